"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch."""
import collections
import csv
import sys

for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "query_kernel" not in k and "mesh_query" not in k:
            continue
        print(path, k)
        for c, v in cs.items():
            print(f"   {c:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
