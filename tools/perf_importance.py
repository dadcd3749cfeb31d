"""Times vanerf_importance_merge + vanerf_composite_merged at several samples-per-ray counts (171 008 rays)."""
import sys
import torch
sys.path.insert(0, ".")
from vanerf_amd import renderer as R

dev = torch.device("cuda:0")
Rn = 512 * 334
g = torch.Generator(device="cpu").manual_seed(0)
for S in (64, 128, 256):
    contrib = (torch.rand(Rn, S, generator=g) ** 6).to(dev)
    z = torch.sort(torch.rand(Rn, S, generator=g) * 0.3 + 0.8, -1)[0].to(dev)
    for _ in range(3):
        R.importance_merge(contrib, z, S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        R.importance_merge(contrib, z, S)
    e1.record()
    torch.cuda.synchronize()
    print(f"S={S}: importance_merge {e0.elapsed_time(e1) / 20:.3f} ms (includes the output allocations)", flush=True)
