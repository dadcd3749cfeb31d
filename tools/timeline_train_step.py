"""GPU busy / idle timeline of one training step (tools/perf_train_step.py's step) from the PyTorch profiler's kernel records: total busy time, and
the idle gaps above 100 us with the kernels on either side -- where the device waits for the host."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import perf_train_step as P
from torch.profiler import profile, ProfilerActivity
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    P.step(); P.step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range.end > e.time_range.start]
ev.sort(key=lambda e: e.time_range.start)
# second step only: from the first kernel after the middle of the trace's span ... simpler: split at the largest count boundary -- use the optimizer kernels
t_first, t_last = ev[0].time_range.start, ev[-1].time_range.end
names = [e.name for e in ev]
adam = [i for i, n in enumerate(names) if "multi_tensor_apply" in n or "foreach" in n.lower()]
cut = next(i for i in adam if ev[i].time_range.start > (t_first + t_last) / 2 - (t_last - t_first) / 4)  # first Adam kernel of step 1 (roughly mid-trace)
# the first step ends with its last Adam kernel: find the gap in Adam indices
split = next(adam[k] + 1 for k in range(len(adam) - 1) if adam[k + 1] - adam[k] > 50)
step = ev[split:]
t0, t1 = step[0].time_range.start, step[-1].time_range.end
busy, gaps, cur_end, prev = 0.0, [], step[0].time_range.start, step[0]
for e in step:
    s, en = e.time_range.start, e.time_range.end
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, prev.name[:60], e.name[:60]))
        busy += 0
    busy += max(0.0, en - max(s, cur_end))
    if en > cur_end:
        cur_end, prev = en, e
print(f"second step: first kernel to last kernel {1e-3 * (t1 - t0):.2f} ms, device busy {1e-3 * busy:.2f} ms, idle {1e-3 * (t1 - t0 - busy):.2f} ms in {len(gaps)} gaps, {len(step)} kernels")
small = sum(g[0] for g in gaps if g[0] <= 100)
print(f"gaps of <= 100 us: {1e-3 * small:.2f} ms in all; the larger ones (length us, at ms, after -> before):")
for g in sorted(gaps, reverse=True)[:25]:
    if g[0] > 100:
        print(f"  {g[0]:8.0f} us at {1e-3 * g[1]:6.2f} ms   {g[2]}  ->  {g[3]}")
# ---- the step cut into phases at marker kernels ------------------------------------------------------------------------------------------
def first(pred, start=0):
    return next((i for i in range(start, len(step)) if pred(step[i].name)), None)
def last(pred):
    return next((i for i in range(len(step) - 1, -1, -1) if pred(step[i].name)), None)
marks = [("encoders forward, per-frame setup", 0),
         ("render pass (HIP forward of the patch)", first(lambda n: "fold_kernel" in n)),
         ("loss, backward stage 1 (composites' graph)", last(lambda n: "composite" in n) + 1),
         ("backward stage 2 (fused HIP: spill, chain, products, scatters)", first(lambda n: "query_kernel<0, true>" in n)),
         ("table graph, per-frame stacks, encoders backward", max(last(lambda n: "scatter_rows" in n), last(lambda n: "query_backward" in n)) + 1),
         ("Adam", first(lambda n: "multi_tensor_apply" in n))]
marks = [(n, i) for n, i in marks if i is not None] + [("end", len(step))]
for (name, a), (_, b) in zip(marks, marks[1:]):
    seg = step[a:b]
    if not seg:
        continue
    dur = sum(e.time_range.end - e.time_range.start for e in seg)
    span = seg[-1].time_range.end - seg[0].time_range.start
    by = {}
    for e in seg:
        k = e.name[:70]
        c = by.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += e.time_range.end - e.time_range.start
    print(f"\n{name}: {len(seg)} kernels, busy {1e-3 * dur:.2f} ms, span {1e-3 * span:.2f} ms")
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:(24 if "stage 2" in name else 40 if "render pass" in name else 9)]:
        print(f"    {1e-3 * t:6.2f} ms  {c:4d} x  {k}")
