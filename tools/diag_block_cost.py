"""Cost of each 8-row block of the benchmark view on its own, and the rank imbalance (max / mean of the per-rank sums) that dealing the blocks
round robin or in snake order (r, 2N-1-r, 2N+r, ...) would give."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode="bf16x3")
cost = []
for b in range(64):
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, 8 * b, 1, 334, 8, 64, 64); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    cost.append(min(ts))
print("block cost (ms):", " ".join(f"{c:.2f}" for c in cost))
for N in (2, 4, 8):
    rr = [sum(cost[b] for b in range(64) if b % N == r) for r in range(N)]
    sn = [sum(cost[b] for b in range(64) if (b % (2 * N) == r or b % (2 * N) == 2 * N - 1 - r)) for r in range(N)]
    mean = sum(cost) / N
    print(f"N={N}: round robin max/mean {max(rr) / mean:.3f}   snake max/mean {max(sn) / mean:.3f}")
