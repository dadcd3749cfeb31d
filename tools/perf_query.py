"""Times the dominant kernel (query_kernel) alone on the coarse samples of the benchmark view (512x334 x 64 = 10.9 M samples).
Used for A/B runs of kernel variants and as the command under `rocprofv3 --pmc`."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--rows", type=int, default=512)
ap.add_argument("--mode", type=int, default=0, help="0 = fp32 MFMA, 1 = split-bf16 x3")
ap.add_argument("--order", type=int, default=1, help="1: work order from the validity partition (what render_pass does), 0: natural order")
args = ap.parse_args()
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode=args.mode)
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, args.rows, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
q_sdf, q_vis, knn = R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts)
order = R.query_order(fdat, pts) if args.order else None
out = R.query_samples(w, fdat, pts, q_sdf, q_vis, knn, order=order)
torch.cuda.synchronize()
ts = []
for _ in range(args.iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = R.query_samples(w, fdat, pts, q_sdf, q_vis, knn, order=order)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
n = pts.shape[0]
best = min(ts)
short = 32 * w.short_groups() / (args.iters + 1) / n
flop = n * ((1 - short) * 287544 + short * 51840)
if args.mode:
    ref = R.query_samples(R.PackedWeights(sd, mode=0), fdat, pts, q_sdf, q_vis, knn)
    print("mode", args.mode, "max |diff| vs fp32 mode [alpha, sdf, r, g, b]:", [f"{v:.2e}" for v in (out - ref).abs().max(0)[0].tolist()])
print(f"samples {n}  ms min {best:.3f} med {sorted(ts)[len(ts)//2]:.3f}  short-path {short:.3f}  TFLOP/s(alg) {flop / best / 1e9:.2f}  frac {flop / best / 1e9 / 157.3:.3f}  checksum {out.double().sum().item():.6f}")
