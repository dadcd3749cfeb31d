"""Where the expensive 8-row blocks of the benchmark view spend their time (tools/diag_block_cost.py): mesh query and per-sample networks of
the coarse pass, timed separately per block, with the fraction of rays that hit the bounding box and of samples that are valid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode="bf16x3")
def timed(fn):
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), r
for b in (4, 11, 12, 13, 14, 15, 16, 21, 30, 46, 49, 50):
    rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 8 * b, 1, 334, 8, 64, device="cuda")
    pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
    tm, (q_sdf, q_vis, knn) = timed(lambda: R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts, grid=(334, 8, 64)))
    tq, out = timed(lambda: R.query_samples(w, fdat, pts, q_sdf, q_vis, knn))
    valid = out[1].float().mean().item() if isinstance(out, tuple) else float("nan")
    print(f"block {b:2d}: mesh {tm:.3f} ms  networks {tq:.3f} ms  hit {rays['hit'].float().mean().item():.2f}  valid {valid:.2f}", flush=True)
