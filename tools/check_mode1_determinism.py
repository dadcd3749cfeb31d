"""Mode-1 (split-bf16) run-to-run determinism and parity against mode 0 over repeated full-size launches (env VANERF_BLOCKS_PER_CU)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
q_sdf, q_vis, knn = R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts)
ref = R.query_samples(R.PackedWeights(sd, mode=0), fdat, pts, q_sdf, q_vis, knn).clone()
w = R.PackedWeights(sd, mode=1)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
first, ndiff, worst = None, [], 0.0
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ms = []
for i in range(runs):
    ev[0].record(); o = R.query_samples(w, fdat, pts, q_sdf, q_vis, knn); ev[1].record(); torch.cuda.synchronize()
    ms.append(ev[0].elapsed_time(ev[1]))
    worst = max(worst, float((o - ref).abs().max()))
    if first is None: first = o.clone()
    else: ndiff.append(int((o != first).any(1).sum()))
print("blocks/CU", os.environ.get("VANERF_BLOCKS_PER_CU", "default"), "runs", runs, "samples differing from run 0:", ndiff,
      "max |mode1 - mode0|", worst, "ms min", min(ms))
