"""How many samples of the benchmark view are invalid (miss the source image), and how many of them sit in all-invalid 32-sample groups."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode="bf16x3")
out = R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, 64, debug=True)
for name in ("coarse", "fine"):
    c = out[name]
    _, valid = R.query_samples(w, fdat, c["pts"], c["q_sdf"].reshape(-1), c["q_vis"], c["knn"], want_valid=True)
    v = valid.bool()
    g = v.view(-1, 32)
    print(f"{name}: samples {v.numel()}, invalid {1 - v.float().mean().item():.3f}, in all-invalid groups {(~g.any(1)).float().mean().item():.3f}, "
          f"in all-valid groups {g.all(1).float().mean().item():.3f}, mixed groups {(g.any(1) & ~g.all(1)).float().mean().item():.3f}")
