"""Diagnostic: the rays of the orbit cameras of tests/test_novel_views.py::test_render_video_on_hip_path -- only ray_setup and
sample_points run (no mesh / network kernels); prints value ranges and non-finite counts."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
from vanerf_amd.model import get_360cameras
from vanerf_amd.novel_views import camera_to_cam_tar
frame = synth.to_device(synth.make_frame(seed=3, tar_h=256, tar_w=256), "cuda")
tar = frame["cam_tar"]
headpose = torch.inverse(tar["RT"][0])[:3, :4]
trans, sc = 10, 0.1
znear, zfar = (trans - 5.0) * sc, (trans + 5.0) * sc
focal = 256 * 30 + 0.01 * (256 * 0.05 - 256 * 30)
cams = get_360cameras(headpose, focal, trans, sc, 256, 256, znear, zfar, 3)
for i, cam in enumerate(cams):
    ct = camera_to_cam_tar(cam)
    print("camera", i, "RT", [round(float(v), 4) for v in ct["RT"][0].flatten().tolist()], "znear/zfar", ct["znear"], ct["zfar"])
    rays = R.ray_setup(ct, frame["bounds"], 0, 0, 1, 256, 256, 8, device="cuda")
    pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
    torch.cuda.synchronize()
    for k in ("rays_d", "cam_pos", "near", "far", "z"):
        t = rays[k].float()
        print("   ", k, "finite", bool(torch.isfinite(t).all()), "min", float(t[torch.isfinite(t)].min()) if torch.isfinite(t).any() else None,
              "max", float(t[torch.isfinite(t)].max()) if torch.isfinite(t).any() else None, "nonfinite", int((~torch.isfinite(t)).sum()))
    print("    pts nonfinite", int((~torch.isfinite(pts)).sum()), "hit", float(rays["hit"].float().mean()), "bounds", frame["bounds"].flatten().tolist())
