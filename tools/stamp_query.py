"""Diagnostic: per-phase cycle shares of query_kernel (needs a library built with VANERF_HIPCC_FLAGS=-DVANERF_STAMPS)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth  # noqa: E402
from vanerf_amd._ffi import lib  # noqa: E402

NAMES = ["front end", "1-NN", "geo0 gathers", "geo0 layers", "geo1", "mlp0 (PE+geo64)", "softplus+mlp1..3", "pool+head", "ibr", "tex", "store", "-"]
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 0
w = R.PackedWeights(sd, mode=MODE)
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
q_sdf, q_vis, knn = R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts)
n = pts.shape[0]
out = torch.empty(n, 5, device="cuda")
nw = ctypes.c_int(0)
qword = torch.zeros(2, dtype=torch.int32, device="cuda")
fn = lib.vanerf_debug_query_stamps
fn.restype = ctypes.c_int
args = lambda st: (w.handle, ctypes.byref(fdat.c), ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(q_sdf.data_ptr()), ctypes.c_void_p(q_vis.data_ptr()), ctypes.c_void_p(knn.data_ptr()),
                   ctypes.c_int64(n), ctypes.c_void_p(out.data_ptr()), st, ctypes.byref(nw), ctypes.c_void_p(qword.data_ptr()), None)
assert fn(*args(None)) == 0
stamps = torch.zeros(nw.value, 12, dtype=torch.int64, device="cuda")
assert fn(*args(ctypes.c_void_p(stamps.data_ptr()))) == 0
torch.cuda.synchronize()
s = stamps.cpu().double()
rt = s[:, 11].clone()
s[:, 11] = 0
tot = s.sum(1)
print(f"in-kernel clock: {(tot / rt).median().item() * 100:.0f} MHz (s_memtime / s_memrealtime x 100 MHz)")
import time
t0 = time.perf_counter(); assert fn(*args(ctypes.c_void_p(stamps.data_ptr()))) == 0; torch.cuda.synchronize(); wall = time.perf_counter() - t0
q = torch.quantile(tot, torch.tensor([0.0, 0.05, 0.5, 0.95, 1.0], dtype=torch.float64))
print("per-wave total cycles min/5%/median/95%/max:", [f"{v:.3e}" for v in q.tolist()], f"wall {wall*1e3:.2f} ms = {wall*2.326e9:.3e} cycles")
per_cu = tot.view(-1, 4).sum(1)
print("per-block totals min/max:", f"{per_cu.min():.3e} {per_cu.max():.3e}")
groups = (n + 31) // 32
print(f"waves {nw.value}, groups/wave {groups / nw.value:.1f}, mean cycles/wave {tot.mean():.3e}, cycles per group per wave {tot.sum() / groups:.0f}")
for k, name in enumerate(NAMES[:11]):
    print(f"  {name:22s} {100 * s[:, k].sum() / tot.sum():6.2f} %   {s[:, k].sum() / groups:9.0f} cycles/group")
