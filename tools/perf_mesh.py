"""Times mesh_query_accel_kernel alone on the coarse samples of the benchmark view."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth  # noqa: E402

sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
NY = 512
if "--share" in sys.argv:  # rank 0's rows of an N-GPU run (parallel.shard_rows): how the kernel does on a small launch
    from vanerf_amd.parallel import shard_rows
    y0, ys, NY, yb = shard_rows(512, int(sys.argv[sys.argv.index("--share") + 1]), 0)
    rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, y0, 1, 334, NY, 64, device="cuda", y_step=ys, y_block=yb)
else:
    rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
for grid in ((334, NY, 64),) if "--hint-only" in sys.argv else ((334, NY, 64), None):
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts, grid=grid)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"grid hint {grid}: ms min {min(ts):.3f}  hit-bbox fraction {rays['hit'].float().mean().item():.3f}  inside {float((out[0] < 0).float().mean()):.4f}")

if "--phases" in sys.argv:  # needs VANERF_HIPCC_FLAGS=-DVANERF_MESH_PHASES
    import ctypes
    from vanerf_amd._ffi import lib
    buf = (ctypes.c_uint64 * 16)()
    lib.vanerf_debug_mesh_phases.restype = ctypes.c_int
    lib.vanerf_debug_mesh_phases(buf, 1)
    R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts, grid=(334, NY, 64)); torch.cuda.synchronize()
    lib.vanerf_debug_mesh_phases(buf, 1)
    tot = sum(buf[:5])
    for name, v in zip(["point load", "1-NN vertex", "closest face", "inside test", "visibility + stores"], buf):
        print(f"  {name:20s} {100.0 * v / tot:5.1f} %  {v / (pts.shape[0] / 64):9.0f} cycles per 64 points")
    nw = pts.shape[0] / 64
    print(f"  per wave of 64 points (per-lane search): {buf[5] / nw:.1f} clusters pass the wave-level test, {buf[6] / nw:.1f} are opened by some lane, {buf[7] / nw:.1f} exact evaluations")
    nt = max(1, buf[8])
    print(f"  closest face: {buf[8]} waves by the tile search ({buf[10] / nt:.0f} cycles each up to here, {buf[12] / nt:.1f} clusters listed, {buf[11] / nt:.1f} per-lane evaluations), {buf[9]} waves by the per-lane search (tile too wide: {buf[13]}, cluster list full: {buf[14]}, candidate table full: {buf[15]})")
