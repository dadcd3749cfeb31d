"""mesh_query_accel_kernel on small launches (a few rows of the benchmark view): the time of a launch that gives every wave at most one or two
items is the latency of a single item plus the cold start of the kernel (instruction fetch, table staging) -- what bounds an N-GPU rank's share."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
for ny, y0 in ((8, 0), (8, 250), (16, 250), (32, 250), (64, 224)):
    rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, y0, 1, 334, ny, 64, device="cuda")
    pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts, grid=(334, ny, 64)); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"rows {y0}..{y0+ny}: {pts.shape[0]} points, {pts.shape[0]//128} items: {min(ts):.3f} ms, hit {rays['hit'].float().mean().item():.2f}")
if "--phases" in sys.argv:  # needs a -DVANERF_MESH_PHASES build (VANERF_HIP_LIB)
    import ctypes
    from vanerf_amd._ffi import lib
    buf = (ctypes.c_uint64 * 16)()
    lib.vanerf_debug_mesh_phases.restype = ctypes.c_int
    for ny, y0 in ((8, 0), (8, 250)):
        rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, y0, 1, 334, ny, 64, device="cuda")
        pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
        lib.vanerf_debug_mesh_phases(buf, 1)
        R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts, grid=(334, ny, 64)); torch.cuda.synchronize()
        lib.vanerf_debug_mesh_phases(buf, 1)
        nw = pts.shape[0] / 128
        print(f"rows {y0}: per item: point load {buf[0]/nw:.0f}, 1-NN {buf[1]/nw:.0f}, face {buf[2]/nw:.0f}, inside {buf[3]/nw:.0f}, vis+store {buf[4]/nw:.0f} | tile waves {buf[8]}, per-lane waves {buf[9]} "
              f"(too wide {buf[13]}, list full {buf[14]}, cand batches {buf[15]}), listed {buf[12]/max(1,buf[8]):.1f}, per-lane evals {buf[11]/max(1,buf[8]):.1f}")
