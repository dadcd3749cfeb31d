"""Where the per-source-frame setup (renderer.FrameData) spends its time: wall time of its parts, steady state."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
mk = lambda: R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
f0 = mk()
def timed(name, fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name:36s} {1e3 * min(ts):7.3f} ms")
timed("FrameData (all)", mk)
timed("  tex_global_vertex_feature", lambda: R.tex_global_vertex_feature(sdd, fd["feat_tex"], fd["img_in"]))
timed("  vertex_visibility (raster)", lambda: R.vertex_visibility(f0.vert_xy01, f0.vert_z01, f0.faces))
timed("  MeshAccel", lambda: R.MeshAccel(f0.verts3, f0.faces))
