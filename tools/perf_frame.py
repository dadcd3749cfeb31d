"""Per-source-frame setup cost (FrameData: vertex tables, TexVisFusion global feature, visibility raster, mesh acceleration structure)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
for it in range(4):
    frame = synth.make_frame(seed=11 + it, tar_h=512, tar_w=334, orbit_deg=15.0)
    fd = synth.to_device(frame, "cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    gf = R.tex_global_vertex_feature({k: v for k, v in sdd.items()}, fd["feat_tex"], fd["img_in"])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    acc = R.MeshAccel(fdat.verts3, fdat.faces)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"frame {it}: FrameData {1e3 * (t1 - t0):.1f} ms (of which global vertex feature {1e3 * (t2 - t1):.1f} ms, mesh accel {1e3 * (t3 - t2):.1f} ms)")
