"""mesh_query_accel_kernel hands its work out through an atomic queue: which wave searches which tile differs from launch to launch, the results may
not.  N launches over the coarse samples of the benchmark view, every output compared bit for bit with the first launch's."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
ref = [t.clone() for t in R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts, want_face=True, grid=(334, 512, 64))]
bad = 0
for i in range(N):
    out = R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts, want_face=True, grid=(334, 512, 64))
    bad += sum(int(not torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)) for a, b in zip(out, ref))
torch.cuda.synchronize()
print(f"{N} launches over {pts.shape[0]} points: {bad} outputs differ from the first launch's (sdf, visibility, face, 1-NN compared bit for bit)")
