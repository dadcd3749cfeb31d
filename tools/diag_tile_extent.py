"""Extent of the 8x8-pixel x one-depth tiles the mesh query works on (benchmark view): 90 % span < 1.5 cm, the tiles on the silhouette of the
bounding box (rays that hit it next to rays that miss it) up to 0.5 m.  Behind the tile-split experiment recorded in DESIGN.md section 8."""
import sys, torch
sys.path.insert(0, ".")
from vanerf_amd import renderer as R, synth
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"]).view(512, 334, 64, 3)
t = pts[:512, :328].reshape(64, 8, 41, 8, 64, 3).permute(0, 2, 4, 1, 3, 5).reshape(64, 41, 64, 64, 3)
ext = (t.max(3)[0] - t.min(3)[0]).max(-1)[0]
print("tile extent quantiles (m):", [round(float(ext.flatten().quantile(q)), 5) for q in (0.01, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0)])
v = frame["targets"]["vert_world"][0]
print("mesh extent:", (v.max(0)[0] - v.min(0)[0]).tolist())
thr = 0.1 * float(max(v.max(0)[0][1] - v.min(0)[0][1], v.max(0)[0][2] - v.min(0)[0][2]))
print("threshold", thr)
for b in (4, 12, 13, 16, 21, 30, 49):
    e = ext[b]
    print(f"block {b}: fraction of tiles x depths above the threshold {float((e > thr).float().mean()):.3f}, median extent {float(e.median()):.4f}, max {float(e.max()):.3f}")
