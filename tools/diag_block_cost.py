"""Cost of each 8-row block of the benchmark view on its own, and the rank imbalance (max / mean of the per-rank sums) that dealing the blocks
round robin or in snake order (r, 2N-1-r, 2N+r, ...) would give."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode="bf16x3")
cost = []
for b in range(64):
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, 8 * b, 1, 334, 8, 64, 64); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    cost.append(min(ts))
print("block cost (ms):", " ".join(f"{c:.2f}" for c in cost))
for N in (2, 4, 8):
    rr = [sum(cost[b] for b in range(64) if b % N == r) for r in range(N)]
    sn = [sum(cost[b] for b in range(64) if (b % (2 * N) == r or b % (2 * N) == 2 * N - 1 - r)) for r in range(N)]
    mean = sum(cost) / N
    print(f"N={N}: round robin max/mean {max(rr) / mean:.3f}   snake max/mean {max(sn) / mean:.3f}")
# proxies a rank can compute without rendering: rays of the block that hit the hands' bounding box; coarse samples that are valid (inside the
# source view and its foreground mask) -- and how well a deal by each proxy balances the MEASURED costs (parallel.deal_blocks)
from vanerf_amd.parallel import deal_blocks
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, device="cuda")
hits = rays["hit"].view(64, -1).float().sum(1).cpu()
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
q_sdf, q_vis, knn = R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts)
_, valid = R.query_samples(w, fdat, pts, q_sdf, q_vis, knn, want_valid=True, raw=True)
vcount = valid.view(64, -1).float().sum(1).cpu()
print("hit rays per block:", " ".join(f"{int(h)}" for h in hits))
print("valid coarse samples per block (k):", " ".join(f"{v / 1e3:.0f}" for v in vcount))
c = torch.tensor(cost, dtype=torch.float64)
A = torch.stack([torch.ones(64, dtype=torch.float64), hits.double(), vcount.double()], 1)
coef = torch.linalg.lstsq(A, c[:, None]).solution.view(-1)
print("least squares: cost ~", [float(x) for x in coef], "* [1, hit rays, valid samples]; residual rms", float(((A @ coef) - c).pow(2).mean().sqrt()))
for N in (2, 4, 8):
    mean = sum(cost) / N
    for name, proxy in (("measured", c), ("hits", 1.0 + hits.double() / 334.0), ("valid", 1.0 + vcount.double() / vcount.mean()), ("fit", A @ coef)):
        a = deal_blocks(proxy, N)
        loads = [sum(cost[b] for b in r if b < 64) for r in a.tolist()]
        print(f"N={N}: dealt by {name:8s} max/mean {max(loads) / mean:.3f}")
