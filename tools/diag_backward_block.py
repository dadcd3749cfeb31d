"""One block of the fused backward on the benchmark frame's coarse samples: per-layer norms of the spilled dY and of the weight products, and a
checksum of the input gradients -- for A/B runs of kernel builds (same inputs, same seeds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth, hip_backward as HB
torch.backends.cudnn.enabled = False  # (MIOpen picks convolution algorithms per process: the per-frame tables would differ in the last bits from run to run)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w0 = R.PackedWeights(sd, mode="fp32")
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 200, 1, 334, 64, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"]).view(-1, 3)[:n].contiguous()
q_sdf, q_vis, knn = R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts)
g = torch.Generator(device="cuda").manual_seed(0)
d = torch.randn(n, 5, device="cuda", generator=g)
ws = HB.workspace(n, pts.device)
ws.dw.zero_()
ig, nb = HB.run_block(ws, w0, fdat, pts, q_sdf.view(-1), q_vis.view(-1), knn.view(-1), d)
torch.cuda.synchronize()
L = HB.layout()
for li, lay in enumerate(L["layers"]):
    ys = ws.ys[lay["y_row"]:lay["y_row"] + lay["n_out"], :n]
    print(f"layer {li:2d}  |dY| {ys.double().norm().item():.9e}  |dW| {ws.dw_l[li].sum(0).double().norm().item():.9e}")
print("valid", int(ws.valid[:n].sum()), " |ig| %.9e" % ws.ig.view(-1)[: (ws.ig.numel() // ws.block) * 0 + ws.ig.numel()].double().norm().item())
# ---- poisoned workspace: every value the backward reads must have been written by the forward spill of the same block ----------------------
ref_ys, ref_dw = ws.ys.clone(), ws.dw.clone()
for poison in (float("nan"), 1e30):
    for t in (ws.xs, ws.aux, ws.ys, ws.ig, ws.raw):
        t.fill_(poison)
    ws.valid.fill_(255)
    ws.dw.zero_()
    HB.run_block(ws, w0, fdat, pts, q_sdf.view(-1), q_vis.view(-1), knn.view(-1), d)
    torch.cuda.synchronize()
    bad_y = (~torch.isfinite(ws.ys[:, :n])).any(1).nonzero().view(-1).tolist()
    print(f"poison {poison}: non-finite dY rows {bad_y[:20]}; dY equal to the first run: {torch.equal(ws.ys[:, :n], ref_ys[:, :n])}; dW equal: {torch.equal(ws.dw, ref_dw)}")
    if not torch.equal(ws.ys[:, :n], ref_ys[:, :n]):
        diff = (ws.ys[:, :n] != ref_ys[:, :n]).any(1).nonzero().view(-1)
        print("   rows that differ:", diff.tolist()[:40], " columns (first):", (ws.ys[:, :n] != ref_ys[:, :n]).any(0).nonzero().view(-1)[:10].tolist())
if len(sys.argv) > 2:
    m = 8192
    lay = L["layers"]
    torch.save({"ys17": ref_ys[lay[17]["y_row"]:lay[17]["y_row"] + 6, :m].cpu(), "ys16": ref_ys[lay[16]["y_row"]:lay[16]["y_row"] + 96, :m].cpu(),
                "ys18": ref_ys[lay[18]["y_row"]:lay[18]["y_row"] + 96, :m].cpu(),
                "aux": ws.aux[:, :m].cpu(), "xs16": ws.xs[lay[16]["x_row"]:lay[16]["x_row"] + lay[16]["n_slots"], :m].cpu(),
                "xs17": ws.xs[lay[17]["x_row"]:lay[17]["x_row"] + lay[17]["n_slots"], :m].cpu(), "d": d[:m].cpu(), "valid": ws.valid[:m].cpu(), "ys17_all": ref_ys[lay[17]["y_row"]:lay[17]["y_row"] + 6, :n].cpu(), "valid_all": ws.valid[:n].cpu(),
                "aux_tex_all": ws.aux[12:20, :n].cpu()}, sys.argv[2])
