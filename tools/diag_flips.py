import sys; sys.path.insert(0,'.')
import torch
from oracle import vanerf_oracle as orc
from vanerf_amd import synth, renderer as R
from tests.conftest import load_golden
sd = synth.make_full_weights(0)
g = load_golden("pass_64x64_s64")
frame = synth.make_frame(seed=11, tar_h=256, tar_w=256, orbit_deg=15.0)
ref = orc.batch_render(sd, frame, 3, torch.tensor([[[0,0]]]), 64, 64)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd)
out = R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, 0, 4, 64, 64, 64, 64, debug=True)
def cmp(a, b, name):
    e=(a-b).abs(); print(name, "max", e.max().item(), "n>1e-4", int((e>1e-4).sum()), "n>1e-5", int((e>1e-5).sum()))
gf = out["color_fine"].cpu().view(64,64,3).permute(2,0,1)
cmp(ref["tex_fg_fine"][0], g["tex_fg_fine"][0], "oracle(box) vs golden fine")
cmp(gf, g["tex_fg_fine"][0], "gpu vs golden fine")
cmp(gf, ref["tex_fg_fine"][0], "gpu vs oracle(box) fine")
gc = out["color"].cpu().view(64,64,3).permute(2,0,1)
cmp(ref["tex_fg"][0], g["tex_fg"][0], "oracle(box) vs golden coarse")
cmp(gc, ref["tex_fg"][0], "gpu vs oracle(box) coarse")
# sample-level: coarse points / sdf / vis / knn
pc = out["coarse"]["pts"].cpu(); cmp(pc, ref["coarse"]["pts"][0], "coarse pts")
print("coarse pts bit-equal frac", (pc==ref["coarse"]["pts"][0]).all(-1).float().mean().item())
print("q_vis mismatch", (out["coarse"]["q_vis"].cpu().bool()!=ref["coarse"]["q_vis"][0,:,0]).sum().item(), "sdf sign mismatch", ((out["coarse"]["q_sdf"].cpu().view(-1)<0)!=(ref["coarse"]["q_sdf"].view(-1)<0)).sum().item())
cmp(out["coarse"]["q_sdf"].cpu().view(-1), ref["coarse"]["q_sdf"].view(-1), "coarse sdf")
e=(out["coarse"]["rgba"].cpu().view(-1,5)-ref["coarse"]["rgba"].view(-1,5)).abs()
print("coarse rgba: n samples >1e-4:", int((e.max(1)[0]>1e-4).sum()), "of", e.shape[0], "max", e.max().item())
zf=out["z_fine"].cpu(); cmp(zf, ref["z_fine"][0], "z_fine")
