"""Which stage of a pass differs from run to run?  Same inputs, repeated: FrameData tables, render_pass_c outputs, render_pass (Python sequence) intermediates."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=3, tar_h=64, tar_w=64)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
def mk(): return R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
f0 = mk()
names = [n for n in dir(f0) if isinstance(getattr(f0, n), torch.Tensor)]
for r in range(5):
    f1 = mk()
    bad = [n for n in names if getattr(f1, n).shape == getattr(f0, n).shape and not torch.equal(getattr(f1, n), getattr(f0, n))]
    if bad: print("FrameData rebuild", r, "differs in", bad, [float((getattr(f1, n).float() - getattr(f0, n).float()).abs().max()) for n in bad])
w = R.PackedWeights(sd, mode=prec)
first = None
for r in range(reps):
    o = R.render_pass_c(w, f0, frame["cam_tar"], frame["bounds"], 0, 0, 1, 64, 64, 16, 16)
    torch.cuda.synchronize()
    o = {k: v.clone() for k, v in o.items()}
    if first is None: first = o
    else:
        bad = {k: float((o[k].double() - first[k].double()).abs().max()) for k in o if not torch.equal(o[k], first[k])}
        if bad: print("render_pass_c rep", r, "differs:", bad)
first = None
for r in range(reps):
    o = R.render_pass(w, f0, frame["cam_tar"], frame["bounds"], 0, 0, 1, 64, 64, 16, 16, debug=True)
    torch.cuda.synchronize()
    flat = {}
    for k, v in o.items():
        if isinstance(v, torch.Tensor): flat[k] = v.clone()
        elif isinstance(v, dict):
            for kk, vv in v.items():
                if isinstance(vv, torch.Tensor): flat[k + "." + kk] = vv.clone()
    if first is None: first = flat
    else:
        bad = {k: float((flat[k].double() - first[k].double()).abs().max()) for k in flat if flat[k].shape == first[k].shape and not torch.equal(flat[k], first[k])}
        if bad: print("render_pass rep", r, "differs:", bad)
print("done", prec, reps)
