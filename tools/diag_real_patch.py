import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import synth
from vanerf_amd.config import default_config
from vanerf_amd.model import VANeRF
KEYS = ("tex_fg", "depth", "alpha", "tex_fg_fine", "depth_fine", "alpha_fine", "sdf")
frame = synth.to_device(synth.make_frame(seed=3, tar_h=256, tar_w=256), "cuda")
res = []
for hip in (True, False, True):
    torch.manual_seed(0)
    cfg = default_config()
    cfg["models"]["VANeRF"]["dr_kwargs"].update(sample_per_ray_c=64, sample_per_ray_f=64, rand_noise_std=0.01, uniform=False, fine=True)
    net = VANeRF(cfg).cuda().train()
    net.load_state_dict(synth.make_full_weights(0), strict=False)
    net.kwargs["hip_backward"] = hip
    dr = {"img": frame["img_in"], "cam": frame["cam_in"], "cam_tar": frame["cam_tar"], "tar": torch.rand(1, 3, 256, 256, device="cuda"),
          "msk": torch.ones(1, 1, 256, 256, device="cuda")}
    torch.manual_seed(5); np.random.seed(5)
    out = net(frame["img_in"], frame["cam_in"], frame["hand_type"], frame["targets"], None, None, n_views=1, sp_data=dict(frame["sp_data"]),
              dr_data=dr, src_foreground_mask=frame["src_foreground_mask"], bounds=frame["bounds"])["out"]["nerf"]
    g = torch.Generator().manual_seed(1)
    sum((out[k] * torch.randn(out[k].shape, generator=g).cuda()).sum() for k in KEYS).backward()
    res.append({k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()})
    del net, out
for name, (a, b) in (("hip vs torch", (res[0], res[1])), ("hip vs hip again", (res[0], res[2]))):
    print(name)
    for k in a:
        if a[k] is None or b[k] is None or k.startswith(("geo_encoder", "tex_encoder")): continue
        na = max(a[k].norm().item(), b[k].norm().item())
        d = (a[k] - b[k]).norm().item()
        if d > 1e-3 * na + 1e-3: print(f"   {k:55s} rel {d / max(na, 1e-30):.2e}  norm {na:.3g}")
if len(sys.argv) > 1:
    keep = lambda r: {k: v.cpu() for k, v in r.items() if v is not None and not k.startswith(("geo_encoder", "tex_encoder"))}
    torch.save({"hip": keep(res[0]), "torch": keep(res[1])}, sys.argv[1])
