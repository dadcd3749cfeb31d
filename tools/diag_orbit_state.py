"""What changes between two repetitions of (orbit, single renders) on one net?  Snapshots the encoder features and the per-frame tables."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import synth
from vanerf_amd.config import default_config
from vanerf_amd.model import VANeRF, get_360cameras
from vanerf_amd.novel_views import camera_to_cam_tar, render_novel_views
torch.manual_seed(0)
cfg = default_config()
cfg["models"]["VANeRF"]["dr_kwargs"].update(sample_per_ray_c=16, sample_per_ray_f=16)
net = VANeRF(cfg).cuda().eval()
net.load_state_dict(synth.make_full_weights(0), strict=False)
frame_cpu = synth.make_frame(seed=3, tar_h=64, tar_w=64)
frame = synth.to_device(frame_cpu, "cuda")
trb = synth.to_tr_batch(frame)
headpose = torch.eye(4)
headpose[:3, 3] = frame_cpu["targets"]["vert_world"][0].mean(0)
cams = get_360cameras(headpose[:3, :4].cuda(), 256.0, 1.0, 1.0, 64, 64, 0.71, 1.42, n_frames=8)
def snap(tag):
    s = {}
    ec = net._enc_cache
    if ec is not None:
        s["enc_key_im"] = ec[0][:3]
        s["feat_geo0"], s["feat_geo1"], s["feat_tex"] = ec[1][0].clone(), ec[1][1].clone(), ec[2].clone()
    fc = net._frame_cache
    if fc is not None:
        for n in dir(fc[1]):
            v = getattr(fc[1], n)
            if isinstance(v, torch.Tensor): s["fd." + n] = v.clone()
        s["fd_id"] = id(fc[1])
    pw = net._packed
    s["packed_id"] = id(pw[1]) if pw else None
    return s
def diff(a, b, tag):
    for k in a:
        if k not in b: continue
        if isinstance(a[k], torch.Tensor):
            if a[k].shape != b[k].shape or not torch.equal(a[k], b[k]):
                print(tag, k, "differs", float((a[k].double() - b[k].double()).abs().max()) if a[k].shape == b[k].shape else "shape")
        elif a[k] != b[k]:
            print(tag, k, a[k], "->", b[k])
def single(k):
    out = net.render_pifu_nerf(None, net, trb["im"], trb["cam"], trb["hand_type"], trb["targets"], camera_to_cam_tar(cams[k]), level=1,
                               sp_data=dict(trb["sp_data"]), fine=True, uniform=True, sample_per_ray_c=16, sample_per_ray_f=16,
                               src_foreground_mask=trb["src_foreground_mask"], bounds=trb["dr_data"]["bounds"], mask_at_box=None)
    return out["tex_fg_fine"].clone()
prev, prev_s = None, None
for r in range(4):
    rgb, _ = render_novel_views(net, cams, trb, only_renderings=True)
    s1 = snap("orbit")
    sg = torch.stack([single(k) for k in range(8)])
    s2 = snap("single")
    diff(s1, s2, f"rep {r} orbit->single:")
    if prev_s is not None: diff(prev_s, s2, f"rep {r-1}->{r} after singles:")
    if prev is not None and not torch.equal(prev, sg): print(f"rep {r}: singles differ from rep {r-1}", float((prev - sg).abs().max()))
    prev, prev_s = sg, s2
print("done")
