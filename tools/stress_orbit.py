"""Stress the orbit path for run-to-run differences: renders the 120-frame orbit of tests/test_novel_views.py repeatedly (frame schedule) and every
frame alone, in both precisions, and reports every frame whose uint8 image or float image differs between repetitions."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import synth
from vanerf_amd.config import default_config
from vanerf_amd.model import VANeRF, get_360cameras
from vanerf_amd.novel_views import camera_to_cam_tar, render_novel_views

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for precision in ("fp32", "bf16x3"):
    torch.manual_seed(0)
    cfg = default_config()
    cfg["models"]["VANeRF"]["dr_kwargs"].update(sample_per_ray_c=16, sample_per_ray_f=16)
    cfg["models"]["VANeRF"]["mfma_precision"] = precision
    net = VANeRF(cfg).cuda().eval()
    net.load_state_dict(synth.make_full_weights(0), strict=False)
    frame_cpu = synth.make_frame(seed=3, tar_h=64, tar_w=64)
    frame = synth.to_device(frame_cpu, "cuda")
    trb = synth.to_tr_batch(frame)
    headpose = torch.eye(4)
    headpose[:3, 3] = frame_cpu["targets"]["vert_world"][0].mean(0)
    cams = get_360cameras(headpose[:3, :4].cuda(), 256.0, 1.0, 1.0, 64, 64, 0.71, 1.42, n_frames=120)
    first, singles0 = None, None
    for r in range(reps):
        rgb, _ = render_novel_views(net, cams, trb, only_renderings=True)
        if first is None:
            first = rgb
        elif not np.array_equal(rgb, first):
            bad = [k for k in range(120) if not np.array_equal(rgb[k], first[k])]
            print(precision, "rep", r, "orbit frames differ from rep 0:", bad)
        singles = []
        for k in range(120):
          with torch.no_grad():
            out = net.render_pifu_nerf(None, net, trb["im"], trb["cam"], trb["hand_type"], trb["targets"], camera_to_cam_tar(cams[k]), level=1,
                                       sp_data=dict(trb["sp_data"]), fine=True, uniform=True, sample_per_ray_c=16, sample_per_ray_f=16,
                                       src_foreground_mask=trb["src_foreground_mask"], bounds=trb["dr_data"]["bounds"], mask_at_box=None)
            singles.append(out["tex_fg_fine"].clone())
        singles = torch.stack(singles)
        if singles0 is None:
            singles0 = singles
        else:
            d = (singles != singles0).flatten(1).any(1).nonzero().flatten().tolist()
            if d:
                print(precision, "rep", r, "single renders differ (float) from rep 0 at frames", d, "max", float((singles - singles0).abs().max()))
        want = (singles.clamp(0, 1).permute(0, 2, 3, 1) * 255.0).to(torch.uint8).cpu().numpy()
        bad = [k for k in range(120) if not np.array_equal(want[k], rgb[k])]
        if bad:
            for k in bad:
                dd = np.abs(want[k].astype(int) - rgb[k].astype(int))
                print(precision, "rep", r, "frame", k, "orbit != single:", int((dd > 0).sum()), "elements, max", int(dd.max()))
    print(precision, "done", reps, "repetitions")
