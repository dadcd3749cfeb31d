"""Orbit throughput (BASELINE config 4's caller, render_novel_views): frames per second of a 360-degree orbit of 256x256 views at 64 + 64
samples through the model interface, against the GPU time of the bare render passes of the same cameras."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
from vanerf_amd.config import default_config
from vanerf_amd.model import VANeRF, get_360cameras
from vanerf_amd.novel_views import camera_to_cam_tar, render_novel_views

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(0)
cfg = default_config()
cfg["models"]["VANeRF"]["mfma_precision"] = "bf16x3"
net = VANeRF(cfg).cuda().eval()
net.load_state_dict(synth.make_full_weights(0), strict=False)
frame = synth.to_device(synth.make_frame(seed=3, tar_h=256, tar_w=256), "cuda")
trb = synth.to_tr_batch(frame)
tar = frame["cam_tar"]
headpose = torch.inverse(tar["RT"][0])[:3, :4]
dist = float(tar["RT"][0][:3, 3].norm())
cams = get_360cameras(headpose, float(tar["K"][0, 0, 0]), dist, 1.0, 256, 256, tar["znear"], tar["zfar"], n_frames=n_frames)
render_novel_views(net, cams[:3], trb, only_renderings=True)  # warm-up: encoders, MIOpen solvers, frame tables
torch.cuda.synchronize()
t0 = time.perf_counter()
rgb, _ = render_novel_views(net, cams, trb, only_renderings=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"render_novel_views: {n_frames} frames of 256x256 in {dt:.3f} s = {n_frames / dt:.1f} frames/s ({1e3 * dt / n_frames:.2f} ms per frame)", flush=True)
# the bare passes of the same cameras
fd = net.frame_data(trb["im"], trb["cam"], trb["targets"], net.attach_geo_feat(trb["im"], return_val=True), net.attach_tex_feat(trb["im"], return_val=True),
                    trb["sp_data"], trb["src_foreground_mask"])
w = net.packed_weights()
cts = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in camera_to_cam_tar(c).items()} for c in cams]
bounds = trb["dr_data"]["bounds"].cpu()
for ct in cts[:3]:
    R.render_pass(w, fd, ct, bounds, 0, 0, 1, 256, 256, 64, 64)
torch.cuda.synchronize()
t0 = time.perf_counter()
for ct in cts:
    R.render_pass(w, fd, ct, bounds, 0, 0, 1, 256, 256, 64, 64)
torch.cuda.synchronize()
dt2 = time.perf_counter() - t0
print(f"bare render passes:  {1e3 * dt2 / n_frames:.2f} ms per frame", flush=True)
