"""The evaluation view through the model interface (render_pifu_nerf, 512x334, 64 + 64 samples, bf16x3) against the bare render pass."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
from vanerf_amd.config import default_config
from vanerf_amd.model import VANeRF
torch.manual_seed(0)
cfg = default_config()
cfg["models"]["VANeRF"]["mfma_precision"] = "bf16x3"
net = VANeRF(cfg).cuda().eval()
net.load_state_dict(synth.make_full_weights(0), strict=False)
frame = synth.to_device(synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0), "cuda")
trb = synth.to_tr_batch(frame)
def call():
    with torch.no_grad():
        return net.render_pifu_nerf(None, net, trb["im"], trb["cam"], trb["hand_type"], trb["targets"], frame["cam_tar"], level=1, sp_data=dict(trb["sp_data"]),
                                    fine=True, uniform=True, sample_per_ray_c=64, sample_per_ray_f=64, src_foreground_mask=trb["src_foreground_mask"],
                                    bounds=trb["dr_data"]["bounds"], mask_at_box=None)
for _ in range(3):
    call()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    out = call()
torch.cuda.synchronize()
print(f"render_pifu_nerf 512x334: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per view; keys {sorted(out)[:6]}...", flush=True)
