"""torch.profiler view of one per-source-frame setup (renderer.FrameData): device time per kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fd = synth.to_device(synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0), "cuda")
mk = lambda: R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
for _ in range(3):
    mk()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    mk()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
