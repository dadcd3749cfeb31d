"""torch.profiler view of rank 0's share of the benchmark view at N = 8 (rows dealt by parallel.shard_rows) beside the full view: device time per kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from vanerf_amd import renderer as R, synth
from vanerf_amd.parallel import shard_rows
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode="bf16x3")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
y0, ys, n, yb = shard_rows(512, N, 0)
share = lambda: R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, y0, 1, 334, n, 64, 64, y_step=ys, y_block=yb)
full = lambda: R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, 64)
for name, fn in (("share", share), ("full", full)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    print(name)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
