"""One rank's share of the benchmark view at N ranks (default 8), repeated: the command behind profiles/r01_i_kernel_stats_rank_share_n8.csv
(rocprofv3 --kernel-trace --stats): per-kernel time of what a rank of the N-GPU bench runs per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
from vanerf_amd.parallel import shard_rows
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode="bf16x3")
y0, ys, ny, yb = shard_rows(512, N, rank)
fn = lambda: R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, y0, 1, 334, ny, 64, 64, y_step=ys, y_block=yb)
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    fn()
torch.cuda.synchronize()
print(f"rank {rank} of {N}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per step", flush=True)
