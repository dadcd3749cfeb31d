"""Cost of one rank's share of the benchmark view for different row-sharding schemes (run on one GPU): full step time of the rows that
rank 0 of N would render, row-interleaved (rows r, r+N, ...) against contiguous blocks of rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode="bf16x3")
def timed(fn, n=4):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
RP = R.render_pass_c if "--c" in sys.argv else R.render_pass  # --c: the one-call C entry point (one ctypes call per pass)
full = timed(lambda: RP(w, fdat, frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, 64))
print(f"full view: {full:.2f} ms")
for N in (2, 4, 8):
    ny = 512 // N
    inter = [timed(lambda r=r: RP(w, fdat, frame["cam_tar"], frame["bounds"], 0, r, 1, 334, ny, 64, 64, y_step=N)) for r in range(N)]
    block = [timed(lambda r=r: RP(w, fdat, frame["cam_tar"], frame["bounds"], 0, r * ny, 1, 334, ny, 64, 64)) for r in range(N)]
    from vanerf_amd.parallel import shard_rows
    dealt = []
    for r in range(N):
        y0, ys, n, yb = shard_rows(512, N, r)
        dealt.append(timed(lambda: RP(w, fdat, frame["cam_tar"], frame["bounds"], 0, y0, 1, 334, n, 64, 64, y_step=ys, y_block=yb)))
    print(f"N={N}: ideal {full / N:.2f} ms | single interleaved rows max {max(inter):.2f} | contiguous blocks max {max(block):.2f} min {min(block):.2f} | "
          f"blocks of 8 rows dealt round robin (shard_rows) max {max(dealt):.2f} min {min(dealt):.2f}")
