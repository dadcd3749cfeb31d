"""Host time to enqueue one render pass (no synchronisation) against its GPU time, for the full view and for one rank's share of 8:
at N = 8 a rank's GPU work is ~3.4 ms per step, so the Python / launch path must stay well below that."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
from vanerf_amd.parallel import shard_rows
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w = R.PackedWeights(sd, mode="bf16x3")
for N in (1, 8):
    y0, ys, ny, yb = shard_rows(512, N, 0)
    fn = lambda: R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, y0, 1, 334, ny, 64, 64, y_step=ys, y_block=yb)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms per pass, wall {1e3 * (t2 - t0) / n:.3f} ms per pass", flush=True)

# where the host time goes: per-call host time of the renderer's entry points over 20 un-synchronised full-view passes
import collections
acc = collections.defaultdict(float)
def wrap(name):
    f = getattr(R, name)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[name] += time.perf_counter() - t
        return r
    setattr(R, name, g)
for nm in ("ray_setup", "sample_points", "mesh_query_accel", "query_order", "query_samples", "composite", "importance_merge", "composite_merged"):
    wrap(nm)
torch.cuda.synchronize()
for _ in range(20):
    R.render_pass(w, fdat, frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, 64)
torch.cuda.synchronize()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} {1e3 * v / 20:8.3f} ms host per pass", flush=True)
