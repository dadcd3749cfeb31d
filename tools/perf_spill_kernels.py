"""Launch times of the fused backward's two kernels (vanerf_query_forward_spill, vanerf_query_backward) and of the weight products on one block of
samples of the benchmark frame (HIP events, 10 launches each).  usage: perf_spill_kernels.py [n]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth, hip_backward as HB
from vanerf_amd._ffi import lib, check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
w0 = R.PackedWeights(sd, mode="fp32")
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 200, 1, 334, 64, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"]).view(-1, 3)[:n].contiguous()
q_sdf, q_vis, knn = (t.view(-1) for t in R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts))
d = torch.randn(n, 5, device="cuda")
ws = HB.workspace(n, pts.device)
npad = ws.block
P = HB._ptr
st = R._stream()
qw = R._queue_word(pts.device)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def fwd():
    qw.zero_()
    check(lib.vanerf_query_forward_spill(w0.handle, ctypes.byref(fdat.c), P(pts), P(q_sdf), P(q_vis), P(knn), n, npad, P(ws.raw), P(ws.valid), P(ws.xs), P(ws.aux), P(qw), st))
def bwd():
    check(lib.vanerf_query_backward(w0.handle, P(d), None, None, None, P(ws.raw), P(ws.valid), n, npad, P(ws.xs), P(ws.aux), P(ws.ys), P(ws.ig), st))
print(f"n = {n}: forward spill {timed(fwd):.3f} ms   backward chain {timed(bwd):.3f} ms   weight products {timed(lambda: HB._weight_products_on(ws, ws.xs, ws.ys, npad)):.3f} ms")
