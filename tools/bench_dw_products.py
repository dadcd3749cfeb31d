"""The weight products of one block of the fused backward: vanerf_weight_products (one launch, all layers) against the sliced torch.baddbmm
path it replaced (one launch per layer) and an unsliced product, HIP events.  usage: bench_dw_products.py [npad]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import hip_backward as HB
npad = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
if len(sys.argv) > 2:
    HB.SLICES = int(sys.argv[2])  # the block's samples are cut into this many separately accumulated parts
ws = HB.Workspace(npad, "cuda")
ws.xs.normal_(); ws.ys.normal_()
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
L = HB.layout()
flops = 2.0 * npad * sum(l["n_out"] * l["n_slots"] for l in L["layers"])
byts = 4.0 * npad * (L["x_rows"] + L["y_rows"])
t_k = timed(lambda: HB._weight_products_on(ws, ws.xs, ws.ys, npad))
t_t = timed(lambda: HB._weight_products_on(ws, ws.xs, ws.ys, npad, use_torch=True))
def unsliced():
    for lay in L["layers"]:
        ws.ys[lay["y_row"]:lay["y_row"] + lay["n_out"]] @ ws.xs[lay["x_row"]:lay["x_row"] + lay["n_slots"]].t()
t_u = timed(unsliced)
print(f"npad {npad}, {ws.slices} slices: vanerf_weight_products {t_k:.3f} ms ({flops / t_k / 1e9:.1f} TFLOP/s fp32, {byts / t_k / 1e9:.2f} TB/s of unique operand bytes)   "
      f"sliced baddbmm x 20 {t_t:.3f} ms   unsliced matmul x 20 {t_u:.3f} ms")
