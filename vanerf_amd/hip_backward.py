"""Host side of the fused HIP backward pass of the per-sample networks (csrc/query_backward.hip).

For a block of samples two launches do what `torch_graph.networks_at` + `torch.autograd.grad` did in ~2 000 eager kernels:
`vanerf_query_forward_spill` (the fp32 forward once more, spilling every layer's operands) and `vanerf_query_backward` (dX = W^T dY through
all twenty layers in registers, spilling every layer's dY and the gradients of the gathered inputs).  What is left for this file:
* the weight gradients, one batched matrix product per layer over the block's samples, dW'[out][slot] = Ys_l Xs_l^T, accumulated over the
  blocks in ONE flat buffer and mapped to the reference's parameters at the end (slot -> input channel: `vanerf_layer_slots`; weight-norm
  through autograd of the fold itself) -- sums in a fixed order, so the gradient is reproducible run to run;
* the scatters of the input gradients into the feature maps and the per-vertex tables (`vanerf_scatter_add_rows`);
* eval_func's derivative between the two launches (elementwise on (n, 5)).
Reference: the autograd of VANeRF.query / query_color (src/model.py:748-957) inside training_step (src/model.py:381-459)."""
import ctypes

import torch

from . import renderer as R
from ._ffi import NUM_LAYERS, check, lib

NUM_V = 779
SLICES = 64  # the reduction over a block's samples is cut into this many slices (one batched GEMM: the output is tiny, see torch_graph._Linear)

# layer -> parameters of the reference module: ("conv", key, rows) bias-free Conv1d(k = 1) whose first `rows` output channels the kernel uses;
# ("wn", prefix) weight-normed Linear (weight_v, weight_g, bias); ("lin", prefix) plain Linear (weight, bias)
LAYER_PARAMS = [
    ("conv", "geo_vis_fusion.fconv_at.0.weight"), ("conv", "geo_vis_fusion.fconv_at.2.weight"),
    ("conv", "geo_vis_fusion.fconv_ated.0.weight"), ("conv", "geo_vis_fusion.fconv_ated.2.weight"),
    ("conv", "geo_vis_fusion.fconv_at1.0.weight"), ("conv", "geo_vis_fusion.fconv_at1.2.weight"),
    ("conv", "geo_vis_fusion.fconv_ated1.0.weight"), ("conv", "geo_vis_fusion.fconv_ated1.2.weight"),
    ("wn", "mlp_geo.layers1.layers.0.linear"), ("wn", "mlp_geo.layers1.layers.1.linear"), ("wn", "mlp_geo.layers1.layers.2.linear"),
    ("lin", "mlp_geo.layers1.layers.3.linear"),
    ("wn", "mlp_geo.layers2.layers.0.linear"), ("wn", "mlp_geo.layers2.layers.1.linear"), ("lin", "mlp_geo.layers2.layers.2.linear"),
    ("lin", "ibr_compress_gfeat"),
    ("conv", "tex_vis_fusion.fconv_at.0.weight"), ("conv", "tex_vis_fusion.fconv_at.2.weight"),
    ("conv", "tex_vis_fusion.fconv.0.weight"), ("conv", "tex_vis_fusion.fconv.2.weight"),
]

_LAYOUT = None


def layout():
    """Row bases and slot tables of the spills (from the library: layer_spec.h / weights_pack.cpp are the one description)."""
    global _LAYOUT
    if _LAYOUT is None:
        rows = [ctypes.c_int() for _ in range(4)]
        check(lib.vanerf_spill_rows(*[ctypes.byref(r) for r in rows]))
        layers, at = [], 0
        for l in range(NUM_LAYERS):
            xr, yr, no = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            check(lib.vanerf_layer_rows(l, ctypes.byref(xr), ctypes.byref(yr), ctypes.byref(no)))
            n_slots = lib.vanerf_layer_slots(l, None, 0)
            if n_slots < 0:
                check(n_slots)
            buf = (ctypes.c_int32 * n_slots)()
            check(min(0, lib.vanerf_layer_slots(l, ctypes.cast(buf, ctypes.c_void_p), n_slots)))
            layers.append({"x_row": xr.value, "y_row": yr.value, "n_out": no.value, "n_slots": n_slots, "slots": torch.tensor(list(buf), dtype=torch.long),
                           "flat": at})
            at += no.value * n_slots
        ig = []
        for which in range(9):
            o, c, st = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            check(min(0, lib.vanerf_ig_tensor(which, ctypes.byref(o), ctypes.byref(c), ctypes.byref(st))))
            ig.append((o.value, c.value, st.value))
        _LAYOUT = {"x_rows": rows[0].value, "y_rows": rows[1].value, "aux_rows": rows[2].value, "ig_rows": rows[3].value, "layers": layers, "flat": at,
                   "ig": dict(zip(("pix0", "nn0", "tw0", "pix1", "nn1", "tw1", "row_nn", "row_tw", "tex_xy"), ig))}
    return _LAYOUT


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Workspace:
    """Spills of one block of samples (reused block after block) and the flat accumulator of the weight gradients."""

    def __init__(self, block, device):
        L = layout()
        self.block = npad = (int(block) + 31) // 32 * 32
        f32 = torch.float32
        self.xs = torch.empty(L["x_rows"], npad, dtype=f32, device=device)
        self.aux = torch.empty(L["aux_rows"], npad, dtype=f32, device=device)
        self.ys = torch.empty(L["y_rows"], npad, dtype=f32, device=device)
        self.ig = torch.empty(L["ig_rows"] * npad, dtype=f32, device=device)  # nine row-major tensors, one after the other (vanerf_ig_tensor)
        self.raw = torch.empty(npad, 5, dtype=f32, device=device)
        self.valid = torch.empty(npad, dtype=torch.uint8, device=device)
        # weight-gradient accumulators: per layer (SLICES, n_out, n_slots), summed over the slices once per step (parameter_gradients)
        self.slices = SLICES if npad % SLICES == 0 and npad // SLICES >= 32 else 1
        self.dw = torch.zeros(self.slices * L["flat"], dtype=f32, device=device)  # [slice][all layers]: one sum over the slices closes a step
        by_slice = self.dw.view(self.slices, L["flat"])
        self.dw_l = [by_slice[:, lay["flat"]:lay["flat"] + lay["n_out"] * lay["n_slots"]].view(self.slices, lay["n_out"], lay["n_slots"]) for lay in L["layers"]]

    def bytes(self):
        return sum(t.numel() * t.element_size() for t in (self.xs, self.aux, self.ys, self.ig, self.raw, self.valid, self.dw))


def run_block(ws, w0, fdat, pts, q_sdf, q_vis, knn, d, noise=None, d2=None, noise2=None):
    """One block of n samples: forward spill, eval_func's derivative, backward chain, weight products.  d, d2: (n, 5) gradients with respect to
    eval_func's outputs [alpha, sdf, r, g, b] (d2 / noise2: the second set of draws the coarse points carry inside the fine batch, or None).
    Returns the block's input gradients (input_gradients(): views into the workspace, valid until the next block) and n."""
    n = pts.shape[0]
    npad = (n + 31) // 32 * 32
    assert npad <= ws.block
    st = R._stream()
    xs, aux, ys = (t[:, :npad] if npad == ws.block else None for t in (ws.xs, ws.aux, ws.ys))
    if npad != ws.block:  # a shorter last block: compact spills of its own width (the kernels address rows with stride npad)
        xs, aux, ys = (t.view(-1)[:t.shape[0] * npad].view(t.shape[0], npad) for t in (ws.xs, ws.aux, ws.ys))
    ig = ws.ig
    check(lib.vanerf_query_forward_spill(w0.handle, ctypes.byref(fdat.c), _ptr(pts), _ptr(q_sdf), _ptr(q_vis), _ptr(knn), n, npad, _ptr(ws.raw),
                                         _ptr(ws.valid), _ptr(xs), _ptr(aux), _ptr(R._queue_word(pts.device)), st))
    c = lambda t: None if t is None else t.reshape(-1).contiguous() if t.dim() == 1 or t.shape[-1] != 5 else t.contiguous()
    d, d2, noise, noise2 = c(d), c(d2), c(noise), c(noise2)  # (eval_func's derivative happens inside the kernel)
    check(lib.vanerf_query_backward(w0.handle, _ptr(d), _ptr(d2), _ptr(noise), _ptr(noise2), _ptr(ws.raw), _ptr(ws.valid), n, npad, _ptr(xs),
                                    _ptr(aux), _ptr(ys), _ptr(ig), st))
    _weight_products_on(ws, xs, ys, npad)
    return input_gradients(ig, n, npad), n


def _weight_products_on(ws, xs, ys, npad, use_torch=False):
    """dW'_l += Ys_l Xs_l^T for every layer.  The reduction over the block's samples is cut into slices -- the output is tiny and an unsliced
    product runs on a handful of CUs -- and every slice accumulates in place; the slices are summed once per step.  One launch of
    vanerf_weight_products for all twenty layers (csrc/weight_products.hip); use_torch: one sliced torch.baddbmm per layer, the path it replaced
    (160 launches and 3.7 ms per step; kept as the checker, tools/bench_dw_products.py)."""
    L = layout()
    S = ws.slices if npad % (32 * ws.slices) == 0 else 1
    if not use_torch:
        check(lib.vanerf_weight_products(_ptr(xs), _ptr(ys), npad, S, ws.slices, _ptr(ws.dw), R._stream()))
        return
    per = npad // S
    for lay, acc in zip(L["layers"], ws.dw_l):
        g = ys[lay["y_row"]:lay["y_row"] + lay["n_out"]]
        x = xs[lay["x_row"]:lay["x_row"] + lay["n_slots"]]
        a = acc[:S]
        torch.baddbmm(a, g.view(lay["n_out"], S, per).transpose(0, 1), x.view(lay["n_slots"], S, per).permute(1, 2, 0), out=a)


_MAPS = {}


def _channel_maps(device, P):
    """Once per device: where every slot of the flat accumulator goes in a flat [all layers][n_out][kin] channel-space vector (slots that carry
    no channel, and the bias slots, are left out), the bias slots' positions, and the layers' offsets / shapes there."""
    key = str(device)
    if key not in _MAPS:
        L = layout()
        src, dst, bias_src, shapes, at = [], [], [], [], 0
        for lay, spec in zip(L["layers"], LAYER_PARAMS):
            kind = spec[0]
            ref = P[spec[1]] if kind == "conv" else P[spec[1] + (".weight_v" if kind == "wn" else ".weight")]
            kin, n_out, n_slots, sl = ref.shape[1], lay["n_out"], lay["n_slots"], lay["slots"]
            cols = (sl >= 0).nonzero().view(-1)
            o = torch.arange(n_out)[:, None]
            src.append((lay["flat"] + o * n_slots + cols[None, :]).reshape(-1))
            dst.append((at + o * kin + sl[cols][None, :]).reshape(-1))
            b = (sl == -2).nonzero().view(-1)
            bias_src.append(lay["flat"] + torch.arange(n_out) * n_slots + int(b[0]) if b.numel() else None)
            shapes.append((at, n_out, kin))
            at += n_out * kin
        _MAPS[key] = {"src": torch.cat(src).to(device), "dst": torch.cat(dst).to(device), "bias": [None if b is None else b.to(device) for b in bias_src],
                      "shapes": shapes, "total": at}
    return _MAPS[key]


def parameter_gradients(ws, P):
    """The flat accumulator(s) -> gradients of the reference's parameters (dict key -> tensor).  P: name -> leaf tensor.
    ws: a Workspace or a list of them (summed in list order).  One sum over the slices and one index_add for all twenty layers (slot -> input
    channel; duplicated operands add up), then views; weight-norm through autograd of the fold itself."""
    L = layout()
    wss = ws if isinstance(ws, (list, tuple)) else [ws]
    flat = wss[0].dw.view(wss[0].slices, L["flat"]).sum(0)
    for other in wss[1:]:
        flat = flat + other.dw.view(other.slices, L["flat"]).sum(0)
    M = _channel_maps(flat.device, P)
    chan = torch.zeros(M["total"], dtype=torch.float32, device=flat.device).index_add_(0, M["dst"], flat[M["src"]])
    out = {}
    for li, (lay, spec) in enumerate(zip(L["layers"], LAYER_PARAMS)):
        at, n_out, kin = M["shapes"][li]
        dw = chan[at:at + n_out * kin].view(n_out, kin)
        kind = spec[0]
        if kind == "conv":
            ref = P[spec[1]]
            g = torch.zeros_like(ref)
            g[:n_out, :, 0] = dw  # (fconv.2 has 40 output channels, of which one view uses 3: src/model.py:1613, 1635)
            out[spec[1]] = g
            continue
        out[spec[1] + ".bias"] = flat[M["bias"][li]]
        if kind == "lin":
            out[spec[1] + ".weight"] = dw
        else:  # weight-norm fold W = v g / ||v||_row (src/utils.py:674-675), differentiated by autograd itself
            v = P[spec[1] + ".weight_v"].detach().requires_grad_(True)
            gg = P[spec[1] + ".weight_g"].detach().requires_grad_(True)
            with torch.enable_grad():
                w = v * (gg / v.norm(2, dim=1, keepdim=True))
            gv, g_g = torch.autograd.grad(w, [v, gg], dw)
            out[spec[1] + ".weight_v"], out[spec[1] + ".weight_g"] = gv, g_g
    return out


def input_gradients(ig, n, npad):
    """The block's IG spill -> the gradients of the gathered inputs: dict of (n, C) views (no copies; rows `stride` floats apart).
    pix0 / nn0 / tw0 (64), pix1 / nn1 / tw1 (8): GeoVisFusion's pixel feature, nearest and twin vertex rows of both scales;
    row_nn / row_tw (29): TexVisFusion's vertex rows [img3 | tex8 | global18]; tex_xy (8): the texture map's pixel feature."""
    return {k: ig[o * npad:(o + st) * npad].view(npad, st)[:n, :c] for k, (o, c, st) in layout()["ig"].items()}


_WS = {}


def workspace(block, device):
    key = (int(block), str(device))
    if key not in _WS:
        _WS.clear()  # one block size at a time
        _WS[key] = Workspace(block, device)
    return _WS[key]


def _taps(xy, H, W, use_torch=False):
    """The four bilinear taps of feat_sample (src/utils.py:136-151; border padding, align_corners) at (n, 2) coordinates in [-1, 1]:
    int32 row indices of the channel-last map [4][n] and their weights [4][n] -- the kernel's bilin_setup, same arithmetic
    (vanerf_bilinear_taps: one launch; use_torch: the ~15 element-wise launches it replaced, kept as the checker)."""
    if not use_torch:
        n = xy.shape[0]
        xy = xy.to(torch.float32).contiguous()
        idx = torch.empty(4, n, dtype=torch.int32, device=xy.device)
        w = torch.empty(4, n, dtype=torch.float32, device=xy.device)
        check(lib.vanerf_bilinear_taps(_ptr(xy), n, int(H), int(W), _ptr(idx), _ptr(w), R._stream()))
        return idx, w
    x = ((xy[:, 0] + 1.0) * (0.5 * (W - 1))).clamp(0.0, W - 1.0)
    y = ((xy[:, 1] + 1.0) * (0.5 * (H - 1))).clamp(0.0, H - 1.0)
    x0, y0 = x.floor(), y.floor()
    wx, wy = x - x0, y - y0
    x0, y0 = x0.long(), y0.long()
    x1, y1 = (x0 + 1).clamp(max=W - 1), (y0 + 1).clamp(max=H - 1)
    idx = torch.stack([y0 * W + x0, y0 * W + x1, y1 * W + x0, y1 * W + x1]).to(torch.int32).contiguous()
    w = torch.stack([(1.0 - wx) * (1.0 - wy), wx * (1.0 - wy), (1.0 - wx) * wy, wx * wy]).contiguous()
    return idx, w


class InputScatter:
    """Accumulates the input gradients of all blocks into the feature maps (channel-last rows) and the per-vertex tables."""

    def __init__(self, frame, device):
        g0, g1, tx = frame["feat_geo"][0], frame["feat_geo"][1], frame["feat_tex"]
        z = lambda r, c: torch.zeros(r, c, dtype=torch.float32, device=device)
        self.shapes = {"map0": g0.shape, "map1": g1.shape, "tex": tx.shape}
        self.acc = {"map0": z(g0.shape[2] * g0.shape[3], g0.shape[1]), "map1": z(g1.shape[2] * g1.shape[3], g1.shape[1]),
                    "tex": z(tx.shape[2] * tx.shape[3], tx.shape[1]), "vtab0": z(2 * NUM_V, g0.shape[1]), "vtab1": z(2 * NUM_V, g1.shape[1]),
                    "table29": z(2 * NUM_V, 29)}
        self.vis = frame["vert_vis"].float()

    def prepare(self, xy, knn):
        """Tap indices / weights and vertex indices / visibilities of ALL samples of a chunk of rays, once (the blocks then take slices)."""
        self.taps = {name: _taps(xy, self.shapes[name][2], self.shapes[name][3]) for name in ("map0", "map1", "tex")}
        self.knn = knn.contiguous()
        self.twin = torch.where(knn >= NUM_V, knn - NUM_V, knn + NUM_V).contiguous()
        self.vn, self.vt = self.vis[self.knn.long()].contiguous(), self.vis[self.twin.long()].contiguous()

    def add(self, sl, g):
        """sl: the block's slice of the prepared samples; g: input_gradients() of the block."""
        for name, key in (("map0", "pix0"), ("map1", "pix1"), ("tex", "tex_xy")):
            idx, w = self.taps[name]
            R.scatter_add_taps(self.acc[name], idx, w, sl, g[key])
        for name, kn, kt in (("vtab0", "nn0", "tw0"), ("vtab1", "nn1", "tw1"), ("table29", "row_nn", "row_tw")):
            R.scatter_add_rows2(self.acc[name], self.knn[sl], g[kn], self.vn[sl], self.twin[sl], g[kt], self.vt[sl])

    def map_gradient(self, name):
        """(1, C, H, W) gradient of a feature map from its pixel taps."""
        _, C, H, W = self.shapes[name]
        return self.acc[name].view(H, W, C).permute(2, 0, 1)[None].contiguous()
