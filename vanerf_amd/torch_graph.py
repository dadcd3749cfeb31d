"""Differentiable PyTorch statement of the per-sample networks and the compositing, used ONLY to build the autograd graph of a
training step (first stage: "fused HIP forward + PyTorch autograd backward", BASELINE config 5).

The values of a training step come from the HIP path (vanerf_amd.renderer.render_pass); this module re-evaluates the same networks at
the SAME sample points -- the points, importance samples, mesh queries (signed distance, visibility, nearest vertex) and noise draws
are taken from the HIP pass, none of them carries gradient in the reference either (importance_sample runs under no_grad,
src/model.py:1432; kaolin / pytorch3d inputs are constants) -- with torch ops on the device, so that gradients reach the module's
parameters and the encoder feature maps.  `straight_through` then returns HIP values with this graph's gradients.

It is not a fallback: nothing here runs unless autograd is recording, and it cannot produce a frame on its own (it has no ray
generation, no mesh query, no sampling).  Reference anchors: VANeRF.query / query_color src/model.py:748-957, eval_func 1140-1160,
sdf_activation + rgba2out 879-882, 1464-1494, GeoVisFusion / TexVisFusion src/networks.py:27-106, 219-293, MLPUNetFusion
src/utils.py:609-880, SpatialEncoder src/spatial.py:20-117.  n_views == 1, batch 1 (as everywhere in this package)."""
import math

import torch
import torch.nn.functional as F

from .renderer import _avg_pool3

NUM_V = 779  # vertices per hand: the "other hand" twin of vertex i is (i + 779) mod 1558 (src/networks.py:30-32)


def _scatter_rows(n_rows, idx32, g, w=None):
    """sum over samples of w[i] g[i] into row idx[i] of a zero (n_rows, C) table: vanerf_scatter_add_rows (LDS-resident table slices; as
    torch's index_add_ -- contended global atomics, hundreds of samples per row -- this was 20 ms of a 73 ms step)."""
    from . import renderer as R
    return R.scatter_add_rows(torch.zeros(n_rows, g.shape[1], dtype=torch.float32, device=g.device), idx32, g.contiguous(), w)


class _Rows(torch.autograd.Function):
    """table.index_select(0, idx) with the HIP scatter as its backward."""

    @staticmethod
    def forward(ctx, table, idx64, idx32):
        ctx.save_for_backward(idx32)
        ctx.n_rows = table.shape[0]
        return table.index_select(0, idx64)

    @staticmethod
    def backward(ctx, g):
        return _scatter_rows(ctx.n_rows, ctx.saved_tensors[0], g), None, None


class _Bilinear(torch.autograd.Function):
    """Four-tap blend of rows of a channel-last map; backward: four weighted scatters (the tap weights ride along in the kernel)."""

    @staticmethod
    def forward(ctx, rows, i64, i32, wx, wy):
        ctx.save_for_backward(i32, wx, wy)
        ctx.n_rows = rows.shape[0]
        t = [rows.index_select(0, i64[k]) for k in range(4)]
        return (t[0] * (1.0 - wx) + t[1] * wx) * (1.0 - wy) + (t[2] * (1.0 - wx) + t[3] * wx) * wy

    @staticmethod
    def backward(ctx, g):
        i32, wx, wy = ctx.saved_tensors
        from . import renderer as R
        g = g.contiguous()
        out = torch.zeros(ctx.n_rows, g.shape[1], dtype=torch.float32, device=g.device)
        ws = ((1.0 - wx) * (1.0 - wy), wx * (1.0 - wy), (1.0 - wx) * wy, wx * wy)
        for k in range(4):
            R.scatter_add_rows(out, i32[k], g, ws[k].reshape(-1).contiguous())
        return out, None, None, None, None


def sample_map(feat, xy):
    """feat_sample (src/utils.py:136-151): (1,C,H,W) map at (N,2) coordinates in [-1,1] -> (N,C); bilinear, border, align_corners.
    Written as four row gathers from the channel-last map: the coordinates carry no gradient here, and the backward of a row gather is
    a scatter-add of contiguous C-float rows (vanerf_scatter_add_rows), where grid_sampler_2d_backward issues one strided atomic per
    channel and tap (4.2 ms per call on the 64-channel map)."""
    _, C, H, W = feat.shape
    with torch.no_grad():
        x = ((xy[:, 0] + 1.0) * (0.5 * (W - 1))).clamp(0.0, W - 1.0)
        y = ((xy[:, 1] + 1.0) * (0.5 * (H - 1))).clamp(0.0, H - 1.0)
        x0, y0 = x.floor(), y.floor()
        wx, wy = (x - x0)[:, None], (y - y0)[:, None]
        x0, y0 = x0.long(), y0.long()
        x1, y1 = (x0 + 1).clamp(max=W - 1), (y0 + 1).clamp(max=H - 1)
        i64 = torch.stack([y0 * W + x0, y0 * W + x1, y1 * W + x0, y1 * W + x1])
    rows = feat[0].permute(1, 2, 0).reshape(H * W, C)
    if not (rows.is_cuda and rows.requires_grad):
        tap = lambda k: rows.index_select(0, i64[k])
        return (tap(0) * (1.0 - wx) + tap(1) * wx) * (1.0 - wy) + (tap(2) * (1.0 - wx) + tap(3) * wx) * wy
    return _Bilinear.apply(rows, i64, i64.to(torch.int32), wx, wy)


class _Linear(torch.autograd.Function):
    """y = x W^T + b over N samples (N ~ 8e5 in a training step) with a backward that keeps the whole chip busy: the weight gradient
    dW = dY^T X is a reduction over N into a tiny (out x in) matrix, which the GEMM library tiles by OUTPUT (48 workgroups for 128 x 358, on
    256 CUs: 17 TFLOP/s fp32, 39 % of the step); here the reduction is cut into 64 slices (one batched GEMM, 3 072 workgroups) that are
    summed afterwards."""

    SLICES = 64

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            n, S = x.shape[0], _Linear.SLICES
            per = n // S
            if per >= 256:
                head = per * S
                gw = torch.bmm(g[:head].view(S, per, -1).transpose(1, 2), x[:head].view(S, per, -1)).sum(0)
                if head < n:
                    gw = gw + g[head:].t() @ x[head:]
            else:
                gw = g.t() @ x
        gb = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _linear_wn(P, prefix, x):
    """Linear of MLPUNetFusion (src/utils.py:670-685): weight-normed, W = g v / ||v||_row, except the last layer of each stack."""
    if prefix + ".weight_v" in P:
        v, g = P[prefix + ".weight_v"], P[prefix + ".weight_g"]
        return _Linear.apply(x, v * (g / v.norm(2, dim=1, keepdim=True)), P[prefix + ".bias"])
    return _Linear.apply(x, P[prefix + ".weight"], P[prefix + ".bias"])


def _conv1(P, key, x):
    return _Linear.apply(x, P[key][:, :, 0], None)  # bias-free Conv1d(k = 1) on (N,C)


def _softplus(x):
    return F.softplus(x, beta=100, threshold=20)  # src/utils.py:656


def project(pts, cam):
    """src/model.py:780-788: world points (N,3) -> xy in [-1,1] (N,2), z in [-1,1] (N,1) of the source view."""
    vh = pts @ cam["KRT"][0, :3, :3].t() + cam["KRT"][0, :3, 3]
    z = vh[:, 2:3]
    xy = vh[:, :2] / z
    xy = torch.stack([2.0 * (xy[:, 0] / (cam["width"] - 1.0)) - 1.0, 2.0 * (xy[:, 1] / (cam["height"] - 1.0)) - 1.0], -1)
    return xy, 2.0 * (z - cam["znear"]) / (cam["zfar"] - cam["znear"]) - 1.0


def project_vertices(vert, cam):
    """src/model.py:845-853 (z + 1e-8 in the divide)."""
    vh = vert @ cam["KRT"][0, :3, :3].t() + cam["KRT"][0, :3, 3]
    xy = vh[:, :2] / (vh[:, 2:3] + 1e-8)
    return torch.stack([2.0 * (xy[:, 0] / (cam["width"] - 1.0)) - 1.0, 2.0 * (xy[:, 1] / (cam["height"] - 1.0)) - 1.0], -1)


_FREQ = {}


def positional_encoding(pts, kpt3d, extrin, levels=3, scale=1.0, sigma=0.1):
    """SpatialEncoder 'rel_z_decay' (src/spatial.py:59-84, 109-117, 20-43): (N,3) -> (N, (1 + 2 levels) K), blocks [dz | sin.. | cos..] x w."""
    Rm, t = extrin[0, :3, :3], extrin[0, :3, 3]
    c = pts @ Rm.t() + t
    k = kpt3d[0] @ Rm.t() + t
    d = c[:, None] - k[None]                                          # (N,K,3)
    w = torch.exp(-(d ** 2).sum(-1) / (2.0 * sigma ** 2))             # (N,K)
    x = scale * d[..., 2]                                             # (N,K)
    key = (levels, str(pts.device))
    if key not in _FREQ:  # (built once: a host-to-device copy has no place inside a captured graph)
        _FREQ[key] = torch.tensor([math.pi * 2 ** l for l in range(levels)], dtype=torch.float32, device=pts.device)
    freq = _FREQ[key]
    y = x[:, None, :] * freq[None, :, None]                           # (N,L,K)
    blocks = torch.cat([x[:, None], torch.stack((torch.sin(y), torch.cos(y)), 2).reshape(x.shape[0], -1, x.shape[1])], 1)  # [x | sin l0 | cos l0 | sin l1 | ..]
    return (blocks * w[:, None]).reshape(pts.shape[0], -1)


def _nearest_rows(table, vis, idx):
    """KNN_vis (src/networks.py:27-33) with the nearest-vertex index given: rows of the nearest vertex and of its twin, x visibility."""
    twin = (idx + NUM_V) % (2 * NUM_V)
    vi, vt = vis[idx, None], vis[twin, None]
    if not (table.is_cuda and table.requires_grad):
        return table.index_select(0, idx) * vi, table.index_select(0, twin) * vt, vi, vt
    return _Rows.apply(table, idx, idx.to(torch.int32)) * vi, _Rows.apply(table, twin, twin.to(torch.int32)) * vt, vi, vt


def geo_fusion(P, geo_maps, pix, vert_xy, idx, vert_vis, q_vis, q_sdf, pre="geo_vis_fusion."):
    """GeoVisFusion.forward (src/networks.py:75-106): two scales, gates then a gated 2-layer MLP.  -> [(N,64), (N,8)]."""
    out = []
    for i, (at, ated) in enumerate((("fconv_at", "fconv_ated"), ("fconv_at1", "fconv_ated1"))):
        nn_f, tw_f, vis_nn, vis_tw = _nearest_rows(sample_map(geo_maps[i], vert_xy), vert_vis, idx)
        tail = [q_sdf, q_vis, vis_nn, vis_tw]
        a = torch.sigmoid(_conv1(P, pre + at + ".2.weight", torch.relu(_conv1(P, pre + at + ".0.weight", torch.cat([pix[i], nn_f, tw_f] + tail, 1)))))
        g = torch.cat([pix[i] * a[:, 0:1], nn_f * a[:, 1:2], tw_f * a[:, 2:3]] + tail, 1)
        out.append(_conv1(P, pre + ated + ".2.weight", torch.relu(_conv1(P, pre + ated + ".0.weight", g))))
    return out


def geometry_mlp(P, pe, fused, weight, pre="mlp_geo."):
    """MLPUNetFusion.forward for the shipped config (src/utils.py:633-649, 709-719, 744-779, 822-880), one view:
    layers1 with skips at 0 and 2, mean / var pooling with the pixel weight, layers2.  -> (N,2) [sdf_pred, rad], latent (N,128)."""
    x = pe
    for i in range(4):
        if i == 0:
            x = torch.cat([x, fused[0]], -1)
        elif i == 2:
            x = torch.cat([x, fused[1]], -1)
        x = _linear_wn(P, f"{pre}layers1.layers.{i}.linear", x)
        if i != 3:
            x = _softplus(x)
    mean = weight * x
    var = weight * (x - mean) ** 2
    latent = torch.cat([mean, var], -1)
    x = latent
    for i in range(3):
        x = _linear_wn(P, f"{pre}layers2.layers.{i}.linear", x)
        if i != 2:
            x = _softplus(x)
    return x, latent


def texture_vertex_table(P, vert_xy, feat_tex, img, pre="tex_vis_fusion."):
    """Per-frame part of TexVisFusion.forward (src/networks.py:270-279): (NV,29) = [img 3 | tex 8 | global 18]."""
    def stack(x, name):
        hw = x.shape[-1]
        x = F.conv2d(x, P[pre + name + ".0.weight"], padding=1)
        x = torch.relu(F.layer_norm(x, [hw, hw], P[pre + name + ".1.weight"], P[pre + name + ".1.bias"], 1e-6))
        x = F.conv2d(x, P[pre + name + ".3.weight"], padding=1)
        x = torch.relu(F.layer_norm(x, [hw, hw], P[pre + name + ".4.weight"], P[pre + name + ".4.bias"], 1e-6))
        return _avg_pool3(x).reshape(1, x.shape[1], 9)  # AdaptiveAvgPool2d(3) as two small products (renderer._avg_pool3; differentiable)

    gf = torch.cat([stack(img, "fconv4"), stack(feat_tex, "fconv3")], -1)  # (1, NV, 18): the conv stacks have NV output channels
    x = F.conv1d(gf, P[pre + "fconv_gt.0.weight"], padding=1)
    x = torch.relu(F.layer_norm(x, [18], P[pre + "fconv_gt.1.weight"], P[pre + "fconv_gt.1.bias"], 1e-6))
    x = F.conv1d(x, P[pre + "fconv_gt.3.weight"], padding=1)
    x = torch.relu(F.layer_norm(x, [18], P[pre + "fconv_gt.4.weight"], P[pre + "fconv_gt.4.bias"], 1e-6))
    return torch.cat([sample_map(img, vert_xy), sample_map(feat_tex, vert_xy), x[0]], 1)


def texture_fusion(P, table29, tex_xy, img_xy, idx, vert_vis, q_vis, latent24, pre="tex_vis_fusion."):
    """Per-sample part of TexVisFusion.forward (src/networks.py:281-293) -> (N,40); at one view the colour is its first 3 channels
    (IBRRenderingHead is a softmax over a single view: src/model.py:1613, 1635)."""
    nn_f, tw_f, vis_nn, vis_tw = _nearest_rows(table29, vert_vis, idx)
    q = torch.cat([img_xy, tex_xy], 1)
    tail = [q_vis, vis_nn, vis_tw]
    parts = [q, nn_f[:, :11], tw_f[:, :11], nn_f[:, 11:], tw_f[:, 11:], latent24]
    a = torch.sigmoid(_conv1(P, pre + "fconv_at.2.weight", torch.relu(_conv1(P, pre + "fconv_at.0.weight", torch.cat(parts + tail, 1)))))
    g = torch.cat([p * a[:, j:j + 1] for j, p in enumerate(parts)] + tail, 1)
    return _conv1(P, pre + "fconv.2.weight", torch.relu(_conv1(P, pre + "fconv.0.weight", g)))


def networks_at(P, frame, pts, q_sdf, q_vis, knn, noise=None, sp_args=None):
    """VANeRF.query + query_color + eval_func at N given points -> (N,5) [alpha, sdf, r, g, b] with gradient.
    frame: dict(cam, img (1,3,H,W), feat_geo [2 maps], feat_tex, fg_mask (1,1,H,W), verts (NV,3), vert_vis (NV,), kpt3d, extrin,
    table29 (optional, from texture_vertex_table)); q_sdf (N,), q_vis (N,) in {0,1}, knn (N,) int64, noise (N,) or None."""
    sp = sp_args or {"sp_level": 3, "scale": 1.0, "sigma": 0.1}
    cam = frame["cam"]
    xy, z = project(pts, cam)
    eps = 1e-2
    inside = ((xy >= -1.0 - eps) & (xy <= 1.0 + eps)).all(-1, keepdim=True) & (z >= -1.0)
    mask = (inside & (sample_map(frame["fg_mask"].float(), xy) > 0.1)).float()                      # (N,1)
    xyz = 0.5 * torch.cat([xy, z], -1) + 0.5
    pw = torch.sigmoid(5.0 * (torch.min(xyz, 1.0 - xyz) / 0.1 - 1.0)).prod(-1, keepdim=True) * mask
    weight = pw / (pw + 1e-6)                                                                       # one view: pw / (sum over views + 1e-6)
    vert_xy = project_vertices(frame["verts"], cam)
    vis = frame["vert_vis"].float()
    qs, qv = q_sdf.view(-1, 1).float(), q_vis.view(-1, 1).float()
    # Samples outside the source view / foreground mask (about half of a pass) have pixel weight 0: their pooled latent is exactly zero
    # (mean = 0 * x, var = 0 * ..), eval_func masks their sdf and rad, and no gradient flows back through the weight -- so GeoVisFusion, the
    # positional encoding and the geometry MLP (85 % of the per-sample arithmetic) are evaluated on the valid samples only; same values,
    # same gradients.  The texture branch below still runs on every sample: eval_func does not mask the colour.
    keep = None
    if COMPACT_VALID:
        keep = mask.view(-1).nonzero().view(-1)
        if keep.numel() == mask.shape[0]:
            keep = None
    if keep is not None and keep.numel() == 0:  # a block without a valid sample: no geometry branch at all
        geo, latent = torch.zeros(mask.shape[0], 2, device=pts.device), torch.zeros(mask.shape[0], 128, device=pts.device)
    else:
        sel = (lambda t: t) if keep is None else (lambda t: t.index_select(0, keep))
        xy_k = sel(xy)
        pix = [sample_map(f, xy_k) for f in frame["feat_geo"]]
        fused = geo_fusion(P, frame["feat_geo"], pix, vert_xy, sel(knn), vis, sel(qv), sel(qs))
        pe = positional_encoding(sel(pts), frame["kpt3d"], frame["extrin"], sp["sp_level"], sp["scale"], sp["sigma"])
        geo, latent = geometry_mlp(P, pe, fused, sel(weight))
        if keep is not None:
            geo = torch.zeros(mask.shape[0], geo.shape[1], device=geo.device).index_copy(0, keep, geo)
            latent = torch.zeros(mask.shape[0], latent.shape[1], device=latent.device).index_copy(0, keep, latent)
    latent24 = _Linear.apply(latent, P["ibr_compress_gfeat.weight"], P["ibr_compress_gfeat.bias"])
    table29 = frame.get("table29")
    if table29 is None:
        table29 = texture_vertex_table(P, vert_xy, frame["feat_tex"], frame["img"])
    rgb = texture_fusion(P, table29, sample_map(frame["feat_tex"], xy), sample_map(frame["img"], xy), knn, vis, qv, latent24)[:, :3]
    # eval_func (src/model.py:1140-1160); a tuple of noise vectors gives a tuple of outputs that share the networks' evaluation (the coarse
    # points of a training pass appear in the coarse and in the fine composite with different draws)
    sdf = mask * geo[:, 0:1] + (1.0 - mask) * (0.1 / cam["nml_scale"])

    def with_noise(nz):
        rad = geo[:, 1:2] if nz is None else geo[:, 1:2] + nz.view(-1, 1)
        return torch.cat([mask * torch.relu(rad), sdf, rgb], -1)

    return tuple(with_noise(nz) for nz in noise) if isinstance(noise, tuple) else with_noise(noise)


def composite(P, rgba, z, mesh_sdf):
    """sdf_activation + rgba2out (src/model.py:879-882, 1464-1494): rgba (R,S,5), z (R,S), mesh_sdf (R,S) ->
    colour (R,3), depth, alpha, sdf (R,)."""
    beta = torch.clamp(P["sigmoid_beta"], min=2e-3)
    sigma = torch.sigmoid(-(rgba[..., 0] + mesh_sdf) / beta) / beta
    dist = torch.cat([z[:, 1:] - z[:, :-1], 1e10 * torch.ones_like(z[:, :1])], -1)
    c = 1.0 - torch.exp(-sigma * dist)
    w = c * torch.cumprod(torch.cat([torch.ones_like(c[:, :1]), 1.0 - c[:, :-1]], -1), -1)
    acc = w.sum(-1)
    return (rgba[..., 2:] * w[..., None]).sum(-2), (z * w).sum(-1) / (acc + 1e-8), acc, (rgba[..., 1] * w).sum(-1) / (acc + 1e-8)


def straight_through(value, graph):
    """HIP value, this module's gradient."""
    return graph + (value - graph).detach()


# Rays per chunk of the backward pass (PassGradient); None = the whole patch at once.  Chunking by rays repeats both stages per chunk; the block
# size below bounds memory more cheaply (it only cuts the second stage).  Measured on the 64x64 patch at 64 + 64 samples (
# tools/perf_train_step.py, one MI355X; 524 k network evaluations per step): whole patch 42 ms / 7.5 GiB; blocks of 262 144 samples 53 ms / 4.2 GiB;
# 131 072: 63 ms / 2.5 GiB.  With 288 GB of HBM the default is speed; model config keys `grad_rays_per_chunk`, `grad_samples_per_block`.
# (bf16 operands for this graph's GEMMs were measured too: 9 % faster, and the parameter gradients moved by 4e-2 relative -- dropped.)
GRAD_RAYS_PER_CHUNK = None
# Samples per block of the second stage of the backward pass (PassGradient); None = all samples of a chunk of rays in one block.
GRAD_SAMPLES_PER_BLOCK = None


COMPACT_VALID = True  # networks_at: evaluate the geometry branch on valid samples only (tests compare both settings)


class _BlockGraph:
    """Second stage of PassGradient for blocks of ONE fixed size, captured once as a HIP graph and replayed block after block (config key
    `grad_graph_blocks`, with `grad_samples_per_block`).  Small blocks bound the step's memory, but eagerly every block re-launches the
    graph's ~2 000 kernels and the step turns host-bound (131 072 samples per block: 63 ms); replayed, a block costs the host a few input
    copies.  What capture needs: inputs, per-frame tensors and gradient accumulators at fixed addresses (copied in, read out), parameters that
    stay where they are (in-place optimizer updates; a moved parameter re-captures), no data-dependent shapes (the valid-sample compaction of
    networks_at is off inside: every sample of a block is evaluated)."""
    cache = {}

    def __init__(self, leaves, names, frame, table, block, two, with_noise, sp_args):
        dev = table.device
        f32 = torch.float32
        self.block, self.two = block, two
        self.pts, self.qs = torch.zeros(block, 3, device=dev), torch.zeros(block, device=dev)
        self.qv, self.knn = torch.zeros(block, dtype=frame_dtype(frame, "q_vis"), device=dev), torch.zeros(block, dtype=torch.int32, device=dev)
        self.nz = torch.zeros(block, device=dev) if with_noise else None
        self.nz2 = torch.zeros(block, device=dev) if two else None
        self.d, self.d2 = torch.zeros(block, 5, device=dev), (torch.zeros(block, 5, device=dev) if two else None)
        # leaves: parameters are read where they live; the encoders' feature maps (new tensors every step) and the vertex table get fixed homes
        self.static = {n: (t.detach().clone() if n.startswith("@") else t.detach()).requires_grad_(True) for n, t in zip(names, leaves)}
        self.table = table.detach().clone().requires_grad_(True)
        self.frame = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in frame.items() if k not in ("cam", "feat_geo", "feat_tex", "table29")}
        self.frame["cam"] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in frame["cam"].items()}
        order = list(self.static.values()) + [self.table]
        flat = torch.zeros(sum(t.numel() for t in order), dtype=f32, device=dev)
        self.flat, self.acc, at = flat, [], 0
        for t in order:
            self.acc.append(flat[at:at + t.numel()].view(t.shape))
            at += t.numel()
        P = {k: v for k, v in self.static.items() if not k.startswith("@")}
        fr = dict(self.frame, feat_geo=[self.static["@feat_geo0"], self.static["@feat_geo1"]], feat_tex=self.static["@feat_tex"], table29=self.table)

        def body():
            global COMPACT_VALID
            keep, COMPACT_VALID = COMPACT_VALID, False
            try:
                noise = (self.nz, self.nz2) if two else self.nz
                outs = networks_at(P, fr, self.pts, self.qs, self.qv, self.knn.long(), noise, sp_args)
                grads = torch.autograd.grad(list(outs) if two else outs, order, [self.d, self.d2] if two else self.d, allow_unused=True)
            finally:
                COMPACT_VALID = keep
            for a, g in zip(self.acc, grads):
                if g is not None:
                    a.add_(g)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.enable_grad():
            for _ in range(2):  # warm-up outside capture (library handles, workspaces, autotuning)
                body()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.enable_grad(), torch.cuda.graph(self.graph):
            body()

    def begin(self, leaves, names, frame, table):
        """New step: this step's feature maps, per-frame tensors and vertex table into their fixed homes, accumulators to zero."""
        with torch.no_grad():
            for n, t in zip(names, leaves):
                if n.startswith("@"):
                    self.static[n].copy_(t)
            self.table.copy_(table)
            for k, v in self.frame.items():
                if torch.is_tensor(v):
                    v.copy_(frame[k])
            for k, v in self.frame["cam"].items():
                if torch.is_tensor(v):
                    v.copy_(frame["cam"][k])
            self.flat.zero_()

    def run(self, pts, qs, qv, knn, nz, nz2, d, d2):
        n = pts.shape[0]
        with torch.no_grad():
            for dst, src in ((self.pts, pts), (self.qs, qs), (self.qv, qv), (self.knn, knn), (self.nz, nz), (self.nz2, nz2), (self.d, d), (self.d2, d2)):
                if dst is not None:
                    dst[:n].copy_(src)
            if n < self.block:  # the last, shorter block: the tail keeps old samples with zero output gradients -- they add nothing
                self.d[n:].zero_()
                if self.d2 is not None:
                    self.d2[n:].zero_()
        self.graph.replay()

    def results(self):
        out = self.flat.clone()
        res, at = [], 0
        for a in self.acc:
            res.append(out[at:at + a.numel()].view(a.shape))
            at += a.numel()
        return res

    @classmethod
    def get(cls, leaves, names, frame, table, block, two, with_noise, sp_args):
        cam = frame["cam"]
        key = (block, two, with_noise, tuple(sorted(sp_args.items())), tuple(t.data_ptr() for n, t in zip(names, leaves) if not n.startswith("@")),
               tuple(tuple(t.shape) for n, t in zip(names, leaves) if n.startswith("@")), tuple((k, float(v)) for k, v in sorted(cam.items()) if not torch.is_tensor(v)),
               tuple((k, tuple(v.shape)) for k, v in sorted(frame.items()) if torch.is_tensor(v)), str(table.device))
        if key not in cls.cache:
            cls.cache.clear()  # one configuration at a time (each holds a block's activations in its private pool)
            cls.cache[key] = cls(leaves, names, frame, table, block, two, with_noise, sp_args)
        return cls.cache[key]


def frame_dtype(frame, key):
    return torch.uint8


class PassGradient(torch.autograd.Function):
    """forward: the HIP pass's images, unchanged.  backward: the gradients of this module's graph at the samples of that pass with respect to
    the leaves (the module's parameters and the encoders' feature maps), in two stages: the composites are differentiated at the pass's own
    per-sample values (HIP), which gives the gradient with respect to every sample's network outputs; then the per-sample networks are
    evaluated and differentiated block of samples by block (coarse batch, fine batch, optionally smaller blocks / chunks of rays): samples
    are independent, so the gradient is the sum over blocks, and only one block's activations exist at a time.  The per-frame vertex table of
    TexVisFusion (two conv stacks over the source image) is shared by all samples: its graph is built once, the chunks accumulate the
    gradient with respect to the table, and one backward through the stacks closes the step."""

    @staticmethod
    def forward(ctx, spec, *leaves):
        ctx.spec = spec
        ctx.save_for_backward(*leaves)
        return tuple(v.clone() for v in spec["values"])

    @staticmethod
    def backward(ctx, *gouts):
        spec, names = ctx.spec, ctx.spec["names"]
        import os, time
        marks = [] if os.environ.get("VANERF_TIME_BACKWARD") else None  # diagnostic: synchronised wall time per section (tools/perf_train_parts.py)

        def mark(what):
            if marks is not None:
                torch.cuda.synchronize()
                marks.append((what, time.perf_counter()))
        mark("start")
        with torch.enable_grad():
            loc = [t.detach().requires_grad_(True) for t in ctx.saved_tensors]
            L = dict(zip(names, loc))
            P = {k: v for k, v in L.items() if not k.startswith("@")}
            frame = dict(spec["frame"], feat_geo=[L["@feat_geo0"], L["@feat_geo1"]], feat_tex=L["@feat_tex"])
            total = [None] * len(loc)

            def accumulate(grads):
                for i, g in enumerate(grads):
                    if g is not None:
                        total[i] = g if total[i] is None else total[i] + g

            hip_state = None
            table_graph = texture_vertex_table(P, project_vertices(frame["verts"], frame["cam"]), frame["feat_tex"], frame["img"])
            table = table_graph.detach().requires_grad_(True)
            g_table = torch.zeros_like(table)
            o, keys = spec["pass"], spec["keys"]
            R = o["z"].shape[0]
            c, f = o["coarse"], o.get("fine")
            step = spec["rays_per_chunk"] or R
            for r0 in range(0, R, step):
                r1 = min(R, r0 + step)
                # (1) the composites, differentiated at the HIP pass's own per-sample values: a small graph over (rays, samples, 5) tensors that
                #     yields the gradient with respect to every sample's [alpha, sdf, r, g, b] (and to sigmoid_beta)
                cf = o.get("coarse_in_fine") if f is not None else None
                n_fine = 0 if f is None else (o["z_fine"].shape[1] if o.get("z_fine") is not None else f["rgba"].shape[1])
                if spec.get("hip_backward") is not None and max(c["rgba"].shape[1], n_fine) <= 256:
                    # vanerf_composite_backward: one launch per composite instead of this graph's ~500 (rays x samples x 5 element-wise kernels, a
                    # cumprod whose backward blocks the host); sigmoid_beta is the handle's device copy (this step's parameter, clamped)
                    from . import renderer as HR
                    w0 = spec["hip_backward"]["w0"]
                    gk = {}
                    for k, g in zip(keys, gouts):
                        if g is not None:
                            gk[k] = (g.reshape(3, -1).t()[r0:r1] if k.startswith("tex_fg") else g.reshape(-1)[r0:r1]).contiguous()
                    with torch.no_grad():
                        d_rc, _, db = HR.composite_backward(w0, c["rgba"][r0:r1], o["z"][r0:r1], c["q_sdf"][r0:r1], gk.get("tex_fg"), gk.get("depth"), gk.get("alpha"))
                        grads = [d_rc]
                        if f is not None:
                            gf = (gk.get("tex_fg_fine"), gk.get("depth_fine"), gk.get("alpha_fine"), gk.get("sdf"))
                            if o.get("fine_src") is not None:
                                rcf = c["rgba"] if cf is None else cf["rgba"]
                                d_rcf, d_rf, db_f = HR.composite_backward(w0, rcf[r0:r1], o["z_fine"][r0:r1], c["q_sdf"][r0:r1], *gf, rgba_n=f["rgba"][r0:r1],
                                                                          sdf_n=f["q_sdf"][r0:r1], src=o["fine_src"][r0:r1])
                                if cf is None:
                                    grads = [d_rc + d_rcf, d_rf]
                                else:
                                    grads = [d_rc, d_rf, d_rcf]
                            else:
                                d_rf, _, db_f = HR.composite_backward(w0, f["rgba"][r0:r1], o["z_fine"][r0:r1], f["q_sdf"][r0:r1], *gf)
                                grads = [d_rc, d_rf]
                            db = db.sum() + db_f.sum()
                        else:
                            db = db.sum()
                        pb = P["sigmoid_beta"]
                        g_beta = [None] * len(loc)
                        g_beta[names.index("sigmoid_beta")] = (db * (pb.detach() >= 2e-3).to(db.dtype)).reshape(pb.shape)  # clamp(min = 2e-3)'s derivative
                    accumulate(g_beta)
                    per_sample = grads  # (only its length is used below)
                    mark("composite backward (HIP)")
                else:
                    rc = c["rgba"][r0:r1].detach().clone().requires_grad_(True)
                    col, dep, acc, _ = composite(P, rc, o["z"][r0:r1], c["q_sdf"][r0:r1])
                    outs = {"tex_fg": col, "depth": dep, "alpha": acc}
                    per_sample = [rc]
                    if f is not None:
                        rf = f["rgba"][r0:r1].detach().clone().requires_grad_(True)
                        per_sample.append(rf)
                        rgba_f, msdf = rf, f["q_sdf"][r0:r1]
                        if o.get("fine_src") is not None:  # the pass re-used the coarse evaluations: merge [coarse | new] by the origin map
                            src = o["fine_src"][r0:r1].long()
                            take = torch.where(src >= 0, src, rc.shape[1] + (-src - 1))
                            rcf = rc
                            if cf is not None:  # (training noise: the coarse points carry other draws inside the fine batch)
                                rcf = cf["rgba"][r0:r1].detach().clone().requires_grad_(True)
                                per_sample.append(rcf)
                            rgba_f = torch.gather(torch.cat([rcf, rf], 1), 1, take[..., None].expand(-1, -1, 5))
                            msdf = torch.gather(torch.cat([c["q_sdf"][r0:r1], f["q_sdf"][r0:r1]], 1), 1, take)
                        col, dep, acc, sdf = composite(P, rgba_f, o["z_fine"][r0:r1], msdf)
                        outs.update({"tex_fg_fine": col, "depth_fine": dep, "alpha_fine": acc, "sdf": sdf})
                    pairs = []
                    for k, g in zip(keys, gouts):
                        if g is None:
                            continue
                        # images are (1,3,h,w) / (1,h,w) over the patch's rays in row-major order: the chunk's rays are a slice of the flattened image
                        gk = g.reshape(3, -1).t()[r0:r1] if k.startswith("tex_fg") else g.reshape(-1)[r0:r1]
                        pairs.append((outs[k], gk))
                    mark("table graph + composite forward")
                    grads = torch.autograd.grad([a for a, _ in pairs], per_sample + loc, [b for _, b in pairs], allow_unused=True)
                    accumulate(grads[len(per_sample):])
                    mark("composite backward")
                    del outs, pairs, col, dep, acc
                # (2) the per-sample networks, one block of samples after the other (samples are independent): only one block's graph exists at a
                #     time, which is what bounds the step's memory -- the coarse and the fine batch are never alive together
                d_per = list(grads[:len(per_sample)])
                # every sample of the chunk in one list: [coarse | new]; a second (noise, gradient) column exists when the coarse points appear
                # in the fine composite with other draws (zeros for the new samples there)
                parts = [(c, d_per[0], None if cf is None else (cf["noise"], d_per[2]))]
                if f is not None:
                    parts.append((f, d_per[1], None))
                cols = {"pts": [], "q_sdf": [], "q_vis": [], "knn": [], "noise": [], "d": [], "noise2": [], "d2": []}
                for part, d, second in parts:
                    S = part["pts"].shape[0] // R
                    sl = slice(r0 * S, r1 * S)
                    n = sl.stop - sl.start
                    cols["pts"].append(part["pts"][sl]); cols["q_sdf"].append(part["q_sdf"].reshape(-1)[sl]); cols["q_vis"].append(part["q_vis"][sl])
                    cols["knn"].append(part["knn"][sl])
                    cols["noise"].append(None if part["noise"] is None else part["noise"][sl])
                    cols["d"].append(torch.zeros(n, 5, device=part["pts"].device) if d is None else d.reshape(-1, 5))
                    if cf is not None:
                        cols["noise2"].append(torch.zeros(n, device=part["pts"].device) if second is None else second[0][sl])
                        cols["d2"].append(torch.zeros(n, 5, device=part["pts"].device) if second is None or second[1] is None else second[1].reshape(-1, 5))
                cat = lambda k: torch.cat(cols[k], 0) if len(cols[k]) > 1 else cols[k][0]
                pts_a, qs_a, qv_a, knn_a, d_a = cat("pts"), cat("q_sdf"), cat("q_vis"), cat("knn"), cat("d")
                nz_a = None if cols["noise"][0] is None else cat("noise")
                nz2_a, d2_a = (cat("noise2"), cat("d2")) if cf is not None else (None, None)
                n_all = pts_a.shape[0]
                if spec.get("hip_backward") is not None:
                    # the fused HIP backward (csrc/query_backward.hip, hip_backward.py): two launches and twenty matrix products per block of samples
                    from . import hip_backward as HB
                    hb = spec["hip_backward"]
                    if hip_state is None:  # the first chunk of rays is a full one: its sample count bounds every later block
                        blk = min(int(hb["block"]), (n_all + 31) // 32 * 32)
                        ws = HB.workspace(blk, pts_a.device)
                        ws.dw.zero_()
                        hip_state = {"ws": ws, "blk": blk, "scatter": HB.InputScatter(frame, pts_a.device)}
                    ws, blk = hip_state["ws"], hip_state["blk"]
                    mark("sample lists")
                    with torch.no_grad():
                        pts_a, qs_a, qv_a, knn_a = pts_a.contiguous(), qs_a.contiguous(), qv_a.contiguous(), knn_a.contiguous()
                        hip_state["scatter"].prepare(project(pts_a, frame["cam"])[0], knn_a)
                        mark("taps")
                        for b0 in range(0, n_all, blk):
                            sl = slice(b0, min(n_all, b0 + blk))
                            ig, nb = HB.run_block(ws, hb["w0"], hb["fdat"], pts_a[sl], qs_a[sl], qv_a[sl], knn_a[sl], d_a[sl],
                                                  None if nz_a is None else nz_a[sl], None if d2_a is None else d2_a[sl], None if nz2_a is None else nz2_a[sl])
                            hip_state["scatter"].add(sl, ig)
                    mark("fused backward blocks")
                    continue
                block = spec.get("samples_per_block") or n_all
                if spec.get("graph_blocks") and spec.get("samples_per_block"):
                    runner = _BlockGraph.get(list(ctx.saved_tensors), names, frame, table, block, nz2_a is not None, nz_a is not None, spec["sp_args"])
                    runner.begin(list(ctx.saved_tensors), names, frame, table)
                    for b0 in range(0, n_all, block):
                        sl = slice(b0, min(n_all, b0 + block))
                        runner.run(pts_a[sl], qs_a[sl], qv_a[sl], knn_a[sl], None if nz_a is None else nz_a[sl], None if nz2_a is None else nz2_a[sl],
                                   d_a[sl], None if d2_a is None else d2_a[sl])
                    res = runner.results()
                    accumulate(res[:-1])
                    g_table += res[-1]
                    continue
                for b0 in range(0, n_all, block):
                    sl = slice(b0, min(n_all, b0 + block))
                    noise = None if nz_a is None else nz_a[sl]
                    if nz2_a is not None:
                        noise = (noise, nz2_a[sl])
                    outs_b = networks_at(P, dict(frame, table29=table), pts_a[sl], qs_a[sl], qv_a[sl], knn_a[sl].long(), noise, spec["sp_args"])
                    if nz2_a is not None:
                        g_block = torch.autograd.grad(list(outs_b), loc + [table], [d_a[sl], d2_a[sl]], allow_unused=True)
                    else:
                        g_block = torch.autograd.grad(outs_b, loc + [table], d_a[sl], allow_unused=True)
                    del outs_b
                    accumulate(g_block[:-1])
                    if g_block[-1] is not None:
                        g_table += g_block[-1]
            if hip_state is not None:
                from . import hip_backward as HB
                sc = hip_state["scatter"]
                by_name = dict(HB.parameter_gradients(hip_state["ws"], P))
                by_name["@feat_geo0"], by_name["@feat_geo1"], by_name["@feat_tex"] = sc.map_gradient("map0"), sc.map_gradient("map1"), sc.map_gradient("tex")
                accumulate([by_name.get(n) for n in names])
                # the per-vertex tables are bilinear samples of the maps at the projected vertices (src/networks.py:86-87, 94-95): their gradient goes
                # back through that gather (and table29's through the per-frame stacks below)
                vert_xy = project_vertices(frame["verts"], frame["cam"])
                tabs = [sample_map(frame["feat_geo"][0], vert_xy), sample_map(frame["feat_geo"][1], vert_xy)]
                accumulate(torch.autograd.grad(tabs, loc, [sc.acc["vtab0"], sc.acc["vtab1"]], allow_unused=True))
                g_table = g_table + sc.acc["table29"]
            mark("parameter gradients")
            accumulate(torch.autograd.grad(table_graph, loc, g_table, allow_unused=True))
            mark("per-frame stacks backward")
        if marks:
            print("PassGradient.backward sections (ms):", [(b[0], round(1e3 * (b[1] - a[1]), 2)) for a, b in zip(marks, marks[1:])])
        return (None, *total)
