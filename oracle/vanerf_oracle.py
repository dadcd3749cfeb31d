"""CPU oracle: a plain-PyTorch fp32 restatement of the VANeRF volume-rendering hot path.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module; the product path (`vanerf_amd/`) never does and fails
loudly when its HIP library is missing.

Pinning: every function below is checked in `tests/test_oracle_golden.py` against vectors
produced by running the reference's own code in the build container
(`oracle/gen_golden.py` -> `tests/golden/*.npz`).  The third-party boundaries -- pytorch3d
`knn_points`, kaolin `point_to_mesh_distance`/`check_sign`, pytorch3d `rasterize_meshes` --
are absent from the reference tree and from this image: PARITY UNPINNED there (see
`oracle/mesh_oracle.c`, which restates their documented semantics).

All `file:line` citations are relative to the reference tree (/root/reference).
Weights are passed as a flat dict with the reference's `state_dict()` key names
(`VANeRF.state_dict()`, SURVEY.md section 5 "Checkpoint / resume").
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

NUM_V = 779  # src/networks.py:25

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _mesh_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libmesh_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
    return _LIB


def _fp(t):
    return ctypes.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------------------------
# third-party boundary restatements (parity unpinned; exact op order in oracle/mesh_oracle.c)
# --------------------------------------------------------------------------------------------
def knn1(query, vert):
    """pytorch3d.ops.knn_points(query, vert, K=1).idx  (src/networks.py:28).  (N,3),(NV,3) -> (N,) int64."""
    q = query.detach().float().contiguous().cpu()
    v = vert.detach().float().contiguous().cpu()
    idx = torch.empty(q.shape[0], dtype=torch.int32)
    _mesh_lib().knn1(_fp(q), ctypes.c_int64(q.shape[0]), _fp(v), ctypes.c_int(v.shape[0]), _fp(idx))
    return idx.long()


def vertex_visibility(vert_xy01, vert_z01, faces, size=256):
    """get_visibility (src/lib/dataset/mesh_util.py:284-318).  (NV,2),(NV,1),(NF,3) -> (NV,1) float {0,1}."""
    xy = vert_xy01.detach().float().contiguous().cpu()
    z = vert_z01.detach().float().contiguous().cpu()
    f = faces.detach().to(torch.int32).contiguous().cpu()
    vis = torch.empty(xy.shape[0], dtype=torch.float32)
    _mesh_lib().mesh_vertex_visibility(_fp(xy), _fp(z), ctypes.c_int(xy.shape[0]), _fp(f), ctypes.c_int(f.shape[0]),
                                       ctypes.c_int(size), _fp(vis), ctypes.c_void_p(0))
    return vis[:, None]


def cal_vis_sdf_batch(verts, faces, points, vert_xy, vert_z):
    """src/lib/dataset/mesh_util.py:498-524.  verts (1,NV,3), faces (1,NF,3) long, points (1,N,3),
    vert_xy (1,NV,2) in [0,1], vert_z (1,NV,1) -> sdf (1,N), vis (1,N,1) bool, vert_vis (1,NV,1), closest_face (1,N,3)."""
    V = verts[0].detach().float().contiguous().cpu()
    Fc = faces[0].detach().to(torch.int32).contiguous().cpu()
    P = points[0].detach().float().contiguous().cpu()
    vert_vis = vertex_visibility(vert_xy[0], vert_z[0], Fc)
    n = P.shape[0]
    sdf = torch.empty(n, dtype=torch.float32)
    vis = torch.empty(n, dtype=torch.uint8)
    face = torch.empty(n, dtype=torch.int32)
    _mesh_lib().mesh_query(_fp(V), ctypes.c_int(V.shape[0]), _fp(Fc), ctypes.c_int(Fc.shape[0]),
                           _fp(vert_vis.contiguous()), _fp(P), ctypes.c_int64(n), _fp(sdf), _fp(vis), _fp(face))
    closest_face = Fc.long()[face.long()]
    return sdf[None], vis.bool()[None, :, None], vert_vis[None], closest_face[None]


# --------------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------------
def feat_sample(feat, uv):
    """src/utils.py:136-151: bilinear, border padding, align_corners=True.  (B,C,H,W),(B,N,2) -> (B,N,C)."""
    out = F.grid_sample(feat, uv[:, :, None], mode="bilinear", padding_mode="border", align_corners=True)
    return out.view(*out.shape[:2], -1).permute(0, 2, 1)


def spatial_encode(v, kpt3d, extrin, sp_level=3, scale=1.0, sigma=0.1):
    """SpatialEncoder 'rel_z_decay' (src/spatial.py:59-84, 109-117, 20-43).  v (B,N,3), kpt3d (B,K,3) -> (B,N,(1+2L)K)."""
    R, t = extrin[:, :3, :3], extrin[:, :3, 3]
    cxyz = v @ R.transpose(1, 2) + t[:, None]
    kxyz = kpt3d @ R.transpose(1, 2) + t[:, None]
    dz = scale * (cxyz[:, :, None, 2:3] - kxyz[:, None, :, 2:3])
    dxyz = cxyz[:, :, None] - kxyz[:, None, :]
    w = torch.exp(-(dxyz ** 2).sum(-1, keepdim=True) / (2.0 * (sigma ** 2)))
    w = w.view(*w.shape[:2], -1)
    x = dz.view(*dz.shape[:2], -1)  # (B,N,K)
    freq = torch.from_numpy(np.asarray([np.pi * (2 ** l) for l in range(sp_level)], dtype=np.float32))  # pe_vector, scale=1.0 default
    y = x[:, :, None, :] * freq[None, None, :, None]
    pe = torch.cat((torch.sin(y), torch.cos(y)), -1).view(x.shape[0], x.shape[1], -1)
    out = torch.cat([x, pe], -1)
    out = out.view(*out.shape[:2], -1, w.shape[-1]) * w[:, :, None]
    return out.view(*out.shape[:2], -1)


def position_embedding(x, nlevels, scale=1.0):
    """src/spatial.py:20-43."""
    if nlevels <= 0:
        return x
    vec = torch.from_numpy(np.asarray([scale * np.pi * (2 ** l) for l in range(nlevels)], dtype=np.float32))
    B, N, _ = x.shape
    y = x[:, :, None, :] * vec[None, None, :, None]
    z = torch.cat((torch.sin(y), torch.cos(y)), -1).view(B, N, -1)
    return torch.cat([x, z], -1)


def _softplus100(x):
    return F.softplus(x, beta=100, threshold=20)  # src/utils.py:656


def _wn_linear(sd, prefix, x):
    """weight-normed Linear (src/utils.py:670-685): W = g * v / ||v||_row."""
    if prefix + ".weight_v" in sd:
        v, g = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]
        W = v * (g / v.norm(2, dim=1, keepdim=True))
    else:
        W = sd[prefix + ".weight"]
    return F.linear(x, W, sd[prefix + ".bias"])


def _conv1(sd, key, x):
    """bias-free Conv1d(k=1) on (B,N,C) tensors."""
    return F.linear(x, sd[key][:, :, 0])


def knn_vis(query, vert, vert_feat, vert_vis):
    """KNN_vis (src/networks.py:27-33), K=1; batch-0 indices are used for every batch."""
    idx = knn1(query[0], vert[0])
    feat_knn = vert_feat[:, idx] * vert_vis[:, idx]
    feat_toh = torch.cat([vert_feat[:, NUM_V:], vert_feat[:, :NUM_V]], 1)
    vis_toh = torch.cat([vert_vis[:, NUM_V:], vert_vis[:, :NUM_V]], 1)
    feat_knn_toh = feat_toh[:, idx] * vis_toh[:, idx]
    return feat_knn, feat_knn_toh, vert_vis[:, idx], vis_toh[:, idx], idx


def geo_vis_fusion(sd, vert_xy, fg, feat_sampled, vert, v, vert_vis, query_vis, query_sdf, pre="geo_vis_fusion."):
    """GeoVisFusion.forward (src/networks.py:75-106).  feat_sampled: [(B,1,N,64),(B,1,N,8)] -> same shapes."""
    B = vert_xy.shape[0]
    out = []
    for i, (at, ated) in enumerate((("fconv_at", "fconv_ated"), ("fconv_at1", "fconv_ated1"))):
        vfeat = feat_sample(fg[i], vert_xy)
        knn, toh, vis_th, vis_toh, _ = knn_vis(v, vert, vfeat, vert_vis)
        pix = feat_sampled[i].squeeze(1)
        tail = [query_sdf, query_vis, vis_th, vis_toh]
        f = torch.cat([pix, knn, toh] + tail, 2).float()
        a = torch.sigmoid(_conv1(sd, pre + at + ".2.weight", torch.relu(_conv1(sd, pre + at + ".0.weight", f))))
        g = torch.cat([pix * a[:, :, 0:1], knn * a[:, :, 1:2], toh * a[:, :, 2:3]] + tail, 2).float()
        h = _conv1(sd, pre + ated + ".2.weight", torch.relu(_conv1(sd, pre + ated + ".0.weight", g)))
        out.append(h.view(B, 1, *h.shape[-2:]))
    return out


def mlp_geo(sd, y, f, a, w, pre="mlp_geo."):
    """MLPUNetFusion.forward (src/utils.py:633-649) with MLPUNet (822-852), PoolModule (744-779), pool_ops (854-880),
    MLP (709-719) for the shipped config: n_dims1 [294,128,128,120,64], skip_layers [0,2], pool mean+var, n_dims2 [128,64,64,2]."""
    x = y
    skip = {0: 0, 2: 1}
    n1 = 4
    for i in range(n1):
        if i in skip:
            x = torch.cat([x, f[skip[i]]], -1)
        x = _wn_linear(sd, f"{pre}layers1.layers.{i}.linear", x)
        if i != n1 - 1:
            x = _softplus100(x)
    x_view = x
    a_sum = a.sum(1)
    mean = (w * x_view).sum(1)
    var = (w * (x_view - mean[:, None]).pow(2.0)).sum(1)
    x_pool = torch.cat([mean, var], -1)
    valid = a_sum > 0.0
    x = x_pool
    for i in range(3):
        x = _wn_linear(sd, f"{pre}layers2.layers.{i}.linear", x)
        if i != 2:
            x = _softplus100(x)
    return x, valid, x_view, x_pool


def tex_vertex_features(sd, vert_xy, ft1, img, pre="tex_vis_fusion."):
    """Per-frame part of TexVisFusion.forward (src/networks.py:270-279): -> vert_feat (B,NV,29) = [img3|tex8|gf18]."""
    vert_feat = feat_sample(ft1, vert_xy)
    vert_img = feat_sample(img, vert_xy)
    vert_feat = torch.cat([vert_img, vert_feat], 2)

    def stack(x, name, hw):
        x = F.conv2d(x, sd[pre + name + ".0.weight"], padding=1)
        x = torch.relu(F.layer_norm(x, [hw, hw], sd[pre + name + ".1.weight"], sd[pre + name + ".1.bias"], 1e-6))
        x = F.conv2d(x, sd[pre + name + ".3.weight"], padding=1)
        x = torch.relu(F.layer_norm(x, [hw, hw], sd[pre + name + ".4.weight"], sd[pre + name + ".4.bias"], 1e-6))
        return F.adaptive_avg_pool2d(x, 3)

    gf = stack(ft1, "fconv3", ft1.shape[-1])
    gf = gf.reshape(*gf.shape[:2], -1)
    gf_img = stack(img, "fconv4", img.shape[-1])
    gf_img = gf_img.reshape(*gf_img.shape[:2], -1)
    gf = torch.cat([gf_img, gf], -1)  # (B,42,18)
    x = F.conv1d(gf, sd[pre + "fconv_gt.0.weight"], padding=1)
    x = torch.relu(F.layer_norm(x, [18], sd[pre + "fconv_gt.1.weight"], sd[pre + "fconv_gt.1.bias"], 1e-6))
    x = F.conv1d(x, sd[pre + "fconv_gt.3.weight"], padding=1)
    x = torch.relu(F.layer_norm(x, [18], sd[pre + "fconv_gt.4.weight"], sd[pre + "fconv_gt.4.bias"], 1e-6))
    return torch.cat([vert_feat, x], 2)


def tex_vis_fusion(sd, vert_feat29, ft_xy, vert, v, vert_vis, query_vis, img_xy, latent24, pre="tex_vis_fusion."):
    """Per-sample part of TexVisFusion.forward (src/networks.py:281-293) -> (B,N,40)."""
    knn, toh, vis_th, vis_toh, _ = knn_vis(v, vert, vert_feat29, vert_vis)
    knn_gf, toh_gf = knn[:, :, 11:], toh[:, :, 11:]
    knn, toh = knn[:, :, :11], toh[:, :, :11]
    q = torch.cat([img_xy, ft_xy], 2)
    tail = [query_vis, vis_th, vis_toh]
    y = torch.cat([q, knn, toh, knn_gf, toh_gf, latent24] + tail, 2).float()
    a = torch.sigmoid(_conv1(sd, pre + "fconv_at.2.weight", torch.relu(_conv1(sd, pre + "fconv_at.0.weight", y))))
    g = torch.cat([q * a[:, :, 0:1], knn * a[:, :, 1:2], toh * a[:, :, 2:3], knn_gf * a[:, :, 3:4],
                   toh_gf * a[:, :, 4:5], latent24 * a[:, :, 5:6]] + tail, 2)
    return _conv1(sd, pre + "fconv.2.weight", torch.relu(_conv1(sd, pre + "fconv.0.weight", g)))


def ibr_head(sd, rgb_feats, ray_diffs, proj_mask, pre="mlp_tex."):
    """IBRRenderingHead.forward (src/model.py:1600-1636).  (R,S,V,40),(R,S,V,4),(R,S,V,1) -> (R,S,3)."""
    def lin(name, x):
        return F.linear(x, sd[pre + name + ".weight"], sd[pre + name + ".bias"])

    V = rgb_feats.shape[2]
    dir_feat = F.elu(lin("ray_encoder.2", F.elu(lin("ray_encoder.0", ray_diffs))))
    src_rgb = rgb_feats[..., :3]
    nd = dir_feat.shape[-1]
    rgb_feats = torch.cat((rgb_feats[..., :nd] + dir_feat, rgb_feats[..., nd:]), -1)
    dot_prod = ray_diffs[..., 3:4]
    e = torch.exp(torch.abs(sd[pre + "ani_al"]) * (dot_prod - 1))
    weight = (e - torch.min(e, dim=2, keepdim=True)[0]) * proj_mask
    weight = weight / (torch.sum(weight, dim=2, keepdim=True) + 1e-8)
    mean = torch.sum(rgb_feats * weight, dim=2, keepdim=True)  # fused_mean_variance, src/utils.py:153-157
    var = torch.sum(weight * (rgb_feats - mean) ** 2, dim=2, keepdim=True)
    fused = torch.cat([mean, var], -1)
    x = F.elu(lin("base_layer.2", F.elu(lin("base_layer.0", torch.cat([fused.expand(-1, -1, V, -1), rgb_feats], -1)))))
    pv = F.elu(lin("vis_layer1.2", F.elu(lin("vis_layer1.0", x * weight))))
    res, vis = pv[..., :-1], pv[..., -1:]
    x = x + res
    vis = torch.sigmoid(lin("vis_layer2.2", F.elu(lin("vis_layer2.0", x * torch.sigmoid(vis) * proj_mask)))) * proj_mask
    o = lin("out_layer.4", F.elu(lin("out_layer.2", F.elu(lin("out_layer.0", torch.cat([x, vis, ray_diffs], -1))))))
    o = o.masked_fill(proj_mask == 0, -1e4)
    return torch.sum(src_rgb * torch.softmax(o, dim=2), dim=2)


def sdf_activation(sd, x):
    """src/model.py:879-882 (the clamp is applied to a copy; the reference clamps the parameter in place)."""
    beta = torch.clamp(sd["sigmoid_beta"], min=2e-3)
    return torch.sigmoid(x / beta) / beta


def rgba2out(sd, rgba, z, vert_sdf):
    """src/model.py:1464-1494.  rgba (B,R,S,5)=[alpha,sdf,rgb], z (B,R,S), vert_sdf (B,R,S,1)."""
    alpha = sdf_activation(sd, -(rgba[..., 0] + vert_sdf.squeeze(-1)))
    sdf = rgba[..., 1]
    rgb = rgba[..., 2:]
    dist = torch.cat([(z[..., 1:] - z[..., :-1]), 1e10 * torch.ones_like(z[..., :1])], -1)
    contrib = 1.0 - torch.exp(-alpha * dist)
    contrib = contrib * torch.cumprod(torch.cat([torch.ones_like(contrib[..., :1]), 1 - contrib[..., :-1]], -1), -1)
    color = (rgb * contrib[..., None]).sum(-2)
    acc = contrib.sum(-1)
    sdf = (sdf * contrib).sum(-1) / (acc + 1e-8)
    depth = (z * contrib).sum(-1) / (acc + 1e-8)
    return color, depth, acc, contrib, sdf


def importance_sample(contrib, z, sample_per_ray, uniform=True, u=None, return_idx=False):
    """src/model.py:1424-1462.  contrib (B,R,D-2), z (B,R,D-1) -> (B,R,sample_per_ray).
    `u` replaces th.rand when uniform=False (training RNG drawn on the host)."""
    assert contrib.shape[-1] == z.shape[-1] - 1
    contrib = contrib + 1e-5
    # The reference's float .sum() has an implementation-defined order; accumulate in fp64 and round once so the
    # result (and the integer searchsorted indices that depend on it) is order independent.  CPU cumsum already
    # accumulates in fp64 (at::acc_type<float, /*is_cuda=*/false>) and rounds each prefix to fp32.
    pdf = contrib / contrib.double().sum(-1, keepdim=True).float()
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :, :1]), cdf], 2)
    if uniform:
        sample = torch.linspace(0.0, 1.0, steps=sample_per_ray)[None, None, :].expand(*cdf.shape[:-1], -1)
    else:
        sample = u
    idx = torch.searchsorted(cdf, sample.contiguous(), right=True)
    idx_prev = (idx - 1).clamp(min=0)
    idx = idx.clamp(max=cdf.shape[-1] - 1)
    cdf_prev, cdf_next = torch.gather(cdf, -1, idx_prev), torch.gather(cdf, -1, idx)
    z_prev, z_next = torch.gather(z, -1, idx_prev), torch.gather(z, -1, idx)
    num = sample - cdf_prev
    den = cdf_next - cdf_prev
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    out = z_prev + (num / den) * (z_next - z_prev)
    if return_idx:
        return out, idx_prev, idx
    return out


def ray_bbox_intersection(bounds, orig, direct, boffset=(-0.01, 0.01)):
    """src/model.py:1496-1570.  bounds (B,2,3), orig (B,1,3), direct (B,R,3) -> near,far (B,R,1), hit (B,R,1) bool."""
    nears, fars, hits = [], [], []
    for b in range(bounds.shape[0]):
        bd = bounds[b] + torch.tensor([boffset[0], boffset[1]])[:, None]
        d = direct[b].detach().clone()
        o = orig[b].expand(d.shape[0], -1)
        d[d.abs() < 1e-5] = 1e-5
        t = ((bd[None] - o[:, None]) / d[:, None]).reshape(-1, 6)
        p = t[..., None] * d[:, None] + o[:, None]
        lo, hi = bd[0], bd[1]
        eps = 1e-6
        inside = ((p[..., 0] >= lo[0] - eps) * (p[..., 0] <= hi[0] + eps) * (p[..., 1] >= lo[1] - eps) *
                  (p[..., 1] <= hi[1] + eps) * (p[..., 2] >= lo[2] - eps) * (p[..., 2] <= hi[2] + eps))
        hit = inside.sum(-1) == 2
        pts = p[hit][inside[hit]].reshape(-1, 2, 3)
        nr = torch.linalg.norm(d[hit], dim=1)
        d0 = torch.linalg.norm(pts[:, 0] - o[hit], dim=1) / nr
        d1 = torch.linalg.norm(pts[:, 1] - o[hit], dim=1) / nr
        near = torch.ones(d.shape[0])
        far = torch.ones(d.shape[0])
        near[hit] = torch.minimum(d0, d1)
        far[hit] = torch.maximum(d0, d1)
        nears.append(near[None, :, None])
        fars.append(far[None, :, None])
        hits.append(hit[None, :, None])
    return torch.cat(nears, 0), torch.cat(fars, 0), torch.cat(hits, 0)


# --------------------------------------------------------------------------------------------
# per-sample query (src/model.py:748-957) and the whole pass (src/model.py:1102-1422)
# --------------------------------------------------------------------------------------------
def apply_transf(xy, cam):
    """The optional 2-D affine behind the projection, cam['transf'] (B,2,3) (src/model.py:783-785, 848-850, 979-981, 1249-1251, 1261-1263)."""
    if "transf" in cam:
        t = cam["transf"]
        xy = xy @ t[:, :2, :2].transpose(1, 2) + t[:, :, 2][:, None]
    return xy


def project(pts, cam):
    """src/model.py:780-788: (B,N,3) -> xy in [-1,1] (B,N,2), z in [-1,1] (B,N,1)."""
    vh = pts @ cam["KRT"][:, :3, :3].transpose(1, 2) + cam["KRT"][:, :3, 3][:, None]
    z = vh[..., 2:3]
    xy = apply_transf(vh[..., :2] / z, cam)
    xy = torch.stack([2.0 * (xy[..., 0] / (cam["width"] - 1.0)) - 1.0, 2.0 * (xy[..., 1] / (cam["height"] - 1.0)) - 1.0], -1)
    z = 2.0 * (z - cam["znear"]) / (cam["zfar"] - cam["znear"]) - 1.0
    return xy, z


def project_verts(vert, cam):
    """src/model.py:845-853: vertices -> [-1,1] image coordinates (z + 1e-8 in the divide)."""
    vh = vert @ cam["KRT"][:, :3, :3].transpose(1, 2) + cam["KRT"][:, :3, 3][:, None]
    z = vh[..., 2:3]
    xy = apply_transf(vh[..., :2] / (z + 1e-8), cam)
    return torch.stack([2.0 * (xy[..., 0] / (cam["width"] - 1.0)) - 1.0, 2.0 * (xy[..., 1] / (cam["height"] - 1.0)) - 1.0], -1)


def query(sd, pts, cam, targets, feat_geo, feat_tex, vert_vis, query_vis, query_sdf, sp_data, img, view,
          fg_mask, sp_args=None, want=None):
    """VANeRF.query + query_color for n_views == 1 (src/model.py:748-957) -> out (B,N,5) = [sdf_pred, rad, r, g, b], valid (B,N,1).
    `want`: optional dict that receives named intermediates."""
    sp_args = sp_args or {"sp_level": 3, "scale": 1.0, "sigma": 0.1}
    B, N = pts.shape[:2]
    xy, z = project(pts, cam)
    eps = 1e-2
    mask_xy = (xy >= -1.0 - eps) & (xy <= 1.0 + eps)
    out_mask = (mask_xy[..., 0] & mask_xy[..., 1] & (z >= -1.0)[..., 0])[..., None].float()
    out_mask = out_mask.view(-1, 1, *out_mask.shape[1:])  # (B,V=1,N,1)
    fg = fg_mask.view(-1, 1, *fg_mask.shape[-2:]).float()
    fg_xy = feat_sample(fg, xy).view(-1, 1, N, 1)
    out_mask = out_mask * (fg_xy > 0.1).all(1, keepdim=True) * out_mask.bool().all(1, keepdim=True)
    xyz = 0.5 * torch.cat([xy, z], -1) + 0.5
    dist_b = torch.min(xyz, 1.0 - xyz)
    pw = torch.sigmoid(5.0 * (dist_b / 0.1 - 1.0))
    pw = (pw[..., 0] * pw[..., 1] * pw[..., 2]).view(-1, 1, N, 1) * out_mask
    pix_weight = pw / (pw.sum(1, keepdim=True) + 1e-6)
    feat_sampled = [feat_sample(f, xy).view(-1, 1, N, f.shape[1]) for f in feat_geo]
    y = spatial_encode(pts, sp_data["kpt3d"], sp_data["extrin"], sp_args["sp_level"], sp_args["scale"], sp_args["sigma"])
    y = y.view(-1, 1, *y.shape[1:])
    vert3d = targets["vert_world"]
    vert_xy = project_verts(vert3d, cam)
    vv = vert_vis.type(torch.int)
    fused = geo_vis_fusion(sd, vert_xy, feat_geo, feat_sampled, vert3d, pts, vv, query_vis, query_sdf.unsqueeze(-1))
    out, valid, x_view, latent = mlp_geo(sd, y, fused, out_mask, pix_weight)
    # query_color (src/model.py:884-957)
    img_xy = feat_sample(img, xy)
    ft_xy = feat_sample(feat_tex, xy)
    latent24 = F.linear(latent, sd["ibr_compress_gfeat.weight"], sd["ibr_compress_gfeat.bias"])
    vfeat29 = tex_vertex_features(sd, vert_xy, feat_tex, img)
    rgb_feat = tex_vis_fusion(sd, vfeat29, ft_xy, vert3d, pts, vv, query_vis, img_xy, latent24)
    cam_pos = torch.inverse(cam["KRT"].float())[:, :3, 3:4]
    cam_rays = F.normalize(pts - cam_pos.view(-1, 1, 3), p=2, dim=-1)
    ray_diff = view - cam_rays
    ray_dot = (cam_rays * view).sum(-1, keepdim=True)
    ray_diff = torch.cat([ray_diff / torch.clamp(torch.norm(ray_diff, dim=-1, keepdim=True), min=1e-6), ray_dot], -1)
    rgb = ibr_head(sd, rgb_feat.view(B * N, 1, 1, -1), ray_diff.view(B * N, 1, 1, 4), out_mask.view(B * N, 1, 1, 1)).view(B, N, 3)
    if want is not None:
        want.update(xy=xy, z=z, out_mask=out_mask, pix_weight=pix_weight, y=y, geo_fused0=fused[0], geo_fused1=fused[1],
                    x_view=x_view, latent=latent, latent24=latent24, vert_feat29=vfeat29, rgb_feat=rgb_feat,
                    ray_diff=ray_diff, vert_xy=vert_xy)
    return torch.cat([out, rgb], -1), valid


def eval_func(sd, rgba, mask, nml_scale, noise=None):
    """src/model.py:1140-1160 -> (B,N,5) = [alpha, sdf, r, g, b]."""
    mask = mask.float()
    sdf = mask * rgba[..., :1] + (1.0 - mask) * (0.1 / nml_scale)
    rad = rgba[..., 1:2]
    if noise is not None:
        rad = rad + noise
    alpha = mask * torch.relu(rad)
    return torch.cat([alpha, sdf, rgba[..., 2:]], -1)


def pixel_grid(width, height, level, stride_xy):
    """Eval branch of src/model.py:1191-1201: grids (1,R,2) int64, index (1,R) int64."""
    st = 2 ** (level - 1)
    yg, xg = torch.meshgrid(torch.arange(0, height, st), torch.arange(0, width, st), indexing="ij")
    grids = torch.stack([xg, yg], -1).view(-1, 2)[None] + stride_xy
    index = grids[..., 0] + grids[..., 1] * width
    return grids, index


def train_window(msk, pick, out_h, out_w, width, height):
    """Training branch of src/model.py:1172-1189: a out_h x out_w window of pixels centred on mask pixel number `pick` (the draw of
    np.random.randint(0, n_mask_pixels), model.py:1182), clamped to [0, min(W-1, H-1)].  msk (H,W) bool -> grids (1,R,2) int64, index (1,R)."""
    coords = torch.stack(torch.where(msk)[::-1], -1)
    centre = coords[pick:pick + 1] if coords.shape[0] > 0 else torch.zeros((1, 2), dtype=torch.int64)
    yg, xg = torch.meshgrid(torch.arange(0, out_h), torch.arange(0, out_w), indexing="ij")
    grids = torch.stack([xg, yg], -1).view(-1, 2) + (centre - out_h // 2)
    grids = grids.clamp(0, min(width - 1, height - 1))[None]
    return grids, grids[..., 0] + grids[..., 1] * width


def stratified(S, jitter):
    """Coarse depth fractions of src/model.py:1222-1230 with uniform=False: jitter (B,R,S) in [0,1) = the draw of th.rand_like(z)."""
    z = torch.linspace(0.0, 1.0, steps=S)[None, None, :].expand(*jitter.shape[:2], -1)
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
    z_lower = torch.cat([z[..., :1], z_mid], -1)
    z_upper = torch.cat([z_mid, z[..., -1:]], -1)
    return z_lower + jitter * (z_upper - z_lower)


def gt_gathers(index, out_h, out_w, tar_img=None, msk=None, fg_mask=None, img_in=None):
    """GT gathers of src/model.py:1361-1418 at the pixel index of the pass (source-sized tensors are indexed with the TARGET index, as
    the reference does): tar_img (B,3,H,W), msk (B,H,W), fg_mask (B,1,1,H,W) or (B,1,H,W), img_in (B,3,H,W)."""
    out = {}

    def g(t, ch):
        flat = t.reshape(t.shape[0], ch, -1)
        return torch.gather(flat, 2, index[:, None].expand(-1, ch, -1)).view(t.shape[0], ch, out_h, out_w)

    if tar_img is not None:
        out["tar_img"] = g(tar_img, 3)
        if msk is not None:
            out["tar_alpha"] = g(msk.reshape(msk.shape[0], 1, -1), 1).float()
    if fg_mask is not None:
        out["input_mask"] = g(fg_mask.reshape(fg_mask.shape[0], 1, *fg_mask.shape[-2:]), 1)
    if img_in is not None:
        out["img_in"] = g(img_in, 3)
    return out


def generate_rays(grids, cam_tar, bounds, znear, zfar):
    """src/model.py:1201-1220 -> cam_rays (B,R,3), cam_pos (B,1,3), znear_rays, zfar_rays (B,R,1), hit."""
    grids = grids.float()
    gh = torch.cat([grids, torch.ones_like(grids[..., :1])], -1)
    inv_K = torch.inverse(cam_tar["K"][:, :3, :3]).transpose(1, 2)
    cam_rays = torch.bmm(gh, inv_K)
    znear_rays = torch.norm(torch.bmm(znear * gh, inv_K), p=2, dim=-1, keepdim=True)
    zfar_rays = torch.norm(torch.bmm(zfar * gh, inv_K), p=2, dim=-1, keepdim=True)
    cam_rays = F.normalize(torch.bmm(cam_rays, cam_tar["RT"][:, :3, :3]), p=2, dim=-1)
    cam_pos = -torch.bmm(cam_tar["RT"][:, :3, 3][:, None], cam_tar["RT"][:, :3, :3])
    z1, z2, hit = ray_bbox_intersection(bounds, cam_pos, cam_rays)
    m1 = (hit & (z1 > znear_rays)).float()
    znear_rays = m1 * z1 + (1.0 - m1) * znear_rays
    m2 = (hit & (z2 < zfar_rays)).float()
    zfar_rays = m2 * z2 + (1.0 - m2) * zfar_rays
    return cam_rays, cam_pos, znear_rays, zfar_rays, hit


def source_vert_xyz01(vert3d, cam):
    """src/model.py:1245-1255: source-view vertex coordinates fed to the visibility rasteriser."""
    vh = vert3d @ cam["KRT"][:, :3, :3].transpose(1, 2) + cam["KRT"][:, :3, 3][:, None]
    vz = vh[..., 2:3]
    xy = apply_transf(vh[..., :2] / (vz + 1e-8), cam)
    xy = torch.stack([xy[..., 0] / (cam["width"] - 1.0), xy[..., 1] / (cam["height"] - 1.0)], -1)
    vz = (vz - cam["znear"]) / (cam["zfar"] - cam["znear"])
    return xy, vz


def batch_render(sd, frame, level, stride_xy, sample_per_ray_c=64, sample_per_ray_f=64, fine=True,
                 sp_args=None, want=None, grids=None, draws=None, tar_img=None, msk=None):
    """VANeRF.batch_render_pifu_nerf (src/model.py:1102-1422) for B = V = 1.  draws=None: the evaluation pass (uniform=True).
    draws = dict(jitter (1,R,Sc), u (1,R,Sf), noise_c (1,R*Sc,1), noise_f (1,R*(Sc+Sf),1), std): the training pass (uniform=False,
    rand_noise_std=std) with the random numbers the reference draws at model.py:1229, 1443, 1156 (coarse, then fine) passed in.
    tar_img / msk: adds the GT gathers of 1361-1418 (render_vis's images, 1377-1389, are discriminator input and not restated).
    Returns dict(tex_fg, depth, alpha, tex_fg_fine, depth_fine, alpha_fine, sdf, index, z, z_fine, ...)."""
    cam_in, cam_tar, targets = frame["cam_in"], frame["cam_tar"], frame["targets"]
    width = cam_tar.get("width", cam_in["width"])
    height = cam_tar.get("height", cam_in["height"])
    znear = cam_tar.get("znear", cam_in["znear"])
    zfar = cam_tar.get("zfar", cam_in["zfar"])
    st = 2 ** (level - 1)
    assert width % st == 0 and height % st == 0
    if grids is None:
        grids, index = pixel_grid(width, height, level, stride_xy)
        out_w, out_h = width // st, height // st
    else:  # explicit pixel list (tests at sizes / strides the reference's grid cannot express)
        index = grids[..., 0] + grids[..., 1] * width
        out_h, out_w = frame["out_hw"]
    cam_rays, cam_pos, znear_rays, zfar_rays, hit = generate_rays(grids, cam_tar, frame["bounds"], znear, zfar)
    S = sample_per_ray_c
    if draws is None:
        z = torch.linspace(0.0, 1.0, steps=S)[None, None, :].expand(*znear_rays.shape[:2], -1)
    else:
        z = stratified(S, draws["jitter"])
    z = znear_rays + (zfar_rays - znear_rays) * z
    vert3d = targets["vert_world"]
    face = targets["face_world"].long()
    vert_xy01, vert_z01 = source_vert_xyz01(vert3d, cam_in)
    img, fg_mask = frame["img_in"], frame["src_foreground_mask"]

    def march(zs, noise=None):
        S_ = zs.shape[-1]
        pts = (cam_pos[:, :, None] + cam_rays[:, :, None] * zs[..., None]).view(1, -1, 3)
        view = cam_rays[:, :, None, :].expand(-1, -1, S_, -1).reshape(1, -1, 3)
        q_sdf, q_vis, vert_vis, _ = cal_vis_sdf_batch(vert3d, face, pts, vert_xy01, vert_z01)
        w = {} if want is not None else None
        rgba, valid = query(sd, pts, cam_in, targets, frame["feat_geo"], frame["feat_tex"], vert_vis, q_vis, q_sdf,
                            frame["sp_data"], img, view, fg_mask, sp_args, w)
        rgba = eval_func(sd, rgba, valid, cam_in["nml_scale"], noise=None if noise is None else noise * draws["std"]).view(1, -1, S_, 5)
        q_sdf = q_sdf.view(1, -1, S_, 1)
        color, depth, alpha, contrib, sdf = rgba2out(sd, rgba, zs, q_sdf)
        return dict(pts=pts, q_sdf=q_sdf, q_vis=q_vis, vert_vis=vert_vis, rgba=rgba, color=color, depth=depth, alpha=alpha,
                    contrib=contrib, sdf=sdf, inter=w)

    c = march(z, None if draws is None else draws.get("noise_c"))
    out = {"tex_fg": c["color"].view(1, out_h, out_w, 3).permute(0, 3, 1, 2), "depth": c["depth"].view(1, out_h, out_w),
           "alpha": c["alpha"].view(1, out_h, out_w), "index": index, "z": z, "hit": hit, "cam_rays": cam_rays,
           "cam_pos": cam_pos, "vert_vis": c["vert_vis"], "coarse": c}
    if fine:
        z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
        if draws is None:
            z_new = importance_sample(c["contrib"][..., 1:-1], z_mid, sample_per_ray_f, uniform=True)
        else:
            z_new = importance_sample(c["contrib"][..., 1:-1], z_mid, sample_per_ray_f, uniform=False, u=draws["u"])
        z_fine = torch.sort(torch.cat([z, z_new], -1), -1)[0]
        f = march(z_fine, None if draws is None else draws.get("noise_f"))
        out.update({"tex_fg_fine": f["color"].view(1, out_h, out_w, 3).permute(0, 3, 1, 2),
                    "depth_fine": f["depth"].view(1, out_h, out_w), "alpha_fine": f["alpha"].view(1, out_h, out_w),
                    "sdf": f["sdf"].view(1, out_h, out_w), "z_fine": z_fine, "z_new": z_new, "fine": f})
    if tar_img is not None:
        out.update(gt_gathers(index, out_h, out_w, tar_img, msk, fg_mask, img))
    return out


def psnr(a, b):
    """src/evaluator.py:16-19."""
    mse = torch.mean((a - b) ** 2)
    return float(-10.0 * torch.log10(mse)) if mse > 0 else math.inf
