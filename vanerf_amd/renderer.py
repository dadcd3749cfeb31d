"""Host side of the MI355X render path: torch tensors in, C-ABI calls into libvanerf_hip.so, torch tensors out.

PyTorch is used here for device memory, streams and the per-frame (not per-sample) preparation
that the north-star keeps on PyTorch-ROCm (feature un-projection at the mesh vertices, the
TexVisFusion per-frame conv stack).  Every per-ray / per-sample operation runs in the HIP library.
There is no CPU or eager fallback: every entry point requires CUDA(ROCm) tensors.
"""
import ctypes
from ctypes import byref, c_float, c_int64, c_uint, c_void_p

import torch
import torch.nn.functional as F

from . import _ffi
from ._ffi import VanerfFrame, VanerfMeshAccel, VanerfPassDesc, VanerfPassOut, VanerfWeightTable, check, lib

NV, NV_HAND, NKPT = 1558, 779, 42


def _ptr(t, dtype=None, cuda=True):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if cuda and not t.is_cuda:
        raise ValueError("the HIP render path needs device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _farr(values, n):
    vals = [float(v) for v in values]
    assert len(vals) == n, (len(vals), n)
    return (c_float * n)(*vals)


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
_CONV_KEYS = {
    "geo_at0_w1": "geo_vis_fusion.fconv_at.0.weight", "geo_at0_w2": "geo_vis_fusion.fconv_at.2.weight",
    "geo_ated0_w1": "geo_vis_fusion.fconv_ated.0.weight", "geo_ated0_w2": "geo_vis_fusion.fconv_ated.2.weight",
    "geo_at1_w1": "geo_vis_fusion.fconv_at1.0.weight", "geo_at1_w2": "geo_vis_fusion.fconv_at1.2.weight",
    "geo_ated1_w1": "geo_vis_fusion.fconv_ated1.0.weight", "geo_ated1_w2": "geo_vis_fusion.fconv_ated1.2.weight",
    "tex_at_w1": "tex_vis_fusion.fconv_at.0.weight", "tex_at_w2": "tex_vis_fusion.fconv_at.2.weight",
    "tex_w1": "tex_vis_fusion.fconv.0.weight", "tex_w2": "tex_vis_fusion.fconv.2.weight",
    "l1_w3": "mlp_geo.layers1.layers.3.linear.weight", "l1_b3": "mlp_geo.layers1.layers.3.linear.bias",
    "l2_w2": "mlp_geo.layers2.layers.2.linear.weight", "l2_b2": "mlp_geo.layers2.layers.2.linear.bias",
    "ibr_w": "ibr_compress_gfeat.weight", "ibr_b": "ibr_compress_gfeat.bias",
}
_EXPECT = {
    "geo_at0_w1": (10, 196), "geo_at0_w2": (3, 10), "geo_ated0_w1": (64, 196), "geo_ated0_w2": (64, 64),
    "geo_at1_w1": (10, 28), "geo_at1_w2": (3, 10), "geo_ated1_w1": (8, 28), "geo_ated1_w2": (8, 8),
    "tex_at_w1": (96, 96), "tex_at_w2": (6, 96), "tex_w1": (96, 96), "tex_w2": (40, 96),
    "l1_w3": (64, 120), "l1_b3": (64,), "l2_w2": (2, 64), "l2_b2": (2,), "ibr_w": (24, 128), "ibr_b": (24,),
}
_L1 = [(128, 358), (128, 128), (120, 136)]
_L2 = [(64, 128), (64, 64)]


def weight_table(sd, on_device=False):
    """Reference state_dict (key names of VANeRF.state_dict()) -> (VanerfWeightTable, keep-alive list of the tensors it points into).
    on_device: the table points at the parameters' own device storage (vanerf_weights_update; no copy) instead of host copies."""
    keep = []

    def host(key, shape):
        t = sd[key].detach()
        if on_device:
            n = 1
            for d in shape:
                n *= d
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
                raise ValueError(f"{key}: the device packer takes contiguous fp32 device tensors of {n} elements")
            keep.append(t)
            return c_void_p(t.data_ptr())
        t = t.to("cpu", torch.float32)
        if t.dim() == 3 and t.shape[-1] == 1:
            t = t[:, :, 0]
        t = t.reshape(shape).contiguous()
        keep.append(t)
        return c_void_p(t.data_ptr())

    tab = VanerfWeightTable()
    for field, key in _CONV_KEYS.items():
        setattr(tab, field, host(key, _EXPECT[field]))
    for i, (o, k) in enumerate(_L1):
        p = f"mlp_geo.layers1.layers.{i}.linear."
        tab.l1_v[i], tab.l1_g[i], tab.l1_b[i] = host(p + "weight_v", (o, k)), host(p + "weight_g", (o,)), host(p + "bias", (o,))
    for i, (o, k) in enumerate(_L2):
        p = f"mlp_geo.layers2.layers.{i}.linear."
        tab.l2_v[i], tab.l2_g[i], tab.l2_b[i] = host(p + "weight_v", (o, k)), host(p + "weight_g", (o,)), host(p + "bias", (o,))
    if on_device:
        tab.sigmoid_beta = 0.0  # taken from the device (PackedWeights.update)
        keep.append(sd["sigmoid_beta"].detach())
    else:
        tab.sigmoid_beta = float(sd["sigmoid_beta"].detach().reshape(-1)[0])
    return tab, keep


# Arithmetic of the per-sample dense layers (vanerf_weights_pack `mode`, include/vanerf_hip.h):
#   "fp32"   v_mfma_f32_32x32x2_f32 on the fp32 weights and activations (1e-5 of the fp32 oracle: summation order only)
#   "bf16x3" v_mfma_f32_32x32x16_bf16 on weights and activations split into two bf16 parts each, three products, fp32
#            accumulate (3e-5 of the "fp32" kernel on the outputs; about twice as fast)
PRECISIONS = {"fp32": 0, "bf16x3": 1}


class PackedWeights:
    """Device-resident MFMA-fragment copy of the per-sample network weights (vanerf_weights_pack)."""

    def __init__(self, sd, mode=0):
        mode = PRECISIONS.get(mode, mode)
        if mode not in (0, 1):
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.mode = mode
        tab, keep = weight_table(sd)
        h = c_void_p()
        check(lib.vanerf_weights_pack(byref(tab), mode, byref(h)))
        self.handle = h
        self.device_index = torch.cuda.current_device() if torch.cuda.is_available() else None  # the handle lives on the device current at the pack
        self._beta = max(float(tab.sigmoid_beta), 2e-3)  # sdf_activation clamp (src/model.py:880)
        self._beta_src = None

    @property
    def beta(self):
        """sigmoid_beta as the handle carries it (the per-stage debug path passes it by value; after an update on the device this read blocks)."""
        if self._beta_src is not None:
            self._beta = max(float(self._beta_src.reshape(-1)[0]), 2e-3)
        return self._beta

    def update(self, sd):
        """Re-pack in place from parameters on the device (vanerf_weights_update): no host copies, nothing blocks, bit-identical to a fresh pack.
        Ordered on the current stream after whatever still reads the old weights."""
        tab, keep = weight_table(sd, on_device=True)
        beta = keep[-1]
        if not (beta.is_cuda and beta.dtype == torch.float32):
            raise ValueError("sigmoid_beta: the device packer takes an fp32 device tensor")
        check(lib.vanerf_weights_update(self.handle, byref(tab), c_void_p(beta.data_ptr()), _stream()))
        self._beta_src = beta
        return self

    def short_groups(self):
        """Groups of 32 samples (since packing) whose query took the all-invalid short path (blocking read; diagnostics)."""
        n = ctypes.c_uint64()
        check(lib.vanerf_weights_short_groups(self.handle, byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "handle", None) and lib is not None:  # `lib` is None during interpreter shutdown
            lib.vanerf_weights_free(self.handle)
            self.handle = None

    __del__ = close


def stream_host(sd, which):
    """Host-only packed stream `which` (0 fp32 forward, 1 bf16x3 forward, 2 backward) as raw int32 words (tests)."""
    tab, keep = weight_table(sd)
    n = c_int64()
    check(lib.vanerf_weights_stream_host(byref(tab), which, None, 0, byref(n)))
    out = torch.empty(n.value, dtype=torch.float32)
    check(lib.vanerf_weights_stream_host(byref(tab), which, c_void_p(out.data_ptr()), n.value, byref(n)))
    return out.view(torch.int32)


def stream_device(weights, which):
    """The stream a handle holds on the device (0 forward of its mode, 2 backward), copied back, as raw int32 words (tests; blocking)."""
    n = c_int64()
    check(lib.vanerf_weights_download(weights.handle, which, None, 0, byref(n)))
    out = torch.empty(n.value, dtype=torch.float32)
    check(lib.vanerf_weights_download(weights.handle, which, c_void_p(out.data_ptr()), n.value, byref(n)))
    return out.view(torch.int32)


def pack_weights_host(sd):
    """CPU-only view of the packed stream (tests of the packer): (float32 tensor, offsets list)."""
    tab, keep = weight_table(sd)
    n = c_int64()
    offs = (c_uint * _ffi.NUM_LAYERS)()
    check(lib.vanerf_weights_pack_host(byref(tab), None, 0, byref(n), offs))
    out = torch.empty(n.value, dtype=torch.float32)
    check(lib.vanerf_weights_pack_host(byref(tab), c_void_p(out.data_ptr()), n.value, byref(n), offs))
    return out, list(offs)


# ------------------------------------------------------------------------------------------------
# per-frame preparation (torch on the device; src/model.py:845-853, src/networks.py:83, 96, 270-279)
# ------------------------------------------------------------------------------------------------
def _grid_sample(feat, uv):
    out = F.grid_sample(feat, uv[:, :, None], mode="bilinear", padding_mode="border", align_corners=True)
    return out.view(*out.shape[:2], -1).permute(0, 2, 1)


def _project(pts, KRT):
    vh = pts @ KRT[:, :3, :3].transpose(1, 2) + KRT[:, :3, 3][:, None]
    return vh[..., :2], vh[..., 2:3]


_POOL3 = {}


def _avg_pool3(x):
    """AdaptiveAvgPool2d(3) of (B,C,H,W) as two small GEMMs with the bins' averaging matrix (bin i = [floor(i n / 3), ceil((i+1) n / 3)), the
    bins overlap when 3 does not divide n): torch's adaptive_average_pool kernel takes 0.33 ms per 779-channel map, this 0.05 ms."""
    def bins(n, dev):
        key = (n, dev)
        if key not in _POOL3:
            m = torch.zeros(n, 3)
            for i in range(3):
                a, b = (i * n) // 3, -((-(i + 1) * n) // 3)
                m[a:b, i] = 1.0 / (b - a)
            _POOL3[key] = m.to(dev)
        return _POOL3[key]

    return bins(x.shape[-2], x.device).t() @ (x @ bins(x.shape[-1], x.device))


def tex_global_vertex_feature(sd, feat_tex, img, pre="tex_vis_fusion."):
    """TexVisFusion's per-frame global feature (src/networks.py:273-278): (B,1558,18)."""
    def conv3x3(x, w):
        # Conv2d(k=3, padding=1, bias=False) as im2col + one GEMM (rocBLAS): for these 3 / 8 / 21-channel, 64^2 / 256^2 maps MIOpen's
        # solver search lands on naive_conv_ab_nonpacked_fwd_nchw (2.3 ms per call in the first frame's search, profiles/r01_i_*)
        B, C, H, W = x.shape
        cols = F.unfold(x, 3, padding=1)  # (B, C*9, H*W), channel-major then kernel position: the layout of w.reshape(O, C*9)
        return (w.reshape(w.shape[0], -1) @ cols).view(B, w.shape[0], H, W)

    def stack(x, name, hw):
        x = conv3x3(x, sd[pre + name + ".0.weight"])
        x = torch.relu(F.layer_norm(x, [hw, hw], sd[pre + name + ".1.weight"], sd[pre + name + ".1.bias"], 1e-6))
        x = conv3x3(x, sd[pre + name + ".3.weight"])
        x = torch.relu(F.layer_norm(x, [hw, hw], sd[pre + name + ".4.weight"], sd[pre + name + ".4.bias"], 1e-6))
        return _avg_pool3(x)

    gf = stack(feat_tex, "fconv3", feat_tex.shape[-1]).reshape(feat_tex.shape[0], 42, -1)
    gf_img = stack(img, "fconv4", img.shape[-1]).reshape(img.shape[0], 42, -1)
    gf = torch.cat([gf_img, gf], -1)
    def conv1d_k3(x, w):
        # Conv1d(k=3, padding=1) over the 18-long axis as one GEMM (rocBLAS): MIOpen falls back to a naive kernel (1.6 ms per call,
        # 48 calls while it searches) for these 42->779->1558-channel, length-18 convolutions
        xp = F.pad(x, (1, 1))
        cols = torch.stack([xp[..., 0:-2], xp[..., 1:-1], xp[..., 2:]], 2).reshape(x.shape[0], -1, x.shape[-1])  # (B, C*3, L)
        return w.reshape(w.shape[0], -1) @ cols

    x = conv1d_k3(gf, sd[pre + "fconv_gt.0.weight"])
    x = torch.relu(F.layer_norm(x, [18], sd[pre + "fconv_gt.1.weight"], sd[pre + "fconv_gt.1.bias"], 1e-6))
    x = conv1d_k3(x, sd[pre + "fconv_gt.3.weight"])
    return torch.relu(F.layer_norm(x, [18], sd[pre + "fconv_gt.4.weight"], sd[pre + "fconv_gt.4.bias"], 1e-6))


class MeshAccel:
    """Per-frame acceleration structure of vanerf_mesh_query_accel: Morton-sorted triangle / vertex clusters with their bounds (closest-face
    and 1-NN searches) and a (y,z) cell grid (inside test), built on the device by vanerf_mesh_accel_build -- seven small launches on the
    current stream, no host synchronisation (0.1 ms; the torch-op builder it replaces took 1.9 ms per source frame)."""
    CL = lib.vanerf_mesh_cluster_size()  # triangles / vertices per cluster the library was built with

    def __init__(self, verts3, faces_i32, grid=64, cell_capacity=None):
        if not (verts3.is_cuda and verts3.dtype == torch.float32 and verts3.is_contiguous() and faces_i32.dtype == torch.int32 and faces_i32.is_contiguous()):
            raise ValueError("MeshAccel needs contiguous device tensors: verts (NV,3) fp32, faces (NF,3) int32")
        nv, nf = verts3.shape[0], faces_i32.shape[0]
        cap = int(cell_capacity) if cell_capacity is not None else 64 * nf
        nbytes = lib.vanerf_mesh_accel_bytes(nv, nf, int(grid), cap)
        check(min(int(nbytes), 0))
        self.tables = torch.empty(int(nbytes), dtype=torch.uint8, device=verts3.device)  # every pointer of self.c points into this block
        self.verts3, self.faces = verts3, faces_i32
        self.c = VanerfMeshAccel()
        check(lib.vanerf_mesh_accel_build(_ptr(verts3, torch.float32), nv, _ptr(faces_i32, torch.int32), nf, int(grid), cap, _ptr(self.tables),
                                          int(nbytes), byref(self.c), _stream()))

    def table(self, name, dtype, shape):
        """One table of the block as a tensor view (tests, diagnostics)."""
        off = getattr(self.c, name) - self.tables.data_ptr()
        n = int(torch.tensor(shape).prod()) * torch.empty((), dtype=dtype).element_size()
        return self.tables[off:off + n].view(dtype).view(*shape)


class FrameData:
    """Everything the per-sample kernel needs about one source frame, resident in HBM (struct VanerfFrame)."""

    def __init__(self, sd, img, feat_geo, feat_tex, fg_mask, cam_in, targets, sp_data, sp_args=None, gf=None):
        sp_args = sp_args or {"scale": 1.0, "sigma": 0.1}
        dev = img.device
        if not img.is_cuda:
            raise ValueError("FrameData needs device tensors (no CPU fallback)")
        if img.shape[0] != 1 or cam_in["KRT"].shape[0] != 1:
            raise AssertionError("n_views == 1 and batch == 1 (src/networks.py:86,94 is single-view)")
        f32 = torch.float32
        KRT = cam_in["KRT"].to(dev, f32)
        verts = targets["vert_world"].to(dev, f32)
        faces = targets["face_world"].to(dev).long()
        assert verts.shape == (1, NV, 3), verts.shape
        if faces.numel() == 0:
            raise ValueError("face_world is empty")
        ext = sp_data["extrin"].to(dev, f32)
        # one read-back for everything the host needs from the device: both camera matrices (they go to the kernels by value) and the range of
        # the face indices (the mesh kernels index the vertex table with them)
        host = torch.cat([KRT[0, :3, :4].reshape(-1), ext[0, :3, :4].reshape(-1), faces.min().to(f32)[None], faces.max().to(f32)[None]]).tolist()
        if host[24] < 0 or host[25] >= NV:
            raise ValueError("face_world holds vertex indices outside [0, 1558)")
        W, H = float(cam_in["width"]), float(cam_in["height"])
        znear, zfar = float(cam_in["znear"]), float(cam_in["zfar"])
        # vertices in the source view (src/model.py:845-853 for sampling, 1245-1255 for the visibility raster)
        vxy, vz = _project(verts, KRT)
        vxy = vxy / (vz + 1e-8)
        vert_uv = torch.stack([2.0 * (vxy[..., 0] / (W - 1.0)) - 1.0, 2.0 * (vxy[..., 1] / (H - 1.0)) - 1.0], -1)
        self.vert_xy01 = torch.stack([vxy[..., 0] / (W - 1.0), vxy[..., 1] / (H - 1.0)], -1)[0].contiguous()
        self.vert_z01 = ((vz - znear) / (zfar - znear))[0, :, 0].contiguous()
        self.faces = faces[0].to(torch.int32).contiguous()
        self.verts3 = verts[0].contiguous()
        self.vert_vis = vertex_visibility(self.vert_xy01, self.vert_z01, self.faces)
        self.accel = MeshAccel(self.verts3, self.faces)
        vis = self.vert_vis[None, :, None]
        # per-vertex features, pre-multiplied by visibility (KNN_vis, src/networks.py:29-32)
        img = img.to(f32)
        self.vfeat0 = (_grid_sample(feat_geo[0], vert_uv) * vis)[0].contiguous()
        self.vfeat1 = (_grid_sample(feat_geo[1], vert_uv) * vis)[0].contiguous()
        if gf is None:
            gf = tex_global_vertex_feature(sd, feat_tex, img)
        vt = torch.cat([_grid_sample(img, vert_uv), _grid_sample(feat_tex, vert_uv), gf, torch.zeros(1, NV, 3, device=dev)], 2) * vis
        self.vfeat_tex = vt[0].contiguous()
        assert self.vfeat0.shape == (NV, 64) and self.vfeat1.shape == (NV, 8) and self.vfeat_tex.shape == (NV, 32)
        # channel-last maps
        self.geo0 = feat_geo[0][0].permute(1, 2, 0).contiguous()
        self.geo1 = feat_geo[1][0].permute(1, 2, 0).contiguous()
        self.tex = feat_tex[0].permute(1, 2, 0).contiguous()
        self.img = torch.cat([img[0], torch.zeros_like(img[0, :1])], 0).permute(1, 2, 0).contiguous()
        self.mask = fg_mask.reshape(fg_mask.shape[-2], fg_mask.shape[-1]).to(dev, f32).contiguous()
        assert self.geo0.shape[-1] == 64 and self.geo1.shape[-1] == 8 and self.tex.shape[-1] == 8
        assert self.mask.shape == self.img.shape[:2]
        self.verts4 = torch.cat([verts[0], torch.zeros(NV, 1, device=dev)], 1).contiguous()
        kpt = sp_data["kpt3d"].to(dev, f32)
        assert kpt.shape == (1, NKPT, 3), kpt.shape
        kc = kpt @ ext[:, :3, :3].transpose(1, 2) + ext[:, :3, 3][:, None]
        self.kpt_cam = torch.cat([kc[0], torch.zeros(NKPT, 1, device=dev)], 1).contiguous()

        c = VanerfFrame()
        c.geo0, c.geo1, c.tex, c.img, c.mask = (_ptr(t, f32) for t in (self.geo0, self.geo1, self.tex, self.img, self.mask))
        c.h0, c.w0 = self.geo0.shape[:2]
        c.h1, c.w1 = self.geo1.shape[:2]
        c.ht, c.wt = self.tex.shape[:2]
        c.hi, c.wi = self.img.shape[:2]
        c.verts, c.vfeat0, c.vfeat1, c.vfeat_tex = (_ptr(t, f32) for t in (self.verts4, self.vfeat0, self.vfeat1, self.vfeat_tex))
        c.vert_vis, c.kpt_cam = _ptr(self.vert_vis, f32), _ptr(self.kpt_cam, f32)
        c.KRT = _farr(host[:12], 12)
        c.extrin = _farr(host[12:24], 12)
        c.width, c.height, c.znear, c.zfar = W, H, znear, zfar
        c.invalid_sdf = 0.1 / float(cam_in["nml_scale"])
        c.pe_scale = float(sp_args.get("scale", 1.0))
        c.pe_inv_2sigma2 = 1.0 / (2.0 * float(sp_args.get("sigma", 0.1)) ** 2)
        self.c = c


# ------------------------------------------------------------------------------------------------
# thin wrappers over the C ABI (one per entry point)
# ------------------------------------------------------------------------------------------------
def vertex_visibility(vert_xy01, vert_z01, faces_i32, raster=256):
    """get_visibility (mesh_util.py:284-318) -> (NV,) float {0,1}."""
    nv = vert_xy01.shape[0]
    vis = torch.empty(nv, dtype=torch.float32, device=vert_xy01.device)
    scratch = torch.empty(raster * raster, dtype=torch.int32, device=vert_xy01.device)
    check(lib.vanerf_vertex_visibility(_ptr(vert_xy01, torch.float32), _ptr(vert_z01, torch.float32), nv, _ptr(faces_i32, torch.int32),
                                       faces_i32.shape[0], raster, _ptr(scratch), _ptr(vis), _stream()))
    return vis


def mesh_query(verts3, faces_i32, vert_vis, pts, want_face=False):
    """cal_vis_sdf_batch (mesh_util.py:498-524) -> sdf (N,), vis (N,) uint8[, closest face (N,) int32]."""
    n = pts.shape[0]
    sdf = torch.empty(n, dtype=torch.float32, device=pts.device)
    vis = torch.empty(n, dtype=torch.uint8, device=pts.device)
    face = torch.empty(n, dtype=torch.int32, device=pts.device) if want_face else None
    check(lib.vanerf_mesh_query(_ptr(verts3, torch.float32), verts3.shape[0], _ptr(faces_i32, torch.int32), faces_i32.shape[0],
                                _ptr(vert_vis, torch.float32), _ptr(pts, torch.float32), n, _ptr(sdf), _ptr(vis), _ptr(face), _stream()))
    return (sdf, vis, face) if want_face else (sdf, vis)


def _queue_word(dev):
    """Eight bytes of device memory for the work-queue head of ONE launch (vanerf_mesh_query_accel / vanerf_query_samples zero it on the stream).
    A fresh block per call: the caching allocator hands it out again only to later work of the same stream, so no two launches in flight share one."""
    return torch.empty(2, dtype=torch.int32, device=dev)


def mesh_query_accel(accel, verts3, faces_i32, vert_vis, pts, want_face=False, want_knn=True, grid=None):
    """Same results as mesh_query (+ knn1), through the per-frame acceleration structure: sdf, vis[, face][, knn].
    grid = (nx, ny, S): pts are the samples of an nx x ny ray grid, S per ray, ray-major (speed hint only)."""
    gnx, gny, gs = grid if grid is not None else (0, 0, 0)
    n = pts.shape[0]
    sdf = torch.empty(n, dtype=torch.float32, device=pts.device)
    vis = torch.empty(n, dtype=torch.uint8, device=pts.device)
    face = torch.empty(n, dtype=torch.int32, device=pts.device) if want_face else None
    knn = torch.empty(n, dtype=torch.int32, device=pts.device) if want_knn else None
    check(lib.vanerf_mesh_query_accel(byref(accel.c), _ptr(verts3, torch.float32), verts3.shape[0], _ptr(faces_i32, torch.int32),
                                      faces_i32.shape[0], _ptr(vert_vis, torch.float32), _ptr(pts, torch.float32), n, _ptr(sdf), _ptr(vis),
                                      _ptr(face), _ptr(knn), int(gnx), int(gny), int(gs), _ptr(_queue_word(pts.device)), _stream()))
    return tuple(t for t in (sdf, vis, face, knn) if t is not None)


def knn1(verts4, pts):
    idx = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
    check(lib.vanerf_knn1(_ptr(verts4, torch.float32), verts4.shape[0], _ptr(pts, torch.float32), pts.shape[0], _ptr(idx), _stream()))
    return idx


def query_order(frame, pts):
    """vanerf_query_order: the stable partition [samples that hit the source view and its mask | the others] as an int32 permutation.
    query_samples(order=...) then meets all-valid and all-invalid 32-sample groups only (same results, the invalid ones take its short path)."""
    n = pts.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=pts.device)
    nbytes = int(lib.vanerf_query_order_scratch(n))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    check(lib.vanerf_query_order(byref(frame.c), _ptr(pts, torch.float32), n, _ptr(order, torch.int32), _ptr(scratch, torch.uint8), nbytes, _stream()))
    return order


def query_samples(weights, frame, pts, query_sdf, query_vis, knn_idx, noise=None, want_valid=False, raw=False, order=None):
    """VANeRF.query + eval_func (src/model.py:748-957, 1140-1160): (N,3),(N,),(N,)u8,(N,)i32 -> (N,5) [alpha, sdf, r, g, b].
    order: optional permutation from query_order (work order only; results are unchanged)."""
    n = pts.shape[0]
    out = torch.empty(n, 5, dtype=torch.float32, device=pts.device)
    valid = torch.empty(n, dtype=torch.uint8, device=pts.device) if want_valid else None
    if order is not None and order.shape != (n,):
        raise ValueError("order must hold one index per sample")
    check(lib.vanerf_query_samples(weights.handle, byref(frame.c), _ptr(pts, torch.float32), _ptr(query_sdf, torch.float32),
                                   _ptr(query_vis, torch.uint8), _ptr(knn_idx, torch.int32), _ptr(noise, torch.float32), _ptr(order, torch.int32),
                                   int(bool(raw)), n, _ptr(out), _ptr(valid), _ptr(_queue_word(pts.device)), _stream()))
    return (out, valid) if want_valid else out


def composite(rgba, z, mesh_sdf, beta, want_contrib=True):
    """sdf_activation + rgba2out (src/model.py:879-882, 1464-1494).  rgba (R,S,5), z (R,S), mesh_sdf (R,S).
    beta: a number, or the PackedWeights whose device copy of sigmoid_beta the kernel reads (no host value involved)."""
    R, S = z.shape
    dev = z.device
    color = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth, alpha, sdf = (torch.empty(R, dtype=torch.float32, device=dev) for _ in range(3))
    contrib = torch.empty(R, S, dtype=torch.float32, device=dev) if want_contrib else None
    if isinstance(beta, PackedWeights):
        check(lib.vanerf_composite_handle(beta.handle, _ptr(rgba, torch.float32), _ptr(z, torch.float32), _ptr(mesh_sdf, torch.float32), S, None, None, 0,
                                          None, R, _ptr(color), _ptr(depth), _ptr(alpha), _ptr(sdf), _ptr(contrib), _stream()))
    else:
        check(lib.vanerf_composite(_ptr(rgba, torch.float32), _ptr(z, torch.float32), _ptr(mesh_sdf, torch.float32), R, S, float(beta),
                                   _ptr(color), _ptr(depth), _ptr(alpha), _ptr(sdf), _ptr(contrib), _stream()))
    return color, depth, alpha, contrib, sdf


def composite_backward(weights, rgba, z, mesh_sdf, g_color=None, g_depth=None, g_alpha=None, g_sdf=None, rgba_n=None, sdf_n=None, src=None):
    """Backward of composite() / composite_merged() (vanerf_composite_backward): gradients with respect to the composite's outputs in, gradient with
    respect to every table entry out -> (d_rgba (R,Sa,5), d_rgba_n (R,Sn,5) or None, d_beta (R,): the rays' shares of d / d clamp(sigmoid_beta)).
    sigmoid_beta is the weight handle's device copy; at most 256 samples per ray."""
    R, S = z.shape
    Sa = mesh_sdf.shape[1]
    Sn = 0 if rgba_n is None else sdf_n.shape[1]
    assert Sa + Sn == S and (src is None) == (rgba_n is None)
    dev = z.device
    f32 = torch.float32
    d_a = torch.empty(R, Sa, 5, dtype=f32, device=dev)
    d_n = torch.empty(R, Sn, 5, dtype=f32, device=dev) if Sn else None
    d_beta = torch.empty(R, dtype=f32, device=dev)
    c = lambda t: None if t is None else t.to(f32).contiguous()
    g_color, g_depth, g_alpha, g_sdf = c(g_color), c(g_depth), c(g_alpha), c(g_sdf)
    rgba, z, mesh_sdf, rgba_n, sdf_n = c(rgba), c(z), c(mesh_sdf), c(rgba_n), c(sdf_n)
    src = None if src is None else src.to(torch.int32).contiguous()
    check(lib.vanerf_composite_backward(weights.handle, _ptr(rgba), _ptr(z), _ptr(mesh_sdf), Sa, _ptr(rgba_n), _ptr(sdf_n), Sn, _ptr(src, torch.int32), R,
                                        _ptr(g_color), _ptr(g_depth), _ptr(g_alpha), _ptr(g_sdf), _ptr(d_a), _ptr(d_n), _ptr(d_beta), _stream()))
    return d_a, d_n, d_beta


def composite_merged(rgba_c, sdf_c, rgba_n, sdf_n, src, z_fine, beta, want_contrib=False):
    """Fine composite over [coarse samples | new importance samples] in merged depth order (vanerf_composite_merged)."""
    R, S = z_fine.shape
    Sc, Sn = sdf_c.shape[1], sdf_n.shape[1]
    assert Sc + Sn == S and src.shape == (R, S)
    dev = z_fine.device
    color = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth, alpha, sdf = (torch.empty(R, dtype=torch.float32, device=dev) for _ in range(3))
    contrib = torch.empty(R, S, dtype=torch.float32, device=dev) if want_contrib else None
    if isinstance(beta, PackedWeights):
        check(lib.vanerf_composite_handle(beta.handle, _ptr(rgba_c, torch.float32), _ptr(z_fine, torch.float32), _ptr(sdf_c, torch.float32), Sc,
                                          _ptr(rgba_n, torch.float32), _ptr(sdf_n, torch.float32), Sn, _ptr(src, torch.int32), R,
                                          _ptr(color), _ptr(depth), _ptr(alpha), _ptr(sdf), _ptr(contrib), _stream()))
    else:
        check(lib.vanerf_composite_merged(_ptr(rgba_c, torch.float32), _ptr(sdf_c, torch.float32), Sc, _ptr(rgba_n, torch.float32),
                                          _ptr(sdf_n, torch.float32), Sn, _ptr(src, torch.int32), _ptr(z_fine, torch.float32), R, float(beta),
                                          _ptr(color), _ptr(depth), _ptr(alpha), _ptr(sdf), _ptr(contrib), _stream()))
    return color, depth, alpha, contrib, sdf


_HOST = {}


def _host_key(t):
    return (t.data_ptr(), t._version, t.numel(), t.dtype)  # contiguous tensors only: a reshaped view of the same memory is the same numbers


def host_copy(t):
    """fp32 CPU copy of a small tensor (camera matrices, bounds: they are passed to the kernels by value).  Reading a device tensor back
    waits for everything queued before it, so copies of device tensors are remembered while the tensor is alive and unmodified; a caller
    that knows its next cameras hands them to prefetch_host_copies first (one read-back for all of them)."""
    if not t.is_cuda or not t.is_contiguous():
        return t.detach().to("cpu", torch.float32)
    key = _host_key(t)
    hit = _HOST.get(key)
    if hit is None:
        if len(_HOST) >= 1024:
            _HOST.clear()
        hit = _HOST[key] = (t.detach().reshape(-1).to("cpu", torch.float32), t)  # (the tensor itself is kept: its address stays its own)
    return hit[0].view(t.shape)


def prefetch_host_copies(tensors):
    """One device -> host transfer for many small tensors; host_copy() then finds them without touching the device."""
    todo = [t for t in tensors if t.is_cuda and t.is_contiguous() and _host_key(t) not in _HOST]
    if not todo:
        return
    if len(_HOST) + len(todo) >= 1024:
        _HOST.clear()
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in todo]).cpu()
    off = 0
    for t in todo:
        n = t.numel()
        _HOST[_host_key(t)] = (flat[off:off + n].clone(), t)
        off += n


_T_LIN = {}


def _t_lin(S, dev):
    """th.linspace(0, 1, S) on the device, built once per (S, device): a host -> device copy from pageable memory waits for everything
    queued on the stream, which stalled the host twice per pass (tools/perf_host_enqueue.py)."""
    key = (int(S), torch.device(dev))
    if key not in _T_LIN:
        _T_LIN[key] = torch.linspace(0.0, 1.0, steps=int(S)).to(dev)
    return _T_LIN[key]


def importance_merge(contrib, z, sample_per_ray, u=None, want_idx=False):
    """importance_sample + sort-merge (src/model.py:1424-1462, 1301-1307).  contrib, z: (R,Sc) -> z_new (R,Sf), z_fine (R,Sc+Sf), src."""
    R, Sc = z.shape
    Sf = int(sample_per_ray)
    dev = z.device
    t_lin = _t_lin(Sf, dev) if u is None else None
    z_new = torch.empty(R, Sf, dtype=torch.float32, device=dev)
    z_fine = torch.empty(R, Sc + Sf, dtype=torch.float32, device=dev)
    src = torch.empty(R, Sc + Sf, dtype=torch.int32, device=dev)
    idx = torch.empty(R, Sf, dtype=torch.int32, device=dev) if want_idx else None
    check(lib.vanerf_importance_merge(_ptr(contrib, torch.float32), _ptr(z, torch.float32), _ptr(u, torch.float32), _ptr(t_lin), R, Sc, Sf,
                                      _ptr(z_new), _ptr(z_fine), _ptr(src), _ptr(idx), _stream()))
    return (z_new, z_fine, src, idx) if want_idx else (z_new, z_fine, src)


def importance_from_midpoints(contrib_inner, z_mid, sample_per_ray, u=None, want_idx=False):
    """importance_sample in the reference's call shape (src/model.py:1304): contrib[..., 1:-1] (R,D-2), z_mid (R,D-1) -> (R,Sf)."""
    Rn, nb = contrib_inner.shape
    assert z_mid.shape == (Rn, nb + 1)
    dev = z_mid.device
    Sf = int(sample_per_ray)
    t_lin = _t_lin(Sf, dev) if u is None else None
    z_new = torch.empty(Rn, Sf, dtype=torch.float32, device=dev)
    idx = torch.empty(Rn, Sf, dtype=torch.int32, device=dev) if want_idx else None
    check(lib.vanerf_importance_sample(_ptr(contrib_inner, torch.float32), _ptr(z_mid, torch.float32), _ptr(u, torch.float32), _ptr(t_lin), Rn, nb, Sf,
                                       _ptr(z_new), _ptr(idx), _stream()))
    return (z_new, idx) if want_idx else z_new


def ray_bbox(bounds, orig, dirs):
    """ray_bbox_intersection (src/model.py:1496-1570): bounds (2,3), orig (3,), dirs (R,3) device -> near, far (R,), hit (R,) uint8."""
    Rn = dirs.shape[0]
    near, far = torch.empty(Rn, dtype=torch.float32, device=dirs.device), torch.empty(Rn, dtype=torch.float32, device=dirs.device)
    hit = torch.empty(Rn, dtype=torch.uint8, device=dirs.device)
    check(lib.vanerf_ray_bbox(_farr(bounds.detach().reshape(-1).tolist(), 6), _farr(orig.detach().reshape(-1).tolist(), 3), _ptr(dirs, torch.float32),
                              Rn, _ptr(near), _ptr(far), _ptr(hit), _stream()))
    return near, far, hit


def ray_setup(cam_tar, bounds, x0, y0, step, nx, ny, S, jitter=None, device=None, y_step=None, pixels=None, y_block=1, row_blocks=None):
    """Pixel grid + rays + bbox clip + coarse depths (src/model.py:1191-1238, 1496-1570).
    pixels: optional explicit (R,2) int32 device tensor of (x, y) (training patches); then nx*ny must equal R.
    y_step, y_block: rows are y0 + (iy // y_block) * y_step + (iy % y_block) * step (multi-GPU shards: blocks of y_block rows).
    row_blocks: optional (ny // y_block,) int32 device tensor, the first row of every block (shards dealt by cost: parallel.deal_blocks)."""
    dev = device or bounds.device
    K, RT = host_copy(cam_tar["K"]), host_copy(cam_tar["RT"])
    inv_K_T = torch.inverse(K[:, :3, :3]).transpose(1, 2)[0].contiguous()  # th.inverse(...).transpose(1, 2), model.py:1208
    R = nx * ny
    index = torch.empty(R, dtype=torch.int64, device=dev)
    rays_d = torch.empty(R, 3, dtype=torch.float32, device=dev)
    cam_pos = torch.empty(3, dtype=torch.float32, device=dev)
    near, far = torch.empty(R, dtype=torch.float32, device=dev), torch.empty(R, dtype=torch.float32, device=dev)
    hit = torch.empty(R, dtype=torch.uint8, device=dev)
    z = torch.empty(R, S, dtype=torch.float32, device=dev)
    t_lin = _t_lin(S, dev)
    cam_args = (_farr(inv_K_T.reshape(-1).tolist(), 9), _farr(RT[0, :3, :4].reshape(-1).tolist(), 12), float(cam_tar["znear"]),
                float(cam_tar["zfar"]), _farr(host_copy(bounds).reshape(-1).tolist(), 6), int(S), _ptr(t_lin), _ptr(jitter, torch.float32),
                _ptr(index), _ptr(rays_d), _ptr(cam_pos), _ptr(near), _ptr(far), _ptr(hit), _ptr(z), _stream())
    if pixels is not None:
        assert pixels.shape == (R, 2)
        check(lib.vanerf_ray_setup_pixels(_ptr(pixels, torch.int32), R, int(cam_tar["width"]), *cam_args))
    elif row_blocks is not None:
        assert row_blocks.dtype == torch.int32 and row_blocks.is_cuda and row_blocks.numel() * int(y_block) == ny
        check(lib.vanerf_ray_setup_blocks(_ptr(row_blocks, torch.int32), int(x0), int(step), int(y_block), int(nx), int(ny), int(cam_tar["width"]), *cam_args))
    else:
        check(lib.vanerf_ray_setup(int(x0), int(y0), int(step), int(y_step or step), int(y_block), int(nx), int(ny), int(cam_tar["width"]), *cam_args))
    return dict(index=index, rays_d=rays_d, cam_pos=cam_pos, near=near, far=far, hit=hit, z=z)


def sample_points(rays_d, cam_pos, z):
    R, S = z.shape
    pts = torch.empty(R * S, 3, dtype=torch.float32, device=z.device)
    check(lib.vanerf_sample_points(_ptr(rays_d, torch.float32), _ptr(cam_pos, torch.float32), _ptr(z, torch.float32), R, S, _ptr(pts), _stream()))
    return pts


# ------------------------------------------------------------------------------------------------
# one pass: rays -> coarse march -> importance -> fine march (src/model.py:1102-1360)
# ------------------------------------------------------------------------------------------------
# launches of at least this many samples are partitioned by validity first (query_order); below it the three small kernels cost more than they save
PARTITION_MIN_SAMPLES = 1 << 18


def render_pass(weights, frame, cam_tar, bounds, x0, y0, step, nx, ny, sample_per_ray_c=64, sample_per_ray_f=64, fine=True,
                jitter=None, u=None, noise_std=0.0, generator=None, debug=False, kernel_events=None, y_step=None, reuse_coarse=True,
                pixels=None, y_block=1, inject=None, noise_draws=None, row_blocks=None):
    """Returns flat per-ray tensors: color/depth/alpha (coarse), color_fine/depth_fine/alpha_fine/sdf (fine), index, z, z_fine.

    noise_draws: optional (coarse, fine) standard-normal draws (flat, one per evaluated sample) used instead of fresh ones when
    noise_std > 0 -- th.randn_like(rad) of src/model.py:1156 handed in (tests replay the reference's recorded draws).
    inject: optional dict of device tensors that REPLACE what the ray kernels would produce, for parity tests that march the oracle's own
    rays: "rays_d" (R,3), "cam_pos" (3,), "z" (R,Sc) and, for the fine march, "z_fine" (R,Sc+Sf) (then all Sc+Sf depths are evaluated: no
    importance kernel, no coarse re-use).  Last-bit differences between torch-CPU and the HIP ray generator otherwise reach the discrete
    decisions of the mesh query (closest face, inside test, visibility >= 0.1, 1-NN) on ~1e-5 of the samples.

    reuse_coarse: the fine composite needs the networks at the Sc coarse and the Sf new depths of every ray.  The reference
    evaluates all Sc+Sf again (src/model.py:1305-1345); the per-sample networks are pure functions of the position, so by
    default only the Sf new samples are evaluated and the coarse results are merged in by depth (same bits, 1/3 less work).
    With per-sample noise (training, rand_noise_std > 0) the reference draws fresh noise for the re-evaluated coarse samples -- but the
    noise is added to the networks' output (rad += randn * std, eval_func, src/model.py:1155-1156), not to their input, and with one source
    view nothing else is random in VANeRF.query (the view dropout of 804-810 needs n_views > 1): the networks are evaluated once per point
    here too (raw outputs + validity), and eval_func is applied twice to the coarse points, with the coarse pass's draws and with the draws
    the fine batch holds at their sorted positions.  Same bits as re-evaluating (tests/test_hip_parity.py), 1/3 less work."""
    Sc, Sf = int(sample_per_ray_c), int(sample_per_ray_f)
    rays = ray_setup(cam_tar, bounds, x0, y0, step, nx, ny, Sc, jitter=jitter, device=frame.verts3.device, y_step=y_step, pixels=pixels,
                     y_block=y_block, row_blocks=row_blocks)
    R = nx * ny
    if inject is not None:
        for k in ("rays_d", "cam_pos", "z"):
            if k in inject:
                if inject[k].shape != rays[k].shape or inject[k].dtype != torch.float32:
                    raise ValueError(f"inject[{k!r}] must be float32 of shape {tuple(rays[k].shape)}")
                rays[k] = inject[k].contiguous()
    if inject is not None and "z_fine" in inject:
        reuse_coarse = False
    noisy_reuse = noise_std > 0.0 and fine and reuse_coarse  # raw network outputs, eval_func (noise, mask) applied here
    if noise_draws is not None and (noise_draws[0].numel() != R * Sc or (fine and noise_draws[1].numel() != R * (Sc + Sf))):
        raise ValueError("noise_draws: one draw per evaluated sample (R*Sc coarse, R*(Sc+Sf) fine)")

    def scaled(draws, n):  # th.randn_like(rad) * rand_noise_std (src/model.py:1155-1156), drawn on the device (or handed in)
        dev = frame.verts3.device
        draws = torch.randn(n, device=dev, generator=generator) if draws is None else draws.reshape(-1).to(dev, torch.float32)
        return (draws * noise_std).contiguous()

    def eval_func(raw, valid, noise):
        """src/model.py:1140-1160 on the kernel's raw outputs [sdf_pred, rad, r, g, b] (vanerf_eval_func: the arithmetic of the kernel's own
        epilogue, csrc/query_kernel.hip, so the same bits; one launch instead of eight element-wise ones)."""
        n = raw.shape[0]
        out = torch.empty(n, 5, dtype=torch.float32, device=raw.device)
        check(lib.vanerf_eval_func(_ptr(raw, torch.float32), _ptr(valid, torch.uint8), None, None, None, _ptr(noise, torch.float32), 1, 0, n,
                                   float(frame.c.invalid_sdf), _ptr(out), None, _stream()))
        return out

    def evaluate(z, draws=None, raw=False):
        pts = sample_points(rays["rays_d"], rays["cam_pos"], z)
        grid = (nx, ny, z.shape[1]) if pixels is None else None
        q_sdf, q_vis, knn = mesh_query_accel(frame.accel, frame.verts3, frame.faces, frame.vert_vis, pts, grid=grid)
        noise = scaled(draws, pts.shape[0]) if noise_std > 0.0 and not raw else None
        order = query_order(frame, pts) if pts.shape[0] >= PARTITION_MIN_SAMPLES else None
        if kernel_events is not None:  # HIP events around the dominant kernel alone (the partition kernels are outside), on the stream it is launched on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        res = query_samples(weights, frame, pts, q_sdf, q_vis, knn, noise, order=order, raw=raw, want_valid=raw)
        if kernel_events is not None:
            e1.record()
            kernel_events.append((e0, e1, pts.shape[0]))
        d = dict(pts=pts, q_sdf=q_sdf.view(R, -1), q_vis=q_vis, knn=knn, noise=noise)
        if raw:
            d["raw"], d["valid"] = res
        else:
            d["rgba"] = res.view(R, -1, 5)
        return d

    if noisy_reuse:
        c = evaluate(rays["z"], raw=True)
        c["noise"] = scaled(None if noise_draws is None else noise_draws[0], R * Sc)
        c["rgba"] = eval_func(c["raw"], c["valid"], c["noise"]).view(R, Sc, 5)
    else:
        c = evaluate(rays["z"], None if noise_draws is None else noise_draws[0])
    c["color"], c["depth"], c["alpha"], c["contrib"], c["sdf"] = composite(c["rgba"], rays["z"], c["q_sdf"], weights)
    out = {"color": c["color"], "depth": c["depth"], "alpha": c["alpha"], "index": rays["index"], "z": rays["z"], "hit": rays["hit"],
           "rays_d": rays["rays_d"], "cam_pos": rays["cam_pos"]}
    if debug:
        out["coarse"] = c
    if fine:
        if inject is not None and "z_fine" in inject:
            z_new, z_fine, src = None, inject["z_fine"].contiguous(), None
            if z_fine.shape != (R, Sc + Sf):
                raise ValueError("inject['z_fine'] must have shape (R, Sc + Sf)")
        else:
            z_new, z_fine, src = importance_merge(c["contrib"], rays["z"], Sf, u=u)
        cf = None
        if noisy_reuse:
            # position of every coarse / new sample in the fine batch's sorted order: src[r][j] >= 0 is a coarse sample id, < 0 is ~(new sample id)
            ids = torch.where(src >= 0, src, Sc + (-src - 1)).long()
            pos = torch.empty_like(ids).scatter_(1, ids, torch.arange(Sc + Sf, device=ids.device).expand(R, -1))
            noise_f = scaled(None if noise_draws is None else noise_draws[1], R * (Sc + Sf)).view(R, Sc + Sf)
            f = evaluate(z_new, raw=True)
            f["noise"] = noise_f.gather(1, pos[:, Sc:]).reshape(-1).contiguous()
            f["rgba"] = eval_func(f["raw"], f["valid"], f["noise"]).view(R, Sf, 5)
            cf = {"noise": noise_f.gather(1, pos[:, :Sc]).reshape(-1).contiguous()}  # the coarse points as the fine batch sees them
            cf["rgba"] = eval_func(c["raw"], c["valid"], cf["noise"]).view(R, Sc, 5)
            f["color"], f["depth"], f["alpha"], f["contrib"], f["sdf"] = composite_merged(cf["rgba"], c["q_sdf"], f["rgba"], f["q_sdf"], src,
                                                                                          z_fine, weights, want_contrib=debug)
        elif reuse_coarse:
            f = evaluate(z_new)
            f["color"], f["depth"], f["alpha"], f["contrib"], f["sdf"] = composite_merged(c["rgba"], c["q_sdf"], f["rgba"], f["q_sdf"], src,
                                                                                          z_fine, weights, want_contrib=debug)
        else:
            f = evaluate(z_fine, None if noise_draws is None else noise_draws[1])
            f["color"], f["depth"], f["alpha"], f["contrib"], f["sdf"] = composite(f["rgba"], z_fine, f["q_sdf"], weights, want_contrib=debug)
        out.update({"color_fine": f["color"], "depth_fine": f["depth"], "alpha_fine": f["alpha"], "sdf": f["sdf"], "z_fine": z_fine, "z_new": z_new})
        if debug:
            out["fine"] = f
            out["fine_src"] = src if reuse_coarse else None
            out["coarse_in_fine"] = cf  # (noise, rgba) of the coarse points inside the fine batch when they carry other draws there; else None
    return out


def render_pass_c(weights, frame, cam_tar, bounds, x0, y0, step, nx, ny, sample_per_ray_c=64, sample_per_ray_f=64, fine=True, jitter=None, u=None,
                  noise_std=0.0, generator=None, y_step=None, reuse_coarse=True, pixels=None, y_block=1, noise_draws=None, row_blocks=None):
    """The same pass through the single C entry point vanerf_render_pass (include/vanerf_hip.h): one ctypes call enqueues every kernel of
    render_pass() above, in the same order with the same arguments -- the outputs are bit-identical (tests/test_hip_parity.py) -- with all
    temporaries in one scratch block.  This is what a non-Python host binds; the model's eval / no-grad passes go through it too."""
    Sc, Sf, R = int(sample_per_ray_c), int(sample_per_ray_f), nx * ny
    dev = frame.verts3.device
    K, RT = host_copy(cam_tar["K"]), host_copy(cam_tar["RT"])
    d = VanerfPassDesc()
    d.x0, d.y0, d.step_x, d.step_y, d.y_block, d.nx, d.ny = int(x0), int(y0), int(step), int(y_step or step), int(y_block), int(nx), int(ny)
    d.pixels_xy = _ptr(pixels, torch.int32)
    d.row_blocks = _ptr(row_blocks, torch.int32)
    d.width = int(cam_tar["width"])
    d.invK_T = _farr(torch.inverse(K[:, :3, :3]).transpose(1, 2)[0].reshape(-1).tolist(), 9)
    d.RT = _farr(RT[0, :3, :4].reshape(-1).tolist(), 12)
    d.znear, d.zfar = float(cam_tar["znear"]), float(cam_tar["zfar"])
    d.bounds = _farr(host_copy(bounds).reshape(-1).tolist(), 6)
    d.Sc, d.Sf, d.fine = Sc, Sf, int(bool(fine))
    noise = None
    if noise_std > 0.0:  # th.randn_like(rad) * rand_noise_std (src/model.py:1155-1156), drawn on the device (or handed in)
        draws = noise_draws if noise_draws is not None else (torch.randn(R * Sc, device=dev, generator=generator),
                                                             torch.randn(R * (Sc + Sf), device=dev, generator=generator) if fine else None)
        noise = tuple(None if t is None else (t.reshape(-1).to(dev, torch.float32) * noise_std).contiguous() for t in draws)
    d.reuse_coarse = int(bool(reuse_coarse))  # (under noise: raw outputs once per point, eval_func per set of draws, as render_pass does)
    t_c, t_f = _t_lin(Sc, dev), _t_lin(Sf, dev) if fine else None
    d.t_lin_c, d.t_lin_f = _ptr(t_c), _ptr(t_f)
    d.jitter, d.u = _ptr(jitter, torch.float32), _ptr(u, torch.float32)
    d.noise_c, d.noise_f = (_ptr(noise[0]), _ptr(noise[1])) if noise is not None else (None, None)
    f32 = torch.float32
    out = {"index": torch.empty(R, dtype=torch.int64, device=dev), "hit": torch.empty(R, dtype=torch.uint8, device=dev),
           "z": torch.empty(R, Sc, dtype=f32, device=dev), "color": torch.empty(R, 3, dtype=f32, device=dev),
           "depth": torch.empty(R, dtype=f32, device=dev), "alpha": torch.empty(R, dtype=f32, device=dev)}
    if fine:
        out.update({"color_fine": torch.empty(R, 3, dtype=f32, device=dev), "depth_fine": torch.empty(R, dtype=f32, device=dev),
                    "alpha_fine": torch.empty(R, dtype=f32, device=dev), "sdf": torch.empty(R, dtype=f32, device=dev),
                    "z_fine": torch.empty(R, Sc + Sf, dtype=f32, device=dev)})
    o = VanerfPassOut()
    for k in ("index", "hit", "z", "color", "depth", "alpha", "color_fine", "depth_fine", "alpha_fine", "sdf", "z_fine"):
        setattr(o, k, _ptr(out.get(k)))
    nbytes = int(lib.vanerf_render_pass_scratch(R, Sc, Sf, d.fine, 2 if d.reuse_coarse and noise is not None else d.reuse_coarse))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib.vanerf_render_pass(weights.handle, byref(frame.c), byref(frame.accel.c), _ptr(frame.verts3, f32), frame.verts3.shape[0],
                                 _ptr(frame.faces, torch.int32), frame.faces.shape[0], byref(d), byref(o), _ptr(scratch), nbytes, _stream()))
    return out


def _rows(g):
    """(n, C) with unit stride along the channels and rows any distance >= C apart (a column slice of a wider row-major tensor) is read in place."""
    return g if g.dim() == 2 and g.stride(1) == 1 and g.stride(0) >= g.shape[1] else g.contiguous()


def _rows_ptr(g):
    if not (g.is_cuda and g.dtype == torch.float32 and g.dim() == 2 and g.stride(1) == 1):
        raise ValueError("rows of fp32 channels on the device")
    return ctypes.c_void_p(g.data_ptr())


def scatter_add_taps(table, idx4, w4, sl, g):
    """table[idx4[k][i]] += w4[k][i] * g[i - sl.start] for the samples i of slice `sl`, k < 4 (vanerf_scatter_add_taps): idx4 (4, N) int32, w4 (4, N)."""
    n, C = g.shape
    assert idx4.shape == w4.shape and idx4.shape[0] == 4 and idx4.is_contiguous() and w4.is_contiguous() and table.is_contiguous() and sl.stop - sl.start == n
    if table.shape[0] * 4 > 128 * 1024:
        for k in range(4):
            table.index_add_(0, idx4[k, sl].long(), g * w4[k, sl].reshape(-1, 1))
        return table
    g = _rows(g)
    check(lib.vanerf_scatter_add_taps(ctypes.c_void_p(idx4.data_ptr() + 4 * sl.start), ctypes.c_void_p(w4.data_ptr() + 4 * sl.start), idx4.shape[1],
                                      _rows_ptr(g), g.stride(0), n, C, _ptr(table, torch.float32), table.shape[0], _stream()))
    return table


def scatter_add_rows2(table, idx, g, w, idx2, g2, w2):
    """table[idx[i]] += w[i] g[i] and table[idx2[i]] += w2[i] g2[i] in one launch (vanerf_scatter_add_rows2)."""
    n, C = g.shape
    assert table.shape[1] == C and idx.shape == (n,) and idx2.shape == (n,) and g2.shape == g.shape and table.is_contiguous()
    g, g2 = _rows(g), _rows(g2)
    if table.shape[0] * 4 > 128 * 1024 or g.stride(0) != g2.stride(0):
        scatter_add_rows(table, idx, g, w)
        return scatter_add_rows(table, idx2, g2, w2)
    check(lib.vanerf_scatter_add_rows2(_ptr(idx, torch.int32), _ptr(w, torch.float32), _rows_ptr(g), _ptr(idx2, torch.int32), _ptr(w2, torch.float32), _rows_ptr(g2),
                                       g.stride(0), n, C, _ptr(table, torch.float32), table.shape[0], _stream()))
    return table


def scatter_add_rows(table, idx, g, w=None):
    """table[idx[i]] += w[i] * g[i] (vanerf_scatter_add_rows): the backward of a row gather over ~1e6 samples into a table of ~1e3..1e4 rows."""
    n, C = g.shape
    assert table.shape[1] == C and idx.shape == (n,) and table.is_contiguous()
    if table.shape[0] * 4 > 128 * 1024:  # a table of more than 32 768 rows (a feature map beyond 181 x 181) does not fit the kernel's LDS slice
        table.index_add_(0, idx.long(), g if w is None else g * w.reshape(-1, 1))
        return table
    g = _rows(g)
    check(lib.vanerf_scatter_add_rows(_ptr(idx, torch.int32), _ptr(w, torch.float32), _rows_ptr(g), g.stride(0), n, C,
                                      _ptr(table, torch.float32), table.shape[0], _stream()))
    return table
