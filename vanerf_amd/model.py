"""Drop-in mirror of the reference's renderer interface (class VANeRF, reference src/model.py:604-1570) on top of the
HIP hot path.  Method names, argument order, dict layouts, returned keys/shapes, state_dict key names and assertion
behaviour follow the reference so that `VANeRFLightningModule` (src/model.py:42-602), train.py --run_val and
render_dynamic.py can call it unchanged; INTEGRATION.md shows the two-line change on the reference side.

What runs where
  * per-ray / per-sample work (a1-a19 of SURVEY.md section 8): libvanerf_hip.so through vanerf_amd/renderer.py;
  * per-frame work (image encoders, vertex un-projection, TexVisFusion's global vertex feature): PyTorch-ROCm;
  * n_views == 1 and batch == 1, as in both shipped configs (the non-spconv reference path is structurally single-view:
    src/networks.py:86,94; render_pifu_nerf hard-codes n_views = 1, src/model.py:1044).

Training: `forward()` in train mode under autograd returns the HIP values with the gradients of `vanerf_amd.torch_graph` -- the same
networks re-evaluated with torch ops at the samples of the HIP pass ("fused HIP forward + PyTorch autograd backward", SURVEY.md
section 8 row f-4, first stage; a fused HIP backward is not built).  Under no_grad / eval nothing of that runs.  `forward()` returns the
reference's `dict(loss, err_dict, out)`: the loss is `vanerf_amd.losses.compute_error` (src/utils.py:159-178) with `self.vgg_loss` as the
perceptual term -- `None` unless the caller attaches one (the reference constructs a pretrained torchvision VGG19 there).
A fresh module carries the reference's initial weights (`init_weights`, src/model.py:660-698; tests/golden/init_checksums.npz).
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as thf

from . import renderer as R
from .losses import compute_error

NUM_V = 779


# ------------------------------------------------------------------------------------------------------------------
# parameter containers with the reference's module tree (=> identical state_dict keys)
# ------------------------------------------------------------------------------------------------------------------
def _conv1(i, o):
    return nn.Conv1d(i, o, 1, padding=0, bias=False)


class GeoVisFusion(nn.Module):
    """Parameters of src/networks.py:43-73 (forward runs inside the HIP kernel)."""

    def __init__(self):
        super().__init__()
        self.fconv_at = nn.Sequential(_conv1(196, 10), nn.ReLU(True), _conv1(10, 3), nn.Sigmoid())
        self.fconv_ated = nn.Sequential(_conv1(196, 64), nn.ReLU(True), _conv1(64, 64))
        self.fconv_at1 = nn.Sequential(_conv1(28, 10), nn.ReLU(True), _conv1(10, 3), nn.Sigmoid())
        self.fconv_ated1 = nn.Sequential(_conv1(28, 8), nn.ReLU(True), _conv1(8, 8))


class TexVisFusion(nn.Module):
    """Parameters of src/networks.py:219-266; the per-frame stack (fconv3, fconv4, fconv_gt) runs in renderer.py."""

    def __init__(self, q_feat_in=96, q_feat_out=40):
        super().__init__()
        self.fconv = nn.Sequential(_conv1(q_feat_in, q_feat_in), nn.ReLU(True), _conv1(q_feat_in, q_feat_out))
        self.fconv_at = nn.Sequential(_conv1(q_feat_in, q_feat_in), nn.ReLU(True), _conv1(q_feat_in, 6), nn.Sigmoid())
        self.fconv_gt = nn.Sequential(nn.Conv1d(42, NUM_V, 3, padding=1, bias=False), nn.LayerNorm(18, 1e-6), nn.ReLU(True),
                                      nn.Conv1d(NUM_V, NUM_V * 2, 3, padding=1, bias=False), nn.LayerNorm(18, 1e-6), nn.ReLU(True))
        self.fconv3 = nn.Sequential(nn.Conv2d(8, 21, 3, padding=1, bias=False), nn.LayerNorm([64, 64], 1e-6), nn.ReLU(True),
                                    nn.Conv2d(21, 42, 3, padding=1, bias=False), nn.LayerNorm([64, 64], 1e-6), nn.ReLU(True),
                                    nn.AdaptiveAvgPool2d(3))
        self.fconv4 = nn.Sequential(nn.Conv2d(3, 21, 3, padding=1, bias=False), nn.LayerNorm([256, 256], 1e-6), nn.ReLU(True),
                                    nn.Conv2d(21, 42, 3, padding=1, bias=False), nn.LayerNorm([256, 256], 1e-6), nn.ReLU(True),
                                    nn.AdaptiveAvgPool2d(3))


class _Linear(nn.Module):
    """src/utils.py:670-685: `.linear` is a (weight-normed) nn.Linear -> keys linear.{weight_g,weight_v,bias} / linear.{weight,bias}."""

    def __init__(self, n_in, n_out, wn):
        super().__init__()
        self.linear = nn.utils.weight_norm(nn.Linear(n_in, n_out)) if wn else nn.Linear(n_in, n_out)


class _MLP(nn.Module):
    def __init__(self, dims_in, dims_out, norm):
        super().__init__()
        n = len(dims_in)
        self.layers = nn.ModuleList([_Linear(i, o, norm == "weight" and k != n - 1) for k, (i, o) in enumerate(zip(dims_in, dims_out))])


class MLPUNetFusion(nn.Module):
    """Parameters of src/utils.py:609-649 (layers1 = MLPUNet with skip concatenation, layers2 = MLP)."""

    def __init__(self, n_dims1, n_dims2, skip_dims, skip_layers, norm="weight", **kwargs):
        super().__init__()
        skip = dict(zip(skip_layers, skip_dims))
        ins1 = [n_dims1[i] + skip.get(i, 0) for i in range(len(n_dims1) - 1)]
        self.layers1 = _MLP(ins1, n_dims1[1:], norm)
        self.layers2 = _MLP(n_dims2[:-1], n_dims2[1:], norm)


class IBRRenderingHead(nn.Module):
    """src/model.py:1572-1636.  At n_views == 1 its output is exactly rgb_feat[..., :3] (softmax over a single view, src/model.py:1613,
    1635), so the HIP path never evaluates it; the parameters exist for checkpoint compatibility, forward() for completeness (V >= 1)."""

    def __init__(self, in_channels=32 + 5, **kwargs):
        super().__init__()
        c = in_channels + 3
        self.ani_al = nn.Parameter(torch.tensor(0.2))
        self.ray_encoder = nn.Sequential(nn.Linear(4, 16), nn.ELU(inplace=True), nn.Linear(16, c), nn.ELU(inplace=True))
        self.base_layer = nn.Sequential(nn.Linear(c * 3, 64), nn.ELU(inplace=True), nn.Linear(64, 32), nn.ELU(inplace=True))
        self.vis_layer1 = nn.Sequential(nn.Linear(32, 32), nn.ELU(inplace=True), nn.Linear(32, 33), nn.ELU(inplace=True))
        self.vis_layer2 = nn.Sequential(nn.Linear(32, 32), nn.ELU(inplace=True), nn.Linear(32, 1), nn.Sigmoid())
        self.out_layer = nn.Sequential(nn.Linear(37, 16), nn.ELU(inplace=True), nn.Linear(16, 8), nn.ELU(inplace=True), nn.Linear(8, 1))

    def forward(self, rgb_feats, ray_diffs, proj_mask):
        """src/model.py:1600-1636 for any number of views V: (R,S,V,40), (R,S,V,4), (R,S,V,1) -> (R,S,3).  Plain PyTorch (the per-sample kernel
        covers V = 1, where this reduces to rgb_feats[..., :3]); here so that the module is whole for callers that evaluate the head themselves."""
        views = rgb_feats.shape[2]
        colours = rgb_feats[..., :3]
        enc = self.ray_encoder(ray_diffs)
        feats = torch.cat([rgb_feats[..., :enc.shape[-1]] + enc, rgb_feats[..., enc.shape[-1]:]], -1)
        closeness = torch.exp(self.ani_al.abs() * (ray_diffs[..., 3:4] - 1.0))
        w = (closeness - closeness.amin(2, keepdim=True)) * proj_mask
        w = w / (w.sum(2, keepdim=True) + 1e-8)
        mean = (feats * w).sum(2, keepdim=True)                      # fused_mean_variance, src/utils.py:153-157
        var = (w * (feats - mean).square()).sum(2, keepdim=True)
        x = self.base_layer(torch.cat([mean.expand(-1, -1, views, -1), var.expand(-1, -1, views, -1), feats], -1))
        extra = self.vis_layer1(x * w)
        x = x + extra[..., :-1]
        vis = self.vis_layer2(x * torch.sigmoid(extra[..., -1:]) * proj_mask) * proj_mask
        score = self.out_layer(torch.cat([x, vis, ray_diffs], -1)).masked_fill(proj_mask == 0, -1e4)
        return (colours * torch.softmax(score, 2)).sum(2)


class SpatialEncoder(nn.Module):
    """Config holder of src/spatial.py (the 'rel_z_decay' encoding itself is fused into the HIP kernel)."""

    def __init__(self, sp_level, sp_type, scale, n_kpt, **kwargs):
        super().__init__()
        self.sp_type, self.sp_level, self.n_kpt, self.scale, self.kwargs = sp_type, sp_level, n_kpt, scale, kwargs
        self.register_buffer("center", torch.Tensor(kwargs.get("center", [0.0, 0.0, 0.0])).float())

    def get_dim(self):
        if self.sp_type != "rel_z_decay":
            raise NotImplementedError(f"sp_type {self.sp_type!r}: only 'rel_z_decay' (both shipped configs) is implemented")
        return (1 + 2 * self.sp_level) * self.n_kpt


# ------------------------------------------------------------------------------------------------------------------
class VANeRF(nn.Module):
    def __init__(self, cfg, geo_encoder=None, tex_encoder=None):
        super().__init__()
        model_cfg = cfg["models"]["VANeRF"]
        if model_cfg.get("sp_conv"):
            raise NotImplementedError("sp_conv=True (sparse-voxel fusion) is outside the hot path; both shipped configs set it to false")
        self.train_out_h = model_cfg.get("train_out_h", 64)
        self.train_out_w = model_cfg.get("train_out_w", 64)
        self.disable_fg_mask = model_cfg.get("disable_fg_mask", False)
        if self.disable_fg_mask:
            raise NotImplementedError("disable_fg_mask is not used by the shipped configs")
        self.nkpt_r, self.nkpt_l = 21, 21
        self.sigmoid_beta = nn.Parameter(0.1 * torch.ones(1))
        self.geo_vis_fusion = GeoVisFusion()
        self.tex_vis_fusion = TexVisFusion()
        sp_encoder = SpatialEncoder(**model_cfg["sp_args"])
        mlp_geo_args = copy.deepcopy(model_cfg["mlp_geo_args"])
        mlp_geo_args["n_dims1"][0] = sp_encoder.get_dim()
        if (mlp_geo_args["n_dims1"], mlp_geo_args["n_dims2"], mlp_geo_args["skip_dims"], mlp_geo_args["skip_layers"]) != \
                ([294, 128, 128, 120, 64], [128, 64, 64, 2], [64, 8], [0, 2]) or mlp_geo_args.get("pool_types") != ["mean", "var"]:
            raise NotImplementedError("the HIP kernel is specialised on the shipped mlp_geo_args (configs/vanerf.json:62-90)")
        self.mlp_geo = MLPUNetFusion(**mlp_geo_args)
        self.mlp_tex = IBRRenderingHead(**model_cfg["mlp_tex_args"]["args"])
        self.ibr_compress_gfeat = nn.Linear(model_cfg["mlp_tex_args"]["gcompress"]["in_ch"], model_cfg["mlp_tex_args"]["gcompress"]["out_ch"])
        if geo_encoder is None or tex_encoder is None:
            from .encoders import HGFilterV2, ResBlkEncoder
            geo_encoder = geo_encoder or HGFilterV2(**model_cfg["geo_args"])
            tex_encoder = tex_encoder or ResBlkEncoder(**model_cfg["tex_args"])
        self.geo_encoder, self.tex_encoder = geo_encoder, tex_encoder
        self.sp_encoder = sp_encoder
        self.sp_encoder_r = SpatialEncoder(**model_cfg["sp_args"])
        self.sp_encoder_l = SpatialEncoder(**model_cfg["sp_args"])
        self.sp_encoder_postfusion = None
        self.ds_geo = model_cfg.get("ds_geo", 0)
        self.ds_tex = model_cfg.get("ds_tex", 0)
        self.v_level = model_cfg.get("v_level", 0)
        self.dr_level = model_cfg.get("dr_level", 5)
        self.feat_geo = None
        self.feat_tex = None
        self.kwargs = model_cfg
        self.disable_bg = True
        # same three calls, same order as the reference's constructor (src/model.py:660-662)
        self.init_weights(self)
        self.init_weights(self.mlp_geo, "kaiming", nl="relu")
        self.init_weights(self.mlp_tex, "kaiming", nl="leaky_relu")
        self.vgg_loss = None  # perceptual term of forward()'s loss: a callable (pred, target) -> scalar, attached by the caller
        # "bf16x3" (default: bf16 MFMA on hi + lo split operands, within 3.3e-5 of the fp32 kernel on every sample, 1e-4 of the oracle with
        # identical rays, bit-reproducible; the configuration every published number is measured in) | "fp32" (fp32 MFMA, 2.2x slower);
        # renderer.PRECISIONS; not a reference key
        self.precision = model_cfg.get("mfma_precision", "bf16x3")
        self._packed = None  # (version key, PackedWeights)
        self._frame_cache = None
        self._encoders_graphed = False
        self._enc_cache = None  # (key, feat_geo, feat_tex, image)

    @staticmethod
    def init_weights(net, init_type="normal", gain=0.02, nl="relu"):
        """src/model.py:660-698.  Every module re-seeds the global generator with 125 before its draw, so a layer's initial weight depends
        only on its shape; weight-normed Linears keep PyTorch's default init (`m.weight` is their derived tensor, not a parameter), which is
        why the constructor creates the fusion blocks and mlp_geo in the reference's order.  Not reproduced: the reference also sets
        torch.backends.cudnn.deterministic = True here (a process-wide solver restriction, not a property of the weights)."""
        def init_func(m):
            torch.manual_seed(125)
            classname = m.__class__.__name__
            if hasattr(m, "weight") and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
                if init_type == "normal":
                    nn.init.normal_(m.weight.data, 0.0, gain)
                elif init_type == "xavier":
                    nn.init.xavier_normal_(m.weight.data, gain=gain)
                elif init_type == "kaiming":
                    nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in", nonlinearity=nl)
                elif init_type == "orthogonal":
                    nn.init.orthogonal_(m.weight.data, gain=gain)
                else:
                    raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
                if hasattr(m, "bias") and m.bias is not None:
                    nn.init.constant_(m.bias.data, 0.0)
            elif classname.find("BatchNorm2d") != -1:
                nn.init.normal_(m.weight.data, 1.0, gain)
                nn.init.constant_(m.bias.data, 0.0)

        net.apply(init_func)

    # ---- image features (src/model.py:700-746) ---------------------------------------------------------------------
    def attach_im_feat(self, im, return_val=False):
        if return_val:
            out = {"feat_geo": self.attach_geo_feat(im, return_val)}
            feat_tex = self.attach_tex_feat(im, return_val)
            if feat_tex is not None:
                out["feat_tex"] = feat_tex
            return out
        self.attach_geo_feat(im, return_val)
        self.attach_tex_feat(im, return_val)

    def _graph_encoders(self, im):
        """Optional (config key `graph_encoders`): the two image encoders as HIP graphs, forward and backward (torch.cuda.make_graphed_callables).
        Their ~800 small launches per training step cost the host more than the GPU (the step is host-bound while they are enqueued); as
        graphs they cost two launches each.  Needs what graph capture needs: source images of one fixed shape, parameters that stay where
        they are (in-place optimizer updates).  Same values and gradients (tests/test_autograd.py); eval-mode calls keep the eager path."""
        def sample(ds):
            x = im.view(-1, *im.shape[2:]) if len(im.shape) == 5 else im
            for _ in range(ds):
                x = thf.avg_pool2d(x, 2, stride=2)
            return (2.0 * x - 1.0).detach().clone()

        mods, args = [self.geo_encoder], [(sample(self.ds_geo),)]
        if self.tex_encoder is not None:
            mods.append(self.tex_encoder)
            args.append((sample(self.ds_tex),))
        graphed = torch.cuda.make_graphed_callables(tuple(mods), tuple(args), allow_unused_input=True)
        self.geo_encoder = graphed[0]
        if self.tex_encoder is not None:
            self.tex_encoder = graphed[1]
        self._encoders_graphed = True

    def attach_geo_feat(self, im, return_val=False):
        if not return_val:
            self.im = im.clone()
        if len(im.shape) == 5:
            im = im.view(-1, *im.shape[2:])
        for _ in range(self.ds_geo):
            im = thf.avg_pool2d(im, 2, stride=2)
        self.feat_geo = self.geo_encoder(2.0 * im - 1.0)
        if return_val:
            return self.feat_geo

    def attach_tex_feat(self, im, return_val=False):
        if self.tex_encoder is None:
            return None
        if len(im.shape) == 5:
            im = im.view(-1, *im.shape[2:])
        for _ in range(self.ds_tex):
            im = thf.avg_pool2d(im, 2, stride=2)
        self.feat_tex = self.tex_encoder(2.0 * im - 1.0)
        if return_val:
            return self.feat_tex

    def encoded(self, im):
        """Both encoders' feature maps of a source image.  Outside autograd they are kept for as long as the caller passes the same (unmodified)
        image tensor and the encoders' parameters and buffers do not change (eval mode only): render_pifu_nerf is called once per target view (120 times per orbit) and the
        reference runs both encoders in every call (src/model.py:1049-1050)."""
        if torch.is_grad_enabled() or self.training:  # (in train mode a forward also updates the BatchNorm statistics)
            return self.attach_geo_feat(im, return_val=True), self.attach_tex_feat(im, return_val=True)
        state = [t for enc in (self.geo_encoder, self.tex_encoder) if enc is not None for t in (*enc.parameters(), *enc.buffers())]
        key = (im.data_ptr(), im._version, tuple(im.shape), self.ds_geo, self.ds_tex, tuple((t.data_ptr(), t._version) for t in state))
        if self._enc_cache is None or self._enc_cache[0] != key:
            self._enc_cache = (key, self.attach_geo_feat(im, return_val=True), self.attach_tex_feat(im, return_val=True), im)
        return self._enc_cache[1], self._enc_cache[2]

    def detach_im_feat(self):
        self.feat_geo = None
        self.feat_tex = None

    # ---- HIP-side state ----------------------------------------------------------------------------------------------
    def _hot_state(self):
        """name -> tensor of everything outside the two encoders, as state_dict() names it.  state_dict() itself walks ~3 400 entries per call
        (1 ms, twice per training step), so the (module, name, tensor) triples are kept and only checked for replaced parameters."""
        cache = getattr(self, "_hot_cache", None)
        if cache is None or any(store.get(n) is not t for store, n, t, _ in cache):
            cache = []
            for mname, m in self.named_modules():
                if mname.split(".")[0] in ("geo_encoder", "tex_encoder"):
                    continue
                prefix = mname + "." if mname else ""
                cache += [(m._parameters, n, t, prefix + n) for n, t in m._parameters.items() if t is not None]
                cache += [(m._buffers, n, t, prefix + n) for n, t in m._buffers.items() if t is not None and n not in m._non_persistent_buffers_set]
            self._hot_cache = cache
        return {k: t.detach() for _, _, t, k in cache}

    def _packed_for(self, slot, precision, sd, key):
        """The handle of `precision`, packed on first use and re-packed when a parameter changed: in place on the device when the parameters
        live there (training: vanerf_weights_update, nothing blocks), through the host otherwise."""
        held = getattr(self, slot, None)
        if held is not None and held[0] == key:
            return held[1]
        hot = {k: v for k, v in sd.items() if k.startswith(self._PACKED_PREFIXES)}
        if held is not None and all(v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() for v in hot.values()) \
                and held[1].device_index == next(iter(hot.values())).device.index == torch.cuda.current_device():
            w = held[1].update(sd)
        else:
            w = R.PackedWeights(sd, mode=precision)
        setattr(self, slot, (key, w))
        return w

    _PACKED_PREFIXES = ("geo_vis_fusion.", "mlp_geo.", "ibr_compress_gfeat.", "tex_vis_fusion.fconv.", "tex_vis_fusion.fconv_at.", "sigmoid_beta")

    def packed_weights(self, precision=None):
        """MFMA-fragment copy of the per-sample weights, re-packed whenever a parameter changed (training steps, load_state_dict).
        precision: None = the module's own; "fp32" = the handle the fused backward runs on (it carries the transposed streams)."""
        sd = self._hot_state()
        precision = precision or self.precision
        key = (precision,) + tuple((k, v._version, v.data_ptr()) for k, v in sd.items() if k.startswith(self._PACKED_PREFIXES))
        return self._packed_for("_packed" if precision == self.precision else "_packed_alt", precision, sd, key)

    def fold_transf(self, cam):
        """cam['transf'] (B,2,3): the optional 2-D affine the reference applies to every projected point, xy' = A (x/z, y/z) + t
        (src/model.py:783-785, 848-850, 979-981, 1249-1251, 1261-1263).  It composes into the projection itself -- xy' = (A (x, y) + t z) / z -- so the
        kernels, which take KRT by value, never see it: rows 0 and 1 of KRT become A KRT[:2] + t KRT[2].  (Rounding differs from the reference's
        two-step form in the last bits; checked against the oracle, which applies the affine as written.)  No shipped config sets the key.
        The folded dict is kept while the same KRT / transf tensors come in, so the per-frame cache keyed on KRT's identity still hits."""
        if "transf" not in cam:
            return cam
        KRT, t = cam["KRT"], cam["transf"]
        key = (KRT.data_ptr(), KRT._version, tuple(KRT.shape), t.data_ptr(), t._version, tuple(t.shape))
        hit = getattr(self, "_transf_cache", None)
        if hit is None or hit[0] != key:
            M = torch.eye(3, dtype=KRT.dtype, device=KRT.device).repeat(KRT.shape[0], 1, 1)
            M[:, :2, :2] = t[:, :2, :2].to(KRT)
            M[:, :2, 2] = t[:, :, 2].to(KRT)
            new = KRT.clone()
            new[:, :3, :] = M @ KRT[:, :3, :]
            folded = {k: v for k, v in cam.items() if k != "transf"}
            folded["KRT"] = new
            self._transf_cache = hit = (key, folded, (KRT, t))  # the inputs are kept alive with the entry (their addresses are the key)
        out = dict(hit[1])
        out.update({k: v for k, v in cam.items() if k not in ("transf", "KRT")})
        return out

    def frame_data(self, img_in, cam_in, targets, feat_geo, feat_tex, sp_data, fg_mask):
        """Per-source-frame device data (vertex features, visibility, acceleration structure); cached on the identity of its inputs
        because render_pifu_nerf / render_novel_views call batch_render_pifu_nerf many times per source frame."""
        # identity AND version of every input; the tensors themselves are kept with the entry, so none of their addresses can be handed to a
        # different tensor while the entry is alive (an address alone would match a new batch that the allocator placed where the old one was)
        deps = (img_in, feat_geo[0], feat_geo[1], feat_tex, targets["vert_world"], targets["face_world"], cam_in["KRT"], sp_data["kpt3d"],
                sp_data["extrin"], fg_mask)
        scalars = tuple(float(cam_in[k]) for k in ("width", "height", "znear", "zfar", "nml_scale")) + tuple(sorted(self.kwargs["sp_args"].items()))
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in deps) + (scalars, tuple(p._version for p in self.tex_vis_fusion.parameters()))
        # Under autograd nothing is kept: a training step changes the encoders' weights, and with graph_encoders the feature maps are the
        # captured graph's static output buffers -- same address, same version counter after every replay -- so the key above cannot see it.
        fresh = torch.is_grad_enabled()  # (forward() hands this method DETACHED maps, so requires_grad says nothing here)
        if fresh or self._frame_cache is None or self._frame_cache[0] != key:
            sd = {"tex_vis_fusion." + k: v for k, v in self.tex_vis_fusion.state_dict().items()}
            fd = R.FrameData(sd, img_in, feat_geo, feat_tex, fg_mask, cam_in, targets, sp_data, self.kwargs["sp_args"])
            if fresh:
                self._frame_cache = None
                return fd
            self._frame_cache = (key, fd, deps)
        return self._frame_cache[1]

    # ---- per-sample query (src/model.py:748-877) ---------------------------------------------------------------------
    def query(self, pts, cam, hand_type, targets, feat_geo=None, feat_tex=None, vert=None, vert_vis=None, query_vis=None, query_sdf=None,
              closest_face=None, n_views=1, sp_data={}, tx_data={}, view=None, n_pts_samples=-1, **kwargs):
        """(B=1, N, 3) -> out (1, N, 5) = [sdf_pred, rad, r, g, b], valid (1, N, 1) bool.  `vert_vis` / `closest_face` / `view` are accepted for
        signature compatibility: visibility is recomputed per frame by the same rasteriser and `view` only feeds the value-dead IBR head."""
        assert n_views == 1 and pts.shape[0] == 1, "the non-spconv path is single-view (src/networks.py:86,94)"
        feat_geo = self.feat_geo if feat_geo is None else feat_geo
        feat_tex = self.feat_tex if feat_tex is None else feat_tex
        fd = self.frame_data(tx_data["img"], self.fold_transf(cam), targets, feat_geo, feat_tex, sp_data, kwargs["src_foreground_mask"])
        p = pts[0].contiguous().float()
        if query_sdf is None or query_vis is None:
            q_sdf, q_vis, knn = R.mesh_query_accel(fd.accel, fd.verts3, fd.faces, fd.vert_vis, p)
        else:
            q_sdf, q_vis = query_sdf.reshape(-1).float().contiguous(), query_vis.reshape(-1).to(torch.uint8).contiguous()
            knn = R.knn1(fd.verts4, p)
        out, valid = R.query_samples(self.packed_weights(), fd, p, q_sdf, q_vis, knn, want_valid=True, raw=True)
        sp_data.update({"n_view": n_views, "pts": pts, "v": pts})  # the reference mutates sp_data in place (model.py:833-834)
        sp_data.update(cam)
        return out[None], valid.bool()[None, :, None]

    def sdf_activation(self, input):
        self.sigmoid_beta.data.clamp_(min=2e-3)  # in-place clamp of the parameter (src/model.py:880)
        return torch.sigmoid(input / self.sigmoid_beta) / self.sigmoid_beta

    @staticmethod
    def rgba2out(self, rgba, z, vert_sdf):
        """src/model.py:1464-1494 -> color (B,R,3), depth, alpha (B,R), contrib (B,R,S), sdf (B,R)."""
        self.sigmoid_beta.data.clamp_(min=2e-3)
        B, Rn, S = z.shape
        c, d, a, con, s = R.composite(rgba.reshape(B * Rn, S, 5).float().contiguous(), z.reshape(B * Rn, S).float().contiguous(),
                                      vert_sdf.reshape(B * Rn, S).float().contiguous(), float(self.sigmoid_beta.detach()))
        return c.view(B, Rn, 3), d.view(B, Rn), a.view(B, Rn), con.view(B, Rn, S), s.view(B, Rn)

    @staticmethod
    def importance_sample(contrib, z, sample_per_ray, uniform=False):
        """src/model.py:1424-1462: contrib (B,N,D-2), z (B,N,D-1) mid-points -> (B,N,sample_per_ray)."""
        assert contrib.shape[-1] == z.shape[-1] - 1
        B, N, D2 = contrib.shape
        u = None if uniform else torch.rand(B * N, sample_per_ray, device=contrib.device)  # th.rand(...), src/model.py:1443
        out = R.importance_from_midpoints(contrib.reshape(B * N, D2).float().contiguous(), z.reshape(B * N, D2 + 1).float().contiguous(),
                                          sample_per_ray, u)
        return out.view(B, N, sample_per_ray)

    @staticmethod
    def ray_bbox_intersection(bounds, orig, direct, boffset=(-0.01, 0.01)):
        """src/model.py:1496-1570: bounds (B,2,3), orig (B,1,3), direct (B,N,3) -> near, far (B,N,1), hit (B,N,1) bool."""
        assert tuple(boffset) == (-0.01, 0.01), "the kernel hard-codes the reference's default offsets"
        outs = [R.ray_bbox(bounds[b], orig[b, 0], direct[b].contiguous().float()) for b in range(bounds.shape[0])]
        near, far, hit = (torch.stack([o[i] for o in outs], 0)[..., None] for i in range(3))
        return near, far, hit.bool()

    # ---- whole passes ------------------------------------------------------------------------------------------------
    @staticmethod
    def batch_render_pifu_nerf(net, img_in, cam_in, hand_type, targets, n_views, cam_tar, level=2, stride=0, tar_img=None, feat_geo=None,
                               feat_tex=None, mano_vert_world=None, sp_data={}, objcenter=None, **config):
        """src/model.py:1102-1422.  Same config keys (sample_per_ray_c/f, fine, uniform, rand_noise_std, src_foreground_mask, bounds, msk)."""
        batch_size = cam_tar["K"].shape[0]
        assert batch_size == 1 and n_views == 1, "val_batch_size = 1 and one source view (configs/vanerf.json:24; src/model.py:1044)"
        cam_in = net.fold_transf(cam_in)  # (cam_tar['transf'] is not read by the reference's march either: rays come from K and RT, src/model.py:1203-1213)
        Sc = config.get("sample_per_ray_c", 64)
        Sf = config.get("sample_per_ray_f", 64)
        fine = config.get("fine", False)
        uniform = config.get("uniform", False)
        if config.get("separate_cf", False):
            raise NotImplementedError("separate_cf is not used by the shipped configs")
        noise_std = config.get("rand_noise_std", 0.0)  # applied whenever the caller passes it, train or eval (src/model.py:1128, 1155)
        if feat_geo is None:
            feat_geo = net.attach_geo_feat(img_in, return_val=True)
        if feat_tex is None:
            feat_tex = net.attach_tex_feat(img_in, return_val=True)
        width = cam_tar.get("width", cam_in["width"])
        height = cam_tar.get("height", cam_in["height"])
        st = 2 ** (level - 1)
        assert width % st == 0 and height % st == 0
        if isinstance(stride, int):
            assert stride < st
            off = (stride, stride)
        elif isinstance(stride, torch.Tensor):
            assert stride.max().item() < st
            off = tuple(int(v) for v in stride.reshape(-1, 2)[0].tolist())
        else:
            raise NotImplementedError("unsupported stride type")
        dev = img_in.device
        cam_t = dict(cam_tar, width=width, height=height, znear=cam_tar.get("znear", cam_in["znear"]), zfar=cam_tar.get("zfar", cam_in["zfar"]))
        fd = net.frame_data(img_in, cam_in, targets, feat_geo, feat_tex, sp_data, config["src_foreground_mask"])
        pixels = None
        draws = config.get("_draws")  # tests: the random numbers of the pass handed in (pick, jitter, u, noise_c, noise_f) instead of drawn here
        if net.training and "msk" in config:  # 64x64 window around a random mask pixel, clamped (src/model.py:1172-1189)
            out_h, out_w = net.train_out_h, net.train_out_w
            msk = config["msk"][0].squeeze()
            coords = torch.stack(torch.where(msk)[::-1], -1)
            pick = np.random.randint(0, coords.shape[0], 1) if coords.shape[0] > 0 and draws is None else np.asarray(draws["pick"]).reshape(1) if draws else None
            centre = coords[pick] if coords.shape[0] > 0 else torch.zeros((1, 2), device=msk.device)
            yg, xg = torch.meshgrid(torch.arange(0, out_h, device=dev), torch.arange(0, out_w, device=dev), indexing="ij")
            grids = torch.stack([xg, yg], -1).view(-1, 2) + (centre.to(dev) - out_h // 2)
            pixels = grids.clamp(0, min(width - 1, height - 1)).to(torch.int32).contiguous()
            nx, ny, x0, y0, step = out_w * out_h, 1, 0, 0, 1
        else:
            out_w, out_h = width // st, height // st
            nx, ny, x0, y0, step = out_w, out_h, off[0], off[1], st
        Rn = out_w * out_h
        if draws is not None and not uniform:
            jitter, u = (draws[k].reshape(Rn, -1).to(dev, torch.float32).contiguous() for k in ("jitter", "u"))
        else:
            jitter = None if uniform else torch.rand(Rn, Sc, device=dev)          # th.rand_like(z), src/model.py:1229
            u = None if uniform else torch.rand(Rn, Sf, device=dev)                # th.rand(...), src/model.py:1443
        noise_draws = (draws["noise_c"], draws["noise_f"]) if draws is not None and noise_std > 0.0 else None
        want_graph = bool(config.get("_autograd", False))  # forward() under autograd: keep what vanerf_amd.torch_graph needs
        # one C call (vanerf_render_pass) unless the autograd graph needs the intermediates of the Python sequence
        run = R.render_pass if want_graph else R.render_pass_c
        o = run(net.packed_weights(), fd, cam_t, config["bounds"], x0, y0, step, nx, ny, Sc, Sf, fine=fine, jitter=jitter, u=u,
                noise_std=float(noise_std), pixels=pixels, noise_draws=noise_draws, **({"debug": True} if want_graph else {}))
        if want_graph:
            net._last_pass = (o, fd, cam_in)
        out = {"tex_fg": o["color"].view(1, out_h, out_w, 3).permute(0, 3, 1, 2), "depth": o["depth"].view(1, out_h, out_w),
               "alpha": o["alpha"].view(1, out_h, out_w)}
        if fine:
            out.update({"tex_fg_fine": o["color_fine"].view(1, out_h, out_w, 3).permute(0, 3, 1, 2), "depth_fine": o["depth_fine"].view(1, out_h, out_w),
                        "alpha_fine": o["alpha_fine"].view(1, out_h, out_w), "sdf": o["sdf"].view(1, out_h, out_w)})
        index = o["index"][None]
        # largest pixel index of the pass, on the host when the rays are a grid (reading it back would stall the host behind the whole pass)
        index_max = int(index.max()) if pixels is not None else (y0 + (ny - 1) * step) * int(width) + x0 + (nx - 1) * step

        def gather(t, ch):  # GT gathers at `index` (src/model.py:1361-1418); the reference indexes source-sized tensors with target
            flat = t.reshape(t.shape[0], ch, -1)  # pixel indices and fails for targets larger than the source -- guarded here
            if index_max >= flat.shape[-1]:
                return None
            return torch.gather(flat, 2, index[:, None].expand(-1, ch, -1)).view(t.shape[0], ch, out_h, out_w)

        if tar_img is not None:
            g = gather(tar_img, 3)
            if g is not None:
                out["tar_img"] = g
            if "msk" in config:
                g = gather(config["msk"].reshape(1, 1, -1), 1)
                if g is not None:
                    out["tar_alpha"] = g.float()
        # render_vis (pytorch3d soft rasteriser, discriminator supervision only) is out of scope: zeros of the reference's shape
        out["vis_img_all"] = torch.zeros(1, 1, 256, 256, device=dev)
        for key, t, ch in (("vis_img", out["vis_img_all"], 1), ("input_mask", config["src_foreground_mask"].reshape(1, 1, *config["src_foreground_mask"].shape[-2:]), 1),
                           ("img_in", img_in, 3)):
            g = gather(t, ch)
            if g is not None:
                out[key] = g
        for key in ("input_densepose", "tar_densepose"):
            if key in targets:
                g = gather(targets[key], 3)
                if g is not None:
                    out[key] = g
        out["vert_vis"] = fd.vert_vis[None, :, None]
        return out

    @staticmethod
    def render_pifu_nerf(self, net, img_in, cam_in, hand_type, targets, cam_tar, level=5, sp_data={}, bkg_emb=None, camcenter=None, objcenter=None,
                         tar_img=None, **config):
        """src/model.py:1026-1100.  The reference renders stride^2 pixel-interleaved passes and pixel_shuffles them because its unfused
        intermediates do not fit; rays are independent, so ONE full-resolution launch (level 1) yields the same image."""
        feat_geo, feat_tex = net.encoded(img_in)
        out = net.batch_render_pifu_nerf(net, img_in, cam_in, hand_type, targets, 1, cam_tar, 1, 0, tar_img, feat_geo, feat_tex, None, sp_data,
                                         objcenter, **config)
        if "input_densepose" in out and self is not None and hasattr(self, "discriminator"):
            rendered = out["tex_fg_fine"].clamp(min=0.0, max=1.0)
            _, out["fake_vis_pred"] = self.discriminator(out["img_in"], out["input_densepose"], out["tar_densepose"], rendered.detach())
            _, out["real_vis_pred"] = self.discriminator(out["img_in"], out["input_densepose"], out["tar_densepose"], out["tar_img"])
        ret = {}
        for k, v in out.items():
            if v is None or len(v.shape) < 3:
                continue
            ret[k] = v[0] if len(v.shape) == 4 else v  # (C,H,W); 3-D tensors (B,H,W) already read as (1,H,W)
        vert3d = targets["vert_world"]
        vimg = vert3d @ cam_tar["KRT"][:, :3, :3].transpose(1, 2) + cam_tar["KRT"][:, :3, 3][:, None]
        ret["vert_xy"] = vimg[..., :2] / (vimg[..., 2:3] + 1e-8)
        if "transf" in cam_tar:  # src/model.py:1093-1095
            ret["vert_xy"] = ret["vert_xy"] @ cam_tar["transf"][:, :2, :2].transpose(1, 2) + cam_tar["transf"][:, :, 2][:, None]
        ret["vert_vis"] = out["vert_vis"]
        return ret

    def attach_autograd(self, out, img_in, feat_geo, feat_tex, targets, sp_data, fg_mask):
        """Replaces the differentiable entries of `out` (a batch_render_pifu_nerf result computed with _autograd=True) by tensors that
        carry the HIP values and, in backward, the gradients of vanerf_amd.torch_graph evaluated at the same samples (same points, same
        importance samples, same mesh queries, same noise draws) -- torch_graph.PassGradient: the graph is built and differentiated chunk of
        rays by chunk of rays inside backward, so a training step never holds the activations of the whole patch."""
        from . import torch_graph as G
        o, fd, cam_in = self._last_pass
        self._last_pass = (o, fd, cam_in) if getattr(self, "_keep_last_pass", False) else None  # tests inspect the pass
        self._hot_state()  # (refreshes the kept list; named_parameters() walks the encoders' ~1 700 entries too: 1 ms per step)
        named = [(k, t) for _, _, t, k in self._hot_cache if isinstance(t, nn.Parameter) and not k.startswith("mlp_tex.")]
        names = [k for k, _ in named] + ["@feat_geo0", "@feat_geo1", "@feat_tex"]
        leaves = [p for _, p in named] + [feat_geo[0], feat_geo[1], feat_tex]
        frame = {"cam": cam_in, "img": img_in, "fg_mask": fg_mask.reshape(1, 1, *fg_mask.shape[-2:]), "verts": targets["vert_world"][0],
                 "vert_vis": fd.vert_vis, "kpt3d": sp_data["kpt3d"], "extrin": sp_data["extrin"]}
        keys = [k for k in ("tex_fg", "depth", "alpha", "tex_fg_fine", "depth_fine", "alpha_fine", "sdf") if k in out]
        spec = {"values": [out[k] for k in keys], "keys": keys, "names": names, "frame": frame, "pass": o, "sp_args": self.kwargs["sp_args"],
                "rays_per_chunk": self.kwargs.get("grad_rays_per_chunk", G.GRAD_RAYS_PER_CHUNK),
                "samples_per_block": self.kwargs.get("grad_samples_per_block", G.GRAD_SAMPLES_PER_BLOCK),
                "graph_blocks": bool(self.kwargs.get("grad_graph_blocks", False)), "hip_backward": None}
        if self.kwargs.get("hip_backward", True):
            # the per-sample networks' gradient on the fused HIP backward (config key hip_backward, default on; off = the PyTorch graph of
            # torch_graph.networks_at, which stays as the independent checker): fp32 weights of this step, the pass's per-frame tables
            spec["hip_backward"] = {"w0": self.packed_weights("fp32"), "fdat": fd, "block": int(self.kwargs.get("hip_backward_block", 65536))}
        for k, v in zip(keys, G.PassGradient.apply(spec, *leaves)):
            out[k] = v
        return out

    def forward(self, im, cam, hand_type, targets, data, bbox, n_views=1, sp_data={}, dr_data=None, **kwargs):
        """src/model.py:959-1024.  Returns dict(loss, err_dict, out={'nerf': out_nerf}) with the reference's out_nerf keys; the loss is
        compute_error(inter_loss=None, out_nerf, vggloss=self.vgg_loss, lambdas=cfg lambdas) as at src/model.py:1023."""
        assert len(im.shape) == 4 and len(cam["KRT"].shape) == 3
        # Training under autograd ("fused HIP forward + PyTorch autograd backward", SURVEY.md section 8 row f-4, first stage): the values
        # come from the HIP pass, the gradients from vanerf_amd.torch_graph evaluated at the same samples (attach_autograd below).
        autograd = torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())
        dr_kwargs = self.kwargs.get("dr_kwargs", {})
        if autograd and self.kwargs.get("graph_encoders") and not self._encoders_graphed and im.is_cuda:
            self._graph_encoders(im)
        feat_geo = self.attach_geo_feat(im, return_val=True)
        feat_tex = self.attach_tex_feat(im, return_val=True)
        n_batch = im.shape[0] // n_views
        stride = 0 if self.dr_level == 1 else torch.randint(high=(2 ** (self.dr_level - 1) - 1), size=(n_batch, 2))
        with torch.no_grad():
            out_nerf = self.batch_render_pifu_nerf(self, dr_data["img"], dr_data["cam"], hand_type, targets, n_views, dr_data["cam_tar"], self.dr_level,
                                                   stride, dr_data["tar"], [f.detach() for f in feat_geo], feat_tex.detach(), None, sp_data,
                                                   dr_data.get("objcenter", None), msk=dr_data["msk"], src_foreground_mask=kwargs["src_foreground_mask"],
                                                   bounds=kwargs["bounds"], _autograd=autograd, **{k: kwargs[k] for k in ("_draws",) if k in kwargs},
                                                   **dr_kwargs)
        if autograd:
            self.attach_autograd(out_nerf, dr_data["img"], feat_geo, feat_tex, targets, sp_data, kwargs["src_foreground_mask"])
        if self.disable_bg:
            out_nerf["tex_bg"] = 0.0
        out_nerf["tex"] = out_nerf["tex_cal"] = out_nerf["tex_fg"]
        if "tex_fg_fine" in out_nerf:
            out_nerf["tex_fine"] = out_nerf["tex_cal_fine"] = out_nerf["tex_fg_fine"]
        loss, err_dict = compute_error(inter_loss=None, out_nerf=out_nerf, vggloss=self.vgg_loss, lambdas=self.kwargs.get("lambdas", {}))
        return dict(loss=loss, err_dict=err_dict, out={"nerf": out_nerf})


def get_360cameras(headpose, focal, trans, sc_factor, im_w, im_h, znear, zfar, n_frames=90):
    """Orbit cameras of src/utils.py:63-134 (cv2.Rodrigues about the y axis written out; device follows `headpose`)."""
    device = headpose.device
    # inverse of the head pose.  The reference writes `T_i[:3, :3] = T_i[:3, :3].t()` (src/utils.py:67): source and destination alias.  A
    # device copy kernel reads every element before it writes (a true transpose, the intended inverse -- the reference's cameras live on the GPU);
    # torch's sequential CPU copy would leave a symmetrised matrix instead.  Here the transpose is taken from `headpose` itself: same on every device.
    T_i = torch.eye(4, device=device)
    T_i[:3, :3] = headpose[:3, :3].t()
    T_i[:3, 3] = -T_i[:3, :3] @ headpose[:3, 3]
    cams, theta0, theta1 = [], 0.0, 0.0
    for idx in range(n_frames):
        th32 = float(np.float32(theta0))  # the reference stores the angle in a float32 rotation vector before cv2.Rodrigues (src/utils.py:79-83)
        c, s = math.cos(th32), math.sin(th32)
        dR = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float32)  # Rodrigues((0, theta0, 0)), rounded to float32
        extrin_tar = torch.eye(4)
        intrin_tar = torch.eye(4)
        extrin_tar[:3, :3] = dR
        extrin_tar[:3, 3] = torch.tensor([0.0, 0.0, trans])
        intrin_tar[:3, :3] = torch.tensor([[focal, 0, im_w / 2], [0, focal, im_h / 2], [0, 0, 1]], dtype=torch.float32)
        extrinsic = torch.matmul(extrin_tar.to(device), T_i).clone()
        extrinsic[:3, 3] *= sc_factor
        i = idx + 0.0001
        d = 5.0 * math.pi * 0.1 / n_frames
        if 0 <= i <= n_frames / 10:
            theta0 += d; theta1 += d
        elif n_frames / 10 < i < n_frames * 3 / 10:
            theta0 -= d
        elif n_frames * 3 / 10 < i < n_frames * 5 / 10:
            theta1 -= d
        elif n_frames * 5 / 10 < i < n_frames * 7 / 10:
            theta0 += d
        elif n_frames * 7 / 10 < i < n_frames * 9 / 10:
            theta1 += d
        elif i >= n_frames * 9 / 10:
            theta0 -= d; theta1 -= d
        theta0 += 2.0 * math.pi / n_frames
        cams.append({"w2cs": extrinsic.to(device), "c2ws": torch.inverse(extrinsic).to(device), "intrinsics": intrin_tar.to(device).unsqueeze(0),
                     "im_w": im_w, "im_h": im_h, "znear": znear, "zfar": zfar})
    return cams
