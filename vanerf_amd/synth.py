"""Seeded synthetic frames for tests, smoke() and bench.py (no dataset, no MANO pickle).

Shapes follow what the reference renderer consumes: two closed
genus-0 "hands" of 779 vertices / 1554 faces each (the vertex/face counts of a sealed MANO
hand, reference src/dataset.py:35-52, so NV = 1558, NF = 3108 and the twin-vertex roll of
src/networks.py:30-32 is valid), 42 key points, a 256x256 source image and the three feature
maps the image encoders produce (src/model.py:971-972).
"""
import math

import numpy as np
import torch

NV_HAND = 779
NF_HAND = 1554


def uv_sphere(rings=21, segs=37):
    """Unit sphere, outward CCW winding: rings*segs + 2 vertices, 2*segs*rings faces."""
    verts = [(0.0, 0.0, 1.0)]
    for r in range(1, rings + 1):
        th = math.pi * r / (rings + 1)
        for s in range(segs):
            ph = 2.0 * math.pi * s / segs
            verts.append((math.sin(th) * math.cos(ph), math.sin(th) * math.sin(ph), math.cos(th)))
    verts.append((0.0, 0.0, -1.0))
    faces = []
    ring0 = lambda r: 1 + r * segs
    for s in range(segs):
        faces.append((0, ring0(0) + s, ring0(0) + (s + 1) % segs))
    for r in range(rings - 1):
        for s in range(segs):
            a, b = ring0(r) + s, ring0(r) + (s + 1) % segs
            c, d = ring0(r + 1) + s, ring0(r + 1) + (s + 1) % segs
            faces.append((a, c, d))
            faces.append((a, d, b))
    last = len(verts) - 1
    for s in range(segs):
        faces.append((last, ring0(rings - 1) + (s + 1) % segs, ring0(rings - 1) + s))
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


def two_hand_mesh(seed=0, radius=0.05, offset=0.04, depth=1.0, bumpy=True):
    """(1558,3) float32 vertices, (3108,3) int64 faces; hand 1 is a translated copy of hand 0's topology."""
    rng = np.random.RandomState(seed)
    sv, sf = uv_sphere()
    assert sv.shape[0] == NV_HAND and sf.shape[0] == NF_HAND
    hands = []
    for h, cx in enumerate((-offset, offset)):
        v = sv.copy()
        if bumpy:  # low-frequency radial displacement keeps the surface closed and star-shaped
            amp = rng.uniform(0.05, 0.15, size=3)
            frq = rng.randint(1, 4, size=3)
            ph = rng.uniform(0, 2 * math.pi, size=3)
            r = 1.0 + amp[0] * np.sin(frq[0] * np.arctan2(v[:, 1], v[:, 0]) + ph[0]) * (1 - v[:, 2] ** 2) \
                + amp[1] * np.sin(frq[1] * math.pi * v[:, 2] + ph[1]) * (1 - v[:, 2] ** 2)
            v = v * r[:, None]
            v = v * np.array([1.0, 1.3, 0.7])
        v = v * radius + np.array([cx, 0.003 * (2 * h - 1), depth])
        hands.append(v)
    verts = np.concatenate(hands, 0).astype(np.float32)
    faces = np.concatenate([sf, sf + NV_HAND], 0)
    return verts, faces


def look_at_extrinsic(eye, target, up=(0.0, -1.0, 0.0)):
    """World->camera 4x4 (x right, y down, z forward)."""
    eye, target, up = (np.asarray(a, dtype=np.float64) for a in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(-up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], 0)
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ eye
    return E.astype(np.float32)


def make_frame(seed=0, tar_h=64, tar_w=64, src_hw=256, orbit_deg=8.0, device="cpu", half_mask=False,
               focal_src=1500.0):
    """One synthetic frame: dict with every tensor `batch_render_pifu_nerf` needs (B = V = 1).

    Keys mirror the reference call (src/model.py:1102-1120, 313-317, 345-349):
      img_in (1,3,256,256), feat_geo [(1,64,32,32),(1,8,128,128)], feat_tex (1,8,64,64),
      cam_in, cam_tar, targets{vert_world, face_world, tar_cam}, sp_data{extrin,kpt3d},
      src_foreground_mask (1,1,1,256,256), bounds (1,2,3).
    """
    g = torch.Generator().manual_seed(seed)
    verts_np, faces_np = two_hand_mesh(seed)
    verts = torch.from_numpy(verts_np)[None]
    faces = torch.from_numpy(faces_np)[None].float()  # reference passes faces as float and calls .long()
    centre = verts[0].mean(0)

    kpts = []
    for h in range(2):
        c = verts[0, h * NV_HAND:(h + 1) * NV_HAND].mean(0)
        kpts.append(c[None] + 0.03 * torch.randn(21, 3, generator=g))
    kpt3d = torch.cat(kpts, 0)[None]

    znear, zfar = 0.71, 1.42
    # source camera: identity extrinsic, principal point at the image centre
    K_src = torch.eye(4)
    K_src[0, 0] = K_src[1, 1] = focal_src
    K_src[0, 2] = K_src[1, 2] = src_hw / 2.0
    E_src = torch.eye(4)
    cam_in = {
        "KRT": (K_src @ E_src)[None], "K": K_src[None], "Rt": E_src[None, :3, :4], "extrin": E_src[None],
        "znear": znear, "zfar": zfar, "width": src_hw, "height": src_hw, "nml_scale": 100.0,
    }
    # target camera: orbit about the scene centre
    ang = math.radians(orbit_deg)
    dist = float(centre[2])
    eye = np.array([centre[0].item() + dist * math.sin(ang), centre[1].item() - 0.02, centre[2].item() - dist * math.cos(ang)])
    E_tar = torch.from_numpy(look_at_extrinsic(eye, centre.numpy()))
    K_tar = torch.eye(4)
    f_tar = focal_src * tar_w / src_hw * 0.9
    K_tar[0, 0] = K_tar[1, 1] = f_tar
    K_tar[0, 2] = tar_w / 2.0
    K_tar[1, 2] = tar_h / 2.0
    cam_tar = {
        "K": K_tar[None], "RT": E_tar[None], "KRT": (K_tar @ E_tar)[None],
        "width": tar_w, "height": tar_h, "nml_scale": 100.0, "znear": znear, "zfar": zfar,
    }
    bmin = verts[0].min(0)[0].clone()
    bmax = verts[0].max(0)[0].clone()
    bmin[2] -= 0.05
    bmax[2] += 0.05
    bounds = torch.stack([bmin, bmax], 0)[None]

    img = torch.rand(1, 3, src_hw, src_hw, generator=g)
    mask = torch.ones(1, 1, 1, src_hw, src_hw)
    if half_mask:
        mask[..., : src_hw // 2 - 9] = 0.0
    feat_geo = [torch.rand(1, 64, 32, 32, generator=g) * 2 - 1, torch.rand(1, 8, 128, 128, generator=g) * 2 - 1]
    feat_tex = torch.rand(1, 8, 64, 64, generator=g) * 2 - 1
    targets = {
        "vert_world": verts, "face_world": faces,
        "tar_cam": {"tar_R": E_tar[None, :3, :3], "tar_T": E_tar[None, :3, 3],
                    "tar_focal": torch.tensor([[f_tar, f_tar]]), "tar_princpt": torch.tensor([[tar_w / 2.0, tar_h / 2.0]])},
    }
    frame = {
        "img_in": img, "feat_geo": feat_geo, "feat_tex": feat_tex, "cam_in": cam_in, "cam_tar": cam_tar,
        "targets": targets, "sp_data": {"extrin": E_src[None].clone(), "kpt3d": kpt3d},
        "src_foreground_mask": mask, "bounds": bounds, "hand_type": torch.ones(1, 2),
    }
    return to_device(frame, device)


def to_tr_batch(frame):
    """The dict VANeRFLightningModule.decode_batch hands to the renderer (reference src/model.py:213-262), filled from a synthetic frame."""
    return {"im": frame["img_in"], "cam": frame["cam_in"], "hand_type": frame["hand_type"], "targets": frame["targets"], "sp_data": frame["sp_data"],
            "src_foreground_mask": frame["src_foreground_mask"],
            "dr_data": {"tar": None, "cam_tar": frame["cam_tar"], "objcenter": None, "bounds": frame["bounds"], "mask_at_box": None}}


def to_device(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, dict):
        return {k: to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_device(v, device) for v in obj)
    return obj


# state_dict keys / shapes of the per-sample networks (reference VANeRF.state_dict(), shipped config)
HOT_SHAPES = {
    "sigmoid_beta": (1,),
    "geo_vis_fusion.fconv_at.0.weight": (10, 196, 1), "geo_vis_fusion.fconv_at.2.weight": (3, 10, 1),
    "geo_vis_fusion.fconv_ated.0.weight": (64, 196, 1), "geo_vis_fusion.fconv_ated.2.weight": (64, 64, 1),
    "geo_vis_fusion.fconv_at1.0.weight": (10, 28, 1), "geo_vis_fusion.fconv_at1.2.weight": (3, 10, 1),
    "geo_vis_fusion.fconv_ated1.0.weight": (8, 28, 1), "geo_vis_fusion.fconv_ated1.2.weight": (8, 8, 1),
    "tex_vis_fusion.fconv.0.weight": (96, 96, 1), "tex_vis_fusion.fconv.2.weight": (40, 96, 1),
    "tex_vis_fusion.fconv_at.0.weight": (96, 96, 1), "tex_vis_fusion.fconv_at.2.weight": (6, 96, 1),
    "ibr_compress_gfeat.weight": (24, 128), "ibr_compress_gfeat.bias": (24,),
    "mlp_geo.layers1.layers.0.linear.weight_v": (128, 358), "mlp_geo.layers1.layers.1.linear.weight_v": (128, 128),
    "mlp_geo.layers1.layers.2.linear.weight_v": (120, 136), "mlp_geo.layers1.layers.3.linear.weight": (64, 120),
    "mlp_geo.layers2.layers.0.linear.weight_v": (64, 128), "mlp_geo.layers2.layers.1.linear.weight_v": (64, 64),
    "mlp_geo.layers2.layers.2.linear.weight": (2, 64),
}


def make_hot_weights(seed=0):
    """'Trained-like' random weights for the per-sample networks: activations are O(0.1-1) at every layer, both
    sides of every ReLU / Softplus knee / validity branch are exercised and the density head produces alpha > 0 for
    roughly half of the samples.  (The reference's own init leaves alpha == 0 everywhere and colours ~1e-2, which
    would make a 1e-4 absolute parity bar meaningless.)  Returns {state_dict key: fp32 tensor}."""
    g = torch.Generator().manual_seed(1000 + seed)
    sd = {}
    for k, shp in HOT_SHAPES.items():
        if k == "sigmoid_beta":
            sd[k] = torch.full(shp, 0.1)
            continue
        fan_in = shp[1] if len(shp) > 1 else shp[0]
        if k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan_in)
    for k in list(sd):
        if k.endswith("weight_v"):
            base = k[: -len("weight_v")]
            sd[base + "weight_g"] = 0.25 + 0.5 * torch.rand(sd[k].shape[0], 1, generator=g)
            sd[base + "bias"] = 0.04 * torch.randn(sd[k].shape[0], generator=g)
    sd["mlp_geo.layers1.layers.3.linear.weight"] *= 0.7
    sd["mlp_geo.layers1.layers.3.linear.bias"] = 0.05 * torch.randn(64, generator=g)
    for i in (0, 1):  # the pooled latent is small (mean ~0.05, variance ~1e-3): give the density head some gain
        sd[f"mlp_geo.layers2.layers.{i}.linear.weight_g"] = 2.0 + 2.0 * torch.rand(64, 1, generator=g)
        sd[f"mlp_geo.layers2.layers.{i}.linear.bias"] = 0.1 * torch.randn(64, generator=g)
    w_last = sd["mlp_geo.layers2.layers.2.linear.weight"]
    w_last -= w_last.mean(1, keepdim=True)  # inputs are post-softplus (positive): zero-mean rows keep rad centred
    sd["mlp_geo.layers2.layers.2.linear.weight"] = w_last * 1.2
    sd["mlp_geo.layers2.layers.2.linear.bias"] = torch.tensor([0.28, -0.47])
    sd["geo_vis_fusion.fconv_at.2.weight"] *= 2.0
    sd["geo_vis_fusion.fconv_at1.2.weight"] *= 2.0
    sd["tex_vis_fusion.fconv_at.2.weight"] *= 2.0
    sd["tex_vis_fusion.fconv.2.weight"] *= 0.6
    return sd


def make_texframe_weights():
    """Per-frame TexVisFusion conv stack with the reference's init recipe (src/model.py:669-698): torch.manual_seed(125)
    before each module, normal_(0, 0.02); LayerNorm affine = (1, 0)."""
    shapes = {"fconv_gt.0.weight": (779, 42, 3), "fconv_gt.3.weight": (1558, 779, 3), "fconv3.0.weight": (21, 8, 3, 3),
              "fconv3.3.weight": (42, 21, 3, 3), "fconv4.0.weight": (21, 3, 3, 3), "fconv4.3.weight": (42, 21, 3, 3)}
    ln = {"fconv_gt.1": (18,), "fconv_gt.4": (18,), "fconv3.1": (64, 64), "fconv3.4": (64, 64), "fconv4.1": (256, 256), "fconv4.4": (256, 256)}
    sd = {}
    state = torch.get_rng_state()
    for k, s in shapes.items():
        torch.manual_seed(125)
        sd["tex_vis_fusion." + k] = torch.empty(s).normal_(0.0, 0.02)
    torch.set_rng_state(state)
    for k, s in ln.items():
        sd[f"tex_vis_fusion.{k}.weight"] = torch.ones(s)
        sd[f"tex_vis_fusion.{k}.bias"] = torch.zeros(s)
    return sd


def make_full_weights(seed=0):
    sd = make_hot_weights(seed)
    sd.update(make_texframe_weights())
    sd.update(make_ibr_weights(seed))
    return sd


IBR_SHAPES = {
    "ray_encoder.0": (16, 4), "ray_encoder.2": (40, 16), "base_layer.0": (64, 120), "base_layer.2": (32, 64),
    "vis_layer1.0": (32, 32), "vis_layer1.2": (33, 32), "vis_layer2.0": (32, 32), "vis_layer2.2": (1, 32),
    "out_layer.0": (16, 37), "out_layer.2": (8, 16), "out_layer.4": (1, 8),
}


def make_ibr_weights(seed=0):
    """IBRRenderingHead parameters (src/model.py:1575-1591).  At V = 1 the head returns rgb_feat[..., :3] exactly
    (softmax over one view), so these only matter to the oracle's restatement and to state_dict completeness."""
    g = torch.Generator().manual_seed(2000 + seed)
    sd = {"mlp_tex.ani_al": torch.tensor(0.2)}
    for k, (o, i) in IBR_SHAPES.items():
        sd[f"mlp_tex.{k}.weight"] = torch.randn(o, i, generator=g) * math.sqrt(2.0 / i)
        sd[f"mlp_tex.{k}.bias"] = torch.zeros(o)
    return sd
