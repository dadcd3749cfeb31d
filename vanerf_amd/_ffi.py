"""ctypes binding of libvanerf_hip.so (the C ABI in include/vanerf_hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing this
module raises.  Build it with `python -m vanerf_amd.build` (or `__graft_entry__.build()`).
"""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: torch ships its own libamdhip64; if libvanerf_hip.so (linked against /opt/rocm's) is loaded before
#                          torch, two HIP runtimes live in the process and the second reports "no ROCm-capable device"
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_uint, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VANERF_HIP_LIB") or os.path.join(_HERE, "lib", "libvanerf_hip.so")  # the override is for A/B runs of kernel builds (tools/)
ABI_VERSION = 10
NUM_LAYERS = 20

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: the HIP extension is required (run `python -m vanerf_amd.build`)")
lib = ctypes.CDLL(LIB_PATH)

_FP = c_void_p  # device or host pointer to fp32 data, passed as an integer address


class VanerfWeightTable(Structure):
    _fields_ = [
        ("geo_at0_w1", _FP), ("geo_at0_w2", _FP), ("geo_ated0_w1", _FP), ("geo_ated0_w2", _FP),
        ("geo_at1_w1", _FP), ("geo_at1_w2", _FP), ("geo_ated1_w1", _FP), ("geo_ated1_w2", _FP),
        ("l1_v", _FP * 3), ("l1_g", _FP * 3), ("l1_b", _FP * 3), ("l1_w3", _FP), ("l1_b3", _FP),
        ("l2_v", _FP * 2), ("l2_g", _FP * 2), ("l2_b", _FP * 2), ("l2_w2", _FP), ("l2_b2", _FP),
        ("ibr_w", _FP), ("ibr_b", _FP),
        ("tex_at_w1", _FP), ("tex_at_w2", _FP), ("tex_w1", _FP), ("tex_w2", _FP),
        ("sigmoid_beta", c_float),
    ]


class VanerfFrame(Structure):
    _fields_ = [
        ("geo0", _FP), ("geo1", _FP), ("tex", _FP), ("img", _FP), ("mask", _FP),
        ("h0", c_int), ("w0", c_int), ("h1", c_int), ("w1", c_int), ("ht", c_int), ("wt", c_int), ("hi", c_int), ("wi", c_int),
        ("verts", _FP), ("vfeat0", _FP), ("vfeat1", _FP), ("vfeat_tex", _FP), ("vert_vis", _FP), ("kpt_cam", _FP),
        ("KRT", c_float * 12), ("extrin", c_float * 12),
        ("width", c_float), ("height", c_float), ("znear", c_float), ("zfar", c_float),
        ("invalid_sdf", c_float), ("pe_scale", c_float), ("pe_inv_2sigma2", c_float),
    ]


class VanerfMeshAccel(Structure):
    _fields_ = [
        ("tri", _FP), ("sphere", _FP), ("tnorm", _FP), ("orig", _FP), ("cbox", _FP), ("nfp", c_int), ("nc", c_int),
        ("cell_start", _FP), ("cell_tri", _FP), ("cell_rec", _FP), ("grid", _FP),
        ("vsort", _FP), ("vbox", _FP), ("nvc", c_int), ("cdisc", _FP),
    ]


class VanerfPassDesc(Structure):
    _fields_ = [
        ("x0", c_int), ("y0", c_int), ("step_x", c_int), ("step_y", c_int), ("y_block", c_int), ("nx", c_int), ("ny", c_int),
        ("pixels_xy", _FP), ("row_blocks", _FP), ("width", c_int), ("invK_T", c_float * 9), ("RT", c_float * 12), ("znear", c_float), ("zfar", c_float),
        ("bounds", c_float * 6), ("Sc", c_int), ("Sf", c_int), ("fine", c_int), ("reuse_coarse", c_int),
        ("t_lin_c", _FP), ("t_lin_f", _FP), ("jitter", _FP), ("u", _FP), ("noise_c", _FP), ("noise_f", _FP),
    ]


class VanerfPassOut(Structure):
    _fields_ = [("index", _FP), ("hit", _FP), ("z", _FP), ("color", _FP), ("depth", _FP), ("alpha", _FP), ("color_fine", _FP), ("depth_fine", _FP),
                ("alpha_fine", _FP), ("sdf", _FP), ("z_fine", _FP)]


_SIGS = {
    "vanerf_abi_version": (c_int, []),
    "vanerf_last_error": (c_char_p, []),
    "vanerf_mesh_cluster_size": (c_int, []),
    "vanerf_mesh_accel_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "vanerf_mesh_accel_build": (c_int, [_FP, c_int, _FP, c_int, c_int, c_int, c_void_p, c_int64, POINTER(VanerfMeshAccel), c_void_p]),
    "vanerf_weights_pack": (c_int, [POINTER(VanerfWeightTable), c_int, POINTER(c_void_p)]),
    "vanerf_weights_free": (c_int, [c_void_p]),
    "vanerf_weights_update": (c_int, [c_void_p, POINTER(VanerfWeightTable), _FP, c_void_p]),
    "vanerf_weights_short_groups": (c_int, [c_void_p, POINTER(c_uint64)]),
    "vanerf_weights_pack_host": (c_int, [POINTER(VanerfWeightTable), _FP, c_int64, POINTER(c_int64), POINTER(c_uint)]),
    "vanerf_weights_stream_host": (c_int, [POINTER(VanerfWeightTable), c_int, _FP, c_int64, POINTER(c_int64)]),
    "vanerf_weights_download": (c_int, [c_void_p, c_int, _FP, c_int64, POINTER(c_int64)]),
    "vanerf_ray_setup": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_float, c_float,
                                 POINTER(c_float), c_int, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_ray_setup_pixels": (c_int, [_FP, c_int, c_int, POINTER(c_float), POINTER(c_float), c_float, c_float, POINTER(c_float), c_int, _FP, _FP,
                                        _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_ray_setup_blocks": (c_int, [_FP, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_float, c_float,
                                        POINTER(c_float), c_int, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_sample_points": (c_int, [_FP, _FP, _FP, c_int, c_int, _FP, c_void_p]),
    "vanerf_vertex_visibility": (c_int, [_FP, _FP, c_int, _FP, c_int, c_int, _FP, _FP, c_void_p]),
    "vanerf_mesh_query": (c_int, [_FP, c_int, _FP, c_int, _FP, _FP, c_int64, _FP, _FP, _FP, c_void_p]),
    "vanerf_mesh_query_accel": (c_int, [POINTER(VanerfMeshAccel), _FP, c_int, _FP, c_int, _FP, _FP, c_int64, _FP, _FP, _FP, _FP, c_int, c_int, c_int,
                                        _FP, c_void_p]),
    "vanerf_knn1": (c_int, [_FP, c_int, _FP, c_int64, _FP, c_void_p]),
    "vanerf_query_samples": (c_int, [c_void_p, POINTER(VanerfFrame), _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int64, _FP, _FP, _FP, c_void_p]),
    "vanerf_query_forward_spill": (c_int, [c_void_p, POINTER(VanerfFrame), _FP, _FP, _FP, _FP, c_int64, c_int64, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_query_backward": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, c_int64, c_int64, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_weight_products": (c_int, [_FP, _FP, c_int64, c_int, c_int, _FP, c_void_p]),
    "vanerf_weight_products_size": (c_int, [POINTER(c_int64)]),
    "vanerf_spill_rows": (c_int, [POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "vanerf_layer_slots": (c_int, [c_int, _FP, c_int]),
    "vanerf_layer_rows": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "vanerf_query_order": (c_int, [POINTER(VanerfFrame), _FP, c_int64, _FP, _FP, c_int64, c_void_p]),
    "vanerf_query_order_scratch": (c_int64, [c_int64]),
    "vanerf_composite": (c_int, [_FP, _FP, _FP, c_int, c_int, c_float, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_composite_merged": (c_int, [_FP, _FP, c_int, _FP, _FP, c_int, _FP, _FP, c_int, c_float, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_eval_func": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_float, _FP, _FP, c_void_p]),
    "vanerf_composite_handle": (c_int, [c_void_p, _FP, _FP, _FP, c_int, _FP, _FP, c_int, _FP, c_int, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_composite_backward": (c_int, [c_void_p, _FP, _FP, _FP, c_int, _FP, _FP, c_int, _FP, c_int, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_importance_merge": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, _FP, _FP, _FP, _FP, c_void_p]),
    "vanerf_importance_sample": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, _FP, _FP, c_void_p]),
    "vanerf_render_pass_scratch": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "vanerf_render_pass": (c_int, [c_void_p, POINTER(VanerfFrame), POINTER(VanerfMeshAccel), _FP, c_int, _FP, c_int, POINTER(VanerfPassDesc),
                                   POINTER(VanerfPassOut), _FP, c_int64, c_void_p]),
    "vanerf_scatter_add_rows": (c_int, [_FP, _FP, _FP, c_int64, c_int64, c_int, _FP, c_int, c_void_p]),
    "vanerf_bilinear_taps": (c_int, [_FP, c_int64, c_int, c_int, _FP, _FP, c_void_p]),
    "vanerf_scatter_add_rows2": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_int64, c_int64, c_int, _FP, c_int, c_void_p]),
    "vanerf_scatter_add_taps": (c_int, [_FP, _FP, c_int64, _FP, c_int64, c_int64, c_int, _FP, c_int, c_void_p]),
    "vanerf_ig_tensor": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "vanerf_ray_bbox": (c_int, [POINTER(c_float), POINTER(c_float), _FP, c_int, _FP, _FP, _FP, c_void_p]),
}
EXPORTS = tuple(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export what the header declares
    _fn.restype = _res
    _fn.argtypes = _args

if lib.vanerf_abi_version() != ABI_VERSION:
    raise ImportError(f"libvanerf_hip.so ABI {lib.vanerf_abi_version()} != binding {ABI_VERSION}: rebuild")


class VanerfError(RuntimeError):
    pass


def check(rc):
    """C ABI convention: 0 = ok, negative = error with a thread-local message."""
    if rc != 0:
        msg = lib.vanerf_last_error()
        raise VanerfError(f"libvanerf_hip error {rc}: {msg.decode() if msg else '?'}")
