"""CPU-side checks of the C ABI: the library loads, exports every symbol include/vanerf_hip.h declares, validates
its arguments without touching a GPU, and the weight packer places every weight exactly once."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ffi():
    from vanerf_amd import build
    build.build()  # in-tree hipcc build (cross-compiles gfx950 without a GPU); no-op when up to date
    from vanerf_amd import _ffi
    return _ffi


def test_exports_match_header(ffi):
    hdr = open(os.path.join(REPO, "include", "vanerf_hip.h")).read()
    declared = set(re.findall(r"\b(vanerf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(ffi.EXPORTS)
    for name in declared:
        assert hasattr(ffi.lib, name), name
    assert ffi.lib.vanerf_abi_version() == ffi.ABI_VERSION == 10


def test_error_convention(ffi):
    # null arguments: negative return code + message, no exception across the ABI, no GPU touched
    rc = ffi.lib.vanerf_composite(None, None, None, 4, 4, 0.1, None, None, None, None, None, None)
    assert rc == -22 and b"null" in ffi.lib.vanerf_last_error()
    rc = ffi.lib.vanerf_knn1(ctypes.c_void_p(8), 0, ctypes.c_void_p(8), 1, ctypes.c_void_p(8), None)
    assert rc == -22 and b"nv=0" in ffi.lib.vanerf_last_error()
    assert ffi.lib.vanerf_knn1(None, 4, None, 0, None, None) == 0  # an empty batch is valid
    with pytest.raises(ffi.VanerfError):
        ffi.check(rc)


def test_packer_places_every_weight_once(ffi, hot_weights):
    from vanerf_amd import renderer
    sd = dict(hot_weights)
    stream, offs = renderer.pack_weights_host(sd)
    assert len(offs) == 20 and offs[0] == 0 and all(b > a for a, b in zip(offs, offs[1:]))

    def layer(i):
        end = offs[i + 1] if i + 1 < len(offs) else stream.numel() - 2 * 64 * 4
        return stream[offs[i]:end]

    def eff(prefix):
        v, g = sd[prefix + "weight_v"], sd[prefix + "weight_g"]
        return torch.cat([(v * (g / v.norm(2, dim=1, keepdim=True))).flatten(), sd[prefix + "bias"]])

    expect = {
        0: sd["geo_vis_fusion.fconv_at.0.weight"].flatten(), 3: sd["geo_vis_fusion.fconv_ated.2.weight"].flatten(),
        8: eff("mlp_geo.layers1.layers.0.linear."), 9: eff("mlp_geo.layers1.layers.1.linear."), 10: eff("mlp_geo.layers1.layers.2.linear."),
        11: torch.cat([sd["mlp_geo.layers1.layers.3.linear.weight"].flatten(), sd["mlp_geo.layers1.layers.3.linear.bias"]]),
        12: eff("mlp_geo.layers2.layers.0.linear."), 15: torch.cat([sd["ibr_compress_gfeat.weight"].flatten(), sd["ibr_compress_gfeat.bias"]]),
        16: sd["tex_vis_fusion.fconv_at.0.weight"].flatten(), 19: sd["tex_vis_fusion.fconv.2.weight"][:3].flatten(),
    }
    for i, w in expect.items():
        got = layer(i)
        got = torch.sort(got[got != 0])[0]
        want = torch.sort(w[w != 0])[0]
        assert got.numel() == want.numel(), (i, got.numel(), want.numel())
        assert (got - want).abs().max() <= 1e-6 * max(1.0, want.abs().max().item()), i


def test_product_path_has_no_cpu_fallback(ffi):
    from vanerf_amd import renderer
    with pytest.raises(ValueError):
        renderer.knn1(torch.zeros(4, 4), torch.zeros(2, 3))  # CPU tensors are refused, never silently computed


def test_backward_spill_layout_is_consistent(ffi, hot_weights):
    """The host-only description of the fused backward's spills (vanerf_spill_rows / vanerf_layer_rows / vanerf_layer_slots): row bases tile the
    spills without overlap, and every layer's slot table names each input channel of the reference's parameter exactly as often as the kernel
    gathers it (once, except the duplicated pool/visibility operands), with a bias slot exactly where the reference layer has a bias."""
    from vanerf_amd import hip_backward as hb
    L = hb.layout()
    assert (L["x_rows"], L["y_rows"], L["aux_rows"], L["ig_rows"]) == (2074, 965, 110, 288)
    assert len(L["layers"]) == ffi.NUM_LAYERS == len(hb.LAYER_PARAMS)
    sd = dict(hot_weights)
    x_at = y_at = flat = 0
    for lay, spec in zip(L["layers"], hb.LAYER_PARAMS):
        assert lay["x_row"] == x_at and lay["y_row"] == y_at and lay["flat"] == flat
        x_at += lay["n_slots"]; y_at += lay["n_out"]; flat += lay["n_out"] * lay["n_slots"]
        ref = sd[spec[1]] if spec[0] == "conv" else sd[spec[1] + (".weight_v" if spec[0] == "wn" else ".weight")]
        kin = ref.shape[1]
        sl = lay["slots"]
        assert lay["n_out"] <= ref.shape[0]  # fconv.2: 3 of 40 output channels (one view)
        assert ((sl >= -2) & (sl < kin)).all()
        assert int((sl == -2).sum()) == (0 if spec[0] == "conv" else 1), spec
        seen = torch.bincount(sl[sl >= 0], minlength=kin)
        assert (seen >= 1).all(), (spec, (seen == 0).nonzero().view(-1).tolist())  # no channel of the parameter without a slot
    assert (x_at, y_at, flat) == (L["x_rows"], L["y_rows"], L["flat"])
    # argument checks of the host-only entry points
    assert ffi.lib.vanerf_layer_slots(20, None, 0) < 0 and ffi.lib.vanerf_layer_slots(-1, None, 0) < 0
    buf = (ctypes.c_int32 * 4)()
    assert ffi.lib.vanerf_layer_slots(8, ctypes.cast(buf, ctypes.c_void_p), 4) < 0  # capacity below the layer's slot count


def test_training_entry_points_validate_their_arguments(ffi):
    """The round-3 training entry points refuse null / inconsistent arguments with an error code before anything touches a GPU."""
    lib = ffi.lib
    n = ctypes.c_int64()
    assert lib.vanerf_weight_products_size(ctypes.byref(n)) == 0 and n.value > 100000
    assert lib.vanerf_weight_products(None, None, 65536, 64, 64, None, None) == -22 and b"null" in lib.vanerf_last_error()
    p = ctypes.c_void_p(8)
    assert lib.vanerf_weight_products(p, p, 1000, 64, 64, p, None) == -22 and b"multiple of 32" in lib.vanerf_last_error()
    assert lib.vanerf_weight_products(p, p, 65536, 128, 64, p, None) == -22  # more slices than the accumulator was laid out for
    assert lib.vanerf_composite_backward(None, p, p, p, 8, None, None, 0, None, 4, None, None, None, None, p, None, None, None) == -22
    assert lib.vanerf_weights_update(None, None, None, None) == -22 and b"null" in lib.vanerf_last_error()
    o, c, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.vanerf_ig_tensor(9, ctypes.byref(o), ctypes.byref(c), ctypes.byref(s)) < 0
    seen = []
    for which in range(9):
        assert lib.vanerf_ig_tensor(which, ctypes.byref(o), ctypes.byref(c), ctypes.byref(s)) == 9
        assert c.value <= s.value
        seen.append((o.value, s.value))
    assert all(a[0] + a[1] == b[0] for a, b in zip(seen, seen[1:])) and seen[-1][0] + seen[-1][1] == 288  # the tensors tile the 288 floats per sample
