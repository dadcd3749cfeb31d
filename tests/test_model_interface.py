"""The drop-in boundary: vanerf_amd.model.VANeRF keeps the reference's constructor, method signatures, state_dict keys and
encoder output shapes (tests/golden/state_dict_keys.json and encoder_shapes.json were dumped from the reference module)."""
import inspect
import json
import os

import pytest
import torch

from tests.conftest import GOLDEN
from vanerf_amd import synth
from vanerf_amd.config import default_config


@pytest.fixture(scope="module")
def net():
    from vanerf_amd import build
    build.build()
    from vanerf_amd.model import VANeRF
    torch.manual_seed(0)
    return VANeRF(default_config()).eval()


def test_state_dict_keys_match_reference(net):
    want = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    got = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:10]
    assert got == want


def test_signatures_match_reference(net):
    from vanerf_amd.model import VANeRF
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(VANeRF.forward) == ["self", "im", "cam", "hand_type", "targets", "data", "bbox", "n_views", "sp_data", "dr_data", "kwargs"]
    assert sig(VANeRF.render_pifu_nerf) == ["self", "net", "img_in", "cam_in", "hand_type", "targets", "cam_tar", "level", "sp_data", "bkg_emb",
                                            "camcenter", "objcenter", "tar_img", "config"]
    assert sig(VANeRF.batch_render_pifu_nerf) == ["net", "img_in", "cam_in", "hand_type", "targets", "n_views", "cam_tar", "level", "stride", "tar_img",
                                                  "feat_geo", "feat_tex", "mano_vert_world", "sp_data", "objcenter", "config"]
    assert sig(VANeRF.query) == ["self", "pts", "cam", "hand_type", "targets", "feat_geo", "feat_tex", "vert", "vert_vis", "query_vis", "query_sdf",
                                 "closest_face", "n_views", "sp_data", "tx_data", "view", "n_pts_samples", "kwargs"]
    assert sig(VANeRF.importance_sample) == ["contrib", "z", "sample_per_ray", "uniform"]
    assert sig(VANeRF.rgba2out) == ["self", "rgba", "z", "vert_sdf"]
    assert sig(VANeRF.ray_bbox_intersection) == ["bounds", "orig", "direct", "boffset"]
    for name in ("attach_im_feat", "attach_geo_feat", "attach_tex_feat", "detach_im_feat", "sdf_activation"):
        assert callable(getattr(net, name))
    assert net.kwargs["dr_kwargs"]["sample_per_ray_c"] == 64 and net.dr_level == 5 and net.train_out_h == 64


def test_encoder_output_shapes(net):
    want = json.load(open(os.path.join(GOLDEN, "encoder_shapes.json")))
    with torch.no_grad():
        x = torch.rand(1, 3, 256, 256)
        fg = net.attach_geo_feat(x, return_val=True)
        ft = net.attach_tex_feat(x, return_val=True)
    assert [list(t.shape) for t in fg] == want["feat_geo"] and list(ft.shape) == want["feat_tex"]


def test_preconditions_raise_like_reference(net):
    frame = synth.make_frame(seed=3, tar_h=60, tar_w=64)
    kw = dict(fine=True, uniform=True, src_foreground_mask=frame["src_foreground_mask"], bounds=frame["bounds"])
    with pytest.raises(AssertionError):  # height % 2^(level-1) != 0 (src/model.py:1162)
        net.batch_render_pifu_nerf(net, frame["img_in"], frame["cam_in"], frame["hand_type"], frame["targets"], 1, frame["cam_tar"], 4, 0, None,
                                   frame["feat_geo"], frame["feat_tex"], None, frame["sp_data"], None, **kw)
    frame = synth.make_frame(seed=3, tar_h=64, tar_w=64)
    with pytest.raises(NotImplementedError):  # unsupported stride type (src/model.py:1169)
        net.batch_render_pifu_nerf(net, frame["img_in"], frame["cam_in"], frame["hand_type"], frame["targets"], 1, frame["cam_tar"], 4, 1.5, None,
                                   frame["feat_geo"], frame["feat_tex"], None, frame["sp_data"], None, **kw)
    with pytest.raises(AssertionError):  # stride >= 2^(level-1) (src/model.py:1164)
        net.batch_render_pifu_nerf(net, frame["img_in"], frame["cam_in"], frame["hand_type"], frame["targets"], 1, frame["cam_tar"], 2, 2, None,
                                   frame["feat_geo"], frame["feat_tex"], None, frame["sp_data"], None, **kw)


def test_orbit_cameras():
    from vanerf_amd.model import get_360cameras
    cams = get_360cameras(torch.eye(4)[:3, :4], 1500.0, 1.0, 1.0, 256, 256, 0.71, 1.42, n_frames=20)
    assert len(cams) == 20 and cams[0]["w2cs"].shape == (4, 4) and cams[0]["intrinsics"].shape == (1, 4, 4)
    assert torch.allclose(cams[0]["w2cs"] @ cams[0]["c2ws"], torch.eye(4), atol=1e-5)
    assert not torch.allclose(cams[0]["w2cs"], cams[7]["w2cs"])


def test_bicubic_upsampling_as_matrix_products_matches_the_library_call():
    """HourGlass up-sampling under autograd (vanerf_amd/encoders.py:_bicubic_up2) == F.interpolate(bicubic, align_corners=True), values and gradients."""
    import torch.nn.functional as F
    from vanerf_amd.encoders import _bicubic_up2
    g = torch.Generator().manual_seed(0)
    for shape in ((1, 5, 16, 12), (2, 3, 7, 9), (1, 4, 2, 2)):
        x = torch.randn(shape, generator=g, requires_grad=True)
        a, b = _bicubic_up2(x), F.interpolate(x, scale_factor=2, mode="bicubic", align_corners=True)
        assert a.shape == b.shape and (a - b).abs().max() <= 2e-6
        go = torch.randn(a.shape, generator=g)
        ga, = torch.autograd.grad(a, x, go)
        gb, = torch.autograd.grad(b, x, go)
        assert (ga - gb).abs().max() <= 2e-5
        with torch.no_grad():
            assert torch.equal(_bicubic_up2(x.detach()), b.detach())  # inference: the library call itself
