"""The image-space loss `VANeRF.forward` returns (reference src/utils.py:159-178 compute_error, 222-287 compute_error_nerf, 289-328 pix_loss).

Host-side torch arithmetic on the rendered patch -- a few thousand pixels, not part of the per-sample hot path -- restated because the
reference's training_step adds to `forward(...)['loss']` (src/model.py:405): a drop-in forward has to return it.  The perceptual term is a
callable the caller supplies (`vggloss`; the reference builds a pretrained torchvision VGG19 in the constructor, src/utils.py:917-937, which
this package neither downloads nor ships): `None` leaves it out, exactly as the reference's own `if vggloss is not None` does.
Checked against values produced by the reference's functions in tests/test_model_interface.py (tests/golden/pass_train_16x16_s16.npz).
"""
import torch
import torch.nn.functional as thf


def pix_loss(src, tar, w_losses={"l1": 1.0}):
    """src/utils.py:289-328 (its `vis_img` argument is overwritten with None on entry, line 292, so it never weights anything)."""
    losses = {}
    for k, v in w_losses.items():
        if v <= 0.0:
            continue
        if k == "l1":
            losses[k] = (v * (src - tar).abs()).mean()
        elif k == "l2":
            losses[k] = (v * (src - tar).pow(2.0)).mean()
        elif k == "lp":
            losses[k] = (v * ((src - tar).abs() + 1e-4).pow(0.4)).mean()
        elif "l1top" in k or "l2top" in k:
            ratio = float(k[5:]) / 100.0
            d = (src - tar).abs() if "l1top" in k else (src - tar).pow(2.0)
            loss = torch.sort(v * d.sum(1).view(src.shape[0], -1), dim=-1, descending=True)[0]
            losses[k] = loss[:, :int(loss.shape[1] * ratio)].mean()
    return losses


def compute_error_nerf(out_nerf, lambdas, vggloss):
    """src/utils.py:222-287: coarse L1, fine pixel terms, optional alpha (mask) terms, optional perceptual term."""
    err = {}
    lambda_l1_c = lambdas.get("lambda_l1_c", 10.0)
    lambda_vgg = lambdas.get("lambda_vgg", 1.0)
    lambda_aux = lambdas.get("lambda_aux", 1.0)
    lambda_mloss = lambdas.get("lambda_mloss", 0.0)
    pix_weights = {"l1": lambdas.get("lambda_l1", 10.0), "l2": lambdas.get("lambda_l2", 0.0), "lp": lambdas.get("lambda_lp", 0.0),
                   "ssim": lambdas.get("lambda_ssim", 0.0)}
    tar = out_nerf["tar_img"]
    loss_pix_c = 0.0
    if "tex_cal" in out_nerf and lambda_l1_c > 0.0:
        loss_pix_c = loss_pix_c + pix_loss(out_nerf["tex_cal"], tar, {"l1": lambda_l1_c})["l1"]
    if "tex_aux_cal" in out_nerf and lambda_l1_c > 0.0 and lambda_aux > 0.0:
        loss_pix_c = loss_pix_c + lambda_aux * pix_loss(out_nerf["tex_aux_cal"], tar, {"l1": lambda_l1_c})["l1"]
    if loss_pix_c > 0.0:
        err["e_pix_c"] = loss_pix_c
    if "tex_cal_fine" in out_nerf:
        for k, v in pix_loss(out_nerf["tex_cal_fine"], tar, pix_weights).items():
            err[f"e_pix_{k}"] = v
    if "tex_aux_cal_fine" in out_nerf and lambda_aux > 0.0:
        for k, v in pix_loss(out_nerf["tex_aux_cal_fine"], tar, pix_weights).items():
            err[f"e_pix_{k}a"] = lambda_aux * v
    for key, name in (("alpha", "mask_loss_c"), ("alpha_fine", "mask_loss_f")):
        if key in out_nerf and "tar_alpha" in out_nerf and lambda_mloss > 0.0:
            err[name] = lambda_mloss * thf.mse_loss(out_nerf[key].clip(-0.001, 1.0).squeeze(), out_nerf["tar_alpha"].squeeze())
    if vggloss is not None:
        loss_vgg = 0.0
        if "tex_cal" in out_nerf:
            loss_vgg = loss_vgg + lambda_vgg * vggloss(out_nerf["tex_cal"], tar)
        if "tex_cal_fine" in out_nerf:
            loss_vgg = loss_vgg + lambda_vgg * vggloss(out_nerf["tex_cal_fine"], tar)
        if "tex_aux_cal_fine" in out_nerf and lambda_aux > 0.0:
            loss_vgg = loss_vgg + lambda_aux * lambda_vgg * vggloss(out_nerf["tex_aux_cal_fine"], tar)
        if loss_vgg > 0.0:
            err["e_vgg"] = loss_vgg
    return err


def compute_error(inter_loss=None, out_nerf=None, vggloss=None, lambdas={}):
    """src/utils.py:159-178 -> (loss, err_dict); err_dict['e_all'] is the sum.  The `uv_rendered_dict` branch (164-165) belongs to a UV
    renderer the shipped model never builds (no such key is ever produced by VANeRF.forward)."""
    if "uv_rendered_dict" in out_nerf and lambdas.get("lambda_l1_uv", 0.0) != 0:
        raise NotImplementedError("uv_rendered_dict: the UV renderer is not part of VANeRF.forward")
    err = dict(compute_error_nerf(out_nerf, lambdas, vggloss))
    loss = 0.0
    for v in err.values():
        loss = loss + v
    if inter_loss is not None:
        for k, v in inter_loss.items():
            loss = loss + v * 0.01
            err[k] = v
    err["e_all"] = loss
    return loss, err
