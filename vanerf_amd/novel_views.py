"""Novel-view driver on top of the HIP renderer: the callers of the hot path for orbit videos (BASELINE config 4).

Mirrors, with the same names and argument meaning,
  VANeRFLightningModule.render_novel_views   reference src/model.py:513-545
  VANeRFLightningModule.render_video         reference src/model.py:140-197
  _arrange_nerf_images / _arrange_src_images reference src/model.py:461-486
  get_360cameras                             reference src/utils.py:63-134 (vanerf_amd.model.get_360cameras)

What is different, by design:
  * the source frame's features, vertex tables and mesh acceleration structure are built ONCE (VANeRF.frame_data caches them on the
    identity of the inputs); the reference re-runs both encoders for every stride pass of every frame and calls
    torch.cuda.empty_cache() + gc.collect() after every batch (src/model.py:187-188);
  * with world_size > 1 (one process per GPU, torch.distributed over RCCL) the orbit is sharded either by FRAMES (default: rank r renders
    frames r, r + world, ... -- no collective on the data path; an optional all-gather returns the whole stack to every rank) or by RAYS
    (shard="rays", BASELINE config 4 as written: every rank marches its interleaved 8-row blocks of EVERY frame and one all_gather of the
    tiles per frame assembles the image on every rank, vanerf_amd.parallel.gather_image -- lowest latency per frame, the same layout as
    bench.py --gpus N);
  * PNG encoding runs on worker threads behind the renderer: the device -> host copy of a frame is issued on a side stream into
    pinned memory and the worker waits on its event, so the GPU never waits for the encoder.
cv2 / imageio are not required: PNG and GIF go through PIL (the reference's .mp4 needs cv2.VideoWriter and is not written).
"""
import math
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.nn.functional as F

from .model import get_360cameras


def camera_to_cam_tar(camera):
    """The cam_tar dict the reference builds per orbit camera (src/model.py:523-529)."""
    w2c = camera["w2cs"].unsqueeze(0)
    return {"K": camera["intrinsics"], "RT": w2c, "KRT": camera["intrinsics"] @ w2c, "width": camera["im_w"], "height": camera["im_h"],
            "nml_scale": 100.0, "znear": camera["znear"], "zfar": camera["zfar"]}


def arrange_nerf_images(out_nerf):
    """src/model.py:461-465: (1,3,H,W) or (3,H,W) fine colour -> (H,W,3) float32 in [0,1], still on the device."""
    tex = out_nerf["tex_fg_fine"].clamp(min=0.0, max=1.0)
    if tex.dim() == 4:
        tex = tex.squeeze(0)
    return tex.permute(1, 2, 0).contiguous()


def arrange_src_images(im, render_size=None):
    """src/model.py:467-486: (B,3,H,W) -> (H', B*W', 3) float32 numpy, scaled so that max(H,W) == render_size.
    cv2.INTER_AREA is torch's 'area' interpolation for the integer shrink factors the callers use (256 -> 256, 512 -> 256)."""
    if render_size is not None:
        sc = render_size / max(im.shape[-2:])
        if sc != 1.0:
            im = F.interpolate(im, size=(int(round(im.shape[-2] * sc)), int(round(im.shape[-1] * sc))), mode="area" if sc < 1.0 else "bilinear",
                               **({} if sc < 1.0 else {"align_corners": False}))
    return np.concatenate([im[b].permute(1, 2, 0).cpu().numpy() for b in range(im.shape[0])], axis=1)


def frames_of_rank(n_frames, rank, world):
    """Frame indices rendered by `rank`: interleaved, so ranks stay balanced when the cost varies smoothly along the orbit."""
    return list(range(rank, n_frames, world))


def _default_render(net, tr_batch, cam_tar, nerf_level):
    """render_full_nerf_image (src/model.py:488-511) on the HIP path."""
    dr = tr_batch["dr_data"]
    kw = net.kwargs["dr_kwargs"]
    return net.render_pifu_nerf(None, net, tr_batch["im"], tr_batch["cam"], tr_batch["hand_type"], tr_batch["targets"], cam_tar, level=nerf_level,
                                sp_data=dict(tr_batch["sp_data"]), objcenter=dr.get("objcenter"), tar_img=dr.get("tar"), fine=kw["fine"], uniform=True,
                                objrad=250.0, blur=3, sample_per_ray_c=kw["sample_per_ray_c"], sample_per_ray_f=kw["sample_per_ray_f"],
                                src_foreground_mask=tr_batch["src_foreground_mask"], bounds=dr["bounds"], mask_at_box=dr.get("mask_at_box"))


def _default_render_rows(net, tr_batch, cam_tar, rank, world):
    """This rank's rows of one frame on the HIP path: (rows * width, 3) fine colour, rows dealt out by parallel.shard_rows."""
    from . import renderer as R
    from .parallel import shard_rows
    kw = net.kwargs["dr_kwargs"]
    feat_geo, feat_tex = net.encoded(tr_batch["im"])
    fd = net.frame_data(tr_batch["im"], tr_batch["cam"], tr_batch["targets"], feat_geo, feat_tex, tr_batch["sp_data"], tr_batch["src_foreground_mask"])
    cam_t = dict(cam_tar, znear=cam_tar.get("znear", tr_batch["cam"]["znear"]), zfar=cam_tar.get("zfar", tr_batch["cam"]["zfar"]))
    y0, y_step, ny, yb = shard_rows(int(cam_t["height"]), world, rank)
    o = R.render_pass(net.packed_weights(), fd, cam_t, tr_batch["dr_data"]["bounds"], 0, y0, 1, int(cam_t["width"]), ny, kw["sample_per_ray_c"],
                      kw["sample_per_ray_f"], fine=kw["fine"], y_step=y_step, y_block=yb)
    return o["color_fine"] if kw["fine"] else o["color"]


def _gather_rows(tile, h, w, world):
    """parallel.gather_image with a host-staged fall-back for the gloo backend (CPU rehearsals)."""
    import torch.distributed as dist
    from .parallel import deinterleave, gather_image
    if world == 1 or dist.get_backend() == "nccl":
        return gather_image(tile, h, w, world)
    parts = [torch.empty_like(tile, device="cpu") for _ in range(world)]
    dist.all_gather(parts, tile.cpu().contiguous())
    return deinterleave(torch.cat(parts, 0), h, w, world, tile.shape[1]).to(tile.device)


@torch.no_grad()
def render_novel_views(net, cameras, tr_batch, only_renderings=False, rank=0, world=1, gather=False, render_fn=None, on_frame=None,
                       shard="frames", render_rows_fn=None):
    """src/model.py:513-545.  cameras: list of get_360cameras dicts; tr_batch: the decode_batch dict (im, cam, hand_type, targets,
    sp_data, src_foreground_mask, dr_data{bounds, objcenter, mask_at_box}).
    Returns uint8 (N,H,W,3) renderings with the source views pasted to their left, or (renderings, source images) when
    only_renderings.  With world > 1 a rank returns ITS frames (frames_of_rank) unless gather=True, which all-gathers the stack.
    on_frame(frame_index, uint8 HWC device tensor) is called as each frame completes (render_video hands them to the PNG writers).
    render_fn(net, tr_batch, cam_tar, level) -> out_nerf replaces the renderer (tests of the scheduling use a stub).
    shard="rays": every rank renders ITS ROWS of every frame (render_rows_fn(net, tr_batch, cam_tar, rank, world) -> (rows * W, 3) tile,
    default: the HIP path) and the frame is assembled on every rank by one all_gather; every rank then returns the whole orbit."""
    if shard not in ("frames", "rays"):
        raise ValueError("shard must be 'frames' or 'rays'")
    by_rays = shard == "rays" and world > 1
    render_fn = render_fn or _default_render
    render_rows_fn = render_rows_fn or _default_render_rows
    if hasattr(net, "encoded"):
        net.encoded(tr_batch["im"])  # once per source frame (src/model.py:517); the maps are kept for every target view of the orbit
    elif hasattr(net, "attach_im_feat"):
        net.attach_im_feat(tr_batch["im"])
    tr_batch["dr_data"]["tar"] = None
    mine = list(range(len(cameras))) if by_rays else frames_of_rank(len(cameras), rank, world)
    if (render_rows_fn is _default_render_rows if by_rays else render_fn is _default_render) and mine:  # the camera matrices go to the kernels by value: one read-back for the whole orbit, not one per frame
        from . import renderer as R
        R.prefetch_host_copies([cameras[fi][k] for fi in mine for k in ("intrinsics", "w2cs")] + [tr_batch["dr_data"]["bounds"]])
    frames = []
    for fi in mine:
        camera = cameras[fi]
        nerf_level = max(0, int(math.log(camera["im_h"], 2)) - 5)
        cam_tar = camera_to_cam_tar(camera)
        tr_batch["dr_data"]["cam_tar"] = cam_tar
        if by_rays:
            full = _gather_rows(render_rows_fn(net, tr_batch, cam_tar, rank, world), int(camera["im_h"]), int(camera["im_w"]), world)
            img = (full.clamp(min=0.0, max=1.0) * 255.0).to(torch.uint8)  # (H, W, 3), as arrange_nerf_images
        else:
            img = (arrange_nerf_images(render_fn(net, tr_batch, cam_tar, nerf_level)) * 255.0).to(torch.uint8)
        if on_frame is not None:
            on_frame(fi, img)
        frames.append(img)
    h, w = cameras[0]["im_h"], cameras[0]["im_w"]
    dev = frames[0].device if frames else tr_batch["im"].device
    stack = torch.stack(frames) if frames else torch.empty(0, h, w, 3, dtype=torch.uint8, device=dev)
    if world > 1 and gather and not by_rays:
        import torch.distributed as dist
        per = (len(cameras) + world - 1) // world
        pad = torch.zeros(per, h, w, 3, dtype=torch.uint8, device=dev)
        pad[: stack.shape[0]] = stack
        if dist.get_backend() == "nccl":
            allf = torch.empty(world * per, h, w, 3, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(allf, pad)
        else:
            parts = [torch.empty_like(pad, device="cpu") for _ in range(world)]
            dist.all_gather(parts, pad.cpu())
            allf = torch.cat(parts, 0).to(dev)
        # rank r's k-th frame is frame r + k * world
        stack = allf.view(world, per, h, w, 3).permute(1, 0, 2, 3, 4).reshape(world * per, h, w, 3)[: len(cameras)]
    rgb = stack.cpu().numpy()
    if only_renderings:
        src = (np.stack([tr_batch["im"][b].permute(1, 2, 0).cpu().numpy() for b in range(tr_batch["im"].shape[0])]) * 255.0).astype(np.uint8)
        return rgb, src
    src = (arrange_src_images(tr_batch["im"], h) * 255.0).astype(np.uint8)
    return np.concatenate((np.repeat(src[None], rgb.shape[0], axis=0), rgb), axis=-2)


class AsyncImageWriter:
    """PNG writers behind the renderer.  submit(path, uint8 HWC device tensor): the copy to pinned host memory is issued on a side
    stream ordered after the producer's stream; a worker thread waits for it and encodes.  close() drains."""

    def __init__(self, workers=4):
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.futures = []
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.lock = threading.Lock()

    def submit(self, path, img):
        if img.is_cuda:
            host = torch.empty(img.shape, dtype=img.dtype, pin_memory=True)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                host.copy_(img, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            img.record_stream(self.stream)
        else:
            host, ev = img, None

        def work():
            from PIL import Image
            if ev is not None:
                ev.synchronize()
            Image.fromarray(host.numpy()).save(path)
            return path

        with self.lock:
            self.futures.append(self.pool.submit(work))

    def close(self):
        done = [f.result() for f in self.futures]
        self.pool.shutdown()
        return done


@torch.no_grad()
def render_video(net, batches, save_dir, decode_batch=lambda b: b, sc_factor=1.0, label="", n_frames=20, rank=0, world=1, render_fn=None,
                 video_dirname="video", shard="frames", render_rows_fn=None):
    """src/model.py:140-197: one orbit per batch, frames as <save_dir>/<video_dirname><label>/<session>/<identity>/%06d.png, then a GIF per
    identity (PIL; the reference also writes an .mp4 through cv2, which this image does not have).  `batches` yields the dataloader's
    dicts ('index'.'segment', 'human', 'headpose', optional 'near_fars'); decode_batch maps one to the tr_batch of render_novel_views.
    Camera constants are the reference's (trans 10, 256x256, focal from the 30x..0.05x sweep at 1 %)."""
    trans = 10
    znear, zfar = (trans - 5.0) * sc_factor, (trans + 5.0) * sc_factor
    im_w, im_h = 256, 256
    fstart, fend = im_w * 30, im_w * 0.05
    focal = fstart + 0.01 * (fend - fstart)
    dst_dir = os.path.join(save_dir, f"{video_dirname}{label}")
    cameras, sub_dirs = {}, set()
    writer = AsyncImageWriter()
    for batch in batches:
        if "near_fars" in batch:
            batch["near_fars"][..., 0] = znear
            batch["near_fars"][..., 1] = zfar
        session, identity = str(batch["index"]["segment"][0]), str(int(batch["human"][0]))
        sub = os.path.join(dst_dir, session, identity)
        sub_dirs.add(sub)
        os.makedirs(sub, exist_ok=True)
        if identity not in cameras:
            cameras[identity] = get_360cameras(batch["headpose"][0], focal, trans, sc_factor, im_w, im_h, znear, zfar, n_frames)
        tr_batch = decode_batch(batch)
        src = (arrange_src_images(tr_batch["im"], im_h) * 255.0).astype(np.uint8)
        src_dev = torch.from_numpy(src).to(tr_batch["im"].device)

        def on_frame(fi, img, sub=sub, src_dev=src_dev):
            if shard == "rays" and world > 1 and fi % world != rank:  # every rank holds every frame: the PNGs are dealt out like the frames are otherwise
                return
            writer.submit(os.path.join(sub, f"{fi:06d}.png"), torch.cat((src_dev, img), dim=1))  # source views | rendering, as the reference

        render_novel_views(net, cameras[identity], tr_batch, only_renderings=True, rank=rank, world=world, render_fn=render_fn, on_frame=on_frame,
                           shard=shard, render_rows_fn=render_rows_fn)
    written = writer.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if rank == 0:
        from PIL import Image
        for sub in sorted(sub_dirs):
            frames = [Image.open(os.path.join(sub, f"{fi:06d}.png")).convert("RGB") for fi in range(n_frames)]
            frames[0].save(f"{sub}_nvs.gif", save_all=True, append_images=frames[1:], duration=100, loop=0)
    return written
