"""Honesty check of the CPU baseline: the REFERENCE's own batch_render_pifu_nerf and this repo's oracle port timed on the
same pass in the build container (64x64 strided rays of a 256x256 view, 64 + 64 samples per ray, all host threads).  bench.py can only time the
port on the GPU box (the reference does not travel); this script says how the two relate.  Build-container only: python -m oracle.time_reference"""
import copy
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import vanerf_oracle as orc  # noqa: E402
from oracle.ref_import import import_reference  # noqa: E402
from vanerf_amd import synth  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count())
    ref = import_reference()
    M, Nw = ref.model, ref.networks
    Nw.knn_points = lambda q, v, K=1: (None, torch.stack([orc.knn1(q[b], v[b]) for b in range(q.shape[0])], 0)[..., None], None)
    M.cal_vis_sdf_batch = orc.cal_vis_sdf_batch
    M.render_vis = lambda *a, **k: (torch.zeros(1, 3, 256, 256), torch.zeros(1, 1, 256, 256))
    cfg = json.load(open("/root/reference/configs/vanerf.json"))
    torch.manual_seed(0)
    net = M.VANeRF(cfg).eval()
    net.load_state_dict(synth.make_hot_weights(0), strict=False)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    frame = synth.make_frame(seed=11, tar_h=256, tar_w=256, orbit_deg=15.0)
    S = 64

    def run_ref():
        with torch.no_grad():
            return M.VANeRF.batch_render_pifu_nerf(net, frame["img_in"], frame["cam_in"], frame["hand_type"], frame["targets"], 1, frame["cam_tar"], 3,
                                                   torch.tensor([[0.0, 0.0]]), None, frame["feat_geo"], frame["feat_tex"], None, copy.copy(frame["sp_data"]),
                                                   None, fine=True, uniform=True, sample_per_ray_c=S, sample_per_ray_f=S,
                                                   src_foreground_mask=frame["src_foreground_mask"], bounds=frame["bounds"], mask_at_box=None)

    def run_port():
        with torch.no_grad():
            return orc.batch_render(sd, frame, 3, torch.tensor([[[0, 0]]]), S, S)

    res = {}
    for name, fn in (("reference", run_ref), ("port", run_port)):
        fn()  # warm-up
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        res[name] = 4096 / min(ts)
        print(f"{name}: {min(ts):.1f} s per 64x64x(64+64) pass = {res[name]:.0f} rays/s on {torch.get_num_threads()} threads", flush=True)
    print(f"port / reference = {res['port'] / res['reference']:.2f}")


if __name__ == "__main__":
    main()
