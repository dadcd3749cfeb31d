"""Time of one training step of the drop-in module on one GPU: HIP forward of the 64x64 patch (values) + PyTorch graph at the same samples
(gradients) + backward (BASELINE config 5, first stage of SURVEY.md section 8 row f-4)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import synth
from vanerf_amd.config import default_config
from vanerf_amd.model import VANeRF
torch.manual_seed(0)
import numpy as np
np.random.seed(0)  # the training window is drawn with numpy (src/model.py:1172-1189): the same patches in every run
cfg = default_config()
for key in ("grad_rays_per_chunk", "grad_samples_per_block"):  # e.g. --grad_samples_per_block 131072
    if "--" + key in sys.argv:
        cfg["models"]["VANeRF"][key] = int(sys.argv[sys.argv.index("--" + key) + 1])
if "--grad_graph_blocks" in sys.argv:  # the blocks of the backward's second stage as replays of one HIP graph
    cfg["models"]["VANeRF"]["grad_graph_blocks"] = True
if "--torch_graph" in sys.argv:  # the PyTorch graph instead of the fused HIP backward (the independent checker)
    cfg["models"]["VANeRF"]["hip_backward"] = False
if "--hip_backward_block" in sys.argv:
    cfg["models"]["VANeRF"]["hip_backward_block"] = int(sys.argv[sys.argv.index("--hip_backward_block") + 1])
if "--graph_encoders" in sys.argv:  # the two image encoders as HIP graphs (forward and backward)
    cfg["models"]["VANeRF"]["graph_encoders"] = True
# --cudnn_benchmark: the reference's trainer runs with `benchmark=True` (train.py:60: PyTorch Lightning sets torch.backends.cudnn.benchmark), i.e. MIOpen
# searches the convolution algorithms of the two image encoders during the first steps.  Measured: 28.1 / 28.8 ms (min / median) against 28.5 / 29.5 without,
# for minutes of search at start-up -- not the default here.
torch.backends.cudnn.benchmark = "--cudnn_benchmark" in sys.argv
net = VANeRF(cfg).cuda().train()
net.load_state_dict(synth.make_full_weights(0), strict=False)
if "--channels_last" in sys.argv:  # experiment: the two image encoders' convolutions in NHWC
    for enc in (net.geo_encoder, net.tex_encoder):
        if enc is not None:
            enc.to(memory_format=torch.channels_last)
frame = synth.to_device(synth.make_frame(seed=3, tar_h=256, tar_w=256), "cuda")
dr = {"img": frame["img_in"], "cam": frame["cam_in"], "cam_tar": frame["cam_tar"], "tar": torch.rand(1, 3, 256, 256, device="cuda"),
      "msk": torch.ones(1, 1, 256, 256, device="cuda")}
opt = torch.optim.Adam(net.parameters(), lr=1e-5, fused=("--fused_adam" in sys.argv) or None)  # (--fused_adam: one launch per dtype group; default: PyTorch's foreach path)
PIPELINED = "--pipelined" in sys.argv  # no read of the loss per step: the host runs ahead of the GPU across the step boundary (a training loop
#                                       that logs every few steps); the time reported is then the average of 10 back-to-back steps
def step():
    out = net(frame["img_in"], frame["cam_in"], frame["hand_type"], frame["targets"], None, None, n_views=1, sp_data=dict(frame["sp_data"]),
              dr_data=dr, src_foreground_mask=frame["src_foreground_mask"], bounds=frame["bounds"])["out"]["nerf"]
    loss = (out["tex_fg_fine"] - out["tar_img"]).abs().mean() + (out["tex_fg"] - out["tar_img"]).abs().mean() + 0.1 * out["alpha_fine"].mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss.detach() if PIPELINED else float(loss.detach())
for _ in range(4 if torch.backends.cudnn.benchmark else 2):
    step()
torch.cuda.synchronize()
ts = []
if PIPELINED:
    t0 = time.perf_counter()
    for _ in range(10):
        l = step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10); l = float(l)
else:
    for _ in range(10):
        t0 = time.perf_counter(); l = step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
with torch.no_grad():
    net.eval(); torch.cuda.synchronize(); t0 = time.perf_counter()
    net.train()
print(f"training step (64x64 patch, 64+64 samples, encoders + HIP forward + backward + Adam): min {1e3 * min(ts):.1f} ms, median {1e3 * sorted(ts)[len(ts) // 2]:.1f} ms, loss {l:.4f}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
