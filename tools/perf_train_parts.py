"""Where a training step's wall time goes (tools/perf_train_step.py's step cut at synchronisation points): forward / backward / Adam, and inside the
backward the fused HIP stage (forward spill + chain + products + scatters) alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import perf_train_step as P
net, frame, dr, opt = P.net, P.frame, P.dr, P.opt
def part(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t)
rows = []
for _ in range(5):
    out, t_f = part(lambda: net(frame["img_in"], frame["cam_in"], frame["hand_type"], frame["targets"], None, None, n_views=1, sp_data=dict(frame["sp_data"]),
                                dr_data=dr, src_foreground_mask=frame["src_foreground_mask"], bounds=frame["bounds"])["out"]["nerf"])
    loss = (out["tex_fg_fine"] - out["tar_img"]).abs().mean() + (out["tex_fg"] - out["tar_img"]).abs().mean() + 0.1 * out["alpha_fine"].mean()
    opt.zero_grad(set_to_none=True)
    _, t_b = part(lambda: loss.backward())
    _, t_o = part(lambda: opt.step())
    rows.append((t_f, t_b, t_o))
print("forward / backward / Adam (ms, synchronised, min over 5):", [round(min(r[i] for r in rows), 2) for i in range(3)])
from vanerf_amd import hip_backward as HB, torch_graph as G
import cProfile, pstats
out = net(frame["img_in"], frame["cam_in"], frame["hand_type"], frame["targets"], None, None, n_views=1, sp_data=dict(frame["sp_data"]),
          dr_data=dr, src_foreground_mask=frame["src_foreground_mask"], bounds=frame["bounds"])["out"]["nerf"]
loss = (out["tex_fg_fine"] - out["tar_img"]).abs().mean() + (out["tex_fg"] - out["tar_img"]).abs().mean() + 0.1 * out["alpha_fine"].mean()
opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); loss.backward(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
