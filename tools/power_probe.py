"""Samples rocm-smi power / clocks while query_kernel runs back to back (evidence for the sustained clock under bf16 MFMA load)."""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanerf_amd import renderer as R, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
sd = synth.make_full_weights(0)
frame = synth.make_frame(seed=11, tar_h=512, tar_w=334, orbit_deg=15.0)
fd = synth.to_device(frame, "cuda")
sdd = {k: v.cuda() for k, v in sd.items() if k.startswith("tex_vis_fusion.")}
fdat = R.FrameData(sdd, fd["img_in"], fd["feat_geo"], fd["feat_tex"], fd["src_foreground_mask"], fd["cam_in"], fd["targets"], fd["sp_data"])
rays = R.ray_setup(frame["cam_tar"], frame["bounds"], 0, 0, 1, 334, 512, 64, device="cuda")
pts = R.sample_points(rays["rays_d"], rays["cam_pos"], rays["z"])
q_sdf, q_vis, knn = R.mesh_query_accel(fdat.accel, fdat.verts3, fdat.faces, fdat.vert_vis, pts)
w = R.PackedWeights(sd, mode=mode)
stop = False
samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.split(":", 1)[1].strip() for l in out.splitlines() if ("sclk" in l or "Power (W)" in l or "Temperature (Sensor junction)" in l)]
            samples.append(keep)
        except Exception as e:  # noqa: BLE001
            samples.append([repr(e)])
        time.sleep(0.4)
t = threading.Thread(target=sampler); t.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(20):
        R.query_samples(w, fdat, pts, q_sdf, q_vis, knn)
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
stop = True; t.join()
print(f"mode {mode}: {n} launches in {dt:.2f} s = {1e3 * dt / n:.2f} ms per launch back to back")
for s_ in samples[2:14]:
    print("  ", s_)
