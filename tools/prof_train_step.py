"""torch.profiler view of one training step (tools/perf_train_step.py): device time per operator, to see what the 178 ms are."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
import perf_train_step as P  # runs the warm-up and the timed steps, leaves `step` ready
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    P.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=35, max_name_column_width=60))
