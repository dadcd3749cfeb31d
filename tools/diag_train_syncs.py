"""Blocking calls inside a training step (tools/perf_train_step.py's step): torch's sync debug mode warns at every one PyTorch makes
(.item(), copies to the host, nonzero, ...), with the Python stack of the first occurrence of each."""
import os, sys, traceback, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import perf_train_step as P
seen = {}
def show(message, category, filename, lineno, file=None, line=None):
    stack = [f for f in traceback.extract_stack()[:-1] if "/vanerf_amd/" in f.filename or "/tools/" in f.filename]
    key = tuple((f.filename, f.lineno) for f in stack[-3:])
    if key not in seen:
        seen[key] = 0
        print("SYNC:", str(message)[:100])
        for f in stack[-4:]:
            print(f"    {os.path.basename(f.filename)}:{f.lineno} {f.name}: {f.line}")
    seen[key] += 1
warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
P.step()
torch.cuda.set_sync_debug_mode("default")
print("blocking calls per step:", sum(seen.values()), "at", len(seen), "places")
