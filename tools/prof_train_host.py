"""Host side of a training step (tools/perf_train_step.py's step): cProfile over five steps, by own time and by cumulative time -- what the
interpreter and the runtime calls cost between the launches."""
import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import perf_train_step as P
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    P.step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
