#!/usr/bin/env python3
"""cProfile of the HOST side of the wrapper step (B from the environment): where the Python time of one fwd + bwd goes."""
import cProfile, io, os, pstats, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ns = runpy.run_path(os.path.join(ROOT, "tools", "wrapper_time.py"), run_name="wrapper_time_import")
import torch
step = ns["step"]
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:7000])
