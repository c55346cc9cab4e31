#!/usr/bin/env python3
"""torch.profiler (CPU side) of the wrapper step: which operators / autograd nodes the HOST time of forward + backward is spent in
(the backward runs on the autograd engine's thread, which cProfile does not see)."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ns = runpy.run_path(os.path.join(ROOT, "tools", "wrapper_time.py"), run_name="wrapper_time_import")
import torch
from torch.profiler import profile, ProfilerActivity
step = ns["step"]
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
