#!/usr/bin/env python3
"""Which framework (aten) operators put kernels / device copies into the wrapper's training step, with the Python line that called
them: torch.profiler (CPU + device activities, stacks) over the step of tools/wrapper_time.py.  Usage (GPU box):
    B=4 REAL=1 TRAINER=1 python tools/wrapper_gpu_ops.py"""
import collections, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ns = runpy.run_path(os.path.join(ROOT, "tools", "wrapper_time.py"), run_name="wrapper_time_import")
import torch
from torch.profiler import profile, ProfilerActivity
step = ns["step"]
for _ in range(4):
    step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))
# aten ops that launch device work, by the innermost repo frame of their stack
by_site = collections.Counter()
t_site = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "/repo/" in fr and "wrapper_gpu_ops" not in fr:
            site = fr.split("/repo/")[-1]
            break
    by_site[(ev.name, site)] += 1
    t_site[(ev.name, site)] += ev.self_device_time_total
print(f"\naten operators with device time, per step ({N} steps), by calling line:")
for (name, site), t in sorted(t_site.items(), key=lambda kv: -kv[1])[:60]:
    print(f"  {t / N:8.1f} us  x{by_site[(name, site)] / N:5.1f}  {name:28s} {site}")
