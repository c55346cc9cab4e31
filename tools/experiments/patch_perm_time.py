"""K1 / K9 permutation kernels at the reference's FPN level shapes (batch 4): microseconds and GB/s per call (round 6).
Before (round 5, rocprofv3 inside the wrapper step): col2im 9.6 .. 96 us, im2col_tiled 16.5 .. 64 us per level."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from transfusion_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 4))
for C, H, p in ((256, 112, 4), (512, 56, 4), (1024, 28, 2), (2048, 14, 1)):
    feat = torch.randn(B, C, H, H, device=dev)
    rows = ops.patchify(feat, p, p)
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    nb = feat.numel() * 4 + rows.numel() * 2
    tg = t(lambda: ops.patchify(feat, p, p))
    ts = t(lambda: ops.regroup(rows, H, H, p, p, out_dtype=torch.float32))
    print(f"C={C:5d} {H}x{H} p={p}: gather {tg:7.1f} us ({nb / tg / 1e3:7.1f} GB/s)   scatter {ts:7.1f} us ({nb / ts / 1e3:7.1f} GB/s)   [{nb / 1e6:.1f} MB] (times include the output allocation)")
