"""Calibration only (not on the product path): the training step's GEMM shapes through torch.mm (hipBLASLt / rocBLAS on ROCm), back to
back, to see what the vendor library reaches on the same MI355X for the shapes our hand-written kernels serve.
usage (GPU box): python tools/lib_gemm_calibration.py [rows]"""
import sys, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16600
dev = torch.device("cuda:0")
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f"rows M = {M}")
for name, N, K in (("qkv fwd", 2304, 768), ("ffn_up fwd", 1536, 768), ("ffn_down fwd", 768, 1536), ("out_proj fwd", 768, 768), ("in_proj dgrad", 768, 2304)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.mm(a, w.t()))
    print(f"  NT  {name:14s} [{M} x {K}] x [{N} x {K}]^T : {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
tot = 0.0
for name, N, K in (("in_proj", 2304, 768), ("out_proj", 768, 768), ("linear1", 1536, 768), ("linear2", 768, 1536)):
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16); x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.mm(dy.t(), x))
    tot += us
    print(f"  TN  wgrad {name:9s} [{M} x {N}]^T x [{M} x {K}] : {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
print(f"  a layer's four weight gradients: {tot:.1f} us  {2.0 * M * 768 * (2304 + 768 + 1536 + 1536) / tot / 1e6:.1f} TFLOP/s (bf16 output, no bias sums, no accumulate)")
for n in (4096, 8192):
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16); b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.mm(a, b.t()), 10)
    print(f"  NT  square {n}: {us:8.1f} us  {2.0 * n ** 3 / us / 1e6:7.1f} TFLOP/s")
