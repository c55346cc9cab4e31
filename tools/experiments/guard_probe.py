"""Which guarded operand makes tf_gemm_fwd(256, 256, 768) come out wrong?  (round 6 diagnostic)"""
import math, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from guard_alloc import GuardPool
from transfusion_amd import _lib as L, ops

def bf(t): return t.to(torch.bfloat16)
dev = torch.device("cuda:0")
pool = GuardPool()
print("granularity", pool.gran)
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 256, 768))]
g = torch.Generator().manual_seed(M + N + K)
A0 = bf(torch.randn(M, K, generator=g)); W0 = bf(torch.randn(N, K, generator=g) / math.sqrt(K))
ref = (A0.float() @ W0.float().t())
for which in ("none", "A", "W", "C", "AW", "AWC"):
    A = pool.place(A0) if "A" in which else A0.to(dev)
    W = pool.place(W0) if "W" in which else W0.to(dev)
    C = pool.place(torch.zeros(M, N, dtype=torch.bfloat16)) if "C" in which else torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    torch.cuda.synchronize()
    ops.gemm(A, W, C, N, K, L.TF_EPI_NONE)
    torch.cuda.synchronize()
    c = C.float().cpu()
    bad = ((c - ref).abs() > 0.05 + 0.02 * ref.abs())
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print(which, "ptrs", hex(A.data_ptr()), hex(W.data_ptr()), hex(C.data_ptr()), "bad", int(bad.sum()),
          "rows", (int(rows.min()), int(rows.max())) if len(rows) else None, "cols", (int(cols.min()), int(cols.max())) if len(cols) else None,
          "zeros among bad", int((c[bad] == 0).sum()))
    # again: is it reproducible / does a second launch fix it
    ops.gemm(A, W, C, N, K, L.TF_EPI_NONE); torch.cuda.synchronize()
    c2 = C.float().cpu(); print("   second launch bad", int(((c2 - ref).abs() > 0.05 + 0.02 * ref.abs()).sum()))
pool.close()
