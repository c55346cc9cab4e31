#!/usr/bin/env python3
"""Where the HOST time of the backward goes at B = 4 (the autograd engine runs it in its own thread, out of cProfile's sight): wall time
of _EncoderFn.backward, of the native tf_encoder_bwd call inside it, of _bind_grads, and of the whole loss.backward() call."""
import importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from transfusion_amd import _lib as L
from transfusion_amd.modeling.cross_fusion.ego_fusion import cross_f_box_layers as M
from transfusion_amd.runner.trainer import FusionTrainStep
dev = torch.device("cuda", 0)
Dm = int(os.environ.get("D", 768))
enc = b.make_encoder(dev, d=Dm); enc.train()
tr = FusionTrainStep(enc, lr=1e-4, weight_decay=2e-4, grad_clip=1.0)
batches = [b.make_batch(int(os.environ.get("B", 4)), dev, 0, d=Dm, variant=v) for v in range(2)]
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
orig_call = L.call
def call(fn_name, *a):
    t0 = time.perf_counter(); r = orig_call(fn_name, *a); acc["native " + fn_name] = acc.get("native " + fn_name, 0.0) + time.perf_counter() - t0; return r
L.call = call; M.L.call = call
M._EncoderFn.backward = staticmethod(timed("_EncoderFn.backward", M._EncoderFn.backward))
M._EncoderFn.forward = staticmethod(timed("_EncoderFn.forward", M._EncoderFn.forward))
enc._bind_grads = timed("_bind_grads", enc._bind_grads)
enc._make_desc = timed("_make_desc", enc._make_desc)
from transfusion_amd import ops as _ops
_ops._SqLossFn.backward = staticmethod(timed("loss.backward fn", _ops._SqLossFn.backward))
_ops._SqLossFn.forward = staticmethod(timed("loss.forward fn", _ops._SqLossFn.forward))
b.loss_fn = timed("loss_fn (module forward + loss)", b.loss_fn)
tr.zero_grad = timed("zero_grad", tr.zero_grad)
tr.mark_parameters_updated = timed("mark_parameters_updated", tr.mark_parameters_updated)
tr.check_errors = timed("check_errors", tr.check_errors)
orig_bw = torch.Tensor.backward
torch.Tensor.backward = timed("Tensor.backward()", orig_bw)
tr.opt.step = timed("opt.step", tr.opt.step)
tr.opt.grad_sumsq = timed("opt.grad_sumsq", tr.opt.grad_sumsq)
for i in range(6): tr.step([batches[i % 2]], b.loss_fn)
torch.cuda.synchronize()
acc.clear()
n = 30
t0 = time.perf_counter()
for i in range(n): tr.step([batches[i % 2]], b.loss_fn)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:32s} {1e6 * v / n:8.1f} us/step")
