#!/usr/bin/env python3
"""One-rank RCCL rehearsal (phantom peer) of tests/test_gpu_ddp.py's wrapper worker with the streams of every K1 / K9 backward and of
every post-accumulate hook logged.  usage: debug_tree2.py <out.pt> ; env TF_TEST_WGRAD_DELAY_US, LEVELS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TF_REHEARSE_COLLECTIVES="1", TF_REHEARSE_PHANTOM_PEERS="1")
import torch
import test_gpu_ddp as T
from transfusion_amd import ops
from transfusion_amd.runner import trainer

LOG = os.environ.get("DBG_LOG") == "1"
sid = lambda: hex(torch.cuda.current_stream().cuda_stream)
orig_bwd = ops._LinearFn.backward
def bwd(ctx, gy):
    if LOG: print(f"  _LinearFn.backward N={ctx.meta[2]} K={ctx.meta[1]} on stream {sid()}", flush=True)
    return orig_bwd(ctx, gy)
ops._LinearFn.backward = staticmethod(bwd)
orig_hook = trainer.OrderedRangeReducer._make_param_hook
def mk(self, u):
    h = orig_hook(self, u)
    def hook(param):
        if LOG: print(f"  post-accumulate hook unit {self.units[u]['key']} shape {tuple(param.shape)} on stream {sid()} pending {self._pending.get(u)}", flush=True)
        return h(param)
    return hook
trainer.OrderedRangeReducer._make_param_hook = mk
orig_complete = trainer.OrderedRangeReducer._complete
def comp(self, u):
    if LOG: print(f"  complete unit {self.units[u]['key']} on stream {sid()}", flush=True)
    return orig_complete(self, u)
trainer.OrderedRangeReducer._complete = comp
levels = getattr(T, os.environ.get("LEVELS", "_LEVELS_RAGGED"))
code = T._TREE_WORKER.format(root=ROOT, out=sys.argv[1], levels=levels)
exec(compile(code, "worker", "exec"))
