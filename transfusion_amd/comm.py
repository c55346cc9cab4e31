"""Host side of the C ABI's gradient exchange (include/tfusion.h: tf_comm_*, tf_allreduce_bucket): one RCCL communicator per
process, created from a unique id that rank 0 draws and torch.distributed (any backend: it only carries 128 bytes)
broadcasts.  Stands where Lightning's strategy="ddp" stands in the reference (runner/run_experiment.py:452).

Selected with ``TF_COMM=rccl`` (or ``FusionTrainStep(comm="rccl")``); the default exchange stays torch.distributed's
process group, which on ROCm is the same RCCL underneath.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from transfusion_amd import _lib as L

ID_BYTES = L.CONSTS["TF_COMM_ID_BYTES"]


class BucketComm:
    """``all_reduce_(t, stream)``: in-place SUM over ranks of a contiguous fp32 CUDA tensor, enqueued on ``stream``."""

    def __init__(self, world: int, rank: int, unique_id: bytes, device: torch.device):
        if len(unique_id) != ID_BYTES:
            raise ValueError(f"unique id must be {ID_BYTES} bytes")
        self.lib = L.load()
        self.lib.tf_allreduce_bucket.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        self.lib.tf_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        self.lib.tf_comm_stats.argtypes = [C.c_void_p] * 5
        self.lib.tf_comm_destroy.argtypes = [C.c_void_p]
        self.world, self.rank, self.device = world, rank, torch.device(device)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):                       # the communicator binds the calling thread's current device
            L.check(self.lib.tf_comm_create(C.byref(self.handle), unique_id, world, rank), "tf_comm_create")

    @staticmethod
    def new_unique_id() -> bytes:
        buf = C.create_string_buffer(ID_BYTES)
        L.check(L.load().tf_comm_unique_id(buf), "tf_comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, device, group=None) -> "BucketComm":
        """Collective over ``group``: every rank must call it."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        return cls(world, rank, box[0], device)

    def all_reduce_(self, t: torch.Tensor, stream=None):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise L.TfError("tf_allreduce_bucket needs a contiguous fp32 CUDA tensor")
        if t.device != self.device:
            raise L.TfError(f"communicator lives on {self.device}, tensor on {t.device}")
        stream = stream if stream is not None else torch.cuda.current_stream(t.device)
        L.check(self.lib.tf_allreduce_bucket(self.handle, t.data_ptr(), t.numel(), stream.cuda_stream), "tf_allreduce_bucket")
        # the collective reads and writes the raw pointer later, on `stream`: tell the caching allocator, so that the block is not
        # handed to another tensor before that work has run (a view of a long-lived flat buffer today; a temporary tomorrow)
        t.record_stream(stream)

    def stats(self):
        w, r, calls, elems = C.c_int(), C.c_int(), C.c_longlong(), C.c_longlong()
        L.check(self.lib.tf_comm_stats(self.handle, C.byref(w), C.byref(r), C.byref(calls), C.byref(elems)), "tf_comm_stats")
        return {"world": w.value, "rank": r.value, "calls": calls.value, "elems": elems.value}

    def close(self):
        if self.handle:
            torch.cuda.synchronize(self.device)
            L.check(self.lib.tf_comm_destroy(self.handle), "tf_comm_destroy")
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
