"""Guard-page device allocator for the kernel tests (TEST INFRASTRUCTURE, not product).

Why: round 5's driver run died with SIGABRT inside ``test_attention_fwd_bwd[False-2-64-1-96]``.  The cause was an over-READ of 4 - 8 bytes
past the end of the keep-bit image by ``attn_bwd_dkv16_kernel`` (waves whose keys all lie past S indexed a word the row does not have).
With operands from torch's caching allocator such a read faults only when the block happens to be the last one of a mapped segment:
placement-dependent, once in a dozen full runs.  Here every operand is placed FLUSH AGAINST THE END of a mapping made with HIP's virtual
memory API, and the virtual pages behind it are reserved but never mapped: a read or write one byte past the operand faults every time,
in the test that does it.  (``flush`` = up to the 16-byte alignment the C ABI requires of its pointers.)

    pool = GuardPool()                     # raises GuardUnavailable when the runtime lacks the VMM entry points
    t = pool.place(torch_tensor_on_cpu_or_gpu)      # -> a CUDA tensor of the same shape / dtype / contents in guarded memory
    pool.close()                           # after a device synchronise

The pool binds the HIP runtime that torch has already loaded (found in /proc/self/maps), so its mappings live in the same context.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch


class GuardUnavailable(RuntimeError):
    pass


class _Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class _AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class _Prop(C.Structure):          # hipMemAllocationProp (hip_runtime_api.h)
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", _Loc), ("win32HandleMetaData", C.c_void_p),
                ("allocFlags", _AllocFlags)]


class _Access(C.Structure):        # hipMemAccessDesc
    _fields_ = [("location", _Loc), ("flags", C.c_int)]


_PINNED, _DEVICE, _RW, _GRAN_MIN = 1, 1, 3, 0


def _loaded_hip_runtime() -> C.CDLL:
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.rsplit(" ", 1)[-1].strip()
            if "libamdhip64.so" in os.path.basename(path):
                return C.CDLL(path)
    raise GuardUnavailable("libamdhip64 is not mapped into this process (import torch and touch the GPU first)")


class _Array:
    """``__cuda_array_interface__`` view of raw device memory (torch.as_tensor takes it without copying)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class GuardPool:
    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise GuardUnavailable("no GPU")
        torch.cuda.init()
        torch.zeros(1, device=f"cuda:{device}")              # the context exists, the runtime is mapped
        self.hip = _loaded_hip_runtime()
        for name in ("hipMemAddressReserve", "hipMemCreate", "hipMemMap", "hipMemSetAccess", "hipMemUnmap", "hipMemRelease",
                     "hipMemAddressFree", "hipMemGetAllocationGranularity"):
            if not hasattr(self.hip, name):
                raise GuardUnavailable(f"{name} missing from the HIP runtime")
        self.device = device
        self.prop = _Prop(type=_PINNED, requestedHandleType=0, location=_Loc(_DEVICE, device), win32HandleMetaData=None,
                          allocFlags=_AllocFlags(0, 0, 0))
        gran = C.c_size_t(0)
        rc = self.hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(self.prop), _GRAN_MIN)
        if rc != 0 or gran.value == 0:
            raise GuardUnavailable(f"hipMemGetAllocationGranularity -> {rc}")
        self.gran = gran.value
        self.live = []                                       # (va, reserved bytes, mapped bytes, handle)
        self.tensors = []

    def _map(self, nbytes: int):
        mapped = max(1, math.ceil(nbytes / self.gran)) * self.gran
        reserved = mapped + self.gran                        # one granule of reserved, never mapped, address space behind the data
        va = C.c_void_p(0)
        rc = self.hip.hipMemAddressReserve(C.byref(va), C.c_size_t(reserved), C.c_size_t(self.gran), None, C.c_ulonglong(0))
        if rc != 0:
            raise GuardUnavailable(f"hipMemAddressReserve({reserved}) -> {rc}")
        handle = C.c_void_p(0)
        rc = self.hip.hipMemCreate(C.byref(handle), C.c_size_t(mapped), C.byref(self.prop), C.c_ulonglong(0))
        if rc != 0:
            self.hip.hipMemAddressFree(va, C.c_size_t(reserved))
            raise GuardUnavailable(f"hipMemCreate({mapped}) -> {rc}")
        rc = self.hip.hipMemMap(va, C.c_size_t(mapped), C.c_size_t(0), handle, C.c_ulonglong(0))
        if rc == 0:
            acc = _Access(_Loc(_DEVICE, self.device), _RW)
            rc = self.hip.hipMemSetAccess(va, C.c_size_t(mapped), C.byref(acc), C.c_size_t(1))
        if rc != 0:
            self.hip.hipMemRelease(handle)
            self.hip.hipMemAddressFree(va, C.c_size_t(reserved))
            raise GuardUnavailable(f"hipMemMap / hipMemSetAccess -> {rc}")
        self.live.append((va.value, reserved, mapped, handle))
        return va.value, mapped

    def place(self, t: torch.Tensor | None, align: int = 16) -> torch.Tensor | None:
        """A copy of ``t`` whose LAST byte is the last mapped byte (rounded down to ``align`` for the first byte)."""
        if t is None:
            return None
        src = t.detach().contiguous()
        nbytes = src.numel() * src.element_size()
        if nbytes == 0:
            return src.to(f"cuda:{self.device}")
        va, mapped = self._map(nbytes)
        start = va + ((mapped - nbytes) // align) * align
        raw = torch.as_tensor(_Array(start, nbytes), device=f"cuda:{self.device}")
        assert raw.data_ptr() == start and raw.numel() == nbytes
        out = raw.view(src.dtype).view(src.shape)
        out.copy_(src.to(out.device))
        self.tensors.append(raw)
        return out

    def end_gap(self, t: torch.Tensor) -> int:
        """Bytes between the end of ``t`` and the first unmapped byte (0 .. align - 1)."""
        end = t.data_ptr() + t.numel() * t.element_size()
        for va, _, mapped, _ in self.live:
            if va <= t.data_ptr() < va + mapped:
                return va + mapped - end
        raise KeyError("not a guarded tensor")

    def close(self, free_va: bool = False):
        """Give the physical memory back.  The address ranges stay RESERVED by default and are never handed out again in this process: on
        this runtime a range that was unmapped, freed and reserved again at the same address served stale translations to some compute
        units (a GEMM's output tile landed in the old pages: tools/experiments/guard_probe.py) -- a fresh range per operand cannot alias."""
        torch.cuda.synchronize()
        self.tensors.clear()
        for va, reserved, mapped, handle in self.live:
            self.hip.hipMemUnmap(C.c_void_p(va), C.c_size_t(mapped))
            self.hip.hipMemRelease(handle)
            if free_va:
                self.hip.hipMemAddressFree(C.c_void_p(va), C.c_size_t(reserved))
        self.live.clear()
