"""Data-parallel training step for the fusion block: the counterpart of what PyTorch-Lightning's
``strategy="ddp"`` + ``EgoNAOTrainer`` do around the model in the reference (run_experiment.py:437-454;
ego_nao_trainer.py:259-380; optimiser groups abc_nao_trainer.py:179-201), reduced to the hot path:

    forward -> backward (gradients land directly in ONE flat fp32 buffer) -> bucketed all-reduce of that
    buffer over RCCL (torch.distributed backend "nccl") / gloo on CPU -> global-norm clip -> fused RAdam.

One process per GPU.  Parameters never receiving a gradient (``heatmap_token``, frozen tensors) are excluded
statically, which replaces Lightning's ``find_unused_parameters=True``.
Gradient accumulation (``accumulate_grad_batches``) reduces only on the last micro-batch.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import os

import torch
import torch.distributed as dist
from torch import nn


def _rehearsal_switches(world: int):
    """(collectives issued in a one-rank group?, phantom peers).  The two TEST-ONLY environment switches of the one-GPU rehearsal, read
    ONCE -- when a reducer is constructed -- and kept as attributes (like ``_debug_break_edge``): the step's hot path reads no environment,
    and a variable leaked into a real job's environment cannot scale gradients: phantom peers are refused when there are real ones."""
    rehearse = os.environ.get("TF_REHEARSE_COLLECTIVES") == "1"
    k = int(os.environ.get("TF_REHEARSE_PHANTOM_PEERS", "0")) if rehearse else 0
    if k and world > 1:
        raise RuntimeError("TF_REHEARSE_PHANTOM_PEERS is a one-rank rehearsal switch: refused with %d real ranks" % world)
    return rehearse, k


def _live_world(group=None):
    """(world size, whether the collectives are issued).  They are issued whenever more than one rank takes part -- and, for a REHEARSAL
    of the RCCL path on one GPU, also in a one-rank process group when TF_REHEARSE_COLLECTIVES=1: every all-reduce, event and stream of the
    N > 1 step then runs against the real backend (bench.py's rehearsal entry; tests/test_gpu_ddp.py), only without peers.  Called at
    construction time only (see _rehearsal_switches)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1, False
    world = dist.get_world_size(group)
    return world, world > 1 or _rehearsal_switches(world)[0]


def _phantom_peers(k, g, handle=None):
    """Rehearsal only (k = the reducer's ``_phantom_k``: TF_REHEARSE_PHANTOM_PEERS with TF_REHEARSE_COLLECTIVES=1, resolved when the
    reducer was built): a one-rank sum all-reduce leaves the buffer as it was, so a collective that ran BEFORE its gradients were written
    (a missing event edge) would go unnoticed.  With k phantom peers the reduced range is multiplied by 1 + k right behind the collective,
    on the collective's stream -- as if k peers had contributed the same gradient: whatever is written into the range after the collective
    misses the factor, and tests/test_gpu_ddp.py sees it."""
    if not k:
        return handle
    if handle is not None:
        handle.wait()                       # (RCCL: the CURRENT stream waits for the collective; the host does not)
    g.mul_(float(1 + k))
    return None


class FlatParams:
    """Re-homes a module's trainable parameters (and their gradients) as views into two flat fp32 buffers."""

    def __init__(self, module: nn.Module, exclude: Iterable[str] = ("heatmap_token", "class_token")):
        # heatmap_token / class_token are created by the reference constructor but never used by its forward
        # (cross_f_box_layers.py:38-43): their grad stays None there and RAdam skips them (no weight decay either)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and not any(n.endswith(e) for e in exclude)]
        if not named:
            raise ValueError("no trainable parameters")
        dev = named[0][1].device
        total = 0
        self.slices = []
        for n, p in named:
            size = (p.numel() + 63) // 64 * 64          # 256-B aligned starts
            self.slices.append((n, p, total, p.numel()))
            total += size
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for n, p, off, num in self.slices:
            self.flat[off:off + num].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + num].view_as(p)
            p.grad = self.grad[off:off + num].view_as(p)
        self.names = [n for n, _, _, _ in self.slices]
        self.numel = sum(num for _, _, _, num in self.slices)

    def check_bound(self):
        """The parameters' ``.grad`` must still be the views into ``self.grad`` (``module.zero_grad(set_to_none=True)`` or an
        optimiser that replaces ``p.grad`` silently detaches them: the all-reduce and the fused optimiser would then see zeros)."""
        base = self.grad.data_ptr()
        for n, p, off, num in self.slices:
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                raise RuntimeError(f"gradient of {n} is no longer a view of the flat gradient buffer (set_to_none / replaced .grad?)")


class DataParallelReducer:
    """Sum all-reduce of the flat gradient buffer in a few large buckets (xGMI rings are per-link bound: fewer,
    larger messages).  Works with any initialised torch.distributed backend; a no-op at world size 1."""

    def __init__(self, flat_grad: torch.Tensor, bucket_mb: float = 64.0, group=None, bucket_comm=None):
        self.grad = flat_grad
        self.group = group
        self.bucket_comm = bucket_comm          # transfusion_amd.comm.BucketComm: the C ABI's tf_allreduce_bucket instead of torch's
        self.world, self.live = _live_world(group)
        self._phantom_k = _rehearsal_switches(self.world)[1]      # test-only, resolved once
        n = flat_grad.numel()
        per = max(1, int(bucket_mb * (1 << 20) // 4))
        self.buckets = [(s, min(n, s + per)) for s in range(0, n, per)]
        self.bytes_per_step = n * 4

    def all_reduce(self, async_op: bool = False):
        if not self.live:
            return []
        handles = []
        if self.bucket_comm is not None:              # stream-ordered: nothing to wait for on the host
            for s, e in self.buckets:
                self.bucket_comm.all_reduce_(self.grad[s:e])
                _phantom_peers(self._phantom_k, self.grad[s:e])
            return handles
        for s, e in self.buckets:
            h = _phantom_peers(self._phantom_k, self.grad[s:e], dist.all_reduce(self.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
            if async_op and h is not None:
                handles.append(h)
        return handles


class LayerwiseReducer:
    """All-reduce that starts as soon as a layer's gradients exist: the encoder runtime is called once per layer
    (top layer first) and after each call the contiguous slice of the flat gradient buffer that belongs to that layer
    (plus final-norm with the top layer, kind embeddings with layer 0) is reduced asynchronously; torch's RCCL process
    group orders the collective after the kernels already enqueued and runs it on its own stream, so it overlaps the
    remaining backward.  ``finish()`` joins before the optimiser."""

    def __init__(self, flat: FlatParams, group=None, bucket_comm=None):
        self.flat, self.group = flat, group
        self.bucket_comm = bucket_comm          # transfusion_amd.comm.BucketComm or None (torch.distributed's process group)
        self.world, self.live = _live_world(group)
        self._phantom_k = _rehearsal_switches(self.world)[1]      # test-only, resolved once
        self.bytes_per_step = flat.grad.numel() * 4
        self.handles = []
        self.comm = None
        self.ranges = {}
        ends = {}
        for name, _, off, num in flat.slices:
            m = [tok for tok in name.split(".")]
            layer = int(m[m.index("layers") + 1]) if "layers" in m else None
            key = layer if layer is not None else ("tail" if "final_norm" in name else "head")
            lo, hi = ends.get(key, (off, off + num))
            ends[key] = (min(lo, off), max(hi, off + num))
        self.num_layers = 1 + max(k for k in ends if isinstance(k, int))
        for layer in range(self.num_layers):
            lo, hi = ends[layer]
            if layer == self.num_layers - 1 and "tail" in ends:
                lo, hi = min(lo, ends["tail"][0]), max(hi, ends["tail"][1])
            if layer == 0 and "head" in ends:
                lo, hi = min(lo, ends["head"][0]), max(hi, ends["head"][1])
            self.ranges[layer] = (lo, hi)
        # the ranges must tile the flat buffer exactly (alignment padding between parameters belongs to the range before it; it holds
        # zeros): nothing reduced twice, nothing forgotten, no gap
        order = sorted(self.ranges, key=lambda l: self.ranges[l][0])
        total = flat.grad.numel()
        for a, b in zip(order, order[1:] + [None]):
            nxt = total if b is None else self.ranges[b][0]
            if self.ranges[a][1] > nxt:
                raise ValueError(f"layer gradient ranges overlap: {self.ranges}")
            self.ranges[a] = (self.ranges[a][0], nxt)
        if self.ranges[order[0]][0] != 0:
            raise ValueError(f"layer gradient ranges do not start at 0: {self.ranges}")
        self.active = True            # False: the hook does nothing (micro-batches of an accumulation window before the last one)
        self.collectives = 0          # all-reduces issued so far (every rank must issue the same number: tests assert it)

    joins_overlap = True      # finish() joins the encoder runtime's side stream: the per-layer backward calls need not
    _debug_break_edge = None  # tests only ("side" / "main"): drop one event edge, the negative control of tests/test_gpu_ddp.py's RCCL rehearsal

    def hook(self, module, layer):
        """Called right after layer `layer`'s backward has been ENQUEUED.  Its weight gradients are produced partly on the
        chain's stream and partly on the runtime's side stream, which the per-layer calls do not join (a join stalls the chain
        until that layer's last wgrad has finished: -2.5 % on one GPU).  The collective therefore runs behind a communication
        stream that waits for an event on each of the two."""
        if not self.live or not self.active:
            return
        from transfusion_amd import ops
        lo, hi = self.ranges[layer]
        self.collectives += 1
        g = self.flat.grad
        main = torch.cuda.current_stream(g.device) if g.is_cuda else None
        side = ops.side_stream(g.device) if g.is_cuda else None
        if side is None:
            self.handles.append(dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=g.device)
        if self._debug_break_edge != "main":
            self.comm.wait_event(main.record_event())
        if self._debug_break_edge != "side":
            self.comm.wait_event(side.record_event())
        with torch.cuda.stream(self.comm):
            if self.bucket_comm is not None:
                self.bucket_comm.all_reduce_(g[lo:hi], stream=self.comm)     # tf_allreduce_bucket, enqueued on the communication stream
                _phantom_peers(self._phantom_k, g[lo:hi])
                return
            h = _phantom_peers(self._phantom_k, g[lo:hi], dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            if h is not None:
                self.handles.append(h)

    def finish(self):
        from transfusion_amd import ops
        if self.flat.grad.is_cuda:
            ops.join_overlap(self.flat.grad.device)          # the optimiser's stream waits for the side stream ...
        for h in self.handles:
            h.wait()                                         # ... and for the collectives
        if self.comm is not None:
            torch.cuda.current_stream(self.flat.grad.device).wait_stream(self.comm)
        self.handles = []


class OrderedRangeReducer:
    """Gradient all-reduce overlapped with the backward for ANY module tree -- the fusion wrapper's four encoders with their patch
    embedding (K1) and back-projection (K9) GEMMs, an encoder with RoI heads, ... -- where ``LayerwiseReducer`` knows one bare encoder.

    The flat gradient buffer is cut into contiguous UNITS: one per encoder layer of every module that offers a ``layer_grad_hook``
    (final norm with its top layer, kind embeddings with layer 0), and one per remaining sub-module (``patches_to_token.2``,
    ``tokens_to_features.0``, ``heads.box_regressor`` ...), whose parameters receive their gradients through autograd: a
    post-accumulate hook per parameter counts the unit down.  A unit that is complete is all-reduced asynchronously behind a
    communication stream that waits for the streams its gradients were produced on.

    Every rank must issue the same collectives in the same order, whatever order its own backward happens to finish things in, so
    units are FIRED IN A FIXED ORDER: the order in which they completed during the first optimiser step (which itself reduces the
    whole buffer after its backward, un-overlapped), checked to be identical on all ranks; later steps fire unit k as soon as it and
    all units before it are complete.  Units that never complete (a branch without gradient this step) are flushed by ``finish()``.
    What the reference gets from torch DDP's bucket order (runner/run_experiment.py:444-446, 452)."""

    joins_overlap = True
    _debug_break_edge = None      # tests only ("side" / "accum"): drop one event edge, the negative control of tests/test_gpu_ddp.py's delay probe

    def __init__(self, flat: FlatParams, module: nn.Module, group=None, bucket_comm=None):
        self.flat, self.group, self.bucket_comm = flat, group, bucket_comm
        self.world, self.live = _live_world(group)
        self._phantom_k = _rehearsal_switches(self.world)[1]      # test-only, resolved once
        self.bytes_per_step = flat.grad.numel() * 4
        total = flat.grad.numel()
        slice_of = {n: (off, num) for n, _, off, num in flat.slices}
        params = {n: p for n, p, _, _ in flat.slices}
        claimed = {}
        self.units = []                      # {"key": str, "lo": int, "hi": int}
        self._enc_units = {}                 # id(encoder module) -> {layer: unit index}
        for prefix, m in module.named_modules():
            if not hasattr(m, "layer_grad_hook"):
                continue
            pre = prefix + "." if prefix else ""
            ends = {}
            for n in slice_of:
                if not n.startswith(pre):
                    continue
                toks = n[len(pre):].split(".")
                layer = int(toks[toks.index("layers") + 1]) if "layers" in toks else None
                key = layer if layer is not None else ("tail" if "final_norm" in n else "head")
                off, num = slice_of[n]
                lo, hi = ends.get(key, (off, off + num))
                ends[key] = (min(lo, off), max(hi, off + num))
                claimed[n] = True
            layers = sorted(k for k in ends if isinstance(k, int))
            if not layers:
                continue
            table = {}
            for layer in layers:
                lo, hi = ends[layer]
                if layer == layers[-1] and "tail" in ends:
                    lo, hi = min(lo, ends["tail"][0]), max(hi, ends["tail"][1])
                if layer == layers[0] and "head" in ends:
                    lo, hi = min(lo, ends["head"][0]), max(hi, ends["head"][1])
                table[layer] = len(self.units)
                self.units.append({"key": f"{pre}layer{layer}", "lo": lo, "hi": hi})
            self._enc_units[id(m)] = table
            m.layer_grad_hook = self._make_encoder_hook(id(m))
        # everything else: one unit per owning sub-module (the parameter name without its last component)
        self._pending_init, self._param_unit = {}, {}
        by_owner = {}
        for n, _, off, num in flat.slices:
            if n in claimed:
                continue
            by_owner.setdefault(n.rsplit(".", 1)[0] if "." in n else n, []).append(n)
        for owner, names in by_owner.items():
            lo = min(slice_of[n][0] for n in names)
            hi = max(sum(slice_of[n]) for n in names)
            u = len(self.units)
            self.units.append({"key": owner, "lo": lo, "hi": hi})
            self._pending_init[u] = len(names)
            for n in names:
                self._param_unit[n] = u
                params[n].register_post_accumulate_grad_hook(self._make_param_hook(u))
        # the units must tile the flat buffer: alignment padding belongs to the unit in front of it
        order = sorted(range(len(self.units)), key=lambda u: self.units[u]["lo"])
        for a, b in zip(order, order[1:] + [None]):
            nxt = total if b is None else self.units[b]["lo"]
            if self.units[a]["hi"] > nxt:
                raise ValueError(f"gradient units overlap: {self.units[a]} / {self.units[b]}")
            self.units[a]["hi"] = nxt
        if self.units[order[0]]["lo"] != 0:
            raise ValueError("gradient units do not start at 0")
        self.order = None                    # firing order, learnt in the first optimiser step
        self.agreed = None                   # set after the first step: did every rank complete its units in the same order
        self.active = True
        self.collectives = 0
        self.handles, self.comm = [], None
        self._begin()

    @property
    def ranges(self):
        """unit index -> (lo, hi) of the flat gradient buffer (the slices the collectives cover)"""
        return {u: (unit["lo"], unit["hi"]) for u, unit in enumerate(self.units)}

    # ---- per optimiser step ----
    def _begin(self):
        self._ready = [False] * len(self.units)
        self._fired = [False] * len(self.units)
        self._pending = dict(self._pending_init)
        self._seen = []                      # completion order of this step (first step: becomes self.order)
        self._next = 0

    def _make_encoder_hook(self, enc_id):
        def hook(mod, layer):
            self._complete(self._enc_units[enc_id][layer])
        hook.__self__ = self                 # (the encoder's backward asks its hook's owner whether it joins the side stream)
        return hook

    def _make_param_hook(self, u):
        def hook(param):
            if not self.active:
                return
            self._pending[u] -= 1
            if self._pending[u] == 0:
                self._complete(u)
        return hook

    def _complete(self, u):
        if not self.live or not self.active or self._ready[u]:
            return
        self._ready[u] = True
        self._seen.append(u)
        g = self.flat.grad
        if g.is_cuda:                        # where this unit's gradients were produced: the current stream and its wgrad side stream
            from transfusion_amd import ops
            main = torch.cuda.current_stream(g.device)
            side = ops.side_stream(g.device)
            evs = [main.record_event()] + ([side.record_event()] if side is not None else [])
            if self._debug_break_edge == "side":
                evs = evs[:1]
            elif self._debug_break_edge == "accum" and u in self._pending_init:
                evs = evs[1:]
            self.units[u]["events"] = evs
        if self.order is None:
            return                           # first step: learn the order, reduce everything in finish()
        while self._next < len(self.order) and self._ready[self.order[self._next]]:
            self._fire(self.order[self._next])
            self._next += 1

    def _fire(self, u):
        unit = self.units[u]
        g = self.flat.grad[unit["lo"]:unit["hi"]]
        self._fired[u] = True
        self.collectives += 1
        if not g.is_cuda:
            self.handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=g.device)
        for ev in unit.pop("events", []):
            self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            if self.bucket_comm is not None:
                self.bucket_comm.all_reduce_(g, stream=self.comm)
                _phantom_peers(self._phantom_k, g)
                return
            h = _phantom_peers(self._phantom_k, g, dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            if h is not None:
                self.handles.append(h)

    def finish(self):
        g = self.flat.grad
        if g.is_cuda:
            from transfusion_amd import ops
            ops.join_overlap(g.device)
        if self.live:
            if self.order is None:
                # first step: everything at once, then agree on the order (identical on every rank, or no overlap at all)
                if g.is_cuda:                # ... behind every stream a unit's gradients came from (level streams and their side streams)
                    cur = torch.cuda.current_stream(g.device)
                    for unit in self.units:
                        for ev in unit.pop("events", []):
                            cur.wait_event(ev)
                self.collectives += 1
                if self.bucket_comm is not None:
                    self.bucket_comm.all_reduce_(g)
                else:
                    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                _phantom_peers(self._phantom_k, g)
                mine = self._seen + [u for u in range(len(self.units)) if u not in self._seen]
                orders = [None] * self.world
                dist.all_gather_object(orders, mine, group=self.group)
                self.order = mine if all(o == orders[0] for o in orders) else []
                self.agreed = bool(self.order)   # False: the ranks disagree -- keep reducing after the backward (one collective), never overlap
            else:
                if g.is_cuda:                # units flushed here were produced on streams the current one has already joined
                    ev = torch.cuda.current_stream(g.device).record_event()
                if not self.order:           # no agreed order: one collective over the whole buffer
                    self.collectives += 1
                    if self.bucket_comm is not None:
                        self.bucket_comm.all_reduce_(g)
                    else:
                        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                    _phantom_peers(self._phantom_k, g)
                else:
                    for u in self.order[self._next:]:
                        if g.is_cuda:
                            self.units[u].setdefault("events", []).append(ev)
                        self._fire(u)
                for h in self.handles:
                    h.wait()
                if self.comm is not None:
                    torch.cuda.current_stream(g.device).wait_stream(self.comm)
        self.handles = []
        self._begin()


class FusionTrainStep:
    """One optimiser step over ``accumulate`` micro-batches for a module that writes into ``p.grad`` directly."""

    def __init__(self, module: nn.Module, lr=1e-4, weight_decay=2e-4, grad_clip: Optional[float] = 1.0, accumulate: int = 1,
                 bucket_mb: float = 64.0, optimizer_cls=None, overlap: bool = True, comm: Optional[str] = None,
                 zero_grads_in_optimizer: bool = False, lr_scale=None):
        """``comm``: "torch" (default; torch.distributed's process group) or "rccl" (the C ABI's own communicator,
        tf_allreduce_bucket; also selected by TF_COMM=rccl) -- both are RCCL on a GPU.
        ``lr_scale``: ``callable(parameter_name) -> float`` -- the reference's parameter groups (ego_nao_trainer.py:440-497: ``lr / div_rate``
        for the backbone and the language model, ``lr / ttc_rate`` for the TTC head): consecutive parameters of equal scale become one
        range of the flat buffer = one optimiser group with ``lr * scale`` (one fused launch per range; clipping stays global)."""
        from transfusion_amd.optim import FusedRAdam
        comm = comm or os.environ.get("TF_COMM", "torch")
        if comm not in ("torch", "rccl"):
            raise ValueError(f"comm must be 'torch' or 'rccl', not {comm!r}")
        self.module = module
        self.flat = FlatParams(module)
        for m in module.modules():
            if hasattr(m, "accumulate_into_grad"):
                m.accumulate_into_grad = True
        self.bucket_comm = None
        if comm == "rccl" and _live_world()[1]:
            if not self.flat.grad.is_cuda:
                raise ValueError("comm='rccl' needs the parameters on a GPU")
            from transfusion_amd.comm import BucketComm
            self.bucket_comm = BucketComm.from_process_group(self.flat.grad.device)
        self.reducer = DataParallelReducer(self.flat.grad, bucket_mb, bucket_comm=self.bucket_comm)
        self.world = self.reducer.world
        live = self.reducer.live                  # collectives are issued (world > 1, or a one-rank rehearsal: _live_world)
        self.layerwise = None
        encoders = [m for m in module.modules() if hasattr(m, "layer_grad_hook")]
        force = os.environ.get("TF_FORCE_LAYERWISE") == "1"      # measurement hook: the per-layer call path on one GPU (no-op reduce)
        if overlap and (live or force) and len(encoders) == 1 and encoders[0] is module:
            self.layerwise = LayerwiseReducer(self.flat, bucket_comm=self.bucket_comm)
            module.layer_grad_hook = self.layerwise.hook
        elif overlap and (live or force):
            # any other module tree (the 4-level wrapper, encoder + heads, ...): units fired in a learnt, rank-agreed order
            self.layerwise = OrderedRangeReducer(self.flat, module, bucket_comm=self.bucket_comm)
        if lr_scale is None:
            self.opt = (optimizer_cls or FusedRAdam)([self.flat_param()], lr=lr, weight_decay=weight_decay)
        else:
            ranges = []                                            # [lo, hi, scale] over the flat buffer, in layout order
            for n, _, off, num in self.flat.slices:
                sc = float(lr_scale(n))
                end = off + (num + 63) // 64 * 64                  # (slices start 256-B aligned; the pad floats are zeros with zero gradients)
                if ranges and ranges[-1][2] == sc and ranges[-1][1] == off:
                    ranges[-1][1] = end
                else:
                    ranges.append([off, end, sc])
            groups = []
            self._range_params = []
            for lo, hi, sc in ranges:
                p = self.flat.flat[lo:hi]                          # a view: shares storage (and the optimiser's raw-pointer updates)
                p.grad = self.flat.grad[lo:hi]
                self._range_params.append(p)
                groups.append({"params": [p], "lr": lr * sc})
            self.opt = (optimizer_cls or FusedRAdam)(groups, lr=lr, weight_decay=weight_decay)
            self.lr_ranges = [tuple(r) for r in ranges]
        self.grad_clip = grad_clip
        self.accumulate = accumulate
        # True: the fused optimiser zeroes each gradient as it reads it, and step() skips its own zero fill from the second step on
        # (the flat gradient buffer is then all zeros BETWEEN steps -- a caller who reads gradients after step() keeps this False)
        import inspect
        self.zero_in_opt = bool(zero_grads_in_optimizer) and "zero_grad" in inspect.signature(self.opt.step).parameters
        self._grads_clean = False
        self._norm = torch.zeros(1, dtype=torch.float32, device=self.flat.flat.device)
        # (walked once: the module tree is fixed, and this runs every step of a host-bound loop)
        # one GPU: the linear-type modules (K1 / K9) add their weight gradients straight into the flat gradient buffer too (with a
        # reducer their post-accumulate hooks are what reports a range complete, so there autograd keeps accumulating)
        for m in module.modules():
            if hasattr(m, "accumulate_linear_grad"):
                m.accumulate_linear_grad = not live
        self._all_params = [p for _, p, _, _ in self.flat.slices]
        self._shadow_owners = [m for m in module.modules() if hasattr(m, "mark_weights_updated")]

    def flat_param(self):
        p = self.flat.flat
        p.grad = self.flat.grad
        return p

    def zero_grad(self):
        self.flat.grad.zero_()

    def _root_grad(self, loss, value):
        cache = self.__dict__.setdefault("_root_grads", {})
        key = (loss.device, loss.dtype, tuple(loss.shape), float(value))
        t = cache.get(key)
        if t is None:
            t = cache[key] = torch.full(tuple(loss.shape), float(value), dtype=loss.dtype, device=loss.device)
        return t

    def step(self, micro_batches: List, loss_fn, on_clock: bool = False):
        """``loss_fn(module, batch) -> scalar``; returns the last loss (detached).  ``on_clock``: see FusedRAdam.step."""
        self.check_errors(sync=False)                  # deferred device-side findings of EARLIER steps that have reached the host
        if not (self.zero_in_opt and self._grads_clean):
            self.zero_grad()
        self.flat.check_bound()
        loss = None
        for i, mb in enumerate(micro_batches):
            if self.layerwise is not None:
                # accumulate_grad_batches: gradients are reduced once per optimiser step, during the LAST micro-batch's backward
                # (Lightning's no_sync on the others, run_experiment.py:444-446)
                self.layerwise.active = i == len(micro_batches) - 1
            loss = loss_fn(self.module, mb)
            # d(loss / n) handed to autograd as the root gradient 1 / n (a cached device scalar): no division kernel, no ones-fill and
            # no multiply in front of the backward (three launches of ~5 us each per micro-batch; Lightning divides the loss,
            # run_experiment.py:444-446 -- the gradients are the same numbers)
            loss.backward(gradient=self._root_grad(loss, 1.0 / len(micro_batches)))
        if self.flat.grad.is_cuda:
            # streams whose backward added into the flat gradient buffer without handing autograd a gradient (ops.note_grad_writer): the
            # exchange, the norm and the optimiser below are ordered behind them explicitly
            from transfusion_amd import ops
            ops.join_grad_writers(self.flat.grad.device)
        if self.layerwise is not None:
            self.layerwise.finish()
        else:
            self.reducer.all_reduce()
        scale = 1.0 / self.world
        extra = {}
        if on_clock:
            extra["on_clock"] = True
        if self.zero_in_opt:
            extra["zero_grad"] = True
        if self.grad_clip:
            g = self.flat.grad
            if g.is_cuda:
                # the optimiser's parameters tile the flat buffer (one tensor, or contiguous ranges whose pad floats have zero gradients):
                # its norm is ONE launch that overwrites the scalar (tf_sumsq_set) -- no zero fill, no launch per range; stays on the
                # device: no host sync in the step
                from transfusion_amd import _lib as L, ops
                L.check(L.load().tf_sumsq_set(L.ptr(g), g.numel(), L.ptr(self._norm), ops._stream()), "tf_sumsq_set")
            else:
                self._norm.zero_()
                self.opt.grad_sumsq(self._norm)
            self.opt.step(grad_scale=scale, sumsq=self._norm, clip=self.grad_clip, **extra)
        else:
            self.opt.step(grad_scale=scale, **extra)
        self._grads_clean = self.zero_in_opt
        self.mark_parameters_updated()
        return loss.detach()

    @staticmethod
    def check_errors(sync: bool = True):
        """What the kernels found wrong WITHOUT stopping the stream: noun / verb labels outside their class range (the loss kernel
        drops such samples; torch's cross_entropy would have raised IndexError -- ``ops.check_label_errors``) and a ``lang_valid_rows``
        that disagreed with its padding mask (``check_packed_row_errors``).  ``step()`` looks at what has already reached the host
        (``sync=False``: no stall) at the start of every step, so a bad batch raises one or two steps later; call this with
        ``sync=True`` where a wait is cheap -- the end of an epoch, before a checkpoint, at teardown -- so that none stays unseen."""
        from transfusion_amd import ops
        from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import check_packed_row_errors
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
        ops.check_label_errors(sync=sync)
        check_packed_row_errors(sync=sync)

    def mark_parameters_updated(self):
        """The fused optimiser wrote through the FLAT buffer: it bumped that tensor's version counter, but a re-homed parameter
        (``p.data = flat[off:off+n]``) keeps a version counter of its own, and every bf16 weight-shadow cache of the library keys on
        ``(p.data_ptr(), p._version)`` (ops._weight_shadows, QKVEncoder._shadows, CrossTransformerModuleBox._wpack_dirty).  Without
        this bump PatchToToken / RegroupPatchesLayerBox / the heads / the asymmetric layers would keep multiplying by their step-0
        weights."""
        torch._C._increment_version(self._all_params)
        for m in self._shadow_owners:
            m.mark_weights_updated()
