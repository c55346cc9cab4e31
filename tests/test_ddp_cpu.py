"""Data-parallel path on CPU with the gloo backend, world_size 2: the flat-buffer gradient all-reduce of
transfusion_amd.runner.trainer equals the single-process gradient of the concatenated batch, parameters that
never receive gradients are excluded statically, and bucket boundaries do not matter."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(6, 5)
        self.b = torch.nn.Linear(5, 3)
        self.heatmap_token = torch.nn.Parameter(torch.randn(1, 1, 6))      # never used, like the reference's
        self.frozen = torch.nn.Parameter(torch.randn(4), requires_grad=False)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transfusion_amd.runner.trainer import DataParallelReducer, FlatParams
    torch.manual_seed(123)
    data = torch.randn(8, 6)
    model = Toy()
    flat = FlatParams(model)
    assert "heatmap_token" not in flat.names and "frozen" not in flat.names
    red = DataParallelReducer(flat.grad, bucket_mb=1e-4)        # ~26 floats per bucket: several buckets
    assert red.world == world and len(red.buckets) > 1
    shard = data[rank * 4:(rank + 1) * 4]
    model(shard).pow(2).sum().backward()                         # autograd accumulates into the flat views
    for h in red.all_reduce(async_op=True):
        h.wait()
    if rank == 0:
        torch.save({"grad": flat.grad.clone(), "names": flat.names, "slices": [(n, o, k) for n, _, o, k in flat.slices]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_flat_allreduce(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(123)
    data = torch.randn(8, 6)
    ref = Toy()
    ref(data).pow(2).sum().backward()
    named = dict(ref.named_parameters())
    for n, off, k in got["slices"]:
        torch.testing.assert_close(got["grad"][off:off + k], named[n].grad.reshape(-1), rtol=1e-5, atol=1e-6)


def test_single_process_reducer_is_noop():
    from transfusion_amd.runner.trainer import DataParallelReducer
    g = torch.ones(10)
    r = DataParallelReducer(g)
    assert r.world == 1 and r.all_reduce() == [] and torch.equal(g, torch.ones(10))


def test_layerwise_ranges_cover_flat_buffer_once():
    """The per-layer all-reduce slices (top layer + final norm first, layer 0 + kind embeddings last) tile the flat
    gradient buffer exactly: nothing reduced twice, nothing forgotten."""
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    from transfusion_amd.runner.trainer import FlatParams, LayerwiseReducer
    pe = PositionalEmbeddingLayer("sin1d", 64, 32)
    enc = CrossTransformerModuleBox(no_patches=64, pos_embedding_layer=pe, lang_pos_embedding=None, num_layers=3, patch_dropout=0.1,
                                    num_heads=2, fforward_multiplier=2, token_dropout=0.1, back_to_img_fn="regroup", activ_f="gelu",
                                    final_norm="ln", input_f_size=32)
    flat = FlatParams(enc)
    red = LayerwiseReducer(flat)
    assert red.num_layers == 3 and red.world == 1
    spans = sorted(red.ranges.values())
    assert spans[0][0] == 0 and spans[-1][1] == flat.grad.numel()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0                      # exact tiling: no gap, no overlap
    # every parameter slice lies in exactly one range, final norm with the top layer, kind embeddings with layer 0
    where = {}
    for name, _, off, num in flat.slices:
        hits = [l for l, (lo, hi) in red.ranges.items() if lo <= off and off + num <= hi]
        assert len(hits) == 1, name
        where[name] = hits[0]
    assert where["final_norm_layer.weight"] == 2 and where["image_kind_embedding"] == 0
    assert where["t_encoder.layers.1.linear1.weight"] == 1


# ----------------------------------------------------------------------------------------------------------------------
# the DEFAULT N > 1 path: LayerwiseReducer.hook driven layer by layer (top first) from inside the backward
# ----------------------------------------------------------------------------------------------------------------------
class _StubEncoderFn(torch.autograd.Function):
    """Backward of a stack of tanh(Linear) layers written the way the encoder runtime's is: one layer at a time, top first,
    gradients ACCUMULATED straight into p.grad, the module's layer_grad_hook called after each layer."""

    @staticmethod
    def forward(ctx, mod, x, *params):       # params: only so that autograd builds a node (gradients go straight to p.grad)
        ctx.nparams = len(params)
        acts = [x]
        for lin in mod.t_encoder.layers:
            acts.append(torch.tanh(acts[-1] @ lin.weight.t() + lin.bias))
        ctx.mod, ctx.acts = mod, acts
        return acts[-1] * mod.final_norm_layer.weight + mod.image_kind_embedding.view(-1)

    @staticmethod
    def backward(ctx, g):
        mod, acts = ctx.mod, ctx.acts
        mod.final_norm_layer.weight.grad += (g * acts[-1]).sum(0)
        g_kind = g.sum(0)
        g = g * mod.final_norm_layer.weight
        for layer in range(len(mod.t_encoder.layers) - 1, -1, -1):
            lin = mod.t_encoder.layers[layer]
            gz = g * (1 - acts[layer + 1] ** 2)
            lin.weight.grad += gz.t() @ acts[layer]
            lin.bias.grad += gz.sum(0)
            g = gz @ lin.weight
            if layer == 0:
                mod.image_kind_embedding.grad += g_kind.view(1, 1, -1)
            if mod.layer_grad_hook is not None:
                mod.layer_grad_hook(mod, layer)
        return (None, g) + (None,) * ctx.nparams


class StubEncoder(torch.nn.Module):
    """Parameter names of CrossTransformerModuleBox (kind embedding, t_encoder.layers.{j}.*, final_norm_layer.*) so that
    LayerwiseReducer derives its ranges exactly as for the real encoder."""

    def __init__(self, d=6, L=3):
        super().__init__()
        torch.manual_seed(1)
        self.image_kind_embedding = torch.nn.Parameter(torch.randn(1, 1, d))
        self.heatmap_token = torch.nn.Parameter(torch.randn(1, 1, d))
        self.t_encoder = torch.nn.Module()
        self.t_encoder.layers = torch.nn.ModuleList([torch.nn.Linear(d, d) for _ in range(L)])
        self.final_norm_layer = torch.nn.Module()
        self.final_norm_layer.weight = torch.nn.Parameter(torch.rand(d) + 0.5)
        self.layer_grad_hook = None
        self.accumulate_into_grad = False

    def forward(self, x):
        return _StubEncoderFn.apply(self, x, *[p for n, p in self.named_parameters() if n != "heatmap_token"])


class _PlainSGD:
    """CPU stand-in for the fused optimiser (the HIP RAdam needs a GPU): same call protocol as FusedRAdam.step."""

    def __init__(self, params, lr, weight_decay):
        self.params, self.lr = list(params), lr

    def grad_sumsq(self, out):
        for p in self.params:
            out += p.grad.double().pow(2).sum().float()

    def step(self, grad_scale=1.0, sumsq=None, clip=0.0):
        for p in self.params:
            p.data.add_(p.grad, alpha=-self.lr * grad_scale)


def _layerwise_worker(rank, world, port, out, accumulate):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transfusion_amd.runner import trainer as T
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(int(t.numel()))
        return real(t, *a, **k)

    T.dist.all_reduce = counting
    torch.manual_seed(7)
    data = torch.randn(2 * world * accumulate, 6)
    model = StubEncoder()
    tr = T.FusionTrainStep(model, lr=0.1, weight_decay=0.0, grad_clip=None, accumulate=accumulate, optimizer_cls=_PlainSGD)
    assert tr.layerwise is not None and model.layer_grad_hook is not None        # the default N > 1 path
    mine = data[rank::world]                                                     # this rank's samples
    mbs = list(mine.chunk(accumulate))
    loss_fn = lambda m, b: m(b).pow(2).sum()
    tr.step(mbs, loss_fn)
    # second step: the reducer state (handles, active flag) must have been left clean
    before2 = tr.flat.flat.clone()
    tr.step(mbs, loss_fn)
    counts = [None] * world
    dist.all_gather_object(counts, calls)
    if rank == 0:
        torch.save({"grad": tr.flat.grad.clone(), "param": tr.flat.flat.clone(), "before2": before2, "counts": counts,
                    "slices": [(n, o, k) for n, _, o, k in tr.flat.slices], "ranges": dict(tr.layerwise.ranges)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("accumulate", [1, 2])
def test_gloo_world2_layerwise_hook_path(tmp_path, accumulate):
    """FusionTrainStep with world 2 takes the LayerwiseReducer path: per-layer all-reduces issued from inside the backward.
    The reduced gradient equals the single-process gradient over all samples (x 1/accumulate for the micro-batch scaling,
    x 1 for the SUM over ranks), both ranks issue the SAME collectives (count and sizes: a mismatch is the hang of round 1's
    rehearsal), and with accumulate_grad_batches = 2 the reduction happens once per optimiser step, on the last micro-batch."""
    out = str(tmp_path / "lw.pt")
    world = 2
    mp.spawn(_layerwise_worker, args=(world, _free_port(), out, accumulate), nprocs=world, join=True)
    got = torch.load(out)
    assert got["counts"][0] == got["counts"][1]
    L = 3
    assert len(got["counts"][0]) == 2 * L                                       # L collectives per optimiser step, 2 steps
    assert sum(got["counts"][0][:L]) == got["grad"].numel()                      # the ranges tile the flat buffer exactly
    # reference: one process, all samples, parameters as they were before the second step
    torch.manual_seed(7)
    data = torch.randn(2 * world * accumulate, 6)
    ref = StubEncoder()
    named = dict(ref.named_parameters())
    for n, off, k in got["slices"]:
        named[n].data.copy_(got["before2"][off:off + k].view_as(named[n]))
        named[n].grad = torch.zeros_like(named[n])
    ref(data).pow(2).sum().backward()
    for n, off, k in got["slices"]:
        torch.testing.assert_close(got["grad"][off:off + k], named[n].grad.reshape(-1) / accumulate, rtol=1e-4, atol=1e-5)
        # and the update used grad / world (FusionTrainStep's mean over ranks)
        want = named[n].data.reshape(-1) - 0.1 * named[n].grad.reshape(-1) / accumulate / world
        torch.testing.assert_close(got["param"][off:off + k], want, rtol=1e-4, atol=1e-5)


def test_train_step_bumps_version_of_rehomed_parameters():
    """A re-homed parameter (``p.data = flat[off:off+n]``) has a version counter of its own, and the fused optimiser only writes the
    flat buffer: FusionTrainStep must bump every parameter's version after the step, or the bf16 weight-shadow caches keyed on
    ``(data_ptr, _version)`` (ops._weight_shadows, QKVEncoder._shadows) keep serving the step-0 weights."""
    from transfusion_amd.runner import trainer as T
    model = Toy()
    tr = T.FusionTrainStep(model, lr=0.1, weight_decay=0.0, grad_clip=None, optimizer_cls=_PlainSGD)
    ps = [p for _, p, _, _ in tr.flat.slices]
    ptrs = [p.data_ptr() for p in ps]
    for it in range(3):
        before = [p._version for p in ps]
        w0 = model.a.weight.detach().clone()
        tr.step([torch.randn(4, 6)], lambda m, b: m(b).pow(2).sum())
        assert all(p._version > v for p, v in zip(ps, before)), it
        assert not torch.equal(w0, model.a.weight)                     # the step did move the weights ...
    assert ptrs == [p.data_ptr() for p in ps]                          # ... in place: data_ptr alone would never have told


# ----------------------------------------------------------------------------------------------------------------------
# OrderedRangeReducer: gradient exchange overlapped with the backward for a module TREE (the wrapper's shape: several encoders,
# each between a patch-embedding and a back-projection Linear, plus a head)
# ----------------------------------------------------------------------------------------------------------------------
class StubWrapper(torch.nn.Module):
    """Two feature levels, each: Linear (K1) -> StubEncoder (per-layer backward with layer_grad_hook) -> Linear (K9); one head.  Parameter
    names follow CrossFusionBoxWrapper's (cross_fusion_encoders.{i}..., patches_to_token.{i}, tokens_to_features.{i})."""

    def __init__(self, d=6):
        super().__init__()
        torch.manual_seed(3)
        self.cross_fusion_encoders = torch.nn.ModuleList([StubEncoder(d, 2), StubEncoder(d, 3)])
        self.patches_to_token = torch.nn.ModuleList([torch.nn.Linear(4, d, bias=False), torch.nn.Linear(5, d, bias=False)])
        self.tokens_to_features = torch.nn.ModuleList([torch.nn.Linear(d, 4), torch.nn.Linear(d, 5)])
        self.head = torch.nn.Linear(9, 3)
        self.unused = torch.nn.Linear(2, 2)              # trainable, but no gradient in any step (a branch that is switched off)

    def forward(self, xs):
        outs = [self.tokens_to_features[i](self.cross_fusion_encoders[i](self.patches_to_token[i](xs[i]))) for i in range(2)]
        return self.head(torch.cat(outs, dim=-1))


def _tree_worker(rank, world, port, out, accumulate):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transfusion_amd.runner import trainer as T
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(int(t.numel()))
        return real(t, *a, **k)

    T.dist.all_reduce = counting
    torch.manual_seed(11)
    n = 2 * world * accumulate
    data = [torch.randn(n, 4), torch.randn(n, 5)]
    model = StubWrapper()
    tr = T.FusionTrainStep(model, lr=0.05, weight_decay=0.0, grad_clip=None, accumulate=accumulate, optimizer_cls=_PlainSGD)
    assert isinstance(tr.layerwise, T.OrderedRangeReducer)
    mine = [d[rank::world] for d in data]
    mbs = [[a, b] for a, b in zip(mine[0].chunk(accumulate), mine[1].chunk(accumulate))]
    loss_fn = lambda m, b: m(b).pow(2).sum()
    steps = []
    for it in range(3):
        before = tr.flat.flat.clone()
        n0 = len(calls)
        tr.step(mbs, loss_fn)
        steps.append({"before": before, "grad": tr.flat.grad.clone(), "param": tr.flat.flat.clone(), "calls": calls[n0:]})
    counts = [None] * world
    dist.all_gather_object(counts, calls)
    if rank == 0:
        torch.save({"steps": steps, "counts": counts, "slices": [(n_, o, k) for n_, _, o, k in tr.flat.slices], "agreed": tr.layerwise.agreed,
                    "units": [(u["key"], u["lo"], u["hi"]) for u in tr.layerwise.units], "order": tr.layerwise.order}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("accumulate", [1, 2])
def test_gloo_world2_ordered_range_reducer_on_a_module_tree(tmp_path, accumulate):
    """World 2, a wrapper-shaped tree: step 1 reduces the whole buffer in one collective and learns the completion order; steps 2 and 3
    fire one collective per unit from inside the backward, in that order, on both ranks alike; the units tile the flat buffer; the unit
    that never receives a gradient is flushed at the end; reduced gradients and updated parameters equal the single-process run."""
    out = str(tmp_path / "tree.pt")
    world = 2
    mp.spawn(_tree_worker, args=(world, _free_port(), out, accumulate), nprocs=world, join=True)
    got = torch.load(out)
    assert got["agreed"] is True
    assert got["counts"][0] == got["counts"][1]
    units = sorted(got["units"], key=lambda u: u[1])
    total = got["steps"][0]["grad"].numel()
    assert units[0][1] == 0 and units[-1][2] == total and all(a[2] == b[1] for a, b in zip(units, units[1:]))
    keys = [u[0] for u in got["units"]]
    # 2 + 3 encoder layers, two K1, two K9, the head, the unused Linear
    assert sum(k.startswith("cross_fusion_encoders.0.layer") for k in keys) == 2 and sum(k.startswith("cross_fusion_encoders.1.layer") for k in keys) == 3
    assert {"patches_to_token.0", "patches_to_token.1", "tokens_to_features.0", "tokens_to_features.1", "head", "unused"} <= set(keys)
    assert got["steps"][0]["calls"] == [total]                                        # step 1: one collective over everything
    sizes = {i: hi - lo for i, (_, lo, hi) in enumerate(got["units"])}
    for st in got["steps"][1:]:                                                       # later steps: one per unit, in the learnt order
        assert st["calls"] == [sizes[u] for u in got["order"]]
    assert got["order"][-1] == keys.index("unused")                                   # never completes: flushed last
    # backward order: the head first, then level 1 (K9, layers top-down, K1), then level 0
    first = [keys[u] for u in got["order"][:3]]
    assert first[0] == "head" and first[1] == "tokens_to_features.1" and first[2] == "cross_fusion_encoders.1.layer2"
    # numerics: every step against a single process over all samples, from the parameters that step started with
    torch.manual_seed(11)
    n = 2 * world * accumulate
    data = [torch.randn(n, 4), torch.randn(n, 5)]
    for st in got["steps"]:
        ref = StubWrapper()
        named = dict(ref.named_parameters())
        for nme, off, k in got["slices"]:
            named[nme].data.copy_(st["before"][off:off + k].view_as(named[nme]))
            named[nme].grad = torch.zeros_like(named[nme])
        ref(data).pow(2).sum().backward()
        for nme, off, k in got["slices"]:
            torch.testing.assert_close(st["grad"][off:off + k], named[nme].grad.reshape(-1) / accumulate, rtol=1e-4, atol=1e-5)
            want = named[nme].data.reshape(-1) - 0.05 * named[nme].grad.reshape(-1) / accumulate / world
            torch.testing.assert_close(st["param"][off:off + k], want, rtol=1e-4, atol=1e-5)


class FlippableWrapper(StubWrapper):
    """StubWrapper whose two levels can be evaluated in either order: autograd replays independent branches latest-created first, so
    ``flip`` reverses the order in which this rank's units COMPLETE in the backward -- the ranks of a real job see such differences
    through timing (level streams, side streams); here they are made deterministic."""
    flip = False

    def forward(self, xs):
        outs = {}
        for i in ((1, 0) if self.flip else (0, 1)):
            outs[i] = self.tokens_to_features[i](self.cross_fusion_encoders[i](self.patches_to_token[i](xs[i])))
        return self.head(torch.cat([outs[0], outs[1]], dim=-1))


def _uneven_worker(rank, world, port, out, mode):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transfusion_amd.runner import trainer as T
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(int(t.numel()))
        return real(t, *a, **k)

    T.dist.all_reduce = counting
    torch.manual_seed(11)
    n = 2 * world
    data = [torch.randn(n, 4), torch.randn(n, 5)]
    model = FlippableWrapper()
    odd = rank % 2 == 1
    model.flip = odd and mode == "first_step"
    tr = T.FusionTrainStep(model, lr=0.05, weight_decay=0.0, grad_clip=None, optimizer_cls=_PlainSGD)
    mine = [d[rank::world] for d in data]
    steps, local_orders = [], []
    for it in range(3):
        if mode == "later_steps" and it >= 1:
            model.flip = odd                                   # the order was agreed on in step 1; from step 2 on the odd ranks complete differently
        before = tr.flat.flat.clone()
        n0 = len(calls)
        tr.step([[mine[0], mine[1]]], lambda m, b: m(b).pow(2).sum())
        steps.append({"before": before, "grad": tr.flat.grad.clone(), "param": tr.flat.flat.clone(), "calls": calls[n0:]})
    counts, agreed = [None] * world, [None] * world
    dist.all_gather_object(counts, calls)
    dist.all_gather_object(agreed, bool(tr.layerwise.agreed))
    if rank == 0:
        torch.save({"steps": steps, "counts": counts, "agreed": agreed, "slices": [(n_, o, k) for n_, _, o, k in tr.flat.slices],
                    "units": [(u["key"], u["lo"], u["hi"]) for u in tr.layerwise.units], "order": tr.layerwise.order}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["first_step", "later_steps"])
def test_gloo_world4_ordered_range_reducer_with_uneven_completion_order(tmp_path, mode):
    """FOUR ranks whose units complete in DIFFERENT orders (what the one-rank RCCL rehearsal on the GPU box cannot show: it has no peer
    to disagree with).  ``first_step``: the odd ranks see another order already in the learning step -- the all-gather of the orders
    must find the disagreement on EVERY rank and the reducer must never overlap (one collective over the whole buffer per step);
    ``later_steps``: the order is agreed on in step 1 and the odd ranks complete differently afterwards -- every rank must still issue
    the collectives in the agreed order (a unit that completes early waits for its turn).  Either way: identical collective sequences
    on all four ranks (anything else is a hang on real hardware) and gradients / parameters equal to one process over all samples."""
    out = str(tmp_path / f"uneven_{mode}.pt")
    world = 4
    mp.spawn(_uneven_worker, args=(world, _free_port(), out, mode), nprocs=world, join=True)
    got = torch.load(out)
    assert all(c == got["counts"][0] for c in got["counts"]), got["counts"]                # the SAME sequence of collective sizes everywhere
    total = got["steps"][0]["grad"].numel()
    sizes = {i: hi - lo for i, (_, lo, hi) in enumerate(got["units"])}
    if mode == "first_step":
        assert got["agreed"] == [False] * world                                           # every rank knows, none overlaps
        for st in got["steps"]:
            assert st["calls"] == [total]
    else:
        assert got["agreed"] == [True] * world
        assert got["steps"][0]["calls"] == [total]
        for st in got["steps"][1:]:
            assert st["calls"] == [sizes[u] for u in got["order"]]                        # the agreed order, whatever completed first locally
    torch.manual_seed(11)
    n = 2 * world
    data = [torch.randn(n, 4), torch.randn(n, 5)]
    for st in got["steps"]:
        ref = StubWrapper()
        named = dict(ref.named_parameters())
        for nme, off, k in got["slices"]:
            named[nme].data.copy_(st["before"][off:off + k].view_as(named[nme]))
            named[nme].grad = torch.zeros_like(named[nme])
        ref(data).pow(2).sum().backward()
        for nme, off, k in got["slices"]:
            torch.testing.assert_close(st["grad"][off:off + k], named[nme].grad.reshape(-1), rtol=1e-4, atol=1e-5)
            want = named[nme].data.reshape(-1) - 0.05 * named[nme].grad.reshape(-1) / world
            torch.testing.assert_close(st["param"][off:off + k], want, rtol=1e-4, atol=1e-5)


def test_lr_scale_builds_contiguous_ranges_with_their_own_rate():
    """FusionTrainStep(lr_scale=...): the reference's parameter groups (lr / div_rate, lr / ttc_rate) as ranges of the flat buffer."""
    import torch
    from transfusion_amd.runner.trainer import FusionTrainStep

    class Opt(torch.optim.Optimizer):                              # records what it was given; never steps (no GPU here)
        def __init__(self, params, lr, weight_decay):
            super().__init__(params, dict(lr=lr, weight_decay=weight_decay))

    m = torch.nn.Sequential(torch.nn.Linear(4, 6), torch.nn.Linear(6, 3), torch.nn.Linear(3, 2))
    tr = FusionTrainStep(m, lr=1e-3, weight_decay=0.0, grad_clip=None, optimizer_cls=Opt,
                         lr_scale=lambda name: 0.1 if name.startswith("1.") else 1.0)
    starts = [off for _, _, off, _ in tr.flat.slices] + [tr.flat.flat.numel()]           # slices start 256-B aligned
    assert [(lo, hi) for lo, hi, _ in tr.lr_ranges] == [(starts[0], starts[2]), (starts[2], starts[4]), (starts[4], starts[6])]
    assert [round(g["lr"], 8) for g in tr.opt.param_groups] == [1e-3, 1e-4, 1e-3]
    for g, (lo, hi, _) in zip(tr.opt.param_groups, tr.lr_ranges):
        p = g["params"][0]
        assert p.data_ptr() == tr.flat.flat.data_ptr() + 4 * lo and p.numel() == hi - lo
        assert p.grad.data_ptr() == tr.flat.grad.data_ptr() + 4 * lo


def test_graph_capture_guard_names_the_nested_stream_fork():
    """graph_step.check_capturable: a wrapper that ran its levels on level streams while the encoders fork weight-gradient side
    streams cannot be captured (hipStreamEndCapture crashes on the nested fork, ROCm 7.2) -- a ValueError, not a core dump."""
    import torch
    from graph_step import check_capturable

    class Wrapper(torch.nn.Module):
        def __init__(self, path):
            super().__init__()
            self.lin = torch.nn.Linear(2, 2)
            self._last_path = path
    tree = torch.nn.Sequential(torch.nn.Linear(2, 2), Wrapper("streams"))
    with pytest.raises(ValueError, match="TF_LEVEL_STREAMS=0"):
        check_capturable(tree, True)
    check_capturable(tree, False)                                  # no side streams: one fork level, capturable
    check_capturable(torch.nn.Sequential(Wrapper("grouped"), Wrapper("loop")), True)
