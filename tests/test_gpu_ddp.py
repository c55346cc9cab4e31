"""The N > 1 code paths on ONE MI355X: two ranks share the device over gloo (no second GPU is available to these tests; RCCL itself is
exercised only by the driver's multi-GPU run).  What this does cover is everything around the collective that differs from the CPU tests:
the real encoder's per-layer backward with `defer_join`, `LayerwiseReducer.hook` on its CUDA branch (communication stream behind one
event of the chain and one of the side stream, asynchronous all-reduce), `finish()`, the fused optimiser on the flat buffer -- and
bench.py's own N = 2 control flow end to end under a timeout."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from cases import make_encoder_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
from cases import make_encoder_inputs
from test_gpu_encoder import build
from transfusion_amd.optim import FusedRAdam
from transfusion_amd.runner.trainer import FusionTrainStep
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rehearse = os.environ.get("TF_REHEARSE_COLLECTIVES") == "1"          # ONE rank over RCCL with a phantom peer (trainer._phantom_peers)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if rehearse:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
if int(os.environ.get("TF_TEST_WGRAD_DELAY_US", "0")):
    from transfusion_amd import ops
    ops.debug_delay_wgrad(int(os.environ["TF_TEST_WGRAD_DELAY_US"]))
if os.environ.get("TF_TEST_BREAK_EDGE"):
    from transfusion_amd.runner.trainer import LayerwiseReducer
    LayerwiseReducer._debug_break_edge = os.environ["TF_TEST_BREAK_EDGE"]
cfg = dict(B=4, Nv=24, Nl=30, d=64, h=4, L=3, seed=91)
enc, _ = build(cfg, dev)
enc.train()
sgd = lambda ps, lr, weight_decay: FusedRAdam(ps, lr=lr, weight_decay=weight_decay, degenerated_to_sgd=True)
tr = FusionTrainStep(enc, lr=5e-2, weight_decay=0.0, grad_clip=None, accumulate={acc}, optimizer_cls=sgd)
assert tr.layerwise is not None and enc.layer_grad_hook is not None, "the default N > 1 path must be the layer-wise reducer"
x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], [30, 11, 22, 30])
t = lambda a: torch.from_numpy(a).to(dev)
mine = slice(0, 4) if rehearse else slice(rank * 2, rank * 2 + 2)       # this rank's two samples (the rehearsal's one rank: all four)
def loss_fn(m, b):
    sl = b
    v, l_, _, _ = m(t(x[sl]), t(lang[sl]), t(mask[sl]))
    return (v * t(gv[sl])).sum() + (l_ * t(gl[sl])).sum()
mbs = [mine] if {acc} == 1 else [slice(rank * 2, rank * 2 + 1), slice(rank * 2 + 1, rank * 2 + 2)]
before = tr.flat.flat.clone()
tr.step(mbs, loss_fn)
torch.cuda.synchronize()
# every rank must hold bit-identical parameters afterwards
chk = tr.flat.flat.double().sum().reshape(1)
chk = chk if rehearse else chk.cpu()                       # (RCCL reduces device tensors, gloo is given host ones)
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert lo.item() == hi.item(), (lo.item(), hi.item())
if rank == 0:
    torch.save({{"grad": tr.flat.grad.cpu(), "param": tr.flat.flat.cpu(), "before": before.cpu(), "collectives": tr.layerwise.collectives,
                "backend": dist.get_backend(), "own_comm": tr.bucket_comm is not None}}, {out!r})
dist.barrier()
dist.destroy_process_group()
'''


def _encoder_reference_grad(before):
    """single process, all four samples, same starting parameters"""
    from test_gpu_encoder import build
    from transfusion_amd.runner.trainer import FlatParams
    dev = torch.device("cuda:0")
    cfg = dict(B=4, Nv=24, Nl=30, d=64, h=4, L=3, seed=91)
    enc, _ = build(cfg, dev)
    enc.train()
    flat = FlatParams(enc)
    assert torch.equal(flat.flat.cpu(), before)
    enc.accumulate_into_grad = True
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], [30, 11, 22, 30])
    t = lambda a: torch.from_numpy(a).to(dev)
    v, l_, _, _ = enc(t(x), t(lang), t(mask))
    ((v * t(gv)).sum() + (l_ * t(gl)).sum()).backward()
    torch.cuda.synchronize()
    return flat.grad.cpu()


@pytest.mark.parametrize("acc,delay_us", [(1, 0), (2, 0), (1, 3000)])
def test_two_ranks_one_gpu_layerwise_reducer_on_the_real_encoder(tmp_path, acc, delay_us):
    """delay_us: every weight-gradient launch sits behind a spin of that many microseconds (ops.debug_delay_wgrad): a collective that
    lacked an event edge to the side stream would reduce zeros, every time."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    out = str(tmp_path / "ddp.pt")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=out, acc=acc))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   TF_TEST_WGRAD_DELAY_US=str(delay_us))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank run hung (a rank issued a different number of collectives?)")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    got = torch.load(out)
    assert got["collectives"] == 3                               # one all-reduce per layer per optimiser step (also with accumulation)
    ref = _encoder_reference_grad(got["before"]) / acc            # micro-batch losses are scaled by 1 / accumulate
    err = ((got["grad"] - ref).norm() / ref.norm()).item()
    assert err < 2e-3, err                                        # same products, different fp32 summation order (atomics, ranks)
    # SGD-degenerated first RAdam step: p -= lr * step_size * m, m = (1 - beta1) * g / world -- the parameters moved by the mean gradient
    moved = got["param"] - got["before"]
    want = -5e-2 * (1.0 / (1 - 0.9)) * (1 - 0.9) * (ref / 2)
    assert ((moved - want).norm() / want.norm()).item() < 2e-3


def test_bench_two_ranks_on_one_gpu_does_not_hang(tmp_path):
    """bench.py --gpus 2 through torch.distributed.run, both ranks on cuda:0 over gloo (the rehearsal hooks TF_FORCE_DEVICE /
    TF_DIST_BACKEND): warm-up, timed steps, traced steps on EVERY rank, the all-reduce bandwidth probe, one JSON line from rank 0."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import json
    env = dict(os.environ, TF_FORCE_DEVICE="0", TF_DIST_BACKEND="gloo", TF_CHECK_SYNC="1", TF_BENCH_WATCHDOG_S="200", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--trace-steps", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 64 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["allreduce"]["collectives_per_step"] == 4 and res["allreduce"]["bytes"] > 7e7
    assert "parameter checksum identical on 2 ranks" in r.stderr
    assert res["rank_sync"]["identical_on_ranks"] == 2                      # the self-check runs by default at N > 1
    # strong scaling next to the weak number: the reference divides the GLOBAL batch of 32 by the device count (run_experiment.py:373)
    assert res["strong"]["global_batch"] == 32 and res["strong"]["batch_per_gpu"] == 16 and res["strong"]["samples_s"] > 0
    # ... and the reference's real module (four-level wrapper, real FPN geometry, batch 4 per rank) under the ordered range reducer
    w = res["wrapper_b4_real_dp"]
    assert "error" not in w and w["ms_per_step"] > 0 and w["reducer"] == "OrderedRangeReducer" and w["batch_per_gpu"] == 4, w


def test_bench_bare_gpus_2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself as a child process and passes
    rank 0's JSON line through (what the driver's SCALE run calls; on this one-GPU box both ranks share cuda:0 over gloo)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TF_FORCE_DEVICE="0", TF_DIST_BACKEND="gloo", TF_BENCH_WATCHDOG_S="200", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-census", "--no-legs"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 64 and res["value"] > 0
    assert res["allreduce"]["group_world"] == 2 and res["allreduce"]["backend"] == "gloo"
    assert res["rank_sync"]["identical_on_ranks"] == 2


_TREE_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
from test_gpu_wrapper import StubDetector
from transfusion_amd.modeling.model_factory import get_fusion_model
from transfusion_amd.optim import FusedRAdam
from transfusion_amd.runner.config import load_fusion_config
from transfusion_amd.runner.trainer import FusionTrainStep, OrderedRangeReducer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
single = os.environ.get("TF_TREE_SINGLE") == "1"
rehearse = os.environ.get("TF_REHEARSE_COLLECTIVES") == "1"          # ONE rank over RCCL with a phantom peer (trainer._phantom_peers)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if rehearse:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
elif not single:
    dist.init_process_group("gloo", rank=rank, world_size=world)
if int(os.environ.get("TF_TEST_WGRAD_DELAY_US", "0")):
    from transfusion_amd import ops
    ops.debug_delay_wgrad(int(os.environ["TF_TEST_WGRAD_DELAY_US"]))
if os.environ.get("TF_TEST_BREAK_EDGE"):                    # negative control: the reducer WITHOUT one of its event edges
    OrderedRangeReducer._debug_break_edge = os.environ["TF_TEST_BREAK_EDGE"]
d, h, L = 64, 4, 2
levels = {levels}
fusion = load_fusion_config(os.path.join({root!r}, "transfusion_amd", "runner", "configs", "cross_fusion_config_sym_ego_res50.yml"))
fusion.update({{"fpn_features": [0, 1], "replace_fpn_features": True, "patch_h": [l["p"] for l in levels], "patch_w": [l["p"] for l in levels],
               "backproj_dropout": 0.0}})
fusion["args"].update({{"num_layers": [L] * 2, "num_heads": h, "patch_dropout": 0.0, "token_dropout": 0.0, "input_f_size": d}})
run_cfg = {{"experiment": "egonao", "narr_fusion": fusion, "criterion": {{"lm": 0}}, "precision": 16,
           "narration_embeds": {{"use": True, "args": {{"text_pooling": "slowfast", "strategy": "current", "out_mlp": 0, "size": d,
                                                     "out_dropout": 0.0, "out_tanh": False, "train_ep": 0}}}}}}
torch.manual_seed(5)
model = get_fusion_model(StubDetector([(l["H"], l["W"]) for l in levels], [l["C"] for l in levels]), {{}}, run_cfg, None).to(dev).train()
sgd = lambda ps, lr, weight_decay: FusedRAdam(ps, lr=lr, weight_decay=weight_decay, degenerated_to_sgd=True)
tr = FusionTrainStep(model, lr=2e-2, weight_decay=0.0, grad_clip=1.0, optimizer_cls=sgd)
if not single:
    assert isinstance(tr.layerwise, OrderedRangeReducer), "a module tree must take the ordered range reducer"
g = torch.Generator().manual_seed(77)
B = 4
feats = [torch.randn(B, l["C"], l["H"], l["W"], generator=g) for l in levels]
lens = [9, 3, 11, 6]
lang = [torch.randn(n, d, generator=g) for n in lens]
cots = [torch.randn(B, l["C"], l["H"], l["W"], generator=g) for l in levels]
mine = list(range(B)) if single or rehearse else [2 * rank, 2 * rank + 1]
def loss_fn(m, idx):
    out = m({{"image": [f[idx].to(dev) for f in feats], "language_f": [lang[i].to(dev) for i in idx]}})
    return sum((out["features"][str(i)].float() * cots[i][idx].to(dev)).sum() for i in range(2))
hist = []
for step in range(3):
    before = tr.flat.flat.clone()
    tr.step([mine], loss_fn)
    torch.cuda.synchronize()
    hist.append(dict(before=before.cpu(), grad=tr.flat.grad.cpu().clone(), param=tr.flat.flat.cpu().clone()))
if not single and not rehearse and not os.environ.get("TF_TEST_BREAK_EDGE"):      # (with an edge removed the ranks may well disagree: that IS the finding)
    chk = tr.flat.flat.double().sum().reshape(1).cpu()
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert lo.item() == hi.item(), (lo.item(), hi.item())
grouped = int(getattr(model.cross_fusion_encoders[0], "_last_desc").groups)
if rank == 0:
    extra = dict(grouped=grouped) if single else dict(grouped=grouped, agreed=tr.layerwise.agreed, collectives=tr.layerwise.collectives, nunits=len(tr.layerwise.units),
                                          keys=[u["key"] for u in tr.layerwise.units], order=tr.layerwise.order,
                                          ranges=[(u["lo"], u["hi"]) for u in tr.layerwise.units], backend=dist.get_backend(),
                                          own_comm=tr.bucket_comm is not None)
    torch.save(dict(hist=hist, **extra), {out!r})
if not single:
    dist.barrier()
    dist.destroy_process_group()
'''


_LEVELS_RAGGED = "[dict(C=16, H=12, W=10, p=2), dict(C=8, H=9, W=9, p=3)]"          # 30 and 9 visual tokens: the level loop
_LEVELS_EQUAL = "[dict(C=16, H=12, W=12, p=2), dict(C=8, H=18, W=18, p=3)]"          # 36 and 36: ONE grouped encoder call


def _run_tree(tmp_path, levels, tag, delay_us=0, break_edge=None, single=True, hw_queues=""):
    """Two ranks sharing the GPU over gloo, then (``single``) one process that sees all four samples.  ``delay_us``: every weight-gradient
    launch of BOTH runs behind a spin (ops.debug_delay_wgrad); ``break_edge``: the two-rank run with one event edge of the reducer removed
    (OrderedRangeReducer._debug_break_edge -- the negative control of the probe)."""
    out2, out1 = str(tmp_path / f"{tag}2.pt"), str(tmp_path / f"{tag}1.pt")
    s2, s1 = tmp_path / f"{tag}_w2.py", tmp_path / f"{tag}_w1.py"
    s2.write_text(_TREE_WORKER.format(root=ROOT, out=out2, levels=levels))
    s1.write_text(_TREE_WORKER.format(root=ROOT, out=out1, levels=levels))
    port = _free_port()
    procs = []
    # (hw_queues -> GPU_MAX_HW_QUEUES: ROCm multiplexes streams onto a few hardware queues -- 4 by default -- and two streams that share
    # one run in submission order, which can HIDE a missing event edge from the probe: see test_the_delay_probe_sees_a_missing_edge)
    # (the delayed runs keep the fused K1 / K9 nodes on the LEVEL streams, round 5's default, so that both stream layouts stay under the
    # two-rank tests; the undelayed runs take round 6's default, the main stream)
    extra = {"TF_TEST_WGRAD_DELAY_US": str(delay_us), "TF_K_LEVEL_STREAMS": "1" if delay_us else "0"}
    if hw_queues:
        extra["GPU_MAX_HW_QUEUES"] = hw_queues
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        if break_edge:
            env["TF_TEST_BREAK_EDGE"] = break_edge
        procs.append(subprocess.Popen([sys.executable, str(s2)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank run hung (a rank issued a different collective sequence?)")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    if not single:
        return torch.load(out2), None
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", TF_TREE_SINGLE="1", **extra)
    r = subprocess.run([sys.executable, str(s1)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return torch.load(out2), torch.load(out1)


def _unit_errors(two, st2, st1):
    return [(k, round(((st2["grad"][lo:hi] - st1["grad"][lo:hi]).norm() / st1["grad"][lo:hi].norm().clamp_min(1e-30)).item(), 4))
            for k, (lo, hi) in zip(two["keys"], two["ranges"])]


def _check_tree(two, one, grouped):
    """rank-agreed order, one collective per unit from step 2 on, gradients of every step equal to one process that sees all samples
    (world 2 sums the two ranks' gradients of their own 2 samples; the single process saw all 4: same sum), same first movement (the
    optimiser uses the MEAN over ranks, grad_scale 1 / world)."""
    if grouped:
        assert two["grouped"] == 2 and one["grouped"] == 2, (two["grouped"], one["grouped"])
    assert two["agreed"] is True
    # 2 levels x (K1 + 2 encoder layers + K9) = 8 units; step 1: one collective, steps 2 and 3: one per unit
    assert two["nunits"] == 8 and two["collectives"] == 1 + 2 * 8, (two["nunits"], two["collectives"], two["keys"])
    for step, (st2, st1) in enumerate(zip(two["hist"], one["hist"])):
        err = ((st2["grad"] - st1["grad"]).norm() / st1["grad"].norm()).item()
        assert err < 5e-3, (step, err, _unit_errors(two, st2, st1))          # (a failure says WHERE: per reducer unit)
    m2 = two["hist"][0]["param"] - two["hist"][0]["before"]
    m1 = one["hist"][0]["param"] - one["hist"][0]["before"]
    assert float(m1.abs().max()) > 0 and ((m2 - m1).norm() / m1.norm()).item() < 2e-2


@pytest.mark.parametrize("delay_us", [0, 3000])
def test_two_ranks_grouped_levels_under_the_ordered_reducer(tmp_path, delay_us):
    """Levels with equal token counts run as ONE grouped encoder call (TfEncoderDesc.groups) -- also under a data-parallel reducer: the
    grouped backward then goes layer by layer and reports each layer of EVERY member encoder to its hook.  With ``delay_us`` every
    producer of a gradient range that is NOT on the chain's stream -- the encoder's weight gradients (side stream), K1 / K9's (level
    streams) -- starts 3 ms late: each consumer (the AccumulateGrad add, the unit's collective, the first step's whole-buffer reduce,
    the optimiser) must then be held by its event edge, or it reads zeros.  DESIGN.md (e) lists writer x edge."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    two, one = _run_tree(tmp_path, _LEVELS_EQUAL, "grp", delay_us=delay_us)
    _check_tree(two, one, grouped=True)


@pytest.mark.parametrize("edge", ["side", "accum"])
def test_the_delay_probe_sees_a_missing_edge(tmp_path, edge):
    """Negative control of the probe above: the same two-rank run with ONE event edge of OrderedRangeReducer removed -- "side": a unit's
    collective no longer waits for the weight-gradient side stream; "accum": no longer for the stream its K1 / K9 gradients were added
    on -- must come out WRONG behind the spins (the delayed producers land after their range was reduced), in one run.
    Whether a missing edge SHOWS depends on how the runtime maps streams onto hardware queues (two streams on one queue run in
    submission order and hide it: with the default 4 queues "side" shows and with 16 it does not, round 5): the control tries queue
    counts until the broken reducer comes out wrong, and the INTACT reducer must then be right under that same mapping -- the
    delayed run above is only as good as this control."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _, one = _run_tree(tmp_path, _LEVELS_EQUAL, "ref", delay_us=0)
    seen = {}
    for q in ("", "16", "8", "2"):
        bad, _ = _run_tree(tmp_path, _LEVELS_EQUAL, f"bad{q}", delay_us=3000, break_edge=edge, single=False, hw_queues=q)
        worst = 0.0
        for st2, st1 in zip(bad["hist"][1:], one["hist"][1:]):           # (step 1 reduces everything after the backward: no unit events)
            worst = max(worst, ((st2["grad"] - st1["grad"]).norm() / st1["grad"].norm()).item())
        seen[q or "default"] = round(worst, 6)
        if worst > 5e-2:
            ok, _ = _run_tree(tmp_path, _LEVELS_EQUAL, f"ok{q}", delay_us=3000, single=False, hw_queues=q)
            _check_tree(ok, one, grouped=True)
            return
    # (which stream lands on which hardware queue is the runtime's business: a box on which no mapping shows the removed edge tells
    # nothing about the product -- the control is then void here, not failed)
    pytest.skip(f"no hardware-queue mapping exposed the removed {edge!r} edge on this box: {seen}")


@pytest.mark.parametrize("delay_us", [0, 3000])
def test_two_ranks_one_gpu_ordered_reducer_on_the_real_wrapper(tmp_path, delay_us):
    """The wrapper (two feature levels on their own HIP streams: patch embedding, 2-layer encoder, back-projection each) under
    FusionTrainStep with world 2: OrderedRangeReducer's CUDA branch -- unit events on the level streams and their wgrad side streams, a
    communication stream, collectives fired in the learnt order from inside the backward -- against ONE process that sees all four
    samples: same gradients (atomics order aside), same parameters after three steps, ranks bit-identical, identical packed-row counts."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    two, one = _run_tree(tmp_path, _LEVELS_RAGGED, "tree", delay_us=delay_us)
    _check_tree(two, one, grouped=False)
    keys = [two["keys"][u] for u in two["order"]]
    assert keys[0].startswith("tokens_to_features.1") and keys[-1].startswith("patches_to_token.0"), keys     # backward order: level 1 first


# ---- the N > 1 step against the REAL backend, on one GPU: a one-rank RCCL process group with a phantom peer ----
def _run_one(tmp_path, template, tag, env_extra, **fmt):
    out = str(tmp_path / f"{tag}.pt")
    script = tmp_path / f"{tag}.py"
    script.write_text(template.format(root=ROOT, out=out, **fmt))
    env = {k: v for k, v in os.environ.items() if not k.startswith("TF_REHEARSE")}
    env.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update({k: v for k, v in env_extra.items() if v != ""})
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return torch.load(out)


def _rehearsal_env(comm, delay_us, break_edge="", hw_queues=""):
    return dict(TF_REHEARSE_COLLECTIVES="1", TF_REHEARSE_PHANTOM_PEERS="1", TF_COMM=comm, TF_TEST_WGRAD_DELAY_US=str(delay_us),
                TF_TEST_BREAK_EDGE=break_edge, GPU_MAX_HW_QUEUES=hw_queues)


def _tree_rehearsal_error(got, one, steps=slice(0, None)):
    """the phantom peer doubles every reduced range: the rehearsal's gradients must be exactly twice the single process's"""
    return max(((st2["grad"] - 2 * st1["grad"]).norm() / (2 * st1["grad"]).norm()).item() for st2, st1 in zip(got["hist"][steps], one["hist"][steps]))


@pytest.mark.parametrize("comm", ["torch", "rccl"])
def test_rccl_rehearsal_of_the_ordered_reducer_with_a_phantom_peer(tmp_path, comm):
    """What the two-rank runs above cannot show: they share one GPU and therefore go over gloo, whose CUDA all-reduce stages through the
    host and synchronises far more than RCCL does.  Here the N > 1 step runs against the real backend -- a ONE-rank "nccl" process group
    (TF_REHEARSE_COLLECTIVES=1), through torch's process group and through the library's own communicator (tf_allreduce_bucket) --
    with asynchronous collectives on the communication stream, every weight gradient 3 ms late, and a phantom peer that doubles each
    range right behind its collective (TF_REHEARSE_PHANTOM_PEERS=1): a gradient written after its range was reduced misses the factor."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    one = _run_one(tmp_path, _TREE_WORKER, "ref", dict(TF_TREE_SINGLE="1"), levels=_LEVELS_EQUAL)
    got = _run_one(tmp_path, _TREE_WORKER, "reh", _rehearsal_env(comm, 3000), levels=_LEVELS_EQUAL)
    assert got["backend"] == "nccl" and got["own_comm"] == (comm == "rccl")
    assert got["agreed"] is True and got["nunits"] == 8 and got["collectives"] == 1 + 2 * 8, (got["nunits"], got["collectives"])
    assert _tree_rehearsal_error(got, one) < 5e-3, _tree_rehearsal_error(got, one)


@pytest.mark.parametrize("comm", ["torch", "rccl"])
def test_rccl_rehearsal_of_the_layerwise_reducer_with_a_phantom_peer(tmp_path, comm):
    """The bare encoder's reducer (bench.py's path) the same way: one all-reduce per layer behind the communication stream, over RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    got = _run_one(tmp_path, _WORKER, "lw", _rehearsal_env(comm, 3000), acc=1)
    assert got["backend"] == "nccl" and got["own_comm"] == (comm == "rccl") and got["collectives"] == 3
    ref = _encoder_reference_grad(got["before"])
    err = ((got["grad"] - 2 * ref).norm() / (2 * ref).norm()).item()
    assert err < 2e-3, err


def test_the_rccl_rehearsal_sees_a_missing_edge(tmp_path):
    """Negative control of the two rehearsals: with the side-stream edge of a reducer removed the delayed weight gradients land after the
    phantom peer's factor, and the run must come out WRONG (under some stream -> hardware-queue mapping; the intact reducer must then be
    right under the same one)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    one = _run_one(tmp_path, _TREE_WORKER, "ref", dict(TF_TREE_SINGLE="1"), levels=_LEVELS_EQUAL)
    seen = {}
    for q in ("", "16", "8", "2"):
        bad = _run_one(tmp_path, _TREE_WORKER, f"bad{q}", _rehearsal_env("torch", 3000, "side", q), levels=_LEVELS_EQUAL)
        worst = _tree_rehearsal_error(bad, one, slice(1, None))         # (step 1 reduces everything after the backward: no unit events)
        seen[q or "default"] = round(worst, 6)
        if worst > 5e-2:
            ok = _run_one(tmp_path, _TREE_WORKER, f"ok{q}", _rehearsal_env("torch", 3000, "", q), levels=_LEVELS_EQUAL)
            assert _tree_rehearsal_error(ok, one) < 5e-3
            return
    pytest.skip(f"no hardware-queue mapping exposed the removed edge on this box: {seen}")


def test_bench_rehearses_the_rccl_path_on_one_gpu():
    """TF_REHEARSE_COLLECTIVES=1 python bench.py: the headline workload with the process group, the layer-wise all-reduces, the rank
    check, the bandwidth probe and the library's own communicator live on ONE GPU -- what the driver's N > 1 runs execute, minus peers."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TF_REHEARSE_COLLECTIVES="1", TF_BENCH_WATCHDOG_S="250", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for comm in ("torch", "rccl"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-census", "--legs", "b4", "--comm", comm,
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert res["n_gpus"] == 1 and res["value"] > 0 and "rehearsal" in res
        ar = res["allreduce"]
        assert ar["backend"] == "nccl" and ar["group_world"] == 1 and ar["collectives_per_step"] == 4 and ar["bytes"] > 7e7
        assert ar["rccl"].get("matches_process_group") is True and ar["rccl"]["world"] == 1, ar["rccl"]
        assert res["rank_sync"]["identical_on_ranks"] == 1
        w = res["wrapper_b4_real_dp"]                            # the real module's data-parallel step over RCCL (one rank)
        assert "error" not in w and w["ms_per_step"] > 0 and w["reducer"] == "OrderedRangeReducer", w
