#!/usr/bin/env python3
"""Forward + backward time of the full CrossFusionBoxWrapper path (4 feature levels: patch-embedding GEMM, patchify, 4-layer
encoder, regroup + fold each) with a stub detector, at the shipped config's geometry: level maps 14p x 14p for p = (4, 4, 2, 1),
C = (256, 512, 1024, 2048), 512 language tokens, d = 768.  A sanity data point (nothing here is a bench line)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from test_gpu_wrapper import StubDetector
from transfusion_amd.modeling.model_factory import get_fusion_model
from transfusion_amd.runner.config import load_fusion_config
from transfusion_amd import _lib as Lb

B, NL, D = int(os.environ.get("B", 8)), 512, int(os.environ.get("D", 768))
dev = torch.device("cuda", 0)
torch.manual_seed(42)
ps, chans = [4, 4, 2, 1], [256, 512, 1024, 2048]
grids = [28, 14, 14, 14] if os.environ.get("REAL") == "1" else [14, 14, 14, 14]        # REAL=1: the reference's FPN geometry (bench leg wrapper_b4_real)
if os.environ.get("GRIDS"):                          # e.g. GRIDS=19,19,19,19: four equal levels with about the real geometry's total token count
    grids = [int(x) for x in os.environ["GRIDS"].split(",")]
shapes = [(g * p, g * p) for g, p in zip(grids, ps)]
fusion = load_fusion_config(os.path.join(ROOT, "transfusion_amd", "runner", "configs", "cross_fusion_config_sym_ego_res50.yml"))
fusion.update({"fpn_features": [0, 1, 2, 3], "replace_fpn_features": True})       # run.narr_fusion.* keys of the experiment YAML
fusion["args"].update({"input_f_size": D})
run_cfg = {"experiment": "egonao", "narr_fusion": fusion, "criterion": {"lm": 0}, "precision": int(os.environ.get("PRECISION", 16)),
           "narration_embeds": {"use": True, "args": {"text_pooling": "slowfast", "strategy": "current", "out_mlp": 0, "size": D,
                                                     "out_dropout": 0.0, "out_tanh": False, "train_ep": 0}}}
model = get_fusion_model(StubDetector(shapes, chans), {}, run_cfg, None).to(dev).train()
g = torch.Generator().manual_seed(1)
feats = [torch.randn(B, c, h, w, generator=g).to(dev).requires_grad_(True) for c, (h, w) in zip(chans, shapes)]
lens = torch.randint(NL // 4, NL + 1, (B,), generator=g).tolist()
lang = [torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=-1).to(dev) for n in lens]

def step():
    model.zero_grad(set_to_none=True)                # as a training loop does (optimizer.zero_grad)
    out = model({"image": feats, "language_f": lang})
    loss = sum(f.float().square().mean() for f in out["features"].values())
    loss.backward()
    return loss

if os.environ.get("FLAT") == "1":
    # forward + backward only, but with the parameters re-homed into flat buffers (what lets the levels run as one grouped encoder
    # call) and gradients accumulated in place: FusionTrainStep's layout without its optimiser
    from transfusion_amd.runner.trainer import FusionTrainStep
    _tr = FusionTrainStep(model, lr=0.0, weight_decay=0.0, grad_clip=None)

    def step():
        _tr.zero_grad()
        for f in feats:
            f.grad = None
        out = model({"image": feats, "language_f": lang})
        loss = sum(f.float().square().mean() for f in out["features"].values())
        loss.backward()
        return loss

if os.environ.get("TRAINER") == "1":
    # the library's own training step around the whole wrapper: flat parameter / gradient buffers, the encoders accumulate straight
    # into .grad (no AccumulateGrad node per parameter), bucketed reduce (no-op on one GPU), clip + fused RAdam -- MORE work per step
    # than the bare forward + backward above, on fewer host operations
    from transfusion_amd.runner.trainer import FusionTrainStep
    trainer = FusionTrainStep(model, lr=1e-4, weight_decay=2e-4, grad_clip=1.0)

    def _loss(m, batch):
        out = m({"image": feats, "language_f": lang})
        fs = list(out["features"].values())
        if all(f.dtype == torch.float32 and f.is_contiguous() and f.numel() % 4 == 0 for f in fs):      # the bench legs' loss: the library's kernels (round 6)
            from transfusion_amd import ops
            return ops.sq_loss([(f, None, 1.0 / f.numel()) for f in fs])
        return sum(f.float().square().mean() for f in fs)

    def step():
        return trainer.step([None], _loss)

    if os.environ.get("GRAPH") == "1":
        # the same step captured once in a HIP graph and replayed (dense rows; fresh dropout masks per replay through the step clock)
        from graph_step import GraphedTrainStep
        for f in feats:
            f.grad = None
        gs = GraphedTrainStep(trainer, None, _loss, warmup=3)
        step = gs.replay

for _ in range(3):
    step()
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    loss = step()
t_host = (time.perf_counter() - t0) / n * 1e3
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"host enqueue {t_host:.2f} ms/step")
print(f"wrapper fwd+bwd, B={B}, 4 levels x 4 layers: {ms:.2f} ms/step ({B / ms * 1e3:.1f} samples/s), loss {float(loss):.4f}")
if os.environ.get("GRAPH") == "1":
    sys.exit(0)                                     # (the launch tracer sees launches, a replay makes none)
if os.environ.get("PROFILE") == "1":
    # where the HOST time of a step goes: cProfile over 10 steps (the backward runs on the autograd engine's thread, which the profiler
    # does not see: it shows as the wall time of loss.backward() on this thread), sorted by own time and by cumulative time
    import cProfile, pstats, gc
    gc.collect(); gc.freeze()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        print(f"---- host profile, 10 steps, by {key} ----")
        pstats.Stats(pr).strip_dirs().sort_stats(key).print_stats(int(os.environ.get("TOP", 45)))
    sys.exit(0)
lib = Lb.load()
Lb.check(lib.tf_trace_start(), "trace")
for _ in range(2):
    step()
cap = 1 << 14
recs = (Lb.TfTraceRecord * cap)()
nrec = lib.tf_trace_stop(ctypes.addressof(recs), cap)
agg = {}
for i in range(min(nrec, cap)):
    r = recs[i]
    a = agg.setdefault(r.name.decode(), [0.0, 0])
    a[0] += r.us; a[1] += 1
tot = sum(v[0] for v in agg.values()) / 2
for name, (us, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", 14))]:
    print(f"  {name:36s} {us / cnt:9.1f} us x{cnt / 2:6.1f} = {us / 2:8.1f} us/step")
print(f"  library kernels (sum of durations) {tot:.0f} us/step")
