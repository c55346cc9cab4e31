"""Soak run of the REAL module: N training steps of the four-level CrossFusionBoxWrapper (real FPN geometry, pass-through detector,
dropout on) on a FIXED batch under FusionTrainStep -- the loss must fall, nothing may become non-finite, the allocator must not grow, no
packed-row error may be pending.  Run once through the ragged grouped call and once with TF_RAGGED_GROUPS=0 (level 0 beside a three-level
group): the two trajectories differ only by their dropout masks.  usage: python tools/soak_wrapper.py [steps] [batch] [lr]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from transfusion_amd.modeling.model_factory import get_fusion_model  # noqa: E402
from transfusion_amd.runner.config import load_fusion_config  # noqa: E402
from transfusion_amd.runner.trainer import FusionTrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-4
dev = torch.device("cuda:0")
D, NL = 768, 512
ps, chans, grids = [4, 4, 2, 1], [256, 512, 1024, 2048], [28, 14, 14, 14]
shapes = [(n * p, n * p) for n, p in zip(grids, ps)]


def run(ragged):
    os.environ["TF_RAGGED_GROUPS"] = "1" if ragged else "0"
    fusion = load_fusion_config(os.path.join(ROOT, "transfusion_amd", "runner", "configs", "cross_fusion_config_sym_ego_res50.yml"))
    fusion.update({"fpn_features": [0, 1, 2, 3], "replace_fpn_features": True})
    fusion["args"].update({"input_f_size": D})
    run_cfg = {"experiment": "egonao", "narr_fusion": fusion, "criterion": {"lm": 0}, "precision": 16,
               "narration_embeds": {"use": True, "args": {"text_pooling": "slowfast", "strategy": "current", "out_mlp": 0, "size": D,
                                                         "out_dropout": 0.0, "out_tanh": False, "train_ep": 0}}}
    torch.manual_seed(42)
    model = get_fusion_model(bench._PassThroughDetector(shapes, chans), {}, run_cfg, None).to(dev).train()
    g = torch.Generator().manual_seed(4242)
    feats = [torch.randn(B, c, h, w, generator=g).to(dev) for c, (h, w) in zip(chans, shapes)]
    targets = [torch.randn(B, c, h, w, generator=g).to(dev) * 0.1 for c, (h, w) in zip(chans, shapes)]
    lens = torch.randint(NL // 4, NL + 1, (B,), generator=g).tolist()
    lang = [torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=-1).to(dev) for n in lens]
    tr = FusionTrainStep(model, lr=lr, weight_decay=2e-4, grad_clip=1.0)

    def loss_fn(m, _):
        out = m({"image": feats, "language_f": lang})
        return sum((out["features"][str(i)].float() - targets[i]).square().mean() for i in range(4))
    first = last = None
    mem0 = mem_half = None
    for i in range(steps):
        loss = tr.step([None], loss_fn)
        if i == 10:
            torch.cuda.synchronize()
            mem0 = torch.cuda.memory_reserved(dev)
        if i % 50 == 0 or i == steps - 1 or i == steps // 2:
            v = float(loss.item())
            if not math.isfinite(v):
                raise SystemExit(f"non-finite loss at step {i}")
            first = v if first is None else first
            last = v
            print(f"  ragged={int(ragged)} step {i:4d} loss {v:.6f}   allocated {torch.cuda.memory_allocated(dev) / 2**20:7.0f} MiB, reserved "
                  f"{torch.cuda.memory_reserved(dev) / 2**20:7.0f} MiB", flush=True)
            if i == steps // 2:
                mem_half = torch.cuda.memory_reserved(dev)
    torch.cuda.synchronize()
    grew = torch.cuda.memory_reserved(dev) - mem_half          # (before the checks below allocate their own temporaries)
    tr.check_errors(sync=True)
    path, groups = model._last_path, int(model.cross_fusion_encoders[0]._last_desc.groups)
    finite = bool(torch.isfinite(tr.flat.flat).all().item())
    # (the pool of reserved blocks settles during the first dozens of steps -- workspaces of forwards whose backward is pending, the
    # collector's rhythm --: what must not happen is growth in the SECOND half of the run)
    ok = finite and last < first and grew <= (64 << 20) and (groups == 4 if ragged else groups in (0, 1))
    print(f"ragged={int(ragged)}: path {path}, groups of encoder 0's last call {groups}; parameters finite {finite}; loss {first:.5f} -> {last:.5f}; "
          f"reserved memory grew by {grew / 2**20:.0f} MiB over the second half (it was {mem0 / 2**20:.0f} MiB at step 10): {'OK' if ok else 'FAIL'}", flush=True)
    del tr, model
    torch.cuda.empty_cache()
    return ok, last


ok1, l1 = run(True)
ok0, l0 = run(False)
same = abs(l1 - l0) <= 0.05 * max(l0, l1)
print(f"final losses {l1:.5f} (ragged) / {l0:.5f} (split): {'within 5 %' if same else 'DIFFER'}")
sys.exit(0 if (ok1 and ok0 and same) else 1)
