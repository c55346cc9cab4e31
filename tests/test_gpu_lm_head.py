"""Language auxiliary head (reference lm_layers.py) on the HIP path: fused pool + LayerNorm + GELU kernel and the
MFMA Linear heads, against the golden fixtures made by the reference itself and against the CPU oracle at a real size.
Tolerances: the pooling kernel is fp32 (1e-5); logits and gradients pass through bf16 GEMMs (1e-2 / 3e-2 relative)."""
import os

import numpy as np
import pytest
import torch

from cases import LM_CASES, make_lm_case

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _build(cfg, params, dev):
    from transfusion_amd.modeling.cross_fusion.ego_fusion import lm_layers as LM
    pooling = {"type": cfg["pool"], "ln": cfg["ln"], "repr_size": cfg["repr_size"]}
    clzz = {False: LM.PoolPredictor, True: LM.MultiPoolPredictor, "sep": LM.MultiPoolPredictorSep}[cfg["multi"]]
    head = clzz(pooling, cfg["d"], cfg["nouns"], cfg["verbs"])
    head.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)    # reference names
    return head.to(dev).train()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("name", list(LM_CASES))
def test_lm_head_golden(golden_dir, name, precision):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = torch.device("cuda:0")
    TOL, GTOL = (1e-2, 3e-2) if precision == "bf16" else (1e-3, 1e-3)      # fp32 mode: the Linears take hi + lo planes
    cfg = LM_CASES[name]
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    params, tokens, att, cot_noun, cot_verb = make_lm_case(cfg)
    head = _build(cfg, params, dev)
    for m in head.modules():
        if hasattr(m, "precision"):
            m.precision = precision
    toks = [torch.from_numpy(t).to(dev).requires_grad_(True) for t in tokens]
    out = head(toks if cfg["multi"] else toks[0], torch.from_numpy(att).to(dev))
    assert out["noun_logits"].shape == g["noun_logits"].shape
    assert rel(out["noun_logits"], g["noun_logits"]) < TOL
    loss = (out["noun_logits"].float() * torch.from_numpy(cot_noun).to(dev)).sum()
    if cfg["verbs"]:
        assert rel(out["verb_logits"], g["verb_logits"]) < TOL
        loss = loss + (out["verb_logits"].float() * torch.from_numpy(cot_verb).to(dev)).sum()
    else:
        assert out["verb_logits"] is None
    loss.backward()
    for i, t in enumerate(toks):
        assert rel(t.grad, g[f"grad_tokens/{i}"]) < GTOL, i
        assert t.grad[~torch.from_numpy(att).to(dev)].abs().max().item() == 0.0       # padded rows: exactly zero
    for k, p in head.named_parameters():
        assert rel(p.grad, g["gradp/" + k]) < GTOL, k


@pytest.mark.parametrize("pool", ["mean", "max"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lm_pool_kernel_exact(pool, dtype):
    """The pooling stage alone is fp32 arithmetic: compare with the oracle's formula at 1e-5, index-exact arg-max routing."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from oracle import fusion_oracle as O
    from transfusion_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)
    B, Lt, d = 5, 33, 768
    x = torch.randn(B, Lt, d, generator=gen).to(dtype)
    att = torch.zeros(B, Lt, dtype=torch.bool)
    for b, n in enumerate([33, 1, 17, 0, 32]):           # includes a fully padded sample
        att[b, :n] = True
    w, bb = 1 + 0.1 * torch.randn(d, generator=gen), 0.1 * torch.randn(d, generator=gen)
    cot = torch.randn(B, d, generator=gen)
    xr = x.float().clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), bb.clone().requires_grad_(True)
    xm = xr * att.unsqueeze(2)
    fr = O.gelu(O.layer_norm(xm.max(dim=1)[0] if pool == "max" else xm.sum(1) / Lt, wr, br))
    (fr * cot).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), bb.to(dev).requires_grad_(True)
    f = ops.lm_pool(xd, att.to(dev), pool, wd, bd, 1e-5, gelu=True)
    assert f.dtype == torch.float32
    assert (f.cpu() - fr.detach()).abs().max().item() < 2e-5
    (f * cot.to(dev)).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 1e-2       # bf16 tokens: dx is stored in bf16
    assert rel(xd.grad, xr.grad) < tol
    assert rel(wd.grad, wr.grad) < 2e-5 and rel(bd.grad, br.grad) < 2e-5
    assert xd.grad[~att.to(dev)].abs().max().item() == 0.0


def test_lm_head_real_size_vs_oracle():
    """Ego4D class counts (87 nouns / 74 verbs: not multiples of 8) at the real token width, three FPN scales, shared head."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from oracle import fusion_oracle as O
    dev = torch.device("cuda:0")
    cfg = dict(B=32, Nl=20, d=256, pool="mean", ln=True, repr_size=512, nouns=87, verbs=74, multi=True, scales=3,
               mask_lens=[(7 * b) % 20 + 1 for b in range(32)], seed=311)
    params, tokens, att, cot_noun, cot_verb = make_lm_case(cfg)
    head = _build(cfg, params, dev)
    toks = [torch.from_numpy(t).to(dev).requires_grad_(True) for t in tokens]
    out = head(toks, torch.from_numpy(att).to(dev))
    (out["noun_logits"] * torch.from_numpy(cot_noun).to(dev)).sum().backward(retain_graph=True)
    (out["verb_logits"] * torch.from_numpy(cot_verb).to(dev)).sum().backward()
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    tr = [torch.from_numpy(t).requires_grad_(True) for t in tokens]
    ref = O.lm_multi_pool_predictor(sd, tr, torch.from_numpy(att), "mean")
    ((ref["noun_logits"] * torch.from_numpy(cot_noun)).sum() + (ref["verb_logits"] * torch.from_numpy(cot_verb)).sum()).backward()
    assert out["noun_logits"].shape == (32, 87) and out["verb_logits"].shape == (32, 74)
    assert rel(out["noun_logits"], ref["noun_logits"]) < 1e-2 and rel(out["verb_logits"], ref["verb_logits"]) < 1e-2
    for t, r in zip(toks, tr):
        assert rel(t.grad, r.grad) < 3e-2
    for k, p in head.named_parameters():
        assert rel(p.grad, sd[k].grad) < 3e-2, k
