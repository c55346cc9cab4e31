"""SURVEY.md 8f-2 on the MI355X: the NAO RoI heads (box / noun / verb / TTC Linears on the MFMA GEMM) and their losses (one row
kernel each way) against fixtures produced by the reference's own head modules, its own ``box_loss`` and the trainer's criterion
objects (tests/golden/make_golden.py::run_heads_case).  bf16 logits: 1e-2 (north_star), losses 1e-2 relative, gradients 3e-2."""
import os

import numpy as np
import pytest
import torch

from cases import HEADS_CASES, IGNORE_VERB_IDX_BG, make_heads_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("name", ["heads_v1", "heads_v2_bg", "heads_allbg"])
def test_heads_and_losses_against_reference_fixture(dev, golden_dir, name, precision):
    """bf16 compute: north_star's 1e-2 (gradients 3e-2).  fp32 mode (run.precision 32: hi + lo planes through the same GEMM, loss and
    softplus kernels): logits, losses and every gradient within 1e-3 of the reference's fp32 values."""
    from transfusion_amd.modeling.obj_detection.nao_heads import NaoHeadLosses, NaoRoIHeads
    TOL, GTOL, TTC_TOL = (1e-2, 3e-2, 3e-2) if precision == "bf16" else (1e-3, 1e-3, 1e-3)
    cfg = HEADS_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    params, feats, noun, verb, ttc, reg, noun_w, verb_w = make_heads_case(cfg)
    heads = NaoRoIHeads(cfg["repr"], cfg["nouns"], cfg["verbs"]).to(dev)
    assert sorted(heads.state_dict().keys()) == sorted(params.keys())          # the reference's checkpoint keys under roi_heads.
    heads.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    heads.train()
    heads.precision = precision
    crit = NaoHeadLosses(noun_w, verb_w, cfg["verb_bg"], cfg["ttc_bg"], cfg["ttc_bg_val"], cfg["ttc_beta"]).to(dev)
    x = torch.from_numpy(feats).to(dev).requires_grad_(True)
    out = heads(x)
    for k in ("box_regression", "class_logits", "verb_logits"):
        assert tuple(out[k].shape) == g[k].shape and rel(out[k], g[k]) < TOL, k
    assert (out["ttcs"].detach().cpu() - torch.from_numpy(g["ttcs"])).abs().max() < TTC_TOL
    h = cfg["R"] // 2
    t = lambda a: torch.from_numpy(a).to(dev)
    # per-image lists, as roi_heads.select_training_samples hands them over (two images)
    losses = crit(out, [t(noun[:h]), t(noun[h:])], [t(verb[:h]), t(verb[h:])], [t(ttc[:h]), t(ttc[h:])], [t(reg[:h]), t(reg[h:])])
    got = torch.stack([losses["bbox_loss"], losses["noun_loss"], losses["verb_loss"], losses["ttc_loss"]]).detach().cpu().double().numpy()
    assert np.abs(got - g["losses"]).max() < TOL * (1 + np.abs(g["losses"]).max()), (got, g["losses"])
    if name == "heads_allbg":
        assert got[0] == 0 and got[2] == 0 and got[3] == 0       # no positive RoI: box / verb / TTC terms are exactly zero
    total = sum(float(c) * losses[k] for c, k in zip(g["cot"], ("bbox_loss", "noun_loss", "verb_loss", "ttc_loss")))
    total.backward()
    assert rel(x.grad, g["grad_feats"]) < GTOL
    named = dict(heads.named_parameters())
    for k in params:
        ref = g["gradp/" + k]
        if np.abs(ref).max() == 0:
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0, k
        else:
            assert rel(named[k].grad, ref) < GTOL, k


def test_heads_loss_kernel_against_oracle_on_given_logits(dev):
    """The row kernel alone, fed bf16 logits: losses and logit gradients against the oracle evaluated on the SAME (bf16-rounded)
    logits, i.e. without the GEMM's rounding in the way: 1e-4 on the losses, 2e-2 on the bf16-stored gradients; class-weighted
    normalisers, verb background dropped / kept, TTC selection, the quadratic and the linear zone of both smooth-L1 terms."""
    from oracle import fusion_oracle as O
    from transfusion_amd import ops
    cfg = HEADS_CASES["heads_v2_bg"]
    _, _, noun, verb, ttc, reg, noun_w, verb_w = make_heads_case(cfg)
    R, Cn, Cv = cfg["R"], cfg["nouns"], cfg["verbs"]
    g = torch.Generator().manual_seed(9)
    cls = (2 * torch.randn(R, Cn + Cv + 1, generator=g)).to(torch.bfloat16)
    box = torch.randn(R, 4 * Cn, generator=g).to(torch.bfloat16)
    for verb_bg, ttc_bg in ((False, False), (True, True)):
        cd = torch.zeros(R, 256, dtype=torch.bfloat16, device=dev)[:, :Cn + Cv + 1]
        cd.copy_(cls)
        cd.requires_grad_(True)
        bd = box.to(dev).requires_grad_(True)
        ttcs = ops.softplus_col(cd, Cn + Cv)
        t = lambda a: torch.from_numpy(a).to(dev)
        losses = ops.nao_head_losses(cd, bd, ttcs, Cn, Cv, t(noun), t(verb), t(ttc), t(reg), t(noun_w), t(verb_w), IGNORE_VERB_IDX_BG, verb_bg, ttc_bg,
                                     1.5, 0.5)
        cot = torch.tensor([0.7, 1.3, 0.9, 1.1])
        (losses * cot.to(dev)).sum().backward()
        c64 = cls.double().requires_grad_(True)
        b64 = box.double().requires_grad_(True)
        z = c64[:, Cn + Cv]
        out = {"class_logits": c64[:, :Cn], "verb_logits": c64[:, Cn:Cn + Cv], "box_regression": b64,
               "ttcs": torch.where(z > 20, z, torch.log1p(torch.exp(z)))}
        ref = O.nao_losses(out, torch.from_numpy(noun), torch.from_numpy(verb), torch.from_numpy(ttc).double(), torch.from_numpy(reg).double(),
                           torch.from_numpy(noun_w).double(), torch.from_numpy(verb_w).double(), IGNORE_VERB_IDX_BG, verb_bg, ttc_bg, 1.5, 0.5)
        refv = torch.stack([ref["box"], ref["noun"], ref["verb"], ref["ttc"]])
        assert (losses.cpu().double() - refv.detach()).abs().max() < 1e-4 * (1 + refv.abs().max())
        (refv * cot.double()).sum().backward()
        assert rel(cd.grad, c64.grad) < 2e-2 and rel(bd.grad, b64.grad) < 2e-2
        assert (ttcs.cpu().double() - out["ttcs"].detach()).abs().max() < 1e-5
