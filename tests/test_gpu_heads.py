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
        # the north_star names these outputs element by element ("noun/verb logits, bbox and TTC outputs ... within 1e-3 fp32 / 1e-2
        # bf16"): max-abs against the largest reference magnitude as well, not only the relative L2 norm
        ref_k = torch.from_numpy(g[k]).double()
        assert (out[k].detach().double().cpu() - ref_k).abs().max() < TOL * max(1.0, float(ref_k.abs().max())), k
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


def test_out_of_range_labels_select_nothing_and_are_reported(dev):
    """Labels are range-checked by the kernel, not by host-side min / max syncs: an out-of-range noun or verb label reads no class
    weight / logit / box slot (the result equals the same call with that RoI's terms removed), -100 (torch's ignore_index) is skipped
    silently, anything else is counted and ``ops.check_label_errors()`` raises torch's IndexError for it afterwards."""
    from transfusion_amd import ops
    cfg = HEADS_CASES["heads_v2_bg"]
    _, _, noun, verb, ttc, reg, noun_w, verb_w = make_heads_case(cfg)
    R, Cn, Cv = cfg["R"], cfg["nouns"], cfg["verbs"]
    g = torch.Generator().manual_seed(3)
    cls = (2 * torch.randn(R, Cn + Cv + 1, generator=g)).to(torch.bfloat16).to(dev)
    box = torch.randn(R, 4 * Cn, generator=g).to(torch.bfloat16).to(dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def run(noun_l, verb_l):
        c = cls.clone().requires_grad_(True)
        b = box.clone().requires_grad_(True)
        ttcs = ops.softplus_col(c, Cn + Cv)
        losses = ops.nao_head_losses(c, b, ttcs, Cn, Cv, t(noun_l), t(verb_l), t(ttc), t(reg), t(noun_w), t(verb_w), IGNORE_VERB_IDX_BG, False, False,
                                     0.0, 1.0)
        losses.sum().backward()
        return losses.detach().cpu(), c.grad.detach().cpu().float(), b.grad.detach().cpu().float()

    ops.check_label_errors(sync=True)                       # nothing pending from earlier tests
    base = run(noun, verb)
    ops.check_label_errors(sync=True)                       # in-range labels: no error
    assert all(torch.isfinite(x).all() for x in base)
    # torch's ignore_index: silently skipped
    n2, v2 = noun.copy(), verb.copy()
    rows = [i for i in range(R) if noun[i] > 0 and verb[i] != IGNORE_VERB_IDX_BG][:2]
    n2[rows[0]] = -100
    v2[rows[1]] = -100
    ign = run(n2, v2)
    ops.check_label_errors(sync=True)
    assert all(torch.isfinite(x).all() for x in ign)
    assert float(ign[1][rows[0], :Cn].abs().max()) == 0 and float(ign[2][rows[0]].abs().max()) == 0      # no noun / box gradient for that RoI
    assert float(ign[1][rows[1], Cn:Cn + Cv].abs().max()) == 0
    # far out of range (would be a wild read): same values as the ignored case, plus the error afterwards
    n3, v3 = noun.copy(), verb.copy()
    n3[rows[0]] = 1 << 40
    v3[rows[1]] = Cv + 12345
    bad = run(n3, v3)
    for a, b in zip(bad, ign):
        # (equal up to the order of the fp32 atomics behind the normalisers; gradients are stored in bf16)
        assert float((a - b).abs().max()) <= 1e-2 * float(b.abs().max()) and torch.allclose(bad[0], ign[0], rtol=1e-5, atol=1e-6)
    with pytest.raises(IndexError):
        ops.check_label_errors(sync=True)
    ops.check_label_errors(sync=True)                       # reported once
    # the trainer's entry: FusionTrainStep.step() looks (without waiting) at the start of every step, check_errors(sync=True) waits
    from transfusion_amd.runner.trainer import FusionTrainStep
    run(n3, v3)
    with pytest.raises(IndexError):
        FusionTrainStep.check_errors(sync=True)
    FusionTrainStep.check_errors(sync=True)
    with pytest.raises(RuntimeError):                       # class-weight vector shorter than the class count
        ops.nao_head_losses(cls, box, None, Cn, Cv, t(noun), t(verb), None, t(reg), t(noun_w[:-1]), t(verb_w), IGNORE_VERB_IDX_BG)
