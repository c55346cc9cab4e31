"""Randomised shape sweep of the encoder against the oracle (seeded, so every run tests the same cases): widths whose head dim needs
padding, one-token language inputs, single samples, sequences shorter than one attention tile and ones that straddle several, odd
layer counts, both activations, fully padded samples, both arithmetic modes.  The golden fixtures pin the oracle to the reference at a
few shapes; this walks the HIP path's tails, clamps and kernel-selection thresholds between them."""
import numpy as np
import pytest
import torch

from cases import make_encoder_inputs, make_encoder_params

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _cases():
    rs = np.random.RandomState(20261004)
    out = []
    for i in range(14):
        h = int(rs.choice([1, 2, 4, 8]))
        hd = int(rs.choice([8, 18, 24, 32, 40, 64, 96]))
        d = h * hd
        if d % 8 or d > 520:
            hd = 32
            d = h * hd
        B = int(rs.randint(1, 4))
        Nv = int(rs.choice([1, 7, 16, 49, 63, 64, 65, 130]))
        Nl = int(rs.choice([1, 2, 9, 33, 64, 100]))
        L = int(rs.randint(1, 4))
        lens = [int(rs.randint(0, Nl + 1)) for _ in range(B)]
        if i % 5 == 0:
            lens = None                                     # no mask passed at all
        elif max(lens) == 0:
            lens[0] = Nl
        out.append(dict(B=B, Nv=Nv, Nl=Nl, d=d, h=h, L=L, mask_lens=lens, seed=900 + i, activ="relu" if i % 3 == 0 else "gelu"))
    return out


CASES = _cases()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("idx", range(len(CASES)))
def test_random_shape_against_oracle(idx, precision):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from oracle import fusion_oracle as O
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    cfg = CASES[idx]
    dev = torch.device("cuda:0")
    d, L, h = cfg["d"], cfg["L"], cfg["h"]
    enc = CrossTransformerModuleBox(no_patches=8192, pos_embedding_layer=PositionalEmbeddingLayer("sin1d", 8192, d), lang_pos_embedding=None,
                                    num_layers=L, patch_dropout=0.0, num_heads=h, fforward_multiplier=2, token_dropout=0.0,
                                    back_to_img_fn="regroup", activ_f=cfg["activ"], final_norm="ln", input_f_size=d)
    params = make_encoder_params(cfg["seed"], d, L)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    enc = enc.to(dev).train()
    enc.precision = precision
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], d, cfg["mask_lens"])
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    xd, ld = t(x).requires_grad_(True), t(lang).requires_grad_(True)
    v, lo, _, _ = enc(xd, ld, t(mask))
    ((v * t(gv)).sum() + (lo * t(gl)).sum()).backward()
    sd = {k: torch.from_numpy(p).clone().requires_grad_(True) for k, p in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, d)
    xr, lr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
    vr, lor = O.encoder_forward(sd, xr, lr, None if mask is None else torch.from_numpy(mask), h, L, activation=cfg["activ"])
    ((vr * torch.from_numpy(gv)).sum() + (lor * torch.from_numpy(gl)).sum()).backward()
    if precision == "fp32":
        ftol = gtol = 1e-3
    else:
        ftol = 1e-2
        gtol = 1.5e-1 if cfg["activ"] == "relu" else 3e-2     # ReLU's step derivative at toy widths (see enc_relu)
    valid = torch.ones(cfg["B"], cfg["Nl"], dtype=torch.bool) if mask is None else ~torch.from_numpy(mask)
    assert torch.isfinite(v).all() and torch.isfinite(lo).all()
    assert rel(v, vr.detach()) < ftol, cfg
    # language outputs: every row is defined (padded queries still attend to the valid keys), compare all of them
    assert rel(lo, lor.detach()) < ftol, cfg
    assert rel(xd.grad, xr.grad) < gtol, cfg
    if valid.any():
        assert rel(ld.grad, lr.grad) < gtol, cfg
    for k, p in enc.named_parameters():
        if k in sd and sd[k].grad is not None and float(sd[k].grad.abs().max()) > 0:
            assert rel(p.grad, sd[k].grad) < gtol, (k, cfg)
