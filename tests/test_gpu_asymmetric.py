"""SURVEY.md 8f-4 on the MI355X: cross attention with its own query set (Nq != Nk), the QKVEncoder layer and the asymmetric fusion
encoder (``type: asymmetric``) against fixtures derived from the reference's own pieces (tests/golden/make_golden.py: the reference
QKVEncoder instances driven through their forward statements with the reference's vendored three-value attention; the classes themselves
cannot run as shipped).  bf16 compute: outputs 1e-2, gradients 3e-2 (relative L2)."""
import math
import os

import numpy as np
import pytest
import torch

from cases import ASYM_CASES, QKV_CASES, make_asym_case, make_qkv_case

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


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("hd,Sq,Sk,p", [(64, 70, 200, 0.0), (192, 150, 333, 0.15), (32, 300, 90, 0.1)])
def test_cross_attention_kernels_against_fp64(dev, hd, Sq, Sk, p, split):
    """tf_attn_fwd / tf_attn_bwd with TfAttnArgs.q (own query rows, Sq != Sk), bf16 and fp32-accuracy (lo planes) kernels, key padding
    mask, dropout bits replayed, LSE, dQ / dK / dV against fp64 attention on the same operand values."""
    from transfusion_amd import _lib as L, ops
    B, H = 2, 2
    g = torch.Generator().manual_seed(hd + Sq)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.7)
    q32, kv32, do32 = mk(B * Sq, H * hd), mk(B * Sk, 3 * H * hd), mk(B * Sq, H * hd)
    planes = lambda t: ops.split_planes(t.to(dev)) if split else (t.to(dev).to(torch.bfloat16), None)
    qh, ql = planes(q32)
    kh, kl = planes(kv32)
    dh, dl = planes(do32)
    join = lambda h_, l_: h_.double().cpu() + (0 if l_ is None else l_.double().cpu())
    km = torch.zeros(B, Sk, dtype=torch.uint8)
    km[0, Sk - 29:] = 1
    bf = torch.bfloat16
    z = lambda *s: torch.zeros(*s, dtype=bf, device=dev)
    oh, ol, gqh, gql, gkh, gkl = z(B * Sq, H * hd), z(B * Sq, H * hd), z(B * Sq, H * hd), z(B * Sq, H * hd), z(B * Sk, 3 * H * hd), z(B * Sk, 3 * H * hd)
    lse, delta = torch.empty(B * H * Sq, device=dev), torch.empty(B * H * Sq, device=dev)
    drop = ops.drop_params(p, 13, 5)
    bits = None
    if p > 0:
        bits = torch.empty(B * H * Sq * ((Sk + 63) // 64), dtype=torch.int64, device=dev)
        L.check(L.load().tf_attn_dropmask_rows(L.ptr(bits), B * H * Sq, Sk, drop[1], drop[0], ops._stream()), "dropmask")
    kmd = km.to(dev)
    lo = (lambda t: L.ptr(t)) if split else (lambda t: 0)
    a = L.TfAttnArgs(qkv=L.ptr(kh), qkv_lo=lo(kl), ld_qkv=3 * H * hd, out=L.ptr(oh), out_lo=lo(ol), ld_out=H * hd, lse=L.ptr(lse), key_mask=L.ptr(kmd),
                     B=B, S=Sk, H=H, HDP=hd, scale=1 / math.sqrt(hd), drop_thr=drop[0], drop_key=drop[1], drop_scale=drop[2], drop_bits=L.ptr(bits),
                     dout=L.ptr(dh), dout_lo=lo(dl), ld_dout=H * hd, dqkv=L.ptr(gkh), dqkv_lo=lo(gkl), ld_dqkv=3 * H * hd, delta=L.ptr(delta),
                     q=L.ptr(qh), q_lo=lo(ql), ld_q=H * hd, Sq=Sq, dq=L.ptr(gqh), dq_lo=lo(gql), ld_dq=H * hd)
    L.call("tf_attn_fwd", a, ops._stream())
    L.call("tf_attn_bwd", a, ops._stream())
    torch.cuda.synchronize()
    keep = torch.ones(B, H, Sq, Sk, dtype=torch.float64)
    if p > 0:
        keep = ops.dropout_mask(B * H * Sq * Sk, p, 13, 5, dev).cpu().view(B, H, Sq, Sk).double() * drop[2]
    qx = join(qh, ql).view(B, Sq, H, hd).permute(0, 2, 1, 3).requires_grad_(True)
    kvx = join(kh, kl).view(B, Sk, 3, H, hd).requires_grad_(True)
    k, v = kvx[:, :, 1].permute(0, 2, 1, 3), kvx[:, :, 2].permute(0, 2, 1, 3)
    sc = ((qx / math.sqrt(hd)) @ k.transpose(-1, -2)).masked_fill(km.bool().view(B, 1, 1, Sk), float("-inf"))
    o = ((torch.softmax(sc, -1) * keep) @ v).permute(0, 2, 1, 3).reshape(B * Sq, H * hd)
    o.backward(join(dh, dl))
    tol_o, tol_g = (1e-4, 1e-4) if split else (8e-3, 2e-2)
    assert rel(join(oh, ol), o.detach()) < tol_o
    assert (lse.cpu().view(B, H, Sq).double() - torch.logsumexp(sc, -1).detach() * math.log2(math.e)).abs().max() < (1e-4 if split else 2e-2)
    assert rel(join(gqh, gql), qx.grad.permute(0, 2, 1, 3).reshape(B * Sq, H * hd)) < tol_g
    gkv = kvx.grad.reshape(B * Sk, 3 * H * hd)
    got = join(gkh, gkl)
    assert rel(got[:, H * hd:2 * H * hd], gkv[:, H * hd:2 * H * hd]) < tol_g and rel(got[:, 2 * H * hd:], gkv[:, 2 * H * hd:]) < tol_g
    assert float(got[:, :H * hd].abs().max()) == 0          # the Q third of dqkv is untouched: the query gradient went to dq


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("name", ["qkv_layer", "qkv_layer_gelu_hd18"])
def test_qkv_encoder_layer_against_reference_fixture(dev, golden_dir, name, precision):
    from transfusion_amd.modeling.cross_fusion.cross_qkv_layers import QKVEncoder
    cfg = QKV_CASES[name]
    ftol, itol = (1e-2, 3e-2) if precision == "bf16" else (1e-3, 1e-3)
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    params, q, kv, mask, cot = make_qkv_case(cfg)
    layer = QKVEncoder(cfg["d"], cfg["d"], cfg["h"], dim_feedforward=cfg["ff"], dropout=0.0, activation=cfg["activ"])
    assert sorted(layer.state_dict().keys()) == sorted(str(k) for k in g["state_dict_keys"])
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    layer = layer.to(dev).train()
    layer.precision = precision
    tq, tkv = torch.from_numpy(q).to(dev).requires_grad_(True), torch.from_numpy(kv).to(dev).requires_grad_(True)
    out, att, vs = layer(tq, tkv, tkv, src_key_padding_mask=None if mask is None else torch.from_numpy(mask).to(dev))
    assert att is None and vs is None
    assert rel(out, g["out"]) < ftol
    (out * torch.from_numpy(cot).to(dev)).sum().backward()
    assert rel(tq.grad, g["grad_q"]) < itol and rel(tkv.grad, g["grad_kv"]) < itol
    gtol = 1e-3 if precision == "fp32" else (1.5e-1 if cfg["activ"] == "relu" else 3e-2)   # bf16: ReLU's step derivative at toy width
    for k, p in layer.named_parameters():
        assert rel(p.grad, g["gradp/" + k]) < gtol, k


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_asymmetric_encoder_against_reference_fixture(dev, golden_dir, precision):
    """``type: asymmetric`` through the wrapper's registry: three visual and two language cross-attention layers over the concatenated
    tokens, the reference's layer schedule, outputs and every gradient against the fixture."""
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_wrapper import get_cross_box_encoder
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    cfg = ASYM_CASES["asym_small"]
    g = np.load(os.path.join(golden_dir, "asym_small.npz"))
    params, x, lang, cv, cl = make_asym_case(cfg)
    clzz = get_cross_box_encoder("asymmetric", class_token_only=False)
    enc = clzz(no_patches=8192, pos_embedding_layer=PositionalEmbeddingLayer("sin1d", 8192, cfg["d"]), patch_dropout=0.0, input_f_size=cfg["d"],
               vis_layers=cfg["vis_layers"], lang_layers=cfg["lang_layers"], num_heads=cfg["h"], fforward_multiplier=cfg["ff_mult"], vis_dropout=0.0,
               lang_dropout=0.0, back_to_img_fn="regroup", activ_f=cfg["activ"])
    missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    assert set(missing) == {"padding_mask", "pos_embedding_layer.pos_embedding", "heatmap_token"} and not unexpected
    enc = enc.to(dev).train()
    for m in enc.modules():
        if isinstance(getattr(m, "precision", None), str):
            m.precision = precision             # what CrossFusionBoxWrapper.set_precision does
    ftol, itol, ptol = (1e-2, 5e-2, 1.5e-1) if precision == "bf16" else (1e-3, 1e-3, 1e-3)
    tx, tl = torch.from_numpy(x).to(dev).requires_grad_(True), torch.from_numpy(lang).to(dev).requires_grad_(True)
    pad = torch.zeros(cfg["B"], cfg["Nl"], dtype=torch.bool, device=dev)
    pad[1, 4:] = True                          # the reference builds a padding mask and never applies it: it must change nothing
    vis, lo, att, _ = enc(tx, tl, pad)
    assert att is None and rel(vis, g["vis"]) < ftol and rel(lo, g["lang"]) < ftol
    ((vis * torch.from_numpy(cv).to(dev)).sum() + (lo * torch.from_numpy(cl).to(dev)).sum()).backward()
    assert rel(tx.grad, g["grad_x"]) < itol and rel(tl.grad, g["grad_lang"]) < itol
    worst = 0.0
    for k, p in enc.named_parameters():
        if "gradp/" + k in g and np.abs(g["gradp/" + k]).max() > 0:
            worst = max(worst, rel(p.grad, g["gradp/" + k]))
    assert worst < ptol                         # bf16: ReLU, d = 32, five stacked layers
    assert enc.heatmap_token.grad is None
