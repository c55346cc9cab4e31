"""Encoder-level parity on the MI355X: transfusion_amd's CrossTransformerModuleBox (HIP, bf16 compute)
against (a) the committed golden fixtures produced by the reference itself and (b) the CPU oracle.

Tolerance (BASELINE.json north_star): outputs within 1e-2 for bf16 compute.  Asserted as relative L2
error <= 1e-2 for forward outputs (measured 4.9e-3 .. 6.9e-3 on the fixtures) and <= 2e-2 for gradients
(measured: inputs <= 1.1e-2, worst parameter <= 1.2e-2).  Where that error comes from (tests/diag_fp32.py budget):
against the oracle evaluated on bf16-ROUNDED parameters the numbers barely move (6.9e-3 -> 6.0e-3), i.e. it is
the bf16 storage of every intermediate activation, not operand quantisation; an element of an O(1..4)
output stored in bf16 carries up to 2^-8 * 4 = 1.6e-2 of rounding by itself, so the elementwise bound is 5e-2
(measured 3.1e-2 .. 3.9e-2).  The elementwise 1e-3 statement of the north_star is the fp32-accuracy mode's:
tests/test_gpu_fp32_mode.py asserts it, max-abs included, on the same fixtures."""
import os

import numpy as np
import pytest
import torch

from cases import ENCODER_CASES, make_encoder_inputs, make_encoder_params

pytestmark = pytest.mark.gpu

FWD_TOL, GRAD_TOL = 1e-2, 2e-2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def build(cfg, dev, p_tok=0.0, p_patch=0.0):
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    pe = PositionalEmbeddingLayer("sin1d", 8192, cfg["d"])
    enc = CrossTransformerModuleBox(no_patches=8192, pos_embedding_layer=pe, lang_pos_embedding=None, num_layers=cfg["L"],
                                    patch_dropout=p_patch, num_heads=cfg["h"], fforward_multiplier=2, token_dropout=p_tok,
                                    back_to_img_fn="regroup", activ_f="gelu", final_norm="ln", input_f_size=cfg["d"])
    params = make_encoder_params(cfg["seed"], cfg["d"], cfg["L"])
    missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    assert set(missing) == {"padding_mask", "pos_embedding_layer.pos_embedding"} and not unexpected
    return enc.to(dev), params


@pytest.mark.parametrize("name", ["enc_small", "enc_hd18", "enc_nomask"])
def test_golden_small(dev, golden_dir, name):
    cfg = ENCODER_CASES[name]
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    enc, params = build(cfg, dev)
    enc.train()     # p = 0: every row computed, as in the fixture's train_* entries
    x = torch.from_numpy(g["in_x"]).to(dev).requires_grad_(True)
    lang = torch.from_numpy(g["in_lang"]).to(dev).requires_grad_(True)
    mask = torch.from_numpy(g["in_mask"]).to(dev) if "in_mask" in g else None
    vis, lo, att, _ = enc(x, lang, mask)
    assert att is None
    valid = np.ones(lang.shape[:2], bool) if mask is None else ~g["in_mask"]
    e_v, e_l = rel(vis, g["train_vis"]), rel(lo.detach().cpu().numpy()[valid], g["train_lang"][valid])
    assert e_v < FWD_TOL and e_l < FWD_TOL, (e_v, e_l)
    assert (vis.detach().cpu() - torch.from_numpy(g["train_vis"])).abs().max() < 5e-2
    loss = (vis * torch.from_numpy(g["cot_vis"]).to(dev)).sum() + (lo * torch.from_numpy(g["cot_lang"]).to(dev)).sum()
    loss.backward()
    assert rel(x.grad, g["grad_x"]) < GRAD_TOL
    assert rel(lang.grad, g["grad_lang"]) < GRAD_TOL
    worst = 0.0
    for k, p in enc.named_parameters():
        if "gradp/" + k in g:
            e = rel(p.grad, g["gradp/" + k])
            worst = max(worst, e)
            assert e < GRAD_TOL, (k, e)
    assert enc.heatmap_token.grad is None
    # eval mode (no dropout, no grad) gives the same numbers as the fixture's eval entries on unpadded rows
    enc.eval()
    with torch.no_grad():
        v2, l2, _, _ = enc(x.detach(), lang.detach(), mask)
    assert rel(v2, g["eval_vis"]) < FWD_TOL
    assert rel(l2.cpu().numpy()[valid], g["eval_lang"][valid]) < FWD_TOL


@pytest.mark.parametrize("name", ["enc_d768", "enc_d712", "enc_d896"])
def test_golden_real_width(dev, golden_dir, name):
    """d = 768 and the reference's TRUE widths: 712 (Ego4Dv1: head dim 178, padded to 192 inside the weight shadows) and 896
    (Ego4Dv2: head dim 224) -- reference-generated fixtures, outputs and every gradient."""
    cfg = ENCODER_CASES[name]
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    enc, _ = build(cfg, dev)
    enc.train()
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    x = torch.from_numpy(x).to(dev).requires_grad_(True)
    lang = torch.from_numpy(lang).to(dev).requires_grad_(True)
    vis, lo, _, _ = enc(x, lang, torch.from_numpy(mask).to(dev))
    assert rel(vis[:, ::14], g["train_vis_rows"]) < FWD_TOL
    assert rel(lo[:, ::8], g["train_lang_rows"]) < FWD_TOL
    ((vis * torch.from_numpy(gv).to(dev)).sum() + (lo * torch.from_numpy(gl).to(dev)).sum()).backward()
    assert rel(x.grad[:, ::14], g["grad_x_rows"]) < GRAD_TOL
    for k, p in enc.named_parameters():
        if "gradp_head/" + k in g:
            assert rel(p.grad.reshape(-1)[:256], g["gradp_head/" + k]) < 5e-2, k
            assert abs(p.grad.double().abs().sum().item() - float(g["gradp_abs/" + k])) < 3e-2 * float(g["gradp_abs/" + k]), k


def test_dropout_replay_against_oracle(dev):
    """Training mode with dropout ON: the device masks of every site are exported through the C ABI and
    replayed in the oracle, so forward AND gradients are compared under the same random stream."""
    from oracle import fusion_oracle as O
    from transfusion_amd import ops
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import SITE_PATCH, site_of
    cfg = dict(B=2, Nv=24, Nl=40, d=64, h=4, L=2, mask_lens=[25, 40], seed=77)
    p_tok, p_patch = 0.15, 0.1
    enc, params = build(cfg, dev, p_tok, p_patch)
    enc.train()
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    ld = torch.from_numpy(lang).to(dev).requires_grad_(True)
    vis, lo, _, _ = enc(xd, ld, torch.from_numpy(mask).to(dev))
    ((vis * torch.from_numpy(gv).to(dev)).sum() + (lo * torch.from_numpy(gl).to(dev)).sum()).backward()
    seed = enc._last_seed
    B, Nv, Nl, d, H, L = cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["h"], cfg["L"]
    S, M = Nv + Nl, B * (Nv + Nl)
    hdp, dp, ffp = 32, 128, 128          # plan for d=64,h=4 (hd 16 -> 32), ff = 128
    masks = {}
    mk = lambda n, p, site: ops.dropout_mask(n, p, seed, site, dev).cpu()
    masks["patch"] = mk(M * dp, p_patch, SITE_PATCH).view(B, S, dp)[:, :Nv, :d].float()
    for l in range(L):
        pre = f"t_encoder.layers.{l}."
        masks[pre + "attn"] = mk(B * H * S * S, p_tok, site_of(l, 1)).view(B, H, S, S).float()
        masks[pre + "dropout1"] = mk(M * dp, p_tok, site_of(l, 2)).view(B, S, dp)[..., :d].float()
        masks[pre + "dropout"] = mk(M * ffp, p_tok, site_of(l, 3)).view(B, S, ffp)[..., : 2 * d].float()
        masks[pre + "dropout2"] = mk(M * dp, p_tok, site_of(l, 4)).view(B, S, dp)[..., :d].float()
    keep_rate = masks["t_encoder.layers.0.attn"].mean().item()
    assert abs(keep_rate - (1 - p_tok)) < 0.01
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, d)
    xr, lr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
    v_ref, l_ref = O.encoder_forward(sd, xr, lr, torch.from_numpy(mask), H, L, masks=masks, token_dropout=p_tok, patch_dropout=p_patch)
    ((v_ref * torch.from_numpy(gv)).sum() + (l_ref * torch.from_numpy(gl)).sum()).backward()
    valid = ~mask
    assert rel(vis, v_ref.detach()) < FWD_TOL
    assert rel(lo.detach().cpu()[valid], l_ref.detach()[valid]) < FWD_TOL
    assert rel(xd.grad, xr.grad) < GRAD_TOL and rel(ld.grad, lr.grad) < GRAD_TOL
    for k, p in enc.named_parameters():
        if k in sd and sd[k].grad is not None:
            assert rel(p.grad, sd[k].grad) < GRAD_TOL, k


def test_full_size_properties(dev):
    """BASELINE shape (B=32 x [196 + 512] tokens, d=768, 4 layers): size-independent properties.
    (1) batch independence: samples 0..3 of the full batch equal a separate B=4 run bit-for-bit in eval;
    (2) the oracle agrees on a 2-sample slice; (3) gradients of the full batch equal the sum over two
    half batches (atomics order only); (4) padded language rows never influence visual outputs."""
    from oracle import fusion_oracle as O
    cfg = dict(B=32, Nv=196, Nl=512, d=768, h=4, L=4, seed=9)
    torch.manual_seed(0)
    enc, _ = build(dict(cfg, seed=5), dev)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(cfg["B"], cfg["Nv"], cfg["d"], generator=g)
    lang = torch.nn.functional.normalize(torch.randn(cfg["B"], cfg["Nl"], cfg["d"], generator=g), dim=-1)
    lens = torch.randint(cfg["Nl"] // 4, cfg["Nl"] + 1, (cfg["B"],), generator=g)
    mask = torch.arange(cfg["Nl"]).view(1, -1) >= lens.view(-1, 1)
    xd, ld, md = x.to(dev), lang.to(dev), mask.to(dev)
    enc.eval()
    with torch.no_grad():
        v_full, l_full, _, _ = enc(xd, ld, md)
        v4, l4, _, _ = enc(xd[:4], ld[:4], md[:4])
        assert torch.equal(v_full[:4], v4) and torch.equal(l_full[:4], l4)
        # (4) garbage in padded language rows changes nothing
        ld2 = ld.clone()
        ld2[md] = 1e3
        v_g, _, _, _ = enc(xd, ld2, md)
        assert torch.equal(v_g, v_full)
    assert torch.isfinite(v_full).all() and torch.isfinite(l_full).all()
    sd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    with torch.no_grad():
        v_ref, l_ref = O.encoder_forward(sd, x[:2], lang[:2], mask[:2], cfg["h"], cfg["L"])
    assert rel(v_full[:2], v_ref) < FWD_TOL
    assert rel(l_full[:2].cpu()[~mask[:2]], l_ref[~mask[:2]]) < FWD_TOL
    # (3) gradient additivity over the batch (train mode, p = 0)
    enc.train()
    enc.token_dropout = 0.0
    enc.patch_dropout = 0.0
    cot = torch.randn(cfg["B"], cfg["Nv"], cfg["d"], generator=g).to(dev)

    def grads(sl):
        enc.zero_grad(set_to_none=True)
        v, l, _, _ = enc(xd[sl], ld[sl], md[sl])
        (v * cot[sl]).sum().backward()
        return {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}

    g_all, g_a, g_b = grads(slice(0, 32)), grads(slice(0, 16)), grads(slice(16, 32))
    for k in g_all:
        assert rel(g_all[k], g_a[k] + g_b[k]) < 2e-3, k


def test_layerwise_backward_equals_single_call(dev):
    """The per-layer backward (used to overlap the gradient all-reduce) produces the gradients of the single-call
    backward; differences are fp32 atomic-add ordering only."""
    cfg = dict(B=3, Nv=40, Nl=50, d=128, h=4, L=3, mask_lens=[50, 20, 33], seed=31)
    enc, _ = build(cfg, dev)
    enc.train()
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])

    def run(hook):
        enc.zero_grad(set_to_none=True)
        enc.layer_grad_hook = hook
        xd = torch.from_numpy(x).to(dev).requires_grad_(True)
        ld = torch.from_numpy(lang).to(dev).requires_grad_(True)
        v, l, _, _ = enc(xd, ld, torch.from_numpy(mask).to(dev))
        ((v * torch.from_numpy(gv).to(dev)).sum() + (l * torch.from_numpy(gl).to(dev)).sum()).backward()
        enc.layer_grad_hook = None
        return xd.grad.clone(), {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}

    order = []
    gx1, g1 = run(None)
    gx2, g2 = run(lambda m, layer: order.append(layer))
    assert order == [2, 1, 0]
    assert torch.equal(gx1, gx2)
    for k in g1:
        assert rel(g2[k], g1[k]) < 1e-5, k


@pytest.mark.gpu
def test_wgrad_side_stream_matches_single_stream(monkeypatch):
    """tf_encoder_bwd with a TfOverlap (weight-gradient GEMMs on a side stream, fork/join by events) must produce the
    same gradients as the single-stream schedule; repeated to give a missing dependency a chance to show."""
    import torch
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    from transfusion_amd import ops

    def run(overlap):
        monkeypatch.setenv("TF_WGRAD_OVERLAP", "1" if overlap else "0")
        torch.manual_seed(3)
        ops._seed_counter[0] = 0                    # same dropout streams in both schedules
        pe = PositionalEmbeddingLayer("sin1d", 8192, 256)
        m = CrossTransformerModuleBox(8192, 0.1, 256, pe, num_layers=3, num_heads=4, fforward_multiplier=2, token_dropout=0.15,
                                      activ_f="gelu", final_norm="ln").cuda().train()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(6, 49, 256, generator=g).cuda().requires_grad_()
        lang = torch.randn(6, 77, 256, generator=g).cuda().requires_grad_()
        pad = torch.zeros(6, 77, dtype=torch.bool)
        pad[1, 60:] = True
        pad[4, 30:] = True
        outs = []
        for _ in range(3):
            m.zero_grad(set_to_none=True)
            v, l_, _, _ = m(x, lang, pad.cuda(), None)
            (v.float().square().mean() + l_.float().square().mean()).backward()
            outs.append([p.grad.detach().float().cpu().clone() for p in m._param_list()] + [x.grad.cpu().clone(), lang.grad.cpu().clone()])
            x.grad = None
            lang.grad = None
        return outs

    a = run(False)
    b = run(True)
    for ga, gb in zip(a[-1], b[-1]):
        scale = ga.abs().max().item() + 1e-12
        assert (ga - gb).abs().max().item() <= 2e-3 * scale      # fp32 atomics: summation order differs between runs


@pytest.mark.gpu
def test_stress_shape_against_oracle(dev):
    """BASELINE.json configs[3] (long-context stress): 28x28 visual grid + 1024 narration tokens, d = 1024, 4 heads, i.e.
    head dim 256 -- the <256> instantiations of the attention kernels and S = 1808 (29 key tiles, ragged last tile).
    B = 2 samples and 2 layers keep the CPU oracle to a few seconds; dropout off, training mode (every row computed)."""
    from oracle import fusion_oracle as O
    cfg = dict(B=2, Nv=784, Nl=1024, d=1024, h=4, L=2, mask_lens=[1024, 333], seed=11)
    enc, params = build(cfg, dev)
    enc.train()
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    ld = torch.from_numpy(lang).to(dev).requires_grad_(True)
    vis, lo, _, _ = enc(xd, ld, torch.from_numpy(mask).to(dev))
    ((vis * torch.from_numpy(gv).to(dev)).sum() + (lo * torch.from_numpy(gl).to(dev)).sum()).backward()
    assert torch.isfinite(vis).all() and torch.isfinite(lo).all()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, cfg["d"])
    xr, lr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
    v_ref, l_ref = O.encoder_forward(sd, xr, lr, torch.from_numpy(mask), cfg["h"], cfg["L"])
    ((v_ref * torch.from_numpy(gv)).sum() + (l_ref * torch.from_numpy(gl)).sum()).backward()
    valid = ~mask
    assert rel(vis, v_ref.detach()) < FWD_TOL
    assert rel(lo.detach().cpu()[valid], l_ref.detach()[valid]) < FWD_TOL
    assert rel(xd.grad, xr.grad) < GRAD_TOL and rel(ld.grad, lr.grad) < GRAD_TOL
    for k, p in enc.named_parameters():
        if k in sd and sd[k].grad is not None:
            assert rel(p.grad, sd[k].grad) < GRAD_TOL, k


@pytest.mark.gpu
def test_golden_local_mask(dev, golden_dir):
    """vis_mask_type "local_1" on a 4x5 visual grid: fixture produced by the reference itself (tests/golden/make_golden.py),
    here through the HIP path with the block-bit attention mask."""
    cfg = ENCODER_CASES["enc_local1"]
    g = dict(np.load(os.path.join(golden_dir, "enc_local1.npz")))
    enc, params = build(cfg, dev)
    enc.train()
    x = torch.from_numpy(g["in_x"]).to(dev).requires_grad_(True)
    lang = torch.from_numpy(g["in_lang"]).to(dev).requires_grad_(True)
    mask = torch.from_numpy(g["in_mask"]).to(dev)
    vmask = torch.from_numpy(g["in_vis_tokens_mask"])
    vis, lo, _, _ = enc(x, lang, mask, vis_tokens_mask=vmask)
    valid = ~g["in_mask"]
    assert rel(vis, g["train_vis"]) < FWD_TOL
    assert rel(lo.detach().cpu().numpy()[valid], g["train_lang"][valid]) < FWD_TOL
    ((vis * torch.from_numpy(g["cot_vis"]).to(dev)).sum() + (lo * torch.from_numpy(g["cot_lang"]).to(dev)).sum()).backward()
    assert rel(x.grad, g["grad_x"]) < GRAD_TOL and rel(lang.grad, g["grad_lang"]) < GRAD_TOL
    for k, p in enc.named_parameters():
        if "gradp/" + k in g:
            assert rel(p.grad, g["gradp/" + k]) < GRAD_TOL, k
    # the mask matters: without it the visual outputs differ
    v0, _, _, _ = enc(x.detach(), lang.detach(), mask)
    assert rel(v0, g["train_vis"]) > 5 * FWD_TOL


@pytest.mark.gpu
def test_fp8_projections_against_bf16(dev):
    """BASELINE.json configs[4]: fp8 (e4m3) MFMA with fp32 accumulation for the QKV / FFN projections, "parity check vs bf16
    within tol".  Same module, same inputs, dropout off; forward projections in fp8, everything else (attention, out_proj,
    LayerNorm, the whole backward) as in the bf16 path.  Tolerance: two e4m3 operands (3 mantissa bits, per-token and
    per-output-channel scales) put ~3-4 % relative L2 error on each projection; LayerNorm keeps it from compounding, so the
    encoder outputs stay within 6e-2 of the bf16 path and the gradients within 1e-1."""
    cfg = dict(B=4, Nv=196, Nl=128, d=768, h=4, L=2, mask_lens=[128, 40, 77, 100], seed=21)
    enc, params = build(cfg, dev)
    enc.train()
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    outs = {}
    for mode in (False, True):
        enc.fp8_projections = mode
        enc.zero_grad(set_to_none=True)
        xd = torch.from_numpy(x).to(dev).requires_grad_(True)
        ld = torch.from_numpy(lang).to(dev).requires_grad_(True)
        vis, lo, _, _ = enc(xd, ld, torch.from_numpy(mask).to(dev))
        ((vis * torch.from_numpy(gv).to(dev)).sum() + (lo * torch.from_numpy(gl).to(dev)).sum()).backward()
        outs[mode] = (vis.detach().cpu(), lo.detach().cpu(), xd.grad.cpu(), {k: p.grad.detach().cpu().clone() for k, p in enc.named_parameters() if p.grad is not None})
    enc.fp8_projections = False
    valid = ~mask
    e_v, e_l = rel(outs[True][0], outs[False][0]), rel(outs[True][1][valid], outs[False][1][valid])
    assert 1e-4 < e_v < 6e-2 and e_l < 6e-2, (e_v, e_l)          # really a different arithmetic, and within tolerance
    assert rel(outs[True][2], outs[False][2]) < 1e-1
    for k in ("t_encoder.layers.0.self_attn.in_proj_weight", "t_encoder.layers.1.linear2.weight", "t_encoder.layers.0.norm1.weight"):
        assert rel(outs[True][3][k], outs[False][3][k]) < 1e-1, k
    # ... and against the fp32 oracle itself (not only against this library's own bf16 path)
    from oracle import fusion_oracle as O
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, cfg["d"])
    xr, lr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
    v_ref, l_ref = O.encoder_forward(sd, xr, lr, torch.from_numpy(mask), cfg["h"], cfg["L"])
    ((v_ref * torch.from_numpy(gv)).sum() + (l_ref * torch.from_numpy(gl)).sum()).backward()
    assert rel(outs[True][0], v_ref.detach()) < 6e-2 and rel(outs[True][1][valid], l_ref.detach()[valid]) < 6e-2
    assert rel(outs[True][2], xr.grad) < 1e-1
    for k in ("t_encoder.layers.0.self_attn.in_proj_weight", "t_encoder.layers.1.linear2.weight", "t_encoder.layers.0.norm1.weight"):
        assert rel(outs[True][3][k], sd[k].grad) < 1e-1, k


@pytest.mark.gpu
def test_fully_padded_language_sample_against_oracle(dev):
    """One sample has NO valid language token (its keys are the visual tokens only; whole key tiles are skipped), another has all
    of them: outputs on valid rows and all gradients against the oracle."""
    from oracle import fusion_oracle as O
    cfg = dict(B=3, Nv=70, Nl=90, d=64, h=2, L=2, mask_lens=[0, 90, 1], seed=31)
    enc, params = build(cfg, dev)
    enc.train()
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    gl = gl * (~mask)[..., None]                     # cotangents only where the reference's outputs are meaningful
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    ld = torch.from_numpy(lang).to(dev).requires_grad_(True)
    vis, lo, _, _ = enc(xd, ld, torch.from_numpy(mask).to(dev))
    ((vis * torch.from_numpy(gv).to(dev)).sum() + (lo * torch.from_numpy(gl).to(dev)).sum()).backward()
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, cfg["d"])
    xr, lr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
    v_ref, l_ref = O.encoder_forward(sd, xr, lr, torch.from_numpy(mask), cfg["h"], cfg["L"])
    ((v_ref * torch.from_numpy(gv)).sum() + (l_ref * torch.from_numpy(gl)).sum()).backward()
    valid = ~mask
    assert rel(vis, v_ref.detach()) < FWD_TOL
    assert rel(lo.detach().cpu()[valid], l_ref.detach()[valid]) < FWD_TOL
    assert rel(xd.grad, xr.grad) < GRAD_TOL and rel(ld.grad, lr.grad) < GRAD_TOL
    for k, p in enc.named_parameters():
        if k in sd and sd[k].grad is not None:
            assert rel(p.grad, sd[k].grad) < GRAD_TOL, k


@pytest.mark.gpu
def test_gradient_accumulation_equals_concatenated_batch(dev):
    """accumulate_into_grad: two micro-batches accumulated into p.grad give the gradients of the concatenated batch (dropout off;
    the per-sample arithmetic does not depend on the batch a sample travels in), and eval-mode forwards are bitwise repeatable."""
    cfg = dict(B=4, Nv=49, Nl=40, d=128, h=4, L=2, mask_lens=[40, 13, 25, 40], seed=41)
    enc, _ = build(cfg, dev)
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    t = lambda a: torch.from_numpy(a).to(dev)

    def run(slices):
        enc.train()
        enc.accumulate_into_grad = True
        for p in enc.parameters():
            p.grad = None
        for sl in slices:
            v, l_, _, _ = enc(t(x[sl]), t(lang[sl]), t(mask[sl]))
            ((v * t(gv[sl])).sum() + (l_ * t(gl[sl])).sum()).backward()
        enc.accumulate_into_grad = False
        return {k: p.grad.detach().cpu().clone() for k, p in enc.named_parameters() if p.grad is not None}

    whole = run([slice(0, 4)])
    parts = run([slice(0, 2), slice(2, 4)])
    for k in whole:
        assert rel(parts[k], whole[k]) < 2e-3, k           # same products, different fp32 summation order (atomics)
    enc.eval()
    with torch.no_grad():
        a = enc(t(x), t(lang), t(mask))[0]
        b = enc(t(x), t(lang), t(mask))[0]
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_training_steps_with_changing_batches_track_the_oracle(dev):
    """Three optimiser steps of FusionTrainStep, every step a DIFFERENT batch (new tensors, new padding lengths): after each
    update the gradients of the next batch are compared with the oracle evaluated at the parameters read back from the device.
    Covers what single-call tests cannot: weight re-pack after the fused RAdam wrote through raw pointers, mask / work-buffer
    reuse across calls, accumulation into the flat gradient buffer."""
    from oracle import fusion_oracle as O
    from transfusion_amd.runner.trainer import FusionTrainStep
    cfg = dict(B=3, Nv=36, Nl=50, d=64, h=4, L=2, mask_lens=None, seed=51)
    enc, _ = build(cfg, dev)
    enc.train()
    tr = FusionTrainStep(enc, lr=3e-2, weight_decay=1e-3, grad_clip=1.0)       # a large step: the parameters really move
    names = [n for n, _, _, _ in tr.flat.slices]
    losses = []
    for step, lens in enumerate(([50, 7, 31], [12, 50, 50], [1, 44, 20])):
        x, lang, mask, gv, gl = make_encoder_inputs(100 + step, cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], lens)
        gl = gl * (~mask)[..., None]
        t = lambda a: torch.from_numpy(a).to(dev)
        # --- gradients of this batch at the CURRENT parameters, on the device (what trainer.step does before the update) ---
        tr.zero_grad()
        vis, lo, _, _ = enc(t(x), t(lang), t(mask))
        loss = (vis * t(gv)).sum() + (lo * t(gl)).sum()
        loss.backward()
        losses.append(float(loss.detach()))
        # --- the same on the CPU oracle with the parameters read back ---
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in enc.state_dict().items() if v.is_floating_point() and "pos_embedding" not in k}
        sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, cfg["d"])
        v_ref, l_ref = O.encoder_forward(sd, torch.from_numpy(x), torch.from_numpy(lang), torch.from_numpy(mask), cfg["h"], cfg["L"])
        ((v_ref * torch.from_numpy(gv)).sum() + (l_ref * torch.from_numpy(gl)).sum()).backward()
        assert rel(vis, v_ref.detach()) < FWD_TOL, step
        for n, p in enc.named_parameters():
            if n in names and sd[n].grad is not None:
                assert rel(p.grad, sd[n].grad) < GRAD_TOL, (step, n)
        # --- update: the fused RAdam call of trainer.step (no movement yet in its first, un-rectified steps, as in the reference's
        # optimiser) followed by a plain SGD step THROUGH THE FLAT BUFFER so that the parameters really move between batches ---
        before = tr.flat.flat.clone()
        tr._norm.zero_()
        tr.opt.grad_sumsq(tr._norm)
        tr.opt.step(grad_scale=1.0, sumsq=tr._norm, clip=tr.grad_clip)
        tr.flat.flat.add_(tr.flat.grad, alpha=-2e-3)
        enc.mark_weights_updated()
        assert torch.isfinite(tr.flat.flat).all()
        assert (tr.flat.flat - before).abs().max().item() > 1e-3
    assert all(np.isfinite(losses))


@pytest.mark.gpu
def test_two_forwards_before_their_backwards(dev):
    """loss = f(enc(a)) + f(enc(b)): the second forward must not recycle the workspace (saved activations, masks, dropout
    streams) of the first while its backward is still pending.  Gradients of the sum of the two losses against the oracle."""
    from oracle import fusion_oracle as O
    cfg = dict(B=2, Nv=16, Nl=24, d=64, h=4, L=2, seed=61)
    enc, params = build(cfg, dev)
    enc.train()
    xa, la, ma, gva, gla = make_encoder_inputs(601, cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], [24, 10])
    xb, lb, mb, gvb, glb = make_encoder_inputs(602, cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], [5, 17])
    t = lambda a: torch.from_numpy(a).to(dev)
    xad, xbd = t(xa).requires_grad_(True), t(xb).requires_grad_(True)
    va, loa, _, _ = enc(xad, t(la), t(ma))
    vb, lob, _, _ = enc(xbd, t(lb), t(mb))          # same shape: would reuse the same workspace if it were released
    (pool,) = enc._work_pool.values()
    assert len(pool) == 2
    loss = (va * t(gva)).sum() + (loa * t(gla)).sum() + (vb * t(gvb)).sum() + (lob * t(glb)).sum()
    loss.backward()
    assert all(not it["busy"] for it in pool)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, cfg["d"])
    xar, xbr = torch.from_numpy(xa).requires_grad_(True), torch.from_numpy(xb).requires_grad_(True)
    var, lar = O.encoder_forward(sd, xar, torch.from_numpy(la), torch.from_numpy(ma), cfg["h"], cfg["L"])
    vbr, lbr = O.encoder_forward(sd, xbr, torch.from_numpy(lb), torch.from_numpy(mb), cfg["h"], cfg["L"])
    ((var * torch.from_numpy(gva)).sum() + (lar * torch.from_numpy(gla)).sum() + (vbr * torch.from_numpy(gvb)).sum()
     + (lbr * torch.from_numpy(glb)).sum()).backward()
    assert rel(va, var.detach()) < FWD_TOL and rel(vb, vbr.detach()) < FWD_TOL
    assert rel(xad.grad, xar.grad) < GRAD_TOL and rel(xbd.grad, xbr.grad) < GRAD_TOL
    for k, p in enc.named_parameters():
        if k in sd and sd[k].grad is not None:
            assert rel(p.grad, sd[k].grad) < GRAD_TOL, k
    # a second backward through a released-and-recycled workspace fails loudly instead of using another call's activations
    v1, _, _, _ = enc(xad, t(la), t(ma))
    s1 = v1.sum()
    s1.backward(retain_graph=True)
    v2, _, _, _ = enc(xbd, t(lb), t(mb))            # recycles the workspace v1's graph points at
    from transfusion_amd._lib import TfError
    with pytest.raises(TfError):
        s1.backward()
    v2.sum().backward()


@pytest.mark.gpu
def test_workspace_pool_is_bounded_under_changing_token_counts(dev):
    """A loader that pads the language tokens to the longest sample of each batch changes Nl almost every step: the encoder must not keep
    one workspace per length ever seen, a forward whose workspace was evicted before its backward must still be correct, and coming back
    to an earlier length (fresh workspace) must reproduce its first result bit for bit."""
    from oracle import fusion_oracle as O
    cfg = dict(B=2, Nv=16, d=64, h=4, L=2, seed=67)
    enc, params = build(dict(cfg, Nl=24), dev)
    enc.eval()
    t = lambda a: torch.from_numpy(a).to(dev)
    first = {}
    pending = None
    for step, nl in enumerate([24, 40, 17, 33, 64, 24, 9, 40]):
        x, l, m, gv, gl = make_encoder_inputs(700 + nl, cfg["B"], cfg["Nv"], nl, cfg["d"], [nl, max(1, nl // 3)])
        xd = t(x).requires_grad_(True)
        v, lo, _, _ = enc(xd, t(l), t(m))
        assert len(enc._work_pool) <= enc.MAX_WORK_SHAPES
        assert sum(len(p) for p in enc._work_pool.values()) <= enc.MAX_WORK_SHAPES * 4
        if nl in first:
            assert torch.equal(v, first[nl][0]) and torch.equal(lo, first[nl][1])
        else:
            first[nl] = (v.detach().clone(), lo.detach().clone())
        if step == 0:
            pending = (xd, x, l, m, gv, gl, v, lo)        # its backward runs after its workspace has left the pool
    xd, x, l, m, gv, gl, v, lo = pending
    assert (cfg["B"], cfg["Nv"], 24, enc.precision) in enc._work_pool      # came back at step 5 -- with a NEW workspace
    ((v * t(gv)).sum() + (lo * t(gl)).sum()).backward()
    sd = {k: torch.from_numpy(p).clone().requires_grad_(True) for k, p in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, cfg["d"])
    xr = torch.from_numpy(x).requires_grad_(True)
    vr, lr = O.encoder_forward(sd, xr, torch.from_numpy(l), torch.from_numpy(m), cfg["h"], cfg["L"])
    ((vr * torch.from_numpy(gv)).sum() + (lr * torch.from_numpy(gl)).sum()).backward()
    assert rel(v, vr.detach()) < FWD_TOL and rel(xd.grad, xr.grad) < GRAD_TOL
    for k, p in enc.named_parameters():
        if k in sd and sd[k].grad is not None:
            assert rel(p.grad, sd[k].grad) < GRAD_TOL, k


@pytest.mark.gpu
def test_standalone_optimizer_step_refreshes_weight_shadows(dev):
    """FusedRAdam used the way the reference uses RAdam (opt = cls(model.parameters()); loss.backward(); opt.step()): the raw-pointer
    update bumps the parameter versions, so the next forward re-packs the bf16 weight shadows and its output changes."""
    from transfusion_amd.optim import FusedRAdam
    cfg = dict(B=2, Nv=16, Nl=8, d=64, h=4, L=1, seed=71)
    enc, _ = build(cfg, dev)
    enc.train()
    opt = FusedRAdam([p for n, p in enc.named_parameters() if n != "heatmap_token"], lr=5e-2, degenerated_to_sgd=True)
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], [8, 3])
    t = lambda a: torch.from_numpy(a).to(dev)
    outs = []
    for _ in range(2):
        opt.zero_grad()
        v, l_, _, _ = enc(t(x), t(lang), t(mask))
        outs.append(v.detach().clone())
        ((v * t(gv)).sum()).backward()
        opt.step()
    assert rel(outs[1], outs[0]) > 1e-2           # the update reached the kernels
