"""TfEncoderDesc.groups: G encoders of identical shape and different weights (the wrapper's FPN levels) as ONE runtime call -- every
row range against its own parameters in the GEMM, weight-gradient, LayerNorm and assemble kernels.  The yardstick is the same G
encoders called one after the other (the path every other test pins to the reference): outputs, input gradients and every parameter
gradient must agree."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _encoders(dev, G, d, H, layers, p_tok, p_patch, final_norm="ln"):
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    mods = []
    for k in range(G):
        torch.manual_seed(100 + k)                                 # different weights per level
        pe = PositionalEmbeddingLayer("sin1d", 256, d)
        m = CrossTransformerModuleBox(no_patches=256, pos_embedding_layer=pe, lang_pos_embedding=None, num_layers=layers, patch_dropout=p_patch,
                                      num_heads=H, fforward_multiplier=2, token_dropout=p_tok, back_to_img_fn="regroup", activ_f="gelu",
                                      final_norm=final_norm, input_f_size=d)
        # LayerNorm weights / kind embeddings away from their defaults, so that a group reading ANOTHER group's vector shows
        with torch.no_grad():
            for q in m.parameters():
                if q.dim() == 1:
                    q.add_(0.3 * torch.randn_like(q))
        mods.append(m)
    return torch.nn.ModuleList(mods).to(dev).train()


@pytest.mark.parametrize("G,B,Nv,Nl,d,H,layers,packed,precision", [
    (4, 2, 36, 40, 64, 2, 2, False, "bf16"),
    (4, 2, 36, 40, 64, 2, 2, True, "bf16"),
    (3, 3, 49, 24, 192, 4, 1, True, "bf16"),
    (2, 4, 196, 128, 768, 4, 2, True, "bf16"),       # the wrapper's level shape (d = 768, 196 visual tokens), large-tile kernels
    (2, 2, 36, 40, 64, 2, 2, True, "fp32"),
])
def test_grouped_call_equals_separate_calls(G, B, Nv, Nl, d, H, layers, packed, precision):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from transfusion_amd.runner.trainer import FusionTrainStep
    dev = torch.device("cuda:0")
    mods = _encoders(dev, G, d, H, layers, 0.0, 0.0)              # dropout off: element indices (hence masks) differ between the two paths
    for m in mods:
        m.precision = precision
    tr = FusionTrainStep(mods, lr=0.0, weight_decay=0.0, grad_clip=None)     # flat parameter / gradient buffers: one common stride
    g = torch.Generator().manual_seed(G * 1000 + Nv)
    xs = [torch.randn(B, Nv, d, generator=g).to(dev).requires_grad_(True) for _ in range(G)]
    lang = torch.randn(B, Nl, d, generator=g).to(dev).requires_grad_(True)
    lens = torch.randint(Nl // 3, Nl + 1, (B,), generator=g)
    lens[0] = Nl
    pad = (torch.arange(Nl).view(1, -1) >= lens.view(-1, 1)).to(dev)
    n_valid = int(lens.sum())
    gv = [torch.randn(B, Nv, d, generator=g).to(dev) for _ in range(G)]
    gl = [(torch.randn(B, Nl, d, generator=g) * (~pad.cpu()).unsqueeze(-1)).to(dev) for _ in range(G)]
    assert mods[0].group_stride(list(mods)) is not None

    # ---- separate calls ----
    tr.zero_grad()
    outs, louts = [], []
    for k, m in enumerate(mods):
        v, lo, _, _ = m(xs[k], lang, pad, lang_valid_rows=n_valid if packed else None)
        outs.append(v)
        louts.append(lo)
    torch.autograd.backward(outs + louts, gv + gl)
    torch.cuda.synchronize()
    ref_grad = tr.flat.grad.detach().clone()
    ref_dx = [x.grad.detach().clone() for x in xs]
    ref_dlang = lang.grad.detach().clone()
    for x in xs:
        x.grad = None
    lang.grad = None

    # ---- one grouped call ----
    tr.zero_grad()
    X = torch.cat(xs, dim=0)
    Lg, Pg = lang.repeat(G, 1, 1), pad.repeat(G, 1)
    V, LO, _, _ = mods[0].forward_grouped(list(mods), X, Lg, Pg, lang_valid_rows=G * n_valid if packed else None)
    torch.autograd.backward([V, LO], [torch.cat(gv, dim=0), torch.cat(gl, dim=0)])
    torch.cuda.synchronize()
    if packed:
        assert mods[0].packed_row_error() == 0
    tol = 1e-4 if precision == "fp32" else 6e-3                    # tile shapes differ between the two paths: bf16 rounding, nothing more
    for k in range(G):
        assert rel(V[k * B:(k + 1) * B], outs[k]) < tol, ("vis", k)
        keep = (~pad).unsqueeze(-1)
        assert rel(LO[k * B:(k + 1) * B] * keep, louts[k] * keep) < tol, ("lang", k)
        assert rel(xs[k].grad, ref_dx[k]) < 3 * tol, ("dx", k)
    assert rel(lang.grad, ref_dlang) < 3 * tol
    # every parameter gradient, encoder by encoder
    for name, p, off, n in tr.flat.slices:
        a, b = tr.flat.grad[off:off + n], ref_grad[off:off + n]
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, name
        else:
            assert rel(a, b) < 4 * tol, (name, rel(a, b))


def test_group_stride_refuses_what_cannot_be_grouped():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = torch.device("cuda:0")
    mods = _encoders(dev, 2, 64, 2, 1, 0.0, 0.0)
    assert mods[0].group_stride(list(mods)) is None               # separate allocations, gradients through autograd: no common stride
    from transfusion_amd.runner.trainer import FusionTrainStep
    FusionTrainStep(mods, lr=0.0, weight_decay=0.0, grad_clip=None)
    assert mods[0].group_stride(list(mods)) is not None
    mods[1].token_dropout = 0.3
    mods[0]._group_check = None
    assert mods[0].group_stride(list(mods)) is None               # different configuration


@pytest.mark.parametrize("nvs,B,Nl,d,H,layers,precision", [
    ([64, 16, 16, 16], 2, 40, 64, 2, 2, "bf16"),        # the real FPN geometry in small: 4N / N / N / N visual tokens
    ([9, 36], 3, 24, 192, 4, 1, "bf16"),                # the LARGEST group need not come first
    ([196, 49], 4, 128, 768, 4, 2, "bf16"),             # d = 768: the large-tile kernels walk the ragged ranges
    ([64, 16, 16], 2, 40, 64, 2, 2, "fp32"),            # fp32-accuracy mode
    ([25, 64, 9], 2, 40, 64, 2, 1, "bf16-nonorm"),      # final_norm: false -- the visual rows leave / enter through plain copies
])
def test_ragged_grouped_call_equals_separate_calls(nvs, B, Nl, d, H, layers, precision):
    """TfEncoderDesc.group_nv: encoders that differ in their visual token count as ONE grouped call on packed rows -- visual tokens,
    outputs and their gradients as the concatenation [sum_g B nv_g, d] -- against the same encoders called one by one: outputs, input
    gradients, the shared narration tokens' gradient (summed over the groups) and EVERY parameter gradient."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from transfusion_amd.runner.trainer import FusionTrainStep
    dev = torch.device("cuda:0")
    G = len(nvs)
    final_norm = False if precision.endswith("-nonorm") else "ln"
    precision = precision.split("-")[0]
    mods = _encoders(dev, G, d, H, layers, 0.0, 0.0, final_norm)  # dropout off: element indices (hence masks) differ between the two paths
    for m in mods:
        m.precision = precision
    tr = FusionTrainStep(mods, lr=0.0, weight_decay=0.0, grad_clip=None)
    g = torch.Generator().manual_seed(G * 1000 + sum(nvs))
    xs = [torch.randn(B, nv, d, generator=g).to(dev).requires_grad_(True) for nv in nvs]
    lang = torch.randn(B, Nl, d, generator=g).to(dev).requires_grad_(True)
    lens = torch.randint(Nl // 3, Nl + 1, (B,), generator=g)
    lens[0] = Nl
    pad = (torch.arange(Nl).view(1, -1) >= lens.view(-1, 1)).to(dev)
    n_valid = int(lens.sum())
    gv = [torch.randn(B, nv, d, generator=g).to(dev) for nv in nvs]
    gl = [(torch.randn(B, Nl, d, generator=g) * (~pad.cpu()).unsqueeze(-1)).to(dev) for _ in range(G)]
    assert mods[0].group_stride(list(mods), ragged=True) is not None

    tr.zero_grad()
    outs, louts = [], []
    for k, m in enumerate(mods):
        v, lo, _, _ = m(xs[k], lang, pad, lang_valid_rows=n_valid)
        outs.append(v)
        louts.append(lo)
    torch.autograd.backward(outs + louts, gv + gl)
    torch.cuda.synchronize()
    ref_grad = tr.flat.grad.detach().clone()
    ref_dx = [x.grad.detach().clone() for x in xs]
    ref_dlang = lang.grad.detach().clone()
    for x in xs:
        x.grad = None
    lang.grad = None

    tr.zero_grad()
    X = torch.cat([x.reshape(-1, d) for x in xs], dim=0)          # [sum_g B nv_g, d]
    offs = [0]
    for nv in nvs:
        offs.append(offs[-1] + B * nv)
    V, LO, _, _ = mods[0].forward_grouped(list(mods), X, lang.repeat(G, 1, 1), pad.repeat(G, 1), lang_valid_rows=G * n_valid, group_nv=nvs)
    assert tuple(V.shape) == (offs[-1], d) and list(mods[0]._last_desc.group_nv)[:G] == nvs
    torch.autograd.backward([V, LO], [torch.cat([t.reshape(-1, d) for t in gv], dim=0), torch.cat(gl, dim=0)])
    torch.cuda.synchronize()
    assert mods[0].packed_row_error() == 0
    tol = 1e-4 if precision == "fp32" else 6e-3
    keep = (~pad).unsqueeze(-1)
    for k in range(G):
        assert rel(V[offs[k]:offs[k + 1]], outs[k].reshape(-1, d)) < tol, ("vis", k)
        assert rel(LO[k * B:(k + 1) * B] * keep, louts[k] * keep) < tol, ("lang", k)
        assert rel(xs[k].grad, ref_dx[k]) < 3 * tol, ("dx", k)
    assert rel(lang.grad, ref_dlang) < 3 * tol
    for name, p, off, n in tr.flat.slices:
        a, b = tr.flat.grad[off:off + n], ref_grad[off:off + n]
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, name
        else:
            assert rel(a, b) < 4 * tol, (name, rel(a, b))
    # a count that does not split into equal language shares, or dense rows, is refused before anything is launched
    from transfusion_amd import _lib as L
    with pytest.raises(L.TfError):
        mods[0].forward_grouped(list(mods), X.detach(), lang.detach().repeat(G, 1, 1), pad.repeat(G, 1), group_nv=nvs)


def test_wrapper_levels_grouped_equal_level_loop(monkeypatch):
    """CrossFusionBoxWrapper under FusionTrainStep takes the grouped path (one encoder call for all FPN levels); TF_GROUP_LEVELS=0 keeps
    the level loop.  Same weights, same inputs, dropout off: fused feature maps, language tokens and every gradient must agree."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_wrapper import StubDetector
    from transfusion_amd.modeling.model_factory import get_fusion_model
    from transfusion_amd.runner.config import load_fusion_config
    from transfusion_amd.runner.trainer import FusionTrainStep
    dev = torch.device("cuda:0")
    B, NL, D = 2, 24, 128
    ps, chans = [4, 4, 2, 1], [8, 16, 32, 64]
    shapes = [(6 * p, 6 * p) for p in ps]                         # 36 visual tokens on every level
    fusion = load_fusion_config(os.path.join(ROOT, "transfusion_amd", "runner", "configs", "cross_fusion_config_sym_ego_res50.yml"))
    fusion.update({"fpn_features": [0, 1, 2, 3], "replace_fpn_features": True, "backproj_dropout": 0.0})
    fusion["args"].update({"input_f_size": D, "num_heads": 2, "num_layers": [2, 2, 2, 2], "patch_dropout": 0.0, "token_dropout": 0.0})
    run_cfg = {"experiment": "egonao", "narr_fusion": fusion, "criterion": {"lm": 0}, "precision": 16,
               "narration_embeds": {"use": True, "args": {"text_pooling": "slowfast", "strategy": "current", "out_mlp": 0, "size": D,
                                                         "out_dropout": 0.0, "out_tanh": False, "train_ep": 0}}}
    torch.manual_seed(7)
    model = get_fusion_model(StubDetector(shapes, chans), {}, run_cfg, None).to(dev).train()
    tr = FusionTrainStep(model, lr=0.0, weight_decay=0.0, grad_clip=None)
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(B, c, h, w, generator=g).to(dev).requires_grad_(True) for c, (h, w) in zip(chans, shapes)]
    lang = [torch.randn(n, D, generator=g).to(dev) for n in (NL, NL - 7)]
    gouts = None
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TF_GROUP_LEVELS", mode)
        tr.zero_grad()
        # fresh leaves per mode: a leaf's AccumulateGrad node remembers the stream of its first use, and the two modes deliver the
        # feature-map gradients on different streams (level streams / the grouped call's) -- in training one mode runs for good
        feats = [f.detach().clone().requires_grad_(True) for f in feats]
        out = model({"image": feats, "language_f": lang})
        fs = [out["features"][str(i)] for i in range(4)]
        if gouts is None:
            gouts = [torch.randn(f.shape, generator=g).to(dev).to(f.dtype) for f in fs]
        torch.autograd.backward(fs, gouts)
        torch.cuda.synchronize()
        res[mode] = ([f.detach().float().cpu() for f in fs], [f.grad.detach().float().cpu() for f in feats], tr.flat.grad.detach().cpu().clone())
    enc0 = model.cross_fusion_encoders[0]
    assert enc0.group_stride(list(model.cross_fusion_encoders)) is not None and enc0._last_desc.groups == 4      # the grouped path did run
    for a, b in zip(res["1"][0], res["0"][0]):
        assert rel(a, b) < 6e-3
    for a, b in zip(res["1"][1], res["0"][1]):
        assert rel(a, b) < 2e-2
    for name, p, off, n in tr.flat.slices:
        a, b = res["1"][2][off:off + n], res["0"][2][off:off + n]
        if float(b.abs().max()) > 0:
            assert rel(a, b) < 2.5e-2, (name, rel(a, b))


@pytest.mark.parametrize("d,ragged", [(128, False), (72, False), (128, True), (72, True)])
def test_level_nodes_with_dropout_against_torch(monkeypatch, d, ragged):
    """level_ops (K1 / K9 of all levels as one autograd node each) with the back-projection dropout ON: forward and every gradient
    against fp64 torch on the bf16-rounded operands, the dropout mask replayed from the library (tf_dropout_mask: same key, same
    element index = position inside the level's [B * Nv, d] input).  ``d`` = 72: a token width that is no multiple of 64 (the Ego4D v1
    config's 712): padded copies in front of the GEMMs that contract over d.  ``ragged``: levels of unequal token counts (the real FPN
    geometry): the nodes then take / return the CONCATENATION [sum_g B nv_g, d]."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from transfusion_amd import level_ops, ops
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_wrapper import PatchToToken
    from transfusion_amd.modeling.cross_fusion.utils import RegroupPatchesLayerBox
    dev = torch.device("cuda:0")
    SEED = 0x5151
    monkeypatch.setattr(ops, "next_seed", lambda: SEED)
    B, p = 2, 0.25
    ps, chans = [4, 2, 1], [8, 32, 64]
    shapes = [(6 * q, 6 * q) for q in ps]                        # 36 tokens per level
    if ragged:
        shapes[0] = (12 * ps[0], 6 * ps[0])                      # ... 72 on level 0
    nvs = [(h // q) * (w // q) for (h, w), q in zip(shapes, ps)]
    offs = [0]
    for n in nvs:
        offs.append(offs[-1] + B * n)
    torch.manual_seed(5)
    k1 = [PatchToToken(c, d, q, q).to(dev) for c, q in zip(chans, ps)]
    k9 = [RegroupPatchesLayerBox(d, h, w, q, q, c, backproj_dropout=p).to(dev).train() for c, q, (h, w) in zip(chans, ps, shapes)]
    g = torch.Generator().manual_seed(9)
    feats = [torch.randn(B, c, h, w, generator=g).to(dev).requires_grad_(True) for c, (h, w) in zip(chans, shapes)]
    streams = [torch.cuda.Stream() for _ in ps]
    bfr = lambda t: t.detach().to(torch.bfloat16).double().cpu()

    # ---- K1 ----
    x = level_ops.levels_patch_embed(k1, feats, streams)          # [G * B, Nv, d], or (ragged) [sum_g B nv_g, d]
    assert tuple(x.shape) == ((offs[-1], d) if ragged else (len(ps) * B, nvs[0], d))
    gx = torch.randn(x.shape, generator=g).to(dev).to(torch.bfloat16)
    x.backward(gx)
    torch.cuda.synchronize()
    x2, gx2 = x.reshape(-1, d), gx.reshape(-1, d)
    for i, (m, f) in enumerate(zip(k1, feats)):
        q, Nv = ps[i], nvs[i]
        rows = torch.nn.functional.unfold(bfr(f), kernel_size=q, stride=q).transpose(1, 2).reshape(B * Nv, -1)      # [B * Nv, C * q * q]
        w = bfr(m.weight).reshape(d, -1)
        assert rel(x2[offs[i]:offs[i + 1]], rows @ w.t()) < 6e-3, i
        gy = gx2[offs[i]:offs[i + 1]].double().cpu()
        assert rel(m.weight.grad.reshape(d, -1), gy.t() @ rows) < 2e-5 + 6e-3, i
        dcols = (gy @ w).reshape(B, Nv, -1).transpose(1, 2)
        dref = torch.nn.functional.fold(dcols, output_size=shapes[i], kernel_size=q, stride=q)
        assert rel(f.grad, dref) < 8e-3, i

    # ---- K9 with dropout ----
    fused = torch.randn(*((offs[-1], d) if ragged else (len(ps) * B, nvs[0], d)), generator=g).to(dev).to(torch.bfloat16).requires_grad_(True)
    outs = level_ops.levels_back_project(k9, fused, streams)
    gouts = [torch.randn(o.shape, generator=g).to(dev) for o in outs]
    torch.autograd.backward(outs, gouts)
    torch.cuda.synchronize()
    thr, key, scale = ops.drop_params(p, SEED, 7)
    f2, fg2 = fused.detach().reshape(-1, d), fused.grad.reshape(-1, d)
    for i, m in enumerate(k9):
        q, (h, w_), Nv = ps[i], shapes[i], nvs[i]
        keep = ops.dropout_mask(B * Nv * d, p, SEED, 7, dev).view(B * Nv, d).double().cpu() * scale
        xi = f2[offs[i]:offs[i + 1]].double().cpu()
        xd = (xi * keep).to(torch.bfloat16).double()              # the kernel rounds the dropped-out input to bf16
        W, bias = bfr(m.linear.weight), m.linear.bias.detach().double().cpu()
        y = (xd @ W.t() + bias).to(torch.bfloat16).double()       # [B * Nv, C * q * q] (stored as bf16 before the fold)
        ref = torch.nn.functional.fold(y.reshape(B, Nv, -1).transpose(1, 2), output_size=(h, w_), kernel_size=q, stride=q)
        assert rel(outs[i], ref) < 6e-3, i
        gy = torch.nn.functional.unfold(gouts[i].double().cpu(), kernel_size=q, stride=q).transpose(1, 2).reshape(B * Nv, -1)
        gy = gy.to(torch.bfloat16).double()
        assert rel(m.linear.weight.grad, gy.t() @ xd) < 8e-3, i
        assert rel(m.linear.bias.grad, gy.sum(0)) < 8e-3, i
        assert rel(fg2[offs[i]:offs[i + 1]], (gy @ W) * keep) < 1e-2, i
