"""Pin the CPU oracle (oracle/fusion_oracle.py) against fixtures produced by the
reference itself (tests/golden/make_golden.py).  fp32 tolerance: 2e-5 abs on O(1) values."""
import os

import numpy as np
import pytest
import torch

from cases import (ENCODER_CASES, LEVEL_CASES, MLEVEL_CASES, make_encoder_inputs, make_encoder_params)
from oracle import fusion_oracle as O

TOL = 2e-5


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False))


def _sd(params, d):
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, d)
    return sd


def _close(a, b, tol=TOL, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), f"{what}: max abs err {err}"


@pytest.mark.parametrize("name", [n for n, c in ENCODER_CASES.items() if not c.get("big")])
def test_encoder_small_cases(golden_dir, name):
    cfg = ENCODER_CASES[name]
    g = _load(golden_dir, name)
    params = {k[6:]: v for k, v in g.items() if k.startswith("param/")}
    # the committed inputs equal what the seeded generator makes (so GPU tests may regenerate them)
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    assert np.array_equal(x, g["in_x"]) and np.array_equal(lang, g["in_lang"])
    regen = make_encoder_params(cfg["seed"], cfg["d"], cfg["L"])
    assert all(np.array_equal(regen[k], params[k]) for k in params)

    sd = _sd(params, cfg["d"])
    if cfg.get("lang_pos") == "sin1d":
        sd["lang_pos_embedding.pos_embedding"] = O.sin1d_table(256, cfg["d"])          # wrapper :100-105
    tx = torch.from_numpy(x).requires_grad_(True)
    tl = torch.from_numpy(lang).requires_grad_(True)
    tm = None if mask is None else torch.from_numpy(mask)
    vm = torch.from_numpy(g["in_vis_tokens_mask"]) if "in_vis_tokens_mask" in g else None
    vis, lo = O.encoder_forward(sd, tx, tl, tm, cfg["h"], cfg["L"], vis_tokens_mask=vm, activation=cfg.get("activ", "gelu"))
    _close(vis, g["train_vis"], what="train_vis")
    _close(lo, g["train_lang"], what="train_lang")
    # eval fast path: visual rows equal, language rows equal where not padded (padded rows are zeros there)
    _close(vis, g["eval_vis"], what="eval_vis")
    valid = np.ones(lang.shape[:2], bool) if mask is None else ~mask
    _close(lo.detach().numpy()[valid], g["eval_lang"][valid], what="eval_lang[valid]")
    if mask is not None and mask.any() and vm is None:
        # nested-tensor fast path (not taken when a src mask is given) zeroes the padded rows
        assert np.all(g["eval_lang"][mask] == 0)

    loss = (vis * torch.from_numpy(gv)).sum() + (lo * torch.from_numpy(gl)).sum()
    loss.backward()
    _close(tx.grad, g["grad_x"], what="grad_x")
    _close(tl.grad, g["grad_lang"], what="grad_lang")
    for k, v in g.items():
        if k.startswith("gradp/"):
            _close(sd[k[6:]].grad, v, tol=5e-5, what=k)
    assert sd["heatmap_token"].grad is None
    assert "gradp/heatmap_token" not in g


@pytest.mark.parametrize("name", ["enc_d768", "enc_d712", "enc_d896"])
def test_encoder_real_width_subsampled(golden_dir, name):
    """d = 768 (BASELINE's synthetic width) and the reference's true widths 712 / 896 (head dims 178 / 224)."""
    cfg = ENCODER_CASES[name]
    g = _load(golden_dir, name)
    params = make_encoder_params(cfg["seed"], cfg["d"], cfg["L"])
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    sd = _sd(params, cfg["d"])
    tx = torch.from_numpy(x).requires_grad_(True)
    tl = torch.from_numpy(lang).requires_grad_(True)
    vis, lo = O.encoder_forward(sd, tx, tl, torch.from_numpy(mask), cfg["h"], cfg["L"])
    _close(vis[:, ::14], g["train_vis_rows"], what="vis rows")
    _close(lo[:, ::8], g["train_lang_rows"], what="lang rows")
    assert abs(vis.double().sum().item() - g["train_vis_sum"]) < 2e-2
    ((vis * torch.from_numpy(gv)).sum() + (lo * torch.from_numpy(gl)).sum()).backward()
    _close(tx.grad[:, ::14], g["grad_x_rows"], tol=5e-5, what="grad_x rows")
    for k, v in g.items():
        if k.startswith("gradp_head/"):
            _close(sd[k[11:]].grad.reshape(-1)[:256], v, tol=1e-4, what=k)


@pytest.mark.parametrize("name", list(LEVEL_CASES))
def test_level_cases(golden_dir, name):
    cfg = LEVEL_CASES[name]
    g = _load(golden_dir, name)
    params = {k[6:]: v for k, v in g.items() if k.startswith("param/")}
    sd = _sd(params, cfg["d"])
    feat = torch.from_numpy(g["in_feat"]).requires_grad_(True)
    lang = torch.from_numpy(g["in_lang"]).requires_grad_(True)
    conv_w = torch.from_numpy(g["conv_w"]).requires_grad_(True)
    reg_w = torch.from_numpy(g["reg_w"]).requires_grad_(True)
    reg_b = torch.from_numpy(g["reg_b"]).requires_grad_(True)
    tok = O.patch_embed(feat, conv_w)
    _close(tok, g["tokens"], what="tokens")
    fused, lo = O.fusion_level_forward(feat, conv_w, sd, lang, torch.from_numpy(g["in_mask"]), cfg["h"], cfg["L"],
                                       reg_w, reg_b, cfg["p"], cfg["p"])
    _close(fused, g["fused"], what="fused")
    _close(lo, g["lang_out"], what="lang_out")
    ((fused * torch.from_numpy(g["cot_out"])).sum() + (lo * torch.from_numpy(g["cot_lang"])).sum()).backward()
    _close(feat.grad, g["grad_feat"], tol=5e-5, what="grad_feat")
    _close(lang.grad, g["grad_lang"], tol=5e-5, what="grad_lang")
    _close(conv_w.grad, g["grad_conv_w"], tol=5e-5, what="grad_conv_w")
    _close(reg_w.grad, g["grad_reg_w"], tol=5e-5, what="grad_reg_w")
    _close(reg_b.grad, g["grad_reg_b"], tol=5e-5, what="grad_reg_b")
    for k, v in g.items():
        if k.startswith("gradp/"):
            _close(sd[k[6:]].grad, v, tol=5e-5, what=k)


@pytest.mark.parametrize("name", list(MLEVEL_CASES))
def test_multi_level_loop(golden_dir, name):
    """The wrapper's level loop over levels of unequal token count with one shared narration input (cross_f_box_wrapper.py:177-212):
    the oracle's per-level restatement, looped, against the reference's fused maps and gradients -- the summed gradient of the shared
    language tokens included."""
    cfg = MLEVEL_CASES[name]
    g = _load(golden_dir, name)
    lang = torch.from_numpy(g["in_lang"]).requires_grad_(True)
    mask = torch.from_numpy(g["in_mask"])
    loss, keep = 0.0, []
    language_f = lang                     # forward_language_f (cross_f_box_wrapper.py:202-209): level i's fused tokens feed level i + 1
    for i, lv in enumerate(cfg["levels"]):
        pre = f"l{i}/"
        sd = _sd({k[len(pre) + 6:]: v for k, v in g.items() if k.startswith(pre + "param/")}, cfg["d"])
        feat = torch.from_numpy(g[pre + "in_feat"]).requires_grad_(True)
        conv_w = torch.from_numpy(g[pre + "conv_w"]).requires_grad_(True)
        reg_w = torch.from_numpy(g[pre + "reg_w"]).requires_grad_(True)
        reg_b = torch.from_numpy(g[pre + "reg_b"]).requires_grad_(True)
        vmask = None
        if "local_k" in cfg:              # vis_mask_type "local_k" on this level's token grid (cross_f_box_wrapper.py:184; utils.py:14-30)
            vmask = O.local_visual_mask(lv["H"] // lv["p"], lv["W"] // lv["p"], cfg["local_k"])
        fused, lo = O.fusion_level_forward(feat, conv_w, sd, language_f, mask, cfg["h"], cfg["L"], reg_w, reg_b, lv["p"], lv["p"], vis_tokens_mask=vmask)
        if cfg.get("fwd_lang") == "direct":
            language_f = lo
        elif cfg.get("fwd_lang") == "sum":
            language_f = language_f + lo
        _close(fused, g[pre + "fused"], what=pre + "fused")
        _close(lo, g[pre + "lang_out"], what=pre + "lang_out")
        loss = loss + (fused * torch.from_numpy(g[pre + "cot_out"])).sum()
        keep.append((pre, sd, feat, conv_w, reg_w, reg_b))
    loss.backward()
    _close(lang.grad, g["grad_lang"], tol=5e-5, what="grad_lang")
    for pre, sd, feat, conv_w, reg_w, reg_b in keep:
        _close(feat.grad, g[pre + "grad_feat"], tol=5e-5, what=pre + "grad_feat")
        _close(conv_w.grad, g[pre + "grad_conv_w"], tol=5e-5, what=pre + "grad_conv_w")
        _close(reg_w.grad, g[pre + "grad_reg_w"], tol=5e-5, what=pre + "grad_reg_w")
        _close(reg_b.grad, g[pre + "grad_reg_b"], tol=5e-5, what=pre + "grad_reg_b")
        for k, v in g.items():
            if k.startswith(pre + "gradp/"):
                _close(sd[k[len(pre) + 6:]].grad, v, tol=5e-5, what=k)


def test_sin1d_and_local_mask(golden_dir):
    g = _load(golden_dir, "sin1d_768")
    pe = O.sin1d_table(8192, 768)
    _close(pe[0, torch.from_numpy(g["idx"])], g["rows"], tol=1e-6, what="sin1d rows")
    assert abs(pe.double().sum().item() - float(g["total"])) < 1e-2
    m = _load(golden_dir, "local_mask_3x4_k1")["mask"]
    assert np.array_equal(O.local_visual_mask(3, 4, 1).numpy(), m)


# ---- language auxiliary head (lm_layers.py) -------------------------------------------------------------------------
from cases import LM_CASES, make_lm_case  # noqa: E402


@pytest.mark.parametrize("name", list(LM_CASES))
def test_lm_head_cases(golden_dir, name):
    cfg = LM_CASES[name]
    g = _load(golden_dir, name)
    params, tokens, att, cot_noun, cot_verb = make_lm_case(cfg)
    assert all(np.array_equal(params[k], g["param/" + k]) for k in params)            # fixtures == seeded generator
    assert all(np.array_equal(t, g[f"tokens/{i}"]) for i, t in enumerate(tokens)) and np.array_equal(att, g["att_mask"])
    assert sorted(params) == list(g["state_dict_keys"])                               # the reference's parameter names
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    toks = [torch.from_numpy(t).requires_grad_(True) for t in tokens]
    if cfg["multi"]:
        out = O.lm_multi_pool_predictor(sd, toks, torch.from_numpy(att), cfg["pool"], separate=cfg["multi"] == "sep")
    else:
        out = O.lm_pool_predictor(sd, toks[0], torch.from_numpy(att), cfg["pool"])
    _close(out["noun_logits"], g["noun_logits"], what="noun_logits")
    loss = (out["noun_logits"] * torch.from_numpy(cot_noun)).sum()
    if cfg["verbs"]:
        _close(out["verb_logits"], g["verb_logits"], what="verb_logits")
        loss = loss + (out["verb_logits"] * torch.from_numpy(cot_verb)).sum()
    else:
        assert out["verb_logits"] is None
    loss.backward()
    for i, t in enumerate(toks):
        _close(t.grad, g[f"grad_tokens/{i}"], what=f"grad_tokens/{i}")
        assert np.all(t.grad.numpy()[~att] == 0)                                      # padded rows receive no gradient
    for k, v in sd.items():
        _close(v.grad, g["gradp/" + k], what="gradp/" + k)


@pytest.mark.parametrize("name", ["radam_wd", "radam_groups", "radam_sgd"])
def test_radam_oracle_matches_reference_optimizer(golden_dir, name):
    """oracle.radam_step against parameters the reference's own RAdam class produced (per-group lr, weight decay, the
    un-rectified first five steps, the SGD-degenerated mode)."""
    from cases import RADAM_CASES, make_radam_case
    cfg = RADAM_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    params, grads = make_radam_case(cfg)
    moved_early = False
    for gi, grp in enumerate(cfg["groups"]):
        lr = grp.get("lr", cfg["lr"])
        for ti in range(len(grp["shapes"])):
            p = torch.from_numpy(params[gi][ti])
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            for step in range(cfg["steps"]):
                p, m, v = O.radam_step(p, torch.from_numpy(grads[step][gi][ti]), m, v, step + 1, lr, cfg["betas"], cfg["eps"],
                                       cfg["weight_decay"], cfg["degenerated_to_sgd"])
                ref = torch.from_numpy(g[f"p/{step}/{gi}/{ti}"])
                assert (p - ref).abs().max().item() <= 2e-6 * (1 + ref.abs().max().item()), (step, gi, ti)
                if step == 2 and not torch.equal(ref, torch.from_numpy(params[gi][ti])):
                    moved_early = True
            assert torch.allclose(m, torch.from_numpy(g[f"exp_avg/{gi}/{ti}"]), rtol=1e-5, atol=1e-7)
            assert torch.allclose(v, torch.from_numpy(g[f"exp_avg_sq/{gi}/{ti}"]), rtol=1e-5, atol=1e-8)
    assert moved_early == cfg["degenerated_to_sgd"]      # only the SGD-degenerated mode moves parameters before step 6


@pytest.mark.parametrize("name", ["heads_v1", "heads_v2_bg", "heads_allbg"])
def test_heads_and_losses_oracle_matches_reference(golden_dir, name):
    """oracle.nao_heads_forward / nao_losses against the reference's head modules + its own box_loss + the trainer's criterion objects
    (tests/golden/make_golden.py::run_heads_case): logits, ttcs, the four losses and every gradient of their weighted sum."""
    from cases import HEADS_CASES, IGNORE_VERB_IDX_BG, make_heads_case
    cfg = HEADS_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    params, feats, noun, verb, ttc, reg, noun_w, verb_w = make_heads_case(cfg)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    x = torch.from_numpy(feats).requires_grad_(True)
    out = O.nao_heads_forward(sd, x)
    for k in ("box_regression", "class_logits", "verb_logits", "ttcs"):
        _close(out[k], g[k], tol=1e-5, what=k)
    losses = O.nao_losses(out, torch.from_numpy(noun), torch.from_numpy(verb), torch.from_numpy(ttc), torch.from_numpy(reg), torch.from_numpy(noun_w),
                          torch.from_numpy(verb_w), IGNORE_VERB_IDX_BG, cfg["verb_bg"], cfg["ttc_bg"], cfg["ttc_bg_val"], cfg["ttc_beta"])
    got = torch.stack([losses["box"], losses["noun"], losses["verb"], losses["ttc"]])
    assert np.abs(got.detach().numpy() - g["losses"]).max() < 1e-5 * (1 + np.abs(g["losses"]).max())
    (got * torch.from_numpy(g["cot"])).sum().backward()
    _close(x.grad, g["grad_feats"], tol=1e-5, what="grad_feats")
    for k in params:
        gr = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        _close(gr, g["gradp/" + k], tol=1e-5, what=k)


@pytest.mark.parametrize("name", ["qkv_layer", "qkv_layer_gelu_hd18"])
def test_qkv_encoder_layer_oracle_matches_reference(golden_dir, name):
    """oracle.qkv_encoder_layer against the reference QKVEncoder instance driven through its own forward statements with the vendored
    three-value attention (tests/golden/make_golden.py::_ref_qkv_forward): Nq != Nk, key padding mask, relu / gelu, head dim 18."""
    from cases import QKV_CASES, make_qkv_case
    cfg = QKV_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    params, q, kv, mask, cot = make_qkv_case(cfg)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    tq, tkv = torch.from_numpy(q).requires_grad_(True), torch.from_numpy(kv).requires_grad_(True)
    out = O.qkv_encoder_layer(sd, "", tq, tkv, cfg["h"], None if mask is None else torch.from_numpy(mask), cfg["activ"])
    _close(out, g["out"], what="out")
    (out * torch.from_numpy(cot)).sum().backward()
    _close(tq.grad, g["grad_q"], tol=5e-5, what="grad_q")
    _close(tkv.grad, g["grad_kv"], tol=5e-5, what="grad_kv")
    for k in params:
        _close(sd[k].grad, g["gradp/" + k], tol=5e-5, what=k)


def test_asymmetric_forward_oracle_matches_reference(golden_dir):
    from cases import ASYM_CASES, make_asym_case
    cfg = ASYM_CASES["asym_small"]
    g = np.load(os.path.join(golden_dir, "asym_small.npz"))
    params, x, lang, cv, cl = make_asym_case(cfg)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, cfg["d"])
    tx, tl = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
    vis, lo = O.asymmetric_forward(sd, tx, tl, cfg["h"], cfg["vis_layers"], cfg["lang_layers"], cfg["activ"])
    _close(vis, g["vis"], what="vis")
    _close(lo, g["lang"], what="lang")
    ((vis * torch.from_numpy(cv)).sum() + (lo * torch.from_numpy(cl)).sum()).backward()
    _close(tx.grad, g["grad_x"], tol=5e-5, what="grad_x")
    _close(tl.grad, g["grad_lang"], tol=5e-5, what="grad_lang")
    for k in params:
        gr = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        _close(gr, g["gradp/" + k], tol=1e-4, what=k)


@pytest.mark.parametrize("name", ["pool_mlp_tanh", "pool_mlp", "pool_plain", "pool_single"])
def test_slowfast_pooling_restatement_equals_reference(golden_dir, name):
    """oracle.slowfast_pooling against fixtures produced by the reference's own SlowFastPooling class
    (modeling/narration_embeds/datasets/slowfast_features_dsets.py:207-240, imported by make_golden.py::run_pool_case): tokens, the all-ones
    attention mask, and the gradients of the inputs and of out_mlp."""
    from cases import POOL_CASES, make_pool_case
    cfg = POOL_CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    params, xs, cot = make_pool_case(cfg)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    tx = [torch.from_numpy(x).requires_grad_(True) for x in xs]
    tokens, att = O.slowfast_pooling(sd, tx, cfg["out_tanh"])
    assert tokens.shape == g["tokens"].shape and torch.equal(att, torch.from_numpy(g["att_mask"]))
    assert (tokens.detach() - torch.from_numpy(g["tokens"])).abs().max() < 2e-6
    (tokens * torch.from_numpy(cot)).sum().backward()
    assert (torch.stack([t.grad for t in tx]) - torch.from_numpy(g["grad_x"])).abs().max() < 2e-5
    for k in params:
        assert (sd[k].grad - torch.from_numpy(g["gradp/" + k])).abs().max() < 2e-5, k
