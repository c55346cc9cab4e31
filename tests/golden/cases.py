"""Seeded case definitions shared by make_golden.py (run once, with the reference
importable) and the tests (run anywhere).  Inputs and weights come from numpy's
MT19937 ``RandomState`` so they can be regenerated bit-identically without
storing them; the small cases store them in the fixture anyway.
"""
from __future__ import annotations

import numpy as np

# name -> config.  "mask_lens": valid language length per sample (None = no mask passed at all)
ENCODER_CASES = {
    "enc_small": dict(B=2, Nv=12, Nl=9, d=64, h=4, L=2, mask_lens=[6, 9], seed=101),
    "enc_hd18": dict(B=2, Nv=20, Nl=7, d=72, h=4, L=1, mask_lens=[7, 3], seed=102),
    "enc_nomask": dict(B=1, Nv=9, Nl=5, d=32, h=2, L=1, mask_lens=None, seed=103),
    "enc_local1": dict(B=2, Nv=20, Nl=6, d=64, h=4, L=1, mask_lens=[4, 6], seed=104, grid=(4, 5), local_k=1),
    # the reference constructor's default activation (cross_f_box_layers.py:26) and a language positional table (:77-78, wrapper :100-105)
    "enc_relu": dict(B=2, Nv=10, Nl=8, d=64, h=4, L=2, mask_lens=[8, 5], seed=108, activ="relu"),
    "enc_langpos": dict(B=2, Nv=9, Nl=7, d=32, h=2, L=1, mask_lens=[7, 4], seed=109, lang_pos="sin1d"),
    # real width, weights regenerated from the seed (not stored), outputs stored sub-sampled
    "enc_d768": dict(B=1, Nv=196, Nl=64, d=768, h=4, L=1, mask_lens=[40], seed=105, big=True),
    # the reference's TRUE widths: out_mlp = 712 (Ego4Dv1, head dim 178 -> padded to 192 inside the runtime) and
    # 896 (Ego4Dv2, head dim 224); sub-sampled like enc_d768
    "enc_d712": dict(B=2, Nv=196, Nl=128, d=712, h=4, L=1, mask_lens=[128, 57], seed=106, big=True),
    "enc_d896": dict(B=2, Nv=196, Nl=128, d=896, h=4, L=1, mask_lens=[90, 128], seed=107, big=True),
}

LEVEL_CASES = {
    # feature map [B,C,H,W], patch p -> Nv = (H//p)*(W//p); H % p != 0 exercises F.fold's zero border
    "level_p2": dict(B=2, C=8, H=9, W=10, p=2, Nl=5, d=32, h=2, L=1, mask_lens=[5, 2], seed=201),
    "level_p1": dict(B=1, C=16, H=3, W=4, p=1, Nl=4, d=32, h=4, L=2, mask_lens=[3], seed=202),
}


# The wrapper's level loop over SEVERAL FPN levels with the reference's real token geometry (cross_fusion_config_sym_ego_res50.yml:8-17:
# patches of 4, 4, 2, 1 on maps of stride 4 / 8 / 16 / 32 -> level 0 holds four times the tokens of levels 1 - 3), one shared narration
# input with padding, every level with its own encoder / patch embedding / back-projection (cross_f_box_wrapper.py:177-212).
MLEVEL_CASES = {
    "mlevel_4n_n_n_n": dict(B=2, d=64, h=4, L=2, Nl=7, mask_lens=[7, 3], seed=401,
                            levels=[dict(C=16, H=8, W=8, p=2), dict(C=16, H=4, W=4, p=2), dict(C=64, H=2, W=2, p=1), dict(C=64, H=2, W=2, p=1)]),
    # the wrapper's two switches the shipped YAML leaves at their defaults (cross_f_box_wrapper.py:184, 202-209): vis_mask_type "local_k"
    # (a visual token attends the visual tokens within k grid steps, utils.py:14-30) and forward_language_f "sum" / "direct" (the fused
    # narration tokens of level i feed level i + 1: the levels are no longer independent)
    "mlevel_local1": dict(B=2, d=64, h=4, L=2, Nl=7, mask_lens=[7, 3], seed=411, local_k=1,
                          levels=[dict(C=16, H=10, W=8, p=2), dict(C=16, H=6, W=6, p=2), dict(C=32, H=3, W=4, p=1)]),
    "mlevel_fwd_sum": dict(B=2, d=64, h=4, L=2, Nl=7, mask_lens=[5, 7], seed=421, fwd_lang="sum",
                           levels=[dict(C=16, H=8, W=8, p=2), dict(C=16, H=4, W=4, p=2), dict(C=32, H=3, W=3, p=1)]),
    "mlevel_fwd_direct_local1": dict(B=2, d=64, h=4, L=2, Nl=7, mask_lens=[7, 2], seed=431, fwd_lang="direct", local_k=1,
                                     levels=[dict(C=16, H=8, W=6, p=2), dict(C=16, H=4, W=4, p=2), dict(C=32, H=3, W=3, p=1)]),
}


def make_mlevel_case(cfg):
    """-> (lang [B, Nl, d], mask [B, Nl] bool (True = pad), per level: dict(params, feat, conv_w, reg_w, reg_b, gout))"""
    B, d, L = cfg["B"], cfg["d"], cfg["L"]
    _, lang, mask, _, _ = make_encoder_inputs(cfg["seed"], B, 1, cfg["Nl"], d, cfg["mask_lens"])
    levels = []
    for i, lv in enumerate(cfg["levels"]):
        feat, conv_w, reg_w, reg_b, gout = make_level_extras(cfg["seed"] + 10 * (i + 1), B, lv["C"], lv["H"], lv["W"], lv["p"], d)
        levels.append(dict(params=make_encoder_params(cfg["seed"] + 10 * (i + 1), d, L), feat=feat, conv_w=conv_w, reg_w=reg_w, reg_b=reg_b, gout=gout))
    return lang, mask, levels


def encoder_param_shapes(d: int, L: int, ff_mult: int = 2):
    ff = int(d * ff_mult)
    shapes = {
        "image_kind_embedding": (1, 1, d),
        "lang_kind_embedding": (1, 1, d),
        "heatmap_token": (1, 1, d),
    }
    for j in range(L):
        p = f"t_encoder.layers.{j}."
        shapes.update({
            p + "self_attn.in_proj_weight": (3 * d, d),
            p + "self_attn.in_proj_bias": (3 * d,),
            p + "self_attn.out_proj.weight": (d, d),
            p + "self_attn.out_proj.bias": (d,),
            p + "linear1.weight": (ff, d),
            p + "linear1.bias": (ff,),
            p + "linear2.weight": (d, ff),
            p + "linear2.bias": (d,),
            p + "norm1.weight": (d,),
            p + "norm1.bias": (d,),
            p + "norm2.weight": (d,),
            p + "norm2.bias": (d,),
        })
    shapes["final_norm_layer.weight"] = (d,)
    shapes["final_norm_layer.bias"] = (d,)
    return shapes


def make_encoder_params(seed: int, d: int, L: int):
    """Non-trivial values for EVERY parameter (biases and LN affine included)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shp in encoder_param_shapes(d, L).items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "final_norm_layer.weight":
            out[name] = (1.0 + 0.1 * rs.randn(*shp)).astype(np.float32)
        elif name.endswith("weight") and len(shp) == 2:
            out[name] = (rs.randn(*shp) / np.sqrt(shp[1])).astype(np.float32)
        elif "kind_embedding" in name or name == "heatmap_token":
            out[name] = rs.randn(*shp).astype(np.float32)
        else:
            out[name] = (0.1 * rs.randn(*shp)).astype(np.float32)
    return out


def make_encoder_inputs(seed: int, B: int, Nv: int, Nl: int, d: int, mask_lens):
    rs = np.random.RandomState(seed + 7919)
    x = rs.randn(B, Nv, d).astype(np.float32)
    lang = rs.randn(B, Nl, d).astype(np.float32)
    gv = rs.randn(B, Nv, d).astype(np.float32)      # cotangents
    gl = rs.randn(B, Nl, d).astype(np.float32)
    mask = None
    if mask_lens is not None:
        mask = np.zeros((B, Nl), dtype=bool)        # True = pad / ignore (torch convention)
        for b, n in enumerate(mask_lens):
            mask[b, n:] = True
        gl = gl * (~mask)[..., None]                 # padded rows carry no gradient
    return x, lang, mask, gv, gl


def make_level_extras(seed: int, B, C, H, W, p, d):
    rs = np.random.RandomState(seed + 104729)
    feat = rs.randn(B, C, H, W).astype(np.float32)
    conv_w = (rs.randn(d, C, p, p) / np.sqrt(C * p * p)).astype(np.float32)
    reg_w = (rs.randn(p * p * C, d) / np.sqrt(d)).astype(np.float32)
    reg_b = (0.1 * rs.randn(p * p * C)).astype(np.float32)
    gout = rs.randn(B, C, H, W).astype(np.float32)
    return feat, conv_w, reg_w, reg_b, gout


# language auxiliary head (lm_layers.py).  "multi": False | True (shared head over the scales) | "sep"
LM_CASES = {
    "lm_mean_ln_repr": dict(B=3, Nl=7, d=32, pool="mean", ln=True, repr_size=24, nouns=11, verbs=5, multi=False,
                            mask_lens=[7, 4, 1], seed=301),
    "lm_max_plain": dict(B=2, Nl=5, d=16, pool="max", ln=False, repr_size=None, nouns=87, verbs=74, multi=False,
                         mask_lens=[3, 5], seed=302),
    "lm_max_ln_noverb": dict(B=2, Nl=6, d=24, pool="max", ln=True, repr_size=None, nouns=9, verbs=0, multi=False,
                             mask_lens=[6, 2], seed=303),
    "lm_multi_mean": dict(B=2, Nl=6, d=32, pool="mean", ln=True, repr_size=16, nouns=10, verbs=6, multi=True,
                          mask_lens=[5, 6], seed=304, scales=3),
    "lm_sep_max": dict(B=2, Nl=4, d=16, pool="max", ln=False, repr_size=8, nouns=7, verbs=3, multi="sep",
                       mask_lens=[4, 2], seed=305, scales=3),
}


def lm_param_shapes(cfg, prefix=""):
    d = cfg["d"]
    r = cfg["repr_size"] or d
    shapes = {}
    if cfg["ln"]:
        shapes[prefix + "ln.weight"] = (d,)
        shapes[prefix + "ln.bias"] = (d,)
    if cfg["repr_size"]:
        shapes[prefix + "repr_mlp.1.weight"] = (r, d)
        shapes[prefix + "repr_mlp.1.bias"] = (r,)
    shapes[prefix + "mlp_noun.weight"] = (cfg["nouns"], r)
    shapes[prefix + "mlp_noun.bias"] = (cfg["nouns"],)
    if cfg["verbs"]:
        shapes[prefix + "mlp_verb.weight"] = (cfg["verbs"], r)
        shapes[prefix + "mlp_verb.bias"] = (cfg["verbs"],)
    return shapes


def make_lm_case(cfg):
    """-> (params, tokens [scales][B, Nl, d], att_mask [B, Nl] bool (True = token), cot_noun, cot_verb | None)"""
    rs = np.random.RandomState(cfg["seed"])
    shapes = {}
    if cfg["multi"] == "sep":
        for i in range(3):
            shapes.update(lm_param_shapes(cfg, f"predictors.{i}."))
    else:
        shapes = lm_param_shapes(cfg)
    params = {}
    for name, shp in shapes.items():
        if name.endswith("ln.weight"):
            params[name] = (1.0 + 0.1 * rs.randn(*shp)).astype(np.float32)
        elif len(shp) == 2:
            params[name] = (rs.randn(*shp) / np.sqrt(shp[1])).astype(np.float32)
        else:
            params[name] = (0.1 * rs.randn(*shp)).astype(np.float32)
    n_scales = cfg.get("scales", 1)
    tokens = [rs.randn(cfg["B"], cfg["Nl"], cfg["d"]).astype(np.float32) for _ in range(n_scales)]
    att = np.zeros((cfg["B"], cfg["Nl"]), dtype=bool)
    for b, n in enumerate(cfg["mask_lens"]):
        att[b, :n] = True
    cot_noun = rs.randn(cfg["B"], cfg["nouns"]).astype(np.float32)
    cot_verb = rs.randn(cfg["B"], cfg["verbs"]).astype(np.float32) if cfg["verbs"] else None
    return params, tokens, att, cot_noun, cot_verb


# fused RAdam (runner/metrics_losses/radam_optim.py): tensors per group, per-group lr, weight decay, SGD-degenerated mode.
# With beta2 = 0.999 the update is un-rectified (N_sma < 5) for steps 1-5: >= 12 steps cover both regimes.
RADAM_CASES = {
    "radam_wd": dict(steps=14, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=2e-4, degenerated_to_sgd=False,
                     groups=[dict(shapes=[(257,), (33, 7)])], seed=401),
    "radam_groups": dict(steps=12, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, degenerated_to_sgd=False,
                         groups=[dict(shapes=[(130,)], lr=2e-4), dict(shapes=[(64, 3), (5,)])], seed=402),
    "radam_sgd": dict(steps=12, lr=1e-3, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.0, degenerated_to_sgd=True,
                      groups=[dict(shapes=[(300,)])], seed=403),
}


def make_radam_case(cfg):
    """-> (params [group][tensor] fp32, grads [step][group][tensor] fp32)"""
    rs = np.random.RandomState(cfg["seed"])
    params = [[rs.randn(*shp).astype(np.float32) for shp in g["shapes"]] for g in cfg["groups"]]
    grads = [[[(rs.randn(*shp) * (0.5 + rs.rand())).astype(np.float32) for shp in g["shapes"]] for g in cfg["groups"]]
             for _ in range(cfg["steps"])]
    return params, grads


# RoI heads + losses (faster_rcnn_wrapper.py:93-100, roi_wrappers.py:204-231, losses.py:98-135, ego_nao_trainer.py:307-359).
# Class counts of the two label mappings: Ego4Dv1 88 nouns / 75 verbs, Ego4Dv2 129 / 82 (background included).
HEADS_CASES = {
    "heads_v1": dict(R=96, repr=64, nouns=88, verbs=75, verb_bg=False, ttc_bg=False, ttc_bg_val=0.0, ttc_beta=1.0, seed=501),
    "heads_v2_bg": dict(R=70, repr=48, nouns=129, verbs=82, verb_bg=True, ttc_bg=True, ttc_bg_val=2.5, ttc_beta=0.5, seed=502),
    "heads_allbg": dict(R=12, repr=32, nouns=9, verbs=5, verb_bg=False, ttc_bg=False, ttc_bg_val=0.0, ttc_beta=1.0, seed=503, all_bg=True),
}
IGNORE_VERB_IDX_BG = 999      # modeling/obj_detection/roi_wrappers.py:21


def make_heads_case(cfg):
    """-> (params, box_features [R, repr], noun_labels, verb_labels (999 = background), ttc_targets, reg_targets [R, 4], noun_w, verb_w)"""
    rs = np.random.RandomState(cfg["seed"])
    R, D, Cn, Cv = cfg["R"], cfg["repr"], cfg["nouns"], cfg["verbs"]
    params = {
        "box_regressor.1.weight": (rs.randn(4 * Cn, D) / np.sqrt(D)).astype(np.float32), "box_regressor.1.bias": (0.1 * rs.randn(4 * Cn)).astype(np.float32),
        "noun_classifier.weight": (rs.randn(Cn, D) / np.sqrt(D)).astype(np.float32), "noun_classifier.bias": (0.1 * rs.randn(Cn)).astype(np.float32),
        "verb_classifier.weight": (rs.randn(Cv, D) / np.sqrt(D)).astype(np.float32), "verb_classifier.bias": (0.1 * rs.randn(Cv)).astype(np.float32),
        "ttc_pred_layer.weight": (rs.randn(1, D) / np.sqrt(D)).astype(np.float32), "ttc_pred_layer.bias": (0.1 * rs.randn(1)).astype(np.float32),
    }
    feats = (2.0 * rs.randn(R, D)).astype(np.float32)
    noun = rs.randint(0, Cn, size=R).astype(np.int64)
    noun[rs.rand(R) < 0.4] = 0                                   # background RoIs
    if cfg.get("all_bg"):
        noun[:] = 0
    verb = rs.randint(0, Cv - 1, size=R).astype(np.int64)
    verb[noun == 0] = IGNORE_VERB_IDX_BG                         # background RoIs carry the ignore index (roi_wrappers.py)
    ttc = (rs.rand(R) * 3.0).astype(np.float32)
    ttc[noun == 0] = float(IGNORE_VERB_IDX_BG)
    reg = (0.5 * rs.randn(R, 4)).astype(np.float32)
    reg[::7] *= 0.05                                             # some residuals inside the quadratic zone of smooth-L1
    noun_w = (0.5 + rs.rand(Cn)).astype(np.float32)
    verb_w = (0.5 + rs.rand(Cv)).astype(np.float32)
    return params, feats, noun, verb, ttc, reg, noun_w, verb_w


# asymmetric encoder (type: asymmetric): cross-attention layers with their own query set (cross_qkv_layers.py, cross_f_box_asymm.py)
QKV_CASES = {
    "qkv_layer": dict(B=2, Nq=9, Nk=14, d=32, h=2, ff=48, activ="relu", mask_lens=[14, 8], seed=601),        # with a key padding mask
    "qkv_layer_gelu_hd18": dict(B=1, Nq=5, Nk=21, d=72, h=4, ff=72, activ="gelu", mask_lens=None, seed=602),  # head dim 18 -> padded 32
}
ASYM_CASES = {
    "asym_small": dict(B=2, Nv=10, Nl=7, d=32, h=2, vis_layers=3, lang_layers=2, ff_mult=1, activ="relu", seed=611),
}


def qkv_param_shapes(d, ff, prefix=""):
    return {prefix + "self_attn.in_proj_weight": (3 * d, d), prefix + "self_attn.in_proj_bias": (3 * d,),
            prefix + "self_attn.out_proj.weight": (d, d), prefix + "self_attn.out_proj.bias": (d,),
            prefix + "linear1.weight": (ff, d), prefix + "linear1.bias": (ff,), prefix + "linear2.weight": (d, ff), prefix + "linear2.bias": (d,),
            prefix + "norm1.weight": (d,), prefix + "norm1.bias": (d,), prefix + "norm2.weight": (d,), prefix + "norm2.bias": (d,)}


def _fill(rs, shapes):
    out = {}
    for name, shp in shapes.items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight"):
            out[name] = (1.0 + 0.1 * rs.randn(*shp)).astype(np.float32)
        elif len(shp) == 2:
            out[name] = (rs.randn(*shp) / np.sqrt(shp[1])).astype(np.float32)
        elif len(shp) == 3:
            out[name] = rs.randn(*shp).astype(np.float32)
        else:
            out[name] = (0.1 * rs.randn(*shp)).astype(np.float32)
    return out


def make_qkv_case(cfg):
    """-> (params, q [B,Nq,d], kv [B,Nk,d], key padding mask [B,Nk] bool | None, cotangent [B,Nq,d])"""
    rs = np.random.RandomState(cfg["seed"])
    params = _fill(rs, qkv_param_shapes(cfg["d"], cfg["ff"]))
    q = rs.randn(cfg["B"], cfg["Nq"], cfg["d"]).astype(np.float32)
    kv = rs.randn(cfg["B"], cfg["Nk"], cfg["d"]).astype(np.float32)
    cot = rs.randn(cfg["B"], cfg["Nq"], cfg["d"]).astype(np.float32)
    mask = None
    if cfg["mask_lens"] is not None:
        mask = np.zeros((cfg["B"], cfg["Nk"]), dtype=bool)
        for b, n in enumerate(cfg["mask_lens"]):
            mask[b, n:] = True
    return params, q, kv, mask, cot


def make_asym_case(cfg):
    """-> (params, x [B,Nv,d], lang [B,Nl,d], cot_vis, cot_lang)"""
    rs = np.random.RandomState(cfg["seed"])
    d, ff = cfg["d"], int(cfg["d"] * cfg["ff_mult"])
    shapes = {"image_kind_embedding": (1, 1, d), "lang_kind_embedding": (1, 1, d)}
    for i in range(cfg["vis_layers"]):
        shapes.update(qkv_param_shapes(d, ff, f"cross_vis_layers.{i}."))
    for i in range(cfg["lang_layers"]):
        shapes.update(qkv_param_shapes(d, ff, f"cross_lang_layers.{i}."))
    params = _fill(rs, shapes)
    x = rs.randn(cfg["B"], cfg["Nv"], d).astype(np.float32)
    lang = rs.randn(cfg["B"], cfg["Nl"], d).astype(np.float32)
    return params, x, lang, rs.randn(cfg["B"], cfg["Nv"], d).astype(np.float32), rs.randn(cfg["B"], cfg["Nl"], d).astype(np.float32)


# tensor-in narration pooling layer (SlowFastPooling, modeling/narration_embeds/datasets/slowfast_features_dsets.py:207-240)
POOL_CASES = {
    "pool_mlp_tanh": dict(B=3, T=6, size=40, out_mlp=64, out_tanh=True, seed=701),        # Linear + tanh + L2 normalisation over the tokens
    "pool_mlp": dict(B=2, T=5, size=72, out_mlp=32, out_tanh=False, seed=702),            # Linear + normalisation
    "pool_plain": dict(B=2, T=4, size=24, out_mlp=0, out_tanh=True, seed=703),            # no projection
    "pool_single": dict(B=3, T=1, size=16, out_mlp=24, out_tanh=False, seed=704),         # T = 1: the normalisation is skipped (:232)
}


def make_pool_case(cfg):
    """-> (params {out_mlp.weight, out_mlp.bias} or {}, list of B [T, size] tensors, cotangent [B, T, d_out])"""
    rs = np.random.RandomState(cfg["seed"])
    params = {}
    if cfg["out_mlp"]:
        params = {"out_mlp.weight": (rs.randn(cfg["out_mlp"], cfg["size"]) / np.sqrt(cfg["size"])).astype(np.float32),
                  "out_mlp.bias": (0.1 * rs.randn(cfg["out_mlp"])).astype(np.float32)}
    xs = [rs.randn(cfg["T"], cfg["size"]).astype(np.float32) for _ in range(cfg["B"])]
    cot = rs.randn(cfg["B"], cfg["T"], cfg["out_mlp"] or cfg["size"]).astype(np.float32)
    return params, xs, cot
