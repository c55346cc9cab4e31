"""CPU oracle for the TransFusion cross-fusion hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``transfusion_amd``) never routes through this file and
fails loudly when the HIP library is missing.

It is a restatement -- explicit arithmetic in plain PyTorch CPU ops (no
``nn.TransformerEncoder``, no ``nn.MultiheadAttention``, no ``F.fold``) -- of
the reference algorithm; every function cites the reference file:line it
follows (paths relative to the reference checkout).  The arithmetic that lives
in the reference's third-party dependency torch==1.9.1 (requirements.txt:239)
is restated from the reference's own vendored copy of it,
``modeling/cross_fusion/ego_fusion/torch18_adapters.py``.

Parity pin: the reference holds NO golden vectors or known-answer tests for
this path.  The oracle is therefore pinned against outputs
of the reference itself, imported in the build container, and committed as
fixtures under ``tests/golden/`` by ``tests/golden/make_golden.py``
(``tests/test_oracle_golden.py`` checks them on CPU).

All functions are differentiable torch code, so oracle gradients come from
autograd over the restated forward.  Dropout is expressed through explicit
keep-masks (``masks`` dict) so that a device RNG stream can be replayed here.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

LN_EPS = 1e-5  # torch18_adapters.py:63 (layer_norm_eps default)


# ----------------------------------------------------------------------------
# token plumbing
# ----------------------------------------------------------------------------
def sin1d_table(no_embeds: int, dim: int) -> torch.Tensor:
    """modeling/cross_fusion/utils.py:267-273 (get_sin1d_embed) -> [1, n, dim]."""
    position = torch.arange(no_embeds).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2) * (-math.log(10000.0) / dim))
    pe = torch.zeros(no_embeds, dim)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def patch_embed(feat: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """Conv2d(k=s=p, bias=False) + patchify_image(.,1,1) as one GEMM.

    cross_f_box_wrapper.py:266-274 (conv), :183-185; utils.py:35-39.
    feat [B,C,H,W], weight [d,C,ph,pw] -> tokens [B, H'*W', d] (h outer, w inner).
    """
    B, C, H, W = feat.shape
    d, _, ph, pw = weight.shape
    Hp, Wp = H // ph, W // pw
    x = feat[:, :, : Hp * ph, : Wp * pw].reshape(B, C, Hp, ph, Wp, pw)
    x = x.permute(0, 2, 4, 1, 3, 5).reshape(B, Hp * Wp, C * ph * pw)  # K order = (c, i, j)
    return x @ weight.reshape(d, C * ph * pw).t()


def regroup(tokens: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, init_h: int, init_w: int,
            ph: int, pw: int, keep_mask: Optional[torch.Tensor] = None, p_drop: float = 0.0) -> torch.Tensor:
    """RegroupPatchesLayerBox.forward with identity activation / norm.

    utils.py:114-119 (dropout -> linear) and :42-46 (transpose + F.fold with
    kernel == stride, a pure permutation; the border not covered by whole
    patches stays zero).  tokens [B,Nv,d], weight [ph*pw*C, d] -> [B,C,init_h,init_w].
    """
    if keep_mask is not None:
        tokens = tokens * keep_mask / (1.0 - p_drop)
    y = tokens @ weight.t() + bias  # [B, Nv, C*ph*pw], column order (c, i, j) as F.fold expects
    B, Nv, CK = y.shape
    C = CK // (ph * pw)
    Hp, Wp = init_h // ph, init_w // pw
    assert Hp * Wp == Nv, (Hp, Wp, Nv)
    y = y.reshape(B, Hp, Wp, C, ph, pw).permute(0, 3, 1, 4, 2, 5).reshape(B, C, Hp * ph, Wp * pw)
    out = y.new_zeros(B, C, init_h, init_w)
    out[:, :, : Hp * ph, : Wp * pw] = y
    return out


def local_visual_mask(h: int, w: int, k: int) -> torch.Tensor:
    """utils.py:14-30 (get_visual_token_mask "local_k"): [h*w, h*w], 1 = blocked.

    Row i = query token (r, c); the clamped (2k+1)^2 window around it is 0.
    """
    mask = torch.ones(h * w, h, w)
    for i in range(h * w):
        c0, r0 = i % w, i // w
        for j1 in range(-k, k + 1):
            for j2 in range(-k, k + 1):
                c = max(0, min(c0 + j1, w - 1))
                r = max(0, min(r0 + j2, h - 1))
                mask[i, r, c] = 0
    return mask.flatten(1)


# ----------------------------------------------------------------------------
# transformer arithmetic
# ----------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.LayerNorm over the last dim, biased variance, eps 1e-5 (torch18_adapters.py:80-81)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * w + b


def gelu(x: torch.Tensor) -> torch.Tensor:
    """Exact erf GELU (activ_f: gelu, cross_fusion_config_sym_ego_res50.yml:39)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def _drop(x, masks, key, p):
    if masks is None or key not in masks or p == 0.0:
        return x
    return x * masks[key].to(x.dtype) / (1.0 - p)


def mha(x: torch.Tensor, in_w, in_b, out_w, out_b, num_heads: int,
        key_padding_mask: Optional[torch.Tensor], attn_mask: Optional[torch.Tensor] = None,
        masks: Optional[Dict[str, torch.Tensor]] = None, prefix: str = "", p: float = 0.0) -> torch.Tensor:
    """Self-attention of one encoder layer.

    torch18_adapters.py:647-700 (_in_projection_packed: one [3d,d] linear, chunked
    q|k|v), :530-540 (reshape to heads), :578-597 (key padding -> -inf additive
    mask, OR-ed with a bool attn_mask), :788-799 (q/sqrt(hd) . k^T + mask ->
    softmax -> dropout -> . v), :606-608 (merge heads, out_proj).
    x [B,S,d]; key_padding_mask [B,S] bool True = ignore; attn_mask [S,S] bool True = blocked.
    """
    B, S, d = x.shape
    hd = d // num_heads
    qkv = x @ in_w.t() + in_b
    q, k, v = qkv.split(d, dim=-1)

    def heads(t):
        return t.reshape(B, S, num_heads, hd).permute(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    scores = (q / math.sqrt(hd)) @ k.transpose(-2, -1)  # [B,h,S,S]
    neg = torch.zeros(B, 1, S, S, dtype=x.dtype)
    if key_padding_mask is not None:
        neg = neg.masked_fill(key_padding_mask.view(B, 1, 1, S), float("-inf"))
    if attn_mask is not None:
        neg = neg.masked_fill(attn_mask.view(1, 1, S, S), float("-inf"))
    scores = scores + neg
    m = scores.max(dim=-1, keepdim=True).values
    e = torch.exp(scores - m)
    prob = e / e.sum(dim=-1, keepdim=True)
    prob = _drop(prob, masks, prefix + "attn", p)
    o = (prob @ v).permute(0, 2, 1, 3).reshape(B, S, d)
    return o @ out_w.t() + out_b


def encoder_layer(x, sd: Dict[str, torch.Tensor], pre: str, num_heads: int, key_padding_mask, attn_mask=None,
                  masks=None, p: float = 0.0, activation: str = "gelu"):
    """Post-norm TransformerEncoderLayer, torch18_adapters.py:108-113.

    src = norm1(src + dropout1(attn)); src = norm2(src + dropout2(W2 . dropout(gelu(W1 . src)))).
    """
    a = mha(x, sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"],
            sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"], num_heads,
            key_padding_mask, attn_mask, masks, pre, p)
    x = layer_norm(x + _drop(a, masks, pre + "dropout1", p), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    pre_act = x @ sd[pre + "linear1.weight"].t() + sd[pre + "linear1.bias"]
    hdn = gelu(pre_act) if activation == "gelu" else torch.clamp(pre_act, min=0.0)     # activ_f: "gelu" | "relu" (cross_f_box_layers.py:26,56)
    hdn = _drop(hdn, masks, pre + "dropout", p)
    y = hdn @ sd[pre + "linear2.weight"].t() + sd[pre + "linear2.bias"]
    return layer_norm(x + _drop(y, masks, pre + "dropout2", p), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])


def encoder_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, lang: torch.Tensor,
                    lang_pad_mask: Optional[torch.Tensor], num_heads: int, num_layers: int,
                    vis_tokens_mask: Optional[torch.Tensor] = None, final_norm: bool = True,
                    masks: Optional[Dict[str, torch.Tensor]] = None, token_dropout: float = 0.0,
                    patch_dropout: float = 0.0, activation: str = "gelu"):
    """CrossTransformerModuleBox.forward, cross_f_box_layers.py:69-108.

    sd uses the reference's state_dict names.  x [B,Nv,d],
    lang [B,Nl,d], lang_pad_mask [B,Nl] bool True = ignore.
    Returns (vis [B,Nv,d], lang [B,Nl,d]).  Every row is computed (train-mode /
    torch-1.9 semantics); modern torch's eval fast path zeroes padded rows, so
    parity on language rows is asserted only where the mask is False.
    """
    B, Nv, d = x.shape
    Nl = lang.shape[1]
    pe = sd["pos_embedding_layer.pos_embedding"]
    x = x + pe[:, :Nv]                                   # utils.py:209-214
    x = x + sd["image_kind_embedding"]                   # :73
    x = _drop(x, masks, "patch", patch_dropout)          # :74
    lang = lang + sd["lang_kind_embedding"]              # :76
    if "lang_pos_embedding.pos_embedding" in sd:         # :77-78 (PositionalEmbeddingLayer.forward, utils.py:209-214)
        lang = lang + sd["lang_pos_embedding.pos_embedding"][:, :Nl]
    kpm = None
    if lang_pad_mask is not None:                        # :80-82
        kpm = torch.cat([torch.zeros(B, Nv, dtype=torch.bool), lang_pad_mask], dim=1)
    attn_mask = None
    if vis_tokens_mask is not None:                      # :87-95
        S = Nv + Nl
        attn_mask = torch.zeros(S, S, dtype=torch.bool)
        attn_mask[:Nv, :Nv] = vis_tokens_mask.to(torch.bool)
    h = torch.cat([x, lang], dim=1)                      # :86
    for j in range(num_layers):                          # :97
        h = encoder_layer(h, sd, f"t_encoder.layers.{j}.", num_heads, kpm, attn_mask, masks, token_dropout, activation)
    vis = h[:, :Nv]
    if final_norm:                                       # :104-107
        vis = layer_norm(vis, sd["final_norm_layer.weight"], sd["final_norm_layer.bias"])
    return vis, h[:, Nv:]


def fusion_level_forward(feat: torch.Tensor, conv_w: torch.Tensor, enc_sd, lang, lang_pad_mask, num_heads, num_layers,
                         reg_w, reg_b, ph: int, pw: int, vis_tokens_mask=None):
    """One iteration of the per-FPN-level loop, cross_f_box_wrapper.py:177-212 (eval / p=0)."""
    H, W = feat.shape[2:]
    tok = patch_embed(feat, conv_w)
    vis, lang_out = encoder_forward(enc_sd, tok, lang, lang_pad_mask, num_heads, num_layers, vis_tokens_mask)
    # F.fold needs Nv == (H//ph)*(W//pw); the zero border comes from init_h/init_w = live H, W (:180-181)
    return regroup(vis, reg_w, reg_b, H, W, ph, pw), lang_out


# ----------------------------------------------------------------------------
# language auxiliary head
# ----------------------------------------------------------------------------
def lm_pool_predictor(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, att_mask: Optional[torch.Tensor], pool_type: str,
                      prefix: str = "") -> Dict[str, Optional[torch.Tensor]]:
    """modeling/cross_fusion/ego_fusion/lm_layers.py:59-81 (PoolPredictor.forward).  ``att_mask`` is the HF-convention
    mask (True = real token).  Which stages exist is read off the state dict, as the constructor (:47-57) creates them:
    ``ln.*`` (optional), ``repr_mlp.1.*`` (optional, preceded by a GELU), ``mlp_noun.*``, ``mlp_verb.*`` (optional)."""
    x = tokens if att_mask is None else tokens * att_mask.unsqueeze(2).to(tokens.dtype)         # :60-61
    if pool_type == "max":
        f = x.max(dim=1)[0]                                                                     # :63-64
    elif pool_type == "mean":
        f = x.sum(dim=1) / x.shape[1]                                                           # :65-66 (padded rows count)
    else:
        raise NotImplementedError(pool_type)
    if prefix + "ln.weight" in sd:
        f = layer_norm(f, sd[prefix + "ln.weight"], sd[prefix + "ln.bias"])                     # :68-69
    if prefix + "repr_mlp.1.weight" in sd:
        f = gelu(f) @ sd[prefix + "repr_mlp.1.weight"].t() + sd[prefix + "repr_mlp.1.bias"]     # :71-72
    noun = f @ sd[prefix + "mlp_noun.weight"].t() + sd[prefix + "mlp_noun.bias"]                # :74
    verb = None
    if prefix + "mlp_verb.weight" in sd:
        verb = f @ sd[prefix + "mlp_verb.weight"].t() + sd[prefix + "mlp_verb.bias"]            # :76-77
    return {"noun_logits": noun, "verb_logits": verb}


def lm_multi_pool_predictor(sd, tokens_per_scale, att_mask, pool_type: str, separate: bool = False):
    """lm_layers.py:84-100 (MultiPoolPredictor: one shared head) and :103-125 (MultiPoolPredictorSep: ``predictors.{i}.``):
    per-scale logits averaged over the scales."""
    outs = [lm_pool_predictor(sd, t, att_mask, pool_type, prefix=f"predictors.{i}." if separate else "")
            for i, t in enumerate(tokens_per_scale)]
    noun = torch.stack([o["noun_logits"] for o in outs]).mean(dim=0)
    verb = None if outs[0]["verb_logits"] is None else torch.stack([o["verb_logits"] for o in outs]).mean(dim=0)
    return {"noun_logits": noun, "verb_logits": verb}


# ----------------------------------------------------------------------------
# optimiser
# ----------------------------------------------------------------------------
def radam_step(p: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int, lr: float,
               betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, degenerated_to_sgd: bool = False):
    """One RAdam step on one tensor, runner/metrics_losses/radam_optim.py:47-100; ``step`` is the 1-based count AFTER the
    increment of :63.  Moments are always updated (:60-61); the parameter moves only when the variance is rectifiable
    (N_sma >= 5, :87-92) or in the SGD-degenerated mode (:93-97); weight decay is ``p += -wd * lr * p`` before the update.
    Returns the new (p, exp_avg, exp_avg_sq)."""
    beta1, beta2 = betas
    exp_avg_sq = exp_avg_sq * beta2 + (1 - beta2) * grad * grad            # :60
    exp_avg = exp_avg * beta1 + (1 - beta1) * grad                         # :61
    beta2_t = beta2 ** step                                                # :69
    n_sma_max = 2 / (1 - beta2) - 1                                        # :70
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)                 # :71
    if n_sma >= 5:                                                         # :75-83
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / (
            1 - beta1 ** step)
        if weight_decay != 0:
            p = p + (-weight_decay * lr) * p                               # :89
        p = p + (-step_size * lr) * exp_avg / (exp_avg_sq.sqrt() + eps)    # :90-91
    elif degenerated_to_sgd:                                               # :84-85, :93-97
        step_size = 1.0 / (1 - beta1 ** step)
        if weight_decay != 0:
            p = p + (-weight_decay * lr) * p
        p = p + (-step_size * lr) * exp_avg
    return p, exp_avg, exp_avg_sq


# ----------------------------------------------------------------------------
# RoI heads and their losses
# ----------------------------------------------------------------------------
def nao_heads_forward(sd: Dict[str, torch.Tensor], box_features: torch.Tensor, keep_box: Optional[torch.Tensor] = None,
                      keep_cls: Optional[torch.Tensor] = None, p_box: float = 0.0, p_cls: float = 0.0):
    """modeling/obj_detection/roi_wrappers.py:204-231 on box features [R, repr] (after box_head), with the heads of
    faster_rcnn_wrapper.py:93-100: box_regressor = Dropout(box_2_dropout) -> Linear(repr, 4*Cn) on the features; classif_dropout,
    then noun / verb classifiers and ttcs = softplus(ttc_pred_layer(.)).squeeze(-1) (roi_wrappers.py:228-229, 306).
    Dropout through explicit keep masks, as elsewhere in this file."""
    xb = box_features if keep_box is None else box_features * keep_box / (1.0 - p_box)
    box_regression = xb @ sd["box_regressor.1.weight"].t() + sd["box_regressor.1.bias"]              # :209
    xc = box_features if keep_cls is None else box_features * keep_cls / (1.0 - p_cls)               # :211
    out = {"box_regression": box_regression,
           "class_logits": xc @ sd["noun_classifier.weight"].t() + sd["noun_classifier.bias"], "verb_logits": None, "ttcs": None}
    if "verb_classifier.weight" in sd:
        out["verb_logits"] = xc @ sd["verb_classifier.weight"].t() + sd["verb_classifier.bias"]     # :218-221
    if "ttc_pred_layer.weight" in sd:
        z = (xc @ sd["ttc_pred_layer.weight"].t() + sd["ttc_pred_layer.bias"]).squeeze(-1)
        out["ttcs"] = torch.where(z > 20.0, z, torch.log1p(torch.exp(z)))                           # F.softplus, :229
    return out


def _smooth_l1(d: torch.Tensor, beta: float) -> torch.Tensor:
    a = d.abs()
    return torch.where(a < beta, 0.5 * d * d / beta, a - 0.5 * beta) if beta > 0 else a


def _weighted_ce(logits: torch.Tensor, targets: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """torch.nn.CrossEntropyLoss(weight, reduction="mean") (abc_nao_trainer.py:53-54): sum_i w[y_i] * nll_i / sum_i w[y_i]."""
    lse = torch.logsumexp(logits, dim=1)
    nll = lse - logits.gather(1, targets.view(-1, 1)).squeeze(1)
    w = weight[targets]
    return (w * nll).sum() / w.sum()


def nao_losses(out, noun_labels: torch.Tensor, verb_labels: Optional[torch.Tensor], ttc_targets: Optional[torch.Tensor],
               reg_targets: torch.Tensor, noun_w: torch.Tensor, verb_w: Optional[torch.Tensor], verb_ignore: int = 999,
               verb_bg: bool = False, ttc_bg: bool = False, ttc_bg_val: float = 0.0, ttc_beta: float = 1.0):
    """-> dict(box, noun, verb, ttc).  Labels / targets are the concatenation over the images of the batch.
    box: runner/metrics_losses/losses.py:98-135 (smooth-L1 beta 1/9, sum over the positive RoIs' own class slots, / number of RoIs);
    noun / verb: runner/nao/ego_nao_trainer.py:307-322 (+1e-6 on the logits, background verbs mapped to the last class or dropped);
    ttc: :347-359 with abc_nao_trainer.py:56 (SmoothL1Loss(beta=ttc_beta), mean over the kept RoIs)."""
    res = {}
    pos = torch.where(noun_labels > 0)[0]                                                           # losses.py:122
    n = out["class_logits"].shape[0]
    br = out["box_regression"].reshape(n, -1, 4)                                                    # :125
    res["box"] = _smooth_l1(br[pos, noun_labels[pos]] - reg_targets[pos], 1.0 / 9).sum() / max(noun_labels.numel(), 1)    # :127-133
    res["noun"] = _weighted_ce(out["class_logits"] + 1e-6, noun_labels, noun_w)                     # trainer :310
    zero = out["class_logits"].new_zeros(())
    res["verb"], res["ttc"] = zero, zero
    if out["verb_logits"] is not None and verb_labels is not None:
        v_targets = torch.where(verb_labels == verb_ignore, out["verb_logits"].shape[1] - 1, verb_labels)    # :316
        v_logits = out["verb_logits"]
        if not verb_bg:                                                                             # :317-320
            idx = torch.where(verb_labels != verb_ignore)[0]
            v_logits, v_targets = v_logits[idx], verb_labels[idx]
        if v_targets.numel():
            res["verb"] = _weighted_ce(v_logits + 1e-6, v_targets, verb_w)                          # :322
    if out["ttcs"] is not None and ttc_targets is not None:
        t_logits, t_targets = out["ttcs"], ttc_targets
        if not ttc_bg:                                                                              # :349-352 ("targets" there = the verb labels)
            idx = torch.where(verb_labels != verb_ignore)[0]
            t_logits, t_targets = t_logits[idx], t_targets[idx]
        else:                                                                                       # :353-356
            t_targets = torch.where(t_targets == float(verb_ignore), torch.tensor(ttc_bg_val, dtype=t_targets.dtype), t_targets)
        if t_logits.shape[0] > 0:                                                                   # :358-359
            res["ttc"] = _smooth_l1(t_logits - t_targets, ttc_beta).mean()
    return res


# ----------------------------------------------------------------------------
# asymmetric cross attention
# ----------------------------------------------------------------------------
def qkv_encoder_layer(sd: Dict[str, torch.Tensor], pre: str, q: torch.Tensor, kv: torch.Tensor, num_heads: int,
                      key_padding_mask: Optional[torch.Tensor] = None, activation: str = "relu",
                      masks: Optional[Dict[str, torch.Tensor]] = None, p: float = 0.0) -> torch.Tensor:
    """QKVEncoder.forward, modeling/cross_fusion/cross_qkv_layers.py:73-81, with k = v = kv (how AsymmetricCrossFModuleBox calls it).
    Attention: torch18_adapters.py:647-700 (_in_projection_packed with q is not k, k is v: w.split([E, 2E]); q from the first E rows of
    in_proj_weight, k | v from the other 2E), :530-540 (heads), :578-597 (key padding), :788-799 (softmax(q/sqrt(hd) k^T) v), :606-608.
    q [B,Nq,d], kv [B,Nk,d] -> [B,Nq,d]."""
    B, Nq, d = q.shape
    Nk = kv.shape[1]
    hd = d // num_heads
    w, b = sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"]
    qp = q @ w[:d].t() + b[:d]
    kp = kv @ w[d:2 * d].t() + b[d:2 * d]
    vp = kv @ w[2 * d:].t() + b[2 * d:]
    heads = lambda t, n: t.reshape(B, n, num_heads, hd).permute(0, 2, 1, 3)
    scores = (heads(qp, Nq) / math.sqrt(hd)) @ heads(kp, Nk).transpose(-2, -1)             # [B,h,Nq,Nk]
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask.view(B, 1, 1, Nk), float("-inf"))
    prob = torch.softmax(scores, dim=-1)
    prob = _drop(prob, masks, pre + "attn", p)
    o = (prob @ heads(vp, Nk)).permute(0, 2, 1, 3).reshape(B, Nq, d)
    q2 = o @ sd[pre + "self_attn.out_proj.weight"].t() + sd[pre + "self_attn.out_proj.bias"]
    x = layer_norm(q + _drop(q2, masks, pre + "dropout1", p), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])       # :76-77
    u = x @ sd[pre + "linear1.weight"].t() + sd[pre + "linear1.bias"]
    hdn = _drop(gelu(u) if activation == "gelu" else torch.clamp(u, min=0.0), masks, pre + "dropout", p)              # :78
    y = hdn @ sd[pre + "linear2.weight"].t() + sd[pre + "linear2.bias"]
    return layer_norm(x + _drop(y, masks, pre + "dropout2", p), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])     # :79-80


def asymmetric_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, lang: torch.Tensor, num_heads: int, vis_layers: int,
                       lang_layers: int, activation: str = "relu", back_to_img_fn: str = "regroup"):
    """AsymmetricCrossFModuleBox.forward, modeling/cross_fusion/ego_fusion/cross_f_box_asymm.py:72-120 (eval / p = 0).  The padding
    mask the reference builds at :85-86 is never handed to a layer: no key is masked.  Returns (vis, lang)."""
    n = x.shape[1]
    x = x + sd["pos_embedding_layer.pos_embedding"][:, :n]                               # :75
    x = x + sd["image_kind_embedding"]                                                   # :76
    lang = lang + sd["lang_kind_embedding"]                                              # :80
    v_k = torch.cat((x, lang), dim=1)                                                    # :87
    lang = qkv_encoder_layer(sd, "cross_lang_layers.0.", lang, v_k, num_heads, None, activation)        # :88
    x = qkv_encoder_layer(sd, "cross_vis_layers.0.", x, v_k, num_heads, None, activation)               # :93
    for i in range(1, lang_layers):                                                      # :97-103
        v_k = torch.cat((x, lang), dim=1)
        x = qkv_encoder_layer(sd, f"cross_vis_layers.{i}.", x, v_k, num_heads, None, activation)
        lang = qkv_encoder_layer(sd, f"cross_lang_layers.{i}.", lang, v_k, num_heads, None, activation)
    for i in range(lang_layers, vis_layers):                                             # :106-110
        v_k = torch.cat((x, lang), dim=1)
        x = qkv_encoder_layer(sd, f"cross_vis_layers.{i}.", x, v_k, num_heads, None, activation)
    return (x[:, 0] if back_to_img_fn == "token" else x[:, :n]), lang                    # :112-115


# ----------------------------------------------------------------------------
# tensor-in narration pooling layer
# ----------------------------------------------------------------------------
def slowfast_pooling(sd: Dict[str, torch.Tensor], tensors, out_tanh: bool, keep: Optional[torch.Tensor] = None, p_out: float = 0.0):
    """SlowFastPooling.forward, modeling/narration_embeds/datasets/slowfast_features_dsets.py:223-240: stack the B [T, size] tensors (:224),
    out_mlp Linear when present (:226-227; narr_pooling_layers.py:93-97 is the same projection for the SBERT layer), tanh (:229-230),
    L2 normalisation over the TOKEN axis when T > 1 (:232-233: F.normalize(p=2, dim=1), eps 1e-12), out_dropout (:235, through an explicit
    keep mask here), and an all-ones HuggingFace-style attention mask (:238).  -> (tokens [B, T, d], att_mask [B, T])."""
    x = torch.stack(list(tensors), dim=0)
    if "out_mlp.weight" in sd:
        x = x @ sd["out_mlp.weight"].t() + sd["out_mlp.bias"]
    if out_tanh:
        x = torch.tanh(x)
    if x.shape[1] > 1:
        x = x / x.norm(p=2, dim=1, keepdim=True).clamp_min(1e-12)
    if keep is not None:
        x = x * keep / (1.0 - p_out)
    return x, torch.ones(x.shape[:2])
