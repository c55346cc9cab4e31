"""Generate tests/golden/*.npz from the REFERENCE itself (run in the build container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference's fusion encoder imports ``modeling.obj_detection.wrapper_utils`` only
for ``is_torch_18v`` (cross_f_box_layers.py:5-10), and that module imports torchvision,
which this image lacks; as SURVEY.md 8(c) records, the two-line predicate
(wrapper_utils.py:18-19) is supplied through ``sys.modules`` so that the reference
classes import UNMODIFIED.  Nothing from the reference is copied into the fixtures:
they hold inputs, parameter values we generated, and the reference's outputs/gradients.

Each fixture records the mode that produced it:
  eval_*   module.eval() + no_grad  (modern torch nested-tensor fast path: padded language rows are 0)
  train_*  module.train() with every dropout p = 0 (all rows computed, torch-1.9 semantics) + autograd grads
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cases import (ASYM_CASES, ENCODER_CASES, HEADS_CASES, IGNORE_VERB_IDX_BG, LEVEL_CASES, LM_CASES, MLEVEL_CASES, POOL_CASES, QKV_CASES,  # noqa: E402
                   RADAM_CASES, make_asym_case, make_encoder_inputs, make_encoder_params, make_heads_case, make_level_extras, make_lm_case,
                   make_mlevel_case, make_pool_case, make_qkv_case, make_radam_case)

REF = os.environ.get("TF_REFERENCE", "/root/reference")


def import_reference():
    sys.path.insert(0, REF)
    pkg = types.ModuleType("modeling.obj_detection")
    pkg.__path__ = []
    wu = types.ModuleType("modeling.obj_detection.wrapper_utils")
    wu.is_torch_18v = lambda v: v == "1.8.1+cu101"
    sys.modules["modeling.obj_detection"] = pkg
    sys.modules["modeling.obj_detection.wrapper_utils"] = wu
    from modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from modeling.cross_fusion import utils as ref_utils
    return CrossTransformerModuleBox, ref_utils


def build_encoder(Enc, ref_utils, cfg, p_tok=0.0, p_patch=0.0):
    pe = ref_utils.PositionalEmbeddingLayer("sin1d", 8192, cfg["d"])
    lpe = ref_utils.PositionalEmbeddingLayer(cfg["lang_pos"], 256, cfg["d"]) if cfg.get("lang_pos") else None    # wrapper :100-105
    enc = Enc(no_patches=8192, pos_embedding_layer=pe, lang_pos_embedding=lpe, num_layers=cfg["L"],
              patch_dropout=p_patch, num_heads=cfg["h"], fforward_multiplier=2, token_dropout=p_tok,
              back_to_img_fn="regroup", activ_f=cfg.get("activ", "gelu"), final_norm="ln", input_f_size=cfg["d"])
    params = make_encoder_params(cfg["seed"], cfg["d"], cfg["L"])
    missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    assert set(missing) == {"padding_mask", "pos_embedding_layer.pos_embedding"} | ({"lang_pos_embedding.pos_embedding"} if lpe else set()), missing
    assert not unexpected, unexpected
    return enc, params


def run_encoder_case(name, cfg, Enc, ref_utils):
    enc, params = build_encoder(Enc, ref_utils, cfg)
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    tx, tl = torch.from_numpy(x), torch.from_numpy(lang)
    tm = None if mask is None else torch.from_numpy(mask)
    vmask = None
    if "local_k" in cfg:
        ref_utils.cache_masks.clear()
        vmask = ref_utils.get_visual_token_mask(cfg["grid"], f"local_{cfg['local_k']}")
    out = {}
    enc.eval()
    with torch.no_grad():
        v, l, att, _ = enc(tx, tl, tm, vis_tokens_mask=vmask)
    assert att is None
    out["eval_vis"], out["eval_lang"] = v.numpy(), l.numpy()

    enc.train()
    tx2, tl2 = tx.clone().requires_grad_(True), tl.clone().requires_grad_(True)
    v, l, _, _ = enc(tx2, tl2, tm, vis_tokens_mask=vmask)
    loss = (v * torch.from_numpy(gv)).sum() + (l * torch.from_numpy(gl)).sum()
    loss.backward()
    out["train_vis"], out["train_lang"] = v.detach().numpy(), l.detach().numpy()
    out["grad_x"], out["grad_lang"] = tx2.grad.numpy(), tl2.grad.numpy()
    grads = {k: p.grad for k, p in enc.named_parameters()}
    assert grads["heatmap_token"] is None          # never used
    sd_keys = sorted(enc.state_dict().keys())
    if cfg.get("big"):
        # weights/inputs are regenerated from the seed by the tests; store sub-sampled outputs + checksums
        sub = {}
        for k in ("eval_vis", "train_vis", "grad_x"):
            sub[k + "_rows"] = out[k][:, ::14]
            sub[k + "_sum"] = np.float64(out[k].astype(np.float64).sum())
            sub[k + "_abs"] = np.float64(np.abs(out[k].astype(np.float64)).sum())
        for k in ("eval_lang", "train_lang", "grad_lang"):
            sub[k + "_rows"] = out[k][:, ::8]
        for k, g in grads.items():
            if g is not None:
                g = g.numpy()
                sub["gradp_sum/" + k] = np.float64(g.astype(np.float64).sum())
                sub["gradp_abs/" + k] = np.float64(np.abs(g.astype(np.float64)).sum())
                sub["gradp_head/" + k] = g.reshape(-1)[:256].copy()
        out = sub
    else:
        out.update({"in_x": x, "in_lang": lang, "cot_vis": gv, "cot_lang": gl})
        if mask is not None:
            out["in_mask"] = mask
        if vmask is not None:
            out["in_vis_tokens_mask"] = vmask.numpy()
        for k, v_ in params.items():
            out["param/" + k] = v_
        for k, g in grads.items():
            if g is not None:
                out["gradp/" + k] = g.numpy()
    out["state_dict_keys"] = np.array(sd_keys)
    out["pos_embedding_head"] = enc.state_dict()["pos_embedding_layer.pos_embedding"][:, :64].numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", {k: getattr(v_, "shape", None) for k, v_ in list(out.items())[:6]})


def run_level_case(name, cfg, Enc, ref_utils):
    """Drives the importable pieces in the order of cross_f_box_wrapper.py:177-212
    (Conv2d -> patchify_image(.,1,1) -> encoder -> RegroupPatchesLayerBox with live init_h/init_w)."""
    enc, params = build_encoder(Enc, ref_utils, cfg)
    B, C, H, W, p, d = cfg["B"], cfg["C"], cfg["H"], cfg["W"], cfg["p"], cfg["d"]
    Nv = (H // p) * (W // p)
    _, lang, mask, _, gl = make_encoder_inputs(cfg["seed"], B, Nv, cfg["Nl"], d, cfg["mask_lens"])
    feat, conv_w, reg_w, reg_b, gout = make_level_extras(cfg["seed"], B, C, H, W, p, d)
    conv = torch.nn.Conv2d(C, d, kernel_size=(p, p), stride=(p, p), bias=False)   # wrapper :268-274
    conv.weight.data.copy_(torch.from_numpy(conv_w))
    reg = ref_utils.RegroupPatchesLayerBox(d, 1, 1, p, p, C, 0.0, None)            # wrapper :126-137
    reg.linear.weight.data.copy_(torch.from_numpy(reg_w))
    reg.linear.bias.data.copy_(torch.from_numpy(reg_b))
    enc.train(); conv.train(); reg.train()
    tf = torch.from_numpy(feat).requires_grad_(True)
    tl = torch.from_numpy(lang).requires_grad_(True)
    tm = torch.from_numpy(mask)
    reg.init_h, reg.init_w = H, W                                                  # wrapper :180-181
    tok = ref_utils.patchify_image(conv(tf), 1, 1)                                 # wrapper :183-185
    vis, lang_out, _, _ = enc(tok, tl, tm, vis_tokens_mask=None)
    fused = reg(vis)                                                               # wrapper :211
    loss = (fused * torch.from_numpy(gout)).sum() + (lang_out * torch.from_numpy(gl)).sum()
    loss.backward()
    out = {"in_feat": feat, "in_lang": lang, "in_mask": mask, "cot_out": gout, "cot_lang": gl,
           "conv_w": conv_w, "reg_w": reg_w, "reg_b": reg_b,
           "tokens": tok.detach().numpy(), "fused": fused.detach().numpy(), "lang_out": lang_out.detach().numpy(),
           "grad_feat": tf.grad.numpy(), "grad_lang": tl.grad.numpy(),
           "grad_conv_w": conv.weight.grad.numpy(), "grad_reg_w": reg.linear.weight.grad.numpy(),
           "grad_reg_b": reg.linear.bias.grad.numpy()}
    for k, v_ in params.items():
        out["param/" + k] = v_
    for k, p_ in enc.named_parameters():
        if p_.grad is not None:
            out["gradp/" + k] = p_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", fused.shape)


def run_mlevel_case(name, cfg, Enc, ref_utils):
    """The wrapper's level loop (cross_f_box_wrapper.py:177-212) over several levels of unequal token count, driven through the importable
    reference pieces in the wrapper's order; the narration tokens and their padding mask are shared by the levels
    (forward_language_f: False), the loss takes every level's fused feature map.  Records per level: the fused map, the gradient of its
    feature map, of its patch embedding / back-projection and of EVERY encoder parameter; and the gradient of the shared narration
    tokens (the sum over the levels)."""
    lang, mask, levels = make_mlevel_case(cfg)
    B, d = cfg["B"], cfg["d"]
    tl = torch.from_numpy(lang).requires_grad_(True)
    tm = torch.from_numpy(mask)
    out = {"in_lang": lang, "in_mask": mask}
    loss = 0.0
    keep = []
    # cfg["fwd_lang"] ("sum" / "direct"): forward_language_f, wrapper :202-209 -- `language_f` is the pooling layer's output there (a
    # non-leaf tensor the wrapper adds to IN PLACE); cfg["local_k"]: vis_mask_type "local_k", wrapper :184
    language_f = tl * 1.0 if cfg.get("fwd_lang") else tl
    for i, (lv, data) in enumerate(zip(cfg["levels"], levels)):
        C, H, W, p = lv["C"], lv["H"], lv["W"], lv["p"]
        pe = ref_utils.PositionalEmbeddingLayer("sin1d", 8192, d)
        enc = Enc(no_patches=8192, pos_embedding_layer=pe, lang_pos_embedding=None, num_layers=cfg["L"], patch_dropout=0.0,
                  num_heads=cfg["h"], fforward_multiplier=2, token_dropout=0.0, back_to_img_fn="regroup", activ_f="gelu", final_norm="ln",
                  input_f_size=d)
        missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in data["params"].items()}, strict=False)
        assert set(missing) == {"padding_mask", "pos_embedding_layer.pos_embedding"} and not unexpected
        conv = torch.nn.Conv2d(C, d, kernel_size=(p, p), stride=(p, p), bias=False)     # wrapper :268-274
        conv.weight.data.copy_(torch.from_numpy(data["conv_w"]))
        reg = ref_utils.RegroupPatchesLayerBox(d, 1, 1, p, p, C, 0.0, None)              # wrapper :126-137
        reg.linear.weight.data.copy_(torch.from_numpy(data["reg_w"]))
        reg.linear.bias.data.copy_(torch.from_numpy(data["reg_b"]))
        enc.train(); conv.train(); reg.train()
        tf = torch.from_numpy(data["feat"]).requires_grad_(True)
        reg.init_h, reg.init_w = H, W                                                    # wrapper :180-181
        tok4 = conv(tf)
        vmask = None
        if "local_k" in cfg:
            ref_utils.cache_masks.clear()
            vmask = ref_utils.get_visual_token_mask(tok4.shape[2:], f"local_{cfg['local_k']}")    # wrapper :184
        tok = ref_utils.patchify_image(tok4, 1, 1)                                       # wrapper :183-185
        vis, lang_out, _, _ = enc(tok, language_f, tm, vis_tokens_mask=vmask)            # (no forwarding: the SAME language tokens on every level)
        if cfg.get("fwd_lang") == "direct":                                              # wrapper :202-209
            language_f = lang_out
        elif cfg.get("fwd_lang") == "sum":
            language_f += lang_out
        fused = reg(vis)                                                                 # wrapper :211
        loss = loss + (fused * torch.from_numpy(data["gout"])).sum()
        keep.append((tf, conv, reg, enc, fused, lang_out))
        for k, v_ in data["params"].items():
            out[f"l{i}/param/{k}"] = v_
        out.update({f"l{i}/in_feat": data["feat"], f"l{i}/cot_out": data["gout"], f"l{i}/conv_w": data["conv_w"], f"l{i}/reg_w": data["reg_w"],
                    f"l{i}/reg_b": data["reg_b"]})
    loss.backward()
    for i, (tf, conv, reg, enc, fused, lang_out) in enumerate(keep):
        out.update({f"l{i}/fused": fused.detach().numpy(), f"l{i}/lang_out": lang_out.detach().numpy(), f"l{i}/grad_feat": tf.grad.numpy(),
                    f"l{i}/grad_conv_w": conv.weight.grad.numpy(), f"l{i}/grad_reg_w": reg.linear.weight.grad.numpy(),
                    f"l{i}/grad_reg_b": reg.linear.bias.grad.numpy()})
        for k, p_ in enc.named_parameters():
            if p_.grad is not None:
                out[f"l{i}/gradp/{k}"] = p_.grad.numpy()
    out["grad_lang"] = tl.grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", [tuple(k[4].shape) for k in keep])


def run_lm_case(name, cfg):
    """The reference's language head (lm_layers.py imports only torch) on seeded tokens; outputs + autograd gradients."""
    from modeling.cross_fusion.ego_fusion import lm_layers as ref_lm
    params, tokens, att, cot_noun, cot_verb = make_lm_case(cfg)
    pooling = {"type": cfg["pool"], "ln": cfg["ln"], "repr_size": cfg["repr_size"]}
    clzz = {False: ref_lm.PoolPredictor, True: ref_lm.MultiPoolPredictor, "sep": ref_lm.MultiPoolPredictorSep}[cfg["multi"]]
    head = clzz(pooling, cfg["d"], cfg["nouns"], cfg["verbs"])
    missing, unexpected = head.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    assert not missing and not unexpected
    head.train()
    toks = [torch.from_numpy(t).requires_grad_(True) for t in tokens]
    outs = head(toks if cfg["multi"] else toks[0], torch.from_numpy(att))
    loss = (outs["noun_logits"] * torch.from_numpy(cot_noun)).sum()
    if cfg["verbs"]:
        loss = loss + (outs["verb_logits"] * torch.from_numpy(cot_verb)).sum()
    loss.backward()
    out = {"att_mask": att, "cot_noun": cot_noun, "noun_logits": outs["noun_logits"].detach().numpy(),
           "state_dict_keys": np.array(sorted(head.state_dict().keys()))}
    if cfg["verbs"]:
        out["cot_verb"] = cot_verb
        out["verb_logits"] = outs["verb_logits"].detach().numpy()
    for i, t in enumerate(toks):
        out[f"tokens/{i}"] = tokens[i]
        out[f"grad_tokens/{i}"] = t.grad.numpy()
    for k, v_ in params.items():
        out["param/" + k] = v_
    for k, p_ in head.named_parameters():
        out["gradp/" + k] = p_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", out["noun_logits"].shape)


def run_radam_case(name, cfg):
    """The reference's own optimiser class (radam_optim.py imports only math and torch) stepped on seeded parameters and
    gradients; parameters after every step and the final moments are the fixture."""
    from runner.metrics_losses.radam_optim import RAdam
    import warnings
    params, grads = make_radam_case(cfg)
    tp = [[torch.nn.Parameter(torch.from_numpy(p.copy())) for p in grp] for grp in params]
    groups = []
    for g, ps in zip(cfg["groups"], tp):
        d = {"params": ps}
        if "lr" in g:
            d["lr"] = g["lr"]
        groups.append(d)
    opt = RAdam(groups, lr=cfg["lr"], betas=cfg["betas"], eps=cfg["eps"], weight_decay=cfg["weight_decay"],
                degenerated_to_sgd=cfg["degenerated_to_sgd"])
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # the deprecated add_(Number, Tensor) overloads still execute on torch 2.10
        for step in range(cfg["steps"]):
            for gi, ps in enumerate(tp):
                for ti, p in enumerate(ps):
                    p.grad = torch.from_numpy(grads[step][gi][ti].copy())
            opt.step()
            for gi, ps in enumerate(tp):
                for ti, p in enumerate(ps):
                    out[f"p/{step}/{gi}/{ti}"] = p.detach().numpy().copy()
    for gi, ps in enumerate(tp):
        for ti, p in enumerate(ps):
            out[f"exp_avg/{gi}/{ti}"] = opt.state[p]["exp_avg"].numpy().copy()
            out[f"exp_avg_sq/{gi}/{ti}"] = opt.state[p]["exp_avg_sq"].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", len(out))


def run_heads_case(name, cfg):
    """RoI heads + losses.  The head modules are the nn.Linear / nn.Sequential(Dropout, Linear) objects the reference builds
    (faster_rcnn_wrapper.py:93-100; roi_wrappers.py:306) and the forward is its roi_wrappers.py:209-229 statement by statement;
    ``box_loss`` is the reference's own function.  runner/metrics_losses/losses.py imports, at module level,
    data_preprocessing.utils.dataset_utils (needs cv2) and runner.metrics_losses.hmap_metrics (needs torchmetrics) -- both absent
    from this image -- for two names (MAX_STD, t_unravel_index) that box_loss does not use; as for is_torch_18v above, the two
    names are supplied through sys.modules for the import and losses.py itself is imported unmodified.  The noun / verb / TTC losses
    live inside EgoNAOTrainer.training_step (ego_nao_trainer.py:307-359, a LightningModule: not importable here) and are produced
    with the criterion objects its constructor builds (abc_nao_trainer.py:53-56) following those lines."""
    import torch.nn.functional as F
    for modname, attrs in (("data_preprocessing", {}), ("data_preprocessing.utils", {}),
                           ("data_preprocessing.utils.dataset_utils", {"MAX_STD": 0.0}),
                           ("runner.metrics_losses.hmap_metrics", {"t_unravel_index": None})):
        if modname not in sys.modules:
            m = types.ModuleType(modname)
            m.__path__ = []
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[modname] = m
    from runner.metrics_losses.losses import box_loss
    params, feats, noun, verb, ttc, reg, noun_w, verb_w = make_heads_case(cfg)
    D, Cn, Cv = cfg["repr"], cfg["nouns"], cfg["verbs"]
    box_regressor = torch.nn.Sequential(torch.nn.Identity(), torch.nn.Linear(D, 4 * Cn))       # box_2_dropout = 0 -> nn.Identity (:91)
    noun_classifier, verb_classifier, ttc_pred_layer = torch.nn.Linear(D, Cn), torch.nn.Linear(D, Cv), torch.nn.Linear(D, 1)
    mods = {"box_regressor.1": box_regressor[1], "noun_classifier": noun_classifier, "verb_classifier": verb_classifier, "ttc_pred_layer": ttc_pred_layer}
    for k, m in mods.items():
        m.weight.data.copy_(torch.from_numpy(params[k + ".weight"]))
        m.bias.data.copy_(torch.from_numpy(params[k + ".bias"]))
    x = torch.from_numpy(feats).requires_grad_(True)
    box_regression = box_regressor(x)                                   # roi_wrappers.py:209
    class_logits = noun_classifier(x)                                   # :213
    verb_logits = verb_classifier(x)                                    # :219
    ttcs = F.softplus(ttc_pred_layer(x)).squeeze(-1)                    # :228-229
    t_noun, t_verb, t_ttc, t_reg = torch.from_numpy(noun), torch.from_numpy(verb), torch.from_numpy(ttc), torch.from_numpy(reg)
    # the trainer hands box_loss per-image lists (ego_nao_trainer.py:289-296): two images here
    h = cfg["R"] // 2
    l_box = box_loss(class_logits, box_regression, [t_noun[:h], t_noun[h:]], [t_reg[:h], t_reg[h:]])
    noun_criterion = torch.nn.CrossEntropyLoss(torch.from_numpy(noun_w), reduction="mean")          # abc_nao_trainer.py:53
    verb_criterion = torch.nn.CrossEntropyLoss(torch.from_numpy(verb_w), reduction="mean")          # :54
    ttc_criterion = torch.nn.SmoothL1Loss(beta=cfg["ttc_beta"])                                     # :56
    l_noun = noun_criterion(class_logits + 1e-6, t_noun)                                            # ego_nao_trainer.py:310
    zero = torch.zeros(())
    targets = t_verb
    v_targets = torch.where(targets == IGNORE_VERB_IDX_BG, Cv - 1, targets)                         # :316
    v_logits = verb_logits
    if not cfg["verb_bg"]:                                                                          # :317-320
        v_idxs = torch.where(targets != IGNORE_VERB_IDX_BG)[0]
        v_logits, v_targets = verb_logits[v_idxs], targets[v_idxs]
    l_verb = verb_criterion(v_logits + 1e-6, v_targets) if v_targets.numel() else zero              # :322 (an empty selection gives nan there)
    ttc_logits, ttc_targets = ttcs, t_ttc                                                           # :347-348
    if not cfg["ttc_bg"]:                                                                           # :349-352
        ttc_idxs = torch.where(targets != IGNORE_VERB_IDX_BG)[0]
        ttc_logits, ttc_targets = ttc_logits[ttc_idxs], ttc_targets[ttc_idxs]
    else:                                                                                           # :353-356
        ttc_targets = torch.where(ttc_targets == IGNORE_VERB_IDX_BG, cfg["ttc_bg_val"], ttc_targets.double()).float()
    l_ttc = ttc_criterion(ttc_logits, ttc_targets) if ttc_logits.shape[0] > 0 else zero             # :358-359
    cot = np.array([0.7, 1.3, 0.9, 1.1], dtype=np.float32)                                          # weights of the four losses in the total
    (cot[0] * l_box + cot[1] * l_noun + cot[2] * l_verb + cot[3] * l_ttc).backward()
    out = {"cot": cot, "losses": np.array([float(l_box), float(l_noun), float(l_verb), float(l_ttc)], dtype=np.float64),
           "box_regression": box_regression.detach().numpy(), "class_logits": class_logits.detach().numpy(),
           "verb_logits": verb_logits.detach().numpy(), "ttcs": ttcs.detach().numpy(), "grad_feats": x.grad.numpy()}
    for k, m in mods.items():
        # a head whose loss has no selected RoI gets no gradient at all (None in the reference): stored as zeros
        out["gradp/" + k + ".weight"] = np.zeros_like(params[k + ".weight"]) if m.weight.grad is None else m.weight.grad.numpy()
        out["gradp/" + k + ".bias"] = np.zeros_like(params[k + ".bias"]) if m.bias.grad is None else m.bias.grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok", out["losses"])


def _ref_qkv_forward(layer, q, k, v, key_padding_mask=None):
    """``QKVEncoder.forward`` (cross_qkv_layers.py:73-81) executed statement by statement on a reference ``QKVEncoder`` INSTANCE -- its own
    Linear / LayerNorm / Dropout submodules and its own parameters -- with ONE substitution: the call ``self.self_attn(q, k, v, ...)`` of
    :73-75 unpacks three values, which on this torch raises (stock ``nn.MultiheadAttention`` returns two; checked: ValueError); the
    three-value attention is the reference's vendored ``MultiheadAttentionBFirst`` (torch18_adapters.py:270-345), which cannot be
    instantiated on torch 2.x (``_LinearWithBias``).  Its forward is therefore followed line by line here (batch_first transposes :300,
    the vendored functional ``multi_head_attention_forward`` :322-339 on the instance's parameters, transpose back :341-343)."""
    from modeling.cross_fusion.ego_fusion.torch18_adapters import multi_head_attention_forward
    a = layer.self_attn
    qt, kt, vt = [t.transpose(1, 0) for t in (q, k, v)]                                   # torch18_adapters.py:300
    attn_output, attentions, vs = multi_head_attention_forward(                          # :322-339
        qt, kt, vt, a.embed_dim, a.num_heads, a.in_proj_weight, a.in_proj_bias, a.bias_k, a.bias_v, a.add_zero_attn, a.dropout,
        a.out_proj.weight, a.out_proj.bias, training=layer.training, key_padding_mask=key_padding_mask, need_weights=False, attn_mask=None)
    q2 = attn_output.transpose(1, 0)                                                      # :341
    q = q + layer.dropout1(q2)                                                            # cross_qkv_layers.py:76
    q = layer.norm1(q)                                                                    # :77
    q2 = layer.linear2(layer.dropout(layer.activation(layer.linear1(q))))                 # :78
    q = q + layer.dropout2(q2)                                                            # :79
    q = layer.norm2(q)                                                                    # :80
    return q, attentions, vs


def _load_qkv(layer, params, prefix=""):
    missing, unexpected = layer.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in params.items() if k.startswith(prefix)}, strict=True)
    assert not missing and not unexpected


def run_qkv_case(name, cfg):
    from modeling.cross_fusion.cross_qkv_layers import QKVEncoder
    params, q, kv, mask, cot = make_qkv_case(cfg)
    layer = QKVEncoder(cfg["d"], cfg["d"], cfg["h"], dim_feedforward=cfg["ff"], dropout=0.0, activation=cfg["activ"])
    _load_qkv(layer, params)
    layer.train()
    tq, tkv = torch.from_numpy(q).requires_grad_(True), torch.from_numpy(kv).requires_grad_(True)
    out, att, vs = _ref_qkv_forward(layer, tq, tkv, tkv, None if mask is None else torch.from_numpy(mask))
    assert att is None and vs is None
    (out * torch.from_numpy(cot)).sum().backward()
    res = {"out": out.detach().numpy(), "grad_q": tq.grad.numpy(), "grad_kv": tkv.grad.numpy(),
           "state_dict_keys": np.array(sorted(layer.state_dict().keys()))}
    for k, p_ in layer.named_parameters():
        res["gradp/" + k] = p_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
    print(name, "ok", res["out"].shape)


def run_asym_case(name, cfg, ref_utils):
    """``AsymmetricCrossFModuleBox.forward`` (cross_f_box_asymm.py:72-120) executed statement by statement: the class itself cannot be
    constructed (its ``super().__init__`` call passes ``pos_embedding=`` / ``final_ln=``, TypeError -- checked), so the pieces it would
    hold are built as its constructor builds them (:48-71: two reference ``QKVEncoder`` prototypes cloned with torch's ``_get_clones``, the
    reference ``PositionalEmbeddingLayer``, the kind embeddings) and its forward's statements are executed on them in order."""
    from modeling.cross_fusion.cross_qkv_layers import QKVEncoder
    from torch.nn.modules.transformer import _get_clones
    params, x, lang, cv, cl = make_asym_case(cfg)
    d, ff = cfg["d"], int(cfg["d"] * cfg["ff_mult"])
    vis = _get_clones(QKVEncoder(d, d, cfg["h"], dim_feedforward=ff, dropout=0.0, activation=cfg["activ"]), cfg["vis_layers"])      # :53-60, :70
    lng = _get_clones(QKVEncoder(d, d, cfg["h"], dim_feedforward=ff, dropout=0.0, activation=cfg["activ"]), cfg["lang_layers"])     # :61-68, :71
    for i, l_ in enumerate(vis):
        _load_qkv(l_, params, f"cross_vis_layers.{i}.")
    for i, l_ in enumerate(lng):
        _load_qkv(l_, params, f"cross_lang_layers.{i}.")
    pe = ref_utils.PositionalEmbeddingLayer("sin1d", 8192, d)
    kind_v = torch.from_numpy(params["image_kind_embedding"]).requires_grad_(True)
    kind_l = torch.from_numpy(params["lang_kind_embedding"]).requires_grad_(True)
    tx, tl = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
    n = tx.shape[1]
    xx = pe(tx)                                                                          # :75
    xx = xx + kind_v                                                                     # :76
    language_f = tl + kind_l                                                             # :80
    v_k = torch.cat((xx, language_f), dim=1)                                             # :87
    language_f, _, _ = _ref_qkv_forward(lng[0], language_f, v_k, v_k)                    # :88
    xx, _, _ = _ref_qkv_forward(vis[0], xx, v_k, v_k)                                    # :93
    for i in range(1, cfg["lang_layers"]):                                               # :97-103
        v_k = torch.cat((xx, language_f), dim=1)
        xx, _, _ = _ref_qkv_forward(vis[i], xx, v_k, v_k)
        language_f, _, _ = _ref_qkv_forward(lng[i], language_f, v_k, v_k)
    for i in range(cfg["lang_layers"], cfg["vis_layers"]):                               # :106-110
        v_k = torch.cat((xx, language_f), dim=1)
        xx, _, _ = _ref_qkv_forward(vis[i], xx, v_k, v_k)
    hmap_token = xx[:, :n]                                                               # :115 (back_to_img_fn != "token")
    ((hmap_token * torch.from_numpy(cv)).sum() + (language_f * torch.from_numpy(cl)).sum()).backward()
    res = {"vis": hmap_token.detach().numpy(), "lang": language_f.detach().numpy(), "grad_x": tx.grad.numpy(), "grad_lang": tl.grad.numpy(),
           "gradp/image_kind_embedding": kind_v.grad.numpy(), "gradp/lang_kind_embedding": kind_l.grad.numpy()}
    for i, l_ in enumerate(vis):
        for k, p_ in l_.named_parameters():
            res[f"gradp/cross_vis_layers.{i}.{k}"] = p_.grad.numpy()
    for i, l_ in enumerate(lng):
        for k, p_ in l_.named_parameters():
            res[f"gradp/cross_lang_layers.{i}.{k}"] = np.zeros_like(params[f"cross_lang_layers.{i}.{k}"]) if p_.grad is None else p_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
    print(name, "ok", res["vis"].shape, res["lang"].shape)


def run_pool_case(name, cfg):
    """The reference's tensor-in narration pooling layer, ``SlowFastPooling`` (slowfast_features_dsets.py:207-240), on seeded inputs.  Its
    module imports, at the top, ``data_preprocessing.datasets.readers.SFastFeaturesReader`` (dataset reader: cv2 / pandas file readers,
    not used by the pooling layer); as for ``is_torch_18v`` and ``losses.py`` above that ONE name is supplied through ``sys.modules`` and
    the reference file itself is imported unmodified."""
    for modname, attrs in (("data_preprocessing", {}), ("data_preprocessing.datasets", {}),
                           ("data_preprocessing.datasets.readers", {"SFastFeaturesReader": object})):
        if modname not in sys.modules:
            m = types.ModuleType(modname)
            m.__path__ = []
            sys.modules[modname] = m
        for k, v in attrs.items():
            if not hasattr(sys.modules[modname], k):
                setattr(sys.modules[modname], k, v)
    from modeling.narration_embeds.datasets.slowfast_features_dsets import SlowFastPooling
    params, xs, cot = make_pool_case(cfg)
    layer = SlowFastPooling({"strategy": "current", "out_mlp": cfg["out_mlp"], "size": cfg["size"], "out_dropout": 0.0, "out_tanh": cfg["out_tanh"]})
    assert sorted(layer.state_dict().keys()) == sorted(params.keys())
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    layer.train()
    tx = [torch.from_numpy(x).requires_grad_(True) for x in xs]
    tokens, none, att = layer(tx, pad_mask=True)
    assert none is None
    (tokens * torch.from_numpy(cot)).sum().backward()
    out = {"tokens": tokens.detach().numpy(), "att_mask": att.numpy(), "grad_x": np.stack([t.grad.numpy() for t in tx])}
    for k, p in layer.named_parameters():
        out["gradp/" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: tokens {tuple(tokens.shape)} |tokens| {tokens.abs().mean():.4f}")


def main():
    """python tests/golden/make_golden.py [case-name ...]   (no names: every fixture)"""
    only = set(sys.argv[1:])
    want = lambda name: not only or name in only
    torch.manual_seed(0)
    torch.set_num_threads(4)
    Enc, ref_utils = import_reference()
    for name, cfg in ENCODER_CASES.items():
        if want(name):
            run_encoder_case(name, cfg, Enc, ref_utils)
    for name, cfg in LEVEL_CASES.items():
        if want(name):
            run_level_case(name, cfg, Enc, ref_utils)
    for name, cfg in MLEVEL_CASES.items():
        if want(name):
            run_mlevel_case(name, cfg, Enc, ref_utils)
    for name, cfg in LM_CASES.items():
        if want(name):
            run_lm_case(name, cfg)
    for name, cfg in RADAM_CASES.items():
        if want(name):
            run_radam_case(name, cfg)
    for name, cfg in HEADS_CASES.items():
        if want(name):
            run_heads_case(name, cfg)
    for name, cfg in QKV_CASES.items():
        if want(name):
            run_qkv_case(name, cfg)
    for name, cfg in ASYM_CASES.items():
        if want(name):
            run_asym_case(name, cfg, ref_utils)
    for name, cfg in POOL_CASES.items():
        if want(name):
            run_pool_case(name, cfg)
    if want("sin1d_768"):
        # sin1d table spot values (utils.py:267-273) at the real width
        pe = ref_utils.get_sin1d_embed(8192, 768)
        np.savez_compressed(os.path.join(HERE, "sin1d_768.npz"), rows=pe[0, [0, 1, 2, 195, 4000, 8191]].numpy(),
                            idx=np.array([0, 1, 2, 195, 4000, 8191]), total=np.float64(pe.double().sum().item()))
    if want("local_mask_3x4_k1"):
        # local visual mask (utils.py:14-30)
        ref_utils.cache_masks.clear()
        np.savez_compressed(os.path.join(HERE, "local_mask_3x4_k1.npz"),
                            mask=ref_utils.get_visual_token_mask((3, 4), "local_1").numpy())


if __name__ == "__main__":
    main()
