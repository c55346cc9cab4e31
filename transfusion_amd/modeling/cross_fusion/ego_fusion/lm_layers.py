"""Language auxiliary head -- host-side mirror of the reference's
``modeling/cross_fusion/ego_fusion/lm_layers.py:5-125`` (get_lm_layer, PoolPredictor, MultiPoolPredictor,
MultiPoolPredictorSep): same constructors, same ``forward(tokens, att_mask)`` returning
``{"noun_logits", "verb_logits"}``, same parameter names (``ln.*``, ``repr_mlp.1.*``, ``mlp_noun.*``, ``mlp_verb.*``,
``predictors.{i}.*``) so reference checkpoints load.

Device path: one fused HIP kernel for mask-multiply + mean/max pooling + LayerNorm + GELU (``tf_lm_pool_fwd/bwd``,
reference :60-72 up to the GELU that opens ``repr_mlp``), then the Linear layers on the MFMA GEMM (``ops.linear``).
The class counts (87 nouns / 74 verbs for Ego4D) are not multiples of 8: ``ops.linear`` zero-pads the weight rows.
"""
from __future__ import annotations

import torch
from torch import nn

from transfusion_amd import ops


def get_lm_layer(cross_fusion_box_wrapper):
    """reference :5-27.  One class is subtracted for the background noun / verb."""
    cross_encoder_args = cross_fusion_box_wrapper.cross_encoder_args
    no_nouns = cross_fusion_box_wrapper.rcnn_model.noun_classes - 1
    no_verbs = cross_fusion_box_wrapper.rcnn_model.verb_classes - 1
    lm_args = cross_encoder_args["lm_args"]
    if lm_args["pooling"]["type"] in {"mean", "max"}:
        multi = lm_args.get("multi", False)
        if multi == True:  # noqa: E712  (the reference distinguishes True from the string "sep")
            clzz = MultiPoolPredictor
        elif multi == "sep":
            clzz = MultiPoolPredictorSep
        else:
            clzz = PoolPredictor
        return clzz(lm_args["pooling"], cross_fusion_box_wrapper.token_dim, no_nouns, no_verbs)
    raise NotImplementedError


class PoolPredictor(nn.Module):
    def __init__(self, pooling_args, token_dim, no_nouns, no_verbs):
        super().__init__()
        self.pooling_args = pooling_args
        self.token_dim = token_dim
        self.repr_size = token_dim
        self.ln = None
        self.repr_mlp = None
        self.mlp_verb = None
        if self.pooling_args["type"] not in ("mean", "max"):
            raise NotImplementedError(self.pooling_args["type"])
        if self.pooling_args.get("ln", None):
            self.ln = nn.LayerNorm(token_dim)
        if self.pooling_args.get("repr_size", None):
            # index 0 stays a GELU so the Linear keeps the reference's state-dict key ``repr_mlp.1``; the
            # activation itself runs fused in the pooling kernel
            self.repr_mlp = nn.Sequential(nn.GELU(), nn.Linear(self.token_dim, pooling_args["repr_size"]))
            self.repr_size = pooling_args["repr_size"]
        self.mlp_noun = nn.Linear(self.repr_size, no_nouns)
        if no_verbs:
            self.mlp_verb = nn.Linear(self.repr_size, no_verbs)
        self.precision = "bf16"          # "fp32": run.precision 32 (CrossFusionBoxWrapper.set_precision); the pooling kernel is fp32 either way

    def forward(self, fused_l_tokens, att_mask=None):
        ln_w = ln_b = None
        eps = 1e-5
        if self.ln is not None:
            ln_w, ln_b, eps = self.ln.weight, self.ln.bias, self.ln.eps
        features = ops.lm_pool(fused_l_tokens, att_mask, self.pooling_args["type"], ln_w, ln_b, eps,
                               gelu=self.repr_mlp is not None)
        if self.repr_mlp is not None:
            features = ops.linear(features, self.repr_mlp[1].weight, self.repr_mlp[1].bias, precision=self.precision)
        noun_logits = ops.linear(features, self.mlp_noun.weight, self.mlp_noun.bias, precision=self.precision)
        verb_logits = None
        if self.mlp_verb is not None:
            verb_logits = ops.linear(features, self.mlp_verb.weight, self.mlp_verb.bias, precision=self.precision)
        return {"noun_logits": noun_logits, "verb_logits": verb_logits}


def _mean_over_scales(outs, key):
    return torch.stack([o[key].float() for o in outs]).mean(dim=0)


class MultiPoolPredictor(PoolPredictor):
    """One shared predictor applied to the fused language tokens of every FPN level, logits averaged (reference :84-100)."""

    def forward(self, x, att_mask=None):
        outs = [super(MultiPoolPredictor, self).forward(x[i], att_mask) for i in range(len(x))]
        noun_logits = _mean_over_scales(outs, "noun_logits")
        # the reference leaves verb_logits unbound when there is no verb head (:97-100 raises); None is returned instead
        verb_logits = _mean_over_scales(outs, "verb_logits") if self.mlp_verb is not None else None
        return {"noun_logits": noun_logits, "verb_logits": verb_logits}


class MultiPoolPredictorSep(nn.Module):
    """A separate predictor per FPN level (three of them), logits averaged (reference :103-125)."""

    def __init__(self, pooling_args, token_dim, no_nouns, no_verbs):
        super().__init__()
        self.no_fpns = 3
        self.predictors = nn.ModuleList([PoolPredictor(pooling_args, token_dim, no_nouns, no_verbs) for _ in range(self.no_fpns)])
        self.no_verbs = no_verbs

    def forward(self, x, att_mask=None):
        outs = [self.predictors[i].forward(x[i], att_mask) for i in range(len(x))]
        noun_logits = _mean_over_scales(outs, "noun_logits")
        verb_logits = _mean_over_scales(outs, "verb_logits") if self.no_verbs else None
        return {"noun_logits": noun_logits, "verb_logits": verb_logits}
