"""RoI heads of the NAO detector and their losses on the MI355X -- what sits between the fused features and the
scalar the optimiser sees.  The detector itself (torchvision Faster R-CNN: backbone, FPN, RPN, RoIAlign, box_head) is out of
scope; these heads take the [R, representation_size] box features it would hand over.

Mirrors, with the reference's parameter names (checkpoint keys ``...roi_heads.{box_regressor.1, noun_classifier, verb_classifier,
ttc_pred_layer}.{weight, bias}``):
  modeling/obj_detection/faster_rcnn_wrapper.py:91-106   box_dropout, box_regressor = Sequential(box_dropout, Linear(repr, 4*Cn)),
                                                         noun_classifier, verb_classifier
  modeling/obj_detection/roi_wrappers.py:57-58, 306      classif_dropout, ttc_pred_layer = Linear(repr, 1)
  modeling/obj_detection/roi_wrappers.py:204-242         forward: box_regression, class_logits, verb_logits, ttcs = softplus(.)
  runner/metrics_losses/losses.py:98-135, runner/nao/ego_nao_trainer.py:307-359, runner/abc_nao_trainer.py:53-56   the losses
Device side: two MFMA GEMMs (box head; noun | verb | ttc concatenated) + one row kernel each way (csrc/heads.hip).
"""
from __future__ import annotations

import torch
from torch import nn

from transfusion_amd import _lib as L
from transfusion_amd import ops

IGNORE_VERB_IDX_BG = 999          # modeling/obj_detection/roi_wrappers.py:21


class NaoRoIHeads(nn.Module):
    def __init__(self, representation_size=1024, noun_classes=88, verb_classes=75, box_2_dropout=0.0, classif_dropout=0.0, ttc_pred=True):
        super().__init__()
        assert noun_classes or verb_classes                        # faster_rcnn_wrapper.py:89
        if not noun_classes:
            raise NotImplementedError("verb-only heads (noun_classes = 0) are not used by the egonao configs")
        self.representation_size = representation_size
        self.box_dropout = nn.Dropout(box_2_dropout) if box_2_dropout else nn.Identity()
        self.box_regressor = nn.Sequential(self.box_dropout, nn.Linear(representation_size, 4 * noun_classes))
        self.noun_classifier = nn.Linear(representation_size, noun_classes)
        self.verb_classifier = nn.Linear(representation_size, verb_classes) if verb_classes else None
        self.classify_verb = bool(verb_classes)
        self.classif_dropout = nn.Dropout(classif_dropout) if classif_dropout else nn.Identity()
        self.ttc_pred = ttc_pred
        if ttc_pred:
            self.ttc_pred_layer = nn.Linear(representation_size, 1)
        self.precision = "bf16"          # "fp32": run.precision 32 -- the fp32-accuracy mode of the GEMMs, fp32 logits

    def forward(self, box_features):
        """box_features [R, repr] (or [R, repr, 1, 1], flattened as roi_wrappers.py:205-207 does) -> the reference's dict."""
        if box_features.dim() == 4:
            assert list(box_features.shape[2:]) == [1, 1]
        box_features = box_features.flatten(start_dim=1)
        p_box = self.box_dropout.p if self.training and isinstance(self.box_dropout, nn.Dropout) else 0.0
        p_cls = self.classif_dropout.p if self.training and isinstance(self.classif_dropout, nn.Dropout) else 0.0
        lin = self.box_regressor[1]
        box_regression = ops.linear(box_features, lin.weight, lin.bias, p_drop_in=p_box, precision=self.precision)
        ws, bs = [self.noun_classifier.weight], [self.noun_classifier.bias]
        Cn, Cv = self.noun_classifier.out_features, 0
        if self.verb_classifier is not None:
            ws.append(self.verb_classifier.weight); bs.append(self.verb_classifier.bias)
            Cv = self.verb_classifier.out_features
        if self.ttc_pred:
            ws.append(self.ttc_pred_layer.weight); bs.append(self.ttc_pred_layer.bias)
        cls = ops.linear(box_features, torch.cat(ws, 0), torch.cat(bs, 0), p_drop_in=p_cls, precision=self.precision,
                         weight_sources=ws)   # one GEMM: noun | verb | ttc; bf16 shadows cached on the three Parameters
        ttcs = ops.softplus_col(cls, Cn + Cv) if self.ttc_pred else None
        return {"class_logits": cls[:, :Cn], "verb_logits": cls[:, Cn:Cn + Cv] if Cv else None, "ttcs": ttcs,
                "box_regression": box_regression, "box_features": box_features, "_cls": cls, "_dims": (Cn, Cv)}


class NaoHeadLosses(nn.Module):
    """The four loss terms of EgoNAOTrainer.training_step (ego_nao_trainer.py:289-359) on the heads' outputs.  ``forward`` takes the
    dict of ``NaoRoIHeads`` and the per-image label lists the reference's ``select_training_samples`` produces."""

    def __init__(self, noun_weights, verb_weights=None, verb_bg=False, ttc_bg=False, ttc_bg_val=0.0, ttc_beta=1.0):
        super().__init__()
        self.register_buffer("noun_w", torch.as_tensor(noun_weights, dtype=torch.float32))         # abc_nao_trainer.py:53-54
        self.register_buffer("verb_w", None if verb_weights is None else torch.as_tensor(verb_weights, dtype=torch.float32))
        self.verb_bg, self.ttc_bg, self.ttc_bg_val, self.ttc_beta = verb_bg, ttc_bg, float(ttc_bg_val), float(ttc_beta)

    def forward(self, roi_outputs, noun_labels, verb_labels=None, ttc_targets=None, reg_targets=None):
        cat = lambda t: None if t is None else (torch.cat(list(t), dim=0) if isinstance(t, (list, tuple)) else t)
        noun, verb, ttc, reg = cat(noun_labels), cat(verb_labels), cat(ttc_targets), cat(reg_targets)
        Cn, Cv = roi_outputs["_dims"]
        losses = ops.nao_head_losses(roi_outputs["_cls"], roi_outputs["box_regression"], roi_outputs["ttcs"], Cn, Cv, noun, verb, ttc, reg,
                                     self.noun_w, self.verb_w, IGNORE_VERB_IDX_BG, self.verb_bg, self.ttc_bg, self.ttc_bg_val, self.ttc_beta)
        return {"bbox_loss": losses[0], "noun_loss": losses[1], "verb_loss": losses[2], "ttc_loss": losses[3]}
