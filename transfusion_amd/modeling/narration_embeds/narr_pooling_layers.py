"""Language-side pooling layers -- the registry of the reference's
``modeling/narration_embeds/narr_pooling_layers.py:23-33`` with the one member that needs no third-party
language model: a tensor-in pooling layer with the contract of ``SlowFastPooling``
(modeling/narration_embeds/datasets/slowfast_features_dsets.py:207-240; selected by
``narration_embeds.slowfast_f: True``, run_experiment.py:90-92).  It takes a list of [T, size] embedding tensors
(precomputed narration / clip embeddings), applies ``out_mlp`` (Linear size -> out_mlp) on the MFMA GEMM,
optional tanh, L2 normalisation over the token axis and ``out_dropout``, and returns
``(tokens [B,T,d], None, hf_mask [B,T])`` with the HuggingFace mask convention (1 = real token).

SBERT / GPT-2 / T5 layers wrap pretrained models that must be fetched by name: out of scope.
Ragged inputs are right-padded and the mask marks the padding (the reference's ``torch.stack`` requires equal T).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from transfusion_amd import ops

LEARNABLE_LM = {"sbert_finetune", "gpt2", "t5-wikihow", "slowfast"}


class SlowFastPooling(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        cfg = args[0]
        self.no_prev = cfg["strategy"]
        out_mlp = cfg["out_mlp"]
        self.size = cfg["size"]
        self.out_dropout = nn.Dropout(cfg["out_dropout"])
        self.use_out_tanh = cfg["out_tanh"]
        self.out_mlp = nn.Linear(self.size, out_mlp) if out_mlp else None
        self.precision = "bf16"          # "fp32": run.precision 32 (CrossFusionBoxWrapper.set_precision)

    def unfreeze_embeddings(self):
        pass

    def forward(self, tensor, *args, **kwargs):
        lens = [t.shape[0] for t in tensor]
        T = max(lens)
        if min(lens) != T:
            tensor = [F.pad(t, (0, 0, 0, T - t.shape[0])) for t in tensor]
        tensor = torch.stack(tensor, dim=0)
        # HF style: 1 = keep, 0 = cancelled; built on the host in one piece (no per-sample device writes)
        # (on a GPU the lengths travel through pinned memory without blocking: a pageable copy would stall the host until the
        # stream has drained, once per step)
        cached = getattr(self, "_lens_cache", None)
        if cached is not None and cached[0] == lens and cached[1].device == tensor.device:
            # same lengths as the previous call: the device copy is still right.  (Also what lets a step with fixed-shape inputs be
            # captured in a HIP graph -- a host-to-device copy of per-call host data cannot be.)
            lens_t = cached[1]
        else:
            lens_t = torch.tensor(lens, dtype=torch.int32)
            if tensor.is_cuda:
                lens_t = lens_t.pin_memory().to(tensor.device, non_blocking=True)
            self._lens_cache = (list(lens), lens_t)
        att_mask = (torch.arange(T, device=tensor.device, dtype=torch.int32).unsqueeze(0) < lens_t.unsqueeze(1)).to(torch.float32)
        if self.out_mlp:
            tensor = ops.linear(tensor, self.out_mlp.weight, self.out_mlp.bias, precision=self.precision)
        # tanh, zeroing of the padded rows (ragged extension: they would be tanh(bias) and weigh on the token-axis norm of every real
        # token), F.normalize(p=2, dim=1) and out_dropout in ONE kernel each way (tf_pool_norm_fwd / bwd); device tensors only, like
        # every other op of this package
        p_out = float(self.out_dropout.p) if self.training else 0.0
        tensor = ops.pool_norm(tensor, lens_t if min(lens) != T else None, self.use_out_tanh, p_out)
        # the number of real tokens in the batch, known here on the host: with it the fusion encoders drop the padded tokens from their
        # row-wise kernels (CrossTransformerModuleBox.forward(..., lang_valid_rows=...)) instead of carrying them as dead rows.  It is
        # kept on the LAYER (the reference's three-value return stays as it is; an attribute hung on the mask tensor would be lost by
        # the first op applied to the mask): the wrapper reads `valid_tokens` right after this call.
        self.valid_tokens = int(sum(lens))
        return tensor, None, att_mask


class IdentityLayer(torch.nn.Identity):
    def forward(self, tensor, *args, **kwargs):
        return tensor, None, None


def get_narr_pooling_layer(typey):
    if typey == "slowfast":
        return SlowFastPooling
    elif typey in ("sbert_finetune", "gpt2", "t5-wikihow"):
        raise NotImplementedError(f"text_pooling={typey!r} wraps a pretrained language model fetched by name; out of scope "
                                  "(feed precomputed embeddings with narration_embeds.slowfast_f: True)")
    else:
        return IdentityLayer
