"""Asymmetric fusion encoder -- host-side mirror of the reference's ``AsymmetricCrossFModuleBox``
(``modeling/cross_fusion/ego_fusion/cross_f_box_asymm.py:10-120``; YAML ``type: asymmetric``): two stacks of cross-attention
layers (``QKVEncoder``), one refining the visual tokens and one the language tokens, each attending over the CONCATENATION of the
current visual and language tokens.

The reference class cannot run as shipped, on any torch: its constructor hands ``pos_embedding=`` / ``final_ln=`` to a parent that
has no such keywords (:31-46, TypeError), ``QKVEncoder.forward`` unpacks three values from a two-value attention call
(cross_qkv_layers.py:73-75), ``forward`` takes three positional arguments and returns a 3-tuple where the wrapper passes
``vis_tokens_mask=`` and unpacks four (cross_f_box_wrapper.py:195-197), and the padding mask it builds (:85-86) is never handed to a
layer.  This mirror keeps the constructor signature (the two dead keywords are accepted and ignored), the parameter names
(``cross_vis_layers.{i}.*``, ``cross_lang_layers.{i}.*``, the kind embeddings, ``heatmap_token``), the layer schedule of :88-112
statement by statement -- including attending over padded language keys, since that is what the shipped arithmetic does -- and adapts
only the call contract to the wrapper's: ``forward(x, language_f, mask, vis_tokens_mask=None) -> (vis, lang, None, None)``.
"""
from __future__ import annotations

import torch
from torch import nn

from transfusion_amd.modeling.cross_fusion.cross_qkv_layers import QKVEncoder


class AsymmetricCrossFModuleBox(nn.Module):
    """Used for cross fusion using all tokens from the LM encoder"""

    def __init__(self, no_patches, patch_dropout, input_f_size, pos_embedding_layer, vis_layers=3, lang_layers=2, num_heads=4,
                 fforward_multiplier=1, vis_dropout=0.1, lang_dropout=0.1, back_to_img_fn="token", activ_f="relu", pos_embedding="learned",
                 patch_norm=False, final_ln=False, lang_pos_embedding=None, num_layers=None, final_norm=False, **unused):
        super().__init__()
        if lang_layers > vis_layers:
            raise ValueError("the reference assumes at least as many visual as language layers (cross_f_box_asymm.py:105)")
        self.no_patches = no_patches
        self.back_to_img_fn = back_to_img_fn
        self.token_dim = input_f_size
        self.pos_embedding_layer = pos_embedding_layer
        self.lang_pos_embedding = lang_pos_embedding
        self.image_kind_embedding = nn.Parameter(torch.randn(1, 1, self.token_dim))
        self.lang_kind_embedding = nn.Parameter(torch.randn(1, 1, self.token_dim))
        self.heatmap_token = nn.Parameter(torch.randn(1, 1, self.token_dim))     # created by the parent constructor, never used
        self.patch_dropout = patch_dropout
        self.register_buffer("padding_mask", torch.zeros(size=(1,), dtype=torch.bool))
        self.no_vis_layers, self.no_lang_layers = vis_layers, lang_layers
        self.get_attentions = False
        ff = int(self.token_dim * fforward_multiplier)
        first_v = QKVEncoder(self.token_dim, self.token_dim, num_heads, dim_feedforward=ff, dropout=vis_dropout, activation=activ_f)
        first_l = QKVEncoder(self.token_dim, self.token_dim, num_heads, dim_feedforward=ff, dropout=lang_dropout, activation=activ_f)

        def clones(first, n, drop):                              # torch's _get_clones deep-copies: identical initial weights (:70-71)
            layers = [first]
            for _ in range(n - 1):
                nxt = QKVEncoder(self.token_dim, self.token_dim, num_heads, dim_feedforward=ff, dropout=drop, activation=activ_f)
                nxt.load_state_dict(first.state_dict())
                layers.append(nxt)
            return nn.ModuleList(layers)

        self.cross_vis_layers = clones(first_v, vis_layers, vis_dropout)
        self.cross_lang_layers = clones(first_l, lang_layers, lang_dropout)

    def forward(self, x, language_f, language_tokens_att_maks=None, vis_tokens_mask=None):
        if vis_tokens_mask is not None:
            raise NotImplementedError("vis_mask_type local_k with the asymmetric encoder: the reference forward takes no visual mask")
        bs, n, _ = x.shape
        x = self.pos_embedding_layer(x)                                                  # :75
        x = x + self.image_kind_embedding                                                # :76
        x = nn.functional.dropout(x, self.patch_dropout, self.training)                  # :77
        language_f = language_f + self.lang_kind_embedding                               # :80
        # (:85-86 build a padding mask that no layer receives: padded language keys ARE attended, as in the reference)
        v_k = torch.cat((x, language_f), dim=1)                                          # :87
        language_f, _, _ = self.cross_lang_layers[0](language_f, v_k, v_k)               # :88
        x, _, _ = self.cross_vis_layers[0](x, v_k, v_k)                                  # :93
        for i in range(1, self.no_lang_layers):                                          # :97-103
            v_k = torch.cat((x, language_f), dim=1)
            x, _, _ = self.cross_vis_layers[i](x, v_k, v_k)
            language_f, _, _ = self.cross_lang_layers[i](language_f, v_k, v_k)
        for i in range(self.no_lang_layers, self.no_vis_layers):                         # :106-110
            v_k = torch.cat((x, language_f), dim=1)
            x, _, _ = self.cross_vis_layers[i](x, v_k, v_k)
        vis = x[:, 0] if self.back_to_img_fn == "token" else x[:, :n]                    # :112-115
        return vis, language_f, None, None
