"""Fusion wrapper -- host-side mirror of the reference's
``modeling/cross_fusion/ego_fusion/cross_f_box_wrapper.py:24-303`` (registry, CrossFusionBoxWrapper): same
constructor, same ``forward(x, targets)`` flow over the FPN levels, same sub-module / parameter names
(``cross_fusion_encoders.{i}``, ``patches_to_token.{i}.weight``, ``tokens_to_features.{i}.linear.*``).

Per level (reference :177-212): K1 patch embedding = device im2col + MFMA GEMM (the Conv2d with k = s = p,
bias=False, is exactly that GEMM), fused encoder, K9 back-projection GEMM + fold scatter.  The detector
(``rcnn_model``) is whatever the caller supplies through the reference's own interface
(get_dsampled_shapes / get_features_out_channels / forward_features / apply_fpn / apply_rpn_roi_on_features ...).
"""
from __future__ import annotations

import os

import torch
from torch import nn

from transfusion_amd import _lib as L
from transfusion_amd import ops
from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
from transfusion_amd.modeling.cross_fusion.ego_fusion.lm_layers import get_lm_layer
from transfusion_amd.modeling.cross_fusion.utils import (PositionalEmbeddingLayer, RegroupPatchesLayerBox,
                                                          get_visual_token_mask)
from transfusion_amd.modeling.narration_embeds.narr_pooling_layers import get_narr_pooling_layer

MAX_NUM_PATCHES = 8192


def get_cross_box_encoder(cross_type, class_token_only):
    """reference :24-38.  Only the default joint-attention encoder is built."""
    if cross_type == "cross_transformer":
        if class_token_only:
            raise NotImplementedError("narr_out_mode: embedding (CrossTransformerTokenModule) is broken in the reference "
                                      "(cross_f_box_layers.py:143) and out of scope")
        return CrossTransformerModuleBox
    elif cross_type == "asymmetric":
        if class_token_only:
            raise NotImplementedError("AsymmetricCrossFTokenModuleBox is an empty class in the reference (cross_f_box_asymm.py:123-124)")
        from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_asymm import AsymmetricCrossFModuleBox
        return AsymmetricCrossFModuleBox
    elif cross_type == "space_time":
        raise NotImplementedError(f"type: {cross_type} is not selected by the shipped configs (SURVEY.md 2 #1: out of scope)")
    else:
        raise ValueError(f"{cross_type=} not implemented")


class PatchToToken(nn.Module):
    """``nn.Conv2d(C, d, kernel=stride=(ph, pw), bias=False)`` (reference :266-274) as a GEMM; the parameter keeps
    the conv layout [d, C, ph, pw] and the name ``weight`` so reference checkpoints load."""

    def __init__(self, in_channels, token_dim, patch_h, patch_w):
        super().__init__()
        holder = nn.Conv2d(in_channels, token_dim, kernel_size=(patch_h, patch_w), stride=(patch_h, patch_w), bias=False)
        self.weight = nn.Parameter(holder.weight.detach().clone())      # same init family as the reference
        self.patch_h, self.patch_w = patch_h, patch_w
        self.precision = "bf16"          # "fp32": run.precision 32 (set_precision)
        self.accumulate_linear_grad = False       # FusionTrainStep (one GPU): the weight gradient goes straight into .grad (ops.linear)

    def forward(self, feat):
        """[B,C,H,W] -> tokens [B, H'*W', d] (already token-major: the reference's patchify_image(.,1,1) is fused)."""
        B, Cc, H, W = feat.shape
        K = Cc * self.patch_h * self.patch_w
        if self.precision == "fp32":
            # run.precision 32: the gather splits the fp32 map into hi + lo operand planes itself, the GEMM returns fp32 tokens
            if K % 8 == 0:
                return ops.patch_embed_fp32(feat, self.weight, self.patch_h, self.patch_w)
            ph, pw = self.patch_h, self.patch_w               # (odd C p^2: torch gathers, the GEMM still runs on planes)
            Hp, Wp = H // ph, W // pw
            rows = feat[:, :, : Hp * ph, : Wp * pw].float().reshape(B, Cc, Hp, ph, Wp, pw).permute(0, 2, 4, 1, 3, 5).reshape(B * Hp * Wp, K)
            return ops.linear(rows, self.weight, None, precision="fp32").view(B, Hp * Wp, -1)
        rows = ops.patchify(feat, self.patch_h, self.patch_w, ld=(K + 63) // 64 * 64)
        tok = ops.linear(rows[:, :K] if rows.shape[1] != K else rows, self.weight, None, accumulate=self.accumulate_linear_grad)
        return tok.view(B, (H // self.patch_h) * (W // self.patch_w), -1)


class CrossFusionBoxWrapper(nn.Module):
    def __init__(self, rcnn_model, cross_layer_args, narr_embed_args, criterion=None):
        super().__init__()
        self.rcnn_model = rcnn_model
        self.narr_embed_args = narr_embed_args
        if "final_ln" in cross_layer_args["args"]:
            w_ln = cross_layer_args["args"].pop("final_ln")
            cross_layer_args["args"]["final_norm"] = "ln" if w_ln else False

        self._level_streams = None
        self.cross_encoder_args = cross_layer_args
        self.forward_language_f = self.cross_encoder_args.get("forward_language_f", False)
        self.vis_mask_type = self.cross_encoder_args.get("vis_mask_type", "global")

        self.dsampled_shapes = rcnn_model.get_dsampled_shapes()
        self.in_rgb_channels = rcnn_model.get_features_out_channels()
        self.fpn_features_idx = self.cross_encoder_args["fpn_features"][: len(self.dsampled_shapes)]
        self.vis_input_key = "image"
        self.token_dim = self.cross_encoder_args["args"]["input_f_size"]

        self.narr_pooling_layer = get_narr_pooling_layer(narr_embed_args["text_pooling"])(
            narr_embed_args, cross_layer_args["narr_out_mode"]
        )

        cross_fusion_encoders = self.setup_cross_fusion_encoders(self.cross_encoder_args)
        patches_to_token = self.setup_patches_to_token()
        tokens_to_features_layers = self.setup_token_to_features_layers()

        self.cross_fusion_encoders = nn.ModuleList(cross_fusion_encoders)
        self.patches_to_token = nn.ModuleList(patches_to_token)
        self.tokens_to_features = nn.ModuleList(tokens_to_features_layers)

        self.criterion = criterion
        if criterion.get("lm", None):
            self.lm_layer = get_lm_layer(self)         # reference :77-78
        self.lm_on = criterion.get("lm", False)
        self.use_lm_f = self.cross_encoder_args["lm_args"].get("use_lm_f", False)
        self.multi_lm = self.cross_encoder_args["lm_args"].get("multi", False) and self.lm_on and not self.use_lm_f

    def set_precision(self, precision):
        """``run.precision`` of the reference's run YAML (run_experiment.py:450): 32 -> the fp32-accuracy mode of the fusion encoders
        (Ego4Dv2 YAML), 16 / "bf16" -> bf16 compute (the Ego4Dv1 YAML's 16 is fp16 autocast in the reference)."""
        mode = "fp32" if str(precision) in ("32", "32-true", "fp32") else "bf16"
        for m in self.modules():               # the encoders, the patch-embedding / back-projection GEMMs (K1 / K9), the pooling layer's out_mlp
            if m is not self and isinstance(getattr(m, "precision", None), str):
                m.precision = mode
        return mode

    def setup_cross_fusion_encoders(self, cross_encoder_args):
        cross_fusion_encoders = []
        cross_encoder_clzz = get_cross_box_encoder(
            cross_encoder_args["type"], class_token_only=cross_encoder_args["narr_out_mode"] == "embedding"
        )
        all_num_layers = cross_encoder_args["args"].pop("num_layers")
        if not isinstance(all_num_layers, list):
            all_num_layers = [all_num_layers] * len(self.dsampled_shapes)
        for i in range(len(self.dsampled_shapes)):
            pos_embedding_layer = PositionalEmbeddingLayer(cross_encoder_args["pos_embedding"], MAX_NUM_PATCHES, self.token_dim)
            lang_pos_embedding_layer = None
            if cross_encoder_args.get("lang_pos_embedding", False):
                lang_pos_embedding_layer = PositionalEmbeddingLayer(cross_encoder_args["lang_pos_embedding"]["embedding_type"], 256, self.token_dim)
            cross_fusion_encoders.append(
                cross_encoder_clzz(no_patches=MAX_NUM_PATCHES, pos_embedding_layer=pos_embedding_layer,
                                   lang_pos_embedding=lang_pos_embedding_layer, num_layers=all_num_layers[i], **cross_encoder_args["args"])
            )
        return cross_fusion_encoders

    def setup_token_to_features_layers(self):
        tokens_to_features = []
        for i, shape in enumerate(self.dsampled_shapes):
            tokens_to_features.append(
                RegroupPatchesLayerBox(self.token_dim, shape[0], shape[1], self.cross_encoder_args["patch_h"][i],
                                       self.cross_encoder_args["patch_w"][i], self.in_rgb_channels[i],
                                       self.cross_encoder_args["backproj_dropout"], self.cross_encoder_args.get("backproj_activ_f", None))
            )
        return tokens_to_features

    def setup_patches_to_token(self):
        patches_to_token = []
        for i, _ in enumerate(self.dsampled_shapes):
            patches_to_token.append(
                self.setup_patch_to_token(self.cross_encoder_args["patch_norm"], None, self.token_dim, in_channels=self.in_rgb_channels[i],
                                          patch_h=self.cross_encoder_args["patch_h"][i], patch_w=self.cross_encoder_args["patch_w"][i])
            )
        return patches_to_token

    def setup_patch_to_token(self, patch_norm, patch_dim, token_dim, in_channels=None, patch_h=None, patch_w=None):
        if in_channels is None:
            raise NotImplementedError("flat nn.Linear patch projection is not used by the egonao path")
        if patch_norm["visual"]:
            raise NotImplementedError(f"patch_norm.visual={patch_norm['visual']!r}: the shipped configs use null")
        return PatchToToken(in_channels, token_dim, patch_h, patch_w)

    def forward(self, x, targets=None):
        visual_data = x[self.vis_input_key]
        features_dict = self.rcnn_model.forward_features(visual_data, targets)
        if self.cross_encoder_args["narr_out_mode"] == "embedding":
            raise NotImplementedError("narr_out_mode: embedding")
        language_f, att_w, att_mask = self.narr_pooling_layer(x["language_f"], pad_mask=True)
        if att_mask is None:
            raise RuntimeError("the pooling layer returned no attention mask (IdentityLayer trap, narr_pooling_layers.py:409-414)")
        pad_mask = ~(att_mask.type(torch.bool))     # HF mask (1 = token) -> torch convention (True = ignore), reference :196
        # host-side count of real language tokens, when the pooling layer knows it (SlowFastPooling.valid_tokens, set by the call
        # above): the encoders then run on the packed token rows
        n_valid = getattr(self.narr_pooling_layer, "valid_tokens", None)
        fused_l_features = None
        mscale_l_features = []
        # The feature levels are independent of each other unless language_f is forwarded from level to level (forward_language_f,
        # False in the shipped configs): each level then runs on its own HIP stream.  At the reference's real per-GPU batch (4-8 samples)
        # one level's kernels cover a fraction of the 256 CUs (grids of 50-150 workgroups); four levels side by side fill the chip.
        # Autograd replays every node's backward on the stream its forward ran on, so the backward is level-parallel too.
        main = torch.cuda.current_stream(language_f.device) if language_f.is_cuda else None
        # Better than side by side: the levels as ONE grouped encoder call (TfEncoderDesc.groups) when their encoders allow it --
        # identical shape, parameters at one common stride (FusionTrainStep's flat buffers), same token count on every level: a
        # quarter of the launches (the step is host-bound at the reference's batch sizes) on grids four times the size.
        grouped = None
        if main is not None and not self.forward_language_f and len(self.fpn_features_idx) > 1 and os.environ.get("TF_GROUP_LEVELS", "1") != "0":
            grouped = self._grouped_levels(features_dict, language_f, pad_mask, n_valid)
        if grouped is not None:
            for i, key in enumerate(self.fpn_features_idx):
                features_dict["features"][str(key)] = grouped[0][i]
                if self.multi_lm:
                    mscale_l_features.append(grouped[1][i])
            fused_l_features = grouped[1][-1]
        parallel = (grouped is None and main is not None and not self.forward_language_f and len(self.fpn_features_idx) > 1
                    and os.environ.get("TF_LEVEL_STREAMS", "1") != "0")
        # which way the levels went (read by tests/graph_step.py: check_capturable: level streams + side streams cannot be graph-captured)
        others_on_streams = grouped is not None and getattr(self, "_grouped_others", 0) > 0 and os.environ.get("TF_LEVEL_STREAMS", "1") != "0"
        self._last_path = ("streams" if others_on_streams else "grouped") if grouped is not None else ("streams" if parallel else "loop")
        if main is not None:
            # the GEMMs plan their tile grids for their share of the chip while the levels run side by side (backward included:
            # autograd replays the levels on the same streams)
            ops.set_gemm_concurrency(len(self.fpn_features_idx) if parallel else (1 + getattr(self, "_grouped_others", 0) if grouped is not None else 1))
        if parallel:
            self._refuse_nested_fork_capture()
        if parallel and (self._level_streams is None or self._level_streams[0].device != language_f.device):
            # TF_LEVEL_STREAMS = n > 1: n streams shared round-robin by the levels (default: one per level)
            n_st = int(os.environ.get("TF_LEVEL_STREAMS", "1"))
            pool = [torch.cuda.Stream(device=language_f.device) for _ in range(n_st if n_st > 1 else len(self.fpn_features_idx))]
            self._level_streams = [pool[i % len(pool)] for i in range(len(self.fpn_features_idx))]
        for i, key in enumerate(() if grouped is not None else self.fpn_features_idx):
            key = str(key)
            feat = features_dict["features"][key]
            self.tokens_to_features[i].init_h = feat.shape[2]
            self.tokens_to_features[i].init_w = feat.shape[3]
            hp, wp = feat.shape[2] // self.patches_to_token[i].patch_h, feat.shape[3] // self.patches_to_token[i].patch_w
            vis_tokens_mask = get_visual_token_mask((hp, wp), self.vis_mask_type)
            if parallel:
                st = self._level_streams[i]
                st.wait_stream(main)                              # inputs (features, language tokens, masks) were produced on main
                with torch.cuda.stream(st):
                    vis_tokens = self.patches_to_token[i](feat)
                    fused_features, fused_l_features, atts, _ = self.cross_fusion_encoders[i](
                        vis_tokens, language_f, pad_mask, vis_tokens_mask=vis_tokens_mask, **self._pack_kw(i, n_valid)
                    )
                    out_i = self.tokens_to_features[i](fused_features)
                for t in (out_i, fused_l_features):
                    t.record_stream(main)                         # consumed on main below: keep the allocator from recycling them early
            else:
                vis_tokens = self.patches_to_token[i](feat)
                fused_features, fused_l_features, atts, _ = self.cross_fusion_encoders[i](
                    vis_tokens, language_f, pad_mask, vis_tokens_mask=vis_tokens_mask, **self._pack_kw(i, n_valid)
                )
                out_i = self.tokens_to_features[i](fused_features)
            if self.multi_lm:
                mscale_l_features.append(fused_l_features)
            if self.forward_language_f:
                if self.forward_language_f == "direct":
                    language_f = fused_l_features
                elif self.forward_language_f == "sum":
                    language_f = language_f + fused_l_features
                else:
                    raise NotImplementedError()
            features_dict["features"][key] = out_i
        if parallel:
            for st in dict.fromkeys(self._level_streams):
                main.wait_stream(st)

        features_dict = self.rcnn_model.apply_fpn(features_dict)
        if "hand_boxes" in x:
            features_dict["hand_boxes"] = x["hand_boxes"]
        if "hand_poses" in x:
            features_dict["hand_poses"] = x["hand_poses"]
        rcnn_outs = self.rcnn_model.apply_rpn_roi_on_features(features_dict)
        if self.lm_on:                                  # reference :223-228
            rcnn_outs["lm"] = self.lm_layer(
                mscale_l_features if self.multi_lm else fused_l_features if not self.use_lm_f else language_f,
                att_mask.type(torch.bool),
            )
        return rcnn_outs

    @staticmethod
    def _refuse_nested_fork_capture():
        """A step whose feature levels run on LEVEL STREAMS while every level's encoder forks a side stream of its own for the weight
        gradients cannot be captured in a HIP graph on ROCm 7.2: every fork is joined -- the capture is legal by the API's rules, and
        either fork level alone captures and replays -- but hipStreamEndCapture segfaults on the nested fork (round 3,
        gpurun_out/wg.txt: a crash in the runtime, not an error code).  Raised BEFORE any level is launched, so that a user who wraps
        FusionTrainStep.step in torch.cuda.graph with the defaults gets an exception that names the ways out instead of a core dump."""
        if torch.cuda.is_current_stream_capturing() and ops.wgrad_overlap_enabled():
            raise ValueError(
                "CrossFusionBoxWrapper: this forward would run feature levels on their own streams while every level's encoder forks a side "
                "stream for its weight gradients; hipStreamEndCapture crashes on that nested fork (ROCm 7.2).  Capture with the levels on "
                "one stream (TF_LEVEL_STREAMS=0), without the side streams (TF_WGRAD_OVERLAP=0), or with ALL levels as one grouped call "
                "(parameters in FusionTrainStep's flat layout, equal token grids).")

    def _grouped_levels(self, features_dict, language_f, pad_mask, n_valid):
        """The levels that share a token grid through ONE grouped encoder call (TfEncoderDesc.groups), the others beside it on their own
        streams; None when no two levels can be grouped (then the level loop runs).  The reference's real FPN geometry -- patches of
        4, 4, 2, 1 on maps of stride 4 / 8 / 16 / 32 (cross_fusion_config_sym_ego_res50.yml:8-17) -- gives level 0 four times the tokens
        of levels 1 - 3: the grouped call then covers levels 1 - 3 and level 0 runs concurrently.
        -> (fused feature maps per level, fused language tokens per level)."""
        encs = list(self.cross_fusion_encoders)
        if not hasattr(encs[0], "group_stride") or self.vis_mask_type != "global":          # (a local visual mask: per-level block bits)
            return None
        keys = [str(k) for k in self.fpn_features_idx]
        feats = [features_dict["features"][k] for k in keys]
        nlev, B = len(encs), language_f.shape[0]
        if any(f.shape[0] != B for f in feats):
            return None
        n_tok = [(f.shape[2] // self.patches_to_token[i].patch_h) * (f.shape[3] // self.patches_to_token[i].patch_w) for i, f in enumerate(feats)]
        # ALL levels as one RAGGED grouped call (TfEncoderDesc.group_nv) when they differ in their token counts and the rows are packed
        # (the batch's un-masked token count is known): the reference's real geometry, 4N / N / N / N tokens.  TF_RAGGED_GROUPS=0: off
        members, ragged = None, False
        if (len(set(n_tok)) > 1 and n_valid is not None and getattr(encs[0], "pack_tokens", False) and pad_mask is not None
                and nlev <= L.CONSTS["TF_MAX_GROUPS"] and os.environ.get("TF_RAGGED_GROUPS", "1") != "0"
                and encs[0].group_stride(encs, ragged=True) is not None):
            members, ragged = list(range(nlev)), True
        # otherwise the largest class of levels with one token count whose encoders sit at a common stride (ties: the first)
        if members is None:
            for cnt in sorted(set(n_tok), key=lambda c: -n_tok.count(c)):
                idx = [i for i in range(nlev) if n_tok[i] == cnt]
                if len(idx) >= 2 and encs[idx[0]].group_stride([encs[i] for i in idx]) is not None:
                    members = idx
                    break
        if members is None:
            return None
        others = [i for i in range(nlev) if i not in members]
        lead, G = encs[members[0]], len(members)
        # the per-level pieces either side of the grouped call (patch embedding, back-projection + fold: different shapes per level) run
        # side by side on the level streams, forward and -- autograd replays a node on its forward stream -- backward
        main = torch.cuda.current_stream(language_f.device)
        use_streams = os.environ.get("TF_LEVEL_STREAMS", "1") != "0"
        # ... each side of the grouped call as ONE autograd node when the shapes allow it (level_ops: the host issues a level's two or three
        # kernels instead of four autograd nodes and a dozen torch / stream calls per level and direction); TF_LEVEL_OPS=0 keeps the
        # per-level modules
        from transfusion_amd import level_ops
        fused_ops = os.environ.get("TF_LEVEL_OPS", "1") != "0"
        p2t, t2f, gfeats = [self.patches_to_token[i] for i in members], [self.tokens_to_features[i] for i in members], [feats[i] for i in members]
        k1_fused = fused_ops and level_ops.k1_supported(p2t, gfeats, self.token_dim)
        # EVERY "cannot group" decision is taken here, before anything is launched (a level started on its stream and then abandoned
        # would keep reading feats / language_f / pad_mask beside the fallback loop that recomputes it): the member levels' token
        # dtypes must agree for the torch.cat of the per-level path (bf16 from the bf16 path, fp32 from the fp32-accuracy mode)
        if not k1_fused and len({getattr(m, "precision", "bf16") for m in p2t}) != 1:
            return None
        if use_streams and others:
            self._refuse_nested_fork_capture()
        if use_streams and (self._level_streams is None or self._level_streams[0].device != language_f.device):
            self._level_streams = [torch.cuda.Stream(device=language_f.device) for _ in self.fpn_features_idx]

        def per_level(fn, levels, args):
            if not use_streams:
                return [fn(i, a) for i, a in zip(levels, args)]
            res = []
            for i, a in zip(levels, args):
                st = self._level_streams[i]
                st.wait_stream(main)
                # `a` was allocated on main and is about to be read (and possibly SAVED for the backward) on the level stream: tell the
                # allocator, or the block is recycled the moment the last reference drops -- see ops._LinearFn.forward
                a.record_stream(st)
                with torch.cuda.stream(st):
                    r = fn(i, a)
                r.record_stream(main)
                res.append(r)
            for i in levels:
                main.wait_stream(self._level_streams[i])
            return res
        for i, feat in enumerate(feats):
            self.tokens_to_features[i].init_h, self.tokens_to_features[i].init_w = feat.shape[2], feat.shape[3]
        outs, fused_ls = [None] * nlev, [None] * nlev
        # ---- the levels outside the group: a whole level each (patch embedding, encoder, back-projection) on its stream, started first so
        #      that it runs beside the grouped call
        self._grouped_others = len(others)
        if others:
            ops.set_gemm_concurrency(1 + len(others))
        for i in others:
            def whole_level(i=i):
                vis_tokens = self.patches_to_token[i](feats[i])
                fused_features, fl, _, _ = self.cross_fusion_encoders[i](vis_tokens, language_f, pad_mask, vis_tokens_mask=None,
                                                                         **self._pack_kw(i, n_valid))
                return self.tokens_to_features[i](fused_features), fl
            if use_streams:
                st = self._level_streams[i]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    outs[i], fused_ls[i] = whole_level()
                outs[i].record_stream(main)
                fused_ls[i].record_stream(main)
            else:
                outs[i], fused_ls[i] = whole_level()
        # The fused K1 / K9 nodes of the group's members run on the MAIN stream by default (TF_K_LEVEL_STREAMS=1: on the level streams, as
        # in rounds 3 - 5).  Side by side the levels' gathers and GEMMs did overlap on the device, but every stream region costs the host
        # ~50 us (wait_stream, the stream context, record_stream) and the legs follow the host's enqueue time: same box, three runs each,
        # on / off: wrapper_b4 4.94 / 4.72 ms, wrapper_b4_real 5.58 / 5.53 ms (gpurun_out/r6_levelstreams.txt) -- since round 6 the
        # permutations are 6x shorter and the levels' weight gradients one launch, so there is less to overlap
        sts = [self._level_streams[i] for i in members] if (use_streams and os.environ.get("TF_K_LEVEL_STREAMS", "0") != "0") else None
        if k1_fused:
            x = level_ops.levels_patch_embed(p2t, gfeats, sts)
        else:
            toks = per_level(lambda i, feat: self.patches_to_token[i](feat), members, gfeats)
            x = torch.cat([t.reshape(-1, t.shape[-1]) for t in toks], dim=0) if ragged else torch.cat(toks, dim=0)   # [G * B, Nv, d], group-major
        lang_g, pad_g = language_f.repeat(G, 1, 1), pad_mask.repeat(G, 1)
        kw = {}
        if n_valid is not None and getattr(lead, "pack_tokens", False):
            kw["lang_valid_rows"] = G * n_valid
        if ragged:
            kw["group_nv"] = [n_tok[i] for i in members]
        fused, fused_l, _, _ = lead.forward_grouped([encs[i] for i in members], x, lang_g, pad_g, **kw)
        if fused_ops and level_ops.k9_supported(t2f, self.token_dim):
            gouts = level_ops.levels_back_project(t2f, fused, sts)
        elif ragged:                                               # fused: the concatenation [sum_g B nv_g, d]
            offs = [0]
            for i in members:
                offs.append(offs[-1] + B * n_tok[i])
            gouts = per_level(lambda i, f: self.tokens_to_features[i](f), members,
                              [fused[offs[k]:offs[k + 1]].view(B, n_tok[i], fused.shape[-1]) for k, i in enumerate(members)])
        else:
            gouts = per_level(lambda i, f: self.tokens_to_features[i](f), members, [fused[k * B:(k + 1) * B] for k in range(G)])
        for k, (i, fl) in enumerate(zip(members, fused_l.chunk(G, dim=0))):
            outs[i], fused_ls[i] = gouts[k], fl
        if use_streams:
            for i in others:
                main.wait_stream(self._level_streams[i])
        return outs, fused_ls

    def _pack_kw(self, i, n_valid):
        """``lang_valid_rows`` for encoders that can drop masked tokens (the joint-attention encoder; the asymmetric variant attends over
        padded language keys by construction, cross_f_box_asymm.py:85-93, and takes none).  The fused language tokens of masked
        positions then come back as zeros -- every consumer here multiplies them by the mask (lm_layers.py:59-61) or, with
        forward_language_f, feeds them to the next level under the same mask."""
        if n_valid is None or not getattr(self.cross_fusion_encoders[i], "pack_tokens", False):
            return {}
        return {"lang_valid_rows": n_valid}

    def call_model_epoch_triggers(self, epoch):
        if epoch >= self.narr_embed_args["train_ep"] and self.narr_embed_args["train_ep"] != -1:
            self.narr_pooling_layer.unfreeze_embeddings()
        self.rcnn_model.call_model_epoch_triggers(epoch)

    def dets_from_outs(self, outs, orig_img_shapes=None, targets=None, hand_poses=None, hand_boxes=None):
        return self.rcnn_model.dets_from_outs(outs, orig_img_shapes, targets=targets, hand_poses=hand_poses, hand_boxes=hand_boxes)

    def forward_w_dets(self, x, targets=None):
        outs = self(x, targets)
        original_image_shapes = [tuple(img.shape[1:]) for img in x["image"]]
        return self.dets_from_outs(outs, orig_img_shapes=original_image_shapes, targets=targets, hand_poses=x.get("hand_poses"),
                                   hand_boxes=x.get("hand_boxes"))

    def postprocess_detections(self, detections, proposals, image_sizes, original_image_shapes):
        return self.rcnn_model.postprocess_detections(detections, proposals, image_sizes, original_image_shapes)

    def compute_rpn_loss(self, objectness, pred_bbox_deltas, labels, regression_targets):
        return self.rcnn_model.compute_rpn_loss(objectness, pred_bbox_deltas, labels, regression_targets)
