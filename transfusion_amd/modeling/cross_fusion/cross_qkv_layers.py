"""Cross-attention encoder layer -- host-side mirror of the reference's ``QKVEncoder``
(``modeling/cross_fusion/cross_qkv_layers.py:19-81``): queries from one token set, keys / values from another (in
``AsymmetricCrossFModuleBox`` the concatenation of both modalities), then the post-norm residual FFN of an encoder layer.

    q2 = MHA(q, k, v);  q = norm1(q + dropout1(q2));  q = norm2(q + dropout2(linear2(dropout(act(linear1(q))))))

Same constructor arguments and parameter names (``self_attn.in_proj_weight`` ... ``norm2.bias``).  The reference's ``forward`` unpacks
THREE values from ``self.self_attn(...)`` (:73-75), which only its vendored torch-1.8 attention returns; on any other torch the call
raises.  This mirror returns ``(q, None, None)`` -- what the reference returns with ``get_attentions=False`` (the vendored functional
gives ``None, None`` for weights and values then, torch18_adapters.py:644-645).

Device side: one autograd node per layer; every FLOP in libtfusion_hip.so through the per-op C-ABI entries -- Q and K|V projections
(``tf_gemm_fwd``), attention with its own query set (``TfAttnArgs.q``: Nq != Nk), out-proj + dropout + residual epilogue, LayerNorm,
FFN GEMMs with the activation / dropout / residual epilogues, and the mirrored backward (dgrad GEMMs on W^T shadows, ``tf_gemm_wgrad``,
``tf_attn_bwd``, ``tf_layernorm_bwd``).  bf16 compute (or, with ``precision = "fp32"``, hi + lo planes through the same entries), fp32
parameters / statistics / gradients.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from transfusion_amd import _lib as L
from transfusion_amd import ops

BIG = ops.BIG
_up = ops._up


class _LayerFn(torch.autograd.Function):
    """``mod.precision == "fp32"``: every bf16 tensor below is a (hi, lo) pair of planes (value = hi + lo) and every op gets both -- the
    same C-ABI entries in their fp32-accuracy mode (three MFMA passes per product, fp32 epilogues / statistics)."""

    @staticmethod
    def forward(ctx, mod, q_in, kv_in, key_padding_mask, *params):
        ops._require_cuda(q_in, kv_in, params[0])
        B, Nq, d = q_in.shape
        Nk = kv_in.shape[1]
        H, ff = mod.nhead, mod.dim_feedforward
        hd = d // H
        hdp, ffp = _up(hd, 32), _up(ff, 64)
        hq = H * hdp
        dp, ldkv = _up(hq, 64), _up(3 * hq, 64)
        Mq, Mk = B * Nq, B * Nk
        dev = q_in.device
        bf = torch.bfloat16
        sp = mod.precision == "fp32"
        W = mod._shadows(params, d, H, ff, sp)
        wl = (lambda k: W[k + "_lo"]) if sp else (lambda k: None)
        in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b = params
        p = float(mod.dropout_p) if mod.training else 0.0
        seed = ops.next_seed() if p > 0 else 0
        site = lambda which: 16 + which                      # the runtime's site numbering of layer 0 (tf_api.hip)
        act = 1 if mod.activation_name == "relu" else 0

        def buf(rows, cols, zero=False):                     # (hi, lo) planes; lo is None in the bf16 mode
            mk = torch.zeros if zero else torch.empty
            return mk(rows, cols, dtype=bf, device=dev), (mk(rows, cols, dtype=bf, device=dev) if sp else None)

        def planes(x2d):
            return ops._planes_padded(x2d, dp) if sp else (ops.to_bf16_padded(x2d, dp), None)

        xq, xq_l = planes(q_in.reshape(Mq, d))
        xkv, xkv_l = (xq, xq_l) if (kv_in is q_in) else planes(kv_in.reshape(Mk, d))
        Q, Q_l = buf(Mq, dp, zero=True)
        KV, KV_l = buf(Mk, ldkv, zero=True)                   # thirds: (unused Q) | K | V, as tf_attn_* expects
        cut = (lambda t: None if t is None else t[hq:]) if True else None
        ops.gemm(xq, W["win"], Q, hq, dp, L.TF_EPI_BIAS, bias=W["bin"], A_lo=xq_l, W_lo=wl("win"), C_lo=Q_l)
        ops.gemm(xkv, W["win"][hq:], KV[:, hq:], 2 * hq, dp, L.TF_EPI_BIAS, bias=W["bin"][hq:], A_lo=xkv_l, W_lo=cut(wl("win")),
                 C_lo=None if KV_l is None else KV_l[:, hq:])
        O, O_l = buf(Mq, dp)
        lse = torch.empty(B * H * Nq, dtype=torch.float32, device=dev)
        km = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
        drop_a = ops.drop_params(p, seed, site(1))
        bits = None
        if drop_a[0]:
            lib = L.load()
            bits = torch.empty(B * H * Nq * ((Nk + 63) // 64), dtype=torch.int64, device=dev)
            L.check(lib.tf_attn_dropmask_rows(L.ptr(bits), B * H * Nq, Nk, drop_a[1], drop_a[0], ops._stream()), "tf_attn_dropmask_rows")
        att = L.TfAttnArgs(qkv=L.ptr(KV), qkv_lo=L.ptr(KV_l), ld_qkv=ldkv, out=L.ptr(O), out_lo=L.ptr(O_l), ld_out=dp, lse=L.ptr(lse),
                           key_mask=L.ptr(km), B=B, S=Nk, H=H, HDP=hdp,
                           scale=1.0 / math.sqrt(hd), drop_thr=drop_a[0], drop_key=drop_a[1], drop_scale=drop_a[2], drop_bits=L.ptr(bits),
                           q=L.ptr(Q), q_lo=L.ptr(Q_l), ld_q=dp, Sq=Nq)
        L.call("tf_attn_fwd", att, ops._stream())
        z1, z1_l = buf(Mq, dp)
        drop_1 = ops.drop_params(p, seed, site(2))
        ops.gemm(O, W["wo"], z1, dp, dp, L.TF_EPI_BIAS_DROP_RES, bias=W["bo"], R=xq, drop=drop_1, A_lo=O_l, W_lo=wl("wo"), C_lo=z1_l, R_lo=xq_l)
        x1, x1_l = buf(Mq, dp)
        st1 = torch.empty(2, Mq, dtype=torch.float32, device=dev)
        ops.layernorm_fwd(z1, x1, n1w, n1b, st1[0], st1[1], Mq, d, x_lo=z1_l, y_lo=x1_l)
        G, G_l = buf(Mq, ffp)
        Hh, Hh_l = buf(Mq, ffp)
        drop_f = ops.drop_params(p, seed, site(3))
        ops.gemm(x1, W["w1"], G, ffp, dp, L.TF_EPI_BIAS_GELU_DROP_G, bias=W["b1"], C2=Hh, drop=drop_f, act=act, A_lo=x1_l, W_lo=wl("w1"),
                 C_lo=G_l, C2_lo=Hh_l)
        z2, z2_l = buf(Mq, dp)
        drop_2 = ops.drop_params(p, seed, site(4))
        ops.gemm(Hh, W["w2"], z2, dp, ffp, L.TF_EPI_BIAS_DROP_RES, bias=W["b2"], R=x1, drop=drop_2, A_lo=Hh_l, W_lo=wl("w2"), C_lo=z2_l, R_lo=x1_l)
        out = torch.empty(B, Nq, d, dtype=torch.float32 if sp else q_in.dtype, device=dev)
        st2 = torch.empty(2, Mq, dtype=torch.float32, device=dev)
        ops.layernorm_fwd(z2, out.view(Mq, d), n2w, n2b, st2[0], st2[1], Mq, d, x_lo=z2_l)
        ctx.saved = (xq, xkv, Q, KV, O, lse, km, bits, z1, st1, x1, G, Hh, z2, st2, W, att)
        ctx.lo = (xq_l, xkv_l, Q_l, KV_l, O_l, z1_l, x1_l, G_l, Hh_l, z2_l)
        ctx.meta = (B, Nq, Nk, d, H, ff, hd, hdp, hq, dp, ffp, ldkv, (drop_a, drop_1, drop_f, drop_2), act, q_in.dtype, kv_in.dtype, kv_in is q_in)
        ctx.mod = mod
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, g_out):
        xq, xkv, Q, KV, O, lse, km, bits, z1, st1, x1, G, Hh, z2, st2, W, att = ctx.saved
        xq_l, xkv_l, Q_l, KV_l, O_l, z1_l, x1_l, G_l, Hh_l, z2_l = ctx.lo
        B, Nq, Nk, d, H, ff, hd, hdp, hq, dp, ffp, ldkv, drops, act, q_dtype, kv_dtype, same = ctx.meta
        drop_a, drop_1, drop_f, drop_2 = drops
        in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b = ctx.params
        Mq, Mk = B * Nq, B * Nk
        dev, bf = g_out.device, torch.bfloat16
        sp = xq_l is not None
        wl = (lambda k: W[k + "_lo"]) if sp else (lambda k: None)
        col = lambda t, c0: None if t is None else t[:, c0:]

        def buf(rows, cols, zero=False):
            mk = torch.zeros if zero else torch.empty
            return mk(rows, cols, dtype=bf, device=dev), (mk(rows, cols, dtype=bf, device=dev) if sp else None)

        g_out = g_out.contiguous().view(Mq, d)
        zf = lambda t: torch.zeros_like(t, dtype=torch.float32, memory_format=torch.contiguous_format)
        g = {k: zf(t) for k, t in zip(("in_w", "in_b", "out_w", "out_b", "w1", "b1", "w2", "b2", "n1w", "n1b", "n2w", "n2b"), ctx.params)}
        # LN2 backward -> dz (residual path) and dy2 (dropout2-masked, feeds linear2's backward)
        dz, dz_l = buf(Mq, dp)
        dy, dy_l = buf(Mq, dp) if drop_2[0] else (None, None)
        ops.layernorm_bwd(z2, n2w, st2[0], st2[1], g_out, dz, g["n2w"], g["n2b"], Mq, d, dx_drop=dy, drop=drop_2, x_lo=z2_l, dx_lo=dz_l, dx_drop_lo=dy_l)
        dy2, dy2_l = (dy, dy_l) if drop_2[0] else (dz, dz_l)
        du, du_l = buf(Mq, ffp)
        ops.gemm(dy2, W["w2T"], du, ffp, dp, L.TF_EPI_MUL, R=G, A_lo=dy2_l, W_lo=wl("w2T"), C_lo=du_l, R_lo=G_l)      # dU = dH . G
        ops.wgrad(dy2, dp, Hh, ffp, g["w2"], g["b2"], n_src=d, k_src=ff, dY_lo=dy2_l, X_lo=Hh_l)
        ops.wgrad(du, ffp, x1, dp, g["w1"], g["b1"], n_src=ff, k_src=d, dY_lo=du_l, X_lo=x1_l)
        dxb, dxb_l = buf(Mq, dp)
        ops.gemm(du, W["w1T"], dxb, dp, ffp, L.TF_EPI_ADD, R=dz, A_lo=du_l, W_lo=wl("w1T"), C_lo=dxb_l, R_lo=dz_l)
        # LN1 backward
        dzb, dzb_l = buf(Mq, dp)
        dyb, dyb_l = buf(Mq, dp) if drop_1[0] else (None, None)
        ops.layernorm_bwd(z1, n1w, st1[0], st1[1], dxb, dzb, g["n1w"], g["n1b"], Mq, d, dx_drop=dyb, drop=drop_1, x_lo=z1_l, dy_lo=dxb_l,
                          dx_lo=dzb_l, dx_drop_lo=dyb_l)
        dy1, dy1_l = (dyb, dyb_l) if drop_1[0] else (dzb, dzb_l)
        ops.wgrad(dy1, dp, O, dp, g["out_w"], g["out_b"], n_src=d, cg=hd, cgp=hdp, k_src=d, dY_lo=dy1_l, X_lo=O_l)
        d_o, d_o_l = buf(Mq, dp)
        ops.gemm(dy1, W["woT"], d_o, dp, dp, L.TF_EPI_NONE, A_lo=dy1_l, W_lo=wl("woT"), C_lo=d_o_l)
        dQ, dQ_l = buf(Mq, dp, zero=True)
        dKV, dKV_l = buf(Mk, ldkv, zero=True)
        delta = torch.empty(B * H * Nq, dtype=torch.float32, device=dev)
        att.dout, att.ld_dout, att.dqkv, att.ld_dqkv, att.delta, att.dq, att.ld_dq = L.ptr(d_o), dp, L.ptr(dKV), ldkv, L.ptr(delta), L.ptr(dQ), dp
        att.dout_lo, att.dqkv_lo, att.dq_lo = L.ptr(d_o_l), L.ptr(dKV_l), L.ptr(dQ_l)
        L.call("tf_attn_bwd", att, ops._stream())
        # in_proj: rows [0, d) of the packed weight from the queries, rows [d, 3d) from the keys / values
        ops.wgrad(dQ, hq, xq, dp, g["in_w"][:d], g["in_b"][:d], rg=hd, rgp=hdp, n_src=d, k_src=d, dY_lo=dQ_l, X_lo=xq_l)
        ops.wgrad(dKV[:, hq:], 2 * hq, xkv, dp, g["in_w"][d:], g["in_b"][d:], rg=hd, rgp=hdp, n_src=2 * d, k_src=d, dY_lo=col(dKV_l, hq), X_lo=xkv_l)
        d_xq, d_xq_l = buf(Mq, dp)
        ops.gemm(dQ, W["winT"][:, :dp], d_xq, dp, dp, L.TF_EPI_ADD, R=dzb, A_lo=dQ_l, W_lo=None if not sp else W["winT_lo"][:, :dp],
                 C_lo=d_xq_l, R_lo=dzb_l)                                                    # columns [hq, dp) of dQ are zero
        d_xkv, d_xkv_l = buf(Mk, dp)
        ops.gemm(dKV[:, hq:], W["winT"][:, hq:], d_xkv, dp, 2 * hq, L.TF_EPI_NONE, A_lo=col(dKV_l, hq),
                 W_lo=None if not sp else W["winT_lo"][:, hq:], C_lo=d_xkv_l)
        need_q, need_kv = ctx.needs_input_grad[1], ctx.needs_input_grad[2]

        def unpad(h_, l_, dtype, shape):
            if l_ is None:
                return ops.from_padded(h_, d, dtype).view(shape)
            return (h_[:, :d].float() + l_[:, :d].float()).to(dtype).view(shape)

        if same:                                       # self-attention call: both gradients belong to the one input
            d_q_in = (unpad(d_xq, d_xq_l, torch.float32, (B, Nq, d)) + unpad(d_xkv, d_xkv_l, torch.float32, (B, Nq, d))).to(q_dtype) if need_q else None
            d_kv_in = None
        else:
            d_q_in = unpad(d_xq, d_xq_l, q_dtype, (B, Nq, d)) if need_q else None
            d_kv_in = unpad(d_xkv, d_xkv_l, kv_dtype, (B, Nk, d)) if need_kv else None
        return (None, d_q_in, d_kv_in, None) + tuple(g[k] for k in ("in_w", "in_b", "out_w", "out_b", "w1", "b1", "w2", "b2", "n1w", "n1b", "n2w", "n2b"))


class _SelfAttnParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names and initialisation."""

    def __init__(self, d, nhead):
        super().__init__()
        self.embed_dim, self.num_heads = d, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class QKVEncoder(nn.Module):
    def __init__(self, vdim, qdim, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", layer_norm_eps=1e-5) -> None:
        super().__init__()
        if vdim is None:
            vdim = qdim
        if vdim != qdim:
            raise NotImplementedError("QKVEncoder with vdim != qdim is not used by AsymmetricCrossFModuleBox (cross_f_box_asymm.py:53-68)")
        if activation not in ("relu", "gelu"):
            raise RuntimeError("activation should be relu/gelu, not {}".format(activation))
        if qdim % nhead or qdim % 8:
            raise ValueError(f"qdim={qdim} must be divisible by nhead={nhead} and by 8")
        if layer_norm_eps != 1e-5:
            raise NotImplementedError("layer_norm_eps other than the reference's default")
        self.nhead, self.dim_feedforward, self.dropout_p, self.activation_name = nhead, dim_feedforward, dropout, activation
        self.self_attn = _SelfAttnParams(qdim, nhead)
        self.linear1 = nn.Linear(vdim, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, vdim)
        self.norm1 = nn.LayerNorm(vdim, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(vdim, eps=layer_norm_eps)
        self._shadow_key, self._shadow = None, None
        self.precision = "bf16"          # "fp32": the fp32-accuracy mode (run.precision 32; set by the module that owns the layer)

    def _params(self):
        a = self.self_attn
        return (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, self.linear1.weight, self.linear1.bias,
                self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias)

    def _shadows(self, params, d, H, ff, planes=False):
        """bf16 shadows (and transposes) of the four weight matrices, heads padded from hd to hdp; re-packed when a parameter changed.
        ``planes``: also their lo planes (w - bf16(w)) under the keys ``<name>_lo``."""
        key = tuple((p.data_ptr(), p._version) for p in params) + (bool(planes),)
        if key == self._shadow_key:
            return self._shadow
        in_w, in_b, out_w, out_b, w1, b1, w2, b2 = params[:8]
        hd = d // H
        hdp, ffp = _up(hd, 32), _up(ff, 64)
        hq = H * hdp
        dp, ldq = _up(hq, 64), _up(3 * hq, 64)
        W = {}
        W["win"], W["winT"] = ops.pack_weight(in_w, 3 * hq, dp, dp, ldq, rg=hd, rgp=hdp)
        W["bin"] = ops.pack_bias(in_b, ldq, cg=hd, cgp=hdp)
        W["wo"], W["woT"] = ops.pack_weight(out_w, dp, dp, dp, dp, cg=hd, cgp=hdp)
        W["bo"] = ops.pack_bias(out_b, dp)
        W["w1"], W["w1T"] = ops.pack_weight(w1, ffp, dp, dp, ffp)
        W["b1"] = ops.pack_bias(b1, ffp)
        W["w2"], W["w2T"] = ops.pack_weight(w2, dp, ffp, ffp, dp)
        W["b2"] = ops.pack_bias(b2, dp)
        if planes:
            res = lambda w: w.detach().float() - w.detach().float().to(torch.bfloat16).float()      # exact in fp32
            W["win_lo"], W["winT_lo"] = ops.pack_weight(res(in_w), 3 * hq, dp, dp, ldq, rg=hd, rgp=hdp)
            W["wo_lo"], W["woT_lo"] = ops.pack_weight(res(out_w), dp, dp, dp, dp, cg=hd, cgp=hdp)
            W["w1_lo"], W["w1T_lo"] = ops.pack_weight(res(w1), ffp, dp, dp, ffp)
            W["w2_lo"], W["w2T_lo"] = ops.pack_weight(res(w2), dp, ffp, ffp, dp)
        self._shadow_key, self._shadow = key, W
        return W

    def forward(self, q, k, v, src_mask=None, src_key_padding_mask=None, get_attentions=False):
        if k is not v:
            raise NotImplementedError("QKVEncoder: keys and values are the same token set in every call of the reference "
                                      "(cross_f_box_asymm.py:92-112 passes v_k, v_k)")
        if src_mask is not None:
            raise NotImplementedError("QKVEncoder with an attention mask (src_mask) is not used by the reference")
        if get_attentions:
            raise NotImplementedError("get_attentions=True: the fused kernels never materialise the attention matrix")
        if q.shape[-1] != self.self_attn.embed_dim or k.shape[-1] != q.shape[-1] or k.shape[0] != q.shape[0]:
            raise RuntimeError(f"token shapes {tuple(q.shape)} / {tuple(k.shape)} do not match qdim={self.self_attn.embed_dim}")
        qc = q.contiguous()
        kc = qc if k is q else k.contiguous()
        out = _LayerFn.apply(self, qc, kc, src_key_padding_mask, *self._params())
        return out, None, None
