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
``tf_attn_bwd``, ``tf_layernorm_bwd``).  bf16 compute, fp32 parameters / statistics / gradients.
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
        W = mod._shadows(params, d, H, ff)
        in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b = params
        p = float(mod.dropout_p) if mod.training else 0.0
        seed = ops.next_seed() if p > 0 else 0
        site = lambda which: 16 + which                      # the runtime's site numbering of layer 0 (tf_api.hip)
        act = 1 if mod.activation_name == "relu" else 0

        xq = ops.to_bf16_padded(q_in.reshape(Mq, d), dp)
        xkv = xq if (kv_in is q_in) else ops.to_bf16_padded(kv_in.reshape(Mk, d), dp)
        Q = torch.zeros(Mq, dp, dtype=bf, device=dev)
        KV = torch.zeros(Mk, ldkv, dtype=bf, device=dev)      # thirds: (unused Q) | K | V, as tf_attn_* expects
        ops.gemm(xq, W["win"], Q, hq, dp, L.TF_EPI_BIAS, bias=W["bin"])
        ops.gemm(xkv, W["win"][hq:], KV[:, hq:], 2 * hq, dp, L.TF_EPI_BIAS, bias=W["bin"][hq:])
        O = torch.empty(Mq, dp, dtype=bf, device=dev)
        lse = torch.empty(B * H * Nq, dtype=torch.float32, device=dev)
        km = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
        drop_a = ops.drop_params(p, seed, site(1))
        bits = None
        if drop_a[0]:
            lib = L.load()
            bits = torch.empty(B * H * Nq * ((Nk + 63) // 64), dtype=torch.int64, device=dev)
            L.check(lib.tf_attn_dropmask_rows(L.ptr(bits), B * H * Nq, Nk, drop_a[1], drop_a[0], ops._stream()), "tf_attn_dropmask_rows")
        att = L.TfAttnArgs(qkv=L.ptr(KV), ld_qkv=ldkv, out=L.ptr(O), ld_out=dp, lse=L.ptr(lse), key_mask=L.ptr(km), B=B, S=Nk, H=H, HDP=hdp,
                           scale=1.0 / math.sqrt(hd), drop_thr=drop_a[0], drop_key=drop_a[1], drop_scale=drop_a[2], drop_bits=L.ptr(bits),
                           q=L.ptr(Q), ld_q=dp, Sq=Nq)
        L.call("tf_attn_fwd", att, ops._stream())
        z1 = torch.empty(Mq, dp, dtype=bf, device=dev)
        drop_1 = ops.drop_params(p, seed, site(2))
        ops.gemm(O, W["wo"], z1, dp, dp, L.TF_EPI_BIAS_DROP_RES, bias=W["bo"], R=xq, drop=drop_1)
        x1 = torch.empty(Mq, dp, dtype=bf, device=dev)
        st1 = torch.empty(2, Mq, dtype=torch.float32, device=dev)
        ops.layernorm_fwd(z1, x1, n1w, n1b, st1[0], st1[1], Mq, d)
        G = torch.empty(Mq, ffp, dtype=bf, device=dev)
        Hh = torch.empty(Mq, ffp, dtype=bf, device=dev)
        drop_f = ops.drop_params(p, seed, site(3))
        ops.gemm(x1, W["w1"], G, ffp, dp, L.TF_EPI_BIAS_GELU_DROP_G, bias=W["b1"], C2=Hh, drop=drop_f, act=act)
        z2 = torch.empty(Mq, dp, dtype=bf, device=dev)
        drop_2 = ops.drop_params(p, seed, site(4))
        ops.gemm(Hh, W["w2"], z2, dp, ffp, L.TF_EPI_BIAS_DROP_RES, bias=W["b2"], R=x1, drop=drop_2)
        out = torch.empty(B, Nq, d, dtype=q_in.dtype, device=dev)
        st2 = torch.empty(2, Mq, dtype=torch.float32, device=dev)
        ops.layernorm_fwd(z2, out.view(Mq, d), n2w, n2b, st2[0], st2[1], Mq, d)
        ctx.saved = (xq, xkv, Q, KV, O, lse, km, bits, z1, st1, x1, G, Hh, z2, st2, W, att)
        ctx.meta = (B, Nq, Nk, d, H, ff, hd, hdp, hq, dp, ffp, ldkv, (drop_a, drop_1, drop_f, drop_2), act, q_in.dtype, kv_in.dtype, kv_in is q_in)
        ctx.mod = mod
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, g_out):
        xq, xkv, Q, KV, O, lse, km, bits, z1, st1, x1, G, Hh, z2, st2, W, att = ctx.saved
        B, Nq, Nk, d, H, ff, hd, hdp, hq, dp, ffp, ldkv, drops, act, q_dtype, kv_dtype, same = ctx.meta
        drop_a, drop_1, drop_f, drop_2 = drops
        in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b = ctx.params
        Mq, Mk = B * Nq, B * Nk
        dev, bf = g_out.device, torch.bfloat16
        g_out = g_out.contiguous().view(Mq, d)
        zf = lambda t: torch.zeros_like(t, dtype=torch.float32, memory_format=torch.contiguous_format)
        g = {k: zf(t) for k, t in zip(("in_w", "in_b", "out_w", "out_b", "w1", "b1", "w2", "b2", "n1w", "n1b", "n2w", "n2b"), ctx.params)}
        # LN2 backward -> dz (residual path) and dy2 (dropout2-masked, feeds linear2's backward)
        dz = torch.empty(Mq, dp, dtype=bf, device=dev)
        dy = torch.empty(Mq, dp, dtype=bf, device=dev) if drop_2[0] else None
        ops.layernorm_bwd(z2, n2w, st2[0], st2[1], g_out, dz, g["n2w"], g["n2b"], Mq, d, dx_drop=dy, drop=drop_2)
        dy2 = dy if drop_2[0] else dz
        du = torch.empty(Mq, ffp, dtype=bf, device=dev)
        ops.gemm(dy2, W["w2T"], du, ffp, dp, L.TF_EPI_MUL, R=G)                              # dU = dH . G
        ops.wgrad(dy2, dp, Hh, ffp, g["w2"], g["b2"], n_src=d, k_src=ff)
        ops.wgrad(du, ffp, x1, dp, g["w1"], g["b1"], n_src=ff, k_src=d)
        dxb = torch.empty(Mq, dp, dtype=bf, device=dev)
        ops.gemm(du, W["w1T"], dxb, dp, ffp, L.TF_EPI_ADD, R=dz)
        # LN1 backward
        dzb = torch.empty(Mq, dp, dtype=bf, device=dev)
        dyb = torch.empty(Mq, dp, dtype=bf, device=dev) if drop_1[0] else None
        ops.layernorm_bwd(z1, n1w, st1[0], st1[1], dxb, dzb, g["n1w"], g["n1b"], Mq, d, dx_drop=dyb, drop=drop_1)
        dy1 = dyb if drop_1[0] else dzb
        ops.wgrad(dy1, dp, O, dp, g["out_w"], g["out_b"], n_src=d, cg=hd, cgp=hdp, k_src=d)
        d_o = torch.empty(Mq, dp, dtype=bf, device=dev)
        ops.gemm(dy1, W["woT"], d_o, dp, dp, L.TF_EPI_NONE)
        dQ = torch.zeros(Mq, dp, dtype=bf, device=dev)
        dKV = torch.zeros(Mk, ldkv, dtype=bf, device=dev)
        delta = torch.empty(B * H * Nq, dtype=torch.float32, device=dev)
        att.dout, att.ld_dout, att.dqkv, att.ld_dqkv, att.delta, att.dq, att.ld_dq = L.ptr(d_o), dp, L.ptr(dKV), ldkv, L.ptr(delta), L.ptr(dQ), dp
        L.call("tf_attn_bwd", att, ops._stream())
        # in_proj: rows [0, d) of the packed weight from the queries, rows [d, 3d) from the keys / values
        ops.wgrad(dQ, hq, xq, dp, g["in_w"][:d], g["in_b"][:d], rg=hd, rgp=hdp, n_src=d, k_src=d)
        ops.wgrad(dKV[:, hq:], 2 * hq, xkv, dp, g["in_w"][d:], g["in_b"][d:], rg=hd, rgp=hdp, n_src=2 * d, k_src=d)
        d_xq = torch.empty(Mq, dp, dtype=bf, device=dev)
        ops.gemm(dQ, W["winT"][:, :dp], d_xq, dp, dp, L.TF_EPI_ADD, R=dzb)                   # columns [hq, dp) of dQ are zero
        d_xkv = torch.empty(Mk, dp, dtype=bf, device=dev)
        ops.gemm(dKV[:, hq:], W["winT"][:, hq:], d_xkv, dp, 2 * hq, L.TF_EPI_NONE)
        need_q, need_kv = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if same:                                       # self-attention call: both gradients belong to the one input
            d_q_in = (ops.from_padded(d_xq, d, torch.float32) + ops.from_padded(d_xkv, d, torch.float32)).to(q_dtype).view(B, Nq, d) if need_q else None
            d_kv_in = None
        else:
            d_q_in = ops.from_padded(d_xq, d, q_dtype).view(B, Nq, d) if need_q else None
            d_kv_in = ops.from_padded(d_xkv, d, kv_dtype).view(B, Nk, d) if need_kv else None
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

    def _params(self):
        a = self.self_attn
        return (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, self.linear1.weight, self.linear1.bias,
                self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias)

    def _shadows(self, params, d, H, ff):
        """bf16 shadows (and transposes) of the four weight matrices, heads padded from hd to hdp; re-packed when a parameter changed."""
        key = tuple((p.data_ptr(), p._version) for p in params)
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
