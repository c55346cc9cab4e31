"""Diagnostic (GPU): localise an error of the fp32-accuracy mode stage by stage -- every peeked activation of layer 0 against a torch
fp32 computation from the PREVIOUS peeked stage -- and sweep the split attention kernels over head dims / lengths."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import ENCODER_CASES, make_encoder_inputs, make_encoder_params  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def encoder_stages(name, precision="fp32"):
    from test_gpu_fp32_mode import build
    from oracle import fusion_oracle as O
    dev = torch.device("cuda:0")
    cfg = ENCODER_CASES[name]
    enc, params = build(cfg, dev, precision)
    enc.train()
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    xd, ld, md = torch.from_numpy(x).to(dev), torch.from_numpy(lang).to(dev), torch.from_numpy(mask).to(dev)
    desc, keep = enc._make_desc(xd, ld, md)
    B, Nv, Nl, d, H = cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["h"]
    vis_out = torch.empty(B, Nv, d, device=dev)
    lang_out = torch.empty(B, Nl, d, device=dev)
    from transfusion_amd import _lib as L, ops
    desc.vis_out, desc.vis_out_is_f32, desc.lang_out, desc.lang_out_is_f32, desc.repack = vis_out.data_ptr(), 1, lang_out.data_ptr(), 1, 1
    L.call("tf_encoder_fwd", desc, ops._stream())
    torch.cuda.synchronize()
    pk = lambda n: enc.peek((desc, keep), n).float().cpu()
    S, hd = Nv + Nl, d // H
    P = {k: torch.from_numpy(v) for k, v in params.items()}
    pre = "t_encoder.layers.0."
    x0 = pk("x0")[:, :d]
    pe = O.sin1d_table(8192, d)
    x0_ref = torch.cat([torch.from_numpy(x) + pe[:, :Nv] + P["image_kind_embedding"], torch.from_numpy(lang) + P["lang_kind_embedding"]], 1).reshape(B * S, d)
    print(f"{name} [{precision}] x0 {rel(x0, x0_ref):.2e}")
    hdp = (hd + 31) // 32 * 32
    qkv = pk("qkv0")
    qkv_u = qkv[:, :3 * H * hdp].reshape(B * S, 3 * H, hdp)[:, :, :hd].reshape(B * S, 3 * d)
    qkv_ref = x0 @ P[pre + "self_attn.in_proj_weight"].t() + P[pre + "self_attn.in_proj_bias"]
    print(f"  qkv {rel(qkv_u, qkv_ref):.2e}")
    q, k, v = [t.reshape(B, S, H, hd).permute(0, 2, 1, 3) for t in qkv_u.split(d, dim=-1)]
    sc = (q / math.sqrt(hd)) @ k.transpose(-1, -2)
    kpm = torch.cat([torch.zeros(B, Nv, dtype=torch.bool), torch.from_numpy(mask)], 1)
    sc = sc.masked_fill(kpm.view(B, 1, 1, S), float("-inf"))
    o_ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, d)
    o = pk("o0")[:, :H * hdp].reshape(B * S, H, hdp)[:, :, :hd].reshape(B * S, d)
    print(f"  attn out {rel(o, o_ref):.2e}   per head: " + " ".join(f"{rel(o.view(B*S,H,hd)[:,i], o_ref.view(B*S,H,hd)[:,i]):.1e}" for i in range(H)))
    for qb in range((S + 127) // 128):
        rows = torch.arange(qb * 128, min(S, qb * 128 + 128))
        print(f"    q-block {qb}: {rel(o.view(B, S, d)[:, rows], o_ref.view(B, S, d)[:, rows]):.2e}")
    z1 = pk("z1_0")[:, :d]
    z1_ref = x0 + o @ P[pre + "self_attn.out_proj.weight"].t() + P[pre + "self_attn.out_proj.bias"]
    print(f"  z1 {rel(z1, z1_ref):.2e}")
    x1 = pk("x1_0")[:, :d]
    print(f"  x1 {rel(x1, O.layer_norm(z1, P[pre + 'norm1.weight'], P[pre + 'norm1.bias'])):.2e}")
    hh = pk("h0")[:, :2 * d]
    print(f"  h {rel(hh, O.gelu(x1 @ P[pre + 'linear1.weight'].t() + P[pre + 'linear1.bias'])):.2e}")
    z2 = pk("z2_0")[:, :d]
    print(f"  z2 {rel(z2, x1 + hh @ P[pre + 'linear2.weight'].t() + P[pre + 'linear2.bias']):.2e}")
    print(f"  x(1) {rel(pk('x1')[:, :d], O.layer_norm(z2, P[pre + 'norm2.weight'], P[pre + 'norm2.bias'])):.2e}")
    sd = {k: v.clone() for k, v in P.items()}
    sd["pos_embedding_layer.pos_embedding"] = pe
    v_ref, l_ref = O.encoder_forward(sd, torch.from_numpy(x), torch.from_numpy(lang), torch.from_numpy(mask), H, cfg["L"])
    print(f"  final vis {rel(vis_out, v_ref):.2e}  lang(valid) {rel(lang_out.cpu()[~torch.from_numpy(mask)], l_ref[~torch.from_numpy(mask)]):.2e}")
    xg, lg = xd.clone().requires_grad_(True), ld.clone().requires_grad_(True)
    v2, l2, _, _ = enc(xg, lg, md)
    print(f"  through the module: vis {rel(v2, v_ref):.2e}")
    ((v2 * torch.from_numpy(gv).to(dev)).sum() + (l2 * torch.from_numpy(gl).to(dev)).sum()).backward()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "pos_embedding_layer.pos_embedding"}
    sdg["pos_embedding_layer.pos_embedding"] = pe
    xr, lr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
    vr, lr_o = O.encoder_forward(sdg, xr, lr, torch.from_numpy(mask), H, cfg["L"])
    ((vr * torch.from_numpy(gv)).sum() + (lr_o * torch.from_numpy(gl)).sum()).backward()
    print(f"  grads: x {rel(xg.grad, xr.grad):.2e} lang {rel(lg.grad, lr.grad):.2e} " + " ".join(
        f"{k.split('.')[-2][:4]}.{k.split('.')[-1][:1]} {rel(p.grad, sdg[k].grad):.1e}" for k, p in enc.named_parameters() if k in sdg and sdg[k].grad is not None and p.grad is not None))


def attention_sweep():
    from transfusion_amd import _lib as L, ops
    from test_gpu_fp32_mode import planes, joined
    dev = torch.device("cuda:0")
    for hd, S, p, H in [(192, 130, 0.0, 2), (192, 333, 0.0, 2), (192, 333, 0.15, 2), (192, 260, 0.0, 4), (224, 333, 0.15, 2), (160, 333, 0.15, 2), (128, 333, 0.0, 2)]:
        B = 2
        g = torch.Generator().manual_seed(hd + S)
        qkv = torch.randn(B * S, 3 * H * hd, generator=g) * 0.7
        dout = torch.randn(B * S, H * hd, generator=g)
        qh, ql = planes(qkv, dev)
        dh, dl = planes(dout, dev)
        km = torch.zeros(B, S, dtype=torch.uint8)
        km[0, S - 37:] = 1
        oh = torch.empty(B * S, H * hd, dtype=torch.bfloat16, device=dev)
        ol = torch.empty_like(oh)
        lse, delta = torch.empty(B * H * S, device=dev), torch.empty(B * H * S, device=dev)
        gh = torch.zeros(B * S, 3 * H * hd, dtype=torch.bfloat16, device=dev)
        gl = torch.zeros_like(gh)
        drop = ops.drop_params(p, 11, 5)
        bits = ops.attn_dropmask(B, H, S, p, 11, 5, dev) if p > 0 else None
        kmd = km.to(dev)
        a = L.TfAttnArgs(qkv=L.ptr(qh), qkv_lo=L.ptr(ql), ld_qkv=3 * H * hd, out=L.ptr(oh), out_lo=L.ptr(ol), ld_out=H * hd, lse=L.ptr(lse),
                         key_mask=L.ptr(kmd), B=B, S=S, H=H, HDP=hd, scale=1 / math.sqrt(hd), drop_thr=drop[0], drop_key=drop[1],
                         drop_scale=drop[2], drop_bits=L.ptr(bits), dout=L.ptr(dh), dout_lo=L.ptr(dl), ld_dout=H * hd, dqkv=L.ptr(gh),
                         dqkv_lo=L.ptr(gl), ld_dqkv=3 * H * hd, delta=L.ptr(delta))
        L.call("tf_attn_fwd", a, ops._stream())
        L.call("tf_attn_bwd", a, ops._stream())
        torch.cuda.synchronize()
        keep = torch.ones(B, H, S, S, dtype=torch.float64)
        if p > 0:
            keep = ops.dropout_mask(B * H * S * S, p, 11, 5, dev).cpu().view(B, H, S, S).double() * drop[2]
        x = joined(qh, ql).cpu().view(B, S, 3, H, hd).requires_grad_(True)
        q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
        sc = ((q / math.sqrt(hd)) @ k.transpose(-1, -2)).masked_fill(km.bool().view(B, 1, 1, S), float("-inf"))
        o = ((torch.softmax(sc, dim=-1) * keep) @ v).permute(0, 2, 1, 3).reshape(B * S, H * hd)
        o.backward(joined(dh, dl).cpu())
        gref, got = x.grad.reshape(B * S, 3 * H * hd), joined(gh, gl).cpu()
        e = [rel(got[:, i * H * hd:(i + 1) * H * hd], gref[:, i * H * hd:(i + 1) * H * hd]) for i in range(3)]
        print(f"attn hd={hd} S={S} p={p} H={H}: O {rel(joined(oh, ol), o.detach()):.1e}  dq {e[0]:.1e} dk {e[1]:.1e} dv {e[2]:.1e}")


def bf16_error_budget(name):
    """bf16 compute against (a) the fp32 oracle and (b) the oracle evaluated on bf16-ROUNDED parameters and inputs: (b) removes the
    operand quantisation every bf16 implementation shares, what remains is this implementation's own rounding of intermediates."""
    from test_gpu_fp32_mode import build
    from oracle import fusion_oracle as O
    dev = torch.device("cuda:0")
    cfg = ENCODER_CASES[name]
    enc, params = build(cfg, dev, "bf16")
    enc.train()
    x, lang, mask, gv, gl = make_encoder_inputs(cfg["seed"], cfg["B"], cfg["Nv"], cfg["Nl"], cfg["d"], cfg["mask_lens"])
    xd, ld = torch.from_numpy(x).to(dev).requires_grad_(True), torch.from_numpy(lang).to(dev).requires_grad_(True)
    md = None if mask is None else torch.from_numpy(mask).to(dev)
    vis, lo, _, _ = enc(xd, ld, md)
    ((vis * torch.from_numpy(gv).to(dev)).sum() + (lo * torch.from_numpy(gl).to(dev)).sum()).backward()
    rb = lambda t: t.to(torch.bfloat16).float()
    for label, rnd in (("fp32 oracle", lambda t: t), ("bf16-rounded oracle", rb)):
        sd = {k: rnd(torch.from_numpy(v)).clone().requires_grad_(True) for k, v in params.items()}
        for k in list(sd):
            if "norm" in k or "bias" in k or "kind" in k:            # fp32 inside the runtime as well
                sd[k] = torch.from_numpy(params[k]).clone().requires_grad_(True)
        sd["pos_embedding_layer.pos_embedding"] = O.sin1d_table(8192, cfg["d"])
        xr, lr = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(lang).requires_grad_(True)
        mk = None if mask is None else torch.from_numpy(mask)
        v_ref, l_ref = O.encoder_forward(sd, xr, lr, mk, cfg["h"], cfg["L"])
        ((v_ref * torch.from_numpy(gv)).sum() + (l_ref * torch.from_numpy(gl)).sum()).backward()
        worst = max(rel(p.grad, sd[k].grad) for k, p in enc.named_parameters() if k in sd and sd[k].grad is not None and p.grad is not None)
        print(f"{name} bf16 vs {label}: vis rel {rel(vis, v_ref):.2e} max-abs {(vis.detach().cpu() - v_ref.detach()).abs().max():.2e}  dX {rel(xd.grad, xr.grad):.2e}  worst param grad {worst:.2e}")


if __name__ == "__main__":
    if sys.argv[1:2] == ["budget"]:
        for n in ("enc_small", "enc_hd18", "enc_d768", "enc_d896"):
            bf16_error_budget(n)
        sys.exit(0)
    attention_sweep()
    for n in sys.argv[1:] or ["enc_d768", "enc_d896"]:
        encoder_stages(n)
