"""FusionTrainStep over modules whose bf16 weight shadows are cached on ``(data_ptr, _version)`` of their Parameters: PatchToToken
(K1), a QKVEncoder layer and the RoI heads.  FlatParams re-homes the parameters into a flat buffer and the fused optimiser writes that
buffer through a raw pointer, so a re-homed parameter's own version counter does not move unless the train step bumps it -- and then
the forward keeps multiplying by the step-0 weights while the (correctly computed) weight gradients make the loss look plausible.
Every step's forward is therefore compared with an fp32 recomputation (the oracle) from the parameters READ BACK at that step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


class _Composite(torch.nn.Module):
    def __init__(self, C, p, d, H, nouns, verbs):
        super().__init__()
        from transfusion_amd.modeling.cross_fusion.cross_qkv_layers import QKVEncoder
        from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_wrapper import PatchToToken
        from transfusion_amd.modeling.obj_detection.nao_heads import NaoRoIHeads
        self.k1 = PatchToToken(C, d, p, p)
        self.layer = QKVEncoder(d, d, H, dim_feedforward=2 * d, dropout=0.0, activation="gelu")
        self.heads = NaoRoIHeads(d, nouns, verbs)
        self.H = H

    def forward(self, feat):
        tok = self.k1(feat)
        out, _, _ = self.layer(tok, tok, tok)
        rois = out.float().mean(dim=1)                       # [B, d] stand-in for the box features
        return tok, out, self.heads(rois)


def _oracle_forward(sd, feat, H):
    from oracle import fusion_oracle as O
    tok = O.patch_embed(feat, sd["k1.weight"])
    lsd = {k[len("layer."):]: v for k, v in sd.items() if k.startswith("layer.")}
    out = O.qkv_encoder_layer(lsd, "", tok, tok, H, activation="gelu")
    hsd = {k[len("heads."):]: v for k, v in sd.items() if k.startswith("heads.")}
    return tok, out, O.nao_heads_forward(hsd, out.mean(dim=1))


def test_every_step_uses_the_updated_weights():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from transfusion_amd.runner.trainer import FusionTrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    C, p, d, H, nouns, verbs = 16, 2, 64, 2, 11, 7
    model = _Composite(C, p, d, H, nouns, verbs).to(dev).train()
    from transfusion_amd.optim import FusedRAdam
    # RAdam does not move the parameters in its first five (un-rectified) steps unless degenerated_to_sgd (radam_optim.py:64-84); with it
    # the first steps are lr * g -- large here, so that stale weights are unmistakable
    opt_cls = lambda params, lr, weight_decay: FusedRAdam(params, lr=lr, weight_decay=weight_decay, degenerated_to_sgd=True)
    tr = FusionTrainStep(model, lr=10.0, weight_decay=0.0, grad_clip=None, optimizer_cls=opt_cls)
    feats = [torch.randn(4, C, 12, 12, generator=torch.Generator().manual_seed(50 + i)) for i in range(4)]
    first = {n: q.detach().clone() for n, q in model.named_parameters()}

    def loss_fn(m, feat):
        tok, out, ho = m(feat)
        return ho["class_logits"].float().pow(2).mean() + ho["verb_logits"].float().pow(2).mean() + ho["box_regression"].float().pow(2).mean() \
            + out.float().pow(2).mean()

    for step in range(4):
        feat = feats[step].to(dev)
        sd = {n: q.detach().float().cpu().clone() for n, q in model.named_parameters()}        # the parameters as they are NOW
        with torch.no_grad():
            tok, out, ho = model(feat)
        r_tok, r_out, r_ho = _oracle_forward(sd, feats[step], H)
        assert rel(tok, r_tok) < 1e-2, step
        assert rel(out, r_out) < 1e-2, step
        for k in ("class_logits", "verb_logits", "box_regression"):
            assert rel(ho[k], r_ho[k]) < 2e-2, (step, k)
        if step > 0:
            # and the step DID move every weight far enough for a stale shadow to fail the checks above
            moved = {n: rel(sd[n], first[n].cpu()) for n in ("k1.weight", "layer.linear1.weight", "heads.noun_classifier.weight")}
            assert min(moved.values()) > 3e-2, moved
            o_tok, _, _ = _oracle_forward({n: v.cpu() for n, v in first.items()}, feats[step], H)
            assert rel(tok, o_tok) > 3e-2                   # ... i.e. the step-0 weights give a visibly different answer
        tr.step([feat], loss_fn)
    torch.cuda.synchronize()
