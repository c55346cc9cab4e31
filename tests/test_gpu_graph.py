"""A training step captured in a HIP graph (tests/graph_step.py: test infrastructure since round 4) and the pieces that make its replays NEW steps:
the library's step clock folded into every dropout key (tf_clock_ptr), RAdam's schedule formed on the device from that clock.
The eager path -- the reference semantics, the one every other test exercises -- is the yardstick: eager steps on the same clock
and replays of the captured step must leave the same parameters."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def test_clock_salts_every_key_and_zero_is_identity():
    dev = _dev()
    from transfusion_amd import ops
    n, p = 1 << 16, 0.25
    ops.clock_set(0)
    m0 = ops.dropout_mask(n, p, 7, 3, dev).clone()
    b0 = ops.attn_dropmask(2, 2, 96, p, 7, 3, dev).clone()
    ops.clock_advance()
    m1 = ops.dropout_mask(n, p, 7, 3, dev).clone()
    b1 = ops.attn_dropmask(2, 2, 96, p, 7, 3, dev).clone()
    ops.clock_advance()
    m2 = ops.dropout_mask(n, p, 7, 3, dev).clone()
    ops.clock_set(0)
    m0b = ops.dropout_mask(n, p, 7, 3, dev)
    assert torch.equal(m0, m0b)                                   # clock 0: the keys the caller passed (mix(0) = 0)
    for a, b in ((m0, m1), (m1, m2), (m0, m2)):
        agree = (a == b).float().mean().item()
        assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.02, agree     # independent masks of the same rate
        assert abs(b.float().mean().item() - (1 - p)) < 0.01
    assert not torch.equal(b0, b1)


@pytest.mark.parametrize("degenerated", [False, True])
def test_radam_schedule_on_the_device_equals_the_host_schedule(degenerated):
    dev = _dev()
    from transfusion_amd import ops
    from transfusion_amd.optim import FusedRAdam
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(4099, generator=g)
    grads = [torch.randn(4099, generator=g) * 0.1 for _ in range(9)]
    res = []
    for on_clock in (False, True):
        ops.clock_set(100)                                        # any origin: the optimiser keeps step0 = step - clock
        p = torch.nn.Parameter(p0.clone().to(dev))
        opt = FusedRAdam([p], lr=1e-2, weight_decay=1e-3, degenerated_to_sgd=degenerated)
        for gr in grads:                                          # steps 1 .. 9 cross the rectification threshold (N_sma >= 5 at step 6)
            p.grad = gr.to(dev)
            if on_clock:
                ops.clock_advance()
                opt.step(on_clock=True)
            else:
                opt.step()
        res.append(p.detach().cpu().clone())
    ops.clock_set(0)
    assert (res[0] - p0).abs().max() > 1e-3                       # the parameters moved
    assert torch.allclose(res[0], res[1], rtol=2e-6, atol=1e-7), (res[0] - res[1]).abs().max()


def _encoder(dev, d=64, H=2, layers=2):
    from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_layers import CrossTransformerModuleBox
    from transfusion_amd.modeling.cross_fusion.utils import PositionalEmbeddingLayer
    torch.manual_seed(42)
    pe = PositionalEmbeddingLayer("sin1d", 512, d)
    enc = CrossTransformerModuleBox(no_patches=512, pos_embedding_layer=pe, lang_pos_embedding=None, num_layers=layers, patch_dropout=0.1,
                                    num_heads=H, fforward_multiplier=2, token_dropout=0.15, back_to_img_fn="regroup", activ_f="gelu",
                                    final_norm="ln", input_f_size=d)
    return enc.to(dev).train()


def _batch(dev, k, B=3, nv=36, nl=40, d=64):
    g = torch.Generator().manual_seed(900 + k)
    x = torch.randn(B, nv, d, generator=g)
    lang = torch.randn(B, nl, d, generator=g)
    lens = torch.tensor([40, 23, 31])
    pad = torch.arange(nl).view(1, -1) >= lens.view(-1, 1)
    return [x.to(dev), lang.to(dev), pad.to(dev)]


def _loss(m, batch):
    x, lang, pad = batch
    vis, lo, _, _ = m(x, lang, pad)
    return vis.float().pow(2).mean() + (lo.float() * (~pad).unsqueeze(-1)).pow(2).mean()


def test_replays_equal_eager_steps_on_the_same_clock(monkeypatch):
    """Same initial weights, same batches, same base seed, same clock values: WARM eager steps + R replays of the captured step
    against WARM + R eager steps.  Dropout is on at every site, RAdam is in its SGD-degenerated mode so that every step moves the
    weights (and a replay that re-used the captured step's masks, step number or weights would show)."""
    dev = _dev()
    from transfusion_amd import ops
    from transfusion_amd.optim import FusedRAdam
    from transfusion_amd.runner.trainer import FusionTrainStep
    from graph_step import GraphedTrainStep
    monkeypatch.setattr(ops, "next_seed", lambda: 0x1234ABCD)    # eager calls draw a new seed per forward; the graph bakes one
    # five steps in all: RAdam's rectified branch starts at step 6, and there sqrt(v) normalises the update of entries whose true
    # gradient is zero (the K third of in_proj_bias) by their rounding noise -- no two runs agree on those, eager or not; the branch
    # itself is covered by test_radam_schedule_on_the_device_equals_the_host_schedule
    WARM, R = 2, 3
    opt_cls = lambda params, lr, weight_decay: FusedRAdam(params, lr=lr, weight_decay=weight_decay, degenerated_to_sgd=True)
    batches = [_batch(dev, k) for k in range(WARM + R)]
    finals, losses = [], []
    for graphed in (False, True):
        ops.clock_set(0)
        enc = _encoder(dev)
        enc.pack_tokens = False
        tr = FusionTrainStep(enc, lr=0.05, weight_decay=1e-3, grad_clip=1.0, optimizer_cls=opt_cls)
        ls = []
        if not graphed:
            for b in batches:
                ops.clock_advance()
                ls.append(float(tr.step([b], _loss, on_clock=True)))
        else:
            static = [t.clone() for t in batches[0]]
            # the warm-up inside the constructor runs WARM eager steps on `static`: feed it the first WARM batches through a hook
            it = iter(batches[:WARM])

            def loss_hook(m, batch):
                nxt = next(it, None)
                if nxt is not None:
                    for d_, s_ in zip(static, nxt):
                        d_.copy_(s_)
                return _loss(m, static)
            gs = GraphedTrainStep(tr, static, loss_hook, warmup=WARM)
            assert ops.clock_value() == WARM
            for b in batches[WARM:]:
                gs.load(b)
                ls.append(float(gs.replay()))
            assert ops.clock_value() == WARM + R and gs.replays == R
            gs.finish()
        torch.cuda.synchronize()
        finals.append({n: q.detach().float().cpu().clone() for n, q in enc.named_parameters()})
        losses.append(ls)
    ops.clock_set(0)
    assert max(abs(a - b) for a, b in zip(losses[0][WARM:], losses[1])) < 2e-3 * max(losses[0]), (losses[0][WARM:], losses[1])
    diffs = {n: (finals[0][n] - finals[1][n]).abs().max().item() for n in finals[0]}
    worst = max(diffs.values())
    assert worst < 2e-5, sorted(diffs.items(), key=lambda kv: -kv[1])[:4]      # fp32 atomics in another order, nothing more
    init = {n: q.detach().float().cpu() for n, q in _encoder(dev).named_parameters()}
    moved = max(((finals[1][n] - init[n]).norm() / (init[n].norm() + 1e-12)).item() for n in init)
    assert moved > 5e-2, moved                                    # and the replays did train


def test_replays_draw_new_masks(monkeypatch):
    dev = _dev()
    from transfusion_amd import ops
    from transfusion_amd.runner.trainer import FusionTrainStep
    from graph_step import GraphedTrainStep
    ops.clock_set(0)
    enc = _encoder(dev)
    tr = FusionTrainStep(enc, lr=0.0, weight_decay=0.0, grad_clip=None)       # lr 0: the weights stay, only the masks change
    static = _batch(dev, 0)
    gs = GraphedTrainStep(tr, static, _loss, warmup=1)
    ls = [float(gs.replay()) for _ in range(4)]
    gs.finish()
    ops.clock_set(0)
    assert len({round(v, 7) for v in ls}) == 4, ls                # four replays of one batch at fixed weights: four different losses
    assert max(ls) - min(ls) < 0.2 * max(ls)


def test_optimizer_can_zero_the_gradients_it_reads():
    """FusionTrainStep(zero_grads_in_optimizer=True): the fused optimiser zeroes each gradient as it reads it and the step skips its
    own zero fill from the second step on -- same parameters as the plain form, gradient buffer all zeros between steps."""
    dev = _dev()
    from transfusion_amd.optim import FusedRAdam
    from transfusion_amd.runner.trainer import FusionTrainStep
    opt_cls = lambda params, lr, weight_decay: FusedRAdam(params, lr=lr, weight_decay=weight_decay, degenerated_to_sgd=True)
    finals = []
    for fused in (False, True):
        enc = _encoder(dev)
        enc.token_dropout = enc.patch_dropout = 0.0
        tr = FusionTrainStep(enc, lr=0.05, weight_decay=1e-3, grad_clip=1.0, optimizer_cls=opt_cls, zero_grads_in_optimizer=fused)
        for k in range(4):
            tr.step([_batch(dev, k)], _loss)
            if fused:
                assert float(tr.flat.grad.abs().max()) == 0.0
        torch.cuda.synchronize()
        finals.append(tr.flat.flat.detach().cpu().clone())
    assert (finals[0] - finals[1]).abs().max().item() < 2e-5


def test_lr_scale_ranges_equal_parameter_groups():
    """FusionTrainStep(lr_scale=...) -- ranges of the flat buffer with their own learning rate -- against FusedRAdam over the separate
    parameters with the same groups (the reference's lr / div_rate, lr / ttc_rate layout, ego_nao_trainer.py:440-497)."""
    dev = _dev()
    from transfusion_amd.optim import FusedRAdam
    from transfusion_amd.runner.trainer import FusionTrainStep
    torch.manual_seed(3)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(16, 24), torch.nn.Tanh(), torch.nn.Linear(24, 8), torch.nn.Tanh(), torch.nn.Linear(8, 4)).to(dev)
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    scale = lambda name: 0.25 if name.startswith("2.") else 1.0
    sgd = lambda ps, lr, weight_decay: FusedRAdam(ps, lr=lr, weight_decay=weight_decay, degenerated_to_sgd=True)
    tr = FusionTrainStep(a, lr=0.05, weight_decay=1e-3, grad_clip=None, optimizer_cls=sgd, lr_scale=scale)
    assert len(tr.opt.param_groups) == 3
    ref = FusedRAdam([{"params": [p], "lr": 0.05 * scale(n)} for n, p in b.named_parameters()], lr=0.05, weight_decay=1e-3, degenerated_to_sgd=True)
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        x = torch.randn(32, 16, generator=g).to(dev)
        tr.step([x], lambda m, xb: m(xb).pow(2).mean())
        ref.zero_grad()
        b(x).pow(2).mean().backward()
        ref.step()
    torch.cuda.synchronize()
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert (p - q).abs().max().item() < 1e-6, n


def test_replays_follow_the_lr_scheduler():
    """A captured step reads its learning rate from a device scalar (TfRadamArgs.lr_dev) that replay() refreshes from
    ``param_groups``: replays under a schedule that moves the rate equal eager steps under the same schedule (dropout off, one batch)."""
    dev = _dev()
    from transfusion_amd import ops
    from transfusion_amd.optim import FusedRAdam
    from transfusion_amd.runner.trainer import FusionTrainStep
    from graph_step import GraphedTrainStep
    opt_cls = lambda params, lr, weight_decay: FusedRAdam(params, lr=lr, weight_decay=weight_decay, degenerated_to_sgd=True)
    rates = [0.05, 0.05, 0.01, 0.002, 0.03]
    finals = []
    for graphed in (False, True):
        ops.clock_set(0)
        enc = _encoder(dev)
        enc.token_dropout = enc.patch_dropout = 0.0
        for m in enc.modules():
            if hasattr(m, "pack_tokens"):
                m.pack_tokens = False
        tr = FusionTrainStep(enc, lr=rates[0], weight_decay=1e-3, grad_clip=1.0, optimizer_cls=opt_cls)
        batch = _batch(dev, 0)
        if graphed:
            gs = GraphedTrainStep(tr, batch, _loss, warmup=1)               # step 1 at rates[0]
            for r in rates[1:]:
                for g in tr.opt.param_groups:
                    g["lr"] = r
                gs.replay()
            gs.finish()
        else:
            for r in rates:
                for g in tr.opt.param_groups:
                    g["lr"] = r
                ops.clock_advance()
                tr.step([batch], _loss, on_clock=True)
        torch.cuda.synchronize()
        finals.append(tr.flat.flat.detach().cpu().clone())
    ops.clock_set(0)
    assert (finals[0] - finals[1]).abs().max().item() < 2e-5
    # ... and the schedule mattered: constant-rate steps end elsewhere
    assert (finals[0] - _constant_rate_reference(dev, rates[0], len(rates), opt_cls)).abs().max().item() > 1e-3


def _constant_rate_reference(dev, lr, steps, opt_cls):
    from transfusion_amd import ops
    from transfusion_amd.runner.trainer import FusionTrainStep
    ops.clock_set(0)
    enc = _encoder(dev)
    enc.token_dropout = enc.patch_dropout = 0.0
    for m in enc.modules():
        if hasattr(m, "pack_tokens"):
            m.pack_tokens = False
    tr = FusionTrainStep(enc, lr=lr, weight_decay=1e-3, grad_clip=1.0, optimizer_cls=opt_cls)
    batch = _batch(dev, 0)
    for _ in range(steps):
        ops.clock_advance()
        tr.step([batch], _loss, on_clock=True)
    torch.cuda.synchronize()
    ops.clock_set(0)
    return tr.flat.flat.detach().cpu().clone()
