"""Fused RAdam on the device (tf_radam_step) with the reference's exact rectification schedule
(runner/metrics_losses/radam_optim.py:30-104) and global-norm clipping (run_experiment.py:444-446,
``gradient_clip_val`` with the norm algorithm).  The reference runs a Python loop of ~10 torch ops per
parameter tensor between backward and the next all-reduce; here one launch covers a whole tensor (ideally one flat
buffer holding every parameter).

``FusedRAdam`` is a ``torch.optim.Optimizer``: same constructor as the reference class (radam_optim.py:7-25), per-group
``lr`` / ``betas`` / ``eps`` / ``weight_decay`` (ego_nao_trainer.py:440-497 builds groups with ``lr / div_rate``), per-parameter
``state["step"]`` driving the rectification schedule, ``state_dict()`` / ``load_state_dict()`` (moments and step counts are
checkpointed), usable under the reference's LR schedulers (abc_nao_trainer.py:203-235).  Parameters whose ``grad`` is None
are skipped entirely, as in the reference (no moment update, no weight decay).
"""
from __future__ import annotations

import math

import torch

from transfusion_amd import _lib as L
from transfusion_amd import ops


def radam_schedule(step: int, beta1: float, beta2: float, degenerated_to_sgd: bool = False):
    """(N_sma, step_size, mode) -- radam_optim.py:64-84.  mode 1 = rectified update, 2 = SGD-degenerated, 0 = none."""
    beta2_t = beta2 ** step
    n_sma_max = 2 / (1 - beta2) - 1
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / (
            1 - beta1 ** step)
        return n_sma, step_size, 1
    if degenerated_to_sgd:
        return n_sma, 1.0 / (1 - beta1 ** step), 2
    return n_sma, -1.0, 0


class FusedRAdam(torch.optim.Optimizer):
    """``step(grad_scale=..., sumsq=..., clip=...)``: the gradient is multiplied by ``grad_scale`` first (1 / world size, loss
    scale) and, with ``sumsq`` + ``clip``, by torch's clip_grad_norm_ coefficient computed on the device."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        self.degenerated_to_sgd = degenerated_to_sgd
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # device scalars of the learning rates, by group INDEX -- kept OUTSIDE param_groups: state_dict() stays what the reference's RAdam
        # writes (no CUDA tensors in checkpoints), and load_state_dict(), which replaces the group dicts, leaves them -- and the pointer a
        # captured graph holds -- in place
        self._lr_dev = {}         # group index -> [tensor, value it holds]

    def lr_device(self, group, device):
        """The group's learning rate as a one-element device tensor (created on first use, holding the current ``group['lr']``): what
        ``step(on_clock=True)`` hands to the kernel, so that a step replayed from a HIP graph follows the LR scheduler.  The tensor is
        created once per group and refilled in place ever after; creating it inside a capture would bake the fill into the graph (every
        replay would reset the rate), so that raises."""
        gi = next(i for i, g in enumerate(self.param_groups) if g is group)
        ent = self._lr_dev.get(gi)
        if ent is None or ent[0].device != device:
            if torch.cuda.is_current_stream_capturing():
                raise L.TfError("FusedRAdam.lr_device: the rate's device scalar must exist before a capture (run one on_clock step eagerly first)")
            ent = self._lr_dev[gi] = [torch.full((1,), float(group["lr"]), dtype=torch.float32, device=device), float(group["lr"])]
        return ent[0]

    def refresh_lr(self):
        """Copies every group's current ``lr`` into its device scalar (no-op for groups whose rate has not changed).  Call OUTSIDE a graph
        capture, before a replay: the replay helper of tests/graph_step.py does."""
        for gi, ent in self._lr_dev.items():
            lr = float(self.param_groups[gi]["lr"])
            if ent[1] != lr:
                ent[0].fill_(lr)
                ent[1] = lr

    def load_state_dict(self, state_dict):
        """torch's load replaces the group dicts (and with them ``lr``): the device scalars are refilled IN PLACE from the loaded rates."""
        super().load_state_dict(state_dict)
        if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
            self.refresh_lr()

    def grad_sumsq(self, out: torch.Tensor):
        """Accumulates sum(g^2) of every gradient into the 1-element fp32 tensor ``out`` (device side)."""
        lib = L.load()
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    L.check(lib.tf_sumsq(L.ptr(p.grad), p.grad.numel(), L.ptr(out), ops._stream()), "tf_sumsq")

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, sumsq: torch.Tensor = None, clip: float = 0.0, on_clock: bool = False,
             zero_grad: bool = False):
        """``sumsq`` (1-element device tensor holding sum(g^2) over ALL gradients) + ``clip`` > 0 apply torch's
        clip_grad_norm_ coefficient on the device, with no host synchronisation.  ``on_clock``: the step number is read from the
        library's step clock on the device (``ops.clock_*``) and the schedule terms are formed there -- what a step captured in a
        HIP graph needs, since its host-computed terms would be frozen into the graph (tests/graph_step.py).  ``zero_grad``: every gradient
        is zeroed by the kernel once it has been read (the next step's zero fill, fused into this pass)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st = ops._stream()
        if on_clock and not torch.cuda.is_current_stream_capturing():
            self.refresh_lr()             # eager calls on the clock read the rate from the device scalar too: keep it current
        for g in self.param_groups:
            beta1, beta2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                ops._require_cuda(p)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise L.TfError("FusedRAdam needs contiguous fp32 parameters and gradients")
                s = self.state[p]
                if len(s) == 0:
                    s["step"] = 0
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                s["step"] += 1
                step = int(s["step"])
                n_sma, step_size, mode = radam_schedule(step, beta1, beta2, self.degenerated_to_sgd)
                a = L.TfRadamArgs(p=L.ptr(p), g=L.ptr(p.grad), m=L.ptr(s["exp_avg"]), v=L.ptr(s["exp_avg_sq"]), n=p.numel(),
                                  lr=g["lr"], beta1=beta1, beta2=beta2, eps=g["eps"], weight_decay=g["weight_decay"],
                                  beta2_t=beta2 ** step, bias1=1 - beta1 ** step, n_sma=n_sma,
                                  step_size=step_size, rectified=mode, grad_scale=grad_scale, sumsq=L.ptr(sumsq),
                                  clip=float(clip), zero_grad=1 if zero_grad else 0)
                if on_clock:      # step = step0 + *clock: right now, and after every advance of the clock (one per replay)
                    a.step_clock, a.step0, a.degenerated_to_sgd = ops.clock_ptr(), step - ops.clock_value(), int(self.degenerated_to_sgd)
                    a.lr_dev = L.ptr(self.lr_device(g, p.device))     # ... and the rate is read from the device (refresh_lr())
                L.call("tf_radam_step", a, st)
                # the kernel wrote through a raw pointer: tell autograd / the encoders' bf16 weight-shadow cache
                # (CrossTransformerModuleBox._wpack_dirty keys on (data_ptr, _version)) that the tensor changed
                torch._C._increment_version((p,))          # (takes an iterable of tensors)
        return loss
