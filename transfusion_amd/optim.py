"""Fused RAdam on the device (tf_radam_step) with the reference's exact rectification schedule
(runner/metrics_losses/radam_optim.py:30-104) and global-norm clipping (run_experiment.py:444-446,
``gradient_clip_val`` with the norm algorithm).  The reference runs a Python loop of ~10 torch ops per
parameter tensor between backward and the next all-reduce; here one launch covers a whole flat buffer.
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


class FusedRAdam:
    """Groups of fp32 tensors (ideally a few large flat buffers).  ``step(grad_scale)`` multiplies the
    gradient first (1/world_size, loss scale, clip coefficient)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, degenerated_to_sgd=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if isinstance(params, (list, tuple)) and len(params) and isinstance(params[0], dict):
            self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **g) for g in params]
        else:
            self.param_groups = [dict(params=list(params), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.degenerated_to_sgd = degenerated_to_sgd
        self.state = {}
        self.step_count = 0

    def zero_grad(self):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    p.grad.zero_()

    def grad_sumsq(self, out: torch.Tensor):
        """Accumulates sum(g^2) of every gradient into the 1-element fp32 tensor ``out`` (device side)."""
        lib = L.load()
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    L.check(lib.tf_sumsq(L.ptr(p.grad), p.grad.numel(), L.ptr(out), ops._stream()), "tf_sumsq")

    def step(self, grad_scale: float = 1.0, sumsq: torch.Tensor = None, clip: float = 0.0):
        """``sumsq`` (1-element device tensor holding sum(g^2) over ALL gradients) + ``clip`` > 0 apply torch's
        clip_grad_norm_ coefficient on the device, with no host synchronisation."""
        self.step_count += 1
        st = ops._stream()
        for g in self.param_groups:
            beta1, beta2 = g["betas"]
            n_sma, step_size, mode = radam_schedule(self.step_count, beta1, beta2, self.degenerated_to_sgd)
            for p in g["params"]:
                if p.grad is None:
                    continue
                ops._require_cuda(p)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise L.TfError("FusedRAdam needs contiguous fp32 parameters and gradients")
                s = self.state.get(id(p))
                if s is None:
                    s = {"exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                    self.state[id(p)] = s
                a = L.TfRadamArgs(p=L.ptr(p), g=L.ptr(p.grad), m=L.ptr(s["exp_avg"]), v=L.ptr(s["exp_avg_sq"]), n=p.numel(),
                                  lr=g["lr"], beta1=beta1, beta2=beta2, eps=g["eps"], weight_decay=g["weight_decay"],
                                  beta2_t=beta2 ** self.step_count, bias1=1 - beta1 ** self.step_count, n_sma=n_sma,
                                  step_size=step_size, rectified=mode, grad_scale=grad_scale, sumsq=L.ptr(sumsq),
                                  clip=float(clip))
                L.call("tf_radam_step", a, st)
