"""Optimizer + LR schedule of the stage-1 step on the device.

  ParamArena     one flat fp32 arena for parameters and one for gradients (+ the EMA statistics
                 of both quantizers), so Adam is ONE launch and data-parallel training is ONE
                 RCCL all-reduce per step instead of DDP buckets + 4 small ones.
  FusedAdam      torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) math (train_vqvae.py:185)
  CycleScheduler /root/reference/scheduler.py:221-320 (the only scheduler stage 1 uses)
"""
import ctypes as C
from math import cos, pi

import torch

from . import ops
from ._lib import lib, check


class ParamArena:
    """Re-homes `params` into one contiguous buffer (p.data become views; Parameter identity and
    state_dict are unchanged) and pre-assigns each parameter's gradient slot: the wgrad / bias-grad
    launches then write straight into the flat gradient buffer (ops.conv_wgrad / bias_grad look at
    `param._vq2_grad`).  `extra` floats are appended to the gradient buffer for the VQ statistics."""

    def __init__(self, params, extra=0):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("ParamArena: no parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("ParamArena: parameters must live on the MI355X (model.cuda() first)")
        self.params = params
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]  # keep every slot 16-byte aligned
        self.n = sum(sizes)
        self.flat_p = torch.zeros(self.n, device=dev, dtype=torch.float32)
        # gradient buffer = [extra (VQ statistics) | parameter gradients in registration order]; backward
        # produces gradients roughly in REVERSE registration order, so the tail of this buffer is
        # complete first and can be all-reduced while the encoder is still back-propagating
        self.n_extra = (extra + 3) // 4 * 4
        self.flat_g = torch.zeros(self.n_extra + self.n, device=dev, dtype=torch.float32)
        self.extra = self.flat_g[:extra]
        self.gp = self.flat_g[self.n_extra:]
        self.offset = {}
        off = 0
        with torch.no_grad():
            for p, sz in zip(params, sizes):
                view = self.flat_p[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._vq2_grad = self.gp[off:off + p.numel()].view(p.shape)
                self.offset[id(p)] = off
                off += sz
        ops.WEIGHT_EPOCH[0] += 1

    def grads_ready(self):
        return all(p.grad is not None and p.grad.data_ptr() == p._vq2_grad.data_ptr() for p in self.params)

    def zero_grad(self):
        for p in self.params:
            p.grad = None


class FusedAdam(torch.optim.Optimizer):
    """Adam over libvq2's vq2_adam_step.  With a ParamArena whose gradients all landed in the flat
    buffer the whole update is one launch; otherwise one launch per parameter tensor."""

    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, arena=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.arena = arena
        self.grad_scale = 1.0  # 1/world_size when gradients were SUM-all-reduced
        self._t = 0
        if arena is not None:
            self._m = torch.zeros(arena.n, device=arena.flat_p.device)
            self._v = torch.zeros(arena.n, device=arena.flat_p.device)

    @torch.no_grad()
    def step(self, closure=None):
        s = ops._stream()
        group = self.param_groups[0]
        lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
        if self.arena is not None and len(self.param_groups) == 1 and self.arena.grads_ready():
            self._t += 1
            a = self.arena
            check(lib.vq2_adam_step(ops._p(a.flat_p), ops._p(a.gp), ops._p(self._m), ops._p(self._v), a.n,
                                    lr, b1, b2, eps, self._t, self.grad_scale, s), "adam_step")
        else:
            for group in self.param_groups:
                lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    st = self.state[p]
                    if not st:
                        st["step"] = 0
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["step"] += 1
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    check(lib.vq2_adam_step(ops._p(p.data), ops._p(g), ops._p(st["exp_avg"]),
                                            ops._p(st["exp_avg_sq"]), p.numel(), lr, b1, b2, eps, st["step"],
                                            self.grad_scale, s), "adam_step")
        ops.WEIGHT_EPOCH[0] += 1  # raw-pointer update: invalidate packed weights
        return None


# ------------------------------------------------------------------ scheduler.py:221-320
def anneal_linear(start, end, proportion):
    return start + proportion * (end - start)


def anneal_cos(start, end, proportion):
    return end + (start - end) / 2 * (cos(pi * proportion) + 1)


class Phase:
    def __init__(self, start, end, n_iter, anneal_fn):
        self.start, self.end, self.n_iter, self.anneal_fn = start, end, n_iter, anneal_fn
        self.n = 0

    def step(self):
        self.n += 1
        return self.anneal_fn(self.start, self.end, self.n / self.n_iter)

    def reset(self):
        self.n = 0

    @property
    def is_done(self):
        return self.n >= self.n_iter


class CycleScheduler:
    """Linear warm-up then cosine anneal of lr (and optionally beta1), restarting after n_iter."""

    def __init__(self, optimizer, lr_max, n_iter, momentum=(0.95, 0.85), divider=25, warmup_proportion=0.3,
                 phase=("linear", "cos")):
        self.optimizer = optimizer
        phase1 = int(n_iter * warmup_proportion)
        phase2 = n_iter - phase1
        lr_min = lr_max / divider
        fns = {"linear": anneal_linear, "cos": anneal_cos}
        self.lr_phase = [Phase(lr_min, lr_max, phase1, fns[phase[0]]),
                         Phase(lr_max, lr_min / 1e4, phase2, fns[phase[1]])]
        self.momentum = momentum
        if momentum is not None:
            m1, m2 = momentum
            self.momentum_phase = [Phase(m1, m2, phase1, fns[phase[0]]), Phase(m2, m1, phase2, fns[phase[1]])]
        else:
            self.momentum_phase = []
        self.phase = 0

    def step(self):
        lr = self.lr_phase[self.phase].step()
        momentum = self.momentum_phase[self.phase].step() if self.momentum is not None else None
        for group in self.optimizer.param_groups:
            group["lr"] = lr
            if self.momentum is not None:
                if "betas" in group:
                    group["betas"] = (momentum, group["betas"][1])
                else:
                    group["momentum"] = momentum
        if self.lr_phase[self.phase].is_done:
            self.phase += 1
        if self.phase >= len(self.lr_phase):
            for ph in self.lr_phase:
                ph.reset()
            for ph in self.momentum_phase:
                ph.reset()
            self.phase = 0
        return lr, momentum
