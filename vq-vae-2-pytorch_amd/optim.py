"""Optimizer + LR schedule of the stage-1 step on the device.

  ParamArena     one flat fp32 arena for parameters and one for gradients (+ the EMA statistics
                 of both quantizers), so Adam is ONE launch and data-parallel training is ONE
                 RCCL all-reduce per step instead of DDP buckets + 4 small ones.
  FusedAdam      torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) math (train_vqvae.py:185)
  CycleScheduler the only scheduler stage 1 uses (/root/reference/scheduler.py:251-320), as a closed form of
                 one step counter
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

    def __init__(self, params, extra=0, last=()):
        """`last`: parameters to place at the END of both buffers (the ones whose gradients arrive last in the backward
        pass; Stage1Trainer puts enc_b there so that each data-parallel bucket is ONE contiguous slice)."""
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("ParamArena: no parameters")
        self.registration_order = list(params)
        tail_ids = {id(p) for p in last}
        params = [p for p in params if id(p) not in tail_ids] + [p for p in params if id(p) in tail_ids]
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
        ops.touch_weights(params)

    def canonical(self, flat):
        """A per-parameter flat state vector (Adam moments) re-ordered from arena order into REGISTRATION order -- the
        layout checkpoints carry, whatever order this arena uses internally."""
        return torch.cat([flat[self.offset[id(p)]:self.offset[id(p)] + p.numel()] for p in self.registration_order])

    def from_canonical(self, src, dst):
        off = 0
        for p in self.registration_order:
            dst[self.offset[id(p)]:self.offset[id(p)] + p.numel()].copy_(src[off:off + p.numel()])
            off += p.numel()
        if off != src.numel():
            raise RuntimeError("ParamArena: state vector belongs to a different model")

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
        if self.arena is not None:
            if len(self.param_groups) != 1:
                raise RuntimeError("FusedAdam: a ParamArena supports a single param group")
            if not self.arena.grads_ready():
                # falling back to the per-tensor path here would silently fork the optimizer state (fresh m/v)
                missing = [i for i, p in enumerate(self.arena.params)
                           if p.grad is None or p.grad.data_ptr() != p._vq2_grad.data_ptr()]
                raise RuntimeError(f"FusedAdam: {len(missing)} gradient(s) did not land in the flat arena (first: "
                                   f"parameter #{missing[0]}); every arena parameter must receive exactly one "
                                   "gradient per step (zero_grad() before each backward)")
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
        ops.touch_weights(p for g in self.param_groups for p in g["params"])  # raw-pointer update
        return None


# ------------------------------------------------------------------ CycleScheduler
# value after a proportion u of a segment that runs start -> end; the arithmetic is written exactly as the reference
# evaluates it, so the LR trajectory (a double handed to Adam) is BIT-identical to the reference scheduler's
_ANNEAL = {
    "linear": lambda start, end, u: start + u * (end - start),                      # scheduler.py:221-222
    "cos": lambda start, end, u: end + (start - end) / 2 * (cos(pi * u) + 1),       # scheduler.py:225-228
}


class CycleScheduler:
    """The one LR schedule stage 1 uses (/root/reference/scheduler.py:251-320; train_vqvae.py:188-195 builds it
    with momentum=None, warmup_proportion=0.05): a warm-up segment lr_max/divider -> lr_max over
    int(n_iter*warmup_proportion) steps, then an anneal segment lr_max -> lr_max/divider/1e4 over the rest, then the
    cycle restarts; `momentum=(hi, lo)` moves beta1 (or SGD momentum) hi -> lo -> hi along the same segments.

    Same constructor and step() contract as the reference (step() sets param_groups and returns (lr, momentum)),
    but stateless in form: the whole schedule is a function of ONE counter `t` (steps taken in the current cycle),
    which is also all that state_dict() has to carry for an exact resume."""

    def __init__(self, optimizer, lr_max, n_iter, momentum=(0.95, 0.85), divider=25, warmup_proportion=0.3,
                 phase=("linear", "cos")):
        self.optimizer = optimizer
        self.momentum = momentum
        warm = int(n_iter * warmup_proportion)
        self.lengths = (warm, n_iter - warm)
        self.anneal = (_ANNEAL[phase[0]], _ANNEAL[phase[1]])
        lo = lr_max / divider
        self.lr_ends = ((lo, lr_max), (lr_max, lo / 1e4))
        self.mom_ends = None if momentum is None else ((momentum[0], momentum[1]), (momentum[1], momentum[0]))
        self.t = 0

    def values(self, t):
        """(lr, momentum) after `t` steps of a cycle, 1 <= t <= n_iter."""
        seg = 0 if t <= self.lengths[0] else 1
        done = t if seg == 0 else t - self.lengths[0]
        if self.lengths[seg] == 0:
            # the reference divides by the segment length on its first step (scheduler.py:241): same failure
            raise ZeroDivisionError("CycleScheduler: a schedule segment has no iterations (n_iter too small)")
        u = done / self.lengths[seg]
        lr = self.anneal[seg](*self.lr_ends[seg], u)
        mom = None
        if self.mom_ends is not None:
            mom = self.anneal[seg](*self.mom_ends[seg], u)
        return lr, mom

    def step(self):
        if self.lengths[0] == 0:
            raise ZeroDivisionError("CycleScheduler: the warm-up segment has no iterations (n_iter too small)")
        self.t += 1
        lr, mom = self.values(self.t)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
            if mom is not None:
                if "betas" in group:
                    group["betas"] = (mom, group["betas"][1])
                else:
                    group["momentum"] = mom
        if self.t >= self.lengths[0] + self.lengths[1]:
            self.t = 0      # the next step starts a new cycle
        return lr, mom

    def state_dict(self):
        return {"t": self.t}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
