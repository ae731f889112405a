"""The canonical stage-1 step of /root/reference/train_vqvae.py:83-91 (+166-171, 185-206) as an
object: forward, MSE + 0.25*latent, backward, packed all-reduces (gradients + the EMA sums of both
quantizers) over RCCL on a side stream, deferred EMA update, one-launch Adam, optional CycleScheduler,
and a checkpoint that resumes the run exactly (model in the reference's format + optimizer + schedule).
"""
import os

import torch
from torch import distributed as dist

from . import distributed as dist_fn
from . import ops
from .optim import CycleScheduler, FusedAdam, ParamArena

LATENT_LOSS_WEIGHT = 0.25  # train_vqvae.py:34


def stage1_loss(dec, diff, img, latent_loss_weight=LATENT_LOSS_WEIGHT):
    """(loss, recon_loss, latent_loss) = MSE(dec,img) + w * diff.mean()  (train_vqvae.py:83-85)."""
    return ops.Stage1LossFn.apply(dec, diff, img, latent_loss_weight)


class Stage1Trainer:
    """Replaces `DistributedDataParallel(model)` + `optim.Adam` + the loop body of train_vqvae.py:83-91.

    Data parallelism (one process per GPU): like DDP's constructor (train_vqvae.py:166-171) the trainer first
    broadcasts rank 0's parameters and buffers, so replicas built from different RNG states start identical; every
    step then SUM-all-reduces the flat gradient buffer (Adam divides by the world size) in two buckets.
    VQ2_DP_FORCE=1 takes this path whenever a process group exists, also at world size 1."""

    def __init__(self, model, lr=3e-4, sched=None, n_iter=None, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        live = model.live_parameters() if hasattr(model, "live_parameters") else list(model.parameters())
        self.quantizers = [m for m in model.modules() if type(m).__name__ == "Quantize"]
        extra = sum(q.n_embed * (q.dim + 1) for q in self.quantizers)
        self.arena = ParamArena(live, extra=extra)
        off = 0
        for q in self.quantizers:  # EMA statistics ride in the head of the gradient buffer
            n = q.n_embed * (q.dim + 1)
            q.deferred_stats = self.arena.extra[off:off + n]
            off += n
        self.optimizer = FusedAdam(live, lr=lr, betas=betas, eps=eps, arena=self.arena)
        self.scheduler = None
        if sched == "cycle":  # train_vqvae.py:188-195
            self.scheduler = CycleScheduler(self.optimizer, lr, n_iter=n_iter, momentum=None, warmup_proportion=0.05)
        # this trainer's backward-pass state hangs on ITS parameters (no process-global state)
        wgrad_stream = torch.cuda.Stream() if os.environ.get("VQ2_WGRAD_STREAM", "1") != "0" else None
        self.ctx = ops.StepContext(wgrad_stream)
        for p in self.arena.params:
            p._vq2_ctx = self.ctx
        if os.environ.get("VQ2_STATS_STREAM", "1") != "0":
            for q in self.quantizers:      # the EMA statistics are consumed after backward: off the main stream
                q.stats_stream = wgrad_stream
        # every weight panel (forward and data-gradient layouts) re-packed by one launch per step
        layers = []
        for name, mod in model.named_modules():
            if hasattr(mod, "spec") and hasattr(mod, "weight") and not name.startswith("dec_ir"):
                layers.append((mod.spec, mod.weight, name != "enc_b.blocks.0"))
        self.pack_plan = ops.PackPlan(layers)

        self.world = dist_fn.get_world_size()
        forced = os.environ.get("VQ2_DP_FORCE", "0") != "0" and dist.is_available() and dist.is_initialized()
        self.dp = self.world > 1 or forced
        self.optimizer.grad_scale = 1.0 / self.world  # DDP averages gradients (train_vqvae.py:166-171)
        self.comm_stream = torch.cuda.Stream() if self.dp else None
        self.comm = dist_fn.data_comm() if self.dp else None   # torch.distributed group or libvq2's own communicator
        if self.dp:
            self._sync_initial_state()
        self.pack_plan.run()
        # Overlap: when the gradients of the layers that back-propagate first (decoder side; they sit at the
        # TAIL of the arena) are complete, their slice is all-reduced on the side stream while the
        # encoder is still back-propagating; the head (encoder gradients + EMA statistics) follows.
        self.split_off = None
        self.tail_params = []
        self._late_seen = 0
        self._bucket_sent = False
        self.early_buckets = 0      # steps in which the tail bucket went out from the backward hook
        split = getattr(model, "quantize_conv_b", None)
        self.early_flush = os.environ.get("VQ2_EARLY_FLUSH", "1") != "0" and self.ctx.stream is not None
        if split is not None and (self.early_flush or (self.dp and os.environ.get("VQ2_DP_OVERLAP", "1") != "0")):
            first = self.arena.offset[id(split.weight)]
            if self.dp and os.environ.get("VQ2_DP_OVERLAP", "1") != "0":
                self.split_off = self.arena.n_extra + first
            self.tail_params = [p for p in self.arena.params if self.arena.offset[id(p)] >= first]
            split.weight.register_post_accumulate_grad_hook(self._late_grad_ready)
            split.bias.register_post_accumulate_grad_hook(self._late_grad_ready)

    # ------------------------------------------------------------------ data parallel plumbing
    def _sync_initial_state(self):
        """What DistributedDataParallel's constructor does (train_vqvae.py:166-171): rank 0's parameters and
        buffers everywhere.  One broadcast for the whole arena, one per tensor outside it (dead dec_ir, codebooks)."""
        with torch.no_grad():
            self.comm.broadcast(self.arena.flat_p, 0)
            inside = {id(p) for p in self.arena.params}
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                if id(t) not in inside:
                    self.comm.broadcast(t.detach(), 0)     # (detach() shares the version counter; .data does not)
        ops.touch_weights(self.arena.params)
        # the codebooks were just overwritten (NativeComm writes through raw pointers, which no version counter sees):
        # anything a Quantize derived from its OLD codebook is stale now
        for q in self.quantizers:
            q.invalidate_prepared()

    def _late_grad_ready(self, _param):
        self._late_seen += 1
        if self._late_seen != 2 or self._bucket_sent:
            return
        # The early bucket is legal only if every parameter of the tail slice has ALREADY produced its gradient
        # (autograd happens to run upsample_t / dec before quantize_conv_b; nothing else guarantees it): otherwise
        # keep everything for the single all-reduce after backward.
        if not all(p.grad is not None and p.grad.data_ptr() == p._vq2_grad.data_ptr() for p in self.tail_params):
            return
        # The decoder-side split-K slabs (most of the 546 MB) are reduced into the arena NOW, on the side stream when
        # there is one, beside the encoder's matrix-bound backward launches -- instead of in one HBM-bound pass after
        # backward; data parallel: that slice of the gradient buffer then goes out while the encoder back-propagates.
        side = self.ctx.stream
        if side is not None:
            side.wait_event(torch.cuda.current_stream().record_event())
            with torch.cuda.stream(side):
                self.ctx.batch.flush()
                ev = side.record_event()
        else:
            self.ctx.batch.flush()
            ev = torch.cuda.current_stream().record_event()
        if self.split_off is None:       # single GPU: nothing to send
            self._bucket_sent = True
            return
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            self.comm.all_reduce(self.arena.flat_g[self.split_off:])
        self._bucket_sent = True
        self.early_buckets += 1

    # ------------------------------------------------------------------ the step
    def step(self, img, return_dec=False):
        """One training step on this rank's batch (img: NCHW); returns device scalars (no host sync)."""
        model = self.model
        model.train()
        self.arena.zero_grad()   # the EMA statistics slots are fully rewritten by vq2_vq_stats: nothing to zero
        if hasattr(model, "forward_nhwc"):
            # loss evaluated in the kernels' own NHWC4 layout: no layout conversion of the
            # reconstruction or of its gradient (the zero pad lane contributes nothing)
            x = ops.to_nhwc(img)
            dec, diff = model.forward_nhwc(x)
            loss, recon, latent, d_dec, d_diff = ops.stage1_loss_and_seeds(dec, diff, x, LATENT_LOSS_WEIGHT, img.numel())
            roots, seeds = (dec, diff), (d_dec, d_diff)
        else:
            dec, diff = model(img)
            loss, recon, latent = stage1_loss(dec, diff, img)
            roots, seeds = (loss,), (None,)
        self._late_seen, self._bucket_sent = 0, False
        self.ctx.active = True
        try:
            torch.autograd.backward(roots, seeds)
        finally:
            self.ctx.active = False
        if self.ctx.stream is not None:
            torch.cuda.current_stream().wait_stream(self.ctx.stream)   # all slabs written
        self.ctx.batch.flush()   # one launch reduces the split-K slabs of every layer into the arena
        if self.dp:
            if not self.arena.grads_ready():
                raise RuntimeError("Stage1Trainer: a gradient did not land in the flat arena")
            # gradients (SUM; Adam divides by world) and EMA sums (SUM, vqvae.py:58-59) in one collective,
            # issued on a side stream so the host can already enqueue the next step's input work
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self._bucket_sent:
                    self.comm.all_reduce(self.arena.flat_g[:self.split_off])
                else:
                    self.comm.all_reduce(self.arena.flat_g)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for q in self.quantizers:
            q.apply_deferred_update()
        if self.scheduler is not None:
            self.scheduler.step()
        self.optimizer.step()
        self.pack_plan.run()
        out = {"loss": loss.detach(), "recon": recon, "latent": latent}
        if return_dec:
            d = dec.detach()
            out["dec"] = ops.from_nhwc(d, img.shape[1]) if hasattr(model, "forward_nhwc") else d
        return out

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """"model" is exactly what the reference saves (train_vqvae.py:205-206: model.state_dict(), loadable by
        extract_code.py / sample.py); the rest is what the reference lacks for an exact resume: Adam moments and
        step count, learning rate / betas, the schedule position."""
        group = self.optimizer.param_groups[0]
        return {"model": {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                "adam_m": self.optimizer._m.clone(), "adam_v": self.optimizer._v.clone(), "adam_t": self.optimizer._t,
                "lr": group["lr"], "betas": tuple(group["betas"]),
                "scheduler": None if self.scheduler is None else self.scheduler.state_dict()}

    def load_state_dict(self, sd):
        """Inverse of state_dict(); also accepts a bare model state_dict (the reference's --resume file,
        train_vqvae.py:173-182, with or without DDP's "module." prefix): optimizer state then starts fresh."""
        full = "model" in sd and isinstance(sd["model"], dict)
        model_sd = sd["model"] if full else sd
        model_sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in model_sd.items()}
        self.model.load_state_dict(model_sd)          # copies INTO the arena views: parameters stay re-homed
        ops.touch_weights(self.arena.params)
        if full:
            if sd["adam_m"].numel() != self.optimizer._m.numel():
                raise RuntimeError("Stage1Trainer.load_state_dict: optimizer state belongs to a different model")
            self.optimizer._m.copy_(sd["adam_m"])
            self.optimizer._v.copy_(sd["adam_v"])
            self.optimizer._t = int(sd["adam_t"])
            for group in self.optimizer.param_groups:
                group["lr"], group["betas"] = sd["lr"], tuple(sd["betas"])
            if (self.scheduler is None) != (sd.get("scheduler") is None):
                raise RuntimeError("Stage1Trainer.load_state_dict: checkpoint and trainer disagree about --sched")
            if self.scheduler is not None:
                self.scheduler.load_state_dict(sd["scheduler"])
        self.pack_plan.run()
