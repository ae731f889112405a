"""The canonical stage-1 step of /root/reference/train_vqvae.py:83-91 (+166-171, 185-206) as an
object: forward, MSE + 0.25*latent, backward, ONE packed all-reduce (gradients + the EMA sums of
both quantizers) over RCCL, deferred EMA update, one-launch Adam, optional CycleScheduler.
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
    def __init__(self, model, lr=3e-4, sched=None, n_iter=None, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        live = model.live_parameters() if hasattr(model, "live_parameters") else list(model.parameters())
        self.quantizers = [m for m in model.modules() if type(m).__name__ == "Quantize"]
        extra = sum(q.n_embed * (q.dim + 1) for q in self.quantizers)
        self.arena = ParamArena(live, extra=extra)
        off = 0
        for q in self.quantizers:  # EMA statistics ride in the tail of the gradient buffer
            n = q.n_embed * (q.dim + 1)
            q.deferred_stats = self.arena.extra[off:off + n]
            off += n
        self.optimizer = FusedAdam(live, lr=lr, betas=betas, eps=eps, arena=self.arena)
        self.scheduler = None
        if sched == "cycle":  # train_vqvae.py:188-195
            self.scheduler = CycleScheduler(self.optimizer, lr, n_iter=n_iter, momentum=None, warmup_proportion=0.05)
        # every weight panel (forward and data-gradient layouts) re-packed by one launch per step
        layers = []
        for name, mod in model.named_modules():
            if hasattr(mod, "spec") and hasattr(mod, "weight") and not name.startswith("dec_ir"):
                layers.append((mod.spec, mod.weight, name != "enc_b.blocks.0"))
        self.wgrad_batch = ops.WgradBatch()
        self.pack_plan = ops.PackPlan(layers)
        self.pack_plan.run()
        self.world = dist_fn.get_world_size()
        self.optimizer.grad_scale = 1.0 / self.world  # DDP averages gradients (train_vqvae.py:166-171)
        self.comm_stream = torch.cuda.Stream() if self.world > 1 else None
        # Overlap: when the gradients of the layers that back-propagate first (decoder side; they sit at the
        # TAIL of the arena) are complete, their slice is all-reduced on the side stream while the
        # encoder is still back-propagating; the head (encoder gradients + EMA statistics) follows.
        self.split_off = None
        self._late_seen = 0
        self._bucket_sent = False
        split = getattr(model, "quantize_conv_b", None)
        if self.world > 1 and split is not None and os.environ.get("VQ2_DP_OVERLAP", "1") != "0":
            self.split_off = self.arena.n_extra + self.arena.offset[id(split.weight)]
            split.weight.register_post_accumulate_grad_hook(self._late_grad_ready)
            split.bias.register_post_accumulate_grad_hook(self._late_grad_ready)
        self.wgrad_stream = torch.cuda.Stream() if os.environ.get("VQ2_WGRAD_STREAM", "0") != "0" else None

    def _late_grad_ready(self, _param):
        self._late_seen += 1
        if self._late_seen == 2 and not self._bucket_sent:
            if self.wgrad_stream is not None:
                torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self.wgrad_batch.flush()                      # reduce the split-K slabs produced so far
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                dist.all_reduce(self.arena.flat_g[self.split_off:])
            self._bucket_sent = True

    def step(self, img, return_dec=False):
        """One training step on this rank's batch (img: NCHW); returns device scalars (no host sync)."""
        model = self.model
        model.train()
        self.arena.zero_grad()
        self.arena.extra.zero_()
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
        ops.WGRAD_STREAM[0] = self.wgrad_stream
        ops.WGRAD_BATCH[0] = self.wgrad_batch
        try:
            torch.autograd.backward(roots, seeds)
        finally:
            ops.WGRAD_STREAM[0] = None
            ops.WGRAD_BATCH[0] = None
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)   # all slabs written
        self.wgrad_batch.flush()   # one launch reduces the split-K slabs of every layer into the arena
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)   # all weight gradients have landed
        if self.world > 1:
            if not self.arena.grads_ready():
                raise RuntimeError("Stage1Trainer: a gradient did not land in the flat arena")
            # gradients (SUM; Adam divides by world) and EMA sums (SUM, vqvae.py:58-59) in one collective,
            # issued on a side stream so the host can already enqueue the next step's input work
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self._bucket_sent:
                    dist.all_reduce(self.arena.flat_g[:self.split_off])
                else:
                    dist.all_reduce(self.arena.flat_g)
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

    def state_dict(self):
        """Checkpoint in the reference's format (train_vqvae.py:205-206) plus optimizer state."""
        return {"model": self.model.state_dict(), "adam_m": self.optimizer._m, "adam_v": self.optimizer._v,
                "adam_t": self.optimizer._t}
