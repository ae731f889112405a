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
        # enc_b back-propagates LAST: its parameters go to the end of the arena, so that the gradient buffer reads
        # [EMA statistics | enc_t quantize_conv_t dec_t | quantize_conv_b upsample_t dec | enc_b] and each of the three
        # data-parallel buckets (below) is one contiguous slice
        enc_b = getattr(model, "enc_b", None)
        self.arena = ParamArena(live, extra=extra, last=list(enc_b.parameters()) if enc_b is not None else ())
        off = 0
        for q in self.quantizers:  # EMA statistics ride in the head of the gradient buffer
            n = q.n_embed * (q.dim + 1)
            q.deferred_stats = self.arena.extra[off:off + n]
            off += n
        self.optimizer = FusedAdam(live, lr=lr, betas=betas, eps=eps, arena=self.arena)
        self.scheduler = None
        if sched == "cycle":  # train_vqvae.py:188-195
            self.scheduler = CycleScheduler(self.optimizer, lr, n_iter=n_iter, momentum=None, warmup_proportion=0.05)
        # Stream priorities: the step's main chain should win every CU slot it can use, the side streams (small weight
        # gradients, EMA statistics, early slab reduction, collectives) only what it leaves idle -- mostly the tails of
        # its launches.  HIP knows two levels (0 and -1 = high), so the trainer installs a HIGH-priority stream as this
        # thread's current stream, once (ordered after the work already queued on the previous one), and creates its
        # side streams at the default priority: 6.78 -> 6.74 ms per step.  VQ2_MAIN_PRIO=0 leaves the current stream alone.
        if os.environ.get("VQ2_MAIN_PRIO", "1") != "0" and torch.cuda.current_stream().priority == 0:
            hp = torch.cuda.Stream(priority=-1)
            hp.wait_stream(torch.cuda.current_stream())
            torch.cuda.set_stream(hp)
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
        # Overlap (train_vqvae.py:166-171 is DDP's bucketed all-reduce; vqvae.py:58-59 the EMA sums).  The backward pass
        # produces gradients in the order dec, upsample_t, quantize_conv_b | dec_t, quantize_conv_t, enc_t | enc_b, so
        # the gradient buffer goes out in THREE contiguous slices, each all-reduced on the communication stream while
        # the next group is still back-propagating:
        #   tail    [quantize_conv_b upsample_t dec]             when quantize_conv_b's gradient lands   (~60 %)
        #   middle  [EMA statistics | enc_t quantize_conv_t dec_t] when enc_t's first conv's gradient lands
        #   head    [enc_b]                                       after backward: the only exposed part (~1.4 MB)
        # A bucket goes out early only if every parameter of its slice already owns its gradient slot (autograd happens
        # to order the graph this way; nothing else guarantees it) -- whatever was not sent early is sent after backward.
        self.early_buckets = 0      # buckets that went out from a backward hook (all steps)
        self.bucket_bytes = {}      # name -> bytes of the slice (bench line)
        self._sent = []             # [lo, hi) slices of flat_g already handed to the communicator in this step
        self._hooks_seen = {}
        self.buckets = []           # (name, lo, hi, params of the slice)
        self._send_early = False
        self.early_flush = os.environ.get("VQ2_EARLY_FLUSH", "1") != "0" and self.ctx.stream is not None
        overlap = self.dp and os.environ.get("VQ2_DP_OVERLAP", "1") != "0"
        split = getattr(model, "quantize_conv_b", None)
        enc_t = getattr(model, "enc_t", None)
        if split is not None and enc_b is not None and enc_t is not None and (self.early_flush or overlap):
            a = self.arena
            off = lambda p: a.n_extra + a.offset[id(p)]
            lo_tail = off(split.weight)
            lo_head = min(off(p) for p in enc_b.parameters())
            total = a.flat_g.numel()
            inside = lambda lo, hi: [p for p in a.params if lo <= off(p) < hi]
            self.buckets = [("tail", lo_tail, lo_head, inside(lo_tail, lo_head)),
                            ("middle", 0, lo_tail, inside(0, lo_tail)),
                            ("head", lo_head, total, inside(lo_head, total))]
            self.bucket_bytes = {name: 4 * (hi - lo) for name, lo, hi, _ in self.buckets}
            self._send_early = overlap
            for prm in (split.weight, split.bias):
                prm.register_post_accumulate_grad_hook(lambda _p: self._group_done("tail", 2))
            if overlap:     # (single GPU: one early slab reduction is enough, a second one only adds a launch)
                first = enc_t.blocks[0]
                for prm in (first.weight, first.bias):
                    prm.register_post_accumulate_grad_hook(lambda _p: self._group_done("middle", 2))

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

    def _group_done(self, name, need):
        """Post-accumulate hook of the LAST layer of a backward group: reduce the split-K slabs produced so far into the
        arena (side stream) and, data parallel, hand the group's slice to the communicator while the next group
        back-propagates."""
        seen = self._hooks_seen[name] = self._hooks_seen.get(name, 0) + 1
        if seen != need:
            return
        _, lo, hi, params = next(b for b in self.buckets if b[0] == name)
        if not all(p.grad is not None and p.grad.data_ptr() == p._vq2_grad.data_ptr() for p in params):
            return      # autograd ordered the graph differently: keep the slice for the collective after backward
        side = self.ctx.stream
        if side is not None:
            side.wait_event(torch.cuda.current_stream().record_event())
            with torch.cuda.stream(side):
                self.ctx.batch.flush()
                ev = side.record_event()
        else:
            self.ctx.batch.flush()
            ev = torch.cuda.current_stream().record_event()
        if not self._send_early:
            return
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            self.comm.all_reduce(self.arena.flat_g[lo:hi])
        self._sent.append((lo, hi))
        self.early_buckets += 1

    def _unsent_slices(self):
        """Complement of the slices already sent, as maximal contiguous [lo, hi) ranges of flat_g."""
        out, pos = [], 0
        for lo, hi in sorted(self._sent):
            if lo > pos:
                out.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.arena.flat_g.numel():
            out.append((pos, self.arena.flat_g.numel()))
        return out

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
        self._sent, self._hooks_seen = [], {}
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
                for lo, hi in self._unsent_slices():     # normally just enc_b's slice; everything if no hook fired
                    self.comm.all_reduce(self.arena.flat_g[lo:hi])
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
                # Adam moments in REGISTRATION order (the arena's internal order is an implementation detail)
                "adam_m": self.arena.canonical(self.optimizer._m), "adam_v": self.arena.canonical(self.optimizer._v),
                "adam_t": self.optimizer._t,
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
            n_real = sum(p.numel() for p in self.arena.params)
            if sd["adam_m"].numel() == self.optimizer._m.numel() and sd["adam_m"].numel() != n_real:
                # a round-2 checkpoint: moments in that arena's padded registration order
                raise RuntimeError("Stage1Trainer.load_state_dict: optimizer state uses the pre-round-3 padded layout; "
                                   "re-save it with that version or resume from the model weights alone")
            if sd["adam_m"].numel() != n_real:
                raise RuntimeError("Stage1Trainer.load_state_dict: optimizer state belongs to a different model")
            dev = self.optimizer._m.device
            self.arena.from_canonical(sd["adam_m"].to(dev), self.optimizer._m)
            self.arena.from_canonical(sd["adam_v"].to(dev), self.optimizer._v)
            self.optimizer._t = int(sd["adam_t"])
            for group in self.optimizer.param_groups:
                group["lr"], group["betas"] = sd["lr"], tuple(sd["betas"])
            if (self.scheduler is None) != (sd.get("scheduler") is None):
                raise RuntimeError("Stage1Trainer.load_state_dict: checkpoint and trainer disagree about --sched")
            if self.scheduler is not None:
                self.scheduler.load_state_dict(sd["scheduler"])
        self.pack_plan.run()
