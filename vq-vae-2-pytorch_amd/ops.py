"""torch.autograd.Function wrappers over the C ABI of libvq2.so (include/vq2.h).

Activations inside this module are NHWC tensors of shape [N,H,W,C] whose last dim is
contiguous and whose pixel stride `ld = t.stride(2)` may exceed C (a channel slice of a
wider buffer).  PyTorch is used for device memory, streams and autograd bookkeeping only;
every FLOP and every byte moved on this path is a libvq2 kernel.
"""
from dataclasses import dataclass
import ctypes as C
import os

import torch
from torch.autograd import Function

from ._lib import lib, check, ConvDesc, PackJob, WgradJob

VQ2_RELU_IN = 1
VQ2_RELU_OUT = 2
VQ2_MASK_AFTER_RESIDUAL = 4
PACK_FWD = 0
PACK_DGRAD = 1

# There is NO process-global mutable state on the autograd path (SURVEY 8b threading row): the packed-weight
# cache lives on each weight tensor (`_vq2_packs`), raw-pointer weight updates are tracked per parameter
# (`_vq2_epoch`, see touch_weights), and the deferred weight-gradient reduction / side stream of a training step
# belong to the StepContext its Stage1Trainer hangs on its own parameters (`_vq2_ctx`), so two trainers or
# models can coexist in one process and back-propagate on different autograd threads.


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def ceil4(c):
    return (c + 3) // 4 * 4


def _require_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"vqvae2_amd: {what} must be a float32 tensor on the MI355X (got "
                           f"{getattr(t, 'dtype', type(t))} on {getattr(t, 'device', '?')}); "
                           "this package has no CPU path")


def is_nhwc_dense(t):
    """[N,H,W,C] with contiguous channels and a uniform pixel stride, 16-byte aligned."""
    if t.dim() != 4 or t.stride(3) != 1:
        return False
    n, h, w, c = t.shape
    ld = t.stride(2)
    if w == 1:
        ld = t.stride(1) if h > 1 else (t.stride(0) if n > 1 else c)
    return (ld >= c and ld % 4 == 0 and (w == 1 or t.stride(2) == ld) and (h == 1 or t.stride(1) == w * ld)
            and (n == 1 or t.stride(0) == h * w * ld) and t.data_ptr() % 16 == 0)


def ld_of(t):
    n, h, w, c = t.shape
    if w > 1:
        return t.stride(2)
    if h > 1:
        return t.stride(1)
    if n > 1:
        return t.stride(0)
    return c


def as_nhwc(t):
    """Return t if it already is a dense NHWC view (any pixel stride), else a packed copy."""
    _require_cuda(t, "activation")
    if is_nhwc_dense(t):
        return t
    return t.contiguous()  # exotic layout handed in by foreign code; never hit on the stage-1 path


def packed(t):
    """Dense NHWC view -> pixel stride == C (slice-copy kernel when it is a channel slice)."""
    if t.is_contiguous():
        return t
    t = as_nhwc(t)
    if t.is_contiguous():
        return t
    n, h, w, c = t.shape
    out = torch.empty((n, h, w, c), device=t.device, dtype=torch.float32)
    check(lib.vq2_slice_copy(_p(t), ld_of(t), _p(out), c, n * h * w, c, 0, _stream()), "slice_copy")
    return out


# ----------------------------------------------------------------------------- conv plumbing
@dataclass(frozen=True)
class ConvSpec:
    transposed: bool
    cin: int
    cout: int
    k: int
    stride: int
    pad: int

    @property
    def ci(self):
        return ceil4(self.cin)

    @property
    def co(self):
        return ceil4(self.cout)

    def out_hw(self, h, w):
        if self.transposed:
            return 2 * h, 2 * w
        return (h + 2 * self.pad - self.k) // self.stride + 1, (w + 2 * self.pad - self.k) // self.stride + 1


def _desc(spec, n, h, w, ldx, ldy):
    d = ConvDesc()
    d.N, d.H, d.W, d.Ci, d.Co = n, h, w, spec.ci, spec.co
    d.KH = d.KW = spec.k
    d.stride, d.pad, d.transposed = spec.stride, spec.pad, int(spec.transposed)
    d.ldx, d.ldy = ldx, ldy
    d.Cir, d.Cor = spec.cin, spec.cout
    return d


def touch_weights(params):
    """Tell the packed-weight caches that `params` were updated through raw pointers (vq2_adam_step), which
    does not bump tensor._version."""
    for p in params:
        p._vq2_epoch = getattr(p, "_vq2_epoch", 0) + 1


def _pack_version(spec, weight):
    return (weight.data_ptr(), weight._version, getattr(weight, "_vq2_epoch", 0), spec)


def packed_weight(spec, weight, which):
    """Kernel-layout copy of a reference-layout weight, cached ON the weight until it changes."""
    packs = weight.__dict__.setdefault("_vq2_packs", {})
    ver = _pack_version(spec, weight)
    hit = packs.get(which)
    if hit is not None and hit[0] == ver:
        return hit[1]
    _require_cuda(weight, "weight")
    wsrc = weight.detach()
    if not wsrc.is_contiguous():
        wsrc = wsrc.contiguous()
    n = spec.ci * spec.co * spec.k * spec.k
    buf = hit[1] if (hit is not None and hit[1].numel() == n and hit[1].device == weight.device) else \
        torch.empty(n, device=weight.device, dtype=torch.float32)
    d = _desc(spec, 1, max(spec.k, 2), max(spec.k, 2), spec.ci, spec.co)
    check(lib.vq2_pack_weight(C.byref(d), which, _p(wsrc), _p(buf), _stream()), "pack_weight")
    packs[which] = (ver, buf)
    return buf


class PackPlan:
    """Every (layer, fwd|dgrad) weight panel of a model re-packed by ONE launch per step
    (vq2_pack_weights_batched) instead of one launch per panel; feeds the same cache that
    packed_weight() reads, so the conv wrappers need no change."""

    def __init__(self, layers):
        """layers: iterable of (ConvSpec, weight Parameter, needs_dgrad)."""
        import numpy as np
        jobs = []
        for spec, weight, needs_dgrad in layers:
            for which in ((PACK_FWD, PACK_DGRAD) if needs_dgrad else (PACK_FWD,)):
                jobs.append((spec, weight, which))
        n = sum(s.ci * s.co * s.k * s.k for s, _, _ in jobs)
        dev = jobs[0][1].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.entries = []
        arr = (PackJob * len(jobs))()
        off = 0
        for i, (spec, weight, which) in enumerate(jobs):
            numel = spec.ci * spec.co * spec.k * spec.k
            buf = self.flat[off:off + numel]
            d = _desc(spec, 1, max(spec.k, 2), max(spec.k, 2), spec.ci, spec.co)
            if not weight.is_contiguous():
                raise RuntimeError("PackPlan: weights must be contiguous")
            check(lib.vq2_pack_job_init(C.byref(d), which, _p(weight), _p(buf), C.byref(arr[i])), "pack_job_init")
            arr[i].offset = off
            assert arr[i].numel == numel
            self.entries.append((spec, weight, which, buf))
            off += numel
        self.total = off
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self.jobs_dev = torch.from_numpy(raw).to(dev)
        self.njobs = len(jobs)
        self._ptrs = [w.data_ptr() for _, w, _, _ in self.entries]

    def run(self):
        if any(w.data_ptr() != p for (_, w, _, _), p in zip(self.entries, self._ptrs)):
            raise RuntimeError("PackPlan: a parameter was re-allocated; rebuild the plan")
        check(lib.vq2_pack_weights_batched(_p(self.jobs_dev), self.njobs, self.total, _stream()), "pack_batched")
        for spec, weight, which, buf in self.entries:
            weight.__dict__.setdefault("_vq2_packs", {})[which] = (_pack_version(spec, weight), buf)


def conv_forward(spec, x, weight, bias, flags=0, residual=None, out=None):
    n, h, w, c = x.shape
    if c != spec.ci:
        raise RuntimeError(f"conv_forward: input has {c} channels, expected {spec.ci}")
    ho, wo = spec.out_hw(h, w)
    if out is None:
        out = torch.empty((n, ho, wo, spec.co), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (n, ho, wo, spec.co) or not is_nhwc_dense(out):
        raise RuntimeError("conv_forward: bad `out` buffer")
    d = _desc(spec, n, h, w, ld_of(x), ld_of(out))
    wp = packed_weight(spec, weight, PACK_FWD)
    ldres = ld_of(residual) if residual is not None else 0
    check(lib.vq2_conv_fwd(C.byref(d), flags, _p(x), _p(wp), _p(bias), _p(residual), ldres, _p(out), _stream()),
          "conv_fwd")
    return out


def conv_dgrad(spec, xshape, dy, weight, mask=None, residual=None, out=None, mask_after=False):
    n, h, w, _ = xshape
    if out is None:
        out = torch.empty((n, h, w, spec.ci), device=dy.device, dtype=torch.float32)
    d = _desc(spec, n, h, w, spec.ci, ld_of(dy))
    wp = packed_weight(spec, weight, PACK_DGRAD)
    check(lib.vq2_conv_dgrad_ex(C.byref(d), VQ2_MASK_AFTER_RESIDUAL if mask_after else 0, _p(dy), _p(wp), _p(mask),
                                ld_of(mask) if mask is not None else 0, _p(residual),
                                ld_of(residual) if residual is not None else 0, _p(out), ld_of(out), _stream()),
          "conv_dgrad")
    return out


def _grad_slot(param):
    """ParamArena: the gradient of `param` lands directly in the flat gradient buffer."""
    slot = getattr(param, "_vq2_grad", None) if param is not None else None
    if slot is not None and param.grad is None:   # a second backward (accumulation) must not overwrite the first
        return slot.view_as(slot)  # fresh tensor object so autograd can adopt it as .grad without a copy
    return None if param is None else torch.empty_like(param, memory_format=torch.contiguous_format)


class WgradBatch:
    """Deferred weight-gradient reduction for a whole backward pass: every conv_wgrad writes only its
    split-K slabs into a persistent per-layer workspace; flush() reduces ALL layers with one launch
    (vq2_wgrad_reduce_batched) straight into the ParamArena gradient slots."""

    def __init__(self):
        self.entries = {}
        self.order = []
        self.tables = {}
        self.dirty = True

    def launch(self, spec, x, dy, relu_in, weight, bias, want_db, side=None):
        n, h, w, _ = x.shape
        key = (id(weight), n, h, w, ld_of(x), ld_of(dy), relu_in)
        d = _desc(spec, n, h, w, ld_of(x), ld_of(dy))
        ent = self.entries.get(key)
        if ent is None:
            nbytes = lib.vq2_conv_wgrad_workspace_bytes(C.byref(d))
            ws = torch.empty(max(nbytes // 4, 4), device=x.device, dtype=torch.float32)
            dw = weight._vq2_grad
            db = bias._vq2_grad if (bias is not None and want_db) else None
            job = WgradJob()
            check(lib.vq2_wgrad_job_init(C.byref(d), _p(ws), _p(dw), _p(db), C.byref(job)), "wgrad_job_init")
            ent = {"ws": ws, "nbytes": nbytes, "job": job, "dw": dw, "db": db, "used": False}
            self.entries[key] = ent
            self.dirty = True
        if side is not None and n * h * w > WGRAD_STREAM_MAX_PIXELS:
            side = None        # a launch that fills the chip by itself gains nothing from a second stream
        if side is not None:   # off the backward critical path: overlaps the next layers' data gradients
            side.wait_event(torch.cuda.current_stream().record_event())
            with torch.cuda.stream(side):
                check(lib.vq2_conv_wgrad_partial(C.byref(d), VQ2_RELU_IN if relu_in else 0, _p(x), _p(dy), _p(ent["db"]),
                                                 _p(ent["ws"]), ent["nbytes"], _stream()), "conv_wgrad_partial")
            x.record_stream(side)
            dy.record_stream(side)
        else:
            check(lib.vq2_conv_wgrad_partial(C.byref(d), VQ2_RELU_IN if relu_in else 0, _p(x), _p(dy), _p(ent["db"]),
                                             _p(ent["ws"]), ent["nbytes"], _stream()), "conv_wgrad_partial")
        if not ent["used"]:
            ent["used"] = True
            self.order.append(key)
        dw, db = ent["dw"], ent["db"]
        return dw.view_as(dw), (None if db is None else db.view_as(db))

    def resblock_w2(self, n, h, w, c, cm, weight, bias):
        """Workspace for the 1x1 weight/bias gradient that vq2_resblock_bwd_data leaves per workgroup; its job
        joins the batched reduction like any other layer's slabs.  Returns (workspace tensor, dw view, db view)."""
        key = (id(weight), n, h, w, "rbw2")
        ent = self.entries.get(key)
        if ent is None:
            nbytes = lib.vq2_resblock_w2_workspace_bytes(n, h, w, c, cm)
            ws = torch.empty(max(nbytes // 4, 4), device=weight.device, dtype=torch.float32)
            dw, db = weight._vq2_grad, bias._vq2_grad
            job = WgradJob()
            check(lib.vq2_resblock_w2_job_init(n, h, w, c, cm, _p(ws), _p(dw), _p(db), C.byref(job)), "resblock_w2_job_init")
            ent = {"ws": ws, "nbytes": nbytes, "job": job, "dw": dw, "db": db, "used": False}
            self.entries[key] = ent
            self.dirty = True
        if not ent["used"]:
            ent["used"] = True
            self.order.append(key)
        return ent["ws"], ent["dw"].view_as(ent["dw"]), ent["db"].view_as(ent["db"])

    def flush(self):
        if not self.order:
            return
        sig = tuple(self.order)
        if self.dirty:
            self.tables = {}
            self.dirty = False
        tab = self.tables.get(sig)
        if tab is None:   # job table of this subset of layers: built and uploaded once, then reused every step
            import numpy as np
            arr = (WgradJob * len(self.order))()
            off = 0
            for i, key in enumerate(self.order):
                job = self.entries[key]["job"]
                job.unit_offset = off
                arr[i] = job
                off += job.n_units_w + job.n_units_b
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            tab = (torch.from_numpy(raw).to(self.entries[self.order[0]]["ws"].device), off)
            self.tables[sig] = tab
        check(lib.vq2_wgrad_reduce_batched(_p(tab[0]), len(self.order), tab[1], _stream()), "wgrad_reduce_batched")
        for key in self.order:
            self.entries[key]["used"] = False
        self.order = []


class StepContext:
    """What one Stage1Trainer's backward pass shares between its layers: the deferred split-K reduction
    (WgradBatch) and the optional side stream for weight gradients.  Hung on the trainer's own parameters as
    `_vq2_ctx`; `active` only while that trainer's backward runs (a stand-alone .backward() on the same model
    takes the immediate per-layer path).  Weight gradients are off the backward critical path (only the optimizer
    consumes them), so on a side stream they overlap the next layers' data-gradient launches; only legal when
    they land in arena slots that nobody reads before the trainer joins the streams."""

    def __init__(self, wgrad_stream=None):
        self.batch = WgradBatch()
        self.stream = wgrad_stream
        self.active = False


def _step_ctx(weight):
    ctx = getattr(weight, "_vq2_ctx", None)
    return ctx if (ctx is not None and ctx.active) else None


# Only launches too small to fill the chip go to the side stream (measured on MI355X, batch 32: the 32x32-resolution
# layers, <= 32,768 pixels: 7.19 -> 7.13 ms per step; every weight gradient there: 7.34 -- two chip-filling kernels
# of different streams do not share CUs usefully)
WGRAD_STREAM_MAX_PIXELS = int(os.environ.get("VQ2_WGRAD_STREAM_MAXPIX", "40000"))


def conv_wgrad(spec, x, dy, relu_in, weight, bias=None, want_dw=True, want_db=True):
    """(dw, db) in the reference layouts; the bias gradient is fused into the same launches."""
    n, h, w, _ = x.shape
    d = _desc(spec, n, h, w, ld_of(x), ld_of(dy))
    nbytes = lib.vq2_conv_wgrad_workspace_bytes(C.byref(d))
    sc = _step_ctx(weight)
    if sc is not None and getattr(weight, "_vq2_grad", None) is not None and weight.grad is None and \
            (bias is None or (getattr(bias, "_vq2_grad", None) is not None and bias.grad is None)) and \
            (bias is None or bias.numel() == spec.cout):
        dw, db = sc.batch.launch(spec, x, dy, relu_in, weight, bias, want_db, sc.stream)
        return (dw if want_dw else None), db
    dw = _grad_slot(weight)
    db = _grad_slot(bias) if (bias is not None and want_db) else None
    side = sc.stream if sc is not None else None
    if side is not None and getattr(weight, "_vq2_grad", None) is not None and dw.data_ptr() == weight._vq2_grad.data_ptr():
        side.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(side):
            ws = torch.empty(max(nbytes // 4, 4), device=x.device, dtype=torch.float32)
            check(lib.vq2_conv_wgrad(C.byref(d), VQ2_RELU_IN if relu_in else 0, _p(x), _p(dy), _p(dw), _p(db), _p(ws),
                                     nbytes, _stream()), "conv_wgrad")
        x.record_stream(side)   # keep the caching allocator from recycling these while the side stream reads
        dy.record_stream(side)
        return (dw if want_dw else None), db
    ws = torch.empty(max(nbytes // 4, 4), device=x.device, dtype=torch.float32)
    check(lib.vq2_conv_wgrad(C.byref(d), VQ2_RELU_IN if relu_in else 0, _p(x), _p(dy), _p(dw), _p(db), _p(ws), nbytes,
                             _stream()), "conv_wgrad")
    return (dw if want_dw else None), db


def relu_bwd(dy, y):
    """g = dy * (y > 0) for dense NHWC operands of any pixel stride."""
    n, h, w, c = dy.shape
    g = torch.empty((n, h, w, c), device=dy.device, dtype=torch.float32)
    check(lib.vq2_relu_bwd(_p(dy), ld_of(dy), _p(y), ld_of(y), _p(g), c, n * h * w, c, _stream()), "relu_bwd")
    return g


def add_(a, b, alpha=1.0):
    """a + alpha*b into a fresh packed tensor (own kernel; used for autograd fan-out sums)."""
    a, b = packed(a), packed(b)
    out = torch.empty_like(a)
    check(lib.vq2_axpby(_p(a), _p(b), float(alpha), _p(out), a.numel(), _stream()), "axpby")
    return out


# ----------------------------------------------------------------------------- layout boundary
class NchwToNhwc(Function):
    """[N,C,H,W] (any strides) -> packed NHWC [N,H,W,ceil4(C)] (zero padded)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x, "input")
        n, c, h, w = x.shape
        ctx.c = c
        xc = x if x.is_contiguous() else x.contiguous()
        out = torch.empty((n, h, w, ceil4(c)), device=x.device, dtype=torch.float32)
        check(lib.vq2_nchw_to_nhwc(_p(xc), _p(out), n, c, h, w, ceil4(c), _stream()), "nchw_to_nhwc")
        return out

    @staticmethod
    def backward(ctx, g):
        g = as_nhwc(g)
        n, h, w, cp = g.shape
        out = torch.empty((n, ctx.c, h, w), device=g.device, dtype=torch.float32)
        check(lib.vq2_nhwc_to_nchw(_p(g), _p(out), n, ctx.c, h, w, ld_of(g), _stream()), "nhwc_to_nchw")
        return out


class NhwcToNchw(Function):
    """NHWC [N,H,W,Cp] -> contiguous NCHW [N,C,H,W] dropping padded channels."""

    @staticmethod
    def forward(ctx, x, c):
        n, h, w, cp = x.shape
        ctx.cp = cp
        out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
        check(lib.vq2_nhwc_to_nchw(_p(x), _p(out), n, c, h, w, ld_of(x), _stream()), "nhwc_to_nchw")
        return out

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = g.shape
        gc = g if g.is_contiguous() else g.contiguous()
        out = torch.empty((n, h, w, ctx.cp), device=g.device, dtype=torch.float32)
        check(lib.vq2_nchw_to_nhwc(_p(gc), _p(out), n, c, h, w, ctx.cp, _stream()), "nchw_to_nhwc")
        return out, None


def to_nhwc(x):
    """Module-boundary input: NCHW tensor -> internal NHWC (zero-copy when x is channels-last)."""
    _require_cuda(x, "input")
    if x.dim() != 4:
        raise RuntimeError("expected a 4-D NCHW tensor")
    c = x.shape[1]
    if c % 4 == 0:
        v = x.permute(0, 2, 3, 1)
        if is_nhwc_dense(v):
            return v
    return NchwToNhwc.apply(x)


def from_nhwc(y, c):
    """Internal NHWC -> NCHW-shaped result (zero-copy channels-last view when C % 4 == 0)."""
    if y.shape[3] == c:
        return y.permute(0, 3, 1, 2)
    return NhwcToNchw.apply(y, c)


class ReluFn(Function):
    """Stand-alone ReLU (only used when a ReLU cannot be fused into a neighbouring conv)."""

    @staticmethod
    def forward(ctx, x):
        x = as_nhwc(x)
        y = relu_bwd(x, x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return relu_bwd(as_nhwc(g), y)


# ----------------------------------------------------------------------------- fused conv op
class GradStash:
    """Side channel for a tensor that is consumed twice (enc_b -> enc_t and the concat, vqvae.py:225,233):
    the consumer that back-propagates FIRST (CatViewFn) parks its gradient here instead of returning it,
    the one that runs LAST (the first conv of enc_t) adds it in its dgrad epilogue -- the sum autograd would
    form with a separate add kernel (+ a slice copy) happens inside a launch that exists anyway.  Either
    order is correct: a stash that was already taken refuses the put and the gradient flows normally."""

    def __init__(self, strict=False):
        self.tensor = None
        self.closed = False
        self.strict = strict   # the taker also applies a ReLU mask to the sum: falling back would be wrong, so raise

    def put(self, t):
        if self.closed:
            return False
        self.tensor = t
        return True

    def take(self):
        t, self.tensor, self.closed = self.tensor, None, True
        return t


class ConvFn(Function):
    """y = [relu]( conv|convT([relu] x) + b [+ residual] ), NHWC."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, spec, flags, out, grad_stash=None, mask_input=False, premasked=False):
        """mask_input: x is the output of a fused trailing ReLU and this op is the one that applies that ReLU's
        backward mask (to its complete input gradient); premasked: the consumers of y do the same for this op's
        own VQ2_RELU_OUT, so the incoming gradient is already masked."""
        ctx.grad_stash, ctx.mask_input, ctx.premasked = grad_stash, mask_input, premasked
        x = as_nhwc(x)
        if residual is not None:
            residual = as_nhwc(residual)
        y = conv_forward(spec, x, weight, bias, flags, residual, out)
        ctx.spec, ctx.flags = spec, flags
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y if (flags & VQ2_RELU_OUT) else None, bias)
        if out is not None:
            y = y.view_as(y)  # fresh tensor object aliasing the caller's buffer
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y, bias = ctx.saved_tensors
        spec, flags = ctx.spec, ctx.flags
        g = as_nhwc(dy)
        if (flags & VQ2_RELU_OUT) and not ctx.premasked:
            g = relu_bwd(g, y)
        relu_in = bool(flags & VQ2_RELU_IN)
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            other = ctx.grad_stash.take() if ctx.grad_stash is not None else None   # gradient of x's second consumer
            dx = conv_dgrad(spec, x.shape, g, weight, mask=x if (relu_in or ctx.mask_input) else None, residual=other,
                            mask_after=ctx.mask_input)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = conv_wgrad(spec, x, g, relu_in, weight, bias, ctx.needs_input_grad[1],
                                ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = g
        return dx, dw, db, dres, None, None, None, None, None, None


def conv_op(x, weight, bias, spec, relu_in=False, relu_out=False, residual=None, out=None, grad_stash=None,
            mask_input=False, premasked=False):
    flags = (VQ2_RELU_IN if relu_in else 0) | (VQ2_RELU_OUT if relu_out else 0)
    if grad_stash is not None and relu_in and not mask_input:
        grad_stash.closed = True      # the epilogue would mask before it adds
        grad_stash = None
    return ConvFn.apply(x, weight, bias, residual, spec, flags, out, grad_stash, mask_input, premasked)


# one-launch ResBlock forward (csrc/vq2_resblock.hip) where the channel counts allow it
RESBLOCK_FUSED = [os.environ.get("VQ2_RB_FUSED", "1") != "0"]


class ResBlockFn(Function):
    """vqvae.py:81-96 as two launches forward and four backward:
        r = relu(conv3x3(relu(x)) + b1)            (ReLU in + ReLU out fused)
        y = [relu]( conv1x1(r) + b2 + x )          (residual, optional trailing ReLU fused)
    backward fuses both ReLU masks and the skip-path add into the dgrad epilogues."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, spec1, spec2, relu_out, out, premasked=False):
        ctx.premasked = premasked
        x = as_nhwc(x)
        n, h, w, c = x.shape
        if RESBLOCK_FUSED[0] and lib.vq2_resblock_supported(c, spec1.co) and spec1.k == 3 and spec2.k == 1:
            r = torch.empty((n, h, w, spec1.co), device=x.device, dtype=torch.float32)
            y = out if out is not None else torch.empty((n, h, w, c), device=x.device, dtype=torch.float32)
            if tuple(y.shape) != (n, h, w, c) or not is_nhwc_dense(y):
                raise RuntimeError("ResBlock: bad `out` buffer")
            check(lib.vq2_resblock_fwd(n, h, w, c, spec1.co, VQ2_RELU_OUT if relu_out else 0, _p(x), ld_of(x),
                                       _p(packed_weight(spec1, w1, PACK_FWD)), _p(b1),
                                       _p(packed_weight(spec2, w2, PACK_FWD)), _p(b2), _p(r), ld_of(r), _p(y), ld_of(y),
                                       _stream()), "resblock_fwd")
        else:
            r = conv_forward(spec1, x, w1, b1, VQ2_RELU_IN | VQ2_RELU_OUT)
            y = conv_forward(spec2, r, w2, b2, VQ2_RELU_OUT if relu_out else 0, residual=x, out=out)
        ctx.spec1, ctx.spec2, ctx.relu_out = spec1, spec2, relu_out
        ctx.save_for_backward(x, r, w1, w2, y if relu_out else None, b1, b2)
        if out is not None:
            y = y.view_as(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, r, w1, w2, y, b1, b2 = ctx.saved_tensors
        s1, s2 = ctx.spec1, ctx.spec2
        g = as_nhwc(dy)
        if ctx.relu_out and not ctx.premasked:
            g = relu_bwd(g, y)
        n, h, w, c = x.shape
        if RESBLOCK_FUSED[0] and ctx.needs_input_grad[0] and lib.vq2_resblock_supported(c, s1.co) and s1.k == 3 and s2.k == 1:
            # both data gradients in one launch (dh is recomputed on each tile's halo and kept in LDS)
            dh = torch.empty((n, h, w, s1.co), device=x.device, dtype=torch.float32)
            dx = torch.empty((n, h, w, c), device=x.device, dtype=torch.float32)
            # with the deferred reduction active the same launch also produces the 1x1 conv's weight/bias gradient
            # partials (it holds g and r anyway): no separate wgrad launch for conv[3]
            sc = _step_ctx(w2)
            w2_ws = None
            if (sc is not None and ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and
                    getattr(w2, "_vq2_grad", None) is not None and getattr(b2, "_vq2_grad", None) is not None and
                    w2.grad is None and b2.grad is None):
                w2_ws, dw2, db2 = sc.batch.resblock_w2(n, h, w, c, s1.co, w2, b2)
            check(lib.vq2_resblock_bwd_data(n, h, w, c, s1.co, _p(g), ld_of(g), _p(r), ld_of(r), _p(x), ld_of(x),
                                            _p(packed_weight(s2, w2, PACK_DGRAD)), _p(packed_weight(s1, w1, PACK_DGRAD)),
                                            _p(dh), ld_of(dh), _p(dx), ld_of(dx), _p(w2_ws), _stream()), "resblock_bwd_data")
            if w2_ws is not None:
                dw1, db1 = conv_wgrad(s1, x, dh, True, w1, b1, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
                return dx, dw1, db1, dw2, db2, None, None, None, None, None
        else:
            # through conv1x1 and the inner ReLU (mask r > 0)
            dh = conv_dgrad(s2, r.shape, g, w2, mask=r)
            # through conv3x3 and the outer ReLU (mask x > 0), plus the skip gradient
            dx = conv_dgrad(s1, x.shape, dh, w1, mask=x, residual=g) if ctx.needs_input_grad[0] else None
        dw2, db2 = conv_wgrad(s2, r, g, False, w2, b2, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        dw1, db1 = conv_wgrad(s1, x, dh, True, w1, b1, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw1, db1, dw2, db2, None, None, None, None, None


class CatViewFn(Function):
    """torch.cat([a, b], channel) where a and b were *produced into* adjacent channel slices
    of `buf` (vqvae.py:218,233): forward is free, backward hands out slices of the gradient."""

    @staticmethod
    def forward(ctx, a, b, buf, stash=None):
        ctx.ca = a.shape[3]
        ctx.stash = stash
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        g = as_nhwc(g)
        gb = g[..., ctx.ca:]
        if ctx.stash is not None:
            if ctx.stash.put(gb):
                gb = None      # b's other consumer adds it inside its own dgrad launch (GradStash)
            elif ctx.stash.strict:
                raise RuntimeError("GradStash: the consumer that applies b's ReLU mask ran before this gradient arrived")
        return g[..., :ctx.ca], gb, None, None


class CatFn(Function):
    """torch.cat([a, b], channel) of two NHWC tensors that already exist (vqvae_deep.py:276,298: the operands are
    handed in by the caller, so they cannot be produced into slices of one buffer like CatViewFn's): two
    slice-copy launches forward, slices of the gradient backward."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = as_nhwc(a), as_nhwc(b)
        n, h, w, ca = a.shape
        cb = b.shape[3]
        if tuple(b.shape[:3]) != (n, h, w):
            raise RuntimeError("CatFn: spatial sizes differ")
        out = torch.empty((n, h, w, ca + cb), device=a.device, dtype=torch.float32)
        pix = n * h * w
        check(lib.vq2_slice_copy(_p(a), ld_of(a), _p(out), ca + cb, pix, ca, 0, _stream()), "slice_copy")
        check(lib.vq2_slice_copy(_p(b), ld_of(b), _p(out[..., ca:]), ca + cb, pix, cb, 0, _stream()), "slice_copy")
        ctx.ca = ca
        return out

    @staticmethod
    def backward(ctx, g):
        g = as_nhwc(g)
        return g[..., :ctx.ca], g[..., ctx.ca:]


class AdaINFn(Function):
    """vqvae_deep.py:99-109 (+ the F.relu_ that follows it at :129,131 when relu=True):
    y = [relu]((1 + gamma) * instance_norm(x) + beta) with (gamma | beta) = h [N, 2C] = fc(style)."""

    @staticmethod
    def forward(ctx, x, h, relu, eps):
        x = as_nhwc(x)
        n, hh, w, c = x.shape
        _require_cuda(h, "AdaIN style projection")
        h2 = h.reshape(n, 2 * c)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        mean = torch.empty((n, c), device=x.device, dtype=torch.float32)
        rstd = torch.empty((n, c), device=x.device, dtype=torch.float32)
        y = torch.empty((n, hh, w, c), device=x.device, dtype=torch.float32)
        check(lib.vq2_instnorm_stats(_p(x), ld_of(x), n, hh * w, c, float(eps), _p(mean), _p(rstd), _stream()), "instnorm_stats")
        check(lib.vq2_adain_fwd(_p(x), ld_of(x), _p(mean), _p(rstd), _p(h2), n, hh * w, c, VQ2_RELU_OUT if relu else 0,
                                _p(y), c, _stream()), "adain_fwd")
        ctx.save_for_backward(x, mean, rstd, h2, y if relu else None)
        ctx.h_shape = h.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, h2, y = ctx.saved_tensors
        n, hh, w, c = x.shape
        g = as_nhwc(dy)
        dh = torch.empty((n, 2 * c), device=x.device, dtype=torch.float32)
        dx = torch.empty((n, hh, w, c), device=x.device, dtype=torch.float32)
        check(lib.vq2_adain_bwd(_p(g), ld_of(g), _p(y), c, _p(x), ld_of(x), _p(mean), _p(rstd), _p(h2), n, hh * w, c,
                                _p(dh), _p(dx), c, _stream()), "adain_bwd")
        return dx, dh.reshape(ctx.h_shape), None, None


class FanOutFn(Function):
    """One tensor consumed twice (enc_b -> enc_t and the concat, vqvae.py:225,233; quant_t ->
    dec_t and upsample_t, vqvae.py:232,217).  Forward: two aliases; backward: one add kernel."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None:
            return g2
        if g2 is None:
            return g1
        return add_(as_nhwc(g1), as_nhwc(g2))


class AddScalarsFn(Function):
    """diff_t + diff_b (vqvae.py:240) on the device without an ATen launch."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        check(lib.vq2_axpby(_p(a), _p(b), 1.0, _p(out), 1, _stream()), "axpby")
        return out

    @staticmethod
    def backward(ctx, g):
        return g.reshape(()), g.reshape(())


# ----------------------------------------------------------------------------- Quantize
def vq_prepare(embed):
    d, k = embed.shape
    embed_t = torch.empty((k, d), device=embed.device, dtype=torch.float32)
    enorm = torch.empty(k, device=embed.device, dtype=torch.float32)
    check(lib.vq2_vq_prepare(_p(embed), _p(embed_t), _p(enorm), d, k, _stream()), "vq_prepare")
    return embed_t, enorm


class QuantizeFn(Function):
    """vqvae.py:42-75 minus the EMA update (done by the module after the all-reduce).
    Returns (ste_out [N,H,W,D], diff 0-dim, idx int64 [N,H,W], stats) where stats is the flat
    fp32 buffer [counts (K) | sumsT (K*D)] of vqvae.py:55-56 (None in eval mode)."""

    @staticmethod
    def forward(ctx, x, embed, want_stats, stats_buf, out_buf, prep=None, stats_stream=None):
        """prep: (embedT, enorm) of `embed` if the caller already holds them (Quantize caches what
        vq2_vq_ema_update_prepare leaves), else they are computed here.
        stats_stream: side stream for the EMA statistics when their consumer is far away (Stage1Trainer applies the
        EMA update after backward): five small latency-bound launches that then run beside the matrix-bound
        forward / backward kernels instead of between them.  The caller joins the stream before it reads stats_buf."""
        x = as_nhwc(x)
        n, h, w, d = x.shape
        k = embed.shape[1]
        m = n * h * w
        embed_t, enorm = prep if prep is not None else vq_prepare(embed)
        idx = torch.empty((n, h, w), device=x.device, dtype=torch.int64)
        if out_buf is None:
            out = torch.empty((n, h, w, d), device=x.device, dtype=torch.float32)
        else:
            if tuple(out_buf.shape) != (n, h, w, d) or not is_nhwc_dense(out_buf):
                raise RuntimeError("QuantizeFn: bad output buffer")
            out = out_buf.view_as(out_buf)
        part = torch.empty(lib.vq2_vq_fwd_workspace_floats(m, d, k), device=x.device, dtype=torch.float32)
        check(lib.vq2_vq_fwd(_p(x), ld_of(x), _p(embed), _p(embed_t), _p(enorm), m, d, k, _p(idx), _p(out), ld_of(out),
                             _p(part), _stream()), "vq_fwd")
        stats = None
        if want_stats:
            if stats_buf is not None:
                if stats_buf.numel() != k + k * d or not stats_buf.is_contiguous():
                    raise RuntimeError("QuantizeFn: stats buffer must be contiguous with K + K*D floats")
                stats = stats_buf.view_as(stats_buf)
            else:
                stats = torch.empty(k + k * d, device=x.device, dtype=torch.float32)
            # every element of stats is written (deterministic sort + ordered sums, no atomics, no zeroing)
            nbytes = lib.vq2_vq_stats_workspace_bytes(m, d, k)
            ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.int32)
            if stats_stream is not None and stats_buf is not None:
                stats_stream.wait_event(torch.cuda.current_stream().record_event())
                with torch.cuda.stream(stats_stream):
                    check(lib.vq2_vq_stats(_p(x), ld_of(x), _p(idx), m, d, k, _p(stats[:k]), _p(stats[k:]), _p(ws), nbytes,
                                           _stream()), "vq_stats")
                for tns in (x, idx, ws):
                    tns.record_stream(stats_stream)
            else:
                check(lib.vq2_vq_stats(_p(x), ld_of(x), _p(idx), m, d, k, _p(stats[:k]), _p(stats[k:]), _p(ws), nbytes,
                                       _stream()), "vq_stats")
        diff = torch.empty((), device=x.device, dtype=torch.float32)
        check(lib.vq2_vq_loss(_p(part), m, d, _p(diff), _stream()), "vq_loss")
        ctx.save_for_backward(x, idx, embed_t)
        ctx.k = k
        ctx.set_materialize_grads(False)   # no zero-filled gradients for idx / stats (two fill launches per call)
        ctx.mark_non_differentiable(idx)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return out, diff, idx, stats

    @staticmethod
    def backward(ctx, g_out, g_diff, _gi, _gs):
        x, idx, embed_t = ctx.saved_tensors
        n, h, w, d = x.shape
        if g_out is None and g_diff is None:
            return None, None, None, None, None, None, None
        if g_out is not None:
            g_out = as_nhwc(g_out)
        if g_diff is not None and not g_diff.is_contiguous():
            g_diff = g_diff.contiguous()
        dx = torch.empty((n, h, w, d), device=x.device, dtype=torch.float32)
        check(lib.vq2_vq_bwd(_p(g_out), ld_of(g_out) if g_out is not None else d, _p(g_diff), _p(x), ld_of(x),
                             _p(idx), _p(embed_t), n * h * w, d, ctx.k, _p(dx), d, _stream()), "vq_bwd")
        return dx, None, None, None, None, None, None


def vq_ema_update(embed, cluster_size, embed_avg, stats, decay, eps, prep_out=None):
    """prep_out: (embedT [K,D], enorm [K]) buffers that receive the prepared form of the UPDATED codebook (same launch)."""
    d, k = embed.shape
    counts, sums_t = stats[:k], stats[k:]
    scratch = torch.empty(4, device=embed.device, dtype=torch.float32)
    if prep_out is not None and 256 % d == 0:
        check(lib.vq2_vq_ema_update_prepare(_p(embed), _p(cluster_size), _p(embed_avg), _p(counts), _p(sums_t), d, k,
                                            float(decay), float(eps), _p(scratch), _p(prep_out[0]), _p(prep_out[1]),
                                            _stream()), "vq_ema_update_prepare")
        return True
    check(lib.vq2_vq_ema_update(_p(embed), _p(cluster_size), _p(embed_avg), _p(counts), _p(sums_t), d, k,
                                float(decay), float(eps), _p(scratch), _stream()), "vq_ema_update")
    return False


def vq_gather(idx, embed, prep=None):
    """embed_code (vqvae.py:77-78): idx [...] int64 -> [..., D]."""
    d, k = embed.shape
    embed_t, _ = prep if prep is not None else vq_prepare(embed)
    idx_c = idx.contiguous()
    m = idx_c.numel()
    out = torch.empty((*idx.shape, d), device=embed.device, dtype=torch.float32)
    check(lib.vq2_vq_gather(_p(idx_c), _p(embed_t), m, d, k, _p(out), d, _stream()), "vq_gather")
    return out


# ----------------------------------------------------------------------------- loss
def _scale_by(src, scalar, alpha=1.0):
    """src * scalar[0] * alpha with the scalar read on the device (no host sync)."""
    out = torch.empty_like(src)
    sc = scalar.reshape(1) if scalar.is_contiguous() else scalar.contiguous().reshape(1)
    check(lib.vq2_scale(_p(src), _p(sc), float(alpha), _p(out), src.numel(), _stream()), "scale")
    return out


class MseLossFn(Function):
    """nn.MSELoss() (train_vqvae.py:31,83); the gradient 2(a-b)/n is produced in the same pass."""

    @staticmethod
    def forward(ctx, a, b):
        _require_cuda(a, "mse input")
        ac = a if a.is_contiguous() else a.contiguous()
        bc = b if b.is_contiguous() else b.contiguous()
        n = ac.numel()
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        grad = torch.empty_like(ac) if ctx.needs_input_grad[0] else None
        ws = torch.empty(lib.vq2_mse_workspace_bytes(n) // 4, device=a.device, dtype=torch.float32)
        check(lib.vq2_mse_fwd_bwd(_p(ac), _p(bc), n, n, None, _p(loss), _p(grad), _p(ws), ws.numel() * 4, _stream()),
              "mse_fwd_bwd")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (None if grad is None else _scale_by(grad, g)), None


def mse_loss(a, b):
    return MseLossFn.apply(a, b)


def stage1_loss_and_seeds(dec, diff, img, weight, denom):
    """Trainer fast path of Stage1LossFn: the loss values AND the gradients of (dec, diff) for a backward
    seed of exactly 1 (what loss.backward() means), so the pass can start with
    torch.autograd.backward((dec, diff), seeds) -- no ones_like fill, no gradient re-scaling launches.
    Returns (loss, recon, latent, d_dec, d_diff)."""
    dc, ic = dec.detach(), img.detach()
    if not (dc.is_contiguous() and ic.is_contiguous()) or diff.numel() != 1:
        raise RuntimeError("stage1_loss_and_seeds: contiguous dec/img and the [1]-shaped latent loss expected")
    n = dc.numel()
    recon = torch.empty((), device=dc.device, dtype=torch.float32)
    loss = torch.empty((), device=dc.device, dtype=torch.float32)
    grad = torch.empty_like(dc)
    latent = diff.detach().reshape(())
    ws = torch.empty(lib.vq2_mse_workspace_bytes(n) // 4, device=dc.device, dtype=torch.float32)
    check(lib.vq2_stage1_loss(_p(dc), _p(ic), n, int(denom), _p(latent), float(weight), _p(recon), _p(loss), _p(grad),
                              _p(ws), ws.numel() * 4, _stream()), "stage1_loss")
    key = (dc.device, float(weight), tuple(diff.shape))
    seed = _DIFF_SEEDS.get(key)
    if seed is None:
        seed = _DIFF_SEEDS[key] = torch.full(tuple(diff.shape), float(weight), device=dc.device, dtype=torch.float32)
    return loss, recon, latent, grad, seed


_DIFF_SEEDS = {}


class Stage1LossFn(Function):
    """loss = MSE(dec, img) + 0.25 * diff.mean()  (train_vqvae.py:83-85) -> (loss, recon, latent)."""

    @staticmethod
    def forward(ctx, dec, diff, img, weight, denom=None):
        """`denom`: number of real elements when dec/img carry zero padding (NHWC4 image layout)."""
        _require_cuda(dec, "dec")
        dc = dec if dec.is_contiguous() else dec.contiguous()
        ic = img if img.is_contiguous() else img.contiguous()
        n = dc.numel()
        denom = n if denom is None else int(denom)
        recon = torch.empty((), device=dec.device, dtype=torch.float32)
        grad = torch.empty_like(dc) if ctx.needs_input_grad[0] else None
        ws = torch.empty(lib.vq2_mse_workspace_bytes(n) // 4, device=dec.device, dtype=torch.float32)
        check(lib.vq2_mse_fwd_bwd(_p(dc), _p(ic), n, denom, None, _p(recon), _p(grad), _p(ws), ws.numel() * 4,
                                  _stream()), "mse_fwd_bwd")
        if diff.numel() != 1:
            raise RuntimeError("stage1 loss expects the [1]-shaped latent loss of VQVAE.forward")
        latent = diff.detach().reshape(())  # mean of a single element
        loss = torch.empty((), device=dec.device, dtype=torch.float32)
        check(lib.vq2_axpby(_p(recon), _p(latent.contiguous()), float(weight), _p(loss), 1, _stream()), "axpby")
        ctx.save_for_backward(grad)
        ctx.weight = weight
        ctx.diff_shape = diff.shape
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(recon, latent)
        return loss, recon, latent

    @staticmethod
    def backward(ctx, g, _gr, _gl):
        (grad,) = ctx.saved_tensors
        if g is None:
            return None, None, None, None, None
        d_dec = None if grad is None else _scale_by(grad, g)
        d_diff = _scale_by(torch.ones(ctx.diff_shape, device=g.device), g, ctx.weight) if ctx.needs_input_grad[1] else None
        return d_dec, d_diff, None, None, None
