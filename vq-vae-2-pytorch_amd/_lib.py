"""ctypes binding of libvq2.so (include/vq2.h).  No fallback: if the HIP library is
missing the import fails loudly -- the product path never routes around it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VQ2_LIB: an alternative build of the SAME sources (kernel A/B experiments, scripts/build_variant.sh)
LIB_PATH = os.environ.get("VQ2_LIB") or os.path.join(_HERE, "libvq2.so")

c_f32p = C.c_void_p
c_stream = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "H", "W", "Ci", "Co", "KH", "KW", "stride", "pad", "transposed", "ldx", "ldy", "Cir", "Cor")]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("offset", C.c_int64), ("numel", C.c_int64)] + \
               [(n, C.c_int32) for n in ("Or", "Ir", "Op", "Ip", "KH", "KW", "mode", "reserved")]


class WgradJob(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dw", C.c_void_p), ("bias_ws", C.c_void_p), ("db", C.c_void_p),
                ("unit_offset", C.c_int64)] + \
               [(n, C.c_int32) for n in ("O", "I", "Or", "Ir", "taps", "S", "n_units_w", "n_units_b", "swapped", "bias_splits")]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(vq-vae-2-pytorch_amd/csrc/build.sh).  There is no CPU or eager fallback for this path.")
    import torch  # noqa: F401  (loads torch's libamdhip64 first so both share one HIP runtime)
    lib = C.CDLL(LIB_PATH)
    P, I32, I64, SZ, F, D = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float, C.c_double
    DP = C.POINTER(ConvDesc)
    sig = {
        "vq2_version": (C.c_int, []),
        "vq2_last_error": (C.c_char_p, []),
        "vq2_prof_enable": (C.c_int, [C.c_int]),
        "vq2_prof_report": (C.c_int, [C.c_char_p, SZ]),
        "vq2_pack_weight": (C.c_int, [DP, C.c_int, P, P, P]),
        "vq2_pack_job_init": (C.c_int, [DP, C.c_int, P, P, C.POINTER(PackJob)]),
        "vq2_pack_weights_batched": (C.c_int, [P, I32, I64, P]),
        "vq2_conv_fwd": (C.c_int, [DP, C.c_int, P, P, P, P, I32, P, P]),
        "vq2_conv_dgrad": (C.c_int, [DP, P, P, P, I32, P, I32, P, I32, P]),
        "vq2_conv_dgrad_ex": (C.c_int, [DP, C.c_int, P, P, P, I32, P, I32, P, I32, P]),
        "vq2_conv_wgrad_workspace_bytes": (SZ, [DP]),
        "vq2_conv_wgrad": (C.c_int, [DP, C.c_int, P, P, P, P, P, SZ, P]),
        "vq2_conv_wgrad_partial": (C.c_int, [DP, C.c_int, P, P, P, P, SZ, P]),
        "vq2_wgrad_job_init": (C.c_int, [DP, P, P, P, C.POINTER(WgradJob)]),
        "vq2_wgrad_reduce_batched": (C.c_int, [P, I32, I64, P]),
        "vq2_colsum_workspace_bytes": (SZ, [I64, I32]),
        "vq2_colsum": (C.c_int, [P, I64, I32, I32, P, P, SZ, P]),
        "vq2_nchw_to_nhwc": (C.c_int, [P, P, I32, I32, I32, I32, I32, P]),
        "vq2_nhwc_to_nchw": (C.c_int, [P, P, I32, I32, I32, I32, I32, P]),
        "vq2_relu_bwd": (C.c_int, [P, I32, P, I32, P, I32, I64, I32, P]),
        "vq2_resblock_supported": (C.c_int, [I32, I32]),
        "vq2_resblock_bwd_data": (C.c_int, [I32, I32, I32, I32, I32, P, I32, P, I32, P, I32, P, P, P, I32, P, I32, P, P]),
        "vq2_resblock_w2_workspace_bytes": (SZ, [I32, I32, I32, I32, I32]),
        "vq2_resblock_w2_job_init": (C.c_int, [I32, I32, I32, I32, I32, P, P, P, C.POINTER(WgradJob)]),
        "vq2_resblock_fwd": (C.c_int, [I32, I32, I32, I32, I32, C.c_int, P, I32, P, P, P, P, P, I32, P, I32, P]),
        "vq2_slice_copy": (C.c_int, [P, I32, P, I32, I64, I32, C.c_int, P]),
        "vq2_vq_prepare": (C.c_int, [P, P, P, I32, I32, P]),
        "vq2_vq_fwd_workspace_floats": (SZ, [I64, I32, I32]),
        "vq2_vq_fwd": (C.c_int, [P, I32, P, P, P, I64, I32, I32, P, P, I32, P, P]),
        "vq2_vq_stats_workspace_bytes": (SZ, [I64, I32, I32]),
        "vq2_vq_stats": (C.c_int, [P, I32, P, I64, I32, I32, P, P, P, SZ, P]),
        "vq2_vq_loss": (C.c_int, [P, I64, I32, P, P]),
        "vq2_vq_bwd": (C.c_int, [P, I32, P, P, I32, P, P, I64, I32, I32, P, I32, P]),
        "vq2_vq_ema_update": (C.c_int, [P, P, P, P, P, I32, I32, D, D, P, P]),
        "vq2_vq_ema_update_prepare": (C.c_int, [P, P, P, P, P, I32, I32, D, D, P, P, P, P]),
        "vq2_vq_gather": (C.c_int, [P, P, I64, I32, I32, P, I32, P]),
        "vq2_instnorm_stats": (C.c_int, [P, I32, I32, I64, I32, D, P, P, P]),
        "vq2_adain_fwd": (C.c_int, [P, I32, P, P, P, I32, I64, I32, C.c_int, P, I32, P]),
        "vq2_adain_bwd": (C.c_int, [P, I32, P, I32, P, I32, P, P, P, I32, I64, I32, P, P, I32, P]),
        "vq2_mse_workspace_bytes": (SZ, [I64]),
        "vq2_mse_fwd_bwd": (C.c_int, [P, P, I64, I64, P, P, P, P, SZ, P]),
        "vq2_stage1_loss": (C.c_int, [P, P, I64, I64, P, F, P, P, P, P, SZ, P]),
        "vq2_adam_step": (C.c_int, [P, P, P, P, I64, D, D, D, D, I32, D, P]),
        "vq2_axpby": (C.c_int, [P, P, F, P, I64, P]),
        "vq2_scale": (C.c_int, [P, P, F, P, I64, P]),
        "vq2_comm_unique_id": (C.c_int, [P]),
        "vq2_comm_init": (C.c_int, [P, I32, I32]),
        "vq2_comm_world": (C.c_int, []),
        "vq2_comm_rank": (C.c_int, []),
        "vq2_comm_allreduce_sum": (C.c_int, [P, I64, P]),
        "vq2_comm_broadcast": (C.c_int, [P, I64, I32, P]),
        "vq2_comm_destroy": (C.c_int, []),
        "vq2_debug_mfma_peak": (C.c_int, [P, I32, I32, P]),
        "vq2_debug_mfma_peak16": (C.c_int, [P, I32, I32, P]),
        "vq2_debug_set_rb_stamps": (C.c_int, [P]),
        "vq2_debug_set_stamps": (C.c_int, [P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib, tuple(sig.keys())


def _header_api_version():
    """VQ2_API_VERSION of include/vq2.h (the header this binding was written against)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "vq2.h")
    with open(hdr) as f:
        return int(re.search(r"#define\s+VQ2_API_VERSION\s+(\d+)", f.read()).group(1))


lib, EXPORTS = _load()
API_VERSION = _header_api_version()
if lib.vq2_version() != API_VERSION:
    raise ImportError(f"{LIB_PATH} implements ABI revision {lib.vq2_version()} but include/vq2.h declares "
                      f"{API_VERSION}: rebuild the library (vq-vae-2-pytorch_amd/csrc/build.sh)")


def check(code, what=""):
    if code != 0:
        msg = lib.vq2_last_error().decode(errors="replace")
        raise RuntimeError(f"libvq2 {what} failed (code {code}): {msg}")
