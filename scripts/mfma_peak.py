"""Measure the sustained v_mfma_f32_32x32x2_f32 rate of this GPU (register-resident loop)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqvae2_amd  # noqa: E402
lib = vqvae2_amd._lib.lib
buf = torch.zeros(16, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for blocks in (256, 512, 1024, 2048):
    iters = 4000
    # v_mfma_f32_16x16x4_f32: 64 MFMAs per iteration and wave, 2*16*16*4 FLOP each
    lib.vq2_debug_mfma_peak16(C.c_void_p(buf.data_ptr()), blocks, 200, s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        lib.vq2_debug_mfma_peak16(C.c_void_p(buf.data_ptr()), blocks, iters, s)
    b.record()
    torch.cuda.synchronize()
    ms16 = a.elapsed_time(b) / 5
    print(f"blocks={blocks:5d} 16x16x4: {ms16:8.3f} ms  {blocks * 4 * iters * 64 * (2.0 * 16 * 16 * 4) / ms16 / 1e9:7.1f} TFLOP/s", flush=True)
    lib.vq2_debug_mfma_peak(C.c_void_p(buf.data_ptr()), blocks, 200, s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        lib.vq2_debug_mfma_peak(C.c_void_p(buf.data_ptr()), blocks, iters, s)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    flop = blocks * 4 * iters * 32 * (2.0 * 32 * 32 * 2)
    print(f"blocks={blocks:5d} waves/SIMD={blocks*4/1024:4.1f}  {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s", flush=True)
