// Launch parameters shared by the conv kernels of libvq2 (vq2_conv.hip, vq2_wino.hip).
#pragma once
#include "vq2_common.h"

namespace vq2 {

struct ConvGemmParams {
    const float *x;   // [N,H,W,ldx]
    const float *w;   // [phases][Co][K]   (K = KH*KW*Ci, ci fastest)
    const float *bias;  // [Co] or null
    const float *mask;  // shape of y (pixel stride ldm) or null: y *= (mask > 0)
    const float *res;   // shape of y (pixel stride ldr) or null: y += res
    float *y;           // [N,Hy,Wy,ldy]
    int N, H, W, Ci, ldx;
    int Ho, Wo, Co, ldy;  // virtual output grid (rows of the GEMM) and channels
    int KH, KW, stride, pad_h, pad_w;
    int K, M;             // K = KH*KW*Ci, M = N*Ho*Wo
    int phases;           // 1, or 4 for the sub-pixel transposed conv
    int Hy, Wy;           // real output image size
    int ldm, ldr;
    int relu_in, relu_out;
    int mask_after;       // apply the mask to (acc + residual) instead of to acc alone
    int nbias;            // bias has nbias entries (real output channels)
    int c4_tpw;           // conv_k4s2_c4_kernel: tiles per workgroup
    int ci_real;          // real input channels (<= Ci; the rest are zero padding), 0 = Ci
    double flops, bytes;  // algorithmic work of this launch (for the profiler only)
    unsigned long long *stamps;  // diagnostic build only (STAMP): per-phase cycle totals of workgroup 0
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = 0x7FFFFFF0;  // >= num_records of every descriptor: loads return 0, stores are dropped
constexpr unsigned RSRC_FLAGS = 0x00020000;

__device__ __forceinline__ float4 as_f4(u32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// vq2_wino.hip: 3x3 stride-1 pad-1 convolutions as F(2,3) Winograd along the image rows
bool wino3_ok(const ConvGemmParams &P);
int launch_wino3(const ConvGemmParams &P, hipStream_t s);

}  // namespace vq2
