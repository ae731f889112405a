// Fused ResBlock kernels with the 3x3 conv in the Winograd domain (vq2_rbwino.hip); dispatched by vq2_resblock.hip.
#pragma once
#include "vq2_common.h"

namespace vq2 {

struct RbwFwdParams {
    const float *x;    // [N,H,W,ldx]
    const float *w1;   // VQ2_PACK_FWD panel of the 3x3 weight: [32][9*128]
    const float *b1;   // [32]
    const float *w2;   // VQ2_PACK_FWD panel of the 1x1 weight: [128][32]
    const float *b2;   // [128]
    float *r;          // [N,H,W,ldr]
    float *y;          // [N,H,W,ldy]
    int N, H, W, ldx, ldr, ldy;
    int relu_out;
};

bool rbw_fwd_ok(const RbwFwdParams &P);
int launch_rbw_fwd(const RbwFwdParams &P, hipStream_t s);

}  // namespace vq2
