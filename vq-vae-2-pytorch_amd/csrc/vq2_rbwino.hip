// Fused ResBlock forward (vqvae.py:81-96) with the 3x3 conv in the Winograd domain -- F(2,3) along the image rows, see
// vq2_wino.hip for the algebra -- for images whose rows are whole 64-pixel segments (the 64x64 / 128x128 levels):
//     r = relu(conv3x3(relu(x)) + b1)      stage 1: four GEMMs of depth 3*128 over column PAIRS instead of one of depth 9*128
//     y = [relu](conv1x1(r) + b2 + x)      stage 2: from the LDS-resident r tile, direct form
// 90 % of the block's multiplications are the 3x3's; 2/3 of those remain.
//
// Workgroup = 4 rows x 32 pairs (64 pixels) of one image; wave w owns row w: 32 pairs x 32 middle channels x 4 Winograd
// indices (64 accumulator registers).  Stage 1 walks the 16 eight-channel blocks of x; per block the raw 6 x 66 halo patch
// (ReLU applied) and the transformed taps of ALL THREE kernel rows are staged (one barrier per 48 MFMAs of a wave), loads of
// block c+1 behind the MFMAs of block c, double-buffered.  The output transform leaves r in registers: bias, ReLU, one copy
// to memory (the backward pass reads it), one into the LDS tile [256 px][32] that is stage 2's A operand; the 1x1 panel
// [128][32] and the skip path x (straight into stage 2's accumulators) are fetched meanwhile.  Stage 2: each wave its own
// 64 pixels x 128 channels (128 accumulator registers, 128 MFMAs) on top of x + b2; epilogue = optional ReLU and stores.
// (Measured alternative: 2-row tiles with the Winograd indices split over wave pairs -- 1,024 workgroups, two resident
//  rounds, 24 MFMAs per barrier: 89.5 us against this form's 83.7 and the direct kernel's 90.6.)
#include "vq2_conv.h"
#include "vq2_rbwino.h"

namespace vq2 {
namespace rbw {

constexpr int CC = 128, CM = 32;
constexpr int BK = 8, LDK = BK + 4;
constexpr int TR = 4, TPW = 32, PR = TR + 2, PW = 2 * TPW + 2, NPX = PR * PW;   // 6 x 66 = 396 patch pixels
constexpr int A_FLOATS = (NPX + 1) * LDK;                 // + dump row
constexpr int A_ITEMS = NPX * 2, A_LD = (A_ITEMS + 255) / 256;                  // 792 items, 4 slots
constexpr int B_FLOATS = 3 * 4 * CM * LDK;                // [kh][v][co][LDK]
constexpr int LDR = CM + 4;                               // 144-byte rows: conflict-free ds_read_b128
constexpr int R_FLOATS = TR * 64 * LDR, W2_FLOATS = CC * LDR;
constexpr int S1_FLOATS = 2 * (A_FLOATS + B_FLOATS), S2_FLOATS = R_FLOATS + W2_FLOATS;
constexpr size_t LDS_BYTES = (size_t)(S1_FLOATS > S2_FLOATS ? S1_FLOATS : S2_FLOATS) * sizeof(float);   // 74,976

__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(256, 2) void rbw_fwd_kernel(const RbwFwdParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                    // [2][A_FLOATS]
    float *Bs = smem + 2 * A_FLOATS;     // [2][B_FLOATS]
    float *Rs = smem;                    // stage 2 (aliases the stage-1 buffers): [256][LDR]
    float *W2s = smem + R_FLOATS;        //                                         [128][LDR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tw = P.W / 64, th = P.H / TR;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int wb = vid % tw, hb = (vid / tw) % th, n = vid / (tw * th);
    const int h0 = hb * TR, w0 = wb * 64;

    const int npix = P.N * P.H * P.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, npix * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w1), 0, CM * 9 * CC * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w2), 0, CC * CM * 4, RSRC_FLAGS);

    // ---- stage-1 staging coordinates (unconditional loads: out-of-image / out-of-patch items read 0 into the dump row)
    int a_off[A_LD], a_dst[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int it = tid + 256 * j;
        const bool ok = it < A_ITEMS;
        const int px = ok ? (it >> 1) : 0, q = it & 1;
        const int pr = px / PW, pc = px - pr * PW;
        const int row = h0 - 1 + pr, col = w0 - 1 + pc;
        const bool in = ok && (unsigned)row < (unsigned)P.H && (unsigned)col < (unsigned)P.W;
        a_off[j] = in ? (((n * P.H + row) * P.W + col) * P.ldx + 4 * q) * 4 : (int)0x80000000;
        a_dst[j] = (ok ? px : NPX) * LDK + 4 * q;
    }
    // weight items (kh, co, quad): threads 0..191; the others read out of range and store into a scratch row of their own
    const bool role_b = tid < 3 * CM * 2;
    const int bkh = role_b ? tid / (CM * 2) : 0, bco = (tid % (CM * 2)) >> 1, bq = tid & 1;
    const int b_off = role_b ? (bco * 9 * CC + bkh * 3 * CC + 4 * bq) * 4 : (int)0x80000000;
    const int b_dst = (bkh * 4 * CM + bco) * LDK + 4 * bq;

    // The patch of block c+2 is requested while block c is multiplied and lands in LDS a block later: an activation line
    // that comes from HBM needs ~2 us, a block's MFMAs take 1.3 -- with the patch only one block ahead a workgroup alone on
    // its CU ran at half the pipe's rate (49 us for 24 us of MFMAs) and two resident ones stalled in step.  The taps (L2
    // hits) stay one block ahead.
    u32x4 ra[2][A_LD], rb[3];
    auto load_b = [&](int cb) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) rb[kw] = __builtin_amdgcn_raw_buffer_load_b128(rw1, b_off + (kw * CC + cb) * 4, 0, 0);
    };
    auto load_a = [&](int set, int cb) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) ra[set][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, a_off[j] + cb * 4, 0, 0);
    };
    auto store_b = [&](float *b) {
        if (role_b) {
            const float4 g0 = as_f4(rb[0]), g1 = as_f4(rb[1]), g2 = as_f4(rb[2]);
            const float4 t = add4(g0, g2);
            *reinterpret_cast<float4 *>(b + 0 * CM * LDK + b_dst) = g0;
            *reinterpret_cast<float4 *>(b + 1 * CM * LDK + b_dst) = add4(t, g1);     // (x 1/2 in the output transform)
            *reinterpret_cast<float4 *>(b + 2 * CM * LDK + b_dst) = sub4(t, g1);
            *reinterpret_cast<float4 *>(b + 3 * CM * LDK + b_dst) = g2;
        }
    };
    auto store_a = [&](int set, float *a) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) *reinterpret_cast<float4 *>(a + a_dst[j]) = relu4(as_f4(ra[set][j]));
    };

    f32x16 acc[4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;

    const int frag_row = lane & 31, frag_k = 4 * (lane >> 5);
    const int lane_a = (wave * PW + 2 * frag_row) * LDK + frag_k;       // row `wave` of the tile, pair frag_row
    const int lane_b = frag_row * LDK + frag_k;

    // fragments double-buffered by hand: the eight LDS reads of kernel row kh+1 go out BEFORE the 16 MFMAs of row kh (left
    // to itself the scheduler issues them after, and every row starts with an exposed LDS round trip and VALU -> MFMA hazards)
    auto compute = [&](const float *a, const float *b) {
        float4 d[2][4], fb[2][4];
        auto read_row = [&](int kh, int set) {
            const float *ap = a + lane_a + kh * PW * LDK;
#pragma unroll
            for (int c = 0; c < 4; ++c) d[set][c] = *reinterpret_cast<const float4 *>(ap + c * LDK);
#pragma unroll
            for (int v = 0; v < 4; ++v) fb[set][v] = *reinterpret_cast<const float4 *>(b + (kh * 4 + v) * CM * LDK + lane_b);
        };
        read_row(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                        // row 0's reads
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int cur = kh & 1;
            if (kh + 1 < 3) read_row(kh + 1, cur ^ 1);
            float4 fv[4];
            fv[0] = sub4(d[cur][0], d[cur][2]);
            fv[1] = add4(d[cur][1], d[cur][2]);
            fv[2] = sub4(d[cur][2], d[cur][1]);
            fv[3] = sub4(d[cur][1], d[cur][3]);
#define VQ2_RBW_STEP(C) \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[v].C, fb[cur][v].C, acc[v], 0, 0, 0);
            VQ2_RBW_STEP(x) VQ2_RBW_STEP(y) VQ2_RBW_STEP(z) VQ2_RBW_STEP(w)
#undef VQ2_RBW_STEP
            if (kh + 1 < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // next row's reads
            __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);                   // this row's transform
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                   // this row's MFMAs
        }
    };

    constexpr int NCB = CC / BK;
    load_a(0, 0);
    load_b(0);
    load_a(1, BK);                                   // block 1's patch: stored at the end of block 0
    store_a(0, As);
    store_b(Bs);
    __syncthreads();
    for (int cbi = 0; cbi < NCB; cbi += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {                // block c = cbi + u: patch set u is free, set u ^ 1 holds block c+1
            const int c = cbi + u;
            const int c1 = (c + 1 < NCB ? c + 1 : c) * BK, c2 = (c + 2 < NCB ? c + 2 : c) * BK;   // (the tail re-loads itself)
            load_b(c1);
            load_a(u, c2);
            __builtin_amdgcn_sched_barrier(0);
            compute(As + (c & 1) * A_FLOATS, Bs + (c & 1) * B_FLOATS);
            __builtin_amdgcn_sched_barrier(0);
            store_b(Bs + ((c + 1) & 1) * B_FLOATS);
            store_a(u ^ 1, As + ((c + 1) & 1) * A_FLOATS);
            __syncthreads();
        }
    }

    // ---- 1x1 panel and the skip path on their way while the output transform runs; then r -> memory and the LDS tile
    u32x4 rw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rw[j] = __builtin_amdgcn_raw_buffer_load_b128(rw2, (tid + 256 * j) * 16, 0, 0);
    const int colq = lane & 31, rowq = 4 * (lane >> 5);
    const int pix_row = (n * P.H + h0 + wave) * P.W + w0;          // first pixel of this wave's row
    const int ldx4 = P.ldx * 4, ldy4 = P.ldy * 4;
    f32x16 acc2[2][4];                                              // stage 2 accumulates on top of x (+ b2 below)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = pix_row + i * 32 + rowq + (r & 3) + 8 * (r >> 2);
                acc2[i][j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, pix * ldx4 + (j * 32 + colq) * 4, 0, 0));
            }
    __builtin_amdgcn_sched_barrier(0);      // (keep the loads HERE: the scheduler would sink them to their first use)
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(P.r, 0, npix * P.ldr * 4, RSRC_FLAGS);
    {
        const float bv = P.b1[colq];
        const int ldr4 = P.ldr * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pair = rowq + (r & 3) + 8 * (r >> 2);
            const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
            const float y0 = relu1(m0 + 0.5f * (m1 + m2) + bv);
            const float y1 = relu1(0.5f * (m1 - m2) - m3 + bv);
            const int pl = wave * 64 + 2 * pair;                    // pixel of the tile
            Rs[pl * LDR + colq] = y0;
            Rs[(pl + 1) * LDR + colq] = y1;
            const int off = (pix_row + 2 * pair) * ldr4 + colq * 4;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y0), rr, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y1), rr, off + ldr4, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int it = tid + 256 * j;                               // float4 index into [128][32]: co = it / 8, quad = it % 8
        *reinterpret_cast<float4 *>(W2s + (it >> 3) * LDR + (it & 7) * 4) = as_f4(rw[j]);
    }
    __syncthreads();

    // ---- stage 2: 64 pixels x 128 channels per wave from the r tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float bv = P.b2[j * 32 + colq];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] += bv;
    }
    {
        const float *a = Rs + (wave * 64 + frag_row) * LDR + frag_k;
        const float *b = W2s + frag_row * LDR + frag_k;
#pragma unroll
        for (int k8 = 0; k8 < CM / 8; ++k8) {
            float4 fa[2], fb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float4 *>(a + i * 32 * LDR + k8 * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const float4 *>(b + j * 32 * LDR + k8 * 8);
#define VQ2_RBW_STEP2(C)                                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
        acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].C, fb[j].C, acc2[i][j], 0, 0, 0);
            VQ2_RBW_STEP2(x) VQ2_RBW_STEP2(y) VQ2_RBW_STEP2(z) VQ2_RBW_STEP2(w)
#undef VQ2_RBW_STEP2
        }
    }

    // ---- epilogue: optional ReLU, stores
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, npix * P.ldy * 4, RSRC_FLAGS);
    const int relu_bits = P.relu_out ? 0 : (int)0x80000000;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co4 = (j * 32 + colq) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = pix_row + i * 32 + rowq + (r & 3) + 8 * (r >> 2);
                const float v = relu_floor(acc2[i][j][r], relu_bits);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, pix * ldy4 + co4, 0, 0);
            }
    }
}

}  // namespace rbw

bool rbw_fwd_ok(const RbwFwdParams &P) {
    static const int on = getenv("VQ2_RB_WINO") ? atoi(getenv("VQ2_RB_WINO")) : 1;
    const long gib = 1L << 30;
    const int ldmax = P.ldx > P.ldy ? P.ldx : P.ldy;
    return on && P.W % 64 == 0 && P.H % rbw::TR == 0 && (long)P.N * P.H * P.W * ldmax * 4 < gib && P.ldx % 4 == 0;
}

int launch_rbw_fwd(const RbwFwdParams &P, hipStream_t s) {
    allow_big_lds(rbw::rbw_fwd_kernel, rbw::LDS_BYTES);
    const unsigned grid = (unsigned)(P.N * (P.H / rbw::TR) * (P.W / 64));
    hipLaunchKernelGGL(rbw::rbw_fwd_kernel, dim3(grid), dim3(256), rbw::LDS_BYTES, s, P);
    return check_launch("rbw_fwd_kernel");
}

}  // namespace vq2
