// 3x3 stride-1 pad-1 convolution (forward, and the data gradient, which is the same operation on the flipped
// panel) as Winograd F(2,3) ALONG THE IMAGE ROWS, on the exact-fp32 matrix instruction.
//
// Two neighbouring output pixels of a row (a "column pair" t: columns 2t, 2t+1) share the four input columns
// 2t-1 .. 2t+2.  With  V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3  (d = the four input columns, any row,
// any channel) and  U0 = g0, U1 = g0 + g1 + g2, U2 = g0 - g1 + g2, U3 = g2  (g = the three taps of one kernel row):
//     M_v = sum over kernel rows kh and input channels ci of  V_v[row + kh - 1][ci] * U_v[kh][ci][co]      v = 0..3
//     y[2t]   = M0 + (M1 + M2) / 2          y[2t+1] = (M1 - M2) / 2 - M3
// Four GEMMs of depth 3*Ci over HALF as many rows instead of one GEMM of depth 9*Ci: 12 multiplications per output
// pair and channel pair instead of 18 -- 2/3 of the matrix instructions of the direct form (vq2_conv.hip), which is
// what bounds this layer (0.87 of the fp32 MFMA peak there; 157 TFLOP/s is all the chip has).  Everything is still
// fp32 with fp32 accumulation; the rounding differs from the direct sum the way any other summation order does
// (measured against fp64: 1.5 - 2x the direct form's 2e-7 relative error; F(4,3) and the 2-D F(2x2,3x3) were not
// taken: 16 accumulator sets per output tile do not fit the register file next to a 32x32 MFMA tile).
//
// Workgroup = TR image rows x TPW column pairs (64 pairs = 128 output pixels) x 128 (or 64) output channels; wave (wm, wn)
// owns 32 pairs x 32*NT channels for all four v (lane-local output transform).  Per 8-channel block the RAW halo patch
// ((TR+2) x (2*TPW+2) pixels, ReLU applied) is staged once and serves the three kernel rows; a lane builds its four V
// fragments from four patch reads.  The weight panel stays in the direct kernel's packed layout [co][(kh,kw,ci)]
// (include/vq2.h: same ABI): the three taps of a kernel row are transformed while they are staged.
#include "vq2_conv.h"

#ifndef VQ2_WINO_EXP
#define VQ2_WINO_EXP 0   // timing experiments (wrong results): 1 no loads in the loop, 2 no loads + no LDS stores, 3 no input
#endif                   // transform, 4 no weight transform, 5 no barrier in the loop, 6 no output stores

namespace vq2 {
namespace wino {

constexpr int TP = 64;             // column pairs per workgroup
constexpr int BN = 128;            // output channels per workgroup
constexpr int BK = 8, LDK = BK + 4;   // 48-byte LDS rows: conflict-free ds_read_b128 (vq2_conv.hip)

// TPW: pairs per tile row (32: rows of whole 64-pixel segments, 2-row tiles; 16: 32-pixel segments, 4-row tiles -- a wave's
// 32 pairs are then two rows).  (NT, BNT): a wave owns 32 pairs x 32*NT channels of a BNT-channel tile -- (2, 128), or (1, 64)
// for layers with 64-channel outputs and for the 32x32 level, where 128-wide tiles would leave one workgroup per CU.
template <int TPW, int NT, int BNT = BN>
struct Geo {
    static constexpr int NWN = BNT / (32 * NT), NTHR = 256;
    static_assert(NWN == 2, "two waves along the pairs, two along the channels");
    static constexpr int TR = TP / TPW, PR = TR + 2, PW = 2 * TPW + 2, NPX = PR * PW;
    static constexpr int A_FLOATS = (NPX + 1) * LDK;          // + one dump row for the items past the patch
    static constexpr int A_ITEMS = NPX * 2;                    // (pixel, 4-channel quad)
    static constexpr int A_LD = (A_ITEMS + NTHR - 1) / NTHR;
    static constexpr int B_FLOATS_T = 4 * BNT * LDK;
    static constexpr size_t LDS_BYTES = (size_t)2 * (A_FLOATS + B_FLOATS_T) * sizeof(float);
};

__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Epilogue shared by the kernels of this file: lane (lane & 31) of a wave holds the first output pixel of its pair in
// pix_lane (shared through ds_bpermute), the second one is dpix further; vals(j, r, y0, y1) yields the two outputs of
// accumulator row r of column block j.
// Then exactly the direct kernel's epilogue (vq2_conv.hip): bias, ReLU mask, residual, ReLU, strided store.
template <int NT, class F>
__device__ __forceinline__ void store_pairs(const ConvGemmParams &P, int co0, int lane, int pix_lane, F vals, int dpix = 1) {
    const int ybytes = P.N * P.Hy * P.Wy * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, ybytes * P.ldy, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rmk =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.mask ? P.mask : P.y), 0, P.mask ? ybytes * P.ldm : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.res ? P.res : P.y), 0, P.res ? ybytes * P.ldr : 0, RSRC_FLAGS);
    const bool has_mask = P.mask != nullptr, has_res = P.res != nullptr;
    const bool mask_first = has_mask && !P.mask_after, mask_last = has_mask && P.mask_after;
    const int colq = lane & 31, rowq = 4 * (lane >> 5);
    const int ldy4 = P.ldy * 4, ldm4 = P.ldm * 4, ldr4 = P.ldr * 4;
    const int relu_bits = P.relu_out ? 0 : (int)0x80000000;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = co0 + j * 32 + colq;
        const float bv = (P.bias && co < P.nbias) ? P.bias[co] : 0.f;
        const int co4 = co * 4;
#pragma unroll
        for (int rb4 = 0; rb4 < 16; rb4 += 4) {
            int pix[8];
            float mk[8], rs[8], val[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rb4 + q;
                const int rr = rowq + (r & 3) + 8 * (r >> 2);
                const int p0 = __shfl(pix_lane, rr, 64);
                pix[2 * q] = p0;
                pix[2 * q + 1] = p0 + dpix;
                vals(j, r, val[2 * q], val[2 * q + 1]);
            }
            if (has_mask) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    mk[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmk, pix[q] * ldm4 + co4, 0, 0));
            }
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    rs[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, pix[q] * ldr4 + co4, 0, 0));
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = val[q] + bv;
                if (mask_first) v = (mk[q] > 0.f) ? v : 0.f;
                if (has_res) v += rs[q];
                if (mask_last) v = (mk[q] > 0.f) ? v : 0.f;
                v = relu_floor(v, relu_bits);
                if (VQ2_WINO_EXP != 6 || v == 123.456f)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, pix[q] * ldy4 + co4, 0, 0);
            }
        }
    }
}

// CLOCK (diagnostic instantiation, scripts/clock_probe.py with VQ2_CLOCKPROBE=1): workgroups 8, 264, ... leave their lifetime in
// shader cycles (s_memtime), in 10 ns ticks (s_memrealtime) and their MFMA count per wave.
template <int TPW, int NT, int BNT, bool RELU_IN, bool CLOCK = false>
__global__ __launch_bounds__(256, NT == 1 ? 3 : 2) void wino3_kernel(const ConvGemmParams P) {
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if constexpr (CLOCK) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    using G = Geo<TPW, NT, BNT>;
    constexpr int B_FLOATS_T = G::B_FLOATS_T;
    constexpr int NWN = G::NWN, NTHR = G::NTHR, PW = G::PW, NPX = G::NPX, A_FLOATS = G::A_FLOATS, A_LD = G::A_LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                    // [2][A_FLOATS]   raw patch of one 8-channel block
    float *Bs = smem + 2 * A_FLOATS;     // [2][4][BNT][LDK] transformed taps of one (kernel row, channel block)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int ntn = P.Co / BNT, tw = P.W / (2 * TPW), th = P.H / G::TR;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (vid % ntn) * BNT;
    const int sp = vid / ntn;
    const int wb = sp % tw, hb = (sp / tw) % th, n = sp / (tw * th);
    const int h0 = hb * G::TR, w0 = wb * 2 * TPW;

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.N * P.H * P.W * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w), 0, P.Co * P.K * 4, RSRC_FLAGS);

    // ---- staging coordinates.  Patch items (pixel, quad): loads are UNCONDITIONAL -- an item outside the image or past
    // the patch carries an offset beyond the descriptor's range (reads 0) and an LDS slot in the dump row.
    int a_off[A_LD], a_dst[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int it = tid + NTHR * j;
        const bool ok = it < G::A_ITEMS;
        const int px = ok ? (it >> 1) : 0, q = it & 1;
        const int pr = px / PW, pc = px - pr * PW;
        const int row = h0 - 1 + pr, col = w0 - 1 + pc;
        const bool in = ok && (unsigned)row < (unsigned)P.H && (unsigned)col < (unsigned)P.W;
        a_off[j] = in ? (((n * P.H + row) * P.W + col) * P.ldx + 4 * q) * 4 : (int)0x80000000;
        a_dst[j] = (ok ? px : NPX) * LDK + 4 * q;
    }
    // weight items (co, quad): the first 2 * BNT threads
    const bool role_b = tid < 2 * BNT;
    const int bco = (tid & (2 * BNT - 1)) >> 1, bq = tid & 1;
    const int b_off = ((n0 + bco) * P.K + 4 * bq) * 4;
    const int b_dst = bco * LDK + 4 * bq;
    const int ci4 = P.Ci * 4;

    u32x4 ra[A_LD], rb[3];
    auto load_b = [&](int kh, int cb) {
        const int kb = (kh * 3 * P.Ci + cb) * 4;   // scalar
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) rb[kw] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_off + kb + kw * ci4, 0, 0);
    };
    auto store_b = [&](float *b) {
        const float4 g0 = as_f4(rb[0]), g1 = as_f4(rb[1]), g2 = as_f4(rb[2]);
        const float4 t = add4(g0, g2);
        *reinterpret_cast<float4 *>(b + 0 * BNT * LDK + b_dst) = g0;
        *reinterpret_cast<float4 *>(b + 1 * BNT * LDK + b_dst) = VQ2_WINO_EXP == 4 ? g1 : add4(t, g1);
        *reinterpret_cast<float4 *>(b + 2 * BNT * LDK + b_dst) = VQ2_WINO_EXP == 4 ? g1 : sub4(t, g1);
        *reinterpret_cast<float4 *>(b + 3 * BNT * LDK + b_dst) = g2;
    };
    auto store_a = [&](float *a, int j) {
        const float4 v = as_f4(ra[j]);
        *reinterpret_cast<float4 *>(a + a_dst[j]) = RELU_IN ? relu4(v) : v;
    };

    f32x16 acc[4][NT];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][j][r] = 0.f;

    const int frag_row = lane & 31, frag_k = 4 * (lane >> 5);
    const int pi = wm * 32 + frag_row;                 // this lane's column pair of the tile
    const int pr_l = pi / TPW, pt_l = pi - pr_l * TPW;
    const int lane_a = (pr_l * PW + 2 * pt_l) * LDK + frag_k;
    const int lane_b = (wn * NT * 32 + frag_row) * LDK + frag_k;

    auto compute = [&](const float *a, const float *b, int kh) {
        const float *ap = a + lane_a + kh * PW * LDK;
        const float4 d0 = *reinterpret_cast<const float4 *>(ap);
        const float4 d1 = *reinterpret_cast<const float4 *>(ap + LDK);
        const float4 d2 = *reinterpret_cast<const float4 *>(ap + 2 * LDK);
        const float4 d3 = *reinterpret_cast<const float4 *>(ap + 3 * LDK);
        float4 fb[4][NT];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[v][j] = *reinterpret_cast<const float4 *>(b + (v * BNT + j * 32) * LDK + lane_b);
        float4 fv[4];
        if (VQ2_WINO_EXP == 3) { fv[0] = d0; fv[1] = d1; fv[2] = d2; fv[3] = d3; } else {
        fv[0] = sub4(d0, d2);
        fv[1] = add4(d1, d2);
        fv[2] = sub4(d2, d1);
        fv[3] = sub4(d1, d3);
        }
#define VQ2_WINO_STEP(C)                                                                                          \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) _Pragma("unroll") for (int j = 0; j < NT; ++j)                  \
        acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[v].C, fb[v][j].C, acc[v][j], 0, 0, 0);
        VQ2_WINO_STEP(x) VQ2_WINO_STEP(y) VQ2_WINO_STEP(z) VQ2_WINO_STEP(w)
#undef VQ2_WINO_STEP
    };

    // ---- prologue: patch of channel block 0, taps of (kh 0, block 0)
    const int NCB = P.Ci / BK;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, a_off[j], 0, 0);
    load_b(0, 0);
#pragma unroll
    for (int j = 0; j < A_LD; ++j) store_a(As, j);
    if (role_b) store_b(Bs);
    __syncthreads();

    // ---- main loop: chunk c = (channel block cbi, kernel row kh), kh innermost.  Behind the loads of chunk c+1 run the
    // MFMAs of chunk c; the stores go to the other buffers; one barrier per chunk.
    int c = 0;
    for (int cbi = 0; cbi < NCB; ++cbi) {
        const int cbn = (cbi + 1 < NCB) ? cbi + 1 : cbi;    // (the last block re-loads itself into the idle buffer)
        const float *a_cur = As + (cbi & 1) * A_FLOATS;
        float *a_nxt = As + ((cbi + 1) & 1) * A_FLOATS;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int nkh = (kh + 1) % 3;
            const int ncb = (kh == 2) ? cbn : cbi;
            if (VQ2_WINO_EXP != 1 && VQ2_WINO_EXP != 2) {
                if (role_b && VQ2_WINO_EXP != 8) load_b(nkh, ncb * BK);
                // the next block's patch: issued BEHIND the weight loads of kernel row 0 and stored a chunk later, so that an
                // activation line that has to come from HBM has two chunks of time and never holds up the weight tile
                // (loads return in order: waiting for a younger load waits for every older one)
                if (kh == 0 && VQ2_WINO_EXP != 7) {
#pragma unroll
                    for (int j = 0; j < A_LD; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, a_off[j] + cbn * BK * 4, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the loads to just before their stores)
            compute(a_cur, Bs + (c & 1) * B_FLOATS_T, kh);
            __builtin_amdgcn_sched_barrier(0);
            if (VQ2_WINO_EXP != 2) {
                if (role_b) store_b(Bs + ((c + 1) & 1) * B_FLOATS_T);
                if (kh == 1) {
#pragma unroll
                    for (int j = 0; j < A_LD; ++j) store_a(a_nxt, j);
                }
            }
            if (VQ2_WINO_EXP != 5) __syncthreads();
            ++c;
        }
    }

    // ---- epilogue: output transform in registers, then the direct kernel's epilogue on two pixels per row
    const int pix_lane = ((n * P.H + h0 + pr_l) * P.W + w0 + 2 * pt_l);   // first pixel of pair (wm * 32 + lane & 31)
    store_pairs<NT>(P, n0 + wn * NT * 32, lane, pix_lane, [&](int j, int r, float &y0, float &y1) {
        const float m0 = acc[0][j][r], m1 = acc[1][j][r], m2 = acc[2][j][r], m3 = acc[3][j][r];
        y0 = m0 + 0.5f * (m1 + m2);
        y1 = 0.5f * (m1 - m2) - m3;
    });
    if constexpr (CLOCK) {
        if (P.stamps && tid == 0 && (blockIdx.x & 255) == 8 && blockIdx.x < 1024) {
            const int slot = blockIdx.x >> 8;
            P.stamps[slot * 4 + 0] = __builtin_amdgcn_s_memtime() - clk_t0;
            P.stamps[slot * 4 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
            P.stamps[slot * 4 + 2] = (unsigned long long)NCB * 3 * 16 * NT;   // MFMAs of one wave
        }
    }
}

// ====================================================================== 4x4 stride-2 pad-1 convolution
// (the down-sampling convs, vqvae.py:105-107, and the data gradient of ConvTranspose2d(k4,s2,p1), which is the same
// operation on dy.)  Along a row the four taps split by column parity into two 2-tap stride-1 filters,
//     y[wo] = (g0 o[wo] + g2 o[wo+1]) + (g1 e[wo] + g3 e[wo+1]),   o[i] = x[2i-1], e[i] = x[2i],
// and each of them runs as F(2,2): for the output pair (2t, 2t+1) and d0..d2 = three consecutive entries of o (or e)
//     M1 = (d0 - d1) ga,  M2 = d1 (ga + gb),  M3 = (d1 - d2) gb,   y[2t] = M1 + M2,  y[2t+1] = M2 - M3
// -- three products instead of four: 3/4 of the matrix instructions of the direct form, three accumulator sets
// (summed over kernel rows, parities and channels before the output transform).  Same tile and pipeline as above; the
// patch of one (kernel row, 8-channel block) holds both parities de-interleaved ([row][parity][65]: a lane's three
// reads are then 96 bytes from its neighbour's, conflict-free) and serves two chunks.
// TPW = pairs per tile row: 32 (output rows of whole 64-pixel segments, 2-row tiles) or 16 (32-pixel segments, 4-row tiles:
// the 32x32 level); (NT, BNT) = (2, 128) or (1, 64) as in the 3x3 kernel.
template <int TPW, int BNT>
struct K4 {
    static constexpr int TRO = TP / TPW;                  // output rows of the tile
    static constexpr int NPH = 2 * TPW + 1;               // entries of one parity in a patch row: 2t + {0,1,2}
    static constexpr int NPX = TRO * 2 * NPH;             // [output row of the tile][parity][entry]
    static constexpr int A_FLOATS = (NPX + 1) * LDK;
    static constexpr int A_ITEMS = NPX * 2;
    static constexpr int A_LD = (A_ITEMS + 255) / 256;
    static constexpr int B3_FLOATS = 3 * BNT * LDK;
    static constexpr size_t LDS_BYTES = (size_t)2 * (A_FLOATS + B3_FLOATS) * sizeof(float);
};

template <int TPW, int NT, int BNT, bool RELU_IN>
__global__ __launch_bounds__(256, NT == 1 ? 3 : 2) void wino_k4s2_kernel(const ConvGemmParams P) {
    constexpr int NWN = 2;
    static_assert(BNT == 64 * NT, "two waves along the channels");
    using G = K4<TPW, BNT>;
    constexpr int TRO = G::TRO, NPH = G::NPH, NPX = G::NPX, A_FLOATS = G::A_FLOATS, A_ITEMS = G::A_ITEMS, A_LD = G::A_LD,
                  B3_FLOATS = G::B3_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                    // [2][A_FLOATS]
    float *Bs = smem + 2 * A_FLOATS;     // [2][3][BNT][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int ntn = P.Co / BNT, tw = P.Wo / (2 * TPW), th = P.Ho / TRO;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (vid % ntn) * BNT;
    const int sp = vid / ntn;
    const int wb = sp % tw, hb = (sp / tw) % th, n = sp / (tw * th);
    const int h0 = hb * TRO, w0 = wb * 2 * TPW;   // first output row / column of the tile
    const int cx0 = 2 * w0 - 1;               // input column of entry 0 of the odd parity

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.N * P.H * P.W * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w), 0, P.Co * P.K * 4, RSRC_FLAGS);

    // patch items (entry, quad) for kernel row 0; kernel row kh adds kh input rows.  a_vh: bit kh = that row is inside
    int a_off[A_LD], a_dst[A_LD];
    unsigned a_vh[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int it = tid + 256 * j;
        const bool ok = it < A_ITEMS;
        const int px = ok ? (it >> 1) : 0, q = it & 1;
        const int r = px / (2 * NPH), rem = px - r * 2 * NPH;
        const int par = rem / NPH, i = rem - par * NPH;
        const int col = cx0 + 2 * i + par;
        const int row0 = 2 * (h0 + r) - 1;
        const bool cin = ok && (unsigned)col < (unsigned)P.W;
        a_off[j] = (((n * P.H + row0) * P.W + col) * P.ldx + 4 * q) * 4;
        a_vh[j] = 0;
        for (int kh = 0; kh < 4; ++kh) a_vh[j] |= (cin && (unsigned)(row0 + kh) < (unsigned)P.H ? 1u : 0u) << kh;
        a_dst[j] = (ok ? px : NPX) * LDK + 4 * q;
    }
    const bool role_b = tid < 2 * BNT;
    const int bco = (tid & (2 * BNT - 1)) >> 1, bq = tid & 1;
    const int b_off = ((n0 + bco) * P.K + 4 * bq) * 4;
    const int b_dst = bco * LDK + 4 * bq;
    const int ci4 = P.Ci * 4, row4 = P.W * P.ldx * 4;

    u32x4 ra[A_LD], rb[2];
    auto load_b = [&](int kh, int par, int cb) {
        const int kb = ((kh * 4 + par) * P.Ci + cb) * 4;   // scalar: tap (kh, par); the second tap is (kh, par + 2)
        rb[0] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_off + kb, 0, 0);
        rb[1] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_off + kb + 2 * ci4, 0, 0);
    };
    auto store_b = [&](float *b) {
        const float4 ga = as_f4(rb[0]), gb = as_f4(rb[1]);
        *reinterpret_cast<float4 *>(b + 0 * BNT * LDK + b_dst) = ga;
        *reinterpret_cast<float4 *>(b + 1 * BNT * LDK + b_dst) = add4(ga, gb);
        *reinterpret_cast<float4 *>(b + 2 * BNT * LDK + b_dst) = gb;
    };
    auto load_a = [&](int kh, int cb) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const bool in = ((a_vh[j] >> kh) & 1u) != 0u;
            ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, in ? a_off[j] + kh * row4 + cb * 4 : OOB, 0, 0);
        }
    };
    auto store_a = [&](float *a) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const float4 v = as_f4(ra[j]);
            *reinterpret_cast<float4 *>(a + a_dst[j]) = RELU_IN ? relu4(v) : v;
        }
    };

    f32x16 acc[3][NT];
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][j][r] = 0.f;

    const int frag_row = lane & 31, frag_k = 4 * (lane >> 5);
    const int pi = wm * 32 + frag_row;                                   // this lane's pair of the tile
    const int pr_l = pi / TPW, pt_l = pi - pr_l * TPW;
    const int lane_a = (pr_l * 2 * NPH + 2 * pt_l) * LDK + frag_k;
    const int lane_b = (wn * NT * 32 + frag_row) * LDK + frag_k;

    auto compute = [&](const float *a, const float *b, int par) {
        const float *ap = a + lane_a + par * NPH * LDK;
        const float4 d0 = *reinterpret_cast<const float4 *>(ap);
        const float4 d1 = *reinterpret_cast<const float4 *>(ap + LDK);
        const float4 d2 = *reinterpret_cast<const float4 *>(ap + 2 * LDK);
        float4 fb[3][NT];
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[v][j] = *reinterpret_cast<const float4 *>(b + (v * BNT + j * 32) * LDK + lane_b);
        float4 fv[3];
        fv[0] = sub4(d0, d1);
        fv[1] = d1;
        fv[2] = sub4(d1, d2);
#define VQ2_WINO_STEP(C)                                                                                          \
    _Pragma("unroll") for (int v = 0; v < 3; ++v) _Pragma("unroll") for (int j = 0; j < NT; ++j)                  \
        acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[v].C, fb[v][j].C, acc[v][j], 0, 0, 0);
        VQ2_WINO_STEP(x) VQ2_WINO_STEP(y) VQ2_WINO_STEP(z) VQ2_WINO_STEP(w)
#undef VQ2_WINO_STEP
    };

    // ---- prologue: patch (block 0, kernel row 0), taps of its odd parity
    const int NCB = P.Ci / BK;
    load_a(0, 0);
    load_b(0, 0, 0);
    store_a(As);
    if (role_b) store_b(Bs);
    __syncthreads();

    // ---- main loop over patches ai = (32-channel group, kernel row, 8-channel block), two chunks (parities) per patch.
    // The 8-channel blocks of a group are INNERMOST: the four 32-byte quarters of an activation line are consumed in eight
    // consecutive chunks (L1 / L2 hits) instead of once per sweep over the kernel rows -- with the blocks outermost a line came
    // back from HBM four times (measured: 494 MB fetched per launch for a 134 MB input).  The sum is the same set of products.
    const int GB = (P.Ci % 32 == 0) ? 4 : 1;                     // 8-channel blocks per group
    const int NPATCH = NCB * 4;
    auto patch_of = [&](int a, int &kh, int &cb) {               // scalar
        const int grp = a / (4 * GB), rem = a - grp * 4 * GB;
        kh = rem / GB;
        cb = (grp * GB + rem - kh * GB) * BK;
    };
    int c = 0;
    for (int ai = 0; ai < NPATCH; ++ai) {
        int kh, cb, akh, ncb;
        patch_of(ai, kh, cb);
        patch_of(ai + 1 < NPATCH ? ai + 1 : ai, akh, ncb);          // (the last patch re-loads itself into the idle buffer)
        const float *a_cur = As + (ai & 1) * A_FLOATS;
        float *a_nxt = As + ((ai + 1) & 1) * A_FLOATS;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            if (par == 0) {
                load_b(kh, 1, cb);
                load_a(akh, ncb);                                  // behind the weight loads, stored a chunk later
            } else {
                load_b(akh, 0, ncb);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(a_cur, Bs + (c & 1) * B3_FLOATS, par);
            __builtin_amdgcn_sched_barrier(0);
            if (role_b) store_b(Bs + ((c + 1) & 1) * B3_FLOATS);
            if (par == 1) store_a(a_nxt);
            __syncthreads();
            ++c;
        }
    }

    const int pix_lane = ((n * P.Ho + h0 + pr_l) * P.Wo + w0 + 2 * pt_l);
    store_pairs<NT>(P, n0 + wn * NT * 32, lane, pix_lane, [&](int j, int r, float &y0, float &y1) {
        const float m1 = acc[0][j][r], m2 = acc[1][j][r], m3 = acc[2][j][r];
        y0 = m1 + m2;
        y1 = m2 - m3;
    });
}

template <int TPW, int NT, int BNT>
static int launch_k4s2(const ConvGemmParams &P, hipStream_t s) {
    using G = K4<TPW, BNT>;
    auto kern = P.relu_in ? wino_k4s2_kernel<TPW, NT, BNT, true> : wino_k4s2_kernel<TPW, NT, BNT, false>;
    allow_big_lds(kern, G::LDS_BYTES);
    const unsigned nwg = (unsigned)(P.N * (P.Ho / G::TRO) * (P.Wo / (2 * TPW)) * (P.Co / BNT));
    const char *name = "conv_wino_k4s2";
    if (prof_enabled()) name = prof_label("conv_wino_k4s2<%dx%d,nt%d>|M=%d,N=%d,K=%d", G::TRO, 2 * TPW, NT, P.M, P.Co, P.K);
    ProfScope prof(name, P.flops, P.bytes, s, true);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), G::LDS_BYTES, s, P);
    return check_launch("wino_k4s2_kernel");
}

// ====================================================================== sub-pixel form of ConvTranspose2d(k4,s2,p1)
// (forward, and the data gradient of a 4x4 stride-2 conv.)  Output phase (ph, pw) is a 2x2 stride-1 conv over the input,
//     y[2h+ph][2w+pw] = sum_{a,b} x[h+a-1+ph][w+b-1+pw] g[ph][pw][a][b]      (panel [phase][co][(a, b, ci)]),
// i.e. along a row a 2-tap filter: F(2,2) over pairs of same-phase outputs (input columns 2t, 2t+1), three products instead
// of four.  Workgroup = 4 input rows x 32 pairs x 64 output channels of ONE phase (blockIdx -> phase fastest: the four
// phases of a tile run next to each other on one XCD and share the patch lines in its L2); wave = one row, 32 pairs x 64
// channels; the 5 x 65 patch of an 8-channel block serves both kernel rows a.
// TPW = pairs per tile row: 32 (input rows of whole 64-pixel segments, 4-row tiles, one row per wave) or 16 (32-pixel
// segments, 8-row tiles, two rows per wave: the 32x32 level).
constexpr int BN2 = 64;
template <int TPW>
struct Sp2 {
    static constexpr int TR = 128 / TPW, NPC = 2 * TPW + 1, NPX = (TR + 1) * NPC;    // patch rows x entries (2t + {0,1,2})
    static constexpr int A_FLOATS = (NPX + 1) * LDK;
    static constexpr int A_ITEMS = NPX * 2;
    static constexpr int A_LD = (A_ITEMS + 255) / 256;
    static constexpr int B_FLOATS2 = 3 * BN2 * LDK;
    static constexpr size_t LDS_BYTES = (size_t)2 * (A_FLOATS + B_FLOATS2) * sizeof(float);
};

template <int TPW, bool RELU_IN>
__global__ __launch_bounds__(256, 2) void wino_subpixel_kernel(const ConvGemmParams P) {
    constexpr int NT = 2;
    using G = Sp2<TPW>;
    constexpr int TR = G::TR, NPC = G::NPC, NPX = G::NPX, A_FLOATS = G::A_FLOATS, A_ITEMS = G::A_ITEMS, A_LD = G::A_LD,
                  B_FLOATS2 = G::B_FLOATS2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                    // [2][A_FLOATS]
    float *Bs = smem + 2 * A_FLOATS;     // [2][3][BN2][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = P.Co / BN2, tw = P.W / (2 * TPW), th = P.H / TR;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int phase = vid & 3;
    const int ph = phase >> 1, pw = phase & 1;
    const int v2 = vid >> 2;
    const int n0 = (v2 % ntn) * BN2;
    const int sp = v2 / ntn;
    const int wb = sp % tw, hb = (sp / tw) % th, n = sp / (tw * th);
    const int h0 = hb * TR, w0 = wb * 2 * TPW;
    const int row0 = h0 - 1 + ph, col0 = w0 - 1 + pw;       // input pixel of patch entry (0, 0)

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.N * P.H * P.W * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(P.w + (size_t)phase * P.Co * P.K), 0, P.Co * P.K * 4, RSRC_FLAGS);

    int a_off[A_LD], a_dst[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int it = tid + 256 * j;
        const bool ok = it < A_ITEMS;
        const int px = ok ? (it >> 1) : 0, q = it & 1;
        const int r = px / NPC, e = px - r * NPC;
        const int row = row0 + r, col = col0 + e;
        const bool in = ok && (unsigned)row < (unsigned)P.H && (unsigned)col < (unsigned)P.W;
        a_off[j] = in ? (((n * P.H + row) * P.W + col) * P.ldx + 4 * q) * 4 : (int)0x80000000;
        a_dst[j] = (ok ? px : NPX) * LDK + 4 * q;
    }
    const bool role_b = tid < 2 * BN2;
    const int bco = (tid & (2 * BN2 - 1)) >> 1, bq = tid & 1;
    const int b_off = ((n0 + bco) * P.K + 4 * bq) * 4;
    const int b_dst = bco * LDK + 4 * bq;
    const int ci4 = P.Ci * 4;

    u32x4 ra[A_LD], rb[2];
    auto load_b = [&](int a, int cb) {
        const int kb = (a * 2 * P.Ci + cb) * 4;       // taps (a, 0) and (a, 1)
        rb[0] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_off + kb, 0, 0);
        rb[1] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_off + kb + ci4, 0, 0);
    };
    auto store_b = [&](float *b) {
        const float4 ga = as_f4(rb[0]), gb = as_f4(rb[1]);
        *reinterpret_cast<float4 *>(b + 0 * BN2 * LDK + b_dst) = ga;
        *reinterpret_cast<float4 *>(b + 1 * BN2 * LDK + b_dst) = add4(ga, gb);
        *reinterpret_cast<float4 *>(b + 2 * BN2 * LDK + b_dst) = gb;
    };
    auto load_a = [&](int cb) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, a_off[j] + cb * 4, 0, 0);
    };
    auto store_a = [&](float *a) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const float4 v = as_f4(ra[j]);
            *reinterpret_cast<float4 *>(a + a_dst[j]) = RELU_IN ? relu4(v) : v;
        }
    };

    f32x16 acc[3][NT];
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[v][j][r] = 0.f;

    const int frag_row = lane & 31, frag_k = 4 * (lane >> 5);
    const int pi = wave * 32 + frag_row;                                 // this lane's pair of the tile
    const int pr_l = pi / TPW, pt_l = pi - pr_l * TPW;
    const int lane_a = (pr_l * NPC + 2 * pt_l) * LDK + frag_k;
    const int lane_b = frag_row * LDK + frag_k;

    auto compute = [&](const float *a, const float *b, int arow) {
        const float *ap = a + lane_a + arow * NPC * LDK;
        const float4 d0 = *reinterpret_cast<const float4 *>(ap);
        const float4 d1 = *reinterpret_cast<const float4 *>(ap + LDK);
        const float4 d2 = *reinterpret_cast<const float4 *>(ap + 2 * LDK);
        float4 fb[3][NT];
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[v][j] = *reinterpret_cast<const float4 *>(b + (v * BN2 + j * 32) * LDK + lane_b);
        float4 fv[3];
        fv[0] = sub4(d0, d1);
        fv[1] = d1;
        fv[2] = sub4(d1, d2);
#define VQ2_WINO_STEP(C)                                                                                          \
    _Pragma("unroll") for (int v = 0; v < 3; ++v) _Pragma("unroll") for (int j = 0; j < NT; ++j)                  \
        acc[v][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[v].C, fb[v][j].C, acc[v][j], 0, 0, 0);
        VQ2_WINO_STEP(x) VQ2_WINO_STEP(y) VQ2_WINO_STEP(z) VQ2_WINO_STEP(w)
#undef VQ2_WINO_STEP
    };

    const int NCB = P.Ci / BK;
    load_a(0);
    load_b(0, 0);
    store_a(As);
    if (role_b) store_b(Bs);
    __syncthreads();

    int c = 0;
    for (int cbi = 0; cbi < NCB; ++cbi) {
        const int cbn = (cbi + 1 < NCB) ? (cbi + 1) * BK : cbi * BK;   // (the last block re-loads itself into the idle buffer)
        const float *a_cur = As + (cbi & 1) * A_FLOATS;
        float *a_nxt = As + ((cbi + 1) & 1) * A_FLOATS;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (a == 0) {
                load_b(1, cbi * BK);
                load_a(cbn);                                        // behind the weight loads, stored a chunk later
            } else {
                load_b(0, cbn);
            }
            __builtin_amdgcn_sched_barrier(0);
            compute(a_cur, Bs + (c & 1) * B_FLOATS2, a);
            __builtin_amdgcn_sched_barrier(0);
            if (role_b) store_b(Bs + ((c + 1) & 1) * B_FLOATS2);
            if (a == 1) store_a(a_nxt);
            __syncthreads();
            ++c;
        }
    }

    const int pix_lane = ((n * P.Hy + 2 * (h0 + pr_l) + ph) * P.Wy + 2 * (w0 + 2 * pt_l) + pw);
    store_pairs<NT>(P, n0, lane, pix_lane, [&](int j, int r, float &y0, float &y1) {
        const float m1 = acc[0][j][r], m2 = acc[1][j][r], m3 = acc[2][j][r];
        y0 = m1 + m2;
        y1 = m2 - m3;
    }, 2);
}

template <int TPW>
static int launch_subpixel2(const ConvGemmParams &P, hipStream_t s) {
    using G = Sp2<TPW>;
    auto kern = P.relu_in ? wino_subpixel_kernel<TPW, true> : wino_subpixel_kernel<TPW, false>;
    allow_big_lds(kern, G::LDS_BYTES);
    const unsigned nwg = (unsigned)(4 * P.N * (P.H / G::TR) * (P.W / (2 * TPW)) * (P.Co / BN2));
    const char *name = "conv_wino_subpixel";
    if (prof_enabled()) name = prof_label("conv_wino_subpixel<%dx%d>|M=%d,N=%d,K=%d,ph4", G::TR, 2 * TPW, P.M, P.Co, P.K);
    ProfScope prof(name, P.flops, P.bytes, s, true);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), G::LDS_BYTES, s, P);
    return check_launch("wino_subpixel_kernel");
}

template <int TPW, int NT, int BNT>
static int launch(const ConvGemmParams &P, hipStream_t s) {
    using G = Geo<TPW, NT, BNT>;
    auto kern = P.relu_in ? wino3_kernel<TPW, NT, BNT, true> : wino3_kernel<TPW, NT, BNT, false>;
    if constexpr (TPW == 32 && NT == 2) {
        if (P.stamps && !P.relu_in) kern = wino3_kernel<TPW, NT, BNT, false, true>;
    }
    allow_big_lds(kern, G::LDS_BYTES);
    const unsigned nwg = (unsigned)(P.N * (P.H / G::TR) * (P.W / (2 * TPW)) * (P.Co / BNT));
    const char *name = "conv_wino";
    if (prof_enabled()) name = prof_label("conv_wino3<%dx%d,nt%d>|M=%d,N=%d,K=%d", G::TR, 2 * TPW, NT, P.M, P.Co, P.K);
    ProfScope prof(name, P.flops, P.bytes, s, true);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(G::NTHR), G::LDS_BYTES, s, P);
    return check_launch("wino3_kernel");
}

static int tune(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace wino

// Shapes the Winograd kernels take: 3x3, stride 1, pad 1, output the size of the input, whole 64-channel output tiles,
// 8-channel input blocks (at least 32 channels), rows of whole 64-pixel segments (or 32-pixel ones with H % 4 == 0), tensors
// below 1 GiB (32-bit offsets with an additive out-of-range penalty); likewise the 4x4 stride-2 and sub-pixel forms.
bool wino3_ok(const ConvGemmParams &P) {
    static const int on = wino::tune("VQ2_WINO", 1), on4 = wino::tune("VQ2_WINO_K4", 1);
    const long gib = 1L << 30;
    static const int onsp = wino::tune("VQ2_WINO_SP", 1);
    if (P.phases == 4)   // sub-pixel conv-transpose: F(2,2) per output phase
        return onsp && P.KH == 2 && P.KW == 2 && P.K == 4 * P.Ci && P.Hy == 2 * P.H && P.Wy == 2 * P.W &&
               ((P.W % 64 == 0 && P.H % 4 == 0) || (P.W % 32 == 0 && P.H % 8 == 0)) && P.Ci % wino::BK == 0 && P.Ci >= 32 && P.Co % 64 == 0 && P.ldx % 4 == 0 &&
               (long)P.N * P.H * P.W * P.ldx * 4 < gib && (long)P.N * P.Hy * P.Wy * P.ldy * 4 < gib &&
               (long)P.N * P.Hy * P.Wy * (P.ldm > P.ldr ? P.ldm : P.ldr) * 4 < gib && (long)4 * P.Co * P.K * 4 < gib;
    if (on4 && P.KH == 4 && P.KW == 4 && P.stride == 2 && P.pad_h == 1 && P.pad_w == 1 && P.phases == 1)   // F(2,2) by parity
        return 2 * P.Ho == P.H && 2 * P.Wo == P.W && P.Hy == P.Ho && P.Wy == P.Wo &&
               ((P.Wo % 64 == 0 && P.Ho % 2 == 0) || (P.Wo % 32 == 0 && P.Ho % 4 == 0)) &&
               P.Ci % wino::BK == 0 && P.Ci >= 32 && P.Co % 64 == 0 && P.ldx % 4 == 0 &&
               // (64-channel outputs on 32-pixel rows lose to the direct 64-row tiles -- 128 -> 64 at 32x32 and batch 32: one tile
               //  per CU, 94.5 vs 84.3 us.  The rule must not look at the batch size: results do not depend on it.)
               (P.Co % 128 == 0 || P.Wo % 64 == 0) &&
               (long)P.N * P.H * P.W * P.ldx * 4 < gib && (long)P.N * P.Ho * P.Wo * P.ldy * 4 < gib &&
               (long)P.N * P.Ho * P.Wo * (P.ldm > P.ldr ? P.ldm : P.ldr) * 4 < gib && (long)P.Co * P.K * 4 < gib;
    const bool rows64 = P.W % 64 == 0 && P.H % 2 == 0, rows32 = P.W % 32 == 0 && P.H % 4 == 0;
    static const int minci = wino::tune("VQ2_WINO_MINCI", 32);   // (3x3 32 -> 128 at 64x64: 89.6 -> 74.5 us)
    // (diagnostic only -- a non-zero value makes the choice, and with it the rounding, depend on the batch size)
    static const int minwg = wino::tune("VQ2_WINO_MINWG", 0);    // fewest 64-pair x 64-channel tiles worth a Winograd launch
    if ((long)P.N * P.H * P.W / 128 * (P.Co / 64) < minwg) return false;
    return on && P.KH == 3 && P.KW == 3 && P.stride == 1 && P.pad_h == 1 && P.pad_w == 1 && P.phases == 1 &&
           P.Ho == P.H && P.Wo == P.W && P.Hy == P.H && P.Wy == P.W && P.Ci % wino::BK == 0 && P.Ci >= minci &&
           P.Co % 64 == 0 && (rows64 || rows32) && P.ldx % 4 == 0 &&
           (long)P.N * P.H * P.W * P.ldx * 4 < gib && (long)P.N * P.H * P.W * P.ldy * 4 < gib &&
           (long)P.N * P.H * P.W * (P.ldm > P.ldr ? P.ldm : P.ldr) * 4 < gib && (long)P.Co * P.K * 4 < gib;
}

int launch_wino3(const ConvGemmParams &P, hipStream_t s) {
    if (P.phases == 4) return P.W % 64 == 0 && P.H % 4 == 0 ? wino::launch_subpixel2<32>(P, s) : wino::launch_subpixel2<16>(P, s);
    // 128-channel tiles where they still give every CU two workgroups; 64-channel tiles otherwise (64-channel outputs, the
    // 32x32 level)
    const long wgs128 = (long)P.N * P.Ho * P.Wo / 128 * (P.Co / 128);
    static const int force = wino::tune("VQ2_WINO_TILE", 0);     // 1: 128-channel tiles whenever possible, 2: 64-channel tiles
    const bool wide = P.Co % 128 == 0 && (force == 1 || (force != 2 && wgs128 >= 400));
    if (P.KH == 4) {
        if (P.Wo % 64 == 0 && P.Ho % 2 == 0)
            return wide ? wino::launch_k4s2<32, 2, 128>(P, s) : wino::launch_k4s2<32, 1, 64>(P, s);
        return wide ? wino::launch_k4s2<16, 2, 128>(P, s) : wino::launch_k4s2<16, 1, 64>(P, s);
    }
    if (P.W % 64 == 0 && P.H % 2 == 0) return wide ? wino::launch<32, 2, 128>(P, s) : wino::launch<32, 1, 64>(P, s);
    return wide ? wino::launch<16, 2, 128>(P, s) : wino::launch<16, 1, 64>(P, s);
}

}  // namespace vq2
