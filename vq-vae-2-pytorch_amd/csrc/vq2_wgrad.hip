// Weight gradient of conv2d / conv-transpose2d (NHWC activations, reference-layout output).
//
//   dw[o][i][kh][kw] = sum_{n,ho,wo} G[n,ho,wo,o] * X[n, ho*s+kh-p, wo*s+kw-p, i]
//
//   conv  : X = (relu) input x,  G = dy                  -> dw in OIHW  (o = Co, i = Ci)
//   convT : X = dy (the 2H x 2W tensor), G = (relu) x    -> dw in IOHW  (o = Ci, i = Co)
//
// This is a GEMM [O x M] * [M x K] (K = KH*KW*I) whose reduction dimension M = N*Ho*Wo is
// huge and whose output is tiny, so the reduction is split over `S` workgroups per output
// tile; every split writes its partial tile to a slab in the caller's workspace and a second
// launch sums the slabs in a fixed order (bit-reproducible, no float atomics) and scatters
// into the reference weight layout.  Both operands are consumed "reduction-major" straight
// from NHWC memory: v_mfma_f32_32x32x2_f32 wants A[i][k] / B[k][j] with lanes along i / j,
// and consecutive lanes read consecutive channels -> conflict-free ds_read_b32, no transposes.
#include "vq2_common.h"
#include <stdlib.h>

namespace vq2 {

struct WgradParams {
    const float *x;  // image operand [N,H,W,ldx], I channels
    const float *g;  // grad  operand [N,Ho,Wo,ldg], O channels
    float *ws;       // [S][O][K] partial slabs
    float *bias_ws;  // [S][O] partial column sums of g (bias gradient) or null
    int bias_taps;   // bias gradient from the GATHERED operand (it is dy: exchanged roles, conv-transpose): bit t set =
                     // tap t visits every dy pixel exactly once over the taps of this mask; the column sums of those
                     // k-blocks of the staged X tile are the bias gradient.  0 = bias from the G tile (plain conv)
    int bias_nslots; // popcount(bias_taps): bias_ws is [S][nslots][I]
    int N, H, W, I, ldx;
    int Ho, Wo, O, ldg;
    int KH, KW, stride, pad;
    int K, M;
    int rows_per_split;  // multiple of 32
    int relu_x, relu_g;
};

constexpr int WG_BKR = 32;  // reduction rows (pixels) per staged chunk

// rows staged per pass by the 256 threads when a row has c4 float4: the largest power of two <= 256 / c4
// (a 96-wide tile has 24 float4 per row -> 8 rows per pass, 64 threads sit the X staging out)
constexpr int rows_per_pass(int c4) {
    int r = 32;
    while (r * c4 > 256) r >>= 1;
    return r;
}

template <int WAVES_M, int WAVES_N, int MT, int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams P) {
    constexpr int BMO = WAVES_M * MT * 32;  // tile rows  (o)
    constexpr int BNK = WAVES_N * NT * 32;  // tile cols  (k)
    constexpr int G_C4 = BMO / 4, X_C4 = BNK / 4;
    constexpr int G_RSTEP = rows_per_pass(G_C4), X_RSTEP = rows_per_pass(X_C4);
    constexpr int G_LD = WG_BKR / G_RSTEP, X_LD = WG_BKR / X_RSTEP;
    static_assert(G_C4 * G_RSTEP == 256, "every thread stages G");
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(G_C4 <= 256 && X_C4 <= 256 && G_LD >= 1 && X_LD >= 1, "tile/thread mapping");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Gs = smem;                       // [2][32][BMO]
    float *Xs = smem + 2 * WG_BKR * BMO;    // [2][32][BNK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // 1-D grid, XCD-aware: the k-tiles and o-tiles of one split share its G/X pixels -> same L2
    const int ktiles = (P.K + BNK - 1) / BNK;
    const int otiles = (P.O + BMO - 1) / BMO;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int k0 = (vid % ktiles) * BNK;
    const int o0 = ((vid / ktiles) % otiles) * BMO;
    const int split = vid / (ktiles * otiles);
    const int m_begin = split * P.rows_per_split;
    const int m_end = min(P.M, m_begin + P.rows_per_split);

    // fixed per-thread column coordinates
    const int g_c = o0 + (tid % G_C4) * 4;
    const int g_r = tid / G_C4;
    const int x_k = k0 + (tid % X_C4) * 4;
    const int x_r = tid / X_C4;
    const bool g_cv = g_c < P.O;
    const bool x_act = tid < X_C4 * X_RSTEP;       // threads beyond the last full staging row sit X out
    const bool x_kv = x_act && x_k < P.K;
    int kh = 0, kw = 0, ci = 0;
    if (x_kv) {
        const int tap = x_k / P.I;
        ci = x_k - tap * P.I;
        kh = tap / P.KW;
        kw = tap - kh * P.KW;
    }
    const int HoWo = P.Ho * P.Wo;

    // (n, ho, wo) of this thread's X rows, advanced by 32 rows per chunk without divisions
    int xn[X_LD], xho[X_LD], xwo[X_LD];
#pragma unroll
    for (int j = 0; j < X_LD; ++j) {
        const int m = m_begin + x_r + j * X_RSTEP;
        xn[j] = m / HoWo;
        const int r = m - xn[j] * HoWo;
        xho[j] = r / P.Wo;
        xwo[j] = r - xho[j] * P.Wo;
    }
    float4 rg[G_LD], rx[X_LD];
    auto load_chunk = [&](int mbase) {
#pragma unroll
        for (int j = 0; j < G_LD; ++j) {
            const int m = mbase + g_r + j * G_RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_cv && m < m_end) v = *reinterpret_cast<const float4 *>(P.g + (size_t)m * P.ldg + g_c);
            rg[j] = v;
        }
#pragma unroll
        for (int j = 0; j < X_LD; ++j) {
            const int m = mbase + x_r + j * X_RSTEP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x_kv && m < m_end) {
                const int ih = xho[j] * P.stride - P.pad + kh;
                const int iw = xwo[j] * P.stride - P.pad + kw;
                if ((unsigned)ih < (unsigned)P.H && (unsigned)iw < (unsigned)P.W)
                    v = *reinterpret_cast<const float4 *>(P.x + ((size_t)(xn[j] * P.H + ih) * P.W + iw) * P.ldx + ci);
            }
            rx[j] = v;  // ReLU applied at the LDS write, so the load is not waited for here
            // next chunk: 32 rows further in (n, ho, wo) order
            xwo[j] += WG_BKR;
            while (xwo[j] >= P.Wo) {
                xwo[j] -= P.Wo;
                if (++xho[j] == P.Ho) { xho[j] = 0; ++xn[j]; }
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float *gs = Gs + buf * WG_BKR * BMO;
        float *xs = Xs + buf * WG_BKR * BNK;
#pragma unroll
        for (int j = 0; j < G_LD; ++j)
            *reinterpret_cast<float4 *>(gs + (g_r + j * G_RSTEP) * BMO + (tid % G_C4) * 4) = P.relu_g ? relu4(rg[j]) : rg[j];
#pragma unroll
        for (int j = 0; j < X_LD; ++j)
            if (X_C4 * X_RSTEP == 256 || x_act)
                *reinterpret_cast<float4 *>(xs + (x_r + j * X_RSTEP) * BNK + (tid % X_C4) * 4) = P.relu_x ? relu4(rx[j]) : rx[j];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (m_end - m_begin + WG_BKR - 1) / WG_BKR;
    if (nchunks > 0) {
        load_chunk(m_begin);
        store_chunk(0);
    }
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5;
    const bool do_bias = P.bias_ws != nullptr && !P.bias_taps && k0 == 0 && tid < BMO;
    int bx_col = -1;   // this thread's slot in a bias_ws row when its k column belongs to a bias tap
    if (P.bias_ws != nullptr && P.bias_taps && o0 == 0 && tid < BNK && k0 + tid < P.K) {
        const int tap = (k0 + tid) / P.I;
        if ((P.bias_taps >> tap) & 1) bx_col = __popc(P.bias_taps & ((1u << tap) - 1u)) * P.I + (k0 + tid - tap * P.I);
    }
    const bool do_bias_x = bx_col >= 0;
    float bsum = 0.f;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(m_begin + (c + 1) * WG_BKR);
        if (do_bias) {  // bias gradient rides along: column sums of the staged grad tile
            const float *gcol = Gs + buf * WG_BKR * BMO + tid;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < WG_BKR; r += 2) { s0 += gcol[r * BMO]; s1 += gcol[(r + 1) * BMO]; }
            bsum += s0 + s1;
        }
        if (do_bias_x) {
            const float *xcol = Xs + buf * WG_BKR * BNK + tid;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < WG_BKR; r += 2) { s0 += xcol[r * BNK]; s1 += xcol[(r + 1) * BNK]; }
            bsum += s0 + s1;
        }
        // a wave's 32-wide blocks are INTERLEAVED with the other waves' (block i of wave wm = i*WAVES_M + wm): the two
        // fragments of a k-step are then 256 bytes apart and k-steps 1 KiB apart -- one ds_read2st64_b32 with immediate
        // offsets per operand and step, no vector add to rebuild a base (vector instructions cost matrix time)
        const float *gs = Gs + buf * WG_BKR * BMO + fk * BMO + wm * 32 + fr;
        const float *xs = Xs + buf * WG_BKR * BNK + fk * BNK + wn * 32 + fr;
        // fragments double-buffered: the ds_reads of k-step kk+1 are in flight during the MFMAs of step kk
        float fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = gs[i * WAVES_M * 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = xs[j * WAVES_N * 32];
#pragma unroll
        for (int kk = 0; kk < WG_BKR / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < WG_BKR / 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[nxt][i] = gs[(kk + 1) * 2 * BMO + i * WAVES_M * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[nxt][j] = xs[(kk + 1) * 2 * BNK + j * WAVES_N * 32];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
            // pin the order: next step's LDS reads first, then this step's MFMAs (hipcc otherwise sinks the
            // reads to just before their use and waits lgkmcnt(0) in front of every MFMA group)
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
        }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    if (do_bias && o0 + tid < P.O) P.bias_ws[(size_t)split * P.O + o0 + tid] = bsum;
    if (do_bias_x) P.bias_ws[(size_t)split * P.bias_nslots * P.I + bx_col] = bsum;

    // partial tile -> slab [split][o][k]
    float *slab = P.ws + (size_t)split * P.O * P.K;
    const int colq = lane & 31, rowq = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int k = k0 + (j * WAVES_N + wn) * 32 + colq;
        if (k >= P.K) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int ob = o0 + (i * WAVES_M + wm) * 32 + rowq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ob + (r & 3) + 8 * (r >> 2);
                if (o < P.O) slab[(size_t)o * P.K + k] = acc[i][j][r];
            }
        }
    }
}

// ====================================================================== low-VALU variant
// Same math as wgrad_kernel for the common case Wo % 32 == 0: a 32-row chunk then lies inside ONE output
// row, so (n, ho, wo0) of the chunk is wave-uniform and tracked on the scalar unit; every gather offset
// is "uniform chunk base + per-thread constant" -- one vector add per load, fetched with raw buffer loads
// whose range check supplies the zeros (no exec branches, no 64-bit address math).  With two waves per
// SIMD the partner's fp32 MFMAs starve the vector ALU, so the address work is what was limiting.
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
constexpr int WOOB = 0x7FFFFFF0;
constexpr unsigned WRSRC_FLAGS = 0x00020000;

__device__ __forceinline__ float4 as_f4w(u32x4w v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// WINO (3x3 stride-1 pad-1, 128 x 128 tiles, I % 128 == 0, Wo % 64 == 0): the same sum as F(2,3) Winograd along the image
// rows (csrc/vq2_wino.hip has the forward form).  The reduction runs over column PAIRS (P.M, P.rows_per_split, P.Wo are
// in pairs... P.Wo stays in pixels), the K axis is (kernel row kh, Winograd index v, channel): for pair t of a row with
// input columns d0..d3 = 2t-1 .. 2t+2 and gradients g0, g1 of its two pixels
//     S_v[kh] += E_v^T V_v,   E = (g0, g0 + g1, g0 - g1, g1),   V = (d0 - d2, d1 + d2, d2 - d1, d3 - d1)
// and the reduction kernels finish with  dw[kh][0] = S0 + (S1 + S2)/2, dw[kh][1] = (S1 - S2)/2, dw[kh][2] = (S1 + S2)/2 + S3:
// 12 products per pair instead of 18 -- 2/3 of the matrix instructions.  A workgroup owns ONE (kh, v, 128-channel block):
// v is workgroup-uniform, so the transform is one fused multiply-add by +-1 per staged element (exact).
// WINO == 2 (4x4 stride-2 pad-1 conv and conv-transpose, I == 64 or I % 128 == 0): the four taps of a kernel row split by
// column parity into two 2-tap filters (csrc/vq2_wino.hip, wino_k4s2_kernel), each as F(2,2) over output column pairs:
//     S1 += g0^T (d0 - d1),  S2 += (g0 + g1)^T d1,  S3 += g1^T (d1 - d2);   dw[first tap] = S1 + S2, dw[second] = S2 - S3
// with d0..d2 = input columns 4t+pc, 4t+pc+2, 4t+pc+4 (pc = -1: taps 0, 2; pc = 0: taps 1, 3): 3/4 of the matrix
// instructions.  K axis = (v, parity, kh, channel), kh fastest: a 128-wide tile holds two kernel rows of ONE (v, parity).
// WINO == 3 (the exchanged-roles form of a 3x3 conv with 32 output channels -- the ResBlock 3x3 -- on 128 x 96 tiles): the
// STREAMED operand is x and carries V_v (two pixels of the row, with column checks), the gathered operand is dy and carries
// E_v at the three row shifts 1 - kh: S_v[kh] += V_v[r]^T E_v[r - kh + 1].  K axis = (v, kh, co); a tile is one v.
template <int WAVES_M, int WAVES_N, int MT, int NT, bool RELU_X, bool RELU_G, int WINO = 0>
__global__ __launch_bounds__(256) void wgrad_fast_kernel(const WgradParams P) {
    constexpr int BMO = WAVES_M * MT * 32;
    constexpr int BNK = WAVES_N * NT * 32;
    constexpr int G_C4 = BMO / 4, X_C4 = BNK / 4;
    constexpr int G_RSTEP = rows_per_pass(G_C4), X_RSTEP = rows_per_pass(X_C4);
    constexpr int G_LD = WG_BKR / G_RSTEP, X_LD = WG_BKR / X_RSTEP;
    static_assert(G_C4 * G_RSTEP == 256, "every thread stages G");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Gs = smem;
    float *Xs = smem + 2 * WG_BKR * BMO;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ktiles = (P.K + BNK - 1) / BNK;
    const int otiles = (P.O + BMO - 1) / BMO;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int k0 = (vid % ktiles) * BNK;
    const int o0 = ((vid / ktiles) % otiles) * BMO;
    const int split = vid / (ktiles * otiles);
    const int m_begin = split * P.rows_per_split;
    const int m_end = min(P.M, m_begin + P.rows_per_split);

    const __amdgpu_buffer_rsrc_t rg =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.g), 0, (WINO ? 2 : 1) * P.M * P.ldg * 4, WRSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.N * P.H * P.W * P.ldx * 4, WRSRC_FLAGS);

    // per-thread constants
    const int g_c = o0 + (tid % G_C4) * 4;
    const int g_r = tid / G_C4;
    const int x_k = k0 + (tid % X_C4) * 4;
    const int x_r = tid / X_C4;
    const bool g_cv = g_c < P.O;
    const bool x_act = tid < X_C4 * X_RSTEP;       // threads beyond the last full staging row sit X out
    const bool x_kv = x_act && x_k < P.K;
    int kh = 0, kw = 0, ci = 0;
    if (x_kv) {
        const int tap = x_k / P.I;
        ci = x_k - tap * P.I;
        kh = tap / P.KW;
        kw = tap - kh * P.KW;
    }
    // WINO: the K tile is one (kh, v) of all threads (I % BNK == 0); column offsets of the two input pixels of V_v and of
    // the gradient pixel(s) of E_v, relative to the pair's first pixel
    int nu = 0, xo_a = 0, xo_b = 0, go_a = 0;
    bool g_two = false;
    float sv = -1.f, sg = 1.f;
    bool x_two = true, bias2 = false, bias2u = false;                                  // WINO == 2: V of the middle product is one pixel
    int par = 0;
    if constexpr (WINO == 1) {
        const int vt = k0 / P.I;
        kh = vt >> 2; nu = vt & 3; kw = 0;
        ci = x_k - vt * P.I;
        xo_a = (nu == 0) ? -1 : (nu == 1) ? 0 : (nu == 2) ? 1 : 2;     // V = x[a] + sv * x[b]
        xo_b = (nu == 0) ? 1 : (nu == 1) ? 1 : 0;
        sv = (nu == 1) ? 1.f : -1.f;
        go_a = (nu == 3) ? 1 : 0;                                       // E = g[a] (+ sg * g[1])
        g_two = (nu == 1 || nu == 2);
        sg = (nu == 2) ? -1.f : 1.f;
    }
    if constexpr (WINO == 2) {
        const int vt = x_k / P.I;                                       // per thread: the tile may hold two kernel rows
        const int vt0 = k0 / P.I;
        kh = vt & 3; par = (vt0 >> 2) & 1; nu = vt0 >> 3; kw = 0;       // nu, par: workgroup-uniform
        ci = x_k - vt * P.I;
        const int pc = par ? 0 : -1;
        xo_a = (nu == 0) ? pc : pc + 2;
        xo_b = (nu == 0) ? pc + 2 : pc + 4;
        x_two = nu != 1;
        sv = -1.f;
        go_a = (nu == 2) ? 1 : 0;
        g_two = nu == 1;
        sg = 1.f;
        // conv-transpose: the bias gradient is the column sum of the GATHERED operand (dy).  The middle-product tiles read
        // d1 = columns 4t+1 (odd parity) / 4t+2 (even) of rows 2ho (kh 1) / 2ho+1 (kh 2); with the spare second load on
        // columns 4t+3 / 4t they visit every dy pixel exactly once.
        bias2u = P.bias_ws != nullptr && P.bias_taps != 0 && nu == 1 && o0 == 0;   // workgroup-uniform: every thread of
        bias2 = bias2u && (kh == 1 || kh == 2) && x_kv;                            // such a tile loads and adds, two kernel
        if (nu == 1) xo_b = par ? 0 : 3;                                           // rows' threads keep their sums
    }
    int go_b = 1;                                                      // WINO == 3: second pixel of V on the streamed side
    if constexpr (WINO == 3) {
        nu = k0 / BNK;                                                  // workgroup-uniform (one v per 96-wide tile)
        const int kl = x_k - k0;
        kh = kl / P.I; kw = 0; ci = kl - kh * P.I;                      // kh: per thread (three kernel rows per tile)
        go_a = (nu == 0) ? -1 : (nu == 1) ? 0 : (nu == 2) ? 1 : 2;     // V = x[a] + sg * x[b]
        go_b = (nu == 0) ? 1 : (nu == 1) ? 1 : 0;
        g_two = true;
        sg = (nu == 1) ? 1.f : -1.f;
        xo_a = (nu == 3) ? 1 : 0;                                       // E = dy[a] (+ sv * dy[1])
        xo_b = 1;
        x_two = (nu == 1 || nu == 2);
        sv = (nu == 2) ? -1.f : 1.f;
    }
    // WINO on rows of 32 pixels (16 pairs): a 32-pair chunk is TWO image rows.  Staging rows 0..15 are the first, 16..31 the
    // second (G_RSTEP = X_RSTEP = 8, so load j belongs to row j >> 1: a compile-time property of the load).
    const bool two_row = WINO != 0 && P.Wo == 32;
    int g_const[G_LD], x_const[X_LD], x_iw[X_LD];
#pragma unroll
    for (int j = 0; j < G_LD; ++j) {
        const int rj = g_r + j * G_RSTEP, r2 = two_row ? rj >> 4 : 0, tj = rj - 16 * r2;
        g_const[j] = (WINO ? (r2 * P.Wo + 2 * tj) * P.ldg + g_c : rj * P.ldg + g_c) * 4;
    }
#pragma unroll
    for (int j = 0; j < X_LD; ++j) {
        const int rj = x_r + j * X_RSTEP;                    // row inside the 32-row chunk
        if constexpr (WINO) {
            constexpr int CS = WINO == 2 ? 4 : 2;            // input columns per pair
            constexpr int RS = WINO == 2 ? 2 : 1;            // input rows per row of the G grid
            const int r2 = two_row ? rj >> 4 : 0, tj = rj - 16 * r2;
            x_iw[j] = CS * tj;                               // + CS * pair0 + xo = input column
            x_const[j] = (((WINO == 3 ? -kh : kh) + RS * r2) * P.W + CS * tj) * P.ldx * 4 + ci * 4;
        } else {
        x_iw[j] = rj * P.stride - P.pad + kw;                // + wo0*stride = input column
        x_const[j] = ((kh * P.W + kw + rj * P.stride) * P.ldx + ci) * 4;
        }
    }
    // wave-uniform chunk position (scalar registers)
    int un, uho, uwo;
    const int row_len = WINO ? P.Wo / 2 : P.Wo;              // reduction positions per image row (WINO: column pairs)
    {
        const int HoWo = P.Ho * row_len;
        un = m_begin / HoWo;
        const int r = m_begin - un * HoWo;
        uho = r / row_len;
        uwo = r - uho * row_len;
    }

    u32x4w rgv[G_LD], rxv[X_LD];
    u32x4w rgw[WINO ? G_LD : 1], rxw[WINO ? X_LD : 1];       // WINO: the second pixel of every staged element
    float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);          // WINO == 2, conv-transpose: bias partial of this thread's 4 channels
    auto load_chunk_wino = [&](int mbase) {
        const int rows_left = m_end - mbase;                  // uniform (pairs)
        const int gbase = ((un * P.Ho + uho) * P.Wo + 2 * uwo + go_a) * P.ldg * 4;   // uniform
#pragma unroll
        for (int j = 0; j < G_LD; ++j) {
            const bool v = g_cv && (g_r + j * G_RSTEP) < rows_left;
            if constexpr (WINO == 3) {                         // (gbase carries go_a; the second pixel is go_b - go_a further)
                const int col = 2 * (uwo + ((g_r + j * G_RSTEP) & (two_row ? 15 : 31)));
                const bool va = v && (unsigned)(col + go_a) < (unsigned)P.Wo;
                const bool vb = v && (unsigned)(col + go_b) < (unsigned)P.Wo;
                rgv[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, va ? gbase + g_const[j] : WOOB, 0, 0);
                rgw[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, vb ? gbase + g_const[j] + (go_b - go_a) * P.ldg * 4 : WOOB, 0, 0);
            } else {
            rgv[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, v ? gbase + g_const[j] : WOOB, 0, 0);
            rgw[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, (v && g_two) ? gbase + g_const[j] + P.ldg * 4 : WOOB, 0, 0);
            }
        }
        constexpr int CS = WINO == 2 ? 4 : 2;
        const int ihu = WINO == 3 ? uho + 1 : (WINO == 2 ? 2 * uho : uho) - 1;   // pad 1 (WINO 3: row shift 1 - kh)
        const int xbase = ((un * P.H + ihu) * P.W + CS * uwo) * P.ldx * 4;            // uniform
        constexpr int RS = WINO == 2 ? 2 : 1;
        const int ihk = ihu + (WINO == 3 ? -kh : kh);
        const bool hv0 = x_kv && (unsigned)ihk < (unsigned)P.H;                       // first row of the chunk
        const bool hv1 = two_row ? x_kv && (unsigned)(ihk + RS) < (unsigned)P.H : hv0;  // second row (two-row chunks)
        const int iwu = CS * uwo;
        const int xa = xo_a * P.ldx * 4, xb = xo_b * P.ldx * 4;
        const bool want_b = x_two || bias2u;
#pragma unroll
        for (int j = 0; j < X_LD; ++j) {
            const bool v = ((x_r + j * X_RSTEP) >= 16 ? hv1 : hv0) && (x_r + j * X_RSTEP) < rows_left;
            const bool va = v && (unsigned)(iwu + x_iw[j] + xo_a) < (unsigned)P.W;
            const bool vb = v && want_b && (unsigned)(iwu + x_iw[j] + xo_b) < (unsigned)P.W;
            rxv[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, va ? xbase + x_const[j] + xa : WOOB, 0, 0);
            rxw[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, vb ? xbase + x_const[j] + xb : WOOB, 0, 0);
        }
        uwo += WG_BKR;
        if (uwo >= row_len) {
            uwo = 0;
            uho += two_row ? 2 : 1;
            if (uho >= P.Ho) { uho = 0; ++un; }
        }
    };
    auto load_chunk = [&](int mbase) {
        if constexpr (WINO) { load_chunk_wino(mbase); return; }
        const int rows_left = m_end - mbase;                  // uniform
        const int gbase = mbase * P.ldg * 4;                  // uniform
#pragma unroll
        for (int j = 0; j < G_LD; ++j) {
            const bool v = g_cv && (g_r + j * G_RSTEP) < rows_left;
            rgv[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, v ? gbase + g_const[j] : WOOB, 0, 0);
        }
        const int ihu = uho * P.stride - P.pad;               // uniform
        const int xbase = ((un * P.H + ihu) * P.W + uwo * P.stride - P.pad) * P.ldx * 4;  // uniform
        const bool hv = x_kv && (unsigned)(ihu + kh) < (unsigned)P.H;
        const int iwu = uwo * P.stride;
#pragma unroll
        for (int j = 0; j < X_LD; ++j) {
            const bool v = hv && (x_r + j * X_RSTEP) < rows_left && (unsigned)(iwu + x_iw[j]) < (unsigned)P.W;
            rxv[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, v ? xbase + x_const[j] : WOOB, 0, 0);
        }
        // next chunk: 32 rows further along the same output row, or the start of the next one
        uwo += WG_BKR;
        if (uwo >= P.Wo) {
            uwo = 0;
            if (++uho == P.Ho) { uho = 0; ++un; }
        }
    };
    auto store_chunk = [&](int buf) {
        float *gs = Gs + buf * WG_BKR * BMO;
        float *xs = Xs + buf * WG_BKR * BNK;
#pragma unroll
        for (int j = 0; j < G_LD; ++j) {
            float4 v = as_f4w(rgv[j]);
            if (RELU_G) v = relu4(v);
            if constexpr (WINO) {
                if (g_two) {                                   // workgroup-uniform
                    float4 w = as_f4w(rgw[j]);
                    if (RELU_G) w = relu4(w);
                    v = make_float4(fmaf(sg, w.x, v.x), fmaf(sg, w.y, v.y), fmaf(sg, w.z, v.z), fmaf(sg, w.w, v.w));
                }
            }
            *reinterpret_cast<float4 *>(gs + (g_r + j * G_RSTEP) * BMO + (tid % G_C4) * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < X_LD; ++j) {
            float4 v = as_f4w(rxv[j]);
            if constexpr (WINO) {
                float4 w = as_f4w(rxw[j]);
                if (RELU_X) { v = relu4(v); w = relu4(w); }
                if (WINO == 2 && bias2u) {                     // (rows past the split and out-of-range pixels were read as 0)
                    bsum4.x += v.x + w.x; bsum4.y += v.y + w.y; bsum4.z += v.z + w.z; bsum4.w += v.w + w.w;
                }
                if (WINO == 1 || x_two)                        // x_two: workgroup-uniform
                    v = make_float4(fmaf(sv, w.x, v.x), fmaf(sv, w.y, v.y), fmaf(sv, w.z, v.z), fmaf(sv, w.w, v.w));
                if (X_C4 * X_RSTEP == 256 || x_act)
                    *reinterpret_cast<float4 *>(xs + (x_r + j * X_RSTEP) * BNK + (tid % X_C4) * 4) = v;
            } else {
            if (X_C4 * X_RSTEP == 256 || x_act)
                *reinterpret_cast<float4 *>(xs + (x_r + j * X_RSTEP) * BNK + (tid % X_C4) * 4) = RELU_X ? relu4(v) : v;
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (m_end - m_begin + WG_BKR - 1) / WG_BKR;
    if (nchunks > 0) {
        load_chunk(m_begin);
        store_chunk(0);
    }
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5;
    // (WINO: the tile (kh 0, v 1) stages E = g0 + g1 -- its column sums over the pairs are the sums over the pixels)
    // (WINO == 2: the tile (v 1, parity 0, kh 0..1) likewise)
    const bool do_bias = P.bias_ws != nullptr && !P.bias_taps && k0 == (WINO == 2 ? 8 * P.I : WINO == 1 ? P.I : 0) && tid < BMO;
    // exchanged roles: dy is the gathered operand; its centre tap visits every pixel exactly once, so the
    // column sums of that k-block of the staged X tile are the bias gradient (tile width == one tap)
    int bx_col = -1;   // this thread's slot in a bias_ws row when its k column belongs to a bias tap
    if (WINO == 3 && P.bias_ws != nullptr && P.bias_taps && o0 == 0 && k0 == BNK && tid >= P.I && tid < 2 * P.I)
        bx_col = tid - P.I;        // tile v = 1 stages E = dy0 + dy1; its kh = 1 columns (row shift 0) see every dy row once
    if (WINO == 0 && P.bias_ws != nullptr && P.bias_taps && o0 == 0 && tid < BNK && k0 + tid < P.K) {
        const int tap = (k0 + tid) / P.I;
        if ((P.bias_taps >> tap) & 1) bx_col = __popc(P.bias_taps & ((1u << tap) - 1u)) * P.I + (k0 + tid - tap * P.I);
    }
    const bool do_bias_x = bx_col >= 0;
    float bsum = 0.f;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(m_begin + (c + 1) * WG_BKR);
        if (do_bias) {
            const float *gcol = Gs + buf * WG_BKR * BMO + tid;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < WG_BKR; r += 2) { s0 += gcol[r * BMO]; s1 += gcol[(r + 1) * BMO]; }
            bsum += s0 + s1;
        }
        if (do_bias_x) {
            const float *xcol = Xs + buf * WG_BKR * BNK + tid;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < WG_BKR; r += 2) { s0 += xcol[r * BNK]; s1 += xcol[(r + 1) * BNK]; }
            bsum += s0 + s1;
        }
        // a wave's 32-wide blocks are INTERLEAVED with the other waves' (block i of wave wm = i*WAVES_M + wm): the two
        // fragments of a k-step are then 256 bytes apart and k-steps 1 KiB apart -- one ds_read2st64_b32 with immediate
        // offsets per operand and step, no vector add to rebuild a base (vector instructions cost matrix time)
        const float *gs = Gs + buf * WG_BKR * BMO + fk * BMO + wm * 32 + fr;
        const float *xs = Xs + buf * WG_BKR * BNK + fk * BNK + wn * 32 + fr;
        // fragments double-buffered: the ds_reads of k-step kk+1 are in flight during the MFMAs of step kk
        float fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = gs[i * WAVES_M * 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = xs[j * WAVES_N * 32];
#pragma unroll
        for (int kk = 0; kk < WG_BKR / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < WG_BKR / 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[nxt][i] = gs[(kk + 1) * 2 * BMO + i * WAVES_M * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[nxt][j] = xs[(kk + 1) * 2 * BNK + j * WAVES_N * 32];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
            // pin the order: next step's LDS reads first, then this step's MFMAs (hipcc otherwise sinks the
            // reads to just before their use and waits lgkmcnt(0) in front of every MFMA group)
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);
        }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    if (do_bias && o0 + tid < P.O) P.bias_ws[(size_t)split * P.O + o0 + tid] = bsum;
    if (do_bias_x) P.bias_ws[(size_t)split * P.bias_nslots * P.I + bx_col] = bsum;
    if constexpr (WINO == 2) {
        // the eight staging row groups meet in LDS (fixed order); 4 slots per split: (parity, kernel row 1 | 2) -- the
        // reduction kernel adds all S * 4 of them (a long slot list is a long chain of dependent loads there)
        if (bias2u) {                                      // workgroup-uniform
            float4 *red = reinterpret_cast<float4 *>(Xs);  // (the main loop ended with a barrier: the tiles are free)
            red[tid] = bsum4;
            __syncthreads();
            if (bias2 && x_r == 0) {
                float4 t = red[tid];
#pragma unroll
                for (int r = 1; r < 8; ++r) {
                    const float4 u = red[r * X_C4 + tid];
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                const int slot = par * 2 + (kh - 1);
                *reinterpret_cast<float4 *>(P.bias_ws + ((size_t)split * 4 + slot) * P.I + ci) = t;
            }
        }
    }

    float *slab = P.ws + (size_t)split * P.O * P.K;
    const int colq = lane & 31, rowq = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int k = k0 + (j * WAVES_N + wn) * 32 + colq;
        if (k >= P.K) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int ob = o0 + (i * WAVES_M + wm) * 32 + rowq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ob + (r & 3) + 8 * (r >> 2);
                if (o < P.O) slab[(size_t)o * P.K + k] = acc[i][j][r];
            }
        }
    }
}

// Winograd slabs (wgrad_fast_kernel<..., WINO>) -> reference layout.  mode 1: [S][O][(kh 3, v 4, i)], one thread column per
// (o, kh, i), three taps out; mode 2: [S][O][(v 3, parity 2, kh 4, i)], one column per (o, parity, kh, i), two taps out;
// mode 3: [S][O = ci][(v 4, kh 3, i = co)], as mode 1 into the transposed destination.
// Same fixed order as the plain reduction: split lane g adds splits g, g+8, ...; the 8 partials are added in lane order.
__device__ __forceinline__ void wino_reduce_unit(int mode, int t, int lane, int g, const float *__restrict__ ws,
                                                 float *__restrict__ dw, int O, int I, int Or, int Ir, int S,
                                                 float (*part4)[8][33]) {
    const int nv = mode == 2 ? 3 : 4, ntap = mode == 2 ? 16 : 9;
    const int KW_ = (mode == 2 ? 24 : 12) * I, stride = O * KW_;
    const int per_o = (mode == 2 ? 8 : 3) * I, cols = O * per_o;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int o = 0, q = 0, i = 0;     // q = kh (mode 1) or parity * 4 + kh (mode 2)
    if (t < cols) {
        o = t / per_o;
        const int r = t - o * per_o;
        q = r / I; i = r - q * I;
        // column of v: mode 1 (kh * 4 + v) * I + i; mode 2 ((v * 2 + parity) * 4 + kh) * I + i = (v * 8 + q) * I + i
        // mode 3 (exchanged roles, slab rows = input channels): (v * 3 + kh) * I + i
        const float *src = ws + (size_t)o * KW_ + (mode == 1 ? q * 4 * I : q * I) + i;
        const int vstep = mode == 1 ? I : (mode == 2 ? 8 : 3) * I;
        for (int z = g; z < S; z += 8) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (v < nv) s[v] += src[(size_t)z * stride + v * vstep];
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) part4[v][g][lane] = s[v];
    __syncthreads();
    if (g == 0 && t < cols && o < Or && i < Ir) {
        float r4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float a = part4[v][0][lane];
#pragma unroll
            for (int qq = 1; qq < 8; ++qq) a += part4[v][qq][lane];
            r4[v] = a;
        }
        float *dst = dw + (mode == 3 ? (size_t)i * Or + o : (size_t)o * Ir + i) * ntap;
        if (mode != 2) {
            const float h = 0.5f * (r4[1] + r4[2]);
            dst[q * 3 + 0] = r4[0] + h;
            dst[q * 3 + 1] = 0.5f * (r4[1] - r4[2]);
            dst[q * 3 + 2] = h + r4[3];
        } else {
            const int par = q >> 2, kh = q & 3;
            dst[kh * 4 + par] = r4[0] + r4[1];          // taps 0 / 1
            dst[kh * 4 + par + 2] = r4[1] - r4[2];      // taps 2 / 3
        }
    }
    __syncthreads();
}

// Sum the slabs and scatter [o][tap][i] (padded O x I) -> reference layout [Or][Ir][tap].
// 256 threads = 32 outputs x 8 split-lanes: lane g adds splits g, g+8, ... (4 loads in flight), the 8
// partial sums are combined through LDS in a fixed order -> bit-reproducible, and latency is paid
// S/32 times instead of S times.  The last workgroup also folds the bias partials.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, int O,
                                                           int I, int Or, int Ir, int taps, int S,
                                                           const float *__restrict__ bias_ws, float *__restrict__ db,
                                                           int swapped, int bias_splits) {
    __shared__ float part[8][33];
    const int lane = threadIdx.x & 31, g = threadIdx.x >> 5;
    __shared__ float part4[4][8][33];
    const int wmode = (swapped >> 1) & 3;     // 0 plain, 1 F(2,3) rows, 2 F(2,2) by parity
    if (wmode) {
        const int cols = O * (wmode == 2 ? 8 : 3) * I;
        for (int base = blockIdx.x * 32; base < cols; base += gridDim.x * 32)
            wino_reduce_unit(wmode, base + lane, lane, g, ws, dw, O, I, Or, Ir, S, part4);
    }
    const int K = taps * I;
    const int total = wmode ? 0 : O * K;
    swapped = wmode ? 0 : (swapped & 1);
    for (int base = blockIdx.x * 32; base < total; base += gridDim.x * 32) {
        const int t = base + lane;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (t < total) {
            int z = g;
            for (; z + 24 < S; z += 32) {
                s0 += ws[(size_t)z * total + t];
                s1 += ws[(size_t)(z + 8) * total + t];
                s2 += ws[(size_t)(z + 16) * total + t];
                s3 += ws[(size_t)(z + 24) * total + t];
            }
            for (; z < S; z += 8) s0 += ws[(size_t)z * total + t];
        }
        part[g][lane] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (g == 0 && t < total) {
            float s = part[0][lane];
#pragma unroll
            for (int q = 1; q < 8; ++q) s += part[q][lane];
            const int o = t / K, k = t - o * K;
            const int tap = k / I, i = k - tap * I;
            if (o < Or && i < Ir) {
                if (swapped) dw[((size_t)i * Or + o) * taps + (taps - 1 - tap)] = s;   // slab is [ci][flipped tap][co]
                else dw[((size_t)o * Ir + i) * taps + tap] = s;
            }
        }
        __syncthreads();
    }
    if (db && bias_ws && blockIdx.x == gridDim.x - 1) {
        // bias_splits > 0: partials came from the gathered operand, [bias_splits][I]; else from the G tile, [S][O]
        const int bstride = bias_splits ? I : O, blimit = bias_splits ? Ir : Or, bcount = bias_splits ? bias_splits : S;
        for (int base = 0; base < blimit; base += 32) {
            const int o = base + lane;
            float s0 = 0.f;
            if (o < blimit)
                for (int z = g; z < bcount; z += 8) s0 += bias_ws[(size_t)z * bstride + o];
            part[g][lane] = s0;
            __syncthreads();
            if (g == 0 && o < blimit) {
                float s = part[0][lane];
#pragma unroll
                for (int q = 1; q < 8; ++q) s += part[q][lane];
                db[o] = s;
            }
            __syncthreads();
        }
    }
}

static int colsum_impl(const float *dy, int64_t rows, int C, int ld, float *db, int nout, float *ws, hipStream_t s);

// All layers' slab reductions in ONE launch (the per-layer reductions are latency-bound and tiny).
// Index space = "units" of 32 consecutive outputs; a job owns [unit_offset, unit_offset + n_units_w +
// n_units_b): first the weight units, then the bias units.  Same 32 x 8 fixed-order scheme as above.
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const vq2_wgrad_job *__restrict__ jobs, int njobs,
                                                                   int64_t total_units) {
    __shared__ float part[8][33];
    __shared__ float part4[4][8][33];
    const int lane = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (int64_t u = blockIdx.x; u < total_units; u += gridDim.x) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].unit_offset <= u) lo = mid; else hi = mid - 1;
        }
        const vq2_wgrad_job j = jobs[lo];
        const int lu = (int)(u - j.unit_offset);
        const bool is_bias = lu >= j.n_units_w;
        if ((j.swapped & 6) && !is_bias) {   // Winograd slabs: unit = 32 thread columns (wino_reduce_unit)
            wino_reduce_unit((j.swapped >> 1) & 3, lu * 32 + lane, lane, g, j.ws, j.dw, j.O, j.I, j.Or, j.Ir, j.S, part4);
            continue;
        }
        const int K = j.taps * j.I;
        const int total = is_bias ? (j.bias_splits ? j.I : j.O) : j.O * K;      // stride between splits
        const int limit = is_bias ? (j.bias_splits ? j.Ir : j.Or) : j.O * K;
        const int nsplit = (is_bias && j.bias_splits) ? j.bias_splits : j.S;
        const float *src = is_bias ? j.bias_ws : j.ws;
        const int t = (is_bias ? lu - j.n_units_w : lu) * 32 + lane;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (t < limit) {
            int z = g;
            for (; z + 24 < nsplit; z += 32) {
                s0 += src[(size_t)z * total + t];
                s1 += src[(size_t)(z + 8) * total + t];
                s2 += src[(size_t)(z + 16) * total + t];
                s3 += src[(size_t)(z + 24) * total + t];
            }
            for (; z < nsplit; z += 8) s0 += src[(size_t)z * total + t];
        }
        part[g][lane] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (g == 0 && t < limit) {
            float s = part[0][lane];
#pragma unroll
            for (int q = 1; q < 8; ++q) s += part[q][lane];
            if (is_bias) {
                j.db[t] = s;
            } else {
                const int o = t / K, k = t - o * K;
                const int tap = k / j.I, i = k - tap * j.I;
                if (o < j.Or && i < j.Ir) {
                    if (j.swapped & 1) j.dw[((size_t)i * j.Or + o) * j.taps + (j.taps - 1 - tap)] = s;
                    else j.dw[((size_t)o * j.Ir + i) * j.taps + tap] = s;
                }
            }
        }
        __syncthreads();
    }
}

struct WgradPlan {
    int O, I, K, M, S, rows_per_split, bmo, bnk;
    int swapped;  // roles of x and dy exchanged (see plan_wgrad)
    int wino;     // F(2,3) Winograd along the rows: K = 12 * I (kh, v, i), M in column pairs (wgrad_fast_kernel<..., WINO>)
};

static WgradPlan plan_wgrad(const vq2_conv_desc *d) {
    WgradPlan p;
    p.swapped = 0;
    // A stride-1 "same" conv with few output channels (the ResBlock 3x3, 128 -> 32): gathering the im2col
    // of the WIDE tensor x re-reads it KH*KW times through L2.  The same sum with the roles exchanged,
    //   dw[co][ci][kh][kw] = sum_q relu(x)[q][ci] * dy[q - (kh-p, kw-p)][co],
    // gathers the NARROW tensor dy instead (flipped taps, pad' = K-1-p) and streams x once.
    static const int wswap = getenv("VQ2_WSWAP") ? atoi(getenv("VQ2_WSWAP")) : 1;
    if (wswap && !d->transposed && d->stride == 1 && 2 * d->pad == d->KH - 1 && d->KH > 1 && d->Co <= 32 && d->Ci >= 64) {
        p.swapped = 1;
        p.O = d->Ci; p.I = d->Co;
        p.M = d->N * d->H * d->W;
    } else if (!d->transposed) {
        p.O = d->Co; p.I = d->Ci;
        const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
        p.M = d->N * Ho * Wo;
    } else {
        p.O = d->Ci; p.I = d->Co;
        p.M = d->N * d->H * d->W;
    }
    p.K = d->KH * d->KW * p.I;
    p.wino = 0;
    static const int wwino3 = getenv("VQ2_WWINO_SW") ? atoi(getenv("VQ2_WWINO_SW")) : 1;
    static const int wwino = getenv("VQ2_WWINO") ? atoi(getenv("VQ2_WWINO")) : 1;
    static const int wfast = getenv("VQ2_WFAST") ? atoi(getenv("VQ2_WFAST")) : 1;
    if (wwino && wfast && !p.swapped && !d->transposed && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 &&
        p.O % 128 == 0 && p.I % 128 == 0 && (d->W % 64 == 0 || (d->W == 32 && d->H % 2 == 0)) &&
        (long)d->N * d->H * d->W * d->ldx < (1L << 29) &&
        (long)d->N * d->H * d->W * d->ldy < (1L << 29)) {
        p.wino = 1;
        p.K = 12 * p.I;
        p.M = p.M / 2;
    }
    static const int wwino4 = getenv("VQ2_WWINO_K4") ? atoi(getenv("VQ2_WWINO_K4")) : 1;
    {   // 4x4 stride-2 conv / conv-transpose: F(2,2) by column parity over output column pairs
        const int wo = d->transposed ? d->W : d->W / 2;       // width of the G operand's grid
        if (wwino4 && wfast && !p.swapped && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && d->H % 2 == 0 &&
            d->W % 2 == 0 && p.O % 128 == 0 && (p.I == 64 || p.I % 128 == 0) &&
            (wo % 64 == 0 || (wo == 32 && (d->transposed ? d->H : d->H / 2) % 2 == 0)) &&
            (long)d->N * d->H * d->W * d->ldx < (1L << 29) &&
            (long)d->N * d->H * d->W * d->ldy * (d->transposed ? 4 : 1) < (1L << 29)) {
            p.wino = 2;
            p.K = 24 * p.I;
            p.M = p.M / 2;
        }
    }
    // output tile (o x k): the candidate that wastes the least padded work, larger tile on ties
    // 128 x 96 = one kernel row of a 3x3 conv with 32 gathered channels (the ResBlock weight gradient with
    // exchanged roles, K = 288): 48 MFMAs per wave and chunk instead of 16 for the same staging
    static const int cand[6][2] = {{128, 128}, {64, 128}, {32, 256}, {128, 96}, {128, 32}, {64, 64}};
    static const int t96 = getenv("VQ2_W96") ? atoi(getenv("VQ2_W96")) : 1;
    long best = -1;
    for (int c = 0; c < (p.wino ? 1 : 6); ++c) {
        if (cand[c][1] == 96 && (!t96 || p.M < 65536)) continue;   // measured: +6 % at 131072 rows, -9 % at 32768
        const long po = (p.O + cand[c][0] - 1) / cand[c][0] * cand[c][0];
        const long pk = (p.K + cand[c][1] - 1) / cand[c][1] * cand[c][1];
        const long work = po * pk;
        if (best < 0 || work < best) { best = work; p.bmo = cand[c][0]; p.bnk = cand[c][1]; }
    }
    if (wwino3 && wfast && p.swapped && d->KH == 3 && d->pad == 1 && p.I == 32 && p.O % 128 == 0 &&
        (d->W % 64 == 0 || (d->W == 32 && d->H % 2 == 0)) &&
        (long)d->N * d->H * d->W * d->ldx < (1L << 29) && (long)d->N * d->H * d->W * d->ldy < (1L << 29)) {
        p.wino = 3;                 // exchanged roles + F(2,3): four 128 x 96 tiles (v) of three kernel rows each
        p.K = 4 * 96;
        p.M = p.M / 2;
        p.bmo = 128; p.bnk = 96;
    }
    const int tiles = ((p.K + p.bnk - 1) / p.bnk) * ((p.O + p.bmo - 1) / p.bmo);
    const int lds = 2 * WG_BKR * (p.bmo + p.bnk) * 4;
    int per_cu = (160 * 1024) / lds;                 // resident workgroups per CU (LDS-bound), at most 4
    if (per_cu > 4) per_cu = 4;
    int S = (256 * per_cu) / tiles;                  // fill the resident slots without a tail round
    const int max_s = (p.M + 255) / 256;             // at least 8 chunks of 32 rows per split
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    int rps = (p.M + S - 1) / S;
    rps = (rps + 31) / 32 * 32;
    p.S = (p.M + rps - 1) / rps;
    p.rows_per_split = rps;
    return p;
}

// Which taps of the gathered operand cover every dy pixel exactly once (bias gradient = their column sums):
// exchanged roles -> the centre tap (its own flip); conv-transpose k4 s2 p1 -> taps (1,1),(1,2),(2,1),(2,2),
// one per output phase, never out of bounds.
static void bias_taps_of(const vq2_conv_desc *d, const WgradPlan &p, int &taps, int &nslots) {
    taps = 0; nslots = 0;
    if (p.wino == 2 && d->transposed) { taps = 1; nslots = 4; return; }    // flag only: see wgrad_fast_kernel<..., 2>
    if (p.swapped) { taps = 1 << ((d->KH / 2) * d->KW + d->KW / 2); nslots = 1; }
    else if (d->transposed) { taps = (1 << 5) | (1 << 6) | (1 << 9) | (1 << 10); nslots = 4; }
}

static size_t bias_ws_floats(const vq2_conv_desc *d, const WgradPlan &p) {
    int taps, nslots;
    bias_taps_of(d, p, taps, nslots);
    return taps ? (size_t)p.S * nslots * p.I : (size_t)p.S * p.O;
}

template <int WAVES_M, int WAVES_N, int MT, int NT>
static int launch_wgrad(const WgradParams &P, int S, hipStream_t s) {
    constexpr int BMO = WAVES_M * MT * 32, BNK = WAVES_N * NT * 32;
    const size_t lds = (size_t)2 * WG_BKR * (BMO + BNK) * sizeof(float);
    static const int fast = getenv("VQ2_WFAST") ? atoi(getenv("VQ2_WFAST")) : 1;
    const long lim = 1L << 29;
    const bool fast_ok = fast && P.Wo % WG_BKR == 0 && P.rows_per_split % WG_BKR == 0 &&
                         (long)P.N * P.H * P.W * P.ldx < lim && (long)P.M * P.ldg < lim;
    auto kern = wgrad_kernel<WAVES_M, WAVES_N, MT, NT>;
    if (fast_ok) {   // ReLU flags are compile-time in the fast kernel: every vector instruction competes with the MFMAs
        if (P.relu_x) kern = wgrad_fast_kernel<WAVES_M, WAVES_N, MT, NT, true, false>;
        else if (P.relu_g) kern = wgrad_fast_kernel<WAVES_M, WAVES_N, MT, NT, false, true>;
        else kern = wgrad_fast_kernel<WAVES_M, WAVES_N, MT, NT, false, false>;
    }
    allow_big_lds(kern, lds);
    dim3 grid(((P.K + BNK - 1) / BNK) * ((P.O + BMO - 1) / BMO) * S);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, P);
    return check_launch("wgrad_kernel");
}

// ------------------------------------------------------------------ column sums (bias gradient)
constexpr int CS_MAX_BLOCKS = 512;

static inline int64_t cs_rows_per_block(int64_t rows) {
    int64_t rpb = (rows + CS_MAX_BLOCKS - 1) / CS_MAX_BLOCKS;
    return rpb < 256 ? 256 : rpb;
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, int64_t rows, int C, int ld,
                                                             int64_t rpb, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [nrg][C]
    const int C4 = C / 4;
    const int nrg = 256 / C4 > 0 ? 256 / C4 : 1;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = r0 + rpb < rows ? r0 + rpb : rows;
    for (int cbase = 0; cbase < C4; cbase += 256) {
        const int c4 = cbase + (int)(threadIdx.x % (C4 < 256 ? C4 : 256));
        const int rg = threadIdx.x / (C4 < 256 ? C4 : 256);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rg < nrg && c4 < C4) {
            for (int64_t r = r0 + rg; r < r1; r += nrg) {
                const float4 v = *reinterpret_cast<const float4 *>(x + r * ld + c4 * 4);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            *reinterpret_cast<float4 *>(red + (size_t)rg * C + c4 * 4) = a;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            if (c / 4 >= cbase && c / 4 < cbase + 256) {
                float s = 0.f;
                for (int g = 0; g < nrg; ++g) s += red[(size_t)g * C + c];
                part[(size_t)blockIdx.x * C + c] = s;
            }
        }
        __syncthreads();
    }
}

// 256 threads = 32 columns x 8 lanes over the partial blocks; fixed combine order (reproducible)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ part, int nblocks, int C, int nout,
                                                           float *__restrict__ out) {
    __shared__ float red[8][33];
    const int lane = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < nout) {
        int b = g;
        for (; b + 8 < nblocks; b += 16) { s0 += part[(size_t)b * C + c]; s1 += part[(size_t)(b + 8) * C + c]; }
        for (; b < nblocks; b += 8) s0 += part[(size_t)b * C + c];
    }
    red[g][lane] = s0 + s1;
    __syncthreads();
    if (g == 0 && c < nout) {
        float s = red[0][lane];
#pragma unroll
        for (int q = 1; q < 8; ++q) s += red[q][lane];
        out[c] = s;
    }
}

static int colsum_impl(const float *dy, int64_t rows, int C, int ld, float *db, int nout, float *ws, hipStream_t s) {
    const int64_t rpb = cs_rows_per_block(rows);
    const int64_t nb = (rows + rpb - 1) / rpb;
    const int C4 = C / 4;
    const int nrg = 256 / C4 > 0 ? 256 / C4 : 1;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nb), dim3(256), (size_t)nrg * C * sizeof(float), s, dy, rows,
                       C, ld, rpb, ws);
    if (int e = check_launch("colsum_partial_kernel")) return e;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((nout + 31) / 32), dim3(256), 0, s, ws, (int)nb, C, nout, db);
    return check_launch("colsum_final_kernel");
}

}  // namespace vq2

using namespace vq2;

extern "C" size_t vq2_conv_wgrad_workspace_bytes(const vq2_conv_desc *d) {
    if (!d || d->N <= 0 || d->Ci <= 0 || d->Co <= 0) return 0;
    const WgradPlan p = plan_wgrad(d);
    return ((size_t)p.S * p.O * p.K + bias_ws_floats(d, p)) * sizeof(float);
}

// slabs (+ bias partials); reduce == true also runs the per-layer reduction into dw/db
static int wgrad_impl(const vq2_conv_desc *d, int flags, const float *x, const float *dy, float *dw, float *db, void *ws,
                      size_t ws_bytes, vq2_stream_t stream, bool reduce) {
    VQ2_REQUIRE(d && x && dy && ws && (dw || !reduce), "conv_wgrad: null pointer");
    VQ2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ci > 0 && d->Co > 0 && d->Ci % 4 == 0 && d->Co % 4 == 0,
                "conv_wgrad: bad dims");
    VQ2_REQUIRE(d->ldx >= d->Ci && d->ldy >= d->Co && d->ldx % 4 == 0 && d->ldy % 4 == 0, "conv_wgrad: bad strides");
    VQ2_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(ws), "conv_wgrad: pointers must be 16-byte aligned");
    if (d->transposed)
        VQ2_REQUIRE(d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1, "conv_wgrad: convT must be k4 s2 p1");
    else
        VQ2_REQUIRE(d->KH == d->KW && d->KH >= 1 && d->KH <= 7 && (d->stride == 1 || d->stride == 2) && d->pad >= 0,
                    "conv_wgrad: unsupported conv geometry");
    const WgradPlan p = plan_wgrad(d);
    VQ2_REQUIRE(ws_bytes >= vq2_conv_wgrad_workspace_bytes(d), "conv_wgrad: workspace too small");
    WgradParams P{};
    P.ws = static_cast<float *>(ws);
    float *bias_ws = P.ws + (size_t)p.S * p.O * p.K;       // bias partials: [S][O], or [S][nslots][I] from the gathered operand
    bias_taps_of(d, p, P.bias_taps, P.bias_nslots);
    P.bias_ws = db ? bias_ws : nullptr;
    P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad = d->pad;
    P.O = p.O; P.I = p.I; P.K = p.K; P.M = p.M; P.rows_per_split = p.rows_per_split;
    P.N = d->N;
    if (p.swapped) {
        P.x = dy; P.ldx = d->ldy; P.H = d->H; P.W = d->W;       // gathered operand: dy (same spatial size as x)
        P.g = x; P.ldg = d->ldx;
        P.Ho = d->H; P.Wo = d->W;
        P.stride = 1; P.pad = d->KH - 1 - d->pad;
        P.relu_x = 0; P.relu_g = (flags & VQ2_RELU_IN) != 0;
    } else if (!d->transposed) {
        P.x = x; P.ldx = d->ldx; P.H = d->H; P.W = d->W;
        P.g = dy; P.ldg = d->ldy;
        P.Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1; P.Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
        P.relu_x = (flags & VQ2_RELU_IN) != 0; P.relu_g = 0;
    } else {
        P.x = dy; P.ldx = d->ldy; P.H = 2 * d->H; P.W = 2 * d->W;
        P.g = x; P.ldg = d->ldx;
        P.Ho = d->H; P.Wo = d->W;
        P.relu_x = 0; P.relu_g = (flags & VQ2_RELU_IN) != 0;
    }
    VQ2_REQUIRE((int64_t)P.N * P.H * P.W * P.ldx < ((int64_t)1 << 31) && (int64_t)P.M * P.ldg < ((int64_t)1 << 31),
                "conv_wgrad: tensor exceeds 2^31 elements");
    hipStream_t s = to_stream(stream);
    int e;
    const double cir_ = d->Cir ? d->Cir : d->Ci, cor_ = d->Cor ? d->Cor : d->Co;
    const double pix_in_ = (double)d->N * d->H * d->W;
    const double pix_out_ = d->transposed ? 4.0 * pix_in_ : (double)P.M * (p.wino ? 2.0 : 1.0);
    const double macs_ = d->transposed ? pix_in_ * 16.0 * cir_ * cor_ : pix_out_ * d->KH * d->KW * cir_ * cor_;
    const char *pname = "wgrad";
    if (prof_enabled()) pname = prof_label("wgrad<%dx%d>%s|O=%d,K=%d,M=%d,S=%d,k%d", p.bmo, p.bnk, p.wino == 3 ? "swwino" : p.swapped ? "sw" : (p.wino == 1 ? "wino" : p.wino == 2 ? "wino4" : ""), p.O, p.K, p.M, p.S, d->KH);
    ProfScope prof(pname, 2.0 * macs_, 4.0 * (pix_in_ * cir_ + pix_out_ * cor_ + cir_ * cor_ * d->KH * d->KW), s);
    if (p.wino == 3) {
        auto kern = P.relu_g ? wgrad_fast_kernel<4, 1, 1, 3, false, true, 3> : wgrad_fast_kernel<4, 1, 1, 3, false, false, 3>;
        const size_t lds = (size_t)2 * WG_BKR * (128 + 96) * sizeof(float);
        allow_big_lds(kern, lds);
        hipLaunchKernelGGL(kern, dim3(4 * (P.O / 128) * p.S), dim3(256), lds, s, P);
        e = check_launch("wgrad_fast_kernel<wino sw>");
    } else if (p.wino) {
        auto kern = P.relu_x ? wgrad_fast_kernel<2, 2, 2, 2, true, false, 1> : wgrad_fast_kernel<2, 2, 2, 2, false, false, 1>;
        if (p.wino == 2)
            kern = P.relu_x ? wgrad_fast_kernel<2, 2, 2, 2, true, false, 2>
                            : (P.relu_g ? wgrad_fast_kernel<2, 2, 2, 2, false, true, 2> : wgrad_fast_kernel<2, 2, 2, 2, false, false, 2>);
        const size_t lds = (size_t)2 * WG_BKR * 256 * sizeof(float);
        allow_big_lds(kern, lds);
        hipLaunchKernelGGL(kern, dim3((P.K / 128) * ((P.O + 127) / 128) * p.S), dim3(256), lds, s, P);
        e = check_launch("wgrad_fast_kernel<wino>");
    } else if (p.bmo == 128 && p.bnk == 128) e = launch_wgrad<2, 2, 2, 2>(P, p.S, s);
    else if (p.bmo == 64 && p.bnk == 128) e = launch_wgrad<1, 4, 2, 1>(P, p.S, s);
    else if (p.bmo == 32) e = launch_wgrad<1, 4, 1, 2>(P, p.S, s);      // 32 x 256
    else if (p.bmo == 128 && p.bnk == 96) e = launch_wgrad<4, 1, 1, 3>(P, p.S, s);
    else if (p.bmo == 128) e = launch_wgrad<4, 1, 1, 1>(P, p.S, s);     // 128 x 32 (1x1 convs with few inputs)
    else e = launch_wgrad<2, 2, 1, 1>(P, p.S, s);                       // 64 x 64
    if (e) return e;
    if (!reduce) return VQ2_OK;
    const int total = p.O * p.K;
    const int blocks = (total + 31) / 32 < 4096 ? (total + 31) / 32 : 4096;
    const int cir = d->Cir ? d->Cir : d->Ci, cor = d->Cor ? d->Cor : d->Co;
    const bool sw = d->transposed || p.swapped;
    const int Or = sw ? cir : cor, Ir = sw ? cor : cir;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, P.ws, dw, p.O, p.I, Or, Ir, d->KH * d->KW,
                       p.S, P.bias_ws, db, p.swapped | (p.wino << 1), P.bias_taps ? p.S * P.bias_nslots : 0);
    return check_launch("wgrad_reduce_kernel");
}

extern "C" int vq2_conv_wgrad(const vq2_conv_desc *d, int flags, const float *x, const float *dy, float *dw, float *db,
                              void *ws, size_t ws_bytes, vq2_stream_t stream) {
    return wgrad_impl(d, flags, x, dy, dw, db, ws, ws_bytes, stream, true);
}

extern "C" int vq2_conv_wgrad_partial(const vq2_conv_desc *d, int flags, const float *x, const float *dy, float *db,
                                      void *ws, size_t ws_bytes, vq2_stream_t stream) {
    return wgrad_impl(d, flags, x, dy, nullptr, db, ws, ws_bytes, stream, false);
}

extern "C" int vq2_wgrad_job_init(const vq2_conv_desc *d, const void *ws, float *dw, float *db, vq2_wgrad_job *job) {
    VQ2_REQUIRE(d && ws && dw && job, "wgrad_job_init: null pointer");
    const WgradPlan p = plan_wgrad(d);
    const int cir = d->Cir ? d->Cir : d->Ci, cor = d->Cor ? d->Cor : d->Co;
    const float *w = static_cast<const float *>(ws);
    job->ws = w; job->dw = dw; job->db = nullptr; job->bias_ws = nullptr;
    const bool sw = d->transposed || p.swapped;
    job->O = p.O; job->I = p.I; job->Or = sw ? cir : cor; job->Ir = sw ? cor : cir;
    job->swapped = p.swapped | (p.wino << 1); job->bias_splits = 0;
    job->taps = d->KH * d->KW; job->S = p.S;
    job->n_units_w = p.wino ? (p.O * (p.wino == 2 ? 8 : 3) * p.I + 31) / 32 : (p.O * p.K + 31) / 32;
    job->n_units_b = 0;
    if (db) {
        int taps, nslots;
        bias_taps_of(d, p, taps, nslots);
        job->db = db; job->bias_ws = w + (size_t)p.S * p.O * p.K;
        job->bias_splits = taps ? p.S * nslots : 0;
        job->n_units_b = ((taps ? job->Ir : job->Or) + 31) / 32;
    }
    job->unit_offset = 0;
    return VQ2_OK;
}

extern "C" int vq2_wgrad_reduce_batched(const vq2_wgrad_job *jobs_dev, int32_t njobs, int64_t total_units,
                                        vq2_stream_t stream) {
    VQ2_REQUIRE(jobs_dev && njobs > 0 && total_units > 0, "wgrad_reduce_batched: bad arguments");
    const unsigned blocks = (unsigned)(total_units < 8192 ? total_units : 8192);
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(blocks), dim3(256), 0, to_stream(stream), jobs_dev, njobs,
                       total_units);
    return check_launch("wgrad_reduce_batched_kernel");
}

extern "C" size_t vq2_colsum_workspace_bytes(int64_t rows, int32_t C) {
    if (rows <= 0 || C <= 0) return 0;
    const int64_t rpb = cs_rows_per_block(rows);
    return (size_t)((rows + rpb - 1) / rpb) * C * sizeof(float);
}

extern "C" int vq2_colsum(const float *dy, int64_t rows, int32_t C, int32_t ld, float *db, void *ws, size_t ws_bytes,
                          vq2_stream_t stream) {
    VQ2_REQUIRE(dy && db && ws, "colsum: null pointer");
    VQ2_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0 && C <= 4096, "colsum: bad shape");
    VQ2_REQUIRE(aligned16(dy), "colsum: dy must be 16-byte aligned");
    VQ2_REQUIRE(ws_bytes >= vq2_colsum_workspace_bytes(rows, C), "colsum: workspace too small");
    return colsum_impl(dy, rows, C, ld, db, C, static_cast<float *>(ws), to_stream(stream));
}
