// Quantize (vqvae.py:28-78) for gfx950: fused distance GEMM + argmin + gather + STE output +
// commitment-loss partials + EMA statistics, never materialising the [M,K] distance matrix.
//
// Distance GEMM orientation: codes are the MFMA *rows* (A operand, streamed through LDS),
// latent vectors are the *columns* (B operand, held in registers for the whole kernel):
//     S[code][vec] = sum_d E[d][code] * x[vec][d]        (v_mfma_f32_32x32x2_f32, exact fp32)
// so each lane owns ONE latent vector (column = lane&31) and sees 16 codes per tile in its
// accumulator registers: the running (min, argmin) over all K codes is lane-local, and the
// only cross-lane step is one exchange between the two half-waves (lane ^ 32) at the end.
// Ties: codes are visited in increasing order with a strict '<', and the half-wave exchange
// prefers the smaller index -> first minimal index, like (-dist).max(1) at vqvae.py:49.
//
// dist is evaluated exactly as the reference writes it (vqvae.py:44-48):
//     (||x||^2 - 2*(x.e)) + ||e||^2      in fp32, in that order.
#include "vq2_common.h"
#include <stdlib.h>

namespace vq2 {

constexpr int VQ_ROWS = 128;    // latent vectors per loss partial (32 per wave)

// DP = D rounded up to {16,32,64}; NW = waves per workgroup (32 vectors each); VQ_CT = codes staged per LDS tile.
// <DP,4,128>: 128 vectors per workgroup, 32 KB tiles, ~5 workgroups per CU -- small launches.
// <DP,16,512>: 512 vectors per workgroup and the reference's whole 512-code codebook (128 KB) staged ONCE, no
// barrier in the main loop, 16 waves per CU; a quarter of the workgroups also means a quarter of the same-address
// atomic bursts of the statistics when few codes are in use.  Chosen when the launch still fills every CU.
// Epilogue shared by the one-pass kernel and the K-split merge: index, gather, straight-through output
// x + (q - x) (vqvae.py:73), commitment-loss partial per 128 vectors.
template <int DP, int NW>
__device__ __forceinline__ void vq_finish(const float (&xf)[DP / 2], int besti, bool rv, int64_t row, int h, int lane,
                                          int wave, int tid, const float *__restrict__ embedT, int64_t M, int D,
                                          int64_t *__restrict__ idx_out, float *__restrict__ out, int ldo,
                                          float *__restrict__ loss_partial, float *wsum) {
    constexpr int HS = DP / 2;
    if (rv && h == 0) idx_out[row] = (int64_t)besti;
    float lsum = 0.f;
#pragma unroll
    for (int s = 0; s < HS; s += 4) {
        const int d = h * HS + s;
        if (rv && d < D) {
            const float4 q = *reinterpret_cast<const float4 *>(embedT + (size_t)besti * D + d);
            const float t0 = q.x - xf[s], t1 = q.y - xf[s + 1], t2 = q.z - xf[s + 2], t3 = q.w - xf[s + 3];
            lsum += t0 * t0; lsum += t1 * t1; lsum += t2 * t2; lsum += t3 * t3;
            if (out) {
                float4 o;
                o.x = xf[s] + t0; o.y = xf[s + 1] + t1; o.z = xf[s + 2] + t2; o.w = xf[s + 3] + t3;  // vqvae.py:73
                *reinterpret_cast<float4 *>(out + row * ldo + d) = o;
            }
        }
    }
    lsum = wave_sum(lsum);
    if (lane == 0) wsum[wave] = lsum;
    __syncthreads();
    if (tid < NW / 4 && loss_partial && (int64_t)(blockIdx.x * (NW / 4) + tid) * VQ_ROWS < M)   // one partial per 128 vectors
        loss_partial[blockIdx.x * (NW / 4) + tid] = (wsum[4 * tid] + wsum[4 * tid + 1]) + (wsum[4 * tid + 2] + wsum[4 * tid + 3]);
}

// K-split (gridDim.y = S > 1): a launch with few latent vectors and a large codebook (the top level of configs[3]:
// M = 32,768, K = 8,192) would put ONE wave on each SIMD, with nothing to overlap its argmin / staging phases.
// Split s searches codes [s*kper, (s+1)*kper) and leaves (best distance, best index) in pbest / pidx [S][M];
// vq_merge_kernel takes the first minimum over the splits in ascending code order -- the same index as one pass.
template <int DP, int NW, int VQ_CT>
__global__ __launch_bounds__(64 * NW) void vq_fwd_kernel(const float *__restrict__ x, int ldx,
                                                     const float *__restrict__ embed,   // [D][K]
                                                     const float *__restrict__ embedT,  // [K][D]
                                                     const float *__restrict__ enorm,   // [K]
                                                     int64_t M, int D, int K, int64_t *__restrict__ idx_out,
                                                     float *__restrict__ out, int ldo,
                                                     float *__restrict__ loss_partial, int kper,
                                                     float *__restrict__ pbest, int *__restrict__ pidx) {
    constexpr int HS = DP / 2;  // MFMA k-steps; lane half h covers d in [h*HS, (h+1)*HS)
    constexpr int NT = 64 * NW, ROWS = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) float vq_smem[];
    float *Es = vq_smem;                                   // [DP][VQ_CT]
    float *En = vq_smem + DP * VQ_CT;                      // [VQ_CT]
    float *wsum = En + VQ_CT;                              // [NW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * ROWS + wave * 32 + col;
    const bool rv = row < M;

    // this lane's half of its latent vector, as MFMA B fragments
    float xf[HS];
#pragma unroll
    for (int s = 0; s < HS; s += 4) {
        const int d = h * HS + s;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rv && d < D) v = *reinterpret_cast<const float4 *>(x + row * ldx + d);
        xf[s] = v.x; xf[s + 1] = v.y; xf[s + 2] = v.z; xf[s + 3] = v.w;
    }
    float xx = 0.f;
#pragma unroll
    for (int s = 0; s < HS; ++s) xx += xf[s] * xf[s];
    xx += __shfl_xor(xx, 32, 64);

    float best = __builtin_inff();
    int besti = 0;

    const int kbeg = blockIdx.y * kper;
    const int kend = (kbeg + kper < K) ? kbeg + kper : K;
    for (int ct0 = kbeg; ct0 < kend; ct0 += VQ_CT) {
        __syncthreads();
        // stage E[:, ct0:ct0+VQ_CT] (zero-padded) and its norms
        for (int t = tid; t < DP * (VQ_CT / 4); t += NT) {
            const int d = t / (VQ_CT / 4), c4 = (t % (VQ_CT / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d < D && ct0 + c4 < kend) v = *reinterpret_cast<const float4 *>(embed + (size_t)d * K + ct0 + c4);
            *reinterpret_cast<float4 *>(Es + d * VQ_CT + c4) = v;
        }
        // codes past K get norm +inf: their distance never wins, no per-element range test below
        for (int t = tid; t < VQ_CT; t += NT) En[t] = (ct0 + t < kend) ? enorm[ct0 + t] : __builtin_inff();
        __syncthreads();

#pragma unroll 1
        for (int sub = 0; sub < VQ_CT / 32; ++sub) {
            if (ct0 + sub * 32 >= kend) break;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *ea = Es + (h * HS) * VQ_CT + sub * 32 + col;
#pragma unroll
            for (int s = 0; s < HS; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[s * VQ_CT], xf[s], acc, 0, 0, 0);
            // The argmin runs on the vector ALU, which the fp32 MFMAs of the other waves share: keep it to five
            // instructions per distance.  fma(-2, dot, xx) rounds once and 2*dot is exact, so it equals the
            // reference's (xx - 2*dot) bit for bit; the code index is (wave-uniform base + compile-time row) --
            // this lane's constant 4*h is added once after the loop.
            const int cbase = ct0 + sub * 32;          // scalar
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 en = *reinterpret_cast<const float4 *>(En + sub * 32 + 8 * r4 + 4 * h);
                const float e4[4] = {en.x, en.y, en.z, en.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dist = __builtin_fmaf(-2.f, acc[r4 * 4 + q], xx) + e4[q];
                    const bool lt = dist < best;
                    best = lt ? dist : best;
                    besti = lt ? cbase + 8 * r4 + q : besti;
                }
            }
        }
    }
    besti += 4 * h;
    // combine the two half-waves (each saw half of the codes of every tile)
    {
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(besti, 32, 64);
        if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (gridDim.y > 1) {
        if (rv && h == 0) {
            pbest[(size_t)blockIdx.y * M + row] = best;
            pidx[(size_t)blockIdx.y * M + row] = besti;
        }
        return;
    }
    vq_finish<DP, NW>(xf, besti, rv, row, h, lane, wave, tid, embedT, M, D, idx_out, out, ldo, loss_partial, wsum);
}

template <int DP>
__global__ __launch_bounds__(256) void vq_merge_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ embedT,
                                                       int64_t M, int D, int S, const float *__restrict__ pbest,
                                                       const int *__restrict__ pidx, int64_t *__restrict__ idx_out,
                                                       float *__restrict__ out, int ldo, float *__restrict__ loss_partial) {
    constexpr int HS = DP / 2, NW = 4;
    __shared__ float wsum[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * (32 * NW) + wave * 32 + col;
    const bool rv = row < M;
    float xf[HS];
#pragma unroll
    for (int s = 0; s < HS; s += 4) {
        const int d = h * HS + s;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rv && d < D) v = *reinterpret_cast<const float4 *>(x + row * ldx + d);
        xf[s] = v.x; xf[s + 1] = v.y; xf[s + 2] = v.z; xf[s + 3] = v.w;
    }
    float best = __builtin_inff();
    int besti = 0;
    if (rv)
        for (int s = 0; s < S; ++s) {   // ascending code ranges + strict '<': the first minimal index wins
            const float b = pbest[(size_t)s * M + row];
            const int i = pidx[(size_t)s * M + row];
            if (b < best) { best = b; besti = i; }
        }
    vq_finish<DP, NW>(xf, besti, rv, row, h, lane, wave, tid, embedT, M, D, idx_out, out, ldo, loss_partial, wsum);
}

// ----------------------------------------------------------------------------------------------------------------
// EMA statistics (vqvae.py:55-56): counts[k] = #{m : idx[m] == k}, sumsT[k][:] = sum of the rows x[m][:] with idx[m] == k.
// Bit-reproducible by construction (no float atomics): a STABLE counting sort of the row numbers by code, then every
// code's rows are summed in increasing row order along a fixed two-level tree (64 sorted positions per chunk, chunk
// partials folded in chunk order).  The cost does not depend on how many codes are in use: the rows are gathered once
// (M*D*4 bytes at the HBM/Infinity-Cache rate) whether all vectors pick one code or every code is used.
//   hist   per block of ST_RB rows: LDS histogram (integer atomics: exact)             -> blockhist[b][k]
//   scan   per code: exclusive prefix over the blocks, totals                          -> blockhist (in place), ncode, counts
//   place  per block: codeoff = exclusive scan of ncode; stable local ranks            -> perm[pos] = row, scode[pos] = code
//   chunk  per 64 sorted positions (one wave): runs of equal code summed in order      -> sumsT (complete runs) / carry
//   fix    per code (one wave): zero for unused codes, carries folded in chunk order   -> sumsT
constexpr int ST_RB = 1024;   // rows per sort block (= threads of hist / place)
constexpr int ST_CH = 64;     // sorted positions per chunk (= one wave)

__global__ __launch_bounds__(ST_RB) void vq_stats_hist_kernel(const int64_t *__restrict__ idx, int64_t M, int K,
                                                              int *__restrict__ blockhist) {
    extern __shared__ int st_hist[];
    for (int k = threadIdx.x; k < K; k += ST_RB) st_hist[k] = 0;
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * ST_RB + threadIdx.x;
    if (m < M) {
        int c = (int)idx[m];
        c = c < 0 ? 0 : (c >= K ? K - 1 : c);
        atomicAdd(&st_hist[c], 1);
    }
    __syncthreads();
    int *dst = blockhist + (size_t)blockIdx.x * K;
    for (int k = threadIdx.x; k < K; k += ST_RB) dst[k] = st_hist[k];
}

// 16 codes x 16 groups of blocks per workgroup: blockhist[b][k] becomes the number of rows with code k in blocks < b
__global__ __launch_bounds__(256) void vq_stats_scan_kernel(int *__restrict__ blockhist, int NB, int K,
                                                            int *__restrict__ ncode, float *__restrict__ counts) {
    __shared__ int part[16][17];
    const int kq = threadIdx.x & 15, bq = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + kq;
    const int per = (NB + 15) / 16;
    const int b0 = bq * per, b1 = (b0 + per < NB) ? b0 + per : NB;
    int sum = 0;
    if (k < K)
        for (int b = b0; b < b1; ++b) sum += blockhist[(size_t)b * K + k];
    part[bq][kq] = sum;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < bq; ++q) base += part[q][kq];
    if (k < K) {
        int run = base;
        for (int b = b0; b < b1; ++b) {
            const int c = blockhist[(size_t)b * K + k];
            blockhist[(size_t)b * K + k] = run;
            run += c;
        }
        if (bq == 15) {
            ncode[k] = base + sum;
            counts[k] = (float)(base + sum);   // exact: integers below 2^24 (M <= 16.7M rows per call)
        }
    }
}

__global__ __launch_bounds__(ST_RB) void vq_stats_place_kernel(const int64_t *__restrict__ idx, int64_t M, int K,
                                                               const int *__restrict__ blockoff,
                                                               const int *__restrict__ ncode, int *__restrict__ codeoff,
                                                               int *__restrict__ perm, int *__restrict__ scode) {
    extern __shared__ int st_cnt[];   // [K] next free position of a code for this block; then [16] wave totals
    int *wtot = st_cnt + K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    // codeoff = exclusive scan of ncode: KP consecutive codes per thread, wave scan, 16 wave totals
    const int KP = (K + ST_RB - 1) / ST_RB;
    const int k0 = tid * KP;
    int local = 0;
    for (int j = 0; j < KP; ++j)
        if (k0 + j < K) local += ncode[k0 + j];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wtot[w];
    int run = wbase + incl - local;
    const int *boff = blockoff + (size_t)b * K;
    for (int j = 0; j < KP; ++j) {
        const int k = k0 + j;
        if (k < K) {
            st_cnt[k] = run + boff[k];
            if (b == 0) codeoff[k] = run;
            run += ncode[k];
        }
    }
    if (b == 0 && tid == ST_RB - 1) codeoff[K] = run;   // the last thread's running sum is the grand total
    // stable rank inside the wave: lanes with equal codes, in lane (= row) order
    const int64_t m = (int64_t)b * ST_RB + tid;
    const bool valid = m < M;
    int c = -1;
    if (valid) {
        c = (int)idx[m];
        c = c < 0 ? 0 : (c >= K ? K - 1 : c);
    }
    int wrank = 0, wcount = 0;
    bool islast = false;
    unsigned long long rem = __ballot(valid);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    while (rem) {
        const int leader = __ffsll((long long)rem) - 1;
        const int lc = __shfl(c, leader, 64);
        const unsigned long long same = __ballot(valid && c == lc);
        if (valid && c == lc) {
            wrank = __popcll(same & below);
            wcount = __popcll(same);
            islast = (same >> lane) == 1ull;
        }
        rem &= ~same;
    }
    // the 16 waves take their turns in row order: base position of the code, then the code's counter moves on
    int pos = 0;
    for (int w = 0; w < ST_RB / 64; ++w) {
        __syncthreads();
        if (wave == w && valid) {
            const int basep = st_cnt[c];
            pos = basep + wrank;
            if (islast) st_cnt[c] = basep + wcount;
        }
    }
    if (valid) {
        perm[pos] = (int)m;
        scode[pos] = c;
    }
}

template <int NV>   // floats per lane: D <= 64 * NV
__global__ __launch_bounds__(256) void vq_stats_chunk_kernel(const float *__restrict__ x, int ldx, int D, int64_t M,
                                                             const int *__restrict__ perm, const int *__restrict__ scode,
                                                             const int *__restrict__ codeoff, float *__restrict__ sumsT,
                                                             float *__restrict__ carry) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p0 = g * ST_CH;
    if (p0 >= M) return;
    const int n = (M - p0 < ST_CH) ? (int)(M - p0) : ST_CH;
    const int row_l = (lane < n) ? perm[p0 + lane] : 0;
    const int code_l = (lane < n) ? scode[p0 + lane] : -1;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    int cur = __builtin_amdgcn_readlane(code_l, 0);
    auto flush = [&](int k) {
        const int s = codeoff[k], e = codeoff[k + 1];
        float *dst;
        if ((int64_t)s >= p0 && (int64_t)e <= p0 + ST_CH) dst = sumsT + (size_t)k * D;             // the code's whole segment
        else dst = carry + ((size_t)g * 2 + ((int64_t)s <= p0 ? 0 : 1)) * D;                      // head / tail partial
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < D) dst[lane + 64 * i] = acc[i];
    };
    constexpr int U = 16;   // rows in flight per wave
    for (int r0 = 0; r0 < n; r0 += U) {
        float v[U][NV];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int r = (r0 + j < n) ? r0 + j : n - 1;
            const int row = __builtin_amdgcn_readlane(row_l, r);
            const float *src = x + (size_t)row * ldx;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[j][i] = (lane + 64 * i < D) ? src[lane + 64 * i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (r0 + j < n) {                                    // wave-uniform
                const int code = __builtin_amdgcn_readlane(code_l, r0 + j);
                if (code != cur) {
                    flush(cur);
#pragma unroll
                    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
                    cur = code;
                }
#pragma unroll
                for (int i = 0; i < NV; ++i) acc[i] += v[j][i];
            }
        }
    }
    flush(cur);
}

// One workgroup per code.  A code whose rows span n > 1 chunks has one partial per chunk; a single chain over them would
// be a latency chain of n dependent L2 round trips (n = 2,048 when every vector of a 32 x 64 x 64 batch picks the same
// code), so the fold is a fixed two-stage tree instead: CL = 256 / (D/4) interleaved chains (thread (c, q) adds the
// float4 q of chunks g0 + c, g0 + c + CL, ... in order, 16 loads in flight), then the CL chain sums in order of c.
__global__ __launch_bounds__(256) void vq_stats_fix_kernel(int K, int D, const int *__restrict__ codeoff,
                                                           const float *__restrict__ carry, float *__restrict__ sumsT) {
    __shared__ float4 red[256];
    const int k = blockIdx.x;
    const int s = codeoff[k], e = codeoff[k + 1];
    const int Q = D >> 2, CL = 256 / Q;
    const int q = threadIdx.x % Q, c = threadIdx.x / Q;
    float4 *dst = reinterpret_cast<float4 *>(sumsT + (size_t)k * D);
    if (e == s) {
        if (c == 0) dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int g0 = s / ST_CH, g1 = (e - 1) / ST_CH;
    if (g0 == g1) return;   // the chunk kernel wrote the complete sum
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 16;
    for (int gb = g0 + c; gb <= g1; gb += CL * U) {
        float4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int g = gb + j * CL;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g <= g1)
                v[j] = *reinterpret_cast<const float4 *>(carry + ((size_t)g * 2 + (s <= g * ST_CH ? 0 : 1)) * D + 4 * q);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {   // adding the zero of a slot past g1 changes nothing (x + 0 = x, and -0 + 0 = +0 only
            acc.x += v[j].x;            // where the true sum is a zero as well)
            acc.y += v[j].y;
            acc.z += v[j].z;
            acc.w += v[j].w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (c == 0) {
        float4 t = red[q];
        for (int cc = 1; cc < CL; ++cc) {
            const float4 o = red[cc * Q + q];
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        dst[q] = t;
    }
}

// embedT[k][d] = embed[d][k], enorm[k] = sum_d embed[d][k]^2: 64 codes x 4 d-lanes per workgroup
__global__ __launch_bounds__(256) void vq_prepare_kernel(const float *__restrict__ embed, float *__restrict__ embedT,
                                                         float *__restrict__ enorm, int D, int K) {
    __shared__ float part[4][64];
    const int kq = threadIdx.x & 63, dg = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kq;
    float s = 0.f;
    if (k < K) {
        for (int d = dg; d < D; d += 4) {
            const float e = embed[(size_t)d * K + k];
            embedT[(size_t)k * D + d] = e;
            s += e * e;
        }
    }
    part[dg][kq] = s;
    __syncthreads();
    if (dg == 0 && k < K) enorm[k] = (part[0][kq] + part[1][kq]) + (part[2][kq] + part[3][kq]);
}

__global__ void vq_loss_kernel(const float *__restrict__ part, int nparts, float denom, float *__restrict__ diff) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) diff[0] = red[0] / denom;
}

__global__ void vq_bwd_kernel(const float *__restrict__ g_out, int ldg, const float *__restrict__ g_diff,
                              const float *__restrict__ x, int ldx, const int64_t *__restrict__ idx,
                              const float *__restrict__ embedT, int64_t M, int D, float inv_numel,
                              float *__restrict__ dx, int lddx) {
    const int D4 = D / 4;
    const int64_t total = M * D4;
    const float gN = g_diff ? g_diff[0] * inv_numel : 0.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = t / D4;
        const int d = (int)(t - m * D4) * 4;
        const float4 xv = *reinterpret_cast<const float4 *>(x + m * ldx + d);
        const float4 q = *reinterpret_cast<const float4 *>(embedT + (size_t)idx[m] * D + d);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_out) g = *reinterpret_cast<const float4 *>(g_out + m * ldg + d);
        // autograd of vqvae.py:72-73: dx = g_out - (g_diff/numel) * (2 * (q - x))
        float4 o;
        o.x = g.x - gN * (2.f * (q.x - xv.x));
        o.y = g.y - gN * (2.f * (q.y - xv.y));
        o.z = g.z - gN * (2.f * (q.z - xv.z));
        o.w = g.w - gN * (2.f * (q.w - xv.w));
        *reinterpret_cast<float4 *>(dx + m * lddx + d) = o;
    }
}

// vqvae.py:61-66: cluster_size EMA and its total n (one workgroup; n feeds the Laplace smoothing)
__global__ __launch_bounds__(1024) void vq_ema_counts_kernel(float *__restrict__ cluster_size,
                                                             const float *__restrict__ counts, int K, float decay,
                                                             float alpha, float *__restrict__ n_out) {
    __shared__ float red[1024];
    float local = 0.f;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float cs = fmaf(alpha, counts[k], cluster_size[k] * decay);
        cluster_size[k] = cs;
        local += cs;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) n_out[0] = red[0];
}

// vqvae.py:64, 67-70: embed_avg EMA and the normalised codebook; thread index runs over [k][d] so the
// transposed statistics are read coalesced
// PREP (embedT / enorm outputs, 256 % D == 0): the launch also leaves what the NEXT forward's vq2_vq_prepare would
// compute -- the transposed codebook and ||e_k||^2 summed in vq_prepare_kernel's own order (four chains over
// d = g, g+4, ..., then (p0+p1)+(p2+p3)), so distances and indices are bit-identical to preparing separately.  A block's
// 256 threads own 256/D complete codes (256 and the grid stride are multiples of D).
template <bool PREP>
__global__ __launch_bounds__(256) void vq_ema_embed_kernel(float *__restrict__ embed, const float *__restrict__ cluster_size,
                                                           float *__restrict__ embed_avg, const float *__restrict__ sumsT,
                                                           int D, int K, float decay, float alpha, float eps, float keps,
                                                           const float *__restrict__ n_in, float *__restrict__ embedT,
                                                           float *__restrict__ enorm) {
    __shared__ float e2[256];
    __shared__ float part[256];
    const float n = n_in[0];
    const float denom = n + keps;
    const int total = D * K;
    const int rounds = (total + gridDim.x * 256 - 1) / (gridDim.x * 256);
    for (int r = 0; r < rounds; ++r) {
        const int t = (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        const bool tv = t < total;
        const int k = tv ? t / D : 0, d = tv ? t - k * D : 0;
        float e = 0.f;
        if (tv) {
            const size_t o = (size_t)d * K + k;
            const float ea = fmaf(alpha, sumsT[t], embed_avg[o] * decay);
            embed_avg[o] = ea;
            const float cs = (cluster_size[k] + eps) / denom * n;
            e = ea / cs;
            embed[o] = e;
            if (PREP) embedT[t] = e;
        }
        if (PREP) {
            e2[threadIdx.x] = e * e;
            __syncthreads();
            if (tv && d < 4) {
                float sacc = 0.f;
                const int base = threadIdx.x - d;
                for (int dd = d; dd < D; dd += 4) sacc += e2[base + dd];
                part[threadIdx.x] = sacc;
            }
            __syncthreads();
            if (tv && d == 0) {
                const float p0 = part[threadIdx.x], p1 = D > 1 ? part[threadIdx.x + 1] : 0.f;
                const float p2 = D > 2 ? part[threadIdx.x + 2] : 0.f, p3 = D > 3 ? part[threadIdx.x + 3] : 0.f;
                enorm[k] = (p0 + p1) + (p2 + p3);
            }
            __syncthreads();
        }
    }
}

__global__ void vq_gather_kernel(const int64_t *__restrict__ idx, const float *__restrict__ embedT, int64_t M, int D,
                                 int K, float *__restrict__ out, int ldo) {
    const int D4 = D / 4;
    const int64_t total = M * D4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = t / D4;
        const int d = (int)(t - m * D4) * 4;
        int64_t c = idx[m];
        c = c < 0 ? 0 : (c >= K ? K - 1 : c);  // never read outside the codebook
        *reinterpret_cast<float4 *>(out + m * ldo + d) = *reinterpret_cast<const float4 *>(embedT + (size_t)c * D + d);
    }
}

static inline int grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace vq2

using namespace vq2;

extern "C" int vq2_vq_prepare(const float *embed, float *embedT, float *enorm, int32_t D, int32_t K,
                              vq2_stream_t stream) {
    VQ2_REQUIRE(embed && embedT && enorm && D > 0 && K > 0, "vq_prepare: bad arguments");
    hipLaunchKernelGGL(vq_prepare_kernel, dim3((K + 63) / 64), dim3(256), 0, to_stream(stream), embed, embedT, enorm,
                       D, K);
    return check_launch("vq_prepare_kernel");
}

// big = 512-vector workgroups (16 waves, whole 512-code tiles); S = number of K-splits; kper = codes per split
static int vq_tune(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

static void vq_plan(int64_t M, int32_t D, int32_t K, bool &big, int &S, int &kper) {
    big = M >= 512 * 256;   // 512-vector workgroups still cover every CU
    S = 1;
    kper = K;
    if (D > 64) { big = false; return; }   // embed_dim 128 / 256 (VQVAE_Deep): one 8-wave shape, see vq2_vq_fwd
    static const int split_small = vq_tune("VQ2_VQ_SPLIT_SMALL", 1);
    if (!big && K >= 1024) {
        const int64_t want = (512 * 256 + M - 1) / M;           // splits that bring the launch to one 16-wave workgroup per CU
        const int64_t most = K / 512;                            // at least one full 512-code tile per split
        S = (int)(want < most ? want : most);
        big = S > 1;
        if (big) kper = ((K + S - 1) / S + 511) / 512 * 512;
    }
    if (!big && S == 1 && split_small && K >= 256 && (M + 127) / 128 < 1024) {
        // few vectors and a small codebook (the top level of the default model: M = 32,768, K = 512): 128-vector
        // workgroups searching ONE 128-code tile each -- four times the workgroups, no staging loop (45 -> 38 us; the
        // same split of the full-size launch was measured 40 % SLOWER than its 512-vector workgroups and is not taken)
        const int64_t want = (1024 + (M + 127) / 128 - 1) / ((M + 127) / 128);
        const int64_t most = K / 128;
        S = (int)(want < most ? want : most);
        if (S < 1) S = 1;
        if (S > 1) kper = ((K + S - 1) / S + 127) / 128 * 128;
    }
}

extern "C" size_t vq2_vq_fwd_workspace_floats(int64_t M, int32_t D, int32_t K) {
    if (M <= 0 || K <= 0) return 0;
    bool big; int S, kper;
    vq_plan(M, D, K, big, S, kper);
    const size_t nparts = (size_t)((M + VQ_ROWS - 1) / VQ_ROWS);
    return (nparts + 3) / 4 * 4 + (S > 1 ? (size_t)2 * S * M : 0);
}

extern "C" int vq2_vq_fwd(const float *x, int32_t ldx, const float *embed, const float *embedT, const float *enorm,
                          int64_t M, int32_t D, int32_t K, int64_t *idx, float *out, int32_t ldo, float *ws,
                          vq2_stream_t stream) {
    VQ2_REQUIRE(x && embed && embedT && enorm && idx && ws, "vq_fwd: null pointer");
    VQ2_REQUIRE(M > 0 && K > 0 && K % 4 == 0 && D >= 4 && D <= 256 && (D & (D - 1)) == 0,
                "vq_fwd: need D a power of two in 4..256 and K %% 4 == 0 (D=%d K=%d)", D, K);
    VQ2_REQUIRE(ldx >= D && ldx % 4 == 0 && (!out || (ldo >= D && ldo % 4 == 0)), "vq_fwd: bad pixel strides");
    VQ2_REQUIRE(aligned16(x) && aligned16(embed) && aligned16(embedT) && (!out || aligned16(out)),
                "vq_fwd: pointers must be 16-byte aligned");
    hipStream_t s = to_stream(stream);
    ProfScope prof(prof_label("vq_fwd|M=%lld,D=%d,K=%d", (long long)M, D, K), 2.0 * (double)M * D * K,
                   4.0 * ((double)M * D * 2 + (double)D * K), s);
    bool big; int S, kper;
    vq_plan(M, D, K, big, S, kper);
    float *loss_partial = ws;
    const size_t nparts = (size_t)((M + VQ_ROWS - 1) / VQ_ROWS);
    float *pbest = S > 1 ? ws + (nparts + 3) / 4 * 4 : nullptr;
    int *pidx = S > 1 ? reinterpret_cast<int *>(pbest + (size_t)S * M) : nullptr;
#define VQ2_LAUNCH_VQ(DP, NW, CT)                                                                                    \
    do {                                                                                                             \
        const size_t lds = ((size_t)DP * CT + CT + NW) * sizeof(float);                                              \
        const unsigned grid = (unsigned)((M + 32 * NW - 1) / (32 * NW));                                             \
        allow_big_lds(vq_fwd_kernel<DP, NW, CT>, lds);                                                               \
        hipLaunchKernelGGL((vq_fwd_kernel<DP, NW, CT>), dim3(grid, S), dim3(64 * NW), lds, s, x, ldx, embed, embedT, enorm, \
                           M, D, K, idx, out, ldo, loss_partial, kper, pbest, pidx);                                 \
        if (int e = check_launch("vq_fwd_kernel")) return e;                                                         \
        if (S > 1)                                                                                                   \
            hipLaunchKernelGGL((vq_merge_kernel<(DP <= 64 ? DP : 64)>), dim3((unsigned)((M + 127) / 128)), dim3(256), 0, s,  \
                               x, ldx, embedT, M, D, S, pbest, pidx, idx, out, ldo, loss_partial);                   \
    } while (0)
    // D = 128 / 256: this lane's half of the vector is 64 / 128 B-fragment registers, so 8 waves per workgroup
    // (two per SIMD, <= 256 VGPRs) and code tiles of 256 / 128 (128 KB of LDS)
    if (D > 128) VQ2_LAUNCH_VQ(256, 8, 128);
    else if (D > 64) VQ2_LAUNCH_VQ(128, 8, 256);
    else if (D <= 16) { if (big) VQ2_LAUNCH_VQ(16, 16, 512); else VQ2_LAUNCH_VQ(16, 4, 128); }
    else if (D <= 32) { if (big) VQ2_LAUNCH_VQ(32, 16, 512); else VQ2_LAUNCH_VQ(32, 4, 128); }
    else { if (big) VQ2_LAUNCH_VQ(64, 16, 512); else VQ2_LAUNCH_VQ(64, 4, 128); }
#undef VQ2_LAUNCH_VQ
    return check_launch("vq_fwd_kernel");
}

namespace {
struct StatsLayout {
    size_t blockhist, ncode, codeoff, perm, scode, carry, total;
    int NB;
    int64_t NCH;
};
inline size_t up16(size_t b) { return (b + 15) / 16 * 16; }
inline StatsLayout stats_layout(int64_t M, int32_t D, int32_t K) {
    StatsLayout L;
    L.NB = (int)((M + ST_RB - 1) / ST_RB);
    L.NCH = (M + ST_CH - 1) / ST_CH;
    size_t o = 0;
    L.blockhist = o; o += up16((size_t)L.NB * K * sizeof(int));
    L.ncode = o;     o += up16((size_t)K * sizeof(int));
    L.codeoff = o;   o += up16((size_t)(K + 1) * sizeof(int));
    L.perm = o;      o += up16((size_t)M * sizeof(int));
    L.scode = o;     o += up16((size_t)M * sizeof(int));
    L.carry = o;     o += up16((size_t)L.NCH * 2 * D * sizeof(float));
    L.total = o;
    return L;
}
}  // namespace

extern "C" size_t vq2_vq_stats_workspace_bytes(int64_t M, int32_t D, int32_t K) {
    if (M <= 0 || D <= 0 || K <= 0) return 0;
    return stats_layout(M, D, K).total;
}

extern "C" int vq2_vq_stats(const float *x, int32_t ldx, const int64_t *idx, int64_t M, int32_t D, int32_t K,
                            float *counts, float *sumsT, void *ws, size_t ws_bytes, vq2_stream_t stream) {
    VQ2_REQUIRE(x && idx && counts && sumsT && ws, "vq_stats: null pointer");
    VQ2_REQUIRE(M > 0 && M < (1ll << 24) && D >= 4 && D <= 256 && (D & (D - 1)) == 0 && K > 0 && K <= 16384 && ldx >= D,
                "vq_stats: need 0 < M < 2^24, D a power of two in 4..256, K <= 16384 (M=%lld D=%d K=%d)", (long long)M, D, K);
    VQ2_REQUIRE(aligned16(sumsT), "vq_stats: sumsT must be 16-byte aligned");
    const StatsLayout L = stats_layout(M, D, K);
    if (ws_bytes < L.total) return set_error(VQ2_ERR_WORKSPACE, "vq_stats: workspace %zu < %zu bytes", ws_bytes, L.total);
    VQ2_REQUIRE(aligned16(ws), "vq_stats: workspace must be 16-byte aligned");
    hipStream_t s = to_stream(stream);
    char *base = static_cast<char *>(ws);
    int *blockhist = reinterpret_cast<int *>(base + L.blockhist), *ncode = reinterpret_cast<int *>(base + L.ncode);
    int *codeoff = reinterpret_cast<int *>(base + L.codeoff), *perm = reinterpret_cast<int *>(base + L.perm);
    int *scode = reinterpret_cast<int *>(base + L.scode);
    float *carry = reinterpret_cast<float *>(base + L.carry);
    ProfScope prof(prof_label("vq_stats|M=%lld,D=%d,K=%d", (long long)M, D, K), (double)M * D,
                   4.0 * ((double)M * D + (double)K * D) + 8.0 * M, s);
    const size_t lds_h = (size_t)K * sizeof(int), lds_p = ((size_t)K + 16) * sizeof(int);
    allow_big_lds(vq_stats_hist_kernel, lds_h);
    hipLaunchKernelGGL(vq_stats_hist_kernel, dim3(L.NB), dim3(ST_RB), lds_h, s, idx, M, K, blockhist);
    if (int e = check_launch("vq_stats_hist_kernel")) return e;
    hipLaunchKernelGGL(vq_stats_scan_kernel, dim3((K + 15) / 16), dim3(256), 0, s, blockhist, L.NB, K, ncode, counts);
    if (int e = check_launch("vq_stats_scan_kernel")) return e;
    allow_big_lds(vq_stats_place_kernel, lds_p);
    hipLaunchKernelGGL(vq_stats_place_kernel, dim3(L.NB), dim3(ST_RB), lds_p, s, idx, M, K, blockhist, ncode, codeoff,
                       perm, scode);
    if (int e = check_launch("vq_stats_place_kernel")) return e;
    const unsigned cgrid = (unsigned)((L.NCH + 3) / 4);
    if (D <= 64)
        hipLaunchKernelGGL(vq_stats_chunk_kernel<1>, dim3(cgrid), dim3(256), 0, s, x, ldx, D, M, perm, scode, codeoff, sumsT, carry);
    else if (D <= 128)
        hipLaunchKernelGGL(vq_stats_chunk_kernel<2>, dim3(cgrid), dim3(256), 0, s, x, ldx, D, M, perm, scode, codeoff, sumsT, carry);
    else
        hipLaunchKernelGGL(vq_stats_chunk_kernel<4>, dim3(cgrid), dim3(256), 0, s, x, ldx, D, M, perm, scode, codeoff, sumsT, carry);
    if (int e = check_launch("vq_stats_chunk_kernel")) return e;
    hipLaunchKernelGGL(vq_stats_fix_kernel, dim3(K), dim3(256), 0, s, K, D, codeoff, carry, sumsT);
    return check_launch("vq_stats_fix_kernel");
}

extern "C" int vq2_vq_loss(const float *loss_partial, int64_t M, int32_t D, float *diff, vq2_stream_t stream) {
    VQ2_REQUIRE(loss_partial && diff && M > 0 && D > 0, "vq_loss: bad arguments");
    const int nparts = (int)((M + VQ_ROWS - 1) / VQ_ROWS);
    hipLaunchKernelGGL(vq_loss_kernel, dim3(1), dim3(256), 0, to_stream(stream), loss_partial, nparts,
                       (float)((double)M * (double)D), diff);
    return check_launch("vq_loss_kernel");
}

extern "C" int vq2_vq_bwd(const float *g_out, int32_t ldg, const float *g_diff, const float *x, int32_t ldx,
                          const int64_t *idx, const float *embedT, int64_t M, int32_t D, int32_t K, float *dx,
                          int32_t lddx, vq2_stream_t stream) {
    VQ2_REQUIRE(x && idx && embedT && dx, "vq_bwd: null pointer");
    VQ2_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && K > 0, "vq_bwd: bad dims");
    VQ2_REQUIRE(ldx >= D && lddx >= D && (!g_out || ldg >= D) && ldx % 4 == 0 && lddx % 4 == 0 && ldg % 4 == 0,
                "vq_bwd: bad pixel strides");
    hipLaunchKernelGGL(vq_bwd_kernel, dim3(grid_for(M * (D / 4))), dim3(256), 0, to_stream(stream), g_out, ldg, g_diff,
                       x, ldx, idx, embedT, M, D, (float)(1.0 / ((double)M * (double)D)), dx, lddx);
    return check_launch("vq_bwd_kernel");
}

static int ema_update_impl(float *embed, float *cluster_size, float *embed_avg, const float *counts, const float *sumsT,
                           int32_t D, int32_t K, double decay, double eps, float *scratch, float *embedT, float *enorm,
                           hipStream_t s) {
    // Python forms (1 - decay) and n_embed * eps in double before the fp32 ops (vqvae.py:62,67)
    const float alpha = (float)(1.0 - decay);
    hipLaunchKernelGGL(vq_ema_counts_kernel, dim3(1), dim3(1024), 0, s, cluster_size, counts, K, (float)decay, alpha,
                       scratch);
    if (int e = check_launch("vq_ema_counts_kernel")) return e;
    const int blocks = (D * K + 255) / 256;
    const dim3 grid(blocks > 1024 ? 1024 : blocks);
    if (embedT)
        hipLaunchKernelGGL(vq_ema_embed_kernel<true>, grid, dim3(256), 0, s, embed, cluster_size, embed_avg, sumsT, D, K,
                           (float)decay, alpha, (float)eps, (float)((double)K * eps), scratch, embedT, enorm);
    else
        hipLaunchKernelGGL(vq_ema_embed_kernel<false>, grid, dim3(256), 0, s, embed, cluster_size, embed_avg, sumsT, D, K,
                           (float)decay, alpha, (float)eps, (float)((double)K * eps), scratch, embedT, enorm);
    return check_launch("vq_ema_embed_kernel");
}

extern "C" int vq2_vq_ema_update(float *embed, float *cluster_size, float *embed_avg, const float *counts,
                                 const float *sumsT, int32_t D, int32_t K, double decay, double eps, float *scratch,
                                 vq2_stream_t stream) {
    VQ2_REQUIRE(embed && cluster_size && embed_avg && counts && sumsT && scratch && D > 0 && K > 0,
                "vq_ema_update: bad arguments");
    return ema_update_impl(embed, cluster_size, embed_avg, counts, sumsT, D, K, decay, eps, scratch, nullptr, nullptr,
                           to_stream(stream));
}

extern "C" int vq2_vq_ema_update_prepare(float *embed, float *cluster_size, float *embed_avg, const float *counts,
                                         const float *sumsT, int32_t D, int32_t K, double decay, double eps,
                                         float *scratch, float *embedT, float *enorm, vq2_stream_t stream) {
    VQ2_REQUIRE(embed && cluster_size && embed_avg && counts && sumsT && scratch && embedT && enorm && D > 0 && K > 0,
                "vq_ema_update_prepare: bad arguments");
    VQ2_REQUIRE(D <= 256 && 256 % D == 0, "vq_ema_update_prepare: D must divide 256 (D=%d)", D);
    return ema_update_impl(embed, cluster_size, embed_avg, counts, sumsT, D, K, decay, eps, scratch, embedT, enorm,
                           to_stream(stream));
}

extern "C" int vq2_vq_gather(const int64_t *idx, const float *embedT, int64_t M, int32_t D, int32_t K, float *out,
                             int32_t ldo, vq2_stream_t stream) {
    VQ2_REQUIRE(idx && embedT && out && M > 0 && D > 0 && D % 4 == 0 && K > 0 && ldo >= D && ldo % 4 == 0,
                "vq_gather: bad arguments");
    hipLaunchKernelGGL(vq_gather_kernel, dim3(grid_for(M * (D / 4))), dim3(256), 0, to_stream(stream), idx, embedT, M, D,
                       K, out, ldo);
    return check_launch("vq_gather_kernel");
}
