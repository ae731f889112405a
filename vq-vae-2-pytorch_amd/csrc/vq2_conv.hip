// NHWC conv2d / conv-transpose2d forward and data-gradient for gfx950 (MI355X).
//
// One kernel family: an implicit GEMM whose A operand is gathered on the fly from the
// NHWC activation (im2col never materialised), B operand is a packed weight panel, and
// whose inner product is the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32
// (bit-identical to an fmaf chain, 64 FLOP/clk/SIMD = the fp32 peak of the chip).
//
//   GEMM rows  m = (n, ho, wo) of a "virtual" output grid
//   GEMM cols  co
//   GEMM depth k = (kh, kw, ci), ci fastest  -> 16-byte coalesced loads along ci
//
// Everything the reference does around a conv on this path is fused:
//   - ReLU on the input (vqvae.py:86,88,107,109,...)      -> applied while staging A into LDS
//   - bias                                                 -> epilogue
//   - `out += input` of ResBlock (vqvae.py:94)             -> residual in the epilogue
//   - trailing in-place ReLU (vqvae.py:122,144)            -> epilogue
//   - ReLU backward (mask by pre-activation)               -> epilogue of the dgrad launch
//   - torch.cat (vqvae.py:218,233)                         -> pixel strides ldx/ldy on channel slices
//
// ConvTranspose2d(k4,s2,p1) and the data-gradient of a stride-2 4x4 conv are the same
// sub-pixel decomposition: 4 output phases, each a 2x2 stride-1 conv; blockIdx.z = phase.
#include "vq2_conv.h"
#include <type_traits>
#include <stdlib.h>

namespace vq2 {


// BK = depth of one staged chunk; LDS rows are padded to BK+4 floats (144 B / 80 B), which makes the
// ds_read_b128 fragment reads conflict-free (16-byte slot index = row*9 resp. row*5 mod 16).
template <int WAVES_M, int WAVES_N, int MT, int NT, int BK, bool STAMP = false>
__global__ __launch_bounds__(256, (MT * NT == 4) ? (BK == 16 ? 3 : 2) : 1) void conv_gemm_kernel(const ConvGemmParams P) {
    constexpr int LDK = BK + 4;
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int BN = WAVES_N * NT * 32;
    constexpr int KQ = BK / 4;            // float4 per staged row
    constexpr int RPP = 256 / KQ;         // rows staged per pass of the 256 threads
    constexpr int A_LD = BM / RPP;        // float4 loads per thread per chunk (A)
    constexpr int B_LD = (BN + RPP - 1) / RPP;  // BN < RPP: only the first BN staging rows carry weights
    static_assert(BM % RPP == 0 && (BN % RPP == 0 || BN < RPP), "tile vs staging shape");
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                  // [2][BM][LDK]
    float *Bs = smem + 2 * BM * LDK;   // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    // 1-D grid, XCD-aware: virtual id -> (n-tile fastest, then m-tile, then phase)
    const int ntiles = (P.Co + BN - 1) / BN;
    const int mtiles = (P.M + BM - 1) / BM;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (vid % ntiles) * BN;
    const int m0 = ((vid / ntiles) % mtiles) * BM;
    const int phase = vid / (ntiles * mtiles);

    int pad_h = P.pad_h, pad_w = P.pad_w, oh = 0, ow = 0, os = 1;
    const float *wp = P.w;
    if (P.phases == 4) {
        const int ph = phase >> 1, pw = phase & 1;
        pad_h = 1 - ph; pad_w = 1 - pw; oh = ph; ow = pw; os = 2;
        wp += (size_t)phase * P.Co * P.K;
    }

    // ---- per-thread staging coordinates (fixed over the K loop)
    const int lrow = tid / KQ;        // 0..RPP-1
    const int lk = (tid % KQ) * 4;    // float4 column inside a chunk
    int a_pix[A_LD], a_h[A_LD], a_w[A_LD];
    const int HoWo = P.Ho * P.Wo;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int m = m0 + lrow + RPP * j;
        if (m < P.M) {
            const int n = m / HoWo;
            const int r = m - n * HoWo;
            const int ho = r / P.Wo;
            const int wo = r - ho * P.Wo;
            a_h[j] = ho * P.stride - pad_h;
            a_w[j] = wo * P.stride - pad_w;
            a_pix[j] = (n * P.H + a_h[j]) * P.W + a_w[j];
        } else {
            a_h[j] = -(1 << 24); a_w[j] = 0; a_pix[j] = 0;  // every bounds test fails
        }
    }
    // k -> (kh, kw, ci) of this thread's float4, advanced incrementally
    int kglob = lk;
    int ci, kh, kw;
    {
        const int tap = kglob / P.Ci;
        ci = kglob - tap * P.Ci;
        kh = tap / P.KW;
        kw = tap - kh * P.KW;
    }

    float4 ra0[A_LD], rb0[B_LD];
    auto load_chunk = [&](float4 (&ra)[A_LD], float4 (&rb)[B_LD]) {
        const bool kv = kglob < P.K;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const bool v = kv && (unsigned)(a_h[j] + kh) < (unsigned)P.H && (unsigned)(a_w[j] + kw) < (unsigned)P.W;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v) val = *reinterpret_cast<const float4 *>(P.x + (size_t)(a_pix[j] + kh * P.W + kw) * P.ldx + ci);
            ra[j] = val;  // ReLU is applied when the registers are written to LDS (no wait on the load here)
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int co = n0 + lrow + RPP * j;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kv && co < P.Co && lrow + RPP * j < BN) val = *reinterpret_cast<const float4 *>(wp + (size_t)co * P.K + kglob);
            rb[j] = val;
        }
    };
    auto advance_k = [&]() {
        kglob += BK;
        ci += BK;
        while (ci >= P.Ci) {
            ci -= P.Ci;
            if (++kw == P.KW) { kw = 0; ++kh; }
        }
    };
    auto store_chunk = [&](int buf, const float4 (&ra)[A_LD], const float4 (&rb)[B_LD]) {
        float *a = As + buf * BM * LDK;
        float *b = Bs + buf * BN * LDK;
#pragma unroll
        for (int j = 0; j < A_LD; ++j)
            *reinterpret_cast<float4 *>(a + (lrow + RPP * j) * LDK + lk) = P.relu_in ? relu4(ra[j]) : ra[j];
#pragma unroll
        for (int j = 0; j < B_LD; ++j)
            if (BN % RPP == 0 || lrow + RPP * j < BN) *reinterpret_cast<float4 *>(b + (lrow + RPP * j) * LDK + lk) = rb[j];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = 4 * (lane >> 5);
    auto compute = [&](int buf) {
        const float *a = As + buf * BM * LDK + (wm * MT * 32 + frag_row) * LDK + frag_k;
        const float *b = Bs + buf * BN * LDK + (wn * NT * 32 + frag_row) * LDK + frag_k;
        // fragment registers are double-buffered: the ds_reads of k-group g+1 are issued before the MFMAs
        // of group g, so LDS latency never sits between two MFMA groups
        float4 fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = *reinterpret_cast<const float4 *>(a + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = *reinterpret_cast<const float4 *>(b + j * 32 * LDK);
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            const int cur = k8 & 1, nxt = cur ^ 1;
            if (k8 + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[nxt][i] = *reinterpret_cast<const float4 *>(a + i * 32 * LDK + (k8 + 1) * 8);
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[nxt][j] = *reinterpret_cast<const float4 *>(b + j * 32 * LDK + (k8 + 1) * 8);
            }
            // round-robin over the MT*NT independent accumulators: consecutive MFMAs never depend on each other
#define VQ2_MFMA_STEP(C)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j)                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].C, fb[cur][j].C, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(1);
            VQ2_MFMA_STEP(x) VQ2_MFMA_STEP(y) VQ2_MFMA_STEP(z) VQ2_MFMA_STEP(w)
            __builtin_amdgcn_s_setprio(0);
#undef VQ2_MFMA_STEP
        }
    };

    const int nchunks = (P.K + BK - 1) / BK;
    load_chunk(ra0, rb0);
    store_chunk(0, ra0, rb0);
    __syncthreads();
    {
        unsigned long long t_load = 0, t_mfma = 0, t_store = 0, t_bar = 0;
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if constexpr (STAMP) s0 = __builtin_amdgcn_s_memtime();
            if (c + 1 < nchunks) {
                advance_k();
                load_chunk(ra0, rb0);  // global loads in flight while the matrix pipe works on chunk c
            }
            if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); s1 = __builtin_amdgcn_s_memtime(); }
            compute(buf);
            if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); s2 = __builtin_amdgcn_s_memtime(); }
            if (c + 1 < nchunks) store_chunk(buf ^ 1, ra0, rb0);
            if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s3 = __builtin_amdgcn_s_memtime(); }
            __syncthreads();
            if constexpr (STAMP) {
                const unsigned long long s4 = __builtin_amdgcn_s_memtime();
                t_load += s1 - s0; t_mfma += s2 - s1; t_store += s3 - s2; t_bar += s4 - s3;
            }
        }
        if constexpr (STAMP) {
            if (P.stamps && blockIdx.x == 64 && lane == 0) {
                P.stamps[wave * 4 + 0] = t_load; P.stamps[wave * 4 + 1] = t_mfma;
                P.stamps[wave * 4 + 2] = t_store; P.stamps[wave * 4 + 3] = t_bar;
            }
        }
    }

    // ---- epilogue: lane holds column (lane&31) of 16 rows per 32x32 tile.
    // All mask/residual loads of a tile are issued BEFORE its stores: y may alias neither, but the
    // compiler cannot know that, and a load->store->load chain would serialise on HBM latency.
    const float *__restrict__ maskp = P.mask;
    const float *__restrict__ resp = P.res;
    float *__restrict__ yp = P.y;
    const int colq = lane & 31;
    const int rowq = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = n0 + (wn * NT + j) * 32 + colq;
        const bool cv = co < P.Co;
        const float bv = (P.bias && co < P.nbias) ? P.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mb = m0 + (wm * MT + i) * 32 + rowq;
#pragma unroll
            for (int rb = 0; rb < 16; rb += 8) {     // 8 rows at a time keeps the register footprint small
                int pix[8];
                bool ok[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = rb + q;
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    ok[q] = cv && m < P.M;
                    if (os == 1) {
                        pix[q] = m;
                    } else {
                        const int n = m / HoWo;
                        const int rr = m - n * HoWo;
                        const int ho = rr / P.Wo;
                        const int wo = rr - ho * P.Wo;
                        pix[q] = (n * P.Hy + (ho * os + oh)) * P.Wy + (wo * os + ow);
                    }
                }
                float mk[8], rs[8];
                if (maskp) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) mk[q] = ok[q] ? maskp[(size_t)pix[q] * P.ldm + co] : 0.f;
                }
                if (resp) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) rs[q] = ok[q] ? resp[(size_t)pix[q] * P.ldr + co] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = acc[i][j][rb + q] + bv;
                    if (maskp && !P.mask_after) v = (mk[q] > 0.f) ? v : 0.f;
                    if (resp) v += rs[q];
                    if (maskp && P.mask_after) v = (mk[q] > 0.f) ? v : 0.f;
                    v = relu_floor(v, P.relu_out ? 0 : (int)0x80000000);
                    if (ok[q]) yp[(size_t)pix[q] * P.ldy + co] = v;
                }
            }
        }
    }
}


// ====================================================================== low-VALU variant of the kernel
// Cycle stamps (scripts/stamp_conv.py) showed that with two waves per SIMD the partner's fp32 MFMAs
// starve a wave's vector ALU: the ~200 address/bounds instructions per chunk of the kernel above stretch
// from ~1.0k to ~4.8k cycles, and the LDS-store/barrier phases end up unhidden.  This variant does the
// same math with ~1/5 of the VALU work:
//   * operands are fetched with raw BUFFER loads: 32-bit byte offsets, out-of-range lanes get a poisoned
//     offset and the hardware range check returns zeros (no exec-mask branches, no 64-bit address math)
//   * per row: base offset + a row mask and a column mask of in-bounds taps, computed once
//   * per chunk: (tap, ci) advance without loops, tap offset from a small LDS table
//   * epilogue: 32-bit offsets, one add per element; sub-pixel phases share one division per tile row
// Limits (checked by the launcher): KH*KW <= 32, every tensor < 2 GiB.
static int tune(const char *name, int dflt);

// OCC4: four workgroups per CU (BK = 16, <= 128 VGPRs, exactly 40 KiB of LDS: the tap table is replaced by a
// per-thread (kh, kw) counter) -- all 1,024 tiles of a 64x64-resolution layer are then resident at once: ONE
// round, so one exposed prologue and one epilogue burst per launch instead of two.
// UNI (Ci % BK == 0, K % BK == 0: every layer but the 3-channel ones): see the scalar k tracking below.
// CLOCK (diagnostic instantiation only, scripts/stamp_conv.py): four workgroups leave their lifetime in shader cycles
// (s_memtime) and in 10 ns ticks (s_memrealtime) -- the in-kernel clock the chip holds under this kernel's load.
template <int WAVES_M, int WAVES_N, int MT, int NT, int BK, bool PIPE, bool RELU_IN, bool OCC4 = false, bool UNI = OCC4, bool TAPIN = false,
          bool CLOCK = false>
__global__ __launch_bounds__(256, OCC4 ? 4 : ((MT * NT == 4) ? (BK == 16 ? 3 : 2) : 1)) void conv_gemm_fast_kernel(const ConvGemmParams P) {
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if constexpr (CLOCK) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int LDK = BK + 4;
    constexpr int BM = WAVES_M * MT * 32;
    constexpr int BN = WAVES_N * NT * 32;
    constexpr int KQ = BK / 4;
    constexpr int RPP = 256 / KQ;
    constexpr int A_LD = BM / RPP;
    constexpr int B_LD = (BN + RPP - 1) / RPP;
    static_assert(BM % RPP == 0 && (BN % RPP == 0 || BN < RPP), "tile vs staging shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + 2 * BM * LDK;
    int *tap_off = reinterpret_cast<int *>(smem + 2 * (BM + BN) * LDK);  // [32] byte offset of tap (kh,kw)
    int *tap_khw = tap_off + 32;                                        // [32] kh | kw << 8

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntiles = (P.Co + BN - 1) / BN;
    const int mtiles = (P.M + BM - 1) / BM;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (vid % ntiles) * BN;
    const int m0 = ((vid / ntiles) % mtiles) * BM;
    const int phase = vid / (ntiles * mtiles);
    int pad_h = P.pad_h, pad_w = P.pad_w, oh = 0, ow = 0, os = 1;
    const float *wp = P.w;
    if (P.phases == 4) {
        const int ph = phase >> 1, pw = phase & 1;
        pad_h = 1 - ph; pad_w = 1 - pw; oh = ph; ow = pw; os = 2;
        wp += (size_t)phase * P.Co * P.K;
    }
    const int ntaps = P.KH * P.KW;
    if (!OCC4 && tid < ntaps) {
        const int kh = tid / P.KW, kw = tid - kh * P.KW;
        tap_off[tid] = (kh * P.W + kw) * P.ldx * 4;
        tap_khw[tid] = kh | (kw << 8);
    }
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.N * P.H * P.W * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wp), 0, P.Co * P.K * 4, RSRC_FLAGS);

    // ---- per-thread staging coordinates
    const int lrow = tid / KQ;
    const int lk = (tid % KQ) * 4;
    const int HoWo = P.Ho * P.Wo;
    int a_base[A_LD];
    unsigned a_vw[A_LD];   // bit t: tap t = (kh, kw) of this row lies inside the image
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int m = m0 + lrow + RPP * j;
        a_base[j] = 0; a_vw[j] = 0;
        if (m < P.M) {
            const int n = m / HoWo;
            const int r = m - n * HoWo;
            const int ho = r / P.Wo;
            const int wo = r - ho * P.Wo;
            const int hb = ho * P.stride - pad_h, wb = wo * P.stride - pad_w;
            a_base[j] = ((n * P.H + hb) * P.W + wb) * P.ldx * 4;
            unsigned hm = 0, wmk = 0;
            for (int q = 0; q < P.KH; ++q) hm |= ((unsigned)(hb + q) < (unsigned)P.H ? 1u : 0u) << q;
            for (int q = 0; q < P.KW; ++q) wmk |= ((unsigned)(wb + q) < (unsigned)P.W ? 1u : 0u) << q;
            for (int q = 0; q < P.KH; ++q)
                if ((hm >> q) & 1u) a_vw[j] |= wmk << (q * P.KW);
        }
    }
    int b_base[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int co = n0 + lrow + RPP * j;
        b_base[j] = (co < P.Co && lrow + RPP * j < BN) ? co * P.K * 4 : -1;
    }
    int kglob = lk;
    int tap = kglob / P.Ci;
    int ci = kglob - tap * P.Ci;
    const int tstep = BK / P.Ci, cstep = BK - tstep * P.Ci;
    // UNI (Ci % BK == 0): every thread of the workgroup is in the SAME tap and BK-channel block of a chunk, so the
    // tap, its (kh, kw) and the channel base are tracked on the scalar unit; a thread only adds its own 4*lk bytes
    // (folded into a_base / b_base once).  ~25 vector instructions per chunk less next to the MFMAs.
    int u_tap = 0, u_kh = 0, u_kw = 0, u_cb = 0;
    bool b_ok[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) b_ok[j] = b_base[j] >= 0;
    // ... and "out of the image / not a weight row" becomes an ADDITIVE 2^31 (refreshed for the A rows only when the
    // tap changes): base + scalar + penalty is one v_add3 per load, no compare / select.  As an unsigned buffer offset
    // anything >= 2^31 - (tensor bytes) is out of range; the launcher takes this path only for tensors below 1 GiB.
    int a_pen[A_LD];
    auto refresh_pen = [&]() {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) a_pen[j] = ((a_vw[j] >> u_tap) & 1u) ? 0 : (int)0x80000000;
    };
    if (UNI) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) a_base[j] += lk * 4;
#pragma unroll
        for (int j = 0; j < B_LD; ++j) b_base[j] = b_ok[j] ? b_base[j] + lk * 4 : (int)0x80000000;
        refresh_pen();
    }
    __syncthreads();  // tap table visible

    u32x4 ra[A_LD], rb[B_LD];
    // The byte offsets of a chunk's loads are computed one chunk AHEAD (prep_offsets, placed inside the
    // previous chunk's MFMA phase), so that at the top of a chunk the wave only has to issue its 8 loads:
    // no dependent LDS-table/VALU chain sits between the barrier and the first MFMA.
    int off_a[A_LD], off_b[B_LD];
    // K % BK == 0 (every layer of this model): a chunk never runs past K, no tail predicate needed
    const bool k_aligned = (P.K % BK) == 0;
    auto prep_offsets = [&]() {
        if constexpr (UNI) {
            const int koff_s = ((u_kh * P.W + u_kw) * P.ldx + u_cb) * 4;     // scalar
            const int kb_s = (u_tap * P.Ci + u_cb) * 4;                      // scalar
#pragma unroll
            for (int j = 0; j < A_LD; ++j) off_a[j] = a_base[j] + koff_s + a_pen[j];
#pragma unroll
            for (int j = 0; j < B_LD; ++j) off_b[j] = b_base[j] + kb_s;
            return;
        }
        const bool kv = k_aligned || tap < ntaps;
        const int tsel = kv ? tap : 0;
        const int koff = tap_off[tsel] + ci * 4;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const bool v = kv && ((a_vw[j] >> tsel) & 1u) != 0u;
            off_a[j] = v ? a_base[j] + koff : OOB;
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) off_b[j] = (kv && b_base[j] >= 0) ? b_base[j] + kglob * 4 : OOB;
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, off_a[j], 0, 0);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, off_b[j], 0, 0);
    };
    auto load_chunk = [&]() { prep_offsets(); issue_loads(); };
    auto advance_k = [&]() {
        if constexpr (UNI) {
            // Depth order.  With the channel block innermost (k = (tap, ci) as packed) a workgroup sweeps all Ci
            // channels of its 128 pixels per tap and comes back to (nearly) the same lines one tap = Ci/BK chunks later:
            // at four resident workgroups per CU that working set (92 KB per workgroup, 11.8 MB per XCD) falls out of
            // L1 and of the 4 MiB L2, and every 128-byte line crosses the fabric ~2x per tap (measured 658 MB per launch
            // for a 67 MB input).  With the TAPS innermost the nine shifted reads of one BK-channel block follow each
            // other directly (23 KB of whole lines per workgroup, ~2 MB of distinct lines per XCD): most are cache hits.
            // The sum is the same set of products in another order (the panel offset of a chunk is still (tap*Ci + cb)).
            if constexpr (TAPIN) {   // (32-channel block, tap, channel): the block size is fixed so that every tile shape (BK 16
                u_cb += BK;      // or 32) adds the products in the SAME order -- results do not depend on the tile choice
                if ((u_cb & 31) == 0) {
                    u_cb -= 32;
                    ++u_tap;
                    if (++u_kw == P.KW) { u_kw = 0; ++u_kh; }
                    if (u_tap == ntaps) { u_tap = 0; u_kh = 0; u_kw = 0; u_cb += 32; }
                    refresh_pen();
                }
                return;
            }
            u_cb += BK;
            if (u_cb >= P.Ci) {
                u_cb = 0; ++u_tap;
                if (++u_kw == P.KW) { u_kw = 0; ++u_kh; }
                refresh_pen();
            }
            return;
        }
        kglob += BK;
        ci += cstep;
        tap += tstep;
        if (ci >= P.Ci) { ci -= P.Ci; ++tap; }
    };
    auto store_chunk = [&](int buf) {
        float *a = As + buf * BM * LDK;
        float *b = Bs + buf * BN * LDK;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const float4 v = as_f4(ra[j]);
            *reinterpret_cast<float4 *>(a + (lrow + RPP * j) * LDK + lk) = RELU_IN ? relu4(v) : v;   // compile-time: no select
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j)
            if (BN % RPP == 0 || lrow + RPP * j < BN) *reinterpret_cast<float4 *>(b + (lrow + RPP * j) * LDK + lk) = as_f4(rb[j]);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = 4 * (lane >> 5);
    auto compute = [&](int buf, bool prep_next) {
        const float *a = As + buf * BM * LDK + (wm * MT * 32 + frag_row) * LDK + frag_k;
        const float *b = Bs + buf * BN * LDK + (wn * NT * 32 + frag_row) * LDK + frag_k;
        float4 fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = *reinterpret_cast<const float4 *>(a + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = *reinterpret_cast<const float4 *>(b + j * 32 * LDK);
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            const int cur = k8 & 1, nxt = cur ^ 1;
            if (k8 + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[nxt][i] = *reinterpret_cast<const float4 *>(a + i * 32 * LDK + (k8 + 1) * 8);
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[nxt][j] = *reinterpret_cast<const float4 *>(b + j * 32 * LDK + (k8 + 1) * 8);
            }
#define VQ2_MFMA_STEP(C)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < NT; ++j)                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].C, fb[cur][j].C, acc[i][j], 0, 0, 0);
            VQ2_MFMA_STEP(x) VQ2_MFMA_STEP(y) VQ2_MFMA_STEP(z) VQ2_MFMA_STEP(w)
#undef VQ2_MFMA_STEP
            if (PIPE && k8 == 0 && prep_next) { advance_k(); prep_offsets(); }   // offsets of chunk c+2, in the MFMA shadow
        }
    };

    const int nchunks = (P.K + BK - 1) / BK;
    load_chunk();
    store_chunk(0);
    __syncthreads();
    if constexpr (PIPE) {
        if (nchunks > 1) { advance_k(); prep_offsets(); }      // offsets of chunk 1
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) issue_loads();                // chunk c+1 (offsets ready since last iteration)
            compute(buf, c + 2 < nchunks);
            if (c + 1 < nchunks) store_chunk(buf ^ 1);
            __syncthreads();
        }
    } else {
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) {
                advance_k();
                load_chunk();
            }
            compute(buf, false);
            if (c + 1 < nchunks) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue with 32-bit offsets through buffer descriptors
    const int ybytes = P.N * P.Hy * P.Wy * 4;  // per unit of pixel stride
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, ybytes * P.ldy, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rmk =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.mask ? P.mask : P.y), 0, P.mask ? ybytes * P.ldm : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.res ? P.res : P.y), 0, P.res ? ybytes * P.ldr : 0, RSRC_FLAGS);
    const bool has_mask = P.mask != nullptr, has_res = P.res != nullptr;
    const bool mask_first = has_mask && !P.mask_after, mask_last = has_mask && P.mask_after;
    const int colq = lane & 31;
    const int rowq = 4 * (lane >> 5);
    const int ldy4 = P.ldy * 4, ldm4 = P.ldm * 4, ldr4 = P.ldr * 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int mt0 = m0 + (wm * MT + i) * 32;   // first GEMM row of this 32-row tile
        // pixel index of row (lane & 31) of the tile: identity for a dense grid, one decomposition per
        // lane (shared through ds_bpermute) for the strided sub-pixel phases
        int pix_lane;
        {
            const int m = mt0 + colq;
            if (os == 1) {
                pix_lane = m;
            } else {
                const int n = m / HoWo;
                const int rr = m - n * HoWo;
                const int ho = rr / P.Wo;
                const int wo = rr - ho * P.Wo;
                pix_lane = (n * P.Hy + (ho * 2 + oh)) * P.Wy + (wo * 2 + ow);
            }
            if (m >= P.M) pix_lane = -1;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = n0 + (wn * NT + j) * 32 + colq;
            const bool cv = co < P.Co;
            const float bv = (P.bias && co < P.nbias) ? P.bias[co] : 0.f;
            const int co4 = co * 4;
#pragma unroll
            for (int rb8 = 0; rb8 < 16; rb8 += 8) {
                int pix[8];
                float mk[8], rs[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = rb8 + q;
                    const int rr = rowq + (r & 3) + 8 * (r >> 2);
                    pix[q] = (os == 1) ? ((mt0 + rr < P.M) ? mt0 + rr : -1) : __shfl(pix_lane, rr, 64);
                    if (!cv) pix[q] = -1;
                }
                if (has_mask) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        mk[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmk, pix[q] >= 0 ? pix[q] * ldm4 + co4 : OOB, 0, 0));
                }
                if (has_res) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        rs[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, pix[q] >= 0 ? pix[q] * ldr4 + co4 : OOB, 0, 0));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = acc[i][j][rb8 + q] + bv;
                    if (mask_first) v = (mk[q] > 0.f) ? v : 0.f;
                    if (has_res) v += rs[q];
                    if (mask_last) v = (mk[q] > 0.f) ? v : 0.f;
                    v = relu_floor(v, P.relu_out ? 0 : (int)0x80000000);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, pix[q] >= 0 ? pix[q] * ldy4 + co4 : OOB, 0, 0);
                }
            }
        }
    }
    if constexpr (CLOCK) {
        if (P.stamps && tid == 0 && (blockIdx.x & 255) == 8 && blockIdx.x < 1024) {
            const int slot = blockIdx.x >> 8;
            P.stamps[slot * 4 + 0] = __builtin_amdgcn_s_memtime() - clk_t0;
            P.stamps[slot * 4 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
            P.stamps[slot * 4 + 2] = (unsigned long long)nchunks * (BK / 2) * MT * NT;   // MFMAs of one wave
        }
    }
}

static unsigned long long *g_stamps = nullptr;  // set by vq2_debug_set_stamps: diagnostic cycle stamps

template <int WAVES_M, int WAVES_N, int MT, int NT, int BK, bool OCC4 = false>
static int launch_conv_gemm_fast(const ConvGemmParams &P, hipStream_t s) {
    constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32;
    const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float) + (OCC4 ? 0 : 64 * sizeof(int));
    const unsigned nwg = ((P.M + BM - 1) / BM) * ((P.Co + BN - 1) / BN) * P.phases;
    // uniform k tracking whenever a chunk never straddles two taps (every layer except the 3-channel ones)
    const long gib = 1L << 30;
    const bool small = (long)P.N * P.H * P.W * P.ldx * 4 < gib && (long)P.Co * P.K * P.phases * 4 < gib;
    const bool uni = small && (OCC4 || (P.Ci % BK == 0 && P.K % BK == 0));
    static const int tapin = tune("VQ2_TAPIN", 1);
    const bool tap_inner = uni && tapin && P.KH * P.KW > 1 && P.Ci > 32 && P.Ci % 32 == 0;
    auto kern = P.relu_in ? (uni ? (tap_inner ? conv_gemm_fast_kernel<WAVES_M, WAVES_N, MT, NT, BK, true, true, OCC4, true, true>
                                              : conv_gemm_fast_kernel<WAVES_M, WAVES_N, MT, NT, BK, true, true, OCC4, true, false>)
                                 : conv_gemm_fast_kernel<WAVES_M, WAVES_N, MT, NT, BK, true, true, OCC4, OCC4>)
                          : (uni ? (tap_inner ? conv_gemm_fast_kernel<WAVES_M, WAVES_N, MT, NT, BK, true, false, OCC4, true, true>
                                              : conv_gemm_fast_kernel<WAVES_M, WAVES_N, MT, NT, BK, true, false, OCC4, true, false>)
                                 : conv_gemm_fast_kernel<WAVES_M, WAVES_N, MT, NT, BK, true, false, OCC4, OCC4>);
    ConvGemmParams Q = P;
    if constexpr (OCC4) {   // in-kernel clock probe of the dominant instantiation (vq2_debug_set_stamps + VQ2_CLOCKPROBE=1)
        static const int probe = tune("VQ2_CLOCKPROBE", 0);
        if (probe && g_stamps && tap_inner && !P.relu_in) {
            kern = conv_gemm_fast_kernel<WAVES_M, WAVES_N, MT, NT, BK, true, false, OCC4, true, true, true>;
            Q.stamps = g_stamps;
        }
    }
    allow_big_lds(kern, lds);
    dim3 grid(nwg);
    const char *name = "conv_gemm";
    if (prof_enabled())
        name = prof_label("conv_gemm<%dx%dx%d>|M=%d,N=%d,K=%d,k%d,s%d,ph%d", BM, BN, BK, P.M, P.Co, P.K, P.KH, P.stride,
                          P.phases);
    ProfScope prof(name, P.flops, P.bytes, s, BM == 128 && BN == 128);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, Q);
    return check_launch("conv_gemm_fast_kernel");
}

template <int WAVES_M, int WAVES_N, int MT, int NT, int BK, bool STAMP = false>
static int launch_conv_gemm(const ConvGemmParams &P, hipStream_t s) {
    constexpr int BM = WAVES_M * MT * 32, BN = WAVES_N * NT * 32;
    static const int lds_pad = getenv("VQ2_LDS_PAD") ? atoi(getenv("VQ2_LDS_PAD")) : 0;  // experiments: force 1 WG/CU
    const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float) + lds_pad;
    auto kern = conv_gemm_kernel<WAVES_M, WAVES_N, MT, NT, BK, STAMP>;
    allow_big_lds(kern, lds);
    dim3 grid(((P.M + BM - 1) / BM) * ((P.Co + BN - 1) / BN) * P.phases);
    const char *name = "conv_gemm";
    if (prof_enabled())
        name = prof_label("conv_gemm<%dx%dx%d>|M=%d,N=%d,K=%d,k%d,s%d,ph%d", BM, BN, BK, P.M, P.Co, P.K, P.KH, P.stride,
                          P.phases);
    ProfScope prof(name, P.flops, P.bytes, s, BM == 128 && BN == 128 && BK == 32);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, P);
    return check_launch("conv_gemm_kernel");
}

// ====================================================================== sub-pixel conv from an LDS patch
// ConvTranspose2d(k4,s2,p1) forward and the data gradient of a k4,s2,p1 conv: four output phases, each a 2x2
// stride-1 conv over the SAME 3x3 input neighbourhood.  As four independent GEMM launches-in-one (the kernel
// above with phases = 4) every phase re-reads the input tensor (measured 640 MB of HBM fetch per launch for a
// 67 MB input).  Here a workgroup owns an 8x16 tile of the INPUT grid and 64 output channels: the 10x18 halo
// patch of a 16-channel slice is staged into LDS once and feeds all 16 (phase, tap) products by shifting the
// fragment base; only the 2x2 weight panel of one phase is streamed per step (64 MFMAs per wave between
// barriers, 4.75 staging loads instead of 8).  Accumulators: 4 phases x (32 px x 64 co) per wave = 128 VGPRs.
namespace sp {
constexpr int TH = 8, TW = 16, PW = TW + 2, NPATCH = (TH + 2) * PW;   // 180 patch pixels
constexpr int CS = 16, LDK = CS + 4;
constexpr int A_F4 = NPATCH * CS / 4;        // 720
constexpr int A_FLOATS = NPATCH * LDK;       // 3600
constexpr int B_ROWS = 4 * 64;               // (tap, co) rows of one phase panel
constexpr int B_FLOATS = B_ROWS * LDK;       // 5120
constexpr size_t LDS_BYTES = (size_t)2 * (A_FLOATS + B_FLOATS) * sizeof(float);   // 69,760: two workgroups per CU
constexpr int SOOB = 0x7F000000;
}  // namespace sp

template <bool RELU_IN>
__global__ __launch_bounds__(256, 2) void subpixel_conv_kernel(const ConvGemmParams P) {
    using namespace sp;
    constexpr int A_LD = (A_F4 + 255) / 256;   // 3
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                    // [2][A_FLOATS]
    float *Bs = smem + 2 * A_FLOATS;     // [2][B_FLOATS]
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;
    const int l31 = lane & 31, fk = 4 * (lane >> 5), rowq = 4 * (lane >> 5);
    const int ncg = P.Co / 64;
    const int tiles_x = (P.W + TW - 1) / TW, tiles_y = (P.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = vid % ncg;
    const int tv = vid / ncg;
    const int n = tv / tiles;
    const int t = tv - n * tiles;
    const int tyi = t / tiles_x;
    const int y0 = tyi * TH, x0 = (t - tyi * tiles_x) * TW;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.N * P.H * P.W * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w), 0, 4 * P.Co * P.K * 4, RSRC_FLAGS);

    int a_off[A_LD], b_off[4];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int f = tid + 256 * j;
        a_off[j] = SOOB;
        if (f < A_F4) {
            const int pp = f >> 2;
            const int pr = pp / PW, pc = pp - pr * PW;
            const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
            if ((unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W)
                a_off[j] = ((n * P.H + gy) * P.W + gx) * P.ldx * 4 + (f & 3) * 16;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (tid >> 2) + 64 * j;     // tap * 64 + co
        b_off[j] = ((cg * 64 + (row & 63)) * P.K + (row >> 6) * P.Ci) * 4 + (tid & 3) * 16;
    }
    const int st_off = (tid >> 2) * LDK + (tid & 3) * 4;
    const int phase_stride = P.Co * P.K * 4;     // bytes between the panels of two phases
    u32x4 ra[A_LD], rb[4];
    auto issue_a = [&](int s) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, a_off[j], s * CS * 4, 0);
    };
    auto issue_b = [&](int s, int ph) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_off[j], ph * phase_stride + s * CS * 4, 0);
    };
    auto store_a = [&](int buf) {
        float *a = As + buf * A_FLOATS + st_off;
#pragma unroll
        for (int j = 0; j < A_LD; ++j)
            if ((j + 1) * 256 <= A_F4 || tid + 256 * j < A_F4) {
                const float4 v = as_f4(ra[j]);
                *reinterpret_cast<float4 *>(a + j * 64 * LDK) = RELU_IN ? relu4(v) : v;
            }
    };
    auto store_b = [&](int buf) {
        float *b = Bs + buf * B_FLOATS + st_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(b + j * 64 * LDK) = as_f4(rb[j]);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][j][r] = 0.f;

    // GEMM row l31 of this wave = tile pixel (2*wq + l31/16, tx): second row rotated by 14 columns so that every
    // tap's ds_read_b128 hits the LDS slots like 32 consecutive rows (see vq2_resblock.hip)
    const int txr = l31 < 16 ? l31 : ((l31 + 14) & 15);
    const int a_frag = ((2 * wq + (l31 >> 4)) * PW + txr) * LDK + fk;
    const int b_frag = l31 * LDK + fk;
    const int NS = P.Ci / CS;

    issue_a(0);
    issue_b(0, 0);
    store_a(0);
    store_b(0);
    __syncthreads();
    for (int s = 0; s < NS; ++s) {
        const float *Ab = As + (s & 1) * A_FLOATS + a_frag;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool last = (s + 1 == NS) && p == 3;
            if (!last) issue_b(p == 3 ? s + 1 : s, (p + 1) & 3);
            if (p == 0 && s + 1 < NS) issue_a(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            {
                const float *Bb = Bs + (p & 1) * B_FLOATS + b_frag;
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const float *a = Ab + (((p >> 1) + (tap >> 1)) * PW + (p & 1) + (tap & 1)) * LDK;
                    const float *b = Bb + tap * 64 * LDK;
#pragma unroll
                    for (int k8 = 0; k8 < CS / 8; ++k8) {
                        const float4 fa = *reinterpret_cast<const float4 *>(a + 8 * k8);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float4 fb = *reinterpret_cast<const float4 *>(b + j * 32 * LDK + 8 * k8);
                            acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[p][j], 0, 0, 0);
                            acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[p][j], 0, 0, 0);
                            acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[p][j], 0, 0, 0);
                            acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[p][j], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!last) store_b((p + 1) & 1);
            if (p == 3 && s + 1 < NS) store_a((s + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: output pixel (2*gy + ph, 2*gx + pw) of phase p = 2*ph + pw
    const int ybytes = P.N * P.Hy * P.Wy * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, ybytes * P.ldy, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rmk =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.mask ? P.mask : P.y), 0, P.mask ? ybytes * P.ldm : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.res ? P.res : P.y), 0, P.res ? ybytes * P.ldr : 0, RSRC_FLAGS);
    const bool has_mask = P.mask != nullptr, has_res = P.res != nullptr;
    const bool mask_first = has_mask && !P.mask_after, mask_last = has_mask && P.mask_after;
    int pb_lane;   // output pixel of phase (0,0) for GEMM row l31, or -1
    {
        const int gy = y0 + 2 * wq + (l31 >> 4), gx = x0 + txr;
        pb_lane = (gy < P.H && gx < P.W) ? (n * P.Hy + 2 * gy) * P.Wy + 2 * gx : -1;
    }
    const int ldy4 = P.ldy * 4, ldm4 = P.ldm * 4, ldr4 = P.ldr * 4;
#pragma unroll
    for (int rb8 = 0; rb8 < 16; rb8 += 8) {
        int pb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = rb8 + q;
            pb[q] = __shfl(pb_lane, rowq + (r & 3) + 8 * (r >> 2), 64);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int poff = (p >> 1) * P.Wy + (p & 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = cg * 64 + j * 32 + l31;
                const float bv = (P.bias && co < P.nbias) ? P.bias[co] : 0.f;
                const int co4 = co * 4;
                float mk[8], rs[8];
                if (has_mask) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        mk[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmk, pb[q] >= 0 ? (pb[q] + poff) * ldm4 + co4 : SOOB, 0, 0));
                }
                if (has_res) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        rs[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, pb[q] >= 0 ? (pb[q] + poff) * ldr4 + co4 : SOOB, 0, 0));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = acc[p][j][rb8 + q] + bv;
                    if (mask_first) v = (mk[q] > 0.f) ? v : 0.f;
                    if (has_res) v += rs[q];
                    if (mask_last) v = (mk[q] > 0.f) ? v : 0.f;
                    v = relu_floor(v, P.relu_out ? 0 : (int)0x80000000);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, pb[q] >= 0 ? (pb[q] + poff) * ldy4 + co4 : SOOB, 0, 0);
                }
            }
        }
    }
}

static int launch_subpixel(const ConvGemmParams &P, hipStream_t s) {
    const int tiles = ((P.W + sp::TW - 1) / sp::TW) * ((P.H + sp::TH - 1) / sp::TH);
    dim3 grid(P.N * tiles * (P.Co / 64));
    const char *name = "subpixel_conv";
    if (prof_enabled()) name = prof_label("subpixel_conv|M=%d,N=%d,K=%d,ph4", P.M, P.Co, P.K);
    ProfScope prof(name, P.flops, P.bytes, s);
    auto kern = P.relu_in ? subpixel_conv_kernel<true> : subpixel_conv_kernel<false>;
    allow_big_lds(kern, sp::LDS_BYTES);
    hipLaunchKernelGGL(kern, grid, dim3(256), sp::LDS_BYTES, s, P);
    return check_launch("subpixel_conv_kernel");
}

// ====================================================================== k4 s2 p1 conv out of 4 (3 + pad) channels
// The image-side layers -- Conv2d(3 -> 64, k4 s2 p1) forward (vqvae.py:105) and the data gradient of the
// reconstruction ConvTranspose2d(64 -> 3) (vqvae.py:157), which is the same strided conv of the 4-channel dy -- have a
// depth of only K = 16 taps x 4 channels = 64: as an implicit GEMM with staged chunks they are all prologue and
// epilogue (2.2-2.8 TB/s of algorithmic bytes on the generic tile).  They are HBM-bound (AI ~ 20): this kernel keeps
// the whole 64 x 64 weight panel in registers, stages the 18 x 34-pixel input patch of an 8 x 16 output tile in LDS
// (one 16-byte pixel = one tap of the depth: a fragment read IS the im2col), walks four tiles per workgroup with the
// next tile's patch in flight behind the current tile's MFMAs, and swaps the operand roles (weights = MFMA rows,
// pixels = columns) so that a lane ends up with 4 CONSECUTIVE output channels of one pixel: bias, mask and store are
// 16-byte accesses (8 stores per lane instead of 32).
namespace c4 {
constexpr int TH = 8, TW = 16;                    // output pixels per tile
constexpr int PH = 2 * TH + 2, PW = 2 * TW + 2;   // input patch
constexpr int NP = PH * PW;                       // 612 pixels of 16 bytes
constexpr int P_LD = (NP + 255) / 256;            // 3 patch loads per thread
constexpr int COOB = 0x7F000000;
}  // namespace c4

template <bool HAS_MASK, bool SWAP, bool C4>
__global__ __launch_bounds__(256, 3) void conv_k4s2_c4_kernel(const ConvGemmParams P) {
    using namespace c4;
    __shared__ float4 patch[NP];
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = (P.Wo + TW - 1) / TW, tiles_y = (P.Ho + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y, total = P.N * tiles;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.N * P.H * P.W * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w), 0, 64 * 64 * 4, RSRC_FLAGS);
    const int ybytes = P.N * P.Ho * P.Wo * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, ybytes * P.ldy, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rmk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(HAS_MASK ? P.mask : P.y), 0,
                                                                         HAS_MASK ? ybytes * P.ldm : 0, RSRC_FLAGS);
    // weight fragments (MFMA A operand): row co = 32*j + l31, depth k = 8*k8 + 4*h .. +3  (panel [Co][tap][Ci] = [64][64])
    float4 wf[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8)
            wf[j][k8] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rw, ((j * 32 + l31) * 64 + 8 * k8 + 4 * h) * 4, 0, 0));
    // bias (zero-padded to 64) in LDS: the epilogue reads this lane's 4 consecutive channels as one float4
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    if (tid < 64) bias_s[tid] = (P.bias && tid < P.nbias) ? P.bias[tid] : 0.f;
    const bool relu_in = P.relu_in != 0;
    u32x4 pr[P_LD];
    auto tile_of = [&](int v, int &n, int &y0, int &x0) {
        n = v / tiles;
        const int t = v - n * tiles;
        const int tyi = t / tiles_x;
        y0 = tyi * TH; x0 = (t - tyi * tiles_x) * TW;
    };
    auto issue_patch = [&](int v) {
        int n, y0, x0;
        tile_of(v, n, y0, x0);
#pragma unroll
        for (int q = 0; q < P_LD; ++q) {
            const int f = tid + 256 * q;
            const int prow = f / PW, pcol = f - prow * PW;
            const int gy = 2 * y0 - 1 + prow, gx = 2 * x0 - 1 + pcol;
            const bool ok = v < total && f < NP && (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W;
            pr[q] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? ((n * P.H + gy) * P.W + gx) * P.ldx * 4 : COOB, 0, 0);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int q = 0; q < P_LD; ++q) {
            const int f = tid + 256 * q;
            if (f < NP) patch[f] = relu_in ? relu4(as_f4(pr[q])) : as_f4(pr[q]);
        }
    };
    const int TPW = P.c4_tpw;
    const int v0 = xcd_remap(blockIdx.x, gridDim.x) * TPW;   // consecutive tiles of one image row band per workgroup
    issue_patch(v0);
    const int ty = 2 * wq + (l31 >> 4), tx = l31 & 15;       // this lane's output pixel inside a tile (MFMA column l31)
    for (int it = 0; it < TPW; ++it) {
        const int v = v0 + it;
        if (v >= total) break;
        store_patch();
        __syncthreads();
        if (it + 1 < TPW) issue_patch(v + 1);                 // in flight behind this tile's MFMAs and stores
        int n, y0, x0;
        tile_of(v, n, y0, x0);
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        u32x4 mk[2][4];
        if (HAS_MASK && SWAP) {    // the mask of this lane's 16 outputs, requested BEHIND the MFMAs instead of after them
            const int oy = y0 + ty, ox = x0 + tx;
            const bool pv = oy < P.Ho && ox < P.Wo;
            const int pix = (n * P.Ho + oy) * P.Wo + ox;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    mk[j][g] = __builtin_amdgcn_raw_buffer_load_b128(rmk, pv ? pix * P.ldm * 4 + (32 * j + 8 * g + 4 * h) * 4 : COOB, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            // depth 8*k8 + 4*h .. +3 = the 4 channels of tap 2*k8 + h: kh = k8/2, kw = 2*(k8&1) + h
            const float4 a = patch[(2 * ty + (k8 >> 1)) * PW + 2 * tx + 2 * (k8 & 1) + h];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // the 4th channel is zero padding on both operands when the layer has 3 real channels: its product
                // adds an exact 0 and is skipped (a quarter of the matrix work)
                if (SWAP) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][k8].x, a.x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][k8].y, a.y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][k8].z, a.z, acc[j], 0, 0, 0);
                    if (C4) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j][k8].w, a.w, acc[j], 0, 0, 0);
                } else {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wf[j][k8].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wf[j][k8].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wf[j][k8].z, acc[j], 0, 0, 0);
                    if (C4) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wf[j][k8].w, acc[j], 0, 0, 0);
                }
            }
        }
        const int relu_floor_bits = P.relu_out ? 0 : (int)0x80000000;
        if (SWAP) {
            // epilogue: lane = pixel (ty, tx); registers 4g..4g+3 of block j = channels 32*j + 8*g + 4*h + (0..3)
            const int oy = y0 + ty, ox = x0 + tx;
            const bool pv = oy < P.Ho && ox < P.Wo;
            const int pix = (n * P.Ho + oy) * P.Wo + ox;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co4 = (32 * j + 8 * g + 4 * h) * 4;
                    const float4 bv = *reinterpret_cast<const float4 *>(bias_s + 32 * j + 8 * g + 4 * h);
                    float4 o = make_float4(acc[j][4 * g] + bv.x, acc[j][4 * g + 1] + bv.y, acc[j][4 * g + 2] + bv.z,
                                           acc[j][4 * g + 3] + bv.w);
                    if (HAS_MASK) {
                        const float4 m = as_f4(mk[j][g]);
                        o.x = m.x > 0.f ? o.x : 0.f; o.y = m.y > 0.f ? o.y : 0.f; o.z = m.z > 0.f ? o.z : 0.f; o.w = m.w > 0.f ? o.w : 0.f;
                    }
                    o.x = relu_floor(o.x, relu_floor_bits); o.y = relu_floor(o.y, relu_floor_bits);
                    o.z = relu_floor(o.z, relu_floor_bits); o.w = relu_floor(o.w, relu_floor_bits);
                    u32x4 u;
                    u.x = __float_as_uint(o.x); u.y = __float_as_uint(o.y); u.z = __float_as_uint(o.z); u.w = __float_as_uint(o.w);
                    __builtin_amdgcn_raw_buffer_store_b128(u, ry, pv ? pix * P.ldy * 4 + co4 : COOB, 0, 0);
                }
        } else {
            // lane = channel l31 (+32 j); register r = pixel 4*h + (r&3) + 8*(r>>2) of this wave's 32
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = 4 * h + (r & 3) + 8 * (r >> 2);
                const int oy = y0 + 2 * wq + (pp >> 4), ox = x0 + (pp & 15);
                const bool pv = oy < P.Ho && ox < P.Wo;
                const int pix = (n * P.Ho + oy) * P.Wo + ox;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int co4 = (32 * j + l31) * 4;
                    float v = acc[j][r] + bias_s[32 * j + l31];
                    if (HAS_MASK) {
                        const float m = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmk, pv ? pix * P.ldm * 4 + co4 : COOB, 0, 0));
                        v = m > 0.f ? v : 0.f;
                    }
                    v = relu_floor(v, relu_floor_bits);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, pv ? pix * P.ldy * 4 + co4 : COOB, 0, 0);
                }
            }
        }
        __syncthreads();   // every wave is done reading the patch before the next tile overwrites it
    }
}

// ====================================================================== 1x1 conv out of 64 channels, row-streaming
// (the data gradient of quantize_conv_b, vqvae.py:189: 64 -> 192 channels at 64x64, and the other 1x1 layers fed by
// embed_dim = 64.)  HBM-bound: one 256-byte row in, Co * 4 bytes out per pixel, 2 * 64 * Co FLOP.  As a GEMM tile kernel
// (64 x 192 x 16: four chunks, then 96 stores per lane) a workgroup is a load -> compute -> store chain too short to overlap
// with itself: 55 us stand-alone for 134 MB (2.4 TB/s), 85 us inside the step.  Here the [Co][64] panel is staged ONCE per
// workgroup of eight waves (one per CU: 52 KB of panel + 8 x 8.7 KB of row tiles), every wave streams 32-pixel row blocks of
// its own through a private LDS tile (no workgroup barrier in the loop; LDS operations of one wave execute in order, so the
// tile needs no second buffer: the next block's loads are in flight while this block's MFMAs and stores run) and walks a
// strided range of blocks.  Epilogue = the generic one (bias, ReLU mask, residual, ReLU).
namespace k64 {
constexpr int KC = 64, LDW = KC + 4;               // 272-byte LDS rows: conflict-free ds_read_b128
constexpr int X_FLOATS = 32 * LDW;                 // one wave's row block
constexpr int NWAVES = 8;                          // 512 threads: ONE workgroup per CU shares the panel, two waves per SIMD
template <int NB> constexpr size_t lds_bytes() { return (size_t)(NB * 32 * LDW + NWAVES * X_FLOATS) * sizeof(float); }
}  // namespace k64

template <int NB, bool RELU_IN>     // NB = Co / 32 column blocks (<= 6)
__global__ __launch_bounds__(512, 2) void conv1x1_k64_kernel(const ConvGemmParams P) {
    using namespace k64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ws = smem;                                        // [NB*32][LDW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *Xs = smem + NB * 32 * LDW + wave * X_FLOATS;      // this wave's [32][LDW]

    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, P.M * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w), 0, P.Co * KC * 4, RSRC_FLAGS);
    // the panel: NB * 32 rows x 16 float4
    for (int it = tid; it < NB * 32 * 16; it += 64 * NWAVES) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, it * 16, 0, 0);
        *reinterpret_cast<float4 *>(Ws + (it >> 4) * LDW + (it & 15) * 4) = as_f4(v);
    }
    // row blocks of this wave: gw, gw + nwaves, ...
    const int nblk = (P.M + 31) / 32, gw = blockIdx.x * NWAVES + wave, nwaves = gridDim.x * NWAVES;
    // staging: lane -> (row = (lane >> 4) + 4 j, quad = lane & 15), 8 loads per block
    const int srow = lane >> 4, sq = lane & 15;
    u32x4 rx8[8];
    auto load_block = [&](int b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = b * 32 + srow + 4 * j;
            rx8[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (b < nblk && m < P.M) ? (m * P.ldx + 4 * sq) * 4 : OOB, 0, 0);
        }
    };
    auto store_block = [&](float *xs) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = as_f4(rx8[j]);
            *reinterpret_cast<float4 *>(xs + (srow + 4 * j) * LDW + 4 * sq) = RELU_IN ? relu4(v) : v;
        }
    };
    const int ybytes = P.M * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, ybytes * P.ldy, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rmk =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.mask ? P.mask : P.y), 0, P.mask ? ybytes * P.ldm : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.res ? P.res : P.y), 0, P.res ? ybytes * P.ldr : 0, RSRC_FLAGS);
    const bool has_mask = P.mask != nullptr, has_res = P.res != nullptr;
    const bool mask_first = has_mask && !P.mask_after, mask_last = has_mask && P.mask_after;
    const int colq = lane & 31, rowq = 4 * (lane >> 5), fk = 4 * (lane >> 5);
    const int ldy4 = P.ldy * 4, ldm4 = P.ldm * 4, ldr4 = P.ldr * 4;
    const int relu_bits = P.relu_out ? 0 : (int)0x80000000;
    float bv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[j] = (P.bias && j * 32 + colq < P.nbias) ? P.bias[j * 32 + colq] : 0.f;

    load_block(gw);
    __syncthreads();                       // the panel (the only workgroup-wide dependency)
    for (int b = gw; b < nblk; b += nwaves) {
        float *xs = Xs;
        store_block(xs);                   // wave-private tile: program order + the LDS counter are the only sync needed
        load_block(b + nwaves);            // next block of this wave, in flight behind the MFMAs and stores below
        f32x16 acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float *a = xs + colq * LDW + fk;
        const float *w = Ws + colq * LDW + fk;
#pragma unroll
        for (int k8 = 0; k8 < KC / 8; ++k8) {
            const float4 fa = *reinterpret_cast<const float4 *>(a + 8 * k8);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float4 fb = *reinterpret_cast<const float4 *>(w + j * 32 * LDW + 8 * k8);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[j], 0, 0, 0);
            }
        }
        const int m0 = b * 32;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int co4 = (j * 32 + colq) * 4;
#pragma unroll
            for (int rb8 = 0; rb8 < 16; rb8 += 8) {
                int pix[8];
                float mk[8], rs[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = rb8 + q;
                    const int m = m0 + rowq + (r & 3) + 8 * (r >> 2);
                    pix[q] = m < P.M ? m : -1;
                }
                if (has_mask) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        mk[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmk, pix[q] >= 0 ? pix[q] * ldm4 + co4 : OOB, 0, 0));
                }
                if (has_res) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        rs[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, pix[q] >= 0 ? pix[q] * ldr4 + co4 : OOB, 0, 0));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v = acc[j][rb8 + q] + bv[j];
                    if (mask_first) v = (mk[q] > 0.f) ? v : 0.f;
                    if (has_res) v += rs[q];
                    if (mask_last) v = (mk[q] > 0.f) ? v : 0.f;
                    v = relu_floor(v, relu_bits);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, pix[q] >= 0 ? pix[q] * ldy4 + co4 : OOB, 0, 0);
                }
            }
        }
    }
}

static bool conv_k64_ok(const ConvGemmParams &P) {
    static const int on = getenv("VQ2_K64") ? atoi(getenv("VQ2_K64")) : 1;
    const long gib = 1L << 30;
    return on && P.phases == 1 && P.KH == 1 && P.KW == 1 && P.stride == 1 && P.pad_h == 0 && P.pad_w == 0 && P.Ci == 64 &&
           P.K == 64 && P.Co % 32 == 0 && P.Co >= 64 && P.Co <= 192 && P.M >= 16384 && P.ldx % 4 == 0 &&
           (long)P.M * P.ldx * 4 < gib && (long)P.M * P.ldy * 4 < gib && (long)P.M * (P.ldm > P.ldr ? P.ldm : P.ldr) * 4 < gib;
}

template <int NB>
static int launch_conv_k64_nb(const ConvGemmParams &P, hipStream_t s) {
    auto kern = P.relu_in ? conv1x1_k64_kernel<NB, true> : conv1x1_k64_kernel<NB, false>;
    const size_t lds = k64::lds_bytes<NB>();
    allow_big_lds(kern, lds);
    const int nblk = (P.M + 31) / 32;
    int grid = 256;                               // one 8-wave workgroup per CU, each wave walks nblk / 2048 row blocks
    if (grid * k64::NWAVES > nblk) grid = (nblk + k64::NWAVES - 1) / k64::NWAVES;
    const char *name = "conv1x1_k64";
    if (prof_enabled()) name = prof_label("conv1x1_k64|M=%d,N=%d,K=64", P.M, P.Co);
    ProfScope prof(name, P.flops, P.bytes, s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * k64::NWAVES), lds, s, P);
    return check_launch("conv1x1_k64_kernel");
}

static int launch_conv_k64(const ConvGemmParams &P, hipStream_t s) {
    switch (P.Co / 32) {
        case 2: return launch_conv_k64_nb<2>(P, s);
        case 3: return launch_conv_k64_nb<3>(P, s);
        case 4: return launch_conv_k64_nb<4>(P, s);
        case 5: return launch_conv_k64_nb<5>(P, s);
        default: return launch_conv_k64_nb<6>(P, s);
    }
}

static bool conv_c4_ok(const ConvGemmParams &P) {
    const long big = 0x7F000000L / 4;
    return P.phases == 1 && P.KH == 4 && P.KW == 4 && P.stride == 2 && P.pad_h == 1 && P.pad_w == 1 && P.Ci == 4 &&
           P.Co == 64 && !P.res && P.H % 2 == 0 && P.W % 2 == 0 && P.ldy % 4 == 0 && (!P.mask || P.ldm % 4 == 0) &&
           (long)P.N * P.H * P.W * P.ldx < big && (long)P.N * P.Ho * P.Wo * (P.ldy > P.ldm ? P.ldy : P.ldm) < big &&
           (reinterpret_cast<uintptr_t>(P.y) & 15u) == 0 && (!P.mask || (reinterpret_cast<uintptr_t>(P.mask) & 15u) == 0);
}

static int launch_conv_c4(const ConvGemmParams &P, hipStream_t s) {
    const int tiles = ((P.Wo + c4::TW - 1) / c4::TW) * ((P.Ho + c4::TH - 1) / c4::TH) * P.N;
    const char *name = "conv_k4s2_c4";
    if (prof_enabled()) name = prof_label("conv_k4s2_c4|M=%d,N=%d,K=%d%s", P.M, P.Co, P.K, P.mask ? ",mask" : "");
    ProfScope prof(name, P.flops, P.bytes, s);
    // measured (scripts/microbench.py c4s2_3_64 / t_64_3; generic tile: 75 / 88 us): the plain layer streams best with
    // coalesced 4-byte stores and 4 tiles per workgroup (45 us, 3.7 TB/s of algorithmic bytes); the masked one with the
    // 16-byte form whose mask loads sit behind the MFMAs and 8 tiles per workgroup (72 us, 4.2 TB/s)
    static const int swap_m = tune("VQ2_C4_SWAP_MASK", 1), swap_p = tune("VQ2_C4_SWAP", 0);
    static const int tpw_m = tune("VQ2_C4_TPW_MASK", 8), tpw_p = tune("VQ2_C4_TPW", 4);
    const bool swap = P.mask ? swap_m : swap_p;
    const int tpw = P.mask ? tpw_m : tpw_p;
    const bool c4 = !(P.ci_real > 0 && P.ci_real <= 3);
    auto pick = [&](auto m, auto sw) {
        constexpr bool M = decltype(m)::value, S = decltype(sw)::value;
        return c4 ? conv_k4s2_c4_kernel<M, S, true> : conv_k4s2_c4_kernel<M, S, false>;
    };
    auto kern = P.mask ? (swap ? pick(std::true_type{}, std::true_type{}) : pick(std::true_type{}, std::false_type{}))
                       : (swap ? pick(std::false_type{}, std::true_type{}) : pick(std::false_type{}, std::false_type{}));
    ConvGemmParams Q = P;
    Q.c4_tpw = tpw < 1 ? 1 : tpw;
    hipLaunchKernelGGL(kern, dim3((tiles + Q.c4_tpw - 1) / Q.c4_tpw), dim3(256), 0, s, Q);
    return check_launch("conv_k4s2_c4_kernel");
}

static int tune(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// vq2_debug_set_stamps without VQ2_CLOCKPROBE: the phase-stamp build of the older 128x128x32 kernel takes every launch
static bool legacy_stamps() {
    static const int probe = tune("VQ2_CLOCKPROBE", 0);
    return g_stamps != nullptr && !probe;
}

static int run_conv_gemm(const ConvGemmParams &P, hipStream_t s) {
    // Tile by output-channel count (GEMM N).  Chunk depth per tile measured on MI355X: the 128x128 tile is
    // register-bound at 2 waves/SIMD and prefers BK=32; the narrower tiles run 4+ waves/SIMD with BK=16.
    static const int bk128 = tune("VQ2_BK128", 32), bk64 = tune("VQ2_BK64", 16), bk32 = tune("VQ2_BK32", 16);
    static const int small_m = tune("VQ2_SMALL_M", 1);
    // few row tiles (the 32x32-resolution layers): halve the tile height so every CU still holds >= 2
    // workgroups and the matrix pipe of a SIMD always has a second wave to switch to
    const long wgs128 = (long)((P.M + 127) / 128) * ((P.Co + 127) / 128) * P.phases;
    static const int fast = tune("VQ2_FAST", 1);
    const long lim = (1L << 29);  // elements: every tensor below 2 GiB so that 32-bit byte offsets suffice
    const bool fast_ok = fast && P.KH * P.KW <= 32 && (long)P.N * P.H * P.W * P.ldx < lim &&
                         (long)P.N * P.Hy * P.Wy * P.ldy < lim && (long)P.N * P.Hy * P.Wy * (P.ldm > P.ldr ? P.ldm : P.ldr) < lim &&
                         (long)P.Co * P.K * P.phases < lim;
    static const int c4k = tune("VQ2_C4", 1);
    if (c4k && fast_ok && !legacy_stamps() && conv_c4_ok(P)) return launch_conv_c4(P, s);
    if (fast_ok && !legacy_stamps() && conv_k64_ok(P)) return launch_conv_k64(P, s);
    if (fast_ok && !legacy_stamps() && wino3_ok(P)) {   // vq2_wino.hip
        ConvGemmParams Q = P;
        Q.stamps = g_stamps;        // (non-null only under vq2_debug_set_stamps + VQ2_CLOCKPROBE=1: the clock-probe instantiation)
        return launch_wino3(Q, s);
    }
    static const int subpix = tune("VQ2_SUBPIX", 1);
    const long big = 0x7F000000L / 4;   // the patch kernel's out-of-range sentinel must stay above every tensor
    // (a launch of <= 256 workgroups with a short depth is better off with the 64-row GEMM tiles: measured)
    const long sp_wgs = (long)P.N * ((P.W + sp::TW - 1) / sp::TW) * ((P.H + sp::TH - 1) / sp::TH) * (P.Co / 64);
    if (fast_ok && !legacy_stamps() && subpix && P.phases == 4 && P.Co % 64 == 0 && P.Ci % 16 == 0 && P.K == 4 * P.Ci &&
        (sp_wgs >= 512 || P.K >= 512) &&
        (long)P.N * P.H * P.W * P.ldx < big && (long)P.N * P.Hy * P.Wy * P.ldy < big &&
        (long)P.N * P.Hy * P.Wy * (P.ldm > P.ldr ? P.ldm : P.ldr) < big)
        return launch_subpixel(P, s);
    if (fast_ok && !legacy_stamps()) {
        static const int t32 = tune("VQ2_T32", 1), tk = tune("VQ2_TSHORTK", 0), t64 = tune("VQ2_T64", 0),
                         tsm = tune("VQ2_TSM", 1);
        if (small_m && wgs128 < 400 && P.Co > 32) {
            if (P.Co > 64) {
                if (tsm == 1) return launch_conv_gemm_fast<2, 2, 1, 2, 32>(P, s);
                return launch_conv_gemm_fast<2, 2, 1, 2, 16>(P, s);
            }
            if (tsm == 1) return launch_conv_gemm_fast<2, 2, 1, 1, 32>(P, s);
            return launch_conv_gemm_fast<2, 2, 1, 1, 16>(P, s);
        }
        if (P.Co > 64) {
            // K <= 64 (the 1x1 convs out of 32/64 channels): one or two chunks, HBM/epilogue-bound ->
            // 64-row tiles double the workgroups in flight (measured +12 % on 32 -> 128)
            if (P.K <= 64) {
                // N = 192 (data gradient of quantize_conv_b, vqvae.py:189): a 192-wide tile covers the row in ONE pass --
                // with 128-wide tiles the second column tile is half empty (a quarter of the MFMAs wasted) and the dy
                // rows are read twice
                static const int t192 = tune("VQ2_T192", 1);
                if (P.Co > 128 && P.Co <= 192 && t192 == 1) return launch_conv_gemm_fast<2, 2, 1, 3, 16>(P, s);
                if (P.Co > 128 && P.Co <= 192 && t192 == 2) return launch_conv_gemm_fast<2, 2, 2, 3, 16>(P, s);
                if (P.Co > 128 && P.Co <= 192 && t192 == 3) return launch_conv_gemm_fast<4, 1, 1, 6, 16>(P, s);
                return launch_conv_gemm_fast<2, 2, 1, 2, 16>(P, s);
            }
            if (tk == 1 && P.K <= 512) return launch_conv_gemm_fast<2, 2, 1, 2, 16>(P, s);   // 64 x 128 for short K
            if (tk == 2 && P.K <= 512) return launch_conv_gemm_fast<2, 2, 1, 2, 32>(P, s);
            static const int t128 = tune("VQ2_T128", 0);
            if (t128 == 1) return launch_conv_gemm_fast<2, 2, 2, 2, 16>(P, s);   // 3 workgroups per CU
            // 513..1024 tiles (every 64x64-resolution layer at batch 32): four workgroups per CU hold ALL tiles at
            // once -- one round instead of two in lock-step (measured +1..3 % per launch, 7.30 -> 7.23 ms/step)
            const bool below_gib = (long)P.N * P.H * P.W * P.ldx * 4 < (1L << 30) && (long)P.Co * P.K * P.phases * 4 < (1L << 30);
            if (P.Ci % 16 == 0 && below_gib && (t128 == 2 || (t128 == 0 && wgs128 > 512 && wgs128 <= 1024)))
                return launch_conv_gemm_fast<2, 2, 2, 2, 16, true>(P, s);
            return launch_conv_gemm_fast<2, 2, 2, 2, 32>(P, s);
        }
        if (P.Co > 32) {
            if (t64 == 1) return launch_conv_gemm_fast<2, 2, 2, 1, 32>(P, s);
            return launch_conv_gemm_fast<2, 2, 2, 1, 16>(P, s);
        }
        if (t32 == 1) return launch_conv_gemm_fast<4, 1, 1, 1, 32>(P, s);
        if (t32 == 2) return launch_conv_gemm_fast<4, 1, 2, 1, 16>(P, s);
        if (t32 == 3) return launch_conv_gemm_fast<4, 1, 2, 1, 32>(P, s);
        return launch_conv_gemm_fast<4, 1, 1, 1, 16>(P, s);
    }
    if (small_m && wgs128 < 400 && P.Co > 32) {
        if (P.Co > 64) return launch_conv_gemm<2, 2, 1, 2, 16>(P, s);      // 64 x 128
        return launch_conv_gemm<2, 2, 1, 1, 16>(P, s);                     // 64 x 64
    }
    if (P.Co > 64) {
        if (bk128 == 16) return launch_conv_gemm<2, 2, 2, 2, 16>(P, s);
        if (legacy_stamps()) { ConvGemmParams Q = P; Q.stamps = g_stamps; return launch_conv_gemm<2, 2, 2, 2, 32, true>(Q, s); }
        return launch_conv_gemm<2, 2, 2, 2, 32>(P, s);                     // 128 x 128
    }
    if (P.Co > 32) {
        static const int tall64 = tune("VQ2_TALL64", 0);
        if (tall64) return launch_conv_gemm<4, 1, 2, 2, 16>(P, s);         // 256 x 64: each wave 64 x 64
        if (bk64 == 32) return launch_conv_gemm<2, 2, 2, 1, 32>(P, s);
        return launch_conv_gemm<2, 2, 2, 1, 16>(P, s);                     // 128 x 64
    }
    if (bk32 == 32) return launch_conv_gemm<4, 1, 1, 1, 32>(P, s);
    return launch_conv_gemm<4, 1, 1, 1, 16>(P, s);                         // 128 x 32
}

// ------------------------------------------------------------------ conv-transpose to <= 4 channels
// The reconstruction layer (vqvae.py:157: ConvTranspose2d(64 -> 3, k4 s2 p1)) has 3 output channels:
// as a 32x32 GEMM tile it would fill 3/32 of the matrix instruction, and it is HBM-bound anyway (AI ~ 20).
// (Round 1's vector-ALU kernel for this layer was removed in round 3: nothing reached it any more -- tensors too
//  large for the kernel below take the generic implicit-GEMM path.)
// Matrix-core form: v_mfma_f32_4x4x1_16B_f32 computes 16 independent 4x4 outer products per
// instruction at the full fp32 rate, and "4 columns" is exactly the padded channel count of the reconstruction.
// Block b of a wave = 4 consecutive input positions (rows of the 4x4), columns = the 4 output channels; a lane
// (4b + i) feeds its own position's activations as A and the weight of channel (lane & 3) as B, one input
// channel per instruction; the four output phases keep one 4-VGPR accumulator each.  The 10x34 halo patch of a
// 16-channel slice and the 16 x 4 weight rows of that slice are staged in LDS; every (neighbour, channel quad)
// is read once and used by up to four phases.
typedef float f32x4m __attribute__((ext_vector_type(4)));
namespace ctm {
constexpr int TH = 16, TW = 32, PW = TW + 2, NP = (TH + 2) * PW;   // 612 patch pixels
constexpr int CS = 8, LD = CS + 4;          // 48-byte LDS rows: conflict-free ds_read_b128 for consecutive pixels
constexpr int X_FLOATS = NP * LD, W_FLOATS = 64 * LD;
constexpr size_t LDS_BYTES = (size_t)2 * (X_FLOATS + W_FLOATS) * sizeof(float);   // 64,896: two workgroups per CU
constexpr int COOB = 0x7F000000;
}  // namespace ctm

// Each wave owns TWO sets of 64 input positions (rows 4*wq + 2u + lane/32): a weight fragment read from LDS is
// used by both, which halves the broadcast reads that otherwise saturate the LDS pipe before the matrix pipe.
__global__ __launch_bounds__(256, 2) void convT_small_mfma_kernel(const float *__restrict__ x, int ldx,
                                                                  const float *__restrict__ wk,   // [16 taps][4 co][Ci]
                                                                  const float *__restrict__ bias, int nbias,
                                                                  float *__restrict__ y, int ldy, int N, int H, int W, int Ci,
                                                                  int relu_in) {
    using namespace ctm;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;                     // [2][X_FLOATS]
    float *Ws = smem + 2 * X_FLOATS;      // [2][W_FLOATS]
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = vid / (tiles_x * tiles_y);
    const int tt = vid - n * tiles_x * tiles_y;
    const int i0 = (tt / tiles_x) * TH, j0 = (tt % tiles_x) * TW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, N * H * W * ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wk), 0, 64 * Ci * 4, RSRC_FLAGS);

    constexpr int QPR = CS / 4;                          // float4 per staged row
    constexpr int X_F4 = NP * QPR;                       // 1224
    constexpr int X_LD = (X_F4 + 255) / 256;             // 5
    int x_off[X_LD];
#pragma unroll
    for (int j = 0; j < X_LD; ++j) {
        const int f = tid + 256 * j;
        x_off[j] = COOB;
        if (f < X_F4) {
            const int pp = f / QPR;
            const int gi = i0 - 1 + pp / PW, gj = j0 - 1 + pp % PW;
            if ((unsigned)gi < (unsigned)H && (unsigned)gj < (unsigned)W) x_off[j] = ((n * H + gi) * W + gj) * ldx * 4 + (f % QPR) * 16;
        }
    }
    const bool w_act = tid < 64 * QPR;
    const int w_off = w_act ? (tid / QPR) * Ci * 4 + (tid % QPR) * 16 : COOB;     // row (tap, co) = tid / QPR
    const int st_off = (tid / QPR) * LD + (tid % QPR) * 4;
    u32x4 rxv[X_LD], rwv;
    auto issue = [&](int s) {
#pragma unroll
        for (int j = 0; j < X_LD; ++j) rxv[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, x_off[j], s * CS * 4, 0);
        rwv = __builtin_amdgcn_raw_buffer_load_b128(rw, w_off, s * CS * 4, 0);
    };
    auto store = [&](int buf) {
        float *xs = Xs + buf * X_FLOATS + st_off;
#pragma unroll
        for (int j = 0; j < X_LD; ++j)
            if ((j + 1) * 256 <= X_F4 || tid + 256 * j < X_F4) {
                const float4 v = as_f4(rxv[j]);
                *reinterpret_cast<float4 *>(xs + j * (256 / QPR) * LD) = relu_in ? relu4(v) : v;
            }
        if (w_act) *reinterpret_cast<float4 *>(Ws + buf * W_FLOATS + st_off) = as_f4(rwv);
    };

    f32x4m acc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[u][p] = f32x4m{0.f, 0.f, 0.f, 0.f};
    const int a_base = ((4 * wq + (lane >> 5)) * PW + (lane & 31)) * LD;   // set u adds 2 patch rows
    const int b_base = (lane & 3) * LD;
    const int NS = Ci / CS;
    issue(0);
    store(0);
    __syncthreads();
    for (int s = 0; s < NS; ++s) {
        const int buf = s & 1;
        if (s + 1 < NS) issue(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float *xs = Xs + buf * X_FLOATS + a_base;
        const float *ws = Ws + buf * W_FLOATS + b_base;
#pragma unroll
        for (int kq = 0; kq < QPR; ++kq) {
            float4 av[2][9];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int nb = 0; nb < 9; ++nb)
                    av[u][nb] = *reinterpret_cast<const float4 *>(xs + ((2 * u + nb / 3) * PW + nb % 3) * LD + kq * 4);
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            // output phase (ph, pw) reads neighbour (ph + a, pw + b) through kernel tap (3-2a-ph, 3-2b-pw)
                            const int tap = (3 - 2 * a - ph) * 4 + (3 - 2 * b - pw);
                            const float4 bv = *reinterpret_cast<const float4 *>(ws + tap * 4 * LD + kq * 4);
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const float4 xv = av[u][(ph + a) * 3 + pw + b];
                                f32x4m c = acc[u][ph * 2 + pw];
                                c = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.x, bv.x, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.y, bv.y, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.z, bv.z, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_4x4x1f32(xv.w, bv.w, c, 0, 0, 0);
                                acc[u][ph * 2 + pw] = c;
                            }
                        }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < NS) store(buf ^ 1);
        __syncthreads();
    }
    // accumulator VGPR r of lane (4b + j) = position 4b + r of the wave's set, channel j
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, N * 4 * H * W * ldy * 4, RSRC_FLAGS);
    const int co = lane & 3;
    const float bvs = (bias && co < nbias) ? bias[co] : 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pos = (lane & ~3) + r;
            const int gi = i0 + 4 * wq + 2 * u + (pos >> 5), gj = j0 + (pos & 31);
            const bool ok = gi < H && gj < W;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int pix = (n * 2 * H + 2 * gi + (p >> 1)) * (2 * W) + 2 * gj + (p & 1);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[u][p][r] + bvs), ry, ok ? (pix * ldy + co) * 4 : COOB, 0, 0);
            }
        }
}

static bool use_convT_small(const vq2_conv_desc *d) {
    return d->transposed && d->Co == 4 && d->Cor >= 1 && d->Cor <= 3 && d->Ci % 16 == 0 && d->N <= 65535;
}

static int check_desc(const vq2_conv_desc *d) {
    VQ2_REQUIRE(d != nullptr, "conv desc is null");
    VQ2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ci > 0 && d->Co > 0, "conv desc: non-positive dims");
    VQ2_REQUIRE(d->Ci % 4 == 0 && d->Co % 4 == 0, "conv desc: Ci=%d, Co=%d must be multiples of 4", d->Ci, d->Co);
    VQ2_REQUIRE(d->ldx >= d->Ci && d->ldx % 4 == 0, "conv desc: ldx=%d must be >= Ci and a multiple of 4", d->ldx);
    VQ2_REQUIRE(d->ldy >= d->Co && d->ldy % 4 == 0, "conv desc: ldy=%d must be >= Co and a multiple of 4", d->ldy);
    VQ2_REQUIRE(d->Cir >= 0 && d->Cir <= d->Ci && d->Cor >= 0 && d->Cor <= d->Co, "conv desc: Cir/Cor out of range");
    if (d->transposed) {
        VQ2_REQUIRE(d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1,
                    "conv-transpose supports k4 s2 p1 only (vqvae.py:150-160,191-193)");
    } else {
        VQ2_REQUIRE(d->KH >= 1 && d->KH <= 7 && d->KW == d->KH, "conv: square kernel 1..7 required");
        VQ2_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride 1 or 2");
        VQ2_REQUIRE(d->pad >= 0 && d->pad < d->KH, "conv: 0 <= pad < KH");
        VQ2_REQUIRE(d->H + 2 * d->pad >= d->KH && d->W + 2 * d->pad >= d->KW, "conv: kernel larger than padded input");
        if (d->stride == 2) VQ2_REQUIRE(d->KH == 4 && d->pad == 1 && d->H % 2 == 0 && d->W % 2 == 0,
                                        "stride-2 conv supports k4 p1 on even sizes (vqvae.py:105,107,114)");
    }
    int Ho, Wo;
    if (d->transposed) { Ho = 2 * d->H; Wo = 2 * d->W; }
    else { Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1; Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1; }
    const int64_t lim = (int64_t)1 << 31;
    VQ2_REQUIRE((int64_t)d->N * d->H * d->W * d->ldx < lim && (int64_t)d->N * Ho * Wo * d->ldy < lim,
                "conv: tensor exceeds 2^31 elements (int32 indexing)");
    return VQ2_OK;
}

static void out_dims(const vq2_conv_desc *d, int &Ho, int &Wo) {
    if (d->transposed) { Ho = 2 * d->H; Wo = 2 * d->W; }
    else { Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1; Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1; }
}

}  // namespace vq2

using namespace vq2;

// mode 3 = pack_convT_small layout
static int pack_job_of(const vq2_conv_desc *d, int which, vq2_pack_job *j) {
    const int cir = d->Cir ? d->Cir : d->Ci, cor = d->Cor ? d->Cor : d->Co;
    j->KH = d->KH; j->KW = d->KW;
    if (which == VQ2_PACK_FWD && use_convT_small(d)) {
        j->mode = 3; j->Or = d->Ci; j->Ir = cor; j->Op = d->Ci; j->Ip = 4;
    } else if (!d->transposed) {
        j->Or = cor; j->Ir = cir; j->Op = d->Co; j->Ip = d->Ci;      // w is [Co][Ci][KH][KW]
        j->mode = (which == VQ2_PACK_FWD) ? 0 : ((d->stride == 1) ? 1 : 2);
    } else {
        j->Or = cir; j->Ir = cor; j->Op = d->Ci; j->Ip = d->Co;      // w is [Ci][Co][4][4]
        j->mode = (which == VQ2_PACK_FWD) ? 2 : 0;
    }
    j->numel = (int64_t)j->Op * j->Ip * j->KH * j->KW;
    return VQ2_OK;
}

__device__ __forceinline__ float pack_elem(const float *__restrict__ w, int t, int Or, int Ir, int Op, int Ip, int KH,
                                           int KW, int mode) {
    const int taps = KH * KW;
    int o, i, kh, kw;
    if (mode == 0) {
        i = t % Ip; const int tap = (t / Ip) % taps; o = t / (Ip * taps);
        kh = tap / KW; kw = tap % KW;
    } else if (mode == 1) {
        o = t % Op; const int tapf = (t / Op) % taps; i = t / (Op * taps);
        kh = KH - 1 - tapf / KW; kw = KW - 1 - tapf % KW;
    } else if (mode == 2) {
        o = t % Op; const int ab = (t / Op) % 4; i = (t / (Op * 4)) % Ip; const int phase = t / (Op * 4 * Ip);
        kh = 3 - 2 * (ab >> 1) - (phase >> 1); kw = 3 - 2 * (ab & 1) - (phase & 1);
    } else {  // [tap][4][Ci]: o = ci (dim 0 of w), i = co (dim 1)
        o = t % Op; i = (t / Op) % 4; const int tap = t / (4 * Op);
        kh = tap / 4; kw = tap % 4;
    }
    return (o < Or && i < Ir) ? w[((o * Ir + i) * KH + kh) * KW + kw] : 0.f;
}

// all weight panels of a model in ONE launch: job j owns [offset_j, offset_j + numel_j) of the index space
__global__ __launch_bounds__(256) void pack_batched_kernel(const vq2_pack_job *__restrict__ jobs, int njobs, int64_t total) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {  // last job with offset <= g
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].offset <= g) lo = mid; else hi = mid - 1;
        }
        const vq2_pack_job j = jobs[lo];
        const int t = (int)(g - j.offset);
        j.packed[t] = pack_elem(j.w, t, j.Or, j.Ir, j.Op, j.Ip, j.KH, j.KW, j.mode);
    }
}

extern "C" int vq2_pack_job_init(const vq2_conv_desc *d, int which, const float *w, float *packed, vq2_pack_job *job) {
    if (int e = check_desc(d)) return e;
    VQ2_REQUIRE(w && packed && job, "pack_job_init: null pointer");
    VQ2_REQUIRE(which == VQ2_PACK_FWD || which == VQ2_PACK_DGRAD, "pack_job_init: bad `which`");
    job->w = w; job->packed = packed; job->offset = 0;
    return pack_job_of(d, which, job);
}

extern "C" int vq2_pack_weights_batched(const vq2_pack_job *jobs_dev, int32_t njobs, int64_t total, vq2_stream_t stream) {
    VQ2_REQUIRE(jobs_dev && njobs > 0 && total > 0, "pack_weights_batched: bad arguments");
    const int64_t b = (total + 255) / 256;
    hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)(b > 2048 ? 2048 : b)), dim3(256), 0, to_stream(stream), jobs_dev,
                       njobs, total);
    return check_launch("pack_batched_kernel");
}

__global__ void pack_single_kernel(const float *__restrict__ w, float *__restrict__ p, int Or, int Ir, int Op, int Ip,
                                   int KH, int KW, int mode, int total) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x)
        p[t] = pack_elem(w, t, Or, Ir, Op, Ip, KH, KW, mode);
}

extern "C" int vq2_pack_weight(const vq2_conv_desc *d, int which, const float *w, float *packed, vq2_stream_t stream) {
    vq2_pack_job j;
    if (int e = vq2_pack_job_init(d, which, w, packed, &j)) return e;
    const int total = (int)j.numel;
    const int blocks = (total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024;
    hipLaunchKernelGGL(pack_single_kernel, dim3(blocks), dim3(256), 0, to_stream(stream), w, packed, j.Or, j.Ir, j.Op, j.Ip,
                       j.KH, j.KW, j.mode, total);
    return check_launch("pack_single_kernel");
}

extern "C" int vq2_conv_fwd(const vq2_conv_desc *d, int flags, const float *x, const float *wp, const float *bias,
                            const float *residual, int32_t ldres, float *y, vq2_stream_t stream) {
    if (int e = check_desc(d)) return e;
    VQ2_REQUIRE(x && wp && y, "conv_fwd: null pointer");
    VQ2_REQUIRE(aligned16(x) && aligned16(wp) && aligned16(y), "conv_fwd: pointers must be 16-byte aligned");
    VQ2_REQUIRE(!residual || (ldres >= d->Co), "conv_fwd: ldres < Co");
    if (use_convT_small(d) && !residual && !(flags & VQ2_RELU_OUT)) {
        // (the weight panel of this layer is packed for THIS kernel: there is no other path to fall through to)
        if (!((double)d->N * 4 * d->H * d->W * d->ldy * 4 < (double)ctm::COOB && (double)d->N * d->H * d->W * d->ldx * 4 < (double)ctm::COOB))
            return set_error(VQ2_ERR_UNSUPPORTED, "conv_fwd: a conv-transpose to <= 4 channels is limited to tensors below %d bytes "
                             "(split the batch)", ctm::COOB);
        hipStream_t s = to_stream(stream);
        const int cor = d->Cor ? d->Cor : d->Co;
        const char *name = "convT_small";
        if (prof_enabled()) name = prof_label("convT_small|N=%d,H=%d,W=%d,Ci=%d", d->N, d->H, d->W, d->Ci);
        ProfScope prof(name, 2.0 * d->N * d->H * d->W * 16.0 * d->Ci * cor,
                       4.0 * ((double)d->N * d->H * d->W * d->Ci + 4.0 * d->N * d->H * d->W * cor), s);
        const int g = ((d->W + ctm::TW - 1) / ctm::TW) * ((d->H + ctm::TH - 1) / ctm::TH) * d->N;
        allow_big_lds(convT_small_mfma_kernel, ctm::LDS_BYTES);
        hipLaunchKernelGGL(convT_small_mfma_kernel, dim3(g), dim3(256), ctm::LDS_BYTES, s, x, d->ldx, wp, bias, cor, y,
                           d->ldy, d->N, d->H, d->W, d->Ci, (flags & VQ2_RELU_IN) != 0);
        return check_launch("convT_small_mfma_kernel");
    }
    ConvGemmParams P{};
    P.x = x; P.w = wp; P.bias = bias; P.mask = nullptr; P.res = residual; P.y = y;
    P.N = d->N; P.H = d->H; P.W = d->W; P.Ci = d->Ci; P.ldx = d->ldx;
    P.Co = d->Co; P.ldy = d->ldy; P.ldr = ldres; P.ldm = 0;
    P.relu_in = (flags & VQ2_RELU_IN) != 0; P.relu_out = (flags & VQ2_RELU_OUT) != 0;
    P.nbias = d->Cor ? d->Cor : d->Co;
    P.ci_real = d->Cir ? d->Cir : d->Ci;
    out_dims(d, P.Hy, P.Wy);
    if (!d->transposed) {
        P.KH = d->KH; P.KW = d->KW; P.stride = d->stride; P.pad_h = P.pad_w = d->pad;
        P.Ho = P.Hy; P.Wo = P.Wy; P.phases = 1;
    } else {
        P.KH = 2; P.KW = 2; P.stride = 1; P.pad_h = P.pad_w = 1;
        P.Ho = d->H; P.Wo = d->W; P.phases = 4;
    }
    P.K = P.KH * P.KW * P.Ci; P.M = P.N * P.Ho * P.Wo;
    {   // algorithmic work: real channels, every tap once; x read once, y written once (+ residual read)
        const double cir = d->Cir ? d->Cir : d->Ci, cor = d->Cor ? d->Cor : d->Co;
        const double pix_in = (double)d->N * d->H * d->W, pix_out = (double)d->N * P.Hy * P.Wy;
        const double macs = d->transposed ? pix_in * 16.0 * cir * cor : pix_out * d->KH * d->KW * cir * cor;
        P.flops = 2.0 * macs;
        P.bytes = 4.0 * (pix_in * cir + pix_out * cor * (residual ? 2.0 : 1.0) + cir * cor * d->KH * d->KW);
    }
    return run_conv_gemm(P, to_stream(stream));
}

extern "C" int vq2_conv_dgrad(const vq2_conv_desc *d, const float *dy, const float *wp, const float *mask,
                              int32_t ldmask, const float *residual, int32_t ldres, float *dx, int32_t lddx,
                              vq2_stream_t stream) {
    return vq2_conv_dgrad_ex(d, 0, dy, wp, mask, ldmask, residual, ldres, dx, lddx, stream);
}

extern "C" int vq2_conv_dgrad_ex(const vq2_conv_desc *d, int flags, const float *dy, const float *wp, const float *mask,
                                 int32_t ldmask, const float *residual, int32_t ldres, float *dx, int32_t lddx,
                                 vq2_stream_t stream) {
    if (int e = check_desc(d)) return e;
    VQ2_REQUIRE((flags & ~VQ2_MASK_AFTER_RESIDUAL) == 0, "conv_dgrad: unknown flag");
    VQ2_REQUIRE(dy && wp && dx, "conv_dgrad: null pointer");
    VQ2_REQUIRE(aligned16(dy) && aligned16(wp) && aligned16(dx), "conv_dgrad: pointers must be 16-byte aligned");
    VQ2_REQUIRE(lddx >= d->Ci && lddx % 4 == 0, "conv_dgrad: lddx=%d must be >= Ci and a multiple of 4", lddx);
    VQ2_REQUIRE((!mask || ldmask >= d->Ci) && (!residual || ldres >= d->Ci), "conv_dgrad: ldmask/ldres < Ci");
    int Hy, Wy;
    out_dims(d, Hy, Wy);
    ConvGemmParams P{};
    // the gradient GEMM reads dy [N,Hy,Wy,Co] and produces dx [N,H,W,Ci]
    P.x = dy; P.w = wp; P.bias = nullptr; P.mask = mask; P.res = residual; P.y = dx;
    P.N = d->N; P.H = Hy; P.W = Wy; P.Ci = d->Co; P.ldx = d->ldy;
    P.Co = d->Ci; P.ldy = lddx; P.ldm = ldmask; P.ldr = ldres;
    P.relu_in = 0; P.relu_out = 0;
    P.ci_real = d->Cor ? d->Cor : d->Co;   // the gradient GEMM's input channels are the forward op's output channels
    P.mask_after = (flags & VQ2_MASK_AFTER_RESIDUAL) != 0;
    P.Hy = d->H; P.Wy = d->W;
    if (!d->transposed && d->stride == 1) {
        // full correlation with the flipped kernel: pad' = KH-1-pad
        P.KH = d->KH; P.KW = d->KW; P.stride = 1; P.pad_h = P.pad_w = d->KH - 1 - d->pad;
        P.Ho = d->H; P.Wo = d->W; P.phases = 1;
    } else if (!d->transposed) {
        // stride-2 k4 p1: dx = conv_transpose(dy): sub-pixel phases over the dy grid
        P.KH = 2; P.KW = 2; P.stride = 1; P.pad_h = P.pad_w = 1;
        P.Ho = Hy; P.Wo = Wy; P.phases = 4;
    } else {
        // conv-transpose: dx = strided conv of dy (k4 s2 p1)
        P.KH = 4; P.KW = 4; P.stride = 2; P.pad_h = P.pad_w = 1;
        P.Ho = d->H; P.Wo = d->W; P.phases = 1;
    }
    P.K = P.KH * P.KW * P.Ci; P.M = P.N * P.Ho * P.Wo;
    {
        const double cir = d->Cir ? d->Cir : d->Ci, cor = d->Cor ? d->Cor : d->Co;
        const double pix_in = (double)d->N * d->H * d->W, pix_out = (double)d->N * Hy * Wy;
        const double macs = d->transposed ? pix_in * 16.0 * cir * cor : pix_out * d->KH * d->KW * cir * cor;
        P.flops = 2.0 * macs;
        P.bytes = 4.0 * (pix_out * cor + pix_in * cir * (1.0 + (mask ? 1.0 : 0.0) + (residual ? 1.0 : 0.0)) +
                         cir * cor * d->KH * d->KW);
    }
    return run_conv_gemm(P, to_stream(stream));
}

// diagnostic only: when set, the 128x128x32 conv tile runs its STAMP build and workgroup 64 writes, per
// wave, the summed s_memtime cycles of {load issue, MFMA phase, LDS store, barrier} to buf[16]
extern "C" int vq2_debug_set_stamps(unsigned long long *buf) {
    g_stamps = buf;
    return VQ2_OK;
}
