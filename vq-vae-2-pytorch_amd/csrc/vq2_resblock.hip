// Fused ResBlock kernels for gfx950 (MI355X):  vqvae.py:81-96
//
//     r = relu(conv3x3(relu(x)) + b1)          [N,H,W,32]   (saved for the backward pass)
//     y = [relu]( conv1x1(r) + b2 + x )        [N,H,W,128]
//
// as ONE launch.  The unfused path (two conv_gemm launches) re-reads every input pixel 9 times
// through L2 -- with only 32 output channels per tile that staging, not the matrix pipe, is what
// bounds it.  Here a workgroup owns an 8x16 pixel tile: the 10x18 halo patch of the input is staged
// into LDS ONCE per 16-channel slice and all nine taps are fed from it by shifting the fragment
// base address (1.4x the tensor instead of 9x, 72 MFMAs between barriers instead of 16); the 32
// mid channels never leave the CU before the 1x1 GEMM consumes them from LDS.
//
//   stage 1: acc1[128 px x 32]  = sum over 8 channel slices, 9 taps, 16 ci      (v_mfma_f32_32x32x2_f32)
//   stage 2: acc2[128 px x 128] = r_tile[128 x 32] . W2[32 x 128]  + b2 + x     (same instruction)
//
// Every tile is computed the same way whatever the batch size: results do not depend on N (an 8-wave
// variant that split the depth between two wave groups was measured equal at 32x32 and dropped).
#include "vq2_common.h"
#include "vq2_rbwino.h"
#include <stdlib.h>

namespace vq2 {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef VQ2_RB_WRITE_AFTER
#define VQ2_RB_WRITE_AFTER 1   // staging order of the slice pipelines (A/B builds: scripts/build_variant.sh)
#endif

namespace rb {
constexpr bool WRITE_AFTER = VQ2_RB_WRITE_AFTER != 0;
#ifndef VQ2_RB_INTERLEAVE
#define VQ2_RB_INTERLEAVE 1
#endif
constexpr bool INTERLEAVE = VQ2_RB_INTERLEAVE != 0;   // staging instructions woven into the MFMA stream (forward kernel)
constexpr unsigned RSRC_FLAGS = 0x00020000;
constexpr int OOB = 0x7F000000;   // >= num_records of every descriptor (tensors are checked to be smaller), and
                                  // OOB + any in-tensor slice offset does not wrap
constexpr int TH = 8, TW = 16;            // output pixels of one workgroup
constexpr int PW = TW + 2, PH = TH + 2;   // halo patch
constexpr int NPATCH = PW * PH;           // 180
constexpr int CS = 16;                    // input channels per staged slice
constexpr int LDK = CS + 4;               // LDS row pitch (floats): conflict-free ds_read_b128 for 16 consecutive rows
constexpr int CM = 32;                    // mid channels (n_res_channel)
constexpr int CC = 128;                   // block channels
constexpr int A_F4 = NPATCH * CS / 4;     // 720 float4 per activation slice
constexpr int B_F4 = 9 * CM * CS / 4;     // 1152 float4 per weight slice
// LDS buffers are padded to whole staging passes (256 threads x float4): every thread then stores every float4 it
// loaded, the out-of-patch ones (offset OOB: the range check returned zeros) into rows nobody reads.  With the stores
// predicated instead, hipcc sinks the LOAD into the predicated block and waits vmcnt(0) right behind it.
constexpr int A_FLOATS = ((A_F4 + 255) / 256) * 64 * LDK;    // 192 rows: 3840
constexpr int B_FLOATS = ((B_F4 + 255) / 256) * 64 * LDK;    // 320 rows: 6400
constexpr int LDR = CM + 4;               // pitch of the r tile and the W2 panel in LDS
constexpr size_t LDS_BYTES = (size_t)2 * (A_FLOATS + B_FLOATS) * sizeof(float);   // 81,920: exactly two workgroups per CU
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert(128 * LDR <= 2 * (A_FLOATS + B_FLOATS), "stage-2 aliases fit in the stage-1 buffers");
}  // namespace rb

struct ResFwdParams {
    const float *x;    // [N,H,W,ldx]
    const float *w1;   // VQ2_PACK_FWD panel of the 3x3 weight: [32][9*128], (kh,kw,ci) with ci fastest
    const float *b1;   // [32]
    const float *w2;   // VQ2_PACK_FWD panel of the 1x1 weight: [128][32]
    const float *b2;   // [128]
    float *r;          // [N,H,W,ldr]
    float *y;          // [N,H,W,ldy]
    int N, H, W, ldx, ldr, ldy;
    int tiles_x, tiles_y;
    int relu_out;
    int dephase, first_round;   // see dephase_start
    unsigned long long *stamps; // diagnostic (vq2_debug_set_rb_stamps): s_memtime at the phase boundaries of 2 workgroups
};

__device__ __forceinline__ float4 u4_as_f4(u32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// De-phased start.  Two workgroups share a CU (one wave of each per SIMD).  Launched together they run in lock-step:
// both fetch their first slices together, share the matrix pipe through their MFMA phases and burst their epilogue
// stores together, so memory phases never sit beside matrix phases.  The workgroup that arrives in the SECOND wave slot
// of its SIMD (HW_REG_HW_ID[3:0] != 0) during the launch's first resident round waits `cycles` before it starts; from
// then on the two stay out of phase (a freed slot is refilled when ITS workgroup ends).  While the second one waits,
// the first has the matrix pipe to itself, so the wait is not lost time.  Speed only: results cannot depend on it.
__device__ __forceinline__ void dephase_start(int cycles, int first_round_blocks) {
    if (cycles <= 0 || (int)blockIdx.x >= first_round_blocks) return;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((hw & 15u) == 0u) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)cycles) __builtin_amdgcn_s_sleep(8);
}

__global__ __launch_bounds__(256, 2) void resblock_fwd_kernel(const ResFwdParams P) {
    using namespace rb;
    constexpr int NT = 256;
    constexpr int A_LD = (A_F4 + NT - 1) / NT;
    constexpr int B_LD = (B_F4 + NT - 1) / NT;
    constexpr int NS = CC / CS;          // channel slices
    constexpr int NJ = CC / 32;          // stage-2 column blocks per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                    // [2][A_FLOATS]
    float *Bs = smem + 2 * A_FLOATS;     // [2][B_FLOATS]
    float *Rs = smem;                    // stage 2 (aliases): r tile  [128][LDR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wq = wave;
    const int l31 = lane & 31, fk = 4 * (lane >> 5);
    const int stamp_slot = (P.stamps && (blockIdx.x == 8 || blockIdx.x == 520)) ? (blockIdx.x == 8 ? 0 : 1) : -1;
    auto stamp = [&](int i) {
        if (stamp_slot >= 0 && lane == 0) {
            P.stamps[(stamp_slot * 4 + wq) * 8 + i] = __builtin_amdgcn_s_memtime();
            if (i == 0 || i == 5) P.stamps[(stamp_slot * 4 + wq) * 8 + (i == 0 ? 6 : 7)] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    dephase_start(P.dephase, P.first_round);
    const int tiles = P.tiles_x * P.tiles_y;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = vid / tiles;
    const int t = vid - n * tiles;
    const int tyi = t / P.tiles_x;
    const int y0 = tyi * TH, x0 = (t - tyi * P.tiles_x) * TW;

    const int npix = P.N * P.H * P.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, npix * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w1), 0, CM * 9 * CC * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w2), 0, CC * CM * 4, RSRC_FLAGS);

    // ---- staging coordinates: float4 number f = tid + NT*j of a slice goes to LDS row f>>2, column (f&3)*4
    int a_off[A_LD], b_off[B_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int f = tid + NT * j;
        a_off[j] = OOB;
        if (f < A_F4) {
            const int pp = f >> 2;
            const int pr = pp / PW, pc = pp - pr * PW;
            const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
            if ((unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W)
                a_off[j] = ((n * P.H + gy) * P.W + gx) * P.ldx * 4 + (f & 3) * 16;
        }
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int f = tid + NT * j;
        b_off[j] = OOB;
        if (f < B_F4) {
            const int row = f >> 2;            // tap * 32 + co
            b_off[j] = ((row & 31) * 9 * CC + (row >> 5) * CC) * 4 + (f & 3) * 16;
        }
    }
    const int st_off = (tid >> 2) * LDK + (tid & 3) * 4;   // + (NT/4)*LDK per j
    struct Slice { u32x4 a[A_LD], b[B_LD]; };
    auto issue_loads = [&](int s, Slice &r) {
        const int soff = s * CS * 4;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) r.a[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, a_off[j], soff, 0);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) r.b[j] = __builtin_amdgcn_raw_buffer_load_b128(rw1, b_off[j], soff, 0);
    };
    auto store_slice = [&](int buf, const Slice &r) {
        float *a = As + buf * A_FLOATS + st_off;
        float *b = Bs + buf * B_FLOATS + st_off;
#pragma unroll
        for (int j = 0; j < A_LD; ++j)
            *reinterpret_cast<float4 *>(a + j * (NT / 4) * LDK) = relu4(u4_as_f4(r.a[j]));   // first ReLU of the block
#pragma unroll
        for (int j = 0; j < B_LD; ++j)
            *reinterpret_cast<float4 *>(b + j * (NT / 4) * LDK) = u4_as_f4(r.b[j]);
    };

    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;

    // fragment bases: GEMM row i of this wave = tile pixel (2*wq + i/16, tx(i)).  The second tile row is rotated by
    // 14 columns: its patch rows then sit at 18 + (i+14)%16 = i (mod 16) from the first row's, i.e. the 32 lanes
    // hit the 16-byte LDS slots exactly like 32 consecutive rows do -- conflict-free ds_read_b128 for every tap.
    const int a_frag = ((2 * wq + (l31 >> 4)) * PW + (l31 < 16 ? l31 : ((l31 + 14) & 15))) * LDK + fk;
    const int b_frag = l31 * LDK + fk;
    // Staging INSIDE the MFMA stream (round 3).  Fine stamps of a slice iteration (VQ2_RB_FINE build): 4,651 cycles of
    // MFMAs (72 x 64: the chain itself is perfect), but 909 cycles to ISSUE the eight buffer loads of the slice after next
    // and 525 for the eight LDS stores of the next one, all in front of the first MFMA -- a quarter of the iteration with an
    // idle matrix pipe when the wave is alone on its SIMD.  A vector-memory or LDS-store instruction issued right behind an
    // MFMA costs nothing (the MFMA executes for 64 cycles): MODE 1 stores slice s+1 from the registers one LDS store per
    // MFMA group (groups 0..7) and requests slice s+2 one load per group (groups 8..15); MODE 2 (last slice) spreads the
    // stage-2 prefetch the same way.
    auto stage_one_store = [&](int buf, const Slice &r, int j) {
        if (j < A_LD) *reinterpret_cast<float4 *>(As + buf * A_FLOATS + st_off + j * (NT / 4) * LDK) = relu4(u4_as_f4(r.a[j]));
        else *reinterpret_cast<float4 *>(Bs + buf * B_FLOATS + st_off + (j - A_LD) * (NT / 4) * LDK) = u4_as_f4(r.b[j - A_LD]);
    };
    auto stage_one_load = [&](int s, Slice &r, int j) {
        if (j < A_LD) r.a[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, a_off[j], s * CS * 4, 0);
        else r.b[j - A_LD] = __builtin_amdgcn_raw_buffer_load_b128(rw1, b_off[j - A_LD], s * CS * 4, 0);
    };
    auto compute = [&](int buf) {
        const float *a0 = As + buf * A_FLOATS + a_frag;
        const float *b0 = Bs + buf * B_FLOATS + b_frag;
        // fragment pairs double-buffered by hand: pair p+1 is requested before the four MFMAs of pair p (left to itself
        // the compiler reuses one register set and waits for every pair in front of its MFMAs)
        constexpr int NP = 9 * (CS / 8);
        float4 fa[2], fb[2];
        fa[0] = *reinterpret_cast<const float4 *>(a0);
        fb[0] = *reinterpret_cast<const float4 *>(b0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int cur = p & 1, nxt = cur ^ 1;
            if (p + 1 < NP) {
                const int tap = (p + 1) / (CS / 8), k8 = (p + 1) % (CS / 8);
                fa[nxt] = *reinterpret_cast<const float4 *>(a0 + ((tap / 3) * PW + (tap % 3)) * LDK + 8 * k8);
                fb[nxt] = *reinterpret_cast<const float4 *>(b0 + tap * 32 * LDK + 8 * k8);
            }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur].z, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur].w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // next pair's two LDS reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then this pair's MFMAs
        }
    };

    // pixel of GEMM row l31 of this wave, and per accumulator register (row = rowq + (r&3) + 8*(r>>2)) the byte
    // offsets of that pixel in x / y at this lane's channel: column block j adds the immediate j*128
    int pix_lane;
    {
        const int gy = y0 + 2 * wq + (l31 >> 4), gx = x0 + (l31 < 16 ? l31 : ((l31 + 14) & 15));
        pix_lane = (gy < P.H && gx < P.W) ? (n * P.H + gy) * P.W + gx : -1;
    }
    const int rowq = 4 * (lane >> 5);
    int xoff[16], yoff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pix = __shfl(pix_lane, rowq + (r & 3) + 8 * (r >> 2), 64);
        xoff[r] = pix >= 0 ? pix * (P.ldx * 4) + l31 * 4 : OOB;
        yoff[r] = pix >= 0 ? pix * (P.ldy * 4) + l31 * 4 : OOB;
    }
    f32x16 acc2[NJ];     // starts as the skip path x (vqvae.py:94); the 1x1 GEMM accumulates on top
    float4 w2f[NJ][CM / 8];   // this lane's B fragments of the 1x1 weight, straight from L2 (16 KB panel, no LDS trip)
    float b1v = 0.f, b2v[NJ];   // this lane's bias values (requested with the stage-2 prefetch, not when they are needed)
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.b1), 0, CM * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rb2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.b2), 0, CC * 4, RSRC_FLAGS);

    // Pipeline (one barrier per slice): the registers hold slice s+1 when iteration s begins (requested a whole MFMA
    // phase earlier); it goes to the LDS buffer that every wave finished reading before the last barrier, slice s+2 is
    // requested at once, then the 72 MFMAs of slice s -- the barrier follows them directly (no load wait and no LDS
    // stores between the last MFMA of one slice and the first of the next).  Slices 0 and 1 are requested back to back
    // into two register sets, so the peeled first iteration does not wait for a request it has just made.
    // compute(buf) with the staging of slice s+1 (store) and s+2 (load) woven in; do_store / do_load are uniform
    auto stage2_load = [&](int i) {      // one of the 16 + 64 loads of stage2_prefetch
        constexpr int NW2 = NJ * (CM / 8);
        if (i < NW2) {
            const int j = i / (CM / 8), k8 = i % (CM / 8);
            w2f[j][k8] = u4_as_f4(__builtin_amdgcn_raw_buffer_load_b128(rw2, ((j * 32 + l31) * CM + fk + 8 * k8) * 4, 0, 0));
        } else if (i < NW2 + NJ * 16) {
            const int j = (i - NW2) / 16, r = (i - NW2) % 16;
            acc2[j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xoff[r] + j * 128, 0, 0));
        } else if (i == NW2 + NJ * 16) {
            b1v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb1, l31 * 4, 0, 0));
        } else if (i <= NW2 + NJ * 16 + NJ) {
            const int j = i - (NW2 + NJ * 16 + 1);
            b2v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb2, (j * 32 + l31) * 4, 0, 0));
        }
    };
    auto compute_staged = [&](int buf, int s, Slice &r, bool do_store, bool do_load, bool do_stage2) {
        const float *a0 = As + buf * A_FLOATS + a_frag;
        const float *b0 = Bs + buf * B_FLOATS + b_frag;
        constexpr int NP = 9 * (CS / 8);
        constexpr int NST = A_LD + B_LD;
        static_assert(2 * NST <= NP, "stores then loads fit in the MFMA groups of one slice");
        float4 fa[2], fb[2];
        fa[0] = *reinterpret_cast<const float4 *>(a0);
        fb[0] = *reinterpret_cast<const float4 *>(b0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int cur = p & 1, nxt = cur ^ 1;
            if (p + 1 < NP) {
                const int tap = (p + 1) / (CS / 8), k8 = (p + 1) % (CS / 8);
                fa[nxt] = *reinterpret_cast<const float4 *>(a0 + ((tap / 3) * PW + (tap % 3)) * LDK + 8 * k8);
                fb[nxt] = *reinterpret_cast<const float4 *>(b0 + tap * 32 * LDK + 8 * k8);
            }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur].z, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur].w, acc1, 0, 0, 0);
            if (p < NST) { if (do_store) stage_one_store(buf ^ 1, r, p); }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // next pair's two LDS reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then this pair's MFMAs
            if (p < NST) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // ... one LDS store behind them
            if (p >= NST && p < 2 * NST && do_load) {
                // (a group barrier does not hold a buffer load in place -- hipcc sinks all eight to the end of the slice,
                //  where their issue is exposed again; a full scheduling fence on either side does)
                __builtin_amdgcn_sched_barrier(0);
                stage_one_load(s + 2, r, p - NST);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (do_stage2) {     // last slice: the 80 loads stage 2 needs, five per MFMA group
                constexpr int PER = (NJ * (CM / 8) + NJ * 16 + 1 + NJ + NP - 1) / NP;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < PER; ++q) stage2_load(p * PER + q);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    Slice R0, R1;
    issue_loads(0, R0);
    issue_loads(1, R1);
    store_slice(0, R0);
    __syncthreads();
    stamp(1);
    auto stage2_prefetch = [&]() {   // what stage 2 needs, fetched behind the last slice's 72 MFMAs
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int k8 = 0; k8 < CM / 8; ++k8)
                w2f[j][k8] = u4_as_f4(__builtin_amdgcn_raw_buffer_load_b128(
                    rw2, ((j * 32 + l31) * CM + fk + 8 * k8) * 4, 0, 0));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc2[j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xoff[r] + j * 128, 0, 0));
        b1v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb1, l31 * 4, 0, 0));
#pragma unroll
        for (int j = 0; j < NJ; ++j) b2v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb2, (j * 32 + l31) * 4, 0, 0));
    };
    if (WRITE_AFTER && INTERLEAVE) {
        // slice 0: the registers R1 hold slice 1 (requested up front), R0 is free for slice 2
        store_slice(1, R1);
        __builtin_amdgcn_sched_barrier(0);
        compute_staged(0, 0, R0, false, true, false);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        for (int s = 1; s < NS; ++s) {
            const int buf = s & 1;
            __builtin_amdgcn_sched_barrier(0);
            compute_staged(buf, s, R0, s + 1 < NS, s + 2 < NS, s + 1 == NS);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    } else if (WRITE_AFTER) {
        store_slice(1, R1);
        issue_loads(2, R0);
        __builtin_amdgcn_sched_barrier(0);   // the fetches must be in flight BEFORE the 72 MFMAs, not sunk behind them
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
#ifdef VQ2_RB_FINE   // diagnostic variant build (scripts/build_variant.sh fine -DVQ2_RB_FINE): where a slice iteration goes
        unsigned long long f_store = 0, f_issue = 0, f_mfma = 0, f_bar = 0;
#define VQ2_FT(acc, stmt) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); \
                            stmt; __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
                            acc += __builtin_amdgcn_s_memtime() - t0_; }
#else
#define VQ2_FT(acc, stmt) { stmt; }
#endif
        for (int s = 1; s < NS; ++s) {
            const int buf = s & 1;
            VQ2_FT(f_store, if (s + 1 < NS) store_slice(buf ^ 1, R0))
            VQ2_FT(f_issue, if (s + 2 < NS) issue_loads(s + 2, R0); else if (s + 1 == NS) stage2_prefetch())
            __builtin_amdgcn_sched_barrier(0);
            VQ2_FT(f_mfma, compute(buf))
            __builtin_amdgcn_sched_barrier(0);
            VQ2_FT(f_bar, __syncthreads())
        }
#undef VQ2_FT
#ifdef VQ2_RB_FINE
        if (stamp_slot >= 0 && lane == 0) {
            unsigned long long *fs = P.stamps + 64 + (stamp_slot * 4 + wq) * 4;   // behind the 64 coarse stamp words
            fs[0] = f_store; fs[1] = f_issue; fs[2] = f_mfma; fs[3] = f_bar;
        }
#endif
    } else {   // write-before-barrier order: slice s+1 is requested in front of the MFMAs of slice s and stored behind them
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        store_slice(1, R1);
        __syncthreads();
        for (int s = 1; s < NS; ++s) {
            const int buf = s & 1;
            if (s + 1 < NS) issue_loads(s + 1, R0); else stage2_prefetch();
            __builtin_amdgcn_sched_barrier(0);
            compute(buf);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < NS) store_slice(buf ^ 1, R0);
            __syncthreads();
        }
    }
    stamp(2);
    // every wave is past its last fragment read: the staging buffers may be overwritten
    // r = relu(acc1 + b1): to LDS for stage 2 now; its 16 stores to HBM (the backward pass needs r) are issued behind the
    // MFMAs of stage 2's first column block, and the 16 stores of every finished column block behind the MFMAs of the
    // next one (column block OUTER, k INNER: a block is complete after 16 MFMAs) -- only the last block's stores are
    // exposed.  The biases were requested with the stage-2 prefetch.
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(P.r, 0, npix * P.ldr * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(P.y, 0, npix * P.ldy * 4, RSRC_FLAGS);
    float rv[16];
    int roff[16];
    {
        const int ldr4 = P.ldr * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rowq + (r & 3) + 8 * (r >> 2);
            const int pix = __shfl(pix_lane, row, 64);
            rv[r] = relu1(acc1[r] + b1v);
            Rs[(32 * wq + row) * LDR + l31] = rv[r];
            roff[r] = pix >= 0 ? pix * ldr4 + l31 * 4 : OOB;
        }
    }
    // a wave reads back only the 32 rows it wrote itself (LDS operations of one wave complete in order): no barrier
    stamp(3);
    // ---- stage 2: 1x1 conv on the r tile, on top of x + b2
    const int relu_floor_bits = P.relu_out ? 0 : (int)0x80000000;
    {
        const float *a = Rs + (32 * wq + l31) * LDR + fk;
        float4 fa[CM / 8];
#pragma unroll
        for (int k8 = 0; k8 < CM / 8; ++k8) fa[k8] = *reinterpret_cast<const float4 *>(a + 8 * k8);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] += b2v[j];
#pragma unroll
            for (int k8 = 0; k8 < CM / 8; ++k8) {
                const float4 fb = w2f[j][k8];
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k8].x, fb.x, acc2[j], 0, 0, 0);
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k8].y, fb.y, acc2[j], 0, 0, 0);
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k8].z, fb.z, acc2[j], 0, 0, 0);
                acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k8].w, fb.w, acc2[j], 0, 0, 0);
                // four stores of the PREVIOUS block (or of r) behind these four MFMAs
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * k8 + q;
                    if (j == 0) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rv[r]), rr, roff[r], 0, 0);
                    } else {
                        const float v = relu_floor(acc2[j - 1][r], relu_floor_bits);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, yoff[r] + (j - 1) * 128, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    stamp(4);
    // ---- epilogue: optional trailing ReLU (vqvae.py:122,144) and store of the last column block
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = relu_floor(acc2[NJ - 1][r], relu_floor_bits);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, yoff[r] + (NJ - 1) * 128, 0, 0);
    }
    if (stamp_slot >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(5);
}


// ====================================================================== backward, data path
//     dh = (r > 0) * conv1x1_dgrad(g)                 [N,H,W,32]   (saved: both weight gradients need it)
//     dx = (x > 0) * conv3x3_dgrad(dh) + g            [N,H,W,128]
// One launch per block instead of two.  Phase A recomputes dh on the tile's 10x18 halo patch (1.4x of a
// GEMM that is 1/9 of the work) from 32-channel slices of g staged through LDS; the patch then stays in LDS
// and phase B feeds all nine taps of the 3x3 data gradient from it by shifting the fragment base, streaming
// only the weight panel of one tap at a time.  dh is never re-read from memory by this kernel.
namespace rbb {
using namespace rb;
constexpr int SA = 32;                 // g channels per phase-A slice
constexpr int LDA = SA + 4;
constexpr int PROWS = 192;             // patch rows padded to 6 MFMA row blocks
constexpr int GA_FLOATS = PROWS * LDA; // one g-patch slice buffer (also the size of the dh patch)
constexpr int WA_FLOATS = CM * LDA;    // one slice of the 1x1 panel [32 cm][32 co]
constexpr int WB_FLOATS = CC * LDA;    // one tap of the 3x3 panel [128 ci][32 cm]
constexpr size_t LDS_BYTES = (size_t)(2 * GA_FLOATS + 2 * WA_FLOATS) * sizeof(float);   // 64,512
constexpr int WR_FLOATS = 16 * 64 + 32;   // one wave's partial 32x32 tile of the 1x1 weight gradient + 32 bias partials
constexpr size_t LDS_BYTES_W2 = LDS_BYTES + (size_t)4 * WR_FLOATS * sizeof(float);     // 81,408: still two per CU
static_assert(GA_FLOATS + 2 * WB_FLOATS <= 2 * GA_FLOATS + 2 * WA_FLOATS, "phase-B buffers alias the phase-A ones");
}  // namespace rbb

struct ResBwdParams {
    const float *g;    // [N,H,W,ldg]  gradient of the block output
    const float *r;    // [N,H,W,ldr]  saved relu(conv3x3) activation (mask of the inner ReLU)
    const float *x;    // [N,H,W,ldx]  block input (mask of the outer ReLU)
    const float *w2d;  // VQ2_PACK_DGRAD panel of the 1x1 weight: [32 cm][128 co]
    const float *w1d;  // VQ2_PACK_DGRAD panel of the 3x3 weight: [128 ci][9 flipped taps][32 cm]
    float *dh;         // [N,H,W,lddh]
    float *dx;         // [N,H,W,lddx]
    float *w2_slab;    // [grid][128 co][32 cm] partial 1x1 weight gradients of each workgroup's own pixels, or null
    float *b2_slab;    // [grid][128 co]        partial 1x1 bias gradients (with w2_slab)
    int N, H, W, ldg, ldr, ldx, lddh, lddx;
    int tiles_x, tiles_y;
    int dephase, first_round;     // see dephase_start
    unsigned long long *stamps;   // diagnostic (vq2_debug_set_stamps): s_memtime at the phase boundaries of 2 workgroups
};

template <bool SAME_LD>
__global__ __launch_bounds__(256, 2) void resblock_bwd_data_kernel(const ResBwdParams P) {
    using namespace rbb;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ga = smem;                          // [2][GA_FLOATS]   phase A: g patch slices
    float *Wa = smem + 2 * GA_FLOATS;          // [2][WA_FLOATS]   phase A: 1x1 panel slices
    float *Dh = smem;                          // phase B: dh patch [192][LDA]          (aliases Ga[0])
    float *Wb = smem + GA_FLOATS;              // phase B: [2][WB_FLOATS] 3x3 tap panels (aliases Ga[1], Wa)

    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;
    const int l31 = lane & 31, fk = 4 * (lane >> 5), rowq = 4 * (lane >> 5);
    const int stamp_slot = (P.stamps && (blockIdx.x == 8 || blockIdx.x == 520)) ? (blockIdx.x == 8 ? 0 : 1) : -1;
    auto stamp = [&](int i) {
        if (stamp_slot >= 0 && lane == 0) {
            P.stamps[(stamp_slot * 4 + wq) * 8 + i] = __builtin_amdgcn_s_memtime();
            // slots 6 / 7: the 100 MHz real-time counter at the first / last stamp (in-kernel clock = cycles / time)
            if (i == 0 || i == 5) P.stamps[(stamp_slot * 4 + wq) * 8 + (i == 0 ? 6 : 7)] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    dephase_start(P.dephase, P.first_round);
    const int tiles = P.tiles_x * P.tiles_y;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int n = vid / tiles;
    const int t = vid - n * tiles;
    const int tyi = t / P.tiles_x;
    const int y0 = tyi * TH, x0 = (t - tyi * P.tiles_x) * TW;
    const int npix = P.N * P.H * P.W;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.g), 0, npix * P.ldg * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.r), 0, npix * P.ldr * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.x), 0, npix * P.ldx * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w2d), 0, CM * CC * 4, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w1d), 0, CC * 9 * CM * 4, RSRC_FLAGS);

    // pixel index of a patch row (or -1 outside the patch / the image)
    auto patch_pix = [&](int pp) {
        const int pr = pp / PW, pc = pp - pr * PW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        return (pp < NPATCH && (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W) ? (n * P.H + gy) * P.W + gx : -1;
    };

    // ---------------------------------------------------------------- phase A: dh on the halo patch
    constexpr int GA_LD = (NPATCH * (SA / 4) + 255) / 256;   // 6 float4 per thread per slice (1440 in all)
    int ga_off[GA_LD];
#pragma unroll
    for (int j = 0; j < GA_LD; ++j) {
        const int f = tid + 256 * j;
        const int pix = patch_pix(f >> 3);
        ga_off[j] = pix >= 0 ? pix * P.ldg * 4 + (f & 7) * 16 : OOB;
    }
    const int wa_off = ((tid >> 3) * CC) * 4 + (tid & 7) * 16;          // row cm = tid>>3 of the [32][128] panel
    const int st8 = (tid >> 3) * LDA + (tid & 7) * 4;                   // LDS float offset of float4 number tid (+32 rows per j)
    constexpr int NSA = CC / SA;
    struct SliceA { u32x4 g[GA_LD], w; };
    auto issue_a = [&](int s, SliceA &r) {
        const int soff = s * SA * 4;
#pragma unroll
        for (int j = 0; j < GA_LD; ++j) r.g[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, ga_off[j], soff, 0);
        r.w = __builtin_amdgcn_raw_buffer_load_b128(rw2, wa_off, soff, 0);
    };
    auto store_a = [&](int s, const SliceA &r) {
        const int buf = s & 1;
        float *a = Ga + buf * GA_FLOATS + st8;
        // GA_LD * 256 float4 = 192 rows exactly: every thread stores everything it loaded (rows >= 180: zeros)
#pragma unroll
        for (int j = 0; j < GA_LD; ++j) *reinterpret_cast<float4 *>(a + j * 32 * LDA) = u4_as_f4(r.g[j]);
        *reinterpret_cast<float4 *>(Wa + buf * WA_FLOATS + st8) = u4_as_f4(r.w);
    };
    static_assert(GA_LD * 256 == PROWS * (SA / 4), "the staging passes cover the padded patch exactly");
    // this wave's patch row blocks: wq, and wq + 4 for waves 0 and 1 (6 blocks of 32 rows cover the 180 patch rows)
    const int nblk = wq < 2 ? 2 : 1;
    SliceA RA0, RA1;
    issue_a(0, RA0);    // first in the queue: loads return in order, and only this slice gates the first MFMA
    issue_a(1, RA1);
    int pixA[2];        // pixel of patch row 32*b + l31 (this lane's GEMM row), per block
    float rmask[2][16];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        pixA[bi] = patch_pix(32 * (wq + 4 * bi) + l31);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            // (the load is UNCONDITIONAL with a poisoned offset for the waves that have no second block: written as
            //  `cond ? load : 0` hipcc branches around every load and waits vmcnt(0) behind each -- 16 dependent memory
            //  round trips in front of the first MFMA, the whole "prologue" of round 2's stamps)
            const int pix = __shfl(pixA[bi], rowq + (q & 3) + 8 * (q >> 2), 64);
            rmask[bi][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rr, (bi < nblk && pix >= 0) ? pix * P.ldr * 4 + l31 * 4 : OOB, 0, 0));
        }
    }
    f32x16 accA[2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int q = 0; q < 16; ++q) accA[bi][q] = 0.f;

    // Fused 1x1 weight gradient (w2_slab != null): dW2[co][cm] += sum over this tile's own pixels of g[px][co]*r[px][cm].
    // Phase A has g in LDS one 32-channel slice at a time, so slice s yields the 32x32 block of output channels
    // 32s..32s+31: every wave reduces over its own 32 pixels (16 MFMAs, A = g column from the slice buffer, B = r
    // fragments held in registers for the whole phase), the four partial blocks meet in an LDS scratch, and each
    // wave sums and stores a quarter.  Replaces a whole launch that re-read g and r from memory (HBM-bound).
    const bool fuse_w2 = P.w2_slab != nullptr;
    float *Wr = smem + 2 * GA_FLOATS + 2 * WA_FLOATS;      // [4 waves][WR_FLOATS], only allocated with fuse_w2
    float rB[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int q = 2 * kk + (lane >> 5);                // this lane's pixel of MFMA step kk: tile row 2*wq + q/16
        const int gy = y0 + 2 * wq + (q >> 4), gx = x0 + (q & 15);
        const int pix = (gy < P.H && gx < P.W) ? (n * P.H + gy) * P.W + gx : -1;
        rB[kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, (fuse_w2 && pix >= 0) ? pix * P.ldr * 4 + l31 * 4 : OOB, 0, 0));
    }
    // first tap panel of phase B: fetched behind the last phase-A slice
    const int wb_goff = ((tid >> 3) * 9 * CM) * 4 + (tid & 7) * 16;     // row ci = tid>>3 (+32 per j), tap 0
    u32x4 rwb[4];
    auto issue_b = [&](int tap) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rwb[j] = __builtin_amdgcn_raw_buffer_load_b128(rw1, wb_goff + j * 32 * 9 * CM * 4, tap * CM * 4, 0);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(Wb + buf * WB_FLOATS + st8 + j * 32 * LDA) = u4_as_f4(rwb[j]);
    };

    store_a(0, RA0);
    __syncthreads();
    stamp(1);
    // same pipeline as the forward kernel: slice s+1 goes from registers to LDS at the START of iteration s, the next
    // request follows at once, the barrier directly follows the MFMAs
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
        const int buf = s & 1;
        if (WRITE_AFTER) {
            if (s + 1 < NSA) store_a(s + 1, s == 0 ? RA1 : RA0);   // (slices 0 and 1 were requested back to back)
            if (s + 2 < NSA) issue_a(s + 2, RA0); else if (s + 2 == NSA) issue_b(0);
        } else {
            if (s > 0 && s + 1 < NSA) issue_a(s + 1, RA0); else if (s + 1 == NSA) issue_b(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float *b = Wa + buf * WA_FLOATS + l31 * LDA + fk;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                if (bi < nblk) {
                    const float *a = Ga + buf * GA_FLOATS + (32 * (wq + 4 * bi) + l31) * LDA + fk;
#pragma unroll
                    for (int k8 = 0; k8 < SA / 8; ++k8) {
                        const float4 fa = *reinterpret_cast<const float4 *>(a + 8 * k8);
                        const float4 fb = *reinterpret_cast<const float4 *>(b + 8 * k8);
                        accA[bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, accA[bi], 0, 0, 0);
                        accA[bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, accA[bi], 0, 0, 0);
                        accA[bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, accA[bi], 0, 0, 0);
                        accA[bi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, accA[bi], 0, 0, 0);
                    }
                }
            }
        }
        if (fuse_w2) {
            f32x16 aw;
#pragma unroll
            for (int r = 0; r < 16; ++r) aw[r] = 0.f;
            float bs = 0.f;
            const float *ga = Ga + buf * GA_FLOATS + l31;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int q = 2 * kk + (lane >> 5);
                const float a = ga[((2 * wq + (q >> 4) + 1) * PW + (q & 15) + 1) * LDA];
                aw = __builtin_amdgcn_mfma_f32_32x32x2f32(a, rB[kk], aw, 0, 0, 0);
                bs += a;
            }
            float *wr = Wr + wq * WR_FLOATS;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[r * 64 + lane] = aw[r];
            bs += __shfl_xor(bs, 32, 64);
            if (lane < 32) wr[16 * 64 + lane] = bs;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!WRITE_AFTER && s + 1 < NSA) store_a(s + 1, s == 0 ? RA1 : RA0);
        __syncthreads();
        if (fuse_w2) {   // wave wq sums accumulator rows 4*wq..4*wq+3 of the four partial blocks; wave 0 the bias partials
            float *slab = P.w2_slab + ((size_t)blockIdx.x * CC + s * 32) * CM;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * wq + i;
                const float v = (Wr[r * 64 + lane] + Wr[WR_FLOATS + r * 64 + lane]) + (Wr[2 * WR_FLOATS + r * 64 + lane] + Wr[3 * WR_FLOATS + r * 64 + lane]);
                slab[(rowq + (r & 3) + 8 * (r >> 2)) * CM + l31] = v;
            }
            if (wq == 0 && lane < 32)
                P.b2_slab[(size_t)blockIdx.x * CC + s * 32 + lane] =
                    (Wr[16 * 64 + lane] + Wr[WR_FLOATS + 16 * 64 + lane]) + (Wr[2 * WR_FLOATS + 16 * 64 + lane] + Wr[3 * WR_FLOATS + 16 * 64 + lane]);
            __syncthreads();   // the scratch is rewritten by the next slice
        }
    }
    stamp(2);
    // every wave is past its last phase-A fragment read: Dh and Wb may overwrite the slice buffers.  Tap 0's panel
    // (requested two slices ago) goes to LDS first and tap 1's is requested before the dh write, so that neither is
    // waited for at a barrier.
    store_b(0);
    if (WRITE_AFTER) issue_b(1);
    {
        const __amdgpu_buffer_rsrc_t rdh = __builtin_amdgcn_make_buffer_rsrc(P.dh, 0, npix * P.lddh * 4, RSRC_FLAGS);
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
            if (bi < nblk) {
                // interior patch rows (the tile's own pixels) also go to memory for the weight-gradient kernels
                const int pp = 32 * (wq + 4 * bi) + l31;
                const int pr = pp / PW, pc = pp - pr * PW;
                const int own = (pr >= 1 && pr <= TH && pc >= 1 && pc <= TW) ? pixA[bi] : -1;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = rowq + (q & 3) + 8 * (q >> 2);
                    const float v = rmask[bi][q] > 0.f ? accA[bi][q] : 0.f;
                    Dh[(32 * (wq + 4 * bi) + row) * LDA + l31] = v;
                    const int pix = __shfl(own, row, 64);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rdh, pix >= 0 ? pix * P.lddh * 4 + l31 * 4 : OOB, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    stamp(3);

    // ---------------------------------------------------------------- phase B: 3x3 data gradient from the dh patch
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    const int txr = l31 < 16 ? l31 : ((l31 + 14) & 15);      // rotated second row: conflict-free reads (see forward)
    const int a_frag = ((2 * wq + (l31 >> 4)) * PW + txr) * LDA + fk;
    // The epilogue operands are fetched BEHIND the MFMAs of phase B instead of in a burst after it (all resident
    // workgroups reach their epilogue together: 200 MB of mask + skip + store traffic in lock-step otherwise):
    // taps 0-3 fetch the outer-ReLU mask x of column block `tap` (folded to one bit per element a tap later),
    // taps 5-8 fetch the skip gradient g of column block `tap - 5` into registers.
    // byte offset of (pixel of accumulator register q, channel l31) in x -- and in g and dx when their pixel strides equal
    // x's (SAME_LD; every launch of the train step): computed ONCE, the column block goes into the immediate/scalar
    // offset.  (Recomputed per load it was ~6 vector instructions + an exec-mask pair per load, 16 loads per tap.)
    int poff[16];
    int pixr[SAME_LD ? 1 : 16];
    {
        const int gy = y0 + 2 * wq + (l31 >> 4), gx = x0 + txr;
        const int pix_lane = (gy < P.H && gx < P.W) ? (n * P.H + gy) * P.W + gx : -1;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int pix = __shfl(pix_lane, rowq + (q & 3) + 8 * (q >> 2), 64);
            poff[q] = pix >= 0 ? pix * (P.ldx * 4) + l31 * 4 : OOB;
            if (!SAME_LD) pixr[q] = pix;
        }
    }
    float mtmp[16], gres[4][16];
    unsigned mbits[2] = {0u, 0u};
    auto issue_b_one = [&](int tap, int j) {
        rwb[j] = __builtin_amdgcn_raw_buffer_load_b128(rw1, wb_goff + j * 32 * 9 * CM * 4, tap * CM * 4, 0);
    };
    auto store_b_one = [&](int buf, int j) {
        *reinterpret_cast<float4 *>(Wb + buf * WB_FLOATS + st8 + j * 32 * LDA) = u4_as_f4(rwb[j]);
    };
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int buf = tap & 1;
        if (!(WRITE_AFTER && INTERLEAVE)) {
            if (WRITE_AFTER) {
                if (tap + 1 < 9) store_b(buf ^ 1);
                if (tap + 2 < 9) issue_b(tap + 2);
            } else if (tap + 1 < 9) {
                issue_b(tap + 1);
            }
            if (tap < 4) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    mtmp[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, poff[q], tap * 128, 0));
            }
            if (tap >= 5) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    gres[tap - 5][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rg, SAME_LD ? poff[q] : (pixr[SAME_LD ? 0 : q] >= 0 ? pixr[SAME_LD ? 0 : q] * P.ldg * 4 + l31 * 4 : OOB),
                        (tap - 5) * 128, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float *a = Dh + a_frag + ((tap / 3) * PW + (tap % 3)) * LDA;
            const float *b = Wb + buf * WB_FLOATS + l31 * LDA + fk;
            if (WRITE_AFTER && INTERLEAVE) {
                // 16 groups of four MFMAs (k8 outer, column block inner); the fragments of group g+1 are requested before
                // the MFMAs of group g, and behind every group sit (between scheduling fences, see the forward kernel):
                //   groups 0-3    one LDS store of the NEXT tap's panel (registers requested during the previous tap)
                //   groups 4-7    one load of the panel after next
                //   groups 0-7    two mask loads (taps 0-3: they have eight groups to land before the fold below)
                //   groups 0-15   one skip-gradient load (taps 5-8: consumed in the epilogue only)
                float4 fa[2], fb[2];
                fa[0] = *reinterpret_cast<const float4 *>(a);
                fb[0] = *reinterpret_cast<const float4 *>(b);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int k8 = g >> 2, j = g & 3, cur = g & 1, nxt = cur ^ 1, ca = k8 & 1;
                    if (g + 1 < 16) {
                        const int k8n = (g + 1) >> 2, jn = (g + 1) & 3;
                        fb[nxt] = *reinterpret_cast<const float4 *>(b + jn * 32 * LDA + 8 * k8n);
                        if (jn == 0) fa[k8n & 1] = *reinterpret_cast<const float4 *>(a + 8 * k8n);
                    }
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ca].x, fb[cur].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ca].y, fb[cur].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ca].z, fb[cur].z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ca].w, fb[cur].w, acc[j], 0, 0, 0);
                    if (g + 1 < 16) {     // next group's LDS reads first, then this group's MFMAs
                        if (((g + 1) & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g < 4 && tap + 1 < 9) store_b_one(buf ^ 1, g);
                    if (g >= 4 && g < 8 && tap + 2 < 9) issue_b_one(tap + 2, g - 4);
                    if (tap < 4 && g < 8) {
#pragma unroll
                        for (int q = 2 * g; q < 2 * g + 2; ++q)
                            mtmp[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, poff[q], tap * 128, 0));
                    }
                    if (tap >= 5)
                        gres[tap - 5][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rg, SAME_LD ? poff[g] : (pixr[SAME_LD ? 0 : g] >= 0 ? pixr[SAME_LD ? 0 : g] * P.ldg * 4 + l31 * 4 : OOB),
                            (tap - 5) * 128, 0));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int k8 = 0; k8 < CM / 8; ++k8) {
                    const float4 fa = *reinterpret_cast<const float4 *>(a + 8 * k8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 fb = *reinterpret_cast<const float4 *>(b + j * 32 * LDA + 8 * k8);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[j], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tap < 4) {   // the mask values have landed by now
#pragma unroll
            for (int q = 0; q < 16; ++q) mbits[tap >> 1] |= (mtmp[q] > 0.f ? 1u : 0u) << ((tap & 1) * 16 + q);
        }
        if (!WRITE_AFTER && tap + 1 < 9) store_b(buf ^ 1);
        __syncthreads();
    }
    stamp(4);
    // ---------------------------------------------------------------- epilogue: outer ReLU mask, skip gradient
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(P.dx, 0, npix * P.lddx * 4, RSRC_FLAGS);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float v = ((mbits[j >> 1] >> ((j & 1) * 16 + q)) & 1u) ? acc[j][q] : 0.f;
            v += gres[j][q];
            __builtin_amdgcn_raw_buffer_store_b32(
                __float_as_uint(v), rdx,
                SAME_LD ? poff[q] : (pixr[SAME_LD ? 0 : q] >= 0 ? pixr[SAME_LD ? 0 : q] * P.lddx * 4 + l31 * 4 : OOB), j * 128, 0);
        }
    if (stamp_slot >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(5);
}

}  // namespace vq2

// De-phased start (dephase_start above): only when the launch has more workgroups than one per CU (otherwise nobody
// shares a SIMD) -- cycles from VQ2_RB_DEPHASE_FWD / VQ2_RB_DEPHASE_BWD (0 = off).
static void rb_dephase(int grid, int bwd, int *cycles, int *first_round) {
    static const int c_fwd = getenv("VQ2_RB_DEPHASE_FWD") ? atoi(getenv("VQ2_RB_DEPHASE_FWD")) : 0;
    static const int c_bwd = getenv("VQ2_RB_DEPHASE_BWD") ? atoi(getenv("VQ2_RB_DEPHASE_BWD")) : 0;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    *cycles = grid > cus ? (bwd ? c_bwd : c_fwd) : 0;
    *first_round = 2 * cus;
}

static unsigned long long *g_rb_stamps = nullptr, *g_rb_stamps_fwd = nullptr;
// buf[64]: backward kernel; buf + 64 (another 64 words): forward kernel (+ 32 more in the VQ2_RB_FINE diagnostic build)
extern "C" int vq2_debug_set_rb_stamps(unsigned long long *buf) { g_rb_stamps = buf; g_rb_stamps_fwd = buf ? buf + 64 : nullptr; return VQ2_OK; }

extern "C" int vq2_resblock_supported(int32_t C, int32_t Cm) { return (C == vq2::rb::CC && Cm == vq2::rb::CM) ? 1 : 0; }

extern "C" int vq2_resblock_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm, int flags, const float *x,
                                int32_t ldx, const float *w1p, const float *b1, const float *w2p, const float *b2,
                                float *r, int32_t ldr, float *y, int32_t ldy, vq2_stream_t stream) {
    using namespace vq2;
    VQ2_REQUIRE(N > 0 && H > 0 && W > 0, "resblock_fwd: empty tensor");
    if (!vq2_resblock_supported(C, Cm))
        return set_error(VQ2_ERR_UNSUPPORTED, "resblock_fwd: fused kernel is built for channel=%d, n_res_channel=%d (got %d, %d)",
                         rb::CC, rb::CM, C, Cm);
    VQ2_REQUIRE(x && w1p && b1 && w2p && b2 && r && y, "resblock_fwd: null pointer");
    VQ2_REQUIRE(aligned16(x) && aligned16(w1p) && aligned16(w2p) && aligned16(r) && aligned16(y),
                "resblock_fwd: pointers must be 16-byte aligned");
    VQ2_REQUIRE(ldx >= C && ldy >= C && ldr >= Cm && ldx % 4 == 0 && ldy % 4 == 0 && ldr % 4 == 0,
                "resblock_fwd: pixel strides must cover the channels and be multiples of 4");
    VQ2_REQUIRE((flags & ~VQ2_RELU_OUT) == 0, "resblock_fwd: only VQ2_RELU_OUT is a valid flag");
    const double npix = (double)N * H * W;
    const int ldmax = ldx > ldy ? ldx : ldy;
    VQ2_REQUIRE(npix * ldmax * 4.0 < (double)rb::OOB, "resblock_fwd: tensors must be smaller than %d bytes", rb::OOB);
    {   // rows of whole 64-pixel segments: the 3x3 in the Winograd domain (vq2_rbwino.hip)
        RbwFwdParams Q{};
        Q.x = x; Q.w1 = w1p; Q.b1 = b1; Q.w2 = w2p; Q.b2 = b2; Q.r = r; Q.y = y;
        Q.N = N; Q.H = H; Q.W = W; Q.ldx = ldx; Q.ldr = ldr; Q.ldy = ldy; Q.relu_out = (flags & VQ2_RELU_OUT) != 0;
        if (ldr >= Cm && rbw_fwd_ok(Q) && !g_rb_stamps_fwd) {
            hipStream_t s = to_stream(stream);
            const char *name = "resblock_fwd_wino";
            if (prof_enabled()) name = prof_label("resblock_fwd_wino|M=%d,C=%d,Cm=%d", N * H * W, C, Cm);
            ProfScope prof(name, 2.0 * npix * (9.0 * C * Cm + (double)Cm * C), 4.0 * npix * (2.0 * C + Cm), s);
            return launch_rbw_fwd(Q, s);
        }
    }
    ResFwdParams P{};
    P.x = x; P.w1 = w1p; P.b1 = b1; P.w2 = w2p; P.b2 = b2; P.r = r; P.y = y;
    P.N = N; P.H = H; P.W = W; P.ldx = ldx; P.ldr = ldr; P.ldy = ldy;
    P.tiles_x = (W + rb::TW - 1) / rb::TW; P.tiles_y = (H + rb::TH - 1) / rb::TH;
    P.relu_out = (flags & VQ2_RELU_OUT) != 0;
    const int grid = N * P.tiles_x * P.tiles_y;
    rb_dephase(grid, 0, &P.dephase, &P.first_round);
    P.stamps = g_rb_stamps_fwd;
    hipStream_t s = to_stream(stream);
    const char *name = "resblock_fwd";
    if (prof_enabled()) name = prof_label("resblock_fwd|M=%d,C=%d,Cm=%d", N * H * W, C, Cm);
    ProfScope prof(name, 2.0 * npix * (9.0 * C * Cm + (double)Cm * C), 4.0 * npix * (2.0 * C + Cm), s);
    allow_big_lds(resblock_fwd_kernel, rb::LDS_BYTES);
    hipLaunchKernelGGL(resblock_fwd_kernel, dim3(grid), dim3(256), rb::LDS_BYTES, s, P);
    return check_launch("resblock_fwd_kernel");
}

static int rb_grid(int N, int H, int W) { return N * ((W + vq2::rb::TW - 1) / vq2::rb::TW) * ((H + vq2::rb::TH - 1) / vq2::rb::TH); }

extern "C" size_t vq2_resblock_w2_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm) {
    if (N <= 0 || H <= 0 || W <= 0 || !vq2_resblock_supported(C, Cm)) return 0;
    return (size_t)rb_grid(N, H, W) * ((size_t)C * Cm + C) * sizeof(float);
}

extern "C" int vq2_resblock_w2_job_init(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm, const void *ws, float *dw,
                                        float *db, vq2_wgrad_job *job) {
    VQ2_REQUIRE(ws && dw && job && vq2_resblock_supported(C, Cm) && N > 0 && H > 0 && W > 0, "resblock_w2_job_init: bad arguments");
    const int S = rb_grid(N, H, W);
    const float *w = static_cast<const float *>(ws);
    job->ws = w; job->dw = dw; job->bias_ws = db ? w + (size_t)S * C * Cm : nullptr; job->db = db;
    job->unit_offset = 0;
    job->O = C; job->I = Cm; job->Or = C; job->Ir = Cm; job->taps = 1; job->S = S;
    job->n_units_w = (C * Cm + 31) / 32; job->n_units_b = db ? (C + 31) / 32 : 0;
    job->swapped = 0; job->bias_splits = 0;
    return VQ2_OK;
}

extern "C" int vq2_resblock_bwd_data(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm, const float *g, int32_t ldg,
                                     const float *r, int32_t ldr, const float *x, int32_t ldx, const float *w2d,
                                     const float *w1d, float *dh, int32_t lddh, float *dx, int32_t lddx, void *w2_ws,
                                     vq2_stream_t stream) {
    using namespace vq2;
    VQ2_REQUIRE(N > 0 && H > 0 && W > 0, "resblock_bwd_data: empty tensor");
    if (!vq2_resblock_supported(C, Cm))
        return set_error(VQ2_ERR_UNSUPPORTED, "resblock_bwd_data: fused kernel is built for channel=%d, n_res_channel=%d (got %d, %d)",
                         rb::CC, rb::CM, C, Cm);
    VQ2_REQUIRE(g && r && x && w2d && w1d && dh && dx, "resblock_bwd_data: null pointer");
    VQ2_REQUIRE(aligned16(g) && aligned16(r) && aligned16(x) && aligned16(w2d) && aligned16(w1d) && aligned16(dh) && aligned16(dx),
                "resblock_bwd_data: pointers must be 16-byte aligned");
    VQ2_REQUIRE(ldg >= C && ldx >= C && lddx >= C && ldr >= Cm && lddh >= Cm && ldg % 4 == 0 && ldx % 4 == 0 &&
                    lddx % 4 == 0 && ldr % 4 == 0 && lddh % 4 == 0,
                "resblock_bwd_data: pixel strides must cover the channels and be multiples of 4");
    const double npix = (double)N * H * W;
    int ldmax = ldg > ldx ? ldg : ldx;
    if (lddx > ldmax) ldmax = lddx;
    VQ2_REQUIRE(npix * ldmax * 4.0 < (double)rb::OOB, "resblock_bwd_data: tensors must be smaller than %d bytes", rb::OOB);
    ResBwdParams P{};
    P.g = g; P.r = r; P.x = x; P.w2d = w2d; P.w1d = w1d; P.dh = dh; P.dx = dx;
    VQ2_REQUIRE(!w2_ws || aligned16(w2_ws), "resblock_bwd_data: workspace must be 16-byte aligned");
    P.N = N; P.H = H; P.W = W; P.ldg = ldg; P.ldr = ldr; P.ldx = ldx; P.lddh = lddh; P.lddx = lddx;
    P.tiles_x = (W + rb::TW - 1) / rb::TW; P.tiles_y = (H + rb::TH - 1) / rb::TH;
    const int grid = N * P.tiles_x * P.tiles_y;
    P.stamps = g_rb_stamps;
    hipStream_t s = to_stream(stream);
    const char *name = "resblock_bwd_data";
    if (prof_enabled()) name = prof_label("resblock_bwd_data|M=%d,C=%d,Cm=%d", N * H * W, C, Cm);
    ProfScope prof(name, 2.0 * npix * (9.0 * C * Cm + (double)Cm * C), 4.0 * npix * (4.0 * C + 2.0 * Cm), s);
    P.w2_slab = static_cast<float *>(w2_ws);
    P.b2_slab = w2_ws ? P.w2_slab + (size_t)grid * C * Cm : nullptr;
    const size_t lds = w2_ws ? rbb::LDS_BYTES_W2 : rbb::LDS_BYTES;
    rb_dephase(grid, 1, &P.dephase, &P.first_round);
    if (ldg == ldx && lddx == ldx) {
        allow_big_lds(resblock_bwd_data_kernel<true>, lds);
        hipLaunchKernelGGL(resblock_bwd_data_kernel<true>, dim3(grid), dim3(256), lds, s, P);
    } else {
        allow_big_lds(resblock_bwd_data_kernel<false>, lds);
        hipLaunchKernelGGL(resblock_bwd_data_kernel<false>, dim3(grid), dim3(256), lds, s, P);
    }
    return check_launch("resblock_bwd_data_kernel");
}
