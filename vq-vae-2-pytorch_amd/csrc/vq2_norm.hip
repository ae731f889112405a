// AdaIN of the VQVAE_Deep decoder (vqvae_deep.py:99-109) for gfx950:
//     y = [relu]( (1 + gamma[n,c]) * (x - mean[n,c]) * rstd[n,c] + beta[n,c] ),   (gamma | beta) = fc(style)[n, 2C]
// with mean / rstd = nn.InstanceNorm2d(C, affine=False) statistics (biased variance, eps inside the root) over the
// H*W pixels of image n.  NHWC: a wave reads 64 consecutive channels of a pixel (256 B, coalesced); the pixel
// dimension is split over the 16 waves of a workgroup in a fixed pattern and folded in a fixed order, so every
// statistic and gradient sum is bit-reproducible.  All of it is HBM-bound elementwise / column-reduction work.
#include "vq2_common.h"

namespace vq2 {

constexpr int NRM_PG = 16;   // pixel groups (waves) per workgroup; 64 channels per workgroup

// out[g][cl] partials of a per-(n, channel) sum over pixels: thread (cl, pg) covers pixels pg, pg + 16, ...
template <typename F>
__device__ __forceinline__ float column_sum(F term, int64_t HW, int pg, float (*red)[64], int cl) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains keep four loads in flight
    int64_t p = pg;
    for (; p + 3 * NRM_PG < HW; p += 4 * NRM_PG) {
        s0 += term(p);
        s1 += term(p + NRM_PG);
        s2 += term(p + 2 * NRM_PG);
        s3 += term(p + 3 * NRM_PG);
    }
    for (; p < HW; p += NRM_PG) s0 += term(p);
    red[pg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < NRM_PG; ++g) t += red[g][cl];   // every thread of a column forms the same sum, in the same order
    __syncthreads();
    return t;
}

// grid (ceil(C/64), N), block 1024
__global__ __launch_bounds__(64 * NRM_PG) void instnorm_stats_kernel(const float *__restrict__ x, int ldx, int64_t HW, int C,
                                                                    float eps, float *__restrict__ mean,
                                                                    float *__restrict__ rstd) {
    __shared__ float red[NRM_PG][64];
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
    const bool cv = c < C;
    const float *xp = x + (size_t)n * HW * ldx + (cv ? c : 0);
    const float inv = 1.f / (float)HW;
    const float m = column_sum([&](int64_t p) { return cv ? xp[p * ldx] : 0.f; }, HW, pg, red, cl) * inv;
    const float v = column_sum([&](int64_t p) { const float d = (cv ? xp[p * ldx] : m) - m; return d * d; }, HW, pg, red, cl) * inv;
    if (cv && pg == 0) {
        mean[(size_t)n * C + c] = m;
        rstd[(size_t)n * C + c] = 1.f / sqrtf(v + eps);
    }
}

__global__ __launch_bounds__(256) void adain_fwd_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ mean,
                                                        const float *__restrict__ rstd, const float *__restrict__ h,
                                                        int64_t HW, int C, int relu, float *__restrict__ y, int ldy,
                                                        int64_t total4) {
    const int C4 = C >> 2;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256) {
        const int64_t pix = t / C4;
        const int c = (int)(t - pix * C4) * 4;
        const int64_t n = pix / HW;
        const float4 xv = *reinterpret_cast<const float4 *>(x + pix * ldx + c);
        const float4 mu = *reinterpret_cast<const float4 *>(mean + n * C + c);
        const float4 rs = *reinterpret_cast<const float4 *>(rstd + n * C + c);
        const float4 ga = *reinterpret_cast<const float4 *>(h + n * 2 * C + c);
        const float4 be = *reinterpret_cast<const float4 *>(h + n * 2 * C + C + c);
        float4 o;   // (1 + gamma) * norm(x) + beta, in the reference's order of operations (vqvae_deep.py:109)
        o.x = (1.f + ga.x) * ((xv.x - mu.x) * rs.x) + be.x;
        o.y = (1.f + ga.y) * ((xv.y - mu.y) * rs.y) + be.y;
        o.z = (1.f + ga.z) * ((xv.z - mu.z) * rs.z) + be.z;
        o.w = (1.f + ga.w) * ((xv.w - mu.w) * rs.w) + be.w;
        if (relu) o = relu4(o);
        *reinterpret_cast<float4 *>(y + pix * ldy + c) = o;
    }
}

// dh[n][c] = sum_p dz * xhat (gradient of gamma), dh[n][C + c] = sum_p dz (gradient of beta); dz = dy * (y > 0)
__global__ __launch_bounds__(64 * NRM_PG) void adain_bwd_reduce_kernel(const float *__restrict__ dy, int lddy,
                                                                      const float *__restrict__ y, int ldy,
                                                                      const float *__restrict__ x, int ldx,
                                                                      const float *__restrict__ mean,
                                                                      const float *__restrict__ rstd, int64_t HW, int C,
                                                                      float *__restrict__ dh) {
    __shared__ float red[NRM_PG][64];
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
    const bool cv = c < C;
    const size_t base = (size_t)n * HW;
    const int cc = cv ? c : 0;
    const float m = mean[(size_t)n * C + cc], r = rstd[(size_t)n * C + cc];
    auto dz = [&](int64_t p) {
        const float g = dy[(base + p) * lddy + cc];
        return (!y || y[(base + p) * ldy + cc] > 0.f) ? g : 0.f;
    };
    const float sb = column_sum([&](int64_t p) { return cv ? dz(p) : 0.f; }, HW, pg, red, cl);
    const float sg = column_sum([&](int64_t p) { return cv ? dz(p) * ((x[(base + p) * ldx + cc] - m) * r) : 0.f; }, HW, pg, red, cl);
    if (cv && pg == 0) {
        dh[(size_t)n * 2 * C + c] = sg;
        dh[(size_t)n * 2 * C + C + c] = sb;
    }
}

// dx = rstd * (1 + gamma) * (dz - mean_p(dz) - xhat * mean_p(dz * xhat))
__global__ __launch_bounds__(256) void adain_bwd_apply_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ y,
                                                              int ldy, const float *__restrict__ x, int ldx,
                                                              const float *__restrict__ mean, const float *__restrict__ rstd,
                                                              const float *__restrict__ h, const float *__restrict__ dh,
                                                              int64_t HW, int C, float *__restrict__ dx, int lddx,
                                                              int64_t total4) {
    const int C4 = C >> 2;
    const float inv = 1.f / (float)HW;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256) {
        const int64_t pix = t / C4;
        const int c = (int)(t - pix * C4) * 4;
        const int64_t n = pix / HW;
        const float4 g = *reinterpret_cast<const float4 *>(dy + pix * lddy + c);
        float4 z = g;
        if (y) {
            const float4 yv = *reinterpret_cast<const float4 *>(y + pix * ldy + c);
            z.x = yv.x > 0.f ? g.x : 0.f; z.y = yv.y > 0.f ? g.y : 0.f; z.z = yv.z > 0.f ? g.z : 0.f; z.w = yv.w > 0.f ? g.w : 0.f;
        }
        const float4 xv = *reinterpret_cast<const float4 *>(x + pix * ldx + c);
        const float4 mu = *reinterpret_cast<const float4 *>(mean + n * C + c);
        const float4 rs = *reinterpret_cast<const float4 *>(rstd + n * C + c);
        const float4 ga = *reinterpret_cast<const float4 *>(h + n * 2 * C + c);
        const float4 sg = *reinterpret_cast<const float4 *>(dh + n * 2 * C + c);
        const float4 sb = *reinterpret_cast<const float4 *>(dh + n * 2 * C + C + c);
        float4 o;
        o.x = rs.x * (1.f + ga.x) * (z.x - sb.x * inv - ((xv.x - mu.x) * rs.x) * (sg.x * inv));
        o.y = rs.y * (1.f + ga.y) * (z.y - sb.y * inv - ((xv.y - mu.y) * rs.y) * (sg.y * inv));
        o.z = rs.z * (1.f + ga.z) * (z.z - sb.z * inv - ((xv.z - mu.z) * rs.z) * (sg.z * inv));
        o.w = rs.w * (1.f + ga.w) * (z.w - sb.w * inv - ((xv.w - mu.w) * rs.w) * (sg.w * inv));
        *reinterpret_cast<float4 *>(dx + pix * lddx + c) = o;
    }
}

static inline int nrm_grid(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace vq2

using namespace vq2;

#define VQ2_NORM_COMMON(what)                                                                                          \
    VQ2_REQUIRE(N > 0 && N <= 65535 && HW > 0 && C > 0 && C % 4 == 0, what ": need 0 < N <= 65535, HW > 0, C %% 4 == 0"); \
    VQ2_REQUIRE((double)N * (double)HW * 4.0 < 2147483648.0 * 4.0, what ": tensor too large")

extern "C" int vq2_instnorm_stats(const float *x, int32_t ldx, int32_t N, int64_t HW, int32_t C, double eps, float *mean,
                                  float *rstd, vq2_stream_t stream) {
    VQ2_REQUIRE(x && mean && rstd, "instnorm_stats: null pointer");
    VQ2_NORM_COMMON("instnorm_stats");
    VQ2_REQUIRE(ldx >= C, "instnorm_stats: pixel stride below the channel count");
    hipLaunchKernelGGL(instnorm_stats_kernel, dim3((C + 63) / 64, N), dim3(64 * NRM_PG), 0, to_stream(stream), x, ldx, HW, C,
                       (float)eps, mean, rstd);
    return check_launch("instnorm_stats_kernel");
}

extern "C" int vq2_adain_fwd(const float *x, int32_t ldx, const float *mean, const float *rstd, const float *h, int32_t N,
                             int64_t HW, int32_t C, int flags, float *y, int32_t ldy, vq2_stream_t stream) {
    VQ2_REQUIRE(x && mean && rstd && h && y, "adain_fwd: null pointer");
    VQ2_NORM_COMMON("adain_fwd");
    VQ2_REQUIRE(ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(h) &&
                    aligned16(mean) && aligned16(rstd), "adain_fwd: strides must be multiples of 4 >= C, pointers 16-byte aligned");
    VQ2_REQUIRE((flags & ~VQ2_RELU_OUT) == 0, "adain_fwd: only VQ2_RELU_OUT is a valid flag");
    const int64_t total4 = (int64_t)N * HW * (C / 4);
    hipLaunchKernelGGL(adain_fwd_kernel, dim3(nrm_grid(total4)), dim3(256), 0, to_stream(stream), x, ldx, mean, rstd, h, HW, C,
                       (flags & VQ2_RELU_OUT) ? 1 : 0, y, ldy, total4);
    return check_launch("adain_fwd_kernel");
}

extern "C" int vq2_adain_bwd(const float *dy, int32_t lddy, const float *y, int32_t ldy, const float *x, int32_t ldx,
                             const float *mean, const float *rstd, const float *h, int32_t N, int64_t HW, int32_t C,
                             float *dh, float *dx, int32_t lddx, vq2_stream_t stream) {
    VQ2_REQUIRE(dy && x && mean && rstd && h && dh && dx, "adain_bwd: null pointer");
    VQ2_NORM_COMMON("adain_bwd");
    VQ2_REQUIRE(lddy >= C && ldx >= C && lddx >= C && (!y || ldy >= C) && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 &&
                    (!y || ldy % 4 == 0), "adain_bwd: pixel strides must be multiples of 4 >= C");
    VQ2_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(h) && aligned16(dh) && (!y || aligned16(y)) &&
                    aligned16(mean) && aligned16(rstd), "adain_bwd: pointers must be 16-byte aligned");
    hipStream_t s = to_stream(stream);
    hipLaunchKernelGGL(adain_bwd_reduce_kernel, dim3((C + 63) / 64, N), dim3(64 * NRM_PG), 0, s, dy, lddy, y, ldy, x, ldx, mean,
                       rstd, HW, C, dh);
    if (int e = check_launch("adain_bwd_reduce_kernel")) return e;
    const int64_t total4 = (int64_t)N * HW * (C / 4);
    hipLaunchKernelGGL(adain_bwd_apply_kernel, dim3(nrm_grid(total4)), dim3(256), 0, s, dy, lddy, y, ldy, x, ldx, mean, rstd, h, dh,
                       HW, C, dx, lddx, total4);
    return check_launch("adain_bwd_apply_kernel");
}
