// HBM-bound helpers of the stage-1 step: layout conversion at the NCHW module boundary,
// fused ReLU backward, channel-slice copy/accumulate, MSE loss + gradient (train_vqvae.py:31,83),
// Adam (train_vqvae.py:185).  All 16-byte vectorised, grid-stride, deterministic reductions.
#include "vq2_common.h"

namespace vq2 {

static inline int grid_for(int64_t work_items, int cap = 4096) {
    int64_t b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// [N][C][HW] -> [N][HW][ld] through a 32x33 LDS tile; channels >= C are written as zeros
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ src, float *__restrict__ dst, int C,
                                                           int HW, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < HW && c < ld) dst[((size_t)n * HW + p) * ld + c] = tile[tx][j];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ src, float *__restrict__ dst, int C,
                                                           int HW, int ld) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        tile[j][tx] = (p < HW && c < C) ? src[((size_t)n * HW + p) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (c < C && p < HW) dst[((size_t)n * C + c) * HW + p] = tile[tx][j];
    }
}

// C <= 4 with pixel stride 4 (the image / reconstruction): one thread per pixel, plane reads coalesce
// across threads, one 16-byte store (resp. load) per pixel
__global__ void nchw_to_nhwc4_kernel(const float *__restrict__ src, float4 *__restrict__ dst, int C, int64_t HW,
                                     int64_t total) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = g / HW, p = g - n * HW;
        const float *s = src + n * C * HW + p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = s[0];
        if (C > 1) v.y = s[HW];
        if (C > 2) v.z = s[2 * HW];
        if (C > 3) v.w = s[3 * HW];
        dst[g] = v;
    }
}

__global__ void nhwc4_to_nchw_kernel(const float4 *__restrict__ src, float *__restrict__ dst, int C, int64_t HW,
                                     int64_t total) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = g / HW, p = g - n * HW;
        const float4 v = src[g];
        float *d = dst + n * C * HW + p;
        d[0] = v.x;
        if (C > 1) d[HW] = v.y;
        if (C > 2) d[2 * HW] = v.z;
        if (C > 3) d[3 * HW] = v.w;
    }
}

__global__ void relu_bwd_kernel(const float *__restrict__ dy, int lddy, const float *__restrict__ y, int ldy,
                                float *__restrict__ g, int ldg, int64_t pixels, int C4) {
    const int64_t total = pixels * C4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / C4;
        const int c = (int)(t - p * C4) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(dy + p * lddy + c);
        const float4 b = *reinterpret_cast<const float4 *>(y + p * ldy + c);
        float4 o;
        o.x = b.x > 0.f ? a.x : 0.f; o.y = b.y > 0.f ? a.y : 0.f; o.z = b.z > 0.f ? a.z : 0.f; o.w = b.w > 0.f ? a.w : 0.f;
        *reinterpret_cast<float4 *>(g + p * ldg + c) = o;
    }
}

__global__ void slice_copy_kernel(const float *__restrict__ src, int lds_, float *__restrict__ dst, int ldd,
                                  int64_t pixels, int C4, int accumulate) {
    const int64_t total = pixels * C4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / C4;
        const int c = (int)(t - p * C4) * 4;
        float4 v = *reinterpret_cast<const float4 *>(src + p * lds_ + c);
        float4 *o = reinterpret_cast<float4 *>(dst + p * ldd + c);
        if (accumulate) { const float4 w = *o; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        *o = v;
    }
}

constexpr int MSE_BLOCKS = 1024;

__global__ __launch_bounds__(256) void mse_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          int64_t n, float coef, const float *__restrict__ gscale,
                                                          float *__restrict__ grad, float *__restrict__ part) {
    __shared__ float wsum[4];
    const float gs = coef * (gscale ? gscale[0] : 1.f);
    float s = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        s += d0 * d0; s += d1 * d1; s += d2 * d2; s += d3 * d3;
        if (grad) reinterpret_cast<float4 *>(grad)[i] = make_float4(gs * d0, gs * d1, gs * d2, gs * d3);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // ragged tail
        const int64_t i = n4 * 4 + threadIdx.x;
        const float d = a[i] - b[i];
        s += d * d;
        if (grad) grad[i] = gs * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// latent != null: also total[0] = loss + weight * latent[0], the `recon_loss + 0.25 * latent_loss` of train_vqvae.py:85
// (same two fp32 operations as the stand-alone axpby launch it replaces)
__global__ void mse_final_kernel(const float *__restrict__ part, int nparts, float denom, float *__restrict__ loss,
                                 const float *__restrict__ latent, float weight, float *__restrict__ total) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l = red[0] / denom;
        loss[0] = l;
        if (latent) total[0] = l + weight * latent[0];
    }
}

// torch.optim.Adam single-tensor math (lerp / addcmul / addcdiv), one pass over a flat arena
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, int64_t n, float w1, float b2, float w2, float bc2_sqrt, float eps,
                            float neg_step, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        float mi = m[i], vi = v[i];
        mi = mi + w1 * (gi - mi);                 // exp_avg.lerp_(grad, 1-beta1)
        vi = vi * b2 + (w2 * gi) * gi;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] + neg_step * (mi / denom);    // param.addcdiv_(exp_avg, denom, value=-step_size)
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ void axpby_kernel(const float *__restrict__ a, const float *__restrict__ b, float alpha,
                             float *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = a[i] + alpha * b[i];
}

__global__ void scale_kernel(const float *__restrict__ src, const float *__restrict__ scalar, float alpha,
                             float *__restrict__ dst, int64_t n) {
    const float s = scalar[0] * alpha;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i] * s;
}

}  // namespace vq2

using namespace vq2;

extern "C" int vq2_nchw_to_nhwc(const float *src, float *dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ld,
                                vq2_stream_t stream) {
    VQ2_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "nchw_to_nhwc: bad arguments");
    if (ld == 4 && aligned16(dst)) {
        const int64_t HW4 = (int64_t)H * W, total = HW4 * N;
        hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(grid_for(total)), dim3(256), 0, to_stream(stream), src,
                           reinterpret_cast<float4 *>(dst), C, HW4, total);
        return check_launch("nchw_to_nhwc4_kernel");
    }
    VQ2_REQUIRE(N <= 65535, "nchw_to_nhwc: batch > 65535");
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (ld + 31) / 32, N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, to_stream(stream), src, dst, C, HW, ld);
    return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int vq2_nhwc_to_nchw(const float *src, float *dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ld,
                                vq2_stream_t stream) {
    VQ2_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, "nhwc_to_nchw: bad arguments");
    if (ld == 4 && aligned16(src)) {
        const int64_t HW4 = (int64_t)H * W, total = HW4 * N;
        hipLaunchKernelGGL(nhwc4_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, to_stream(stream),
                           reinterpret_cast<const float4 *>(src), dst, C, HW4, total);
        return check_launch("nhwc4_to_nchw_kernel");
    }
    VQ2_REQUIRE(N <= 65535, "nhwc_to_nchw: batch > 65535");
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, to_stream(stream), src, dst, C, HW, ld);
    return check_launch("nhwc_to_nchw_kernel");
}

extern "C" int vq2_relu_bwd(const float *dy, int32_t lddy, const float *y, int32_t ldy, float *g, int32_t ldg,
                            int64_t pixels, int32_t C, vq2_stream_t stream) {
    VQ2_REQUIRE(dy && y && g && pixels > 0 && C > 0 && C % 4 == 0, "relu_bwd: need C %% 4 == 0");
    VQ2_REQUIRE(lddy >= C && ldy >= C && ldg >= C && lddy % 4 == 0 && ldy % 4 == 0 && ldg % 4 == 0,
                "relu_bwd: bad pixel strides");
    VQ2_REQUIRE(aligned16(dy) && aligned16(y) && aligned16(g), "relu_bwd: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, to_stream(stream), dy, lddy, y,
                       ldy, g, ldg, pixels, C / 4);
    return check_launch("relu_bwd_kernel");
}

extern "C" int vq2_slice_copy(const float *src, int32_t lds_, float *dst, int32_t ldd, int64_t pixels, int32_t C,
                              int accumulate, vq2_stream_t stream) {
    VQ2_REQUIRE(src && dst && pixels > 0 && C > 0 && C % 4 == 0 && lds_ >= C && ldd >= C && lds_ % 4 == 0 && ldd % 4 == 0,
                "slice_copy: bad arguments");
    VQ2_REQUIRE(aligned16(src) && aligned16(dst), "slice_copy: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(slice_copy_kernel, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, to_stream(stream), src, lds_,
                       dst, ldd, pixels, C / 4, accumulate);
    return check_launch("slice_copy_kernel");
}

extern "C" size_t vq2_mse_workspace_bytes(int64_t numel) { return numel > 0 ? MSE_BLOCKS * sizeof(float) : 0; }

static int mse_impl(const float *a, const float *b, int64_t numel, int64_t denom, const float *gscale, float *loss,
                    float *grad, const float *latent, float weight, float *total, void *ws, size_t ws_bytes,
                    vq2_stream_t stream);

extern "C" int vq2_mse_fwd_bwd(const float *a, const float *b, int64_t numel, int64_t denom, const float *gscale,
                               float *loss, float *grad, void *ws, size_t ws_bytes, vq2_stream_t stream) {
    return mse_impl(a, b, numel, denom, gscale, loss, grad, nullptr, 0.f, nullptr, ws, ws_bytes, stream);
}

extern "C" int vq2_stage1_loss(const float *a, const float *b, int64_t numel, int64_t denom, const float *latent,
                               float weight, float *recon, float *total, float *grad, void *ws, size_t ws_bytes,
                               vq2_stream_t stream) {
    VQ2_REQUIRE(latent && total, "stage1_loss: null pointer");
    return mse_impl(a, b, numel, denom, nullptr, recon, grad, latent, weight, total, ws, ws_bytes, stream);
}

static int mse_impl(const float *a, const float *b, int64_t numel, int64_t denom, const float *gscale, float *loss,
                    float *grad, const float *latent, float weight, float *total, void *ws, size_t ws_bytes,
                    vq2_stream_t stream) {
    VQ2_REQUIRE(a && b && loss && ws && numel > 0 && denom > 0, "mse: bad arguments");
    VQ2_REQUIRE(aligned16(a) && aligned16(b) && (!grad || aligned16(grad)), "mse: pointers must be 16-byte aligned");
    VQ2_REQUIRE(ws_bytes >= MSE_BLOCKS * sizeof(float), "mse: workspace too small");
    const int blocks = grid_for(numel / 4 + 1, MSE_BLOCKS);
    hipStream_t s = to_stream(stream);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, s, a, b, numel, (float)(2.0 / (double)denom),
                       gscale, grad, static_cast<float *>(ws));
    if (int e = check_launch("mse_partial_kernel")) return e;
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, static_cast<const float *>(ws), blocks,
                       (float)denom, loss, latent, weight, total);
    return check_launch("mse_final_kernel");
}

extern "C" int vq2_adam_step(float *p, const float *g, float *m, float *v, int64_t n, double lr, double beta1,
                             double beta2, double eps, int32_t step, double grad_scale, vq2_stream_t stream) {
    VQ2_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam: bad arguments");
    // scalar prep in double exactly as torch/optim/adam.py does on the host
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1;
    const double bc2_sqrt = sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, to_stream(stream), p, g, m, v, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)bc2_sqrt, (float)eps,
                       (float)(-step_size), (float)grad_scale);
    return check_launch("adam_kernel");
}

extern "C" int vq2_axpby(const float *a, const float *b, float alpha, float *dst, int64_t n, vq2_stream_t stream) {
    VQ2_REQUIRE(a && b && dst && n > 0, "axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, to_stream(stream), a, b, alpha, dst, n);
    return check_launch("axpby_kernel");
}

extern "C" int vq2_scale(const float *src, const float *scalar, float alpha, float *dst, int64_t n,
                         vq2_stream_t stream) {
    VQ2_REQUIRE(src && scalar && dst && n > 0, "scale: bad arguments");
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, to_stream(stream), src, scalar, alpha, dst, n);
    return check_launch("scale_kernel");
}

// ------------------------------------------------------------------ calibration (not on the product path)
// Register-resident v_mfma_f32_32x32x2_f32 loop: what the chip sustains for this instruction at the
// clock it holds under load -- the practical ceiling the conv kernels are measured against in DESIGN.md.
__global__ __launch_bounds__(256) void mfma_peak_kernel(float *out, int iters, float seed) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            a += 1e-6f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;  // keep the loop alive
}

// the same arithmetic rate through v_mfma_f32_16x16x4_f32 (32 cycles per instruction, half the FLOPs): does the chip
// hold a different clock for this shape (MI355X_MICROARCH.md, DVFS note 7)?
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_peak16_kernel(float *out, int iters, float seed) {
    f32x4v acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            a += 1e-6f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

extern "C" int vq2_debug_mfma_peak16(float *scratch, int32_t blocks, int32_t iters, vq2_stream_t stream) {
    VQ2_REQUIRE(scratch && blocks > 0 && iters > 0, "mfma_peak16: bad arguments");
    hipLaunchKernelGGL(mfma_peak16_kernel, dim3(blocks), dim3(256), 0, to_stream(stream), scratch, iters, 0.5f);
    return check_launch("mfma_peak16_kernel");
}

extern "C" int vq2_debug_mfma_peak(float *scratch, int32_t blocks, int32_t iters, vq2_stream_t stream) {
    VQ2_REQUIRE(scratch && blocks > 0 && iters > 0, "mfma_peak: bad arguments");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, to_stream(stream), scratch, iters, 0.5f);
    return check_launch("mfma_peak_kernel");
}
