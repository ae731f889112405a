/*
 * vq2.h -- C ABI of libvq2.so: the MI355X (gfx950) implementation of the
 * VQ-VAE-2 stage-1 hot path of alehdaghi/vq-vae-2-pytorch.
 *
 * The reference has NO native/FFI boundary on this path: every op
 * in vqvae.py is an ATen call made from Python.  Each entry point below therefore
 * cites the reference *Python call site* (file:line under /root/reference) whose
 * ATen kernel it replaces.  The host-side mirror of the reference's nn.Module
 * interface lives in vq-vae-2-pytorch_amd/vqvae.py and calls these through ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 unless stated; the caller owns all
 *    memory (tensors and workspaces); the library allocates nothing persistent.
 *  - activations are NHWC: element (n,h,w,c) at ((n*H+h)*W+w)*ld + c where the
 *    pixel stride ld >= C lets an op read/write a channel slice of a wider buffer
 *    (that is how torch.cat at vqvae.py:233 and :218 disappears).
 *    Channel counts and pixel strides must be multiples of 4 (16-byte vectors).
 *  - `stream` is a hipStream_t (NULL = default stream).  All work is enqueued on
 *    it; no entry point synchronises the device.
 *  - return value: VQ2_OK or an error class; vq2_last_error() returns the
 *    message of the calling thread's last failure (thread-local, re-entrant:
 *    forward runs on the main thread, backward on an autograd thread).
 */
#ifndef VQ2_H
#define VQ2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQ2_OK 0
#define VQ2_ERR_INVALID 1     /* bad argument (shape, alignment, null pointer)  */
#define VQ2_ERR_UNSUPPORTED 2 /* valid but not implemented configuration        */
#define VQ2_ERR_WORKSPACE 3   /* workspace too small                            */
#define VQ2_ERR_LAUNCH 4      /* hipLaunch / runtime error                      */

typedef void *vq2_stream_t;

/* ABI revision of THIS header.  It moves whenever an entry point changes its argument list or the meaning of an
 * argument / workspace (revision 2: vq2_vq_fwd lost its counts/sumsT arguments and vq2_vq_fwd_workspace_floats
 * went from (M) to (M, D, K); revision 3: round-3 additions, see INTEGRATION.md "ABI history").  vq2_version()
 * returns the revision the LIBRARY was built from: a host must refuse to run when the two differ (a mismatched
 * workspace size would let a kernel write past the caller's buffer). */
#define VQ2_API_VERSION 3

int vq2_version(void);
const char *vq2_last_error(void);

/* Measurement aid (bench.py): level 1 brackets every conv/wgrad/VQ launch with two HIP events on its
 * own stream, level 2 only the dominant kernel (the 128x128x32 conv tile) so that the timed region is
 * barely perturbed, 0 = off.  vq2_prof_report waits for the events (the ONLY synchronising entry
 * point), writes one line per kernel+shape "name launches total_ms algorithmic_flops
 * algorithmic_bytes" and clears the records. */
int vq2_prof_enable(int level);
int vq2_prof_report(char *buf, size_t cap);

/* ------------------------------------------------------------------ conv
 * One descriptor describes the FORWARD op; fwd/dgrad/wgrad entry points all
 * take the same descriptor so callers never swap roles by hand.
 *   conv  (transposed=0): y[N,Ho,Wo,Co] = conv2d(x[N,H,W,Ci], w[Co,Ci,KH,KW], stride, pad)
 *                         Ho = (H+2*pad-KH)/stride+1            (vqvae.py:87,89,105-116,137,184,189)
 *   convT (transposed=1): y[N,2H,2W,Co] = conv_transpose2d(x, w[Ci,Co,4,4], stride 2, pad 1)
 *                                                               (vqvae.py:150-160,191-193)
 */
typedef struct vq2_conv_desc {
    int32_t N, H, W, Ci; /* input, NHWC                                         */
    int32_t Co;          /* output channels                                     */
    int32_t KH, KW, stride, pad;
    int32_t transposed;  /* 0 conv, 1 conv-transpose (KH=KW=4, stride 2, pad 1) */
    int32_t ldx, ldy;    /* pixel strides of x and y buffers (elements)         */
    int32_t Cir, Cor;    /* channel counts of the reference weight/bias tensors (<= Ci, Co; 0 = same).
                            Ci/Co are rounded up to a multiple of 4 (the 3-channel image and
                            reconstruction, vqvae.py:105,157); padded channels read as weight 0,
                            are written as 0 and are skipped in dw/db. */
} vq2_conv_desc;

/* epilogue / prologue fusion flags */
#define VQ2_RELU_IN 1  /* operand is relu(x): ReLU fused into the load (vqvae.py:86,88,107,...) */
#define VQ2_RELU_OUT 2 /* y = relu(...): trailing in-place ReLU (vqvae.py:122,144)            */

/* weight packing: reference layouts (OIHW for Conv2d, IOHW for ConvTranspose2d,
 * i.e. the state_dict tensors as they are) -> kernel layouts.  Every packed
 * buffer has exactly w.numel() floats. */
#define VQ2_PACK_FWD 0   /* operand of vq2_conv_fwd   */
#define VQ2_PACK_DGRAD 1 /* operand of vq2_conv_dgrad */
int vq2_pack_weight(const vq2_conv_desc *d, int which, const float *w, float *packed, vq2_stream_t stream);

/* Re-packing every layer after an optimizer step in ONE launch: fill one job per (layer, which) on
 * the host with vq2_pack_job_init, set job.offset to the running sum of job.numel, copy the array to
 * the device once, then call vq2_pack_weights_batched(jobs_dev, njobs, sum of numel) every step. */
typedef struct vq2_pack_job {
    const float *w;  /* reference-layout weight (device)            */
    float *packed;   /* destination panel (device), numel floats    */
    int64_t offset;  /* start of this job in the batched index space */
    int64_t numel;
    int32_t Or, Ir, Op, Ip, KH, KW, mode, reserved;
} vq2_pack_job;
int vq2_pack_job_init(const vq2_conv_desc *d, int which, const float *w, float *packed, vq2_pack_job *job);
int vq2_pack_weights_batched(const vq2_pack_job *jobs_dev, int32_t njobs, int64_t total, vq2_stream_t stream);

/* y = [relu]( conv_or_convT([relu]x, w) + bias [+ residual] )
 * wp: VQ2_PACK_FWD packing of w.  bias may be NULL.  residual (same shape as y,
 * pixel stride ldres) may be NULL: the `out += input` of vqvae.py:94. */
int vq2_conv_fwd(const vq2_conv_desc *d, int flags, const float *x, const float *wp, const float *bias,
                 const float *residual, int32_t ldres, float *y, vq2_stream_t stream);

/* Fused ResBlock forward (vqvae.py:81-96), one launch:
 *     r = relu(conv3x3(relu(x)) + b1)        [N,H,W,Cm]  pixel stride ldr  (saved for the backward pass)
 *     y = [relu]( conv1x1(r) + b2 + x )      [N,H,W,C]   pixel stride ldy
 * w1p / w2p: VQ2_PACK_FWD panels of the 3x3 (Cm x C) and 1x1 (C x Cm) weights.  flags: VQ2_RELU_OUT only
 * (the trailing ReLU of Encoder/Decoder, vqvae.py:122,144).  Built for C = 128, Cm = 32 (the reference's
 * channel / n_res_channel defaults): vq2_resblock_supported() says whether a (C, Cm) pair can take this
 * path; otherwise the caller composes the block from two vq2_conv_fwd launches. */
int vq2_resblock_supported(int32_t C, int32_t Cm);
int vq2_resblock_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm, int flags, const float *x, int32_t ldx,
                     const float *w1p, const float *b1, const float *w2p, const float *b2, float *r, int32_t ldr,
                     float *y, int32_t ldy, vq2_stream_t stream);

/* Fused ResBlock backward, data path (one launch instead of two dgrad launches):
 *     dh = (r > 0) * dgrad_1x1(g)                    [N,H,W,Cm]  (output: both weight gradients read it)
 *     dx = (x > 0) * dgrad_3x3(dh) + g               [N,H,W,C]
 * g: gradient of the block output (already masked if the block had VQ2_RELU_OUT); r, x: saved by
 * vq2_resblock_fwd; w2d / w1d: VQ2_PACK_DGRAD panels of the 1x1 and 3x3 weights.  Same (C, Cm) support
 * as vq2_resblock_fwd. *
 * w2_ws (optional, vq2_resblock_w2_workspace_bytes): the kernel also leaves, per workgroup, the partial 1x1
 * weight and bias gradients of its own pixels there (it holds g and r anyway), which saves the separate
 * vq2_conv_wgrad launch of the 1x1 conv: fill a vq2_wgrad_job with vq2_resblock_w2_job_init and hand it to
 * vq2_wgrad_reduce_batched together with the other layers' jobs. */
size_t vq2_resblock_w2_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm);
int vq2_resblock_bwd_data(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm, const float *g, int32_t ldg,
                          const float *r, int32_t ldr, const float *x, int32_t ldx, const float *w2d, const float *w1d,
                          float *dh, int32_t lddh, float *dx, int32_t lddx, void *w2_ws, vq2_stream_t stream);

/* dx = dgrad(dy) [* (mask > 0)] [+ residual]
 * wp: VQ2_PACK_DGRAD packing of w.  mask (shape of x, pixel stride ldmask): the
 * pre-ReLU input when the forward op had VQ2_RELU_IN (ReLU backward fused);
 * residual (shape of x): the skip-path gradient of a ResBlock.  dx has pixel
 * stride lddx; dy has pixel stride d->ldy. */
int vq2_conv_dgrad(const vq2_conv_desc *d, const float *dy, const float *wp, const float *mask, int32_t ldmask,
                   const float *residual, int32_t ldres, float *dx, int32_t lddx, vq2_stream_t stream);
/* Same with flags.  VQ2_MASK_AFTER_RESIDUAL: dx = (dgrad(dy) + residual) * (mask > 0) -- x is itself the
 * output of a fused trailing ReLU (vqvae.py:122,144) whose backward mask is applied to the COMPLETE
 * gradient here (residual = the gradient of x's other consumer), instead of by a vq2_relu_bwd pass. */
#define VQ2_MASK_AFTER_RESIDUAL 4
int vq2_conv_dgrad_ex(const vq2_conv_desc *d, int flags, const float *dy, const float *wp, const float *mask,
                      int32_t ldmask, const float *residual, int32_t ldres, float *dx, int32_t lddx,
                      vq2_stream_t stream);

/* dw (reference layout, OIHW or IOHW) = wgrad([relu]x, dy) and, if db != NULL, db[Cor] = sum over
 * pixels of dy (bias gradient, fused).  Deterministic split-K: partial slabs in `ws`, then an
 * ordered reduction.  flags: VQ2_RELU_IN. */
size_t vq2_conv_wgrad_workspace_bytes(const vq2_conv_desc *d);
int vq2_conv_wgrad(const vq2_conv_desc *d, int flags, const float *x, const float *dy, float *dw, float *db, void *ws,
                   size_t ws_bytes, vq2_stream_t stream);

/* Deferred form for a whole backward pass: vq2_conv_wgrad_partial writes only the slabs and the bias
 * partials (db != NULL asks for them; db itself is written by the reduction), one
 * vq2_wgrad_reduce_batched launch at the end reduces EVERY layer.  Fill one
 * job per layer with vq2_wgrad_job_init (ws must stay alive and private to the layer until the batched
 * launch), set unit_offset to the running sum of n_units_w + n_units_b, upload the array once. */
typedef struct vq2_wgrad_job {
    const float *ws;      /* the layer's slab workspace                         */
    float *dw;            /* destination, reference layout                      */
    const float *bias_ws; /* bias partials inside ws or NULL                    */
    float *db;            /* bias gradient destination or NULL                  */
    int64_t unit_offset;  /* start of this job in the batched unit space        */
    int32_t O, I, Or, Ir, taps, S, n_units_w, n_units_b;
    int32_t swapped;      /* bit 0: slab is [ci][flipped tap][co] (roles of x and dy exchanged);
                             bits 1-2: Winograd slab layout written by vq2_conv_wgrad_partial (0 none, 1 F(2,3) over
                             column pairs, 2 F(2,2) by column parity, 3 F(2,3) with exchanged roles) -- filled by
                             vq2_wgrad_job_init, interpreted by vq2_wgrad_reduce_batched                        */
    int32_t bias_splits;  /* > 0: bias partials are [bias_splits][I] (taken from the gathered dy
                             operand: exchanged roles, conv-transpose); 0: [S][O]                  */
} vq2_wgrad_job;
int vq2_conv_wgrad_partial(const vq2_conv_desc *d, int flags, const float *x, const float *dy, float *db, void *ws,
                           size_t ws_bytes, vq2_stream_t stream);
int vq2_wgrad_job_init(const vq2_conv_desc *d, const void *ws, float *dw, float *db, vq2_wgrad_job *job);
int vq2_wgrad_reduce_batched(const vq2_wgrad_job *jobs_dev, int32_t njobs, int64_t total_units, vq2_stream_t stream);
/* job of the 1x1 weight gradient that vq2_resblock_bwd_data left in `ws` (see there) */
int vq2_resblock_w2_job_init(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cm, const void *ws, float *dw, float *db,
                             vq2_wgrad_job *job);

/* out[c] = sum over rows of x[.,c] (stand-alone column sums).  ws: >= vq2_colsum_workspace_bytes. */
size_t vq2_colsum_workspace_bytes(int64_t rows, int32_t C);
int vq2_colsum(const float *dy, int64_t rows, int32_t C, int32_t ld, float *db, void *ws, size_t ws_bytes,
               vq2_stream_t stream);

/* ------------------------------------------------------------------ layout + elementwise */
/* NCHW [N,C,H,W] -> NHWC with pixel stride ld (>= C); channels C..ld-1 are written as 0 */
int vq2_nchw_to_nhwc(const float *src, float *dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ld,
                     vq2_stream_t stream);
int vq2_nhwc_to_nchw(const float *src, float *dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ld,
                     vq2_stream_t stream);
/* g = dy * (y > 0) over [pixels, C] with pixel strides: backward of a fused VQ2_RELU_OUT
 * (vqvae.py:122,144); with dy == y it is the forward ReLU itself */
int vq2_relu_bwd(const float *dy, int32_t lddy, const float *y, int32_t ldy, float *g, int32_t ldg, int64_t pixels,
                 int32_t C, vq2_stream_t stream);
/* dst[p*ldd + c] (+)= src[p*lds + c], c < C: channel-slice copy / accumulate */
int vq2_slice_copy(const float *src, int32_t lds, float *dst, int32_t ldd, int64_t pixels, int32_t C,
                   int accumulate, vq2_stream_t stream);

/* ------------------------------------------------------------------ Quantize (vqvae.py:28-78)
 * x[M,D] rows (NHWC latents), embed[D,K] (reference layout).
 * vq2_vq_prepare: embedT[K,D] and enorm[K] = sum_d embed[d,k]^2      (vqvae.py:47)
 * vq2_vq_fwd:  idx[M] (int64) = first argmin_k ||x||^2 - 2 x.e_k + ||e_k||^2 (vqvae.py:44-49)
 *              out[M,D] = x + (e_idx - x)                            (vqvae.py:52,73)
 *              ws (>= vq2_vq_fwd_workspace_floats(M, D, K) floats): per-128-vector sums of (e_idx - x)^2
 *              (vqvae.py:72) at its start -- the operand of vq2_vq_loss -- followed by the (distance, index)
 *              candidates of a K-split search (few vectors x large codebook: splits searched by separate
 *              workgroups, first minimum taken in code order: same indices as one pass)
 * vq2_vq_stats: counts[K] = one-hot sum, sumsT[K,D] = x rows summed per code   (vqvae.py:55-56)
 *              every element is WRITTEN (no zeroing by the caller) and the result is bit-reproducible:
 *              a stable counting sort of the row numbers by code, then each code's rows are added in
 *              increasing row order along a fixed tree -- no float atomics.  ws >= vq2_vq_stats_workspace_bytes.
 * vq2_vq_loss: diff = sum(loss_partial) / (M*D)
 * vq2_vq_bwd:  dx = g_out + (2/(M*D)) * g_diff * (x - e_idx)         (autograd of vqvae.py:72-73)
 * vq2_vq_ema_update: in-place EMA of cluster_size/embed_avg/embed from (all-reduced) counts/sumsT
 *                                                                   (vqvae.py:61-70)
 * vq2_vq_gather: out[M,D] = embedT[idx]                              (vqvae.py:77-78 embed_code)
 */
int vq2_vq_prepare(const float *embed, float *embedT, float *enorm, int32_t D, int32_t K, vq2_stream_t stream);
size_t vq2_vq_fwd_workspace_floats(int64_t M, int32_t D, int32_t K);
int vq2_vq_fwd(const float *x, int32_t ldx, const float *embed, const float *embedT, const float *enorm, int64_t M,
               int32_t D, int32_t K, int64_t *idx, float *out, int32_t ldo, float *ws, vq2_stream_t stream);
size_t vq2_vq_stats_workspace_bytes(int64_t M, int32_t D, int32_t K);
int vq2_vq_stats(const float *x, int32_t ldx, const int64_t *idx, int64_t M, int32_t D, int32_t K, float *counts,
                 float *sumsT, void *ws, size_t ws_bytes, vq2_stream_t stream);
int vq2_vq_loss(const float *loss_partial, int64_t M, int32_t D, float *diff, vq2_stream_t stream);
int vq2_vq_bwd(const float *g_out, int32_t ldg, const float *g_diff, const float *x, int32_t ldx,
               const int64_t *idx, const float *embedT, int64_t M, int32_t D, int32_t K, float *dx, int32_t lddx,
               vq2_stream_t stream);
int vq2_vq_ema_update(float *embed, float *cluster_size, float *embed_avg, const float *counts, const float *sumsT,
                      int32_t D, int32_t K, double decay, double eps, float *scratch /* >= 1 float */,
                      vq2_stream_t stream);
/* the same update, also leaving embedT / enorm of the UPDATED codebook (what vq2_vq_prepare would compute before
 * the next forward, bit for bit): one launch less per quantizer and step.  D must divide 256. */
int vq2_vq_ema_update_prepare(float *embed, float *cluster_size, float *embed_avg, const float *counts,
                              const float *sumsT, int32_t D, int32_t K, double decay, double eps, float *scratch,
                              float *embedT, float *enorm, vq2_stream_t stream);
int vq2_vq_gather(const int64_t *idx, const float *embedT, int64_t M, int32_t D, int32_t K, float *out, int32_t ldo,
                  vq2_stream_t stream);

/* ------------------------------------------------------------------ AdaIN (VQVAE_Deep decoder, vqvae_deep.py:99-134)
 * vq2_instnorm_stats: mean / rstd [N,C] of nn.InstanceNorm2d(C, affine=False) over the HW pixels of each image
 *                     (biased variance, rstd = 1/sqrt(var + eps))                            (vqvae_deep.py:102)
 * vq2_adain_fwd:      y = [relu]((1 + gamma) * (x - mean) * rstd + beta), h = [N, 2C] = (gamma | beta) = fc(style)
 *                     (vqvae_deep.py:105-109; flags VQ2_RELU_OUT fuses the F.relu_ of vqvae_deep.py:129,131)
 * vq2_adain_bwd:      dz = dy * (y > 0) (y NULL: no ReLU);  dh[N,2C] = (sum_p dz*xhat | sum_p dz);
 *                     dx = rstd * (1 + gamma) * (dz - mean_p dz - xhat * mean_p(dz * xhat))
 * Fixed-order reductions (bit-reproducible).  The fc itself is a 1x1 conv on a [N,1,1,style_dim] tensor. */
int vq2_instnorm_stats(const float *x, int32_t ldx, int32_t N, int64_t HW, int32_t C, double eps, float *mean,
                       float *rstd, vq2_stream_t stream);
int vq2_adain_fwd(const float *x, int32_t ldx, const float *mean, const float *rstd, const float *h, int32_t N,
                  int64_t HW, int32_t C, int flags, float *y, int32_t ldy, vq2_stream_t stream);
int vq2_adain_bwd(const float *dy, int32_t lddy, const float *y, int32_t ldy, const float *x, int32_t ldx,
                  const float *mean, const float *rstd, const float *h, int32_t N, int64_t HW, int32_t C, float *dh,
                  float *dx, int32_t lddx, vq2_stream_t stream);

/* ------------------------------------------------------------------ loss + optimizer
 * vq2_mse_fwd_bwd: loss = sum((a-b)^2)/denom (train_vqvae.py:31,83) over `numel` contiguous
 *   elements (denom = numel, or the unpadded count when both operands carry zero padding);
 *   grad (may be NULL) = (*gscale) * 2*(a-b)/denom  (gscale NULL = 1).  ws >= vq2_mse_workspace_bytes.
 * vq2_adam_step: torch.optim.Adam (betas, eps, no weight decay; train_vqvae.py:185) over a
 *   flat fp32 arena; `step` is the 1-based step count. */
size_t vq2_mse_workspace_bytes(int64_t numel);
int vq2_mse_fwd_bwd(const float *a, const float *b, int64_t numel, int64_t denom, const float *gscale, float *loss,
                    float *grad, void *ws, size_t ws_bytes, vq2_stream_t stream);
/* the whole stage-1 loss in the same two launches: recon = MSE(a, b), total = recon + weight * latent[0]
 * (train_vqvae.py:83-85), grad = 2*(a-b)/denom */
int vq2_stage1_loss(const float *a, const float *b, int64_t numel, int64_t denom, const float *latent, float weight,
                    float *recon, float *total, float *grad, void *ws, size_t ws_bytes, vq2_stream_t stream);
int vq2_adam_step(float *p, const float *g, float *m, float *v, int64_t n, double lr, double beta1, double beta2,
                  double eps, int32_t step, double grad_scale, vq2_stream_t stream);
/* dst = a + alpha * b (flat): the 0.25 * latent_loss accumulation and small host-free scalar math */
int vq2_axpby(const float *a, const float *b, float alpha, float *dst, int64_t n, vq2_stream_t stream);
/* dst = src * scalar[0] * alpha, scalar read on the device (upstream gradient of a loss) */
int vq2_scale(const float *src, const float *scalar, float alpha, float *dst, int64_t n, vq2_stream_t stream);

/* ------------------------------------------------------------------ data-parallel exchange (RCCL over xGMI)
 * One communicator per process (= per GPU), owned by the library -- its only persistent state.  Replaces what
 * the reference moves with dist.all_reduce at vqvae.py:58-59 (through distributed/distributed.py:64-72: the EMA
 * sums, SUM) and with DistributedDataParallel at train_vqvae.py:166-171 (gradient all-reduce; initial broadcast
 * of rank 0's parameters and buffers); bring-up replaces distributed/launch.py:60-66.
 *   rank 0: vq2_comm_unique_id(id) -> ship the VQ2_COMM_ID_BYTES bytes to every rank (any side channel) ->
 *   every rank, after selecting its device: vq2_comm_init(id, rank, world).
 * Collectives are enqueued on `stream` (the caller orders them against compute with events); in place, fp32.
 * RCCL itself is loaded on first use, so a single-GPU process never needs it. */
#define VQ2_COMM_ID_BYTES 128
int vq2_comm_unique_id(void *id);
int vq2_comm_init(const void *id, int32_t rank, int32_t world);
int vq2_comm_world(void); /* 0 = no communicator */
int vq2_comm_rank(void);  /* -1 = no communicator */
int vq2_comm_allreduce_sum(float *buf, int64_t count, vq2_stream_t stream);
int vq2_comm_broadcast(float *buf, int64_t count, int32_t root, vq2_stream_t stream);
int vq2_comm_destroy(void);

/* calibration only: register-resident fp32-MFMA loop (blocks x 256 threads, iters x 32 MFMAs per wave);
 * 2*32*32*2 FLOP per MFMA.  Used by scripts/mfma_peak.py to measure the ceiling the chip sustains. */
int vq2_debug_mfma_peak(float *scratch, int32_t blocks, int32_t iters, vq2_stream_t stream);
/* same through v_mfma_f32_16x16x4_f32: iters x 64 MFMAs per wave, 2*16*16*4 FLOP each */
int vq2_debug_mfma_peak16(float *scratch, int32_t blocks, int32_t iters, vq2_stream_t stream);
/* diagnostic only: per-phase cycle stamps of the 128x128x32 conv tile into buf[16] (NULL = off) */
int vq2_debug_set_rb_stamps(unsigned long long *buf); /* same for the fused ResBlock backward kernel: buf[64] */
int vq2_debug_set_stamps(unsigned long long *buf);

#ifdef __cplusplus
}
#endif
#endif /* VQ2_H */
