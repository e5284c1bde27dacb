// Sampler-side kernels: CCDM fused posterior+sample, DDIM update, layout movers, small fp32 linears.
#include "gg_common.h"
#include "gg_posterior.h"

// ------------------------------------------------------------------------------------------------------------
// CCDM reverse step, one voxel per thread (gg_posterior.h holds the arithmetic).
// ------------------------------------------------------------------------------------------------------------
template <int KMAX, int KS = 0>
__global__ __launch_bounds__(256) void ccdm_posterior_kernel(const float *__restrict__ head, int head_stride, int is_logits,
                                                             const int *__restrict__ xt, const float *__restrict__ E,
                                                             uint64_t seed, const long long *__restrict__ offset_dev, int draw,
                                                             const float *__restrict__ scalars, int K, long long M,
                                                             int *__restrict__ labels_out, float *__restrict__ probs_out,
                                                             bf16_t *__restrict__ onehot_out, int onehot_stride)
{
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float p0[KMAX];
    const float *hp = head + m * head_stride;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) p0[c] = (c < (KS > 0 ? KS : K)) ? hp[c] : 0.f;
    const long long off = (draw && !E && offset_dev) ? offset_dev[0] : 0;
    const int best = ccdm_posterior_voxel<KMAX, KS>(p0, is_logits, xt[m], scalars[0], scalars[1], K, m, draw, E ? E + m * K : nullptr, seed, off,
                                                probs_out ? probs_out + m * K : nullptr);
    labels_out[m] = best;
    if (onehot_out) ccdm_onehot_row<KMAX, KS>(onehot_out + m * onehot_stride, best, K);
}

extern "C" int gg_ccdm_posterior_sample(const float *head, int32_t head_stride, int32_t head_is_logits, const int32_t *xt,
                                        const float *E, uint64_t philox_seed, const int64_t *philox_offset_dev, int32_t draw,
                                        const float *scalars_dev, int32_t K, int64_t M, int32_t *labels_out, float *probs_out,
                                        void *onehot_out, int32_t onehot_stride, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!head || !xt || !scalars_dev || !labels_out) GG_FAIL(GG_ERR_BAD_SHAPE, "ccdm_posterior_sample: null pointer");
    if (K < 2 || K > 32) GG_FAIL(GG_ERR_UNSUPPORTED, "ccdm_posterior_sample: K=%d outside [2, 32]", K);
    if (head_stride < K) GG_FAIL(GG_ERR_BAD_SHAPE, "ccdm_posterior_sample: head_stride < K");
    if (onehot_out && onehot_stride < K) GG_FAIL(GG_ERR_BAD_SHAPE, "ccdm_posterior_sample: onehot_stride < K");
    if (onehot_out && ((onehot_stride & 1) || ((uintptr_t)onehot_out & 3))) GG_FAIL(GG_ERR_BAD_SHAPE, "ccdm_posterior_sample: onehot rows must be 4-byte aligned (even stride)");
    if (M <= 0) return GG_OK;
    dim3 grid((unsigned)((M + 255) / 256));
    if (K == 14)        // the class count of the shipped configs (13 organs + background), compiled in
        hipLaunchKernelGGL((ccdm_posterior_kernel<16, 14>), grid, dim3(256), 0, stream, head, head_stride, head_is_logits, xt, E,
                           philox_seed, (const long long *)philox_offset_dev, draw, scalars_dev, K, (long long)M, labels_out,
                           probs_out, (bf16_t *)onehot_out, onehot_stride);
    else if (K <= 16)
        hipLaunchKernelGGL(ccdm_posterior_kernel<16>, grid, dim3(256), 0, stream, head, head_stride, head_is_logits, xt, E,
                           philox_seed, (const long long *)philox_offset_dev, draw, scalars_dev, K, (long long)M, labels_out,
                           probs_out, (bf16_t *)onehot_out, onehot_stride);
    else
        hipLaunchKernelGGL(ccdm_posterior_kernel<32>, grid, dim3(256), 0, stream, head, head_stride, head_is_logits, xt, E,
                           philox_seed, (const long long *)philox_offset_dev, draw, scalars_dev, K, (long long)M, labels_out,
                           probs_out, (bf16_t *)onehot_out, onehot_stride);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

__global__ __launch_bounds__(256) void labels_to_onehot_kernel(const int *__restrict__ labels, long long M, int K,
                                                               bf16_t *__restrict__ out, int stride)
{
    // one thread per (voxel, 8-channel piece): writes the full padded row (zeros beyond K)
    const int P = stride >> 3;
    const long long total = M * P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long m = i / P;
        int c0 = (int)(i - m * P) * 8;
        int lab = labels[m];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((c0 + j == lab && c0 + j < K) ? 1.0f : 0.0f);
        *reinterpret_cast<bf16x8 *>(out + m * stride + c0) = o;
    }
}

extern "C" int gg_labels_to_onehot(const int32_t *labels, int64_t M, int32_t K, void *onehot_out, int32_t stride, void *stream_)
{
    if (!labels || !onehot_out) GG_FAIL(GG_ERR_BAD_SHAPE, "labels_to_onehot: null pointer");
    if (stride % 8 || stride < K) GG_FAIL(GG_ERR_BAD_SHAPE, "labels_to_onehot: stride %d (K=%d)", stride, K);
    if (M <= 0) return GG_OK;
    long long total = (long long)M * (stride / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(labels_to_onehot_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, labels, (long long)M, K,
                       (bf16_t *)onehot_out, stride);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// DDIM update (ddim.py:190-204), fp32; same expression order as oracle/samplers.py:ddim_step.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ddim_step_kernel(float *__restrict__ x, const float *__restrict__ eps, int eps_stride,
                                                        const float *__restrict__ noise, const float *__restrict__ sc,
                                                        long long M, int C, float *__restrict__ pred_x0_out,
                                                        bf16_t *__restrict__ unet_in, int unet_in_stride)
{
#pragma clang fp contract(off)
    const float a_t = sc[0], a_prev = sc[1], sigma = sc[2], s1m = sc[3];
    const float sqrt_at = sqrtf(a_t), sqrt_ap = sqrtf(a_prev), dirc = sqrtf(1.0f - a_prev - sigma * sigma);
    const long long total = M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long m = i / C;
        int c = (int)(i - m * C);
        float e = eps[m * eps_stride + c];
        float xv = x[i];
        float px0 = (xv - s1m * e) / sqrt_at;
        float xn = sqrt_ap * px0 + dirc * e;
        if (noise) xn = xn + sigma * noise[i];
        x[i] = xn;
        if (pred_x0_out) pred_x0_out[i] = px0;
        if (unet_in) unet_in[m * unet_in_stride + c] = (bf16_t)xn;
    }
}

extern "C" int gg_ddim_step(float *x, const float *eps, int32_t eps_stride, const float *noise, const float *scalars_dev,
                            int64_t M, int32_t C, float *pred_x0_out, void *unet_in, int32_t unet_in_stride, void *stream_)
{
    if (!x || !eps || !scalars_dev) GG_FAIL(GG_ERR_BAD_SHAPE, "ddim_step: null pointer");
    if (eps_stride < C || (unet_in && unet_in_stride < C)) GG_FAIL(GG_ERR_BAD_SHAPE, "ddim_step: stride < C");
    long long total = (long long)M * C;
    if (total <= 0) return GG_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, x, eps, eps_stride, noise,
                       scalars_dev, (long long)M, C, pred_x0_out, (bf16_t *)unet_in, unet_in_stride);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Ancestral DDPM step (LatentDiffusion.p_sample, ldm/models/diffusion/ddpm.py:217-230,1060-1120), fp32, same order:
//   x_recon = sqrt_recip_ac*x - sqrt_recipm1_ac*eps ; mean = coef1*x_recon + coef2*x ; x_prev = mean + sigma*noise
//   scalars device fp32[5] = {sqrt_recip_alphas_cumprod[t], sqrt_recipm1_alphas_cumprod[t], posterior_mean_coef1[t],
//                             posterior_mean_coef2[t], (t > 0) * exp(0.5*posterior_log_variance_clipped[t])}
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ddpm_step_kernel(float *__restrict__ x, const float *__restrict__ eps, int eps_stride,
                                                        const float *__restrict__ noise, const float *__restrict__ sc, long long M,
                                                        int C, bf16_t *__restrict__ unet_in, int unet_in_stride)
{
#pragma clang fp contract(off)
    const float a = sc[0], b = sc[1], c1 = sc[2], c2 = sc[3], sg = sc[4];
    const long long total = M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        const float xv = x[i];
        const float xr = a * xv - b * eps[m * eps_stride + c];
        float xn = c1 * xr + c2 * xv;
        if (noise) xn = xn + sg * noise[i];
        x[i] = xn;
        if (unet_in) unet_in[m * unet_in_stride + c] = (bf16_t)xn;
    }
}

extern "C" int gg_ddpm_step(float *x, const float *eps, int32_t eps_stride, const float *noise, const float *scalars_dev, int64_t M,
                            int32_t C, void *unet_in, int32_t unet_in_stride, void *stream_)
{
    if (!x || !eps || !scalars_dev) GG_FAIL(GG_ERR_BAD_SHAPE, "ddpm_step: null pointer");
    if (eps_stride < C || (unet_in && unet_in_stride < C)) GG_FAIL(GG_ERR_BAD_SHAPE, "ddpm_step: stride < C");
    long long total = (long long)M * C;
    if (total <= 0) return GG_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, x, eps, eps_stride, noise, scalars_dev,
                       (long long)M, C, (bf16_t *)unet_in, unet_in_stride);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// PLMS noise-estimate combination (ldm/models/diffusion/plms.py:218-232), fp32, same left-to-right order as the reference
// expression:  out = (c0*e0 + c1*e1 + c2*e2 + c3*e3) / denom   (terms with a NULL pointer are skipped)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lincomb4_kernel(const float *__restrict__ e0, const float *__restrict__ e1,
                                                       const float *__restrict__ e2, const float *__restrict__ e3, float c0, float c1,
                                                       float c2, float c3, float denom, long long n, float *__restrict__ out)
{
#pragma clang fp contract(off)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float a = c0 * e0[i];
        if (e1) a = a + c1 * e1[i];
        if (e2) a = a + c2 * e2[i];
        if (e3) a = a + c3 * e3[i];
        out[i] = a / denom;
    }
}

extern "C" int gg_lincomb4(const float *e0, const float *e1, const float *e2, const float *e3, float c0, float c1, float c2, float c3,
                           float denom, int64_t n, float *out, void *stream_)
{
    if (!e0 || !out) GG_FAIL(GG_ERR_BAD_SHAPE, "lincomb4: null pointer");
    if (n <= 0) return GG_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(lincomb4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, e0, e1, e2, e3, c0, c1, c2, c3, denom,
                       (long long)n, out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// min-max normalisation over a whole tensor (ordered-uint atomics; deterministic)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

__global__ void minmax_init_kernel(uint32_t *ws) { ws[0] = 0xFFFFFFFFu; ws[1] = 0u; }

__global__ __launch_bounds__(256) void minmax_reduce_kernel(const float *__restrict__ src, long long n, uint32_t *ws)
{
    float mn = INFINITY, mx = -INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = src[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&ws[0], f2ord(mn)); atomicMax(&ws[1], f2ord(mx)); }
}

__global__ __launch_bounds__(256) void minmax_apply_kernel(const float *__restrict__ src, long long n, const uint32_t *ws,
                                                           float *__restrict__ dst)
{
#pragma clang fp contract(off)
    const float mn = ord2f(ws[0]), mx = ord2f(ws[1]);
    const float den = mx - mn;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (src[i] - mn) / den;
}

extern "C" int gg_minmax_normalise(const float *src, int64_t n, float *dst, float *workspace2, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst || !workspace2) GG_FAIL(GG_ERR_BAD_SHAPE, "minmax_normalise: null pointer");
    if (n <= 0) return GG_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, stream, (uint32_t *)workspace2);
    hipLaunchKernelGGL(minmax_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, (long long)n, (uint32_t *)workspace2);
    hipLaunchKernelGGL(minmax_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, (long long)n, (const uint32_t *)workspace2, dst);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Stage glue: CCDM label volume -> conditioning slice of the LDM loop, on the device.
//   up[d,y,x]  = labels[n, floor(d*Dm/D), floor(y*Hm/H), floor(x*Wm/W)]      nearest (order-0) upsample
//   rot[i,j]   = up[slice, H-1-j, i]                                          torch.rot90(k=3) on (H, W)
//   cond[n,i,j,0] = prev[n,i,j] (previous generated slice, [0,1]) ; cond[n,i,j,1] = rot[i,j]/255 ; other lanes 0
// (latentdiffusion/sample_diffusion.py:199-210; value convention ldm/data/ruijin_pimage_and_mask.py:127-131)
// ------------------------------------------------------------------------------------------------------------
// Index rule of scipy.ndimage.zoom(order=0) with its defaults (mode="constant", grid_mode=False), which is what the reference's
// stage-glue recipe calls (latentdiffusion/sample_diffusion.py:199-200): output index o reads input index
//   floor(o * zf + 0.5),  zf = (in - 1) / (out - 1)  in IEEE double  (NI_ZoomShift: cc = o * zoom; start = floor(cc + 0.5)).
// It differs from F.interpolate(nearest)'s floor(o * in / out) on 16 % of the indices at 128 -> 512.  The product and the sum are
// rounded separately (no FMA contraction), as the C code of scipy does, so that the index is bit-identical.
__device__ __forceinline__ int zoom0_index(int o, double zf, int n_in)
{
    const int i = (int)floor(__dadd_rn(__dmul_rn((double)o, zf), 0.5));
    return i < 0 ? 0 : (i > n_in - 1 ? n_in - 1 : i);
}

__global__ __launch_bounds__(256) void mask_to_cond_slice_kernel(const int *__restrict__ labels, int N, int Dm, int Hm, int Wm,
                                                                 int slice, int D, int H, int W, double zd, double zh, double zw,
                                                                 const float *__restrict__ prev, bf16_t *__restrict__ cond, int stride,
                                                                 float *__restrict__ mask_out)
{
    const long long total = (long long)N * H * W;
    const int sd = zoom0_index(slice, zd, Dm);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        int j = (int)(t % W);
        int i = (int)((t / W) % H);
        int n = (int)(t / ((long long)W * H));
        int y = H - 1 - j, x = i;                         // rot90(k=3) on (H, W): out[i][j] = up[H-1-j][i]  (H == W)
        int sy = zoom0_index(y, zh, Hm), sx = zoom0_index(x, zw, Wm);
        int lab = labels[(((long long)n * Dm + sd) * Hm + sy) * Wm + sx];
        float mv = (float)lab / 255.0f;
        float pv = prev ? prev[t] : 0.f;
        bf16_t *row = cond + t * stride;
        bf16x8 o = {(bf16_t)pv, (bf16_t)mv, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        *reinterpret_cast<bf16x8 *>(row) = o;
        bf16x8 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        for (int c = 8; c < stride; c += 8) *reinterpret_cast<bf16x8 *>(row + c) = z;
        if (mask_out) mask_out[t] = mv;
    }
}

static double zoom0_factor(int n_in, int n_out) { return n_out > 1 ? (double)(n_in - 1) / (double)(n_out - 1) : 1.0; }

extern "C" int gg_mask_to_cond_slice(const int32_t *labels, int32_t N, int32_t Dm, int32_t Hm, int32_t Wm, int32_t slice, int32_t D,
                                     int32_t H, int32_t W, const float *prev, void *cond_cl, int32_t stride, float *mask_out,
                                     void *stream_)
{
    if (!labels || !cond_cl) GG_FAIL(GG_ERR_BAD_SHAPE, "mask_to_cond_slice: null pointer");
    if (H != W) GG_FAIL(GG_ERR_UNSUPPORTED, "mask_to_cond_slice: rot90 needs H == W");
    if (stride % 8 || stride < 8) GG_FAIL(GG_ERR_BAD_SHAPE, "mask_to_cond_slice: stride");
    if (slice < 0 || slice >= D) GG_FAIL(GG_ERR_BAD_SHAPE, "mask_to_cond_slice: slice %d outside [0,%d)", slice, D);
    if (N < 1 || Dm < 1 || Hm < 1 || Wm < 1) GG_FAIL(GG_ERR_BAD_SHAPE, "mask_to_cond_slice: empty label volume");
    long long total = (long long)N * H * W;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_to_cond_slice_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, labels, N, Dm, Hm, Wm, slice,
                       D, H, W, zoom0_factor(Dm, D), zoom0_factor(Hm, H), zoom0_factor(Wm, W), prev, (bf16_t *)cond_cl, stride, mask_out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// layout movers
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_cl_kernel(const float *__restrict__ src, int N, int C, long long S,
                                                         bf16_t *__restrict__ dst, int C_pad, int c_off, int zero_fill)
{
    // thread per (n, s, 8-channel piece of the padded row)
    const int P = C_pad >> 3;
    const long long total = (long long)N * S * P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // order: piece slowest within (n), s fastest -> coalesced fp32 reads along s
        long long ns = i % ((long long)N * S);
        int piece = (int)(i / ((long long)N * S));
        long long n = ns / S, s = ns - n * S;
        int c0 = piece * 8;
        bool any = false;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = c0 + j - c_off;
            float v = 0.f;
            if (c >= 0 && c < C) { v = src[(n * C + c) * S + s]; any = true; }
            o[j] = (bf16_t)v;
        }
        bf16_t *d = dst + (n * S + s) * C_pad + c0;
        if (zero_fill) {
            *reinterpret_cast<bf16x8 *>(d) = o;
        } else if (any) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int c = c0 + j - c_off;
                if (c >= 0 && c < C) d[j] = o[j];
            }
        }
    }
}

extern "C" int gg_nchw_f32_to_cl_bf16(const float *src, int32_t N, int32_t C, int64_t S, void *dst, int32_t C_pad,
                                      int32_t c_offset, int32_t zero_fill, void *stream_)
{
    if (!src || !dst) GG_FAIL(GG_ERR_BAD_SHAPE, "nchw_to_cl: null pointer");
    if (C_pad % 8 || c_offset < 0 || c_offset + C > C_pad) GG_FAIL(GG_ERR_BAD_SHAPE, "nchw_to_cl: C=%d offset=%d C_pad=%d", C, c_offset, C_pad);
    long long total = (long long)N * S * (C_pad / 8);
    if (total <= 0) return GG_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(nchw_to_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, src, N, C, (long long)S,
                       (bf16_t *)dst, C_pad, c_offset, zero_fill);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void cl_to_nchw_kernel(const T *__restrict__ src, int N, int C, long long S, int C_pad,
                                                         float *__restrict__ dst)
{
    const long long total = (long long)N * C * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long s = i % S;
        long long nc = i / S;
        int c = (int)(nc % C);
        long long n = nc / C;
        dst[i] = (float)src[(n * S + s) * C_pad + c];
    }
}

extern "C" int gg_cl_to_nchw_f32(const void *src, int32_t src_dtype, int32_t N, int32_t C, int64_t S, int32_t C_pad, float *dst,
                                 void *stream_)
{
    if (!src || !dst) GG_FAIL(GG_ERR_BAD_SHAPE, "cl_to_nchw: null pointer");
    if (C > C_pad) GG_FAIL(GG_ERR_BAD_SHAPE, "cl_to_nchw: C > C_pad");
    long long total = (long long)N * C * S;
    if (total <= 0) return GG_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (src_dtype == GG_BF16)
        hipLaunchKernelGGL(cl_to_nchw_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)src, N, C,
                           (long long)S, C_pad, dst);
    else if (src_dtype == GG_F32)
        hipLaunchKernelGGL(cl_to_nchw_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const float *)src, N, C,
                           (long long)S, C_pad, dst);
    else
        GG_FAIL(GG_ERR_BAD_DTYPE, "cl_to_nchw: dtype");
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// small fp32 linear (one wave per output element) and sinusoidal embedding
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_f32_kernel(const float *__restrict__ in, int M, int I, const float *__restrict__ W,
                                                         const float *__restrict__ b, int O, int act_in, float *__restrict__ out,
                                                         long long out_stride)
{
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= (long long)M * O) return;
    const int m = (int)(w / O), o = (int)(w - (long long)m * O);
    float acc = 0.f;
    for (int i = lane; i < I; i += 64) {
        float v = in[(long long)m * I + i];
        if (act_in) v = v / (1.0f + expf(-v));
        acc += v * W[(long long)o * I + i];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if (lane == 0) out[(long long)m * out_stride + o] = acc + (b ? b[o] : 0.f);
}

extern "C" int gg_linear_f32(const float *in, int32_t M, int32_t I, const float *W, const float *b, int32_t O, int32_t act_in,
                             float *out, int64_t out_stride, void *stream_)
{
    if (!in || !W || !out) GG_FAIL(GG_ERR_BAD_SHAPE, "linear_f32: null pointer");
    if (M <= 0 || I <= 0 || O <= 0 || out_stride < O) GG_FAIL(GG_ERR_BAD_SHAPE, "linear_f32: bad shape");
    long long waves = (long long)M * O;
    hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, in, M, I, W, b, O, act_in,
                       out, (long long)out_stride);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

__global__ void timestep_embedding_kernel(const float *__restrict__ t, int M, int dim, float max_period, float *__restrict__ out)
{
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * dim) return;
    const int m = i / dim, j = i - m * dim;
    float v = 0.f;
    if (j < 2 * half) {
        int f = (j < half) ? j : j - half;
        float freq = expf(-logf(max_period) * (float)f / (float)half);
        float arg = t[m] * freq;
        v = (j < half) ? cosf(arg) : sinf(arg);
    }
    out[i] = v;
}

extern "C" int gg_timestep_embedding(const float *t, int32_t M, int32_t dim, float max_period, float *out, void *stream_)
{
    if (!t || !out || M <= 0 || dim <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "timestep_embedding: bad args");
    int total = M * dim;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, t, M, dim, max_period, out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void gg_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *gg_last_error(void) { return g_err; }
extern "C" int gg_version(void) { return 100; }
