/* guidegen_hip.h -- C-ABI of libguidegen_hip.so (MI355X / gfx950 only).
 *
 * The reference (OvO1111/JointImageGeneration) has no native boundary: every hot-path "kernel" is an
 * ATen op dispatched from Python.  Each entry point below replaces the ATen call sites
 * cited beside it; the Python modules in jointimagegeneration_amd/ call these through ctypes exactly
 * where the reference modules call torch (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions:
 *   - plain C: device pointers as void*, explicit sizes, no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = default stream);
 *   - the library never allocates, frees or retains device memory and never synchronises the device:
 *     every call only enqueues kernels on `stream`, so it is legal inside hipGraph capture;
 *   - every function returns 0 (GG_OK) or a negative gg_status; gg_last_error() gives a thread-local message;
 *   - activations are channels-last ("CL"): [N, D, H, W, C] with C contiguous, bf16 unless stated,
 *     2-D tensors use D == 1; the channel count of a bf16 CL tensor is padded to a multiple of 32
 *     and the pad lanes hold zeros.
 */
#ifndef GUIDEGEN_HIP_H
#define GUIDEGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GG_OK = 0,
    GG_ERR_BAD_SHAPE = -1,
    GG_ERR_BAD_DTYPE = -2,
    GG_ERR_UNSUPPORTED = -3,
    GG_ERR_WORKSPACE_TOO_SMALL = -4,
    GG_ERR_HIP = -5
} gg_status;

typedef enum { GG_BF16 = 0, GG_F32 = 1 } gg_dtype;

const char *gg_last_error(void);
int gg_version(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA (v_mfma_f32_16x16x32_bf16), bf16 in / fp32 accumulate.
 * Replaces nn.Conv{1,2,3}d call sites:
 *   ResBlock in_layers[2]/out_layers[3]/skip_connection  ccdm/.../unet_openai/unet.py:188-228, ldm/.../openaimodel.py:204-244
 *   stem / head conv                                      unet.py:522,719 ; openaimodel.py:522,688
 *   Downsample.op (stride 2, pad 1)                       unet.py:135-139 ; openaimodel.py:152-156
 *   Upsample: F.interpolate(nearest x2) + conv            unet.py:106-116 ; openaimodel.py:109-119 (fused: upsample=1)
 *   AttentionBlock qkv / proj_out (1x1)                   unet.py:291-301 ; openaimodel.py:307-317
 *   AE ResnetBlock / Downsample(pad (0,1)) / Upsample     ldm/modules/diffusionmodules/model.py:42-145
 *   nn.Linear on token rows (SpatialTransformer)          ldm/modules/attention.py:152-215 (ksize=1)
 *   th.cat([h, hs.pop()], 1) before a conv                unet.py:812 ; openaimodel.py:739 (fused: src2/C2)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t N, D, H, W;        /* input extent (before the fused x2 upsample); D = 1 for 2-D            */
    int32_t C1, C2;            /* channels of src1 / src2 (C2 = 0: single source); padded, multiple of 32 */
    int32_t Cout;              /* logical output channels                                                */
    int32_t Cout_pad;          /* channel stride of `out` (multiple of 32 for bf16 outputs)              */
    int32_t kd, kh, kw;        /* kernel extent per dim: 1 or 3 (kd = 1 for 2-D)                         */
    int32_t stride;            /* 1 or 2 (applied to every dim with k > 1 or to all dims if k == 1)     */
    int32_t pad;               /* leading zero padding per dim with k == 3 (1 = 'same'; 0 = AE Downsample) */
    int32_t upsample;          /* 1: nearest-neighbour x2 on D(if kd==3),H,W is fused in front of the conv */
    int32_t Do, Ho, Wo;        /* output extent                                                          */
    int32_t out_dtype;         /* GG_BF16 or GG_F32                                                      */
    int32_t prologue_act;      /* fused GroupNorm prologue applied while gathering (zero padding stays zero):
                                  0 none, 1: silu(x*gn_scale + gn_shift), 2: x*gn_scale + gn_shift        */
    int32_t path_hint;         /* 0 = production dispatch.  1 (tests only) = take the halo-tile kernel whenever the shape is inside
                                  its envelope, even when the grid would under-fill the chip (the production gate then prefers
                                  the box / split-K kernels): lets small test shapes exercise the kernel the big shapes use, with its
                                  512-position box.  4 / 6 (tests only) = the same with the 256-position box the production dispatch
                                  picks for 3-D grids of <= 256 workgroups / the 1024-position box it picks for grids of >= 512 */
    const void *src1;          /* bf16 CL [N,D,H,W,C1]                                                   */
    const void *src2;          /* bf16 CL [N,D,H,W,C2] or NULL                                           */
    const void *weight;        /* packed by gg_conv_pack_weight                                          */
    const float *bias;         /* fp32 [*, Cout_pad]; row n * bias_stride is added to sample n           */
    int64_t bias_stride;       /* 0: one row shared by the batch (plain conv bias);
                                  Cout_pad: per-sample rows (conv bias + timestep-embedding projection) */
    const void *residual;      /* optional bf16 CL [N,Do,Ho,Wo,Cout_pad] added in the epilogue          */
    void *out;                 /* CL [N,Do,Ho,Wo,Cout_pad]                                              */
    const float *gn_scale;     /* fp32 [N, C1+C2] or NULL (see prologue_act)                             */
    const float *gn_shift;
    void *workspace;           /* caller-owned scratch (split-K slabs); size from gg_conv_workspace_bytes */
    int64_t workspace_bytes;
    void *reserved_ptr;        /* must be NULL (was: tickets of an in-launch split-K combine, measured slower and removed) */
    int64_t *gn_acc;           /* optional [N][S][Cout_pad][2] int64 (S = gg_conv_emits_stats(desc) stripes by position tile, summed by the consumer), caller-zeroed: the epilogue adds, per output channel, the sum
                                  and the sum of squares of the bf16-rounded outputs in fixed point (2^28 / 2^20 fractional
                                  bits; integer atomics commute, so the result is bit-reproducible).  It is the GroupNorm
                                  statistics of the NEXT norm, consumed by gg_groupnorm_apply_acc (the 1-stripe layout of the box / 160-step kernels) or gg_groupnorm_scale_shift_acc.
                                  Only filled when gg_conv_emits_stats(desc) != 0, which also gives the stripe count (1: box / 160-step
                                  kernels without split-K; 32: halo-tile kernel); bf16 output only. */
    /* Optional fused DDIM update as the epilogue of the UNet HEAD conv (ldm/models/diffusion/ddim.py:190-204; replaces a separate
     * gg_ddim_step launch).  Only when gg_conv_fuses_ddim(desc) == 1 (box kernel, Cout == 4 == channels of the state, fp32 output): for
     * every output position the epilogue, besides storing eps, computes pred_x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t) and x_prev =
     * sqrt(a_prev) pred_x0 + sqrt(1 - a_prev - sigma^2) eps (sigma-noise NOT added: deterministic eta = 0 steps only) with the same
     * fp32 expression order as gg_ddim_step, and writes x_prev over ddim_x, pred_x0 to ddim_pred_x0, bf16(x_prev) into channels [0, 4)
     * of ddim_unet_in.  All NULL = plain conv. */
    float *ddim_x;               /* fp32 CL [M, 4] state, updated in place                                 */
    const float *ddim_scalars;   /* device fp32[4]: a_t, a_prev, sigma, sqrt(1 - a_t)                      */
    float *ddim_pred_x0;         /* fp32 CL [M, 4] or NULL                                                 */
    void *ddim_unet_in;          /* bf16 CL [M, ddim_unet_in_stride] or NULL                               */
    int64_t ddim_unet_in_stride;
    /* Optional fused GEGLU epilogue of the feed-forward projection (ldm/modules/attention.py:37-44 `x, gate = proj(x).chunk(2);
     * x * gelu(gate)`; replaces a separate gg_geglu launch and the round trip of the [M, 2*inner] projection).  1x1(x1) conv, bf16
     * output, Cout = 2*inner with inner % 16 == 0, no residual, and the weight / bias rows arranged in groups of 32 as [16 value rows
     * j0..j0+15 | their 16 gate rows inner+j0..inner+j0+15]: the output then has inner channels, out[m, j] =
     * (acc_value + bias_value)[j] * gelu_erf((acc_gate + bias_gate)[j]) from the fp32 accumulators, row stride Cout_pad / 2.
     * 0 = plain conv. */
    int32_t epilogue_geglu;
    /* Optional GroupNorm prologue computed FROM ACCUMULATORS inside the conv (no statistics launch, no scale / shift launch, no apply
     * launch): with prologue_act != 0, gn_scale == gn_shift == NULL and pro_acc1 != NULL, every workgroup folds the per-channel
     * fixed-point (sum, sumsq) accumulators the PRODUCING convs left for src1 / src2 (the 1-stripe layout [N][1][C][2] of
     * gg_conv_desc.gn_acc, i.e. producers with gg_conv_emits_stats == 1) into the GroupNorm(32) scale / shift table in LDS and applies
     * normalise * affine (* SiLU) in place to its staged input box.  Only where gg_conv_prologue_from_acc(desc) == 1 (box kernel:
     * affine-only norms always, SiLU norms where few cout tiles share a box or the image is tiny). */
    int32_t pro_c_logical;       /* logical channels of cat[src1, src2] (multiple of 32 groups)            */
    const int64_t *pro_acc1;     /* [N][1][C1][2]                                                          */
    const int64_t *pro_acc2;     /* [N][1][C2][2] or NULL (C2 == 0)                                        */
    const float *pro_gamma;      /* fp32 [pro_c_logical]                                                   */
    const float *pro_beta;
    float pro_eps;
    /* Optional K-concatenated 1x1 skip projection of a ResBlock (out = conv3x3(h) + conv1x1(x): unet.py:228-262, openaimodel.py:244-278
     * `self.skip_connection(x) + h`): the conv takes x (one or two sources, the skip concat) as extra input channels at the centre
     * tap, so the separate skip-conv launch and the residual read disappear.  skip_weight: the 1x1 weight packed by gg_conv_pack_weight
     * (ntaps = 1, Cin_pad = skip_C1 + skip_C2); `bias` must already hold conv bias + skip bias; `residual` must be NULL.  Only where
     * gg_conv_fuses_skip(desc) == 1 (box kernel, 3x3, stride 1, no upsample).  skip_C1 == 0: plain conv. */
    int32_t skip_C1, skip_C2;    /* channels of skip_src1 / skip_src2 (padded, multiples of 32)             */
    int32_t reserved_tail;
    const void *skip_src1;       /* bf16 CL [N,D,H,W,skip_C1] (same extent as the conv's input)            */
    const void *skip_src2;       /* bf16 CL [N,D,H,W,skip_C2] or NULL                                      */
    const void *skip_weight;
    /* Optional fused CCDM reverse step as the epilogue of the UNet HEAD conv (ccdm/ddpm/models/DenoisingModel: softmax of the head,
     * posterior q(x_{t-1} | x_t, x_0) summed over the predicted x_0, categorical draw; replaces a separate gg_ccdm_posterior_sample launch and
     * the round trip of the fp32 logits, 128 B per voxel).  Only when gg_conv_fuses_posterior(desc) == 1 (halo-tile kernel with the
     * 1024-position 3-D box, Cout = K <= 16, fp32 output, no residual).  For every output position m (= voxel index n, d, h, w) the
     * epilogue runs, on the fp32 accumulator + bias (the logits), exactly the arithmetic of gg_ccdm_posterior_sample (same device
     * function, so the same labels bit for bit), writes the new label to post_labels_out[m] (may alias post_xt) and, when
     * post_onehot_out != NULL, channels [0, K) of row m of the one-hot UNet input (bf16, row stride post_onehot_stride, even).
     * `out` is NOT written in this mode (may be NULL).  post_xt == NULL: plain conv. */
    const int32_t *post_xt;        /* int32 [M] current labels x_t                                            */
    int32_t *post_labels_out;      /* int32 [M]                                                               */
    const float *post_scalars;     /* device fp32[2], as gg_ccdm_posterior_sample's scalars_dev               */
    const float *post_E;           /* optional fp32 [M, K] exponentials (a tape); NULL: Philox                */
    uint64_t post_philox_seed;
    const int64_t *post_philox_offset_dev;   /* device int64[1] step offset of the Philox counter, or NULL (0)  */
    void *post_onehot_out;         /* bf16 [M, post_onehot_stride] or NULL                                    */
    int64_t post_onehot_stride;
    int32_t post_draw;             /* 1: categorical draw (exponential race); 0: argmax of the posterior      */
    int32_t reserved_tail2;
} gg_conv_desc;

/* Bytes of the packed weight for a conv with the given logical shape. */
int64_t gg_conv_packed_weight_bytes(int32_t Cout, int32_t Cin_pad, int32_t ntaps);
/* Repack an fp32 OI[D]HW weight (the checkpoint layout) into the MFMA tile order
 * [Cout_pad/32][tap][Cin_pad/32][32 co][32 ci] bf16, zero-filling padded rows/cols.
 * `w_f32` is a device pointer, logical shape [Cout, Cin, ntaps]; cin_map (host, may be NULL) is not used. */
int gg_conv_pack_weight(const float *w_f32, int32_t Cout, int32_t Cin, int32_t Cin_pad, int32_t ntaps,
                        void *packed_bf16, void *stream);
/* Scratch bytes gg_conv_forward needs for this shape (0 for most; > 0 when the under-filled grid is split over K). */
int64_t gg_conv_workspace_bytes(const gg_conv_desc *desc);
/* 1 if the caller should hand the GroupNorm (* SiLU) in front of this conv to the conv (gn_scale / gn_shift + prologue_act: applied
 * once per staged element, the normalised activation never written to HBM), 0 if a separate gg_groupnorm_apply pass in front of a
 * prologue-free conv is the faster form.  Halo-tile shapes CAN always fuse it (path_hint != 0 answers that); with path_hint 0 the answer
 * follows the measured rule in gg_conv_halo.hip (separate where a box is re-staged by several cout groups).  Box-kernel shapes: where the box is staged by at most two cout tiles.  Pointers are not read. */
int gg_conv_fuses_prologue(const gg_conv_desc *desc);
/* 1 if gg_conv_forward(desc) can run the CCDM reverse step as this (head) conv's epilogue (gg_conv_desc.post_xt).  Pointers are not
 * dereferenced; residual, gn_acc, ddim_x and the skip projection must be unset (they have no meaning on a head conv with this epilogue). */
int gg_conv_fuses_posterior(const gg_conv_desc *desc);
/* 1 if gg_conv_forward(desc) runs this shape on the halo-tile kernel (3x3(x3), stride 1, filled grid). Pointers are not read. */
int gg_conv_runs_halo_tile(const gg_conv_desc *desc);
/* 0 if gg_conv_forward(desc) will not fill desc->gn_acc, else the number of stripes S of the [N][S][Cout_pad][2] accumulator it fills
 * (1 for the box / 160-step kernels, 32 for the halo-tile kernel); pointers are not read. */
int gg_conv_emits_stats(const gg_conv_desc *desc);
/* 1 if gg_conv_forward(desc) can compute the GroupNorm prologue desc->prologue_act from accumulators (gg_conv_desc.pro_acc1) on its
 * own; pointers are not read. */
int gg_conv_prologue_from_acc(const gg_conv_desc *desc);
/* 1 if gg_conv_forward(desc) can take desc->skip_C1 / skip_C2 channels of a K-concatenated 1x1 skip projection; pointers are not read. */
int gg_conv_fuses_skip(const gg_conv_desc *desc);
/* 1 if gg_conv_forward(desc) can run the fused DDIM epilogue (see gg_conv_desc.ddim_x); pointers are not read. */
int gg_conv_fuses_ddim(const gg_conv_desc *desc);
int gg_conv_forward(const gg_conv_desc *desc, void *stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm(32 groups) statistics and fused normalise*affine(+SiLU).
 * Replaces GroupNorm32 / Normalize + nn.SiLU / nonlinearity:
 *   ccdm/.../unet_openai/nn.py:17-19,93-100 ; ldm/modules/diffusionmodules/util.py:199-216 ;
 *   ldm/modules/diffusionmodules/model.py:33-39 ; ldm/modules/attention.py:76-77
 * Two-source aware (fuses the skip concat).  Statistics are accumulated in fp32 per block and combined
 * in fp64; results are written as per-(n,c) fp32 scale/shift so that y = x*scale + shift.
 * ------------------------------------------------------------------------------------------------ */
int64_t gg_groupnorm_workspace_bytes(int32_t N, int64_t S, int32_t C);
int gg_groupnorm_stats(const void *src1, int32_t C1, const void *src2, int32_t C2, int32_t N, int64_t S,
                       int32_t C_logical, const float *gamma, const float *beta, float eps,
                       float *scale_out, float *shift_out, void *workspace, int64_t workspace_bytes, void *stream);
/* y = act(x*scale[n,c] + shift[n,c]) ; act: 0 none, 1 SiLU.  out: bf16 CL [N,S,C1+C2]. */
int gg_groupnorm_apply(const void *src1, int32_t C1, const void *src2, int32_t C2, int32_t N, int64_t S,
                       const float *scale, const float *shift, int32_t act, void *out, void *stream);
/* Statistics + normalise*affine(+SiLU) in ONE launch for small tensors (the <= 16x16 UNet levels): a block keeps its group in registers
 * between the two phases.  gg_groupnorm_fused_supported says whether (S, C1, C2, C_logical) fits (C_logical == C1 + C2 required);
 * same result as gg_groupnorm_stats + gg_groupnorm_apply. */
int gg_groupnorm_fused_supported(int64_t S, int32_t C1, int32_t C2, int32_t C_logical);
int gg_groupnorm_fused(const void *src1, int32_t C1, const void *src2, int32_t C2, int32_t N, int64_t S, int32_t C_logical,
                       const float *gamma, const float *beta, float eps, int32_t act, void *out, void *stream);
/* The same normalise*affine(+SiLU), with the statistics taken from the per-channel fixed-point accumulators that the producing
 * convs left behind (gg_conv_desc.gn_acc): acc1 [N][1][C1][2], acc2 [N][1][C2][2] (NULL iff C2 == 0) -- ONLY the 1-stripe layout,
 * i.e. accumulators of convs for which gg_conv_emits_stats(desc) == 1 (box / 160-step kernels); the call has no stripe arguments,
 * so the 32-stripe accumulators of the halo-tile kernel (gg_conv_emits_stats == 32) must go through gg_groupnorm_scale_shift_acc
 * instead (handing them here would read 1/32 of the sums without an error).  Every block folds the accumulators into the
 * per-channel scale/shift table in LDS (fp64), so no statistics launch is needed. */
int gg_groupnorm_apply_acc(const void *src1, int32_t C1, const int64_t *acc1, const void *src2, int32_t C2, const int64_t *acc2,
                           int32_t N, int64_t S, int32_t C_logical, const float *gamma, const float *beta, float eps,
                           int32_t act, void *out, void *stream);

/* Per-(n, c) fp32 scale / shift (y = x*scale + shift == GroupNorm(32)(cat[src1, src2])) folded from the accumulators the producing convs
 * left (acc1 [N][stripes1][C1][2], acc2 [N][stripes2][C2][2] or NULL), for consumers that apply the norm themselves (the halo-tile conv's
 * fused prologue).  Replaces the gg_groupnorm_stats pass over the tensor. */
int gg_groupnorm_scale_shift_acc(const int64_t *acc1, int32_t stripes1, int32_t C1, const int64_t *acc2, int32_t stripes2, int32_t C2,
                                 int32_t N, int64_t S, int32_t C_logical, const float *gamma, const float *beta, float eps,
                                 float *scale_out, float *shift_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Attention: out = softmax(scale * Q K^T) V, flash-style (no TxT buffer), MFMA 16x16x32 bf16, fp32 softmax.
 * Replaces QKVAttentionLegacy (unet.py:334-360, openaimodel.py:349-371), CrossAttention
 * (ldm/modules/attention.py:170-193) and AttnBlock2d's bmm/softmax/bmm (model.py:243-257).
 * Element (n, t, h, d) of X in {Q,K,V,O} lives at X + ((n*T + t)*ld + h*hs + d).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t N, heads, head_dim;     /* head_dim in {32, 64, 128, 256, 384, 512}                          */
    int32_t Tq, Tkv;
    int64_t ldq, hsq, ldk, hsk, ldv, hsv, ldo, hso;
    float scale;
    int32_t reserved;
    const void *q, *k, *v;          /* bf16 */
    void *out;                      /* bf16 */
    void *workspace;                /* optional caller-owned scratch of gg_attention_workspace_bytes(desc) bytes: under-filled single-head
                                       grids (AE mid-block attention, one head of 384 / 512 channels over 4096 tokens) then split the KEYS over
                                       several workgroups and merge the online-softmax states in a second launch; NULL / too small: unsplit */
    int64_t workspace_bytes;
} gg_attention_desc;
int gg_attention_forward(const gg_attention_desc *desc, void *stream);
/* Scratch bytes with which gg_attention_forward splits the keys of this shape (0: it never splits it); pointers are not read. */
int64_t gg_attention_workspace_bytes(const gg_attention_desc *desc);

/* LayerNorm over the last dim (nn.LayerNorm eps 1e-5, attention.py:203-205), bf16 rows -> bf16 rows. */
int gg_layernorm(const void *x, int64_t rows, int32_t C, const float *gamma, const float *beta, float eps,
                 void *out, void *stream);
/* GEGLU: out[r, j] = h[r, j] * gelu_erf(h[r, inner + j]) (attention.py:37-44). */
int gg_geglu(const void *h, int64_t rows, int32_t inner, void *out, void *stream);
/* out = a + b (bf16, elementwise; residual adds of SpatialTransformer / attention). */
int gg_add(const void *a, const void *b, int64_t n, void *out, void *stream);

/* Small fp32 linear: out[m, o] = sum_i act(in[m, i]) * W[o, i] + b[o]; act: 0 none, 1 SiLU on the input.
 * Replaces time_embed / emb_layers (unet.py:205-211,511-515 ; openaimodel.py:221-227,510-514). */
int gg_linear_f32(const float *in, int32_t M, int32_t I, const float *W, const float *b, int32_t O,
                  int32_t act_in, float *out, int64_t out_stride, void *stream);
/* Sinusoidal embedding cos||sin (nn.py:103-121 ; util.py:151-171), t as fp32. */
int gg_timestep_embedding(const float *t, int32_t M, int32_t dim, float max_period, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Layout / dtype movers at the module boundary (NC[D]HW fp32 <-> CL bf16/fp32).
 * ------------------------------------------------------------------------------------------------ */
int gg_nchw_f32_to_cl_bf16(const float *src, int32_t N, int32_t C, int64_t S, void *dst, int32_t C_pad,
                           int32_t c_offset, int32_t zero_fill, void *stream);
int gg_cl_to_nchw_f32(const void *src, int32_t src_dtype, int32_t N, int32_t C, int64_t S, int32_t C_pad,
                      float *dst, void *stream);

/* ------------------------------------------------------------------------------------------------
 * CCDM categorical reverse step (per voxel, fully fused):
 *   [softmax over K logits] -> theta_post_prob -> clamp 1e-12 -> renormalise -> argmax_k p_k/E_k
 * Replaces nn.Softmax head (unet.py:715-721), DiffusionModel.theta_post_prob
 * (ccdm/ddpm/models/diffusion_denoising.py:105-139), torch.clamp (:216) and
 * OneHotCategoricalBCHW.sample/max_prob_sample/prob_sample (one_hot_categorical.py:30-55).
 *   head       fp32 CL [M, head_stride]: probabilities (head_is_logits = 0) or logits (1)
 *   xt         int32 [M] current labels
 *   E          fp32 [M, K] exponential tape, or NULL
 *   philox_seed/philox_offset: used when E == NULL and draw != 0 (counter-based Exp(1) generator)
 *   draw       1: sample (t > 1); 0: argmax of the normalised posterior (t == 1)
 *   scalars    device fp32[2] = {alphas[t-1] (0 at t==1), cumalphas[t-2] (1 at t==1)}
 * Outputs: labels_out int32 [M] (may alias xt); probs_out fp32 [M, K] normalised posterior or NULL;
 *          onehot_out bf16 CL [M, onehot_stride] (channels < K one-hot, rest untouched; onehot_stride even, rows 4-byte aligned) or NULL.
 * ------------------------------------------------------------------------------------------------ */
int gg_ccdm_posterior_sample(const float *head, int32_t head_stride, int32_t head_is_logits, const int32_t *xt,
                             const float *E, uint64_t philox_seed, const int64_t *philox_offset_dev, int32_t draw,
                             const float *scalars_dev, int32_t K, int64_t M, int32_t *labels_out, float *probs_out,
                             void *onehot_out, int32_t onehot_stride, void *stream);
/* labels -> one-hot bf16 CL rows (x_T assembly; evaluator.py:135-136 + unet.py:774-775 concat with zeros). */
int gg_labels_to_onehot(const int32_t *labels, int64_t M, int32_t K, void *onehot_out, int32_t stride, void *stream);

/* ------------------------------------------------------------------------------------------------
 * DDIM update (ldm/models/diffusion/ddim.py:190-204), fp32, elementwise on CL tensors:
 *   pred_x0 = (x - sqrt(1-a_t) e)/sqrt(a_t);  x_prev = sqrt(a_prev) pred_x0 + sqrt(1-a_prev-sigma^2) e + sigma*noise
 *   scalars device fp32[4] = {a_t, a_prev, sigma_t, sqrt_one_minus_a_t};  noise may be NULL (treated as 0).
 *   x [M, C] fp32 (in/out), eps [M, eps_stride] fp32, pred_x0_out optional;
 *   unet_in optional bf16 CL [M, unet_in_stride]: channels [0, C) are refreshed with x_prev.
 * ------------------------------------------------------------------------------------------------ */
int gg_ddim_step(float *x, const float *eps, int32_t eps_stride, const float *noise, const float *scalars_dev,
                 int64_t M, int32_t C, float *pred_x0_out, void *unet_in, int32_t unet_in_stride, void *stream);

/* Ancestral DDPM step (LatentDiffusion.p_sample: ldm/models/diffusion/ddpm.py:217-230,1060-1120), fp32 elementwise:
 *   x_recon = s[0]*x - s[1]*eps ; mean = s[2]*x_recon + s[3]*x ; x <- mean + s[4]*noise   (scalars device fp32[5],
 *   s[4] = (t > 0) * exp(0.5 * posterior_log_variance_clipped[t]); noise may be NULL). unet_in as in gg_ddim_step. */
int gg_ddpm_step(float *x, const float *eps, int32_t eps_stride, const float *noise, const float *scalars_dev, int64_t M, int32_t C,
                 void *unet_in, int32_t unet_in_stride, void *stream);

/* PLMS multistep combination of noise estimates (ldm/models/diffusion/plms.py:218-232), fp32, evaluated left to right:
 *   out = (c0*e0 + c1*e1 + c2*e2 + c3*e3) / denom ; e1..e3 may be NULL (skipped). */
int gg_lincomb4(const float *e0, const float *e1, const float *e2, const float *e3, float c0, float c1, float c2, float c3,
                float denom, int64_t n, float *out, void *stream);

/* Slice normalisation (ds - min)/(max - min) over the whole tensor (latentdiffusion/sample_diffusion.py:222).
 * workspace: >= 2 floats, zero-initialised by the call. */
int gg_minmax_normalise(const float *src, int64_t n, float *dst, float *workspace2, void *stream);

/* Stage glue: CCDM labels -> LDM conditioning slice on the device, replacing the host recipe
 * rot90(scipy.ndimage.zoom(mask, target / shape, order=0), k=3) / 255 (latentdiffusion/sample_diffusion.py:199-200):
 * order-0 zoom of the label volume [N,Dm,Hm,Wm] to (D,H,W) with scipy's index rule (output o reads input
 * floor(o * (in-1)/(out-1) + 0.5) in IEEE double, NOT F.interpolate's floor(o*in/out)), torch.rot90(k=3) on (H,W), value
 * label/255 in channel 1, previous generated slice `prev` fp32 [N,H,W] (or NULL = zeros) in channel 0, remaining lanes of
 * the bf16 CL row zeroed (sample_diffusion.py:208-210). mask_out (optional) receives the fp32 mask slice. */
int gg_mask_to_cond_slice(const int32_t *labels, int32_t N, int32_t Dm, int32_t Hm, int32_t Wm, int32_t slice, int32_t D,
                          int32_t H, int32_t W, const float *prev, void *cond_cl, int32_t stride, float *mask_out,
                          void *stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 VALIDATION mode of the CCDM path (gg_f32.hip): the same network functions on fp32 channels-last tensors with fp32 weights
 * and fp32 FMA accumulation in a fixed order, so that integer outputs (labels) can be compared exactly with the fp32 CPU
 * reference (the reference's own precision switch: ccdm/ddpm/models/unet_openai/unet.py:447,742-756).  Not a fast path.
 * ------------------------------------------------------------------------------------------------ */
/* gg_conv_desc with fp32 tensors: src1 / src2 / residual / out are fp32 CL, out_dtype must be GG_F32, `weight` is fp32
 * [taps][C1+C2][Cout_pad] (tap-major, zero padded), bias as in gg_conv_forward; prologue_act, gn_acc, ddim_x, epilogue_geglu must be 0. */
int gg_conv_forward_f32(const gg_conv_desc *desc, void *stream);
/* act(GroupNorm(32)(cat[src1, src2])) on fp32 CL tensors: statistics in fp64, (x - mean) * rstd * gamma + beta in fp32 (ATen's
 * order), act 1 = SiLU with an IEEE division.  out fp32 CL [N, S, C1+C2] (pad lanes zero); workspace: 64 * N floats. */
int gg_groupnorm_f32(const float *src1, int32_t C1, const float *src2, int32_t C2, int32_t N, int64_t S, int32_t C_logical,
                     const float *gamma, const float *beta, float eps, int32_t act, float *out, float *workspace, void *stream);
/* gg_attention_desc with fp32 q / k / v / out (head_dim <= 64): softmax((q a)(k a)^T) v, a = sqrt(scale), all fp32. */
int gg_attention_forward_f32(const gg_attention_desc *desc, void *stream);

/* ------------------------------------------------------------------------------------------------
 * On-box peak measurements for the benchmark's roofline (gg_ubench.hip; SURVEY.md 8d, BASELINE.md 3: fractions are quoted against
 * the vendor peaks AND against what this box delivers).  Measurement infrastructure: nothing on the sampling path calls them.
 * No counterpart in the reference (it publishes no numbers: README.md:1-41).
 * ------------------------------------------------------------------------------------------------ */
/* Register-resident bf16 MFMA loop on every CU: cus * waves_per_simd workgroups of 256 threads, `iters` iterations of 262 144 FLOP
 * per wave (shape 0: 16 x v_mfma_f32_16x16x32_bf16, shape 1: 8 x v_mfma_f32_32x32x16_bf16), random non-zero operands.  The caller
 * times the launch with events on `stream`; *flops_out (host) receives the FLOP count of the launch.  sink: >= 1 float (device). */
int gg_ubench_mfma_bf16(int32_t shape, int32_t iters, int32_t waves_per_simd, float *sink, double *flops_out, void *stream);
/* 16 bytes per lane grid-stride copy src -> dst (device pointers, 16-byte aligned, bytes % 16 == 0): moves 2 * bytes through HBM. */
int gg_ubench_stream_copy(const void *src, void *dst, int64_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GUIDEGEN_HIP_H */
