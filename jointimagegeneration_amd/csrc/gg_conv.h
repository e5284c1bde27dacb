// Shared between gg_conv.hip (generic gather kernel) and gg_conv_halo.hip (halo-tile kernel).
#pragma once
#include "gg_common.h"

struct ConvParams {
    int N, D, H, W, C1, C2, Cout, Cout_pad;
    int kd, kh, kw, stride, pad, upsample;
    int Do, Ho, Wo, out_dtype, prologue_act;
    int path_hint;            // gg_conv_desc.path_hint (1: tests force the halo kernel below the grid-fill gate)
    int nchunk1, nchunk, ntaps;
    long long M;              // N*Do*Ho*Wo
    long long bias_stride;
    const bf16_t *src1, *src2, *weight, *residual;
    const float *bias, *gn_scale, *gn_shift;
    void *out;
    float *ws;                // split-K slabs [splitk][M][Cout_pad] fp32 (splitk > 1)
    int splitk;
    long long *gn_acc;        // per-channel fixed-point (sum, sumsq) accumulators of the outputs [N][Cout_pad][2], or nullptr
    float *ddim_x, *ddim_pred_x0;      // fused DDIM epilogue of the UNet head conv (gg_conv_desc.ddim_x), box kernel only
    const float *ddim_scalars;
    bf16_t *ddim_unet_in;
    long long ddim_unet_in_stride;
    // multiply-high magics (gg_fastdiv) of the output-position decode m -> (n, od, oh, ow); 0 where M * divisor >= 2^32
    unsigned mg_osp, mg_ohw, mg_wo;
    int epi_geglu;            // gg_conv_desc.epilogue_geglu (generic gather kernel without split-K only)
    // GroupNorm prologue from accumulators (gg_conv_desc.pro_acc1; box kernel only)
    const long long *pro_acc1, *pro_acc2;
    const float *pro_gamma, *pro_beta;
    float pro_eps;
    int pro_clog;
    // K-concatenated 1x1 skip projection (gg_conv_desc.skip_src1; box kernel, 3x3 stride 1 only)
    const bf16_t *skip_src1, *skip_src2, *skip_weight;
    int skip_C1, skip_C2;
    // fused CCDM reverse step of the UNet head conv (gg_conv_desc.post_xt; halo-tile kernel, 1024-position 3-D box only)
    const int *post_xt;
    int *post_labels_out;
    const float *post_scalars, *post_E;
    unsigned long long post_seed;
    const long long *post_offset_dev;
    bf16_t *post_onehot_out;
    long long post_onehot_stride;
    int post_draw;
};

// fixed-point scales of the GroupNorm accumulators: |sum| < 2^35, sumsq < 2^43 per channel and sample
#define GG_ACC_SUM_SCALE 268435456.0f   /* 2^28 */
#define GG_ACC_SQ_SCALE 1048576.0f      /* 2^20 */
// accumulators of the box / 160-step kernels: [N][GG_ACC_STRIPES][C][2].  One stripe: at most a few dozen position tiles add to
// one address, and every block of gn_apply_acc re-reads all stripes (4 stripes: 1606 us per latent-UNet forward, 1 stripe: 1595)
#ifndef GG_ACC_STRIPES
#define GG_ACC_STRIPES 1
#endif
// the halo-tile kernel (thousands of workgroups per launch) stripes 32-way: at most P/32 workgroups add to one address
#define GG_ACC_STRIPES_HALO 32

// 16-byte chunk swizzle for 64-byte LDS rows read by ds_read_b128 with lane -> (row = r0 + (l&15), chunk = l>>4).
// chunk ^ ((row>>1)&2) puts the 16 lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) on 16
// distinct 16-byte slots of the 256-byte bank row for EVERY start row r0 (exhaustive search over r0 = 0..15):
// the shifted reads of the conv taps (row offsets +1, +18, +180) stay conflict-free, not only the aligned ones.
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((row >> 1) & 2); }


// Sum over the 16 lanes of a row (lanes with equal l >> 4) with DPP moves only: quad xor 1, quad xor 2, half-row mirror, row
// mirror.  Every lane of the row ends up with the row total (fixed order: deterministic).  4 VALU ops instead of 4 ds_bpermute.
__device__ __forceinline__ float gg_row16_sum(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
    return x;
}
