// Shared between gg_conv.hip (generic gather kernel) and gg_conv_halo.hip (halo-tile kernel).
#pragma once
#include "gg_common.h"

struct ConvParams {
    int N, D, H, W, C1, C2, Cout, Cout_pad;
    int kd, kh, kw, stride, pad, upsample;
    int Do, Ho, Wo, out_dtype, prologue_act;
    int nchunk1, nchunk, ntaps;
    long long M;              // N*Do*Ho*Wo
    long long bias_stride;
    const bf16_t *src1, *src2, *weight, *residual;
    const float *bias, *gn_scale, *gn_shift;
    void *out;
    float *ws;                // split-K slabs [splitk][M][Cout_pad] fp32 (splitk > 1)
    int *counters;            // per output tile arrival tickets (zero on entry, zero on exit) or nullptr
    int splitk;
};

// 16-byte chunk swizzle for 64-byte LDS rows read by ds_read_b128 with lane -> (row = r0 + (l&15), chunk = l>>4).
// chunk ^ ((row>>1)&2) puts the 16 lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) on 16
// distinct 16-byte slots of the 256-byte bank row for EVERY start row r0 (exhaustive search over r0 = 0..15):
// the shifted reads of the conv taps (row offsets +1, +18, +180) stay conflict-free, not only the aligned ones.
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((row >> 1) & 2); }

