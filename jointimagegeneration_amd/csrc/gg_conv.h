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

// 16-byte chunk swizzle for 64-byte LDS rows read by ds_read_b128 with lane -> (row = l&15, chunk = l>>4):
// conflict-free for 16 consecutive rows (derivation in DESIGN.md, "LDS images").
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((0 - (row >> 2)) & 3); }

