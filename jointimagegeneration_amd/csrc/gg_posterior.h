// CCDM reverse step of ONE voxel (shared by the stand-alone sampler kernel in gg_sampler.hip and by the fused epilogue of the UNet head
// conv in gg_conv_halo.hip): softmax of the head's logits, posterior q(x_{t-1} | x_t, x_0) summed over the predicted x_0, clamp,
// normalise, exponential race (or argmax).  The arithmetic is, expression for expression and in the same left-to-right fp32 order (no FMA
// contraction, IEEE division), the C restatement oracle/ccdm_posterior.c, so the labels agree bit-for-bit when probabilities (not
// logits) are fed -- and both callers produce the SAME label from the same fp32 logits.
#pragma once
#include "gg_common.h"

// ------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (counter-based): counter = (voxel lo, voxel hi, draw index, step offset), key = seed.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// p0: the head's K values of voxel m (logits or probabilities), overwritten; x: the voxel's current label; a / abar: the step's scalars
// (alpha_t, cumulative alpha of t - 1); E_row: K exponentials of a tape or nullptr (then Philox draws them); probs_row: K normalised
// posterior probabilities out, or nullptr.  Returns the new label.
// KS > 0: the class count as a compile-time constant (K must equal it): no per-class bound checks (they are scalar branches, ~500 per voxel
// with a run-time K); KS == 0: run-time K <= KMAX.
template <int KMAX, int KS = 0>
__device__ __forceinline__ int ccdm_posterior_voxel(float (&p0)[KMAX], const int is_logits, const int x, const float a, const float abar, const int K_rt,
                                                    const long long m, const int draw, const float *__restrict__ E_row, const uint64_t seed,
                                                    const long long off, float *__restrict__ probs_row)
{
#pragma clang fp contract(off)
    const int K = KS > 0 ? KS : K_rt;
    const float Kf = (float)K;
    const float u = (1.0f - a) / Kf;
    const float v = (1.0f - abar) / Kf;
    const float bd = abar * 1.0f + v;
    const float bo = abar * 0.0f + v;
    float out[KMAX];
    if (is_logits) {   // nn.Softmax(dim=1) of the UNet head, fp32
        float mx = p0[0];
#pragma unroll
        for (int c = 1; c < KMAX; ++c) if (c < K) mx = fmaxf(mx, p0[c]);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) if (c < K) { p0[c] = expf(p0[c] - mx); s = s + p0[c]; }
#pragma unroll
        for (int c = 0; c < KMAX; ++c) if (c < K) p0[c] = p0[c] / s;
    }
    // q(x_{t-1} = c | x_t = x, x_0 = d) = A[c] B[c][d] / den[d] with A[c] = a [c == x] + u and B[c][d] = abar [c == d] + v: A takes TWO
    // values (c == x or not) and B two (c == d or not), so the K numerators of a column d are three distinct fp32 products and the K IEEE
    // divisions three -- the same operands, hence the same bits, as dividing every entry (42 divisions per voxel instead of 196: the
    // kernel is VALU-bound, 4.7 k -> 2.9 k instructions per voxel).  The sums keep the restatement's left-to-right order.
    const float Ax = a * 1.0f + u, Ao = a * 0.0f + u;
    const float n_xd = Ax * bd, n_xo = Ax * bo, n_od = Ao * bd, n_oo = Ao * bo;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) out[c] = 0.f;
#pragma unroll
    for (int d = 0; d < KMAX; ++d) {
        if (d < K) {
            float den = 0.f;
#pragma unroll
            for (int c = 0; c < KMAX; ++c) if (c < K) den = den + ((c == x) ? (c == d ? n_xd : n_xo) : (c == d ? n_od : n_oo));
            const float pd = p0[d];
            const float q_dd = ((d == x) ? n_xd : n_od) / den, q_xo = n_xo / den, q_oo = n_oo / den;
#pragma unroll
            for (int c = 0; c < KMAX; ++c) if (c < K) {
                const float post = (c == d) ? q_dd : ((c == x) ? q_xo : q_oo);
                out[c] = out[c] + post * pd;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) if (c < K) {
        if (out[c] < 1e-12f) out[c] = 1e-12f;
        s = s + out[c];
    }
    // exponential race
    float Ev[KMAX];
    const bool use_race = draw != 0;
    if (use_race) {
        if (E_row) {
#pragma unroll
            for (int c = 0; c < KMAX; ++c) Ev[c] = (c < K) ? E_row[c] : 1.f;
        } else {
#pragma unroll
            for (int q4 = 0; q4 < KMAX / 4; ++q4) {
                uint32_t ctr[4] = {(uint32_t)m, (uint32_t)(m >> 32), (uint32_t)q4, (uint32_t)off};
                philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float uu = (float)((ctr[j] >> 8) + 1u) * 5.9604644775390625e-8f;   // (0, 1]
                    Ev[q4 * 4 + j] = -__logf(uu) + 1e-30f;
                }
            }
        }
    }
    int best = 0;
    float bestv = -1.0f;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) if (c < K) {
        const float pn = out[c] / s;
        const float r = use_race ? pn / Ev[c] : pn;
        if (probs_row) probs_row[c] = pn;
        if (r > bestv) { bestv = r; best = c; }
    }
    return best;
}

// channels [0, K) of the voxel's one-hot row as 4-byte pairs (+ one 2-byte tail for odd K): 7 stores instead of 14 two-byte ones at K = 14
// (two-byte stores cost ~12x a 16-byte store per byte on this memory system, MI355X_MICROARCH.md); channel K and beyond (the
// condition image, the padding lanes) are NOT touched
template <int KMAX, int KS = 0>
__device__ __forceinline__ void ccdm_onehot_row(bf16_t *__restrict__ oh, const int best, const int K_rt)
{
    const int K = KS > 0 ? KS : K_rt;
#pragma unroll
    for (int c = 0; c + 1 < KMAX; c += 2) {
        if (c + 1 < K) {
            bf16x2 pr;
            pr[0] = (bf16_t)(c == best ? 1.0f : 0.0f);
            pr[1] = (bf16_t)(c + 1 == best ? 1.0f : 0.0f);
            *reinterpret_cast<bf16x2 *>(oh + c) = pr;
        } else if (c < K) {
            oh[c] = (bf16_t)(c == best ? 1.0f : 0.0f);
        }
    }
}
