// GroupNorm(32) statistics + fused normalise*affine(+SiLU), LayerNorm, GEGLU, add. HBM-bound kernels:
// 16-byte (8 x bf16) accesses per lane, fp32 math, two-source aware (fused skip concat).
#include "gg_conv.h"

// ------------------------------------------------------------------------------------------------------------
// Stage 1: per-block partial sums. grid = (nblk, N). Block b of sample n owns rows [b*rpb, (b+1)*rpb).
// A thread always sees the same 8-channel piece (piece = tid % P), so its 8+8 running sums stay in registers.
// Partials are written as part[n][b][c][2] (fp32: sum, sumsq per CHANNEL) -- deterministic, no atomics.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t *__restrict__ s1, int C1,
                                                         const bf16_t *__restrict__ s2, int C2, long long S,
                                                         long long rows_per_block, float *__restrict__ part)
{
    const int C = C1 + C2;
    const int P = C >> 3;                     // 16-byte pieces per row
    const int rpi = blockDim.x / P;           // rows per iteration
    const int tid = threadIdx.x;
    const int piece = tid % P, rsub = tid / P;
    const int n = blockIdx.y;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > S) r1 = S;
    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = sq[j] = 0.f;
    if (rsub < rpi) {
        const int c0 = piece * 8;
        const bool second = c0 >= C1;
        const bf16_t *base = second ? s2 + (long long)n * S * C2 + (c0 - C1) : s1 + (long long)n * S * C1 + c0;
        const int Cs = second ? C2 : C1;
        for (long long r = r0 + rsub; r < r1; r += rpi) {
            bf16x8 v = *reinterpret_cast<const bf16x8 *>(base + r * Cs);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (float)v[j];
                sum[j] += f;
                sq[j] += f * f;
            }
        }
    }
    // reduce the rpi row-slots of every piece through LDS
    extern __shared__ float red[];            // [256][16]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid * 16 + j] = sum[j];
        red[tid * 16 + 8 + j] = sq[j];
    }
    __syncthreads();
    if (tid < P) {
        float a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = 0.f;
        for (int k = 0; k < rpi; ++k)
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] += red[(k * P + tid) * 16 + j];
        float *dst = part + (((long long)n * gridDim.x + blockIdx.x) * C + tid * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dst[2 * j] = a[j];
            dst[2 * j + 1] = a[8 + j];
        }
    }
}

// Stage 2: one block per (n, group): combine the block partials of the group's channels in fp64 with a fixed
// reduction tree (deterministic), then fold gamma/beta into per-(n,c) scale/shift:
//   y = x*scale + shift,  scale = rstd*gamma,  shift = beta - mean*rstd*gamma.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float *__restrict__ part, int nblk, int C, int C_logical,
                                                          long long S, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps,
                                                          float *__restrict__ scale, float *__restrict__ shift)
{
    const int n = blockIdx.y, g = blockIdx.x;
    const int cpg = C_logical / 32;
    const int tid = threadIdx.x;
    __shared__ double ra[256], rb[256];
    double a = 0.0, b = 0.0;
    const int total = nblk * cpg;
    // 4 independent 8-byte loads in flight per thread (the fold of 1024 partial blocks is otherwise 16+ serial round trips)
    for (int i0 = tid; i0 < total; i0 += 1024) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            v[u] = float2{0.f, 0.f};
            if (i < total) {
                const int k = i / cpg, j = i - k * cpg;
                v[u] = *reinterpret_cast<const float2 *>(part + (((long long)n * nblk + k) * C + g * cpg + j) * 2);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a += (double)v[u].x; b += (double)v[u].y; }
    }
    ra[tid] = a;
    rb[tid] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { ra[tid] += ra[tid + s]; rb[tid] += rb[tid + s]; }
        __syncthreads();
    }
    const double cnt = (double)S * cpg;
    const double mean = ra[0] / cnt;
    double var = rb[0] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float fmean = (float)mean, frstd = (float)(1.0 / sqrt(var + (double)eps));
    if (tid < cpg) {
        const int c = g * cpg + tid;
        const float sc = frstd * gamma[c];
        scale[(long long)n * C + c] = sc;
        shift[(long long)n * C + c] = beta[c] - fmean * sc;
    }
    if (g == 0)   // zero the pad lanes once
        for (int c = C_logical + tid; c < C; c += 256) { scale[(long long)n * C + c] = 0.f; shift[(long long)n * C + c] = 0.f; }
}

// Small-tensor fast path: ONE launch. Block (g, n) reduces its group's S x cpg elements (fp32 per thread, fp64 tree),
// then writes the folded scale/shift of its channels. Used when N*S*C is small enough that launch latency, not HBM,
// is the cost (LDM latent UNet, deep CCDM levels).
// x + (x of the DPP-selected lane) in fp64: two DPP moves + one add, no LDS round trip (a 64-bit __shfl_xor is two ds_bpermute, ~150 cycles
// of dependent latency per butterfly step: six steps were ~0.4 us of a 3.4 us kernel)
template <int CTRL>
__device__ __forceinline__ double gg_dpp_add_f64(double x)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return x + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum over the 64 lanes of a wave, fixed order (deterministic): the 16 lanes of a row by DPP (quad xor 1, quad xor 2, half-row mirror,
// row mirror), the four rows by two xor shuffles
__device__ __forceinline__ double gg_wave_sum_f64(double x)
{
    x = gg_dpp_add_f64<0xB1>(x);
    x = gg_dpp_add_f64<0x4E>(x);
    x = gg_dpp_add_f64<0x141>(x);
    x = gg_dpp_add_f64<0x140>(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

__global__ __launch_bounds__(256) void gn_stats_small_kernel(const bf16_t *__restrict__ s1, int C1, const bf16_t *__restrict__ s2,
                                                             int C2, long long S, int C_logical, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps, float *__restrict__ scale,
                                                             float *__restrict__ shift)
{
    // every argument in one scalar-load batch (gg_pin)
    s1 = gg_pin(s1); C1 = gg_pin(C1); s2 = gg_pin(s2); C2 = gg_pin(C2); S = gg_pin(S); C_logical = gg_pin(C_logical);
    gamma = gg_pin(gamma); beta = gg_pin(beta); eps = gg_pin(eps); scale = gg_pin(scale); shift = gg_pin(shift);
    const int C = C1 + C2;
    const int n = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int cpg = C_logical / 32;
    const bf16_t *b1 = s1 + (long long)n * S * C1;
    const bf16_t *b2 = s2 ? s2 + (long long)n * S * C2 : nullptr;
    float a = 0.f, b = 0.f;
    // gamma / beta of this thread's channel are requested up front: one memory round trip for the whole kernel, not two
    float gmm = 0.f, bta = 0.f;
    if (tid < cpg) { gmm = gamma[g * cpg + tid]; bta = beta[g * cpg + tid]; }
    // 16-byte loads: the group's channels [c_lo, c_hi) live in pieces p_lo..p_hi of a row; lanes mask the foreign channels
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    const int p_lo = c_lo >> 3, p_hi = (c_hi - 1) >> 3;
    const int np = p_hi - p_lo + 1;
    const int work = (int)S * np;                       // < 2^19 (host gate: S * C <= 2^19)
    const float rnp = __builtin_amdgcn_rcpf((float)np);
    for (int i = tid; i < work; i += 256) {
        const int r = gg_div_small(i, rnp);
        const int c0 = (p_lo + (i - r * np)) * 8;
        const bf16_t *src = (c0 < C1) ? b1 + (long long)r * C1 + c0 : b2 + (long long)r * C2 + (c0 - C1);
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(src);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            const float f = (c >= c_lo && c < c_hi) ? (float)v[j] : 0.f;
            a += f;
            b += f * f;
        }
    }
    // fixed-order reduction: fp64 butterfly inside each wave (6 steps), then the 4 waves through LDS: one barrier, not eight
    const double da = gg_wave_sum_f64((double)a), db = gg_wave_sum_f64((double)b);
    __shared__ double ra[4], rb[4];
    if ((tid & 63) == 0) { ra[tid >> 6] = da; rb[tid >> 6] = db; }
    __syncthreads();
    const double sa = (ra[0] + ra[1]) + (ra[2] + ra[3]), sb = (rb[0] + rb[1]) + (rb[2] + rb[3]);
    // (no fp64 division / square root: each is a ~50-100-instruction sequence every thread would issue between the reduction and
    // its stores; the same formulas as gn_apply_acc_kernel: fp32 reciprocal of the exact count, v_rsq_f32)
    const double cnt = (double)S * (double)cpg; double inv = (double)(1.0f / (float)cnt); inv = inv * (2.0 - cnt * inv);      // fp32 reciprocal + one Newton step in fp64 (error ~1e-14: no fp64 division, and mean^2 is not polluted when |mean| >> std)
    const double mean = sa * inv;
    double var = sb * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    const float fmean = (float)mean, frstd = rsqrtf((float)var + eps);
    if (tid < cpg) {
        const int c = g * cpg + tid;
        const float sc = frstd * gmm;
        scale[(long long)n * C + c] = sc;
        shift[(long long)n * C + c] = bta - fmean * sc;
    }
    if (g == 0)
        for (int c = C_logical + tid; c < C; c += 256) { scale[(long long)n * C + c] = 0.f; shift[(long long)n * C + c] = 0.f; }
}

// Small tensors, ONE launch for the whole norm: block (g, n) keeps its group's S x cpg elements in registers (as the 16-byte pieces
// that cover the group's channels; foreign lanes masked), reduces them (fp32 per thread, fp64 butterfly, fixed order), then
// normalises * affine (* SiLU) the same registers and stores them: interior pieces as 16 bytes, the pieces it shares with the
// neighbouring groups' blocks element by element (byte-enabled 2-byte stores: no block writes a foreign channel).
// Replaces gn_stats_small + gn_apply (two dependent launches) at the 8x8 / 4x4 UNet levels only (see gn_fused_small_ok).
#define GG_GN_FUSED_MAXP 2      /* pieces per thread kept in registers */
__global__ __launch_bounds__(256) void gn_fused_small_kernel(const bf16_t *__restrict__ s1, int C1, const bf16_t *__restrict__ s2, int C2,
                                                             long long S, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float eps, int act, bf16_t *__restrict__ out)
{
    // every argument in one scalar-load batch (gg_pin)
    s1 = gg_pin(s1); C1 = gg_pin(C1); s2 = gg_pin(s2); C2 = gg_pin(C2); S = gg_pin(S); gamma = gg_pin(gamma); beta = gg_pin(beta);
    eps = gg_pin(eps); act = gg_pin(act); out = gg_pin(out);
    const int C = C1 + C2;
    const int n = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / 32;
    const bf16_t *b1 = s1 + (long long)n * S * C1;
    const bf16_t *b2 = s2 ? s2 + (long long)n * S * C2 : nullptr;
    bf16_t *o = out + (long long)n * S * C;
    __shared__ float gb[2][64];                             // gamma / beta of the group's channels (cpg <= 64)
    if (tid < cpg) { gb[0][tid] = gamma[g * cpg + tid]; gb[1][tid] = beta[g * cpg + tid]; }
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    const int p_lo = c_lo >> 3, p_hi = (c_hi - 1) >> 3;
    const int np = p_hi - p_lo + 1;
    const int work = (int)S * np;
    const float rnp = __builtin_amdgcn_rcpf((float)np);
    u32x4 v[GG_GN_FUSED_MAXP];
#pragma unroll
    for (int k = 0; k < GG_GN_FUSED_MAXP; ++k) {
        const int i = tid + 256 * k;
        v[k] = u32x4{0u, 0u, 0u, 0u};
        if (i < work) {
            const int r = gg_div_small(i, rnp);
            const int c0 = (p_lo + (i - r * np)) * 8;
            v[k] = *reinterpret_cast<const u32x4 *>((c0 < C1) ? b1 + (long long)r * C1 + c0 : b2 + (long long)r * C2 + (c0 - C1));
        }
    }
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < GG_GN_FUSED_MAXP; ++k) {
        const int i = tid + 256 * k;
        if (i < work) {
            const int r = gg_div_small(i, rnp);
            const int c0 = (p_lo + (i - r * np)) * 8;
            const bf16x8 x = __builtin_bit_cast(bf16x8, v[k]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j;
                const float f = (c >= c_lo && c < c_hi) ? (float)x[j] : 0.f;
                a += f;
                b += f * f;
            }
        }
    }
    const double da = gg_wave_sum_f64((double)a), db = gg_wave_sum_f64((double)b);
    __shared__ double ra[4], rb[4];
    if ((tid & 63) == 0) { ra[tid >> 6] = da; rb[tid >> 6] = db; }
    __syncthreads();
    const double sa = (ra[0] + ra[1]) + (ra[2] + ra[3]), sb = (rb[0] + rb[1]) + (rb[2] + rb[3]);
    const double cnt = (double)S * (double)cpg; double inv = (double)(1.0f / (float)cnt); inv = inv * (2.0 - cnt * inv);      // fp32 reciprocal + Newton step in fp64; no fp64 division / sqrt (see above)
    const double mean = sa * inv;
    double var = sb * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    const float fmean = (float)mean, frstd = rsqrtf((float)var + eps);
#pragma unroll
    for (int k = 0; k < GG_GN_FUSED_MAXP; ++k) {
        const int i = tid + 256 * k;
        if (i < work) {
            const int r = gg_div_small(i, rnp);
            const int c0 = (p_lo + (i - r * np)) * 8;
            const bf16x8 x = __builtin_bit_cast(bf16x8, v[k]);
            bf16x8 y;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int cl = c0 + j - c_lo;                         // channel inside the group (clamped for the foreign lanes: not stored)
                cl = cl < 0 ? 0 : (cl >= cpg ? cpg - 1 : cl);
                const float sc = frstd * gb[0][cl];             // the same fp32 scale / shift the two-launch path tabulates
                float t = (float)x[j] * sc + (gb[1][cl] - fmean * sc);
                if (act) t = gg_silu(t);
                y[j] = (bf16_t)t;
            }
            bf16_t *dst = o + (long long)r * C + c0;
            if (c0 >= c_lo && c0 + 8 <= c_hi) {
                *reinterpret_cast<bf16x8 *>(dst) = y;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c0 + j >= c_lo && c0 + j < c_hi) dst[j] = y[j];
            }
        }
    }
}

static bool gn_fused_small_ok(long long S, int C1, int C2, int C_logical)
{
    const int C = C1 + C2;
    if (C_logical != C || C % 32 || C / 32 > 64 || S > (1 << 20)) return false;
    const int cpg = C / 32;
    int np = 0;                                              // most pieces any group's channel range touches
    for (int g = 0; g < 32; ++g) {
        const int n = (((g + 1) * cpg - 1) >> 3) - ((g * cpg) >> 3) + 1;
        np = n > np ? n : np;
    }
    // measured (tools/probe_gn.py, us, two launches vs this kernel): 8x8x640 5.0 vs 4.5, 4x4x800 4.9 vs 4.3, but 16x16x640 5.8 vs 7.2 and
    // 32x32x320 7.7 vs 14.0: past two pieces per thread the serial work of the 32 blocks costs more than the saved launch
    return S * np <= 512;
}

extern "C" int gg_groupnorm_fused_supported(int64_t S, int32_t C1, int32_t C2, int32_t C_logical)
{
    return (C1 > 0 && C1 % 32 == 0 && C2 >= 0 && C2 % 32 == 0 && gn_fused_small_ok(S, C1, C2, C_logical)) ? 1 : 0;
}

extern "C" int gg_groupnorm_fused(const void *src1, int32_t C1, const void *src2, int32_t C2, int32_t N, int64_t S, int32_t C_logical,
                                  const float *gamma, const float *beta, float eps, int32_t act, void *out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (C1 <= 0 || C1 % 32 || C2 % 32 || C2 < 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_fused: C1/C2 must be multiples of 32");
    if (!src1 || (C2 && !src2) || !gamma || !beta || !out || N <= 0 || S <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_fused: null pointer / empty");
    if (!gn_fused_small_ok(S, C1, C2, C_logical)) GG_FAIL(GG_ERR_UNSUPPORTED, "groupnorm_fused: tensor too large for the single-launch path (gg_groupnorm_fused_supported)");
    hipLaunchKernelGGL(gn_fused_small_kernel, dim3(32, N), dim3(256), 0, stream, (const bf16_t *)src1, C1, (const bf16_t *)src2, C2,
                       (long long)S, gamma, beta, eps, act, (bf16_t *)out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

extern "C" int64_t gg_groupnorm_workspace_bytes(int32_t N, int64_t S, int32_t C)
{
    (void)S;
    return (int64_t)N * 1024 * C * 2 * 4;   // up to 1024 partial blocks per sample
}

static int gn_nblk(int N, long long S, int C)
{
    // enough blocks to fill the chip (256 CUs x ~8), at least ~64 rows per block
    long long want = (2048 + N - 1) / N;
    long long maxb = (S + 63) / 64;
    long long nb = want < maxb ? want : maxb;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    (void)C;
    return (int)nb;
}

extern "C" int gg_groupnorm_stats(const void *src1, int32_t C1, const void *src2, int32_t C2, int32_t N, int64_t S,
                                  int32_t C_logical, const float *gamma, const float *beta, float eps, float *scale_out,
                                  float *shift_out, void *workspace, int64_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    const int C = C1 + C2;
    if (C1 <= 0 || C1 % 32 || C2 % 32 || C2 < 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm: C1/C2 must be multiples of 32");
    if (C_logical % 32 || C_logical > C || C_logical <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm: logical channels %d not divisible by 32 groups", C_logical);
    if (C / 8 > 256) GG_FAIL(GG_ERR_UNSUPPORTED, "groupnorm: C > 2048");
    if (!src1 || (C2 && !src2) || !gamma || !beta || !scale_out || !shift_out || !workspace) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm: null pointer");
    if ((long long)S * C <= (1 << 19)) {   // small tensors (deep UNet levels): single-launch path
        hipLaunchKernelGGL(gn_stats_small_kernel, dim3(32, N), dim3(256), 0, stream, (const bf16_t *)src1, C1, (const bf16_t *)src2, C2,
                           (long long)S, C_logical, gamma, beta, eps, scale_out, shift_out);
        GG_CHECK_LAUNCH();
        return GG_OK;
    }
    const int nblk = gn_nblk(N, S, C);
    if ((int64_t)N * nblk * C * 8 > workspace_bytes) GG_FAIL(GG_ERR_WORKSPACE_TOO_SMALL, "groupnorm: workspace %lld < %lld", (long long)workspace_bytes, (long long)N * nblk * C * 8);
    const long long rpb = (S + nblk - 1) / nblk;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, N), dim3(256), 256 * 16 * sizeof(float), stream, (const bf16_t *)src1, C1,
                       (const bf16_t *)src2, C2, (long long)S, rpb, (float *)workspace);
    GG_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, N), dim3(256), 0, stream, (const float *)workspace, nblk, C, C_logical,
                       (long long)S, gamma, beta, eps, scale_out, shift_out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t *__restrict__ s1, int C1, const bf16_t *__restrict__ s2,
                                                       int C2, long long S, long long total_pieces,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       int act, bf16_t *__restrict__ out, unsigned pmagic, int N)
{
    // every argument in one scalar-load batch (gg_pin)
    s1 = gg_pin(s1); C1 = gg_pin(C1); s2 = gg_pin(s2); C2 = gg_pin(C2); S = gg_pin(S); total_pieces = gg_pin(total_pieces);
    scale = gg_pin(scale); shift = gg_pin(shift); act = gg_pin(act); out = gg_pin(out); pmagic = gg_pin(pmagic); N = gg_pin(N);
    const int C = C1 + C2;
    const int P = C >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total_pieces; i += (long long)gridDim.x * 256) {
        long long row = gg_fastdiv(i, P, pmagic);                 // row over N*S
        int piece = (int)(i - row * P);
        int c0 = piece * 8;
        int n = N == 1 ? 0 : (int)(row / S);
        const bf16_t *src = (c0 >= C1) ? s2 + row * C2 + (c0 - C1) : s1 + row * C1 + c0;
        bf16x8 v = *reinterpret_cast<const bf16x8 *>(src);
        const float *sc = scale + (long long)n * C + c0;
        const float *sh = shift + (long long)n * C + c0;
        f32x4 a0 = *reinterpret_cast<const f32x4 *>(sc), a1 = *reinterpret_cast<const f32x4 *>(sc + 4);
        f32x4 b0 = *reinterpret_cast<const f32x4 *>(sh), b1 = *reinterpret_cast<const f32x4 *>(sh + 4);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y0 = (float)v[j] * a0[j] + b0[j];
            float y1 = (float)v[j + 4] * a1[j] + b1[j];
            if (act) { y0 = gg_silu(y0); y1 = gg_silu(y1); }
            o[j] = (bf16_t)y0;
            o[j + 4] = (bf16_t)y1;
        }
        *reinterpret_cast<bf16x8 *>(out + row * C + c0) = o;
    }
}

extern "C" int gg_groupnorm_apply(const void *src1, int32_t C1, const void *src2, int32_t C2, int32_t N, int64_t S,
                                  const float *scale, const float *shift, int32_t act, void *out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (C1 <= 0 || C1 % 32 || C2 % 32 || C2 < 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_apply: C1/C2 must be multiples of 32");
    if (!src1 || (C2 && !src2) || !scale || !shift || !out) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_apply: null pointer");
    long long total = (long long)N * S * ((C1 + C2) / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t *)src1, C1,
                       (const bf16_t *)src2, C2, (long long)S, total, scale, shift, act, (bf16_t *)out, gg_magic_u32(total, (C1 + C2) / 8), N);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// normalise*affine(+SiLU) with the statistics read from the per-channel fixed-point accumulators left behind by the
// producing convs (gg_conv_desc.gn_acc): no statistics launch at all.  Every block folds (sum, sumsq) of its sample's
// channels into the scale/shift table in LDS (fp64 per group), then applies its share of the rows.  All global reads
// (accumulators, gamma/beta, the thread's first pieces) are issued up front: one memory round trip per block.
// Diagnostic build only (-DGG_GN_STAMPS, tools/experiments/probe_gn_stamps.py): s_memrealtime phase stamps of thread 0 of every block
#ifdef GG_GN_STAMPS
__device__ unsigned long long gg_gn_stamp_buf[4096 * 8];
extern "C" int gg_gn_stamps_read(unsigned long long *host, int n) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(gg_gn_stamp_buf), (size_t)n * 8) == hipSuccess ? 0 : 1; }
#define GG_GSTAMP(K) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) gg_gn_stamp_buf[blockIdx.x * 8 + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GG_GSTAMP(K) do { } while (0)
#endif
#ifndef GG_GN_APPLY_ACC_LEAN
#define GG_GN_APPLY_ACC_LEAN 1           /* one-piece-per-thread form of gn_apply_acc for batch-1 shapes (A/B: tools/experiments) */
#endif
#define GG_ACC_SUM_SCALE_D 268435456.0   /* 2^28, must match gg_conv.h */
#define GG_ACC_SQ_SCALE_D 1048576.0      /* 2^20 */
__global__ __launch_bounds__(256) void gn_apply_acc_kernel(const bf16_t *__restrict__ s1, int C1, const long long *__restrict__ acc1,
                                                           const bf16_t *__restrict__ s2, int C2, const long long *__restrict__ acc2,
                                                           long long S, int C_logical, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float eps, int act, bf16_t *__restrict__ out, unsigned pmagic)
{
    // every argument in one scalar-load batch (gg_pin)
    s1 = gg_pin(s1); C1 = gg_pin(C1); acc1 = gg_pin(acc1); s2 = gg_pin(s2); C2 = gg_pin(C2); acc2 = gg_pin(acc2); S = gg_pin(S);
    C_logical = gg_pin(C_logical); gamma = gg_pin(gamma); beta = gg_pin(beta); eps = gg_pin(eps); act = gg_pin(act); out = gg_pin(out);
    pmagic = gg_pin(pmagic);
    GG_GSTAMP(0);
    const int C = C1 + C2;
    const int P = C >> 3;
    const int tid = threadIdx.x, n = blockIdx.y;
    const int cpg = C_logical / 32;
    const float rcpg = __builtin_amdgcn_rcpf((float)cpg);
    extern __shared__ float ss[];                      // scale[C], shift[C]
    __shared__ unsigned long long gacc[32][2];         // per-group integer (sum, sumsq): LDS atomics, exact in any order
    __shared__ float gmean[32], grstd[32];
    const long long pieces = S * P;
    const bf16_t *b1 = s1 + (long long)n * S * C1;
    const bf16_t *b2 = s2 ? s2 + (long long)n * S * C2 : nullptr;
    bf16_t *o = out + (long long)n * S * C;
    constexpr int U = 2, CPT = 8;                      // pieces prefetched per thread; channels per thread (C <= 2048)
    const long long stride = (long long)gridDim.x * 256;
    const long long i0 = (long long)blockIdx.x * 256 + tid;
    if (tid < 64) gacc[tid >> 1][tid & 1] = 0ull;
    u32x4 pv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * stride;
        pv[u] = u32x4{0u, 0u, 0u, 0u};
        if (i < pieces) {
            const long long row = gg_fastdiv(i, P, pmagic);
            const int c0 = (int)(i - row * P) * 8;
            pv[u] = *reinterpret_cast<const u32x4 *>((c0 >= C1) ? b2 + row * C2 + (c0 - C1) : b1 + row * C1 + c0);
        }
    }
    // this thread's channels: GG_ACC_STRIPES x (sum, sumsq) as 16-byte loads, gamma / beta; everything requested before the first wait
    float gam[CPT], bet[CPT];
    long long sa[CPT], sb[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = tid + 256 * k;
        gam[k] = 0.f;
        bet[k] = 0.f;
        sa[k] = 0;
        sb[k] = 0;
        if (c < C_logical) {
            const long long *q = (c < C1) ? acc1 + ((long long)n * GG_ACC_STRIPES * C1 + c) * 2 : acc2 + ((long long)n * GG_ACC_STRIPES * C2 + (c - C1)) * 2;
            const long long cs = (c < C1) ? (long long)C1 * 2 : (long long)C2 * 2;
            typedef __attribute__((ext_vector_type(2))) long long i64x2;
#pragma unroll
            for (int st = 0; st < GG_ACC_STRIPES; ++st) {
                const i64x2 v = *reinterpret_cast<const i64x2 *>(q + st * cs);
                sa[k] += v[0];
                sb[k] += v[1];
            }
            gam[k] = gamma[c];
            bet[k] = beta[c];
        }
    }
    GG_GSTAMP(1);
    __syncthreads();                                   // gacc zeroed
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = tid + 256 * k;
        if (c < C_logical) {
            const int g = gg_div_small(c, rcpg);
            if (k == 0) GG_GSTAMP(2);
            atomicAdd(&gacc[g][0], (unsigned long long)sa[k]);
            atomicAdd(&gacc[g][1], (unsigned long long)sb[k]);
        }
    }
    __syncthreads();
    GG_GSTAMP(3);
    if (tid < 32) {
        const double a = (double)(long long)gacc[tid][0] * (1.0 / GG_ACC_SUM_SCALE_D);
        const double b = (double)(long long)gacc[tid][1] * (1.0 / GG_ACC_SQ_SCALE_D);
        const double cnt = (double)S * (double)cpg; double inv = (double)(1.0f / (float)cnt); inv = inv * (2.0 - cnt * inv);      // fp32 reciprocal + Newton step in fp64 (as gn_stats_small)
        const double mean = a * inv;
        double var = b * inv - mean * mean;
        if (var < 0.0) var = 0.0;
        gmean[tid] = (float)mean;
        grstd[tid] = rsqrtf((float)var + eps);
    }
    __syncthreads();
    GG_GSTAMP(4);
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = tid + 256 * k;
        if (c < C) {
            float sc = 0.f, sh = 0.f;
            if (c < C_logical) {
                const int g = gg_div_small(c, rcpg);
                sc = grstd[g] * gam[k];
                sh = bet[k] - gmean[g] * sc;
            }
            ss[c] = sc;
            ss[C + c] = sh;
        }
    }
    __syncthreads();
    GG_GSTAMP(5);
    auto emit = [&](long long i, const u32x4 raw) {
        const long long row = gg_fastdiv(i, P, pmagic);
        const int c0 = (int)(i - row * P) * 8;
        const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(ss + c0), a1 = *reinterpret_cast<const f32x4 *>(ss + c0 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4 *>(ss + C + c0), h1 = *reinterpret_cast<const f32x4 *>(ss + C + c0 + 4);
        bf16x8 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y0 = (float)v[j] * a0[j] + h0[j];
            float y1 = (float)v[j + 4] * a1[j] + h1[j];
            if (act) { y0 = gg_silu(y0); y1 = gg_silu(y1); }
            y[j] = (bf16_t)y0;
            y[j + 4] = (bf16_t)y1;
        }
        *reinterpret_cast<bf16x8 *>(o + row * C + c0) = y;
    };
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (i0 + u * stride < pieces) emit(i0 + u * stride, pv[u]);
    for (long long i = i0 + U * stride; i < pieces; i += stride) {
        const long long row = gg_fastdiv(i, P, pmagic);
        const int c0 = (int)(i - row * P) * 8;
        emit(i, *reinterpret_cast<const u32x4 *>((c0 >= C1) ? b2 + row * C2 + (c0 - C1) : b1 + row * C1 + c0));
    }
    GG_GSTAMP(6);
}

// Batch-1 latent-UNet form of the above (one 16-byte piece per thread, < 2^31 elements, one sample per grid row): at ~2 ns per wave
// instruction the kernel above spends 0.6 us before its first load (64-bit piece arithmetic) and ~1.5 us in its fold (LDS integer
// atomics, four barriers, a scale / shift table for all channels).  Here: 32-bit piece arithmetic; the thread's piece, the (sum, sumsq)
// of its fold channels and gamma / beta of ITS 8 channels all requested up front; per-channel sums parked in LDS (one barrier), 32
// threads add their group's cpg entries (integers: exact, any order) and derive mean / rstd with the same formulas (second barrier);
// every thread then forms the scale / shift of its own 8 channels from its groups' statistics -- no table, no third barrier.
// Results are bit-identical to gn_apply_acc_kernel (same integer sums, same fp64 / fp32 expressions).
template <int CPT>      // fold channels per thread: C <= 256 * CPT
__global__ __launch_bounds__(256) void gn_apply_acc_lean_kernel(const bf16_t *__restrict__ s1, int C1, const long long *__restrict__ acc1,
                                                                const bf16_t *__restrict__ s2, int C2, const long long *__restrict__ acc2,
                                                                int S, int C_logical, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float eps, int act, bf16_t *__restrict__ out, unsigned pmagic)
{
    s1 = gg_pin(s1); C1 = gg_pin(C1); acc1 = gg_pin(acc1); s2 = gg_pin(s2); C2 = gg_pin(C2); acc2 = gg_pin(acc2); S = gg_pin(S);
    C_logical = gg_pin(C_logical); gamma = gg_pin(gamma); beta = gg_pin(beta); pmagic = gg_pin(pmagic);
    GG_GSTAMP(0);
    typedef __attribute__((ext_vector_type(2))) long long i64x2;
    const int C = C1 + C2, P = C >> 3;
    const int tid = threadIdx.x, n = blockIdx.y;
    const int pieces = S * P;
    extern __shared__ __attribute__((aligned(16))) char lean_smem[];
    i64x2 *csum = reinterpret_cast<i64x2 *>(lean_smem);          // [C_logical] (sum, sumsq), fixed point
    __shared__ float gmean[32], grstd[32];
    // the thread's piece and the affine of its 8 channels
    // every XCD (blockIdx & 7) takes a contiguous eighth of the rows, roughly the rows the position-major box convs on either side of this
    // norm give the same XCD: more of the hand-over stays inside one L2 (captured latent-UNet forward 1.2857 -> 1.2807 ms, three A/B pairs)
    const int bx = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int i = bx * 256 + tid;
    const bool live = i < pieces;
    const int row = live ? (int)__umulhi((unsigned)i, pmagic) : 0;            // i / P (pmagic != 0: host gate)
    const int c0 = live ? (i - row * P) * 8 : 0;
    const bool logical = live && c0 < C_logical;                                // (C_logical % 32 == 0: a piece is all logical or all padding)
    const unsigned e1 = (unsigned)n * (unsigned)S * (unsigned)C1, e2 = (unsigned)n * (unsigned)S * (unsigned)C2;   // < 2^31 elements (host gate)
    u32x4 pv = u32x4{0u, 0u, 0u, 0u};
    if (live) pv = *reinterpret_cast<const u32x4 *>((c0 >= C1) ? s2 + e2 + (unsigned)row * (unsigned)C2 + (unsigned)(c0 - C1) : s1 + e1 + (unsigned)row * (unsigned)C1 + (unsigned)c0);
    f32x4 g0 = f32x4{0.f, 0.f, 0.f, 0.f}, g1 = g0, b0 = g0, b1 = g0;
    if (logical) {
        g0 = *reinterpret_cast<const f32x4 *>(gamma + c0); g1 = *reinterpret_cast<const f32x4 *>(gamma + c0 + 4);
        b0 = *reinterpret_cast<const f32x4 *>(beta + c0); b1 = *reinterpret_cast<const f32x4 *>(beta + c0 + 4);
    }
    // fold channels tid + 256k: (sum, sumsq) over the stripes, parked in LDS
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = tid + 256 * k;
        if (c < C_logical) {
            const long long *q = (c < C1) ? acc1 + ((long long)n * GG_ACC_STRIPES * C1 + c) * 2 : acc2 + ((long long)n * GG_ACC_STRIPES * C2 + (c - C1)) * 2;
            const long long cs = (c < C1) ? (long long)C1 * 2 : (long long)C2 * 2;
            i64x2 a = *reinterpret_cast<const i64x2 *>(q);
#pragma unroll
            for (int st = 1; st < GG_ACC_STRIPES; ++st) a += *reinterpret_cast<const i64x2 *>(q + st * cs);
            csum[c] = a;
        }
    }
    const int cpg = C_logical >> 5;
    GG_GSTAMP(1);
    __syncthreads();
    GG_GSTAMP(2);
    if (tid < 32) {
        i64x2 t = i64x2{0, 0};
        for (int j = 0; j < cpg; ++j) t += csum[tid * cpg + j];
        const double a = (double)t[0] * (1.0 / GG_ACC_SUM_SCALE_D);
        const double b = (double)t[1] * (1.0 / GG_ACC_SQ_SCALE_D);
        const double cnt = (double)S * (double)cpg; double inv = (double)(1.0f / (float)cnt); inv = inv * (2.0 - cnt * inv);      // (as gn_apply_acc_kernel)
        const double mean = a * inv;
        double var = b * inv - mean * mean;
        if (var < 0.0) var = 0.0;
        gmean[tid] = (float)mean;
        grstd[tid] = rsqrtf((float)var + eps);
        GG_GSTAMP(3);
    }
    __syncthreads();
    GG_GSTAMP(4);
    if (!live) return;
    const float rcpg = __builtin_amdgcn_rcpf((float)cpg);
    const bf16x8 v = __builtin_bit_cast(bf16x8, pv);
    bf16x8 y;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float sc = 0.f, sh = 0.f;
        if (logical) {
            const int g = gg_div_small(c0 + j, rcpg);
            sc = grstd[g] * (j < 4 ? g0[j & 3] : g1[j & 3]);
            sh = (j < 4 ? b0[j & 3] : b1[j & 3]) - gmean[g] * sc;
        }
        float t = (float)v[j] * sc + sh;
        if (act) t = gg_silu(t);
        y[j] = (bf16_t)t;
    }
    *reinterpret_cast<bf16x8 *>(out + ((unsigned)n * (unsigned)S + (unsigned)row) * (unsigned)C + (unsigned)c0) = y;
    GG_GSTAMP(5);
}

extern "C" int gg_groupnorm_apply_acc(const void *src1, int32_t C1, const int64_t *acc1, const void *src2, int32_t C2,
                                      const int64_t *acc2, int32_t N, int64_t S, int32_t C_logical, const float *gamma,
                                      const float *beta, float eps, int32_t act, void *out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    const int C = C1 + C2;
    if (C1 <= 0 || C1 % 32 || C2 % 32 || C2 < 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_apply_acc: C1/C2 must be multiples of 32");
    if (C_logical % 32 || C_logical > C || C_logical <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_apply_acc: logical channels %d not divisible by 32 groups", C_logical);
    if (C > 2048) GG_FAIL(GG_ERR_UNSUPPORTED, "groupnorm_apply_acc: C > 2048");
    if (!src1 || !acc1 || (C2 && (!src2 || !acc2)) || !gamma || !beta || !out || N <= 0 || S <= 0)
        GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_apply_acc: null pointer / empty");
    const long long pieces = (long long)S * (C / 8);
    long long blocks = (pieces + 255) / 256;   // one piece per thread (256 / 512 / 1024 pieces per block: 1477 / 1496 / 1545 us per latent-UNet forward: the parallelism is worth more than the per-block fold)
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    // batch-1 latent-UNet shapes: one piece per thread, 32-bit element offsets, multiply-high piece decode
    const unsigned pm = gg_magic_u32(pieces, C / 8);
    if (GG_GN_APPLY_ACC_LEAN && blocks * 256 >= pieces && (long long)N * S * C < (1LL << 31) && (pm != 0 || C == 8) && C / 8 > 1) {
        const size_t lds = (size_t)C_logical * 16;
#define GG_LEAN(CPT) hipLaunchKernelGGL(gn_apply_acc_lean_kernel<CPT>, dim3((unsigned)blocks, N), dim3(256), lds, stream, (const bf16_t *)src1, C1, \
                       (const long long *)acc1, (const bf16_t *)src2, C2, (const long long *)acc2, (int)S, C_logical, gamma, beta, eps, act, (bf16_t *)out, pm)
        if (C_logical <= 256) GG_LEAN(1); else if (C_logical <= 512) GG_LEAN(2); else if (C_logical <= 1024) GG_LEAN(4); else GG_LEAN(8);
#undef GG_LEAN
        GG_CHECK_LAUNCH();
        return GG_OK;
    }
    hipLaunchKernelGGL(gn_apply_acc_kernel, dim3((unsigned)blocks, N), dim3(256), C * 2 * sizeof(float), stream, (const bf16_t *)src1, C1,
                       (const long long *)acc1, (const bf16_t *)src2, C2, (const long long *)acc2, (long long)S, C_logical, gamma, beta, eps,
                       act, (bf16_t *)out, gg_magic_u32(pieces, C / 8));
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// GroupNorm scale / shift from the accumulators the producing convs left (gg_conv_desc.gn_acc), for consumers that apply the
// norm themselves (the halo-tile conv's fused prologue): replaces the statistics PASS over the tensor (268-805 MB per norm at
// the 128^3 CCDM levels) by a fold of N x stripes x C x 2 integers.  One block per (group, sample).
__global__ __launch_bounds__(256) void gn_scale_shift_acc_kernel(const long long *__restrict__ acc1, int stripes1, int C1,
                                                                 const long long *__restrict__ acc2, int stripes2, int C2,
                                                                 long long S, int C_logical, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float eps, float *__restrict__ scale,
                                                                 float *__restrict__ shift)
{
    // block (g, n): the (stripe, channel) pairs of the group are spread over the threads, so the fold is ONE memory round trip
    // (a block per sample walking 32 stripes per channel serially took 11.8 us per norm in the CCDM forward)
    const int C = C1 + C2, g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C_logical / 32;
    __shared__ unsigned long long gacc[2];             // the group's integer (sum, sumsq): LDS atomics, exact in any order
    if (tid < 2) gacc[tid] = 0ull;
    float gmm = 0.f, bta = 0.f;
    if (tid < cpg) { gmm = gamma[g * cpg + tid]; bta = beta[g * cpg + tid]; }
    __syncthreads();
    typedef __attribute__((ext_vector_type(2))) long long i64x2;
    const int smax = stripes1 > stripes2 ? stripes1 : stripes2;
    long long sa = 0, sb = 0;
    const float rcpg = __builtin_amdgcn_rcpf((float)cpg);
    for (int i = tid; i < cpg * smax; i += 256) {
        const int k = gg_div_small(i, rcpg), c = g * cpg + (i - k * cpg);          // (i < 64 * 32: exact; an integer division is ~30 instructions)
        const bool first = c < C1;
        const int st = first ? stripes1 : stripes2, Cs = first ? C1 : C2, cc = first ? c : c - C1;
        if (k < st) {
            const i64x2 v = *reinterpret_cast<const i64x2 *>((first ? acc1 : acc2) + (((long long)n * st + k) * Cs + cc) * 2);
            sa += v[0];
            sb += v[1];
        }
    }
    atomicAdd(&gacc[0], (unsigned long long)sa);
    atomicAdd(&gacc[1], (unsigned long long)sb);
    __syncthreads();
    const double a = (double)(long long)gacc[0] * (1.0 / GG_ACC_SUM_SCALE_D);
    const double b = (double)(long long)gacc[1] * (1.0 / GG_ACC_SQ_SCALE_D);
    // (as gn_apply_acc_kernel: fp32 reciprocal of the exact count + one Newton step in fp64, v_rsq_f32; the fp64 divisions and the fp64
    //  square root were ~150 instructions on the critical path of a 7 us kernel, 27-29 launches per CCDM forward / AE pass)
    const double cnt = (double)S * (double)cpg; double inv = (double)(1.0f / (float)cnt); inv = inv * (2.0 - cnt * inv);
    const double mean = a * inv;
    double var = b * inv - mean * mean;
    if (var < 0.0) var = 0.0;
    const float fmean = (float)mean, frstd = rsqrtf((float)var + eps);
    if (tid < cpg) {
        const int c = g * cpg + tid;
        const float sc = frstd * gmm;
        scale[(long long)n * C + c] = sc;
        shift[(long long)n * C + c] = bta - fmean * sc;
    }
    if (g == 0)   // zero the pad lanes once
        for (int c = C_logical + tid; c < C; c += 256) { scale[(long long)n * C + c] = 0.f; shift[(long long)n * C + c] = 0.f; }
}

extern "C" int gg_groupnorm_scale_shift_acc(const int64_t *acc1, int32_t stripes1, int32_t C1, const int64_t *acc2, int32_t stripes2,
                                            int32_t C2, int32_t N, int64_t S, int32_t C_logical, const float *gamma, const float *beta,
                                            float eps, float *scale_out, float *shift_out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (C1 <= 0 || C1 % 32 || C2 % 32 || C2 < 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_scale_shift_acc: C1/C2 must be multiples of 32");
    if (C_logical <= 0 || C_logical % 32 || C_logical > C1 + C2) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_scale_shift_acc: C_logical must be a multiple of 32 groups");
    if (!acc1 || (C2 && !acc2) || !gamma || !beta || !scale_out || !shift_out || N <= 0 || S <= 0 || stripes1 <= 0 || (C2 && stripes2 <= 0))
        GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_scale_shift_acc: null pointer / empty");
    hipLaunchKernelGGL(gn_scale_shift_acc_kernel, dim3(32, (unsigned)N), dim3(256), 0, stream, (const long long *)acc1, stripes1, C1,
                       (const long long *)acc2, stripes2, C2, (long long)S, C_logical, gamma, beta, eps, scale_out, shift_out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, fp32 two-pass (row cached in registers: C <= 64*8*4 = 2048)
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t *__restrict__ x, long long rows, int C,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float eps, bf16_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int P = C >> 3;
    f32x8 v[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int piece = lane + 64 * k;
        if (piece < P) {
            v[k] = gg_bf16x8_to_f32(*reinterpret_cast<const bf16x8 *>(x + row * C + piece * 8));
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[k][j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int piece = lane + 64 * k;
        if (piece < P) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float d = v[k][j] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int piece = lane + 64 * k;
        if (piece < P) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int c = piece * 8 + j;
                o[j] = (bf16_t)((v[k][j] - mean) * rstd * gamma[c] + beta[c]);
            }
            *reinterpret_cast<bf16x8 *>(out + row * C + piece * 8) = o;
        }
    }
}

extern "C" int gg_layernorm(const void *x, int64_t rows, int32_t C, const float *gamma, const float *beta, float eps,
                            void *out, void *stream_)
{
    if (C % 8 || C > 2048 || C <= 0) GG_FAIL(GG_ERR_UNSUPPORTED, "layernorm: C=%d (need multiple of 8, <= 2048)", C);
    if (!x || !gamma || !beta || !out) GG_FAIL(GG_ERR_BAD_SHAPE, "layernorm: null pointer");
    if (rows <= 0) return GG_OK;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)x,
                       (long long)rows, C, gamma, beta, eps, (bf16_t *)out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geglu_kernel(const bf16_t *__restrict__ h, long long rows, int inner,
                                                    bf16_t *__restrict__ out)
{
    const int P = inner >> 3;
    const long long total = rows * P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i / P;
        int c0 = (int)(i - r * P) * 8;
        bf16x8 a = *reinterpret_cast<const bf16x8 *>(h + r * 2 * inner + c0);
        bf16x8 g = *reinterpret_cast<const bf16x8 *>(h + r * 2 * inner + inner + c0);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gv = (float)g[j];
            float ge = 0.5f * gv * (1.0f + erff(gv * 0.70710678118654752f));   // F.gelu, exact erf form
            o[j] = (bf16_t)((float)a[j] * ge);
        }
        *reinterpret_cast<bf16x8 *>(out + r * inner + c0) = o;
    }
}

extern "C" int gg_geglu(const void *h, int64_t rows, int32_t inner, void *out, void *stream_)
{
    if (inner % 8 || inner <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "geglu: inner %% 8");
    long long total = (long long)rows * (inner / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) return GG_OK;
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)h, (long long)rows,
                       inner, (bf16_t *)out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

__global__ __launch_bounds__(256) void add_kernel(const bf16_t *__restrict__ a, const bf16_t *__restrict__ b, long long n8,
                                                  bf16_t *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        bf16x8 x = *reinterpret_cast<const bf16x8 *>(a + i * 8);
        bf16x8 y = *reinterpret_cast<const bf16x8 *>(b + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)x[j] + (float)y[j]);
        *reinterpret_cast<bf16x8 *>(out + i * 8) = o;
    }
}

extern "C" int gg_add(const void *a, const void *b, int64_t n, void *out, void *stream_)
{
    if (n % 8) GG_FAIL(GG_ERR_BAD_SHAPE, "add: n %% 8");
    long long n8 = n / 8;
    long long blocks = (n8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) return GG_OK;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)a, (const bf16_t *)b,
                       n8, (bf16_t *)out);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
