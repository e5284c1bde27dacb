// fp32 VALIDATION path of the CCDM sampler (include/guidegen_hip.h "fp32 validation mode"): the same network functions as the bf16
// production kernels, on fp32 channels-last tensors with fp32 weights and fp32 FMA accumulation in a fixed order, so that the
// categorical sampler's integer outputs (labels) can be compared EXACTLY with the fp32 CPU reference (north_star: "bit-exact for
// the argmax mask labels"; SURVEY.md section 7, hard part 1).  Speed is not a goal here (a 32^3 CCDM forward of 197 GFLOP takes a
// few tens of milliseconds, a 128^3 one seconds); clarity and a fixed summation order are.  The reference's own precision switch:
// ccdm/ddpm/models/unet_openai/unet.py:447,742-756 (fp32 torso unless use_fp16).
//
//   conv_f32_kernel       nn.Conv{1,2,3}d (stride 1 / 2, zero padding 1 or AE (0,1), fused nearest x2 upsample, skip concat as a second
//                         source, per-sample bias = conv bias + timestep embedding, residual add)   unet.py:188-228,106-139
//   gn_f32_stats / apply  GroupNorm32 (+ SiLU): statistics in fp64, (x - mean) * rstd * gamma + beta in fp32 as ATen evaluates it
//                         nn.py:17-19,93-100
//   attn_f32_kernel       QKVAttentionLegacy: softmax((q * s)(k * s)^T) v with s = ch^-1/4, fp32   unet.py:334-360
#include "gg_common.h"

struct ConvF32 {
    int N, D, H, W, C1, C2, Cout, Cout_pad, kd, kh, kw, stride, pad, upsample, Do, Ho, Wo;
    long long M, bias_stride;
    const float *src1, *src2, *weight, *bias, *residual;
    float *out;
};

// One workgroup: 64 output positions x 32 output channels; thread -> positions (tid & 31), (tid & 31) + 32 and 4 channels.
// Per (tap, 32-channel chunk): the 64 x 32 input tile (zero padding, upsample and concat resolved here) and the 32 x 32 weight tile
// go through LDS; accumulation order = taps outer, input channels inner, one fp32 FMA each: fixed, independent of the grid.
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvF32 p)
{
    __shared__ float xs[32][65];          // [ci][position] (+1: the transposing stores spread over the banks)
    __shared__ float ws[32][32];          // [ci][co]
    const int tid = threadIdx.x;
    const int C = p.C1 + p.C2;
    const long long m0 = (long long)blockIdx.x * 64;
    const int co0 = blockIdx.y * 32;
    // staging duty: position (tid >> 2), channels (tid & 3) * 8 .. + 7
    const int sp = tid >> 2, sc = (tid & 3) * 8;
    const long long ms = m0 + sp;
    int n = 0, od = 0, oh = 0, ow = 0;
    const bool mvalid = ms < p.M;
    if (mvalid) {
        long long r = ms;
        ow = (int)(r % p.Wo); r /= p.Wo;
        oh = (int)(r % p.Ho); r /= p.Ho;
        od = (int)(r % p.Do); r /= p.Do;
        n = (int)r;
    }
    const int pq = tid & 31, cg = tid >> 5;
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // extent of the (virtually upsampled) input per dim; a dim with kernel extent 1 is never upsampled by itself
    const int upD = p.upsample && p.kd == 3, upH = p.upsample, upW = p.upsample;
    const int eD = upD ? 2 * p.D : p.D, eH = upH ? 2 * p.H : p.H, eW = upW ? 2 * p.W : p.W;
    const int ntaps = p.kd * p.kh * p.kw;
    for (int tap = 0; tap < ntaps; ++tap) {
        const int kz = tap / (p.kh * p.kw), ky = (tap / p.kw) % p.kh, kx = tap % p.kw;
        // input coordinate of this thread's staging position for this tap (1-extent dims: stride applies, no padding)
        const int zd = (p.kd == 3) ? od * p.stride - p.pad + kz : od * p.stride;
        const int zh = (p.kh == 3) ? oh * p.stride - p.pad + ky : oh * p.stride;
        const int zw = (p.kw == 3) ? ow * p.stride - p.pad + kx : ow * p.stride;
        const bool inb = mvalid && zd >= 0 && zd < eD && zh >= 0 && zh < eH && zw >= 0 && zw < eW;
        const int id = upD ? zd >> 1 : zd, ih = upH ? zh >> 1 : zh, iw = upW ? zw >> 1 : zw;
        const long long pos = (((long long)n * p.D + id) * p.H + ih) * p.W + iw;
        for (int c0 = 0; c0 < C; c0 += 32) {
            f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (inb) {
                const int c = c0 + sc;                       // 8 channels of one source (C1 is a multiple of 32)
                const float *src = (c < p.C1) ? p.src1 + pos * p.C1 + c : p.src2 + pos * p.C2 + (c - p.C1);
                v0 = *reinterpret_cast<const f32x4 *>(src);
                v1 = *reinterpret_cast<const f32x4 *>(src + 4);
            }
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(p.weight + ((long long)tap * C + c0 + (tid >> 3)) * p.Cout_pad + co0 + (tid & 7) * 4);
            __syncthreads();                                 // the previous tile has been consumed
#pragma unroll
            for (int j = 0; j < 4; ++j) { xs[sc + j][sp] = v0[j]; xs[sc + 4 + j][sp] = v1[j]; }
            *reinterpret_cast<f32x4 *>(&ws[tid >> 3][(tid & 7) * 4]) = wv;
            __syncthreads();
#pragma unroll 8
            for (int ci = 0; ci < 32; ++ci) {
                const float x0 = xs[ci][pq], x1 = xs[ci][pq + 32];
                const f32x4 w = *reinterpret_cast<const f32x4 *>(&ws[ci][cg * 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][j] = __builtin_fmaf(x0, w[j], acc[0][j]);
                    acc[1][j] = __builtin_fmaf(x1, w[j], acc[1][j]);
                }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long m = m0 + pq + 32 * h;
        if (m >= p.M) continue;
        const long long osp = (long long)p.Do * p.Ho * p.Wo;
        const int nn = (int)(m / osp);
        const int co = co0 + cg * 4;
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[h][j];
            if (p.bias) v += p.bias[(long long)nn * p.bias_stride + co + j];
            if (p.residual) v += p.residual[m * p.Cout_pad + co + j];
            r[j] = (co + j < p.Cout) ? v : 0.f;
        }
        *reinterpret_cast<f32x4 *>(p.out + m * p.Cout_pad + co) = r;
    }
}

extern "C" int gg_conv_forward_f32(const gg_conv_desc *d, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d) GG_FAIL(GG_ERR_BAD_SHAPE, "conv_f32: null desc");
    if (d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->C2 < 0) GG_FAIL(GG_ERR_BAD_SHAPE, "conv_f32: C1/C2 must be multiples of 32 (got %d, %d)", d->C1, d->C2);
    if (d->Cout_pad % 32 || d->Cout > d->Cout_pad || d->Cout <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "conv_f32: bad Cout/Cout_pad %d/%d", d->Cout, d->Cout_pad);
    auto okk = [](int k) { return k == 1 || k == 3; };
    if (!okk(d->kd) || !okk(d->kh) || !okk(d->kw)) GG_FAIL(GG_ERR_UNSUPPORTED, "conv_f32: kernel extent must be 1 or 3");
    if ((d->stride != 1 && d->stride != 2) || (d->upsample && d->stride != 1)) GG_FAIL(GG_ERR_UNSUPPORTED, "conv_f32: stride must be 1 or 2, no stride with upsample");
    if (d->out_dtype != GG_F32) GG_FAIL(GG_ERR_BAD_DTYPE, "conv_f32: the validation path stores fp32");
    if (d->prologue_act || d->gn_acc || d->ddim_x || d->epilogue_geglu)
        GG_FAIL(GG_ERR_UNSUPPORTED, "conv_f32: no fused prologue / statistics / DDIM / GEGLU on the validation path (separate fp32 launches)");
    if (!d->src1 || !d->weight || !d->out || (d->C2 && !d->src2)) GG_FAIL(GG_ERR_BAD_SHAPE, "conv_f32: null pointer");
    if (d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Do <= 0 || d->Ho <= 0 || d->Wo <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "conv_f32: empty extent");
    auto expect = [&](int in, int k, int up) {
        int e = up ? in * 2 : in;
        if (k == 1) return (d->upsample && !up) ? e : (e - 1) / d->stride + 1;
        return (d->stride == 1) ? e + 2 * d->pad - 2 : (d->pad == 1 ? (e + 2 - 3) / 2 + 1 : (e + 1 - 3) / 2 + 1);
    };
    const int upD = d->upsample && d->kd == 3, upHW = d->upsample;
    if (d->Do != expect(d->D, d->kd, upD) || d->Ho != expect(d->H, d->kh, upHW) || d->Wo != expect(d->W, d->kw, upHW))
        GG_FAIL(GG_ERR_BAD_SHAPE, "conv_f32: output extent (%d,%d,%d) inconsistent with input (%d,%d,%d)", d->Do, d->Ho, d->Wo, d->D, d->H, d->W);
    ConvF32 p;
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.C1 = d->C1; p.C2 = d->C2; p.Cout = d->Cout; p.Cout_pad = d->Cout_pad;
    p.kd = d->kd; p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.upsample = d->upsample;
    p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo;
    p.M = (long long)d->N * d->Do * d->Ho * d->Wo;
    p.bias_stride = d->bias_stride;
    p.src1 = (const float *)d->src1; p.src2 = (const float *)d->src2; p.weight = (const float *)d->weight;
    p.bias = d->bias; p.residual = (const float *)d->residual; p.out = (float *)d->out;
    const long long mb = (p.M + 63) / 64;
    if (mb >= (1LL << 31)) GG_FAIL(GG_ERR_UNSUPPORTED, "conv_f32: too many positions");
    hipLaunchKernelGGL(conv_f32_kernel, dim3((unsigned)mb, (unsigned)(p.Cout_pad / 32)), dim3(256), 0, stream, p);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ---------------------------------------------------------------------------------------------------------------- GroupNorm
// One block per (group, sample): fp64 sum / sum of squares over the group's S x cpg elements (both sources of a concat).
__global__ __launch_bounds__(256) void gn_f32_stats_kernel(const float *__restrict__ s1, int C1, const float *__restrict__ s2, int C2, long long S,
                                                           int C_logical, float eps, float *__restrict__ mean_out, float *__restrict__ rstd_out)
{
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C_logical / 32;
    double a = 0.0, b = 0.0;
    const long long total = S * cpg;
    for (long long i = tid; i < total; i += 256) {
        const long long row = i / cpg;
        const int c = g * cpg + (int)(i - row * cpg);
        const float v = (c < C1) ? s1[((long long)n * S + row) * C1 + c] : s2[((long long)n * S + row) * C2 + (c - C1)];
        a += (double)v;
        b += (double)v * (double)v;
    }
    __shared__ double ra[256], rb[256];
    ra[tid] = a; rb[tid] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {          // fixed tree: deterministic
        if (tid < s) { ra[tid] += ra[tid + s]; rb[tid] += rb[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)total, mean = ra[0] / cnt;
        double var = rb[0] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_out[n * 32 + g] = (float)mean;
        rstd_out[n * 32 + g] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ __launch_bounds__(256) void gn_f32_apply_kernel(const float *__restrict__ s1, int C1, const float *__restrict__ s2, int C2, long long S,
                                                           int C_logical, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd, int act,
                                                           float *__restrict__ out, long long total)
{
    const int C = C1 + C2, cpg = C_logical / 32;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / C;               // n * S + position
        const int c = (int)(i - row * C);
        float y = 0.f;
        if (c < C_logical) {
            const int n = (int)(row / S), g = c / cpg;
            const float v = (c < C1) ? s1[row * C1 + c] : s2[row * C2 + (c - C1)];
            {   // ATen's order: (x - mean) * rstd * gamma + beta, every step rounded to fp32
#pragma clang fp contract(off)
                y = (v - mean[n * 32 + g]) * rstd[n * 32 + g];
                y = y * gamma[c] + beta[c];
            }
            if (act) y = y / (1.0f + expf(-y));     // x * sigmoid(x), IEEE division, full-precision expf
        }
        out[i] = y;
    }
}

extern "C" int gg_groupnorm_f32(const float *src1, int32_t C1, const float *src2, int32_t C2, int32_t N, int64_t S, int32_t C_logical,
                                const float *gamma, const float *beta, float eps, int32_t act, float *out, float *workspace, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (C1 <= 0 || C1 % 32 || C2 % 32 || C2 < 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_f32: C1/C2 must be multiples of 32");
    if (C_logical <= 0 || C_logical % 32 || C_logical > C1 + C2) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_f32: logical channels %d not divisible into 32 groups", C_logical);
    if (C2 && C_logical < C1) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_f32: two sources need an unpadded first source");
    if (!src1 || (C2 && !src2) || !gamma || !beta || !out || !workspace || N <= 0 || S <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "groupnorm_f32: null pointer / empty");
    float *mean = workspace, *rstd = workspace + (size_t)N * 32;
    hipLaunchKernelGGL(gn_f32_stats_kernel, dim3(32, (unsigned)N), dim3(256), 0, stream, src1, C1, src2, C2, (long long)S, C_logical, eps, mean, rstd);
    GG_CHECK_LAUNCH();
    const long long total = (long long)N * S * (C1 + C2);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gn_f32_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src1, C1, src2, C2, (long long)S, C_logical, gamma, beta,
                       (const float *)mean, (const float *)rstd, act, out, total);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// ---------------------------------------------------------------------------------------------------------------- attention
// One thread per query, online softmax in fp32; q and k are each multiplied by a = sqrt(scale) first, as QKVAttentionLegacy does
// with a = ch^-1/4 (unet.py:349-354).  All threads of a wave read the same key / value row: broadcast loads.
template <int DMAX>
__global__ __launch_bounds__(64) void attn_f32_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                      float *__restrict__ out, int heads, int D, int Tq, int Tkv, long long ldq, long long hsq,
                                                      long long ldk, long long hsk, long long ldv, long long hsv, long long ldo, long long hso, float a)
{
    const int t = blockIdx.x * 64 + threadIdx.x, h = blockIdx.y, n = blockIdx.z;
    if (t >= Tq) return;
    float qr[DMAX], o[DMAX];
    const float *qp = q + ((long long)n * Tq + t) * ldq + h * hsq;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) { qr[d] = d < D ? qp[d] * a : 0.f; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < Tkv; ++j) {
        const float *kp = k + ((long long)n * Tkv + j) * ldk + h * hsk;
        const float *vp = v + ((long long)n * Tkv + j) * ldv + h * hsv;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
            if (d < D) s = __builtin_fmaf(qr[d], kp[d] * a, s);
        const float mn = fmaxf(m, s);
        const float corr = expf(m - mn), e = expf(s - mn);
        l = l * corr + e;
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
            if (d < D) o[d] = o[d] * corr + e * vp[d];
        m = mn;
    }
    float *op = out + ((long long)n * Tq + t) * ldo + h * hso;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
        if (d < D) op[d] = o[d] / l;
}

extern "C" int gg_attention_forward_f32(const gg_attention_desc *d, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || !d->q || !d->k || !d->v || !d->out) GG_FAIL(GG_ERR_BAD_SHAPE, "attention_f32: null pointer");
    if (d->N <= 0 || d->heads <= 0 || d->Tq <= 0 || d->Tkv <= 0 || d->head_dim <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "attention_f32: empty");
    if (d->head_dim > 64) GG_FAIL(GG_ERR_UNSUPPORTED, "attention_f32: head_dim <= 64 on the validation path (CCDM uses 32)");
    const float a = sqrtf(d->scale);
    dim3 grid((unsigned)((d->Tq + 63) / 64), (unsigned)d->heads, (unsigned)d->N);
    if (d->head_dim <= 32)
        hipLaunchKernelGGL(attn_f32_kernel<32>, grid, dim3(64), 0, stream, (const float *)d->q, (const float *)d->k, (const float *)d->v, (float *)d->out,
                           d->heads, d->head_dim, d->Tq, d->Tkv, d->ldq, d->hsq, d->ldk, d->hsk, d->ldv, d->hsv, d->ldo, d->hso, a);
    else
        hipLaunchKernelGGL(attn_f32_kernel<64>, grid, dim3(64), 0, stream, (const float *)d->q, (const float *)d->k, (const float *)d->v, (float *)d->out,
                           d->heads, d->head_dim, d->Tq, d->Tkv, d->ldq, d->hsq, d->ldk, d->hsk, d->ldv, d->hsv, d->ldo, d->hso, a);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
