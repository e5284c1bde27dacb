// Implicit-GEMM convolution on MFMA for gfx950 (bf16 in, fp32 accumulate).
//
// GEMM view (operands swapped so that each lane ends up holding 4 consecutive output CHANNELS of one
// output position -> 8-byte bf16 stores along the contiguous channel axis):
//     D[co, m] = sum_{tap, ci}  Wp[co, tap, ci] * X[pos(m) + tap, ci]
//   MFMA A operand  = weight rows  (row = co,  k = ci)   16 B per lane from the packed weight tile
//   MFMA B operand  = activation   (col = m,   k = ci)   16 B per lane from the gathered position row
//
// Two kernels share the MFMA/epilogue code:
//   conv_gather  : generic. Rows of the activation tile are gathered per (tap, 32-channel chunk) from HBM/L2
//                  with bounds-checked 16-byte loads (zero padding, stride 2, fused nearest x2 upsample,
//                  two-source concat, optional GroupNorm*SiLU prologue), staged through registers into a
//                  swizzled, double-buffered LDS tile.
//   conv_halo    : fast path for 3x3(x3) stride-1 convs on large extents (see gg_conv_halo.hip).
#include "gg_conv.h"
#include <stdlib.h>

template <int NT, int PF, int PRO>
__global__ __launch_bounds__(256) void conv_gather_kernel(const ConvParams p)
{
    constexpr int BM = 128;
    constexpr int XBYTES = BM * 64;
    constexpr int WBYTES = NT * 32 * 64;
    constexpr int STAGE = XBYTES + WBYTES;
    constexpr int WITER = (NT * 128 + 255) / 256;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const long long m0 = (long long)blockIdx.x * BM;
    const int g0 = blockIdx.y * NT;

    // ---- per-thread gather duties: rows r and r+64, 16-byte piece q of the 64-byte channel chunk.
    // All per-k-step address arithmetic is 32-bit (in-sample element offsets); the 64-bit part is a per-row sample base.
    const int xq = tid & 3;
    int bn[2], bd[2], bh[2], bw[2];
    bool rv[2];
    const bf16_t *rb1[2], *rb2[2];
    const unsigned osp = (unsigned)(p.Do * p.Ho * p.Wo);
    const unsigned ohw = (unsigned)(p.Ho * p.Wo);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long m = m0 + (tid >> 2) + 64 * i;
        rv[i] = m < p.M;
        unsigned mu = rv[i] ? (unsigned)m : 0u;              // host guarantees M < 2^31
        unsigned n = mu / osp;
        unsigned r = mu - n * osp;
        unsigned od = r / ohw;
        r -= od * ohw;
        unsigned oh = r / (unsigned)p.Wo;
        unsigned ow = r - oh * (unsigned)p.Wo;
        bn[i] = (int)n;
        // coordinates in the (possibly upsampled) input frame of tap (0,0,0)
        bd[i] = (p.kd == 1) ? (int)od * p.stride : (int)od * p.stride - p.pad;
        bh[i] = (p.kh == 1) ? (int)oh * p.stride : (int)oh * p.stride - p.pad;
        bw[i] = (p.kw == 1) ? (int)ow * p.stride : (int)ow * p.stride - p.pad;
        const long long sp = (long long)p.D * p.H * p.W;
        rb1[i] = p.src1 + (long long)n * sp * p.C1 + xq * 8;
        rb2[i] = p.src2 ? p.src2 + (long long)n * sp * p.C2 + xq * 8 : nullptr;
    }
    const int upD = (p.upsample && p.kd == 3) ? 1 : 0;  // 2-D convs never upsample the dummy D axis
    const int upHW = p.upsample ? 1 : 0;
    const unsigned limD = (unsigned)(p.D << upD), limH = (unsigned)(p.H << upHW), limW = (unsigned)(p.W << upHW);
    const int gnC = p.C1 + p.C2;

    // register prefetch ring: PF k-steps of global loads are in flight (the loop is otherwise one load latency per k-step)
    u32x4 xreg[PF][2];
    u32x4 wreg[PF][WITER];
    int xso[PF][2];             // GroupNorm scale/shift offset of the piece, -1: padding / out of range (stays zero)

    // split-K: blockIdx.z owns k-steps [ks_begin, ks_end) of the (chunk outer, tap inner) sequence
    const int KS_all = p.ntaps * p.nchunk;
    const int ks_begin = (int)(((long long)KS_all * blockIdx.z) / p.splitk);
    const int ks_end = (int)(((long long)KS_all * (blockIdx.z + 1)) / p.splitk);
    int chunk = ks_begin / p.ntaps, tap = ks_begin - (ks_begin / p.ntaps) * p.ntaps;   // counters for the NEXT global load
    int tkd = tap / (p.kh * p.kw), tkh = (tap / p.kw) % p.kh, tkw = tap % p.kw;

    // weight tile offset advances incrementally with the (chunk outer, tap inner) walk
    long long woff = ((((long long)g0 * p.ntaps + tap) * p.nchunk + chunk) << 10);
    const long long wstep_tap = (long long)p.nchunk << 10;
    const long long wstep_wrap = (1LL << 10) - (long long)p.ntaps * wstep_tap;          // tap wraps to 0, chunk + 1
    const long long wgroup = ((long long)p.ntaps * p.nchunk) << 10;

    auto load_regs = [&](u32x4 (&xr)[2], u32x4 (&wr)[WITER], int (&so)[2]) {
        const bool second = chunk >= p.nchunk1;
        const int Cs = second ? p.C2 : p.C1;
        const int coff = (second ? chunk - p.nchunk1 : chunk) * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned ud = (unsigned)(bd[i] + tkd), uh = (unsigned)(bh[i] + tkh), uw = (unsigned)(bw[i] + tkw);
            const bool ok = rv[i] && ud < limD && uh < limH && uw < limW;         // unsigned compare also rejects negatives
            u32x4 v = {0u, 0u, 0u, 0u};
            so[i] = -1;
            if (ok) {
                const unsigned pos = (((ud >> upD) * (unsigned)p.H + (uh >> upHW)) * (unsigned)p.W + (uw >> upHW));
                const bf16_t *base = second ? rb2[i] : rb1[i];
                v = *reinterpret_cast<const u32x4 *>(base + (pos * (unsigned)Cs + (unsigned)coff));
                so[i] = bn[i] * gnC + chunk * 32 + xq * 8;
            }
            xr[i] = v;
        }
#pragma unroll
        for (int j = 0; j < WITER; ++j) {
            int i = tid + 256 * j;
            if (i < NT * 128) wr[j] = *reinterpret_cast<const u32x4 *>(p.weight + woff + (long long)(i >> 7) * wgroup + (i & 127) * 8);
        }
        // advance (chunk outer, tap inner)
        ++tap;
        woff += wstep_tap;
        if (++tkw == p.kw) {
            tkw = 0;
            if (++tkh == p.kh) {
                tkh = 0;
                if (++tkd == p.kd) { tkd = 0; tap = 0; ++chunk; woff += wstep_wrap; }
            }
        }
    };

    auto write_lds = [&](int buf, u32x4 (&xr)[2], u32x4 (&wr)[WITER], int (&so)[2]) {
        char *xb = smem + buf * STAGE;
        char *wb = xb + XBYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r = (tid >> 2) + 64 * i;
            u32x4 v = xr[i];
            if (PRO && so[i] >= 0) {   // fused GroupNorm(*SiLU): y = act(x*scale + shift); zero padding stays zero
                f32x4 s0 = *reinterpret_cast<const f32x4 *>(p.gn_scale + so[i]);
                f32x4 s1 = *reinterpret_cast<const f32x4 *>(p.gn_scale + so[i] + 4);
                f32x4 h0 = *reinterpret_cast<const f32x4 *>(p.gn_shift + so[i]);
                f32x4 h1 = *reinterpret_cast<const f32x4 *>(p.gn_shift + so[i] + 4);
                bf16x8 xb8 = __builtin_bit_cast(bf16x8, v);
                bf16x8 yb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y0 = (float)xb8[j] * s0[j] + h0[j], y1 = (float)xb8[j + 4] * s1[j] + h1[j];
                    if (p.prologue_act == 1) {
                        y0 = y0 * __builtin_amdgcn_rcpf(1.0f + __expf(-y0));
                        y1 = y1 * __builtin_amdgcn_rcpf(1.0f + __expf(-y1));
                    }
                    yb[j] = (bf16_t)y0;
                    yb[j + 4] = (bf16_t)y1;
                }
                v = __builtin_bit_cast(u32x4, yb);
            }
            *reinterpret_cast<u32x4 *>(xb + r * 64 + swz64(r, xq) * 16) = v;
        }
#pragma unroll
        for (int j = 0; j < WITER; ++j) {
            int i = tid + 256 * j;
            if (i < NT * 128) *reinterpret_cast<u32x4 *>(wb + i * 16) = wr[j];   // image pre-swizzled at pack time
        }
    };

    f32x4 acc[2][2 * NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2 * NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
        const char *xb = smem + buf * STAGE;
        const char *wb = xb + XBYTES;
        bf16x8 xf[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            int r = wave * 32 + pt * 16 + fr;
            xf[pt] = *reinterpret_cast<const bf16x8 *>(xb + r * 64 + swz64(r, fq) * 16);
        }
#pragma unroll
        for (int ct = 0; ct < 2 * NT; ++ct) {
            int r = ct * 16 + fr;
            bf16x8 wf = *reinterpret_cast<const bf16x8 *>(wb + r * 64 + swz64(r, fq) * 16);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
                acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[pt], acc[pt][ct], 0, 0, 0);
        }
    };

    const int KS = ks_end - ks_begin;
#pragma unroll
    for (int j = 0; j < PF; ++j)
        if (j < KS) load_regs(xreg[j], wreg[j], xso[j]);
    for (int ks0 = 0; ks0 < KS; ks0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int ks = ks0 + j;
            if (ks < KS) {
                const int buf = ks & 1;
                write_lds(buf, xreg[j], wreg[j], xso[j]);       // waits only for slot j's loads (counted vmcnt)
                __syncthreads();                                 // one barrier per k-step: a wave that passed barrier(ks+1)
                if (ks + PF < KS) load_regs(xreg[j], wreg[j], xso[j]);   // knows every wave finished compute(ks) => buf reuse at ks+2 is safe
                compute(buf);
            }
        }
    }

    if (p.splitk > 1) {
        // raw fp32 partial tile -> this block's slab
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            long long m = m0 + wave * 32 + pt * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int ct = 0; ct < 2 * NT; ++ct) {
                int co = (g0 * 32) + ct * 16 + fq * 4;
                *reinterpret_cast<f32x4 *>(p.ws + ((long long)blockIdx.z * p.M + m) * p.Cout_pad + co) = acc[pt][ct];
            }
        }
        // combine + epilogue: conv_splitk_reduce_kernel (an in-launch combine by the last-arriving K slice was measured slower:
        // 3.66 vs 2.70 ms per latent-UNet forward, an agent-scope release per K-slice block costs more than the reduce launch)
        return;
    }
    if (p.epi_geglu) {
        // ---- GEGLU epilogue (gg_conv_desc.epilogue_geglu): cout tile 2k holds 16 value channels, tile 2k+1 their gates, so a lane has
        //      both halves of its 4 output channels in registers; out[m, j] = value * gelu_erf(gate), inner = Cout / 2 channels
        const int ostride = p.Cout_pad >> 1, inner = p.Cout >> 1;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const long long m = m0 + wave * 32 + pt * 16 + fr;
            if (m >= p.M) continue;
            const int n = (int)((unsigned)m / osp);
            const float *brow = p.bias ? p.bias + (long long)n * p.bias_stride : nullptr;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int co = (g0 * 32) + k * 32 + fq * 4;            // value rows co.., gate rows co + 16..
                f32x4 v = acc[pt][2 * k], g = acc[pt][2 * k + 1];
                if (brow) {
                    v += *reinterpret_cast<const f32x4 *>(brow + co);
                    g += *reinterpret_cast<const f32x4 *>(brow + co + 16);
                }
                const int oc = (co >> 5) * 16 + fq * 4;                 // output channel of the lane's first element
                bf16x4 ob;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ge = 0.5f * g[j] * (1.0f + erff(g[j] * 0.70710678118654752f));   // F.gelu, exact erf form (as gg_geglu)
                    ob[j] = (bf16_t)((oc + j < inner) ? v[j] * ge : 0.f);
                }
                *reinterpret_cast<bf16x4 *>((bf16_t *)p.out + m * ostride + oc) = ob;
            }
        }
        return;
    }
    // ---- epilogue: + bias[n] (+ residual) -> bf16 / fp32, 4 consecutive channels per lane.  Every bias / residual load stands in front of
    // the first store: gfx950 counts loads and stores in one in-order counter, so a load behind a store waits for the store's round trip
    f32x4 bv[2][2 * NT];
    bf16x4 resv[2][2 * NT];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const long long m = m0 + wave * 32 + pt * 16 + fr;
        const long long mc = m < p.M ? m : p.M - 1;                 // (rows beyond M: a valid address, never stored)
        const int n = (int)((unsigned)mc / osp);
        const float *brow = p.bias ? p.bias + (long long)n * p.bias_stride : nullptr;
#pragma unroll
        for (int ct = 0; ct < 2 * NT; ++ct) {
            const int co = (g0 * 32) + ct * 16 + fq * 4;
            bv[pt][ct] = brow ? *reinterpret_cast<const f32x4 *>(brow + co) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.residual) resv[pt][ct] = *reinterpret_cast<const bf16x4 *>(p.residual + mc * p.Cout_pad + co);
        }
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        long long m = m0 + wave * 32 + pt * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int ct = 0; ct < 2 * NT; ++ct) {
            int co = (g0 * 32) + ct * 16 + fq * 4;
            f32x4 v = acc[pt][ct];
            if (p.bias) v += bv[pt][ct];
            long long o = m * p.Cout_pad + co;
            if (p.residual) {
                const bf16x4 r = resv[pt][ct];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (co + j >= p.Cout) v[j] = 0.f;
            if (p.out_dtype == GG_F32) {
                *reinterpret_cast<f32x4 *>((float *)p.out + o) = v;
            } else {
                bf16x4 ob;
#pragma unroll
                for (int j = 0; j < 4; ++j) ob[j] = (bf16_t)v[j];
                *reinterpret_cast<bf16x4 *>((bf16_t *)p.out + o) = ob;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ 160-channel-step gather
// Variant for channel counts that are multiples of 160 (the LDM UNet: 160/320/480/.../1600): one k-step covers FIVE
// 32-channel chunks of one tap.  A thread gathers 5 x 16 B of ONE position row per step (one bounds check / address), the
// weight tile of a step is 10 KiB contiguous in the packed layout, and the per-k-step overhead (barrier, loop, waits,
// address math: ~110 instructions in the 32-channel kernel) is paid once per 160 channels.  Tile 64 positions x 32 couts.
template <int PRO>
__global__ __launch_bounds__(256) void conv_gather5_kernel(const ConvParams p)
{
    constexpr int BM = 64, CPS = 5;
    constexpr int XT = BM * 64, WT = 32 * 64;                       // one 32-channel tile of X / W in LDS
    constexpr int STAGE = CPS * (XT + WT);                           // 30 KiB
    constexpr int GNMAX = 2560;                                      // max C1+C2 with a fused prologue (scale+shift staged in LDS)
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + (PRO ? 2 * GNMAX * 4 : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const long long m0 = (long long)blockIdx.x * BM;
    const int g0 = blockIdx.y;
    float *gns = reinterpret_cast<float *>(smem + 2 * STAGE);       // [C] scale then [C] shift of this block's sample

    // gather duty: row tid>>2, 16-byte slot tid&3 of each of the 5 chunks
    const int xrow = tid >> 2, xq = tid & 3;
    const unsigned osp = (unsigned)(p.Do * p.Ho * p.Wo), ohw = (unsigned)(p.Ho * p.Wo);
    bool rv;
    int bd, bh, bw;
    const bf16_t *rb1, *rb2;
    {
        long long m = m0 + xrow;
        rv = m < p.M;
        unsigned mu = rv ? (unsigned)m : 0u;
        unsigned n = (unsigned)gg_fastdiv(mu, (int)osp, p.mg_osp), r = mu - n * osp;
        unsigned od = (unsigned)gg_fastdiv(r, (int)ohw, p.mg_ohw);
        r -= od * ohw;
        unsigned oh = (unsigned)gg_fastdiv(r, p.Wo, p.mg_wo), ow = r - oh * (unsigned)p.Wo;
        bd = (p.kd == 1) ? (int)od * p.stride : (int)od * p.stride - p.pad;
        bh = (p.kh == 1) ? (int)oh * p.stride : (int)oh * p.stride - p.pad;
        bw = (p.kw == 1) ? (int)ow * p.stride : (int)ow * p.stride - p.pad;
        const long long sp = (long long)p.D * p.H * p.W;
        rb1 = p.src1 + (long long)n * sp * p.C1 + xq * 8;
        rb2 = p.src2 ? p.src2 + (long long)n * sp * p.C2 + xq * 8 : nullptr;
    }
    const int upD = (p.upsample && p.kd == 3) ? 1 : 0, upHW = p.upsample ? 1 : 0;
    const unsigned limD = (unsigned)(p.D << upD), limH = (unsigned)(p.H << upHW), limW = (unsigned)(p.W << upHW);
    const int gnC = p.C1 + p.C2;

    if (PRO) {   // host guarantees: all rows of a block belong to one sample (Do*Ho*Wo % 64 == 0) and C1+C2 <= GNMAX
        const int nblk = (int)((unsigned)m0 / osp);
        for (int i = tid; i < gnC; i += 256) {
            gns[i] = p.gn_scale[(long long)nblk * gnC + i];
            gns[GNMAX + i] = p.gn_shift[(long long)nblk * gnC + i];
        }
        __syncthreads();
    }
    // k-steps: (group of 5 chunks) outer, tap inner
    const int ngrp = p.nchunk / CPS, ngrp1 = p.nchunk1 / CPS;
    const int KS_all = ngrp * p.ntaps;
    // (the common cases decode without a division: a wave issues its instructions one by one, and the ten divisions of the general
    // decode were ~300 of the ~650 instructions in front of this kernel's first load)
    int ks_begin = 0, ks_end = KS_all;
    if (p.splitk != 1) {
        ks_begin = (int)(((long long)KS_all * blockIdx.z) / p.splitk);
        ks_end = (int)(((long long)KS_all * (blockIdx.z + 1)) / p.splitk);
    }
    int grp = ks_begin, tap = 0, tkd = 0, tkh = 0, tkw = 0;
    if (p.ntaps != 1) {
        grp = ks_begin / p.ntaps; tap = ks_begin - grp * p.ntaps;
        tkd = tap / (p.kh * p.kw); tkh = (tap / p.kw) % p.kh; tkw = tap % p.kw;
    }

    u32x4 xreg[2][CPS], wreg[2][3];
    int xso[2];
    auto load_regs = [&](u32x4 (&xr)[CPS], u32x4 (&wr)[3], int &so) {
        const bool second = grp >= ngrp1;
        const int Cs = second ? p.C2 : p.C1;
        const int coff = (second ? grp - ngrp1 : grp) * (CPS * 32);
        const unsigned ud = (unsigned)(bd + tkd), uh = (unsigned)(bh + tkh), uw = (unsigned)(bw + tkw);
        const bool ok = rv && ud < limD && uh < limH && uw < limW;
        so = -1;
#pragma unroll
        for (int c = 0; c < CPS; ++c) xr[c] = u32x4{0u, 0u, 0u, 0u};
        if (ok) {
            const unsigned pos = ((ud >> upD) * (unsigned)p.H + (uh >> upHW)) * (unsigned)p.W + (uw >> upHW);
            const bf16_t *src = (second ? rb2 : rb1) + (pos * (unsigned)Cs + (unsigned)coff);
#pragma unroll
            for (int c = 0; c < CPS; ++c) xr[c] = *reinterpret_cast<const u32x4 *>(src + c * 32);
            so = grp * (CPS * 32) + xq * 8;
        }
        // 5 consecutive packed tiles (chunks 5*grp..5*grp+4 of this tap): 10 KiB contiguous
        const bf16_t *wsrc = p.weight + ((((long long)g0 * p.ntaps + tap) * p.nchunk + grp * CPS) << 10);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + 256 * j;
            if (i < CPS * 128) wr[j] = *reinterpret_cast<const u32x4 *>(wsrc + i * 8);
        }
        if (++tkw == p.kw) {
            tkw = 0;
            if (++tkh == p.kh) {
                tkh = 0;
                if (++tkd == p.kd) { tkd = 0; tap = -1; ++grp; }
            }
        }
        ++tap;
    };
    auto write_lds = [&](int buf, u32x4 (&xr)[CPS], u32x4 (&wr)[3], int so) {
        char *xb = smem + buf * STAGE;
        char *wb = xb + CPS * XT;
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
            u32x4 v = xr[c];
            if (PRO && so >= 0) {
                const float *sc = gns + so + c * 32, *sh = gns + GNMAX + so + c * 32;
                f32x4 s0 = *reinterpret_cast<const f32x4 *>(sc), s1 = *reinterpret_cast<const f32x4 *>(sc + 4);
                f32x4 h0 = *reinterpret_cast<const f32x4 *>(sh), h1 = *reinterpret_cast<const f32x4 *>(sh + 4);
                bf16x8 xb8 = __builtin_bit_cast(bf16x8, v), yb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float y0 = (float)xb8[j] * s0[j] + h0[j], y1 = (float)xb8[j + 4] * s1[j] + h1[j];
                    if (p.prologue_act == 1) {
                        y0 = y0 * __builtin_amdgcn_rcpf(1.0f + __expf(-y0));
                        y1 = y1 * __builtin_amdgcn_rcpf(1.0f + __expf(-y1));
                    }
                    yb[j] = (bf16_t)y0;
                    yb[j + 4] = (bf16_t)y1;
                }
                v = __builtin_bit_cast(u32x4, yb);
            }
            *reinterpret_cast<u32x4 *>(xb + c * XT + xrow * 64 + swz64(xrow, xq) * 16) = v;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + 256 * j;
            if (i < CPS * 128) *reinterpret_cast<u32x4 *>(wb + i * 16) = wr[j];     // 5 pre-swizzled 2 KiB tiles, linear
        }
    };

    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto compute = [&](int buf) {
        const char *xb = smem + buf * STAGE;
        const char *wb = xb + CPS * XT;
        const int r = wave * 16 + fr;
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8 *>(xb + c * XT + r * 64 + swz64(r, fq) * 16);
            const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(wb + c * WT + fr * 64 + swz64(fr, fq) * 16);
            const bf16x8 w1 = *reinterpret_cast<const bf16x8 *>(wb + c * WT + (16 + fr) * 64 + swz64(16 + fr, fq) * 16);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xf, acc[1], 0, 0, 0);
        }
    };

    // the epilogue's bias and residual values are requested here, a whole k-loop ahead of their use (an exposed round trip at the end
    // of a 5-10 us kernel otherwise)
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2e;
    f32x4 bias_pf[2];
    u32x2e res_pf[2];
    {
        const long long mp = m0 + wave * 16 + fr;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int co = g0 * 32 + ct * 16 + fq * 4;
            bias_pf[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            res_pf[ct] = u32x2e{0u, 0u};
            if (mp < p.M && p.splitk == 1) {
                if (p.bias) bias_pf[ct] = *reinterpret_cast<const f32x4 *>(p.bias + (long long)((unsigned)mp / osp) * p.bias_stride + co);
                if (p.residual) res_pf[ct] = *reinterpret_cast<const u32x2e *>(p.residual + mp * p.Cout_pad + co);
            }
        }
    }
    const int KS = ks_end - ks_begin;
    if (KS > 0) load_regs(xreg[0], wreg[0], xso[0]);
    if (KS > 1) load_regs(xreg[1], wreg[1], xso[1]);
    for (int ks0 = 0; ks0 < KS; ks0 += 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ks = ks0 + j;
            if (ks < KS) {
                const int buf = ks & 1;
                write_lds(buf, xreg[j], wreg[j], xso[j]);
                __syncthreads();
                if (ks + 2 < KS) load_regs(xreg[j], wreg[j], xso[j]);
                compute(buf);
            }
        }
    }

    const long long m = m0 + wave * 16 + fr;
    const bool mvalid = m < p.M;
    const bool stats = p.gn_acc && p.splitk == 1 && p.out_dtype != GG_F32;     // GroupNorm statistics of the NEXT norm
    float sv[2][4], sq[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int co = g0 * 32 + ct * 16 + fq * 4;
        f32x4 v = acc[ct];
        const long long o = m * p.Cout_pad + co;
#pragma unroll
        for (int j = 0; j < 4; ++j) sv[ct][j] = sq[ct][j] = 0.f;
        if (!mvalid) continue;
        if (p.splitk > 1) {
            *reinterpret_cast<f32x4 *>(p.ws + (long long)blockIdx.z * p.M * p.Cout_pad + o) = v;
            continue;
        }
        v += bias_pf[ct];
        if (p.residual) {
            const bf16x4 r = __builtin_bit_cast(bf16x4, res_pf[ct]);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (co + j >= p.Cout) v[j] = 0.f;
        if (p.out_dtype == GG_F32) {
            *reinterpret_cast<f32x4 *>((float *)p.out + o) = v;
        } else {
            bf16x4 ob;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ob[j] = (bf16_t)v[j];
                const float f = (float)ob[j];           // statistics of what the next norm will read
                sv[ct][j] = f;
                sq[ct][j] = f * f;
            }
            *reinterpret_cast<bf16x4 *>((bf16_t *)p.out + o) = ob;
        }
    }
    if (stats) {   // host guarantees Do*Ho*Wo % 64 == 0: the 64 rows of the block belong to one sample
        __shared__ float statp[4][32][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = gg_row16_sum(sv[ct][j]), b = gg_row16_sum(sq[ct][j]);   // over the wave's 16 rows
                if (fr == 0) { statp[wave][ct * 16 + fq * 4 + j][0] = a; statp[wave][ct * 16 + fq * 4 + j][1] = b; }
            }
        __syncthreads();
        if (tid < 64) {
            const int c = tid >> 1, which = tid & 1;
            const float t = ((statp[0][c][which] + statp[1][c][which]) + statp[2][c][which]) + statp[3][c][which];
            const long long fx = __double2ll_rn((double)t * (double)(which ? GG_ACC_SQ_SCALE : GG_ACC_SUM_SCALE));
            const long long nb = (long long)((unsigned)m0 / osp);
            atomicAdd(reinterpret_cast<unsigned long long *>(p.gn_acc + (((nb * GG_ACC_STRIPES + (blockIdx.x & (GG_ACC_STRIPES - 1))) * p.Cout_pad + g0 * 32 + c) * 2 + which)),
                      (unsigned long long)fx);
        }
    }
}

// ------------------------------------------------------------------------------------------ split-K reduce
// out[m, co] = sum_z slab[z][m][co] (fixed order: deterministic) + bias[n][co] (+ residual) -> bf16 / fp32
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvParams p)
{
    const int q4 = p.Cout_pad >> 2;
    const long long total = p.M * q4;
    const long long osp = (long long)p.Do * p.Ho * p.Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long m = i / q4;
        int co = (int)(i - m * q4) * 4;
        long long o = m * p.Cout_pad + co;
        // the slabs of a batch are requested together (one round trip per 16 slabs instead of one per few: the kernel is a chain of
        // dependent L2 round trips at batch 1), then added in slab order: the sum is the same, bit for bit
        constexpr int ZB = 16;
        const long long zs = p.M * p.Cout_pad;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z0 = 0; z0 < p.splitk; z0 += ZB) {
            f32x4 t[ZB];
#pragma unroll
            for (int k = 0; k < ZB; ++k) {
                t[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (z0 + k < p.splitk) t[k] = *reinterpret_cast<const f32x4 *>(p.ws + (long long)(z0 + k) * zs + o);
            }
#pragma unroll
            for (int k = 0; k < ZB; ++k) v += t[k];
        }
        if (p.bias) {
            int n = (int)(m / osp);
            v += *reinterpret_cast<const f32x4 *>(p.bias + (long long)n * p.bias_stride + co);
        }
        if (p.residual) {
            bf16x4 r = *reinterpret_cast<const bf16x4 *>(p.residual + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (co + j >= p.Cout) v[j] = 0.f;
        if (p.out_dtype == GG_F32) {
            *reinterpret_cast<f32x4 *>((float *)p.out + o) = v;
        } else {
            bf16x4 ob;
#pragma unroll
            for (int j = 0; j < 4; ++j) ob[j] = (bf16_t)v[j];
            *reinterpret_cast<bf16x4 *>((bf16_t *)p.out + o) = ob;
        }
    }
}

// ------------------------------------------------------------------------------------------ weight packing
__global__ void conv_pack_weight_kernel(const float *__restrict__ w, int Cout, int Cin, int Cin_pad, int ntaps,
                                        int Cout_pad, bf16_t *__restrict__ dst)
{
    const int nchunk = Cin_pad >> 5;
    const long long total = (long long)Cout_pad * ntaps * Cin_pad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int cil = (int)(i & 31);
        int col = (int)((i >> 5) & 31);
        long long t = i >> 10;
        int chunk = (int)(t % nchunk);
        t /= nchunk;
        int tap = (int)(t % ntaps);
        int g = (int)(t / ntaps);
        // physical 16-byte slot (cil>>3) of row `col` holds the LOGICAL chunk swz64(col, slot) (involution): the linear LDS
        // image of a tile is then bank-conflict free for ds_read_b128 without any swizzle at staging time
        int co = g * 32 + col, ci = chunk * 32 + swz64(col, cil >> 3) * 8 + (cil & 7);
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((long long)co * Cin + ci) * ntaps + tap];
        dst[i] = (bf16_t)v;
    }
}

extern "C" int64_t gg_conv_packed_weight_bytes(int32_t Cout, int32_t Cin_pad, int32_t ntaps)
{
    int64_t cp = ((int64_t)Cout + 31) / 32 * 32;
    return cp * ntaps * Cin_pad * 2;
}

extern "C" int gg_conv_pack_weight(const float *w, int32_t Cout, int32_t Cin, int32_t Cin_pad, int32_t ntaps,
                                   void *packed, void *stream)
{
    if (Cin_pad % 32 || Cin > Cin_pad || Cout <= 0 || ntaps <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "pack_weight: bad shape");
    int Cout_pad = (Cout + 31) / 32 * 32;
    long long total = (long long)Cout_pad * ntaps * Cin_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, Cin_pad,
                       ntaps, Cout_pad, (bf16_t *)packed);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

// halo fast path (gg_conv_halo.hip); returns GG_ERR_UNSUPPORTED when the shape is outside its envelope
int gg_conv_halo_try(const ConvParams &p, hipStream_t stream);
bool gg_conv_halo_prefers_separate_norm(const ConvParams &p);
#ifndef GG_HALO_SEPARATE_NORM
#define GG_HALO_SEPARATE_NORM 1      /* A/B switch: 0 = halo-tile convs always fuse the GroupNorm prologue */
#endif
// box-resident 2-D path for under-filled grids (gg_conv_box.hip); same contract as gg_conv_halo_try
int gg_conv_box_try(const ConvParams &p, hipStream_t stream);
bool gg_conv_box_fuses_prologue(const ConvParams &p);
// tiny-M weight-streaming path (gg_conv_tiny.hip): plan returns 0 (not applicable) or the K split
int gg_conv_tiny_plan(long long M, int Cout_pad, int KS, int prologue_act);
int gg_conv_tiny_launch(const ConvParams &p, hipStream_t stream);

template <int NT>
static int launch_gather(const ConvParams &p, hipStream_t stream)
{
    dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)(p.Cout_pad / (32 * NT)), (unsigned)p.splitk);
    if (p.prologue_act)
        hipLaunchKernelGGL((conv_gather_kernel<NT, (NT <= 2 ? 4 : 2), 1>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((conv_gather_kernel<NT, (NT <= 2 ? 4 : 2), 0>), grid, dim3(256), 0, stream, p);
    GG_CHECK_LAUNCH();
    if (p.splitk > 1) {
        long long total = p.M * (p.Cout_pad / 4);
        long long blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
        GG_CHECK_LAUNCH();
    }
    return GG_OK;
}

// Tile/split plan of the gather kernel. Large grids: widest cout tile (NT in {4,5,3,2,1}) and no split. Under-filled
// grids (small M: deep UNet levels at batch 1) are weight-streaming/latency bound: shrink the cout tile and split K
// over blockIdx.z until ~2 blocks per CU exist, keeping >= 8 k-steps per block.
struct GatherPlan { int NT, splitk; };
static GatherPlan plan_gather(long long M, int Cout_pad, int KS)
{
    const int G = Cout_pad / 32;
    int NT = (G % 4 == 0) ? 4 : (G % 5 == 0) ? 5 : (G % 3 == 0) ? 3 : (G % 2 == 0) ? 2 : 1;
    const long long mb = (M + 127) / 128;
    int splitk = 1;
    constexpr int k_target = 512, k_minsteps = 8;          // ~2 blocks per CU, >= 8 k-steps per block (tuned on the latent UNet)
    if (mb * (G / NT) < 192) {
        if (G % 2 == 0 && NT > 2 && mb * (G / 2) <= 1024) NT = 2;
        if (mb * (G / NT) < 192 && NT > 1) NT = 1;
        long long blocks = mb * (G / NT);
        long long want = (k_target + blocks - 1) / blocks;
        long long maxs = KS / k_minsteps > 0 ? KS / k_minsteps : 1;
        splitk = (int)(want < maxs ? want : maxs);
        if (splitk < 1) splitk = 1;
        if (splitk > 64) splitk = 64;
    }
    return {NT, splitk};
}

// 160-channel-step variant: 0 = not applicable, else the K split
static int plan_gather5(long long M, int C1, int C2, int Cout_pad, int ntaps)
{
    if (C1 % 160 || C2 % 160) return 0;
    const long long blocks = ((M + 63) / 64) * (Cout_pad / 32);
    if (blocks >= 4096) return 0;                                   // big grids: the wide-tile kernel has more MFMA per LDS byte
    const int KS5 = ((C1 + C2) / 160) * ntaps;
    constexpr int g5_target = 320, g5_minsteps = 5;
    long long sk = (g5_target + blocks - 1) / blocks;
    long long maxs = KS5 / g5_minsteps > 0 ? KS5 / g5_minsteps : 1;
    if (sk > maxs) sk = maxs;
    if (sk < 1) sk = 1;
    if (sk > 32) sk = 32;
    return (int)sk;
}

extern "C" int64_t gg_conv_workspace_bytes(const gg_conv_desc *d)
{
    if (!d || d->epilogue_geglu) return 0;           // (the GEGLU epilogue runs on the single-pass gather kernel)
    long long M = (long long)d->N * d->Do * d->Ho * d->Wo;
    int KS = d->kd * d->kh * d->kw * ((d->C1 + d->C2) / 32);
    if (int tsk = gg_conv_tiny_plan(M, d->Cout_pad, KS, d->prologue_act)) return tsk > 1 ? (int64_t)tsk * M * d->Cout_pad * 4 : 0;
    if (int s5 = plan_gather5(M, d->C1, d->C2, d->Cout_pad, d->kd * d->kh * d->kw)) return s5 > 1 ? (int64_t)s5 * M * d->Cout_pad * 4 : 0;
    GatherPlan pl = plan_gather(M, d->Cout_pad, KS);
    return pl.splitk > 1 ? (int64_t)pl.splitk * M * d->Cout_pad * 4 : 0;
}

static void fill_params(const gg_conv_desc *d, ConvParams &p)
{
    p.N = d->N; p.D = d->D; p.H = d->H; p.W = d->W; p.C1 = d->C1; p.C2 = d->C2; p.Cout = d->Cout; p.Cout_pad = d->Cout_pad;
    p.kd = d->kd; p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.upsample = d->upsample;
    p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.out_dtype = d->out_dtype; p.prologue_act = d->prologue_act; p.path_hint = d->path_hint;
    p.nchunk1 = d->C1 / 32; p.nchunk = (d->C1 + d->C2) / 32; p.ntaps = d->kd * d->kh * d->kw;
    p.M = (long long)d->N * d->Do * d->Ho * d->Wo;
    p.bias_stride = d->bias_stride;
    p.src1 = (const bf16_t *)d->src1; p.src2 = (const bf16_t *)d->src2; p.weight = (const bf16_t *)d->weight;
    p.residual = (const bf16_t *)d->residual; p.bias = d->bias; p.gn_scale = d->gn_scale; p.gn_shift = d->gn_shift;
    p.out = d->out;
    p.ws = nullptr;
    p.splitk = 1;
    p.gn_acc = (long long *)d->gn_acc;
    p.ddim_x = d->ddim_x; p.ddim_pred_x0 = d->ddim_pred_x0; p.ddim_scalars = d->ddim_scalars;
    p.ddim_unet_in = (bf16_t *)d->ddim_unet_in; p.ddim_unet_in_stride = d->ddim_unet_in_stride;
    p.epi_geglu = d->epilogue_geglu;
    p.pro_acc1 = (const long long *)d->pro_acc1; p.pro_acc2 = (const long long *)d->pro_acc2; p.pro_gamma = d->pro_gamma; p.pro_beta = d->pro_beta;
    p.pro_eps = d->pro_eps; p.pro_clog = d->pro_c_logical;
    if (d->gn_scale || d->gn_shift || !d->prologue_act) p.pro_acc1 = p.pro_acc2 = nullptr;          // external tables win; no prologue: unused
    p.skip_src1 = (const bf16_t *)d->skip_src1; p.skip_src2 = (const bf16_t *)d->skip_src2; p.skip_weight = (const bf16_t *)d->skip_weight;
    p.skip_C1 = d->skip_C1; p.skip_C2 = d->skip_C2;
    p.post_xt = d->post_xt; p.post_labels_out = d->post_labels_out; p.post_scalars = d->post_scalars; p.post_E = d->post_E;
    p.post_seed = d->post_philox_seed; p.post_offset_dev = (const long long *)d->post_philox_offset_dev;
    p.post_onehot_out = (bf16_t *)d->post_onehot_out; p.post_onehot_stride = d->post_onehot_stride; p.post_draw = d->post_draw;
    p.mg_osp = gg_magic_u32(p.M, d->Do * d->Ho * d->Wo); p.mg_ohw = gg_magic_u32(p.M, d->Ho * d->Wo); p.mg_wo = gg_magic_u32(p.M, d->Wo);
}

static bool halo_try_dry(const ConvParams &p) { return gg_conv_halo_try(p, (hipStream_t)-1) == GG_OK; }

extern "C" int gg_conv_runs_halo_tile(const gg_conv_desc *d)
{
    if (!d || d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->Cout_pad % 32 || d->epilogue_geglu) return 0;
    ConvParams p;
    fill_params(d, p);
    return gg_conv_halo_try(p, (hipStream_t)-1) == GG_OK ? 1 : 0;
}

extern "C" int gg_conv_fuses_prologue(const gg_conv_desc *d)
{
    if (!d || d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->Cout_pad % 32 || d->epilogue_geglu) return 0;
    ConvParams p;
    fill_params(d, p);
    // halo-tile convs CAN always fuse it; the answer is whether they should (measured rule in gg_conv_halo.hip)
    if (gg_conv_halo_try(p, (hipStream_t)-1) == GG_OK) return (GG_HALO_SEPARATE_NORM && d->path_hint == 0 && gg_conv_halo_prefers_separate_norm(p)) ? 0 : 1;
    if (gg_conv_box_try(p, (hipStream_t)-1) == GG_OK) return gg_conv_box_fuses_prologue(p) ? 1 : 0;
    // GroupNorm*SiLU inside the 160-step gather loop was measured slower (26 vs 16.6 us per conv: SiLU on the load -> LDS critical
    // path once per tap), so the gather kernels never fuse the prologue
    return 0;
}

bool gg_conv_box_emits_stats(const ConvParams &p);
bool gg_conv_box_prologue_from_acc(const ConvParams &p);
bool gg_conv_halo_fuses_posterior(const ConvParams &p);

extern "C" int gg_conv_fuses_posterior(const gg_conv_desc *d)
{
    if (!d || d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->Cout_pad % 32 || d->epilogue_geglu || d->residual || d->out_dtype != GG_F32 || d->gn_acc ||
        d->ddim_x || d->skip_C1) return 0;
    ConvParams p;
    fill_params(d, p);
    return gg_conv_halo_fuses_posterior(p) ? 1 : 0;
}

extern "C" int gg_conv_prologue_from_acc(const gg_conv_desc *d)
{
    if (!d || d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->Cout_pad % 32 || d->epilogue_geglu || !d->prologue_act) return 0;
    // two sources: the fold takes channel c < C1 from pro_acc1 and c - C1 from pro_acc2, i.e. every channel of the FIRST source must be a
    // logical one (pro_c_logical >= C1 says exactly that: padding lanes may only sit at the end of the second source)
    if (d->pro_c_logical <= 0 || d->pro_c_logical % 32 || d->pro_c_logical > d->C1 + d->C2 || (d->C2 && d->pro_c_logical < d->C1)) return 0;
    ConvParams p;
    fill_params(d, p);
    if (halo_try_dry(p)) return 0;
    return gg_conv_box_try(p, (hipStream_t)-1) == GG_OK && gg_conv_box_prologue_from_acc(p) ? 1 : 0;
}

// K-concatenated 1x1 skip projection: box kernel only (3x3, stride 1, no upsample: plan_box checks).
extern "C" int gg_conv_fuses_skip(const gg_conv_desc *d)
{
    if (!d || d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->Cout_pad % 32 || d->epilogue_geglu || d->skip_C1 <= 0 || d->skip_C1 % 32 || d->skip_C2 % 32 ||
        d->skip_C2 < 0 || d->residual || d->ddim_x)
        return 0;
    ConvParams p;
    fill_params(d, p);
    if (halo_try_dry(p)) return 0;
    return gg_conv_box_try(p, (hipStream_t)-1) == GG_OK ? 1 : 0;
}

// The fused DDIM epilogue lives in the box kernel's epilogue (the latent UNet's head conv at batch 1..4 runs there).
extern "C" int gg_conv_fuses_ddim(const gg_conv_desc *d)
{
    if (!d || d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->Cout_pad % 32 || d->out_dtype != GG_F32 || d->Cout != 4 || d->residual) return 0;
    ConvParams p;
    fill_params(d, p);
    if (halo_try_dry(p)) return 0;
    return gg_conv_box_try(p, (hipStream_t)-1) == GG_OK ? 1 : 0;
}


// Which path a desc takes is decided by the same plan functions gg_conv_forward uses.
extern "C" int gg_conv_emits_stats(const gg_conv_desc *d)
{
    if (!d || d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->Cout_pad % 32 || d->out_dtype != GG_BF16 || d->epilogue_geglu) return 0;
    ConvParams p;
    fill_params(d, p);
    if (halo_try_dry(p)) return GG_ACC_STRIPES_HALO;
    if (gg_conv_box_try(p, (hipStream_t)-1) == GG_OK) return gg_conv_box_emits_stats(p) ? GG_ACC_STRIPES : 0;
    if (gg_conv_tiny_plan(p.M, p.Cout_pad, p.ntaps * p.nchunk, p.prologue_act)) return 0;
    const long long osp = (long long)d->Do * d->Ho * d->Wo;
    return (plan_gather5(p.M, p.C1, p.C2, p.Cout_pad, p.ntaps) == 1 && osp % 64 == 0) ? GG_ACC_STRIPES : 0;
}

#ifdef GG_EXP_WPREFETCH
__global__ __launch_bounds__(256) void exp_wprefetch_kernel(const u32x4 *w, long long n16, unsigned *sink)
{
    unsigned a = 0;
    if (n16 <= (3 << 20) / 16 * 1) {        // <= 3 MB: EVERY XCD (blockIdx & 7) reads all of it into its own L2
        for (long long i = (long long)(blockIdx.x >> 3) * 256 + threadIdx.x; i < n16; i += 32 * 256) { const u32x4 v = w[i]; a ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
    } else
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += 256 * 256) { const u32x4 v = w[i]; a ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
    if (a == 0x12345679u) sink[0] = a;      // practically never
}
#endif
extern "C" int gg_conv_forward(const gg_conv_desc *d, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: null desc");
    if (d->C1 <= 0 || d->C1 % 32 || d->C2 % 32 || d->C2 < 0) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: C1/C2 must be multiples of 32 (got %d, %d)", d->C1, d->C2);
    if (d->Cout_pad % 32 || d->Cout > d->Cout_pad || d->Cout <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: bad Cout/Cout_pad %d/%d", d->Cout, d->Cout_pad);
    auto okk = [](int k) { return k == 1 || k == 3; };
    if (!okk(d->kd) || !okk(d->kh) || !okk(d->kw)) GG_FAIL(GG_ERR_UNSUPPORTED, "conv: kernel extent must be 1 or 3");
    if (d->stride != 1 && d->stride != 2) GG_FAIL(GG_ERR_UNSUPPORTED, "conv: stride must be 1 or 2");
    if (d->upsample && d->stride != 1) GG_FAIL(GG_ERR_UNSUPPORTED, "conv: upsample with stride");
    if (d->out_dtype != GG_BF16 && d->out_dtype != GG_F32) GG_FAIL(GG_ERR_BAD_DTYPE, "conv: out dtype");
    if (!d->src1 || !d->weight || (!d->out && !d->post_xt) || (d->C2 && !d->src2)) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: null pointer");
    if (d->post_xt) {
        if (!gg_conv_fuses_posterior(d)) GG_FAIL(GG_ERR_UNSUPPORTED, "conv: this shape cannot run the fused CCDM reverse step (gg_conv_fuses_posterior)");
        if (!d->post_labels_out || !d->post_scalars) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: fused CCDM reverse step needs labels_out and scalars");
        if (d->post_onehot_out && (d->post_onehot_stride < d->Cout || (d->post_onehot_stride & 1) || ((uintptr_t)d->post_onehot_out & 3)))
            GG_FAIL(GG_ERR_BAD_SHAPE, "conv: fused CCDM reverse step: one-hot rows must be 4-byte aligned (even stride >= K)");
    }
    const bool pro_acc = d->prologue_act && !d->gn_scale && !d->gn_shift && d->pro_acc1;
    if (pro_acc) {
        if (!gg_conv_prologue_from_acc(d)) GG_FAIL(GG_ERR_UNSUPPORTED, "conv: this shape cannot compute its GroupNorm prologue from accumulators (gg_conv_prologue_from_acc)");
        if (!d->pro_gamma || !d->pro_beta || (d->C2 && !d->pro_acc2)) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: accumulator prologue needs gamma / beta and one accumulator per source");
    } else if (d->prologue_act && (!d->gn_scale || !d->gn_shift)) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: prologue without scale/shift");
    if (d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Do <= 0 || d->Ho <= 0 || d->Wo <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: empty extent");
    // output extent must match the conv arithmetic (what nn.ConvNd / F.interpolate would produce)
    auto expect = [&](int in, int k, int up) {
        int e = up ? in * 2 : in;
        if (k == 1) return (d->upsample && !up) ? e : (e - 1) / d->stride + 1;
        // pad lo = d->pad, pad hi chosen so that 'same' (pad 1) or AE (0,1) arithmetic holds
        return (d->stride == 1) ? e + 2 * d->pad - 2 : (d->pad == 1 ? (e + 2 - 3) / 2 + 1 : (e + 1 - 3) / 2 + 1);
    };
    const int upD = d->upsample && d->kd == 3, upHW = d->upsample;
    if (d->Do != expect(d->D, d->kd, upD) || d->Ho != expect(d->H, d->kh, upHW) || d->Wo != expect(d->W, d->kw, upHW))
        GG_FAIL(GG_ERR_BAD_SHAPE, "conv: output extent (%d,%d,%d) inconsistent with input (%d,%d,%d) k=(%d,%d,%d) stride %d pad %d up %d",
                d->Do, d->Ho, d->Wo, d->D, d->H, d->W, d->kd, d->kh, d->kw, d->stride, d->pad, d->upsample);

    if (d->skip_C1) {
        if (!gg_conv_fuses_skip(d)) GG_FAIL(GG_ERR_UNSUPPORTED, "conv: this shape cannot take a K-concatenated skip projection (gg_conv_fuses_skip)");
        if (!d->skip_src1 || !d->skip_weight || (d->skip_C2 && !d->skip_src2)) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: skip projection without source / weight");
    }
    if (d->ddim_x && !gg_conv_fuses_ddim(d)) GG_FAIL(GG_ERR_UNSUPPORTED, "conv: this shape cannot run the fused DDIM epilogue (gg_conv_fuses_ddim)");
    if (d->ddim_x && (!d->ddim_scalars || (d->ddim_unet_in && d->ddim_unet_in_stride < 4))) GG_FAIL(GG_ERR_BAD_SHAPE, "conv: fused DDIM epilogue needs scalars / a unet_in stride >= 4");
    ConvParams p;
    fill_params(d, p);
    if (p.M >= (1LL << 31) || (long long)d->D * d->H * d->W * (d->C1 > d->C2 ? d->C1 : d->C2) >= (1LL << 31))
        GG_FAIL(GG_ERR_UNSUPPORTED, "conv: tensor too large for 32-bit in-sample offsets");

    if (d->epilogue_geglu) {
        // fused GEGLU epilogue: implemented in the generic gather kernel's single-pass (no split-K) epilogue only
        if (d->kd != 1 || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->upsample || d->residual || d->out_dtype != GG_BF16 || d->gn_acc ||
            d->ddim_x || d->Cout % 32 || d->bias_stride)
            GG_FAIL(GG_ERR_UNSUPPORTED, "conv: the GEGLU epilogue needs a 1x1 conv, bf16 output, Cout = 2*inner with inner %% 16 == 0, a shared bias, no residual / gn_acc / ddim");
        const GatherPlan plg = plan_gather(p.M, p.Cout_pad, p.ntaps * p.nchunk);
        switch (plg.NT) {
            case 5: return launch_gather<5>(p, stream);
            case 4: return launch_gather<4>(p, stream);
            case 3: return launch_gather<3>(p, stream);
            case 2: return launch_gather<2>(p, stream);
            default: return launch_gather<1>(p, stream);
        }
    }
#ifdef GG_EXP_WPREFETCH
    // experiment (tools/experiments/README.md, "warm weights"): every conv preceded by a launch that reads its packed weights with the whole
    // chip, so that the conv finds them in L2 / the memory-side cache -- bounds what a CONCURRENT weight prefetcher could buy
    {
        const long long wbytes = (long long)(p.Cout_pad / 32) * p.ntaps * p.nchunk * 2048;
        hipLaunchKernelGGL(exp_wprefetch_kernel, dim3(256), dim3(256), 0, stream, (const u32x4 *)p.weight, wbytes >> 4, (unsigned *)d->out);
    }
#endif
#ifdef GG_H3_STAMPS
    if (d->path_hint == 7 && d->workspace) p.ws = (float *)d->workspace;      // diagnostic build: phase stamps of the team halo kernel
#endif
    int rc = gg_conv_halo_try(p, stream);
#ifdef GG_H3_STAMPS
    p.ws = nullptr;
#endif
    if (rc != GG_ERR_UNSUPPORTED) return rc;
#ifdef GG_BOX_STAMPS
    if (d->path_hint == 98) p.ws = (float *)d->workspace;      // diagnostic build: phase stamps of the box kernel
#endif
    rc = gg_conv_box_try(p, stream);
#ifdef GG_BOX_STAMPS
    p.ws = nullptr;
#endif
    if (rc != GG_ERR_UNSUPPORTED) return rc;

    if (int tsk = gg_conv_tiny_plan(p.M, p.Cout_pad, p.ntaps * p.nchunk, p.prologue_act)) {
        if (tsk > 1) {
            const long long need = (long long)tsk * p.M * p.Cout_pad * 4;
            if (!d->workspace || d->workspace_bytes < need)
                GG_FAIL(GG_ERR_WORKSPACE_TOO_SMALL, "conv: split-K needs %lld workspace bytes (gg_conv_workspace_bytes), got %lld", need, (long long)d->workspace_bytes);
            p.ws = (float *)d->workspace;
        }
        p.splitk = tsk;
        rc = gg_conv_tiny_launch(p, stream);
        if (rc != GG_OK || tsk == 1) return rc;
        long long total = p.M * (p.Cout_pad / 4);
        long long blocks = (total + 255) / 256;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
        GG_CHECK_LAUNCH();
        return GG_OK;
    }

    if (int s5 = plan_gather5(p.M, p.C1, p.C2, p.Cout_pad, p.ntaps)) {
        if (s5 > 1) {
            const long long need = (long long)s5 * p.M * p.Cout_pad * 4;
            if (!d->workspace || d->workspace_bytes < need)
                GG_FAIL(GG_ERR_WORKSPACE_TOO_SMALL, "conv: split-K needs %lld workspace bytes (gg_conv_workspace_bytes), got %lld", need, (long long)d->workspace_bytes);
            p.ws = (float *)d->workspace;
        }
        p.splitk = s5;
        dim3 grid((unsigned)((p.M + 63) / 64), (unsigned)(p.Cout_pad / 32), (unsigned)s5);
        if (p.prologue_act && (((long long)p.Do * p.Ho * p.Wo) % 64 || p.C1 + p.C2 > 2560))
            GG_FAIL(GG_ERR_UNSUPPORTED, "conv: fused prologue on the 160-step kernel needs Do*Ho*Wo %% 64 == 0 and C <= 2560");
        if (p.prologue_act) hipLaunchKernelGGL(conv_gather5_kernel<1>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(conv_gather5_kernel<0>, grid, dim3(256), 0, stream, p);
        GG_CHECK_LAUNCH();
        if (s5 > 1) {
            long long total = p.M * (p.Cout_pad / 4);
            long long blocks = (total + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
            GG_CHECK_LAUNCH();
        }
        return GG_OK;
    }

    GatherPlan pl = plan_gather(p.M, p.Cout_pad, p.ntaps * p.nchunk);
    if (pl.splitk > 1) {
        const long long need = (long long)pl.splitk * p.M * p.Cout_pad * 4;
        if (!d->workspace || d->workspace_bytes < need)
            GG_FAIL(GG_ERR_WORKSPACE_TOO_SMALL, "conv: split-K needs %lld workspace bytes (gg_conv_workspace_bytes), got %lld", need, (long long)d->workspace_bytes);
        p.ws = (float *)d->workspace;
        p.splitk = pl.splitk;
    }
    switch (pl.NT) {
        case 5: return launch_gather<5>(p, stream);
        case 4: return launch_gather<4>(p, stream);
        case 3: return launch_gather<3>(p, stream);
        case 2: return launch_gather<2>(p, stream);
        default: return launch_gather<1>(p, stream);
    }
}
