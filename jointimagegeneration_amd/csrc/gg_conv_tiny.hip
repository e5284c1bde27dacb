// Tiny-M convolution: M = N*Do*Ho*Wo <= 128 output positions (deep UNet levels at batch 1: 4x4, 8x8 latents).
// These layers are pure WEIGHT STREAMING (e.g. 1600->800 3x3 at 4x4: 23 MB of weights for 16 positions), so the kernel is
// built like a GEMV: no LDS staging, no barrier in the main loop.  Each wave owns a contiguous run of k-steps
// (tap-major order == the packed weight order, so its weight tiles are one contiguous stream), loads the 32x32 weight
// tile of every k-step straight into VGPRs (two 1-KiB wave loads), gathers the tiny activation operand from L2 with
// bounds-checked 16-byte loads, and accumulates with MFMA 16x16x32.  The 4 waves of a workgroup are combined through LDS
// once at the end; K is additionally split over blockIdx.x (deterministic slab reduce as in the gather kernel).
#include "gg_conv.h"

template <int PT>          // position tiles of 16 (PT*16 >= M)
__global__ __launch_bounds__(256) void conv_tinym_kernel(const ConvParams p)
{
    constexpr int MP = PT * 16;
    __shared__ __attribute__((aligned(16))) float red[4 * MP * 32];          // [wave][m][co]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int g = blockIdx.y, z = blockIdx.x;
    const int KS = p.ntaps * p.nchunk;
    const int kb0 = (int)(((long long)KS * z) / p.splitk), kb1 = (int)(((long long)KS * (z + 1)) / p.splitk);
    const int span = kb1 - kb0;
    const int k0 = kb0 + (span * wave) / 4, k1 = kb0 + (span * (wave + 1)) / 4;

    // ---- per-lane output positions (one per position tile): coordinates of tap (0,0,0) and per-sample source bases
    const unsigned osp = (unsigned)(p.Do * p.Ho * p.Wo), ohw = (unsigned)(p.Ho * p.Wo);
    int bd[PT], bh[PT], bw[PT];
    bool rv[PT];
    const bf16_t *rb1[PT], *rb2[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const unsigned m = (unsigned)(pt * 16 + fr);
        rv[pt] = m < (unsigned)p.M;
        const unsigned mu = rv[pt] ? m : 0u;
        const unsigned n = mu / osp;
        unsigned r = mu - n * osp;
        const unsigned od = r / ohw;
        r -= od * ohw;
        const unsigned oh = r / (unsigned)p.Wo, ow = r - (r / (unsigned)p.Wo) * (unsigned)p.Wo;
        bd[pt] = (p.kd == 1) ? (int)od * p.stride : (int)od * p.stride - p.pad;
        bh[pt] = (p.kh == 1) ? (int)oh * p.stride : (int)oh * p.stride - p.pad;
        bw[pt] = (p.kw == 1) ? (int)ow * p.stride : (int)ow * p.stride - p.pad;
        const long long sp = (long long)p.D * p.H * p.W;
        rb1[pt] = p.src1 + (long long)n * sp * p.C1 + fq * 8;
        rb2[pt] = p.src2 ? p.src2 + (long long)n * sp * p.C2 + fq * 8 : nullptr;
    }
    const int upD = (p.upsample && p.kd == 3) ? 1 : 0, upHW = p.upsample ? 1 : 0;
    const unsigned limD = (unsigned)(p.D << upD), limH = (unsigned)(p.H << upHW), limW = (unsigned)(p.W << upHW);

    f32x4 acc[PT][2];
#pragma unroll
    for (int a = 0; a < PT; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // weight stream of this wave: tiles (g, ks) for ks in [k0, k1), 1024 elements each, rows pre-swizzled at pack time
    const bf16_t *wlane0 = p.weight + (((long long)g * KS + k0) << 10) + fr * 32 + swz64(fr, fq) * 8;
    const bf16_t *wlane1 = p.weight + (((long long)g * KS + k0) << 10) + (16 + fr) * 32 + swz64(16 + fr, fq) * 8;

    int ks = k0;
    while (ks < k1) {
        // ---- one tap segment: chunks [c, cend) of tap `tap`
        const int tap = ks / p.nchunk;
        int c = ks - tap * p.nchunk;
        const int cend_tap = (k1 - tap * p.nchunk < p.nchunk) ? k1 - tap * p.nchunk : p.nchunk;
        const int tkd = tap / (p.kh * p.kw), tkh = (tap / p.kw) % p.kh, tkw = tap % p.kw;
        unsigned pos[PT];
        bool ok[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const unsigned ud = (unsigned)(bd[pt] + tkd), uh = (unsigned)(bh[pt] + tkh), uw = (unsigned)(bw[pt] + tkw);
            ok[pt] = rv[pt] && ud < limD && uh < limH && uw < limW;
            pos[pt] = ((ud >> upD) * (unsigned)p.H + (uh >> upHW)) * (unsigned)p.W + (uw >> upHW);
        }
        // ---- two source segments (fused skip concat): src1 chunks [0, nchunk1), src2 chunks [nchunk1, nchunk)
#pragma unroll 1
        for (int seg = 0; seg < 2; ++seg) {
            const int s0 = seg == 0 ? 0 : p.nchunk1, s1 = seg == 0 ? p.nchunk1 : p.nchunk;
            int cb = c > s0 ? c : s0;
            const int ce = cend_tap < s1 ? cend_tap : s1;
            if (cb >= ce) continue;
            const int Cs = seg == 0 ? p.C1 : p.C2;
            const bf16_t *xp[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
                xp[pt] = (seg == 0 ? rb1[pt] : rb2[pt]) + (long long)pos[pt] * Cs + (cb - s0) * 32;
            const int nsteps = ce - cb;
            int i = 0;
            // 4 k-steps per trip: all 8 weight loads and 4*PT activation loads are issued before the first MFMA
            for (; i + 4 <= nsteps; i += 4) {
                bf16x8 w0[4], w1[4], xf[4][PT];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w0[u] = *reinterpret_cast<const bf16x8 *>(wlane0 + u * 1024);
                    w1[u] = *reinterpret_cast<const bf16x8 *>(wlane1 + u * 1024);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        u32x4 v = {0u, 0u, 0u, 0u};
                        if (ok[pt]) v = *reinterpret_cast<const u32x4 *>(xp[pt] + u * 32);
                        xf[u][pt] = __builtin_bit_cast(bf16x8, v);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        acc[pt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[u], xf[u][pt], acc[pt][0], 0, 0, 0);
                        acc[pt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[u], xf[u][pt], acc[pt][1], 0, 0, 0);
                    }
                wlane0 += 4096;
                wlane1 += 4096;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) xp[pt] += 128;
            }
            for (; i < nsteps; ++i) {
                const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(wlane0);
                const bf16x8 w1 = *reinterpret_cast<const bf16x8 *>(wlane1);
                bf16x8 xf[PT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (ok[pt]) v = *reinterpret_cast<const u32x4 *>(xp[pt]);
                    xf[pt] = __builtin_bit_cast(bf16x8, v);
                    xp[pt] += 32;
                }
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    acc[pt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xf[pt], acc[pt][0], 0, 0, 0);
                    acc[pt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xf[pt], acc[pt][1], 0, 0, 0);
                }
                wlane0 += 1024;
                wlane1 += 1024;
            }
            ks += nsteps;
            c = ce;
        }
    }

    // ---- combine the 4 waves (fixed order: deterministic), then slab or final epilogue
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
            *reinterpret_cast<f32x4 *>(&red[(wave * MP + pt * 16 + fr) * 32 + ct * 16 + fq * 4]) = acc[pt][ct];
    __syncthreads();
    for (int i = tid; i < MP * 32; i += 256) {
        const int m = i >> 5, cl = i & 31;
        if (m >= (int)p.M) continue;
        float v = red[i];
        v += red[MP * 32 + i];
        v += red[2 * MP * 32 + i];
        v += red[3 * MP * 32 + i];
        const int co = g * 32 + cl;
        const long long o = (long long)m * p.Cout_pad + co;
        if (p.splitk > 1) {
            p.ws[(long long)z * p.M * p.Cout_pad + o] = v;
            continue;
        }
        if (p.bias) v += p.bias[(long long)((unsigned)m / osp) * p.bias_stride + co];
        if (p.residual) v += (float)p.residual[o];
        if (co >= p.Cout) v = 0.f;
        if (p.out_dtype == GG_F32) ((float *)p.out)[o] = v;
        else ((bf16_t *)p.out)[o] = (bf16_t)v;
    }
}

template <int PT>
static void launch_tiny(const ConvParams &p, hipStream_t stream)
{
    dim3 grid((unsigned)p.splitk, (unsigned)(p.Cout_pad / 32));
    hipLaunchKernelGGL(conv_tinym_kernel<PT>, grid, dim3(256), 0, stream, p);
}

// plan: 0 = not applicable, else the K split
int gg_conv_tiny_plan(long long M, int Cout_pad, int KS, int prologue_act)
{
    if (M > 128 || prologue_act) return 0;
    const int G = Cout_pad / 32;
    long long want = (768 + G - 1) / G;              // ~3 workgroups per CU
    long long maxs = KS / 16 > 0 ? KS / 16 : 1;      // >= 16 k-steps per workgroup (4 per wave)
    long long sk = want < maxs ? want : maxs;
    if (sk < 1) sk = 1;
    if (sk > 64) sk = 64;
    return (int)sk;
}

int gg_conv_tiny_launch(const ConvParams &p, hipStream_t stream)
{
    if (p.M <= 16) launch_tiny<1>(p, stream);
    else if (p.M <= 32) launch_tiny<2>(p, stream);
    else if (p.M <= 64) launch_tiny<4>(p, stream);
    else launch_tiny<8>(p, stream);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
