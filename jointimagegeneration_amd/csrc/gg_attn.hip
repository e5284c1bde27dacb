// Flash-style attention on MFMA 16x16x32 bf16 (gfx950): out = softmax(scale * Q K^T) V without a TxT buffer.
//
// Orientation (per wave = 16 queries, per step = 32 keys), chosen so that no score ever changes lane:
//   S^T[key, q]  = K Q^T      A = K rows from LDS (row = key, k = d), B = Q rows held in registers (col = q)
//                             -> lane (q = l&15, g = l>>4) holds keys {4g..4g+3} and {16+4g..16+4g+3}
//   softmax over keys         = in-lane over 8 values + two xor-shuffles (lanes l^16, l^32); state (m, l) per lane
//   O^T[d, q]   += V^T P^T    B = P^T straight from the score registers (k-slot (g,j) <-> key 4g+j | 16+4g+j-4),
//                             A = V^T via ds_read_b64_tr_b16 on the row-major V tile with the SAME k-slot map
//   epilogue                  lane holds 4 consecutive d of one query -> 8-byte stores
#include "gg_common.h"

#ifndef GG_ATTN_WS2
#define GG_ATTN_WS2 1          /* in-workgroup key split for under-filled grids (A/B: tools/experiments) */
#endif
struct AttnParams {
    int N, heads, Tq, Tkv;
    long long ldq, hsq, ldk, hsk, ldv, hsv, ldo, hso;
    float scale_log2;
    const bf16_t *q, *k, *v;
    bf16_t *out;
    // key split (flash-decoding): block s of `ksplit` handles keys [s * kchunk, (s + 1) * kchunk) and leaves its UNNORMALISED state
    // (o fp32 [.., D], m, l) in the workspace; attn_merge_kernel combines the states.  ksplit == 1: plain kernel.
    int ksplit, kchunk;
    float *ws_o, *ws_ml;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

// chunk (16 B) swizzle inside a row of CH chunks so that 16 consecutive rows at one column chunk hit distinct banks
template <int CH>
__device__ __forceinline__ int swz_row(int row, int chunk)
{
    if constexpr (CH == 4) return chunk ^ ((row >> 1) & 2);
    else if constexpr (CH == 8) return chunk ^ ((row >> 1) & 7);
    else return chunk ^ (row & 15);
}

// WS = 2: key split INSIDE the workgroup (8 waves): waves 0-3 and waves 4-7 take the same 64 queries and one half of the keys each, with
// K/V tiles of their own, and merge their online-softmax states through LDS at the end.  For under-filled grids with long key loops (the
// latent UNet's 32x32 attention at batch 1: 160 workgroups, T = 1024: 8 softmax steps of 128 keys per wave, the kernel is bound by that
// serial chain): half the steps per wave for the same staging work per thread.
template <int D, int KT, int WS = 1>
__global__ __launch_bounds__(256 * WS) void attn_kernel(const AttnParams p)
{
    static_assert(WS == 1 || (WS == 2 && D < 256), "in-workgroup key split: register-staged tiles only");
    constexpr int CH = D / 8;          // 16-byte chunks per row
    constexpr int ROWB = D * 2;
    constexpr int KS = D / 32;         // k-steps of the S^T product
    constexpr int DT = D / 16;         // d tiles of O^T
    // D >= 256 (AE mid-block attention: one head of 384 / 512 channels): a K/V tile is 48-64 KiB and the register staging path
    // cannot prefetch it; there the tiles are double-buffered in LDS and filled by LDS-DMA one tile ahead (one wave instruction
    // per K or V row), hand-counted vmcnt.  (Inline-asm DMA: the compiler would drain all LDS-DMA before every ds_read.)
    constexpr bool DMA = D >= 256;
    constexpr int TILEB = 2 * KT * ROWB;               // K tile + V tile
    __shared__ __attribute__((aligned(1024))) char smem[(DMA ? 2 : WS) * TILEB];
    const int grp = WS == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;      // key half of this wave group
    char *ksm = smem + grp * TILEB, *vsm = ksm + KT * ROWB;

    const int tid = threadIdx.x & 255, lane = tid & 63;            // (thread index inside the wave group)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform (LDS-DMA destinations live in M0)
    const int fr = lane & 15, g = lane >> 4;
    const int n = blockIdx.z, h = blockIdx.y;
    const int qt = p.ksplit > 1 ? blockIdx.x / p.ksplit : blockIdx.x, ksp = p.ksplit > 1 ? blockIdx.x - qt * p.ksplit : 0;
    const int q0 = qt * 64 + wave * 16;
    int kbeg = ksp * p.kchunk, kend = (p.ksplit > 1 && kbeg + p.kchunk < p.Tkv) ? kbeg + p.kchunk : p.Tkv;     // kchunk % KT == 0
    int ntile = (kend - kbeg + KT - 1) / KT;                       // tiles of this workgroup's key range
    if constexpr (WS == 2) {                                       // both groups walk the SAME number of tiles (whole-workgroup barriers)
        ntile = (ntile + 1) >> 1;
        kbeg += grp * ntile * KT;
        const int ke = kbeg + ntile * KT;
        kend = ke < kend ? ke : kend;                              // (the second group's range may be short or empty: masked / skipped steps)
    }

    // Q fragments (B operand: col = query fr, k = 8g + j)
    bf16x8 qf[KS];
    {
        int qi = q0 + fr;
        if (qi >= p.Tq) qi = p.Tq - 1;
        const bf16_t *qp = p.q + ((long long)n * p.Tq + qi) * p.ldq + (long long)h * p.hsq + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(qp + ks * 32);
    }

    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const bf16_t *kbase = p.k + (long long)n * p.Tkv * p.ldk + (long long)h * p.hsk;
    const bf16_t *vbase = p.v + (long long)n * p.Tkv * p.ldv + (long long)h * p.hsv;

    // K/V tiles go global -> registers -> LDS; the registers of tile t+1 are requested before tile t is consumed, so the
    // load round trip (1.5-2 us) hides behind the MFMA/softmax work instead of adding to every tile
    constexpr int NP = (KT * CH + 255) / 256;          // 16-byte pieces per thread and operand
    constexpr bool PREFETCH = NP <= 4;                 // 32 VGPRs at most (D <= 256)
    u32x4 kreg[NP], vreg[NP];
    auto fetch = [&](int key0) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + 256 * j;
            const int r = i / CH, c = i - r * CH;
            kreg[j] = u32x4{0u, 0u, 0u, 0u};
            vreg[j] = u32x4{0u, 0u, 0u, 0u};
            if (i < KT * CH && key0 + r < p.Tkv) {
                kreg[j] = *reinterpret_cast<const u32x4 *>(kbase + (long long)(key0 + r) * p.ldk + c * 8);
                vreg[j] = *reinterpret_cast<const u32x4 *>(vbase + (long long)(key0 + r) * p.ldv + c * 8);
            }
        }
    };
    const unsigned smem_off = (unsigned)(size_t)((__attribute__((address_space(3))) char *)smem);
    auto stage_dma = [&](int key0, int buf) {          // this wave's rows r = wave, wave + 4, ...: 2 * KT / 4 DMA instructions
#pragma unroll
        for (int rr = 0; rr < KT / 4; ++rr) {
            const int r = rr * 4 + wave;
            int key = key0 + r;
            if (key >= p.Tkv) key = p.Tkv - 1;         // rows past the end are masked (-inf scores), any finite data will do
            const int c = lane ^ (r & 15);             // LDS chunk slot `lane` of row r holds global chunk c (swz_row is an involution)
            if (lane < CH) {
                const bf16_t *ks = kbase + (long long)key * p.ldk + c * 8;
                const bf16_t *vs = vbase + (long long)key * p.ldv + c * 8;
                const unsigned kd = __builtin_amdgcn_readfirstlane(smem_off + buf * TILEB + r * ROWB), vd = kd + KT * ROWB;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(ks), "s"(kd) : "memory", "m0");
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vs), "s"(vd) : "memory", "m0");
            }
        }
    };
    if (DMA) stage_dma(kbeg, 0);
    if (!DMA && PREFETCH) fetch(kbeg);
    int tile = 0;
    for (int key0 = kbeg; tile < ntile; key0 += KT, ++tile) {
        if (DMA) {
            const bool more = key0 + KT < kend;
            if (more) stage_dma(key0 + KT, (tile + 1) & 1);          // the other buffer: last read one tile ago, barrier since
            // this wave's 2*KT/4 DMAs of the current tile have landed once at most the next tile's are outstanding
            if (more) __builtin_amdgcn_s_waitcnt(((2 * KT / 4) & 15) | (((2 * KT / 4) >> 4) << 14) | 0x0F70);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ksm = smem + (tile & 1) * TILEB;
            vsm = ksm + KT * ROWB;
        } else {
        __syncthreads();   // previous tile fully consumed
        if (!PREFETCH) fetch(key0);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + 256 * j;
            const int r = i / CH, c = i - r * CH;
            if (i < KT * CH) {
                const int off = r * ROWB + swz_row<CH>(r, c) * 16;
                *reinterpret_cast<u32x4 *>(ksm + off) = kreg[j];
                *reinterpret_cast<u32x4 *>(vsm + off) = vreg[j];
            }
        }
        __syncthreads();
        if (PREFETCH && key0 + KT < kend) fetch(key0 + KT);
        }

        // One online-softmax step covers G x 32 keys (up to 128): the max / sum lane reductions and the rescale of O happen once
        // per step, and the G S^T tiles, 8G exponentials and G PV products inside a step are independent, so a single wave per
        // SIMD (under-filled grids: T <= 1024 at batch 1) is not serialised on 4 shuffle round trips per 32 keys.
        // (Splitting the keys of a 16-query workgroup over its 4 waves instead was slower: 4x the K/V staging per query.)
        constexpr int G = KT / 32 >= 4 ? 4 : KT / 32;
#pragma unroll
        for (int sub0 = 0; sub0 < KT / 32; sub0 += G) {
            if (key0 + sub0 * 32 >= kend) break;
            // ---- S^T for G x 32 keys: 2G 16-key tiles
            f32x4 s[G][2];
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    s[gi][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int r = (sub0 + gi) * 32 + kt * 16 + fr;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        bf16x8 kf = *reinterpret_cast<const bf16x8 *>(ksm + r * ROWB + swz_row<CH>(r, ks * 4 + g) * 16);
                        s[gi][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[gi][kt], 0, 0, 0);
                    }
                }
            // ---- online softmax (fp32, base-2).  The kernel is VALU-issue bound at batch 1 (one wave per SIMD, 64 scores per lane and
            //      tile), so the per-score work is kept to max, one FMA, one raw v_exp_f32, one add and half a pack: the scale is
            //      applied inside the exponent's FMA (scale > 0 commutes with the max), the key mask exists only on a ragged LAST tile
            //      (wave-uniform branch), and v_exp_f32 is used bare (arguments <= 0; results below 2^-126 may flush to zero, they are
            //      rounded to bf16 anyway).
            const bool ragged = key0 + (sub0 + G) * 32 > kend;
            if (ragged) {
#pragma unroll
                for (int gi = 0; gi < G; ++gi)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = key0 + (sub0 + gi) * 32 + kt * 16 + 4 * g + r;
                            s[gi][kt][r] = (key < kend) ? s[gi][kt][r] : -INFINITY;
                        }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[gi][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx * p.scale_log2);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float rs = 0.f;
            bf16x8 pf[G];
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[gi][kt][r], p.scale_log2, -m_new));
                        rs += pv;
                        pf[gi][kt * 4 + r] = (bf16_t)pv;
                    }
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            // ---- O^T = alpha * O^T + V^T P^T
            const int qrow = fr >> 2, pcol = fr & 3;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int chunk = 2 * dt + (pcol >> 1);
                const int sub8 = (pcol & 1) * 8;
                f32x4 acc = o[dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] *= alpha;
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    const int vr0 = (sub0 + gi) * 32 + 4 * g + qrow, vr1 = vr0 + 16;
                    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(vsm + vr0 * ROWB + swz_row<CH>(vr0, chunk) * 16 + sub8));
                    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(vsm + vr1 * ROWB + swz_row<CH>(vr1, chunk) * 16 + sub8));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x8 vv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf[gi], acc, 0, 0, 0);
                }
                o[dt] = acc;
            }
        }
        if (DMA) {                                     // every wave is done with this buffer before it is refilled two tiles on
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this wave's LDS reads have returned
            __builtin_amdgcn_s_barrier();
        }
    }

    if constexpr (WS == 2) {
        // merge the two key halves: out = (w0 o0 + w1 o1) / (w0 l0 + w1 l1), w = 2^(m - max m) (the exact combination, as attn_merge_kernel)
        __syncthreads();                                   // every wave is done with its tiles
        float *st = reinterpret_cast<float *>(smem);       // [DT * 4 + 2][256]
        if (grp == 1) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(dt * 4 + r) * 256 + tid] = o[dt][r];
            st[(DT * 4) * 256 + tid] = m_run;
            st[(DT * 4 + 1) * 256 + tid] = l_run;
        }
        __syncthreads();
        if (grp == 1) return;
        const float m1 = st[(DT * 4) * 256 + tid], l1 = st[(DT * 4 + 1) * 256 + tid];
        const float mm = fmaxf(m_run, m1);
        const float w0 = __builtin_amdgcn_exp2f(m_run - mm), w1 = __builtin_amdgcn_exp2f(m1 - mm);      // (m1 = -inf for an empty half: w1 = 0)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] = o[dt][r] * w0 + st[(dt * 4 + r) * 256 + tid] * w1;
        l_run = l_run * w0 + l1 * w1;
        m_run = mm;
    }
    const int qi = q0 + fr;
    if (p.ksplit > 1) {
        if (qi < p.Tq) {          // unnormalised partial state of this key range
            const long long row = (((long long)n * p.heads + h) * p.ksplit + ksp) * p.Tq + qi;
            float *op = p.ws_o + row * D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4 *>(op + dt * 16) = o[dt];
            if (g == 0) { p.ws_ml[row * 2] = m_run; p.ws_ml[row * 2 + 1] = l_run; }
        }
        return;
    }
    if (qi < p.Tq) {
        const float inv = 1.0f / l_run;
        bf16_t *op = p.out + ((long long)n * p.Tq + qi) * p.ldo + (long long)h * p.hso + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)(o[dt][r] * inv);
            *reinterpret_cast<bf16x4 *>(op + dt * 16) = ob;
        }
    }
}

// out[q, :] = sum_s w_s o_s[q, :] / sum_s w_s l_s with w_s = 2^(m_s - max_s m_s): the exact combination of the key ranges' online-softmax
// states (base-2 exponents as in the kernel).  One thread per (query, 4 channels).
__global__ __launch_bounds__(256) void attn_merge_kernel(const float *__restrict__ ws_o, const float *__restrict__ ws_ml, bf16_t *__restrict__ out,
                                                         int heads, int Tq, int D, int ksplit, long long ldo, long long hso, long long total)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int d4 = D / 4;
    const int c = (int)(i % d4);
    const long long r = i / d4;              // (n * heads + h) * Tq + q
    const int q = (int)(r % Tq);
    const long long nh = r / Tq;
    float m = -INFINITY;
    for (int s = 0; s < ksplit; ++s) m = fmaxf(m, ws_ml[((nh * ksplit + s) * Tq + q) * 2]);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const long long row = (nh * ksplit + s) * Tq + q;
        const float w = __builtin_amdgcn_exp2f(ws_ml[row * 2] - m);
        L += w * ws_ml[row * 2 + 1];
        const f32x4 o = *reinterpret_cast<const f32x4 *>(ws_o + row * D + c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += w * o[j];
    }
    const float inv = 1.0f / L;
    const int h = (int)(nh % heads);
    const long long n = nh / heads;
    bf16x4 ob;
#pragma unroll
    for (int j = 0; j < 4; ++j) ob[j] = (bf16_t)(acc[j] * inv);
    *reinterpret_cast<bf16x4 *>(out + (n * Tq + q) * ldo + (long long)h * hso + c * 4) = ob;
}

// Key split for UNDER-FILLED single-head grids (AE mid-block attention: one head of 384 / 512 channels over T = 4096: 64 workgroups, each
// streaming all 8 MB of K / V at what one CU can take in): split the keys over `ksplit` workgroups so that the grid fills the chip.
static int attn_ksplit(int D, int KT, long long blocks, int Tkv)
{
    if (D < 256 || blocks >= 128 || Tkv < 8 * KT) return 1;
    int s = (int)(256 / blocks);
    if (s > 8) s = 8;
    while (s > 1 && (Tkv + s - 1) / s < 4 * KT) --s;
    return s < 2 ? 1 : s;
}

template <int D, int KT>
static int launch_attn(AttnParams p, const gg_attention_desc *d, hipStream_t stream)
{
    const long long qtiles = (p.Tq + 63) / 64, blocks = qtiles * p.heads * p.N;
    int ks = attn_ksplit(D, KT, blocks, p.Tkv);
    const long long rows = (long long)p.N * p.heads * ks * p.Tq, need = rows * (D + 2) * 4;
    if (ks > 1 && (!d->workspace || d->workspace_bytes < need)) ks = 1;          // no workspace: the unsplit kernel (slower, same result)
    p.ksplit = ks;
    p.kchunk = ks > 1 ? (int)((((p.Tkv + ks - 1) / ks) + KT - 1) / KT * KT) : p.Tkv;
    p.ws_o = (float *)d->workspace;
    p.ws_ml = p.ws_o ? p.ws_o + rows * D : nullptr;
    dim3 grid((unsigned)(qtiles * ks), (unsigned)p.heads, (unsigned)p.N);
    // in-workgroup key split: under-filled grids with at least four K/V tiles per workgroup (the 32x32 attention blocks at batch 1)
    bool ws2 = false;
    if constexpr (D < 256) ws2 = GG_ATTN_WS2 && ks == 1 && blocks <= 256 && p.Tkv >= 4 * KT;
    if constexpr (D < 256) {
        if (ws2) hipLaunchKernelGGL((attn_kernel<D, KT, 2>), grid, dim3(512), 0, stream, p);
    }
    if (!ws2) hipLaunchKernelGGL((attn_kernel<D, KT>), grid, dim3(256), 0, stream, p);
    GG_CHECK_LAUNCH();
    if (ks > 1) {
        const long long total = (long long)p.N * p.heads * p.Tq * (D / 4);
        hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const float *)p.ws_o, (const float *)p.ws_ml,
                           p.out, p.heads, p.Tq, D, ks, p.ldo, p.hso, total);
        GG_CHECK_LAUNCH();
    }
    return GG_OK;
}

extern "C" int64_t gg_attention_workspace_bytes(const gg_attention_desc *d)
{
    if (!d || d->N <= 0 || d->heads <= 0 || d->Tq <= 0 || d->Tkv <= 0) return 0;
    const int KT = d->head_dim == 32 ? 256 : d->head_dim <= 128 ? 64 : 32;
    const long long blocks = (long long)((d->Tq + 63) / 64) * d->heads * d->N;
    const int ks = attn_ksplit(d->head_dim, KT, blocks, d->Tkv);
    return ks > 1 ? (int64_t)d->N * d->heads * ks * d->Tq * (d->head_dim + 2) * 4 : 0;
}

extern "C" int gg_attention_forward(const gg_attention_desc *d, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d || !d->q || !d->k || !d->v || !d->out) GG_FAIL(GG_ERR_BAD_SHAPE, "attention: null pointer");
    if (d->N <= 0 || d->heads <= 0 || d->Tq <= 0 || d->Tkv <= 0) GG_FAIL(GG_ERR_BAD_SHAPE, "attention: empty extent");
    auto al8 = [](long long x) { return (x % 8) == 0; };
    if (!al8(d->ldq) || !al8(d->hsq) || !al8(d->ldk) || !al8(d->hsk) || !al8(d->ldv) || !al8(d->hsv) || !al8(d->ldo) || !al8(d->hso))
        GG_FAIL(GG_ERR_BAD_SHAPE, "attention: strides must be multiples of 8 elements (16-byte rows)");
    AttnParams p;
    p.N = d->N; p.heads = d->heads; p.Tq = d->Tq; p.Tkv = d->Tkv;
    p.ldq = d->ldq; p.hsq = d->hsq; p.ldk = d->ldk; p.hsk = d->hsk; p.ldv = d->ldv; p.hsv = d->hsv; p.ldo = d->ldo; p.hso = d->hso;
    p.scale_log2 = d->scale * 1.4426950408889634f;
    p.q = (const bf16_t *)d->q; p.k = (const bf16_t *)d->k; p.v = (const bf16_t *)d->v; p.out = (bf16_t *)d->out;
    p.ksplit = 1; p.kchunk = d->Tkv; p.ws_o = nullptr; p.ws_ml = nullptr;
    switch (d->head_dim) {
        case 32: return launch_attn<32, 256>(p, d, stream);
        case 64: return launch_attn<64, 64>(p, d, stream);
        case 128: return launch_attn<128, 64>(p, d, stream);
        case 256: return launch_attn<256, 32>(p, d, stream);
        case 384: return launch_attn<384, 32>(p, d, stream);
        case 512: return launch_attn<512, 32>(p, d, stream);
        default: GG_FAIL(GG_ERR_UNSUPPORTED, "attention: head_dim %d not in {32,64,128,256,384,512}", d->head_dim);
    }
}
