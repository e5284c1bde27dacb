// Halo-tile implicit-GEMM convolution for gfx950: 3x3(x3), stride 1, pad 1 (optionally with the nearest x2 upsample
// fused in front), bf16 in / fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// One workgroup owns an output box of 512 positions (3-D: 4x8x16, 2-D: 1x32x16; HB: 256 positions, 4x4x16) x BN = 32*NT output
// channels.
// Per 32-channel chunk of the input:
//   1. the INPUT box that the 27 (9) taps touch (3-D: 6x10x18 rows, 2-D: 34x18; upsample: 4x6x10 / 18x10) is staged ONCE into
//      LDS as 64-byte rows (zero padding, two-source concat, and GroupNorm*SiLU applied here, once per element);
//   2. for every tap the 32x(32*NT) weight tile is streamed into a double-buffered LDS slot by global_load_lds (the packed
//      weight is pre-swizzled, so the linear DMA image is already the bank-conflict-free one) while the previous tap's
//      MFMAs run; the activation operand of tap (kd,kh,kw) is the same LDS box read at a shifted row.
// HBM/L2 traffic per block and chunk: one box (41 KB) instead of 27 gathered tiles (27 x 16 KB) in the generic kernel.
#include <atomic>
#include "gg_conv.h"
#include "gg_posterior.h"
#include <stdlib.h>
#ifndef GG_HALO_G3
#define GG_HALO_G3 1
#endif
#ifndef GG_HALO_G3_2D
#define GG_HALO_G3_2D 1        /* the same for the 2-D kernel at NT >= 3 (AE convs; same-box A/B: decode 4.137 -> 4.05 ms, cond-encode 1.79 -> 1.75 ms) */
#endif
#ifndef GG_HALO_2D_HB1
#define GG_HALO_2D_HB1 1       /* 256-position boxes for under-filled 2-D grids at NT >= 3 (A/B switch) */
#endif
#ifndef GG_HALO_G3_HB1
#define GG_HALO_G3_HB1 1      /* three taps per barrier on the 256-position 3-D boxes too: 53 KiB of LDS, still three workgroups per CU; captured CCDM forward 15.45 -> 14.94 ms (A/B switch) */
#endif
#ifndef GG_HALO_TSTORE
#define GG_HALO_TSTORE 1       /* epilogue stores and residual loads transposed through LDS into full-line runs (0: straight in the accumulator layout) */
#endif
#ifndef GG_HALO_W16_2D
#define GG_HALO_W16_2D 1
#endif
#ifndef GG_HALO_W16_3D
#define GG_HALO_W16_3D 0
#endif
#define GG_HALO_W16(D3, NT, HB) ((GG_HALO_W16_2D && !(D3) && (NT) >= 3 && (HB) != 1) || (GG_HALO_W16_3D && (D3) && ((HB) == 2 || (NT) >= 3)))
#ifndef GG_HALO_DMA_UPPER
#define GG_HALO_DMA_UPPER 0
#endif
#ifndef GG_HALO_PRIO_UPPER
#define GG_HALO_PRIO_UPPER 0
#endif
#ifndef GG_HALO_WPS
#define GG_HALO_WPS(NT) 2      /* measured: 4 waves/SIMD forces scratch spills (NT=2) and is not faster */
#endif

template <int D3, int NT, int UP, int HB, int POST = 0>
__global__ __launch_bounds__((GG_HALO_W16(D3, NT, HB) ? 1024 : (NT <= 2 && HB != 2) ? 256 : 512), (GG_HALO_W16(D3, NT, HB) ? 4 : (HB == 1 && D3) ? 3 : 2)) void conv_halo_kernel(const ConvParams p, const int tiles_d, const int tiles_h, const int tiles_w)
{
    // HB 0: 512-position box 4x8x16 (2-D: 1x32x16); 1: 256 positions 4x4x16 (under-filled 3-D grids); 2: 1024 positions 8x8x16, one
    // 8-wave workgroup per CU (halo redundancy 1.76x instead of 2.1x: less staging work per output)
    // 2-D, HB 1: 256 positions 1x16x16 for grids that 512-position tiles leave under-filled at the widest cout tile (AE 512 / 384 channels
    // @128x128: 128 / 96 workgroups on 256 CUs); 8 waves x 2 position tiles
    constexpr int TD = D3 ? (HB == 2 ? 8 : 4) : 1, TH = D3 ? (HB == 1 ? 4 : 8) : (HB == 1 ? 16 : 32), TW = 16;
    // NT <= 2: 4 waves x 8 position-tiles (128 pos x 32*NT couts per wave, 2 workgroups per CU overlap staging and MFMA);
    // NT >= 3: 8 waves x 4 position-tiles (the accumulator would not fit otherwise)
    // 2-D, NT >= 3 (AE convs at 512^2 / 256^2: 9 taps per staged chunk, one workgroup per CU): 16 waves x 2 position tiles = 4 waves per SIMD
    // at <= 64 accumulator registers each, so that staging round trips and operand reads of one wave hide behind the others' MFMAs
    // (same-box A/B at NT 4: AE decode 3.75 -> 3.60-3.68 ms)
    constexpr int NWAVE = GG_HALO_W16(D3, NT, HB) ? 16 : (NT <= 2 && HB != 2) ? 4 : 8;
    constexpr int TPW = (TD * TH) / NWAVE;
    constexpr int NTHR = NWAVE * 64;
    constexpr int KD = D3 ? 3 : 1;
    constexpr int NTAPS = KD * 9;
    constexpr int HD = D3 ? (UP ? TD / 2 + 2 : TD + 2) : 1;
    constexpr int HH = UP ? TH / 2 + 2 : TH + 2;
    constexpr int HW = UP ? TW / 2 + 2 : TW + 2;
    constexpr int NROWS = HD * HH * HW;
    constexpr int NPIECE = NROWS * 4;
    constexpr int JMAX = (NPIECE + NTHR - 1) / NTHR;
    constexpr int XBYTES = ((NROWS * 64 + 1023) / 1024) * 1024;
    constexpr int WBYTES = NT * 2048;                    // one tap: 32*NT cout rows x 64 B
    // G3: the three kw taps of a (kd, kh) line share ONE barrier and ONE weight DMA round (3 tiles, double-buffered: 6 * WBYTES), where
    // the workgroup is alone on its CU anyway (3-D, 8 waves: the 1024-position boxes and NT >= 3), so the extra LDS costs no residency:
    // 9 instead of 27 barriers per chunk (same-box A/B, CCDM forward @128^3: 16.54 -> 16.42 ms).  Hand-ordered software pipelines of the
    // tap loop on top of it (fragments of tap s+1 requested under the MFMAs of tap s; one or two register sets; immediate-offset
    // addressing) were NOT faster: 21.6 / 16.75 / 18.7 ms, see tools/experiments/README.md and gg_conv_halo_tap_pipeline.hip.txt.
    constexpr bool G3 = (GG_HALO_G3 && D3 && (HB == 2 || (GG_HALO_G3_HB1 && HB == 1) || NT >= 3)) || (GG_HALO_G3_2D && !D3 && NT >= 3);
    constexpr int GSZ = G3 ? 3 : 1;                      // taps per weight slot
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // XBYTES + 2 * GSZ * WBYTES (launch_halo)
    char *xs = smem;
    char *wsm = smem + XBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: tile rows live in SGPRs
    const int fr = lane & 15, fq = lane >> 4;

    // ---- tile coordinates; consecutive (remapped) block ids walk W, then H, then D tiles: neighbours share their halo in L2.
    // XCD-aware remap (blocks b and b+8 share an XCD): give every XCD a contiguous run of tiles when the grid allows it.
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    int t = bid;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int td = t % tiles_d;
    const int n = t / tiles_d;
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;          // output-box origin
    const int g0 = blockIdx.y * NT;

    // input coordinates of halo row (0,0,0)
    const int id0 = D3 ? (UP ? d0 / 2 - 1 : d0 - 1) : 0;
    const int ih0 = UP ? h0 / 2 - 1 : h0 - 1;
    const int iw0 = UP ? w0 / 2 - 1 : w0 - 1;

    // ---- per-thread staging duties: piece = 16 B = 8 channels; the channel piece xq is the same for all of a thread's pieces
    const int xq = tid & 3;

    // ---- per-lane activation-operand row offsets: row(tap) = RD(tile,kd) + RH(tile,kh) + rw[kw]; only rw depends on the lane
    int rw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rw[k] = UP ? ((fr + k + 1) >> 1) : (fr + k);

    // Box image swizzle: chunk ^ f(hw) with f depending only on the position INSIDE a W-line (hw), found by exhaustive search
    // to keep every ds_read_b128 lane group of the three kw taps on 16 distinct 16-byte slots.  Unlike a row-based map it is
    // invariant under the kd / kh shifts (whole W-lines), so an operand address is ONE add: lane_off[kw] + line * (HW * 64)
    // (the row-based map cost 5 VALU per read, 40 per tap; same-box A/B on 64->64 @128^3: 545-565 vs 568-578 us).
    constexpr unsigned FMASK = UP ? 0x3C0u : 0xFC30u;      // f(hw) = 2 for hw in {6..9} (upsample) / {4,5,10..15}
    auto fsw = [&](int hw) -> int { return (int)((FMASK >> hw) & 1u) << 1; };
    int lane_off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) lane_off[k] = rw[k] * 64 + ((fq ^ fsw(rw[k])) * 16);
    f32x4 acc[TPW][2 * NT];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int b = 0; b < 2 * NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight tile DMA: tile (g0.., tap, chunk) is NT contiguous 2 KiB groups strided by ntaps*nchunk*2 KiB; a slot holds GSZ taps
    auto issue_w = [&](int ks, int buf) {               // ks: tap index (GSZ == 1) or index of the first tap of a line (GSZ == 3)
        const int chunk = ks / NTAPS, tap0 = ks - chunk * NTAPS;
        constexpr int NP = GSZ * NT * 2;                // 1 KiB pieces of the slot
        // GG_HALO_DMA_UPPER: only the upper half of the waves (the SIMD partners of waves 0 .. NWAVE/2-1) issue the weight DMAs, so the
        // two waves of a SIMD are not in lockstep after the barrier: one starts its operand reads at once, the other a few DMAs later
        constexpr int DW = (GG_HALO_DMA_UPPER && G3) ? NWAVE / 2 : NWAVE, DW0 = NWAVE - DW;
#pragma unroll
        for (int i = 0; i < (NP + DW - 1) / DW; ++i) {
            const int piece = (wave - DW0) + DW * i;
            if (wave >= DW0 && piece < NP) {
                const int u = piece / (NT * 2), piece1k = piece - u * (NT * 2);
                const int g = piece1k >> 1, half = piece1k & 1;
                const bf16_t *src = p.weight + ((((long long)(g0 + g) * NTAPS + tap0 + u) * p.nchunk + chunk) << 10) + half * 512 + lane * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(wsm + (buf * GSZ + u) * WBYTES + piece1k * 1024), 16, 0, 0);
            }
        }
    };

    const int KS = p.nchunk * NTAPS;
    if (GG_HALO_PRIO_UPPER && G3 && wave >= NWAVE / 2) __builtin_amdgcn_s_setprio(1);      // static priority for the younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    issue_w(0, 0);

    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        // ================= stage the input box of this chunk (all threads) =================
        {
            const bool second = chunk >= p.nchunk1;
            const bf16_t *src = second ? p.src2 : p.src1;
            const int Cs = second ? p.C2 : p.C1;
            const int coff = (second ? chunk - p.nchunk1 : chunk) * 32 + xq * 8;
            f32x4 s0, s1, b0, b1;
            if (p.prologue_act) {
                const long long so = (long long)n * (p.C1 + p.C2) + chunk * 32 + xq * 8;
                s0 = *reinterpret_cast<const f32x4 *>(p.gn_scale + so); s1 = *reinterpret_cast<const f32x4 *>(p.gn_scale + so + 4);
                b0 = *reinterpret_cast<const f32x4 *>(p.gn_shift + so); b1 = *reinterpret_cast<const f32x4 *>(p.gn_shift + so + 4);
            }
            constexpr int JG = 3;                                   // pieces in flight per thread (bounds the VGPR footprint)
            // The piece of step j is box row (tid >> 2) + (NTHR / 4) j: its (hd, hh, hw) is decoded once per chunk and then stepped
            // with carries (no integer division per piece: the staging phase is VALU-bound, in-kernel stamps in DESIGN.md 5.2);
            // padding rows load from a clamped address and are zeroed by a select, so the loads of a round issue back to back.
            constexpr int RSTEP = NTHR / 4;
            constexpr int SD = RSTEP / (HH * HW), SREM = RSTEP % (HH * HW), SH = SREM / HW, SW = SREM % HW;
            int chd, chh, chw;
            {
                const int row0 = tid >> 2;
                chd = row0 / (HH * HW);
                const int rem = row0 - chd * (HH * HW);
                chh = rem / HW;
                chw = rem - chh * HW;
            }
            int crow = tid >> 2;
            const bf16_t *srcn = src + (long long)n * p.D * p.H * p.W * Cs + coff;
#pragma unroll 1
            for (int j0 = 0; j0 < JMAX; j0 += JG) {
                u32x4 v[JG];
                bool ok[JG];
                int lrow[JG], lsw[JG];
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) {
                    const int id = id0 + chd, ih = ih0 + chh, iw = iw0 + chw;
                    ok[jj] = crow < NROWS && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    const unsigned pos = ok[jj] ? (unsigned)((id * p.H + ih) * p.W + iw) : 0u;
                    v[jj] = *reinterpret_cast<const u32x4 *>(srcn + pos * (unsigned)Cs);
                    lrow[jj] = crow;
                    lsw[jj] = fsw(chw);
                    crow += RSTEP;
                    chw += SW; chh += SH; chd += SD;
                    if (chw >= HW) { chw -= HW; chh += 1; }
                    if (chh >= HH) { chh -= HH; chd += 1; }
                }
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) {
                    if (p.prologue_act) {
                        bf16x8 xb = __builtin_bit_cast(bf16x8, v[jj]);
                        bf16x8 yb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float y0 = (float)xb[e] * s0[e] + b0[e], y1 = (float)xb[e + 4] * s1[e] + b1[e];
                            if (p.prologue_act == 1) {
                                y0 = y0 * __builtin_amdgcn_rcpf(1.0f + __expf(-y0));
                                y1 = y1 * __builtin_amdgcn_rcpf(1.0f + __expf(-y1));
                            }
                            yb[e] = (bf16_t)y0;
                            yb[e + 4] = (bf16_t)y1;
                        }
                        v[jj] = __builtin_bit_cast(u32x4, yb);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[jj][e] = ok[jj] ? v[jj][e] : 0u;          // zero padding stays zero
                    if (lrow[jj] < NROWS) *reinterpret_cast<u32x4 *>(xs + lrow[jj] * 64 + (xq ^ lsw[jj]) * 16) = v[jj];
                }
            }
        }
        __syncthreads();          // box visible; (vmcnt(0) inside: the first weight tile of the chunk has landed too)

        // keep the 27x4 operand addresses from being hoisted out of the chunk loop (they would pin >100 VGPRs)
        asm volatile("" : "+v"(lane_off[0]), "+v"(lane_off[1]), "+v"(lane_off[2]));
        // ================= 27 (9) taps from LDS =================
#pragma unroll 1
        for (int kd = 0; kd < KD; ++kd) {
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh) {
                if constexpr (G3) {      // one weight round and one barrier per (kd, kh) line
                    const int ks0 = chunk * NTAPS + (kd * 3 + kh) * 3, gi = ks0 / 3;
                    if (ks0 + 3 < KS) issue_w(ks0 + 3, (gi + 1) & 1);
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tap = (kd * 3 + kh) * 3 + kw;
                    const int ks = chunk * NTAPS + tap;
                    if constexpr (!G3) {
                        if (ks + 1 < KS) issue_w(ks + 1, (ks + 1) & 1);
                    }
                    const char *wb = G3 ? wsm + (((ks / 3) & 1) * 3 + kw) * WBYTES : wsm + (ks & 1) * WBYTES;
                    bf16x8 xf[TPW];
#pragma unroll
                    for (int tt = 0; tt < TPW; ++tt) {
                        const int tile = wave * TPW + tt;             // one W-row of 16 output positions
                        const int od = D3 ? tile / TH : 0, oh = D3 ? tile % TH : tile;
                        const int hd = D3 ? (UP ? ((od + kd + 1) >> 1) : od + kd) : 0;
                        const int hh = UP ? ((oh + kh + 1) >> 1) : oh + kh;
                        xf[tt] = *reinterpret_cast<const bf16x8 *>(xs + (hd * HH + hh) * (HW * 64) + lane_off[kw]);
                    }
#pragma unroll
                    for (int ct = 0; ct < (POST ? 1 : 2 * NT); ++ct) {      // POST: K <= 16 classes = the first 16-cout tile only
                        const int r = ct * 16 + fr;
                        const bf16x8 wf = *reinterpret_cast<const bf16x8 *>(wb + r * 64 + swz64(r, fq) * 16);   // image pre-swizzled at pack time
#pragma unroll
                        for (int tt = 0; tt < TPW; ++tt)
                            acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[tt], acc[tt][ct], 0, 0, 0);
                    }
                    if constexpr (!G3) __syncthreads();  // next tap's weights landed (vmcnt(0)); this tap's slot / the box may be overwritten
                }
                if constexpr (G3) __syncthreads();       // next line's weights landed (vmcnt(0)); this line's slot / the box may be overwritten
            }
        }
    }

    // ================= epilogue: + bias[n] (+ residual) -> bf16 / fp32 =================
    const float *brow = p.bias ? p.bias + (long long)n * p.bias_stride : nullptr;
    // GroupNorm statistics of the NEXT norm (gg_conv_desc.gn_acc): per output channel, sum and sum of squares of the bf16-rounded
    // values this workgroup stores
    const bool stats = p.gn_acc && p.out_dtype != GG_F32;
    float ssum[2 * NT][4], ssq[2 * NT][4];
#pragma unroll
    for (int a = 0; a < 2 * NT; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[a][j] = 0.f; ssq[a][j] = 0.f; }
    // gfx950 counts loads and stores in ONE in-order counter (vmcnt): a load issued behind a store cannot be waited for without waiting
    // for the store's whole round trip.  With the bias / residual load of every tile behind the previous tile's store, the epilogue of
    // a 1024-position box took ~23 k cycles (half a chunk's taps; in-kernel stamps of the same pattern in gg_conv_halo3.hip: 25 k cycles
    // for 32 tiles).  So: the bias vectors and ALL residual pieces are loaded in front of the first store.
    if constexpr (POST) {
        // ===== fused CCDM reverse step (gg_conv_desc.post_xt): the logits never leave the CU =====
        // The accumulator layout gives a lane 4 of a position's K <= 16 logits; through LDS (the box image is dead) every lane gets ALL
        // logits of two of the wave's 16 TPW positions and runs gg_posterior.h on them: the very function of the stand-alone sampler kernel.
        static_assert(NT == 1 && !GG_HALO_W16(D3, NT, HB) && (TPW * 16) % 64 == 0, "fused posterior: one 32-cout group, 64-lane position groups");
        const f32x4 bv = brow ? *reinterpret_cast<const f32x4 *>(brow + g0 * 32 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        float *scrp = reinterpret_cast<float *>(smem + 8192) + wave * (TPW * 16 * 16);        // [TPW * 16 positions][16 logits] fp32
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            f32x4 v = acc[tt][0];
            if (brow) v += bv;
            *reinterpret_cast<f32x4 *>(scrp + (tt * 16 + fr) * 16 + fq * 4) = v;
        }
        const float pa = p.post_scalars[0], pabar = p.post_scalars[1];
        const long long poff = (p.post_draw && !p.post_E && p.post_offset_dev) ? p.post_offset_dev[0] : 0;
#pragma unroll 1
        for (int r = 0; r < TPW * 16 / 64; ++r) {
            const int pp = r * 64 + lane, tt = pp >> 4, pw = pp & 15;
            const int tile = wave * TPW + tt;
            const int od = D3 ? tile / TH : 0, oh = D3 ? tile % TH : tile;
            const long long m = (((long long)n * p.Do + (d0 + od)) * p.Ho + (h0 + oh)) * p.Wo + (w0 + pw);
            const int x = p.post_xt[m];
            float p0[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(scrp + pp * 16 + i * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) p0[i * 4 + j] = q[j];
            }
            const float *erow = p.post_E ? p.post_E + m * p.Cout : nullptr;
            int best;
            if (p.Cout == 14) best = ccdm_posterior_voxel<16, 14>(p0, 1, x, pa, pabar, 14, m, p.post_draw, erow, (uint64_t)p.post_seed, poff, nullptr);
            else best = ccdm_posterior_voxel<16>(p0, 1, x, pa, pabar, p.Cout, m, p.post_draw, erow, (uint64_t)p.post_seed, poff, nullptr);
            p.post_labels_out[m] = best;
            if (p.post_onehot_out) ccdm_onehot_row<16>(p.post_onehot_out + m * p.post_onehot_stride, best, p.Cout);
        }
        return;
    }
    if constexpr (!GG_HALO_W16(D3, NT, HB)) {
        f32x4 bvec[2 * NT];
#pragma unroll
        for (int ct = 0; ct < 2 * NT; ++ct) bvec[ct] = brow ? *reinterpret_cast<const f32x4 *>(brow + g0 * 32 + ct * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        auto out_off = [&](int tt) -> long long {
            const int tile = wave * TPW + tt;
            const int od = D3 ? tile / TH : 0, oh = D3 ? tile % TH : tile;
            const long long m = (((long long)n * p.Do + (d0 + od)) * p.Ho + (h0 + oh)) * p.Wo + (w0 + fr);
            return m * p.Cout_pad + g0 * 32 + fq * 4;
        };
        // all residual pieces of the wave's tiles in flight at once (TPW x 2 NT x 2 registers: 64 where the wave owns 128 positions x 64
        // couts; the operand fragments are dead by now): ONE memory round trip in front of the stores instead of one per tile row
        // (NT >= 3: 64 statistics + 32 bias registers beside the 128 accumulators leave no room for that; there the pieces of tile row
        // tt + 1 are requested before the stores of row tt, so that the wait for them skips those stores)
        // Output stores (bf16) go through LDS: the MFMA accumulator layout gives every lane 4 couts of ONE position, i.e. a store
        // instruction of 16 scattered 32-byte segments, 2 NT of them per tile row; transposed through a per-wave scratch (the box image
        // is dead by now) a tile row leaves as NT instructions of 16 bytes per lane that cover whole 64 NT-byte runs per position (one
        // 2 KiB run where the wave owns all couts).  Scratch rows are padded by 16 B: conflict-free 8-byte writes for NT 1..4.
        // Measured (same-box A/B): 64->64 @128^3 with residual 566 -> 530 us, without 487 -> 476; on the 512-position boxes (NT 3 / 4) and
        // in 2-D it is 1-2 % SLOWER (shorter epilogues, and NT 3 spills 50 registers), so only the 1024-position boxes take it.
        constexpr bool TS = GG_HALO_TSTORE && D3 && HB == 2;
        constexpr int TS_PPR = 4 * NT, TS_RS = 64 * NT + 16;
        char *scr = smem + 8192 + wave * (16 * TS_RS);      // the first 8 KiB take the statistics exchange below
        constexpr bool ALLRES = NT <= 2;
        constexpr int RD = ALLRES ? TPW : 2;
        // the residual comes in the same way: NT 16-byte pieces per lane in memory order, redistributed to the accumulator layout
        // through the wave's scratch rows
        u32x4 rres16[TS ? RD : 1][NT];
        bf16x4 rres[TS ? 1 : RD][2 * NT];
        auto prefetch = [&](int tt) {
            if constexpr (TS) {
                const long long mrow = out_off(tt) - (long long)fr * p.Cout_pad - fq * 4;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int idx = i * 64 + lane, pos = idx / TS_PPR, pc = idx - pos * TS_PPR;
                    rres16[tt % RD][i] = *reinterpret_cast<const u32x4 *>(p.residual + mrow + (long long)pos * p.Cout_pad + pc * 8);
                }
            } else {
                const long long o1 = out_off(tt);
#pragma unroll
                for (int ct = 0; ct < 2 * NT; ++ct) rres[tt % RD][ct] = *reinterpret_cast<const bf16x4 *>(p.residual + o1 + ct * 16);
            }
        };
        if (p.residual) {
#pragma unroll
            for (int tt = 0; tt < (ALLRES ? TPW : 1); ++tt) prefetch(tt);
        }
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const long long ob = out_off(tt);
            if (!ALLRES && p.residual && tt + 1 < TPW) prefetch(tt + 1);
            if (TS && p.residual) {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int idx = i * 64 + lane, pos = idx / TS_PPR, pc = idx - pos * TS_PPR;
                    *reinterpret_cast<u32x4 *>(scr + pos * TS_RS + pc * 16) = rres16[TS ? tt % RD : 0][i];
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2 * NT; ++ct) {
                const int co = g0 * 32 + ct * 16 + fq * 4;
                f32x4 v = acc[tt][ct];
                if (brow) v += bvec[ct];
                const long long o = ob + ct * 16;
                if (p.residual) {
                    const bf16x4 r = TS ? *reinterpret_cast<const bf16x4 *>(scr + fr * TS_RS + (ct * 16 + fq * 4) * 2) : rres[TS ? 0 : tt % RD][ct];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (co + j >= p.Cout) v[j] = 0.f;
                if (p.out_dtype == GG_F32) {
                    *reinterpret_cast<f32x4 *>((float *)p.out + o) = v;
                } else {
                    bf16x4 ob4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ob4[j] = (bf16_t)v[j];
                        const float f = (float)ob4[j];               // what the next norm will read
                        ssum[ct][j] += f;
                        ssq[ct][j] += f * f;
                    }
                    if constexpr (TS) *reinterpret_cast<bf16x4 *>(scr + fr * TS_RS + (ct * 16 + fq * 4) * 2) = ob4;
                    else *reinterpret_cast<bf16x4 *>((bf16_t *)p.out + o) = ob4;
                }
            }
            if (TS && p.out_dtype != GG_F32) {
                // the tile row (16 positions x 32 NT couts) back out of LDS as 16-byte pieces in memory order: position = idx / PPR
                const long long mrow = ob - (long long)fr * p.Cout_pad - fq * 4;          // element offset of (position 0, cout g0 * 32)
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int idx = i * 64 + lane, pos = idx / TS_PPR, pc = idx - pos * TS_PPR;
                    const u32x4 v16 = *reinterpret_cast<const u32x4 *>(scr + pos * TS_RS + pc * 16);
                    *reinterpret_cast<u32x4 *>((bf16_t *)p.out + mrow + (long long)pos * p.Cout_pad + pc * 8) = v16;
                }
            }
        }
    } else {
        // 16 waves at a 128-register budget (64 accumulator + 64 statistics registers at NT 4): no room for whole vector sets, so the bias
        // and residual piece of tile k + 1 are requested before tile k is stored -- ONE tile of look-ahead (6 registers): the wait for them
        // then skips that store instead of waiting for its round trip
        constexpr int NTILE = TPW * 2 * NT;
        auto tile_off = [&](int k, long long &o, int &co) {
            const int tt = k / (2 * NT), ct = k - tt * (2 * NT);
            const int tile = wave * TPW + tt;
            const int od = D3 ? tile / TH : 0, oh = D3 ? tile % TH : tile;
            const long long m = (((long long)n * p.Do + (d0 + od)) * p.Ho + (h0 + oh)) * p.Wo + (w0 + fr);
            co = g0 * 32 + ct * 16 + fq * 4;
            o = m * p.Cout_pad + co;
        };
        f32x4 bnx = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x4 rnx = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        {
            long long o0; int c0;
            tile_off(0, o0, c0);
            if (brow) bnx = *reinterpret_cast<const f32x4 *>(brow + c0);
            if (p.residual) rnx = *reinterpret_cast<const bf16x4 *>(p.residual + o0);
        }
#pragma unroll
        for (int k = 0; k < NTILE; ++k) {
            const int tt = k / (2 * NT), ct = k - tt * (2 * NT);
            long long o; int co;
            tile_off(k, o, co);
            const f32x4 bcur = bnx;
            const bf16x4 rcur = rnx;
            if (k + 1 < NTILE) {
                long long o1; int c1;
                tile_off(k + 1, o1, c1);
                if (brow) bnx = *reinterpret_cast<const f32x4 *>(brow + c1);
                if (p.residual) rnx = *reinterpret_cast<const bf16x4 *>(p.residual + o1);
            }
            f32x4 v = acc[tt][ct];
            if (brow) v += bcur;
            if (p.residual) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += (float)rcur[j];
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
                    const float f = (float)ob[j];               // what the next norm will read
                    ssum[ct][j] += f;
                    ssq[ct][j] += f * f;
                }
                *reinterpret_cast<bf16x4 *>((bf16_t *)p.out + o) = ob;
            }
        }
    }
    if (stats) {
        // the 16 positions of a lane row by DPP moves, the waves through LDS in a fixed order (the box / weight image is dead: every
        // wave has passed the last tap's barrier), then 64-bit fixed-point integer atomics: the sums are exact and order-independent
        float *statp = reinterpret_cast<float *>(smem);          // [NWAVE][32 * NT couts][sum | sumsq]
#pragma unroll
        for (int ct = 0; ct < 2 * NT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = gg_row16_sum(ssum[ct][j]), b = gg_row16_sum(ssq[ct][j]);
                if (fr == 0) {
                    float *q = statp + ((wave * (32 * NT) + ct * 16 + fq * 4 + j) << 1);
                    q[0] = a;
                    q[1] = b;
                }
            }
        __syncthreads();
        if (tid < 64 * NT) {
            const int which = tid & 1, c = tid >> 1;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) t += statp[((w * (32 * NT) + c) << 1) + which];
            const long long fx = __double2ll_rn((double)t * (double)(which ? GG_ACC_SQ_SCALE : GG_ACC_SUM_SCALE));
            const int stripe = blockIdx.x % GG_ACC_STRIPES_HALO;
            atomicAdd(reinterpret_cast<unsigned long long *>(p.gn_acc + ((((long long)n * GG_ACC_STRIPES_HALO + stripe) * p.Cout_pad + g0 * 32 + c) * 2 + which)),
                      (unsigned long long)fx);
        }
    }
}

template <int D3, int NT, int UP, int HB = 0, int POST = 0>
static int launch_halo(const ConvParams &p, hipStream_t stream)
{
    constexpr int TD = D3 ? (HB == 2 ? 8 : 4) : 1, TH = D3 ? (HB == 1 ? 4 : 8) : (HB == 1 ? 16 : 32), TW = 16;
    constexpr int HD = D3 ? (UP ? TD / 2 + 2 : TD + 2) : 1, HH = UP ? TH / 2 + 2 : TH + 2, HW = UP ? TW / 2 + 2 : TW + 2;
    constexpr bool G3 = (GG_HALO_G3 && D3 && (HB == 2 || (GG_HALO_G3_HB1 && HB == 1) || NT >= 3)) || (GG_HALO_G3_2D && !D3 && NT >= 3);
    constexpr int LDSB = ((HD * HH * HW * 64 + 1023) / 1024) * 1024 + (G3 ? 6 : 2) * NT * 2048;
    // the attribute is per device: one bit per device ordinal (setting it twice from two threads is harmless)
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GG_ERR_HIP;
    const unsigned long long dev_bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & dev_bit)) {
        if (hipFuncSetAttribute((const void *)conv_halo_kernel<D3, NT, UP, HB, POST>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)
            return GG_ERR_UNSUPPORTED;
        attr_mask.fetch_or(dev_bit, std::memory_order_release);
    }
    const int tiles_d = p.Do / TD, tiles_h = p.Ho / TH, tiles_w = p.Wo / TW;
    dim3 grid((unsigned)(p.N * tiles_d * tiles_h * tiles_w), (unsigned)(p.Cout_pad / (32 * NT)));
    hipLaunchKernelGGL((conv_halo_kernel<D3, NT, UP, HB, POST>), grid, dim3(GG_HALO_W16(D3, NT, HB) ? 1024 : (NT <= 2 && HB != 2) ? 256 : 512), LDSB, stream, p, tiles_d, tiles_h, tiles_w);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

template <int D3, int UP>
static int dispatch_nt(const ConvParams &p, int NT, hipStream_t stream)
{
    switch (NT) {
        case 4: return launch_halo<D3, 4, UP>(p, stream);
        case 3: return launch_halo<D3, 3, UP>(p, stream);
        case 2: return launch_halo<D3, 2, UP>(p, stream);
        default: return launch_halo<D3, 1, UP>(p, stream);
    }
}

// team kernel for filled 3-D grids (gg_conv_halo3.hip); same contract
int gg_conv_halo3_try(const ConvParams &p, hipStream_t stream);

// widest cout tile (fewest re-stagings of the box) that still gives >= 256 workgroups; else the widest with >= 128.
// 3-D: 27 taps per staged box, so filling the chip comes first (>= 256 workgroups, else >= 128).  2-D: only 9 taps per staged
// box, staging dominates, so the widest cout tile that still gives 128 workgroups wins (AE 512->512 @128x128: NT 4 x 128
// workgroups instead of NT 2 x 256)
static int halo_pick_nt(bool d3, int G, long long tiles)
{
    // (3-D grids that only fill the chip with one 32-cout group per workgroup -- 128 couts @32^3: 64 tiles x 4 groups -- on two groups
    // per 256-position box instead, the same 256 workgroups: captured CCDM forward 15.40 -> 15.53 ms, not kept)
    for (int want : {!d3 ? 128 : 256, 128})
        for (int cand : {4, 3, 2, 1})
            if (G % cand == 0 && tiles * (G / cand) >= want) return cand;
    return 1;
}

// Whether GroupNorm * SiLU in front of this conv is cheaper as its OWN pass than fused into the staging pass (the caller has checked
// that gg_conv_halo_try takes the shape).  Fused, every staged element costs ~25 vector instructions incl. two transcendentals, once per
// cout group and per box that holds it (halo redundancy 1.2x in 2-D, 2.1x for the 512-position 3-D boxes, 1.76x for the 1024-position
// ones), in issue slots the MFMAs of the co-resident waves want; the separate pass is one HBM-bound read + write.  Measured on one box
// (tools/experiments/probe_halo_2d_pro.py; fused vs prologue-free conv + apply pass, us):
//   2-D 128->128 @512^2 112 vs 91 + 21 | 256->256 @256^2 96 vs 79 + 14 | 512->512 @128^2 136 vs 110 + 12 | 512->512 @64^2 59 vs 36 + 12
//       96->96 @512^2 69 vs 60 + 17   | 192->192 @256^2 61 vs 53 + 15 | 384->384 @128^2 88 vs 69 + 12   | 384->384 @64^2 47 vs 30 + 12
//   3-D 64->64 @128^3 477 vs 416 + 98 | 128->128 @64^3 222 vs 191 + 21 | 256->128 @64^3 398 vs 343 + 40 | 256->256 @32^3 156 vs 125 + 15
//       512->256 @32^3 288 vs 236 + 13
// => separate wherever the box is re-staged by >= 3 cout groups or by 2 groups of the 128-cout tile.  The one-group 3-D shapes (CCDM
// 128 channels @64^3 / @32^3) look like wins above (-4 %), but IN the captured CCDM forward the rule applied to them lost: 15.47 -> 15.55 ms
// (their apply pass reads a tensor the memory-side cache no longer holds); they stay fused.  With the rule as it is (captured, same box):
// AE decode 3.51 -> 3.44 ms, cond-encode 1.458 -> 1.423 ms.
bool gg_conv_halo_prefers_separate_norm(const ConvParams &p)
{
    const bool d3 = (p.kd == 3);
    const int TD = d3 ? 4 : 1, TH = d3 ? 8 : 32, TW = 16;
    const int G = p.Cout_pad / 32;
    const long long tiles = (long long)p.N * (p.Do / TD) * (p.Ho / TH) * (p.Wo / TW);
    const int NT = halo_pick_nt(d3, G, tiles);
    const int groups = G / NT;
    return groups >= 3 || (groups == 2 && NT == 4);
}

static bool halo_uses_1024_box(const ConvParams &p, bool d3, int NT, long long blocks)
{
    return d3 && NT <= 2 && !p.upsample && (p.Do % 8) == 0 && (p.path_hint == 6 || ((p.path_hint == 0 || p.path_hint == 8) && blocks >= 512));
}

// The CCDM reverse step as the epilogue of the head conv: the 1024-position 3-D box kernel with ONE 32-cout group holding all K <= 16 classes
bool gg_conv_halo_fuses_posterior(const ConvParams &p)
{
    if (!(p.kd == 3 && p.kh == 3 && p.kw == 3) || p.stride != 1 || p.pad != 1 || p.upsample) return false;
    if (p.Cout > 16 || p.Cout_pad != 32 || p.out_dtype != GG_F32 || p.residual || p.gn_acc) return false;
    if (p.Wo % 16 || p.Ho % 8 || p.Do % 8) return false;
    const long long tiles = (long long)p.N * (p.Do / 4) * (p.Ho / 8) * (p.Wo / 16);
    if (p.path_hint != 6 && tiles < 128) return false;      // (the under-filled-grid gate of gg_conv_halo_try)
    return halo_uses_1024_box(p, true, 1, tiles);
}

// Returns GG_ERR_UNSUPPORTED (silently, no error text) when the shape is outside the envelope: the caller then uses the
// generic gather kernel.  stream == (hipStream_t)-1: dry run (only answers whether the halo kernel would be used).
int gg_conv_halo_try(const ConvParams &p, hipStream_t stream)
{
    const bool d3 = (p.kd == 3);
    if (!(p.kh == 3 && p.kw == 3 && (p.kd == 3 || p.kd == 1))) return GG_ERR_UNSUPPORTED;
    if (p.stride != 1 || p.pad != 1) return GG_ERR_UNSUPPORTED;
    if (!d3 && p.D != 1) return GG_ERR_UNSUPPORTED;
    // path_hint 7 (tests, A/B probes): the team kernel wherever its envelope allows; 8: production dispatch WITHOUT the team kernel.
    // GG_HALO3_DEFAULT decides what production (path_hint 0) does.
#ifndef GG_HALO3_DEFAULT
#define GG_HALO3_DEFAULT 0
#endif
    if (d3 && !p.post_xt && ((GG_HALO3_DEFAULT && p.path_hint == 0) || p.path_hint == 7)) {
        const int rc3 = gg_conv_halo3_try(p, stream);
        if (rc3 != GG_ERR_UNSUPPORTED) return rc3;
    }
    const int TD = d3 ? 4 : 1, TH = d3 ? 8 : 32, TW = 16;
    if (p.Wo % TW || p.Ho % TH || p.Do % TD) return GG_ERR_UNSUPPORTED;
    const int G = p.Cout_pad / 32;
    const long long tiles = (long long)p.N * (p.Do / TD) * (p.Ho / TH) * (p.Wo / TW);
    constexpr int wide2d = 1;
    const int NT = halo_pick_nt(d3, G, tiles);
    const long long blocks = tiles * (G / NT);
    // under-filled grids: the box / split-K gather paths are faster (2-D under one workgroup per CU: AE 512->512 @64x64 is 136 us
    // here at 128 workgroups); path_hint 1 / 4 / 6 (tests) lift the gate so that small shapes run on this kernel
    const long long min_blocks = (d3 || (wide2d && NT > 1)) ? 128 : 256;
    if (p.path_hint != 1 && p.path_hint != 4 && p.path_hint != 6 && p.path_hint != 7 && blocks < min_blocks) return GG_ERR_UNSUPPORTED;
    if (stream == (hipStream_t)-1) return GG_OK;
    // 3-D grids of at most one 512-position workgroup per CU: 256-position boxes (HB: 4x4x16, three workgroups per CU) double the
    // grid; same-box A/B 256->256 @32^3: 141 vs 156 us.  On filled grids the two box sizes are within +-3 % (64->64 @128^3
    // 525-534 vs 519-571 us, 192->64 1464-1467 vs 1370-1442 us), so those keep the box with the smaller halo.
    // path_hint (tests): 1 = always the 512-position box, 4 = the 256-position box wherever it is instantiated (NT <= 2), 6 = the 1024-one.
    // 3-D grids that still give every CU a workgroup with 1024-position boxes (8x8x16, one 8-wave workgroup per CU, 124 KiB of LDS):
    // the halo redundancy drops from 2.1x to 1.76x, i.e. less staging (loads, GroupNorm*SiLU, LDS writes) per output.  Same-box A/B at
    // 128^3 with the fused prologue: 64->64 500 -> 468 us, 192->64 1338 -> 1240 us, 32->64 300 -> 280 us (bit-identical results);
    // 256->256 @32^3 would lose (177 vs 156 us: half the workgroups), hence the grid condition.  path_hint 6 (tests) forces it.
    if (halo_uses_1024_box(p, d3, NT, blocks)) {
        if (NT == 2) return launch_halo<1, 2, 0, 2>(p, stream);
        if (p.post_xt) return launch_halo<1, 1, 0, 2, 1>(p, stream);       // UNet head with the CCDM reverse step as its epilogue
        return launch_halo<1, 1, 0, 2>(p, stream);
    }
    if (p.post_xt) return GG_ERR_UNSUPPORTED;      // (gg_conv_forward has asked gg_conv_fuses_posterior: not reached)
    if (d3 && NT <= 2 && (p.path_hint == 4 || ((p.path_hint == 0 || p.path_hint == 8) && blocks <= 256))) {
        if (NT == 2) return p.upsample ? launch_halo<1, 2, 1, 1>(p, stream) : launch_halo<1, 2, 0, 1>(p, stream);
        return p.upsample ? launch_halo<1, 1, 1, 1>(p, stream) : launch_halo<1, 1, 0, 1>(p, stream);
    }
    if (d3) return p.upsample ? dispatch_nt<1, 1>(p, NT, stream) : dispatch_nt<1, 0>(p, NT, stream);
    // 2-D grids that the 512-position tiles under-fill at the widest cout tile: 256-position boxes, twice the workgroups (path_hint 4: tests)
    if (GG_HALO_2D_HB1 && NT >= 3 && (p.path_hint == 4 || ((p.path_hint == 0 || p.path_hint == 8) && blocks < 224))) {
        if (NT == 4) return p.upsample ? launch_halo<0, 4, 1, 1>(p, stream) : launch_halo<0, 4, 0, 1>(p, stream);
        return p.upsample ? launch_halo<0, 3, 1, 1>(p, stream) : launch_halo<0, 3, 0, 1>(p, stream);
    }
    return p.upsample ? dispatch_nt<0, 1>(p, NT, stream) : dispatch_nt<0, 0>(p, NT, stream);
}
