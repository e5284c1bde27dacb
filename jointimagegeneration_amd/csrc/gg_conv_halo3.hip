// Team halo conv for gfx950: 3x3x3, stride 1, pad 1 on large 3-D extents (the 128^3 / 64^3 levels of the CCDM UNet,
// ccdm/ddpm/models/unet_openai/unet.py:188-262 ResBlock convs), bf16 in / fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// Why a second kernel: in conv_halo_kernel (gg_conv_halo.hip) every wave of a CU stages the input box of a chunk (global loads,
// GroupNorm*SiLU, LDS writes: ~21 k cycles, matrix pipes idle) and then every wave runs the 27 taps (~27.6 k cycles); in-kernel
// stamps showed the phases in lockstep chip-wide, i.e. the matrix pipes idle 43 % of the time (tools/experiments/README.md).
// Here ONE persistent 8-wave workgroup per CU is split into two TEAMS of four waves (one wave per SIMD each).  Each team owns a
// 4x8x16 output box (the two halves of an 8x8x16 box) with a 6x10x18-row input box of its own in LDS, and the teams run in
// ANTIPHASE: while team A runs the taps of its chunk (hand-scheduled inline assembly, gg_conv_halo3_asm.inc: operands of tap s+1
// are read under the MFMAs of tap s, so one wave per SIMD keeps the pipe busy), team B stages its next chunk (or stores its
// finished box) in the issue slots the MFMAs leave free -- then they swap.  Both teams pass the same 10 workgroup barriers per
// phase (one per (kd, kh) line + one at the end), so the antiphase is by construction, not by luck of the dispatcher.
//
// Results are bit-identical to conv_halo_kernel: same MFMA, same tap / chunk order per accumulator, same epilogue arithmetic.
#include <atomic>
#include <type_traits>
#include "gg_conv.h"
// wave priorities of the two roles (s_setprio): what the tapping team's waves run at, and the staging team's (A/B-measured)
#ifndef GG_H3_TPRIO
#define GG_H3_TPRIO "0"
#endif
#ifndef GG_H3_SPRIO
#define GG_H3_SPRIO 0
#endif
#ifdef GG_H3_ABL_NOMFMA            /* timing ablation only (wrong results): the tap phase without its MFMAs */
#define GG_H3_M(x) ""
#else
#define GG_H3_M(x) x
#endif
#include "gg_conv_halo3_asm.inc"

namespace {
constexpr int H3_HH = 10, H3_HW = 18;                       // box lines per plane, positions per line (8 + 2, 16 + 2)
constexpr int H3_NROWS = 6 * H3_HH * H3_HW;                 // 1080 rows of 64 B (6 planes)
constexpr int H3_XB = 69632;                                // one team's box, rounded up to 1 KiB
constexpr int H3_WOFF = 2 * H3_XB;                          // weight slots behind the two boxes
constexpr int H3_WSLOT = 3 * 4096;                          // one (kd, kh) line: 3 taps x (64 couts x 64 B)
constexpr int H3_LDS = H3_WOFF + 2 * H3_WSLOT;              // 163 840 B = all of the CU's LDS
[[maybe_unused]] constexpr int H3_SCRATCH = 65536;                           // statistics scratch inside the team's box (rows >= 1024: written last by the staging)

// workgroup barrier as inline assembly: a compiler-level memory barrier too, and no implicit s_waitcnt (the staging team keeps its
// global loads in flight across the line barriers)
__device__ __forceinline__ void h3_bar() { asm volatile("s_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned h3_lds_addr(const void *p)
{
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void *)p;
}
}  // namespace

template <int PRO>       // gg_conv_desc.prologue_act: 0 none, 1 GroupNorm * SiLU, 2 GroupNorm affine only (compile-time: the staging code has no branch)
__global__ __launch_bounds__(512, 2) void conv_halo3_team_kernel(const ConvParams p, const int tiles_d, const int tiles_h, const int tiles_w,
                                                                  const int ncg, const int nitems)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, w = wave & 3;               // wave-uniform: every branch on them is a scalar branch
    const unsigned lds0 = h3_lds_addr(smem);

    // W-line swizzle of the box image (gg_conv_halo.hip): chunk ^ f(hw), f = 2 for hw in {4,5,10..15}
    constexpr unsigned FMASK = 0xFC30u;
    auto fsw = [&](int hw) -> int { return (int)((FMASK >> hw) & 1u) << 1; };

    const int m0b = (int)lds0 + H3_WOFF + w * 1024;
    const int nchunk = p.nchunk;
    const int ws = 3 * nchunk * 2048;                       // DMA source step per (kd, kh) line
    const int lw0 = (((w >> 1) * 27 * nchunk) << 11) + (w & 1) * 1024;                 // this wave's 1 KiB piece of a tap tile

    // ---- this workgroup's items (persistent): virtual block ids b, b + grid, ... remapped so that an XCD owns a contiguous item range
    const int grid = gridDim.x;
    const int K = (nitems - (int)blockIdx.x + grid - 1) / grid;
    const int J = K * nchunk;
    const bool remap = !(nitems & 7) && !(grid & 7);
    auto item_of = [&](int k) -> int {
        int v = (int)blockIdx.x + k * grid;
        if (remap) v = (v & 7) * (nitems >> 3) + (v >> 3);
        return v;
    };
    struct Item { int n, d0, h0, w0, g0, idx; };
    auto decode = [&](int item) -> Item {
        Item it;
        it.idx = item;
        const int cg = item % ncg;
        int t = item / ncg;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        const int td = t % tiles_d;
        it.n = t / tiles_d;
        it.d0 = td * 8 + team * 4; it.h0 = th * 8; it.w0 = tw * 16; it.g0 = cg * 2;
        return it;
    };
    auto wbase_off = [&](int g0, int chunk) -> int { return ((g0 * 27) * nchunk + chunk) << 11; };

#ifdef GG_H3_STAMPS
    // diagnostic build: s_memtime stamps of workgroup 0, waves 0 (team A) and 4 (team B): [wave sel][phase 0..63][slot 0..31] through p.ws
    long long *stampb = (p.ws && blockIdx.x == 0 && w == 0) ? reinterpret_cast<long long *>(p.ws) + team * 64 * 32 : nullptr;
    int stamp_phase = 0;
#define H3_STAMP(slot) do { if (stampb && stamp_phase < 64 && (tid & 63) == 0) stampb[stamp_phase * 32 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define H3_NEXT_PHASE() (++stamp_phase)
#else
#define H3_STAMP(slot) do { } while (0)
#define H3_NEXT_PHASE() do { } while (0)
#endif
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    int jk = 0, jc = 0;                                     // the team's current job: item index k, chunk c
    const bool stats = p.gn_acc != nullptr;
    const bool has_bias = p.bias != nullptr, has_res = p.residual != nullptr;

    // ======================================= P-phase: in-place pass over the box a team has just staged, by ALL EIGHT waves =======================================
    // GroupNorm affine (* SiLU) of every staged element and zeros in the padding rows, in place in LDS.  Next to the other team's MFMAs this
    // pass is slow (in-kernel stamps: ~1000 cycles per 16-byte piece and wave instead of ~350: one wave per SIMD, every LDS / transcendental
    // latency exposed, and 1.8x slower again beside the matrix stream, whatever the MFMA shape), and there it made the staging team the
    // critical path (24 k cycles against 15 k of taps).  Between two tap phases, with both teams' waves on it (two per SIMD), the whole box
    // takes ~3 k cycles: the matrix pipes idle for that long, instead of waiting 9 k for the staging team.
    auto pphase = [&](const int owner, const int pk, const int pc, const bool staged) {
#ifdef GG_H3_STAMPS
        if (stampb && stamp_phase > 0 && stamp_phase <= 64 && (tid & 63) == 0) stampb[(stamp_phase - 1) * 32 + 24] = (long long)__builtin_amdgcn_s_memtime();
#endif
        if (staged) {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            const int lane = tl & 63, slot = lane & 3;
            // item of the OWNER team (the two teams stage the two halves of the same 8x8x16 box)
            Item it = decode(item_of(pk));
            it.d0 += (owner - team) * 4;
            const int id0 = it.d0 - 1, ih0 = it.h0 - 1, iw0 = it.w0 - 1;
            const int pD = gg_pin(p.D), pH = gg_pin(p.H), pW = gg_pin(p.W);          // (else re-read from the kernarg segment in every round)
            char *bx = smem + owner * H3_XB;
            const float *ssb = reinterpret_cast<const float *>(bx + H3_NROWS * 64);
            // piece r of this wave: DMA piece 8 r + wave = box rows 16 (8 r + wave) + (lane >> 2): 128 rows per round = 7 lines + 2 positions
            const int rowb = wave * 16 + (lane >> 2);
            int chd = rowb / (H3_HH * H3_HW), chh, chw;
            {
                const int rem = rowb - chd * (H3_HH * H3_HW);
                chh = rem / H3_HW;
                chw = rem - chh * H3_HW;
            }
            // interior boxes of a conv without prologue need nothing at all (uniform test)
            const bool edge = id0 < 0 || ih0 < 0 || iw0 < 0 || id0 + 6 > pD || ih0 + H3_HH > pH || iw0 + H3_HW > pW;
            if (PRO != 0 || edge) {
                // two-stage software pipeline over the rounds (compile-time indices): the LDS reads of round r + 1 are in flight while round r
                // is computed, and no branch stands between them: a lone pair of waves per SIMD has nothing else to hide the LDS latency with
                struct Piece { bf16x8 x; f32x4 s0, s1, b0, b1; int keep; };
                Piece pa, pb;
                auto load = [&](Piece &q, const int r) {
                    const int id = id0 + chd, ih = ih0 + chh, iw = iw0 + chw;
                    const int o = (int)((unsigned)id < (unsigned)pD) & (int)((unsigned)ih < (unsigned)pH) & (int)((unsigned)iw < (unsigned)pW);
                    q.keep = -o;                            // all ones inside the tensor, zero in the padding
                    const int gq = slot ^ fsw(chw);
                    const int row = rowb + 128 * r < H3_NROWS ? rowb + 128 * r : rowb;       // (beyond the box: a harmless re-read, never written back)
                    q.x = *reinterpret_cast<const bf16x8 *>(bx + row * 64 + slot * 16);
                    if constexpr (PRO != 0) {
                        q.s0 = *reinterpret_cast<const f32x4 *>(ssb + gq * 8); q.s1 = *reinterpret_cast<const f32x4 *>(ssb + gq * 8 + 4);
                        q.b0 = *reinterpret_cast<const f32x4 *>(ssb + 32 + gq * 8); q.b1 = *reinterpret_cast<const f32x4 *>(ssb + 32 + gq * 8 + 4);
                    }
                    chw += 2; chh += 7;
                    if (chw >= H3_HW) { chw -= H3_HW; chh += 1; }
                    if (chh >= H3_HH) { chh -= H3_HH; chd += 1; }
                };
                auto work = [&](const Piece &q, const int r, const bool guard) {
                    u32x4 t = __builtin_bit_cast(u32x4, q.x);
                    if constexpr (PRO != 0) {
                        bf16x8 yb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float y0 = (float)q.x[e] * q.s0[e] + q.b0[e], y1 = (float)q.x[e + 4] * q.s1[e] + q.b1[e];
                            if constexpr (PRO == 1) {
                                y0 = y0 * __builtin_amdgcn_rcpf(1.0f + __expf(-y0));
                                y1 = y1 * __builtin_amdgcn_rcpf(1.0f + __expf(-y1));
                            }
                            yb[e] = (bf16_t)y0;
                            yb[e + 4] = (bf16_t)y1;
                        }
                        t = __builtin_bit_cast(u32x4, yb);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] &= (unsigned)q.keep;                   // zero padding stays zero
                    if (!guard || rowb + 128 * r < H3_NROWS) *reinterpret_cast<u32x4 *>(bx + (rowb + 128 * r) * 64 + slot * 16) = t;
                };
                // a ROLLED loop of two rounds per iteration: 1.5 KB of code that stays in the instruction cache, instead of 7 KB of cold
                // straight-line code per call site
                load(pa, 0);
#pragma unroll 1
                for (int r = 0; r < 8; r += 2) {
                    load(pb, r + 1);
                    work(pa, r, false);
                    load(pa, r + 2);
                    work(pb, r + 1, false);
                }
                if (wave < 4) work(pa, 8, true);            // pieces 64 .. 67 (the last one covers 8 rows only)
            }
        }
#ifdef GG_H3_STAMPS
        if (stampb && stamp_phase > 0 && stamp_phase <= 64 && (tid & 63) == 0) stampb[(stamp_phase - 1) * 32 + 25] = (long long)__builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef GG_H3_STAMPS
        if (stampb && stamp_phase > 0 && stamp_phase <= 64 && (tid & 63) == 0) stampb[(stamp_phase - 1) * 32 + 26] = (long long)__builtin_amdgcn_s_memtime();
#endif
    };

    auto dummy_phase = [&]() {                              // pipeline fill (team B) / drain (team A): keep the barrier count
#pragma unroll
        for (int i = 0; i < 9; ++i) h3_bar();
        __syncthreads();
    };

    // ======================================= S-phase: staging of job (jk, jc) + (epilogue of the finished item) =======================================
    // Staging = LDS-DMA + in-place pass.  A wave's 17 DMA pieces (16 box rows x 64 B each, lane -> 16 bytes; the W-line swizzle of the
    // box image is applied on the SOURCE side: the lane at slot s of row r fetches channel piece s ^ f(hw(r))) are issued six ahead of
    // the pass that consumes them -- not all at once: the CU's address unit takes ~90 cycles per scattered 1 KiB piece (stamps: 6 k
    // cycles for the 68 pieces of a box, during which the issuing waves are parked), so the issue is spread under the passes.
    // Intervals 1..6 run the in-place pass, three pieces per interval, each lane on exactly the 16 bytes it fetched itself (no barrier
    // between landing and pass): wait for the interval's last piece (s_waitcnt vmcnt(n): DMAs land in issue order), read the pieces
    // back, GroupNorm affine (* SiLU), zero the padding rows, write them back.  All vector-memory instructions of the staging are inline
    // assembly with hand-counted waits: the compiler's own wait-count insertion drained EVERY outstanding load at the first use of one
    // (s_waitcnt vmcnt(0) behind any branch), which made a register-staged version latency-bound (27 k cycles per phase).
    // The epilogue of the finished item runs in intervals 6..9, a quarter per interval, its loads one part ahead of their use.
    auto sphase = [&](const bool do_epi, const bool stage) {
        // Lane-derived values are re-derived per phase from a laundered thread id: computed once, the compiler keeps them in VGPRs across
        // the tap phase, whose assembly needs 231 of the 256.
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const int lane = tl & 63, fr = lane & 15, fq = lane >> 4;
        if (GG_H3_SPRIO) __builtin_amdgcn_s_setprio(GG_H3_SPRIO);
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // (compiler-visible vmcnt(0): nothing of this wave is outstanding here; see the T-phase)
        const int slot = lane & 3, row0 = w * 16 + (lane >> 2);             // piece k of this lane: box row row0 + 64 k, 16-byte slot `slot`
        const Item it = decode(item_of(stage ? jk : 0));
        const bool second = jc >= p.nchunk1;
        const bf16_t *src = second ? p.src2 : p.src1;
        const int Cs = second ? p.C2 : p.C1;
        const int coff = (second ? jc - p.nchunk1 : jc) * 32;
        const int id0 = it.d0 - 1, ih0 = it.h0 - 1, iw0 = it.w0 - 1;
        const bf16_t *srcn = src + (long long)it.n * p.D * p.H * p.W * Cs + coff;
        // ([scale 32 | shift 32] of the chunk are parked in the 512 spare bytes behind the box rows: smem + team * H3_XB + H3_NROWS * 64)
        // box row of piece k is row0 + 64 k: (hd, hh, hw) stepped with carries (64 = 3 * 18 + 10)
        int chd = row0 / (H3_HH * H3_HW), chh, chw;
        {
            const int rem = row0 - chd * (H3_HH * H3_HW);
            chh = rem / H3_HW;
            chw = rem - chh * H3_HW;
        }
        auto dma = [&](auto kc) {                           // issue piece k (pieces are issued in order: the coordinates step along)
            constexpr int k = decltype(kc)::value;
            const int id = id0 + chd, ih = ih0 + chh, iw = iw0 + chw;
            const bool o = (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const int f = fsw(chw);
            const unsigned pos = o ? (unsigned)((id * p.H + ih) * p.W + iw) : 0u;
            const bf16_t *gp = srcn + pos * (unsigned)Cs + ((slot ^ f) * 8);
            const int ldsd = (int)lds0 + team * H3_XB + (w * 16 + 64 * k) * 64;              // 16 rows = 1 KiB per wave instruction
            if (k < 16 || row0 < H3_NROWS - 1024)           // (the last piece covers 8 rows only: the upper lanes would land in the scale / shift rows)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(ldsd) : "memory", "m0");
            chw += 10; chh += 3;
            if (chw >= H3_HW) { chw -= H3_HW; chh += 1; }
            if (chh >= H3_HH) { chh -= H3_HH; chd += 1; }
        };
        // ---- epilogue of the finished item, a quarter (two W-lines of this wave's plane) per interval 6..9: + bias[n] (+ residual) -> bf16,
        // GroupNorm statistics of what is stored.  The loads of part k + 1 are issued before part k is computed (straight-line code: the
        // compiler counts them exactly), and no load stands behind a store it must wait for: gfx950 counts loads and stores in ONE in-order
        // counter (stamps: 25 k cycles for 32 tiles with a load behind every store).  A missing bias / residual is loaded from the output
        // tensor's own (valid) addresses and discarded by a select: no branch.
        Item ei = it;
        if (do_epi) ei = decode(item_of(jk - 1));
        float ssum[4][4], ssq[4][4];
        f32x4 bv[4];
        bf16x4 rv[2][2][4];                                 // [part parity][line][cout tile]
        const float *brow = (has_bias ? p.bias + (long long)ei.n * p.bias_stride : reinterpret_cast<const float *>(p.out)) + ei.g0 * 32 + fq * 4;
        const bf16_t *resp = has_res ? p.residual : reinterpret_cast<const bf16_t *>(p.out);
        auto epi_off = [&](int tt) -> long long {
            const long long m = (((long long)ei.n * p.Do + (ei.d0 + w)) * p.Ho + (ei.h0 + tt)) * p.Wo + (ei.w0 + fr);
            return m * p.Cout_pad + ei.g0 * 32 + fq * 4;
        };
        auto epi_loads = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k == 0) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bv[ct] = *reinterpret_cast<const f32x4 *>(brow + ct * 16);
            }
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const long long o = epi_off(2 * k + t2);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) rv[k & 1][t2][ct] = *reinterpret_cast<const bf16x4 *>(resp + o + ct * 16);
            }
        };
        auto epi_part = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k == 0) {                         // (zeroed here, not at the top of the phase: 32 registers the in-place passes can use)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ssum[a][j] = 0.f; ssq[a][j] = 0.f; }
            }
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int tt = 2 * k + t2;
                const long long o = epi_off(tt);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int co = ei.g0 * 32 + ct * 16 + fq * 4;
                    f32x4 vv = acc[tt][ct];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vv[j] += has_bias ? bv[ct][j] : 0.f;
                        vv[j] += has_res ? (float)rv[k & 1][t2][ct][j] : 0.f;
                        if (co + j >= p.Cout) vv[j] = 0.f;
                    }
                    bf16x4 ob;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ob[j] = (bf16_t)vv[j];
                        const float f = (float)ob[j];                   // what the next norm will read
                        ssum[ct][j] += f;
                        ssq[ct][j] += f * f;
                    }
                    *reinterpret_cast<bf16x4 *>((bf16_t *)p.out + o + ct * 16) = ob;
                }
            }
        };
        // the ten intervals, each with a compile-time index (a `#pragma unroll` loop was NOT unrolled by hipcc here: the accumulator array
        // was then indexed at run time and moved to scratch)
        auto interval = [&](auto ivc) {
            constexpr int iv = decltype(ivc)::value;
            using std::integral_constant;
            if constexpr (iv == 0) {
                if (stage && PRO != 0 && w == 0) {          // scale / shift rows of this chunk by ONE 16-lane DMA of wave 0
                    if (lane < 16) {
                        const float *sp = (lane < 8 ? p.gn_scale : p.gn_shift) + (long long)it.n * (p.C1 + p.C2) + jc * 32 + (lane & 7) * 4;
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(sp), "s"((int)lds0 + team * H3_XB + H3_NROWS * 64) : "memory", "m0");
                    }
                }
            }
            if constexpr (iv <= 5) {                        // three DMA pieces per interval: the address unit takes ~90 cycles per piece and CU
                if (stage) {
                    dma(integral_constant<int, 3 * iv>{});
                    dma(integral_constant<int, 3 * iv + 1>{});
                    if constexpr (iv < 5) dma(integral_constant<int, 3 * iv + 2>{});
                }
            }
            if constexpr (iv == 5) {
                if (do_epi) epi_loads(integral_constant<int, 0>{});
            }
            if constexpr (iv >= 6) {
                if (do_epi) {
                    if constexpr (iv < 9) epi_loads(integral_constant<int, iv - 5>{});
                    epi_part(integral_constant<int, iv - 6>{});
                    if (iv == 9 && stats) {
                        // per-channel sums of this WAVE's 128 positions (16 positions of a lane row by DPP moves), one 64-bit fixed-point integer atomic
                        // per channel and quantity: integer adds commute, so the totals are exact sums of the waves' fp32 partials in any order
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float a = gg_row16_sum(ssum[ct][j]), b = gg_row16_sum(ssq[ct][j]);
                                if (fr < 2) {
                                    const long long fx = __double2ll_rn((double)(fr ? b : a) * (double)(fr ? GG_ACC_SQ_SCALE : GG_ACC_SUM_SCALE));
                                    const int stripe = (ei.idx * 8 + wave) % GG_ACC_STRIPES_HALO;
                                    atomicAdd(reinterpret_cast<unsigned long long *>(p.gn_acc + ((((long long)ei.n * GG_ACC_STRIPES_HALO + stripe) * p.Cout_pad + ei.g0 * 32 + ct * 16 + fq * 4 + j) * 2 + fr)),
                                              (unsigned long long)fx);
                                }
                            }
                    }
                }
            }
            H3_STAMP(2 * iv);
            if constexpr (iv < 9) h3_bar();
            else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every DMA piece of this wave has landed: the P-phase behind the barrier reads them
                __syncthreads();
            }
            H3_STAMP(2 * iv + 1);
        };
        interval(std::integral_constant<int, 0>{}); interval(std::integral_constant<int, 1>{}); interval(std::integral_constant<int, 2>{});
        interval(std::integral_constant<int, 3>{}); interval(std::integral_constant<int, 4>{}); interval(std::integral_constant<int, 5>{});
        interval(std::integral_constant<int, 6>{}); interval(std::integral_constant<int, 7>{}); interval(std::integral_constant<int, 8>{});
        interval(std::integral_constant<int, 9>{});
        if (GG_H3_SPRIO) __builtin_amdgcn_s_setprio(0);
        if (do_epi) {                                       // the next item starts from zero
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2) acc[a][b2] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        H3_NEXT_PHASE();
    };

    // Global phase g: team A runs its own phase g, team B its phase g - 1 (S, T, S, T, ...); after every global phase both teams run the
    // P-phase over the box staged in it: in an even phase that is team A's (its job g / 2), in an odd one team B's (job (g - 1) / 2).
    // (A single loop over g with the phase kinds as branches makes hipcc shuffle the 128 accumulator registers between the arms:
    // 300 spills; this straight S, P, T, P order per team has none.)
    if (team == 1) { dummy_phase(); H3_NEXT_PHASE(); pphase(0, 0, 0, J > 0); }

    for (int q = 0; q <= J; ++q) {
        if (q == 0 && team == 0) {                          // weights of the first two lines of the first tap phase
            const Item it0 = decode(item_of(0));
            const char *wsrc = (const char *)p.weight + wbase_off(it0.g0, 0) + lw0 + (tid & 63) * 16;
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(wsrc + (((3 * l + u) * nchunk) << 11)),
                                 "s"((int)lds0 + H3_WOFF + l * H3_WSLOT + u * 4096 + w * 1024) : "memory", "m0");
        }
#ifdef GG_H3_ABL_NOSTAGE                 /* timing ablation only (wrong results): the staging team only keeps the barriers */
        sphase(false, false);
        pphase(team, jk, jc, false);
#else
        sphase(jc == 0 && jk > 0, jk < K);
        // the P-phase behind this team's S-phase: its own box.  (Team B's S-phase q runs beside team A's T-phase q: same job.)
        pphase(team, jk < K ? jk : 0, jc, jk < K);
#endif
        if (q == J) break;
        {
            // ======================================= T-phase: the 27 taps of job (jk, jc) =======================================
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int lane2 = t2 & 63, fr2 = lane2 & 15, fq2 = lane2 >> 4;
            int ax[3];                                      // operand lane bases (LDS byte addresses)
#pragma unroll
            for (int k = 0; k < 3; ++k) ax[k] = (int)lds0 + team * H3_XB + w * (H3_HH * H3_HW * 64) + (fr2 + k) * 64 + ((fq2 ^ fsw(fr2 + k)) * 16);
            const int aw = (int)lds0 + H3_WOFF + fr2 * 64 + ((fq2 ^ ((fr2 >> 1) & 2)) * 16);
            const int lw = lw0 + lane2 * 16;
            const Item it = decode(item_of(jk));
            int nk = jk, nc = jc;                           // whose weights the last two DMA rounds fetch: the NEXT tap phase's job
            if (team == 1) { nc = jc + 1; if (nc == nchunk) { nc = 0; nk = jk + 1; } }
            const int base_c = wbase_off(it.g0, jc);
            int base_n = base_c;
            if (nk < K) base_n = wbase_off(nk == jk ? it.g0 : decode(item_of(nk)).g0, nc);
            const int dn = base_n - base_c - 9 * ws;
            int vo[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) vo[u] = base_c + lw + (((6 + u) * nchunk) << 11);      // line 2, tap u
            const void *wbase = (const void *)p.weight;
            u32x4 xa[8], wa[4], xb[8], wb[4];
            H3_STAMP(20);
            // team A starts every tap phase on an even global line (slot 0, register set a), team B on an odd one: one scalar branch
#ifdef GG_H3_ABL_MFMA32               /* timing ablation only (wrong results): 32x32x16 MFMAs on the same fragments */
            typedef float f32x16 __attribute__((ext_vector_type(16)));
            f32x16 acc16[8];
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc16[a][b2 * 4 + e] = acc[a][b2][e];
            asm volatile("s_cmp_eq_u32 %[team], 0\n"
                         "s_cbranch_scc1 LH3P0_%=\n" GG_H3_TPHASE32_P1 "s_branch LH3END_%=\n"
                         "LH3P0_%=:\n" GG_H3_TPHASE32_P0 "LH3END_%=:\n" GG_H3_ASM_OPERANDS32);
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[a][b2][e] = acc16[a][b2 * 4 + e];
#else
            asm volatile("s_cmp_eq_u32 %[team], 0\n"
                         "s_cbranch_scc1 LH3P0_%=\n" GG_H3_TPHASE_P1 "s_branch LH3END_%=\n"
                         "LH3P0_%=:\n" GG_H3_TPHASE_P0 "LH3END_%=:\n" GG_H3_ASM_OPERANDS);
#endif
            // Tell the COMPILER that no vector-memory operation is outstanding (true: the assembly ends in s_waitcnt vmcnt(0)).  Without
            // it its model still holds the epilogue's last stores / atomics from the previous S-phase as pending, and the first LDS read of
            // the next S-phase that reuses one of their data registers gets a compiler-inserted s_waitcnt vmcnt(0), which drains the DMAs.
            __builtin_amdgcn_s_waitcnt(0x0F70);
            H3_STAMP(21);
            __syncthreads();
            H3_STAMP(22);
            H3_NEXT_PHASE();
            {
                int ok2 = jk, oc2 = jc;                    // the job the OTHER team has staged beside these taps: the same one (team B lags) or the next (team A leads)
                if (team == 1) { if (++oc2 == nchunk) { oc2 = 0; ++ok2; } }
#ifdef GG_H3_ABL_NOSTAGE
                pphase(team ^ 1, ok2, oc2, false);
#else
                pphase(team ^ 1, ok2 < K ? ok2 : 0, oc2, ok2 < K);
#endif
            }
            if (++jc == nchunk) { jc = 0; ++jk; }
        }
    }
    if (team == 0) { dummy_phase(); pphase(1, 0, 0, false); }
}

// Returns GG_ERR_UNSUPPORTED (silently) outside the envelope; stream == (hipStream_t)-1: dry run.
int gg_conv_halo3_try(const ConvParams &p, hipStream_t stream)
{
    if (!(p.kd == 3 && p.kh == 3 && p.kw == 3) || p.stride != 1 || p.pad != 1 || p.upsample) return GG_ERR_UNSUPPORTED;
    if (p.out_dtype != GG_BF16 || (p.Cout_pad % 64) || (p.Do % 8) || (p.Ho % 8) || (p.Wo % 16)) return GG_ERR_UNSUPPORTED;
    if (p.ddim_x || p.epi_geglu || p.skip_src1 || p.pro_acc1) return GG_ERR_UNSUPPORTED;
    const int tiles_d = p.Do / 8, tiles_h = p.Ho / 8, tiles_w = p.Wo / 16, ncg = p.Cout_pad / 64;
    const long long items = (long long)p.N * tiles_d * tiles_h * tiles_w * ncg;
    if ((long long)p.Cout_pad / 32 * 27 * p.nchunk * 2048 >= (1LL << 31)) return GG_ERR_UNSUPPORTED;        // 32-bit DMA offsets
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return GG_ERR_HIP;
    // persistent: one workgroup per CU; under-filled grids (fewer items than ~3/4 of the CUs) stay on the 256-position boxes of conv_halo_kernel
    if (p.path_hint != 7 && items < (long long)cus * 3 / 4) return GG_ERR_UNSUPPORTED;
    if (stream == (hipStream_t)-1) return GG_OK;
    static std::atomic<unsigned long long> attr_mask{0};
    const unsigned long long dev_bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & dev_bit)) {
        if (hipFuncSetAttribute((const void *)conv_halo3_team_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_halo3_team_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_halo3_team_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS) != hipSuccess)
            return GG_ERR_UNSUPPORTED;
        attr_mask.fetch_or(dev_bit, std::memory_order_release);
    }
    const int grid = (int)(items < cus ? items : cus);
    if (p.prologue_act == 1)
        hipLaunchKernelGGL(conv_halo3_team_kernel<1>, dim3((unsigned)grid), dim3(512), H3_LDS, stream, p, tiles_d, tiles_h, tiles_w, ncg, (int)items);
    else if (p.prologue_act == 2)
        hipLaunchKernelGGL(conv_halo3_team_kernel<2>, dim3((unsigned)grid), dim3(512), H3_LDS, stream, p, tiles_d, tiles_h, tiles_w, ncg, (int)items);
    else
        hipLaunchKernelGGL(conv_halo3_team_kernel<0>, dim3((unsigned)grid), dim3(512), H3_LDS, stream, p, tiles_d, tiles_h, tiles_w, ncg, (int)items);
    GG_CHECK_LAUNCH();
    return GG_OK;
}
