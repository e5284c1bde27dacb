// Pipelined halo-tile implicit-GEMM convolution for gfx950, 3-D: 3x3x3, stride 1, pad 1 (optionally with the nearest x2 upsample
// fused in front), bf16 in / fp32 accumulate on v_mfma_f32_16x16x32_bf16.  It replaces conv_halo_kernel<1,NT,UP> (gg_conv_halo.hip)
// on grids that fill the chip; the differences are all about OVERLAP (the MFMA pipe of the old kernel was 42 % busy):
//
//   * ONE persistent workgroup of 8 waves per CU walks a contiguous run of work items, item = (output box of 4x8x16 = 512
//     positions x 32*NT output channels, 32-channel input chunk).  All 160 KiB of LDS belong to it:
//       2 x 68 KiB input-box buffers  +  2 x (3 taps x NT x 2 KiB) weight buffers.
//   * The input box of item k+1 is fetched by LDS-DMA (global_load_lds: no VGPRs, everything in flight at once) into the idle box
//     buffer while the 27 taps of item k run, and GroupNorm*SiLU (+ the zero padding) is applied to it IN PLACE, a few 16-byte
//     pieces per thread between the tap groups of item k.  So staging costs issue slots, never a round trip, also across the
//     boundary between two output boxes (the old kernel exposed one staging round trip per workgroup and 6 per chunk).
//   * Weights arrive by LDS-DMA one (kd, kh) group of 3 taps ahead: 9 workgroup barriers per chunk instead of 27.
//   * Each wave owns 64 positions x 64 channels (4 x 4 MFMA tiles, 64 accumulator VGPRs): the operand fragments of a whole tap
//     group fit in registers next to the accumulators, so the compiler can run the LDS reads ahead of the MFMAs.
//   * The weight rows of a 32-channel group are permuted while they are DMA-ed (free: the per-lane source address), so that a
//     lane's two accumulators of a group hold 8 CONSECUTIVE output channels: the epilogue stores 16 bytes per lane.
#include "gg_conv.h"
#include <type_traits>

#define GG_PIPE_WAITCNT_ALL0 0x0070                                                  /* vmcnt 0, lgkmcnt 0 */
#define GG_PIPE_WAITCNT_VM_LGKM0(VM) ((((VM) & 15) | (((VM) >> 4) << 14)) | 0x0070)     /* vmcnt VM, lgkmcnt 0 */

struct PipeUnit {
    int n, d0, h0, w0, cg;          // sample, output-box origin, 32*NT-channel output group
    int id0, ih0, iw0;              // input coordinates of box row (0,0,0)
};

// ABL: timing ablations for tools/probe_conv3d.py (results are garbage): 1 no box DMA / in-place pass, 2 no weight DMA, 4 no barriers,
// 8 no LDS fragment reads
template <int NT, int UP, int PRO, int ABL = 0>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_halo3d_pipe_kernel(const ConvParams p, const int tiles_d, const int tiles_h, const int tiles_w,
                                                                   const int ncg, const int nunits)
{
    constexpr int TD = 4, TH = 8, TW = 16;
    constexpr int NTHR = 512;
    constexpr int HD = UP ? TD / 2 + 2 : TD + 2;
    constexpr int HH = UP ? TH / 2 + 2 : TH + 2;
    constexpr int HW = UP ? TW / 2 + 2 : TW + 2;
    constexpr int NROWS = HD * HH * HW;                       // 1080 (240 with the fused upsample)
    constexpr int NPIECE = NROWS * 4;
    constexpr int XBUF = ((NROWS * 64 + 512 + 1023) / 1024) * 1024;   // box rows + >= 512 B of slack (dummy slots of the in-place pass)
    constexpr int NDMA = XBUF / 1024;                         // LDS-DMA instructions per box (16 rows each)
    constexpr int JT = (NPIECE + NTHR - 1) / NTHR;            // in-place transform pieces per thread and item
    constexpr int WTAP = NT * 2048, WGRP = 3 * WTAP;
    constexpr int NWP = 6 * NT;                               // 1 KiB weight pieces per tap group
    constexpr int NGRP = 9;                                   // (kd, kh) tap groups per chunk
    __shared__ __attribute__((aligned(1024))) char smem[2 * XBUF + 2 * WGRP];
    char *const xbase = smem;
    char *const wbase = smem + 2 * XBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // ---- this block's run of units; blocks b and b + 8 share an XCD, so every XCD gets a contiguous region (halo reuse in its L2)
    const int nb = gridDim.x;
    int b = blockIdx.x;
    if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);
    const int u0 = (int)(((long long)nunits * b) / nb), u1 = (int)(((long long)nunits * (b + 1)) / nb);
    const int nitems = (u1 - u0) * p.nchunk;
    if (nitems <= 0) return;

    auto decode = [&](int u) -> PipeUnit {
        PipeUnit q;
        q.cg = u % ncg;
        int t = u / ncg;
        const int tw = t % tiles_w; t /= tiles_w;
        const int th = t % tiles_h; t /= tiles_h;
        const int td = t % tiles_d;
        q.n = t / tiles_d;
        q.d0 = td * TD; q.h0 = th * TH; q.w0 = tw * TW;
        q.id0 = UP ? q.d0 / 2 - 1 : q.d0 - 1;
        q.ih0 = UP ? q.h0 / 2 - 1 : q.h0 - 1;
        q.iw0 = UP ? q.w0 / 2 - 1 : q.w0 - 1;
        return q;
    };

    constexpr unsigned FMASK = UP ? 0x3C0u : 0xFC30u;         // box swizzle f(hw) (see gg_conv_halo.hip): 2 for hw in {6..9} / {4,5,10..15}
    auto fsw = [&](int hw) -> int { return (int)((FMASK >> hw) & 1u) << 1; };

    // ---- staging geometry without integer division in the loop: this lane's first box row of the DMA stream (waves 4..7: DMA
    //      instruction i = wave-4 + 4k moves box rows 16i .. 16i+15, this lane 16 bytes of row 16i + lane/4; rows advance by 64 per k)
    //      and of the in-place pass (piece tid + 512 j: rows advance by 128 per j) are decoded once; stepping is carry arithmetic
    auto advance = [&](int step, int &hd, int &hh, int &hw) {          // (hd, hh, hw) += step rows (compile-time step)
        const int dhd = step / (HH * HW), rem = step % (HH * HW);
        hw += rem % HW; hh += rem / HW; hd += dhd;
        if (hw >= HW) { hw -= HW; hh += 1; }
        if (hh >= HH) { hh -= HH; hd += 1; }
    };
    constexpr int NBK = (NDMA + 7) / 8;                       // box DMA instructions per wave (instruction i = wave + 8k)
    constexpr int NBK0 = (NBK + 1) / 2;                       // issued during tap group 0, the rest during group 1
    const int xq = tid & 3;
    int bx_hd, bx_hh, bx_hw, tr_hd, tr_hh, tr_hw;
    {
        const int r0 = wave * 16 + (lane >> 2);
        bx_hd = r0 / (HH * HW); bx_hh = (r0 - bx_hd * (HH * HW)) / HW; bx_hw = r0 - bx_hd * (HH * HW) - bx_hh * HW;
        const int r1 = tid >> 2;
        tr_hd = r1 / (HH * HW); tr_hh = (r1 - tr_hd * (HH * HW)) / HW; tr_hw = r1 - tr_hd * (HH * HW) - tr_hh * HW;
    }
    // per-UNIT staging metadata (the box geometry is the same for every chunk of a unit):
    //   tr_oob / tr_swz : bit j = piece j of this thread lies on a padding row / sits in slot xq ^ 2
    unsigned tr_oob = 0, tr_swz = 0;
    auto unit_meta = [&](const PipeUnit &q) {
        {
            int hd = tr_hd, hh = tr_hh, hw = tr_hw;
            tr_oob = 0; tr_swz = 0;
#pragma unroll 1                       /* rolled on purpose: unrolled, the (unit-independent) row sequence is hoisted and pinned in ~27 VGPRs */
            for (int j = 0; j < JT; ++j) {
                const bool inb = (unsigned)(q.id0 + hd) < (unsigned)p.D && (unsigned)(q.ih0 + hh) < (unsigned)p.H && (unsigned)(q.iw0 + hw) < (unsigned)p.W;
                tr_oob |= (inb ? 0u : 1u) << j;
                tr_swz |= (unsigned)(fsw(hw) >> 1) << j;
                advance(128, hd, hh, hw);
            }
        }
    };

    // ---- box DMA of one chunk, instructions k0 <= k < k1 of this wave (instruction i = wave + 8k moves box rows 16i .. 16i+15; this
    //      lane 16 bytes of row 16i + lane/4).  NO lane is masked off: a padding row fetches a valid dummy address (the in-place
    //      pass overwrites padding rows with zeros after the box has landed), so every wave issues exactly k1 - k0 instructions
    //      and the counted vmcnt waits of waves 0..3 (weights first, box behind them) stay exact.
    auto issue_box = [&](const PipeUnit &q, int chunk, int buf, int k0, int k1) {
        const bool second = chunk >= p.nchunk1;
        const int Cs = second ? p.C2 : p.C1;
        const bf16_t *src = (second ? p.src2 : p.src1) + (long long)q.n * p.D * p.H * p.W * Cs + (second ? chunk - p.nchunk1 : chunk) * 32;
        int hd = bx_hd, hh = bx_hh, hw = bx_hw;
        for (int k = 0; k < k0; ++k) advance(128, hd, hh, hw);
#pragma unroll 1
        for (int k = k0; k < k1; ++k) {
            const int i = wave + 8 * k;
            const int id = q.id0 + hd, ih = q.ih0 + hh, iw = q.iw0 + hw;
            const bool ok = hd < HD && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const unsigned pos = ok ? (unsigned)((id * p.H + ih) * p.W + iw) : 0u;
            const unsigned off = pos * (unsigned)Cs + ((unsigned)((lane & 3) ^ fsw(hw)) << 3);
            if (i < NDMA)                                                  // wave-uniform (only the last k of waves 4..7 can be short)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(xbase + buf * XBUF + i * 1024), 16, 0, 0);
            advance(128, hd, hh, hw);
        }
    };

    // ---- weight DMA (waves 0..3): tap group g (taps 3g .. 3g+2) of (cout group, chunk) -> weight buffer `buf`.
    // LDS row (ct*16 + i) of a 32-channel group receives packed row (i>>2)*8 + (ct&1)*4 + (i&3): see the header comment.
    auto issue_w = [&](int cg, int chunk, int g, int buf) {
        const int i16 = lane >> 2, sl = lane & 3;
#pragma unroll
        for (int k = 0; k < (NWP + 3) / 4; ++k) {
            const int piece = wave + 4 * k;                               // 1 KiB piece: (kw, 32-channel group m, ct parity)
            if (piece < NWP) {
                const int kw = piece / (2 * NT), rest = piece - kw * (2 * NT);
                const int m = rest >> 1, ctl = rest & 1;
                const int kk = sl ^ ((i16 >> 1) & 2);                     // logical k-chunk this LDS slot must hold
                const int rg = ((i16 >> 2) << 3) + ctl * 4 + (i16 & 3);   // packed (global) row
                const int sp = kk ^ (ctl << 1);                           // its physical slot in the packed image
                const bf16_t *srcw = p.weight + ((((long long)(cg * NT + m) * 27 + (g * 3 + kw)) * p.nchunk + chunk) << 10) + rg * 32 + sp * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)srcw,
                                                 (__attribute__((address_space(3))) void *)(wbase + buf * WGRP + kw * WTAP + m * 2048 + ctl * 1024), 16, 0, 0);
            }
        }
    };

    // ---- GroupNorm scale / shift of this thread's channel piece (tid & 3) for one chunk
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
    auto load_gn = [&](const PipeUnit &q, int chunk) {
        if (PRO != 0) {
            const long long so = (long long)q.n * (p.C1 + p.C2) + chunk * 32 + xq * 8;
            s0 = *reinterpret_cast<const f32x4 *>(p.gn_scale + so); s1 = *reinterpret_cast<const f32x4 *>(p.gn_scale + so + 4);
            b0 = *reinterpret_cast<const f32x4 *>(p.gn_shift + so); b1 = *reinterpret_cast<const f32x4 *>(p.gn_shift + so + 4);
        }
    };

    // ---- in-place pass over the next item's box, NPG pieces (16 bytes each) per thread and tap group, groups 4..8: padding rows
    //      <- 0, data rows <- act(x * s + b).  Branch-free, in three steps (LDS reads / arithmetic / LDS writes) that sit in the same
    //      basic block as the group's MFMAs, so that the scheduler can lay the VALU work into the MFMA shadow (sched_group_barrier
    //      pipeline below).  A piece that does not exist (the last j is ragged) works on a 16-byte dummy slot in the slack behind
    //      the box rows instead of being branched around.
    constexpr int NPG = (JT + 4) / 5;
    constexpr int TR_VALU = PRO == 1 ? 60 : (PRO == 2 ? 24 : 8);      // rough VALU instructions per piece (for the pipeline spec)
    auto tr_addr = [&](int buf, int j) -> char * {
        const bool live = j < JT && tid + NTHR * j < NPIECE;
        const int real = ((tid >> 2) + 128 * j) * 64 + ((xq ^ (((tr_swz >> j) & 1u) << 1)) << 4);
        const int dummy = NROWS * 64 + (tid & 31) * 16;
        return xbase + buf * XBUF + (live ? real : dummy);
    };
    // one dword (channels 2e, 2e + 1 of the piece) of the in-place pass
    auto tr_dword = [&](unsigned in, int e, bool oob) -> unsigned {
        unsigned o = in;
        if (PRO != 0) {
            const float sa = e < 2 ? s0[2 * e] : s1[2 * e - 4], sb = e < 2 ? s0[2 * e + 1] : s1[2 * e - 3];
            const float ba = e < 2 ? b0[2 * e] : b1[2 * e - 4], bb = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
            float ya = __builtin_bit_cast(float, in << 16) * sa + ba;
            float yb = __builtin_bit_cast(float, in & 0xffff0000u) * sb + bb;
            if (PRO == 1) {
                ya = ya * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ya * -1.44269504088896340736f));
                yb = yb * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(yb * -1.44269504088896340736f));
            }
            bf16x2 pk;
            pk[0] = (bf16_t)ya;
            pk[1] = (bf16_t)yb;
            o = __builtin_bit_cast(unsigned, pk);
        }
        return oob ? 0u : o;
    };
    auto tr_math = [&](const bf16x8 xb, bool oob) -> u32x4 {
        const u32x4 in = __builtin_bit_cast(u32x4, xb);
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = tr_dword(in[e], e, oob);
        return out;
    };
    // the un-overlapped form (first item of a block)
    auto transform_all = [&](int buf) {
#pragma unroll 1
        for (int j = 0; j < JT; ++j) {
            char *pc = tr_addr(buf, j);
            const bf16x8 xb = *reinterpret_cast<const bf16x8 *>(pc);
            *reinterpret_cast<u32x4 *>(pc) = tr_math(xb, (tr_oob >> j) & 1u);
        }
    };

    // ---- per-lane operand offsets
    int lane_off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int rw = UP ? ((fr + k + 1) >> 1) : (fr + k);
        lane_off[k] = rw * 64 + ((fq ^ fsw(rw)) << 4);
    }
    const int od = wave >> 1, ohb = (wave & 1) * 4;              // this wave's 4 W-lines: (od, ohb .. ohb+3)
    int wrow_off[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = (ct & 1) * 16 + fr;                         // row inside a 32-channel group image (swizzle bits are those of fr)
        wrow_off[ct] = (ct >> 1) * 2048 + r * 64 + (swz64(r, fq) << 4);
    }

    f32x4 acc[4][2 * NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 2 * NT; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ================= prologue: item 0 is staged without overlap =================
    PipeUnit cur = decode(u0);
    unit_meta(cur);
    load_gn(cur, 0);
    if (wave < 4) issue_w(cur.cg, 0, 0, 0);
    issue_box(cur, 0, 0, 0, NBK);
    __builtin_amdgcn_s_waitcnt(GG_PIPE_WAITCNT_ALL0);
    __builtin_amdgcn_s_barrier();
    transform_all(0);
    __builtin_amdgcn_s_waitcnt(GG_PIPE_WAITCNT_ALL0);
    __builtin_amdgcn_s_barrier();

    int chunk = 0, unit = u0;
    int G = 0;                                                   // running tap-group counter (weight buffer = G & 1)
#pragma unroll 1
    for (int it = 0; it < nitems; ++it) {
        const int xb = it & 1;
        const bool has_next = it + 1 < nitems;
        const int nchunk_next = (chunk + 1 == p.nchunk) ? 0 : chunk + 1;
        const int nunit = (chunk + 1 == p.nchunk) ? unit + 1 : unit;
        PipeUnit nxt = cur;
        if (has_next && nunit != unit) {
            nxt = decode(nunit);
            unit_meta(nxt);                                    // the box of `cur` was fully staged during the previous item
        }
        const char *const xs = xbase + xb * XBUF;

#pragma unroll 1
        for (int g = 0; g < NGRP; ++g, ++G) {
            auto stamp = [&](int k) {
                if ((ABL & 16) && blockIdx.x == 0 && it == 2 && lane == 0) ((long long *)p.ws)[(wave * NGRP + g) * 8 + k] = (long long)__builtin_amdgcn_s_memtime();
            };
            stamp(0);
            // ---- DMA issue for what comes next: weights of the next tap group, the box of the next item (once, at g == 0)
            if (g == 0 && has_next) load_gn(nxt, nchunk_next);     // oldest: the counted waits below never have to cover them
            if (wave < 4 && !(ABL & 2)) {
                if (g + 1 < NGRP) issue_w(cur.cg, chunk, g + 1, (G + 1) & 1);
                else if (has_next) issue_w(nxt.cg, nchunk_next, 0, (G + 1) & 1);
            }
            // box DMA of the next item: waves 0..3 issue their share BEFORE the group's MFMAs, waves 4..7 (their SIMD partners) AFTER
            // theirs, so that one half's address arithmetic / VMEM issue runs beside the other half's matrix work
            auto box_share = [&]() {
                if (g == 0) issue_box(nxt, nchunk_next, xb ^ 1, 0, NBK0);
                else if (g == 1) issue_box(nxt, nchunk_next, xb ^ 1, NBK0, NBK);
            };
            if (has_next && !(ABL & 1) && wave < 4) box_share();
            stamp(1);

            // ---- 3 taps (kd, kh fixed; kw = 0..2) out of LDS, with this thread's NPG pieces of the in-place pass over the NEXT item's
            //      box (groups 4..8; the box landed and became visible at the barrier that ended g == 3) laid into the MFMA shadow
            const bool do_tr = has_next && g >= 4 && !(ABL & 1);
            const int kd = g / 3, kh = g - kd * 3;
            const int hd = UP ? ((od + kd + 1) >> 1) : od + kd;
            const char *const wb = wbase + (G & 1) * WGRP;
            auto group_body = [&](auto tr_tag) {
                constexpr bool TR = decltype(tr_tag)::value;
                // activation fragments are double-buffered across taps; a weight fragment is refilled in place right after the last
                // MFMA of the tap that reads it (an MFMA reads its operands at issue), 12 MFMAs before its next use
                bf16x8 xf[2][4], wf[2 * NT];
                auto load_x = [&](int kw, bf16x8 (&xa)[4]) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        if (ABL & 8) { asm volatile("" : "=v"(xa[tt])); continue; }
                        const int oh = ohb + tt;
                        const int hh = UP ? ((oh + kh + 1) >> 1) : oh + kh;
                        xa[tt] = *reinterpret_cast<const bf16x8 *>(xs + (hd * HH + hh) * (HW * 64) + lane_off[kw]);
                    }
                };
                auto load_w = [&](int kw, int ct) {
                    if (ABL & 8) { asm volatile("" : "=v"(wf[ct])); return; }
                    wf[ct] = *reinterpret_cast<const bf16x8 *>(wb + kw * WTAP + wrow_off[ct]);
                };
                // The in-place pass is laid between the MFMA quads BY HAND (sched_barrier fences; the scheduler's own interleaving
                // via sched_group_barrier blew the register budget): after quad s of the group's NS = 6 NT quads comes step s:
                //   piece 0: DPS dwords per step from step 0, its LDS write + the LDS read of piece 1 at step 4 / DPS,
                //   piece 1 (NPG == 2): from step P1, its LDS write at step P1 + 4 / DPS.
                // The quad's MFMAs are in flight (64 pipe cycles, 32 issue cycles) while the step's VALU issue.
                constexpr int DPS = NT == 2 ? 1 : 2, P0END = 4 / DPS, P1 = NT == 2 ? 6 : 3;
                char *pc0 = nullptr, *pc1 = nullptr;
                u32x4 tin0, tin1, tout0, tout1;
                const int j0 = NPG * (g - 4);
                const bool oob0 = j0 < JT ? ((tr_oob >> j0) & 1u) : 0u, oob1 = j0 + 1 < JT ? ((tr_oob >> (j0 + 1)) & 1u) : 0u;
                load_x(0, xf[0]);
#pragma unroll
                for (int ct = 0; ct < 2 * NT; ++ct) load_w(0, ct);
                if (TR) {
                    pc0 = tr_addr(xb ^ 1, j0);
                    tin0 = *reinterpret_cast<const u32x4 *>(pc0);
                    if (NPG == 2) pc1 = tr_addr(xb ^ 1, j0 + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                    for (int ct = 0; ct < 2 * NT; ++ct) {
                        const int step = kw * 2 * NT + ct;
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt)
                            acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], xf[kw & 1][tt], acc[tt][ct], 0, 0, 0);
                        if (kw + 1 < 3) {
                            if (ct == 0) load_x(kw + 1, xf[(kw + 1) & 1]);
                            load_w(kw + 1, ct);
                        }
                        if (TR) {
                            if (step < P0END) {
#pragma unroll
                                for (int d = 0; d < DPS; ++d) tout0[step * DPS + d] = tr_dword(tin0[step * DPS + d], step * DPS + d, oob0);
                            } else if (step == P0END) {
                                *reinterpret_cast<u32x4 *>(pc0) = tout0;
                                if (NPG == 2) tin1 = *reinterpret_cast<const u32x4 *>(pc1);
                            } else if (NPG == 2 && step >= P1 && step < P1 + P0END) {
#pragma unroll
                                for (int d = 0; d < DPS; ++d) tout1[(step - P1) * DPS + d] = tr_dword(tin1[(step - P1) * DPS + d], (step - P1) * DPS + d, oob1);
                            } else if (NPG == 2 && step == P1 + P0END) {
                                *reinterpret_cast<u32x4 *>(pc1) = tout1;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            stamp(2);
            if (do_tr) group_body(std::true_type{});
            else group_body(std::false_type{});
            stamp(3);
            if (has_next && !(ABL & 1) && wave >= 4) box_share();
            stamp(4);

            // ---- waves 0..3: the next group's weights must have landed (they were issued BEFORE this group's box instructions, which
            //      may stay in flight: counted vmcnt); the whole box must be in LDS at the barrier that ends g == 3 (all waves)
            if (g == 3 || (wave < 4 && (g > 1 || !has_next || (ABL & 1)))) __builtin_amdgcn_s_waitcnt(GG_PIPE_WAITCNT_ALL0);
            else if (wave < 4 && g == 0) __builtin_amdgcn_s_waitcnt(GG_PIPE_WAITCNT_VM_LGKM0(NBK0));
            else if (wave < 4) __builtin_amdgcn_s_waitcnt(GG_PIPE_WAITCNT_VM_LGKM0(NBK - NBK0));
            else __builtin_amdgcn_s_waitcnt(0xC07F);               // waves 4..7: lgkmcnt(0) only (their LDS writes / reads are done)
            stamp(5);
            if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            stamp(6);
        }

        // ================= end of a unit: + bias[n] (+ residual) -> bf16 / fp32, 8 consecutive channels per lane =================
        if (chunk + 1 == p.nchunk) {
            const float *brow = p.bias ? p.bias + (long long)cur.n * p.bias_stride : nullptr;
            f32x4 bv[NT][2];
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                const int co = (cur.cg * NT + m) * 32 + fq * 8;
                bv[m][0] = brow ? *reinterpret_cast<const f32x4 *>(brow + co) : f32x4{0.f, 0.f, 0.f, 0.f};
                bv[m][1] = brow ? *reinterpret_cast<const f32x4 *>(brow + co + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // two W-lines at a time: their residual loads are in flight together (the registers for all four would spill)
#pragma unroll
            for (int th = 0; th < 4; th += 2) {
                long long obase[2];
                bf16x8 rv[2][NT];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    obase[t2] = ((((long long)cur.n * p.Do + (cur.d0 + od)) * p.Ho + (cur.h0 + ohb + th + t2)) * p.Wo + (cur.w0 + fr)) * p.Cout_pad
                                + cur.cg * NT * 32 + fq * 8;
                    if (p.residual) {
#pragma unroll
                        for (int m = 0; m < NT; ++m) rv[t2][m] = *reinterpret_cast<const bf16x8 *>(p.residual + obase[t2] + m * 32);
                    }
                }
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int tt = th + t2;
#pragma unroll
                    for (int m = 0; m < NT; ++m) {
                        const int co = (cur.cg * NT + m) * 32 + fq * 8;
                        f32x4 v0 = acc[tt][2 * m] + bv[m][0], v1 = acc[tt][2 * m + 1] + bv[m][1];
                        const long long o = obase[t2] + m * 32;
                        if (p.residual) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { v0[j] += (float)rv[t2][m][j]; v1[j] += (float)rv[t2][m][j + 4]; }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (co + j >= p.Cout) v0[j] = 0.f;
                            if (co + 4 + j >= p.Cout) v1[j] = 0.f;
                        }
                        if (p.out_dtype == GG_F32) {
                            *reinterpret_cast<f32x4 *>((float *)p.out + o) = v0;
                            *reinterpret_cast<f32x4 *>((float *)p.out + o + 4) = v1;
                        } else {
                            bf16x8 ob;
#pragma unroll
                            for (int j = 0; j < 4; ++j) { ob[j] = (bf16_t)v0[j]; ob[j + 4] = (bf16_t)v1[j]; }
                            *reinterpret_cast<bf16x8 *>((bf16_t *)p.out + o) = ob;
                        }
                        acc[tt][2 * m] = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[tt][2 * m + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        chunk = nchunk_next;
        unit = nunit;
        cur = nxt;
    }
}

template <int NT, int UP, int PRO, int ABL = 0>
static int launch_pipe(const ConvParams &p, hipStream_t stream)
{
    const int tiles_d = p.Do / 4, tiles_h = p.Ho / 8, tiles_w = p.Wo / 16;
    const int ncg = p.Cout_pad / (32 * NT);
    const long long nunits = (long long)p.N * tiles_d * tiles_h * tiles_w * ncg;
    static const int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
        return 256;
    }();
    const unsigned grid = (unsigned)(nunits < ncu ? nunits : ncu);          // one persistent workgroup per CU (all of its LDS)
    hipLaunchKernelGGL((conv_halo3d_pipe_kernel<NT, UP, PRO, ABL>), dim3(grid), dim3(512), 0, stream, p, tiles_d, tiles_h, tiles_w, ncg, (int)nunits);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

template <int NT, int UP>
static int dispatch_pro_nt(const ConvParams &p, hipStream_t stream)
{
    switch (p.prologue_act) {
        case 1: return launch_pipe<NT, UP, 1>(p, stream);
        case 2: return launch_pipe<NT, UP, 2>(p, stream);
        default: return launch_pipe<NT, UP, 0>(p, stream);
    }
}

static int dispatch_pro(const ConvParams &p, int NT, hipStream_t stream)
{
    if (NT == 2) return p.upsample ? dispatch_pro_nt<2, 1>(p, stream) : dispatch_pro_nt<2, 0>(p, stream);
    return p.upsample ? dispatch_pro_nt<1, 1>(p, stream) : dispatch_pro_nt<1, 0>(p, stream);
}

// 3-D halo shapes only.  Production gate: at least one unit per CU (smaller grids stay on conv_halo_kernel, whose 128..255
// workgroups of 256 threads spread better); path_hint 3 (tests) lifts the gate.  stream == (hipStream_t)-1: dry run.
int gg_conv_halo_pipe_try(const ConvParams &p, hipStream_t stream)
{
    if (!(p.kd == 3 && p.kh == 3 && p.kw == 3) || p.stride != 1 || p.pad != 1) return GG_ERR_UNSUPPORTED;
    if (p.Wo % 16 || p.Ho % 8 || p.Do % 4) return GG_ERR_UNSUPPORTED;
    const int G = p.Cout_pad / 32;
    const int NT = (G % 2 == 0) ? 2 : 1;
    const long long nunits = (long long)p.N * (p.Do / 4) * (p.Ho / 8) * (p.Wo / 16) * (G / NT);
    if (p.path_hint != 3 && nunits < 256) return GG_ERR_UNSUPPORTED;
    if (nunits >= (1LL << 30)) return GG_ERR_UNSUPPORTED;
    if (stream == (hipStream_t)-1) return GG_OK;
#ifdef GG_PIPE_ABLATIONS
    if (NT == 2 && !p.upsample && p.path_hint >= 16 && p.prologue_act == 1) {
        switch (p.path_hint - 16) {
            case 1: return launch_pipe<2, 0, 1, 1>(p, stream);
            case 2: return launch_pipe<2, 0, 1, 2>(p, stream);
            case 3: return launch_pipe<2, 0, 1, 3>(p, stream);
            case 7: return launch_pipe<2, 0, 1, 7>(p, stream);
            case 11: return launch_pipe<2, 0, 1, 11>(p, stream);
            case 15: return launch_pipe<2, 0, 1, 15>(p, stream);
            case 16: return launch_pipe<2, 0, 1, 16>(p, stream);
            default: break;
        }
    }
#endif
    return dispatch_pro(p, NT, stream);
}
