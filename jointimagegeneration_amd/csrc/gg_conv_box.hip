// Box-resident 2-D convolution for UNDER-FILLED grids (latent UNet levels at batch 1: 64x64 .. 16x16, 160..1280 channels):
// 3x3 (stride 1, pad 1, optional fused nearest x2 upsample) and 1x1, bf16 in / fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// These layers are 1-8 GFLOP each: neither MFMA nor HBM bound, but bound by (1) what ONE CU can take in (tools/experiments/
// ubench_intake.hip: 48 GB/s per CU for 64-byte rows, 70 for 128-byte lines, ~80 contiguous, L2-resident, 240 workgroups) and (2) the
// number of instructions a wave has to issue, one by one, in front of and between its memory instructions (~2 ns each at one or two
// waves per SIMD: phase stamps, -DGG_BOX_STAMPS).  The kernel therefore moves the minimum number of bytes into each CU, has every
// byte of the box in flight at once, has no barrier in its main loop, and keeps its scalar bookkeeping out of the inner loops:
//   * a workgroup (8 waves) owns MT position tiles (16 positions each: 1x16, 2x8 or 4x4, so 8x8 and 4x4 levels fit too) x 16*CT
//     output channels; the plan (plan_box) picks MT and CT so that max-over-CUs of (weight slice + input box) bytes is smallest for
//     a single round of <= 256 workgroups; block ids decode by multiply-high with host-provided magics, and the arguments the
//     first DMAs need are pinned into one scalar-load batch (gg_pin);
//   * the input box the 9 taps touch ((TH+2) x 18 rows; upsample: (TH/2+2) x 10) is staged into LDS ONCE for ALL input channels
//     of a stage (<= 128 KiB; two-source concat and zero padding applied here) as one swizzled 64-byte-row plane per 32-channel
//     chunk, by global_load_lds (1 KiB = 16 rows per wave instruction, no VGPRs).  Units (16-row block, chunk) are walked
//     block-major in runs: consecutive chunks are +64 B in memory (the two halves of a 128-byte line go out back to back) and
//     +PLANE in LDS, 6.5 instructions per unit; padding slots are zeroed by the issuing wave after its DMAs have landed.
//     GroupNorm affine (* SiLU), where fused (<= 2 cout tiles per box), then runs IN PLACE in LDS, once per staged element;
//   * the k-steps of the stage are split evenly over the 8 waves as (kh, chunk) units with the three kw taps unrolled (kw is a
//     compile-time constant: operand addresses are lane constant + uniform + immediate, the bookkeeping is paid once per three
//     k-steps).  Each wave streams the weight tiles of ITS units straight from L2 into VGPRs (ring of 2 units, counted vmcnt; no
//     LDS copy, no redundancy between waves; deeper rings are slower: the stream is intake-bound) and reads the activation operand
//     from the LDS box at a shifted row, all MT reads of a k-step ahead of its MFMAs;
//   * the partial accumulators of the 8 waves are combined through LDS in a fixed order (deterministic), then bias / residual /
//     store (+ GroupNorm sums of the next norm, + the fused DDIM update on the UNet head);
//   * workgroups are renumbered so that the ones sharing a weight slice (weight-heavy layers) or an input box
//     (activation-heavy layers) run on the same XCD and hit its L2 (blockIdx round-robins over the 8 XCDs).
#include <atomic>
#include "gg_conv.h"
#ifndef GG_BOX_K1_MAX_M_ACC
#define GG_BOX_K1_MAX_M_ACC 256
#endif
#ifndef GG_BOX_ACC_SILU_MAX_ELEMS
#define GG_BOX_ACC_SILU_MAX_ELEMS 0          /* elements of a workgroup's box up to which a SiLU norm is folded into the conv (0: never; A/B: tools/experiments) */
#endif
#ifndef GG_BOX_NW
#define GG_BOX_NW 8                          /* waves per workgroup (8: two per SIMD; 4: one per SIMD -- A/B: tools/experiments) */
#endif
#ifndef GG_BOX_STRIDE2
#define GG_BOX_STRIDE2 1                     /* stride-2 3x3 convs (UNet Downsample) on the box kernel (A/B: tools/experiments) */
#endif
#ifndef GG_BOX_COUT_SUBSPLIT
#define GG_BOX_COUT_SUBSPLIT 1               /* 3x3 convs of the 8x8 / 4x4 levels: 2 or 4 workgroups per 16-cout tile (A/B: tools/experiments) */
#endif
#include <stdlib.h>

// The compiler hoists loop-invariant address arithmetic of the conditional blocks inside the stage loop (accumulator fold, in-place
// prologue, first-stage operand offsets: ~350 instructions, 0.7 us at batch 1) in front of the loop, i.e. in front of the box DMAs.
// A value laundered through a volatile asm INSIDE a block pins everything computed from it to that block.
__device__ __forceinline__ int gg_here(int v) { asm volatile("" : "+v"(v)); return v; }

// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14), as builtins so that the
// compiler's own wait-count pass sees them
#define GG_WAITCNT_IMM(VM) ((((VM) & 15) | (((VM) >> 4) << 14)) | 0x70)
#define GG_BOX_WAIT_BARRIER(VM) do { __builtin_amdgcn_s_waitcnt(GG_WAITCNT_IMM(VM)); __builtin_amdgcn_s_barrier(); } while (0)
#define GG_BOX_LDS_BARRIER() do { __builtin_amdgcn_s_waitcnt(GG_WAITCNT_IMM(63)); __builtin_amdgcn_s_barrier(); } while (0)

// Diagnostic build only (tools/experiments/README.md): -DGG_BOX_STAMPS records s_memrealtime (100 MHz) phase stamps of waves 0 and 7
// of every workgroup into gg_conv_desc.workspace when path_hint == 98.
#ifdef GG_BOX_STAMPS
#define GG_STAMP(K) do { if (p.path_hint == 98 && p.ws && lane == 0 && (wave == 0 || wave == GG_BOX_NW - 1)) \
    reinterpret_cast<unsigned long long *>(p.ws)[(blockIdx.x * 2 + (wave ? 1 : 0)) * 16 + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GG_STAMP(K) do { } while (0)
#endif

// Exact division of small non-negative integers by run-time constants: q = (n * ceil(2^32 / d)) >> 32 for n * d < 2^32 (block ids,
// k-steps and DMA units are all < 2^16).  One s_mul_hi_u32 instead of the ~30-instruction 32-bit division sequence: at batch 1 the
// ~520 serially issued instructions between kernel entry and the first DMA were 1.1 us of a 10 us kernel (phase stamps).
struct BoxMagic { unsigned pq, tw, th, nch, nch_last; int Q; unsigned nch_s, nch_s_last; int nstage_s, nch_stage_s; };
__device__ __forceinline__ int gg_mdiv(int n, unsigned magic) { return (int)__umulhi((unsigned)n, magic); }
static unsigned gg_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }   // d == 1: handled by the caller

template <int TWI, int MT, int CT, int UP, int K3, int SK = 0, int NS = 1>
__global__ __launch_bounds__(GG_BOX_NW * 64) void conv_box2d_kernel(const ConvParams p_arg, const int tiles_h_arg, const int tiles_w_arg, const int nstage_arg,
                                                         const int nch_stage_arg, const int gn_bytes_arg, const int q_major_arg, const int nblocks_arg, const BoxMagic mg_arg)
{
    // the arguments the block decode and the first DMAs need, in ONE scalar-load batch (gg_pin); the rest load lazily
    ConvParams p = p_arg;
    p.N = gg_pin(p_arg.N); p.H = gg_pin(p_arg.H); p.W = gg_pin(p_arg.W); p.C1 = gg_pin(p_arg.C1); p.C2 = gg_pin(p_arg.C2);
    p.nchunk1 = gg_pin(p_arg.nchunk1); p.nchunk = gg_pin(p_arg.nchunk); p.src1 = gg_pin(p_arg.src1); p.src2 = gg_pin(p_arg.src2);
    p.bias = gg_pin(p_arg.bias); p.bias_stride = gg_pin(p_arg.bias_stride); p.prologue_act = gg_pin(p_arg.prologue_act);
    // ... and one field of every other 64-byte line of the kernarg segment (0xc0, 0x100, 0x140): a later scalar load of a line nobody
    // has touched is a full memory round trip, and the compiler had five of them, serial, between kernel entry and the first DMA
    p.Cout = gg_pin(p_arg.Cout); p.Cout_pad = gg_pin(p_arg.Cout_pad);
    p.gn_acc = gg_pin(p_arg.gn_acc); p.pro_acc1 = gg_pin(p_arg.pro_acc1); p.pro_clog = gg_pin(p_arg.pro_clog); p.skip_C1 = gg_pin(p_arg.skip_C1);
    const int tiles_h = gg_pin(tiles_h_arg), tiles_w = gg_pin(tiles_w_arg), nstage = gg_pin(nstage_arg), nch_stage = gg_pin(nch_stage_arg),
              gn_bytes = gg_pin(gn_bytes_arg), q_major = gg_pin(q_major_arg), nblocks = gg_pin(nblocks_arg);
    BoxMagic mg;
    mg.pq = gg_pin(mg_arg.pq); mg.tw = gg_pin(mg_arg.tw); mg.th = gg_pin(mg_arg.th); mg.nch = gg_pin(mg_arg.nch);
    mg.nch_last = gg_pin(mg_arg.nch_last); mg.Q = gg_pin(mg_arg.Q);
    mg.nch_s = mg_arg.nch_s; mg.nch_s_last = mg_arg.nch_s_last; mg.nstage_s = mg_arg.nstage_s; mg.nch_stage_s = mg_arg.nch_stage_s;
    // an MFMA position tile (16 positions) is RPT rows x TWI columns: one 16-wide row, 2 x 8 or 4 x 4 (deep UNet levels)
    constexpr int TW = TWI, NW = GG_BOX_NW, NTH = NW * 64;
    // Weight tiles of the 8- / 4-wide shapes (the 8x8 / 4x4 levels: 400 of the 535 MB of weights, each byte read by one to four
    // workgroups per forward) by non-temporal loads (gg_common.h): with the default policy they evict what the next kernels read.
    // Where many position tiles share a weight slice (64x64 .. 16x16) the default policy keeps their L2 hits.  Per captured forward:
    // none 1385 us, every shape 1330, 8- / 4-wide only 1308 (N = 1 @64x64); N = 4 @32x32 1435 -> 1365; N = 8 @64x64 3866 -> 3878.
    constexpr bool WNT = TWI < 16;
    auto WLOAD = [](const auto *ptr) { if constexpr (WNT) return GG_STREAM_LOAD(ptr); else return *ptr; };
    constexpr int RPT = 16 / TWI;
    constexpr int TH = MT * RPT;                      // output rows of the workgroup
    // Weight trips (4 k-steps each) kept in flight per wave: measured on the latent-UNet forward (same box, hipGraph replay):
    // 1 trip 1707 us, 2 trips 1675, 3 trips (CT 1) 1694, 6 / 4 trips where the registers allow 1744.  What a CU can take in is the
    // bound, so weight tiles requested early only delay the landing of the box, i.e. the start of the k-loop.
    // A deeper ring topped up AFTER the box has landed (6 / 4 trips, host-gated to shares that fill it) is slower too (1648 vs 1606 us):
    // the phase stamps show the k-loop at the same 3.5-3.8 us either way, i.e. it is not a latency chain but the same intake bound.
    // Cout sub-split: a weight load instruction moves 1 / NS of the bytes, so the ring is NS times deeper for the same bytes in flight
    // (phase stamps of the 800 -> 800 conv at 4x4: 50 workgroups x 230 KB and 200 x 58 KB both take ~8 us entry to end -- 1.7 us to
    // the first DMA, ~2.8 us until the box has landed, ~2 us of k-loop, 1.2 us of combine and epilogue; the split buys ~0.5 us).
    constexpr int NTRIP = 2 * NS * (8 / NW);   // (3x3 with (kh, chunk) units: 2 units = 6 k-steps in flight 1511 us per forward, 3 units 1514)
    constexpr int PADK = K3 ? 1 : 0, NTAPS = K3 ? 9 : 1;      // 3x3 pad 1, or 1x1 (the box is then the tile itself)
    // UP: 0 plain, 1 fused nearest x2 upsample, 2 STRIDE 2 (the UNet's Downsample convs: 3x3, pad 1): the box is (2 TH + 1) x (2 TW + 1)
    // input positions, its columns stored de-interleaved inside a line (slots 0 .. TW: even columns, TW + 1 .. 2 TW: odd columns), so
    // that the 16 lanes of an operand read (input columns 2 c + kw) touch consecutive slots as in the stride-1 box
    constexpr bool S2 = UP == 2;
    static_assert(!S2 || K3, "stride 2: 3x3 only");
    constexpr int HH = UP == 1 ? TH / 2 + 2 : S2 ? 2 * TH + 1 : TH + 2 * PADK;
    constexpr int HW = UP == 1 ? TW / 2 + 2 : S2 ? 2 * TW + 1 : TW + 2 * PADK;
    auto colof = [](int hw) -> int { return S2 ? (hw <= TW ? 2 * hw : 2 * (hw - TW - 1) + 1) : hw; };       // input column of a line slot
    constexpr int NROWS = HH * HW;
    constexpr int NRB = (NROWS + 15) / 16;            // 1 KiB DMA blocks (16 rows) per chunk plane
    constexpr int PLANE = NRB * 1024;                 // one 32-channel chunk of the box
    // Box image swizzle (16-byte chunk ^ sw).  3x3 boxes use a function of the position INSIDE a W-line only (exhaustive search
    // over the ds_read_b128 lane groups for the three kw taps: TWI 16 -> hw in {4,5,10..15}, upsample {6..9}; TWI 8 -> {2,3,6,7};
    // TWI 4 -> {2,3}): unlike the row-based map it is invariant under whole-line shifts (kh, the position tile), so an operand
    // address is a per-lane constant + a wave-uniform term + an immediate.  The row-based map cost 5 VALU per operand read,
    // 60 per k-step at 12 position tiles, MORE issue cycles than the k-step's 12 MFMAs (timing ablation: 1.76 -> 1.70 ms per
    // latent-UNet forward).  No such map exists for the upsampled 8- / 4-wide boxes; they keep the row-based one (2 launches per
    // forward), and so do 1x1 boxes, where the row-based map is already a lane constant.
    constexpr bool LINE_SWZ = K3 && !S2 && (TWI == 16 || !UP);
    constexpr unsigned FMASK = TWI == 16 ? (UP == 1 ? 0x3C0u : 0xFC30u) : TWI == 8 ? 0xCCu : 0x0Cu;
    constexpr bool LANE_ADDR = LINE_SWZ || !K3;       // operand address = lane constant + uniform + immediate
    auto bsw = [&](int row, int hw) -> int { return LINE_SWZ ? (int)((FMASK >> hw) & 1u) << 1 : (row >> 1) & 2; };
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float *gns = reinterpret_cast<float *>(smem);     // [nch*32] scale, [nch*32] shift of the stage (fused prologue only)
    char *box = smem + gn_bytes;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int pos_r = fr / TWI, pos_c = fr % TWI;     // this lane's position inside a position tile
    GG_STAMP(0);

    // ---- workgroup -> (position tile, cout tile).  Consecutive hardware ids round-robin over the 8 XCDs; give every XCD a
    //      contiguous run of virtual ids, then decode them cout-major (a run shares weights) or position-major (shares boxes).
    const int P = p.N * tiles_h * tiles_w, Q = mg.Q;
    int v = blockIdx.x;
    if ((nblocks & 7) == 0) v = (v & 7) * (nblocks >> 3) + (v >> 3);
    // (divisors of 1 have magic 0: the quotient is the dividend)
    auto mdiv = [](int a, int d, unsigned m) { return d == 1 ? a : gg_mdiv(a, m); };
    const int vq = mdiv(v, q_major ? P : Q, mg.pq);                     // v / P (cout-major) or v / Q (position-major)
    int by = q_major ? vq : v - vq * Q;
    // Cout sub-split (NS 2 / 4, weight-bound 3x3 convs of the 8x8 / 4x4 levels): NS workgroups share a 16-cout tile and take 16 / NS of
    // its weight rows each (the other lanes' operand registers are zeros: no load), so the level's weight stream spreads over 4x the
    // CUs without a cross-workgroup reduction; each stores (and adds the GroupNorm sums of) its own couts only.
    static_assert(NS == 1 || (CT == 1 && (NS == 2 || NS == 4)), "cout sub-split: single cout tile, 2 or 4 ways");
    constexpr int CSZ = 16 / NS;
    const int sub = by & (NS - 1);
    by >>= (NS == 4 ? 2 : NS == 2 ? 1 : 0);
    const bool wact = NS == 1 || (fr / CSZ) == sub;                       // this lane's weight row belongs to the workgroup
    const bool oact = NS == 1 || (((lane >> 4) * 4) / CSZ) == sub;        // this lane's 4 accumulator couts do
    int t = q_major ? v - vq * P : vq;
    const int stripe = t & (GG_ACC_STRIPES - 1);       // GroupNorm accumulator stripe of this position tile
    const int t1 = mdiv(t, tiles_w, mg.tw);
    const int tw = t - t1 * tiles_w;
    const int n = mdiv(t1, tiles_h, mg.th);
    const int th = t1 - n * tiles_h;
    const int h0 = th * TH, w0 = tw * TW;
    const int g = CT == 2 ? by : by >> 1, half = CT == 2 ? 0 : by & 1;
    const int ih0 = UP == 1 ? h0 / 2 - 1 : S2 ? 2 * h0 - 1 : h0 - PADK;
    const int iw0 = UP == 1 ? w0 / 2 - 1 : S2 ? 2 * w0 - 1 : w0 - PADK;

    // final pass: thread -> f32x4 slot (tid & 63) of slices (tid >> 6) + 8k; with CT | 8 its 4 couts are the same for every k,
    // so the bias is fetched here, a whole kernel ahead of its use
    int co_thr = 0;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};           // (requested right after the first stage's DMAs have been issued)

    // accumulators, weight base and operand lane offsets are set up AFTER the first stage's DMAs have been issued (a wave issues its
    // instructions one by one: whatever precedes the DMAs delays the landing of the box)
    f32x4 acc[MT][CT];
    const bf16_t *wbase = nullptr;
    int wl0 = 0;
    int lane_off[3] = {0, 0, 0};

    // staging duty of a lane inside a 16-row DMA block: row (lane>>2), LDS slot (lane&3)
    const int lrow = lane >> 2, lslot = lane & 3;

    // GroupNorm prologue FROM ACCUMULATORS (gg_conv_desc.pro_acc1): this thread's channels tid + 512k of the per-channel fixed-point
    // (sum, sumsq) the producing convs left, gamma and beta, requested ahead of the box DMAs (vmcnt counts in order: they have
    // landed when the box has); folded into the scale / shift table of ALL input channels once the first box is in LDS.
    const bool acc_mode = p.prologue_act && p.pro_acc1 != nullptr;
    constexpr int ACPT = 2048 / NTH;                   // channels per thread: C1 + C2 <= 2048 (host gate)
    float pgam[ACPT], pbet[ACPT];
    long long psum = 0;                                // 8 lanes per (group, sum | sumsq) task
    __shared__ float pro_gmean[32], pro_grstd[32];
    if (acc_mode) {
        const int tidh = gg_here(tid);
#pragma unroll
        for (int k = 0; k < ACPT; ++k) {
            const int c = tidh + NTH * k;
            pgam[k] = 0.f;
            pbet[k] = 0.f;
            if (c < p.pro_clog) {
                pgam[k] = p.pro_gamma[c];
                pbet[k] = p.pro_beta[c];
            }
        }
    }

    for (int st = 0; st < nstage; ++st) {
        const int cbase = st * nch_stage;
        const int nch = (p.nchunk - cbase < nch_stage) ? p.nchunk - cbase : nch_stage;
        const int S = NTAPS * nch;
        const int s0 = (S * wave) / NW, s1 = (S * (wave + 1)) / NW;
        const unsigned mnch = nch == nch_stage ? mg.nch : mg.nch_last;       // magic of this stage's chunk count

        // GroupNorm scale/shift rows of the stage -> LDS, by DMA as well (256 floats per wave instruction)
        if (p.prologue_act && !acc_mode) {
            const int gn_units = 2 * ((nch + 7) >> 3);
            for (int u = wave; u < gn_units; u += NW) {
                const int which = u & 1, blk = u >> 1;
                const float *gsrc = (which ? p.gn_shift : p.gn_scale) + (long long)n * (p.C1 + p.C2) + cbase * 32 + blk * 256 + lane * 4;
                if (blk * 256 + lane * 4 < nch * 32)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                                     (__attribute__((address_space(3))) void *)(gns + which * nch * 32 + blk * 256), 16, 0, 0);
            }
        }
        // ---- stage the box of this stage's channels (global_load_lds, everything in flight); padding rows are zeros.
        //      Units (16-row block rbk, chunk c) are walked block-major, each wave a contiguous eighth: everything that depends on the
        //      block only (the lane's box row -> input position, padding, swizzle, element offsets) is set up when rbk changes, so a
        //      unit costs ~a dozen instructions.  (Chunk-major units recomputed it per DMA: ~70 instructions, and the phase stamps
        //      showed 4-5 us between kernel entry and the last DMA issued for a 640->640 conv at 16x16.)
        const int nunit = nch * NRB;
        const int u0 = (nunit * wave) / NW, u1 = (nunit * (wave + 1)) / NW;
        bool inr = false, valid = false;
        unsigned off1 = 0u, off2 = 0u;          // byte offsets inside the sample: < 2^32 (checked on the host)
        auto setup = [&](int rb) {
            const int row = rb * 16 + lrow;
            const int hh = row / HW, hw = row - hh * HW;
            const int ih = ih0 + hh, iw = iw0 + colof(hw);
            inr = row < NROWS;
            valid = inr && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const unsigned pos = valid ? (unsigned)(ih * p.W + iw) : 0u;
            const unsigned q8 = (unsigned)((lslot ^ bsw(row, hw)) * 8);
            off1 = (pos * (unsigned)p.C1 + q8) * 2u;
            off2 = (pos * (unsigned)p.C2 + q8) * 2u;
        };
        const char *s1n = reinterpret_cast<const char *>(p.src1 + (long long)n * p.H * p.W * p.C1);
        const char *s2n = reinterpret_cast<const char *>(p.src2 + (long long)n * p.H * p.W * p.C2);
        // DMA issue: every lane always issues (padding and past-the-box rows from a clamped, legal address; the padding slots are
        // zeroed by their own wave after its DMAs have landed, below).  A wave issues its instructions one by one, so the instruction
        // count per unit IS the staging time at batch 1 (stamps: 170 ns per unit with a predicated DMA / zero-store pair and the
        // (block, chunk) bookkeeping per unit; 15 units per wave).  Hence runs: within one 16-row block and one source tensor,
        // consecutive chunks are +64 B in global memory and +PLANE in LDS, and nothing else changes.
        {
            int rbk = mdiv(u0, nch, mnch), c = u0 - rbk * nch;
            int left = u1 - u0;
            GG_STAMP(8);
            while (left > 0) {
                setup(rbk);
                int run = nch - c < left ? nch - c : left;                       // units of this block
                left -= run;
                int gc = cbase + c;
                char *dst = box + c * PLANE + rbk * 1024;
                // first source, then (two-source concat) second source
                int n1 = p.nchunk1 - gc;
                n1 = n1 < 0 ? 0 : (n1 > run ? run : n1);
                const char *sb = s1n + gc * 64;
#pragma unroll 2
                for (int i = 0; i < n1; ++i) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + off1),
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                    sb += 64;
                    dst += PLANE;
                }
                sb = s2n + (gc + n1 - p.nchunk1) * 64;
#pragma unroll 2
                for (int i = n1; i < run; ++i) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + off2),
                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                    sb += 64;
                    dst += PLANE;
                }
                c = 0;
                ++rbk;
            }
        }
        GG_STAMP(1);
        if (st == 0) {
            const int gh = gg_pin(g), halfh = gg_pin(half), nh = gg_pin(n);      // (pinned: what follows stays behind the DMA issue)
            {
                const int tidh = gg_here(tid);
                co_thr = gh * 32 + halfh * 16 + ((tidh >> 6) % CT) * 16 + ((tidh & 63) >> 4) * 4;
                if (p.bias) bias4 = *reinterpret_cast<const f32x4 *>(p.bias + (long long)nh * p.bias_stride + co_thr);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            wbase = p.weight + ((long long)gh * NTAPS * p.nchunk << 10) + halfh * 512;
            const int laneh = gg_here(lane), frh = laneh & 15, fqh = laneh >> 4;      // (not hoisted in front of the DMAs)
            wl0 = frh * 32 + swz64(frh, fqh) * 8;      // pre-swizzled packed rows: cout row fr (and 16 + fr at +512 elements)
            // per-lane part of the activation-operand address for the three kw taps (1x1: one)
            if constexpr (LINE_SWZ) {
                const int pos_rh = frh / TWI, pos_ch = frh % TWI;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int rwk = UP == 1 ? ((pos_ch + k + 1) >> 1) : (pos_ch + k);
                    lane_off[k] = (UP == 1 ? 0 : pos_rh * (HW * 64)) + rwk * 64 + ((fqh ^ bsw(0, rwk)) * 16);
                }
            } else if constexpr (!K3) {
                lane_off[0] = frh * 64 + ((fqh ^ ((frh >> 1) & 2)) * 16);       // row = 16 * tile + fr: the row-based map only sees fr
            }
        }
        // ---- weight stream of this wave.  3x3: the wave owns units q in [q0, q1) of the (kh, chunk) grid, a unit = the three kw taps
        //      of one chunk plane and one kh, so kw is a compile-time constant in the k-loop (operand lane offsets, weight tile stride)
        //      and the scalar bookkeeping is paid once per three k-steps: at batch 1 the k-loop was bound by its ~50 scalar / branch
        //      instructions per k-step, not by the 3-12 MFMAs in it (a deeper weight ring did not shorten it).  1x1: steps = chunks,
        //      four per trip.  Loads past the end re-read the last unit (unconditional, branch-free: the vmcnt counts stay exact).
        //      Issued AFTER the box so the box lands first.
        constexpr int SPT = K3 ? 3 : 4;                                           // k-steps per trip
        const int TU = K3 ? 3 * nch : nch;                                        // trips-units of the stage: (kh, chunk) units / chunks
        const int q0 = K3 ? (TU * wave) / NW : s0, q1 = K3 ? (TU * (wave + 1)) / NW : s1;
        int lkh = K3 ? mdiv(q0, nch, mnch) : 0, lc = q0 - lkh * nch, lidx = q0;   // load iterator
        bf16x8 wr[NTRIP][SPT][CT];
        if constexpr (NS > 1) {
            // an MFMA output row (cout) depends on its own weight row only: the rows of the other sub-workgroups are never loaded, their
            // accumulator rows never stored (zeros once, so that no NaN pattern is ever fed to the matrix core)
#pragma unroll
            for (int r = 0; r < NTRIP; ++r)
#pragma unroll
                for (int u = 0; u < SPT; ++u)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) wr[r][u][ct] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        auto load_w = [&](bf16x8 (&a)[SPT][CT]) {
            if constexpr (K3) {
                const bf16_t *tile = wbase + (((long long)(lkh * 3) * p.nchunk + cbase + lc) << 10) + wl0;
                const long long kws = (long long)p.nchunk << 10;                  // next tap of the same chunk
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        if constexpr (NS == 1) a[u][ct] = WLOAD(reinterpret_cast<const bf16x8 *>(tile + u * kws + ct * 512));
                        else if (wact) a[u][ct] = WLOAD(reinterpret_cast<const bf16x8 *>(tile + u * kws + ct * 512));     // (foreign rows: stale registers)
                    }
                const int adv = (lidx + 1 < q1) ? 1 : 0;
                lidx += adv;
                lc += adv;
                const int wrap = (lc == nch) ? 1 : 0;
                lc = wrap ? 0 : lc;
                lkh += wrap;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bf16_t *tile = wbase + ((long long)(cbase + lc) << 10) + wl0;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) a[u][ct] = WLOAD(reinterpret_cast<const bf16x8 *>(tile + ct * 512));
                    lc += (lc + 1 < q1) ? 1 : 0;
                }
            }
        };
        asm volatile("" ::: "memory");                 // keep the weight loads behind the DMA issue
#pragma unroll
        for (int r = 0; r < NTRIP; ++r) load_w(wr[r]);
        // the box (and the scale/shift rows) have landed once at most this wave's NTRIP*4*CT weight loads are outstanding
        GG_STAMP(2);
        if (acc_mode && st == 0) {
            // 64 tasks (32 groups x {sum, sumsq}), 8 lanes each: lane `part` reads channels part, part + 8, ... of the group's cpg <= 64
            // per-channel fixed-point sums straight from L2 -- at most 8 loads per lane, ALL in flight at once, while the box is landing
            // (no LDS atomics, no staging area; integer adds are exact in any order)
            const int tidh = gg_here(tid);
            constexpr int LPT = NTH / 64;                     // lanes per task (8 waves: 8, 4 waves: 4)
            const int task = tidh / LPT, part = tidh % LPT, gg = task >> 1, which = task & 1;
            const int cpg = p.pro_clog >> 5;
            long long v[64 / LPT];
#pragma unroll
            for (int i = 0; i < 64 / LPT; ++i) {
                const int j = part + LPT * i, c = gg * cpg + j;
                v[i] = 0;
                if (j < cpg) {
                    const long long *q = (c < p.C1) ? p.pro_acc1 + ((long long)n * p.C1 + c) * 2 : p.pro_acc2 + ((long long)n * p.C2 + (c - p.C1)) * 2;
                    v[i] = q[which];
                }
            }
#pragma unroll
            for (int i = 0; i < 64 / LPT; ++i) psum += v[i];
        }
        // this wave's DMAs have landed once only its NTRIP*SPT*CT weight loads are outstanding; then zero ITS padding slots; then barrier
        __builtin_amdgcn_s_waitcnt(GG_WAITCNT_IMM(NTRIP * SPT * CT));
        if (ih0 < 0 || iw0 < 0 || ih0 + HH > p.H || iw0 + HW > p.W) {        // border workgroups only (wave-uniform)
            int rbk = mdiv(u0, nch, mnch), c = u0 - rbk * nch;
            setup(rbk);
#pragma unroll 1
            for (int u = u0; u < u1; ++u) {
                if (inr && !valid) *reinterpret_cast<u32x4 *>(box + c * PLANE + rbk * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
                if (++c == nch) { c = 0; ++rbk; setup(rbk); }
            }
        }
        GG_BOX_LDS_BARRIER();

        if (acc_mode && st == 0) {
            // groups -> mean / rstd (fp64; the sum | sumsq lanes of a group are neighbours), channels -> scale / shift rows of ALL chunks
            const int tidh = gg_here(tid);
            const int cpg = gg_pin(p.pro_clog) >> 5;              // (pinned here: the fp64 reciprocal below is not hoisted in front of the DMAs)
            const float rcpg = __builtin_amdgcn_rcpf((float)cpg);
            {
                constexpr int LPT = NTH / 64;
                psum += __shfl_xor(psum, 1);
                psum += __shfl_xor(psum, 2);
                if constexpr (LPT == 8) psum += __shfl_xor(psum, 4);      // the LPT parts of a task
                const long long other = __shfl_xor(psum, LPT);            // sumsq task of the same group sits LPT lanes up
                if ((tidh & (2 * LPT - 1)) == 0) {
                    const double a = (double)psum * (1.0 / (double)GG_ACC_SUM_SCALE);
                    const double b = (double)other * (1.0 / (double)GG_ACC_SQ_SCALE);
                    const double cnt = (double)p.H * (double)p.W * (double)cpg;
                    double inv = (double)(1.0f / (float)cnt);
                    inv = inv * (2.0 - cnt * inv);             // fp32 reciprocal + one Newton step in fp64 (as gn_apply_acc_kernel)
                    const double mean = a * inv;
                    double var = b * inv - mean * mean;
                    if (var < 0.0) var = 0.0;
                    pro_gmean[tidh / (2 * LPT)] = (float)mean;
                    pro_grstd[tidh / (2 * LPT)] = rsqrtf((float)var + p.pro_eps);
                }
            }
            GG_BOX_LDS_BARRIER();
            const int Ct = p.nchunk * 32;
#pragma unroll
            for (int k = 0; k < ACPT; ++k) {
                const int c = tidh + NTH * k;
                if (c < Ct) {
                    float sc = 0.f, sh = 0.f;
                    if (c < p.pro_clog) {
                        const int gg = gg_div_small(c, rcpg);
                        sc = pro_grstd[gg] * pgam[k];
                        sh = pbet[k] - pro_gmean[gg] * sc;
                    }
                    gns[c] = sc;
                    gns[Ct + c] = sh;
                }
            }
            GG_BOX_LDS_BARRIER();
        }
        if (p.prologue_act) {     // GroupNorm affine (* SiLU) in place, once per staged element; padding stays zero
            const int laneh = gg_here(lane), lrowh = laneh >> 2, lsloth = laneh & 3;
            // rows of the table: external tables hold this stage's chunks only, the accumulator fold holds all chunks of the conv
            const float *gsc = gns + (acc_mode ? cbase * 32 : 0);
            const int gsh = (acc_mode ? p.nchunk : nch) * 32;
            auto xform = [&](char *pc, const f32x4 sc0, const f32x4 sc1, const f32x4 sh0, const f32x4 sh1) {
                bf16x8 xb = *reinterpret_cast<const bf16x8 *>(pc), yb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y0 = (float)xb[e] * sc0[e] + sh0[e], y1 = (float)xb[e + 4] * sc1[e] + sh1[e];
                    if (p.prologue_act == 1) {
                        y0 = y0 * __builtin_amdgcn_rcpf(1.0f + __expf(-y0));
                        y1 = y1 * __builtin_amdgcn_rcpf(1.0f + __expf(-y1));
                    }
                    yb[e] = (bf16_t)y0;
                    yb[e + 4] = (bf16_t)y1;
                }
                *reinterpret_cast<bf16x8 *>(pc) = yb;
            };
            if constexpr (!K3) {
                // 1x1 boxes (row-based swizzle): the lane's 8 channels of a chunk are the same in every 16-row block, so its scale / shift
                // rows are read once per chunk, not once per piece
                const int q = lsloth ^ bsw(lrowh, 0);
                for (int c = wave; c < nch; c += NW) {
                    const float *sc = gsc + c * 32 + q * 8, *sh = sc + gsh;
                    const f32x4 sc0 = *reinterpret_cast<const f32x4 *>(sc), sc1 = *reinterpret_cast<const f32x4 *>(sc + 4);
                    const f32x4 sh0 = *reinterpret_cast<const f32x4 *>(sh), sh1 = *reinterpret_cast<const f32x4 *>(sh + 4);
#pragma unroll 2
                    for (int rbk = 0; rbk < NRB; ++rbk) {
                        const int row = rbk * 16 + lrowh;
                        const int hh = row / HW, hw = row - hh * HW;
                        const int ih = ih0 + hh, iw = iw0 + colof(hw);
                        if (row < NROWS && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) xform(box + c * PLANE + rbk * 1024 + laneh * 16, sc0, sc1, sh0, sh1);
                    }
                }
            } else {
#pragma unroll 2
                for (int unit = wave; unit < nunit; unit += NW) {
                    const int c = unit / NRB, rbk = unit - c * NRB;
                    const int row = rbk * 16 + lrowh;
                    const int hh = row / HW, hw = row - hh * HW;
                    const int ih = ih0 + hh, iw = iw0 + colof(hw);
                    if (row < NROWS && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) {
                        const int q = lsloth ^ bsw(row, hw);
                        const float *sc = gsc + c * 32 + q * 8, *sh = sc + gsh;
                        xform(box + c * PLANE + rbk * 1024 + laneh * 16, *reinterpret_cast<const f32x4 *>(sc), *reinterpret_cast<const f32x4 *>(sc + 4),
                              *reinterpret_cast<const f32x4 *>(sh), *reinterpret_cast<const f32x4 *>(sh + 4));
                    }
                }
            }
            GG_BOX_LDS_BARRIER();
        }

        GG_STAMP(3);
        // ---- this wave's k-steps
        int ckh = K3 ? mdiv(q0, nch, mnch) : 0, cc = q0 - ckh * nch;            // compute iterator: (kh, chunk) unit / chunk
        auto kstep = [&](const bf16x8 (&w)[CT], const char *plane, const int kh, const int kw) {     // kw: compile-time after unrolling
            bf16x8 xf[MT];
            if constexpr (LANE_ADDR) {
                const int lo = lane_off[K3 ? kw : 0];
                if constexpr (!UP) {
                    const char *pa = plane + kh * (HW * 64) + lo;                 // one VALU add per k-step
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt) xf[tt] = *reinterpret_cast<const bf16x8 *>(pa + tt * (K3 ? RPT * HW * 64 : 1024));
                } else {                                                          // TWI == 16: line (tt + kh + 1) >> 1
                    const char *pe = plane + ((kh + 1) >> 1) * (HW * 64) + lo, *po = plane + ((kh + 2) >> 1) * (HW * 64) + lo;
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt)
                        xf[tt] = *reinterpret_cast<const bf16x8 *>(((tt & 1) ? po : pe) + (tt >> 1) * (HW * 64));
                }
            } else {
                // per-lane column slot of the operand row (stride 2: input column 2 c + kw -> even slot c + kw / 2, or odd slot TW + 1 + c)
                const int rwk = UP == 1 ? ((pos_c + kw + 1) >> 1) : S2 ? ((kw & 1) ? TW + 1 + pos_c : pos_c + (kw >> 1)) : (pos_c + kw);
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) {
                    const int orow = tt * RPT + pos_r;
                    const int hh = UP == 1 ? ((orow + kh + 1) >> 1) : S2 ? 2 * orow + kh : orow + kh;
                    const int row = hh * HW + rwk;
                    xf[tt] = *reinterpret_cast<const bf16x8 *>(plane + row * 64 + swz64(row, fq) * 16);
                }
            }
            // all MT operand reads are issued before the first MFMA (the scheduler would otherwise pair them two by two to save
            // registers and expose the LDS latency once per pair); the MFMAs then drain them under counted lgkmcnt
            if constexpr (MT * CT <= 12) __builtin_amdgcn_sched_barrier(0);      // (12 x 2: the 48 operand registers would spill)
#pragma unroll
            for (int tt = 0; tt < MT; ++tt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ct], xf[tt], acc[tt][ct], 0, 0, 0);
        };
        auto trip = [&](const bf16x8 (&a)[SPT][CT], int q) {
            if constexpr (K3) {
                if (q < q1) {
                    const char *plane = box + cc * PLANE;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) kstep(a[kw], plane, ckh, kw);
                    if (++cc == nch) { cc = 0; ++ckh; }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (q + u < q1) { kstep(a[u], box + cc * PLANE, 0, 0); ++cc; }
            }
        };
        int q = q0;
#pragma unroll 1
        for (; q + (K3 ? 1 : 4) * NTRIP < q1; q += (K3 ? 1 : 4) * NTRIP) {       // steady state: at least one of the refilled trips is real
#pragma unroll
            for (int r = 0; r < NTRIP; ++r) {
                trip(wr[r], q + (K3 ? 1 : 4) * r);
                load_w(wr[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < NTRIP; ++r) trip(wr[r], q + (K3 ? 1 : 4) * r);      // drain: nothing left to load
        GG_STAMP(4);
        GG_BOX_WAIT_BARRIER(0);   // all waves done with the box: it may be overwritten (next stage / the reduction area)
        GG_STAMP(5);
    }

    // ---- K-concatenated 1x1 skip projection (gg_conv_desc.skip_src1): out = conv3x3(act(GN(h1))) + conv1x1(x) in ONE launch.  The raw x
    //      tile of the workgroup's own positions (no halo: TH x TW rows, the 1x1 geometry with the row-based swizzle) is staged into
    //      the same LDS region in stages of its own after the 3x3 stages, and its k-steps (one per 32-channel chunk) add into the
    //      same accumulators: the separate skip-conv launch (~6.4 us at batch 1, 18 per latent-UNet forward) and the residual round
    //      trip of the epilogue are gone; the price is the x tile's bytes into every CU.
    if constexpr (SK) {
        constexpr int PLANE1 = MT * 1024;                 // one chunk plane of the tile: MT 16-row blocks
        constexpr int SPT1 = 3;                           // chunk k-steps per weight trip (reuses the 3x3 ring's shape)
        const int Cs1 = p.skip_C1, Cs2 = p.skip_C2, nck1 = Cs1 >> 5, nck = (Cs1 + Cs2) >> 5;
        const int lo1 = fr * 64 + ((fq ^ ((fr >> 1) & 2)) * 16);            // row = 16 * tile + fr: the row-based map only sees fr
        const bf16_t *wsk = p.skip_weight + ((long long)g * nck << 10) + half * 512;      // [Cout_pad / 32][1 tap][nck][32 co][32 ci]
        const char *x1n = reinterpret_cast<const char *>(p.skip_src1 + (long long)n * p.H * p.W * Cs1);
        const char *x2n = reinterpret_cast<const char *>(p.skip_src2 + (long long)n * p.H * p.W * Cs2);
        for (int st = 0; st < mg.nstage_s; ++st) {
            const int cbase = st * mg.nch_stage_s;
            const int nch = (nck - cbase < mg.nch_stage_s) ? nck - cbase : mg.nch_stage_s;
            const unsigned mnch = nch == mg.nch_stage_s ? mg.nch_s : mg.nch_s_last;
            const int nunit = nch * MT;
            const int u0 = (nunit * wave) / NW, u1 = (nunit * (wave + 1)) / NW;
            {
                int rbk = mdiv(u0, nch, mnch), c = u0 - rbk * nch;
                int left = u1 - u0;
                while (left > 0) {
                    const int row = rbk * 16 + lrow;
                    const int hh = row / TW, hw = row - hh * TW;
                    const int ih = h0 + hh, iw = w0 + hw;
                    const unsigned pos = (ih < p.H) ? (unsigned)(ih * p.W + iw) : 0u;   // rows past a ragged last tile: any legal address (outputs not stored)
                    const unsigned q8 = (unsigned)((lslot ^ ((row >> 1) & 2)) * 8);
                    const unsigned o1 = (pos * (unsigned)Cs1 + q8) * 2u, o2 = (pos * (unsigned)Cs2 + q8) * 2u;
                    int run = nch - c < left ? nch - c : left;
                    left -= run;
                    const int gc = cbase + c;
                    char *dst = box + c * PLANE1 + rbk * 1024;
                    int n1 = nck1 - gc;
                    n1 = n1 < 0 ? 0 : (n1 > run ? run : n1);
                    const char *sb = x1n + gc * 64;
#pragma unroll 2
                    for (int i = 0; i < n1; ++i) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + o1),
                                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                        sb += 64;
                        dst += PLANE1;
                    }
                    sb = x2n + (gc + n1 - nck1) * 64;
#pragma unroll 2
                    for (int i = n1; i < run; ++i) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + o2),
                                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                        sb += 64;
                        dst += PLANE1;
                    }
                    c = 0;
                    ++rbk;
                }
            }
            // weight stream: this wave's chunks [q0, q1), three per trip, straight from L2 into VGPRs (as the 3x3 stages)
            const int q0 = (nch * wave) / NW, q1 = (nch * (wave + 1)) / NW;
            int lc = q0;
            bf16x8 ws[2][SPT1][CT];
            if constexpr (NS > 1) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int u = 0; u < SPT1; ++u)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) ws[r][u][ct] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
            auto load_ws = [&](bf16x8 (&a)[SPT1][CT]) {
#pragma unroll
                for (int u = 0; u < SPT1; ++u) {
                    const bf16_t *tile = wsk + ((long long)(cbase + lc) << 10) + wl0;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        if constexpr (NS == 1) a[u][ct] = WLOAD(reinterpret_cast<const bf16x8 *>(tile + ct * 512));
                        else if (wact) a[u][ct] = WLOAD(reinterpret_cast<const bf16x8 *>(tile + ct * 512));
                    }
                    lc += (lc + 1 < q1) ? 1 : 0;
                }
            };
            asm volatile("" ::: "memory");                 // keep the weight loads behind the DMA issue
            load_ws(ws[0]);
            load_ws(ws[1]);
            __builtin_amdgcn_s_waitcnt(GG_WAITCNT_IMM(2 * SPT1 * CT));      // this wave's DMAs have landed
            GG_BOX_LDS_BARRIER();
            int cc = q0;
            auto trip_s = [&](const bf16x8 (&a)[SPT1][CT], int q) {
#pragma unroll
                for (int u = 0; u < SPT1; ++u)
                    if (q + u < q1) {
                        const char *pa = box + cc * PLANE1 + lo1;
                        bf16x8 xf[MT];
#pragma unroll
                        for (int tt = 0; tt < MT; ++tt) xf[tt] = *reinterpret_cast<const bf16x8 *>(pa + tt * 1024);
#pragma unroll
                        for (int tt = 0; tt < MT; ++tt)
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct)
                                acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][ct], xf[tt], acc[tt][ct], 0, 0, 0);
                        ++cc;
                    }
            };
            int q = q0;
#pragma unroll 1
            for (; q + 2 * SPT1 < q1; q += 2 * SPT1) {
                trip_s(ws[0], q);
                load_ws(ws[0]);
                trip_s(ws[1], q + SPT1);
                load_ws(ws[1]);
            }
            trip_s(ws[0], q);
            trip_s(ws[1], q + SPT1);
            GG_BOX_WAIT_BARRIER(0);
        }
    }

    // ---- combine the 8 waves (fixed order), then bias / residual / store.  red[wave][tt][ct][lane] is lane-contiguous:
    //      conflict-free 1 KiB wave writes and reads.
    // the thread's residual values of the final pass are requested now, a barrier and the 8-wave combine ahead of their use
    constexpr int EPI = (MT * CT * 64 + NTH - 1) / NTH;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    u32x2 resv[EPI];
#pragma unroll
    for (int kk = 0; kk < EPI; ++kk) {
        const int i = tid + NTH * kk;
        resv[kk] = u32x2{0u, 0u};
        const int oh = h0 + ((i >> 6) / CT) * RPT + (i & 15) / TWI;
        if (p.residual && i < MT * CT * 64 && oh < p.Ho && oact)
            resv[kk] = *reinterpret_cast<const u32x2 *>(p.residual + (((long long)n * p.Ho + oh) * p.Wo + (w0 + (i & 15) % TWI)) * p.Cout_pad + co_thr);
    }
    f32x4 *red = reinterpret_cast<f32x4 *>(box);
    // MT * CT > 16 (12 position tiles x 2 cout tiles: the 320 -> 320 upsample conv to 64x64 on 240 instead of 480 workgroups): eight slabs
    // would not fit the LDS, so waves 4-7 ADD their accumulators into the slabs of waves 0-3 in a second phase (fixed order)
    constexpr bool TWO_PHASE = NW == 8 && MT * CT > 16;
    constexpr int NWR = TWO_PHASE ? 4 : NW;
    if constexpr (!TWO_PHASE) {
#pragma unroll
        for (int tt = 0; tt < MT; ++tt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) red[((wave * MT + tt) * CT + ct) * 64 + lane] = acc[tt][ct];
        GG_BOX_LDS_BARRIER();
    } else {
        if (wave < 4) {
#pragma unroll
            for (int tt = 0; tt < MT; ++tt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) red[((wave * MT + tt) * CT + ct) * 64 + lane] = acc[tt][ct];
        }
        GG_BOX_LDS_BARRIER();
        if (wave >= 4) {
#pragma unroll
            for (int tt = 0; tt < MT; ++tt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) red[(((wave - 4) * MT + tt) * CT + ct) * 64 + lane] += acc[tt][ct];
        }
        GG_BOX_LDS_BARRIER();
    }
    GG_STAMP(6);
    const bool stats = p.gn_acc && p.out_dtype != GG_F32;       // GroupNorm statistics of the NEXT norm (see gg_conv_desc.gn_acc)
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < EPI; ++kk) {
        const int i = tid + NTH * kk;
        if (i >= MT * CT * 64) break;
        if (!oact) continue;                                // cout sub-split: another workgroup's couts
        f32x4 a = red[i];
#pragma unroll
        for (int w = 1; w < NWR; ++w) a += red[w * MT * CT * 64 + i];
        const int l = i & 63, tt = (i >> 6) / CT;
        a += bias4;
        const int oh = h0 + tt * RPT + (l & 15) / TWI;
        if (oh >= p.Ho) continue;                           // ragged last row tile (Ho not a multiple of the tile height)
        const long long mo = ((long long)n * p.Ho + oh) * p.Wo + (w0 + (l & 15) % TWI);
        const long long o = mo * p.Cout_pad + co_thr;
        if (p.residual) {
            const bf16x4 r = __builtin_bit_cast(bf16x4, resv[kk]);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += (float)r[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (co_thr + j >= p.Cout) a[j] = 0.f;
        if (p.out_dtype == GG_F32) {
            // (the DDIM state and scalars are requested BEFORE the eps store: loads and stores share one in-order counter on gfx950, a
            //  load behind the store would wait for the store's round trip)
            const bool dd = p.ddim_x && co_thr == 0;
            f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f}, dsc = f32x4{1.f, 1.f, 0.f, 0.f};
            if (dd) {
                xv = *reinterpret_cast<const f32x4 *>(p.ddim_x + mo * 4);
                dsc = *reinterpret_cast<const f32x4 *>(p.ddim_scalars);
            }
            *reinterpret_cast<f32x4 *>((float *)p.out + o) = a;
            if (dd) {
                // fused DDIM update (ddim.py:190-204), the UNet head conv's eps still in registers; same fp32 expression order as
                // ddim_step_kernel (bit-identical results)
#pragma clang fp contract(off)
                const float a_t = dsc[0], a_prev = dsc[1], sigma = dsc[2], s1m = dsc[3];
                const float sqrt_at = sqrtf(a_t), sqrt_ap = sqrtf(a_prev), dirc = sqrtf(1.0f - a_prev - sigma * sigma);
                f32x4 px0, xn;
                bf16x4 xb;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    px0[j] = (xv[j] - s1m * a[j]) / sqrt_at;
                    xn[j] = sqrt_ap * px0[j] + dirc * a[j];
                    xb[j] = (bf16_t)xn[j];
                }
                *reinterpret_cast<f32x4 *>(p.ddim_x + mo * 4) = xn;
                if (p.ddim_pred_x0) *reinterpret_cast<f32x4 *>(p.ddim_pred_x0 + mo * 4) = px0;
                if (p.ddim_unet_in) *reinterpret_cast<bf16x4 *>(p.ddim_unet_in + mo * p.ddim_unet_in_stride) = xb;
            }
        } else {
            bf16x4 ob;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ob[j] = (bf16_t)a[j];
                const float f = (float)ob[j];               // what the next norm will read
                ssum[j] += f;
                ssq[j] += f * f;
            }
            *reinterpret_cast<bf16x4 *>((bf16_t *)p.out + o) = ob;
        }
    }
    if (stats) {
        // a thread's slices (tid>>6) + 8k share their 4 couts; its lane's position (l & 15) is reduced over the 16 lanes of the row
        // by DPP moves, the 8 waves through LDS in a fixed order, then ONE wave instruction of 64-bit integer atomics per block
        __shared__ float statp[NW][16][2];                    // [wave][cq * 4 + j][sum | sumsq]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = gg_row16_sum(ssum[j]), b = gg_row16_sum(ssq[j]);
            if ((lane & 15) == 0) { statp[wave][(lane >> 4) * 4 + j][0] = a; statp[wave][(lane >> 4) * 4 + j][1] = b; }
        }
        __syncthreads();
        if (tid < 32 * CT) {
            const int which = tid & 1, c = tid >> 1;         // channel c of the block's 16*CT couts
            const int ct = c >> 4, cw = c & 15;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (w % CT == ct) t += statp[w][cw][which];  // waves whose slices carry cout tile ct, fixed order
            const long long fx = __double2ll_rn((double)t * (double)(which ? GG_ACC_SQ_SCALE : GG_ACC_SUM_SCALE));
            if (NS == 1 || cw / CSZ == sub)                  // cout sub-split: the sums of this workgroup's couts only
                atomicAdd(reinterpret_cast<unsigned long long *>(p.gn_acc + ((((long long)n * GG_ACC_STRIPES + stripe) * p.Cout_pad + g * 32 + half * 16 + c) * 2 + which)),
                          (unsigned long long)fx);
        }
    }
    GG_STAMP(7);
}

struct BoxPlan { int TWI, MT, CT, nstage, nch_stage, gn_bytes, q_major; long long smem; int nstage_s, nch_stage_s, NS; };

// Cost model: bytes one CU has to take in (its weight slice + its input box), times the number of rounds the grid needs on
// 256 CUs.  Smallest wins; ties go to the larger tile (fewer redundant halo bytes overall).
static bool plan_box(const ConvParams &p, BoxPlan &pl)
{
    constexpr long long max_blocks = 1024, lds_cap = 131072;     // <= 4 rounds of 256 workgroups; box + GroupNorm rows <= 128 KiB
    const bool k3 = p.kh == 3 && p.kw == 3 && p.pad == 1, k1 = p.kh == 1 && p.kw == 1 && p.pad == 0 && !p.upsample;
    const bool s2 = GG_BOX_STRIDE2 && p.stride == 2 && k3 && !p.upsample && p.skip_C1 == 0;      // the UNet's Downsample convs
    if (!(p.kd == 1 && p.D == 1 && (p.stride == 1 || s2) && (k3 || k1))) return false;
    const int halo = k3 ? 2 : 0;
    // 1x1: only where the grid is under-filled (measured: 8x8 4.2 vs 8.6 us on the tiny-M kernel, but 64x64 12.0 vs 7.8 us on gather5)
    constexpr long long k1_max_m = 256;
    // ... unless the conv folds a GroupNorm from accumulators itself: then the box kernel also replaces the norm's launch
    const long long k1_lim = (p.prologue_act && p.pro_clog > 0) ? GG_BOX_K1_MAX_M_ACC : k1_max_m;
    if (k1 && p.M > k1_lim) return false;
    const int TWI = p.Wo % 16 == 0 ? 16 : p.Wo % 8 == 0 ? 8 : p.Wo % 4 == 0 ? 4 : 0;   // width of a 16-position MFMA tile
    if (!TWI) return false;
    const int RPT = 16 / TWI;
    const long long wbytes16 = 16LL * (k3 ? 9 : 1) * p.nchunk * 32 * 2;   // weight slice of 16 output channels
    double best = 0;
    int bMT = 0, bCT = 0, bNS = 1;
    // tile heights: powers of two, plus 12 / 6 / 3 rows for 16-wide tiles so that 240 (not 160 or 320) workgroups cover the
    // 64 / 32 / 16-row levels; the last row tile may be ragged (rows >= Ho are computed on zero padding and not stored)
    for (int MT : {12, 8, 6, 4, 3, 2, 1}) {
        const int TH = MT * RPT;
        if ((p.upsample && (TH & 1)) || TH > p.Ho) continue;
        if (TWI != 16 && (MT == 12 || MT == 6 || MT == 3 || p.Ho % TH)) continue;
        if ((TWI == 16 && MT == 1) || (TWI == 8 && MT == 8) || (TWI == 4 && MT != 1)) continue;   // instantiated shapes only
        const int rows = p.upsample ? (TH / 2 + 2) * (TWI / 2 + 2) : s2 ? (2 * TH + 1) * (2 * TWI + 1) : (TH + halo) * (TWI + halo);
        const long long boxb = (long long)rows * p.nchunk * 64;
        for (int CT : {2, 1}) {
            const long long blocks = (long long)p.N * ((p.Ho + TH - 1) / TH) * (p.Wo / TWI) * (p.Cout_pad / (16 * CT));
            if (blocks > max_blocks || (GG_BOX_NW == 8 && MT * CT > 16 ? 4LL : (long long)(GG_BOX_NW < 8 ? GG_BOX_NW : 8)) * MT * CT * 1024 > lds_cap) continue;      // grid cap; the combine area (8 slabs, or 4 in two phases) must fit
            // every extra LDS stage is another exposed staging round trip
            const long long plane_c = (long long)((rows + 15) / 16) * 1024;
            const long long cap_c = lds_cap / plane_c > 0 ? lds_cap / plane_c : 1;
            const long long nst = (p.nchunk + cap_c - 1) / cap_c;
            const double cost = (double)(wbytes16 * CT + boxb) * (double)((blocks + 255) / 256) * (1.0 + 0.15 * (double)(nst - 1));
            if (!bMT || cost < best * 0.97) { best = cost; bMT = MT; bCT = CT; bNS = 1; }
            // cout sub-split (weight-bound 3x3 convs of the 8x8 / 4x4 levels; instantiated shapes only): NS workgroups per 16-cout tile,
            // each taking in 1 / NS of its weight rows and the whole box
            if (GG_BOX_COUT_SUBSPLIT && CT == 1 && k3 && !p.upsample && !s2 && ((TWI == 4 && MT == 1) || (TWI == 8 && MT == 2)))
                for (int NS : {2, 4}) {
                    if (blocks * NS > 256) continue;                       // one round only
                    const double cs = (double)(wbytes16 / NS + boxb) * (1.0 + 0.15 * (double)(nst - 1));
                    if (cs < best * 0.97) { best = cs; bMT = MT; bCT = CT; bNS = NS; }
                }
        }
    }
    if (!bMT) return false;                                          // filled grids: the halo / wide-tile kernels win
    const int MT = bMT, CT = bCT, NS = bNS, TH = MT * RPT;
    const int rows = p.upsample ? (TH / 2 + 2) * (TWI / 2 + 2) : s2 ? (2 * TH + 1) * (2 * TWI + 1) : (TH + halo) * (TWI + halo);
    const long long plane = (long long)((rows + 15) / 16) * 1024;   // whole 16-row DMA blocks
    long long cap = lds_cap / plane;
    if (cap < 1) return false;
    const int nstage = (int)((p.nchunk + cap - 1) / cap);
    if (TWI == 4 && nstage > 1) return false;                        // 4x4 levels with > 1 stage: the split-K tiny kernel fills more CUs
    const int nch_stage = (p.nchunk + nstage - 1) / nstage;
    long long smem = nch_stage * plane;
    const long long red = (GG_BOX_NW == 8 && MT * CT > 16 ? 4LL : (long long)GG_BOX_NW) * MT * CT * 64 * 16;      // [wave][tt][ct][lane] f32x4 (two-phase combine above 16 tiles)
    if (smem < red) smem = red;
    // scale / shift rows in front of the box: external tables are DMA'd per stage, the accumulator fold keeps all chunks of the conv
    const int gn_bytes = p.prologue_act ? ((p.pro_acc1 ? p.nchunk : nch_stage) * 32 * 8 + 1023) / 1024 * 1024 : 0;
    // XCD locality: a run of workgroups shares weights (cout-major) when the weights are the bigger re-fetch, else boxes
    const long long P = (long long)p.N * ((p.Ho + TH - 1) / TH) * (p.Wo / TWI), Q = p.Cout_pad / (16 * CT);
    const long long wtot = wbytes16 * (p.Cout_pad / 16), xtot = (long long)p.N * p.H * p.W * p.nchunk * 64;
    const long long cost_q = wtot + xtot * (Q < 8 ? Q : 8), cost_p = wtot * (P < 8 ? P : 8) + xtot;
    // K-concatenated 1x1 skip projection: stages of the x tile (MT KiB per chunk) in the same region
    int nstage_s = 0, nch_stage_s = 0;
    if (p.skip_C1 > 0) {
        if (!k3 || p.upsample || p.skip_C1 % 32 || p.skip_C2 % 32) return false;
        const long long plane1 = (long long)MT * 1024, nck = (p.skip_C1 + p.skip_C2) / 32;
        const long long cap1 = lds_cap / plane1;
        nstage_s = (int)((nck + cap1 - 1) / cap1);
        nch_stage_s = (int)((nck + nstage_s - 1) / nstage_s);
        if (smem < nch_stage_s * plane1) smem = nch_stage_s * plane1;
    }
    pl = {TWI, MT, CT, nstage, nch_stage, gn_bytes, cost_q <= cost_p ? 1 : 0, smem + gn_bytes, nstage_s, nch_stage_s, NS};
    return true;
}

template <int TWI, int MT, int CT, int UP, int K3, int SK = 0, int NS = 1>
static int launch_box(const ConvParams &p, const BoxPlan &pl, hipStream_t stream)
{
    // the attribute is per device: one bit per device ordinal (setting it twice from two threads is harmless)
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GG_ERR_HIP;
    const unsigned long long dev_bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & dev_bit)) {
        if (hipFuncSetAttribute((const void *)conv_box2d_kernel<TWI, MT, CT, UP, K3, SK, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
            return GG_ERR_UNSUPPORTED;
        attr_mask.fetch_or(dev_bit, std::memory_order_release);
    }
    const int tiles_h = (p.Ho + MT * (16 / TWI) - 1) / (MT * (16 / TWI)), tiles_w = p.Wo / TWI;
    dim3 grid((unsigned)(p.N * tiles_h * tiles_w * (p.Cout_pad / (16 * CT)) * NS));
    const int Pn = p.N * tiles_h * tiles_w, Qn = p.Cout_pad / (16 * CT) * NS;      // (cout sub-split: NS workgroups per cout tile)
    const int nch_last = p.nchunk - (pl.nstage - 1) * pl.nch_stage;
    const int nck_s = (p.skip_C1 + p.skip_C2) / 32, nch_s_last = pl.nstage_s ? nck_s - (pl.nstage_s - 1) * pl.nch_stage_s : 0;
    const BoxMagic mg = {gg_magic(pl.q_major ? Pn : Qn), gg_magic(tiles_w), gg_magic(tiles_h), gg_magic(pl.nch_stage), gg_magic(nch_last), Qn,
                         gg_magic(pl.nch_stage_s), gg_magic(nch_s_last), pl.nstage_s, pl.nch_stage_s};
    hipLaunchKernelGGL((conv_box2d_kernel<TWI, MT, CT, UP, K3, SK, NS>), grid, dim3(GG_BOX_NW * 64), (size_t)pl.smem, stream, p, tiles_h, tiles_w, pl.nstage,
                       pl.nch_stage, pl.gn_bytes, pl.q_major, (int)grid.x, mg);
    GG_CHECK_LAUNCH();
    return GG_OK;
}

template <int CT, int UP, int K3, int SK = 0>
static int dispatch_box(const ConvParams &p, const BoxPlan &pl, hipStream_t stream)
{
    switch (pl.TWI * 10 + pl.MT) {
        case 172: return launch_box<16, 12, CT, UP, K3, SK>(p, pl, stream);
        case 168: return launch_box<16, 8, CT, UP, K3, SK>(p, pl, stream);
        case 166: return launch_box<16, 6, CT, UP, K3, SK>(p, pl, stream);
        case 163: return launch_box<16, 3, CT, UP, K3, SK>(p, pl, stream);
        case 164: return launch_box<16, 4, CT, UP, K3, SK>(p, pl, stream);
        case 162: return launch_box<16, 2, CT, UP, K3, SK>(p, pl, stream);
        case 84: return launch_box<8, 4, CT, UP, K3, SK>(p, pl, stream);
        case 82:
            if constexpr (CT == 1 && UP == 0 && K3 == 1) {
                if (pl.NS == 2) return launch_box<8, 2, CT, UP, K3, SK, 2>(p, pl, stream);
                if (pl.NS == 4) return launch_box<8, 2, CT, UP, K3, SK, 4>(p, pl, stream);
            }
            return pl.NS == 1 ? launch_box<8, 2, CT, UP, K3, SK>(p, pl, stream) : GG_ERR_UNSUPPORTED;
        case 81: return launch_box<8, 1, CT, UP, K3, SK>(p, pl, stream);
        case 41:
            if constexpr (CT == 1 && UP == 0 && K3 == 1) {
                if (pl.NS == 2) return launch_box<4, 1, CT, UP, K3, SK, 2>(p, pl, stream);
                if (pl.NS == 4) return launch_box<4, 1, CT, UP, K3, SK, 4>(p, pl, stream);
            }
            return pl.NS == 1 ? launch_box<4, 1, CT, UP, K3, SK>(p, pl, stream) : GG_ERR_UNSUPPORTED;
        default: return GG_ERR_UNSUPPORTED;
    }
}

// The in-place prologue activates the whole box once per workgroup, i.e. once per cout tile: with Q cout tiles it is Q x 1.4
// times the work of a separate GroupNorm-apply launch, and transcendental-bound (2 per element).  Fusing pays only when few
// cout tiles share a box (measured: Q = 5 breaks even with the 3 us apply launch, Q = 20 costs +6 us).
bool gg_conv_box_fuses_prologue(const ConvParams &p)
{
    constexpr int fuse_q = 2;
    BoxPlan pl;
    if (!plan_box(p, pl)) return false;
    return p.Cout_pad / (16 * pl.CT) <= fuse_q;
}

bool gg_conv_box_emits_stats(const ConvParams &p) { return p.out_dtype != GG_F32; }

// Prologue computed from accumulators inside the conv (gg_conv_desc.pro_acc1).  The fold itself is ~0.5 us per workgroup; what decides
// is the in-place transform, redone by every cout tile that shares a box: an affine-only norm (attention / SpatialTransformer: no
// SiLU) is a handful of VALU per 16-byte piece and always pays against the ~4.5 us GroupNorm launch it removes; a SiLU norm is
// transcendental-bound (2 per element) and pays only where few cout tiles share a box (as with external tables) or the workgroup's box
// is small (GG_BOX_ACC_SILU_MAX_ELEMS elements; A/B in tools/experiments/README.md).
bool gg_conv_box_prologue_from_acc(const ConvParams &p)
{
    BoxPlan pl;
    if (!plan_box(p, pl) || p.C1 + p.C2 > 2048 || !p.prologue_act) return false;
    if (p.prologue_act == 2) return true;
    // SiLU norm: what a workgroup pays is the in-place pass over ITS box (rows x input channels, 2 transcendentals per element)
    const int RPT = 16 / pl.TWI, TH = pl.MT * RPT;
    const long long rows = p.upsample ? (long long)(TH / 2 + 2) * (pl.TWI / 2 + 2) : (long long)(TH + 2 * (p.kh == 3)) * (pl.TWI + 2 * (p.kw == 3));
    constexpr long long silu_max_elems = GG_BOX_ACC_SILU_MAX_ELEMS;
    return p.Cout_pad / (16 * pl.CT) <= 2 || rows * (p.C1 + p.C2) <= silu_max_elems;
}

// Returns GG_ERR_UNSUPPORTED (silently) when the shape is outside the envelope.  stream == (hipStream_t)-1: dry run.
int gg_conv_box_try(const ConvParams &p, hipStream_t stream)
{
    BoxPlan pl;
    if (!plan_box(p, pl)) return GG_ERR_UNSUPPORTED;
    if (stream == (hipStream_t)-1) return GG_OK;
    if (p.skip_C1 > 0) return pl.CT == 2 ? dispatch_box<2, 0, 1, 1>(p, pl, stream) : dispatch_box<1, 0, 1, 1>(p, pl, stream);     // (plan_box: 3x3, no upsample)
    if (p.kh == 1) return pl.CT == 2 ? dispatch_box<2, 0, 0>(p, pl, stream) : dispatch_box<1, 0, 0>(p, pl, stream);
    if (p.stride == 2) return pl.CT == 2 ? dispatch_box<2, 2, 1>(p, pl, stream) : dispatch_box<1, 2, 1>(p, pl, stream);
    if (pl.CT == 2) return p.upsample ? dispatch_box<2, 1, 1>(p, pl, stream) : dispatch_box<2, 0, 1>(p, pl, stream);
    return p.upsample ? dispatch_box<1, 1, 1>(p, pl, stream) : dispatch_box<1, 0, 1>(p, pl, stream);
}
