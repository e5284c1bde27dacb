#!/usr/bin/env python3
"""Generates jointimagegeneration_amd/csrc/gg_conv_halo3_asm.inc: the hand-scheduled tap phase of the team halo conv
(gg_conv_halo3.hip) as inline-assembly text for gfx950.

One T-phase = the 27 taps of one 32-channel chunk for ONE wave of the tapping team: 8 position tiles x 4 cout tiles of
v_mfma_f32_16x16x32_bf16 per tap (A = weights, B = activations, as gg_conv_halo.hip), 864 MFMAs in all.  What hipcc could not hold
at this register budget (tools/experiments/README.md, "hand-ordered software pipelines") is written out here:

  * two operand register sets (8 activation + 4 weight fragments each): while the MFMAs of tap s run on one set, the 12
    ds_read_b128 of tap s+1 are issued into the other, one read per two MFMAs, so that a wave ALONE on its SIMD keeps the matrix
    pipe busy (the other wave of the SIMD belongs to the staging team and issues VALU / memory instructions in the MFMA shadow);
  * every LDS address is a lane constant + immediate (box rows: line * 1152 + tile * 1152; weight rows: slot * 12288 + tap * 4096
    + cout tile * 1024), so the loop has no address arithmetic at all;
  * one workgroup barrier per (kd, kh) line, placed in front of the line's LAST tap: by then every wave holds that tap's weights in
    registers, so the line's weight slot is free and the DMA of line + 2 is issued right behind the barrier, and the first tap of
    the next line is prefetched under the last tap's MFMAs.

Operands (see GG_H3_ASM_OPERANDS): c{tile*4+ct} accumulators; xa*/wa*, xb*/wb* the two fragment sets; ax0..2 box lane bases per
kw; aw weight lane base; vo0..2 the wave's DMA source offsets (one per tap of a line); ws (s) DMA offset step per line; dn (s)
offset jump from "line 9 of this phase" to line 0 of the next phase; m0b (s) LDS address of the wave's 1 KiB DMA piece in slot 0 /
tap 0; wbase (s, 64-bit) packed-weight base.
"""
import os

HW_ROW = 18 * 64           # bytes of one W-line of the box image (18 positions x 64 B)
PLANE = 10 * HW_ROW        # one D-plane (10 lines)
WTAP = 4096                # one tap's weight tile: 64 cout rows x 64 B
WSLOT = 3 * WTAP           # one (kd, kh) line = 3 taps
TPW, CT = 8, 4


def lineoff(L):
    kd, kh = divmod(L, 3)
    return kd * PLANE + kh * HW_ROW


def reads(dst, ax, lo, wimm):
    """the 12 operand reads of one tap into set `dst` ('a' | 'b'), in the order the next tap consumes them"""
    r = [f"ds_read_b128 %[w{dst}0], %[aw] offset:{wimm}"]
    r += [f"ds_read_b128 %[x{dst}{t}], %[{ax}] offset:{lo + t * HW_ROW}" for t in range(TPW)]
    r += [f"ds_read_b128 %[w{dst}{c}], %[aw] offset:{wimm + c * 1024}" for c in range(1, CT)]
    return r


def tap32(cur, nxt_reads):
    """TIMING ABLATION ONLY (numerically meaningless): the tap's 32 v_mfma 16x16x32 replaced by 16 v_mfma_f32_32x32x16_bf16 on the same
    fragment registers and 16-register accumulator tuples -- same FLOPs, same reads, half the MFMA issue slots"""
    out = []
    q = list(nxt_reads)
    i = 0
    for ct in range(CT):
        for tt in range(0, TPW, 2):
            out.append(f"v_mfma_f32_32x32x16_bf16 %[d{(tt // 2) * 2 + (ct & 1)}], %[w{cur}{ct}], %[x{cur}{tt + (ct >> 1)}], %[d{(tt // 2) * 2 + (ct & 1)}]")
            i += 1
            if q:
                out.append(q.pop(0))
    while q:
        out.append(q.pop(0))
    return out


def tap(cur, nxt_reads):
    """32 MFMAs on set `cur`, with `nxt_reads` (<= 12) interleaved one per two MFMAs"""
    out = []
    q = list(nxt_reads)
    i = 0
    for ct in range(CT):
        for tt in range(TPW):
            out.append(f"v_mfma_f32_16x16x32_bf16 %[c{tt * CT + ct}], %[w{cur}{ct}], %[x{cur}{tt}], %[c{tt * CT + ct}]")
            i += 1
            if q and i % 2 == 0:
                out.append(q.pop(0))
    assert not q
    return out


def tphase(P, tap=None):
    tap = tap or globals()["tap"]
    s = ["s_setprio TPRIO"]
    sets = "ab"
    s += reads(sets[P], "ax0", lineoff(0), P * WSLOT)
    for L in range(9):
        par = (P + L) & 1
        S, T = sets[par], sets[1 - par]
        slot = par * WSLOT
        lo = lineoff(L)
        s.append("s_waitcnt lgkmcnt(0)")
        s += tap(S, reads(T, "ax1", lo, slot + WTAP))
        s.append("s_waitcnt lgkmcnt(0)")
        s += tap(T, reads(S, "ax2", lo, slot + 2 * WTAP))
        s.append("s_waitcnt lgkmcnt(0)")          # tap 2's operands are in registers: nobody reads this line's weight slot any more
        s.append("s_waitcnt vmcnt(0)")            # my pieces of line L+1's weights have landed
        s.append("s_barrier")
        if L == 7:                                # the two last DMA rounds of a phase fetch lines 0 and 1 of the NEXT phase
            s += [f"v_add_u32 %[vo{u}], %[vo{u}], %[dn]" for u in range(3)]
        for u in range(3):                        # line L+2 into the slot this line just released
            s += [f"s_add_i32 m0, %[m0b], {slot + u * WTAP}", "s_nop 0", f"global_load_lds_dwordx4 %[vo{u}], %[wbase]"]
        s += [f"v_add_u32 %[vo{u}], %[vo{u}], %[ws]" for u in range(3)]
        if L < 8:
            s += tap(S, reads(T, "ax0", lineoff(L + 1), (1 - par) * WSLOT))
        else:
            s += tap(S, [])
    s.append("s_waitcnt vmcnt(0)")                # lines 0 / 1 of the next phase have landed before the phase-end barrier
    s.append("s_setprio 0")
    return s


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "jointimagegeneration_amd", "csrc", "gg_conv_halo3_asm.inc")
    L = ["// GENERATED by tools/gen_halo3_asm.py -- do not edit; the schedule is documented there.", "#pragma once"]
    for P in (0, 1):
        body = tphase(P)
        L.append(f"#define GG_H3_TPHASE_P{P} \\")
        for ins in body:
            if ins == "s_setprio TPRIO":
                L.append('    "s_setprio " GG_H3_TPRIO "\\n" \\')
            elif ins.startswith("v_mfma"):
                L.append(f'    GG_H3_M("{ins}\\n") \\')          # (timing ablations drop the MFMAs: -DGG_H3_ABL_NOMFMA)
            else:
                L.append(f'    "{ins}\\n" \\')
        L.append('    ""')
    for P in (0, 1):
        body = tphase(P, tap32)
        L.append(f"#define GG_H3_TPHASE32_P{P} \\")
        for ins in body:
            if ins == "s_setprio TPRIO":
                L.append('    "s_setprio " GG_H3_TPRIO "\\n" \\')
            else:
                L.append(f'    "{ins}\\n" \\')
        L.append('    ""')
    outs = [f'[c{i}] "+v"(acc[{i // CT}][{i % CT}])' for i in range(TPW * CT)]
    for st in "ab":
        outs += [f'[x{st}{t}] "=&v"(x{st}[{t}])' for t in range(TPW)]
        outs += [f'[w{st}{c}] "=&v"(w{st}[{c}])' for c in range(CT)]
    outs += [f'[vo{u}] "+v"(vo[{u}])' for u in range(3)]
    ins = ['[ax0] "v"(ax[0])', '[ax1] "v"(ax[1])', '[ax2] "v"(ax[2])', '[aw] "v"(aw)', '[ws] "s"(ws)', '[dn] "s"(dn)', '[m0b] "s"(m0b)',
           '[wbase] "s"(wbase)', '[team] "s"(team)']
    L.append("#define GG_H3_ASM_OPERANDS \\")
    L.append("    : " + ", ".join(outs) + " \\")
    L.append("    : " + ", ".join(ins) + " \\")
    L.append('    : "memory", "m0", "scc"')
    outs32 = [f'[d{i}] "+v"(acc16[{i}])' for i in range(8)] + outs[TPW * CT:]
    L.append("#define GG_H3_ASM_OPERANDS32 \\")
    L.append("    : " + ", ".join(outs32) + " \\")
    L.append("    : " + ", ".join(ins) + " \\")
    L.append('    : "memory", "m0", "scc"')
    with open(out, "w") as f:
        f.write("\n".join(L) + "\n")
    print("wrote", os.path.normpath(out), sum(1 for x in tphase(0) if x.startswith("v_mfma")), "MFMAs per phase")


if __name__ == "__main__":
    main()
