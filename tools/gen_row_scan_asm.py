#!/usr/bin/env python3
"""Generates open-msspe-design_amd/csrc/row_scan_pinned.inc: the predecessor scan of thal_pairs_row.hip's 13-base
instance as ONE inline-asm block over a slot table that lives in hand-assigned registers.

Why: the table (52 slots x {value G, word W} = 104 VGPRs) is read with compile-time register numbers and written
through a wave-uniform index.  As compiler-managed register tuples (v32i + v16i + v4i per plane) every cell paid
register copies the allocator could not coalesce: two per skipped chunk on the way out of the scan's nested exits, and
about 45 64-bit copies around the indexed write of every slot from 32 up.  The VALU of this kernel is 93 % busy at a
flat 4 cycles per instruction (profiles/r03_pmc_k_pairs_row.txt: SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles), so
instructions are time.  Here the tuples are bound to FIXED registers wherever they are touched -- the scan and the
publish are inline asm whose tuple operands carry explicit register ranges ("{v[62:93]}" ...), so the allocator keeps
them there for good and everything else of the kernel around them -- and the scan is what it is on paper -- far
visit: sub, lshr, [ds_read], add3, mov, med3, min_f64 -- with one exit and no copies; the publish is two moves for
any slot (GPR-index mode over the whole 52-register plane, tuple boundaries or not).

Register map (168 VGPRs = three waves per SIMD):
    v57           G2: runner-up of the running minimum            (the scan's output, where it stands)
    v58:59, v60:61  operand pairs (value : word) of v_min_f64     (scan-internal)
    v62 .. v113   G[0 .. 51]   slot values (+ kRowZero)           = Ga v[62:93], Gb v[94:109], Gc v[110:113]
    v114 .. v165  W[0 .. 51]   slot words                         = Wa v[114:145], Wb v[146:161], Wc v[162:165]
    v166:167      running minimum (value : word)                  (the scan's output, where it stands)
    everything else: the compiler's
usage: tools/gen_row_scan_asm.py > open-msspe-design_amd/csrc/row_scan_pinned.inc
"""
KC = 4


class Cfg:
    def __init__(self, name, ns, g2, pairs, g0, w0, acc, tuples, clamp):
        self.name, self.ns, self.g2, self.pairs, self.g0, self.w0, self.acc = name, ns, g2, pairs, g0, w0, acc
        self.tuples = tuples      # [(first register, length)] of Ga, Gb, (Gc), Wa, Wb, (Wc) in operand order
        self.clamp = clamp        # the table address is clamped onto the "not available" entry behind the table


# 13 bases: 768 threads, 168 VGPRs, 52 slots as v32i + v16i + v4i per plane; addresses need no clamp (kRowZero)
ROW13 = Cfg("ROW13", 52, 57, (58, 60), 62, 114, 166,
            [(62, 32), (94, 16), (110, 4), (114, 32), (146, 16), (162, 4)], False)
# 14 / 15 bases: 512 threads, 256 VGPRs, 64 slots as two v32i per plane; clamped addresses
#     v120 .. v183  G[0 .. 63], v184 .. v247  W[0 .. 63], v[248:249] running minimum, v[250:251] v[252:253] pairs, v254 G2
ROW64 = Cfg("ROW64", 64, 254, (250, 252), 120, 184, 248,
            [(120, 32), (152, 32), (184, 32), (216, 32)], True)
NS = G2 = PAIR = G0 = W0 = ACC = CLAMP = None


def use(cfg):
    global NS, G2, PAIR, G0, W0, ACC, CLAMP
    NS, G2, PAIR, G0, W0, ACC, CLAMP = cfg.ns, cfg.g2, cfg.pairs, cfg.g0, cfg.w0, cfg.acc, cfg.clamp


def scan_asm():
    L = []
    a = lambda s: L.append(s)
    # operands: %[C] %[Y] v; %[IDX] %[NFAR] %[NRUN] %[NEAR] s; outputs %[SG] %[SW] v, %[HV] s64, %[BG] %[BW] %[B2] v;
    # temps %[a0..a3] %[t0..t3] v (two sets of four: address -> loaded value in place); immediates %[TOFF] %[INIT]
    nch = NS // KC
    SETS = (["%[a0]", "%[a1]", "%[a2]", "%[a3]"], ["%[t0]", "%[t1]", "%[t2]", "%[t3]"])
    G = lambda pc, e: G0 + pc * KC + e
    W = lambda pc, e: W0 + pc * KC + e

    def tail(e):   # runner-up and minimum of visit e (its operand pair is ready)
        p = PAIR[e & 1]
        a(f"v_med3_i32 v{G2}, v{ACC + 1}, v{G2}, v{p + 1}")
        a(f"v_min_f64 v[{ACC}:{ACC + 1}], v[{ACC}:{ACC + 1}], v[{p}:{p + 1}]")

    def loads(pc, addr, data):
        for e in range(KC):
            a(f"v_sub_u32 {addr[e]}, %[C], v{W(pc, e)}")
        for e in range(KC):
            a(f"v_lshrrev_b32 {addr[e]}, 15, {addr[e]}")
        if CLAMP:   # a predecessor that is not up-left of the cell: onto the "not available" entry behind the table
            for e in range(KC):
                a(f"v_min_u32 {addr[e]}, %[TB], {addr[e]}")
        for e in range(KC):
            a(f"ds_read_b32 {data[e]}, {addr[e]} offset:%c[TOFF]")

    def far_process(pc, data, newer):   # newer: loads issued after this chunk's (the next chunk's prefetch)
        a(f"s_waitcnt lgkmcnt({newer})")   # (one wait for the four: they were issued a chunk ago and return together)
        for e in range(KC):
            p = PAIR[e & 1]
            a(f"v_add3_u32 v{p + 1}, {data[e]}, %[Y], v{G(pc, e)}")
            a(f"v_mov_b32 v{p}, v{W(pc, e)}")
            if e >= 1:
                tail(e - 1)
        tail(KC - 1)

    a(f"v_mov_b32 v{ACC + 1}, %[INIT]")
    a(f"v_mov_b32 v{ACC}, 0")
    a(f"v_bfrev_b32 v{G2}, -2")               # 0x7fffffff
    a("v_mov_b32 %[SG], 0")
    a("v_mov_b32 %[SW], 0")
    a("s_mov_b64 %[HV], 0")
    # ---- far chunks (slots of rows i-2 and above: every entry takes the cell-side term), software-pipelined: the next
    #      chunk's addresses and table reads are issued before this chunk's candidates are reduced; the loaded value
    #      lands in its address register (two sets of four, used alternately)
    a("s_cmp_lt_i32 %[NFAR], 1")
    a("s_cbranch_scc1 Lnear0_%=")
    loads(0, SETS[0], SETS[0])
    for pc in range(nch):
        cur = SETS[pc & 1]
        if pc + 1 < nch:
            a(f"s_cmp_gt_i32 %[NFAR], {pc + 1}")
            a(f"s_cbranch_scc0 Lfl{pc}_%=")
            loads(pc + 1, SETS[(pc + 1) & 1], SETS[(pc + 1) & 1])
            far_process(pc, cur, KC)
            a(f"s_branch Lff{pc + 1}_%=")
            a(f"Lfl{pc}_%=:")
        far_process(pc, cur, 0)
        if pc + 1 < nch:
            a(f"s_branch Lnear{pc + 1}_%=")
            a(f"Lff{pc + 1}_%=:")
    a("s_branch Ldone%=")
    # ---- the chunks behind the far ones: the one that straddles rows i-2 and i-1 (the first rem slots take the
    #      cell-side term) and the ones of row i-1 alone; the cell (i-1, j-1) is among these slots
    def stk(pc, e, addr):
        a(f"v_cmp_eq_u32_e32 vcc, %[IDX], {addr[e]}")
        a(f"v_cndmask_b32_e32 %[SG], %[SG], v{G(pc, e)}, vcc")
        a(f"v_cndmask_b32_e32 %[SW], %[SW], v{W(pc, e)}, vcc")
        a("s_or_b64 %[HV], %[HV], vcc")
    for pc in range(nch):
        addr, data = SETS
        a(f"Lnear{pc}_%=:")
        a(f"s_cmp_gt_i32 %[NRUN], {pc}")
        a("s_cbranch_scc0 Ldone%=")
        loads(pc, addr, data)
        a(f"s_cmp_gt_i32 %[NEAR], {pc * KC}")
        a(f"s_cbranch_scc0 Lnr{pc}_%=")
        for e in range(KC):   # straddle
            p = PAIR[e & 1]
            a(f"s_cmp_gt_i32 %[NEAR], {pc * KC + e}")
            a("s_cselect_b64 vcc, -1, 0")
            a("v_cndmask_b32_e32 %[yt], 0, %[Y], vcc")
            if e == 0:
                a("s_waitcnt lgkmcnt(0)")
            a(f"v_add3_u32 v{p + 1}, {data[e]}, %[yt], v{G(pc, e)}")
            a(f"v_mov_b32 v{p}, v{W(pc, e)}")
            if e >= 1:
                tail(e - 1)
            stk(pc, e, addr)
        tail(KC - 1)
        if pc + 1 < nch:
            a(f"s_branch Lnear{pc + 1}_%=")
        else:
            a("s_branch Ldone%=")
        a(f"Lnr{pc}_%=:")
        for e in range(KC):   # row i-1 only
            p = PAIR[e & 1]
            if e == 0:
                a("s_waitcnt lgkmcnt(0)")
            a(f"v_add_u32 v{p + 1}, {data[e]}, v{G(pc, e)}")
            a(f"v_mov_b32 v{p}, v{W(pc, e)}")
            if e >= 1:
                tail(e - 1)
            stk(pc, e, addr)
        tail(KC - 1)
    a("Ldone%=:")   # (the running minimum and its runner-up are the asm's outputs where they stand: MSSPE_*_ACC_OUT)
    return L


def clobbers():
    return [PAIR[0], PAIR[0] + 1, PAIR[1], PAIR[1] + 1]


def cstr(lines):
    return "\n".join('    "' + l + '\\n\\t"' for l in lines)


print("// GENERATED by tools/gen_row_scan_asm.py -- do not edit (see that file for the why and the register maps)")
for cfg in (ROW13, ROW64):
    use(cfg)
    n = cfg.name
    print(f"#define MSSPE_{n}_G0 {G0}")
    print(f"#define MSSPE_{n}_W0 {W0}")
    print(f"#define MSSPE_{n}_SCAN_ASM \\")
    print(" \\\n".join('    "' + l + '\\n\\t"' for l in scan_asm()))
    print(f"#define MSSPE_{n}_SCAN_CLOBBERS " + ", ".join(f'"v{r}"' for r in clobbers()) + ', "vcc", "scc", "m0", "memory"')
    print(f'#define MSSPE_{n}_ACC_OUT(hi, lo, g2) "=&{{v{ACC + 1}}}"(hi), "=&{{v{ACC}}}"(lo), "=&{{v{G2}}}"(g2)')
    names = ["Ga", "Gb", "Gc", "Wa", "Wb", "Wc"] if len(cfg.tuples) == 6 else ["Ga", "Gb", "Wa", "Wb"]
    for kind, pre in (("IN", ""), ("INOUT", "+")):
        print(f"#define MSSPE_{n}_TUPLES_{kind}(" + ", ".join(names) + ") " +
              ", ".join('"%s{v[%d:%d]}"(%s)' % (pre, r0, r0 + ln - 1, nm) for (r0, ln), nm in zip(cfg.tuples, names)))
