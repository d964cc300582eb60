// thal_pairs_int.hip -- all-pairs cross-dimer kernel, exact-integer DP (thal ANY per ordered pair).
//
// Same job and same outputs as thal_pairs.hip (the reference's "format N^2 lines -> ntthal ->
// parse" loop, /root/reference/od-msspe/src/delta_g.rs:61-153; Primer3 2.6.1 thal() restated from
// SURVEY.md Appendix C.3), but the O(cells^2) part -- every earlier cell tried as the predecessor
// of every cell -- runs on int32 instead of f64:
//   * a cell's value is the exact triple (h, s, n) of fast_tables.hpp (IntTables); its dG at
//     37 C is the int32 G = 20000 h + 600000 n - 6203 s, and (G, h) is all a slot keeps: the
//     total entropy is (20000 h - G) / 620300.  fillMatrix's acceptance test
//     "dG(candidate) < dG(current)" is decided on G: one LDS gather T[d][po] + one add3 + one
//     compare per predecessor, where d = 16 l1 + l2 falls out of a single subtraction of packed
//     coordinates.  Exact values that differ are >= 5e-4 cal/mol apart, so the double comparison
//     Primer3 makes gives the same answer; only exact ties can go either way in doubles.
//   * maxTM's "extend the helix if Tm rises" compares two quotients; it is evaluated in f64 from
//     the integer state (cross-multiplied, no division) and is decisive unless the two sides
//     agree to 1e-9.
//   * a pair that meets an exact tie that matters (terminal pick; a cell of the optimal path), a
//     near-tie of Tm, a rejected (H > 0, S > 0) winner or more cells than the table holds is
//     NOT answered here: it goes to the hand-over list and is finished by the f64 kernels.
//     Nothing is approximated.
//   * each cell records its predecessor; the optimal path is walked by pointer, written to an LDS
//     scratch [step][thread] and then REPLAYED forwards in f64 with Primer3's own operation order
//     (pair_core.hpp cand_* = the f64 kernel's formulas), so dS, dH, dG and t carry the same bits
//     as the CPU oracle.
// CDNA4 mapping: lane = ordered pair, persistent blocks (one per CU): the slots x {G, W} live in
// register tuples (static reads, one indexed write per cell); LDS holds the loop table (48 / 61
// KB), the compact f64 / int tables, the predecessor bytes [slot][thread] and the path scratch.
//   k_pairs_int       matrix mode: wave = one row x 64 composition-sorted columns.  k <= 13: 768
//                     threads, 48 slots, 168 VGPRs = three waves per SIMD; else 512 threads, 56
//                     slots.  Lanes whose table is too large, or far above their wave's, are
//                     handed on (lock-step work ~ slots^2)
//   k_pairs_int_list  list mode: the handed-on pairs without a "needs f64" mark, 64 slots, batches
//                     counting-sorted by table size in LDS so that a wave's lanes are alike
#include "int_core.hpp"

namespace msspe {

namespace {


template <int NS, int THREADS, int TROWS>
struct SharedI {
    static constexpr int kThreads = THREADS, kRows = TROWS;
    int T[TROWS * 64];
    Lds F;                              // f64 S + int H (replay, end terms)
    int g[FastTables::kCount];
    double cq[100];                     // 620300 * (init_S + rS + RC) per right-end context (maxTM)
    unsigned char pred[NS][THREADS];
    unsigned short path[kPathMax][THREADS];
};

struct ICell {
    int cgeo;      // (im1 - 1) * 16 + (jm1 - 1)
    int cstk;      // packed coordinates of (i-1, j-1), or a value no slot holds (first row / column)
    int jm1p;      // jm1 - 1
    int a16;       // cell base << 4 (bulge column, pre-multiplied by 4 like po)
    int yTS, yMM;  // cell-side mismatch term of interior / 1x1 loops (G units)
};


// the f64 kernel's word (h << 14 | po << 8 | im1 << 4 | jm1) of pair_core.hpp's cand_* helpers
__device__ __forceinline__ int core_word(int W)
{
    return ((W >> 16) << 14) | ((W >> 2) & 0x3f00) | (W & 0xff);
}

// One predecessor slot against cell c.
struct Visit {
    int idx4;    // byte offset into T (0 = the kBig row when the geometry is invalid)
    int y;       // cell-side term for this kind of loop
    bool geo;    // predecessor lies strictly up-left of the cell, or is the cell (i-1, j-1)
    bool stack;  // ... the latter
};


__device__ __forceinline__ Visit visit_geometry(const ICell &c, int Wp)
{
    Visit v;
    const int d = c.cgeo - (Wp & 0xff);
    const int jj = Wp & 15;
    v.geo = (jj <= c.jm1p) & (d >= 0);
    v.stack = (Wp & 0xff) == c.cstk;         // the cell (i-1, j-1)
    const int po4 = (Wp >> 8) & 0xff;        // po << 2 (bits 8, 9 of W are zero)
    const bool bulge = (d < 16) | (jj == c.jm1p);   // l1 == 0 or l2 == 0 (for a valid geometry)
    const int pe4 = bulge ? ((po4 & 12) | c.a16) : po4;
    // a valid geometry has 0 <= d <= 238 and pe4 < 256: in range.  Everything else reads T[0], the
    // stacked-pair row, which holds kBig: no clamp and no separate validity mask in the compares.
    v.idx4 = v.geo ? ((d << 8) | pe4) : 0;
    v.y = d == 0x11 ? c.yMM : (bulge ? 0 : c.yTS);
    return v;
}

// The same for a slot that is known to lie at least three rows above the cell (l1 >= 2): no
// stacked pair, no 1 x 1 loop, a bulge only through l2 = 0, and no row test.
__device__ __forceinline__ Visit visit_geometry_far(const ICell &c, int Wp)
{
    Visit v;
    const int d = c.cgeo - (Wp & 0xff);
    const int jj = Wp & 15;
    v.geo = jj <= c.jm1p;
    v.stack = false;
    const int po4 = (Wp >> 8) & 0xff;
    const bool bulge = jj == c.jm1p;   // l2 == 0
    const int pe4 = bulge ? ((po4 & 12) | c.a16) : po4;
    v.idx4 = v.geo ? ((d << 8) | pe4) : 0;
    v.y = bulge ? 0 : c.yTS;
    return v;
}

__device__ __forceinline__ void visit_finish_far(const Visit &v, int t, int Gp, int Wp, RBest &best,
                                                 ScanMasks &m)
{
    const int cand = t + v.y + Gp;
    const bool better = cand < best.G;
    best.G2 = med3_i32(best.G, best.G2, cand);   // second smallest so far: a tie shows as G2 == G at the end
    best.G = min(cand, best.G);
    best.W = better ? Wp : best.W;
}

__device__ __forceinline__ void visit_finish(const Visit &v, int t, int Gp, int Wp, RBest &best,
                                             IBest &stk, ScanMasks &m)
{
    const int cand = t + v.y + Gp;   // unavailable / invalid: kBig + ..., never below best.G <= kValid
    const bool better = cand < best.G;
    best.G2 = med3_i32(best.G, best.G2, cand);
    best.G = min(cand, best.G);
    best.W = better ? Wp : best.W;
    stk.G = v.stack ? Gp : stk.G;
    stk.W = v.stack ? Wp : stk.W;
    m.stHave |= __builtin_amdgcn_ballot_w64(v.stack);
}

// All earlier slots as predecessors of cell c, kC at a time.  The chunks are unrolled with
// compile-time register numbers and left through a wave-uniform branch at the first chunk that
// holds no computed slot (later slots are empty and would fail the geometry test anyway).
template <int NS, int PC = 0>
__device__ __forceinline__ void scan_fill_int(MSSPE_TAB_PARAMS, int upto, int far_upto, const char *T,
                                              const ICell &c, RBest &best, IBest &stk, ScanMasks &m)
{
    if constexpr (PC * kC < NS) {
        if (PC * kC < upto) {   // wave-uniform
            Visit v[kC];
            int t[kC];
            if (PC * kC + kC <= far_upto) {   // wave-uniform: every lane has these slots >= 3 rows up
                asm volatile("" ::"n"(PC));   // keeps the chunks from being merged into selects
#pragma unroll
                for (int e = 0; e < kC; ++e) v[e] = visit_geometry_far(c, slot_of<NS>(Wa, Wb, Wc, PC * kC + e));
#pragma unroll
                for (int e = 0; e < kC; ++e) t[e] = *(const int *)(T + v[e].idx4);
#pragma unroll
                for (int e = 0; e < kC; ++e)
                    visit_finish_far(v[e], t[e], slot_of<NS>(Ga, Gb, Gc, PC * kC + e), slot_of<NS>(Wa, Wb, Wc, PC * kC + e),
                                     best, m);
            } else {
                asm volatile("" ::"n"(PC + 64));
#pragma unroll
                for (int e = 0; e < kC; ++e) v[e] = visit_geometry(c, slot_of<NS>(Wa, Wb, Wc, PC * kC + e));
#pragma unroll
                for (int e = 0; e < kC; ++e) t[e] = *(const int *)(T + v[e].idx4);
#pragma unroll
                for (int e = 0; e < kC; ++e)
                    visit_finish(v[e], t[e], slot_of<NS>(Ga, Gb, Gc, PC * kC + e), slot_of<NS>(Wa, Wb, Wc, PC * kC + e), best,
                                 stk, m);
            }
            scan_fill_int<NS, PC + 1>(MSSPE_TAB_ARGS, upto, far_upto, T, c, best, stk, m);
        }
    }
}


// thal ANY for the lane's pair.  n_cells == 0: idle lane.  n_cells counts every complementary cell; the
// lane's last-row cells (the last ones of its row-major order) may lie beyond slot NS - 1: they are computed
// and may be picked, but only cells of earlier rows are ever read back (wave_pairs sizes the table by them).
// RESOLVE: a terminal pick shared by exactly two cells is settled the way Primer3 settles it, by
// replaying both paths and comparing the two doubles (list mode; in matrix mode such pairs are
// handed on, because a second walk would be paid by the whole wave) -- unless the call asks for
// decisions only and no tied structure can reach the cut (int_core.hpp kPickMargin).
template <int NS, bool RESOLVE, class SH>
__device__ __forceinline__ IntResult run_pair_int(SH &sh, const ThalConsts &K, const SeqPair &q,
                                                  unsigned rowmask, int n_cells, int nmax, bool decisions_only)
{
    const Lds &F = sh.F;
    v32i Ga = 0, Wa = kEmptyW;
    typename TabTypes<NS>::B Gb = 0, Wb = kEmptyW;
    typename TabTypes<NS>::C Gc = 0, Wc = kEmptyW;
    int defer = 0;
    CellCtx c;
    c.rS = 0.0;
    c.rH = 0;
    c.im1p = c.jm1p = 0;
    c.yTS = c.yMM = c.bBase = 0;
    unsigned Rrem = rowmask, mrem = 0;
    int im1 = 0, jm1 = 0;
    // first slot of the lane's current row and of its two previous non-empty rows: every slot
    // below row_lo2 lies at least three rows above the current cell
    int row_lo0 = 0, row_lo1 = 0, row_lo2 = 0;
    int pickG = 0x7fffffff, pickW = 0, pickW2 = 0, nTie = 0, tieLo = 0, tieHi = 0;
    int pickP = 0xff, pickP2 = 0xff;   // the picked cells' predecessor byte | "has an equal-valued alternative" << 8
    unsigned long long softTie = 0ull;   // per lane: slots whose value has an equal-valued alternative

    for (int slot_ = 0; slot_ < nmax; ++slot_) {
        const int slot = __builtin_amdgcn_readfirstlane(slot_);
        // ---- next complementary cell in row-major order (as in thal_pairs.hip)
        const bool newrow = mrem == 0;
        const int t = __ffs((int)Rrem) - 1;
        const int a_new = (q.s1 >> (t & 31)) & 3;
        const unsigned m_new = spaced_mask(q.s2, 3 - a_new, q.lenmask);
        im1 = newrow ? (t >> 1) : im1;
        row_lo2 = newrow ? row_lo1 : row_lo2;
        row_lo1 = newrow ? row_lo0 : row_lo1;
        row_lo0 = newrow ? slot : row_lo0;
        Rrem = newrow ? (Rrem & (Rrem - 1)) : Rrem;
        mrem = newrow ? m_new : mrem;
        jm1 = (__ffs((int)mrem) - 1) >> 1;
        mrem &= mrem - 1;
        im1 &= 15;
        jm1 &= 15;
        const CellBases b = cell_bases(q, im1, jm1, c);
        ICell ic;
        ic.cgeo = (im1 - 1) * 16 + (jm1 - 1);
        ic.jm1p = jm1 - 1;
        ic.cstk = ((im1 > 0) & (jm1 > 0)) ? ic.cgeo : 0x100;
        ic.a16 = b.a << 4;
        ic.yTS = sh.g[c.yTS];
        ic.yMM = sh.g[c.yMM];
        // ---- all earlier slots as predecessors
        RBest best;
        IBest stk;
        best.G = IntTables::kValid;
        best.G2 = 0x7fffffff;
        best.W = 0;
        stk.G = stk.W = 0;
        ScanMasks sm;
        sm.tie = sm.stHave = 0ull;
        const int far_upto = wave_min_64(slot < n_cells ? row_lo2 : 63);
        scan_fill_int<NS>(MSSPE_TAB_ARGS, slot, far_upto, (const char *)sh.T, ic, best, stk, sm);
        const bool tie = best.G2 == best.G;   // two loop candidates share the minimum
        const bool stHave = lane_bit(sm.stHave);
        // ---- thal.c maxTM(): helix extension if it raises Tm.  T = A / B with B < 0 on both
        //      sides, so T1 > T0 <=> A1 B0 > A0 B1; 620300 B = (2000 H - G) + cq (exact integers
        //      plus one constant: decisive unless the two sides agree to 1e-9).
        int H0 = F.H[b.idxL], G0 = sh.g[b.idxL], pred = 0xff, flags = 0, cell_soft = 0;
        if (stHave) {
            const int rH = F.H[b.idxR];
            const double cq = sh.cq[b.idxR - FastTables::kEndR];
            const int H1 = (stk.W >> 16) * 10 + F.H[b.wc];
            const int G1 = stk.G + sh.g[b.wc];
            const double A0 = (double)(H0 + 200 + rH), A1 = (double)(H1 + 200 + rH);
            const double B0 = (double)(2000 * H0 - G0) + cq, B1 = (double)(2000 * H1 - G1) + cq;
            const double lhs = A1 * B0, rhs = A0 * B1;
            const bool sure = (B0 < 0.0) & (B1 < 0.0) & (fabs(lhs - rhs) > 1e-9 * (fabs(lhs) + fabs(rhs)));
            flags |= sure ? 0 : kDeferTm;
            if (lhs > rhs) {
                H0 = H1;
                G0 = G1;
                pred = stk.W & 0xff;
            }
        }
        // ---- loops (thal.c calc_bulge_internal acceptance: dG of the candidate strictly lower)
        if (best.G <= G0) {
            // exact enthalpy of the best candidate from the compact tables
            const CandGeom g = cand_geometry(c, core_word(best.W));
            const int Hw = F.H[g.lx] + F.H[g.y] + (best.W >> 16) * 10;
            if (best.G < G0) {
                // two candidates tie for the minimum: doubles could order them either way
                flags |= tie ? kDeferLoopTie : 0;
                // thal.c rejects a candidate with H > 0 and S > 0 (620300 S = 2000 H - G): never
                // the case for a sensible minimum; if it is, leave the pair to the f64 kernel
                flags |= ((Hw > 0) & (2000 * Hw - best.G > -1000)) ? kDeferBad : 0;
                H0 = Hw;
                G0 = best.G;
                pred = best.W & 0xff;
            } else if (Hw == H0) {
                // same value either way: only the path (and the rounding along it) could differ
                cell_soft = 0x100;
                softTie |= ((slot < n_cells) & (slot < 64)) ? (1ull << (slot & 63)) : 0ull;
            } else {
                flags |= kDeferLoopEq;
            }
        }
        const int Wcell = ((H0 / 10) << 16) | (b.po_c << 10) | (im1 << 4) | jm1;
        const bool in = slot < n_cells;   // lanes past their last cell compute garbage
        defer |= in ? flags : 0;
        // ---- terminal pick (thal.c thal(): strict minimum of dG incl. the right end term, first
        //      in row-major order = slot order; the 1e-6 nudges are the same on every cell and
        //      drop out)
        {
            const int Gt = G0 + sh.g[b.idxR];
            const bool pick = in & (Gt < pickG);
            const bool same = in & (Gt == pickG);
            pickP2 = (same & (nTie == 0)) ? (pred | cell_soft) : pickP2;
            pickW2 = (same & (nTie == 0)) ? Wcell : pickW2;   // the first later cell with the same value
            {   // range of the tied later cells' enthalpies (right end included, units of 10 cal/mol)
                const int ht = (H0 + F.H[b.idxR]) / 10;   // both multiples of 10 (build_int_tables)
                tieLo = same ? ((nTie == 0) ? ht : min(tieLo, ht)) : tieLo;
                tieHi = same ? ((nTie == 0) ? ht : max(tieHi, ht)) : tieHi;
            }
            nTie = pick ? 0 : (nTie + (same ? 1 : 0));
            pickG = pick ? Gt : pickG;
            pickW = pick ? Wcell : pickW;
            pickP = pick ? (pred | cell_soft) : pickP;
        }
        // ---- publish the cell (idle lanes write a slot nobody reads).  From slot NS - 1 on every lane still
        //      at work is in its last row (wave_pairs), whose cells are nobody's predecessors: they all land in
        //      slot NS - 1, which nothing reads
        const int ps = min(slot, NS - 1);   // wave-uniform; slot NS - 1 is kept free for them (wave_pairs)
        if constexpr (NS == 64) {
            // list mode: the four tuples bound to fixed registers where they are written (explicit register ranges on
            // the asm's operands, as in thal_pairs_row.hip): ONE indexed write per plane over the whole 64-register plane.
            // Left to the allocator, a write into the upper tuples copied both of them to fresh registers and back
            // (some sixty 64-bit moves per cell from slot 32 on -- half the cells of the tables this stage gets).
            asm volatile("s_set_gpr_idx_on %[SLOT], gpr_idx(DST)\n\t"
                         "v_mov_b32 v120, %[G]\n\t"
                         "v_mov_b32 v184, %[W]\n\t"
                         "s_set_gpr_idx_off"
                         : "+{v[120:151]}"(Ga), "+{v[152:183]}"(Gb), "+{v[184:215]}"(Wa), "+{v[216:247]}"(Wb)
                         : [G] "v"(G0), [W] "v"(Wcell), [SLOT] "s"(ps)
                         : "m0");
        } else
        if (ps < 32) {   // wave-uniform slot number: one indexed register write per plane
            Ga[ps & 31] = G0;
            Wa[ps & 31] = Wcell;
        } else if constexpr (NS == 56) {
            if (ps < 48) {
                Gb[(ps - 32) & 15] = G0;
                Wb[(ps - 32) & 15] = Wcell;
            } else {
                Gc[(ps - 48) & 7] = G0;
                Wc[(ps - 48) & 7] = Wcell;
            }
        } else if constexpr (NS == 48) {
            Gb[(ps - 32) & 15] = G0;
            Wb[(ps - 32) & 15] = Wcell;
        } else if constexpr (NS == 52) {
            if (ps < 48) {
                Gb[(ps - 32) & 15] = G0;
                Wb[(ps - 32) & 15] = Wcell;
            } else {
                Gc[(ps - 48) & 3] = G0;
                Wc[(ps - 48) & 3] = Wcell;
            }
        } else {
            Gb[(ps - 32) & 31] = G0;
            Wb[(ps - 32) & 31] = Wcell;
        }
        sh.pred[ps][threadIdx.x] = (unsigned char)pred;
    }

    IntResult out;
    out.r.none = n_cells == 0;
    out.r.dG = INFINITY;
    out.r.t = 0.0;
    out.r.conflict = false;

    const int nch = (min(nmax, NS) + kC - 1) / kC;
    // (kDeferPick is decided at the end: int_core.hpp kPickMargin)
    bool second = RESOLVE && !out.r.none && nTie == 1;
    bool any_second = RESOLVE && __builtin_amdgcn_ballot_w64(second) != 0ull;   // wave-uniform

    // ---- walk from the picked cell (and, RESOLVE, from the one that ties with it): traceback by
    //      pointer into the LDS scratch, then replay forwards in f64 with Primer3's operation order
    //      (fillMatrix / maxTM), then the terminal dG as thal() compares it
    double S = 0.0, S_1 = 0.0, Gt_1 = 0.0;
    int H = 0, P = 0, H_1 = 0, P_1 = 0, dpath = 0, dpath_1 = 0, endW = pickW;
    bool unsettled = false;   // RESOLVE: the two walks were compared, but one of them may not be the walk thal() made
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            if (!any_second) break;
            S_1 = S;
            H_1 = H;
            P_1 = P;
            dpath_1 = dpath;
            endW = pickW2;
        }
        P = 0;
        dpath = 0;
        {
            // the picked cell (which may have no slot) opens the path; the walk goes on from its predecessor
            const int endP = pass == 0 ? pickP : pickP2;
            const bool walk = !(out.r.none | ((pass == 1) & !second));
            int cur = endP & 0xff;
            bool done = !walk | (cur == 0xff);
            if (walk) {
                sh.path[0][threadIdx.x] = (unsigned short)(core_word(endW) & 0x3fff);
                P = 1;
                dpath |= (endP & 0x100) ? kDeferPathTie : 0;
            }
            for (int pc_ = nch - 1; pc_ >= 0; --pc_) {
                const int pc = __builtin_amdgcn_readfirstlane(pc_);
                int W[kC];
#pragma unroll
                for (int e = 0; e < kC; ++e) W[e] = slot_of<NS>(Wa, Wb, Wc, pc * kC + e);
#pragma unroll
                for (int e = kC - 1; e >= 0; --e) {
                    const int slot = pc * kC + e;
                    const int pr = sh.pred[slot][threadIdx.x];
                    const bool hit = !done & (slot < n_cells) & ((W[e] & 0xff) == cur);   // predecessors only: cur is never the end cell
                    if (hit) sh.path[P & (kPathMax - 1)][threadIdx.x] = (unsigned short)(core_word(W[e]) & 0x3fff);
                    dpath |= (hit & (((softTie >> (slot & 63)) & 1ull) != 0ull)) ? kDeferPathTie : 0;
                    P += hit ? 1 : 0;
                    cur = hit ? pr : cur;
                    done = done | (hit & (pr == 0xff));
                }
            }
        }
        S = 0.0;
        H = 0;
        {
            int prevW = 0;
            const int maxP = wave_max_u8(P);
            for (int step_ = 0; step_ < maxP; ++step_) {
                const int step = __builtin_amdgcn_readfirstlane(step_);
                const int e = P - 1 - step;
                if (e >= 0) {
                    const int Wstep = sh.path[e & (kPathMax - 1)][threadIdx.x];
                    CellCtx cc;
                    const CellBases b = cell_bases(q, (Wstep >> 4) & 15, Wstep & 15, cc);
                    if (step == 0) {
                        S = F.S[b.idxL];
                        H = F.H[b.idxL];
                    } else if (((Wstep & 0xff) - (prevW & 0xff)) == 0x11) {
                        S = S + F.S[b.wc];
                        H = H + F.H[b.wc];
                    } else {
                        const CandGeom g = cand_geometry(cc, prevW);
                        const CandLoad v = cand_gather(F, g);
                        S = ((v.sLX + v.sY) + v.sZ) + S;
                        H = v.hLX + v.hY + H;
                    }
                    prevW = Wstep;
                }
            }
        }
        // the replayed enthalpy must be the tracked one; anything else is handed on
        dpath |= (!out.r.none & (H != (endW >> 16) * 10) & ((pass == 0) | second)) ? kDeferReplay : 0;
        if (RESOLVE) {
            // thal.c thal(): the nudged dG the terminal pick compares
            CellCtx cc;
            const CellBases b = cell_bases(q, (endW >> 4) & 15, endW & 15, cc);
            const double rSn = F.S[b.idxR] + kTiny, rHn = (double)F.H[b.idxR] + kTiny;
            const double Gt = (((double)H + rHn) + K.init_H) - kT37 * ((S + rSn) + K.init_S);
            if (pass == 0) {
                Gt_1 = Gt;
                if (decisions_only) {   // wave-uniform: a decision that is not close needs no second walk
                    // what drawDimer() would report for this walk (the totals at the end of this function)
                    const double G0 = (double)(H + F.H[b.idxR] + 200) -
                                      (K.temp_k * (((S + F.S[b.idxR]) + K.init_S) + ((P - 1) * K.salt)));
                    const int ht0 = (H + F.H[b.idxR]) / 10;
                    second = second & !tied_pick_cannot_conflict(K, G0, P - 1, tieLo - ht0, tieHi - ht0);
                    any_second = __builtin_amdgcn_ballot_w64(second) != 0ull;
                }
            } else {
                // The comparison is between the doubles of BOTH walks, and they differ by rounding alone: a walk through a
                // cell with an equal-valued alternative (or one whose replay missed the tracked enthalpy) may carry other
                // last bits than the walk thal() made, whichever of the two is kept -- the pick stays open then.
                // (Found by the pair campaign, seed 301: GCGGCGGCCGCCGC x GCCGGCCGGGCGGG answered 15e-12 cal/mol off,
                // TCTAGACTAGCCAGCA x TGAAGAAAGCTAAGTC with the other cell's structure.)
                unsettled = second & ((dpath | dpath_1) != 0);
                if (second & !(Gt < Gt_1)) {   // strict: the first cell stays unless the second is lower
                    S = S_1;
                    H = H_1;
                    P = P_1;
                    dpath = dpath_1;
                    endW = pickW;
                }
            }
        }
    }
    if (RESOLVE && any_second && !second) {   // lanes without a second walk keep their first one
        S = S_1;
        H = H_1;
        P = P_1;
        dpath = dpath_1;
        endW = pickW;
    }
    // (a path tie is looked at again below: it only changes the number of pairs of the walked structure)
    defer |= dpath & ~kDeferPathTie;
    // ---- thal.c drawDimer(): totals
    {
        CellCtx cc;
        const CellBases b = cell_bases(q, (endW >> 4) & 15, endW & 15, cc);
        const double rS = F.S[b.idxR];
        const int rH = F.H[b.idxR];
        const double dH = (double)(H + rH + 200);
        const double dS = (S + rS) + K.init_S;
        const int N = P - 1;
        const double t = (dH / ((dS + (N * K.salt)) + K.RC)) - kAbsZero;
        const double G = dH - (K.temp_k * (dS + (N * K.salt)));
        if (!out.r.none) {
            out.r.dG = G;
            out.r.t = t;
            out.r.conflict = G <= K.g_cut;
        }
        // a tied terminal pick: resolved above (list mode, one tie), final as "no conflict" when only decisions
        // are asked for and this one is not close (int_core.hpp kPickMargin), handed on otherwise
        const bool open_tie = RESOLVE ? (nTie > 1) | unsettled : (nTie > 0);
        const int ht = (H + rH) / 10;
        defer |= (open_tie && !(decisions_only && tied_pick_cannot_conflict(K, G, N, tieLo - ht, tieHi - ht))) ? kDeferPick : 0;
        // a cell of the path with an equal-valued alternative of the same enthalpy: the other path gives the same
        // (dH, dS) with another N, so the same bound with an empty enthalpy range
        defer |= ((dpath & kDeferPathTie) && !(decisions_only && tied_pick_cannot_conflict(K, G, N, 0, 0))) ? kDeferPathTie : 0;
    }
    out.defer = out.r.none ? 0 : defer;
    return out;
}


template <class SH>
__device__ __forceinline__ void load_tables_int(SH &sh, const IntArgs &a)
{
    constexpr int kThreadsI = SH::kThreads;
    for (int e = threadIdx.x; e < SH::kRows * 64; e += kThreadsI) sh.T[e] = a.it->T[e];
    for (int e = threadIdx.x; e < FastTables::kCount; e += kThreadsI) {
        sh.F.S[e] = a.f.ft->S[e];
        sh.F.H[e] = a.f.ft->H[e];
        sh.g[e] = a.it->g[e];
    }
    for (int e = threadIdx.x; e < 100; e += kThreadsI)
        sh.cq[e] = 620300.0 * ((a.f.c.init_S + a.f.ft->S[FastTables::kEndR + e]) + a.f.c.RC);
    __syncthreads();
}

// One lock-step DP of the wave: lane = pair (row, col); `take` lanes are computed, `pass_on`
// lanes go to the output list untouched (flag kept).  same_row: all lanes share `row`.
template <int NS, bool RESOLVE, class SH>
__device__ __forceinline__ void wave_pairs(SH &sh, const IntArgs &a, int row, int col, uint64_t pa,
                                           uint64_t pb, bool inside, bool pass_on, unsigned pass_flag,
                                           bool same_row)
{
    const int lane = threadIdx.x & 63;
    SeqPair q;
    unsigned rowmask;
    int n_cells = setup_pair(pa, pb, a.f.k, q, rowmask);
    const bool sym = self_complementary(pa, a.f.k) && self_complementary(pb, a.f.k);
    // the cells of the lane's last row (the last ones of its row-major order) take no slot: run_pair_int
    const int last_row = __popc(spaced_mask(q.s2, 3 - (int)((q.s1 >> (2 * (a.f.k - 1))) & 3u), q.lenmask));
    bool spill = inside & (pass_on | (n_cells - last_row > NS - 1) | sym);   // one slot stays free for the last row's writes
    unsigned flag = pass_on ? pass_flag : 0u;
    if (!inside | spill) n_cells = 0;
    int nmax = wave_max_u8(n_cells);
    // Lock-step lanes pay for the largest table of their wave (work ~ slots^2).  A few lanes
    // far above the rest (mixed compositions at bin boundaries) are cheaper in a sorted list
    // stage than as a drag on 64 lanes.  (Matrix mode only: the list stage's lanes are sorted by table size
    // already, and which pairs it runs must not depend on the order the list was filled in -- the stage
    // counters are then the same in every run.)
    for (int round = 0; round < (RESOLVE ? 0 : 6); ++round) {
        const int next = wave_max_u8(n_cells < nmax ? n_cells : 0);
        const int m = __popcll(__ballot(n_cells == nmax));
        if (next == 0 || nmax * nmax - next * next <= kDragCost * m) break;   // wave-uniform
        if (n_cells == nmax) {
            spill = true;
            n_cells = 0;
        }
        nmax = next;
    }
    if (nmax == 0) {   // wave-uniform: nothing to compute
        if (spill) {
            const uint32_t at = atomicAdd(a.f.ovf_count, 1u);
            if (at < a.f.ovf_cap) a.f.ovf_list[at] = make_uint2((unsigned)row | flag, (unsigned)col);
        }
        // lanes inside the block that are not handed on have no complementary cell at all: thal() finds no
        // structure (dG = inf, t = 0, no conflict) -- the planes are the caller's memory and must say so
        if (inside & !spill) {
            const size_t orow = (size_t)(row - a.f.sinks.row0), ocol = (size_t)(col - a.f.sinks.col0);
            if (a.f.sinks.dg) a.f.sinks.dg[orow * (size_t)a.f.sinks.ncols + ocol] = INFINITY;
            if (a.f.sinks.tm) a.f.sinks.tm[orow * (size_t)a.f.sinks.ncols + ocol] = 0.0;
        }
        return;
    }
    const bool decisions_only = a.f.sinks.dg == nullptr && a.f.sinks.tm == nullptr;   // wave-uniform
    const IntResult r = run_pair_int<NS, RESOLVE, SH>(sh, a.f.c, q, rowmask, n_cells, nmax, decisions_only);
    const bool deferred = inside & !spill & (r.defer != 0);
    if (deferred) flag = kNeedsF64;
    spill |= deferred;
    if (spill) {
        const uint32_t at = atomicAdd(a.f.ovf_count, 1u);
        if (at < a.f.ovf_cap) a.f.ovf_list[at] = make_uint2((unsigned)row | flag, (unsigned)col);
    }
    if (a.reasons) {
        const unsigned long long dm = __ballot(deferred);
        if (dm) {   // wave-uniform
            if (lane == 0) atomicAdd(&a.reasons[a.stat_off], (unsigned long long)__popcll(dm));
#pragma unroll
            for (int bit = 0; bit < 7; ++bit) {
                const unsigned long long bm = __ballot(deferred & ((r.defer >> bit) & 1));
                if (lane == 0 && bm) atomicAdd(&a.reasons[a.stat_off + 1 + bit], (unsigned long long)__popcll(bm));
            }
            // a few samples for diagnostics: row << 40 | col << 16 | reasons
            if (deferred && a.stat_off == 0 && a.reasons[8] < 1024ull) {
                const unsigned long long at = atomicAdd(&a.reasons[8], 1ull);
                if (at < 1024ull)
                    a.reasons[9 + at] = ((unsigned long long)row << 40) | ((unsigned long long)col << 16) |
                                        (unsigned long long)r.defer;
            }
        }
    }
    // ---- sinks (conflicts are rare: one atomic OR per conflicting pair, one add per wave)
    const bool live = inside & !spill;
    const bool hit = live & r.r.conflict;
    const size_t orow = (size_t)(row - a.f.sinks.row0);
    const size_t ocol = (size_t)(col - a.f.sinks.col0);
    if (hit && a.f.sinks.bitmap)
        atomicOr((unsigned long long *)&a.f.sinks.bitmap[orow * (size_t)a.f.sinks.words + (ocol >> 6)],
                 1ull << (ocol & 63));
    if (hit) sink_edge(a.f.sinks, row, col, r.r.dG);
    if (a.f.sinks.row_conflicts) {
        if (same_row) {
            const unsigned long long bits = __ballot(hit);
            if (lane == 0 && bits) atomicAdd(&a.f.sinks.row_conflicts[row], (unsigned)__popcll(bits));
        } else if (hit) {
            atomicAdd(&a.f.sinks.row_conflicts[row], 1u);
        }
    }
    if (live) {
        if (a.f.sinks.dg) a.f.sinks.dg[orow * (size_t)a.f.sinks.ncols + ocol] = r.r.dG;
        if (a.f.sinks.tm) a.f.sinks.tm[orow * (size_t)a.f.sinks.ncols + ocol] = r.r.t;
    }
}

// Matrix mode: wave = one row x 64 consecutive entries of the composition-sorted column list.
template <int NS, int THREADS, int TROWS>
__global__ void __launch_bounds__(THREADS) k_pairs_int(IntArgs a)
{
    typedef SharedI<NS, THREADS, TROWS> SH;
    __shared__ SH sh;
    load_tables_int(sh, a);
    const int lane = threadIdx.x & 63;
    const int ncolg = (a.f.col1 - a.f.col0 + 63) >> 6;
    // Work items (one row x 64 consecutive sorted columns) are handed out to the WAVES from a
    // counter: a block's waves need not march together, and the composition-sorted column groups
    // (whose cost differs by up to 20 % between the ends and the middle of the order) no longer
    // fall to fixed blocks.
    const unsigned n_items = (unsigned)ncolg * (unsigned)(a.f.row1 - a.f.row0);
    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.work_counter, 1u);
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= n_items) break;   // wave-uniform
        // (the fetch above relies on the whole wave arriving here together: wave_pairs returns only through
        // wave-uniform branches -- keep it that way, or a lane runs ahead of the readfirstlane)
        const int row = a.f.row0 + (int)(item / (unsigned)ncolg), cg = (int)(item % (unsigned)ncolg);
        const int cq = a.f.col0 + cg * 64 + lane;
        const bool inside = cq < a.f.col1;
        const uint64_t pa = a.f.pool[row];
        const uint64_t pb = a.f.cols_sorted[inside ? cq : a.f.col0];
        const int col = (int)a.f.perm[inside ? cq : a.f.col0];
        wave_pairs<NS, false, SH>(sh, a, row, col, pa, pb, inside, false, 0u, true);
    }
}

// List mode: the pairs the matrix-mode kernel sent away get a second chance in lanes sorted by
// table size (64 slots), with the two-cell terminal pick settled by a second walk; what meets
// another kind of tie or still does not fit goes to the output list.  Batches of kListBatchI x 512 entries are
// counting-sorted in LDS (the predecessor rows double as scratch) and written back in place.
constexpr int kListBatchI = 7;
__global__ void __launch_bounds__(kThreadsI) k_pairs_int_list(IntArgs a)
{
    typedef SharedI<kSlotsList, kThreadsI, IntTables::kRows> SH;
    __shared__ SH sh;
    if (*a.f.in_count == 0u) return;   // an empty list: no table loads
    load_tables_int(sh, a);
    static_assert(sizeof(sh.pred) >= sizeof(uint2) * kListBatchI * kThreadsI + sizeof(unsigned) * 256,
                  "the predecessor rows must hold one sorted batch");
    uint2 *sorted = reinterpret_cast<uint2 *>(&sh.pred[0][0]);
    unsigned *hist = reinterpret_cast<unsigned *>(sorted + kListBatchI * kThreadsI);
    const long n_work = (long)min(*a.f.in_count, a.f.ovf_cap);
    // A batch is `depth` x 512 entries, sorted together and run as `depth` lock-step passes.  Long lists take the
    // full depth (better sorted lanes); a short list -- the screen of a reference-sized pool hands on some ten
    // thousand pairs -- is spread over all the blocks instead of keeping two dozen of them busy for seven passes
    // (2,000 primers: 2.30 ms -> one pass).  Which pairs the stage answers does not depend on the depth.
    const long per_pass = (long)gridDim.x * kThreadsI;
    const int depth = (int)min((long)kListBatchI, max(1L, (n_work + per_pass - 1) / per_pass));
    const long batch = (long)depth * kThreadsI;
    const long n_batches = (n_work + batch - 1) / batch;
    __shared__ unsigned next_batch;
    for (;;) {
        // batches are handed out from a counter (a block's batches differ in cost with the table sizes)
        if (threadIdx.x == 0) next_batch = atomicAdd(a.work_counter, 1u);
        __syncthreads();
        const long bt = (long)next_batch;
        if (bt >= n_batches) break;   // block-uniform
        uint2 mine[kListBatchI];
        int key[kListBatchI];
        for (int e = threadIdx.x; e < 256; e += kThreadsI) hist[e] = 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kListBatchI; ++j) {
            const long w = bt * batch + (long)j * kThreadsI + threadIdx.x;
            const bool inside = (j < depth) & (w < n_work);   // passes beyond the depth hold padding only
            mine[j] = inside ? a.f.in_list[w] : make_uint2(0xffffffffu, 0u);
            int nc = 255;   // padding sorts last
            if (inside) {
                SeqPair q;
                unsigned rowmask;
                nc = min(setup_pair(a.f.pool[mine[j].x & ~kNeedsF64], a.f.pool[mine[j].y], a.f.k, q, rowmask), 254);
            }
            key[j] = nc;
            atomicAdd(&hist[nc], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {   // exclusive scan of the 256 bins
            unsigned v[4], sum = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = hist[threadIdx.x * 4 + q];
                sum += v[q];
            }
            unsigned incl = sum;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned up = __shfl_up(incl, off);
                if ((int)threadIdx.x >= off) incl += up;
            }
            unsigned run = incl - sum;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hist[threadIdx.x * 4 + q] = run;
                run += v[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kListBatchI; ++j) sorted[atomicAdd(&hist[key[j]], 1u)] = mine[j];
        __syncthreads();
        uint2 *own = const_cast<uint2 *>(a.f.in_list) + bt * batch;
        const long n_own = min(batch, n_work - bt * batch);
#pragma unroll
        for (int j = 0; j < kListBatchI; ++j) {
            const long e = (long)j * kThreadsI + threadIdx.x;
            if (e < n_own) own[e] = sorted[e];
        }
        __syncthreads();   // the predecessor rows are the DP's again; the writes are visible to the block
        for (int j = 0; j < depth; ++j) {
            const long e = (long)j * kThreadsI + threadIdx.x;
            const bool inside = e < n_own;
            const uint2 pr = own[inside ? e : 0];
            // marked entries are retried as well: most of them met nothing but a terminal pick
            // shared by two cells, which this mode settles by walking both
            const int row = (int)(pr.x & ~kNeedsF64), col = (int)pr.y;
            wave_pairs<kSlotsList, true, SH>(sh, a, row, col, a.f.pool[inside ? row : 0], a.f.pool[inside ? col : 0], inside,
                                         false, 0u, false);
        }
        __syncthreads();
    }
}

}  // namespace

int pairs_int_slots() { return kSlotsMatrix; }
int pairs_int_slots_small() { return kSlotsSmall; }

hipError_t launch_pairs_int(const PairKernelArgs &a, const IntTables *it, unsigned long long *reasons, int n_cu,
                            hipStream_t stream)
{
    IntArgs x;
    FastArgs &f = x.f;
    f.ft = a.ft;
    f.c = a.c;
    f.pool = a.pool;
    f.cols_sorted = a.cols_sorted;
    f.perm = a.perm;
    f.k = a.k;
    f.row0 = a.row0;
    f.row1 = a.row1;
    f.col0 = a.col0;
    f.col1 = a.col1;
    f.sinks = a.sinks;
    f.ovf_list = a.overflow_list;
    f.ovf_count = a.overflow_count;
    f.ovf_cap = a.overflow_cap;
    f.in_list = nullptr;
    f.in_count = nullptr;
    x.it = it;
    x.reasons = reasons;
    x.stat_off = 0;
    x.work_counter = a.work_counter;
    if (hipError_t e = hipMemsetAsync(a.work_counter, 0, sizeof(unsigned), stream); e != hipSuccess) return e;
    // one persistent block per CU (about 150 KB of LDS each)
    if (a.k <= 13) {
        constexpr int kRowsPerBlock = kThreadsSmall / 64;
        const long tiles = (long)((a.col1 - a.col0 + 63) / 64) *
                           (long)((a.row1 - a.row0 + kRowsPerBlock - 1) / kRowsPerBlock);
        if (tiles <= 0) return hipSuccess;
        const int grid = (int)(tiles < (long)n_cu ? tiles : (long)n_cu);
        hipLaunchKernelGGL((k_pairs_int<kSlotsSmall, kThreadsSmall, kRowsSmall>), dim3(grid), dim3(kThreadsSmall), 0,
                           stream, x);
    } else {
        constexpr int kRowsPerBlock = kThreadsI / 64;
        const long tiles = (long)((a.col1 - a.col0 + 63) / 64) *
                           (long)((a.row1 - a.row0 + kRowsPerBlock - 1) / kRowsPerBlock);
        if (tiles <= 0) return hipSuccess;
        const int grid = (int)(tiles < (long)n_cu ? tiles : (long)n_cu);
        hipLaunchKernelGGL((k_pairs_int<kSlotsMatrix, kThreadsI, IntTables::kRows>), dim3(grid), dim3(kThreadsI), 0,
                           stream, x);
    }
    return hipGetLastError();
}

hipError_t launch_pairs_int_list(const PairKernelArgs &a, const IntTables *it, const uint2 *in_list,
                                 const uint32_t *in_count, unsigned long long *reasons, int n_cu, hipStream_t stream)
{
    IntArgs x;
    FastArgs &f = x.f;
    f.ft = a.ft;
    f.c = a.c;
    f.pool = a.pool;
    f.cols_sorted = nullptr;
    f.perm = nullptr;
    f.k = a.k;
    f.row0 = a.row0;
    f.row1 = a.row1;
    f.col0 = a.col0;
    f.col1 = a.col1;
    f.sinks = a.sinks;
    f.ovf_list = a.overflow_list;
    f.ovf_count = a.overflow_count;
    f.ovf_cap = a.overflow_cap;
    f.in_list = in_list;
    f.in_count = in_count;
    x.it = it;
    x.reasons = reasons;
    x.stat_off = 1033;
    x.work_counter = a.work_counter;
    if (hipError_t e = hipMemsetAsync(a.work_counter, 0, sizeof(unsigned), stream); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pairs_int_list, dim3(n_cu), dim3(kThreadsI), 0, stream, x);
    return hipGetLastError();
}

}  // namespace msspe
