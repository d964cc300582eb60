// thal_pairs_split.hip -- all-pairs cross-dimer kernel for LONG oligos (15 .. 32 bases): the
// exact-integer DP of thal_pairs_int.hip with 5-bit cell coordinates and the DP table of one pair
// SPLIT over Q lanes.
//
// Same job and outputs as thal_pairs_int.hip (the reference's "format N^2 lines -> ntthal -> parse"
// loop, /root/reference/od-msspe/src/delta_g.rs:61-153, for --kmer-size 15 and above; Primer3 2.6.1
// thal() restated from SURVEY.md Appendix C.3).  A pair of k-mers has about k^2 / 4 complementary
// cells (100 at k = 20, 196 at k = 28): too many for one lane's registers.  Here Q = 2, 4 or 8
// neighbouring lanes share one pair:
//   * cell number c (row-major order of the complementary cells) lives in lane c % Q, register slot
//     c / Q, as (G, W): G = exact 2000 * dG, W = (H / 10 + bias) << 16 | po << 10 | im1 << 5 | jm1;
//   * all Q lanes enumerate the cells together; for cell c each lane tries its own earlier slots
//     as predecessors (compile-time register numbers, split_tables.hpp: two LDS gathers per try),
//     then the Q partial minima, tie flags and the (i-1, j-1) neighbour are merged with DPP
//     quad permutes; every lane then evaluates the cell (maxTM, loop acceptance, terminal pick)
//     redundantly and the owner publishes it;
//   * the traceback walks the cells downwards, the lane holding the current cell announcing its
//     predecessor to the group; the path is replayed in f64 with Primer3's operation order, so
//     dS, dH, dG and t carry the same bits as the CPU oracle.  A terminal pick shared by two cells
//     is settled by walking both (as the list mode of thal_pairs_int.hip does).
// Pairs that meet another exact tie, a near-tie of Tm, more than 64 Q cells, or two
// self-complementary oligos go to the hand-over list and are finished by the generic f64 kernel.
// Nothing is approximated.
#include <cstdlib>

#include "split_core.hpp"

namespace msspe {

namespace {

typedef int v32i __attribute__((ext_vector_type(32)));

constexpr int kC = 4;               // slots per chunk
constexpr int kLaneSlots = 64;      // register slots per lane (two planes of 32 + 32)
constexpr int kThreadsS = 512;
constexpr int kPathMaxS = 32;       // a path has at most k <= 32 cells
constexpr int kEmptyS = 0x3ff;      // coordinates (31, 31): fails every geometry test
constexpr int kNoCell = 0x3ff;      // "no predecessor"

struct SharedS {
    int L[1024];
    int X[W_::kXCount];
    double S[W_::kCount];
    int H[W_::kCount];
    int g[W_::kCount];
    double cq[100];                                   // 620300 * (init_S + rS + RC) per right-end context
    int h_bias10;                                     // 10 * bias of the enthalpy field of the packed words
    unsigned short pred[kLaneSlots][kThreadsS];       // predecessor coordinates of the lane's slots
    unsigned short path[kPathMaxS][kThreadsS];        // po << 10 | im1 << 5 | jm1 of the path cells
};

struct ICellS {
    int cgeo;      // (im1 - 1) * 32 + (jm1 - 1)
    int cstk;      // coordinates of (i-1, j-1), or a value no slot holds
    int jm1p;
    int a16;       // cell base << 4
    int yTS, yMM;  // cell-side mismatch terms (G units)
};
struct IBestS {
    int G, W;
    int W2;   // a later candidate with the same value (meaningful while the tie flag is set)
};
struct MasksS {
    unsigned long long tie, multi, stHave;   // multi: three or more candidates share the minimum
};

// One predecessor slot against the cell, in three phases so that a chunk issues all its LDS gathers
// before any is consumed: geometry (integer VALU) -> two gathers -> finish.
struct VisitS {
    int li4, xi4;   // byte offsets into L and X
    int y;          // cell-side term for this kind of loop
    bool stack;     // the predecessor is the cell (i-1, j-1)
};
__device__ __forceinline__ VisitS visit_geo_s(const ICellS &c, int Wp)
{
    VisitS v;
    const int gp = Wp & 0x3ff;
    const int d = c.cgeo - gp;
    const int jj = Wp & 31;
    const bool geo = (jj <= c.jm1p) & (d >= 0);
    v.stack = gp == c.cstk;
    const bool l1z = d < 32, l2z = jj == c.jm1p;
    const bool bulge = l1z | l2z;
    const int po4 = (Wp >> 8) & 0xfc;
    const int pe4 = (po4 & 12) | c.a16;
    const int bx4 = (l1z ? (d << 6) : ((d << 1) + 4 * (W_::kXB2 - W_::kXB1))) + (4 * W_::kXB1 + pe4);
    const bool m11 = d == 0x21;
    const int xi4 = bulge ? bx4 : (po4 + (m11 ? 4 * W_::kXMM : 0));
    // an impossible geometry reads L[0] = kBig (the stacked pair's row) and X[0]
    v.li4 = geo ? (d << 2) : 0;
    v.xi4 = geo ? xi4 : 0;
    v.y = m11 ? c.yMM : (bulge ? 0 : c.yTS);
    return v;
}
__device__ __forceinline__ void visit_fin_s(const VisitS &v, int lv, int xv, int Gp, int Wp, IBestS &best,
                                            IBestS &stk, MasksS &m)
{
    const int cand = lv + xv + v.y + Gp;
    const bool better = cand < best.G;
    const bool eq = cand == best.G;
    const unsigned long long bm = __builtin_amdgcn_ballot_w64(better), em = __builtin_amdgcn_ballot_w64(eq);
    m.multi = (m.multi & ~bm) | (m.tie & em);
    m.tie = (m.tie & ~bm) | em;
    best.G = better ? cand : best.G;
    best.W = better ? Wp : best.W;
    best.W2 = eq ? Wp : best.W2;
    stk.G = v.stack ? Gp : stk.G;
    stk.W = v.stack ? Wp : stk.W;
    m.stHave |= __builtin_amdgcn_ballot_w64(v.stack);
}

__device__ __forceinline__ int slot_s(const v32i a, const v32i b, int x)
{
    return x < 32 ? a[x & 31] : b[(x - 32) & 31];
}

// The lane's slots below `upto` as predecessors of the cell, kC at a time (compile-time register
// numbers; left through a wave-uniform branch at the first chunk nobody has filled).
template <int PC = 0>
__device__ __forceinline__ void scan_s(const v32i Ga, const v32i Wa, const v32i Gb, const v32i Wb, int upto,
                                       const char *L, const char *X, const ICellS &c, IBestS &best,
                                       IBestS &stk, MasksS &m)
{
    if constexpr (PC * kC < kLaneSlots) {
        if (PC * kC < upto) {   // wave-uniform
            asm volatile("" ::"n"(PC));   // keeps the chunks from being merged into selects
            VisitS v[kC];
            int lv[kC], xv[kC];
#pragma unroll
            for (int e = 0; e < kC; ++e) v[e] = visit_geo_s(c, slot_s(Wa, Wb, PC * kC + e));
#pragma unroll
            for (int e = 0; e < kC; ++e) {
                lv[e] = *(const int *)(L + v[e].li4);
                xv[e] = *(const int *)(X + v[e].xi4);
            }
#pragma unroll
            for (int e = 0; e < kC; ++e)
                visit_fin_s(v[e], lv[e], xv[e], slot_s(Ga, Gb, PC * kC + e), slot_s(Wa, Wb, PC * kC + e), best, stk, m);
            scan_s<PC + 1>(Ga, Wa, Gb, Wb, upto, L, X, c, best, stk, m);
        }
    }
}

// Lane permutations inside a group of Q lanes (DPP quad_perm: no LDS traffic).
template <int MASK>
__device__ __forceinline__ int lane_xor(int v)
{
    static_assert(MASK == 1 || MASK == 2 || MASK == 4, "quad permutes and the half-row mirror");
    // MASK 4: row_half_mirror pairs lane i with lane 7 - i of its group of eight, which merges the two
    // quads just as well as i ^ 4 would (both already agree within themselves)
    return __builtin_amdgcn_mov_dpp(v, MASK == 1 ? 0xB1 : (MASK == 2 ? 0x4E : 0x141), 0xf, 0xf, false);
}

// tie: 0 = one candidate holds the minimum, 1 = two (b.W, b.W2), 2 = more
template <int Q, int MASK = 1>
__device__ __forceinline__ void merge_min(int lane, IBestS &b, int &tie)
{
    if constexpr (MASK < Q) {
        const int oG = lane_xor<MASK>(b.G), oW = lane_xor<MASK>(b.W), oW2 = lane_xor<MASK>(b.W2),
                  oT = lane_xor<MASK>(tie);
        const bool lt = oG < b.G, eq = oG == b.G;
        const bool other_first = (lane & MASK) != 0;   // equal values: every lane keeps the lower lane's words
        const int lowW = other_first ? oW : b.W, highW = other_first ? b.W : oW;
        const int both = tie + oT + 1;                 // candidates at the shared minimum, minus one
        b.W2 = lt ? oW2 : (eq ? highW : b.W2);
        b.W = lt ? oW : (eq ? lowW : b.W);
        tie = lt ? oT : (eq ? min(both, 2) : tie);
        b.G = lt ? oG : b.G;
        merge_min<Q, MASK * 2>(lane, b, tie);
    }
}

template <int Q, int MASK = 1>
__device__ __forceinline__ int merge_or(int v)
{
    if constexpr (MASK < Q) return merge_or<Q, MASK * 2>(v | lane_xor<MASK>(v));
    else return v;
}

enum : int {
    kDeferTm = 1,
    kDeferLoopEq = 2,
    kDeferLoopTie = 4,
    kDeferBad = 8,
    kDeferPick = 16,
    kDeferReplay = 32,
    kDeferPathTie = 64,
};

struct SplitResult {
    double dG, t;
    bool none, conflict;
    int defer;
};

__device__ __forceinline__ int wave_max_s(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}

// thal ANY for the pair shared by the Q lanes of a group (all of them return the same result).
// n_cells == 0: idle group.
template <int Q>
__device__ __forceinline__ SplitResult run_pair_split(SharedS &sh, const ThalConsts &K, const SeqW &q,
                                                      unsigned long long rowmask, int n_cells, int nmax)
{
    const int lane = threadIdx.x & 63, ql = lane & (Q - 1);
    v32i Ga = 0, Wa = kEmptyS, Gb = 0, Wb = kEmptyS;
    const int hb10 = sh.h_bias10;   // H of a packed word = (W >> 16, unsigned) * 10 - hb10
    int defer = 0;
    unsigned long long Rrem = rowmask, mrem = 0;
    int im1 = 0, jm1 = 0;
    int pickG = 0x7fffffff, pickW = 0, pickW2 = 0, nTie = 0;
    unsigned long long softTie = 0ull;   // per lane: own slots whose value has an equal-valued alternative

    for (int slot_ = 0; slot_ < nmax; ++slot_) {
        const int slot = __builtin_amdgcn_readfirstlane(slot_);
        // ---- next complementary cell in row-major order
        const bool newrow = mrem == 0;
        const int t = __ffsll((long long)Rrem) - 1;
        const int a_new = (int)((q.s1 >> (t & 63)) & 3);
        const unsigned long long m_new = spaced_mask64(q.s2, 3 - a_new, q.lenmask);
        im1 = newrow ? (t >> 1) : im1;
        Rrem = newrow ? (Rrem & (Rrem - 1)) : Rrem;
        mrem = newrow ? m_new : mrem;
        jm1 = (__ffsll((long long)mrem) - 1) >> 1;
        mrem &= mrem - 1;
        im1 &= 31;
        jm1 &= 31;
        const CellS b = cell_s(q, im1, jm1);
        ICellS ic;
        ic.cgeo = (im1 - 1) * 32 + (jm1 - 1);
        ic.jm1p = jm1 - 1;
        ic.cstk = ((im1 > 0) & (jm1 > 0)) ? ic.cgeo : 0x400;
        ic.a16 = b.a << 4;
        ic.yTS = sh.g[b.yTS];
        ic.yMM = sh.g[b.yMM];
        // ---- this lane's earlier slots as predecessors, then the group's minimum
        IBestS best, stk;
        best.G = W_::kValid;
        best.W = best.W2 = 0;
        stk.G = stk.W = stk.W2 = 0;
        MasksS sm;
        sm.tie = sm.multi = sm.stHave = 0ull;
        scan_s(Ga, Wa, Gb, Wb, (slot + Q - 1) / Q, (const char *)sh.L, (const char *)sh.X, ic, best, stk, sm);
        int tie = (int)((sm.tie >> lane) & 1ull) + (int)((sm.multi >> lane) & 1ull);
        const bool stMine = (sm.stHave >> lane) & 1ull;
        merge_min<Q>(lane, best, tie);
        stk.G = merge_or<Q>(stMine ? stk.G : 0);
        stk.W = merge_or<Q>(stMine ? stk.W : 0);
        const bool stHave = merge_or<Q>((int)stMine) != 0;
        // ---- thal.c maxTM(): helix extension if it raises Tm (see thal_pairs_int.hip)
        int H0 = sh.H[b.idxL], G0 = sh.g[b.idxL], pred = kNoCell, flags = 0;
        if (stHave) {
            const int rH = sh.H[b.idxR];
            const double cq = sh.cq[b.idxR - W_::kEndR];
            const int H1 = (int)((unsigned)stk.W >> 16) * 10 - hb10 + sh.H[b.wc];
            const int G1 = stk.G + sh.g[b.wc];
            const double A0 = (double)(H0 + 200 + rH), A1 = (double)(H1 + 200 + rH);
            const double B0 = (double)(2000 * H0 - G0) + cq, B1 = (double)(2000 * H1 - G1) + cq;
            const double lhs = A1 * B0, rhs = A0 * B1;
            const bool sure = (B0 < 0.0) & (B1 < 0.0) & (fabs(lhs - rhs) > 1e-9 * (fabs(lhs) + fabs(rhs)));
            flags |= sure ? 0 : kDeferTm;
            if (lhs > rhs) {
                H0 = H1;
                G0 = G1;
                pred = stk.W & 0x3ff;
            }
        }
        // ---- loops (thal.c calc_bulge_internal acceptance: dG of the candidate strictly lower)
        if (best.G <= G0) {
            const LoopIx g = loop_indices(b, best.W & 0xffff);
            const int Hw = sh.H[g.lx] + sh.H[g.y] + (int)((unsigned)best.W >> 16) * 10 - hb10;
            const unsigned long long my_bit = ((slot < n_cells) & ((slot & (Q - 1)) == ql)) ? (1ull << (slot / Q)) : 0ull;
            if (best.G < G0) {
                // two loop candidates with one value: if their enthalpies agree as well the cell's
                // value is the same either way and only a path through this cell is ambiguous
                if (tie == 1) {
                    const LoopIx g2 = loop_indices(b, best.W2 & 0xffff);
                    const int Hw2 = sh.H[g2.lx] + sh.H[g2.y] + (int)((unsigned)best.W2 >> 16) * 10 - hb10;
                    if (Hw2 == Hw) softTie |= my_bit;
                    else flags |= kDeferLoopTie;
                } else if (tie > 1) {
                    flags |= kDeferLoopTie;
                }
                flags |= ((Hw > 0) & (2000 * Hw - best.G > -1000)) ? kDeferBad : 0;
                H0 = Hw;
                G0 = best.G;
                pred = best.W & 0x3ff;
            } else if (Hw == H0) {
                softTie |= my_bit;
            } else {
                flags |= kDeferLoopEq;
            }
        }
        const int Wcell = (int)((unsigned)((H0 + hb10) / 10) << 16) | (b.po_c << 10) | (im1 << 5) | jm1;
        const bool in = slot < n_cells;
        defer |= in ? flags : 0;
        // ---- terminal pick (strict minimum of dG incl. the right end term, first in slot order)
        {
            const int Gt = G0 + sh.g[b.idxR];
            const bool pick = in & (Gt < pickG);
            const bool same = in & (Gt == pickG);
            pickW2 = (same & (nTie == 0)) ? Wcell : pickW2;
            nTie = pick ? 0 : (nTie + (same ? 1 : 0));
            pickG = pick ? Gt : pickG;
            pickW = pick ? Wcell : pickW;
        }
        // ---- the owner publishes the cell (wave-uniform register number)
        const int ls = slot / Q;
        const bool owner = (slot & (Q - 1)) == ql;
        if (ls < 32) {
            const int oldG = Ga[ls & 31], oldW = Wa[ls & 31];
            Ga[ls & 31] = owner ? G0 : oldG;
            Wa[ls & 31] = owner ? Wcell : oldW;
        } else {
            const int oldG = Gb[(ls - 32) & 31], oldW = Wb[(ls - 32) & 31];
            Gb[(ls - 32) & 31] = owner ? G0 : oldG;
            Wb[(ls - 32) & 31] = owner ? Wcell : oldW;
        }
        if (owner) sh.pred[ls][threadIdx.x] = (unsigned short)pred;
    }

    SplitResult out;
    out.none = n_cells == 0;
    out.dG = INFINITY;
    out.t = 0.0;
    out.conflict = false;
    defer |= nTie > 1 ? kDeferPick : 0;
    const bool second = !out.none && nTie == 1;
    const bool any_second = __builtin_amdgcn_ballot_w64(second) != 0ull;   // wave-uniform

    // ---- walk from the picked cell (and from the one that ties with it), replay forwards in f64
    double S = 0.0, S_1 = 0.0, Gt_1 = 0.0;
    int H = 0, P = 0, H_1 = 0, P_1 = 0, dpath = 0, dpath_1 = 0, endW = pickW;
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
            int cur = endW & 0x3ff;
            bool done = out.none | ((pass == 1) & !second);
            for (int c_ = nmax - 1; c_ >= 0; --c_) {
                const int c = __builtin_amdgcn_readfirstlane(c_);
                const int ls = c / Q;
                const int Wl = ls < 32 ? Wa[ls & 31] : Wb[(ls - 32) & 31];
                const int pr = sh.pred[ls][threadIdx.x];
                const bool mine = ((c & (Q - 1)) == ql) & !done & (c < n_cells) & ((Wl & 0x3ff) == cur);
                // the holder announces: bit 31, the cell's po | coordinates, its predecessor
                const int said = merge_or<Q>(mine ? (int)(0x80000000u | ((unsigned)(Wl & 0xffff) << 10) | (unsigned)pr) : 0);
                const bool hit = said != 0;
                if (hit) sh.path[P & (kPathMaxS - 1)][threadIdx.x] = (unsigned short)((said >> 10) & 0xffff);
                dpath |= (mine & (((softTie >> ls) & 1ull) != 0ull)) ? kDeferPathTie : 0;
                P += hit ? 1 : 0;
                cur = hit ? (said & 0x3ff) : cur;
                done = done | (hit & ((said & 0x3ff) == kNoCell));
            }
        }
        S = 0.0;
        H = 0;
        {
            int prevW = 0;
            const int maxP = wave_max_s(P);
            for (int step_ = 0; step_ < maxP; ++step_) {
                const int step = __builtin_amdgcn_readfirstlane(step_);
                const int e = P - 1 - step;
                if (e >= 0) {
                    const int Wstep = sh.path[e & (kPathMaxS - 1)][threadIdx.x];
                    const CellS b = cell_s(q, (Wstep >> 5) & 31, Wstep & 31);
                    if (step == 0) {
                        S = sh.S[b.idxL];
                        H = sh.H[b.idxL];
                    } else if (((Wstep & 0x3ff) - (prevW & 0x3ff)) == 0x21) {
                        S = S + sh.S[b.wc];
                        H = H + sh.H[b.wc];
                    } else {
                        const LoopIx g = loop_indices(b, prevW);
                        S = ((sh.S[g.lx] + sh.S[g.y]) + sh.S[g.zi]) + S;
                        H = sh.H[g.lx] + sh.H[g.y] + H;
                    }
                    prevW = Wstep;
                }
            }
        }
        // the replayed enthalpy must be the tracked one; anything else is handed on
        dpath |= (!out.none & (H != (int)((unsigned)endW >> 16) * 10 - hb10) & ((pass == 0) | second)) ? kDeferReplay : 0;
        {
            // thal.c thal(): the nudged dG the terminal pick compares
            const CellS b = cell_s(q, (endW >> 5) & 31, endW & 31);
            const double rSn = sh.S[b.idxR] + kTiny, rHn = (double)sh.H[b.idxR] + kTiny;
            const double Gt = (((double)H + rHn) + K.init_H) - kT37 * ((S + rSn) + K.init_S);
            if (pass == 0) {
                Gt_1 = Gt;
            } else {
                // both walks' doubles are compared: a flag on either of them leaves the pick open (thal_pairs_int.hip)
                if (second & ((dpath | dpath_1) != 0)) defer |= kDeferPick;
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
    if (any_second && !second) {   // lanes without a second walk keep their first one
        S = S_1;
        H = H_1;
        P = P_1;
        dpath = dpath_1;
        endW = pickW;
    }
    defer |= dpath;
    // ---- thal.c drawDimer(): totals
    {
        const CellS b = cell_s(q, (endW >> 5) & 31, endW & 31);
        const double rS = sh.S[b.idxR];
        const int rH = sh.H[b.idxR];
        const double dH = (double)(H + rH + 200);
        const double dS = (S + rS) + K.init_S;
        const int N = P - 1;
        const double t = (dH / ((dS + (N * K.salt)) + K.RC)) - kAbsZero;
        const double G = dH - (K.temp_k * (dS + (N * K.salt)));
        if (!out.none) {
            out.dG = G;
            out.t = t;
            out.conflict = G <= K.g_cut;
        }
    }
    const int group_defer = merge_or<Q>(defer);
    out.defer = out.none ? 0 : group_defer;
    return out;
}

struct SplitArgs {
    const SplitTables *st;
    ThalConsts c;
    const uint64_t *pool;
    const uint64_t *cols_sorted;
    const uint32_t *perm;
    int k;
    int row0, row1, col0, col1;
    PairSinks sinks;
    uint2 *ovf_list;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
    unsigned long long *reasons;   // optional statistics, layout of thal_pairs_int.hip's matrix mode
    unsigned *work_counter;        // next work item, zero at launch
    const uint2 *in_list;          // list mode: explicit pairs (row | kNeedsF64 mark, col)
    const uint32_t *in_count;
};
constexpr unsigned kMarkF64 = 0x80000000u;   // int_core.hpp kNeedsF64: an exact tie was met, only the f64 kernels answer

__device__ __forceinline__ void load_tables_s(SharedS &sh, const SplitArgs &a)
{
    for (int e = threadIdx.x; e < 1024; e += kThreadsS) sh.L[e] = a.st->L[e];
    for (int e = threadIdx.x; e < W_::kXCount; e += kThreadsS) sh.X[e] = a.st->X[e];
    for (int e = threadIdx.x; e < W_::kCount; e += kThreadsS) {
        sh.S[e] = a.st->S[e];
        sh.H[e] = a.st->H[e];
        sh.g[e] = a.st->g[e];
    }
    for (int e = threadIdx.x; e < 100; e += kThreadsS)
        sh.cq[e] = 620300.0 * ((a.c.init_S + a.st->S[W_::kEndR + e]) + a.c.RC);
    if (threadIdx.x == 0) sh.h_bias10 = a.st->h_bias * 10;
    __syncthreads();
}

// Matrix mode: wave = one row x 64 / Q consecutive entries of the composition-sorted column list.
// List mode (LIST): wave = 64 / Q consecutive entries of a hand-over list.  Entries that carry the tie mark are
// passed on untouched (this kernel would meet the same tie); the others left the integer stages for their size
// only, and two lanes per pair hold tables twice as large.
template <int Q, bool LIST>
__global__ void __launch_bounds__(kThreadsS) k_pairs_split(SplitArgs a)
{
    __shared__ SharedS sh;
    if (LIST && *a.in_count == 0u) return;   // nothing was handed on: not worth 53 KB of table loads per block
    load_tables_s(sh, a);
    constexpr int kPairsPerWave = 64 / Q;
    const int lane = threadIdx.x & 63;
    const int ncolg = LIST ? 1 : (a.col1 - a.col0 + kPairsPerWave - 1) / kPairsPerWave;
    const unsigned n_list = LIST ? min(*a.in_count, a.ovf_cap) : 0u;
    // work items (one row x 64 / Q consecutive sorted columns, or 64 / Q list entries) are handed to the waves
    // from a counter (fixed strides would tie every block to a few composition classes of very different cost)
    const unsigned n_items = LIST ? (n_list + kPairsPerWave - 1) / kPairsPerWave
                                  : (unsigned)ncolg * (unsigned)(a.row1 - a.row0);
    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = atomicAdd(a.work_counter, 1u);
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= n_items) break;   // wave-uniform
        int row, col;
        bool inside, marked = false;
        uint64_t pa, pb;
        if (LIST) {
            const unsigned e = item * kPairsPerWave + (unsigned)(lane / Q);
            inside = e < n_list;
            const uint2 pr = a.in_list[inside ? e : 0];
            marked = (pr.x & kMarkF64) != 0u;
            row = (int)(pr.x & ~kMarkF64);
            col = (int)pr.y;
            pa = a.pool[row];
            pb = a.pool[col];
        } else {
            row = a.row0 + (int)(item / (unsigned)ncolg);
            const int cg = (int)(item % (unsigned)ncolg);
            const int cq = a.col0 + cg * kPairsPerWave + lane / Q;
            inside = cq < a.col1;
            pa = a.pool[row];
            pb = a.cols_sorted[inside ? cq : a.col0];
            col = (int)a.perm[inside ? cq : a.col0];
        }
        const bool head = (lane & (Q - 1)) == 0;   // the lane that reports for its group
        SeqW q;
        unsigned long long rowmask;
        int n_cells = setup_pair_w(pa, pb, a.k, q, rowmask);
        const bool sym = self_complementary(pa, a.k) && self_complementary(pb, a.k);
        bool spill = inside & ((n_cells > kLaneSlots * Q) | sym | marked);
        if (!inside | spill) n_cells = 0;
        const int nmax = wave_max_s(n_cells);
        SplitResult r;
        r.none = true;
        r.conflict = false;
        r.defer = 0;
        r.dG = INFINITY;
        r.t = 0.0;
        if (nmax > 0) r = run_pair_split<Q>(sh, a.c, q, rowmask, n_cells, nmax);   // wave-uniform
        const bool deferred = inside & !spill & (r.defer != 0);
        spill |= deferred;
        if (spill & head) {
            const uint32_t at = atomicAdd(a.ovf_count, 1u);
            if (at < a.ovf_cap)
                a.ovf_list[at] = make_uint2((unsigned)row | ((LIST && (marked | deferred)) ? kMarkF64 : 0u), (unsigned)col);
        }
        if (a.reasons) {
            const unsigned long long dm = __ballot(deferred & head);
            if (dm) {   // wave-uniform
                if (lane == 0) atomicAdd(&a.reasons[0], (unsigned long long)__popcll(dm));
#pragma unroll
                for (int bit = 0; bit < 7; ++bit) {
                    const unsigned long long bm = __ballot(deferred & head & ((r.defer >> bit) & 1));
                    if (lane == 0 && bm) atomicAdd(&a.reasons[1 + bit], (unsigned long long)__popcll(bm));
                }
            }
        }
        const bool live = inside & !spill & head;
        const bool hit = live & r.conflict;
        const size_t orow = (size_t)(row - a.sinks.row0);
        const size_t ocol = (size_t)(col - a.sinks.col0);
        if (hit && a.sinks.bitmap)
            atomicOr((unsigned long long *)&a.sinks.bitmap[orow * (size_t)a.sinks.words + (ocol >> 6)],
                     1ull << (ocol & 63));
        if (hit) sink_edge(a.sinks, row, col, r.dG);
        if (a.sinks.row_conflicts) {
            if (LIST) {
                if (hit) atomicAdd(&a.sinks.row_conflicts[row], 1u);   // the lanes' rows differ
            } else {
                const unsigned long long bits = __ballot(hit);
                if (lane == 0 && bits) atomicAdd(&a.sinks.row_conflicts[row], (unsigned)__popcll(bits));
            }
        }
        if (live) {
            if (a.sinks.dg) a.sinks.dg[orow * (size_t)a.sinks.ncols + ocol] = r.dG;
            if (a.sinks.tm) a.sinks.tm[orow * (size_t)a.sinks.ncols + ocol] = r.t;
        }
    }
}

}  // namespace

int pairs_split_lanes(int k) { return k <= 20 ? 2 : (k <= 28 ? 4 : 8); }

hipError_t launch_pairs_split(const PairKernelArgs &a, const SplitTables *st, unsigned long long *reasons,
                              int lanes, hipStream_t stream)
{
    SplitArgs x;
    x.st = st;
    x.c = a.c;
    x.pool = a.pool;
    x.cols_sorted = a.cols_sorted;
    x.perm = a.perm;
    x.k = a.k;
    x.row0 = a.row0;
    x.row1 = a.row1;
    x.col0 = a.col0;
    x.col1 = a.col1;
    x.sinks = a.sinks;
    x.ovf_list = a.overflow_list;
    x.ovf_count = a.overflow_count;
    x.ovf_cap = a.overflow_cap;
    x.reasons = reasons;
    x.work_counter = a.work_counter;
    if (hipError_t e = hipMemsetAsync(a.work_counter, 0, sizeof(unsigned), stream); e != hipSuccess) return e;
    const int Q = (lanes == 2 || lanes == 4 || lanes == 8) ? lanes : pairs_split_lanes(a.k);   // option "split_lanes"
    const long tiles = (long)((a.col1 - a.col0 + 64 / Q - 1) / (64 / Q)) * (long)((a.row1 - a.row0 + 7) / 8);
    if (tiles <= 0) return hipSuccess;
    const int grid = (int)(tiles < 256L ? tiles : 256L);   // one persistent block per CU (about 150 KB of LDS)
    x.in_list = nullptr;
    x.in_count = nullptr;
    if (Q == 2) hipLaunchKernelGGL((k_pairs_split<2, false>), dim3(grid), dim3(kThreadsS), 0, stream, x);
    else if (Q == 4) hipLaunchKernelGGL((k_pairs_split<4, false>), dim3(grid), dim3(kThreadsS), 0, stream, x);
    else hipLaunchKernelGGL((k_pairs_split<8, false>), dim3(grid), dim3(kThreadsS), 0, stream, x);
    return hipGetLastError();
}

// List mode, two lanes per pair: the unmarked entries of a hand-over list (tables too large for the integer
// list stage) in exact integers; marked entries and what this kernel cannot answer go to the output list.
hipError_t launch_pairs_split_list(const PairKernelArgs &a, const SplitTables *st, const uint2 *in_list,
                                   const uint32_t *in_count, int n_cu, hipStream_t stream)
{
    SplitArgs x;
    x.st = st;
    x.c = a.c;
    x.pool = a.pool;
    x.cols_sorted = nullptr;
    x.perm = nullptr;
    x.k = a.k;
    x.row0 = a.row0;
    x.row1 = a.row1;
    x.col0 = a.col0;
    x.col1 = a.col1;
    x.sinks = a.sinks;
    x.ovf_list = a.overflow_list;
    x.ovf_count = a.overflow_count;
    x.ovf_cap = a.overflow_cap;
    x.reasons = nullptr;
    x.work_counter = a.work_counter;
    x.in_list = in_list;
    x.in_count = in_count;
    if (hipError_t e = hipMemsetAsync(a.work_counter, 0, sizeof(unsigned), stream); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_pairs_split<2, true>), dim3(n_cu > 0 ? n_cu : 256), dim3(kThreadsS), 0, stream, x);
    return hipGetLastError();
}

}  // namespace msspe
