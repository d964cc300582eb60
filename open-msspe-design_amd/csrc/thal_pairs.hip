// thal_pairs.hip -- the tuned all-pairs cross-dimer kernel (thal ANY for every ordered pair).
//
// Replaces the reference's hot loop "format N^2 lines -> ntthal -> parse"
// (/root/reference/od-msspe/src/delta_g.rs:61-153); one "check" = one ordered pair through Primer3
// 2.6.1 thal() ANY (fillMatrix / maxTM / calc_bulge_internal / terminal pick / traceback /
// drawDimer, restated from SURVEY.md Appendix C.3) down to the reference's conflict decision.
//
// CDNA4 design
//   * one lane = one ordered pair; one wave = one primer `a` (row) x 64 consecutive primers `b`;
//     a 256-thread block covers 4 rows x 64 columns and loops over tiles (persistent grid).
//   * the DP is SPARSE: only complementary cells (about 42 of 169 for random 13-mers) exist; they
//     are enumerated per lane in row-major order and numbered 0..n-1 ("slots").
//   * the per-pair DP table lives in VGPRs, not in an LDS image: slot s keeps S (f64) and a
//     packed word {H/10, predecessor context, i, j}.  All lanes walk slots in lock-step, so the
//     slot number is wave-uniform: a switch over the (uniform) chunk number copies 8 slots into
//     working registers with compile-time register numbers (24 v_mov per 8 predecessors).  Slots
//     40..55 overflow into a small LDS extension laid out [slot][thread] (conflict-free because
//     the slot is uniform).  No spills, 256 VGPRs, two 256-thread blocks per CU.
//   * every predecessor is evaluated in three phases (integer geometry -> five LDS gathers ->
//     f64 finish) and the scan is software-pipelined over groups of two predecessors, so LDS
//     latency overlaps the previous group's arithmetic.
//   * LDS holds the 2.6k-entry thermodynamic tables (fast_tables.hpp), gathered per lane.
//   * enthalpies are exact integers (checked on the host) and are summed in int32; entropies are
//     summed in f64 in Primer3's operation order (file built with -ffp-contract=off), so dS, dH,
//     dG and t are bit-identical to the CPU oracle and decisions are identical by construction.
//   * min-dG over predecessors is order independent except for exact ties, which are resolved by
//     Primer3's visiting order (key = loop size, then row distance).
//   * the columns of a launch are sorted by base composition (pool_sort.hip), which gives the 64
//     lanes of a wave equal DP sizes; conflicts (0.5 % of random pairs) leave the wave as one
//     atomic OR per conflicting pair at its ORIGINAL column and one popcount atomic per wave for
//     the per-row conflict count.
// Pairs whose DP has more complementary cells than the register table holds go to an overflow
// list and are finished by the wide instantiation of this kernel (list mode) or the generic kernel.
#include <cstring>

#include "pair_core.hpp"

namespace msspe {

namespace {

constexpr int kChunk = 8;
constexpr int kEmptySlot = 0xff;   // packed word of a slot that holds no cell yet: i-1 = j-1 = 15
constexpr int kInterleave = 2;   // predecessor evaluations the scheduler may overlap (register budget)

// The DP table of one lane.  Slots [0, NREG*8) live in VGPRs: every access uses a compile-time
// slot number (switch over the wave-uniform chunk number), so the arrays dissolve into registers.
// Slots beyond that (rarely reached once the columns are composition-sorted) live in an LDS
// extension laid out [slot][thread]: the slot number is wave-uniform, so those accesses are
// conflict-free ds_read/ds_write with a per-lane base.
template <int NREG, int NEXT>
struct Slots {
    static constexpr int kRegSlots = NREG * kChunk;
    static constexpr int kSlots = (NREG + NEXT) * kChunk;
    double S[NREG * kChunk];
    int W[NREG * kChunk];
    double *xS;   // LDS: &extS[0][threadIdx.x]
    int *xW;
    int xstride;  // threads per block (elements between consecutive extension slots)
};

template <int NEXT, int THREADS>
struct LdsExt {
    double S[NEXT * kChunk > 0 ? NEXT * kChunk : 1][THREADS];
    int W[NEXT * kChunk > 0 ? NEXT * kChunk : 1][THREADS];
};

// ---- register-table access: slot numbers are compile-time constants ---------------------------

// Fill: one predecessor slot against cell c.  `on` masks slots that are not computed yet.
__device__ __forceinline__ void fill_step(const Cand &k, const CellCtx &c, double Sp, int Wp,
                                          Best &best, double &stS, int &stH, bool &stHave)
{
    const double G1 = (double)(k.H + c.rH) - kT37 * (k.S + c.rS);
    const bool better = k.ok & ((G1 < best.G) | ((G1 == best.G) & (k.key < best.key)));
    best.G = better ? G1 : best.G;
    best.S = better ? k.S : best.S;
    best.H = better ? k.H : best.H;
    best.key = better ? k.key : best.key;
    const bool st = k.isStack;
    stS = st ? Sp : stS;
    stH = st ? (Wp >> 14) * 10 : stH;
    stHave = stHave | st;
}

struct TraceHit {
    unsigned key;
    double S;
    int H, im1, jm1, slot;
};

// Traceback: does predecessor slot p reproduce the current cell's value?
__device__ __forceinline__ void trace_step(const Cand &k, double Sp, int Wp, int p,
                                           double curS, int curH, int curSlot, double wcS, int wcH,
                                           TraceHit &h)
{
    const int Hp = (Wp >> 14) * 10;
    const double candS = k.isStack ? wcS + Sp : k.S;
    const int candH = k.isStack ? wcH + Hp : k.H;
    const unsigned key = k.isStack ? 0u : k.key;
    const bool hit = (p < curSlot) & (k.ok | k.isStack) & (candH == curH) &
                     (fabs(curS - candS) < 1e-5) & (key < h.key);
    h.key = hit ? key : h.key;
    h.S = hit ? Sp : h.S;
    h.H = hit ? Hp : h.H;
    h.im1 = hit ? k.iim1 : h.im1;
    h.jm1 = hit ? k.jjm1 : h.jm1;
    h.slot = hit ? p : h.slot;
}

// Copies chunk `pc` (wave-uniform, dynamic) of the table into 8 working registers.  The switch
// keeps every register-table access compile-time indexed; only the 24 v_mov of one case execute.
template <int NREG, int NEXT>
__device__ __forceinline__ void fetch_chunk(const Slots<NREG, NEXT> &st, int pc, double (&S)[kChunk],
                                            int (&W)[kChunk])
{
    if (NEXT > 0 && pc >= NREG) {
        const int base = (pc - NREG) * kChunk;
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            S[q] = st.xS[(base + q) * st.xstride];
            W[q] = st.xW[(base + q) * st.xstride];
        }
        return;
    }
#define MSSPE_FETCH(PC)                                                   \
    case PC:                                                              \
        if constexpr (PC < NREG) {                                        \
            _Pragma("unroll") for (int q = 0; q < kChunk; ++q) {          \
                S[q] = st.S[(PC < NREG ? PC : 0) * kChunk + q];           \
                W[q] = st.W[(PC < NREG ? PC : 0) * kChunk + q];           \
            }                                                             \
        }                                                                 \
        break;
    switch (pc) {
        MSSPE_FETCH(0) MSSPE_FETCH(1) MSSPE_FETCH(2) MSSPE_FETCH(3) MSSPE_FETCH(4) MSSPE_FETCH(5)
        MSSPE_FETCH(6) MSSPE_FETCH(7) MSSPE_FETCH(8) MSSPE_FETCH(9) MSSPE_FETCH(10) MSSPE_FETCH(11)
        MSSPE_FETCH(12) MSSPE_FETCH(13) MSSPE_FETCH(14) MSSPE_FETCH(15)
    default: break;
    }
#undef MSSPE_FETCH
}

template <int NREG, int NEXT>
__device__ __forceinline__ void scan_fill_all(const Slots<NREG, NEXT> &st, int upto, const Lds &T,
                                              const CellCtx &c, Best &best, double &stS, int &stH,
                                              bool &stHave)
{
    const int nch = (upto + kChunk - 1) / kChunk;   // wave-uniform
    for (int pc_ = 0; pc_ < nch; ++pc_) {
        const int pc = __builtin_amdgcn_readfirstlane(pc_);
        double S[kChunk];
        int W[kChunk];
        fetch_chunk<NREG, NEXT>(st, pc, S, W);
        // software pipeline over groups of kInterleave predecessors: the gathers of group n+1 are
        // in flight while group n is finished
        constexpr int kGroups = kChunk / kInterleave;
        CandGeom g[2][kInterleave];
        CandLoad v[2][kInterleave];
#pragma unroll
        for (int e = 0; e < kInterleave; ++e) g[0][e] = cand_geometry(c, W[e]);
#pragma unroll
        for (int e = 0; e < kInterleave; ++e) v[0][e] = cand_gather(T, g[0][e]);
#pragma unroll
        for (int grp = 0; grp < kGroups; ++grp) {
            const int q0 = grp * kInterleave;
            const int cur = grp & 1, nxt = cur ^ 1;
            const bool more = grp + 1 < kGroups && pc * kChunk + q0 + kInterleave < upto;   // wave-uniform
            if (more) {
#pragma unroll
                for (int e = 0; e < kInterleave; ++e) g[nxt][e] = cand_geometry(c, W[q0 + kInterleave + e]);
#pragma unroll
                for (int e = 0; e < kInterleave; ++e) v[nxt][e] = cand_gather(T, g[nxt][e]);
            }
#pragma unroll
            for (int e = 0; e < kInterleave; ++e) {
                const Cand k = cand_finish(g[cur][e], v[cur][e], S[q0 + e], W[q0 + e]);
                fill_step(k, c, S[q0 + e], W[q0 + e], best, stS, stH, stHave);
            }
            if (!more) break;
        }
    }
}

template <int NREG, int NEXT>
__device__ __forceinline__ void scan_trace_all(const Slots<NREG, NEXT> &st, int upto, const Lds &T,
                                               const CellCtx &c, double curS, int curH, int curSlot,
                                               double wcS, int wcH, TraceHit &h)
{
    const int nch = (upto + kChunk - 1) / kChunk;   // wave-uniform
    for (int pc_ = 0; pc_ < nch; ++pc_) {
        const int pc = __builtin_amdgcn_readfirstlane(pc_);
        double S[kChunk];
        int W[kChunk];
        fetch_chunk<NREG, NEXT>(st, pc, S, W);
#pragma unroll
        for (int q0 = 0; q0 < kChunk; q0 += kInterleave) {
            if (pc * kChunk + q0 >= upto) break;   // wave-uniform
            CandGeom g[kInterleave];
            CandLoad v[kInterleave];
#pragma unroll
            for (int e = 0; e < kInterleave; ++e) g[e] = cand_geometry(c, W[q0 + e]);
#pragma unroll
            for (int e = 0; e < kInterleave; ++e) v[e] = cand_gather(T, g[e]);
#pragma unroll
            for (int e = 0; e < kInterleave; ++e) {
                const Cand k = cand_finish(g[e], v[e], S[q0 + e], W[q0 + e]);
                trace_step(k, S[q0 + e], W[q0 + e], pc * kChunk + q0 + e, curS, curH, curSlot, wcS, wcH, h);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NREG, int NEXT, int PC = 0>
__device__ __forceinline__ void store_slot(Slots<NREG, NEXT> &st, int slot, double S, int W)
{
    if constexpr (PC < NREG) {
        if (slot < (PC + 1) * kChunk) {
            switch (slot - PC * kChunk) {
            case 0: st.S[PC * kChunk + 0] = S; st.W[PC * kChunk + 0] = W; break;
            case 1: st.S[PC * kChunk + 1] = S; st.W[PC * kChunk + 1] = W; break;
            case 2: st.S[PC * kChunk + 2] = S; st.W[PC * kChunk + 2] = W; break;
            case 3: st.S[PC * kChunk + 3] = S; st.W[PC * kChunk + 3] = W; break;
            case 4: st.S[PC * kChunk + 4] = S; st.W[PC * kChunk + 4] = W; break;
            case 5: st.S[PC * kChunk + 5] = S; st.W[PC * kChunk + 5] = W; break;
            case 6: st.S[PC * kChunk + 6] = S; st.W[PC * kChunk + 6] = W; break;
            default: st.S[PC * kChunk + 7] = S; st.W[PC * kChunk + 7] = W; break;
            }
        } else {
            store_slot<NREG, NEXT, PC + 1>(st, slot, S, W);
        }
    } else if constexpr (NEXT > 0) {
        st.xS[(slot - NREG * kChunk) * st.xstride] = S;
        st.xW[(slot - NREG * kChunk) * st.xstride] = W;
    }
}


// thal.c fillMatrix() for the lane's pair: the table of all complementary cells.  n_cells == 0 means "lane idle".
template <int NREG, int NEXT>
__device__ __forceinline__ void fill_pair(const Lds &T, const ThalConsts &K, const SeqPair &q, unsigned rowmask,
                                          int nmax, Slots<NREG, NEXT> &st)
{
    // Every slot starts as "not computed yet": coordinates (16,16) lie beyond any cell, so such a
    // slot fails the geometry test of every cell and needs no separate mask in the scans.
#pragma unroll
    for (int x = 0; x < NREG * kChunk; ++x) {
        st.S[x] = 0.0;
        st.W[x] = kEmptySlot;
    }
    if constexpr (NEXT > 0) {
#pragma unroll
        for (int x = 0; x < NEXT * kChunk; ++x) st.xW[x * st.xstride] = kEmptySlot;
    }
    CellCtx c;
    c.rS = 0.0;
    c.rH = 0;
    c.im1p = c.jm1p = 0;
    c.yTS = c.yMM = c.bBase = 0;
    unsigned Rrem = rowmask, mrem = 0;
    int im1 = 0, jm1 = 0;

    for (int slot_ = 0; slot_ < nmax; ++slot_) {
        const int slot = __builtin_amdgcn_readfirstlane(slot_);
        // ---- next complementary cell in row-major order
        const bool newrow = mrem == 0;
        const int t = __ffs((int)Rrem) - 1;
        const int a_new = (q.s1 >> (t & 31)) & 3;
        const unsigned m_new = spaced_mask(q.s2, 3 - a_new, q.lenmask);
        im1 = newrow ? (t >> 1) : im1;
        Rrem = newrow ? (Rrem & (Rrem - 1)) : Rrem;
        mrem = newrow ? m_new : mrem;
        jm1 = (__ffs((int)mrem) - 1) >> 1;
        mrem &= mrem - 1;
        im1 &= 15;
        jm1 &= 15;
        const CellBases b = cell_bases(q, im1, jm1, c);
        c.rS = T.S[b.idxR];
        c.rH = T.H[b.idxR];
        // ---- all earlier slots as predecessors
        Best best;
        best.G = INFINITY;
        best.S = 0.0;
        best.H = 0;
        best.key = 0xffffffffu;
        double stS = 0.0;
        int stH = 0;
        bool stHave = false;
        scan_fill_all<NREG, NEXT>(st, slot, T, c, best, stS, stH, stHave);
        // ---- thal.c maxTM(): helix extension if it raises Tm
        double S0 = T.S[b.idxL];
        int H0 = T.H[b.idxL];
        if (stHave) {
            const double T0 = (double)(H0 + 200 + c.rH) / (((S0 + K.init_S) + c.rS) + K.RC);
            const double S1 = stS + T.S[b.wc];
            const int H1 = stH + T.H[b.wc];
            const double T1 = (double)(H1 + 200 + c.rH) / (((S1 + K.init_S) + c.rS) + K.RC);
            if (T1 > T0) {
                S0 = S1;
                H0 = H1;
            }
        }
        // ---- loops (thal.c calc_bulge_internal acceptance: dG of the candidate strictly lower)
        const double G2 = (double)(H0 + c.rH) - kT37 * (S0 + c.rS);
        if (best.G < G2) {
            S0 = best.S;
            H0 = best.H;
        }
        // ---- publish the cell (idle lanes write a slot nobody reads)
        store_slot<NREG, NEXT>(st, slot, S0, ((H0 / 10) << 14) | (b.po_c << 8) | (im1 << 4) | jm1);
    }
}

// Terminal pick, traceback and totals over a filled table.  END1 (thal type 2, libprimer3's SELF_END: only
// structures that close on the 3' base of oligo 1, i.e. cells of the last row; without one thal() falls back to
// cell (1, 1) and reports a structure only if that cell is a base pair) differs from ANY in the pick alone, so a
// caller that wants both fills once and finishes twice (stage B, k_self_list).
template <int NREG, int NEXT, bool END1 = false>
__device__ __forceinline__ PairResult finish_pair(const Lds &T, const ThalConsts &K, const SeqPair &q, int n_cells,
                                                  int nmax, const Slots<NREG, NEXT> &st)
{
    CellCtx c;
    c.rS = 0.0;
    c.rH = 0;
    c.im1p = c.jm1p = 0;
    c.yTS = c.yMM = c.bBase = 0;
    PairResult r;
    r.none = n_cells == 0;
    r.dG = INFINITY;
    r.t = 0.0;
    r.conflict = false;

    // ---- terminal pick (thal.c thal(): strict minimum of the nudged dG over all cells, first in
    //      row-major order = slot order).  Done as one pass over the finished table so that no
    //      pick state is carried through the fill loop.
    double pickG = INFINITY, pickS = 0.0, pickRS = 0.0;
    int pickH = 0, pickRH = 0, pickI = 0, pickJ = 0, pickSlot = 0;
    {
        const int nch = (nmax + kChunk - 1) / kChunk;
        for (int pc_ = 0; pc_ < nch; ++pc_) {
            const int pc = __builtin_amdgcn_readfirstlane(pc_);
            double S[kChunk];
            int W[kChunk];
            fetch_chunk<NREG, NEXT>(st, pc, S, W);
#pragma unroll
            for (int e = 0; e < kChunk; ++e) {
                const int slot = pc * kChunk + e;
                const int ci = (W[e] >> 4) & 15, cj = W[e] & 15, H0 = (W[e] >> 14) * 10;
                CellCtx dummy;
                const CellBases b = cell_bases(q, ci, cj, dummy);
                const double rS = T.S[b.idxR];
                const int rH = T.H[b.idxR];
                const double rSn = rS + kTiny, rHn = (double)rH + kTiny;
                const double Gt = (((double)H0 + rHn) + K.init_H) - kT37 * ((S[e] + rSn) + K.init_S);
                const bool pick = (slot < n_cells) & (Gt < pickG) & (!END1 | (ci == q.len - 1));
                pickG = pick ? Gt : pickG;
                pickS = pick ? S[e] : pickS;
                pickH = pick ? H0 : pickH;
                pickRS = pick ? rS : pickRS;
                pickRH = pick ? rH : pickRH;
                pickI = pick ? ci : pickI;
                pickJ = pick ? cj : pickJ;
                pickSlot = pick ? slot : pickSlot;
            }
        }
    }

    if constexpr (END1) {
        // no cell in the last row: cell (1, 1) if it is a base pair (slot 0 then, the first in row-major order)
        double S[kChunk];
        int W[kChunk];
        fetch_chunk<NREG, NEXT>(st, 0, S, W);
        const bool have = pickG < INFINITY;
        const bool first = !have & (n_cells > 0) & ((W[0] & 0xff) == 0);
        CellCtx dummy;
        const CellBases b0 = cell_bases(q, 0, 0, dummy);
        pickS = first ? S[0] : pickS;
        pickH = first ? (W[0] >> 14) * 10 : pickH;
        pickRS = first ? T.S[b0.idxR] : pickRS;
        pickRH = first ? T.H[b0.idxR] : pickRH;
        pickI = first ? 0 : pickI;
        pickJ = first ? 0 : pickJ;
        pickSlot = first ? 0 : pickSlot;
        r.none = r.none | !(have | first);
    }
    // ---- thal.c traceback(): count the base pairs of the optimal structure
    double curS = pickS;
    int curH = pickH, curI = pickI, curJ = pickJ, curSlot = pickSlot, P = 1;
    bool done = r.none;
    for (int step = 0; step < 2 * 16 + 2; ++step) {
        if (__builtin_amdgcn_readfirstlane((int)__all(done))) break;
        const CellBases b = cell_bases(q, curI, curJ, c);
        const double leftS = T.S[b.idxL];
        const int leftH = T.H[b.idxL];
        done = done | ((leftH == curH) & (fabs(curS - leftS) < 1e-5));
        const int bound = wave_max(done ? 0 : curSlot);
        TraceHit h;
        h.key = 0xffffffffu;
        h.S = 0.0;
        h.H = h.im1 = h.jm1 = h.slot = 0;
        scan_trace_all<NREG, NEXT>(st, bound, T, c, curS, curH, curSlot, T.S[b.wc], T.H[b.wc], h);
        const bool moved = !done & (h.key != 0xffffffffu);
        done = done | !moved;
        curS = moved ? h.S : curS;
        curH = moved ? h.H : curH;
        curI = moved ? h.im1 : curI;
        curJ = moved ? h.jm1 : curJ;
        curSlot = moved ? h.slot : curSlot;
        P += moved ? 1 : 0;
    }
    // ---- thal.c drawDimer(): totals
    const double dH = (double)(pickH + pickRH + 200);
    const double dS = (pickS + pickRS) + K.init_S;
    const int N = P - 1;
    const double t = (dH / ((dS + (N * K.salt)) + K.RC)) - kAbsZero;
    const double G = dH - (K.temp_k * (dS + (N * K.salt)));
    if (!r.none) {
        r.dG = G;
        r.t = t;
        r.conflict = G <= K.g_cut;
    }
    return r;
}

// The whole thal ANY computation for the lane's pair.
template <int NREG, int NEXT>
__device__ __forceinline__ PairResult run_pair(const Lds &T, const ThalConsts &K, const SeqPair &q,
                                               unsigned rowmask, int n_cells, int nmax,
                                               Slots<NREG, NEXT> &st)
{
    fill_pair<NREG, NEXT>(T, K, q, rowmask, nmax, st);
    return finish_pair<NREG, NEXT, false>(T, K, q, n_cells, nmax, st);
}

__device__ __forceinline__ void load_tables(Lds &T, const FastTables *ft)
{
    for (int e = threadIdx.x; e < FastTables::kCount; e += blockDim.x) {
        T.S[e] = ft->S[e];
        T.H[e] = ft->H[e];
    }
    __syncthreads();
}

// Matrix mode: wave = one row x 64 consecutive entries of the composition-sorted column list.
// col0/col1 index that sorted list; perm[] maps an entry back to its pool index for the outputs.
template <int NREG, int NEXT, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_pairs_fast(FastArgs a)
{
    // one LDS object with the tables FIRST: their byte offsets stay below 64 KB, so every table
    // gather folds its array base into the ds_read immediate offset
    struct Shared {
        Lds T;
        LdsExt<NEXT, 256> X;
    };
    __shared__ Shared sh;
    Lds &T = sh.T;
    load_tables(T, a.ft);
    Slots<NREG, NEXT> st;
    st.xS = &sh.X.S[0][threadIdx.x];
    st.xW = &sh.X.W[0][threadIdx.x];
    st.xstride = 256;
    constexpr int NCH = NREG + NEXT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncolg = (a.col1 - a.col0 + 63) >> 6;
    const int nrowg = (a.row1 - a.row0 + 3) >> 2;
    const long tiles = (long)ncolg * nrowg;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int rg = (int)(tile / ncolg), cg = (int)(tile % ncolg);
        const int row = a.row0 + rg * 4 + wave;
        const int cq = a.col0 + cg * 64 + lane;
        if (row >= a.row1) continue;   // wave-uniform
        const bool inside = cq < a.col1;
        const uint64_t pa = a.pool[row];
        const uint64_t pb = a.cols_sorted[inside ? cq : a.col0];
        const int col = (int)a.perm[inside ? cq : a.col0];
        SeqPair q;
        unsigned rowmask;
        int n_cells = setup_pair(pa, pb, a.k, q, rowmask);
        const bool sym = self_complementary(pa, a.k) && self_complementary(pb, a.k);
        const bool spill = inside & ((n_cells > NCH * kChunk) | sym);
        if (spill) {
            const uint32_t at = atomicAdd(a.ovf_count, 1u);
            if (at < a.ovf_cap) a.ovf_list[at] = make_uint2((unsigned)row, (unsigned)col);
        }
        if (!inside | spill) n_cells = 0;
        const int nmax = wave_max(n_cells);
        const PairResult r = run_pair<NREG, NEXT>(T, a.c, q, rowmask, n_cells, nmax, st);
        // ---- sinks (conflicts are rare: one atomic OR per conflicting pair, one add per wave)
        const bool live = inside & !spill;
        const bool hit = live & r.conflict;
        const unsigned long long bits = __ballot(hit);
        const size_t orow = (size_t)(row - a.sinks.row0);
        const size_t ocol = (size_t)(col - a.sinks.col0);
        if (hit && a.sinks.bitmap)
            atomicOr((unsigned long long *)&a.sinks.bitmap[orow * (size_t)a.sinks.words + (ocol >> 6)],
                     1ull << (ocol & 63));
        if (hit) sink_edge(a.sinks, row, col, r.dG);
        if (lane == 0 && a.sinks.row_conflicts && bits)
            atomicAdd(&a.sinks.row_conflicts[row], (unsigned)__popcll(bits));
        if (live) {
            if (a.sinks.dg) a.sinks.dg[orow * (size_t)a.sinks.ncols + ocol] = r.dG;
            if (a.sinks.tm) a.sinks.tm[orow * (size_t)a.sinks.ncols + ocol] = r.t;
        }
    }
}

// List mode: lane = one explicit pair (the list a previous stage left behind).  THREADS-wide
// blocks: 40 register slots + NEXT * 8 LDS slots per lane, two blocks per CU.
// SELF (stage B, od-msspe/src/primer.rs:143-166: PRIMER_LEFT_0_SELF_ANY_TH / SELF_END_TH of libprimer3's
// oligo_compl_thermod): the entries are (i, i); ONE fill serves thal ANY and thal END1, each finished on its own
// (finish_pair), self_any[i] / self_end[i] = max(0, t) (either may be null).
template <int NREG, int NEXT, int THREADS, bool SELF>
__device__ __forceinline__ void list_body(const FastArgs &a, double *self_any, double *self_end)
{
    struct Shared {
        Lds T;
        LdsExt<NEXT, THREADS> X;
    };
    __shared__ Shared sh;
    Lds &T = sh.T;
    if (*a.in_count == 0u) return;   // an empty list: no table loads
    load_tables(T, a.ft);
    Slots<NREG, NEXT> st;
    st.xS = &sh.X.S[0][threadIdx.x];
    st.xW = &sh.X.W[0][threadIdx.x];
    st.xstride = THREADS;
    constexpr int NCH = NREG + NEXT;
    // Lock-step lanes pay for the largest table of their wave, and a list arrives in no useful
    // order: every block takes kListBatch * THREADS entries at a time and counting-sorts them by
    // table size in LDS (the slot extension doubles as scratch), so that the 64 lanes of a wave
    // get neighbours of the sorted batch.
    constexpr int kListBatch = 4;
    static_assert(sizeof(sh.X) >= sizeof(uint2) * kListBatch * THREADS + sizeof(unsigned) * 256,
                  "the slot extension must hold one sorted batch");
    uint2 *sorted = reinterpret_cast<uint2 *>(&sh.X);
    unsigned *hist = reinterpret_cast<unsigned *>(sorted + kListBatch * THREADS);
    const long n_work = (long)(unsigned)__builtin_amdgcn_readfirstlane((int)min(*a.in_count, a.ovf_cap));
    // a short list is spread over the blocks (one pass each) instead of being run four passes deep by a few of them
    const long per_pass = (long)gridDim.x * THREADS;
    const int depth = (int)min((long)kListBatch, max(1L, (n_work + per_pass - 1) / per_pass));
    const long batch = (long)depth * THREADS;
    const long n_batches = (n_work + batch - 1) / batch;
    for (long bt = blockIdx.x; bt < n_batches; bt += gridDim.x) {
        // ---- 1. keys and histogram
        uint2 mine[kListBatch];
        int key[kListBatch];
        __threadfence_block();
        for (int e = threadIdx.x; e < 256; e += THREADS) hist[e] = 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kListBatch; ++j) {
            const long w = bt * batch + (long)j * THREADS + threadIdx.x;
            const bool inside = (j < depth) & (w < n_work);   // passes beyond the depth hold padding only
            mine[j] = inside ? a.in_list[w] : make_uint2(0xffffffffu, 0u);
            int nc = 255;   // padding entries sort last
            if (inside) {
                SeqPair q;
                unsigned rowmask;
                nc = min(setup_pair(a.pool[mine[j].x & 0x7fffffffu], a.pool[mine[j].y], a.k, q, rowmask), 254);
            }
            key[j] = nc;
            atomicAdd(&hist[nc], 1u);
        }
        __syncthreads();
        // ---- 2. exclusive scan of the 256 bins (one wave)
        if (threadIdx.x < 64) {
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
        // ---- 3. scatter into the sorted batch and write it back over the batch's own part of the
        //         list (this block is its only reader), so that nothing is held in registers
        //         across the DP
#pragma unroll
        for (int j = 0; j < kListBatch; ++j) sorted[atomicAdd(&hist[key[j]], 1u)] = mine[j];
        __syncthreads();
        uint2 *own = const_cast<uint2 *>(a.in_list) + bt * batch;
        const long n_own = min(batch, n_work - bt * batch);
#pragma unroll
        for (int j = 0; j < kListBatch; ++j) {
            const long e = (long)j * THREADS + threadIdx.x;
            if (e < n_own) own[e] = sorted[e];   // padding entries sorted last
        }
        __syncthreads();   // the extension is the DP's again; the writes are visible to the block
        // ---- 4. the pairs
        for (int j = 0; j < depth; ++j) {
            const long e = (long)j * THREADS + threadIdx.x;
            const bool inside = e < n_own;
            uint2 pr = own[inside ? e : 0];
            pr.x &= 0x7fffffffu;   // bit 31: "needs the f64 kernels" mark of the integer stage
            const uint64_t pa = a.pool[inside ? pr.x : 0], pb = a.pool[inside ? pr.y : 0];
            SeqPair q;
            unsigned rowmask;
            int n_cells = setup_pair(pa, pb, a.k, q, rowmask);
            const bool sym = self_complementary(pa, a.k) && self_complementary(pb, a.k);
            const bool spill = inside & ((n_cells > NCH * kChunk) | sym);
            if (spill) {
                const uint32_t at = atomicAdd(a.ovf_count, 1u);
                if (at < a.ovf_cap) a.ovf_list[at] = pr;
            }
            if (!inside | spill) n_cells = 0;
            const int nmax = wave_max(n_cells);
            if (nmax == 0) continue;   // wave-uniform: nothing but padding / handed-on pairs
            if constexpr (SELF) {
                fill_pair<NREG, NEXT>(T, a.c, q, rowmask, nmax, st);
                if (self_any) {   // kernel argument: uniform
                    const PairResult r = finish_pair<NREG, NEXT, false>(T, a.c, q, n_cells, nmax, st);
                    if (inside & !spill) self_any[pr.x] = (r.none || r.t < 0.0) ? 0.0 : r.t;   // libprimer3 align_thermod()
                }
                if (self_end) {
                    const PairResult r = finish_pair<NREG, NEXT, true>(T, a.c, q, n_cells, nmax, st);
                    if (inside & !spill) self_end[pr.x] = (r.none || r.t < 0.0) ? 0.0 : r.t;
                }
                continue;
            }
            const PairResult r = run_pair<NREG, NEXT>(T, a.c, q, rowmask, n_cells, nmax, st);
            if (inside & !spill) {
                const size_t orow = (size_t)((int)pr.x - a.sinks.row0);
                const size_t ocol = (size_t)((int)pr.y - a.sinks.col0);
                if (a.sinks.dg) a.sinks.dg[orow * (size_t)a.sinks.ncols + ocol] = r.dG;
                if (a.sinks.tm) a.sinks.tm[orow * (size_t)a.sinks.ncols + ocol] = r.t;
                if (r.conflict) {
                    if (a.sinks.row_conflicts) atomicAdd(&a.sinks.row_conflicts[pr.x], 1u);
                    if (a.sinks.bitmap)
                        atomicOr((unsigned long long *)&a.sinks.bitmap[orow * (size_t)a.sinks.words + (ocol >> 6)],
                                 1ull << (ocol & 63));
                    sink_edge(a.sinks, (int)pr.x, (int)pr.y, r.dG);
                }
            }
        }
        __syncthreads();   // before the next batch reuses the scratch
    }
}

template <int NREG, int NEXT, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 128) k_pairs_list(FastArgs a)
{
    list_body<NREG, NEXT, THREADS, false>(a, nullptr, nullptr);
}

template <int NREG, int NEXT, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 128) k_self_list(FastArgs a, double *self_any, double *self_end)
{
    list_body<NREG, NEXT, THREADS, true>(a, self_any, self_end);
}

// stage B's work list: entry i = (i, i)
__global__ void k_self_entries(uint2 *list, uint32_t *count, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) list[i] = make_uint2((unsigned)i, (unsigned)i);
    if (i == 0) *count = (uint32_t)n;
}

constexpr int kNregMain = 5;   // 40 slots in VGPRs ...
constexpr int kNextMain = 2;   // ... + 16 slots in LDS = 56: 98.8 % of random 13-mer pairs, 2 waves / SIMD
constexpr int kNchMain = kNregMain + kNextMain;
constexpr int kNregWide = 5;   // overflow list: 40 slots in VGPRs ...
constexpr int kNextWide = 4;   // ... + 32 in LDS = 72 (all but 0.012 % of random 13-mer pairs)
constexpr int kNchWide = kNregWide + kNextWide;

}  // namespace

int pairs_fast_max_k() { return 16; }
int pairs_fast_main_slots() { return kNchMain * kChunk; }
int pairs_fast_wide_slots() { return kNchWide * kChunk; }

hipError_t launch_pairs_fast(const PairKernelArgs &a, hipStream_t stream)
{
    FastArgs f;
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
    const long tiles = (long)((a.col1 - a.col0 + 63) / 64) * (long)((a.row1 - a.row0 + 3) / 4);
    if (tiles <= 0) return hipSuccess;
    const int grid = (int)(tiles < 256L * 8 ? tiles : 256L * 8);
    hipLaunchKernelGGL((k_pairs_fast<kNregMain, kNextMain, 2>), dim3(grid), dim3(256), 0, stream, f);
    return hipGetLastError();
}

namespace {
FastArgs list_args(const PairKernelArgs &a, const uint2 *in_list, const uint32_t *in_count)
{
    FastArgs f;
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
    return f;
}
}  // namespace

// The main table (56 slots) over an explicit pair list: finishes what the integer stage handed on.
hipError_t launch_pairs_main_list(const PairKernelArgs &a, const uint2 *in_list,
                                  const uint32_t *in_count, hipStream_t stream)
{
    const FastArgs f = list_args(a, in_list, in_count);
    hipLaunchKernelGGL((k_pairs_list<kNregMain, kNextMain, 256>), dim3(256 * 2), dim3(256), 0, stream, f);
    return hipGetLastError();
}

// Stage B, one lane per oligo: the f64 register-table kernels over the list of (i, i), n oligos.  list_a / list_b:
// two lists of at least n entries, counters[0 .. 3): device words (zeroed here).  What is left -- self-complementary
// oligos, tables beyond 72 cells -- ends up in list_a with counters[2] entries, for launch_self_wave.
hipError_t launch_self_lists(const FastTables *ft, const ThalConsts &c, const uint64_t *pool, int n, int k,
                             double *self_any, double *self_end, uint2 *list_a, uint2 *list_b, uint32_t *counters,
                             uint32_t list_cap, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (hipError_t e = hipMemsetAsync(counters, 0, 3 * sizeof(uint32_t), stream); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_self_entries, dim3((n + 255) / 256), dim3(256), 0, stream, list_a, counters, n);
    PairKernelArgs a;
    std::memset(&a, 0, sizeof a);
    a.ft = ft;
    a.c = c;
    a.pool = pool;
    a.k = k;
    a.overflow_cap = list_cap;
    a.overflow_list = list_b;
    a.overflow_count = counters + 1;
    FastArgs f = list_args(a, list_a, counters);
    hipLaunchKernelGGL((k_self_list<kNregMain, kNextMain, 256>), dim3(256 * 2), dim3(256), 0, stream, f, self_any, self_end);
    a.overflow_list = list_a;
    a.overflow_count = counters + 2;
    f = list_args(a, list_b, counters + 1);
    hipLaunchKernelGGL((k_self_list<kNregWide, kNextWide, 128>), dim3(256 * 4), dim3(128), 0, stream, f, self_any, self_end);
    return hipGetLastError();
}

hipError_t launch_pairs_wide(const PairKernelArgs &a, const uint2 *in_list,
                             const uint32_t *in_count, hipStream_t stream)
{
    FastArgs f;
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
    hipLaunchKernelGGL((k_pairs_list<kNregWide, kNextWide, 128>), dim3(256 * 4), dim3(128), 0, stream, f);
    return hipGetLastError();
}

}  // namespace msspe
