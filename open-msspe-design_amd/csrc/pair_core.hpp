// pair_core.hpp -- device code shared by the all-pairs kernels (thal_pairs.hip: f64 tables in
// registers; thal_pairs_int.hip: exact integer DP + f64 replay of the optimal path).
// Packed-sequence helpers, the per-cell table indices and the one-formula loop candidate of
// fast_tables.hpp.  Primer3 2.6.1 thal.c restated from SURVEY.md Appendix C.3; the reference call
// site is /root/reference/od-msspe/src/delta_g.rs:61-153.
#pragma once

#include "fast_tables.hpp"
#include "kernels.hpp"
#include "thal_dense.hpp"

namespace msspe {

namespace {

struct Lds {
    double S[FastTables::kCount];
    int H[FastTables::kCount];
};

struct CellCtx {
    int im1p, jm1p;          // cell coordinates minus one: l1 = im1p - ii, l2 = jm1p - jj (0-based)
    int yTS, yMM, bBase;     // table indices that depend on the cell only
    double rS;               // right end term of the cell
    int rH;
};

struct Cand {
    double S;
    int H;
    bool ok;       // a valid, finite loop candidate
    bool isStack;  // predecessor is (i-1, j-1)
    unsigned key;  // Primer3's visiting order among loop candidates (smaller = earlier)
    int iim1, jjm1;
};

// Candidate value of the loop (or stack) between predecessor slot (Sp, Wp) and cell c, in three
// phases so that a group of predecessors can issue all its LDS gathers before any is consumed
// (one s_waitcnt per group instead of three per predecessor):
//   geometry  -> table indices          (integer VALU)
//   gather    -> five LDS reads         (ds_read)
//   finish    -> sums, rejection tests  (f64 VALU)
// One formula for every kind of loop (fast_tables.hpp); lanes with an impossible geometry read
// clamped table entries and are masked by `ok`.
struct CandGeom {
    int l1, l2, t, sz;
    unsigned lx;
    int y, zi;
    int iim1, jjm1;
};
struct CandLoad {
    double sLX, sY, sZ;
    int hLX, hY;
};

__device__ __forceinline__ CandGeom cand_geometry(const CellCtx &c, int Wp)
{
    CandGeom g;
    g.jjm1 = Wp & 15;
    g.iim1 = (Wp >> 4) & 15;
    const int po = (Wp >> 8) & 63;
    g.l1 = c.im1p - g.iim1;
    g.l2 = c.jm1p - g.jjm1;
    g.sz = g.l1 + g.l2;
    g.t = min(g.l1, g.l2);
    const bool bulge = g.t == 0;
    const int lxN = g.sz * 64 + po + (FastTables::kNB - 2 * 64);
    const int lxB = g.sz * 4 + (po & 3) + c.bBase;
    g.lx = min((unsigned)(bulge ? lxB : lxN), (unsigned)(FastTables::kCount - 1));
    const bool m11 = ((g.l1 << 4) | g.l2) == 0x11;
    g.y = bulge ? FastTables::kZero : (m11 ? c.yMM : c.yTS);
    g.zi = bulge ? FastTables::kZero : (g.l1 - g.l2 + (FastTables::kZT + 32));
    return g;
}

__device__ __forceinline__ CandLoad cand_gather(const Lds &T, const CandGeom &g)
{
    CandLoad v;
    v.sLX = T.S[g.lx];
    v.sY = T.S[g.y];
    v.sZ = T.S[g.zi];
    v.hLX = T.H[g.lx];
    v.hY = T.H[g.y];
    return v;
}

__device__ __forceinline__ Cand cand_finish(const CandGeom &g, const CandLoad &v, double Sp, int Wp)
{
    Cand r;
    r.iim1 = g.iim1;
    r.jjm1 = g.jjm1;
    r.S = ((v.sLX + v.sY) + v.sZ) + Sp;
    r.H = v.hLX + v.hY + (Wp >> 14) * 10;
    const bool bad = (r.H > 0) & (r.S > 0.0);   // also true for unavailable table entries
    r.isStack = (g.l1 | g.l2) == 0;
    r.ok = (g.t >= 0) & !bad;
    r.key = (unsigned)(g.sz * 32 + g.l1);
    return r;
}

struct Best {
    double G, S;
    int H;
    unsigned key;
};

struct SeqPair {
    unsigned s1, s2;     // 2 bits per base; s2 = oligo 2 reversed
    unsigned lenmask;
    int len;
};

// Bases around cell (im1, jm1) and every table index that depends on the cell only.
struct CellBases {
    int a, idxL, idxR, wc, po_c;
};
__device__ __forceinline__ CellBases cell_bases(const SeqPair &q, int im1, int jm1, CellCtx &c)
{
    CellBases b;
    const int t1 = 2 * im1, t2 = 2 * jm1;
    b.a = (q.s1 >> t1) & 3;
    const int oaL = im1 > 0 ? (int)((q.s1 >> (t1 - 2)) & 3) : 4;
    const int oaR = im1 < q.len - 1 ? (int)((q.s1 >> (t1 + 2)) & 3) : 4;
    const int obL = jm1 > 0 ? (int)((q.s2 >> (t2 - 2)) & 3) : 4;
    const int obR = jm1 < q.len - 1 ? (int)((q.s2 >> (t2 + 2)) & 3) : 4;
    b.idxL = FastTables::kEndL + b.a * 25 + oaL * 5 + obL;
    b.idxR = FastTables::kEndR + b.a * 25 + oaR * 5 + obR;
    const int ci = (((3 - b.a) * 4 + (obL & 3)) * 4 + (oaL & 3)) & 63;
    b.wc = FastTables::kWC + (oaL & 3) * 4 + b.a;
    b.po_c = b.a | ((oaR & 3) << 2) | ((obR & 3) << 4);
    c.im1p = im1 - 1;
    c.jm1p = jm1 - 1;
    c.yTS = FastTables::kTSc + ci;
    c.yMM = FastTables::kMMc + ci;
    c.bBase = FastTables::kBU + b.a * FastTables::kBUStride;
    return b;
}

__device__ __forceinline__ int wave_max(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ unsigned spaced_mask(unsigned s, int base, unsigned lenmask)
{
    // bit 2p set iff the 2-bit field p of s equals base
    const unsigned x = s ^ (unsigned)(base * 0x55555555u);
    return ~(x | (x >> 1)) & 0x55555555u & lenmask;
}

__device__ __forceinline__ unsigned reverse2(unsigned s, int len)
{
    // reverse the order of the len 2-bit fields
    unsigned r = __brev(s);                                   // bit reversal
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);  // restore bit order inside fields
    return r >> (32 - 2 * len);
}

struct PairResult {
    double dG, t;
    bool none, conflict;
};

__device__ __forceinline__ int setup_pair(uint64_t pa, uint64_t pb, int k, SeqPair &q,
                                          unsigned &rowmask)
{
    const unsigned lenmask = k == 16 ? 0xffffffffu : ((1u << (2 * k)) - 1u);
    q.len = k;
    q.lenmask = lenmask;
    q.s1 = (unsigned)pa & lenmask;
    q.s2 = reverse2((unsigned)pb & lenmask, k);
    int n_cells = 0;
    rowmask = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const unsigned m1 = spaced_mask(q.s1, x, lenmask);
        const unsigned m2 = spaced_mask(q.s2, 3 - x, lenmask);
        n_cells += __popc(m1) * __popc(m2);
        rowmask |= m2 ? m1 : 0u;
    }
    return n_cells;
}

struct FastArgs {
    const FastTables *ft;
    ThalConsts c;
    const uint64_t *pool;
    const uint64_t *cols_sorted;
    const uint32_t *perm;
    int k;
    int row0, row1, col0, col1;   // tile range of this launch (matrix mode)
    PairSinks sinks;
    uint2 *ovf_list;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
    const uint2 *in_list;         // list mode: explicit pairs
    const uint32_t *in_count;
};

}  // namespace

}  // namespace msspe
