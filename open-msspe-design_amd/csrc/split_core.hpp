// split_core.hpp -- device code shared by the long-oligo kernels (thal_pairs_split.hip: exact-integer
// DP with the table split over lanes; thal_pairs_wave.hip: f64 DP, one wave per pair): 64-bit packed
// sequences, 5-bit cell coordinates, the per-cell table indices of split_tables.hpp.
// Primer3 2.6.1 thal.c restated from SURVEY.md Appendix C.3; reference call site
// /root/reference/od-msspe/src/delta_g.rs:61-153.
#pragma once

#include "kernels.hpp"
#include "split_tables.hpp"
#include "thal_dense.hpp"

namespace msspe {

namespace {

typedef SplitTables W_;

struct SeqW {
    unsigned long long s1, s2, lenmask;   // 2 bits per base; s2 = oligo 2 reversed
    int len;
};

__device__ __forceinline__ unsigned long long spaced_mask64(unsigned long long s, int base,
                                                            unsigned long long lenmask)
{
    const unsigned long long x = s ^ ((unsigned long long)base * 0x5555555555555555ull);
    return ~(x | (x >> 1)) & 0x5555555555555555ull & lenmask;
}

__device__ __forceinline__ unsigned long long reverse2_64(unsigned long long s, int len)
{
    unsigned long long r = __brevll(s);
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    return r >> (64 - 2 * len);
}

__device__ __forceinline__ int setup_pair_w(uint64_t pa, uint64_t pb, int k, SeqW &q,
                                            unsigned long long &rowmask)
{
    const unsigned long long lenmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    q.len = k;
    q.lenmask = lenmask;
    q.s1 = pa & lenmask;
    q.s2 = reverse2_64(pb & lenmask, k);
    int n_cells = 0;
    rowmask = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const unsigned long long m1 = spaced_mask64(q.s1, x, lenmask);
        const unsigned long long m2 = spaced_mask64(q.s2, 3 - x, lenmask);
        n_cells += __popcll(m1) * __popcll(m2);
        rowmask |= m2 ? m1 : 0ull;
    }
    return n_cells;
}

// Bases around cell (im1, jm1) and every table index that depends on the cell only.
struct CellS {
    int a, idxL, idxR, wc, po_c;
    int yTS, yMM, bBase;
    int im1p, jm1p;
};
__device__ __forceinline__ CellS cell_s(const SeqW &q, int im1, int jm1)
{
    CellS b;
    const int t1 = 2 * im1, t2 = 2 * jm1;
    b.a = (int)((q.s1 >> t1) & 3);
    const int oaL = im1 > 0 ? (int)((q.s1 >> ((t1 - 2) & 63)) & 3) : 4;
    const int oaR = im1 < q.len - 1 ? (int)((q.s1 >> ((t1 + 2) & 63)) & 3) : 4;
    const int obL = jm1 > 0 ? (int)((q.s2 >> ((t2 - 2) & 63)) & 3) : 4;
    const int obR = jm1 < q.len - 1 ? (int)((q.s2 >> ((t2 + 2) & 63)) & 3) : 4;
    b.idxL = W_::kEndL + b.a * 25 + oaL * 5 + obL;
    b.idxR = W_::kEndR + b.a * 25 + oaR * 5 + obR;
    const int ci = (((3 - b.a) * 4 + (obL & 3)) * 4 + (oaL & 3)) & 63;
    b.wc = W_::kWC + (oaL & 3) * 4 + b.a;
    b.po_c = b.a | ((oaR & 3) << 2) | ((obR & 3) << 4);
    b.im1p = im1 - 1;
    b.jm1p = jm1 - 1;
    b.yTS = W_::kTSc + ci;
    b.yMM = W_::kMMc + ci;
    b.bBase = W_::kBU + b.a * W_::kBUStride;
    return b;
}

// The f64 / exact-H terms of the loop closed by predecessor word Wp (po << 10 | im1 << 5 | jm1)
// and the cell: indices as in pair_core.hpp's cand_geometry, 5-bit coordinates.
struct LoopIx {
    int l1, l2;
    unsigned lx;
    int y, zi;
};
__device__ __forceinline__ LoopIx loop_indices(const CellS &c, int Wp)
{
    LoopIx g;
    const int jj = Wp & 31, ii = (Wp >> 5) & 31, po = (Wp >> 10) & 63;
    g.l1 = c.im1p - ii;
    g.l2 = c.jm1p - jj;
    const int sz = g.l1 + g.l2;
    const bool bulge = min(g.l1, g.l2) == 0;
    const int lxN = sz * 64 + po + (W_::kNB - 2 * 64);
    const int lxB = sz * 4 + (po & 3) + c.bBase;
    g.lx = min((unsigned)(bulge ? lxB : lxN), (unsigned)(W_::kCount - 1));
    const bool m11 = (g.l1 == 1) & (g.l2 == 1);
    g.y = bulge ? W_::kZero : (m11 ? c.yMM : c.yTS);
    g.zi = bulge ? W_::kZero : (g.l1 - g.l2 + (W_::kZT + 32));
    return g;
}

}  // namespace

}  // namespace msspe
