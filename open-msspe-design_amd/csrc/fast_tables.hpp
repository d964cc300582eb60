// fast_tables.hpp -- LDS-resident tables of the tuned all-pairs kernel (thal_pairs.hip).
//
// One flat array of entries (S: double, H: int32, same index in both planes).  The candidate of a
// bulge / interior loop closed by predecessor p and cell c is, in Primer3's operation order
// (thal.c calc_bulge_internal):
//      S = (((L + X) + Y) + Z) + S_p        H = L_H + X_H + Y_H + H_p
// and (L + X) depends only on (loop size, predecessor context), so it is folded on the host with
// the very same IEEE-754 addition the kernel would execute:
//      kind       LX = fl(L + X)                         Y                     Z
//      interior   interior[sz-1] + tstack[po]            tstack[ci]            ILAS * |l1 - l2|
//      bulge >=2  bulge[sz-1]    + AT(a_p)               AT(a_c)               0
//      bulge 1    bulge[0]       + stack[a_p][a_c] (*)   0                     0
//      1 x 1      stackmm[po]                            stackmm[ci]           0
//  (*) thal.c rejects a single-base bulge whose own (S,H) has H > 0 or S > 0 before the
//      predecessor is added; that test is applied here once per table entry.
// po = a_p | s1[ii+1] << 2 | s2[jj+1] << 4 ; ci = (s2[j] * 4 + s2[j-1]) * 4 + s1[i-1].
#pragma once

#include <cstdint>

#include "nn_params.hpp"

namespace msspe {

struct FastTables {
    static constexpr int kLxI = 0;                 // [sz-1][po]      30 * 64
    static constexpr int kLxB = kLxI + 30 * 64;    // [sz-1][a_p]     30 * 4
    static constexpr int kLxB1 = kLxB + 30 * 4;    // [a_c][a_p]      16
    static constexpr int kMM = kLxB1 + 16;         // [idx]           64   (table[x][y][3-x][z], idx = x + 4y + 16z for po; see below)
    static constexpr int kTS = kMM + 64;           // [idx]           64
    static constexpr int kMMc = kTS + 64;          // [ci]            64   same numbers, cell-side index order
    static constexpr int kTSc = kMMc + 64;         // [ci]            64
    static constexpr int kAT = kTSc + 64;          // [a]             4
    static constexpr int kZero = kAT + 4;          // 1 (+3 pad)
    static constexpr int kEndL = kZero + 4;        // [a*25 + oa*5 + ob]  100
    static constexpr int kEndR = kEndL + 100;      // 100
    static constexpr int kWC = kEndR + 100;        // [x*4 + y]       16
    static constexpr int kCount = kWC + 16;
    double S[kCount];
    int32_t H[kCount];
    int32_t usable;      // 1 when the kernel's exactness preconditions hold (see build_fast_tables)
    int32_t max_k;       // largest oligo length the preconditions were verified for
};

// Preconditions checked here (else usable = 0 and the caller must use the generic kernel):
//   * every finite enthalpy is an integer multiple of 10 cal/mol (packed as H/10 in 18 bits);
//   * no accepted value can reach the MinEntropyCutoff clamp for oligos up to max_k bases.
bool build_fast_tables(const NNTables &t, const PairTables &pt, int max_k, FastTables &out);

}  // namespace msspe
