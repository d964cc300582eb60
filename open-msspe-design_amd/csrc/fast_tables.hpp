// fast_tables.hpp -- LDS-resident tables of the tuned all-pairs kernel (thal_pairs.hip).
//
// One flat array of entries (S: double, H: int32, same index in both planes).  The candidate of a
// bulge / interior loop closed by predecessor p and cell c is, in Primer3's operation order
// (thal.c calc_bulge_internal):
//      S = (((L + X) + Y) + Z) + S_p        H = L_H + X_H + Y_H + H_p
// Everything that does not depend on BOTH p and c is folded on the host with the very same
// IEEE-754 additions, in the same order, that the kernel would otherwise execute:
//      kind        first gather (index)                              second gather     third
//      interior    NB[sz][po]   = interior[sz-1] + tstack[po]        TSc[ci]           ZT[l1-l2]
//      1 x 1       NB[2][po]    = stackmm[po]  (interior row 2 is otherwise unused)   MMc[ci]   ZT[0]
//      bulge >= 2  BU[a_c][sz][a_p] = (bulge[sz-1] + AT(a_p)) + AT(a_c)                ZERO      ZERO
//      bulge 1     BU[a_c][1][a_p]  = bulge[0] + stack[a_p][a_c]  (*)                   ZERO      ZERO
//      (stack)     BU[a_c][0][a_p]  = not available: the stacked pair is not a loop candidate
//  (*) thal.c rejects a single-base bulge whose own (S,H) has H > 0 or S > 0 before the
//      predecessor is added; that test is applied here once per table entry.
// ZT[d] = ILAS * |d| exactly as the kernel would multiply it (ILAS * 0 is -0.0 and stays so);
// adding the ZERO entry (+0.0) is an exact no-op, so all kinds share one formula.
// po = a_p | s1[ii+1] << 2 | s2[jj+1] << 4 ; ci = (s2[j] * 4 + s2[j-1]) * 4 + s1[i-1].
#pragma once

#include <cstdint>

#include "nn_params.hpp"

namespace msspe {

struct FastTables {
    static constexpr int kMaxSz = 28;                          // loop sizes 0..28 (= 2 * 16 - 4) are addressable
    static constexpr int kNB = 0;                              // [sz-2][po], sz = 2..28: 27 * 64
    static constexpr int kBU = kNB + (kMaxSz - 1) * 64;        // [a_c][sz][a_p], sz = 0..28: 4 * 29 * 4
    static constexpr int kBUStride = (kMaxSz + 1) * 4;
    static constexpr int kTSc = kBU + 4 * kBUStride;           // [ci] 64
    static constexpr int kMMc = kTSc + 64;                     // [ci] 64
    static constexpr int kZero = kMMc + 64;                    // 1 (+3 pad)
    static constexpr int kZT = kZero + 4;                      // [32 + (l1 - l2)], 64 entries
    static constexpr int kEndL = kZT + 64;                     // [a*25 + oa*5 + ob]  100
    static constexpr int kEndR = kEndL + 100;                  // 100
    static constexpr int kWC = kEndR + 100;                    // [x*4 + y]       16
    static constexpr int kCount = kWC + 16;
    double S[kCount];
    int32_t H[kCount];
    int32_t usable;      // 1 when the kernel's exactness preconditions hold (see build_fast_tables)
    int32_t max_k;       // largest oligo length the preconditions were verified for
};

// Preconditions checked here (else usable = 0 and the caller must use the generic kernel):
//   * every finite enthalpy is an integer multiple of 10 cal/mol (packed as H/10 in 18 bits);
//   * no accepted value can reach the MinEntropyCutoff clamp for oligos up to max_k bases;
//   * 2 * max_k - 4 <= kMaxSz (every loop an oligo pair can form has a table row).
bool build_fast_tables(const NNTables &t, const PairTables &pt, int max_k, FastTables &out);

}  // namespace msspe
