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

// Integer image of the same tables for the exact-integer kernel (thal_pairs_int.hip).
// Every table entropy is a multiple of 0.01 e.u. and every enthalpy a multiple of 10 cal/mol, and
// the interior-loop asymmetry term is ILAS * |l1 - l2| with 310.15 * ILAS = -300 exactly, so the
// value of a DP cell is the integer triple (h = H / 10, s = 100 * S_tables, n = sum |l1 - l2|) and
//      2000 * dG(37 C) = 20000 h + 600000 n - 6203 s         (dG = H - 310.15 S)
// is an exact int32.  Primer3 compares dG values as doubles; two candidates whose exact values
// differ are at least 5e-4 cal/mol apart, far beyond double rounding, so integer comparison
// gives the same answer.  Exact ties (where double rounding decides) are detected and those
// pairs are handed to the f64 kernel.
struct IntTables {
    static constexpr int kMaxL = 14;                       // l1, l2 <= k - 2
    static constexpr int kRows = kMaxL * 17 + 1;           // row d = l1 * 16 + l2
    // "not available" is a large value, not a flag.  A candidate is the sum of a loop entry, a
    // cell-side entry and the predecessor's value: two unavailable terms must not wrap
    // (2 kBig + kReach < 2^31) and a single one must land above kValid (kBig - kReach >= kValid),
    // where kReach bounds every reachable |value| (checked by build_int_tables).
    static constexpr int32_t kBig = 900000000;             // not available
    static constexpr int32_t kValid = 500000000;           // candidates at or above this are void
    static constexpr int32_t kReach = 300000000;
    static constexpr int32_t kGh = 20000, kGn = 600000, kGs = 6203;
    // loop term of predecessor p -> cell c without the cell-side mismatch term:
    //   interior / 1x1 : column po (fast_tables.hpp), asymmetry included
    //   bulge          : column a_p | a_c << 2
    //   d = 0 (stack)  : kBig, the stacked pair is not a loop candidate
    int32_t T[kRows * 64];
    int32_t g[FastTables::kCount];   // kGh * (H / 10) - kGs * s per FastTables entry (kBig: unavailable)
    int32_t s[FastTables::kCount];   // round(100 * S)
    int32_t usable, max_k;
};
// usable = 0 unless every finite entropy is a multiple of 0.01 within 1e-7 and the packed fields
// (s: 18 bits signed, sums below kValid) cannot overflow for oligos up to max_k bases.
bool build_int_tables(const FastTables &ft, int max_k, IntTables &out);

}  // namespace msspe
