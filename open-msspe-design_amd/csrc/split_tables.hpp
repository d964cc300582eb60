// split_tables.hpp -- tables of the exact-integer all-pairs kernel for LONG oligos
// (thal_pairs_split.hip: 15 .. 32 bases, 5-bit cell coordinates, a pair's DP table split over
// several lanes).
//
// Same exactness argument as IntTables (fast_tables.hpp): every table entropy is a multiple of
// 0.01 e.u., every enthalpy a multiple of 10 cal/mol, 310.15 * ILAS = -300, so
//      2000 * dG(37 C) = 20000 h + 600000 n - 6203 s
// is an exact int32 per DP cell and Primer3's "dG(candidate) < dG(current)" is decided on it.
//
// With 5-bit coordinates the loop table T[d][po] of the short-oligo kernel (d = 32 l1 + l2) would
// need 256 KB, so the loop term is kept in its two additive parts (one LDS gather each):
//      candidate = L[d] + X[xi] + y(cell) + G(predecessor)
//   kind            L[d]                                  xi
//   interior        g(interior[sz-1]) + 600000 |l1-l2|    kXP  + po            g(tstack[po])
//   1 x 1           0                                     kXMM + po            g(stackmm[po])
//   bulge, l1 = 0   0                                     kXB1 + l2 * 16 + pe  g(BU[a_c][sz][a_p])
//   bulge, l2 = 0   0                                     kXB2 + l1 * 16 + pe  (the same values)
//   stack (d = 0), loops beyond max_loop, impossible geometry:  L = kBig
// po = a_p | s1[ii+1] << 2 | s2[jj+1] << 4, pe = a_p | a_c << 2 (fast_tables.hpp).
// The f64 / int32 planes S, H, g follow FastTables' layout with loop sizes up to 30: they serve
// the cell-side terms, the end terms and the f64 replay of the optimal path.
#pragma once

#include <cstdint>

#include "nn_params.hpp"

namespace msspe {

struct SplitTables {
    static constexpr int kMaxSz = 30;                          // thal.c MAX_LOOP
    static constexpr int kNB = 0;                              // [sz-2][po], sz = 2..30
    static constexpr int kBU = kNB + (kMaxSz - 1) * 64;        // [a_c][sz][a_p], sz = 0..30
    static constexpr int kBUStride = (kMaxSz + 1) * 4;
    static constexpr int kTSc = kBU + 4 * kBUStride;           // [ci] 64
    static constexpr int kMMc = kTSc + 64;                     // [ci] 64
    static constexpr int kZero = kMMc + 64;                    // 1 (+3 pad)
    static constexpr int kZT = kZero + 4;                      // [32 + (l1 - l2)], 64 entries
    static constexpr int kEndL = kZT + 64;                     // [a*25 + oa*5 + ob]  100
    static constexpr int kEndR = kEndL + 100;                  // 100
    static constexpr int kWC = kEndR + 100;                    // [x*4 + y]       16
    static constexpr int kCount = kWC + 16;
    // "not available": three such terms plus a reachable value must not wrap, one alone must
    // land above kValid (3 kBig + kReach < 2^31, kBig - kReach >= kValid >= kReach)
    static constexpr int32_t kBig = 600000000;
    static constexpr int32_t kValid = 300000000;
    static constexpr int32_t kReach = 250000000;
    static constexpr int32_t kGh = 20000, kGn = 600000, kGs = 6203;
    static constexpr int kXP = 0, kXMM = 64, kXB1 = 128, kXB2 = kXB1 + 512, kXCount = kXB2 + 512;

    double S[kCount];
    int32_t H[kCount];
    int32_t g[kCount];       // kGh * (H / 10) - kGs * round(100 S); kBig: not available
    int32_t L[1024];         // [l1 * 32 + l2]
    int32_t X[kXCount];
    int32_t usable;          // 0: the preconditions do not hold, use the generic kernel
    int32_t max_k;           // longest oligo the range checks cover (0 if unusable)
    int32_t h_bias;          // a cell keeps H / 10 + h_bias in 16 unsigned bits
    int32_t f64_max_k;       // longest oligo for which the S / H planes alone are exact for an f64 DP
                             // (thal_pairs_wave.hip): integral enthalpies, MinEntropyCutoff out of reach
};

// max_loop: the chemistry's loop-size limit (thal.c maxLoop, <= 30).  Returns out.usable != 0.
bool build_split_tables(const PairTables &pt, int max_loop, SplitTables &out);

}  // namespace msspe
