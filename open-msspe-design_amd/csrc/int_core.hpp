// int_core.hpp -- pieces shared by the exact-integer all-pairs kernels: thal_pairs_int.hip (list mode and
// the 14..16-base matrix shape) and thal_pairs_row.hip (matrix mode with a table specialised to the
// block's row primer).  Reference path: /root/reference/od-msspe/src/delta_g.rs:61-153 (one thal ANY per
// ordered pair); Primer3 2.6.1 thal() restated from SURVEY.md Appendix C.3.
#pragma once

#include "pair_core.hpp"

namespace msspe {

namespace {

constexpr int kC = 4;              // slots per chunk (= predecessor evaluations in flight)
// Two shapes of the matrix-mode kernel.  Oligos up to 13 bases need 189 rows of the loop table,
// which leaves LDS for a 768-thread block: with 48 slots the kernel fits 168 VGPRs, i.e. THREE
// waves per SIMD (the scan is VALU-bound, the rest of a cell latency-bound: the third wave fills
// the gaps); pairs with more cells (11 %) go to the list mode.  Longer oligos: 512 threads, 56 slots.
constexpr int kSlotsSmall = 52, kThreadsSmall = 768, kRowsSmall = 11 * 17 + 2;   // k <= 13
constexpr int kSlotsMatrix = 56;   // k <= 16 (the largest tables drag their waves)
constexpr int kSlotsList = 64;     // table of the list-mode kernel (lanes arrive sorted by table size)
constexpr int kThreadsI = 512;
constexpr int kPathMax = 16;       // a path has at most k <= 16 cells
// A tie in the terminal pick: thal() walks from one of the cells that share the minimal DP value g =
// dH - 310.15 dS.  What it reports for the walked structure is dH - T (dS + salt N) (drawDimer; T = the
// chemistry's temperature, N = base pairs - 1), i.e. relative to the structure this kernel walked
//      dG' - dG = (dH' - dH) (1 - T / 310.15) - T salt (N' - N).
// The enthalpies of all tied cells are known without a walk (the slot words carry them): the pick keeps their
// range.  When a call asks for decisions only (no dG / Tm planes; the edge list takes conflicts, whose ties are
// still handed on) and even the lowest value a tied structure could report (the enthalpy at the end of the
// range that lowers it, N' = 0 for the usual negative salt term, kPathMax - 1 otherwise) stays above the cut
// by a margin, the pair is final as "no conflict" whichever cell the reference walks from.
constexpr double kPickMargin = 0.5;   // cal/mol; exact values are multiples of 0.0005
// dh_min / dh_max: range over the tied cells of (their enthalpy - the walked cell's), in units of 10 cal/mol
__device__ __forceinline__ bool tied_pick_cannot_conflict(const ThalConsts &K, double G, int N, int dh_min, int dh_max)
{
    const double psi = 1.0 - K.temp_k / 310.15;
    const double dH = psi * 10.0 * (psi >= 0.0 ? (double)min(dh_min, 0) : (double)max(dh_max, 0));   // <= 0
    const double per = K.temp_k * K.salt;
    const double dN = per < 0.0 ? per * (double)N : -per * (double)(kPathMax - 1 - N);              // <= 0
    return (G + dH) + dN > K.g_cut + kPickMargin;
}
constexpr int kDragCost = 32;      // slots^2 a lane must save its wave to be sent to the list stage (tuned on 65,536 primers)
constexpr int kEmptyW = 0xff;      // coordinates (15, 15): fails every geometry test

// slot s: G[s] = exact 2000 * dG of the cell value; W[s] = h << 16 | po << 10 | im1 << 4 | jm1
// (bits 8, 9 zero, so that bits 8..15 read as po * 4, a byte offset).  The predecessor of the
// cell (im1 << 4 | jm1, 0xff: none) is only read by the traceback and lives in LDS.
// The table is kept as register tuples, plain local values (32 + 16 + 8 elements per plane for 56
// slots, 32 + 32 for 64): reads use compile-time element numbers, and the one write per cell goes
// through the wave-uniform slot number (s_set_gpr_idx + v_mov), so publishing a cell needs no
// branch tree and no register copies.  (They must stay plain locals passed by value: behind a
// struct or a reference the compiler leaves them in scratch memory.)
typedef int v32i __attribute__((ext_vector_type(32)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
template <int NS>
struct TabTypes;
template <>
struct TabTypes<56> {
    typedef v16i B;
    typedef v8i C;
};
template <>
struct TabTypes<48> {
    typedef v16i B;
    typedef v8i C;   // unused
};
typedef int v4i __attribute__((ext_vector_type(4)));
template <>
struct TabTypes<52> {
    typedef v16i B;
    typedef v4i C;
};
template <>
struct TabTypes<64> {
    typedef v32i B;
    typedef v8i C;   // unused
};
#define MSSPE_TAB_PARAMS                                                                                   \
    const v32i Ga, const v32i Wa, const typename TabTypes<NS>::B Gb, const typename TabTypes<NS>::B Wb, \
        const typename TabTypes<NS>::C Gc, const typename TabTypes<NS>::C Wc
#define MSSPE_TAB_ARGS Ga, Wa, Gb, Wb, Gc, Wc

template <int NS>
__device__ __forceinline__ int slot_of(const v32i a, const typename TabTypes<NS>::B b,
                                       const typename TabTypes<NS>::C c, int x)
{
    if constexpr (NS == 56) return x < 32 ? a[x & 31] : (x < 48 ? b[(x - 32) & 15] : c[(x - 48) & 7]);
    else if constexpr (NS == 48) return x < 32 ? a[x & 31] : b[(x - 32) & 15];
    else if constexpr (NS == 52) return x < 32 ? a[x & 31] : (x < 48 ? b[(x - 32) & 15] : c[(x - 48) & 3]);
    else return x < 32 ? a[x & 31] : b[(x - 32) & 31];
}

struct IBest {
    int G, W;   // candidate value; predecessor's packed word
};

// wave-uniform lane masks carried through the scan (scalar registers, no VALU work)
struct ScanMasks {
    unsigned long long tie;      // lanes whose running minimum is shared by two candidates
    unsigned long long stHave;   // lanes that met their (i-1, j-1) predecessor
};

// The lane's bit of a wave-uniform 64-bit mask (a ballot result): the mask IS a lane predicate, so one
// v_cndmask reads it; shifting it by the lane number costs a 64-bit shift and a live register pair.
__device__ __forceinline__ bool lane_bit(unsigned long long m)
{
    int r;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(m));
    return r != 0;
}

// median of three = the second smallest: with a <= b it keeps the two smallest values seen in (min, med3)
__device__ __forceinline__ int med3_i32(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// running minimum of the scan: value, the slot word that gave it, and the second smallest value (two
// candidates tie for the minimum iff G2 == G at the end: no per-visit mask bookkeeping)
struct RBest {
    int G, W, G2;
};

// Largest value over the lanes of the wave for 0 <= v < 256: bisection on ballots.  (The shuffle-based
// wave_max of pair_core.hpp keeps six lane-address vectors alive across the whole DP -- registers this
// kernel does not have.)
__device__ __forceinline__ int wave_max_u8(int v)
{
    int lo = 0;   // the answer is in [lo, lo + 2 * span)
#pragma unroll
    for (int span = 128; span > 0; span >>= 1) {
        const bool above = v >= lo + span;
        lo += __builtin_amdgcn_ballot_w64(above) ? span : 0;
    }
    return lo;
}

// Smallest value over the lanes of the wave for 0 <= v < 64: bisection on ballots (no LDS traffic).
__device__ __forceinline__ int wave_min_64(int v)
{
    int lo = 0;   // answer is in [lo, lo + span)
#pragma unroll
    for (int span = 32; span > 0; span >>= 1) {
        const bool below = v < lo + span;
        lo += __builtin_amdgcn_ballot_w64(below) ? 0 : span;
    }
    return lo;
}

// why a pair is not answered here (bit mask; statistics in IntArgs::reasons)
enum : int {
    kDeferTm = 1,        // maxTM: the two quotients agree to 1e-9
    kDeferLoopEq = 2,    // best loop candidate ties with the cell's stack / start value
    kDeferLoopTie = 4,   // two loop candidates tie for the minimum
    kDeferBad = 8,       // the minimum has H > 0 and S > 0 (thal.c would reject it)
    kDeferPick = 16,     // two cells tie in the terminal pick
    kDeferReplay = 32,   // replayed enthalpy differs from the tracked one (never expected)
    kDeferPathTie = 64,  // a cell of the optimal path has an equal-valued alternative
};

struct IntResult {
    PairResult r;
    int defer;   // not answered here: OR of the reasons above
};

struct IntArgs {
    FastArgs f;
    const IntTables *it;
    unsigned long long *reasons;   // optional statistics: [0] pairs handed on, [1 + b] reason bit b (b < 7),
                                   // [8] samples kept, [9 ... 1032] samples; the list mode counts at
                                   // [1033 ...] (same layout, no samples)
    int stat_off;                  // 0 (matrix mode) or 1033 (list mode)
    unsigned *work_counter;        // matrix mode: next work item (one row x 64 sorted columns), zero at launch
};

// A list entry whose pair needs the f64 kernels (an exact tie was met) carries this bit in .x;
// entries without it only left their wave because of their table size and may be retried here.
constexpr unsigned kNeedsF64 = 0x80000000u;

}  // namespace

}  // namespace msspe
