// thal_pairs_row.hip -- all-pairs cross-dimer kernel, exact-integer DP, matrix mode for oligos of up
// to 13 bases with the loop table SPECIALISED TO THE BLOCK'S ROW PRIMER.
//
// Same job, same recurrence, same exactness argument and same outputs as thal_pairs_int.hip (the
// reference's "format N^2 lines -> ntthal -> parse" loop, /root/reference/od-msspe/src/delta_g.rs:61-153;
// Primer3 2.6.1 thal() restated from SURVEY.md Appendix C.3).  What changes is the price of one
// predecessor visit, which is what the kernel is made of (about 1,000 lock-step visits per pair):
//
//   * gfx950 issues v_add / v_sub / v_and / v_or / v_lshr / v_mov at one wave-instruction per ~2.4
//     cycles per SIMD, and everything else the scan needs (v_cndmask, v_cmp, v_min, v_add3, v_bfe,
//     v_lshl_or, SDWA forms) at one per ~4.3 cycles (tools/valu_peak.hip, profiles/r02_valu_peak.txt).
//     The general visit of thal_pairs_int.hip spends 16 (far) to 24 (near) mostly half-rate
//     instructions, most of them on classifying the loop (bulge / 1 x 1 / interior, which cell-side
//     term, which table column) from the coordinates of cell and predecessor.
//   * A work item here is one ROW primer x a segment of the composition-sorted columns, processed by
//     the twelve waves of one block.  Every lane of the block shares oligo 1, so everything a loop
//     term takes from oligo 1 -- the predecessor's pair and its 3' neighbour, the cell's pair and its
//     5' neighbour, hence the kind-specific tables AND the cell-side term of every loop kind whose
//     cell-side term does not need a base of oligo 2 that the predecessor does not already carry --
//     is folded into a table T[l2][i][i - ii][n2] built in LDS per row (37 KB, rebuilt in ~2 us
//     from the chemistry's general table).  A slot word carries the predecessor's coordinates as
//     K = 772 jj + 4 ii + n2  (n2 = the base of oligo 2 right of the predecessor), so that ONE
//     subtraction from a per-cell constant yields the table index:
//         C - K = 772 (j - 1 - jj) + 4 (14 i + (i - ii)) + (3 - n2)
//     A predecessor right of the cell (j - 1 - jj < 0) makes the 17-bit field wrap onto an address beyond the
//     block's LDS allocation, which reads 0 = "not available" (kRowZero below): no validity test, no loop
//     classification, no column fix-up.  772 = 4 (mod 64): the 64 lanes of a wave sit on nearly the same rows (same slot number),
//     what differs between them is l2 and n2, and with this stride the 48 (l2, n2) combinations fall
//     into 48 different LDS banks (a power-of-two layout puts them all into one: measured 16-way
//     conflicts, the kernel then gains 4 % instead of 30 %).
//     Nothing is left of the classification: the cell-side mismatch term of an interior loop with l2 >= 2 needs the
//     base left of the cell on oligo 2, which only the lane knows, so the lane adds it to EVERY entry of the rows
//     two or more above the cell, and the entries that must not get it hold the loop term minus it (their m2 is
//     known when the table is built: build_row_table).
//   * far visit (predecessor known to lie two or more rows up):  sub, lshr, [ds_read], add3, med3, mov, min_f64
//     = 3 full-rate + 3 half-rate instructions (general kernel: 16); near visit (the row above the cell):
//     + cmp, 2 x cndmask that catch the cell (i-1, j-1) for maxTM.
//
// Ties, maxTM, the terminal pick, the traceback by pointer and the f64 replay of the optimal path are
// those of thal_pairs_int.hip; a pair this kernel does not answer goes to the same hand-over list.
#include "int_core.hpp"
#include "row_scan_pinned.inc"


namespace msspe {

namespace {

// One class per (longest oligo, block shape): 13 bases with 768 threads and 52 stored cells (three waves per SIMD, the
// table read without an address clamp), 15 bases with 512 threads and 64 stored cells (two waves per SIMD, clamped
// addresses).  Everything below is the same code; the constants follow from the four parameters.
template <int ROWK, int ROWT, int ROWS, bool ROWOOB>
struct RowKernel {
static constexpr int kRowK = ROWK;                      // longest oligo of this instance
// The 13-base instance binds its slot table to FIXED registers: its scan and its publish are inline asm whose tuple
// operands carry explicit register ranges, the scan one generated block (tools/gen_row_scan_asm.py ->
// row_scan_pinned.inc: register map and reasons there).  The other instances leave the tuples to the allocator.
static constexpr bool kPin = ROWK == 13 && ROWT == 768 && ROWS == 52 && ROWOOB;
// ... and so do the 14- and 15-base instances (64 slots as two v32i per plane, 256 VGPRs, clamped table addresses)
static constexpr bool kPin64 = ROWS == 64 && ROWT == 512 && !ROWOOB;
static constexpr int kRowThreads = ROWT, kRowSlots = ROWS;   // 13 bases: 768 threads (three waves per SIMD), 52 stored cells per pair
static constexpr int kRowL2 = kRowK - 1;                // l2 = j - 1 - jj = 0 .. k - 2
static constexpr int kRowR = kRowK + 1;                 // digits of the row index: 14 i + (i - ii) at 13 bases
// stride of l2: 4 * kRowR * kRowK entries used, rounded up to = 4 (mod 64) (13 bases: 728 -> 772; 15: 960 -> 964)
static constexpr int kRowA = (4 * kRowR * kRowK + 59) / 64 * 64 + 4;
static_assert(kRowA % 64 == 4 && kRowA >= 4 * kRowR * kRowK, "48 (l2, n2) combinations in 48 different LDS banks");
static constexpr int kRowTEntries = kRowL2 * kRowA;
static constexpr int kRowTBytes = kRowTEntries * 4;     // byte offset of the "not available" entry behind the table
static constexpr int kHBias = 16384;                    // h = H / 10 is kept as h + kHBias in 15 bits
static constexpr int kRowGBase = FastTables::kTSc;      // first compact-table entry kept in LDS
static constexpr int kRowGCount = FastTables::kCount - kRowGBase;
// The running minimum of a scan is ONE v_min_f64 on the pair (candidate value : slot word): every vector
// instruction of this loop issues at the same rate, f64 or not, and a minimum of two 64-bit
// patterns with the value on top is compare + two selects in one.  For that the patterns must be positive
// normal doubles ordered like the values: every table entry carries + kRowD, which puts the reachable
// candidates (|value| < kReach) and the void ones (kRowU instead of kBig in this table and in the cell-side
// terms) into [2^20, 0x7fefffff] as high words.
static constexpr int kRowD = 350000000;
static constexpr int kRowU = 700000000;                 // "not available" in T and in the cell-side terms: a candidate that holds one is
                                                 // beyond anything a cell can take (every cell value is a reachable one, < kReach)
static constexpr int kRowInit = kRowD + IntTables::kReach + 50000000;   // the scan's starting minimum: above every valid candidate, below every void one
static_assert(kRowD - IntTables::kReach >= (1 << 20), "smallest candidate must be a normal double's high word");
// (a predecessor's value is always a reachable one: pairs_row_tables_ok() admits only chemistries whose end and
//  stacked-pair terms are all finite, and a cell takes its value from those or from a non-void candidate)
static_assert((long long)kRowU + kRowU + IntTables::kReach + kRowD <= 0x7fefffffLL,
              "largest candidate (void loop + void cell side + reachable predecessor) must stay a finite double");
static_assert(kRowU + kRowD - IntTables::kReach > kRowInit && kRowInit - kRowD > IntTables::kReach,
              "a candidate that holds a void term stays above the starting minimum, which no cell can take");
// No address clamp: a predecessor that is not up-left of the cell makes the 17-bit address field wrap to at least
// kRowWrapMin; with the table placed so that those addresses lie beyond the block's LDS allocation the read
// returns 0 (gfx950: every ds_read at or beyond the allocation, rounded up to 1,280 bytes, reads 0;
// tools/lds_oob_probe.hip, checked again by pairs_row_lds_reads_zero() when an engine is made -- a device that
// fails it, and a context with option row_oob = 0, runs the general integer kernel instead), and 0 is made
// to mean "not available": every table entry carries - kRowZero and every slot value + kRowZero.  That saves
// the unsigned min that clamped the address in every visit.
static constexpr int kRowZero = kRowU + kRowD;          // the stored pattern of "not available" before the shift: now 0
static_assert((long long)IntTables::kReach + kRowZero < 0x7fffffffLL, "slot values + kRowZero are int32");
static_assert(-(long long)IntTables::kReach - kRowU + kRowD - kRowZero > -0x7fffffffLL,
              "table entries (a folded-in void cell-side term included) - kRowZero are int32");
static constexpr int kSegGroups = 256;                  // column groups (of 64) per work item
// slot word:  bits 31..17  K = 772 jj + 4 ii + n2     (bits 15, 16 zero: the byte offset is K << 2)
//             bits 14..0   h + kHBias
// an empty slot: K one beyond the largest real one (12 * 772 + 4 * 12 + 3), i.e. beyond every cell's minuend, so
// that the difference is negative for every cell (and as large as possible after the wrap, see kRowWrapMin)
static constexpr int kEmptyRowK = (kRowK - 1) * kRowA + 4 * (kRowK - 1) + 3 + 1;
static constexpr int kEmptyRowW = (kEmptyRowK << 17) | kHBias;
// smallest table address (bytes) of a predecessor that is not up-left of the cell: either the minuend is the
// smallest possible (column 1, row 0: K = 3) and the word the empty one, or the cell is in column 0, whose
// minuend wraps to 2^15 - 769 + 60 i
static constexpr int kRowWrapMin = 4 * ((1 << 15) - (kRowA - 3) - kEmptyRowK) < (1 << 17) - 4 * (kEmptyRowK - 3)
                                ? 4 * ((1 << 15) - (kRowA - 3) - kEmptyRowK) : (1 << 17) - 4 * (kEmptyRowK - 3);
static_assert(kEmptyRowK < (1 << 15), "K is a 15-bit field");

// Order matters.  LDS instructions take a 16-bit immediate offset, so what is addressed as "lane + constant" or
// "table + index" lives in the first 64 KB and needs no address arithmetic; and T sits so far
// back that its wrapped addresses fall off the end of the allocation (static_asserts below the struct).
static constexpr int kPredLo = 24;   // rows of the predecessor bytes in front of T (the row number is a scalar: no cost)
struct SharedRow {
    // per-lane state that is touched once per cell (registers are the scarce resource: 104 of a lane's
    // 168 hold the table, and what does not fit goes to scratch memory, i.e. to HBM latency)
    int pick[4][kRowThreads];           // terminal pick: value, word, 0 or the tied later cells' enthalpy range (max << 16 | min, biased by
                                        // 32768), the picked cell's predecessor byte | its "equal-valued alternative" bit << 8
    unsigned soft[2][kRowThreads];      // slots whose value has an equal-valued alternative (bit mask)
    int yts[64];                        // [i][m2]: cell-side mismatch term of an interior loop, G units
    int ytsh[64];                       // ... and its enthalpy / 10
    // the entries of the compact tables that a cell reads (cell-side mismatch, end and stacked-pair terms:
    // FastTables::kTSc and up); the loop entries below that are read once per step of the replay, from global
    int g[kRowGCount];
    int h[kRowGCount];                  // enthalpy / 10 (every finite one is a multiple of 10 cal/mol)
    short TH[kRowTEntries + 4];         // enthalpy / 10 of the loop term T holds, << 1 | "the lane adds ytsh" (same index)
    double cq[100];                     // 620300 * (init_S + rS + RC) per right-end context (maxTM)
    unsigned char pred_lo[kPredLo][kRowThreads];
    unsigned next_group;
    int item;
    int T[kRowTEntries + 4];            // [l2 * 772 + (14 i + (i - ii)) * 4 + (3 - n2)]; entry kRowTEntries: not available
    unsigned char pred_hi[kRowSlots + 1 - kPredLo][kRowThreads];   // + 1: the row the last row of a full table writes (never read)
    unsigned short path[kPathMax][kRowThreads];
    __device__ __forceinline__ unsigned char &pred(int slot, int tid)   // slot: wave-uniform
    {
        return slot < kPredLo ? pred_lo[slot][tid] : pred_hi[slot - kPredLo][tid];
    }
};
static_assert(offsetof(SharedRow, T) <= 65532, "T is addressed with an immediate offset");
// ROWOOB: the scan reads the table without clamping its address (see kRowZero); otherwise the address is clamped
// onto the "not available" entry behind the table (one more instruction per visit; the 15-base instance, whose
// table is too long for its wrapped addresses to clear the allocation)
static_assert(!ROWOOB || offsetof(SharedRow, T) + kRowWrapMin >= (sizeof(SharedRow) + 1279) / 1280 * 1280,
              "wrapped table addresses must lie beyond the block's LDS allocation (granule: 1,280 bytes)");
static_assert(kRowTBytes <= kRowWrapMin, "valid addresses stay inside the table, wrapped ones beyond it");
static_assert(sizeof(SharedRow) <= 160 * 1024, "one block per CU");

struct KParts {
    int ii, jj, n2;
};
// K = kRowA jj + 4 ii + n2: jj = K / kRowA by multiplication (exact for every K a slot word can hold: checked below)
static constexpr unsigned kRowDivMagic = ((1u << 24) + (unsigned)kRowA - 1u) / (unsigned)kRowA;
static constexpr bool div_magic_ok()
{
    for (unsigned K = 0; K <= (unsigned)kEmptyRowK; ++K)
        if (((K * kRowDivMagic) >> 24) != K / (unsigned)kRowA) return false;
    return true;
}
static __device__ __forceinline__ KParts k_parts(unsigned K)
{
    KParts p;
    p.jj = (int)((K * kRowDivMagic) >> 24);
    const unsigned rest = K - (unsigned)p.jj * (unsigned)kRowA;
    p.ii = (int)(rest >> 2);
    p.n2 = (int)(rest & 3u);
    return p;
}
// The predecessor byte of a cell (pred_lo / pred_hi), made from the predecessor's slot word once per cell; only the
// walk back reads it, through sig_of_cw.  With kRowA = 772 = 768 + 4 (13 bases) K = 768 jj + 4 (ii + jj) + n2, so the
// word's bits 25.. are 3 jj and its bits 19..23 ii + jj: the byte is 4 (3 jj) + (ii + jj) = 13 jj + ii, three
// instructions.  Otherwise jj << 4 | ii by k_parts' division.  0xff: no predecessor (the word kNoPredW gives it).
static constexpr bool kCompactCw = kRowA == 772;
static constexpr int kNoPredW = kCompactCw ? (int)((56u << 25) | (31u << 19)) : (int)((unsigned)(15 * kRowA + 60) << 17);
static_assert(15 * kRowA + 60 < (1 << 14), "kNoPredW is a positive 32-bit word");   // (cw_ok: checked below the struct)
static __device__ __host__ __forceinline__ constexpr int word_cw(int W)
{
    if constexpr (kCompactCw) {
        return (int)((((unsigned)W >> 25) << 2) + (((unsigned)W >> 19) & 31u));
    } else {
        const unsigned K = (unsigned)W >> 17;
        const unsigned jj = (K * kRowDivMagic) >> 24;
        return (int)((jj << 4) | ((K - jj * (unsigned)kRowA) >> 2));
    }
}
// what the traceback compares: K without n2
static __device__ __forceinline__ unsigned word_sig(int W) { return ((unsigned)W >> 17) & ~3u; }
static __device__ __host__ __forceinline__ constexpr unsigned sig_of_cw(int cw)
{
    if constexpr (kCompactCw) {
        const unsigned jj = ((unsigned)cw * 79u) >> 10;   // cw / 13 for cw < 169
        return jj * (unsigned)(kRowA - 4 * 13) + ((unsigned)cw << 2);   // kRowA jj + 4 (cw - 13 jj)
    } else {
        return (unsigned)(cw >> 4) * (unsigned)kRowA + (unsigned)((cw & 15) << 2);
    }
}
static constexpr bool cw_ok()
{
    for (int jj = 0; jj < kRowK; ++jj)
        for (int ii = 0; ii < kRowK; ++ii)
            for (int n2 = 0; n2 < 4; ++n2) {
                const unsigned K = (unsigned)(jj * kRowA + 4 * ii + n2);
                const int cw = word_cw((int)(K << 17) | 0x7fff);
                if (cw == 0xff || (cw & ~0xff) || sig_of_cw(cw) != (K & ~3u)) return false;
            }
    return word_cw(kNoPredW) == 0xff && word_cw(kNoPredW | 0x7fff) == 0xff;
}
static __device__ __forceinline__ int word_h(int W) { return (W & 0x7fff) - kHBias; }
// the f64 kernels' word (h << 14 | po << 8 | im1 << 4 | jm1) of pair_core.hpp's cand_* helpers;
// po = pair base | 3' neighbour on oligo 1 << 2 | right neighbour on oligo 2 << 4
static __device__ __forceinline__ int core_of_k(unsigned K, unsigned s1, int h)
{
    const KParts p = k_parts(K);
    const int po = (int)((s1 >> (2 * p.ii)) & 15u) | (p.n2 << 4);
    return (h << 14) | (po << 8) | (p.ii << 4) | p.jj;
}
static __device__ __forceinline__ int core_word_row(int W, unsigned s1) { return core_of_k((unsigned)W >> 17, s1, word_h(W)); }

struct RCell {
    unsigned C;     // per-cell minuend of the address subtraction
    int yTS;        // cell-side mismatch term of an interior loop with l2 >= 2
    int idxStk;     // table address that the cell (i-1, j-1) produces
};

// Table address (byte offset) of predecessor word W seen from the cell with minuend C; a predecessor that
// is not up-left of the cell lands on a "not available" entry.  (The clamped form: used once per cell, for the
// winner's enthalpy; the scan itself uses scan_index().)
static __device__ __forceinline__ unsigned row_index(unsigned C, unsigned W)
{
    return min((C - W) >> 15, (unsigned)kRowTBytes);
}

// Running minimum of a scan (all values carry + kRowD): GW = (value : slot word) as one double, G2 = second
// smallest value so far (two candidates tie for the minimum iff G2 == the minimum at the end).
struct RowBest {
    double GW;
    int G2;
};
static __device__ __forceinline__ int best_g(const RowBest &b) { return __double2hiint(b.GW); }
static __device__ __forceinline__ int best_w(const RowBest &b) { return __double2loint(b.GW); }
static __device__ __forceinline__ void take_min(RowBest &b, int cand, int Wp)
{
    b.G2 = med3_i32(best_g(b), b.G2, cand);   // second smallest so far
    const double cd = __hiloint2double(cand, Wp);
    // (asm: fmin() would canonicalise its operands first; both are normal numbers here by construction, and
    //  idle lanes, whose patterns may be anything, publish nothing)
    asm("v_min_f64 %0, %1, %2" : "=v"(b.GW) : "v"(b.GW), "v"(cd));
}

// the scan's address: no clamp (a wrapped address lies beyond the allocation and reads 0 = not available)
static __device__ __forceinline__ unsigned scan_index(unsigned C, unsigned W)
{
    if constexpr (ROWOOB) return (C - W) >> 15;
    else return min((C - W) >> 15, (unsigned)kRowTBytes);
}

// The table addresses and the (still outstanding) table values of one chunk of slots.
struct ChunkLoad {
    unsigned idx[kC];
    int t[kC];
};

template <int NS, int PC>
static __device__ __forceinline__ void chunk_issue(MSSPE_TAB_PARAMS, const char *T, const RCell &c, ChunkLoad &L)
{
#pragma unroll
    for (int e = 0; e < kC; ++e)
        L.idx[e] = scan_index(c.C, (unsigned)slot_of<NS>(Wa, Wb, Wc, PC * kC + e));
#pragma unroll
    for (int e = 0; e < kC; ++e) L.t[e] = *(const int *)(T + L.idx[e]);
}

// All slots below `upto` as predecessors of the cell, kC at a time; chunks at or above near_from (the
// row above the cell) also catch the cell (i-1, j-1).  The chunks are unrolled with compile-time register
// numbers and left through a wave-uniform branch.
// run / far: one bit per chunk (set by the caller once per ROW of the DP: which chunks hold slots of the rows above,
// which of them lie entirely in rows i-2 and above).  Two scalar words instead of two comparisons per chunk whose
// results the compiler would keep alive, as lane masks, across the whole cell.
template <int NS, int PC = 0>
static __device__ __forceinline__ void scan_fill_row(MSSPE_TAB_PARAMS, unsigned run, unsigned far, int near_from, const char *T,
                                              const RCell &c, RowBest &best, IBest &stk, ScanMasks &m)
{
    if constexpr (PC * kC < NS) {
        if ((run >> PC) & 1u) {   // wave-uniform
            ChunkLoad cur;
            chunk_issue<NS, PC>(MSSPE_TAB_ARGS, T, c, cur);
            if ((far >> PC) & 1u) {   // wave-uniform: slots of rows i-2 and above
                asm volatile("" ::"n"(PC));   // keeps the chunks from being merged into selects
#pragma unroll
                for (int e = 0; e < kC; ++e) {
                    const int Gp = slot_of<NS>(Ga, Gb, Gc, PC * kC + e), Wp = slot_of<NS>(Wa, Wb, Wc, PC * kC + e);
                    // every entry of these rows (r >= 2) takes the cell-side term: where the loop has none of its own
                    // (l2 <= 1) the table holds the loop term MINUS it (build_row_table).  Unavailable: kRowU + ...
                    const int cand = cur.t[e] + c.yTS + Gp;
                    take_min(best, cand, Wp);
                }
            } else {
                asm volatile("" ::"n"(PC + 64));
                int rem = near_from - PC * kC;   // slots of this chunk that still belong to row i-2
                asm volatile("" : "+s"(rem));    // (computed here, per use: hoisted out of the cell loop it costs a register per slot)
#pragma unroll
                for (int e = 0; e < kC; ++e) {
                    const int Gp = slot_of<NS>(Ga, Gb, Gc, PC * kC + e), Wp = slot_of<NS>(Wa, Wb, Wc, PC * kC + e);
                    // the row above the cell (r = 1) takes no cell-side term; slots of row i-2 at the head of this chunk do
                    const int y = c.yTS & (e < rem ? -1 : 0);   // a wave-uniform mask (one s_cselect_b32)
                    const int cand = cur.t[e] + y + Gp;
                    take_min(best, cand, Wp);
                    const bool isstk = cur.idx[e] == (unsigned)c.idxStk;   // the cell (i-1, j-1)
                    stk.G = isstk ? Gp : stk.G;
                    stk.W = isstk ? Wp : stk.W;
                    m.stHave |= __builtin_amdgcn_ballot_w64(isstk);
                }
            }
            scan_fill_row<NS, PC + 1>(MSSPE_TAB_ARGS, run, far, near_from, T, c, best, stk, m);
        }
    }
}

// ---- the 13-base instance's scan and publish over the register-bound table (kPin) -------------------------------------
struct PinScan {
    int bestG, bestW, G2, stkG, stkW;
    unsigned long long stHave;
};
static __device__ __forceinline__ PinScan pin_scan(const v32i Ga, const v32i Wa, const v16i Gb, const v16i Wb, const v4i Gc,
                                            const v4i Wc, const RCell &c, int n_far, int n_run, int near_from)
{
    PinScan r;
    int a0, a1, a2, a3, t0, t1, t2, t3, yt;
    asm volatile(MSSPE_ROW13_SCAN_ASM
                 : [SG] "=&v"(r.stkG), [SW] "=&v"(r.stkW), [HV] "=&s"(r.stHave), MSSPE_ROW13_ACC_OUT(r.bestG, r.bestW, r.G2),
                   [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [t0] "=&v"(t0),
                   [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [yt] "=&v"(yt)
                 : [C] "v"(c.C), [Y] "v"(c.yTS), [IDX] "s"(c.idxStk), [NFAR] "s"(n_far), [NRUN] "s"(n_run), [NEAR] "s"(near_from),
                   [TOFF] "n"((int)offsetof(SharedRow, T)), [INIT] "n"(kRowInit), MSSPE_ROW13_TUPLES_IN(Ga, Gb, Gc, Wa, Wb, Wc)
                 : MSSPE_ROW13_SCAN_CLOBBERS);
    return r;
}
// slot: wave-uniform, < kRowSlots.  One indexed write per plane, whatever tuple the slot falls into.
static __device__ __forceinline__ void pin_publish(v32i &Ga, v32i &Wa, v16i &Gb, v16i &Wb, v4i &Gc, v4i &Wc, int slot, int G0s,
                                            int Wcell)
{
    asm volatile("s_set_gpr_idx_on %[SLOT], gpr_idx(DST)\n\t"
                 "v_mov_b32 v%c[G0], %[G]\n\t"
                 "v_mov_b32 v%c[W0], %[W]\n\t"
                 "s_set_gpr_idx_off"
                 : MSSPE_ROW13_TUPLES_INOUT(Ga, Gb, Gc, Wa, Wb, Wc)
                 : [G] "v"(G0s), [W] "v"(Wcell), [SLOT] "s"(slot), [G0] "n"(MSSPE_ROW13_G0), [W0] "n"(MSSPE_ROW13_W0)
                 : "m0");
}

static __device__ __forceinline__ PinScan pin_scan64(const v32i Ga, const v32i Wa, const v32i Gb, const v32i Wb, const RCell &c,
                                              int n_far, int n_run, int near_from)
{
    PinScan r;
    int a0, a1, a2, a3, t0, t1, t2, t3, yt;
    asm volatile(MSSPE_ROW64_SCAN_ASM
                 : [SG] "=&v"(r.stkG), [SW] "=&v"(r.stkW), [HV] "=&s"(r.stHave), MSSPE_ROW64_ACC_OUT(r.bestG, r.bestW, r.G2),
                   [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [t0] "=&v"(t0),
                   [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [yt] "=&v"(yt)
                 : [C] "v"(c.C), [Y] "v"(c.yTS), [IDX] "s"(c.idxStk), [NFAR] "s"(n_far), [NRUN] "s"(n_run), [NEAR] "s"(near_from),
                   [TOFF] "n"((int)offsetof(SharedRow, T)), [INIT] "n"(kRowInit), [TB] "n"(kRowTBytes),
                   MSSPE_ROW64_TUPLES_IN(Ga, Gb, Wa, Wb)
                 : MSSPE_ROW64_SCAN_CLOBBERS);
    return r;
}
static __device__ __forceinline__ void pin_publish64(v32i &Ga, v32i &Wa, v32i &Gb, v32i &Wb, int slot, int G0s, int Wcell)
{
    asm volatile("s_set_gpr_idx_on %[SLOT], gpr_idx(DST)\n\t"
                 "v_mov_b32 v%c[G0], %[G]\n\t"
                 "v_mov_b32 v%c[W0], %[W]\n\t"
                 "s_set_gpr_idx_off"
                 : MSSPE_ROW64_TUPLES_INOUT(Ga, Gb, Wa, Wb)
                 : [G] "v"(G0s), [W] "v"(Wcell), [SLOT] "s"(slot), [G0] "n"(MSSPE_ROW64_G0), [W0] "n"(MSSPE_ROW64_W0)
                 : "m0");
}

// thal ANY for the lane's pair (oligo 1 = the block's row primer).  !active: idle lane.  n_slots: the slots the
// wave's rows take (all rows but the last, padded to the widest lane); wmax4: the widest lane's count of each base.
template <int NS>
static __device__ __forceinline__ IntResult run_pair_row(SharedRow &sh, const ThalConsts &K, const double *gS, const int *gH,
                                                  const SeqPair &q, bool active, unsigned wmax4, int n_slots,
                                                  bool decisions_only, bool edge_values)
{
    v32i Ga = kRowZero, Wa = kEmptyRowW;   // slot values carry + kRowZero
    typename TabTypes<NS>::B Gb = kRowZero, Wb = kEmptyRowW;
    typename TabTypes<NS>::C Gc = kRowZero, Wc = kEmptyRowW;
    int defer = 0;
    CellCtx c;
    c.rS = 0.0;
    c.rH = 0;
    c.im1p = c.jm1p = 0;
    c.yTS = c.yMM = c.bBase = 0;
    int pick_g = 0x7fffffff;                // the terminal pick's value so far (a register: it is read by every cell)
    sh.pick[1][threadIdx.x] = 0;            // pickW
    sh.pick[2][threadIdx.x] = 0;            // nTie
    sh.pick[3][threadIdx.x] = 0xff;
    sh.soft[0][threadIdx.x] = sh.soft[1][threadIdx.x] = 0u;

    // Rows are walked by the WAVE: every lane shares oligo 1, so row i holds the cells whose oligo-2 base
    // is 3 - s1[i], and a row takes as many slots as its widest lane has cells there (lanes with fewer
    // publish an empty slot, which fails every geometry test).  Composition-sorted lanes differ by a
    // base or two, so the padding is about 1 % of the slots; in exchange the row number, the row's
    // first slot, the near / far boundary and everything cell_bases() takes from oligo 1 are scalars.
    int slot_ = 0, start_im1 = 0;   // first slot of row i-1
    for (int i_ = 0; i_ < q.len; ++i_) {
      const int im1 = __builtin_amdgcn_readfirstlane(i_);
      const int a_row = (int)((q.s1 >> (2 * im1)) & 3u);
      const int w_row = (int)((wmax4 >> (8 * (3 - a_row))) & 0xffu);   // widest lane's cells in this row
      const int row_start = __builtin_amdgcn_readfirstlane(slot_);
      // The cells of the last row are nobody's predecessors: they are computed and may be picked, but take no
      // slot (a pair needs cells - cells of the last row slots: three or four more pairs in a hundred fit)
      const bool stored = im1 < q.len - 1;   // wave-uniform
      // chunks that hold slots below row_start / that lie entirely below the first slot of row i-1 (scalars)
      const unsigned run_chunks = (unsigned)__builtin_amdgcn_readfirstlane((int)((1u << ((row_start + kC - 1) / kC)) - 1u));
      const unsigned far_chunks = (unsigned)__builtin_amdgcn_readfirstlane((int)((1u << (start_im1 / kC)) - 1u));
      unsigned mrem = active ? spaced_mask(q.s2, 3 - a_row, q.lenmask) : 0u;
      for (int c_ = 0; c_ < w_row; ++c_, slot_ += stored ? 1 : 0) {
        const int slot = __builtin_amdgcn_readfirstlane(slot_);
        // ---- the lane's next complementary cell of this row (none: an empty slot)
        const bool in = mrem != 0u;
        const int jm1 = ((__ffs((int)mrem) - 1) >> 1) & 15;
        mrem &= mrem - 1;
        RCell rc;
        rc.C = ((unsigned)((jm1 - 1) * kRowA + im1 * (4 * (kRowR + 1)) + 3) << 17) | 0x7fffu;
        // m2 = base left of the cell on oligo 2 (0 in column 0, where no loop can close)
        rc.yTS = sh.yts[(im1 << 2) | (int)(((q.s2 << 2) >> (2 * jm1)) & 3u)];
        rc.idxStk = ((im1 * kRowR + 1) * 4 + (int)((q.s1 >> (2 * im1)) & 3u)) * 4;   // l2 = 0, i - ii = 1, 3 - n2 = base of the cell
        // ---- all earlier slots as predecessors
        RowBest rb;
        IBest stk;
        rb.GW = __hiloint2double(kRowInit, 0);
        rb.G2 = 0x7fffffff;
        stk.G = stk.W = 0;
        ScanMasks sm;
        sm.tie = sm.stHave = 0ull;
        // predecessors: every slot of the rows above; row i-1 (where the cell (i-1, j-1) lives) through
        // the code that catches it
        if constexpr (kPin) {
            const PinScan ps = pin_scan(Ga, Wa, Gb, Wb, Gc, Wc, rc, start_im1 / kC, (row_start + kC - 1) / kC, start_im1);
            rb.GW = __hiloint2double(ps.bestG, ps.bestW);
            rb.G2 = ps.G2;
            stk.G = ps.stkG;
            stk.W = ps.stkW;
            sm.stHave = ps.stHave;
        } else if constexpr (kPin64) {
            const PinScan ps = pin_scan64(Ga, Wa, Gb, Wb, rc, start_im1 / kC, (row_start + kC - 1) / kC, start_im1);
            rb.GW = __hiloint2double(ps.bestG, ps.bestW);
            rb.G2 = ps.G2;
            stk.G = ps.stkG;
            stk.W = ps.stkW;
            sm.stHave = ps.stHave;
        } else {
            unsigned run = run_chunks, far = far_chunks;
            asm volatile("" : "+s"(run), "+s"(far));   // the per-chunk bit tests stay where they are used
            scan_fill_row<NS>(MSSPE_TAB_ARGS, run, far, start_im1, (const char *)sh.T, rc, rb, stk, sm);
        }
        const bool tie = rb.G2 == best_g(rb);   // two loop candidates share the minimum
        RBest best;
        best.G = best_g(rb) - kRowD;
        best.W = best_w(rb);
        const bool stHave = __builtin_amdgcn_inverse_ballot_w64(sm.stHave);   // the mask IS the lane predicate: no select + compare
        const CellBases b = cell_bases(q, im1, jm1, c);   // after the scan: nothing of it is live across it
        // ---- thal.c maxTM(): helix extension if it raises Tm (see thal_pairs_int.hip).  Enthalpies in units
        //      of 10 cal/mol (h): T = A / B with A = 10 (h + 20 + rh), 620300 B = 20000 h - G + cq; the
        //      factor 10 drops out of A1 B0 > A0 B1
        // (every LDS read of the cell's own terms is issued here, in one group: one wait, not four)
        int h0 = sh.h[b.idxL - kRowGBase], G0 = sh.g[b.idxL - kRowGBase], predW = kNoPredW, flags = 0, cell_soft = 0;
        const int rh = sh.h[b.idxR - kRowGBase], gR = sh.g[b.idxR - kRowGBase], hwc = sh.h[b.wc - kRowGBase], gwc = sh.g[b.wc - kRowGBase];
        const double cq = sh.cq[b.idxR - FastTables::kEndR];
        const int pickG = pick_g;
        if (stHave) {
            const int h1 = word_h(stk.W) + hwc;
            const int G1 = stk.G + (gwc - kRowZero);
            const double A0 = (double)(h0 + 20 + rh), A1 = (double)(h1 + 20 + rh);
            // (|h| < 2^14: a 24-bit multiply -- the 32-bit one issues at a quarter of the rate)
            const double B0 = (double)(__mul24(20000, h0) - G0) + cq, B1 = (double)(__mul24(20000, h1) - G1) + cq;
            const double lhs = A1 * B0, rhs = A0 * B1;
            const bool sure = (B0 < 0.0) & (B1 < 0.0) & (fabs(lhs - rhs) > 1e-9 * (fabs(lhs) + fabs(rhs)));
            flags |= sure ? 0 : kDeferTm;
            if (lhs > rhs) {
                h0 = h1;
                G0 = G1;
                predW = stk.W;
            }
        }
        // ---- loops (thal.c calc_bulge_internal acceptance: dG of the candidate strictly lower)
        if (best.G <= G0) {
            // exact enthalpy of the best candidate: the loop term's from the table that mirrors T
            // (TH's entry of table byte address A is the short at byte A / 2: one shift, the clamp in the halved domain --
            //  floor commutes with min -- and the cell-side enthalpy read beside it, unconditionally, then multiplied by
            //  the entry's flag: one LDS round trip and no exec juggling)
            const unsigned th_at = min((rc.C - (unsigned)best.W) >> 16, (unsigned)kRowTBytes >> 1) & ~1u;
            const int th = *reinterpret_cast<const short *>(reinterpret_cast<const char *>(sh.TH) + th_at);
            const int ysh = sh.ytsh[(im1 << 2) | (int)(((q.s2 << 2) >> (2 * jm1)) & 3u)];
            const int hw = (th >> 1) + __mul24(th & 1, ysh) + word_h(best.W);
            if (best.G < G0) {
                flags |= tie ? kDeferLoopTie : 0;
                // thal.c rejects a candidate with H > 0 and S > 0 (620300 S = 2000 H - G)
                // (H > 0 alone hands the pair on: S > 0 as well -- 620300 S = 20000 hw - G -- is what thal.c asks, but a
                //  winning loop candidate of positive enthalpy is rare enough that the second half is not worth
                //  three instructions on every cell; the stages behind this one apply the full rule)
                flags |= (hw > 0) ? kDeferBad : 0;
                h0 = hw;
                G0 = best.G;
                predW = best.W;
            } else if (hw == h0) {
                cell_soft = 0x100;
                if (in) sh.soft[slot >> 5][threadIdx.x] |= 1u << (slot & 31);
            } else {
                flags |= kDeferLoopEq;
            }
        }
        const int pred = word_cw(predW);   // (once, behind both choices: the word is what they select)
        const int hb = h0 + kHBias;
        flags |= ((unsigned)hb > 0x7fffu) ? kDeferReplay : 0;   // enthalpy beyond the 15-bit field: hand the pair on
        const int Wcell = in ? ((int)((unsigned)(jm1 * kRowA + (im1 << 2) + ((b.po_c >> 4) & 3)) << 17) | (hb & 0x7fff))
                             : kEmptyRowW;   // lanes without a cell here compute garbage and publish an empty slot
        defer |= in ? flags : 0;
        // an empty slot's value is never a candidate's winner (its word fails every geometry test), but it is
        // an operand of the sum that v_min_f64 orders: keep it a number the bounds above cover
        G0 = in ? G0 : 0;
        const int G0s = G0 + kRowZero;   // as the slots hold it
        // ---- terminal pick (strict minimum of dG incl. the right end term, first in slot order)
        {
            const int Gt = G0 + gR;
            if (in & (Gt < pickG)) {
                pick_g = Gt;
                sh.pick[1][threadIdx.x] = Wcell;
                sh.pick[2][threadIdx.x] = 0;
                sh.pick[3][threadIdx.x] = pred | cell_soft;
            } else if (in & (Gt == pickG)) {
                // a tie: keep the range of the tied cells' enthalpies (right end included), 16 bits each, biased
                const unsigned u = (unsigned)min(max(h0 + rh + 32768, 1), 65535), t2 = (unsigned)sh.pick[2][threadIdx.x];
                const unsigned lo = t2 ? min(t2 & 0xffffu, u) : u, hi = t2 ? max(t2 >> 16, u) : u;
                sh.pick[2][threadIdx.x] = (int)((hi << 16) | lo);   // never 0: u >= 1
            }
        }
        // ---- publish the cell (idle lanes write a slot nobody reads)
        // (the cells of the last row all take the first free slot number -- or 52, which is written nowhere --:
        //  nothing reads it, no row follows and the walk back only matches predecessors)
        if constexpr (kPin) {
            if (slot < NS) pin_publish(Ga, Wa, Gb, Wb, Gc, Wc, slot, G0s, Wcell);   // (slot 52: cells of the last row of a full table, written nowhere)
        } else if constexpr (kPin64) {
            if (slot < NS) pin_publish64(Ga, Wa, Gb, Wb, slot, G0s, Wcell);
        } else
        if (slot < 32) {   // wave-uniform slot number: one indexed register write per plane
            Ga[slot & 31] = G0s;
            Wa[slot & 31] = Wcell;
        } else if constexpr (NS == 64) {
            Gb[(slot - 32) & 31] = G0s;
            Wb[(slot - 32) & 31] = Wcell;
        } else if constexpr (NS == 48) {
            Gb[(slot - 32) & 15] = G0s;
            Wb[(slot - 32) & 15] = Wcell;
        } else {
            if (slot < 48) {
                Gb[(slot - 32) & 15] = G0s;
                Wb[(slot - 32) & 15] = Wcell;
            } else if constexpr (NS == 52) {
                if (slot < 52) {   // slot 52: the cells of the last row of a full table, written nowhere
                    Gc[(slot - 48) & 3] = G0s;
                    Wc[(slot - 48) & 3] = Wcell;
                }
            } else {
                Gc[(slot - 48) & 7] = G0s;
                Wc[(slot - 48) & 7] = Wcell;
            }
        }
        sh.pred(slot, threadIdx.x) = (unsigned char)pred;
      }
      start_im1 = row_start;
    }

    IntResult out;
    out.r.none = !active;
    out.r.dG = INFINITY;
    out.r.t = 0.0;
    out.r.conflict = false;

    const unsigned pick_ties = (unsigned)sh.pick[2][threadIdx.x];   // a second walk would be paid by the whole wave: handed on, unless (below)
    const unsigned long long softTie =
        (unsigned long long)sh.soft[0][threadIdx.x] | ((unsigned long long)sh.soft[1][threadIdx.x] << 32);

    // ---- walk from the picked cell: traceback by pointer into the LDS scratch, then replay forwards
    //      in f64 with Primer3's operation order (fillMatrix / maxTM)
    double S = 0.0;
    int H = 0, P = 0, dpath = 0;
    const int endW = sh.pick[1][threadIdx.x];
    {
        // The picked cell (which may have no slot) opens the path; the walk goes on from its predecessor byte to that
        // cell's, and so on: a step per cell of the LONGEST path of the wave, not one per slot of the table (the slot
        // words live in registers, which a lane cannot index -- but all the walk needs of a cell is its predecessor
        // byte, and the cell's slot follows from its coordinates: the first slot of its row, a wave-uniform table of
        // bytes in four scalars, plus the number of earlier columns of oligo 2 with the same base).
        unsigned rs_pack[4] = {0u, 0u, 0u, 0u};   // byte i: first slot of row i
        {
            int acc = 0;
#pragma unroll
            for (int i = 0; i < kRowK; ++i) {   // (rows beyond the oligo's length: never looked up)
                rs_pack[i >> 2] |= (unsigned)acc << (8 * (i & 3));
                const int a_row = (int)((q.s1 >> (2 * i)) & 3u);
                acc += (i < kRowK - 1) ? (int)((wmax4 >> (8 * (3 - a_row))) & 0xffu) : 0;
            }
        }
        const int end3 = sh.pick[3][threadIdx.x];
        int cw = end3 & 0xff;
        bool done = out.r.none | (cw == 0xff);
        if (!out.r.none) {
            sh.path[0][threadIdx.x] = (unsigned short)((unsigned)endW >> 17);
            P = 1;
            dpath |= (end3 & 0x100) ? kDeferPathTie : 0;
        }
        const unsigned char *const pred_bytes = reinterpret_cast<const unsigned char *>(&sh);
        constexpr int kOffLo = (int)offsetof(SharedRow, pred_lo), kOffHi = (int)offsetof(SharedRow, pred_hi) - kPredLo * kRowThreads;
        for (int step = 0; step < kPathMax; ++step) {   // wave-uniform
            if (__ballot(!done) == 0ull) break;
            int ii, jj;
            if constexpr (kCompactCw) {
                jj = (int)(((unsigned)cw * 79u) >> 10);
                ii = cw - 13 * jj;
            } else {
                jj = cw >> 4;
                ii = cw & 15;
            }
            const int b2 = 2 * jj;
            const unsigned base = (q.s2 >> b2) & 3u;            // = 3 - s1[ii]: the cell is a complementary one
            // the base in every 2-bit field below the cell's column (24-bit multiplies: the 32-bit one issues at a quarter
            // of the rate; columns 0 .. 11 are all a 13-base oligo's cell can have before it)
            unsigned rep = (unsigned)__mul24((int)base, 0x555555);
            if constexpr (kRowK > 13) rep |= (unsigned)__mul24((int)base, 0x55) << 24;
            const unsigned x = q.s2 ^ rep;
            const unsigned differ = (x | (x >> 1)) & 0x55555555u & ((1u << b2) - 1u);   // earlier columns with another base
            const int rank = jj - __popc(differ);
            const unsigned sel = 0x0c0c0c00u | (unsigned)(ii & 7);
            const unsigned rs_lo = __builtin_amdgcn_perm(rs_pack[1], rs_pack[0], sel), rs_hi = __builtin_amdgcn_perm(rs_pack[3], rs_pack[2], sel);
            const int slot = done ? 0 : (int)(ii < 8 ? rs_lo : rs_hi) + rank;
            const unsigned K = (unsigned)(jj * kRowA + 4 * ii) + ((q.s2 >> (b2 + 2)) & 3u);   // the slot word's K (n2: cell_bases' obR & 3)
            if (!done) {
                sh.path[P & (kPathMax - 1)][threadIdx.x] = (unsigned short)K;
                dpath |= ((softTie >> slot) & 1ull) != 0ull ? kDeferPathTie : 0;
                P += 1;
            }
            const int pr = pred_bytes[(slot < kPredLo ? kOffLo : kOffHi) + __mul24(slot, kRowThreads) + (int)threadIdx.x];
            cw = done ? cw : pr;
            done = done | (pr == 0xff);
        }
    }
    // The picked cell's right end terms (thal.c drawDimer adds them to the path's sums)
    const KParts pe = k_parts((unsigned)endW >> 17);
    CellCtx ecc;
    const CellBases eb = cell_bases(q, pe.ii, pe.jj, ecc);
    const int rH = sh.h[eb.idxR - kRowGBase];
    const int N = P - 1;
    // A call that asks for decisions only does not need the walked structure's sums in Primer3's own order of f64
    // additions: the DP carries them exactly.  The picked value is pick_g = 2000 (H - 310.15 S) over the path and the
    // right end, the word's enthalpy field H / 10, hence 620300 S = 20000 h - pick_g as an integer, and
    //      dG = dH - T (dS + N salt),   dH = 10 h + 200,   dS = S + init_S
    // follows with two roundings where the reference's sum has a dozen: the two differ by less than 1e-8 cal/mol.
    // Unless a lane of the wave comes closer to the cut than kCutMargin, the decision is made from that value and the
    // f64 replay below (a trip per cell of the longest path, table entries from global memory) is skipped.
    constexpr double kCutMargin = 1e-3;   // cal/mol
    bool exact = !decisions_only;
    double Gfast = 0.0;
    if (decisions_only) {
        const int ht = word_h(endW) + rH;
        const double dHf = (double)(ht * 10 + 200);
        const double dSf = (double)(__mul24(20000, ht) - pick_g) / 620300.0 + K.init_S;
        Gfast = dHf - (K.temp_k * (dSf + (N * K.salt)));
        // (an edge record carries its conflict's dG: the reference's own bits)
        exact = __ballot(!out.r.none && (fabs(Gfast - K.g_cut) < kCutMargin || (edge_values && Gfast <= K.g_cut))) != 0ull;   // wave-uniform
    }
    if (exact) {
        int prevCore = 0;
        const int maxP = wave_max_u8(P);
        for (int step_ = 0; step_ < maxP; ++step_) {
            const int step = __builtin_amdgcn_readfirstlane(step_);
            const int e = P - 1 - step;
            if (e >= 0) {
                const int core = core_of_k(sh.path[e & (kPathMax - 1)][threadIdx.x], q.s1, 0);
                CellCtx cc;
                const CellBases b = cell_bases(q, (core >> 4) & 15, core & 15, cc);
                if (step == 0) {
                    S = gS[b.idxL];
                    H = sh.h[b.idxL - kRowGBase];
                } else if (((core & 0xff) - (prevCore & 0xff)) == 0x11) {   // the cell (i-1, j-1): stacked pair
                    S = S + gS[b.wc];
                    H = H + sh.h[b.wc - kRowGBase];
                } else {
                    const CandGeom g = cand_geometry(cc, prevCore);
                    S = ((gS[g.lx] + gS[g.y]) + gS[g.zi]) + S;   // thal.c's order (pair_core.hpp cand_finish)
                    H = gH[g.lx] / 10 + sh.h[g.y - kRowGBase] + H;   // the loop entry from global, like its entropy
                }
                prevCore = core;
            }
        }
        // the replayed enthalpy must be the tracked one; anything else is handed on
        dpath |= (!out.r.none & (H != word_h(endW))) ? kDeferReplay : 0;   // H in units of 10 cal/mol here
    } else {
        H = word_h(endW);
    }
    // (a path tie is looked at again below: it only changes the number of pairs of the walked structure)
    defer |= dpath & ~kDeferPathTie;
    // ---- thal.c drawDimer(): totals
    {
        double G = Gfast, t = 0.0;
        if (exact) {
            const double rS = gS[eb.idxR];
            const double dH = (double)((H + rH) * 10 + 200);
            const double dS = (S + rS) + K.init_S;
            t = (dH / ((dS + (N * K.salt)) + K.RC)) - kAbsZero;
            G = dH - (K.temp_k * (dS + (N * K.salt)));
        }
        if (!out.r.none) {
            out.r.dG = G;
            out.r.t = t;
            out.r.conflict = G <= K.g_cut;
        }
        // int_core.hpp kPickMargin: the decision of a pair with a tied pick stands when it is not a close one
        const int ht = H + rH;   // the walked cell's (H == word_h(endW), or the pair is handed on anyway)
        const int dh_min = (int)(pick_ties & 0xffffu) - 32768 - ht, dh_max = (int)(pick_ties >> 16) - 32768 - ht;
        defer |= (pick_ties != 0u && !(decisions_only && tied_pick_cannot_conflict(K, G, N, dh_min, dh_max))) ? kDeferPick : 0;
        // a cell of the path with an equal-valued alternative of the same enthalpy: the other path gives the same
        // (dH, dS) with another N, so the same bound with an empty enthalpy range
        defer |= ((dpath & kDeferPathTie) && !(decisions_only && tied_pick_cannot_conflict(K, G, N, 0, 0))) ? kDeferPathTie : 0;
    }
    out.defer = out.r.none ? 0 : defer;
    return out;
}

// One lock-step DP of the wave: lane = (row, col), all lanes share `row`.
template <int NS>
static __device__ __forceinline__ void wave_pairs_row(SharedRow &sh, const IntArgs &a, int row, int col, uint64_t pa,
                                               uint64_t pb, bool inside)
{
    const int lane = threadIdx.x & 63;
    SeqPair q;
    unsigned rowmask;
    const int k = __builtin_amdgcn_readfirstlane(a.f.k);
    const int n_all = setup_pair(pa, pb, k, q, rowmask);
    q.s1 = (unsigned)__builtin_amdgcn_readfirstlane((int)q.s1);   // the block's row primer: a scalar
    q.len = k;
    q.lenmask = (unsigned)__builtin_amdgcn_readfirstlane((int)q.lenmask);
    const unsigned lenmask = q.lenmask;
    // A pair's table holds its complementary cells except those of the last row (run_pair_row): n_cells below
    // is that number, the quantity every sizing decision is about.
    int c2[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) c2[b] = __popc(spaced_mask(q.s2, b, lenmask));
    const int last_base = (int)((q.s1 >> (2 * (k - 1))) & 3u);   // scalar
    int n_cells = n_all - c2[3 - last_base];
    const bool sym = self_complementary(pa, k) && self_complementary(pb, k);
    // the whole table for 52 slots (the last row's cells then take the number 52 and are written nowhere); the
    // other shapes keep one slot free for them
    constexpr int kFit = NS == 52 ? NS : NS - 1;
    bool spill = inside & ((n_cells > kFit) | sym);
    unsigned flag = 0u;
    bool active = inside & !spill & (n_all > 0);   // lanes that run the DP
    if (!active) n_cells = 0;
    int nmax = wave_max_u8(n_cells);
    // Lock-step lanes pay for the largest table of their wave (work ~ slots^2).  A few lanes far above
    // the rest (mixed compositions at bin boundaries) are cheaper in a sorted list stage.
    for (int round = 0; round < 6; ++round) {
        const int next = wave_max_u8(n_cells < nmax ? n_cells : 0);
        const int m = __popcll(__ballot(active && n_cells == nmax));
        if (next == 0 || nmax * nmax - next * next <= kDragCost * m) break;   // wave-uniform
        if (active && n_cells == nmax) {
            spill = true;
            active = false;
            n_cells = 0;
        }
        nmax = next;
    }
    // Slots the wave needs: row i (but the last) takes the widest lane's count of base 3 - s1[i].  If that is
    // more than the table holds, the lanes with the most cells leave for the list stage until it fits.
    const unsigned rowsmask = lenmask >> 2;   // rows 0 .. k - 2
    int w4[4] = {0, 0, 0, 0}, n_slots = 0;
    for (;;) {
        if (__ballot(active) == 0ull) break;   // wave-uniform
        n_slots = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            w4[b] = wave_max_u8(active ? c2[b] : 0);
            n_slots += w4[b] * __popc(spaced_mask(q.s1, 3 - b, rowsmask));   // stored rows whose cells have base b on oligo 2
        }
        if (n_slots <= kFit) break;
        // Too many slots: the wave mixes compositions (it straddles a bin boundary, or the pool is small).
        // Keep the larger party: lanes with the composition of the first active lane stay if they are at
        // least half of the active lanes, else they are the ones to go.  (Lanes of one composition need
        // exactly their own cell count, which fits: larger tables left above.)
        const unsigned sig = (unsigned)c2[0] | ((unsigned)c2[1] << 8) | ((unsigned)c2[2] << 16) | ((unsigned)c2[3] << 24);
        const unsigned long long act = __ballot(active);
        const unsigned sig0 = (unsigned)__shfl((int)sig, __ffsll((long long)act) - 1);
        const unsigned long long same = __ballot(active && sig == sig0);
        const bool keep_same = 2 * __popcll(same) >= __popcll(act);
        const bool uniform = same == act;   // cannot happen with n_slots > NS; never loop on it
        if (active && (uniform ? n_cells == nmax : ((sig == sig0) != keep_same))) {
            spill = true;
            active = false;
            n_cells = 0;
        }
        nmax = wave_max_u8(n_cells);
    }
    if (__ballot(active) == 0ull) {   // wave-uniform: nothing to compute
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
    const unsigned wmax4 = (unsigned)w4[0] | ((unsigned)w4[1] << 8) | ((unsigned)w4[2] << 16) | ((unsigned)w4[3] << 24);
    const bool decisions_only = a.f.sinks.dg == nullptr && a.f.sinks.tm == nullptr;   // wave-uniform
    const IntResult r = run_pair_row<NS>(sh, a.f.c, a.f.ft->S, a.f.ft->H, q, active, wmax4, n_slots, decisions_only,
                                         a.f.sinks.edge_count != nullptr);
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
        const unsigned long long bits = __ballot(hit);
        if (lane == 0 && bits) atomicAdd(&a.f.sinks.row_conflicts[row], (unsigned)__popcll(bits));
    }
    if (live) {
        if (a.f.sinks.dg) a.f.sinks.dg[orow * (size_t)a.f.sinks.ncols + ocol] = r.r.dG;
        if (a.f.sinks.tm) a.f.sinks.tm[orow * (size_t)a.f.sinks.ncols + ocol] = r.r.t;
    }
}

// The loop table of row primer s1 (2 bits per base): for every (l2, i, r = i - ii, n2) the loop term of
// the general table (fast_tables.hpp IntTables::T, row d = 16 l1 + l2, l1 = r - 1) with the column that
// oligo 1 determines, plus the cell-side term where it needs no base of oligo 2 beyond n2:
//   bulges          column a_p | a_c << 2 (the general table holds both closing pairs)
//   l2 == 1         the base between predecessor and cell on oligo 2 IS n2: 1 x 1 loops get their
//                   cell-side mismatch, interior loops their cell-side terminal mismatch
//   l2 >= 2, r >= 2 pred side only; the lane adds yts[i][m2], m2 = the base left of the cell on oligo 2.
// The lane adds that term to EVERY entry of the rows r >= 2 (one add, no select per visit), so the entries of those
// rows that must not get it hold the loop term minus it: their m2 is known here -- l2 == 1: m2 is n2; l2 == 0: the
// cell sits right of the predecessor, whose base on oligo 2 is the complement of oligo 1's at ii.  The row above the
// cell (r == 1) takes no cell-side term at all.
// r == 0 (same row), ii < 0, the stacked pair and loops the chemistry has no entry for: not available.
static __device__ __forceinline__ void build_row_table(SharedRow &sh, const IntArgs &a, unsigned s1)
{
    const int32_t *Tg = a.it->T;
    const int32_t *Hg = a.f.ft->H;
    if (threadIdx.x < 64) {
        const int i = threadIdx.x >> 2, m2 = threadIdx.x & 3;
        const int a_c = (int)((s1 >> (2 * i)) & 3u), m1 = i > 0 ? (int)((s1 >> (2 * i - 2)) & 3u) : 0;
        {
            const int y = sh.g[FastTables::kTSc - kRowGBase + (((3 - a_c) * 4 + m2) * 4 + m1)];
            sh.yts[threadIdx.x] = y >= IntTables::kValid ? kRowU : y;   // void stays void, and in range (kRowD)
        }
        sh.ytsh[threadIdx.x] = sh.h[FastTables::kTSc - kRowGBase + (((3 - a_c) * 4 + m2) * 4 + m1)];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kRowTEntries; e += kRowThreads) {
        const int l2 = e / kRowA, rem = e - l2 * kRowA;
        const int n2 = 3 - (rem & 3), ir = rem >> 2, i = ir / kRowR, r = ir - i * kRowR;
        const int ii = i - r, l1 = r - 1;
        int v = IntTables::kBig, hv = 0;
        bool needs_y = false;
        if (r >= 1 && ii >= 0 && i < a.f.k && (l1 | l2) != 0 && l1 <= IntTables::kMaxL) {
            const int d = l1 * 16 + l2;
            const int a_p = (int)((s1 >> (2 * ii)) & 3u), a_c = (int)((s1 >> (2 * i)) & 3u);
            const int sz = l1 + l2;
            if (l1 == 0 || l2 == 0) {
                v = Tg[d * 64 + (a_p | (a_c << 2))];
                hv = Hg[FastTables::kBU + a_c * FastTables::kBUStride + sz * 4 + a_p];
            } else {
                const int n1 = (int)((s1 >> (2 * ii + 2)) & 3u);
                const int po = a_p | (n1 << 2) | (n2 << 4);
                v = Tg[d * 64 + po];
                hv = Hg[FastTables::kNB + (sz - 2) * 64 + po];
                if (l2 == 1) {
                    const int m1 = (int)((s1 >> (2 * i - 2)) & 3u);
                    const int ci = ((3 - a_c) * 4 + n2) * 4 + m1;
                    const int ye = (d == 0x11 ? FastTables::kMMc : FastTables::kTSc) + ci;
                    const int y = sh.g[ye - kRowGBase];
                    v = (v >= IntTables::kValid || y >= IntTables::kValid) ? IntTables::kBig : v + y;
                    hv += sh.h[ye - kRowGBase] * 10;
                } else {
                    needs_y = v < IntTables::kValid;
                }
            }
            if (v >= IntTables::kValid) hv = 0;
        }
        if (v >= IntTables::kValid) {
            v = kRowU;
        } else if (!needs_y && r >= 2) {
            // the lane will add yts[i][m2] to this entry although the loop has no such term: take it out here
            const int m2 = l2 == 1 ? n2 : 3 - (int)((s1 >> (2 * ii)) & 3u);
            v -= sh.yts[(i << 2) | m2];
        }
        sh.T[e] = v + kRowD - kRowZero;   // "not available" is 0
        sh.TH[e] = (short)(((hv / 10) << 1) | (needs_y ? 1 : 0));
    }
    if (threadIdx.x < 4) {
        sh.T[kRowTEntries + threadIdx.x] = 0;
        sh.TH[kRowTEntries + threadIdx.x] = 0;
    }
}

template <int NS>
static __device__ __forceinline__ void kernel_body(const IntArgs &a)
{
    __shared__ SharedRow sh;
    for (int e = threadIdx.x; e < kRowGCount; e += kRowThreads) {
        sh.h[e] = a.f.ft->H[kRowGBase + e] / 10;   // finite entries are multiples of 10 (build_int_tables); "not available" stays huge
        sh.g[e] = a.it->g[kRowGBase + e];
    }
    for (int e = threadIdx.x; e < 100; e += kRowThreads)
        sh.cq[e] = 620300.0 * ((a.f.c.init_S + a.f.ft->S[FastTables::kEndR + e]) + a.f.c.RC);
    const int lane = threadIdx.x & 63;
    const int ncolg = (a.f.col1 - a.f.col0 + 63) >> 6;
    const int n_seg = (ncolg + kSegGroups - 1) / kSegGroups;
    const unsigned n_items = (unsigned)n_seg * (unsigned)(a.f.row1 - a.f.row0);
    int built_row = -1;
    // Work items (one row x up to kSegGroups column groups) are handed to the BLOCKS from a counter; inside
    // an item the column groups are handed to the waves from a counter in LDS, so that a block's
    // waves need not march together (the cost of a group varies with its compositions).
    for (;;) {
        if (threadIdx.x == 0) {
            sh.item = (int)atomicAdd(a.work_counter, 1u);
            sh.next_group = 0u;
        }
        __syncthreads();
        const unsigned item = (unsigned)sh.item;
        if (item >= n_items) break;   // block-uniform
        const int row = a.f.row0 + (int)(item / (unsigned)n_seg), seg = (int)(item % (unsigned)n_seg);
        const uint64_t pa = a.f.pool[row];
        if (row != built_row) {
            const unsigned lenmask = (1u << (2 * a.f.k)) - 1u;
            build_row_table(sh, a, (unsigned)pa & lenmask);
            built_row = row;
        }
        __syncthreads();
        const unsigned g_lo = (unsigned)seg * kSegGroups;
        const unsigned g_hi = min((unsigned)ncolg, g_lo + (unsigned)kSegGroups);
        for (;;) {
            unsigned grp = 0;
            if (lane == 0) grp = atomicAdd(&sh.next_group, 1u);
            grp = g_lo + (unsigned)__builtin_amdgcn_readfirstlane((int)grp);
            if (grp >= g_hi) break;   // wave-uniform
            // (the fetch relies on the whole wave arriving here together: wave_pairs_row returns only through
            // wave-uniform branches -- keep it that way, or a lane runs ahead of the readfirstlane)
            const int cq = a.f.col0 + (int)grp * 64 + lane;
            const bool inside = cq < a.f.col1;
            const uint64_t pb = a.f.cols_sorted[inside ? cq : a.f.col0];
            const int col = (int)a.f.perm[inside ? cq : a.f.col0];
            wave_pairs_row<NS>(sh, a, row, col, pa, pb, inside);
        }
        __syncthreads();   // every wave is done with the table (and with sh.item) before the next item
    }
}

};   // struct RowKernel

using Row13 = RowKernel<13, 768, 52, true>;
using Row14 = RowKernel<14, 512, 64, false>;
using Row15 = RowKernel<15, 512, 64, false>;
static_assert(Row13::div_magic_ok() && Row14::div_magic_ok() && Row15::div_magic_ok(), "K / kRowA by multiply-shift");
static_assert(Row13::cw_ok() && Row14::cw_ok() && Row15::cw_ok(), "the predecessor byte names every cell, and 0xff none");
static_assert(Row13::kCompactCw && !Row15::kCompactCw, "13 bases: the three-instruction byte");
static_assert(Row13::kRowA == 772 && Row14::kRowA == 900 && Row15::kRowA == 964, "table strides");

template <class RK>
__global__ void __launch_bounds__(RK::kRowThreads) k_pairs_row(IntArgs a)
{
    RK::template kernel_body<RK::kRowSlots>(a);
}

}  // namespace

int pairs_row_max_k() { return Row15::kRowK; }
int pairs_row_oob_max_k() { return Row13::kRowK; }   // up to here the instance without an address clamp runs (it needs the probe below)

namespace {
// The probe behind pairs_row_lds_reads_zero(), two launches.
//   k_lds_paint: one block per CU with the largest LDS allocation a block can have writes a non-zero pattern
//       over the CU's whole LDS (the blocks wait for each other, bounded, so that they spread over all CUs).
//   k_lds_probe: blocks with the ROW KERNEL'S OWN static allocation (the same `__shared__ SharedRow`) paint
//       their allocation as well and read every address a wrapped table address can produce.  What lies beyond
//       the allocation but inside the CU's LDS still holds k_lds_paint's pattern, so a device that did not
//       bounds-check the read would return it; every read must return 0.
constexpr unsigned kLdsPaintBytes = 160 * 1024;
__device__ __forceinline__ void wait_for_peers(unsigned *arrived, unsigned want)
{
    // bounded: the point is to keep the block on its CU while the others are placed, not a barrier anyone
    // depends on (a busy device may not hold all the blocks at once)
    if (threadIdx.x == 0) {
        atomicAdd(arrived, 1u);
        const unsigned long long t0 = wall_clock64();   // 100 MHz
        while (atomicAdd(arrived, 0u) < want && wall_clock64() - t0 < 20000ull) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}
__global__ void __launch_bounds__(256) k_lds_paint(unsigned *arrived)
{
    extern __shared__ unsigned paint[];
    for (unsigned e = threadIdx.x; e < kLdsPaintBytes / 4; e += blockDim.x) paint[e] = 0xa5a5a5a5u;
    __syncthreads();
    wait_for_peers(arrived, gridDim.x);
    if (paint[(threadIdx.x * 97u) % (kLdsPaintBytes / 4)] != 0xa5a5a5a5u) __builtin_trap();   // keeps the stores
}
__global__ void __launch_bounds__(Row13::kRowThreads) k_lds_probe(unsigned lo, unsigned hi, unsigned *arrived, unsigned *nonzero)
{
    __shared__ Row13::SharedRow sh;
    unsigned *fill = (unsigned *)&sh;
    for (unsigned e = threadIdx.x; e < sizeof(Row13::SharedRow) / 4; e += blockDim.x) fill[e] = 0xffffffffu;
    __syncthreads();
    wait_for_peers(arrived, gridDim.x);
    unsigned bad = 0;
    for (unsigned addr = lo + 4u * threadIdx.x; addr < hi; addr += 4u * blockDim.x) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        bad |= v;
    }
    if (bad) atomicOr(nonzero, 1u);
    if (fill[threadIdx.x] != 0xffffffffu) atomicOr(nonzero, 2u);
}
}  // namespace

// The row kernel has no address clamp (see kRowZero): checked once per engine, on the engine's device.
hipError_t pairs_row_lds_reads_zero(hipStream_t stream, int n_cu, bool *ok)
{
    *ok = false;
    {   // the kernel's real LDS size (the static_asserts speak for SharedRow; a compiler could add to it), and the
        // probe's must be the same
        hipFuncAttributes fa, fp;
        hipError_t ea = hipFuncGetAttributes(&fa, (const void *)k_pairs_row<Row13>);
        if (ea == hipSuccess) ea = hipFuncGetAttributes(&fp, (const void *)k_lds_probe);
        if (ea != hipSuccess) return ea;
        if (fa.sharedSizeBytes != fp.sharedSizeBytes) return hipSuccess;
        if (offsetof(Row13::SharedRow, T) + (size_t)Row13::kRowWrapMin < (fa.sharedSizeBytes + 1279) / 1280 * 1280) return hipSuccess;
    }
    unsigned *d_flag = nullptr, h_flag[3] = {0, 0, 1};
    hipError_t e = hipMalloc((void **)&d_flag, 3 * sizeof(unsigned));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_flag, 0, 3 * sizeof(unsigned), stream);
    if (e == hipSuccess) {
        const unsigned lo = (unsigned)offsetof(Row13::SharedRow, T) + (unsigned)Row13::kRowWrapMin;
        const unsigned hi = (unsigned)offsetof(Row13::SharedRow, T) + (1u << 17);
        const int grid = n_cu > 0 ? n_cu : 256;
        // (a runtime that refuses a block of the CU's whole LDS leaves the probe without the painted background:
        //  what it reads beyond its allocation is then whatever earlier kernels left there)
        if (hipFuncSetAttribute((const void *)k_lds_paint, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kLdsPaintBytes) == hipSuccess)
            hipLaunchKernelGGL(k_lds_paint, dim3(grid), dim3(256), kLdsPaintBytes, stream, d_flag);
        (void)hipGetLastError();
        hipLaunchKernelGGL(k_lds_probe, dim3(grid), dim3(Row13::kRowThreads), 0, stream, lo, hi, d_flag + 1, d_flag + 2);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_flag, d_flag, sizeof h_flag, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_flag);
    if (e != hipSuccess) return e;
    *ok = h_flag[2] == 0;
    return hipSuccess;
}

// The row kernel's running minimum needs every value a cell can publish to be a reachable one (kRowD above):
// the end terms and the stacked-pair terms must all exist.  Primer3's parameter set has them all; a
// chemistry that does not goes to the general integer kernel.
bool pairs_row_tables_ok(const IntTables &it)
{
    if (!it.usable) return false;
    for (int e = FastTables::kEndL; e < FastTables::kWC + 16; ++e)
        if (it.g[e] >= IntTables::kValid) return false;
    return true;
}

hipError_t launch_pairs_row(const PairKernelArgs &a, const IntTables *it, unsigned long long *reasons, int n_cu,
                            hipStream_t stream)
{
    if (a.k > Row15::kRowK || a.k < 2) return hipErrorInvalidValue;
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
    const long ncolg = (a.col1 - a.col0 + 63) / 64;
    const long items = ((ncolg + Row13::kSegGroups - 1) / Row13::kSegGroups) * (long)(a.row1 - a.row0);
    if (items <= 0) return hipSuccess;
    if (hipError_t e = hipMemsetAsync(a.work_counter, 0, sizeof(unsigned), stream); e != hipSuccess) return e;
    // one persistent block per CU (about 153 KB of LDS each)
    const int grid = (int)(items < (long)n_cu ? items : (long)n_cu);
    // up to 13 bases: three waves per SIMD, 52 stored cells, no address clamp; 14 / 15 bases: two waves, 56 / 64 cells
    if (a.k <= Row13::kRowK) hipLaunchKernelGGL((k_pairs_row<Row13>), dim3(grid), dim3(Row13::kRowThreads), 0, stream, x);
    else if (a.k <= Row14::kRowK) hipLaunchKernelGGL((k_pairs_row<Row14>), dim3(grid), dim3(Row14::kRowThreads), 0, stream, x);
    else hipLaunchKernelGGL((k_pairs_row<Row15>), dim3(grid), dim3(Row15::kRowThreads), 0, stream, x);
    return hipGetLastError();
}

}  // namespace msspe
