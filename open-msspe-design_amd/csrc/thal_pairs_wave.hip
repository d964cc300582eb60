// thal_pairs_wave.hip -- thal ANY in f64 with ONE WAVE PER PAIR: the kernel behind the exact-integer
// stages for oligos of any length up to 32 bases.
//
// Job: the pairs the integer kernels hand on (exact ties, rejected minima: about 1 % of the pairs of
// long oligos), and whole blocks of the pair matrix when the integer preconditions do not hold
// (29 .. 32 bases, parameter files off the 0.01 e.u. grid).  Same outputs as every other pair
// kernel (the reference's "format N^2 lines -> ntthal -> parse" loop,
// /root/reference/od-msspe/src/delta_g.rs:61-153; Primer3 2.6.1 thal() restated from SURVEY.md
// Appendix C.3), same f64 operation order as thal_pairs.hip (file built with -ffp-contract=off), so
// dS, dH, dG and t carry the bits of the CPU oracle.
//
// The dense generic kernel keeps a k x k plane per LANE in global memory and is bound by that
// traffic (0.4 .. 2 M pairs/s).  Here the 64 lanes of a wave share one pair:
//   * the complementary cells (about k^2 / 4) are listed in row-major order in a wave-private LDS
//     table (S f64, H int32, po | coordinates);
//   * cell c: lane l evaluates predecessors l, l + 64, ... < c (split_tables.hpp's folded S / H
//     planes, five LDS gathers each), the wave reduces to the minimum dG with Primer3's visiting
//     order as tie-break (key = loop size, then row distance), and maxTM / the acceptance test run
//     once per wave;
//   * terminal pick and thal.c's value-matching traceback are the same parallel sweep + reduction.
// Pairs of two self-complementary oligos (other RC constant) and pairs with more cells than the LDS
// table holds go to the next list, for the dense kernel.
#include <cstring>

#include "split_core.hpp"

namespace msspe {

namespace {

constexpr int kWaveCells = 384;      // cells per pair in LDS (random 32-mers: 256 +- 14); two 8-wave blocks per CU
constexpr int kWavesPerBlock = 8;
constexpr int kThreadsW = 64 * kWavesPerBlock;

struct SharedWv {
    double S[W_::kCount];
    int H[W_::kCount];
    double cS[kWavesPerBlock][kWaveCells];
    int cH[kWavesPerBlock][kWaveCells];
    unsigned short cW[kWavesPerBlock][kWaveCells];   // po << 10 | im1 << 5 | jm1
};

struct WaveArgs {
    const SplitTables *st;
    ThalConsts c;
    const uint64_t *pool;
    int k;
    int row0, row1, col0, col1;      // matrix mode (in_list == nullptr): the block of the pair matrix
    const uint2 *in_list;            // list mode: explicit pairs (bit 31 of .x is a mark, ignored)
    const uint32_t *in_count;
    uint32_t in_cap;
    PairSinks sinks;
    uint2 *ovf_list;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
    // self mode (stage B, libprimer3 align_thermod of an oligo with itself; on when self_any or self_end is
    // set): work item w = pair (row0 + w, row0 + w), or (in_list[w].x, in_list[w].x) with a list; ONE fill,
    // then thal ANY -> self_any[row] = max(0, t) and thal END1 -> self_end[row] (either may be null)
    double *self_any, *self_end;
    unsigned *work_counter;          // next work item, zero at launch
};

// all lanes of the wave end up with the same value
__device__ __forceinline__ int wave_sum(int v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Minimum of (G, key) over the wave, smaller key first among equal G; returns the winning lane.
__device__ __forceinline__ int wave_argmin(double G, unsigned key)
{
    int who = threadIdx.x & 63;
    for (int off = 32; off > 0; off >>= 1) {
        const double oG = __shfl_xor(G, off);
        const unsigned oK = (unsigned)__shfl_xor((int)key, off);
        const int oW = __shfl_xor(who, off);
        // (G, key, lane) lexicographic: the lane breaks what the key leaves equal, so that every
        // lane converges on the same winner
        const bool take = (oG < G) | ((oG == G) & ((oK < key) | ((oK == key) & (oW < who))));
        G = take ? oG : G;
        key = take ? oK : key;
        who = take ? oW : who;
    }
    return who;
}

struct WaveResult {
    double dG, t;
    bool none, conflict;
};

// Candidate of the loop closed by predecessor p (word Wp, value Sp / Hp) and the cell.
struct CandW {
    double S;
    int H;
    bool ok, isStack;
    unsigned key;
};
__device__ __forceinline__ CandW candidate(const SharedWv &sh, const CellS &b, int max_loop, int Wp, double Sp,
                                           int Hp)
{
    const LoopIx g = loop_indices(b, Wp);
    CandW r;
    r.S = ((sh.S[g.lx] + sh.S[g.y]) + sh.S[g.zi]) + Sp;
    r.H = sh.H[g.lx] + sh.H[g.y] + Hp;
    const bool bad = (r.H > 0) & (r.S > 0.0);   // thal.c's both-positive rule; also unavailable entries
    r.isStack = (g.l1 | g.l2) == 0;
    r.ok = (g.l1 >= 0) & (g.l2 >= 0) & !r.isStack & (g.l1 + g.l2 <= max_loop) & !bad;
    r.key = (unsigned)((g.l1 + g.l2) * 32 + g.l1);
    return r;
}

// thal.c fillMatrix() for one pair, computed by the whole wave: the cells' values in the wave's LDS table.
// Returns the number of cells, or -1 when the pair does not fit the table.
__device__ int fill_pair_wave(SharedWv &sh, int wave, const ThalConsts &K, uint64_t pa, uint64_t pb, int k, SeqW &q)
{
    const int lane = threadIdx.x & 63;
    double *cS = sh.cS[wave];
    int *cH = sh.cH[wave];
    unsigned short *cW = sh.cW[wave];
    unsigned long long rowmask;
    const int n = setup_pair_w(pa, pb, k, q, rowmask);
    if (n > kWaveCells) return -1;
    if (n == 0) return 0;
    // ---- the cells in row-major order: lane = row, exclusive scan of the row lengths
    {
        unsigned long long m = 0ull;
        if (lane < k) m = spaced_mask64(q.s2, 3 - (int)((q.s1 >> (2 * lane)) & 3), q.lenmask);
        const int cnt = __popcll(m);
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        int at = incl - cnt;
        while (m) {
            const int jm1 = (__ffsll((long long)m) - 1) >> 1;
            m &= m - 1;
            const CellS b = cell_s(q, lane, jm1);
            cW[at++] = (unsigned short)((b.po_c << 10) | (lane << 5) | jm1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- thal.c fillMatrix(): cells in order, predecessors in parallel
    for (int c_ = 0; c_ < n; ++c_) {
        const int c = __builtin_amdgcn_readfirstlane(c_);
        const int Wc = cW[c];
        const CellS b = cell_s(q, (Wc >> 5) & 31, Wc & 31);
        const double rS = sh.S[b.idxR];
        const int rH = sh.H[b.idxR];
        double bestG = INFINITY, bestS = 0.0, stS = 0.0;
        int bestH = 0, stH = 0;
        unsigned bestKey = 0xffffffffu;
        bool stMine = false;
        for (int p = lane; p < c; p += 64) {
            const int Wp = cW[p];
            const double Sp = cS[p];
            const int Hp = cH[p];
            const CandW r = candidate(sh, b, K.max_loop, Wp, Sp, Hp);
            const double G1 = (double)(r.H + rH) - kT37 * (r.S + rS);
            const bool better = r.ok & ((G1 < bestG) | ((G1 == bestG) & (r.key < bestKey)));
            bestG = better ? G1 : bestG;
            bestS = better ? r.S : bestS;
            bestH = better ? r.H : bestH;
            bestKey = better ? r.key : bestKey;
            stS = r.isStack ? Sp : stS;
            stH = r.isStack ? Hp : stH;
            stMine |= r.isStack;
        }
        const int who = wave_argmin(bestG, bestKey);
        bestG = __shfl(bestG, who);
        bestS = __shfl(bestS, who);
        bestH = __shfl(bestH, who);
        const unsigned long long stMask = __ballot(stMine);
        // ---- thal.c maxTM(): helix extension if it raises Tm
        double S0 = sh.S[b.idxL];
        int H0 = sh.H[b.idxL];
        if (stMask) {   // wave-uniform
            const int sl = __ffsll((long long)stMask) - 1;
            stS = __shfl(stS, sl);
            stH = __shfl(stH, sl);
            const double T0 = (double)(H0 + 200 + rH) / (((S0 + K.init_S) + rS) + K.RC);
            const double S1 = stS + sh.S[b.wc];
            const int H1 = stH + sh.H[b.wc];
            const double T1 = (double)(H1 + 200 + rH) / (((S1 + K.init_S) + rS) + K.RC);
            if (T1 > T0) {
                S0 = S1;
                H0 = H1;
            }
        }
        // ---- loops (calc_bulge_internal acceptance: dG of the candidate strictly lower)
        const double G2 = (double)(H0 + rH) - kT37 * (S0 + rS);
        if (bestG < G2) {
            S0 = bestS;
            H0 = bestH;
        }
        if (lane == 0) {
            cS[c] = S0;
            cH[c] = H0;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return n;
}

// Terminal pick (thal ANY, or END1: the last row only), thal.c's value-matching traceback and the totals over a
// filled table of n > 0 cells.
__device__ void finish_pair_wave(const SharedWv &sh, int wave, const ThalConsts &K, const SeqW &q, int k, int n, bool end1,
                                 WaveResult &out)
{
    const int lane = threadIdx.x & 63;
    const double *cS = sh.cS[wave];
    const int *cH = sh.cH[wave];
    const unsigned short *cW = sh.cW[wave];
    out.none = false;
    out.dG = INFINITY;
    out.t = 0.0;
    out.conflict = false;
    // ---- terminal pick: strict minimum of the nudged dG, first in row-major order
    double pickG = INFINITY;
    int pickSlot = 0x7fffffff;
    for (int p = lane; p < n; p += 64) {
        const int Wp = cW[p];
        const CellS b = cell_s(q, (Wp >> 5) & 31, Wp & 31);
        const double rSn = sh.S[b.idxR] + kTiny, rHn = (double)sh.H[b.idxR] + kTiny;
        const double Gt = (((double)cH[p] + rHn) + K.init_H) - kT37 * ((cS[p] + rSn) + K.init_S);
        // thal END1: only structures that close on the 3' base of oligo 1 (the last row)
        const bool pick = (Gt < pickG) & (!end1 | (((Wp >> 5) & 31) == k - 1));
        pickG = pick ? Gt : pickG;
        pickSlot = pick ? p : pickSlot;
    }
    {
        const int who = wave_argmin(pickG, (unsigned)pickSlot);
        pickG = __shfl(pickG, who);
        pickSlot = __shfl(pickSlot, who);
    }
    if (!(pickG < INFINITY)) {
        // no candidate: thal() falls back to cell (1, 1) and reports no structure unless that cell
        // is a base pair (only reachable in END1 mode, where the last row may hold no cell)
        if ((cW[0] & 0x3ff) != 0) {
            out.none = true;
            return;
        }
        pickSlot = 0;
    }
    const double pickS = cS[pickSlot];
    const int pickH = cH[pickSlot];
    // ---- thal.c traceback(): follow the first candidate (stack, then loops in visiting order) that
    //      reproduces the cell's value; count the base pairs
    int cur = pickSlot, P = 1;
    for (int step = 0; step < 2 * 32 + 2; ++step) {
        const int Wc = cW[cur];
        const double curS = cS[cur];
        const int curH = cH[cur];
        const CellS b = cell_s(q, (Wc >> 5) & 31, Wc & 31);
        if ((sh.H[b.idxL] == curH) & (fabs(curS - sh.S[b.idxL]) < 1e-5)) break;   // wave-uniform
        const double wcS = sh.S[b.wc];
        const int wcH = sh.H[b.wc];
        unsigned hitKey = 0xffffffffu;
        int hitSlot = 0;
        for (int p = lane; p < cur; p += 64) {
            const int Wp = cW[p];
            const double Sp = cS[p];
            const int Hp = cH[p];
            const CandW r = candidate(sh, b, K.max_loop, Wp, Sp, Hp);
            const double candS = r.isStack ? wcS + Sp : r.S;
            const int candH = r.isStack ? wcH + Hp : r.H;
            const unsigned key = r.isStack ? 0u : r.key;
            const bool hit = (r.ok | r.isStack) & (candH == curH) & (fabs(curS - candS) < 1e-5) & (key < hitKey);
            hitKey = hit ? key : hitKey;
            hitSlot = hit ? p : hitSlot;
        }
        const int who = wave_argmin(0.0, hitKey);
        hitKey = (unsigned)__shfl((int)hitKey, who);
        hitSlot = __shfl(hitSlot, who);
        if (hitKey == 0xffffffffu) break;   // wave-uniform
        cur = hitSlot;
        ++P;
    }
    // ---- thal.c drawDimer(): totals
    {
        const int Wp = cW[pickSlot];
        const CellS b = cell_s(q, (Wp >> 5) & 31, Wp & 31);
        const double rS = sh.S[b.idxR];
        const int rH = sh.H[b.idxR];
        const double dH = (double)(pickH + rH + 200);
        const double dS = (pickS + rS) + K.init_S;
        const int N = P - 1;
        out.t = (dH / ((dS + (N * K.salt)) + K.RC)) - kAbsZero;
        out.dG = dH - (K.temp_k * (dS + (N * K.salt)));
        out.conflict = out.dG <= K.g_cut;
    }
}

__global__ void __launch_bounds__(kThreadsW) k_pairs_wave(WaveArgs a)
{
    __shared__ SharedWv sh;
    for (int e = threadIdx.x; e < W_::kCount; e += kThreadsW) {
        sh.S[e] = a.st->S[e];
        sh.H[e] = a.st->H[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ncols = a.col1 - a.col0;
    const bool self = a.self_any || a.self_end;
    const long n_work = a.in_list ? (long)min(*a.in_count, a.in_cap)
                                  : (self ? (long)(a.row1 - a.row0) : (long)(a.row1 - a.row0) * ncols);
    for (;;) {
        // pairs are handed to the waves from a counter: their cost ranges over an order of magnitude
        unsigned next = 0;
        if (lane == 0) next = atomicAdd(a.work_counter, 1u);
        const long w = (long)(unsigned)__builtin_amdgcn_readfirstlane((int)next);
        if (w >= n_work) break;   // wave-uniform
        int row, col;
        if (a.in_list) {
            const uint2 pr = a.in_list[w];
            row = (int)(pr.x & 0x7fffffffu);
            col = self ? row : (int)pr.y;
        } else if (self) {
            row = col = a.row0 + (int)w;
        } else {
            row = a.row0 + (int)(w / ncols);
            col = a.col0 + (int)(w % ncols);
        }
        const uint64_t pa = a.pool[row], pb = a.pool[col];
        const bool sym = self_complementary(pa, a.k) && self_complementary(pb, a.k);
        WaveResult r;
        r.none = true;
        r.conflict = false;
        r.dG = INFINITY;
        r.t = 0.0;
        SeqW q;
        const int n_cells = sym ? -1 : fill_pair_wave(sh, wave, a.c, pa, pb, a.k, q);   // wave-uniform
        const bool fits = n_cells >= 0;
        WaveResult r_end = r;
        if (n_cells > 0) {
            if (!self || a.self_any) finish_pair_wave(sh, wave, a.c, q, a.k, n_cells, false, r);
            if (self && a.self_end) finish_pair_wave(sh, wave, a.c, q, a.k, n_cells, true, r_end);
        }
        // lane 0 reports; no lane may run ahead into the next fetch (readfirstlane reads the first
        // ACTIVE lane), so there is no early `continue` here: the wave reconverges at the loop's end
        if (lane == 0) {
            if (!fits) {
                const uint32_t at = atomicAdd(a.ovf_count, 1u);
                if (at < a.ovf_cap) a.ovf_list[at] = make_uint2((unsigned)row, (unsigned)col);
            } else if (self) {
                if (a.self_any) a.self_any[row] = (r.none || r.t < 0.0) ? 0.0 : r.t;   // libprimer3 align_thermod()
                if (a.self_end) a.self_end[row] = (r_end.none || r_end.t < 0.0) ? 0.0 : r_end.t;
            } else {
                const size_t orow = (size_t)(row - a.sinks.row0);
                const size_t ocol = (size_t)(col - a.sinks.col0);
                if (r.conflict) {
                    if (a.sinks.bitmap)
                        atomicOr((unsigned long long *)&a.sinks.bitmap[orow * (size_t)a.sinks.words + (ocol >> 6)],
                                 1ull << (ocol & 63));
                    if (a.sinks.row_conflicts) atomicAdd(&a.sinks.row_conflicts[row], 1u);
                    sink_edge(a.sinks, row, col, r.dG);
                }
                if (a.sinks.dg) a.sinks.dg[orow * (size_t)a.sinks.ncols + ocol] = r.dG;
                if (a.sinks.tm) a.sinks.tm[orow * (size_t)a.sinks.ncols + ocol] = r.t;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// in_list == nullptr: matrix mode over rows [a.row0, a.row1) x pool columns [a.col0, a.col1).
hipError_t launch_pairs_wave(const PairKernelArgs &a, const SplitTables *st, const uint2 *in_list,
                             const uint32_t *in_count, hipStream_t stream)
{
    WaveArgs x;
    x.st = st;
    x.c = a.c;
    x.pool = a.pool;
    x.k = a.k;
    x.row0 = a.row0;
    x.row1 = a.row1;
    x.col0 = a.col0;
    x.col1 = a.col1;
    x.in_list = in_list;
    x.in_count = in_count;
    x.in_cap = a.overflow_cap;
    x.sinks = a.sinks;
    x.ovf_list = a.overflow_list;
    x.ovf_count = a.overflow_count;
    x.ovf_cap = a.overflow_cap;
    x.self_any = x.self_end = nullptr;
    x.work_counter = a.work_counter;
    if (!in_list && ((long)(a.row1 - a.row0) * (long)(a.col1 - a.col0) <= 0)) return hipSuccess;
    if (hipError_t e = hipMemsetAsync(a.work_counter, 0, sizeof(unsigned), stream); e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pairs_wave, dim3(256 * 2), dim3(kThreadsW), 0, stream, x);
    return hipGetLastError();
}

// Stage B: thal ANY and / or thal END1 of oligos with themselves from ONE fill per oligo, self_any[row] /
// self_end[row] = max(0, t) (either may be null).  The oligos: [row0, row1) of the pool, or -- in_list set -- the
// entries' .x (what launch_self_lists left over).  Oligos it does not take (self-complementary ones, oversized
// tables) are appended to list as (row, row) for launch_dimer_generic.
hipError_t launch_self_wave(const SplitTables *st, const ThalConsts &c, const uint64_t *pool, int k, int row0,
                            int row1, double *self_any, double *self_end, const uint2 *in_list,
                            const uint32_t *in_count, uint2 *list, uint32_t *list_count, uint32_t list_cap,
                            uint32_t *work_counter, hipStream_t stream)
{
    WaveArgs x;
    std::memset(&x, 0, sizeof x);
    x.st = st;
    x.c = c;
    x.pool = pool;
    x.k = k;
    x.row0 = row0;
    x.row1 = row1;
    x.col0 = 0;
    x.col1 = 1;
    x.in_list = in_list;
    x.in_count = in_count;
    x.in_cap = list_cap;
    x.ovf_list = list;
    x.ovf_count = list_count;
    x.ovf_cap = list_cap;
    x.self_any = self_any;
    x.self_end = self_end;
    x.work_counter = work_counter;
    if ((!self_any && !self_end) || (!in_list && row1 <= row0)) return hipSuccess;
    if (hipError_t e = hipMemsetAsync(work_counter, 0, sizeof(unsigned), stream); e != hipSuccess) return e;
    const int blocks = in_list ? 512 : (row1 - row0 + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(k_pairs_wave, dim3(blocks < 512 ? blocks : 512), dim3(kThreadsW), 0, stream, x);
    return hipGetLastError();
}

}  // namespace msspe
