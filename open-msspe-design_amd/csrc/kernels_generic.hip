// kernels_generic.hip -- generic (one lane per problem, dense DP in a global workspace) kernels:
// thal ANY / END1 for arbitrary ordered pairs, thal HAIRPIN, and oligotm.
// Built for gfx950 only, with -ffp-contract=off (see thal_dense.hpp).
#include "kernels.hpp"
#include "thal_dense.hpp"
#include "thal_hairpin_dense.hpp"

namespace msspe {

namespace {

__device__ __forceinline__ void sink_pair(const PairSinks &s, const ThalConsts &c, int row, int col,
                                          const ThalOut &o)
{
    const bool conflict = !o.none && o.dG <= c.g_cut;
    const size_t r = (size_t)(row - s.row0), q = (size_t)(col - s.col0);
    if (s.dg) s.dg[r * (size_t)s.ncols + q] = o.none ? INFINITY : o.dG;
    if (s.tm) s.tm[r * (size_t)s.ncols + q] = o.none ? 0.0 : o.t;
    if (conflict) {
        if (s.row_conflicts) atomicAdd(&s.row_conflicts[row], 1u);
        if (s.bitmap)
            atomicOr((unsigned long long *)&s.bitmap[r * (size_t)s.words + (q >> 6)],
                     1ull << (q & 63));
        sink_edge(s, row, col, o.dG);
    }
}

__global__ void __launch_bounds__(256) k_dimer_generic(GenericDimerArgs a)
{
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    long n_work = a.n_work;
    if (a.list_count) {
        const long cnt = (long)*a.list_count;
        n_work = cnt < n_work ? cnt : n_work;
    }
    for (long w = (long)lane; w < n_work; w += (long)a.ws_lanes) {
        int row, col;
        if (a.list) {
            row = (int)(a.list[w].x & 0x7fffffffu);   // bit 31: mark of the integer stage
            col = (int)a.list[w].y;
        } else if (a.self_mode) {
            row = col = (int)w;
        } else {
            row = a.sinks.row0 + (int)(w / a.sinks.ncols);
            col = a.sinks.col0 + (int)(w % a.sinks.ncols);
        }
        const uint64_t pa = a.pool[row], pb = a.pool[col];
        const int sym = (self_complementary(pa, a.k) && self_complementary(pb, a.k)) ? 1 : 0;
        DimerCtx ctx;
        ctx.pt = a.pt + sym;
        ctx.c = a.c[sym];
        ctx.s1 = Seq{pa, a.k};
        ctx.s2 = Seq{reverse_packed(pb, a.k), a.k};
        ctx.m = Planes{a.wsS + lane, a.wsH + lane, a.ws_lanes, a.k};
        ThalOut o;
        ThalDetail *det = a.detail ? (ThalDetail *)a.detail + w : nullptr;
        if (det) {
            for (int q = 0; q < 32; ++q) det->ps1[q] = det->ps2[q] = 0;
        }
        ctx.run(a.mode, o, det);
        if (det) {
            det->dS = o.dS;
            det->dH = o.dH;
            det->dG = o.dG;
            det->t = o.t;
            det->no_structure = o.none;
            det->n_pairs = o.n_pairs;
        } else if (a.self_mode) {
            a.self_t[row] = (o.none || o.t < 0.0) ? 0.0 : o.t;   // libprimer3 align_thermod()
        } else {
            sink_pair(a.sinks, ctx.c, row, col, o);
        }
    }
}

__global__ void __launch_bounds__(256) k_hairpin_generic(HairpinArgs a)
{
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (long w = (long)lane; w < a.n_work; w += (long)a.ws_lanes) {
        HairpinCtx ctx;
        ctx.tb = a.tb;
        ctx.c = a.c;
        ctx.s = Seq{a.pool[w], a.k};
        ctx.S = a.wsS + lane;
        ctx.H = a.wsH + lane;
        ctx.stride = a.ws_lanes;
        ThalOut o;
        ctx.run(o);
        a.out_t[w] = (o.none || o.t < 0.0) ? 0.0 : o.t;
    }
}

// Primer3 2.6.1 oligotm.c oligotm(), tm_method = santalucia_auto, salt_corrections = santalucia
// (SURVEY.md Appendix C.2).  NN sums are exact integers; log() terms arrive precomputed.
__constant__ int kNN_S[16] = {222, 224, 210, 204, 227, 199, 272, 210,
                              222, 244, 199, 224, 213, 222, 227, 222};
__constant__ int kNN_H[16] = {79, 84, 78, 72, 85, 80, 106, 78, 82, 98, 80, 84, 72, 82, 85, 79};

__global__ void __launch_bounds__(256) k_oligo_tm(const uint64_t *pool, int n, int k,
                                                  double salt_term, double log_c4, double log_c1,
                                                  double *tm, double *gc)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const uint64_t x = pool[w];
    const bool sym = self_complementary(x, k);
    int dh = 0, ds = 0, ngc = 0;
    if (sym) ds += 14;
    const int first = (int)(x & 3), last = (int)((x >> (2 * (k - 1))) & 3);
    if (first == 0 || first == 3) { ds += -41; dh += -23; } else { ds += 28; dh += -1; }
    if (last == 0 || last == 3) { ds += -41; dh += -23; } else { ds += 28; dh += -1; }
    for (int p = 0; p < k; ++p) {
        const int b0 = (int)((x >> (2 * p)) & 3);
        if (b0 == 1 || b0 == 2) ++ngc;
        if (p + 1 < k) {
            const int b1 = (int)((x >> (2 * (p + 1))) & 3);
            ds += kNN_S[b0 * 4 + b1];
            dh += kNN_H[b0 * 4 + b1];
        }
    }
    const double delta_H = dh * -100.0;
    double delta_S = ds * -0.1;
    delta_S = delta_S + salt_term;   // 0.368 * (k - 1) * log(K_mM / 1000)
    if (tm) tm[w] = delta_H / (delta_S + (sym ? log_c1 : log_c4)) - 273.15;
    if (gc) gc[w] = 100.0 * ((double)ngc) / k;
}

}  // namespace

hipError_t launch_dimer_generic(const GenericDimerArgs &a, hipStream_t stream)
{
    if (a.n_work <= 0) return hipSuccess;
    const int block = 256;
    const long lanes = (long)a.ws_lanes;
    long want = a.n_work < lanes ? a.n_work : lanes;
    const int grid = (int)((want + block - 1) / block);
    hipLaunchKernelGGL(k_dimer_generic, dim3(grid), dim3(block), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_hairpin_generic(const HairpinArgs &a, hipStream_t stream)
{
    if (a.n_work <= 0) return hipSuccess;
    const int block = 256;
    const long lanes = (long)a.ws_lanes;
    long want = a.n_work < lanes ? a.n_work : lanes;
    const int grid = (int)((want + block - 1) / block);
    hipLaunchKernelGGL(k_hairpin_generic, dim3(grid), dim3(block), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_oligo_tm(const uint64_t *pool, int n, int k, double dna_conc, double mv,
                           double dv, double dntp, double *tm, double *gc, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    // oligotm.c divalent_to_monovalent() and the salt / concentration logarithms, on the host
    if (dv == 0) dntp = 0;
    if (dv < dntp) dv = dntp;
    const double K_mM = mv + 120 * (sqrt(dv - dntp));
    const double salt_term = 0.368 * (k - 1) * log(K_mM / 1000.0);
    const double log_c4 = 1.987 * log(dna_conc / 4000000000.0);
    const double log_c1 = 1.987 * log(dna_conc / 1000000000.0);
    hipLaunchKernelGGL(k_oligo_tm, dim3((n + 255) / 256), dim3(256), 0, stream, pool, n, k,
                       salt_term, log_c4, log_c1, tm, gc);
    return hipGetLastError();
}

}  // namespace msspe
