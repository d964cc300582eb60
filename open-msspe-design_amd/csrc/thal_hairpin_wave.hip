// thal_hairpin_wave.hip -- hairpin (monomer) melting temperature, one WAVE per oligo, DP planes in LDS.
//
// Replaces primer3_core's PRIMER_LEFT_0_HAIRPIN_TH for od-msspe's check_primers call
// (/root/reference/od-msspe/src/primer.rs:104-106,143-166; filter at src/main.rs:501-502).  The
// recurrence is the one of thal_hairpin_dense.hpp (Primer3 2.6.1 thal.c type 4, restated from SURVEY.md
// Appendix C.5): the very same device functions run here, so the doubles are those of the one-lane
// kernel and of the CPU oracle, bit for bit.
//
// CDNA4 mapping: a cell (i, j) of the upper-triangular DP reads only cells strictly inside it, so all
// cells of one span d = j - i are independent: thal.c's "j outer, i inner" fill becomes len - 4 sweeps
// over the spans, the (up to 28) cells of a sweep evaluated by the lanes of the wave, with the two planes
// ((len + 1)^2 doubles each, 17 KB at 32 bases) in LDS instead of a [cell][lane] workspace in HBM.
// The 1-D exterior-loop pass and the traceback are serial by nature and run on lane 0 out of the same
// LDS planes.  A block is one wave: no block-level barrier anywhere, nine oligos in flight per CU at
// 32 bases and sixteen at 13.
#include "kernels.hpp"
#include "thal_hairpin_dense.hpp"

namespace msspe {

namespace {

constexpr int kHpMaxLen = 32;
constexpr int kHpPlane = (kHpMaxLen + 1) * (kHpMaxLen + 1);

__global__ void __launch_bounds__(64) k_hairpin_wave(HairpinArgs a)
{
    __shared__ double S[kHpPlane], H[kHpPlane];
    const int lane = threadIdx.x;
    for (long w = blockIdx.x; w < a.n_work; w += gridDim.x) {   // block-uniform
        HairpinCtx ctx;
        ctx.tb = a.tb;
        ctx.c = a.c;
        ctx.s = Seq{a.pool[w], a.k};
        ctx.S = S;
        ctx.H = H;
        ctx.stride = 1;
        const int n = a.k;
        // thal.c initMatrix2(): every cell of the (n + 1) x (n + 1) planes, row 0 is set by the terminal pass
        for (int e = lane; e < (n + 1) * (n + 1); e += 64) {
            const int i = e / (n + 1), j = e - i * (n + 1);
            const bool open = i >= 1 && j >= i && j - i >= kMinHairpinLoop + 1 &&
                              HairpinCtx::pairs(ctx.b(i), ctx.b(j));
            S[e] = open ? kMinEntropy : -1.0;
            H[e] = open ? 0.0 : INFINITY;
        }
        __syncthreads();
        // thal.c fillMatrix2(), span by span
        for (int d = kMinHairpinLoop + 1; d <= n - 1; ++d) {
            for (int i = 1 + lane; i + d <= n; i += 64) ctx.fill_cell(i, i + d);
            __syncthreads();
        }
        if (lane == 0) {
            ThalOut o;
            ctx.finish(o, true);   // only max(0, t) leaves this kernel
            a.out_t[w] = (o.none || o.t < 0.0) ? 0.0 : o.t;   // libprimer3 oligo_hairpin(): max(0, t)
        }
        __syncthreads();   // the planes are reused by the next oligo
    }
}

}  // namespace

hipError_t launch_hairpin_wave(const HairpinArgs &a, int n_cu, hipStream_t stream)
{
    if (a.n_work <= 0) return hipSuccess;
    if (a.k > kHpMaxLen) return hipErrorInvalidValue;
    const long want = a.n_work < (long)n_cu * 16 ? a.n_work : (long)n_cu * 16;
    hipLaunchKernelGGL(k_hairpin_wave, dim3((unsigned)want), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace msspe
