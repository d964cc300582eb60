// thal_dense.hpp -- device code of the GENERIC thermodynamic-alignment kernels: one lane per
// problem, dense DP planes in a global-memory workspace laid out [cell][lane] (coalesced).
//
// These kernels are the general path (any oligo length <= 32, ANY / END1 / hairpin, both-
// self-complementary pairs, pairs whose DP overflows the register-resident table of the
// all-pairs kernel).  They favour clarity; the all-pairs kernel in thal_pairs.hip is the tuned
// path.  Every double expression is evaluated in Primer3's operation order and the file is
// compiled with -ffp-contract=off, so results are bit-identical to a non-FMA x86-64 ntthal.
//
// Algorithm: Primer3 2.6.1 thal.c (fillMatrix, maxTM, calc_bulge_internal, traceback, drawDimer;
// fillMatrix2, maxTM2, CBI, calc_bulge_internal2, calc_hairpin, calc_terminal_bp, END5_1..4,
// tracebacku, drawHairpin) as restated in SURVEY.md Appendix C.3-C.5.  Reference call sites:
// /root/reference/od-msspe/src/delta_g.rs:93-145 (ntthal) and src/primer.rs:151-160 (primer3_core).
#pragma once

#include <hip/hip_runtime.h>

#include "nn_params.hpp"

namespace msspe {

constexpr double kT37 = 310.15;
constexpr double kAbsZero = 273.15;
constexpr double kMinEntropyCutoff = -2500.0;
constexpr double kMinEntropy = -3224.0;
constexpr double kILAS = (-300 / 310.15);
constexpr double kTiny = 0.000001;

enum { kModeAny = 1, kModeEnd1 = 2, kModeHairpin = 4 };

struct ThalOut {
    double dS, dH, dG, t;
    int n_pairs;
    int none;   // 1: no structure
};

// Full per-pair record for callers that print what ntthal prints (the protocol shim): the
// thermodynamic values plus the base pairs of the traced structure.  Layout == msspe_thal_detail.
struct ThalDetail {
    double dS, dH, dG, t;
    int32_t no_structure, n_pairs;
    uint8_t ps1[32];   // ps1[i-1] = partner j in the REVERSED oligo 2 (1-based), 0 = unpaired
    uint8_t ps2[32];
};

// 2-bit packed oligo with N sentinels outside [1, len]
struct Seq {
    uint64_t bits;
    int len;
    __device__ __forceinline__ int at(int p) const   // 1-based
    {
        return (p < 1 || p > len) ? kBaseN : (int)((bits >> (2 * (p - 1))) & 3);
    }
};

__device__ __forceinline__ uint64_t reverse_packed(uint64_t x, int len)
{
    uint64_t r = 0;
    for (int p = 0; p < len; ++p) r |= ((x >> (2 * p)) & 3ull) << (2 * (len - 1 - p));
    return r;
}

__device__ __forceinline__ bool self_complementary(uint64_t x, int len)
{
    if (len & 1) return false;
    for (int p = 0; p < len / 2; ++p)
        if (((x >> (2 * p)) & 3) + ((x >> (2 * (len - 1 - p))) & 3) != 3) return false;
    return true;
}

__device__ __forceinline__ double h_of(int32_t h) { return h >= kHInf ? INFINITY : (double)h; }
__device__ __forceinline__ double at_pen_S(int a) { return (a == 0 || a == 3) ? 6.9 : 0.0; }
__device__ __forceinline__ double at_pen_H(int a) { return (a == 0 || a == 3) ? 2200.0 : 0.0; }
__device__ __forceinline__ bool nearly(double a, double b)
{
    if (!isfinite(a) || !isfinite(b)) return false;
    return fabs(a - b) < 1e-5;
}

// Dense planes of one lane inside the shared workspace: element (i,j), 1-based.
struct Planes {
    double *S, *H;
    size_t stride;   // lanes in the workspace
    int len2;
    __device__ __forceinline__ size_t ix(int i, int j) const
    {
        return (size_t)((i - 1) * len2 + (j - 1)) * stride;
    }
    __device__ __forceinline__ double s(int i, int j) const { return S[ix(i, j)]; }
    __device__ __forceinline__ double h(int i, int j) const { return H[ix(i, j)]; }
    __device__ __forceinline__ void set(int i, int j, double s_, double h_)
    {
        S[ix(i, j)] = s_;
        H[ix(i, j)] = h_;
    }
};

struct DimerCtx {
    const PairTables *pt;
    ThalConsts c;
    Seq s1, s2;   // s2 = oligo 2 reversed
    Planes m;

    __device__ __forceinline__ bool pair(int i, int j) const { return s1.at(i) + s2.at(j) == 3; }
    __device__ __forceinline__ void left(int i, int j, double &S, double &H) const
    {
        const int idx = s1.at(i) * 25 + s1.at(i - 1) * 5 + s2.at(j - 1);
        S = pt->endL_S[idx];
        H = h_of(pt->endL_H[idx]);
    }
    __device__ __forceinline__ void right(int i, int j, double &S, double &H) const
    {
        const int idx = s1.at(i) * 25 + s1.at(i + 1) * 5 + s2.at(j + 1);
        S = pt->endR_S[idx];
        H = h_of(pt->endR_H[idx]);
    }

    // thal.c calc_bulge_internal(): loop closed by predecessor (pi,pj) and cell (i,j); returns the
    // candidate including the predecessor's value, (-1, inf) when rejected.
    __device__ void loop_candidate(int pi, int pj, int i, int j, double &S, double &H) const
    {
        const int l1 = i - pi - 1, l2 = j - pj - 1, idx = l1 + l2 - 1;
        const int a_p = s1.at(pi), a_c = s1.at(i);
        if ((l1 == 0 && l2 > 0) || (l2 == 0 && l1 > 0)) {
            if (l1 + l2 == 1) {
                H = h_of(pt->loopH[1][idx]) + h_of(pt->wc_H[a_p * 4 + a_c]);
                S = pt->loopS[1][idx] + pt->wc_S[a_p * 4 + a_c];
                if (H > 0 || S > 0) {
                    H = INFINITY;
                    S = -1.0;
                }
                H += m.h(pi, pj);
                S += m.s(pi, pj);
                if (!isfinite(H)) {
                    H = INFINITY;
                    S = -1.0;
                }
                return;
            }
            H = h_of(pt->loopH[1][idx]) + at_pen_H(a_p) + at_pen_H(a_c);
            H += m.h(pi, pj);
            S = pt->loopS[1][idx] + at_pen_S(a_p) + at_pen_S(a_c);
            S += m.s(pi, pj);
        } else if (l1 == 1 && l2 == 1) {
            const int po = (a_p * 4 + s1.at(pi + 1)) * 4 + s2.at(pj + 1);
            const int ci = (s2.at(j) * 4 + s2.at(j - 1)) * 4 + s1.at(i - 1);
            S = pt->mm_S[po] + pt->mm_S[ci];
            S += m.s(pi, pj);
            H = h_of(pt->mm_H[po]) + h_of(pt->mm_H[ci]);
            H += m.h(pi, pj);
        } else {
            const int po = (a_p * 4 + s1.at(pi + 1)) * 4 + s2.at(pj + 1);
            const int ci = (s2.at(j) * 4 + s2.at(j - 1)) * 4 + s1.at(i - 1);
            const int asym = l1 > l2 ? l1 - l2 : l2 - l1;
            H = h_of(pt->loopH[0][idx]) + h_of(pt->ts_H[po]) + h_of(pt->ts_H[ci]) + (0.0 * asym);
            H += m.h(pi, pj);
            S = pt->loopS[0][idx] + pt->ts_S[po] + pt->ts_S[ci] + (kILAS * asym);
            S += m.s(pi, pj);
        }
        if (!isfinite(H)) {
            H = INFINITY;
            S = -1.0;
        }
        if (H > 0 && S > 0) {
            H = INFINITY;
            S = -1.0;
        }
    }

    __device__ void fill()
    {
        for (int i = 1; i <= s1.len; ++i)
            for (int j = 1; j <= s2.len; ++j) {
                if (!pair(i, j)) {
                    m.set(i, j, -1.0, INFINITY);
                    continue;
                }
                double S0, H0;
                left(i, j, S0, H0);
                if (i > 1 && j > 1) {
                    double rS, rH;
                    right(i, j, rS, rH);
                    // thal.c maxTM()
                    const double T0 = (H0 + c.init_H + rH) / (S0 + c.init_S + rS + c.RC);
                    double S1, H1, T1;
                    const double pH = m.h(i - 1, j - 1);
                    const int wc = s1.at(i - 1) * 4 + s1.at(i);
                    if (isfinite(pH)) {   // predecessor complementary => the WC stack exists
                        S1 = m.s(i - 1, j - 1) + pt->wc_S[wc];
                        H1 = pH + h_of(pt->wc_H[wc]);
                        T1 = (H1 + c.init_H + rH) / (S1 + c.init_S + rS + c.RC);
                    } else {
                        S1 = -1.0;
                        H1 = INFINITY;
                        T1 = (H1 + c.init_H) / (S1 + c.init_S + c.RC);
                    }
                    if (S1 < kMinEntropyCutoff) {
                        S1 = kMinEntropy;
                        H1 = 0.0;
                    }
                    if (S0 < kMinEntropyCutoff) {
                        S0 = kMinEntropy;
                        H0 = 0.0;
                    }
                    if (T1 > T0) {
                        S0 = S1;
                        H0 = H1;
                    }
                    double G2 = H0 + rH - kT37 * (S0 + rS);
                    for (int d = 3; d <= c.max_loop + 2; ++d) {
                        int ii = i - 1;
                        int jj = -ii - d + (j + i);
                        if (jj < 1) {
                            ii -= (1 - jj);
                            jj = 1;
                        }
                        for (; ii > 0 && jj < j; --ii, ++jj) {
                            if (!isfinite(m.h(ii, jj))) continue;
                            double S, H;
                            loop_candidate(ii, jj, i, j, S, H);
                            const double G1 = H + rH - kT37 * (S + rS);
                            if (G1 < G2) {
                                if (S < kMinEntropyCutoff) {
                                    S = kMinEntropy;
                                    H = 0.0;
                                }
                                S0 = S;
                                H0 = H;
                                G2 = H0 + rH - kT37 * (S0 + rS);
                            }
                        }
                    }
                }
                m.set(i, j, S0, H0);
            }
    }

    // thal.c traceback(): returns the number of base pairs on the optimal path from (i,j);
    // records them in d->ps1 / d->ps2 when a detail record is requested
    __device__ int traceback(int i, int j, ThalDetail *det = nullptr) const
    {
        int pairs = 1;
        if (det) {
            det->ps1[i - 1] = (uint8_t)j;
            det->ps2[j - 1] = (uint8_t)i;
        }
        int guard = 4 * (s1.len + s2.len) + 8;
        while (guard-- > 0) {
            double lS, lH;
            left(i, j, lS, lH);
            const double cS = m.s(i, j), cH = m.h(i, j);
            if (nearly(cS, lS) && nearly(cH, lH)) break;
            bool done = false;
            if (i > 1 && j > 1 && isfinite(m.h(i - 1, j - 1))) {
                const int wc = s1.at(i - 1) * 4 + s1.at(i);
                if (nearly(cS, pt->wc_S[wc] + m.s(i - 1, j - 1)) &&
                    nearly(cH, h_of(pt->wc_H[wc]) + m.h(i - 1, j - 1))) {
                    --i;
                    --j;
                    ++pairs;
                    done = true;
                    if (det) {
                        det->ps1[i - 1] = (uint8_t)j;
                        det->ps2[j - 1] = (uint8_t)i;
                    }
                }
            }
            for (int d = 3; !done && d <= c.max_loop + 2; ++d) {
                int ii = i - 1;
                int jj = -ii - d + (j + i);
                if (jj < 1) {
                    ii -= (1 - jj);
                    jj = 1;
                }
                for (; !done && ii > 0 && jj < j; --ii, ++jj) {
                    if (!isfinite(m.h(ii, jj))) continue;   // candidate would be +inf: never "nearly"
                    double S, H;
                    loop_candidate(ii, jj, i, j, S, H);
                    if (nearly(cS, S) && nearly(cH, H)) {
                        i = ii;
                        j = jj;
                        ++pairs;
                        done = true;
                        if (det) {
                            det->ps1[i - 1] = (uint8_t)j;
                            det->ps2[j - 1] = (uint8_t)i;
                        }
                    }
                }
            }
            if (!done) break;
        }
        return pairs;
    }

    __device__ void run(int mode, ThalOut &o, ThalDetail *d = nullptr)
    {
        fill();
        int bi = 0, bj = 0;
        double bestG = INFINITY;
        const int i_lo = (mode == kModeAny) ? 1 : s1.len;
        for (int i = i_lo; i <= s1.len; ++i)
            for (int j = 1; j <= s2.len; ++j) {
                if (!pair(i, j)) continue;   // +inf cell can never be a strict minimum
                double rS, rH;
                right(i, j, rS, rH);
                rS = rS + kTiny;
                rH = rH + kTiny;
                const double G1 = (m.h(i, j) + rH + c.init_H) - kT37 * (m.s(i, j) + rS + c.init_S);
                if (G1 < bestG) {
                    bestG = G1;
                    bi = i;
                    bj = j;
                }
            }
        if (mode != kModeAny) bi = s1.len;
        if (!isfinite(bestG)) bi = bj = 1;
        o.none = 1;
        o.dS = o.dH = o.dG = 0.0;
        o.t = 0.0;
        o.n_pairs = 0;
        if (!pair(bi, bj)) return;
        double rS, rH;
        right(bi, bj, rS, rH);
        const double dH = m.h(bi, bj) + rH + c.init_H;
        const double dS = m.s(bi, bj) + rS + c.init_S;
        const int P = traceback(bi, bj, d);
        const int N = P - 1;
        o.none = 0;
        o.n_pairs = P;
        o.t = (dH / (dS + (N * c.salt) + c.RC)) - kAbsZero;
        o.dG = dH - (c.temp_k * (dS + (N * c.salt)));
        o.dS = dS + (N * c.salt);
        o.dH = dH;
    }
};

}  // namespace msspe
