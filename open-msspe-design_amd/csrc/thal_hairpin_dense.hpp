// thal_hairpin_dense.hpp -- device code of the generic hairpin (monomer) kernel: one lane per
// oligo, dense upper-triangular DP in a global workspace laid out [cell][lane].
//
// Replaces primer3_core's PRIMER_LEFT_0_HAIRPIN_TH for od-msspe's check_primers call
// (/root/reference/od-msspe/src/primer.rs:104-106,151-160; filter at src/main.rs:501-502).
// Algorithm: Primer3 2.6.1 thal.c type-4 path (initMatrix2, fillMatrix2, maxTM2, CBI,
// calc_bulge_internal2, calc_hairpin, calc_terminal_bp, END5_1..4, tracebacku, drawHairpin),
// restated from SURVEY.md Appendix C.5.  No reference vector pins a positive hairpin Tm
// ("restated, unpinned"); parity is against the CPU oracle.
#pragma once

#include "thal_dense.hpp"

namespace msspe {

constexpr int kMinHairpinLoop = 3;

struct HairpinCtx {
    const NNTables *tb;
    ThalConsts c;
    Seq s;
    double *S, *H;     // (len+1) x (len+1) cells, row 0 holds the 5'-prefix arrays (send5/hend5)
    size_t stride;

    __device__ __forceinline__ size_t ix(int i, int j) const
    {
        return (size_t)(i * (s.len + 1) + j) * stride;
    }
    __device__ __forceinline__ double ms(int i, int j) const { return S[ix(i, j)]; }
    __device__ __forceinline__ double mh(int i, int j) const { return H[ix(i, j)]; }
    __device__ __forceinline__ void mset(int i, int j, double s_, double h_)
    {
        S[ix(i, j)] = s_;
        H[ix(i, j)] = h_;
    }
    __device__ __forceinline__ double s5(int k) const { return S[ix(0, k)]; }
    __device__ __forceinline__ double h5(int k) const { return H[ix(0, k)]; }
    __device__ __forceinline__ void set5(int k, double s_, double h_)
    {
        S[ix(0, k)] = s_;
        H[ix(0, k)] = h_;
    }
    __device__ __forceinline__ int b(int p) const { return s.at(p); }
    __device__ __forceinline__ static bool pairs(int x, int y) { return x < 4 && y < 4 && x + y == 3; }
    __device__ __forceinline__ static double atS(int x, int y)
    {
        return (x + y == 3 && (x == 0 || x == 3)) ? 6.9 : 0.0;
    }
    __device__ __forceinline__ static double atH(int x, int y)
    {
        return (x + y == 3 && (x == 0 || x == 3)) ? 2200.0 : 0.0;
    }
    __device__ __forceinline__ double tau(double Hh, double Ss) const
    {
        return (Hh + c.init_H) / (Ss + c.init_S + c.RC);
    }

    // stack of pair (i,j) on the inner pair (i+1,j-1)  (thal.c Ss/Hs with k == 2)
    __device__ double stS(int i, int j) const
    {
        if (i >= j || i == s.len || j == s.len + 1) return -1.0;
        return tb->stackS[b(i)][b(i + 1)][b(j)][b(j - 1)];
    }
    __device__ double stH(int i, int j) const
    {
        if (i >= j || i == s.len || j == s.len + 1) return INFINITY;
        const double h = tb->stackH[b(i)][b(i + 1)][b(j)][b(j - 1)];
        return isfinite(h) ? h : INFINITY;
    }

    // thal.c RSH() on the single strand (outer bases i+1 and j+1)
    __device__ void right_end(int i, int j, double &outS, double &outH) const
    {
        const int a = b(i), bb = b(j);
        if (!pairs(a, bb)) {
            outS = -1.0;
            outH = INFINITY;
            return;
        }
        const int oa = b(i + 1), ob = b(j + 1);
        const double aS = atS(a, bb), aH = atH(a, bb);
        double S1 = aS + tb->tstack2S[a][oa][bb][ob];
        double H1 = aH + tb->tstack2H[a][oa][bb][ob];
        double G1 = H1 - kT37 * S1, T1 = -INFINITY;
        if (!isfinite(H1) || G1 > 0) {
            H1 = INFINITY;
            S1 = -1.0;
            G1 = 1.0;
        }
        const double d3S = tb->d3S[a][oa][bb], d3H = tb->d3H[a][oa][bb];
        const double d5S = tb->d5S[a][bb][ob], d5H = tb->d5H[a][bb][ob];
        bool haveD = false;
        double S2 = -1.0, H2 = INFINITY;
        if (!pairs(oa, ob)) {
            if (isfinite(d3H) && isfinite(d5H)) {
                S2 = aS + d3S + d5S;
                H2 = aH + d3H + d5H;
                haveD = true;
            } else if (isfinite(d3H)) {
                S2 = aS + d3S;
                H2 = aH + d3H;
                haveD = true;
            } else if (isfinite(d5H)) {
                S2 = aS + d5S;
                H2 = aH + d5H;
                haveD = true;
            }
        }
        if (haveD) {
            double G2 = H2 - kT37 * S2;
            if (!isfinite(H2) || G2 > 0) {
                H2 = INFINITY;
                S2 = -1.0;
                G2 = 1.0;
            }
            const double T2 = tau(H2, S2);
            if (isfinite(H1) && G1 < 0) {
                T1 = tau(H1, S1);
                if (T1 < T2 && G2 < 0) {
                    S1 = S2;
                    H1 = H2;
                    T1 = T2;
                }
            } else if (G2 < 0) {
                S1 = S2;
                H1 = H2;
                T1 = T2;
            }
        }
        const double Tb = tau(aH, aS);
        if (isfinite(H1) && !(T1 < Tb)) {
            outS = S1;
            outH = H1;
        } else {
            outS = aS;
            outH = aH;
        }
    }

    // thal.c calc_bulge_internal2(): tb_mode 0 fill, 1 traceback (loop energy alone), 2 scan
    __device__ void loop2(int i, int j, int ii, int jj, double &oS, double &oH, int tb_mode) const
    {
        const int l1 = ii - i - 1, l2 = j - jj - 1;
        if (l1 + l2 > c.max_loop) {
            oS = -1.0;
            oH = INFINITY;
            return;
        }
        const int idx = l1 + l2 - 1;
        double Sx, Hx;
        bool strict_1x1 = false;
        if ((l1 == 0 && l2 > 0) || (l2 == 0 && l1 > 0)) {
            if (l1 + l2 == 1) {
                Hx = tb->bulgeH[idx] + tb->stackH[b(i)][b(ii)][b(j)][b(jj)];
                Sx = tb->bulgeS[idx] + tb->stackS[b(i)][b(ii)][b(j)][b(jj)];
                if (tb_mode != 1) {
                    Hx += mh(ii, jj);
                    Sx += ms(ii, jj);
                }
            } else {
                Hx = tb->bulgeH[idx] + atH(b(i), b(j)) + atH(b(ii), b(jj));
                if (tb_mode != 1) Hx += mh(ii, jj);
                Sx = tb->bulgeS[idx] + atS(b(i), b(j)) + atS(b(ii), b(jj));
                if (tb_mode != 1) Sx += ms(ii, jj);
            }
        } else if (l1 == 1 && l2 == 1) {
            Sx = tb->mmS[b(i)][b(i + 1)][b(j)][b(j - 1)] + tb->mmS[b(jj)][b(jj + 1)][b(ii)][b(ii - 1)];
            if (tb_mode != 1) Sx += ms(ii, jj);
            Hx = tb->mmH[b(i)][b(i + 1)][b(j)][b(j - 1)] + tb->mmH[b(jj)][b(jj + 1)][b(ii)][b(ii - 1)];
            if (tb_mode != 1) Hx += mh(ii, jj);
            strict_1x1 = true;
        } else {
            const int asym = l1 > l2 ? l1 - l2 : l2 - l1;
            Hx = tb->interiorH[idx] + tb->tstackH[b(i)][b(i + 1)][b(j)][b(j - 1)] +
                 tb->tstackH[b(jj)][b(jj + 1)][b(ii)][b(ii - 1)] + (0.0 * asym);
            if (tb_mode != 1) Hx += mh(ii, jj);
            Sx = tb->interiorS[idx] + tb->tstackS[b(i)][b(i + 1)][b(j)][b(j - 1)] +
                 tb->tstackS[b(jj)][b(jj + 1)][b(ii)][b(ii - 1)] + (kILAS * asym);
            if (tb_mode != 1) Sx += ms(ii, jj);
        }
        if (!isfinite(Hx)) {
            Hx = INFINITY;
            Sx = -1.0;
        }
        const double T1 = (Hx + c.init_H) / ((Sx + c.init_S) + c.RC);
        const double T2 = (mh(i, j) + c.init_H) / ((ms(i, j)) + c.init_S + c.RC);
        bool take;
        if (strict_1x1) {
            take = (!((T1 - T2) < 0.000001) || tb_mode) && ((T1 - T2 >= 0.000001) || tb_mode == 1);
        } else {
            take = (T1 > T2) || ((tb_mode && T1 >= T2) || tb_mode == 1);
        }
        if (take) {
            oS = Sx;
            oH = Hx;
        }
    }

    // thal.c CBI()
    __device__ void inner_loops(int i, int j, double &oS, double &oH, int tb_mode)
    {
        for (int d = j - i - 3; d >= kMinHairpinLoop + 1 && d >= j - i - 2 - c.max_loop; --d)
            for (int ii = i + 1; ii < j - d && ii <= s.len; ++ii) {
                const int jj = d + ii;
                if (tb_mode == 0) {
                    oS = -1.0;
                    oH = INFINITY;
                }
                if (isfinite(mh(ii, jj)) && isfinite(mh(i, j))) {
                    loop2(i, j, ii, jj, oS, oH, tb_mode);
                    if (isfinite(oH)) {
                        if (oS < kMinEntropyCutoff) {
                            oS = kMinEntropy;
                            oH = 0.0;
                        }
                        if (tb_mode == 0) mset(i, j, oS, oH);
                    }
                }
            }
    }

    __device__ bool tloop(int keylen, int i, double &vS, double &vH) const
    {
        uint32_t key = 0;
        for (int q = 0; q < keylen; ++q) key = (key << 3) | (uint32_t)b(i + q);
        const int n = keylen == 5 ? tb->n_tri : tb->n_tet;
        const uint32_t *keys = keylen == 5 ? tb->triKey : tb->tetKey;
        for (int q = 0; q < n; ++q)
            if (keys[q] == key) {
                vS = keylen == 5 ? tb->triS[q] : tb->tetS[q];
                vH = keylen == 5 ? tb->triH[q] : tb->tetH[q];
                return true;
            }
        return false;
    }

    // thal.c calc_hairpin()
    __device__ void closure(int i, int j, double &oS, double &oH, int traceback) const
    {
        const int ls = j - i - 1;
        if (ls < kMinHairpinLoop) {
            oS = -1.0;
            oH = INFINITY;
            return;
        }
        const int li = ls <= 30 ? ls - 1 : 29;
        oH = tb->hairpinH[li];
        oS = tb->hairpinS[li];
        if (ls > 3) {
            oH += tb->tstack2H[b(i)][b(i + 1)][b(j)][b(j - 1)];
            oS += tb->tstack2S[b(i)][b(i + 1)][b(j)][b(j - 1)];
        } else if (ls == 3) {
            oH += atH(b(i), b(j));
            oS += atS(b(i), b(j));
        }
        double vS, vH;
        if (ls == 3 && tloop(5, i, vS, vH)) {
            oH += vH;
            oS += vS;
        } else if (ls == 4 && tloop(6, i, vS, vH)) {
            oH += vH;
            oS += vS;
        }
        if (!isfinite(oH)) {
            oH = INFINITY;
            oS = -1.0;
        }
        if (oH > 0 && oS > 0 && (!(mh(i, j) > 0) || !(ms(i, j) > 0))) {
            oH = INFINITY;
            oS = -1.0;
        }
        double rS, rH;
        right_end(i, j, rS, rH);
        const double G1 = oH + rH - kT37 * (oS + rS);
        const double G2 = mh(i, j) + rH - kT37 * (ms(i, j) + rS);
        if (G2 < G1 && traceback == 0) {
            oS = ms(i, j);
            oH = mh(i, j);
        }
    }

    // geometry of the four exterior-loop families (thal.c END5_1..4)
    __device__ __forceinline__ static int e5_kmax(int kind, int i)
    {
        return i - kMinHairpinLoop - (kind == 1 ? 2 : kind == 4 ? 4 : 3);
    }
    // energy of family `kind` at prefix split k, without the prefix term; pair = (pi,pj)
    __device__ void e5_terms(int kind, int i, int k, int &pi, int &pj, double &xS, double &xH,
                             double &aS_, double &aH_, double &eS_, double &eH_) const
    {
        pi = (kind == 1 || kind == 3) ? k + 1 : k + 2;
        pj = (kind <= 2) ? i : i - 1;
        aS_ = atS(b(pi), b(pj));
        aH_ = atH(b(pi), b(pj));
        eS_ = 0.0;
        eH_ = 0.0;
        if (kind == 2) {
            eS_ = tb->d5S[b(i)][b(k + 2)][b(k + 1)];
            eH_ = tb->d5H[b(i)][b(k + 2)][b(k + 1)];
        } else if (kind == 3) {
            eS_ = tb->d3S[b(i - 1)][b(i)][b(k + 1)];
            eH_ = tb->d3H[b(i - 1)][b(i)][b(k + 1)];
        } else if (kind == 4) {
            eS_ = tb->tstack2S[b(i - 1)][b(i)][b(k + 2)][b(k + 1)];
            eH_ = tb->tstack2H[b(i - 1)][b(i)][b(k + 2)][b(k + 1)];
        }
        if (kind == 1) {
            xS = aS_ + ms(pi, pj);
            xH = aH_ + mh(pi, pj);
        } else {
            xS = aS_ + eS_ + ms(pi, pj);
            xH = aH_ + eH_ + mh(pi, pj);
        }
    }
    __device__ void e5_with_prefix(int kind, int k, int pi, int pj, double aS_, double aH_,
                                   double eS_, double eH_, double &yS, double &yH) const
    {
        if (kind == 1) {
            yS = s5(k) + aS_ + ms(pi, pj);
            yH = h5(k) + aH_ + mh(pi, pj);
        } else {
            yS = s5(k) + aS_ + eS_ + ms(pi, pj);
            yH = h5(k) + aH_ + eH_ + mh(pi, pj);
        }
    }

    __device__ void end5(int i, int kind, double &outS, double &outH) const
    {
        double H_max = INFINITY, S_max = -1.0, max_tm = -INFINITY;
        const int kmax = e5_kmax(kind, i);
        const double T2 = (0 + c.init_H) / (0 + c.init_S + c.RC);
        for (int k = 0; k <= kmax; ++k) {
            double T1 = tau(h5(k), s5(k));
            int pi, pj;
            double xS, xH, aS_, aH_, eS_, eH_, Sx, Hx;
            e5_terms(kind, i, k, pi, pj, xS, xH, aS_, aH_, eS_, eH_);
            if (T1 >= T2) {
                e5_with_prefix(kind, k, pi, pj, aS_, aH_, eS_, eH_, Sx, Hx);
            } else {
                Sx = 0 + xS;
                Hx = 0 + xH;
            }
            if (!isfinite(Hx) || Hx > 0 || Sx > 0) {
                Hx = INFINITY;
                Sx = -1.0;
            }
            T1 = tau(Hx, Sx);
            if (max_tm < T1 && Sx > kMinEntropyCutoff) {
                H_max = Hx;
                S_max = Sx;
                max_tm = T1;
            }
        }
        outS = S_max;
        outH = H_max;
    }

    // thal.c calc_terminal_bp()
    __device__ void terminal_bp()
    {
        set5(0, -1.0, INFINITY);
        set5(1, -1.0, INFINITY);
        for (int i = 2; i <= s.len; ++i) set5(i, kMinEntropy, 0);
        for (int i = 2; i <= s.len; ++i) {
            double eS[5], eH[5], T[5];
            eS[0] = s5(i - 1);
            eH[0] = h5(i - 1);
            for (int k = 1; k <= 4; ++k) end5(i, k, eS[k], eH[k]);
            for (int k = 0; k < 5; ++k) T[k] = tau(eH[k], eS[k]);
            int m;
            if (T[0] > T[1] && T[0] > T[2] && T[0] > T[3] && T[0] > T[4]) m = 0;
            else if (T[1] > T[2] && T[1] > T[3] && T[1] > T[4]) m = 1;
            else if (T[2] > T[3] && T[2] > T[4]) m = 2;
            else if (T[3] > T[4]) m = 3;
            else m = 4;
            if (m != 0) {
                const double G = eH[m] - (c.temp_k * (eS[m]));
                if (!(G < 0.0)) m = 0;
            }
            set5(i, eS[m], eH[m]);
        }
    }

    // thal.c tracebacku(): counts paired POSITIONS the way drawHairpin does (it stops one short of
    // the last base).  bpmask bit (p-1) set iff base p is paired.
    __device__ uint64_t traceback()
    {
        uint64_t bpmask = 0;
        // explicit stack of (i, j, kind) records, each packed in one int
        int stack[96];
        int sp = 0, guard = 64 * 34;
        stack[sp++] = (s.len << 8) | (0 << 2) | 1;
        while (sp > 0 && guard-- > 0) {
            const int top = stack[--sp];
            int i = top >> 8, j = (top >> 2) & 63;
            const int mtrx = top & 3;
            if (mtrx == 1) {
                while (i > 0 && nearly(s5(i), s5(i - 1)) && nearly(h5(i), h5(i - 1))) --i;
                if (i == 0) continue;
                bool matched = false;
                for (int kind = 1; kind <= 4 && !matched; ++kind) {
                    double eS, eH;
                    end5(i, kind, eS, eH);
                    if (!(nearly(s5(i), eS) && nearly(h5(i), eH))) continue;
                    matched = true;
                    const int kmax = e5_kmax(kind, i);
                    for (int k = 0; k <= kmax; ++k) {
                        int pi, pj;
                        double xS, xH, aS_, aH_, eS_, eH_, yS, yH;
                        e5_terms(kind, i, k, pi, pj, xS, xH, aS_, aH_, eS_, eH_);
                        e5_with_prefix(kind, k, pi, pj, aS_, aH_, eS_, eH_, yS, yH);
                        if (nearly(s5(i), xS) && nearly(h5(i), xH)) {
                            if (sp < 94) stack[sp++] = (pi << 8) | (pj << 2) | 0;
                            break;
                        } else if (nearly(s5(i), yS) && nearly(h5(i), yH)) {
                            if (sp < 93) {
                                stack[sp++] = (pi << 8) | (pj << 2) | 0;
                                stack[sp++] = (k << 8) | (0 << 2) | 1;
                            }
                            break;
                        }
                    }
                }
            } else {
                bpmask |= (1ull << (i - 1)) | (1ull << (j - 1));
                double S1 = -1.0, H1 = INFINITY, S2 = -1.0, H2 = INFINITY;
                closure(i, j, S1, H1, 1);
                inner_loops(i, j, S2, H2, 2);
                const double cS = ms(i, j), cH = mh(i, j);
                if (nearly(cS, stS(i, j) + ms(i + 1, j - 1)) && nearly(cH, stH(i, j) + mh(i + 1, j - 1))) {
                    if (sp < 94) stack[sp++] = ((i + 1) << 8) | ((j - 1) << 2) | 0;
                } else if (nearly(cS, S1) && nearly(cH, H1)) {
                    // hairpin loop closes here
                } else if (nearly(cS, S2) && nearly(cH, H2)) {
                    bool done = false;
                    for (int d = j - i - 3;
                         d >= kMinHairpinLoop + 1 && d >= j - i - 2 - c.max_loop && !done; --d)
                        for (int ii = i + 1; ii < j - d; ++ii) {
                            const int jj = d + ii;
                            double eS = -1.0, eH = INFINITY;
                            loop2(i, j, ii, jj, eS, eH, 1);
                            if (nearly(cS, eS + ms(ii, jj)) && nearly(cH, eH + mh(ii, jj))) {
                                if (sp < 94) stack[sp++] = (ii << 8) | (jj << 2) | 0;
                                done = true;
                                break;
                            }
                        }
                }
            }
        }
        return bpmask;
    }

    // thal.c fillMatrix2() for ONE cell (maxTM2, CBI, calc_hairpin): reads cells strictly inside (i, j)
    // and the cell itself, writes the cell.
    __device__ void fill_cell(int i, int j)
    {
        if (!isfinite(mh(i, j))) return;
        {   // maxTM2()
            double S0 = ms(i, j), H0 = mh(i, j);
            const double T0 = tau(H0, S0);
            double S1 = (ms(i + 1, j - 1) + stS(i, j));
            double H1 = (mh(i + 1, j - 1) + stH(i, j));
            const double T1 = tau(H1, S1);
            if (S1 < kMinEntropyCutoff) {
                S1 = kMinEntropy;
                H1 = 0.0;
            }
            if (S0 < kMinEntropyCutoff) {
                S0 = kMinEntropy;
                H0 = 0.0;
            }
            if (T1 > T0) mset(i, j, S1, H1);
            else mset(i, j, S0, H0);
        }
        double oS = -1.0, oH = INFINITY;
        inner_loops(i, j, oS, oH, 0);
        oS = -1.0;
        oH = INFINITY;
        closure(i, j, oS, oH, 0);
        if (isfinite(oH)) {
            if (oS < kMinEntropyCutoff) {
                oS = kMinEntropy;
                oH = 0.0;
            }
            mset(i, j, oS, oH);
        }
    }

    // thal.c drawHairpin()'s totals from the filled planes (terminal pass, traceback, Tm)
    __device__ void finish(ThalOut &o, bool tm_only = false)
    {
        const int n = s.len;
        terminal_bp();
        const double mh_ = h5(n), ms_ = s5(n);
        o.none = 1;
        o.t = 0.0;
        o.dS = o.dH = o.dG = 0.0;
        o.n_pairs = 0;
        if (!isfinite(mh_)) return;
        if (tm_only) {
            // The caller only wants max(0, t) (libprimer3 oligo_hairpin()).  t = mh / (ms + x salt) - 273.15 with
            // x = N / 2 - 1, N the paired positions the traceback would count (-1 <= x <= len / 2): where the
            // denominator keeps its sign over that range, t is monotone in x, so if it is negative at both ends
            // it is negative for whatever the traceback finds -- the result is 0 without walking.
            const double d0 = ms_ + ((-1) * c.salt), d1 = ms_ + ((n / 2) * c.salt);
            if ((d0 < 0.0) == (d1 < 0.0) && d0 != 0.0 && d1 != 0.0) {
                const double t0 = (mh_ / d0) - kAbsZero, t1 = (mh_ / d1) - kAbsZero;
                if (t0 < 0.0 && t1 < 0.0) {
                    o.none = 0;
                    o.t = -1.0;   // any negative value: the caller reports 0
                    return;
                }
            }
        }
        const uint64_t bp = traceback();
        const int N = __popcll(bp & ((1ull << (n - 1)) - 1));   // drawHairpin: i = 1 .. len-1
        o.n_pairs = __popcll(bp) / 2;
        o.none = 0;
        o.t = (mh_ / (ms_ + (((N / 2) - 1) * c.salt))) - kAbsZero;
        o.dH = mh_;
        o.dS = ms_ + (((N / 2) - 1) * c.salt);
        o.dG = mh_ - (c.temp_k * (ms_ + (((N / 2) - 1) * c.salt)));
    }

    __device__ void run(ThalOut &o)
    {
        const int n = s.len;
        // thal.c initMatrix2()
        for (int i = 1; i <= n; ++i)
            for (int j = 1; j <= n; ++j) {
                if (j >= i && j - i >= kMinHairpinLoop + 1 && pairs(b(i), b(j))) mset(i, j, kMinEntropy, 0.0);
                else mset(i, j, -1.0, INFINITY);
            }
        // thal.c fillMatrix2()
        for (int j = 2; j <= n; ++j)
            for (int i = j - kMinHairpinLoop - 1; i >= 1; --i) {
                if (!isfinite(mh(i, j))) continue;
                {   // maxTM2()
                    double S0 = ms(i, j), H0 = mh(i, j);
                    const double T0 = tau(H0, S0);
                    double S1 = (ms(i + 1, j - 1) + stS(i, j));
                    double H1 = (mh(i + 1, j - 1) + stH(i, j));
                    const double T1 = tau(H1, S1);
                    if (S1 < kMinEntropyCutoff) {
                        S1 = kMinEntropy;
                        H1 = 0.0;
                    }
                    if (S0 < kMinEntropyCutoff) {
                        S0 = kMinEntropy;
                        H0 = 0.0;
                    }
                    if (T1 > T0) mset(i, j, S1, H1);
                    else mset(i, j, S0, H0);
                }
                double oS = -1.0, oH = INFINITY;
                inner_loops(i, j, oS, oH, 0);
                oS = -1.0;
                oH = INFINITY;
                closure(i, j, oS, oH, 0);
                if (isfinite(oH)) {
                    if (oS < kMinEntropyCutoff) {
                        oS = kMinEntropy;
                        oH = 0.0;
                    }
                    mset(i, j, oS, oH);
                }
            }
        terminal_bp();
        const double mh_ = h5(n), ms_ = s5(n);
        o.none = 1;
        o.t = 0.0;
        o.dS = o.dH = o.dG = 0.0;
        o.n_pairs = 0;
        if (!isfinite(mh_)) return;
        const uint64_t bp = traceback();
        const int N = __popcll(bp & ((1ull << (n - 1)) - 1));   // drawHairpin: i = 1 .. len-1
        o.n_pairs = __popcll(bp) / 2;
        o.none = 0;
        o.t = (mh_ / (ms_ + (((N / 2) - 1) * c.salt))) - kAbsZero;
        o.dH = mh_;
        o.dS = ms_ + (((N / 2) - 1) * c.salt);
        o.dG = mh_ - (c.temp_k * (ms_ + (((N / 2) - 1) * c.salt)));
    }
};

}  // namespace msspe
