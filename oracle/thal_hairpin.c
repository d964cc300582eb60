/*
 * thal_hairpin.c -- oracle restatement of Primer3 2.6.1 thal() type 4 (monomer / hairpin).
 * TEST INFRASTRUCTURE ONLY (see msspe_oracle.h).
 *
 * Reference call site: od-msspe/src/primer.rs:151-160 (primer3_core `check_primers`;
 * PRIMER_LEFT_0_HAIRPIN_TH read back at primer.rs:104-106, filtered at main.rs:501-502).
 * Inside primer3_core: libprimer3 oligo_hairpin() -> thal(oligo, oligo, hairpin args, THL_FAST).
 * The arithmetic (thal.c initMatrix2, fillMatrix2, maxTM2, CBI, calc_bulge_internal2,
 * calc_hairpin, calc_terminal_bp, END5_1..4, tracebacku, drawHairpin) is not in the reference
 * tree; it is restated from the published source as summarised in SURVEY.md Appendix C.5.
 *
 * PARITY UNPINNED: the reference pins only HAIRPIN_TH = 0.00 for one oligo
 * (od-msspe/src/primer.rs:244-250).  Positive hairpin temperatures are "restated, unpinned".
 */
#include "msspe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ILAS (-300 / 310.15)
#define ILAH 0.0
#define AT_H 2200.0
#define AT_S 6.9
#define MIN_ENTROPY_CUTOFF (-2500.0)
#define MIN_ENTROPY (-3224.0)
#define ABS_ZERO 273.15
#define T37 310.15
#define MIN_HRPN_LOOP 3

typedef struct {
    const orc_tables *tb;
    int len, max_loop;
    unsigned char s[ORC_MAX_OLIGO + 2];
    double S[ORC_MAX_OLIGO + 2][ORC_MAX_OLIGO + 2];
    double H[ORC_MAX_OLIGO + 2][ORC_MAX_OLIGO + 2];
    double s5[ORC_MAX_OLIGO + 2], h5[ORC_MAX_OLIGO + 2]; /* send5 / hend5 */
    double init_H, init_S, RC, salt;
    long ops;
} hp_ctx;

static int code_of(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}
static int is_pair(int a, int b) { return a + b == 3 && a < 4 && b < 4; }
static double at_S(int a, int b) { return (a + b == 3 && (a == 0 || a == 3)) ? AT_S : 0.0; }
static double at_H(int a, int b) { return (a + b == 3 && (a == 0 || a == 3)) ? AT_H : 0.0; }

static double salt_correction(double mv, double dv, double dntp)
{
    if (dv <= 0) dntp = dv;
    return 0.368 * (log((mv + 120 * (sqrt(fmax(0.0, dv - dntp)))) / 1000));
}

static int nearly(double a, double b)
{
    if (!isfinite(a) || !isfinite(b)) return 0;
    return fabs(a - b) < 1e-5;
}

/* thal.c Ss/Hs with k == 2: stack of pair (i,j) on the inner pair (i+1,j-1) */
static double stack2_S(const hp_ctx *c, int i, int j)
{
    if (i >= j) return -1.0;
    if (i == c->len || j == c->len + 1) return -1.0;
    return c->tb->stackS[c->s[i]][c->s[i + 1]][c->s[j]][c->s[j - 1]];
}
static double stack2_H(const hp_ctx *c, int i, int j)
{
    if (i >= j) return INFINITY;
    if (i == c->len || j == c->len + 1) return INFINITY;
    const double h = c->tb->stackH[c->s[i]][c->s[i + 1]][c->s[j]][c->s[j - 1]];
    return isfinite(h) ? h : INFINITY;
}

/* thal.c RSH() on the single strand: outer bases are s[i+1] and s[j+1] */
static void right_end(hp_ctx *c, int i, int j, double *outS, double *outH)
{
    const orc_tables *tb = c->tb;
    const int a = c->s[i], b = c->s[j];
    if (!is_pair(a, b)) {
        *outS = -1.0;
        *outH = INFINITY;
        return;
    }
    const int oa = c->s[i + 1], ob = c->s[j + 1];
    const double aS = at_S(a, b), aH = at_H(a, b);
    double S1 = aS + tb->tstack2S[a][oa][b][ob];
    double H1 = aH + tb->tstack2H[a][oa][b][ob];
    double G1 = H1 - T37 * S1;
    double T1 = -INFINITY, T2, S2 = -1.0, H2 = INFINITY, G2;
    if (!isfinite(H1) || G1 > 0) {
        H1 = INFINITY;
        S1 = -1.0;
        G1 = 1.0;
    }
    const double d3S = tb->d3S[a][oa][b], d3H = tb->d3H[a][oa][b];
    const double d5S = tb->d5S[a][b][ob], d5H = tb->d5H[a][b][ob];
    const int outer_unpaired = !is_pair(oa, ob);
    int have_d = 0;
    if (outer_unpaired && isfinite(d3H) && isfinite(d5H)) {
        S2 = aS + d3S + d5S;
        H2 = aH + d3H + d5H;
        have_d = 1;
    } else if (outer_unpaired && isfinite(d3H)) {
        S2 = aS + d3S;
        H2 = aH + d3H;
        have_d = 1;
    } else if (outer_unpaired && isfinite(d5H)) {
        S2 = aS + d5S;
        H2 = aH + d5H;
        have_d = 1;
    }
    if (have_d) {
        G2 = H2 - T37 * S2;
        if (!isfinite(H2) || G2 > 0) {
            H2 = INFINITY;
            S2 = -1.0;
            G2 = 1.0;
        }
        T2 = (H2 + c->init_H) / (S2 + c->init_S + c->RC);
        if (isfinite(H1) && G1 < 0) {
            T1 = (H1 + c->init_H) / (S1 + c->init_S + c->RC);
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
    S2 = aS;
    H2 = aH;
    T2 = (H2 + c->init_H) / (S2 + c->init_S + c->RC);
    if (isfinite(H1)) {
        if (T1 < T2) {
            *outS = S2;
            *outH = H2;
        } else {
            *outS = S1;
            *outH = H1;
        }
    } else {
        *outS = S2;
        *outH = H2;
    }
}

/* thal.c maxTM2() */
static void stack_step(hp_ctx *c, int i, int j)
{
    double S0 = c->S[i][j], H0 = c->H[i][j], S1, H1, T0, T1;
    T0 = (H0 + c->init_H) / (S0 + c->init_S + c->RC);
    if (isfinite(c->H[i][j])) {
        S1 = (c->S[i + 1][j - 1] + stack2_S(c, i, j));
        H1 = (c->H[i + 1][j - 1] + stack2_H(c, i, j));
    } else {
        S1 = -1.0;
        H1 = INFINITY;
    }
    T1 = (H1 + c->init_H) / (S1 + c->init_S + c->RC);
    if (S1 < MIN_ENTROPY_CUTOFF) {
        S1 = MIN_ENTROPY;
        H1 = 0.0;
    }
    if (S0 < MIN_ENTROPY_CUTOFF) {
        S0 = MIN_ENTROPY;
        H0 = 0.0;
    }
    if (T1 > T0) {
        c->S[i][j] = S1;
        c->H[i][j] = H1;
    } else {
        c->S[i][j] = S0;
        c->H[i][j] = H0;
    }
}

/*
 * thal.c calc_bulge_internal2(): loop between outer pair (i,j) and inner pair (ii,jj).
 * tb_mode 0 = fill (inner value added, strict Tm test), 1 = traceback (loop energy alone, always
 * returned), 2 = traceback scan inside CBI (inner value added, >= test).
 */
static void loop2(hp_ctx *c, int i, int j, int ii, int jj, double *SH, int tb_mode)
{
    const orc_tables *tb = c->tb;
    const unsigned char *s = c->s;
    const int l1 = ii - i - 1, l2 = j - jj - 1;
    double S = MIN_ENTROPY, H = 0.0, T1, T2;
    if (l1 + l2 > c->max_loop) {
        SH[0] = -1.0;
        SH[1] = INFINITY;
        return;
    }
    const int idx = l1 + l2 - 1;
    if ((l1 == 0 && l2 > 0) || (l2 == 0 && l1 > 0)) {
        if (l2 == 1 || l1 == 1) {
            H = tb->bulgeH[idx] + tb->stackH[s[i]][s[ii]][s[j]][s[jj]];
            S = tb->bulgeS[idx] + tb->stackS[s[i]][s[ii]][s[j]][s[jj]];
            if (tb_mode != 1) {
                H += c->H[ii][jj];
                S += c->S[ii][jj];
            }
            if (!isfinite(H)) {
                H = INFINITY;
                S = -1.0;
            }
            T1 = (H + c->init_H) / ((S + c->init_S) + c->RC);
            T2 = (c->H[i][j] + c->init_H) / ((c->S[i][j]) + c->init_S + c->RC);
            if ((T1 > T2) || ((tb_mode && T1 >= T2) || tb_mode == 1)) {
                SH[0] = S;
                SH[1] = H;
            }
        } else {
            H = tb->bulgeH[idx] + at_H(s[i], s[j]) + at_H(s[ii], s[jj]);
            if (tb_mode != 1) H += c->H[ii][jj];
            S = tb->bulgeS[idx] + at_S(s[i], s[j]) + at_S(s[ii], s[jj]);
            if (tb_mode != 1) S += c->S[ii][jj];
            if (!isfinite(H)) {
                H = INFINITY;
                S = -1.0;
            }
            T1 = (H + c->init_H) / ((S + c->init_S) + c->RC);
            T2 = (c->H[i][j] + c->init_H) / (c->S[i][j] + c->init_S + c->RC);
            if ((T1 > T2) || ((tb_mode && T1 >= T2) || (tb_mode == 1))) {
                SH[0] = S;
                SH[1] = H;
            }
        }
    } else if (l1 == 1 && l2 == 1) {
        S = tb->mmS[s[i]][s[i + 1]][s[j]][s[j - 1]] + tb->mmS[s[jj]][s[jj + 1]][s[ii]][s[ii - 1]];
        if (tb_mode != 1) S += c->S[ii][jj];
        H = tb->mmH[s[i]][s[i + 1]][s[j]][s[j - 1]] + tb->mmH[s[jj]][s[jj + 1]][s[ii]][s[ii - 1]];
        if (tb_mode != 1) H += c->H[ii][jj];
        if (!isfinite(H)) {
            H = INFINITY;
            S = -1.0;
        }
        T1 = (H + c->init_H) / ((S + c->init_S) + c->RC);
        T2 = (c->H[i][j] + c->init_H) / (c->S[i][j] + c->init_S + c->RC);
        /* thal.c: DBL_EQ(T1,T2) == 2, i.e. T1 - T2 >= 1e-6 */
        if (!((T1 - T2) < 0.000001) || tb_mode) {
            if ((T1 - T2 >= 0.000001) || tb_mode == 1) {
                SH[0] = S;
                SH[1] = H;
            }
        }
    } else {
        H = tb->interiorH[idx] + tb->tstackH[s[i]][s[i + 1]][s[j]][s[j - 1]] +
            tb->tstackH[s[jj]][s[jj + 1]][s[ii]][s[ii - 1]] + (ILAH * abs(l1 - l2));
        if (tb_mode != 1) H += c->H[ii][jj];
        S = tb->interiorS[idx] + tb->tstackS[s[i]][s[i + 1]][s[j]][s[j - 1]] +
            tb->tstackS[s[jj]][s[jj + 1]][s[ii]][s[ii - 1]] + (ILAS * abs(l1 - l2));
        if (tb_mode != 1) S += c->S[ii][jj];
        if (!isfinite(H)) {
            H = INFINITY;
            S = -1.0;
        }
        T1 = (H + c->init_H) / ((S + c->init_S) + c->RC);
        T2 = (c->H[i][j] + c->init_H) / ((c->S[i][j]) + c->init_S + c->RC);
        if ((T1 > T2) || ((tb_mode && T1 >= T2) || (tb_mode == 1))) {
            SH[0] = S;
            SH[1] = H;
        }
    }
}

/* thal.c CBI(): all bulge / interior loops from (i,j) to inner pairs */
static void inner_loops(hp_ctx *c, int i, int j, double *SH, int tb_mode)
{
    for (int d = j - i - 3; d >= MIN_HRPN_LOOP + 1 && d >= j - i - 2 - c->max_loop; --d)
        for (int ii = i + 1; ii < j - d && ii <= c->len; ++ii) {
            const int jj = d + ii;
            if (tb_mode == 0) {
                SH[0] = -1.0;
                SH[1] = INFINITY;
            }
            if (isfinite(c->H[ii][jj]) && isfinite(c->H[i][j])) {
                loop2(c, i, j, ii, jj, SH, tb_mode);
                if (isfinite(SH[1])) {
                    if (SH[0] < MIN_ENTROPY_CUTOFF) {
                        SH[0] = MIN_ENTROPY;
                        SH[1] = 0.0;
                    }
                    if (tb_mode == 0) {
                        c->H[i][j] = SH[1];
                        c->S[i][j] = SH[0];
                    }
                }
            }
        }
}

static int tloop_find(const orc_tloop *tab, int n, const unsigned char *key, int keylen,
                      double *value)
{
    for (int k = 0; k < n; k++)
        if (!memcmp(tab[k].key, key, (size_t)keylen)) {
            *value = tab[k].value;
            return 1;
        }
    return 0;
}

/* thal.c calc_hairpin(): closing the hairpin loop at pair (i,j) */
static void hairpin_closure(hp_ctx *c, int i, int j, double *SH, int traceback)
{
    const orc_tables *tb = c->tb;
    const unsigned char *s = c->s;
    const int ls = j - i - 1;
    if (ls < MIN_HRPN_LOOP) {
        SH[0] = -1.0;
        SH[1] = INFINITY;
        return;
    }
    if (i <= c->len && c->len < j) {
        SH[0] = -1.0;
        SH[1] = INFINITY;
        return;
    }
    if (ls <= 30) {
        SH[1] = tb->hairpinH[ls - 1];
        SH[0] = tb->hairpinS[ls - 1];
    } else {
        SH[1] = tb->hairpinH[29];
        SH[0] = tb->hairpinS[29];
    }
    if (ls > 3) {
        SH[1] += tb->tstack2H[s[i]][s[i + 1]][s[j]][s[j - 1]];
        SH[0] += tb->tstack2S[s[i]][s[i + 1]][s[j]][s[j - 1]];
    } else if (ls == 3) {
        SH[1] += at_H(s[i], s[j]);
        SH[0] += at_S(s[i], s[j]);
    }
    double v;
    if (ls == 3) {
        if (tloop_find(tb->triH, tb->n_tri_h, s + i, 5, &v)) SH[1] += v;
        if (tloop_find(tb->triS, tb->n_tri_s, s + i, 5, &v)) SH[0] += v;
    } else if (ls == 4) {
        if (tloop_find(tb->tetH, tb->n_tet_h, s + i, 6, &v)) SH[1] += v;
        if (tloop_find(tb->tetS, tb->n_tet_s, s + i, 6, &v)) SH[0] += v;
    }
    if (!isfinite(SH[1])) {
        SH[1] = INFINITY;
        SH[0] = -1.0;
    }
    if (SH[1] > 0 && SH[0] > 0 && (!(c->H[i][j] > 0) || !(c->S[i][j] > 0))) {
        SH[1] = INFINITY;
        SH[0] = -1.0;
    }
    double rS, rH;
    right_end(c, i, j, &rS, &rH);
    const double G1 = SH[1] + rH - T37 * (SH[0] + rS);
    const double G2 = c->H[i][j] + rH - T37 * (c->S[i][j] + rS);
    if (G2 < G1 && traceback == 0) {
        SH[0] = c->S[i][j];
        SH[1] = c->H[i][j];
    }
}

/* thal.c END5_1..4: best 5'-prefix structure ending in a helix closed at/near position i.
 * kind 1: pair (k+1,i); 2: pair (k+2,i) + 5' dangle; 3: pair (k+1,i-1) + 3' dangle;
 * 4: pair (k+2,i-1) + terminal mismatch. */
static void end5(const hp_ctx *c, int i, int kind, double *outS, double *outH)
{
    const orc_tables *tb = c->tb;
    const unsigned char *s = c->s;
    double H_max = INFINITY, S_max = -1.0, max_tm = -INFINITY;
    int kmax;
    switch (kind) {
    case 1: kmax = i - MIN_HRPN_LOOP - 2; break;
    case 2: kmax = i - MIN_HRPN_LOOP - 3; break;
    case 3: kmax = i - MIN_HRPN_LOOP - 3; break;
    default: kmax = i - MIN_HRPN_LOOP - 4; break;
    }
    for (int k = 0; k <= kmax; ++k) {
        double T1 = (c->h5[k] + c->init_H) / (c->s5[k] + c->init_S + c->RC);
        const double T2 = (0 + c->init_H) / (0 + c->init_S + c->RC);
        double addS, addH;
        switch (kind) {
        case 1:
            addH = at_H(s[k + 1], s[i]) + c->H[k + 1][i];
            addS = at_S(s[k + 1], s[i]) + c->S[k + 1][i];
            break;
        case 2:
            addH = at_H(s[k + 2], s[i]) + tb->d5H[s[i]][s[k + 2]][s[k + 1]] + c->H[k + 2][i];
            addS = at_S(s[k + 2], s[i]) + tb->d5S[s[i]][s[k + 2]][s[k + 1]] + c->S[k + 2][i];
            break;
        case 3:
            addH = at_H(s[k + 1], s[i - 1]) + tb->d3H[s[i - 1]][s[i]][s[k + 1]] +
                   c->H[k + 1][i - 1];
            addS = at_S(s[k + 1], s[i - 1]) + tb->d3S[s[i - 1]][s[i]][s[k + 1]] +
                   c->S[k + 1][i - 1];
            break;
        default:
            addH = at_H(s[k + 2], s[i - 1]) + tb->tstack2H[s[i - 1]][s[i]][s[k + 2]][s[k + 1]] +
                   c->H[k + 2][i - 1];
            addS = at_S(s[k + 2], s[i - 1]) + tb->tstack2S[s[i - 1]][s[i]][s[k + 2]][s[k + 1]] +
                   c->S[k + 2][i - 1];
            break;
        }
        double H, S;
        /* thal.c writes "HEND5(k) + atPenalty + [dangle] + DPT" left to right */
        if (T1 >= T2) {
            switch (kind) {
            case 1:
                H = c->h5[k] + at_H(s[k + 1], s[i]) + c->H[k + 1][i];
                S = c->s5[k] + at_S(s[k + 1], s[i]) + c->S[k + 1][i];
                break;
            case 2:
                H = c->h5[k] + at_H(s[k + 2], s[i]) + tb->d5H[s[i]][s[k + 2]][s[k + 1]] +
                    c->H[k + 2][i];
                S = c->s5[k] + at_S(s[k + 2], s[i]) + tb->d5S[s[i]][s[k + 2]][s[k + 1]] +
                    c->S[k + 2][i];
                break;
            case 3:
                H = c->h5[k] + at_H(s[k + 1], s[i - 1]) + tb->d3H[s[i - 1]][s[i]][s[k + 1]] +
                    c->H[k + 1][i - 1];
                S = c->s5[k] + at_S(s[k + 1], s[i - 1]) + tb->d3S[s[i - 1]][s[i]][s[k + 1]] +
                    c->S[k + 1][i - 1];
                break;
            default:
                H = c->h5[k] + at_H(s[k + 2], s[i - 1]) +
                    tb->tstack2H[s[i - 1]][s[i]][s[k + 2]][s[k + 1]] + c->H[k + 2][i - 1];
                S = c->s5[k] + at_S(s[k + 2], s[i - 1]) +
                    tb->tstack2S[s[i - 1]][s[i]][s[k + 2]][s[k + 1]] + c->S[k + 2][i - 1];
                break;
            }
        } else {
            H = 0 + addH;
            S = 0 + addS;
        }
        if (!isfinite(H) || H > 0 || S > 0) {
            H = INFINITY;
            S = -1.0;
        }
        T1 = (H + c->init_H) / (S + c->init_S + c->RC);
        if (max_tm < T1) {
            if (S > MIN_ENTROPY_CUTOFF) {
                H_max = H;
                S_max = S;
                max_tm = T1;
            }
        }
    }
    *outS = S_max;
    *outH = H_max;
}

static int max5(double a, double b, double c, double d, double e)
{
    if (a > b && a > c && a > d && a > e) return 1;
    else if (b > c && b > d && b > e) return 2;
    else if (c > d && c > e) return 3;
    else if (d > e) return 4;
    else return 5;
}

/* thal.c calc_terminal_bp() */
static void terminal_bp(hp_ctx *c, double temp)
{
    c->s5[0] = c->s5[1] = -1.0;
    c->h5[0] = c->h5[1] = INFINITY;
    for (int i = 2; i <= c->len; i++) {
        c->s5[i] = MIN_ENTROPY;
        c->h5[i] = 0;
    }
    for (int i = 2; i <= c->len; ++i) {
        double eS[5], eH[5], T[5];
        eS[0] = c->s5[i - 1];
        eH[0] = c->h5[i - 1];
        for (int k = 1; k <= 4; k++) end5(c, i, k, &eS[k], &eH[k]);
        for (int k = 0; k < 5; k++) T[k] = (eH[k] + c->init_H) / (eS[k] + c->init_S + c->RC);
        const int m = max5(T[0], T[1], T[2], T[3], T[4]);
        if (m == 1) {
            c->s5[i] = c->s5[i - 1];
            c->h5[i] = c->h5[i - 1];
        } else {
            const double G = eH[m - 1] - (temp * (eS[m - 1]));
            if (G < 0.0) {
                c->s5[i] = eS[m - 1];
                c->h5[i] = eH[m - 1];
            } else {
                c->s5[i] = c->s5[i - 1];
                c->h5[i] = c->h5[i - 1];
            }
        }
    }
}

typedef struct {
    int i, j, mtrx;
} tracer;

/* thal.c tracebacku(): fills bp[] with the partners of the optimal monomer structure */
static void traceback_u(hp_ctx *c, int *bp)
{
    tracer stack[8 * ORC_MAX_OLIGO];
    int sp = 0, guard = 64 * ORC_MAX_OLIGO;
    const unsigned char *s = c->s;
    const orc_tables *tb = c->tb;
    stack[sp++] = (tracer){c->len, 0, 1};
    while (sp > 0 && guard-- > 0) {
        tracer top = stack[--sp];
        int i = top.i, j = top.j;
        if (top.mtrx == 1) {
            while (i > 0 && nearly(c->s5[i], c->s5[i - 1]) && nearly(c->h5[i], c->h5[i - 1])) --i;
            if (i == 0) continue;
            double eS, eH;
            int matched = 0;
            for (int kind = 1; kind <= 4 && !matched; kind++) {
                end5(c, i, kind, &eS, &eH);
                if (!(nearly(c->s5[i], eS) && nearly(c->h5[i], eH))) continue;
                matched = 1;
                int kmax, pi_off, pj;
                switch (kind) {
                case 1: kmax = i - MIN_HRPN_LOOP - 2; pi_off = 1; pj = i; break;
                case 2: kmax = i - MIN_HRPN_LOOP - 3; pi_off = 2; pj = i; break;
                case 3: kmax = i - MIN_HRPN_LOOP - 3; pi_off = 1; pj = i - 1; break;
                default: kmax = i - MIN_HRPN_LOOP - 4; pi_off = 2; pj = i - 1; break;
                }
                for (int k = 0; k <= kmax; ++k) {
                    const int pi = k + pi_off;
                    double aS = at_S(s[pi], s[pj]), aH = at_H(s[pi], s[pj]);
                    double xS, xH; /* energy without the prefix */
                    switch (kind) {
                    case 1:
                        xS = aS + c->S[pi][pj];
                        xH = aH + c->H[pi][pj];
                        break;
                    case 2:
                        xS = aS + tb->d5S[s[i]][s[k + 2]][s[k + 1]] + c->S[pi][pj];
                        xH = aH + tb->d5H[s[i]][s[k + 2]][s[k + 1]] + c->H[pi][pj];
                        break;
                    case 3:
                        xS = aS + tb->d3S[s[i - 1]][s[i]][s[k + 1]] + c->S[pi][pj];
                        xH = aH + tb->d3H[s[i - 1]][s[i]][s[k + 1]] + c->H[pi][pj];
                        break;
                    default:
                        xS = aS + tb->tstack2S[s[i - 1]][s[i]][s[k + 2]][s[k + 1]] + c->S[pi][pj];
                        xH = aH + tb->tstack2H[s[i - 1]][s[i]][s[k + 2]][s[k + 1]] + c->H[pi][pj];
                        break;
                    }
                    double yS, yH; /* with the prefix, in thal.c's left-to-right order */
                    switch (kind) {
                    case 1:
                        yS = c->s5[k] + aS + c->S[pi][pj];
                        yH = c->h5[k] + aH + c->H[pi][pj];
                        break;
                    case 2:
                        yS = c->s5[k] + aS + tb->d5S[s[i]][s[k + 2]][s[k + 1]] + c->S[pi][pj];
                        yH = c->h5[k] + aH + tb->d5H[s[i]][s[k + 2]][s[k + 1]] + c->H[pi][pj];
                        break;
                    case 3:
                        yS = c->s5[k] + aS + tb->d3S[s[i - 1]][s[i]][s[k + 1]] + c->S[pi][pj];
                        yH = c->h5[k] + aH + tb->d3H[s[i - 1]][s[i]][s[k + 1]] + c->H[pi][pj];
                        break;
                    default:
                        yS = c->s5[k] + aS + tb->tstack2S[s[i - 1]][s[i]][s[k + 2]][s[k + 1]] +
                             c->S[pi][pj];
                        yH = c->h5[k] + aH + tb->tstack2H[s[i - 1]][s[i]][s[k + 2]][s[k + 1]] +
                             c->H[pi][pj];
                        break;
                    }
                    if (nearly(c->s5[i], xS) && nearly(c->h5[i], xH)) {
                        stack[sp++] = (tracer){pi, pj, 0};
                        break;
                    } else if (nearly(c->s5[i], yS) && nearly(c->h5[i], yH)) {
                        stack[sp++] = (tracer){pi, pj, 0};
                        stack[sp++] = (tracer){k, 0, 1};
                        break;
                    }
                }
            }
        } else {
            bp[i - 1] = j;
            bp[j - 1] = i;
            double SH1[2] = {-1.0, INFINITY}, SH2[2] = {-1.0, INFINITY};
            hairpin_closure(c, i, j, SH1, 1);
            inner_loops(c, i, j, SH2, 2);
            if (nearly(c->S[i][j], stack2_S(c, i, j) + c->S[i + 1][j - 1]) &&
                nearly(c->H[i][j], stack2_H(c, i, j) + c->H[i + 1][j - 1])) {
                stack[sp++] = (tracer){i + 1, j - 1, 0};
            } else if (nearly(c->S[i][j], SH1[0]) && nearly(c->H[i][j], SH1[1])) {
                /* hairpin loop closes here */
            } else if (nearly(c->S[i][j], SH2[0]) && nearly(c->H[i][j], SH2[1])) {
                int done = 0;
                for (int d = j - i - 3;
                     d >= MIN_HRPN_LOOP + 1 && d >= j - i - 2 - c->max_loop && !done; --d)
                    for (int ii = i + 1; ii < j - d; ++ii) {
                        const int jj = d + ii;
                        double E[2] = {-1.0, INFINITY};
                        loop2(c, i, j, ii, jj, E, 1);
                        if (nearly(c->S[i][j], E[0] + c->S[ii][jj]) &&
                            nearly(c->H[i][j], E[1] + c->H[ii][jj])) {
                            stack[sp++] = (tracer){ii, jj, 0};
                            ++done;
                            break;
                        }
                    }
            }
        }
        if (sp > (int)(sizeof stack / sizeof stack[0]) - 4) break;
    }
}

int orc_thal_hairpin(const orc_tables *t, const char *oligo, const orc_thal_args *a,
                     orc_thal_result *r);

int orc_thal_hairpin(const orc_tables *t, const char *oligo, const orc_thal_args *a,
                     orc_thal_result *r)
{
    const int n = (int)strlen(oligo);
    if (n < 1 || n > ORC_MAX_OLIGO) return -1;
    hp_ctx *c = (hp_ctx *)calloc(1, sizeof *c);
    if (!c) return -1;
    memset(r, 0, sizeof *r);
    c->tb = t;
    c->len = n;
    c->max_loop = a->max_loop;
    for (int i = 1; i <= n; i++) c->s[i] = (unsigned char)code_of(oligo[i - 1]);
    c->s[0] = c->s[n + 1] = 4;
    c->init_H = 0.0;
    c->init_S = -0.00000000001;
    c->RC = 0;
    c->salt = salt_correction(a->mv, a->dv, a->dntp);

    /* thal.c initMatrix2() */
    for (int i = 0; i <= n + 1; i++)
        for (int j = 0; j <= n + 1; j++) {
            c->H[i][j] = INFINITY;
            c->S[i][j] = -1.0;
        }
    for (int i = 1; i <= n; ++i)
        for (int j = i; j <= n; ++j)
            if (j - i < MIN_HRPN_LOOP + 1 || !is_pair(c->s[i], c->s[j])) {
                c->H[i][j] = INFINITY;
                c->S[i][j] = -1.0;
            } else {
                c->H[i][j] = 0.0;
                c->S[i][j] = MIN_ENTROPY;
            }
    /* thal.c fillMatrix2() */
    for (int j = 2; j <= n; ++j)
        for (int i = j - MIN_HRPN_LOOP - 1; i >= 1; --i) {
            if (!isfinite(c->H[i][j])) continue;
            double SH[2] = {-1.0, INFINITY};
            stack_step(c, i, j);
            inner_loops(c, i, j, SH, 0);
            SH[0] = -1.0;
            SH[1] = INFINITY;
            hairpin_closure(c, i, j, SH, 0);
            if (isfinite(SH[1])) {
                if (SH[0] < MIN_ENTROPY_CUTOFF) {
                    SH[0] = MIN_ENTROPY;
                    SH[1] = 0.0;
                }
                c->S[i][j] = SH[0];
                c->H[i][j] = SH[1];
            }
        }
    terminal_bp(c, a->temp_k);
    const double mh = c->h5[n], ms = c->s5[n];
    if (isfinite(mh)) {
        traceback_u(c, r->bp);
        /* thal.c drawHairpin(): note the loop stops one short of the last base */
        int N = 0;
        for (int i = 1; i < n; ++i)
            if (r->bp[i - 1] > 0) N++;
        int np = 0;
        for (int i = 0; i < n; ++i)
            if (r->bp[i] > 0) np++;
        r->n_pairs = np / 2;
        const double tm = (mh / (ms + (((N / 2) - 1) * c->salt))) - ABS_ZERO;
        r->t = tm;
        r->dH = mh;
        r->dS_raw = ms;
        r->dS = ms + (((N / 2) - 1) * c->salt);
        r->dG = mh - (a->temp_k * (ms + (((N / 2) - 1) * c->salt)));
        r->no_structure = 0;
    } else {
        r->no_structure = 1;
        r->t = 0.0;
    }
    free(c);
    return 0;
}
