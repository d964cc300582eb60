/*
 * thal_dimer.c -- oracle restatement of Primer3 2.6.1 thal() for duplexes (ANY / END1 / END2).
 * TEST INFRASTRUCTURE ONLY (see msspe_oracle.h).
 *
 * Reference call sites: od-msspe/src/delta_g.rs:93-145 (ntthal -a ANY ... -i; one thal ANY per
 * ordered pair) and od-msspe/src/primer.rs:151-160 (primer3_core: SELF_ANY_TH = thal ANY,
 * SELF_END_TH = thal END1 of the oligo against itself).  The arithmetic lives in Primer3 2.6.1
 * src/thal.c, which is not in the reference tree; this file restates its published algorithm
 * (thal, fillMatrix, maxTM, LSH, RSH, calc_bulge_internal, traceback, drawDimer) as summarised in
 * SURVEY.md Appendix C.3/C.4, with every floating-point expression kept in Primer3's operation
 * order (compile with -ffp-contract=off).  Unlike thal.c there are no globals: all state lives in
 * a per-call context, so the oracle is re-entrant and OpenMP-safe.
 *
 * Pinned by tests/golden/ntthal_dimer.json (od-msspe/src/delta_g.rs:196-230, 5 vectors).
 */
#include "msspe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GAS_R 1.9872
#define ILAS (-300 / 310.15)
#define ILAH 0.0
#define AT_H 2200.0
#define AT_S 6.9
#define MIN_ENTROPY_CUTOFF (-2500.0)
#define MIN_ENTROPY (-3224.0)
#define ABS_ZERO 273.15
#define T37 310.15
#define TINY 0.000001

typedef struct {
    const orc_tables *tb;
    int len1, len2, max_loop;
    unsigned char s1[ORC_MAX_OLIGO + 2]; /* 1-based, N sentinels at 0 and len+1 */
    unsigned char s2[ORC_MAX_OLIGO + 2]; /* oligo 2 REVERSED (reads 3'->5')     */
    double S[ORC_MAX_OLIGO + 1][ORC_MAX_OLIGO + 1];
    double H[ORC_MAX_OLIGO + 1][ORC_MAX_OLIGO + 1];
    double init_H, init_S, RC, salt;
    long ops, n_cells, n_loop, n_end, end_ops;
} dimer_ctx;

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

/* thal.c symmetry_thermo(): even length and self-complementary */
static int self_complementary(const char *s, int n)
{
    if (n % 2) return 0;
    for (int i = 0; i < n / 2; i++) {
        int a = code_of(s[i]), b = code_of(s[n - 1 - i]);
        if (a > 3 || b > 3 || a + b != 3) return 0;
    }
    return 1;
}

/* thal.c saltCorrectS() */
static double salt_correction(double mv, double dv, double dntp)
{
    if (dv <= 0) dntp = dv;
    return 0.368 * (log((mv + 120 * (sqrt(fmax(0.0, dv - dntp)))) / 1000));
}

/*
 * End terms (thal.c LSH / RSH).  `left` selects which side is looked at:
 *   left : outer bases are s1[i-1], s2[j-1]; tstack2[s2[j]][s2[j-1]][s1[i]][s1[i-1]],
 *          dangle3[s2[j]][s2[j-1]][s1[i]], dangle5[s2[j]][s1[i]][s1[i-1]]
 *   right: outer bases are s1[i+1], s2[j+1]; tstack2[s1[i]][s1[i+1]][s2[j]][s2[j+1]],
 *          dangle3[s1[i]][s1[i+1]][s2[j]], dangle5[s1[i]][s2[j]][s2[j+1]]
 */
static void end_term(dimer_ctx *c, int i, int j, int left, double *outS, double *outH)
{
    const orc_tables *tb = c->tb;
    const int a = c->s1[i], b = c->s2[j];
    const long ops_in = c->ops;
    c->n_end++;
    if (!is_pair(a, b)) {
        *outS = -1.0;
        *outH = INFINITY;
        return;
    }
    int oa, ob; /* outer bases on strand 1 / strand 2 */
    double tS, tH, d3S, d3H, d5S, d5H;
    if (left) {
        oa = c->s1[i - 1];
        ob = c->s2[j - 1];
        tS = tb->tstack2S[b][ob][a][oa];
        tH = tb->tstack2H[b][ob][a][oa];
        d3S = tb->d3S[b][ob][a];
        d3H = tb->d3H[b][ob][a];
        d5S = tb->d5S[b][a][oa];
        d5H = tb->d5H[b][a][oa];
    } else {
        oa = c->s1[i + 1];
        ob = c->s2[j + 1];
        tS = tb->tstack2S[a][oa][b][ob];
        tH = tb->tstack2H[a][oa][b][ob];
        d3S = tb->d3S[a][oa][b];
        d3H = tb->d3H[a][oa][b];
        d5S = tb->d5S[a][b][ob];
        d5H = tb->d5H[a][b][ob];
    }
    const double aS = at_S(a, b), aH = at_H(a, b);
    double S1 = aS + tS, H1 = aH + tH;
    double G1 = H1 - T37 * S1;
    double T1 = -INFINITY, T2;
    double S2 = -1.0, H2 = INFINITY, G2;
    c->ops += 5;
    if (!isfinite(H1) || G1 > 0) {
        H1 = INFINITY;
        S1 = -1.0;
        G1 = 1.0;
    }
    const int outer_unpaired = !is_pair(oa, ob);
    int have_d = 0;
    if (outer_unpaired && isfinite(d3H) && isfinite(d5H)) {
        S2 = aS + d3S + d5S;
        H2 = aH + d3H + d5H;
        have_d = 1;
        c->ops += 4;
    } else if (outer_unpaired && isfinite(d3H)) {
        S2 = aS + d3S;
        H2 = aH + d3H;
        have_d = 1;
        c->ops += 2;
    } else if (outer_unpaired && isfinite(d5H)) {
        S2 = aS + d5S;
        H2 = aH + d5H;
        have_d = 1;
        c->ops += 2;
    }
    if (have_d) {
        G2 = H2 - T37 * S2;
        if (!isfinite(H2) || G2 > 0) {
            H2 = INFINITY;
            S2 = -1.0;
            G2 = 1.0;
        }
        T2 = (H2 + c->init_H) / (S2 + c->init_S + c->RC);
        c->ops += 7;
        if (isfinite(H1) && G1 < 0) {
            T1 = (H1 + c->init_H) / (S1 + c->init_S + c->RC);
            c->ops += 5;
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
    /* bare closing pair (AT penalty only); note T1 stays -inf when no dangle option was
     * considered, which makes the bare pair win -- a Primer3 quirk kept on purpose */
    S2 = aS;
    H2 = aH;
    T2 = (H2 + c->init_H) / (S2 + c->init_S + c->RC);
    c->ops += 5;
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
    c->end_ops += c->ops - ops_in;
}

/*
 * Energy of the bulge / interior loop closed by predecessor (pi,pj) and cell (i,j)
 * (thal.c calc_bulge_internal).  Returns the candidate (S,H) including the
 * predecessor's value, or (-1, inf) when rejected.  When `accept_test` is set, the candidate is
 * additionally required to beat the cell's current value by G1 < G2 (fill); traceback calls it
 * with accept_test = 0.
 */
static void loop_candidate(dimer_ctx *c, int pi, int pj, int i, int j, int accept_test,
                           double *outS, double *outH)
{
    const orc_tables *tb = c->tb;
    const unsigned char *s1 = c->s1, *s2 = c->s2;
    const int l1 = i - pi - 1, l2 = j - pj - 1;
    const int idx = l1 + l2 - 1;
    double S = -1.0, H = INFINITY;
    *outS = -1.0;
    *outH = INFINITY;
    c->n_loop++;
    if ((l1 == 0 && l2 > 0) || (l2 == 0 && l1 > 0)) {
        if (l2 == 1 || l1 == 1) {
            /* single-base bulge: the flanking pairs stack on each other */
            H = tb->bulgeH[idx] + tb->stackH[s1[pi]][s1[i]][s2[pj]][s2[j]];
            S = tb->bulgeS[idx] + tb->stackS[s1[pi]][s1[i]][s2[pj]][s2[j]];
            c->ops += 2;
            if (H > 0 || S > 0) {
                H = INFINITY;
                S = -1.0;
            }
            H += c->H[pi][pj];
            S += c->S[pi][pj];
            c->ops += 2;
            if (!isfinite(H)) {
                H = INFINITY;
                S = -1.0;
            }
        } else {
            H = tb->bulgeH[idx] + at_H(s1[pi], s2[pj]) + at_H(s1[i], s2[j]);
            H += c->H[pi][pj];
            S = tb->bulgeS[idx] + at_S(s1[pi], s2[pj]) + at_S(s1[i], s2[j]);
            S += c->S[pi][pj];
            c->ops += 6;
            if (!isfinite(H)) {
                H = INFINITY;
                S = -1.0;
            }
            if (H > 0 && S > 0) {
                H = INFINITY;
                S = -1.0;
            }
        }
    } else if (l1 == 1 && l2 == 1) {
        S = tb->mmS[s1[pi]][s1[pi + 1]][s2[pj]][s2[pj + 1]] +
            tb->mmS[s2[j]][s2[j - 1]][s1[i]][s1[i - 1]];
        S += c->S[pi][pj];
        H = tb->mmH[s1[pi]][s1[pi + 1]][s2[pj]][s2[pj + 1]] +
            tb->mmH[s2[j]][s2[j - 1]][s1[i]][s1[i - 1]];
        H += c->H[pi][pj];
        c->ops += 4;
        if (!isfinite(H)) {
            H = INFINITY;
            S = -1.0;
        }
        if (H > 0 && S > 0) {
            H = INFINITY;
            S = -1.0;
        }
    } else {
        H = tb->interiorH[idx] + tb->tstackH[s1[pi]][s1[pi + 1]][s2[pj]][s2[pj + 1]] +
            tb->tstackH[s2[j]][s2[j - 1]][s1[i]][s1[i - 1]] + (ILAH * abs(l1 - l2));
        H += c->H[pi][pj];
        S = tb->interiorS[idx] + tb->tstackS[s1[pi]][s1[pi + 1]][s2[pj]][s2[pj + 1]] +
            tb->tstackS[s2[j]][s2[j - 1]][s1[i]][s1[i - 1]] + (ILAS * abs(l1 - l2));
        S += c->S[pi][pj];
        c->ops += 10;
        if (!isfinite(H)) {
            H = INFINITY;
            S = -1.0;
        }
        if (H > 0 && S > 0) {
            H = INFINITY;
            S = -1.0;
        }
    }
    if (accept_test) {
        double rS, rH;
        end_term(c, i, j, 0, &rS, &rH);
        const double G1 = H + rH - T37 * (S + rS);
        const double G2 = c->H[i][j] + rH - T37 * (c->S[i][j] + rS);
        c->ops += 9;
        if (!(G1 < G2)) return;
    }
    *outS = S;
    *outH = H;
}

/* thal.c maxTM(): extend the helix by one stacked pair if that raises Tm */
static void stack_step(dimer_ctx *c, int i, int j)
{
    const orc_tables *tb = c->tb;
    double S0 = c->S[i][j], H0 = c->H[i][j], S1, H1, T0, T1, rS, rH;
    end_term(c, i, j, 0, &rS, &rH);
    T0 = (H0 + c->init_H + rH) / (S0 + c->init_S + rS + c->RC);
    c->ops += 6;
    const double stH = tb->stackH[c->s1[i - 1]][c->s1[i]][c->s2[j - 1]][c->s2[j]];
    const double stS = tb->stackS[c->s1[i - 1]][c->s1[i]][c->s2[j - 1]][c->s2[j]];
    if (isfinite(c->H[i - 1][j - 1]) && isfinite(stH)) {
        S1 = c->S[i - 1][j - 1] + stS;
        H1 = c->H[i - 1][j - 1] + stH;
        T1 = (H1 + c->init_H + rH) / (S1 + c->init_S + rS + c->RC);
        c->ops += 8;
    } else {
        S1 = -1.0;
        H1 = INFINITY;
        T1 = (H1 + c->init_H) / (S1 + c->init_S + c->RC);
        c->ops += 4;
    }
    if (S1 < MIN_ENTROPY_CUTOFF) {
        S1 = MIN_ENTROPY;
        H1 = 0.0;
    }
    if (S0 < MIN_ENTROPY_CUTOFF) {
        S0 = MIN_ENTROPY;
        H0 = 0.0;
    }
    c->ops += 3;
    if (T1 > T0) {
        c->S[i][j] = S1;
        c->H[i][j] = H1;
    } else if (T0 >= T1) {
        c->S[i][j] = S0;
        c->H[i][j] = H0;
    }
}

/* thal.c initMatrix() + fillMatrix() */
static void fill(dimer_ctx *c)
{
    for (int i = 1; i <= c->len1; i++)
        for (int j = 1; j <= c->len2; j++) {
            if (is_pair(c->s1[i], c->s2[j])) {
                c->H[i][j] = 0.0;
                c->S[i][j] = MIN_ENTROPY;
            } else {
                c->H[i][j] = INFINITY;
                c->S[i][j] = -1.0;
            }
        }
    for (int i = 1; i <= c->len1; i++)
        for (int j = 1; j <= c->len2; j++) {
            if (!isfinite(c->H[i][j])) continue;
            c->n_cells++;
            double lS, lH;
            end_term(c, i, j, 1, &lS, &lH);
            if (isfinite(lH)) {
                c->S[i][j] = lS;
                c->H[i][j] = lH;
            }
            if (i > 1 && j > 1) {
                stack_step(c, i, j);
                for (int d = 3; d <= c->max_loop + 2; d++) {
                    int ii = i - 1;
                    int jj = -ii - d + (j + i);
                    if (jj < 1) {
                        ii -= abs(jj - 1);
                        jj = 1;
                    }
                    for (; ii > 0 && jj < j; --ii, ++jj) {
                        if (!isfinite(c->H[ii][jj])) continue;
                        double cS, cH;
                        loop_candidate(c, ii, jj, i, j, 1, &cS, &cH);
                        if (cS < MIN_ENTROPY_CUTOFF) {
                            cS = MIN_ENTROPY;
                            cH = 0.0;
                        }
                        c->ops += 1;
                        if (isfinite(cH)) {
                            c->H[i][j] = cH;
                            c->S[i][j] = cS;
                        }
                    }
                }
            }
        }
}

static int nearly(double a, double b)
{
    if (!isfinite(a) || !isfinite(b)) return 0;
    return fabs(a - b) < 1e-5;
}

/* thal.c traceback(): ps1[i-1] = j, ps2[j-1] = i for every pair on the optimal path */
static void traceback(dimer_ctx *c, int i, int j, int *ps1, int *ps2)
{
    ps1[i - 1] = j;
    ps2[j - 1] = i;
    int guard = 4 * (c->len1 + c->len2) + 8;
    while (guard-- > 0) {
        double lS, lH;
        end_term(c, i, j, 1, &lS, &lH);
        c->ops += 2;
        if (nearly(c->S[i][j], lS) && nearly(c->H[i][j], lH)) break;
        int done = 0;
        if (i > 1 && j > 1) {
            const double stS = c->tb->stackS[c->s1[i - 1]][c->s1[i]][c->s2[j - 1]][c->s2[j]];
            const double stH = c->tb->stackH[c->s1[i - 1]][c->s1[i]][c->s2[j - 1]][c->s2[j]];
            c->ops += 4;
            if (nearly(c->S[i][j], stS + c->S[i - 1][j - 1]) &&
                nearly(c->H[i][j], stH + c->H[i - 1][j - 1])) {
                i = i - 1;
                j = j - 1;
                ps1[i - 1] = j;
                ps2[j - 1] = i;
                done = 1;
            }
        }
        for (int d = 3; !done && d <= c->max_loop + 2; ++d) {
            int ii = i - 1;
            int jj = -ii - d + (j + i);
            if (jj < 1) {
                ii -= abs(jj - 1);
                jj = 1;
            }
            for (; !done && ii > 0 && jj < j; --ii, ++jj) {
                double cS, cH;
                loop_candidate(c, ii, jj, i, j, 0, &cS, &cH);
                c->ops += 2;
                if (nearly(c->S[i][j], cS) && nearly(c->H[i][j], cH)) {
                    i = ii;
                    j = jj;
                    ps1[i - 1] = j;
                    ps2[j - 1] = i;
                    done = 1;
                    break;
                }
            }
        }
        if (!done) break; /* thal.c would spin here; never observed, kept finite on purpose */
    }
}

static int setup(dimer_ctx *c, const orc_tables *t, const char *o1, const char *o2, int mode,
                 const orc_thal_args *a)
{
    const char *f = o1, *r = o2;
    if (mode == ORC_THAL_END2) { /* thal.c: type 3 swaps the oligos, then behaves like END1 */
        f = o2;
        r = o1;
    }
    const int n1 = (int)strlen(f), n2 = (int)strlen(r);
    if (n1 < 1 || n2 < 1 || n1 > ORC_MAX_OLIGO || n2 > ORC_MAX_OLIGO) return -1;
    memset(c, 0, sizeof *c);
    c->tb = t;
    c->len1 = n1;
    c->len2 = n2;
    c->max_loop = a->max_loop;
    for (int i = 1; i <= n1; i++) c->s1[i] = (unsigned char)code_of(f[i - 1]);
    for (int j = 1; j <= n2; j++) c->s2[j] = (unsigned char)code_of(r[n2 - j]); /* reversed */
    c->s1[0] = c->s1[n1 + 1] = c->s2[0] = c->s2[n2 + 1] = 4;
    c->init_H = 200;
    c->init_S = -5.7;
    if (self_complementary(f, n1) && self_complementary(r, n2))
        c->RC = GAS_R * log(a->dna_conc / 1000000000.0);
    else
        c->RC = GAS_R * log(a->dna_conc / 4000000000.0);
    c->salt = salt_correction(a->mv, a->dv, a->dntp);
    return 0;
}

int orc_thal_dimer_planes(const orc_tables *t, const char *oligo1, const char *oligo2,
                          const orc_thal_args *a, double *S, double *H)
{
    dimer_ctx *c = (dimer_ctx *)malloc(sizeof *c);
    if (!c) return -1;
    if (setup(c, t, oligo1, oligo2, ORC_THAL_ANY, a)) {
        free(c);
        return -1;
    }
    fill(c);
    for (int i = 1; i <= c->len1; i++)
        for (int j = 1; j <= c->len2; j++) {
            S[(i - 1) * c->len2 + (j - 1)] = c->S[i][j];
            H[(i - 1) * c->len2 + (j - 1)] = c->H[i][j];
        }
    free(c);
    return 0;
}

int orc_thal_dimer(const orc_tables *t, const char *oligo1, const char *oligo2, int mode,
                   const orc_thal_args *a, orc_thal_result *r);

int orc_thal_dimer(const orc_tables *t, const char *oligo1, const char *oligo2, int mode,
                   const orc_thal_args *a, orc_thal_result *r)
{
    dimer_ctx *c = (dimer_ctx *)malloc(sizeof *c);
    if (!c) return -1;
    memset(r, 0, sizeof *r);
    if (setup(c, t, oligo1, oligo2, mode, a)) {
        free(c);
        return -1;
    }
    fill(c);

    /* terminal pick (thal.c thal(), dG criterion of Primer3 >= 2.5) */
    int bi = 0, bj = 0;
    double bestG = INFINITY;
    if (mode == ORC_THAL_ANY) {
        for (int i = 1; i <= c->len1; i++)
            for (int j = 1; j <= c->len2; j++) {
                double rS, rH;
                end_term(c, i, j, 0, &rS, &rH);
                rS = rS + TINY;
                rH = rH + TINY;
                const double G1 = (c->H[i][j] + rH + c->init_H) -
                                  T37 * (c->S[i][j] + rS + c->init_S);
                c->ops += 9;
                if (G1 < bestG) {
                    bestG = G1;
                    bi = i;
                    bj = j;
                }
            }
    } else {
        bi = c->len1; /* 3' end of oligo 1 must be paired */
        const int i = c->len1;
        for (int j = 1; j <= c->len2; j++) {
            double rS, rH;
            end_term(c, i, j, 0, &rS, &rH);
            rS = rS + TINY;
            rH = rH + TINY;
            const double G1 =
                (c->H[i][j] + rH + c->init_H) - T37 * (c->S[i][j] + rS + c->init_S);
            c->ops += 9;
            if (G1 < bestG) {
                bestG = G1;
                bj = j;
            }
        }
    }
    if (!isfinite(bestG)) bi = bj = 1;
    double rS, rH;
    end_term(c, bi, bj, 0, &rS, &rH);
    const double dH = c->H[bi][bj] + rH + c->init_H;
    const double dS = c->S[bi][bj] + rS + c->init_S;
    c->ops += 4;
    r->end1 = bi;
    r->end2 = bj;
    if (isfinite(c->H[bi][bj])) {
        traceback(c, bi, bj, r->ps1, r->ps2);
        /* thal.c drawDimer() */
        int N = 0;
        for (int i = 0; i < c->len1; i++)
            if (r->ps1[i] > 0) ++N;
        for (int j = 0; j < c->len2; j++)
            if (r->ps2[j] > 0) ++N;
        r->n_pairs = N / 2;
        N = (N / 2) - 1;
        const double tm = (dH / (dS + (N * c->salt) + c->RC)) - ABS_ZERO;
        const double G = dH - (a->temp_k * (dS + (N * c->salt)));
        c->ops += 9;
        r->t = tm;
        r->dH = dH;
        r->dS_raw = dS;
        r->dS = dS + (N * c->salt);
        r->dG = G;
        r->no_structure = 0;
    } else {
        r->no_structure = 1;
        r->t = 0.0;
        r->dG = 0.0;
    }
    r->n_cells = c->n_cells;
    r->n_loop_evals = c->n_loop;
    r->n_end_evals = c->n_end;
    r->n_f64_ops = c->ops;
    r->n_end_ops = c->end_ops;
    free(c);
    return 0;
}
