/*
 * oligo_stats.c -- oracle restatement of what primer3_core reports for one oligo, and of the text
 * rounding od-msspe applies when it reads Primer3 / ntthal output back.
 * TEST INFRASTRUCTURE ONLY (see msspe_oracle.h).
 *
 * Reference call sites: od-msspe/src/primer.rs:125-166 (Boulder-IO record with only size/Tm
 * bounds, so primer3_core runs with ITS default chemistry: mv 50, dv 1.5, dNTP 0.6, DNA 50 nM),
 * od-msspe/src/primer.rs:67-114 (values parsed back as f32), od-msspe/src/delta_g.rs:33-36
 * (dG token parsed as f32 and compared with the threshold).
 * Arithmetic: Primer3 2.6.1 src/oligotm.c oligotm() with tm_method = santalucia_auto and
 * salt_corrections = santalucia (the 2.6.1 defaults), restated per SURVEY.md Appendix C.2;
 * libprimer3.cc oligo_compl_thermod / oligo_hairpin / align_thermod (max(0, t)).
 * Pinned by tests/golden/primer3_check_primers.json (od-msspe/src/primer.rs:238-250).
 */
#include "msspe_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_thal_dimer(const orc_tables *t, const char *oligo1, const char *oligo2, int mode,
                   const orc_thal_args *a, orc_thal_result *r);
int orc_thal_hairpin(const orc_tables *t, const char *oligo, const orc_thal_args *a,
                     orc_thal_result *r);

void orc_thal_default_args(orc_thal_args *a)
{
    /* od-msspe/src/constants.rs:7-15 -> ntthal -mv 50 -dv 3 -n 0 -d 250 -t 25 */
    a->mv = 50.0;
    a->dv = 3.0;
    a->dntp = 0.0;
    a->dna_conc = 250.0;
    a->temp_k = 25.0 + 273.15;
    a->max_loop = 30;
}

void orc_p3_default_args(orc_thal_args *a)
{
    /* Primer3 2.6.1 p3 global defaults (PRIMER_SALT_MONOVALENT 50, PRIMER_SALT_DIVALENT 1.5,
     * PRIMER_DNTP_CONC 0.6, PRIMER_DNA_CONC 50), thal temp 37 C */
    a->mv = 50.0;
    a->dv = 1.5;
    a->dntp = 0.6;
    a->dna_conc = 50.0;
    a->temp_k = 310.15;
    a->max_loop = 30;
}

int orc_thal(const orc_tables *t, const char *oligo1, const char *oligo2, int mode,
             const orc_thal_args *a, orc_thal_result *r)
{
    if (mode == ORC_THAL_HAIRPIN) return orc_thal_hairpin(t, oligo1, a, r);
    if (mode == ORC_THAL_ANY || mode == ORC_THAL_END1 || mode == ORC_THAL_END2)
        return orc_thal_dimer(t, oligo1, oligo2, mode, a, r);
    return -1;
}

/* SantaLucia (1998) unified NN parameters, 5'->3' dinucleotide; dS in 0.1 cal/(K mol) and dH in
 * 100 cal/mol, both sign-flipped, exactly as oligotm.c keeps them in integers. */
static const int NN_S[4][4] = {
    /*        A    C    G    T  */
    /* A */ {222, 224, 210, 204},
    /* C */ {227, 199, 272, 210},
    /* G */ {222, 244, 199, 224},
    /* T */ {213, 222, 227, 222},
};
static const int NN_H[4][4] = {
    {79, 84, 78, 72},
    {85, 80, 106, 78},
    {82, 98, 80, 84},
    {72, 82, 85, 79},
};

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

static int symmetric(const char *s, int n)
{
    if (n % 2) return 0;
    for (int i = 0; i < n / 2; i++) {
        int a = code_of(s[i]), b = code_of(s[n - 1 - i]);
        if (a > 3 || b > 3 || a + b != 3) return 0;
    }
    return 1;
}

double orc_oligotm(const char *oligo, double dna_conc, double mv, double dv, double dntp)
{
    const int n = (int)strlen(oligo);
    if (n < 2) return -999999.9999;
    for (int i = 0; i < n; i++)
        if (code_of(oligo[i]) > 3) return -999999.9999;
    const int sym = symmetric(oligo, n);
    int dh = 0, ds = 0;
    if (sym) ds += 14;
    for (int end = 0; end < 2; end++) {
        const int c = code_of(end ? oligo[n - 1] : oligo[0]);
        if (c == 0 || c == 3) {
            ds += -41;
            dh += -23;
        } else {
            ds += 28;
            dh += -1;
        }
    }
    for (int i = 0; i + 1 < n; i++) {
        ds += NN_S[code_of(oligo[i])][code_of(oligo[i + 1])];
        dh += NN_H[code_of(oligo[i])][code_of(oligo[i + 1])];
    }
    double delta_H = dh * -100.0;
    double delta_S = ds * -0.1;
    /* divalent_to_monovalent() */
    if (dv == 0) dntp = 0;
    if (dv < 0 || dntp < 0) return -999999.9999;
    if (dv < dntp) dv = dntp;
    double K_mM = mv + 120 * (sqrt(dv - dntp));
    delta_S = delta_S + 0.368 * (n - 1) * log(K_mM / 1000.0);
    double Tm;
    if (sym)
        Tm = delta_H / (delta_S + 1.987 * log(dna_conc / 1000000000.0)) - 273.15;
    else
        Tm = delta_H / (delta_S + 1.987 * log(dna_conc / 4000000000.0)) - 273.15;
    return Tm;
}

double orc_gc_percent(const char *oligo)
{
    const int n = (int)strlen(oligo);
    int gc = 0;
    for (int i = 0; i < n; i++) {
        const int c = code_of(oligo[i]);
        if (c == 1 || c == 2) gc++;
    }
    return n ? 100.0 * ((double)gc) / n : 0.0;
}

/* The reference filters an edge twice: parse_ntthal_output keeps it when the %g text of dG, parsed as
 * f32, is below the threshold (od-msspe/src/delta_g.rs:33-36) and stores it as "{:.2}" (:45); main.rs:758
 * then re-parses that text (delta_g.rs:10-15) and tests `< threshold` again.  Below |dG| = 1000 the
 * first text has more than two decimals, so the second test can drop an edge the first one kept. */
int orc_edge_decision(double dG, float threshold)
{
    const float first = orc_round_g_f32(dG);
    if (!(first < threshold)) return 0;
    return orc_round_fixed_f32((double)first, 2) < threshold;
}

float orc_round_g_f32(double x)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%g", x);
    return strtof(buf, NULL);
}

float orc_round_fixed_f32(double x, int decimals)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.*f", decimals, x);
    return strtof(buf, NULL);
}

int orc_check_primer(const orc_tables *t, const char *oligo, orc_primer_info *out)
{
    orc_thal_args a;
    orc_thal_result r;
    orc_p3_default_args(&a);
    memset(out, 0, sizeof *out);
    out->tm = orc_oligotm(oligo, a.dna_conc, a.mv, a.dv, a.dntp);
    out->gc = orc_gc_percent(oligo);
    if (orc_thal_dimer(t, oligo, oligo, ORC_THAL_ANY, &a, &r)) return -1;
    out->self_any_th = (r.no_structure || r.t < 0.0) ? 0.0 : r.t;
    if (orc_thal_dimer(t, oligo, oligo, ORC_THAL_END1, &a, &r)) return -1;
    out->self_end_th = (r.no_structure || r.t < 0.0) ? 0.0 : r.t;
    if (orc_thal_hairpin(t, oligo, &a, &r)) return -1;
    out->hairpin_th = (r.no_structure || r.t < 0.0) ? 0.0 : r.t;
    out->tm_f32 = orc_round_fixed_f32(out->tm, 3);
    out->gc_f32 = orc_round_fixed_f32(out->gc, 3);
    out->self_any_f32 = orc_round_fixed_f32(out->self_any_th, 2);
    out->self_end_f32 = orc_round_fixed_f32(out->self_end_th, 2);
    out->hairpin_f32 = orc_round_fixed_f32(out->hairpin_th, 2);
    return 0;
}

int orc_pair_conflict(const orc_tables *t, const char *a, const char *b,
                      const orc_thal_args *args, float threshold, double *dg_out)
{
    orc_thal_result r;
    if (orc_thal_dimer(t, a, b, ORC_THAL_ANY, args, &r)) return -1;
    if (r.no_structure) { /* ntthal prints nothing; this build defines "no edge" (App. B) */
        if (dg_out) *dg_out = INFINITY;
        return 0;
    }
    if (dg_out) *dg_out = r.dG;
    return orc_edge_decision(r.dG, threshold);
}

int orc_is_run(const char *kmer)
{
    int runs = 0;
    char last = ' ';
    for (const char *p = kmer; *p; p++) {
        if (*p == last) runs += 1;
        else runs = 0;
        last = *p;
    }
    return runs >= 5;
}

void orc_tm_stat(const float *tm, int n, int sample_divisor, float *mean, float *std)
{
    /* main.rs:462-467: f32 sequential sum / n ; sigma from crate std-dev 0.1.0 (divisor
     * unverifiable offline: exposed as a switch, PARITY UNPINNED) */
    float sum = 0.0f;
    for (int i = 0; i < n; i++) sum += tm[i];
    const float m = sum / (float)n;
    float acc = 0.0f;
    for (int i = 0; i < n; i++) {
        const float d = tm[i] - m;
        acc += d * d;
    }
    const float div = (float)(sample_divisor ? (n - 1) : n);
    *mean = m;
    *std = (n > (sample_divisor ? 1 : 0)) ? sqrtf(acc / div) : 0.0f;
}
