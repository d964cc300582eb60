/*
 * thermo_tables.c -- oracle loader for the Primer3 nearest-neighbour parameter tables.
 * TEST INFRASTRUCTURE ONLY (see msspe_oracle.h).
 *
 * Data: /root/reference/od-msspe/primer3_config/{stack,stackmm,tstack2}.{ds,dh}, tstack.dh,
 * tstack_tm_inf.ds, dangle.{ds,dh}, loops.{ds,dh}, triloop.{ds,dh}, tetraloop.{ds,dh}
 * (handed to ntthal at od-msspe/src/delta_g.rs:90,107-108).  Index order and the sentinel rules
 * for N follow SURVEY.md Appendix C.1, which restates Primer3 2.6.1 thal.c getStack/getStackint2/
 * getTstack/getTstack2/getDangle/getLoop/getTriloop/getTetraloop.
 */
#include "msspe_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    char **tok;
    int n;
} toklist;

static void toklist_free(toklist *l)
{
    for (int i = 0; i < l->n; i++) free(l->tok[i]);
    free(l->tok);
    l->tok = NULL;
    l->n = 0;
}

static int toklist_push(toklist *l, const char *s, size_t len, int *cap)
{
    if (l->n == *cap) {
        *cap = *cap ? *cap * 2 : 256;
        char **nt = (char **)realloc(l->tok, (size_t)*cap * sizeof(char *));
        if (!nt) return -1;
        l->tok = nt;
    }
    char *c = (char *)malloc(len + 1);
    if (!c) return -1;
    memcpy(c, s, len);
    c[len] = 0;
    l->tok[l->n++] = c;
    return 0;
}

static int tokenize(const char *text, toklist *l)
{
    int cap = 0;
    l->tok = NULL;
    l->n = 0;
    const char *p = text;
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++;
        if (!*p) break;
        const char *q = p;
        while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') q++;
        if (toklist_push(l, p, (size_t)(q - p), &cap)) return -1;
        p = q;
    }
    return 0;
}

static char *slurp(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (!buf) {
        fclose(f);
        return NULL;
    }
    size_t got = fread(buf, 1, (size_t)n, f);
    buf[got] = 0;
    fclose(f);
    return buf;
}

/* thal.c readDouble(): the token "inf" is +infinity, everything else goes through strtod. */
static double tok_double(const char *s)
{
    if (!strncmp(s, "inf", 3)) return INFINITY;
    return strtod(s, NULL);
}

static int base_code(char c)
{
    switch (c) {
    case 'A': case 'a': case '0': return 0;
    case 'C': case 'c': case '1': return 1;
    case 'G': case 'g': case '2': return 2;
    case 'T': case 't': case '3': return 3;
    default: return 4;
    }
}

/* 16 named sections, in the order tools/make_param_bundle.py writes them */
enum {
    SEC_STACK_DS, SEC_STACKMM_DS, SEC_TSTACK2_DS, SEC_STACK_DH, SEC_STACKMM_DH, SEC_TSTACK2_DH,
    SEC_TSTACK_TM_INF_DS, SEC_TSTACK_DH, SEC_DANGLE_DS, SEC_DANGLE_DH, SEC_LOOPS_DS, SEC_LOOPS_DH,
    SEC_TRILOOP_DS, SEC_TRILOOP_DH, SEC_TETRALOOP_DS, SEC_TETRALOOP_DH, SEC_COUNT
};
static const char *SEC_NAME[SEC_COUNT] = {
    "stack.ds", "stackmm.ds", "tstack2.ds", "stack.dh", "stackmm.dh", "tstack2.dh",
    "tstack_tm_inf.ds", "tstack.dh", "dangle.ds", "dangle.dh", "loops.ds", "loops.dh",
    "triloop.ds", "triloop.dh", "tetraloop.ds", "tetraloop.dh"
};

/* Watson-Crick / mismatch stacks: any N -> (-1, inf); non-finite entry -> (-1, inf). */
static int fill_stack(const toklist *ds, const toklist *dh, double S[5][5][5][5],
                      double H[5][5][5][5])
{
    if (ds->n != 256 || dh->n != 256) return -1;
    int k = 0;
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            for (int c = 0; c < 5; c++)
                for (int d = 0; d < 5; d++) {
                    if (a == 4 || b == 4 || c == 4 || d == 4) {
                        S[a][b][c][d] = -1.0;
                        H[a][b][c][d] = INFINITY;
                    } else {
                        double s = tok_double(ds->tok[k]), h = tok_double(dh->tok[k]);
                        k++;
                        if (!isfinite(s) || !isfinite(h)) {
                            s = -1.0;
                            h = INFINITY;
                        }
                        S[a][b][c][d] = s;
                        H[a][b][c][d] = h;
                    }
                }
    return 0;
}

/* Terminal stacks: a or c = N -> (-1, inf); b or d = N -> (1e-11, 0): how sequence ends are
 * absorbed; non-finite entry -> (-1, inf). */
static int fill_tstack(const toklist *ds, const toklist *dh, double S[5][5][5][5],
                       double H[5][5][5][5])
{
    if (ds->n != 256 || dh->n != 256) return -1;
    int k = 0;
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            for (int c = 0; c < 5; c++)
                for (int d = 0; d < 5; d++) {
                    if (a == 4 || c == 4) {
                        S[a][b][c][d] = -1.0;
                        H[a][b][c][d] = INFINITY;
                    } else if (b == 4 || d == 4) {
                        S[a][b][c][d] = 0.00000000001;
                        H[a][b][c][d] = 0.0;
                    } else {
                        double s = tok_double(ds->tok[k]), h = tok_double(dh->tok[k]);
                        k++;
                        if (!isfinite(s) || !isfinite(h)) {
                            s = -1.0;
                            h = INFINITY;
                        }
                        S[a][b][c][d] = s;
                        H[a][b][c][d] = h;
                    }
                }
    return 0;
}

/* dangle file: first 64 numbers = 3' block, file index X*16+Z*4+Y stored at [X][Y][Z];
 * next 64 = 5' block, file index Z*16+X*4+Y stored at [Z][X][Y]. */
static int fill_dangle(const toklist *ds, const toklist *dh, orc_tables *t)
{
    if (ds->n != 128 || dh->n != 128) return -1;
    int k = 0;
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++)
            for (int y = 0; y < 5; y++) {
                if (i == 4 || j == 4 || y == 4) {
                    t->d3S[i][y][j] = -1.0;
                    t->d3H[i][y][j] = INFINITY;
                } else {
                    double s = tok_double(ds->tok[k]), h = tok_double(dh->tok[k]);
                    k++;
                    if (!isfinite(s) || !isfinite(h)) {
                        s = -1.0;
                        h = INFINITY;
                    }
                    t->d3S[i][y][j] = s;
                    t->d3H[i][y][j] = h;
                }
            }
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++)
            for (int y = 0; y < 5; y++) {
                if (i == 4 || j == 4 || y == 4) {
                    t->d5S[i][j][y] = -1.0;
                    t->d5H[i][j][y] = INFINITY;
                } else {
                    double s = tok_double(ds->tok[k]), h = tok_double(dh->tok[k]);
                    k++;
                    if (!isfinite(s) || !isfinite(h)) {
                        s = -1.0;
                        h = INFINITY;
                    }
                    t->d5S[i][j][y] = s;
                    t->d5H[i][j][y] = h;
                }
            }
    return 0;
}

/* loops file: 30 lines "size interior bulge hairpin" */
static int fill_loops(const toklist *ds, const toklist *dh, orc_tables *t)
{
    if (ds->n != 120 || dh->n != 120) return -1;
    for (int k = 0; k < 30; k++) {
        t->interiorS[k] = tok_double(ds->tok[4 * k + 1]);
        t->bulgeS[k] = tok_double(ds->tok[4 * k + 2]);
        t->hairpinS[k] = tok_double(ds->tok[4 * k + 3]);
        t->interiorH[k] = tok_double(dh->tok[4 * k + 1]);
        t->bulgeH[k] = tok_double(dh->tok[4 * k + 2]);
        t->hairpinH[k] = tok_double(dh->tok[4 * k + 3]);
    }
    return 0;
}

static int fill_tloop(const toklist *l, int keylen, orc_tloop *out, int *n_out)
{
    if (l->n % 2) return -1;
    int n = l->n / 2;
    if (n > ORC_MAX_TLOOP) return -1;
    for (int i = 0; i < n; i++) {
        const char *key = l->tok[2 * i];
        if ((int)strlen(key) != keylen) return -1;
        memset(out[i].key, 0, sizeof out[i].key);
        for (int c = 0; c < keylen; c++) out[i].key[c] = (unsigned char)base_code(key[c]);
        out[i].value = tok_double(l->tok[2 * i + 1]);
    }
    *n_out = n;
    return 0;
}

static int assemble(toklist sec[SEC_COUNT], orc_tables *t)
{
    if (fill_stack(&sec[SEC_STACK_DS], &sec[SEC_STACK_DH], t->stackS, t->stackH)) return -2;
    if (fill_stack(&sec[SEC_STACKMM_DS], &sec[SEC_STACKMM_DH], t->mmS, t->mmH)) return -3;
    if (fill_tstack(&sec[SEC_TSTACK_TM_INF_DS], &sec[SEC_TSTACK_DH], t->tstackS, t->tstackH))
        return -4;
    if (fill_tstack(&sec[SEC_TSTACK2_DS], &sec[SEC_TSTACK2_DH], t->tstack2S, t->tstack2H))
        return -5;
    if (fill_dangle(&sec[SEC_DANGLE_DS], &sec[SEC_DANGLE_DH], t)) return -6;
    if (fill_loops(&sec[SEC_LOOPS_DS], &sec[SEC_LOOPS_DH], t)) return -7;
    if (fill_tloop(&sec[SEC_TRILOOP_DS], 5, t->triS, &t->n_tri_s)) return -8;
    if (fill_tloop(&sec[SEC_TRILOOP_DH], 5, t->triH, &t->n_tri_h)) return -8;
    if (fill_tloop(&sec[SEC_TETRALOOP_DS], 6, t->tetS, &t->n_tet_s)) return -9;
    if (fill_tloop(&sec[SEC_TETRALOOP_DH], 6, t->tetH, &t->n_tet_h)) return -9;
    return 0;
}

int orc_tables_load_dir(const char *dir, orc_tables *t)
{
    toklist sec[SEC_COUNT];
    memset(sec, 0, sizeof sec);
    int rc = 0;
    for (int s = 0; s < SEC_COUNT && !rc; s++) {
        char path[4096];
        size_t dl = strlen(dir);
        snprintf(path, sizeof path, "%s%s%s", dir, (dl && dir[dl - 1] == '/') ? "" : "/",
                 SEC_NAME[s]);
        char *text = slurp(path);
        if (!text) {
            rc = -1;
            break;
        }
        if (tokenize(text, &sec[s])) rc = -1;
        free(text);
    }
    if (!rc) rc = assemble(sec, t);
    for (int s = 0; s < SEC_COUNT; s++) toklist_free(&sec[s]);
    return rc;
}

int orc_tables_load_bundle(const char *file, orc_tables *t)
{
    char *text = slurp(file);
    if (!text) return -1;
    toklist all;
    if (tokenize(text, &all)) {
        free(text);
        return -1;
    }
    free(text);
    toklist sec[SEC_COUNT];
    memset(sec, 0, sizeof sec);
    int caps[SEC_COUNT];
    memset(caps, 0, sizeof caps);
    int rc = 0, i = 0;
    /* header comment lines start with '#': skip tokens until the first '@' */
    while (i < all.n && strcmp(all.tok[i], "@")) i++;
    while (i < all.n && !rc) {
        if (strcmp(all.tok[i], "@") || i + 2 >= all.n) {
            rc = -1;
            break;
        }
        const char *name = all.tok[i + 1];
        int cnt = atoi(all.tok[i + 2]);
        i += 3;
        int s;
        for (s = 0; s < SEC_COUNT; s++)
            if (!strcmp(name, SEC_NAME[s])) break;
        if (s == SEC_COUNT || i + cnt > all.n) {
            rc = -1;
            break;
        }
        for (int k = 0; k < cnt; k++)
            if (toklist_push(&sec[s], all.tok[i + k], strlen(all.tok[i + k]), &caps[s])) rc = -1;
        i += cnt;
    }
    if (!rc) rc = assemble(sec, t);
    for (int s = 0; s < SEC_COUNT; s++) toklist_free(&sec[s]);
    toklist_free(&all);
    return rc;
}

orc_tables *orc_tables_new(void) { return (orc_tables *)calloc(1, sizeof(orc_tables)); }
void orc_tables_free(orc_tables *t) { free(t); }
