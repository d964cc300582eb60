/*
 * batch.c -- oracle batch drivers (all ordered pairs of a pool; a list of oligos).
 * TEST INFRASTRUCTURE ONLY (see msspe_oracle.h).  Used by tests/ as the checker and by bench.py's
 * cpu_baseline leg as the timed "CPU restatement of the reference path"
 * (od-msspe/src/delta_g.rs:61-153: N^2 ordered pairs, self pairs included, one thal ANY each).
 */
#include "msspe_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_thal_dimer(const orc_tables *t, const char *oligo1, const char *oligo2, int mode,
                   const orc_thal_args *a, orc_thal_result *r);

/*
 * pool: n oligos of k chars each, concatenated (no NULs).  Rows [row0,row1) x all n columns.
 * dg_out       (optional) (row1-row0)*n doubles, +inf where thal finds no structure
 * conflict_out (optional) (row1-row0)*n bytes, 1 where the reference keeps the edge (orc_edge_decision)
 * returns the number of conflicts, or -1.
 */
long orc_pool_pairs(const orc_tables *t, const char *pool, int n, int k, int row0, int row1,
                    const orc_thal_args *args, float threshold, int mode, int n_threads,
                    double *dg_out, unsigned char *conflict_out, double *t_out);

long orc_pool_pairs(const orc_tables *t, const char *pool, int n, int k, int row0, int row1,
                    const orc_thal_args *args, float threshold, int mode, int n_threads,
                    double *dg_out, unsigned char *conflict_out, double *t_out)
{
    if (k < 1 || k >= ORC_MAX_OLIGO || row0 < 0 || row1 > n || row0 > row1) return -1;
    long conflicts = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : conflicts)
#else
    (void)n_threads;
#endif
    for (int i = row0; i < row1; i++) {
        char a[ORC_MAX_OLIGO], b[ORC_MAX_OLIGO];
        memcpy(a, pool + (size_t)i * k, (size_t)k);
        a[k] = 0;
        for (int j = 0; j < n; j++) {
            memcpy(b, pool + (size_t)j * k, (size_t)k);
            b[k] = 0;
            orc_thal_result r;
            orc_thal_dimer(t, a, b, mode, args, &r);
            const size_t o = (size_t)(i - row0) * (size_t)n + (size_t)j;
            int c = 0;
            if (!r.no_structure) c = orc_edge_decision(r.dG, threshold);
            if (dg_out) dg_out[o] = r.no_structure ? INFINITY : r.dG;
            if (t_out) t_out[o] = r.no_structure ? 0.0 : r.t;
            if (conflict_out) conflict_out[o] = (unsigned char)c;
            conflicts += c;
        }
    }
    return conflicts;
}

/* Averaged instrumentation over all ordered pairs of a pool (measured op count). */
int orc_pool_op_stats(const orc_tables *t, const char *pool, int n, int k,
                      const orc_thal_args *args, double *mean_cells, double *mean_loop_evals,
                      double *mean_end_evals, double *mean_f64_ops, double *mean_end_ops);

int orc_pool_op_stats(const orc_tables *t, const char *pool, int n, int k,
                      const orc_thal_args *args, double *mean_cells, double *mean_loop_evals,
                      double *mean_end_evals, double *mean_f64_ops, double *mean_end_ops)
{
    double c = 0, l = 0, e = 0, f = 0, g = 0;
    char a[ORC_MAX_OLIGO], b[ORC_MAX_OLIGO];
    for (int i = 0; i < n; i++) {
        memcpy(a, pool + (size_t)i * k, (size_t)k);
        a[k] = 0;
        for (int j = 0; j < n; j++) {
            memcpy(b, pool + (size_t)j * k, (size_t)k);
            b[k] = 0;
            orc_thal_result r;
            orc_thal_dimer(t, a, b, ORC_THAL_ANY, args, &r);
            c += (double)r.n_cells;
            l += (double)r.n_loop_evals;
            e += (double)r.n_end_evals;
            f += (double)r.n_f64_ops;
            g += (double)r.n_end_ops;
        }
    }
    const double nn = (double)n * (double)n;
    *mean_cells = c / nn;
    *mean_loop_evals = l / nn;
    *mean_end_evals = e / nn;
    *mean_f64_ops = f / nn;
    *mean_end_ops = g / nn;
    return 0;
}

/* primer3_core view of a list of oligos (n x k chars). */
int orc_check_primers(const orc_tables *t, const char *pool, int n, int k, orc_primer_info *out);
int orc_check_primers(const orc_tables *t, const char *pool, int n, int k, orc_primer_info *out)
{
    if (k < 1 || k >= ORC_MAX_OLIGO) return -1;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8)
#endif
    for (int i = 0; i < n; i++) {
        char a[ORC_MAX_OLIGO];
        memcpy(a, pool + (size_t)i * k, (size_t)k);
        a[k] = 0;
        if (orc_check_primer(t, a, &out[i])) rc = -1;
    }
    return rc;
}
