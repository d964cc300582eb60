/*
 * stage_a.c -- oracle restatement of od-msspe's k-mer candidate generation.
 * TEST INFRASTRUCTURE ONLY (see msspe_oracle.h).
 *
 * Follows /root/reference/od-msspe/src/main.rs:
 *   reverse_complement              :148-161
 *   find_kmers                      :163-171  (valid = all chars in "ATCGU"; first-occurrence dedup)
 *   partitioning_sequence           :173-181  (full windows at offsets 0, stride, 2*stride ...)
 *   get_sequence_on_search_windows  :183-187
 *   get_segment_manager             :196-235
 *   make_kmer_segments_windows_mapping :237-255
 *   partition_tie_score             :261-283  (f32 accumulation, ascending segment index)
 *   find_most_freq_kmer             :285-329  (recount every iteration; ties: score, then
 *                                              lexicographically smallest word)
 *   find_candidates_kmers           :331-406
 * The recount-per-iteration structure of the reference is kept on purpose: this is the checker,
 * not the product.  Pinned by tests/golden/stage_a_unit.json (main.rs:868-1235).
 */
#include "msspe_oracle.h"

#include <stdlib.h>
#include <string.h>

struct orc_segments {
    int k;
    int n_seg;
    int *seq_index;
    uint16_t *partition_no;
    int *koff[2];   /* n_seg + 1 offsets into words[dir] (in k-mers) */
    char *words[2]; /* flat k-char words                              */
    int n_words[2];
    /* inverted index over (dir, word) */
    int n_keys;     /* distinct (word, dir)                           */
    int *key_of[2]; /* per instance: key id                           */
    int *key_dir;   /* per key                                        */
    char *key_word; /* per key: k chars                               */
    int *post_off;  /* n_keys + 1                                     */
    int *post;      /* segment indices, ascending per key             */
};

void orc_reverse_complement(const char *seq, size_t n, char *out)
{
    for (size_t i = 0; i < n; i++) {
        char c = seq[n - 1 - i];
        switch (c) {
        case 'A': c = 'T'; break;
        case 'T': c = 'A'; break;
        case 'U': c = 'A'; break;
        case 'C': c = 'G'; break;
        case 'G': c = 'C'; break;
        default: break;
        }
        out[i] = c;
    }
}

static int valid_base(char c) { return c == 'A' || c == 'T' || c == 'C' || c == 'G' || c == 'U'; }

int orc_find_kmers(const char *seq, size_t n, int k, char *out, int cap)
{
    int cnt = 0;
    if (k <= 0 || (size_t)k > n) return 0;
    for (size_t p = 0; p + (size_t)k <= n; p++) {
        int ok = 1;
        for (int c = 0; c < k; c++)
            if (!valid_base(seq[p + c])) {
                ok = 0;
                break;
            }
        if (!ok) continue;
        int dup = 0;
        for (int e = 0; e < cnt; e++)
            if (!memcmp(out + (size_t)e * k, seq + p, (size_t)k)) {
                dup = 1;
                break;
            }
        if (dup) continue;
        if (cnt == cap) return cnt;
        memcpy(out + (size_t)cnt * k, seq + p, (size_t)k);
        cnt++;
    }
    return cnt;
}

int orc_partition_count(size_t len, int size, int stride)
{
    if (size <= 0 || stride <= 0 || len < (size_t)size) return 0;
    return (int)((len - (size_t)size) / (size_t)stride) + 1;
}

typedef struct {
    const char *w;
    int dir;
    int seg;
    int slot; /* instance index within words[dir] */
} inst;

static int g_k; /* qsort context (oracle is single-threaded here) */
static int inst_cmp(const void *a, const void *b)
{
    const inst *x = (const inst *)a, *y = (const inst *)b;
    if (x->dir != y->dir) return x->dir - y->dir;
    int c = memcmp(x->w, y->w, (size_t)g_k);
    if (c) return c;
    return x->seg - y->seg;
}

orc_segments *orc_segments_build(const char *const *seqs, const size_t *lens, int n_seq,
                                 const orc_partition_opt *opt)
{
    if (opt->overlap_size < opt->window_size) return NULL; /* main.rs:201-203 panics */
    if (opt->segment_size < opt->window_size || opt->kmer_size < 1 ||
        opt->kmer_size >= ORC_MAX_OLIGO)
        return NULL;
    const int k = opt->kmer_size, W = opt->window_size;
    orc_segments *m = (orc_segments *)calloc(1, sizeof *m);
    m->k = k;
    int total = 0;
    for (int r = 0; r < n_seq; r++)
        total += orc_partition_count(lens[r], opt->segment_size, opt->overlap_size);
    m->n_seg = total;
    m->seq_index = (int *)malloc(sizeof(int) * (size_t)(total + 1));
    m->partition_no = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(total + 1));
    const int per_win = W >= k ? W - k + 1 : 0;
    for (int d = 0; d < 2; d++) {
        m->koff[d] = (int *)malloc(sizeof(int) * (size_t)(total + 1));
        m->words[d] = (char *)malloc((size_t)(total > 0 ? total : 1) * (size_t)(per_win + 1) *
                                     (size_t)k);
    }
    char *win = (char *)malloc((size_t)W + 1);
    char *tmp = (char *)malloc((size_t)(per_win + 1) * (size_t)k);
    int seg = 0, cur[2] = {0, 0};
    for (int r = 0; r < n_seq; r++) {
        const int np = orc_partition_count(lens[r], opt->segment_size, opt->overlap_size);
        for (int j = 0; j < np; j++) {
            const char *part = seqs[r] + (size_t)j * (size_t)opt->overlap_size;
            m->seq_index[seg] = r;
            m->partition_no[seg] = (uint16_t)j;
            /* head window, as-is */
            m->koff[0][seg] = cur[0];
            int c0 = orc_find_kmers(part, (size_t)W, k, m->words[0] + (size_t)cur[0] * k, per_win);
            cur[0] += c0;
            /* tail window, each k-mer reverse-complemented */
            m->koff[1][seg] = cur[1];
            memcpy(win, part + opt->segment_size - W, (size_t)W);
            int c1 = orc_find_kmers(win, (size_t)W, k, tmp, per_win);
            for (int e = 0; e < c1; e++)
                orc_reverse_complement(tmp + (size_t)e * k, (size_t)k,
                                       m->words[1] + (size_t)(cur[1] + e) * k);
            cur[1] += c1;
            seg++;
        }
    }
    m->koff[0][seg] = cur[0];
    m->koff[1][seg] = cur[1];
    m->n_words[0] = cur[0];
    m->n_words[1] = cur[1];
    free(win);
    free(tmp);

    /* inverted index */
    const int n_inst = cur[0] + cur[1];
    inst *all = (inst *)malloc(sizeof(inst) * (size_t)(n_inst > 0 ? n_inst : 1));
    int q = 0;
    for (int d = 0; d < 2; d++)
        for (int s = 0; s < m->n_seg; s++)
            for (int e = m->koff[d][s]; e < m->koff[d][s + 1]; e++) {
                all[q].w = m->words[d] + (size_t)e * k;
                all[q].dir = d;
                all[q].seg = s;
                all[q].slot = e;
                q++;
            }
    g_k = k;
    qsort(all, (size_t)n_inst, sizeof(inst), inst_cmp);
    m->key_of[0] = (int *)malloc(sizeof(int) * (size_t)(cur[0] > 0 ? cur[0] : 1));
    m->key_of[1] = (int *)malloc(sizeof(int) * (size_t)(cur[1] > 0 ? cur[1] : 1));
    m->key_dir = (int *)malloc(sizeof(int) * (size_t)(n_inst > 0 ? n_inst : 1));
    m->key_word = (char *)malloc((size_t)(n_inst > 0 ? n_inst : 1) * (size_t)k);
    m->post_off = (int *)malloc(sizeof(int) * (size_t)(n_inst + 2));
    m->post = (int *)malloc(sizeof(int) * (size_t)(n_inst > 0 ? n_inst : 1));
    int nk = 0;
    for (int e = 0; e < n_inst; e++) {
        if (e == 0 || all[e].dir != all[e - 1].dir || memcmp(all[e].w, all[e - 1].w, (size_t)k)) {
            m->key_dir[nk] = all[e].dir;
            memcpy(m->key_word + (size_t)nk * k, all[e].w, (size_t)k);
            m->post_off[nk] = e;
            nk++;
        }
        m->key_of[all[e].dir][all[e].slot] = nk - 1;
        m->post[e] = all[e].seg;
    }
    m->post_off[nk] = n_inst;
    m->n_keys = nk;
    free(all);
    return m;
}

void orc_segments_free(orc_segments *m)
{
    if (!m) return;
    free(m->seq_index);
    free(m->partition_no);
    for (int d = 0; d < 2; d++) {
        free(m->koff[d]);
        free(m->words[d]);
        free(m->key_of[d]);
    }
    free(m->key_dir);
    free(m->key_word);
    free(m->post_off);
    free(m->post);
    free(m);
}

int orc_segments_count(const orc_segments *m) { return m->n_seg; }
int orc_segment_partition_no(const orc_segments *m, int seg) { return m->partition_no[seg]; }
int orc_segment_seq_index(const orc_segments *m, int seg) { return m->seq_index[seg]; }
int orc_segment_kmer_count(const orc_segments *m, int seg, int dir)
{
    return m->koff[dir][seg + 1] - m->koff[dir][seg];
}
const char *orc_segment_kmer(const orc_segments *m, int seg, int dir, int idx)
{
    return m->words[dir] + (size_t)(m->koff[dir][seg] + idx) * m->k;
}
int orc_mapping_key_count(const orc_segments *m) { return m->n_keys; }
int orc_mapping_postings(const orc_segments *m, const char *word, int dir)
{
    for (int key = 0; key < m->n_keys; key++)
        if (m->key_dir[key] == dir && !memcmp(m->key_word + (size_t)key * m->k, word, (size_t)m->k))
            return m->post_off[key + 1] - m->post_off[key];
    return 0;
}

/* partition_tie_score (main.rs:261-283) */
static float tie_score(const orc_segments *m, int key, const unsigned char *ignored,
                       const int *coverage, int *seen_stamp, int stamp)
{
    float score = 0.0f;
    for (int e = m->post_off[key]; e < m->post_off[key + 1]; e++) {
        const int idx = m->post[e];
        if (ignored[idx]) continue;
        const int p = m->partition_no[idx];
        if (seen_stamp[p] != stamp) {
            seen_stamp[p] = stamp;
            score += 1.0f / ((float)coverage[p] + 1.0f);
        }
    }
    return score;
}

/* find_most_freq_kmer (main.rs:285-329).  Returns key id or -1; *freq_out = max frequency. */
static int most_freq(const orc_segments *m, int direction, const unsigned char *ignored,
                     const int *coverage, int *counts, int *seen_stamp, int *stamp, int *freq_out)
{
    memset(counts, 0, sizeof(int) * (size_t)(m->n_keys > 0 ? m->n_keys : 1));
    int any = 0;
    for (int s = 0; s < m->n_seg; s++) {
        if (ignored[s]) continue;
        for (int e = m->koff[direction][s]; e < m->koff[direction][s + 1]; e++) {
            counts[m->key_of[direction][e]]++;
            any = 1;
        }
    }
    if (!any) return -1;
    int max_freq = 0;
    for (int key = 0; key < m->n_keys; key++)
        if (m->key_dir[key] == direction && counts[key] > max_freq) max_freq = counts[key];
    int best = -1;
    float best_score = 0.0f;
    for (int key = 0; key < m->n_keys; key++) {
        if (m->key_dir[key] != direction || counts[key] != max_freq) continue;
        ++*stamp;
        const float sc = tie_score(m, key, ignored, coverage, seen_stamp, *stamp);
        /* keys are visited in ascending word order, so "greater score, else first seen" gives the
         * lexicographically smallest word among equal scores (main.rs:320-324) */
        if (best < 0 || sc > best_score) {
            best = key;
            best_score = sc;
        }
    }
    *freq_out = max_freq;
    return best;
}

int orc_find_candidates(const orc_segments *m, int direction, int max_iterations,
                        int max_mismatch_segments, orc_candidate *out, int cap)
{
    unsigned char *ignored = (unsigned char *)calloc((size_t)m->n_seg + 1, 1);
    int *coverage = (int *)calloc(65536, sizeof(int));
    int *seen = (int *)calloc(65536, sizeof(int));
    int *seen_cov = (int *)calloc(65536, sizeof(int));
    int *counts = (int *)malloc(sizeof(int) * (size_t)(m->n_keys + 1));
    int stamp = 0, n_out = 0, cov_stamp = 0;
    for (int it = 0; it < max_iterations; it++) {
        int freq = 0;
        const int key = most_freq(m, direction, ignored, coverage, counts, seen, &stamp, &freq);
        if (key < 0) break;
        if (freq == 1) break;
        if (n_out < cap) {
            memcpy(out[n_out].word, m->key_word + (size_t)key * m->k, (size_t)m->k);
            out[n_out].word[m->k] = 0;
            out[n_out].frequency = freq;
        }
        n_out++;
        ++cov_stamp;
        for (int e = m->post_off[key]; e < m->post_off[key + 1]; e++) {
            const int idx = m->post[e];
            ignored[idx] = 1;
            const int p = m->partition_no[idx];
            if (seen_cov[p] != cov_stamp) {
                seen_cov[p] = cov_stamp;
                coverage[p] += 1;
            }
        }
        if (freq < max_mismatch_segments) break;
    }
    free(ignored);
    free(coverage);
    free(seen);
    free(seen_cov);
    free(counts);
    return n_out < cap ? n_out : cap;
}

int orc_find_most_freq_kmer(const orc_segments *m, int direction, orc_candidate *out)
{
    unsigned char *ignored = (unsigned char *)calloc((size_t)m->n_seg + 1, 1);
    int *coverage = (int *)calloc(65536, sizeof(int));
    int *seen = (int *)calloc(65536, sizeof(int));
    int *counts = (int *)malloc(sizeof(int) * (size_t)(m->n_keys + 1));
    int stamp = 0, freq = 0;
    const int key = most_freq(m, direction, ignored, coverage, counts, seen, &stamp, &freq);
    if (key >= 0) {
        memcpy(out->word, m->key_word + (size_t)key * m->k, (size_t)m->k);
        out->word[m->k] = 0;
        out->frequency = freq;
    }
    free(ignored);
    free(coverage);
    free(seen);
    free(counts);
    return key >= 0 ? 1 : 0;
}
