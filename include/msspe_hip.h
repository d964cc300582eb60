/*
 * msspe_hip.h -- C ABI of the MI355X primer-screening engine (libmsspe_hip.so).
 *
 * Drop-in boundary for od-msspe's hot path.  The reference has no library API: it reaches its
 * thermodynamic arithmetic through two child processes chosen by --ntthal / --primer3
 * (/root/reference/od-msspe/src/config.rs:142-147, :179-202).  Every entry point below replaces
 * one of those process boundaries or one in-process stage-A function, and cites it.
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add to od-msspe.
 *
 * Conventions:
 *   - every call returns an int status (MSSPE_OK = 0); msspe_last_error(ctx) gives the text;
 *   - plain pointers and sizes only, caller owns every buffer, no exceptions cross the boundary;
 *   - one context per host thread per device; calls on different contexts are concurrent-safe;
 *   - `_dev` entry points take DEVICE pointers and enqueue on the context's HIP stream; in the steady
 *     state they neither allocate nor synchronise, but the FIRST call with a new chemistry / threshold, or
 *     with a larger problem than any before, builds tables and (re)allocates work buffers, which
 *     synchronises.  Warm a context up with one call of the final size before capturing a HIP graph.  The
 *     others take HOST pointers, copy, run and synchronise;
 *   - there is NO CPU fallback: without a usable gfx950 device msspe_create() fails.
 *
 * Oligo encoding on the device: one uint64 per oligo, base p (0-based from the 5' end) in bits
 * [2p, 2p+1], A=0 C=1 G=2 T=3, k <= 32.  Oligos must be pure ACGT (stage A only emits such words,
 * od-msspe/src/main.rs:167).
 */
#ifndef MSSPE_HIP_H
#define MSSPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msspe_ctx msspe_ctx;

enum msspe_status {
    MSSPE_OK = 0,
    MSSPE_ERR_ARG = 1,      /* NULL pointer, bad range, non-ACGT oligo ...            */
    MSSPE_ERR_K = 2,        /* unsupported oligo length                               */
    MSSPE_ERR_TABLES = 3,   /* thermodynamic parameter files missing / malformed      */
    MSSPE_ERR_DEVICE = 4,   /* HIP error (no device, launch failure, out of memory)   */
    MSSPE_ERR_CAPACITY = 5, /* caller-supplied output capacity exceeded               */
    MSSPE_ERR_NOMEM = 6
};

/* Chemistry handed to ntthal by od-msspe/src/delta_g.rs:93-110 (-mv -dv -n -d -t, all "{:.2}"
 * strings of f32) and defaults of od-msspe/src/constants.rs:7-15. */
typedef struct {
    double mv;        /* monovalent cations, mM          (--mv-conc,  50)   */
    double dv;        /* divalent cations, mM            (--dv-conc,  3)    */
    double dntp;      /* dNTP, mM                        (--dntp-conc, 0)   */
    double dna_conc;  /* oligo concentration, nM         (--dna-conc, 250)  */
    double temp_c;    /* temperature dG is reported at   (--annealing-temp, 25) */
    int max_loop;     /* ntthal -maxloop, 30                                 */
} msspe_chem;

void msspe_chem_ntthal_defaults(msspe_chem *c);   /* 50 / 3 / 0 / 250 nM / 25 C  */
void msspe_chem_primer3_defaults(msspe_chem *c);  /* 50 / 1.5 / 0.6 / 50 nM / 37 C: what
                                                     primer3_core uses for od-msspe's records
                                                     (od-msspe/src/primer.rs:125-140)        */

/* ---- context ---------------------------------------------------------------------------- */

/* device: HIP device ordinal.  params_path: a Primer3-format directory (what ntthal gets via
 * `-path`, od-msspe/src/delta_g.rs:90,107-108), or a consolidated bundle file, or NULL for the
 * bundle shipped next to the library. */
int msspe_create(int device, const char *params_path, msspe_ctx **out);
void msspe_destroy(msspe_ctx *ctx);
const char *msspe_last_error(const msspe_ctx *ctx);
const char *msspe_version(void);
/* Use the caller's hipStream_t (e.g. PyTorch's current stream) for every later call; NULL is
 * HIP's default (null) stream.  msspe_reset_stream() returns to the context's own stream. */
/* Engine options: which kernels later calls may use.  Nothing in the library reads the process
 * environment; the defaults are the product path and the other values exist so that the parity tests
 * can run every stage against every other (the reference has no equivalent: its only knobs are the
 * --ntthal / --primer3 paths, od-msspe/src/config.rs:142-147).  Unknown key / bad value: MSSPE_ERR_ARG.
 *   "pair_kernel"    "auto" | "f64" (f64 register-table kernel first) | "int" (general integer kernel first)
 *   "force_generic"  "0" | "1"    dense one-lane-per-pair kernels only
 *   "split_min_k"    "2".."99"    shortest oligo that goes to the split-table kernel (16; 14- and 15-mers run the row-specialised first stage)
 *   "wave_kernel"    "0" | "1"    one-wave-per-pair f64 kernel in the chain (1)
 *   "list_cap_log2"  "0" | "20".."30"   fixed hand-over list size (0: sized by the call)
 *   "split_lanes"    "0" | "2" | "4" | "8"
 *   "split_list"     "0" | "1"    short oligos: tables too large for the integer list stage go to the split-table
 *                                 kernel's list mode (1) or straight to the f64 kernels (0)
 *   "row_oob"        "0" | "1"    the row-specialised first stage, which reads LDS beyond its allocation and takes the
 *                                 0 gfx950 returns there for "not available" (1; it also needs the per-engine probe to
 *                                 pass), or the general integer kernel, which never leaves its allocation (0: for
 *                                 debugger / trap-handler sessions that raise MEM_VIOL on such reads)
 *   "stage_a_graph"  "0" | "1"    hipGraph replay of stage A's greedy loop (1)
 *   "stage_a_candidates" "0" | "1"  greedy loop over the list of words near the maximum (1), or over all the
 *                                 words on every iteration (0); the winners are the same */
int msspe_set_option(msspe_ctx *ctx, const char *key, const char *value);
/* Facts about the context's device and what it will run (engine-only diagnostics, no reference counterpart):
 *   "device"          HIP ordinal           "n_cu"  compute units
 *   "lds_reads_zero"  1: the per-engine probe found that LDS reads beyond a block's allocation return 0 (every
 *                     gfx950 seen), i.e. the row-specialised first stage may run; 0: the general kernel runs
 *   "row_kernel"      1: 13-mer pools go to the row-specialised first stage with the current options
 *   "stage_a_fast_iterations" / "stage_a_general_iterations" / "stage_a_rebuilds" / "stage_a_idle_iterations"
 *                     the last msspe_kmer_candidates* call's greedy loop: iterations that recorded winners from the
 *                     partitions' leaders alone / after a walk over posting lists, candidate lists made, idle
 *                     iterations at the end of the last batch */
int msspe_get_info(msspe_ctx *ctx, const char *key, long long *value_out);
int msspe_set_stream(msspe_ctx *ctx, void *hip_stream);
int msspe_reset_stream(msspe_ctx *ctx);
int msspe_synchronize(msspe_ctx *ctx);

/* ---- packing ------------------------------------------------------------------------------ */

/* ASCII (n oligos x k chars, no separators) -> packed uint64.  Host-side helper. */
int msspe_pack_oligos(const char *ascii, int n, int k, uint64_t *packed_out);
void msspe_unpack_oligo(uint64_t packed, int k, char *ascii_out /* k+1 bytes */);

/* ---- stage C: all-pairs cross-dimer (replaces run_ntthal, od-msspe/src/delta_g.rs:83-153) - */

/*
 * Evaluates thal ANY (Primer3 2.6.1, what `ntthal -a ANY` computes per input line) for every
 * ORDERED pair (a = pool[i], b = pool[j]), i in [row0,row1), j in [col0,col1), self pairs
 * included (od-msspe/src/delta_g.rs:64-78), and applies the reference's decision
 * "%g-rounded dG parsed as f32 < threshold" (od-msspe/src/delta_g.rs:33-36).
 *
 * Outputs (each optional, device pointers):
 *   row_conflicts  uint32[n]            += number of conflicting columns for each row i
 *   bitmap         uint64[(row1-row0) * words], words = ceil((col1-col0)/64): bit (j-col0) of
 *                  row (i-row0) set iff (i,j) conflicts; the block is cleared by the call and
 *                  conflicts (rare) are then set with atomic ORs
 *   dg             double[(row1-row0)*(col1-col0)] raw dG in cal/mol before %g rounding;
 *                  +inf where thal finds no structure (ntthal prints nothing for such a pair;
 *                  this engine defines "no edge")
 *   tm             double[...] melting temperature t (Celsius), 0 where no structure
 * A call without the dg / tm planes asks for decisions only: where thal()'s terminal pick is a tie that doubles
 * would settle by rounding, such a call may let the pair stand as "no conflict" without settling it, if no
 * structure of the tie can reach the cut.  The bits, counts and edges are the same either way.
 */
int msspe_cross_dimer_dev(msspe_ctx *ctx, const uint64_t *d_pool, int n, int k,
                          const msspe_chem *chem, float dg_threshold,
                          int row0, int row1, int col0, int col1,
                          uint32_t *d_row_conflicts, uint64_t *d_bitmap, double *d_dg,
                          double *d_tm);

/* Host-buffer convenience: packs, uploads, runs the full n x n matrix, downloads. */
int msspe_cross_dimer(msspe_ctx *ctx, const char *pool_ascii, int n, int k,
                      const msspe_chem *chem, float dg_threshold,
                      uint32_t *row_conflicts, uint64_t *bitmap, double *dg, double *tm);

/* The same screen as an edge list (the reference keeps one edge with its dG per conflicting
 * ordered pair, od-msspe/src/delta_g.rs:33-46, and reads it back through Edge::get_dg(), :10-15).
 *   _dev: enqueue only.  d_edges[capacity] receives (a, b, raw double dG) in no particular order, *d_count the
 *         number of conflict edges of the block; a count above the capacity means the list is truncated (the
 *         first `capacity` to arrive are kept).  d_row_conflicts optional.
 *   host: whole pool; edges sorted by (a, b) as the reference's nested loops produce them, dg = the value
 *         get_dg() yields ("%g" -> f32 -> "{:.2}" -> f32).  *count_out = number of conflict edges; when it
 *         exceeds `capacity` the call returns MSSPE_ERR_CAPACITY with the first `capacity` edges (of the
 *         arrival order, then sorted) filled in, so the caller can retry with count_out entries. */
typedef struct {
    uint32_t a, b;   /* pool indices of the ordered pair */
    double dg;       /* cal/mol, unrounded */
} msspe_edge_dev;
typedef struct {
    uint32_t a, b;
    float dg;
} msspe_edge;
int msspe_cross_dimer_edges_dev(msspe_ctx *ctx, const uint64_t *d_pool, int n, int k, const msspe_chem *chem,
                                float dg_threshold, int row0, int row1, int col0, int col1,
                                uint32_t *d_row_conflicts, msspe_edge_dev *d_edges, uint64_t capacity,
                                uint64_t *d_count);
int msspe_cross_dimer_edges(msspe_ctx *ctx, const char *pool_ascii, int n, int k, const msspe_chem *chem,
                            float dg_threshold, msspe_edge *edges, uint64_t capacity, uint64_t *count_out);

/* Number of pairs the last cross-dimer call routed to the generic (slow) kernel because their
 * DP did not fit the fast kernel's register-resident table. */
int msspe_last_overflow_pairs(msspe_ctx *ctx, uint64_t *count_out);

/* Diagnostics of the exact-integer stages of the cross-dimer path since the last call (engine-only,
 * no reference counterpart).  out[0..7]: the matrix-mode kernel -- out[0] = pairs it did not
 * answer because Primer3's double comparisons could go either way (they are retried in list
 * mode), out[1..7] = how many of them met each reason (Tm near-tie, loop == stack/start value with
 * another enthalpy, tie between loops, rejected minimum, tie in the terminal pick, replay
 * mismatch, equal-valued alternative on the optimal path; the row-specialised matrix kernel counts every
 * winning loop candidate of positive enthalpy as a rejected minimum and leaves the entropy half of thal.c's
 * rule to the list mode).  out[8..15]: the same for the list-mode
 * kernel; out[8] is the number of pairs that needed the f64 kernels.  Reading resets the counters. */
int msspe_pair_stage_stats(msspe_ctx *ctx, uint64_t out[16]);
/* Up to 1024 of those pairs (call before msspe_pair_stage_stats, which resets the sample count):
 * out[i] = row << 40 | col << 16 | reason bits (1 Tm, 2 loop == value, 4 loop tie, 8 rejected
 * minimum, 16 pick tie, 32 replay, 64 alternative on the path). */
int msspe_pair_stage_samples(msspe_ctx *ctx, uint64_t *out, int capacity, int *n_out);
/* Host only, no device: the folded tables the all-pairs kernels keep in LDS (csrc/fast_tables.hpp),
 * so that CPU tests can restate the integer recurrence.  fast_S / fast_H / int_g: 2604 entries,
 * int_T: 239 * 64; consts = init_S, RC, salt, temp_k, g_cut, f64 tables usable, int tables usable,
 * entry count. */
int msspe_host_pair_tables(const char *params_path, const msspe_chem *chem, float dg_threshold,
                           double *fast_S, int32_t *fast_H, int32_t *int_g, int32_t *int_T,
                           double consts[8]);
/* The same for the long-oligo kernel (csrc/split_tables.hpp): S / H / g hold info[2] entries, L 1024,
 * X info[3]; info = usable, longest oligo covered, entry count, X count. */
int msspe_host_split_tables(const char *params_path, const msspe_chem *chem, double *S, int32_t *H,
                            int32_t *g, int32_t *L, int32_t *X, int32_t info[4]);

/* Full thal record for explicit pairs (a_i, b_i), i < n -- what `ntthal` prints per input line
 * (od-msspe/src/delta_g.rs:206-230): dS (salt-corrected), dH, dG, t and the base pairs of the
 * traced structure.  mode 1 = ANY, 2 = END1.  Used by the ntthal protocol shim. */
typedef struct {
    double dS, dH, dG, t;
    int32_t no_structure, n_pairs;
    uint8_t ps1[32];   /* ps1[i-1] = partner position in the REVERSED oligo 2 (1-based), 0 = unpaired */
    uint8_t ps2[32];
} msspe_thal_detail;
int msspe_thal_detail_pairs(msspe_ctx *ctx, const char *a_ascii, const char *b_ascii, int n, int k,
                            const msspe_chem *chem, int mode, msspe_thal_detail *out);

/* Measurement aid: when enabled, every launch of the dominant kernel (the all-pairs kernel) is
 * bracketed by HIP events on the context's stream; msspe_profile_read() synchronises and returns
 * the number of launches and their summed device time since the last read. */
int msspe_profile_enable(msspe_ctx *ctx, int on);
int msspe_profile_read(msspe_ctx *ctx, uint64_t *launches, double *total_ms);

/* ---- stage B: per-oligo statistics (replaces check_primers -> primer3_core,
 *      od-msspe/src/primer.rs:143-166) ---------------------------------------------------- */

/*
 * For each oligo: Tm (oligotm, SantaLucia), GC %, and max(0, t) of thal ANY / END1 of the oligo
 * against itself and of thal HAIRPIN -- PRIMER_LEFT_0_{TM,GC_PERCENT,SELF_ANY_TH,SELF_END_TH,
 * HAIRPIN_TH} (od-msspe/src/primer.rs:79-111).  Raw doubles; the text rounding primer3_core /
 * od-msspe apply (%.3f / %.2f -> f32) is msspe_round_fixed_f32().
 */
int msspe_oligo_stats_dev(msspe_ctx *ctx, const uint64_t *d_pool, int n, int k,
                          const msspe_chem *chem, double *d_tm, double *d_gc,
                          double *d_self_any, double *d_self_end, double *d_hairpin);
int msspe_oligo_stats(msspe_ctx *ctx, const char *pool_ascii, int n, int k,
                      const msspe_chem *chem, double *tm, double *gc, double *self_any,
                      double *self_end, double *hairpin);

/* ---- stage A: k-mer candidates (replaces get_segment_manager + find_candidates_kmers,
 *      od-msspe/src/main.rs:196-235, :331-406) -------------------------------------------- */

typedef struct {
    int segment_size;          /* --window-size           500 */
    int overlap_size;          /* --overlap-size          250 */
    int search_window_size;    /* --search-windows-size    50 */
    int kmer_size;             /* --kmer-size              13 */
    int max_iterations;        /* --max-iterations       1000 */
    int max_mismatch_segments; /* --max-mismatch-segments (auto rule is the caller's) */
} msspe_kmer_opt;

/*
 * seqs: n_seq aligned sequences of equal length seq_len, row-major bytes, already upper-cased
 * with U->T (od-msspe/src/main.rs:115-118).  direction 0 = head windows as-is, 1 = tail windows
 * reverse-complemented.  Winners are written in selection order: words as packed uint64 and
 * their frequency.  *n_out = number of winners (<= capacity, else MSSPE_ERR_CAPACITY).
 * words_out / freq_out / n_out are HOST buffers in both variants (at most max_iterations small
 * records); only the sequences differ (host pointer vs. device pointer).
 */
int msspe_kmer_candidates(msspe_ctx *ctx, const uint8_t *seqs, int n_seq, size_t seq_len,
                          const msspe_kmer_opt *opt, int direction,
                          uint64_t *words_out, uint32_t *freq_out, int capacity, int *n_out);
int msspe_kmer_candidates_dev(msspe_ctx *ctx, const uint8_t *d_seqs, int n_seq, size_t seq_len,
                              const msspe_kmer_opt *opt, int direction,
                              uint64_t *words_out, uint32_t *freq_out, int capacity, int *n_out);

/* ---- coverage of the final primer set: replaces the per-segment string search of
 * coverage_report() (od-msspe/src/main.rs:518-594).  seqs as for msspe_kmer_candidates; fwd_words /
 * rev_words: packed primers (msspe_pack_oligos) of the two directions, any order;
 * hit_out[seq * P + partition] (host, n_seq * P bytes, P = (seq_len - segment_size) / overlap_size
 * + 1) = 1 when the segment's head window holds a forward primer or its tail window holds the
 * reverse complement of a reverse primer.  Totals per sequence / partition stay on the host. */
int msspe_segment_coverage(msspe_ctx *ctx, const uint8_t *seqs, int n_seq, size_t seq_len,
                           const msspe_kmer_opt *opt, const uint64_t *fwd_words, int n_fwd,
                           const uint64_t *rev_words, int n_rev, uint8_t *hit_out);
int msspe_segment_coverage_dev(msspe_ctx *ctx, const uint8_t *d_seqs, int n_seq, size_t seq_len,
                               const msspe_kmer_opt *opt, const uint64_t *fwd_words, int n_fwd,
                               const uint64_t *rev_words, int n_rev, uint8_t *hit_out);

/* Staging for hosts without a HIP binding (the reference is Rust): copy a host buffer to the
 * context's device once and use the *_dev entry points on it (the alignment is read by both
 * directions of stage A and by the coverage report). */
int msspe_device_put(msspe_ctx *ctx, const void *host, size_t bytes, void **device_out);
/* The same for a matrix given as separate rows (an alignment held as one string per record): row r is
 * rows[r][0 .. row_bytes[r]) followed by `pad` bytes up to row_len; staged through pinned buffers, the
 * host never builds the rectangular copy. */
int msspe_device_put_rows(msspe_ctx *ctx, const char *const *rows, const size_t *row_bytes, int n_rows,
                          size_t row_len, int pad, void **device_out);
/* The alignment in its compact device form: 2-bit bases + 1 validity bit per column (SURVEY.md 2.2, 8f-3;
 * replaces the per-window char vectors of od-msspe/src/main.rs:163-235).  A packed row is
 * msspe_packed_row_words(row_len) uint64: first (row_len + 31) / 32 words of bases (A 0, C 1, G 2, T 3; column
 * c in bits [2 (c % 32), +1] of word c / 32), then (row_len + 63) / 64 words of validity bits (1 = A / C / G / T;
 * '-', N, IUPAC codes and the padding of short rows are 0, which invalidates every k-mer that covers them,
 * main.rs:167).  msspe_device_put_rows_packed uploads the rows 16 MB at a time and packs each chunk on the
 * device; the *_packed_dev entry points are msspe_kmer_candidates_dev / msspe_segment_coverage_dev on that form
 * (same outputs; 3/8 of the bytes resident and read). */
/* Diagnostics of the last msspe_kmer_candidates* call (engine-only): per winner, iteration << 8 | how the greedy
 * loop selected it -- 1 a partition's leader read off the counts, 2 a word with postings in several partitions,
 * 3 such a word after its key was re-computed, 4 after a walk over posting lists, 0 the all-words loop.
 * *n_out = number of winners (also when capacity is smaller). */
int msspe_kmer_trace(msspe_ctx *ctx, uint32_t *out, int capacity, int *n_out);
size_t msspe_packed_row_words(size_t seq_len);
int msspe_device_put_rows_packed(msspe_ctx *ctx, const char *const *rows, const size_t *row_bytes, int n_rows,
                                 size_t row_len, void **device_out);
int msspe_kmer_candidates_packed_dev(msspe_ctx *ctx, const uint64_t *d_packed, int n_seq, size_t seq_len,
                                     const msspe_kmer_opt *opt, int direction,
                                     uint64_t *words_out, uint32_t *freq_out, int capacity, int *n_out);
/* Both directions of one alignment in one call (the reference runs find_candidates_kmers twice, main.rs:673-690):
 * direction 0 on the context's stream, direction 1 on a second stream of the context from a second host thread -- each
 * direction is a chain of small dependent launches and host round trips, which overlap.  Same results as two
 * msspe_kmer_candidates_packed_dev calls; msspe_kmer_trace and the stage_a_* infos then describe direction 0. */
int msspe_kmer_candidates_both_packed_dev(msspe_ctx *ctx, const uint64_t *d_packed, int n_seq, size_t seq_len,
                                          const msspe_kmer_opt *opt, uint64_t *words_fwd, uint32_t *freq_fwd, int *n_fwd,
                                          uint64_t *words_rev, uint32_t *freq_rev, int *n_rev, int capacity);
int msspe_segment_coverage_packed_dev(msspe_ctx *ctx, const uint64_t *d_packed, int n_seq, size_t seq_len,
                                      const msspe_kmer_opt *opt, const uint64_t *fwd_words, int n_fwd,
                                      const uint64_t *rev_words, int n_rev, uint8_t *hit_out);
int msspe_device_free(msspe_ctx *ctx, void *device);


/* ---- several devices of one node ----------------------------------------------------------
 *
 * One process, one context and one host thread per device.  What shards is the reference's N^2 pair loop
 * (od-msspe/src/delta_g.rs:61-81, called at main.rs:739-752: every ordered pair is independent) and the per-oligo
 * statistics (od-msspe/src/primer.rs:143-166); stage A's greedy loop is sequential and runs on member 0's context
 * (msspe_group_member(g, 0) with the msspe_kmer_candidates* calls).
 *
 * A screen: every member uploads the slice of the pool it "produced" (contiguous n / N candidates), ONE all-gather
 * assembles the packed pool on every device, every member screens its rows against all columns, ONE all-reduce
 * merges the per-primer conflict counts; bitmap rows and edge lists stay with their member until the host call
 * collects them.  Rows are dealt out in groups of 256, round robin (row r belongs to member (r / 256) mod N), so
 * that every member gets a sample of the whole pool whatever its order (msspe_group_rows lists a member's rows).
 * Results are identical to the single-context calls (tests/test_gpu_group.py: members that share one card -- the
 * device-copy transport -- and RCCL as a group of one rank).  NOT YET VERIFIED ON HARDWARE: two or more DISTINCT
 * devices (grouped ncclAllGather / ncclAllReduce on N communicators from one thread, peer access, cross-device copies);
 * no multi-GPU node has been reachable from the build, the two-device test skips on one card.
 *
 * transport: "auto" (NULL) | "rccl" | "device-copy".  "rccl": RCCL over xGMI, librccl.so loaded when the group is
 * made, needs distinct devices.  "device-copy": device-to-device copies and a summing kernel -- for members that
 * share a card (a device may be listed more than once: how the tests rehearse N members on one GPU) and where
 * RCCL cannot be loaded.  auto = rccl for two or more distinct devices if it loads, else device-copy.
 */
typedef struct msspe_group msspe_group;
int msspe_group_create(const int *devices, int n_devices, const char *params_path, const char *transport,
                       msspe_group **out);
void msspe_group_destroy(msspe_group *g);
const char *msspe_group_last_error(const msspe_group *g);
int msspe_group_size(const msspe_group *g);
const char *msspe_group_transport(const msspe_group *g);          /* "single" | "rccl" | "device-copy" */
/* "" unless transport "auto" wanted RCCL and runs the copies instead: then why (librccl.so not loadable,
 * ncclCommInitAll's error, or the failure of the first grouped collective, which msspe_group_create runs with a
 * known answer before any screen depends on the fabric) */
const char *msspe_group_transport_reason(const msspe_group *g);
/* Host only, no device: 1 if RCCL (library: NULL = the names msspe_group_create tries) can be loaded with every
 * collective entry point the group uses, else 0 and the reason in why[0 .. why_capacity) */
int msspe_group_rccl_available(const char *library, char *why, int why_capacity);
msspe_ctx *msspe_group_member(msspe_group *g, int member);        /* owned by the group */
int msspe_group_set_option(msspe_group *g, const char *key, const char *value);   /* msspe_set_option on every member */
/* pool rows member `member` of a group of n_members screens in a pool of n (ascending).  Host only, no device.
 * *n_rows_out is set even when capacity is 0 (sizing call); a capacity that is too small: MSSPE_ERR_CAPACITY */
int msspe_group_rows(int n, int n_members, int member, uint32_t *rows_out, int capacity, int *n_rows_out);
/* msspe_cross_dimer over the group (host buffers; row_conflicts[n] merged, bitmap[n * ceil(n/64)] optional) */
int msspe_cross_dimer_group(msspe_group *g, const char *pool_ascii, int n, int k, const msspe_chem *chem,
                            float dg_threshold, uint32_t *row_conflicts, uint64_t *bitmap);
/* msspe_cross_dimer_edges over the group: same order, same rounding, same capacity contract */
int msspe_cross_dimer_edges_group(msspe_group *g, const char *pool_ascii, int n, int k, const msspe_chem *chem,
                                  float dg_threshold, msspe_edge *edges, uint64_t capacity, uint64_t *count_out);
/* msspe_oligo_stats over the group (contiguous slices of the oligos, one per member) */
int msspe_oligo_stats_group(msspe_group *g, const char *pool_ascii, int n, int k, const msspe_chem *chem,
                            double *tm, double *gc, double *self_any, double *self_end, double *hairpin);

/* ---- text rounding at the reference's process boundary ----------- */

float msspe_round_g_f32(double x);                  /* "%g"   -> f32 (od-msspe/src/delta_g.rs:33-35) */
float msspe_round_fixed_f32(double x, int decimals);/* "%.Nf" -> f32 (od-msspe/src/primer.rs:94-106) */
/* Largest double X such that msspe_round_g_f32(x) < threshold  <=>  x <= X (the decision cut the
 * kernels compare against; exact by construction, found by bisection over doubles). */
double msspe_g_cut(float threshold);

#ifdef __cplusplus
}
#endif
#endif
