// kmer_stage.hip -- stage A on the device: k-mer candidate generation.
//
// Replaces get_segment_manager + make_kmer_segments_windows_mapping + find_candidates_kmers
// (/root/reference/od-msspe/src/main.rs:196-255, :261-406; exact semantics in SURVEY.md Appendix A):
//   1. every aligned genome is cut into segments (size 500, stride 250); the k-mers of the head
//      window (direction 0) or of the tail window, reverse-complemented (direction 1), are
//      extracted: valid = k consecutive A/C/G/T columns, first occurrence per window only;
//   2. an inverted index word -> ascending segment list is built (radix sort of 2k-bit keys);
//   3. the greedy loop picks, up to max_iterations times, the word present in most uncovered
//      segments (ties: partition_tie_score in f32, then the lexicographically smallest word),
//      covers its segments and decrements the live counts of every word they hold.
// The reference rebuilds a string-keyed HashMap over all live segments on every iteration
// (O(iterations x instances)); here the counts are maintained incrementally, which is exact
// because a segment is covered at most once.
//
// Kernels are HBM-bound integer/byte work: instance arrays are laid out segment-major so that
// extraction writes and the cover step's reads are coalesced; the sort is rocPRIM's radix sort.
#include "kmer_stage.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <cstddef>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

namespace msspe {

namespace {

// Loop state of the greedy selection, resident on the device so that one iteration is a fixed
// sequence of launches with constant arguments (hipGraph-replayable, no host round trip).
struct Status {
    int maxf;          // highest live frequency of this iteration
    unsigned n_tied;   // words tied at maxf
    int winner;        // id of the selected word
    int n_win;         // winners recorded so far
    int stop;          // 1: the loop has ended (every kernel becomes a no-op)
    int stop_next;     // 1: end after this iteration's push (frequency < max_mismatch_segments)
    int max_iter;      // --max-iterations
    int min_freq;      // --max-mismatch-segments
    unsigned n_long;   // tied words with a long posting list (kept at the back of `tied`)
    unsigned ticket;   // last-block detection
    unsigned long long best;   // winner_key() maximum over the tied words
    // candidate-list loop (k_select / k_cover<true>): every word whose live count is >= theta is in cand[]
    int theta;
    unsigned n_cand;
    int need_rebuild;  // 1: the candidates' maximum fell below theta; the rest of the batch is a no-op
    unsigned it1;      // stamp of the iteration in flight (n_win + 1)
    int last_max;      // the maximum the last k_max_count saw (the host's capacity check)
    int too_many;      // 1: the candidate list is longer than k_select reads; the host runs a five-launch batch
    unsigned long long best2[2];   // `best` of the candidate-list loop, alternating with the iteration
};
static_assert(sizeof(Status) <= 128, "the winners' arrays start 128 bytes into the status buffer");

// Segment ids are genome-major (seg = genome * P + partition: the order the reference walks
// them in), but a winner's postings are mostly one partition of many genomes.  The per-segment
// tables the cover step gathers from (word ids, covered flags) are therefore stored
// partition-major, so that those rows are neighbours in memory (same pages, same cache lines).
__device__ __forceinline__ uint32_t row_of(uint32_t seg, uint32_t P, uint32_t G)
{
    return (seg % P) * G + seg / P;
}

__device__ __forceinline__ int base2(uint8_t c)
{
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

// Column `col` of record `rec` as an ASCII letter ('N' for anything that is not A, C, G or T), from the
// byte matrix or from the packed rows (2-bit bases + validity bit): consecutive columns of a packed row
// share a word, so the lanes of a wave that read consecutive columns issue one broadcast load each.
__device__ __forceinline__ uint8_t seq_at(const SeqView &v, size_t rec, size_t col)
{
    if (v.ascii) return v.ascii[rec * v.seq_len + col];
    const size_t bw = (v.seq_len + 31) / 32, rw = bw + (v.seq_len + 63) / 64;
    const uint64_t *row = v.packed + rec * rw;
    const bool ok = (row[bw + (col >> 6)] >> (col & 63)) & 1ull;
    const int code = (int)((row[col >> 5] >> (2 * (col & 31))) & 3ull);
    return ok ? (uint8_t)"ACGT"[code] : (uint8_t)'N';
}

// ASCII rows -> packed rows: one thread per output word (32 columns of bases or 64 of validity).
__global__ void __launch_bounds__(256) k_pack_rows(const uint8_t *ascii, int n_rows, size_t row_len, uint64_t *packed)
{
    const size_t bw = (row_len + 31) / 32, rw = bw + (row_len + 63) / 64;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_rows * rw) return;
    const size_t r = t / rw, w = t % rw;
    const uint8_t *row = ascii + r * row_len;
    uint64_t out = 0;
    if (w < bw) {
        const size_t c0 = w * 32;
        for (int q = 0; q < 32 && c0 + q < row_len; ++q) {
            const int b = base2(row[c0 + q]);
            out |= (uint64_t)(b >= 0 ? b : 0) << (2 * q);   // columns without a base: bits 00, validity 0
        }
    } else {
        const size_t c0 = (w - bw) * 64;
        for (int q = 0; q < 64 && c0 + q < row_len; ++q) out |= (uint64_t)(base2(row[c0 + q]) >= 0) << q;
    }
    packed[t] = out;
}

// One thread per (segment, window position).  key = lexicographic-order code of the emitted word
// (first base in the most significant bits); invalid / duplicate positions get the sentinel.
// Key = uint32_t when the 2k + 1 key bits fit (k <= 15: a third less sort traffic), uint64_t otherwise.
template <class Key>
__global__ void __launch_bounds__(256) k_extract(const SeqView seqs, size_t seq_len, int n_seg,
                                                 int P, int seg_size, int stride, int W, int k,
                                                 int direction, int per, int wins_per_block,
                                                 Key *key_out, uint32_t *val_out)
{
    extern __shared__ unsigned char smem[];
    uint64_t *fkey = (uint64_t *)smem;                           // [wins_per_block][per] forward keys
    uint8_t *win = smem + sizeof(uint64_t) * wins_per_block * per;   // [wins_per_block][W]
    const int t = threadIdx.x;
    const int w = t / per, p = t % per;
    const long seg0 = (long)blockIdx.x * wins_per_block;
    // stage the windows
    for (int e = t; e < wins_per_block * W; e += blockDim.x) {
        const long seg = seg0 + e / W;
        uint8_t c = 'N';
        if (seg < n_seg) {
            const long rec = seg / P, part = seg % P;
            const size_t col = (size_t)part * stride + (direction ? seg_size - W : 0) + (e % W);
            c = seq_at(seqs, (size_t)rec, col);
        }
        win[e] = c;
    }
    __syncthreads();
    const long seg = seg0 + w;
    const bool mine = w < wins_per_block && seg < n_seg;
    uint64_t fk = ~0ull, word = 0;
    if (mine) {
        bool ok = true;
        uint64_t f = 0, rc = 0;
        for (int q = 0; q < k; ++q) {
            const int b = base2(win[w * W + p + q]);
            ok = ok && b >= 0;
            f = (f << 2) | (uint64_t)(b & 3);                      // forward word, MSB first
            rc |= (uint64_t)(3 - (b & 3)) << (2 * q);              // reverse complement, MSB first
        }
        if (ok) {
            fk = f;
            word = direction ? rc : f;
        }
        fkey[w * per + p] = fk;
    }
    __syncthreads();
    if (!mine) return;
    bool keep = fk != ~0ull;
    for (int q = 0; keep && q < p; ++q) keep = fkey[w * per + q] != fk;   // first occurrence only
    const size_t inst = (size_t)seg * per + p;
    key_out[inst] = (Key)(keep ? word : (1ull << (2 * k)));
    val_out[inst] = (uint32_t)inst;
}

template <class Key>
__global__ void k_heads(const Key *key, size_t n, uint64_t sentinel, uint32_t *head)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    head[i] = (k != sentinel && (i == 0 || key[i - 1] != k)) ? 1u : 0u;
}

template <class Key>
__global__ void k_index(const Key *key, const uint32_t *val, const uint32_t *head,
                        const uint32_t *hscan, size_t n, uint64_t sentinel, int per, int P, int G,
                        int32_t *kid_of_inst, uint32_t *post, uint32_t *post_off, uint64_t *ukeys)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    const uint32_t inst = val[i];
    const uint32_t seg = inst / (uint32_t)per, q = inst % (uint32_t)per;
    const size_t slot = (size_t)row_of(seg, (uint32_t)P, (uint32_t)G) * per + q;
    if (k == sentinel) {
        kid_of_inst[slot] = -1;
        return;
    }
    const uint32_t kid = hscan[i] + head[i] - 1;
    kid_of_inst[slot] = (int32_t)kid;
    post[i] = seg;
    if (head[i]) {
        post_off[kid] = (uint32_t)i;
        ukeys[kid] = k;
    }
}

// lower bound of the sentinel in the sorted key array (single thread, log n steps)
template <class Key>
__global__ void k_tail(const Key *key, size_t n, uint64_t sentinel, uint32_t *post_off, int M)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (key[mid] < sentinel) lo = mid + 1;
        else hi = mid;
    }
    post_off[M] = (uint32_t)lo;
}

__global__ void k_init_counts(const uint32_t *post_off, int M, int32_t *count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) count[i] = (int32_t)(post_off[i + 1] - post_off[i]);
}

// ---- greedy loop -------------------------------------------------------------------------------
// One iteration = k_max_count -> k_collect_tied -> k_tie_scores + k_tie_long -> k_cover, all with
// constant arguments.  Single-thread bookkeeping rides on the last block of k_max_count (decide)
// and of k_cover (record the winner), found with a ticket counter.

constexpr unsigned kLongList = 512;
constexpr int kTieUnroll = 4;        // 1024-posting chunks whose loads k_tie_long keeps in flight   // posting lists above this get a whole block in k_tie_long

template <bool kRebuild>
__global__ void __launch_bounds__(256) k_max_count(const int32_t *count, int M, Status *st)
{
    __shared__ int part[4];
    if (st->stop) return;
    int m = 0;
    {
        // 16-byte loads (the count array is 256-byte aligned), the last M % 4 words one by one
        const int4 *c4 = reinterpret_cast<const int4 *>(count);
        const int M4 = M >> 2, tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
        for (int i = tid; i < M4; i += stride) {
            const int4 v = c4[i];
            m = max(m, max(max(v.x, v.y), max(v.z, v.w)));
        }
        for (int i = (M4 << 2) + tid; i < M; i += stride) m = max(m, count[i]);
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic per block: same-address atomics serialise
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m > 0) atomicMax(&st->maxf, m);
        __threadfence();
        if (atomicAdd(&st->ticket, 1u) == gridDim.x - 1) {
            // last block: the maximum is final.  main.rs:344-366: the loop ends when no word is
            // left, when the best word is in one segment only, or after max_iterations winners
            const int mf = atomicMax(&st->maxf, 0);
            if (mf <= 1 || st->n_win >= st->max_iter || st->stop_next) st->stop = 1;
            st->last_max = mf;
            if (kRebuild) {
                // head of a batch of the candidate-list loop: counts only fall, so until the maximum drops
                // below theta the winner and everything tied with it are among the words collected now
                st->theta = max(2, mf / 2);
                st->n_cand = 0;
                st->need_rebuild = 0;
                st->too_many = 0;
                st->maxf = 0;
                st->best2[0] = 0;
                st->best2[1] = 0;
            }
            st->n_tied = 0;
            st->n_long = 0;
            st->ticket = 0;
        }
    }
}

// Words tied at the maximum: short posting lists go to the front of `tied`, long ones to the back
// (slot M-1-j).  Wave-aggregated so that a million-way tie costs thousands of atomics, not millions.
__global__ void __launch_bounds__(256) k_collect_tied(const int32_t *count, int M, Status *st,
                                                      const uint32_t *post_off, uint32_t *tied)
{
    if (st->stop) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool hit = i < M && count[i] == st->maxf;
    const bool is_long = hit && post_off[i + 1] - post_off[i] > kLongList;
    const unsigned long long ms = __ballot(hit && !is_long), ml = __ballot(is_long);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (ms) {
        unsigned base = 0;
        if (lane == __ffsll((long long)ms) - 1) base = atomicAdd(&st->n_tied, (unsigned)__popcll(ms));
        base = __shfl(base, __ffsll((long long)ms) - 1);
        if (hit && !is_long) tied[base + (unsigned)__popcll(ms & below)] = (uint32_t)i;
    }
    if (ml) {
        unsigned base = 0;
        if (lane == __ffsll((long long)ml) - 1) base = atomicAdd(&st->n_long, (unsigned)__popcll(ml));
        base = __shfl(base, __ffsll((long long)ml) - 1);
        if (is_long) tied[(unsigned)M - 1u - (base + (unsigned)__popcll(ml & below))] = (uint32_t)i;
    }
}

// Words with a live count of at least theta, in any order (wave-aggregated append); bit 31 marks a long
// posting list (the class k_tie_long / tie_score_block serves).
constexpr uint32_t kCandLong = 0x80000000u;

__global__ void __launch_bounds__(256) k_collect_cand(const int32_t *count, int M, Status *st,
                                                      const uint32_t *post_off, uint32_t *cand)
{
    if (st->stop) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool hit = i < M && count[i] >= st->theta;
    const unsigned long long mh = __ballot(hit);
    if (mh) {
        const int first = __ffsll((long long)mh) - 1;
        unsigned base = 0;
        if (lane == first) base = atomicAdd(&st->n_cand, (unsigned)__popcll(mh));
        base = __shfl(base, first);
        if (hit)
            cand[base + (unsigned)__popcll(mh & ((1ull << lane) - 1ull))] =
                (uint32_t)i | (post_off[i + 1] - post_off[i] > kLongList ? kCandLong : 0u);
    }
}

// winner = highest score, then smallest word (= smallest id: ids follow the sorted key order).
// Scores are positive f32, so their bit patterns order like the values.
__device__ __forceinline__ unsigned long long winner_key(float score, uint32_t kid)
{
    return ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xffffffffu - kid);
}

// partition_tie_score (main.rs:261-283): walk the posting list in ascending segment order, skip
// covered segments, and on the first sight of each partition add 1 / (coverage + 1) in f32 (the
// order of the additions is the reference's).  One wave per tied word; `seen` is a per-wave bitmap
// in LDS.
__device__ __forceinline__ float tie_score_wave(uint32_t kid, unsigned *seen, int words, int lane,
                                                const uint32_t *post_off, const uint32_t *post,
                                                const uint8_t *ignored, const uint32_t *coverage, int P, int G)
{
    constexpr int kDeep = 8;   // 64-posting chunks whose (post -> ignored) loads are in flight together
    for (int wd = lane; wd < words; wd += 64) seen[wd] = 0u;
    const uint32_t b = post_off[kid], e = post_off[kid + 1];
    float acc = 0.0f;
    for (uint32_t base = b; base < e; base += 64 * kDeep) {
        uint32_t seg[kDeep];
        bool live[kDeep];
#pragma unroll
        for (int u = 0; u < kDeep; ++u) {
            const uint32_t i = base + u * 64 + lane;
            seg[u] = i < e ? post[i] : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < kDeep; ++u)
            live[u] = seg[u] != 0xffffffffu && !ignored[row_of(seg[u], (uint32_t)P, (uint32_t)G)];
#pragma unroll
        for (int u = 0; u < kDeep; ++u) {
            const int part = live[u] ? (int)(seg[u] % (uint32_t)P) : -1;
            const int ps = live[u] ? part : 0;
            unsigned long long m = __ballot(live[u] && !((seen[ps >> 5] >> (ps & 31)) & 1u));
            while (m) {   // wave-uniform: distinct new partitions in ascending posting order
                const int l = __ffsll((long long)m) - 1;
                const int pl = __shfl(part, l);
                m &= ~__ballot(part == pl);
                if (lane == 0) seen[pl >> 5] |= 1u << (pl & 31);
                acc += 1.0f / ((float)coverage[pl] + 1.0f);
            }
        }
    }
    return acc;
}

__global__ void __launch_bounds__(256) k_tie_scores(const uint32_t *tied, Status *st,
                                                    const uint32_t *post_off, const uint32_t *post,
                                                    const uint8_t *ignored, const uint32_t *coverage,
                                                    int P, int G)
{
    extern __shared__ unsigned char smem[];
    if (st->stop) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int words = (P + 31) / 32;
    unsigned *seen = (unsigned *)smem + (size_t)wave * words;
    const unsigned n = st->n_tied;
    if (n + st->n_long == 1) {   // a single candidate wins whatever its score
        if (n == 1 && blockIdx.x == 0 && threadIdx.x == 0) st->best = winner_key(1.0f, tied[0]);
        return;
    }
    unsigned long long best = 0;
    for (unsigned tix = blockIdx.x * 4 + wave; tix < n; tix += gridDim.x * 4) {
        const uint32_t kid = tied[tix];
        const float acc = tie_score_wave(kid, seen, words, lane, post_off, post, ignored, coverage, P, G);
        const unsigned long long key = winner_key(acc, kid);
        best = key > best ? key : best;
    }
    if (lane == 0 && best) atomicMax(&st->best, best);
}

// The same walk for a long posting list, by a whole 1024-thread block: the block loads and
// filters 1024 postings at a time.  A chunk that holds not-yet-seen partitions (normally just the first
// chunk, and one partition) gives them up one at a time, earliest posting first: the block finds the
// smallest thread index among the fresh postings and that thread adds its partition's term.  A chunk with
// many new partitions goes over to the ordered path after kExtract rounds, where the 16 waves resolve
// their candidates in turn.  Every thread of the block calls it; the score is in sh->acc afterwards.
struct TieBlockShared {
    float acc;
    int first[3];   // smallest fresh thread index; round r uses first[r % 3]
    int part;       // the partition being taken
};

__device__ __forceinline__ void tie_score_block(uint32_t kid, unsigned *seen, TieBlockShared *sh, int words,
                                                const uint32_t *post_off, const uint32_t *post,
                                                const uint8_t *ignored, const uint32_t *coverage, int P, int G)
{
    constexpr int kExtract = 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int wd = threadIdx.x; wd < words; wd += 1024) seen[wd] = 0u;
    if (threadIdx.x == 0) {
        sh->acc = 0.0f;
        sh->first[0] = 0x7fffffff;
        sh->first[1] = 0x7fffffff;
        sh->first[2] = 0x7fffffff;
    }
    const uint32_t b = post_off[kid], e = post_off[kid + 1];
    __syncthreads();
    int slot = 0;
    for (uint32_t base = b; base < e; base += 1024 * kTieUnroll) {
        // kTieUnroll chunks of (post -> ignored) dependent loads in flight at once
        uint32_t seg[kTieUnroll];
        bool live[kTieUnroll];
#pragma unroll
        for (int u = 0; u < kTieUnroll; ++u) {
            const uint32_t i = base + u * 1024 + threadIdx.x;
            seg[u] = i < e ? post[i] : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < kTieUnroll; ++u)
            live[u] = seg[u] != 0xffffffffu && !ignored[row_of(seg[u], (uint32_t)P, (uint32_t)G)];
#pragma unroll
        for (int u = 0; u < kTieUnroll; ++u) {
            const int part = live[u] ? (int)(seg[u] % (uint32_t)P) : -1;
            const int ps = live[u] ? part : 0;   // in-bounds bitmap index for idle lanes
            bool fresh = live[u] && !((seen[ps >> 5] >> (ps & 31)) & 1u);
            bool more = true;
            for (int round = 0; round < kExtract; ++round) {
                // first[slot] is 0x7fffffff here and nobody reads it any more: it was reset two rounds ago,
                // after the barrier that followed its last read
                const unsigned long long mf = __ballot(fresh);
                if (mf && lane == __ffsll((long long)mf) - 1) atomicMin(&sh->first[slot], (int)threadIdx.x);
                __syncthreads();
                const int f = sh->first[slot];
                const int prev = slot == 0 ? 2 : slot - 1;
                if (threadIdx.x == 0) sh->first[prev] = 0x7fffffff;   // its readers are all past the barrier
                slot = slot == 2 ? 0 : slot + 1;
                if (f == 0x7fffffff) {   // block-uniform: nothing fresh is left in this chunk
                    more = false;
                    break;
                }
                if ((int)threadIdx.x == f) {
                    sh->part = part;
                    seen[part >> 5] |= 1u << (part & 31);
                    sh->acc += 1.0f / ((float)coverage[part] + 1.0f);
                }
                __syncthreads();
                if (part == sh->part) fresh = false;
            }
            if (more && __syncthreads_or(fresh)) {
                for (int w = 0; w < 16; ++w) {
                    if (wave == w) {
                        // partitions taken by the earlier waves of this chunk are visible now
                        if (fresh) fresh = !((seen[ps >> 5] >> (ps & 31)) & 1u);
                        unsigned long long m = __ballot(fresh);
                        float acc = sh->acc;
                        while (m) {
                            const int l = __ffsll((long long)m) - 1;
                            const int pl = __shfl(part, l);
                            m &= ~__ballot(part == pl);
                            if (lane == 0) seen[pl >> 5] |= 1u << (pl & 31);
                            acc += 1.0f / ((float)coverage[pl] + 1.0f);
                        }
                        if (lane == 0) sh->acc = acc;
                    }
                    __syncthreads();
                }
            }
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) k_tie_long(const uint32_t *tied, int M, Status *st,
                                                   const uint32_t *post_off, const uint32_t *post,
                                                   const uint8_t *ignored, const uint32_t *coverage,
                                                   int P, int G)
{
    extern __shared__ unsigned char smem[];
    __shared__ TieBlockShared tb;
    if (st->stop) return;
    unsigned *seen = (unsigned *)smem;
    const int words = (P + 31) / 32;
    const unsigned n = st->n_long;
    if (n + st->n_tied == 1) {   // a single candidate wins whatever its score
        if (n == 1 && blockIdx.x == 0 && threadIdx.x == 0)
            st->best = winner_key(1.0f, tied[(unsigned)M - 1u]);
        return;
    }
    for (unsigned j = blockIdx.x; j < n; j += gridDim.x) {
        const uint32_t kid = tied[(unsigned)M - 1u - j];
        tie_score_block(kid, seen, &tb, words, post_off, post, ignored, coverage, P, G);
        if (threadIdx.x == 0) atomicMax(&st->best, winner_key(tb.acc, kid));
        __syncthreads();
    }
}

// ---- candidate-list loop -------------------------------------------------------------------------------
// One iteration = k_select -> k_cover<true>.  A batch starts with k_max_count<true> + k_collect_cand, which
// gather the words whose live count is at least theta = max / 2 (a few thousand, against millions of words).
// Counts only fall, so while the candidates' maximum stays >= theta it is the global maximum and every word
// tied with it is a candidate: k_select finds the maximum and the tied words by reading the candidates only,
// every block for itself (no grid-wide step), scores its share of them and posts winner_key() maxima.
// When the maximum falls below theta the rest of the batch is a no-op and the next batch starts from a
// fresh list.  The decisions are those of the five-launch iteration: same maximum, same tied set, same
// scores, same winner.
constexpr unsigned kCandCap = 32768;   // longer lists go the five-launch way: every block reads all of it
constexpr int kSelectGrid = 128;
constexpr int kOwn = (int)(kCandCap / kSelectGrid);   // a block's share of the tied words fits: no second pass
constexpr int kNarrowMaxP = 8192;      // 17 partition bitmaps in LDS

__global__ void __launch_bounds__(1024) k_select(Status *st, const int32_t *count, const uint32_t *cand,
                                                 const uint32_t *post_off, const uint32_t *post,
                                                 const uint8_t *ignored, const uint32_t *coverage, int P, int G,
                                                 int parity)
{
    constexpr int kKeep = 4;   // rounds of 1024 candidates whose (word, count) stay in registers
    extern __shared__ unsigned char smem[];
    __shared__ int red[16];
    __shared__ unsigned wl[kKeep * 16], ws[kKeep * 16];   // tied words per (round, wave), then their prefix sums
    __shared__ unsigned tot_sh[2];
    __shared__ uint32_t own_long[kOwn], own_short[kOwn];
    __shared__ TieBlockShared tb;
    // stop / need_rebuild are written by block 0 of this very kernel as well: every block reaches the same
    // verdict from the same inputs, so a block that sees the flag early only skips work it would have skipped
    if (st->stop || st->need_rebuild) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int words = (P + 31) / 32;
    unsigned *seen_blk = (unsigned *)smem;
    unsigned *seen_w = seen_blk + (size_t)words * (1 + wave);
    const int n_win = st->n_win;
    if (n_win >= st->max_iter || st->stop_next) {   // main.rs:344-366
        if (blockIdx.x == 0 && tid == 0) st->stop = 1;
        return;
    }
    const unsigned n_cand = st->n_cand;
    if (n_cand > kCandCap) {   // every block reads the whole list: too long for that, the host goes the other way
        if (blockIdx.x == 0 && tid == 0) {
            st->need_rebuild = 1;
            st->too_many = 1;
        }
        return;
    }
    // the first kKeep rounds of the list stay in registers for the second look
    uint32_t ck[kKeep];
    int cc[kKeep];
#pragma unroll
    for (int r = 0; r < kKeep; ++r) {
        const unsigned i = (unsigned)r * 1024u + (unsigned)tid;
        ck[r] = i < n_cand ? cand[i] : 0xffffffffu;
    }
#pragma unroll
    for (int r = 0; r < kKeep; ++r) cc[r] = ck[r] != 0xffffffffu ? count[ck[r] & ~kCandLong] : 0;
    int m = 0;
#pragma unroll
    for (int r = 0; r < kKeep; ++r) m = max(m, cc[r]);
    for (unsigned i = kKeep * 1024u + (unsigned)tid; i < n_cand; i += 1024) m = max(m, count[cand[i] & ~kCandLong]);
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = max(m, red[w]);
    if (m < st->theta) {   // a word outside the list may be ahead now
        if (blockIdx.x == 0 && tid == 0) st->need_rebuild = 1;
        return;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    // number the tied words (long and short posting lists apart) in list order; word t of a class belongs
    // to block t % gridDim.x, which keeps its share in own_long / own_short
    unsigned long long bl[kKeep], bs[kKeep];
#pragma unroll
    for (int r = 0; r < kKeep; ++r) {
        const bool hit = cc[r] == m;   // m >= 2: the padding (count 0) never hits
        const bool lg = hit && (ck[r] & kCandLong);
        bl[r] = __ballot(lg);
        bs[r] = __ballot(hit && !lg);
        if (lane == 0) {
            wl[r * 16 + wave] = (unsigned)__popcll(bl[r]);
            ws[r * 16 + wave] = (unsigned)__popcll(bs[r]);
        }
    }
    __syncthreads();
    if (wave == 0) {   // exclusive prefix sums over the kKeep * 16 (round, wave) cells, in list order
        unsigned a = wl[lane], c = ws[lane];
        unsigned ia = a, ic = c;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned ta = __shfl_up(ia, off), tc = __shfl_up(ic, off);
            if (lane >= off) {
                ia += ta;
                ic += tc;
            }
        }
        wl[lane] = ia - a;
        ws[lane] = ic - c;
        if (lane == 63) {
            tot_sh[0] = ia;
            tot_sh[1] = ic;
        }
    }
    __syncthreads();
    const unsigned keep_l = tot_sh[0], keep_s = tot_sh[1];
    unsigned long long best = 0;
#pragma unroll
    for (int r = 0; r < kKeep; ++r) {
        const bool hit = cc[r] == m;
        if (hit) {
            const bool lg = (ck[r] & kCandLong) != 0;
            const unsigned t = lg ? wl[r * 16 + wave] + (unsigned)__popcll(bl[r] & below)
                                  : ws[r * 16 + wave] + (unsigned)__popcll(bs[r] & below);
            if (t % gridDim.x == blockIdx.x) (lg ? own_long : own_short)[t / gridDim.x] = ck[r] & ~kCandLong;
        }
    }
    unsigned tot_l = keep_l, tot_s = keep_s;
    for (unsigned base = kKeep * 1024u; base < n_cand; base += 1024) {   // the rest of a long list
        const unsigned i = base + (unsigned)tid;
        uint32_t kid = 0;
        bool hit = false, lg = false;
        if (i < n_cand) {
            kid = cand[i];
            hit = count[kid & ~kCandLong] == m;
            lg = hit && (kid & kCandLong);
        }
        const unsigned long long xl = __ballot(lg), xs = __ballot(hit && !lg);
        __syncthreads();   // red is read by everybody before it is rewritten
        if (lane == 0) red[wave] = (int)((unsigned)__popcll(xl) << 16 | (unsigned)__popcll(xs));
        __syncthreads();
        unsigned pl = tot_l, ps = tot_s;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned v = (unsigned)red[w];
            if (w < wave) {
                pl += v >> 16;
                ps += v & 0xffffu;
            }
            tot_l += v >> 16;
            tot_s += v & 0xffffu;
        }
        if (hit) {
            const unsigned t = lg ? pl + (unsigned)__popcll(xl & below) : ps + (unsigned)__popcll(xs & below);
            if (t % gridDim.x == blockIdx.x) (lg ? own_long : own_short)[t / gridDim.x] = kid & ~kCandLong;
        }
    }
    __syncthreads();   // the lists are complete
    // t / gridDim.x < n_cand / gridDim.x <= kOwn: the share always fits
    const unsigned nl = tot_l > blockIdx.x ? (tot_l - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const unsigned ns = tot_s > blockIdx.x ? (tot_s - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (tot_l + tot_s == 1) {   // a single candidate wins whatever its score
        if (blockIdx.x == 0 && tid == 0) best = winner_key(1.0f, tot_l ? own_long[0] : own_short[0]);
    } else {
        for (unsigned j = 0; j < nl; ++j) {
            const uint32_t kid = own_long[j];
            tie_score_block(kid, seen_blk, &tb, words, post_off, post, ignored, coverage, P, G);
            if (tid == 0) {
                const unsigned long long key = winner_key(tb.acc, kid);
                best = key > best ? key : best;
            }
            __syncthreads();
        }
        for (unsigned j = wave; j < ns; j += 16) {
            const uint32_t kid = own_short[j];
            const float acc = tie_score_wave(kid, seen_w, words, lane, post_off, post, ignored, coverage, P, G);
            const unsigned long long key = winner_key(acc, kid);
            best = key > best ? key : best;
        }
    }
    if (lane == 0 && best) atomicMax(&st->best2[parity], best);
    if (blockIdx.x == 0 && tid == 0) {
        st->maxf = m;
        st->it1 = (unsigned)n_win + 1u;
        st->best2[parity ^ 1] = 0;   // the next iteration's; k_cover of the last one is over
    }
}

// Cover every segment that holds the winner: bump the partition coverage once per distinct
// partition (main.rs:371-378, covered segments included) and, for segments covered now, take
// one off the live count of every word they hold.
// A block takes 64 postings at a time.  Wave 0 does the per-posting bookkeeping; the four waves
// then gather the 64 segments' word ids into an LDS tile (64 window positions per pass, 16 loads per
// thread in flight) and split the window positions between them with lane = posting: neighbouring
// postings are the same window of near-identical genomes, so equal targets are merged across the
// wave before the atomic (same-address atomics serialise in L2).  The last block records the
// winner.
// kNarrow: the candidate-list loop's variant.  The winner comes from best2[parity], the stamp from it1, and
// block 0 records the winner on its own (nothing else in this kernel reads the fields it writes; k_select of
// the next iteration does), so there is no last-block step.
template <bool kNarrow>
__global__ void __launch_bounds__(256) k_cover(Status *st, const uint32_t *post_off,
                                               const uint32_t *post, uint8_t *ignored,
                                               uint32_t *coverage, uint32_t *stamp, int P, int G,
                                               int per, const int32_t *kid_of_inst, int32_t *count,
                                               const uint64_t *ukeys, uint64_t *out_key,
                                               uint32_t *out_freq, int parity)
{
    __shared__ int32_t tile[64 * 65];
    __shared__ uint32_t rows_s[64];   // partition-major row of each posting, ~0u: nothing to do
    __shared__ int any_live;
    if (st->stop || (kNarrow && st->need_rebuild)) return;
    const uint32_t it1 = kNarrow ? st->it1 : (uint32_t)st->n_win + 1u;   // unique stamp of this iteration
    const unsigned long long best = kNarrow ? st->best2[parity] : st->best;
    const uint32_t kid = 0xffffffffu - (uint32_t)(best & 0xffffffffull);
    if (kNarrow && blockIdx.x == 0 && threadIdx.x == 0) {
        const int mf = st->maxf, nw = st->n_win;
        out_key[nw] = ukeys[kid];
        out_freq[nw] = (uint32_t)mf;
        st->winner = (int)kid;
        st->n_win = nw + 1;
        if (mf < st->min_freq) st->stop_next = 1;   // main.rs:387-390: stop after the push
        st->maxf = 0;   // k_max_count starts from zero
    }
    const uint32_t b = post_off[kid], e = post_off[kid + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = b + blockIdx.x * 64u; base < e; base += gridDim.x * 64u) {
        bool rep = false;          // wave 0: this lane speaks for its partition
        uint32_t old_stamp = it1;
        int part = -1;
        if (wave == 0) {
            const uint32_t i = base + lane;
            uint32_t row = 0;
            bool live = false;
            if (i < e) {
                const uint32_t seg = post[i];
                part = (int)(seg % (uint32_t)P);
                row = (uint32_t)part * (uint32_t)G + seg / (uint32_t)P;
                live = !ignored[row];
                ignored[row] = 1;   // a segment appears once per posting list: no race
            }
            rows_s[lane] = live ? row : 0xffffffffu;
            const unsigned long long lives = __ballot(live);
            if (lane == 0) any_live = lives != 0ull;
            unsigned long long m = __ballot(part >= 0);
            while (m) {   // once per distinct partition of the wave
                const int l = __ffsll((long long)m) - 1;
                const int pl = __builtin_amdgcn_readlane(part, l);
                m &= ~__ballot(part == pl);
                rep = rep || lane == l;
            }
            // all the partitions at once; the answer is looked at after the count updates are on their way
            if (rep) old_stamp = atomicExch(&stamp[part], it1);
        }
        __syncthreads();
        if (any_live) {
            for (int q0 = 0; q0 < per; q0 += 64) {
                int32_t v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {   // 16 loads per thread in flight before the LDS stores
                    const int r = 4 * j + wave, q = q0 + lane;
                    const uint32_t row_r = rows_s[r];
                    const bool ok = row_r != 0xffffffffu && q < per;
                    const int32_t x = kid_of_inst[ok ? (size_t)row_r * per + q : 0];
                    v[j] = ok ? x : -1;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) tile[(4 * j + wave) * 65 + lane] = v[j];
                __syncthreads();
                const int nq = min(64, per - q0);
                for (int qq = wave; qq < nq; qq += 4) {
                    const int32_t k2 = tile[lane * 65 + qq];
                    // merge the common words of this window position (up to three rounds: the
                    // consensus word and its most frequent variants), the rest go one by one
                    unsigned long long mk = __ballot(k2 >= 0);
                    for (int round = 0; round < 3 && mk; ++round) {
                        const int l = __ffsll((long long)mk) - 1;
                        const int32_t kl = __builtin_amdgcn_readlane(k2, l);
                        const unsigned long long same = __ballot(k2 == kl);
                        if (lane == l) atomicSub(&count[kl], (int)__popcll(same));
                        mk &= ~same;
                    }
                    if ((mk >> lane) & 1ull) atomicSub(&count[k2], 1);
                }
                __syncthreads();
            }
        }
        if (rep && old_stamp != it1) atomicAdd(&coverage[part], 1u);   // first posting of the partition this iteration
        __syncthreads();   // rows_s / any_live are rewritten by the next group
    }
    __syncthreads();
    if (!kNarrow && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&st->ticket, 1u) == gridDim.x - 1) {
            const int mf = st->maxf;
            out_key[st->n_win] = ukeys[kid];
            out_freq[st->n_win] = (uint32_t)mf;
            st->winner = (int)kid;
            st->n_win += 1;
            if (mf < st->min_freq) st->stop_next = 1;   // main.rs:387-390: stop after the push
            st->maxf = 0;
            st->best = 0;
            st->ticket = 0;
        }
    }
}

// Coverage of the final primer set (main.rs:518-594): a segment is covered when its head window
// holds one of the forward primers or its tail window holds the reverse complement of one of the
// reverse primers.  One wave per segment, lane = window position; the primers are sorted packed
// words (2 bits per base, base p in bits [2p, 2p+1]) searched by bisection.
__device__ __forceinline__ bool has_word(const uint64_t *set, int n, uint64_t w)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (set[mid] < w) lo = mid + 1;
        else hi = mid;
    }
    return lo < n && set[lo] == w;
}

__global__ void __launch_bounds__(256) k_segment_hits(const SeqView seqs, size_t seq_len, int n_seg, int P,
                                                      int seg_size, int stride, int W, int k,
                                                      const uint64_t *fwd, int n_fwd, const uint64_t *rev,
                                                      int n_rev, uint8_t *hit)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    const int per = W - k + 1;
    for (int seg = wave; seg < n_seg; seg += n_waves) {
        const size_t rec = (size_t)(seg / P), col0 = (size_t)(seg % P) * (size_t)stride;
        bool found = false;
        for (int p = lane; p < per; p += 64) {
            uint64_t wf = 0, wr = 0;
            bool okf = true, okr = true;
            const size_t head = col0 + p, tail = col0 + (size_t)(seg_size - W) + p;
            for (int q = 0; q < k; ++q) {
                const int cf = base2(seq_at(seqs, rec, head + q)), cr = base2(seq_at(seqs, rec, tail + (k - 1 - q)));
                okf &= cf >= 0;
                okr &= cr >= 0;
                wf |= (uint64_t)(cf & 3) << (2 * q);
                wr |= (uint64_t)((3 - cr) & 3) << (2 * q);   // reverse complement of the tail word
            }
            found |= okf && has_word(fwd, n_fwd, wf);
            found |= okr && has_word(rev, n_rev, wr);
        }
        const unsigned long long any = __ballot(found);
        if (lane == 0) hit[seg] = any != 0ull;
    }
}

uint64_t lex_to_packed(uint64_t lex, int k)
{
    uint64_t w = 0;
    for (int p = 0; p < k; ++p) w |= ((lex >> (2 * (k - 1 - p))) & 3ull) << (2 * p);
    return w;
}

}  // namespace

int KmerStage::ensure(int slot, size_t bytes, std::string &err)
{
    if (cap_[slot] >= bytes) return MSSPE_OK;
    if (buf_[slot]) (void)hipFree(buf_[slot]);
    buf_[slot] = nullptr;
    cap_[slot] = 0;
    const hipError_t e = hipMalloc(&buf_[slot], bytes ? bytes : 16);
    if (e != hipSuccess) {
        err = std::string("hipMalloc (stage A): ") + hipGetErrorString(e);
        return MSSPE_ERR_DEVICE;
    }
    cap_[slot] = bytes;
    return MSSPE_OK;
}

void KmerStage::release()
{
    for (int s = 0; s < 16; ++s) {
        if (buf_[s]) (void)hipFree(buf_[s]);
        buf_[s] = nullptr;
        cap_[s] = 0;
    }
}

// the greedy loop's graph objects, released on every way out of KmerStage::run
struct GraphGuard {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ~GraphGuard()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
    }
};

#define KM_TRY(expr)                                                        \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) {                                            \
            err = std::string(#expr) + ": " + hipGetErrorString(e__);       \
            return MSSPE_ERR_DEVICE;                                        \
        }                                                                   \
    } while (0)

hipError_t launch_pack_rows(const uint8_t *d_ascii, int n_rows, size_t row_len, uint64_t *d_packed, hipStream_t stream)
{
    const size_t words = (size_t)n_rows * SeqView::row_words(row_len);
    if (!words) return hipSuccess;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, d_ascii, n_rows,
                       row_len, d_packed);
    return hipGetLastError();
}

int KmerStage::run(const SeqView &d_seqs, int n_seq, size_t seq_len, const msspe_kmer_opt &opt,
                   int direction, uint64_t *words_out, uint32_t *freq_out, int capacity,
                   int *n_out, hipStream_t stream, std::string &err)
{
    *n_out = 0;
    const int k = opt.kmer_size, W = opt.search_window_size;
    if (opt.overlap_size < W) {   // main.rs:201-203 panics
        err = "Overlap windows size must be greater or equal than search windows size";
        return MSSPE_ERR_ARG;
    }
    if (k < 1 || k > 31 || W < k || opt.segment_size < W || opt.overlap_size < 1 ||
        (direction != 0 && direction != 1) || n_seq < 0 || opt.max_iterations < 0) {
        err = "stage A: unsupported options (need 1 <= k <= 31, k <= window <= segment, stride >= 1)";
        return k < 1 || k > 31 ? MSSPE_ERR_K : MSSPE_ERR_ARG;
    }
    const long P = seq_len < (size_t)opt.segment_size
                       ? 0
                       : (long)((seq_len - (size_t)opt.segment_size) / (size_t)opt.overlap_size) + 1;
    const long n_seg_l = P * n_seq;
    if (n_seg_l == 0 || opt.max_iterations == 0) return MSSPE_OK;
    if (P > 65536 || n_seg_l > 0x7fffffffL / (W - k + 1)) {
        err = "stage A: alignment too large for 32-bit instance indices";
        return MSSPE_ERR_ARG;
    }
    const int n_seg = (int)n_seg_l;
    const int per = W - k + 1;
    const size_t n_inst = (size_t)n_seg * per;
    const uint64_t sentinel = 1ull << (2 * k);
    int rc;
    // buffers: 0/1 keys, 2/3 vals, 4 head, 5 hscan, 6 kid_of_inst, 7 post, 8 post_off, 9 ukeys,
    // 10 count+tied, 11 ignored, 12 coverage+stamp, 13 status/out, 14 cub temp
    if ((rc = ensure(0, n_inst * 8, err)) || (rc = ensure(1, n_inst * 8, err)) ||
        (rc = ensure(2, n_inst * 4, err)) || (rc = ensure(3, n_inst * 4, err)) ||
        (rc = ensure(4, n_inst * 4, err)) || (rc = ensure(5, n_inst * 4, err)) ||
        (rc = ensure(6, n_inst * 4, err)) || (rc = ensure(7, n_inst * 4, err)) ||
        (rc = ensure(8, (n_inst + 1) * 4, err)) || (rc = ensure(9, n_inst * 8, err)) ||
        (rc = ensure(10, n_inst * 8, err)) || (rc = ensure(11, (size_t)n_seg, err)) ||
        (rc = ensure(12, (size_t)P * 8, err)) ||
        (rc = ensure(13, 128 + (size_t)opt.max_iterations * 12, err)))
        return rc;
    uint64_t *key_a = (uint64_t *)buf_[0], *key_b = (uint64_t *)buf_[1];
    uint32_t *val_a = (uint32_t *)buf_[2], *val_b = (uint32_t *)buf_[3];
    uint32_t *head = (uint32_t *)buf_[4], *hscan = (uint32_t *)buf_[5];
    int32_t *kid_of_inst = (int32_t *)buf_[6];
    uint32_t *post = (uint32_t *)buf_[7], *post_off = (uint32_t *)buf_[8];
    uint64_t *ukeys = (uint64_t *)buf_[9];
    int32_t *count = (int32_t *)buf_[10];
    uint32_t *tied = (uint32_t *)buf_[10] + n_inst;
    uint8_t *ignored = (uint8_t *)buf_[11];
    uint32_t *coverage = (uint32_t *)buf_[12], *stamp = coverage + P;
    Status *st = (Status *)buf_[13];
    uint64_t *out_key = (uint64_t *)((char *)buf_[13] + 128);
    uint32_t *out_freq = (uint32_t *)(out_key + opt.max_iterations);

    // 1. extraction, 2. inverted index (the key type by the word length)
    const int wins_per_block = std::max(1, 256 / per);
    const int ext_grid = (n_seg + wins_per_block - 1) / wins_per_block;
    const size_t ext_lds = sizeof(uint64_t) * wins_per_block * per + (size_t)wins_per_block * W;
    if (per > 256) {
        err = "stage A: search window too wide for the extraction kernel";
        return MSSPE_ERR_ARG;
    }
    int M = 0;
    auto build_index = [&](auto key_tag) -> int {
        using Key = decltype(key_tag);
        Key *ka = (Key *)key_a, *kb = (Key *)key_b;
        hipLaunchKernelGGL(k_extract<Key>, dim3(ext_grid), dim3(256), ext_lds, stream, d_seqs, seq_len, n_seg,
                           (int)P, opt.segment_size, opt.overlap_size, W, k, direction, per,
                           wins_per_block, ka, val_a);
        KM_TRY(hipGetLastError());
        // stable radix sort on the 2k+1 key bits keeps ascending segment order
        size_t tmp_bytes = 0, tmp2 = 0;
        KM_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ka, kb, val_a, val_b, n_inst, 0u,
                                         (unsigned)(2 * k + 1), stream));
        KM_TRY(rocprim::exclusive_scan(nullptr, tmp2, head, hscan, 0u, n_inst, rocprim::plus<uint32_t>(), stream));
        int rc2;
        if ((rc2 = ensure(14, std::max(tmp_bytes, tmp2), err))) return rc2;
        KM_TRY(rocprim::radix_sort_pairs(buf_[14], tmp_bytes, ka, kb, val_a, val_b, n_inst, 0u,
                                         (unsigned)(2 * k + 1), stream));
        const int g_inst = (int)((n_inst + 255) / 256);
        hipLaunchKernelGGL(k_heads<Key>, dim3(g_inst), dim3(256), 0, stream, kb, n_inst, sentinel, head);
        KM_TRY(rocprim::exclusive_scan(buf_[14], tmp2, head, hscan, 0u, n_inst, rocprim::plus<uint32_t>(), stream));
        uint32_t last_head = 0, last_scan = 0;
        KM_TRY(hipMemcpyAsync(&last_head, head + n_inst - 1, 4, hipMemcpyDeviceToHost, stream));
        KM_TRY(hipMemcpyAsync(&last_scan, hscan + n_inst - 1, 4, hipMemcpyDeviceToHost, stream));
        KM_TRY(hipStreamSynchronize(stream));
        M = (int)(last_head + last_scan);
        if (M == 0) return MSSPE_OK;
        // number of valid instances = first sentinel position: post_off[M]
        hipLaunchKernelGGL(k_index<Key>, dim3(g_inst), dim3(256), 0, stream, kb, val_b, head, hscan, n_inst,
                           sentinel, per, (int)P, n_seq, kid_of_inst, post, post_off, ukeys);
        // post_off[M] = number of non-sentinel instances (sentinels sort last)
        hipLaunchKernelGGL(k_tail<Key>, dim3(1), dim3(1), 0, stream, kb, n_inst, sentinel, post_off, M);
        return MSSPE_OK;
    };
    if ((rc = 2 * k + 1 <= 32 ? build_index(uint32_t{}) : build_index(uint64_t{}))) return rc;
    if (M == 0) return MSSPE_OK;
    hipLaunchKernelGGL(k_init_counts, dim3((M + 255) / 256), dim3(256), 0, stream, post_off, M, count);
    KM_TRY(hipMemsetAsync(ignored, 0, (size_t)n_seg, stream));
    KM_TRY(hipMemsetAsync(coverage, 0, (size_t)P * 8, stream));
    KM_TRY(hipGetLastError());

    // 3. greedy loop: one iteration = five launches with constant arguments, captured once into a
    //    hipGraph (kBatch iterations per graph) and replayed; the loop state (Status) lives on the
    //    device and the host only looks at the stop flag between graphs.
    const int red_grid = std::min(128, (M + 255) / 256);
    const size_t tie_lds = 4 * sizeof(unsigned) * (size_t)((P + 31) / 32);
    Status h0;
    std::memset(&h0, 0, sizeof h0);
    h0.winner = -1;
    h0.max_iter = std::min(opt.max_iterations, capacity);
    h0.min_freq = opt.max_mismatch_segments;
    KM_TRY(hipMemcpyAsync(st, &h0, sizeof h0, hipMemcpyHostToDevice, stream));
    KM_TRY(hipStreamSynchronize(stream));
    auto enqueue_iteration = [&](hipStream_t s_, int /*node*/) {
        hipLaunchKernelGGL(k_max_count<false>, dim3(red_grid), dim3(256), 0, s_, count, M, st);
        hipLaunchKernelGGL(k_collect_tied, dim3((M + 255) / 256), dim3(256), 0, s_, count, M, st,
                           post_off, tied);
        hipLaunchKernelGGL(k_tie_scores, dim3(256), dim3(256), tie_lds, s_, tied, st, post_off, post,
                           ignored, coverage, (int)P, n_seq);
        hipLaunchKernelGGL(k_tie_long, dim3(64), dim3(1024), tie_lds / 4, s_, tied, M, st, post_off,
                           post, ignored, coverage, (int)P, n_seq);
        hipLaunchKernelGGL(k_cover<false>, dim3(256), dim3(256), 0, s_, st, post_off, post, ignored, coverage,
                           stamp, (int)P, n_seq, per, kid_of_inst, count, ukeys, out_key, out_freq, 0);
    };
    // the candidate-list iteration (two launches); `tied` doubles as the candidate list
    uint32_t *cand = tied;
    const size_t sel_lds = 17 * sizeof(unsigned) * (size_t)((P + 31) / 32);
    auto enqueue_narrow = [&](hipStream_t s_, int node) {
        if (node == 0) {
            // head of the batch: the maximum, theta and the candidate list (and the stop decision)
            hipLaunchKernelGGL(k_max_count<true>, dim3(red_grid), dim3(256), 0, s_, count, M, st);
            hipLaunchKernelGGL(k_collect_cand, dim3((M + 255) / 256), dim3(256), 0, s_, count, M, st, post_off,
                               cand);
        }
        hipLaunchKernelGGL(k_select, dim3(kSelectGrid), dim3(1024), sel_lds, s_, st, count, cand, post_off, post,
                           ignored, coverage, (int)P, n_seq, node & 1);
        hipLaunchKernelGGL(k_cover<true>, dim3(256), dim3(256), 0, s_, st, post_off, post, ignored, coverage,
                           stamp, (int)P, n_seq, per, kid_of_inst, count, ukeys, out_key, out_freq, node & 1);
    };
    // a batch of kBatch iterations is ONE graph (fewer graph launches than one graph per iteration:
    // 29 -> 27 ms per direction at 10,000 genomes)
    constexpr int kBatch = 32;
    GraphGuard gg[2];   // 0: five-launch iterations, 1: candidate-list iterations
    bool use_graph = use_graph_;   // option "stage_a_graph" (0: plain launches, a testing aid)
    auto capture = [&](int which) -> bool {
        // capture on a private stream so that the caller's stream may be of any kind
        hipStream_t cs = nullptr;
        bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
        if (ok && hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            for (int b = 0; b < kBatch; ++b) which ? enqueue_narrow(cs, b) : enqueue_iteration(cs, b);
            ok = hipStreamEndCapture(cs, &gg[which].graph) == hipSuccess && gg[which].graph &&
                 hipGraphInstantiate(&gg[which].exec, gg[which].graph, nullptr, nullptr, 0) == hipSuccess;
        } else {
            ok = false;
        }
        if (cs) (void)hipStreamDestroy(cs);
        (void)hipGetLastError();
        return ok;
    };
    const bool narrow_ok = narrow_loop_ && P <= kNarrowMaxP;
    Status h = h0;
    int wide_left = 0;   // five-launch batches still to run after a candidate list came out too long
    for (int batch = 0; !h.stop; ++batch) {
        if (narrow_ok ? batch >= h0.max_iter + 2 : batch * kBatch >= h0.max_iter + 1) break;
        const int which = narrow_ok && wide_left == 0 ? 1 : 0;
        if (wide_left) --wide_left;
        if (use_graph && !gg[which].exec && !capture(which)) use_graph = false;
        if (use_graph) {
            KM_TRY(hipGraphLaunch(gg[which].exec, stream));
        } else {
            for (int b = 0; b < kBatch; ++b) which ? enqueue_narrow(stream, b) : enqueue_iteration(stream, b);
        }
        KM_TRY(hipGetLastError());
        KM_TRY(hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, stream));
        KM_TRY(hipStreamSynchronize(stream));
        if (which && h.too_many) wide_left = 4;
    }
    const int n_win = h.n_win;
    if (narrow_ok && !h.stop_next && n_win >= capacity && n_win < opt.max_iterations) {
        // the candidate-list loop stops at max_iter without looking at the words: is anything left?
        KM_TRY(hipMemsetAsync((char *)st + offsetof(Status, stop), 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_max_count<true>, dim3(red_grid), dim3(256), 0, stream, count, M, st);
        KM_TRY(hipGetLastError());
        KM_TRY(hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, stream));
        KM_TRY(hipStreamSynchronize(stream));
    }
    if (h.last_max > 1 && !h.stop_next && n_win >= capacity && n_win < opt.max_iterations) {
        // the caller's buffers ended the loop, not the reference's rules
        err = "stage A: more winners than the caller's capacity";
        return MSSPE_ERR_CAPACITY;
    }
    if (n_win) {
        std::vector<uint64_t> hk((size_t)n_win);
        KM_TRY(hipMemcpyAsync(hk.data(), out_key, sizeof(uint64_t) * n_win, hipMemcpyDeviceToHost, stream));
        KM_TRY(hipMemcpyAsync(freq_out, out_freq, sizeof(uint32_t) * n_win, hipMemcpyDeviceToHost, stream));
        KM_TRY(hipStreamSynchronize(stream));
        for (int i = 0; i < n_win; ++i) words_out[i] = lex_to_packed(hk[i], k);
    }
    *n_out = n_win;
    return MSSPE_OK;
}

int KmerStage::coverage(const SeqView &d_seqs, int n_seq, size_t seq_len, const msspe_kmer_opt &opt,
                        const uint64_t *fwd_words, int n_fwd, const uint64_t *rev_words, int n_rev,
                        uint8_t *hit_out, hipStream_t stream, std::string &err)
{
    const int k = opt.kmer_size, W = opt.search_window_size;
    if (k < 1 || k > 31 || W < k || opt.segment_size < W || opt.overlap_size < 1 || n_seq < 0 || n_fwd < 0 ||
        n_rev < 0) {
        err = "coverage: unsupported options (need 1 <= k <= 31, k <= window <= segment, stride >= 1)";
        return k < 1 || k > 31 ? MSSPE_ERR_K : MSSPE_ERR_ARG;
    }
    const long P = seq_len < (size_t)opt.segment_size
                       ? 0
                       : (long)((seq_len - (size_t)opt.segment_size) / (size_t)opt.overlap_size) + 1;
    const long n_seg = P * n_seq;
    if (n_seg == 0) return MSSPE_OK;
    if (n_seg > 0x7fffffffL) {
        err = "coverage: alignment too large for 32-bit segment indices";
        return MSSPE_ERR_ARG;
    }
    std::vector<uint64_t> f(fwd_words, fwd_words + n_fwd), r(rev_words, rev_words + n_rev);
    std::sort(f.begin(), f.end());
    std::sort(r.begin(), r.end());
    int rc;
    if ((rc = ensure(15, sizeof(uint64_t) * (size_t)(n_fwd + n_rev + 2) + (size_t)n_seg, err))) return rc;
    uint64_t *d_f = (uint64_t *)buf_[15], *d_r = d_f + n_fwd + 1;
    uint8_t *d_hit = (uint8_t *)(d_r + n_rev + 1);
    if (n_fwd) KM_TRY(hipMemcpyAsync(d_f, f.data(), sizeof(uint64_t) * n_fwd, hipMemcpyHostToDevice, stream));
    if (n_rev) KM_TRY(hipMemcpyAsync(d_r, r.data(), sizeof(uint64_t) * n_rev, hipMemcpyHostToDevice, stream));
    const int grid = (int)std::min<long>(4096, (n_seg + 3) / 4);
    hipLaunchKernelGGL(k_segment_hits, dim3(grid), dim3(256), 0, stream, d_seqs, seq_len, (int)n_seg, (int)P,
                       opt.segment_size, opt.overlap_size, W, k, d_f, n_fwd, d_r, n_rev, d_hit);
    KM_TRY(hipGetLastError());
    KM_TRY(hipMemcpyAsync(hit_out, d_hit, (size_t)n_seg, hipMemcpyDeviceToHost, stream));
    KM_TRY(hipStreamSynchronize(stream));   // the sorted host copies must outlive the uploads
    return MSSPE_OK;
}

}  // namespace msspe
