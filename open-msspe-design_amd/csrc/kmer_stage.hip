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
//      covers its segments and decrements the live counts of every word they hold -- here a whole run of
//      such picks per iteration, exactly (k_fast / k_score + k_prefix below).
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

#include <rocprim/detail/various.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
// rocPRIM's radix sort copies its input to scratch when the pass count is odd and "the input may alias the output",
// which it assumes of every iterator that is not a pointer (360 MB, 80 us at 45 M instances).  The values here come
// from a counting iterator, which aliases nothing: say so, ahead of the sort's definition (its call is qualified).
BEGIN_ROCPRIM_NAMESPACE
namespace detail
{
template <class V>
inline bool can_iterators_alias(::rocprim::counting_iterator<V>, V *, const size_t)
{
    return false;
}
}  // namespace detail
END_ROCPRIM_NAMESPACE
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
    // candidate-list loop (k_score / k_prefix / k_cover_multi): every word whose live count is >= theta is in cand[]
    int theta;
    unsigned n_cand;
    int need_rebuild;  // 1: a new candidate list is wanted (the iteration's remaining kernels are no-ops); 2: rebuilt in this iteration
    int last_max;      // the maximum the last k_max_count saw (the host's capacity check)
    int too_many;      // 1: the candidate list is longer than k_score reads, or too many words tie: the host runs five-launch batches
    // statistics of the candidate-list loop: iterations that recorded winners through k_fast / through k_score + k_prefix,
    // iterations that rebuilt the list, and iterations that found the loop over or waiting for a rebuild
    int it_fast, it_general, it_rebuild, it_idle;
    unsigned n_mcand;  // candidates with postings in several partitions (the list behind cand[])
    unsigned epoch;    // number of the candidate list (cand_flag[word] == epoch: the word is on it)
    int want_general;  // k_fast could not settle the iteration: the host runs k_score / k_prefix / k_cover_multi once
};
static_assert(sizeof(Status) <= 256, "the winners' arrays start 256 bytes into the status buffer");

// Segment ids are genome-major (seg = genome * P + partition: the order the reference walks
// them in), but a winner's postings are mostly one partition of many genomes.  The per-segment
// tables the cover step gathers from (word ids, covered flags) are therefore stored
// partition-major, so that those rows are neighbours in memory (same pages, same cache lines).
__device__ __forceinline__ uint32_t row_of(uint32_t seg, uint32_t P, uint32_t G)
{
    return (seg % P) * G + seg / P;
}

// The word id of window position q of the segment in partition-major row `row`: kid_of_inst is laid out
// [partition][window position][genome], so that the 64 genomes a wave handles at one window position are 64
// neighbouring words -- for the index pass that writes them (a word's postings are the same window position of many
// genomes, in ascending genome order) and for the cover and marking passes that read them with lane = genome.
__device__ __forceinline__ size_t inst_slot(uint32_t row, uint32_t q, uint32_t per, uint32_t G)
{
    const uint32_t part = row / G, genome = row - part * G;
    return ((size_t)part * per + q) * G + genome;
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

// Extraction (main.rs:163-235): one wave per search window, lane = window position, from the PACKED alignment.
// A window's 2-bit bases are a run of bits in two or three 64-bit words of its row, which all the lanes of the wave
// read (one broadcast load each); lane p cuts its k bases out with two shifts -- x, base q of the k-mer in bits
// [2q, 2q+1] -- and its k validity bits the same way.  The emitted key is the word's lexicographic code (first
// base in the most significant bits): for the head window the 2-bit groups of x reversed (bit reverse + swap
// within pairs), for the tail window, whose k-mers are stored reverse-complemented (main.rs:213-224), simply the
// complement of x -- reversing the order and reading LSB-first instead of MSB-first cancel.
// First occurrence per window (itertools::unique, main.rs:168), without memory: the lanes that share an 8-bit
// hash of x are found with eight ballots (a lane's mask = the AND over the hash bits of the ballot or its
// complement); a lane then compares its x with the x of the earlier lanes of its mask only, fetched by shuffle --
// almost always none (38 positions in 256 buckets), so the loop runs once or twice per window.  Windows wider than
// 64 positions take several passes of 64; a later pass also compares with the earlier passes' words, kept in LDS.
// Only the keys are written, 4 (k <= 15) or 8 bytes per window position in one contiguous run per wave; the
// instance numbers the sort carries along come from a counting iterator.
template <class Key>
__global__ void __launch_bounds__(256) k_extract(const uint64_t *packed, size_t seq_len, int n_seg, int P, int seg_size,
                                                 int stride, int W, int k, int direction, int per, Key *key_out)
{
    __shared__ unsigned long long prev_x[4][256];   // the valid words of the window's earlier passes (per > 64 only)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
    const size_t bw = (seq_len + 31) / 32, vw = (seq_len + 63) / 64, rw = bw + vw;
    const unsigned long long xmask = (1ull << (2 * k)) - 1ull, vmask = (1ull << k) - 1ull;
    const unsigned long long below = (1ull << lane) - 1ull;
    const Key sentinel = (Key)(1ull << (2 * k));
    for (int seg = wave; seg < n_seg; seg += n_waves) {   // wave-uniform
        const int rec = seg / P, part = seg - rec * P;
        const uint64_t *row = packed + (size_t)rec * rw;
        const size_t c0 = (size_t)part * (size_t)stride + (size_t)(direction ? seg_size - W : 0);
        for (int p0 = 0; p0 < per; p0 += 64) {   // wave-uniform
            const int p = p0 + lane;
            const bool in = p < per;
            const size_t col = c0 + (size_t)(in ? p : 0);
            const size_t w0 = col >> 5, v0 = col >> 6;
            const int sh = (int)(col & 31) * 2, vs = (int)(col & 63);
            const unsigned long long lo = row[w0], hi = w0 + 1 < bw ? row[w0 + 1] : 0ull;
            const unsigned long long vlo = row[bw + v0], vhi = v0 + 1 < vw ? row[bw + v0 + 1] : 0ull;
            const unsigned long long x = ((lo >> sh) | (sh ? hi << (64 - sh) : 0ull)) & xmask;
            const unsigned long long vb = ((vlo >> vs) | (vs ? vhi << (64 - vs) : 0ull)) & vmask;
            const bool ok = in && vb == vmask;
            // lanes with my hash (valid ones only)
            const unsigned hsh = (unsigned)((x * 0x9e3779b97f4a7c15ull) >> 56);
            unsigned long long same = __ballot(ok);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const unsigned long long bm = __ballot((hsh >> bit) & 1u);
                same &= ((hsh >> bit) & 1u) ? bm : ~bm;
            }
            unsigned long long cand = ok ? (same & below) : 0ull;   // earlier lanes that may hold my word
            bool dup = false;
            while (__ballot(cand != 0ull)) {   // wave-uniform trip count: the longest chain of equal hashes (mostly 0)
                const int src = cand ? __ffsll((long long)cand) - 1 : lane;
                const unsigned xl = (unsigned)__shfl((int)(unsigned)x, src), xh = (unsigned)__shfl((int)(unsigned)(x >> 32), src);
                dup = dup || (cand && (((unsigned long long)xh << 32) | xl) == x);
                cand &= cand - 1ull;
            }
            if (p0 > 0 && ok && !dup)   // (wide windows) the earlier passes' words
                for (int q = 0; q < p0 && !dup; ++q) dup = prev_x[wv][q] == x;
            if (per > 64) {
                prev_x[wv][p & 255] = ok ? x : ~0ull;
                __builtin_amdgcn_wave_barrier();
            }
            unsigned long long word;
            if (direction) {
                word = ~x & xmask;
            } else {
                unsigned long long r = __brevll(x) >> (64 - 2 * k);
                word = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
            }
            if (in) key_out[(size_t)seg * per + p] = (ok && !dup) ? (Key)word : sentinel;
        }
    }
}

// Numbering the words of the sorted key array (a word's number = the first instances before its own): the first
// instances are counted per block of 256 keys (k_head_count), a small scan turns the 176 k counts into the blocks'
// bases, and k_index finishes the numbering inside its block with ballots.  (A scan over the 45 M flags, and the
// flags themselves, cost 0.37 ms more; reading them through a transform iterator 0.25.)
constexpr int kIndexBlock = 256;   // keys per index block: one WAVE's, four consecutive keys per lane (16- / 32-byte loads)

// The four keys (and, for k_index, instance numbers) of lane `lane` of index block `blk`, and the element before them.
template <class T>
__device__ __forceinline__ void load4(const T *p, size_t i0, size_t n, T (&v)[4], T &before)
{
    if (i0 + 4 <= n) {
        if constexpr (sizeof(T) == 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(p + i0);
            v[0] = (T)q.x, v[1] = (T)q.y, v[2] = (T)q.z, v[3] = (T)q.w;
        } else {
            const ulonglong2 q0 = *reinterpret_cast<const ulonglong2 *>(p + i0), q1 = *reinterpret_cast<const ulonglong2 *>(p + i0 + 2);
            v[0] = (T)q0.x, v[1] = (T)q0.y, v[2] = (T)q1.x, v[3] = (T)q1.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = i0 + j < n ? p[i0 + j] : (T)0;
    }
    before = i0 > 0 && i0 <= n ? p[i0 - 1] : (T)0;
}

// bit j: element j of the lane's four is the first instance of its word
template <class Key>
__device__ __forceinline__ unsigned first_instances4(const Key (&k)[4], Key before, size_t i0, size_t n, uint64_t sentinel)
{
    unsigned f = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t prev = j ? (uint64_t)k[j - 1] : (uint64_t)before;
        const bool first = i0 + j < n && (uint64_t)k[j] != sentinel && (i0 + j == 0 || prev != (uint64_t)k[j]);
        f |= first ? 1u << j : 0u;
    }
    return f;
}

template <class Key>
__global__ void __launch_bounds__(256) k_head_count(const Key *key, size_t n, uint64_t sentinel, uint32_t *block_heads, size_t n_iblk)
{
    const int lane = threadIdx.x & 63;
    const size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
    if (blk >= n_iblk) return;
    const size_t i0 = blk * kIndexBlock + 4 * (size_t)lane;
    Key k[4], before;
    load4(key, i0, n, k, before);
    const unsigned f = first_instances4(k, before, i0, n, sentinel);
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c += (unsigned)__popcll(__ballot((f >> j) & 1u));
    if (lane == 0) block_heads[blk] = c;
}

template <class Key>
__global__ void __launch_bounds__(256) k_index(const Key *key, const uint32_t *val,
                        const uint32_t *block_base, size_t n, size_t n_iblk, uint64_t sentinel, int per, int P, int G, int M,
                        int32_t *kid_of_inst, uint32_t *post, uint32_t *post_off, uint64_t *ukeys,
                        uint16_t *word_part, uint8_t *word_multi)
{
    const int lane = threadIdx.x & 63;
    const size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
    if (blk >= n_iblk) return;
    const size_t i0 = blk * kIndexBlock + 4 * (size_t)lane;
    Key k[4], kbefore;
    uint32_t inst[4], ibefore;
    load4(key, i0, n, k, kbefore);
    load4(val, i0, n, inst, ibefore);
    const unsigned f = first_instances4(k, kbefore, i0, n, sentinel);
    // first instances before this lane's elements: the blocks before, the lanes before
    uint32_t before = block_base[blk];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) before += (uint32_t)__popcll(__ballot((f >> j) & 1u) & below);
    uint32_t segs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t i = i0 + j;
        const uint32_t seg = inst[j] / (uint32_t)per, q = inst[j] % (uint32_t)per;
        segs[j] = seg;
        if (i >= n) continue;
        const size_t slot = ((size_t)(seg % (uint32_t)P) * per + q) * (uint32_t)G + seg / (uint32_t)P;   // inst_slot of its row
        const uint64_t kk = (uint64_t)k[j];
        const uint64_t kprev = j ? (uint64_t)k[j - 1] : (uint64_t)kbefore;
        // post_off[M] = number of non-sentinel instances (sentinels sort last)
        if (kk == sentinel) {
            kid_of_inst[slot] = -1;
            if (i == 0 || kprev != sentinel) post_off[M] = (uint32_t)i;
            continue;
        }
        if (i == n - 1) post_off[M] = (uint32_t)n;
        const bool first = (f >> j) & 1u;
        before += first ? 1u : 0u;      // (now: first instances up to and including this element)
        const uint32_t kid = before - 1;
        kid_of_inst[slot] = (int32_t)kid;
        if (first) {
            post_off[kid] = (uint32_t)i;
            ukeys[kid] = kk;
            word_part[kid] = (uint16_t)(seg % (uint32_t)P);   // the partition of the word's first posting ...
        } else {
            const uint32_t iprev = j ? inst[j - 1] : ibefore;
            if ((iprev / (uint32_t)per) % (uint32_t)P != seg % (uint32_t)P)
                word_multi[kid] = 1;   // ... and whether any two neighbouring postings differ in theirs (cleared by the host)
        }
    }
    // post[i] = the instance's segment (read up to post_off[M] only: the sentinel run's entries are never looked at)
    if (i0 + 4 <= n) {
        *reinterpret_cast<uint4 *>(post + i0) = make_uint4(segs[0], segs[1], segs[2], segs[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) post[i0 + j] = segs[j];
    }
}

// A run's byte fills in one launch (seven memsets were seven launches, 7 us each with their gaps): job j sets n[j]
// bytes at p[j] to v[j]; 16-byte stores between the aligned ends, bytes at the ends.
struct FillJobs {
    void *p[8];
    size_t n[8];
    unsigned char v[8];
    int count;
};
__global__ void __launch_bounds__(256) k_fill_jobs(FillJobs jobs)
{
    for (int j = 0; j < jobs.count; ++j) {
        unsigned char *p = (unsigned char *)jobs.p[j];
        const size_t n = jobs.n[j];
        const unsigned w = 0x01010101u * jobs.v[j];
        const uintptr_t a0 = ((uintptr_t)p + 15) & ~(uintptr_t)15, a1 = ((uintptr_t)p + n) & ~(uintptr_t)15;
        if (a1 <= a0) {   // shorter than one aligned chunk
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = jobs.v[j];
            continue;
        }
        uint4 *body = (uint4 *)a0;
        const size_t chunks = (a1 - a0) / 16;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (size_t)gridDim.x * 256)
            body[i] = make_uint4(w, w, w, w);
        if (blockIdx.x == 0 && threadIdx.x < 32) {
            const size_t head = a0 - (uintptr_t)p, tail = (uintptr_t)p + n - a1;
            if (threadIdx.x < 16 && threadIdx.x < head) p[threadIdx.x] = jobs.v[j];
            if (threadIdx.x >= 16 && threadIdx.x - 16 < tail) ((unsigned char *)a1)[threadIdx.x - 16] = jobs.v[j];
        }
    }
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

// kRebuild: head of the candidate-list loop's iteration.  It runs only when a new list was asked for
// (st->need_rebuild == 1, or `force`); its last block turns the request into "rebuilt in this iteration" (2), which
// k_collect_cand waits for and k_prefix takes back to 0.  (The flags are written by the last block only, i.e.
// after every block has passed this test: no block can see them change under it.)
template <bool kRebuild>
__global__ void __launch_bounds__(256) k_max_count(const int32_t *count, int M, Status *st, int force, int32_t *live_part,
                                                   int P)
{
    __shared__ int part[4];
    __shared__ int last_sh;
    if (threadIdx.x == 0) last_sh = 0;
    if (st->stop) return;
    if (kRebuild && !force && st->need_rebuild == 0) return;
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
                st->n_mcand = 0;
                st->epoch += 1;
                last_sh = 1;
                st->need_rebuild = 2;
                st->too_many = 0;
                st->maxf = 0;
            }
            st->n_tied = 0;
            st->n_long = 0;
            st->ticket = 0;
        }
    }
    if (kRebuild) {
        // a new list: live_part[p] is counted afresh by k_mark (segments of p that hold a word of the list)
        __syncthreads();
        if (last_sh && live_part)
            for (int p = threadIdx.x; p < P; p += 256) live_part[p] = 0;
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

// mcand: the candidates with postings in several partitions, once more (k_multi walks those; at most kMaxMulti are
// stored, the count goes on)
constexpr unsigned kMaxMulti = 512;

__global__ void __launch_bounds__(256) k_collect_cand(const int32_t *count, int M, Status *st,
                                                      const uint32_t *post_off, uint32_t *cand,
                                                      const uint8_t *word_multi, uint32_t *mcand, uint32_t *cand_flag)
{
    if (st->stop || st->need_rebuild != 2) return;   // only behind a k_max_count<true> that rebuilt
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
        if (hit) cand_flag[i] = st->epoch;
        if (hit && word_multi[i]) {   // rare: one atomic each
            const unsigned at = atomicAdd(&st->n_mcand, 1u);
            if (at < kMaxMulti) mcand[at] = (uint32_t)i;
        }
    }
}

// The first list of a run needs no marking pass: every segment is live, every segment counts (the bound that comes
// of it is loose, but the first leaders cover most of their partitions: it does not bind).
__global__ void __launch_bounds__(256) k_live_all(const Status *st, int32_t *live_part, int P, int G)
{
    if (st->stop || st->need_rebuild != 2) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) live_part[p] = G;
}

// Behind a new candidate list: which live segments hold a word of the list (marked), and how many of them each
// partition has (live_part).  A word of the list can only lose marked segments, and after a cover of f live segments
// of partition p every word living in p has at most live_part[p] - f left: k_fast's bound.  (Segments that hold rare
// words only -- most of what is left late in the loop -- do not count.)  A wave takes 64 consecutive rows, lane = row
// (rows are partition-major: mostly one partition, one atomic), and walks the window positions.
__global__ void __launch_bounds__(256) k_mark(const Status *st, const int32_t *kid_of_inst, const uint8_t *ignored,
                                              const uint32_t *cand_flag, int n_seg, int per, int G, uint8_t *marked,
                                              int32_t *live_part)
{
    if (st->stop || st->need_rebuild != 2) return;
    const uint32_t epoch = st->epoch;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (int r0 = wave * 64; r0 < n_seg; r0 += n_waves * 64) {   // wave-uniform
        const int row = r0 + lane;
        const bool in = row < n_seg;
        const int part = in ? row / G : -1;
        bool any = false;
        if (in && !ignored[row]) {
            const int32_t *col = kid_of_inst + ((size_t)part * per) * G + (row - part * G);
            for (int q = 0; q < per && !any; q += 4) {   // four loads in flight; a marked segment stops early
                int32_t kid[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) kid[j] = q + j < per ? col[(size_t)(q + j) * G] : -1;
#pragma unroll
                for (int j = 0; j < 4; ++j) any = any || (kid[j] >= 0 && cand_flag[kid[j]] == epoch);
            }
        }
        if (in) marked[row] = any ? 1 : 0;
        unsigned long long m = __ballot(any);
        while (m) {   // one atomic per distinct partition of the 64 rows (one, at a boundary two)
            const int l = __ffsll((long long)m) - 1;
            const int pl = __builtin_amdgcn_readlane(part, l);
            const unsigned long long same = __ballot(any && part == pl);
            if (lane == l) atomicAdd(&live_part[pl], (int)__popcll(same));
            m &= ~same;
        }
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
// allp (optional): a second bitmap that receives the partitions of ALL the postings, covered segments included
// (what main.rs:371-378 bumps the coverage of when the word wins).
__device__ __forceinline__ float tie_score_wave(uint32_t kid, unsigned *seen, unsigned *allp, int words, int lane,
                                                const uint32_t *post_off, const uint32_t *post,
                                                const uint8_t *ignored, const uint32_t *coverage, int P, int G)
{
    constexpr int kDeep = 8;   // 64-posting chunks whose (post -> ignored) loads are in flight together
    for (int wd = lane; wd < words; wd += 64) {
        seen[wd] = 0u;
        if (allp) allp[wd] = 0u;
    }
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
            const int pall = seg[u] != 0xffffffffu ? (int)(seg[u] % (uint32_t)P) : -1;
            const int part = live[u] ? pall : -1;
            const int ps = live[u] ? part : 0;
            if (allp) {   // wave-uniform
                const int pq = pall >= 0 ? pall : 0;
                unsigned long long ma = __ballot(pall >= 0 && !((allp[pq >> 5] >> (pq & 31)) & 1u));
                while (ma) {
                    const int l = __ffsll((long long)ma) - 1;
                    const int pl = __shfl(pall, l);
                    ma &= ~__ballot(pall == pl);
                    if (lane == 0) allp[pl >> 5] |= 1u << (pl & 31);
                }
            }
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
        const float acc = tie_score_wave(kid, seen, nullptr, words, lane, post_off, post, ignored, coverage, P, G);
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

__device__ __forceinline__ void tie_score_block(uint32_t kid, unsigned *seen, unsigned *allp, TieBlockShared *sh, int words,
                                                const uint32_t *post_off, const uint32_t *post,
                                                const uint8_t *ignored, const uint32_t *coverage, int P, int G)
{
    constexpr int kExtract = 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int wd = threadIdx.x; wd < words; wd += 1024) {
        seen[wd] = 0u;
        if (allp) allp[wd] = 0u;   // allp (optional): the partitions of ALL the postings, see tie_score_wave
    }
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
            const int pall = seg[u] != 0xffffffffu ? (int)(seg[u] % (uint32_t)P) : -1;
            const int part = live[u] ? pall : -1;
            const int ps = live[u] ? part : 0;   // in-bounds bitmap index for idle lanes
            if (allp) {   // block-uniform; a wave merges its lanes, the waves' bits meet in an LDS atomic
                const int pq = pall >= 0 ? pall : 0;
                unsigned long long ma = __ballot(pall >= 0 && !((allp[pq >> 5] >> (pq & 31)) & 1u));
                while (ma) {
                    const int l = __ffsll((long long)ma) - 1;
                    const int pl = __shfl(pall, l);
                    ma &= ~__ballot(pall == pl);
                    if (lane == 0) atomicOr(&allp[pl >> 5], 1u << (pl & 31));
                }
            }
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
        tie_score_block(kid, seen, nullptr, &tb, words, post_off, post, ignored, coverage, P, G);
        if (threadIdx.x == 0) atomicMax(&st->best, winner_key(tb.acc, kid));
        __syncthreads();
    }
}

// ---- candidate-list loop: several winners per iteration ---------------------------------------------------
// A batch of iterations starts with k_max_count<true> + k_collect_cand, which gather the words whose live count
// is at least theta = max / 2 (a few thousand, against millions of words).  Counts only fall, so while the
// candidates' maximum stays >= theta it is the global maximum and every word near it is a candidate.
//
// One iteration = k_score -> k_prefix -> k_cover_multi and selects a whole PREFIX of the greedy order, exactly:
// let w1, w2, ... be the words in the loop's order (count desc, partition_tie_score desc, word asc) right now.
// Covering w1 (main.rs:371-378) only LOWERS other words' keys -- it takes live segments away (count, score) and
// raises the coverage of the partitions its posting list touches (score) -- and leaves a word untouched whose
// live postings lie in none of those partitions.  So if w2 is untouched by w1 it is the winner of the next
// iteration with the frequency it has now, w3 is the one after that if untouched by w1 and w2, and so on; the
// first touched word ends the prefix, because its new place in the order is unknown.  With 119 partitions of
// near-equal words a prefix runs until it meets a partition for the second time: about 14 winners per
// iteration, 50 iterations where the one-winner loop needed 656.
//   k_score        every block reads the candidates, finds the maximum and the set E of words at the top of the
//                  order -- all words with count >= some c, about kETarget of them, never cutting through a run of
//                  equal counts -- and scores its share of E: tie score, bitmap of the partitions of its live
//                  postings, bitmap of the partitions of all its postings.
//   k_prefix       one block: ranks E, takes the longest prefix whose words are untouched by the ones before
//                  them, applies the reference's stop rules winner by winner (main.rs:344-366, :387-390), records
//                  the winners and bumps the partition coverage.
//   k_cover_multi  covers the segments of all the winners and takes the live ones off the counts of their words.
// When the maximum falls below theta, k_prefix asks for a new candidate list (need_rebuild = 1): the next
// iteration's k_max_count<true> + k_collect_cand, no-ops otherwise, rebuild it.
constexpr unsigned kCandCap = 32768;   // longer lists go the five-launch way: every block reads all of it
constexpr int kSelectGrid = 128;
constexpr int kEMax = 2048;            // words scored per iteration at most (more words tied at the maximum: five-launch way)
constexpr int kETarget = 128;          // ... and aimed at
constexpr int kOwn = kEMax / kSelectGrid;   // a block's share of E per class
constexpr int kMaxPick = 64;           // winners per iteration at most
constexpr int kCoverGrid = 512;        // k_cover_multi: two 1,024-thread blocks per CU, 16 chunks of 64 postings each at a time
constexpr int kHistBins = 256;         // counts from the maximum down that E may reach
constexpr int kNarrowMaxP = 8192;      // 34 partition bitmaps in LDS

struct PickState {
    int fast_done;           // k_fast dealt with this iteration: k_score / k_prefix have nothing to do
    int valid;               // k_score ran to its end in this iteration (k_prefix takes it back)
    int m;                   // the candidates' maximum count
    unsigned tot_l, tot_s;   // words of E with long / short posting lists: result slots [0, tot_l) and [kEMax - tot_s, kEMax)
    int want_rebuild;        // the maximum fell below theta: a word outside the list may be ahead
    int too_many;            // list longer than kCandCap, or more than kEMax words tied at the maximum
    unsigned n_pick;         // winners of this iteration (k_prefix)
    unsigned n_chunks;       // 64-posting chunks of all of them
    uint32_t kid[kMaxPick];
    uint32_t cum[kMaxPick + 1];   // chunks of the winners before winner j
};

// The iteration's fast path: no posting walks but for a handful of words.  Nearly every word's postings lie in ONE
// partition (the same window of many genomes); word_part / word_multi say which, and for which words that is not so.
// For one-partition words the loop's order needs no walk: within a partition all of them have the same tie score,
// 1 / (coverage + 1) (one live partition: main.rs:268-282), so a partition's best word -- its LEADER -- is the one
// with the highest count, smallest word first; and covering a word changes nothing but the words of the partitions
// its postings lie in and those partitions' coverage.  The greedy order is therefore a merge of per-partition
// sequences, and a whole run of it can be read off an ordered list of the leaders and of the (few) candidates with
// postings in several partitions, for which k_multi has walked the postings (live postings per partition in
// first-seen order, hence the exact tie score; all partitions).  Walk that list in the loop's order (count, score,
// word); keep, per partition p touched by an accepted word, ub[p] = the live segments p has left, an upper bound
// on the new count of anything that lives in p; then
//   * an entry none of whose partitions is touched has the key it had: it is the next winner if its count is above
//     `bound`, the largest count anything passed over or left behind can still have;
//   * an entry with a touched partition is passed over, and what it can still have (its live postings in untouched
//     partitions + min(its postings, ub) in touched ones) goes into `bound`;
//   * every other word stands behind its partition's leader, or in a touched partition (<= ub[p] <= bound).
// In the first iterations that is every partition's leader at once (119 winners in two iterations of 64).  An
// entry the fast path cannot judge (a word with more than kMultiParts partitions at the top, too many such
// candidates, P above kFastMaxP) leaves the iteration to k_score / k_prefix, which walk whatever they need.
constexpr int kFastMaxP = 2048;    // entries ranked by counting, n^2 / 1024 steps per thread
constexpr int kMultiParts = 4;     // partitions of a several-partition word the fast path keeps
constexpr int kFastTop = 256;      // entries of the order the walk may look at (accepted + passed over)
constexpr int kFastEnt = kFastMaxP + (int)kMaxMulti;
constexpr int kFastMiCache = 64;   // several-partition records the walk finds in LDS

constexpr int kMinorSegs = 4;      // a partition with at most this many live postings of the word: their segments are kept

struct MultiInfo {   // k_multi's result for mcand[j]
    uint32_t kid;
    int count;                                  // live postings = sum of live_cnt
    unsigned char n_live, n_all, overflow, pad; // overflow: more partitions than kMultiParts (live or all)
    unsigned short live_part[kMultiParts];      // in first-seen order of the live postings (ascending segment index)
    unsigned short all_part[kMultiParts];
    int live_cnt[kMultiParts];
    uint32_t first_seg[kMultiParts];            // the first live posting (segment index) of each live partition
    uint32_t seg[kMultiParts][kMinorSegs];      // live_cnt <= kMinorSegs: the live postings themselves, ascending
};

// One block per several-partition candidate: per-partition tallies of its live postings, the segment at which each
// partition is first seen live, the set of all its partitions, and the live postings of the partitions that hold
// only a few of them.
__global__ void __launch_bounds__(1024) k_multi(const Status *st, const uint32_t *mcand, const uint32_t *post_off,
                                                const uint32_t *post, const uint8_t *ignored, int P, int G,
                                                MultiInfo *out)
{
    __shared__ int cnt[kFastMaxP];
    __shared__ unsigned first[kFastMaxP];
    __shared__ unsigned allb[kFastMaxP / 32];
    __shared__ int n_l, n_a;
    __shared__ unsigned short lp[8], ap[8];
    __shared__ int lc[8], nseg[8];
    __shared__ unsigned lf[8];
    __shared__ uint32_t segs[8][kMinorSegs];
    const unsigned n = st->n_mcand;   // (this kernel writes no flag)
    if (st->stop || st->need_rebuild == 1 || st->want_general || n > kMaxMulti || P > kFastMaxP) return;
    const int tid = threadIdx.x, lane = tid & 63;
    for (unsigned j = blockIdx.x; j < n; j += gridDim.x) {
        const uint32_t kid = mcand[j];
        for (int e = tid; e < P; e += 1024) {
            cnt[e] = 0;
            first[e] = 0xffffffffu;
        }
        if (tid < kFastMaxP / 32) allb[tid] = 0u;
        if (tid < 8) nseg[tid] = 0;
        if (tid == 0) n_l = n_a = 0;
        __syncthreads();
        const uint32_t b = post_off[kid], e = post_off[kid + 1];
        for (uint32_t base = b; base < e; base += 4096) {
            uint32_t seg[4];
            bool live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = base + u * 1024 + tid;
                seg[u] = i < e ? post[i] : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                live[u] = seg[u] != 0xffffffffu && !ignored[row_of(seg[u], (uint32_t)P, (uint32_t)G)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int part = seg[u] != 0xffffffffu ? (int)(seg[u] % (uint32_t)P) : -1;
                unsigned long long m = __ballot(part >= 0);
                while (m) {   // wave-uniform: once per distinct partition of the wave's 64 postings
                    const int l = __ffsll((long long)m) - 1;
                    const int pl = __builtin_amdgcn_readlane(part, l);
                    const unsigned long long same = __ballot(part == pl);
                    const unsigned long long lives = __ballot(live[u] && part == pl);
                    if (lives) {   // wave-uniform
                        const uint32_t fs = (uint32_t)__builtin_amdgcn_readlane((int)seg[u], __ffsll((long long)lives) - 1);
                        if (lane == l) {
                            atomicAdd(&cnt[pl], (int)__popcll(lives));
                            atomicMin(&first[pl], fs);   // postings ascend: the lowest live lane holds the wave's first
                        }
                    }
                    if (lane == l) atomicOr(&allb[pl >> 5], 1u << (pl & 31));
                    m &= ~same;
                }
            }
        }
        __syncthreads();
        for (int p0 = tid; p0 < P; p0 += 1024) {
            if (cnt[p0] > 0) {
                const int at = atomicAdd(&n_l, 1);
                if (at < 8) {
                    lp[at] = (unsigned short)p0;
                    lc[at] = cnt[p0];
                    lf[at] = first[p0];
                }
            }
            if ((allb[p0 >> 5] >> (p0 & 31)) & 1u) {
                const int at = atomicAdd(&n_a, 1);
                if (at < 8) ap[at] = (unsigned short)p0;
            }
        }
        __syncthreads();
        const int nl = min(n_l, 8);
        // second look at the postings for the partitions with a few live ones: which segments they are
        bool any_minor = false;
        for (int x = 0; x < nl; ++x) any_minor = any_minor || lc[x] <= kMinorSegs;
        if (any_minor && n_l <= kMultiParts) {   // block-uniform
            for (uint32_t i = b + tid; i < e; i += 1024) {
                const uint32_t sg = post[i];
                const int part = (int)(sg % (uint32_t)P);
                if (cnt[part] <= kMinorSegs && !ignored[row_of(sg, (uint32_t)P, (uint32_t)G)]) {
                    int x = 0;
                    while (x < nl && lp[x] != (unsigned short)part) ++x;
                    const int at = atomicAdd(&nseg[x], 1);
                    if (at < kMinorSegs) segs[x][at] = sg;
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            MultiInfo mi;
            mi.kid = kid;
            mi.pad = 0;
            mi.overflow = (n_l > kMultiParts || n_a > kMultiParts) ? 1 : 0;
            mi.n_live = (unsigned char)min(n_l, kMultiParts);
            mi.n_all = (unsigned char)min(n_a, kMultiParts);
            int order[8];
            for (int x = 0; x < nl; ++x) order[x] = x;
            for (int x = 1; x < nl; ++x)   // first-seen order (a handful of entries)
                for (int y = x; y > 0 && lf[order[y]] < lf[order[y - 1]]; --y) {
                    const int t = order[y];
                    order[y] = order[y - 1];
                    order[y - 1] = t;
                }
            int total = 0;
            for (int x = 0; x < kMultiParts; ++x) {
                const int o = x < nl ? order[x] : 0;
                mi.live_part[x] = x < nl ? lp[o] : (unsigned short)0;
                mi.live_cnt[x] = x < nl ? lc[o] : 0;
                mi.first_seg[x] = x < nl ? lf[o] : 0u;
                mi.all_part[x] = x < min(n_a, 8) ? ap[x] : (unsigned short)0;
                for (int q = 0; q < kMinorSegs; ++q) mi.seg[x][q] = 0u;
                if (x < nl && lc[o] <= kMinorSegs) {
                    uint32_t v[kMinorSegs];
                    const int ns = min(nseg[o], kMinorSegs);
                    for (int q = 0; q < ns; ++q) v[q] = segs[o][q];
                    for (int q = 1; q < ns; ++q)   // ascending
                        for (int y = q; y > 0 && v[y] < v[y - 1]; --y) {
                            const uint32_t t = v[y];
                            v[y] = v[y - 1];
                            v[y - 1] = t;
                        }
                    for (int q = 0; q < ns; ++q) mi.seg[x][q] = v[q];
                }
            }
            for (int x = 0; x < nl; ++x) total += lc[x];
            mi.count = total;
            out[j] = mi;
        }
        __syncthreads();
    }
}

constexpr int kFastTasks = 1024;   // (candidate, minor partition, segment) checks of one iteration
constexpr int kMaxPend = 8;        // several-partition words waiting with a re-computed key
constexpr uint32_t kByMulti = 0xffffffffu;

// A several-partition word some of whose partitions were touched by winners of this iteration.  If every such
// partition either holds no live posting of the word, or holds a few (<= kMinorSegs) and was touched by nothing but
// its partition's one-partition leader, the word's new key is exact: the leader took the segments that hold it
// (hit bits, found before the walk from the segments' word lists), the rest stay, the partition's coverage is one up.
struct WalkWord {
    int ent;                  // entry index
    int j;                    // index into multi[]
    int cnt[kMultiParts];     // live postings left per live partition
    unsigned rem;             // 4 bits per live partition: which of its kept segments are left
    unsigned char seen[kMultiParts];   // per all_part: the partition's touch count the key accounts for
    unsigned long long key;   // count << 32 | score bits, as of now
};

__global__ void __launch_bounds__(1024) k_fast(Status *st, PickState *ps, const int32_t *count, const uint32_t *cand,
                                               const uint16_t *word_part, const uint8_t *word_multi,
                                               const MultiInfo *multi, const int32_t *live_part, uint32_t *coverage,
                                               const uint32_t *post_off, const uint64_t *ukeys,
                                               const int32_t *kid_of_inst, int per, int G, uint64_t *out_key,
                                               uint32_t *out_freq, uint32_t *out_trace, int P)
{
    __shared__ unsigned long long lead[kFastMaxP];   // per partition: count << 32 | ~word id of its one-partition leader
    __shared__ int second[kFastMaxP];                // ... and the highest count among its other one-partition candidates
    __shared__ unsigned long long lkey[kFastEnt];    // entries: count << 32 | score bits
    __shared__ uint32_t lkid[kFastEnt];
    __shared__ unsigned short lwho[kFastEnt];        // partition of a leader, or 0x8000 | index of a several-partition candidate
    __shared__ int ub[kFastMaxP];                    // live segments left in a partition touched in this iteration, -1: untouched
    __shared__ uint32_t who[kFastMaxP];              // the one-partition winner that touched it, kByMulti: something else did
    __shared__ unsigned char tcount[kFastMaxP], cov_add[kFastMaxP];   // touches / coverage bumps of this iteration
    __shared__ unsigned hitmask[kMaxMulti];          // bit 4 x + q: segment q of live partition x holds that partition's leader
    __shared__ unsigned char hit_known[kMaxMulti];
    __shared__ uint32_t task_row[kFastTasks], task_kid[kFastTasks];
    __shared__ unsigned short task_bit[kFastTasks], task_j[kFastTasks];
    __shared__ int top[kFastTop];
    __shared__ WalkWord walk_s[kMaxPend + 1];        // the walk's waiting words and the word in hand (one thread's; in
                                                     // private memory every access was a trip to scratch: 1 us per winner)
    __shared__ int rank_s[kFastEnt];                 // entries ahead of entry i (summed over the threads that share it)
    __shared__ unsigned long long ent_word[kFastEnt]; // the walk's global data, fetched by all the threads before it:
    __shared__ unsigned ent_chunks[kFastEnt];        //   an entry's word and 64-posting chunks (top entries only),
    __shared__ int lp_s[kFastMaxP];                  //   the partitions' live segments,
    __shared__ MultiInfo mi_s[kFastMiCache];         //   the several-partition records of the first such entries
    __shared__ unsigned char mi_slot[kMaxMulti];     //   (mi_slot[j]: where multi[j] is, 0xff: not cached)
    __shared__ int n_mi_s;
    __shared__ unsigned short bump[kMaxPick * kMultiParts];   // partitions whose coverage goes up
    __shared__ int red[16];
    __shared__ int n_ent_sh, n_bump, n_task;
    __shared__ int ws_r, ws_bound, ws_np, ws_nb, ws_nwin, ws_stop_next, ws_stop_now, ws_done;   // the walk's state between turns
    __shared__ unsigned ws_chunks;
    __shared__ uint32_t ws_last;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (single block, and the only writer of the loop's flags besides k_prefix, which does not run when this did)
    const int stop = st->stop, rebuild_state = st->need_rebuild;
    const unsigned n_cand = st->n_cand, n_mcand = st->n_mcand;
    const int theta = st->theta, n_win0 = st->n_win, max_iter = st->max_iter, stop_next0 = st->stop_next;
    const int want_general0 = st->want_general;
    __syncthreads();
    if (tid == 0) {
        ps->fast_done = 1;
        ps->valid = 0;
        ps->want_rebuild = 0;
        ps->too_many = 0;
        ps->n_pick = 0;
        ps->n_chunks = 0;
        n_ent_sh = 0;
        n_bump = 0;
        n_task = 0;
        n_mi_s = 0;
    }
    if (stop || rebuild_state == 1 || want_general0) {   // the host looks at these between batches
        if (tid == 0) st->it_idle += 1;
        return;
    }
    if (tid == 0 && rebuild_state == 2) {   // the list was rebuilt just before this iteration
        st->need_rebuild = 0;
        st->it_rebuild += 1;
    }
    if (n_win0 >= max_iter || stop_next0) {   // main.rs:344-366: the loop head
        if (tid == 0) st->stop = 1;
        return;
    }
    if (n_cand > kCandCap) {   // a list this long goes the five-launch way
        if (tid == 0) {
            st->need_rebuild = 1;
            st->too_many = 1;
        }
        return;
    }
    if (P > kFastMaxP || n_mcand > kMaxMulti) {
        if (tid == 0) {
            ps->fast_done = 0;
            st->want_general = 1;
        }
        return;
    }
    for (int e = tid; e < P; e += 1024) {
        lead[e] = 0ull;
        lp_s[e] = live_part[e];
        second[e] = 0;
        ub[e] = -1;
        who[e] = kByMulti;
        tcount[e] = 0;
        cov_add[e] = 0;
    }
    for (unsigned e = (unsigned)tid; e < n_mcand; e += 1024) {
        hitmask[e] = 0u;
        hit_known[e] = 1;
        mi_slot[e] = 0xff;
    }
    __syncthreads();
    int m = 0;   // highest count of all the candidates
    for (unsigned i = (unsigned)tid; i < n_cand; i += 1024) {
        const uint32_t kid = cand[i] & ~kCandLong;
        const int c = count[kid];
        m = max(m, c);
        if (!word_multi[kid] && c > 0) {
            // the partition's leader, and its runner-up in the same pass: whichever of the two keys loses this exchange
            // -- the candidate, or the leader it displaces -- is a word of the partition that is not its leader (the
            // final leader wins every exchange it is in), and every such word loses exactly one
            const int p = word_part[kid];
            const unsigned long long key = ((unsigned long long)(unsigned)c << 32) | (unsigned long long)(0xffffffffu - kid);
            const unsigned long long old = atomicMax(&lead[p], key);
            const int loser = (int)(min(old, key) >> 32);
            if (loser > 0) atomicMax(&second[p], loser);
        }
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = max(m, red[w]);
    if (m < theta) {   // a word outside the list may be ahead now
        if (tid == 0) st->need_rebuild = 1;
        return;
    }
    // (second[p], every partition's runner-up: once the leader is covered nothing that lives in the partition alone can
    // have more than the runner-up has now -- counts only fall -- however many live segments the partition keeps: the
    // bound that still works late in the loop, when a leader covers a small part of what its partition has left)
    __syncthreads();
    // the entries: leaders with a count of at least theta (nothing below it may be accepted from this list) ...
    for (int p0 = 0; p0 < P; p0 += 1024) {
        const int p = p0 + tid;
        const unsigned long long l = p < P ? lead[p] : 0ull;
        const bool have = (int)(l >> 32) >= theta;
        const unsigned long long mk = __ballot(have);
        int base = 0;
        if (mk && lane == 0) base = atomicAdd(&n_ent_sh, (int)__popcll(mk));
        base = __shfl(base, 0);
        if (have) {
            const int at = base + (int)__popcll(mk & ((1ull << lane) - 1ull));
            lkey[at] = (l & 0xffffffff00000000ull) | (unsigned long long)__float_as_uint(1.0f / ((float)coverage[p] + 1.0f));
            lkid[at] = 0xffffffffu - (uint32_t)(l & 0xffffffffull);
            lwho[at] = (unsigned short)p;
        }
    }
    // ... and the several-partition candidates with theirs: the score as partition_tie_score adds it up, in f32, one term
    // per live partition in first-seen order (an overflowing one gets no score: the walk stops at it).  For each of
    // their partitions with a few live postings: does a segment hold that partition's leader?  (One check per
    // segment and window position, all of them in flight together.)
    for (unsigned j = (unsigned)tid; j < n_mcand; j += 1024) {
        const MultiInfo &mi = multi[j];
        const int mcount = mi.overflow ? count[mi.kid] : mi.count;   // (an overflowing word's tallies are not all kept)
        if (mcount >= theta) {
            float acc = 0.0f;
            for (int x = 0; x < mi.n_live; ++x) acc += 1.0f / ((float)coverage[mi.live_part[x]] + 1.0f);
            // (a word with more partitions than are kept has no score here: it goes in front of everything of its
            //  count, where the walk stops at it before it could accept a word that word may be ahead of)
            if (mi.overflow) acc = __uint_as_float(0x7f7fffffu);
            const int at = atomicAdd(&n_ent_sh, 1);
            lkey[at] = ((unsigned long long)(unsigned)mcount << 32) | (unsigned long long)__float_as_uint(acc);
            lkid[at] = mi.kid;
            lwho[at] = (unsigned short)(0x8000u | j);
            if (!mi.overflow)
                for (int x = 0; x < mi.n_live; ++x) {
                    const int p = mi.live_part[x];
                    if (mi.live_cnt[x] > kMinorSegs || (int)(lead[p] >> 32) < theta) continue;   // no leader that could touch it
                    for (int q = 0; q < mi.live_cnt[x]; ++q) {
                        const int at2 = atomicAdd(&n_task, 1);
                        if (at2 < kFastTasks) {
                            task_row[at2] = row_of(mi.seg[x][q], (uint32_t)P, (uint32_t)G);
                            task_kid[at2] = 0xffffffffu - (uint32_t)(lead[p] & 0xffffffffull);
                            task_bit[at2] = (unsigned short)(4 * x + q);
                            task_j[at2] = (unsigned short)j;
                        } else {
                            hit_known[j] = 0;
                        }
                    }
                }
        }
    }
    __syncthreads();
    {
        const int nt = min(n_task, kFastTasks);
        for (int id = tid; id < nt * per; id += 1024) {
            const int t = id / per, q = id - t * per;
            if ((uint32_t)kid_of_inst[inst_slot(task_row[t], (uint32_t)q, (uint32_t)per, (uint32_t)G)] == task_kid[t]) atomicOr(&hitmask[task_j[t]], 1u << task_bit[t]);
        }
    }
    const int n_ent = n_ent_sh;
    __syncthreads();
    // rank by counting; equal (count, score): the smaller word first.  1024 / n_ent threads share an entry.
    for (int i = tid; i < n_ent; i += 1024) rank_s[i] = 0;
    __syncthreads();
    {
        const int share = max(1, 1024 / max(1, n_ent)), piece = (n_ent + share - 1) / share;
        for (int t = tid; t < n_ent * share; t += 1024) {
            const int i = t / share, j0 = (t - i * share) * piece, j1 = min(n_ent, j0 + piece);
            const unsigned long long ki = lkey[i];
            const uint32_t di = lkid[i];
            int rank = 0;
            for (int j = j0; j < j1; ++j) {
                const unsigned long long kj = lkey[j];
                rank += (kj > ki || (kj == ki && lkid[j] < di)) ? 1 : 0;
            }
            if (rank) atomicAdd(&rank_s[i], rank);
        }
    }
    __syncthreads();
    for (int i = tid; i < n_ent; i += 1024)
        if (rank_s[i] < kFastTop) top[rank_s[i]] = i;
    __syncthreads();
    {
        // The walk below is one thread's, and every global load in it would be a dependent one (about 200 ns each, six per
        // winner when it read them itself): all the threads fetch what it will read -- the entries' words and posting-list
        // lengths into LDS, the several-partition records and the coverage into this CU's L1.
        unsigned sink = 0;
        const int lim = min(n_ent, kFastTop);
        for (int r = tid; r < lim; r += 1024) {
            const int i = top[r];
            const uint32_t kid = lkid[i];
            ent_word[i] = ukeys[kid];
            ent_chunks[i] = (post_off[kid + 1] - post_off[kid] + 63u) / 64u;
            if (lwho[i] & 0x8000u) {
                const int at = atomicAdd(&n_mi_s, 1);   // (in no particular order: whichever 64 come first)
                if (at < kFastMiCache) mi_slot[lwho[i] & 0x7fffu] = (unsigned char)at;
            }
        }
        __syncthreads();
        static_assert(sizeof(MultiInfo) % 4 == 0, "copied word by word");
        constexpr int kMiWords = (int)(sizeof(MultiInfo) / 4);
        for (int t = tid; t < lim * 32; t += 1024) {   // 32 threads per entry
            const int i = top[t >> 5];
            if (!(lwho[i] & 0x8000u)) continue;
            const int j = lwho[i] & 0x7fffu, slot = mi_slot[j];
            if (slot == 0xff) continue;
            const unsigned *src = reinterpret_cast<const unsigned *>(&multi[j]);
            unsigned *dst = reinterpret_cast<unsigned *>(&mi_s[slot]);
            for (int wd = t & 31; wd < kMiWords; wd += 32) dst[wd] = src[wd];
        }
        for (int p = tid; p < P; p += 1024) sink += coverage[p];
        if (sink == 0x9e3779b9u) red[0] = (int)sink;   // (keeps the loads)
    }
    __syncthreads();
    const int min_freq = st->min_freq;
    const uint32_t it_now = (uint32_t)(st->it_fast + st->it_general + 1);
    const int limit = min(n_ent, kFastTop);
    if (tid == 0) {
        ws_r = 0;
        ws_bound = 0;
        ws_np = 0;
        ws_nb = 0;
        ws_chunks = 0u;
        ws_nwin = n_win0;
        ws_stop_next = stop_next0;
        ws_stop_now = 0;
        ws_done = 0;
        ws_last = 0u;
    }
    __syncthreads();
    // The order is walked in turns.  Wave 0 takes runs of partition LEADERS, for which the walk is a scan: lane r takes
    // the entry of rank r; a leader whose partition is untouched is accepted when its count is above the bound its
    // predecessors leave -- the running maximum of what each of them can still leave behind: min(live segments its
    // partition keeps, its partition's runner-up) for an accepted leader, min(count, live segments left) for a leader
    // whose partition a several-partition winner has touched (passed over) -- and the first one that is not ends the
    // iteration: a prefix maximum and two ballots, the accepted lanes record their winners side by side.  Thread 0 takes
    // what the scan cannot: several-partition words (their keys are re-computed against what has been accepted, and may
    // wait in `pend`), and the loop's stop rules.  (All of it was thread 0's once: 1 us per entry, dependent LDS trips.)
    int npend = 0;   // thread 0's, across its turns
    for (int turn = 0; turn < kFastTop + 2; ++turn) {   // block-uniform; every turn finishes the iteration or takes an entry
        if (wave == 0) {
            const int r0 = ws_r, np0 = ws_np, nb0 = ws_nb, nwin0 = ws_nwin, bound0 = ws_bound, sn0 = ws_stop_next;
            const unsigned chunks0 = ws_chunks;
            const int idx = r0 + lane;
            const int i = idx < limit ? top[idx] : -1;
            const bool leader = i >= 0 && !(lwho[i] & 0x8000u);
            const unsigned long long others = __ballot(!leader);
            const int n_lead = others ? __ffsll((long long)others) - 1 : 64;
            const unsigned long long below = (1ull << lane) - 1ull;
            int f = 0, p = 0, b = 0;
            uint32_t kid = 0;
            unsigned ch = 0;
            bool fresh = false;   // a leader whose partition nothing has touched in this iteration
            if (lane < n_lead) {
                f = (int)(lkey[i] >> 32);
                p = lwho[i];
                kid = lkid[i];
                const int u = ub[p];
                fresh = u < 0;
                b = fresh ? min(lp_s[p] - f, max(second[p], theta - 1)) : min(f, u);
                ch = fresh ? ent_chunks[i] : 0u;
            }
            const unsigned long long fresh_m = __ballot(fresh);
            int pm = b;   // inclusive prefix maximum of b, inclusive prefix sum of the fresh leaders' chunks
            unsigned cs = ch;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(pm, off);
                const unsigned u = __shfl_up(cs, off);
                if (lane >= off) {
                    pm = max(pm, t);
                    cs += u;
                }
            }
            int before = __shfl_up(pm, 1);   // the bound entry r meets
            before = lane == 0 ? bound0 : max(bound0, before);
            const int a_idx = (int)__popcll(fresh_m & below);   // fresh leaders ahead of this lane: all accepted, if it gets that far
            // main.rs:387-390: a winner below min_freq is the last one
            const unsigned long long low_m = __ballot(fresh && f < min_freq);
            const bool sn = sn0 || (low_m & below) != 0ull;
            const bool ends = fresh && f <= before;   // something passed over or left behind may be ahead: the next iteration decides
            const bool hands = fresh && !ends && (f <= 1 || nwin0 + a_idx >= max_iter || sn || np0 + a_idx >= kMaxPick);
            const unsigned long long stop_m = __ballot(ends || hands);
            const int s_lane = stop_m ? __ffsll((long long)stop_m) - 1 : 64;   // everything before it is settled
            const int n_set = min(s_lane, n_lead);
            if (fresh && lane < n_set) {
                out_key[nwin0 + a_idx] = ent_word[i];
                out_freq[nwin0 + a_idx] = (uint32_t)f;
                out_trace[nwin0 + a_idx] = (it_now << 8) | 1u;
                ps->kid[np0 + a_idx] = kid;
                ps->cum[np0 + a_idx] = chunks0 + cs - ch;
                ub[p] = lp_s[p] - f;
                who[p] = kid;
                tcount[p] = 1;
                cov_add[p] += 1;
                bump[nb0 + a_idx] = (unsigned short)p;
            }
            const unsigned long long acc_m = fresh_m & (n_set >= 64 ? ~0ull : (1ull << n_set) - 1ull);
            const int n_acc = (int)__popcll(acc_m);
            const int last_lane = acc_m ? 63 - __clzll((long long)acc_m) : 0;
            const uint32_t last_kid = __shfl(kid, last_lane);
            const int pm_set = __shfl(pm, max(n_set, 1) - 1);
            const unsigned cs_set = __shfl(cs, max(n_set, 1) - 1);
            const bool ends_s = __shfl(ends ? 1 : 0, s_lane & 63) != 0;
            if (lane == 0) {
                ws_r = r0 + n_set;
                ws_np = np0 + n_acc;
                ws_nb = nb0 + n_acc;
                ws_nwin = nwin0 + n_acc;
                ws_bound = n_set ? max(bound0, pm_set) : bound0;
                ws_chunks = chunks0 + (n_set ? cs_set : 0u);
                ws_stop_next = sn0 || (low_m & acc_m) != 0ull;
                if (n_acc) ws_last = last_kid;
                if (s_lane < n_lead && ends_s) ws_done = 1;
            }
        }
        __syncthreads();
        if (tid == 0 && !ws_done) {
            int n_win = ws_nwin, np = ws_np, nb = ws_nb, stop_next = ws_stop_next, stop_now = 0, bound = ws_bound;
            unsigned chunks = ws_chunks;
            uint32_t last_kid = ws_last;
            WalkWord *pend = walk_s;
            WalkWord &w = walk_s[kMaxPend];
            int r = ws_r;
            bool handed = false, first = true;
            while (np < kMaxPick) {
                // a leader next and no word waiting: wave 0's turn again (after at least one entry of this turn)
                if (!first && npend == 0 && r < limit && !(lwho[top[r]] & 0x8000u)) {
                    handed = true;
                    break;
                }
                first = false;
                // the next entry of the order: the list's, or a waiting word whose new key is ahead of it
                int pk = -1;
                for (int q = 0; q < npend; ++q)
                    if (pk < 0 || pend[q].key > pend[pk].key ||
                        (pend[q].key == pend[pk].key && lkid[pend[q].ent] < lkid[pend[pk].ent]))
                        pk = q;
                const int mi_i = r < limit ? top[r] : -1;
                if (pk < 0 && mi_i < 0) break;
                bool from_pend = mi_i < 0;
                if (pk >= 0 && mi_i >= 0)
                    from_pend = pend[pk].key > lkey[mi_i] || (pend[pk].key == lkey[mi_i] && lkid[pend[pk].ent] < lkid[mi_i]);
                if (from_pend) {
                    w = pend[pk];
                    pend[pk] = pend[--npend];
                } else {
                    ++r;
                    w.ent = mi_i;
                    w.key = lkey[mi_i];
                    w.j = (lwho[mi_i] & 0x8000u) ? (int)(lwho[mi_i] & 0x7fffu) : -1;
                    w.rem = 0xffffu;
                    if (w.j >= 0) {
                        const MultiInfo *m0 = mi_slot[w.j] != 0xff ? &mi_s[mi_slot[w.j]] : &multi[w.j];
                        for (int x = 0; x < kMultiParts; ++x) {
                            w.cnt[x] = m0->live_cnt[x];
                            w.seen[x] = 0;
                        }
                    }
                }
                const int f = (int)(w.key >> 32), ent = w.ent, wj = w.j;
                const uint32_t kid = lkid[ent];
                const MultiInfo *mi = wj >= 0 ? (mi_slot[wj] != 0xff ? &mi_s[mi_slot[wj]] : &multi[wj]) : nullptr;
                if (!mi) {   // a partition's leader
                    const int p = lwho[ent];
                    if (ub[p] >= 0) {   // its partition was touched: it is passed over, with what it can still have
                        bound = max(bound, min(f, ub[p]));
                        continue;
                    }
                } else {
                    if (mi->overflow) break;   // the fast path cannot judge this word: nothing behind it is safe
                    bool need = false, bad = false;
                    for (int x = 0; x < mi->n_all; ++x) need = need || tcount[mi->all_part[x]] != w.seen[x];
                    if (need) {
                        for (int x = 0; x < mi->n_all && !bad; ++x) {
                            const int p = mi->all_part[x];
                            if (tcount[p] == w.seen[x]) continue;
                            int xl = -1;
                            for (int y = 0; y < mi->n_live; ++y)
                                if (mi->live_part[y] == p) xl = y;
                            if (xl >= 0 && w.cnt[xl] > 0) {
                                // live postings in a touched partition: exact only for a few of them against the partition's leader
                                if (mi->live_cnt[xl] > kMinorSegs || tcount[p] != 1 || w.seen[x] != 0 || who[p] == kByMulti ||
                                    who[p] != 0xffffffffu - (uint32_t)(lead[p] & 0xffffffffull) || !hit_known[wj]) {
                                    bad = true;
                                    break;
                                }
                                const unsigned hits = (hitmask[wj] >> (4 * xl)) & ((1u << mi->live_cnt[xl]) - 1u);
                                w.rem &= ~(hits << (4 * xl));
                                w.cnt[xl] -= (int)__popc(hits);
                            }
                            w.seen[x] = tcount[p];
                        }
                        if (bad) {   // passed over with what it can still have
                            int still = 0;
                            for (int y = 0; y < mi->n_live; ++y) {
                                const int u = ub[mi->live_part[y]];
                                still += u >= 0 ? min(w.cnt[y], u) : w.cnt[y];
                            }
                            bound = max(bound, still);
                            continue;
                        }
                        // its key as of now: the live partitions in first-seen order, the coverage as this iteration leaves it
                        int total = 0, ord[kMultiParts], no = 0;
                        uint32_t fs[kMultiParts];
                        for (int y = 0; y < mi->n_live; ++y) {
                            if (w.cnt[y] <= 0) continue;
                            total += w.cnt[y];
                            uint32_t f0 = mi->first_seg[y];
                            if (mi->live_cnt[y] <= kMinorSegs) {
                                const unsigned left = (w.rem >> (4 * y)) & 0xfu;
                                f0 = mi->seg[y][__ffs((int)left) - 1];
                            }
                            int at = no++;
                            while (at > 0 && fs[at - 1] > f0) {
                                fs[at] = fs[at - 1];
                                ord[at] = ord[at - 1];
                                --at;
                            }
                            fs[at] = f0;
                            ord[at] = y;
                        }
                        float acc = 0.0f;
                        for (int q = 0; q < no; ++q) {
                            const int p = mi->live_part[ord[q]];
                            acc += 1.0f / ((float)(coverage[p] + (uint32_t)cov_add[p]) + 1.0f);
                        }
                        w.key = ((unsigned long long)(unsigned)total << 32) | (unsigned long long)__float_as_uint(acc);
                        if (total >= theta) {
                            if (npend < kMaxPend) pend[npend++] = w;
                            else bound = max(bound, total);
                        }
                        continue;   // (below theta: behind everything this list may yield)
                    }
                }
                if (f <= bound) break;   // something passed over or left behind may be ahead: the next iteration decides
                if (n_win >= max_iter || stop_next) {   // main.rs:344: the loop head
                    stop_now = 1;
                    break;
                }
                if (f <= 1) {   // main.rs:353-366 (cannot happen: f >= theta >= 2)
                    stop_now = 1;
                    break;
                }
                out_key[n_win] = ent_word[ent];
                out_freq[n_win] = (uint32_t)f;
                // trace: iteration (fast + general + 1) << 8 | 1 leader, 2 several-partition word, 3 the same after a re-computed key
                out_trace[n_win] = (it_now << 8) | (mi ? (from_pend ? 3u : 2u) : 1u);
                ++n_win;
                ps->kid[np] = kid;
                ps->cum[np] = chunks;
                chunks += ent_chunks[ent];
                last_kid = kid;
                ++np;
                if (f < min_freq) stop_next = 1;   // main.rs:387-390: stop after the push
                if (mi) {
                    // partitions it only has covered segments in lose nothing, but their coverage goes up
                    for (int x = 0; x < mi->n_all; ++x) {
                        const int p = mi->all_part[x];
                        if (ub[p] < 0) ub[p] = lp_s[p];
                        who[p] = kByMulti;
                        tcount[p] = (unsigned char)min(255, tcount[p] + 1);
                        cov_add[p] += 1;
                        bump[nb++] = (unsigned short)p;
                    }
                    for (int y = 0; y < mi->n_live; ++y)
                        if (w.cnt[y] > 0) ub[mi->live_part[y]] -= w.cnt[y];
                    // what a word living in one of these partitions can still have: no more than the partition has left,
                    // and no more than it had (its partition's leader's count, or less than theta if it is not on the list)
                    for (int x = 0; x < mi->n_all; ++x) {
                        const int p = mi->all_part[x];
                        bound = max(bound, min(ub[p], max((int)(lead[p] >> 32), theta - 1)));
                    }
                } else {
                    const int p = lwho[ent];
                    ub[p] = lp_s[p] - f;   // live segments p keeps (what a several-partition word's postings there are measured against)
                    who[p] = kid;
                    tcount[p] = 1;
                    cov_add[p] += 1;
                    // what a word living in p alone can still have: no more than that, and no more than p's runner-up had
                    // (a word that is not on the list has less than theta)
                    bound = max(bound, min(ub[p], max(second[p], theta - 1)));
                    bump[nb++] = (unsigned short)p;
                }
            }
            ws_r = r;
            ws_np = np;
            ws_nb = nb;
            ws_nwin = n_win;
            ws_bound = bound;
            ws_chunks = chunks;
            ws_stop_next = stop_next;
            ws_last = last_kid;
            if (stop_now) ws_stop_now = 1;
            if (!handed) ws_done = 1;
        }
        __syncthreads();
        const int done = ws_done;
        __syncthreads();   // (wave 0 writes the state again in its next turn)
        if (done) break;
    }
    if (tid == 0) {
        const int n_win = ws_nwin, np = ws_np, nb = ws_nb, stop_next = ws_stop_next, stop_now = ws_stop_now;
        const unsigned chunks = ws_chunks;
        const uint32_t last_kid = ws_last;
        ps->cum[np] = chunks;
        ps->n_pick = (unsigned)np;
        ps->n_chunks = chunks;
        st->n_win = n_win;
        st->stop_next = stop_next;
        if (stop_now) st->stop = 1;
        if (np) st->winner = (int)last_kid;
        st->maxf = 0;
        if (np == 0 && !stop_now) {   // left to k_score / k_prefix
            ps->fast_done = 0;
            st->want_general = 1;
        }
        if (np) st->it_fast += 1;
        n_bump = nb;
    }
    __syncthreads();
    if (tid < n_bump) atomicAdd(&coverage[bump[tid]], 1u);   // main.rs:371-378: once per partition of a winner's posting list
}

__global__ void __launch_bounds__(1024) k_score(const Status *st, PickState *ps, const int32_t *count,
                                                const uint32_t *cand, const uint32_t *post_off, const uint32_t *post,
                                                const uint8_t *ignored, const uint32_t *coverage, int P, int G,
                                                unsigned long long *res_key, uint32_t *res_kid, unsigned *res_bits)
{
    constexpr int kKeep = 4;   // rounds of 1024 candidates whose (word, count) stay in registers
    extern __shared__ unsigned char smem[];
    __shared__ int red[16];
    __shared__ unsigned wl[kKeep * 16], ws[kKeep * 16];   // E's words per (round, wave), then their prefix sums
    __shared__ unsigned tot_sh[2];
    __shared__ uint32_t own_long[kOwn], own_short[kOwn];
    __shared__ unsigned hist[kHistBins];
    __shared__ int sel_sh[2];   // deepest bin of E, its size
    __shared__ int go_sh;
    __shared__ TieBlockShared tb;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the loop's flags once per block (this kernel writes none of them: k_prefix does)
    if (tid == 0)
        go_sh = st->want_general && !(st->stop || st->need_rebuild == 1 || st->n_win >= st->max_iter || st->stop_next);
    if (tid < kHistBins) hist[tid] = 0u;
    __syncthreads();
    if (!go_sh) return;
    const int words = (P + 31) / 32;
    unsigned *seen_blk = (unsigned *)smem, *all_blk = seen_blk + words;
    unsigned *seen_w = all_blk + (size_t)words * (1 + 2 * wave), *all_w = seen_w + words;
    const unsigned n_cand = st->n_cand;
    const int theta = st->theta;
    if (n_cand > kCandCap) {   // every block reads the whole list: too long for that, the host goes the other way
        if (blockIdx.x == 0 && tid == 0) {
            ps->too_many = 1;
            ps->valid = 1;
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
    if (m < theta) {   // a word outside the list may be ahead now
        if (blockIdx.x == 0 && tid == 0) {
            ps->want_rebuild = 1;
            ps->valid = 1;
        }
        return;
    }
    // E = the words with count >= m - d for the largest d that keeps |E| <= kETarget and the counts >= theta;
    // d = 0 (the words tied at the maximum) whatever their number
#pragma unroll
    for (int r = 0; r < kKeep; ++r)
        if (ck[r] != 0xffffffffu && m - cc[r] < kHistBins) atomicAdd(&hist[m - cc[r]], 1u);
    for (unsigned i = kKeep * 1024u + (unsigned)tid; i < n_cand; i += 1024) {
        const int d = m - count[cand[i] & ~kCandLong];
        if (d < kHistBins) atomicAdd(&hist[d], 1u);
    }
    __syncthreads();
    if (wave == 0) {
        unsigned v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = hist[lane * 4 + q];
            sum += v[q];
        }
        unsigned incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        unsigned run = incl - sum;
        int ok = 0;   // bins of this lane that E may reach (the admissible bins are a prefix of all the bins)
        unsigned cum_at[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            run += v[q];
            cum_at[q] = run;
            const int bin = lane * 4 + q;
            ok += (bin == 0 || (run <= (unsigned)kETarget && m - bin >= theta)) ? 1 : 0;
        }
        int total = ok;
        for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
        const int dsel = total - 1;   // >= 0: bin 0 always counts
        if (lane == (dsel >> 2)) {
            sel_sh[0] = dsel;
            sel_sh[1] = (int)cum_at[dsel & 3];
        }
    }
    __syncthreads();
    const int dsel = sel_sh[0];
    if (sel_sh[1] > kEMax) {   // (only possible with d = 0) more words tied at the maximum than one iteration scores
        if (blockIdx.x == 0 && tid == 0) {
            ps->too_many = 1;
            ps->valid = 1;
        }
        return;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    // number E's words (long and short posting lists apart) in list order; word t of a class belongs to block
    // t % gridDim.x, which keeps its share in own_long / own_short, and owns result slot t (long) or
    // kEMax - 1 - t (short)
    unsigned long long bl[kKeep], bs[kKeep];
#pragma unroll
    for (int r = 0; r < kKeep; ++r) {
        const bool hit = ck[r] != 0xffffffffu && m - cc[r] <= dsel;
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
#pragma unroll
    for (int r = 0; r < kKeep; ++r) {
        const bool hit = ck[r] != 0xffffffffu && m - cc[r] <= dsel;
        if (hit) {
            const bool lg = (ck[r] & kCandLong) != 0;
            const unsigned t = lg ? wl[r * 16 + wave] + (unsigned)__popcll(bl[r] & below)
                                  : ws[r * 16 + wave] + (unsigned)__popcll(bs[r] & below);
            if (t % gridDim.x == blockIdx.x) (lg ? own_long : own_short)[t / gridDim.x] = ck[r] & ~kCandLong;
        }
    }
    unsigned tot_l = tot_sh[0], tot_s = tot_sh[1];
    for (unsigned base = kKeep * 1024u; base < n_cand; base += 1024) {   // the rest of a long list
        const unsigned i = base + (unsigned)tid;
        uint32_t kid = 0;
        bool hit = false, lg = false;
        if (i < n_cand) {
            kid = cand[i];
            hit = m - count[kid & ~kCandLong] <= dsel;
            lg = hit && (kid & kCandLong);
        }
        const unsigned long long xl = __ballot(lg), xs = __ballot(hit && !lg);
        __syncthreads();   // red is read by everybody before it is rewritten
        if (lane == 0) red[wave] = (int)((unsigned)__popcll(xl) << 16 | (unsigned)__popcll(xs));
        __syncthreads();
        unsigned pl = tot_l, pss = tot_s;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned v = (unsigned)red[w];
            if (w < wave) {
                pl += v >> 16;
                pss += v & 0xffffu;
            }
            tot_l += v >> 16;
            tot_s += v & 0xffffu;
        }
        if (hit) {
            const unsigned t = lg ? pl + (unsigned)__popcll(xl & below) : pss + (unsigned)__popcll(xs & below);
            if (t % gridDim.x == blockIdx.x) (lg ? own_long : own_short)[t / gridDim.x] = kid & ~kCandLong;
        }
    }
    __syncthreads();   // the lists are complete (tot_l + tot_s = |E| <= kEMax: a block's share always fits)
    const unsigned nl = tot_l > blockIdx.x ? (tot_l - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const unsigned ns = tot_s > blockIdx.x ? (tot_s - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    for (unsigned j = 0; j < nl; ++j) {
        const uint32_t kid = own_long[j];
        const unsigned slot = j * gridDim.x + blockIdx.x;
        tie_score_block(kid, seen_blk, all_blk, &tb, words, post_off, post, ignored, coverage, P, G);
        if (tid == 0) {
            res_key[slot] = ((unsigned long long)(unsigned)count[kid] << 32) | (unsigned long long)__float_as_uint(tb.acc);
            res_kid[slot] = kid;
        }
        for (int wd = tid; wd < words; wd += 1024) {
            res_bits[((size_t)slot * 2) * words + wd] = seen_blk[wd];
            res_bits[((size_t)slot * 2 + 1) * words + wd] = all_blk[wd];
        }
        __syncthreads();
    }
    for (unsigned j = wave; j < ns; j += 16) {
        const uint32_t kid = own_short[j];
        const unsigned slot = (unsigned)kEMax - 1u - (j * gridDim.x + blockIdx.x);
        const float acc = tie_score_wave(kid, seen_w, all_w, words, lane, post_off, post, ignored, coverage, P, G);
        if (lane == 0) {
            res_key[slot] = ((unsigned long long)(unsigned)count[kid] << 32) | (unsigned long long)__float_as_uint(acc);
            res_kid[slot] = kid;
        }
        for (int wd = lane; wd < words; wd += 64) {
            res_bits[((size_t)slot * 2) * words + wd] = seen_w[wd];
            res_bits[((size_t)slot * 2 + 1) * words + wd] = all_w[wd];
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        ps->m = m;
        ps->tot_l = tot_l;
        ps->tot_s = tot_s;
        ps->valid = 1;
    }
}

// One block.  Ranks E by (count, score) descending and word ascending (= id ascending: ids follow the sorted key
// order), takes the longest prefix of untouched words, applies the stop rules and records the winners.
__global__ void __launch_bounds__(1024) k_prefix(Status *st, PickState *ps, const unsigned long long *res_key,
                                                 const uint32_t *res_kid, const unsigned *res_bits,
                                                 const uint32_t *post_off, const uint64_t *ukeys, uint32_t *coverage,
                                                 uint64_t *out_key, uint32_t *out_freq, uint32_t *out_trace, int P)
{
    __shared__ unsigned long long key_s[kEMax];
    __shared__ uint32_t kid_s[kEMax];
    __shared__ unsigned short slot_s[kEMax];
    __shared__ int top[kMaxPick + 1];   // position in key_s of the word of rank r
    __shared__ int first_bad, n_acc;
    const int tid = threadIdx.x;
    const int words = (P + 31) / 32;
    // (everything this kernel reads was written by earlier kernels; it is the only writer of the loop's flags)
    const int valid = ps->valid, want_rebuild = ps->want_rebuild, too_many = ps->too_many;
    const unsigned tot_l = ps->tot_l, tot_s = ps->tot_s;
    const int stop = st->stop, rebuild_state = st->need_rebuild;
    if (!st->want_general) return;   // block-uniform: only behind a k_fast that left the iteration to the walking path
    __syncthreads();
    if (tid == 0) st->want_general = 0;
    if (tid == 0) {
        ps->valid = 0;
        ps->want_rebuild = 0;
        ps->too_many = 0;
        ps->n_pick = 0;
        ps->n_chunks = 0;
        first_bad = kMaxPick + 1;
        n_acc = 0;
    }
    if (stop || rebuild_state == 1) return;
    if (tid == 0 && rebuild_state == 2) st->need_rebuild = 0;   // the list was rebuilt at the head of this iteration
    if (!valid) {
        // k_score left at its loop-head test (main.rs:344-366): the loop is over
        if (tid == 0 && (st->n_win >= st->max_iter || st->stop_next)) st->stop = 1;
        return;
    }
    if (want_rebuild || too_many) {
        if (tid == 0) {
            st->need_rebuild = 1;
            if (too_many) st->too_many = 1;
        }
        return;
    }
    const int n_e = (int)(tot_l + tot_s);
    for (int i = tid; i < n_e; i += 1024) {
        const unsigned slot = (unsigned)i < tot_l ? (unsigned)i : (unsigned)kEMax - 1u - ((unsigned)i - tot_l);
        key_s[i] = res_key[slot];
        kid_s[i] = res_kid[slot];
        slot_s[i] = (unsigned short)slot;
    }
    __syncthreads();
    // rank by counting (|E| is about a hundred; a few thousand only when that many words tie at the maximum)
    for (int i = tid; i < n_e; i += 1024) {
        const unsigned long long ki = key_s[i];
        const uint32_t di = kid_s[i];
        int rank = 0;
        for (int j = 0; j < n_e; ++j) {
            const unsigned long long kj = key_s[j];
            rank += (kj > ki || (kj == ki && kid_s[j] < di)) ? 1 : 0;
        }
        if (rank <= kMaxPick) top[rank] = i;
    }
    __syncthreads();
    const int n_top = min(n_e, kMaxPick + 1);
    // first word of the order that an earlier one touches: live partitions of r against all partitions of q < r
    for (int pr = tid; pr < n_top * n_top; pr += 1024) {
        const int q = pr / n_top, r = pr % n_top;
        if (q >= r || r >= first_bad) continue;
        const unsigned *aq = res_bits + ((size_t)slot_s[top[q]] * 2 + 1) * words;
        const unsigned *lr = res_bits + ((size_t)slot_s[top[r]] * 2) * words;
        unsigned hit = 0;
        for (int wd = 0; wd < words; ++wd) hit |= aq[wd] & lr[wd];
        if (hit) atomicMin(&first_bad, r);
    }
    __syncthreads();
    if (tid == 0) {
        const int limit = min(min(first_bad, n_top), kMaxPick);
        int n_win = st->n_win, np = 0;
        unsigned chunks = 0;
        const int max_iter = st->max_iter, min_freq = st->min_freq;
        int stop_next = st->stop_next, stop_now = 0;
        for (int r = 0; r < limit; ++r) {
            if (n_win >= max_iter || stop_next) {   // main.rs:344: the loop head
                stop_now = 1;
                break;
            }
            const int i = top[r];
            const int f = (int)(key_s[i] >> 32);
            if (f <= 1) {   // main.rs:353-366: no word, or the best word is in one segment only
                stop_now = 1;
                break;
            }
            const uint32_t kid = kid_s[i];
            out_key[n_win] = ukeys[kid];
            out_freq[n_win] = (uint32_t)f;
            out_trace[n_win] = ((uint32_t)(st->it_fast + st->it_general + 1) << 8) | 4u;   // 4: after posting walks
            ++n_win;
            ps->kid[np] = kid;
            ps->cum[np] = chunks;
            chunks += (post_off[kid + 1] - post_off[kid] + 63u) / 64u;
            ++np;
            if (f < min_freq) stop_next = 1;   // main.rs:387-390: stop after the push
        }
        ps->cum[np] = chunks;
        ps->n_pick = (unsigned)np;
        ps->n_chunks = chunks;
        st->n_win = n_win;
        st->stop_next = stop_next;
        if (stop_now) st->stop = 1;
        if (np) st->winner = (int)ps->kid[np - 1];
        st->maxf = 0;
        if (np) st->it_general += 1;
        n_acc = np;
    }
    __syncthreads();
    // main.rs:371-378: the coverage of every partition a winner's posting list touches goes up by one
    const int np = n_acc;
    for (int e = tid; e < np * words; e += 1024) {
        const int r = e / words, wd = e % words;
        unsigned bits = res_bits[((size_t)slot_s[top[r]] * 2 + 1) * words + wd];
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            atomicAdd(&coverage[wd * 32 + b], 1u);
        }
    }
}

// Cover every segment that holds a winner: take one off the live count of every word of the segments that are
// covered now.  A WAVE takes 64 postings at a time, lane = posting: it does the per-posting bookkeeping, then walks
// the window positions and reads the 64 segments' word ids straight from the position-major table (neighbouring
// postings are the same window of neighbouring genomes: neighbouring words, and mostly the SAME word id, so equal
// targets are merged across the wave before the atomic -- same-address atomics serialise in L2).  No block barrier:
// every wave of the chip has its own chunk in flight.
__device__ __forceinline__ void cover_wave(uint32_t row, bool ok, int per, int G, const int32_t *kid_of_inst, int32_t *count)
{
    constexpr int kDeep = 8;   // window positions whose loads are in flight together
    const int lane = threadIdx.x & 63;
    const uint32_t part = ok ? row / (uint32_t)G : 0u;
    const int32_t *col = kid_of_inst + ((size_t)part * per) * G + (ok ? row - part * (uint32_t)G : 0u);
    for (int q0 = 0; q0 < per; q0 += kDeep) {   // wave-uniform
        int32_t v[kDeep];
#pragma unroll
        for (int j = 0; j < kDeep; ++j) v[j] = (ok && q0 + j < per) ? col[(size_t)(q0 + j) * G] : -1;
#pragma unroll
        for (int j = 0; j < kDeep; ++j) {
            const int32_t k2 = v[j];
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
    }
}

// The same with the count updates collected in a table in LDS first (k_cover_multi: the 16 waves of a block take 16
// neighbouring chunks of one winner and meet the same words over and over -- the consensus word of a window position
// and its variants; measured on 64 winners of 8,000 postings each, the global atomics were three quarters of the
// kernel's time).  Open addressing, linear probing, keys claimed with a compare-and-swap; a lane that finds no place
// within kProbeMax steps (a table fuller than it should ever be) sends its update to memory directly.
constexpr int kCoverTbl = 4096;    // entries (32 KB): a block's 1,024 segments x ~40 window positions hold ~1,000 distinct words
constexpr int kProbeMax = 32;
constexpr unsigned kCoverHeavy = 128;   // work items (of 16 chunks) from which an iteration counts as heavy

__device__ __forceinline__ void table_sub(int32_t *tkey, int32_t *tval, int32_t kid, int n, int32_t *count)
{
    unsigned h = ((unsigned)kid * 2654435761u) >> 20;
#pragma unroll 1
    for (int step = 0; step < kProbeMax; ++step) {
        const int32_t old = atomicCAS(&tkey[h], -1, kid);
        if (old == -1 || old == kid) {
            atomicAdd(&tval[h], n);
            return;
        }
        h = (h + 1u) & (unsigned)(kCoverTbl - 1);
    }
    atomicSub(&count[kid], n);
}

// `heavy`: an iteration with thousands of chunks, where the variants' updates go through the table as well (their
// global atomics were half the kernel's time there); in a light one they go to memory directly, fire and forget --
// a table update is a compare-and-swap the lane waits for, one LDS round trip per window position, and a late
// iteration has nothing else to hide it behind.  The consensus words of a round of window positions are handed to
// the lanes, one position each, so that their table updates are ONE round trip, not one per position.
__device__ __forceinline__ void cover_wave_table(uint32_t row, bool ok, int per, int G, const int32_t *kid_of_inst,
                                                 int32_t *count, int32_t *tkey, int32_t *tval, bool heavy)
{
    constexpr int kDeep = 20;   // window positions whose loads are in flight together (a 50-base window of 13-mers: two rounds)
    const int lane = threadIdx.x & 63;
    const uint32_t part = ok ? row / (uint32_t)G : 0u;
    const int32_t *col = kid_of_inst + ((size_t)part * per) * G + (ok ? row - part * (uint32_t)G : 0u);
    for (int q0 = 0; q0 < per; q0 += kDeep) {   // wave-uniform
        int32_t v[kDeep];
#pragma unroll
        for (int j = 0; j < kDeep; ++j) v[j] = (ok && q0 + j < per) ? col[(size_t)(q0 + j) * G] : -1;
        int32_t my_kid = -1;
        int my_n = 0;
#pragma unroll
        for (int j = 0; j < kDeep; ++j) {
            const int32_t k2 = v[j];
            const unsigned long long mk = __ballot(k2 >= 0);
            if (mk) {   // the window position's most likely word: one update for all its lanes (lane j's, below)
                const int l = __ffsll((long long)mk) - 1;
                const int32_t kl = __builtin_amdgcn_readlane(k2, l);
                const unsigned long long same = __ballot(k2 == kl);
                if (lane == j) {
                    my_kid = kl;
                    my_n = (int)__popcll(same);
                }
                if (k2 == kl) v[j] = -1;   // settled
            }
        }
        if (my_kid >= 0) table_sub(tkey, tval, my_kid, my_n, count);
#pragma unroll
        for (int j = 0; j < kDeep; ++j) {   // the variants
            const int32_t k2 = v[j];
            if (k2 >= 0) {
                if (heavy) table_sub(tkey, tval, k2, 1, count);
                else atomicSub(&count[k2], 1);
            }
        }
    }
}

// live_part[p] = segments of partition p that are not covered yet (k_fast's bound): the newly covered ones of a
// wave's 64 postings leave it, one atomic per distinct partition
__device__ __forceinline__ void take_live(int32_t *live_part, bool live, int part, int lane)
{
    // (`live` here: newly covered AND marked, i.e. counted in live_part by k_mark)
    unsigned long long ml = __ballot(live);
    while (ml) {   // wave-uniform
        const int l = __ffsll((long long)ml) - 1;
        const int pl = __builtin_amdgcn_readlane(part, l);
        const unsigned long long same = __ballot(live && part == pl);
        if (lane == l) atomicSub(&live_part[pl], (int)__popcll(same));
        ml &= ~same;
    }
}

// The five-launch iteration's cover step: one winner (st->best), the partition coverage bumped once per
// distinct partition of its posting list (main.rs:371-378, covered segments included) through a stamp per
// partition; the last block records the winner.
__global__ void __launch_bounds__(256) k_cover(Status *st, const uint32_t *post_off,
                                               const uint32_t *post, uint8_t *ignored,
                                               uint32_t *coverage, uint32_t *stamp, int P, int G,
                                               int per, const int32_t *kid_of_inst, int32_t *count,
                                               const uint64_t *ukeys, uint64_t *out_key,
                                               uint32_t *out_freq, int32_t *live_part, const uint8_t *marked)
{
    if (st->stop) return;
    const uint32_t it1 = (uint32_t)st->n_win + 1u;   // unique stamp of this iteration
    const unsigned long long best = st->best;
    const uint32_t kid = 0xffffffffu - (uint32_t)(best & 0xffffffffull);
    const uint32_t b = post_off[kid], e = post_off[kid + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = b + (blockIdx.x * 4u + wave) * 64u; base < e; base += gridDim.x * 256u) {   // wave-uniform
        bool rep = false;          // this lane speaks for its partition
        uint32_t old_stamp = it1;
        int part = -1;
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
        take_live(live_part, live && marked[row], part, lane);
        unsigned long long m = __ballot(part >= 0);
        while (m) {   // once per distinct partition of the wave
            const int l = __ffsll((long long)m) - 1;
            const int pl = __builtin_amdgcn_readlane(part, l);
            m &= ~__ballot(part == pl);
            rep = rep || lane == l;
        }
        // all the partitions at once; the answer is looked at after the count updates are on their way
        if (rep) old_stamp = atomicExch(&stamp[part], it1);
        if (__ballot(live)) cover_wave(row, live, per, G, kid_of_inst, count);
        if (rep && old_stamp != it1) atomicAdd(&coverage[part], 1u);   // first posting of the partition this iteration
    }
    __syncthreads();
    if (threadIdx.x == 0) {
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

// The candidate-list iteration's cover step: all the winners k_prefix recorded, a 64-posting chunk of one of
// them per work item.  The winners' LIVE segments lie in pairwise different partitions (that is what made them a
// prefix), so the count updates of different winners never meet in a segment -- with one exception, a
// several-partition word accepted with a re-computed key, which is why a segment is claimed with an atomic before
// it is covered (below).  (The partition coverage is k_fast's / k_prefix's.)
__global__ void __launch_bounds__(1024) k_cover_multi(const PickState *ps, const uint32_t *post_off,
                                                      const uint32_t *post, uint8_t *ignored, int P, int G, int per,
                                                      const int32_t *kid_of_inst, int32_t *count, int32_t *live_part,
                                                      const uint8_t *marked)
{
    constexpr unsigned kSpan = 16;   // chunks per work item: one per wave
    __shared__ int32_t tkey[kCoverTbl], tval[kCoverTbl];
    __shared__ uint32_t cum_s[kMaxPick + 1], kid_s[kMaxPick], span_s[kMaxPick + 1];
    const unsigned np = ps->n_pick;
    if (np == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {   // np <= 64: the winners' spans and their running sum in one wave
        const unsigned c0 = (unsigned)lane <= np ? ps->cum[lane] : 0u;
        const unsigned c1 = (unsigned)lane < np ? ps->cum[lane + 1] : 0u;
        const unsigned mine = (unsigned)lane < np ? (c1 - c0 + kSpan - 1) / kSpan : 0u;
        unsigned acc = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(acc, off);
            if (lane >= off) acc += t;
        }
        cum_s[lane] = c0;
        if (lane == 63 && np == 64) cum_s[64] = c1;
        span_s[lane] = acc - mine;
        if (lane == 63) span_s[64] = acc;
        if ((unsigned)lane < np) kid_s[lane] = ps->kid[lane];
    }
    __syncthreads();
    const unsigned total = span_s[np < 64 ? np : 64];
    if (blockIdx.x >= total) return;   // block-uniform: most blocks of a late iteration
    for (int e = tid; e < kCoverTbl; e += 1024) {
        tkey[e] = -1;
        tval[e] = 0;
    }
    __syncthreads();
    for (unsigned item = blockIdx.x; item < total; item += gridDim.x) {   // block-uniform
        unsigned lo = 0, hi = np;   // the winner this span belongs to: the last j with span[j] <= item
        while (hi - lo > 1) {
            const unsigned mid = (lo + hi) >> 1;
            if (span_s[mid] <= item) lo = mid;
            else hi = mid;
        }
        const uint32_t kid = kid_s[lo];
        const unsigned chunk = (item - span_s[lo]) * kSpan + (unsigned)wave;
        if (chunk < cum_s[lo + 1] - cum_s[lo]) {   // wave-uniform
            const uint32_t i = post_off[kid] + chunk * 64u + lane, e = post_off[kid + 1];
            uint32_t row = 0;
            int part = -1;
            bool live = false, mk = false;
            if (i < e) {
                const uint32_t seg = post[i];
                part = (int)(seg % (uint32_t)P);
                row = (uint32_t)part * (uint32_t)G + seg / (uint32_t)P;
                mk = marked[row] != 0;   // (issued with the claim, not behind it)
                // Two winners of one iteration may share a segment that is still live: a several-partition word
                // accepted with a re-computed key lists the segments of its minor partition that the partition's
                // leader takes as well.  The reference covers it once (the second winner finds it ignored), so the
                // segment is CLAIMED: one atomic on the flag's 32-bit word, whoever sets the byte covers it.
                const unsigned sh8 = 8u * (row & 3u);
                const unsigned old = atomicOr(reinterpret_cast<unsigned *>(ignored + (row & ~3u)), 1u << sh8);
                live = ((old >> sh8) & 0xffu) == 0u;
            }
            take_live(live_part, live && mk, part, lane);
            if (__ballot(live)) cover_wave_table(row, live, per, G, kid_of_inst, count, tkey, tval, total >= kCoverHeavy);
        }
        __syncthreads();
        for (int e = tid; e < kCoverTbl; e += 1024) {   // the table goes to memory and is empty again
            const int32_t k = tkey[e];
            if (k >= 0) {
                atomicSub(&count[k], tval[e]);
                tkey[e] = -1;
                tval[e] = 0;
            }
        }
        __syncthreads();
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

void KmerStage::drop_loop_graph()
{
    if (loop_exec_) (void)hipGraphExecDestroy(loop_exec_);
    if (loop_graph_) (void)hipGraphDestroy(loop_graph_);
    loop_exec_ = nullptr;
    loop_graph_ = nullptr;
    loop_sig_.clear();
}

void KmerStage::release()
{
    drop_loop_graph();
    if (pinned_) (void)hipHostFree(pinned_);
    pinned_ = nullptr;
    for (int s = 0; s < 19; ++s) {
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
    // buffers: 0/1 keys, 2/3 vals, 4 (unused), 5 first instances per index block and their scan, 6 kid_of_inst, 7 post, 8 post_off, 9 ukeys,
    // 10 count+tied, 11 ignored, 12 coverage+stamp, 13 status/out, 14 cub temp, 16 the candidate-list loop's
    // PickState + per-word results (key, id, two partition bitmaps)
    if ((rc = ensure(0, n_inst * 8, err)) || (rc = ensure(1, n_inst * 8, err)) ||
        (rc = ensure(3, n_inst * 4, err)) ||
        (rc = ensure(5, 2 * 4 * ((n_inst + kIndexBlock - 1) / kIndexBlock), err)) ||
        (rc = ensure(6, n_inst * 4, err)) || (rc = ensure(7, n_inst * 4, err)) ||
        (rc = ensure(8, (n_inst + 1) * 4, err)) || (rc = ensure(9, n_inst * 8, err)) ||
        (rc = ensure(10, n_inst * 8, err)) || (rc = ensure(11, ((size_t)n_seg + 7) & ~(size_t)3, err)) ||
        (rc = ensure(12, (size_t)P * 8, err)) ||
        (rc = ensure(13, 256 + (size_t)opt.max_iterations * 16, err)))
        return rc;
    const size_t pwords = (size_t)((P + 31) / 32);
    const size_t pick_bytes = (sizeof(PickState) + 255) & ~(size_t)255;
    const size_t res_bytes = ((size_t)kEMax * (8 + 4 + 2 * 4 * pwords) + 255) & ~(size_t)255;
    const size_t multi_bytes = ((size_t)kMaxMulti * sizeof(MultiInfo) + 255) & ~(size_t)255;
    if ((rc = ensure(16, pick_bytes + res_bytes + multi_bytes + (size_t)kMaxMulti * 4, err))) return rc;
    uint64_t *key_a = (uint64_t *)buf_[0], *key_b = (uint64_t *)buf_[1];
    uint32_t *val_b = (uint32_t *)buf_[3];
    const size_t n_iblk = (n_inst + kIndexBlock - 1) / kIndexBlock;
    uint32_t *block_heads = (uint32_t *)buf_[5], *block_base = block_heads + n_iblk;
    int32_t *kid_of_inst = (int32_t *)buf_[6];
    uint32_t *post = (uint32_t *)buf_[7], *post_off = (uint32_t *)buf_[8];
    uint64_t *ukeys = (uint64_t *)buf_[9];
    int32_t *count = (int32_t *)buf_[10];
    uint32_t *tied = (uint32_t *)buf_[10] + n_inst;
    uint8_t *ignored = (uint8_t *)buf_[11];
    uint32_t *coverage = (uint32_t *)buf_[12], *stamp = coverage + P;
    Status *st = (Status *)buf_[13];
    uint64_t *out_key = (uint64_t *)((char *)buf_[13] + 256);
    uint32_t *out_freq = (uint32_t *)(out_key + opt.max_iterations);
    uint32_t *out_trace = out_freq + opt.max_iterations;   // per winner: how it was selected (diagnostics)
    PickState *ps = (PickState *)buf_[16];
    unsigned long long *res_key = (unsigned long long *)((char *)buf_[16] + pick_bytes);
    uint32_t *res_kid = (uint32_t *)(res_key + kEMax);
    unsigned *res_bits = (unsigned *)(res_kid + kEMax);
    MultiInfo *multi = (MultiInfo *)((char *)buf_[16] + pick_bytes + res_bytes);
    uint32_t *mcand = (uint32_t *)((char *)multi + multi_bytes);

    // 1. extraction (from the packed alignment: byte rows are packed first), 2. inverted index (the key type by
    //    the word length)
    const uint64_t *d_packed = d_seqs.packed;
    if (!d_packed) {
        if ((rc = ensure(18, sizeof(uint64_t) * (size_t)n_seq * SeqView::row_words(seq_len), err))) return rc;
        KM_TRY(launch_pack_rows(d_seqs.ascii, n_seq, seq_len, (uint64_t *)buf_[18], stream));
        d_packed = (const uint64_t *)buf_[18];
    }
    const int ext_grid = std::min((n_seg + 3) / 4, 8192);   // one window per wave at a time
    if (per > 256) {
        err = "stage A: search window too wide for the extraction kernel";
        return MSSPE_ERR_ARG;
    }
    int M = 0;
    uint16_t *word_part = nullptr;   // [M] partition of a word's first posting
    uint8_t *word_multi = nullptr;   // [M] 1: the word has postings in several partitions
    int32_t *live_part = nullptr;    // [P] live segments of the partition that hold a word of the current candidate list
    uint32_t *cand_flag = nullptr;   // [M] the list (Status::epoch) a word was last on
    uint8_t *marked = nullptr;       // [n_seg] the segment was counted in live_part
    auto build_index = [&](auto key_tag) -> int {
        using Key = decltype(key_tag);
        Key *ka = (Key *)key_a, *kb = (Key *)key_b;
        hipLaunchKernelGGL(k_extract<Key>, dim3(ext_grid), dim3(256), 0, stream, d_packed, seq_len, n_seg,
                           (int)P, opt.segment_size, opt.overlap_size, W, k, direction, per, ka);
        KM_TRY(hipGetLastError());
        // stable radix sort on the 2k+1 key bits keeps ascending segment order; the values (instance numbers) are
        // not stored anywhere before the sort: a counting iterator supplies them
        const rocprim::counting_iterator<uint32_t> val_a(0u);
        size_t tmp_bytes = 0, tmp2 = 0;
        // 9 bits per pass where that saves a pass (13-mers: 27 key bits, three passes instead of four)
        using Sort9 = rocprim::radix_sort_config<
            rocprim::default_config, rocprim::default_config,
            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 9,
                                                rocprim::block_radix_rank_algorithm::match>>;
        const unsigned key_bits = (unsigned)(2 * k + 1);
        const bool nine = (key_bits + 8) / 9 < (key_bits + 7) / 8;
        KM_TRY(nine ? rocprim::radix_sort_pairs<Sort9>(nullptr, tmp_bytes, ka, kb, val_a, val_b, n_inst, 0u, key_bits, stream)
                    : rocprim::radix_sort_pairs(nullptr, tmp_bytes, ka, kb, val_a, val_b, n_inst, 0u, key_bits, stream));
        KM_TRY(rocprim::exclusive_scan(nullptr, tmp2, block_heads, block_base, 0u, n_iblk, rocprim::plus<uint32_t>(), stream));
        int rc2;
        if ((rc2 = ensure(14, std::max(tmp_bytes, tmp2), err))) return rc2;
        KM_TRY(nine ? rocprim::radix_sort_pairs<Sort9>(buf_[14], tmp_bytes, ka, kb, val_a, val_b, n_inst, 0u, key_bits, stream)
                    : rocprim::radix_sort_pairs(buf_[14], tmp_bytes, ka, kb, val_a, val_b, n_inst, 0u, key_bits, stream));
        hipLaunchKernelGGL(k_head_count<Key>, dim3((unsigned)((n_iblk + 3) / 4)), dim3(256), 0, stream, kb, n_inst, sentinel, block_heads, n_iblk);
        KM_TRY(rocprim::exclusive_scan(buf_[14], tmp2, block_heads, block_base, 0u, n_iblk, rocprim::plus<uint32_t>(), stream));
        uint32_t last_heads = 0, last_base = 0;   // the number of words: the last block's base + its own first instances
        KM_TRY(hipMemcpyAsync(&last_heads, block_heads + n_iblk - 1, 4, hipMemcpyDeviceToHost, stream));
        KM_TRY(hipMemcpyAsync(&last_base, block_base + n_iblk - 1, 4, hipMemcpyDeviceToHost, stream));
        KM_TRY(hipStreamSynchronize(stream));
        M = (int)(last_heads + last_base);
        if (M == 0) return MSSPE_OK;
        // number of valid instances = first sentinel position: post_off[M]
        const size_t wp_bytes = ((size_t)M * 3 + 15) & ~(size_t)15, lp_bytes = (4 * (size_t)P + 15) & ~(size_t)15;
        if ((rc2 = ensure(17, wp_bytes + lp_bytes + 4 * (size_t)M + (size_t)n_seg, err))) return rc2;
        word_part = (uint16_t *)buf_[17];
        word_multi = (uint8_t *)(word_part + M);
        live_part = (int32_t *)((char *)buf_[17] + wp_bytes);
        cand_flag = (uint32_t *)((char *)buf_[17] + wp_bytes + lp_bytes);
        marked = (uint8_t *)(cand_flag + M);
        {
            FillJobs fj{};
            int nj = 0;
            auto job = [&](void *ptr, size_t bytes, unsigned char value) {
                fj.p[nj] = ptr;
                fj.n[nj] = bytes;
                fj.v[nj++] = value;
            };
            job(live_part, lp_bytes + 4 * (size_t)M, 0);   // live_part, cand_flag: epoch 0 = on no list
            job(marked, (size_t)n_seg, 1);                  // the first list counts every segment (k_live_all)
            job(word_multi, (size_t)M, 0);                  // (k_index sets it)
            job(ignored, (size_t)n_seg, 0);
            job(coverage, (size_t)P * 8, 0);
            job(ps, sizeof(PickState), 0);
            job(out_trace, sizeof(uint32_t) * (size_t)opt.max_iterations, 0);
            fj.count = nj;
            hipLaunchKernelGGL(k_fill_jobs, dim3(512), dim3(256), 0, stream, fj);
        }
        hipLaunchKernelGGL(k_index<Key>, dim3((unsigned)((n_iblk + 3) / 4)), dim3(256), 0, stream, kb, val_b, block_base, n_inst, n_iblk,
                           sentinel, per, (int)P, n_seq, M, kid_of_inst, post, post_off, ukeys, word_part, word_multi);
        return MSSPE_OK;
    };
    if ((rc = 2 * k + 1 <= 32 ? build_index(uint32_t{}) : build_index(uint64_t{}))) return rc;
    if (M == 0) return MSSPE_OK;
    hipLaunchKernelGGL(k_init_counts, dim3((M + 255) / 256), dim3(256), 0, stream, post_off, M, count);
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
    h0.need_rebuild = 1;   // the candidate-list loop starts by making its list
    if (!pinned_) KM_TRY(hipHostMalloc(&pinned_, 256, hipHostMallocDefault));
    *(Status *)pinned_ = h0;   // (read back into only after this copy, in stream order)
    KM_TRY(hipMemcpyAsync(st, pinned_, sizeof h0, hipMemcpyHostToDevice, stream));
    auto enqueue_iteration = [&](hipStream_t s_, int /*node*/) {
        hipLaunchKernelGGL(k_max_count<false>, dim3(red_grid), dim3(256), 0, s_, count, M, st, 0, (int32_t *)nullptr, 0);
        hipLaunchKernelGGL(k_collect_tied, dim3((M + 255) / 256), dim3(256), 0, s_, count, M, st,
                           post_off, tied);
        hipLaunchKernelGGL(k_tie_scores, dim3(256), dim3(256), tie_lds, s_, tied, st, post_off, post,
                           ignored, coverage, (int)P, n_seq);
        hipLaunchKernelGGL(k_tie_long, dim3(64), dim3(1024), tie_lds / 4, s_, tied, M, st, post_off,
                           post, ignored, coverage, (int)P, n_seq);
        hipLaunchKernelGGL(k_cover, dim3(256), dim3(256), 0, s_, st, post_off, post, ignored, coverage,
                           stamp, (int)P, n_seq, per, kid_of_inst, count, ukeys, out_key, out_freq, live_part, marked);
    };
    // the candidate-list iteration; `tied` doubles as the candidate list
    uint32_t *cand = tied;
    const size_t sel_lds = 34 * sizeof(unsigned) * pwords;   // a pair of partition bitmaps for the block and for each wave
    // the candidate-list loop's three pieces.  A batch (graph) holds iterations only; the host looks at the flags
    // between batches and puts a list rebuild or one walking iteration in front of the next batch when asked to
    // (the remaining iterations of a batch that asked are no-ops)
    bool first_list = true;
    auto enqueue_rebuild = [&](hipStream_t s_) {
        // the maximum, theta, the candidate list, the marked segments (and the stop decision)
        hipLaunchKernelGGL(k_max_count<true>, dim3(red_grid), dim3(256), 0, s_, count, M, st, 0, live_part, (int)P);
        hipLaunchKernelGGL(k_collect_cand, dim3((M + 255) / 256), dim3(256), 0, s_, count, M, st, post_off, cand,
                           word_multi, mcand, cand_flag);
        if (first_list) {   // nothing is covered yet: every segment counts (marked was set to ones with the other state)
            hipLaunchKernelGGL(k_live_all, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s_, st, live_part, (int)P, n_seq);
            first_list = false;
        } else {
            hipLaunchKernelGGL(k_mark, dim3(1024), dim3(256), 0, s_, st, kid_of_inst, ignored, cand_flag, n_seg, per,
                               n_seq, marked, live_part);
        }
    };
    auto enqueue_general = [&](hipStream_t s_) {
        // an iteration k_fast could not settle: posting walks for the words at the top of the order
        hipLaunchKernelGGL(k_score, dim3(kSelectGrid), dim3(1024), sel_lds, s_, st, ps, count, cand, post_off, post,
                           ignored, coverage, (int)P, n_seq, res_key, res_kid, res_bits);
        hipLaunchKernelGGL(k_prefix, dim3(1), dim3(1024), 0, s_, st, ps, res_key, res_kid, res_bits, post_off, ukeys,
                           coverage, out_key, out_freq, out_trace, (int)P);
        hipLaunchKernelGGL(k_cover_multi, dim3(kCoverGrid), dim3(1024), 0, s_, ps, post_off, post, ignored, (int)P, n_seq, per,
                           kid_of_inst, count, live_part, marked);
    };
    auto enqueue_narrow = [&](hipStream_t s_, int /*node*/) {
        // the few candidates with postings in several partitions are walked; everything else is read off the counts
        hipLaunchKernelGGL(k_multi, dim3(kSelectGrid), dim3(1024), 0, s_, st, mcand, post_off, post, ignored, (int)P,
                           n_seq, multi);
        hipLaunchKernelGGL(k_fast, dim3(1), dim3(1024), 0, s_, st, ps, count, cand, word_part, word_multi, multi,
                           live_part, coverage, post_off, ukeys, kid_of_inst, per, n_seq, out_key, out_freq, out_trace,
                           (int)P);
        hipLaunchKernelGGL(k_cover_multi, dim3(kCoverGrid), dim3(1024), 0, s_, ps, post_off, post, ignored, (int)P, n_seq, per,
                           kid_of_inst, count, live_part, marked);
    };
    // a batch of kBatch iterations is ONE graph (fewer graph launches than one graph per iteration)
    constexpr int kBatch = 32;
    constexpr int kBatchN = 8;    // candidate-list iterations per graph (an iteration selects up to 64 winners)
    GraphGuard gg[2];   // 0: five-launch iterations, 1: candidate-list iterations
    bool use_graph = use_graph_;   // option "stage_a_graph" (0: plain launches, a testing aid)
    auto capture = [&](int which) -> bool {
        // capture on a private stream so that the caller's stream may be of any kind
        hipStream_t cs = nullptr;
        bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
        if (ok && hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            for (int b = 0; b < (which ? kBatchN : kBatch); ++b) which ? enqueue_narrow(cs, b) : enqueue_iteration(cs, b);
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
    // the candidate-list graph of an earlier run serves this one if every argument enqueue_narrow passes is unchanged
    const std::vector<uint64_t> narrow_sig = {
        (uint64_t)(uintptr_t)st, (uint64_t)(uintptr_t)ps, (uint64_t)(uintptr_t)mcand, (uint64_t)(uintptr_t)multi,
        (uint64_t)(uintptr_t)post_off, (uint64_t)(uintptr_t)post, (uint64_t)(uintptr_t)ignored, (uint64_t)(uintptr_t)count,
        (uint64_t)(uintptr_t)cand, (uint64_t)(uintptr_t)word_part, (uint64_t)(uintptr_t)word_multi,
        (uint64_t)(uintptr_t)live_part, (uint64_t)(uintptr_t)coverage, (uint64_t)(uintptr_t)ukeys,
        (uint64_t)(uintptr_t)kid_of_inst, (uint64_t)(uintptr_t)out_key, (uint64_t)(uintptr_t)out_freq,
        (uint64_t)(uintptr_t)out_trace, (uint64_t)(uintptr_t)marked, (uint64_t)P, (uint64_t)n_seq, (uint64_t)per};
    if (loop_exec_ && loop_sig_ != narrow_sig) drop_loop_graph();
    if (!pinned_) KM_TRY(hipHostMalloc(&pinned_, 256, hipHostMallocDefault));
    Status *h_pin = (Status *)pinned_;
    Status h = h0;
    int wide_left = 0;   // five-launch batches still to run after a candidate list came out too long
    bool out_of_batches = false;
    for (int batch = 0; !h.stop; ++batch) {
        if (narrow_ok ? batch >= 2 * h0.max_iter + 4 : batch * kBatch >= h0.max_iter + 1) {
            out_of_batches = true;
            break;
        }
        const int which = narrow_ok && wide_left == 0 ? 1 : 0;
        if (wide_left) --wide_left;
        if (which && h.need_rebuild == 1) enqueue_rebuild(stream);
        if (which && h.want_general && h.need_rebuild != 1) enqueue_general(stream);
        if (use_graph && which == 1 && !loop_exec_) {
            if (capture(1)) {   // kept by the stage (drop_loop_graph), not by this call's guard
                loop_graph_ = gg[1].graph;
                loop_exec_ = gg[1].exec;
                gg[1].graph = nullptr;
                gg[1].exec = nullptr;
                loop_sig_ = narrow_sig;
            } else {
                use_graph = false;
            }
        } else if (use_graph && which == 0 && !gg[0].exec && !capture(0)) {
            use_graph = false;
        }
        if (use_graph) {
            KM_TRY(hipGraphLaunch(which ? loop_exec_ : gg[0].exec, stream));
        } else {
            for (int b = 0; b < (which ? kBatchN : kBatch); ++b) which ? enqueue_narrow(stream, b) : enqueue_iteration(stream, b);
        }
        KM_TRY(hipGetLastError());
        KM_TRY(hipMemcpyAsync(h_pin, st, sizeof h, hipMemcpyDeviceToHost, stream));   // pinned: no staging copy per batch
        KM_TRY(hipStreamSynchronize(stream));
        h = *h_pin;
        if (which && h.too_many) wide_left = 4;
    }
    if (out_of_batches && !h.stop) {
        // every batch selects a winner or stops the loop; a loop that used up its bound without the device's stop
        // flag would return a truncated list as if it were whole
        err = "stage A: the greedy loop did not finish within its batch bound (" + std::to_string(h.n_win) + " winners so far)";
        return MSSPE_ERR_DEVICE;
    }
    const int n_win = h.n_win;
    loop_stats_[0] = h.it_fast;
    loop_stats_[1] = h.it_general;
    loop_stats_[2] = h.it_rebuild;
    loop_stats_[3] = h.it_idle;
    if (narrow_ok && !h.stop_next && n_win >= capacity && n_win < opt.max_iterations) {
        // the candidate-list loop stops at max_iter without looking at the words: is anything left?
        KM_TRY(hipMemsetAsync((char *)st + offsetof(Status, stop), 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_max_count<true>, dim3(red_grid), dim3(256), 0, stream, count, M, st, 1, (int32_t *)nullptr, 0);
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
        trace_.resize((size_t)n_win);
        const size_t mi = (size_t)opt.max_iterations, all = 16 * mi;
        if (all <= (1u << 20)) {
            // words, frequencies and trace lie one behind the other: one copy (each pageable copy is a host round trip)
            std::vector<unsigned char> host(all);
            KM_TRY(hipMemcpyAsync(host.data(), out_key, all, hipMemcpyDeviceToHost, stream));
            KM_TRY(hipStreamSynchronize(stream));
            const uint64_t *hk = reinterpret_cast<const uint64_t *>(host.data());
            for (int i = 0; i < n_win; ++i) words_out[i] = lex_to_packed(hk[i], k);
            std::memcpy(freq_out, host.data() + 8 * mi, sizeof(uint32_t) * (size_t)n_win);
            std::memcpy(trace_.data(), host.data() + 12 * mi, sizeof(uint32_t) * (size_t)n_win);
        } else {
            std::vector<uint64_t> hk((size_t)n_win);
            KM_TRY(hipMemcpyAsync(hk.data(), out_key, sizeof(uint64_t) * n_win, hipMemcpyDeviceToHost, stream));
            KM_TRY(hipMemcpyAsync(freq_out, out_freq, sizeof(uint32_t) * n_win, hipMemcpyDeviceToHost, stream));
            KM_TRY(hipMemcpyAsync(trace_.data(), out_trace, sizeof(uint32_t) * n_win, hipMemcpyDeviceToHost, stream));
            KM_TRY(hipStreamSynchronize(stream));
            for (int i = 0; i < n_win; ++i) words_out[i] = lex_to_packed(hk[i], k);
        }
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

