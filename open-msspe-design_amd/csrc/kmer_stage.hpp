// kmer_stage.hpp -- stage A on the device: k-mer candidate generation
// (replaces get_segment_manager + find_candidates_kmers,
//  /root/reference/od-msspe/src/main.rs:196-235 and :331-406).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/msspe_hip.h"

namespace msspe {

// The alignment as the kernels read it.  ascii != nullptr: n_seq x seq_len bytes, upper case, U -> T
// (od-msspe/src/main.rs:115-118).  Otherwise packed rows of row_words() uint64 each: first
// (seq_len + 31) / 32 words of 2-bit bases (A 0, C 1, G 2, T 3; column c in bits [2 (c % 32), +1] of word
// c / 32), then (seq_len + 63) / 64 words of validity bits (1 = the column holds A, C, G or T).
struct SeqView {
    const uint8_t *ascii;
    const uint64_t *packed;
    size_t seq_len;
    static size_t base_words(size_t seq_len) { return (seq_len + 31) / 32; }
    static size_t row_words(size_t seq_len) { return (seq_len + 31) / 32 + (seq_len + 63) / 64; }
};

// device ASCII rows (n_rows x row_len) -> packed rows (n_rows x SeqView::row_words(row_len))
hipError_t launch_pack_rows(const uint8_t *d_ascii, int n_rows, size_t row_len, uint64_t *d_packed,
                            hipStream_t stream);

class KmerStage {
public:
    // d_seqs: device, n_seq x seq_len bytes.  words_out / freq_out / n_out: host buffers.
    int run(const SeqView &seqs, int n_seq, size_t seq_len, const msspe_kmer_opt &opt,
            int direction, uint64_t *words_out, uint32_t *freq_out, int capacity, int *n_out,
            hipStream_t stream, std::string &err);
    // Segment coverage of a primer set (main.rs:518-594): hit_out[seq * P + partition] = 1 when the
    // segment's head window holds a forward primer or its tail window the reverse complement of a
    // reverse primer.  fwd_words / rev_words / hit_out: host buffers.
    int coverage(const SeqView &seqs, int n_seq, size_t seq_len, const msspe_kmer_opt &opt,
                 const uint64_t *fwd_words, int n_fwd, const uint64_t *rev_words, int n_rev,
                 uint8_t *hit_out, hipStream_t stream, std::string &err);
    void release();
    void set_use_graph(bool on) { use_graph_ = on; }
    // false: every iteration of the greedy loop scans all the words (the five-launch iteration)
    void set_narrow_loop(bool on) { narrow_loop_ = on; }
    // the last run's candidate-list loop: iterations that recorded winners without a posting walk / with one,
    // iterations that rebuilt the list, idle iterations at the end of the last batch
    const int *loop_stats() const { return loop_stats_; }
    // per winner of the last run: iteration << 8 | how it was selected (1 a partition's leader, 2 a several-partition
    // word, 3 the same with a re-computed key, 4 after posting walks, 0 the all-words loop)
    const std::vector<uint32_t> &trace() const { return trace_; }

private:
    void *buf_[19] = {};
    size_t cap_[19] = {};
    void *pinned_ = nullptr;   // the loop state's host copy (pinned: read back once per batch)
    bool use_graph_ = true;
    bool narrow_loop_ = true;
    int loop_stats_[4] = {};
    std::vector<uint32_t> trace_;
    // the candidate-list loop's graph (eight iterations), kept across runs for as long as its kernels' arguments are
    // what they were (same buffers, same shape: the two directions of one alignment, repeated calls); capturing and
    // instantiating it costs 0.9 ms
    hipGraph_t loop_graph_ = nullptr;
    hipGraphExec_t loop_exec_ = nullptr;
    std::vector<uint64_t> loop_sig_;
    void drop_loop_graph();
    int ensure(int slot, size_t bytes, std::string &err);
};

}  // namespace msspe
