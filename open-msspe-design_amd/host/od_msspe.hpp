// od_msspe.hpp -- C++ host side above the C ABI, mirroring the reference's (Rust) interface for
// the hot path and the rows SURVEY.md 8f marks "next": same names, argument meaning and error
// behaviour as /root/reference/od-msspe/src/{main,primer,delta_g,config,constants}.rs.
// Everything thermodynamic or k-mer related is computed by libmsspe_hip.so (no CPU fallback);
// what stays on the host is what the north star leaves on the host: FASTA I/O, the outer
// pipeline, the (tiny, sequential) vertex cover, the report and the CSV.
#pragma once

#include <cstdint>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/msspe_hip.h"

namespace od_msspe {

// constants.rs:1-26
constexpr int KMER_SIZE = 13, WINDOW_SIZE = 500, OVERLAP_SIZE = 250, MAX_ITERATIONS = 1000,
              SEARCH_WINDOWS_SIZE = 50;
constexpr float MV_CONC = 50.0f, DV_CONC = 3.0f, DNTP_CONC = 0.0f, DNA_CONC = 250.0f,
                ANNEALING_TEMP = 25.0f, PRIMER_MIN_TM = 30.0f, PRIMER_MAX_TM = 60.0f,
                PRIMER_MAX_SELF_ANY_TH = 47.0f, PRIMER_MAX_SELF_END_TH = 47.0f,
                PRIMER_MAX_HAIRPIN_TH = 24.0f, DELTA_G_THRESHOLD = -9000.0f;
constexpr uint8_t SEQ_DIR_FWD = 0, SEQ_DIR_REV = 1;

struct UsageError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct Panic : std::runtime_error {   // where the reference panics
    using std::runtime_error::runtime_error;
};

// config.rs:11-148 (clap Args; every option except -i/-o also reads an environment variable)
struct Args {
    std::string input, output;
    int kmer_size = KMER_SIZE, window_size = WINDOW_SIZE, overlap_size = OVERLAP_SIZE;
    int max_mismatch_segments = -1;   // Option<usize>: -1 = not given
    int max_iterations = MAX_ITERATIONS, search_windows_size = SEARCH_WINDOWS_SIZE;
    float mv_conc = MV_CONC, dv_conc = DV_CONC, dntp_conc = DNTP_CONC, dna_conc = DNA_CONC,
          annealing_temp = ANNEALING_TEMP, min_tm = PRIMER_MIN_TM, max_tm = PRIMER_MAX_TM,
          max_self_dimer_any_tm = PRIMER_MAX_SELF_ANY_TH, max_self_dimer_end_tm = PRIMER_MAX_SELF_END_TH,
          max_hairpin_tm = PRIMER_MAX_HAIRPIN_TH, delta_g_threshold = DELTA_G_THRESHOLD;
    std::string keep_all = "false", check_cross_dimers = "true", check_self_dimers = "true",
                check_hairpin = "true", disable_tm_stddev = "false", disable_min_max_tm = "false",
                do_align = "true";
    float tm_stddev = 2.0f;
    std::string ntthal = "ntthal", primer3 = "primer3_core";   // accepted for compatibility, unused
    // engine-only switches (not in the reference)
    int device = 0;
    std::string devices;           // "0,1,2,...": the N^2 pair loop and stage B run on a group of devices (msspe_group_*)
    std::string params_path;       // Primer3 config directory; empty = bundled tables
    bool stddev_population = false;  // crate std-dev 0.1.0's divisor is unpinned
    static Args parse(int argc, const char *const *argv);   // throws UsageError
    static std::string usage();
};

struct PrimerConfig {   // config.rs:150-158
    int kmer_size;
    float min_tm, max_tm, max_self_dimer_any_tm, max_self_dimer_end_tm, max_hairpin_tm;
};

struct ProgramConfig {   // config.rs:160-177
    int max_iterations, max_mismatch_segments;
    bool keep_all, check_cross_dimers, check_self_dimers, check_hairpin;
    float tm_stddev;
    bool disable_tm_stddev, disable_min_max_tm;
    PrimerConfig primer_config;
    bool stddev_population;
};

struct SequenceRecord {   // main.rs:21-24
    std::string name, sequence;
};
std::vector<SequenceRecord> to_records(const std::string &fasta);   // main.rs:108-122
std::vector<SequenceRecord> to_records(const char *fasta, size_t size);
std::string reverse_complement(const std::string &s);               // main.rs:148-161
bool ntthal_pair_sent(const std::string &a, const std::string &b, const ProgramConfig &cfg);   // delta_g.rs:64-73
std::string format_ntthal_input(const std::vector<std::string> &primers, const ProgramConfig &cfg);   // delta_g.rs:61-81

struct KmerFrequency {   // main.rs:47-51
    std::string word;
    uint8_t direction;
    size_t frequency;
};

struct PrimerInfo {   // primer.rs:8-15 (values as od-msspe reads them back: text -> f32)
    std::string id;
    float tm = 0, gc = 0, self_any_th = 0, self_end_th = 0, hairpin_th = 0;
};

struct KmerStat {   // main.rs:67-80
    std::string word;
    uint8_t direction;
    float gc_percent, mean, std, tm;
    bool tm_ok;
    float self_any_th, self_end_th, hairpin_th;
    bool runs;
};

struct NtthalOptions {   // delta_g.rs:18-25
    float mv, dv, dntp, conc, t, dg;
};

// Conflict relation produced by the cross-dimer stage: the reference's string-keyed GraphDB
// (graphdb.rs) restricted to what main.rs:754-771 consumes.
struct ConflictGraph {
    std::vector<std::string> nodes;                       // unique primer words, first-seen order
    std::map<std::string, std::set<std::string>> edges;   // directed: a -> {b : dG(a,b) < threshold}
};

class Engine {   // owns one msspe_ctx, or a group of them (one per device of --devices)
public:
    Engine(int device, const std::string &params_path);
    Engine(const std::vector<int> &devices, const std::string &params_path);
    ~Engine();
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    msspe_ctx *ctx() const { return ctx_; }         // with a group: member 0 (stage A, the coverage report)
    msspe_group *group() const { return group_; }   // nullptr: one device
    [[noreturn]] void fail(int rc) const;
    [[noreturn]] void fail_group(int rc) const;

private:
    msspe_ctx *ctx_ = nullptr;
    msspe_group *group_ = nullptr;
};

// The alignment as the device sees it: one rectangular byte matrix (rows shorter than the longest
// padded with '-'), copied to the GPU once and read by both directions of stage A and by the
// coverage report.
class DeviceAlignment {
public:
    DeviceAlignment(Engine &eng, const std::vector<SequenceRecord> &records);
    ~DeviceAlignment();
    DeviceAlignment(const DeviceAlignment &) = delete;
    DeviceAlignment &operator=(const DeviceAlignment &) = delete;
    const uint64_t *device() const { return static_cast<const uint64_t *>(dev_); }   // packed rows (msspe_device_put_rows_packed)
    int rows() const { return rows_; }
    size_t length() const { return len_; }

private:
    Engine &eng_;
    void *dev_ = nullptr;
    int rows_ = 0;
    size_t len_ = 0;
};

// main.rs:331-406 (+ :196-255): winners of one direction, in selection order
std::vector<KmerFrequency> find_candidates_kmers(Engine &eng, const std::vector<SequenceRecord> &records,
                                                 uint8_t direction, const ProgramConfig &cfg,
                                                 int segment_size, int overlap_size, int window_size);
std::pair<std::vector<KmerFrequency>, std::vector<KmerFrequency>> find_candidates_kmers_both(
    Engine &eng, const DeviceAlignment &aln, const ProgramConfig &cfg, int segment_size, int overlap_size, int window_size);
std::vector<KmerFrequency> find_candidates_kmers(Engine &eng, const DeviceAlignment &aln, uint8_t direction,
                                                 const ProgramConfig &cfg, int segment_size,
                                                 int overlap_size, int window_size);
// primer.rs:143-166
std::vector<PrimerInfo> check_primers(Engine &eng, const std::vector<std::string> &primers);
// main.rs:408-455, :462-471, :478-490, :492-516
std::vector<KmerStat> get_kmer_stats(Engine &eng, const std::vector<KmerFrequency> &kmers,
                                     const ProgramConfig &cfg);
void get_tm_stat(const std::vector<PrimerInfo> &info, bool population, float &mean, float &std);
bool tm_in_threshold(float tm, float mean, float std, float diff);
bool is_run(const std::string &kmer);
std::vector<KmerStat> filter_kmers(const std::vector<KmerStat> &stats, const ProgramConfig &cfg);
// delta_g.rs:61-153
ConflictGraph run_ntthal(Engine &eng, const std::vector<std::string> &primers,
                         const NtthalOptions &opts, const ProgramConfig &cfg);
// main.rs:754-798: primers removed by the greedy vertex cover
std::set<std::string> vertex_cover(const std::vector<std::string> &primers, const ConflictGraph &g);
// main.rs:518-594 (text goes to `out`): the per-segment search runs on the device
// (msspe_segment_coverage_dev), the totals per sequence / partition and the text on the host
std::string coverage_report(Engine &eng, const DeviceAlignment &aln, const std::vector<KmerStat> &fwd,
                            const std::vector<KmerStat> &rev, const std::vector<SequenceRecord> &records,
                            int segment_size, int overlap_size, int window_size, int kmer_size);
std::string coverage_report(Engine &eng, const std::vector<KmerStat> &fwd, const std::vector<KmerStat> &rev,
                            const std::vector<SequenceRecord> &records, int segment_size,
                            int overlap_size, int window_size, int kmer_size);
// main.rs:834-858
std::string primers_csv(const std::vector<KmerStat> &fwd, const std::vector<KmerStat> &rev);

// main.rs:596-861 without the MAFFT call: returns the process exit code; report -> stdout
int run(const Args &args, std::string &stdout_text);

}  // namespace od_msspe
