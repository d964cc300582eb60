// od_msspe.cpp -- see od_msspe.hpp.  Mirrors /root/reference/od-msspe/src/main.rs, primer.rs,
// delta_g.rs and config.rs for the hot path and its immediate callers; all arithmetic of the hot
// path happens in libmsspe_hip.so.
#include "od_msspe.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <thread>

#include <fcntl.h>
#include <spawn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <unordered_set>

extern char **environ;

namespace od_msspe {

// ---------------------------------------------------------------------------------------------
// CLI (config.rs:11-148)
// ---------------------------------------------------------------------------------------------
namespace {

struct OptSpec {
    const char *flag, *env;
    enum Kind { Int, OptInt, Float, Bool, Str } kind;
    size_t offset;
};
#define OFF(f) offsetof(Args, f)
const OptSpec kOpts[] = {
    {"--kmer-size", "KMER_SIZE", OptSpec::Int, OFF(kmer_size)},
    {"--window-size", "WINDOW_SIZE", OptSpec::Int, OFF(window_size)},
    {"--overlap-size", "OVERLAP_SIZE", OptSpec::Int, OFF(overlap_size)},
    {"--max-mismatch-segments", "MAX_MISMATCH_SEGMENTS", OptSpec::OptInt, OFF(max_mismatch_segments)},
    {"--max-iterations", "MAX_ITERATIONS", OptSpec::Int, OFF(max_iterations)},
    {"--search-windows-size", "SEARCH_WINDOWS_SIZE", OptSpec::Int, OFF(search_windows_size)},
    {"--mv-conc", "MV_CONC", OptSpec::Float, OFF(mv_conc)},
    {"--dv-conc", "DV_CONC", OptSpec::Float, OFF(dv_conc)},
    {"--dntp-conc", "DNTP_CONC", OptSpec::Float, OFF(dntp_conc)},
    {"--dna-conc", "DNA_CONC", OptSpec::Float, OFF(dna_conc)},
    {"--annealing-temp", "ANNEALING_TEMP", OptSpec::Float, OFF(annealing_temp)},
    {"--min-tm", "MIN_TM", OptSpec::Float, OFF(min_tm)},
    {"--max-tm", "MAX_TM", OptSpec::Float, OFF(max_tm)},
    {"--max-self-dimer-any-tm", "MAX_SELF_DIMER_ANY_TM", OptSpec::Float, OFF(max_self_dimer_any_tm)},
    {"--max-self-dimer-end-tm", "MAX_SELF_DIMER_END_TM", OptSpec::Float, OFF(max_self_dimer_end_tm)},
    {"--max-hairpin-tm", "MAX_HAIRPIN_TM", OptSpec::Float, OFF(max_hairpin_tm)},
    {"--delta-g-threshold", "DELTA_G_THRESHOLD", OptSpec::Float, OFF(delta_g_threshold)},
    {"--keep-all", "KEEP_ALL", OptSpec::Bool, OFF(keep_all)},
    {"--check-cross-dimers", "CHECK_CROSS_DIMERS", OptSpec::Bool, OFF(check_cross_dimers)},
    {"--check-self-dimers", "CHECK_SELF_DIMERS", OptSpec::Bool, OFF(check_self_dimers)},
    {"--check-hairpin", "CHECK_HAIRPIN", OptSpec::Bool, OFF(check_hairpin)},
    {"--tm-stddev", "TM_STDDEV", OptSpec::Float, OFF(tm_stddev)},
    {"--disable-tm-stddev", "DISABLE_TM_STDDEV", OptSpec::Bool, OFF(disable_tm_stddev)},
    {"--disable-min-max-tm", "DISABLE_MIN_MAX_TM", OptSpec::Bool, OFF(disable_min_max_tm)},
    {"--do-align", "DO_ALIGN", OptSpec::Bool, OFF(do_align)},
    {"--ntthal", "NTTHAL", OptSpec::Str, OFF(ntthal)},
    {"--primer3", "PRIMER3", OptSpec::Str, OFF(primer3)},
    {"--params-path", "MSSPE_PARAMS_PATH", OptSpec::Str, OFF(params_path)},
    {"--device", "MSSPE_DEVICE", OptSpec::Int, OFF(device)},
    {"--devices", "MSSPE_DEVICES", OptSpec::Str, OFF(devices)},
};
#undef OFF

void assign(Args &a, const OptSpec &o, const std::string &v)
{
    char *base = reinterpret_cast<char *>(&a);
    char *end = nullptr;
    switch (o.kind) {
    case OptSpec::Int:
    case OptSpec::OptInt: {
        const long x = std::strtol(v.c_str(), &end, 10);
        if (v.empty() || *end || x < 0)
            throw UsageError(std::string("error: invalid value '") + v + "' for '" + o.flag + "'");
        *reinterpret_cast<int *>(base + o.offset) = (int)x;
        break;
    }
    case OptSpec::Float: {
        const float x = std::strtof(v.c_str(), &end);
        if (v.empty() || *end)
            throw UsageError(std::string("error: invalid value '") + v + "' for '" + o.flag + "'");
        *reinterpret_cast<float *>(base + o.offset) = x;
        break;
    }
    case OptSpec::Bool:
        if (v != "true" && v != "false")   // config.rs value_parser = ["true", "false"]
            throw UsageError(std::string("error: invalid value '") + v + "' for '" + o.flag +
                             " <...>'\n  [possible values: true, false]");
        *reinterpret_cast<std::string *>(base + o.offset) = v;
        break;
    case OptSpec::Str:
        *reinterpret_cast<std::string *>(base + o.offset) = v;
        break;
    }
}

}  // namespace

std::string Args::usage()
{
    std::string u = "Usage: od-msspe-hip --input <INPUT> --output <OUTPUT> [OPTIONS]\n\nOptions:\n"
                    "  -i, --input <INPUT>\n  -o, --output <OUTPUT>\n";
    for (const auto &o : kOpts) u += std::string("      ") + o.flag + " <...>  [env: " + o.env + "=]\n";
    u += "      --stddev-population   divide the Tm variance by n instead of n-1\n";
    return u;
}

Args Args::parse(int argc, const char *const *argv)
{
    Args a;
    for (const auto &o : kOpts)   // environment first, the command line overrides it (clap)
        if (const char *e = std::getenv(o.env))
            if (*e) assign(a, o, e);
    for (int i = 1; i < argc; ++i) {
        std::string tok = argv[i], val;
        bool has_val = false;
        const size_t eq = tok.find('=');
        if (tok.rfind("--", 0) == 0 && eq != std::string::npos) {
            val = tok.substr(eq + 1);
            tok = tok.substr(0, eq);
            has_val = true;
        }
        auto need = [&]() -> std::string {
            if (has_val) return val;
            if (i + 1 >= argc)
                throw UsageError("error: a value is required for '" + tok + " <...>' but none was supplied");
            return argv[++i];
        };
        if (tok == "-h" || tok == "--help") throw UsageError(usage());
        if (tok == "-i" || tok == "--input") { a.input = need(); continue; }
        if (tok == "-o" || tok == "--output") { a.output = need(); continue; }
        if (tok == "--stddev-population") { a.stddev_population = true; continue; }
        bool found = false;
        for (const auto &o : kOpts)
            if (tok == o.flag) {
                assign(a, o, need());
                found = true;
                break;
            }
        if (!found) throw UsageError("error: unexpected argument '" + tok + "' found\n\n" + usage());
    }
    if (a.input.empty() || a.output.empty())
        throw UsageError("error: the following required arguments were not provided:\n"
                         "  --input <INPUT>\n  --output <OUTPUT>\n\n" + usage());
    return a;
}

// ---------------------------------------------------------------------------------------------
// FASTA (main.rs:108-122; seq_io semantics: id = header up to the first space, lines joined)
// ---------------------------------------------------------------------------------------------
namespace {

// records of the byte range [p, end): one pass; sequence bytes go through a 256-entry table
// (upper case, U -> T).  Bytes before the first header of the range are ignored.
std::vector<SequenceRecord> parse_fasta_range(const char *p, const char *end)
{
    static const auto table = [] {
        std::array<char, 256> t{};
        for (int c = 0; c < 256; ++c) {
            char u = (char)std::toupper(c);
            t[(size_t)c] = u == 'U' ? 'T' : u;
        }
        return t;
    }();
    std::vector<SequenceRecord> out;
    size_t reserve_hint = 0;
    while (p < end) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
        const char *stop = nl ? nl : end;
        const char *last = stop;
        if (last > p && last[-1] == '\r') --last;
        if (last > p && *p == '>') {
            if (!out.empty()) reserve_hint = std::max(reserve_hint, out.back().sequence.size());
            const char *sp = static_cast<const char *>(std::memchr(p, ' ', (size_t)(last - p)));
            out.push_back({std::string(p + 1, sp ? sp : last), std::string()});
            out.back().sequence.reserve(reserve_hint);
        } else if (!out.empty()) {
            std::string &seq = out.back().sequence;
            const size_t at = seq.size();
            seq.resize(at + (size_t)(last - p));
            for (size_t q = 0; q < (size_t)(last - p); ++q) seq[at + q] = table[(unsigned char)p[q]];
        }
        p = nl ? nl + 1 : end;
    }
    return out;
}

}  // namespace

std::vector<SequenceRecord> to_records(const std::string &fasta)
{
    return to_records(fasta.data(), fasta.size());
}

std::vector<SequenceRecord> to_records(const char *base, size_t size)
{
    const char *end = base + size;
    // large inputs: cut at header lines and parse the pieces on the host's cores
    const size_t n_threads = std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), 16,
                                               size / (8u << 20)});
    if (n_threads < 2) return parse_fasta_range(base, end);
    std::vector<const char *> cut{base};
    for (size_t t = 1; t < n_threads; ++t) {
        const char *p = base + size / n_threads * t;
        const char *hit = nullptr;
        while (p < end) {   // next line that starts with '>'
            const char *nl = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
            if (!nl || nl + 1 >= end) break;
            if (nl[1] == '>') {
                hit = nl + 1;
                break;
            }
            p = nl + 1;
        }
        if (hit && hit > cut.back()) cut.push_back(hit);
    }
    cut.push_back(end);
    std::vector<std::vector<SequenceRecord>> parts(cut.size() - 1);
    std::vector<std::thread> pool;
    for (size_t t = 0; t + 1 < cut.size(); ++t)
        pool.emplace_back([&, t] { parts[t] = parse_fasta_range(cut[t], cut[t + 1]); });
    for (auto &th : pool) th.join();
    std::vector<SequenceRecord> out;
    size_t total = 0;
    for (const auto &v : parts) total += v.size();
    out.reserve(total);
    for (auto &v : parts)
        for (auto &r : v) out.push_back(std::move(r));
    return out;
}

std::string reverse_complement(const std::string &s)
{
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) switch (c) {
        case 'A': c = 'T'; break;
        case 'T': c = 'A'; break;
        case 'U': c = 'A'; break;
        case 'C': c = 'G'; break;
        case 'G': c = 'C'; break;
        default: break;
        }
    return r;
}

// ---------------------------------------------------------------------------------------------
// engine handle
// ---------------------------------------------------------------------------------------------
Engine::Engine(int device, const std::string &params_path)
{
    const int rc = msspe_create(device, params_path.empty() ? nullptr : params_path.c_str(), &ctx_);
    if (rc) {
        const std::string msg = ctx_ ? msspe_last_error(ctx_) : "allocation failed";
        if (ctx_) msspe_destroy(ctx_);
        ctx_ = nullptr;
        throw std::runtime_error("msspe_create: " + msg);
    }
}
Engine::Engine(const std::vector<int> &devices, const std::string &params_path)
{
    const int rc = msspe_group_create(devices.data(), (int)devices.size(), params_path.empty() ? nullptr : params_path.c_str(),
                                      nullptr, &group_);
    if (rc) {
        const std::string msg = group_ ? msspe_group_last_error(group_) : "allocation failed";
        if (group_) msspe_group_destroy(group_);
        group_ = nullptr;
        throw std::runtime_error("msspe_group_create: " + msg);
    }
    ctx_ = msspe_group_member(group_, 0);
}
Engine::~Engine()
{
    if (group_) msspe_group_destroy(group_);   // owns its members
    else if (ctx_) msspe_destroy(ctx_);
}
void Engine::fail_group(int rc) const
{
    throw std::runtime_error(std::string("libmsspe_hip status ") + std::to_string(rc) + ": " + msspe_group_last_error(group_));
}
void Engine::fail(int rc) const
{
    throw std::runtime_error(std::string("libmsspe_hip status ") + std::to_string(rc) + ": " +
                             msspe_last_error(ctx_));
}

// ---------------------------------------------------------------------------------------------
// stage A (main.rs:196-235, 331-406)
// ---------------------------------------------------------------------------------------------
DeviceAlignment::DeviceAlignment(Engine &eng, const std::vector<SequenceRecord> &records) : eng_(eng)
{
    // The device path takes one rectangular byte matrix.  Rows shorter than the longest are
    // padded with '-': partition j starts at the same column in every row (main.rs:173-181), pad
    // columns invalidate every k-mer that touches them (main.rs:167), so the extra all-pad
    // partitions of a short row hold no k-mers and can neither be counted nor covered -- the
    // winners are exactly those of the reference's per-record partitioning.
    for (const auto &r : records) len_ = std::max(len_, r.sequence.size());
    rows_ = (int)records.size();
    // rows go to the device through the library's pinned staging; no rectangular host copy
    std::vector<const char *> rows(records.size());
    std::vector<size_t> bytes(records.size());
    for (size_t r = 0; r < records.size(); ++r) {
        rows[r] = records[r].sequence.data();
        bytes[r] = records[r].sequence.size();
    }
    // packed on the device: 2-bit bases + validity bit (3/8 byte per column stay resident)
    const int rc = msspe_device_put_rows_packed(eng.ctx(), rows.data(), bytes.data(), rows_, len_, &dev_);
    if (rc) eng.fail(rc);
}

DeviceAlignment::~DeviceAlignment()
{
    if (dev_) (void)msspe_device_free(eng_.ctx(), dev_);
}

std::vector<KmerFrequency> find_candidates_kmers(Engine &eng, const DeviceAlignment &aln, uint8_t direction,
                                                 const ProgramConfig &cfg, int segment_size,
                                                 int overlap_size, int window_size)
{
    if (overlap_size < window_size)   // main.rs:201-203
        throw Panic("Overlap windows size must be greater or equal than search windows size");
    std::vector<KmerFrequency> out;
    if (aln.rows() == 0) return out;
    msspe_kmer_opt opt{segment_size, overlap_size, window_size, cfg.primer_config.kmer_size,
                       cfg.max_iterations, cfg.max_mismatch_segments};
    const int cap = std::max(1, cfg.max_iterations);
    std::vector<uint64_t> words((size_t)cap);
    std::vector<uint32_t> freq((size_t)cap);
    int n = 0;
    const int rc = msspe_kmer_candidates_packed_dev(eng.ctx(), aln.device(), aln.rows(), aln.length(), &opt, direction,
                                             words.data(), freq.data(), cap, &n);
    if (rc) eng.fail(rc);
    std::vector<char> buf((size_t)opt.kmer_size + 1);
    for (int i = 0; i < n; ++i) {
        msspe_unpack_oligo(words[(size_t)i], opt.kmer_size, buf.data());
        out.push_back({std::string(buf.data()), direction, freq[(size_t)i]});
    }
    return out;
}

// Both directions in one engine call (msspe_kmer_candidates_both_packed_dev: two streams, two host threads); what the
// reference gets from its two find_candidates_kmers calls, main.rs:673-690.
std::pair<std::vector<KmerFrequency>, std::vector<KmerFrequency>> find_candidates_kmers_both(
    Engine &eng, const DeviceAlignment &aln, const ProgramConfig &cfg, int segment_size, int overlap_size, int window_size)
{
    if (overlap_size < window_size)   // main.rs:201-203
        throw Panic("Overlap windows size must be greater or equal than search windows size");
    std::pair<std::vector<KmerFrequency>, std::vector<KmerFrequency>> out;
    if (aln.rows() == 0) return out;
    msspe_kmer_opt opt{segment_size, overlap_size, window_size, cfg.primer_config.kmer_size,
                       cfg.max_iterations, cfg.max_mismatch_segments};
    const int cap = std::max(1, cfg.max_iterations);
    std::vector<uint64_t> words[2] = {std::vector<uint64_t>((size_t)cap), std::vector<uint64_t>((size_t)cap)};
    std::vector<uint32_t> freq[2] = {std::vector<uint32_t>((size_t)cap), std::vector<uint32_t>((size_t)cap)};
    int n[2] = {0, 0};
    const int rc = msspe_kmer_candidates_both_packed_dev(eng.ctx(), aln.device(), aln.rows(), aln.length(), &opt,
                                                         words[0].data(), freq[0].data(), &n[0], words[1].data(),
                                                         freq[1].data(), &n[1], cap);
    if (rc) eng.fail(rc);
    std::vector<char> buf((size_t)opt.kmer_size + 1);
    for (int d = 0; d < 2; ++d) {
        auto &dst = d ? out.second : out.first;
        for (int i = 0; i < n[d]; ++i) {
            msspe_unpack_oligo(words[d][(size_t)i], opt.kmer_size, buf.data());
            dst.push_back({std::string(buf.data()), (uint8_t)(d ? SEQ_DIR_REV : SEQ_DIR_FWD), freq[d][(size_t)i]});
        }
    }
    return out;
}

std::vector<KmerFrequency> find_candidates_kmers(Engine &eng, const std::vector<SequenceRecord> &records,
                                                 uint8_t direction, const ProgramConfig &cfg,
                                                 int segment_size, int overlap_size, int window_size)
{
    if (overlap_size < window_size)   // main.rs:201-203
        throw Panic("Overlap windows size must be greater or equal than search windows size");
    if (records.empty()) return {};
    const DeviceAlignment aln(eng, records);
    return find_candidates_kmers(eng, aln, direction, cfg, segment_size, overlap_size, window_size);
}

// ---------------------------------------------------------------------------------------------
// stage B (primer.rs:143-166; main.rs:408-516)
// ---------------------------------------------------------------------------------------------
std::vector<PrimerInfo> check_primers(Engine &eng, const std::vector<std::string> &primers)
{
    std::vector<PrimerInfo> out;
    if (primers.empty()) return out;
    const int n = (int)primers.size(), k = (int)primers[0].size();
    std::string flat;
    for (const auto &p : primers) {
        if ((int)p.size() != k) throw std::runtime_error("primers of unequal length");
        flat += p;
    }
    std::vector<double> tm((size_t)n), gc((size_t)n), any((size_t)n), end((size_t)n), hp((size_t)n);
    msspe_chem chem;
    msspe_chem_primer3_defaults(&chem);   // primer.rs:125-140 sends only size / Tm bounds
    if (eng.group()) {   // oligos are independent: a slice per device
        const int rc = msspe_oligo_stats_group(eng.group(), flat.data(), n, k, &chem, tm.data(), gc.data(), any.data(),
                                               end.data(), hp.data());
        if (rc) eng.fail_group(rc);
    } else {
        const int rc = msspe_oligo_stats(eng.ctx(), flat.data(), n, k, &chem, tm.data(), gc.data(),
                                         any.data(), end.data(), hp.data());
        if (rc) eng.fail(rc);
    }
    for (int i = 0; i < n; ++i) {
        PrimerInfo p;
        p.id = primers[(size_t)i];
        p.tm = msspe_round_fixed_f32(tm[(size_t)i], 3);            // PRIMER_LEFT_0_TM=%.3f -> f32
        p.gc = msspe_round_fixed_f32(gc[(size_t)i], 3);
        p.self_any_th = msspe_round_fixed_f32(any[(size_t)i], 2);  // ..._TH=%.2f -> f32
        p.self_end_th = msspe_round_fixed_f32(end[(size_t)i], 2);
        p.hairpin_th = msspe_round_fixed_f32(hp[(size_t)i], 2);
        out.push_back(p);
    }
    return out;
}

void get_tm_stat(const std::vector<PrimerInfo> &info, bool population, float &mean, float &std)
{
    float sum = 0.0f;
    for (const auto &p : info) sum += p.tm;   // main.rs:464: sequential f32 sum
    mean = sum / (float)info.size();
    float acc = 0.0f;
    for (const auto &p : info) {
        const float d = p.tm - mean;
        acc += d * d;
    }
    const float div = population ? (float)info.size() : (float)(info.size() - 1);
    std = info.size() > (population ? 0u : 1u) ? std::sqrt(acc / div) : 0.0f;
}

bool tm_in_threshold(float tm, float mean, float std, float diff)
{
    return std::fabs(tm - mean) <= (diff * std);
}

bool is_run(const std::string &kmer)   // main.rs:478-490: only the trailing run counts
{
    int runs = 0;
    char last = ' ';
    for (char c : kmer) {
        if (c == last) runs += 1;
        else runs = 0;
        last = c;
    }
    return runs >= 5;
}

std::vector<KmerStat> get_kmer_stats(Engine &eng, const std::vector<KmerFrequency> &kmers,
                                     const ProgramConfig &cfg)
{
    std::vector<KmerStat> out;
    if (kmers.empty()) return out;
    std::vector<std::string> primers;
    for (const auto &k : kmers) primers.push_back(k.word);
    const auto info = check_primers(eng, primers);
    std::map<std::string, const PrimerInfo *> by_id;
    for (const auto &p : info) by_id.emplace(p.id, &p);   // first entry wins (or_insert)
    float mean, std;
    get_tm_stat(info, cfg.stddev_population, mean, std);
    for (const auto &k : kmers) {
        const PrimerInfo *p = by_id.at(k.word);
        out.push_back({k.word, k.direction, p->gc, mean, std, p->tm,
                       tm_in_threshold(p->tm, mean, std, cfg.tm_stddev), p->self_any_th,
                       p->self_end_th, p->hairpin_th, is_run(k.word)});
    }
    return out;
}

std::vector<KmerStat> filter_kmers(const std::vector<KmerStat> &stats, const ProgramConfig &cfg)
{
    const PrimerConfig &pc = cfg.primer_config;
    std::vector<KmerStat> out;
    for (const auto &s : stats) {
        const bool pass_self_any = !cfg.check_self_dimers || (s.self_any_th < pc.max_self_dimer_any_tm);
        const bool pass_self_end = !cfg.check_self_dimers || (s.self_end_th < pc.max_self_dimer_end_tm);
        const bool pass_hairpin = !cfg.check_hairpin || (s.hairpin_th < pc.max_hairpin_tm);
        const bool pass_min_max = cfg.disable_min_max_tm || (s.tm > pc.min_tm && s.tm < pc.max_tm);
        const bool pass_stddev = cfg.disable_tm_stddev || s.tm_ok;
        if (pass_self_any && pass_self_end && pass_hairpin && pass_min_max && pass_stddev && !s.runs)
            out.push_back(s);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// stage C (delta_g.rs:61-153) and the vertex cover (main.rs:754-798)
// ---------------------------------------------------------------------------------------------
// delta_g.rs:64-73: is the ordered pair (a, b) sent to ntthal at all?
bool ntthal_pair_sent(const std::string &a, const std::string &b, const ProgramConfig &cfg)
{
    if (!cfg.check_self_dimers && (a == b || reverse_complement(b) == a)) return false;
    return cfg.check_cross_dimers;
}

// delta_g.rs:61-81: every ordered pair that is sent, "a,b" per line, no trailing newline.  The engine
// takes the packed pool instead of this text; the function exists as the statement of WHICH pairs count
// (run_ntthal below applies the same predicate to the bitmap) and for the ntthal-compatible shim's tests.
std::string format_ntthal_input(const std::vector<std::string> &primers, const ProgramConfig &cfg)
{
    std::string out;
    for (const auto &a : primers)
        for (const auto &b : primers)
            if (ntthal_pair_sent(a, b, cfg)) out += a + "," + b + "\n";
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();   // .trim()
    return out;
}

ConflictGraph run_ntthal(Engine &eng, const std::vector<std::string> &primers,
                         const NtthalOptions &opts, const ProgramConfig &cfg)
{
    ConflictGraph g;
    // the reference's graph is keyed by the primer string: duplicates collapse into one node
    std::unordered_set<std::string> seen;
    for (const auto &p : primers)
        if (seen.insert(p).second) g.nodes.push_back(p);
    if (!cfg.check_cross_dimers || g.nodes.empty()) return g;   // delta_g.rs:71-73: no input at all
    const int n = (int)g.nodes.size(), k = (int)g.nodes[0].size();
    std::string flat;
    for (const auto &p : g.nodes) flat += p;
    auto two = [](float v) {   // ntthal receives "{:.2}" strings of the f32 options (delta_g.rs:98-106)
        char b[64];
        std::snprintf(b, sizeof b, "%.2f", (double)v);
        return std::strtod(b, nullptr);
    };
    msspe_chem chem{two(opts.mv), two(opts.dv), two(opts.dntp), two(opts.conc), two(opts.t), 30};
    // the conflict edges as a list (about 0.5 % of the ordered pairs at the default threshold), not the
    // dense n x n bitmap; if the first guess is too small the call says how many there are
    std::vector<msspe_edge> edges((size_t)std::max<uint64_t>(4096, (uint64_t)n * (uint64_t)n / 64));
    uint64_t count = 0;
    // (with --devices: the rows of the pair matrix dealt out over the group, the same edges in the same order)
    auto screen = [&]() {
        return eng.group() ? msspe_cross_dimer_edges_group(eng.group(), flat.data(), n, k, &chem, opts.dg, edges.data(),
                                                           edges.size(), &count)
                           : msspe_cross_dimer_edges(eng.ctx(), flat.data(), n, k, &chem, opts.dg, edges.data(), edges.size(),
                                                     &count);
    };
    int rc = screen();
    if (rc == MSSPE_ERR_CAPACITY) {
        edges.resize((size_t)count);
        rc = screen();
    }
    if (rc) eng.group() ? eng.fail_group(rc) : eng.fail(rc);
    for (uint64_t e = 0; e < count; ++e) {
        const std::string &a = g.nodes[edges[(size_t)e].a], &b = g.nodes[edges[(size_t)e].b];
        // delta_g.rs:66-69: pairs never sent to ntthal when self-dimer checking is off
        if (!ntthal_pair_sent(a, b, cfg)) continue;
        g.edges[a].insert(b);
    }
    return g;
}

std::set<std::string> vertex_cover(const std::vector<std::string> &primers, const ConflictGraph &g)
{
    // main.rs:754-771: symmetric adjacency over conflict edges, self loops included
    std::map<std::string, std::set<std::string>> conflicts;
    for (const auto &p : primers) {
        auto it = g.edges.find(p);
        if (it != g.edges.end())
            for (const auto &b : it->second) {
                conflicts[p].insert(b);
                conflicts[b].insert(p);
            }
    }
    // main.rs:776-798: repeatedly delete the primer with most live conflicts; ties go to the
    // lexicographically greatest string.  Same rule on integer ids (the map iterates in
    // lexicographic order, so a node's id is its rank) with live degrees kept up to date.
    std::vector<const std::string *> name;
    std::map<std::string, int> id;
    for (const auto &kv : conflicts) {
        id.emplace(kv.first, (int)name.size());
        name.push_back(&kv.first);
    }
    const int n = (int)name.size();
    std::vector<std::vector<int>> adj((size_t)n);
    std::vector<int> live((size_t)n, 0);
    std::vector<char> gone((size_t)n, 0);
    for (const auto &kv : conflicts) {
        auto &list = adj[(size_t)id[kv.first]];
        for (const auto &nb : kv.second) list.push_back(id[nb]);
        live[(size_t)id[kv.first]] = (int)list.size();
    }
    std::set<std::string> deleted;
    for (;;) {
        int worst = -1;
        for (int v = 0; v < n; ++v)   // ascending rank: ">=" keeps the greatest string among equals
            if (!gone[(size_t)v] && live[(size_t)v] > 0 && (worst < 0 || live[(size_t)v] >= live[(size_t)worst])) worst = v;
        if (worst < 0) break;
        gone[(size_t)worst] = 1;
        deleted.insert(*name[(size_t)worst]);
        for (int nb : adj[(size_t)worst]) --live[(size_t)nb];   // conflicts are symmetric
    }
    return deleted;
}

// ---------------------------------------------------------------------------------------------
// report + CSV (main.rs:518-594, 834-858): host-side text
// ---------------------------------------------------------------------------------------------
namespace {

std::string fmt(const char *f, double v)
{
    char b[64];
    std::snprintf(b, sizeof b, f, v);
    return b;
}

std::vector<uint64_t> pack_words(Engine &eng, const std::vector<KmerStat> &list, int k)
{
    std::vector<uint64_t> out(list.size());
    if (list.empty()) return out;
    std::string flat;
    for (const auto &p : list) flat += p.word;
    const int rc = msspe_pack_oligos(flat.data(), (int)list.size(), k, out.data());
    if (rc) eng.fail(rc);
    return out;
}

}  // namespace

std::string coverage_report(Engine &eng, const std::vector<KmerStat> &fwd, const std::vector<KmerStat> &rev,
                            const std::vector<SequenceRecord> &records, int segment_size,
                            int overlap_size, int window_size, int kmer_size)
{
    const DeviceAlignment aln(eng, records);
    return coverage_report(eng, aln, fwd, rev, records, segment_size, overlap_size, window_size, kmer_size);
}

std::string coverage_report(Engine &eng, const DeviceAlignment &aln, const std::vector<KmerStat> &fwd,
                            const std::vector<KmerStat> &rev, const std::vector<SequenceRecord> &records,
                            int segment_size, int overlap_size, int window_size, int kmer_size)
{
    // per-segment search on the device: hit[record * P + partition]
    const size_t L = aln.length();
    const size_t P = L < (size_t)segment_size ? 0 : (L - (size_t)segment_size) / (size_t)overlap_size + 1;
    std::vector<uint8_t> hit(records.size() * P + 1);
    if (P) {
        const auto wf = pack_words(eng, fwd, kmer_size), wr = pack_words(eng, rev, kmer_size);
        msspe_kmer_opt opt{segment_size, overlap_size, window_size, kmer_size, 0, 0};
        const int rc = msspe_segment_coverage_packed_dev(eng.ctx(), aln.device(), aln.rows(), L, &opt, wf.data(),
                                                  (int)wf.size(), wr.data(), (int)wr.size(), hit.data());
        if (rc) eng.fail(rc);
    }
    size_t total = 0, covered = 0;
    std::map<std::string, std::pair<size_t, size_t>> seq_stats;        // name -> (covered, total)
    std::map<uint16_t, std::pair<size_t, size_t>> partition_stats;
    for (size_t ri = 0; ri < records.size(); ++ri) {
        const auto &r = records[ri];
        const size_t len = r.sequence.size();
        auto &se = seq_stats[r.name];
        for (size_t j = 0; (size_t)segment_size <= len && j * (size_t)overlap_size + (size_t)segment_size <= len; ++j) {
            const bool h = hit[ri * P + j] != 0;
            auto &pe = partition_stats[(uint16_t)j];
            se.second += 1;
            pe.second += 1;
            total += 1;
            if (h) {
                se.first += 1;
                pe.first += 1;
                covered += 1;
            }
        }
    }
    float min_cov = INFINITY, max_cov = -INFINITY;
    size_t well = 0;
    for (const auto &kv : seq_stats) {
        const float c = (float)kv.second.first / (float)kv.second.second * 100.0f;
        min_cov = std::fmin(min_cov, c);
        max_cov = std::fmax(max_cov, c);
        if (c >= 80.0f) ++well;
    }
    std::string out = "\nCoverage report:\n";
    out += "  Segments:  " + std::to_string(covered) + "/" + std::to_string(total) + " covered (" +
           fmt("%.1f", (double)(100.0f * (float)covered / (float)total)) + "%)\n";
    out += "  Sequences: " + std::to_string(well) + "/" + std::to_string(seq_stats.size()) +
           " at \xe2\x89\xa5" "80% coverage (min " + fmt("%.1f", (double)min_cov) + "%, max " +
           fmt("%.1f", (double)max_cov) + "%)\n";
    std::string unc;
    for (const auto &kv : partition_stats)
        if (kv.second.first == 0) unc += (unc.empty() ? "" : ", ") + std::to_string(kv.first);
    if (unc.empty()) out += "  All partitions have primer coverage\n";
    else out += "  Uncovered partitions: [" + unc + "]\n";
    return out;
}

std::string primers_csv(const std::vector<KmerStat> &fwd, const std::vector<KmerStat> &rev)
{
    std::string out = "direction,name,primers,gc,avg,std,tm\n";
    for (const auto *list : {&fwd, &rev}) {
        size_t idx = 0;
        for (const auto &p : *list) {
            const char *d = p.direction == SEQ_DIR_FWD ? "F" : "R";
            out += std::string(d) + ",Primer_" + std::to_string(idx++) + "_" + d + "," + p.word + "," +
                   fmt("%.2f", (double)(p.gc_percent / 100.0f)) + "," + fmt("%.2f", (double)p.mean) + "," +
                   fmt("%.2f", (double)p.std) + "," + fmt("%.2f", (double)p.tm) + "\n";
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// main.rs:596-861
// ---------------------------------------------------------------------------------------------
namespace {
// MSSPE_HOST_TIMING=1: wall time of the pipeline phases on stderr (development aid)
struct PhaseTimer {
    bool on = std::getenv("MSSPE_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[od-msspe-hip] %-28s %8.1f ms\n", what,
                     std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
    ~PhaseTimer() { lap("teardown"); }   // declared first in run(): destroyed after everything else
};
}  // namespace

// main.rs:127-146 align_sequences(): `mafft --auto --quiet --thread -1 --op 1.53 --ep 0.123 --jtt 200 <file>`
// as a child process, its stdout is the aligned FASTA.  Like the reference (Command::output()) the exit
// status is not looked at and stderr is discarded; a mafft that cannot be started panics with
// std::process's message.  Runs before the engine is created: no process that holds a GPU spawns anything.
std::string align_sequences(const std::string &filepath)
{
    int fds[2];
    if (::pipe(fds) != 0) throw Panic("failed to execute MAFFT: cannot create a pipe");
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, fds[1], STDOUT_FILENO);
    posix_spawn_file_actions_addopen(&fa, STDERR_FILENO, "/dev/null", O_WRONLY, 0);
    posix_spawn_file_actions_addopen(&fa, STDIN_FILENO, "/dev/null", O_RDONLY, 0);
    posix_spawn_file_actions_addclose(&fa, fds[0]);
    posix_spawn_file_actions_addclose(&fa, fds[1]);
    const char *argv[] = {"mafft", "--auto", "--quiet", "--thread", "-1", "--op", "1.53", "--ep", "0.123",
                          "--jtt", "200", filepath.c_str(), nullptr};
    pid_t pid = 0;
    const int rc = ::posix_spawnp(&pid, "mafft", &fa, nullptr, const_cast<char *const *>(argv), environ);
    posix_spawn_file_actions_destroy(&fa);
    ::close(fds[1]);
    if (rc != 0) {
        ::close(fds[0]);
        throw Panic(std::string("failed to execute MAFFT: Os { code: ") + std::to_string(rc) + ", kind: " +
                    (rc == ENOENT ? "NotFound" : "Other") + ", message: \"" + std::strerror(rc) + "\" }");
    }
    std::string out;
    char buf[1 << 16];
    for (;;) {
        const ssize_t got = ::read(fds[0], buf, sizeof buf);
        if (got > 0) out.append(buf, (size_t)got);
        else if (got == 0 || errno != EINTR) break;
    }
    ::close(fds[0]);
    int status = 0;
    while (::waitpid(pid, &status, 0) < 0 && errno == EINTR) {
    }
    return out;
}

int run(const Args &args, std::string &stdout_text)
{
    PhaseTimer timer;
    std::vector<SequenceRecord> records;
    if (args.do_align == "true") {   // the reference's default (config.rs:131-138)
        const std::string aligned = align_sequences(args.input);
        records = to_records(aligned);
        timer.lap("mafft");
    } else {
        // regular files are mapped and parsed in place (no copy of a multi-hundred-MB input);
        // anything else (pipes) is read through a stream
        const int fd = ::open(args.input.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot read " + args.input);
        struct stat st;
        void *map = MAP_FAILED;
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0)
            map = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map != MAP_FAILED) {
            (void)::madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);
            records = to_records(static_cast<const char *>(map), (size_t)st.st_size);
            (void)::munmap(map, (size_t)st.st_size);
            ::close(fd);
        } else {
            ::close(fd);
            std::ifstream f(args.input, std::ios::binary);
            if (!f) throw std::runtime_error("cannot read " + args.input);
            std::ostringstream all;
            all << f.rdbuf();
            records = to_records(all.str());
        }
    }
    if (records.empty()) throw Panic("No sequences found in the input file");
    timer.lap("read + to_records");

    const int auto_mm = (int)std::min<size_t>(10, std::max<size_t>(1, (records.size() + 49) / 50));
    ProgramConfig cfg;
    cfg.max_iterations = args.max_iterations;
    cfg.max_mismatch_segments = args.max_mismatch_segments >= 0 ? args.max_mismatch_segments : auto_mm;
    cfg.keep_all = args.keep_all == "true";
    cfg.check_cross_dimers = args.check_cross_dimers == "true";
    cfg.check_self_dimers = args.check_self_dimers == "true";
    cfg.check_hairpin = args.check_hairpin == "true";
    cfg.tm_stddev = args.tm_stddev;
    cfg.disable_tm_stddev = args.disable_tm_stddev == "true";
    cfg.disable_min_max_tm = args.disable_min_max_tm == "true";
    cfg.primer_config = {args.kmer_size, args.min_tm, args.max_tm, args.max_self_dimer_any_tm,
                         args.max_self_dimer_end_tm, args.max_hairpin_tm};
    cfg.stddev_population = args.stddev_population;

    std::vector<int> devices;   // --devices 0,1,...: a group; otherwise the one context of --device
    for (size_t at = 0; at < args.devices.size();) {
        size_t comma = args.devices.find(',', at);
        if (comma == std::string::npos) comma = args.devices.size();
        const std::string tok = args.devices.substr(at, comma - at);
        char *end = nullptr;
        const long d = std::strtol(tok.c_str(), &end, 10);
        if (tok.empty() || *end || d < 0) throw UsageError("error: invalid value '" + args.devices + "' for '--devices'");
        devices.push_back((int)d);
        at = comma + 1;
    }
    std::unique_ptr<Engine> eng_owner(devices.empty() ? new Engine(args.device, args.params_path)
                                                      : new Engine(devices, args.params_path));
    Engine &eng = *eng_owner;
    const DeviceAlignment aln(eng, records);   // one upload for stage A (both directions) and the report
    timer.lap("engine + alignment upload");
    const auto cand_both = find_candidates_kmers_both(eng, aln, cfg, args.window_size, args.overlap_size,
                                                      args.search_windows_size);
    const auto &cand_f = cand_both.first, &cand_r = cand_both.second;
    timer.lap("stage A (both directions)");
    const auto stats_f = get_kmer_stats(eng, cand_f, cfg);
    const auto stats_r = get_kmer_stats(eng, cand_r, cfg);
    const auto prim_f = cfg.keep_all ? stats_f : filter_kmers(stats_f, cfg);
    const auto prim_r = cfg.keep_all ? stats_r : filter_kmers(stats_r, cfg);

    timer.lap("stage B + filter");
    std::vector<std::string> primers;
    for (const auto &s : prim_f) primers.push_back(s.word);
    for (const auto &s : prim_r) primers.push_back(s.word);
    const NtthalOptions opts{args.mv_conc, args.dv_conc, args.dntp_conc, args.dna_conc,
                             args.annealing_temp, args.delta_g_threshold};
    const ConflictGraph graph = run_ntthal(eng, primers, opts, cfg);
    const auto deleted = vertex_cover(primers, graph);
    std::vector<KmerStat> good_f, good_r;
    for (const auto &p : prim_f)
        if (cfg.keep_all || !deleted.count(p.word)) good_f.push_back(p);
    for (const auto &p : prim_r)
        if (cfg.keep_all || !deleted.count(p.word)) good_r.push_back(p);

    timer.lap("stage C + vertex cover");
    stdout_text = coverage_report(eng, aln, good_f, good_r, records, args.window_size, args.overlap_size,
                                  args.search_windows_size, args.kmer_size);
    std::ofstream out(args.output, std::ios::binary);
    if (!out) throw std::runtime_error("cannot write " + args.output);
    out << primers_csv(good_f, good_r);
    timer.lap("coverage report + csv");
    return 0;
}

}  // namespace od_msspe
