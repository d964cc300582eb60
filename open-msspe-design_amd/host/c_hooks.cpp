// c_hooks.cpp -- extern "C" test hooks over the host layer (used by tests/ through ctypes).
#include <cstring>
#include <sstream>

#include "od_msspe.hpp"

using namespace od_msspe;

namespace {
int emit(const std::string &s, char *out, size_t cap)
{
    if (s.size() + 1 > cap) {   // -3: the caller's buffer is too small (and says so: it is left an empty string)
        if (cap) out[0] = '\0';
        return -3;
    }
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
std::vector<std::string> lines(const char *text)
{
    std::vector<std::string> v;
    std::istringstream in(text ? text : "");
    std::string l;
    while (std::getline(in, l))
        if (!l.empty()) v.push_back(l);
    return v;
}
}  // namespace

extern "C" {

// returns 0 ok, 2 usage error, 101 panic, 1 other; message / report text in `out`
int odm_run_cli(int argc, const char *const *argv, char *out, size_t cap)
{
    try {
        const Args a = Args::parse(argc, argv);
        std::string report;
        const int rc = run(a, report);
        emit(report, out, cap);
        return rc;
    } catch (const UsageError &e) {
        emit(e.what(), out, cap);
        return 2;
    } catch (const Panic &e) {
        emit(e.what(), out, cap);
        return 101;
    } catch (const std::exception &e) {
        emit(e.what(), out, cap);
        return 1;
    }
}

// parse only: writes "key=value" lines of the resolved options
int odm_parse_args(int argc, const char *const *argv, char *out, size_t cap)
{
    try {
        const Args a = Args::parse(argc, argv);
        std::ostringstream o;
        o << "input=" << a.input << "\noutput=" << a.output << "\nkmer_size=" << a.kmer_size
          << "\nwindow_size=" << a.window_size << "\noverlap_size=" << a.overlap_size
          << "\nmax_mismatch_segments=" << a.max_mismatch_segments << "\nmax_iterations=" << a.max_iterations
          << "\nsearch_windows_size=" << a.search_windows_size << "\nmv_conc=" << a.mv_conc
          << "\ndelta_g_threshold=" << a.delta_g_threshold << "\nkeep_all=" << a.keep_all
          << "\ncheck_hairpin=" << a.check_hairpin << "\ndo_align=" << a.do_align
          << "\ntm_stddev=" << a.tm_stddev << "\n";
        emit(o.str(), out, cap);
        return 0;
    } catch (const UsageError &e) {
        emit(e.what(), out, cap);
        return 2;
    }
}

// primers: one per line; edges: "a,b" per line (directed conflict a -> b).  Output: deleted
// primers, one per line, sorted.
int odm_vertex_cover(const char *primers_nl, const char *edges_nl, char *out, size_t cap)
{
    ConflictGraph g;
    const auto primers = lines(primers_nl);
    g.nodes = primers;
    for (const auto &e : lines(edges_nl)) {
        const size_t c = e.find(',');
        g.edges[e.substr(0, c)].insert(e.substr(c + 1));
    }
    std::string s;
    for (const auto &d : vertex_cover(primers, g)) s += d + "\n";
    return emit(s, out, cap);
}

int odm_is_run(const char *kmer) { return is_run(kmer) ? 1 : 0; }

// primers: one per line -> the text the reference would pipe into ntthal (delta_g.rs:61-81)
int odm_format_ntthal_input(const char *primers_nl, int check_cross_dimers, int check_self_dimers, char *out,
                            size_t cap)
{
    ProgramConfig cfg{};
    cfg.check_cross_dimers = check_cross_dimers != 0;
    cfg.check_self_dimers = check_self_dimers != 0;
    return emit(format_ntthal_input(lines(primers_nl), cfg), out, cap);
}

// FASTA text -> "name\tsequence" lines
int odm_to_records(const char *fasta, char *out, size_t cap)
{
    std::string s;
    for (const auto &r : to_records(fasta)) s += r.name + "\t" + r.sequence + "\n";
    return emit(s, out, cap);
}

// rows: "word,direction,gc,mean,std,tm" per line, forward rows first
int odm_primers_csv(const char *rows_nl, char *out, size_t cap)
{
    std::vector<KmerStat> f, r;
    for (const auto &l : lines(rows_nl)) {
        std::istringstream in(l);
        std::string w, d, gc, mean, sd, tm;
        std::getline(in, w, ','); std::getline(in, d, ','); std::getline(in, gc, ',');
        std::getline(in, mean, ','); std::getline(in, sd, ','); std::getline(in, tm, ',');
        KmerStat k{w, (uint8_t)std::stoi(d), std::stof(gc), std::stof(mean), std::stof(sd), std::stof(tm),
                   true, 0, 0, 0, false};
        (k.direction == SEQ_DIR_FWD ? f : r).push_back(k);
    }
    return emit(primers_csv(f, r), out, cap);
}

// records: "name\tsequence" lines; fwd / rev: selected primer words, one per line.  The per-segment
// search runs on the device (there is no CPU version of it in the product).
int odm_coverage_report(const char *records_nl, const char *fwd_nl, const char *rev_nl, int segment,
                        int stride, int window, int k, char *out, size_t cap)
{
    std::vector<SequenceRecord> recs;
    for (const auto &l : lines(records_nl)) {
        const size_t t = l.find('\t');
        recs.push_back({l.substr(0, t), l.substr(t + 1)});
    }
    std::vector<KmerStat> f, r;
    for (const auto &w : lines(fwd_nl)) f.push_back({w, SEQ_DIR_FWD, 0, 0, 0, 0, true, 0, 0, 0, false});
    for (const auto &w : lines(rev_nl)) r.push_back({w, SEQ_DIR_REV, 0, 0, 0, 0, true, 0, 0, 0, false});
    try {
        Engine eng(0, "");
        return emit(coverage_report(eng, f, r, recs, segment, stride, window, k), out, cap);
    } catch (const std::exception &e) {
        emit(e.what(), out, cap);
        return -2;   // no usable GPU: the text is the reason
    }
}

float odm_tm_stat(const float *tm, int n, int population, float *std_out)
{
    std::vector<PrimerInfo> v((size_t)n);
    for (int i = 0; i < n; ++i) v[(size_t)i].tm = tm[i];
    float mean, sd;
    get_tm_stat(v, population != 0, mean, sd);
    *std_out = sd;
    return mean;
}

}  // extern "C"
