// ntthal-hip: speaks the `ntthal` command-line / pipe protocol od-msspe uses
// (/root/reference/od-msspe/src/delta_g.rs:93-145) on top of libmsspe_hip.so, so that an UNMODIFIED
// od-msspe binary can be pointed at the GPU engine with `--ntthal /path/to/ntthal-hip`.
//   argv : -a ANY|END1 -mv <mM> -dv <mM> -n <mM> -d <nM> -t <C> -maxloop <n> -path <dir>/ -i
//          (or -s1 <seq> -s2 <seq> instead of -i)
//   stdin: one "SEQ1,SEQ2" per line
//   stdout per line: "Calculated thermodynamical parameters for dimer:\tdS = %g\tdH = %g\tdG = %g\tt = %g"
//          and the four SEQ/SEQ/STR/STR alignment rows (Primer3 2.6.1 thal.c drawDimer).  Like
//          ntthal, nothing is printed for a pair without any structure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "../../include/msspe_hip.h"

namespace {

std::string reversed(const std::string &s) { return std::string(s.rbegin(), s.rend()); }

void draw_dimer(const std::string &o1, const std::string &o2_5to3, const uint8_t *ps1, const uint8_t *ps2)
{
    const std::string o2 = reversed(o2_5to3);
    const int len1 = (int)o1.size(), len2 = (int)o2.size();
    std::string d[4];
    int n1 = 0, n2 = 0;
    while (n1 < len1 && ps1[n1] == 0) ++n1;
    while (n2 < len2 && ps2[n2] == 0) ++n2;
    if (n1 >= n2) {
        d[0] += o1.substr(0, (size_t)n1);
        d[1].append((size_t)n1, ' ');
        d[2].append((size_t)n1, ' ');
        d[3].append((size_t)(n1 - n2), ' ');
        d[3] += o2.substr(0, (size_t)n2);
    } else {
        d[3] += o2.substr(0, (size_t)n2);
        d[1].append((size_t)n2, ' ');
        d[2].append((size_t)n2, ' ');
        d[0].append((size_t)(n2 - n1), ' ');
        d[0] += o1.substr(0, (size_t)n1);
    }
    int i = n1 + 1, j = n2 + 1;
    while (i <= len1) {
        while (i <= len1 && ps1[i - 1] != 0 && j <= len2 && ps2[j - 1] != 0) {
            d[0] += ' ';
            d[1] += o1[(size_t)i - 1];
            d[2] += o2[(size_t)j - 1];
            d[3] += ' ';
            ++i;
            ++j;
        }
        int s1 = 0, s2 = 0;
        while (i <= len1 && ps1[i - 1] == 0) {
            d[0] += o1[(size_t)i - 1];
            d[1] += ' ';
            ++s1;
            ++i;
        }
        while (j <= len2 && ps2[j - 1] == 0) {
            d[2] += ' ';
            d[3] += o2[(size_t)j - 1];
            ++s2;
            ++j;
        }
        if (s1 < s2) {
            d[0].append((size_t)(s2 - s1), '-');
            d[1].append((size_t)(s2 - s1), ' ');
        } else if (s1 > s2) {
            d[2].append((size_t)(s1 - s2), ' ');
            d[3].append((size_t)(s1 - s2), '-');
        }
    }
    std::printf("SEQ\t%s\nSEQ\t%s\nSTR\t%s\nSTR\t%s\n", d[0].c_str(), d[1].c_str(), d[2].c_str(), d[3].c_str());
}

}  // namespace

int main(int argc, char **argv)
{
    msspe_chem chem;
    msspe_chem_ntthal_defaults(&chem);
    chem.temp_c = 37.0;   // ntthal's own default; od-msspe always passes -t
    int mode = 1;
    bool interactive = false;
    std::string path, s1, s2;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "ntthal-hip: option %s needs a value\n", a.c_str());
                std::exit(1);
            }
            return argv[++i];
        };
        if (a == "-a") {
            const std::string m = val();
            if (m == "ANY") mode = 1;
            else if (m == "END1") mode = 2;
            else {
                std::fprintf(stderr, "ntthal-hip: alignment type %s is not supported (ANY, END1)\n", m.c_str());
                return 1;
            }
        } else if (a == "-mv") chem.mv = std::atof(val());
        else if (a == "-dv") chem.dv = std::atof(val());
        else if (a == "-n") chem.dntp = std::atof(val());
        else if (a == "-d") chem.dna_conc = std::atof(val());
        else if (a == "-t") chem.temp_c = std::atof(val());
        else if (a == "-maxloop") chem.max_loop = std::atoi(val());
        else if (a == "-path") path = val();
        else if (a == "-s1") s1 = val();
        else if (a == "-s2") s2 = val();
        else if (a == "-i") interactive = true;
        else if (a == "-r") { /* "only Tm" flag of ntthal: accepted, full output is printed anyway */ }
        else {
            std::fprintf(stderr, "ntthal-hip: unknown option %s\n", a.c_str());
            return 1;
        }
    }
    std::vector<std::pair<std::string, std::string>> pairs;
    if (interactive) {
        std::string line;
        while (std::getline(std::cin, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            const size_t c = line.find(',');
            if (c == std::string::npos) {
                std::fprintf(stderr, "ntthal-hip: expected SEQ1,SEQ2, got '%s'\n", line.c_str());
                return 1;
            }
            pairs.emplace_back(line.substr(0, c), line.substr(c + 1));
        }
    } else if (!s1.empty() && !s2.empty()) {
        pairs.emplace_back(s1, s2);
    } else {
        std::fprintf(stderr, "ntthal-hip: use -i (pairs on stdin) or -s1 SEQ -s2 SEQ\n");
        return 1;
    }
    const char *dev_env = std::getenv("MSSPE_DEVICE");
    msspe_ctx *ctx = nullptr;
    int rc = msspe_create(dev_env ? std::atoi(dev_env) : 0, path.empty() ? nullptr : path.c_str(), &ctx);
    if (rc) {
        std::fprintf(stderr, "ntthal-hip: %s\n", ctx ? msspe_last_error(ctx) : "allocation failed");
        if (ctx) msspe_destroy(ctx);
        return 1;
    }
    // group by oligo length (od-msspe sends equal-length k-mers); results keep the input order
    std::map<size_t, std::vector<size_t>> by_len;
    for (size_t p = 0; p < pairs.size(); ++p) {
        if (pairs[p].first.size() != pairs[p].second.size()) {
            std::fprintf(stderr, "ntthal-hip: oligos of one pair must have equal length\n");
            msspe_destroy(ctx);
            return 1;
        }
        by_len[pairs[p].first.size()].push_back(p);
    }
    std::vector<msspe_thal_detail> det(pairs.size());
    for (auto &kv : by_len) {
        std::string A, B;
        for (size_t p : kv.second) {
            A += pairs[p].first;
            B += pairs[p].second;
        }
        std::vector<msspe_thal_detail> part(kv.second.size());
        rc = msspe_thal_detail_pairs(ctx, A.c_str(), B.c_str(), (int)kv.second.size(), (int)kv.first, &chem,
                                     mode, part.data());
        if (rc) {
            std::fprintf(stderr, "ntthal-hip: %s\n", msspe_last_error(ctx));
            msspe_destroy(ctx);
            return 1;
        }
        for (size_t q = 0; q < kv.second.size(); ++q) det[kv.second[q]] = part[q];
    }
    for (size_t p = 0; p < pairs.size(); ++p) {
        const msspe_thal_detail &r = det[p];
        if (r.no_structure) continue;   // ntthal prints nothing to stdout in this case
        std::printf("Calculated thermodynamical parameters for dimer:\tdS = %g\tdH = %g\tdG = %g\tt = %g\n",
                    r.dS, r.dH, r.dG, r.t);
        draw_dimer(pairs[p].first, pairs[p].second, r.ps1, r.ps2);
    }
    msspe_destroy(ctx);
    return 0;
}
