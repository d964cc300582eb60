// primer3_core-hip: speaks the slice of primer3_core's Boulder-IO protocol od-msspe uses
// (/root/reference/od-msspe/src/primer.rs:125-166: task check_primers, one SEQUENCE_PRIMER per
// record) on top of libmsspe_hip.so, so that an UNMODIFIED od-msspe binary can be pointed at the
// GPU engine with `--primer3 /path/to/primer3_core-hip`.
// For every record the input tags are echoed and the tags od-msspe reads back are printed:
//   PRIMER_LEFT_0_TM (%.3f), _GC_PERCENT (%.3f), _SELF_ANY_TH, _SELF_END_TH, _HAIRPIN_TH (%.2f).
// As in primer3_core, chemistry comes from the record (PRIMER_SALT_MONOVALENT, PRIMER_SALT_DIVALENT,
// PRIMER_DNTP_CONC, PRIMER_DNA_CONC) or Primer3's defaults 50 / 1.5 / 0.6 / 50.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/msspe_hip.h"

struct Record {
    std::vector<std::pair<std::string, std::string>> tags;
    std::string primer;
    msspe_chem chem;
};

int main()
{
    std::vector<Record> recs;
    Record cur;
    msspe_chem_primer3_defaults(&cur.chem);
    std::string line;
    bool any = false;
    while (std::getline(std::cin, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line == "=") {
            recs.push_back(cur);
            cur = Record();
            msspe_chem_primer3_defaults(&cur.chem);
            any = false;
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        const std::string key = line.substr(0, eq), val = line.substr(eq + 1);
        cur.tags.emplace_back(key, val);
        any = true;
        if (key == "SEQUENCE_PRIMER") cur.primer = val;
        else if (key == "PRIMER_SALT_MONOVALENT") cur.chem.mv = std::atof(val.c_str());
        else if (key == "PRIMER_SALT_DIVALENT") cur.chem.dv = std::atof(val.c_str());
        else if (key == "PRIMER_DNTP_CONC") cur.chem.dntp = std::atof(val.c_str());
        else if (key == "PRIMER_DNA_CONC") cur.chem.dna_conc = std::atof(val.c_str());
    }
    if (any) recs.push_back(cur);
    if (recs.empty()) return 0;
    const char *dev_env = std::getenv("MSSPE_DEVICE");
    const char *params = std::getenv("MSSPE_PARAMS_PATH");
    msspe_ctx *ctx = nullptr;
    int rc = msspe_create(dev_env ? std::atoi(dev_env) : 0, params, &ctx);
    if (rc) {
        std::fprintf(stderr, "primer3_core-hip: %s\n", ctx ? msspe_last_error(ctx) : "allocation failed");
        if (ctx) msspe_destroy(ctx);
        return 1;
    }
    for (const Record &r : recs) {
        for (const auto &t : r.tags) std::printf("%s=%s\n", t.first.c_str(), t.second.c_str());
        if (r.primer.empty()) {
            std::printf("PRIMER_ERROR=Missing SEQUENCE_PRIMER\n=\n");
            continue;
        }
        double tm, gc, sa, se, hp;
        rc = msspe_oligo_stats(ctx, r.primer.c_str(), 1, (int)r.primer.size(), &r.chem, &tm, &gc, &sa, &se, &hp);
        if (rc) {
            std::printf("PRIMER_ERROR=%s\n=\n", msspe_last_error(ctx));
            continue;
        }
        std::printf("PRIMER_LEFT_NUM_RETURNED=1\nPRIMER_RIGHT_NUM_RETURNED=0\nPRIMER_INTERNAL_NUM_RETURNED=0\n"
                    "PRIMER_PAIR_NUM_RETURNED=0\nPRIMER_LEFT_0_SEQUENCE=%s\nPRIMER_LEFT_0=0,%zu\n"
                    "PRIMER_LEFT_0_TM=%.3f\nPRIMER_LEFT_0_GC_PERCENT=%.3f\nPRIMER_LEFT_0_SELF_ANY_TH=%.2f\n"
                    "PRIMER_LEFT_0_SELF_END_TH=%.2f\nPRIMER_LEFT_0_HAIRPIN_TH=%.2f\n=\n",
                    r.primer.c_str(), r.primer.size(), tm, gc, sa, se, hp);
    }
    msspe_destroy(ctx);
    return 0;
}
