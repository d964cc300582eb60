// nn_params.cpp -- host loader for the nearest-neighbour tables, the per-chemistry constants and
// the compact device tables of the all-pairs kernel.
//
// Replaces, on the host side, what ntthal does at start-up with `-path`
// (/root/reference/od-msspe/src/delta_g.rs:90,107-108): Primer3 2.6.1 thal.c get_thermodynamic_values()
// and its getStack / getStackint2 / getTstack / getTstack2 / getDangle / getLoop / getTriloop /
// getTetraloop readers, restated from SURVEY.md Appendix C.1.
#include "nn_params.hpp"
#include "fast_tables.hpp"
#include "split_tables.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <sys/stat.h>

namespace msspe {
namespace {

using Tokens = std::vector<std::string>;

const char *const kSections[] = {"stack.ds", "stack.dh", "stackmm.ds", "stackmm.dh",
                                 "tstack_tm_inf.ds", "tstack.dh", "tstack2.ds", "tstack2.dh",
                                 "dangle.ds", "dangle.dh", "loops.ds", "loops.dh",
                                 "triloop.ds", "triloop.dh", "tetraloop.ds", "tetraloop.dh"};

bool read_tokens(const std::string &path, Tokens &out)
{
    std::ifstream f(path);
    if (!f) return false;
    std::string tok;
    while (f >> tok) out.push_back(tok);
    return true;
}

double number(const std::string &tok)
{
    if (tok.compare(0, 3, "inf") == 0) return INFINITY;   // Primer3's spelling of "not available"
    return std::strtod(tok.c_str(), nullptr);
}

int code(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return kBaseN;
    }
}

// A four-index table as laid out in the files: 256 numbers, index ((a*4+b)*4+c)*4+d.
// mode 0: any N or any non-finite entry -> (-1, +inf)                       (stack, stackmm)
// mode 1: a or c = N -> (-1, +inf); b or d = N -> (1e-11, 0)               (tstack, tstack2)
bool fill4(const Tokens &ds, const Tokens &dh, int mode, double S[5][5][5][5],
           double H[5][5][5][5])
{
    if (ds.size() != 256 || dh.size() != 256) return false;
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            for (int c = 0; c < 5; ++c)
                for (int d = 0; d < 5; ++d) {
                    double s = -1.0, h = INFINITY;
                    const bool inner_n = (a == 4 || c == 4), outer_n = (b == 4 || d == 4);
                    if (mode == 1 && !inner_n && outer_n) {
                        s = 0.00000000001;
                        h = 0.0;
                    } else if (!inner_n && !outer_n) {
                        const int k = ((a * 4 + b) * 4 + c) * 4 + d;
                        s = number(ds[k]);
                        h = number(dh[k]);
                        if (!std::isfinite(s) || !std::isfinite(h)) {
                            s = -1.0;
                            h = INFINITY;
                        }
                    }
                    S[a][b][c][d] = s;
                    H[a][b][c][d] = h;
                }
    return true;
}

bool fill_dangles(const Tokens &ds, const Tokens &dh, NNTables &t)
{
    if (ds.size() != 128 || dh.size() != 128) return false;
    for (int x = 0; x < 5; ++x)
        for (int y = 0; y < 5; ++y)
            for (int z = 0; z < 5; ++z) {
                t.d3S[x][y][z] = t.d5S[x][y][z] = -1.0;
                t.d3H[x][y][z] = t.d5H[x][y][z] = INFINITY;
            }
    // 3' block: file index X*16 + Z*4 + Y ; 5' block: 64 + Z*16 + X*4 + Y
    for (int p = 0; p < 4; ++p)
        for (int q = 0; q < 4; ++q)
            for (int r = 0; r < 4; ++r) {
                const int k = p * 16 + q * 4 + r;
                double s = number(ds[k]), h = number(dh[k]);
                if (std::isfinite(s) && std::isfinite(h)) {
                    t.d3S[p][r][q] = s;
                    t.d3H[p][r][q] = h;
                }
                s = number(ds[64 + k]);
                h = number(dh[64 + k]);
                if (std::isfinite(s) && std::isfinite(h)) {
                    t.d5S[p][q][r] = s;
                    t.d5H[p][q][r] = h;
                }
            }
    return true;
}

bool fill_loops(const Tokens &ds, const Tokens &dh, NNTables &t)
{
    if (ds.size() != 120 || dh.size() != 120) return false;
    for (int k = 0; k < 30; ++k) {
        t.interiorS[k] = number(ds[4 * k + 1]);
        t.bulgeS[k] = number(ds[4 * k + 2]);
        t.hairpinS[k] = number(ds[4 * k + 3]);
        t.interiorH[k] = number(dh[4 * k + 1]);
        t.bulgeH[k] = number(dh[4 * k + 2]);
        t.hairpinH[k] = number(dh[4 * k + 3]);
    }
    return true;
}

// tri-/tetraloop files hold the same keys in .ds and .dh; Primer3 looks each file up on its own.
bool fill_tloops(const Tokens &ds, const Tokens &dh, size_t keylen, int cap, uint32_t *keys,
                 double *S, double *H, int &n)
{
    std::map<uint32_t, std::pair<double, double>> m;
    auto pack = [&](const std::string &key, uint32_t &out) {
        if (key.size() != keylen) return false;
        out = 0;
        for (char ch : key) out = (out << 3) | (uint32_t)code(ch);
        return true;
    };
    if (ds.size() % 2 || dh.size() % 2) return false;
    for (size_t i = 0; i + 1 < ds.size(); i += 2) {
        uint32_t k;
        if (!pack(ds[i], k)) return false;
        if (!m.count(k)) m[k] = {0.0, 0.0};
        m[k].first = number(ds[i + 1]);
    }
    for (size_t i = 0; i + 1 < dh.size(); i += 2) {
        uint32_t k;
        if (!pack(dh[i], k)) return false;
        if (!m.count(k)) m[k] = {0.0, 0.0};
        m[k].second = number(dh[i + 1]);
    }
    if ((int)m.size() > cap) return false;
    n = 0;
    for (auto &kv : m) {
        keys[n] = kv.first;
        S[n] = kv.second.first;
        H[n] = kv.second.second;
        ++n;
    }
    return true;
}

bool assemble(std::map<std::string, Tokens> &sec, NNTables &t, std::string &err)
{
    for (const char *name : kSections)
        if (!sec.count(name)) {
            err = std::string("thermodynamic table missing: ") + name;
            return false;
        }
    std::memset(&t, 0, sizeof t);
    bool ok = fill4(sec["stack.ds"], sec["stack.dh"], 0, t.stackS, t.stackH) &&
              fill4(sec["stackmm.ds"], sec["stackmm.dh"], 0, t.mmS, t.mmH) &&
              fill4(sec["tstack_tm_inf.ds"], sec["tstack.dh"], 1, t.tstackS, t.tstackH) &&
              fill4(sec["tstack2.ds"], sec["tstack2.dh"], 1, t.tstack2S, t.tstack2H) &&
              fill_dangles(sec["dangle.ds"], sec["dangle.dh"], t) &&
              fill_loops(sec["loops.ds"], sec["loops.dh"], t) &&
              fill_tloops(sec["triloop.ds"], sec["triloop.dh"], 5, 32, t.triKey, t.triS, t.triH,
                          t.n_tri) &&
              fill_tloops(sec["tetraloop.ds"], sec["tetraloop.dh"], 6, 128, t.tetKey, t.tetS,
                          t.tetH, t.n_tet);
    if (!ok) err = "thermodynamic tables malformed (wrong token count)";
    return ok;
}

double at_S(int a, int b) { return (a + b == 3 && (a == 0 || a == 3)) ? 6.9 : 0.0; }
double at_H(int a, int b) { return (a + b == 3 && (a == 0 || a == 3)) ? 2200.0 : 0.0; }
bool pairs(int a, int b) { return a < 4 && b < 4 && a + b == 3; }

}  // namespace

bool load_nn_tables(const std::string &path, NNTables &out, std::string &err)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) {
        err = "cannot stat thermodynamic parameter path: " + path;
        return false;
    }
    std::map<std::string, Tokens> sec;
    if (S_ISDIR(st.st_mode)) {
        for (const char *name : kSections) {
            std::string p = path;
            if (!p.empty() && p.back() != '/') p += '/';
            if (!read_tokens(p + name, sec[name])) {
                err = "cannot read " + p + name;
                return false;
            }
        }
    } else {
        Tokens all;
        if (!read_tokens(path, all)) {
            err = "cannot read " + path;
            return false;
        }
        size_t i = 0;
        while (i < all.size() && all[i] != "@") ++i;   // skip the '#' comment header
        while (i < all.size()) {
            if (all[i] != "@" || i + 2 >= all.size()) {
                err = "bundle syntax error in " + path;
                return false;
            }
            const std::string name = all[i + 1];
            const size_t cnt = (size_t)std::strtoul(all[i + 2].c_str(), nullptr, 10);
            i += 3;
            if (i + cnt > all.size()) {
                err = "bundle truncated in section " + name;
                return false;
            }
            sec[name].assign(all.begin() + (long)i, all.begin() + (long)(i + cnt));
            i += cnt;
        }
    }
    return assemble(sec, out, err);
}

// thal.c saltCorrectS()
static double salt_correction(double mv, double dv, double dntp)
{
    if (dv <= 0) dntp = dv;
    return 0.368 * (std::log((mv + 120 * (std::sqrt(std::fmax(0.0, dv - dntp)))) / 1000));
}

ThalConsts make_dimer_consts(double mv, double dv, double dntp, double dna_conc, double temp_c,
                             int max_loop, bool both_self_complementary, float dg_threshold)
{
    ThalConsts c;
    c.init_S = -5.7;
    c.init_H = 200;
    c.RC = 1.9872 * std::log(dna_conc / (both_self_complementary ? 1000000000.0 : 4000000000.0));
    c.salt = salt_correction(mv, dv, dntp);
    c.temp_k = temp_c + 273.15;
    c.g_cut = g_cut(dg_threshold);
    c.max_loop = max_loop;
    return c;
}

ThalConsts make_hairpin_consts(double mv, double dv, double dntp, double temp_k, int max_loop)
{
    ThalConsts c;
    c.init_S = -0.00000000001;
    c.init_H = 0.0;
    c.RC = 0;
    c.salt = salt_correction(mv, dv, dntp);
    c.temp_k = temp_k;
    c.g_cut = 0;
    c.max_loop = max_loop;
    return c;
}

void end_term(const NNTables &t, const ThalConsts &c, int a, int b, int oa, int ob, bool left,
              double &outS, double &outH)
{
    const double T37 = 310.15;
    if (!pairs(a, b)) {
        outS = -1.0;
        outH = INFINITY;
        return;
    }
    double tS, tH, d3S, d3H, d5S, d5H;
    if (left) {   // outer bases i-1 / j-1; strand 2 is the "first" strand of the lookup
        tS = t.tstack2S[b][ob][a][oa];
        tH = t.tstack2H[b][ob][a][oa];
        d3S = t.d3S[b][ob][a];
        d3H = t.d3H[b][ob][a];
        d5S = t.d5S[b][a][oa];
        d5H = t.d5H[b][a][oa];
    } else {      // outer bases i+1 / j+1
        tS = t.tstack2S[a][oa][b][ob];
        tH = t.tstack2H[a][oa][b][ob];
        d3S = t.d3S[a][oa][b];
        d3H = t.d3H[a][oa][b];
        d5S = t.d5S[a][b][ob];
        d5H = t.d5H[a][b][ob];
    }
    const double aS = at_S(a, b), aH = at_H(a, b);
    // option T: terminal mismatch stack
    double S1 = aS + tS, H1 = aH + tH, G1 = H1 - T37 * S1, T1 = -INFINITY;
    if (!std::isfinite(H1) || G1 > 0) {
        H1 = INFINITY;
        S1 = -1.0;
        G1 = 1.0;
    }
    // option D: dangling ends, only when the outer bases cannot pair
    bool haveD = false;
    double S2 = -1.0, H2 = INFINITY;
    if (!pairs(oa, ob)) {
        if (std::isfinite(d3H) && std::isfinite(d5H)) {
            S2 = aS + d3S + d5S;
            H2 = aH + d3H + d5H;
            haveD = true;
        } else if (std::isfinite(d3H)) {
            S2 = aS + d3S;
            H2 = aH + d3H;
            haveD = true;
        } else if (std::isfinite(d5H)) {
            S2 = aS + d5S;
            H2 = aH + d5H;
            haveD = true;
        }
    }
    if (haveD) {
        double G2 = H2 - T37 * S2;
        if (!std::isfinite(H2) || G2 > 0) {
            H2 = INFINITY;
            S2 = -1.0;
            G2 = 1.0;
        }
        const double T2 = (H2 + c.init_H) / (S2 + c.init_S + c.RC);
        if (std::isfinite(H1) && G1 < 0) {
            T1 = (H1 + c.init_H) / (S1 + c.init_S + c.RC);
            if (T1 < T2 && G2 < 0) {
                S1 = S2;
                H1 = H2;
                T1 = T2;
            }
        } else if (G2 < 0) {
            S1 = S2;
            H1 = H2;
            T1 = T2;
        }
    }
    // bare closing pair; T1 is still -inf when no dangle option was considered (Primer3 quirk)
    const double Tb = (aH + c.init_H) / (aS + c.init_S + c.RC);
    if (std::isfinite(H1) && !(T1 < Tb)) {
        outS = S1;
        outH = H1;
    } else {
        outS = aS;
        outH = aH;
    }
}

bool build_pair_tables(const NNTables &t, const ThalConsts &c, PairTables &out, std::string &err)
{
    std::memset(&out, 0, sizeof out);
    bool integral = true;
    auto as_int = [&](double h) -> int32_t {
        if (!std::isfinite(h)) return kHInf;
        if (h != std::nearbyint(h) || std::fabs(h) > 1e6) integral = false;
        return (int32_t)h;
    };
    for (int a = 0; a < 4; ++a)
        for (int oa = 0; oa < 5; ++oa)
            for (int ob = 0; ob < 5; ++ob) {
                const int idx = a * 25 + oa * 5 + ob;
                double S, H;
                end_term(t, c, a, 3 - a, oa, ob, true, S, H);
                out.endL_S[idx] = S;
                out.endL_H[idx] = as_int(H);
                end_term(t, c, a, 3 - a, oa, ob, false, S, H);
                out.endR_S[idx] = S;
                out.endR_H[idx] = as_int(H);
            }
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y) {
            out.wc_S[x * 4 + y] = t.stackS[x][y][3 - x][3 - y];
            out.wc_H[x * 4 + y] = as_int(t.stackH[x][y][3 - x][3 - y]);
            for (int z = 0; z < 4; ++z) {
                const int idx = (x * 4 + y) * 4 + z;
                out.ts_S[idx] = t.tstackS[x][y][3 - x][z];
                out.ts_H[idx] = as_int(t.tstackH[x][y][3 - x][z]);
                out.mm_S[idx] = t.mmS[x][y][3 - x][z];
                out.mm_H[idx] = as_int(t.mmH[x][y][3 - x][z]);
            }
        }
    for (int k = 0; k < 32; ++k) {
        out.loopS[0][k] = k < 30 ? t.interiorS[k] : -1.0;
        out.loopH[0][k] = k < 30 ? as_int(t.interiorH[k]) : kHInf;
        out.loopS[1][k] = k < 30 ? t.bulgeS[k] : -1.0;
        out.loopH[1][k] = k < 30 ? as_int(t.bulgeH[k]) : kHInf;
    }
    out.h_is_integral = integral ? 1 : 0;
    if (!integral) err = "enthalpy tables are not integral: the all-pairs kernel cannot be used";
    return integral;
}

float round_g_f32(double x)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%g", x);
    return std::strtof(buf, nullptr);
}

float round_fixed_f32(double x, int decimals)
{
    char buf[400];
    std::snprintf(buf, sizeof buf, "%.*f", decimals, x);
    return std::strtof(buf, nullptr);
}

// Doubles ordered as unsigned integers: monotone map double -> uint64.
static uint64_t ordered_key(double d)
{
    uint64_t k;
    std::memcpy(&k, &d, sizeof k);
    return (k & 0x8000000000000000ull) ? ~k : (k | 0x8000000000000000ull);
}
static double from_key(uint64_t k)
{
    k = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    double d;
    std::memcpy(&d, &k, sizeof d);
    return d;
}

double g_cut(float threshold)
{
    // The reference keeps an edge when the %g text of dG, parsed as f32, is below the threshold
    // (delta_g.rs:33-36), stores it as "{:.2}" (:45) and tests the re-parsed text again (main.rs:758,
    // delta_g.rs:10-15): below |dG| = 1000 the second text is coarser than the first and can lift a value
    // back to the threshold.  Both roundings are monotone, so the conjunction is (true for small x):
    // return the largest double for which it is true, so that the kernels can test dG <= cut.
    if (std::isnan(threshold)) return -INFINITY;
    auto dec = [&](double x) {
        const float first = round_g_f32(x);
        return first < threshold && round_fixed_f32((double)first, 2) < threshold;
    };
    uint64_t lo = ordered_key(-DBL_MAX), hi = ordered_key(DBL_MAX);
    if (!dec(from_key(lo))) return -INFINITY;
    if (dec(from_key(hi))) return DBL_MAX;
    while (hi - lo > 1) {   // invariant: dec(lo) true, dec(hi) false
        const uint64_t mid = lo + (hi - lo) / 2;
        if (dec(from_key(mid))) lo = mid;
        else hi = mid;
    }
    return from_key(lo);
}

// The compact S / H planes shared by FastTables and SplitTables (same layout, other loop-size
// range): T provides the k* offsets, kMaxSz, S[] and H[].
template <class T>
static void fill_compact_planes(const PairTables &pt, T &out, bool &ok, double &min_S)
{
    auto put = [&](int idx, double S, int32_t H) {
        // an unavailable entry carries a huge positive entropy next to its huge enthalpy, so that
        // the kernel's single rejection test "H > 0 and S > 0" (thal.c's both-positive rule)
        // also catches it, whatever finite terms are added on top
        out.S[idx] = H >= kHInf ? 1e300 : S;
        out.H[idx] = H;
        if (H < kHInf) {
            if (H % 10 != 0) ok = false;
            if (S < min_S) min_S = S;
        }
    };
    auto addH = [](int32_t a, int32_t b) -> int32_t { return (a >= kHInf || b >= kHInf) ? kHInf : a + b; };
    const double atS[4] = {6.9, 0.0, 0.0, 6.9};
    const int32_t atH[4] = {2200, 0, 0, 2200};
    // po = a | oa << 2 | ob << 4  ->  PairTables index (a*4 + oa)*4 + ob
    auto src_of = [](int po) { return ((po & 3) * 4 + ((po >> 2) & 3)) * 4 + (po >> 4); };
    for (int sz = 2; sz <= T::kMaxSz; ++sz)
        for (int po = 0; po < 64; ++po) {
            const int idx = T::kNB + (sz - 2) * 64 + po;
            if (sz == 2) put(idx, pt.mm_S[src_of(po)], pt.mm_H[src_of(po)]);
            else put(idx, pt.loopS[0][sz - 1] + pt.ts_S[src_of(po)], addH(pt.loopH[0][sz - 1], pt.ts_H[src_of(po)]));
        }
    for (int ac = 0; ac < 4; ++ac)
        for (int sz = 0; sz <= T::kMaxSz; ++sz)
            for (int ap = 0; ap < 4; ++ap) {
                const int idx = T::kBU + ac * T::kBUStride + sz * 4 + ap;
                double S = -1.0;
                int32_t H = kHInf;
                if (sz == 1) {
                    S = pt.loopS[1][0] + pt.wc_S[ap * 4 + ac];
                    H = addH(pt.loopH[1][0], pt.wc_H[ap * 4 + ac]);
                    if (H >= kHInf || H > 0 || S > 0) {   // thal.c: isPositive(H) || isPositive(S)
                        S = -1.0;
                        H = kHInf;
                    }
                } else if (sz >= 2) {
                    S = (pt.loopS[1][sz - 1] + atS[ap]) + atS[ac];
                    H = addH(addH(pt.loopH[1][sz - 1], atH[ap]), atH[ac]);
                }
                put(idx, S, H);
            }
    for (int ci = 0; ci < 64; ++ci) {
        put(T::kTSc + ci, pt.ts_S[ci], pt.ts_H[ci]);   // cell side: (x*4+y)*4+z already
        put(T::kMMc + ci, pt.mm_S[ci], pt.mm_H[ci]);
    }
    for (int q = 0; q < 4; ++q) put(T::kZero + q, 0.0, 0);
    const double ilas = (-300 / 310.15);
    for (int d = -32; d < 32; ++d) put(T::kZT + 32 + d, ilas * (d < 0 ? -d : d), 0);
    for (int q = 0; q < 100; ++q) {
        put(T::kEndL + q, pt.endL_S[q], pt.endL_H[q]);
        put(T::kEndR + q, pt.endR_S[q], pt.endR_H[q]);
    }
    for (int q = 0; q < 16; ++q) put(T::kWC + q, pt.wc_S[q], pt.wc_H[q]);
}

bool build_fast_tables(const NNTables &t, const PairTables &pt, int max_k, FastTables &out)
{
    (void)t;
    std::memset(&out, 0, sizeof out);
    bool ok = pt.h_is_integral != 0;
    double min_S = 0.0;     // most negative entropy any single table term can add
    fill_compact_planes(pt, out, ok, min_S);
    // clamp reachability: a path has at most max_k pairs and every step adds at most three table
    // terms, each >= min_S; plus two end terms
    if (3.0 * min_S * (max_k + 2) < -2500.0) ok = false;
    if (2 * max_k - 4 > FastTables::kMaxSz) ok = false;
    out.usable = ok ? 1 : 0;
    out.max_k = max_k;
    return ok;
}

bool build_int_tables(const FastTables &ft, int max_k, IntTables &out)
{
    std::memset(&out, 0, sizeof out);
    bool ok = ft.usable != 0 && max_k <= IntTables::kMaxL + 2;
    long max_abs_g = 0, max_abs_s = 0;
    for (int e = 0; e < FastTables::kCount; ++e) {
        const bool zt = e >= FastTables::kZT && e < FastTables::kZT + 64;   // asymmetry: carried as n
        if (ft.H[e] >= kHInf) {
            out.g[e] = IntTables::kBig;
            out.s[e] = 0;
            continue;
        }
        if (zt) {
            const int d = e - (FastTables::kZT + 32);
            out.g[e] = IntTables::kGn * (d < 0 ? -d : d);
            out.s[e] = 0;
            continue;
        }
        const double s100 = ft.S[e] * 100.0;
        const double r = std::nearbyint(s100);
        if (std::fabs(s100 - r) > 1e-7 || std::fabs(r) > 1e5 || ft.H[e] % 10 != 0) ok = false;
        const long si = (long)r, hi = ft.H[e] / 10;
        const long g = (long)IntTables::kGh * hi - (long)IntTables::kGs * si;
        if (std::labs(g) > max_abs_g) max_abs_g = std::labs(g);
        if (std::labs(si) > max_abs_s) max_abs_s = std::labs(si);
        out.g[e] = (int32_t)g;
        out.s[e] = (int32_t)si;
    }
    // Range of a reachable value: a path spends the 2 * max_k bases of the two oligos, a stacked
    // pair costs 2 of them and a loop of size sz costs sz + 2, so |G| <= 2 * max_k * (largest
    // |g| per base) + two end terms.  It must stay below kReach (fast_tables.hpp).
    {
        (void)max_abs_g;
        (void)max_abs_s;
        auto mag = [&](int e) -> double {
            return out.g[e] == IntTables::kBig ? 0.0 : std::fabs((double)out.g[e]);
        };
        double per_base = 0.0, mm = 0.0, en = 0.0;
        for (int e = FastTables::kWC; e < FastTables::kWC + 16; ++e) per_base = std::max(per_base, mag(e) / 2.0);
        for (int e = FastTables::kTSc; e < FastTables::kZero; ++e) mm = std::max(mm, mag(e));
        for (int sz = 2; sz <= FastTables::kMaxSz; ++sz)
            for (int po = 0; po < 64; ++po)
                per_base = std::max(per_base, (mag(FastTables::kNB + (sz - 2) * 64 + po) + mm +
                                               (double)IntTables::kGn * sz) / (sz + 2));
        for (int ac = 0; ac < 4; ++ac)
            for (int sz = 1; sz <= FastTables::kMaxSz; ++sz)
                for (int ap = 0; ap < 4; ++ap)
                    per_base = std::max(per_base,
                                        mag(FastTables::kBU + ac * FastTables::kBUStride + sz * 4 + ap) / (sz + 2));
        for (int e = FastTables::kEndL; e < FastTables::kWC; ++e) en = std::max(en, mag(e));
        if (2.0 * max_k * per_base + 2.0 * en + mm >= IntTables::kReach) ok = false;
    }
    for (int d = 0; d < IntTables::kRows; ++d)
        for (int pe = 0; pe < 64; ++pe) {
            const int l1 = d >> 4, l2 = d & 15, sz = l1 + l2;
            int32_t v = IntTables::kBig;
            if (d != 0 && l1 <= IntTables::kMaxL && l2 <= IntTables::kMaxL && sz <= FastTables::kMaxSz) {
                if (l1 == 0 || l2 == 0) {
                    if (pe < 16)
                        v = out.g[FastTables::kBU + (pe >> 2) * FastTables::kBUStride + sz * 4 + (pe & 3)];
                } else {
                    v = out.g[FastTables::kNB + (sz - 2) * 64 + pe];
                    if (v != IntTables::kBig && d != 0x11)
                        v += IntTables::kGn * (l1 > l2 ? l1 - l2 : l2 - l1);
                }
            }
            out.T[d * 64 + pe] = v;
        }
    out.usable = ok ? 1 : 0;
    out.max_k = max_k;
    return ok;
}

bool build_split_tables(const PairTables &pt, int max_loop, SplitTables &out)
{
    typedef SplitTables W;
    std::memset(&out, 0, sizeof out);
    bool ok = pt.h_is_integral != 0;
    double min_S = 0.0;
    fill_compact_planes(pt, out, ok, min_S);
    const bool planes_ok = ok;
    // integer image of one (S, H) term
    auto gi = [&](double S, int32_t H) -> int32_t {
        if (H >= kHInf) return W::kBig;
        const double s100 = S * 100.0, r = std::nearbyint(s100);
        if (std::fabs(s100 - r) > 1e-7 || std::fabs(r) > 1e5 || H % 10 != 0) {
            ok = false;
            return W::kBig;
        }
        return (int32_t)((long)W::kGh * (H / 10) - (long)W::kGs * (long)r);
    };
    for (int e = 0; e < W::kCount; ++e) {
        if (e >= W::kZT && e < W::kZT + 64) {   // asymmetry: carried as n
            const int d = e - (W::kZT + 32);
            out.g[e] = W::kGn * (d < 0 ? -d : d);
        } else {
            out.g[e] = gi(out.S[e], out.H[e]);
        }
    }
    auto src_of = [](int po) { return ((po & 3) * 4 + ((po >> 2) & 3)) * 4 + (po >> 4); };
    const int lim = std::min(max_loop, W::kMaxSz);
    for (int d = 0; d < 1024; ++d) {
        const int l1 = d >> 5, l2 = d & 31, sz = l1 + l2;
        int32_t v = W::kBig;
        if (d != 0 && sz <= lim) {
            if (l1 == 0 || l2 == 0 || d == 0x21) v = 0;
            else {
                v = gi(pt.loopS[0][sz - 1], pt.loopH[0][sz - 1]);
                if (v != W::kBig) v += W::kGn * (l1 > l2 ? l1 - l2 : l2 - l1);
            }
        }
        out.L[d] = v;
    }
    for (int po = 0; po < 64; ++po) {
        out.X[W::kXP + po] = gi(pt.ts_S[src_of(po)], pt.ts_H[src_of(po)]);
        out.X[W::kXMM + po] = gi(pt.mm_S[src_of(po)], pt.mm_H[src_of(po)]);
    }
    for (int sz = 0; sz < 32; ++sz)
        for (int pe = 0; pe < 16; ++pe) {
            const int32_t v = (sz >= 1 && sz <= W::kMaxSz)
                                  ? out.g[W::kBU + (pe >> 2) * W::kBUStride + sz * 4 + (pe & 3)] : W::kBig;
            out.X[W::kXB1 + sz * 16 + pe] = v;
            out.X[W::kXB2 + sz * 16 + pe] = v;
        }
    // the two-part loop term must reproduce the folded entry (it does when both parts sit on the grid)
    for (int sz = 3; sz <= W::kMaxSz && ok; ++sz)
        for (int po = 0; po < 64; ++po) {
            const int32_t whole = out.g[W::kNB + (sz - 2) * 64 + po];
            const int32_t a = gi(pt.loopS[0][sz - 1], pt.loopH[0][sz - 1]), b = out.X[W::kXP + po];
            if (whole == W::kBig || a == W::kBig || b == W::kBig) {
                if (!(whole == W::kBig && (a == W::kBig || b == W::kBig))) ok = false;
            } else if (whole != a + b) ok = false;
        }
    // Ranges.  A stored cell value is the left end term plus stacked pairs plus accepted loops.  A
    // loop is only accepted when it LOWERS the value, a stacked pair adds g(wc): so the value never
    // exceeds (largest end term) + (stacks) * max(0, g(wc)); from below it is bounded per base
    // spent (a path uses the 2 k bases of the two oligos, a stacked pair costs 2, a loop of size sz
    // costs sz + 2) by the most negative term sums.  Enthalpy and entropy: magnitudes per base.
    auto gv = [&](int e) -> double { return out.g[e] == W::kBig ? 0.0 : (double)out.g[e]; };
    auto hmag = [&](int e) -> double { return out.H[e] >= kHInf ? 0.0 : std::fabs((double)out.H[e]); };
    auto sneg = [&](int e) -> double { return (out.H[e] >= kHInf || out.S[e] > 0) ? 0.0 : -out.S[e]; };
    auto hpos = [&](int e) -> double { return out.H[e] >= kHInf ? 0.0 : std::max(0.0, (double)out.H[e]); };
    double hp = 0, hpmm = 0, hpen = 0;   // positive enthalpy per base spent / per mismatch / per end
    for (int e = W::kWC; e < W::kWC + 16; ++e) hp = std::max(hp, hpos(e) / 2.0);
    for (int e = W::kTSc; e < W::kZero; ++e) hpmm = std::max(hpmm, hpos(e));
    for (int sz = 2; sz <= W::kMaxSz; ++sz)
        for (int po = 0; po < 64; ++po) hp = std::max(hp, (hpos(W::kNB + (sz - 2) * 64 + po) + hpmm) / (sz + 2));
    for (int ac = 0; ac < 4; ++ac)
        for (int sz = 1; sz <= W::kMaxSz; ++sz)
            for (int ap = 0; ap < 4; ++ap) hp = std::max(hp, hpos(W::kBU + ac * W::kBUStride + sz * 4 + ap) / (sz + 2));
    for (int e = W::kEndL; e < W::kWC; ++e) hpen = std::max(hpen, hpos(e));
    double gneg = 0, gpos_wc = 0, hb = 0, sb = 0, gmm_lo = 0, hmm = 0, smm = 0, gen = 0, hen = 0, sen = 0;
    for (int e = W::kWC; e < W::kWC + 16; ++e) {
        gneg = std::max(gneg, -gv(e) / 2.0);
        gpos_wc = std::max(gpos_wc, gv(e));
        hb = std::max(hb, hmag(e) / 2.0);
        sb = std::max(sb, sneg(e) / 2.0);
    }
    for (int e = W::kTSc; e < W::kZero; ++e) {
        gmm_lo = std::max(gmm_lo, -gv(e));
        hmm = std::max(hmm, hmag(e));
        smm = std::max(smm, sneg(e));
    }
    for (int sz = 2; sz <= W::kMaxSz; ++sz)
        for (int po = 0; po < 64; ++po) {
            const int e = W::kNB + (sz - 2) * 64 + po;
            gneg = std::max(gneg, (-gv(e) + gmm_lo) / (sz + 2));   // the asymmetry term is >= 0
            hb = std::max(hb, (hmag(e) + hmm) / (sz + 2));
            sb = std::max(sb, (sneg(e) + smm + 0.97 * sz) / (sz + 2));
        }
    for (int ac = 0; ac < 4; ++ac)
        for (int sz = 1; sz <= W::kMaxSz; ++sz)
            for (int ap = 0; ap < 4; ++ap) {
                const int e = W::kBU + ac * W::kBUStride + sz * 4 + ap;
                gneg = std::max(gneg, -gv(e) / (sz + 2));
                hb = std::max(hb, hmag(e) / (sz + 2));
                sb = std::max(sb, sneg(e) / (sz + 2));
            }
    for (int e = W::kEndL; e < W::kWC; ++e) {
        gen = std::max(gen, std::fabs(gv(e)));
        hen = std::max(hen, hmag(e));
        sen = std::max(sen, sneg(e));
    }
    // a candidate in flight adds one loop term, one predecessor-side and one cell-side term
    double cand_hi = 0;
    for (int d = 1; d < 1024; ++d)
        if (out.L[d] != W::kBig) cand_hi = std::max(cand_hi, (double)out.L[d]);
    {
        double x_hi = 0, y_hi = 0;
        for (int e = 0; e < W::kXCount; ++e)
            if (out.X[e] != W::kBig) x_hi = std::max(x_hi, (double)out.X[e]);
        for (int e = W::kTSc; e < W::kZero; ++e) y_hi = std::max(y_hi, gv(e));
        cand_hi += x_hi + y_hi;
    }
    // enthalpy H / 10 is kept in 16 unsigned bits with a bias: mostly negative values
    constexpr int kHBias = 37500;   // stock tables: -368,000 .. +255,000 cal/mol for 32-mers
    out.h_bias = kHBias;
    int max_k = 0;
    for (int k = 32; k >= 2 && ok; --k) {
        const bool fits = 2.0 * k * gneg + 2.0 * gen + gmm_lo < (double)W::kReach &&     // int32 sums, low side
                          2.0 * gen + k * gpos_wc + cand_hi < (double)W::kReach &&         // ... high side
                          2.0 * k * hb + 2.0 * hen + hmm + 200.0 < 10.0 * kHBias &&                 // H / 10 + bias in 16 bits: low side
                          2.0 * k * hp + 2.0 * hpen + hpmm + 200.0 < 10.0 * (65000 - kHBias) &&  // ... high side
                          2.0 * k * sb + 2.0 * sen + smm + 6.0 < 2400.0;            // MinEntropyCutoff (-2500) out of reach
        if (fits) {
            max_k = k;
            break;
        }
    }
    out.usable = (ok && max_k >= 2) ? 1 : 0;
    out.max_k = out.usable ? max_k : 0;
    out.f64_max_k = 0;
    for (int k = 32; k >= 2 && planes_ok; --k)
        if (2.0 * k * sb + 2.0 * sen + smm + 6.0 < 2400.0) {
            out.f64_max_k = k;
            break;
        }
    return out.usable != 0;
}

}  // namespace msspe
