// nn_params.hpp -- nearest-neighbour thermodynamic parameters on the host and their device images.
//
// Source data: the Primer3 parameter files the reference hands to ntthal
// (/root/reference/od-msspe/src/delta_g.rs:90,107-108 -> `-path <cwd>/primer3_config/`), or the
// consolidated bundle shipped in open-msspe-design_amd/data/.  Index order and the N-sentinel
// rules are those of Primer3 2.6.1 thal.c's table readers.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace msspe {

constexpr int kBaseN = 4;  // sentinel code for "no base" (sequence ends)

// POD image of every table, indexed with 5-ary base codes (A C G T N).  Uploaded verbatim to
// the device for the generic kernels; the pair kernel uses the compact tables below.
struct NNTables {
    double stackS[5][5][5][5], stackH[5][5][5][5];      // Watson-Crick stacks
    double mmS[5][5][5][5], mmH[5][5][5][5];            // single internal mismatch
    double tstackS[5][5][5][5], tstackH[5][5][5][5];    // terminal stack inside loops
    double tstack2S[5][5][5][5], tstack2H[5][5][5][5];  // terminal stack at helix ends
    double d3S[5][5][5], d3H[5][5][5];                  // 3' dangle  [X][Y][Z]
    double d5S[5][5][5], d5H[5][5][5];                  // 5' dangle  [Z][X][Y]
    double interiorS[30], interiorH[30];
    double bulgeS[30], bulgeH[30];
    double hairpinS[30], hairpinH[30];
    int n_tri, n_tet;
    // tri-/tetraloop bonuses: key = 5 / 6 base codes packed 3 bits each
    uint32_t triKey[32], tetKey[128];
    double triS[32], triH[32], tetS[128], tetH[128];
};

// Loads from a Primer3-format directory or from a bundle file.  Returns false and sets err.
bool load_nn_tables(const std::string &path, NNTables &out, std::string &err);

// Chemistry-dependent constants of one thal() run (thal.c thal(): dplx_init_*, RC,
// saltCorrection) and of the decision rule.
struct ThalConsts {
    double init_S, init_H;  // duplex initiation (-5.7, 200); hairpin (-1e-11, 0)
    double RC;              // R ln(C/4e9) for duplexes of non-self-complementary oligos, 0 for hairpins
    double salt;            // 0.368 ln((mv + 120 sqrt(max(0, dv - dntp))) / 1000)
    double temp_k;          // temperature dG is reported at
    double g_cut;           // conflict iff dG <= g_cut (== "%g -> f32 < threshold", exact)
    int max_loop;
};
ThalConsts make_dimer_consts(double mv, double dv, double dntp, double dna_conc, double temp_c,
                             int max_loop, bool both_self_complementary, float dg_threshold);
ThalConsts make_hairpin_consts(double mv, double dv, double dntp, double temp_k, int max_loop);

// Compact duplex tables for the all-pairs kernel: only the entries a complementary cell can
// touch, with the end terms (thal.c LSH / RSH) evaluated once per chemistry on the host.
//   endL / endR : [a][oa][ob]        a = s1[i] (0..3), oa / ob = outer base on strand 1 / 2 (0..4)
//   ts / mm     : [x][y][z]          table[x][y][3-x][z], x,y,z in 0..3
//   wc          : [x][y]             stack[x][y][3-x][3-y]
// H values are exact integers (cal/mol) in every Primer3 table; they are kept as int32 with
// kHInf marking "not available".
constexpr int32_t kHInf = 1 << 28;
struct PairTables {
    double endL_S[100], endR_S[100];
    int32_t endL_H[100], endR_H[100];
    double ts_S[64], mm_S[64];
    int32_t ts_H[64], mm_H[64];
    double wc_S[16];
    int32_t wc_H[16];
    double loopS[2][32];   // [0] interior, [1] bulge ; index = loop size - 1
    int32_t loopH[2][32];
    int32_t h_is_integral; // 1 when every finite enthalpy is an integer (required by the kernel)
};
bool build_pair_tables(const NNTables &t, const ThalConsts &c, PairTables &out, std::string &err);

// thal.c LSH()/RSH(): the state-independent end term of a closing pair (a,b) with outer bases
// (oa on strand 1, ob on strand 2).  left = looking towards i-1 / j-1.
void end_term(const NNTables &t, const ThalConsts &c, int a, int b, int oa, int ob, bool left,
              double &S, double &H);

// Text rounding at the reference's process boundary.
float round_g_f32(double x);
float round_fixed_f32(double x, int decimals);
double g_cut(float threshold);

}  // namespace msspe
