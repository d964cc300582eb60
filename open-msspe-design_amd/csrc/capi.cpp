// capi.cpp -- extern "C" boundary of libmsspe_hip.so (declared in include/msspe_hip.h).
//
// Each entry point replaces one process boundary or stage-A function of the reference:
//   msspe_cross_dimer*     run_ntthal            /root/reference/od-msspe/src/delta_g.rs:83-153
//   msspe_oligo_stats*     check_primers         /root/reference/od-msspe/src/primer.rs:143-166
//   msspe_kmer_candidates* get_segment_manager + find_candidates_kmers  src/main.rs:196-235,331-406
// There is no CPU fallback: every compute entry point needs a gfx950 device.
#include "../../include/msspe_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "kmer_stage.hpp"
#include "nn_params.hpp"
#include "thal_dense.hpp"

using namespace msspe;

namespace {

struct ChemEntry {
    msspe_chem chem;
    float threshold;
    ThalConsts c[2];
    PairTables *d_pt = nullptr;   // 2 entries: ordinary, both self-complementary
    FastTables *d_ft = nullptr;   // tables of the tuned all-pairs kernel (ordinary pairs)
    bool fast_ok = false;
    IntTables *d_it = nullptr;    // integer image for the exact-integer kernel
    bool int_ok = false;
    bool row_ok = false;          // ... and the row-specialised kernel (thal_pairs_row.hip)
    SplitTables *d_st = nullptr;  // long oligos (thal_pairs_split.hip)
    int split_max_k = 0;          // 0: not usable
    int wave_max_k = 0;           // f64 one-wave-per-pair kernel (thal_pairs_wave.hip)
};

constexpr long kChunkPairs = 1L << 29;       // pairs per launch of the all-pairs kernel (a launch's tail: 2.4 % at 2^24, 1.4 % at 2^26; a 65,536-primer
                                             // pool: 1850 ms at 2^27, 1822 at 2^29, 1819 at 2^30 -- and a hand-over list of as many entries, 4 GB, twice)
constexpr int kHairpinLaneFrom = 8192;       // oligos per call from which HAIRPIN_TH runs one lane per oligo (msspe_oligo_stats_dev)
constexpr long kListCapMin = 1L << 20;       // hand-over list entries (grows with the call up to kListCapMax): one launch can never overrun it
constexpr long kListCapMax = 1L << 30;       // 8 GB per list (two of them, 6 % of the card's memory): the stages behind the first
                                             // run every two launches of 2^29 pairs, so that a list cannot be overrun even if
                                             // every pair were handed on (2.7 % are); k_accumulate_overflow checks the counters
                                             // against it all the same.  The small kernels of a flush do not fill the card:
                                             // 2^28 -> 2^30 is 17 -> 5 flushes per 65,536^2 screen and 1.5 % of its time
constexpr size_t kGenericLanes = 1u << 16;   // lanes of the generic kernels' workspace

}  // namespace

// Engine options (msspe_set_option): which kernels a call may use.  Defaults are the product path;
// the rest exists for the parity tests (every stage against every other) and for experiments.
struct EngineOptions {
    int pair_kernel = 0;      // 0 auto | 1 f64 register-table kernel first | 2 general integer kernel first
    bool force_generic = false;
    int split_min_k = 16;     // 14- and 15-mers: the row-specialised first stage (round 3); 16 and up: the split-table kernel
    bool wave_kernel = true;
    int list_cap_log2 = 0;    // 0: sized by the call; 20..30: fixed (forces flushes mid-screen)
    int split_lanes = 0;      // 0: by oligo length; 2 / 4 / 8
    bool row_oob = true;      // the row-specialised first stage (it reads LDS beyond its allocation: thal_pairs_row.hip) may run
    bool split_list = true;   // short oligos: tables too large for the integer list stage go to the split kernel's list mode
    bool short_chain = true;  // screens of up to 2^23 pairs: integer list stage -> one wave per pair (no register-table stages between)
    int self_lane_from = 81920;   // oligos per call from which SELF_ANY / SELF_END run one lane per oligo (msspe_oligo_stats_dev)
};

struct msspe_ctx {
    int device = 0;
    int n_cu = 256;
    bool lds_reads_zero = false;       // pairs_row_lds_reads_zero(): the row kernel may run
    EngineOptions opt;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    NNTables host_tb;
    NNTables *d_tb = nullptr;
    std::vector<ChemEntry> chem_cache;
    double *wsS = nullptr, *wsH = nullptr;
    size_t ws_cells = 0;
    uint2 *ovf_list = nullptr;         // pairs the main kernel could not hold
    uint2 *ovf_list2 = nullptr;        // pairs the wide kernel could not hold either
    uint32_t *ovf_count = nullptr;     // list counters of the stages (8): [0] first, [1] second, ...
    long list_cap = 0;                 // entries per hand-over list
    long list_cap_ceiling = 1L << 30;  // lowered when an allocation of that size failed (not tried again)
    uint64_t *d_ovf_total = nullptr;   // [0] pairs handed on so far, [1] != 0: a list counter went past its capacity
    unsigned long long *d_reasons = nullptr;   // [8] statistics of the integer stage
    uint64_t *d_sorted = nullptr;      // column primers grouped by composition
    uint32_t *d_perm = nullptr;
    void *d_sort_scratch = nullptr;    // keys, values and rocPRIM temporary storage of the composition sort
    size_t sort_cap = 0, sort_scratch_bytes = 0;
    std::string err;
    KmerStage kmer;
    KmerStage kmer_rev;                // direction 1 of msspe_kmer_candidates_both_packed_dev (its own buffers and loop graph)
    hipStream_t stream_rev = nullptr;  // ... and its stream
    hipEvent_t ev_rev = nullptr;
    // optional profiling of the dominant kernel (k_pairs_fast) with HIP events on ctx->stream
    bool prof_on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
};

namespace {

int fail(msspe_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

int hip_fail(msspe_ctx *ctx, hipError_t e, const char *what)
{
    return fail(ctx, MSSPE_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(ctx, expr)                                              \
    do {                                                                \
        hipError_t e__ = (expr);                                        \
        if (e__ != hipSuccess) return hip_fail((ctx), e__, #expr);      \
    } while (0)

std::string default_bundle_path()
{
    Dl_info info;
    if (dladdr((void *)&msspe_version, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t slash = p.find_last_of('/');
        p = slash == std::string::npos ? std::string(".") : p.substr(0, slash);
        return p + "/data/nn_params.bundle";
    }
    return "data/nn_params.bundle";
}

bool same_chem(const msspe_chem &a, const msspe_chem &b)
{
    return a.mv == b.mv && a.dv == b.dv && a.dntp == b.dntp && a.dna_conc == b.dna_conc &&
           a.temp_c == b.temp_c && a.max_loop == b.max_loop;
}

int chem_entry(msspe_ctx *ctx, const msspe_chem &chem, float threshold, ChemEntry **out)
{
    for (auto &e : ctx->chem_cache)
        if (same_chem(e.chem, chem) && (e.threshold == threshold ||
                                        (std::isnan(e.threshold) && std::isnan(threshold)))) {
            *out = &e;
            return MSSPE_OK;
        }
    if (!(chem.dna_conc > 0) || chem.max_loop < 0 || chem.max_loop > 30)
        return fail(ctx, MSSPE_ERR_ARG, "chemistry: dna_conc must be > 0 and 0 <= max_loop <= 30");
    ChemEntry e;
    e.chem = chem;
    e.threshold = threshold;
    PairTables host_pt[2];
    for (int sym = 0; sym < 2; ++sym) {
        e.c[sym] = make_dimer_consts(chem.mv, chem.dv, chem.dntp, chem.dna_conc, chem.temp_c,
                                     chem.max_loop, sym == 1, threshold);
        std::string err;
        if (!build_pair_tables(ctx->host_tb, e.c[sym], host_pt[sym], err))
            return fail(ctx, MSSPE_ERR_TABLES, err);
    }
    HIP_TRY(ctx, hipMalloc((void **)&e.d_pt, sizeof host_pt));
    HIP_TRY(ctx, hipMemcpy(e.d_pt, host_pt, sizeof host_pt, hipMemcpyHostToDevice));
    {
        auto ft = std::make_unique<FastTables>();
        e.fast_ok = build_fast_tables(ctx->host_tb, host_pt[0], pairs_fast_max_k(), *ft);
        HIP_TRY(ctx, hipMalloc((void **)&e.d_ft, sizeof(FastTables)));
        HIP_TRY(ctx, hipMemcpy(e.d_ft, ft.get(), sizeof(FastTables), hipMemcpyHostToDevice));
        auto it = std::make_unique<IntTables>();
        e.int_ok = e.fast_ok && build_int_tables(*ft, pairs_fast_max_k(), *it);
        e.row_ok = e.int_ok && pairs_row_tables_ok(*it);
        HIP_TRY(ctx, hipMalloc((void **)&e.d_it, sizeof(IntTables)));
        HIP_TRY(ctx, hipMemcpy(e.d_it, it.get(), sizeof(IntTables), hipMemcpyHostToDevice));
        auto st = std::make_unique<SplitTables>();
        e.split_max_k = build_split_tables(host_pt[0], chem.max_loop, *st) ? st->max_k : 0;
        e.wave_max_k = st->f64_max_k;
        HIP_TRY(ctx, hipMalloc((void **)&e.d_st, sizeof(SplitTables)));
        HIP_TRY(ctx, hipMemcpy(e.d_st, st.get(), sizeof(SplitTables), hipMemcpyHostToDevice));
    }
    ctx->chem_cache.push_back(e);
    *out = &ctx->chem_cache.back();
    return MSSPE_OK;
}

int ensure_workspace(msspe_ctx *ctx, size_t cells_per_lane)
{
    if (ctx->ws_cells >= cells_per_lane) return MSSPE_OK;
    if (ctx->wsS) (void)hipFree(ctx->wsS);
    if (ctx->wsH) (void)hipFree(ctx->wsH);
    ctx->wsS = ctx->wsH = nullptr;
    ctx->ws_cells = 0;
    const size_t bytes = cells_per_lane * kGenericLanes * sizeof(double);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->wsS, bytes));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->wsH, bytes));
    ctx->ws_cells = cells_per_lane;
    return MSSPE_OK;
}

// The lists are sized by the call (every pair could be handed on between two flushes): small
// screens keep small lists, the 65,536-primer screen flushes every 8 launches.
int ensure_overflow(msspe_ctx *ctx, long total_pairs)
{
    // the small counter buffers first and unconditionally: a failed list allocation must not leave a
    // context whose counters are missing
    if (!ctx->ovf_count) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->ovf_count, sizeof(uint32_t) * 8));
        HIP_TRY(ctx, hipMemsetAsync(ctx->ovf_count, 0, sizeof(uint32_t) * 8, ctx->stream));
    }
    if (!ctx->d_ovf_total) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_ovf_total, 2 * sizeof(uint64_t)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_ovf_total, 0, 2 * sizeof(uint64_t), ctx->stream));
    }
    if (!ctx->d_reasons) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_reasons, (9 + 1024 + 8) * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_reasons, 0, (9 + 1024 + 8) * sizeof(unsigned long long), ctx->stream));
    }
    long want = kListCapMin;
    while (want < total_pairs && want < kListCapMax && want < ctx->list_cap_ceiling) want <<= 1;
    const bool fixed = ctx->opt.list_cap_log2 >= 20 && ctx->opt.list_cap_log2 <= 30;
    if (fixed) want = 1L << ctx->opt.list_cap_log2;
    if (ctx->ovf_list && ctx->ovf_list2 && (ctx->list_cap == want || (ctx->list_cap > want && !fixed))) return MSSPE_OK;
    if (ctx->ovf_list || ctx->ovf_list2) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ovf_list) (void)hipFree(ctx->ovf_list);
        if (ctx->ovf_list2) (void)hipFree(ctx->ovf_list2);
        ctx->ovf_list = ctx->ovf_list2 = nullptr;
        ctx->list_cap = 0;
    }
    // The large sizes only buy fewer flushes: when the card is short of memory (the caller's own tensors), a
    // smaller pair of lists does, down to what one launch can fill.
    const long floor_cap = std::max(kListCapMin, std::min(want, kChunkPairs));
    for (;;) {
        hipError_t e1 = hipMalloc((void **)&ctx->ovf_list, sizeof(uint2) * (size_t)want);
        hipError_t e2 = e1 == hipSuccess ? hipMalloc((void **)&ctx->ovf_list2, sizeof(uint2) * (size_t)want) : e1;
        if (e1 == hipSuccess && e2 == hipSuccess) break;
        // all or nothing: a later call starts from a clean state
        if (ctx->ovf_list) (void)hipFree(ctx->ovf_list);
        if (ctx->ovf_list2) (void)hipFree(ctx->ovf_list2);
        ctx->ovf_list = ctx->ovf_list2 = nullptr;
        ctx->list_cap = 0;
        (void)hipGetLastError();
        if (fixed || want <= floor_cap) return hip_fail(ctx, e1 != hipSuccess ? e1 : e2, "hipMalloc(hand-over lists)");
        want >>= 1;
        ctx->list_cap_ceiling = want;
    }
    ctx->list_cap = want;
    return MSSPE_OK;
}

int ensure_sort(msspe_ctx *ctx, size_t ncols)
{
    if (ctx->sort_cap >= ncols) return MSSPE_OK;
    if (ctx->d_sorted) (void)hipFree(ctx->d_sorted);
    if (ctx->d_perm) (void)hipFree(ctx->d_perm);
    if (ctx->d_sort_scratch) (void)hipFree(ctx->d_sort_scratch);
    ctx->d_sorted = nullptr;
    ctx->d_perm = nullptr;
    ctx->d_sort_scratch = nullptr;
    ctx->sort_cap = 0;
    ctx->sort_scratch_bytes = pool_sort_scratch_bytes(ncols);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_sorted, sizeof(uint64_t) * ncols));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_perm, sizeof(uint32_t) * ncols));
    HIP_TRY(ctx, hipMalloc(&ctx->d_sort_scratch, ctx->sort_scratch_bytes));
    ctx->sort_cap = ncols;
    return MSSPE_OK;
}

// End of a flush: totals for the statistics, and the check that no stage's list counter went past the
// capacity of its list (entries beyond it are not stored: the host sizes the flushes so that this cannot
// happen, and a screen during which it did must not be trusted).
__global__ void k_accumulate_overflow(const uint32_t *count, uint64_t *total, uint32_t cap)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        total[0] += count[0];
        for (int q = 0; q < 7; ++q)
            if (count[q] > cap) total[1] = 1;
    }
}

}  // namespace

namespace msspe {
hipStream_t ctx_stream(msspe_ctx *ctx) { return ctx->stream; }   // group.hip: collectives are enqueued on the members' streams
}

extern "C" {

const char *msspe_version(void) { return "msspe-hip 0.1.0 (gfx950)"; }

void msspe_chem_ntthal_defaults(msspe_chem *c)
{
    if (!c) return;
    c->mv = 50.0;
    c->dv = 3.0;
    c->dntp = 0.0;
    c->dna_conc = 250.0;
    c->temp_c = 25.0;
    c->max_loop = 30;
}

void msspe_chem_primer3_defaults(msspe_chem *c)
{
    if (!c) return;
    c->mv = 50.0;
    c->dv = 1.5;
    c->dntp = 0.6;
    c->dna_conc = 50.0;
    c->temp_c = 37.0;
    c->max_loop = 30;
}

int msspe_create(int device, const char *params_path, msspe_ctx **out)
{
    if (!out) return MSSPE_ERR_ARG;
    *out = nullptr;
    msspe_ctx *ctx = new (std::nothrow) msspe_ctx();
    if (!ctx) return MSSPE_ERR_NOMEM;
    *out = ctx;   // returned even on failure so that msspe_last_error() works; destroy it anyway
    ctx->device = device;
    std::string err;
    const std::string path = params_path && *params_path ? params_path : default_bundle_path();
    if (!load_nn_tables(path, ctx->host_tb, err)) return fail(ctx, MSSPE_ERR_TABLES, err);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(ctx, MSSPE_ERR_DEVICE,
                    "no HIP device available: this engine has no CPU fallback (needs gfx950)");
    if (device < 0 || device >= ndev) return fail(ctx, MSSPE_ERR_ARG, "device ordinal out of range");
    HIP_TRY(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ctx, MSSPE_ERR_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tb, sizeof(NNTables)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_tb, &ctx->host_tb, sizeof(NNTables), hipMemcpyHostToDevice));
    // the row-specialised kernel drops its address clamp where LDS reads beyond the allocation return 0
    // (thal_pairs_row.hip kRowZero): every gfx950 seen does; a device that does not, and a context with option
    // row_oob = 0 (debugger sessions that trap on out-of-range LDS accesses), runs the general integer kernel
    HIP_TRY(ctx, pairs_row_lds_reads_zero(ctx->stream, ctx->n_cu, &ctx->lds_reads_zero));
    return MSSPE_OK;
}

int msspe_set_option(msspe_ctx *ctx, const char *key, const char *value)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!key || !value) return fail(ctx, MSSPE_ERR_ARG, "msspe_set_option: null key or value");
    const std::string k(key), v(value);
    char *end = nullptr;
    const long num = std::strtol(value, &end, 10);
    const bool is_num = end && end != value && *end == '\0';
    auto bad = [&]() { return fail(ctx, MSSPE_ERR_ARG, "msspe_set_option: bad value '" + v + "' for '" + k + "'"); };
    if (k == "pair_kernel") {
        if (v == "auto") ctx->opt.pair_kernel = 0;
        else if (v == "f64") ctx->opt.pair_kernel = 1;
        else if (v == "int") ctx->opt.pair_kernel = 2;
        else return bad();
    } else if (k == "force_generic") {
        if (!is_num || num < 0 || num > 1) return bad();
        ctx->opt.force_generic = num != 0;
    } else if (k == "split_min_k") {
        if (!is_num || num < 2 || num > 99) return bad();
        ctx->opt.split_min_k = (int)num;
    } else if (k == "wave_kernel") {
        if (!is_num || num < 0 || num > 1) return bad();
        ctx->opt.wave_kernel = num != 0;
    } else if (k == "list_cap_log2") {
        if (!is_num || !(num == 0 || (num >= 20 && num <= 30))) return bad();
        ctx->opt.list_cap_log2 = (int)num;
    } else if (k == "split_lanes") {
        if (!is_num || !(num == 0 || num == 2 || num == 4 || num == 8)) return bad();
        ctx->opt.split_lanes = (int)num;
    } else if (k == "split_list") {
        if (!is_num || num < 0 || num > 1) return bad();
        ctx->opt.split_list = num != 0;
    } else if (k == "short_chain") {
        if (!is_num || num < 0 || num > 1) return bad();
        ctx->opt.short_chain = num != 0;
    } else if (k == "self_lane_from") {
        if (!is_num || num < 0) return bad();
        ctx->opt.self_lane_from = (int)num;
    } else if (k == "row_oob") {
        if (!is_num || num < 0 || num > 1) return bad();
        ctx->opt.row_oob = num != 0;
    } else if (k == "stage_a_graph") {
        if (!is_num || num < 0 || num > 1) return bad();
        ctx->kmer.set_use_graph(num != 0);
        ctx->kmer_rev.set_use_graph(num != 0);
    } else if (k == "stage_a_candidates") {
        if (!is_num || num < 0 || num > 1) return bad();
        ctx->kmer.set_narrow_loop(num != 0);
        ctx->kmer_rev.set_narrow_loop(num != 0);
    } else {
        return fail(ctx, MSSPE_ERR_ARG, "msspe_set_option: unknown option '" + k + "'");
    }
    return MSSPE_OK;
}

int msspe_get_info(msspe_ctx *ctx, const char *key, long long *value_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!key || !value_out) return fail(ctx, MSSPE_ERR_ARG, "msspe_get_info: null key or output");
    const std::string k(key);
    if (k == "device") *value_out = ctx->device;
    else if (k == "n_cu") *value_out = ctx->n_cu;
    else if (k == "lds_reads_zero") *value_out = ctx->lds_reads_zero ? 1 : 0;
    else if (k == "row_kernel")
        *value_out = ctx->lds_reads_zero && ctx->opt.row_oob && ctx->opt.pair_kernel == 0 && !ctx->opt.force_generic ? 1 : 0;
    else if (k == "stage_a_fast_iterations") *value_out = ctx->kmer.loop_stats()[0];
    else if (k == "stage_a_general_iterations") *value_out = ctx->kmer.loop_stats()[1];
    else if (k == "stage_a_rebuilds") *value_out = ctx->kmer.loop_stats()[2];
    else if (k == "stage_a_idle_iterations") *value_out = ctx->kmer.loop_stats()[3];
    else return fail(ctx, MSSPE_ERR_ARG, "msspe_get_info: unknown key '" + k + "'");
    return MSSPE_OK;
}

int msspe_kmer_trace(msspe_ctx *ctx, uint32_t *out, int capacity, int *n_out)
{
    if (!ctx || !n_out || capacity < 0 || (capacity && !out)) return MSSPE_ERR_ARG;
    const auto &t = ctx->kmer.trace();
    *n_out = (int)t.size();
    for (int i = 0; i < capacity && i < (int)t.size(); ++i) out[i] = t[(size_t)i];
    return MSSPE_OK;
}

void msspe_destroy(msspe_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->own_stream) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->kmer.release();
        ctx->kmer_rev.release();
        if (ctx->ev_rev) (void)hipEventDestroy(ctx->ev_rev);
        if (ctx->stream_rev) {
            (void)hipStreamSynchronize(ctx->stream_rev);
            (void)hipStreamDestroy(ctx->stream_rev);
        }
        for (auto &e : ctx->chem_cache)
        {
            if (e.d_pt) (void)hipFree(e.d_pt);
            if (e.d_ft) (void)hipFree(e.d_ft);
            if (e.d_it) (void)hipFree(e.d_it);
            if (e.d_st) (void)hipFree(e.d_st);
        }
        if (ctx->wsS) (void)hipFree(ctx->wsS);
        if (ctx->wsH) (void)hipFree(ctx->wsH);
        if (ctx->ovf_list) (void)hipFree(ctx->ovf_list);
        if (ctx->ovf_list2) (void)hipFree(ctx->ovf_list2);
        if (ctx->ovf_count) (void)hipFree(ctx->ovf_count);
        if (ctx->d_ovf_total) (void)hipFree(ctx->d_ovf_total);
        if (ctx->d_reasons) (void)hipFree(ctx->d_reasons);
        if (ctx->d_sorted) (void)hipFree(ctx->d_sorted);
        if (ctx->d_perm) (void)hipFree(ctx->d_perm);
        if (ctx->d_sort_scratch) (void)hipFree(ctx->d_sort_scratch);
        if (ctx->d_tb) (void)hipFree(ctx->d_tb);
        for (auto &ev : ctx->prof_events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        (void)hipStreamDestroy(ctx->own_stream);
    }
    delete ctx;
}

const char *msspe_last_error(const msspe_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int msspe_set_stream(msspe_ctx *ctx, void *hip_stream)
{
    if (!ctx) return MSSPE_ERR_ARG;
    ctx->stream = (hipStream_t)hip_stream;   // NULL is HIP's default (null) stream
    return MSSPE_OK;
}

int msspe_reset_stream(msspe_ctx *ctx)
{
    if (!ctx) return MSSPE_ERR_ARG;
    ctx->stream = ctx->own_stream;
    return MSSPE_OK;
}

// after a synchronisation: did any stage's hand-over list run past its capacity since the last check?
static int check_list_overrun(msspe_ctx *ctx)
{
    if (!ctx->d_ovf_total) return MSSPE_OK;
    uint64_t flag = 0;
    HIP_TRY(ctx, hipMemcpy(&flag, ctx->d_ovf_total + 1, sizeof flag, hipMemcpyDeviceToHost));
    if (!flag) return MSSPE_OK;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_ovf_total + 1, 0, sizeof flag, ctx->stream));
    return fail(ctx, MSSPE_ERR_DEVICE, "a hand-over list was overrun: results of the last screen are incomplete");
}

int msspe_synchronize(msspe_ctx *ctx)
{
    if (!ctx) return MSSPE_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_list_overrun(ctx);
}

int msspe_pack_oligos(const char *ascii, int n, int k, uint64_t *packed_out)
{
    if (!ascii || !packed_out || n < 0) return MSSPE_ERR_ARG;
    if (k < 1 || k > 32) return MSSPE_ERR_K;
    for (int i = 0; i < n; ++i) {
        uint64_t w = 0;
        for (int p = 0; p < k; ++p) {
            uint64_t code;
            switch (ascii[(size_t)i * k + p]) {
            case 'A': case 'a': code = 0; break;
            case 'C': case 'c': code = 1; break;
            case 'G': case 'g': code = 2; break;
            case 'T': case 't': code = 3; break;
            default: return MSSPE_ERR_ARG;
            }
            w |= code << (2 * p);
        }
        packed_out[i] = w;
    }
    return MSSPE_OK;
}

void msspe_unpack_oligo(uint64_t packed, int k, char *ascii_out)
{
    for (int p = 0; p < k; ++p) ascii_out[p] = "ACGT"[(packed >> (2 * p)) & 3];
    ascii_out[k] = 0;
}

static int cross_dimer_impl(msspe_ctx *ctx, const uint64_t *d_pool, int n, int k, const msspe_chem *chem,
                            float dg_threshold, int row0, int row1, int col0, int col1,
                            uint32_t *d_row_conflicts, uint64_t *d_bitmap, double *d_dg, double *d_tm,
                            EdgeRecord *d_edges, unsigned long long *d_edge_count, unsigned long long edge_cap);

int msspe_cross_dimer_dev(msspe_ctx *ctx, const uint64_t *d_pool, int n, int k,
                          const msspe_chem *chem, float dg_threshold, int row0, int row1,
                          int col0, int col1, uint32_t *d_row_conflicts, uint64_t *d_bitmap,
                          double *d_dg, double *d_tm)
{
    return cross_dimer_impl(ctx, d_pool, n, k, chem, dg_threshold, row0, row1, col0, col1, d_row_conflicts,
                            d_bitmap, d_dg, d_tm, nullptr, nullptr, 0);
}

int msspe_cross_dimer_edges_dev(msspe_ctx *ctx, const uint64_t *d_pool, int n, int k, const msspe_chem *chem,
                                float dg_threshold, int row0, int row1, int col0, int col1,
                                uint32_t *d_row_conflicts, msspe_edge_dev *d_edges, uint64_t capacity,
                                uint64_t *d_count)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!d_count || (capacity && !d_edges)) return fail(ctx, MSSPE_ERR_ARG, "edge list: null count or buffer");
    static_assert(sizeof(msspe_edge_dev) == sizeof(EdgeRecord), "edge record layouts differ");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(uint64_t), ctx->stream));
    return cross_dimer_impl(ctx, d_pool, n, k, chem, dg_threshold, row0, row1, col0, col1, d_row_conflicts,
                            nullptr, nullptr, nullptr, reinterpret_cast<EdgeRecord *>(d_edges),
                            reinterpret_cast<unsigned long long *>(d_count), capacity);
}

static int cross_dimer_impl(msspe_ctx *ctx, const uint64_t *d_pool, int n, int k, const msspe_chem *chem,
                            float dg_threshold, int row0, int row1, int col0, int col1,
                            uint32_t *d_row_conflicts, uint64_t *d_bitmap, double *d_dg, double *d_tm,
                            EdgeRecord *d_edges, unsigned long long *d_edge_count, unsigned long long edge_cap)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!d_pool || !chem || n < 0) return fail(ctx, MSSPE_ERR_ARG, "null pool/chemistry");
    if (k < 2 || k > 32) return fail(ctx, MSSPE_ERR_K, "oligo length must be 2..32");
    if (row0 < 0 || row1 > n || row0 > row1 || col0 < 0 || col1 > n || col0 > col1)
        return fail(ctx, MSSPE_ERR_ARG, "row/column range outside the pool");
    if (row0 == row1 || col0 == col1) return MSSPE_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ChemEntry *ce = nullptr;
    int rc = chem_entry(ctx, *chem, dg_threshold, &ce);
    if (rc) return rc;
    if ((rc = ensure_workspace(ctx, (size_t)k * (size_t)k))) return rc;
    if ((rc = ensure_overflow(ctx, (long)(row1 - row0) * (long)(col1 - col0)))) return rc;
    const long kListCap = ctx->list_cap;

    const int ncols = col1 - col0;
    const int words = (ncols + 63) / 64;
    // long oligos: exact-integer kernel with a pair's table split over lanes (honours max_loop)
    // (also short oligos under a loop-size limit the register-table kernels do not implement)
    const bool split = !ctx->opt.force_generic && !(ctx->opt.pair_kernel == 1) && k <= ce->split_max_k &&
                       (k >= ctx->opt.split_min_k || chem->max_loop < 2 * k - 4);
    // f64, one wave per pair: behind the split kernel, and as the first stage where neither the split
    // kernel nor the register-table chain applies (29 .. 32 bases, parameter files off the grid)
    const bool wave_ok = !ctx->opt.force_generic && k <= ce->wave_max_k && ctx->opt.wave_kernel;
    const bool wave_matrix = wave_ok && !split && (k > pairs_fast_max_k() || chem->max_loop < 2 * k - 4);
    const bool fast = split || wave_matrix ||
                      (!ctx->opt.force_generic && k <= pairs_fast_max_k() && ce->fast_ok &&
                       chem->max_loop >= 2 * k - 4);   // the tuned kernel has no loop-size cut-off
    // the conflict bitmap is produced with atomic ORs: clear the caller's block first
    if (d_bitmap)
        HIP_TRY(ctx, hipMemsetAsync(d_bitmap, 0, sizeof(uint64_t) * (size_t)(row1 - row0) * (size_t)words,
                                    ctx->stream));
    PairSinks sinks;
    sinks.row_conflicts = d_row_conflicts;
    sinks.bitmap = d_bitmap;
    sinks.dg = d_dg;
    sinks.tm = d_tm;
    sinks.row0 = row0;
    sinks.col0 = col0;
    sinks.ncols = ncols;
    sinks.words = words;
    sinks.edges = d_edges;
    sinks.edge_count = d_edge_count;
    sinks.edge_cap = edge_cap;
    GenericDimerArgs g;
    std::memset(&g, 0, sizeof g);
    g.pt = ce->d_pt;
    g.c[0] = ce->c[0];
    g.c[1] = ce->c[1];
    g.pool = d_pool;
    g.k = k;
    g.mode = 1;
    g.sinks = sinks;
    g.wsS = ctx->wsS;
    g.wsH = ctx->wsH;
    g.ws_lanes = kGenericLanes;
    // A launch covers at most kChunkPairs pairs, and never more than one hand-over list holds (a fixed
    // list_cap_log2 below 29, or lists shrunk because the card is short of memory): even a launch that handed
    // every pair on cannot overrun its list.
    const long chunk_pairs = std::min(kChunkPairs, kListCap);
    // The block's rows are split evenly over as few launches as the limit allows (a 65,536-row block: nine launches of
    // about 7,296 rows rather than eight of 8,184 and one of 64; the 8,192 rows a rank of eight owns: one launch, not
    // 8,184 + 8), in whole row groups of the first-stage kernels (12 or 8 waves per block: no idle waves in the last
    // tile row of a launch that is not the block's last).
    const long max_rows = std::max(1L, chunk_pairs / ncols), n_rows_all = (long)row1 - row0;
    long rows_per_chunk = max_rows;
    if (n_rows_all > max_rows) {
        const long cap_rows = max_rows > 24 ? max_rows - max_rows % 24 : max_rows;   // the largest whole-group launch
        const long n_launch = (n_rows_all + cap_rows - 1) / cap_rows;
        rows_per_chunk = (n_rows_all + n_launch - 1) / n_launch;
        if (rows_per_chunk > 24) rows_per_chunk = std::min(cap_rows, (rows_per_chunk + 23) / 24 * 24);
    } else if (n_rows_all > 0) {
        rows_per_chunk = n_rows_all;
    }
    if (rows_per_chunk < 1) rows_per_chunk = 1;
    if (!fast) {
        // generic kernel over the whole block, a band of rows per launch (matrix mode derives
        // (row, col) from sinks.row0 / sinks.col0, so the output base pointers move with the band)
        for (int r = row0; r < row1; r += (int)rows_per_chunk) {
            const int r_end = (int)std::min<long>(row1, (long)r + rows_per_chunk);
            GenericDimerArgs gb = g;
            gb.sinks.row0 = r;
            gb.n_work = (long)(r_end - r) * (long)ncols;
            const size_t roff = (size_t)(r - row0);
            if (gb.sinks.bitmap) gb.sinks.bitmap += roff * (size_t)words;
            if (gb.sinks.dg) gb.sinks.dg += roff * (size_t)ncols;
            if (gb.sinks.tm) gb.sinks.tm += roff * (size_t)ncols;
            HIP_TRY(ctx, launch_dimer_generic(gb, ctx->stream));
        }
        return MSSPE_OK;
    }
    const bool int_stage = !split && ce->int_ok && !(ctx->opt.pair_kernel == 1);
    if (!wave_matrix) {
        if ((rc = ensure_sort(ctx, (size_t)ncols))) return rc;
        HIP_TRY(ctx, sort_columns_by_composition(d_pool, col0, ncols, k, ctx->d_sort_scratch,
                                                 ctx->sort_scratch_bytes, ctx->d_sorted, ctx->d_perm,
                                                 ctx->stream));
    }
    // Overflow pairs are collected over several launches and finished together: the list kernels
    // have a fixed latency floor, and list_cap entries cannot be overrun by list_cap / kChunkPairs
    // launches even if every pair overflowed.
    // small screens (the reference's are at most 2,000^2): the short chain behind the integer list stage
    const bool short_chain = wave_ok && ctx->opt.short_chain && (long)(row1 - row0) * (long)ncols <= (1L << 23);
    auto flush = [&]() -> int {
        PairKernelArgs a;
        a.ft = ce->d_ft;
        a.c = ce->c[0];
        a.pool = d_pool;
        a.cols_sorted = nullptr;
        a.perm = nullptr;
        a.ncols_sorted = 0;
        a.n = n;
        a.k = k;
        a.row0 = row0;
        a.row1 = row1;
        a.col0 = 0;
        a.col1 = ncols;
        a.sinks = sinks;
        a.work_counter = ctx->ovf_count + 7;
        // list A (ovf_list, counter 0) = what the first stage did not answer
        const uint2 *in_list = ctx->ovf_list;
        uint2 *out_list = ctx->ovf_list2;
        int in_c = 0, out_c = 1;
        if (split || wave_matrix) {
            // long oligos: what the split kernel handed on is answered in f64 by one wave per pair;
            // the dense kernel takes what is left (two self-complementary oligos, huge tables)
            if (split && wave_ok) {
                a.overflow_list = out_list;
                a.overflow_count = ctx->ovf_count + out_c;
                a.overflow_cap = (uint32_t)kListCap;
                HIP_TRY(ctx, launch_pairs_wave(a, ce->d_st, in_list, ctx->ovf_count + in_c, ctx->stream));
                in_list = out_list;
                in_c = out_c;
            }
            g.list = in_list;
            g.list_count = ctx->ovf_count + in_c;
            g.n_work = kListCap;
            HIP_TRY(ctx, launch_dimer_generic(g, ctx->stream));
            hipLaunchKernelGGL(k_accumulate_overflow, dim3(1), dim3(64), 0, ctx->stream, ctx->ovf_count,
                               ctx->d_ovf_total, (uint32_t)kListCap);
            HIP_TRY(ctx, hipMemsetAsync(ctx->ovf_count, 0, 8 * sizeof(uint32_t), ctx->stream));
            return MSSPE_OK;
        }
        auto advance = [&]() {   // the two buffers ping-pong: a stage's input is consumed when it ends
            in_list = out_list;
            out_list = out_list == ctx->ovf_list2 ? ctx->ovf_list : ctx->ovf_list2;
            in_c = out_c;
            out_c = out_c + 1;
        };
        if (int_stage) {
            // (1) the integer kernel again, 64 slots, lanes sorted by table size: pairs that only left
            //     their wave because of their size; (2) the 56-slot f64 table: exact ties
            a.overflow_list = out_list;
            a.overflow_count = ctx->ovf_count + out_c;
            a.overflow_cap = (uint32_t)kListCap;
            HIP_TRY(ctx, launch_pairs_int_list(a, ce->d_it, in_list, ctx->ovf_count + in_c, ctx->d_reasons,
                                               ctx->n_cu, ctx->stream));
            advance();
            if (short_chain) {
                // a reference-sized screen: what the integer list stage leaves (some ten thousand pairs) goes straight to
                // one wave per pair -- each of the three register-table stages in between has a latency floor of 0.4 ...
                // 0.6 ms whatever its list holds, the wave kernel takes 0.13 ms + 16 ns per pair
                a.overflow_list = out_list;
                a.overflow_count = ctx->ovf_count + out_c;
                HIP_TRY(ctx, launch_pairs_wave(a, ce->d_st, in_list, ctx->ovf_count + in_c, ctx->stream));
                g.list = out_list;
                g.list_count = ctx->ovf_count + out_c;
                g.n_work = kListCap;
                HIP_TRY(ctx, launch_dimer_generic(g, ctx->stream));
                hipLaunchKernelGGL(k_accumulate_overflow, dim3(1), dim3(64), 0, ctx->stream, ctx->ovf_count,
                                   ctx->d_ovf_total, (uint32_t)kListCap);
                HIP_TRY(ctx, hipMemsetAsync(ctx->ovf_count, 0, 8 * sizeof(uint32_t), ctx->stream));
                return MSSPE_OK;
            }
            if (ce->split_max_k >= k && ctx->opt.split_list) {
                // (1b) tables beyond the list stage's 63 stored cells: two lanes per pair, still exact integers
                //      (marked entries -- ties -- pass through to the f64 kernels)
                a.overflow_list = out_list;
                a.overflow_count = ctx->ovf_count + out_c;
                HIP_TRY(ctx, launch_pairs_split_list(a, ce->d_st, in_list, ctx->ovf_count + in_c, ctx->n_cu, ctx->stream));
                advance();
            }
            a.overflow_list = out_list;
            a.overflow_count = ctx->ovf_count + out_c;
            HIP_TRY(ctx, launch_pairs_main_list(a, in_list, ctx->ovf_count + in_c, ctx->stream));
            advance();
        }
        // the wide table over the list
        a.overflow_list = out_list;
        a.overflow_count = ctx->ovf_count + out_c;
        a.overflow_cap = (uint32_t)kListCap;
        HIP_TRY(ctx, launch_pairs_wide(a, in_list, ctx->ovf_count + in_c, ctx->stream));
        if (wave_ok) {
            // huge tables: one wave per pair, the table in LDS
            advance();
            a.overflow_list = out_list;
            a.overflow_count = ctx->ovf_count + out_c;
            HIP_TRY(ctx, launch_pairs_wave(a, ce->d_st, in_list, ctx->ovf_count + in_c, ctx->stream));
        }
        // last stage: whatever is left (both-self-complementary pairs)
        g.list = out_list;
        g.list_count = ctx->ovf_count + out_c;
        g.n_work = kListCap;
        HIP_TRY(ctx, launch_dimer_generic(g, ctx->stream));
        hipLaunchKernelGGL(k_accumulate_overflow, dim3(1), dim3(64), 0, ctx->stream, ctx->ovf_count,
                           ctx->d_ovf_total, (uint32_t)kListCap);
        HIP_TRY(ctx, hipMemsetAsync(ctx->ovf_count, 0, 8 * sizeof(uint32_t), ctx->stream));
        return MSSPE_OK;
    };
    HIP_TRY(ctx, hipMemsetAsync(ctx->ovf_count, 0, 8 * sizeof(uint32_t), ctx->stream));
    long pending = 0;   // worst-case entries the list may hold
    for (int r = row0; r < row1; r += (int)rows_per_chunk) {
        const int r_end = (int)std::min<long>(row1, (long)r + rows_per_chunk);
        for (long q0 = 0; q0 < ncols; q0 += chunk_pairs) {   // sorted-column index range (one chunk unless a row is longer than a launch)
            const long q_end = std::min<long>(ncols, q0 + chunk_pairs);
            const long launch_pairs = (long)(r_end - r) * (q_end - q0);
            if (pending + launch_pairs > kListCap) {
                if ((rc = flush())) return rc;
                pending = 0;
            }
            PairKernelArgs a;
            a.ft = ce->d_ft;
            a.c = ce->c[0];
            a.pool = d_pool;
            a.cols_sorted = ctx->d_sorted;
            a.perm = ctx->d_perm;
            a.ncols_sorted = ncols;
            a.n = n;
            a.k = k;
            a.row0 = r;
            a.row1 = r_end;
            a.col0 = (int)q0;
            a.col1 = (int)q_end;
            a.sinks = sinks;
            a.overflow_list = ctx->ovf_list;
            a.overflow_count = ctx->ovf_count;
            a.overflow_cap = (uint32_t)kListCap;
            a.work_counter = ctx->ovf_count + 7;   // the last of the eight stage counters
            if (ctx->prof_on) {
                if (ctx->prof_used == ctx->prof_events.size()) {
                    hipEvent_t e0, e1;
                    HIP_TRY(ctx, hipEventCreate(&e0));
                    HIP_TRY(ctx, hipEventCreate(&e1));
                    ctx->prof_events.emplace_back(e0, e1);
                }
                HIP_TRY(ctx, hipEventRecord(ctx->prof_events[ctx->prof_used].first, ctx->stream));
            }
            if (wave_matrix) {
                a.col0 = col0 + (int)q0;   // pool columns, no composition sort
                a.col1 = col0 + (int)q_end;
                HIP_TRY(ctx, launch_pairs_wave(a, ce->d_st, nullptr, nullptr, ctx->stream));
            } else if (split) HIP_TRY(ctx, launch_pairs_split(a, ce->d_st, ctx->d_reasons, ctx->opt.split_lanes, ctx->stream));
            else if (int_stage && ce->row_ok && k <= pairs_row_max_k() && ctx->opt.pair_kernel != 2 &&
                     (k > pairs_row_oob_max_k() || (ctx->lds_reads_zero && ctx->opt.row_oob)))
                HIP_TRY(ctx, launch_pairs_row(a, ce->d_it, ctx->d_reasons, ctx->n_cu, ctx->stream));
            else if (int_stage) HIP_TRY(ctx, launch_pairs_int(a, ce->d_it, ctx->d_reasons, ctx->n_cu, ctx->stream));
            else HIP_TRY(ctx, launch_pairs_fast(a, ctx->stream));
            if (ctx->prof_on)
                HIP_TRY(ctx, hipEventRecord(ctx->prof_events[ctx->prof_used++].second, ctx->stream));
            pending += launch_pairs;
        }
    }
    if (pending && (rc = flush())) return rc;
    return MSSPE_OK;
}

int msspe_profile_enable(msspe_ctx *ctx, int on)
{
    if (!ctx) return MSSPE_ERR_ARG;
    ctx->prof_on = on != 0;
    ctx->prof_used = 0;
    return MSSPE_OK;
}

int msspe_profile_read(msspe_ctx *ctx, uint64_t *launches, double *total_ms)
{
    if (!ctx || !launches || !total_ms) return MSSPE_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->prof_events[i].first, ctx->prof_events[i].second));
        sum += ms;
    }
    *launches = ctx->prof_used;
    *total_ms = sum;
    ctx->prof_used = 0;
    return MSSPE_OK;
}

int msspe_last_overflow_pairs(msspe_ctx *ctx, uint64_t *count_out)
{
    if (!ctx || !count_out) return MSSPE_ERR_ARG;
    *count_out = 0;
    if (!ctx->d_ovf_total) return MSSPE_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t both[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpy(both, ctx->d_ovf_total, sizeof both, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_ovf_total, 0, sizeof both, ctx->stream));
    *count_out = both[0];
    if (both[1]) return fail(ctx, MSSPE_ERR_DEVICE, "a hand-over list was overrun: results of the last screen are incomplete");
    return MSSPE_OK;
}

int msspe_pair_stage_stats(msspe_ctx *ctx, uint64_t out[16])
{
    if (!ctx || !out) return MSSPE_ERR_ARG;
    for (int q = 0; q < 16; ++q) out[q] = 0;
    if (!ctx->d_reasons) return MSSPE_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_reasons, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(out + 8, ctx->d_reasons + 1033, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_reasons, 0, 9 * sizeof(uint64_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_reasons + 1033, 0, 8 * sizeof(uint64_t), ctx->stream));
    return MSSPE_OK;
}

int msspe_pair_stage_samples(msspe_ctx *ctx, uint64_t *out, int capacity, int *n_out)
{
    if (!ctx || !out || !n_out || capacity < 0) return MSSPE_ERR_ARG;
    *n_out = 0;
    if (!ctx->d_reasons) return MSSPE_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t n = 0;
    HIP_TRY(ctx, hipMemcpy(&n, ctx->d_reasons + 8, sizeof n, hipMemcpyDeviceToHost));
    const int m = (int)std::min<uint64_t>(std::min<uint64_t>(n, 1024), (uint64_t)capacity);
    if (m) HIP_TRY(ctx, hipMemcpy(out, ctx->d_reasons + 9, sizeof(uint64_t) * m, hipMemcpyDeviceToHost));
    *n_out = m;
    return MSSPE_OK;
}

int msspe_cross_dimer(msspe_ctx *ctx, const char *pool_ascii, int n, int k,
                      const msspe_chem *chem, float dg_threshold, uint32_t *row_conflicts,
                      uint64_t *bitmap, double *dg, double *tm)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!pool_ascii || !chem || n < 0) return fail(ctx, MSSPE_ERR_ARG, "null pool/chemistry");
    if (n == 0) return MSSPE_OK;
    std::vector<uint64_t> packed((size_t)n);
    int rc = msspe_pack_oligos(pool_ascii, n, k, packed.data());
    if (rc) return fail(ctx, rc, rc == MSSPE_ERR_K ? "oligo length must be 1..32"
                                                   : "pool holds characters other than ACGT");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t words = ((size_t)n + 63) / 64, nn = (size_t)n * (size_t)n;
    uint64_t *d_pool = nullptr, *d_bitmap = nullptr;
    uint32_t *d_rc = nullptr;
    double *d_dg = nullptr, *d_tm = nullptr;
    auto cleanup = [&]() {
        if (d_pool) (void)hipFree(d_pool);
        if (d_bitmap) (void)hipFree(d_bitmap);
        if (d_rc) (void)hipFree(d_rc);
        if (d_dg) (void)hipFree(d_dg);
        if (d_tm) (void)hipFree(d_tm);
    };
#define TRY_OR_CLEAN(expr)                                         \
    do {                                                           \
        hipError_t e__ = (expr);                                   \
        if (e__ != hipSuccess) {                                   \
            cleanup();                                             \
            return hip_fail(ctx, e__, #expr);                      \
        }                                                          \
    } while (0)
    TRY_OR_CLEAN(hipMalloc((void **)&d_pool, sizeof(uint64_t) * (size_t)n));
    TRY_OR_CLEAN(hipMemcpy(d_pool, packed.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice));
    if (row_conflicts) {
        TRY_OR_CLEAN(hipMalloc((void **)&d_rc, sizeof(uint32_t) * (size_t)n));
        TRY_OR_CLEAN(hipMemsetAsync(d_rc, 0, sizeof(uint32_t) * (size_t)n, ctx->stream));   // on the stream the kernels run on (it is non-blocking: the null stream does not order against it)
    }
    if (bitmap) {
        TRY_OR_CLEAN(hipMalloc((void **)&d_bitmap, sizeof(uint64_t) * (size_t)n * words));
        TRY_OR_CLEAN(hipMemsetAsync(d_bitmap, 0, sizeof(uint64_t) * (size_t)n * words, ctx->stream));
    }
    if (dg) TRY_OR_CLEAN(hipMalloc((void **)&d_dg, sizeof(double) * nn));
    if (tm) TRY_OR_CLEAN(hipMalloc((void **)&d_tm, sizeof(double) * nn));
    rc = msspe_cross_dimer_dev(ctx, d_pool, n, k, chem, dg_threshold, 0, n, 0, n, d_rc, d_bitmap,
                               d_dg, d_tm);
    if (rc) {
        cleanup();
        return rc;
    }
    TRY_OR_CLEAN(hipStreamSynchronize(ctx->stream));
    if ((rc = check_list_overrun(ctx))) {
        cleanup();
        return rc;
    }
    if (row_conflicts)
        TRY_OR_CLEAN(hipMemcpy(row_conflicts, d_rc, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
    if (bitmap)
        TRY_OR_CLEAN(hipMemcpy(bitmap, d_bitmap, sizeof(uint64_t) * (size_t)n * words, hipMemcpyDeviceToHost));
    if (dg) TRY_OR_CLEAN(hipMemcpy(dg, d_dg, sizeof(double) * nn, hipMemcpyDeviceToHost));
    if (tm) TRY_OR_CLEAN(hipMemcpy(tm, d_tm, sizeof(double) * nn, hipMemcpyDeviceToHost));
    cleanup();
    return MSSPE_OK;
}

int msspe_cross_dimer_edges(msspe_ctx *ctx, const char *pool_ascii, int n, int k, const msspe_chem *chem,
                            float dg_threshold, msspe_edge *edges, uint64_t capacity, uint64_t *count_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!pool_ascii || !chem || !count_out || n < 0 || (capacity && !edges))
        return fail(ctx, MSSPE_ERR_ARG, "null pool/chemistry/count, or a capacity without a buffer");
    *count_out = 0;
    if (n == 0) return MSSPE_OK;
    std::vector<uint64_t> packed((size_t)n);
    int rc = msspe_pack_oligos(pool_ascii, n, k, packed.data());
    if (rc) return fail(ctx, rc, rc == MSSPE_ERR_K ? "oligo length must be 1..32"
                                                   : "pool holds characters other than ACGT");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint64_t *d_pool = nullptr, *d_count = nullptr;
    msspe_edge_dev *d_edges = nullptr;
    auto cleanup = [&]() {
        if (d_pool) (void)hipFree(d_pool);
        if (d_count) (void)hipFree(d_count);
        if (d_edges) (void)hipFree(d_edges);
    };
#define TRY_OR_CLEAN2(expr)                                        \
    do {                                                           \
        hipError_t e__ = (expr);                                   \
        if (e__ != hipSuccess) {                                   \
            cleanup();                                             \
            return hip_fail(ctx, e__, #expr);                      \
        }                                                          \
    } while (0)
    TRY_OR_CLEAN2(hipMalloc((void **)&d_pool, sizeof(uint64_t) * (size_t)n));
    TRY_OR_CLEAN2(hipMemcpy(d_pool, packed.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice));
    TRY_OR_CLEAN2(hipMalloc((void **)&d_count, sizeof(uint64_t)));
    if (capacity) TRY_OR_CLEAN2(hipMalloc((void **)&d_edges, sizeof(msspe_edge_dev) * (size_t)capacity));
    rc = msspe_cross_dimer_edges_dev(ctx, d_pool, n, k, chem, dg_threshold, 0, n, 0, n, nullptr, d_edges, capacity,
                                     d_count);
    if (rc) {
        cleanup();
        return rc;
    }
    TRY_OR_CLEAN2(hipStreamSynchronize(ctx->stream));
    if ((rc = check_list_overrun(ctx))) {
        cleanup();
        return rc;
    }
    uint64_t count = 0;
    TRY_OR_CLEAN2(hipMemcpy(&count, d_count, sizeof count, hipMemcpyDeviceToHost));
    *count_out = count;
    const size_t have = (size_t)std::min<uint64_t>(count, capacity);
    std::vector<msspe_edge_dev> raw(have);
    if (have) TRY_OR_CLEAN2(hipMemcpy(raw.data(), d_edges, sizeof(msspe_edge_dev) * have, hipMemcpyDeviceToHost));
    cleanup();
#undef TRY_OR_CLEAN2
    // the kernels append in no particular order: sort by (a, b) as the reference's nested loops emit them
    std::sort(raw.begin(), raw.end(), [](const msspe_edge_dev &x, const msspe_edge_dev &y) {
        return x.a != y.a ? x.a < y.a : x.b < y.b;
    });
    for (size_t e = 0; e < have; ++e) {
        edges[e].a = raw[e].a;
        edges[e].b = raw[e].b;
        // what Edge::get_dg() returns: the %g text as f32, stored as "{:.2}", parsed again (delta_g.rs:10-15, 33-46)
        edges[e].dg = round_fixed_f32((double)round_g_f32(raw[e].dg), 2);
    }
    if (count > capacity)
        return fail(ctx, MSSPE_ERR_CAPACITY, "edge list: " + std::to_string(count) + " conflict edges, capacity " +
                                                 std::to_string(capacity));
    return MSSPE_OK;
}

int msspe_thal_detail_pairs(msspe_ctx *ctx, const char *a_ascii, const char *b_ascii, int n, int k,
                            const msspe_chem *chem, int mode, msspe_thal_detail *out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!a_ascii || !b_ascii || !chem || !out || n < 0 || (mode != 1 && mode != 2))
        return fail(ctx, MSSPE_ERR_ARG, "null argument or unsupported mode (1 = ANY, 2 = END1)");
    if (k < 2 || k > 32) return fail(ctx, MSSPE_ERR_K, "oligo length must be 2..32");
    if (n == 0) return MSSPE_OK;
    static_assert(sizeof(msspe_thal_detail) == sizeof(ThalDetail), "detail layouts differ");
    std::vector<uint64_t> packed((size_t)2 * n);
    int rc = msspe_pack_oligos(a_ascii, n, k, packed.data());
    if (!rc) rc = msspe_pack_oligos(b_ascii, n, k, packed.data() + n);
    if (rc) return fail(ctx, rc, "oligos hold characters other than ACGT");
    std::vector<uint2> list((size_t)n);
    for (int i = 0; i < n; ++i) list[(size_t)i] = make_uint2((unsigned)i, (unsigned)(n + i));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ChemEntry *ce = nullptr;
    if ((rc = chem_entry(ctx, *chem, -9000.0f, &ce))) return rc;
    if ((rc = ensure_workspace(ctx, (size_t)k * (size_t)k))) return rc;
    uint64_t *d_pool = nullptr;
    uint2 *d_list = nullptr;
    ThalDetail *d_det = nullptr;
    auto cleanup = [&]() {
        if (d_pool) (void)hipFree(d_pool);
        if (d_list) (void)hipFree(d_list);
        if (d_det) (void)hipFree(d_det);
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&d_pool, sizeof(uint64_t) * 2 * (size_t)n)) != hipSuccess ||
        (e = hipMalloc((void **)&d_list, sizeof(uint2) * (size_t)n)) != hipSuccess ||
        (e = hipMalloc((void **)&d_det, sizeof(ThalDetail) * (size_t)n)) != hipSuccess ||
        (e = hipMemcpy(d_pool, packed.data(), sizeof(uint64_t) * 2 * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_list, list.data(), sizeof(uint2) * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess) {
        cleanup();
        return hip_fail(ctx, e, "thal detail buffers");
    }
    GenericDimerArgs g;
    std::memset(&g, 0, sizeof g);
    g.pt = ce->d_pt;
    g.c[0] = ce->c[0];
    g.c[1] = ce->c[1];
    g.pool = d_pool;
    g.k = k;
    g.mode = mode;
    g.list = d_list;
    g.n_work = n;
    g.detail = d_det;
    g.wsS = ctx->wsS;
    g.wsH = ctx->wsH;
    g.ws_lanes = kGenericLanes;
    e = launch_dimer_generic(g, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_det, sizeof(ThalDetail) * (size_t)n, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return hip_fail(ctx, e, "thal detail");
    return MSSPE_OK;
}

int msspe_oligo_stats_dev(msspe_ctx *ctx, const uint64_t *d_pool, int n, int k,
                          const msspe_chem *chem, double *d_tm, double *d_gc, double *d_self_any,
                          double *d_self_end, double *d_hairpin)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!d_pool || !chem || n < 0) return fail(ctx, MSSPE_ERR_ARG, "null pool/chemistry");
    if (k < 2 || k > 32) return fail(ctx, MSSPE_ERR_K, "oligo length must be 2..32");
    if (n == 0) return MSSPE_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ChemEntry *ce = nullptr;
    int rc = chem_entry(ctx, *chem, -9000.0f, &ce);   // threshold unused by the self modes
    if (rc) return rc;
    if ((rc = ensure_workspace(ctx, (size_t)(k + 1) * (size_t)(k + 1)))) return rc;
    if (d_tm || d_gc)
        HIP_TRY(ctx, launch_oligo_tm(d_pool, n, k, chem->dna_conc, chem->mv, chem->dv, chem->dntp,
                                     d_tm, d_gc, ctx->stream));
    // SELF_ANY / SELF_END (thal ANY / END1 of the oligo with itself): ONE fill of the DP serves both -- END1 is the
    // same fillMatrix with the terminal pick restricted to the last row.  What the reference's loop
    // produces (<= 2,000 oligos per call, main.rs:344): one wave per oligo (thal_pairs_wave.hip), whose latency is a
    // single oligo's.  Large pools: one LANE per oligo through the f64 register-table kernels over the list of (i, i)
    // (thal_pairs.hip, 56 then 72 slots), the wave kernel behind them for larger tables.  The dense kernel, one lane
    // per oligo over a global workspace, takes what is left (self-complementary oligos: another RC constant).
    const bool wave_ok = !ctx->opt.force_generic && k <= ce->wave_max_k && ctx->opt.wave_kernel;
    const bool lane_ok = wave_ok && n >= ctx->opt.self_lane_from && k <= pairs_fast_max_k() && ce->fast_ok &&
                         chem->max_loop >= 2 * k - 4;
    if (wave_ok && (d_self_any || d_self_end)) {
        if ((rc = ensure_overflow(ctx, n))) return rc;
        if (ctx->list_cap < n) return fail(ctx, MSSPE_ERR_NOMEM, "stage B: no memory for the work lists");
        HIP_TRY(ctx, hipMemsetAsync(ctx->ovf_count, 0, 8 * sizeof(uint32_t), ctx->stream));
        uint32_t *left = ctx->ovf_count + 3;   // entries of ovf_list2 the dense kernel has to take
        if (lane_ok) {
            HIP_TRY(ctx, launch_self_lists(ce->d_ft, ce->c[0], d_pool, n, k, d_self_any, d_self_end, ctx->ovf_list,
                                           ctx->ovf_list2, ctx->ovf_count, (uint32_t)ctx->list_cap, ctx->stream));
            HIP_TRY(ctx, launch_self_wave(ce->d_st, ce->c[0], d_pool, k, 0, n, d_self_any, d_self_end, ctx->ovf_list,
                                          ctx->ovf_count + 2, ctx->ovf_list2, left, (uint32_t)ctx->list_cap,
                                          ctx->ovf_count + 7, ctx->stream));
        } else {
            HIP_TRY(ctx, launch_self_wave(ce->d_st, ce->c[0], d_pool, k, 0, n, d_self_any, d_self_end, nullptr, nullptr,
                                          ctx->ovf_list2, left, (uint32_t)ctx->list_cap, ctx->ovf_count + 7, ctx->stream));
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        double *dst = pass == 0 ? d_self_any : d_self_end;
        if (!dst) continue;
        GenericDimerArgs g;
        std::memset(&g, 0, sizeof g);
        g.pt = ce->d_pt;
        g.c[0] = ce->c[0];
        g.c[1] = ce->c[1];
        g.pool = d_pool;
        g.k = k;
        g.mode = pass == 0 ? 1 : 2;
        g.n_work = n;
        g.self_mode = 1;
        g.self_t = dst;
        g.wsS = ctx->wsS;
        g.wsH = ctx->wsH;
        g.ws_lanes = kGenericLanes;
        if (wave_ok) {
            g.list = ctx->ovf_list2;
            g.list_count = ctx->ovf_count + 3;
        }
        HIP_TRY(ctx, launch_dimer_generic(g, ctx->stream));
    }
    if (wave_ok && (d_self_any || d_self_end)) HIP_TRY(ctx, hipMemsetAsync(ctx->ovf_count, 0, 8 * sizeof(uint32_t), ctx->stream));
    if (d_hairpin) {
        HairpinArgs h;
        h.tb = ctx->d_tb;
        h.c = make_hairpin_consts(chem->mv, chem->dv, chem->dntp, chem->temp_c + 273.15,
                                  chem->max_loop);
        h.pool = d_pool;
        h.k = k;
        h.n_work = n;
        h.out_t = d_hairpin;
        h.wsS = ctx->wsS;
        h.wsH = ctx->wsH;
        h.ws_lanes = kGenericLanes;
        // What the reference's loop produces (<= 2,000 oligos per call, main.rs:344): one wave per oligo with the DP
        // planes in LDS (thal_hairpin_wave.hip), whose latency is a single oligo's.  A pool large enough to give
        // every SIMD several full waves: one LANE per oligo over a global workspace laid out [cell][lane]
        // (kernels_generic.hip) -- measured 1,048,576 13-mers: 7.8 ms against 61.7 ms (the serial exterior-loop pass
        // and traceback run in 64 lanes instead of one); 65,536: 0.51 against 4.05; the crossing is near 8,192.
        // (Option "force_generic" takes the one-lane kernel always.)
        if (ctx->opt.force_generic || k > 32 || n >= kHairpinLaneFrom) HIP_TRY(ctx, launch_hairpin_generic(h, ctx->stream));
        else HIP_TRY(ctx, launch_hairpin_wave(h, ctx->n_cu, ctx->stream));
    }
    return MSSPE_OK;
}

int msspe_oligo_stats(msspe_ctx *ctx, const char *pool_ascii, int n, int k,
                      const msspe_chem *chem, double *tm, double *gc, double *self_any,
                      double *self_end, double *hairpin)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!pool_ascii || !chem || n < 0) return fail(ctx, MSSPE_ERR_ARG, "null pool/chemistry");
    if (n == 0) return MSSPE_OK;
    std::vector<uint64_t> packed((size_t)n);
    int rc = msspe_pack_oligos(pool_ascii, n, k, packed.data());
    if (rc) return fail(ctx, rc, rc == MSSPE_ERR_K ? "oligo length must be 1..32"
                                                   : "pool holds characters other than ACGT");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint64_t *d_pool = nullptr;
    double *d_out = nullptr;
    const size_t nb = sizeof(double) * (size_t)n;
    HIP_TRY(ctx, hipMalloc((void **)&d_pool, sizeof(uint64_t) * (size_t)n));
    hipError_t e = hipMalloc((void **)&d_out, nb * 5);
    if (e != hipSuccess) {
        (void)hipFree(d_pool);
        return hip_fail(ctx, e, "hipMalloc");
    }
    (void)hipMemcpy(d_pool, packed.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice);
    double *host[5] = {tm, gc, self_any, self_end, hairpin};
    double *dev[5];
    for (int q = 0; q < 5; ++q) dev[q] = host[q] ? d_out + (size_t)q * n : nullptr;
    rc = msspe_oligo_stats_dev(ctx, d_pool, n, k, chem, dev[0], dev[1], dev[2], dev[3], dev[4]);
    if (!rc) {
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "hipStreamSynchronize");
    }
    for (int q = 0; q < 5 && !rc; ++q)
        if (host[q]) {
            e = hipMemcpy(host[q], dev[q], nb, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = hip_fail(ctx, e, "hipMemcpy");
        }
    (void)hipFree(d_pool);
    (void)hipFree(d_out);
    return rc;
}

int msspe_kmer_candidates_dev(msspe_ctx *ctx, const uint8_t *d_seqs, int n_seq, size_t seq_len,
                              const msspe_kmer_opt *opt, int direction, uint64_t *words_out,
                              uint32_t *freq_out, int capacity, int *n_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!d_seqs || !opt || !words_out || !freq_out || !n_out || capacity < 0)
        return fail(ctx, MSSPE_ERR_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const SeqView view{d_seqs, nullptr, seq_len};
    const int rc = ctx->kmer.run(view, n_seq, seq_len, *opt, direction, words_out, freq_out,
                                 capacity, n_out, ctx->stream, err);
    if (rc) return fail(ctx, rc, err);
    return MSSPE_OK;
}

int msspe_kmer_candidates_packed_dev(msspe_ctx *ctx, const uint64_t *d_packed, int n_seq, size_t seq_len,
                                     const msspe_kmer_opt *opt, int direction, uint64_t *words_out,
                                     uint32_t *freq_out, int capacity, int *n_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!d_packed || !opt || !words_out || !freq_out || !n_out || capacity < 0)
        return fail(ctx, MSSPE_ERR_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const SeqView view{nullptr, d_packed, seq_len};
    const int rc = ctx->kmer.run(view, n_seq, seq_len, *opt, direction, words_out, freq_out,
                                 capacity, n_out, ctx->stream, err);
    if (rc) return fail(ctx, rc, err);
    return MSSPE_OK;
}

int msspe_kmer_candidates_both_packed_dev(msspe_ctx *ctx, const uint64_t *d_packed, int n_seq, size_t seq_len,
                                          const msspe_kmer_opt *opt, uint64_t *words_fwd, uint32_t *freq_fwd, int *n_fwd,
                                          uint64_t *words_rev, uint32_t *freq_rev, int *n_rev, int capacity)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!d_packed || !opt || !words_fwd || !freq_fwd || !n_fwd || !words_rev || !freq_rev || !n_rev || capacity < 0)
        return fail(ctx, MSSPE_ERR_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->stream_rev) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream_rev, hipStreamNonBlocking));
    if (!ctx->ev_rev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_rev, hipEventDisableTiming));
    // the second stream starts behind whatever the context's stream holds (the upload of the alignment)
    HIP_TRY(ctx, hipEventRecord(ctx->ev_rev, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_rev, ctx->ev_rev, 0));
    const SeqView view{nullptr, d_packed, seq_len};
    std::string err0, err1;
    int rc0 = MSSPE_OK, rc1 = MSSPE_OK;
    // The two directions are independent (main.rs:673-690 runs them one after the other); each is a chain of small
    // dependent launches with host round trips, so two host threads on two streams overlap them almost entirely.
    std::thread rev([&]() {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            rc1 = MSSPE_ERR_DEVICE;
            err1 = "hipSetDevice failed";
            return;
        }
        rc1 = ctx->kmer_rev.run(view, n_seq, seq_len, *opt, 1, words_rev, freq_rev, capacity, n_rev, ctx->stream_rev, err1);
    });
    rc0 = ctx->kmer.run(view, n_seq, seq_len, *opt, 0, words_fwd, freq_fwd, capacity, n_fwd, ctx->stream, err0);
    rev.join();
    if (rc0) return fail(ctx, rc0, err0);
    if (rc1) return fail(ctx, rc1, "direction 1: " + err1);
    return MSSPE_OK;
}

size_t msspe_packed_row_words(size_t seq_len) { return SeqView::row_words(seq_len); }

int msspe_kmer_candidates(msspe_ctx *ctx, const uint8_t *seqs, int n_seq, size_t seq_len,
                          const msspe_kmer_opt *opt, int direction, uint64_t *words_out,
                          uint32_t *freq_out, int capacity, int *n_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!seqs || n_seq < 0) return fail(ctx, MSSPE_ERR_ARG, "null sequences");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    const size_t bytes = (size_t)n_seq * seq_len;
    HIP_TRY(ctx, hipMalloc((void **)&d, bytes ? bytes : 1));
    hipError_t e = hipMemcpy(d, seqs, bytes, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? msspe_kmer_candidates_dev(ctx, d, n_seq, seq_len, opt, direction,
                                                         words_out, freq_out, capacity, n_out)
                             : hip_fail(ctx, e, "hipMemcpy");
    (void)hipFree(d);
    return rc;
}

int msspe_segment_coverage_dev(msspe_ctx *ctx, const uint8_t *d_seqs, int n_seq, size_t seq_len,
                               const msspe_kmer_opt *opt, const uint64_t *fwd_words, int n_fwd,
                               const uint64_t *rev_words, int n_rev, uint8_t *hit_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!d_seqs || !opt || !hit_out || n_fwd < 0 || n_rev < 0 || (n_fwd && !fwd_words) || (n_rev && !rev_words))
        return fail(ctx, MSSPE_ERR_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const SeqView view{d_seqs, nullptr, seq_len};
    const int rc = ctx->kmer.coverage(view, n_seq, seq_len, *opt, fwd_words, n_fwd, rev_words, n_rev,
                                      hit_out, ctx->stream, err);
    if (rc) return fail(ctx, rc, err);
    return MSSPE_OK;
}

int msspe_segment_coverage_packed_dev(msspe_ctx *ctx, const uint64_t *d_packed, int n_seq, size_t seq_len,
                                      const msspe_kmer_opt *opt, const uint64_t *fwd_words, int n_fwd,
                                      const uint64_t *rev_words, int n_rev, uint8_t *hit_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!d_packed || !opt || !hit_out || n_fwd < 0 || n_rev < 0 || (n_fwd && !fwd_words) || (n_rev && !rev_words))
        return fail(ctx, MSSPE_ERR_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    const SeqView view{nullptr, d_packed, seq_len};
    const int rc = ctx->kmer.coverage(view, n_seq, seq_len, *opt, fwd_words, n_fwd, rev_words, n_rev,
                                      hit_out, ctx->stream, err);
    if (rc) return fail(ctx, rc, err);
    return MSSPE_OK;
}

int msspe_segment_coverage(msspe_ctx *ctx, const uint8_t *seqs, int n_seq, size_t seq_len,
                           const msspe_kmer_opt *opt, const uint64_t *fwd_words, int n_fwd,
                           const uint64_t *rev_words, int n_rev, uint8_t *hit_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!seqs || n_seq < 0) return fail(ctx, MSSPE_ERR_ARG, "null sequences");
    void *d = nullptr;
    int rc = msspe_device_put(ctx, seqs, (size_t)n_seq * seq_len, &d);
    if (rc) return rc;
    rc = msspe_segment_coverage_dev(ctx, (const uint8_t *)d, n_seq, seq_len, opt, fwd_words, n_fwd, rev_words,
                                    n_rev, hit_out);
    (void)msspe_device_free(ctx, d);
    return rc;
}

int msspe_device_put(msspe_ctx *ctx, const void *host, size_t bytes, void **device_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!device_out || (bytes && !host)) return fail(ctx, MSSPE_ERR_ARG, "null argument");
    *device_out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, bytes ? bytes : 1));
    const hipError_t e = hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(ctx, e, "hipMemcpy");
    }
    *device_out = d;
    return MSSPE_OK;
}

static int put_rows_impl(msspe_ctx *ctx, const char *const *rows, const size_t *row_bytes, int n_rows,
                         size_t row_len, int pad, bool packed, void **device_out);

int msspe_device_put_rows(msspe_ctx *ctx, const char *const *rows, const size_t *row_bytes, int n_rows,
                          size_t row_len, int pad, void **device_out)
{
    return put_rows_impl(ctx, rows, row_bytes, n_rows, row_len, pad, false, device_out);
}

int msspe_device_put_rows_packed(msspe_ctx *ctx, const char *const *rows, const size_t *row_bytes, int n_rows,
                                 size_t row_len, void **device_out)
{
    return put_rows_impl(ctx, rows, row_bytes, n_rows, row_len, '-', true, device_out);
}

// packed: the ASCII rows only pass through two 16 MB device chunks; each chunk is packed on the device
// (k_pack_rows: 2-bit bases + validity bit, 3/8 of a byte per column) behind its copy, on the copy stream
static int put_rows_impl(msspe_ctx *ctx, const char *const *rows, const size_t *row_bytes, int n_rows,
                         size_t row_len, int pad, bool packed, void **device_out)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!device_out || n_rows < 0 || (n_rows && (!rows || !row_bytes))) return fail(ctx, MSSPE_ERR_ARG, "null argument");
    *device_out = nullptr;
    for (int r = 0; r < n_rows; ++r)
        if (row_bytes[r] > row_len || (row_bytes[r] && !rows[r])) return fail(ctx, MSSPE_ERR_ARG, "row longer than row_len");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t row_out = packed ? SeqView::row_words(row_len) * sizeof(uint64_t) : row_len;
    const size_t total = row_out * (size_t)n_rows;
    char *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d, total ? total : 1));
    if (!total) {
        *device_out = d;
        return MSSPE_OK;
    }
    // two pinned staging buffers: the host fills one (rows copied and padded by a few threads) while
    // the DMA engine drains the other -- no rectangular copy of the whole matrix on the host
    const size_t rows_per_chunk = std::max<size_t>(1, (size_t)(16u << 20) / std::max<size_t>(row_len, 1));
    const size_t chunk_bytes = rows_per_chunk * row_len;
    char *stage[2] = {nullptr, nullptr};
    char *dchunk[2] = {nullptr, nullptr};   // packed: device landing zone of the ASCII chunk
    hipEvent_t drained[2] = {nullptr, nullptr};
    hipStream_t copy = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&copy, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        e = hipHostMalloc((void **)&stage[b], chunk_bytes, hipHostMallocDefault);
        if (e == hipSuccess && packed) e = hipMalloc((void **)&dchunk[b], chunk_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&drained[b], hipEventDisableTiming);
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t n_threads = std::min<size_t>({(size_t)hw, 8, rows_per_chunk});
    int turn = 0;
    for (size_t r0 = 0; r0 < (size_t)n_rows && e == hipSuccess; r0 += rows_per_chunk, turn ^= 1) {
        const size_t r1 = std::min<size_t>((size_t)n_rows, r0 + rows_per_chunk);
        e = hipEventSynchronize(drained[turn]);   // the copy that last read this buffer (none: returns at once)
        if (e != hipSuccess) break;
        char *buf = stage[turn];
        auto fill = [&](size_t a, size_t b) {
            for (size_t r = a; r < b; ++r) {
                char *dst = buf + (r - r0) * row_len;
                if (row_bytes[r]) std::memcpy(dst, rows[r], row_bytes[r]);
                std::memset(dst + row_bytes[r], pad, row_len - row_bytes[r]);
            }
        };
        if (n_threads < 2) {
            fill(r0, r1);
        } else {
            std::vector<std::thread> pool;
            for (size_t t = 0; t < n_threads; ++t)
                pool.emplace_back(fill, r0 + (r1 - r0) * t / n_threads, r0 + (r1 - r0) * (t + 1) / n_threads);
            for (auto &th : pool) th.join();
        }
        if (packed) {
            e = hipMemcpyAsync(dchunk[turn], buf, (r1 - r0) * row_len, hipMemcpyHostToDevice, copy);
            if (e == hipSuccess)
                e = launch_pack_rows((const uint8_t *)dchunk[turn], (int)(r1 - r0), row_len,
                                     (uint64_t *)(d + r0 * row_out), copy);
        } else {
            e = hipMemcpyAsync(d + r0 * row_len, buf, (r1 - r0) * row_len, hipMemcpyHostToDevice, copy);
        }
        if (e == hipSuccess) e = hipEventRecord(drained[turn], copy);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(copy);
    for (int b = 0; b < 2; ++b) {
        if (drained[b]) (void)hipEventDestroy(drained[b]);
        if (stage[b]) (void)hipHostFree(stage[b]);
        if (dchunk[b]) (void)hipFree(dchunk[b]);
    }
    if (copy) (void)hipStreamDestroy(copy);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(ctx, e, "msspe_device_put_rows");
    }
    *device_out = d;
    return MSSPE_OK;
}

int msspe_device_free(msspe_ctx *ctx, void *device)
{
    if (!ctx) return MSSPE_ERR_ARG;
    if (!device) return MSSPE_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(device));
    return MSSPE_OK;
}

int msspe_host_pair_tables(const char *params_path, const msspe_chem *chem, float dg_threshold,
                           double *fast_S, int32_t *fast_H, int32_t *int_g, int32_t *int_T,
                           double consts[8])
{
    // host only (no device needed): what the all-pairs kernels keep in LDS, for the CPU tests
    if (!chem || !fast_S || !fast_H || !int_g || !int_T || !consts) return MSSPE_ERR_ARG;
    auto tb = std::make_unique<NNTables>();
    std::string err;
    const std::string path = params_path && *params_path ? params_path : default_bundle_path();
    if (!load_nn_tables(path, *tb, err)) return MSSPE_ERR_TABLES;
    const ThalConsts c = make_dimer_consts(chem->mv, chem->dv, chem->dntp, chem->dna_conc, chem->temp_c,
                                           chem->max_loop, false, dg_threshold);
    auto pt = std::make_unique<PairTables>();
    if (!build_pair_tables(*tb, c, *pt, err)) return MSSPE_ERR_TABLES;
    auto ft = std::make_unique<FastTables>();
    auto it = std::make_unique<IntTables>();
    const bool fast_ok = build_fast_tables(*tb, *pt, pairs_fast_max_k(), *ft);
    const bool int_ok = fast_ok && build_int_tables(*ft, pairs_fast_max_k(), *it);
    std::memcpy(fast_S, ft->S, sizeof ft->S);
    std::memcpy(fast_H, ft->H, sizeof ft->H);
    std::memcpy(int_g, it->g, sizeof it->g);
    std::memcpy(int_T, it->T, sizeof it->T);
    consts[0] = c.init_S;
    consts[1] = c.RC;
    consts[2] = c.salt;
    consts[3] = c.temp_k;
    consts[4] = c.g_cut;
    consts[5] = fast_ok ? 1.0 : 0.0;
    consts[6] = int_ok ? 1.0 : 0.0;
    consts[7] = (double)FastTables::kCount;
    return MSSPE_OK;
}

int msspe_host_split_tables(const char *params_path, const msspe_chem *chem, double *S, int32_t *H,
                            int32_t *g, int32_t *L, int32_t *X, int32_t info[4])
{
    // host only: what the long-oligo kernel keeps in LDS (csrc/split_tables.hpp)
    if (!chem || !S || !H || !g || !L || !X || !info) return MSSPE_ERR_ARG;
    auto tb = std::make_unique<NNTables>();
    std::string err;
    const std::string path = params_path && *params_path ? params_path : default_bundle_path();
    if (!load_nn_tables(path, *tb, err)) return MSSPE_ERR_TABLES;
    const ThalConsts c = make_dimer_consts(chem->mv, chem->dv, chem->dntp, chem->dna_conc, chem->temp_c,
                                           chem->max_loop, false, -9000.0f);
    auto pt = std::make_unique<PairTables>();
    if (!build_pair_tables(*tb, c, *pt, err)) return MSSPE_ERR_TABLES;
    auto st = std::make_unique<SplitTables>();
    build_split_tables(*pt, chem->max_loop, *st);
    std::memcpy(S, st->S, sizeof st->S);
    std::memcpy(H, st->H, sizeof st->H);
    std::memcpy(g, st->g, sizeof st->g);
    std::memcpy(L, st->L, sizeof st->L);
    std::memcpy(X, st->X, sizeof st->X);
    info[0] = st->usable;
    info[1] = st->max_k;
    info[2] = SplitTables::kCount;
    info[3] = SplitTables::kXCount;
    return MSSPE_OK;
}

float msspe_round_g_f32(double x) { return round_g_f32(x); }
float msspe_round_fixed_f32(double x, int decimals) { return round_fixed_f32(x, decimals); }
double msspe_g_cut(float threshold) { return g_cut(threshold); }

}  // extern "C"
