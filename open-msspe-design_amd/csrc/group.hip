// group.hip -- several devices of one node behind the C ABI: one process, one context and one host thread per
// device, the candidate pool assembled everywhere by ONE all-gather and the per-primer conflict counts merged
// by ONE all-reduce (the loop that shards is the reference's N^2 "a,b" line generator,
// /root/reference/od-msspe/src/delta_g.rs:61-81, called at main.rs:739-752; stage B shards over oligos,
// primer.rs:143-166).  Bitmaps and edge lists stay with the member that produced them until the host asks.
//
// Which rows a member screens: rows are dealt out in GROUPS OF 256, round robin (row r belongs to member
// (r / 256) mod N).  The cost of a row depends on its base composition (tables differ by tens of per cent), and
// pools that come out of stage A are ordered by frequency, i.e. not randomly: contiguous blocks would give the
// members unequal work, interleaved groups give every member a sample of the whole pool.
//
// How a member screens a scattered row set with kernels that take a contiguous row range: it appends its rows
// to its copy of the pool -- P' = [ pool (n, padded to the all-gather's size) | the member's rows (m) ] -- and
// asks its context for rows [n', n' + m) x columns [0, n) of P'.  Columns keep their pool indices (bitmap bits,
// edge.b), row j of the block is pool row rows[j]; counts and edge.a are mapped back through rows[].
//
// The collectives sit behind a two-function interface (all_gather_u64, all_reduce_sum_u32) with two
// implementations: RCCL over xGMI (librccl.so, loaded on first use: a process that screens on one device never
// maps it) for distinct devices, and plain device copies + a summing kernel -- for members that share a card
// (RCCL refuses two ranks per device; the tests rehearse the whole path that way on one GPU) and as the fallback
// when RCCL cannot be loaded.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/msspe_hip.h"
#include "kernels.hpp"

namespace msspe {
hipStream_t ctx_stream(msspe_ctx *ctx);   // capi.cpp
}

namespace {

constexpr int kGroupRows = 256;   // rows dealt to a member at a time

// ---- RCCL, resolved at run time (types as in <rccl/rccl.h>: ncclUint32 = 3, ncclUint64 = 5, ncclSum = 0) ------------
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    // `only`: try this library name alone (msspe_group_rccl_available's argument); else the usual names
    bool load(std::string &err, const char *only = nullptr)
    {
        std::string why;
        auto attempt = [&](const char *name) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!lib) {
                const char *e = dlerror();   // one call: dlerror() clears the message it returns
                why = e ? e : "?";
            }
            return lib != nullptr;
        };
        if (only) attempt(only);
        else
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if (attempt(name)) break;
        if (!lib) {
            err = std::string(only ? only : "librccl.so") + " not loadable: " + why;
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllGather || !AllReduce || !GroupStart || !GroupEnd || !GetErrorString) {
            err = "librccl.so lacks a collective entry point";
            return false;
        }
        return true;
    }
};
constexpr int kNcclUint32 = 3, kNcclUint64 = 5, kNcclSum = 0;

__global__ void k_take_rows(uint64_t *pool, const uint32_t *rows, int m, size_t at)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) pool[at + j] = pool[rows[j]];
}
__global__ void k_counts_home(const uint32_t *block_counts, const uint32_t *rows, int m, uint32_t *full)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) full[rows[j]] = block_counts[j];
}
__global__ void k_edges_home(msspe_edge_dev *edges, unsigned long long n_edges, const uint32_t *rows, uint32_t row_base)
{
    const unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) edges[e].a = rows[edges[e].a - row_base];
}
__global__ void k_add_u32(uint32_t *acc, const uint32_t *src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += src[i];
}

struct Member {
    int device = 0;
    msspe_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;       // the context's own stream
    // work buffers, grown on demand
    uint64_t *shard = nullptr;          // what this member contributes to the all-gather
    uint64_t *pool = nullptr;           // P' = gathered pool | this member's rows
    uint32_t *rows = nullptr;           // pool indices of this member's rows
    uint32_t *block_counts = nullptr;   // counts as the context writes them (indexed by P' row)
    uint32_t *counts = nullptr;         // counts by pool index (what is all-reduced)
    uint32_t *scratch = nullptr;        // copy transport: another member's counts
    msspe_edge_dev *edges = nullptr;    // edge records of the member's rows, kept across calls
    uint64_t *edge_count = nullptr;
    size_t shard_cap = 0, pool_cap = 0, rows_cap = 0, counts_cap = 0, block_cap = 0, scratch_cap = 0, edges_cap = 0,
           edge_count_cap = 0;
    std::vector<uint32_t> h_rows;
    int rc = 0;
    std::string err;
};

}  // namespace

struct msspe_group {
    std::vector<Member> mem;
    std::string transport = "single";   // "single" | "rccl" | "device-copy"
    std::string transport_reason;       // why the copies run where RCCL was wanted ("" otherwise)
    Rccl rccl;
    std::vector<void *> comms;
    std::string err;
};

namespace {

int gfail(msspe_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg;
    return code;
}

#define G_HIP(g, expr)                                                                       \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) return gfail((g), MSSPE_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define M_HIP(m, expr)                                                    \
    do {                                                                  \
        hipError_t e__ = (expr);                                          \
        if (e__ != hipSuccess) {                                          \
            (m).rc = MSSPE_ERR_DEVICE;                                    \
            (m).err = std::string(#expr) + ": " + hipGetErrorString(e__); \
            return;                                                       \
        }                                                                 \
    } while (0)

template <class T>
hipError_t grow(T *&p, size_t &cap, size_t want)
{
    if (cap >= want) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const hipError_t e = hipMalloc((void **)&p, sizeof(T) * want);
    if (e == hipSuccess) cap = want;
    return e;
}

// every member runs `fn(member index)` on its own host thread (hipSetDevice is per thread); the first failure wins
template <class F>
int on_every_member(msspe_group *g, F fn)
{
    std::vector<std::thread> th;
    for (size_t m = 0; m < g->mem.size(); ++m) {
        g->mem[m].rc = 0;
        th.emplace_back([g, m, &fn]() {
            if (hipSetDevice(g->mem[m].device) != hipSuccess) {
                g->mem[m].rc = MSSPE_ERR_DEVICE;
                g->mem[m].err = "hipSetDevice failed";
                return;
            }
            fn((int)m);
        });
    }
    for (auto &t : th) t.join();
    for (size_t m = 0; m < g->mem.size(); ++m)
        if (g->mem[m].rc) return gfail(g, g->mem[m].rc, "member " + std::to_string(m) + " (device " +
                                                             std::to_string(g->mem[m].device) + "): " + g->mem[m].err);
    return MSSPE_OK;
}

int sync_all(msspe_group *g)
{
    for (auto &m : g->mem) {
        G_HIP(g, hipSetDevice(m.device));
        G_HIP(g, hipStreamSynchronize(m.stream));
    }
    return MSSPE_OK;
}

// ---- the collective layer ---------------------------------------------------------------------------------------
// all_gather_u64: every member's `shard` (count entries) -> every member's pool[m * count ...), m = 0 .. N-1.
int all_gather_u64(msspe_group *g, size_t count)
{
    const size_t N = g->mem.size();
    if (g->transport == "rccl") {
        int rc = g->rccl.GroupStart();
        for (size_t m = 0; m < N && rc == 0; ++m)
            rc = g->rccl.AllGather(g->mem[m].shard, g->mem[m].pool, count, kNcclUint64, g->comms[m], g->mem[m].stream);
        const int rc2 = g->rccl.GroupEnd();
        if (rc || rc2) return gfail(g, MSSPE_ERR_DEVICE, std::string("ncclAllGather: ") + g->rccl.GetErrorString(rc ? rc : rc2));
        return MSSPE_OK;
    }
    // device copies: the sources must be complete before another member's stream reads them
    int rc = sync_all(g);
    if (rc) return rc;
    for (size_t d = 0; d < N; ++d) {
        G_HIP(g, hipSetDevice(g->mem[d].device));
        for (size_t s = 0; s < N; ++s)
            G_HIP(g, hipMemcpyAsync(g->mem[d].pool + s * count, g->mem[s].shard, sizeof(uint64_t) * count,
                                    hipMemcpyDeviceToDevice, g->mem[d].stream));
    }
    return MSSPE_OK;
}

// all_reduce_sum_u32: every member's counts[0 .. count) := the sum over the members
int all_reduce_sum_u32(msspe_group *g, size_t count)
{
    const size_t N = g->mem.size();
    if (N == 1 && g->transport != "rccl") return MSSPE_OK;   // (one RCCL rank still goes through RCCL: the tests' rehearsal)
    if (g->transport == "rccl") {
        int rc = g->rccl.GroupStart();
        for (size_t m = 0; m < N && rc == 0; ++m)
            rc = g->rccl.AllReduce(g->mem[m].counts, g->mem[m].counts, count, kNcclUint32, kNcclSum, g->comms[m],
                                   g->mem[m].stream);
        const int rc2 = g->rccl.GroupEnd();
        if (rc || rc2) return gfail(g, MSSPE_ERR_DEVICE, std::string("ncclAllReduce: ") + g->rccl.GetErrorString(rc ? rc : rc2));
        return MSSPE_OK;
    }
    int rc = sync_all(g);
    if (rc) return rc;
    Member &root = g->mem[0];
    G_HIP(g, hipSetDevice(root.device));
    const int grid = (int)((count + 255) / 256);
    for (size_t s = 1; s < N; ++s) {
        G_HIP(g, hipMemcpyAsync(root.scratch, g->mem[s].counts, sizeof(uint32_t) * count, hipMemcpyDeviceToDevice, root.stream));
        hipLaunchKernelGGL(k_add_u32, dim3(grid), dim3(256), 0, root.stream, root.counts, root.scratch, count);
    }
    G_HIP(g, hipStreamSynchronize(root.stream));
    for (size_t d = 1; d < N; ++d) {
        G_HIP(g, hipSetDevice(g->mem[d].device));
        G_HIP(g, hipMemcpyAsync(g->mem[d].counts, root.counts, sizeof(uint32_t) * count, hipMemcpyDeviceToDevice, g->mem[d].stream));
    }
    return MSSPE_OK;
}

void rows_of(int n, int n_members, int member, std::vector<uint32_t> &out)
{
    out.clear();
    for (int g0 = member * kGroupRows; g0 < n; g0 += n_members * kGroupRows)
        for (int r = g0; r < std::min(n, g0 + kGroupRows); ++r) out.push_back((uint32_t)r);
}

// Steps 1-4 of a screen: pack, upload the shards, all-gather, append each member's rows.  After it every member
// holds P' and its row list; *n_base_out = index of the first appended row.
int assemble_pool(msspe_group *g, const char *pool_ascii, int n, int k, size_t *n_base_out)
{
    const size_t N = g->mem.size();
    std::vector<uint64_t> packed((size_t)n);
    const int prc = msspe_pack_oligos(pool_ascii, n, k, packed.data());
    if (prc) return gfail(g, prc, prc == MSSPE_ERR_K ? "oligo length must be 1..32" : "pool holds characters other than ACGT");
    const size_t shard = ((size_t)n + N - 1) / N;   // every member contributes `shard` entries (the last ones padded)
    const size_t n_base = shard * N;
    *n_base_out = n_base;
    packed.resize(n_base, 0);
    int rc = on_every_member(g, [&](int mi) {
        Member &m = g->mem[(size_t)mi];
        rows_of(n, (int)N, mi, m.h_rows);
        M_HIP(m, grow(m.shard, m.shard_cap, shard));
        M_HIP(m, grow(m.pool, m.pool_cap, n_base + m.h_rows.size() + 1));
        M_HIP(m, grow(m.rows, m.rows_cap, m.h_rows.size() + 1));
        M_HIP(m, grow(m.counts, m.counts_cap, n_base));
        M_HIP(m, grow(m.block_counts, m.block_cap, n_base + m.h_rows.size() + 1));   // indexed by P' row
        if (mi == 0 && N > 1) M_HIP(m, grow(m.scratch, m.scratch_cap, n_base));
        // the member's contribution: the candidates "it produced" = its contiguous slice of the host's pool
        M_HIP(m, hipMemcpyAsync(m.shard, packed.data() + (size_t)mi * shard, sizeof(uint64_t) * shard,
                                hipMemcpyHostToDevice, m.stream));
        if (!m.h_rows.empty())
            M_HIP(m, hipMemcpyAsync(m.rows, m.h_rows.data(), sizeof(uint32_t) * m.h_rows.size(), hipMemcpyHostToDevice,
                                    m.stream));
        M_HIP(m, hipStreamSynchronize(m.stream));   // `packed` is the host's
    });
    if (rc) return rc;
    if ((rc = all_gather_u64(g, shard))) return rc;
    return on_every_member(g, [&](int mi) {
        Member &m = g->mem[(size_t)mi];
        const int mr = (int)m.h_rows.size();
        if (mr) hipLaunchKernelGGL(k_take_rows, dim3((mr + 255) / 256), dim3(256), 0, m.stream, m.pool, m.rows, mr, n_base);
        M_HIP(m, hipGetLastError());
    });
}

// The first collective of a group, with a known answer (msspe_group_create): false + `why` when RCCL's grouped
// all-reduce fails or returns anything but the sum, so that "auto" can fall back to the copies before a screen runs.
bool rccl_self_test(msspe_group *g, std::string &why)
{
    const size_t N = g->mem.size();
    std::vector<uint32_t> got(N, 0);
    for (size_t m = 0; m < N; ++m) {
        Member &mb = g->mem[m];
        const uint32_t v = 1u + (uint32_t)m;
        if (hipSetDevice(mb.device) != hipSuccess || grow(mb.counts, mb.counts_cap, 64) != hipSuccess ||
            hipMemcpyAsync(mb.counts, &v, sizeof v, hipMemcpyHostToDevice, mb.stream) != hipSuccess ||
            hipStreamSynchronize(mb.stream) != hipSuccess) {
            why = "RCCL self-test: device buffer on member " + std::to_string(m);
            return false;
        }
    }
    if (all_reduce_sum_u32(g, 1) != MSSPE_OK) {
        why = "RCCL self-test: " + g->err;
        return false;
    }
    for (size_t m = 0; m < N; ++m) {
        Member &mb = g->mem[m];
        if (hipSetDevice(mb.device) != hipSuccess || hipStreamSynchronize(mb.stream) != hipSuccess ||
            hipMemcpy(&got[m], mb.counts, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            why = std::string("RCCL self-test: the all-reduce did not complete on member ") + std::to_string(m) + ": " +
                  hipGetErrorString(hipGetLastError());
            return false;
        }
        if (got[m] != (uint32_t)(N * (N + 1) / 2)) {
            why = "RCCL self-test: all-reduce returned " + std::to_string(got[m]) + " on member " + std::to_string(m) +
                  ", expected " + std::to_string(N * (N + 1) / 2);
            return false;
        }
    }
    return true;
}

}  // namespace

extern "C" {

int msspe_group_create(const int *devices, int n_devices, const char *params_path, const char *transport,
                       msspe_group **out)
{
    if (!out) return MSSPE_ERR_ARG;
    *out = nullptr;
    msspe_group *g = new (std::nothrow) msspe_group();
    if (!g) return MSSPE_ERR_NOMEM;
    *out = g;   // returned even on failure so that msspe_group_last_error() works; destroy it anyway
    if (!devices || n_devices < 1 || n_devices > 64) return gfail(g, MSSPE_ERR_ARG, "msspe_group_create: 1 .. 64 devices");
    const std::string want = transport && *transport ? transport : "auto";
    if (want != "auto" && want != "rccl" && want != "device-copy")
        return gfail(g, MSSPE_ERR_ARG, "msspe_group_create: transport must be auto, rccl or device-copy");
    g->mem.resize((size_t)n_devices);
    bool distinct = true;
    for (int m = 0; m < n_devices; ++m) {
        g->mem[(size_t)m].device = devices[m];
        for (int q = 0; q < m; ++q) distinct = distinct && devices[q] != devices[m];
    }
    for (auto &m : g->mem) {
        const int rc = msspe_create(m.device, params_path, &m.ctx);
        if (rc) return gfail(g, rc, std::string("msspe_create(device ") + std::to_string(m.device) + "): " +
                                        (m.ctx ? msspe_last_error(m.ctx) : "out of memory"));
        m.stream = msspe::ctx_stream(m.ctx);
    }
    if (want == "rccl" && !distinct)
        return gfail(g, MSSPE_ERR_ARG, "msspe_group_create: RCCL needs distinct devices (one rank per device)");
    if (want == "rccl" || (want == "auto" && distinct && n_devices > 1)) {
        std::string why;
        bool ok = g->rccl.load(why);
        if (ok) {
            g->comms.assign((size_t)n_devices, nullptr);
            const int rc = g->rccl.CommInitAll(g->comms.data(), n_devices, devices);
            if (rc) {
                ok = false;
                why = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(rc);
                g->comms.clear();
            }
        }
        // a first grouped collective with a known answer, before any screen depends on the fabric: every member
        // contributes 1 + its index, every member must read N (N + 1) / 2 back
        if (ok) {
            g->transport = "rccl";
            ok = rccl_self_test(g, why);
            if (!ok) {
                for (void *c : g->comms)
                    if (c) (void)g->rccl.CommDestroy(c);
                g->comms.clear();
            }
        }
        if (ok) g->transport = "rccl";
        else if (want == "rccl") return gfail(g, MSSPE_ERR_DEVICE, why);
        else {
            g->transport = "device-copy";   // auto: the copies work between any two devices of a node
            g->transport_reason = why;
        }
    } else {
        g->transport = n_devices > 1 ? "device-copy" : "single";
    }
    if (g->transport == "device-copy" && distinct && n_devices > 1) {
        // copies between devices go peer to peer where the node allows it (xGMI), staged otherwise
        for (auto &a : g->mem)
            for (auto &b : g->mem)
                if (a.device != b.device) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can) {
                        (void)hipSetDevice(a.device);
                        (void)hipDeviceEnablePeerAccess(b.device, 0);
                        (void)hipGetLastError();   // "already enabled" is fine
                    }
                }
    }
    return MSSPE_OK;
}

void msspe_group_destroy(msspe_group *g)
{
    if (!g) return;
    for (auto &m : g->mem) {
        if (!m.ctx) continue;
        (void)hipSetDevice(m.device);
        if (m.stream) (void)hipStreamSynchronize(m.stream);
    }
    for (void *c : g->comms)
        if (c && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(c);
    for (auto &m : g->mem) {
        (void)hipSetDevice(m.device);
        for (void *p : {(void *)m.shard, (void *)m.pool, (void *)m.rows, (void *)m.block_counts, (void *)m.counts, (void *)m.scratch,
                        (void *)m.edges, (void *)m.edge_count})
            if (p) (void)hipFree(p);
        if (m.ctx) msspe_destroy(m.ctx);
    }
    // (librccl stays mapped: unloading a library that owns threads and device state is not safe)
    delete g;
}

const char *msspe_group_last_error(const msspe_group *g) { return g ? g->err.c_str() : "null group"; }
int msspe_group_size(const msspe_group *g) { return g ? (int)g->mem.size() : 0; }
const char *msspe_group_transport(const msspe_group *g) { return g ? g->transport.c_str() : ""; }
const char *msspe_group_transport_reason(const msspe_group *g) { return g ? g->transport_reason.c_str() : ""; }

int msspe_group_rccl_available(const char *library, char *why, int why_capacity)
{
    Rccl r;
    std::string err;
    const bool ok = r.load(err, library && *library ? library : nullptr);
    if (why && why_capacity > 0) {
        std::strncpy(why, ok ? "" : err.c_str(), (size_t)why_capacity - 1);
        why[why_capacity - 1] = 0;
    }
    return ok ? 1 : 0;   // (a library that loaded stays mapped, as in msspe_group_create)
}
msspe_ctx *msspe_group_member(msspe_group *g, int member)
{
    return g && member >= 0 && member < (int)g->mem.size() ? g->mem[(size_t)member].ctx : nullptr;
}

int msspe_group_set_option(msspe_group *g, const char *key, const char *value)
{
    if (!g) return MSSPE_ERR_ARG;
    for (auto &m : g->mem) {
        const int rc = msspe_set_option(m.ctx, key, value);
        if (rc) return gfail(g, rc, msspe_last_error(m.ctx));
    }
    return MSSPE_OK;
}

int msspe_group_rows(int n, int n_members, int member, uint32_t *rows_out, int capacity, int *n_rows_out)
{
    if (!n_rows_out || n < 0 || n_members < 1 || member < 0 || member >= n_members || (capacity > 0 && !rows_out))
        return MSSPE_ERR_ARG;
    std::vector<uint32_t> r;
    rows_of(n, n_members, member, r);
    *n_rows_out = (int)r.size();
    if ((int)r.size() > capacity) return capacity > 0 ? MSSPE_ERR_CAPACITY : MSSPE_OK;
    std::copy(r.begin(), r.end(), rows_out);
    return MSSPE_OK;
}

int msspe_cross_dimer_group(msspe_group *g, const char *pool_ascii, int n, int k, const msspe_chem *chem,
                            float dg_threshold, uint32_t *row_conflicts, uint64_t *bitmap)
{
    if (!g) return MSSPE_ERR_ARG;
    if (!pool_ascii || !chem || n < 0) return gfail(g, MSSPE_ERR_ARG, "null pool/chemistry");
    if (n == 0) return MSSPE_OK;
    size_t n_base = 0;
    int rc = assemble_pool(g, pool_ascii, n, k, &n_base);
    if (rc) return rc;
    const size_t words = ((size_t)n + 63) / 64;
    std::vector<uint64_t *> d_bitmap(g->mem.size(), nullptr);
    rc = on_every_member(g, [&](int mi) {
        Member &m = g->mem[(size_t)mi];
        const int mr = (int)m.h_rows.size();
        M_HIP(m, hipMemsetAsync(m.counts, 0, sizeof(uint32_t) * n_base, m.stream));
        if (!mr) return;
        M_HIP(m, hipMemsetAsync(m.block_counts, 0, sizeof(uint32_t) * (n_base + (size_t)mr), m.stream));
        if (bitmap) M_HIP(m, hipMalloc((void **)&d_bitmap[(size_t)mi], sizeof(uint64_t) * (size_t)mr * words));
        const int prc = msspe_cross_dimer_dev(m.ctx, m.pool, (int)(n_base + (size_t)mr), k, chem, dg_threshold, (int)n_base,
                                              (int)n_base + mr, 0, n, m.block_counts, d_bitmap[(size_t)mi], nullptr, nullptr);
        if (prc) {
            m.rc = prc;
            m.err = msspe_last_error(m.ctx);
            return;
        }
        hipLaunchKernelGGL(k_counts_home, dim3((mr + 255) / 256), dim3(256), 0, m.stream, m.block_counts + n_base, m.rows, mr,
                           m.counts);
        M_HIP(m, hipGetLastError());
    });
    if (!rc) rc = all_reduce_sum_u32(g, n_base);
    if (!rc)
        rc = on_every_member(g, [&](int mi) {
            Member &m = g->mem[(size_t)mi];
            const int src = msspe_synchronize(m.ctx);   // also: was a hand-over list overrun?
            if (src) {
                m.rc = src;
                m.err = msspe_last_error(m.ctx);
                return;
            }
            if (mi == 0 && row_conflicts)
                M_HIP(m, hipMemcpy(row_conflicts, m.counts, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
            if (bitmap && d_bitmap[(size_t)mi]) {
                // the member's rows come in runs of up to 256 consecutive pool rows: one copy per run
                const size_t mr = m.h_rows.size();
                for (size_t j = 0; j < mr; j += (size_t)kGroupRows) {
                    const size_t len = std::min((size_t)kGroupRows, mr - j);
                    M_HIP(m, hipMemcpy(bitmap + (size_t)m.h_rows[j] * words, d_bitmap[(size_t)mi] + j * words,
                                       sizeof(uint64_t) * len * words, hipMemcpyDeviceToHost));
                }
            }
        });
    for (size_t mi = 0; mi < g->mem.size(); ++mi)
        if (d_bitmap[mi]) {
            (void)hipSetDevice(g->mem[mi].device);
            (void)hipFree(d_bitmap[mi]);
        }
    return rc;
}

int msspe_cross_dimer_edges_group(msspe_group *g, const char *pool_ascii, int n, int k, const msspe_chem *chem,
                                  float dg_threshold, msspe_edge *edges, uint64_t capacity, uint64_t *count_out)
{
    if (!g) return MSSPE_ERR_ARG;
    if (!pool_ascii || !chem || !count_out || n < 0 || (capacity && !edges))
        return gfail(g, MSSPE_ERR_ARG, "null pool/chemistry/count, or a capacity without a buffer");
    *count_out = 0;
    if (n == 0) return MSSPE_OK;
    size_t n_base = 0;
    int rc = assemble_pool(g, pool_ascii, n, k, &n_base);
    if (rc) return rc;
    const size_t N = g->mem.size();
    std::vector<std::vector<msspe_edge_dev>> got(N);
    std::vector<uint64_t> counts(N, 0);
    rc = on_every_member(g, [&](int mi) {
        Member &m = g->mem[(size_t)mi];
        const int mr = (int)m.h_rows.size();
        if (!mr) {
            // nothing to screen, but this member's stream still carries the all-gather's copies out of the other
            // members' shards: they must be done before the next call overwrites those shards
            M_HIP(m, hipStreamSynchronize(m.stream));
            return;
        }
        // the member's edge records stay allocated across calls (grown on demand): a member may hold up to
        // `capacity` edges -- which rows conflict is not known beforehand -- but on its own device
        M_HIP(m, grow(m.edge_count, m.edge_count_cap, 1));
        if (capacity) M_HIP(m, grow(m.edges, m.edges_cap, (size_t)capacity));
        msspe_edge_dev *d_edges = capacity ? m.edges : nullptr;
        uint64_t *d_count = m.edge_count;
        int prc = msspe_cross_dimer_edges_dev(m.ctx, m.pool, (int)(n_base + (size_t)mr), k, chem, dg_threshold, (int)n_base,
                                              (int)n_base + mr, 0, n, nullptr, d_edges, capacity, d_count);
        if (!prc) prc = msspe_synchronize(m.ctx);
        if (prc) {
            m.rc = prc;
            m.err = msspe_last_error(m.ctx);
            (void)hipStreamSynchronize(m.stream);
            return;
        }
        uint64_t c = 0;
        hipError_t e = hipMemcpy(&c, d_count, sizeof c, hipMemcpyDeviceToHost);
        const size_t have = (size_t)std::min<uint64_t>(c, capacity);
        if (e == hipSuccess && have) {
            hipLaunchKernelGGL(k_edges_home, dim3((unsigned)((have + 255) / 256)), dim3(256), 0, m.stream, d_edges,
                               (unsigned long long)have, m.rows, (uint32_t)n_base);
            got[(size_t)mi].resize(have);
            e = hipStreamSynchronize(m.stream);
            if (e == hipSuccess) e = hipMemcpy(got[(size_t)mi].data(), d_edges, sizeof(msspe_edge_dev) * have, hipMemcpyDeviceToHost);
        }
        counts[(size_t)mi] = c;
        M_HIP(m, e);
    });
    if (rc) return rc;
    uint64_t total = 0;
    std::vector<msspe_edge_dev> raw;
    for (size_t mi = 0; mi < N; ++mi) {
        total += counts[mi];
        raw.insert(raw.end(), got[mi].begin(), got[mi].end());
    }
    *count_out = total;
    // by (a, b), as the reference's nested loops emit the pairs (delta_g.rs:64-78)
    std::sort(raw.begin(), raw.end(), [](const msspe_edge_dev &x, const msspe_edge_dev &y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
    const size_t fill = (size_t)std::min<uint64_t>(raw.size(), capacity);
    for (size_t e = 0; e < fill; ++e) {
        edges[e].a = raw[e].a;
        edges[e].b = raw[e].b;
        edges[e].dg = msspe_round_fixed_f32((double)msspe_round_g_f32(raw[e].dg), 2);   // Edge::get_dg(), delta_g.rs:10-15
    }
    if (total > capacity)
        return gfail(g, MSSPE_ERR_CAPACITY, "edge list: " + std::to_string(total) + " conflict edges, capacity " + std::to_string(capacity));
    return MSSPE_OK;
}

int msspe_oligo_stats_group(msspe_group *g, const char *pool_ascii, int n, int k, const msspe_chem *chem, double *tm,
                            double *gc, double *self_any, double *self_end, double *hairpin)
{
    if (!g) return MSSPE_ERR_ARG;
    if (!pool_ascii || !chem || n < 0) return gfail(g, MSSPE_ERR_ARG, "null pool/chemistry");
    if (n == 0) return MSSPE_OK;
    const int N = (int)g->mem.size();
    // oligos are independent: contiguous slices, every member fills its part of the caller's arrays
    return on_every_member(g, [&](int mi) {
        Member &m = g->mem[(size_t)mi];
        const int base = n / N, extra = n % N;
        const int r0 = mi * base + std::min(mi, extra), r1 = r0 + base + (mi < extra ? 1 : 0);
        if (r1 <= r0) return;
        const int prc = msspe_oligo_stats(m.ctx, pool_ascii + (size_t)r0 * (size_t)k, r1 - r0, k, chem, tm ? tm + r0 : nullptr,
                                          gc ? gc + r0 : nullptr, self_any ? self_any + r0 : nullptr,
                                          self_end ? self_end + r0 : nullptr, hairpin ? hairpin + r0 : nullptr);
        if (prc) {
            m.rc = prc;
            m.err = msspe_last_error(m.ctx);
        }
    });
}

}  // extern "C"
