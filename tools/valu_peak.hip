// valu_peak.hip -- how many wave64 vector-ALU instructions one gfx950 SIMD issues per cycle.
//
// The all-pairs DP kernel (csrc/thal_pairs_int.hip) is bound by vector-ALU issue, so its roofline
// needs the issue rate of the instructions it is made of, measured, not taken from a data sheet.
// Every op below runs as eight independent chains per wave in a 64-instruction loop body, at 1, 2,
// 3 and 4 waves per SIMD on every CU (one block per CU, forced by a 96 KB LDS declaration).
// Output: one JSON line per (op, waves/SIMD): cycles per wave-instruction per SIMD, from the wall
// time of the launch at the in-kernel clock (s_memtime / s_memrealtime).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_peak tools/valu_peak.hip && tools/valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int kUnroll = 64;   // instructions per loop body (8 chains x 8)

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

// A kernel per op.  BODY(n) is one asm statement on chain n (r##n: 32-bit, d##n: 64-bit);
// x, y: loop-invariant VGPRs; m: a loop-invariant 64-bit SGPR mask; dx, dy: doubles.
#define DEF_KERNEL(NAME, BODY, PER8)                                                                            \
    __global__ void __launch_bounds__(1024) NAME(int iters, unsigned *sink, unsigned long long *cycles, int x_in, \
                                                 int y_in, unsigned long long m_in)                              \
    {                                                                                                            \
        __shared__ char pad[96 * 1024];                                                                          \
        if (x_in == 0x7fffffff) pad[threadIdx.x] = 1;                                                            \
        int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,        \
            r7 = r0 + 7;                                                                                         \
        double d0 = r0, d1 = r1, d2 = r2, d3 = r3, d4 = r4, d5 = r5, d6 = r6, d7 = r7;                            \
        const int x = x_in, y = y_in;                                                                            \
        const unsigned long long m = __builtin_amdgcn_readfirstlane((int)m_in) |                                 \
                                     ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m_in >> 32)) << 32); \
        const double dx = 1.0 + 1e-9 * x_in, dy = 1e-9 * y_in;                                                   \
        (void)x; (void)y; (void)m; (void)dx; (void)dy;                                                           \
        __syncthreads();                                                                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
        const unsigned long long q0 = __builtin_amdgcn_s_memrealtime();                                          \
        for (int it = 0; it < iters; ++it) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < kUnroll / 8; ++u) { PER8(BODY) }                               \
        }                                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
        const unsigned long long q1 = __builtin_amdgcn_s_memrealtime();                                          \
        unsigned acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                                    \
        acc ^= (unsigned)(long long)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);                                     \
        if (acc == 0x12345678u) sink[0] = acc;                                                                   \
        if ((threadIdx.x & 63) == 0) {                                                                           \
            cycles[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t1 - t0;                                        \
            cycles[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = q1 - q0;                                    \
        }                                                                                                        \
    }

#define A32(TXT) asm volatile(TXT : "+v"(RN) : "v"(x), "v"(y), "s"(m) : "vcc", "s40", "s41");
#define A64(TXT) asm volatile(TXT : "+v"(DN) : "v"(dx), "v"(dy), "s"(m) : "vcc", "s40", "s41");

// ---- 32-bit integer ops of the DP scan
#define B(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_add_u32, B, REP8)
#undef B
#define B(n) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_sub_u32, B, REP8)
#undef B
#define B(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_and_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_or_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_xor_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r##n));
DEF_KERNEL(k_lshlrev_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r##n));
DEF_KERNEL(k_lshrrev_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_mov_b32 %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_mov_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_min_i32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_min_i32, B, REP8)
#undef B
#define B(n) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_max_i32, B, REP8)
#undef B
#define B(n) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_min_u32, B, REP8)
#undef B
#define B(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r##n) : "v"(x), "v"(y));
DEF_KERNEL(k_add3_u32, B, REP8)
#undef B
#define B(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(x), "v"(y));
DEF_KERNEL(k_and_or_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r##n) : "v"(y));
DEF_KERNEL(k_lshl_or_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r##n) : "v"(y));
DEF_KERNEL(k_lshl_add_u32, B, REP8)
#undef B
#define B(n) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(r##n) : "v"(y));
DEF_KERNEL(k_add_lshl_u32, B, REP8)
#undef B
#define B(n) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(r##n));
DEF_KERNEL(k_bfe_u32, B, REP8)
#undef B
#define B(n) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r##n) : "v"(x), "v"(y));
DEF_KERNEL(k_bfi_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(x), "v"(y));
DEF_KERNEL(k_perm_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r##n) : "v"(x), "v"(y));
DEF_KERNEL(k_mad_u32_u24, B, REP8)
#undef B
#define B(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_mul_u32_u24, B, REP8)
#undef B
#define B(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_mul_lo_u32, B, REP8)
#undef B
#define B(n) \
    asm volatile("v_sub_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_sub_u32_sdwa, B, REP8)
#undef B
#define B(n) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r##n));
DEF_KERNEL(k_ffbl_b32, B, REP8)
#undef B
// ---- compares and selects
#define B(n) asm volatile("v_cmp_lt_i32_e64 s[40:41], %0, %1" : : "v"(r##n), "v"(x) : "s40", "s41");
DEF_KERNEL(k_cmp_lt_i32_e64, B, REP8)
#undef B
#define B(n) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1" : : "v"(r##n), "v"(x) : "vcc");
DEF_KERNEL(k_cmp_lt_i32_e32, B, REP8)
#undef B
#define B(n) \
    asm volatile("v_cmp_eq_u32_sdwa s[40:41], %0, %1 src0_sel:BYTE_0 src1_sel:DWORD" : : "v"(r##n), "v"(x) : "s40", "s41");
DEF_KERNEL(k_cmp_eq_u32_sdwa, B, REP8)
#undef B
#define B(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##n) : "v"(x), "s"(m));
DEF_KERNEL(k_cndmask_b32_e64_sgpr, B, REP8)
#undef B
#define B(n) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_cndmask_b32_e32_vcc, B, REP8)
#undef B
// compare into an SGPR pair, then the select that uses it (the scan's min / argmin idiom)
#define B(n)                                                                                                 \
    asm volatile("v_cmp_lt_i32_e64 s[40:41], %1, %0\n\tv_cndmask_b32_e64 %0, %0, %2, s[40:41]" : "+v"(r##n) \
                 : "v"(x), "v"(y) : "s40", "s41");
#define REP4(I) I(0) I(1) I(2) I(3)
DEF_KERNEL(k_cmp_then_cndmask_e64, B, REP4)
#undef B
// ---- lane crossing, LDS
#define B(n) asm volatile("v_readlane_b32 s40, %0, 3" : : "v"(r##n) : "s40");
DEF_KERNEL(k_readlane_b32, B, REP8)
#undef B
#define B(n) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r##n));
DEF_KERNEL(k_mov_b32_dpp, B, REP8)
#undef B
// ---- floating point of the replay / maxTM
#define B(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(x), "v"(y));
DEF_KERNEL(k_fma_f32, B, REP8)
#undef B
#define B(n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r##n) : "v"(x));
DEF_KERNEL(k_add_f32, B, REP8)
#undef B
#define B(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d##n) : "v"(dx), "v"(dy));
DEF_KERNEL(k_pk_fma_f32, B, REP8)
#undef B
#define B(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##n) : "v"(dx), "v"(dy));
DEF_KERNEL(k_fma_f64, B, REP8)
#undef B
#define B(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##n) : "v"(dy));
DEF_KERNEL(k_add_f64, B, REP8)
#undef B
#define B(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##n) : "v"(dx));
DEF_KERNEL(k_mul_f64, B, REP8)
#undef B
#define B(n) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(d##n) : "v"(x));
DEF_KERNEL(k_cvt_f64_i32, B, REP8)
#undef B

typedef void (*kern_t)(int, unsigned *, unsigned long long *, int, int, unsigned long long);

struct OpDesc {
    const char *name;
    kern_t fn;
    int valu_per_body;   // vector-ALU instructions in one 8-statement group x 8 groups
};

static void run_op(const OpDesc &op, int n_cu, unsigned *d_sink, unsigned long long *d_cycles)
{
    const int iters = 8000;
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 256 * wps;
        std::vector<unsigned long long> h(n_cu * 32);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(op.fn, dim3(n_cu), dim3(threads), 0, 0, iters / 8, d_sink, d_cycles, 3, 5, 0x5555aaaa5555aaaaull);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(op.fn, dim3(n_cu), dim3(threads), 0, 0, iters, d_sink, d_cycles, 3, 5, 0x5555aaaa5555aaaaull);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), d_cycles, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
        double sum = 0, rsum = 0;
        for (int b = 0; b < n_cu; ++b)
            for (int w = 0; w < 4 * wps; ++w) {
                sum += (double)h[(b * 16 + w) * 2];
                rsum += (double)h[(b * 16 + w) * 2 + 1];
            }
        const double clock_mhz = 100.0 * sum / rsum;                        // shader cycles per 100 MHz tick
        const double per_simd = (double)iters * op.valu_per_body * wps;     // VALU wave-instructions on one SIMD
        printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"valu_per_simd\": %.0f, \"wall_ms\": %.4f, "
               "\"clock_mhz_in_kernel\": %.0f, \"cycles_per_valu_per_simd\": %.3f}\n",
               op.name, wps, per_simd, ms, clock_mhz, (ms * 1e3 * clock_mhz) / per_simd);
        fflush(stdout);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("{\"arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"loop_body\": %d}\n", prop.gcnArchName, n_cu, prop.clockRate,
           kUnroll);
    unsigned *d_sink;
    unsigned long long *d_cycles;
    CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * n_cu * 32));
#define OP(k) {#k, k, kUnroll}
    const OpDesc ops[] = {
        OP(k_add_u32), OP(k_sub_u32), OP(k_and_b32), OP(k_or_b32), OP(k_xor_b32), OP(k_lshlrev_b32), OP(k_lshrrev_b32),
        OP(k_mov_b32), OP(k_min_i32), OP(k_max_i32), OP(k_min_u32), OP(k_add3_u32), OP(k_and_or_b32), OP(k_lshl_or_b32),
        OP(k_lshl_add_u32), OP(k_add_lshl_u32), OP(k_bfe_u32), OP(k_bfi_b32), OP(k_perm_b32), OP(k_mad_u32_u24),
        OP(k_mul_u32_u24), OP(k_mul_lo_u32), OP(k_sub_u32_sdwa), OP(k_ffbl_b32), OP(k_cmp_lt_i32_e64), OP(k_cmp_lt_i32_e32),
        OP(k_cmp_eq_u32_sdwa), OP(k_cndmask_b32_e64_sgpr), OP(k_cndmask_b32_e32_vcc),
        {"k_cmp_then_cndmask_e64", k_cmp_then_cndmask_e64, kUnroll},
        OP(k_readlane_b32), OP(k_mov_b32_dpp), OP(k_fma_f32), OP(k_add_f32), OP(k_pk_fma_f32), OP(k_fma_f64), OP(k_add_f64),
        OP(k_mul_f64), OP(k_cvt_f64_i32)};
    for (const OpDesc &op : ops) run_op(op, n_cu, d_sink, d_cycles);
    return 0;
}
