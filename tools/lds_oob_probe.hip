// Development probe behind DESIGN.md 10 (2): what ds_read_b32 returns at and beyond a block's LDS allocation on gfx950
// (result: the allocation is rounded up to 1,280 B; every address from there on reads 0, up to at least 400 KB).
// hipcc --offload-arch=gfx950 -O2 -o lds_oob_probe tools/lds_oob_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
// What does ds_read_b32 return for byte addresses at and beyond the block's LDS allocation?
template <int N>
__global__ void probe(unsigned *out, unsigned lo, unsigned step, int n)
{
    __shared__ int a[N];
    for (int i = threadIdx.x; i < N; i += blockDim.x) a[i] = 0x07070707;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        unsigned addr = lo + (unsigned)k * step, v;
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        out[k] = v;
    }
}
template <int N> void run(const char *name)
{
    const int n = 4096; unsigned *d, h[n];
    hipMalloc(&d, n * 4);
    const unsigned lo = (unsigned)(N * 4) - 1024u, step = 64;   // from 1 KB inside the array to 255 KB beyond
    probe<N><<<1, 256>>>(d, lo, step, n);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int first_zero = -1, nonzero_after = 0; unsigned last_nz = 0;
    for (int k = 0; k < n; ++k) {
        const unsigned addr = lo + k * step;
        if (h[k] == 0 && first_zero < 0) first_zero = (int)addr;
        if (first_zero >= 0 && h[k] != 0) { ++nonzero_after; last_nz = addr; }
    }
    printf("%s: alloc %d B, first zero at byte %d, non-zero reads after that: %d (last at %u), scanned to %u\n", name, N * 4,
           first_zero, nonzero_after, last_nz, lo + (n - 1) * step);
    hipFree(d);
}
int main()
{
    run<1000>("4000 B");
    run<8192>("32 KB");
    run<16384 + 100>("64 KB + 400");
    run<35000>("140000 B");
    run<40960>("160 KB");
    return 0;
}
