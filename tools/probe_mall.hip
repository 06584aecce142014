// Probe: read bandwidth as a function of working-set size (Infinity Cache residency), and
// read-after-read / read-after-write reuse across kernels.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void read_k(const uint4* __restrict__ src, unsigned* sink, size_t n16) {
    size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x; unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k * 256 < n16) { uint4 v = src[base + k * 256]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x; uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k * 256 < n16) v[k] = src[base + k * 256];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k * 256 < n16) dst[base + k * 256] = v[k];
}
int main() {
    size_t big = (size_t)2 << 30; uint4 *a, *b; unsigned* sink;
    CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMalloc(&sink, 4)); CK(hipMemset(a, 1, big)); CK(hipMemset(b, 2, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
        size_t bytes = mb << 20, n16 = bytes / 16; unsigned grid = (unsigned)((n16 + 1023) / 1024);
        int reps = (int)(8192 / mb); if (reps < 4) reps = 4;
        for (int i = 0; i < 3; ++i) read_k<<<grid, 256>>>(a, sink, n16);
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) read_k<<<grid, 256>>>(a, sink, n16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        float rd = (float)bytes * reps / ms / 1e6;
        for (int i = 0; i < 3; ++i) copy_k<<<grid, 256>>>(a, b, n16);
        CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) copy_k<<<grid, 256>>>(a, b, n16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("working set %5zu MiB: repeated read %8.1f GB/s (%.1f us/launch) | repeated copy a->b %8.1f GB/s (r+w)\n", mb, rd, 1e3 * bytes / rd / 1e6 / 1e3 * 1e0, 2.0f * bytes * reps / ms / 1e6);
    }
    return 0;
}
