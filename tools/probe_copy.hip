// Hardware probe (not product code): which shape of a register-only streaming copy / read reaches the device's
// HBM rate.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe_copy.hip -o tools/bin/probe_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int ITEMS, bool NT>
__global__ __launch_bounds__(256) void copy_span(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * ITEMS) + threadIdx.x;
    u4 v[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) v[k] = NT ? __builtin_nontemporal_load(s + base + k * 256) : s[base + k * 256];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) { if (NT) __builtin_nontemporal_store(v[k], d + base + k * 256); else d[base + k * 256] = v[k]; }
}
template <int ITEMS>
__global__ __launch_bounds__(256) void copy_stride(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n) {
    const size_t step = (size_t)gridDim.x * 256 * ITEMS;
    for (size_t base = (size_t)blockIdx.x * (256 * ITEMS) + threadIdx.x; base + (ITEMS - 1) * 256 < n; base += step) {
        u4 v[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) v[k] = s[base + k * 256];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) d[base + k * 256] = v[k];
    }
}
__global__ void copy_simple(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <int ITEMS>
__global__ __launch_bounds__(256) void read_span(const u4 *__restrict__ s, unsigned *sink, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * ITEMS) + threadIdx.x;
    u4 v[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) v[k] = s[base + k * 256];
    unsigned x = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) x ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    if (x == 0x9E3779B9u && (blockIdx.x ^ threadIdx.x) == 0x5bd1e995u) sink[0] = x;
}

// The engine's own access pattern without its arithmetic: one thread per 8x8 pixel block of interleaved u8 RGB,
// 8 rows x 24 bytes per lane (3 x 8-byte accesses per row), 64 adjacent blocks per wave.
typedef unsigned u2v __attribute__((ext_vector_type(2)));
template <int THREADS, bool NTL, bool NTS, int LDS_PAD>
__global__ __launch_bounds__(THREADS) void pattern_var(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int wb, int nblk,
                                                       size_t frame_stride) {
    __shared__ volatile char pad[LDS_PAD > 0 ? LDS_PAD : 1];
    if (nblk < 0) pad[threadIdx.x] = 0;
    const int f = blockIdx.y;
    int c = blockIdx.x * THREADS + threadIdx.x;
    if (c >= nblk) c = nblk - 1;
    const int bi = c / wb, bj = c - bi * wb;
    const size_t off = (size_t)f * frame_stride + ((size_t)bi * 8 * W + (size_t)bj * 8) * 3;
    const int pitch = W * 3;
    u2v v[8][3];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const u2v *q = reinterpret_cast<const u2v *>(in + off + (size_t)r * pitch);
#pragma unroll
        for (int k = 0; k < 3; ++k) v[r][k] = NTL ? __builtin_nontemporal_load(q + k) : q[k];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        u2v *q = reinterpret_cast<u2v *>(out + off + (size_t)r * pitch);
#pragma unroll
        for (int k = 0; k < 3; ++k) { if (NTS) __builtin_nontemporal_store(v[r][k], q + k); else q[k] = v[r][k]; }
    }
}

template <bool WRITE, int ROWS_IN_FLIGHT>
__global__ __launch_bounds__(256) void pattern_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int wb, int nblk,
                                                      size_t frame_stride, unsigned *sink) {
    const int f = blockIdx.y;
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nblk) c = nblk - 1;
    const int bi = c / wb, bj = c - bi * wb;
    const size_t off = (size_t)f * frame_stride + ((size_t)bi * 8 * W + (size_t)bj * 8) * 3;
    const int pitch = W * 3;
    unsigned x = 0;
#pragma unroll
    for (int r0 = 0; r0 < 8; r0 += ROWS_IN_FLIGHT) {
        uint2 v[ROWS_IN_FLIGHT][3];
#pragma unroll
        for (int r = 0; r < ROWS_IN_FLIGHT; ++r) {
            const uint2 *q = reinterpret_cast<const uint2 *>(in + off + (size_t)(r0 + r) * pitch);
            v[r][0] = q[0]; v[r][1] = q[1]; v[r][2] = q[2];
        }
#pragma unroll
        for (int r = 0; r < ROWS_IN_FLIGHT; ++r) {
            if (WRITE) {
                uint2 *q = reinterpret_cast<uint2 *>(out + off + (size_t)(r0 + r) * pitch);
                q[0] = v[r][0]; q[1] = v[r][1]; q[2] = v[r][2];
            } else {
                x ^= v[r][0].x ^ v[r][0].y ^ v[r][1].x ^ v[r][1].y ^ v[r][2].x ^ v[r][2].y;
            }
        }
    }
    if (!WRITE && x == 0x9E3779B9u && (blockIdx.x ^ threadIdx.x) == 0x5bd1e995u) sink[0] = x;
}

// Same reads as the engine pattern, but every store instruction writes 16 bytes per lane to consecutive addresses
// (what staging the output rows through LDS would give): is the 24-byte-stride store shape what costs the mark kernel?
__global__ __launch_bounds__(256) void pattern_coalesced_write(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int wb,
                                                               int nblk, size_t frame_stride) {
    const int f = blockIdx.y;
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nblk) c = nblk - 1;
    const int bi = c / wb, bj = c - bi * wb;
    const size_t off = (size_t)f * frame_stride + ((size_t)bi * 8 * W + (size_t)bj * 8) * 3;
    const int pitch = W * 3;
    const int lane = threadIdx.x & 63;
    const int c0 = c - lane;                                  // first block of the wave (probe: ignores row straddling)
    const int bi0 = c0 / wb, bj0 = c0 - bi0 * wb;
    const size_t off0 = (size_t)f * frame_stride + ((size_t)bi0 * 8 * W + (size_t)bj0 * 8) * 3;
#pragma unroll
    for (int r0 = 0; r0 < 8; r0 += 2) {
        uint2 v[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint2 *q = reinterpret_cast<const uint2 *>(in + off + (size_t)(r0 + r) * pitch);
            v[r][0] = q[0]; v[r][1] = q[1]; v[r][2] = q[2];
        }
        // 2 rows x 1536 B = 3 full-wave 16-byte stores
        const uint4 a = make_uint4(v[0][0].x, v[0][0].y, v[0][1].x, v[0][1].y);
        const uint4 b = make_uint4(v[0][2].x, v[0][2].y, v[1][0].x, v[1][0].y);
        const uint4 cc = make_uint4(v[1][1].x, v[1][1].y, v[1][2].x, v[1][2].y);
        uint8_t *row0 = out + off0 + (size_t)r0 * pitch, *row1 = row0 + pitch;
        *reinterpret_cast<uint4 *>(row0 + lane * 16) = a;
        *reinterpret_cast<uint4 *>(lane < 32 ? row0 + 1024 + lane * 16 : row1 + (lane - 32) * 16) = b;
        *reinterpret_cast<uint4 *>(row1 + 512 + lane * 16) = cc;
    }
}

// The engine pattern with the workgroup -> tile mapping made XCD-aware.  Workgroups are dispatched round-robin over the 8 XCDs
// in linear order (x fastest, then y), so by default XCD k gets every 8th 48 KiB tile.  MODE 1: XCD k walks ONE contiguous
// eighth of the batch (consecutive workgroups of an XCD are neighbours in memory).  MODE 2: XCD k takes every 8th FRAME
// (frame-contiguous per XCD).  Is the 24-byte-per-lane pattern's 15 % below the plain copy a DRAM-locality effect?
template <int MODE>
__global__ __launch_bounds__(256) void pattern_xcd(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int W, int wb, int nblk,
                                                   size_t frame_stride, int nf) {
    const unsigned gx = gridDim.x, G = gx * (unsigned)nf;
    const unsigned L = blockIdx.y * gx + blockIdx.x;
    unsigned t;
    if (MODE == 1) {
        const unsigned per = (G + 7) / 8;
        t = (L % 8) * per + L / 8;
    } else {
        const unsigned xcd = L % 8, i = L / 8;                 // i-th workgroup of this XCD
        const unsigned fpx = ((unsigned)nf + 7) / 8;           // frames per XCD
        const unsigned fl = i / gx, x = i % gx;
        t = (fl * 8 + xcd) * gx + x;
        if (fl >= fpx) return;
    }
    if (t >= G) return;
    const int f = t / gx;
    int c = (t % gx) * 256 + threadIdx.x;
    if (c >= nblk) c = nblk - 1;
    const int bi = c / wb, bj = c - bi * wb;
    const size_t off = (size_t)f * frame_stride + ((size_t)bi * 8 * W + (size_t)bj * 8) * 3;
    const int pitch = W * 3;
    uint2 v[8][3];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint2 *q = reinterpret_cast<const uint2 *>(in + off + (size_t)r * pitch);
        v[r][0] = q[0]; v[r][1] = q[1]; v[r][2] = q[2];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        uint2 *q = reinterpret_cast<uint2 *>(out + off + (size_t)r * pitch);
        q[0] = v[r][0]; q[1] = v[r][1]; q[2] = v[r][2];
    }
}

template <typename F>
void timeit(const char *name, double bytes, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms/launch  %7.1f GB/s\n", name, ms / 10, 10 * bytes / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = (size_t)300 * 1080 * 1920 * 3;      // the benchmark's frame batch: 1.87 GB
    const size_t n = bytes / 16;                              // multiple of 256*8: 116 640 000 = 2048 * 56953.1 -> trim
    const size_t n8 = n / 2048 * 2048;
    u4 *s, *d; unsigned *sink;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMalloc(&sink, 16);
    hipMemset(s, 0x5a, bytes); hipMemset(d, 0, bytes);
    const double moved = 2.0 * n8 * 16;
    timeit("copy span 4 x16B/lane", moved, [&] { hipLaunchKernelGGL((copy_span<4, false>), dim3(n8 / 1024), dim3(256), 0, 0, s, d, n8); });
    timeit("copy span 8 x16B/lane", moved, [&] { hipLaunchKernelGGL((copy_span<8, false>), dim3(n8 / 2048), dim3(256), 0, 0, s, d, n8); });
    timeit("copy span 2 x16B/lane", moved, [&] { hipLaunchKernelGGL((copy_span<2, false>), dim3(n8 / 512), dim3(256), 0, 0, s, d, n8); });
    timeit("copy span 1 x16B/lane", moved, [&] { hipLaunchKernelGGL((copy_span<1, false>), dim3(n8 / 256), dim3(256), 0, 0, s, d, n8); });
    timeit("copy span 4, nontemporal", moved, [&] { hipLaunchKernelGGL((copy_span<4, true>), dim3(n8 / 1024), dim3(256), 0, 0, s, d, n8); });
    timeit("copy span 8, nontemporal", moved, [&] { hipLaunchKernelGGL((copy_span<8, true>), dim3(n8 / 2048), dim3(256), 0, 0, s, d, n8); });
    timeit("copy simple (1 per thread, 256 thr)", moved, [&] { hipLaunchKernelGGL(copy_simple, dim3(n8 / 256), dim3(256), 0, 0, s, d, n8); });
    timeit("copy simple (1 per thread, 1024 thr)", moved, [&] { hipLaunchKernelGGL(copy_simple, dim3(n8 / 1024), dim3(1024), 0, 0, s, d, n8); });
    timeit("copy grid-stride 2048 WG x4", moved, [&] { hipLaunchKernelGGL((copy_stride<4>), dim3(2048), dim3(256), 0, 0, s, d, n8); });
    timeit("copy grid-stride 4096 WG x4", moved, [&] { hipLaunchKernelGGL((copy_stride<4>), dim3(4096), dim3(256), 0, 0, s, d, n8); });
    timeit("copy grid-stride 1024 WG x8", moved, [&] { hipLaunchKernelGGL((copy_stride<8>), dim3(1024), dim3(256), 0, 0, s, d, n8); });
    timeit("hipMemcpyDtoD", moved, [&] { hipMemcpyAsync(d, s, n8 * 16, hipMemcpyDeviceToDevice, 0); });
    const double rd = 1.0 * n8 * 16;
    timeit("read span 4", rd, [&] { hipLaunchKernelGGL((read_span<4>), dim3(n8 / 1024), dim3(256), 0, 0, s, sink, n8); });
    timeit("read span 8", rd, [&] { hipLaunchKernelGGL((read_span<8>), dim3(n8 / 2048), dim3(256), 0, 0, s, sink, n8); });
    timeit("read span 2", rd, [&] { hipLaunchKernelGGL((read_span<2>), dim3(n8 / 512), dim3(256), 0, 0, s, sink, n8); });
    timeit("read span 1", rd, [&] { hipLaunchKernelGGL((read_span<1>), dim3(n8 / 256), dim3(256), 0, 0, s, sink, n8); });
    {
        const int W = 1920, H = 1080, wb = W / 8, nblk = (H / 8) * wb, nf = 300;
        const size_t fs = (size_t)H * W * 3;
        const dim3 grid((nblk + 255) / 256, nf);
        const uint8_t *in = reinterpret_cast<const uint8_t *>(s);
        uint8_t *out = reinterpret_cast<uint8_t *>(d);
        timeit("engine pattern copy, 8 rows in flight", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_kernel<true, 8>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs, sink); });
        timeit("engine pattern copy, 4 rows in flight", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_kernel<true, 4>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs, sink); });
        timeit("engine pattern copy, 2 rows in flight", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_kernel<true, 2>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs, sink); });
        timeit("engine pattern copy, 1 row in flight", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_kernel<true, 1>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs, sink); });
        timeit("engine pattern copy, 64-thread WGs", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_var<64, false, false, 0>), dim3((nblk + 63) / 64, nf), dim3(64), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern copy, 128-thread WGs", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_var<128, false, false, 0>), dim3((nblk + 127) / 128, nf), dim3(128), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern copy, nt stores", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_var<256, false, true, 0>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern copy, nt loads", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_var<256, true, false, 0>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern copy, nt loads+stores", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_var<256, true, true, 0>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern copy, 4 WG/CU (LDS cap)", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_var<256, false, false, 40000>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern copy, 2 WG/CU (LDS cap)", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_var<256, false, false, 80000>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern copy, 1 WG/CU (LDS cap)", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_var<256, false, false, 160000>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern copy, XCD-contiguous eighths", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_xcd<1>), dim3(grid.x, grid.y + 1), dim3(256), 0, 0, in, out, W, wb, nblk, fs, nf); });
        timeit("engine pattern copy, frames dealt to XCDs", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_xcd<2>), dim3(grid.x, grid.y + 8), dim3(256), 0, 0, in, out, W, wb, nblk, fs, nf); });
        timeit("engine pattern copy, 8 rows in flight (again)", 2.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_kernel<true, 8>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs, sink); });
        timeit("engine reads + coalesced 16-B stores", 2.0 * nf * fs, [&] { hipLaunchKernelGGL(pattern_coalesced_write, grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs); });
        timeit("engine pattern read, 8 rows in flight", 1.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_kernel<false, 8>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs, sink); });
        timeit("engine pattern read, 4 rows in flight", 1.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_kernel<false, 4>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs, sink); });
        timeit("engine pattern read, 2 rows in flight", 1.0 * nf * fs, [&] { hipLaunchKernelGGL((pattern_kernel<false, 2>), grid, dim3(256), 0, 0, in, out, W, wb, nblk, fs, sink); });
    }
    // smaller working sets (Infinity-Cache resident): 128 MiB
    const size_t small = (size_t)(128 << 20) / 16;
    timeit("copy span 4, 128 MiB + 128 MiB", 2.0 * small * 16, [&] { hipLaunchKernelGGL((copy_span<4, false>), dim3(small / 1024), dim3(256), 0, 0, s, d, small); });
    timeit("read span 8, 128 MiB", 1.0 * small * 16, [&] { hipLaunchKernelGGL((read_span<8>), dim3(small / 2048), dim3(256), 0, 0, s, sink, small); });
    return 0;
}
