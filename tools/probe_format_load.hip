// Hardware probe (not product code): can the texture addresser do the byte -> float conversions for free?
// buffer_load_format_xyzw through a buffer descriptor with DATA_FORMAT 8_8_8_8 / NUM_FORMAT USCALED returns four
// floats (0..255) per lane from four bytes, so the 192 v_cvt_f32_ubyte per 8x8 block (1.7 ns each, 22 % of analyze's
// issue time) would disappear.  Questions: are the values exact, and does the engine's access pattern (8 rows x 24 B per
// lane) still stream at HBM rate when every lane's row is fetched as six 4-byte typed loads (48 instead of 16 loads per block,
// four times the return data)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_format_load.hip -o tools/bin/probe_format_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i4 make_rsrc(const void *base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    i4 r;
    r.x = (int)(uint32_t)a;
    r.y = (int)((uint32_t)(a >> 32) & 0xffffu);             // stride 0
    r.z = (int)bytes;                                         // num_records (bytes when stride is 0)
    // word 3 (GFX9 layout): dst_sel x,y,z,w = 4,5,6,7; num_format [14:12] = 2 (USCALED); data_format [18:15] = 10 (8_8_8_8)
    r.w = (int)((4u) | (5u << 3) | (6u << 6) | (7u << 9) | (2u << 12) | (10u << 15));
    return r;
}
__device__ __forceinline__ f4 load_fmt(i4 rsrc, uint32_t byte_off) {
    f4 v;
    asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 offen" : "=v"(v) : "v"(byte_off), "s"(rsrc) : "memory");
    return v;
}

// MODE 0: typed loads, sum the floats;  MODE 1: plain 8-byte loads + 192 v_cvt_f32_ubyte, sum the floats
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ in, float *__restrict__ out, int W, int wb, int nblk, size_t frame_stride) {
    const int f = blockIdx.y;
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nblk) c = nblk - 1;
    const int bi = c / wb, bj = c - bi * wb;
    const uint32_t off = (uint32_t)(((size_t)bi * 8 * W + (size_t)bj * 8) * 3);
    const int pitch = W * 3;
    float acc = 0.f;
    if (MODE == 0) {
        const i4 rsrc = make_rsrc(in + (size_t)f * frame_stride, (uint32_t)frame_stride);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            f4 v[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) v[q] = load_fmt(rsrc, off + (uint32_t)r * pitch + 4 * q);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
#pragma unroll
            for (int q = 0; q < 6; ++q) acc += (v[q].x + v[q].y) + (v[q].z + v[q].w) * (float)(q + 1);
        }
    } else {
        const uint8_t *p = in + (size_t)f * frame_stride + off;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint2 *q2 = reinterpret_cast<const uint2 *>(p + (size_t)r * pitch);
            const uint2 a = q2[0], b = q2[1], d = q2[2];
            const uint32_t w[6] = {a.x, a.y, b.x, b.y, d.x, d.y};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float x = (float)(w[q] & 0xff), y = (float)((w[q] >> 8) & 0xff), z = (float)((w[q] >> 16) & 0xff), ww = (float)(w[q] >> 24);
                acc += (x + y) + (z + ww) * (float)(q + 1);
            }
        }
    }
    if (MODE >= 0) out[(size_t)f * nblk + c] = acc;
}

int main() {
    const int W = 1920, H = 1080, wb = W / 8, nblk = (H / 8) * wb, nf = 300;
    const size_t fs = (size_t)H * W * 3;
    uint8_t *in; float *o0, *o1;
    hipMalloc(&in, fs * nf); hipMalloc(&o0, sizeof(float) * nblk * nf); hipMalloc(&o1, sizeof(float) * nblk * nf);
    std::vector<uint8_t> h(fs);
    for (size_t i = 0; i < fs; ++i) h[i] = (uint8_t)((i * 2654435761u) >> 13);
    for (int f = 0; f < nf; ++f) hipMemcpy(in + f * fs, h.data(), fs, hipMemcpyHostToDevice);
    const dim3 grid((nblk + 255) / 256, nf);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 2; ++i) { if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, in, o0, W, wb, nblk, fs); else hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, in, o1, W, wb, nblk, fs); }
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) { if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, in, o0, W, wb, nblk, fs); else hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, in, o1, W, wb, nblk, fs); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.3f ms per 300 frames  %7.1f GB/s of pixel bytes\n", mode == 0 ? "typed loads (6 x buffer_load_format_xyzw per row)" : "plain loads + 192 v_cvt_f32_ubyte per block", ms / 10, 10.0 * nf * fs / (ms * 1e-3) / 1e9);
    }
    std::vector<float> a(nblk), b(nblk);
    hipMemcpy(a.data(), o0, sizeof(float) * nblk, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o1, sizeof(float) * nblk, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < nblk; ++i) bad += a[i] != b[i];
    printf("typed-load sums differ from converted-byte sums in %d of %d blocks (first: %g vs %g)\n", bad, nblk, a[0], b[0]);
    return 0;
}
