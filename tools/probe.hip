// Hardware probes (not product code): v_cvt_pk_u8_f32 semantics and streaming-copy ceilings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

__global__ void cvt_probe(const float* in, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ src_, uint4* __restrict__ dst_, size_t n16) {
    const u32x4* src = (const u32x4*)src_; u32x4* dst = (u32x4*)dst_;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) { if (NT) __builtin_nontemporal_store(v[k], &dst[i + k * stride]); else dst[i + k * stride] = v[k]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// blocked: each workgroup copies a contiguous span
template <int UNROLL>
__global__ __launch_bounds__(256) void copy_blocked(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    uint4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) if (base + k * 256 < n16) v[k] = src[base + k * 256];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) if (base + k * 256 < n16) dst[base + k * 256] = v[k];
}

__global__ __launch_bounds__(256) void read_k(const uint4* __restrict__ src, unsigned* sink, size_t n16) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class F> float time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    // ---- cvt probe
    std::vector<float> h = {-5.f, -0.5f, -0.4f, 0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, NAN, 127.5f, 128.5f, 0.49999997f, 1.4999999f};
    float* din; unsigned* dout; CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, h.size() * 4));
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    cvt_probe<<<1, 64>>>(din, dout, (int)h.size());
    std::vector<unsigned> r(h.size()); CK(hipMemcpy(r.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) printf("cvt_pk_u8_f32(%g) -> byte %u  word %08x  | rne+sat would be %d\n", h[i], (r[i] >> 8) & 255, r[i],
                                                 std::isnan(h[i]) ? -1 : (int)nearbyintf(fminf(fmaxf(h[i], 0.f), 255.f)));
    // ---- copy ceilings
    size_t bytes = (size_t)1866240000; size_t n16 = bytes / 16;
    uint4 *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 2, bytes));
    unsigned* sink; CK(hipMalloc(&sink, 4));
    auto rep = [&](const char* name, float ms, double mult) { printf("%-34s %8.3f ms  %8.1f GB/s\n", name, ms, mult * bytes / ms / 1e6); };
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "copy stride u1 grid %d", g); rep(nm, time_ms([&] { copy_k<1, false><<<g, 256>>>(s, d, n16); }, 10), 2);
        snprintf(nm, 64, "copy stride u4 grid %d", g); rep(nm, time_ms([&] { copy_k<4, false><<<g, 256>>>(s, d, n16); }, 10), 2);
        snprintf(nm, 64, "copy stride u4 NT grid %d", g); rep(nm, time_ms([&] { copy_k<4, true><<<g, 256>>>(s, d, n16); }, 10), 2);
        snprintf(nm, 64, "read-only grid %d", g); rep(nm, time_ms([&] { read_k<<<g, 256>>>(s, sink, n16); }, 10), 1);
    }
    rep("copy blocked u4", time_ms([&] { copy_blocked<4><<<(unsigned)((n16 + 1023) / 1024), 256>>>(s, d, n16); }, 10), 2);
    rep("copy blocked u8", time_ms([&] { copy_blocked<8><<<(unsigned)((n16 + 2047) / 2048), 256>>>(s, d, n16); }, 10), 2);
    rep("hipMemcpyAsync D2D", time_ms([&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); }, 10), 2);
    return 0;
}
