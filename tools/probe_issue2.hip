// Hardware probe (not product code), part 2: the engine's kernels issue one wave64 VALU instruction per 4.0 cycles per SIMD
// although probe_issue shows 2 cycles with >= 2 waves.  Candidates: many live VGPRs / operand-bank conflicts (the kernels
// use ~90 registers, the probes 20), three distinct VGPR sources per instruction, the real instruction mix (dct8s).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_issue2.hip -o tools/bin/probe_issue2
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr float S0 = 0.353553391f, T1 = 0.198912367f, T2 = 0.414213562f, T3 = 0.668178638f, R13 = 1.179580427f;
__device__ __forceinline__ void dct8s(float (&x)[8]) {
    const float a0 = x[0] + x[7], a1 = x[1] + x[6], a2 = x[2] + x[5], a3 = x[3] + x[4];
    const float b0 = x[0] - x[7], b1 = x[1] - x[6], b2 = x[2] - x[5], b3 = x[3] - x[4];
    const float c0 = a0 + a3, c1 = a1 + a2, c2 = a1 - a2, c3 = a0 - a3;
    x[0] = c0 + c1; x[4] = c0 - c1; x[2] = fmaf(c2, T2, c3); x[6] = fmaf(c3, T2, -c2);
    const float p4 = fmaf(b0, T3, b3), p7 = fmaf(b3, -T3, b0), p5 = fmaf(b1, T1, b2), p6 = fmaf(b2, -T1, b1);
    const float u4 = fmaf(p6, R13, p4), u6 = fmaf(p6, -R13, p4), u7 = fmaf(p5, R13, p7), u5 = fmaf(p5, -R13, p7);
    x[1] = u7 + u4; x[7] = u7 - u4; x[3] = u5; x[5] = u6;
}

// MODE 0: 2-D 8x8 scaled DCT on 64 registers, repeated (the kernels' own arithmetic: 416 VALU per pass, ~90 live VGPRs)
// MODE 1: fmac over NREG accumulators, two VGPR sources + accumulator   MODE 2: v_fma with three distinct VGPR sources
template <int MODE, int NREG>
__global__ __launch_bounds__(64) void k(float *out, int reps, float seed) {
    float s[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) s[i] = seed + i * 0.37f + threadIdx.x * 1e-3f;
    const float m = 0.999f, a = 1e-3f;
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {
            float (*R)[8] = reinterpret_cast<float (*)[8]>(s);
#pragma unroll
            for (int i = 0; i < 8; ++i) dct8s(R[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float col[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) col[i] = R[i][j];
                dct8s(col);
#pragma unroll
                for (int i = 0; i < 8; ++i) R[i][j] = col[i] * S0;
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int rep = 0; rep < 512 / NREG; ++rep)
#pragma unroll
                for (int i = 0; i < NREG; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(m), "v"(a));
        } else {
#pragma unroll
            for (int rep = 0; rep < 512 / NREG; ++rep)
#pragma unroll
                for (int i = 0; i < NREG; ++i)
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s[i]) : "v"(s[(i + 1) % NREG]), "v"(s[(i + NREG / 2) % NREG]), "v"(s[(i + 5) % NREG]));
        }
    }
    float x = 0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) x += s[i];
    if (x == 12345.678f) out[0] = x;
}

template <int MODE, int NREG>
void run(const char *name, int waves_per_simd, int per_rep) {
    float *d;
    hipMalloc(&d, 4);
    const int reps = 200, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NREG>), dim3(blocks), dim3(64), 0, 0, d, 2, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NREG>), dim3(blocks), dim3(64), 0, 0, d, reps, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %d waves/SIMD  %8.3f ms  %.2f ns per wave-instruction per SIMD\n", name, waves_per_simd, ms,
           ms * 1e6 / ((double)waves_per_simd * reps * per_rep));
    hipFree(d);
}

int main() {
    for (int w : {2, 4, 5}) {
        run<0, 64>("2-D scaled DCT on 64 registers (480 VALU per pass)", w, 480);
        run<1, 16>("fmac, 16 accumulators", w, 512);
        run<1, 64>("fmac, 64 accumulators", w, 512);
        run<1, 128>("fmac, 128 accumulators", w, 512);
        run<2, 64>("fma, three distinct VGPR sources, 64 registers", w, 512);
        run<2, 128>("fma, three distinct VGPR sources, 128 registers", w, 512);
    }
    return 0;
}
