// Hardware probe (not product code): what limits VALU issue in long straight-line kernels?  The engine's kernels run at
// 4.0 cycles per wave64 VALU instruction per SIMD (PMC: SQ_ACTIVE_INST_VALU ~ 95 % of the cycles), the tight-loop probe
// (tools/probe_pk.hip) at 2.5-3.  Variables: waves per SIMD, loop vs fully unrolled code (instruction fetch), dependent vs
// independent instructions, 4-byte (VOP2) vs 8-byte (VOP3 / literal) encodings.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_issue.hip -o tools/bin/probe_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
#define FMA_IND(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(m), "v"(a));
#define FMA16_IND FMA_IND(0) FMA_IND(1) FMA_IND(2) FMA_IND(3) FMA_IND(4) FMA_IND(5) FMA_IND(6) FMA_IND(7) FMA_IND(8) FMA_IND(9) FMA_IND(10) FMA_IND(11) FMA_IND(12) FMA_IND(13) FMA_IND(14) FMA_IND(15)
#define FMA_DEP(i) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(s[i & 1]) : "v"(m));
#define FMA16_DEP FMA_DEP(0) FMA_DEP(1) FMA_DEP(2) FMA_DEP(3) FMA_DEP(4) FMA_DEP(5) FMA_DEP(6) FMA_DEP(7) FMA_DEP(8) FMA_DEP(9) FMA_DEP(10) FMA_DEP(11) FMA_DEP(12) FMA_DEP(13) FMA_DEP(14) FMA_DEP(15)
#define FMA_LIT(i) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(s[i]) : "v"(a));
#define FMA16_LIT FMA_LIT(0) FMA_LIT(1) FMA_LIT(2) FMA_LIT(3) FMA_LIT(4) FMA_LIT(5) FMA_LIT(6) FMA_LIT(7) FMA_LIT(8) FMA_LIT(9) FMA_LIT(10) FMA_LIT(11) FMA_LIT(12) FMA_LIT(13) FMA_LIT(14) FMA_LIT(15)

// MODE 0: loop of 16 independent VOP2 fmac;  1: the same 2048 instructions fully unrolled;  2: unrolled, 2 dependent chains;
// 3: unrolled, 8-byte encodings (literal);  4: loop, 8-byte encodings
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int reps, float seed) {
    float s[16];
    for (int i = 0; i < 16; ++i) s[i] = seed + i + threadIdx.x;
    const float m = 0.999f, a = 1e-3f;
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {
            for (int it = 0; it < 128; ++it) { FMA16_IND }
        } else if (MODE == 1) {
            REP16(REP16(FMA16_IND) ) 
        } else if (MODE == 2) {
            REP16(REP16(FMA16_DEP) )
        } else if (MODE == 3) {
            REP16(REP16(FMA16_LIT) )
        } else {
            for (int it = 0; it < 128; ++it) { FMA16_LIT }
        }
    }
    float x = 0;
    for (int i = 0; i < 16; ++i) x += s[i];
    if (x == 12345.678f) out[0] = x;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int per_rep) {
    float *d;
    hipMalloc(&d, 4);
    const int reps = 40, blocks = 256 * 4 * waves_per_simd;      // one 64-thread workgroup = one wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 2, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, reps, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)waves_per_simd * reps * per_rep;
    printf("%-46s %d waves/SIMD  %8.3f ms  %.2f ns per wave-instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / per_simd);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 5, 8}) {
        run<0>("loop, independent, 4-byte fmac", w, 128 * 16);
        run<1>("unrolled 4096, independent, 4-byte fmac", w, 4096);
        run<2>("unrolled 4096, two dependent chains", w, 4096);
        run<3>("unrolled 4096, independent, 8-byte fmamk", w, 4096);
        run<4>("loop, independent, 8-byte fmamk", w, 128 * 16);
    }
    return 0;
}
