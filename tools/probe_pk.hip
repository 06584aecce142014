// Hardware probe (not product code): issue rate of packed-f32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
// against their scalar forms on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe_pk.hip -o probe_pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float s[16];
    f2 p[8];
    for (int i = 0; i < 16; ++i) s[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 8; ++i) p[i] = f2{s[2 * i], s[2 * i + 1]};
    const float m = 0.999f, a = 1e-3f;
    const f2 m2 = {m, m}, a2 = {a, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (MODE == 0) {            // 16 scalar fma
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m), "v"(a));
            } else if (MODE == 1) {     // 8 packed fma = the same 16 flops-pairs
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(a2));
            } else if (MODE == 2) {     // 16 scalar mul
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(m));
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            } else if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
            } else if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
            } else if (MODE == 6) {     // packed add with the second operand's halves swapped (op_sel)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(a2));
            } else if (MODE == 7) {     // byte -> float conversions
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(s[i]));
            } else if (MODE == 8) {     // byte permute (2 bytes -> two f16 magic halves in one op)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m), "v"(a));
            } else if (MODE == 9) {     // packed f16 add (the -1024 of the magic-number byte->f16 conversion)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(s[i]) : "v"(a));
            } else if (MODE == 10) {    // mixed-precision fma reading an f16 half directly
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(s[i]) : "v"(m), "v"(a));
            } else if (MODE == 11) {    // clip + round + pack to u8
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(s[i]) : "v"(m));
            } else if (MODE == 12) {    // float64 fma (the scalar stage)
                double *dd = reinterpret_cast<double *>(s);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[i]) : "v"((double)m), "v"((double)a));
            } else if (MODE == 13) {    // SDWA: integer byte select + convert in one VOP1
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "+v"(s[i]));
            } else if (MODE == 14) {    // 4-byte integer dot product
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m), "v"(a));
            } else if (MODE == 15) {    // rcp (quarter-rate class)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
            } else if (MODE == 16) {    // fma with |x| source modifier (VOP3 encoding)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(s[i]) : "v"(m), "v"(a));
            } else if (MODE == 17) {    // fma with a literal constant (v_fmamk_f32)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(s[i]) : "v"(a));
            } else if (MODE == 18) {    // sub
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i];
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
void run(const char *name, int elems_per_instr) {
    float *d;
    hipMalloc(&d, 4);
    const int iters = 2000, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int per_iter = 8 * (elems_per_instr == 2 ? 8 : 16);
    const double instrs = (double)blocks * 4 * iters * per_iter;          // wave-instructions
    const double per_simd = instrs / (256.0 * 4);
    printf("%-28s %8.3f ms  %.2f ns per wave-instruction per SIMD  -> %.1f G lane-elements/s\n", name, ms,
           ms * 1e6 / per_simd, instrs * 64 * elems_per_instr / (ms * 1e-3) / 1e9);
    hipFree(d);
}

int main() {
    run<0>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 2);
    run<2>("v_mul_f32", 1);
    run<3>("v_pk_mul_f32", 2);
    run<4>("v_add_f32", 1);
    run<5>("v_pk_add_f32", 2);
    run<6>("v_pk_add_f32 op_sel swap", 2);
    run<7>("v_cvt_f32_ubyte1", 1);
    run<8>("v_perm_b32", 1);
    run<9>("v_pk_add_f16", 1);
    run<10>("v_fma_mix_f32 (f16 src)", 1);
    run<11>("v_cvt_pk_u8_f32", 1);
    run<12>("v_fma_f64", 2);
    run<13>("v_cvt_f32_u32_sdwa BYTE_2", 1);
    run<14>("v_dot4_u32_u8", 1);
    run<15>("v_rcp_f32", 1);
    run<16>("v_fma_f32 |src|", 1);
    run<17>("v_fmamk_f32 literal", 1);
    run<18>("v_sub_f32", 1);
    return 0;
}
