// offmark_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels + C ABI for the offmark
// DCT frame-watermark path.  Built with:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize
//
// Reference behaviour restated here (paths relative to the reference root):
//   src/offmark/video/embedder.py:33-39      u8 -> f32 -> "BGR2YUV" -> encode -> "YUV2BGR" -> clip/round/u8
//   src/offmark/embed/dct_encoder.py:18-102  masks from the Y block DCTs, QIM on U coefficient [2][1]
//   src/offmark/extract/dct_decoder.py:10-27 same masks, bit = round(c21/step) odd
//   src/offmark/degenerator/de_shuffler.py:17-18  sums of bits[i::L] (the mean's numerator)
//
// Kernels (DESIGN.md has the full story):
//   analyze  : frame pixels -> 3 floats per 8x8 block {A00, texture-mask code, C21} + a fixed-point
//              sum of the block DCs (the luminance mask needs the frame-global mean first).
//              Shared by embed and detect.  Reads 3 B/px, writes 0.19 B/px.
//   finalize : one thread per block: luminance mask (float64 like the reference), step, then
//              either the QIM delta of C21 (embed) or the read-out bit + bits[i::L] counts.
//   mark     : frame pixels + delta -> marked pixels.  The 8x8 DCT is orthonormal, so
//              idct(dct(U) + d*e21) == U + d * outer(c2, c1): a rank-1 update, no DCT needed.
//              The FUSED variant also analyzes the marked block it has just produced (mark +
//              verify), which saves detect's read of the marked frame.
//
// Work decomposition: ONE THREAD PER 8x8 BLOCK.  A lane owns 8 rows x 24 contiguous bytes; the 64
// lanes of a wavefront own 64 adjacent blocks, so every row load of a wave covers 1536 contiguous
// bytes.  Row DCTs, column DCTs and all per-block sums stay in that lane's registers: no LDS, no
// barrier, no cross-lane traffic.  (A first version staged the 8x8 transpose through LDS with
// 8-lane DPP sums; rocprofv3 showed VALU ~58 % and the LDS pipe ~55 % busy at once -- see
// profiles/ -- and this form needs ~20 % fewer VALU instructions and none of the LDS ones.)
// It also lets the texture-feature sums follow numpy's exact association order.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/offmark_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRec = 3;           // float planes per block record: A00, texture code, C21
constexpr int kSlots = 32;        // fixed-point mean accumulators per frame (spreads atomics)
constexpr int kHistMax = 2048;    // payload lengths up to this use an LDS histogram in finalize

// minimum waves per SIMD the register allocator must leave room for (tuned on MI355X, profiles/)
#ifndef OFMK_ANALYZE_WAVES
#define OFMK_ANALYZE_WAVES 4
#endif
#ifndef OFMK_PREFETCH_ROWS
#define OFMK_PREFETCH_ROWS 4
#endif
#ifndef OFMK_ROW_BARRIER
#define OFMK_ROW_BARRIER 1
#endif
#ifndef OFMK_FUSED_WAVES
#define OFMK_FUSED_WAVES 3
#endif

constexpr int SRC_RGB8 = 0;
constexpr int SRC_YUV32F = 1;

// 0.5*cos(k*pi/16), k = 1..7  (orthonormal 8-point DCT-II basis; k = 4 doubles as the DC scale)
constexpr float H1 = 0.49039264f, H2 = 0.46193977f, H3 = 0.41573481f, H4 = 0.35355339f,
                H5 = 0.27778512f, H6 = 0.19134172f, H7 = 0.09754516f;

// OpenCV float "YUV" constants
constexpr float KY0 = 0.114f, KY1 = 0.587f, KY2 = 0.299f, KU = 0.492f, KV = 0.877f, KDELTA = 0.5f;
constexpr float KI_B = 2.032f, KI_GU = -0.395f, KI_GV = -0.581f;

__device__ __forceinline__ constexpr float c1_of(int x) {   // 0.5*cos((2x+1)*1*pi/16): row basis of [2][1]
    return x == 0 ? H1 : x == 1 ? H3 : x == 2 ? H5 : x == 3 ? H7 : x == 4 ? -H7 : x == 5 ? -H5 : x == 6 ? -H3 : -H1;
}
__device__ __forceinline__ constexpr float c2_of(int r) {   // 0.5*cos((2r+1)*2*pi/16): column basis of [2][1]
    return (r == 0 || r == 7) ? H2 : (r == 1 || r == 6) ? H6 : (r == 2 || r == 5) ? -H6 : -H2;
}

// In-place orthonormal 8-point DCT-II, even/odd decomposition: 36 VALU ops.  The first butterfly
// stage makes every AC output of a constant or mirror-symmetric input an exact zero.
__device__ __forceinline__ void dct8(float (&x)[8]) {
    const float a0 = x[0] + x[7], a1 = x[1] + x[6], a2 = x[2] + x[5], a3 = x[3] + x[4];
    const float b0 = x[0] - x[7], b1 = x[1] - x[6], b2 = x[2] - x[5], b3 = x[3] - x[4];
    const float e0 = a0 + a3, e1 = a1 + a2, e2 = a0 - a3, e3 = a1 - a2;
    x[0] = (e0 + e1) * H4;
    x[4] = (e0 - e1) * H4;
    x[2] = fmaf(e3, H6, e2 * H2);
    x[6] = fmaf(e3, -H2, e2 * H6);
    x[1] = fmaf(b3, H7, fmaf(b2, H5, fmaf(b1, H3, b0 * H1)));
    x[3] = fmaf(b3, -H5, fmaf(b2, -H1, fmaf(b1, -H7, b0 * H3)));
    x[5] = fmaf(b3, H3, fmaf(b2, H7, fmaf(b1, -H1, b0 * H5)));
    x[7] = fmaf(b3, -H1, fmaf(b2, H3, fmaf(b1, -H5, b0 * H7)));
}

// coefficient 1 of the 8-point DCT only (8 ops), butterfly first
__device__ __forceinline__ float proj1(const float (&v)[8]) {
    return fmaf(v[3] - v[4], H7, fmaf(v[2] - v[5], H5, fmaf(v[1] - v[6], H3, (v[0] - v[7]) * H1)));
}
__device__ __forceinline__ void divmod_small(int c, int d, float inv_d, int &q, int &r) {
    q = (int)((float)c * inv_d);       // c < 2^24 is checked on the host
    r = c - q * d;
    if (r < 0) { q -= 1; r += d; }
    else if (r >= d) { q += 1; r -= d; }
}

// Sum of one int per lane over the 64-lane wavefront; result is wave-uniform (an SGPR).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_i<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_i<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_i<0x141>(v);    // row_half_mirror
    v += dpp_i<0x140>(v);    // row_mirror: every lane of a 16-lane row now holds the row sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// ------------------------------------------------------------------------------------------
// pixel access
// ------------------------------------------------------------------------------------------
struct Px8 { uint32_t w[6]; };   // 8 interleaved u8 RGB pixels = 24 bytes

#ifndef OFMK_NT_LOAD
#define OFMK_NT_LOAD 0
#endif
#ifndef OFMK_NT_STORE
#define OFMK_NT_STORE 0
#endif
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool ALIGNED>
__device__ __forceinline__ Px8 load_px8(const uint8_t *p) {
    Px8 v;
    if constexpr (ALIGNED) {
#if OFMK_NT_LOAD
        const u32x2 *q = reinterpret_cast<const u32x2 *>(p);
        const u32x2 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1), c = __builtin_nontemporal_load(q + 2);
#else
        const uint2 *q = reinterpret_cast<const uint2 *>(p);
        const uint2 a = q[0], b = q[1], c = q[2];
#endif
        v.w[0] = a.x; v.w[1] = a.y; v.w[2] = b.x; v.w[3] = b.y; v.w[4] = c.x; v.w[5] = c.y;
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            v.w[k] = (uint32_t)p[4 * k] | ((uint32_t)p[4 * k + 1] << 8) | ((uint32_t)p[4 * k + 2] << 16) |
                     ((uint32_t)p[4 * k + 3] << 24);
    }
    return v;
}

template <bool ALIGNED>
__device__ __forceinline__ void store_px8(uint8_t *p, const Px8 &v) {
    if constexpr (ALIGNED) {
#if OFMK_NT_STORE
        u32x2 *q = reinterpret_cast<u32x2 *>(p);
        u32x2 a = {v.w[0], v.w[1]}, b = {v.w[2], v.w[3]}, c = {v.w[4], v.w[5]};
        __builtin_nontemporal_store(a, q); __builtin_nontemporal_store(b, q + 1); __builtin_nontemporal_store(c, q + 2);
#else
        uint2 *q = reinterpret_cast<uint2 *>(p);
        q[0] = make_uint2(v.w[0], v.w[1]);
        q[1] = make_uint2(v.w[2], v.w[3]);
        q[2] = make_uint2(v.w[4], v.w[5]);
#endif
    } else {
#pragma unroll
        for (int k = 0; k < 24; ++k) p[k] = (uint8_t)(v.w[k >> 2] >> (8 * (k & 3)));
    }
}

// Make the compiler forget what it knows about these registers.  Used where a kernel deliberately
// RECOMPUTES per-pixel values from the raw bytes in a second pass: without it, CSE keeps the first
// pass's 128+ floats alive across the whole SVD and spills.
__device__ __forceinline__ void forget(Px8 &v) {
#pragma unroll
    for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(v.w[k]));
}

__device__ __forceinline__ float px_byte(const Px8 &v, int k) {   // k is a compile-time constant after unrolling
    return (float)((v.w[k >> 2] >> (8 * (k & 3))) & 0xffu);       // -> v_cvt_f32_ubyteN
}

// cvtColor BGR2YUV for one 8-pixel row, per pixel and in OpenCV's fma order, so that a chroma-flat
// block yields bit-identical U samples and hence an exactly zero C21 (np.sign(0) == 0).
__device__ __forceinline__ void row_yu(const Px8 &px, float (&y)[8], float (&u)[8]) {
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const float c0 = px_byte(px, 3 * x);
        y[x] = fmaf(c0, KY0, fmaf(px_byte(px, 3 * x + 1), KY1, px_byte(px, 3 * x + 2) * KY2));
        u[x] = fmaf(c0 - y[x], KU, KDELTA);
    }
}

// ------------------------------------------------------------------------------------------
// geometry and per-block features
// ------------------------------------------------------------------------------------------
struct Geom {
    int W;                // pixels per row
    int wb;               // blocks per block-row (W / 8)
    float inv_wb;
    int nblk;             // (H/8)*(W/8)
    size_t frame_stride;  // elements (bytes for u8, floats for f32) between frames
    size_t plane;         // elements between record planes (frames in flight * nblk)
};

struct BlockFeat { float a00, tex, c21; };

// texture_mask (dct_encoder.py:70-102) as a one-float code: 1, 1.125 or 1.25 stand for themselves
// (exact in float32); a negative value -eh means "the ramp 1 + 1.25*(eh - 290)/1510", which
// finalize evaluates in float64 from the float32 eh, as numpy 1.23 does (eh > 125 there, so the
// sign is unambiguous).  float32 arithmetic exactly where the reference's numpy scalars are
// float32; comparisons with python floats are float64 because the reference pins numpy 1.23
// (legacy promotion: np.float32 scalar (op) python scalar -> float64).  l/e and (l+e)/h may be
// inf or nan; IEEE comparisons with nan are false, as in the reference.
__device__ __forceinline__ float texture_code(float a00abs, float tot, float dcl, float e) {
    // branch-free: everything is computed, the decision tree becomes selects (no control flow in
    // the middle of a register-heavy kernel)
    const float eh = tot - dcl;
    const float h = eh - e;
    const float l = dcl - a00abs;
    const float l_e = l / e;
    const float lpe = l + e;
    const float le_h = lpe / h;
    const bool active = eh > 125.f;
    const bool big = eh > 900.f;
    const double a = big ? 1.4 : 2.3, b = big ? 1.1 : 1.6;
    const double dl_e = (double)l_e, dle_h = (double)le_h;
    const bool cond = (dl_e >= a && dle_h >= b) || (dl_e >= b && dle_h >= a) || (le_h > 4.f);
    const bool ramp = big || (e + h > 290.f);
    const float stepped = lpe <= 400.f ? 1.125f : 1.25f;
    const float inner = cond ? stepped : (ramp ? -eh : 1.0f);
    return active ? inner : 1.0f;
}
__device__ __forceinline__ double texture_value(float code) {
    return code > 0.f ? (double)code : 1.0 + 1.25 * ((double)(-code) - 290.0) / 1510.0;
}


// R[r][k]: row-DCT outputs of the Y block (row r, horizontal frequency k); u1[r]: k=1 projection
// of the U rows.  Column DCTs in place, then the texture-mask features with the reference's own
// association order (dct_encoder.py:80-86; np.sum = 8 running column sums combined pairwise).
// u1[r] for r < 4 first holds row r's projection; when row 7-r arrives it becomes
// a_r = p[r] + p[7-r], the first butterfly stage of the 8-point DCT's coefficient 2
// (same association as dct8 / the oracle, so a vertically symmetric U gives an exact zero).
__device__ __forceinline__ void fold_u1(float (&u1)[4], int r, float p) {
    // Opaque use: C21 is only stored under `if (valid)` at the very end, and without this LLVM sinks
    // the whole U chain (64 pixels' c0 and Y) down into that branch -- 300 bytes of spills per lane.
    asm volatile("" : "+v"(p));
    if (r < 4) u1[r] = p; else u1[7 - r] += p;
}

__device__ __forceinline__ BlockFeat block_features(float (&R)[8][8], const float (&u1)[4]) {
    // One column at a time: DCT it, fold |coefficients| into numpy's 8 running sums (np.sum of the
    // contiguous 8x8 block keeps one accumulator per column j and adds the rows in order), pick the
    // 18 coefficients dcl and e need, and let the column die.  The scheduling barrier pins this order;
    // without it the compiler interleaves all eight columns and spills.
    float rs[8], A[4][4], A0[8], Ai0[8];     // A[i][j] for i,j < 4; A0[j] = |A[0][j]|; Ai0[i] = |A[i][0]|
    BlockFeat ft;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float col[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) col[i] = R[i][j];
        dct8(col);
        if (j == 0) ft.a00 = col[0];
#pragma unroll
        for (int i = 0; i < 8; ++i) col[i] = fabsf(col[i]);
        rs[j] = col[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) rs[j] += col[i];
        A0[j] = col[0];
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) Ai0[i] = col[i];
        }
        if (j < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) A[i][j] = col[i];
        }
#if OFMK_ROW_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    const float tot = ((rs[0] + rs[1]) + (rs[2] + rs[3])) + ((rs[4] + rs[5]) + (rs[6] + rs[7]));
    const float dcl = ((((A[0][0] + A[0][1]) + A[0][2]) + A[1][0]) + A[1][1]) + A[2][0];
    const float e = ((((((((((Ai0[3] + Ai0[4]) + Ai0[5]) + Ai0[6]) + A0[3]) + A0[4]) + A0[5]) + A0[6]) +
                     A[2][1]) + A[1][2]) + A[2][2]) + A[3][3];
    ft.tex = texture_code(A[0][0], tot, dcl, e);
    ft.c21 = fmaf(u1[1] - u1[2], H6, (u1[0] - u1[3]) * H2);   // u1[] holds the folded sums a_r (see fold_u1)
    return ft;
}

// Records are three planes of [frames][nblk] floats; the frame's block DCs also go, as 2^19 fixed
// point, into one of kSlots 64-bit accumulators: integer adds commute, so the frame mean is
// bit-reproducible however the workgroups are scheduled.
__device__ __forceinline__ void emit_block(const BlockFeat &ft, bool valid, int f, int c, const Geom &g,
                                           float *__restrict__ rec, unsigned long long *__restrict__ ysum) {
    if (valid) {
        float *r = rec + (size_t)f * g.nblk + c;
        r[0] = ft.a00;
        r[g.plane] = ft.tex;
        r[2 * g.plane] = ft.c21;
    }
    const int q = valid ? __float2int_rn(ft.a00 * 524288.0f) : 0;     // A00 * 2^19, |A00| <= 2040
    const int lo = wave_sum(q & 0xffff), hi = wave_sum(q >> 16);
    if ((threadIdx.x & 63) == 0) {
        const long long s = (long long)hi * 65536 + lo;
        const int slot = (blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) & (kSlots - 1);
        if (s != 0) atomicAdd(&ysum[(size_t)f * kSlots + slot], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------
// per-block scalar stage shared by finalize and mark (float64 like the reference)
// ------------------------------------------------------------------------------------------
// Frame-global mean of the block means A00/8 (luminance_mask, dct_encoder.py:54-56) from the
// fixed-point accumulators; the first wavefront leaves it in LDS.  Caller adds the barrier.
__device__ __forceinline__ void frame_mean_to_lds(const unsigned long long *__restrict__ ysum, int f, int nblk,
                                                  double *s_mean) {
    const int t = threadIdx.x;
    if (t < 64) {
        long long s = t < kSlots ? (long long)ysum[(size_t)f * kSlots + t] : 0;
#pragma unroll
        for (int d = 1; d < kSlots; d <<= 1) s += __shfl_xor(s, d);
        if (t == 0) *s_mean = ((double)s * (1.0 / 4194304.0)) / (double)nblk;   // sum(A00 * 2^19) -> mean(A00 / 8)
    }
}

// luminance_mask's per-block branch (dct_encoder.py:57-66) for block mean m = A00/8
__device__ __forceinline__ double luminance_value(float a00, double mean_m) {
    const double mean = mean_m > 90.0 ? mean_m : 90.0;
    const double f_ref = 1.0 + (mean - 90.0) * 1.0 / 165.0;
    const double m = (double)a00 / 8.0;
    if (m > mean) return 1.0 + (m - mean) / (255.0 - mean) * (2.0 - f_ref);
    if (m < 15.0) return 1.25;
    if (m < 25.0) return 1.125;
    return 1.0;
}

// QIM, dct_encoder.py:30-35 (float64 on a float32 coefficient; np.sign(0) == 0 keeps a zero at zero)
__device__ __forceinline__ float qim_new_coeff(float c21, double step, int bit) {
    const double step2 = step + step;
    double q = floor(fabs((double)c21) / step2) * step2;
    if (bit) q = q + step;
    const double nv = c21 > 0.f ? q : (c21 < 0.f ? -q : 0.0);
    return (float)nv;
}

// ------------------------------------------------------------------------------------------
// analyze
// ------------------------------------------------------------------------------------------
template <int SRC, bool ALIGNED>
__global__ __launch_bounds__(kThreads, OFMK_ANALYZE_WAVES) void analyze_kernel(const void *__restrict__ frames, Geom g,
                                                           float *__restrict__ rec,
                                                           unsigned long long *__restrict__ ysum) {
    const int f = blockIdx.y;
    const int c = blockIdx.x * kThreads + threadIdx.x;
    const bool valid = c < g.nblk;
    int bi, bj;
    divmod_small(valid ? c : g.nblk - 1, g.wb, g.inv_wb, bi, bj);     // ragged tail recomputes the last block
    const size_t off = (size_t)f * g.frame_stride + ((size_t)bi * 8 * g.W + (size_t)bj * 8) * 3;
    const int pitch = g.W * 3;
    float R[8][8], u1[4];
    if constexpr (SRC == SRC_RGB8) {
        const uint8_t *p = static_cast<const uint8_t *>(frames) + off;
        // rolling prefetch: OFMK_PREFETCH_ROWS rows of raw bytes in flight, each row consumed (colour
        // transform + row DCT) as a unit; the scheduling barrier keeps the compiler from hoisting
        // every row's conversions to the top, which costs ~200 VGPRs and half the occupancy.
        Px8 raw[8];
#pragma unroll
        for (int r = 0; r < OFMK_PREFETCH_ROWS; ++r) raw[r] = load_px8<ALIGNED>(p + (size_t)r * pitch);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r + OFMK_PREFETCH_ROWS < 8)
                raw[r + OFMK_PREFETCH_ROWS] = load_px8<ALIGNED>(p + (size_t)(r + OFMK_PREFETCH_ROWS) * pitch);
            float y[8], u[8];
            row_yu(raw[r], y, u);
            fold_u1(u1, r, proj1(u));
            dct8(y);
#pragma unroll
            for (int k = 0; k < 8; ++k) R[r][k] = y[k];
#if OFMK_ROW_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    } else {
        const float *p = static_cast<const float *>(frames) + off;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float y[8], u[8];
            const float *q = p + (size_t)r * pitch;
            if constexpr (ALIGNED) {
                const float4 *q4 = reinterpret_cast<const float4 *>(q);
                float4 v[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) v[k] = q4[k];
                const float *e = reinterpret_cast<const float *>(v);
#pragma unroll
                for (int x = 0; x < 8; ++x) { y[x] = e[3 * x]; u[x] = e[3 * x + 1]; }
            } else {
#pragma unroll
                for (int x = 0; x < 8; ++x) { y[x] = q[3 * x]; u[x] = q[3 * x + 1]; }
            }
            fold_u1(u1, r, proj1(u));
            dct8(y);
#pragma unroll
            for (int k = 0; k < 8; ++k) R[r][k] = y[k];
        }
    }
    const BlockFeat ft = block_features(R, u1);
    emit_block(ft, valid, f, c, g, rec, ysum);
}

// ------------------------------------------------------------------------------------------
// finalize
// ------------------------------------------------------------------------------------------
struct FinArgs {
    const float *rec;                 // 3 planes of [frames][nblk]: A00, texture code, C21
    size_t plane;
    const unsigned long long *ysum;   // [frames][kSlots]
    int nblk, N, L;
    double alpha;
    const uint8_t *wm;                // [n_wm][N] or null
    const int32_t *wm_row;            // [frames] or null
    float *delta;                     // [frames][nblk]   embed
    int32_t *counts;                  // [frames][L]      detect
    uint8_t *bits;                    // [frames][N]      detect (optional)
    float *y_dc;                      // debug planes, [frames][nblk]
    double *lum, *tex, *step;
    float *c21_pre, *c21_post;
};

__global__ __launch_bounds__(kThreads) void finalize_kernel(FinArgs p) {
    __shared__ int hist[kHistMax];
    __shared__ double s_mean;
    const int t = threadIdx.x;
    const int f = blockIdx.y;
    const int c = blockIdx.x * kThreads + t;
    const bool use_hist = p.counts != nullptr && p.L <= kHistMax;
    if (use_hist)
        for (int k = t; k < p.L; k += kThreads) hist[k] = 0;
    frame_mean_to_lds(p.ysum, f, p.nblk, &s_mean);
    __syncthreads();
    if (c < p.nblk) {
        const float *r = p.rec + (size_t)f * p.nblk + c;
        const float a00 = r[0], tcode = r[p.plane], c21 = r[2 * p.plane];
        const double lum = luminance_value(a00, s_mean);
        const double tex = texture_value(tcode);
        const double step = p.alpha * (tex * lum);
        const size_t o = (size_t)f * p.nblk + c;
        if (p.y_dc) p.y_dc[o] = a00;
        if (p.lum) p.lum[o] = lum;
        if (p.tex) p.tex[o] = tex;
        if (p.step) p.step[o] = step;
        if (p.c21_pre) p.c21_pre[o] = c21;

        if (p.delta || p.c21_post) {
            const int row = p.wm_row ? p.wm_row[f] : 0;
            const float newc = qim_new_coeff(c21, step, p.wm[(size_t)row * p.N + c]);
            if (p.c21_post) p.c21_post[o] = newc;
            if (p.delta) p.delta[o] = newc - c21;
        }
        if (p.counts || p.bits) {
            // dct_decoder.py:24: int(np.around(c21/step) % 2 == 1)
            const double x = rint((double)c21 / step);
            const int bit = fmod(fabs(x), 2.0) == 1.0 ? 1 : 0;
            if (p.bits) p.bits[(size_t)f * p.N + c] = (uint8_t)bit;
            if (p.counts && bit) {
                const int pos = c % p.L;
                if (use_hist) atomicAdd(&hist[pos], 1);
                else atomicAdd(&p.counts[(size_t)f * p.L + pos], 1);
            }
        }
    } else if (c < p.N && p.bits) {
        p.bits[(size_t)f * p.N + c] = 0;     // dct_decoder.py:16: entries past (H/8)*(W/8) stay zero
    }
    if (use_hist) {
        __syncthreads();
        for (int k = t; k < p.L; k += kThreads) {
            const int v = hist[k];
            if (v) atomicAdd(&p.counts[(size_t)f * p.L + k], v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// mark
// ------------------------------------------------------------------------------------------
// np.clip(0,255) -> np.around (half to even) -> uint8, packed into byte `sel` of `word`:
// v_cvt_pk_u8_f32 does exactly that in one instruction (probed on gfx950: tools/probe.hip).
__device__ __forceinline__ uint32_t put_u8(float v, int sel, uint32_t word) {
    return __builtin_amdgcn_cvt_pk_u8_f32(v, sel, word);
}

struct MarkArgs {
    const float *rec;                 // records of the INPUT frames (from analyze)
    const unsigned long long *ysum;   // their mean accumulators
    const uint8_t *wm;                // [n_wm][N]
    const int32_t *wm_row;            // [frames] or null
    int N;
    double alpha;
};

// embedder.py:33-39 for one block per thread, including the per-block scalar stage (masks -> step ->
// QIM of C21, dct_encoder.py:21-35) that turns the input frame's record into this block's delta.
// FUSED: also analyze the marked block (detect's front end on the frame being written), producing
// its records (rec_out may alias m.rec: a thread reads its own entries before it overwrites them)
// and mean accumulator (ysum_out, a different buffer from m.ysum).
// SELF (experiment, tools/upper_bound.py): recompute the input block's record in this kernel instead of
// reading it, i.e. analyze + mark + verify in ONE pass given the frame mean from outside.  It bounds what
// a persistent kernel that keeps a block's pixels in registers across the mean dependency could reach.
template <bool ALIGNED, bool FUSED, bool SELF = false>
__global__ __launch_bounds__(kThreads, FUSED ? OFMK_FUSED_WAVES : 4) void mark_rgb8_kernel(const uint8_t *__restrict__ in,
                                                             uint8_t *__restrict__ out, Geom g, MarkArgs m,
                                                             float *rec_out,
                                                             unsigned long long *__restrict__ ysum_out) {
    __shared__ double s_mean;
    const int f = blockIdx.y;
    const int c = blockIdx.x * kThreads + threadIdx.x;
    const bool valid = c < g.nblk;
    frame_mean_to_lds(m.ysum, f, g.nblk, &s_mean);
    __syncthreads();
    if (!FUSED && !valid) return;
    const int cc = valid ? c : g.nblk - 1;
    int bi, bj;
    divmod_small(cc, g.wb, g.inv_wb, bi, bj);
    const size_t off = (size_t)f * g.frame_stride + ((size_t)bi * 8 * g.W + (size_t)bj * 8) * 3;
    const int pitch = g.W * 3;
    Px8 raw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) raw[r] = load_px8<ALIGNED>(in + off + (size_t)r * pitch);
    float R[8][8], u1[4];
    float d;
    {
        float a00, tcode, c21;
        if constexpr (SELF) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float y[8], u[8];
                row_yu(raw[r], y, u);
                fold_u1(u1, r, proj1(u));
                dct8(y);
#pragma unroll
                for (int k = 0; k < 8; ++k) R[r][k] = y[k];
#if OFMK_ROW_BARRIER
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            const BlockFeat fin = block_features(R, u1);
            a00 = fin.a00; tcode = fin.tex; c21 = fin.c21;
#pragma unroll
            for (int r = 0; r < 8; ++r) forget(raw[r]);
        } else {
            const float *r = m.rec + (size_t)f * g.nblk + cc;
            a00 = r[0]; tcode = r[g.plane]; c21 = r[2 * g.plane];
        }
        const double step = m.alpha * (texture_value(tcode) * luminance_value(a00, s_mean));
        const int row = m.wm_row ? m.wm_row[f] : 0;
        d = qim_new_coeff(c21, step, m.wm[(size_t)row * m.N + cc]) - c21;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const Px8 &px = raw[r];
        const float dr = d * c2_of(r);
        Px8 o = px;       // channel 2 bytes stay: Y + 1.140*(V-0.5) = c2 - 2.2e-4*(c2 - Y), |error| < 0.05
        float yv[8], uv[8];   // FUSED: Y and U of the MARKED row, i.e. of the rounded, clipped u8 pixels detect will see
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const float c0 = px_byte(px, 3 * x), c1 = px_byte(px, 3 * x + 1), c2 = px_byte(px, 3 * x + 2);
            const float t2 = c2 * KY2;
            const float y = fmaf(c0, KY0, fmaf(c1, KY1, t2));
            const float u = fmaf(c0 - y, KU, KDELTA);          // cvtColor BGR2YUV
            const float v = fmaf(c2 - y, KV, KDELTA);
            const float u2 = fmaf(dr, c1_of(x), u);            // idct(dct(U) + d*e21) = U + d*c2[r]*c1[x]
            const float ud = u2 - KDELTA, vd = v - KDELTA;     // cvtColor YUV2BGR
            o.w[(3 * x) >> 2] = put_u8(fmaf(ud, KI_B, y), (3 * x) & 3, o.w[(3 * x) >> 2]);
            o.w[(3 * x + 1) >> 2] = put_u8(fmaf(vd, KI_GV, fmaf(ud, KI_GU, y)), (3 * x + 1) & 3, o.w[(3 * x + 1) >> 2]);
            if constexpr (FUSED) {                             // channel 2 and its product are shared with the pass above
                const float n0 = px_byte(o, 3 * x), n1 = px_byte(o, 3 * x + 1);
                yv[x] = fmaf(n0, KY0, fmaf(n1, KY1, t2));
                uv[x] = fmaf(n0 - yv[x], KU, KDELTA);
            }
        }
        if (valid) store_px8<ALIGNED>(out + off + (size_t)r * pitch, o);
        if constexpr (FUSED) {
            fold_u1(u1, r, proj1(uv));
            dct8(yv);
#pragma unroll
            for (int k = 0; k < 8; ++k) R[r][k] = yv[k];
        }
    }
    if constexpr (FUSED) {
        const BlockFeat ft = block_features(R, u1);
        emit_block(ft, valid, f, c, g, rec_out, ysum_out);
    }
}

// DctEncoder.encode on float32 YUV: only channel 1 changes (dct_encoder.py:20,36-37)
__global__ __launch_bounds__(kThreads) void mark_yuv32f_kernel(float *__restrict__ yuv, Geom g,
                                                               const float *__restrict__ delta) {
    const int f = blockIdx.y;
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= g.nblk) return;
    int bi, bj;
    divmod_small(c, g.wb, g.inv_wb, bi, bj);
    float *p = yuv + (size_t)f * g.frame_stride + ((size_t)bi * 8 * g.W + (size_t)bj * 8) * 3;
    const int pitch = g.W * 3;
    const float d = delta[(size_t)f * g.nblk + c];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float dr = d * c2_of(r);
#pragma unroll
        for (int x = 0; x < 8; ++x) p[(size_t)r * pitch + 3 * x + 1] = fmaf(dr, c1_of(x), p[(size_t)r * pitch + 3 * x + 1]);
    }
}

// ------------------------------------------------------------------------------------------
// DwtDctSvd codec: embed/dwt_dct_svd_encoder.py:19-45, extract/dwt_dct_svd_decoder.py:12-37
// ------------------------------------------------------------------------------------------
// Per 8x8 pixel tile the reference takes the Haar LL band of channel 1 (a 4x4 block B), runs
// cv2.dct, np.linalg.svd, replaces the top singular value by (s0 // scale + 0.25 + 0.5*bit)*scale,
// multiplies back, cv2.idct, inverse Haar.  The 4x4 DCT is orthonormal, so dct(B) has B's singular
// values and u*diag(s')*v maps back to B + (s0' - s0) * u0 * v0^T: a rank-1 update of B, and the
// inverse Haar spreads each LL change over its 2x2 pixels with weight 1/2.  No DCT is computed here.
// There is no frame-global dependency: embed is ONE pass (6 B/px), detect one pass (3 B/px).

constexpr float kHaar = 0.70710678118654752f;   // pywt's haar taps in float32

// Top singular value (and, if WANT_UPDATE, the rank-1 update direction) of a 4x4 matrix by one-sided
// Jacobi (Hestenes): rotate column pairs until they are orthogonal; column norms are then the
// singular values.  Five fixed sweeps reach float32 accuracy for 4x4.
struct Svd4 { float s0; float w[4]; float z[4]; };     // w = s0*u0, z = B^T w = s0^2 * v0

template <bool WANT_UPDATE>
__device__ __forceinline__ Svd4 svd4_top(const float (&B)[4][4]) {
    float A[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) A[i][j] = B[i][j];
#ifndef OFMK_SVD_SWEEPS
#define OFMK_SVD_SWEEPS 4   // reaches float32 accuracy on 4x4 (6e-7 max relative error; 5 and 6 sweeps give the same)
#endif
#pragma unroll 1
    for (int sweep = 0; sweep < OFMK_SVD_SWEEPS; ++sweep) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                float alpha = 0.f, beta = 0.f, gamma = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    alpha = fmaf(A[i][p], A[i][p], alpha);
                    beta = fmaf(A[i][q], A[i][q], beta);
                    gamma = fmaf(A[i][p], A[i][q], gamma);
                }
                // rotation angle that zeroes the pair's inner product; gamma == 0 -> identity (t = 0).
                // Hardware reciprocal / rsqrt (1 ulp) are enough here: Jacobi is self-correcting, any
                // near-orthogonal rotation that shrinks gamma converges to the same singular values.
                const float zeta = (beta - alpha) * __builtin_amdgcn_rcpf(2.f * gamma);
                float t = copysignf(1.f, zeta) * __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.f)));
                t = (gamma == 0.f || !(fabsf(zeta) < 3.0e38f)) ? 0.f : t;
                const float c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
                const float sn = c * t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ap = A[i][p], aq = A[i][q];
                    A[i][p] = fmaf(c, ap, -sn * aq);
                    A[i][q] = fmaf(sn, ap, c * aq);
                }
            }
    }
    float n[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) n[j] = fmaf(A[3][j], A[3][j], fmaf(A[2][j], A[2][j], fmaf(A[1][j], A[1][j], A[0][j] * A[0][j])));
    int k = 0;
    float best = n[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) if (n[j] > best) { best = n[j]; k = j; }
    Svd4 r;
    r.s0 = sqrtf(best);
    if constexpr (WANT_UPDATE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r.w[i] = k == 0 ? A[i][0] : k == 1 ? A[i][1] : k == 2 ? A[i][2] : A[i][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) r.z[j] = fmaf(B[3][j], r.w[3], fmaf(B[2][j], r.w[2], fmaf(B[1][j], r.w[1], B[0][j] * r.w[0])));
    }
    return r;
}

// Top singular value only (detect / verify): classical two-sided Jacobi on the symmetric 4x4 Gram matrix
// G = B^T B, whose largest eigenvalue is s0^2.  ~21 VALU ops per rotation instead of ~48, no vectors.
// Three sweeps leave <= 7e-5 relative error in s0 (float32 emulation over 10^4 blocks incl. random ones);
// the read-out decides (s0 mod scale) > scale/2 on values the embedder put at scale/4 or 3*scale/4.
__device__ __forceinline__ float svd4_top_value(const float (&B)[4][4]) {
    float G[4][4];                       // upper triangle used: G[i][j], i <= j
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j)
            G[i][j] = fmaf(B[3][i], B[3][j], fmaf(B[2][i], B[2][j], fmaf(B[1][i], B[1][j], B[0][i] * B[0][j])));
#pragma unroll 1
    for (int sweep = 0; sweep < 3; ++sweep) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const float gpq = G[p][q];
                const float zeta = (G[q][q] - G[p][p]) * __builtin_amdgcn_rcpf(2.f * gpq);
                float t = copysignf(1.f, zeta) * __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.f)));
                t = (gpq == 0.f || !(fabsf(zeta) < 3.0e38f)) ? 0.f : t;
                const float c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f));
                const float sn = c * t;
                G[p][p] = fmaf(-t, gpq, G[p][p]);
                G[q][q] = fmaf(t, gpq, G[q][q]);
                G[p][q] = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r == p || r == q) continue;
                    float &grp = r < p ? G[r][p] : G[p][r];
                    float &grq = r < q ? G[r][q] : G[q][r];
                    const float a = grp, b = grq;
                    grp = fmaf(c, a, -sn * b);
                    grq = fmaf(sn, a, c * b);
                }
            }
    }
    const float lam = fmaxf(fmaxf(G[0][0], G[1][1]), fmaxf(G[2][2], G[3][3]));
    return sqrtf(fmaxf(lam, 0.f));
}

// np.float32 floor division by a positive scale (numpy: via fmod, exact for near-integers)
__device__ __forceinline__ float floor_div_pos(float a, float b) {
    const float m = fmodf(a, b);
    return rintf((a - m) / b);
}

// One row of the Haar LL band straight from two rows of u8 pixels.  LL = (sum of the 2x2 group's U)/2 and
// U = (c0 - Y)*0.492 + 0.5 with Y linear in the channels, so
//   LL = 0.246 * ((1 - 0.114)*S0 - 0.587*S1 - 0.299*S2) + 1,   S_k = sum of channel k over the 2x2 pixels
// (exact small integers).  24 VALU ops per LL entry instead of 41 via four per-pixel colour transforms;
// it differs from the reference's per-pixel float32 chain by ~1e-5 on values up to ~220, the same size as
// that chain's own rounding noise.
__device__ __forceinline__ void haar_ll_row_px(const Px8 &top, const Px8 &bot, float (&brow)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float S[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            S[k] = (px_byte(top, 6 * j + k) + px_byte(top, 6 * j + 3 + k)) + (px_byte(bot, 6 * j + k) + px_byte(bot, 6 * j + 3 + k));
        const float t = fmaf(S[2], -KY2, fmaf(S[1], -KY1, S[0] * (1.f - KY0)));
        brow[j] = fmaf(t, 0.5f * KU, 1.0f);
    }
}

// One row of the Haar LL band from two pixel rows of U, in PyWavelets' order: axis -2 (the row
// pair) first, then axis -1.  Building B row pair by row pair means U itself is never stored.
__device__ __forceinline__ void haar_ll_row(const float (&top)[8], const float (&bot)[8], float (&brow)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float lo0 = top[2 * j] * kHaar + bot[2 * j] * kHaar;
        const float lo1 = top[2 * j + 1] * kHaar + bot[2 * j + 1] * kHaar;
        brow[j] = lo0 * kHaar + lo1 * kHaar;
    }
}

// The quantisation step of dwt_dct_svd_encoder.py:44 as a change of B: dB[i][j] (already halved for
// the inverse Haar, i.e. the amount to add to each of the 2x2 pixels' U).
__device__ __forceinline__ void svd_update(const float (&B)[4][4], int bit, float scale, float (&dU)[4][4]) {
    const Svd4 r = svd4_top<true>(B);
    const float s_new = (floor_div_pos(r.s0, scale) + 0.25f + 0.5f * (float)bit) * scale;
    if (r.s0 > 0.f) {
        const float g = 0.5f * (s_new - r.s0) / (r.s0 * r.s0 * r.s0);      // (s0'-s0) * (w/s0) (z/s0^2)^T / 2
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dU[i][j] = g * r.w[i] * r.z[j];
    } else {                                                                 // LAPACK on a zero block: u0 = v0 = e0
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dU[i][j] = (i == 0 && j == 0) ? 0.5f * s_new : 0.f;
    }
}

__device__ __forceinline__ int svd_read_bit(const float (&B)[4][4], float scale) {
    return fmodf(svd4_top_value(B), scale) > scale * 0.5f ? 1 : 0;           // dwt_dct_svd_decoder.py:36
}

struct SvdArgs {
    const uint8_t *wm;        // [n_wm][N]           (embed)
    const int32_t *wm_row;    // [frames] or null
    int32_t *counts;          // [frames][L] or null (detect / verify)
    uint8_t *bits;            // [frames][N] or null
    int N, L;
    float scale;
};

constexpr int SVD_DETECT = 0, SVD_EMBED = 1, SVD_EMBED_VERIFY = 2;

template <bool ALIGNED, int MODE>
__global__ __launch_bounds__(kThreads, 3) void svd_rgb8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                               Geom g, SvdArgs a) {
    __shared__ int hist[kHistMax];
    const int t = threadIdx.x;
    const int f = blockIdx.y;
    const int c = blockIdx.x * kThreads + t;
    const bool valid = c < g.nblk;
    const bool want_counts = MODE != SVD_EMBED && a.counts != nullptr;
    const bool use_hist = want_counts && a.L <= kHistMax;
    if (use_hist) {
        for (int k = t; k < a.L; k += kThreads) hist[k] = 0;
        __syncthreads();
    }
    const int cc = valid ? c : g.nblk - 1;
    int bi, bj;
    divmod_small(cc, g.wb, g.inv_wb, bi, bj);
    const size_t off = (size_t)f * g.frame_stride + ((size_t)bi * 8 * g.W + (size_t)bj * 8) * 3;
    const int pitch = g.W * 3;
    Px8 raw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) raw[r] = load_px8<ALIGNED>(in + off + (size_t)r * pitch);
    float B[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) haar_ll_row_px(raw[2 * i], raw[2 * i + 1], B[i]);
    int bit = 0;
    if constexpr (MODE == SVD_DETECT) {
        bit = svd_read_bit(B, a.scale);
    } else {
        float dU[4][4];
        const int row = a.wm_row ? a.wm_row[f] : 0;
        svd_update(B, a.wm[(size_t)row * a.N + cc], a.scale, dU);
        Px8 oprev;                                             // marked pixels of the even row of a pair (verify)
#pragma unroll
        for (int r = 0; r < 8; ++r) forget(raw[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const Px8 &px = raw[r];
            Px8 o = px;                                        // channel 2 is untouched (see mark_rgb8_kernel)
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float c0 = px_byte(px, 3 * x), c1 = px_byte(px, 3 * x + 1), c2 = px_byte(px, 3 * x + 2);
                const float y = fmaf(c0, KY0, fmaf(c1, KY1, c2 * KY2));
                const float u = fmaf(c0 - y, KU, KDELTA);
                const float v = fmaf(c2 - y, KV, KDELTA);
                const float u2 = u + dU[r >> 1][x >> 1];
                const float ud = u2 - KDELTA, vd = v - KDELTA;
                o.w[(3 * x) >> 2] = put_u8(fmaf(ud, KI_B, y), (3 * x) & 3, o.w[(3 * x) >> 2]);
                o.w[(3 * x + 1) >> 2] = put_u8(fmaf(vd, KI_GV, fmaf(ud, KI_GU, y)), (3 * x + 1) & 3, o.w[(3 * x + 1) >> 2]);
            }
            if (valid) store_px8<ALIGNED>(out + off + (size_t)r * pitch, o);
            if constexpr (MODE == SVD_EMBED_VERIFY) {          // what the detector will see: the rounded u8 pixels
                if (r & 1) haar_ll_row_px(oprev, o, B[r >> 1]);
                else oprev = o;
            }
        }
        if constexpr (MODE == SVD_EMBED_VERIFY) bit = svd_read_bit(B, a.scale);
    }
    if constexpr (MODE != SVD_EMBED) {
        if (valid) {
            if (a.bits) a.bits[(size_t)f * a.N + c] = (uint8_t)bit;
            if (want_counts && bit) {
                const int pos = c % a.L;
                if (use_hist) atomicAdd(&hist[pos], 1);
                else atomicAdd(&a.counts[(size_t)f * a.L + pos], 1);
            }
        }
        if (use_hist) {
            __syncthreads();
            for (int k = t; k < a.L; k += kThreads) {
                const int v = hist[k];
                if (v) atomicAdd(&a.counts[(size_t)f * a.L + k], v);
            }
        }
    }
}

// Plugin-level float32 YUV frames: only channel 1 is read (and, for embed, written).
template <int MODE>
__global__ __launch_bounds__(kThreads, 3) void svd_yuv32f_kernel(float *__restrict__ yuv, Geom g, SvdArgs a) {
    const int f = blockIdx.y;
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= g.nblk) return;
    int bi, bj;
    divmod_small(c, g.wb, g.inv_wb, bi, bj);
    float *p = yuv + (size_t)f * g.frame_stride + ((size_t)bi * 8 * g.W + (size_t)bj * 8) * 3 + 1;
    const int pitch = g.W * 3;
    float U[8][8], B[4][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int x = 0; x < 8; ++x) U[r][x] = p[(size_t)r * pitch + 3 * x];
#pragma unroll
    for (int i = 0; i < 4; ++i) haar_ll_row(U[2 * i], U[2 * i + 1], B[i]);
    if constexpr (MODE == SVD_DETECT) {
        a.bits[(size_t)f * a.N + c] = (uint8_t)svd_read_bit(B, a.scale);
    } else {
        float dU[4][4];
        const int row = a.wm_row ? a.wm_row[f] : 0;
        svd_update(B, a.wm[(size_t)row * a.N + c], a.scale, dU);
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int x = 0; x < 8; ++x) p[(size_t)r * pitch + 3 * x] = U[r][x] + dU[r >> 1][x >> 1];
    }
}

// DeShuffler.degenerate's epilogue on the device (de_shuffler.py:17-22) for a batch of frames:
// mean of bits[i::L] from the counts, undo the key permutation, threshold strictly above the
// mid-range of the L means.  One workgroup per frame; float64 like the reference.
__global__ __launch_bounds__(kThreads) void degenerate_kernel(const int32_t *__restrict__ counts, int L, int n_bits,
                                                              const int32_t *__restrict__ perm,
                                                              uint8_t *__restrict__ payload) {
    __shared__ double s_max[kThreads], s_min[kThreads];
    __shared__ int s_nan[kThreads];
    const int t = threadIdx.x;
    const int f = blockIdx.x;
    double mx = -1.0, mn = 2.0;      // means are in [0, 1]
    int has_nan = 0;
    for (int i = t; i < L; i += kThreads) {
        const int len = i < n_bits ? (n_bits - i + L - 1) / L : 0;     // entries of bits[i::L]
        if (len == 0) { has_nan = 1; continue; }                        // numpy: mean of empty slice = nan
        const double m = (double)counts[(size_t)f * L + i] / (double)len;
        mx = m > mx ? m : mx;
        mn = m < mn ? m : mn;
    }
    s_max[t] = mx; s_min[t] = mn; s_nan[t] = has_nan;
    __syncthreads();
    for (int d = kThreads / 2; d > 0; d >>= 1) {
        if (t < d) {
            s_max[t] = s_max[t + d] > s_max[t] ? s_max[t + d] : s_max[t];
            s_min[t] = s_min[t + d] < s_min[t] ? s_min[t + d] : s_min[t];
            s_nan[t] |= s_nan[t + d];
        }
        __syncthreads();
    }
    const double thr = 0.5 * (s_max[0] + s_min[0]);
    const bool poisoned = s_nan[0] != 0;                                // nan threshold: nothing compares greater
    for (int i = t; i < L; i += kThreads) {
        const int len = i < n_bits ? (n_bits - i + L - 1) / L : 0;
        const double m = len ? (double)counts[(size_t)f * L + i] / (double)len : 0.0;
        payload[(size_t)f * L + perm[i]] = (!poisoned && len && m > thr) ? 1 : 0;
    }
}

// Pixels outside the block-aligned region pass through the reference's YUV round trip unchanged.
__global__ void copy_fringe_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int n, int H, int W) {
    const int H8 = (H / 8) * 8, W8 = (W / 8) * 8;
    const size_t per = (size_t)H * W * 3;
    const size_t total = (size_t)n * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t k = i % per;
        const int row = (int)(k / ((size_t)W * 3));
        const int col = (int)((k % ((size_t)W * 3)) / 3);
        if (row >= H8 || col >= W8) out[i] = in[i];
    }
}

// Streaming copy, one 16-byte access per lane, each workgroup a contiguous 16 KiB span.
__global__ __launch_bounds__(kThreads) void copy16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * (kThreads * 4) + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k * kThreads < n16) v[k] = src[base + k * kThreads];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k * kThreads < n16) dst[base + k * kThreads] = v[k];
}

// ------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";
int g_self_experiment = 0;   // tools/upper_bound.py only
int g_fuse_verify = 1;    // ofmk_embed_detect_rgb8: 1 = fused mark+analyze kernel, 0 = separate kernels

// Optional per-launch HIP-event timing (bench.py): events are created by ofmk_timing_enable(),
// recorded on the launch stream around every kernel, and read back by ofmk_timing_collect().
enum { KIND_ANALYZE = 0, KIND_FINALIZE = 1, KIND_MARK = 2, KIND_MARK_FUSED = 3, KIND_SVD = 4, KIND_COUNT = 5 };
struct TimingRec { hipEvent_t a, b; int kind; };
TimingRec *g_trec = nullptr;
int g_trec_cap = 0, g_trec_used = 0;
unsigned g_trec_mask = 0x1F;     // which kernel kinds get bracketed

struct ScopedTiming {      // records the "after" event when it goes out of scope
    hipStream_t s;
    TimingRec *r;
    ScopedTiming(int kind, hipStream_t stream) : s(stream), r(nullptr) {
        if (g_trec && ((g_trec_mask >> kind) & 1u) && g_trec_used < g_trec_cap) {
            r = &g_trec[g_trec_used++];
            r->kind = kind;
            (void)hipEventRecord(r->a, s);
        }
    }
    ~ScopedTiming() { if (r) (void)hipEventRecord(r->b, s); }
};

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return fail(OFMK_E_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

constexpr int kMaxChunk = 65535;   // frames per launch = gridDim.y

struct Workspace {
    float *rec;      // kRec planes of [frames][nblk]
    float *delta;    // [frames][nblk]
    unsigned long long *ysum;    // mean accumulators of the frames analyze() saw
    unsigned long long *ysum2;   // ... of the marked frames (fused mark+verify kernel)
    int frames;      // chunk capacity
    size_t plane;    // frames * nblk
};

size_t per_frame_bytes(int H, int W) {
    const size_t nblk = (size_t)(H / 8) * (W / 8);
    return nblk * (kRec + 1) * sizeof(float) + 2 * kSlots * 8;
}

int carve(void *ws, size_t bytes, int H, int W, int want_frames, Workspace &out) {
    if (!ws) return fail(OFMK_E_ARG, "workspace is null%s");
    if ((uintptr_t)ws % 256) return fail(OFMK_E_ARG, "workspace must be 256-byte aligned%s");
    const size_t per = per_frame_bytes(H, W);
    if (bytes < per + 1024) return fail(OFMK_E_WORKSPACE, "workspace smaller than ofmk_workspace_bytes(1, H, W)%s");
    size_t cap = (bytes - 1024) / per;
    if (want_frames > 0 && (size_t)want_frames < cap) cap = want_frames;
    if (cap > (size_t)kMaxChunk) cap = kMaxChunk;
    const size_t nblk = (size_t)(H / 8) * (W / 8);
    char *p = static_cast<char *>(ws);
    out.frames = (int)cap;
    out.plane = cap * nblk;
    out.rec = reinterpret_cast<float *>(p);
    p += align256(out.plane * kRec * sizeof(float));
    out.delta = reinterpret_cast<float *>(p);
    p += align256(out.plane * sizeof(float));
    out.ysum = reinterpret_cast<unsigned long long *>(p);
    p += align256(cap * kSlots * 8);
    out.ysum2 = reinterpret_cast<unsigned long long *>(p);
    return OFMK_OK;
}

int check_dims(int n, int H, int W) {
    if (n <= 0) return fail(OFMK_E_ARG, "n must be positive%s");
    if (H < 8 || W < 8) return fail(OFMK_E_ARG, "H and W must be at least 8%s");
    if ((long long)H * W >= (1LL << 28)) return fail(OFMK_E_ARG, "frame too large (H*W must be < 2^28)%s");
    return OFMK_OK;
}

Geom make_geom(int H, int W, const Workspace &ws) {
    Geom g;
    g.W = W;
    g.wb = W / 8;
    g.inv_wb = 1.0f / (float)g.wb;
    g.nblk = (H / 8) * (W / 8);
    g.frame_stride = (size_t)H * W * 3;
    g.plane = ws.plane;
    return g;
}

dim3 block_grid(const Geom &g, int n) { return dim3((unsigned)((g.nblk + kThreads - 1) / kThreads), (unsigned)n); }

bool aligned_rows(const void *p, int W, size_t elem) {   // every 8-pixel block row starts on 8 B (u8) / 16 B (f32)
    const size_t need = elem == 1 ? 8 : 16;
    return W % 8 == 0 && (uintptr_t)p % need == 0;
}

int launch_analyze(const void *frames, int src, int n, int H, int W, const Workspace &ws, hipStream_t s) {
    HIP_TRY(hipMemsetAsync(ws.ysum, 0, (size_t)n * kSlots * 8, s));
    const Geom g = make_geom(H, W, ws);
    const dim3 grid = block_grid(g, n);
    const bool al = aligned_rows(frames, W, src == SRC_RGB8 ? 1 : 4);
    ScopedTiming timing(KIND_ANALYZE, s);
    if (src == SRC_RGB8) {
        if (al) hipLaunchKernelGGL((analyze_kernel<SRC_RGB8, true>), grid, dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum);
        else hipLaunchKernelGGL((analyze_kernel<SRC_RGB8, false>), grid, dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum);
    } else {
        if (al) hipLaunchKernelGGL((analyze_kernel<SRC_YUV32F, true>), grid, dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum);
        else hipLaunchKernelGGL((analyze_kernel<SRC_YUV32F, false>), grid, dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum);
    }
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int launch_finalize(FinArgs a, int n, hipStream_t s) {
    const unsigned gx = (unsigned)((a.N + kThreads - 1) / kThreads);
    ScopedTiming timing(KIND_FINALIZE, s);
    hipLaunchKernelGGL(finalize_kernel, dim3(gx, (unsigned)n), dim3(kThreads), 0, s, a);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

// Needs the input frames' records in ws.rec / ws.ysum (launch_analyze).  fused = true also leaves the
// MARKED frames' records in ws.rec and their mean accumulators in ws.ysum2.
int launch_mark_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, const int32_t *wm_row,
                     double alpha, const Workspace &ws, bool fused, hipStream_t s) {
    if (fused) HIP_TRY(hipMemsetAsync(ws.ysum2, 0, (size_t)n * kSlots * 8, s));
    const Geom g = make_geom(H, W, ws);
    const dim3 grid = block_grid(g, n);
    const bool al = aligned_rows(in, W, 1) && aligned_rows(out, W, 1);
    MarkArgs m;
    m.rec = ws.rec;
    m.ysum = ws.ysum;
    m.wm = wm;
    m.wm_row = wm_row;
    m.N = (int)((long long)H * W / 64);
    m.alpha = alpha;
    {
        ScopedTiming timing(fused ? KIND_MARK_FUSED : KIND_MARK, s);
        if (fused && g_self_experiment) {
            hipLaunchKernelGGL((mark_rgb8_kernel<true, true, true>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
        } else if (fused) {
            if (al) hipLaunchKernelGGL((mark_rgb8_kernel<true, true>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
            else hipLaunchKernelGGL((mark_rgb8_kernel<false, true>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
        } else {
            if (al) hipLaunchKernelGGL((mark_rgb8_kernel<true, false>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
            else hipLaunchKernelGGL((mark_rgb8_kernel<false, false>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
        }
    }
    HIP_TRY(hipGetLastError());
    if (in != out && (H % 8 || W % 8)) {
        hipLaunchKernelGGL(copy_fringe_kernel, dim3(512), dim3(256), 0, s, in, out, n, H, W);
        HIP_TRY(hipGetLastError());
    }
    return OFMK_OK;
}

FinArgs fin_base(const Workspace &ws, int H, int W, double alpha) {
    FinArgs a;
    memset(&a, 0, sizeof(a));
    a.rec = ws.rec;
    a.plane = ws.plane;
    a.ysum = ws.ysum;
    a.nblk = (H / 8) * (W / 8);
    a.N = (int)((long long)H * W / 64);
    a.L = 1;
    a.alpha = alpha;
    return a;
}

int finalize_detect(int f0, int cf, int H, int W, int L, double alpha, int32_t *counts, uint8_t *bits,
                    const Workspace &ws, bool after_fused_mark, hipStream_t s) {
    FinArgs a = fin_base(ws, H, W, alpha);
    if (after_fused_mark) a.ysum = ws.ysum2;
    a.L = L;
    a.counts = counts ? counts + (size_t)f0 * L : nullptr;
    a.bits = bits ? bits + (size_t)f0 * a.N : nullptr;
    return launch_finalize(a, cf, s);
}

// analyze + mark for frames [f0, f0+cf); verify = also leave the marked frames' records in the
// workspace (fused kernel), ready for finalize_detect(after_fused_mark = true)
int embed_chunk(const void *in, void *out, int src, int f0, int cf, int H, int W, const uint8_t *wm,
                const int32_t *wm_row, double alpha, const Workspace &ws, bool verify, hipStream_t s) {
    const size_t fs = (size_t)H * W * 3;
    const size_t esz = src == SRC_RGB8 ? 1 : 4;
    const char *pin = static_cast<const char *>(in) + (size_t)f0 * fs * esz;
    char *pout = static_cast<char *>(out) + (size_t)f0 * fs * esz;
    int rc = launch_analyze(pin, src, cf, H, W, ws, s);
    if (rc) return rc;
    const int32_t *rows = wm_row ? wm_row + f0 : nullptr;
    if (src == SRC_RGB8)
        return launch_mark_rgb8(reinterpret_cast<const uint8_t *>(pin), reinterpret_cast<uint8_t *>(pout), cf, H, W, wm,
                                rows, alpha, ws, verify, s);
    // float32 YUV plugin path: separate scalar stage, then the rank-1 update of channel 1
    FinArgs a = fin_base(ws, H, W, alpha);
    a.wm = wm;
    a.wm_row = rows;
    a.delta = ws.delta;
    if ((rc = launch_finalize(a, cf, s))) return rc;
    const Geom g = make_geom(H, W, ws);
    hipLaunchKernelGGL(mark_yuv32f_kernel, block_grid(g, cf), dim3(kThreads), 0, s, reinterpret_cast<float *>(pout), g, ws.delta);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int detect_chunk(const void *in, int src, int f0, int cf, int H, int W, int L, double alpha, int32_t *counts,
                 uint8_t *bits, const Workspace &ws, hipStream_t s) {
    const size_t fs = (size_t)H * W * 3;
    const size_t esz = src == SRC_RGB8 ? 1 : 4;
    const char *pin = static_cast<const char *>(in) + (size_t)f0 * fs * esz;
    int rc = launch_analyze(pin, src, cf, H, W, ws, s);
    if (rc) return rc;
    return finalize_detect(f0, cf, H, W, L, alpha, counts, bits, ws, false, s);
}

// ---- DwtDctSvd codec ---------------------------------------------------------------------------
int launch_svd_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, int mode, SvdArgs a, hipStream_t s) {
    Workspace none;
    none.plane = 0;
    const Geom g = make_geom(H, W, none);
    const bool al = aligned_rows(in, W, 1) && (mode == SVD_DETECT || aligned_rows(out, W, 1));
    if (a.counts) HIP_TRY(hipMemsetAsync(a.counts, 0, (size_t)n * a.L * sizeof(int32_t), s));
    if (a.bits && a.N > g.nblk) HIP_TRY(hipMemsetAsync(a.bits, 0, (size_t)n * a.N, s));   // entries past (H/8)(W/8) stay 0
    for (int f0 = 0; f0 < n; f0 += kMaxChunk) {
        const int cf = n - f0 < kMaxChunk ? n - f0 : kMaxChunk;
        const size_t fo = (size_t)f0 * g.frame_stride;
        SvdArgs b = a;
        if (b.wm_row) b.wm_row += f0;
        if (b.counts) b.counts += (size_t)f0 * a.L;
        if (b.bits) b.bits += (size_t)f0 * a.N;
        const dim3 grid = block_grid(g, cf);
        ScopedTiming timing(KIND_SVD, s);
#define OFMK_SVD_LAUNCH(AL, MD) hipLaunchKernelGGL((svd_rgb8_kernel<AL, MD>), grid, dim3(kThreads), 0, s, in + fo, out ? out + fo : nullptr, g, b)
        if (mode == SVD_DETECT) { if (al) OFMK_SVD_LAUNCH(true, SVD_DETECT); else OFMK_SVD_LAUNCH(false, SVD_DETECT); }
        else if (mode == SVD_EMBED) { if (al) OFMK_SVD_LAUNCH(true, SVD_EMBED); else OFMK_SVD_LAUNCH(false, SVD_EMBED); }
        else { if (al) OFMK_SVD_LAUNCH(true, SVD_EMBED_VERIFY); else OFMK_SVD_LAUNCH(false, SVD_EMBED_VERIFY); }
#undef OFMK_SVD_LAUNCH
    }
    HIP_TRY(hipGetLastError());
    if (mode != SVD_DETECT && in != out && (H % 8 || W % 8)) {
        hipLaunchKernelGGL(copy_fringe_kernel, dim3(512), dim3(256), 0, s, in, out, n, H, W);
        HIP_TRY(hipGetLastError());
    }
    return OFMK_OK;
}

int check_embed_args(const void *in, const void *out, int n, int H, int W, const uint8_t *wm, int n_wm) {
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!in || !out || !wm) return fail(OFMK_E_ARG, "null frame or watermark pointer%s");
    if (n_wm < 1) return fail(OFMK_E_ARG, "n_wm must be >= 1%s");
    return OFMK_OK;
}

int check_detect_args(const void *in, int n, int H, int W, int L, const int32_t *counts, const uint8_t *bits) {
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!in) return fail(OFMK_E_ARG, "null frame pointer%s");
    if (L < 1) return fail(OFMK_E_ARG, "payload length L must be >= 1%s");
    if (!counts && !bits) return fail(OFMK_E_ARG, "both outputs (counts, bits) are null%s");
    return OFMK_OK;
}

}  // namespace

extern "C" {

int ofmk_version(void) { return OFMK_ABI_VERSION; }
const char *ofmk_last_error(void) { return g_err; }

size_t ofmk_workspace_bytes(int frames_in_flight, int H, int W) {
    if (frames_in_flight < 1 || H < 8 || W < 8) return 0;
    return per_frame_bytes(H, W) * (size_t)frames_in_flight + 1024;
}

void ofmk_set_fused_verify(int on) { g_fuse_verify = on ? 1 : 0; g_self_experiment = on == 2 ? 1 : 0; }

int ofmk_embed_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm,
                    const int32_t *wm_row, double alpha, int chunk_frames, void *workspace, size_t workspace_bytes,
                    void *stream) {
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = embed_chunk(in, out, SRC_RGB8, f0, cf, H, W, wm, wm_row, alpha, ws, false, s))) return rc;
    }
    return OFMK_OK;
}

int ofmk_detect_rgb8(const uint8_t *in, int n, int H, int W, int L, double alpha, int32_t *counts, uint8_t *bits,
                     int chunk_frames, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_detect_args(in, n, H, W, L, counts, bits);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (counts) HIP_TRY(hipMemsetAsync(counts, 0, (size_t)n * L * sizeof(int32_t), s));
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = detect_chunk(in, SRC_RGB8, f0, cf, H, W, L, alpha, counts, bits, ws, s))) return rc;
    }
    return OFMK_OK;
}

int ofmk_embed_detect_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm,
                           const int32_t *wm_row, double alpha, int L, int32_t *counts, uint8_t *bits,
                           int chunk_frames, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    if ((rc = check_detect_args(out, n, H, W, L, counts, bits))) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (counts) HIP_TRY(hipMemsetAsync(counts, 0, (size_t)n * L * sizeof(int32_t), s));
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if (g_fuse_verify) {
            if ((rc = embed_chunk(in, out, SRC_RGB8, f0, cf, H, W, wm, wm_row, alpha, ws, true, s))) return rc;
            if ((rc = finalize_detect(f0, cf, H, W, L, alpha, counts, bits, ws, true, s))) return rc;
        } else {
            if ((rc = embed_chunk(in, out, SRC_RGB8, f0, cf, H, W, wm, wm_row, alpha, ws, false, s))) return rc;
            if ((rc = detect_chunk(out, SRC_RGB8, f0, cf, H, W, L, alpha, counts, bits, ws, s))) return rc;
        }
    }
    return OFMK_OK;
}

int ofmk_encode_yuv32f(float *yuv, int n, int H, int W, const uint8_t *wm, int n_wm, const int32_t *wm_row,
                       double alpha, int chunk_frames, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_embed_args(yuv, yuv, n, H, W, wm, n_wm);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = embed_chunk(yuv, yuv, SRC_YUV32F, f0, cf, H, W, wm, wm_row, alpha, ws, false, s))) return rc;
    }
    return OFMK_OK;
}

int ofmk_decode_yuv32f(const float *yuv, int n, int H, int W, int L, double alpha, int32_t *counts, uint8_t *bits,
                       int chunk_frames, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_detect_args(yuv, n, H, W, L, counts, bits);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (counts) HIP_TRY(hipMemsetAsync(counts, 0, (size_t)n * L * sizeof(int32_t), s));
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = detect_chunk(yuv, SRC_YUV32F, f0, cf, H, W, L, alpha, counts, bits, ws, s))) return rc;
    }
    return OFMK_OK;
}

int ofmk_debug_planes(const void *frame, int src_is_yuv32f, int H, int W, double alpha, const uint8_t *wm,
                      float *y_dc, double *lum_mask, double *tex_mask, double *step, float *c21_pre,
                      float *c21_post, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_dims(1, H, W);
    if (rc) return rc;
    if (!frame) return fail(OFMK_E_ARG, "null frame pointer%s");
    if (c21_post && !wm) return fail(OFMK_E_ARG, "c21_post requested without a watermark%s");
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, 1, ws))) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = launch_analyze(frame, src_is_yuv32f ? SRC_YUV32F : SRC_RGB8, 1, H, W, ws, s))) return rc;
    FinArgs a = fin_base(ws, H, W, alpha);
    a.wm = wm;
    a.y_dc = y_dc;
    a.lum = lum_mask;
    a.tex = tex_mask;
    a.step = step;
    a.c21_pre = c21_pre;
    a.c21_post = wm ? c21_post : nullptr;
    return launch_finalize(a, 1, s);
}

int ofmk_stage_analyze_rgb8(const uint8_t *in, int n, int H, int W, void *workspace, size_t workspace_bytes,
                            void *stream) {
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!in) return fail(OFMK_E_ARG, "null frame pointer%s");
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, n, ws))) return rc;
    if (ws.frames < n) return fail(OFMK_E_WORKSPACE, "stage call needs workspace for all n frames%s");
    return launch_analyze(in, SRC_RGB8, n, H, W, ws, static_cast<hipStream_t>(stream));
}

int ofmk_stage_mark_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, double alpha,
                         int fused, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_embed_args(in, out, n, H, W, wm, 1);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, n, ws))) return rc;
    if (ws.frames < n) return fail(OFMK_E_WORKSPACE, "stage call needs workspace for all n frames%s");
    return launch_mark_rgb8(in, out, n, H, W, wm, nullptr, alpha, ws, fused != 0, static_cast<hipStream_t>(stream));
}

int ofmk_svd_embed_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm,
                        const int32_t *wm_row, double scale, void *stream) {
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    if (!(scale > 0)) return fail(OFMK_E_ARG, "scale must be positive%s");
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    a.wm = wm; a.wm_row = wm_row; a.N = (int)((long long)H * W / 64); a.L = 1; a.scale = (float)scale;
    return launch_svd_rgb8(in, out, n, H, W, SVD_EMBED, a, static_cast<hipStream_t>(stream));
}

int ofmk_svd_detect_rgb8(const uint8_t *in, int n, int H, int W, int L, double scale, int32_t *counts, uint8_t *bits,
                         void *stream) {
    int rc = check_detect_args(in, n, H, W, L, counts, bits);
    if (rc) return rc;
    if (!(scale > 0)) return fail(OFMK_E_ARG, "scale must be positive%s");
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    a.counts = counts; a.bits = bits; a.N = (int)((long long)H * W / 64); a.L = L; a.scale = (float)scale;
    return launch_svd_rgb8(in, nullptr, n, H, W, SVD_DETECT, a, static_cast<hipStream_t>(stream));
}

int ofmk_svd_embed_detect_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm,
                               const int32_t *wm_row, double scale, int L, int32_t *counts, uint8_t *bits,
                               void *stream) {
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    if ((rc = check_detect_args(out, n, H, W, L, counts, bits))) return rc;
    if (!(scale > 0)) return fail(OFMK_E_ARG, "scale must be positive%s");
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    a.wm = wm; a.wm_row = wm_row; a.counts = counts; a.bits = bits;
    a.N = (int)((long long)H * W / 64); a.L = L; a.scale = (float)scale;
    return launch_svd_rgb8(in, out, n, H, W, SVD_EMBED_VERIFY, a, static_cast<hipStream_t>(stream));
}

int ofmk_svd_encode_yuv32f(float *yuv, int n, int H, int W, const uint8_t *wm, int n_wm, const int32_t *wm_row,
                           double scale, void *stream) {
    int rc = check_embed_args(yuv, yuv, n, H, W, wm, n_wm);
    if (rc) return rc;
    if (!(scale > 0) || n > kMaxChunk) return fail(OFMK_E_ARG, "bad scale or too many frames%s");
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    a.wm = wm; a.wm_row = wm_row; a.N = (int)((long long)H * W / 64); a.L = 1; a.scale = (float)scale;
    Workspace none;
    none.plane = 0;
    const Geom g = make_geom(H, W, none);
    hipLaunchKernelGGL((svd_yuv32f_kernel<SVD_EMBED>), block_grid(g, n), dim3(kThreads), 0, static_cast<hipStream_t>(stream), yuv, g, a);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int ofmk_svd_decode_yuv32f(const float *yuv, int n, int H, int W, double scale, uint8_t *bits, void *stream) {
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!yuv || !bits) return fail(OFMK_E_ARG, "null pointer%s");
    if (!(scale > 0) || n > kMaxChunk) return fail(OFMK_E_ARG, "bad scale or too many frames%s");
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    a.bits = bits; a.N = (int)((long long)H * W / 64); a.L = 1; a.scale = (float)scale;
    Workspace none;
    none.plane = 0;
    const Geom g = make_geom(H, W, none);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.N > g.nblk) HIP_TRY(hipMemsetAsync(bits, 0, (size_t)n * a.N, s));
    hipLaunchKernelGGL((svd_yuv32f_kernel<SVD_DETECT>), block_grid(g, n), dim3(kThreads), 0, s, const_cast<float *>(yuv), g, a);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int ofmk_payloads_from_counts(const int32_t *counts, int n, int L, int n_bits, const int32_t *perm, uint8_t *payload,
                              void *stream) {
    if (!counts || !perm || !payload) return fail(OFMK_E_ARG, "null pointer%s");
    if (n < 1 || L < 1 || n_bits < 0) return fail(OFMK_E_ARG, "bad sizes%s");
    hipLaunchKernelGGL(degenerate_kernel, dim3((unsigned)n), dim3(kThreads), 0, static_cast<hipStream_t>(stream), counts,
                       L, n_bits, perm, payload);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int ofmk_timing_enable(int max_launches, unsigned kind_mask) {
    if (g_trec) return fail(OFMK_E_ARG, "timing already enabled%s");
    g_trec_mask = kind_mask ? kind_mask : 0x1F;
    if (max_launches < 1) return fail(OFMK_E_ARG, "max_launches must be positive%s");
    g_trec = new TimingRec[max_launches];
    for (int i = 0; i < max_launches; ++i) {
        HIP_TRY(hipEventCreate(&g_trec[i].a));
        HIP_TRY(hipEventCreate(&g_trec[i].b));
    }
    g_trec_cap = max_launches;
    g_trec_used = 0;
    return OFMK_OK;
}

int ofmk_timing_collect(double *ms_by_kind, int *launches_by_kind) {
    if (!g_trec) return fail(OFMK_E_ARG, "timing is not enabled%s");
    if (!ms_by_kind || !launches_by_kind) return fail(OFMK_E_ARG, "null output%s");
    for (int k = 0; k < KIND_COUNT; ++k) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
    for (int i = 0; i < g_trec_used; ++i) {
        HIP_TRY(hipEventSynchronize(g_trec[i].b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_trec[i].a, g_trec[i].b));
        ms_by_kind[g_trec[i].kind] += ms;
        launches_by_kind[g_trec[i].kind] += 1;
    }
    g_trec_used = 0;
    return OFMK_OK;
}

void ofmk_timing_disable(void) {
    if (!g_trec) return;
    for (int i = 0; i < g_trec_cap; ++i) { (void)hipEventDestroy(g_trec[i].a); (void)hipEventDestroy(g_trec[i].b); }
    delete[] g_trec;
    g_trec = nullptr;
    g_trec_cap = g_trec_used = 0;
}

int ofmk_hbm_copy(const void *src, void *dst, size_t bytes, void *stream) {
    if (!src || !dst || bytes % 16 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
        return fail(OFMK_E_ARG, "copy needs 16-byte aligned pointers and size%s");
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)((n16 + kThreads * 4 - 1) / (kThreads * 4))), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

}  // extern "C"
