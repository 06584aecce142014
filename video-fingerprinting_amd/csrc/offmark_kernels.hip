// offmark_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels + C ABI for the offmark
// DCT frame-watermark path.  Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
//
// Reference behaviour restated here (paths relative to the reference root):
//   src/offmark/video/embedder.py:33-39      u8 -> f32 -> "BGR2YUV" -> encode -> "YUV2BGR" -> clip/round/u8
//   src/offmark/embed/dct_encoder.py:18-102  masks from the Y block DCTs, QIM on U coefficient [2][1]
//   src/offmark/extract/dct_decoder.py:10-27 same masks, bit = round(c21/step) odd
//   src/offmark/degenerator/de_shuffler.py:17-18  sums of bits[i::L] (the mean's numerator)
//
// Three kernels (DESIGN.md has the full story):
//   analyze  : frame pixels -> 5 floats per 8x8 block {A00, sum|A|, dcl, e, C21} + a fixed-point
//              sum of the block means (frame-global mean needed by the luminance mask).
//              Shared by embed and detect.  HBM-bound: 3 B/px in, 0.31 B/px out.
//   finalize : one thread per block: luminance/texture masks (float64 like the reference), step,
//              then either the QIM delta of C21 (embed) or the read-out bit + bits[i::L] counts.
//   apply    : frame pixels + delta -> marked pixels.  Because the 8x8 DCT is orthonormal,
//              idct(dct(U) + d*e21) == U + d * outer(c2, c1): a rank-1 update, no DCT needed.
//
// Thread roles in analyze (256 threads = 4 wavefronts of 64, tile = 8 pixel rows x 32 blocks):
//   phase 1: thread (r = t>>5, b = t&31) owns pixel row r of block b: 24 contiguous bytes.  A
//            32-lane half-wave therefore reads 768 contiguous bytes.  Row DCT in registers.
//   LDS    : thread t stores its 8 row-DCT outputs at T[t*8 .. t*8+7] (two ds_write_b128).
//   phase 2: thread (b = t>>3, j = t&7) owns coefficient column j of block b and reads
//            T[i*256 + t], i = 0..7 (eight conflict-free ds_read_b32).  Column DCT in registers,
//            8-lane DPP butterflies for the per-block sums (no LDS, no bpermute).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/offmark_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileBlocks = 32;   // blocks per tile
constexpr int kRec = 5;           // floats per block record
constexpr int kSlots = 32;        // fixed-point mean accumulators per frame (spreads atomics)
constexpr int kHistMax = 2048;    // payload lengths up to this use an LDS histogram in finalize

constexpr int SRC_RGB8 = 0;
constexpr int SRC_YUV32F = 1;

// 0.5*cos(k*pi/16), k = 1..7  (orthonormal 8-point DCT-II basis; k = 4 doubles as the DC scale)
constexpr float H1 = 0.49039264f, H2 = 0.46193977f, H3 = 0.41573481f, H4 = 0.35355339f,
                H5 = 0.27778512f, H6 = 0.19134172f, H7 = 0.09754516f;

// OpenCV float "YUV" constants
constexpr float KY0 = 0.114f, KY1 = 0.587f, KY2 = 0.299f, KU = 0.492f, KV = 0.877f, KDELTA = 0.5f;
constexpr float KI_B = 2.032f, KI_GU = -0.395f, KI_GV = -0.581f;

__constant__ float kC2[8] = {H2, H6, -H6, -H2, -H2, -H6, H6, H2};   // 0.5*cos((2r+1)*2*pi/16)
__device__ __forceinline__ constexpr float c1_of(int x) {               // 0.5*cos((2x+1)*1*pi/16)
    return x == 0 ? H1 : x == 1 ? H3 : x == 2 ? H5 : x == 3 ? H7 : x == 4 ? -H7 : x == 5 ? -H5 : x == 6 ? -H3 : -H1;
}

// In-place orthonormal 8-point DCT-II, even/odd decomposition: 36 VALU ops.
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

// sum_x v[x] * 0.5*cos((2x+1)*pi/16): coefficient 1 of the 8-point DCT only (8 ops)
__device__ __forceinline__ float proj1(const float (&v)[8]) {
    return fmaf(v[3] - v[4], H7, fmaf(v[2] - v[5], H5, fmaf(v[1] - v[6], H3, (v[0] - v[7]) * H1)));
}

// DPP lane exchange inside groups of 8 lanes (wave64: rows of 16 lanes, quads of 4).
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// After this every lane of an aligned 8-lane group holds ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)),
// which is also numpy's combine order for the 8 running sums of np.sum on 64 floats.
__device__ __forceinline__ float sum8(float v) {
    v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp<0x141>(v);   // row_half_mirror
    return v;
}

__device__ __forceinline__ void divmod_small(int c, int d, float inv_d, int &q, int &r) {
    q = (int)((float)c * inv_d);       // c < 2^24 is checked on the host
    r = c - q * d;
    if (r < 0) { q -= 1; r += d; }
    else if (r >= d) { q += 1; r -= d; }
}

// ------------------------------------------------------------------------------------------
// pixel access
// ------------------------------------------------------------------------------------------
struct Px8 { uint32_t w[6]; };   // 8 interleaved u8 RGB pixels = 24 bytes

template <bool ALIGNED>
__device__ __forceinline__ Px8 load_px8(const uint8_t *p) {
    Px8 v;
    if constexpr (ALIGNED) {
        const uint2 *q = reinterpret_cast<const uint2 *>(p);
        const uint2 a = q[0], b = q[1], c = q[2];
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
        uint2 *q = reinterpret_cast<uint2 *>(p);
        q[0] = make_uint2(v.w[0], v.w[1]);
        q[1] = make_uint2(v.w[2], v.w[3]);
        q[2] = make_uint2(v.w[4], v.w[5]);
    } else {
#pragma unroll
        for (int k = 0; k < 24; ++k) p[k] = (uint8_t)(v.w[k >> 2] >> (8 * (k & 3)));
    }
}

__device__ __forceinline__ float px_byte(const Px8 &v, int k) {   // k is a compile-time constant after unrolling
    return (float)((v.w[k >> 2] >> (8 * (k & 3))) & 0xffu);       // -> v_cvt_f32_ubyteN
}

struct Yuv8 { float4 q[6]; };    // 8 interleaved f32 YUV pixels = 96 bytes

template <bool ALIGNED>
__device__ __forceinline__ Yuv8 load_yuv8(const float *p) {
    Yuv8 v;
    if constexpr (ALIGNED) {
        const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll
        for (int k = 0; k < 6; ++k) v.q[k] = q[k];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) v.q[k] = make_float4(p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]);
    }
    return v;
}
__device__ __forceinline__ float yuv_elem(const Yuv8 &v, int k) {
    const float4 &f = v.q[k >> 2];
    return (k & 3) == 0 ? f.x : (k & 3) == 1 ? f.y : (k & 3) == 2 ? f.z : f.w;
}

template <int SRC> struct RawOf;
template <> struct RawOf<SRC_RGB8> { using type = Px8; };
template <> struct RawOf<SRC_YUV32F> { using type = Yuv8; };

// ------------------------------------------------------------------------------------------
// analyze
// ------------------------------------------------------------------------------------------
struct Geom {
    int W;                // pixels per row
    int wb;               // blocks per block-row  (W / 8)
    float inv_wb;
    int nblk;             // (H/8)*(W/8)
    int tiles_per_frame;  // ceil(nblk / 32)
    int total_tiles;      // frames * tiles_per_frame
    int tiles_per_wg;
    size_t frame_stride;  // elements (bytes for u8, floats for f32) between frames
};

template <int SRC, bool ALIGNED>
__device__ __forceinline__ typename RawOf<SRC>::type
load_block_row(const void *frames, const Geom &g, int f, int tb, int r, int b) {
    int c = tb * kTileBlocks + b;
    c = c < g.nblk ? c : g.nblk - 1;              // clamp: the lane computes a duplicate, never stores it
    int bi, bj;
    divmod_small(c, g.wb, g.inv_wb, bi, bj);
    const size_t off = (size_t)f * g.frame_stride + ((size_t)(bi * 8 + r) * g.W + (size_t)bj * 8) * 3;
    if constexpr (SRC == SRC_RGB8) return load_px8<ALIGNED>(static_cast<const uint8_t *>(frames) + off);
    else return load_yuv8<ALIGNED>(static_cast<const float *>(frames) + off);
}

template <int SRC, bool ALIGNED>
__global__ __launch_bounds__(kThreads) void analyze_kernel(const void *__restrict__ frames, Geom g,
                                                           float *__restrict__ rec,
                                                           unsigned long long *__restrict__ ysum) {
    __shared__ __attribute__((aligned(16))) float T[2][kThreads * 8];   // row-DCT outputs, double buffered
    __shared__ float U1[2][kThreads];                                   // per-row k=1 projection of U

    const int t = threadIdx.x;
    const int r1 = t >> 5, b1 = t & 31;   // phase-1 role
    const int b2 = t >> 3, j = t & 7;     // phase-2 role

    // 0/1 lane weights selecting this lane's contribution to dcl and e (texture mask features,
    // dct_encoder.py:81,84-86).  Lane j holds column j: a[i] = |A[i][j]|.
    const float wd0 = j <= 2 ? 1.f : 0.f, wd1 = j <= 1 ? 1.f : 0.f, wd2 = j == 0 ? 1.f : 0.f;
    const float we0 = (j >= 3 && j <= 6) ? 1.f : 0.f;          // (0,3) (0,4) (0,5) (0,6)
    const float we1 = j == 2 ? 1.f : 0.f;                        // (1,2)
    const float we2 = (j == 1 || j == 2) ? 1.f : 0.f;            // (2,1) (2,2)
    const float we3 = (j == 0 || j == 3) ? 1.f : 0.f;            // (3,0) (3,3)
    const float we456 = j == 0 ? 1.f : 0.f;                      // (4,0) (5,0) (6,0)
    const float c2j = kC2[j];

    const int tile0 = blockIdx.x * g.tiles_per_wg;
    int tile_end = tile0 + g.tiles_per_wg;
    tile_end = tile_end < g.total_tiles ? tile_end : g.total_tiles;
    if (tile0 >= tile_end) return;

    // (frame, tile-in-frame) of the current tile and of the prefetched one, advanced incrementally
    int f = tile0 / g.tiles_per_frame, tb = tile0 - f * g.tiles_per_frame;
    int fn = f, tbn = tb;
    long long acc = 0;          // fixed-point (2^-32) sum of block means A00/8, lanes j == 0 only
    int acc_frame = f;

    auto flush = [&](int frame) {
        long long v = acc;
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if ((t & 63) == 0 && v != 0) {
            const int slot = (blockIdx.x * 4 + (t >> 6)) & (kSlots - 1);
            atomicAdd(&ysum[(size_t)frame * kSlots + slot], (unsigned long long)v);
        }
        acc = 0;
    };

    auto raw = load_block_row<SRC, ALIGNED>(frames, g, f, tb, r1, b1);
    int buf = 0;
    for (int tile = tile0; tile < tile_end; ++tile, buf ^= 1) {
        auto cur = raw;
        if (++tbn == g.tiles_per_frame) { tbn = 0; ++fn; }
        if (tile + 1 < tile_end) raw = load_block_row<SRC, ALIGNED>(frames, g, fn, tbn, r1, b1);

        // ---- phase 1: colour transform + row DCT of Y, k=1 projection of U -----------------
        float y[8], u1;
        if constexpr (SRC == SRC_RGB8) {
            // cvtColor BGR2YUV, per pixel and in OpenCV's fma order, so that a chroma-flat block
            // yields bit-identical U samples and hence an exactly zero C21 (np.sign(0) == 0)
            float u[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float c0 = px_byte(cur, 3 * x);
                y[x] = fmaf(c0, KY0, fmaf(px_byte(cur, 3 * x + 1), KY1, px_byte(cur, 3 * x + 2) * KY2));
                u[x] = fmaf(c0 - y[x], KU, KDELTA);
            }
            u1 = proj1(u);
            dct8(y);
        } else {
            float u[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) { y[x] = yuv_elem(cur, 3 * x); u[x] = yuv_elem(cur, 3 * x + 1); }
            u1 = proj1(u);
            dct8(y);
        }
        float4 *dst = reinterpret_cast<float4 *>(&T[buf][t * 8]);
        dst[0] = make_float4(y[0], y[1], y[2], y[3]);
        dst[1] = make_float4(y[4], y[5], y[6], y[7]);
        U1[buf][t] = u1;
        __syncthreads();

        // ---- phase 2: column DCT, per-block features ---------------------------------------
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = T[buf][i * kThreads + t];
        dct8(a);
        const float a00 = a[0];                       // meaningful on lane j == 0
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = fabsf(a[i]);
        float tot = a[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) tot += a[i];      // numpy's running sum r_j
        float dcl = fmaf(wd2, a[2], fmaf(wd1, a[1], wd0 * a[0]));
        float e = fmaf(we456, (a[4] + a[5]) + a[6], fmaf(we3, a[3], fmaf(we2, a[2], fmaf(we1, a[1], we0 * a[0]))));
        float c21 = U1[buf][j * kTileBlocks + b2] * c2j;
        tot = sum8(tot);
        dcl = sum8(dcl);
        e = sum8(e);
        c21 = sum8(c21);

        const int c = tb * kTileBlocks + b2;
        const bool valid = c < g.nblk;
        if (f != acc_frame) { flush(acc_frame); acc_frame = f; }      // wave-uniform
        if (valid) {
            if (j == 0) acc += __float2ll_rn(a00 * 536870912.0f);     // (A00/8) * 2^32
            const float v = j == 0 ? a00 : j == 1 ? tot : j == 2 ? dcl : j == 3 ? e : c21;
            if (j < kRec) rec[((size_t)f * g.nblk + c) * kRec + j] = v;
        }
        f = fn; tb = tbn;
    }
    flush(acc_frame);
}

// ------------------------------------------------------------------------------------------
// finalize
// ------------------------------------------------------------------------------------------
struct FinArgs {
    const float *rec;                 // [frames][nblk][5]
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

// texture_mask, dct_encoder.py:70-102.  float32 arithmetic exactly where the reference's numpy
// scalars are float32; the comparisons with python floats and the ramp are float64 because the
// reference pins numpy 1.23 (legacy promotion: np.float32 scalar (op) python scalar -> float64).
__device__ __forceinline__ double texture_mask(float a00abs, float tot, float dcl, float e) {
    const float eh = tot - dcl;
    double out = 1.0;
    if (eh > 125.f) {
        const float h = eh - e;
        const float l = dcl - a00abs;
        const float l_e = l / e;
        const float lpe = l + e;
        const float le_h = lpe / h;
        const bool big = eh > 900.f;
        const double a = big ? 1.4 : 2.3, b = big ? 1.1 : 1.6;
        const double dl_e = (double)l_e, dle_h = (double)le_h;
        const bool cond = (dl_e >= a && dle_h >= b) || (dl_e >= b && dle_h >= a) || (le_h > 4.f);
        const double ramp = 1.0 + 1.25 * ((double)eh - 290.0) / 1510.0;
        if (cond) out = lpe <= 400.f ? 1.125 : 1.25;
        else if (big) out = ramp;
        else if (e + h > 290.f) out = ramp;
    }
    return out;
}

__global__ __launch_bounds__(kThreads) void finalize_kernel(FinArgs p) {
    __shared__ int hist[kHistMax];
    const int t = threadIdx.x;
    const int f = blockIdx.y;
    const int c = blockIdx.x * kThreads + t;
    const bool use_hist = p.counts != nullptr && p.L <= kHistMax;
    if (use_hist) {
        for (int k = t; k < p.L; k += kThreads) hist[k] = 0;
        __syncthreads();
    }
    if (c < p.nblk) {
        // frame-global mean of the block means (luminance_mask, dct_encoder.py:54-56)
        long long s = 0;
#pragma unroll 8
        for (int k = 0; k < kSlots; ++k) s += (long long)p.ysum[(size_t)f * kSlots + k];
        const double mean_m = ((double)s * (1.0 / 4294967296.0)) / (double)p.nblk;
        const double mean = mean_m > 90.0 ? mean_m : 90.0;
        const double f_ref = 1.0 + (mean - 90.0) * 1.0 / 165.0;

        const float *r = p.rec + ((size_t)f * p.nblk + c) * kRec;
        const float a00 = r[0], tot = r[1], dcl = r[2], e = r[3], c21 = r[4];
        const double m = (double)a00 / 8.0;
        double lum;
        if (m > mean) lum = 1.0 + (m - mean) / (255.0 - mean) * (2.0 - f_ref);
        else if (m < 15.0) lum = 1.25;
        else if (m < 25.0) lum = 1.125;
        else lum = 1.0;
        const double tex = texture_mask(fabsf(a00), tot, dcl, e);
        const double step = p.alpha * (tex * lum);
        const size_t o = (size_t)f * p.nblk + c;
        if (p.y_dc) p.y_dc[o] = a00;
        if (p.lum) p.lum[o] = lum;
        if (p.tex) p.tex[o] = tex;
        if (p.step) p.step[o] = step;
        if (p.c21_pre) p.c21_pre[o] = c21;

        if (p.delta || p.c21_post) {
            // QIM, dct_encoder.py:30-35 (float64 on a float32 coefficient; sign(0) = 0)
            const int row = p.wm_row ? p.wm_row[f] : 0;
            const int bit = p.wm[(size_t)row * p.N + c];
            const double step2 = step + step;
            double q = floor(fabs((double)c21) / step2) * step2;
            if (bit) q = q + step;
            const double nv = c21 > 0.f ? q : (c21 < 0.f ? -q : 0.0);
            const float newc = (float)nv;
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
// apply
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t to_u8(float v) {   // np.clip(0,255) -> np.around -> uint8
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (uint32_t)rintf(v);
}

template <bool ALIGNED>
__global__ __launch_bounds__(kThreads) void apply_rgb8_kernel(const uint8_t *__restrict__ in,
                                                              uint8_t *__restrict__ out, Geom g,
                                                              const float *__restrict__ delta) {
    const int t = threadIdx.x;
    const int r = t >> 5, b = t & 31;
    const float c2r = kC2[r];
    const int tile0 = blockIdx.x * g.tiles_per_wg;
    int tile_end = tile0 + g.tiles_per_wg;
    tile_end = tile_end < g.total_tiles ? tile_end : g.total_tiles;
    int f = tile0 / g.tiles_per_frame, tb = tile0 - f * g.tiles_per_frame - 1;
    for (int tile = tile0; tile < tile_end; ++tile) {
        if (++tb == g.tiles_per_frame) { tb = 0; ++f; }
        const int c = tb * kTileBlocks + b;
        if (c >= g.nblk) continue;
        int bi, bj;
        divmod_small(c, g.wb, g.inv_wb, bi, bj);
        const size_t off = (size_t)f * g.frame_stride + ((size_t)(bi * 8 + r) * g.W + (size_t)bj * 8) * 3;
        const Px8 px = load_px8<ALIGNED>(in + off);
        const float dr = delta[(size_t)f * g.nblk + c] * c2r;
        uint32_t q[24];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const float c0 = px_byte(px, 3 * x), c1 = px_byte(px, 3 * x + 1), c2 = px_byte(px, 3 * x + 2);
            const float y = fmaf(c0, KY0, fmaf(c1, KY1, c2 * KY2));
            const float u = fmaf(c0 - y, KU, KDELTA);          // cvtColor BGR2YUV
            const float v = fmaf(c2 - y, KV, KDELTA);
            const float u2 = fmaf(dr, c1_of(x), u);              // idct(dct(U) + d*e21) = U + d*c2[r]*c1[x]
            const float ud = u2 - KDELTA, vd = v - KDELTA;     // cvtColor YUV2BGR
            q[3 * x] = to_u8(fmaf(ud, KI_B, y));
            q[3 * x + 1] = to_u8(fmaf(vd, KI_GV, fmaf(ud, KI_GU, y)));
            // channel 2 = Y + 1.140*(V-0.5) = c2 - 2.2e-4*(c2 - Y): |error| < 0.05, always rounds back to c2
            q[3 * x + 2] = (px.w[(3 * x + 2) >> 2] >> (8 * ((3 * x + 2) & 3))) & 0xffu;
        }
        Px8 o;
#pragma unroll
        for (int k = 0; k < 6; ++k) o.w[k] = q[4 * k] | (q[4 * k + 1] << 8) | (q[4 * k + 2] << 16) | (q[4 * k + 3] << 24);
        store_px8<ALIGNED>(out + off, o);
    }
}

// DctEncoder.encode on float32 YUV: only channel 1 changes (dct_encoder.py:20,36-37)
__global__ __launch_bounds__(kThreads) void apply_yuv32f_kernel(float *__restrict__ yuv, Geom g,
                                                                const float *__restrict__ delta) {
    const int t = threadIdx.x;
    const int r = t >> 5, b = t & 31;
    const float c2r = kC2[r];
    const int tile0 = blockIdx.x * g.tiles_per_wg;
    int tile_end = tile0 + g.tiles_per_wg;
    tile_end = tile_end < g.total_tiles ? tile_end : g.total_tiles;
    int f = tile0 / g.tiles_per_frame, tb = tile0 - f * g.tiles_per_frame - 1;
    for (int tile = tile0; tile < tile_end; ++tile) {
        if (++tb == g.tiles_per_frame) { tb = 0; ++f; }
        const int c = tb * kTileBlocks + b;
        if (c >= g.nblk) continue;
        int bi, bj;
        divmod_small(c, g.wb, g.inv_wb, bi, bj);
        float *p = yuv + (size_t)f * g.frame_stride + ((size_t)(bi * 8 + r) * g.W + (size_t)bj * 8) * 3;
        const float dr = delta[(size_t)f * g.nblk + c] * c2r;
#pragma unroll
        for (int x = 0; x < 8; ++x) p[3 * x + 1] = fmaf(dr, c1_of(x), p[3 * x + 1]);
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

__global__ __launch_bounds__(kThreads) void copy16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";
int g_analyze_tiles = 8;
int g_apply_tiles = 8;

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

struct Workspace {
    float *rec;
    float *delta;
    unsigned long long *ysum;
    int frames;   // chunk capacity
};

size_t per_frame_bytes(int H, int W) {
    const size_t nblk = (size_t)(H / 8) * (W / 8);
    return align256(nblk * kRec * sizeof(float)) + align256(nblk * sizeof(float)) + align256(kSlots * 8);
}

int carve(void *ws, size_t bytes, int H, int W, int want_frames, Workspace &out) {
    if (!ws) return fail(OFMK_E_ARG, "workspace is null%s");
    if ((uintptr_t)ws % 256) return fail(OFMK_E_ARG, "workspace must be 256-byte aligned%s");
    const size_t per = per_frame_bytes(H, W);
    size_t cap = bytes / per;
    if (cap < 1) return fail(OFMK_E_WORKSPACE, "workspace smaller than ofmk_workspace_bytes(1, H, W)%s");
    if (want_frames > 0 && (size_t)want_frames < cap) cap = want_frames;
    if (cap > (1u << 20)) cap = 1u << 20;
    const size_t nblk = (size_t)(H / 8) * (W / 8);
    char *p = static_cast<char *>(ws);
    out.frames = (int)cap;
    out.rec = reinterpret_cast<float *>(p);
    p += align256(nblk * kRec * sizeof(float)) * cap;
    out.delta = reinterpret_cast<float *>(p);
    p += align256(nblk * sizeof(float)) * cap;
    out.ysum = reinterpret_cast<unsigned long long *>(p);
    return OFMK_OK;
}

int check_dims(int n, int H, int W) {
    if (n <= 0) return fail(OFMK_E_ARG, "n must be positive%s");
    if (H < 8 || W < 8) return fail(OFMK_E_ARG, "H and W must be at least 8%s");
    if ((long long)H * W >= (1LL << 30)) return fail(OFMK_E_ARG, "frame too large (H*W must be < 2^30)%s");
    return OFMK_OK;
}

Geom make_geom(int n, int H, int W, int tiles_per_wg) {
    Geom g;
    g.W = W;
    g.wb = W / 8;
    g.inv_wb = 1.0f / (float)g.wb;
    g.nblk = (H / 8) * (W / 8);
    g.tiles_per_frame = (g.nblk + kTileBlocks - 1) / kTileBlocks;
    g.total_tiles = n * g.tiles_per_frame;
    g.tiles_per_wg = tiles_per_wg;
    g.frame_stride = (size_t)H * W * 3;
    return g;
}

bool aligned_rows(const void *p, int W, size_t elem) {   // every 8-pixel block row starts on 8 B (u8) / 16 B (f32)
    const size_t need = elem == 1 ? 8 : 16;
    return W % 8 == 0 && (uintptr_t)p % need == 0;
}

int launch_analyze(const void *frames, int src, int n, int H, int W, const Workspace &ws, hipStream_t s) {
    HIP_TRY(hipMemsetAsync(ws.ysum, 0, (size_t)n * kSlots * 8, s));
    const Geom g = make_geom(n, H, W, g_analyze_tiles);
    const unsigned grid = (unsigned)((g.total_tiles + g.tiles_per_wg - 1) / g.tiles_per_wg);
    const bool al = aligned_rows(frames, W, src == SRC_RGB8 ? 1 : 4);
    if (src == SRC_RGB8) {
        if (al) hipLaunchKernelGGL((analyze_kernel<SRC_RGB8, true>), dim3(grid), dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum);
        else hipLaunchKernelGGL((analyze_kernel<SRC_RGB8, false>), dim3(grid), dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum);
    } else {
        if (al) hipLaunchKernelGGL((analyze_kernel<SRC_YUV32F, true>), dim3(grid), dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum);
        else hipLaunchKernelGGL((analyze_kernel<SRC_YUV32F, false>), dim3(grid), dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum);
    }
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int launch_finalize(FinArgs a, int n, hipStream_t s) {
    const unsigned gx = (unsigned)((a.N + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(finalize_kernel, dim3(gx, (unsigned)n), dim3(kThreads), 0, s, a);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int launch_apply_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const Workspace &ws, hipStream_t s) {
    const Geom g = make_geom(n, H, W, g_apply_tiles);
    const unsigned grid = (unsigned)((g.total_tiles + g.tiles_per_wg - 1) / g.tiles_per_wg);
    const bool al = aligned_rows(in, W, 1) && aligned_rows(out, W, 1);
    if (al) hipLaunchKernelGGL(apply_rgb8_kernel<true>, dim3(grid), dim3(kThreads), 0, s, in, out, g, ws.delta);
    else hipLaunchKernelGGL(apply_rgb8_kernel<false>, dim3(grid), dim3(kThreads), 0, s, in, out, g, ws.delta);
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
    a.ysum = ws.ysum;
    a.nblk = (H / 8) * (W / 8);
    a.N = (int)((long long)H * W / 64);
    a.L = 1;
    a.alpha = alpha;
    return a;
}

int embed_chunk(const void *in, void *out, int src, int f0, int cf, int H, int W, const uint8_t *wm,
                const int32_t *wm_row, double alpha, const Workspace &ws, hipStream_t s) {
    const size_t fs = (size_t)H * W * 3;
    const size_t esz = src == SRC_RGB8 ? 1 : 4;
    const char *pin = static_cast<const char *>(in) + (size_t)f0 * fs * esz;
    char *pout = static_cast<char *>(out) + (size_t)f0 * fs * esz;
    int rc = launch_analyze(pin, src, cf, H, W, ws, s);
    if (rc) return rc;
    FinArgs a = fin_base(ws, H, W, alpha);
    a.wm = wm;
    a.wm_row = wm_row ? wm_row + f0 : nullptr;
    a.delta = ws.delta;
    rc = launch_finalize(a, cf, s);
    if (rc) return rc;
    if (src == SRC_RGB8)
        return launch_apply_rgb8(reinterpret_cast<const uint8_t *>(pin), reinterpret_cast<uint8_t *>(pout), cf, H, W, ws, s);
    const Geom g = make_geom(cf, H, W, g_apply_tiles);
    const unsigned grid = (unsigned)((g.total_tiles + g.tiles_per_wg - 1) / g.tiles_per_wg);
    hipLaunchKernelGGL(apply_yuv32f_kernel, dim3(grid), dim3(kThreads), 0, s, reinterpret_cast<float *>(pout), g, ws.delta);
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
    FinArgs a = fin_base(ws, H, W, alpha);
    a.L = L;
    a.counts = counts ? counts + (size_t)f0 * L : nullptr;
    a.bits = bits ? bits + (size_t)f0 * a.N : nullptr;
    return launch_finalize(a, cf, s);
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
    return per_frame_bytes(H, W) * (size_t)frames_in_flight;
}

void ofmk_set_tiles_per_workgroup(int analyze_tiles, int apply_tiles) {
    g_analyze_tiles = analyze_tiles > 0 ? analyze_tiles : 8;
    g_apply_tiles = apply_tiles > 0 ? apply_tiles : 8;
}

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
        if ((rc = embed_chunk(in, out, SRC_RGB8, f0, cf, H, W, wm, wm_row, alpha, ws, s))) return rc;
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
        if ((rc = embed_chunk(in, out, SRC_RGB8, f0, cf, H, W, wm, wm_row, alpha, ws, s))) return rc;
        if ((rc = detect_chunk(out, SRC_RGB8, f0, cf, H, W, L, alpha, counts, bits, ws, s))) return rc;
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
        if ((rc = embed_chunk(yuv, yuv, SRC_YUV32F, f0, cf, H, W, wm, wm_row, alpha, ws, s))) return rc;
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

int ofmk_stage_apply_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, void *workspace,
                          size_t workspace_bytes, void *stream) {
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!in || !out) return fail(OFMK_E_ARG, "null frame pointer%s");
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, n, ws))) return rc;
    if (ws.frames < n) return fail(OFMK_E_WORKSPACE, "stage call needs workspace for all n frames%s");
    return launch_apply_rgb8(in, out, n, H, W, ws, static_cast<hipStream_t>(stream));
}

int ofmk_hbm_copy(const void *src, void *dst, size_t bytes, void *stream) {
    if (!src || !dst || bytes % 16 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
        return fail(OFMK_E_ARG, "copy needs 16-byte aligned pointers and size%s");
    hipLaunchKernelGGL(copy16_kernel, dim3(256 * 8), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), bytes / 16);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

}  // extern "C"
