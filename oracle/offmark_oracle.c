/* offmark_oracle.c -- plain-C restatement of the offmark DCT watermark path.  TEST INFRASTRUCTURE ONLY.
 *
 * Same role and same status as oracle/offmark_oracle.py (read its header): the checker for the HIP kernels
 * and the CPU baseline of bench.py, never the product path.  It performs, operation for operation, the
 * arithmetic of the NumPy oracle's vectorised form, so the two agree BIT FOR BIT (tests/test_oracle_c.py);
 * the NumPy oracle in turn reproduces the vectors captured from the reference's own modules.  OpenCV's float
 * arithmetic is restated, not linked: parity unpinned for that part, as documented there.
 *
 * Reference files restated (relative to the reference root):
 *   src/offmark/video/embedder.py:33-39, src/offmark/video/extractor.py:30-34,
 *   src/offmark/embed/dct_encoder.py:18-102, src/offmark/extract/dct_decoder.py:10-27.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double CK[8];           /* cos(k*pi/16) */
static double S0;              /* sqrt(1/8)    */
static int g_init = 0;

static void init_tables(void) {
    if (g_init) return;
    for (int k = 0; k < 8; ++k) CK[k] = cos(k * M_PI / 16.0);
    S0 = sqrt(0.125);
    g_init = 1;
}

/* float32 fma as the NumPy oracle emulates it: exact product in double, one double add, round to float */
static inline float fma32(float a, float b, float c) { return (float)((double)a * (double)b + (double)c); }

static const float CY0 = 0.114f, CY1 = 0.587f, CY2 = 0.299f, CU = 0.492f, CV = 0.877f, DELTA = 0.5f;
static const float I_B = 2.032f, I_GU = -0.395f, I_GV = -0.581f, I_R = 1.140f;

/* orthonormal 8-point DCT-II, even/odd butterflies, float64 (oracle _dct1d_last) */
static void dct1d(const double *x, int stride, double *o, int ostride) {
    const double x0 = x[0], x1 = x[stride], x2 = x[2 * stride], x3 = x[3 * stride], x4 = x[4 * stride],
                 x5 = x[5 * stride], x6 = x[6 * stride], x7 = x[7 * stride];
    const double a0 = x0 + x7, a1 = x1 + x6, a2 = x2 + x5, a3 = x3 + x4;
    const double b0 = x0 - x7, b1 = x1 - x6, b2 = x2 - x5, b3 = x3 - x4;
    const double e0 = a0 + a3, e1 = a1 + a2, e2 = a0 - a3, e3 = a1 - a2;
    o[0] = (e0 + e1) * S0;
    o[4 * ostride] = (e0 - e1) * (0.5 * CK[4]);
    o[2 * ostride] = 0.5 * (e2 * CK[2] + e3 * CK[6]);
    o[6 * ostride] = 0.5 * (e2 * CK[6] - e3 * CK[2]);
    o[1 * ostride] = 0.5 * (((b0 * CK[1] + b1 * CK[3]) + b2 * CK[5]) + b3 * CK[7]);
    o[3 * ostride] = 0.5 * (((b0 * CK[3] - b1 * CK[7]) - b2 * CK[1]) - b3 * CK[5]);
    o[5 * ostride] = 0.5 * (((b0 * CK[5] - b1 * CK[1]) + b2 * CK[7]) + b3 * CK[3]);
    o[7 * ostride] = 0.5 * (((b0 * CK[7] - b1 * CK[5]) + b2 * CK[3]) - b3 * CK[1]);
}

/* inverse (oracle _idct1d_last) */
static void idct1d(const double *X, int stride, double *o, int ostride) {
    const double X0 = X[0], X1 = X[stride], X2 = X[2 * stride], X3 = X[3 * stride], X4 = X[4 * stride],
                 X5 = X[5 * stride], X6 = X[6 * stride], X7 = X[7 * stride];
    const double p0 = X0 * S0 + X4 * (0.5 * CK[4]);
    const double p1 = X0 * S0 - X4 * (0.5 * CK[4]);
    const double q0 = 0.5 * (X2 * CK[2] + X6 * CK[6]);
    const double q1 = 0.5 * (X2 * CK[6] - X6 * CK[2]);
    const double ev0 = p0 + q0, ev3 = p0 - q0, ev1 = p1 + q1, ev2 = p1 - q1;
    const double od0 = 0.5 * (((X1 * CK[1] + X3 * CK[3]) + X5 * CK[5]) + X7 * CK[7]);
    const double od1 = 0.5 * (((X1 * CK[3] - X3 * CK[7]) - X5 * CK[1]) - X7 * CK[5]);
    const double od2 = 0.5 * (((X1 * CK[5] - X3 * CK[1]) + X5 * CK[7]) + X7 * CK[3]);
    const double od3 = 0.5 * (((X1 * CK[7] - X3 * CK[5]) + X5 * CK[3]) - X7 * CK[1]);
    o[0] = ev0 + od0; o[7 * ostride] = ev0 - od0;
    o[1 * ostride] = ev1 + od1; o[6 * ostride] = ev1 - od1;
    o[2 * ostride] = ev2 + od2; o[5 * ostride] = ev2 - od2;
    o[3 * ostride] = ev3 + od3; o[4 * ostride] = ev3 - od3;
}

/* 2-D transform of one 8x8 float32 block (row stride `pitch` floats): last axis first, then the other,
 * float64 throughout, one rounding to float32 at the end (oracle dct8x8 / idct8x8) */
static void block_transform(const float *blk, int pitch, int inverse, float out[64]) {
    double x[64], t[64], c[64];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) x[i * 8 + j] = (double)blk[(size_t)i * pitch + j];
    for (int i = 0; i < 8; ++i) (inverse ? idct1d : dct1d)(x + i * 8, 1, t + i * 8, 1);
    for (int j = 0; j < 8; ++j) (inverse ? idct1d : dct1d)(t + j, 8, c + j, 8);
    for (int k = 0; k < 64; ++k) out[k] = (float)c[k];
}

/* numpy's float64 add.reduce over a contiguous array: 8192-element buffer chunks summed in order, each chunk
 * by pairwise summation (8 accumulators, blocks of <= 128 at the leaves) */
static double pairwise_sum(const double *a, size_t n) {
    if (n < 8) {
        double r = 0.0;
        for (size_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        size_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}
static double numpy_sum(const double *a, size_t n) {
    double acc = 0.0;
    int first = 1;
    for (size_t i = 0; i < n; i += 8192) {
        const double p = pairwise_sum(a + i, n - i < 8192 ? n - i : 8192);
        acc = first ? p : acc + p;
        first = 0;
    }
    return acc;
}

static void bgr2yuv(const uint8_t *in, float *yuv, size_t npx) {
    for (size_t p = 0; p < npx; ++p) {
        const float c0 = in[3 * p], c1 = in[3 * p + 1], c2 = in[3 * p + 2];
        const float y = fma32(c0, CY0, fma32(c1, CY1, c2 * CY2));
        yuv[3 * p] = y;
        yuv[3 * p + 1] = fma32(c0 - y, CU, DELTA);
        yuv[3 * p + 2] = fma32(c2 - y, CV, DELTA);
    }
}

static void yuv2bgr_u8(const float *yuv, uint8_t *out, size_t npx) {
    for (size_t p = 0; p < npx; ++p) {
        const float y = yuv[3 * p], u = yuv[3 * p + 1] - DELTA, v = yuv[3 * p + 2] - DELTA;
        float c[3];
        c[0] = fma32(u, I_B, y);
        c[1] = fma32(v, I_GV, fma32(u, I_GU, y));
        c[2] = fma32(v, I_R, y);
        for (int k = 0; k < 3; ++k) {
            float t = c[k] < 0.f ? 0.f : (c[k] > 255.f ? 255.f : c[k]);      /* np.clip */
            out[3 * p + k] = (uint8_t)rintf(t);                              /* np.around (half to even) */
        }
    }
}

static int ge_const(float x, double c, int legacy) { return legacy ? ((double)x >= c) : (x >= (float)c); }

/* texture_mask for one block from |DCT(Y)| (oracle texture_features + texture_from_features) */
static double texture_one(const float a[64], int legacy) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    for (int i = 1; i < 8; ++i)
        for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i * 8 + j];
    const float tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const float dcl = ((((a[0] + a[1]) + a[2]) + a[8]) + a[9]) + a[16];
    const float eh = tot - dcl;
    float e = a[3 * 8];
    static const int ep[11][2] = {{4, 0}, {5, 0}, {6, 0}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {2, 1}, {1, 2}, {2, 2}, {3, 3}};
    for (int k = 0; k < 11; ++k) e = e + a[ep[k][0] * 8 + ep[k][1]];
    const float h = eh - e, l = dcl - a[0];
    const float l_e = l / e, lpe = l + e, le_h = lpe / h, eph = e + h;
    const int gt4 = le_h > 4.f;
    const int c2 = (ge_const(l_e, 1.4, legacy) && ge_const(le_h, 1.1, legacy)) ||
                   (ge_const(l_e, 1.1, legacy) && ge_const(le_h, 1.4, legacy)) || gt4;
    const int c1 = (ge_const(l_e, 2.3, legacy) && ge_const(le_h, 1.6, legacy)) ||
                   (ge_const(l_e, 1.6, legacy) && ge_const(le_h, 2.3, legacy)) || gt4;
    const double step_val = lpe <= 400.f ? 1.125 : 1.25;
    double ramp;
    if (legacy) ramp = 1 + 1.25 * ((double)eh - 290) / (1800 - 290);
    else {
        float t = eh - 290.f;
        t = 1.25f * t;
        t = t / 1510.f;
        ramp = (double)(1.f + t);
    }
    double out = 1.0;
    const int active = eh > 125.f, big = active && eh > 900.f, small = active && !big;
    if (big && c2) out = step_val;
    if (big && !c2) out = ramp;
    if (small && c1) out = step_val;
    if (small && !c1 && eph > 290.f) out = ramp;
    return out;
}

/* masks for a frame: mask[b] = tex*lum for every block of the Y plane (channel 0 of yuv) */
static void frame_mask(const float *yuv, int H, int W, int legacy, double *mask) {
    const int h8 = H / 8, w8 = W / 8, nb = h8 * w8;
    double *m = (double *)malloc(sizeof(double) * (size_t)nb);
    for (int bi = 0; bi < h8; ++bi)
        for (int bj = 0; bj < w8; ++bj) {
            float blk[64], c[64];
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) blk[i * 8 + j] = yuv[((size_t)(bi * 8 + i) * W + bj * 8 + j) * 3];
            block_transform(blk, 8, 0, c);
            m[bi * w8 + bj] = (double)c[0] / 8;
            for (int k = 0; k < 64; ++k) c[k] = fabsf(c[k]);
            mask[bi * w8 + bj] = texture_one(c, legacy);                /* tex for now */
        }
    double mean = numpy_sum(m, (size_t)nb) / (double)nb;
    if (!(mean > 90.0)) mean = 90.0;                                      /* max(90, mean) */
    const double f_ref = 1 + (mean - 90) * (2 - 1) / (255 - 90);
    for (int b = 0; b < nb; ++b) {
        double lum;
        if (m[b] > mean) lum = 1 + (m[b] - mean) / (255 - mean) * (2 - f_ref);
        else if (m[b] < 15) lum = 1.25;
        else if (m[b] < 25) lum = 1.125;
        else lum = 1.0;
        mask[b] = mask[b] * lum;
    }
    free(m);
}

int ofo_mark_frame(const uint8_t *in, uint8_t *out, int H, int W, const int64_t *wm, double alpha, int legacy) {
    init_tables();
    const size_t npx = (size_t)H * W;
    const int h8 = H / 8, w8 = W / 8;
    float *yuv = (float *)malloc(sizeof(float) * npx * 3);
    double *mask = (double *)malloc(sizeof(double) * (size_t)(h8 * w8 > 0 ? h8 * w8 : 1));
    if (!yuv || !mask) { free(yuv); free(mask); return -1; }
    bgr2yuv(in, yuv, npx);
    frame_mask(yuv, H, W, legacy, mask);
    for (int bi = 0; bi < h8; ++bi)
        for (int bj = 0; bj < w8; ++bj) {
            float blk[64], c[64], back[64];
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) blk[i * 8 + j] = yuv[((size_t)(bi * 8 + i) * W + bj * 8 + j) * 3 + 1];
            block_transform(blk, 8, 0, c);
            const double step = alpha * mask[bi * w8 + bj], step2 = step + step;
            const float c21 = c[2 * 8 + 1];
            double q = floor((double)fabsf(c21) / step2) * step2;
            if (wm[bi * w8 + bj] != 0) q = q + step;
            const float sgn = c21 > 0.f ? 1.f : (c21 < 0.f ? -1.f : 0.f);   /* np.sign */
            c[2 * 8 + 1] = (float)((double)sgn * q);
            block_transform(c, 8, 1, back);
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) yuv[((size_t)(bi * 8 + i) * W + bj * 8 + j) * 3 + 1] = back[i * 8 + j];
        }
    yuv2bgr_u8(yuv, out, npx);
    free(yuv);
    free(mask);
    return 0;
}

/* bits: H*W/64 doubles, the first (H/8)*(W/8) written, the rest zero (dct_decoder.py:16-19) */
int ofo_check_frame(const uint8_t *in, int H, int W, double alpha, int legacy, double *bits) {
    init_tables();
    const size_t npx = (size_t)H * W;
    const int h8 = H / 8, w8 = W / 8;
    float *yuv = (float *)malloc(sizeof(float) * npx * 3);
    double *mask = (double *)malloc(sizeof(double) * (size_t)(h8 * w8 > 0 ? h8 * w8 : 1));
    if (!yuv || !mask) { free(yuv); free(mask); return -1; }
    memset(bits, 0, sizeof(double) * (npx / 64));
    bgr2yuv(in, yuv, npx);
    frame_mask(yuv, H, W, legacy, mask);
    for (int bi = 0; bi < h8; ++bi)
        for (int bj = 0; bj < w8; ++bj) {
            float blk[64], c[64];
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) blk[i * 8 + j] = yuv[((size_t)(bi * 8 + i) * W + bj * 8 + j) * 3 + 1];
            block_transform(blk, 8, 0, c);
            const double x = rint((double)c[2 * 8 + 1] / (alpha * mask[bi * w8 + bj]));   /* np.around */
            bits[bi * w8 + bj] = fmod(fabs(x), 2.0) == 1.0 ? 1.0 : 0.0;
        }
    free(yuv);
    free(mask);
    return 0;
}

/* batches: frames are independent, one OpenMP thread per frame; returns the number of threads used */
int ofo_mark_frames(const uint8_t *in, uint8_t *out, int n, int H, int W, const int64_t *wm, double alpha, int legacy,
                    int threads) {
    const size_t fs = (size_t)H * W * 3;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = threads > 0 ? threads : omp_get_max_threads();
#endif
    init_tables();
#pragma omp parallel for schedule(dynamic)
    for (int f = 0; f < n; ++f) ofo_mark_frame(in + (size_t)f * fs, out + (size_t)f * fs, H, W, wm, alpha, legacy);
    return used;
}

int ofo_check_frames(const uint8_t *in, int n, int H, int W, double alpha, int legacy, double *bits, int threads) {
    const size_t fs = (size_t)H * W * 3, nb = (size_t)H * W / 64;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = threads > 0 ? threads : omp_get_max_threads();
#endif
    init_tables();
#pragma omp parallel for schedule(dynamic)
    for (int f = 0; f < n; ++f) ofo_check_frame(in + (size_t)f * fs, H, W, alpha, legacy, bits + (size_t)f * nb);
    return used;
}
