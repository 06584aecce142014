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
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/offmark_hip.h"

#include "common.hiph"
#include "readout.hiph"
#include "dct_kernels.hiph"
#include "svd_kernels.hiph"
#include "svd8_kernels.hiph"
#include "misc_kernels.hiph"
#include "planar_kernels.hiph"

namespace {

using namespace ofmk;

// ------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";     // the ONLY mutable state of the library: the calling thread's last error text

// Optional per-launch HIP-event timing (bench.py).  The event pool is an object the CALLER owns
// (ofmk_timing_create) and passes in ofmk_opts; no library-wide state.
enum { KIND_ANALYZE = 0, KIND_FINALIZE = 1, KIND_MARK = 2, KIND_MARK_FUSED = 3, KIND_SVD = 4, KIND_PLANAR_ANALYZE = 5, KIND_PLANAR_MARK = 6, KIND_COUNT = 7 };
static_assert(KIND_COUNT == OFMK_TIMING_KINDS, "timing kinds");
struct TimingRec { hipEvent_t a, b; int kind; };

}  // namespace

struct ofmk_timing {
    TimingRec *rec;
    int cap;
    std::atomic<int> used;      // launches that took a pair; may exceed cap (then the surplus launches go untimed)
    unsigned mask;              // which kernel kinds get an event pair
};

namespace {

// What a launcher needs besides its operands: the stream and the (optional) event pool of this call.
struct Ctx {
    hipStream_t s;
    ofmk_timing *t;
    unsigned flags;
    int xcds;         // XCDs the XCD-aware tile order assumes (common.hiph: xcd_tile); which order a launch uses: tile_xcds()
};
constexpr int kDefaultXcds = 8;      // MI355X in SPX mode; ofmk_opts.xcds overrides (ofmk_probe_xcc counts the real ones)
Ctx make_ctx(void *stream, const ofmk_opts *o) {
    Ctx c;
    c.s = static_cast<hipStream_t>(stream);
    c.t = o ? o->timing : nullptr;
    c.flags = o ? o->flags : 0u;
    c.xcds = o && o->xcds ? (int)o->xcds : kDefaultXcds;
    if (c.xcds == 1) c.xcds = 0;
    return c;
}

// Tile order of one launch of the frame-writing DCT kernel (include/offmark_hip.h: OFMK_F_LINEAR_TILES / OFMK_F_XCD_TILES).
// Without a flag the library decides by a STATIC rule on the bytes of frames the launch reads: XCD-aware from
// OFMK_XCD_TILES_MIN_BYTES (192 frames of 1080p) up, linear below.  That is what every interleaved A/B since round 3 says
// (profiles/r4_mark_fused_pass.txt, r4_bench_config4.json, r6_mark_ladder.txt): at 300-384 x 1080p per launch the XCD-aware order
// wins by 1.5-7 % or ties, depending on where the driver placed the caller's frames; at 192 frames the two tie; at 48-96 frames
// linear wins by 1-4 %.  Nothing is measured at run time (rounds 4-5 carried a calibration mode in the Python engine; removed).
int tile_xcds(const Ctx &cx, size_t frame_bytes_per_launch) {
    if (cx.flags & OFMK_F_LINEAR_TILES) return 0;
    if (cx.flags & OFMK_F_XCD_TILES) return cx.xcds;
    return frame_bytes_per_launch >= (size_t)OFMK_XCD_TILES_MIN_BYTES ? cx.xcds : 0;
}

struct ScopedTiming {      // reserves an event pair for the launch that follows (none when timing is off)
    TimingRec *r;
    ScopedTiming(int kind, const Ctx &c) : r(nullptr) {
        if (c.t && ((c.t->mask >> kind) & 1u)) {
            const int k = c.t->used.fetch_add(1, std::memory_order_relaxed);
            if (k < c.t->cap) {
                r = &c.t->rec[k];
                r->kind = kind;
            }
        }
    }
};
// A timed launch hands its event pair to the dispatch itself (hipExtLaunchKernelGGL): the events take the
// kernel's own begin/end timestamps and no marker packets go into the stream, so consecutive kernels still
// dispatch back to back (bracketing with hipEventRecord cost 3-7 % of throughput).
#define OFMK_TIMED_LAUNCH(T, KERNEL, GRID, BLOCK, SHMEM, STREAM, ...)                                            \
    do {                                                                                                         \
        if ((T).r) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, (T).r->a, (T).r->b, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, __VA_ARGS__);                                \
    } while (0)

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

// Zero-fill as a KERNEL, never hipMemsetAsync: every compute call must replay correctly from a captured hipGraph, and on ROCm 7.2
// a graph that holds memset node -> kernel nodes -> memset node OF THE SAME BUFFER -> kernel nodes (two steps of a call sequence
// sharing one counts / accumulator buffer) replays with the memsets out of order from the second replay on: measured with
// tools/graph_memset_order.py (payloads right on the first replay, wrong on every later one; distinct buffers or this kernel: always
// right); tests/test_gpu_parity.py::test_several_steps_in_one_graph_replay_like_eager pins it for both codecs.
hipError_t launch_zero(void *p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const size_t vec = bytes / 16 + 1;
    const unsigned grid = (unsigned)(vec < 256 * 2048 ? (vec + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, s, static_cast<uint8_t *>(p), bytes);
    return hipGetLastError();
}

constexpr int kMaxChunk = 65535;   // frames per launch = gridDim.y

struct Workspace {
    float *rec;      // kRec planes of [frames][nblk]
    float *delta;    // [frames][nblk]
    unsigned long long *ysum;    // [frames][tiles] per-tile partial sums of the block DCs of the frames analyze() saw
    unsigned long long *ysum2;   // ... of the marked frames (fused mark+verify kernel)
    int frames;      // chunk capacity
    int tiles;       // workgroups per frame of the frame kernels
    size_t plane;    // frames * nblk
};

constexpr size_t kFixedBytes = 4096;     // alignment slack of the carved arrays

size_t tiles_per_frame(int H, int W) { return ((size_t)(H / 8) * (W / 8) + kThreads - 1) / kThreads; }

size_t per_frame_bytes(int H, int W) {
    const size_t nblk = (size_t)(H / 8) * (W / 8);
    return nblk * (kRec + 1) * sizeof(float) + 2 * tiles_per_frame(H, W) * 8;
}

int carve(void *ws, size_t bytes, int H, int W, int want_frames, Workspace &out) {
    if (!ws) return fail(OFMK_E_ARG, "workspace is null%s");
    if ((uintptr_t)ws % 256) return fail(OFMK_E_ARG, "workspace must be 256-byte aligned%s");
    const size_t per = per_frame_bytes(H, W);
    if (bytes < per + kFixedBytes) return fail(OFMK_E_WORKSPACE, "workspace smaller than ofmk_workspace_bytes(1, H, W)%s");
    size_t cap = (bytes - kFixedBytes) / per;
    if (want_frames > 0 && (size_t)want_frames < cap) cap = want_frames;
    if (cap > (size_t)kMaxChunk) cap = kMaxChunk;
    const size_t nblk = (size_t)(H / 8) * (W / 8);
    char *p = static_cast<char *>(ws);
    out.frames = (int)cap;
    out.tiles = (int)tiles_per_frame(H, W);
    out.plane = cap * nblk;
    out.rec = reinterpret_cast<float *>(p);
    p += align256(out.plane * kRec * sizeof(float));
    out.delta = reinterpret_cast<float *>(p);
    p += align256(out.plane * sizeof(float));
    out.ysum = reinterpret_cast<unsigned long long *>(p);
    p += align256(cap * out.tiles * 8);
    out.ysum2 = reinterpret_cast<unsigned long long *>(p);
    return OFMK_OK;
}

// every entry point that takes ofmk_opts: unknown flag bits or a non-zero reserved word are a caller bug, not a default
int check_opts(const ofmk_opts *o) {
    if (o && ((o->flags & ~(uint32_t)(OFMK_F_SEPARATE_DETECT | OFMK_F_LINEAR_TILES | OFMK_F_XCD_TILES | OFMK_F_PARTIAL_COUNTS)) || o->xcds > 64u))
        return fail(OFMK_E_ARG, "ofmk_opts: unknown flag bits or xcds > 64%s");
    if (o && (o->flags & OFMK_F_LINEAR_TILES) && (o->flags & OFMK_F_XCD_TILES))
        return fail(OFMK_E_ARG, "ofmk_opts: OFMK_F_LINEAR_TILES and OFMK_F_XCD_TILES exclude each other%s");
    return OFMK_OK;
}

int check_dims(int n, int H, int W) {
    if (n <= 0) return fail(OFMK_E_ARG, "n must be positive%s");
    if (H < 8 || W < 8) return fail(OFMK_E_ARG, "H and W must be at least 8%s");
    if ((long long)H * W >= (1LL << 28)) return fail(OFMK_E_ARG, "frame too large (H*W must be < 2^28)%s");
    return OFMK_OK;
}

Geom make_geom(int H, int W, const Workspace &ws, int frames = 0, int xcds = 0) {
    Geom g;
    g.frames = frames;
    g.xcds = xcds;
    g.W = W;
    g.wb = W / 8;
    g.inv_wb = 1.0f / (float)g.wb;
    g.nblk = (H / 8) * (W / 8);
    g.frame_stride = (size_t)H * W * 3;
    g.plane = ws.plane;
    return g;
}

dim3 block_grid(const Geom &g, int n) { return dim3((unsigned)((g.nblk + kThreads - 1) / kThreads), (unsigned)n); }

// Linear grid of the frame kernels (common.hiph: xcd_tile): tiles x frames workgroups, padded to a multiple of the XCD count
// the tile order assumes (xcds <= 1: linear order, no padding); the kernels take the frame count from their geometry argument.
dim3 xcd_grid(int blocks_per_frame, int n, int xcds = 0) {
    const unsigned tiles = (unsigned)((blocks_per_frame + kThreads - 1) / kThreads);
    const unsigned long long G = (unsigned long long)tiles * (unsigned)n;
    const unsigned long long X = xcds > 1 ? (unsigned long long)xcds : 1ull;
    return dim3((unsigned)(((G + X - 1) / X) * X));
}

bool aligned_rows(const void *p, int W, size_t elem) {   // every 8-pixel block row starts on 8 B (u8) / 16 B (f32)
    const size_t need = elem == 1 ? 8 : 16;
    return W % 8 == 0 && (uintptr_t)p % need == 0;
}

// in -> out for the pixels outside the first Hc rows x Wc columns (frames in chunks of gridDim.y)
void launch_copy_fringe(const uint8_t *in, uint8_t *out, int n, int H, int W, int Hc, int Wc, hipStream_t s) {
    const size_t fringe = (size_t)(H - Hc) * W * 3 + (size_t)Hc * (W - Wc) * 3;
    if (fringe == 0) return;
    const unsigned gx = (unsigned)((fringe + 256 * 4 - 1) / (256 * 4));
    for (int f0 = 0; f0 < n; f0 += kMaxChunk) {
        const int cf = n - f0 < kMaxChunk ? n - f0 : kMaxChunk;
        const size_t fo = (size_t)f0 * H * W * 3;
        hipLaunchKernelGGL(copy_fringe_kernel, dim3(gx < 1024 ? gx : 1024, (unsigned)cf), dim3(256), 0, s, in + fo, out + fo, H, W, Hc, Wc);
    }
}

// zero_counts (optional): [n][L] position sums of these frames, cleared by the kernel for a finalize that follows
int launch_analyze(const void *frames, int src, int n, int H, int W, const Workspace &ws, const Ctx &cx,
                   int32_t *zero_counts = nullptr, int L = 0) {
    hipStream_t s = cx.s;
    // (no fill dispatch in front: the frame mean is summed from per-tile partial sums that every launch writes in full -- common.hiph: emit_block)
    const Geom g = make_geom(H, W, ws, n);
    const dim3 grid = xcd_grid(g.nblk, n);
    const bool al = aligned_rows(frames, W, src == SRC_RGB8 ? 1 : 4);
    ScopedTiming timing(KIND_ANALYZE, cx);
    if (src == SRC_RGB8) {
        // Occupancy cap by an (unused) dynamic LDS reservation: 48 KiB per workgroup = 3 workgroups per CU instead of the
        // 5 the registers allow.  Measured on the shipped kernel, interleaved in one session (profiles/r2_tuning_sweep.txt):
        // 3 per CU 0.341 ms, 4 or 5 per CU 0.350, 2 per CU 0.367, 1 per CU 0.553 -- fewer resident wavefronts keep the
        // window of frame bytes in flight smaller, which the memory system rewards (the bare read pattern shows the same).
        constexpr unsigned kAnalyzeLdsCap = 48u * 1024u;
        if (al) OFMK_TIMED_LAUNCH(timing, (analyze_kernel<SRC_RGB8, true>), grid, dim3(kThreads), kAnalyzeLdsCap, s, frames, g, ws.rec, ws.ysum, zero_counts, L);
        else OFMK_TIMED_LAUNCH(timing, (analyze_kernel<SRC_RGB8, false>), grid, dim3(kThreads), kAnalyzeLdsCap, s, frames, g, ws.rec, ws.ysum, zero_counts, L);
    } else {
        if (al) OFMK_TIMED_LAUNCH(timing, (analyze_kernel<SRC_YUV32F, true>), grid, dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum, zero_counts, L);
        else OFMK_TIMED_LAUNCH(timing, (analyze_kernel<SRC_YUV32F, false>), grid, dim3(kThreads), 0, s, frames, g, ws.rec, ws.ysum, zero_counts, L);
    }
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int launch_finalize(FinArgs a, int n, const Ctx &cx) {
    hipStream_t s = cx.s;
    const int per_wg = kThreads * kFinItems;
    const unsigned gx = (unsigned)((a.N + per_wg - 1) / per_wg);
    ScopedTiming timing(KIND_FINALIZE, cx);
    const bool full = a.delta || a.soft || a.y_dc || a.lum || a.tex || a.step || a.c21_pre || a.c21_post;
    if (full) OFMK_TIMED_LAUNCH(timing, finalize_kernel<true>, dim3(gx, (unsigned)n), dim3(kThreads), 0, s, a);
    else OFMK_TIMED_LAUNCH(timing, finalize_kernel<false>, dim3(gx, (unsigned)n), dim3(kThreads), 0, s, a);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

// Needs the input frames' records in ws.rec / ws.ysum (launch_analyze).  fused = true also leaves the
// MARKED frames' records in ws.rec and their partial sums in ws.ysum2.
int launch_mark_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm, const int32_t *wm_row,
                     double alpha, const Workspace &ws, bool fused, const Ctx &cx) {
    hipStream_t s = cx.s;
    const int xc = tile_xcds(cx, (size_t)n * H * W * 3);
    const Geom g = make_geom(H, W, ws, n, xc);
    const dim3 grid = xcd_grid(g.nblk, n, xc);
    const bool al = aligned_rows(in, W, 1) && aligned_rows(out, W, 1);
    MarkArgs m;
    m.rec = ws.rec;
    m.ysum = ws.ysum;
    m.wm = wm;
    m.wm_row = wm_row;
    m.n_wm = n_wm;
    m.N = (int)((long long)H * W / 64);
    m.alpha = alpha;
    {
        ScopedTiming timing(fused ? KIND_MARK_FUSED : KIND_MARK, cx);
        if (fused) {
            if (al) OFMK_TIMED_LAUNCH(timing, (mark_rgb8_kernel<true, true>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
            else OFMK_TIMED_LAUNCH(timing, (mark_rgb8_kernel<false, true>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
        } else if (xc > 1) {      // XCD-aware order: the marked rows are stored back to back at the end (dct_kernels.hiph: mark_rows HOLD; -2.5 %)
            if (al) OFMK_TIMED_LAUNCH(timing, (mark_rgb8_kernel<true, false, true>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
            else OFMK_TIMED_LAUNCH(timing, (mark_rgb8_kernel<false, false, true>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
        } else {
            if (al) OFMK_TIMED_LAUNCH(timing, (mark_rgb8_kernel<true, false>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
            else OFMK_TIMED_LAUNCH(timing, (mark_rgb8_kernel<false, false>), grid, dim3(kThreads), 0, s, in, out, g, m, ws.rec, ws.ysum2);
        }
    }
    HIP_TRY(hipGetLastError());
    if (in != out && (H % 8 || W % 8)) {
        launch_copy_fringe(in, out, n, H, W, (H / 8) * 8, (W / 8) * 8, s);
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
    a.tiles = ws.tiles;
    a.nblk = (H / 8) * (W / 8);
    a.N = (int)((long long)H * W / 64);
    a.L = 1;
    a.n_wm = 1;
    a.alpha = alpha;
    return a;
}

int finalize_detect(int f0, int cf, int H, int W, int L, double alpha, int32_t *counts, uint8_t *bits,
                    const Workspace &ws, bool after_fused_mark, const Ctx &s) {
    FinArgs a = fin_base(ws, H, W, alpha);
    if (after_fused_mark) a.ysum = ws.ysum2;
    a.L = L;
    a.counts = counts ? counts + (size_t)f0 * L : nullptr;
    a.bits = bits ? bits + (size_t)f0 * a.N : nullptr;
    return launch_finalize(a, cf, s);
}

// analyze + mark for frames [f0, f0+cf); verify = also leave the marked frames' records in the
// workspace (fused kernel), ready for finalize_detect(after_fused_mark = true)
int embed_chunk(const void *in, void *out, int src, int f0, int cf, int H, int W, const uint8_t *wm, int n_wm,
                const int32_t *wm_row, double alpha, const Workspace &ws, bool verify, const Ctx &s,
                int32_t *zero_counts = nullptr, int L = 0) {
    const size_t fs = (size_t)H * W * 3;
    const size_t esz = src == SRC_RGB8 ? 1 : 4;
    const char *pin = static_cast<const char *>(in) + (size_t)f0 * fs * esz;
    char *pout = static_cast<char *>(out) + (size_t)f0 * fs * esz;
    int rc = launch_analyze(pin, src, cf, H, W, ws, s, zero_counts ? zero_counts + (size_t)f0 * L : nullptr, L);
    if (rc) return rc;
    const int32_t *rows = wm_row ? wm_row + f0 : nullptr;
    if (src == SRC_RGB8)
        return launch_mark_rgb8(reinterpret_cast<const uint8_t *>(pin), reinterpret_cast<uint8_t *>(pout), cf, H, W, wm, n_wm,
                                rows, alpha, ws, verify, s);
    // float32 YUV plugin path: separate scalar stage, then the rank-1 update of channel 1
    FinArgs a = fin_base(ws, H, W, alpha);
    a.wm = wm;
    a.wm_row = rows;
    a.n_wm = n_wm;
    a.delta = ws.delta;
    if ((rc = launch_finalize(a, cf, s))) return rc;
    const Geom g = make_geom(H, W, ws);
    hipLaunchKernelGGL(mark_yuv32f_kernel, block_grid(g, cf), dim3(kThreads), 0, s.s, reinterpret_cast<float *>(pout), g, ws.delta);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int launch_payloads(const int32_t *counts, int n, int L, int n_bits, const int32_t *perm, uint8_t *payload, hipStream_t s) {
    hipLaunchKernelGGL(degenerate_kernel, dim3((unsigned)n), dim3(kThreads), 0, s, counts, L, n_bits, perm, payload);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int detect_chunk(const void *in, int src, int f0, int cf, int H, int W, int L, double alpha, int32_t *counts,
                 uint8_t *bits, const Workspace &ws, const Ctx &s) {
    const size_t fs = (size_t)H * W * 3;
    const size_t esz = src == SRC_RGB8 ? 1 : 4;
    const char *pin = static_cast<const char *>(in) + (size_t)f0 * fs * esz;
    int rc = launch_analyze(pin, src, cf, H, W, ws, s, counts ? counts + (size_t)f0 * L : nullptr, L);
    if (rc) return rc;
    return finalize_detect(f0, cf, H, W, L, alpha, counts, bits, ws, false, s);
}

// ---- DwtDctSvd codec ---------------------------------------------------------------------------
int launch_svd_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, int mode, SvdArgs a, const Ctx &cx) {
    hipStream_t s = cx.s;
    Workspace none;
    none.plane = 0;
    const Geom g = make_geom(H, W, none);
    const bool al = aligned_rows(in, W, 1) && (mode == SVD_DETECT || aligned_rows(out, W, 1));
    const size_t tiles = (size_t)(g.nblk + kThreads - 1) / kThreads;
    // partial counts: every workgroup stores its own row, nothing to clear (OFMK_F_PARTIAL_COUNTS); else the frames' [L] sums are added into
    if (a.counts && !a.partial) HIP_TRY(launch_zero(a.counts, (size_t)n * a.L * sizeof(int32_t), s));
    if (a.bits && a.N > g.nblk) HIP_TRY(launch_zero(a.bits, (size_t)n * a.N, s));   // entries past (H/8)(W/8) stay 0
    for (int f0 = 0; f0 < n; f0 += kMaxChunk) {
        const int cf = n - f0 < kMaxChunk ? n - f0 : kMaxChunk;
        const size_t fo = (size_t)f0 * g.frame_stride;
        SvdArgs b = a;
        if (b.wm_row) b.wm_row += f0;
        if (b.counts) b.counts += (size_t)f0 * a.L * (a.partial ? tiles : 1);
        if (b.bits) b.bits += (size_t)f0 * a.N;
        Geom gc = g;
        gc.frames = cf;
        const dim3 grid = xcd_grid(g.nblk, cf);
        ScopedTiming timing(KIND_SVD, cx);
#define OFMK_SVD_LAUNCH(AL, MD, MU) OFMK_TIMED_LAUNCH(timing, (svd_rgb8_kernel<AL, MD, MU>), grid, dim3(kThreads), 0, s, in + fo, out ? out + fo : nullptr, gc, b)
        const bool multi = a.scales[0] > 0.f || a.scales[2] > 0.f || !(a.scales[1] > 0.f);      // anything but the default [0, s, 0]
        if (mode == SVD_DETECT) { if (al) OFMK_SVD_LAUNCH(true, SVD_DETECT, false); else OFMK_SVD_LAUNCH(false, SVD_DETECT, false); }
        else if (mode == SVD_EMBED && !multi) { if (al) OFMK_SVD_LAUNCH(true, SVD_EMBED, false); else OFMK_SVD_LAUNCH(false, SVD_EMBED, false); }
        else if (mode == SVD_EMBED) { if (al) OFMK_SVD_LAUNCH(true, SVD_EMBED, true); else OFMK_SVD_LAUNCH(false, SVD_EMBED, true); }
        else if (!multi) { if (al) OFMK_SVD_LAUNCH(true, SVD_EMBED_VERIFY, false); else OFMK_SVD_LAUNCH(false, SVD_EMBED_VERIFY, false); }
        else { if (al) OFMK_SVD_LAUNCH(true, SVD_EMBED_VERIFY, true); else OFMK_SVD_LAUNCH(false, SVD_EMBED_VERIFY, true); }
#undef OFMK_SVD_LAUNCH
    }
    HIP_TRY(hipGetLastError());
    if (mode != SVD_DETECT && in != out && (H % 8 || W % 8)) {
        launch_copy_fringe(in, out, n, H, W, (H / 8) * 8, (W / 8) * 8, s);
        HIP_TRY(hipGetLastError());
    }
    return OFMK_OK;
}

// blk = 8: 16x16 pixel tiles (svd8_kernels.hiph).  a.N = H*W/64 (watermark row stride), a.N8 = H*W/256 (bits per frame).
Geom8 make_geom8(int H, int W) {
    Geom8 g;
    g.W = W;
    g.wt = ((W / 4) * 2) / 8;
    g.inv_wt = g.wt ? 1.0f / (float)g.wt : 0.f;
    g.ntile = (((H / 4) * 2) / 8) * g.wt;
    g.frames = 0;
    g.frame_stride = (size_t)H * W * 3;
    return g;
}

int launch_svd8_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, int mode, Svd8Args a, const Ctx &cx) {
    hipStream_t s = cx.s;
    const Geom8 g = make_geom8(H, W);
    const size_t tiles = (size_t)(g.ntile + kThreads - 1) / kThreads;
    if (a.counts && !a.partial) HIP_TRY(launch_zero(a.counts, (size_t)n * a.L * sizeof(int32_t), s));
    if (a.bits && a.N8 > g.ntile) HIP_TRY(launch_zero(a.bits, (size_t)n * a.N8, s));    // entries past the tiles stay 0
    const int Hc = (((H / 4) * 2) / 8) * 16, Wc = g.wt * 16;                                    // the region the tiles cover
    if (g.ntile > 0) {
        const bool al = W % 8 == 0 && (uintptr_t)in % 8 == 0 && (mode == SVD_DETECT || (uintptr_t)out % 8 == 0);
        for (int f0 = 0; f0 < n; f0 += kMaxChunk) {
            const int cf = n - f0 < kMaxChunk ? n - f0 : kMaxChunk;
            const size_t fo = (size_t)f0 * g.frame_stride;
            Svd8Args b = a;
            if (b.wm_row) b.wm_row += f0;
            if (b.counts) b.counts += (size_t)f0 * a.L * (a.partial ? tiles : 1);
            if (b.bits) b.bits += (size_t)f0 * a.N8;
            Geom8 gc = g;
            gc.frames = cf;
            const dim3 grid = xcd_grid(g.ntile, cf);
            ScopedTiming timing(KIND_SVD, cx);
#define OFMK_SVD8_LAUNCH(AL, MD, MU) OFMK_TIMED_LAUNCH(timing, (svd8_rgb8_kernel<AL, MD, MU>), grid, dim3(kThreads), 0, s, in + fo, out ? out + fo : nullptr, gc, b)
            const bool multi = a.scales[0] > 0.f || a.scales[2] > 0.f || !(a.scales[1] > 0.f);      // anything but the default [0, s, 0]
            if (mode == SVD_DETECT) { if (al) OFMK_SVD8_LAUNCH(true, SVD_DETECT, false); else OFMK_SVD8_LAUNCH(false, SVD_DETECT, false); }
            else if (mode == SVD_EMBED && !multi) { if (al) OFMK_SVD8_LAUNCH(true, SVD_EMBED, false); else OFMK_SVD8_LAUNCH(false, SVD_EMBED, false); }
            else if (mode == SVD_EMBED) { if (al) OFMK_SVD8_LAUNCH(true, SVD_EMBED, true); else OFMK_SVD8_LAUNCH(false, SVD_EMBED, true); }
            else if (!multi) { if (al) OFMK_SVD8_LAUNCH(true, SVD_EMBED_VERIFY, false); else OFMK_SVD8_LAUNCH(false, SVD_EMBED_VERIFY, false); }
            else { if (al) OFMK_SVD8_LAUNCH(true, SVD_EMBED_VERIFY, true); else OFMK_SVD8_LAUNCH(false, SVD_EMBED_VERIFY, true); }
#undef OFMK_SVD8_LAUNCH
        }
        HIP_TRY(hipGetLastError());
    }
    if (mode != SVD_DETECT && in != out && (Hc != H || Wc != W)) {
        launch_copy_fringe(in, out, n, H, W, Hc, Wc, s);
        HIP_TRY(hipGetLastError());
    }
    return OFMK_OK;
}

// workgroups per frame of the DwtDctSvd frame kernels = rows of a frame's partial counts (OFMK_F_PARTIAL_COUNTS)
int svd_count_tiles(int H, int W, int blk) {
    const long long units = blk == 8 ? (long long)make_geom8(H, W).ntile : (long long)(H / 8) * (W / 8);
    return (int)((units + kThreads - 1) / kThreads);
}

// OFMK_F_PARTIAL_COUNTS on a DwtDctSvd read-out: the workgroup's sums live in its LDS histogram, so L is bounded by it
int check_partial(const ofmk_opts *o, int L, const int32_t *counts, int &partial) {
    partial = (o && (o->flags & OFMK_F_PARTIAL_COUNTS)) ? 1 : 0;
    if (partial && !counts) return fail(OFMK_E_ARG, "OFMK_F_PARTIAL_COUNTS without a counts buffer%s");
    if (partial && L > kHistMax) return fail(OFMK_E_ARG, "OFMK_F_PARTIAL_COUNTS needs L <= 2048 (longer payloads: plain counts)%s");
    return OFMK_OK;
}

int check_blk(int blk) {
    if (blk != 4 && blk != 8)
        return fail(OFMK_E_ARG, "blk must be 4 (the reference's default) or 8%s");
    return OFMK_OK;
}

Svd8Args to_args8(const SvdArgs &a, int H, int W) {
    Svd8Args b;
    memset(&b, 0, sizeof(b));
    b.wm = a.wm; b.wm_row = a.wm_row; b.n_wm = a.n_wm; b.counts = a.counts; b.bits = a.bits; b.partial = a.partial;
    b.N = a.N; b.N8 = (int)((long long)H * W / 256); b.L = a.L;
    for (int k = 0; k < 3; ++k) b.scales[k] = a.scales[k];
    return b;
}

// ---- planar YUV 4:2:0 (I420 / NV12) -------------------------------------------------------------
int check_planar(int layout, int H, int W, const void *a, const void *b) {
    if (layout != OFMK_YUV_I420 && layout != OFMK_YUV_NV12) return fail(OFMK_E_ARG, "layout must be OFMK_YUV_I420 or OFMK_YUV_NV12%s");
    if (H % 8 || W % 8) return fail(OFMK_E_ARG, "planar 4:2:0 entry points need H and W to be multiples of 8%s");
    if ((a && (uintptr_t)a % 8) || (b && (uintptr_t)b % 8)) return fail(OFMK_E_ARG, "planar frame buffers must be 8-byte aligned%s");
    return OFMK_OK;
}

PGeom make_pgeom(int layout, int H, int W, size_t plane) {
    PGeom g;
    g.W = W;
    g.wb = W / 8;
    g.inv_wb = 1.0f / (float)g.wb;
    g.nblk = (H / 8) * (W / 8);
    g.frame_stride = (size_t)H * W * 3 / 2;
    g.u_off = (size_t)H * W;
    g.v_off = layout == OFMK_YUV_I420 ? g.u_off + (size_t)H * W / 4 : g.u_off + 1;
    g.cpitch = layout == OFMK_YUV_I420 ? W / 2 : W;
    g.plane = plane;
    return g;
}

int launch_analyze_yuv420(const uint8_t *frames, int layout, int n, int H, int W, const Workspace &ws, const Ctx &cx,
                          int32_t *zero_counts = nullptr, int L = 0) {
    const PGeom g = make_pgeom(layout, H, W, ws.plane);
    const dim3 grid((unsigned)((g.nblk + kThreads - 1) / kThreads), (unsigned)n);     // 2-D grid: the planar kernels gain nothing from the XCD order
    ScopedTiming timing(KIND_PLANAR_ANALYZE, cx);
    if (layout == OFMK_YUV_I420) OFMK_TIMED_LAUNCH(timing, analyze_yuv420_kernel<FMT_I420>, grid, dim3(kThreads), 0, cx.s, frames, g, ws.rec, ws.ysum, zero_counts, L);
    else OFMK_TIMED_LAUNCH(timing, analyze_yuv420_kernel<FMT_NV12>, grid, dim3(kThreads), 0, cx.s, frames, g, ws.rec, ws.ysum, zero_counts, L);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int launch_mark_yuv420(const uint8_t *in, uint8_t *out, int layout, int n, int H, int W, const uint8_t *wm, int n_wm, const int32_t *wm_row,
                       double alpha, const Workspace &ws, bool fused, const Ctx &cx) {
    const PGeom g = make_pgeom(layout, H, W, ws.plane);
    const dim3 grid((unsigned)((g.nblk + kThreads - 1) / kThreads), (unsigned)n);
    MarkArgs m;
    m.rec = ws.rec;
    m.ysum = ws.ysum;
    m.wm = wm;
    m.wm_row = wm_row;
    m.n_wm = n_wm;
    m.N = (int)((long long)H * W / 64);
    m.alpha = alpha;
    ScopedTiming timing(KIND_PLANAR_MARK, cx);
    if (layout == OFMK_YUV_I420) {
        if (fused) OFMK_TIMED_LAUNCH(timing, (mark_yuv420_kernel<FMT_I420, true>), grid, dim3(kThreads), 0, cx.s, in, out, g, m, ws.rec, ws.ysum2);
        else OFMK_TIMED_LAUNCH(timing, (mark_yuv420_kernel<FMT_I420, false>), grid, dim3(kThreads), 0, cx.s, in, out, g, m, ws.rec, ws.ysum2);
    } else {
        if (fused) OFMK_TIMED_LAUNCH(timing, (mark_yuv420_kernel<FMT_NV12, true>), grid, dim3(kThreads), 0, cx.s, in, out, g, m, ws.rec, ws.ysum2);
        else OFMK_TIMED_LAUNCH(timing, (mark_yuv420_kernel<FMT_NV12, false>), grid, dim3(kThreads), 0, cx.s, in, out, g, m, ws.rec, ws.ysum2);
    }
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

// scales: host double[3], one per YUV channel (dwt_dct_svd_encoder.py:6: scales=[0,15,0]); <= 0 leaves a channel alone
int set_scales(SvdArgs &a, const double *scales, bool need_channel1) {
    if (!scales) return fail(OFMK_E_ARG, "scales is null (host array of 3 doubles)%s");
    bool any = false;
    for (int k = 0; k < 3; ++k) {
        if (!(scales[k] == scales[k]) || scales[k] > 1e30 || scales[k] < -1e30) return fail(OFMK_E_ARG, "scales must be finite%s");
        // decided on the float32 value the kernels use: a positive scale that is no normal float32 >= 1e-3 would mark
        // nothing (flushed to 0) or feed denormals to the quotient / remainder arithmetic
        const float sc = scales[k] > 0 ? (float)scales[k] : 0.f;
        if (scales[k] > 0 && !(sc >= 1.0e-3f)) return fail(OFMK_E_ARG, "a positive scale must be >= 1e-3 as float32%s");
        a.scales[k] = sc;
        any |= sc > 0.f;
    }
    if (!any && !need_channel1) return fail(OFMK_E_ARG, "no channel has a positive scale%s");
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
    return per_frame_bytes(H, W) * (size_t)frames_in_flight + kFixedBytes;
}

int ofmk_embed_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm,
                    const int32_t *wm_row, double alpha, int chunk_frames, void *workspace, size_t workspace_bytes,
                    void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = embed_chunk(in, out, SRC_RGB8, f0, cf, H, W, wm, n_wm, wm_row, alpha, ws, false, cx))) return rc;
    }
    return OFMK_OK;
}

int ofmk_detect_rgb8(const uint8_t *in, int n, int H, int W, int L, double alpha, int32_t *counts, uint8_t *bits,
                     int chunk_frames, void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_detect_args(in, n, H, W, L, counts, bits);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    for (int f0 = 0; f0 < n; f0 += ws.frames) {      // counts are cleared by each chunk's analyze kernel
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = detect_chunk(in, SRC_RGB8, f0, cf, H, W, L, alpha, counts, bits, ws, cx))) return rc;
    }
    return OFMK_OK;
}

// Build extension, not reference semantics: per payload position the sum over the frame's blocks
// of -cos(pi * C21/step) in 2^14 fixed point (positive = the position reads as 1).  One histogram per call, so
// it is a separate entry point from the hard-decision counts.
int ofmk_detect_soft_rgb8(const uint8_t *in, int n, int H, int W, int L, double alpha, long long *soft, int chunk_frames,
                          void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!in || !soft) return fail(OFMK_E_ARG, "null pointer%s");
    if (L < 1) return fail(OFMK_E_ARG, "payload length L must be >= 1%s");
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    HIP_TRY(launch_zero(soft, (size_t)n * L * sizeof(long long), cx.s));
    const size_t fs = (size_t)H * W * 3;
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = launch_analyze(in + (size_t)f0 * fs, SRC_RGB8, cf, H, W, ws, cx))) return rc;
        FinArgs a = fin_base(ws, H, W, alpha);
        a.L = L;
        a.soft = soft + (size_t)f0 * L;
        if ((rc = launch_finalize(a, cf, cx))) return rc;
    }
    return OFMK_OK;
}

int ofmk_embed_detect_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm,
                           const int32_t *wm_row, double alpha, int L, int32_t *counts, uint8_t *bits,
                           int chunk_frames, void *workspace, size_t workspace_bytes, void *stream,
                           const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    if ((rc = check_detect_args(out, n, H, W, L, counts, bits))) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if (!(cx.flags & OFMK_F_SEPARATE_DETECT)) {
            if ((rc = embed_chunk(in, out, SRC_RGB8, f0, cf, H, W, wm, n_wm, wm_row, alpha, ws, true, cx, counts, L))) return rc;
            if ((rc = finalize_detect(f0, cf, H, W, L, alpha, counts, bits, ws, true, cx))) return rc;
        } else {
            if ((rc = embed_chunk(in, out, SRC_RGB8, f0, cf, H, W, wm, n_wm, wm_row, alpha, ws, false, cx))) return rc;
            if ((rc = detect_chunk(out, SRC_RGB8, f0, cf, H, W, L, alpha, counts, bits, ws, cx))) return rc;
        }
    }
    return OFMK_OK;
}

int ofmk_encode_yuv32f(float *yuv, int n, int H, int W, const uint8_t *wm, int n_wm, const int32_t *wm_row,
                       double alpha, int chunk_frames, void *workspace, size_t workspace_bytes, void *stream,
                       const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(yuv, yuv, n, H, W, wm, n_wm);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = embed_chunk(yuv, yuv, SRC_YUV32F, f0, cf, H, W, wm, n_wm, wm_row, alpha, ws, false, cx))) return rc;
    }
    return OFMK_OK;
}

int ofmk_decode_yuv32f(const float *yuv, int n, int H, int W, int L, double alpha, int32_t *counts, uint8_t *bits,
                       int chunk_frames, void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_detect_args(yuv, n, H, W, L, counts, bits);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = detect_chunk(yuv, SRC_YUV32F, f0, cf, H, W, L, alpha, counts, bits, ws, cx))) return rc;
    }
    return OFMK_OK;
}

int ofmk_debug_planes(const void *frame, int src_is_yuv32f, int H, int W, double alpha, const uint8_t *wm,
                      float *y_dc, double *lum_mask, double *tex_mask, double *step, float *c21_pre,
                      float *c21_post, void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_dims(1, H, W);
    if (rc) return rc;
    if (!frame) return fail(OFMK_E_ARG, "null frame pointer%s");
    if (c21_post && !wm) return fail(OFMK_E_ARG, "c21_post requested without a watermark%s");
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, 1, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    if ((rc = launch_analyze(frame, src_is_yuv32f ? SRC_YUV32F : SRC_RGB8, 1, H, W, ws, cx))) return rc;
    FinArgs a = fin_base(ws, H, W, alpha);
    a.wm = wm;
    a.y_dc = y_dc;
    a.lum = lum_mask;
    a.tex = tex_mask;
    a.step = step;
    a.c21_pre = c21_pre;
    a.c21_post = wm ? c21_post : nullptr;
    return launch_finalize(a, 1, cx);
}

int ofmk_stage_analyze_rgb8(const uint8_t *in, int n, int H, int W, void *workspace, size_t workspace_bytes,
                            void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!in) return fail(OFMK_E_ARG, "null frame pointer%s");
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, n, ws))) return rc;
    if (ws.frames < n) return fail(OFMK_E_WORKSPACE, "stage call needs workspace for all n frames%s");
    return launch_analyze(in, SRC_RGB8, n, H, W, ws, make_ctx(stream, opts));
}

int ofmk_stage_mark_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, double alpha,
                         int fused, void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(in, out, n, H, W, wm, 1);
    if (rc) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, n, ws))) return rc;
    if (ws.frames < n) return fail(OFMK_E_WORKSPACE, "stage call needs workspace for all n frames%s");
    return launch_mark_rgb8(in, out, n, H, W, wm, 1, nullptr, alpha, ws, fused != 0, make_ctx(stream, opts));
}

int ofmk_svd_embed_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm,
                        const int32_t *wm_row, const double *scales, int blk, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    if ((rc = check_blk(blk))) return rc;
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    if ((rc = set_scales(a, scales, false))) return rc;
    a.wm = wm; a.wm_row = wm_row; a.n_wm = n_wm; a.N = (int)((long long)H * W / 64); a.L = 1;
    if (blk == 8) return launch_svd8_rgb8(in, out, n, H, W, SVD_EMBED, to_args8(a, H, W), make_ctx(stream, opts));
    return launch_svd_rgb8(in, out, n, H, W, SVD_EMBED, a, make_ctx(stream, opts));
}

// scales[1] <= 0: the reference's decoder returns channel 1's (never written) bit array: all zeros
int ofmk_svd_detect_rgb8(const uint8_t *in, int n, int H, int W, int L, const double *scales, int blk, int32_t *counts, uint8_t *bits,
                         void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_detect_args(in, n, H, W, L, counts, bits);
    if (rc) return rc;
    if ((rc = check_blk(blk))) return rc;
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    if ((rc = set_scales(a, scales, true))) return rc;
    if ((rc = check_partial(opts, L, counts, a.partial))) return rc;
    a.counts = counts; a.bits = bits; a.N = (int)((long long)H * W / 64); a.L = L;
    const size_t bits_per_frame = blk == 8 ? (size_t)((long long)H * W / 256) : (size_t)a.N;     // dwt_dct_svd_decoder.py:14
    if (!(a.scales[1] > 0.f)) {
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (counts) HIP_TRY(launch_zero(counts, (size_t)n * L * sizeof(int32_t) * (a.partial ? (size_t)svd_count_tiles(H, W, blk) : 1), s));
        if (bits) HIP_TRY(launch_zero(bits, (size_t)n * bits_per_frame, s));
        return OFMK_OK;
    }
    if (blk == 8) return launch_svd8_rgb8(in, nullptr, n, H, W, SVD_DETECT, to_args8(a, H, W), make_ctx(stream, opts));
    return launch_svd_rgb8(in, nullptr, n, H, W, SVD_DETECT, a, make_ctx(stream, opts));
}

int ofmk_svd_embed_detect_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W, const uint8_t *wm, int n_wm,
                               const int32_t *wm_row, const double *scales, int blk, int L, int32_t *counts, uint8_t *bits,
                               void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    if ((rc = check_blk(blk))) return rc;
    if ((rc = check_detect_args(out, n, H, W, L, counts, bits))) return rc;
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    if ((rc = set_scales(a, scales, false))) return rc;
    if ((rc = check_partial(opts, L, counts, a.partial))) return rc;
    a.wm = wm; a.wm_row = wm_row; a.n_wm = n_wm; a.counts = counts; a.bits = bits;
    a.N = (int)((long long)H * W / 64); a.L = L;
    if (blk == 8) return launch_svd8_rgb8(in, out, n, H, W, SVD_EMBED_VERIFY, to_args8(a, H, W), make_ctx(stream, opts));
    return launch_svd_rgb8(in, out, n, H, W, SVD_EMBED_VERIFY, a, make_ctx(stream, opts));
}

int ofmk_svd_encode_yuv32f(float *yuv, int n, int H, int W, const uint8_t *wm, int n_wm, const int32_t *wm_row,
                           const double *scales, int blk, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(yuv, yuv, n, H, W, wm, n_wm);
    if (rc) return rc;
    if ((rc = check_blk(blk))) return rc;
    if (n > kMaxChunk) return fail(OFMK_E_ARG, "too many frames%s");
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    if ((rc = set_scales(a, scales, false))) return rc;
    a.wm = wm; a.wm_row = wm_row; a.n_wm = n_wm; a.N = (int)((long long)H * W / 64); a.L = 1;
    const Ctx cx = make_ctx(stream, opts);
    if (blk == 8) {
        const Geom8 g8 = make_geom8(H, W);
        if (g8.ntile > 0) {                               // an event pair is reserved only for a launch that happens
            ScopedTiming timing(KIND_SVD, cx);
            OFMK_TIMED_LAUNCH(timing, (svd8_yuv32f_kernel<SVD_EMBED>), dim3((unsigned)((g8.ntile + kThreads - 1) / kThreads), (unsigned)n), dim3(kThreads), 0,
                              cx.s, yuv, g8, to_args8(a, H, W));
        }
        HIP_TRY(hipGetLastError());
        return OFMK_OK;
    }
    Workspace none;
    none.plane = 0;
    const Geom g = make_geom(H, W, none);
    ScopedTiming timing(KIND_SVD, cx);                    // the plugin path's launches are timed like every other
    OFMK_TIMED_LAUNCH(timing, (svd_yuv32f_kernel<SVD_EMBED>), block_grid(g, n), dim3(kThreads), 0, cx.s, yuv, g, a);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int ofmk_svd_decode_yuv32f(const float *yuv, int n, int H, int W, const double *scales, int blk, uint8_t *bits, void *stream,
                           const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if ((rc = check_blk(blk))) return rc;
    if (!yuv || !bits) return fail(OFMK_E_ARG, "null pointer%s");
    if (n > kMaxChunk) return fail(OFMK_E_ARG, "too many frames%s");
    SvdArgs a;
    memset(&a, 0, sizeof(a));
    if ((rc = set_scales(a, scales, true))) return rc;
    a.bits = bits; a.N = (int)((long long)H * W / 64); a.L = 1;
    const Ctx cx = make_ctx(stream, opts);
    hipStream_t s = cx.s;
    if (blk == 8) {                                       // bits: [n][H*W/256] (dwt_dct_svd_decoder.py:14)
        const Geom8 g8 = make_geom8(H, W);
        const Svd8Args a8 = to_args8(a, H, W);
        if (a8.N8 > g8.ntile || !(a.scales[1] > 0.f)) HIP_TRY(launch_zero(bits, (size_t)n * a8.N8, s));
        if (!(a.scales[1] > 0.f) || g8.ntile == 0) return OFMK_OK;
        ScopedTiming timing(KIND_SVD, cx);
        OFMK_TIMED_LAUNCH(timing, (svd8_yuv32f_kernel<SVD_DETECT>), dim3((unsigned)((g8.ntile + kThreads - 1) / kThreads), (unsigned)n), dim3(kThreads), 0, s,
                          const_cast<float *>(yuv), g8, a8);
        HIP_TRY(hipGetLastError());
        return OFMK_OK;
    }
    Workspace none;
    none.plane = 0;
    const Geom g = make_geom(H, W, none);
    if (a.N > g.nblk || !(a.scales[1] > 0.f)) HIP_TRY(launch_zero(bits, (size_t)n * a.N, s));
    if (!(a.scales[1] > 0.f)) return OFMK_OK;
    ScopedTiming timing(KIND_SVD, cx);
    OFMK_TIMED_LAUNCH(timing, (svd_yuv32f_kernel<SVD_DETECT>), block_grid(g, n), dim3(kThreads), 0, s, const_cast<float *>(yuv), g, a);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int ofmk_payloads_from_counts(const int32_t *counts, int n, int L, int n_bits, const int32_t *perm, uint8_t *payload,
                              void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    if (!counts || !perm || !payload) return fail(OFMK_E_ARG, "null pointer%s");
    if (n < 1 || L < 1 || n_bits < 0) return fail(OFMK_E_ARG, "bad sizes%s");
    return launch_payloads(counts, n, L, n_bits, perm, payload, static_cast<hipStream_t>(stream));
}

int ofmk_svd_count_tiles(int H, int W, int blk) {
    if (H < 8 || W < 8 || check_blk(blk)) return -1;
    return svd_count_tiles(H, W, blk);
}

int ofmk_payloads_from_partial_counts(const int32_t *partials, int tiles, int n, int L, int n_bits, const int32_t *perm,
                                      uint8_t *payload, int32_t *counts, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    if (!partials || (!payload && !counts) || (payload && !perm)) return fail(OFMK_E_ARG, "null pointer%s");
    if (n < 1 || L < 1 || n_bits < 0 || tiles < 0) return fail(OFMK_E_ARG, "bad sizes%s");
    if (L > kHistMax) return fail(OFMK_E_ARG, "partial counts need L <= 2048%s");
    if ((long long)tiles * L >= (1LL << 31)) return fail(OFMK_E_ARG, "tiles * L too large%s");
    hipLaunchKernelGGL(degenerate_partial_kernel, dim3((unsigned)n), dim3(kThreads), 0, static_cast<hipStream_t>(stream), partials, tiles, L,
                       n_bits, perm, payload, counts);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

// ---- planar YUV 4:2:0 entry points ------------------------------------------------------
int ofmk_embed_yuv420(const uint8_t *in, uint8_t *out, int layout, int n, int H, int W, const uint8_t *wm, int n_wm,
                      const int32_t *wm_row, double alpha, int chunk_frames, void *workspace, size_t workspace_bytes,
                      void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    if ((rc = check_planar(layout, H, W, in, out))) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    const size_t fs = (size_t)H * W * 3 / 2;
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = launch_analyze_yuv420(in + (size_t)f0 * fs, layout, cf, H, W, ws, cx))) return rc;
        if ((rc = launch_mark_yuv420(in + (size_t)f0 * fs, out + (size_t)f0 * fs, layout, cf, H, W, wm, n_wm, wm_row ? wm_row + f0 : nullptr,
                                     alpha, ws, false, cx))) return rc;
    }
    return OFMK_OK;
}

int ofmk_detect_yuv420(const uint8_t *in, int layout, int n, int H, int W, int L, double alpha, int32_t *counts, uint8_t *bits,
                       int chunk_frames, void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_detect_args(in, n, H, W, L, counts, bits);
    if (rc) return rc;
    if ((rc = check_planar(layout, H, W, in, nullptr))) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    const size_t fs = (size_t)H * W * 3 / 2;
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        if ((rc = launch_analyze_yuv420(in + (size_t)f0 * fs, layout, cf, H, W, ws, cx, counts ? counts + (size_t)f0 * L : nullptr, L))) return rc;
        if ((rc = finalize_detect(f0, cf, H, W, L, alpha, counts, bits, ws, false, cx))) return rc;
    }
    return OFMK_OK;
}

int ofmk_embed_detect_yuv420(const uint8_t *in, uint8_t *out, int layout, int n, int H, int W, const uint8_t *wm, int n_wm,
                             const int32_t *wm_row, double alpha, int L, int32_t *counts, uint8_t *bits, int chunk_frames,
                             void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_embed_args(in, out, n, H, W, wm, n_wm);
    if (rc) return rc;
    if ((rc = check_detect_args(out, n, H, W, L, counts, bits))) return rc;
    if ((rc = check_planar(layout, H, W, in, out))) return rc;
    Workspace ws;
    if ((rc = carve(workspace, workspace_bytes, H, W, chunk_frames, ws))) return rc;
    const Ctx cx = make_ctx(stream, opts);
    const size_t fs = (size_t)H * W * 3 / 2;
    const bool fused = !(cx.flags & OFMK_F_SEPARATE_DETECT);
    for (int f0 = 0; f0 < n; f0 += ws.frames) {
        const int cf = n - f0 < ws.frames ? n - f0 : ws.frames;
        const uint8_t *pin = in + (size_t)f0 * fs;
        uint8_t *pout = out + (size_t)f0 * fs;
        int32_t *zc = counts ? counts + (size_t)f0 * L : nullptr;
        if ((rc = launch_analyze_yuv420(pin, layout, cf, H, W, ws, cx, fused ? zc : nullptr, L))) return rc;
        if ((rc = launch_mark_yuv420(pin, pout, layout, cf, H, W, wm, n_wm, wm_row ? wm_row + f0 : nullptr, alpha, ws, fused, cx))) return rc;
        if (fused) {
            if ((rc = finalize_detect(f0, cf, H, W, L, alpha, counts, bits, ws, true, cx))) return rc;
        } else {
            if ((rc = launch_analyze_yuv420(pout, layout, cf, H, W, ws, cx, zc, L))) return rc;
            if ((rc = finalize_detect(f0, cf, H, W, L, alpha, counts, bits, ws, false, cx))) return rc;
        }
    }
    return OFMK_OK;
}

int ofmk_yuv420_to_rgb8(const uint8_t *yuv, uint8_t *rgb, int layout, int n, int H, int W, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!yuv || !rgb) return fail(OFMK_E_ARG, "null pointer%s");
    if ((rc = check_planar(layout, H, W, yuv, rgb))) return rc;
    const PGeom g = make_pgeom(layout, H, W, 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int f0 = 0; f0 < n; f0 += kMaxChunk) {
        const int cf = n - f0 < kMaxChunk ? n - f0 : kMaxChunk;
        const dim3 grid((unsigned)((g.nblk + kThreads - 1) / kThreads), (unsigned)cf);
        const uint8_t *pi = yuv + (size_t)f0 * g.frame_stride;
        uint8_t *po = rgb + (size_t)f0 * H * W * 3;
        if (layout == OFMK_YUV_I420) hipLaunchKernelGGL(yuv420_to_rgb8_kernel<FMT_I420>, grid, dim3(kThreads), 0, s, pi, po, g);
        else hipLaunchKernelGGL(yuv420_to_rgb8_kernel<FMT_NV12>, grid, dim3(kThreads), 0, s, pi, po, g);
    }
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int ofmk_rgb8_to_yuv420(const uint8_t *rgb, uint8_t *yuv, int layout, int n, int H, int W, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    int rc = check_dims(n, H, W);
    if (rc) return rc;
    if (!yuv || !rgb) return fail(OFMK_E_ARG, "null pointer%s");
    if ((rc = check_planar(layout, H, W, yuv, rgb))) return rc;
    const PGeom g = make_pgeom(layout, H, W, 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int f0 = 0; f0 < n; f0 += kMaxChunk) {
        const int cf = n - f0 < kMaxChunk ? n - f0 : kMaxChunk;
        const dim3 grid((unsigned)((g.nblk + kThreads - 1) / kThreads), (unsigned)cf);
        const uint8_t *pi = rgb + (size_t)f0 * H * W * 3;
        uint8_t *po = yuv + (size_t)f0 * g.frame_stride;
        if (layout == OFMK_YUV_I420) hipLaunchKernelGGL(rgb8_to_yuv420_kernel<FMT_I420>, grid, dim3(kThreads), 0, s, pi, po, g);
        else hipLaunchKernelGGL(rgb8_to_yuv420_kernel<FMT_NV12>, grid, dim3(kThreads), 0, s, pi, po, g);
    }
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int ofmk_probe_xcc(int32_t *xcc_of_workgroup, int n_workgroups, void *stream, const ofmk_opts *opts) {
    if (int orc = check_opts(opts)) return orc;
    if (!xcc_of_workgroup || n_workgroups < 1 || n_workgroups > (1 << 20)) return fail(OFMK_E_ARG, "probe needs an output array and 1..2^20 workgroups%s");
    hipLaunchKernelGGL(probe_xcc_kernel, dim3((unsigned)n_workgroups), dim3(64), 0, static_cast<hipStream_t>(stream), xcc_of_workgroup, n_workgroups);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

// ---- timing objects (caller-owned; see the header) ------------------------------------------------
int ofmk_timing_create(int max_launches, unsigned kind_mask, ofmk_timing **out) {
    if (!out) return fail(OFMK_E_ARG, "null output%s");
    *out = nullptr;
    if (max_launches < 1) return fail(OFMK_E_ARG, "max_launches must be positive%s");
    ofmk_timing *t = new ofmk_timing;
    t->rec = new TimingRec[max_launches];
    t->cap = max_launches;
    t->used.store(0);
    t->mask = kind_mask ? kind_mask : 0xFFFFFFFFu;
    for (int i = 0; i < max_launches; ++i) { t->rec[i].a = nullptr; t->rec[i].b = nullptr; t->rec[i].kind = -1; }
    for (int i = 0; i < max_launches; ++i) {
        hipError_t e = hipEventCreate(&t->rec[i].a);
        if (e == hipSuccess) e = hipEventCreate(&t->rec[i].b);
        if (e != hipSuccess) {
            ofmk_timing_destroy(t);                 // destroys whichever events exist (the others are null)
            return fail(OFMK_E_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    *out = t;
    return OFMK_OK;
}

int ofmk_timing_collect(ofmk_timing *t, double *ms_by_kind, int *launches_by_kind) {
    if (!t || !ms_by_kind || !launches_by_kind) return fail(OFMK_E_ARG, "null pointer%s");
    for (int k = 0; k < KIND_COUNT; ++k) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
    const int taken = t->used.load();
    const int used = taken < t->cap ? taken : t->cap;
    int rc = OFMK_OK;
    for (int i = 0; i < used; ++i) {
        // a pair whose launch failed after it was reserved was never recorded: skip it, do not poison the pool
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(t->rec[i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, t->rec[i].a, t->rec[i].b);
        if (e == hipSuccess && t->rec[i].kind >= 0 && t->rec[i].kind < KIND_COUNT) {
            ms_by_kind[t->rec[i].kind] += ms;
            launches_by_kind[t->rec[i].kind] += 1;
        } else if (e != hipSuccess && e != hipErrorInvalidHandle && e != hipErrorInvalidResourceHandle && e != hipErrorNotReady) {
            rc = fail(OFMK_E_HIP, "timing collect: %s", hipGetErrorString(e));
        }
        if (e != hipSuccess) (void)hipGetLastError();
        t->rec[i].kind = -1;
    }
    t->used.store(0);
    if (rc) return rc;
    return OFMK_OK;
}

int ofmk_timing_durations(ofmk_timing *t, float *ms_per_launch, int *kind_per_launch, int cap) {
    if (!t || !ms_per_launch || cap < 0) return fail(OFMK_E_ARG, "null pointer%s");
    const int taken = t->used.load();
    const int used = taken < t->cap ? taken : t->cap;
    int n = 0;
    for (int i = 0; i < used && n < cap; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(t->rec[i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, t->rec[i].a, t->rec[i].b);
        if (e != hipSuccess) { (void)hipGetLastError(); continue; }      // a pair whose launch failed was never recorded
        ms_per_launch[n] = ms;
        if (kind_per_launch) kind_per_launch[n] = t->rec[i].kind;
        ++n;
    }
    return n;
}

void ofmk_timing_destroy(ofmk_timing *t) {
    if (!t) return;
    for (int i = 0; i < t->cap; ++i) {
        if (t->rec[i].a) (void)hipEventDestroy(t->rec[i].a);
        if (t->rec[i].b) (void)hipEventDestroy(t->rec[i].b);
    }
    delete[] t->rec;
    delete t;
}

// ---- streaming probes: the device's achievable HBM rates, measured in the same run as the kernels ----
int ofmk_hbm_copy(const void *src, void *dst, size_t bytes, void *stream) {
    if (!src || !dst || bytes % 16 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
        return fail(OFMK_E_ARG, "copy needs 16-byte aligned pointers and size%s");
    const size_t n16 = bytes / 16;
    if (n16 == 0) return OFMK_OK;
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)((n16 + kCopyPerWg - 1) / kCopyPerWg)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16);
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

int ofmk_hbm_read(const void *src, size_t bytes, void *sink, void *stream) {
    if (!src || !sink || bytes % 16 || (uintptr_t)src % 16 || (uintptr_t)sink % 4)
        return fail(OFMK_E_ARG, "read probe needs a 16-byte aligned source and size, and a 4-byte sink%s");
    const size_t n16 = bytes / 16;
    if (n16 == 0) return OFMK_OK;
    hipLaunchKernelGGL(read16_kernel, dim3((unsigned)((n16 + kReadPerWg - 1) / kReadPerWg)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint4 *>(src), n16, static_cast<uint32_t *>(sink));
    HIP_TRY(hipGetLastError());
    return OFMK_OK;
}

}  // extern "C"
