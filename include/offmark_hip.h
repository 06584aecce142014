/* offmark_hip.h -- C ABI of the MI355X (gfx950) DCT frame-watermark engine.
 *
 * The reference (vikasdimaniya/video-fingerprinting, "offmark") is pure Python and has no FFI
 * of its own; each entry point below names the reference interface it replaces
 * (paths relative to the reference root).  The Python package `offmark` in this repository
 * binds these symbols with ctypes (video-fingerprinting_amd/offmark/_hip.py); INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer marked "device" is HBM memory owned by the caller (e.g. torch tensors);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is only
 *     ENQUEUED on it: no allocation, no host synchronisation, no internal threads, so every
 *     compute call may be captured into a hipGraph (unless it carries an ofmk_timing object);
 *   - re-entrant: the library has NO mutable state besides the calling thread's error text.
 *     Everything a call needs arrives in its arguments; per-call options travel in `ofmk_opts`
 *     (NULL = defaults), the LAST argument of EVERY compute entry point (since ABI 3; only the
 *     bandwidth probes and the timing-object functions at the end of this file take none).  Two host threads may drive two engines (own workspace, own stream,
 *     own timing object) concurrently (tests/test_gpu_parity.py::test_two_threads_two_engines);
 *   - return value 0 = OK, negative = error (OFMK_E_*); ofmk_last_error() gives the text for
 *     the calling thread; nothing throws across this boundary;
 *   - frames are interleaved 8-bit, 3 channels, row-major [n][H][W][3] exactly as
 *     FileDecoder.read() delivers them (src/offmark/video/frame_reader.py:53-64).  Channel 0
 *     is treated as "B" by the colour transform, as the reference does
 *     (src/offmark/video/embedder.py:34).
 *   - one watermark bit per 8x8 pixel block, raster order, N = H*W/64 entries per frame of
 *     which the first (H/8)*(W/8) are used (src/offmark/embed/dct_encoder.py:13-16,25-26).
 */
#ifndef OFFMARK_HIP_H
#define OFFMARK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFMK_ABI_VERSION 6

#define OFMK_OK            0
#define OFMK_E_ARG        -1   /* null pointer / non-positive size / H or W < 8 */
#define OFMK_E_WORKSPACE  -2   /* workspace smaller than ofmk_workspace_bytes(1, H, W) */
#define OFMK_E_HIP        -3   /* a HIP call failed; text in ofmk_last_error() */

int ofmk_version(void);
const char *ofmk_last_error(void);

/* Per-call options (the reference's codecs are stateless objects configured by constructor kwargs,
 * src/offmark/embed/dct_encoder.py:6-16; this is the same idea at the C boundary).  Pass NULL for defaults. */
typedef struct ofmk_timing ofmk_timing;     /* opaque, see ofmk_timing_create */
typedef struct ofmk_opts {
    uint32_t flags;          /* OFMK_F_*; unknown bits are rejected (OFMK_E_ARG) */
    uint32_t xcds;           /* XCDs the XCD-aware tile order assumes: 0 = 8 (MI355X, SPX mode), 1 = linear order, > 64 rejected.
                                ABI 3 called this word `reserved` and required 0 */
    ofmk_timing *timing;     /* NULL = launches carry no events */
} ofmk_opts;
/* ofmk_embed_detect_rgb8: embed, then detect the written frames with the stand-alone detect kernels
 * (analyze runs on the marked frames: 12 B/px of traffic) instead of the fused mark+verify kernel
 * (9 B/px).  Same results bit for bit. */
#define OFMK_F_SEPARATE_DETECT 1u
/* Tile order of the frame-WRITING DCT kernel (ofmk_embed_rgb8, ofmk_embed_detect_rgb8, ofmk_stage_mark_rgb8).  The kernel is
 * one linear grid of 48 KiB tiles; the hardware deals consecutive workgroups round-robin to the XCDs.  XCD-aware order: every XCD walks one contiguous 1/xcds of the launch's frames (tile = (L % xcds) * ceil(G / xcds)
 * + L / xcds); linear order: tile = workgroup index.  A pure permutation of the work: results are identical bit for bit either
 * way.  DEFAULT (neither flag, since ABI 5): a static rule on the launch's size -- XCD-aware when the launch reads at least
 * OFMK_XCD_TILES_MIN_BYTES of frames (192 frames of 1080p), linear below -- which is what interleaved A/B runs on MI355X say
 * (large launches: XCD-aware wins by 1.5-6 % or ties, depending on where the driver placed the caller's frames; 48-96 frames of
 * 1080p: linear wins by 1-4 %; profiles/r4_mark_fused_pass.txt, profiles/r6_mark_ladder.txt).
 * The two flags force an order (a caller that has measured its own box); both at once are rejected.  ABI 4 defaulted to the XCD-aware order at every size.  The read-only and one-pass kernels always run in linear
 * order (measured faster there).  The reference has no counterpart: its loop is one frame at a time
 * (src/offmark/video/embedder.py:18-31). */
#define OFMK_F_LINEAR_TILES 2u
#define OFMK_F_XCD_TILES 4u
#define OFMK_XCD_TILES_MIN_BYTES 1194393600ull      /* 192 x 1080 x 1920 x 3 */
/* DwtDctSvd read-outs (ofmk_svd_detect_rgb8, ofmk_svd_embed_detect_rgb8) only: `counts` is the PARTIAL form, device int32
 * [n][tiles][L] with tiles = ofmk_svd_count_tiles(H, W, blk) -- every workgroup of the frame kernel STORES the L sums of its own
 * tile, so the launch clears nothing first (no fill dispatch, no global atomics) and the buffer may hold anything before the
 * call.  ofmk_payloads_from_partial_counts adds the tiles up and runs DeShuffler.degenerate's epilogue
 * (de_shuffler.py:17-22; dwt_dct_svd_decoder.py:12-37 produced the bits).  L <= 2048 (else OFMK_E_ARG: use plain counts);
 * every other entry point ignores the flag. */
#define OFMK_F_PARTIAL_COUNTS 8u
/* Bytes of device scratch needed to process `frames_in_flight` frames per internal chunk.
 * Any workspace >= ofmk_workspace_bytes(1, H, W) is accepted; the engine sizes its chunks to
 * what fits.  Bigger chunks are faster (fewer launches, shorter tails): keeping a chunk resident
 * in the 256 MiB Infinity Cache between the two passes was measured NOT to pay. */
size_t ofmk_workspace_bytes(int frames_in_flight, int H, int W);

/* ---- embed: replaces Embedder.__mark_frame + DctEncoder.encode for a batch of frames ------
 * src/offmark/video/embedder.py:33-39, src/offmark/embed/dct_encoder.py:18-39.
 *   in, out   device u8 [n][H][W][3]; out may alias in (in-place)
 *   wm        device u8 [n_wm][N] of 0/1, N = H*W/64 (DctEncoder.read_wm keeps row 0 of the
 *             generator's (1,N) array; n_wm > 1 lets segments carry different payloads)
 *   wm_row    device int32 [n] giving the wm row of each frame, or NULL = row 0 for all.  The
 *             reference has one watermark per encoder (dct_encoder.py:10-11); the row map is this
 *             build's extension and so is its safety: the kernels clamp every entry into
 *             [0, n_wm) (an out-of-range entry reads the nearest valid row, never out of bounds)
 *   alpha     DctEncoder(alpha=20)
 *   chunk_frames  frames per internal chunk, 0 = as many as the workspace holds            */
int ofmk_embed_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W,
                    const uint8_t *wm, int n_wm, const int32_t *wm_row, double alpha,
                    int chunk_frames, void *workspace, size_t workspace_bytes, void *stream,
                    const ofmk_opts *opts);

/* ---- detect: replaces Extractor.__check_frame + DctDecoder.decode + the bits[i::L] sums ----
 * src/offmark/video/extractor.py:30-34, src/offmark/extract/dct_decoder.py:10-27,
 * src/offmark/degenerator/de_shuffler.py:17-18.
 *   counts    device int32 [n][L]: number of 1 bits among raw_bits[i::L] (host finishes
 *             DeShuffler.degenerate: mean, un-permute, mid-range threshold)
 *   bits      device u8 [n][N] raw per-block bits (DctDecoder.decode's array), or NULL     */
int ofmk_detect_rgb8(const uint8_t *in, int n, int H, int W, int L, double alpha,
                     int32_t *counts, uint8_t *bits,
                     int chunk_frames, void *workspace, size_t workspace_bytes, void *stream,
                     const ofmk_opts *opts);

/* ---- soft-decision read-out (BUILD EXTENSION, not reference semantics) -------------
 * soft: device int64 [n][L]; soft[f][i] = sum over blocks c with c mod L == i of round(-cos(pi*C21/step) * 2^14):
 * positive means position i reads as 1, the magnitude is a confidence.  Sums over frames of a segment can be
 * added before thresholding at 0 (offmark.dist.vote.soft_vote).  The reference's hard decision stays the
 * default everywhere. */
int ofmk_detect_soft_rgb8(const uint8_t *in, int n, int H, int W, int L, double alpha, long long *soft,
                          int chunk_frames, void *workspace, size_t workspace_bytes, void *stream,
                          const ofmk_opts *opts);

/* ---- embed then detect the produced frames, chunk by chunk (mark + verify) ---------------
 * The shape of tests/mark_video_to_hls.py:356-389 (verify every marked copy).  Same results as
 * ofmk_embed_rgb8 followed by ofmk_detect_rgb8 on `out`.  By default the mark kernel also analyzes
 * the marked block it still holds in registers, so detect's read of the frame is saved
 * (opts->flags & OFMK_F_SEPARATE_DETECT runs the literal two-call sequence instead). */
int ofmk_embed_detect_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W,
                           const uint8_t *wm, int n_wm, const int32_t *wm_row, double alpha,
                           int L, int32_t *counts, uint8_t *bits,
                           int chunk_frames, void *workspace, size_t workspace_bytes, void *stream,
                           const ofmk_opts *opts);

/* ---- DwtDctSvd codec (what tests/mark.py and tests/detect.py construct) -----------------------
 * src/offmark/embed/dwt_dct_svd_encoder.py:19-45 (Haar LL of channel 1 -> 4x4 blocks -> DCT -> SVD ->
 * s[0] = (s[0] // scale + 0.25 + 0.5*bit) * scale -> back) and
 * src/offmark/extract/dwt_dct_svd_decoder.py:12-37 (bit = (s[0] % scale) > scale/2), blk = 4.
 *   scales   HOST array of 3 doubles, one per YUV channel as in DwtDctSvdEncoder(scales=[0,15,0])
 *            (dwt_dct_svd_encoder.py:6,19-26): every channel with a positive scale is marked with the same
 *            watermark; the read-out is channel 1's (dwt_dct_svd_decoder.py:24 returns wm_bits[1]), so
 *            detection with scales[1] <= 0 yields zeros, as in the reference.  A positive scale must be a normal
 *            float32 >= 1e-3 after conversion (smaller steps are below the float32 resolution of typical top
 *            singular values; rejected with OFMK_E_ARG), NaN / infinities are rejected.
 *   blk      DwtDctSvdEncoder(blk=4): the LL block size, 4 (the reference's default: 8x8 pixel tiles, one bit per tile,
 *            N = H*W/64 bits) or 8 (16x16 pixel tiles: tile c takes wm[c], so the first quarter of the watermark row is
 *            used, dwt_dct_svd_encoder.py:29-40, and `bits` is [n][H*W/256], dwt_dct_svd_decoder.py:14).  Other values:
 *            OFMK_E_ARG (blk < 4 indexes past the reference's own watermark; larger blocks are not built).
 * Same frame/watermark/counts/bits conventions as the DCT entry points; no workspace (this codec has no
 * frame-global dependency: one pass).                                                            */
int ofmk_svd_embed_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W,
                        const uint8_t *wm, int n_wm, const int32_t *wm_row, const double *scales, int blk, void *stream,
                        const ofmk_opts *opts);
int ofmk_svd_detect_rgb8(const uint8_t *in, int n, int H, int W, int L, const double *scales, int blk,
                         int32_t *counts, uint8_t *bits, void *stream, const ofmk_opts *opts);
int ofmk_svd_embed_detect_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W,
                               const uint8_t *wm, int n_wm, const int32_t *wm_row, const double *scales, int blk,
                               int L, int32_t *counts, uint8_t *bits, void *stream, const ofmk_opts *opts);
/* plugin level, float32 YUV [n][H][W][3] (n <= 65535): encode mutates the marked channels; decode fills bits */
int ofmk_svd_encode_yuv32f(float *yuv, int n, int H, int W,
                           const uint8_t *wm, int n_wm, const int32_t *wm_row, const double *scales, int blk, void *stream,
                           const ofmk_opts *opts);
int ofmk_svd_decode_yuv32f(const float *yuv, int n, int H, int W, const double *scales, int blk, uint8_t *bits, void *stream,
                           const ofmk_opts *opts);

/* ---- planar 8-bit YUV 4:2:0 on either side of the DCT codec ------------------------------
 * The reference moves rgb24 over pipes and has ffmpeg convert to yuv420p on the way out
 * (src/offmark/video/frame_reader.py:42-64, src/offmark/video/frame_writer.py:33-34).  These entry points take and
 * produce the 4:2:0 planes themselves, so a decoder's output stays on the device and HBM / PCIe carry 1.5 instead
 * of 3 bytes per pixel each way.  The result is, bit for bit,
 *     ofmk_yuv420_to_rgb8 -> ofmk_embed_rgb8 / ofmk_detect_rgb8 -> ofmk_rgb8_to_yuv420
 * with the conversions fused into the kernels' loads and stores.  The conversion is BUILD-DEFINED (swscale is not
 * available here and its rounding is not claimed): BT.601 studio swing in float32 fused multiply-adds, clip to
 * [0,255], round half to even; chroma is replicated over its 2x2 pixels on the way in and is the conversion of the
 * 2x2 mean RGB on the way out (csrc/planar_kernels.hiph, restated in oracle/offmark_oracle.py).
 *   layout  OFMK_YUV_I420: per frame [Y: H*W][U: H/2*W/2][V: H/2*W/2];  OFMK_YUV_NV12: [Y: H*W][UV interleaved: H/2*W]
 *           frames are consecutive, 1.5*H*W bytes each; H and W must be multiples of 8, buffers 8-byte aligned
 *   ofmk_embed_detect_yuv420's counts/bits are those of a reader of the WRITTEN planes (== ofmk_detect_yuv420(out)). */
#define OFMK_YUV_I420 0
#define OFMK_YUV_NV12 1
int ofmk_embed_yuv420(const uint8_t *in, uint8_t *out, int layout, int n, int H, int W,
                      const uint8_t *wm, int n_wm, const int32_t *wm_row, double alpha,
                      int chunk_frames, void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts);
int ofmk_detect_yuv420(const uint8_t *in, int layout, int n, int H, int W, int L, double alpha,
                       int32_t *counts, uint8_t *bits,
                       int chunk_frames, void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts);
int ofmk_embed_detect_yuv420(const uint8_t *in, uint8_t *out, int layout, int n, int H, int W,
                             const uint8_t *wm, int n_wm, const int32_t *wm_row, double alpha,
                             int L, int32_t *counts, uint8_t *bits,
                             int chunk_frames, void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts);
int ofmk_yuv420_to_rgb8(const uint8_t *yuv, uint8_t *rgb, int layout, int n, int H, int W, void *stream,
                        const ofmk_opts *opts);
int ofmk_rgb8_to_yuv420(const uint8_t *rgb, uint8_t *yuv, int layout, int n, int H, int W, void *stream,
                        const ofmk_opts *opts);

/* ---- DeShuffler.degenerate's epilogue for a batch, on the device ---------------------------
 * src/offmark/degenerator/de_shuffler.py:17-22: mean of bits[i::L] (from `counts`), undo the key
 * permutation (`perm` = DeShuffler.payload_idx, device int32 [L]), threshold strictly above the
 * mid-range of the L means.  payload: device u8 [n][L].  n_bits = H*W/64.                   */
int ofmk_payloads_from_counts(const int32_t *counts, int n, int L, int n_bits, const int32_t *perm,
                              uint8_t *payload, void *stream, const ofmk_opts *opts);
/* Rows of a frame's partial counts (OFMK_F_PARTIAL_COUNTS) for the DwtDctSvd codec with this blk; negative on bad arguments. */
int ofmk_svd_count_tiles(int H, int W, int blk);
/* The payload epilogue from partial counts [n][tiles][L] (OFMK_F_PARTIAL_COUNTS): per frame, counts[i] = sum over tiles, then
 * exactly ofmk_payloads_from_counts (de_shuffler.py:17-22).  payload: device u8 [n][L] or NULL; counts: device int32 [n][L]
 * or NULL (the summed counts, for callers that want DeShuffler's numerators); at least one of the two.  L <= 2048. */
int ofmk_payloads_from_partial_counts(const int32_t *partials, int tiles, int n, int L, int n_bits, const int32_t *perm,
                                      uint8_t *payload, int32_t *counts, void *stream, const ofmk_opts *opts);

/* ---- plugin-level entry points on float32 YUV frames --------------------------------------
 * DctEncoder.encode(yuv) (dct_encoder.py:18-39; mutates channel 1 in place) and
 * DctDecoder.decode(yuv) (dct_decoder.py:10-27).  yuv: device f32 [n][H][W][3].            */
int ofmk_encode_yuv32f(float *yuv, int n, int H, int W,
                       const uint8_t *wm, int n_wm, const int32_t *wm_row, double alpha,
                       int chunk_frames, void *workspace, size_t workspace_bytes, void *stream,
                       const ofmk_opts *opts);
int ofmk_decode_yuv32f(const float *yuv, int n, int H, int W, int L, double alpha,
                       int32_t *counts, uint8_t *bits,
                       int chunk_frames, void *workspace, size_t workspace_bytes, void *stream,
                       const ofmk_opts *opts);

/* ---- parity / debug planes for ONE frame (any pointer may be NULL) -----------------------
 * DctEncoder.luminance_mask / texture_mask (dct_encoder.py:41-102) and the [2][1] coefficient
 * before and after quantisation (dct_encoder.py:29-35).  All planes are [H/8][W/8].
 *   src_is_yuv32f = 0: frame is u8 [H][W][3];  1: frame is f32 YUV [H][W][3]
 *   wm may be NULL (then c21_post is not produced)                                          */
int ofmk_debug_planes(const void *frame, int src_is_yuv32f, int H, int W, double alpha,
                      const uint8_t *wm,
                      float *y_dc, double *lum_mask, double *tex_mask, double *step,
                      float *c21_pre, float *c21_post,
                      void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts);

/* ---- individual stages (bench.py and tests drive single kernels with these) ----------------
 * analyze : frames -> per-block records (the kernel shared by embed and detect)
 * mark    : frames + the records the analyze stage left in the workspace for the SAME frames +
 *           watermark (row 0 for every frame) -> marked frames; fused != 0 also analyzes the
 *           marked frames (mark + verify kernel)                                             */
int ofmk_stage_analyze_rgb8(const uint8_t *in, int n, int H, int W,
                            void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts);
int ofmk_stage_mark_rgb8(const uint8_t *in, uint8_t *out, int n, int H, int W,
                         const uint8_t *wm, double alpha, int fused,
                         void *workspace, size_t workspace_bytes, void *stream, const ofmk_opts *opts);

/* Streaming probes: bench.py measures the device's achievable HBM rates with them in the same run as the
 * kernels.  ofmk_hbm_copy: device-to-device copy, 16 bytes per lane and access, registers only (reads `bytes`,
 * writes `bytes`).  ofmk_hbm_read: read-only stream of `bytes`; `sink` is a device uint32 the kernel
 * practically never writes (it only keeps the loads alive). */
int ofmk_hbm_copy(const void *src, void *dst, size_t bytes, void *stream);
int ofmk_hbm_read(const void *src, size_t bytes, void *sink, void *stream);

/* Which XCD each workgroup of a linear grid of `n_workgroups` 64-thread workgroups runs on (HW_REG_XCC_ID): device int32
 * [n_workgroups].  HIP promises nothing about that deal; the XCD-aware tile order assumes workgroups L and L + xcds share an
 * XCD (speed only, never correctness), and this lets a host check it and count the XCDs (bench.py: `mark_order.xcc_deal`). */
int ofmk_probe_xcc(int32_t *xcc_of_workgroup, int n_workgroups, void *stream, const ofmk_opts *opts);

/* Per-launch HIP-event timing for bench.py.  A timing object owns 2*max_launches events; while it is passed in
 * ofmk_opts.timing every launch of a kernel kind selected in kind_mask (bit k = kind k, 0 = all) carries an event
 * pair as the dispatch's own start/stop events (hipExtLaunchKernelGGL) on the launch stream, so no marker packets
 * separate consecutive kernels.  collect() waits for the recorded events, returns the summed milliseconds and
 * launch counts per kernel kind (0 analyze, 1 finalize, 2 mark, 3 fused mark+analyze, 4 DwtDctSvd, 5 planar 4:2:0
 * analyze, 6 planar 4:2:0 mark) and rewinds the pool.  One object per engine / host thread; a call that carries one cannot be
 * captured into a hipGraph (events on the dispatch). */
#define OFMK_TIMING_KINDS 7
int ofmk_timing_create(int max_launches, unsigned kind_mask, ofmk_timing **out);
int ofmk_timing_collect(ofmk_timing *t, double *ms_by_kind /*[OFMK_TIMING_KINDS]*/, int *launches_by_kind /*[OFMK_TIMING_KINDS]*/);
/* The recorded launches one by one, in launch order (waits for their events; does NOT rewind the pool: call before collect):
 * duration in ms and kernel kind of up to `cap` launches.  Returns the number written, or a negative error code.  bench.py
 * uses it to show how the dominant kernel's duration moves through the timed region (clock ramp after idle). */
int ofmk_timing_durations(ofmk_timing *t, float *ms_per_launch, int *kind_per_launch /* may be NULL */, int cap);
void ofmk_timing_destroy(ofmk_timing *t);

#ifdef __cplusplus
}
#endif
#endif /* OFFMARK_HIP_H */
