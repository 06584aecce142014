/* abi_demo.c -- the C ABI of liboffmark_hip.so driven from plain C: no Python, no torch.
 *
 * What a host written in any language does through its FFI: allocate device buffers with the HIP runtime, hand
 * pointers and sizes to the library, read the payloads back.  Marks a batch of synthetic frames with the DCT codec and
 * with the DwtDctSvd codec, verifies each in the same call (ofmk_embed_detect_rgb8 / ofmk_svd_embed_detect_rgb8, the shape of
 * tests/mark_video_to_hls.py:356-389), then reads the marked frames again with the stand-alone detectors
 * (Extractor.__check_frame, src/offmark/video/extractor.py:30-34) and finishes DeShuffler.degenerate
 * (src/offmark/degenerator/de_shuffler.py:14-22) on the device.  The watermark is the 8-bit payload tiled over the
 * H*W/64 block bits -- Shuffler.generate_wm with the identity permutation (src/offmark/generator/shuffler.py:16-25).
 *
 * Build with the plain C compiler (tests/test_abi_and_host.py compiles it, tests/test_gpu_parity.py runs it):
 *   gcc -std=c11 examples/abi_demo.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L<dir of the .so> -L/opt/rocm/lib \
 *       -loffmark_hip -lamdhip64 -Wl,-rpath,<dir of the .so> -Wl,-rpath,/opt/rocm/lib -o abi_demo
 * Exit code 0 and a final line "abi_demo OK" when every frame's payload comes back.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "offmark_hip.h"

#define CHECK_HIP(e)                                                                      \
    do {                                                                                  \
        hipError_t r_ = (e);                                                              \
        if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } \
    } while (0)
#define CHECK_OFMK(e)                                                                     \
    do {                                                                                  \
        int r_ = (e);                                                                     \
        if (r_ != OFMK_OK) { fprintf(stderr, "%s -> %d: %s\n", #e, r_, ofmk_last_error()); return 3; } \
    } while (0)

enum { N_FRAMES = 6, H = 240, W = 320, L = 8 };
static const uint8_t PAYLOAD[L] = {0, 1, 1, 0, 0, 1, 0, 1};

/* smooth colour gradients plus a little noise: natural enough for both codecs */
static void make_frames(uint8_t *p) {
    uint32_t s = 12345u;
    for (int f = 0; f < N_FRAMES; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < 3; ++c) {
                    s = s * 1664525u + 1013904223u;
                    int v = 60 + (x * (c + 1)) / 4 % 120 + (y * (3 - c)) / 5 % 60 + 10 * f + (int)(s >> 29);
                    *p++ = (uint8_t)(v > 255 ? 255 : v);
                }
}

static int payloads_ok(const uint8_t *got, const char *what) {
    for (int f = 0; f < N_FRAMES; ++f)
        if (memcmp(got + (size_t)f * L, PAYLOAD, L) != 0) {
            fprintf(stderr, "%s: frame %d decodes to", what, f);
            for (int i = 0; i < L; ++i) fprintf(stderr, " %d", got[f * L + i]);
            fprintf(stderr, "\n");
            return 0;
        }
    printf("%-44s %d of %d frames carry the payload\n", what, N_FRAMES, N_FRAMES);
    return 1;
}

int main(void) {
    const int n_bits = H * W / 64;
    const size_t frame_bytes = (size_t)N_FRAMES * H * W * 3;
    printf("liboffmark_hip ABI version %d (header %d)\n", ofmk_version(), OFMK_ABI_VERSION);
    if (ofmk_version() != OFMK_ABI_VERSION) return 1;

    uint8_t *h_frames = (uint8_t *)malloc(frame_bytes), *h_wm = (uint8_t *)malloc(n_bits), h_payload[N_FRAMES * L];
    int32_t h_perm[L];
    make_frames(h_frames);
    for (int i = 0; i < n_bits; ++i) h_wm[i] = PAYLOAD[i % L];
    for (int i = 0; i < L; ++i) h_perm[i] = i;

    hipStream_t stream;
    uint8_t *d_in, *d_out, *d_wm, *d_payload;
    int32_t *d_counts, *d_perm;
    void *d_ws;
    const size_t ws_bytes = ofmk_workspace_bytes(N_FRAMES, H, W);
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc((void **)&d_in, frame_bytes));
    CHECK_HIP(hipMalloc((void **)&d_out, frame_bytes));
    CHECK_HIP(hipMalloc((void **)&d_wm, n_bits));
    CHECK_HIP(hipMalloc((void **)&d_payload, N_FRAMES * L));
    CHECK_HIP(hipMalloc((void **)&d_counts, N_FRAMES * L * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void **)&d_perm, L * sizeof(int32_t)));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpyAsync(d_in, h_frames, frame_bytes, hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(d_wm, h_wm, n_bits, hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(d_perm, h_perm, sizeof(h_perm), hipMemcpyHostToDevice, stream));

    int ok = 1;
    /* DCT codec: mark + verify in one call, then the stand-alone detector on the written frames */
    CHECK_OFMK(ofmk_embed_detect_rgb8(d_in, d_out, N_FRAMES, H, W, d_wm, 1, NULL, 20.0, L, d_counts, NULL, 0, d_ws, ws_bytes, stream, NULL));
    CHECK_OFMK(ofmk_payloads_from_counts(d_counts, N_FRAMES, L, n_bits, d_perm, d_payload, stream, NULL));
    CHECK_HIP(hipMemcpyAsync(h_payload, d_payload, sizeof(h_payload), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    ok &= payloads_ok(h_payload, "DCT       ofmk_embed_detect_rgb8");
    CHECK_OFMK(ofmk_detect_rgb8(d_out, N_FRAMES, H, W, L, 20.0, d_counts, NULL, 0, d_ws, ws_bytes, stream, NULL));
    CHECK_OFMK(ofmk_payloads_from_counts(d_counts, N_FRAMES, L, n_bits, d_perm, d_payload, stream, NULL));
    CHECK_HIP(hipMemcpyAsync(h_payload, d_payload, sizeof(h_payload), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    ok &= payloads_ok(h_payload, "DCT       ofmk_detect_rgb8 on the output");

    /* DwtDctSvd codec (what tests/mark.py constructs), scales = [0, 15, 0] */
    const double scales[3] = {0.0, 15.0, 0.0};
    CHECK_OFMK(ofmk_svd_embed_detect_rgb8(d_in, d_out, N_FRAMES, H, W, d_wm, 1, NULL, scales, 4, L, d_counts, NULL, stream, NULL));
    CHECK_OFMK(ofmk_payloads_from_counts(d_counts, N_FRAMES, L, n_bits, d_perm, d_payload, stream, NULL));
    CHECK_HIP(hipMemcpyAsync(h_payload, d_payload, sizeof(h_payload), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    ok &= payloads_ok(h_payload, "DwtDctSvd ofmk_svd_embed_detect_rgb8");
    CHECK_OFMK(ofmk_svd_detect_rgb8(d_out, N_FRAMES, H, W, L, scales, 4, d_counts, NULL, stream, NULL));
    CHECK_OFMK(ofmk_payloads_from_counts(d_counts, N_FRAMES, L, n_bits, d_perm, d_payload, stream, NULL));
    CHECK_HIP(hipMemcpyAsync(h_payload, d_payload, sizeof(h_payload), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    ok &= payloads_ok(h_payload, "DwtDctSvd ofmk_svd_detect_rgb8 on the output");

    /* the same codec with blk = 8 (16x16 pixel tiles): H*W/256 bits per frame, tile c carries wm[c] */
    CHECK_OFMK(ofmk_svd_embed_detect_rgb8(d_in, d_out, N_FRAMES, H, W, d_wm, 1, NULL, scales, 8, L, d_counts, NULL, stream, NULL));
    CHECK_OFMK(ofmk_payloads_from_counts(d_counts, N_FRAMES, L, H * W / 256, d_perm, d_payload, stream, NULL));
    CHECK_HIP(hipMemcpyAsync(h_payload, d_payload, sizeof(h_payload), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    ok &= payloads_ok(h_payload, "DwtDctSvd blk=8 ofmk_svd_embed_detect_rgb8");
    CHECK_OFMK(ofmk_svd_detect_rgb8(d_out, N_FRAMES, H, W, L, scales, 8, d_counts, NULL, stream, NULL));
    CHECK_OFMK(ofmk_payloads_from_counts(d_counts, N_FRAMES, L, H * W / 256, d_perm, d_payload, stream, NULL));
    CHECK_HIP(hipMemcpyAsync(h_payload, d_payload, sizeof(h_payload), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    ok &= payloads_ok(h_payload, "DwtDctSvd blk=8 ofmk_svd_detect_rgb8 on the output");

    /* errors are return codes with a per-thread text, never aborts */
    const int rc = ofmk_detect_rgb8(d_out, N_FRAMES, 4, W, L, 20.0, d_counts, NULL, 0, d_ws, ws_bytes, stream, NULL);
    printf("a frame height of 4 is refused with code %d: %s\n", rc, ofmk_last_error());
    ok &= rc == OFMK_E_ARG;

    hipFree(d_ws); hipFree(d_perm); hipFree(d_counts); hipFree(d_payload); hipFree(d_wm); hipFree(d_out); hipFree(d_in);
    hipStreamDestroy(stream);
    free(h_wm); free(h_frames);
    puts(ok ? "abi_demo OK" : "abi_demo FAILED");
    return ok ? 0 : 1;
}
