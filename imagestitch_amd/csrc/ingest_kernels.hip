// ingest_kernels.hip -- one decode per file, both planes on the device (scope row f-1).
//
// The reference reads every file twice: cv2.imdecode(..., 0) for the registration loop (Stitcher.py:68-69) and cv2.imdecode(..., IMREAD_COLOR)
// for the mosaic when isColorMode is set (Stitcher.py:382-403, Main.py:14).  Both are views of ONE entropy decode: libjpeg hands out the
// component planes Y, Cb, Cr (upsampled); IMREAD_GRAYSCALE is the Y plane as it is (out_color_space = JCS_GRAYSCALE) and IMREAD_COLOR is
// jdcolor.c's fixed-point YCbCr -> RGB table arithmetic on the same planes, stored B, G, R.  So the host decodes a file ONCE to YCbCr
// (no colour conversion on the CPU at all) and this kernel writes the gray registration tile and the interleaved BGR canvas tile.
//
// jdcolor.c (build_ycc_rgb_table / ycc_rgb_convert), SCALEBITS = 16, ONE_HALF = 1 << 15, FIX(x) = (int)(x * 65536 + 0.5), x = c - 128:
//     R = clamp(Y + ((FIX(1.40200) * (Cr - 128) + ONE_HALF) >> 16))
//     B = clamp(Y + ((FIX(1.77200) * (Cb - 128) + ONE_HALF) >> 16))
//     G = clamp(Y + ((-FIX(0.34414) * (Cb - 128) + ONE_HALF - FIX(0.71414) * (Cr - 128)) >> 16))        (arithmetic shifts)
// libjpeg-turbo's SIMD converters produce the same bytes as this C code (they are tested against it upstream); the CPU tests hold a
// numpy restatement of the lines above to Pillow's own RGB decode of the reference's demo tiles, the GPU tests hold this kernel to both.
#include "common.h"

#define YCC_FIX_1_40200 91881
#define YCC_FIX_1_77200 116130
#define YCC_FIX_0_71414 46802
#define YCC_FIX_0_34414 22554

__device__ __forceinline__ unsigned ycc_to_bgr(unsigned y, unsigned cb, unsigned cr)
{
    const int b_ = (int)cb - 128, r_ = (int)cr - 128, Y = (int)y;
    int r = Y + ((YCC_FIX_1_40200 * r_ + 32768) >> 16);
    int b = Y + ((YCC_FIX_1_77200 * b_ + 32768) >> 16);
    int g = Y + ((-YCC_FIX_0_34414 * b_ + 32768 - YCC_FIX_0_71414 * r_) >> 16);
    r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
    return (unsigned)b | ((unsigned)g << 8) | ((unsigned)r << 16);
}

// FORMAT 0: one byte per pixel (a grayscale file: IMREAD_COLOR replicates it), 1: Y Cb Cr interleaved, 2: Y Cb Cr X (Pillow's own
// 4-byte pixel storage, handed over without a host-side repack).  Source, gray and BGR planes are densely packed, so the image is a
// flat run of n pixels: a lane converts four of them -- 4 / 3 / 1 source dwords in, 1 gray dword + 3 BGR dwords out, all aligned.
template <int FORMAT>
__global__ void __launch_bounds__(256) k_ingest_split(const uint8_t *__restrict__ src, uint8_t *__restrict__ gray, uint8_t *__restrict__ bgr, long long n)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // quad index
    const long long p0 = q * 4;
    if (p0 >= n) return;
    unsigned px[4];                                           // B | G << 8 | R << 16 | Y << 24
    if (p0 + 4 <= n) {
        if (FORMAT == 2) {
            const uint4 v = *(const uint4 *)(src + p0 * 4);
            const unsigned s[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) px[k] = ycc_to_bgr(s[k] & 255u, (s[k] >> 8) & 255u, (s[k] >> 16) & 255u) | (s[k] << 24);
        } else if (FORMAT == 1) {
            const unsigned *s = (const unsigned *)(src + p0 * 3);
            const unsigned a = s[0], b = s[1], c = s[2];      // Y0 Cb0 Cr0 Y1 | Cb1 Cr1 Y2 Cb2 | Cr2 Y3 Cb3 Cr3
            px[0] = ycc_to_bgr(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u) | (a << 24);
            px[1] = ycc_to_bgr(a >> 24, b & 255u, (b >> 8) & 255u) | (a & 0xff000000u);
            px[2] = ycc_to_bgr((b >> 16) & 255u, b >> 24, c & 255u) | ((b << 8) & 0xff000000u);
            px[3] = ycc_to_bgr((c >> 8) & 255u, (c >> 16) & 255u, c >> 24) | ((c << 16) & 0xff000000u);
        } else {
            const unsigned v = *(const unsigned *)(src + p0);
#pragma unroll
            for (int k = 0; k < 4; k++) { const unsigned g = (v >> (8 * k)) & 255u; px[k] = g * 0x01010101u; }
        }
        if (gray) *(unsigned *)(gray + p0) = (px[0] >> 24) | ((px[1] >> 24) << 8) | ((px[2] >> 24) << 16) | (px[3] & 0xff000000u);
        if (bgr) {
            unsigned *o = (unsigned *)(bgr + p0 * 3);
            o[0] = (px[0] & 0xffffffu) | (px[1] << 24);                       // B0 G0 R0 B1
            o[1] = ((px[1] >> 8) & 0xffffu) | (px[2] << 16);                  // G1 R1 B2 G2
            o[2] = ((px[2] >> 16) & 0xffu) | (px[3] << 8);                    // R2 B3 G3 R3
        }
        return;
    }
    for (long long p = p0; p < n; p++) {                      // the last 1..3 pixels of an image whose area is not a multiple of 4
        unsigned v;
        if (FORMAT == 2) v = ycc_to_bgr(src[p * 4], src[p * 4 + 1], src[p * 4 + 2]) | ((unsigned)src[p * 4] << 24);
        else if (FORMAT == 1) v = ycc_to_bgr(src[p * 3], src[p * 3 + 1], src[p * 3 + 2]) | ((unsigned)src[p * 3] << 24);
        else v = (unsigned)src[p] * 0x01010101u;
        if (gray) gray[p] = (uint8_t)(v >> 24);
        if (bgr) { bgr[p * 3] = (uint8_t)v; bgr[p * 3 + 1] = (uint8_t)(v >> 8); bgr[p * 3 + 2] = (uint8_t)(v >> 16); }
    }
}

int ingest_source_pixel_bytes(int format) { return format == 0 ? 1 : format == 1 ? 3 : format == 2 ? 4 : 0; }

// stream-ordered; src: n pixels densely packed in `format` on the device, gray / bgr: the tiles' buffers (either may be null)
int launch_ingest_split(hipStream_t stream, const uint8_t *src, uint8_t *gray, uint8_t *bgr, long long n, int format)
{
    const unsigned blocks = (unsigned)((((n + 3) / 4) + 255) / 256);
    if (format == 0) hipLaunchKernelGGL(k_ingest_split<0>, dim3(blocks), dim3(256), 0, stream, src, gray, bgr, n);
    else if (format == 1) hipLaunchKernelGGL(k_ingest_split<1>, dim3(blocks), dim3(256), 0, stream, src, gray, bgr, n);
    else if (format == 2) hipLaunchKernelGGL(k_ingest_split<2>, dim3(blocks), dim3(256), 0, stream, src, gray, bgr, n);
    else { vfsms_set_error("ingest: unknown source format %d", format); return VFSMS_ERR_BAD_ARG; }
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}
