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

// ---- round 6: a device stage INSIDE the JPEG decode ----------------------------------------------------------------------------------------
// The host stops behind the IDCT (jpeg_read_raw_data, csrc/jpeg_host.cpp): what arrives are the component planes of a 4:2:0 file as they are
// coded -- Y at full size, Cb and Cr at half size in both directions, all on the iMCU grid (pitch pw = w rounded up to 16).  This kernel
// is libjpeg's upsampler and colour converter in one pass:
//   jdsample.c h2v2_fancy_upsample (the default, do_fancy_upsampling = TRUE; libjpeg-turbo's SIMD routines produce the same bytes): a
//   triangle filter, 3/4 nearer + 1/4 further sample in each direction --
//       colsum(c) = 3 * near_row[c] + far_row[c]            near = row y >> 1, far = the row above (y even) / below (y odd)
//       out(2 c)     = (3 * colsum(c) + colsum(c - 1) + 8) >> 4
//       out(2 c + 1) = (3 * colsum(c) + colsum(c + 1) + 7) >> 4
//   with the image's first / last sample row and column standing in for the missing neighbour (jdmainct.c duplicates the edge rows; the
//   first / last column forms (4 * colsum + 8) >> 4 and (4 * colsum + 7) >> 4 are the formulas above with c - 1, c + 1 clamped);
//   then jdcolor.c's fixed-point conversion (ycc_to_bgr above) for the B G R tile, and Y as it is for the gray tile.
// The restatement is held to the library's own upsampled planes on the CPU (tests/test_host_logic.py) and the kernel to Pillow's two decodes
// on the GPU (test_tile_fill_jpeg_equals_the_two_decodes).  A lane converts four pixels of a row.
__global__ void __launch_bounds__(256) k_ingest_420(const uint8_t *__restrict__ src, int pw, int ph, int h, int w, uint8_t *__restrict__ gray, uint8_t *__restrict__ bgr)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= w || y >= h) return;
    const uint8_t *Yp = src, *Cb = src + (size_t)pw * ph, *Cr = Cb + (size_t)(pw >> 1) * (ph >> 1);
    const int cp = pw >> 1, dh = (h + 1) >> 1, dw = (w + 1) >> 1;
    const int rn = y >> 1, rf = (y & 1) ? min(rn + 1, dh - 1) : max(rn - 1, 0);
    const int c0 = x0 >> 1;
    const int ci[4] = { max(c0 - 1, 0), min(c0, dw - 1), min(c0 + 1, dw - 1), min(c0 + 2, dw - 1) };
    int sb[4], sr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        sb[k] = 3 * (int)Cb[(size_t)rn * cp + ci[k]] + (int)Cb[(size_t)rf * cp + ci[k]];
        sr[k] = 3 * (int)Cr[(size_t)rn * cp + ci[k]] + (int)Cr[(size_t)rf * cp + ci[k]];
    }
    // pixels x0 .. x0 + 3 sit on chroma columns c0, c0, c0 + 1, c0 + 1 (sb / sr index 1, 1, 2, 2); even x look left, odd x look right
    unsigned ub[4], ur[4];
    ub[0] = (unsigned)(3 * sb[1] + sb[0] + 8) >> 4; ub[1] = (unsigned)(3 * sb[1] + sb[2] + 7) >> 4;
    ub[2] = (unsigned)(3 * sb[2] + sb[1] + 8) >> 4; ub[3] = (unsigned)(3 * sb[2] + sb[3] + 7) >> 4;
    ur[0] = (unsigned)(3 * sr[1] + sr[0] + 8) >> 4; ur[1] = (unsigned)(3 * sr[1] + sr[2] + 7) >> 4;
    ur[2] = (unsigned)(3 * sr[2] + sr[1] + 8) >> 4; ur[3] = (unsigned)(3 * sr[2] + sr[3] + 7) >> 4;
    const unsigned yv = *(const unsigned *)(Yp + (size_t)y * pw + x0);          // (pw is a multiple of 16, x0 of 4: aligned, and inside the padded row)
    unsigned px[4];                                                             // B | G << 8 | R << 16 | Y << 24
#pragma unroll
    for (int k = 0; k < 4; k++) { const unsigned yy = (yv >> (8 * k)) & 255u; px[k] = ycc_to_bgr(yy, ub[k], ur[k]) | (yy << 24); }
    const size_t p0 = (size_t)y * w + x0;
    if ((w & 3) == 0) {
        if (gray) *(unsigned *)(gray + p0) = yv;
        if (bgr) {
            unsigned *o = (unsigned *)(bgr + p0 * 3);
            o[0] = (px[0] & 0xffffffu) | (px[1] << 24);
            o[1] = ((px[1] >> 8) & 0xffffu) | (px[2] << 16);
            o[2] = ((px[2] >> 16) & 0xffu) | (px[3] << 8);
        }
        return;
    }
    for (int k = 0; k < 4 && x0 + k < w; k++) {                                 // widths that are not multiples of 4: byte stores
        if (gray) gray[p0 + k] = (uint8_t)(px[k] >> 24);
        if (bgr) { bgr[(p0 + k) * 3] = (uint8_t)px[k]; bgr[(p0 + k) * 3 + 1] = (uint8_t)(px[k] >> 8); bgr[(p0 + k) * 3 + 2] = (uint8_t)(px[k] >> 16); }
    }
}

// stream-ordered; src: the raw planes of jpeg_decode_raw420_host on the device
int launch_ingest_420(hipStream_t stream, const uint8_t *src, int pw, int ph, int h, int w, uint8_t *gray, uint8_t *bgr)
{
    hipLaunchKernelGGL(k_ingest_420, dim3((unsigned)((w + 1023) / 1024), (unsigned)h), dim3(256), 0, stream, src, pw, ph, h, w, gray, bgr);
    HIP_TRY(hipGetLastError());
    return VFSMS_OK;
}
