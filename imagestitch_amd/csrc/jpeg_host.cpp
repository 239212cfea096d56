// jpeg_host.cpp -- host-side JPEG decode straight into pinned staging, for the ingest pipeline (scope row f-1: "host libjpeg-turbo + pinned
// upload").  No kernels here.
//
// The decoder pool of the Python host (Pillow) stops scaling at ~16 threads on the 256-thread host of the MI355X box: every decode maps
// and faults in 4-17 MB of fresh memory and crosses the interpreter lock a few times, and the per-tile decode time grows with the
// parallelism.  This entry decodes with the system's libjpeg-turbo (libjpeg.so.8, ABI version 80: loaded with dlopen at first use, no
// link-time dependency and no header -- the image ships the library without its headers) into a pinned buffer that is REUSED from call to
// call, from any thread, without the interpreter lock (ctypes releases it), and hands the planes to the same device path as
// vfsms_tile_fill_pair.  Grayscale wanted: out_color_space = JCS_GRAYSCALE, the Y plane (cv2.imdecode(..., 0), Stitcher.py:68-69).  Colour
// wanted: out_color_space = JCS_YCbCr, the upsampled planes interleaved, no colour conversion on the host; the device derives the gray and
// the B G R tile (csrc/ingest_kernels.hip).
//
// The other direction, for the mosaic (cv2.imwrite at Stitcher.py:149, 175-179; Main.py writes every result as .jpg): vfsms_jpeg_encode
// turns a horizontal STRIPE of an image into a complete baseline JPEG stream with cv2.imwrite's settings (libjpeg defaults: 4:2:0, standard
// Huffman tables, quality 95 unless told otherwise).  Stripes whose heights are multiples of the MCU height share no state -- the DCT
// blocks, the 2x2 chroma box filter and the quantisation are local to an MCU row -- so the host encodes them on as many threads as it
// likes and joins their entropy-coded segments into ONE file as restart intervals (vfsms_jpeg_join: DRI = the MCUs of a stripe, an RSTn
// marker between two stripes).  The coefficients, hence the decoded pixels, are those of the one-thread encode of the whole image.
//
// ABI knowledge used (libjpeg 8 / libjpeg-turbo 2.x on x86-64, `boolean` = int): the public head of jpeg_decompress_struct up to
// output_scanline and the layout of jpeg_error_mgr -- restated below field by field.  Two guards make a mismatch a clean refusal
// (VFSMS_ERR_UNSUPPORTED -> the host falls back to Pillow) instead of a wrong image: the struct SIZE is taken from the library itself
// (jpeg_CreateDecompress is first called with a wrong size; its JERR_BAD_STRUCT_SIZE error carries the size the library expects), and
// after jpeg_read_header the image size read through these offsets must equal the size this file parses from the SOF marker itself.
#include "common.h"
#include <dlfcn.h>
#include <setjmp.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>

namespace {

// ---- the ABI, restated ----------------------------------------------------------------------------------------------------------------
struct jerr_abi {                       // struct jpeg_error_mgr
    void (*error_exit)(void *);
    void (*emit_message)(void *, int);
    void (*output_message)(void *);
    void (*format_message)(void *, char *);
    void (*reset_error_mgr)(void *);
    int msg_code;
    union { int i[8]; char s[80]; } msg_parm;
    int trace_level;
    long num_warnings;
    const char *const *jpeg_message_table;
    int last_jpeg_message;
    const char *const *addon_message_table;
    int first_addon_message, last_addon_message;
};
struct jdec_head_abi {                  // struct jpeg_decompress_struct, its public head (all that is touched here)
    jerr_abi *err; void *mem; void *progress; void *client_data; int is_decompressor; int global_state;      // jpeg_common_fields
    void *src;
    unsigned image_width, image_height; int num_components; int jpeg_color_space;
    int out_color_space; unsigned scale_num, scale_denom;
    double output_gamma;
    int buffered_image, raw_data_out, dct_method, do_fancy_upsampling, do_block_smoothing;
    int quantize_colors, dither_mode, two_pass_quantize, desired_number_of_colors, enable_1pass_quant, enable_external_quant, enable_2pass_quant;
    unsigned output_width, output_height; int out_color_components, output_components, rec_outbuf_height;
    int actual_number_of_colors; void *colormap;
    unsigned output_scanline;
};
static_assert(offsetof(jdec_head_abi, image_width) == 48 && offsetof(jdec_head_abi, out_color_space) == 64 && offsetof(jdec_head_abi, output_gamma) == 80 &&
              offsetof(jdec_head_abi, output_width) == 136 && offsetof(jdec_head_abi, output_components) == 148 && offsetof(jdec_head_abi, output_scanline) == 168,
              "jpeg_decompress_struct head: unexpected layout");
static_assert(offsetof(jerr_abi, msg_parm) == 44 && offsetof(jerr_abi, num_warnings) == 128 && sizeof(jerr_abi) == 168, "jpeg_error_mgr: unexpected layout");
struct jcomp_head_abi {                 // struct jpeg_compress_struct, its public head
    jerr_abi *err; void *mem; void *progress; void *client_data; int is_decompressor; int global_state;
    void *dest;
    unsigned image_width, image_height; int input_components; int in_color_space;
    double input_gamma;
};
static_assert(offsetof(jcomp_head_abi, image_width) == 48 && offsetof(jcomp_head_abi, in_color_space) == 60, "jpeg_compress_struct head: unexpected layout");
enum { JCS_GRAYSCALE_ = 1, JCS_RGB_ = 2, JCS_YCbCr_ = 3 };
#define JPEG_ABI_VERSION 80
#define JPEG_CINFO_BYTES 2048           // room for the whole struct (632 bytes in libjpeg-turbo 2.1, ABI 8)

struct JpegApi {
    jerr_abi *(*std_error)(jerr_abi *);
    void (*create)(void *, int, size_t);
    void (*mem_src)(void *, const unsigned char *, unsigned long);
    int (*read_header)(void *, int);
    int (*start)(void *);
    unsigned (*read_scanlines)(void *, unsigned char **, unsigned);
    unsigned (*read_raw)(void *, unsigned char ***, unsigned);          // jpeg_read_raw_data (may be absent: the raw 4:2:0 path is then off)
    int (*finish)(void *);
    void (*destroy)(void *);
    size_t cinfo_size;                  // as the library reports it
    bool ok;
    // the compressor
    void (*c_create)(void *, int, size_t);
    void (*c_mem_dest)(void *, unsigned char **, unsigned long *);
    void (*c_defaults)(void *);
    void (*c_quality)(void *, int, int);
    void (*c_start)(void *, int);
    unsigned (*c_write)(void *, unsigned char **, unsigned);
    void (*c_finish)(void *);
    void (*c_destroy)(void *);
    size_t c_size;
    bool c_ok;
};
struct Guard { jmp_buf jb; int code; int parm0, parm1; char text[200]; };

void on_error(void *cinfo)                 // (compressor or decompressor: err and client_data are jpeg_common_fields of both)
{
    jdec_head_abi *c = (jdec_head_abi *)cinfo;
    Guard *g = (Guard *)c->client_data;
    g->code = c->err->msg_code; g->parm0 = c->err->msg_parm.i[0]; g->parm1 = c->err->msg_parm.i[1];
    g->text[0] = 0;
    if (c->err->format_message) { char buf[256]; buf[0] = 0; c->err->format_message(cinfo, buf); strncpy(g->text, buf, sizeof(g->text) - 1); g->text[sizeof(g->text) - 1] = 0; }
    longjmp(g->jb, 1);
}
void on_message(void *) {}              // warnings (e.g. "extraneous bytes before marker") are not printed

const JpegApi &api()
{
    static JpegApi A = [] {
        JpegApi a; memset(&a, 0, sizeof(a));
        void *h = dlopen("libjpeg.so.8", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.std_error = (jerr_abi * (*)(jerr_abi *)) dlsym(h, "jpeg_std_error");
        a.create = (void (*)(void *, int, size_t))dlsym(h, "jpeg_CreateDecompress");
        a.mem_src = (void (*)(void *, const unsigned char *, unsigned long))dlsym(h, "jpeg_mem_src");
        a.read_header = (int (*)(void *, int))dlsym(h, "jpeg_read_header");
        a.start = (int (*)(void *))dlsym(h, "jpeg_start_decompress");
        a.read_scanlines = (unsigned (*)(void *, unsigned char **, unsigned))dlsym(h, "jpeg_read_scanlines");
        a.read_raw = (unsigned (*)(void *, unsigned char ***, unsigned))dlsym(h, "jpeg_read_raw_data");
        a.finish = (int (*)(void *))dlsym(h, "jpeg_finish_decompress");
        a.destroy = (void (*)(void *))dlsym(h, "jpeg_destroy_decompress");
        if (!a.std_error || !a.create || !a.mem_src || !a.read_header || !a.start || !a.read_scanlines || !a.finish || !a.destroy) return a;
        // the struct size, from the library: a create call with a size no struct has fails with JERR_BAD_STRUCT_SIZE(library's, caller's)
        alignas(16) unsigned char cinfo[JPEG_CINFO_BYTES]; memset(cinfo, 0, sizeof(cinfo));
        jerr_abi err; memset(&err, 0, sizeof(err));
        Guard g; memset(&g.code, 0, sizeof(g) - offsetof(Guard, code));
        jdec_head_abi *c = (jdec_head_abi *)cinfo;
        c->err = a.std_error(&err);
        err.error_exit = on_error; err.output_message = on_message;
        c->client_data = &g;
        if (setjmp(g.jb) == 0) { a.create(cinfo, JPEG_ABI_VERSION, 7); return a; }       // (a size of 7 cannot be right: the error branch is the expected one)
        if (g.parm1 != 7 || g.parm0 < (int)sizeof(jdec_head_abi) || g.parm0 > JPEG_CINFO_BYTES) return a;    // not the size error, or an implausible size
        a.cinfo_size = (size_t)g.parm0;
        a.ok = true;
        a.c_create = (void (*)(void *, int, size_t))dlsym(h, "jpeg_CreateCompress");
        a.c_mem_dest = (void (*)(void *, unsigned char **, unsigned long *))dlsym(h, "jpeg_mem_dest");
        a.c_defaults = (void (*)(void *))dlsym(h, "jpeg_set_defaults");
        a.c_quality = (void (*)(void *, int, int))dlsym(h, "jpeg_set_quality");
        a.c_start = (void (*)(void *, int))dlsym(h, "jpeg_start_compress");
        a.c_write = (unsigned (*)(void *, unsigned char **, unsigned))dlsym(h, "jpeg_write_scanlines");
        a.c_finish = (void (*)(void *))dlsym(h, "jpeg_finish_compress");
        a.c_destroy = (void (*)(void *))dlsym(h, "jpeg_destroy_compress");
        if (!a.c_create || !a.c_mem_dest || !a.c_defaults || !a.c_quality || !a.c_start || !a.c_write || !a.c_finish || !a.c_destroy) return a;
        memset(cinfo, 0, sizeof(cinfo)); memset(&err, 0, sizeof(err)); memset(&g.code, 0, sizeof(g) - offsetof(Guard, code));
        c->err = a.std_error(&err);
        err.error_exit = on_error; err.output_message = on_message;
        c->client_data = &g;
        if (setjmp(g.jb) == 0) { a.c_create(cinfo, JPEG_ABI_VERSION, 7); return a; }
        if (g.parm1 != 7 || g.parm0 < (int)sizeof(jcomp_head_abi) || g.parm0 > JPEG_CINFO_BYTES) return a;
        a.c_size = (size_t)g.parm0;
        a.c_ok = true;
        return a;
    }();
    return A;
}

// (rows, cols, components) from the first SOFn marker: the independent witness for the struct offsets
bool sof_size(const unsigned char *p, size_t n, int *h, int *w, int *nc, int *samp = nullptr)
{
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) return false;
    size_t q = 2;
    while (q + 9 < n) {
        if (p[q] != 0xFF) return false;
        const unsigned m = p[q + 1];
        if (m == 0xFF) { q++; continue; }
        if ((m >= 0xD0 && m <= 0xD9) || m == 0x01) { q += 2; continue; }
        const size_t seg = ((size_t)p[q + 2] << 8) | p[q + 3];
        if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            *h = (p[q + 5] << 8) | p[q + 6]; *w = (p[q + 7] << 8) | p[q + 8]; *nc = p[q + 9];
            if (samp) {                                       // (H << 4 | V) of up to three components
                samp[0] = samp[1] = samp[2] = 0;
                for (int k = 0; k < *nc && k < 3 && q + 10 + 3 * k + 2 < n; k++) samp[k] = p[q + 10 + 3 * k + 1];
            }
            return *h > 0 && *w > 0;
        }
        q += 2 + seg;
    }
    return false;
}

}  // namespace

// Decode `jpeg` into `out` (rows of *w_out * (*comp_out) bytes, densely packed).  want_planes != 0 and a 3-component YCbCr file: the Y Cb Cr
// planes interleaved (comp 3); otherwise the grayscale decode (comp 1).  Returns VFSMS_ERR_UNSUPPORTED when the library or the file is
// not one this path handles (the caller decodes some other way), VFSMS_ERR_CAPACITY when `cap` is too small (h, w, comp are set).
int jpeg_decode_host(const unsigned char *jpeg, size_t nbytes, int want_planes, unsigned char *out, size_t cap, int *h_out, int *w_out, int *comp_out)
{
    const JpegApi &A = api();
    if (!A.ok) { vfsms_set_error("jpeg: libjpeg.so.8 (ABI 8) is not available on this host"); return VFSMS_ERR_UNSUPPORTED; }
    int sh = 0, sw = 0, snc = 0;
    if (!jpeg || !sof_size(jpeg, nbytes, &sh, &sw, &snc) || (snc != 1 && snc != 3)) { vfsms_set_error("jpeg: not a 1- or 3-component JPEG"); return VFSMS_ERR_UNSUPPORTED; }
    alignas(16) unsigned char cinfo[JPEG_CINFO_BYTES]; memset(cinfo, 0, sizeof(cinfo));
    jerr_abi err; memset(&err, 0, sizeof(err));
    Guard g; memset(&g.code, 0, sizeof(g) - offsetof(Guard, code));
    jdec_head_abi *c = (jdec_head_abi *)cinfo;
    volatile bool created = false;
    if (setjmp(g.jb)) {
        if (created) A.destroy(cinfo);
        vfsms_set_error("jpeg: %s", g.text[0] ? g.text : "decode error");
        return VFSMS_ERR_BAD_ARG;                            // a damaged file: an error of the input, not of the path
    }
    c->err = A.std_error(&err);
    err.error_exit = on_error; err.output_message = on_message;
    c->client_data = &g;
    A.create(cinfo, JPEG_ABI_VERSION, A.cinfo_size);
    created = true;
    c->client_data = &g;                                     // (create zeroes the struct but for err and client_data; set again to be sure)
    A.mem_src(cinfo, jpeg, (unsigned long)nbytes);
    A.read_header(cinfo, 1);
    if ((int)c->image_height != sh || (int)c->image_width != sw || c->num_components != snc) {      // the offsets are not this library's
        A.destroy(cinfo);
        vfsms_set_error("jpeg: libjpeg.so.8 does not have the expected struct layout");
        return VFSMS_ERR_UNSUPPORTED;
    }
    const bool planes = want_planes && snc == 3 && c->jpeg_color_space == JCS_YCbCr_;
    if (want_planes && snc == 3 && !planes) {                // an RGB / Adobe-transform-0 file has no Y Cb Cr planes: let the caller decode it
        A.destroy(cinfo);
        vfsms_set_error("jpeg: 3-component file that is not YCbCr");
        return VFSMS_ERR_UNSUPPORTED;
    }
    c->out_color_space = planes ? JCS_YCbCr_ : JCS_GRAYSCALE_;
    const int comp = planes ? 3 : 1;
    *h_out = sh; *w_out = sw; *comp_out = comp;
    if (!out || cap < (size_t)sh * sw * comp) { A.destroy(cinfo); vfsms_set_error("jpeg: output buffer too small"); return VFSMS_ERR_CAPACITY; }
    A.start(cinfo);
    if ((int)c->output_width != sw || (int)c->output_height != sh || c->output_components != comp) {
        A.destroy(cinfo);
        vfsms_set_error("jpeg: unexpected output geometry");
        return VFSMS_ERR_UNSUPPORTED;
    }
    const size_t pitch = (size_t)sw * comp;
    while (c->output_scanline < c->output_height) {
        unsigned char *rows[16];
        const unsigned y0 = c->output_scanline;
        const unsigned nr = c->output_height - y0 < 16 ? c->output_height - y0 : 16;
        for (unsigned r = 0; r < nr; r++) rows[r] = out + (size_t)(y0 + r) * pitch;
        if (A.read_scanlines(cinfo, rows, nr) == 0) break;
    }
    const bool complete = c->output_scanline == c->output_height;
    A.finish(cinfo);
    const long warnings = err.num_warnings;                  // e.g. "premature end of data segment": libjpeg pads a truncated file with gray
    A.destroy(cinfo);
    if (!complete || warnings) { vfsms_set_error("jpeg: truncated or damaged file (%ld decoder warnings)", warnings); return VFSMS_ERR_BAD_ARG; }
    return VFSMS_OK;
}

// Round 6: the DOWNSAMPLED planes of a 4:2:0 Y Cb Cr file (jpeg_read_raw_data: entropy decode + IDCT on the host, nothing else) --
// chroma upsampling and the colour conversion run on the device (csrc/ingest_kernels.hip: k_ingest_420, libjpeg's h2v2 "fancy" triangle
// filter restated).  The host writes 1.5 bytes per pixel into pinned staging instead of 3 and skips its upsampling and interleaving passes.
// out: Y plane, pitch pw = w rounded up to 16, ph = h rounded up to 16 rows (the iMCU grid: the rows / columns beyond the image hold the
// decoder's edge padding and are never looked at), then the Cb plane (pitch pw / 2, ph / 2 rows), then Cr.  VFSMS_ERR_UNSUPPORTED: not a
// 3-component Y Cb Cr file sampled 2x2, 1x1, 1x1, or no jpeg_read_raw_data in the library -- the caller takes the full decode.
int jpeg_decode_raw420_host(const unsigned char *jpeg, size_t nbytes, unsigned char *out, size_t cap, int *h_out, int *w_out)
{
    const JpegApi &A = api();
    if (!A.ok || !A.read_raw) { vfsms_set_error("jpeg: no raw-data decode on this host"); return VFSMS_ERR_UNSUPPORTED; }
    int sh = 0, sw = 0, snc = 0, samp[3] = {0, 0, 0};
    if (!jpeg || !sof_size(jpeg, nbytes, &sh, &sw, &snc, samp) || snc != 3 || samp[0] != 0x22 || samp[1] != 0x11 || samp[2] != 0x11 || sw < 4 || sh < 2) {
        vfsms_set_error("jpeg: not a 4:2:0 Y Cb Cr file"); return VFSMS_ERR_UNSUPPORTED;
    }
    const size_t pw = ((size_t)sw + 15) & ~(size_t)15, ph = ((size_t)sh + 15) & ~(size_t)15;
    *h_out = sh; *w_out = sw;
    if (!out || cap < pw * ph * 3 / 2) { vfsms_set_error("jpeg: output buffer too small"); return VFSMS_ERR_CAPACITY; }
    alignas(16) unsigned char cinfo[JPEG_CINFO_BYTES]; memset(cinfo, 0, sizeof(cinfo));
    jerr_abi err; memset(&err, 0, sizeof(err));
    Guard g; memset(&g.code, 0, sizeof(g) - offsetof(Guard, code));
    jdec_head_abi *c = (jdec_head_abi *)cinfo;
    volatile bool created = false;
    if (setjmp(g.jb)) {
        if (created) A.destroy(cinfo);
        vfsms_set_error("jpeg: %s", g.text[0] ? g.text : "decode error");
        return VFSMS_ERR_BAD_ARG;
    }
    c->err = A.std_error(&err);
    err.error_exit = on_error; err.output_message = on_message;
    c->client_data = &g;
    A.create(cinfo, JPEG_ABI_VERSION, A.cinfo_size);
    created = true;
    c->client_data = &g;
    A.mem_src(cinfo, jpeg, (unsigned long)nbytes);
    A.read_header(cinfo, 1);
    if ((int)c->image_height != sh || (int)c->image_width != sw || c->num_components != 3 || c->jpeg_color_space != JCS_YCbCr_) {
        A.destroy(cinfo);
        vfsms_set_error("jpeg: not a Y Cb Cr file (or an unexpected struct layout)");
        return VFSMS_ERR_UNSUPPORTED;
    }
    c->out_color_space = JCS_YCbCr_;
    c->raw_data_out = 1;
    A.start(cinfo);
    if ((int)c->output_width != sw || (int)c->output_height != sh) { A.destroy(cinfo); vfsms_set_error("jpeg: unexpected output geometry"); return VFSMS_ERR_UNSUPPORTED; }
    unsigned char *Y = out, *Cb = out + pw * ph, *Cr = Cb + (pw / 2) * (ph / 2);
    bool ok = true;
    while (c->output_scanline < c->output_height) {
        const unsigned y0 = c->output_scanline;                  // a multiple of 16: one iMCU row per call
        if (y0 % 16 || y0 + 16 > ph) { ok = false; break; }
        unsigned char *yr[16], *br[8], *rr[8];
        for (unsigned r = 0; r < 16; r++) yr[r] = Y + (size_t)(y0 + r) * pw;
        for (unsigned r = 0; r < 8; r++) { br[r] = Cb + (size_t)(y0 / 2 + r) * (pw / 2); rr[r] = Cr + (size_t)(y0 / 2 + r) * (pw / 2); }
        unsigned char **planes[3] = { yr, br, rr };
        if (A.read_raw(cinfo, planes, 16) == 0) { ok = false; break; }
    }
    const bool complete = ok && c->output_scanline >= c->output_height;
    if (complete) A.finish(cinfo);
    const long warnings = err.num_warnings;
    A.destroy(cinfo);
    if (!complete || warnings) { vfsms_set_error("jpeg: truncated or damaged file (%ld decoder warnings)", warnings); return VFSMS_ERR_BAD_ARG; }
    return VFSMS_OK;
}

// host-only entry (no context, no GPU): the decode of jpeg_decode_host, for tests and for callers that want the planes themselves
extern "C" int vfsms_jpeg_decode(const uint8_t *jpeg, size_t nbytes, int want_planes, uint8_t *out, size_t cap, int *h, int *w, int *comp)
{
    if (!h || !w || !comp) { vfsms_set_error("jpeg_decode: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    if (want_planes == 2) {                                  // the raw 4:2:0 planes (comp = 420): Y | Cb | Cr on the iMCU grid, see jpeg_decode_raw420_host
        *comp = 420;
        return jpeg_decode_raw420_host(jpeg, nbytes, out, cap, h, w);
    }
    return jpeg_decode_host(jpeg, nbytes, want_planes, out, cap, h, w, comp);
}

// ---- encode -----------------------------------------------------------------------------------------------------------------------------
namespace {
// the segments of a baseline stream this library wrote: [2, sos_end) = the headers up to and including the SOS header, [sos_end, n - 2) =
// the entropy-coded segment, and where the frame header keeps its size
struct StreamMap { size_t sof, sos, sos_end; int h, w, nc, vmax; };
bool map_stream(const unsigned char *p, size_t n, StreamMap *m)
{
    if (n < 6 || p[0] != 0xFF || p[1] != 0xD8 || p[n - 2] != 0xFF || p[n - 1] != 0xD9) return false;
    size_t q = 2; m->sof = 0;
    while (q + 4 <= n) {
        if (p[q] != 0xFF) return false;
        const unsigned mk = p[q + 1];
        const size_t seg = ((size_t)p[q + 2] << 8) | p[q + 3];
        if (q + 2 + seg > n) return false;
        if (mk == 0xC0) {                                        // baseline frame header
            if (seg < 8 + 3) return false;
            m->sof = q; m->h = (p[q + 5] << 8) | p[q + 6]; m->w = (p[q + 7] << 8) | p[q + 8]; m->nc = p[q + 9];
            if (seg < 8 + 3u * m->nc) return false;
            m->vmax = 1;
            for (int c = 0; c < m->nc; c++) { const int v = p[q + 11 + 3 * c] & 15; if (v > m->vmax) m->vmax = v; }
        } else if (mk == 0xDD) return false;                      // a stripe is ONE restart interval: it must not bring a DRI of its own
        else if (mk == 0xDA) { m->sos = q; m->sos_end = q + 2 + seg; return m->sof != 0 && m->sos_end <= n - 2; }
        q += 2 + seg;
    }
    return false;
}
}  // namespace

namespace {
enum { IN_RGB = 0, IN_BGR_SWAP = 1, IN_BGR_EXT = 2 };
#define JCS_EXT_BGR_ 8                  // libjpeg-turbo's extension: B G R input, converted by the same SIMD routine as R G B
// one image -> a malloc'ed stream (*mem, *memsize; the caller frees it)
int encode_impl(const uint8_t *rows, int n_rows, int cols, int channels, int stride, int input, int quality, unsigned char **mem, unsigned long *memsize)
{
    const JpegApi &A = api();
    alignas(16) unsigned char cinfo[JPEG_CINFO_BYTES]; memset(cinfo, 0, sizeof(cinfo));
    jerr_abi err; memset(&err, 0, sizeof(err));
    Guard g; memset(&g.code, 0, sizeof(g) - offsetof(Guard, code));
    jcomp_head_abi *c = (jcomp_head_abi *)cinfo;
    unsigned char *volatile swap = nullptr;
    volatile bool created = false;
    *mem = nullptr; *memsize = 0;                             // (jpeg_mem_dest allocates and grows the buffer with malloc)
    if (setjmp(g.jb)) {
        if (created) A.c_destroy(cinfo);
        free(*mem); *mem = nullptr; free(swap);
        vfsms_set_error("jpeg_encode: %s", g.text[0] ? g.text : "encoder error");
        return VFSMS_ERR_BAD_ARG;
    }
    c->err = A.std_error(&err);
    err.error_exit = on_error; err.output_message = on_message;
    c->client_data = &g;
    A.c_create(cinfo, JPEG_ABI_VERSION, A.c_size);
    created = true;
    c->client_data = &g;
    A.c_mem_dest(cinfo, mem, memsize);
    c->image_width = (unsigned)cols; c->image_height = (unsigned)n_rows; c->input_components = channels;
    c->in_color_space = channels == 1 ? JCS_GRAYSCALE_ : input == IN_BGR_EXT ? JCS_EXT_BGR_ : JCS_RGB_;
    A.c_defaults(cinfo);
    A.c_quality(cinfo, quality, 1);
    A.c_start(cinfo, 1);
    const int CH = 16;
    if (channels == 3 && input == IN_BGR_SWAP) {
        swap = (unsigned char *)malloc((size_t)CH * cols * 3);
        if (!swap) { A.c_destroy(cinfo); free(*mem); *mem = nullptr; vfsms_set_error("jpeg_encode: out of memory"); return VFSMS_ERR_CAPACITY; }
    }
    for (int y0 = 0; y0 < n_rows; y0 += CH) {
        const int nr = n_rows - y0 < CH ? n_rows - y0 : CH;
        unsigned char *ptr[CH];
        for (int r = 0; r < nr; r++) {
            const uint8_t *src = rows + (size_t)(y0 + r) * stride;
            if (swap) {
                unsigned char *d = swap + (size_t)r * cols * 3;
                for (int x = 0; x < cols; x++) { d[3 * x] = src[3 * x + 2]; d[3 * x + 1] = src[3 * x + 1]; d[3 * x + 2] = src[3 * x]; }
                ptr[r] = d;
            } else ptr[r] = (unsigned char *)src;
        }
        int done = 0;
        while (done < nr) {
            const unsigned k = A.c_write(cinfo, ptr + done, (unsigned)(nr - done));
            if (k == 0) { g.text[0] = 0; longjmp(g.jb, 1); }      // (cannot happen with a memory destination)
            done += (int)k;
        }
    }
    A.c_finish(cinfo);
    A.c_destroy(cinfo);
    created = false;
    free(swap); swap = nullptr;
    return VFSMS_OK;
}
// does this libjpeg take B G R rows directly (libjpeg-turbo does), with the very bytes of the swapped encode?  Probed once.
bool ext_bgr_ok()
{
    static std::atomic<int> state{0};
    int st = state.load();
    if (st == 0) {
        uint8_t img[16 * 16 * 3];
        for (int k = 0; k < 16 * 16 * 3; k++) img[k] = (uint8_t)((k * 37 + (k / 48) * 11) & 255);
        unsigned char *m1 = nullptr, *m2 = nullptr; unsigned long n1 = 0, n2 = 0;
        const int r1 = encode_impl(img, 16, 16, 3, 48, IN_BGR_SWAP, 90, &m1, &n1);
        const int r2 = encode_impl(img, 16, 16, 3, 48, IN_BGR_EXT, 90, &m2, &n2);
        st = (r1 == VFSMS_OK && r2 == VFSMS_OK && n1 == n2 && n1 > 0 && memcmp(m1, m2, n1) == 0) ? 1 : -1;
        free(m1); free(m2);
        state.store(st);
    }
    return st == 1;
}
}  // namespace

// `rows`: n_rows x cols pixels of `channels` (1: gray, 3: R G B, or B G R with bgr != 0 -- the canvas order), `stride` bytes apart -> a
// complete baseline JPEG in `out` (cap bytes; *nbytes = the size needed, also on VFSMS_ERR_CAPACITY).  libjpeg defaults + `quality`: what
// cv2.imwrite(path, img) writes (quality 95).
extern "C" int vfsms_jpeg_encode(const uint8_t *rows, int n_rows, int cols, int channels, int stride, int bgr, int quality, uint8_t *out, size_t cap, size_t *nbytes)
{
    const JpegApi &A = api();
    if (!A.c_ok) { vfsms_set_error("jpeg_encode: libjpeg.so.8 (ABI 8) is not available on this host"); return VFSMS_ERR_UNSUPPORTED; }
    if (!rows || !nbytes || n_rows <= 0 || cols <= 0 || n_rows > 65500 || cols > 65500 || (channels != 1 && channels != 3) || stride < cols * channels || quality < 1 || quality > 100) {
        vfsms_set_error("jpeg_encode: bad arguments (1 or 3 channels, at most 65500 pixels a side)"); return VFSMS_ERR_BAD_ARG;
    }
    const int input = (channels == 3 && bgr) ? (ext_bgr_ok() ? IN_BGR_EXT : IN_BGR_SWAP) : IN_RGB;
    unsigned char *mem = nullptr; unsigned long memsize = 0;
    int rc = encode_impl(rows, n_rows, cols, channels, stride, input, quality, &mem, &memsize);
    if (rc != VFSMS_OK) return rc;
    StreamMap m;
    const bool sane = mem && map_stream(mem, memsize, &m) && m.h == n_rows && m.w == cols && m.nc == channels;     // the witness for the struct offsets
    *nbytes = memsize;
    if (!sane) { vfsms_set_error("jpeg_encode: libjpeg.so.8 does not have the expected struct layout"); rc = VFSMS_ERR_UNSUPPORTED; }
    else if (!out || cap < memsize) { vfsms_set_error("jpeg_encode: output buffer too small"); rc = VFSMS_ERR_CAPACITY; }
    else memcpy(out, mem, memsize);
    free(mem);
    return rc;
}

// Join stripes encoded by vfsms_jpeg_encode (same width, channels and quality; every stripe but the last `stripe_rows` high, a multiple of
// the MCU height: 16 for colour, 8 for gray) into ONE JPEG of `total_rows` rows: the headers of stripe 0 with the frame height patched and a
// DRI segment (restart interval = the MCUs of one stripe, at most 65535), then the entropy-coded segments with RSTn between them.
// `streams[k]`, `sizes[k]`: stripe k.  With out == NULL only the size is computed.
extern "C" int vfsms_jpeg_join(const uint8_t *const *streams, const size_t *sizes, int n_stripes, int stripe_rows, int total_rows, uint8_t *out, size_t cap, size_t *nbytes)
{
    if (!streams || !sizes || !nbytes || n_stripes < 1 || stripe_rows < 1 || total_rows < 1 || total_rows > 65500) { vfsms_set_error("jpeg_join: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    StreamMap m0;
    if (!streams[0] || !map_stream(streams[0], sizes[0], &m0)) { vfsms_set_error("jpeg_join: stripe 0 is not a stream of vfsms_jpeg_encode"); return VFSMS_ERR_BAD_ARG; }
    const int mcu_h = 8 * m0.vmax, mcu_w = m0.nc == 1 ? 8 : 16;
    const long long interval = (long long)((m0.w + mcu_w - 1) / mcu_w) * (stripe_rows / mcu_h);
    if (n_stripes > 1 && (stripe_rows % mcu_h || interval > 65535 || (m0.nc == 3 && m0.vmax != 2))) {
        vfsms_set_error("jpeg_join: stripes of %d rows cannot be restart intervals of this image (MCU %d x %d, %lld MCUs per stripe)", stripe_rows, mcu_w, mcu_h, interval);
        return VFSMS_ERR_BAD_ARG;
    }
    size_t need = m0.sos_end + (n_stripes > 1 ? 6 : 0) + 2;
    long long rows = 0;
    for (int k = 0; k < n_stripes; k++) {
        StreamMap m;
        if (!streams[k] || !map_stream(streams[k], sizes[k], &m) || m.w != m0.w || m.nc != m0.nc || (k + 1 < n_stripes && m.h != stripe_rows) ||
            m.sof != m0.sof || m.sos != m0.sos || m.sos_end != m0.sos_end || memcmp(streams[k] + 2, streams[0] + 2, m0.sof - 2) ||       // same tables in front of the frame header
            memcmp(streams[k] + m.sof + 9, streams[0] + m0.sof + 9, m0.sos_end - m0.sof - 9)) {                                        // same components, Huffman tables, scan header
            vfsms_set_error("jpeg_join: stripe %d does not continue stripe 0 (size, channels, tables or height)", k); return VFSMS_ERR_BAD_ARG;
        }
        rows += m.h;
        need += sizes[k] - 2 - m.sos_end + (k ? 2 : 0);
    }
    if (rows != total_rows) { vfsms_set_error("jpeg_join: the stripes hold %lld rows, not %d", rows, total_rows); return VFSMS_ERR_BAD_ARG; }
    *nbytes = need;
    if (!out) return VFSMS_OK;
    if (cap < need) { vfsms_set_error("jpeg_join: output buffer too small"); return VFSMS_ERR_CAPACITY; }
    uint8_t *o = out;
    memcpy(o, streams[0], m0.sos); o += m0.sos;                                         // SOI .. DHT
    out[m0.sof + 5] = (uint8_t)(total_rows >> 8); out[m0.sof + 6] = (uint8_t)total_rows;
    if (n_stripes > 1) { const uint8_t dri[6] = { 0xFF, 0xDD, 0, 4, (uint8_t)(interval >> 8), (uint8_t)interval }; memcpy(o, dri, 6); o += 6; }
    memcpy(o, streams[0] + m0.sos, m0.sos_end - m0.sos); o += m0.sos_end - m0.sos;      // the scan header
    for (int k = 0; k < n_stripes; k++) {
        if (k) { *o++ = 0xFF; *o++ = (uint8_t)(0xD0 + ((k - 1) & 7)); }
        const size_t n = sizes[k] - 2 - m0.sos_end;
        memcpy(o, streams[k] + m0.sos_end, n); o += n;
    }
    *o++ = 0xFF; *o++ = 0xD9;
    return (size_t)(o - out) == need ? VFSMS_OK : VFSMS_ERR_BAD_ARG;
}
