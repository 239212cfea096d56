// jpeg_host.hip -- host-side JPEG decode straight into pinned staging, for the ingest pipeline (scope row f-1: "host libjpeg-turbo + pinned
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
// ABI knowledge used (libjpeg 8 / libjpeg-turbo 2.x on x86-64, `boolean` = int): the public head of jpeg_decompress_struct up to
// output_scanline and the layout of jpeg_error_mgr -- restated below field by field.  Two guards make a mismatch a clean refusal
// (VFSMS_ERR_UNSUPPORTED -> the host falls back to Pillow) instead of a wrong image: the struct SIZE is taken from the library itself
// (jpeg_CreateDecompress is first called with a wrong size; its JERR_BAD_STRUCT_SIZE error carries the size the library expects), and
// after jpeg_read_header the image size read through these offsets must equal the size this file parses from the SOF marker itself.
#include "common.h"
#include <dlfcn.h>
#include <setjmp.h>
#include <string.h>
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
enum { JCS_GRAYSCALE_ = 1, JCS_YCbCr_ = 3 };
#define JPEG_ABI_VERSION 80
#define JPEG_CINFO_BYTES 2048           // room for the whole struct (632 bytes in libjpeg-turbo 2.1, ABI 8)

struct JpegApi {
    jerr_abi *(*std_error)(jerr_abi *);
    void (*create)(void *, int, size_t);
    void (*mem_src)(void *, const unsigned char *, unsigned long);
    int (*read_header)(void *, int);
    int (*start)(void *);
    unsigned (*read_scanlines)(void *, unsigned char **, unsigned);
    int (*finish)(void *);
    void (*destroy)(void *);
    size_t cinfo_size;                  // as the library reports it
    bool ok;
};
struct Guard { jmp_buf jb; int code; int parm0, parm1; char text[200]; };

void on_error(void *cinfo)
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
        a.finish = (int (*)(void *))dlsym(h, "jpeg_finish_decompress");
        a.destroy = (void (*)(void *))dlsym(h, "jpeg_destroy_decompress");
        if (!a.std_error || !a.create || !a.mem_src || !a.read_header || !a.start || !a.read_scanlines || !a.finish || !a.destroy) return a;
        // the struct size, from the library: a create call with a size no struct has fails with JERR_BAD_STRUCT_SIZE(library's, caller's)
        alignas(16) unsigned char cinfo[JPEG_CINFO_BYTES]; memset(cinfo, 0, sizeof(cinfo));
        jerr_abi err; memset(&err, 0, sizeof(err));
        Guard g; memset(&g.code, 0, sizeof(g) - sizeof(g.jb));
        jdec_head_abi *c = (jdec_head_abi *)cinfo;
        c->err = a.std_error(&err);
        err.error_exit = on_error; err.output_message = on_message;
        c->client_data = &g;
        if (setjmp(g.jb) == 0) { a.create(cinfo, JPEG_ABI_VERSION, 7); return a; }       // (a size of 7 cannot be right: the error branch is the expected one)
        if (g.parm1 != 7 || g.parm0 < (int)sizeof(jdec_head_abi) || g.parm0 > JPEG_CINFO_BYTES) return a;    // not the size error, or an implausible size
        a.cinfo_size = (size_t)g.parm0;
        a.ok = true;
        return a;
    }();
    return A;
}

// (rows, cols, components) from the first SOFn marker: the independent witness for the struct offsets
bool sof_size(const unsigned char *p, size_t n, int *h, int *w, int *nc)
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
    Guard g; memset(&g.code, 0, sizeof(g) - sizeof(g.jb));
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

// host-only entry (no context, no GPU): the decode of jpeg_decode_host, for tests and for callers that want the planes themselves
extern "C" int vfsms_jpeg_decode(const uint8_t *jpeg, size_t nbytes, int want_planes, uint8_t *out, size_t cap, int *h, int *w, int *comp)
{
    if (!h || !w || !comp) { vfsms_set_error("jpeg_decode: bad arguments"); return VFSMS_ERR_BAD_ARG; }
    return jpeg_decode_host(jpeg, nbytes, want_planes, out, cap, h, w, comp);
}
