// Image pre-processing on device (reference: starvector/data/util.py:40-68
// `ImageTrainProcessor`): RGBA -> composite on white, white pad to square, Pillow's antialiased BICUBIC resize (what
// torchvision `Resize` runs on a PIL image), ToTensor, Normalize.  A byte / integer path: the output equals the
// reference's float32 tensor bit for bit.  Recipe 1 is the SigLIP tower's HF image processor (image_encoder.py:45-48,
// 116-117): alpha dropped, the image stretched to S x S with the same resampler, rescale by 1/255 in double.
//
//   composite : Pillow Paste.c paste_mask_L on a white background, per channel
//               DIV255(255 * (255 - a) + c * a),  DIV255(t) = ((t' >> 8) + t') >> 8, t' = t + 128
//   resize    : Pillow Resample.c, 8 bits per channel: coefficients in double (host, precompute_coeffs), normalised and
//               converted to fixed point with 22 fractional bits (normalize_coeffs_8bpc), a horizontal pass into a uint8
//               image, then a vertical pass; each output = clip8((2^21 + sum pixel * tap) >> 22)
//   to tensor : float32 u / 255, then (x - mean) / std in float32 (IEEE subtract and divide, as torch does them)
//
// The composite + pad are folded into the horizontal pass (the padded square canvas is never materialised).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include <vector>

#include "kernels.h"

// Stateless and batched: one call handles up to any number of images in chunks of PP_MAXB; every per-image parameter
// travels in the kernel arguments, the fixed-point tap tables are computed ON DEVICE (same double-precision arithmetic as
// Pillow's precompute_coeffs / normalize_coeffs_8bpc, contraction off) into the caller's workspace, and nothing is copied
// from the host or synchronised inside the call: no process-global buffer, no lock, no hipStreamSynchronize.

namespace sv {

#define PP_BITS 22
#define PP_MAXB 32

struct PpImg {
    const uint8_t* px; int W, H, C;      // source pixels, HWC
    int cw, ch, left, top;               // canvas (padded square, or the image itself) and the source offset inside it
    int ksize_h, ksize_v;                // taps per output pixel of the two passes
    int copy;                            // 1: the canvas already has the target size (no resampling)
    unsigned tab_off, tmp_off;           // byte offsets inside the workspace: tables | horizontal-pass output [ch][S][3]
};
struct PpBatch {
    PpImg im[PP_MAXB];
    int n, S, recipe;                    // recipe 0: ImageTrainProcessor (composite on white, pad, u/255 in float32)
                                         //        1: HF SiglipImageProcessor (alpha dropped, stretch, u * (1/255) in double)
    float mean[3], stdv[3];
    char* ws;                            // workspace
    float* out;                          // [n][3][S][S]
};

// table layout of one image at ws + tab_off (int32): bounds_h [S][2] | taps_h [S][ksize_h] | bounds_v [S][2] | taps_v [S][ksize_v]
__host__ __device__ inline size_t pp_tab_ints(int S, int kh, int kv) { return (size_t)S * (4 + kh + kv); }

__device__ __forceinline__ int pp_src(const PpBatch& p, const PpImg& im, int y, int x, int c) {
    const int yy = y - im.top, xx = x - im.left;
    if (yy < 0 || yy >= im.H || xx < 0 || xx >= im.W) return 255;        // white padding
    const uint8_t* q = im.px + ((size_t)yy * im.W + xx) * im.C;
    if (im.C == 3 || p.recipe == 1) return q[c];                        // recipe 1: image.convert("RGB") drops alpha
    const int a = q[3];
    const int t = 255 * (255 - a) + (int)q[c] * a + 128;
    return ((t >> 8) + t) >> 8;
}
__device__ __forceinline__ int pp_clip8(int v) {
    v >>= PP_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}
__device__ __forceinline__ float pp_norm(const PpBatch& p, int u, int c) {
    // recipe 0: ToTensor = float32 u / 255.  recipe 1: HF rescale = float32(double(u) * (1 / 255))
    const float x = p.recipe == 1 ? (float)((double)u * 0.00392156862745098) : (float)u / 255.0f;
    return (x - p.mean[c]) / p.stdv[c];                                   // Normalize
}

// ---- Resample.c precompute_coeffs + normalize_coeffs_8bpc, double precision, NO contraction (host and device) ----
#pragma clang fp contract(off)
__host__ __device__ inline double pp_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
__host__ __device__ inline int pp_ksize(int in_size, int out_size) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}
// one output position xx: bounds (first input index, tap count) and the ksize fixed-point taps
__host__ __device__ inline void pp_coeff_row(int in_size, int out_size, int ksize, int xx, int32_t* bounds2, int32_t* taps) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += pp_bicubic((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < ksize; ++x) {
        int t = 0;
        if (x < xmax) {
            const double w = pp_bicubic((x + xmin - center + 0.5) * ss);
            const double v = ww != 0.0 ? w / ww : w;
            t = v < 0 ? (int)(-0.5 + v * (1 << PP_BITS)) : (int)(0.5 + v * (1 << PP_BITS));
        }
        taps[x] = t;
    }
    bounds2[0] = xmin;
    bounds2[1] = xmax;
}

__global__ __launch_bounds__(64) void pp_tables_kernel(PpBatch p) {
    const PpImg& im = p.im[blockIdx.z];
    if (im.copy) return;
    const int xx = blockIdx.x * 64 + threadIdx.x;
    if (xx >= p.S) return;
    int32_t* tab = reinterpret_cast<int32_t*>(p.ws + im.tab_off);
    if (blockIdx.y == 0) {
        pp_coeff_row(im.cw, p.S, im.ksize_h, xx, tab + 2 * xx, tab + 2 * p.S + (size_t)xx * im.ksize_h);
    } else {
        int32_t* tv = tab + (size_t)p.S * (2 + im.ksize_h);
        pp_coeff_row(im.ch, p.S, im.ksize_v, xx, tv + 2 * xx, tv + 2 * p.S + (size_t)xx * im.ksize_v);
    }
}

__global__ __launch_bounds__(256) void pp_horizontal_kernel(PpBatch p) {
    const PpImg& im = p.im[blockIdx.z];
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (im.copy || y >= im.ch || xo >= p.S) return;
    const int32_t* tab = reinterpret_cast<const int32_t*>(p.ws + im.tab_off);
    const int x0 = tab[2 * xo], n = tab[2 * xo + 1];
    const int32_t* k = tab + 2 * p.S + (size_t)xo * im.ksize_h;
    int acc[3] = {1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1)};
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += pp_src(p, im, y, x0 + i, c) * w;
    }
    uint8_t* o = reinterpret_cast<uint8_t*>(p.ws + im.tmp_off) + ((size_t)y * p.S + xo) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)pp_clip8(acc[c]);
}

__global__ __launch_bounds__(256) void pp_vertical_kernel(PpBatch p) {
    const PpImg& im = p.im[blockIdx.z];
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
    if (xo >= p.S) return;
    float* out = p.out + (size_t)blockIdx.z * 3 * p.S * p.S;
    if (im.copy) {        // the padded square already has the target size: torchvision Resize returns the image as is
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((size_t)c * p.S + yo) * p.S + xo] = pp_norm(p, pp_src(p, im, yo, xo, c), c);
        return;
    }
    const int32_t* tv = reinterpret_cast<const int32_t*>(p.ws + im.tab_off) + (size_t)p.S * (2 + im.ksize_h);
    const int y0 = tv[2 * yo], n = tv[2 * yo + 1];
    const int32_t* k = tv + 2 * p.S + (size_t)yo * im.ksize_v;
    const uint8_t* tmp = reinterpret_cast<const uint8_t*>(p.ws + im.tmp_off);
    int acc[3] = {1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1)};
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
        const uint8_t* q = tmp + ((size_t)(y0 + i) * p.S + xo) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += (int)q[c] * w;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((size_t)c * p.S + yo) * p.S + xo] = pp_norm(p, pp_clip8(acc[c]), c);
}

static void pp_geometry(int width, int height, int out_size, int recipe, PpImg& im) {
    if (recipe == 0) {                    // white pad to square (data/util.py:56-62)
        im.cw = im.ch = width > height ? width : height;
        im.left = (im.cw - width) / 2; im.top = (im.ch - height) / 2;
    } else {                              // stretch: no padding
        im.cw = width; im.ch = height; im.left = im.top = 0;
    }
    im.copy = (im.cw == out_size && im.ch == out_size) ? 1 : 0;
    // a pass whose input size equals the output size has identity taps (bicubic(0) = 1, bicubic(+-1) = 0): the same bytes
    im.ksize_h = im.copy ? 0 : pp_ksize(im.cw, out_size);
    im.ksize_v = im.copy ? 0 : pp_ksize(im.ch, out_size);
}
static size_t pp_al(size_t b) { return (b + 255) & ~(size_t)255; }

size_t preprocess_workspace_bytes(const int32_t* widths, const int32_t* heights, int n, int out_size, int recipe) {
    size_t need = 256;
    for (int i = 0; i < n; ++i) {
        PpImg im;
        pp_geometry(widths[i], heights[i], out_size, recipe, im);
        if (im.copy) continue;
        need += pp_al(pp_tab_ints(out_size, im.ksize_h, im.ksize_v) * 4) + pp_al((size_t)im.ch * out_size * 3);
    }
    return need;
}

// returns a hipError_t value, or -1 when the workspace is too small
int preprocess_images(const uint8_t* const* dev_pixels, const int32_t* widths, const int32_t* heights, const int32_t* channels,
                      int n, int out_size, int recipe, const float* mean3, const float* std3, float* dev_out, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
    if (preprocess_workspace_bytes(widths, heights, n, out_size, recipe) > workspace_bytes) return -1;
    size_t off = 0;
    for (int i0 = 0; i0 < n; i0 += PP_MAXB) {
        PpBatch p;
        p.n = n - i0 < PP_MAXB ? n - i0 : PP_MAXB;
        p.S = out_size; p.recipe = recipe; p.ws = (char*)workspace;
        p.out = dev_out + (size_t)i0 * 3 * out_size * out_size;
        for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }
        int max_ch = 0, any_resize = 0;
        for (int j = 0; j < p.n; ++j) {
            PpImg& im = p.im[j];
            im.px = dev_pixels[i0 + j]; im.W = widths[i0 + j]; im.H = heights[i0 + j]; im.C = channels[i0 + j];
            pp_geometry(im.W, im.H, out_size, recipe, im);
            im.tab_off = im.tmp_off = 0;
            if (!im.copy) {
                im.tab_off = (unsigned)off; off += pp_al(pp_tab_ints(out_size, im.ksize_h, im.ksize_v) * 4);
                im.tmp_off = (unsigned)off; off += pp_al((size_t)im.ch * out_size * 3);
                max_ch = im.ch > max_ch ? im.ch : max_ch;
                any_resize = 1;
            }
        }
        if (any_resize) {
            pp_tables_kernel<<<dim3((out_size + 63) / 64, 2, p.n), 64, 0, st>>>(p);
            pp_horizontal_kernel<<<dim3((out_size + 255) / 256, max_ch, p.n), 256, 0, st>>>(p);
        }
        pp_vertical_kernel<<<dim3((out_size + 255) / 256, out_size, p.n), 256, 0, st>>>(p);
    }
    return (int)hipGetLastError();
}

// one image, no caller workspace: a stream-ordered allocation (no global buffer, no synchronisation)
int preprocess_image(const uint8_t* dev_pixels, int width, int height, int channels, int out_size, int recipe,
                     const float* mean3, const float* std3, float* dev_out, hipStream_t st) {
    const int32_t w = width, h = height, c = channels;
    const size_t need = preprocess_workspace_bytes(&w, &h, 1, out_size, recipe);
    void* ws = nullptr;
    hipError_t e = hipMallocAsync(&ws, need, st);
    if (e != hipSuccess) return (int)e;
    const int r = preprocess_images(&dev_pixels, &w, &h, &c, 1, out_size, recipe, mean3, std3, dev_out, ws, need, st);
    e = hipFreeAsync(ws, st);
    return r ? r : (int)e;
}

// host copy of the table the device computes (CPU test surface)
static int pp_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& taps) {
    const int ksize = pp_ksize(in_size, out_size);
    bounds.assign((size_t)out_size * 2, 0);
    taps.assign((size_t)out_size * ksize, 0);
    for (int xx = 0; xx < out_size; ++xx) pp_coeff_row(in_size, out_size, ksize, xx, &bounds[2 * xx], &taps[(size_t)xx * ksize]);
    return ksize;
}

}  // namespace sv

// Host-only test surface (callable without a GPU): the fixed-point resampling table the device passes consume, so the
// CPU suite can hold it against the oracle's restatement of Pillow's precompute_coeffs / normalize_coeffs_8bpc.
// bounds [out_size][2] (first input index, tap count), taps [out_size][cap] (row stride = cap >= ksize); returns ksize.
extern "C" int sv_debug_resample_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* taps, int32_t cap) {
    if (in_size < 1 || out_size < 1 || !bounds || !taps) return -22;
    std::vector<int32_t> b, t;
    const int ksize = sv::pp_coeffs(in_size, out_size, b, t);
    if (ksize > cap) return -22;
    for (int i = 0; i < out_size; ++i) {
        bounds[2 * i] = b[2 * i];
        bounds[2 * i + 1] = b[2 * i + 1];
        for (int k = 0; k < cap; ++k) taps[(size_t)i * cap + k] = k < ksize ? t[(size_t)i * ksize + k] : 0;
    }
    return ksize;
}

