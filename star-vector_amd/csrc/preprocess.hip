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

#include <mutex>
#include <vector>

#include "kernels.h"

namespace sv {

#define PP_BITS 22

struct PpArgs {
    const uint8_t* px; int W, H, C;      // source pixels, HWC
    int cw, ch, left, top;               // canvas (padded square, or the image itself) and the source offset inside it
    int S;                               // output side
    int recipe;                          // 0: ImageTrainProcessor (composite on white, pad, u/255 in float32)
                                         // 1: HF SiglipImageProcessor (alpha dropped, stretch, u * (1/255) in double)
    const int32_t* bounds_h; const int32_t* taps_h; int ksize_h;   // [S][2], [S][ksize]: canvas width  -> S
    const int32_t* bounds_v; const int32_t* taps_v; int ksize_v;   //                     canvas height -> S
    uint8_t* tmp;                        // [ch][S][3] horizontal pass output
    float* out;                          // [3][S][S]
    float mean[3], stdv[3];
};

__device__ __forceinline__ int pp_src(const PpArgs& p, int y, int x, int c) {
    const int yy = y - p.top, xx = x - p.left;
    if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) return 255;          // white padding
    const uint8_t* q = p.px + ((size_t)yy * p.W + xx) * p.C;
    if (p.C == 3 || p.recipe == 1) return q[c];                         // recipe 1: image.convert("RGB") drops alpha
    const int a = q[3];
    const int t = 255 * (255 - a) + (int)q[c] * a + 128;
    return ((t >> 8) + t) >> 8;
}
__device__ __forceinline__ int pp_clip8(int v) {
    v >>= PP_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}
__device__ __forceinline__ float pp_norm(const PpArgs& p, int u, int c) {
    // recipe 0: ToTensor = float32 u / 255.  recipe 1: HF rescale = float32(double(u) * (1 / 255))
    const float x = p.recipe == 1 ? (float)((double)u * 0.00392156862745098) : (float)u / 255.0f;
    return (x - p.mean[c]) / p.stdv[c];                                   // Normalize
}

__global__ __launch_bounds__(256) void pp_horizontal_kernel(PpArgs p) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xo >= p.S) return;
    const int x0 = p.bounds_h[2 * xo], n = p.bounds_h[2 * xo + 1];
    const int32_t* k = p.taps_h + (size_t)xo * p.ksize_h;
    int acc[3] = {1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1)};
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += pp_src(p, y, x0 + i, c) * w;
    }
    uint8_t* o = p.tmp + ((size_t)y * p.S + xo) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)pp_clip8(acc[c]);
}

__global__ __launch_bounds__(256) void pp_vertical_kernel(PpArgs p) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
    if (xo >= p.S) return;
    const int y0 = p.bounds_v[2 * yo], n = p.bounds_v[2 * yo + 1];
    const int32_t* k = p.taps_v + (size_t)yo * p.ksize_v;
    int acc[3] = {1 << (PP_BITS - 1), 1 << (PP_BITS - 1), 1 << (PP_BITS - 1)};
    for (int i = 0; i < n; ++i) {
        const int w = k[i];
        const uint8_t* q = p.tmp + ((size_t)(y0 + i) * p.S + xo) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += (int)q[c] * w;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) p.out[((size_t)c * p.S + yo) * p.S + xo] = pp_norm(p, pp_clip8(acc[c]), c);
}

// the padded square already has the target size: no resampling (torchvision Resize returns the image as is)
__global__ __launch_bounds__(256) void pp_copy_kernel(PpArgs p) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
    if (xo >= p.S) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) p.out[((size_t)c * p.S + yo) * p.S + xo] = pp_norm(p, pp_src(p, yo, xo, c), c);
}

// ---- host: Resample.c precompute_coeffs + normalize_coeffs_8bpc (double precision, no contraction) ---------------
#pragma clang fp contract(off)
static double pp_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static int pp_coeffs(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& taps) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    taps.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = pp_bicubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? k[x] / ww : k[x];
            taps[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PP_BITS)) : (int)(0.5 + v * (1 << PP_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

// workspace shared by all calls (grown on demand); one call at a time
static std::mutex g_pp_mu;
static void* g_pp_buf = nullptr;
static size_t g_pp_bytes = 0;

int preprocess_image(const uint8_t* dev_pixels, int width, int height, int channels, int out_size, int recipe,
                     const float* mean3, const float* std3, float* dev_out, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_pp_mu);
    PpArgs p;
    p.px = dev_pixels; p.W = width; p.H = height; p.C = channels; p.recipe = recipe;
    if (recipe == 0) {                    // white pad to square (data/util.py:56-62)
        p.cw = p.ch = width > height ? width : height;
        p.left = (p.cw - width) / 2; p.top = (p.ch - height) / 2;
    } else {                              // stretch: no padding
        p.cw = width; p.ch = height; p.left = p.top = 0;
    }
    p.S = out_size; p.out = dev_out;
    for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }
    p.bounds_h = p.bounds_v = nullptr; p.taps_h = p.taps_v = nullptr; p.ksize_h = p.ksize_v = 0; p.tmp = nullptr;
    const dim3 blk(256), grid_out((out_size + 255) / 256, out_size);
    if (p.cw == out_size && p.ch == out_size) {
        pp_copy_kernel<<<grid_out, blk, 0, st>>>(p);
        return (int)hipGetLastError();
    }
    // a pass whose input size equals the output size has identity taps (bicubic(0) = 1, bicubic(+-1) = 0): Pillow skips it
    std::vector<int32_t> bh, th, bv, tv;
    const int kh = pp_coeffs(p.cw, out_size, bh, th), kv = pp_coeffs(p.ch, out_size, bv, tv);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t tmp_bytes = (size_t)p.ch * out_size * 3;
    const size_t need = al(bh.size() * 4) + al(th.size() * 4) + al(bv.size() * 4) + al(tv.size() * 4) + tmp_bytes + 256;
    if (need > g_pp_bytes) {
        if (g_pp_buf) (void)hipFree(g_pp_buf);
        g_pp_buf = nullptr; g_pp_bytes = 0;
        hipError_t e = hipMalloc(&g_pp_buf, need);
        if (e != hipSuccess) return (int)e;
        g_pp_bytes = need;
    }
    char* q = (char*)g_pp_buf;
    int32_t* d_bh = (int32_t*)q; q += al(bh.size() * 4);
    int32_t* d_th = (int32_t*)q; q += al(th.size() * 4);
    int32_t* d_bv = (int32_t*)q; q += al(bv.size() * 4);
    int32_t* d_tv = (int32_t*)q; q += al(tv.size() * 4);
    uint8_t* d_tmp = (uint8_t*)q;
    hipError_t e = hipMemcpyAsync(d_bh, bh.data(), bh.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_th, th.data(), th.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bv, bv.data(), bv.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tv, tv.data(), tv.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);                    // the tables are host temporaries
    if (e != hipSuccess) return (int)e;
    p.bounds_h = d_bh; p.taps_h = d_th; p.ksize_h = kh; p.bounds_v = d_bv; p.taps_v = d_tv; p.ksize_v = kv; p.tmp = d_tmp;
    pp_horizontal_kernel<<<dim3((out_size + 255) / 256, p.ch), blk, 0, st>>>(p);
    pp_vertical_kernel<<<grid_out, blk, 0, st>>>(p);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);                    // the workspace is shared between calls
    return (int)e;
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

