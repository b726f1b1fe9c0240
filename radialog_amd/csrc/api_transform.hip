// librdx C ABI, part 6: the inference image transform on the GPU -- Resize(512) -> CenterCrop(448 | 488) -> ToTensor -> ExpandChannels of
// create_chest_xray_transform_for_inference (model/lavis/data/ReportDataset.py:80-106; call sites demo.py:144, :169, :251) on the 8-bit "L" image of demo.py:205-218.
// On a PIL image torchvision's Resize IS PIL.Image.resize(size, BILINEAR): the arithmetic is Pillow's src/libImaging/Resample.c (third-party, neither vendored nor
// pinned by the reference) -- two integer passes with a down-scale-stretched triangle filter. Restated here from the published algorithm; oracle/pil_resize.py is the
// numpy twin and both are held to Pillow bit for bit (tests/test_transforms.py on the CPU, tests/test_gpu_api.py on the GPU):
//   host  (this file)   output size (shorter side -> `resize`, longer side truncated), crop offsets (half to even), and per output column / row the filter window and its
//                       fixed-point coefficients -- C doubles, the expressions of precompute_coeffs / normalize_coeffs_8bpc (PRECISION_BITS = 22) in their order;
//   GPU   resample_h_k  horizontal pass over the columns the crop keeps and the rows the vertical pass reads: uint8 -> uint8, int32 accumulation from 2^21, >> 22, saturate;
//         resample_v_k  vertical pass of the crop window + ToTensor (/ 255 in fp32, correctly rounded) + the three identical channels.
// Byte / integer work, HBM-trivial (a 3056 x 2544 radiograph is 7.8 MB in, 2.4 MB out): no MFMA, coalesced byte rows, one thread per output element.
#include <cmath>
#include "rdx_ctx.h"

namespace {

constexpr int TF_PREC = 32 - 8 - 2;

struct TfCoeffs { std::vector<int> bounds, kk; int ksize = 0; };

// Pillow Resample.c: precompute_coeffs(inSize, in0 = 0, in1 = inSize, outSize, BILINEAR) + normalize_coeffs_8bpc
TfCoeffs tf_precompute(int in_size, int out_size) {
    TfCoeffs c;
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;                       // BILINEAR.support
    c.ksize = (int)ceil(support) * 2 + 1;
    c.bounds.assign((size_t)out_size * 2, 0);
    c.kk.assign((size_t)out_size * c.ksize, 0);
    std::vector<double> k(c.ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < c.ksize; ++x) k[x] = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            const double w = t < 1.0 ? 1.0 - t : 0.0;
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        c.bounds[(size_t)xx * 2] = xmin;
        c.bounds[(size_t)xx * 2 + 1] = xmax;
        for (int x = 0; x < c.ksize; ++x)
            c.kk[(size_t)xx * c.ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << TF_PREC)) : (int)(0.5 + k[x] * (1 << TF_PREC));
    }
    return c;
}

__device__ __forceinline__ unsigned tf_clip8(int acc) {
    const int v = acc >> TF_PREC;
    return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[y - y0][cx] for the image rows y in [y0, y0 + rows) and the resized columns left + cx, cx < crop
__global__ __launch_bounds__(256) void resample_h_k(const uint8_t* __restrict__ img, int W, int y0, int rows, const int* __restrict__ bounds, const int* __restrict__ kk,
                                                    int ksize, int left, int crop, uint8_t* __restrict__ tmp) {
    const int cx = blockIdx.x * blockDim.x + threadIdx.x, ry = blockIdx.y;
    if (cx >= crop || ry >= rows) return;
    const int xx = left + cx, xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const uint8_t* row = img + (size_t)(y0 + ry) * W + xmin;
    const int* k = kk + (size_t)xx * ksize;
    int acc = 1 << (TF_PREC - 1);
    for (int x = 0; x < cnt; ++x) acc += (int)row[x] * k[x];
    tmp[(size_t)ry * crop + cx] = (uint8_t)tf_clip8(acc);
}

// vertical pass over the crop window + ToTensor + ExpandChannels: out[ch][cy][cx], ch = 0..2
__global__ __launch_bounds__(256) void resample_v_k(const uint8_t* __restrict__ tmp, int y0, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int top,
                                                    int crop, float* __restrict__ out) {
    const int cx = blockIdx.x * blockDim.x + threadIdx.x, cy = blockIdx.y;
    if (cx >= crop) return;
    const int yy = top + cy, ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    int acc = 1 << (TF_PREC - 1);
    for (int y = 0; y < cnt; ++y) acc += (int)tmp[(size_t)(ymin + y - y0) * crop + cx] * k[y];
    const float v = __fdiv_rn((float)tf_clip8(acc), 255.0f);
    const size_t plane = (size_t)crop * crop, o = (size_t)cy * crop + cx;
    out[o] = v; out[plane + o] = v; out[2 * plane + o] = v;
}

// identity tables for a pass Pillow skips (equal sizes): one tap of weight 1.0 = 2^22 -> (2^21 + v 2^22) >> 22 = v
TfCoeffs tf_identity(int n) {
    TfCoeffs c;
    c.ksize = 1;
    c.bounds.resize((size_t)n * 2);
    c.kk.assign(n, 1 << TF_PREC);
    for (int i = 0; i < n; ++i) { c.bounds[2 * i] = i; c.bounds[2 * i + 1] = 1; }
    return c;
}

}  // namespace

extern "C" int rdx_transform_image(rdx_ctx* c, const uint8_t* img, int H, int W, int resize, int crop, float* out) {
    if (!c) return -1;
    if (!img || !out || H <= 0 || W <= 0 || resize <= 0 || crop <= 0) return fail(c, -1, "rdx_transform_image: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    // torchvision Resize(int): the shorter side becomes `resize`, the longer one int(resize * long / short) -- truncation (double arithmetic like Python's)
    const int shrt = W <= H ? W : H, lng = W <= H ? H : W;
    const int new_long = (int)((double)resize * (double)lng / (double)shrt);
    const int nw = W <= H ? resize : new_long, nh = W <= H ? new_long : resize;
    if (nw < crop || nh < crop)
        return fail(c, -1, "rdx_transform_image: a %d x %d image is %d x %d after Resize(%d): smaller than the %d px crop (torchvision would zero-pad; no RaDialog input does)", W, H, nw, nh, resize, crop);
    // CenterCrop: int(round((n - crop) / 2.0)) with Python's round = half to even = nearbyint in the default rounding mode
    const int top = (int)nearbyint((nh - crop) / 2.0), left = (int)nearbyint((nw - crop) / 2.0);
    const TfCoeffs ch = nw != W ? tf_precompute(W, nw) : tf_identity(W);
    const TfCoeffs cv = nh != H ? tf_precompute(H, nh) : tf_identity(H);
    // image rows the vertical pass of the crop window reads
    const int y0 = cv.bounds[(size_t)top * 2];
    const int y1 = cv.bounds[(size_t)(top + crop - 1) * 2] + cv.bounds[(size_t)(top + crop - 1) * 2 + 1];
    const int rows = y1 - y0;
    const size_t n_hb = ch.bounds.size(), n_hk = ch.kk.size(), n_vb = cv.bounds.size(), n_vk = cv.kk.size();
    const size_t ints = n_hb + n_hk + n_vb + n_vk, need = ints * sizeof(int) + (size_t)rows * crop;
    if (need > c->tf_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dfree(c, c->tf_ws); c->tf_bytes = 0;
        ALLOC(c, c->tf_ws, need);
        c->tf_bytes = need;
    }
    std::vector<int> host(ints);
    std::copy(ch.bounds.begin(), ch.bounds.end(), host.begin());
    std::copy(ch.kk.begin(), ch.kk.end(), host.begin() + n_hb);
    std::copy(cv.bounds.begin(), cv.bounds.end(), host.begin() + n_hb + n_hk);
    std::copy(cv.kk.begin(), cv.kk.end(), host.begin() + n_hb + n_hk + n_vb);
    int* d = reinterpret_cast<int*>(c->tf_ws);
    HIPCHK(c, hipMemcpyAsync(d, host.data(), ints * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // `host` is pageable and goes out of scope
    uint8_t* tmp = reinterpret_cast<uint8_t*>(d + ints);
    const dim3 blk(256), gh((crop + 255) / 256, rows), gv((crop + 255) / 256, crop);
    hipLaunchKernelGGL(resample_h_k, gh, blk, 0, c->stream, img, W, y0, rows, d, d + n_hb, ch.ksize, left, crop, tmp);
    hipLaunchKernelGGL(resample_v_k, gv, blk, 0, c->stream, tmp, y0, d + n_hb + n_hk, d + n_hb + n_hk + n_vb, cv.ksize, top, crop, out);
    HIPCHK(c, hipGetLastError());
    return 0;
}
