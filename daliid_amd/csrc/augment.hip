// augment.hip -- the image side of the input pipeline on decoded uint8 images:
//   eval   (getFeatures.py:18-19):            Resize((H,W), bicubic) -> ToTensor -> Normalize
//   train  (train_encodersKIT.py:313-320):    Resize bicubic -> RandomCrop((H,W), padding=10) -> RandomHorizontalFlip ->
//                                             ColorJitter(0.4, 0.3, 0.4, 0) -> ToTensor -> RandomErasing(p=1, 0.05-0.30) -> Normalize
// JPEG decode stays on the host (read_image); the random parameters are drawn on the host in torchvision's order
// (daliid_amd/transforms.py) so this file is deterministic pixel arithmetic, written to reproduce PIL bit for bit on
// the uint8 stages:
//   * Resize = Pillow's ImagingResample: separable, horizontal pass then vertical pass with a uint8 intermediate, fixed
//     point coefficients (22 fractional bits, built by the host exactly as precompute_coeffs / normalize_coeffs_8bpc do),
//     accumulate from 1 << 21, arithmetic shift, clip to 0..255;
//   * ColorJitter on a PIL image = ImageEnhance.{Brightness, Contrast, Color}: Image.blend(degenerate, image, factor) in
//     fp32 with truncation (clipped when the factor extrapolates), degenerate = black / the rounded mean of the L image /
//     the L image, L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16.
// The augmentation kernel keeps one whole image in LDS (256 x 128 x 3 bytes = 96 KiB of the CU's 160 KiB): crop + flip
// on load, the three enhancers in their random order with a block-wide reduction for the contrast mean, erase + normalise
// on store.  HBM-bound: 3 bytes read + 12 written per output pixel.
#include "common.h"

namespace dali {

constexpr int RESAMPLE_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8_shift(int v) {
    v >>= RESAMPLE_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[n][y][xx][c] = clip8((2^21 + sum_k src[n][y][xmin+k][c] * kk[xx][k]) >> 22)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ src_off,
                                                          const int32_t* __restrict__ in_h, const int32_t* __restrict__ in_w,
                                                          const int32_t* __restrict__ table_of, const int32_t* __restrict__ bounds,
                                                          const int32_t* __restrict__ coefs, int ksize, int out_w, int max_in_h,
                                                          uint8_t* __restrict__ tmp) {
    const int n = blockIdx.y;
    const int ih = in_h[n], iw = in_w[n];
    const uint8_t* img = src + src_off[n];
    const int32_t* bnd = bounds + (size_t)table_of[n] * out_w * 2;
    const int32_t* kk = coefs + (size_t)table_of[n] * out_w * ksize;
    uint8_t* out = tmp + (size_t)n * max_in_h * out_w * 3;
    const int total = ih * out_w * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c = i % 3, xx = (i / 3) % out_w, y = i / (3 * out_w);
        const int xmin = bnd[xx * 2], xn = bnd[xx * 2 + 1];
        const int32_t* k = kk + (size_t)xx * ksize;
        const uint8_t* row = img + ((size_t)y * iw + xmin) * 3 + c;
        int ss = 1 << (RESAMPLE_PRECISION_BITS - 1);
        for (int x = 0; x < xn; ++x) ss += (int)row[x * 3] * k[x];
        out[i] = (uint8_t)clip8_shift(ss);
    }
}

// vertical pass: out[n][yy][x][c] from tmp[n][ymin+k][x][c]
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ table_of,
                                                          const int32_t* __restrict__ bounds, const int32_t* __restrict__ coefs, int ksize,
                                                          int out_h, int out_w, int max_in_h, uint8_t* __restrict__ out) {
    const int n = blockIdx.y;
    const int32_t* bnd = bounds + (size_t)table_of[n] * out_h * 2;
    const int32_t* kk = coefs + (size_t)table_of[n] * out_h * ksize;
    const uint8_t* in = tmp + (size_t)n * max_in_h * out_w * 3;
    uint8_t* o = out + (size_t)n * out_h * out_w * 3;
    const int rowb = out_w * 3, total = out_h * rowb;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int xc = i % rowb, yy = i / rowb;
        const int ymin = bnd[yy * 2], yn = bnd[yy * 2 + 1];
        const int32_t* k = kk + (size_t)yy * ksize;
        int ss = 1 << (RESAMPLE_PRECISION_BITS - 1);
        for (int y = 0; y < yn; ++y) ss += (int)in[(size_t)(ymin + y) * rowb + xc] * k[y];
        o[i] = (uint8_t)clip8_shift(ss);
    }
}

// ImagingBlend on one 8-bit channel: in1 = degenerate, in2 = image
__device__ __forceinline__ int blend8(int in1, int in2, float alpha, bool interp) {
    const float t = (float)in1 + alpha * (float)(in2 - in1);
    if (interp) return (int)t;                      // 0 <= alpha <= 1: (UINT8) of an in-range value
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ int luma8(int r, int g, int b) { return (19595 * r + 38470 * g + 7471 * b + 0x8000) >> 16; }

// per-image parameters, 16 int32 / float32 words:
//   [0] crop top  [1] crop left (offsets into the padded image, 0..2*pad)   [2] flip (0/1)   [3..6] op order: 0 brightness,
//   1 contrast, 2 saturation, 3 hue (no-op), -1 skip   [7] erase i  [8] erase j  [9] erase h  [10] erase w (h == 0: none)
//   [11] brightness factor (float bits)  [12] contrast  [13] saturation  [14] pad  [15] augment (0: eval path = normalise only)
constexpr int AUG_WORDS = 16;

__global__ __launch_bounds__(1024) void augment_kernel(const uint8_t* __restrict__ img, const int32_t* __restrict__ params, int H, int W,
                                                        float m0, float m1, float m2, float s0, float s1, float s2, float* __restrict__ out) {
    extern __shared__ uint8_t px[];                 // [H][W][3]
    __shared__ unsigned long long red[16];
    __shared__ int mean_l;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int32_t* p = params + (size_t)n * AUG_WORDS;
    const uint8_t* src = img + (size_t)n * H * W * 3;
    const int npix = H * W;
    const bool aug = p[15] != 0;
    const int top = p[0], left = p[1], flip = p[2], pad = p[14];
    // ---- load: zero-padded crop + horizontal flip ----
    for (int i = tid; i < npix; i += 1024) {
        const int h = i / W, w = i - h * W;
        int r = 0, g = 0, b = 0;
        if (aug) {
            const int ws = flip ? W - 1 - w : w;
            const int sh = h + top - pad, sw = ws + left - pad;
            if ((unsigned)sh < (unsigned)H && (unsigned)sw < (unsigned)W) {
                const uint8_t* q = src + ((size_t)sh * W + sw) * 3;
                r = q[0]; g = q[1]; b = q[2];
            }
        } else {
            const uint8_t* q = src + (size_t)i * 3;
            r = q[0]; g = q[1]; b = q[2];
        }
        px[i * 3] = (uint8_t)r; px[i * 3 + 1] = (uint8_t)g; px[i * 3 + 2] = (uint8_t)b;
    }
    __syncthreads();
    // ---- ColorJitter: the enhancers in the drawn order ----
    if (aug) {
        for (int step = 0; step < 4; ++step) {
            const int op = p[3 + step];
            if (op < 0 || op > 2) continue;          // hue = 0 is a no-op (torchvision drops it)
            const float f = __int_as_float(p[11 + op]);
            const bool interp = f >= 0.f && f <= 1.f;
            if (op == 1) {                           // contrast: degenerate = int(mean(L) + 0.5)
                unsigned long long s = 0;
                for (int i = tid; i < npix; i += 1024) s += (unsigned)luma8(px[i * 3], px[i * 3 + 1], px[i * 3 + 2]);
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if ((tid & 63) == 0) red[tid >> 6] = s;
                __syncthreads();
                if (tid == 0) {
                    unsigned long long t = 0;
                    for (int k = 0; k < 16; ++k) t += red[k];
                    mean_l = (int)((double)t / (double)npix + 0.5);
                }
                __syncthreads();
            }
            const int mean = mean_l;
            for (int i = tid; i < npix; i += 1024) {
                const int r = px[i * 3], g = px[i * 3 + 1], b = px[i * 3 + 2];
                const int d = op == 0 ? 0 : (op == 1 ? mean : luma8(r, g, b));
                px[i * 3] = (uint8_t)blend8(d, r, f, interp);
                px[i * 3 + 1] = (uint8_t)blend8(d, g, f, interp);
                px[i * 3 + 2] = (uint8_t)blend8(d, b, f, interp);
            }
            __syncthreads();
        }
    }
    // ---- ToTensor, RandomErasing (value 0 before normalisation), Normalize; fp32 NCHW ----
    const int ei = p[7], ej = p[8], eh = aug ? p[9] : 0, ew = p[10];
    const float mean3[3] = {m0, m1, m2}, std3[3] = {s0, s1, s2};
    float* o = out + (size_t)n * 3 * npix;
    for (int i = tid; i < npix; i += 1024) {
        const int h = i / W, w = i - h * W;
        const bool erased = eh > 0 && h >= ei && h < ei + eh && w >= ej && w < ej + ew;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = erased ? 0.f : (float)px[i * 3 + c] / 255.0f;
            o[(size_t)c * npix + i] = (v - mean3[c]) / std3[c];
        }
    }
}

}  // namespace dali

using namespace dali;

extern "C" int dali_resize_bicubic_u8(dali_ctx* ctx, void* stream, const uint8_t* src, const int64_t* src_off, const int32_t* in_h,
                                      const int32_t* in_w, const int32_t* table_of, int n, int max_in_h, const int32_t* bounds_h,
                                      const int32_t* coefs_h, int ksize_h, const int32_t* bounds_v, const int32_t* coefs_v, int ksize_v,
                                      int out_h, int out_w, uint8_t* out) {
    DALI_REQUIRE(ctx && src && src_off && in_h && in_w && table_of && bounds_h && coefs_h && bounds_v && coefs_v && out,
                 "dali_resize_bicubic_u8: null argument");
    DALI_REQUIRE(n >= 0 && n <= 65535 && max_in_h > 0 && out_h > 0 && out_w > 0 && ksize_h > 0 && ksize_v > 0,
                 "dali_resize_bicubic_u8: bad sizes n=%d max_in_h=%d out=%dx%d", n, max_in_h, out_h, out_w);
    if (n == 0) return DALI_OK;
    uint8_t* tmp = static_cast<uint8_t*>(workspace(ctx, (size_t)n * max_in_h * out_w * 3));
    if (!tmp) return DALI_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    const int bx_h = (max_in_h * out_w * 3 + 255) / 256, bx_v = (out_h * out_w * 3 + 255) / 256;
    hipLaunchKernelGGL(resample_h_kernel, dim3(bx_h < 64 ? bx_h : 64, n), dim3(256), 0, st, src, src_off, in_h, in_w, table_of, bounds_h, coefs_h,
                       ksize_h, out_w, max_in_h, tmp);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_v_kernel, dim3(bx_v < 64 ? bx_v : 64, n), dim3(256), 0, st, tmp, table_of, bounds_v, coefs_v, ksize_v, out_h, out_w,
                       max_in_h, out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_augment_batch(dali_ctx* ctx, void* stream, const uint8_t* images, const int32_t* params, int n, int h, int w,
                                  const float* mean3, const float* std3, float* out) {
    DALI_REQUIRE(ctx && images && params && mean3 && std3 && out, "dali_augment_batch: null argument");
    DALI_REQUIRE(n >= 0 && h > 0 && w > 0, "dali_augment_batch: bad sizes");
    const size_t lds = (size_t)h * w * 3;
    DALI_REQUIRE(lds <= 150 * 1024, "dali_augment_batch: an image of %d x %d does not fit the 160 KiB LDS (limit 150 KiB)", h, w);
    if (n == 0) return DALI_OK;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&augment_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(augment_kernel, dim3(n), dim3(1024), lds, (hipStream_t)stream, images, params, h, w, mean3[0], mean3[1], mean3[2],
                       std3[0], std3[1], std3[2], out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
