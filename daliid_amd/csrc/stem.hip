// Inference stem in one launch: conv1 (7x7 / stride 2 / pad 3, 3 -> 64) -> bn1 by running statistics (no ReLU: Encoders.py:321-322, :334)
// -> 3x3 / stride 2 / pad 1 max-pool, from the packed [N][H+6][W+8][4] bf16 image to the pooled [N][H/4][W/4][64] bf16 tensor.
// (torchvision resnet50's conv1 / bn1 / maxpool under Encoders.py:33,36; the training forward keeps the three-launch form: the batch statistics
//  of the convolution's output have to exist before the pool can run.)
//
// The three-launch form writes the convolution's output (N x 128 x 64 x 64 bf16 = 1 MB per image), reads it back through the pool and is bound
// by its own per-tile overheads (64-channel tiles: 7 k-steps per 64 x 256 tile).  Here a workgroup owns 2 pooled rows of one image:
//   * the 15 packed input rows under them are ONE contiguous global range; they are staged through registers (requested while the previous
//     tile is multiplied) into an LDS patch;
//   * the B operand of every MFMA is read straight from that patch: for convolution pixel (row, col) and tap row r the 32 k-values
//     (8 taps x 4 channels) are the 64 contiguous bytes at patch row 2 row + r, byte 16 col -- no im2col expansion anywhere;
//   * the weights (64 x 7 x 32) live in registers as 28 A fragments per lane, with the channel order chosen so that a lane's 16 accumulators
//     are 16 CONTIGUOUS channels of one pixel;
//   * bn1's scale / shift act on the fp32 accumulators, the bf16 results of the 5 convolution rows go to an LDS tile as order-preserving
//     16-bit keys, and the pool is nine packed integer maxima per 8 channels.
// Two 4-wave workgroups per CU: one's pool / staging runs under the other's MFMAs.
#include "common.h"
#include "kernels.h"
#include "gemm_tile.h"

namespace dali {
namespace {

typedef short s16x2_t __attribute__((ext_vector_type(2)));
// bf16 bits -> a signed 16-bit key with the same order (negative values: magnitude bits flipped); its own inverse.  Two values per register.
__device__ __forceinline__ uint32_t order_key2(uint32_t x) {
    const s16x2_t v = __builtin_bit_cast(s16x2_t, x);
    const s16x2_t m = {(short)0x7FFF, (short)0x7FFF};
    return __builtin_bit_cast(uint32_t, (s16x2_t)(v ^ ((v >> 15) & m)));
}
__device__ __forceinline__ uint32_t max_key2(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}
__device__ __forceinline__ uint4 max_key8(uint4 a, uint4 b) {
    return make_uint4(max_key2(a.x, b.x), max_key2(a.y, b.y), max_key2(a.z, b.z), max_key2(a.w, b.w));
}

struct StemArgs {
    const uint16_t* ximg;     // [N][Hp][Wp][4] bf16, zero border (stem_pack_image_kernel)
    const uint16_t* w;        // [64][7][8][4] bf16 (stem_pack_weight_kernel)
    const float* scale; const float* shift;      // bn1 by running statistics
    uint16_t* out;            // [N][Ho][Wo][64]
    int N, Hp, Wp, Wc, Ho, Wo, tiles;            // Wc = 2 Wo convolution columns; tiles = N * Ho / 2
};

constexpr int SF_THREADS = 256, SF_NST = 4, SF_PATCH_ROWS = 15, SF_CONV_ROWS = 5;

__global__ __launch_bounds__(SF_THREADS, 2) void stem_conv_bn_pool_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, n16 = lane & 15;
    const int row_bytes = a.Wp * 8;
    const int patch_bytes = SF_PATCH_ROWS * row_bytes, nchunks = patch_bytes >> 4;
    unsigned char* patch = sf_smem;
    unsigned char* ctile = sf_smem + ((patch_bytes + 127) & ~127);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.ximg), 0, a.N * a.Hp * row_bytes, 0x00020000);

    // A fragments: fragment (i, r) row m is channel 16 (m >> 2) + 4 i + (m & 3), so that accumulator (i, j) of lane group q is channel 16 q + 4 i + j
    bf16x8_t A[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int co = 16 * (n16 >> 2) + 4 * i + (n16 & 3);
            A[i][r] = *reinterpret_cast<const bf16x8_t*>(a.w + co * 224 + r * 32 + q * 8);
        }
    float sc[16], sh[16];
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
        const float4 s4 = *reinterpret_cast<const float4*>(a.scale + 16 * q + e), h4 = *reinterpret_cast<const float4*>(a.shift + 16 * q + e);
        sc[e] = s4.x; sc[e + 1] = s4.y; sc[e + 2] = s4.z; sc[e + 3] = s4.w;
        sh[e] = h4.x; sh[e + 1] = h4.y; sh[e + 2] = h4.z; sh[e + 3] = h4.w;
    }

    const int tiles_per_img = a.Ho >> 1;
    // the patch of tile t: packed rows 4 i0 - 2 .. 4 i0 + 12 of image n, one contiguous range (rows -2, -1 of the first tile of an image only feed
    // convolution row -1, which the pool never reads: whatever lies there -- zeros before the buffer, the previous image's last rows -- is unused)
    auto request = [&](int t, uint4 (&st)[SF_NST]) {
        const int n = t / tiles_per_img, i0 = (t - n * tiles_per_img) * 2;
        const int base = (n * a.Hp + 4 * i0 - 2) * row_bytes;
#pragma unroll
        for (int s = 0; s < SF_NST; ++s) {
            const int c = tid + SF_THREADS * s;
            const uint32_t off = c < nchunks ? (uint32_t)(base + c * 16) : DMA_OOB;
            st[s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        }
    };
    uint4 st[SF_NST];
    int t = blockIdx.x;
    if (t < a.tiles) request(t, st);
    const int gpr = a.Wc >> 4, ngroups = SF_CONV_ROWS * gpr;
    for (; t < a.tiles; t += gridDim.x) {
#pragma unroll
        for (int s = 0; s < SF_NST; ++s) {
            const int c = tid + SF_THREADS * s;
            if (c < nchunks) *reinterpret_cast<uint4*>(patch + c * 16) = st[s];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + (int)gridDim.x < a.tiles) request(t + gridDim.x, st);
        // ---- the 5 convolution rows of this tile, 16 pixels x 64 channels per wave and turn ----
        for (int g = wave; g < ngroups; g += 4) {
            const int cr = g / gpr, wc0 = (g - cr * gpr) * 16;
            const unsigned char* b0 = patch + (2 * cr) * row_bytes + (wc0 + n16 + q) * 16;
            bf16x8_t B[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) B[r] = *reinterpret_cast<const bf16x8_t*>(b0 + r * row_bytes);
            f32x4_t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i][r], B[r], acc[i], 0, 0, 0);
            const int px = cr * a.Wc + wc0 + n16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t k[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int e = 8 * h + 2 * p;
                    const float z0 = fmaf(acc[e >> 2][e & 3], sc[e], sh[e]), z1 = fmaf(acc[(e + 1) >> 2][(e + 1) & 3], sc[e + 1], sh[e + 1]);
                    k[p] = order_key2(pack_bf16x2(z0, z1));
                }
                *reinterpret_cast<uint4*>(ctile + px * 128 + (((2 * q + h) ^ (px & 7)) << 4)) = make_uint4(k[0], k[1], k[2], k[3]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- 3x3 / stride 2 max-pool of the tile: window rows / columns outside the image are clamped onto a neighbour inside the window ----
        const int n = t / tiles_per_img, i0 = (t - n * tiles_per_img) * 2;
        const int items = 2 * a.Wo * 8;
        for (int it = tid; it < items; it += SF_THREADS) {
            const int cc = it & 7, pix = it >> 3;
            const int pr = pix / a.Wo, pc = pix - pr * a.Wo;
            int tr[3] = {2 * pr, 2 * pr + 1, 2 * pr + 2};
            if (i0 + pr == 0) tr[0] = 1;
            int tc[3] = {2 * pc - 1, 2 * pc, 2 * pc + 1};
            if (pc == 0) tc[0] = 0;
            uint4 v[9];
#pragma unroll
            for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    const int px = tr[dr] * a.Wc + tc[dc];
                    v[dr * 3 + dc] = *reinterpret_cast<const uint4*>(ctile + px * 128 + ((cc ^ (px & 7)) << 4));
                }
            uint4 m = max_key8(max_key8(max_key8(v[0], v[1]), max_key8(v[2], v[3])), max_key8(max_key8(v[4], v[5]), max_key8(v[6], v[7])));
            m = max_key8(m, v[8]);
            const size_t o = (((size_t)n * a.Ho + i0 + pr) * a.Wo + pc) * 64 + cc * 8;
            *reinterpret_cast<uint4*>(a.out + o) = make_uint4(order_key2(m.x), order_key2(m.y), order_key2(m.z), order_key2(m.w));
        }
    }
}

int stem_cu_count() {
    static int n_cus = 0;
    if (!n_cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
        n_cus = v > 0 ? v : 1;
    }
    return n_cus;
}

}  // namespace

// 64 output channels, an input the packed-row staging covers with 4 requests per thread, a pixel grid in whole groups of 16 columns
bool stem_fused_supported(int N, int H, int W, int C) {
    if (C != 64 || H % 32 != 0 || W % 32 != 0 || W > 128) return false;
    const int Wp = W + 8, Hp = H + 6;
    if (SF_PATCH_ROWS * Wp * 8 / 16 > SF_THREADS * SF_NST) return false;
    if ((long long)N * Hp * Wp * 8 >= 0x7ff00000ll) return false;
    return DALI_ENV_INT("DALI_EVAL_STEM", 1) != 0;
}

int launch_stem_conv_bn_pool(hipStream_t st, const uint16_t* ximg, const uint16_t* w, const float* scale, const float* shift, int N, int H, int W,
                             uint16_t* out) {
    if (!stem_fused_supported(N, H, W, 64)) { set_error("stem_conv_bn_pool: unsupported shape %d x %d x %d", N, H, W); return DALI_ERR_INVALID; }
    StemArgs a{};
    a.ximg = ximg; a.w = w; a.scale = scale; a.shift = shift; a.out = out;
    a.N = N; a.Hp = H + 6; a.Wp = W + 8; a.Wc = W / 2; a.Ho = H / 4; a.Wo = W / 4;
    a.tiles = N * (a.Ho / 2);
    const int patch = ((SF_PATCH_ROWS * a.Wp * 8 + 127) & ~127), lds = patch + SF_CONV_ROWS * a.Wc * 128;
    DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_bn_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)));
    const int n_cus = stem_cu_count();
    if (n_cus <= 0) { set_error("stem_conv_bn_pool: device query failed"); return DALI_ERR_HIP; }
    const int grid = a.tiles < 2 * n_cus ? a.tiles : 2 * n_cus;
    hipLaunchKernelGGL(stem_conv_bn_pool_kernel, dim3(grid), dim3(SF_THREADS), lds, st, a);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

}  // namespace dali

extern "C" int dali_stem_fused_supported(int n, int h, int w) { return dali::stem_fused_supported(n, h, w, 64) ? 1 : 0; }

extern "C" int dali_stem_conv_bn_maxpool(dali_ctx* ctx, void* stream, const float* images, int n, int h, int w, const float* weight, const float* scale,
                                         const float* shift, uint16_t* y) {
    using namespace dali;
    DALI_REQUIRE(ctx && images && weight && scale && shift && y, "dali_stem_conv_bn_maxpool: null argument");
    DALI_REQUIRE(n > 0 && stem_fused_supported(n, h, w, 64), "dali_stem_conv_bn_maxpool: unsupported shape %d x %d x %d (dali_stem_fused_supported)", n, h, w);
    const size_t img_bytes = align_up((size_t)n * (h + 6) * (w + 8) * 8, 256);
    unsigned char* ws = static_cast<unsigned char*>(workspace(ctx, img_bytes + 64 * 224 * 2));
    if (!ws) return DALI_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    uint16_t* ximg = reinterpret_cast<uint16_t*>(ws);
    uint16_t* wp = reinterpret_cast<uint16_t*>(ws + img_bytes);
    int rc;
    if ((rc = launch_stem_pack_image(st, images, n, h, w, ximg))) return rc;
    if ((rc = launch_stem_pack_weight(st, weight, 64, wp))) return rc;
    return launch_stem_conv_bn_pool(st, ximg, wp, scale, shift, n, h, w, y);
}
