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
//   * the bf16 results of the 5 convolution rows go to an LDS tile as order-preserving 16-bit keys of sign(scale) * raw, the pool is nine
//     packed integer maxima per 8 channels, and bn1's scale / shift act on the selected element only.
// Two 4-wave workgroups per CU (256 registers per wave: the weights alone take 112).  Measured at batch 500 (scripts/bench_stem.py, the
// kernel's DALI_STEM_ABLATE switches): 170-190 us against 296 + 168 us of the convolution + pool launches; per tile the MFMA pipe (2240 cycles),
// the LDS port (270 KB = 2100 cycles: 140 KB of B fragments, 74 KB of pool reads, 40 KB of tile writes, 16 KB of patch) and the vector ALU
// (~2500 cycles of keys, maxima and index arithmetic) each need ~60 us and two waves per SIMD overlap them only in part.
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
    int lwo;                  // log2(Wo): W is 32, 64 or 128
    int ablate;               // diagnostic (DALI_STEM_ABLATE, results wrong): 1 no pool stage, 2 no output stage of the groups, 4 no MFMAs, 8 no prefetch requests
};

constexpr int SF_THREADS = 256, SF_NST = 4, SF_PATCH_ROWS = 15, SF_CONV_ROWS = 5;

__global__ __launch_bounds__(SF_THREADS, 2) void stem_conv_bn_pool_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, n16 = lane & 15;
    const int row_bytes = a.Wp * 8;
    const int patch_bytes = SF_PATCH_ROWS * row_bytes, nchunks = patch_bytes >> 4;
    constexpr int patch_pitch = SF_THREADS * SF_NST * 16;      // every staged register has a slot (requests past the patch return zeros into the padding):
                                                               // a register left unwritten behind a branch would still be waited for at the next request
    unsigned char* ctile = sf_smem + 2 * patch_pitch;              // [patch 0 | patch 1 | convolution tile | pool coefficients]
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
    // bn1 acts in the POOL stage, on the selected element only (a fifth of the convolution pixels): z = raw * scale + shift is monotone in raw, rising
    // or falling with the sign of scale, so the tile holds sign(scale) * raw (bf16, as order-preserving keys), the pool takes the maximum of that
    // and bn1 is applied to the one value it selects.  Bit for bit the three-launch form: bf16(raw) -> raw * scale + shift -> max -> bf16.
    // (first form: the affine on every accumulator with the 32 coefficients read from LDS per group -- 8 KB per group next to 7 KB of B fragments,
    //  the LDS port was the bound: 430 KB per tile against 2240 MFMA cycles.)
    uint32_t sflip[8];                                  // sign bits to flip in the packed pairs (channels 16 q + 2 p, + 1) of this lane's accumulators
#pragma unroll
    for (int p = 0; p < 8; ++p)
        sflip[p] = (a.scale[16 * q + 2 * p] < 0.f ? 0x8000u : 0u) | (a.scale[16 * q + 2 * p + 1] < 0.f ? 0x80000000u : 0u);
    // the pool stage's coefficients [|scale| 64 | shift 64] behind the tile (64 bytes per pooled item; in registers they pushed the B-fragment prefetch out)
    float* coef = reinterpret_cast<float*>(ctile + SF_CONV_ROWS * a.Wc * 128);
    if (tid < 64) { coef[tid] = fabsf(a.scale[tid]); coef[64 + tid] = a.shift[tid]; }

    const int tiles_per_img = a.Ho >> 1;
    // the patch of tile t: packed rows 4 i0 - 2 .. 4 i0 + 12 of image n, one contiguous range (rows -2, -1 of the first tile of an image only feed
    // convolution row -1, which the pool never reads: whatever lies there -- zeros before the buffer, the previous image's last rows -- is unused)
    // Order inside a turn: convolution rows of tile t -> the registers holding tile t + 1 into the OTHER patch buffer -> requests of tile t + 2 into
    // the same registers -> pool of tile t (stores).  vmcnt retires in order and the compiler cannot count the pool's stores: whatever it waits for
    // must not sit right behind them.  Here the registers are consumed a whole convolution stage after the stores before them were issued, and the
    // new requests start from an empty queue (requests at the top of the turn were made to wait for the previous turn's stores; one patch buffer,
    // registers consumed after the pool: the same, one store latency per tile).
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
    auto deposit = [&](unsigned char* patch, const uint4 (&st)[SF_NST]) {
#pragma unroll
        for (int s = 0; s < SF_NST; ++s) *reinterpret_cast<uint4*>(patch + (tid + SF_THREADS * s) * 16) = st[s];
    };
    uint4 st[SF_NST];
    int t = blockIdx.x;
    if (t < a.tiles) { request(t, st); deposit(sf_smem, st); }
    if (t + (int)gridDim.x < a.tiles) request(t + gridDim.x, st);
    // the weight fragments have landed before the loop (the compiler otherwise waits for them lazily INSIDE it, with counts that also cover the requests)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 7; ++r) asm volatile("" : "+v"(A[i][r]));
#pragma unroll
    for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(sflip[c]));
    const int lgpr = a.lwo - 3, ngroups = SF_CONV_ROWS << lgpr;        // 16-pixel groups per convolution row: Wc / 16 = Wo / 8
    int buf = 0;
    for (; t < a.tiles; t += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* patch = sf_smem + buf * patch_pitch;
        const bool more = t + (int)gridDim.x < a.tiles;
        // ---- the 5 convolution rows of this tile, 16 pixels x 64 channels per wave and turn ----
        auto b_base = [&](int g) -> const unsigned char* {
            const int cr = g >> lgpr, wc0 = (g - (cr << lgpr)) * 16;
            return patch + (2 * cr) * row_bytes + (wc0 + n16 + q) * 16;
        };
        // the first two tap rows' fragments of a group are requested under the previous group's MFMAs (one LDS round trip per group was exposed;
        // all seven ahead did not fit the 256 registers beside the 112 of the weights)
        bf16x8_t B[7];
        if (wave < ngroups) {
            const unsigned char* b0 = b_base(wave);
            B[0] = *reinterpret_cast<const bf16x8_t*>(b0);
            B[1] = *reinterpret_cast<const bf16x8_t*>(b0 + row_bytes);
        }
        for (int g = wave; g < ngroups; g += 4) {
            const int cr = g >> lgpr, wc0 = (g - (cr << lgpr)) * 16;
            {
                const unsigned char* b0 = b_base(g);
#pragma unroll
                for (int r = 2; r < 7; ++r) B[r] = *reinterpret_cast<const bf16x8_t*>(b0 + r * row_bytes);
            }
            __builtin_amdgcn_sched_barrier(0);           // all requests before the first MFMA (the scheduler otherwise issues them in pairs, each pair's latency exposed)
            f32x4_t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (!(a.ablate & 4)) {
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i][r], B[r], acc[i], 0, 0, 0);
            }
            if (g + 4 < ngroups) {
                const unsigned char* b0 = b_base(g + 4);
                B[0] = *reinterpret_cast<const bf16x8_t*>(b0);
                B[1] = *reinterpret_cast<const bf16x8_t*>(b0 + row_bytes);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int px = cr * a.Wc + wc0 + n16;
            if (!(a.ablate & 2)) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t k[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int e = 8 * h + 2 * p;
                    k[p] = order_key2(pack_bf16x2(acc[e >> 2][e & 3], acc[(e + 1) >> 2][(e + 1) & 3]) ^ sflip[4 * h + p]);
                }
                *reinterpret_cast<uint4*>(ctile + px * 128 + (((2 * q + h) ^ (px & 7)) << 4)) = make_uint4(k[0], k[1], k[2], k[3]);
            }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) deposit(sf_smem + (buf ^ 1) * patch_pitch, st);
        if (t + 2 * (int)gridDim.x < a.tiles && !(a.ablate & 8)) request(t + 2 * gridDim.x, st);
        asm volatile("" ::: "memory");             // (the compiler otherwise sinks these LDS writes below the pool's stores)
        // ---- 3x3 / stride 2 max-pool of the tile: window rows / columns outside the image are clamped onto a neighbour inside the window ----
        const int n = t / tiles_per_img, i0 = (t - n * tiles_per_img) * 2;
        const int items = 2 * a.Wo * 8;
        for (int it = tid; it < items && !(a.ablate & 1); it += SF_THREADS) {
            const int cc = it & 7, pix = it >> 3;
            const int pr = pix >> a.lwo, pc = pix - (pr << a.lwo);
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
            const uint32_t sel[4] = {order_key2(m.x), order_key2(m.y), order_key2(m.z), order_key2(m.w)};        // sign(scale) * raw of the selected elements
            float psc[8], psh[8];
#pragma unroll
            for (int c = 0; c < 8; c += 4) {
                const float4 s4 = *reinterpret_cast<const float4*>(coef + 8 * cc + c), h4 = *reinterpret_cast<const float4*>(coef + 64 + 8 * cc + c);
                psc[c] = s4.x; psc[c + 1] = s4.y; psc[c + 2] = s4.z; psc[c + 3] = s4.w;
                psh[c] = h4.x; psh[c + 1] = h4.y; psh[c + 2] = h4.z; psh[c + 3] = h4.w;
            }
            uint32_t zz[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // |sign(scale) * raw| = |raw|, and its sign times the sign of scale is raw's: raw * scale = (sign(scale) raw) * |scale|, exactly
                const float r0 = bf16_bits_to_f32(sel[p] & 0xffffu), r1 = bf16_bits_to_f32(sel[p] >> 16);
                zz[p] = pack_bf16x2(r0 * psc[2 * p] + psh[2 * p], r1 * psc[2 * p + 1] + psh[2 * p + 1]);
            }
            *reinterpret_cast<uint4*>(a.out + o) = make_uint4(zz[0], zz[1], zz[2], zz[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Training stem convolution: raw0 = conv1(images) (bf16 [N][H/2][W/2][64]) + per-tile partial BatchNorm sums (sum, sum of squares of the fp32
// accumulators, [tiles][64][2] as the implicit-GEMM kernels leave them for reduce_finish).  Same machinery as above (packed rows in an LDS patch,
// B fragments straight from it, weights in registers, two workgroups per CU); a tile is 4 convolution rows of one image: no halo, 13 packed rows.
// The 64 x 256 tile of the implicit-GEMM kernel ran this shape (K = 224: 7 k-steps per tile) at 160 us for 341 MB.  W = 128 only (256 pixels per
// tile = the statistics slab's row count the plan reserves).
// ------------------------------------------------------------------------------------------------
struct StemTrainArgs {
    const uint16_t* ximg; const uint16_t* w; uint16_t* raw; float* stats;
    int N, Hp, Wp, Hc, Wc, tiles;                // tiles = N * Hc / 4
};
constexpr int ST_PATCH_ROWS = 13, ST_CONV_ROWS = 4;
__device__ __forceinline__ float st_row16_sum(float v) {          // sum over the 16 lanes of a DPP row, in every lane of the row
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

__global__ __launch_bounds__(SF_THREADS, 2) void stem_conv_stats_kernel(StemTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, n16 = lane & 15;
    const int row_bytes = a.Wp * 8;
    const int nchunks = ST_PATCH_ROWS * row_bytes >> 4;
    constexpr int patch_pitch = SF_THREADS * SF_NST * 16;
    unsigned char* rtile = sf_smem + 2 * patch_pitch;              // [patch 0 | patch 1 | raw tile 256 px x 128 B | statistics 4 waves x 128 floats]
    float* wstat = reinterpret_cast<float*>(rtile + ST_CONV_ROWS * a.Wc * 128);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.ximg), 0, a.N * a.Hp * row_bytes, 0x00020000);
    bf16x8_t A[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int co = 16 * (n16 >> 2) + 4 * i + (n16 & 3);
            A[i][r] = *reinterpret_cast<const bf16x8_t*>(a.w + co * 224 + r * 32 + q * 8);
        }
    const int tiles_per_img = a.Hc >> 2;
    auto request = [&](int t, uint4 (&st)[SF_NST]) {               // packed rows 8 r4 .. 8 r4 + 12 of image n: one contiguous range
        const int n = t / tiles_per_img, r4 = t - n * tiles_per_img;
        const int base = (n * a.Hp + 8 * r4) * row_bytes;
#pragma unroll
        for (int s = 0; s < SF_NST; ++s) {
            const int c = tid + SF_THREADS * s;
            const uint32_t off = (uint32_t)(base + c * 16) | (c < nchunks ? 0u : DMA_OOB);
            st[s] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        }
    };
    auto deposit = [&](unsigned char* patch, const uint4 (&st)[SF_NST]) {
#pragma unroll
        for (int s = 0; s < SF_NST; ++s) *reinterpret_cast<uint4*>(patch + (tid + SF_THREADS * s) * 16) = st[s];
    };
    uint4 st[SF_NST];
    int t = blockIdx.x;
    if (t < a.tiles) { request(t, st); deposit(sf_smem, st); }
    if (t + (int)gridDim.x < a.tiles) request(t + gridDim.x, st);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 7; ++r) asm volatile("" : "+v"(A[i][r]));
    const int gpr = a.Wc >> 4, ngroups = ST_CONV_ROWS * gpr;
    int buf = 0;
    for (; t < a.tiles; t += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* patch = sf_smem + buf * patch_pitch;
        const bool more = t + (int)gridDim.x < a.tiles;
        float s1[16], s2[16];                                      // this lane's pixel column of the tile: sums over its groups, channels 16 q + e
#pragma unroll
        for (int e = 0; e < 16; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        for (int g = wave; g < ngroups; g += 4) {
            const int cr = g / gpr, wc0 = (g - cr * gpr) * 16;
            const unsigned char* b0 = patch + (2 * cr) * row_bytes + (wc0 + n16 + q) * 16;
            bf16x8_t B[7];
#pragma unroll
            for (int r = 0; r < 7; ++r) B[r] = *reinterpret_cast<const bf16x8_t*>(b0 + r * row_bytes);
            __builtin_amdgcn_sched_barrier(0);
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
                    const float v0 = acc[e >> 2][e & 3], v1 = acc[(e + 1) >> 2][(e + 1) & 3];
                    s1[e] += v0; s2[e] += v0 * v0; s1[e + 1] += v1; s2[e + 1] += v1 * v1;
                    k[p] = pack_bf16x2(v0, v1);
                }
                *reinterpret_cast<uint4*>(rtile + px * 128 + (((2 * q + h) ^ (px & 7)) << 4)) = make_uint4(k[0], k[1], k[2], k[3]);
            }
        }
        // the wave's share of the tile's sums: over its 16 pixel columns (DPP row), one writer per lane group
#pragma unroll
        for (int e = 0; e < 16; ++e) { s1[e] = st_row16_sum(s1[e]); s2[e] = st_row16_sum(s2[e]); }
        if (n16 == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { wstat[wave * 128 + (16 * q + e) * 2] = s1[e]; wstat[wave * 128 + (16 * q + e) * 2 + 1] = s2[e]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) deposit(sf_smem + (buf ^ 1) * patch_pitch, st);
        if (t + 2 * (int)gridDim.x < a.tiles) request(t + 2 * gridDim.x, st);
        asm volatile("" ::: "memory");
        // ---- the tile leaves as one contiguous range of raw0 (4 whole rows of one image), the sums as one row of the statistics slab ----
        uint16_t* dst = a.raw + (size_t)t * (ST_CONV_ROWS * a.Wc * 64);
        for (int c = tid; c < ST_CONV_ROWS * a.Wc * 8; c += SF_THREADS) {
            const int px = c >> 3, cc = c & 7;
            *reinterpret_cast<uint4*>(dst + (size_t)c * 8) = *reinterpret_cast<const uint4*>(rtile + px * 128 + ((cc ^ (px & 7)) << 4));
        }
        if (tid < 128) a.stats[(size_t)t * 128 + tid] = ((wstat[tid] + wstat[128 + tid]) + wstat[256 + tid]) + wstat[384 + tid];
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

// the training stem convolution on the patch kernel: 64 channels, W = 128 (256-pixel tiles: the statistics slab the plan reserves), H % 8 == 0
bool stem_train_supported(int N, int H, int W, int C) {
    if (C != 64 || W != 128 || H % 8 != 0) return false;
    if ((long long)N * (H + 6) * (W + 8) * 8 >= 0x7ff00000ll) return false;
    return DALI_ENV_INT("DALI_TRAIN_STEM", 1) != 0;
}
int stem_train_tiles(int N, int H) { return N * (H / 2 / 4); }
int launch_stem_conv_stats(hipStream_t st, const uint16_t* ximg, const uint16_t* w, int N, int H, int W, uint16_t* raw, float* stats) {
    if (!stem_train_supported(N, H, W, 64)) { set_error("stem_conv_stats: unsupported shape %d x %d x %d", N, H, W); return DALI_ERR_INVALID; }
    StemTrainArgs a{};
    a.ximg = ximg; a.w = w; a.raw = raw; a.stats = stats;
    a.N = N; a.Hp = H + 6; a.Wp = W + 8; a.Hc = H / 2; a.Wc = W / 2; a.tiles = stem_train_tiles(N, H);
    const int lds = 2 * SF_THREADS * SF_NST * 16 + ST_CONV_ROWS * a.Wc * 128 + 4 * 128 * 4;
    DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_stats_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)));
    const int n_cus = stem_cu_count();
    if (n_cus <= 0) { set_error("stem_conv_stats: device query failed"); return DALI_ERR_HIP; }
    const int grid = a.tiles < 2 * n_cus ? a.tiles : 2 * n_cus;
    char what[160];
    snprintf(what, sizeof what, "fwd,Cm=64,K=224,P=%d,taps=7,stride=2,sub=0,fused=0,stats=1,res=0,mask=0,lin=0", N * a.Hc * a.Wc);
    const int slot = gemm_profile_begin(st, 0, 2.0 * 64 * 224 * (double)N * a.Hc * a.Wc, what);     // counted with the GEMM launches, as the kernel it replaces
    hipLaunchKernelGGL(stem_conv_stats_kernel, dim3(grid), dim3(SF_THREADS), lds, st, a);
    gemm_profile_end(st, slot);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

// 64 output channels, an input the packed-row staging covers with 4 requests per thread, a pixel grid in whole groups of 16 columns
bool stem_fused_supported(int N, int H, int W, int C) {
    if (C != 64 || H % 32 != 0 || !(W == 32 || W == 64 || W == 128)) return false;
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
    a.lwo = W == 32 ? 3 : (W == 64 ? 4 : 5);
    const int lds = 2 * SF_THREADS * SF_NST * 16 + SF_CONV_ROWS * a.Wc * 128 + 512;
    a.ablate = DALI_ENV_INT("DALI_STEM_ABLATE", 0);
    DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_bn_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)));
    const int n_cus = stem_cu_count();
    if (n_cus <= 0) { set_error("stem_conv_bn_pool: device query failed"); return DALI_ERR_HIP; }
    const int grid = a.tiles < 2 * n_cus ? a.tiles : 2 * n_cus;          // two resident workgroups per CU (1: 258 us, 2: 177, 3: 190 at batch 500)
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
