// gemm_tile.h -- the bf16 MFMA block-tile engine every contraction in libdaliid_hip is built on
// (pair distance, loss-head similarity GEMMs, implicit-GEMM convolutions, ViT linears).
//
// Shape: block tile TM x TN x 32, 256 threads = 4 wave64 in a 2x2 grid, each wave owns a
// (TM/2) x (TN/2) sub-tile as FM x FN accumulators of v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
// Operand A supplies the MFMA rows (m), operand B the MFMA columns (n); BOTH are presented
// K-contiguous (a "NT" GEMM: C[m][n] = sum_k A[m][k] * B[n][k]), so each lane's fragment is one 16-byte
// LDS read (A[row = lane&15][k = 8*(lane>>4) .. +7]).
//
// LDS image per operand array and stage: [rows][32] bf16 = 64-byte rows, 16-byte chunk index XORed with
// swz(row) = {0,2,3,1}[(row>>2)&3].  With ds_read_b128's four 16-lane service groups on gfx950
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) every group then touches 16 distinct 16-byte slots of
// the 256-byte bank row: conflict-free fragment reads; the 16-byte staging stores cover whole rows.
//
// Pipeline: global -> registers (next k-tile, issued before the MFMAs of the current one) -> LDS
// (double buffered), ONE barrier per k-tile.
//
// NPROD = 1: acc += A0*B0.   NPROD = 3 (split-bf16, near-fp32): acc += A0*B0 + A0*B1 + A1*B0 with
// A0/B0 the bf16 "hi" parts and A1/B1 the bf16 residuals.
#pragma once
#include "common.h"

namespace dali {

__device__ __forceinline__ int lds_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

template <int TM_, int TN_, int NA_, int NB_, int NPROD_>
struct GemmCfg {
    static constexpr int TM = TM_, TN = TN_, NA = NA_, NB = NB_, NPROD = NPROD_;
    static constexpr int BK = 32;
    static constexpr int FM = TM / 32, FN = TN / 32;
    static constexpr int A_ELEMS = TM * BK, B_ELEMS = TN * BK;
    static constexpr int STAGE_ELEMS = NA * A_ELEMS + NB * B_ELEMS;
    static constexpr int LDS_BYTES = 2 * STAGE_ELEMS * 2;
    static constexpr int ACH = TM * 4 / 256, BCH = TN * 4 / 256;   // 16-B chunks per thread per array
    static_assert(TM % 64 == 0 && TN % 64 == 0, "tile must be a multiple of 64");
    static_assert(NPROD == 1 || (NPROD == 3 && NA == 2 && NB == 2), "NPROD 3 needs hi/lo on both operands");
};

// LA / LB: callable (arr, row_in_tile, ktile, chunk) -> uint4 (8 bf16, zero where out of bounds).
template <class Cfg, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop(f32x4_t (&acc)[Cfg::FM][Cfg::FN], const LA& la, const LB& lb,
                                              int ktiles, uint16_t* smem) {
    constexpr int FM = Cfg::FM, FN = Cfg::FN, NA = Cfg::NA, NB = Cfg::NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    uint4 ra[NA][Cfg::ACH], rb[NB][Cfg::BCH];

    auto gload = [&](int kt) {
#pragma unroll
        for (int arr = 0; arr < NA; ++arr)
#pragma unroll
            for (int i = 0; i < Cfg::ACH; ++i) {
                const int c = tid + i * 256;
                ra[arr][i] = la(arr, c >> 2, kt, c & 3);
            }
#pragma unroll
        for (int arr = 0; arr < NB; ++arr)
#pragma unroll
            for (int i = 0; i < Cfg::BCH; ++i) {
                const int c = tid + i * 256;
                rb[arr][i] = lb(arr, c >> 2, kt, c & 3);
            }
    };
    auto sstore = [&](int stage) {
        uint16_t* base = smem + stage * Cfg::STAGE_ELEMS;
#pragma unroll
        for (int arr = 0; arr < NA; ++arr)
#pragma unroll
            for (int i = 0; i < Cfg::ACH; ++i) {
                const int c = tid + i * 256, row = c >> 2, kc = c & 3;
                *reinterpret_cast<uint4*>(base + arr * Cfg::A_ELEMS + row * 32 + ((kc ^ lds_swz(row)) << 3)) = ra[arr][i];
            }
#pragma unroll
        for (int arr = 0; arr < NB; ++arr)
#pragma unroll
            for (int i = 0; i < Cfg::BCH; ++i) {
                const int c = tid + i * 256, row = c >> 2, kc = c & 3;
                *reinterpret_cast<uint4*>(base + NA * Cfg::A_ELEMS + arr * Cfg::B_ELEMS + row * 32 +
                                          ((kc ^ lds_swz(row)) << 3)) = rb[arr][i];
            }
    };

    // per-lane fragment offset inside a 16-row group (frag bases are multiples of 16 rows, so swz only
    // depends on lane&15)
    const int frag_off = (lane & 15) * 32 + (((lane >> 4) ^ lds_swz(lane & 15)) << 3);
    const int a_row0 = wm * (Cfg::TM / 2), b_row0 = wn * (Cfg::TN / 2);

    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const bool more = (kt + 1) < ktiles;
        if (more) gload(kt + 1);
        const uint16_t* sa = smem + (kt & 1) * Cfg::STAGE_ELEMS;
        const uint16_t* sb = sa + NA * Cfg::A_ELEMS;
        bf16x8_t fa[NA][FM];
#pragma unroll
        for (int arr = 0; arr < NA; ++arr)
#pragma unroll
            for (int i = 0; i < FM; ++i)
                fa[arr][i] = *reinterpret_cast<const bf16x8_t*>(sa + arr * Cfg::A_ELEMS + (a_row0 + i * 16) * 32 + frag_off);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            bf16x8_t fb[NB];
#pragma unroll
            for (int arr = 0; arr < NB; ++arr)
                fb[arr] = *reinterpret_cast<const bf16x8_t*>(sb + arr * Cfg::B_ELEMS + (b_row0 + j * 16) * 32 + frag_off);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0], acc[i][j], 0, 0, 0);
                if constexpr (Cfg::NPROD == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[0], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (more) sstore((kt + 1) & 1);
        __syncthreads();
    }
}

// Accumulator element (i, j, r) of this lane is C[m][n] with
//   m = m_tile0 + wm*(TM/2) + i*16 + (lane>>4)*4 + r,   n = n_tile0 + wn*(TN/2) + j*16 + (lane&15).
template <class Cfg>
__device__ __forceinline__ void acc_coords(int& m_base, int& n_base) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    m_base = (wave >> 1) * (Cfg::TM / 2) + (lane >> 4) * 4;
    n_base = (wave & 1) * (Cfg::TN / 2) + (lane & 15);
}

// ---- LDS-DMA helpers (buffer_load ... lds) shared by the conv and distance kernels ----
typedef __attribute__((address_space(3))) void* lds_void_ptr;
constexpr uint32_t DMA_OOB = 0x7ffffff0u;

// wait until at most N of this wave's vector-memory operations (the LDS-DMA loads) are still in flight; also drains
// this wave's LDS reads so the following barrier orders them against the next refill
template <int N>
__device__ __forceinline__ void dma_wait() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else static_assert(N == 0, "unsupported DMA count");
}


// XCD-aware block -> tile map.  Blocks are dealt round-robin to the 8 XCDs (b % 8 labels the XCD group); each group
// walks 64-tile "super-tiles" of SM x SN tiles so the 64 blocks resident on one XCD share SM A panels and SN B panels
// through that XCD's L2, and the 8 XCDs work on neighbouring super-tiles (shared through the Infinity Cache).
// A super-tile holds 2^lg <= 64 tiles (fewer when the problem has < 1024 tiles, so that all 8 XCDs get work);
// SM = min(8, next power of two >= tiles_m), SN = 2^lg / SM: with fewer than 8 tile rows (most convolutions: Cm < 1024)
// a fixed 8 x 8 super-tile would be 50-88 % padding blocks, each of which still costs a workgroup launch and its LDS
// allocation.  Inside a super-tile m runs fastest: the M-tiles of one pixel tile run back to back (the pixel panel is
// fetched from HBM once, and one pixel's output row is completed while its DRAM page is open).
// Returns false for padding blocks.  Speed only, never correctness.
__host__ __device__ __forceinline__ void xcd_super_shape(int tiles_m, int tiles_n, int& lg, int& ms) {
    // super-tile = 2^lg tiles, at most 64 and small enough that every XCD gets at least two of them
    const long long total = (long long)tiles_m * tiles_n;
    lg = total >= 1024 ? 6 : total >= 512 ? 5 : total >= 256 ? 4 : 3;
    ms = tiles_m >= 5 ? 3 : tiles_m >= 3 ? 2 : tiles_m == 2 ? 1 : 0;
    if (ms > lg) ms = lg;
}
// When tiles_m is not a multiple of the super-tile's power-of-two height, the super-tile map pads PERIODICALLY (tiles_m = 3: every 4th
// block of an XCD's sequence is a padding block), and an XCD hands its workgroups to its 4 shader engines round-robin in arrival order:
// one engine then receives nothing but padding blocks.  Measured with HW_ID stamps on ViT's 25216 x 768 outputs (3 x 99 tiles): only
// 192 of the 256 CUs ever ran a tile (scripts/conv_block_timeline.py); tiles_m = 9 (768 x 2304) sends the tiles of every second
// super-tile row to a single engine.  Those shapes take a dense map instead: tiles in the order (band of <= 16 or 8 m-tiles, pixel tile,
// m) -- m fastest, so a pixel panel is still fetched once per XCD -- cut into 8 contiguous ranges, one per XCD; the only padding blocks
// are the < 8 at the very end.
__host__ __device__ __forceinline__ bool xcd_dense_map(int tiles_m, int ms) { return (tiles_m & ((1 << ms) - 1)) != 0; }
__device__ __forceinline__ bool xcd_tile_map(int b, int tiles_m, int tiles_n, int& tm, int& tn) {
    int lg, ms;
    xcd_super_shape(tiles_m, tiles_n, lg, ms);
    if (xcd_dense_map(tiles_m, ms)) {
        const int BM = tiles_m <= 16 ? tiles_m : 8;
        const int T = tiles_m * tiles_n, per = (T + 7) >> 3;
        const int xcd = b & 7, slot = b >> 3;
        const int t = xcd * per + slot;
        if (slot >= per || t >= T) return false;
        const int band_tiles = BM * tiles_n;
        const int band = t / band_tiles;
        const int r = t - band * band_tiles;
        const int left = tiles_m - band * BM;
        const int h = left < BM ? left : BM;
        tn = r / h;
        tm = band * BM + (r - tn * h);
        return true;
    }
    const int ns = lg - ms;
    const int xcd = b & 7, slot = b >> 3;
    const int ssn = (tiles_n + (1 << ns) - 1) >> ns;
    const int st = (slot >> lg) * 8 + xcd;          // super-tile index
    const int w = slot & ((1 << lg) - 1);
    const int sm = st / ssn, sn = st - sm * ssn;
    tm = (sm << ms) + (w & ((1 << ms) - 1));
    tn = (sn << ns) + (w >> ms);
    return tm < tiles_m && tn < tiles_n;
}
inline int xcd_tile_grid(int tiles_m, int tiles_n) {
    int lg, ms;
    xcd_super_shape(tiles_m, tiles_n, lg, ms);
    if (xcd_dense_map(tiles_m, ms)) return ((tiles_m * tiles_n + 7) / 8) * 8;
    const int ns = lg - ms;
    const int ssm = (tiles_m + (1 << ms) - 1) >> ms, ssn = (tiles_n + (1 << ns) - 1) >> ns;
    const int st = ssm * ssn;
    return (((st + 7) / 8) * 8) << lg;
}

}  // namespace dali
