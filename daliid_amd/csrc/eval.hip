// eval.hip -- evaluation path of validateModels.validate (validateModels.py:35-76):
//   row L2 normalisation (:41-42), distmat = 1 - q @ g.T (:47) on MFMA, and the market1501 CMC/mAP
//   arithmetic of torchreid.metrics.evaluate_rank (:68) without a full row sort.
#include "gemm_tile.h"
#include <cstdlib>

namespace dali {

// ------------------------------------------------------------------------------------------------
// Row pre-pass: one wave64 per row.  Optionally normalises (y = x / (|x| + eps)), emits the bf16 operand image
// (zero-padded to Kp columns), |y|^2, and/or the fp32 row.  Operand image layouts (what the distance kernels DMA):
//   layout 1 (bf16):    [n][Kp] bf16, Kp = roundup(d, 64): a 64-deep k-step of one row is one 128-byte cache line;
//   layout 3 (bf16 x3): [n][Kp/32][hi 32 | lo 32] bf16, Kp = roundup(d, 32), lo = bf16(y - hi): the hi and lo parts of a
//                       32-deep k-step of one row share one 128-byte line (the L2 -> LDS path retires whole lines, see conv.hip).
// HBM-bound: reads 4 B/elem once, writes 2-4 B/elem.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_prep_kernel(const float* __restrict__ x, int n, int d, int Kp,
                                                         int normalize, float eps, uint16_t* __restrict__ img,
                                                         int layout, float* __restrict__ sq,
                                                         float* __restrict__ yout, float* __restrict__ norms) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* xr = x + (size_t)row * d;
    float ss = 0.f;
    const bool vec = (d & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (vec && Kp == d && d <= 2048 && !yout && img) {
        // the whole row in registers (<= 8 float4 per lane): all loads in flight at once, one pass over memory.  Same arithmetic, in the same
        // order, as the two-sweep form below (the squares are added chunk by chunk, lane sums by wave_sum).
        float4 v4[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane * 4 + k * 256;
            v4[k] = c < d ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (lane * 4 + k * 256 < d) ss += v4[k].x * v4[k].x + v4[k].y * v4[k].y + v4[k].z * v4[k].z + v4[k].w * v4[k].w;
        ss = wave_sum(ss);
        const float nrm1 = sqrtf(ss);
        const float den1 = normalize ? (nrm1 + eps) : 1.0f;
        if (lane == 0) {
            if (norms) norms[row] = nrm1;
            if (sq) sq[row] = normalize ? (ss / (den1 * den1)) : ss;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane * 4 + k * 256;
            if (c < d) {
                const float v[4] = {v4[k].x / den1, v4[k].y / den1, v4[k].z / den1, v4[k].w / den1};
                uint16_t h[4], l[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    h[t] = f32_to_bf16_bits(v[t]);
                    l[t] = f32_to_bf16_bits(v[t] - bf16_bits_to_f32(h[t]));
                }
                const uint2 hv = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
                if (layout == 3) {
                    uint16_t* dst = img + (size_t)row * (2 * Kp) + (c >> 5) * 64 + (c & 31);
                    const uint2 lv = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
                    *reinterpret_cast<uint2*>(dst) = hv;
                    *reinterpret_cast<uint2*>(dst + 32) = lv;
                } else {
                    *reinterpret_cast<uint2*>(img + (size_t)row * Kp + c) = hv;
                }
            }
        }
        return;
    }
    if (vec) {
        for (int c = lane * 4; c < d; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else {
        for (int c = lane; c < d; c += 64) { const float v = xr[c]; ss += v * v; }
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float den = normalize ? (nrm + eps) : 1.0f;
    if (lane == 0) {
        if (norms) norms[row] = nrm;
        if (sq) sq[row] = normalize ? (ss / (den * den)) : ss;
    }
    if (vec && Kp == d && (reinterpret_cast<uintptr_t>(yout) & 15) == 0) {
        // second sweep, whole float4 chunks (d % 4 == 0, no column padding): 16-byte loads of the L1 / L2-resident row, no per-element
        // predicates (the scalar form below issued four predicated dword loads per chunk, each behind its own branch)
        for (int c = lane * 4; c < d; c += 256) {
            const float4 x4 = *reinterpret_cast<const float4*>(xr + c);
            const float v[4] = {x4.x / den, x4.y / den, x4.z / den, x4.w / den};
            if (yout) *reinterpret_cast<float4*>(yout + (size_t)row * d + c) = make_float4(v[0], v[1], v[2], v[3]);
            if (img) {
                uint16_t h[4], l[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    h[t] = f32_to_bf16_bits(v[t]);
                    l[t] = f32_to_bf16_bits(v[t] - bf16_bits_to_f32(h[t]));
                }
                const uint2 hv = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
                if (layout == 3) {
                    uint16_t* dst = img + (size_t)row * (2 * Kp) + (c >> 5) * 64 + (c & 31);
                    const uint2 lv = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
                    *reinterpret_cast<uint2*>(dst) = hv;
                    *reinterpret_cast<uint2*>(dst + 32) = lv;
                } else {
                    *reinterpret_cast<uint2*>(img + (size_t)row * Kp + c) = hv;
                }
            }
        }
        return;
    }
    // second sweep (the row is L1/L2 resident): write outputs
    for (int c = lane * 4; c < Kp; c += 256) {
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (c + t < d) ? xr[c + t] / den : 0.f;
        if (yout) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (c + t < d) yout[(size_t)row * d + c + t] = v[t];
        }
        if (img) {
            uint16_t h[4], l[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                h[t] = f32_to_bf16_bits(v[t]);
                l[t] = f32_to_bf16_bits(v[t] - bf16_bits_to_f32(h[t]));
            }
            const uint2 hv = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            if (layout == 3) {
                uint16_t* dst = img + (size_t)row * (2 * Kp) + (c >> 5) * 64 + (c & 31);
                const uint2 lv = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
                *reinterpret_cast<uint2*>(dst) = hv;
                *reinterpret_cast<uint2*>(dst + 32) = lv;
            } else {
                *reinterpret_cast<uint2*>(img + (size_t)row * Kp + c) = hv;
            }
        }
    }
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          int n, int d, float eps, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* xr = x + (size_t)row * d;
    const float* gr = dy + (size_t)row * d;
    float ss = 0.f, xg = 0.f;
#pragma unroll 8
    for (int c = lane; c < d; c += 64) { const float v = xr[c]; ss += v * v; xg += v * gr[c]; }
    ss = wave_sum(ss); xg = wave_sum(xg);
    const float nrm = sqrtf(ss), den = nrm + eps;
    const float a = 1.0f / den;
    const float b = (nrm > 0.f) ? xg / (nrm * den * den) : 0.f;
#pragma unroll 8
    for (int c = lane; c < d; c += 64) dx[(size_t)row * d + c] = gr[c] * a - xr[c] * b;
}

// ------------------------------------------------------------------------------------------------
// Pair distance: out[q][g] = 1 - Q[q].G[g]   (or |q|^2 + |g|^2 - 2 q.g)
// MFMA rows (m) = gallery, MFMA columns (n) = query, so each lane's 4 accumulator registers are 4
// consecutive gallery entries of one query row: one 16-byte store.
// Algorithmic work: 2*d FLOP per pair (x3 MFMA issue in split-bf16 mode); bound: MFMA.
// ------------------------------------------------------------------------------------------------
template <int NPROD>
struct PairCfg { using type = GemmCfg<128, 128, (NPROD == 3 ? 2 : 1), (NPROD == 3 ? 2 : 1), NPROD>; };

struct RowLoader {                       // 32-deep k-tile kt of row `row` of an operand image; arr 1 = the lo part (layout 3)
    const uint16_t* p;
    int row0, nrows, pitch, kstep;
    __device__ __forceinline__ uint4 operator()(int arr, int row, int kt, int kc) const {
        const int r = row0 + row;
        if (r >= nrows) return make_uint4(0, 0, 0, 0);
        return *reinterpret_cast<const uint4*>(p + (size_t)r * pitch + kt * kstep + arr * 32 + kc * 8);
    }
};

// Two-model fusion epilogue (evaluateCleanATModels.py:154-157): with `out` holding the first model's distances d1,
//   out = (w1*d1 + w2*d2) / (w1 + w2),  w_m[q,g] = max(qmag_m[q], gmag_m[g])   (null magnitudes: weight 1 = the
// "simple ensemble" (d1+d2)/2 of :126), d2 = this launch's 1 - q.g.
struct PairBlend {
    const float* qmag1; const float* gmag1;
    const float* qmag2; const float* gmag2;
    int on;
};

// Epilogue shared by the distance kernels: metric, optional two-model blend, fp32 store (4 consecutive gallery entries per lane).
// mb / nb: this lane's first gallery row / query column inside the tile (accumulator layout: rows (lane>>4)*4 + r, column lane&15).
template <int FM, int FN>
__device__ __forceinline__ void pairdist_epilogue(f32x4_t (&acc)[FM][FN], int g_tile0, int q_tile0, int mb, int nb, const float* __restrict__ gsq,
                                                  const float* __restrict__ qsq, int ng, int nq, int metric, float* out, const PairBlend& blend) {
    const bool vec_ok = (ng & 3) == 0;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int q = q_tile0 + nb + j * 16;
        if (q >= nq) continue;
        const float qq = (metric == DALI_METRIC_L2SQ) ? qsq[q] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int g0 = g_tile0 + mb + i * 16;
            if (g0 >= ng) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dot = acc[i][j][r];
                if (metric == DALI_METRIC_L2SQ) {
                    const float gg = (g0 + r < ng) ? gsq[g0 + r] : 0.f;
                    v[r] = qq + gg - 2.0f * dot;
                } else if (metric == DALI_METRIC_DOT) {
                    v[r] = dot;
                } else {
                    v[r] = 1.0f - dot;
                }
            }
            float* o = out + (size_t)q * ng + g0;
            if (blend.on) {
                const float qm1 = blend.qmag1 ? blend.qmag1[q] : 1.f, qm2 = blend.qmag2 ? blend.qmag2[q] : 1.f;
                float prev[4];
                if (vec_ok && g0 + 3 < ng) {
                    const float4 p4 = *reinterpret_cast<const float4*>(o);
                    prev[0] = p4.x; prev[1] = p4.y; prev[2] = p4.z; prev[3] = p4.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) prev[r] = (g0 + r < ng) ? o[r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = (g0 + r < ng) ? g0 + r : g0;
                    const float w1 = fmaxf(qm1, blend.gmag1 ? blend.gmag1[g] : 1.f);
                    const float w2 = fmaxf(qm2, blend.gmag2 ? blend.gmag2[g] : 1.f);
                    v[r] = (w1 * prev[r] + w2 * v[r]) / (w1 + w2);
                }
            }
            if (vec_ok && g0 + 3 < ng) {
                // non-temporal: the 4 GB result streams past the Infinity Cache instead of evicting the operand panels every XCD re-reads each round
                // (PMC: 24.7 GB fetched behind the L2 per launch); 10k x 100k x 2048: 9.77 -> 9.44 ms, the ranking kernel behind it unchanged
                typedef float f32x4_nt __attribute__((ext_vector_type(4)));
                const f32x4_nt vv = {v[0], v[1], v[2], v[3]};
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4_nt*>(o));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (g0 + r < ng) o[r] = v[r];
            }
        }
    }
}

template <int NPROD>
__global__ __launch_bounds__(256) void pairdist_kernel(const uint16_t* __restrict__ G, const uint16_t* __restrict__ Q,
                                                       const float* __restrict__ gsq, const float* __restrict__ qsq,
                                                       int ng, int nq, int Kp, int metric, float* out,
                                                       int tiles_m, int tiles_n, PairBlend blend) {
    using Cfg = typename PairCfg<NPROD>::type;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    int tm, tn;
    if (!xcd_tile_map(blockIdx.x, tiles_m, tiles_n, tm, tn)) return;
    f32x4_t acc[Cfg::FM][Cfg::FN];
#pragma unroll
    for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    RowLoader la{G, tm * Cfg::TM, ng, NPROD == 3 ? 2 * Kp : Kp, NPROD == 3 ? 64 : 32};
    RowLoader lb{Q, tn * Cfg::TN, nq, NPROD == 3 ? 2 * Kp : Kp, NPROD == 3 ? 64 : 32};
    gemm_mainloop<Cfg>(acc, la, lb, Kp / 32, smem);

    int mb, nb;
    acc_coords<Cfg>(mb, nb);
    pairdist_epilogue<Cfg::FM, Cfg::FN>(acc, tm * Cfg::TM, tn * Cfg::TN, mb, nb, gsq, qsq, ng, nq, metric, out, blend);
}

// Epilogue of the persistent kernel for interior tiles: whole 128-byte lines per store instruction.  In the accumulator layout a store
// instruction covers 16 query rows x 64 bytes (half a line per row); streamed non-temporally past the caches those halves reached memory as
// partial-line writes (PMC WRITE_SIZE 5.50 GB per launch for a 4.00 GB result, round 3).  Here each wave passes its 64 x 64 sub-tile through a
// PRIVATE 2 KiB strip of LDS behind the ring (3 x 48 + 8 x 2 = all 160 KiB of the CU) (the three ring stages are all busy during the epilogue: the producers are already fetching the
// next tile), one block of 16 query rows x 32 gallery entries at a time: two ds_write_b128 per lane in the accumulator layout, two ds_read_b128 in
// which 8 adjacent lanes hold one row's 128 bytes, two 16-byte stores of 8 rows x one full line each.  No workgroup barrier (the strip is the
// wave's own; a wave's LDS operations execute in order), no registers live across the k-loop.
constexpr int PAIR_STRIP_PITCH = 128;                         // bytes per staged row; 16-byte chunk c of row q sits at chunk c ^ (q & 7): the 8 lanes of a
constexpr int PAIR_STRIP_BYTES = 16 * PAIR_STRIP_PITCH;       // ds_write_b128 group (8 rows, one chunk) land on 8 distinct bank quads
__device__ __forceinline__ void pairdist_epilogue_lines(f32x4_t (&acc)[4][4], int g0w, int q0w, int lane, const float* __restrict__ gsq,
                                                        const float* __restrict__ qsq, int ng, int metric, float* out, char* strip) {
    // g0w / q0w: first gallery entry / query row of this wave's 64 x 64 sub-tile
    char* wr = strip + (lane & 15) * PAIR_STRIP_PITCH;                            // + ((ii * 4 + (lane >> 4)) ^ (lane & 7)) * 16
    const int wsw = lane & 7, wch = lane >> 4;
    const char* rd = strip + (lane >> 3) * PAIR_STRIP_PITCH + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);   // + t * 8 * PITCH (rows + 8: same swizzle)
    float* obase = out + (size_t)(q0w + (lane >> 3)) * ng + g0w + (lane & 7) * 4;
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float qq = 0.f;
        if (metric == DALI_METRIC_L2SQ) qq = qsq[q0w + j * 16 + (lane & 15)];
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = ih * 2 + ii;
                f32x4_t v;
                if (metric == DALI_METRIC_L2SQ) {
                    const float4 gg = *reinterpret_cast<const float4*>(gsq + g0w + i * 16 + (lane >> 4) * 4);
                    v = f32x4_t{qq + gg.x - 2.0f * acc[i][j][0], qq + gg.y - 2.0f * acc[i][j][1], qq + gg.z - 2.0f * acc[i][j][2], qq + gg.w - 2.0f * acc[i][j][3]};
                } else if (metric == DALI_METRIC_DOT) {
                    v = acc[i][j];
                } else {
                    v = f32x4_t{1.0f - acc[i][j][0], 1.0f - acc[i][j][1], 1.0f - acc[i][j][2], 1.0f - acc[i][j][3]};
                }
                *reinterpret_cast<f32x4_t*>(wr + (((ii * 4 + wch) ^ wsw) << 4)) = v;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(rd + t * 8 * PAIR_STRIP_PITCH);
                const f32x4_nt vv = {v[0], v[1], v[2], v[3]};
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4_nt*>(obase + (size_t)(j * 16 + t * 8) * ng + ih * 32));
            }
        }
    }
}

// LDS-DMA version (what the launcher uses when the operands fit 32-bit buffer offsets): 128 gallery rows x 256 query
// rows per 8-wave block (wave (wm, wn) of the 2 x 4 grid owns 64 x 64).  A k-step takes one 128-byte line of every operand
// row ([hi 32 | lo 32] for NPROD = 3, 64 consecutive k for NPROD = 1) by buffer_load_dwordx4 ... lds (a DMA piece = 8 rows,
// no VGPR staging) into a [rows][64] LDS image (physical 16-byte chunk = logical ^ ((row >> 1) & 7), as the k-tile-64 conv
// kernel); 48 KiB per stage, 3-stage ring with counted vmcnt.  Per k-step a wave reads 16 fragments for 48 (NPROD = 3: hi.hi +
// hi.lo + lo.hi) or 32 MFMAs.
// Wave specialisation: waves 8..15 only issue the DMA pieces (60-185 issue cycles each; 6 per wave and k-step used to sit in
// the MFMA waves' instruction streams and the feed time ADDED to the MFMA time: compute-only 0.89 us per k-step, with the feed
// 1.17 us), waves 0..7 only read fragments and issue MFMAs: 1.03 us per k-step (bf16 x3), 0.45 -> 0.37 us (bf16).
// Persistent: one workgroup per CU walks its tiles; the k-steps of all tiles form one stream through the ring, so the next
// tile's first k-steps are fetched under the stores of the finished one.
// Measured and not kept: the two waves of a SIMD issuing their DMA pieces at different points of the k-step (no change); a
// software-pipelined loop (wait + barrier half way through the k-step, next tile's first fragments fetched under the second
// half of the MFMAs: the DMA lead shrinks from two k-steps to one, +5 % / +18 % time); the tile staged through the free ring
// stage and stored as 512-byte contiguous rows (the stores cost 8 of the 11 us a tile spends outside its k-loop, found by
// removing them; staged they cost 6, but the kernel reaches the 128-VGPR cap, spills 15 registers around the k-loop and the
// k-step slows by 10 %).
// Round 4: the tile now leaves through a private 2 KiB strip per consumer wave behind the ring (pairdist_epilogue_lines: whole 128-byte
// lines per store, no spill: 123 VGPRs): 9.03 -> 8.72 ms (bf16 3.99 -> 3.62).  Measured and removed again: (a) 8 consumer + 4 producer waves
// (3 waves per SIMD, 168 VGPRs): 8.93 against 8.91 ms, so four producers do feed the ring; (b) on that kernel, three of a tile's four query
// columns kept in 48 registers and handed to the store path one (column, half) block every ktiles / 6 k-steps of the NEXT tile, so that
// the stores run under its MFMAs: 9.41 ms (+5 %; bf16 3.64 against 3.51) -- stores issued inside the k-loop delay the LDS-DMA loads of the
// same CU by more than the exposed burst costs; (c) start phases per XCD (blockIdx.x & 7) instead of per workgroup: 8.69-8.73 against 8.72.
template <int NPROD>
__global__ __launch_bounds__(1024) void pairdist_dma_kernel(const uint16_t* __restrict__ G, const uint16_t* __restrict__ Q,
                                                            const float* __restrict__ gsq, const float* __restrict__ qsq,
                                                            int ng, int nq, int pitch, int ktiles, int metric, float* out,
                                                            int tiles_m, int tiles_n, int vgrid, PairBlend blend) {
    constexpr int TM = 128, TN = 256;
    constexpr int A_ELEMS = TM * 64, B_ELEMS = TN * 64, STAGE = A_ELEMS + B_ELEMS;
    constexpr int A_PIECES = TM / 8, NDMA = (TM + TN) / 8 / 8;                   // 1 KiB DMA pieces: 16 of A, 6 per producer wave
    static_assert(A_PIECES % 8 == 0, "a producer's piece i is a gallery piece for every producer or for none");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Persistent: this workgroup owns the virtual blocks v = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8, so v
    // stays on this XCD's share of the tile map); the k-steps of all its tiles form one stream through the 3-stage ring, so
    // the producers fetch the next tile's first two k-steps while the consumers store the finished tile.
    int n_tiles = 0;
    for (int v = blockIdx.x; v < vgrid; v += gridDim.x) { int tm, tn; n_tiles += xcd_tile_map(v, tiles_m, tiles_n, tm, tn) ? 1 : 0; }
    if (n_tiles == 0) return;
    // (Measured and removed: a staggered start of the workgroups, in 8 phases an eighth of a tile apart per workgroup -- 9.67 -> 10.1-10.5 ms, it gives up
    // the XCD's L2 panel sharing -- or per XCD, blockIdx.x & 7 -- 8.69-8.73 against 8.72 ms.  The stores' 8 us per tile are not burst contention.)
    if (wave >= 8) {
        // ---- producer waves 8..15 (two per SIMD, next to two consumers): nothing but the DMA ring ----
        const int pw = wave - 8;
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(G), 0, ng * pitch * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Q), 0, nq * pitch * 2, 0x00020000);
        // pieces q = pw + 8*i; q < 16: rows 8q.. of the gallery tile, else rows 8(q-16).. of the query tile; both piece indices
        // have the parity of pw, so one logical chunk per lane serves all of them
        const int r_in = lane >> 3;
        const int kc = (lane & 7) ^ (((pw & 1) << 2) | (r_in >> 1));
        uint32_t off[NDMA];
        int v_next = blockIdx.x, kt_next = 0, st_fill = 0;            // the step the next issue() fetches
        auto seek_tile = [&]() {                                      // advance v_next to the next valid tile and load its row offsets
            int tm = 0, tn = 0;
            while (!xcd_tile_map(v_next, tiles_m, tiles_n, tm, tn)) v_next += gridDim.x;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int q = pw + 8 * i;
                const bool is_a = q < A_PIECES;
                const int row = is_a ? tm * TM + q * 8 + r_in : tn * TN + (q - A_PIECES) * 8 + r_in;
                off[i] = (row < (is_a ? ng : nq)) ? (uint32_t)(row * pitch + kc * 8) * 2u : DMA_OOB;
            }
        };
        auto issue = [&]() {
            if (kt_next == 0) seek_tile();
            uint16_t* base = smem + st_fill * STAGE;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const uint32_t o = (off[i] == DMA_OOB) ? DMA_OOB : off[i] + (uint32_t)(kt_next * 128);
                uint16_t* dst = base + (pw + 8 * i) * 512;
                if (8 * i + 7 < A_PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_void_ptr)dst, 16, o, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (lds_void_ptr)dst, 16, o, 0, 0, 0);
            }
            st_fill = (st_fill == 2) ? 0 : st_fill + 1;
            if (++kt_next == ktiles) { kt_next = 0; v_next += gridDim.x; }
        };
        const int steps = n_tiles * ktiles;
        issue();
        if (steps > 1) { issue(); dma_wait<NDMA>(); } else dma_wait<0>();
        __builtin_amdgcn_s_barrier();                              // step 0 visible
        for (int s = 0; s < steps; ++s) {
            // the stage of step s+2 was last read in step s-1, which every consumer left through the previous barrier
            if (s + 2 < steps) { issue(); dma_wait<NDMA>(); } else dma_wait<0>();     // step s+1 has landed
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    // ---- consumer waves 0..7: fragments + MFMAs + the tile's stores ----
    const int wm = wave >> 2, wn = wave & 3;
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 3);   // first half; second half = ^ 32
    __builtin_amdgcn_s_barrier();
    int st_cur = 0;
    for (int v = blockIdx.x; v < vgrid; v += gridDim.x) {
        int tm, tn;
        if (!xcd_tile_map(v, tiles_m, tiles_n, tm, tn)) continue;
        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < ktiles; ++kt) {
            const uint16_t* sa = smem + st_cur * STAGE;
            const uint16_t* sb = sa + A_ELEMS;
            bf16x8_t a0[4], a1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0[i] = *reinterpret_cast<const bf16x8_t*>(sa + (wm * 64 + i * 16) * 64 + frag_off);
                a1[i] = *reinterpret_cast<const bf16x8_t*>(sa + (wm * 64 + i * 16) * 64 + (frag_off ^ 32));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(sb + (wn * 64 + j * 16) * 64 + frag_off);
                const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(sb + (wn * 64 + j * 16) * 64 + (frag_off ^ 32));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0, acc[i][j], 0, 0, 0);
                    if (NPROD == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b1, acc[i][j], 0, 0, 0);   // hi . lo
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b0, acc[i][j], 0, 0, 0);   // lo . hi
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1, acc[i][j], 0, 0, 0);   // next 32 k
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of this stage are complete before it can be refilled
            __builtin_amdgcn_s_barrier();
            st_cur = (st_cur == 2) ? 0 : st_cur + 1;
        }
        if (!blend.on && (ng & 31) == 0 && (tm + 1) * TM <= ng && (tn + 1) * TN <= nq)
            pairdist_epilogue_lines(acc, tm * TM + wm * 64, tn * TN + wn * 64, lane, gsq, qsq, ng, metric, out,
                                    reinterpret_cast<char*>(smem + 3 * STAGE) + wave * PAIR_STRIP_BYTES);
        else
            pairdist_epilogue<4, 4>(acc, tm * TM, tn * TN, wm * 64 + (lane >> 4) * 4, wn * 64 + (lane & 15), gsq, qsq, ng, nq, metric, out, blend);
    }
}

// ------------------------------------------------------------------------------------------------
// market1501 ranking without a row sort.  One 256-thread block per query.
//   kept(g)  = !(g_pid == q_pid && g_cam == q_cam)                  (junk removal)
//   match(g) = kept(g) && g_pid == q_pid
// The gallery is indexed by identity ONCE per evaluation (rank_index_*: counting sort of the gallery positions by pid, on
// the device), so a query finds its same-identity entries -- matches and junk -- as one slice of that index instead of
// scanning all ng ids (10k x 100k: 4 GB of id reads gone; only the distance row is read, 4 bytes per pair).
// Sort the matches by key (dist, index) in LDS (bitonic).  Every kept gallery entry is binned by the
// number of matches with a smaller key (binary search); with c[b] the bin counts,
//   position (1-based, among kept) of the j-th match = c[0] + ... + c[j]
//   AP = mean_j (j+1) / position_j ,  first-hit rank = c[0] - 1.
// LDS is sized in two tiers (the row scan is HBM-bound and needs many workgroups per CU in flight): RANK_PSMALL same-identity
// entries per query in the first launch (16 KiB of LDS: 8+ workgroups per CU); queries with more are flagged and redone by a
// second launch with room for RANK_PMAX; beyond that DALI_ERR_LIMIT through status[0].
// ------------------------------------------------------------------------------------------------
constexpr int RANK_PSMALL = 512, RANK_PMAX = 4096, RANK_BINS = 1024;
constexpr int RANK_MAX_PID_RANGE = 1 << 20;      // identity codes must span at most this range (the mirrors pass dense codes): status 2 otherwise

__device__ __forceinline__ bool key_less(float da, int ia, float db, int ib) {
    return da < db || (da == db && ia < ib);
}
// (dist, index) as one unsigned 64-bit key with the same order as key_less: IEEE bits made monotonic (negative values
// flipped, sign bit set on the others; -0.0 is folded onto +0.0 first so that equal distances compare by index)
__device__ __forceinline__ unsigned long long rank_key(float d, int g) {
    unsigned int b = __float_as_uint(d + 0.0f);
    b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;
    return ((unsigned long long)b << 32) | (unsigned int)g;
}

// ---- gallery index by identity: info = {min pid, max pid}; counts/starts over [min, max]; order = gallery positions grouped by pid ----
__global__ __launch_bounds__(256) void rank_index_minmax_kernel(const int32_t* __restrict__ g_pids, int ng, int32_t* __restrict__ info) {
    int lo = 0x7fffffff, hi = (int)0x80000000;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) { const int p = g_pids[g]; lo = min(lo, p); hi = max(hi, p); }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&info[0], lo); atomicMax(&info[1], hi); }
}
__global__ __launch_bounds__(256) void rank_index_count_kernel(const int32_t* __restrict__ g_pids, int ng, const int32_t* __restrict__ info,
                                                                int32_t* __restrict__ counts, int32_t* __restrict__ status) {
    const int lo = info[0];
    const long long range = (long long)info[1] - lo + 1;
    if (range > RANK_MAX_PID_RANGE) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(status, 2); return; }
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) atomicAdd(&counts[g_pids[g] - lo], 1);
}
// one block: starts[r] = exclusive prefix of counts (range + 1 entries), cursors zeroed
__global__ __launch_bounds__(1024) void rank_index_scan_kernel(const int32_t* __restrict__ info, const int32_t* __restrict__ counts,
                                                               int32_t* __restrict__ starts, int32_t* __restrict__ cursor) {
    __shared__ int s_part[1024];
    const long long range64 = (long long)info[1] - info[0] + 1;
    if (range64 > RANK_MAX_PID_RANGE) return;
    const int range = (int)range64, tid = threadIdx.x;
    const int per = (range + 1023) / 1024, b = tid * per, e = min(b + per, range);
    int sum = 0;
    for (int r = b; r < e; ++r) sum += counts[r];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int add = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int r = b; r < e; ++r) { starts[r] = run; cursor[r] = 0; run += counts[r]; }
    if (tid == 1023) starts[range] = s_part[1023];
}
__global__ __launch_bounds__(256) void rank_index_scatter_kernel(const int32_t* __restrict__ g_pids, int ng, const int32_t* __restrict__ info,
                                                                  const int32_t* __restrict__ starts, int32_t* __restrict__ cursor,
                                                                  int32_t* __restrict__ order) {
    const int lo = info[0];
    if ((long long)info[1] - lo + 1 > RANK_MAX_PID_RANGE) return;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        const int r = g_pids[g] - lo;
        order[starts[r] + atomicAdd(&cursor[r], 1)] = g;          // order inside an identity is irrelevant: matches are sorted by key below
    }
}

// PASS 0: every query, LDS for PCAP = RANK_PSMALL; larger identities set pending[q].  PASS 1: only the pending queries, PCAP = RANK_PMAX.
template <int PCAP, int PASS>
__global__ __launch_bounds__(256) void rank_query_kernel(const float* __restrict__ distmat, const int32_t* __restrict__ q_pids,
                                                          const int32_t* __restrict__ q_cams, const int32_t* __restrict__ g_cams,
                                                          const int32_t* __restrict__ info, const int32_t* __restrict__ starts,
                                                          const int32_t* __restrict__ order, int nq, int ng,
                                                          float* __restrict__ ap_out, int32_t* __restrict__ first_rank,
                                                          int32_t* __restrict__ pending, int32_t* __restrict__ status) {
    __shared__ unsigned long long s_key[PCAP];    // (orderable distance bits << 32) | gallery index: one 8-byte LDS read per compare
    __shared__ unsigned s_cell[RANK_BINS];
    __shared__ int s_cnt[PCAP + 1];
    __shared__ int s_junk[PCAP];
    __shared__ int s_n, s_nj;
    __shared__ float s_red[4];
    __shared__ int s_scan[256];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (PASS == 1 && pending[q] == 0) return;
    const float* drow = distmat + (size_t)q * ng;
    const int qp = q_pids[q], qc = q_cams[q];
    // 1. this query's identity slice of the gallery index: matches (other camera) and junk (same camera)
    const int lo = info[0], hi = info[1];
    if ((long long)hi - lo + 1 > RANK_MAX_PID_RANGE) { if (tid == 0) { ap_out[q] = 0.f; first_rank[q] = -1; } return; }     // status 2 set by the index build
    int sb = 0, se = 0;
    if (qp >= lo && qp <= hi) { sb = starts[qp - lo]; se = starts[qp - lo + 1]; }
    const int nsame = se - sb;
    if (nsame == 0) {
        if (tid == 0) { ap_out[q] = 0.f; first_rank[q] = -1; if (PASS == 0) pending[q] = 0; }
        return;
    }
    if (nsame > PCAP) {
        if (tid == 0) {
            ap_out[q] = 0.f; first_rank[q] = -1;
            if (PASS == 0) pending[q] = 1; else atomicMax(status, 1);
        }
        return;
    }
    if (tid == 0) { s_n = 0; s_nj = 0; if (PASS == 0) pending[q] = 0; }
    __syncthreads();
    for (int t = tid; t < nsame; t += 256) {
        const int g = order[sb + t];
        if (g_cams[g] != qc) s_key[atomicAdd(&s_n, 1)] = rank_key(drow[g], g);
        else s_junk[atomicAdd(&s_nj, 1)] = g;
    }
    __syncthreads();
    const int np = s_n, nj = s_nj;
    if (np == 0) {
        if (tid == 0) { ap_out[q] = 0.f; first_rank[q] = -1; }
        return;
    }
    const bool vec = (ng & 3) == 0 && (reinterpret_cast<uintptr_t>(drow) & 15) == 0;
    int npad = 1;
    while (npad < np) npad <<= 1;
    for (int t = np + tid; t < npad; t += 256) s_key[t] = ~0ull;                  // above every real key
    for (int t = tid; t <= np; t += 256) s_cnt[t] = 0;
    __syncthreads();
    // 2. bitonic sort of (dist, idx)
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < npad; t += 256) {
                const int p = t ^ j;
                if (p > t) {
                    const bool up = (t & k) == 0;
                    const unsigned long long ka = s_key[t], kb = s_key[p];
                    if (up ? kb < ka : ka < kb) { s_key[t] = kb; s_key[p] = ka; }
                }
            }
            __syncthreads();
        }
    }
    // 3. bin EVERY gallery entry by the number of matches with a smaller key (only the distance row is read: 4 bytes per pair, coalesced
    //    16 B per lane), then take the junk entries back out of their bins.
    //    The lower bound over the sorted matches is NOT searched per entry (7 dependent LDS reads per entry at ~100 matches left the pass
    //    LDS-bound on random distances: 1.7 ms for 10k x 100k against 0.8 ms of HBM time).  The distance axis between the first and the
    //    last match is cut into RANK_BINS uniform cells; s_cell[c] = (matches in lower cells) | (matches in cell c) << 16.  An entry reads
    //    its cell's word: that IS its lower bound unless the cell holds matches itself (about one cell in ten), where a short search
    //    inside the cell's matches finishes it.  floor((d - lo) * scale) is monotone in d, so cells never reorder keys.
    const unsigned long long last_key = s_key[np - 1];
    auto key_dist = [](unsigned long long k) { unsigned b = (unsigned)(k >> 32); b = (b & 0x80000000u) ? (b ^ 0x80000000u) : ~b; return __uint_as_float(b); };
    const float dlo = key_dist(s_key[0]), dhi = key_dist(last_key);
    const float cscale = dhi > dlo ? (float)RANK_BINS / (dhi - dlo) : 0.f;
    auto cell_of = [&](float dn) { const unsigned c = (unsigned)(int)((dn - dlo) * cscale); return (int)(c < (unsigned)RANK_BINS ? c : RANK_BINS - 1); };   // (always in bounds)
    for (int t = tid; t < RANK_BINS; t += 256) s_cell[t] = 0;
    __syncthreads();
    for (int t = tid; t < np; t += 256) atomicAdd(&s_cell[cell_of(key_dist(s_key[t]))], 1u << 16);
    __syncthreads();
    {   // exclusive scan of the per-cell match counts into the low halves (RANK_BINS = 4 * 256: four cells per thread)
        unsigned c4[4], run = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { c4[u] = s_cell[tid * 4 + u] >> 16; run += c4[u]; }
        s_scan[tid] = (int)run;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int add = (tid >= o) ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        unsigned base = (unsigned)s_scan[tid] - run;
#pragma unroll
        for (int u = 0; u < 4; ++u) { s_cell[tid * 4 + u] = base | (c4[u] << 16); base += c4[u]; }
        __syncthreads();
    }
    auto lower_bound_in = [&](unsigned long long k, float dn) {          // number of matches with a key below k, for dlo <= dn and k <= last_key
        const unsigned cw = s_cell[cell_of(dn)];
        int p = (int)(cw & 0xffffu), len = (int)(cw >> 16);
        while (len > 0) {                                               // only cells that hold matches: a search among THEIR keys
            const int half = len >> 1;
            if (s_key[p + half] < k) { p += half + 1; len -= half + 1; } else len = half;
        }
        return p;
    };
    auto bin = [&](float d, int g, int delta) {
        const float dn = d + 0.0f;
        const unsigned long long k = rank_key(d, g);
        if (k > last_key) return;                                       // beyond the last match: affects no position
        atomicAdd(&s_cnt[dn < dlo ? 0 : lower_bound_in(k, dn)], delta);
    };
    auto bin4 = [&](const float4 v, int g, int delta) { bin(v.x, g, delta); bin(v.y, g + 1, delta); bin(v.z, g + 2, delta); bin(v.w, g + 3, delta); };
    if (vec) {
        int g = tid * 4;
        for (; g + 3072 < ng; g += 4096) {                            // 4 independent 16-byte loads in flight
            const float4 v0 = *reinterpret_cast<const float4*>(drow + g);
            const float4 v1 = *reinterpret_cast<const float4*>(drow + g + 1024);
            const float4 v2 = *reinterpret_cast<const float4*>(drow + g + 2048);
            const float4 v3 = *reinterpret_cast<const float4*>(drow + g + 3072);
            bin4(v0, g, 1); bin4(v1, g + 1024, 1); bin4(v2, g + 2048, 1); bin4(v3, g + 3072, 1);
        }
        for (; g < ng; g += 1024) bin4(*reinterpret_cast<const float4*>(drow + g), g, 1);
    } else {
        for (int g = tid; g < ng; g += 256) bin(drow[g], g, 1);
    }
    __syncthreads();
    for (int t = tid; t < nj; t += 256) bin(drow[s_junk[t]], s_junk[t], -1);
    __syncthreads();
    // 4. inclusive scan of the bins + AP (sequential chunks of 256)
    float ap_part = 0.f;
    int carry = 0;
    for (int base = 0; base < np; base += 256) {
        const int t = base + tid;
        const int v = (t < np) ? s_cnt[t] : 0;
        s_scan[tid] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int add = (tid >= o) ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        const int pos = carry + s_scan[tid];
        if (t < np) ap_part += (float)(t + 1) / (float)pos;
        if (t == 0) first_rank[q] = pos - 1;
        carry += s_scan[255];
        __syncthreads();
    }
    ap_part = wave_sum(ap_part);
    if ((tid & 63) == 0) s_red[tid >> 6] = ap_part;
    __syncthreads();
    if (tid == 0) ap_out[q] = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)np;
}

// Single-block deterministic reduction: CMC curve + mAP over valid queries.
__global__ __launch_bounds__(256) void rank_reduce_kernel(const float* __restrict__ ap, const int32_t* __restrict__ first_rank,
                                                           int nq, int max_rank, float* __restrict__ cmc, float* __restrict__ mAP,
                                                           double* __restrict__ map64, int32_t* __restrict__ num_valid) {
    __shared__ double s_sum[256];
    __shared__ int s_valid[256];
    __shared__ int s_hist[1024];
    const int tid = threadIdx.x;
    for (int t = tid; t < 1024; t += 256) s_hist[t] = 0;
    __syncthreads();
    double s = 0.0;
    int nv = 0;
    for (int q = tid; q < nq; q += 256) {
        const int fr = first_rank[q];
        if (fr >= 0) {
            s += (double)ap[q];
            ++nv;
            if (fr < max_rank) atomicAdd(&s_hist[fr], 1);
        }
    }
    s_sum[tid] = s; s_valid[tid] = nv;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_sum[tid] += s_sum[tid + o]; s_valid[tid] += s_valid[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const int n = s_valid[0];
        num_valid[0] = n;
        const double m = n > 0 ? s_sum[0] / (double)n : 0.0;
        mAP[0] = (float)m;
        if (map64) map64[0] = m;
        int run = 0;
        for (int k = 0; k < max_rank; ++k) {
            run += s_hist[k];
            cmc[k] = n > 0 ? (float)((double)run / (double)n) : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Gallery-sharded ranking (the evaluation path over N GPUs): every rank holds the distances of ALL queries to ITS slice of
// the gallery.  The position of a match in the full ranking is 1 + (kept gallery entries of every shard with a smaller key), keys being
// (distance, GLOBAL gallery index), so the merge is a sum of per-shard counts:
//   (1) rank_shard_matches_kernel: the keys of the query's matches inside this shard (identity slice of the shard's index, other camera);
//   (2) [host: all-gather of the keys]
//   (3) rank_shard_bins_kernel: all shards' match keys sorted in LDS (the same order on every rank: keys are unique); every kept entry of THIS
//       shard is binned by the number of matches with a smaller key, exactly as rank_query_kernel bins the whole row;
//   (4) [host: all-reduce SUM of the integer bins]
//   (5) rank_shard_finish_kernel: positions = inclusive scan of the bins, AP and first-hit rank with rank_query_kernel's summation order
//       (bit-identical to the single-GPU result), then rank_reduce_kernel.
// Plain binary search over the sorted keys (no distance cells): this path is bounded by the collectives, not by the bins.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rank_shard_matches_kernel(const float* __restrict__ dist, const int32_t* __restrict__ q_pids,
                                                                  const int32_t* __restrict__ q_cams, const int32_t* __restrict__ g_cams,
                                                                  const int32_t* __restrict__ info, const int32_t* __restrict__ starts,
                                                                  const int32_t* __restrict__ order, int ng, int g_offset, int cap,
                                                                  unsigned long long* __restrict__ keys, int32_t* __restrict__ counts,
                                                                  int32_t* __restrict__ status) {
    __shared__ int s_n;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int lo = info[0], hi = info[1];
    unsigned long long* krow = keys + (size_t)q * cap;
    for (int t = tid; t < cap; t += 256) krow[t] = ~0ull;
    if ((long long)hi - lo + 1 > RANK_MAX_PID_RANGE) { if (tid == 0) counts[q] = 0; return; }        // status 2 set by the index build
    const int qp = q_pids[q], qc = q_cams[q];
    int sb = 0, se = 0;
    if (qp >= lo && qp <= hi) { sb = starts[qp - lo]; se = starts[qp - lo + 1]; }
    if (tid == 0) s_n = 0;
    __syncthreads();
    const float* drow = dist + (size_t)q * ng;
    for (int t = sb + tid; t < se; t += 256) {
        const int g = order[t];
        if (g_cams[g] != qc) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < cap) krow[slot] = rank_key(drow[g], g + g_offset);
        }
    }
    __syncthreads();
    if (tid == 0) {
        counts[q] = s_n < cap ? s_n : cap;
        if (s_n > cap) atomicMax(status, 1);
    }
}

__global__ __launch_bounds__(256) void rank_shard_bins_kernel(const float* __restrict__ dist, const int32_t* __restrict__ q_pids,
                                                               const int32_t* __restrict__ q_cams, const int32_t* __restrict__ g_cams,
                                                               const int32_t* __restrict__ info, const int32_t* __restrict__ starts,
                                                               const int32_t* __restrict__ order, int nq, int ng, int g_offset,
                                                               const unsigned long long* __restrict__ keys_all, const int32_t* __restrict__ counts_all,
                                                               int world, int cap, int32_t* __restrict__ bins, int bins_cap,
                                                               int32_t* __restrict__ status) {
    __shared__ unsigned long long s_key[RANK_PMAX];
    __shared__ int s_cnt[RANK_PMAX + 1];
    const int q = blockIdx.x, tid = threadIdx.x;
    int32_t* brow = bins + (size_t)q * (bins_cap + 1);
    int np = 0;
    for (int r = 0; r < world; ++r) np += counts_all[(size_t)r * nq + q];
    if (np > bins_cap || np > RANK_PMAX) { if (tid == 0) atomicMax(status, 1); np = 0; }
    for (int t = tid; t <= bins_cap; t += 256) brow[t] = 0;
    if (np == 0) return;
    // all shards' match keys of this query, in rank order (any order: they are sorted next)
    int base = 0;
    for (int r = 0; r < world; ++r) {
        const int n = counts_all[(size_t)r * nq + q];
        const unsigned long long* src = keys_all + ((size_t)r * nq + q) * cap;
        for (int t = tid; t < n; t += 256) s_key[base + t] = src[t];
        base += n;
    }
    int npad = 1;
    while (npad < np) npad <<= 1;
    for (int t = np + tid; t < npad; t += 256) s_key[t] = ~0ull;
    for (int t = tid; t <= np; t += 256) s_cnt[t] = 0;
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < npad; t += 256) {
                const int p = t ^ j;
                if (p > t) {
                    const bool up = (t & k) == 0;
                    const unsigned long long ka = s_key[t], kb = s_key[p];
                    if (up ? kb < ka : ka < kb) { s_key[t] = kb; s_key[p] = ka; }
                }
            }
            __syncthreads();
        }
    }
    const unsigned long long last_key = s_key[np - 1];
    auto bin = [&](float d, int g, int delta) {
        const unsigned long long k = rank_key(d, g + g_offset);
        if (k > last_key) return;                                     // beyond the last match: affects no position
        int p = 0, len = np;
        while (len > 0) { const int half = len >> 1; if (s_key[p + half] < k) { p += half + 1; len -= half + 1; } else len = half; }
        atomicAdd(&s_cnt[p], delta);
    };
    const float* drow = dist + (size_t)q * ng;
    for (int g = tid; g < ng; g += 256) bin(drow[g], g, 1);
    // junk of this shard (same identity, same camera) comes back out
    const int lo = info[0], hi = info[1];
    const int qp = q_pids[q], qc = q_cams[q];
    if ((long long)hi - lo + 1 <= RANK_MAX_PID_RANGE && qp >= lo && qp <= hi) {
        const int sb = starts[qp - lo], se = starts[qp - lo + 1];
        __syncthreads();
        for (int t = sb + tid; t < se; t += 256) { const int g = order[t]; if (g_cams[g] == qc) bin(drow[g], g, -1); }
    }
    __syncthreads();
    for (int t = tid; t <= np; t += 256) brow[t] = s_cnt[t];
}

__global__ __launch_bounds__(256) void rank_shard_finish_kernel(const int32_t* __restrict__ bins, const int32_t* __restrict__ counts_all, int world,
                                                                 int nq, int bins_cap, float* __restrict__ ap_out, int32_t* __restrict__ first_rank) {
    __shared__ int s_scan[256];
    __shared__ float s_red[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    int np = 0;
    for (int r = 0; r < world; ++r) np += counts_all[(size_t)r * nq + q];
    if (np == 0 || np > bins_cap) { if (tid == 0) { ap_out[q] = 0.f; first_rank[q] = -1; } return; }
    const int32_t* brow = bins + (size_t)q * (bins_cap + 1);
    // inclusive scan of the bins + AP in sequential chunks of 256: the arithmetic and its order are rank_query_kernel's step 4
    float ap_part = 0.f;
    int carry = 0;
    for (int base = 0; base < np; base += 256) {
        const int t = base + tid;
        const int v = (t < np) ? brow[t] : 0;
        s_scan[tid] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int add = (tid >= o) ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        const int pos = carry + s_scan[tid];
        if (t < np) ap_part += (float)(t + 1) / (float)pos;
        if (t == 0) first_rank[q] = pos - 1;
        carry += s_scan[255];
        __syncthreads();
    }
    ap_part = wave_sum(ap_part);
    if ((tid & 63) == 0) s_red[tid >> 6] = ap_part;
    __syncthreads();
    if (tid == 0) ap_out[q] = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)np;
}

}  // namespace dali

using namespace dali;

extern "C" int dali_l2norm_rows(dali_ctx* ctx, void* stream, const float* x, int n, int d, float eps, float* y,
                                float* norms) {
    DALI_REQUIRE(ctx && x && y, "dali_l2norm_rows: null argument");
    DALI_REQUIRE(n >= 0 && d > 0, "dali_l2norm_rows: bad shape n=%d d=%d", n, d);
    if (n == 0) return DALI_OK;
    hipLaunchKernelGGL(rows_prep_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, n, d, (d + 3) & ~3, 1, eps,
                       (uint16_t*)nullptr, 0, (float*)nullptr, y, norms);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_l2norm_rows_bwd(dali_ctx* ctx, void* stream, const float* x, const float* dy, int n, int d,
                                    float eps, float* dx) {
    DALI_REQUIRE(ctx && x && dy && dx, "dali_l2norm_rows_bwd: null argument");
    DALI_REQUIRE(n >= 0 && d > 0, "dali_l2norm_rows_bwd: bad shape n=%d d=%d", n, d);
    if (n == 0) return DALI_OK;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, n, d, eps, dx);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

// operand image geometry (rows_prep_kernel): columns padded to Kp, `pitch` bf16 elements per row
static inline int pair_kp(int d, bool split) { return split ? (d + 31) & ~31 : (d + 63) & ~63; }
static inline int pair_pitch(int d, bool split) { return split ? 2 * pair_kp(d, true) : pair_kp(d, false); }

static int launch_pairdist(int num_cus, hipStream_t st, const uint16_t* g_img, const float* gsq, const uint16_t* q_img, const float* qsq, int nq, int ng,
                           int d, int metric, bool split, float* out, PairBlend blend = PairBlend{nullptr, nullptr, nullptr, nullptr, 0}) {
    static const bool no_dma = getenv("DALI_PAIRDIST_NODMA") != nullptr;
    const int Kp = pair_kp(d, split), pitch = pair_pitch(d, split);
    if (!no_dma && (long long)ng * pitch * 2 < 0x7ff00000ll && (long long)nq * pitch * 2 < 0x7ff00000ll) {
        const int tm2 = (ng + 127) / 128, tn2 = (nq + 255) / 256;
        const int grid2 = xcd_tile_grid(tm2, tn2);
        const int lds = 3 * (128 + 256) * 64 * 2 + 8 * PAIR_STRIP_BYTES;         // 3 stages x 48 KiB + the consumers' store strips = 160 KiB
        DALI_ONCE_PER_DEVICE({
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pairdist_dma_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pairdist_dma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        });
        const int cap = num_cus / 8 * 8, grid = grid2 < cap ? grid2 : cap;          // persistent: one workgroup per CU (144 KiB of LDS each)
        const int kt = split ? Kp / 32 : Kp / 64;
        if (split) hipLaunchKernelGGL(pairdist_dma_kernel<3>, dim3(grid), dim3(1024), lds, st, g_img, q_img, gsq, qsq, ng, nq, pitch, kt, metric, out, tm2, tn2, grid2, blend);
        else hipLaunchKernelGGL(pairdist_dma_kernel<1>, dim3(grid), dim3(1024), lds, st, g_img, q_img, gsq, qsq, ng, nq, pitch, kt, metric, out, tm2, tn2, grid2, blend);
        DALI_LAUNCH_CHECK();
        return DALI_OK;
    }
    const int tiles_m = (ng + 127) / 128, tiles_n = (nq + 127) / 128;
    const int grid = xcd_tile_grid(tiles_m, tiles_n);
    if (split) {
        using Cfg = PairCfg<3>::type;
        DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pairdist_kernel<3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        hipLaunchKernelGGL(pairdist_kernel<3>, dim3(grid), dim3(256), Cfg::LDS_BYTES, st, g_img, q_img, gsq, qsq, ng, nq, Kp, metric, out, tiles_m, tiles_n, blend);
    } else {
        using Cfg = PairCfg<1>::type;
        hipLaunchKernelGGL(pairdist_kernel<1>, dim3(grid), dim3(256), Cfg::LDS_BYTES, st, g_img, q_img, gsq, qsq, ng, nq, Kp, metric, out, tiles_m, tiles_n, blend);
    }
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" size_t dali_pairdist_operand_bytes(int n, int d, int precision) {
    if (n <= 0 || d <= 0) return 0;
    return (size_t)n * pair_pitch(d, precision == DALI_PREC_BF16X3) * sizeof(uint16_t);
}

extern "C" int dali_pairdist_prepare(dali_ctx* ctx, void* stream, const float* X, int n, int d, int normalize, int precision,
                                     void* image, float* sq) {
    DALI_REQUIRE(ctx && X && image && sq, "dali_pairdist_prepare: null argument");
    DALI_REQUIRE(n >= 0 && d > 0, "dali_pairdist_prepare: bad shape n=%d d=%d", n, d);
    DALI_REQUIRE(precision == DALI_PREC_BF16X3 || precision == DALI_PREC_BF16, "dali_pairdist_prepare: bad precision %d", precision);
    DALI_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, "dali_pairdist_prepare: the operand image must be 16-byte aligned");
    if (n == 0) return DALI_OK;
    const bool split = precision == DALI_PREC_BF16X3;
    hipLaunchKernelGGL(rows_prep_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, n, d, pair_kp(d, split), normalize, 0.0f,
                       static_cast<uint16_t*>(image), split ? 3 : 1, sq, (float*)nullptr, (float*)nullptr);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_pairdist_prepared(dali_ctx* ctx, void* stream, const void* q_image, const float* q_sq, const void* g_image,
                                      const float* g_sq, int nq, int ng, int d, int metric, int precision, float* out) {
    DALI_REQUIRE(ctx && q_image && g_image && q_sq && g_sq && out, "dali_pairdist_prepared: null argument");
    DALI_REQUIRE(nq >= 0 && ng >= 0 && d > 0, "dali_pairdist_prepared: bad shape nq=%d ng=%d d=%d", nq, ng, d);
    DALI_REQUIRE(metric == DALI_METRIC_COSINE || metric == DALI_METRIC_L2SQ || metric == DALI_METRIC_DOT, "dali_pairdist_prepared: bad metric %d", metric);
    DALI_REQUIRE(precision == DALI_PREC_BF16X3 || precision == DALI_PREC_BF16, "dali_pairdist_prepared: bad precision %d", precision);
    DALI_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "dali_pairdist_prepared: out must be 16-byte aligned");
    if (nq == 0 || ng == 0) return DALI_OK;
    return launch_pairdist(ctx->num_cus, (hipStream_t)stream, static_cast<const uint16_t*>(g_image), g_sq, static_cast<const uint16_t*>(q_image), q_sq, nq, ng, d,
                           metric, precision == DALI_PREC_BF16X3, out);
}

// ------------------------------------------------------------------------------------------------
// Small similarity GEMMs (the loss heads' fn @ centers^T, fn @ proxies^T, dS @ centers: 256 x 751 .. 2253 x 2048, losses.py:62, :277) on the fp32
// MFMA: out[m][n] = sum_k Q[m][k] G[n][k], exact fp32 products, no operand pre-pass.  The persistent distance kernel gives a 128 x 256 tile
// to one CU: 6-18 of the 256 CUs worked on these, 46 us per launch + two 8 us operand pre-passes.  Here: one workgroup per 32 x 32 tile
// (192-568 workgroups), its 4 waves split K in 16-deep steps (lane (i, h) takes k = k16 + 8h + t: two 16-byte loads per operand), D steps of
// loads in flight ahead of each wave's MFMA chain, partial tiles summed through LDS in a fixed order.  Rows past nq / ng are clamped onto the
// last row and masked at the store; k-steps past a wave's share are clamped and multiplied by zero; the K % 16 tail is one masked step.
// (Rows are read in 16-byte pieces at 4-byte alignment when ldq / ldg is no multiple of 4 -- d = 751 in the heads' backward: global loads of any
//  width need dword alignment only on gfx950; tests/test_gpu_eval.py covers that shape bit-exactly.)
// ------------------------------------------------------------------------------------------------
namespace dali {
template <int D>
__global__ __launch_bounds__(256) void dot_small_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ G, int ldg, int nq, int ng, int K,
                                                        float* __restrict__ out, int ldo) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const float* ap = Q + (size_t)min(m0 + i, nq - 1) * ldq + 8 * h;
    const float* bp = G + (size_t)min(n0 + i, ng - 1) * ldg + 8 * h;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nfull = K >> 4;                              // whole 16-deep steps; wave w takes steps w, w + 4, ...
    const int ns = (nfull + 3) >> 2;                       // turns per wave (the last may be past a wave's share: clamped, weight 0)
    struct Raw { float4 a0, a1, b0, b1; };
    auto load = [&](int j, Raw& st) {
        const int sc = min(wave + 4 * j, nfull - 1);
        st.a0 = *reinterpret_cast<const float4*>(ap + 16 * sc); st.a1 = *reinterpret_cast<const float4*>(ap + 16 * sc + 4);
        st.b0 = *reinterpret_cast<const float4*>(bp + 16 * sc); st.b1 = *reinterpret_cast<const float4*>(bp + 16 * sc + 4);
    };
    auto fma = [&](const Raw& st, int j) {
        const float z = wave + 4 * j < nfull ? 1.f : 0.f;
        const float a[8] = {st.a0.x * z, st.a0.y * z, st.a0.z * z, st.a0.w * z, st.a1.x * z, st.a1.y * z, st.a1.z * z, st.a1.w * z};
        const float b[8] = {st.b0.x, st.b0.y, st.b0.z, st.b0.w, st.b1.x, st.b1.y, st.b1.z, st.b1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    };
    if (nfull > 0) {
        Raw ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) load(d, ring[d]);
        for (int j0 = 0; j0 < ns; j0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                fma(ring[d], j0 + d);
                load(j0 + d + D, ring[d]);
            }
        }
    }
    if ((K & 15) && wave == 0) {                           // the ragged end of K: one masked step
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int k = 16 * nfull + 8 * h + t;          // (ap / bp already carry the 8h)
            const float a = k < K ? ap[16 * nfull + t] : 0.f, b = k < K ? bp[16 * nfull + t] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int n = n0 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
            if (m < nq && n < ng) out[(size_t)m * ldo + n] = v;
        }
    }
}
// does dali_pairdist take this problem on the small kernel?  plain dot products at fp32 grade whose 128 x 256 tiles would occupy at most a
// quarter of the CUs (DALI_PAIRDIST_SMALL=0: always the persistent kernel, for A/B)
static bool dot_small_applies(int num_cus, int nq, int ng, int d, int metric, int precision, int normalize) {
    if (metric != DALI_METRIC_DOT || normalize || precision != DALI_PREC_BF16X3 || d < 16) return false;
    const long long big_tiles = (long long)((nq + 127) / 128) * ((ng + 255) / 256);
    return big_tiles * 4 <= num_cus && DALI_ENV_INT("DALI_PAIRDIST_SMALL", 1) != 0;
}
static int launch_dot_small(hipStream_t st, const float* Q, const float* G, int nq, int ng, int d, float* out) {
    const dim3 grid((ng + 31) / 32, (nq + 31) / 32);
    const int turns = ((d >> 4) + 3) >> 2;
    if (turns >= 8) hipLaunchKernelGGL(dot_small_kernel<8>, grid, dim3(256), 0, st, Q, d, G, d, nq, ng, d, out, ng);
    else if (turns >= 4) hipLaunchKernelGGL(dot_small_kernel<4>, grid, dim3(256), 0, st, Q, d, G, d, nq, ng, d, out, ng);
    else hipLaunchKernelGGL(dot_small_kernel<2>, grid, dim3(256), 0, st, Q, d, G, d, nq, ng, d, out, ng);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
}  // namespace dali

// gallery image, query image and the squared norms in the context workspace
static int prepare_both(dali_ctx* ctx, void* stream, const float* Q, const float* G, int nq, int ng, int d, int precision, int normalize,
                        uint16_t*& g_img, uint16_t*& q_img, float*& sq) {
    const size_t g_bytes = align_up(dali_pairdist_operand_bytes(ng, d, precision), 256);
    const size_t q_bytes = align_up(dali_pairdist_operand_bytes(nq, d, precision), 256);
    char* ws = static_cast<char*>(workspace(ctx, g_bytes + q_bytes + align_up(((size_t)nq + ng) * sizeof(float), 256)));
    if (!ws) return DALI_ERR_NOMEM;
    g_img = reinterpret_cast<uint16_t*>(ws);
    q_img = reinterpret_cast<uint16_t*>(ws + g_bytes);
    sq = reinterpret_cast<float*>(ws + g_bytes + q_bytes);
    int rc = dali_pairdist_prepare(ctx, stream, G, ng, d, normalize, precision, g_img, sq);          // gallery rows first
    if (rc != DALI_OK) return rc;
    return dali_pairdist_prepare(ctx, stream, Q, nq, d, normalize, precision, q_img, sq + ng);
}

extern "C" int dali_pairdist(dali_ctx* ctx, void* stream, const float* Q, const float* G, int nq, int ng, int d,
                             int metric, int precision, int normalize, float* out) {
    DALI_REQUIRE(ctx && Q && G && out, "dali_pairdist: null argument");
    DALI_REQUIRE(nq >= 0 && ng >= 0 && d > 0, "dali_pairdist: bad shape nq=%d ng=%d d=%d", nq, ng, d);
    DALI_REQUIRE(metric == DALI_METRIC_COSINE || metric == DALI_METRIC_L2SQ || metric == DALI_METRIC_DOT, "dali_pairdist: bad metric %d", metric);
    DALI_REQUIRE(precision == DALI_PREC_BF16X3 || precision == DALI_PREC_BF16, "dali_pairdist: bad precision %d", precision);
    DALI_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "dali_pairdist: out must be 16-byte aligned");
    if (nq == 0 || ng == 0) return DALI_OK;
    if (dot_small_applies(ctx->num_cus, nq, ng, d, metric, precision, normalize)) return launch_dot_small((hipStream_t)stream, Q, G, nq, ng, d, out);
    uint16_t *g_img, *q_img;
    float* sq;
    const int rc = prepare_both(ctx, stream, Q, G, nq, ng, d, precision, normalize, g_img, q_img, sq);
    if (rc != DALI_OK) return rc;
    return launch_pairdist(ctx->num_cus, (hipStream_t)stream, g_img, sq, q_img, sq + ng, nq, ng, d, metric, precision == DALI_PREC_BF16X3, out);
}

extern "C" int dali_pairdist_blend(dali_ctx* ctx, void* stream, const float* Q, const float* G, int nq, int ng, int d,
                                   int precision, int normalize, const float* q_mag_prev, const float* g_mag_prev,
                                   const float* q_mag, const float* g_mag, float* inout) {
    DALI_REQUIRE(ctx && Q && G && inout, "dali_pairdist_blend: null argument");
    DALI_REQUIRE(nq >= 0 && ng >= 0 && d > 0, "dali_pairdist_blend: bad shape nq=%d ng=%d d=%d", nq, ng, d);
    DALI_REQUIRE(precision == DALI_PREC_BF16X3 || precision == DALI_PREC_BF16, "dali_pairdist_blend: bad precision %d", precision);
    DALI_REQUIRE((q_mag_prev == nullptr) == (g_mag_prev == nullptr) && (q_mag == nullptr) == (g_mag == nullptr) &&
                 (q_mag_prev == nullptr) == (q_mag == nullptr),
                 "dali_pairdist_blend: the four magnitude vectors must be all set or all null");
    DALI_REQUIRE((reinterpret_cast<uintptr_t>(inout) & 15) == 0, "dali_pairdist_blend: inout must be 16-byte aligned");
    if (nq == 0 || ng == 0) return DALI_OK;
    uint16_t *g_img, *q_img;
    float* sq;
    const int rc = prepare_both(ctx, stream, Q, G, nq, ng, d, precision, normalize, g_img, q_img, sq);
    if (rc != DALI_OK) return rc;
    return launch_pairdist(ctx->num_cus, (hipStream_t)stream, g_img, sq, q_img, sq + ng, nq, ng, d, DALI_METRIC_COSINE, precision == DALI_PREC_BF16X3, inout,
                           PairBlend{q_mag_prev, g_mag_prev, q_mag, g_mag, 1});
}

// gallery positions counting-sorted by identity (info = {min, max} pid; starts over [min, max]; order = positions grouped by pid)
static int build_gallery_index(hipStream_t st, const int32_t* g_pids, int ng, int32_t* info, int32_t* counts, int32_t* starts, int32_t* cursor,
                               int32_t* order, int32_t* status) {
    const int32_t init[2] = {0x7fffffff, (int32_t)0x80000000};
    DALI_HIP(hipMemcpyAsync(info, init, sizeof(init), hipMemcpyHostToDevice, st));
    const int gb = (ng + 255) / 256 < 1024 ? (ng + 255) / 256 : 1024;
    hipLaunchKernelGGL(rank_index_minmax_kernel, dim3(gb), dim3(256), 0, st, g_pids, ng, info);
    DALI_LAUNCH_CHECK();
    DALI_HIP(hipMemsetAsync(counts, 0, ((size_t)RANK_MAX_PID_RANGE + 1) * 4, st));
    hipLaunchKernelGGL(rank_index_count_kernel, dim3(gb), dim3(256), 0, st, g_pids, ng, info, counts, status);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(rank_index_scan_kernel, dim3(1), dim3(1024), 0, st, info, counts, starts, cursor);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(rank_index_scatter_kernel, dim3(gb), dim3(256), 0, st, g_pids, ng, info, starts, cursor, order);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
// the index of a gallery (shard) in the context workspace: -> info / starts / order
static int shard_index(dali_ctx* ctx, hipStream_t st, const int32_t* g_pids, int ng, int32_t* status, int32_t*& info, int32_t*& starts, int32_t*& order) {
    const size_t b_info = 256, b_order = align_up((size_t)ng * 4, 256), b_tab = align_up(((size_t)RANK_MAX_PID_RANGE + 1) * 4, 256);
    char* ws = static_cast<char*>(workspace(ctx, b_info + b_order + 3 * b_tab));
    if (!ws) return DALI_ERR_NOMEM;
    info = reinterpret_cast<int32_t*>(ws);
    order = reinterpret_cast<int32_t*>(ws + b_info);
    int32_t* counts = reinterpret_cast<int32_t*>(ws + b_info + b_order);
    starts = counts + b_tab / 4;
    return build_gallery_index(st, g_pids, ng, info, counts, starts, starts + b_tab / 4, order, status);
}

extern "C" int dali_rank_shard_matches(dali_ctx* ctx, void* stream, const float* dist_shard, const int32_t* q_pids, const int32_t* g_pids,
                                       const int32_t* q_camids, const int32_t* g_camids, int nq, int ng, int g_offset, int cap,
                                       int64_t* keys, int32_t* counts, int32_t* status) {
    DALI_REQUIRE(ctx && dist_shard && q_pids && g_pids && q_camids && g_camids && keys && counts && status, "dali_rank_shard_matches: null argument");
    DALI_REQUIRE(nq > 0 && ng > 0 && g_offset >= 0 && cap > 0 && cap <= RANK_PMAX, "dali_rank_shard_matches: bad shape nq=%d ng=%d offset=%d cap=%d", nq, ng, g_offset, cap);
    hipStream_t st = (hipStream_t)stream;
    DALI_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    int32_t *info, *starts, *order;
    if (int rc = shard_index(ctx, st, g_pids, ng, status, info, starts, order)) return rc;
    hipLaunchKernelGGL(rank_shard_matches_kernel, dim3(nq), dim3(256), 0, st, dist_shard, q_pids, q_camids, g_camids, info, starts, order, ng, g_offset, cap,
                       reinterpret_cast<unsigned long long*>(keys), counts, status);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_rank_shard_bins(dali_ctx* ctx, void* stream, const float* dist_shard, const int32_t* q_pids, const int32_t* g_pids,
                                    const int32_t* q_camids, const int32_t* g_camids, int nq, int ng, int g_offset, const int64_t* keys_all,
                                    const int32_t* counts_all, int world, int cap, int32_t* bins, int bins_cap, int32_t* status) {
    DALI_REQUIRE(ctx && dist_shard && q_pids && g_pids && q_camids && g_camids && keys_all && counts_all && bins && status, "dali_rank_shard_bins: null argument");
    DALI_REQUIRE(nq > 0 && ng > 0 && g_offset >= 0 && world > 0 && cap > 0 && bins_cap > 0 && bins_cap <= RANK_PMAX,
                 "dali_rank_shard_bins: bad shape nq=%d ng=%d world=%d cap=%d bins_cap=%d (<= %d)", nq, ng, world, cap, bins_cap, RANK_PMAX);
    hipStream_t st = (hipStream_t)stream;
    DALI_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    int32_t *info, *starts, *order;
    if (int rc = shard_index(ctx, st, g_pids, ng, status, info, starts, order)) return rc;
    hipLaunchKernelGGL(rank_shard_bins_kernel, dim3(nq), dim3(256), 0, st, dist_shard, q_pids, q_camids, g_camids, info, starts, order, nq, ng, g_offset,
                       reinterpret_cast<const unsigned long long*>(keys_all), counts_all, world, cap, bins, bins_cap, status);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_rank_shard_finish(dali_ctx* ctx, void* stream, const int32_t* bins, const int32_t* counts_all, int world, int nq, int bins_cap,
                                      int max_rank, float* cmc, float* mAP, double* map64, int32_t* num_valid, float* ap, int32_t* first_rank) {
    DALI_REQUIRE(ctx && bins && counts_all && cmc && mAP && num_valid && ap && first_rank, "dali_rank_shard_finish: null argument");
    DALI_REQUIRE(nq > 0 && world > 0 && bins_cap > 0 && max_rank > 0 && max_rank <= 1024, "dali_rank_shard_finish: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rank_shard_finish_kernel, dim3(nq), dim3(256), 0, st, bins, counts_all, world, nq, bins_cap, ap, first_rank);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(rank_reduce_kernel, dim3(1), dim3(256), 0, st, ap, first_rank, nq, max_rank, cmc, mAP, map64, num_valid);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_rank_eval(dali_ctx* ctx, void* stream, const float* distmat, const int32_t* q_pids,
                              const int32_t* g_pids, const int32_t* q_camids, const int32_t* g_camids, int nq, int ng,
                              int max_rank, float* cmc, float* mAP, double* map64, int32_t* num_valid, float* ap,
                              int32_t* first_rank, int32_t* status) {
    DALI_REQUIRE(ctx && distmat && q_pids && g_pids && q_camids && g_camids && cmc && mAP && num_valid && status,
                 "dali_rank_eval: null argument");
    DALI_REQUIRE(nq > 0 && ng > 0, "dali_rank_eval: bad shape nq=%d ng=%d", nq, ng);
    DALI_REQUIRE(max_rank > 0 && max_rank <= 1024, "dali_rank_eval: max_rank %d outside 1..1024", max_rank);
    hipStream_t st = (hipStream_t)stream;
    float* ap_buf = ap;
    int32_t* fr_buf = first_rank;
    // gallery index by identity (counting sort over [min pid, max pid]) in the context workspace, behind the ap / first_rank scratch
    DALI_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    {
        const size_t head = (!ap || !first_rank) ? align_up((size_t)nq * 4, 256) * 2 : 0;
        const size_t b_info = 256, b_order = align_up((size_t)ng * 4, 256), b_pend = align_up((size_t)nq * 4, 256);
        const size_t b_tab = align_up(((size_t)RANK_MAX_PID_RANGE + 1) * 4, 256);
        char* ws = static_cast<char*>(workspace(ctx, head + b_info + b_order + b_pend + 3 * b_tab));
        if (!ws) return DALI_ERR_NOMEM;
        if (!ap) ap_buf = reinterpret_cast<float*>(ws);
        if (!first_rank) fr_buf = reinterpret_cast<int32_t*>(ws + align_up((size_t)nq * 4, 256));
        int32_t* info = reinterpret_cast<int32_t*>(ws + head);
        int32_t* order = reinterpret_cast<int32_t*>(ws + head + b_info);
        int32_t* pending = reinterpret_cast<int32_t*>(ws + head + b_info + b_order);
        int32_t* counts = reinterpret_cast<int32_t*>(ws + head + b_info + b_order + b_pend);
        int32_t* starts = counts + b_tab / 4;
        int32_t* cursor = starts + b_tab / 4;
        if (int rc = build_gallery_index(st, g_pids, ng, info, counts, starts, cursor, order, status)) return rc;
        hipLaunchKernelGGL((rank_query_kernel<RANK_PSMALL, 0>), dim3(nq), dim3(256), 0, st, distmat, q_pids, q_camids, g_camids, info, starts, order,
                           nq, ng, ap_buf, fr_buf, pending, status);
        DALI_LAUNCH_CHECK();
        hipLaunchKernelGGL((rank_query_kernel<RANK_PMAX, 1>), dim3(nq), dim3(256), 0, st, distmat, q_pids, q_camids, g_camids, info, starts, order,
                           nq, ng, ap_buf, fr_buf, pending, status);
        DALI_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(rank_reduce_kernel, dim3(1), dim3(256), 0, st, ap_buf, fr_buf, nq, max_rank, cmc, mAP, map64, num_valid);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
