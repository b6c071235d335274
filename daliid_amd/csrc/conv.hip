// conv.hip -- implicit-GEMM convolutions on bf16 MFMA for the ResNet-50-ReID trunk
// (Encoders.py:330-339: conv1 7x7/2, bottleneck 1x1 / 3x3 convs, 1x1 downsamples; layer4 at stride 1).
//
// One kernel (igemm_conv_kernel) serves forward and data-gradient, one (igemm_wgrad_kernel) the weight
// gradient.  Activations are NHWC bf16, weights [Cout][R][S][Cin] bf16 (K-contiguous), accumulation fp32.
//
//   forward : O[p][co]  = sum_{r,s,ci} W[co][r][s][ci] * X[n, ho*st-pad+r, wo*st-pad+s, ci]
//   dgrad   : dX[p][ci] = sum_{r,s,co} Wt[ci][r][s][co] * dY[n, (h+pad-r)/st, (w+pad-s)/st, co]   (exact multiples only)
//   wgrad   : dW[co][r][s][ci] = sum_p dY[p][co] * X[n, ho*st-pad+r, wo*st-pad+s, ci]
//
// Fusions: (a) the gathered operand can be transformed on load by a per-channel affine + ReLU
// (the previous BatchNorm+ReLU is never materialised); (b) the forward epilogue accumulates the
// per-channel sum / sum-of-squares partials training-mode BatchNorm needs (deterministic, no atomics);
// (c) the dgrad epilogue can add a residual gradient.
#include "gemm_tile.h"
#include "kernels.h"
#include <cstdlib>
#include <type_traits>
#include <vector>
#include <string>
#include <cstdio>

namespace dali {

__device__ __forceinline__ void decode_pixel(const GatherGeom& g, int p, int& n, int& ho, int& wo) {
    if (g.lhw >= 0 && g.lw >= 0) {
        n = p >> g.lhw;
        const int rem = p & ((1 << g.lhw) - 1);
        ho = rem >> g.lw;
        wo = rem & ((1 << g.lw) - 1);
    } else {
        const int hw = g.Hout * g.Wout;
        n = p / hw;
        const int rem = p - n * hw;
        ho = rem / g.Wout;
        wo = rem - ho * g.Wout;
    }
}

// element index of output pixel p (GEMM N index) in the [pixels][Cm] output tensor
__device__ __forceinline__ size_t out_pixel(const GatherGeom& g, int p) {
    if (!g.sub) return (size_t)p;
    int n, ho, wo;
    decode_pixel(g, p, n, ho, wo);
    return ((size_t)n * g.Hfull + (2 * ho + g.oph)) * g.Wfull + (2 * wo + g.opw);
}

// offset (elements) of tap (r,s) for base coords, or -1 if the tap falls outside / is not hit
__device__ __forceinline__ long long tap_offset(const GatherGeom& g, long long img_base, int h0, int w0, int r, int s) {
    int hi, wi;
    if (g.mode == 0) {
        hi = h0 + r;
        wi = w0 + s;
    } else {
        const int th = h0 - r, tw = w0 - s;
        if (th < 0 || tw < 0) return -1;
        if (g.stride == 2) {
            if ((th | tw) & 1) return -1;
            hi = th >> 1; wi = tw >> 1;
        } else {
            hi = th; wi = tw;
        }
    }
    if (hi < 0 || hi >= g.Hin || wi < 0 || wi >= g.Win) return -1;
    return img_base + (long long)hi * g.row_pitch + (long long)wi * g.pix_pitch;
}

__device__ __forceinline__ uint4 bn_relu_chunk(uint4 v, const float* __restrict__ sc, const float* __restrict__ sh, int relu) {
    const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(sh), b1 = *reinterpret_cast<const float4*>(sh + 4);
    const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float lo = bf16_bits_to_f32(w[t] & 0xffffu) * s[2 * t] + b[2 * t];
        float hi = bf16_bits_to_f32(w[t] >> 16) * s[2 * t + 1] + b[2 * t + 1];
        if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
        o[t] = pack_bf16x2(lo, hi);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// Stride-2 data gradient with all four output-parity classes in one launch (GatherGeom::sub == 2): the grid is four copies of the
// class's tile grid (class_grid blocks each, a multiple of 8, so a block keeps its XCD), the class with the most taps first so that the
// short classes fill the launch's tail.  Sets this workgroup's class fields, returns its block index inside the class.
// Class c: output rows of parity ph = 1 - (c >> 1), columns pw = 1 - (c & 1): for a 3 x 3 / pad 1 kernel 4, 2, 2 and 1 taps.
__device__ __forceinline__ int parity_block(IGemmArgs& a) {
    int b = blockIdx.x;
    if (a.g.sub == 2) {
        const int cls = b / a.g.class_grid;
        b -= cls * a.g.class_grid;
        const int ph = 1 - (cls >> 1), pw = 1 - (cls & 1);
        a.g.sub = 1; a.g.oph = ph; a.g.opw = pw;
        a.g.r0 = (ph + a.g.pad) & 1; a.g.rstep = 2; a.g.nr = (a.g.R - a.g.r0 + 1) / 2;
        a.g.s0 = (pw + a.g.pad) & 1; a.g.sstep = 2; a.g.ns = (a.g.S - a.g.s0 + 1) / 2;
    }
    return b;
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad
// ------------------------------------------------------------------------------------------------
template <int TN, bool IN_BN>
struct PixelLoader {
    static constexpr int BCH = TN * 4 / 256;
    const uint16_t* X;
    const float* sc; const float* sh;
    GatherGeom g;
    int relu;
    long long img_base[BCH];
    int h0[BCH], w0[BCH];
    bool ok[BCH];
    int kc;                   // this thread's 16-byte chunk inside the 32-channel k-tile
    int tiles_per_tap;

    __device__ __forceinline__ void init(const IGemmArgs& a, int p0) {
        X = a.X; sc = a.in_scale; sh = a.in_shift; g = a.g; relu = a.in_relu;
        kc = threadIdx.x & 3;
        tiles_per_tap = g.Ck >> 5;
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int row = (threadIdx.x + i * 256) >> 2;
            const int p = p0 + row;
            ok[i] = p < a.P;
            int n = 0, ho = 0, wo = 0;
            if (ok[i]) decode_pixel(g, p, n, ho, wo);
            img_base[i] = (long long)n * g.img_pitch;
            if (g.mode == 0) { h0[i] = ho * g.stride - g.pad; w0[i] = wo * g.stride - g.pad; }
            else { h0[i] = ho + g.pad; w0[i] = wo + g.pad; }
        }
    }
    // row index is implied by (tid, i): the mainloop calls with row == (tid + i*256) >> 2
    __device__ __forceinline__ uint4 operator()(int /*arr*/, int row, int kt, int /*kc*/) const {
        const int i = row >> 6;                    // rows handled by this thread are 64 apart
        const int tap = kt / tiles_per_tap;
        const int c0 = (kt - tap * tiles_per_tap) * 32 + kc * 8;
        const int r = tap / g.S, s = tap - r * g.S;
        // BCH is small: select this thread's i-th row by unrolled compare (keeps arrays in registers)
        long long off = -1;
#pragma unroll
        for (int t = 0; t < BCH; ++t)
            if (t == i && ok[t]) off = tap_offset(g, img_base[t], h0[t], w0[t], r, s);
        if (off < 0) return make_uint4(0, 0, 0, 0);
        uint4 v = *reinterpret_cast<const uint4*>(X + off + c0);
        if constexpr (IN_BN) v = bn_relu_chunk(v, sc + c0, sh + c0, relu);
        return v;
    }
};

struct WeightLoader {
    const uint16_t* W; int m0, Cm, K;
    __device__ __forceinline__ uint4 operator()(int /*arr*/, int row, int kt, int kc) const {
        const int m = m0 + row;
        if (m >= Cm) return make_uint4(0, 0, 0, 0);
        return *reinterpret_cast<const uint4*>(W + (size_t)m * K + kt * 32 + kc * 8);
    }
};

// workgroup barrier that orders LDS traffic only (no wait for outstanding global loads / stores)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
// same, after also waiting for this wave's global loads (data loaded into registers and then written to LDS)
__device__ __forceinline__ void lds_barrier_vm() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// sum over the 16 lanes of a DPP row, result in every lane of the row (rotate-and-add: 4 VALU ops, no LDS crossbar)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

// Eight values per lane summed over the 16 lanes of a DPP row by a halving butterfly: row_mirror pairs lane l with 15 - l (the low half keeps
// values 0-3, the high half 4-7), row_half_mirror pairs l with 7 - l inside each half, then quad permutes xor 2 and xor 1: 4 + 2 + 1 + 1 = 8 DPP
// adds and 14 selects where eight full row sums take 32 DPP adds.  On return lanes l and l ^ 1 hold the total of value index
// (l >= 8 ? 4 : 0) + ((l & 7) >= 4 ? 2 : 0) + ((l & 2) ? 1 : 0).  (The statistics epilogue: an ablation without its 128 DPP adds per wave ran the
// short-K forward launches 7-12 % faster; the adds are dependent chains with DPP's extra wait states.)
#define DALI_DPP(v, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xf, 0xf, false))
__device__ __forceinline__ float row16_reduce8(const float (&v)[8], int lane) {
    const bool hi8 = (lane & 8) != 0, hi4 = (lane & 4) != 0, hi2 = (lane & 2) != 0;
    float a[4], b[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float keep = hi8 ? v[4 + k] : v[k], send = hi8 ? v[k] : v[4 + k];
        a[k] = keep + DALI_DPP(send, 0x140);                // row_mirror
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = hi4 ? a[2 + k] : a[k], send = hi4 ? a[k] : a[2 + k];
        b[k] = keep + DALI_DPP(send, 0x141);                // row_half_mirror
    }
    const float keep = hi2 ? b[1] : b[0], send = hi2 ? b[0] : b[1];
    const float c = keep + DALI_DPP(send, 0x4e);            // quad_perm [2, 3, 0, 1]
    return c + DALI_DPP(c, 0xb1);                           // quad_perm [1, 0, 3, 2]
}

// Shared epilogue: the per-tile BatchNorm partial statistics and the bf16 store of O (+ residual).
// In-kernel stamps on the short-K layers showed the epilogue, not the memory system, bounding the kernel: 8 us of a 10 us
// workgroup lifetime in VALU / LDS-crossbar work (statistics by 128 ds_bpermute shuffles, per-element predicates and
// 64-bit address arithmetic, 8-byte stores scattered over 16 cache lines per instruction) while the stores themselves
// drained in 0.2 us.  Hence: statistics first, reduced with DPP row rotates; barriers that order LDS traffic only (a
// __syncthreads() would wait for the global stores); and for interior tiles of the plain convolution a lean path that
// stages the tile through LDS as [pixel][TM channels] bf16 and writes 16 bytes per lane, TM/8 adjacent lanes covering
// one pixel's contiguous TM*2 bytes, without per-element predicates.  Values and rounding are identical on every path.
template <int FM_, int FN_> struct EpiShape { static constexpr int FM = FM_, FN = FN_; };
// Generic over the block shape: TM x TN tile, WNW waves along n, NT threads; this wave sits at (wm, wn) and owns an
// (FM*16) x (FN*16) sub-tile.
// keep the bf16 halves of `w` whose mask bits (bit 0: low half, bit 1: high half) are set
__device__ __forceinline__ uint32_t gate_bf16x2(uint32_t w, unsigned bits) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 0, 1) & 0xffffu;      // 0 or 0x0000ffff
    const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)bits, 1, 1) << 16;         // 0 or 0xffff0000
    return w & (lo | hi);
}
// exact-erf GELU (vit_pytorch.py:120-136, nn.GELU) and its derivative.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7: three orders
// below the bf16 rounding of the stored result) on v_rcp / v_exp: ocml's erff + expf were ~60 VALU instructions per element, a third of the
// fc1 / fc2 epilogues' time.  cdf = Phi(v) = 0.5 (1 + erf(v / sqrt 2)), pdf = phi(v); both share exp(-v^2 / 2).
__device__ __forceinline__ void gelu_parts(float v, float& cdf, float& pdf) {
    // explicit fused multiply-adds: the library is built with -ffp-contract=off, which left this polynomial as separate multiplies and
    // adds (~24 VALU issue slots per element; the fc1 / fc2-gradient epilogues spent 7-9 us per half tile in it, 2 us without GELU)
    const float z = fabsf(v) * 0.70710678118654752f;
    const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.44269504088896341f);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float erf_abs = __builtin_fmaf(-(poly * t), e, 1.0f);
    cdf = __builtin_fmaf(0.5f, copysignf(erf_abs, v), 0.5f);
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float v) { float c, p; gelu_parts(v, c, p); return v * c; }
__device__ __forceinline__ float gelu_grad_f(float x) { float c, p; gelu_parts(x, c, p); return __builtin_fmaf(x, p, c); }
__device__ __forceinline__ void unpack8(const uint4& q, float (&f)[8]) {
    f[0] = bf16_bits_to_f32(q.x & 0xffffu); f[1] = bf16_bits_to_f32(q.x >> 16); f[2] = bf16_bits_to_f32(q.y & 0xffffu); f[3] = bf16_bits_to_f32(q.y >> 16);
    f[4] = bf16_bits_to_f32(q.z & 0xffffu); f[5] = bf16_bits_to_f32(q.z >> 16); f[6] = bf16_bits_to_f32(q.w & 0xffffu); f[7] = bf16_bits_to_f32(q.w >> 16);
}
// bit 0 / bit 1: the low / high bf16 half of w is > 0
__device__ __forceinline__ unsigned pos_bits_bf16x2(uint32_t w) {
    return ((int)(int16_t)(w & 0xffffu) > 0 ? 1u : 0u) | (((int)w >> 16) > 0 ? 2u : 0u);
}
// Staged store, one 16-pixel column block of every wave at a time (any FN; used by the 256 x 320 tile, FN = 5).  The staging is a pure
// transpose: the WNW * 16 pixels of step j go through LDS as [pixel][TM channels] fp32 accumulators, and the thread that reads 8 adjacent
// channels of a pixel back applies the whole output stage to them -- bias, pre-activation copy (O2), GELU, GELU' factor, DropPath row
// factor, residual (+ mask) -- with its residual / GELU' argument chunk (16 bytes, requested before the staging so that the transpose
// hides its latency) and stores 16 bytes.  One rounding to bf16 at the end, as on every other path; interior tiles, natural output
// addressing, Cm % 8 == 0.
template <int TM, int TN, int FM_, int FN_, int WNW, int NT>
__device__ __forceinline__ void conv_epilogue_cols(const IGemmArgs& a, f32x4_t (&acc)[FM_][FN_], int tm, int tn, uint16_t* smem, int wm, int wn) {
    constexpr int ROWB = TM * 4 + 16, ROWS = WNW * 16, CPR = TM / 8, ITERS = ROWS * CPR / NT;       // +16: 16 rows of 16-byte writes cover all banks once
    static_assert(ROWS * CPR % NT == 0 && NT % CPR == 0, "staged store: threads must tile the column block evenly");
    const int lane = threadIdx.x & 63;
    const int mb = wm * (FM_ * 16) + (lane >> 4) * 4;
    char* stage = reinterpret_cast<char*>(smem);
    char* my_stage = stage + (wn * 16 + (lane & 15)) * ROWB + mb * 4;                   // + i*64
    const int ch = threadIdx.x % CPR, lp0 = threadIdx.x / CPR;
    const uint16_t* in_tile = a.Res ? a.Res : a.dact_pre;         // (the caller excludes Res together with dact_pre, as the half-tile paths do)
    const float* bias_c = a.bias ? a.bias + tm * TM + ch * 8 : nullptr;
#pragma unroll
    for (int j = 0; j < FN_; ++j) {
        uint32_t gofs[ITERS];                            // byte offset of this thread's chunk (the caller checks the tensor is below 4 GiB)
        uint4 tin[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int lp = lp0 + it * (NT / CPR);
            const int pix = tn * TN + (lp >> 4) * (FN_ * 16) + j * 16 + (lp & 15);
            gofs[it] = ((uint32_t)pix * (uint32_t)a.Cm + (uint32_t)(tm * TM + ch * 8)) * 2u;
            if (in_tile) tin[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(in_tile) + gofs[it]);
        }
#pragma unroll
        for (int i = 0; i < FM_; ++i) *reinterpret_cast<f32x4_t*>(my_stage + i * 64) = acc[i][j];
        lds_barrier();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int lp = lp0 + it * (NT / CPR);
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(stage + lp * ROWB + ch * 32);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(stage + lp * ROWB + ch * 32 + 16);
            if (bias_c) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias_c), b1 = *reinterpret_cast<const float4*>(bias_c + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
            }
            if (a.O2)
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.O2) + gofs[it]) =
                    make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            if (a.act == 1) {
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = gelu_f(v[t]);
            }
            if (a.dact_pre) {
                float x8[8];
                unpack8(tin[it], x8);
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] *= gelu_grad_f(x8[t]);
            }
            if (a.row_scale) {
                const float rs = a.row_scale[gofs[it] / ((uint32_t)a.Cm * 2u)];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] *= rs;
            }
            if (a.Res) {
                uint4 rv = tin[it];
                if (a.res_mask) {
                    const unsigned m = a.res_mask[gofs[it] >> 4];
                    rv.x = gate_bf16x2(rv.x, m); rv.y = gate_bf16x2(rv.y, m >> 2); rv.z = gate_bf16x2(rv.z, m >> 4); rv.w = gate_bf16x2(rv.w, m >> 6);
                }
                float r8[8];
                unpack8(rv, r8);
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] += r8[t];
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.O) + gofs[it]) =
                make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
        if (j + 1 < FN_) lds_barrier();                  // the read-out is done before the next column block overwrites the stage
    }
}

// LIN: 3 = the fused output stage (IGemmArgs::out_scale ... out_mask) in a staged block and in the general path;
// LIN: 0 = a convolution-only instantiation (no linear-layer extras compiled in at all), 1 = extras in the staged block and in the
// general path (the LIN kernel variants), 2 = extras in the general path only (kernels without a LIN variant)
template <int TM, int TN, int FM_, int FN_, int WNW, int NT, int LIN = 2>
__device__ __forceinline__ void conv_epilogue_g(const IGemmArgs& a, f32x4_t (&acc)[FM_][FN_], int tm, int tn, uint16_t* smem, int wm, int wn) {
    using Cfg = EpiShape<FM_, FN_>;
    constexpr bool EVEN = FN_ % 2 == 0;                  // the staged stores below split the tile's pixels in two halves; odd FN: conv_epilogue_cols
    const int lane = threadIdx.x & 63;
    const int mb = wm * (FM_ * 16) + (lane >> 4) * 4, nb = wn * (FN_ * 16) + (lane & 15);
    // Precondition: the caller has synchronised after its main loop (every kernel ends the loop with a barrier that follows
    // each wave's last fragment read), so smem is free.
    // ---- BatchNorm partial statistics: per channel sum / sumsq over this tile's pixels ----
    constexpr int STAGE_BYTES = (TN / 2) * (TM * 2 + 32);      // the staged store's LDS image (below); the partial sums sit behind it
    if (a.stats) {
        float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + STAGE_BYTES);   // [WNW (wn)][TM][2]
#pragma unroll
        for (int i = 0; i < Cfg::FM; ++i) {
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < Cfg::FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; s1[r] += v; s2[r] = __builtin_fmaf(v, v, s2[r]); }      // (explicit: the build has -ffp-contract=off)
            const float v8[8] = {s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3]};
            const float tot = row16_reduce8(v8, lane);   // lanes 0-7 of the row: sums, 8-15: sums of squares; channel r = 2 * ((l & 7) >= 4) + ((l & 2) != 0)
            if ((lane & 1) == 0) {
                const int r = ((lane & 4) >> 1) | ((lane & 2) >> 1);
                red[(wn * TM + mb + i * 16 + r) * 2 + ((lane >> 3) & 1)] = tot;
            }
        }
        lds_barrier();
        for (int t = threadIdx.x; t < TM; t += NT) {
            const int c = tm * TM + t;
            if (c < a.Cm) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WNW; ++w) { const float2 v = *reinterpret_cast<const float2*>(red + (w * TM + t) * 2); s1 += v.x; s2 += v.y; }
                *reinterpret_cast<float2*>(a.stats + ((size_t)tn * a.Cm + c) * 2) = make_float2(s1, s2);
            }
        }
        // no barrier: the staged store below does not touch `red`
    }
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 12 + 5] = __builtin_amdgcn_s_memrealtime();     // stats done

    // ---- lean path: plain convolution (optionally + residual), interior tile, natural output addressing ----
    // LIN == 4: the lean path with the output addressing of a stride-2 data gradient's parity class (GatherGeom::sub): the pixel rows of
    // the tile scatter to every second position of every second image row, each still TM * 2 contiguous bytes.  Its own instantiations
    // (the sub-problems went through the general path before: 8-byte stores per lane, per-element residual loads; layer2's downsample
    // data gradient 169 us against 71 us forward), so that the hot convolution instantiations stay as they are.
    constexpr bool SUB = LIN == 4;
    const bool plain = LIN != 3 && LIN != 5 && !a.O2 && a.act == 0 && !a.dact_pre && !a.row_scale && (SUB || !a.g.sub) && (a.Cm & 7) == 0;   // bias (linear layers) is folded in below
    // the linear layers' GELU / pre-activation copy (O2) / GELU' factor take a second staged block further down, kept apart so that the
    // convolutions' path stays as lean as it was (folding them into one block cost the ResNet step 0.8 ms)
    // (only in the LIN instantiations of the kernels: compiled into every kernel it changed the convolutions' register allocation and
    // cost the ResNet step 0.4 ms even when never taken)
    const bool plain_ext = LIN == 1 && !plain && !a.g.sub && (a.Cm & 7) == 0 && !(a.Res && a.dact_pre);
    const bool interior = (tm + 1) * TM <= a.Cm && (tn + 1) * TN <= a.P;
    if constexpr (!EVEN) {
        if (interior && LIN != 3 && LIN != 5 && !a.g.sub && (a.Cm & 7) == 0 && !(a.Res && a.dact_pre) && (unsigned long long)a.P * a.Cm * 2ull < 0xffffffffull) { conv_epilogue_cols<TM, TN, FM_, FN_, WNW, NT>(a, acc, tm, tn, smem, wm, wn); return; }
    }
    if constexpr (EVEN) if (plain && interior) {
        constexpr int ROWB = TM * 2 + 32;                       // LDS row pitch in bytes (+32: spreads the 8-byte accesses over banks)
        constexpr int HFN = FN_ / 2, WROWS = HFN * 16, ROWS = TN / 2, CPR = TM / 8, ITERS = ROWS * CPR / NT;
        static_assert(ROWS * CPR % NT == 0 && NT % CPR == 0, "staged store: threads must tile the half evenly");
        char* stage = reinterpret_cast<char*>(smem);
        char* my_stage = stage + (wn * WROWS + (lane & 15)) * ROWB + mb * 2;              // + jj*16*ROWB + i*32
        const int ch = threadIdx.x % CPR, lp0 = threadIdx.x / CPR;                        // read-out: 16-byte chunk / first row
        static_assert(ROWS * ROWB == STAGE_BYTES, "layout of the partial sums behind the staged tile");
        float4 bias4[Cfg::FM];                                                            // this lane's 4 channels of every 16-row block
#pragma unroll
        for (int i = 0; i < Cfg::FM; ++i)
            bias4[i] = a.bias ? *reinterpret_cast<const float4*>(a.bias + tm * TM + mb + i * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t gbase = ((size_t)(tn * TN + h * WROWS) * a.Cm + tm * TM + ch * 8) * 2;     // bytes; + pixel q * Cm * 2
            auto row_bytes = [&](int q) -> size_t {             // byte offset of this thread's chunk of tile pixel q (of half h)
                if constexpr (SUB) return (out_pixel(a.g, tn * TN + h * WROWS + q) * a.Cm + tm * TM + ch * 8) * 2;
                else return gbase + (size_t)q * a.Cm * 2;
            };
            if (a.Res) {                                        // residual tile -> LDS with 16-byte loads, same layout as the output
                unsigned rm[ITERS];                             // (mask bytes requested together: see the fused output stage below)
                if (a.res_mask) {
#pragma unroll
                    for (int it = 0; it < ITERS; ++it) {
                        const int lp = lp0 + it * (NT / CPR);
                        rm[it] = a.res_mask[row_bytes((lp / WROWS) * (FN_ * 16) + (lp % WROWS)) >> 4];
                    }
                }
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lp = lp0 + it * (NT / CPR);
                    const int q = (lp / WROWS) * (FN_ * 16) + (lp % WROWS);
                    const size_t rb = row_bytes(q);
                    uint4 rv = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.Res) + rb);
                    if (a.res_mask) {                           // 16 bytes = 8 channels = one mask byte
                        const unsigned m = rm[it];
                        rv.x = gate_bf16x2(rv.x, m); rv.y = gate_bf16x2(rv.y, m >> 2); rv.z = gate_bf16x2(rv.z, m >> 4); rv.w = gate_bf16x2(rv.w, m >> 6);
                    }
                    *reinterpret_cast<uint4*>(stage + lp * ROWB + ch * 16) = rv;
                }
                lds_barrier_vm();                               // the loaded residual is visible to every wave
            }
#pragma unroll
            for (int jj = 0; jj < HFN; ++jj) {
                const int j = h * HFN + jj;
#pragma unroll
                for (int i = 0; i < Cfg::FM; ++i) {
                    float v0 = acc[i][j][0] + bias4[i].x, v1 = acc[i][j][1] + bias4[i].y, v2 = acc[i][j][2] + bias4[i].z, v3 = acc[i][j][3] + bias4[i].w;
                    uint2* slot = reinterpret_cast<uint2*>(my_stage + jj * 16 * ROWB + i * 32);
                    if (a.Res) {
                        const uint2 rv = *slot;
                        v0 += bf16_bits_to_f32(rv.x & 0xffffu); v1 += bf16_bits_to_f32(rv.x >> 16);
                        v2 += bf16_bits_to_f32(rv.y & 0xffffu); v3 += bf16_bits_to_f32(rv.y >> 16);
                    }
                    *slot = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                }
            }
            lds_barrier();
            if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 12 + 6 + 2 * h] = __builtin_amdgcn_s_memrealtime();   // half staged
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int lp = lp0 + it * (NT / CPR);
                const int q = (lp / WROWS) * (FN_ * 16) + (lp % WROWS);                  // pixel inside the tile, minus h*WROWS
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.O) + row_bytes(q)) =
                    *reinterpret_cast<const uint4*>(stage + lp * ROWB + ch * 16);
            }
            if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 12 + 7 + 2 * h] = __builtin_amdgcn_s_memrealtime();   // half's stores issued
            if (h == 0) lds_barrier();                          // the LDS reads are done before the second half overwrites them
        }
        return;
    }

    if constexpr (LIN == 1 && EVEN) if (plain_ext && interior) {            // linear-layer extras: GELU, pre-activation copy, GELU' factor
        constexpr int ROWB = TM * 2 + 32;                       // LDS row pitch in bytes (+32: spreads the 8-byte accesses over banks)
        constexpr int HFN = FN_ / 2, WROWS = HFN * 16, ROWS = TN / 2, CPR = TM / 8, ITERS = ROWS * CPR / NT;
        static_assert(ROWS * CPR % NT == 0 && NT % CPR == 0, "staged store: threads must tile the half evenly");
        char* stage = reinterpret_cast<char*>(smem);
        char* my_stage = stage + (wn * WROWS + (lane & 15)) * ROWB + mb * 2;              // + jj*16*ROWB + i*32
        const int ch = threadIdx.x % CPR, lp0 = threadIdx.x / CPR;                        // read-out: 16-byte chunk / first row
        static_assert(ROWS * ROWB == STAGE_BYTES, "layout of the partial sums behind the staged tile");
        float4 bias4[Cfg::FM];                                                            // this lane's 4 channels of every 16-row block
#pragma unroll
        for (int i = 0; i < Cfg::FM; ++i)
            bias4[i] = a.bias ? *reinterpret_cast<const float4*>(a.bias + tm * TM + mb + i * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        const uint16_t* in_tile = a.Res ? a.Res : a.dact_pre;       // optional input tile staged through LDS in the output layout
        auto store_half = [&](uint16_t* dst, size_t gbase) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int lp = lp0 + it * (NT / CPR);
                const int q = (lp / WROWS) * (FN_ * 16) + (lp % WROWS);                  // pixel inside the tile, minus h*WROWS
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + gbase + (size_t)q * a.Cm * 2) =
                    *reinterpret_cast<const uint4*>(stage + lp * ROWB + ch * 16);
            }
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t gbase = ((size_t)(tn * TN + h * WROWS) * a.Cm + tm * TM + ch * 8) * 2;     // bytes; + pixel q * Cm * 2
            if (a.O2) {                                         // pre-activation copy (kept for the backward) goes out first
#pragma unroll
                for (int jj = 0; jj < HFN; ++jj) {
                    const int j = h * HFN + jj;
#pragma unroll
                    for (int i = 0; i < Cfg::FM; ++i)
                        *reinterpret_cast<uint2*>(my_stage + jj * 16 * ROWB + i * 32) =
                            make_uint2(pack_bf16x2(acc[i][j][0] + bias4[i].x, acc[i][j][1] + bias4[i].y), pack_bf16x2(acc[i][j][2] + bias4[i].z, acc[i][j][3] + bias4[i].w));
                }
                lds_barrier();
                store_half(a.O2, gbase);
                lds_barrier();
            }
            if (in_tile) {                                      // residual / GELU' argument tile -> LDS with 16-byte loads
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lp = lp0 + it * (NT / CPR);
                    const int q = (lp / WROWS) * (FN_ * 16) + (lp % WROWS);
                    const size_t rb = gbase + (size_t)q * a.Cm * 2;
                    uint4 rv = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(in_tile) + rb);
                    if (a.Res && a.res_mask) {                  // 16 bytes = 8 channels = one mask byte
                        const unsigned m = a.res_mask[rb >> 4];
                        rv.x = gate_bf16x2(rv.x, m); rv.y = gate_bf16x2(rv.y, m >> 2); rv.z = gate_bf16x2(rv.z, m >> 4); rv.w = gate_bf16x2(rv.w, m >> 6);
                    }
                    *reinterpret_cast<uint4*>(stage + lp * ROWB + ch * 16) = rv;
                }
                lds_barrier_vm();                               // the loaded tile is visible to every wave
            }
#pragma unroll
            for (int jj = 0; jj < HFN; ++jj) {
                const int j = h * HFN + jj;
#pragma unroll
                for (int i = 0; i < Cfg::FM; ++i) {
                    float v[4] = {acc[i][j][0] + bias4[i].x, acc[i][j][1] + bias4[i].y, acc[i][j][2] + bias4[i].z, acc[i][j][3] + bias4[i].w};
                    uint2* slot = reinterpret_cast<uint2*>(my_stage + jj * 16 * ROWB + i * 32);
                    if (a.act == 1) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = gelu_f(v[t]);
                    }
                    if (a.row_scale) {                          // DropPath: the whole branch output (bias included) times its sample's factor
                        const float rs = a.row_scale[tn * TN + h * WROWS + wn * (FN_ * 16) + jj * 16 + (lane & 15)];
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] *= rs;
                    }
                    if (in_tile) {
                        const uint2 rv = *slot;
                        const float x4[4] = {bf16_bits_to_f32(rv.x & 0xffffu), bf16_bits_to_f32(rv.x >> 16), bf16_bits_to_f32(rv.y & 0xffffu), bf16_bits_to_f32(rv.y >> 16)};
                        if (a.Res) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) v[t] += x4[t];
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                v[t] *= gelu_grad_f(x4[t]);
                        }
                    }
                    *slot = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
            }
            lds_barrier();
            if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 12 + 6 + 2 * h] = __builtin_amdgcn_s_memrealtime();   // half staged
            store_half(a.O, gbase);
            if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 12 + 7 + 2 * h] = __builtin_amdgcn_s_memrealtime();   // half's stores issued
            if (h == 0) lds_barrier();                          // the LDS reads are done before the second half overwrites them
        }
        return;
    }

    // LIN == 5: the fused output stage WITHOUT residual, masks and mask bits (scale / shift / bias / ReLU only): the inference forward's
    // conv + BatchNorm + ReLU launches on the 3x3 and narrow kernels, whose register budgets the full stage's residual prefetch overflowed
    if constexpr ((LIN == 3 || LIN == 5) && EVEN) if (interior && !a.g.sub && (a.Cm & 7) == 0) {     // fused output stage (see IGemmArgs): staged like the lean path
        constexpr bool FULL = LIN == 3;
        const uint16_t* const Res = FULL ? a.Res : nullptr;
        const uint8_t* const out_mask = FULL ? a.out_mask : nullptr;
        const uint8_t* const res_mask = FULL ? a.res_mask : nullptr;
        uint8_t* const bits_out = FULL ? a.bits_out : nullptr;
        constexpr int ROWB = TM * 2 + 32;
        constexpr int HFN = FN_ / 2, WROWS = HFN * 16, ROWS = TN / 2, CPR = TM / 8, ITERS = ROWS * CPR / NT;
        static_assert(ROWS * CPR % NT == 0 && NT % CPR == 0, "staged store: threads must tile the half evenly");
        char* stage = reinterpret_cast<char*>(smem);
        char* my_stage = stage + (wn * WROWS + (lane & 15)) * ROWB + mb * 2;
        const int ch = threadIdx.x % CPR, lp0 = threadIdx.x / CPR;
        float4 sc4[Cfg::FM], sh4[Cfg::FM];
#pragma unroll
        for (int i = 0; i < Cfg::FM; ++i) { sc4[i] = make_float4(1.f, 1.f, 1.f, 1.f); sh4[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (a.out_scale) {                                      // one branch per batch of loads (as for the mask bytes below)
#pragma unroll
            for (int i = 0; i < Cfg::FM; ++i) sc4[i] = *reinterpret_cast<const float4*>(a.out_scale + tm * TM + mb + i * 16);
        }
        if (a.out_shift) {
#pragma unroll
            for (int i = 0; i < Cfg::FM; ++i) sh4[i] = *reinterpret_cast<const float4*>(a.out_shift + tm * TM + mb + i * 16);
        }
        if (a.bias) {
            float4 bi4[Cfg::FM];
#pragma unroll
            for (int i = 0; i < Cfg::FM; ++i) bi4[i] = *reinterpret_cast<const float4*>(a.bias + tm * TM + mb + i * 16);
#pragma unroll
            for (int i = 0; i < Cfg::FM; ++i) { sh4[i].x += bi4[i].x; sh4[i].y += bi4[i].y; sh4[i].z += bi4[i].z; sh4[i].w += bi4[i].w; }
        }
        // the residual tile of BOTH halves is requested up front: the second half's loads fly while the first half is formed and stored
        // (requested per half they exposed one HBM round trip per half: these kernels are short-K, their epilogue is most of their time)
        constexpr bool PRE = FULL && NT == 256;          // (the 512-thread 128 x 256 kernel would leave its 128-register budget = two workgroups per CU)
        uint4 rpre[2][PRE ? ITERS : 1];
        if (PRE && Res) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const size_t gbase = ((size_t)(tn * TN + h * WROWS) * a.Cm + tm * TM + ch * 8) * 2;
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lp = lp0 + it * (NT / CPR);
                    const int q = (lp / WROWS) * (FN_ * 16) + (lp % WROWS);
                    rpre[h][it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Res) + gbase + (size_t)q * a.Cm * 2);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const size_t gbase = ((size_t)(tn * TN + h * WROWS) * a.Cm + tm * TM + ch * 8) * 2;     // bytes; + pixel q * Cm * 2
            // the mask bytes of this half, requested together and early: "if (mask) m = mask[..]" inside the unrolled loops below compiled to a
            // branch and a wait around every byte load (up to 16 dependent round trips per tile; MI355X guide, the per-element select trap)
            unsigned om[ITERS], rm[ITERS];
            if (out_mask) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lp = lp0 + it * (NT / CPR);
                    om[it] = out_mask[(gbase + (size_t)((lp / WROWS) * (FN_ * 16) + (lp % WROWS)) * a.Cm * 2) >> 4];
                }
            }
            if (Res && res_mask) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lp = lp0 + it * (NT / CPR);
                    rm[it] = res_mask[(gbase + (size_t)((lp / WROWS) * (FN_ * 16) + (lp % WROWS)) * a.Cm * 2) >> 4];
                }
            }
            if (Res) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int lp = lp0 + it * (NT / CPR);
                    const int q = (lp / WROWS) * (FN_ * 16) + (lp % WROWS);
                    const size_t rb = gbase + (size_t)q * a.Cm * 2;
                    uint4 rv;
                    if constexpr (PRE) rv = rpre[h][it];
                    else rv = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Res) + rb);
                    if (res_mask) {
                        const unsigned m = rm[it];
                        rv.x = gate_bf16x2(rv.x, m); rv.y = gate_bf16x2(rv.y, m >> 2); rv.z = gate_bf16x2(rv.z, m >> 4); rv.w = gate_bf16x2(rv.w, m >> 6);
                    }
                    *reinterpret_cast<uint4*>(stage + lp * ROWB + ch * 16) = rv;
                }
                if constexpr (PRE) lds_barrier();               // (the register dependence waits for the loads of this half only)
                else lds_barrier_vm();
            }
#pragma unroll
            for (int jj = 0; jj < HFN; ++jj) {
                const int j = h * HFN + jj;
#pragma unroll
                for (int i = 0; i < Cfg::FM; ++i) {
                    float v0 = acc[i][j][0] * sc4[i].x + sh4[i].x, v1 = acc[i][j][1] * sc4[i].y + sh4[i].y;
                    float v2 = acc[i][j][2] * sc4[i].z + sh4[i].z, v3 = acc[i][j][3] * sc4[i].w + sh4[i].w;
                    uint2* slot = reinterpret_cast<uint2*>(my_stage + jj * 16 * ROWB + i * 32);
                    if (Res) {
                        const uint2 rv = *slot;
                        if (a.res_scale) {                      // (loaded per use: L1-resident; held across the tile it cost 16 registers and spills)
                            const float4 rs = *reinterpret_cast<const float4*>(a.res_scale + tm * TM + mb + i * 16);
                            v0 += rs.x * bf16_bits_to_f32(rv.x & 0xffffu); v1 += rs.y * bf16_bits_to_f32(rv.x >> 16);
                            v2 += rs.z * bf16_bits_to_f32(rv.y & 0xffffu); v3 += rs.w * bf16_bits_to_f32(rv.y >> 16);
                        } else {
                            v0 += bf16_bits_to_f32(rv.x & 0xffffu); v1 += bf16_bits_to_f32(rv.x >> 16);
                            v2 += bf16_bits_to_f32(rv.y & 0xffffu); v3 += bf16_bits_to_f32(rv.y >> 16);
                        }
                    }
                    if (a.out_relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    *slot = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                }
            }
            lds_barrier();
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int lp = lp0 + it * (NT / CPR);
                const int q = (lp / WROWS) * (FN_ * 16) + (lp % WROWS);
                const size_t ob = gbase + (size_t)q * a.Cm * 2;
                uint4 v = *reinterpret_cast<const uint4*>(stage + lp * ROWB + ch * 16);
                if (out_mask) {
                    const unsigned m = om[it];
                    v.x = gate_bf16x2(v.x, m); v.y = gate_bf16x2(v.y, m >> 2); v.z = gate_bf16x2(v.z, m >> 4); v.w = gate_bf16x2(v.w, m >> 6);
                }
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.O) + ob) = v;
                if (bits_out) bits_out[ob >> 4] = (uint8_t)(pos_bits_bf16x2(v.x) | (pos_bits_bf16x2(v.y) << 2) | (pos_bits_bf16x2(v.z) << 4) | (pos_bits_bf16x2(v.w) << 6));
            }
            if (h == 0) lds_barrier();
        }
        return;
    }

    // ---- general path: edge tiles, linear-layer epilogues (bias / GELU / GELU' / second output), parity sub-problems ----
#pragma unroll
    for (int j = 0; j < Cfg::FN; ++j) {
        const int p = tn * TN + nb + j * 16;
        const size_t opix = (p < a.P) ? out_pixel(a.g, p) : 0;
#pragma unroll
        for (int i = 0; i < Cfg::FM; ++i) {
            const int c = tm * TM + mb + i * 16;
            unsigned nib = 0;
            if (p < a.P && c < a.Cm) {      // Cm is a multiple of 4: the 4 channels are all valid
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                const size_t o = opix * a.Cm + c;
                if constexpr (LIN == 3 || LIN == 5) {
                    if (a.out_scale) { const float4 sv = *reinterpret_cast<const float4*>(a.out_scale + c); v[0] *= sv.x; v[1] *= sv.y; v[2] *= sv.z; v[3] *= sv.w; }
                    if (a.out_shift) { const float4 sv = *reinterpret_cast<const float4*>(a.out_shift + c); v[0] += sv.x; v[1] += sv.y; v[2] += sv.z; v[3] += sv.w; }
                }
                if (a.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(a.bias + c);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if constexpr (LIN == 1 || LIN == 2) {
                if (a.O2) *reinterpret_cast<uint2*>(a.O2 + o) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                if (a.act == 1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = gelu_f(v[t]);
                }
                if (a.dact_pre) {
                    const uint2 pv = *reinterpret_cast<const uint2*>(a.dact_pre + o);
                    const float x4[4] = {bf16_bits_to_f32(pv.x & 0xffffu), bf16_bits_to_f32(pv.x >> 16), bf16_bits_to_f32(pv.y & 0xffffu),
                                         bf16_bits_to_f32(pv.y >> 16)};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        v[t] *= gelu_grad_f(x4[t]);
                }
                if (a.row_scale) {
                    const float rs = a.row_scale[p];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] *= rs;
                }
                }
                if (LIN != 5 && a.Res) {
                    uint2 rv = *reinterpret_cast<const uint2*>(a.Res + o);
                    if (a.res_mask) {                           // o is a multiple of 4: this lane's nibble of the mask byte
                        const unsigned m = a.res_mask[o >> 3] >> (o & 4);
                        rv.x = gate_bf16x2(rv.x, m); rv.y = gate_bf16x2(rv.y, m >> 2);
                    }
                    float r4[4] = {bf16_bits_to_f32(rv.x & 0xffffu), bf16_bits_to_f32(rv.x >> 16), bf16_bits_to_f32(rv.y & 0xffffu), bf16_bits_to_f32(rv.y >> 16)};
                    if constexpr (LIN == 3) if (a.res_scale) {
                        const float4 rs = *reinterpret_cast<const float4*>(a.res_scale + c);
                        r4[0] *= rs.x; r4[1] *= rs.y; r4[2] *= rs.z; r4[3] *= rs.w;
                    }
                    v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
                }
                uint2 ov = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                if constexpr (LIN == 5) { if (a.out_relu) ov = make_uint2(pack_bf16x2(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)), pack_bf16x2(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f))); }
                if constexpr (LIN == 3) {
                    if (a.out_relu) ov = make_uint2(pack_bf16x2(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)), pack_bf16x2(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)));
                    if (a.out_mask) {
                        const unsigned m = a.out_mask[o >> 3] >> (o & 4);
                        ov.x = gate_bf16x2(ov.x, m); ov.y = gate_bf16x2(ov.y, m >> 2);
                    }
                    nib = pos_bits_bf16x2(ov.x) | (pos_bits_bf16x2(ov.y) << 2);
                }
                *reinterpret_cast<uint2*>(a.O + o) = ov;
            }
            if constexpr (LIN == 3) {
                // this lane's 4 mask bits and those of lane ^ 16 (the other half of the same byte: Cm % 8 == 0 is required) -> one byte store
                const unsigned other = __shfl_xor(nib, 16, 64);
                if (a.bits_out && p < a.P && c < a.Cm && ((lane >> 4) & 1) == 0) a.bits_out[(opix * a.Cm + c) >> 3] = (uint8_t)(nib | (other << 4));
            }
        }
    }
}
template <class Cfg, int LIN = 2>
__device__ __forceinline__ void conv_epilogue(const IGemmArgs& a, f32x4_t (&acc)[Cfg::FM][Cfg::FN], int tm, int tn, uint16_t* smem) {
    const int wave = threadIdx.x >> 6;
    conv_epilogue_g<Cfg::TM, Cfg::TN, Cfg::FM, Cfg::FN, 2, 256, LIN>(a, acc, tm, tn, smem, wave >> 1, wave & 1);
}

template <int TM, int TN, bool IN_BN>
__global__ __launch_bounds__(256) void igemm_conv_kernel(IGemmArgs a, int tiles_m, int tiles_n) {
    using Cfg = GemmCfg<TM, TN, 1, 1, 1>;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    int tm, tn;
    if (!xcd_tile_map(blockIdx.x, tiles_m, tiles_n, tm, tn)) return;
    f32x4_t acc[Cfg::FM][Cfg::FN];
#pragma unroll
    for (int i = 0; i < Cfg::FM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int K = a.g.R * a.g.S * a.g.Ck;
    WeightLoader la{a.W, tm * TM, a.Cm, K};
    PixelLoader<TN, IN_BN> lb;
    lb.init(a, tn * TN);
    gemm_mainloop<Cfg>(acc, la, lb, K >> 5, smem);

    conv_epilogue<Cfg>(a, acc, tm, tn, smem);
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad, LDS-DMA version (the one the net plan uses).  Both operand tiles travel HBM/L2 -> LDS with
// `buffer_load_dwordx4 ... lds` (no VGPR staging, no ds_write); zero padding comes from the buffer range check:
// an out-of-image tap is given an offset past num_records and the DMA writes zeros.  The LDS image is the same
// swizzled [rows][32] bf16 image as gemm_tile.h; because an LDS-DMA wave-instruction writes lane-linear (lane L ->
// 16-byte slot L of a 1 KiB block = row L>>2, physical chunk L&3), the XOR swizzle is applied to the SOURCE chunk
// each lane fetches.  Two LDS stages, one barrier per k-tile: the DMA of tile t+1 is in flight while tile t is
// multiplied.  Tap / channel position advance as wave-uniform scalars (no per-load division).
// ------------------------------------------------------------------------------------------------
// (lds_void_ptr, DMA_OOB, dma_wait<N> live in gemm_tile.h: shared with eval.hip)

// SRC2: K = the channels of X followed by the channels of a second plain [P][channels] tensor X2 (IGemmArgs::X2 / Ck1; 1x1, stride 1): its own
// instantiations, so that the hot ones carry none of it.
template <int TM, int TN, int NSTAGE, int EPI = 0, bool SRC2 = false>        // EPI: 0 convolution, 1 linear-layer extras, 3 fused output stage (conv_epilogue_g)
__global__ __launch_bounds__(256, (TM >= 128 ? (EPI == 3 ? 3 : 4) : 2)) void igemm_conv_dma_kernel(IGemmArgs a, int tiles_m, int tiles_n) {
    // 128 x 128: <= 128 VGPRs (4 waves per SIMD); the 64 x 256 shape carries twice the per-lane gather state and would spill
    using Cfg = GemmCfg<TM, TN, 1, 1, 1>;
    constexpr int FM = Cfg::FM, FN = Cfg::FN;
    constexpr int A_BLK = TM / 16 / 4, B_BLK = TN / 16 / 4;      // 1 KiB DMA blocks (16 rows x 64 B) per wave per k-tile
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    int tm, tn;
    if (!xcd_tile_map(parity_block(a), tiles_m, tiles_n, tm, tn)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12] = __builtin_amdgcn_s_memrealtime();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const GatherGeom g = a.g;
    const int K = g.R * g.S * g.Ck;                    // weight row length
    const int ktiles = (g.nr * g.ns * g.Ck) >> 5;      // taps actually visited (all of them unless g.sub)

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), 0, a.Cm * K * 2, 0x00020000);
    const long long x_bytes = SRC2 ? (long long)a.P * a.Ck1 * 2 : (long long)g.img_pitch * 2 * ((a.P + g.Hout * g.Wout - 1) / (g.Hout * g.Wout));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);
    const int Ck2 = g.Ck - a.Ck1;                                   // (SRC2) channels of the second tensor
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(SRC2 ? a.X2 : a.X), 0, SRC2 ? (int)((long long)a.P * Ck2 * 2) : (int)x_bytes, 0x00020000);

    // lane-constant part of the source swizzle: LDS slot `lane` of a block = row lane>>2, physical chunk lane&3
    const int r_in = lane >> 2;
    const int kc = (lane & 3) ^ lds_swz(r_in);                     // logical 16-byte chunk this lane fetches
    // A (weights): rows m0 + 16*(wave + 4*i) + r_in
    uint32_t a_off[A_BLK];
#pragma unroll
    for (int i = 0; i < A_BLK; ++i) {
        const int m = tm * TM + 16 * (wave + 4 * i) + r_in;
        a_off[i] = (m < a.Cm) ? (uint32_t)(m * K + kc * 8) * 2u : DMA_OOB;
    }
    // B (pixels): rows p0 + 16*(wave + 4*i) + r_in
    int b_pix[B_BLK], b_h0[B_BLK], b_w0[B_BLK];
#pragma unroll
    for (int i = 0; i < B_BLK; ++i) {
        const int p = tn * TN + 16 * (wave + 4 * i) + r_in;
        int n = 0, ho = 0, wo = 0;
        const bool ok = p < a.P;
        if (ok) decode_pixel(g, p, n, ho, wo);
        if (g.sub) { ho = 2 * ho + g.oph; wo = 2 * wo + g.opw; }
        if (g.mode == 0) { b_h0[i] = ho * g.stride - g.pad; b_w0[i] = wo * g.stride - g.pad; }
        else { b_h0[i] = ho + g.pad; b_w0[i] = wo + g.pad; }
        if (!ok) b_h0[i] = -0x40000000;                             // never in range
        b_pix[i] = (int)((long long)n * g.img_pitch) + kc * 8;      // element offset of the image (+ this lane's chunk)
        if constexpr (SRC2) b_pix[i] = ok ? p : -1;                  // plain rows: the pixel index itself (row pitch differs between the two tensors)
    }

    // wave-uniform k position
    int kr = g.r0, ks = g.s0, kc0 = 0;
    const int ks_end = g.s0 + g.sstep * g.ns, kr_end = g.r0 + g.rstep * g.nr;
    auto issue = [&](int stage) {
        uint16_t* sa = smem + stage * Cfg::STAGE_ELEMS;
        uint16_t* sb = sa + Cfg::A_ELEMS;
        const int kbase = ((kr * g.S + ks) * g.Ck + kc0) * 2;      // byte offset of this k-tile inside a weight row
#pragma unroll
        for (int i = 0; i < A_BLK; ++i) {
            const uint32_t off = (a_off[i] == DMA_OOB) ? DMA_OOB : a_off[i] + (uint32_t)kbase;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_ptr)(sa + (wave + 4 * i) * 512), 16, off, 0, 0, 0);
        }
        if constexpr (SRC2) {
            // K position kc0 lies in X (< Ck1) or in X2: a wave-uniform choice per k-tile
            const bool second = kc0 >= a.Ck1;
            const int pitch = second ? Ck2 : a.Ck1, cbase = (second ? kc0 - a.Ck1 : kc0) + kc * 8;
#pragma unroll
            for (int i = 0; i < B_BLK; ++i) {
                const uint32_t off = b_pix[i] >= 0 ? (uint32_t)(b_pix[i] * pitch + cbase) * 2u : DMA_OOB;
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void_ptr)(sb + (wave + 4 * i) * 512), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (wave + 4 * i) * 512), 16, off, 0, 0, 0);
            }
            kc0 += 32;
            return;
        }
#pragma unroll
        for (int i = 0; i < B_BLK; ++i) {
            int hi, wi;
            bool ok;
            if (g.mode == 0) {
                hi = b_h0[i] + kr; wi = b_w0[i] + ks;
                ok = true;
            } else {
                const int th = b_h0[i] - kr, tw = b_w0[i] - ks;
                ok = (th >= 0) && (tw >= 0);
                if (g.stride == 2) { ok = ok && (((th | tw) & 1) == 0); hi = th >> 1; wi = tw >> 1; }
                else { hi = th; wi = tw; }
            }
            ok = ok && ((unsigned)hi < (unsigned)g.Hin) && ((unsigned)wi < (unsigned)g.Win);
            const uint32_t off = ok ? (uint32_t)(b_pix[i] + hi * g.row_pitch + wi * g.pix_pitch + kc0) * 2u : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (wave + 4 * i) * 512), 16, off, 0, 0, 0);
        }
        // advance (channel block fastest, then s, then r)
        // taps fastest, channel block outermost: the R*S taps of one channel block are fetched in consecutive k-steps, so the
        // shifted re-reads of the same pixels hit the XCD's L2 (channels-fastest order spaced them Ck/32 steps apart: with
        // Ck >= 128 the line had left the 4 MiB L2 and came back from the Infinity Cache at HBM-like bandwidth, 9 times)
        ks += g.sstep;
        if (ks >= ks_end) { ks = g.s0; kr += g.rstep; if (kr >= kr_end) { kr = g.r0; kc0 += 32; } }
    };

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frag_off = (lane & 15) * 32 + (((lane >> 4) ^ lds_swz(lane & 15)) << 3);
    const int a_row0 = wm * (TM / 2), b_row0 = wn * (TN / 2);

    // 3-stage ring: the DMA of tile t+2 is issued before tile t is multiplied; the wait at the end of iteration t retires
    // tile t+1 only (counted vmcnt leaves tile t+2's A_BLK+B_BLK DMAs in flight ACROSS the barrier, so a raw s_barrier is
    // used: __syncthreads() would drain them).  A stage is read one iteration after the wait+barrier that retired it, and
    // re-filled two barriers after its last read.
    constexpr int NDMA = A_BLK + B_BLK;
    constexpr int AHEAD = NSTAGE - 1;                   // tiles in flight beyond the one being multiplied
    issue(0);
    if (AHEAD == 2 && ktiles > 1) issue(1);
    if (AHEAD == 2 && ktiles > 1) dma_wait<NDMA>(); else dma_wait<0>();
    __builtin_amdgcn_s_barrier();
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 1] = __builtin_amdgcn_s_memrealtime();
    int st_cur = 0, st_nxt2 = AHEAD;
    for (int kt = 0; kt < ktiles; ++kt) {
        if (kt + AHEAD < ktiles) issue(st_nxt2);
        const uint16_t* sa = smem + st_cur * Cfg::STAGE_ELEMS;
        const uint16_t* sb = sa + Cfg::A_ELEMS;
        bf16x8_t fa[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + (a_row0 + i * 16) * 32 + frag_off);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(sb + (b_row0 + j * 16) * 32 + frag_off);
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
        }
        if (AHEAD == 2 && kt + 2 < ktiles) dma_wait<NDMA>(); else dma_wait<0>();
        __builtin_amdgcn_s_barrier();
        st_cur = (st_cur == NSTAGE - 1) ? 0 : st_cur + 1;
        st_nxt2 = (st_nxt2 == NSTAGE - 1) ? 0 : st_nxt2 + 1;
    }
    __syncthreads();
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 2] = __builtin_amdgcn_s_memrealtime();

    conv_epilogue<Cfg, ((TM == 128 && TN == 128 && NSTAGE == 3) || EPI == 3 || EPI == 5) ? EPI : 2>(a, acc, tm, tn, smem);
    if (a.stamps) {
        const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();   // all stores issued (not yet acknowledged)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have been acknowledged
        if (tid == 0) {
            a.stamps[(size_t)blockIdx.x * 12 + 3] = __builtin_amdgcn_s_memrealtime();
            a.stamps[(size_t)blockIdx.x * 12 + 4] = t_issued;
        }
    }
}

// Wave-grid variant for the large layers: WM x WN waves, each owning a 64 x 64 sub-tile (same per-wave code and register
// budget as the 128 x 128 kernel), block tile (64*WM) x (64*WN).  256 x 256 with 16 waves halves the L2->LDS bytes per
// FLOP (measured limiter of the 128 x 128 kernel: ~47 GB/s per CU of operand traffic at 0.8 PFLOP/s) and leaves room
// for a 4-deep LDS ring (3 k-tiles in flight) at one block per CU.
// (launch bounds: 4 waves per SIMD.  The 8-wave shapes are meant to run two workgroups per CU; their linear-layer instantiation compiled
// to 130 VGPRs = 3 waves per SIMD = ONE workgroup of 8 waves per CU, and fc1 forward / fc2 data gradient ran at 440 TFLOP/s for it.)
template <int WM, int WN, int NSTAGE, int EPI = 0>
__global__ __launch_bounds__(WM * WN * 64, 4) void igemm_conv_wg_kernel(IGemmArgs a, int tiles_m, int tiles_n) {
    constexpr int TM = 64 * WM, TN = 64 * WN, NW = WM * WN, NT = NW * 64;
    constexpr int A_BLK = TM / 16 / NW, B_BLK = TN / 16 / NW, NDMA = A_BLK + B_BLK;
    constexpr int A_ELEMS = TM * 32, B_ELEMS = TN * 32, STAGE_ELEMS = A_ELEMS + B_ELEMS;
    constexpr int AHEAD = NSTAGE - 1;
    static_assert(TM % (16 * NW) == 0 && TN % (16 * NW) == 0, "DMA blocks must divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    int tm, tn;
    if (!xcd_tile_map(parity_block(a), tiles_m, tiles_n, tm, tn)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const GatherGeom g = a.g;
    const int K = g.R * g.S * g.Ck;                    // weight row length
    const int ktiles = (g.nr * g.ns * g.Ck) >> 5;      // taps actually visited (all of them unless g.sub)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), 0, a.Cm * K * 2, 0x00020000);
    const long long x_bytes = (long long)g.img_pitch * 2 * ((a.P + g.Hout * g.Wout - 1) / (g.Hout * g.Wout));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);
    const int r_in = lane >> 2;
    const int kc = (lane & 3) ^ lds_swz(r_in);
    uint32_t a_off[A_BLK];
#pragma unroll
    for (int i = 0; i < A_BLK; ++i) {
        const int m = tm * TM + 16 * (wave + NW * i) + r_in;
        a_off[i] = (m < a.Cm) ? (uint32_t)(m * K + kc * 8) * 2u : DMA_OOB;
    }
    int b_pix[B_BLK], b_h0[B_BLK], b_w0[B_BLK];
#pragma unroll
    for (int i = 0; i < B_BLK; ++i) {
        const int p = tn * TN + 16 * (wave + NW * i) + r_in;
        int n = 0, ho = 0, wo = 0;
        const bool ok = p < a.P;
        if (ok) decode_pixel(g, p, n, ho, wo);
        if (g.sub) { ho = 2 * ho + g.oph; wo = 2 * wo + g.opw; }
        if (g.mode == 0) { b_h0[i] = ho * g.stride - g.pad; b_w0[i] = wo * g.stride - g.pad; }
        else { b_h0[i] = ho + g.pad; b_w0[i] = wo + g.pad; }
        if (!ok) b_h0[i] = -0x40000000;
        b_pix[i] = (int)((long long)n * g.img_pitch) + kc * 8;
    }
    int kr = g.r0, ks = g.s0, kc0 = 0;
    const int ks_end = g.s0 + g.sstep * g.ns, kr_end = g.r0 + g.rstep * g.nr;
    auto issue = [&](int stage) {
        uint16_t* sa = smem + stage * STAGE_ELEMS;
        uint16_t* sb = sa + A_ELEMS;
        const int kbase = ((kr * g.S + ks) * g.Ck + kc0) * 2;
#pragma unroll
        for (int i = 0; i < A_BLK; ++i) {
            const uint32_t off = (a_off[i] == DMA_OOB) ? DMA_OOB : a_off[i] + (uint32_t)kbase;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_ptr)(sa + (wave + NW * i) * 512), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_BLK; ++i) {
            int hi, wi;
            bool ok = true;
            if (g.mode == 0) { hi = b_h0[i] + kr; wi = b_w0[i] + ks; }
            else {
                const int th = b_h0[i] - kr, tw = b_w0[i] - ks;
                ok = (th >= 0) && (tw >= 0);
                if (g.stride == 2) { ok = ok && (((th | tw) & 1) == 0); hi = th >> 1; wi = tw >> 1; }
                else { hi = th; wi = tw; }
            }
            ok = ok && ((unsigned)hi < (unsigned)g.Hin) && ((unsigned)wi < (unsigned)g.Win);
            const uint32_t off = ok ? (uint32_t)(b_pix[i] + hi * g.row_pitch + wi * g.pix_pitch + kc0) * 2u : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (wave + NW * i) * 512), 16, off, 0, 0, 0);
        }
        // taps fastest, channel block outermost: the R*S taps of one channel block are fetched in consecutive k-steps, so the
        // shifted re-reads of the same pixels hit the XCD's L2 (channels-fastest order spaced them Ck/32 steps apart: with
        // Ck >= 128 the line had left the 4 MiB L2 and came back from the Infinity Cache at HBM-like bandwidth, 9 times)
        ks += g.sstep;
        if (ks >= ks_end) { ks = g.s0; kr += g.rstep; if (kr >= kr_end) { kr = g.r0; kc0 += 32; } }
    };
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frag_off = (lane & 15) * 32 + (((lane >> 4) ^ lds_swz(lane & 15)) << 3);
    const int a_row0 = wm * 64, b_row0 = wn * 64;
    // prologue: AHEAD tiles in flight, wait for the first
    int issued = 0;
    for (; issued < AHEAD && issued < ktiles; ++issued) issue(issued);
    if (issued == 1) dma_wait<0>();
    else if (issued == 2) dma_wait<NDMA>();
    else dma_wait<2 * NDMA>();
    __builtin_amdgcn_s_barrier();
    int st_cur = 0, st_fill = AHEAD % NSTAGE;
    for (int kt = 0; kt < ktiles; ++kt) {
        const bool more = kt + AHEAD < ktiles;
        if (more) issue(st_fill);
        const uint16_t* sa = smem + st_cur * STAGE_ELEMS;
        const uint16_t* sb = sa + A_ELEMS;
        bf16x8_t fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + (a_row0 + i * 16) * 32 + frag_off);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(sb + (b_row0 + j * 16) * 32 + frag_off);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
        }
        // retire tile kt+1: everything but the youngest min(AHEAD-1, remaining) tiles must have landed
        const int left = ktiles - 1 - kt;                    // tiles after this one
        int inflight_after = left < AHEAD ? left : AHEAD;      // tiles issued and not yet multiplied (excl. current)
        // allowed still-in-flight tiles = inflight_after - 1 (tile kt+1 must be complete)
        if (inflight_after <= 1) dma_wait<0>();
        else if (inflight_after == 2) dma_wait<NDMA>();
        else dma_wait<2 * NDMA>();
        __builtin_amdgcn_s_barrier();
        st_cur = (st_cur == NSTAGE - 1) ? 0 : st_cur + 1;
        st_fill = (st_fill == NSTAGE - 1) ? 0 : st_fill + 1;
    }
    __syncthreads();
    conv_epilogue_g<TM, TN, 4, 4, WN, NT, (WM == 2 && WN == 4 && NSTAGE == 3) ? EPI : 2>(a, acc, tm, tn, smem, wm, wn);
}

// ------------------------------------------------------------------------------------------------
// k-tile of 64 (128-byte operand rows).  Measured with scripts/micro/dma_segments.hip: the L2 -> LDS path of a CU retires
// about one cache line per 3 clocks whatever part of the line is used, so a [rows][32 bf16] k-tile (64 B = half a line per
// row) feeds 70-79 GB/s per CU and a [rows][64 bf16] k-tile (a full 128-B line per row) 98-132 GB/s; the ablated 256 x 256
// kernel (scripts/ablate_conv.py) spent 0.66 us per 32-deep k-step in the feed alone against 0.43 us of MFMA work.
// Same wave grid, per-wave 64 x 64 sub-tile and epilogue as igemm_conv_wg_kernel; a DMA piece is 8 rows x 128 B, a stage is
// (TM + TN) x 128 B, one barrier per 64 deep k-step (2 x 16 MFMAs per wave).  Swizzle: physical 16-byte chunk =
// logical chunk ^ ((row >> 1) & 7): every 16-lane service group of ds_read_b128 ({0-3,12-15,20-27}, ... = 8 rows of one
// chunk + 8 rows of the next) lands on 16 distinct 16-byte bank slots.  Needs Ck % 64 == 0.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int NSTAGE, int FM = 4, int FN = 4, int EPI = 0, bool SRC2 = false>
__global__ __launch_bounds__(WM * WN * 64) void igemm_conv_k64_kernel(IGemmArgs a, int tiles_m, int tiles_n) {
    constexpr int TM = 16 * FM * WM, TN = 16 * FN * WN, NW = WM * WN, NT = NW * 64;     // per-wave sub-tile 16 FM x 16 FN
    constexpr int A_BLK = TM / 8 / NW, B_BLK = (TN / 8 + NW - 1) / NW, NDMA = A_BLK + B_BLK;
    constexpr bool B_RAGGED = (TN / 8) % NW != 0;        // 320 pixels: 40 pieces over 16 waves, the waves of the upper half skip their third
    constexpr int A_ELEMS = TM * 64, B_ELEMS = TN * 64, STAGE_ELEMS = A_ELEMS + B_ELEMS;
    constexpr int AHEAD = NSTAGE - 1;
    static_assert(NW % 2 == 0 && TM % (8 * NW) == 0 && TN % 8 == 0, "DMA pieces: whole 8-row pieces over an even number of waves");
    static_assert(!B_RAGGED || NSTAGE == 2, "a ragged piece count needs the wait-for-all of the 2-stage ring (dma_wait<NDMA> counts pieces per wave)");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    int tm, tn;
    if (!xcd_tile_map(parity_block(a), tiles_m, tiles_n, tm, tn)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    if (a.stamps && tid == 0) {
        a.stamps[(size_t)blockIdx.x * 12] = __builtin_amdgcn_s_memrealtime();
        unsigned hw, xcc;                              // where this workgroup sits (slot 10: HW_ID | XCC_ID << 32)
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.stamps[(size_t)blockIdx.x * 12 + 10] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
    const GatherGeom g = a.g;
    const int K = g.R * g.S * g.Ck;                    // weight row length
    const int ktiles = (g.nr * g.ns * g.Ck) >> 6;      // taps actually visited (all of them unless g.sub)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), 0, a.Cm * K * 2, 0x00020000);
    const long long x_bytes = SRC2 ? (long long)a.P * a.Ck1 * 2 : (long long)g.img_pitch * 2 * ((a.P + g.Hout * g.Wout - 1) / (g.Hout * g.Wout));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);
    const int Ck2 = g.Ck - a.Ck1;                                   // (SRC2) channels of the second tensor
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(SRC2 ? a.X2 : a.X), 0, SRC2 ? (int)((long long)a.P * Ck2 * 2) : (int)x_bytes, 0x00020000);
    // LDS slot `lane` of piece q (rows 8q .. 8q+7) = row 8q + (lane >> 3), physical chunk lane & 7; q = wave + NW * i has the
    // parity of the wave, so the logical chunk this lane fetches is the same for all of its pieces
    const int r_in = lane >> 3;
    const int kc = (lane & 7) ^ (((wave & 1) << 2) | (r_in >> 1));
    uint32_t a_off[A_BLK];
#pragma unroll
    for (int i = 0; i < A_BLK; ++i) {
        const int m = tm * TM + 8 * (wave + NW * i) + r_in;
        a_off[i] = (m < a.Cm) ? (uint32_t)(m * K + kc * 8) * 2u : DMA_OOB;
    }
    int b_pix[B_BLK], b_h0[B_BLK], b_w0[B_BLK];
#pragma unroll
    for (int i = 0; i < B_BLK; ++i) {
        const int p = tn * TN + 8 * (wave + NW * i) + r_in;
        int n = 0, ho = 0, wo = 0;
        const bool ok = p < a.P;
        if (ok) decode_pixel(g, p, n, ho, wo);
        if (g.sub) { ho = 2 * ho + g.oph; wo = 2 * wo + g.opw; }
        if (g.mode == 0) { b_h0[i] = ho * g.stride - g.pad; b_w0[i] = wo * g.stride - g.pad; }
        else { b_h0[i] = ho + g.pad; b_w0[i] = wo + g.pad; }
        if (!ok) b_h0[i] = -0x40000000;
        b_pix[i] = (int)((long long)n * g.img_pitch) + kc * 8;
        if constexpr (SRC2) b_pix[i] = ok ? p : -1;                  // plain rows: the pixel index itself
    }
    int kr = g.r0, ks = g.s0, kc0 = 0;
    const int ks_end = g.s0 + g.sstep * g.ns, kr_end = g.r0 + g.rstep * g.nr;
    auto issue = [&](int stage) {
        uint16_t* sa = smem + stage * STAGE_ELEMS;
        uint16_t* sb = sa + A_ELEMS;
        const int kbase = ((kr * g.S + ks) * g.Ck + kc0) * 2;
#pragma unroll
        for (int i = 0; i < A_BLK; ++i) {
            const uint32_t off = (a_off[i] == DMA_OOB) ? DMA_OOB : a_off[i] + (uint32_t)kbase;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_ptr)(sa + (wave + NW * i) * 512), 16, off, 0, 0, 0);
        }
        if constexpr (SRC2) {                             // K position kc0 lies in X (< Ck1) or in X2: a wave-uniform choice per k-tile
            const bool second = kc0 >= a.Ck1;
            const int pitch = second ? Ck2 : a.Ck1, cbase = (second ? kc0 - a.Ck1 : kc0) + kc * 8;
#pragma unroll
            for (int i = 0; i < B_BLK; ++i) {
                const uint32_t off = b_pix[i] >= 0 ? (uint32_t)(b_pix[i] * pitch + cbase) * 2u : DMA_OOB;
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void_ptr)(sb + (wave + NW * i) * 512), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (wave + NW * i) * 512), 16, off, 0, 0, 0);
            }
            kc0 += 64;
            return;
        }
#pragma unroll
        for (int i = 0; i < B_BLK; ++i) {
            if (B_RAGGED && wave + NW * i >= TN / 8) continue;          // (wave-uniform)
            int hi, wi;
            bool ok = true;
            if (g.mode == 0) { hi = b_h0[i] + kr; wi = b_w0[i] + ks; }
            else {
                const int th = b_h0[i] - kr, tw = b_w0[i] - ks;
                ok = (th >= 0) && (tw >= 0);
                if (g.stride == 2) { ok = ok && (((th | tw) & 1) == 0); hi = th >> 1; wi = tw >> 1; }
                else { hi = th; wi = tw; }
            }
            ok = ok && ((unsigned)hi < (unsigned)g.Hin) && ((unsigned)wi < (unsigned)g.Win);
            const uint32_t off = ok ? (uint32_t)(b_pix[i] + hi * g.row_pitch + wi * g.pix_pitch + kc0) * 2u : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (wave + NW * i) * 512), 16, off, 0, 0, 0);
        }
        ks += g.sstep;                                   // taps fastest, channel block outermost (see igemm_conv_dma_kernel)
        if (ks >= ks_end) { ks = g.s0; kr += g.rstep; if (kr >= kr_end) { kr = g.r0; kc0 += 64; } }
    };
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fragment of the first 32 k: row lane & 15, logical chunk lane >> 4; the second 32 k are chunk + 4 = physical chunk ^ 4
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 3);
    const int a_row0 = wm * (16 * FM), b_row0 = wn * (16 * FN);
    int issued = 0;
    for (; issued < AHEAD && issued < ktiles; ++issued) issue(issued);
    if (issued == 2) dma_wait<NDMA>(); else dma_wait<0>();
    __builtin_amdgcn_s_barrier();
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 1] = __builtin_amdgcn_s_memrealtime();
    int st_cur = 0, st_fill = AHEAD % NSTAGE;
    for (int kt = 0; kt < ktiles; ++kt) {
        if (kt + AHEAD < ktiles) issue(st_fill);
        const uint16_t* sa = smem + st_cur * STAGE_ELEMS;
        const uint16_t* sb = sa + A_ELEMS;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int fo = frag_off ^ (h << 5);
            bf16x8_t fa[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + (a_row0 + i * 16) * 64 + fo);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(sb + (b_row0 + j * 16) * 64 + fo);
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
            }
        }
        // retire tile kt+1: only the youngest tile (if any beyond kt+1) may still be in flight
        const int left = ktiles - 1 - kt;
        if (AHEAD == 2 && left >= 2) dma_wait<NDMA>(); else dma_wait<0>();
        __builtin_amdgcn_s_barrier();
        st_cur = (st_cur == NSTAGE - 1) ? 0 : st_cur + 1;
        st_fill = (st_fill == NSTAGE - 1) ? 0 : st_fill + 1;
    }
    __syncthreads();
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 2] = __builtin_amdgcn_s_memrealtime();
    conv_epilogue_g<TM, TN, FM, FN, WN, NT, (WM == 4 && WN == 4 && NSTAGE == 2 && FM == 4) ? EPI : 2>(a, acc, tm, tn, smem, wm, wn);
    if (a.stamps) {
        const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();   // all stores issued (not yet acknowledged)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            a.stamps[(size_t)blockIdx.x * 12 + 3] = __builtin_amdgcn_s_memrealtime();
            a.stamps[(size_t)blockIdx.x * 12 + 4] = t_issued;
        }
    }
}

// Wave-specialised k-tile-64 kernel: WM x WN consumer waves (64 x 64 sub-tiles: fragment reads + MFMAs + epilogue) and NP
// producer waves that do nothing but the gather arithmetic and the DMA pieces of the ring.  In the kernels above a DMA piece
// costs the issuing wave 60-185 cycles in which it cannot issue MFMAs, and the feed time added to the MFMA time instead of
// hiding under it (ablations in scripts/ablate_conv.py, same finding and same cure as pairdist_dma_kernel in eval.hip).
// The producers leave after the last k-step; the epilogue's barriers then count the consumers only.
template <int WM, int WN, int NP, int NSTAGE, int FM = 4, int FN = 4, int EPI = 0, bool SRC2 = false>      // EPI: as igemm_conv_dma_kernel
__global__ __launch_bounds__((WM * WN + NP) * 64) void igemm_conv_k64s_kernel(IGemmArgs a, int tiles_m, int tiles_n) {
    constexpr int TM = 16 * FM * WM, TN = 16 * FN * WN, NC = WM * WN, NT = NC * 64;       // consumer sub-tile 16 FM x 16 FN
    constexpr int A_BLK = TM / 8 / NP, B_BLK = TN / 8 / NP, NDMA = A_BLK + B_BLK;
    constexpr int A_ELEMS = TM * 64, B_ELEMS = TN * 64, STAGE_ELEMS = A_ELEMS + B_ELEMS;
    constexpr int AHEAD = NSTAGE - 1;
    static_assert(NP % 2 == 0 && TM % (8 * NP) == 0 && TN % (8 * NP) == 0, "DMA pieces must divide evenly over an even number of producers");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    int tm, tn;
    if (!xcd_tile_map(blockIdx.x, tiles_m, tiles_n, tm, tn)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const GatherGeom g = a.g;
    const int ktiles = (g.nr * g.ns * g.Ck) >> 6;      // taps actually visited (all of them unless g.sub)
    if (wave >= NC) {
        // ---------------- producers ----------------
        const int pw = wave - NC;
        const int K = g.R * g.S * g.Ck;                // weight row length
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), 0, a.Cm * K * 2, 0x00020000);
        const long long x_bytes = SRC2 ? (long long)a.P * a.Ck1 * 2 : (long long)g.img_pitch * 2 * ((a.P + g.Hout * g.Wout - 1) / (g.Hout * g.Wout));
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);
        const int Ck2 = g.Ck - a.Ck1;                                   // (SRC2) channels of the second tensor
        const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(SRC2 ? a.X2 : a.X), 0, SRC2 ? (int)((long long)a.P * Ck2 * 2) : (int)x_bytes, 0x00020000);
        const int r_in = lane >> 3;
        const int kc = (lane & 7) ^ (((pw & 1) << 2) | (r_in >> 1));      // see igemm_conv_k64_kernel
        uint32_t a_off[A_BLK];
#pragma unroll
        for (int i = 0; i < A_BLK; ++i) {
            const int m = tm * TM + 8 * (pw + NP * i) + r_in;
            a_off[i] = (m < a.Cm) ? (uint32_t)(m * K + kc * 8) * 2u : DMA_OOB;
        }
        int b_pix[B_BLK], b_h0[B_BLK], b_w0[B_BLK];
#pragma unroll
        for (int i = 0; i < B_BLK; ++i) {
            const int p = tn * TN + 8 * (pw + NP * i) + r_in;
            int n = 0, ho = 0, wo = 0;
            const bool ok = p < a.P;
            if (ok) decode_pixel(g, p, n, ho, wo);
            if (g.sub) { ho = 2 * ho + g.oph; wo = 2 * wo + g.opw; }
            if (g.mode == 0) { b_h0[i] = ho * g.stride - g.pad; b_w0[i] = wo * g.stride - g.pad; }
            else { b_h0[i] = ho + g.pad; b_w0[i] = wo + g.pad; }
            if (!ok) b_h0[i] = -0x40000000;
            b_pix[i] = (int)((long long)n * g.img_pitch) + kc * 8;
            if constexpr (SRC2) b_pix[i] = ok ? p : -1;                  // plain rows: the pixel index itself
        }
        int kr = g.r0, ks = g.s0, kc0 = 0;
        const int ks_end = g.s0 + g.sstep * g.ns, kr_end = g.r0 + g.rstep * g.nr;
        auto issue = [&](int stage) {
            uint16_t* sa = smem + stage * STAGE_ELEMS;
            uint16_t* sb = sa + A_ELEMS;
            const int kbase = ((kr * g.S + ks) * g.Ck + kc0) * 2;
#pragma unroll
            for (int i = 0; i < A_BLK; ++i) {
                const uint32_t off = (a_off[i] == DMA_OOB) ? DMA_OOB : a_off[i] + (uint32_t)kbase;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_ptr)(sa + (pw + NP * i) * 512), 16, off, 0, 0, 0);
            }
            if constexpr (SRC2) {                             // K position kc0 lies in X (< Ck1) or in X2: a wave-uniform choice per k-tile
                const bool second = kc0 >= a.Ck1;
                const int pitch = second ? Ck2 : a.Ck1, cbase = (second ? kc0 - a.Ck1 : kc0) + kc * 8;
#pragma unroll
                for (int i = 0; i < B_BLK; ++i) {
                    const uint32_t off = b_pix[i] >= 0 ? (uint32_t)(b_pix[i] * pitch + cbase) * 2u : DMA_OOB;
                    if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void_ptr)(sb + (pw + NP * i) * 512), 16, off, 0, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (pw + NP * i) * 512), 16, off, 0, 0, 0);
                }
                kc0 += 64;
                return;
            }
#pragma unroll
            for (int i = 0; i < B_BLK; ++i) {
                int hi, wi;
                bool ok = true;
                if (g.mode == 0) { hi = b_h0[i] + kr; wi = b_w0[i] + ks; }
                else {
                    const int th = b_h0[i] - kr, tw = b_w0[i] - ks;
                    ok = (th >= 0) && (tw >= 0);
                    if (g.stride == 2) { ok = ok && (((th | tw) & 1) == 0); hi = th >> 1; wi = tw >> 1; }
                    else { hi = th; wi = tw; }
                }
                ok = ok && ((unsigned)hi < (unsigned)g.Hin) && ((unsigned)wi < (unsigned)g.Win);
                const uint32_t off = ok ? (uint32_t)(b_pix[i] + hi * g.row_pitch + wi * g.pix_pitch + kc0) * 2u : DMA_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (pw + NP * i) * 512), 16, off, 0, 0, 0);
            }
            ks += g.sstep;                                   // taps fastest, channel block outermost (see igemm_conv_dma_kernel)
            if (ks >= ks_end) { ks = g.s0; kr += g.rstep; if (kr >= kr_end) { kr = g.r0; kc0 += 64; } }
        };
        issue(0);
        if constexpr (AHEAD == 2) { if (ktiles > 1) { issue(1); dma_wait<NDMA>(); } else dma_wait<0>(); }
        else dma_wait<0>();
        __builtin_amdgcn_s_barrier();                              // tile 0 visible
        int st_fill = AHEAD % NSTAGE;
        for (int kt = 0; kt < ktiles; ++kt) {
            // the stage of tile kt+AHEAD was last read in iteration kt-1, which every consumer left through the previous barrier
            if (kt + AHEAD < ktiles) {
                issue(st_fill);
                if constexpr (AHEAD == 2) dma_wait<NDMA>(); else dma_wait<0>();                     // tile kt+1 has landed
            } else dma_wait<0>();
            __builtin_amdgcn_s_barrier();
            st_fill = (st_fill == NSTAGE - 1) ? 0 : st_fill + 1;
        }
        return;
    }
    // ---------------- consumers ----------------
    const int wm = wave / WN, wn = wave % WN;
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 3);
    const int a_row0 = wm * (16 * FM), b_row0 = wn * (16 * FN);
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_barrier();
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 1] = __builtin_amdgcn_s_memrealtime();
    int st_cur = 0;
    for (int kt = 0; kt < ktiles; ++kt) {
        const uint16_t* sa = smem + st_cur * STAGE_ELEMS;
        const uint16_t* sb = sa + A_ELEMS;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int fo = frag_off ^ (h << 5);
            bf16x8_t fa[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + (a_row0 + i * 16) * 64 + fo);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(sb + (b_row0 + j * 16) * 64 + fo);
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my reads of this stage are complete before it can be refilled
        __builtin_amdgcn_s_barrier();
        st_cur = (st_cur == NSTAGE - 1) ? 0 : st_cur + 1;
    }
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 2] = __builtin_amdgcn_s_memrealtime();
    conv_epilogue_g<TM, TN, FM, FN, WN, NT, (WM == 2 && WN == 4 && NP == 8 && NSTAGE == 3) ? EPI : 2>(a, acc, tm, tn, smem, wm, wn);
    if (a.stamps) {
        const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) { a.stamps[(size_t)blockIdx.x * 12 + 3] = __builtin_amdgcn_s_memrealtime(); a.stamps[(size_t)blockIdx.x * 12 + 4] = t_issued; }
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with 64 -> 64 channels (layer1's conv2, forward and data gradient), halo reuse (round 3).
// In the GEMM view a 256-pixel tile fetches 9 x 256 pixel rows of 128 bytes (one per tap) + 9 x 8 KB of weights = 369 KB through the
// L2 -> LDS path, and that path (~50 GB/s per CU), not the 38.7 GFLOP, set the 74-78 us of these launches (0.3 of their HBM roof).  Here a
// workgroup fetches the tile's halo patch ONCE, (256 / W + 2) x (W + 2) pixels x 128 bytes = 43.5 KB, and every tap's B fragments are reads
// of that patch at a shifted pixel: 116 KB per tile.  The weights still stream one tap per k-step through a 3-stage ring (all nine are 72 KB:
// with the patch they would not leave room for a second workgroup per CU).
// Tile = 256 consecutive pixels = 256 / W whole rows of one image (W in {16, 32}, H * W a multiple of 256); 4 consumer waves of 64 x 64 and
// 4 producer waves as in igemm_conv_k64s_kernel<1, 4, 4, .>; epilogue shared.  Patch image: [halo pixel][64 ch], 16-byte chunk ^ ((hp >> 1) & 7):
// conflict-free b128 fragment reads when the fragment's first halo pixel is even, two-way for odd starts (no swizzle of this family is
// conflict-free at every shift -- exhaustive search -- and the B reads are a third of the fragment reads).
// mode 0: input pixel (h - 1 + kr, w - 1 + ks); mode 1 (data gradient, stride 1): (h + 1 - kr, w + 1 - ks): the same patch, taps mirrored.
// ------------------------------------------------------------------------------------------------
constexpr int HALO64_PX = 344;                           // >= (256 / W + 2) * (W + 2) for W = 16 (324), 32 (340); 43 DMA pieces of 8 pixels
template <int NSTAGE, int EPI = 0>                    // EPI = 3: the fused output stage (IGemmArgs::out_scale ...), its own instantiation
__global__ __launch_bounds__(512, 4) void igemm_conv_halo64_kernel(IGemmArgs a, int tiles_n) {
    constexpr int TM = 64, TN = 256, NC = 4, NP = 4, NT = NC * 64, FM = 4, FN = 4;
    constexpr int A_ELEMS = TM * 64, RING = NSTAGE * A_ELEMS, NPIECE = HALO64_PX / 8, PPW = (NPIECE + NP - 1) / NP;   // 43 pieces, 11 per producer
    constexpr int AHEAD = NSTAGE - 1;
    static_assert(NSTAGE == 3, "ring depth (the waits below count 2 weight pieces per producer and k-step)");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* patch = smem + RING;
    const int tn = blockIdx.x;
    if (tn >= tiles_n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const GatherGeom g = a.g;
    const int W = g.Wout, H = g.Hout, HC = W + 2, lw = g.lw, lhw = g.lhw;
    const int TR = 256 >> lw, HP = (TR + 2) * HC;
    const int p0 = tn * TN, n_img = p0 >> lhw, row0 = (p0 & ((1 << lhw) - 1)) >> lw;       // first pixel / image / first image row of the tile
    constexpr int ktiles = 9;
    if (wave >= NC) {
        // ---------------- producers: the halo patch once, then one weight tap per k-step ----------------
        const int pw = wave - NC;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), 0, 64 * 576 * 2, 0x00020000);
        const long long x_bytes = (long long)g.img_pitch * 2 * (a.P >> lhw);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);
        const int img_base = (int)((long long)n_img * g.img_pitch);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = pw + NP * i;                              // piece: halo pixels 8q .. 8q + 7
            if (q < NPIECE) {
                const int hp = 8 * q + (lane >> 3);
                const int lc = (lane & 7) ^ ((hp >> 1) & 7);        // logical chunk stored in this lane's physical slot
                const int hr = hp / HC, hc = hp - hr * HC;
                const int ih = row0 - 1 + hr, iw = hc - 1;
                const bool ok = hp < HP && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                const uint32_t off = ok ? (uint32_t)(img_base + (ih * W + iw) * 64 + lc * 8) * 2u : DMA_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(patch + q * 512), 16, off, 0, 0, 0);
            }
        }
        // weight tap kt: rows m = 8 (pw + 4 i) + lane >> 3 of the [64][64] stage image, k-tile-64 swizzle (see igemm_conv_k64_kernel)
        const int r_in = lane >> 3;
        const int kc = (lane & 7) ^ (((pw & 1) << 2) | (r_in >> 1));
        uint32_t a_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a_off[i] = (uint32_t)((8 * (pw + NP * i) + r_in) * 576 + kc * 8) * 2u;
        auto issue_w = [&](int kt) {
            uint16_t* sa = smem + (kt % NSTAGE) * A_ELEMS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t off = a_off[i] + (uint32_t)(kt * 128);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_ptr)(sa + (pw + NP * i) * 512), 16, off, 0, 0, 0);
            }
        };
        issue_w(0); issue_w(1);
        dma_wait<2>();                                             // the patch and tap 0 have landed (tap 1 may be in flight)
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < ktiles; ++kt) {
            if (kt + AHEAD < ktiles) { issue_w(kt + AHEAD); dma_wait<2>(); } else dma_wait<0>();     // tap kt + 1 has landed
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    // ---------------- consumers: wave wn owns all 64 output channels of pixels 64 wn .. 64 wn + 63 ----------------
    const int wn = wave;
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 3);       // weights: as igemm_conv_k64s_kernel
    int hp_base[FN];                                               // halo pixel of tap (0, 0) [mode 0] for this lane's pixel of fragment j
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int p = wn * 64 + j * 16 + (lane & 15);
        hp_base[j] = (p >> lw) * HC + (p & (W - 1));
    }
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < ktiles; ++kt) {
        const uint16_t* sa = smem + (kt % NSTAGE) * A_ELEMS;
        const int kr = kt / 3, ks = kt - 3 * kr;
        const int tap_off = g.mode == 0 ? kr * HC + ks : (2 - kr) * HC + (2 - ks);
        uint32_t boff[FN];                                         // element offset of the first k half of this lane's B fragment
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int hp = hp_base[j] + tap_off;
            boff[j] = (uint32_t)(hp * 64 + (((lane >> 4) ^ ((hp >> 1) & 7)) << 3));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8_t fa[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + (i * 16) * 64 + (frag_off ^ (h << 5)));
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(patch + (boff[j] ^ (uint32_t)(h << 5)));
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my reads of this weight stage are complete before it can be refilled
        __builtin_amdgcn_s_barrier();
    }
    conv_epilogue_g<TM, TN, FM, FN, 4, NT, (EPI == 3 || EPI == 5) ? EPI : 2>(a, acc, 0, tn, smem, 0, wn);
}

}  // namespace dali
#include "fused1x1.h"
namespace dali {

// ------------------------------------------------------------------------------------------------
// wgrad: M = Cm (channels of dY), N = R*S*Ck, K = pixels.  Both operands are stored pixel-major, so the
// MFMA fragments (8 consecutive k per lane) are produced by ds_read_b64_tr_b16 transposing reads.
// LDS image per operand and stage: [32 pixels][128 channels] bf16 (256-byte rows); the 32-byte chunk index is
// XORed with swz(row) = (row&3) | ((row>>3)&1)<<2 so that the 8 rows one 32-lane half touches per read
// ({k0..k0+3} and {k0+8..k0+11}) fall on 8 distinct 32-byte bank groups of the 256-byte bank row.
// Split-K over pixels: block (tile, ks) writes an fp32 slab; a second kernel sums the slabs in a fixed order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wg_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8_t tr_frag(const uint16_t* tile, int chunk32, int lane) {
    // rows 8g+q (first read) and 8g+4+q (second read), g = lane>>4, q = (lane&15)>>2, columns 4*(lane&3)..+3
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row0 = 8 * g + q, row1 = row0 + 4;
    const int off0 = row0 * 128 + ((chunk32 ^ wg_swz(row0)) << 4) + 4 * p;      // in bf16 elements
    const int off1 = row1 * 128 + ((chunk32 ^ wg_swz(row1)) << 4) + 4 * p;
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + off0));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + off1));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// The same reads through inline asm.  With LDS-DMA writes in flight the compiler puts `s_waitcnt vmcnt(0)` in front of the first
// __builtin_amdgcn_ds_read_tr16_b64 of a k-step (it cannot prove that the read does not alias the DMA destination; plain
// ds_read_b128 loads do not get that wait), which makes every k-step wait for the tile it has just requested: the ring never
// overlaps anything.  The asm form is invisible to that analysis; in exchange the caller must wait for the data itself with
// tr_settle (the halves are passed through it as in/out operands, so nothing that uses them can be scheduled above the wait).
struct TrPair { s16x4_t lo, hi; };
__device__ __forceinline__ s16x4_t ds_read_tr_raw(const uint16_t* p) {
    typedef __attribute__((address_space(3))) const uint16_t* lds_cptr_t;
    const uint32_t addr = (uint32_t)(uintptr_t)(lds_cptr_t)p;
    s16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ TrPair tr_frag_raw(const uint16_t* tile, int chunk32, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row0 = 8 * g + q, row1 = row0 + 4;
    const int off0 = row0 * 128 + ((chunk32 ^ wg_swz(row0)) << 4) + 4 * p;
    const int off1 = row1 * 128 + ((chunk32 ^ wg_swz(row1)) << 4) + 4 * p;
    return TrPair{ds_read_tr_raw(tile + off0), ds_read_tr_raw(tile + off1)};
}
__device__ __forceinline__ bf16x8_t tr_join(const TrPair& t) {
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = {t.lo[0], t.lo[1], t.lo[2], t.lo[3], t.hi[0], t.hi[1], t.hi[2], t.hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
// wait until at most N LDS reads of this wave are outstanding; the four pairs are "used and redefined" by the wait
template <int N>
__device__ __forceinline__ void tr_settle(TrPair& a, TrPair& b, TrPair& c, TrPair& d) {
    static_assert(N == 0 || N == 8 || N == 10 || N == 16 || N == 18, "lgkmcnt immediates used by the callers");
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi));
    else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi));
    else if constexpr (N == 10) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi));
    else if constexpr (N == 16) asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi));
    else asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi));
}

__device__ __forceinline__ void tr_settle_all(TrPair& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo), "+v"(a.hi)); }

// fp32 slab store of a wave's FM x FN accumulator tiles at rows m_first + 16 i + 4 (lane >> 4) + r, columns n_first + 16 j + (lane & 15).
// Interior tiles (the common case) take a path without per-element predicates: one row pointer per (i, r), the FN stores of a row at
// constant offsets (stamps: 3.7 us of a 21 us workgroup went into 128 predicated stores with 64-bit address arithmetic each).
template <int FM, int FN>
__device__ __forceinline__ void store_slab_tiles(float* slab, const f32x4_t (&acc)[FM][FN], int m_first, int n_first, int Cm, int Ntot, int lane) {
    const int mr = m_first + (lane >> 4) * 4, nc = n_first + (lane & 15);
    if (m_first + 16 * FM <= Cm && n_first + 16 * FN <= Ntot) {
        float* row = slab + (size_t)mr * Ntot + nc;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float* p = row + (size_t)(i * 16 + rr) * Ntot;
#pragma unroll
                for (int j = 0; j < FN; ++j) p[j * 16] = acc[i][j][rr];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = nc + j * 16;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = mr + i * 16 + rr;
                if (m < Cm && n < Ntot) slab[(size_t)m * Ntot + n] = acc[i][j][rr];
            }
        }
}

template <bool IN_BN>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(WGradArgs a, int tiles_m, int tiles_n) {
    constexpr int TILE = 32 * 128;                       // elements per operand per stage
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * 2 * TILE];   // 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = tiles_m * tiles_n;
    const int tile = blockIdx.x % tiles, ks = blockIdx.x / tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const GatherGeom g = a.g;

    // this thread's fixed 16-byte column chunk (same for both of its rows): 16 chunks per 128-wide row
    const int c16 = tid & 15;
    const int am = m0 + c16 * 8;                         // dY channel of the chunk
    const bool a_ok = am < a.Cm;
    const int bn = n0 + c16 * 8;                         // column in [0, Ntot): tap*Ck + ci
    const bool b_ok = bn < a.Ntot;
    const int tap = b_ok ? bn / g.Ck : 0;
    const int ci = bn - tap * g.Ck;
    const int r = tap / g.S, s = tap - r * g.S;

    const int p_begin = ks * a.pix_per_split;
    const int p_end = min(a.P, p_begin + a.pix_per_split);
    const int ksteps = (p_end > p_begin) ? (p_end - p_begin + 31) >> 5 : 0;

    uint4 ra[2], rb[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 4) + i * 16;
            const int p = p_begin + kt * 32 + row;
            ra[i] = make_uint4(0, 0, 0, 0);
            rb[i] = make_uint4(0, 0, 0, 0);
            if (p < p_end) {
                if (a_ok) ra[i] = *reinterpret_cast<const uint4*>(a.dY + (size_t)p * a.Cm + am);
                if (b_ok) {
                    int n, ho, wo;
                    decode_pixel(g, p, n, ho, wo);
                    const long long off = tap_offset(g, (long long)n * g.img_pitch, ho * g.stride - g.pad, wo * g.stride - g.pad, r, s);
                    if (off >= 0) {
                        uint4 v = *reinterpret_cast<const uint4*>(a.X + off + ci);
                        if constexpr (IN_BN) v = bn_relu_chunk(v, a.in_scale + ci, a.in_shift + ci, a.in_relu);
                        rb[i] = v;
                    }
                }
            }
        }
    };
    auto sstore = [&](int stage) {
        uint16_t* sa = smem + stage * 2 * TILE;
        uint16_t* sb = sa + TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 4) + i * 16;
            const int off = row * 128 + (((c16 >> 1) ^ wg_swz(row)) << 4) + (c16 & 1) * 8;
            *reinterpret_cast<uint4*>(sa + off) = ra[i];
            *reinterpret_cast<uint4*>(sb + off) = rb[i];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (ksteps > 0) {
        gload(0);
        sstore(0);
        __syncthreads();
        for (int kt = 0; kt < ksteps; ++kt) {
            const bool more = (kt + 1) < ksteps;
            if (more) gload(kt + 1);
            const uint16_t* sa = smem + (kt & 1) * 2 * TILE;
            const uint16_t* sb = sa + TILE;
            bf16x8_t fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = tr_frag(sa, wm * 4 + i, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8_t fb = tr_frag(sb, wn * 4 + j, lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
            }
            if (more) sstore((kt + 1) & 1);
            __syncthreads();
        }
    }
    // epilogue: fp32 slab
    float* slab = a.partial + (size_t)ks * a.Cm * a.Ntot;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + rr;
                if (m < a.Cm && n < a.Ntot) slab[(size_t)m * a.Ntot + n] = acc[i][j][rr];
            }
        }
}

// wgrad, LDS-DMA version (the one the net plan uses): same tile / fragment scheme as igemm_wgrad_kernel, operands
// streamed by `buffer_load_dwordx4 ... lds` (1 KiB block = 4 pixel rows x 256 B), swizzle applied on the source chunk.
template <bool COLSUM>
__global__ __launch_bounds__(256) void igemm_wgrad_dma_kernel(WGradArgs a, int tiles_m, int tiles_n) {
    constexpr int TILE = 32 * 128;
    __shared__ __attribute__((aligned(16))) uint16_t smem[3 * 2 * TILE];          // 3-stage ring, 48 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware (k-slice, tile) map: blocks are dealt round-robin to the 8 XCDs; XCD x owns items [x*W/8, (x+1)*W/8) of the slice-major order:
    // the tiles_n (resp. tiles_m) blocks that re-read the same dY (resp. X) pixel range share ONE L2 instead of
    // missing in eight (measured before this map: 20 GB of L2-miss traffic per step in this kernel alone).
    const int tiles = tiles_m * tiles_n;
    const int work = tiles * a.splits, per_xcd = (work + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);     // each XCD owns a contiguous, slice-major run of (ks, tile)
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= work) return;
    const int ks = item / tiles, tile = item - ks * tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12] = __builtin_amdgcn_s_memrealtime();
    const GatherGeom g = a.g;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.dY), 0, a.P * a.Cm * 2, 0x00020000);
    const long long x_bytes = (long long)g.img_pitch * 2 * ((a.P + g.Hout * g.Wout - 1) / (g.Hout * g.Wout));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);

    // lane-constant source chunk: slot lane&15 of row (4*block + lane>>4); swz(row) is the same for blocks w and w+4
    const int r_in = lane >> 4, ps = lane & 15;
    const int c16 = ((((ps >> 1) ^ wg_swz(4 * wave + r_in)) << 1) | (ps & 1));
    const int am = m0 + c16 * 8;
    const bool a_ok = am < a.Cm;
    const int bn = n0 + c16 * 8;
    const bool b_ok = bn < a.Ntot;
    const int tap = b_ok ? bn / g.Ck : 0;
    const int ci = bn - tap * g.Ck;
    const int r = tap / g.S, s = tap - r * g.S;

    const int p_begin = ks * a.pix_per_split;
    const int p_end = min(a.P, p_begin + a.pix_per_split);
    const int ksteps = (p_end > p_begin) ? (p_end - p_begin + 31) >> 5 : 0;

    auto issue = [&](int kt, int stage) {
        uint16_t* sa = smem + stage * 2 * TILE;
        uint16_t* sb = sa + TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = wave + 4 * i;
            const int p = p_begin + kt * 32 + 4 * blk + r_in;
            uint32_t oa = DMA_OOB, ob = DMA_OOB;
            if (p < p_end) {
                if (a_ok) oa = (uint32_t)(p * a.Cm + am) * 2u;
                if (b_ok) {
                    int n, ho, wo;
                    decode_pixel(g, p, n, ho, wo);
                    const int hi = ho * g.stride - g.pad + r, wi = wo * g.stride - g.pad + s;
                    if ((unsigned)hi < (unsigned)g.Hin && (unsigned)wi < (unsigned)g.Win)
                        ob = (uint32_t)((int)((long long)n * g.img_pitch) + hi * g.row_pitch + wi * g.pix_pitch + ci) * 2u;
                }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_ptr)(sa + blk * 512), 16, oa, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + blk * 512), 16, ob, 0, 0, 0);
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // COLSUM: column sums of dY over this split's pixels = one more MFMA per A fragment against an all-ones B operand (every column of
    // the 16 x 16 result is the sum); only the n-tile-0 / wn-0 waves, whose A fragments cover each channel of the m tile exactly once
    const bool do_cs = COLSUM && tn == 0 && wn == 0;
    f32x4_t cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    typedef short ones_vec_t __attribute__((ext_vector_type(8)));
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_vec_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

    if (ksteps > 0) {                                   // same 3-stage ring / counted-vmcnt schedule as igemm_conv_dma_kernel
        issue(0, 0);
        if (ksteps > 1) issue(1, 1);
        if (ksteps > 1) dma_wait<4>(); else dma_wait<0>();
        __builtin_amdgcn_s_barrier();
        if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 1] = __builtin_amdgcn_s_memrealtime();
        int st_cur = 0, st_nxt2 = 2;
        for (int kt = 0; kt < ksteps; ++kt) {
            if (kt + 2 < ksteps) issue(kt + 2, st_nxt2);
            const uint16_t* sa = smem + st_cur * 2 * TILE;
            const uint16_t* sb = sa + TILE;
            // transposing reads through inline asm (see ds_read_tr_raw): 16 reads in order, A (8) then B (8)
            TrPair ra[4], rb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = tr_frag_raw(sa, wm * 4 + i, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[j] = tr_frag_raw(sb, wn * 4 + j, lane);
            tr_settle<8>(ra[0], ra[1], ra[2], ra[3]);
            tr_settle<0>(rb[0], rb[1], rb[2], rb[3]);
            bf16x8_t fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = tr_join(ra[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8_t fb = tr_join(rb[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
            }
            if constexpr (COLSUM) if (do_cs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, cs[i], 0, 0, 0);
            }
            if (kt + 2 < ksteps) dma_wait<4>(); else dma_wait<0>();
            __builtin_amdgcn_s_barrier();
            st_cur = (st_cur == 2) ? 0 : st_cur + 1;
            st_nxt2 = (st_nxt2 == 2) ? 0 : st_nxt2 + 1;
        }
    }
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 2] = __builtin_amdgcn_s_memrealtime();
    float* slab = a.partial + (size_t)ks * a.Cm * a.Ntot;
    store_slab_tiles<4, 4>(slab, acc, m0 + wm * 64, n0 + wn * 64, a.Cm, a.Ntot, lane);
    if constexpr (COLSUM) if (do_cs && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + rr;
                if (m < a.Cm) a.colsum[(size_t)ks * a.Cm + m] = cs[i][rr];
            }
    }
    if (a.stamps) {
        const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) { a.stamps[(size_t)blockIdx.x * 12 + 3] = __builtin_amdgcn_s_memrealtime(); a.stamps[(size_t)blockIdx.x * 12 + 4] = t_issued; }
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 weight gradient with halo reuse.  In the GEMM view above every 128-column group of N is one
// tap's gather, so a block re-loads the same dY pixels and the (shifted) same X pixels for each of the 9 taps: 18 operand
// tiles per (co tile, ci tile, 32 pixels).  Here one 8-wave block owns 128 output channels x 64 input channels x ALL 9
// taps: per k-step (32 consecutive pixels = 32/W full image rows) it fetches the dY tile [32 px][128 co] and ONE halo patch
// of X, (32/W + 2) x (W + 2) pixels x 64 ci, and every tap's B operand is the same LDS patch read at a shifted row
// (halo row (hh + r) * (W + 2) + col + s): 2 operand tiles instead of 18, 4.7 x fewer L2 -> LDS bytes per FLOP.
// Wave (wm, wn) of the 2 x 4 grid owns 64 co x 16 ci x 9 taps (144 accumulator registers).  The halo patch is pixel-major
// [px][64 ci] (128-byte rows, what an LDS-DMA block of 8 pixels writes) with the 32-byte chunk index XORed with
// halo_swz(px) so that the 2 x 4 rows one LDS cycle of a transposing read touches fall into different banks at any shift.
// Needs W in {8, 16, 32}, H*W a power of two >= 32, Cout % 64 == 0, Cin % 64 == 0.  Split-K over pixels as before.
// ------------------------------------------------------------------------------------------------
// XOR on the 32-byte chunk index (4 per 128-byte halo row): a transposing read serves 32 lanes per LDS cycle = 2 lane groups x
// 4 consecutive rows, the groups 8 rows apart: (row & 1) picks the 128-byte half of the 64 banks, this picks the quarter
__device__ __forceinline__ int halo_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
constexpr int W3_A_ELEMS = 32 * 128, W3_B_ELEMS = 16 * 512, W3_STAGE = W3_A_ELEMS + W3_B_ELEMS;      // 8 KiB + 16 KiB

template <int NSTAGE>
__global__ __launch_bounds__(512) void igemm_wgrad3x3_kernel(WGradArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles = tiles_m * tiles_n;
    const int work = tiles * a.splits, per_xcd = (work + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= work) return;
    const int ks = item / tiles, tile = item - ks * tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 128, ci0 = tn * 64;
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12] = __builtin_amdgcn_s_memrealtime();
    const bool m_active = m0 + wm * 64 < a.Cm && a.ablate != 2;          // Cout = 64: the upper half of the co tile is padding, its waves only help with the DMA
    const bool feed = a.ablate != 1;
    const GatherGeom g = a.g;
    const int W = g.Wout, H = g.Hout, Cin = g.Ck, lw = g.lw, lhw = g.lhw;
    const int RW = 32 >> lw, HC = W + 2, HP = (RW + 2) * HC;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.dY), 0, a.P * a.Cm * 2, 0x00020000);
    const long long x_bytes = (long long)g.img_pitch * 2 * (a.P >> lhw);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);

    // dY DMA (block = wave): row 4*wave + lane>>4 of the [32][128] image, swizzled 16-byte slot as in igemm_wgrad_dma_kernel
    const int r_in = lane >> 4, ps = lane & 15;
    const int c16 = ((((ps >> 1) ^ wg_swz(4 * wave + r_in)) << 1) | (ps & 1));
    const int am = m0 + c16 * 8;
    const bool a_ok = am < a.Cm;
    // halo DMA (blocks wave and wave + 8): halo pixel 8*blk + lane>>3, physical 16-byte chunk lane&7
    int hb_hr[2], hb_off[2];
    bool hb_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int hp = 8 * (wave + 8 * i) + (lane >> 3);
        const int lc = (lane & 7) ^ (halo_swz(hp) << 1);                  // logical 16-byte chunk stored in this physical slot
        const int hr = hp / HC, hc = hp - hr * HC;
        hb_hr[i] = hr;
        hb_ok[i] = hp < HP && (unsigned)(hc - 1) < (unsigned)W && ci0 + lc * 8 < Cin;
        hb_off[i] = ((hc - 1) * Cin + ci0 + lc * 8) * 2;                   // bytes inside an image row (+ row / image part per k-step)
    }
    const int p_begin = ks * a.pix_per_split;
    const int p_end = min(a.P, p_begin + a.pix_per_split);
    const int ksteps = (p_end > p_begin) ? (p_end - p_begin + 31) >> 5 : 0;

    auto issue = [&](int kt, int stage) {
        uint16_t* sa = smem + stage * W3_STAGE;
        uint16_t* sb = sa + W3_A_ELEMS;
        const int p0 = p_begin + kt * 32;
        const int p = p0 + 4 * wave + r_in;
        const uint32_t oa = (p < p_end && a_ok) ? (uint32_t)(p * a.Cm + am) * 2u : DMA_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_ptr)(sa + wave * 512), 16, oa, 0, 0, 0);
        const int n = p0 >> lhw, h_base = (p0 & ((1 << lhw) - 1)) >> lw;       // wave-uniform: first image row of this k-step
        const int row_bytes = W * Cin * 2;
        const int img_base = (int)((long long)n * g.img_pitch * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int h = h_base - 1 + hb_hr[i];
            const bool ok = hb_ok[i] && (unsigned)h < (unsigned)H && p0 < p_end;
            const uint32_t ob = ok ? (uint32_t)(img_base + h * row_bytes + hb_off[i]) : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (wave + 8 * i) * 512), 16, ob, 0, 0, 0);
        }
    };

    f32x4_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // transposing read of the halo patch: lane group gq = lane>>4 covers pixels 8*gq .. 8*gq+7 of the k-step
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int hrow_base = ((8 * gq) >> lw) * HC + ((8 * gq) & (W - 1)) + q;    // halo row of (tap 0,0), first of the two reads
    // the 9 taps' read offsets inside the patch do not depend on the k-step: computed once
    int boff0[9], boff1[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
            const int row0 = hrow_base + r * HC + s2, row1 = row0 + 4;
            boff0[r * 3 + s2] = row0 * 64 + ((wn ^ halo_swz(row0)) << 4) + 4 * pp;
            boff1[r * 3 + s2] = row1 * 64 + ((wn ^ halo_swz(row1)) << 4) + 4 * pp;
        }
    auto load_b = [&](const uint16_t* sb, int t) -> TrPair { return TrPair{ds_read_tr_raw(sb + boff0[t]), ds_read_tr_raw(sb + boff1[t])}; };

    // NSTAGE-deep ring, NSTAGE-1 k-steps in flight: with one 8-wave block per CU the bytes in flight are what hides the
    // ~2 us load latency (3 stages: 1.25 us per k-step, load-latency-bound; the MFMA work is 0.5 us)
    constexpr int AHEAD = NSTAGE - 1;
    auto wait_inflight = [&](int n) {                  // n = k-steps that may stay in flight (each = 3 DMA pieces per wave)
        if (n <= 0) dma_wait<0>();
        else if (n == 1) dma_wait<3>();
        else if (n == 2) dma_wait<6>();
        else if (n == 3) dma_wait<9>();
        else dma_wait<12>();
    };
    static_assert(AHEAD >= 1 && AHEAD <= 5, "ring depth");
    if (ksteps > 0) {
        int issued = 0;
        for (; issued < AHEAD && issued < ksteps; ++issued) issue(issued, issued);
        wait_inflight(issued - 1 > 4 ? 4 : issued - 1);
        __builtin_amdgcn_s_barrier();
        if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 1] = __builtin_amdgcn_s_memrealtime();
        int st_cur = 0, st_fill = AHEAD % NSTAGE;
        for (int kt = 0; kt < ksteps; ++kt) {
            if (kt + AHEAD < ksteps && feed) issue(kt + AHEAD, st_fill);
            const uint16_t* sa = smem + st_cur * W3_STAGE;
            const uint16_t* sb = sa + W3_A_ELEMS;
            if (m_active) {
            // 26 transposing reads (inline asm, see ds_read_tr_raw) in program order: A (8), taps 0..3 (8), taps 4..8 (10); the
            // MFMAs of a group start once the reads before the next group's have returned (LDS reads return in order)
            TrPair ra[4], rb[9];
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = tr_frag_raw(sa, wm * 4 + i, lane);
#pragma unroll
            for (int t = 0; t < 9; ++t) rb[t] = load_b(sb, t);
            tr_settle<18>(ra[0], ra[1], ra[2], ra[3]);              // <= 15 outstanding is all the counter can express: A has landed
            tr_settle<10>(rb[0], rb[1], rb[2], rb[3]);
            bf16x8_t fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = tr_join(ra[i]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8_t fb = tr_join(rb[t]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[t][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                       // keep the first 16 MFMAs above the second wait
            tr_settle<0>(rb[4], rb[5], rb[6], rb[7]);
            tr_settle_all(rb[8]);
#pragma unroll
            for (int t = 4; t < 9; ++t) {
                const bf16x8_t fb = tr_join(rb[t]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[t][i], 0, 0, 0);
            }
            }
            // k-step kt+1 must have landed: everything but the youngest min(AHEAD - 1, remaining - 1) k-steps
            const int left = ksteps - 1 - kt;
            const int keep = (left < AHEAD ? left : AHEAD) - 1;
            wait_inflight(keep > 4 ? 4 : keep);
            __builtin_amdgcn_s_barrier();
            st_cur = (st_cur == NSTAGE - 1) ? 0 : st_cur + 1;
            st_fill = (st_fill == NSTAGE - 1) ? 0 : st_fill + 1;
        }
    }
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 2] = __builtin_amdgcn_s_memrealtime();
    float* slab = a.partial + (size_t)ks * a.Cm * a.Ntot;
    if (m_active) {                                      // one row pointer per (i, r), the nine taps Cin columns apart
        float* row = slab + (size_t)(m0 + wm * 64 + (lane >> 4) * 4) * a.Ntot + ci0 + wn * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float* p = row + (size_t)(i * 16 + rr) * a.Ntot;
#pragma unroll
                for (int t = 0; t < 9; ++t) p[t * Cin] = acc[t][i][rr];
            }
    }
    if (a.stamps) {
        const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) { a.stamps[(size_t)blockIdx.x * 12 + 3] = __builtin_amdgcn_s_memrealtime(); a.stamps[(size_t)blockIdx.x * 12 + 4] = t_issued; }
    }
}

// wgrad, wave-grid variant: WM x WN waves of 64 x 64 sub-tiles; the (64*WM) x (64*WN) block tile is held as (TM+TN)/128
// swizzled [32 px][128 ch] LDS images per stage (same image / tr-read scheme as above), NSTAGE-deep ring.
template <int WM, int WN, int NSTAGE, bool COLSUM = false>
__global__ __launch_bounds__(WM * WN * 64, 4) void igemm_wgrad_wg_kernel(WGradArgs a, int tiles_m, int tiles_n) {      // (4 waves per SIMD = two 8-wave workgroups per CU: the column-sum variant allocated 136 registers without the bound)
    constexpr int TM = 64 * WM, TN = 64 * WN, NW = WM * WN;
    constexpr int IMG = 32 * 128, NIMG_A = TM / 128, NIMG = (TM + TN) / 128;
    constexpr int NBLK = NIMG * 8 / NW;                 // 1 KiB DMA blocks per wave per k-step
    constexpr int AHEAD = NSTAGE - 1;
    static_assert((NIMG * 8) % NW == 0 && TM % 128 == 0 && TN % 128 == 0, "unsupported wave grid");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles = tiles_m * tiles_n;
    const int work = tiles * a.splits, per_xcd = (work + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= work) return;
    const int ks = item / tiles, tile = item - ks * tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const GatherGeom g = a.g;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.dY), 0, a.P * a.Cm * 2, 0x00020000);
    const long long x_bytes = (long long)g.img_pitch * 2 * ((a.P + g.Hout * g.Wout - 1) / (g.Hout * g.Wout));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);
    // this wave's DMA blocks: q = wave + NW*i -> image q>>3, block q&7 (rows 4*(q&7) + lane>>4); (q&7) is the same for every i
    const int r_in = lane >> 4, ps = lane & 15;
    const int blk = wave & 7;
    const int c16 = ((((ps >> 1) ^ wg_swz(4 * blk + r_in)) << 1) | (ps & 1));
    // per-block operand column state
    bool is_a[NBLK], col_ok[NBLK];
    int col[NBLK], tap_r[NBLK], tap_s[NBLK], img_of[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int q = wave + NW * i, img = q >> 3;
        img_of[i] = img;
        is_a[i] = img < NIMG_A;
        if (is_a[i]) {
            col[i] = m0 + img * 128 + c16 * 8;
            col_ok[i] = col[i] < a.Cm;
            tap_r[i] = tap_s[i] = 0;
        } else {
            const int bn = n0 + (img - NIMG_A) * 128 + c16 * 8;
            col_ok[i] = bn < a.Ntot;
            const int tap = col_ok[i] ? bn / g.Ck : 0;
            col[i] = bn - tap * g.Ck;
            tap_r[i] = tap / g.S; tap_s[i] = tap - tap_r[i] * g.S;
        }
    }
    const int p_begin = ks * a.pix_per_split;
    const int p_end = min(a.P, p_begin + a.pix_per_split);
    const int ksteps = (p_end > p_begin) ? (p_end - p_begin + 31) >> 5 : 0;

    auto issue = [&](int kt, int stage) {
        uint16_t* base = smem + stage * NIMG * IMG;
        const int p = p_begin + kt * 32 + 4 * blk + r_in;
        int n = 0, ho = 0, wo = 0;
        const bool pok = p < p_end;
        if (pok) decode_pixel(g, p, n, ho, wo);
        const int pix_base = (int)((long long)n * g.img_pitch);
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            uint32_t off = DMA_OOB;
            if (pok && col_ok[i]) {
                if (is_a[i]) off = (uint32_t)(p * a.Cm + col[i]) * 2u;
                else {
                    const int hi = ho * g.stride - g.pad + tap_r[i], wi = wo * g.stride - g.pad + tap_s[i];
                    if ((unsigned)hi < (unsigned)g.Hin && (unsigned)wi < (unsigned)g.Win)
                        off = (uint32_t)(pix_base + hi * g.row_pitch + wi * g.pix_pitch + col[i]) * 2u;
                }
            }
            uint16_t* dst = base + img_of[i] * IMG + blk * 512;
            if (is_a[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_ptr)dst, 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)dst, 16, off, 0, 0, 0);
        }
    };

    // COLSUM: column sums of dY over this split's pixels ride on the GEMM as one more MFMA per A fragment against an all-ones operand
    // (see igemm_wgrad_dma_kernel), in the wn-0 waves, whose A fragments cover each channel of the m tile exactly once.  All n tiles of an
    // (m tile, split) see the same dY: they share the k-steps round-robin (n tile tn takes kt = tn mod tiles_n) and leave tiles_n partial
    // rows -- with the n-tile-0 workgroups alone doing it the launch waited for their 25 % longer main loops (6.6 against 5.0 ms per step).
    // One accumulator for the 4 A fragments: fragment i is multiplied by an operand that is all ones in result columns 4i .. 4i+3 and zero
    // elsewhere, so column 4i of the 16 x 16 result carries fragment i's row sums (16 accumulator registers for the four sums spilled
    // inside the main loop of the 128-register kernel).
    const bool do_cs = COLSUM && wn == 0;
    int cs_turn = tn;                                    // k-steps until this workgroup's next turn
    f32x4_t cs = f32x4_t{0.f, 0.f, 0.f, 0.f};
    typedef unsigned sel_vec_t __attribute__((ext_vector_type(4)));
    const int cs_col = (lane & 15) >> 2;                 // the fragment whose sums this lane's result column carries
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (ksteps > 0) {
        int issued = 0;
        for (; issued < AHEAD && issued < ksteps; ++issued) issue(issued, issued);
        if (issued == 1) dma_wait<0>();
        else if (issued == 2) dma_wait<NBLK>();
        else dma_wait<2 * NBLK>();
        __builtin_amdgcn_s_barrier();
        int st_cur = 0, st_fill = AHEAD % NSTAGE;
        for (int kt = 0; kt < ksteps; ++kt) {
            if (kt + AHEAD < ksteps) issue(kt + AHEAD, st_fill);
            const uint16_t* sa = smem + st_cur * NIMG * IMG + (wm >> 1) * IMG;
            const uint16_t* sb = smem + st_cur * NIMG * IMG + (NIMG_A + (wn >> 1)) * IMG;
            TrPair ra[4], rb[4];                                     // inline-asm transposing reads, see ds_read_tr_raw
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = tr_frag_raw(sa, (wm & 1) * 4 + i, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[j] = tr_frag_raw(sb, (wn & 1) * 4 + j, lane);
            tr_settle<8>(ra[0], ra[1], ra[2], ra[3]);
            tr_settle<0>(rb[0], rb[1], rb[2], rb[3]);
            bf16x8_t fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = tr_join(ra[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8_t fb = tr_join(rb[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
            }
            if constexpr (COLSUM) if (do_cs) {
                if (cs_turn == 0) {
                    int opaque;                          // the select operands are built here, per use: hoisted out of the loop they cost 16 registers
                    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned w = (cs_col + opaque == i) ? 0x3F803F80u : 0u;        // bf16 (1, 1) or (0, 0)
                        cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], __builtin_bit_cast(bf16x8_t, sel_vec_t{w, w, w, w}), cs, 0, 0, 0);
                    }
                    cs_turn = tiles_n;
                }
                --cs_turn;
            }
            const int left = ksteps - 1 - kt;
            const int inflight_after = left < AHEAD ? left : AHEAD;
            if (inflight_after <= 1) dma_wait<0>();
            else if (inflight_after == 2) dma_wait<NBLK>();
            else dma_wait<2 * NBLK>();
            __builtin_amdgcn_s_barrier();
            st_cur = (st_cur == NSTAGE - 1) ? 0 : st_cur + 1;
            st_fill = (st_fill == NSTAGE - 1) ? 0 : st_fill + 1;
        }
    }
    float* slab = a.partial + (size_t)ks * a.Cm * a.Ntot;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + rr;
                if (m < a.Cm && n < a.Ntot) slab[(size_t)m * a.Ntot + n] = acc[i][j][rr];
            }
        }
    if constexpr (COLSUM) if (do_cs && (lane & 3) == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int m = m0 + wm * 64 + cs_col * 16 + (lane >> 4) * 4 + rr;
            if (m < a.Cm) a.colsum[((size_t)ks * tiles_n + tn) * a.Cm + m] = cs[rr];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Pipelined weight gradient (round 3).  Same GEMM, LDS images ([32 px][128 ch], swizzled, transposing fragment reads) and split-K slabs
// as igemm_wgrad_wg_kernel; what changes is WHEN a consumer wave reads its fragments.  In the kernels above a wave reads the 16
// fragment halves of k-step t, waits for them, multiplies, and meets the other waves at the barrier: the LDS reads and the MFMAs of a
// wave never overlap, and since every wave of the workgroup leaves the same barrier together, the two waves of a SIMD read together and
// multiply together as well (stamps: 0.55 us per 32-pixel k-step of a 128 x 256 tile for 0.21 us of MFMA work).  Here the fragments of
// k-step t+1 are requested BEFORE the MFMAs of k-step t (two register sets, loop unrolled twice so that every access is static), so the
// reads run under the multiplies.  The ring's bookkeeping shifts by one k-step: at the barrier E_t that ends iteration t the producers
// guarantee that k-step t+2 has landed, the consumers that their reads of k-steps <= t+1 are complete; during iteration t >= 1 the
// producers request k-step t + NSTAGE - 1 into the stage k-step t-1 left, so NSTAGE - 2 k-steps are in flight while one is read.
// Wave grid: WM x WN consumers of (16 FM) x (16 FN), NP producers.  1 x 4 consumers of 128 x 64 + 4 producers = 8 waves at <= 256 registers
// (one MFMA stream per SIMD, 12 fragments = 24 reads per 32 MFMAs); 2 x 4 consumers of 64 x 64 + 4 producers = 12 waves at <= 168.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Producer wave pw of igemm_wgrad_p_kernel.  A __device__ function of its own rather than a branch of the kernel body: lambdas inside a
// __global__ function are compiled for the host too, and with this code inside the kernel hipcc 7.2 dropped the kernel's HOST stub
// without a diagnostic (the library then failed to load with an undefined symbol).
template <int TM, int TN, int NP, int NSTAGE>
__device__ __forceinline__ void wgrad_p_produce(const WGradArgs& a, uint16_t* smem, int pw, int m0, int n0, int p_begin, int p_end, int ksteps) {
    constexpr int IMG = 32 * 128, NIMG_A = TM / 128, NIMG = (TM + TN) / 128, STAGE = NIMG * IMG;
    constexpr int NBLK = NIMG * 8 / NP;                 // 1 KiB DMA pieces per producer per k-step
    constexpr int NB = NP >= 8 ? 1 : 8 / NP;            // distinct piece rows (q & 7) a producer serves
    constexpr int LEAD = NSTAGE - 3;                    // k-steps that may still be in flight at the barrier (beyond the one that must have landed)
    const int lane = threadIdx.x & 63;
    const GatherGeom g = a.g;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.dY), 0, a.P * a.Cm * 2, 0x00020000);
    const long long x_bytes = (long long)g.img_pitch * 2 * ((a.P + g.Hout * g.Wout - 1) / (g.Hout * g.Wout));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)x_bytes, 0x00020000);
    // pieces q = pw + NP*i -> image q >> 3, piece row q & 7 (pixels 4*(q&7) + lane>>4 of the k-step); wg_swz(4*blk + r) is the same for blk and blk + 4
    const int r_in = lane >> 4, ps = lane & 15;
    const int c16 = ((((ps >> 1) ^ wg_swz(4 * (pw & 7) + r_in)) << 1) | (ps & 1));
    bool is_a[NBLK], col_ok[NBLK];
    int col[NBLK], tap_r[NBLK], tap_s[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int img = (pw + NP * i) >> 3;
        is_a[i] = img < NIMG_A;
        if (is_a[i]) {
            col[i] = m0 + img * 128 + c16 * 8;
            col_ok[i] = col[i] < a.Cm;
            tap_r[i] = tap_s[i] = 0;
        } else {
            const int bn = n0 + (img - NIMG_A) * 128 + c16 * 8;
            col_ok[i] = bn < a.Ntot;
            const int tap = col_ok[i] ? bn / g.Ck : 0;
            col[i] = bn - tap * g.Ck;
            tap_r[i] = tap / g.S; tap_s[i] = tap - tap_r[i] * g.S;
        }
    }
    // 1x1 / stride 1 convolutions and linear layers: both operands are plain [P][ld] matrices, so a piece's source offset is a base
    // plus kt * 32 rows -- one add per piece instead of the pixel decode / tap arithmetic / bounds tests of the general gather (with four
    // producers of six pieces each, that arithmetic, not the matrix pipe, set the k-step: 126 us against 94 for layer4's conv1).
    // Rows past P fall behind num_records of the descriptors and arrive as zeros; pix_per_split is a multiple of 32, so no row of a
    // k-step belongs to the next split.
    const bool flat = g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.pix_pitch == g.Ck && g.row_pitch == g.Win * g.Ck &&
                      g.img_pitch == (long long)g.Hin * g.Win * g.Ck && g.Hin == g.Hout && g.Win == g.Wout;
    uint32_t off0[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int row = p_begin + 4 * ((pw + NP * i) & 7) + r_in;
        off0[i] = col_ok[i] ? (uint32_t)(row * (is_a[i] ? a.Cm : g.Ck) + col[i]) * 2u : DMA_OOB;
    }
    const uint32_t step_a = 32u * (uint32_t)a.Cm * 2u, step_b = 32u * (uint32_t)g.Ck * 2u;
    auto issue = [&](int kt) {
        uint16_t* base = smem + (kt % NSTAGE) * STAGE;
        if (flat) {
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                const int q = pw + NP * i;
                uint16_t* dst = base + (q >> 3) * IMG + (q & 7) * 512;
                // (DMA_OOB + kt * step stays below 2^32 and above every num_records: the tensors are below 0x7ff00000 bytes)
                const uint32_t off = off0[i] + (uint32_t)kt * (is_a[i] ? step_a : step_b);
                if (is_a[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_ptr)dst, 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)dst, 16, off, 0, 0, 0);
            }
            return;
        }
        int pn[NB], pho[NB], pwo[NB], pp[NB];
        bool pok[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            pp[j] = p_begin + kt * 32 + 4 * ((pw + NP * j) & 7) + r_in;
            pok[j] = pp[j] < p_end;
            int n = 0, ho = 0, wo = 0;
            if (pok[j]) decode_pixel(g, pp[j], n, ho, wo);
            pn[j] = (int)((long long)n * g.img_pitch); pho[j] = ho; pwo[j] = wo;
        }
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            const int j = i % NB, q = pw + NP * i;
            uint32_t off = DMA_OOB;
            if (pok[j] && col_ok[i]) {
                if (is_a[i]) off = (uint32_t)(pp[j] * a.Cm + col[i]) * 2u;
                else {
                    const int hi = pho[j] * g.stride - g.pad + tap_r[i], wi = pwo[j] * g.stride - g.pad + tap_s[i];
                    if ((unsigned)hi < (unsigned)g.Hin && (unsigned)wi < (unsigned)g.Win)
                        off = (uint32_t)(pn[j] + hi * g.row_pitch + wi * g.pix_pitch + col[i]) * 2u;
                }
            }
            uint16_t* dst = base + (q >> 3) * IMG + (q & 7) * 512;
            if (is_a[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_void_ptr)dst, 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)dst, 16, off, 0, 0, 0);
        }
    };
    auto wait_all_but = [&](int n) {                           // all but the youngest n k-steps of this wave's pieces have landed
        if (n <= 0) vm_wait<0>();
        else if (n == 1) vm_wait<NBLK>();
        else if (n == 2) vm_wait<2 * NBLK>();
        else vm_wait<(LEAD > 3 ? 3 : LEAD) * NBLK>();
    };
    constexpr int LEADC = LEAD > 3 ? 3 : LEAD;                 // (the switch above knows 0..3)
    int issued = 0;
    for (; issued < NSTAGE && issued < ksteps; ++issued) issue(issued);
    // k-steps 0 and 1 must have landed before the consumers' first reads
    { const int keep = issued - 2; wait_all_but(keep < 0 ? 0 : (keep > LEADC ? LEADC : keep)); }
    __builtin_amdgcn_s_barrier();                              // P0
    for (int t = 0; t < ksteps; ++t) {
        // k-step t + NSTAGE - 1 into the stage of k-step t - 1, whose reads (iteration t - 2, or the consumers' first reads for k-step 0)
        // were complete at E_{t-2} / E_0: not in iteration 0
        if (t >= 1 && issued < ksteps) { issue(issued); ++issued; }
        // k-step t+2 must have landed at E_t: everything issued beyond it may stay in flight
        const int keep = issued - (t + 3);
        wait_all_but(keep < 0 ? 0 : (keep > LEADC ? LEADC : keep));
        __builtin_amdgcn_s_barrier();                          // E_t
    }
}

template <int WM, int WN, int FM, int FN, int NP, int NSTAGE, bool COLSUM = false>
__global__ __launch_bounds__((WM * WN + NP) * 64) void igemm_wgrad_p_kernel(WGradArgs a, int tiles_m, int tiles_n) {
    constexpr int TM = 16 * FM * WM, TN = 16 * FN * WN, NC = WM * WN;
    constexpr int IMG = 32 * 128, NIMG_A = TM / 128, NIMG = (TM + TN) / 128, STAGE = NIMG * IMG;
    constexpr int NBLK = NIMG * 8 / NP;                 // 1 KiB DMA pieces per producer per k-step
    constexpr int LEAD = NSTAGE - 3;                    // k-steps that may still be in flight at the barrier (beyond the one that must have landed)
    static_assert((NIMG * 8) % NP == 0 && (NP == 4 || NP == 8) && TM % 128 == 0 && TN % 128 == 0, "unsupported wave grid");
    static_assert(NSTAGE >= 4 && LEAD * NBLK <= 63, "ring depth / vmcnt range");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = tiles_m * tiles_n;
    const int work = tiles * a.splits, per_xcd = (work + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= work) return;
    const int ks = item / tiles, tile = item - ks * tiles;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const int p_begin = ks * a.pix_per_split;
    const int p_end = min(a.P, p_begin + a.pix_per_split);
    const int ksteps = (p_end > p_begin) ? (p_end - p_begin + 31) >> 5 : 0;
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12] = __builtin_amdgcn_s_memrealtime();
    if (wave >= NC) {
        // ---------------- producers ---------------- (a __device__ function: see wgrad_p_produce)
        if (ksteps > 0) wgrad_p_produce<TM, TN, NP, NSTAGE>(a, smem, wave - NC, m0, n0, p_begin, p_end, ksteps);
        return;
    }
    // ---------------- consumers ----------------
    const int wm = wave / WN, wn = wave % WN;
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // Byte address of this lane's first read of fragment f inside a stage = base ^ (f << 5): the 32-byte chunk index of a fragment is
    // (first chunk of the wave + f) ^ swizzle(row), the wave's first chunk is a multiple of FM (FN), a power of two, so "+ f" is "^ f" and
    // the whole XOR folds into ONE register per operand (twelve hoisted addresses cost the 128 x 64 variant its last registers); the
    // stage offset (a multiple of 0x6000) is added per k-step and does not touch bits 5-7.  The second read is 4 rows = 1024 bytes on.
    static_assert((FM == 4 || FM == 8) && (FN == 4 || FN == 8) && (STAGE * 2) % 256 == 0, "fragment index folds into the address by XOR");
    const int gq = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int row0 = 8 * gq + q4, swz0 = wg_swz(row0);             // (row0 + 4 has the same swizzle)
    const int ca = wm * FM, cb = wn * FN;                          // first 16-column block of this wave inside the M (N) tile
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)smem;     // (added after the XOR: only the offsets inside the ring need their low bits clear)
    const uint32_t base_a = (uint32_t)((ca >> 3) * IMG * 2 + row0 * 256 + 8 * p4 + (((ca & 7) ^ swz0) << 5));
    const uint32_t base_b = (uint32_t)((NIMG_A + (cb >> 3)) * IMG * 2 + row0 * 256 + 8 * p4 + (((cb & 7) ^ swz0) << 5));
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    auto frag_at = [&](uint32_t addr) -> bf16x8_t {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)addr);
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(uintptr_t)(addr + 1024));
        return __builtin_bit_cast(bf16x8_t, s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    };
    auto load_frags = [&](bf16x8_t (&fa)[FM], bf16x8_t (&fb)[FN], int stage) {
        const uint32_t so = (uint32_t)stage * (uint32_t)(STAGE * 2);
        const uint32_t aa = base_a + so, bb = base_b + so;
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i] = frag_at(lds0 + (aa ^ (uint32_t)(i << 5)));
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j] = frag_at(lds0 + (bb ^ (uint32_t)(j << 5)));
    };
    auto multiply = [&](const bf16x8_t (&fa)[FM], const bf16x8_t (&fb)[FN]) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    // Issue order of one k-step: left alone, the scheduler sinks the next k-step's reads BELOW this k-step's MFMAs (it saves the second
    // register set that way) and the wave then waits for them with an idle matrix pipe.  The pipeline below pins one transposing read
    // between two MFMAs until the 2 (FM + FN) reads are out, the remaining MFMAs follow.
    auto interleave = [&]() {
        constexpr int NRD = 2 * (FM + FN), NMF = FM * FN;
        static_assert(NRD <= NMF, "one read per MFMA gap");
#pragma unroll
        for (int k = 0; k < NRD; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
    };
    // COLSUM (bias gradients of the linear layers): column sums of dY over this split's pixels, as in igemm_wgrad_wg_kernel: one more MFMA per A fragment
    // against a column-select operand (fragment i' of a wave's share lands in result column 4 i'), the n tiles of an (m tile, split) taking the k-steps
    // in turn.  The WN waves of a consumer row hold the same FM A fragments: each sums FM / WN of them (all on the wn-0 wave, 8 extra MFMAs in its turn
    // steps while the other three waited at the barrier: 4.76 ms for the ViT's 48 weight gradients against 4.67 shared).
    constexpr int CSF = FM / WN;
    static_assert(!COLSUM || (FM % WN == 0 && CSF <= 4), "column sums: the fragments must divide over the waves of a row");
    const bool do_cs = COLSUM;
    int cs_turn = tn;
    f32x4_t cs = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int cs_col = (lane & 15) >> 2;
    auto colsum_step = [&](const bf16x8_t (&fa)[FM]) {
        if constexpr (COLSUM) if (do_cs) {
            if (cs_turn == 0) {
                typedef unsigned sel_vec_t __attribute__((ext_vector_type(4)));
                int opaque;                                        // built per use (hoisted they would cost registers across the pipelined loop)
                asm volatile("v_mov_b32 %0, 0" : "=v"(opaque));
                // this wave's share, one branch per wave index with CONSTANT fragment indices inside (a selected fa[q * CSF + i] made fa a
                // dynamically indexed array: 272 bytes of scratch, the kernel four times slower)
                auto share = [&](auto qc) {
                    constexpr int Q = decltype(qc)::value;
#pragma unroll
                    for (int i = 0; i < CSF; ++i) {
                        const unsigned w = (cs_col + opaque == i) ? 0x3F803F80u : 0u;    // bf16 (1, 1) or (0, 0)
                        cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[Q * CSF + i], __builtin_bit_cast(bf16x8_t, sel_vec_t{w, w, w, w}), cs, 0, 0, 0);
                    }
                };
                static_assert(WN == 4, "column sums: four waves per consumer row");
                if (wn == 0) share(std::integral_constant<int, 0>{});
                else if (wn == 1) share(std::integral_constant<int, 1>{});
                else if (wn == 2) share(std::integral_constant<int, 2>{});
                else share(std::integral_constant<int, 3>{});
                cs_turn = tiles_n;
            }
            --cs_turn;
        }
    };
    if (ksteps > 0) {
        bf16x8_t fa0[FM], fb0[FN], fa1[FM], fb1[FN];
        __builtin_amdgcn_s_barrier();                              // P0: k-steps 0 and 1 have landed
        if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 1] = __builtin_amdgcn_s_memrealtime();
        load_frags(fa0, fb0, 0);
        int t = 0, st1 = 1 % NSTAGE;                               // st1 = stage of k-step t + 1
        // two k-steps per trip, no branch inside: the last trip's second request re-reads the last k-step's stage (landed, unused)
        for (; t + 2 <= ksteps; t += 2) {
            load_frags(fa1, fb1, st1);
            multiply(fa0, fb0);
            interleave();
            colsum_step(fa0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of k-step t+1 are complete: its stage may be refilled after E_t
            __builtin_amdgcn_s_barrier();                          // E_t
            const int st2 = (t + 2 < ksteps) ? (st1 + 1 == NSTAGE ? 0 : st1 + 1) : st1;
            load_frags(fa0, fb0, st2);
            multiply(fa1, fb1);
            interleave();
            colsum_step(fa1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // E_{t+1}
            st1 = st2 + 1 == NSTAGE ? 0 : st2 + 1;
        }
        if (t < ksteps) {                                          // odd count: the last k-step is in set 0
            multiply(fa0, fb0);
            colsum_step(fa0);
            __builtin_amdgcn_s_barrier();                          // E_{ksteps-1}
        }
    }
    if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 12 + 2] = __builtin_amdgcn_s_memrealtime();
    float* slab = a.partial + (size_t)ks * a.Cm * a.Ntot;
    store_slab_tiles<FM, FN>(slab, acc, m0 + wm * 16 * FM, n0 + wn * 16 * FN, a.Cm, a.Ntot, lane);
    if constexpr (COLSUM) if (do_cs && (lane & 3) == 0 && cs_col < CSF) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int m = m0 + wm * 16 * FM + (wn * CSF + cs_col) * 16 + (lane >> 4) * 4 + rr;
            if (m < a.Cm) a.colsum[((size_t)ks * tiles_n + tn) * a.Cm + m] = cs[rr];
        }
    }
    if (a.stamps) {
        const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) { a.stamps[(size_t)blockIdx.x * 12 + 3] = __builtin_amdgcn_s_memrealtime(); a.stamps[(size_t)blockIdx.x * 12 + 4] = t_issued; }
    }
}

// Split-K reduce: out[e] (+)= sum_k partial[k][e] in a fixed order (deterministic).  One block covers 64 float4 chunks (1 KiB contiguous
// per slab); wave w of its WAVES waves sums slabs w, w + WAVES, ... in batches of up to 16 independent 16-byte loads per lane, then the
// waves' partial sums are added in wave order through LDS.  Two round-3 findings, both from timing the launch alone
// (scripts/bench_reduce.py, scripts/micro/bench_reduce.hip):
// (1) the batch was written as "k < splits ? load : 0" per element, which hipcc turns into a branch around every load with a wait behind
//     it -- the loads ran one after the other;
// (2) the kernel also held the scalar path for element counts that are not a multiple of 4, which indexed the accumulator through a pointer
//     (sp[e - i4]).  That made the accumulator an array: hipcc moved it to LDS (alloca promotion), indexed by the flattened thread id, whose
//     computation reads the workgroup size from the DISPATCH PACKET (host memory) at the start of every wave.  The path never ran at the plan's
//     shapes and cost them 30 us per launch whatever the wave count (3.5-13 us without it: 34 MB, resp. 76 MB, of slabs).  The odd sizes
//     have their own kernel now; check new kernels for ".amdhsa_user_sgpr_dispatch_ptr 1".
// A second, small job (the column sums that rode on the same weight-gradient GEMM: bias gradients, the Gram scheme's sums) can share the launch:
// the workgroups behind the first job's nb1 take it (a launch of its own cost 5-8 us, 48 times per ViT step).
struct ReduceJob2 { const float* partial; float* out; size_t elems; int splits; unsigned nb1; };
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                  size_t elems, int splits, int accumulate, ReduceJob2 j2) {      // elems % 4 == 0
    __shared__ float4 red[WAVES > 1 ? WAVES * 64 : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned blk = blockIdx.x;
    if (blk >= j2.nb1) { blk -= j2.nb1; partial = j2.partial; out = j2.out; elems = j2.elems; splits = j2.splits; accumulate = 0; }     // (wave-uniform)
    const size_t i4 = ((size_t)blk * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool in = i4 < elems;
    if (in) {
        const float* base = partial + i4;
        int k = w;
        auto batch = [&](auto nconst) {
            constexpr int N = decltype(nconst)::value;
            float4 v[N];
#pragma unroll
            for (int u = 0; u < N; ++u) v[u] = *reinterpret_cast<const float4*>(base + (size_t)(k + u * WAVES) * elems);
#pragma unroll
            for (int u = 0; u < N; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            k += N * WAVES;
        };
        while (k + 15 * WAVES < splits) batch(std::integral_constant<int, 16>{});
        if (k + 7 * WAVES < splits) batch(std::integral_constant<int, 8>{});
        if (k + 3 * WAVES < splits) batch(std::integral_constant<int, 4>{});
        if (k + WAVES < splits) batch(std::integral_constant<int, 2>{});
        if (k < splits) batch(std::integral_constant<int, 1>{});
    }
    if (WAVES > 1) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (w == 0)
            for (int k = 1; k < WAVES; ++k) { const float4 v = red[k * 64 + lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (w == 0 && in) {
        float4* o = reinterpret_cast<float4*>(out + i4);
        if (accumulate) { const float4 p = *o; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
        *o = s;
    }
}

// The same sum for element counts that are not a multiple of 4 (slabs are not 16-byte aligned then: bias gradients of odd widths), one
// element per lane, the same order of additions as the kernel above with `waves` waves (wave partial sums, then wave order).
__global__ __launch_bounds__(256) void splitk_reduce_odd_kernel(const float* __restrict__ partial, float* __restrict__ out, size_t elems,
                                                                int splits, int accumulate, int waves) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= elems) return;
    float s = 0.f;
    for (int w = 0; w < waves; ++w) {
        float p = 0.f;
        for (int k = w; k < splits; k += waves) p += partial[(size_t)k * elems + e];
        s = w ? s + p : p;
    }
    out[e] = accumulate ? out[e] + s : s;
}

static inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace dali

using namespace dali;

namespace dali {

// Tile configuration per problem (measured on MI355X, scripts/bench_convs.py): 64x256 for <= 64 output channels;
// 256x256 / 16 waves / 4-deep ring when both K and Cm are large (operand traffic per FLOP halves: +25..36 % on the
// layer4 3x3); 128x256 / 8 waves for Cm = 128..256 with K >= 1024; 128x128 / 4 waves otherwise (small K: prologue-bound).
enum ConvCfg { CONV_NARROW = 0, CONV_128 = 1, CONV_128x256 = 2, CONV_256x256 = 3, CONV_256x128 = 4, CONV_256x320 = 5 };
static int conv_cfg_override() { return DALI_ENV_INT("DALI_CONV_CFG", -1); }
// DALI_CONV_K64 (A/B aid): 0 = k-tile 32 kernels only, 2 (default) = k-tile 64 kernels on the long-K layers, 6 = the same with
// the unspecialised 128 x 256 kernel, 4 / 3 = k-tile 64 wherever Ck % 64 == 0, the 128 x 128 tile with a 2- / 3-stage ring
static int conv_k64_mode() { return DALI_ENV_INT("DALI_CONV_K64", 2); }
// 256 channels x 320 pixels (k-tile 64, 16 waves of 64 x 80): one workgroup per CU means a launch runs in rounds of 256 tiles, and
// ViT's 25216 x 768 outputs are 297 tiles of 256 x 256 = 2 rounds for 1.16 rounds of work (measured: 145 us, 144 of 256 CUs busy on
// average) but 237 tiles of 256 x 320 = one round of 1.25 x the work.  Taken where rounds x tile size says so by a margin; only for
// launches without BatchNorm statistics (their slab layout is fixed by igemm_conv_stat_tiles before the launch is known).
// (K >= 512; K = 768: 28.9 -> 28.0..28.6 ms per ViT step against a limit of 1024)
static bool conv_prefers_320(int Cm, int P, int K) {
    if (Cm < 512 || P < 16384 || K < 512) return false;
    const long long tm = (Cm + 255) / 256, t256 = tm * ((P + 255) / 256), t320 = tm * ((P + 319) / 320);
    const double c256 = (double)((t256 + 255) / 256), c320 = 1.25 * (double)((t320 + 255) / 256);
    return c320 < 0.85 * c256;
}
int conv_pick_cfg(int Cm, int P, int K) {
    if (Cm <= 64) return CONV_NARROW;
    const int ov = conv_cfg_override();
    if (ov == 0 || ov == 1) return CONV_128;
    if (ov == 4) return Cm >= 256 ? CONV_256x256 : CONV_128;
    if (ov == 6) return Cm >= 256 ? CONV_128x256 : CONV_128;
    if (ov == 7) return Cm >= 256 ? CONV_256x128 : CONV_128;
    if (ov == 9) return Cm >= 128 ? CONV_128x256 : CONV_128;
    if (K >= 1024 && P >= 16384) {
        if (Cm >= 512) return CONV_256x256;              // (the 128 x 256 tile's finer rounds do not pay for ViT's 297-tile launches: +0.7 ms per step)
        if (Cm >= 128) return CONV_128x256;             // Cm = 128 (layer2 3x3): -18 % against 128 x 128
    }
    // ViT's K = 768 layers with 2304 / 3072 outputs: 256 x 256 k-tile 64 (qkv forward 120 -> 107 us, fc1 forward 225 -> 203 us against the
    // 128 x 256 k-tile-32 kernel); ResNet has no K in [768, 1024), and its K = 512 layers lose with either k-tile-64 kernel
    if (K >= 768 && P >= 16384 && Cm >= 1024) return CONV_256x256;
    if (K >= 512 && P >= 16384 && Cm >= 256) return CONV_128x256;     // ViT linears (K = 768, 768 outputs: see conv_prefers_320), layer4 conv3 / layer3 downsample

    return CONV_128;
}
int igemm_conv_stat_tiles(int Cm, int P, int K) {
    const int c = conv_pick_cfg(Cm, P, K);
    return (c == CONV_128 || c == CONV_256x128) ? (P + 127) / 128 : (P + 255) / 256;
}


// ---- live per-launch timing of the MFMA GEMM kernels (bench.py's roofline leg) ------------------------------------
// Between dali_gemm_profile_begin and _end every conv / wgrad MFMA kernel launch is bracketed by two HIP events on
// its own stream; _end synchronises and returns summed durations, launch counts and FLOPs per class
// (0 = igemm_conv_* forward/dgrad, 1 = igemm_wgrad_*; the split-K reduce is not included).  Measurement aid only:
// process-wide, not re-entrant, off by default.
struct GemmProfiler {
    std::vector<hipEvent_t> ev;        // 2 per launch
    std::vector<int> cls;
    std::vector<double> flops;
    std::vector<std::string> desc;     // one line per launch for DALI_GEMM_PROFILE_DUMP
    size_t used = 0, cap = 0;
};
static GemmProfiler* g_prof = nullptr;
struct ProfScope {
    hipStream_t st; bool on = false; size_t i = 0;
    ProfScope(hipStream_t s, int cls, double flops, const char* what = "") : st(s) {
        if (!g_prof || g_prof->used >= g_prof->cap) return;
        on = true; i = g_prof->used++;
        g_prof->cls[i] = cls; g_prof->flops[i] = flops; g_prof->desc[i] = what;
        (void)hipEventRecord(g_prof->ev[2 * i], st);
    }
    ~ProfScope() { if (on) (void)hipEventRecord(g_prof->ev[2 * i + 1], st); }
};

// the same bracket for GEMM launches that live in other translation units (stem.hip): -> slot index, or -1 when no profile step is running
int gemm_profile_begin(hipStream_t st, int cls, double flops, const char* what) {
    if (!g_prof || g_prof->used >= g_prof->cap) return -1;
    const size_t i = g_prof->used++;
    g_prof->cls[i] = cls; g_prof->flops[i] = flops; g_prof->desc[i] = what;
    (void)hipEventRecord(g_prof->ev[2 * i], st);
    return (int)i;
}
void gemm_profile_end(hipStream_t st, int slot) {
    if (slot >= 0 && g_prof) (void)hipEventRecord(g_prof->ev[2 * slot + 1], st);
}

// Host-side launchers shared with the net plan (resnet_plan.hip).
static int launch_igemm_conv_one(hipStream_t st, const IGemmArgs& a);
static bool narrow_cm(int Cm) { return Cm <= 64; }
static unsigned long long* g_conv_stamps = nullptr;      // diagnostic only, see dali_debug_set_conv_stamps

// Stride-2 data gradients are split by output parity: output position (2h'+ph, 2w'+pw) only receives the taps
// kr = (ph+pad) mod 2 + 2i, ks likewise, so each of the four classes is a dense stride-1-like problem on a quarter of
// the pixels with 1, 2, 2 and 4 of a 3x3 kernel's 9 taps (the un-split form multiplies 3/4 zero operands).  A class
// without taps (1x1 kernels: three of four) contributes nothing; it is skipped when the residual is accumulated in
// place (Res == O), otherwise the un-split path is used.
int launch_igemm_conv(hipStream_t st, const IGemmArgs& a) {
    IGemmArgs args = a;
    args.stamps = g_conv_stamps;
    GatherGeom& g = args.g;
    g.sub = 0; g.oph = g.opw = 0; g.Hfull = g.Hout; g.Wfull = g.Wout;
    g.r0 = 0; g.rstep = 1; g.nr = g.R; g.s0 = 0; g.sstep = 1; g.ns = g.S;
    const long long x_bytes = (long long)g.img_pitch * 2 * ((a.P + g.Hout * g.Wout - 1) / (g.Hout * g.Wout));
    const bool dma_ok = !a.in_scale && x_bytes < 0x7ff00000ll && (long long)a.Cm * g.R * g.S * g.Ck * 2 < 0x7ff00000ll;
    if (g.mode == 1 && g.stride == 2 && dma_ok && !a.stats && (g.Hout % 2) == 0 && (g.Wout % 2) == 0 &&
        ilog2_exact(g.Wout / 2) >= 0 && ilog2_exact((g.Hout / 2) * (g.Wout / 2)) >= 0) {
        int nr[2], ns[2];
        for (int ph = 0; ph < 2; ++ph) {
            const int r0 = (ph + g.pad) & 1, s0 = (ph + g.pad) & 1;
            nr[ph] = r0 < g.R ? (g.R - r0 + 1) / 2 : 0;
            ns[ph] = s0 < g.S ? (g.S - s0 + 1) / 2 : 0;
        }
        const bool all_have = nr[0] && nr[1] && ns[0] && ns[1];
        // every class has taps (3x3): ONE launch, the workgroups of the four classes interleaved (parity_block); the four separate launches
        // of a quarter of the pixels each ran one after the other with a short-K tail each (layer2 / layer3 conv2: 96 -> 72 us, 85 -> 53 us)
        if (all_have && !narrow_cm(a.Cm) && (a.Cm & 7) == 0 && conv_cfg_override() < 0) {
            IGemmArgs s = args;
            s.g.sub = 2; s.g.oph = s.g.opw = 0; s.g.Hfull = g.Hout; s.g.Wfull = g.Wout;
            s.g.Hout = g.Hout / 2; s.g.Wout = g.Wout / 2; s.P = a.P / 4;
            s.g.r0 = s.g.s0 = 0; s.g.rstep = s.g.sstep = 2;
            s.g.nr = nr[0] > nr[1] ? nr[0] : nr[1]; s.g.ns = ns[0] > ns[1] ? ns[0] : ns[1];      // the longest class picks the kernel
            return launch_igemm_conv_one(st, s);
        }
        if (all_have || (a.Res != nullptr && a.Res == a.O)) {
            for (int ph = 0; ph < 2; ++ph)
                for (int pw = 0; pw < 2; ++pw) {
                    if (nr[ph] * ns[pw] == 0) continue;
                    IGemmArgs s = args;
                    s.g.sub = 1; s.g.oph = ph; s.g.opw = pw; s.g.Hfull = g.Hout; s.g.Wfull = g.Wout;
                    s.g.Hout = g.Hout / 2; s.g.Wout = g.Wout / 2; s.P = a.P / 4;
                    s.g.r0 = (ph + g.pad) & 1; s.g.rstep = 2; s.g.nr = nr[ph];
                    s.g.s0 = (pw + g.pad) & 1; s.g.sstep = 2; s.g.ns = ns[pw];
                    const int rc = launch_igemm_conv_one(st, s);
                    if (rc) return rc;
                }
            return DALI_OK;
        }
    }
    return launch_igemm_conv_one(st, args);
}

// CUs of the current device rounded down to a multiple of 8 (the persistent streaming kernel's grid: one workgroup per CU, whole XCD groups)
static int f1_cu_count() {
    static int n_cus = 0;
    if (!n_cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { set_error("conv: device query failed"); return -1; }
        n_cus = v > 8 ? v & ~7 : 8;
    }
    return n_cus;
}
// does launch_igemm_conv take [X | X2] (c1 + c2 = K channels, plain rows) with the scale / shift / ReLU output stage at this size?  (dali_conv1x1_cat_act;
// the net plan asks before it folds a downsample branch into conv3)
bool conv_cat_act_supported(int Cm, int c1, int c2, int P, int parts) {
    const int K = parts * (c1 + c2);
    if (parts != 1 && parts != 2) return false;
    if ((c1 & 63) || (c2 & 63) || (Cm % 128) || (long long)P * Cm * 2 >= 0x7ff00000ll || (long long)P * K * 2 >= 0x7ff00000ll) return false;
    if (K <= DALI_ENV_INT("DALI_CONV_PERSIST_KMAX", 256) && DALI_ENV_INT("DALI_CONV_PERSIST", 1) != 0) {
        const int n_cus = f1_cu_count(), tiles_m = Cm / 128, tiles_n = (P + 127) / 128;
        return n_cus > 0 && (long long)tiles_m * tiles_n >= 2ll * n_cus && (n_cus / 8) % tiles_m == 0;
    }
    return parts == 1 && K >= 1024 && Cm >= 512 && P >= 16384 && conv_k64_mode() == 2 && conv_cfg_override() < 0;
}

static int launch_igemm_conv_one(hipStream_t st, const IGemmArgs& a) {
    const bool in_bn = a.in_scale != nullptr;
    const bool narrow = a.Cm <= 64;
    IGemmArgs args = a;
    args.g.lw = ilog2_exact(a.g.Wout);
    args.g.lhw = ilog2_exact(a.g.Hout * a.g.Wout);
    // the LDS-DMA kernel addresses both tensors with 32-bit byte offsets through buffer descriptors
    const long long x_bytes = (long long)a.g.img_pitch * 2 * ((a.P + a.g.Hout * a.g.Wout - 1) / (a.g.Hout * a.g.Wout));
    const bool dma_ok = !in_bn && x_bytes < 0x7ff00000ll && (long long)a.Cm * a.g.R * a.g.S * a.g.Ck * 2 < 0x7ff00000ll;
    const int K = a.g.nr * a.g.ns * a.g.Ck;               // reduction length actually visited
    int cfg = conv_pick_cfg(a.Cm, a.P, K);
    const bool fused_out = a.out_scale || a.out_shift || a.out_relu || a.bits_out || a.out_mask || a.res_scale;
    if (!in_bn && dma_ok && !a.stats && !fused_out && !a.g.sub && a.g.Ck % 64 == 0 && (a.Cm & 7) == 0 && conv_k64_mode() == 2 && conv_cfg_override() < 0 &&
        conv_prefers_320(a.Cm, a.P, K))
        cfg = CONV_256x320;
    if (!dma_ok && !in_bn && a.stats && (cfg == CONV_128x256 || cfg == CONV_256x256 || cfg == CONV_256x128)) {
        set_error("conv: tensors beyond 2 GiB are not supported together with the BatchNorm statistics epilogue");
        return DALI_ERR_LIMIT;
    }
    {
    char what[160] = "";
    if (g_prof)
        snprintf(what, sizeof what, "%s,Cm=%d,K=%d,P=%d,taps=%d,stride=%d,sub=%d,fused=%d,stats=%d,res=%d,mask=%d,lin=%d", a.g.mode ? "dgrad" : "fwd", a.Cm, K, a.P,
                 a.g.R * a.g.S, a.g.stride, a.g.sub, a.out_scale ? 1 : 0, a.stats ? 1 : 0, a.Res ? 1 : 0, (a.res_mask || a.out_mask) ? 1 : 0,
                 (a.bias || a.row_scale || a.act) ? 1 : 0);
    ProfScope prof_scope(st, 0, a.g.sub == 2 ? 2.0 * a.Cm * (double)a.P * a.g.R * a.g.S * a.g.Ck : 2.0 * a.Cm * (double)a.P * K, what);
    // k-tile 64 pays where the main loop dominates (K >= 1024); with a short K or the 128 x 128 tile the smaller k-tile's 2-3
    // co-resident workgroups overlap their epilogues better (measured per layer: 256 x 256 -12..-15 %; 128 x 256 wave-specialised
    // -18..-26 % on the 3x3 layers, -10 % on the K = 1024 1x1 layers; K = 512 layers +12..+20 % with either k-tile-64 kernel)
    const bool lin = a.act != 0 || a.O2 != nullptr || a.dact_pre != nullptr || a.row_scale != nullptr;      // linear-layer epilogue extras: the LIN kernel instantiations
    int k64 = (!in_bn && dma_ok && !narrow && a.g.Ck % 64 == 0) ? conv_k64_mode() : 0;
    int narrow_k64 = (!in_bn && dma_ok && narrow && a.g.Ck % 64 == 0 && K >= 512) ? conv_k64_mode() : 0;   // layer1's 3x3 (Cin = 64: a pixel is one line)
    if (a.X2) {                                         // two operand tensors (IGemmArgs::X2): the SRC2 instantiations of the 128 x 128 / 64 x 256 LDS-DMA kernel
        // a fused output stage beside X2: shift (+ scale) and ReLU only (the inference forward's conv3 + downsample branch as one GEMM), on the two
        // kernels that have that instantiation: the persistent streaming kernel (K <= 256) and the 256 x 256 k-tile-64 kernel
        const bool fo_simple = !a.bits_out && !a.out_mask && !a.res_scale && !a.Res && (a.out_scale || a.out_shift || a.out_relu);
        const bool fo = (a.out_scale || a.out_shift || a.out_relu || a.bits_out || a.out_mask || a.res_scale) && !fo_simple;
        const int xr = a.x_rep > 1 ? a.x_rep : 1;
        if (xr > 1 && !(fo_simple && a.g.Ck % xr == 0 && conv_cat_act_supported(a.Cm, a.Ck1, a.g.Ck / xr - a.Ck1, a.P, xr) && a.g.Ck <= DALI_ENV_INT("DALI_CONV_PERSIST_KMAX", 256))) {
            set_error("conv: split weight images (x_rep = %d) are served by the persistent streaming kernel only (K <= 256, >= 2 tiles per CU)", xr);
            return DALI_ERR_INVALID;
        }
        if (fo_simple && ((a.Ck1 & 63) || ((a.g.Ck / xr - a.Ck1) & 63) || (a.Cm % 128))) {
            set_error("conv: two operand tensors with a fused output stage need channel counts that are multiples of 64 and Cm %% 128 == 0");
            return DALI_ERR_INVALID;
        }
        if (in_bn || !dma_ok || a.stats || fo || lin || a.g.sub || a.g.R != 1 || a.g.S != 1 || a.g.stride != 1 || a.g.pad != 0 || a.Ck1 <= 0 || a.Ck1 >= a.g.Ck / xr ||
            (a.Ck1 & 31) || ((a.g.Ck / xr - a.Ck1) & 31) || (a.Cm & 7)) {
            set_error("conv: a second operand tensor needs a plain 1x1 / stride 1 problem with both channel counts multiples of 32");
            return DALI_ERR_INVALID;
        }
        narrow_k64 = 0;
        if ((a.Ck1 & 63) || ((a.g.Ck / xr - a.Ck1) & 63)) k64 = 0;
        if (!(k64 && (cfg == CONV_256x256 || cfg == CONV_128x256)) && !narrow) { cfg = CONV_128; k64 = 0; }
    }
    if ((k64 == 2 || k64 == 6) && cfg != CONV_256x320 && !((K >= 1024 && (cfg == CONV_256x256 || cfg == CONV_128x256)) || (K >= 768 && cfg == CONV_256x256))) k64 = 0;
    // a second operand tensor is only read by the SRC2 instantiations: the two k-tile-64 kernels above (k64 still set) or, for every other
    // choice -- including a k64 that the K gate has just cleared -- the 128 x 128 / 64 x 256 LDS-DMA kernel
    if (a.X2 && !narrow && !(k64 && (cfg == CONV_256x256 || cfg == CONV_128x256))) { cfg = CONV_128; k64 = 0; }
    // fused output stage (IGemmArgs::out_scale ... out_mask): its own instantiations of three kernels, so that the convolutions' hot
    // instantiations compile none of it (code that is never executed still cost their register allocation 0.4-0.8 ms per step)
    const bool fused = a.out_scale || a.out_shift || a.out_relu || a.bits_out || a.out_mask || a.res_scale;
    if (fused) {
        // (measured, round 4: the 128 x 128 tile for the short-K fused launches instead of 128 x 256: 15.87-15.90 against 15.82-15.84 ms per step)
        if (in_bn || !dma_ok || (a.Cm & 7) || lin) {
            set_error("conv: the fused output stage needs Cm %% 8 == 0, tensors below 2 GiB, no operand transform and no linear-layer extras");
            return DALI_ERR_INVALID;
        }
        DALI_ONCE_PER_DEVICE({
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<4, 4, 2, 4, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (256 + 256) * 64 * 2 * 2));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_wg_kernel<2, 4, 3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (128 + 256) * 32 * 2 * 3));
        });
        // the inference forward (dali_resnet_forward, training = 0) folds every BatchNorm + ReLU into its convolution's output stage, so the
        // 3x3 kernels and the narrow 1x1 kernel have fused instantiations too (the train step never reaches them: its fused launches are the
        // 1x1 conv3 forwards and the masked conv1 data gradients, Cm >= 256 and K <= 512)
        // short-K 1x1 with a residual / mask stream: the persistent streaming kernel (fused1x1.h).  K <= 256: at K = 512 (layer4) its four ring
        // producers cannot issue the 32 DMA pieces of a k-step as fast as the consumers multiply it (141 against 125 us; the block is capped at 16 waves)
        if (DALI_ENV_INT("DALI_CONV_PERSIST", 1) != 0 && a.g.R == 1 && a.g.S == 1 && a.g.stride == 1 && a.g.pad == 0 && !a.g.sub && !a.res_mask &&
            (a.Cm % F1_TM) == 0 && (a.g.Ck & 63) == 0 && a.g.Ck <= DALI_ENV_INT("DALI_CONV_PERSIST_KMAX", 256) && a.g.pix_pitch == a.g.Ck && a.g.row_pitch == a.g.Win * a.g.Ck &&
            a.g.img_pitch == (long long)a.g.Hin * a.g.Win * a.g.Ck && a.g.Hin == a.g.Hout && a.g.Win == a.g.Wout &&
            (long long)a.P * a.Cm * 2 < 0x7ff00000ll && (a.Res || a.out_mask || a.bits_out || a.X2)) {
            const int n_cus = f1_cu_count();
            if (n_cus <= 0) return DALI_ERR_HIP;
            const int tiles_m = a.Cm / F1_TM, tiles_n = (a.P + F1_TN - 1) / F1_TN;
            if ((long long)tiles_m * tiles_n >= 2ll * n_cus && (n_cus / 8) % tiles_m == 0) {     // (every workgroup keeps one channel tile)
                const dim3 f1_block((F1_NC + F1_NP + F1_NR) * 64);
#define DALI_F1_LAUNCH(RES, OM, BITS)                                                                                                                     \
    do {                                                                                                                                                  \
        DALI_ONCE_PER_DEVICE({                                                                                                                            \
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused1x1_persist_kernel<3, 2, true, RES, OM, BITS>), hipFuncAttributeMaxDynamicSharedMemorySize, f1_lds_bytes(3, 2, true, 128))); \
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused1x1_persist_kernel<3, 1, true, RES, OM, BITS>), hipFuncAttributeMaxDynamicSharedMemorySize, f1_lds_bytes(3, 1, true, 256))); \
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused1x1_persist_kernel<3, 1, false, RES, OM, BITS>), hipFuncAttributeMaxDynamicSharedMemorySize, f1_lds_bytes(3, 1, false, 0))); \
        });                                                                                                                                               \
        const int kk = a.g.Ck;                                                                                                                            \
        if (kk <= 128) hipLaunchKernelGGL((fused1x1_persist_kernel<3, 2, true, RES, OM, BITS>), dim3(n_cus), f1_block, f1_lds_bytes(3, 2, true, kk), st, args, tiles_m, tiles_n); \
        else if (kk <= 256) hipLaunchKernelGGL((fused1x1_persist_kernel<3, 1, true, RES, OM, BITS>), dim3(n_cus), f1_block, f1_lds_bytes(3, 1, true, kk), st, args, tiles_m, tiles_n); \
        else hipLaunchKernelGGL((fused1x1_persist_kernel<3, 1, false, RES, OM, BITS>), dim3(n_cus), f1_block, f1_lds_bytes(3, 1, false, kk), st, args, tiles_m, tiles_n); \
    } while (0)
                const bool f_res = a.Res != nullptr, f_om = a.out_mask != nullptr, f_bits = a.bits_out != nullptr;
                if (a.X2) {                                                               // [X | X2] against one weight image, shift + ReLU (inference)
                    DALI_ONCE_PER_DEVICE({
                        DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused1x1_persist_kernel<3, 2, true, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, f1_lds_bytes(3, 2, true, 128)));
                        DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused1x1_persist_kernel<3, 1, true, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, f1_lds_bytes(3, 1, true, 256)));
                    });
                    if (a.g.Ck <= 128) hipLaunchKernelGGL((fused1x1_persist_kernel<3, 2, true, false, false, false, true>), dim3(n_cus), f1_block, f1_lds_bytes(3, 2, true, a.g.Ck), st, args, tiles_m, tiles_n);
                    else hipLaunchKernelGGL((fused1x1_persist_kernel<3, 1, true, false, false, false, true>), dim3(n_cus), f1_block, f1_lds_bytes(3, 1, true, a.g.Ck), st, args, tiles_m, tiles_n);
                } else if (f_res && !f_om && f_bits) DALI_F1_LAUNCH(true, false, true);          // conv3 forward of the train step
                else if (f_res && !f_om && !f_bits) DALI_F1_LAUNCH(true, false, false);   // conv3 forward, inference
                else if (f_res && f_om && !f_bits) DALI_F1_LAUNCH(true, true, false);     // conv1 data gradient + identity gradient, masked
                else if (!f_res && f_om && !f_bits) DALI_F1_LAUNCH(false, true, false);   // masked data gradient without a residual
                else if (f_res && f_om && f_bits) DALI_F1_LAUNCH(true, true, true);
                else if (!f_res && f_om && f_bits) DALI_F1_LAUNCH(false, true, true);
                else DALI_F1_LAUNCH(false, false, true);
#undef DALI_F1_LAUNCH
                DALI_LAUNCH_CHECK();
                return DALI_OK;
            }
        }
        const bool halo_ok = narrow_k64 == 2 && a.Cm == 64 && a.g.Ck == 64 && a.g.R == 3 && a.g.S == 3 && a.g.stride == 1 && a.g.pad == 1 && !a.g.sub &&
                             (a.g.Wout == 16 || a.g.Wout == 32) && args.g.lhw >= 8 && a.g.Hin == a.g.Hout && a.g.Win == a.g.Wout && a.g.pix_pitch == 64 &&
                             a.g.row_pitch == a.g.Win * 64 && a.g.img_pitch == (long long)a.g.Hin * a.g.Win * 64 && a.P % 256 == 0;
        const bool lean = !a.Res && !a.out_mask && !a.bits_out && !a.res_mask && !a.res_scale && !a.X2;      // scale / shift / bias / ReLU only: the EPI = 5 instantiations
        if (halo_ok && lean) {
            const int lds = (3 * 64 * 64 + HALO64_PX * 64) * 2;
            DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_halo64_kernel<3, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
            const int tiles_n = a.P / 256;
            hipLaunchKernelGGL((igemm_conv_halo64_kernel<3, 5>), dim3(tiles_n), dim3(512), lds, st, args, tiles_n);
        } else if (narrow && lean) {
            using Cfg = GemmCfg<64, 256, 1, 1, 1>;
            const int tiles_m = (a.Cm + 63) / 64, tiles_n = (a.P + 255) / 256;
            hipLaunchKernelGGL((igemm_conv_dma_kernel<64, 256, 3, 5>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(256), Cfg::LDS_BYTES / 2 * 3, st, args, tiles_m, tiles_n);
        } else if (lean && k64 && cfg == CONV_128x256 && k64 != 6 && !a.g.sub) {
            const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 255) / 256;
            const int lds = (128 + 256) * 64 * 2 * 3;
            DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64s_kernel<2, 4, 8, 3, 4, 4, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
            hipLaunchKernelGGL((igemm_conv_k64s_kernel<2, 4, 8, 3, 4, 4, 5>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
        } else if (a.X2) {
            // two operand tensors + fused output: the 256 x 256 k-tile-64 kernel only (the dispatch above sends the short-K case to fused1x1)
            if (!(k64 && cfg == CONV_256x256)) {
                set_error("conv: two operand tensors with a fused output stage need K <= 256 or a 256 x 256 k-tile-64 problem (K >= 1024, Cm >= 512, >= 16384 pixels)");
                return DALI_ERR_INVALID;
            }
            const int tiles_m = (a.Cm + 255) / 256, tiles_n = (a.P + 255) / 256;
            DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<4, 4, 2, 4, 4, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (256 + 256) * 64 * 2 * 2)));
            hipLaunchKernelGGL((igemm_conv_k64_kernel<4, 4, 2, 4, 4, 3, true>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), (256 + 256) * 64 * 2 * 2, st, args, tiles_m, tiles_n);
        } else if (k64 && cfg == CONV_256x256) {
            const int tiles_m = (a.Cm + 255) / 256, tiles_n = (a.P + 255) / 256;
            hipLaunchKernelGGL((igemm_conv_k64_kernel<4, 4, 2, 4, 4, 3>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), (256 + 256) * 64 * 2 * 2, st, args, tiles_m, tiles_n);
        } else if (cfg == CONV_128x256 || cfg == CONV_256x256) {
            const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 255) / 256;
            hipLaunchKernelGGL((igemm_conv_wg_kernel<2, 4, 3, 3>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(512), (128 + 256) * 32 * 2 * 3, st, args, tiles_m, tiles_n);
        } else {
            using Cfg = GemmCfg<128, 128, 1, 1, 1>;
            const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 127) / 128;
            hipLaunchKernelGGL((igemm_conv_dma_kernel<128, 128, 3, 3>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(256), Cfg::LDS_BYTES / 2 * 3, st, args, tiles_m, tiles_n);
        }
    } else if (a.g.sub && !in_bn && dma_ok && !narrow && !lin && (a.Cm & 7) == 0 && conv_cfg_override() < 0) {
        // a parity class of a stride-2 data gradient: the staged store with scattered pixel rows (EPI = 4)
        DALI_ONCE_PER_DEVICE({
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<4, 4, 2, 4, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (256 + 256) * 64 * 2 * 2));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_wg_kernel<2, 4, 3, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (128 + 256) * 32 * 2 * 3));
        });
        const int ncls = a.g.sub == 2 ? 4 : 1;          // sub == 2: the four classes in one launch, class_grid workgroups each (parity_block)
        if (k64 && cfg == CONV_256x256) {
            const int tiles_m = (a.Cm + 255) / 256, tiles_n = (a.P + 255) / 256;
            args.g.class_grid = xcd_tile_grid(tiles_m, tiles_n);
            hipLaunchKernelGGL((igemm_conv_k64_kernel<4, 4, 2, 4, 4, 4>), dim3(ncls * args.g.class_grid), dim3(1024), (256 + 256) * 64 * 2 * 2, st, args, tiles_m, tiles_n);
        } else if (cfg == CONV_128x256 || cfg == CONV_256x256) {
            const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 255) / 256;
            args.g.class_grid = xcd_tile_grid(tiles_m, tiles_n);
            hipLaunchKernelGGL((igemm_conv_wg_kernel<2, 4, 3, 4>), dim3(ncls * args.g.class_grid), dim3(512), (128 + 256) * 32 * 2 * 3, st, args, tiles_m, tiles_n);
        } else {
            using Cfg = GemmCfg<128, 128, 1, 1, 1>;
            const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 127) / 128;
            args.g.class_grid = xcd_tile_grid(tiles_m, tiles_n);
            hipLaunchKernelGGL((igemm_conv_dma_kernel<128, 128, 3, 4>), dim3(ncls * args.g.class_grid), dim3(256), Cfg::LDS_BYTES / 2 * 3, st, args, tiles_m, tiles_n);
        }
    } else if (cfg == CONV_256x320) {
        const int tiles_m = (a.Cm + 255) / 256, tiles_n = (a.P + 319) / 320;
        const int lds = (256 + 320) * 64 * 2 * 2;
        DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<4, 4, 2, 4, 5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
        hipLaunchKernelGGL((igemm_conv_k64_kernel<4, 4, 2, 4, 5, 1>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
    } else if (k64 && cfg == CONV_256x256) {
        const int tiles_m = (a.Cm + 255) / 256, tiles_n = (a.P + 255) / 256;
        const int lds = (256 + 256) * 64 * 2 * 2;
        // (8 waves with 128 x 64 per wave, 25 % fewer LDS fragment bytes, 192 VGPRs: measured 3-5 % slower than 16 waves of 64 x 64)
        // and the wave-specialised form (8 consumers of 128 x 64 + 4 producers, 168 VGPRs, 2-stage ring): -2 % on layer4's 3x3, +14 % on
        // the stride-2 downsample dgrad -- a 256 x 256 tile has no room for producers beside 16 consumers (1024 threads per workgroup)
        DALI_ONCE_PER_DEVICE({
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<4, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<4, 4, 2, 4, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        });
        if (a.X2) {
            DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<4, 4, 2, 4, 4, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
            hipLaunchKernelGGL((igemm_conv_k64_kernel<4, 4, 2, 4, 4, 0, true>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
        } else if (lin) hipLaunchKernelGGL((igemm_conv_k64_kernel<4, 4, 2, 4, 4, true>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
        else hipLaunchKernelGGL((igemm_conv_k64_kernel<4, 4, 2>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
    } else if (k64 && cfg == CONV_128x256) {
        const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 255) / 256;
        const int lds = (128 + 256) * 64 * 2 * 3;
        DALI_ONCE_PER_DEVICE({
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<2, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64s_kernel<2, 4, 8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64s_kernel<2, 4, 8, 3, 4, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        });
        if (a.X2) {
            DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64s_kernel<2, 4, 8, 3, 4, 4, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
            hipLaunchKernelGGL((igemm_conv_k64s_kernel<2, 4, 8, 3, 4, 4, 0, true>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
        } else if (k64 == 6) hipLaunchKernelGGL((igemm_conv_k64_kernel<2, 4, 3>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(512), lds, st, args, tiles_m, tiles_n);
        else if (lin) hipLaunchKernelGGL((igemm_conv_k64s_kernel<2, 4, 8, 3, 4, 4, 1>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
        else hipLaunchKernelGGL((igemm_conv_k64s_kernel<2, 4, 8, 3>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
    } else if (k64 && cfg == CONV_128) {
        const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 127) / 128;
        DALI_ONCE_PER_DEVICE({
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 2 * 2));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64_kernel<2, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 64 * 2 * 3));
        });
        if (k64 == 3) hipLaunchKernelGGL((igemm_conv_k64_kernel<2, 2, 3>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(256), 256 * 64 * 2 * 3, st, args, tiles_m, tiles_n);
        else hipLaunchKernelGGL((igemm_conv_k64_kernel<2, 2, 2>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(256), 256 * 64 * 2 * 2, st, args, tiles_m, tiles_n);
    } else if (narrow_k64 == 2 && a.Cm == 64 && a.g.Ck == 64 && a.g.R == 3 && a.g.S == 3 && a.g.stride == 1 && a.g.pad == 1 && !a.g.sub && !lin &&
               (a.g.Wout == 16 || a.g.Wout == 32) && args.g.lhw >= 8 && a.g.Hin == a.g.Hout && a.g.Win == a.g.Wout && a.g.pix_pitch == 64 &&
               a.g.row_pitch == a.g.Win * 64 && a.g.img_pitch == (long long)a.g.Hin * a.g.Win * 64 && a.P % 256 == 0) {
        // layer1's 3x3 (64 -> 64): the halo patch of a 256-pixel tile fetched once, taps read it at shifted pixels (igemm_conv_halo64_kernel)
        const int lds = (3 * 64 * 64 + HALO64_PX * 64) * 2;
        DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_halo64_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
        const int tiles_n = a.P / 256;
        hipLaunchKernelGGL((igemm_conv_halo64_kernel<3>), dim3(tiles_n), dim3(512), lds, st, args, tiles_n);
    } else if (narrow_k64 == 2) {
        // layer1's 3x3 (Cm = Cin = 64, K = 576): k-tile 64 = one full line per pixel and tap, 4 MFMA waves + 4 DMA waves, 2-stage ring,
        // two workgroups per CU: 94 -> 77 us forward, 90 -> 73 us data gradient (unspecialised k-tile 64: 84 / 79; 3-stage ring, one
        // workgroup per CU: 122 / 118)
        const int tiles_m = (a.Cm + 63) / 64, tiles_n = (a.P + 255) / 256;
        const int lds = (64 + 256) * 64 * 2 * 2;
        // (8 producer waves instead of 4: 74 -> 78 us; the L2 -> LDS feed, 9 taps per pixel, bounds it, not the issue of the DMA pieces)
        DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_k64s_kernel<1, 4, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
        hipLaunchKernelGGL((igemm_conv_k64s_kernel<1, 4, 4, 2>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(512), lds, st, args, tiles_m, tiles_n);
    } else if (narrow) {
        using Cfg = GemmCfg<64, 256, 1, 1, 1>;
        const int tiles_m = (a.Cm + 63) / 64, tiles_n = (a.P + 255) / 256;
        const int grid = xcd_tile_grid(tiles_m, tiles_n);
        if (in_bn) hipLaunchKernelGGL((igemm_conv_kernel<64, 256, true>), dim3(grid), dim3(256), Cfg::LDS_BYTES, st, args, tiles_m, tiles_n);
        else if (dma_ok && a.X2) hipLaunchKernelGGL((igemm_conv_dma_kernel<64, 256, 3, 0, true>), dim3(grid), dim3(256), Cfg::LDS_BYTES / 2 * 3, st, args, tiles_m, tiles_n);
        else if (dma_ok) hipLaunchKernelGGL((igemm_conv_dma_kernel<64, 256, 3>), dim3(grid), dim3(256), Cfg::LDS_BYTES / 2 * 3, st, args, tiles_m, tiles_n);
        else hipLaunchKernelGGL((igemm_conv_kernel<64, 256, false>), dim3(grid), dim3(256), Cfg::LDS_BYTES, st, args, tiles_m, tiles_n);
    } else if (!in_bn && dma_ok && cfg == CONV_256x256) {
        const int tiles_m = (a.Cm + 255) / 256, tiles_n = (a.P + 255) / 256;
        const int lds = (256 + 256) * 32 * 2 * 4;
        DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_wg_kernel<4, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
        hipLaunchKernelGGL((igemm_conv_wg_kernel<4, 4, 4>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(1024), lds, st, args, tiles_m, tiles_n);
    } else if (!in_bn && dma_ok && cfg == CONV_256x128) {
        const int tiles_m = (a.Cm + 255) / 256, tiles_n = (a.P + 127) / 128;
        const int lds = (256 + 128) * 32 * 2 * 3;
        DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_wg_kernel<4, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
        hipLaunchKernelGGL((igemm_conv_wg_kernel<4, 2, 3>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(512), lds, st, args, tiles_m, tiles_n);
    } else if (!in_bn && dma_ok && cfg == CONV_128x256) {
        const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 255) / 256;
        const int lds = (128 + 256) * 32 * 2 * 3;
        DALI_ONCE_PER_DEVICE({
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_wg_kernel<2, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_conv_wg_kernel<2, 4, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        });
        if (lin) hipLaunchKernelGGL((igemm_conv_wg_kernel<2, 4, 3, true>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(512), lds, st, args, tiles_m, tiles_n);
        else hipLaunchKernelGGL((igemm_conv_wg_kernel<2, 4, 3>), dim3(xcd_tile_grid(tiles_m, tiles_n)), dim3(512), lds, st, args, tiles_m, tiles_n);
    } else {
        using Cfg = GemmCfg<128, 128, 1, 1, 1>;
        const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.P + 127) / 128;
        const int grid = xcd_tile_grid(tiles_m, tiles_n);
        if (in_bn) hipLaunchKernelGGL((igemm_conv_kernel<128, 128, true>), dim3(grid), dim3(256), Cfg::LDS_BYTES, st, args, tiles_m, tiles_n);
        else if (dma_ok && a.X2) hipLaunchKernelGGL((igemm_conv_dma_kernel<128, 128, 3, 0, true>), dim3(grid), dim3(256), Cfg::LDS_BYTES / 2 * 3, st, args, tiles_m, tiles_n);
        else if (dma_ok && conv_cfg_override() == 0) hipLaunchKernelGGL((igemm_conv_dma_kernel<128, 128, 2>), dim3(grid), dim3(256), Cfg::LDS_BYTES, st, args, tiles_m, tiles_n);
        else if (dma_ok && lin) hipLaunchKernelGGL((igemm_conv_dma_kernel<128, 128, 3, true>), dim3(grid), dim3(256), Cfg::LDS_BYTES / 2 * 3, st, args, tiles_m, tiles_n);
        else if (dma_ok) hipLaunchKernelGGL((igemm_conv_dma_kernel<128, 128, 3>), dim3(grid), dim3(256), Cfg::LDS_BYTES / 2 * 3, st, args, tiles_m, tiles_n);
        else hipLaunchKernelGGL((igemm_conv_kernel<128, 128, false>), dim3(grid), dim3(256), Cfg::LDS_BYTES, st, args, tiles_m, tiles_n);
    }
    }
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

// Chooses the split count so that the grid has ~target blocks; returns slab bytes through *ws_bytes.
// 0: 128x128 (4 waves); 1: 256x256 (16 waves, 4-deep ring) for the large weight matrices
// Tile shape of the weight-gradient GEMM (measured per layer, scripts/bench_convs.py): 0 = 128 x 128 / 4 waves,
// 1 = 256 x 256 / 16 waves, 2 = 128 x 256 / 8 waves.  The wider tiles cut the L2 -> LDS operand bytes per FLOP and win
// on the 1x1 layers with a long pixel (K) dimension; on 3x3 layers (each 128-column group of N is one tap's gather)
// and on the ViT linears (25 k rows: the split-K slabs double) they lose.
// the 128 x 256 weight-gradient kernel: the pipelined wave-specialised form (igemm_wgrad_p_kernel, one workgroup per CU) when the problem has
// few output tiles, the plain 8-wave kernel (two workgroups per CU = 16 MFMA waves) when the tiles alone nearly fill the chip.
// (The limit was 40 tiles until the pipelined kernel carried the column sums: with them it wins at the ViT's 54 / 72 tiles too -- 48 weight
//  gradients 5.46 -> 4.67 ms -- and at layer4's downsample, 64 tiles: 162 -> 125 us.)
static bool wgrad_spec(int Cm, int Ntot) { return ((Cm + 127) / 128) * ((Ntot + 255) / 256) <= 100; }
int wgrad_pick_cfg(int Cm, int Ntot, int taps, int P, int halo_w) {
    const int ov = DALI_ENV_INT("DALI_WGRAD_CFG", -1);
    // 3 = 3x3 halo kernel (128 co x 64 ci x 9 taps per block)
    if (ov != 0 && taps == 9 && (halo_w == 8 || halo_w == 16 || halo_w == 32) && Cm % 64 == 0 && (Ntot / 9) % 64 == 0 && P % 32 == 0) return 3;
    if (ov == 2) return (Ntot >= 256) ? 2 : 0;
    if (ov >= 0) return (ov == 1 && Cm >= 256 && Ntot >= 256) ? 1 : 0;
    if (taps == 7 && Ntot == 224 && P >= 16384) return 2;         // the stem (7 tap rows of 32): one 256-wide n tile, dY read once (187 -> ~155 us)
    return (taps == 1 && Ntot >= 256 && P >= 16384) ? 2 : 0;      // incl. the ViT linears (P = 25216)
}
void wgrad_plan(int Cm, int Ntot, int P, int target_blocks, int* splits, int* pix_per_split, size_t* ws_bytes, int taps, int halo_w) {
    const int cfg = wgrad_pick_cfg(Cm, Ntot, taps, P, halo_w);
    const int TMc = cfg == 1 ? 256 : 128, TNc = cfg == 0 ? 128 : (cfg == 3 ? 9 * 64 : 256);
    if (cfg == 1 || cfg == 3) target_blocks = 256;  // one 16-wave / 8-wave block per CU
    if (cfg == 2) target_blocks = wgrad_spec(Cm, Ntot) ? 256 : 512;     // one 16-wave (specialised) / two 8-wave blocks per CU
    const int tiles = ((Cm + TMc - 1) / TMc) * ((Ntot + TNc - 1) / TNc);
    int sp = (target_blocks + tiles - 1) / tiles;
    const int max_sp = (P + 255) / 256;              // at least 8 k-steps per block
    if (sp > max_sp) sp = max_sp;
    if (sp < 1) sp = 1;
    // the launch runs in rounds of target_blocks workgroups: rounding the split UP can spill a few workgroups into a second round (ViT's
    // 768 x 3072 weight: 72 tiles x 8 = 576 for 512 places; 768 x 768: 18 x 15 = 270 for 256), so take the split with the least
    // rounds(tiles * s) / s instead, the smallest on a tie (fewer slabs).  Power-of-two tile counts (ResNet) keep the split they had.
    {
        int best = sp;
        double best_cost = (double)((tiles * sp + target_blocks - 1) / target_blocks) / sp;
        for (int c = sp - 1; c >= 1 && c >= sp / 2; --c) {
            const double cost = (double)((tiles * c + target_blocks - 1) / target_blocks) / c;
            if (cost <= best_cost + 1e-12) { best = c; best_cost = cost; }
        }
        sp = best;
    }
    int pps = ((P + sp - 1) / sp + 31) & ~31;
    sp = (P + pps - 1) / pps;
    *splits = sp; *pix_per_split = pps;
    *ws_bytes = (size_t)sp * Cm * Ntot * sizeof(float);
}

int launch_igemm_wgrad(hipStream_t st, const WGradArgs& a, float* out, int accumulate, float* colsum_out, int colsum_rows) {
    WGradArgs args = a;
    args.stamps = g_conv_stamps;
    args.ablate = DALI_ENV_INT("DALI_WGRAD_ABLATE", 0);
    args.g.lw = ilog2_exact(a.g.Wout);
    args.g.lhw = ilog2_exact(a.g.Hout * a.g.Wout);
    const int tiles_m = (a.Cm + 127) / 128, tiles_n = (a.Ntot + 127) / 128;
    const int grid = tiles_m * tiles_n * a.splits;
    const long long x_bytes = (long long)a.g.img_pitch * 2 * ((a.P + a.g.Hout * a.g.Wout - 1) / (a.g.Hout * a.g.Wout));
    const bool dma_ok = x_bytes < 0x7ff00000ll && (long long)a.P * a.Cm * 2 < 0x7ff00000ll;
    {
    char what[160] = "";
    if (g_prof)
        snprintf(what, sizeof what, "wgrad,Cm=%d,K=%d,P=%d,taps=%d,stride=%d,sub=0,fused=%d,stats=0,res=0,mask=0,lin=%d", a.Cm, a.Ntot, a.P, a.g.R * a.g.S, a.g.stride,
                 a.in_scale ? 1 : 0, a.colsum ? 1 : 0);
    ProfScope prof_scope(st, 1, 2.0 * a.Cm * (double)a.Ntot * a.P, what);
    const bool halo_ok = a.g.R == 3 && a.g.S == 3 && a.g.stride == 1 && a.g.pad == 1 && a.g.mode == 0 && args.g.lw >= 0 && args.g.lhw >= 7 &&
                         a.g.Hin == a.g.Hout && a.g.Win == a.g.Wout && a.g.pix_pitch == a.g.Ck && a.g.row_pitch == a.g.Win * a.g.Ck;
    const int wcfg = wgrad_pick_cfg(a.Cm, a.Ntot, a.g.R * a.g.S, a.P, halo_ok ? a.g.Wout : 0);
    if (a.colsum && !((wcfg == 0 || wcfg == 2) && dma_ok && !a.in_scale)) {
        set_error("wgrad: column sums ride on the 128 x 128 and 128 x 256 LDS-DMA kernels only (wgrad_colsum_supported)");
        return DALI_ERR_INVALID;
    }
    if (!a.in_scale && dma_ok && wcfg == 3) {
        const int tm3 = (a.Cm + 127) / 128, tn3 = a.g.Ck / 64;
        const int lds = 3 * W3_STAGE * 2;            // 3 stages x 24 KiB (4 and 5 measured the same, before and after the inline-asm reads: the
                                                     // k-step is bound by its 26 transposing reads + 36 MFMAs per wave, two waves per SIMD)
        DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad3x3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
        // (measured and removed, round 3: the same kernel software-pipelined -- layer4 conv2 161 us against 152 -- and as two wave groups half a
        //  k-step apart, 195 us.  PMC on this kernel at the layer4 shape: matrix pipe busy 45 % of the SIMD cycles, LDS active 23 % (28 % of that bank
        //  conflicts), waves parked at s_waitcnt / barriers 26 % of their cycles: neither the issue order nor the LDS bounds it.)
        const dim3 grid3(((tm3 * tn3 * a.splits + 7) / 8) * 8);
        hipLaunchKernelGGL(igemm_wgrad3x3_kernel<3>, grid3, dim3(512), lds, st, args, tm3, tn3);
    } else if (!a.in_scale && dma_ok && wcfg == 1) {
        const int tm2 = (a.Cm + 255) / 256, tn2 = (a.Ntot + 255) / 256;
        const int lds = 4 * 4 * 32 * 128 * 2;       // 4 stages x 4 images x 8 KiB
        DALI_ONCE_PER_DEVICE(DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_wg_kernel<4, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)));
        hipLaunchKernelGGL((igemm_wgrad_wg_kernel<4, 4, 4>), dim3(((tm2 * tn2 * a.splits + 7) / 8) * 8), dim3(1024), lds, st, args, tm2, tn2);
    } else if (!a.in_scale && dma_ok && wcfg == 2) {
        const int tm2 = (a.Cm + 127) / 128, tn2 = (a.Ntot + 255) / 256;
        const int lds = 3 * 3 * 32 * 128 * 2;       // 3 stages x 3 images x 8 KiB
        DALI_ONCE_PER_DEVICE({
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_wg_kernel<2, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_wg_kernel<2, 4, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        });
        const dim3 grid2(((tm2 * tn2 * a.splits + 7) / 8) * 8);
        if (wgrad_spec(a.Cm, a.Ntot)) {                 // few output tiles: the pipelined wave-specialised kernel, one workgroup per CU
            constexpr int lds_p = 6 * 3 * 32 * 128 * 2;   // 6 stages x 3 images x 8 KiB
            DALI_ONCE_PER_DEVICE({
                DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_p_kernel<1, 4, 8, 4, 4, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p));
                DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_p_kernel<2, 4, 4, 4, 8, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p));
                DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_p_kernel<1, 4, 8, 4, 4, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p));
                DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_p_kernel<2, 4, 4, 4, 8, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p));
            });
            // (measured, interleaved: plain [P][C] operands 92 us with 4 consumers of 128 x 64 + 4 producers against 94 with 8 + 8 and 105 before;
            //  strided / gathered operands 83 us with 4 producers of 6 pieces, 75 with 8 of 3)
            const GatherGeom& gg = a.g;
            const bool flat = gg.R == 1 && gg.S == 1 && gg.stride == 1 && gg.pad == 0 && gg.pix_pitch == gg.Ck && gg.row_pitch == gg.Win * gg.Ck &&
                              gg.img_pitch == (long long)gg.Hin * gg.Win * gg.Ck && gg.Hin == gg.Hout && gg.Win == gg.Wout;
            if (a.colsum) {                                     // bias gradients of the linear layers / the Gram scheme's column sums ride on the GEMM
                if (!flat) hipLaunchKernelGGL((igemm_wgrad_p_kernel<2, 4, 4, 4, 8, 6, true>), grid2, dim3(1024), lds_p, st, args, tm2, tn2);
                else hipLaunchKernelGGL((igemm_wgrad_p_kernel<1, 4, 8, 4, 4, 6, true>), grid2, dim3(512), lds_p, st, args, tm2, tn2);
            } else if (!flat) hipLaunchKernelGGL((igemm_wgrad_p_kernel<2, 4, 4, 4, 8, 6>), grid2, dim3(1024), lds_p, st, args, tm2, tn2);
            else hipLaunchKernelGGL((igemm_wgrad_p_kernel<1, 4, 8, 4, 4, 6>), grid2, dim3(512), lds_p, st, args, tm2, tn2);
        } else {
            if (a.colsum) hipLaunchKernelGGL((igemm_wgrad_wg_kernel<2, 4, 3, true>), grid2, dim3(512), lds, st, args, tm2, tn2);
            else hipLaunchKernelGGL((igemm_wgrad_wg_kernel<2, 4, 3>), grid2, dim3(512), lds, st, args, tm2, tn2);
        }
    } else if (a.in_scale) hipLaunchKernelGGL((igemm_wgrad_kernel<true>), dim3(grid), dim3(256), 0, st, args, tiles_m, tiles_n);
    else if (dma_ok && a.colsum) hipLaunchKernelGGL(igemm_wgrad_dma_kernel<true>, dim3(((tiles_m * tiles_n * a.splits + 7) / 8) * 8), dim3(256), 0, st, args, tiles_m, tiles_n);
    else if (dma_ok) hipLaunchKernelGGL(igemm_wgrad_dma_kernel<false>, dim3(((tiles_m * tiles_n * a.splits + 7) / 8) * 8), dim3(256), 0, st, args, tiles_m, tiles_n);
    else hipLaunchKernelGGL((igemm_wgrad_kernel<false>), dim3(grid), dim3(256), 0, st, args, tiles_m, tiles_n);
    }
    DALI_LAUNCH_CHECK();
    if (!out) return DALI_OK;                       // the caller reduces the slabs itself
    // the column sums that rode on the GEMM (a.colsum, colsum_rows partial rows of Cm) are reduced in the same launch
    return launch_splitk_reduce2(st, a.partial, out, (size_t)a.Cm * a.Ntot, a.splits, accumulate, (a.colsum && colsum_out) ? a.colsum : nullptr, colsum_out,
                                 (size_t)a.Cm, colsum_rows);
}

// rows of the column-sum partial slab [rows][Cm] a weight-gradient launch with WGradArgs::colsum leaves: one per split from the
// 128 x 128 kernel (its n-tile-0 workgroups), one per (split, n tile) from the 128 x 256 kernels (every n tile takes a share)
int wgrad_colsum_rows(int Cm, int Ntot, int taps, int P, int splits) {
    return wgrad_pick_cfg(Cm, Ntot, taps, P, 0) == 2 ? splits * ((Ntot + 255) / 256) : splits;
}
bool wgrad_colsum_supported(int Cm, int Ntot, int taps, int P) {
    const int cfg = wgrad_pick_cfg(Cm, Ntot, taps, P, 0);
    return (cfg == 0 || cfg == 2) && (long long)P * Cm * 2 < 0x7ff00000ll && (long long)P * Ntot * 2 < 0x7ff00000ll;
}

int launch_splitk_reduce(hipStream_t st, const float* partial, float* out, size_t elems, int splits, int accumulate) {
    return launch_splitk_reduce2(st, partial, out, elems, splits, accumulate, nullptr, nullptr, 0, 0);
}
// partial2 / out2 (nullable): a second job in the same launch (ReduceJob2); falls back to a launch of its own when either job has an odd element count
int launch_splitk_reduce2(hipStream_t st, const float* partial, float* out, size_t elems, int splits, int accumulate,
                          const float* partial2, float* out2, size_t elems2, int splits2) {
    const size_t chunks = (elems + 3) / 4;
    const unsigned rblocks = (unsigned)((chunks + 63) / 64);
    // waves per chunk column: 16 from 32 slabs on, else 4.  (An adaptive rule -- about 1024 waves per launch, at least 4 slabs per wave --
    // measured the same standalone and 0.1 - 0.2 ms per train step SLOWER in the step; what made the reduce slow
    // was the serialised loads described in the kernel, not the wave count.)  A function of (elems, splits) only, so the summation order
    // of a given weight gradient never changes from step to step.
    const int W = splits >= 32 ? 16 : (splits >= 2 ? 4 : 1);
    const bool second = partial2 && out2 && elems2 > 0;
    if (second && ((elems & 3) || (elems2 & 3))) {        // own launches
        if (int rc = launch_splitk_reduce2(st, partial, out, elems, splits, accumulate, nullptr, nullptr, 0, 0)) return rc;
        return launch_splitk_reduce2(st, partial2, out2, elems2, splits2, 0, nullptr, nullptr, 0, 0);
    }
    if (elems & 3) {
        hipLaunchKernelGGL(splitk_reduce_odd_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, partial, out, elems, splits, accumulate, W);
        DALI_LAUNCH_CHECK();
        return DALI_OK;
    }
    ReduceJob2 j2{nullptr, nullptr, 0, 0, 0xffffffffu};
    unsigned grid = rblocks;
    if (second) { j2 = ReduceJob2{partial2, out2, elems2, splits2, rblocks}; grid += (unsigned)((elems2 / 4 + 63) / 64); }
    switch (W) {
        case 1: hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(grid), dim3(64), 0, st, partial, out, elems, splits, accumulate, j2); break;
        case 4: hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(grid), dim3(256), 0, st, partial, out, elems, splits, accumulate, j2); break;
        default: hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3(grid), dim3(1024), 0, st, partial, out, elems, splits, accumulate, j2); break;
    }
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

}  // namespace dali

// Diagnostic (include/daliid_debug.h): device buffer of 4 x uint64 per block that the LDS-DMA conv kernel fills with
// s_memrealtime stamps (100 MHz); null switches it off.  scripts/conv_block_timeline.py reads it.
extern "C" int dali_debug_set_conv_stamps(void* dev_ptr) { g_conv_stamps = static_cast<unsigned long long*>(dev_ptr); return DALI_OK; }
// Diagnostics for scripts/bench_reduce.py (include/daliid_debug.h): the split count wgrad_plan picks, and the reduce alone.
extern "C" int dali_debug_wgrad_splits(int Cm, int Ntot, int P, int taps, int halo_w) {
    int sp = 0, pps = 0; size_t wsb = 0;
    wgrad_plan(Cm, Ntot, P, 512, &sp, &pps, &wsb, taps, halo_w);
    return sp;
}
extern "C" int dali_debug_splitk_reduce(void* stream, const float* partial, float* out, long long elems, int splits, int accumulate) {
    return launch_splitk_reduce(static_cast<hipStream_t>(stream), partial, out, (size_t)elems, splits, accumulate);
}

extern "C" int dali_gemm_profile_begin(dali_ctx* ctx, int max_launches) {
    DALI_REQUIRE(ctx && max_launches > 0, "dali_gemm_profile_begin: bad argument");
    DALI_REQUIRE(!g_prof, "dali_gemm_profile_begin: a profile is already open");
    GemmProfiler* p = new GemmProfiler();
    p->cap = (size_t)max_launches;
    p->ev.resize(2 * p->cap); p->cls.resize(p->cap); p->flops.resize(p->cap); p->desc.resize(p->cap);
    for (auto& e : p->ev) DALI_HIP(hipEventCreate(&e));
    g_prof = p;
    return DALI_OK;
}

extern "C" int dali_gemm_profile_end(dali_ctx* ctx, double* total_ms, double* total_flops, long long* launches) {
    DALI_REQUIRE(ctx && total_ms && total_flops && launches, "dali_gemm_profile_end: null argument");
    DALI_REQUIRE(g_prof, "dali_gemm_profile_end: no open profile");
    GemmProfiler* p = g_prof;
    g_prof = nullptr;
    for (int c = 0; c < 2; ++c) { total_ms[c] = 0; total_flops[c] = 0; launches[c] = 0; }
    int rc = DALI_OK;
    const char* dump_path = getenv("DALI_GEMM_PROFILE_DUMP");          // development aid: one CSV line per launch (scripts/gemm_launch_table.py)
    FILE* dump = dump_path ? fopen(dump_path, "a") : nullptr;
    for (size_t i = 0; i < p->used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(p->ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]) != hipSuccess) {
            set_error("gemm_profile_end: event query failed"); rc = DALI_ERR_HIP; break;
        }
        total_ms[p->cls[i]] += ms; total_flops[p->cls[i]] += p->flops[i]; launches[p->cls[i]] += 1;
        if (dump) fprintf(dump, "%zu,%s,us=%.2f,gflop=%.3f\n", i, p->desc[i].c_str(), ms * 1e3, p->flops[i] * 1e-9);
    }
    if (dump) fclose(dump);
    for (auto& e : p->ev) (void)hipEventDestroy(e);
    delete p;
    return rc;
}

static int fill_geom(GatherGeom& g, int n_img, int Hin, int Win, int Ck, int Hout, int Wout, int R, int S, int stride, int pad, int mode) {
    g.Hout = Hout; g.Wout = Wout; g.Hin = Hin; g.Win = Win; g.Ck = Ck; g.R = R; g.S = S; g.stride = stride; g.pad = pad;
    g.mode = mode;
    g.pix_pitch = Ck; g.row_pitch = Win * Ck; g.img_pitch = (long long)Hin * Win * Ck;
    g.lw = g.lhw = -1;
    (void)n_img;
    return 0;
}

// ---- single-op C ABI (used by the parity tests; the net plan calls the launchers directly) ----------------
extern "C" int dali_conv2d_fwd(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, uint16_t* y,
                               int n, int h, int wd, int cin, int cout, int r, int s, int stride, int pad,
                               const float* in_scale, const float* in_shift, int in_relu, float* stats) {
    DALI_REQUIRE(ctx && x && w && y, "dali_conv2d_fwd: null argument");
    DALI_REQUIRE(cin % 32 == 0 && cout % 4 == 0, "dali_conv2d_fwd: cin must be a multiple of 32 and cout of 4 (cin=%d cout=%d)", cin, cout);
    DALI_REQUIRE(stride == 1 || stride == 2, "dali_conv2d_fwd: stride %d unsupported", stride);
    DALI_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dali_conv2d_fwd: in_scale/in_shift must come together");
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (wd + 2 * pad - s) / stride + 1;
    IGemmArgs a{};
    a.W = w; a.X = x; a.O = y; a.Res = nullptr; a.in_scale = in_scale; a.in_shift = in_shift; a.stats = stats;
    a.Cm = cout; a.P = n * ho * wo; a.in_relu = in_relu;
    fill_geom(a.g, n, h, wd, cin, ho, wo, r, s, stride, pad, 0);
    return launch_igemm_conv((hipStream_t)stream, a);
}

// y = relu?( conv(x) * out_scale[c] + out_shift[c] ): a convolution with the BatchNorm (+ ReLU) that follows it folded into the output stage
// (the fp32 accumulators are scaled, shifted, clamped and rounded to bf16 once): the inference forward of every bottleneck's conv1 / conv2
extern "C" int dali_conv2d_bn_act(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, uint16_t* y, int n, int h, int wd, int cin, int cout,
                                  int r, int s, int stride, int pad, const float* out_scale, const float* out_shift, int out_relu) {
    DALI_REQUIRE(ctx && x && w && y && out_scale && out_shift, "dali_conv2d_bn_act: null argument");
    DALI_REQUIRE(cin % 32 == 0 && cout % 8 == 0, "dali_conv2d_bn_act: cin must be a multiple of 32 and cout of 8 (cin=%d cout=%d)", cin, cout);
    DALI_REQUIRE(stride == 1 || stride == 2, "dali_conv2d_bn_act: stride %d unsupported", stride);
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (wd + 2 * pad - s) / stride + 1;
    IGemmArgs a{};
    a.W = w; a.X = x; a.O = y;
    a.out_scale = out_scale; a.out_shift = out_shift; a.out_relu = out_relu;
    a.Cm = cout; a.P = n * ho * wo;
    fill_geom(a.g, n, h, wd, cin, ho, wo, r, s, stride, pad, 0);
    return launch_igemm_conv((hipStream_t)stream, a);
}

// output width if the geometry is what the 3x3 halo kernels take (3x3, stride 1, pad 1, power-of-two pixel grid), else 0
static int halo_width(int r, int s, int stride, int pad, int ho, int wo) {
    return (r == 3 && s == 3 && stride == 1 && pad == 1 && ilog2_exact(wo) >= 0 && ilog2_exact(ho * wo) >= 0 && ho * wo >= 128) ? wo : 0;
}
extern "C" int dali_conv2d_stat_tiles(int cout, int cin, int r, int s, int stride, int pad, int n, int ho, int wo, int fused_operand) {
    // the fused-operand (register-staged) kernels always use 128-pixel tiles for cout > 64
    if (fused_operand && cout > 64) return (n * ho * wo + 127) / 128;
    (void)stride; (void)pad;
    return igemm_conv_stat_tiles(cout, n * ho * wo, r * s * cin);
}

extern "C" int dali_conv2d_dgrad(dali_ctx* ctx, void* stream, const uint16_t* dy, const uint16_t* wt, uint16_t* dx,
                                 const uint16_t* residual, const uint8_t* residual_mask, int n, int h, int wd, int cin, int cout, int r, int s,
                                 int stride, int pad) {
    DALI_REQUIRE(ctx && dy && wt && dx, "dali_conv2d_dgrad: null argument");
    DALI_REQUIRE(!residual_mask || (residual && cin % 8 == 0), "dali_conv2d_dgrad: residual_mask needs a residual and cin %% 8 == 0 (cin=%d)", cin);
    DALI_REQUIRE(cout % 32 == 0 && cin % 4 == 0, "dali_conv2d_dgrad: cout must be a multiple of 32 and cin of 4 (cin=%d cout=%d)", cin, cout);
    DALI_REQUIRE(stride == 1 || stride == 2, "dali_conv2d_dgrad: stride %d unsupported", stride);
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (wd + 2 * pad - s) / stride + 1;
    IGemmArgs a{};
    a.W = wt; a.X = dy; a.O = dx; a.Res = residual; a.res_mask = residual_mask; a.in_scale = nullptr; a.in_shift = nullptr; a.stats = nullptr;
    a.Cm = cin; a.P = n * h * wd; a.in_relu = 0;
    fill_geom(a.g, n, ho, wo, cout, h, wd, r, s, stride, pad, 1);
    return launch_igemm_conv((hipStream_t)stream, a);
}

extern "C" int dali_conv2d_wgrad(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* dy, float* dw,
                                 int n, int h, int wd, int cin, int cout, int r, int s, int stride, int pad,
                                 const float* in_scale, const float* in_shift, int in_relu, int accumulate) {
    DALI_REQUIRE(ctx && x && dy && dw, "dali_conv2d_wgrad: null argument");
    DALI_REQUIRE(cin % 8 == 0 && cout % 8 == 0, "dali_conv2d_wgrad: cin and cout must be multiples of 8 (cin=%d cout=%d)", cin, cout);
    DALI_REQUIRE(stride == 1 || stride == 2, "dali_conv2d_wgrad: stride %d unsupported", stride);
    DALI_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "dali_conv2d_wgrad: in_scale/in_shift must come together");
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (wd + 2 * pad - s) / stride + 1;
    WGradArgs a{};
    a.dY = dy; a.X = x; a.in_scale = in_scale; a.in_shift = in_shift; a.in_relu = in_relu;
    a.Cm = cout; a.P = n * ho * wo; a.Ntot = r * s * cin;
    fill_geom(a.g, n, h, wd, cin, ho, wo, r, s, stride, pad, 0);
    size_t ws_bytes;
    wgrad_plan(a.Cm, a.Ntot, a.P, 512, &a.splits, &a.pix_per_split, &ws_bytes, r * s,
               in_scale == nullptr ? halo_width(r, s, stride, pad, ho, wo) : 0);
    a.partial = static_cast<float*>(workspace(ctx, ws_bytes));
    if (!a.partial) return DALI_ERR_NOMEM;
    return launch_igemm_wgrad((hipStream_t)stream, a, dw, accumulate);
}

// ---- Linear layers (ViT: qkv / proj / fc1 / fc2 / patch embedding) on the same engine: a Linear is a 1x1 conv over tokens ----
// 1x1 convolution with the fused output stage (IGemmArgs::out_scale ...): y = gate(relu?(acc*out_scale + out_shift + bias + residual)),
// bits_out = (y > 0).  mode 0: forward (w [cout][cin]); mode 1: data gradient (w = the [cin][cout] image, x = dy).
extern "C" int dali_conv1x1_fused(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, uint16_t* y, int pixels, int cin, int cout,
                                  const float* out_scale, const float* out_shift, const float* bias, const uint16_t* residual, int out_relu,
                                  uint8_t* bits_out, const uint8_t* out_mask, const float* res_scale) {
    DALI_REQUIRE(ctx && x && w && y, "dali_conv1x1_fused: null argument");
    DALI_REQUIRE(cin % 32 == 0 && cout % 8 == 0 && pixels > 0, "dali_conv1x1_fused: cin %% 32, cout %% 8 (cin=%d cout=%d)", cin, cout);
    IGemmArgs a{};
    a.W = w; a.X = x; a.O = y; a.Res = residual; a.bias = bias;
    a.out_scale = out_scale; a.out_shift = out_shift; a.out_relu = out_relu; a.bits_out = bits_out; a.out_mask = out_mask; a.res_scale = res_scale;
    a.Cm = cout; a.P = pixels;
    fill_geom(a.g, 1, 1, pixels, cin, 1, pixels, 1, 1, 1, 0, 0);
    return launch_igemm_conv((hipStream_t)stream, a);
}

// y [pixels][cout] = [x1 | x2] @ w^T (+ bias): one GEMM whose K dimension is the concatenation of two plain [pixels][c] tensors (IGemmArgs::X2)
extern "C" int dali_conv1x1_cat(dali_ctx* ctx, void* stream, const uint16_t* x1, int c1, const uint16_t* x2, int c2, const uint16_t* w, const float* bias,
                                uint16_t* y, int pixels, int cout) {
    DALI_REQUIRE(ctx && x1 && x2 && w && y, "dali_conv1x1_cat: null argument");
    DALI_REQUIRE(c1 > 0 && c2 > 0 && c1 % 32 == 0 && c2 % 32 == 0 && cout % 8 == 0 && pixels > 0, "dali_conv1x1_cat: c1 %% 32, c2 %% 32, cout %% 8 (c1=%d c2=%d cout=%d)", c1, c2, cout);
    IGemmArgs a{};
    a.W = w; a.X = x1; a.X2 = x2; a.Ck1 = c1; a.O = y; a.bias = bias;
    a.Cm = cout; a.P = pixels;
    fill_geom(a.g, 1, 1, pixels, c1 + c2, 1, pixels, 1, 1, 1, 0, 0);
    return launch_igemm_conv((hipStream_t)stream, a);
}

// y [pixels][cout] = relu?( [x1 | x2] @ w^T * out_scale + out_shift ): the two-operand GEMM with the scale / shift / ReLU output stage
extern "C" int dali_conv1x1_cat_act(dali_ctx* ctx, void* stream, const uint16_t* x1, int c1, const uint16_t* x2, int c2, const uint16_t* w, int weight_parts,
                                    const float* out_scale, const float* out_shift, int out_relu, uint16_t* y, int pixels, int cout) {
    DALI_REQUIRE(ctx && x1 && x2 && w && y && out_shift, "dali_conv1x1_cat_act: null argument");
    DALI_REQUIRE(c1 > 0 && c2 > 0 && c1 % 64 == 0 && c2 % 64 == 0 && cout % 128 == 0 && pixels > 0, "dali_conv1x1_cat_act: c1 %% 64, c2 %% 64, cout %% 128 (c1=%d c2=%d cout=%d)", c1, c2, cout);
    IGemmArgs a{};
    DALI_REQUIRE(weight_parts == 1 || weight_parts == 2, "dali_conv1x1_cat_act: weight_parts must be 1 or 2");
    a.W = w; a.X = x1; a.X2 = x2; a.Ck1 = c1; a.x_rep = weight_parts; a.O = y;
    a.out_scale = out_scale; a.out_shift = out_shift; a.out_relu = out_relu;
    a.Cm = cout; a.P = pixels;
    fill_geom(a.g, 1, 1, pixels, weight_parts * (c1 + c2), 1, pixels, 1, 1, 1, 0, 0);
    return launch_igemm_conv((hipStream_t)stream, a);
}

static void linear_geom(GatherGeom& g, int K) {
    g.Hout = 1; g.Wout = 1; g.Hin = 1; g.Win = 1; g.Ck = K; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.mode = 0;
    g.pix_pitch = K; g.row_pitch = K; g.img_pitch = K; g.lw = g.lhw = -1;
}
namespace dali {
int launch_linear_fwd(hipStream_t st, const uint16_t* x, const uint16_t* w, const float* bias, int act, const uint16_t* residual, uint16_t* y,
                      uint16_t* pre, const uint16_t* dact_pre, int rows, int K, int N, const float* row_scale) {
    IGemmArgs a{};
    a.W = w; a.X = x; a.O = y; a.Res = residual; a.bias = bias; a.act = act; a.O2 = pre; a.dact_pre = dact_pre; a.row_scale = row_scale;
    a.Cm = N; a.P = rows;
    linear_geom(a.g, K);
    return launch_igemm_conv(st, a);
}
// dbias (nullable) = column sums of dy: they ride on the weight-gradient GEMM (per-split partial sums in cs_partial [splits][N], reduced
// like the slabs) where the kernel supports it; *bias_done says whether they did (otherwise the caller runs the column-sum pass)
int launch_linear_wgrad(hipStream_t st, const uint16_t* x, const uint16_t* dy, float* dw, int rows, int K, int N, float* slab, float* dbias,
                        float* cs_partial, bool* bias_done) {
    WGradArgs a{};
    a.dY = dy; a.X = x; a.partial = slab; a.Cm = N; a.P = rows; a.Ntot = K;
    linear_geom(a.g, K);
    size_t wsb;
    wgrad_plan(a.Cm, a.Ntot, a.P, 512, &a.splits, &a.pix_per_split, &wsb);
    const bool ride = dbias && cs_partial && wgrad_colsum_supported(N, K, 1, rows);
    if (bias_done) *bias_done = ride;
    if (ride) a.colsum = cs_partial;
    return launch_igemm_wgrad(st, a, dw, 0, ride ? dbias : nullptr, ride ? wgrad_colsum_rows(N, K, 1, rows, a.splits) : 0);
}
size_t linear_wgrad_colsum_floats(int rows, int K, int N) {
    int sp, pps; size_t wsb;
    wgrad_plan(N, K, rows, 512, &sp, &pps, &wsb);
    return (size_t)wgrad_colsum_rows(N, K, 1, rows, sp) * N;
}
size_t linear_wgrad_slab_bytes(int rows, int K, int N) {
    int sp, pps; size_t wsb;
    wgrad_plan(N, K, rows, 512, &sp, &pps, &wsb);
    return wsb;
}
}  // namespace dali

/* y[rows,N] = act(x[rows,K] @ w[N,K]^T + bias) (+ residual);  pre (nullable) receives the pre-activation. */
extern "C" int dali_linear_fwd(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, const float* bias, int act,
                               const uint16_t* residual, uint16_t* y, uint16_t* pre, int rows, int K, int N) {
    DALI_REQUIRE(ctx && x && w && y, "dali_linear_fwd: null argument");
    DALI_REQUIRE(K % 32 == 0 && N % 4 == 0 && rows > 0, "dali_linear_fwd: K must be a multiple of 32 and N of 4 (K=%d N=%d)", K, N);
    DALI_REQUIRE(act == 0 || act == 1, "dali_linear_fwd: act must be 0 (none) or 1 (gelu)");
    return launch_linear_fwd((hipStream_t)stream, x, w, bias, act, residual, y, pre, nullptr, rows, K, N);
}
/* the same with the whole branch output (bias and activation included) times a per-row factor before the residual is added: DropPath's
 * per-sample keep / (1 - p) factors expanded to rows (vit_pytorch.py:45-62, applied at :338) */
extern "C" int dali_linear_fwd_scaled(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, const float* bias, int act,
                                      const uint16_t* residual, const float* row_scale, uint16_t* y, int rows, int K, int N) {
    DALI_REQUIRE(ctx && x && w && y && row_scale, "dali_linear_fwd_scaled: null argument");
    DALI_REQUIRE(K % 32 == 0 && N % 4 == 0 && rows > 0, "dali_linear_fwd_scaled: K must be a multiple of 32 and N of 4 (K=%d N=%d)", K, N);
    DALI_REQUIRE(act == 0 || act == 1, "dali_linear_fwd_scaled: act must be 0 (none) or 1 (gelu)");
    return launch_linear_fwd((hipStream_t)stream, x, w, bias, act, residual, y, nullptr, nullptr, rows, K, N, row_scale);
}
/* dx[rows,K] = (dy[rows,N] @ wt[K,N]^T) * gelu'(gelu_pre) (+ residual);  wt is the transposed weight image [K][N]. */
extern "C" int dali_linear_dgrad(dali_ctx* ctx, void* stream, const uint16_t* dy, const uint16_t* wt, const uint16_t* gelu_pre,
                                 const uint16_t* residual, uint16_t* dx, int rows, int K, int N) {
    DALI_REQUIRE(ctx && dy && wt && dx, "dali_linear_dgrad: null argument");
    DALI_REQUIRE(N % 32 == 0 && K % 4 == 0 && rows > 0, "dali_linear_dgrad: N must be a multiple of 32 and K of 4 (K=%d N=%d)", K, N);
    return launch_linear_fwd((hipStream_t)stream, dy, wt, nullptr, 0, residual, dx, nullptr, gelu_pre, rows, N, K);
}
/* dw[N,K] (fp32) = dy^T @ x;  dbias[N] (nullable, fp32) = column sums of dy. */
extern "C" int dali_linear_wgrad(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, int rows,
                                 int K, int N) {
    DALI_REQUIRE(ctx && x && dy && dw, "dali_linear_wgrad: null argument");
    DALI_REQUIRE(K % 8 == 0 && N % 8 == 0 && rows > 0, "dali_linear_wgrad: K and N must be multiples of 8");
    const size_t slab = align_up(linear_wgrad_slab_bytes(rows, K, N), 256);
    const size_t part = align_up(std::max(colsum_partial_floats(rows, N), linear_wgrad_colsum_floats(rows, K, N)) * 4, 256);
    char* ws = static_cast<char*>(workspace(ctx, slab + part + reduce_scratch_bytes(N, 1)));
    if (!ws) return DALI_ERR_NOMEM;
    bool bias_done = false;
    int rc = launch_linear_wgrad((hipStream_t)stream, x, dy, dw, rows, K, N, reinterpret_cast<float*>(ws), dbias, reinterpret_cast<float*>(ws + slab), &bias_done);
    if (rc || !dbias || bias_done) return rc;
    return launch_colsum((hipStream_t)stream, dy, rows, N, dbias, reinterpret_cast<float*>(ws + slab), reinterpret_cast<double*>(ws + slab + part));
}
