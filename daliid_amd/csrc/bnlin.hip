// bnlin.hip -- training-mode BatchNorm behind a 1x1 convolution WITHOUT ever storing the convolution's output.
//
// Bottleneck tail of torchvision's ResNet-50 as Encoders.ResNet50ReID runs it (Encoders.py:330-339; block = conv1-bn1-relu,
// conv2-bn2-relu, conv3-bn3, + identity, relu):   raw3 = a2 W3^T  (1x1 conv, [P][w] x [C][w]^T),  y = relu(bn3(raw3) + x).
// raw3 is linear in a2, so everything BatchNorm needs of it follows from two small moments of a2:
//     m2[k] = sum_p a2[p,k]                      (column sums, [w])
//     G[k,k'] = sum_p a2[p,k] a2[p,k']           (Gram matrix, [w][w]; an MFMA weight-gradient GEMM with dY = X = a2: 1/4 of conv3's FLOPs)
// Forward:   mean[c] = W3[c,:] . m2 / P,   E[raw3^2][c] = W3[c,:] G W3[c,:]^T / P      -> scale, shift BEFORE conv3 runs;
//            conv3's epilogue then writes y = relu(scale*acc + shift + x) and the ReLU mask directly (IGemmArgs::out_scale ...).
// Backward:  with dz = dy * (y > 0),  G0 = dz^T a2 (the plain weight-gradient GEMM on dz),  s = colsum(dz):
//            sum_p dz*raw3 = rowdot(W3, G0)                         -> dgamma, dbeta and the folded coefficients
//            d_raw3 = A dz + Kc - Q raw3   (A = gamma*invstd, Q = A*invstd*dgamma/P, Kc = -A*dbeta/P + Q*mean; nnops.hip's convention)
//            dW3   = A.G0 + Kc (x) m2 - Q.(W3 G)                     (bnlin_row_kernel)
//            d_a2  = dz (A.W3) - a2 (W3^T diag(Q) W3) + W3^T Kc      (two data-gradient GEMMs: weights A.W3 with bias, then -M accumulated)
// Per block this removes, against the materialised form (raw3 stored; bn_act; bn_bwd reduce + apply): the raw3 write and its three
// re-reads, the dy / d_raw3 round trip of the apply pass -- about 6 of the 11 passes over [P][C] tensors a block made.
// The sums are fp32 MFMA accumulations reduced in a fixed order and finished in fp64: deterministic, and closer to the fp32
// reference than statistics of a bf16-rounded tensor.
#include "kernels.h"
#include "reduce_finish.h"
#include <algorithm>

namespace dali {

namespace {
constexpr int BL_CH = 8;          // channels per workgroup of the row kernel

__device__ __forceinline__ double block_sum_d(double v, double* red) {       // 256 threads; result in every thread
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float ld_f(const float* p) { return *p; }
__device__ __forceinline__ float ld_f(const uint16_t* p) { return bf16_bits_to_f32(*p); }
}  // namespace

// ------------------------------------------------------------------------------------------------
// The two small products of the scheme, W G ([C][w] x [w][w]) and W^T diag(Q) W ([w][C] x [C][w]): 0.5 GMAC each at layer4, exact fp32
// wanted (they carry batch statistics and BatchNorm's mean corrections).  v_mfma_f32_32x32x2_f32 = an fp32 fmaf chain at the fp32
// vector peak without any VALU work.  "NT" form, both operands with k CONTIGUOUS, so that a lane's share of a 16-deep k-step is one
// 16-byte load per operand (8 bf16; two loads for 8 fp32):
//     C[m][n] = sum_k sa[k] * A[m][k] * B[n][k]          A [M][lda] (fp32 or bf16), B [N][ldb] bf16, sa optional [K]
// (the first form read both operands k-major, "TN": one 2- or 4-byte load per lane and k, 32 lanes to a row -- two cache lines per wave
//  instruction with 64-128 useful bytes; at w = 512 either product took 36-39 us for 1.07 GFLOP, bound by the rate of those line requests.)
// One workgroup = one 32 x 32 output tile; its 4 waves split K four ways (k = 16*(4*it + wave) + ...) and are summed through LDS in a
// fixed order: no split-K slabs, 256 tiles x 4 waves fill the chip at w = 512.  Lane (i = lane & 31, h = lane >> 5) feeds MFMA t of a
// 16-deep k-step with k = k16 + 8h + t (any k order works as long as A and B agree).
// Epilogues: C as fp32 [M][ldc]; or bf16 of -C; optionally dot[m-tile][n] = sum_{m in tile} C[m][n] * B2[m][n] (the quadratic form
// w_c G w_c^T per output channel; B2 = B^T, [M][ldb2], read along n) and vsum[n] = sum_k v[k] B[n][k] on the m-tile-0 workgroups (W^T Kc).
// ------------------------------------------------------------------------------------------------
constexpr int BL_SAV_MAX = 2048;         // longest K whose sa / v vectors the deep-prefetch instantiations stage in LDS
// FIN (the forward's product): the LAST workgroup of a column block to arrive (one device-scope counter per block of 32 channels, the arrival
// pattern of reduce_finish.h: sc1 stores of the partials, every wave's vmcnt(0), barrier, one relaxed atomic) sums the column block's quadratic-
// form partials over the m tiles in a fixed order, forms mean[c] = W^T[.,c] . m2 / P and finishes scale / shift / mean / invstd (+ running
// statistics) for its 32 channels: what bnlin_finish_kernel did in a launch of its own (16 launches of 9 us per ResNet step).
struct BnlinFin {
    const float* m2; double count; const float* gamma; const float* beta; float* running_mean; float* running_var; float momentum, eps;
    float *scale, *shift, *mean_out, *invstd_out; unsigned int* ctr; int w;
    double* mu_part;          // [w/32][C]: per-tile shares of W^T m2, behind the quadratic-form partials in the `dot` scratch
};
template <class TA, bool HAS_SA, bool HAS_V, bool FIN = false, int D = 1, bool TN = false>
__global__ __launch_bounds__(256) void bnlin_nt_gemm_kernel(const TA* __restrict__ A, int lda, const uint16_t* __restrict__ B, int ldb, const float* __restrict__ sa,
                                                             int K, float* __restrict__ Cf, uint16_t* __restrict__ Cneg, int ldc,
                                                             const uint16_t* __restrict__ B2, int ldb2, float* __restrict__ dot, int ldd,
                                                             const float* __restrict__ v, float* __restrict__ vsum, BnlinFin fin = BnlinFin{}) {
    __shared__ float red[3][16][64];                    // partial accumulators of waves 1..3
    __shared__ float vred[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float vs = 0.f;
    const bool do_v = HAS_V && blockIdx.y == 0;
    const TA* ap = A + (size_t)(m0 + i) * lda + 8 * h;
    const uint16_t* bp = B + (size_t)(n0 + i) * ldb + 8 * h;
    // K is a multiple of 16: wave w runs the k-steps 16 (4 it + w).  D steps of loads are in flight ahead of the MFMAs, held as loaded (packed)
    // registers: one wave's MFMAs are ONE dependent chain (512 cycles per step) and a load takes 1-2 us, so with a single step ahead (and with
    // the TN form's two) every step waited for its operands -- 36-39 us per product at w = 512 against 7 us of MFMA time.  D > 1 needs
    // K % (64 D) == 0 and, for sa / v, K <= BL_SAV_MAX: they are staged in LDS once (one address per lane half: broadcast reads).
    struct Raw { uint4 a0, a1, b; };                    // a1 only for fp32 A
    __shared__ float lsav[(HAS_SA || HAS_V) && D > 1 ? 2 * BL_SAV_MAX : 1];
    if constexpr ((HAS_SA || HAS_V) && D > 1) {
        for (int k = threadIdx.x; k < K; k += 256) { lsav[k] = HAS_SA ? sa[k] : 1.f; lsav[BL_SAV_MAX + k] = HAS_V ? v[k] : 0.f; }
        __syncthreads();
    }
    auto unpack8 = [](const uint4 q, float (&o)[8]) {
        o[0] = bf16_bits_to_f32(q.x & 0xffffu); o[1] = bf16_bits_to_f32(q.x >> 16); o[2] = bf16_bits_to_f32(q.y & 0xffffu); o[3] = bf16_bits_to_f32(q.y >> 16);
        o[4] = bf16_bits_to_f32(q.z & 0xffffu); o[5] = bf16_bits_to_f32(q.z >> 16); o[6] = bf16_bits_to_f32(q.w & 0xffffu); o[7] = bf16_bits_to_f32(q.w >> 16);
    };
    auto load8f = [](const float* p, float (&o)[8]) {
        const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
        o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
    };
    auto load = [&](int k16, Raw& st) {
        if constexpr (sizeof(TA) == 4) {
            st.a0 = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(ap) + k16);
            st.a1 = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(ap) + k16 + 4);
        } else st.a0 = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(ap) + k16);
        st.b = *reinterpret_cast<const uint4*>(bp + k16);
    };
    auto fma = [&](const Raw& st, int k16) {
        float a[8], b[8], sc[8], vv[8];
        if constexpr (sizeof(TA) == 4) {
            a[0] = __uint_as_float(st.a0.x); a[1] = __uint_as_float(st.a0.y); a[2] = __uint_as_float(st.a0.z); a[3] = __uint_as_float(st.a0.w);
            a[4] = __uint_as_float(st.a1.x); a[5] = __uint_as_float(st.a1.y); a[6] = __uint_as_float(st.a1.z); a[7] = __uint_as_float(st.a1.w);
        } else unpack8(st.a0, a);
        unpack8(st.b, b);
        if constexpr (HAS_SA) load8f((D > 1 ? lsav : sa) + k16 + 8 * h, sc);
        if constexpr (HAS_V) load8f((D > 1 ? lsav + BL_SAV_MAX : v) + k16 + 8 * h, vv);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float av = HAS_SA ? a[t] * sc[t] : a[t];
            if (HAS_V) vs += vv[t] * b[t];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[t], acc, 0, 0, 0);
        }
    };
    int k16 = wave * 16;
    if constexpr (TN) {
        // "TN" operands, both k-major (A [K][lda], B [K][ldb]): a lane's share of an 8-deep k-step is 4 + 4 scalar loads (k = k8 + 2t + h), 32 lanes to a
        // row piece.  Kept for the forward's product G W^T: its A rows are fp32 (2 KB apart at w = 512) and the NT form's 16-byte pieces of 32 different
        // rows per instruction measured slower there (53 against 36 us at w = 512); D steps in flight as above (K % (32 D) == 0).
        static_assert(!HAS_SA && !HAS_V, "TN form: plain product only");
        const TA* apt = A + m0 + i;
        const uint16_t* bpt = B + n0 + i;
        struct RawT { float a[4]; uint32_t b[4]; };
        auto load_t = [&](int k8, RawT& st) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = k8 + 2 * t + h;
                st.a[t] = ld_f(apt + (size_t)k * lda);
                st.b[t] = bpt[(size_t)k * ldb];
            }
        };
        auto fma_t = [&](const RawT& st) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(st.a[t], bf16_bits_to_f32(st.b[t]), acc, 0, 0, 0);
        };
        int k8 = wave * 8;
        RawT ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) load_t(k8 + 32 * d, ring[d]);
        for (; k8 + 32 * D < K; k8 += 32 * D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                fma_t(ring[d]);
                load_t(k8 + 32 * (D + d), ring[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) fma_t(ring[d]);
    } else if constexpr (D > 1) {
        Raw ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) load(k16 + 64 * d, ring[d]);
        for (; k16 + 64 * D < K; k16 += 64 * D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                fma(ring[d], k16 + 64 * d);
                load(k16 + 64 * (D + d), ring[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) fma(ring[d], k16 + 64 * d);
    } else if (k16 < K) {
        Raw s0;
        load(k16, s0);
        for (; k16 + 64 < K; k16 += 64) {
            Raw s1;
            load(k16 + 64, s1);
            fma(s0, k16);
            s0 = s1;
        }
        fma(s0, k16);
    }
    if (!do_v) vs = 0.f;
    // C/D layout: column n = n0 + i, row m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    vred[wave][lane] = vs;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
        float d = 0.f;
        double dm = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + i;
            if (Cf) Cf[(size_t)m * ldc + n] = acc[r];
            if (Cneg) Cneg[(size_t)m * ldc + n] = f32_to_bf16_bits(-acc[r]);
            if (B2) {
                const float b2 = bf16_bits_to_f32(B2[(size_t)m * ldb2 + n]);
                d += acc[r] * b2;
                if constexpr (FIN) dm += (double)b2 * (double)fin.m2[m];         // this tile's share of W^T m2 (the channel means), B2 = W^T
            }
        }
        if (B2) {
            d += __shfl_xor(d, 32, 64);                 // the two row halves of the tile
            if constexpr (FIN) {
                dm += __shfl_xor(dm, 32, 64);
                if (h == 0) __hip_atomic_store(&fin.mu_part[(size_t)blockIdx.y * ldd + n0 + i], dm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (h == 0) {
                if constexpr (FIN) __hip_atomic_store(&dot[(size_t)blockIdx.y * ldd + n0 + i], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: seen by the finishing workgroup on any XCD
                else dot[(size_t)blockIdx.y * ldd + n0 + i] = d;
            }
        }
        if (do_v) {
            float t = ((vred[0][lane] + vred[1][lane]) + vred[2][lane]) + vred[3][lane];
            t += __shfl_xor(t, 32, 64);
            if (h == 0) vsum[n0 + i] = t;
        }
    }
    if constexpr (FIN) {
        __shared__ int s_last;
        __shared__ double fred[2][8][32];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's sc1 stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned prev = __hip_atomic_fetch_add(&fin.ctr[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = prev == gridDim.y - 1;
            if (s_last) __hip_atomic_store(&fin.ctr[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the slot's next launch
        }
        __syncthreads();
        if (!s_last) return;
        const int cx = threadIdx.x & 31, grp = threadIdx.x >> 5, c = n0 + cx;      // 32 channels x 8 groups over k / over the m tiles, fixed order
        double mu = 0.0, q = 0.0;
        // (the means W^T m2 were a loop over k here, 64 dependent rounds of loads at w = 512: 14 of the launch's 36 us; every tile now leaves its share
        //  beside its quadratic-form partial, from the B2 values its epilogue loads anyway)
        for (int t = grp; t < (int)gridDim.y; t += 8) {
            q += (double)__hip_atomic_load(&dot[(size_t)t * ldd + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mu += __hip_atomic_load(&fin.mu_part[(size_t)t * ldd + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        fred[0][grp][cx] = mu; fred[1][grp][cx] = q;
        __syncthreads();
        if (grp == 0) {
            mu = ((fred[0][0][cx] + fred[0][1][cx]) + (fred[0][2][cx] + fred[0][3][cx])) + ((fred[0][4][cx] + fred[0][5][cx]) + (fred[0][6][cx] + fred[0][7][cx]));
            q = ((fred[1][0][cx] + fred[1][1][cx]) + (fred[1][2][cx] + fred[1][3][cx])) + ((fred[1][4][cx] + fred[1][5][cx]) + (fred[1][6][cx] + fred[1][7][cx]));
            const double mean = mu / fin.count;
            double var = q / fin.count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
            const float sc = fin.gamma[c] * invstd;
            fin.scale[c] = sc;
            fin.shift[c] = fin.beta[c] - (float)mean * sc;
            fin.mean_out[c] = (float)mean;
            fin.invstd_out[c] = invstd;
            if (fin.running_mean) {                              // torch's update rule, as bn_finalize_kernel (nnops.hip)
                const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
                fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)mean;
                fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
            }
        }
    }
}

// ---- forward: scale / shift / mean / invstd (+ running statistics) from the per-tile quadratic-form partials and W^T m2 ----
//     E[raw^2][c] = sum_tiles dot[tile][c] / P,   mean[c] = sum_k m2[k] W^T[k][c] / P        (block = 64 channels x 4 k groups)
__global__ __launch_bounds__(256) void bnlin_finish_kernel(const float* __restrict__ dot, int tiles, const uint16_t* __restrict__ Wt, const float* __restrict__ m2,
                                                            int C, int w, double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                                            float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                                            float* __restrict__ invstd_out) {
    __shared__ double red[2][4][64];
    const int cx = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double mu = 0.0, q = 0.0;
    if (c < C) {
        double part[4] = {0.0, 0.0, 0.0, 0.0};
        int k = grp;
        for (; k + 12 < w; k += 16) {
            float wv[4], mv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { wv[u] = bf16_bits_to_f32(Wt[(size_t)(k + 4 * u) * C + c]); mv[u] = m2[k + 4 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) part[u] += (double)wv[u] * (double)mv[u];
        }
        for (; k < w; k += 4) part[0] += (double)bf16_bits_to_f32(Wt[(size_t)k * C + c]) * (double)m2[k];
        mu = (part[0] + part[1]) + (part[2] + part[3]);
        for (int t = grp; t < tiles; t += 4) q += (double)dot[(size_t)t * C + c];
    }
    red[0][grp][cx] = mu; red[1][grp][cx] = q;
    __syncthreads();
    if (grp == 0 && c < C) {
        mu = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
        q = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
        const double mean = mu / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        mean_out[c] = (float)mean;
        invstd_out[c] = invstd;
        if (running_mean) {                              // torch's update rule, as bn_finalize_kernel (nnops.hip)
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// ---- backward, per group of BL_CH output channels: G0 = dz^T a2 (in dW, reduced) -> dgamma, dbeta, folded coefficients, dW in place, A.W ----
// Ut = (W G)^T [w][C] was left by the forward's TN product; s_dz = colsum(dz).
__global__ __launch_bounds__(256) void bnlin_row_kernel(const uint16_t* __restrict__ W, const float* __restrict__ Ut, const float* __restrict__ m2,
                                                         const float* __restrict__ s_dz, int C, int w, double count, const float* __restrict__ scale,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ dW,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, uint16_t* __restrict__ wd1, int ld1,
                                                         float* __restrict__ qk) {
    extern __shared__ __attribute__((aligned(16))) float bl_smem[];
    float* wf = bl_smem;                                    // [BL_CH][w]
    float* g0 = wf + BL_CH * w;                             // [BL_CH][w]
    float* us = g0 + BL_CH * w;                             // [w][BL_CH]: this group's columns of Ut
    float* m2s = us + BL_CH * w;                            // [w]
    __shared__ double red[4][BL_CH];
    __shared__ float coef[BL_CH][4];                        // A, Kc, Q, pad
    const int c0 = blockIdx.x * BL_CH;                      // C % BL_CH == 0 (checked by the launcher)
    // Every global read of the kernel is requested here, in whole 16-byte pieces and unrolled (the first form's element loops had run-time trip
    // counts: one load round trip per iteration, 16 + 16 of them at w = 512 -- 12 us per launch for 50 KB of data).  The 8 rows of G0 and of W are
    // contiguous; Ut's 8 columns are 32-byte pieces of its rows.  w % 32 == 0.
    {
        const float4* gsrc = reinterpret_cast<const float4*>(dW + (size_t)c0 * w);
        const uint4* wsrc = reinterpret_cast<const uint4*>(W + (size_t)c0 * w);
        const int n4 = BL_CH * w / 4, n8 = BL_CH * w / 8;
#pragma unroll 4
        for (int e = threadIdx.x; e < n4; e += 256) reinterpret_cast<float4*>(g0)[e] = gsrc[e];
#pragma unroll 2
        for (int e = threadIdx.x; e < n8; e += 256) {
            const uint4 q = wsrc[e];
            float4 lo, hi;
            lo.x = bf16_bits_to_f32(q.x & 0xffffu); lo.y = bf16_bits_to_f32(q.x >> 16); lo.z = bf16_bits_to_f32(q.y & 0xffffu); lo.w = bf16_bits_to_f32(q.y >> 16);
            hi.x = bf16_bits_to_f32(q.z & 0xffffu); hi.y = bf16_bits_to_f32(q.z >> 16); hi.z = bf16_bits_to_f32(q.w & 0xffffu); hi.w = bf16_bits_to_f32(q.w >> 16);
            reinterpret_cast<float4*>(wf)[2 * e] = lo; reinterpret_cast<float4*>(wf)[2 * e + 1] = hi;
        }
#pragma unroll 4
        for (int e = threadIdx.x; e < 2 * w; e += 256)      // float4 e: row k = e >> 1, half e & 1 of its 8 columns
            reinterpret_cast<float4*>(us)[e] = *reinterpret_cast<const float4*>(Ut + (size_t)(e >> 1) * C + c0 + 4 * (e & 1));
        for (int e = threadIdx.x; e < w / 4; e += 256) reinterpret_cast<float4*>(m2s)[e] = reinterpret_cast<const float4*>(m2)[e];
    }
    __syncthreads();
    {   // T[ch] = sum_p dz * raw3 = rowdot(W, G0): the 8 channels' partial sums per thread, ONE reduction (8 block reductions in a row before)
        double t[BL_CH];
#pragma unroll
        for (int ch = 0; ch < BL_CH; ++ch) {
            t[ch] = 0.0;
            for (int k = threadIdx.x; k < w; k += 256) t[ch] += (double)wf[ch * w + k] * (double)g0[ch * w + k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t[ch] += __shfl_xor(t[ch], o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int ch = 0; ch < BL_CH; ++ch) red[threadIdx.x >> 6][ch] = t[ch];
        }
        __syncthreads();
        if (threadIdx.x < BL_CH) {
            const int ch = threadIdx.x, c = c0 + ch;
            const double tt = (red[0][ch] + red[1][ch]) + (red[2][ch] + red[3][ch]);
            const double s = (double)s_dz[c], iv = (double)invstd[c], mn = (double)mean[c], a = (double)scale[c];
            const double dg = iv * (tt - mn * s);                   // sum dz * xhat
            const double qq = a * iv * dg / count;
            dgamma[c] = (float)dg;
            dbeta[c] = (float)s;
            qk[c] = (float)qq; qk[C + c] = (float)(qq * mn - a * s / count);
            coef[ch][0] = (float)a; coef[ch][1] = (float)(qq * mn - a * s / count); coef[ch][2] = (float)qq;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < BL_CH * w; e += 256) {    // dW[c][k] = A G0 + Kc m2 - Q (W G)   (k fastest: whole rows of dW per channel)
        const int ch = e / w, k = e - ch * w;
        dW[(size_t)(c0 + ch) * w + k] = coef[ch][0] * g0[e] + coef[ch][1] * m2s[k] - coef[ch][2] * us[k * BL_CH + ch];
    }
    for (int k = threadIdx.x; k < w; k += 256) {            // A.W, transposed into the data-gradient image [w][C]: 8 channels = 16 bytes per k
        uint32_t o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = pack_bf16x2(coef[2 * t][0] * wf[(2 * t) * w + k], coef[2 * t + 1][0] * wf[(2 * t + 1) * w + k]);
        *reinterpret_cast<uint4*>(wd1 + (size_t)k * ld1 + c0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
// k-steps of loads in flight per wave: as many as K allows (K % (64 D) == 0), at most 8 (DALI_BNLIN_DEPTH caps it: A/B)
static int bnlin_depth_tn(int K) {         // TN form: 8-deep steps, K % (32 D) == 0; D = 2 is the round-4 kernel
    const int cap = DALI_ENV_INT("DALI_BNLIN_DEPTH_TN", 4);
    int d = 1;
    while (d < 8 && 2 * d <= cap && K % (64 * d) == 0) d *= 2;
    return d;
}
static int bnlin_depth(int K) {
    const int cap = DALI_ENV_INT("DALI_BNLIN_DEPTH", 8);
    int d = 1;
    while (d < 8 && 2 * d <= cap && K % (128 * d) == 0) d *= 2;
    return d;
}
// Wt = W^T [w][C] bf16 (the plain data-gradient image of the convolution); ut [w][C] fp32 and dot ((w/32) * C * 12 bytes: per-tile fp32 quadratic-form and fp64 mean partials) are outputs the
// backward / the finish kernel read
int launch_bnlin_stats(hipStream_t st, const uint16_t* W, const uint16_t* Wt, const float* gram, const float* m2, int C, int w, double count, const float* gamma,
                       const float* beta, float* rm, float* rv, float momentum, float eps, float* ut, float* dot, float* scale, float* shift,
                       float* mean, float* invstd) {
    if (w % 32 != 0 || C % 32 != 0) { set_error("bnlin: width %d and channels %d must be multiples of 32", w, C); return DALI_ERR_INVALID; }
    // Ut[k'][c] = sum_k G[k'][k] W[c][k]  (G symmetric); dot[tile][c] = sum_{k' in tile} Ut[k'][c] Wt[k'][c]
    if (C / 32 <= RF_GROUPS && DALI_ENV_INT("DALI_BNLIN_FUSED_FINISH", 1) != 0) {
        // the statistics finish in the product's own launch, by the last workgroup of every 32-channel column block (see BnlinFin)
        unsigned int* ctr = nullptr;
        if (int rc = rf_counter_base(&ctr)) return rc;
        ctr += (size_t)rf_next_slot() * RF_GROUPS;
        const BnlinFin fin{m2, count, gamma, beta, rm, rv, momentum, eps, scale, shift, mean, invstd, ctr, w, reinterpret_cast<double*>(dot + (size_t)(w / 32) * C)};
#define BL_FWD_FIN(DEPTH, TNF) hipLaunchKernelGGL((bnlin_nt_gemm_kernel<float, false, false, true, DEPTH, TNF>), dim3(C / 32, w / 32), dim3(256), 0, st, gram, w, TNF ? Wt : W, TNF ? C : w, \
                                                  (const float*)nullptr, w, ut, (uint16_t*)nullptr, C, Wt, C, dot, C, (const float*)nullptr, (float*)nullptr, fin)
        if (DALI_ENV_INT("DALI_BNLIN_FWD_NT", 0)) {
            switch (bnlin_depth(w)) { case 8: BL_FWD_FIN(8, false); break; case 4: BL_FWD_FIN(4, false); break; case 2: BL_FWD_FIN(2, false); break; default: BL_FWD_FIN(1, false); }
        } else {
            switch (bnlin_depth_tn(w)) { case 8: BL_FWD_FIN(8, true); break; case 4: BL_FWD_FIN(4, true); break; case 2: BL_FWD_FIN(2, true); break; default: BL_FWD_FIN(1, true); }
        }
#undef BL_FWD_FIN
        DALI_LAUNCH_CHECK();
        return DALI_OK;
    }
    hipLaunchKernelGGL((bnlin_nt_gemm_kernel<float, false, false>), dim3(C / 32, w / 32), dim3(256), 0, st, gram, w, W, w, (const float*)nullptr, w, ut, (uint16_t*)nullptr, C,
                       Wt, C, dot, C, (const float*)nullptr, (float*)nullptr);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(bnlin_finish_kernel, dim3((C + 63) / 64), dim3(256), 0, st, dot, w / 32, Wt, m2, C, w, count, gamma, beta, rm, rv, momentum, eps,
                       scale, shift, mean, invstd);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

// only Ut = (W G)^T [w][C] (a scheme whose forward statistics come from elsewhere: the downsample branch, resnet_plan.hip)
int launch_bnlin_ut(hipStream_t st, const uint16_t* W, const float* gram, int C, int w, float* ut) {
    if (w % 32 != 0 || C % 32 != 0) { set_error("bnlin: width %d and channels %d must be multiples of 32", w, C); return DALI_ERR_INVALID; }
    hipLaunchKernelGGL((bnlin_nt_gemm_kernel<float, false, false>), dim3(C / 32, w / 32), dim3(256), 0, st, gram, w, W, w, (const float*)nullptr, w, ut, (uint16_t*)nullptr, C,
                       (const uint16_t*)nullptr, 0, (float*)nullptr, 0, (const float*)nullptr, (float*)nullptr);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

int launch_bnlin_bwd(hipStream_t st, const uint16_t* W, const uint16_t* Wt, const float* ut, const float* m2, const float* s_dz, int C,
                     int w, double count, const float* scale, const float* mean, const float* invstd, float* dW, float* dgamma, float* dbeta,
                     uint16_t* wd1, uint16_t* wd2, float* bvec, float* qk, int ld1, int ld2) {
    if (w % 32 != 0 || C % 32 != 0) { set_error("bnlin: width %d and channels %d must be multiples of 32", w, C); return DALI_ERR_INVALID; }
    if (ld1 <= 0) ld1 = C;
    if (ld2 <= 0) ld2 = w;
    if ((size_t)(3 * BL_CH + 1) * w * sizeof(float) > 64 * 1024) { set_error("bnlin: width %d: the row kernel keeps 25 w floats in LDS (w <= 640)", w); return DALI_ERR_LIMIT; }
    hipLaunchKernelGGL(bnlin_row_kernel, dim3(C / BL_CH), dim3(256), (size_t)(3 * BL_CH + 1) * w * sizeof(float), st, W, ut, m2, s_dz, C, w,
                       count, scale, mean, invstd, dW, dgamma, dbeta, wd1, ld1, qk);
    DALI_LAUNCH_CHECK();
    // wd2 = -(W^T diag(Q) W) [w][w] (bf16), bvec = W^T Kc: A = B = Wt [w][K = C], scaled by Q along K; v = Kc
#define BL_BWD(DEPTH) hipLaunchKernelGGL((bnlin_nt_gemm_kernel<uint16_t, true, true, false, DEPTH>), dim3(w / 32, w / 32), dim3(256), 0, st, Wt, C, Wt, C, qk, C, (float*)nullptr, wd2, ld2, \
                                         (const uint16_t*)nullptr, 0, (float*)nullptr, 0, qk + C, bvec)
    switch (C <= BL_SAV_MAX ? bnlin_depth(C) : 1) { case 8: BL_BWD(8); break; case 4: BL_BWD(4); break; case 2: BL_BWD(2); break; default: BL_BWD(1); }
#undef BL_BWD
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

}  // namespace dali

// ---- single-op C ABI (parity tests; the net plan calls the launchers with its own buffers) ---------------------------------
using namespace dali;

static GatherGeom bl_geom(int P, int Ck) {
    GatherGeom g{};
    g.Hout = 1; g.Wout = P; g.Hin = 1; g.Win = P; g.Ck = Ck; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.mode = 0;
    g.pix_pitch = Ck; g.row_pitch = P * Ck; g.img_pitch = (long long)P * Ck; g.lw = g.lhw = -1;
    return g;
}

extern "C" int dali_bnlin_fwd(dali_ctx* ctx, void* stream, const uint16_t* a, const uint16_t* W, int P, int C, int w, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* gram, float* m2,
                              float* ut, float* scale, float* shift, float* mean, float* invstd) {
    DALI_REQUIRE(ctx && a && W && gamma && beta && gram && m2 && ut && scale && shift && mean && invstd, "dali_bnlin_fwd: null argument");
    DALI_REQUIRE(P > 0 && C % 32 == 0 && w % 32 == 0, "dali_bnlin_fwd: C %% 32, w %% 32 (C=%d w=%d)", C, w);
    hipStream_t st = (hipStream_t)stream;
    WGradArgs wa{};
    wa.dY = a; wa.X = a; wa.Cm = w; wa.P = P; wa.Ntot = w; wa.g = bl_geom(P, w);
    size_t wsb;
    wgrad_plan(w, w, P, 512, &wa.splits, &wa.pix_per_split, &wsb, 1, 0);
    const bool fused_cs = wgrad_colsum_supported(w, w, 1, P);
    const int cs_rows = wgrad_colsum_rows(w, w, 1, P, wa.splits);             // one row per split, or per (split, n tile) from the 128 x 256 kernels
    const size_t b_cs = align_up(std::max(colsum_partial_floats(P, w), (size_t)cs_rows * w) * 4, 256), b_sc = align_up(reduce_scratch_bytes(w, 1), 256);
    const size_t b_wt = align_up((size_t)C * w * 2, 256), b_dot = align_up((size_t)(w / 32) * C * 12, 256);      // quadratic-form partials (fp32) + mean partials (fp64)
    char* ws = static_cast<char*>(workspace(ctx, align_up(wsb, 256) + b_cs + b_sc + b_wt + b_dot));
    if (!ws) return DALI_ERR_NOMEM;
    wa.partial = reinterpret_cast<float*>(ws);
    char* p = ws + align_up(wsb, 256);
    float* cs_partial = reinterpret_cast<float*>(p); p += b_cs;
    double* scratch = reinterpret_cast<double*>(p); p += b_sc;
    uint16_t* wt = reinterpret_cast<uint16_t*>(p); p += b_wt;
    float* dot = reinterpret_cast<float*>(p);
    int rc;
    if (fused_cs) wa.colsum = cs_partial;
    if ((rc = launch_igemm_wgrad(st, wa, gram, 0, fused_cs ? m2 : nullptr, cs_rows))) return rc;
    if (!fused_cs && (rc = launch_colsum(st, a, P, w, m2, cs_partial, scratch))) return rc;
    if ((rc = launch_weight_transpose(st, W, C, 1, w, wt))) return rc;
    return launch_bnlin_stats(st, W, wt, gram, m2, C, w, (double)P, gamma, beta, running_mean, running_var, momentum, eps, ut, dot, scale, shift, mean, invstd);
}

extern "C" int dali_bnlin_bwd(dali_ctx* ctx, void* stream, const uint16_t* dz, const uint16_t* a, const uint16_t* W, int P, int C, int w,
                              const float* ut, const float* m2, const float* scale, const float* mean, const float* invstd, float* dW,
                              float* dgamma, float* dbeta, uint16_t* wd1, uint16_t* wd2, float* bvec) {
    DALI_REQUIRE(ctx && dz && a && W && ut && m2 && scale && mean && invstd && dW && dgamma && dbeta && wd1 && wd2 && bvec, "dali_bnlin_bwd: null argument");
    DALI_REQUIRE(P > 0 && C % 32 == 0 && w % 32 == 0, "dali_bnlin_bwd: C %% 32, w %% 32 (C=%d w=%d)", C, w);
    hipStream_t st = (hipStream_t)stream;
    WGradArgs wa{};
    wa.dY = dz; wa.X = a; wa.Cm = C; wa.P = P; wa.Ntot = w; wa.g = bl_geom(P, w);
    size_t wsb;
    wgrad_plan(C, w, P, 512, &wa.splits, &wa.pix_per_split, &wsb, 1, 0);
    const bool fused_cs = wgrad_colsum_supported(C, w, 1, P);
    const int cs_rows = wgrad_colsum_rows(C, w, 1, P, wa.splits);
    const size_t b_cs = align_up(std::max(colsum_partial_floats(P, C), (size_t)cs_rows * C) * 4, 256), b_sc = align_up(reduce_scratch_bytes(C, 1), 256);
    const size_t b_v = align_up((size_t)C * 4, 256), b_wt = align_up((size_t)C * w * 2, 256);
    char* ws = static_cast<char*>(workspace(ctx, align_up(wsb, 256) + b_cs + b_sc + 3 * b_v + b_wt));
    if (!ws) return DALI_ERR_NOMEM;
    wa.partial = reinterpret_cast<float*>(ws);
    char* p = ws + align_up(wsb, 256);
    float* cs_partial = reinterpret_cast<float*>(p); p += b_cs;
    double* scratch = reinterpret_cast<double*>(p); p += b_sc;
    float* sdz = reinterpret_cast<float*>(p); p += b_v;
    float* qk = reinterpret_cast<float*>(p); p += 2 * b_v;
    uint16_t* wt = reinterpret_cast<uint16_t*>(p);
    int rc;
    if (fused_cs) wa.colsum = cs_partial;
    if ((rc = launch_weight_transpose(st, W, C, 1, w, wt))) return rc;
    if ((rc = launch_igemm_wgrad(st, wa, dW, 0, fused_cs ? sdz : nullptr, cs_rows))) return rc;                       // G0 -> dW
    if (!fused_cs && (rc = launch_colsum(st, dz, P, C, sdz, cs_partial, scratch))) return rc;
    return launch_bnlin_bwd(st, W, wt, ut, m2, sdz, C, w, (double)P, scale, mean, invstd, dW, dgamma, dbeta, wd1, wd2, bvec, qk);
}
