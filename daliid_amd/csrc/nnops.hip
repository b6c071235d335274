// nnops.hip -- the HBM-bound kernels around the convolutions of the ResNet-50-ReID trunk:
// BatchNorm statistics/finalise/apply/backward, stem image/weight packing, max-pool (with the stem BN fused,
// Encoders.py:333-335: conv1 -> bn1 -> maxpool, NO ReLU), global avg+max pool head (Encoders.py:341-345),
// BatchNorm1d neck (Encoders.py:350), weight casts/transposes.  All activations NHWC bf16, math fp32.
#include "kernels.h"
#include "reduce_finish.h"

namespace dali {

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = bf16_bits_to_f32(v.x & 0xffffu); f[1] = bf16_bits_to_f32(v.x >> 16);
    f[2] = bf16_bits_to_f32(v.y & 0xffffu); f[3] = bf16_bits_to_f32(v.y >> 16);
    f[4] = bf16_bits_to_f32(v.z & 0xffffu); f[5] = bf16_bits_to_f32(v.z >> 16);
    f[6] = bf16_bits_to_f32(v.w & 0xffffu); f[7] = bf16_bits_to_f32(v.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
__device__ __forceinline__ void load8f(const float* __restrict__ p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm finalise: partial (sum, sumsq) [tiles][C][2] -> mean, invstd, scale = gamma*invstd,
// shift = beta - mean*scale; running stats updated as torch does (momentum, unbiased variance).
// fp64 two-level sum and the finish in one launch (reduce_finish.h).
// ------------------------------------------------------------------------------------------------
// second level of reduce_finish_kernel for the forward statistics (v = {sum, sumsq} of channel c)
struct FinBnFwd {
    double count;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    float momentum, eps;
    float *scale, *shift, *mean_out, *invstd_out;
    __device__ void operator()(int c, const double* v) const {
        const double mean = v[0] / count;
        double var = v[1] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        mean_out[c] = (float)mean;
        invstd_out[c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
};

// eval mode: scale/shift from the running statistics
__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// the same for many BatchNorms in one launch (the inference forward folds all of them into conv epilogues: 53 coefficient launches -> 2)
__global__ void bn_eval_coeffs_batched_kernel(BnEvalJobs jobs, float eps) {
    const BnEvalJob j = jobs.job[blockIdx.y];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < j.C; c += gridDim.x * blockDim.x) {
        const float sc = j.gamma[c] / sqrtf(j.rv[c] + eps);
        j.scale[c] = sc;
        j.shift[c] = j.beta[c] - j.rm[c] * sc;
    }
}

// inference: conv3 and a stride-1 downsample convolution of a bottleneck as ONE GEMM over [a2 | x]: the weight image [C][w + cin] with each
// BatchNorm's scale folded into its rows (one rounding, from the fp32 master weights) and the summed shifts
__global__ void fold_cat_weights_kernel(const float* __restrict__ w3, const float* __restrict__ wd, const float* __restrict__ s3, const float* __restrict__ sd,
                                        const float* __restrict__ h3, const float* __restrict__ hd, int C, int w, int cin, int parts,
                                        uint16_t* __restrict__ wcat, float* __restrict__ shcat) {
    // parts == 2: [s3.W3 hi | s3.W3 lo | sd.Wd hi | sd.Wd lo], hi = bf16(v), lo = bf16(v - hi): the folded fp32 weights to ~2^-17
    const int K = parts * (w + cin);
    const size_t n = (size_t)C * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / K), k = (int)(e - (size_t)c * K);
        const bool first = k < parts * w;
        const int kk = first ? k : k - parts * w, width = first ? w : cin, part = kk / width, col = kk - part * width;
        const float v = first ? s3[c] * w3[(size_t)c * w + col] : sd[c] * wd[(size_t)c * cin + col];
        const uint16_t hi = f32_to_bf16_bits(v);
        wcat[e] = part == 0 ? hi : f32_to_bf16_bits(v - bf16_bits_to_f32(hi));
        if (k == 0) shcat[c] = h3[c] + hd[c];
    }
}

// ------------------------------------------------------------------------------------------------
// Block output: y = relu( raw*scale+shift + identity ), identity = idn (bf16) or raw2*scale2+shift2.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_act_kernel(const uint16_t* __restrict__ raw, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const uint16_t* __restrict__ idn,
                                                      const uint16_t* __restrict__ raw2, const float* __restrict__ scale2,
                                                      const float* __restrict__ shift2, int relu, size_t chunks, int C,
                                                      uint16_t* __restrict__ y, uint8_t* __restrict__ mask_out) {
    // A thread's channel chunk is the same in every grid-stride iteration when the stride is a multiple of C (always, for the
    // power-of-two widths of the net): the coefficients are then loaded once, not per 16 bytes of data, and the 64-bit modulo
    // leaves the loop.
    const size_t first = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool fixed_c = ((size_t)gridDim.x * 2048) % (size_t)C == 0;
    int c = (int)((first * 8) % (size_t)C);
    float sc[8], sh[8], sc2[8], sh2[8];
    load8f(scale + c, sc); load8f(shift + c, sh);
    if (raw2) { load8f(scale2 + c, sc2); load8f(shift2 + c, sh2); }
    for (size_t i = first; i < chunks; i += (size_t)gridDim.x * 256) {
        if (!fixed_c) {
            c = (int)((i * 8) % (size_t)C);
            load8f(scale + c, sc); load8f(shift + c, sh);
            if (raw2) { load8f(scale2 + c, sc2); load8f(shift2 + c, sh2); }
        }
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(raw + i * 8), v);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = v[t] * sc[t] + sh[t];
        if (idn) {
            float r[8];
            unpack8(*reinterpret_cast<const uint4*>(idn + i * 8), r);
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] += r[t];
        } else if (raw2) {
            float r[8];
            unpack8(*reinterpret_cast<const uint4*>(raw2 + i * 8), r);
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] += r[t] * sc2[t] + sh2[t];
        }
        if (relu) {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        *reinterpret_cast<uint4*>(y + i * 8) = pack8(v);
        if (mask_out) {                              // bit t = (y[c+t] > 0): what the backward needs of y, at 1/16 of its bytes
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) m |= (v[t] > 0.f ? 1u : 0u) << t;
            mask_out[i] = (uint8_t)m;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward, two passes over [P][C] bf16 tensors.
//   dz = g * mask,  mask = (ymask > 0) if ymask given else (raw*scale+shift > 0) if relu else 1
//   reduce: S1[c] = sum dz, S2[c] = sum dz * xhat          (xhat = (raw-mean)*invstd)
//   apply : draw = scale * (dz - S1/N - xhat * S2/N)
// A second BN sharing the same dz (the downsample branch of a bottleneck) is handled in the same passes.
// Work split: a thread owns one 16-byte channel chunk and strides over pixels; partials per block.
// ------------------------------------------------------------------------------------------------
template <bool DUAL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ ymask,
                                                             const uint8_t* __restrict__ ybits, BnBwdSide a, BnBwdSide b, int relu, int P, int C,
                                                             int rows_per_block, float* __restrict__ partial) {
    extern __shared__ float red[];                  // [rows_in_flight][C][NV]
    constexpr int NV = DUAL ? 3 : 2;                // S1, S2a, (S2b)
    const int cpr = C >> 3;                         // chunks per row
    const int rif = 256 / cpr;                      // rows in flight (C <= 2048)
    const int col = threadIdx.x % cpr, rsub = threadIdx.x / cpr;
    const int c = col * 8;
    const int p0 = blockIdx.x * rows_per_block;
    const int p1 = min(P, p0 + rows_per_block);
    float s1[8], s2a[8], s2b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { s1[t] = 0.f; s2a[t] = 0.f; s2b[t] = 0.f; }
    if (rsub < rif) {
        float ma[8], ia[8], sa[8], ha[8], mb[8], ib[8];
        load8f(a.mean + c, ma); load8f(a.invstd + c, ia);
        if (!ymask && !ybits && relu) { load8f(a.scale + c, sa); load8f(a.shift + c, ha); }
        if (DUAL) { load8f(b.mean + c, mb); load8f(b.invstd + c, ib); }
        // one row's contribution, added in row order whatever the unrolling (the sums stay bit-identical)
        auto add_row = [&](const uint4 gq, const uint4 rq, const unsigned m, const uint4 yq, const uint4 r2q) {
            float gv[8], rv[8];
            unpack8(gq, gv);
            unpack8(rq, rv);
            if (ybits) {
#pragma unroll
                for (int t = 0; t < 8; ++t) gv[t] = ((m >> t) & 1u) ? gv[t] : 0.f;
            } else if (ymask) {
                float yv[8];
                unpack8(yq, yv);
#pragma unroll
                for (int t = 0; t < 8; ++t) gv[t] = yv[t] > 0.f ? gv[t] : 0.f;
            } else if (relu) {
#pragma unroll
                for (int t = 0; t < 8; ++t) gv[t] = (rv[t] * sa[t] + ha[t]) > 0.f ? gv[t] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) { s1[t] += gv[t]; s2a[t] += gv[t] * ((rv[t] - ma[t]) * ia[t]); }
            if (DUAL) {
                float r2[8];
                unpack8(r2q, r2);
#pragma unroll
                for (int t = 0; t < 8; ++t) s2b[t] += gv[t] * ((r2[t] - mb[t]) * ib[t]);
            }
        };
        // four rows' loads are issued before the first is used: with 16 waves per CU and two 16-byte loads per thread in flight
        // the pass was latency-bound at 4.1 TB/s
        constexpr int U = 4;
        int p = p0 + rsub;
        for (; p + (U - 1) * rif < p1; p += U * rif) {
            uint4 gq[U], rq[U], yq[U], r2q[U];
            unsigned m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t o = (size_t)(p + u * rif) * C + c;
                gq[u] = *reinterpret_cast<const uint4*>(g + o);
                rq[u] = *reinterpret_cast<const uint4*>(a.raw + o);
                m[u] = ybits ? ybits[o >> 3] : 0u;
                yq[u] = (!ybits && ymask) ? *reinterpret_cast<const uint4*>(ymask + o) : make_uint4(0, 0, 0, 0);
                r2q[u] = DUAL ? *reinterpret_cast<const uint4*>(b.raw + o) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) add_row(gq[u], rq[u], m[u], yq[u], r2q[u]);
        }
        for (; p < p1; p += rif) {
            const size_t o = (size_t)p * C + c;
            add_row(*reinterpret_cast<const uint4*>(g + o), *reinterpret_cast<const uint4*>(a.raw + o), ybits ? ybits[o >> 3] : 0u,
                    (!ybits && ymask) ? *reinterpret_cast<const uint4*>(ymask + o) : make_uint4(0, 0, 0, 0),
                    DUAL ? *reinterpret_cast<const uint4*>(b.raw + o) : make_uint4(0, 0, 0, 0));
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float* d = red + ((size_t)rsub * C + c + t) * NV;
            d[0] = s1[t]; d[1] = s2a[t];
            if (DUAL) d[2] = s2b[t];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * NV; e += 256) {
        float s = 0.f;
        for (int r = 0; r < rif; ++r) s += red[(size_t)r * C * NV + e];
        partial[(size_t)blockIdx.x * C * NV + e] = s;
    }
}

// partial [blocks][C][NV] -> dgamma = S2, dbeta = S1 and the folded apply coefficients, structure-of-arrays coef [3][C]:
//   draw = scale*(dz - S1/N - xhat*S2/N),  xhat = (raw - mean)*invstd
//        = A*dz + K - Q*raw   with  A = scale,  Q = scale*invstd*S2/N,  K = Q*mean - scale*S1/N
// second level of reduce_finish_kernel for one (NV = 2) or both (NV = 3: v = {S1, S2a, S2b}) BatchNorms behind a gradient
struct FinBnBwdSide {
    const float *scale, *mean, *invstd;
    float *coef, *dgamma, *dbeta;
};
template <int NV>
struct FinBnBwd {
    double count;
    int C;
    FinBnBwdSide side[NV - 1];
    __device__ void operator()(int c, const double* v) const {
#pragma unroll
        for (int k = 0; k < NV - 1; ++k) {
            const FinBnBwdSide& s = side[k];
            const double a = v[0], b = v[1 + k];
            const double sc = (double)s.scale[c];
            const double q = sc * (double)s.invstd[c] * (b / count);
            s.coef[c] = s.scale[c];
            s.coef[C + c] = (float)(q * (double)s.mean[c] - sc * (a / count));
            s.coef[2 * C + c] = (float)q;
            s.dgamma[c] = (float)b;
            s.dbeta[c] = (float)a;
        }
    }
};

template <bool DUAL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ ymask,
                                                            const uint8_t* __restrict__ ybits, BnBwdSide a, BnBwdSide b,
                                                            const float* __restrict__ coef_a, const float* __restrict__ coef_b, int relu,
                                                            size_t chunks, int C, uint16_t* __restrict__ draw_a,
                                                            uint16_t* __restrict__ draw_b, uint16_t* __restrict__ dz_out) {
    // coefficients once per thread when the grid stride is a multiple of C (see bn_act_kernel)
    const size_t first = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool fixed_c = ((size_t)gridDim.x * 2048) % (size_t)C == 0;
    const bool need_relu_coef = !ybits && !ymask && relu;
    int c = (int)((first * 8) % (size_t)C);
    float A[8], K[8], Q[8], A2[8], K2[8], Q2[8], sa[8], ha[8];
    auto load_coef = [&]() {
        load8f(coef_a + c, A); load8f(coef_a + C + c, K); load8f(coef_a + 2 * C + c, Q);
        if (DUAL) { load8f(coef_b + c, A2); load8f(coef_b + C + c, K2); load8f(coef_b + 2 * C + c, Q2); }
        if (need_relu_coef) { load8f(a.scale + c, sa); load8f(a.shift + c, ha); }
    };
    load_coef();
    for (size_t i = first; i < chunks; i += (size_t)gridDim.x * 256) {
        if (!fixed_c) { c = (int)((i * 8) % (size_t)C); load_coef(); }
        float gv[8], rv[8];
        unpack8(*reinterpret_cast<const uint4*>(g + i * 8), gv);
        unpack8(*reinterpret_cast<const uint4*>(a.raw + i * 8), rv);
        if (ybits) {
            const unsigned m = ybits[i];
#pragma unroll
            for (int t = 0; t < 8; ++t) gv[t] = ((m >> t) & 1u) ? gv[t] : 0.f;
        } else if (ymask) {
            float yv[8];
            unpack8(*reinterpret_cast<const uint4*>(ymask + i * 8), yv);
#pragma unroll
            for (int t = 0; t < 8; ++t) gv[t] = yv[t] > 0.f ? gv[t] : 0.f;
        } else if (relu) {
#pragma unroll
            for (int t = 0; t < 8; ++t) gv[t] = (rv[t] * sa[t] + ha[t]) > 0.f ? gv[t] : 0.f;
        }
        float o[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = A[t] * gv[t] + K[t] - Q[t] * rv[t];
        if (DUAL) {
            float r2[8], o2[8];
            unpack8(*reinterpret_cast<const uint4*>(b.raw + i * 8), r2);
#pragma unroll
            for (int t = 0; t < 8; ++t) o2[t] = A2[t] * gv[t] + K2[t] - Q2[t] * r2[t];
            *reinterpret_cast<uint4*>(draw_b + i * 8) = pack8(o2);
        }
        if (dz_out) *reinterpret_cast<uint4*>(dz_out + i * 8) = pack8(gv);     // may alias g (same index, read first)
        *reinterpret_cast<uint4*>(draw_a + i * 8) = pack8(o);                 // may alias g when dz_out is null
    }
}

// ------------------------------------------------------------------------------------------------
// Stem packing.  Image: fp32 NCHW [N,3,H,W] -> bf16 [N][H+6][Wp][4] zero-padded (3 px border, 4th channel 0,
// Wp = W+8).  A 7x7/2 tap row (8 taps x 4 ch = 32 bf16 = 64 B) is then one contiguous, aligned k-tile.
// Weight: fp32 [64][7][7][3] (OHWI, the channels_last storage of conv1.weight) -> bf16 [64][7][8][4].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_pack_image_kernel(const float* __restrict__ img, int N, int H, int W, int Hp, int Wp,
                                                               uint16_t* __restrict__ out) {
    const size_t total = (size_t)N * Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int wp = (int)(i % Wp);
        const int hp = (int)((i / Wp) % Hp);
        const int n = (int)(i / ((size_t)Wp * Hp));
        const int h = hp - 3, w = wp - 3;
        float v[3] = {0.f, 0.f, 0.f};
        if (h >= 0 && h < H && w >= 0 && w < W) {
            const size_t base = ((size_t)n * 3 * H + h) * W + w;
            v[0] = img[base]; v[1] = img[base + (size_t)H * W]; v[2] = img[base + 2 * (size_t)H * W];
        }
        *reinterpret_cast<uint2*>(out + i * 4) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], 0.f));
    }
}
__global__ void stem_pack_weight_kernel(const float* __restrict__ w, int Cout, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // over Cout*7*8*4
    if (i >= Cout * 224) return;
    const int c = i & 3, s = (i >> 2) & 7, r = (i >> 5) % 7, o = i / 224;
    float v = 0.f;
    if (c < 3 && s < 7) v = w[((o * 7 + r) * 7 + s) * 3 + c];
    out[i] = f32_to_bf16_bits(v);
}
__global__ void stem_unpack_wgrad_kernel(const float* __restrict__ padded, int Cout, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // over Cout*7*7*3
    if (i >= Cout * 147) return;
    const int c = i % 3, s = (i / 3) % 7, r = (i / 21) % 7, o = i / 147;
    dw[i] = padded[((o * 7 + r) * 8 + s) * 4 + c];
}

// ------------------------------------------------------------------------------------------------
// 3x3/2 pad-1 max-pool over z = raw*scale+shift (the stem BN, no ReLU), NHWC, 8 channels per thread.
// arg = tap index 0..8 of the first maximum in (r,s) scan order (torch's rule).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_bn_fwd_kernel(const uint16_t* __restrict__ raw, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int N, int H, int W, int C,
                                                              int Ho, int Wo, uint16_t* __restrict__ out, uint8_t* __restrict__ arg) {
    const int cpr = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cc = (int)(i % cpr) * 8;
        const size_t pix = i / cpr;
        const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Wo * Ho));
        float sc[8], sh[8], best[8];
        int bi[8];
        load8f(scale + cc, sc); load8f(shift + cc, sh);
#pragma unroll
        for (int t = 0; t < 8; ++t) { best[t] = -__builtin_inff(); bi[t] = 0; }
        // the nine taps are requested together from clamped addresses and gated by a predicate afterwards (a `continue` per out-of-range tap
        // put every load behind its own branch and wait: 105 us for 369 MB)
        uint4 q[9];
        bool ok[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = ho * 2 - 1 + r, hc = h < 0 ? 0 : (h >= H ? H - 1 : h);
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const int w = wo * 2 - 1 + s2, wc = w < 0 ? 0 : (w >= W ? W - 1 : w);
                ok[r * 3 + s2] = h >= 0 && h < H && w >= 0 && w < W;
                q[r * 3 + s2] = *reinterpret_cast<const uint4*>(raw + (((size_t)n * H + hc) * W + wc) * C + cc);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float v[8];
            unpack8(q[k], v);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float z = v[t] * sc[t] + sh[t];
                if (ok[k] && z > best[t]) { best[t] = z; bi[t] = k; }
            }
        }
        *reinterpret_cast<uint4*>(out + pix * C + cc) = pack8(best);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) { lo |= (uint32_t)bi[t] << (8 * t); hi |= (uint32_t)bi[t + 4] << (8 * t); }
        *reinterpret_cast<uint2*>(arg + pix * C + cc) = make_uint2(lo, hi);
    }
}

// dz[n,h,w,c] = sum over the (<=4) windows containing (h,w) of dp[window] * [arg[window] == tap of (h,w)]
__device__ __forceinline__ void maxpool_gather_dz(const uint16_t* __restrict__ dp, const uint8_t* __restrict__ arg, int n, int h, int w,
                                                  int cc, int C, int Ho, int Wo, float (&dz)[8]) {
    // 3x3 / stride 2 / pad 1: row h lies in window h>>1 (at offset 1 for even h, 2 for odd h) and, for odd h, in window (h>>1)+1 at
    // offset 0; the same for columns.  The (up to) four windows are read unconditionally from clamped addresses and gated by a
    // predicate, so the four load pairs are in flight together (the data-dependent 1..2 x 1..2 loops serialised them).
    const int ho0 = h >> 1, wo0 = w >> 1;
    const int rh[2] = {1 + (h & 1), 0}, rw[2] = {1 + (w & 1), 0};           // offsets inside window 0 / window 1
    const bool vh[2] = {ho0 < Ho, (h & 1) && ho0 + 1 < Ho}, vw[2] = {wo0 < Wo, (w & 1) && wo0 + 1 < Wo};
    uint4 gq[4];
    uint2 aq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = k >> 1, b = k & 1;
        const int ho = min(ho0 + a, Ho - 1), wo = min(wo0 + b, Wo - 1);
        const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + cc;
        aq[k] = *reinterpret_cast<const uint2*>(arg + o);
        gq[k] = *reinterpret_cast<const uint4*>(dp + o);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) dz[t] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                            // same order as the loops it replaces: (ho, wo) ascending
        const int a = k >> 1, b = k & 1;
        const int tap = (vh[a] && vw[b]) ? rh[a] * 3 + rw[b] : -1;
        float g[8];
        unpack8(gq[k], g);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int av = (t < 4 ? (aq[k].x >> (8 * t)) : (aq[k].y >> (8 * (t - 4)))) & 0xff;
            if (av == tap) dz[t] += g[t];
        }
    }
}

// pass 1: partial sums of dz and dz*xhat over the stem output;  pass 2: d_raw = scale*(dz - S1/N - xhat*S2/N)
__global__ __launch_bounds__(256) void maxpool_bn_bwd_reduce_kernel(const uint16_t* __restrict__ dp, const uint8_t* __restrict__ arg,
                                                                     const uint16_t* __restrict__ raw, const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd, int N, int H, int W, int C,
                                                                     int Ho, int Wo, int rows_per_block, float* __restrict__ partial) {
    extern __shared__ float red[];                  // [rif][C][2]
    const int cpr = C >> 3, rif = 256 / cpr;
    const int col = threadIdx.x % cpr, rsub = threadIdx.x / cpr;
    const int cc = col * 8;
    const int P = N * H * W;
    const int p0 = blockIdx.x * rows_per_block, p1 = min(P, p0 + rows_per_block);
    float s1[8], s2[8], m[8], iv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
    if (rsub < rif) {
        load8f(mean + cc, m); load8f(invstd + cc, iv);
        for (int p = p0 + rsub; p < p1; p += rif) {
            const int w = p % W, h = (p / W) % H, n = p / (W * H);
            float dz[8], rv[8];
            maxpool_gather_dz(dp, arg, n, h, w, cc, C, Ho, Wo, dz);
            unpack8(*reinterpret_cast<const uint4*>(raw + (size_t)p * C + cc), rv);
#pragma unroll
            for (int t = 0; t < 8; ++t) { s1[t] += dz[t]; s2[t] += dz[t] * ((rv[t] - m[t]) * iv[t]); }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) { red[((size_t)rsub * C + cc + t) * 2] = s1[t]; red[((size_t)rsub * C + cc + t) * 2 + 1] = s2[t]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * 2; e += 256) {
        float s = 0.f;
        for (int r = 0; r < rif; ++r) s += red[(size_t)r * C * 2 + e];
        partial[(size_t)blockIdx.x * C * 2 + e] = s;
    }
}
__global__ __launch_bounds__(256) void maxpool_bn_bwd_apply_kernel(const uint16_t* __restrict__ dp, const uint8_t* __restrict__ arg,
                                                                    const uint16_t* __restrict__ raw, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, const float* __restrict__ coef,
                                                                    int N, int H, int W, int C, int Ho, int Wo, uint16_t* __restrict__ draw) {
    const int cpr = C >> 3;
    const unsigned total = (unsigned)N * H * W * cpr;              // < 2^32 (checked by the launcher): 32-bit index arithmetic
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int cc = (int)(i % (unsigned)cpr) * 8;
        const unsigned p = i / (unsigned)cpr;
        const unsigned pw = p / (unsigned)W;
        const int w = (int)(p - pw * W), n = (int)(pw / (unsigned)H), h = (int)(pw - (unsigned)n * H);
        float dz[8], rv[8], A[8], K[8], Q[8], o[8];
        maxpool_gather_dz(dp, arg, n, h, w, cc, C, Ho, Wo, dz);
        unpack8(*reinterpret_cast<const uint4*>(raw + (size_t)p * C + cc), rv);
        load8f(coef + cc, A); load8f(coef + C + cc, K); load8f(coef + 2 * C + cc, Q);
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = A[t] * dz[t] + K[t] - Q[t] * rv[t];
        *reinterpret_cast<uint4*>(draw + (size_t)p * C + cc) = pack8(o);
    }
}

// Reduce pass on 2 x 2 pixel quads (H, W even), same window sharing as the apply pass below; a block owns a range of quads.
__global__ __launch_bounds__(256) void maxpool_bn_bwd_reduce_quad_kernel(const uint16_t* __restrict__ dp, const uint8_t* __restrict__ arg,
                                                                          const uint16_t* __restrict__ raw, const float* __restrict__ mean,
                                                                          const float* __restrict__ invstd, int N, int H, int W, int C,
                                                                          int Ho, int Wo, int quads_per_block, float* __restrict__ partial) {
    extern __shared__ float red[];                  // [rif][C][2]
    const int cpr = C >> 3, rif = 256 / cpr;
    const int col = threadIdx.x % cpr, rsub = threadIdx.x / cpr;
    const int cc = col * 8;
    const int Pq = N * Ho * Wo;
    const int q0 = blockIdx.x * quads_per_block, q1 = min(Pq, q0 + quads_per_block);
    float s1[8], s2[8], m[8], iv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
    if (rsub < rif) {
        load8f(mean + cc, m); load8f(invstd + cc, iv);
        for (int q = q0 + rsub; q < q1; q += rif) {
            const int j = q % Wo, k = (q / Wo) % Ho, n = q / (Wo * Ho);
            float g[4][8];
            int av[4][8];
            const bool vk = k + 1 < Ho, vj = j + 1 < Wo;
            uint4 rq[4];                                    // the quad's four raw chunks, requested with the windows' (not behind their arithmetic)
#pragma unroll
            for (int wdx = 0; wdx < 4; ++wdx)
                rq[wdx] = *reinterpret_cast<const uint4*>(raw + (((size_t)n * H + 2 * k + (wdx >> 1)) * W + 2 * j + (wdx & 1)) * C + cc);
#pragma unroll
            for (int wdx = 0; wdx < 4; ++wdx) {
                const int ho = min(k + (wdx >> 1), Ho - 1), wo = min(j + (wdx & 1), Wo - 1);
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + cc;
                const uint2 a2 = *reinterpret_cast<const uint2*>(arg + o);
                unpack8(*reinterpret_cast<const uint4*>(dp + o), g[wdx]);
#pragma unroll
                for (int t = 0; t < 8; ++t) av[wdx][t] = (int)(((t < 4 ? (a2.x >> (8 * t)) : (a2.y >> (8 * (t - 4)))) & 0xff));
            }
#pragma unroll
            for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    float dz[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) dz[t] = 0.f;
#pragma unroll
                    for (int wdx = 0; wdx < 4; ++wdx) {
                        const int da = wdx >> 1, db = wdx & 1;
                        if ((da && !pa) || (db && !pb)) continue;
                        const bool ok = (!da || vk) && (!db || vj);
                        const int tap = ok ? (da ? 0 : 1 + pa) * 3 + (db ? 0 : 1 + pb) : -1;
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (av[wdx][t] == tap) dz[t] += g[wdx][t];
                    }
                    float rv[8];
                    unpack8(rq[pa * 2 + pb], rv);
#pragma unroll
                    for (int t = 0; t < 8; ++t) { s1[t] += dz[t]; s2[t] += dz[t] * ((rv[t] - m[t]) * iv[t]); }
                }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) { red[((size_t)rsub * C + cc + t) * 2] = s1[t]; red[((size_t)rsub * C + cc + t) * 2 + 1] = s2[t]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * 2; e += 256) {
        float sum = 0.f;
        for (int r = 0; r < rif; ++r) sum += red[(size_t)r * C * 2 + e];
        partial[(size_t)blockIdx.x * C * 2 + e] = sum;
    }
}
// Apply pass on 2 x 2 pixel quads (H, W even): the four pixels of a quad draw on the same four pooling windows, so a thread loads each
// (dp, arg) window once instead of once per pixel (the gather above re-reads every window four times: 400 MB of L2 traffic for a
// 67 MB tensor).  Per pixel the windows are added in the same (ho, wo) order as maxpool_gather_dz: bit-identical results.
__global__ __launch_bounds__(256) void maxpool_bn_bwd_apply_quad_kernel(const uint16_t* __restrict__ dp, const uint8_t* __restrict__ arg,
                                                                         const uint16_t* __restrict__ raw, const float* __restrict__ coef,
                                                                         int N, int H, int W, int C, int Ho, int Wo, uint16_t* __restrict__ draw) {
    const int cpr = C >> 3;
    const unsigned total = (unsigned)N * Ho * Wo * cpr;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned q = i / (unsigned)cpr;
        const int cc = (int)(i - q * cpr) * 8;
        const unsigned qrow = q / (unsigned)Wo;
        const int j = (int)(q - qrow * Wo), n = (int)(qrow / (unsigned)Ho), k = (int)(qrow - (unsigned)n * Ho);
        float A[8], K[8], Q[8];
        load8f(coef + cc, A); load8f(coef + C + cc, K); load8f(coef + 2 * C + cc, Q);
        // windows (k, j), (k, j+1), (k+1, j), (k+1, j+1): clamped addresses, validity as predicates
        float g[4][8];
        int av[4][8];
        const bool vk = k + 1 < Ho, vj = j + 1 < Wo;
#pragma unroll
        for (int wdx = 0; wdx < 4; ++wdx) {
            const int ho = min(k + (wdx >> 1), Ho - 1), wo = min(j + (wdx & 1), Wo - 1);
            const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + cc;
            const uint2 a2 = *reinterpret_cast<const uint2*>(arg + o);
            unpack8(*reinterpret_cast<const uint4*>(dp + o), g[wdx]);
#pragma unroll
            for (int t = 0; t < 8; ++t) av[wdx][t] = (int)(((t < 4 ? (a2.x >> (8 * t)) : (a2.y >> (8 * (t - 4)))) & 0xff));
        }
        // (window, tap) lists per pixel (a, b) of the quad, in (ho, wo) order; tap = row offset * 3 + column offset inside the window
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                float dz[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) dz[t] = 0.f;
#pragma unroll
                for (int wdx = 0; wdx < 4; ++wdx) {
                    const int da = wdx >> 1, db = wdx & 1;
                    if ((da && !pa) || (db && !pb)) continue;                 // an even row / column lies in one window only
                    const bool ok = (!da || vk) && (!db || vj);
                    const int tap = ok ? (da ? 0 : 1 + pa) * 3 + (db ? 0 : 1 + pb) : -1;
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (av[wdx][t] == tap) dz[t] += g[wdx][t];
                }
                const size_t p = ((size_t)n * H + 2 * k + pa) * W + 2 * j + pb;
                float rv[8], o8[8];
                unpack8(*reinterpret_cast<const uint4*>(raw + p * C + cc), rv);
#pragma unroll
                for (int t = 0; t < 8; ++t) o8[t] = A[t] * dz[t] + K[t] - Q[t] * rv[t];
                *reinterpret_cast<uint4*>(draw + p * C + cc) = pack8(o8);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Head: f[n,c] = mean_hw x + max_hw x (Encoders.py:341-345), fp32 out, argmax kept for the backward.
// mode: DALI_FEATURE_BOTH / _GAP (mean only) / _GMP (max only) = the `feature` switch of evaluateCleanATModels.py:335-340.
// ------------------------------------------------------------------------------------------------
// 8 lanes share one (image, 8-channel chunk): lane s takes pixels s, s+8, ... and the 8 partial (sum, max, argmax) are combined by
// shuffles (ties: the lowest pixel index wins, as in a sequential scan with `>`).  One thread per chunk walking all HW pixels was
// latency-bound (84 us for 134 MB).
__global__ __launch_bounds__(256) void head_pool_fwd_kernel(const uint16_t* __restrict__ x, int N, int HW, int C, int mode,
                                                             float* __restrict__ f, int16_t* __restrict__ arg) {
    const int cpr = C >> 3;
    const int item = (blockIdx.x * 256 + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    const bool live = item < N * cpr;
    const int cc = live ? (item % cpr) * 8 : 0, n = live ? item / cpr : 0;
    float sum[8], best[8];
    int bi[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { sum[t] = 0.f; best[t] = -__builtin_inff(); bi[t] = 0x7fffffff; }
    if (live)
        for (int p = sub; p < HW; p += 8) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(x + ((size_t)n * HW + p) * C + cc), v);
#pragma unroll
            for (int t = 0; t < 8; ++t) { sum[t] += v[t]; if (v[t] > best[t]) { best[t] = v[t]; bi[t] = p; } }
        }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            sum[t] += __shfl_xor(sum[t], o, 64);
            const float ob = __shfl_xor(best[t], o, 64);
            const int oi = __shfl_xor(bi[t], o, 64);
            if (ob > best[t] || (ob == best[t] && oi < bi[t])) { best[t] = ob; bi[t] = oi; }
        }
    if (live && sub == 0) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f[(size_t)n * C + cc + t] = (mode == DALI_FEATURE_GMP ? 0.f : sum[t] / (float)HW) + (mode == DALI_FEATURE_GAP ? 0.f : best[t]);
            arg[(size_t)n * C + cc + t] = (int16_t)bi[t];
        }
    }
}
__global__ __launch_bounds__(256) void head_pool_bwd_kernel(const float* __restrict__ df, const int16_t* __restrict__ arg, int N, int HW,
                                                             int C, int mode, uint16_t* __restrict__ dx, const uint8_t* __restrict__ ybits) {
    const int cpr = C >> 3;
    const unsigned total = (unsigned)N * HW * cpr;              // < 2^32 (checked by the launcher): 32-bit index arithmetic
    const bool hw_pow2 = (HW & (HW - 1)) == 0;
    const float inv_hw = 1.f / (float)HW;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned pix = i / (unsigned)cpr;
        const int cc = (int)(i - pix * cpr) * 8;
        const int n = (int)(pix / (unsigned)HW), p = (int)(pix - (unsigned)n * HW);
        float g[8], o[8];
        load8f(df + (size_t)n * C + cc, g);
        const uint4 av = *reinterpret_cast<const uint4*>(arg + (size_t)n * C + cc);
        const int a[8] = {(int)(int16_t)(av.x & 0xffff), (int)(int16_t)(av.x >> 16), (int)(int16_t)(av.y & 0xffff), (int)(int16_t)(av.y >> 16),
                          (int)(int16_t)(av.z & 0xffff), (int)(int16_t)(av.z >> 16), (int)(int16_t)(av.w & 0xffff), (int)(int16_t)(av.w >> 16)};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // (HW a power of two -- 16 x 8 at 256 x 128 -- : the product with 1 / HW is the quotient, bit for bit, without eight division sequences)
            const float avg = hw_pow2 ? g[t] * inv_hw : g[t] / (float)HW;
            o[t] = (mode == DALI_FEATURE_GMP ? 0.f : avg) + ((mode != DALI_FEATURE_GAP && a[t] == p) ? g[t] : 0.f);
        }
        if (ybits) {                                  // dz = dy * (y > 0): the block's BatchNorm backward consumes the masked gradient
            const unsigned m = ybits[((size_t)pix * C + cc) >> 3];
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] = ((m >> t) & 1u) ? o[t] : 0.f;
        }
        *reinterpret_cast<uint4*>(dx + (size_t)pix * C + cc) = pack8(o);
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm1d neck on fp32 [N,C].
// ------------------------------------------------------------------------------------------------
// block = 32 channels x 8 row slices (thread: channel tid&31, rows tid>>5, +8, ...); fp64 partial sums, slices added in a
// fixed order.  (A single thread per channel walking all N rows was latency-bound: 120 us for a 2 MB tensor.)
__device__ __forceinline__ double bn1d_slice_sum(double v, double (*red)[32]) {
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    __syncthreads();                                   // previous use of red is over
    red[ry][cx] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][cx];
    return s;
}
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const float* __restrict__ x, int N, int C, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ rm, float* __restrict__ rv,
                                                        int training, float momentum, float eps, float* __restrict__ y,
                                                        float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    __shared__ double red[8][32];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const bool ok = c < C;
    float mean = 0.f, invstd = 1.f;
    if (training) {
        double s = 0.0;
        // (the row loops of this kernel and of bn1d_bwd are unrolled by 8: with a run-time trip count and an fp64 chain the compiler issued ONE load per
        //  round trip -- 32 rounds per pass at batch 256, 17 us for a 2 MB tensor; the additions keep their order)
        if (ok) {
#pragma unroll 8
            for (int n = ry; n < N; n += 8) s += (double)x[(size_t)n * C + c];
        }
        const double m = bn1d_slice_sum(s, red) / N;
        double v = 0.0;
        if (ok) {
#pragma unroll 8
            for (int n = ry; n < N; n += 8) { const double d = (double)x[(size_t)n * C + c] - m; v += d * d; }
        }
        v = bn1d_slice_sum(v, red);
        const double var = v / N;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (rm && ok && ry == 0) {
            const double unbiased = N > 1 ? v / (N - 1) : var;
            rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
            rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unbiased;
        }
    } else if (ok) {
        mean = rm[c];
        invstd = 1.0f / sqrtf(rv[c] + eps);
    }
    if (!ok) return;
    if (mean_out && ry == 0) { mean_out[c] = mean; invstd_out[c] = invstd; }
    const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
#pragma unroll 8
    for (int n = ry; n < N; n += 8) y[(size_t)n * C + c] = x[(size_t)n * C + c] * sc + sh;
}
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ dx,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[8][32];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const bool ok = c < C;
    const float m = ok ? mean[c] : 0.f, iv = ok ? invstd[c] : 1.f;
    double s1 = 0.0, s2 = 0.0;
    if (ok) {
#pragma unroll 8
        for (int n = ry; n < N; n += 8) {
            const float g = dy[(size_t)n * C + c];
            s1 += (double)g; s2 += (double)g * (double)((x[(size_t)n * C + c] - m) * iv);
        }
    }
    s1 = bn1d_slice_sum(s1, red);
    s2 = bn1d_slice_sum(s2, red);
    if (!ok) return;
    if (ry == 0) { dgamma[c] = (float)s2; if (dbeta) dbeta[c] = (float)s1; }       // dbeta null: frozen bias (make_models.py:181)
    const float a = (float)(s1 / N), b = (float)(s2 / N), sc = gamma[c] * iv;
#pragma unroll 8
    for (int n = ry; n < N; n += 8) {
        const float xh = (x[(size_t)n * C + c] - m) * iv;
        dx[(size_t)n * C + c] = sc * (dy[(size_t)n * C + c] - a - xh * b);
    }
}

// ------------------------------------------------------------------------------------------------
// Weight layout helpers: fp32 -> bf16 cast (flat), and [Co][T][Ci] -> [Ci][T][Co] bf16 for dgrad.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, size_t n, uint16_t* __restrict__ y) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            *reinterpret_cast<uint2*>(y + i) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        } else {
            for (size_t e = i; e < n; ++e) y[e] = f32_to_bf16_bits(x[e]);
        }
    }
}
// one block transposes a 32(co) x 32(ci) tile of one tap
__global__ __launch_bounds__(256) void weight_transpose_kernel(const uint16_t* __restrict__ w, int Co, int T, int Ci,
                                                                uint16_t* __restrict__ wt) {
    __shared__ uint16_t tile[32][33];
    const int tiles_ci = (Ci + 31) / 32, tiles_co = (Co + 31) / 32;
    const int b = blockIdx.x;
    const int tci = b % tiles_ci, tco = (b / tiles_ci) % tiles_co, tap = b / (tiles_ci * tiles_co);
    const int x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
    for (int y = y0; y < 32; y += 8) {
        const int co = tco * 32 + y, ci = tci * 32 + x;
        tile[y][x] = (co < Co && ci < Ci) ? w[((size_t)co * T + tap) * Ci + ci] : (uint16_t)0;
    }
    __syncthreads();
    for (int y = y0; y < 32; y += 8) {
        const int ci = tci * 32 + y, co = tco * 32 + x;
        if (ci < Ci && co < Co) wt[((size_t)ci * T + tap) * Co + co] = tile[x][y];
    }
}

// all dgrad weight images of a net in one launch: the job table travels as a kernel argument
constexpr int TRANSPOSE_BATCH = 56;
struct TransposeBatch { TransposeJob job[TRANSPOSE_BATCH]; int n; };
// 64 x 64 tiles moved with 16-byte global accesses (Co, Ci multiples of 64: every conv / linear weight of the nets); the
// 2-byte-per-lane 32 x 32 form below ran the 47 MB of weights at 0.9 TB/s
__global__ __launch_bounds__(256) void weight_transpose64_batched_kernel(TransposeBatch batch) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64][72];
    int ji = 0;
    while (ji + 1 < batch.n && (int)blockIdx.x >= batch.job[ji + 1].first_block) ++ji;       // block-uniform scan
    const TransposeJob jb = batch.job[ji];
    const int Co = jb.Co, T = jb.T, Ci = jb.Ci;
    const int tiles_ci = Ci / 64, tiles_co = Co / 64;
    const int b = blockIdx.x - jb.first_block;
    const int tci = b % tiles_ci, tco = (b / tiles_ci) % tiles_co, tap = b / (tiles_ci * tiles_co);
    const int ch = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = r0 + 32 * k;                    // co inside the tile
        *reinterpret_cast<uint4*>(&tile[row][ch * 8]) =
            *reinterpret_cast<const uint4*>(jb.w + ((size_t)(tco * 64 + row) * T + tap) * Ci + tci * 64 + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = r0 + 32 * k;                    // ci inside the tile
        uint16_t e[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) e[t] = tile[ch * 8 + t][row];
        const uint4 v = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                                   (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
        *reinterpret_cast<uint4*>(jb.wt + ((size_t)(tci * 64 + row) * T + tap) * Co + tco * 64 + ch * 8) = v;
    }
}
__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(TransposeBatch batch) {
    __shared__ uint16_t tile[32][33];
    int ji = 0;
    while (ji + 1 < batch.n && (int)blockIdx.x >= batch.job[ji + 1].first_block) ++ji;       // block-uniform scan
    const TransposeJob jb = batch.job[ji];
    const int Co = jb.Co, T = jb.T, Ci = jb.Ci;
    const int tiles_ci = (Ci + 31) / 32, tiles_co = (Co + 31) / 32;
    const int b = blockIdx.x - jb.first_block;
    const int tci = b % tiles_ci, tco = (b / tiles_ci) % tiles_co, tap = b / (tiles_ci * tiles_co);
    const int x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
    for (int y = y0; y < 32; y += 8) {
        const int co = tco * 32 + y, ci = tci * 32 + x;
        tile[y][x] = (co < Co && ci < Ci) ? jb.w[((size_t)co * T + tap) * Ci + ci] : (uint16_t)0;
    }
    __syncthreads();
    for (int y = y0; y < 32; y += 8) {
        const int ci = tci * 32 + y, co = tco * 32 + x;
        if (ci < Ci && co < Co) jb.wt[((size_t)ci * T + tap) * Co + co] = tile[x][y];
    }
}

static inline int grid_for(size_t work_items, int cap = 16384) {
    size_t b = (work_items + 255) / 256;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- host launchers (shared with the net plan) ----------------------------------------------------
int launch_bn_finalize(hipStream_t st, const float* partial, int tiles, int C, double count, const float* gamma, const float* beta,
                       float* rm, float* rv, float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                       double* scratch) {
    return launch_reduce_finish<2>(st, partial, tiles, C, scratch, FinBnFwd{count, gamma, beta, rm, rv, momentum, eps, scale, shift, mean, invstd});
}
int launch_bn_eval_coeffs(hipStream_t st, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                          float* scale, float* shift) {
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, rm, rv, eps, C, scale, shift);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_bn_eval_coeffs_batched(hipStream_t st, const BnEvalJob* jobs, int n, float eps) {
    for (int first = 0; first < n; first += BnEvalJobs::MAX) {
        BnEvalJobs chunk;
        const int m = n - first < BnEvalJobs::MAX ? n - first : BnEvalJobs::MAX;
        int cmax = 1;
        for (int i = 0; i < m; ++i) { chunk.job[i] = jobs[first + i]; cmax = chunk.job[i].C > cmax ? chunk.job[i].C : cmax; }
        hipLaunchKernelGGL(bn_eval_coeffs_batched_kernel, dim3((cmax + 255) / 256, m), dim3(256), 0, st, chunk, eps);
        DALI_LAUNCH_CHECK();
    }
    return DALI_OK;
}
int launch_fold_cat_weights(hipStream_t st, const float* w3, const float* wd, const float* s3, const float* sd, const float* h3, const float* hd, int C, int w,
                            int cin, int parts, uint16_t* wcat, float* shcat) {
    const size_t n = (size_t)C * parts * (w + cin);
    hipLaunchKernelGGL(fold_cat_weights_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, w3, wd, s3, sd, h3, hd, C, w, cin, parts, wcat, shcat);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_bn_act(hipStream_t st, const uint16_t* raw, const float* scale, const float* shift, const uint16_t* idn, const uint16_t* raw2,
                  const float* scale2, const float* shift2, int relu, size_t elems, int C, uint16_t* y, uint8_t* mask_out) {
    const size_t chunks = elems / 8;
    hipLaunchKernelGGL(bn_act_kernel, dim3(grid_for(chunks)), dim3(256), 0, st, raw, scale, shift, idn, raw2, scale2, shift2, relu, chunks, C, y, mask_out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
// Number of partial blocks the reduce pass uses for a [P][C] tensor
int bn_bwd_blocks(int P, int C, int* rows_per_block) {
    const int rif = 256 / (C / 8);
    int blocks = (P + rif * 8 - 1) / (rif * 8);           // >= 8 rows per thread
    if (blocks > 1024) blocks = 1024;                      // (768 .. 3072 measured the same)
    if (blocks < 1) blocks = 1;
    int rpb = (P + blocks - 1) / blocks;
    rpb = (rpb + rif - 1) / rif * rif;
    blocks = (P + rpb - 1) / rpb;
    *rows_per_block = rpb;
    return blocks;
}
int launch_bn_bwd(hipStream_t st, const uint16_t* g, const uint16_t* ymask, const uint8_t* ybits, const BnBwdSide& a, const BnBwdSide* b, int relu, int P, int C,
                  float* partial, float* coef_a, float* coef_b, float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b,
                  uint16_t* draw_a, uint16_t* draw_b, uint16_t* dz_out, double* scratch) {
    int rpb;
    const int blocks = bn_bwd_blocks(P, C, &rpb);
    const int rif = 256 / (C / 8);
    const bool dual = b != nullptr;
    const int NV = dual ? 3 : 2;
    const size_t lds = (size_t)rif * C * NV * sizeof(float);
    BnBwdSide bb = dual ? *b : a;
    if (dual) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(blocks), dim3(256), lds, st, g, ymask, ybits, a, bb, relu, P, C, rpb, partial);
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(blocks), dim3(256), lds, st, g, ymask, ybits, a, bb, relu, P, C, rpb, partial);
    }
    DALI_LAUNCH_CHECK();
    int rc;
    if (dual) {
        FinBnBwd<3> fin{(double)P, C, {{a.scale, a.mean, a.invstd, coef_a, dgamma_a, dbeta_a}, {bb.scale, bb.mean, bb.invstd, coef_b, dgamma_b, dbeta_b}}};
        rc = launch_reduce_finish<3>(st, partial, blocks, C, scratch, fin);
    } else {
        FinBnBwd<2> fin{(double)P, C, {{a.scale, a.mean, a.invstd, coef_a, dgamma_a, dbeta_a}}};
        rc = launch_reduce_finish<2>(st, partial, blocks, C, scratch, fin);
    }
    if (rc) return rc;
    const size_t chunks = (size_t)P * C / 8;
    if (dual) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid_for(chunks)), dim3(256), 0, st, g, ymask, ybits, a, bb, coef_a, coef_b, relu, chunks, C, draw_a, draw_b, dz_out);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for(chunks)), dim3(256), 0, st, g, ymask, ybits, a, bb, coef_a, coef_a, relu, chunks, C, draw_a, draw_b, dz_out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
size_t bn_bwd_partial_floats(int P, int C, bool dual) {
    int rpb;
    return (size_t)bn_bwd_blocks(P, C, &rpb) * C * (dual ? 3 : 2);
}
int launch_stem_pack_image(hipStream_t st, const float* img, int N, int H, int W, uint16_t* out) {
    const int Hp = H + 6, Wp = W + 8;
    hipLaunchKernelGGL(stem_pack_image_kernel, dim3(grid_for((size_t)N * Hp * Wp)), dim3(256), 0, st, img, N, H, W, Hp, Wp, out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_stem_pack_weight(hipStream_t st, const float* w, int Cout, uint16_t* out) {
    hipLaunchKernelGGL(stem_pack_weight_kernel, dim3((Cout * 224 + 255) / 256), dim3(256), 0, st, w, Cout, out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_stem_unpack_wgrad(hipStream_t st, const float* padded, int Cout, float* dw) {
    hipLaunchKernelGGL(stem_unpack_wgrad_kernel, dim3((Cout * 147 + 255) / 256), dim3(256), 0, st, padded, Cout, dw);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_maxpool_bn_fwd(hipStream_t st, const uint16_t* raw, const float* scale, const float* shift, int N, int H, int W, int C,
                          uint16_t* out, uint8_t* arg) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bn_fwd_kernel, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(256), 0, st, raw, scale, shift, N, H, W, C,
                       Ho, Wo, out, arg);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_maxpool_bn_bwd(hipStream_t st, const uint16_t* dp, const uint8_t* arg, const uint16_t* raw, const float* mean, const float* invstd,
                          const float* scale, int N, int H, int W, int C, float* partial, float* coef, float* dgamma, float* dbeta,
                          uint16_t* draw, double* scratch) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int P = N * H * W;
    if ((long long)P * (C / 8) >= (1ll << 32)) { set_error("maxpool_bn_bwd: more than 2^32 16-byte chunks"); return DALI_ERR_LIMIT; }
    int rpb;
    const bool quads = (H & 1) == 0 && (W & 1) == 0;
    // (the partial buffer is sized for bn_bwd_blocks(P, C): the quad form needs at most as many rows)
    const int blocks = quads ? bn_bwd_blocks(N * Ho * Wo, C, &rpb) : bn_bwd_blocks(P, C, &rpb);
    const int rif = 256 / (C / 8);
    if (quads) hipLaunchKernelGGL(maxpool_bn_bwd_reduce_quad_kernel, dim3(blocks), dim3(256), (size_t)rif * C * 2 * sizeof(float), st, dp, arg, raw, mean,
                                  invstd, N, H, W, C, Ho, Wo, rpb, partial);
    else hipLaunchKernelGGL(maxpool_bn_bwd_reduce_kernel, dim3(blocks), dim3(256), (size_t)rif * C * 2 * sizeof(float), st, dp, arg, raw, mean, invstd,
                            N, H, W, C, Ho, Wo, rpb, partial);
    DALI_LAUNCH_CHECK();
    int rc;
    {
        FinBnBwd<2> fin{(double)P, C, {{scale, mean, invstd, coef, dgamma, dbeta}}};
        if ((rc = launch_reduce_finish<2>(st, partial, blocks, C, scratch, fin))) return rc;
    }
    if ((H & 1) == 0 && (W & 1) == 0)
        hipLaunchKernelGGL(maxpool_bn_bwd_apply_quad_kernel, dim3(grid_for((size_t)N * Ho * Wo * (C / 8))), dim3(256), 0, st, dp, arg, raw, coef, N, H, W, C,
                           Ho, Wo, draw);
    else
        hipLaunchKernelGGL(maxpool_bn_bwd_apply_kernel, dim3(grid_for((size_t)P * (C / 8))), dim3(256), 0, st, dp, arg, raw, mean, invstd, coef, N, H,
                           W, C, Ho, Wo, draw);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_head_pool_fwd(hipStream_t st, const uint16_t* x, int N, int HW, int C, int mode, float* f, int16_t* arg) {
    hipLaunchKernelGGL(head_pool_fwd_kernel, dim3((N * (C / 8) * 8 + 255) / 256), dim3(256), 0, st, x, N, HW, C, mode, f, arg);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_head_pool_bwd(hipStream_t st, const float* df, const int16_t* arg, int N, int HW, int C, int mode, uint16_t* dx, const uint8_t* ybits) {
    if ((long long)N * HW * (C / 8) >= (1ll << 32)) { set_error("head_pool_bwd: more than 2^32 16-byte chunks"); return DALI_ERR_LIMIT; }
    hipLaunchKernelGGL(head_pool_bwd_kernel, dim3(grid_for((size_t)N * HW * (C / 8))), dim3(256), 0, st, df, arg, N, HW, C, mode, dx, ybits);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_bn1d_fwd(hipStream_t st, const float* x, int N, int C, const float* gamma, const float* beta, float* rm, float* rv, int training,
                    float momentum, float eps, float* y, float* mean, float* invstd) {
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((C + 31) / 32), dim3(256), 0, st, x, N, C, gamma, beta, rm, rv, training, momentum, eps, y, mean, invstd);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_bn1d_bwd(hipStream_t st, const float* x, const float* dy, int N, int C, const float* gamma, const float* mean, const float* invstd,
                    float* dx, float* dgamma, float* dbeta) {
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((C + 31) / 32), dim3(256), 0, st, x, dy, N, C, gamma, mean, invstd, dx, dgamma, dbeta);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_cast_bf16(hipStream_t st, const float* x, size_t n, uint16_t* y) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((n + 3) / 4, 4096)), dim3(256), 0, st, x, n, y);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_weight_transpose_batched(hipStream_t st, const TransposeJob* jobs, int n) {
    bool all64 = true;
    for (int i = 0; i < n; ++i)
        all64 = all64 && jobs[i].Ci % 64 == 0 && jobs[i].Co % 64 == 0 && ((reinterpret_cast<uintptr_t>(jobs[i].w) | reinterpret_cast<uintptr_t>(jobs[i].wt)) & 15) == 0;
    const int tile = all64 ? 64 : 32;
    for (int begin = 0; begin < n; begin += TRANSPOSE_BATCH) {
        TransposeBatch batch{};
        batch.n = n - begin < TRANSPOSE_BATCH ? n - begin : TRANSPOSE_BATCH;
        int blocks = 0;
        for (int i = 0; i < batch.n; ++i) {
            batch.job[i] = jobs[begin + i];
            batch.job[i].first_block = blocks;
            blocks += ((batch.job[i].Ci + tile - 1) / tile) * ((batch.job[i].Co + tile - 1) / tile) * batch.job[i].T;
        }
        if (all64) hipLaunchKernelGGL(weight_transpose64_batched_kernel, dim3(blocks), dim3(256), 0, st, batch);
        else hipLaunchKernelGGL(weight_transpose_batched_kernel, dim3(blocks), dim3(256), 0, st, batch);
        DALI_LAUNCH_CHECK();
    }
    return DALI_OK;
}
int launch_weight_transpose(hipStream_t st, const uint16_t* w, int Co, int T, int Ci, uint16_t* wt) {
    const int blocks = ((Ci + 31) / 32) * ((Co + 31) / 32) * T;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(blocks), dim3(256), 0, st, w, Co, T, Ci, wt);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

}  // namespace dali

// ---- single-op C ABI (parity tests; the net plan calls the launchers directly) ------------------------------
using namespace dali;

extern "C" int dali_bn_finalize(dali_ctx* ctx, void* stream, const float* partial, int tiles, int C, double count,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float momentum, float eps, float* scale, float* shift, float* mean, float* invstd) {
    DALI_REQUIRE(ctx && partial && gamma && beta && scale && shift && mean && invstd, "dali_bn_finalize: null argument");
    DALI_REQUIRE(tiles > 0 && C > 0 && count > 0, "dali_bn_finalize: bad sizes");
    double* scratch = static_cast<double*>(workspace(ctx, reduce_scratch_bytes(C, 2)));
    if (!scratch) return DALI_ERR_NOMEM;
    return launch_bn_finalize((hipStream_t)stream, partial, tiles, C, count, gamma, beta, running_mean, running_var, momentum, eps, scale,
                              shift, mean, invstd, scratch);
}

extern "C" int dali_bn_act(dali_ctx* ctx, void* stream, const uint16_t* raw, const float* scale, const float* shift,
                           const uint16_t* identity, const uint16_t* raw2, const float* scale2, const float* shift2, int relu,
                           int64_t pixels, int C, uint16_t* y, uint8_t* mask_out) {
    DALI_REQUIRE(ctx && raw && scale && shift && y, "dali_bn_act: null argument");
    DALI_REQUIRE(C % 8 == 0 && pixels >= 0, "dali_bn_act: C must be a multiple of 8");
    DALI_REQUIRE(!(identity && raw2), "dali_bn_act: identity and raw2 are exclusive");
    DALI_REQUIRE(!raw2 || (scale2 && shift2), "dali_bn_act: raw2 needs scale2/shift2");
    if (pixels == 0) return DALI_OK;
    return launch_bn_act((hipStream_t)stream, raw, scale, shift, identity, raw2, scale2, shift2, relu, (size_t)pixels * C, C, y, mask_out);
}

extern "C" int dali_bn_bwd(dali_ctx* ctx, void* stream, const uint16_t* g, const uint16_t* ymask, const uint8_t* ybits, int relu, int64_t pixels, int C,
                           const uint16_t* raw_a, const float* mean_a, const float* invstd_a, const float* scale_a, const float* shift_a,
                           const uint16_t* raw_b, const float* mean_b, const float* invstd_b, const float* scale_b,
                           float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, uint16_t* draw_a, uint16_t* draw_b,
                           uint16_t* dz_out) {
    DALI_REQUIRE(ctx && g && raw_a && mean_a && invstd_a && scale_a && dgamma_a && dbeta_a && draw_a, "dali_bn_bwd: null argument");
    DALI_REQUIRE(C % 8 == 0 && C <= 2048 && pixels > 0 && pixels < (1ll << 31), "dali_bn_bwd: C must be a multiple of 8, <= 2048");
    DALI_REQUIRE(ymask || ybits || !relu || shift_a, "dali_bn_bwd: relu mask needs ymask, ybits or scale/shift");
    DALI_REQUIRE(!(ymask && ybits), "dali_bn_bwd: ymask and ybits are exclusive");
    const bool dual = raw_b != nullptr;
    DALI_REQUIRE(!dual || (mean_b && invstd_b && scale_b && dgamma_b && dbeta_b && draw_b), "dali_bn_bwd: incomplete second side");
    const size_t pf = bn_bwd_partial_floats((int)pixels, C, dual);
    const size_t need = align_up(pf * 4, 256) + 2 * align_up((size_t)C * 12, 256) + reduce_scratch_bytes(C, 3);
    char* ws = static_cast<char*>(workspace(ctx, need));
    if (!ws) return DALI_ERR_NOMEM;
    float* partial = reinterpret_cast<float*>(ws);
    float* coef_a = reinterpret_cast<float*>(ws + align_up(pf * 4, 256));
    float* coef_b = reinterpret_cast<float*>(ws + align_up(pf * 4, 256) + align_up((size_t)C * 12, 256));
    double* scratch = reinterpret_cast<double*>(ws + align_up(pf * 4, 256) + 2 * align_up((size_t)C * 12, 256));
    BnBwdSide a{raw_a, mean_a, invstd_a, scale_a, shift_a};
    BnBwdSide b{raw_b, mean_b, invstd_b, scale_b, nullptr};
    return launch_bn_bwd((hipStream_t)stream, g, ymask, ybits, a, dual ? &b : nullptr, relu, (int)pixels, C, partial, coef_a, coef_b, dgamma_a, dbeta_a,
                         dgamma_b, dbeta_b, draw_a, draw_b, dz_out, scratch);
}

extern "C" int dali_maxpool_bn_fwd(dali_ctx* ctx, void* stream, const uint16_t* raw, const float* scale, const float* shift, int n, int h,
                                   int w, int C, uint16_t* out, uint8_t* arg) {
    DALI_REQUIRE(ctx && raw && scale && shift && out && arg, "dali_maxpool_bn_fwd: null argument");
    DALI_REQUIRE(C % 8 == 0 && n > 0 && h > 0 && w > 0, "dali_maxpool_bn_fwd: bad shape");
    return launch_maxpool_bn_fwd((hipStream_t)stream, raw, scale, shift, n, h, w, C, out, arg);
}

extern "C" int dali_maxpool_bn_bwd(dali_ctx* ctx, void* stream, const uint16_t* dpool, const uint8_t* arg, const uint16_t* raw,
                                   const float* mean, const float* invstd, const float* scale, int n, int h, int w, int C,
                                   float* dgamma, float* dbeta, uint16_t* draw) {
    DALI_REQUIRE(ctx && dpool && arg && raw && mean && invstd && scale && dgamma && dbeta && draw, "dali_maxpool_bn_bwd: null argument");
    DALI_REQUIRE(C % 8 == 0 && C <= 2048, "dali_maxpool_bn_bwd: C must be a multiple of 8, <= 2048");
    const size_t pf = bn_bwd_partial_floats(n * h * w, C, false);
    const size_t need = align_up(pf * 4, 256) + align_up((size_t)C * 12, 256) + reduce_scratch_bytes(C, 2);
    char* ws = static_cast<char*>(workspace(ctx, need));
    if (!ws) return DALI_ERR_NOMEM;
    return launch_maxpool_bn_bwd((hipStream_t)stream, dpool, arg, raw, mean, invstd, scale, n, h, w, C, reinterpret_cast<float*>(ws),
                                 reinterpret_cast<float*>(ws + align_up(pf * 4, 256)), dgamma, dbeta, draw,
                                 reinterpret_cast<double*>(ws + align_up(pf * 4, 256) + align_up((size_t)C * 12, 256)));
}

extern "C" int dali_head_pool_fwd(dali_ctx* ctx, void* stream, const uint16_t* x, int n, int hw, int C, int mode, float* f, int16_t* arg) {
    DALI_REQUIRE(ctx && x && f && arg, "dali_head_pool_fwd: null argument");
    DALI_REQUIRE(mode >= DALI_FEATURE_BOTH && mode <= DALI_FEATURE_GMP, "dali_head_pool_fwd: bad feature mode %d", mode);
    DALI_REQUIRE(C % 8 == 0 && hw > 0 && hw < 32768, "dali_head_pool_fwd: bad shape");
    return launch_head_pool_fwd((hipStream_t)stream, x, n, hw, C, mode, f, arg);
}
extern "C" int dali_head_pool_bwd(dali_ctx* ctx, void* stream, const float* df, const int16_t* arg, int n, int hw, int C, int mode, uint16_t* dx) {
    DALI_REQUIRE(ctx && df && arg && dx, "dali_head_pool_bwd: null argument");
    DALI_REQUIRE(mode >= DALI_FEATURE_BOTH && mode <= DALI_FEATURE_GMP, "dali_head_pool_bwd: bad feature mode %d", mode);
    DALI_REQUIRE(C % 8 == 0 && hw > 0, "dali_head_pool_bwd: bad shape");
    return launch_head_pool_bwd((hipStream_t)stream, df, arg, n, hw, C, mode, dx);
}
extern "C" int dali_bn1d_fwd(dali_ctx* ctx, void* stream, const float* x, int n, int C, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int training, float momentum, float eps, float* y, float* mean,
                             float* invstd) {
    DALI_REQUIRE(ctx && x && gamma && beta && y, "dali_bn1d_fwd: null argument");
    DALI_REQUIRE(training || (running_mean && running_var), "dali_bn1d_fwd: eval mode needs running statistics");
    return launch_bn1d_fwd((hipStream_t)stream, x, n, C, gamma, beta, running_mean, running_var, training, momentum, eps, y, mean, invstd);
}
extern "C" int dali_bn1d_bwd(dali_ctx* ctx, void* stream, const float* x, const float* dy, int n, int C, const float* gamma,
                             const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta) {
    DALI_REQUIRE(ctx && x && dy && gamma && mean && invstd && dx && dgamma && dbeta, "dali_bn1d_bwd: null argument");
    return launch_bn1d_bwd((hipStream_t)stream, x, dy, n, C, gamma, mean, invstd, dx, dgamma, dbeta);
}
