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

namespace dali {

namespace {
constexpr int BL_CH = 8;          // channels per workgroup of the stats / row kernels

__device__ __forceinline__ double block_sum_d(double v, double* red) {       // 256 threads; result in every thread
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// wf[ch][k] (LDS, fp32) <- bf16 rows c0 .. c0+BL_CH-1 of W [C][w]
__device__ __forceinline__ void load_w_rows(const uint16_t* __restrict__ W, int c0, int C, int w, float* wf) {
    for (int e = threadIdx.x; e < BL_CH * w; e += 256) {
        const int ch = e / w, k = e - ch * w;
        wf[e] = (c0 + ch < C) ? bf16_bits_to_f32(W[(size_t)(c0 + ch) * w + k]) : 0.f;
    }
}

// u[ch] = sum_k wf[ch][k] * gram[k][kp] for this thread's column kp (gram rows are read coalesced across the threads' columns)
__device__ __forceinline__ void gram_column(const float* __restrict__ gram, const float* wf, int w, int kp, float (&u)[BL_CH]) {
#pragma unroll
    for (int ch = 0; ch < BL_CH; ++ch) u[ch] = 0.f;
    for (int k = 0; k < w; k += 4) {
        const float g0 = gram[(size_t)k * w + kp], g1 = gram[(size_t)(k + 1) * w + kp], g2 = gram[(size_t)(k + 2) * w + kp], g3 = gram[(size_t)(k + 3) * w + kp];
#pragma unroll
        for (int ch = 0; ch < BL_CH; ++ch) {
            const float4 wv = *reinterpret_cast<const float4*>(wf + ch * w + k);
            u[ch] += wv.x * g0 + wv.y * g1 + wv.z * g2 + wv.w * g3;
        }
    }
}
}  // namespace

// ---- forward: batch statistics of raw3 = a2 W^T from the moments of a2 -> scale / shift / mean / invstd (+ running statistics) ----
__global__ __launch_bounds__(256) void bnlin_stats_kernel(const uint16_t* __restrict__ W, const float* __restrict__ gram, const float* __restrict__ m2,
                                                           int C, int w, double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                                           float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                                           float* __restrict__ invstd_out) {
    extern __shared__ __attribute__((aligned(16))) float bl_smem[];
    float* wf = bl_smem;                                    // [BL_CH][w]
    __shared__ double red[4];
    const int c0 = blockIdx.x * BL_CH;
    load_w_rows(W, c0, C, w, wf);
    __syncthreads();
    double q[BL_CH], mu[BL_CH];
#pragma unroll
    for (int ch = 0; ch < BL_CH; ++ch) { q[ch] = 0.0; mu[ch] = 0.0; }
    for (int kp = threadIdx.x; kp < w; kp += 256) {
        float u[BL_CH];
        gram_column(gram, wf, w, kp, u);
        const float m = m2[kp];
#pragma unroll
        for (int ch = 0; ch < BL_CH; ++ch) { const float wk = wf[ch * w + kp]; q[ch] += (double)u[ch] * (double)wk; mu[ch] += (double)wk * (double)m; }
    }
#pragma unroll
    for (int ch = 0; ch < BL_CH; ++ch) {
        const double qs = block_sum_d(q[ch], red), ms = block_sum_d(mu[ch], red);
        const int c = c0 + ch;
        if (threadIdx.x == 0 && c < C) {
            const double mean = ms / count;
            double var = qs / count - mean * mean;
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
}

// ---- backward, per group of BL_CH output channels: split-K slabs of G0 = dz^T a2 -> dgamma, dbeta, folded coefficients, dW, A.W ----
__global__ __launch_bounds__(256) void bnlin_row_kernel(const float* __restrict__ slabs, int splits, const uint16_t* __restrict__ W,
                                                         const float* __restrict__ gram, const float* __restrict__ m2, const float* __restrict__ s_dz,
                                                         int C, int w, double count, const float* __restrict__ scale, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, float* __restrict__ dW, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, uint16_t* __restrict__ wd1, float* __restrict__ qk) {
    extern __shared__ __attribute__((aligned(16))) float bl_smem[];
    float* wf = bl_smem;                                    // [BL_CH][w]
    float* g0 = wf + BL_CH * w;                             // [BL_CH][w]
    __shared__ double red[4];
    __shared__ float coef[BL_CH][3];                        // A, Kc, Q
    const int c0 = blockIdx.x * BL_CH;
    load_w_rows(W, c0, C, w, wf);
    const size_t slab = (size_t)C * w;
    for (int e = threadIdx.x; e < BL_CH * w; e += 256) {    // fixed-order sum over the split-K slabs (deterministic)
        const int ch = e / w;
        float acc = 0.f;
        if (c0 + ch < C) {
            const float* p = slabs + (size_t)c0 * w + e;
            int sidx = 0;
            for (; sidx + 4 <= splits; sidx += 4) {
                const float v0 = p[(size_t)sidx * slab], v1 = p[(size_t)(sidx + 1) * slab], v2 = p[(size_t)(sidx + 2) * slab], v3 = p[(size_t)(sidx + 3) * slab];
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; sidx < splits; ++sidx) acc += p[(size_t)sidx * slab];
        }
        g0[e] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < BL_CH; ++ch) {                    // T = sum_p dz * raw3 = rowdot(W, G0)
        double t = 0.0;
        for (int k = threadIdx.x; k < w; k += 256) t += (double)wf[ch * w + k] * (double)g0[ch * w + k];
        t = block_sum_d(t, red);
        const int c = c0 + ch;
        if (threadIdx.x == 0) {
            float A = 0.f, Kc = 0.f, Q = 0.f;
            if (c < C) {
                const double s = (double)s_dz[c], iv = (double)invstd[c], mn = (double)mean[c], a = (double)scale[c];
                const double dg = iv * (t - mn * s);                    // sum dz * xhat
                const double qq = a * iv * dg / count;
                A = (float)a; Q = (float)qq; Kc = (float)(qq * mn - a * s / count);
                dgamma[c] = (float)dg;
                dbeta[c] = (float)s;
                qk[c] = Q; qk[C + c] = Kc;
            }
            coef[ch][0] = A; coef[ch][1] = Kc; coef[ch][2] = Q;
        }
    }
    __syncthreads();
    for (int kp = threadIdx.x; kp < w; kp += 256) {         // dW[c][kp] = A G0 + Kc m2 - Q (W G)
        float u[BL_CH];
        gram_column(gram, wf, w, kp, u);
        const float m = m2[kp];
#pragma unroll
        for (int ch = 0; ch < BL_CH; ++ch)
            if (c0 + ch < C) dW[(size_t)(c0 + ch) * w + kp] = coef[ch][0] * g0[ch * w + kp] + coef[ch][1] * m - coef[ch][2] * u[ch];
    }
    if (c0 + BL_CH <= C) {                                  // A.W, transposed into the data-gradient image [w][C]: 8 channels = 16 bytes per k
        for (int k = threadIdx.x; k < w; k += 256) {
            uint32_t o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = pack_bf16x2(coef[2 * t][0] * wf[(2 * t) * w + k], coef[2 * t + 1][0] * wf[(2 * t + 1) * w + k]);
            *reinterpret_cast<uint4*>(wd1 + (size_t)k * C + c0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (int e = threadIdx.x; e < BL_CH * w; e += 256) {
            const int ch = e / w, k = e - ch * w;
            if (c0 + ch < C) wd1[(size_t)k * C + c0 + ch] = f32_to_bf16_bits(coef[ch][0] * wf[e]);
        }
    }
}

// ---- backward: -M = -(W^T diag(Q) W) as the second data-gradient weight image [w][w] (bf16) and bvec = W^T Kc ----
__global__ __launch_bounds__(256) void bnlin_m_kernel(const uint16_t* __restrict__ W, const float* __restrict__ qk, int C, int w,
                                                       uint16_t* __restrict__ wd2, float* __restrict__ bvec) {
    __shared__ float sa[32][33], sb[32][33], sq[32], sk[32];
    const int ti = blockIdx.y * 32, tj = blockIdx.x * 32;           // rows k (ti), columns k' (tj) of M
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // thread -> outputs (2*ty + {0,1}, 2*tx + {0,1})
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float bacc = 0.f;                                               // bvec[ti + threadIdx.x] on the tj == 0 column of blocks
    for (int c0 = 0; c0 < C; c0 += 32) {
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * 32; e += 256) {
            const int cc = e >> 5, kk = e & 31, c = c0 + cc;
            const bool okc = c < C;
            sa[cc][kk] = (okc && ti + kk < w) ? bf16_bits_to_f32(W[(size_t)c * w + ti + kk]) : 0.f;
            sb[cc][kk] = (okc && tj + kk < w) ? bf16_bits_to_f32(W[(size_t)c * w + tj + kk]) : 0.f;
        }
        if (threadIdx.x < 32) { const int c = c0 + threadIdx.x; sq[threadIdx.x] = c < C ? qk[c] : 0.f; sk[threadIdx.x] = c < C ? qk[C + c] : 0.f; }
        __syncthreads();
#pragma unroll 8
        for (int cc = 0; cc < 32; ++cc) {
            const float q = sq[cc];
            const float a0 = sa[cc][2 * ty] * q, a1 = sa[cc][2 * ty + 1] * q, b0 = sb[cc][2 * tx], b1 = sb[cc][2 * tx + 1];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        if (blockIdx.x == 0 && threadIdx.x < 32)
#pragma unroll 8
            for (int cc = 0; cc < 32; ++cc) bacc += sk[cc] * sa[cc][threadIdx.x];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int k = ti + 2 * ty + r, kp = tj + 2 * tx;
        if (k < w && kp + 1 < w) *reinterpret_cast<uint32_t*>(wd2 + (size_t)k * w + kp) = pack_bf16x2(-acc[r][0], -acc[r][1]);
        else if (k < w && kp < w) wd2[(size_t)k * w + kp] = f32_to_bf16_bits(-acc[r][0]);
    }
    if (blockIdx.x == 0 && threadIdx.x < 32 && ti + threadIdx.x < w) bvec[ti + threadIdx.x] = bacc;
}

// ---- launchers ------------------------------------------------------------------------------------------------------
int launch_bnlin_stats(hipStream_t st, const uint16_t* W, const float* gram, const float* m2, int C, int w, double count, const float* gamma,
                       const float* beta, float* rm, float* rv, float momentum, float eps, float* scale, float* shift, float* mean, float* invstd) {
    if (w % 4 != 0 || w > 2048) { set_error("bnlin: input width %d must be a multiple of 4 and <= 2048", w); return DALI_ERR_INVALID; }
    hipLaunchKernelGGL(bnlin_stats_kernel, dim3((C + BL_CH - 1) / BL_CH), dim3(256), (size_t)BL_CH * w * sizeof(float), st, W, gram, m2, C, w, count, gamma, beta,
                       rm, rv, momentum, eps, scale, shift, mean, invstd);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

int launch_bnlin_bwd(hipStream_t st, const float* slabs, int splits, const uint16_t* W, const float* gram, const float* m2, const float* s_dz, int C,
                     int w, double count, const float* scale, const float* mean, const float* invstd, float* dW, float* dgamma, float* dbeta,
                     uint16_t* wd1, uint16_t* wd2, float* bvec, float* qk) {
    if (w % 4 != 0 || w > 2048 || (C & 7)) { set_error("bnlin: width %d must be a multiple of 4 and <= 2048, C %d a multiple of 8", w, C); return DALI_ERR_INVALID; }
    hipLaunchKernelGGL(bnlin_row_kernel, dim3((C + BL_CH - 1) / BL_CH), dim3(256), (size_t)2 * BL_CH * w * sizeof(float), st, slabs, splits, W, gram, m2, s_dz, C, w,
                       count, scale, mean, invstd, dW, dgamma, dbeta, wd1, qk);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(bnlin_m_kernel, dim3((w + 31) / 32, (w + 31) / 32), dim3(256), 0, st, W, qk, C, w, wd2, bvec);
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
                              float* scale, float* shift, float* mean, float* invstd) {
    DALI_REQUIRE(ctx && a && W && gamma && beta && gram && m2 && scale && shift && mean && invstd, "dali_bnlin_fwd: null argument");
    DALI_REQUIRE(P > 0 && C % 8 == 0 && w % 32 == 0, "dali_bnlin_fwd: C %% 8, w %% 32 (C=%d w=%d)", C, w);
    hipStream_t st = (hipStream_t)stream;
    WGradArgs wa{};
    wa.dY = a; wa.X = a; wa.Cm = w; wa.P = P; wa.Ntot = w; wa.g = bl_geom(P, w);
    size_t wsb;
    wgrad_plan(w, w, P, 512, &wa.splits, &wa.pix_per_split, &wsb, 1, 0);
    const size_t b_cs = align_up(colsum_partial_floats(P, w) * 4, 256), b_sc = align_up(reduce_scratch_bytes(w, 1), 256);
    char* ws = static_cast<char*>(workspace(ctx, align_up(wsb, 256) + b_cs + b_sc));
    if (!ws) return DALI_ERR_NOMEM;
    wa.partial = reinterpret_cast<float*>(ws);
    int rc;
    if ((rc = launch_igemm_wgrad(st, wa, gram, 0))) return rc;
    if ((rc = launch_colsum(st, a, P, w, m2, reinterpret_cast<float*>(ws + align_up(wsb, 256)), reinterpret_cast<double*>(ws + align_up(wsb, 256) + b_cs)))) return rc;
    return launch_bnlin_stats(st, W, gram, m2, C, w, (double)P, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd);
}

extern "C" int dali_bnlin_bwd(dali_ctx* ctx, void* stream, const uint16_t* dz, const uint16_t* a, const uint16_t* W, int P, int C, int w,
                              const float* gram, const float* m2, const float* scale, const float* mean, const float* invstd, float* dW,
                              float* dgamma, float* dbeta, uint16_t* wd1, uint16_t* wd2, float* bvec) {
    DALI_REQUIRE(ctx && dz && a && W && gram && m2 && scale && mean && invstd && dW && dgamma && dbeta && wd1 && wd2 && bvec, "dali_bnlin_bwd: null argument");
    DALI_REQUIRE(P > 0 && C % 32 == 0 && w % 32 == 0, "dali_bnlin_bwd: C %% 32, w %% 32 (C=%d w=%d)", C, w);
    hipStream_t st = (hipStream_t)stream;
    WGradArgs wa{};
    wa.dY = dz; wa.X = a; wa.Cm = C; wa.P = P; wa.Ntot = w; wa.g = bl_geom(P, w);
    size_t wsb;
    wgrad_plan(C, w, P, 512, &wa.splits, &wa.pix_per_split, &wsb, 1, 0);
    const size_t b_cs = align_up(colsum_partial_floats(P, C) * 4, 256), b_sc = align_up(reduce_scratch_bytes(C, 1), 256), b_v = align_up((size_t)C * 4, 256);
    char* ws = static_cast<char*>(workspace(ctx, align_up(wsb, 256) + b_cs + b_sc + 3 * b_v));
    if (!ws) return DALI_ERR_NOMEM;
    wa.partial = reinterpret_cast<float*>(ws);
    char* p = ws + align_up(wsb, 256);
    float* cs_partial = reinterpret_cast<float*>(p); p += b_cs;
    double* scratch = reinterpret_cast<double*>(p); p += b_sc;
    float* sdz = reinterpret_cast<float*>(p); p += b_v;
    float* qk = reinterpret_cast<float*>(p);
    int rc;
    if ((rc = launch_colsum(st, dz, P, C, sdz, cs_partial, scratch))) return rc;
    if ((rc = launch_igemm_wgrad(st, wa, nullptr, 0))) return rc;
    return launch_bnlin_bwd(st, wa.partial, wa.splits, W, gram, m2, sdz, C, w, (double)P, scale, mean, invstd, dW, dgamma, dbeta, wd1, wd2, bvec, qk);
}
