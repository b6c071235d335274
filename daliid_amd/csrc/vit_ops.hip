// vit_ops.hip -- kernels of the TransReID ViT encoder (vit_pytorch.py:120-184, 251-288, 375-408) that are not plain
// linear layers: patch extraction, token assembly (cls + pos_embed), LayerNorm, multi-head self-attention
// (softmax(q k^T * hd^-0.5) v with the score matrix kept on chip), column sums for bias gradients.
// Linear layers run on the implicit-GEMM engine of conv.hip (a Linear is a 1x1 convolution over tokens).
// Activations: tokens [B*T][C] bf16; parameters fp32; statistics / reductions fp32.
#include "kernels.h"
#include "reduce_finish.h"

namespace dali {

__device__ __forceinline__ void unpack8v(const uint4& v, float (&f)[8]) {
    f[0] = bf16_bits_to_f32(v.x & 0xffffu); f[1] = bf16_bits_to_f32(v.x >> 16);
    f[2] = bf16_bits_to_f32(v.y & 0xffffu); f[3] = bf16_bits_to_f32(v.y >> 16);
    f[4] = bf16_bits_to_f32(v.z & 0xffffu); f[5] = bf16_bits_to_f32(v.z >> 16);
    f[6] = bf16_bits_to_f32(v.w & 0xffffu); f[7] = bf16_bits_to_f32(v.w >> 16);
}
__device__ __forceinline__ uint4 pack8v(const float (&f)[8]) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

// ------------------------------------------------------------------------------------------------
// PatchEmbed_overlap (vit_pytorch.py:251-288): the ps x ps / stride conv as a GEMM over extracted patches.
// img fp32 [B,3,H,W] -> patches bf16 [B*ny*nx][3*ps*ps], K order (c, r, s) = the conv weight's flattening.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int B, int H, int W, int ps, int stride,
                                                        int ny, int nx, uint16_t* __restrict__ out) {
    const int K = 3 * ps * ps, cpr = K >> 3;                       // ps % 8 == 0
    const size_t total = (size_t)B * ny * nx * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % cpr) * 8;
        const size_t patch = i / cpr;
        const int px = (int)(patch % nx), py = (int)((patch / nx) % ny), b = (int)(patch / ((size_t)nx * ny));
        const int c = k / (ps * ps), rem = k - c * ps * ps, r = rem / ps, s = rem - r * ps;
        const float* src = img + (((size_t)b * 3 + c) * H + py * stride + r) * W + px * stride + s;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = src[t];
        *reinterpret_cast<uint4*>(out + patch * K + k) = pack8v(v);
    }
}

// x[b,0,:] = cls + pos[0];  x[b,1+i,:] = pe[b*np+i,:] + pos[1+i]      (vit_pytorch.py:379-391, no SIE)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const uint16_t* __restrict__ pe, const float* __restrict__ cls,
                                                               const float* __restrict__ pos, int B, int T, int C, uint16_t* __restrict__ x) {
    const int cpr = C >> 3;
    const size_t total = (size_t)B * T * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cpr) * 8;
        const size_t row = i / cpr;
        const int t = (int)(row % T), b = (int)(row / T);
        float v[8];
        if (t == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = cls[c + q];
        } else {
            unpack8v(*reinterpret_cast<const uint4*>(pe + ((size_t)b * (T - 1) + t - 1) * C + c), v);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += pos[(size_t)t * C + c + q];
        *reinterpret_cast<uint4*>(x + row * C + c) = pack8v(v);
    }
}
// dpos[t,:] = sum_b dx[b,t,:] (fp32; dcls = dpos[0]);  dpe[b*np+i,:] = dx[b,1+i,:]
__global__ __launch_bounds__(256) void assemble_tokens_bwd_kernel(const uint16_t* __restrict__ dx, int B, int T, int C,
                                                                   float* __restrict__ dpos, float* __restrict__ dcls, uint16_t* __restrict__ dpe) {
    const int cpr = C >> 3;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T * cpr) return;
    const int c = (i % cpr) * 8, t = i / cpr;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int b = 0; b < B; ++b) {
        const uint4 raw = *reinterpret_cast<const uint4*>(dx + ((size_t)b * T + t) * C + c);
        float v[8];
        unpack8v(raw, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += v[q];
        if (t > 0 && dpe) *reinterpret_cast<uint4*>(dpe + ((size_t)b * (T - 1) + t - 1) * C + c) = raw;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        dpos[(size_t)t * C + c + q] = acc[q];
        if (t == 0) dcls[c + q] = acc[q];
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (C <= 2048, C % 8 == 0), one wave64 per row.  eps 1e-6 in TransReID (vit_pytorch.py:457).
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXCH = 4;        // 16-byte chunks per lane (C <= 2048)

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const uint16_t* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int rows, int C, float eps,
                                                             uint16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ y32) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[LN_MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        const int c = (lane + k * 64) * 8;
        if (c < C) {
            unpack8v(*reinterpret_cast<const uint4*>(x + (size_t)row * C + c), v[k]);
#pragma unroll
            for (int t = 0; t < 8; ++t) s += v[k][t];
        }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        const int c = (lane + k * 64) * 8;
        if (c < C) {
#pragma unroll
            for (int t = 0; t < 8; ++t) { const float d = v[k][t] - mu; q += d * d; }
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        const int c = (lane + k * 64) * 8;
        if (c < C) {
            float o[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) o[t] = (v[k][t] - mu) * rs * gamma[c + t] + beta[c + t];
            if (y) *reinterpret_cast<uint4*>(y + (size_t)row * C + c) = pack8v(o);
            if (y32) {
#pragma unroll
                for (int t = 0; t < 8; ++t) y32[(size_t)row * C + c + t] = o[t];
            }
        }
    }
}

// dx = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat)) (+ add);  per-block partials of
// dgamma = sum_rows g*xhat and dbeta = sum_rows g   ->  partial[block][C][2]
// NCH = 16-byte chunks per lane (C <= 512 NCH): the per-lane arrays are sized for the actual width (ViT-B: 2), not for the 2048 maximum
// (160 of ~200 VGPRs were dead weight and held the kernel to 2 waves per SIMD)
// MINW = waves per SIMD the register allocation must leave room for: at 173 VGPRs (the 256-thread default bound) two workgroups fit a CU and
// the 768-workgroup launch sized for "three per CU in one round" ran as one and a half rounds
template <int NCH, int MINW = 2>
__global__ __launch_bounds__(256, MINW) void layernorm_bwd_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const uint16_t* __restrict__ add,
                                                             int rows, int C, int rows_per_block, uint16_t* __restrict__ dx,
                                                             float* __restrict__ partial, const float* __restrict__ g32) {
    extern __shared__ float red[];                                  // [4 waves][C][2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[NCH][8], db[NCH][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int t = 0; t < 8; ++t) { dg[k][t] = 0.f; db[k][t] = 0.f; }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float gam[NCH][8];                                        // this lane's gamma values: loaded once, not per row
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = (lane + k * 64) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) gam[k][t] = (c < C) ? gamma[c + t] : 0.f;
    }
    // One row per wave and step, TWO rows of requests ahead (register sets A and B, the loop is unrolled by two): a wave walking its 9 rows
    // load -> reduce -> store one after the other exposed one memory round trip per row (56 us for 116 MB); one row ahead still left the
    // kernel latency-bound (8 waves per CU x 4.6 KB in flight = 3.8 TB/s by Little's law, measured 3.4).  Every load of a request is
    // unconditional (lanes past C re-read column 0, an absent operand re-reads x: the values are never used): "if (c < C) { if (!g32) load;
    // load; if (add) load; }" compiled to a branch and a wait around each load, six dependent round trips per row (47 us for 155 MB).
    const uint16_t* gsrc = g32 ? x : g;
    const uint16_t* asrc = add ? add : x;
    struct RowRegs { uint4 g[NCH], x[NCH], a[NCH]; float mu, rs; };
    auto request = [&](RowRegs& r, int row) {
        r.mu = mean[row]; r.rs = rstd[row];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (lane + k * 64) * 8;
            const size_t o = (size_t)row * C + (c < C ? c : 0);
            r.g[k] = *reinterpret_cast<const uint4*>(gsrc + o);
            r.x[k] = *reinterpret_cast<const uint4*>(x + o);
            r.a[k] = *reinterpret_cast<const uint4*>(asrc + o);
        }
    };
    // consumes r (its registers are free for the next request once this returns the unpacked values) and writes the row
    auto process = [&](RowRegs& r, int row, int next_row) {
        const float mu = r.mu, rs = r.rs;
        float gv[NCH][8], xh[NCH][8];
        uint4 ca[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (lane + k * 64) * 8;
            ca[k] = r.a[k];
            if (c < C) {
                float xv[8];
                if (g32) {                                          // fp32 upstream gradient (final LayerNorm under the BN neck)
#pragma unroll
                    for (int t = 0; t < 8; ++t) gv[k][t] = g32[(size_t)row * C + c + t];
                } else {
                    unpack8v(r.g[k], gv[k]);
                }
                unpack8v(r.x[k], xv);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    xh[k][t] = (xv[t] - mu) * rs;
                    dg[k][t] += gv[k][t] * xh[k][t];
                    db[k][t] += gv[k][t];
                    gv[k][t] *= gam[k][t];
                    s1 += gv[k][t];
                    s2 += gv[k][t] * xh[k][t];
                }
            }
        }
        if (next_row < r1) request(r, next_row);
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (lane + k * 64) * 8;
            if (c < C) {
                float o[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) o[t] = rs * (gv[k][t] - s1 - xh[k][t] * s2);
                if (add) {
                    float a[8];
                    unpack8v(ca[k], a);
#pragma unroll
                    for (int t = 0; t < 8; ++t) o[t] += a[t];
                }
                *reinterpret_cast<uint4*>(dx + (size_t)row * C + c) = pack8v(o);
            }
        }
    };
    RowRegs A, B;
    const int first = r0 + wave;
    if (first < r1) request(A, first);
    if (first + 4 < r1) request(B, first + 4);
    for (int row = first; row < r1; row += 8) {
        process(A, row, row + 8);
        if (row + 4 < r1) process(B, row + 4, row + 12);
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = (lane + k * 64) * 8;
        if (c < C) {
#pragma unroll
            for (int t = 0; t < 8; ++t) { red[((size_t)wave * C + c + t) * 2] = dg[k][t]; red[((size_t)wave * C + c + t) * 2 + 1] = db[k][t]; }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * 2; e += 256)
        partial[(size_t)blockIdx.x * C * 2 + e] = red[e] + red[(size_t)C * 2 + e] + red[(size_t)2 * C * 2 + e] + red[(size_t)3 * C * 2 + e];
}

// partial[block][C] = sum over the block's rows of y[row][:]  (bias gradients of the linear layers)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const uint16_t* __restrict__ y, int rows, int C, int rows_per_block,
                                                              float* __restrict__ partial) {
    extern __shared__ float red[];                                  // [rif][C]
    const int cpr = C >> 3, rif = 256 / min(cpr, 256);
    const int col = threadIdx.x % cpr, rsub = threadIdx.x / cpr;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    // C may exceed 2048 (3072-wide MLP): a thread then owns several chunk columns
    for (int cc = col; cc < cpr; cc += 256) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (rsub < rif) {
            for (int row = r0 + rsub; row < r1; row += rif) {
                float v[8];
                unpack8v(*reinterpret_cast<const uint4*>(y + (size_t)row * C + cc * 8), v);
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] += v[t];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) red[(size_t)rsub * C + cc * 8 + t] = acc[t];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C; e += 256) {
        float s = 0.f;
        for (int r = 0; r < rif; ++r) s += red[(size_t)r * C + e];
        partial[(size_t)blockIdx.x * C + e] = s;
    }
}
// ------------------------------------------------------------------------------------------------
// Multi-head self-attention, one block per (batch, head); head_dim 64; T <= 16 * NTILE tokens, NTILE in {13, 14, 16}
// (197 tokens = ViT-B/16 at 224x224; 211 = TransReID's 256x128 at stride 12, vit_pytorch.py:254-267; up to 256).
// qkv [B*T][3C] bf16 (q | k | v, head h at columns h*64 of each third, as vit_pytorch.py:155 lays them out).
// Forward: O = softmax(Q K^T * scale) V, scores never leave the chip; also writes the row log-sum-exp (lse = max + log(sum)) for the
// backward.  The kernels are the "second form" described further down (attention_fwd2 / bwd_dq / bwd_dkv: probabilities stay in registers);
// the first form (P through a per-wave LDS strip, one workgroup per CU: 90 us forward, 316 us backward per layer) was removed in round 4.
// LDS images are plain row-major [Tp][72] bf16 (64 + 8 pad): both the K-contiguous (ds_read_b128) and the transposing
// (tr_b16) fragment reads work on it.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_LD = 72, ATT_HD = 64;
constexpr int ATT_MAX_T = 256;
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));

// A/B fragment with k contiguous in memory: element (row0 + lane&15, k0 + 8*(lane>>4) .. +7)
__device__ __forceinline__ bf16x8_t frag_k(const uint16_t* base, int ld, int row0, int k0, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(base + (row0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4));
}
template <int TP>
__device__ __forceinline__ void att_load_tile(const uint16_t* __restrict__ src, size_t row_stride, int T, uint16_t* dst) {
    // [T][64] bf16 from global (row stride in elements) -> LDS [TP][ATT_LD], rows >= T zeroed
    for (int i = threadIdx.x; i < TP * 8; i += blockDim.x) {
        const int row = i >> 3, ch = i & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < T) v = *reinterpret_cast<const uint4*>(src + (size_t)row * row_stride + ch * 8);
        *reinterpret_cast<uint4*>(dst + row * ATT_LD + ch * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// DropPath (stochastic depth per sample, vit_pytorch.py:45-62): out[b][t][:] = (res ? res : 0) + scale[b] * branch[b][t][:],
// scale[b] in {0, 1/keep_prob} drawn by the caller.  8 bf16 per thread.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowscale_add_kernel(const uint16_t* __restrict__ branch, const float* __restrict__ scale, int rows_per_sample,
                                                            int C, size_t chunks, const uint16_t* __restrict__ res, uint16_t* __restrict__ out) {
    const int cpr = C / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (size_t)gridDim.x * 256) {
        const size_t row = i / cpr;
        const float sc = scale[row / rows_per_sample];
        float v[8], r[8];
        unpack8v(*reinterpret_cast<const uint4*>(branch + i * 8), v);
        if (res) unpack8v(*reinterpret_cast<const uint4*>(res + i * 8), r);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (res ? r[t] : 0.f) + sc * v[t];
        *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
}

// per-sample DropPath factors [n_vec][B] -> per-row factors [n_vec][B * T] (what the linear epilogues index by output row)
__global__ __launch_bounds__(256) void expand_rowscale_kernel(const float* __restrict__ scale, int n_vec, int B, int T, float* __restrict__ out) {
    const size_t total = (size_t)n_vec * B * T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) out[i] = scale[i / T];
}

static inline int vgrid(size_t items, int cap = 16384) {
    size_t b = (items + 255) / 256;
    if (b > (size_t)cap) b = cap;
    return b < 1 ? 1 : (int)b;
}

// ---- launchers ----------------------------------------------------------------------------------------
int launch_patchify(hipStream_t st, const float* img, int B, int H, int W, int ps, int stride, uint16_t* out) {
    const int ny = (H - ps) / stride + 1, nx = (W - ps) / stride + 1;
    hipLaunchKernelGGL(patchify_kernel, dim3(vgrid((size_t)B * ny * nx * (3 * ps * ps / 8))), dim3(256), 0, st, img, B, H, W, ps, stride, ny, nx, out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_assemble_tokens(hipStream_t st, const uint16_t* pe, const float* cls, const float* pos, int B, int T, int C, uint16_t* x) {
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(vgrid((size_t)B * T * (C / 8))), dim3(256), 0, st, pe, cls, pos, B, T, C, x);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_assemble_tokens_bwd(hipStream_t st, const uint16_t* dx, int B, int T, int C, float* dpos, float* dcls, uint16_t* dpe) {
    hipLaunchKernelGGL(assemble_tokens_bwd_kernel, dim3((T * (C / 8) + 255) / 256), dim3(256), 0, st, dx, B, T, C, dpos, dcls, dpe);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_expand_rowscale(hipStream_t st, const float* scale, int n_vec, int B, int T, float* out) {
    hipLaunchKernelGGL(expand_rowscale_kernel, dim3(vgrid((size_t)n_vec * B * T, 2048)), dim3(256), 0, st, scale, n_vec, B, T, out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_rowscale_add(hipStream_t st, const uint16_t* branch, const float* scale, int samples, int rows_per_sample, int C, const uint16_t* res,
                        uint16_t* out) {
    const size_t chunks = (size_t)samples * rows_per_sample * (C / 8);
    hipLaunchKernelGGL(rowscale_add_kernel, dim3(vgrid(chunks, 4096)), dim3(256), 0, st, branch, scale, rows_per_sample, C, chunks, res, out);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_layernorm_fwd(hipStream_t st, const uint16_t* x, const float* gamma, const float* beta, int rows, int C, float eps,
                         uint16_t* y, float* mean, float* rstd, float* y32) {
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, gamma, beta, rows, C, eps, y, mean, rstd, y32);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
// LayerNorm backward: TWO workgroups per CU in one round (the kernel needs ~200 VGPRs = two 4-wave workgroups per CU; the former 768
// "three per CU" ran as one and a half rounds): 25216 rows x 768, with the reduce behind it, 47.5 us at 768 -> 43.3 at 512, 40.4 with two
// rows of requests ahead; three waves per SIMD spill (59.6 us), four 93.7.
static int ln_block_cap() { return 512; }
static int rows_blocks(int rows, int per_iter, int* rpb, int cap = 2048) {
    int blocks = (rows + per_iter * 8 - 1) / (per_iter * 8);
    if (blocks > cap) blocks = cap;                    // column sums: 8 workgroups per CU (512 left too few rows in flight: 26 -> 17 us)
    if (blocks < 1) blocks = 1;
    int r = (rows + blocks - 1) / blocks;
    r = (r + per_iter - 1) / per_iter * per_iter;
    *rpb = r;
    return (rows + r - 1) / r;
}
size_t layernorm_bwd_partial_floats(int rows, int C) { int rpb; return (size_t)rows_blocks(rows, 4, &rpb, 2048) * C * 2; }   // sized for any cap
int launch_layernorm_bwd(hipStream_t st, const uint16_t* g, const uint16_t* x, const float* gamma, const float* mean, const float* rstd,
                         const uint16_t* add, int rows, int C, uint16_t* dx, float* dgamma, float* dbeta, float* partial, double* scratch,
                         const float* g32) {
    int rpb;
    const int blocks = rows_blocks(rows, 4, &rpb, ln_block_cap());
    const size_t lds = (size_t)4 * C * 2 * sizeof(float);
    if (C <= 512) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(blocks), dim3(256), lds, st, g, x, gamma, mean, rstd, add, rows, C, rpb, dx, partial, g32);
    else if (C <= 1024) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), lds, st, g, x, gamma, mean, rstd, add, rows, C, rpb, dx, partial, g32);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<LN_MAXCH>, dim3(blocks), dim3(256), lds, st, g, x, gamma, mean, rstd, add, rows, C, rpb, dx, partial, g32);
    DALI_LAUNCH_CHECK();
    return launch_reduce_finish<2>(st, partial, blocks, C, scratch, FinStore{{dgamma, dbeta, nullptr, nullptr}, 2});
}
size_t colsum_partial_floats(int rows, int C) { int rpb; const int rif = 256 / ((C / 8) < 256 ? (C / 8) : 256); return (size_t)rows_blocks(rows, rif, &rpb) * C; }
// first level only: partial[*n_rows][C] per-block column sums; the caller finishes the sum (bnlin.hip's row kernel)
int launch_colsum_partials(hipStream_t st, const uint16_t* y, int rows, int C, float* partial, int* n_rows) {
    const int cpr = C / 8, rif = 256 / (cpr < 256 ? cpr : 256);
    int rpb;
    const int blocks = rows_blocks(rows, rif, &rpb);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks), dim3(256), (size_t)rif * C * sizeof(float), st, y, rows, C, rpb, partial);
    DALI_LAUNCH_CHECK();
    *n_rows = blocks;
    return DALI_OK;
}
int launch_colsum(hipStream_t st, const uint16_t* y, int rows, int C, float* out, float* partial, double* scratch) {
    const int cpr = C / 8, rif = 256 / (cpr < 256 ? cpr : 256);
    int rpb;
    const int blocks = rows_blocks(rows, rif, &rpb);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks), dim3(256), (size_t)rif * C * sizeof(float), st, y, rows, C, rpb, partial);
    DALI_LAUNCH_CHECK();
    return launch_reduce_finish<1>(st, partial, blocks, C, scratch, FinStore{{out, nullptr, nullptr, nullptr}, 1});
}
// ------------------------------------------------------------------------------------------------
// Attention, second form (round 3): the probabilities never leave the registers, and only the two operands EVERY wave needs live in LDS.
//
// An MFMA sums over 32 "slots" (lane group g = lane >> 4, position s = 0..7); which k index a slot means is free as long as the A and the B
// operand agree.  The accumulator of a 16 x 16 tile leaves lane (g, i) with rows 4g .. 4g+3 of column i -- exactly four slots of lane group g
// of an A operand whose row is i.  So a score tile computed TRANSPOSED (rows = keys, columns = queries: S^T = K Q^T) is, after the softmax,
// already the A operand of P V for query row i, with slots (g, 0..3) = keys 16 j0 + 4g .. +3 of one key tile and slots (g, 4..7) = the same
// keys of a second tile; the B operand takes the matching rows of V by two transposing reads (frag_tr2).  No per-wave strip in LDS, no
// write / barrier / re-read of P (the first form spent more on those than on its MFMAs: 93 us forward, 317 us backward per layer for
// 5 us and 11 us of matrix work per workgroup).  The same trick gives dQ = dS K from dS^T, and dV = P^T dO, dK = dS^T Q from the
// UN-transposed tiles S = Q K^T, so the backward is two kernels: one workgroup per (batch, head) each, the query (key) tile a wave owns is
// loaded straight from global memory as fragments, K and V (Q and dO) for all tiles sit in LDS: 60 KB, two workgroups per CU, so one loads
// while the other computes.  S and dP are computed once per kernel (the first form computed S three times and dP twice).
// ------------------------------------------------------------------------------------------------
// 4 + 4 rows of one column per lane: rows rowA + 4g + (0..3) and rowB + 4g + (0..3), column n0 + (lane & 15)
__device__ __forceinline__ bf16x8_t frag_tr2(const uint16_t* base, int ld, int rowA, int rowB, int n0, int lane) {
    typedef __attribute__((address_space(3))) s16x4v* lp;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + (rowA + 4 * g + q) * ld + n0 + 4 * p));
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + (rowB + 4 * g + q) * ld + n0 + 4 * p));
    const s16x8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
// K-contiguous fragment straight from global memory: element (row0 + lane&15, k0 + 8*(lane>>4) .. +7), rows >= T are zero
__device__ __forceinline__ bf16x8_t frag_g(const uint16_t* __restrict__ src, size_t row_stride, int row0, int k0, int T, int lane) {
    const int row = row0 + (lane & 15);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < T) v = *reinterpret_cast<const uint4*>(src + (size_t)row * row_stride + k0 + 8 * (lane >> 4));
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t pack_frag(const float (&a)[4], const float (&b)[4]) {
    const uint4 v = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3]));
    return __builtin_bit_cast(bf16x8_t, v);
}

// A 16 x 16 fp32 accumulator (lane (g, i): rows 4g .. 4g+3 of column i) leaves its lane as FOUR CONSECUTIVE COLUMNS of one row: a 4 x 4 transpose inside
// every quad of lanes (two butterfly stages of quad permutes: 4 DPP moves + 12 selects), then one 8-byte store per lane -- row 4g + (i & 3), columns
// 4 (i >> 2) .. + 3 -- where the accumulator layout itself only allows 2-byte stores (16 `global_store_short` per lane and tile in the attention kernels).
__device__ __forceinline__ void acc_row_store(const f32x4_t& a, uint16_t* row0_ptr, size_t row_stride, int rows_left, int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    float v[4] = {a[0], a[1], a[2], a[3]}, u[4];
#pragma unroll
    for (int p = 0; p < 4; p += 2) {                     // stage 1: lane bit 0 <-> register bit 0
        const float send = b0 ? v[p] : v[p + 1];
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xb1, 0xf, 0xf, false));   // quad_perm [1, 0, 3, 2]
        u[p] = b0 ? recv : v[p];
        u[p + 1] = b0 ? v[p + 1] : recv;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {                        // stage 2: lane bit 1 <-> register bit 1
        const float send = b1 ? u[p] : u[p + 2];
        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4e, 0xf, 0xf, false));   // quad_perm [2, 3, 0, 1]
        v[p] = b1 ? recv : u[p];
        v[p + 2] = b1 ? u[p + 2] : recv;
    }
    const int r = (lane >> 4) * 4 + (lane & 3);          // this lane's row inside the 16-row block; v[0..3] = columns 4 ((lane & 15) >> 2) ..
    if (r < rows_left)
        *reinterpret_cast<uint2*>(row0_ptr + (size_t)r * row_stride + 4 * ((lane & 15) >> 2)) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
template <int NTILE, int NW>
__global__ __launch_bounds__(NW * 64, 4) void attention_fwd2_kernel(const uint16_t* __restrict__ qkv, int B, int T, int H, float scale,
                                                                    uint16_t* __restrict__ out, float* __restrict__ lse) {
    constexpr int TP = NTILE * 16, NPAIR = (NTILE + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    uint16_t* sK = sm;
    uint16_t* sV = sK + TP * ATT_LD;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * ATT_HD, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t rs = (size_t)3 * C;
    const uint16_t* base = qkv + (size_t)b * T * rs + h * ATT_HD;
    att_load_tile<TP>(base + C, rs, T, sK);
    att_load_tile<TP>(base + 2 * C, rs, T, sV);
    __syncthreads();
    const float sc2 = scale * 1.44269504088896f;                     // exp(x) = 2^(x log2 e)
    for (int qt = wave; qt < NTILE; qt += NW) {
        if (qt * 16 >= T) break;
        const bf16x8_t qa0 = frag_g(base, rs, qt * 16, 0, T, lane), qa1 = frag_g(base, rs, qt * 16, 32, T, lane);
        f32x4_t s[NTILE];
        float m = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < NTILE; ++j) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sK, ATT_LD, j * 16, 0, lane), qa0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sK, ATT_LD, j * 16, 32, lane), qa1, a, 0, 0, 0);
            const int key0 = j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (key0 + r < T) ? a[r] * sc2 : -__builtin_inff(); m = fmaxf(m, a[r]); }
            s[j] = a;
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < NTILE; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[j][r] = __builtin_amdgcn_exp2f(s[j][r] - m); l += s[j][r]; }
        l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
        const float inv_l = 1.0f / l;
        if (lane < 16) {
            const int row = qt * 16 + lane;
            if (row < T && lse) lse[(size_t)bh * T + row] = (m + log2f(l)) * 0.6931471805599453f;     // natural-log lse = max + log(sum)
        }
        f32x4_t o[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int jp = 0; jp < NPAIR; ++jp) {
            const int j0 = 2 * jp, j1 = 2 * jp + 1 < NTILE ? 2 * jp + 1 : j0;       // odd tile count: the last pair's second half is zero
            float pa[4], pb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pa[r] = s[j0][r] * inv_l; pb[r] = 2 * jp + 1 < NTILE ? s[j1][r] * inv_l : 0.f; }
            const bf16x8_t pf = pack_frag(pa, pb);
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, frag_tr2(sV, ATT_LD, j0 * 16, j1 * 16, d * 16, lane), o[d], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)                          // (acc_row_store here: 51.4 -> 52.1 us; it pays in the backward kernels, three outputs per tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = qt * 16 + (lane >> 4) * 4 + r;
                if (row < T) out[((size_t)b * T + row) * C + h * ATT_HD + d * 16 + (lane & 15)] = f32_to_bf16_bits(o[d][r]);
            }
    }
}

// dQ: one wave per query tile, K and V of the head in LDS.  dS^T = P^T (dP^T - D) scale with P^T = exp(S^T scale - lse), dQ = dS K.
template <int NTILE, int NW>
__global__ __launch_bounds__(NW * 64, 4) void attention_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o,
                                                                      const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                      int B, int T, int H, float scale, uint16_t* __restrict__ dqkv) {
    constexpr int TP = NTILE * 16, NPAIR = (NTILE + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    uint16_t* sK = sm;
    uint16_t* sV = sK + TP * ATT_LD;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * ATT_HD, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t rs = (size_t)3 * C;
    const uint16_t* base = qkv + (size_t)b * T * rs + h * ATT_HD;
    const uint16_t* go = d_o + (size_t)b * T * C + h * ATT_HD;
    const uint16_t* oo = o + (size_t)b * T * C + h * ATT_HD;
    att_load_tile<TP>(base + C, rs, T, sK);
    att_load_tile<TP>(base + 2 * C, rs, T, sV);
    __syncthreads();
    const float sc2 = scale * 1.44269504088896f;
    uint16_t* dq_base = dqkv + (size_t)b * T * rs + h * ATT_HD;
    for (int qt = wave; qt < NTILE; qt += NW) {
        if (qt * 16 >= T) break;
        const bf16x8_t qa0 = frag_g(base, rs, qt * 16, 0, T, lane), qa1 = frag_g(base, rs, qt * 16, 32, T, lane);
        const bf16x8_t ga0 = frag_g(go, C, qt * 16, 0, T, lane), ga1 = frag_g(go, C, qt * 16, 32, T, lane);
        // D_i = sum_d dO[i][d] O[i][d]: this lane's 16 of the 64 products (the d it holds of dO), then over the four lane groups
        float di = 0.f;
        {
            const bf16x8_t oa0 = frag_g(oo, C, qt * 16, 0, T, lane), oa1 = frag_g(oo, C, qt * 16, 32, T, lane);
            float a8[8], b8[8];
            unpack8v(__builtin_bit_cast(uint4, ga0), a8); unpack8v(__builtin_bit_cast(uint4, oa0), b8);
#pragma unroll
            for (int t = 0; t < 8; ++t) di += a8[t] * b8[t];
            unpack8v(__builtin_bit_cast(uint4, ga1), a8); unpack8v(__builtin_bit_cast(uint4, oa1), b8);
#pragma unroll
            for (int t = 0; t < 8; ++t) di += a8[t] * b8[t];
            di += __shfl_xor(di, 16, 64); di += __shfl_xor(di, 32, 64);
        }
        const int qrow = qt * 16 + (lane & 15);
        const float lq = (qrow < T ? lse[(size_t)bh * T + qrow] : 0.f) * 1.44269504088896f;
        f32x4_t dq[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int jp = 0; jp < NPAIR; ++jp) {
            float v[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int j = 2 * jp + hh;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[hh][r] = 0.f;
                if (j < NTILE) {
                    f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sK, ATT_LD, j * 16, 0, lane), qa0, sc, 0, 0, 0);
                    sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sK, ATT_LD, j * 16, 32, lane), qa1, sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sV, ATT_LD, j * 16, 0, lane), ga0, dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sV, ATT_LD, j * 16, 32, lane), ga1, dp, 0, 0, 0);
                    const int key0 = j * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = (key0 + r < T) ? __builtin_amdgcn_exp2f(sc[r] * sc2 - lq) : 0.f;
                        v[hh][r] = p * (dp[r] - di) * scale;
                    }
                }
            }
            const int j0 = 2 * jp, j1 = 2 * jp + 1 < NTILE ? 2 * jp + 1 : j0;
            const bf16x8_t da = pack_frag(v[0], v[1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, frag_tr2(sK, ATT_LD, j0 * 16, j1 * 16, d * 16, lane), dq[d], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) acc_row_store(dq[d], dq_base + (size_t)(qt * 16) * rs + d * 16, rs, T - qt * 16, lane);
    }
}

// dK, dV: one wave per key tile, Q and dO of the head (and lse, D) in LDS.  dV = P^T dO, dK = dS^T Q from the un-transposed tiles S = Q K^T.
template <int NTILE, int NW>
__global__ __launch_bounds__(NW * 64, 4) void attention_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o,
                                                                       const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                       int B, int T, int H, float scale, uint16_t* __restrict__ dqkv) {
    constexpr int TP = NTILE * 16, NPAIR = (NTILE + 1) / 2, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    uint16_t* sQ = sm;
    uint16_t* sD = sQ + TP * ATT_LD;
    float* sLse = reinterpret_cast<float*>(sD + TP * ATT_LD);      // [TP], already times log2 e
    float* sDi = sLse + TP;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * ATT_HD, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t rs = (size_t)3 * C;
    const uint16_t* base = qkv + (size_t)b * T * rs + h * ATT_HD;
    const uint16_t* go = d_o + (size_t)b * T * C + h * ATT_HD;
    att_load_tile<TP>(base, rs, T, sQ);
    att_load_tile<TP>(go, C, T, sD);
    for (int i0 = 0; i0 < TP; i0 += NT / 8) {                        // D_i: 8 lanes x 16 bytes per row
        const int i = i0 + (threadIdx.x >> 3), ch = threadIdx.x & 7;
        float acc = 0.f;
        if (i < T) {
            float ov[8], gv[8];
            unpack8v(*reinterpret_cast<const uint4*>(o + ((size_t)b * T + i) * C + h * ATT_HD + ch * 8), ov);
            unpack8v(*reinterpret_cast<const uint4*>(go + (size_t)i * C + ch * 8), gv);
#pragma unroll
            for (int d = 0; d < 8; ++d) acc += ov[d] * gv[d];
        }
        acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
        if (ch == 0 && i < TP) sDi[i] = acc;
    }
    for (int i = threadIdx.x; i < TP; i += NT) sLse[i] = (i < T) ? lse[(size_t)bh * T + i] * 1.44269504088896f : 0.f;
    __syncthreads();
    const float sc2 = scale * 1.44269504088896f;
    uint16_t* dq_base = dqkv + (size_t)b * T * rs + h * ATT_HD;
    for (int kt = wave; kt < NTILE; kt += NW) {
        if (kt * 16 >= T) break;
        const bf16x8_t ka0 = frag_g(base + C, rs, kt * 16, 0, T, lane), ka1 = frag_g(base + C, rs, kt * 16, 32, T, lane);
        const bf16x8_t va0 = frag_g(base + 2 * C, rs, kt * 16, 0, T, lane), va1 = frag_g(base + 2 * C, rs, kt * 16, 32, T, lane);
        f32x4_t dk[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4_t dv[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const bool key_ok = kt * 16 + (lane & 15) < T;
#pragma unroll
        for (int jp = 0; jp < NPAIR; ++jp) {
            float pv[2][4], dsv[2][4];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int j = 2 * jp + hh;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pv[hh][r] = 0.f; dsv[hh][r] = 0.f; }
                if (j < NTILE) {
                    // rows = queries of tile j, columns = this tile's keys: lane (g, i) holds queries 16 j + 4g .. +3 of key i
                    f32x4_t st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sQ, ATT_LD, j * 16, 0, lane), ka0, st, 0, 0, 0);
                    st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sQ, ATT_LD, j * 16, 32, lane), ka1, st, 0, 0, 0);
                    dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sD, ATT_LD, j * 16, 0, lane), va0, dpt, 0, 0, 0);
                    dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_k(sD, ATT_LD, j * 16, 32, lane), va1, dpt, 0, 0, 0);
                    const int q0 = j * 16 + (lane >> 4) * 4;
                    const float4 lq4 = *reinterpret_cast<const float4*>(sLse + q0), dq4 = *reinterpret_cast<const float4*>(sDi + q0);
                    const float lqv[4] = {lq4.x, lq4.y, lq4.z, lq4.w}, dqv[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = (key_ok && q0 + r < T) ? __builtin_amdgcn_exp2f(st[r] * sc2 - lqv[r]) : 0.f;
                        pv[hh][r] = p;
                        dsv[hh][r] = p * (dpt[r] - dqv[r]) * scale;
                    }
                }
            }
            const int j0 = 2 * jp, j1 = 2 * jp + 1 < NTILE ? 2 * jp + 1 : j0;
            const bf16x8_t pf = pack_frag(pv[0], pv[1]), df = pack_frag(dsv[0], dsv[1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, frag_tr2(sD, ATT_LD, j0 * 16, j1 * 16, d * 16, lane), dv[d], 0, 0, 0);
                dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, frag_tr2(sQ, ATT_LD, j0 * 16, j1 * 16, d * 16, lane), dk[d], 0, 0, 0);
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            acc_row_store(dk[d], dq_base + (size_t)(kt * 16) * rs + C + d * 16, rs, T - kt * 16, lane);
            acc_row_store(dv[d], dq_base + (size_t)(kt * 16) * rs + 2 * C + d * 16, rs, T - kt * 16, lane);
        }
    }
}
template <int NTILE> constexpr size_t att2_lds() { return (size_t)2 * NTILE * 16 * ATT_LD * 2 + 2 * NTILE * 16 * 4; }
static_assert(2 * att2_lds<16>() <= 163840, "two attention workgroups per CU");

template <int NTILE, int NW>
static int att2_fwd_launch(hipStream_t st, const uint16_t* qkv, int B, int T, int H, float scale, uint16_t* out, float* lse) {
    constexpr size_t lds = att2_lds<NTILE>();
    DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_fwd2_kernel<NTILE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((attention_fwd2_kernel<NTILE, NW>), dim3(B * H), dim3(NW * 64), lds, st, qkv, B, T, H, scale, out, lse);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
template <int NTILE, int NW>
static int att2_bwd_launch(hipStream_t st, const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, int B, int T, int H,
                           float scale, uint16_t* dqkv) {
    constexpr size_t lds = att2_lds<NTILE>();
    DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_dq_kernel<NTILE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DALI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_dkv_kernel<NTILE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((attention_bwd_dq_kernel<NTILE, NW>), dim3(B * H), dim3(NW * 64), lds, st, qkv, o, d_o, lse, B, T, H, scale, dqkv);
    hipLaunchKernelGGL((attention_bwd_dkv_kernel<NTILE, NW>), dim3(B * H), dim3(NW * 64), lds, st, qkv, o, d_o, lse, B, T, H, scale, dqkv);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
int launch_attention_fwd(hipStream_t st, const uint16_t* qkv, int B, int T, int H, float scale, uint16_t* out, float* lse) {
    if (T <= 208) return att2_fwd_launch<13, 7>(st, qkv, B, T, H, scale, out, lse);
    if (T <= 224) return att2_fwd_launch<14, 7>(st, qkv, B, T, H, scale, out, lse);
    if (T <= 256) return att2_fwd_launch<16, 8>(st, qkv, B, T, H, scale, out, lse);
    set_error("attention: %d tokens exceed the limit of %d", T, ATT_MAX_T);
    return DALI_ERR_LIMIT;
}
int launch_attention_bwd(hipStream_t st, const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, int B, int T, int H,
                         float scale, uint16_t* dqkv) {
    if (T <= 208) return att2_bwd_launch<13, 7>(st, qkv, o, d_o, lse, B, T, H, scale, dqkv);
    if (T <= 224) return att2_bwd_launch<14, 7>(st, qkv, o, d_o, lse, B, T, H, scale, dqkv);
    if (T <= 256) return att2_bwd_launch<16, 8>(st, qkv, o, d_o, lse, B, T, H, scale, dqkv);
    set_error("attention: %d tokens exceed the limit of %d", T, ATT_MAX_T);
    return DALI_ERR_LIMIT;
}

}  // namespace dali

// ---- single-op C ABI ----------------------------------------------------------------------------------
using namespace dali;

extern "C" int dali_vit_patchify(dali_ctx* ctx, void* stream, const float* img, int B, int H, int W, int patch, int stride, uint16_t* out) {
    DALI_REQUIRE(ctx && img && out, "dali_vit_patchify: null argument");
    DALI_REQUIRE(patch % 8 == 0 && stride > 0 && H >= patch && W >= patch, "dali_vit_patchify: bad geometry");
    return launch_patchify((hipStream_t)stream, img, B, H, W, patch, stride, out);
}
extern "C" int dali_vit_assemble_tokens(dali_ctx* ctx, void* stream, const uint16_t* patch_emb, const float* cls, const float* pos, int B, int T,
                                        int C, uint16_t* x) {
    DALI_REQUIRE(ctx && patch_emb && cls && pos && x && C % 8 == 0, "dali_vit_assemble_tokens: bad argument");
    return launch_assemble_tokens((hipStream_t)stream, patch_emb, cls, pos, B, T, C, x);
}
extern "C" int dali_vit_assemble_tokens_bwd(dali_ctx* ctx, void* stream, const uint16_t* dx, int B, int T, int C, float* dpos, float* dcls,
                                            uint16_t* dpatch_emb) {
    DALI_REQUIRE(ctx && dx && dpos && dcls && C % 8 == 0, "dali_vit_assemble_tokens_bwd: bad argument");
    return launch_assemble_tokens_bwd((hipStream_t)stream, dx, B, T, C, dpos, dcls, dpatch_emb);
}
extern "C" int dali_layernorm_fwd(dali_ctx* ctx, void* stream, const uint16_t* x, const float* gamma, const float* beta, int rows, int C,
                                  float eps, uint16_t* y, float* mean, float* rstd) {
    DALI_REQUIRE(ctx && x && gamma && beta && y && mean && rstd, "dali_layernorm_fwd: null argument");
    DALI_REQUIRE(C % 8 == 0 && C <= 2048 && rows > 0, "dali_layernorm_fwd: C must be a multiple of 8 and <= 2048 (C=%d)", C);
    return launch_layernorm_fwd((hipStream_t)stream, x, gamma, beta, rows, C, eps, y, mean, rstd, nullptr);
}
extern "C" int dali_layernorm_bwd(dali_ctx* ctx, void* stream, const uint16_t* g, const uint16_t* x, const float* gamma, const float* mean,
                                  const float* rstd, const uint16_t* add, int rows, int C, uint16_t* dx, float* dgamma, float* dbeta) {
    DALI_REQUIRE(ctx && g && x && gamma && mean && rstd && dx && dgamma && dbeta, "dali_layernorm_bwd: null argument");
    DALI_REQUIRE(C % 8 == 0 && C <= 2048 && rows > 0, "dali_layernorm_bwd: C must be a multiple of 8 and <= 2048 (C=%d)", C);
    const size_t part = align_up(layernorm_bwd_partial_floats(rows, C) * 4, 256);
    char* ws = static_cast<char*>(workspace(ctx, part + reduce_scratch_bytes(C, 2)));
    if (!ws) return DALI_ERR_NOMEM;
    return launch_layernorm_bwd((hipStream_t)stream, g, x, gamma, mean, rstd, add, rows, C, dx, dgamma, dbeta, reinterpret_cast<float*>(ws),
                                reinterpret_cast<double*>(ws + part), nullptr);
}
extern "C" int dali_attention_fwd(dali_ctx* ctx, void* stream, const uint16_t* qkv, int B, int T, int H, int head_dim, float scale,
                                  uint16_t* out, float* lse) {
    DALI_REQUIRE(ctx && qkv && out, "dali_attention_fwd: null argument");
    DALI_REQUIRE(head_dim == ATT_HD && T > 0 && T <= ATT_MAX_T, "dali_attention_fwd: head_dim must be %d and T <= %d (got %d, %d)", ATT_HD, ATT_MAX_T, head_dim, T);
    return launch_attention_fwd((hipStream_t)stream, qkv, B, T, H, scale, out, lse);
}
extern "C" int dali_attention_bwd(dali_ctx* ctx, void* stream, const uint16_t* qkv, const uint16_t* out, const uint16_t* d_out, const float* lse,
                                  int B, int T, int H, int head_dim, float scale, uint16_t* dqkv) {
    DALI_REQUIRE(ctx && qkv && out && d_out && lse && dqkv, "dali_attention_bwd: null argument");
    DALI_REQUIRE(head_dim == ATT_HD && T > 0 && T <= ATT_MAX_T, "dali_attention_bwd: head_dim must be %d and T <= %d", ATT_HD, ATT_MAX_T);
    return launch_attention_bwd((hipStream_t)stream, qkv, out, d_out, lse, B, T, H, scale, dqkv);
}
