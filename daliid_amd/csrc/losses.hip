// losses.hip -- the two live loss heads of the trainer (train_encodersKIT.py:200-208), fp32:
//   BatchWeightedCenterLoss (losses.py:39-88)  -- distortion-weighted softmax over class centers
//   BatchWeightedProxyLoss  (losses.py:273-341) -- per row: positives = proxies of the row's id, negatives = the k
//                                                  most similar other-id proxies with k = #positives
// The similarity matrices S = fn @ C^T / fn @ P^T come from the split-bf16 MFMA GEMM (dali_pairdist, metric DOT);
// these kernels do the row-wise softmax / top-k arithmetic and the gradients wrt S (center: dense; proxy: <= 2k
// entries per row, applied directly to the proxy rows).  Both losses divide by a batch-global normaliser
// (sum_i w_i * #matches_i, resp. sum of w over rows with a positive): the forward returns the LOCAL numerator and
// denominator so that data-parallel ranks can all-reduce the two scalars before the backward.
#include "common.h"

namespace dali {

// ------------------------------------------------------------------------------------------------
// center head, one wave64 per row.
//   p_ij = softmax_j(S_ij / tau);  num_i = w_i * sum_{j: cl_j == y_i} -log p_ij;  den_i = w_i * #{j: cl_j == y_i}
//   (the reference evaluates exp/sum without max-subtraction, losses.py:62-64; subtracting the row max is the same
//    quantity, evaluated stably)
// rowstat[i] = {num_i, den_i, argmax_j, max_j p_ij}
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void center_rows_kernel(const float* __restrict__ S, const int32_t* __restrict__ y,
                                                           const int32_t* __restrict__ cl, const float* __restrict__ w, float inv_tau,
                                                           int nb, int NC, float* __restrict__ rowstat) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nb) return;
    const float* s = S + (size_t)i * NC;
    float m = -__builtin_inff();
    int am = 0;
#pragma unroll 4
    for (int j = lane; j < NC; j += 64) { const float v = s[j] * inv_tau; if (v > m) { m = v; am = j; } }
    // wave arg-max (first index on ties)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float mo = __shfl_xor(m, o, 64);
        const int ao = __shfl_xor(am, o, 64);
        if (mo > m || (mo == m && ao < am)) { m = mo; am = ao; }
    }
    float se = 0.f, pos = 0.f;
    int cnt = 0;
    const int yi = y[i];
#pragma unroll 4
    for (int j = lane; j < NC; j += 64) {
        const float v = s[j] * inv_tau;
        se += expf(v - m);
        if (cl[j] == yi) { pos += v; ++cnt; }
    }
    se = wave_sum(se); pos = wave_sum(pos); cnt = wave_sum_i(cnt);
    if (lane == 0) {
        const float logz = m + logf(se);
        const float wi = w[i];
        rowstat[i * 4 + 0] = wi * ((float)cnt * logz - pos);
        rowstat[i * 4 + 1] = wi * (float)cnt;
        rowstat[i * 4 + 2] = (float)am;
        rowstat[i * 4 + 3] = 1.0f / se;                     // exp(m - logz)
    }
}

// sums[0] = sum_i rowstat[i][0], sums[1] = sum_i rowstat[i][1]  (single block, fixed order)
__global__ __launch_bounds__(256) void rowstat_reduce_kernel(const float* __restrict__ rowstat, int nb, int stride, float* __restrict__ sums) {
    __shared__ double a[256], b[256];
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) { s0 += (double)rowstat[i * stride]; s1 += (double)rowstat[i * stride + 1]; }
    a[threadIdx.x] = s0; b[threadIdx.x] = s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { a[threadIdx.x] += a[threadIdx.x + o]; b[threadIdx.x] += b[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[0] = (float)a[0]; sums[1] = (float)b[0]; }
}

// dS_ij = gscale * w_i / (tau * Z) * (n_i * p_ij - [cl_j == y_i]),  Z = *denom (global normaliser)
__global__ __launch_bounds__(256) void center_bwd_kernel(const float* __restrict__ S, const int32_t* __restrict__ y,
                                                          const int32_t* __restrict__ cl, const float* __restrict__ w, float inv_tau,
                                                          int nb, int NC, const float* __restrict__ denom, float gscale,
                                                          float* __restrict__ dS) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nb) return;
    const float* s = S + (size_t)i * NC;
    float m = -__builtin_inff();
    for (int j = lane; j < NC; j += 64) m = fmaxf(m, s[j] * inv_tau);
    m = wave_max(m);
    float se = 0.f;
    int cnt = 0;
    const int yi = y[i];
    for (int j = lane; j < NC; j += 64) { se += expf(s[j] * inv_tau - m); cnt += (cl[j] == yi); }
    se = wave_sum(se); cnt = wave_sum_i(cnt);
    const float coef = gscale * w[i] * inv_tau / denom[0];
    const float inv_se = 1.0f / se;
    for (int j = lane; j < NC; j += 64) {
        const float p = expf(s[j] * inv_tau - m) * inv_se;
        dS[(size_t)i * NC + j] = coef * ((float)cnt * p - (cl[j] == yi ? 1.f : 0.f));
    }
}

// ------------------------------------------------------------------------------------------------
// proxy head, one 256-thread block per row.
//   pos = {j : pl_j == y_i}, n = |pos| (<= PROXY_KMAX), neg = top-n of {S_ij : pl_j != y_i}
//   D = sum_pos e^{s/tau} + sum_neg e^{s/tau};  row = -w_i/n * sum_pos (s_p/tau - log D);  valid_i = n > 0
//   d row / d s_p = w_i/tau * (e_p/D - 1/n),   d row / d s_neg = w_i/tau * e_neg/D
// sel_idx[i][0..2K) / sel_coef: selected proxy indices (positives then negatives, -1 padded) and d row/d s.
// Ties among equal negative similarities are broken by ascending index (torch.topk leaves them unspecified).
// ------------------------------------------------------------------------------------------------
constexpr int PROXY_KMAX = 16;

// Rows of up to 256 * PROXY_RV proxies are held in registers (one read of S and of the labels; every round of the top-k selection and the
// positives' values came from global memory before: ~10 dependent passes over the row, 61 us per launch at 256 x 2253); longer rows take the
// same code with the values re-read per round.
constexpr int PROXY_RV = 16;

__global__ __launch_bounds__(256) void proxy_rows_kernel(const float* __restrict__ S, const int32_t* __restrict__ y,
                                                          const int32_t* __restrict__ pl, const float* __restrict__ w, float inv_tau,
                                                          int nb, int NP, float* __restrict__ rowstat, int32_t* __restrict__ sel_idx,
                                                          float* __restrict__ sel_coef, int32_t* __restrict__ status) {
    __shared__ int s_pos[PROXY_KMAX];
    __shared__ float s_posv[PROXY_KMAX];
    __shared__ int s_npos;
    __shared__ float s_rv[4];
    __shared__ int s_ri[4];
    __shared__ float s_negv[PROXY_KMAX];
    __shared__ int s_negi[PROXY_KMAX];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = S + (size_t)i * NP;
    const int yi = y[i];
    const bool in_regs = NP <= 256 * PROXY_RV;
    if (tid == 0) s_npos = 0;
    __syncthreads();
    // negatives' values of this thread (entries tid, tid + 256, ...); positives and entries past the row are -inf: never picked, since at most
    // min(n, NP - n) negatives are asked for
    float rv[PROXY_RV];
    if (in_regs) {
        float val[PROXY_RV];
        int lab[PROXY_RV];
#pragma unroll
        for (int u = 0; u < PROXY_RV; ++u) {
            const int jc = min(tid + 256 * u, NP - 1);              // clamped: unconditional loads, all in flight together
            val[u] = s[jc]; lab[u] = pl[jc];
        }
#pragma unroll
        for (int u = 0; u < PROXY_RV; ++u) {
            const int j = tid + 256 * u;
            const bool pos = j < NP && lab[u] == yi;
            if (pos) {
                const int slot = atomicAdd(&s_npos, 1);
                if (slot < PROXY_KMAX) { s_pos[slot] = j; s_posv[slot] = val[u]; }
            }
            rv[u] = (j < NP && !pos) ? val[u] : -__builtin_inff();
        }
    } else {
        for (int j = tid; j < NP; j += 256) {
            if (pl[j] == yi) {
                const int slot = atomicAdd(&s_npos, 1);
                if (slot < PROXY_KMAX) { s_pos[slot] = j; s_posv[slot] = s[j]; }
            }
        }
    }
    __syncthreads();
    int n = s_npos;
    if (n > PROXY_KMAX) { if (tid == 0) atomicMax(status, 1); n = PROXY_KMAX; }
    int32_t* si = sel_idx + (size_t)i * 2 * PROXY_KMAX;
    float* sc = sel_coef + (size_t)i * 2 * PROXY_KMAX;
    if (n == 0) {
        if (tid < 2 * PROXY_KMAX) { si[tid] = -1; sc[tid] = 0.f; }
        if (tid == 0) { rowstat[i * 2] = 0.f; rowstat[i * 2 + 1] = 0.f; }
        return;
    }
    // deterministic order of the positives (ascending index): tiny insertion sort by one thread
    if (tid == 0) {
        for (int a = 1; a < n; ++a) {
            const int v = s_pos[a]; const float f = s_posv[a];
            int b = a - 1;
            while (b >= 0 && s_pos[b] > v) { s_pos[b + 1] = s_pos[b]; s_posv[b + 1] = s_posv[b]; --b; }
            s_pos[b + 1] = v; s_posv[b + 1] = f;
        }
    }
    __syncthreads();
    // top-n negatives: n rounds of block arg-max over keys strictly below the previous pick ((value desc, index asc) order)
    float pv = __builtin_inff();
    int pi = -1;
    const int nneg_total = NP - n;
    const int k = min(n, nneg_total);
    for (int t = 0; t < k; ++t) {
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < PROXY_RV; ++u) {
                const int j = tid + 256 * u;
                const float v = rv[u];
                const bool below_prev = (v < pv) || (v == pv && j > pi);
                if (v > -__builtin_inff() && below_prev && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
            }
        } else {
            for (int j = tid; j < NP; j += 256) {
                if (pl[j] == yi) continue;
                const float v = s[j];
                const bool below_prev = (v < pv) || (v == pv && j > pi);
                if (below_prev && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float vo = __shfl_xor(bv, o, 64);
            const int io = __shfl_xor(bi, o, 64);
            if (vo > bv || (vo == bv && io < bi)) { bv = vo; bi = io; }
        }
        if (lane == 0) { s_rv[wave] = bv; s_ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < 4; ++q)
                if (s_rv[q] > s_rv[0] || (s_rv[q] == s_rv[0] && s_ri[q] < s_ri[0])) { s_rv[0] = s_rv[q]; s_ri[0] = s_ri[q]; }
            s_negv[t] = s_rv[0]; s_negi[t] = s_ri[0];
        }
        __syncthreads();
        pv = s_negv[t]; pi = s_negi[t];
        __syncthreads();
    }
    if (tid == 0) {
        // exponentials relative to the max selected value (stable); everything is <= 2*PROXY_KMAX terms
        float m = -__builtin_inff();
        for (int a = 0; a < n; ++a) m = fmaxf(m, s_posv[a] * inv_tau);
        for (int a = 0; a < k; ++a) m = fmaxf(m, s_negv[a] * inv_tau);
        float D = 0.f, possum = 0.f;
        for (int a = 0; a < n; ++a) { const float v = s_posv[a] * inv_tau; D += expf(v - m); possum += v; }
        for (int a = 0; a < k; ++a) D += expf(s_negv[a] * inv_tau - m);
        const float logD = m + logf(D);
        const float wi = w[i];
        rowstat[i * 2] = -wi * (possum / (float)n - logD);
        rowstat[i * 2 + 1] = wi;
        for (int a = 0; a < PROXY_KMAX; ++a) {
            if (a < n) { si[a] = s_pos[a]; sc[a] = wi * inv_tau * (expf(s_posv[a] * inv_tau - logD) - 1.0f / (float)n); }
            else { si[a] = -1; sc[a] = 0.f; }
            if (a < k) { si[PROXY_KMAX + a] = s_negi[a]; sc[PROXY_KMAX + a] = wi * inv_tau * expf(s_negv[a] * inv_tau - logD); }
            else { si[PROXY_KMAX + a] = -1; sc[PROXY_KMAX + a] = 0.f; }
        }
    }
}

// dfn[i][:] (+)= gscale / Z * sum_sel coef * P[sel][:]      (one block per row)
__global__ __launch_bounds__(256) void proxy_bwd_kernel(const int32_t* __restrict__ sel_idx, const float* __restrict__ sel_coef,
                                                         const float* __restrict__ P, int nb, int D, const float* __restrict__ denom,
                                                         float gscale, int accumulate, float* __restrict__ dfn) {
    const int i = blockIdx.x;
    const int32_t* si = sel_idx + (size_t)i * 2 * PROXY_KMAX;
    const float* sc = sel_coef + (size_t)i * 2 * PROXY_KMAX;
    const float z = gscale / denom[0];
    // the row's <= 2 K selected proxies once, in registers; an empty slot reads proxy 0 with coefficient 0 (a predicate per slot put every load of
    // the gather behind its own branch and wait: 34 us for 256 x 32 rows of 8 KB)
    int jj[2 * PROXY_KMAX];
    float cf[2 * PROXY_KMAX];
#pragma unroll
    for (int a = 0; a < 2 * PROXY_KMAX; ++a) {
        const int j = si[a];
        jj[a] = j < 0 ? 0 : j;
        cf[a] = j < 0 ? 0.f : sc[a];
    }
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2 * PROXY_KMAX; ++a) acc += cf[a] * P[(size_t)jj[a] * D + d];
        acc *= z;
        dfn[(size_t)i * D + d] = accumulate ? dfn[(size_t)i * D + d] + acc : acc;
    }
}

// ------------------------------------------------------------------------------------------------
// optional in-batch triplet head (BatchWeightedSoftmaxTripletLoss, losses.py:607-654), one wave64 per row.
//   S = fn @ fn^T;  p_i = argmin_{j: y_j == y_i} S_ij (self included),  n_i = argmax_{j: y_j != y_i} S_ij
//   row_i = -w_i log(e^{s_p/tau} / (e^{s_p/tau} + e^{s_n/tau})) = w_i softplus((s_n - s_p)/tau)
//   rowstat[i] = {row_i, w_i};  sel_idx[i] = {p_i, n_i};  sel_coef[i] = d row_i / d s_n = -d row_i / d s_p
// Ties go to the lowest index (torch.topk leaves them unspecified).  A row without any other-id sample has no
// negative: the reference's topk raises there; the row is skipped and status[0] is set to 1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void triplet_rows_kernel(const float* __restrict__ S, const int32_t* __restrict__ y,
                                                            const float* __restrict__ w, float inv_tau, int nb,
                                                            float* __restrict__ rowstat, int32_t* __restrict__ sel_idx,
                                                            float* __restrict__ sel_coef, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nb) return;
    const int yi = y[i];
    const float* row = S + (size_t)i * nb;
    float pmin = INFINITY, nmax = -INFINITY;
    int pi = -1, ni = -1;
    for (int j = lane; j < nb; j += 64) {
        const float s = row[j];
        if (y[j] == yi) { if (s < pmin) { pmin = s; pi = j; } }
        else if (s > nmax) { nmax = s; ni = j; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float op = __shfl_xor(pmin, o); const int opi = __shfl_xor(pi, o);
        const float on = __shfl_xor(nmax, o); const int oni = __shfl_xor(ni, o);
        if (opi >= 0 && (pi < 0 || op < pmin || (op == pmin && opi < pi))) { pmin = op; pi = opi; }
        if (oni >= 0 && (ni < 0 || on > nmax || (on == nmax && oni < ni))) { nmax = on; ni = oni; }
    }
    if (lane != 0) return;
    if (ni < 0) {
        rowstat[2 * i] = 0.f; rowstat[2 * i + 1] = 0.f;
        sel_idx[2 * i] = -1; sel_idx[2 * i + 1] = -1; sel_coef[i] = 0.f;
        status[0] = 1;
        return;
    }
    const float x = (nmax - pmin) * inv_tau;
    const float sp = x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x));       // softplus
    const float sg = 1.f / (1.f + expf(-x));
    rowstat[2 * i] = w[i] * sp; rowstat[2 * i + 1] = w[i];
    sel_idx[2 * i] = pi; sel_idx[2 * i + 1] = ni;
    sel_coef[i] = w[i] * sg * inv_tau;
}

// dS_sym[i][j] = dS[i][j] + dS[j][i] with dS[i][n_i] = +c_i, dS[i][p_i] = -c_i, c_i = gscale/Z * sel_coef[i]
// (the gradient wrt fn of a loss on S = fn fn^T is (dS + dS^T) fn).
__global__ __launch_bounds__(256) void triplet_bwd_kernel(const int32_t* __restrict__ sel_idx, const float* __restrict__ sel_coef, int nb,
                                                           const float* __restrict__ denom, float gscale, float* __restrict__ dS_sym) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)nb * nb) return;
    const int i = (int)(t / nb), j = (int)(t % nb);
    const float z = gscale / denom[0];
    float v = 0.f;
    if (sel_idx[2 * i + 1] == j) v += sel_coef[i];
    if (sel_idx[2 * i] == j) v -= sel_coef[i];
    if (sel_idx[2 * j + 1] == i) v += sel_coef[j];
    if (sel_idx[2 * j] == i) v -= sel_coef[j];
    dS_sym[t] = v * z;
}

}  // namespace dali

using namespace dali;

extern "C" int dali_center_loss_fwd(dali_ctx* ctx, void* stream, const float* S, const int32_t* labels, const int32_t* center_labels,
                                    const float* w, float tau, int nb, int NC, float* rowstat, float* sums) {
    DALI_REQUIRE(ctx && S && labels && center_labels && w && rowstat && sums, "dali_center_loss_fwd: null argument");
    DALI_REQUIRE(nb > 0 && NC > 0 && tau > 0.f, "dali_center_loss_fwd: bad sizes nb=%d NC=%d tau=%g", nb, NC, tau);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(center_rows_kernel, dim3((nb + 3) / 4), dim3(256), 0, st, S, labels, center_labels, w, 1.0f / tau, nb, NC, rowstat);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(rowstat_reduce_kernel, dim3(1), dim3(256), 0, st, rowstat, nb, 4, sums);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_center_loss_bwd(dali_ctx* ctx, void* stream, const float* S, const int32_t* labels, const int32_t* center_labels,
                                    const float* w, float tau, int nb, int NC, const float* denom, float gscale, float* dS) {
    DALI_REQUIRE(ctx && S && labels && center_labels && w && denom && dS, "dali_center_loss_bwd: null argument");
    DALI_REQUIRE(nb > 0 && NC > 0 && tau > 0.f, "dali_center_loss_bwd: bad sizes");
    hipLaunchKernelGGL(center_bwd_kernel, dim3((nb + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, labels, center_labels, w, 1.0f / tau, nb, NC,
                       denom, gscale, dS);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_proxy_loss_fwd(dali_ctx* ctx, void* stream, const float* S, const int32_t* labels, const int32_t* proxy_labels,
                                   const float* w, float tau, int nb, int NP, float* rowstat, float* sums, int32_t* sel_idx,
                                   float* sel_coef, int32_t* status) {
    DALI_REQUIRE(ctx && S && labels && proxy_labels && w && rowstat && sums && sel_idx && sel_coef && status, "dali_proxy_loss_fwd: null argument");
    DALI_REQUIRE(nb > 0 && NP > 0 && tau > 0.f, "dali_proxy_loss_fwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    DALI_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(proxy_rows_kernel, dim3(nb), dim3(256), 0, st, S, labels, proxy_labels, w, 1.0f / tau, nb, NP, rowstat, sel_idx, sel_coef, status);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(rowstat_reduce_kernel, dim3(1), dim3(256), 0, st, rowstat, nb, 2, sums);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_proxy_loss_bwd(dali_ctx* ctx, void* stream, const int32_t* sel_idx, const float* sel_coef, const float* proxies, int nb,
                                   int D, const float* denom, float gscale, int accumulate, float* dfn) {
    DALI_REQUIRE(ctx && sel_idx && sel_coef && proxies && denom && dfn, "dali_proxy_loss_bwd: null argument");
    DALI_REQUIRE(nb > 0 && D > 0, "dali_proxy_loss_bwd: bad sizes");
    hipLaunchKernelGGL(proxy_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, sel_idx, sel_coef, proxies, nb, D, denom, gscale, accumulate, dfn);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_proxy_kmax(void) { return PROXY_KMAX; }

extern "C" int dali_triplet_loss_fwd(dali_ctx* ctx, void* stream, const float* S, const int32_t* labels, const float* w, float tau, int nb,
                                     float* rowstat, float* sums, int32_t* sel_idx, float* sel_coef, int32_t* status) {
    DALI_REQUIRE(ctx && S && labels && w && rowstat && sums && sel_idx && sel_coef && status, "dali_triplet_loss_fwd: null argument");
    DALI_REQUIRE(nb > 0 && tau > 0.f, "dali_triplet_loss_fwd: bad sizes nb=%d tau=%g", nb, tau);
    hipStream_t st = (hipStream_t)stream;
    DALI_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(triplet_rows_kernel, dim3((nb + 3) / 4), dim3(256), 0, st, S, labels, w, 1.0f / tau, nb, rowstat, sel_idx, sel_coef, status);
    DALI_LAUNCH_CHECK();
    hipLaunchKernelGGL(rowstat_reduce_kernel, dim3(1), dim3(256), 0, st, rowstat, nb, 2, sums);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

extern "C" int dali_triplet_loss_bwd(dali_ctx* ctx, void* stream, const int32_t* sel_idx, const float* sel_coef, int nb, const float* denom,
                                     float gscale, float* dS_sym) {
    DALI_REQUIRE(ctx && sel_idx && sel_coef && denom && dS_sym, "dali_triplet_loss_bwd: null argument");
    DALI_REQUIRE(nb > 0 && nb <= 32768, "dali_triplet_loss_bwd: bad batch size %d", nb);
    const size_t total = (size_t)nb * nb;
    hipLaunchKernelGGL(triplet_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sel_idx, sel_coef, nb, denom,
                       gscale, dS_sym);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
