// targets.hip -- the epoch-level loss targets of the trainer on device-resident embeddings
// (train_encodersKIT.py:113-156 with selectProxiesByTriagulation :252-284):
//   per identity: center = mean of the un-normalised embeddings, then L2-normalised (:131-137);
//                 proxies = farthest-point sampling, first pick supplied by the host (the reference draws it with
//                 np.random.choice, :257), then repeatedly the row whose minimum distance to the chosen set is
//                 largest (:267-269), min(5, n) per identity (:259), rows L2-normalised (:139-141);
//                 max pairwise distance among the chosen rows (:278, the "Mean Max Proxies Positive Distances" log).
// One 256-thread block per identity; rows of an identity are addressed through a host-built sorted order.  Distances
// are direct fp32 sum((a-b)^2) in a fixed order (deterministic).  HBM-bound: (1 + num_proxies) reads of the
// identity's rows = 4*D*(1+P) bytes per training image per epoch.
#include "common.h"

namespace dali {

// ||a - b||^2 over D (multiple of 4) by one wave, result in every lane
__device__ __forceinline__ float wave_sqdist(const float* __restrict__ a, const float* __restrict__ b, int d, int lane) {
    float s = 0.f;
    for (int i = lane * 4; i < d; i += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + i);
        const float4 y = *reinterpret_cast<const float4*>(b + i);
        const float e0 = x.x - y.x, e1 = x.y - y.y, e2 = x.z - y.z, e3 = x.w - y.w;
        s += e0 * e0; s += e1 * e1; s += e2 * e2; s += e3 * e3;
    }
    return wave_sum(s);
}

__global__ __launch_bounds__(256) void class_targets_kernel(const float* __restrict__ fvs, int d, const int* __restrict__ order,
                                                             const int* __restrict__ bounds, const int* __restrict__ first_pick,
                                                             int num_proxies, float* __restrict__ running, float* __restrict__ centers,
                                                             float* __restrict__ proxies, int* __restrict__ proxy_rows,
                                                             float* __restrict__ max_dist) {
    __shared__ float red[4];
    __shared__ int red_i[4];
    __shared__ int chosen[16];
    const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lo = bounds[c], n = bounds[c + 1] - lo;
    const int* rows = order + lo;
    float* run = running + lo;
    if (n <= 0) {                                              // an identity without images: defined (zero) outputs
        for (int col = threadIdx.x; col < d; col += 256) centers[(size_t)c * d + col] = 0.f;
        for (int j = 0; j < num_proxies; ++j) {
            for (int col = threadIdx.x; col < d; col += 256) proxies[((size_t)c * num_proxies + j) * d + col] = 0.f;
            if (threadIdx.x == 0) proxy_rows[c * num_proxies + j] = -1;
        }
        if (threadIdx.x == 0) max_dist[c] = 0.f;
        return;
    }

    // ---- center: column sums, mean, L2 norm (no epsilon: train_encodersKIT.py:136-137) ----
    float sq = 0.f;
    for (int col = threadIdx.x * 4; col < d; col += 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < n; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(fvs + (size_t)rows[r] * d + col);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float inv_n = 1.f / (float)n;
        s.x *= inv_n; s.y *= inv_n; s.z *= inv_n; s.w *= inv_n;
        *reinterpret_cast<float4*>(centers + (size_t)c * d + col) = s;
        sq += s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
    }
    sq = wave_sum(sq);
    if (lane == 0) red[w] = sq;
    __syncthreads();
    const float cnorm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    for (int col = threadIdx.x * 4; col < d; col += 1024) {
        float4 s = *reinterpret_cast<float4*>(centers + (size_t)c * d + col);
        s.x /= cnorm; s.y /= cnorm; s.z /= cnorm; s.w /= cnorm;
        *reinterpret_cast<float4*>(centers + (size_t)c * d + col) = s;
    }

    // ---- farthest-point sampling ----
    const int np = num_proxies < n ? num_proxies : n;
    for (int r = threadIdx.x; r < n; r += 256) run[r] = INFINITY;
    if (threadIdx.x == 0) { const int f = first_pick[c]; chosen[0] = f < 0 ? 0 : (f >= n ? n - 1 : f); }
    __syncthreads();
    for (int j = 0; j + 1 < np; ++j) {
        const float* pivot = fvs + (size_t)rows[chosen[j]] * d;
        float best = -1.f; int best_i = -1;
        for (int r = w; r < n; r += 4) {
            const float dist = sqrtf(wave_sqdist(fvs + (size_t)rows[r] * d, pivot, d, lane));
            float m = run[r];
            m = dist < m ? dist : m;
            if (lane == 0) run[r] = m;
            if (m >= best) { best = m; best_i = r; }          // rows ascend within a wave: ties keep the highest index
        }
        if (lane == 0) { red[w] = best; red_i[w] = best_i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = red[0]; int bi = red_i[0];
            for (int k = 1; k < 4; ++k)
                if (red_i[k] >= 0 && (red[k] > b || (red[k] == b && red_i[k] > bi))) { b = red[k]; bi = red_i[k]; }
            chosen[j + 1] = bi;
        }
        __syncthreads();
    }

    // ---- max pairwise distance among the chosen (dist[proxies][:, proxies].max(), :278) ----
    float mx = 0.f;
    int pair = 0;
    for (int a = 0; a < np; ++a)
        for (int b = a + 1; b < np; ++b, ++pair)
            if ((pair & 3) == w) {
                const float dist = sqrtf(wave_sqdist(fvs + (size_t)rows[chosen[a]] * d, fvs + (size_t)rows[chosen[b]] * d, d, lane));
                mx = dist > mx ? dist : mx;
            }
    __syncthreads();
    if (lane == 0) red[w] = mx;
    __syncthreads();
    if (threadIdx.x == 0) max_dist[c] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));

    // ---- proxies: the chosen rows, L2-normalised (no epsilon, :140) ----
    for (int j = w; j < num_proxies; j += 4) {
        float* out = proxies + ((size_t)c * num_proxies + j) * d;
        if (j >= np) {
            for (int i = lane * 4; i < d; i += 256) *reinterpret_cast<float4*>(out + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane == 0) proxy_rows[c * num_proxies + j] = -1;
            continue;
        }
        const int grow = rows[chosen[j]];
        const float* src = fvs + (size_t)grow * d;
        float s = 0.f;
        for (int i = lane * 4; i < d; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            s += v.x * v.x; s += v.y * v.y; s += v.z * v.z; s += v.w * v.w;
        }
        const float nrm = sqrtf(wave_sum(s));
        for (int i = lane * 4; i < d; i += 256) {
            float4 v = *reinterpret_cast<const float4*>(src + i);
            v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
            *reinterpret_cast<float4*>(out + i) = v;
        }
        if (lane == 0) proxy_rows[c * num_proxies + j] = grow;
    }
}

}  // namespace dali

using namespace dali;

extern "C" int dali_class_targets(dali_ctx* ctx, void* stream, const float* fvs, int n, int d, const int32_t* order,
                                  const int32_t* bounds, int n_classes, const int32_t* first_pick, int num_proxies,
                                  float* centers, float* proxies, int32_t* proxy_rows, float* max_dist) {
    DALI_REQUIRE(ctx && fvs && order && bounds && first_pick && centers && proxies && proxy_rows && max_dist,
                 "dali_class_targets: null argument");
    DALI_REQUIRE(n > 0 && n_classes > 0 && d > 0 && d % 4 == 0, "dali_class_targets: need n, n_classes > 0 and d %% 4 == 0 (n=%d classes=%d d=%d)", n, n_classes, d);
    DALI_REQUIRE(num_proxies >= 1 && num_proxies <= 16, "dali_class_targets: num_proxies must be in 1..16 (got %d)", num_proxies);
    float* running = static_cast<float*>(workspace(ctx, (size_t)n * sizeof(float)));
    if (!running) return DALI_ERR_NOMEM;
    hipLaunchKernelGGL(class_targets_kernel, dim3(n_classes), dim3(256), 0, (hipStream_t)stream, fvs, d, order, bounds, first_pick,
                       num_proxies, running, centers, proxies, proxy_rows, max_dist);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
