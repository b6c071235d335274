// optim.hip -- optimizer-side kernels of the trainer hot loop on the flat fp32 parameter storage:
//   Adam with L2 weight decay folded into the gradient (torch.optim.Adam, NOT AdamW; mainKIT.py:99, step at
//   train_encodersKIT.py:214-216), the EMA "momentum" model update (train_encodersKIT.py:218-226) and the
//   sum of squared weights the trainer logs (train_encodersKIT.py:229-231: 161 .item() syncs in the reference,
//   one fused reduction here).  All HBM-bound: Adam 4 reads + 3 writes of 4 B per element.
#include "common.h"

namespace dali {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float beta1, float beta2, float eps,
                                                    float weight_decay, float bc1, float bc2_sqrt, float grad_scale,
                                                    float* __restrict__ wsum_partial) {
    float local = 0.f;
    const float step_size = lr / bc1;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        float4 pv = *reinterpret_cast<const float4*>(p + i);
        float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 mv = *reinterpret_cast<const float4*>(m + i);
        float4 vv = *reinterpret_cast<const float4*>(v + i);
        float* pp = reinterpret_cast<float*>(&pv); float* gp = reinterpret_cast<float*>(&gv);
        float* mp = reinterpret_cast<float*>(&mv); float* vp = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float gg = gp[t] * grad_scale + weight_decay * pp[t];
            mp[t] = beta1 * mp[t] + (1.f - beta1) * gg;
            vp[t] = beta2 * vp[t] + (1.f - beta2) * gg * gg;
            const float denom = sqrtf(vp[t]) / bc2_sqrt + eps;
            pp[t] = pp[t] - step_size * (mp[t] / denom);
            local += pp[t] * pp[t];
        }
        *reinterpret_cast<float4*>(p + i) = pv;
        *reinterpret_cast<float4*>(m + i) = mv;
        *reinterpret_cast<float4*>(v + i) = vv;
    }
    if (wsum_partial) {
        __shared__ float red[4];
        local = wave_sum(local);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) wsum_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ m, const float* __restrict__ theta, size_t n, float beta) {
    const float omb = 1.f - beta;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        float4 a = *reinterpret_cast<const float4*>(m + i);
        const float4 b = *reinterpret_cast<const float4*>(theta + i);
        a.x = beta * a.x + omb * b.x; a.y = beta * a.y + omb * b.y; a.z = beta * a.z + omb * b.z; a.w = beta * a.w + omb * b.w;
        *reinterpret_cast<float4*>(m + i) = a;
    }
}

}  // namespace dali

using namespace dali;

extern "C" int dali_adam_step(dali_ctx* ctx, void* stream, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                              int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                              float* weights_sqsum) {
    DALI_REQUIRE(ctx && params && grads && exp_avg && exp_avg_sq, "dali_adam_step: null argument");
    DALI_REQUIRE(n > 0 && n % 4 == 0 && step >= 1, "dali_adam_step: n must be a positive multiple of 4 and step >= 1 (n=%lld step=%d)", (long long)n, step);
    DALI_REQUIRE(((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
                   reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0, "dali_adam_step: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    float* partial = nullptr;
    if (weights_sqsum) {
        partial = static_cast<float*>(workspace(ctx, 2048 * sizeof(float)));
        if (!partial) return DALI_ERR_NOMEM;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, partial);
    DALI_LAUNCH_CHECK();
    if (weights_sqsum) {
        hipLaunchKernelGGL(partial_sum_kernel, dim3(1), dim3(256), 0, st, partial, (int)blocks, weights_sqsum);
        DALI_LAUNCH_CHECK();
    }
    return DALI_OK;
}

extern "C" int dali_ema_update(dali_ctx* ctx, void* stream, float* momentum, const float* online, int64_t n, float beta) {
    DALI_REQUIRE(ctx && momentum && online, "dali_ema_update: null argument");
    DALI_REQUIRE(n > 0 && n % 4 == 0, "dali_ema_update: n must be a positive multiple of 4");
    DALI_REQUIRE(((reinterpret_cast<uintptr_t>(momentum) | reinterpret_cast<uintptr_t>(online)) & 15) == 0, "dali_ema_update: buffers must be 16-byte aligned");
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, momentum, online, (size_t)n, beta);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}
