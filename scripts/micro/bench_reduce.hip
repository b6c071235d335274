// Micro-benchmark of split-K reduce variants (development aid; not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void red_a(const float* __restrict__ partial, float* __restrict__ out, size_t elems, int splits) {
    __shared__ float4 red[WAVES * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t i4 = ((size_t)blockIdx.x * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 + 3 < elems) {
        const float* base = partial + i4;
        for (int k0 = w; k0 < splits; k0 += 8 * WAVES) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * WAVES;
                v[u] = (k < splits) ? *reinterpret_cast<const float4*>(base + (size_t)k * elems) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
    }
    if (WAVES > 1) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (w == 0)
            for (int k = 1; k < WAVES; ++k) { const float4 v = red[k * 64 + lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (w == 0 && i4 + 3 < elems) *reinterpret_cast<float4*>(out + i4) = s;
}

// B: each thread owns U chunks (stride 64 chunks) and loops over all slices itself; no LDS.
template <int U>
__global__ __launch_bounds__(256) void red_b(const float* __restrict__ partial, float* __restrict__ out, size_t elems, int splits) {
    const size_t c0 = ((size_t)blockIdx.x * 256 * U + threadIdx.x) * 4;
    float4 s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i4 = c0 + (size_t)u * 1024;
            v[u] = (i4 + 3 < elems) ? *reinterpret_cast<const float4*>(partial + (size_t)k * elems + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { s[u].x += v[u].x; s[u].y += v[u].y; s[u].z += v[u].z; s[u].w += v[u].w; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i4 = c0 + (size_t)u * 1024;
        if (i4 + 3 < elems) *reinterpret_cast<float4*>(out + i4) = s[u];
    }
}

// C: one chunk per thread, all slices by the same thread, KU slices in flight; no LDS.
template <int KU>
__global__ __launch_bounds__(256) void red_c(const float* __restrict__ partial, float* __restrict__ out, size_t elems, int splits) {
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 3 >= elems) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < splits; k0 += KU) {
        float4 v[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u)
            v[u] = (k0 + u < splits) ? *reinterpret_cast<const float4*>(partial + (size_t)(k0 + u) * elems + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < KU; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    *reinterpret_cast<float4*>(out + i4) = s;
}

__global__ void fill(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}

template <typename F>
static float time_it(F launch, float* dirty, size_t dirty_n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, dirty, dirty_n);     // producer kernel before, like wgrad
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    struct Case { size_t elems; int splits; } cases[] = {{2359296, 1}, {2359296, 4}, {36864, 128}, {36864, 16}, {65536, 64}, {262144, 32}, {589824, 14}, {1048576, 8}};
    float *partial, *out;
    hipMalloc(&partial, (size_t)64 << 20);
    hipMalloc(&out, (size_t)16 << 20);
    for (auto c : cases) {
        const size_t elems = c.elems; const int splits = c.splits;
        const size_t tot = elems * splits;
        if (tot * 4 > ((size_t)64 << 20)) { printf("skip\n"); continue; }
        const size_t chunks = elems / 4;
        float a16 = time_it([&] { hipLaunchKernelGGL(red_a<16>, dim3((chunks + 63) / 64), dim3(1024), 0, 0, partial, out, elems, splits); }, partial, tot);
        float a4 = time_it([&] { hipLaunchKernelGGL(red_a<4>, dim3((chunks + 63) / 64), dim3(256), 0, 0, partial, out, elems, splits); }, partial, tot);
        float b4 = time_it([&] { hipLaunchKernelGGL(red_b<4>, dim3((chunks + 1023) / 1024), dim3(256), 0, 0, partial, out, elems, splits); }, partial, tot);
        float b8 = time_it([&] { hipLaunchKernelGGL(red_b<8>, dim3((chunks + 2047) / 2048), dim3(256), 0, 0, partial, out, elems, splits); }, partial, tot);
        float c4 = time_it([&] { hipLaunchKernelGGL(red_c<4>, dim3((chunks + 255) / 256), dim3(256), 0, 0, partial, out, elems, splits); }, partial, tot);
        float c8 = time_it([&] { hipLaunchKernelGGL(red_c<8>, dim3((chunks + 255) / 256), dim3(256), 0, 0, partial, out, elems, splits); }, partial, tot);
        float empty = time_it([&] { hipLaunchKernelGGL(fill, dim3(1), dim3(64), 0, 0, out, 64); }, partial, tot);
        printf("elems %8zu splits %4d (%.1f MB): a16 %.1f  a4 %.1f  b4 %.1f  b8 %.1f  c4 %.1f  c8 %.1f us   (tiny kernel %.1f us)\n", elems, splits, tot * 4 / 1e6, a16, a4, b4, b8, c4, c8, empty);
    }
    return 0;
}
