// Micro-benchmark of split-K reduce variants (development aid; not part of the library).  Round 3: the shipped kernel takes ~30 us for
// 34 MB of slabs whatever its wave count; torch's column sum of the same bytes takes ~6.  Variants:
//   cur<W>   the shipped form: a block owns 1 KiB of every slab (64 lanes x 16 B), wave w sums slabs w, w + W, ...; LDS combine
//   xcd<W>   the same, block -> chunk mapping made contiguous per XCD (blocks b, b + 8, ... of an XCD own neighbouring KiB)
//   wide<KU> a block of 256 lanes owns 4 KiB of every slab, every lane sums ALL slabs for its 16 B, KU loads in flight; no LDS
//   widex<KU> wide + the per-XCD contiguous mapping
//   stream   reads the same bytes linearly (what the memory system gives a kernel of this size)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bench_reduce scripts/micro/bench_reduce.hip && scripts/micro/bench_reduce
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcd_contig(unsigned b, unsigned nb) {
    return (nb & 7) ? b : (b & 7) * (nb >> 3) + (b >> 3);
}

template <int WAVES, bool XCD>
__global__ __launch_bounds__(WAVES * 64) void red_cur(const float* __restrict__ partial, float* __restrict__ out, size_t elems, int splits) {
    __shared__ float4 red[WAVES * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned blk = XCD ? xcd_contig(blockIdx.x, gridDim.x) : blockIdx.x;
    const size_t i4 = ((size_t)blk * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 + 3 < elems) {
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
    if (w == 0 && i4 + 3 < elems) *reinterpret_cast<float4*>(out + i4) = s;
}

template <int KU, bool XCD>
__global__ __launch_bounds__(256) void red_wide(const float* __restrict__ partial, float* __restrict__ out, size_t elems, int splits) {
    const unsigned blk = XCD ? xcd_contig(blockIdx.x, gridDim.x) : blockIdx.x;
    const size_t i4 = ((size_t)blk * 256 + threadIdx.x) * 4;
    if (i4 + 3 >= elems) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* base = partial + i4;
    int k = 0;
    auto batch = [&](auto nconst) {
        constexpr int N = decltype(nconst)::value;
        float4 v[N];
#pragma unroll
        for (int u = 0; u < N; ++u) v[u] = *reinterpret_cast<const float4*>(base + (size_t)(k + u) * elems);
#pragma unroll
        for (int u = 0; u < N; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        k += N;
    };
    while (k + KU <= splits) batch(std::integral_constant<int, KU>{});
    while (k < splits) batch(std::integral_constant<int, 1>{});
    *reinterpret_cast<float4*>(out + i4) = s;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void red_lib(const float* __restrict__ partial, float* __restrict__ out,
                                                                  size_t elems, int splits, int accumulate) {
    __shared__ float4 red[WAVES * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t i4 = ((size_t)blockIdx.x * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool vec = (elems & 3) == 0 && i4 + 3 < elems;      // slabs stay 16-byte aligned only then
    if (vec) {
        // batches of 16, 8, 4, 2, 1 slabs, every load of a batch unconditional: a per-load "k < splits ? load : 0" made hipcc branch around
        // each load and wait for it (MI355X guide, trap (c) of the projection-GEMM notes): 32 dependent round trips, 28-37 us per launch
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
    } else if (i4 < elems) {
        float* sp = reinterpret_cast<float*>(&s);
        for (int k = w; k < splits; k += WAVES)
            for (size_t e = i4; e < elems; ++e) sp[e - i4] += partial[(size_t)k * elems + e];
    }
    if (WAVES > 1) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (w == 0)
            for (int k = 1; k < WAVES; ++k) { const float4 v = red[k * 64 + lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (w == 0 && i4 < elems) {
        if (vec) {
            float4* o = reinterpret_cast<float4*>(out + i4);
            if (accumulate) { const float4 p = *o; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
            *o = s;
        } else {
            const float* sp = reinterpret_cast<const float*>(&s);
            for (int t = 0; t < 4 && i4 + t < elems; ++t) out[i4 + t] = accumulate ? out[i4 + t] + sp[t] : sp[t];
        }
    }
}

template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void red_var(const float* __restrict__ partial, float* __restrict__ out,
                                                                  size_t elems, int splits, int accumulate) {
    __shared__ float4 red[WAVES * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t i4 = ((size_t)blockIdx.x * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool vec = (elems & 3) == 0 && i4 + 3 < elems;      // slabs stay 16-byte aligned only then
    if (vec) {
        // batches of 16, 8, 4, 2, 1 slabs, every load of a batch unconditional: a per-load "k < splits ? load : 0" made hipcc branch around
        // each load and wait for it (MI355X guide, trap (c) of the projection-GEMM notes): 32 dependent round trips, 28-37 us per launch
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
    } else if ((MODE & 1) && i4 < elems) {
        float* sp = reinterpret_cast<float*>(&s);
        for (int k = w; k < splits; k += WAVES)
            for (size_t e = i4; e < elems; ++e) sp[e - i4] += partial[(size_t)k * elems + e];
    }
    if (WAVES > 1) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (w == 0)
            for (int k = 1; k < WAVES; ++k) { const float4 v = red[k * 64 + lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (w == 0 && i4 < elems) {
        if (vec) {
            float4* o = reinterpret_cast<float4*>(out + i4);
            if (accumulate) { const float4 p = *o; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
            *o = s;
        } else if (MODE & 2) {
            const float* sp = reinterpret_cast<const float*>(&s);
            for (int t = 0; t < 4 && i4 + t < elems; ++t) out[i4 + t] = accumulate ? out[i4 + t] + sp[t] : sp[t];
        }
    }
}

__global__ __launch_bounds__(256) void stream_read(const float* __restrict__ p, float* __restrict__ out, size_t n4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = s.x;
}

__global__ void fill(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}

template <typename F>
static float time_it(F launch, float* dirty, size_t dirty_n) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // 10 launches back to back between one pair of events (the event pair itself costs ~6 us), slabs rewritten before
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, dirty, dirty_n);
    launch();
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < 10; ++it) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 100.f;
}

int main() {
    struct Case { size_t elems; int splits; } cases[] = {{16384, 256}, {36864, 256}, {65536, 128}, {147456, 128}, {262144, 32}, {589824, 32}, {524288, 16}, {1048576, 8}, {2359296, 8}};
    float *partial, *out;
    CK(hipMalloc(&partial, (size_t)96 << 20));
    CK(hipMalloc(&out, (size_t)16 << 20));
    for (auto c : cases) {
        const size_t elems = c.elems; const int splits = c.splits;
        const size_t tot = elems * splits;
        const size_t chunks = elems / 4;
        const unsigned nb1 = (chunks + 63) / 64, nb4 = (chunks + 255) / 256;
#define T(...) time_it([&] { __VA_ARGS__; }, partial, tot)
        float c16 = T(hipLaunchKernelGGL((red_cur<16, false>), dim3(nb1), dim3(1024), 0, 0, partial, out, elems, splits));
        float c4 = T(hipLaunchKernelGGL((red_cur<4, false>), dim3(nb1), dim3(256), 0, 0, partial, out, elems, splits));
        float c1 = T(hipLaunchKernelGGL((red_cur<1, false>), dim3(nb1), dim3(64), 0, 0, partial, out, elems, splits));
        float x16 = T(hipLaunchKernelGGL((red_cur<16, true>), dim3(nb1), dim3(1024), 0, 0, partial, out, elems, splits));
        float x4 = T(hipLaunchKernelGGL((red_cur<4, true>), dim3(nb1), dim3(256), 0, 0, partial, out, elems, splits));
        float x1 = T(hipLaunchKernelGGL((red_cur<1, true>), dim3(nb1), dim3(64), 0, 0, partial, out, elems, splits));
        float l16 = T(hipLaunchKernelGGL((red_lib<16>), dim3(nb1), dim3(1024), 0, 0, partial, out, elems, splits, 0));
        float l4 = T(hipLaunchKernelGGL((red_lib<4>), dim3(nb1), dim3(256), 0, 0, partial, out, elems, splits, 0));
        float v0 = T(hipLaunchKernelGGL((red_var<4, 0>), dim3(nb1), dim3(256), 0, 0, partial, out, elems, splits, 0));
        float v1 = T(hipLaunchKernelGGL((red_var<4, 1>), dim3(nb1), dim3(256), 0, 0, partial, out, elems, splits, 0));
        float v2 = T(hipLaunchKernelGGL((red_var<4, 2>), dim3(nb1), dim3(256), 0, 0, partial, out, elems, splits, 0));
        float w8 = T(hipLaunchKernelGGL((red_wide<8, false>), dim3(nb4), dim3(256), 0, 0, partial, out, elems, splits));
        float w16 = T(hipLaunchKernelGGL((red_wide<16, false>), dim3(nb4), dim3(256), 0, 0, partial, out, elems, splits));
        float wx8 = T(hipLaunchKernelGGL((red_wide<8, true>), dim3(nb4), dim3(256), 0, 0, partial, out, elems, splits));
        float wx16 = T(hipLaunchKernelGGL((red_wide<16, true>), dim3(nb4), dim3(256), 0, 0, partial, out, elems, splits));
        float st = T(hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, 0, partial, out, tot / 4));
        float empty = T(hipLaunchKernelGGL(fill, dim3(1), dim3(64), 0, 0, out, 64));
        printf("elems %8zu splits %4d (%5.1f MB): lib16 %5.1f lib4 %5.1f | var4: none %5.1f tail-in %5.1f scalar-out %5.1f | cur16 %5.1f cur4 %5.1f cur1 %5.1f | xcd16 %5.1f xcd4 %5.1f xcd1 %5.1f | wide8 %5.1f wide16 %5.1f widex8 %5.1f widex16 %5.1f | stream %5.1f | tiny %4.1f us\n",
               elems, splits, tot * 4 / 1e6, l16, l4, v0, v1, v2, c16, c4, c1, x16, x4, x1, w8, w16, wx8, wx16, st, empty);
        fflush(stdout);
    }
    return 0;
}
