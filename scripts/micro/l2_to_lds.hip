// Micro-benchmark: L2 -> LDS fill rate per CU, LDS-DMA (buffer_load ... lds) vs global_load + ds_write (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// each block streams `iters` tiles of 16 KiB (4 waves x 4 x 1 KiB) from its own 64 KiB-strided window of `src` (wraps inside
// `window` bytes so the data stays L2 / MALL resident)
template <int MODE>
__global__ __launch_bounds__(256) void fill_lds(const uint16_t* __restrict__ src, size_t window_elems, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(src), 0, (int)(window_elems * 2), 0x00020000);
    uint32_t acc = 0;
    size_t base = ((size_t)blockIdx.x * 8192) % window_elems;          // 16 KiB per block start, elements
    for (int it = 0; it < iters; ++it) {
        uint16_t* st = smem + (it & 1) * 8192;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t off = (uint32_t)((base + (size_t)(wave * 4 + i) * 512 + lane * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(st + (wave * 4 + i) * 512), 16, off, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const uint4*>(src + base + (size_t)(wave * 4 + i) * 512 + lane * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(st + (wave * 4 + i) * 512 + lane * 8) = v[i];
        }
        __syncthreads();
        acc += *reinterpret_cast<uint32_t*>(st + ((tid * 37) & 8191 & ~1));
        base += (size_t)gridDim.x * 8192;
        if (base + 8192 > window_elems) base = ((size_t)blockIdx.x * 8192) % window_elems;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F> static float time_us(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) { (void)hipEventRecord(e0, 0); f(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
    return best * 1e3f;
}
int main() {
    uint16_t* src; uint32_t* sink;
    const size_t bytes = (size_t)1 << 30;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&sink, 64);
    (void)hipMemset(src, 1, bytes);
    const int iters = 256;
    for (size_t window_mb : {16, 128, 1024}) {
        const size_t window_elems = window_mb * 1024 * 1024 / 2;
        for (int bpc : {1, 2, 3}) {
            const int grid = 256 * bpc;
            const double total = (double)grid * iters * 16384;
            float a = time_us([&] { hipLaunchKernelGGL(fill_lds<0>, dim3(grid), dim3(256), 32768 + (3 - bpc) * 0, 0, src, window_elems, iters, sink); });
            float b = time_us([&] { hipLaunchKernelGGL(fill_lds<1>, dim3(grid), dim3(256), 32768, 0, src, window_elems, iters, sink); });
            printf("window %4zu MiB, %d blocks/CU: LDS-DMA %.0f GB/s per CU (%.2f TB/s) | load+ds_write %.0f GB/s per CU (%.2f TB/s)\n", window_mb, bpc,
                   total / a / 1e3 / 256, total / a / 1e6, total / b / 1e3 / 256, total / b / 1e6);
        }
    }
    return 0;
}
