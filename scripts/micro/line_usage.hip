// Micro-benchmark: L2 -> LDS fill rate when a k-step takes 64 B (half a cache line) vs 128 B (a full line) of every operand
// row, 2 tiles in flight per block (development aid: does BK = 64 pay for the conv kernels?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// operand = rows of `row_bytes` bytes; a block owns 256 rows (ROWB = 64) or 128 rows (ROWB = 128) per 16 KiB tile and walks
// along the rows ROWB bytes per iteration, moving to the next row group when a row is exhausted.
template <int ROWB>
__global__ __launch_bounds__(256) void fill(const uint8_t* __restrict__ src, size_t window, int row_bytes, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, (int)window, 0x00020000);
    constexpr int LPR = ROWB / 16;                 // lanes per row
    constexpr int ROWS = 16384 / ROWB;             // rows per tile
    const int steps_per_row = row_bytes / ROWB;
    uint32_t acc = 0;
    const size_t group_bytes = (size_t)ROWS * row_bytes;
    size_t group = (size_t)blockIdx.x;
    const size_t n_groups = window / group_bytes;
    int kk = 0;
    for (int it = 0; it < iters + 2; ++it) {
        if (it < iters) {
            uint8_t* st = smem + (it % 3) * 16384;
            const size_t gbase = (group % n_groups) * group_bytes + (size_t)kk * ROWB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int piece = i * 256 + tid;                    // 16-byte piece of the tile
                const int row = piece / LPR, ch = piece % LPR;
                const uint32_t off = (uint32_t)(gbase + (size_t)row * row_bytes + ch * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(st + (i * 256 + (tid & ~63)) * 16), 16, off, 0, 0, 0);
            }
            if (++kk == steps_per_row) { kk = 0; group += gridDim.x; }
        }
        if (it >= 2) {
            if (it < iters) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (it == iters) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            acc += *reinterpret_cast<uint32_t*>(smem + ((it - 2) % 3) * 16384 + ((tid * 148) & 16383 & ~3));
        }
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
    uint8_t* src; uint32_t* sink;
    const size_t bytes = (size_t)1 << 30;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&sink, 64);
    (void)hipMemset(src, 1, bytes);
    const int iters = 256;
    for (size_t window_mb : {64, 1024}) {
        for (int row_bytes : {128, 512, 2048}) {
            for (int bpc : {1, 3}) {
                const int grid = 256 * bpc;
                const double total = (double)grid * iters * 16384;
                const size_t window = window_mb << 20;
                float a = time_us([&] { hipLaunchKernelGGL(fill<64>, dim3(grid), dim3(256), 49152, 0, src, window, row_bytes, iters, sink); });
                float b = time_us([&] { hipLaunchKernelGGL(fill<128>, dim3(grid), dim3(256), 49152, 0, src, window, row_bytes, iters, sink); });
                printf("window %4zu MiB row %4d B, %d blocks/CU: 64 B/row/step %.0f GB/s per CU (%.2f TB/s) | 128 B/row/step %.0f GB/s per CU (%.2f TB/s)\n",
                       window_mb, row_bytes, bpc, total / a / 1e3 / 256, total / a / 1e6, total / b / 1e3 / 256, total / b / 1e6);
            }
        }
    }
    return 0;
}
