// Micro-benchmark: the operand feed of the 256x256 conv kernel alone (LDS-DMA ring, counted waits, one barrier per k-step,
// no MFMA): how does the k-step time depend on the operand row stride and on all workgroups walking K in lockstep?
// (development aid: L2 channel camping test)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// A = [M rows][stride_b bytes], B = [N rows][stride_b bytes]; block (tm, tn) loads rows tm*256.. of A and tn*256.. of B,
// 64 B per row per k-step.  ROT: the block starts its K walk at a block-dependent offset.
template <int NSTAGE, bool ROT, bool CONV = false>
__global__ __launch_bounds__(1024) void feed(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, int tiles_m, int tiles_n,
                                             int stride_b, int ksteps, uint32_t* sink, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware: xcd = b % 8 owns a contiguous range of n tiles, m fastest
    const int b = blockIdx.x, xcd = b & 7, local = b >> 3;
    const int per_xcd = tiles_m * tiles_n / 8;
    const int t = xcd * per_xcd + local;
    const int tm = t % tiles_m, tn = t / tiles_m;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A), 0, tiles_m * 256 * stride_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(B), 0, tiles_n * 256 * stride_b, 0x00020000);
    const int r_in = lane >> 2, ch = lane & 3;
    const uint32_t a_off = (uint32_t)((tm * 256 + wave * 16 + r_in) * stride_b + ch * 16);
    // CONV: A rows are 9216 B (3x3x512 weights), B rows are pixels of 1024 B (512 channels); k-step = (channel block, tap), taps fastest
    const uint32_t b_off = CONV ? (uint32_t)((tn * 256 + wave * 16 + r_in + 64) * 1024 + ch * 16) : (uint32_t)((tn * 256 + wave * 16 + r_in) * stride_b + ch * 16);
    int k = ROT ? (int)((b * 7) % ksteps) : 0;
    auto issue = [&](int stage) {
        uint8_t* sa = smem + stage * 32768;
        if (CONV) {
            const int tap = k % 9, cb = k / 9;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(sa + wave * 1024), 16, a_off + tap * 1024 + cb * 64, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(sa + 16384 + wave * 1024), 16, b_off + ((tap / 3 - 1) * 8 + (tap % 3 - 1)) * 1024 + cb * 64, 0, 0, 0);
        } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(sa + wave * 1024), 16, a_off + k * 64, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_ptr)(sa + 16384 + wave * 1024), 16, b_off + k * 64, 0, 0, 0);
        }
        if (++k == ksteps) k = 0;
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < NSTAGE - 1; ++i) issue(i);
    uint32_t acc = 0;
    int st = 0, fill = NSTAGE - 1;
    for (int kt = 0; kt < ksteps; ++kt) {
        if (kt + NSTAGE - 1 < ksteps) {
            issue(fill);
            if (NSTAGE == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<uint32_t*>(smem + st * 32768 + ((tid * 148) & 32767 & ~3));
        __builtin_amdgcn_s_barrier();
        st = (st + 1) % NSTAGE; fill = (fill + 1) % NSTAGE;
    }
    if (tid == 0) clk[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F> static float time_us(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) { (void)hipEventRecord(e0, 0); f(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
    return best * 1e3f;
}
int main() {
    uint8_t *A, *B; uint32_t* sink; unsigned long long* clk;
    const size_t bytes = (size_t)1 << 30;
    (void)hipMalloc(&A, 64 << 20); (void)hipMalloc(&B, bytes); (void)hipMalloc(&sink, 64); (void)hipMalloc(&clk, 8 * 4096);
    (void)hipMemset(A, 1, 64 << 20); (void)hipMemset(B, 1, bytes);
    const int tiles_m = 2, tiles_n = 128;                       // 512 x 32768 outputs, 256 tiles = one per CU
    for (int stride_b : {9216, 9216 + 128, 2048, 2048 + 128, 4096, 4096 + 128, 1024, 1024 + 128, 512, 512 + 128}) {
        const int ksteps = (stride_b & ~255) / 64 > 144 ? 144 : (stride_b & ~255) / 64;
        const int reps = 144 / ksteps;                          // comparable run length
        (void)reps;
        float t4 = time_us([&] { hipLaunchKernelGGL((feed<4, false>), dim3(256), dim3(1024), 4 * 32768, 0, A, B, tiles_m, tiles_n, stride_b, ksteps, sink, clk); });
        float t4r = time_us([&] { hipLaunchKernelGGL((feed<4, true>), dim3(256), dim3(1024), 4 * 32768, 0, A, B, tiles_m, tiles_n, stride_b, ksteps, sink, clk); });
        float t3 = time_us([&] { hipLaunchKernelGGL((feed<3, false>), dim3(256), dim3(1024), 3 * 32768, 0, A, B, tiles_m, tiles_n, stride_b, ksteps, sink, clk); });
        printf("stride %5d B, %3d k-steps: 4-stage %.3f us/k-step (%.0f GB/s per CU) | rotated start %.3f us/k-step (%.0f GB/s per CU) | 3-stage %.3f\n", stride_b, ksteps,
               t4 / ksteps, 32768.0 / (t4 / ksteps) / 1e3, t4r / ksteps, 32768.0 / (t4r / ksteps) / 1e3, t3 / ksteps);
    }
    {
        const int ksteps = 144;
        float t4 = time_us([&] { hipLaunchKernelGGL((feed<4, false, true>), dim3(256), dim3(1024), 4 * 32768, 0, A, B, tiles_m, tiles_n, 9216, ksteps, sink, clk); });
        float t3 = time_us([&] { hipLaunchKernelGGL((feed<3, false, true>), dim3(256), dim3(1024), 3 * 32768, 0, A, B, tiles_m, tiles_n, 9216, ksteps, sink, clk); });
        float t2 = time_us([&] { hipLaunchKernelGGL((feed<2, false, true>), dim3(256), dim3(1024), 2 * 32768, 0, A, B, tiles_m, tiles_n, 9216, ksteps, sink, clk); });
        printf("conv3x3 pattern (l4.c2: 512 ch, 33.5 MB of pixels, taps fastest): 4-stage %.3f us/k-step (%.0f GB/s per CU) | 3-stage %.3f | 2-stage %.3f\n",
               t4 / ksteps, 32768.0 / (t4 / ksteps) / 1e3, t3 / ksteps, t2 / ksteps);
    }
    return 0;
}
