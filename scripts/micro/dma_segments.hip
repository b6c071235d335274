// Micro-benchmark: L2-resident operands -> LDS by LDS-DMA with 16 waves per CU, as a function of the contiguous segment one
// row contributes to a wave-instruction (64 B = the conv kernels' [rows][32 bf16] k-tile ... 1024 B = fully contiguous).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int SEG, int NSTAGE, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void feed(const uint8_t* __restrict__ A, int rows_total, int stride_b, int ksteps, int instr_per_wave, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LPR = SEG / 16, RPI = 64 / LPR;                 // lanes per row, rows per instruction
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A), 0, rows_total * stride_b, 0x00020000);
    const int kwrap = stride_b / SEG;
    const int row0 = ((blockIdx.x >> 3) % 4) * (NWAVE * RPI * instr_per_wave) + wave * RPI + lane / LPR;
    const uint32_t off0 = (uint32_t)(row0 * stride_b + (lane % LPR) * 16);
    const int stage_bytes = NWAVE * instr_per_wave * 1024;
    int k = 0;
    auto issue = [&](int stage) {
        uint8_t* sa = smem + stage * stage_bytes;
        for (int i = 0; i < instr_per_wave; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_ptr)(sa + (i * NWAVE + wave) * 1024), 16, off0 + (uint32_t)(i * NWAVE * RPI * stride_b) + k * SEG, 0, 0, 0);
        if (++k == kwrap) k = 0;
    };
    for (int i = 0; i < NSTAGE - 1; ++i) issue(i);
    uint32_t acc = 0;
    int st = 0, fill = NSTAGE - 1;
    for (int kt = 0; kt < ksteps; ++kt) {
        issue(fill);
        if (instr_per_wave == 2) {
            if (NSTAGE == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            if (NSTAGE == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<uint32_t*>(smem + st * stage_bytes + ((tid * 148) % stage_bytes & ~3));
        __builtin_amdgcn_s_barrier();
        st = (st + 1) % NSTAGE; fill = (fill + 1) % NSTAGE;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F> static float time_us(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) { (void)hipEventRecord(e0, 0); f(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
    return best * 1e3f;
}
template <int SEG, int NSTAGE, int NWAVE> static void run(const uint8_t* A, uint32_t* sink, int ipw, int bpc) {
    const int stride_b = 2048, ksteps = 512;
    const int rows_total = 4 * NWAVE * (1024 / SEG) * ipw;
    const int lds = NSTAGE * NWAVE * ipw * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&feed<SEG, NSTAGE, NWAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float t = time_us([&] { hipLaunchKernelGGL((feed<SEG, NSTAGE, NWAVE>), dim3(256 * bpc), dim3(NWAVE * 64), lds, 0, A, rows_total, stride_b, ksteps, ipw, sink); });
    const double bytes = (double)NWAVE * ipw * 1024 * ksteps * bpc;
    printf("segment %4d B, %2d waves x %d instr/k-step (%3d KiB/k-step), %d-stage, %d blocks/CU: %.3f us/k-step, %.0f GB/s per CU (%.1f TB/s)\n", SEG, NWAVE, ipw,
           NWAVE * ipw, NSTAGE, bpc, t / ksteps, bytes / t / 1e3, bytes / t / 1e6 * 256);
}
int main() {
    uint8_t* A; uint32_t* sink;
    (void)hipMalloc(&A, 64 << 20); (void)hipMalloc(&sink, 64);
    (void)hipMemset(A, 1, 64 << 20);
    run<64, 3, 16>(A, sink, 2, 1); run<128, 3, 16>(A, sink, 2, 1); run<256, 3, 16>(A, sink, 2, 1); run<1024, 3, 16>(A, sink, 2, 1);
    run<64, 4, 16>(A, sink, 2, 1); run<1024, 4, 16>(A, sink, 2, 1);
    run<64, 3, 4>(A, sink, 2, 1); run<64, 3, 4>(A, sink, 2, 3); run<1024, 3, 4>(A, sink, 2, 3);
    run<64, 3, 4>(A, sink, 4, 3); run<128, 3, 4>(A, sink, 4, 3); run<1024, 3, 4>(A, sink, 4, 3);
    run<64, 3, 8>(A, sink, 2, 2); run<128, 3, 8>(A, sink, 2, 2);
    return 0;
}
