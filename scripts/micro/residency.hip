// How many workgroups of a given shape does the chip hold at once?  Every block stamps its start time and its hardware id
// (XCC / SE / CU from HW_ID + XCC_ID), spins for 30 us, stamps its end and leaves; the host counts the blocks that started in the first
// 3 us, the distinct CUs they sat on, and the peak number of blocks alive at once (= what the chip holds).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/residency scripts/micro/residency.hip && /tmp/residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT, int NREG>
__global__ __launch_bounds__(NT) void spin_kernel(unsigned long long* stamps, unsigned* hwid, float* sink, int spin_ticks) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float r[NREG];                                   // NREG live registers per lane
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s + lds[threadIdx.x];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[blockIdx.x] = t0;
        stamps[gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        hwid[blockIdx.x] = (hw & 0xffffff) | ((xcc & 0xf) << 24);
    }
}

template <int NT, int NREG>
static void run(const char* name, int blocks, int lds_bytes) {
    unsigned long long* stamps; unsigned* hwid; float* sink;
    CK(hipMalloc(&stamps, blocks * 16)); CK(hipMalloc(&hwid, blocks * 4)); CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel<NT, NREG>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((spin_kernel<NT, NREG>), dim3(blocks), dim3(NT), lds_bytes, 0, stamps, hwid, sink, 3000);   // 30 us at 100 MHz
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> t(2 * blocks); std::vector<unsigned> h(blocks);
    CK(hipMemcpy(t.data(), stamps, blocks * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(h.data(), hwid, blocks * 4, hipMemcpyDeviceToHost));
    const unsigned long long t0 = *std::min_element(t.begin(), t.begin() + blocks);
    std::vector<std::pair<unsigned long long, int>> ev;          // peak number of blocks alive at once
    for (int i = 0; i < blocks; ++i) { ev.push_back({t[i], 1}); ev.push_back({t[blocks + i], -1}); }
    std::sort(ev.begin(), ev.end());
    int alive = 0, peak = 0;
    for (auto& e : ev) { alive += e.second; peak = std::max(peak, alive); }
    int first = 0; std::set<unsigned> cus, xccs;
    for (int i = 0; i < blocks; ++i)
        if (t[i] - t0 < 300) {                        // 3 us
            ++first;
            // HW_ID: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se (gfx9 layout); xcc in the top byte here
            cus.insert((h[i] >> 24) << 16 | (h[i] & 0xff00));
            xccs.insert(h[i] >> 24);
        }
    printf("%-44s blocks %4d  lds %6d  started in first 3 us: %4d  on %3zu distinct CUs (%zu XCCs)  peak alive %4d\n", name, blocks, lds_bytes, first, cus.size(), xccs.size(), peak);
    CK(hipFree(stamps)); CK(hipFree(hwid)); CK(hipFree(sink));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, max threads per CU %d, regs per block %d, LDS per block %zu\n", p.gcnArchName, p.multiProcessorCount,
           p.maxThreadsPerMultiProcessor, p.regsPerBlock, p.sharedMemPerBlock);
    run<1024, 8>("1024 threads, ~16 VGPRs", 512, 0);
    run<1024, 96>("1024 threads, ~110 VGPRs", 512, 0);
    run<1024, 8>("1024 threads, ~16 VGPRs, 128 KiB LDS", 512, 131072);
    run<1024, 96>("1024 threads, ~110 VGPRs, 128 KiB LDS", 512, 131072);
    run<512, 96>("512 threads, ~110 VGPRs, 72 KiB LDS", 1024, 73728);
    run<512, 8>("512 threads, ~16 VGPRs, 128 KiB LDS", 512, 131072);
    run<256, 96>("256 threads, ~110 VGPRs, 128 KiB LDS", 512, 131072);
    run<256, 8>("256 threads, ~16 VGPRs, 0 LDS", 4096, 0);
    // registers: 256-thread workgroups (one wave per SIMD each), no LDS: how many fit per CU at a given VGPR count?
    run<256, 24>("256 threads, NREG 24", 4096, 0); run<256, 40>("256 threads, NREG 40", 4096, 0); run<256, 52>("256 threads, NREG 52", 4096, 0);
    run<256, 60>("256 threads, NREG 60", 4096, 0); run<256, 72>("256 threads, NREG 72", 4096, 0); run<256, 84>("256 threads, NREG 84", 4096, 0);
    run<256, 96>("256 threads, NREG 96", 4096, 0); run<256, 116>("256 threads, NREG 116", 4096, 0); run<256, 150>("256 threads, NREG 150", 4096, 0);
    // how much LDS may a workgroup ask for and still share its CU with a second one?
    for (int kb : {32, 48, 56, 60, 64, 68, 72, 76, 80}) { char n[64]; snprintf(n, 64, "512 threads, ~16 VGPRs, %d KiB LDS", kb); run<512, 8>(n, 1024, kb * 1024); }
    for (int kb : {64, 72, 80}) { char n[64]; snprintf(n, 64, "512 threads, ~110 VGPRs, %d KiB LDS", kb); run<512, 96>(n, 1024, kb * 1024); }
    for (int kb : {24, 36, 40, 48, 52, 53}) { char n[64]; snprintf(n, 64, "256 threads, ~110 VGPRs, %d KiB LDS", kb); run<256, 96>(n, 2048, kb * 1024); }
    return 0;
}
