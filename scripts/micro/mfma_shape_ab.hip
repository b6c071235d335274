// Micro-benchmark: the fragment-read + MFMA part of the 256 x 256 k-tile-64 convolution kernel (igemm_conv_k64_kernel<4,4,2>:
// 16 waves of 64 x 64, LDS image [rows][64 bf16] with the chunk ^ ((row >> 1) & 7) swizzle, one barrier per 64-deep k-step) with
// v_mfma_f32_16x16x32_bf16 (what every kernel in csrc/ uses) against v_mfma_f32_32x32x16_bf16 on the SAME per-wave tile, LDS image and
// data.  No DMA refill: the operands stay in LDS, so this isolates issue slots, LDS reads and the clock the chip holds per shape
// (MI355X_MICROARCH.md, DVFS give-back item 7).  Random bf16 operands; wall time by HIP events, interleaved rounds in one process.
// Per wave and 64-deep k-step both shapes read 8 + 8 ds_read_b128 and run 512 MFMA cycles (32 x 16 or 16 x 32).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_shape_ab.hip -o scripts/micro/mfma_shape_ab && scripts/micro/mfma_shape_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int SHAPE>      // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(1024) void loop(const uint16_t* __restrict__ src, int ksteps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];            // [512 rows][64]: A rows 0..255, B rows 256..511
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    for (int i = tid; i < 512 * 8; i += 1024) *reinterpret_cast<uint4*>(smem + i * 8) = *reinterpret_cast<const uint4*>(src + ((size_t)blockIdx.x % 4) * 512 * 64 + i * 8);
    __syncthreads();
    const uint16_t* sa = smem;
    const uint16_t* sb = smem + 256 * 64;
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x4_t acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 3);
        for (int kt = 0; kt < ksteps; ++kt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int fo = frag_off ^ (h << 5);
                bf16x8_t fa[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + (wm * 64 + i * 16) * 64 + fo);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(sb + (wn * 64 + j * 16) * 64 + fo);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    } else {
        f32x16_t acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // lane (r = lane & 31, h = lane >> 5) reads row r, logical 16-byte chunk 2*k16 + h; physical chunk = logical ^ ((row >> 1) & 7)
        const int r = lane & 31, hh = lane >> 5;
        const int row_off = r * 64, swz = (r >> 1) & 7;
        for (int kt = 0; kt < ksteps; ++kt) {
#pragma unroll
            for (int k16 = 0; k16 < 4; ++k16) {
                const int ch = ((2 * k16 + hh) ^ swz) << 3;
                bf16x8_t fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + (wm * 64 + i * 32) * 64 + row_off + ch);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sb + (wn * 64 + j * 32) * 64 + row_off + ch);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int rr = 0; rr < 16; ++rr) sum += acc[i][j][rr];
    }
    out[(size_t)blockIdx.x * 1024 + tid] = sum;
}

int main() {
    const int blocks = 256, ksteps = 2048;
    std::vector<uint16_t> h(4 * 512 * 64);
    srand(12);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    uint16_t* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, (size_t)blocks * 1024 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int lds = 512 * 64 * 2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * 256 * 256 * 64 * (double)ksteps * blocks;
    // checksum agreement of the two shapes (same products, different summation grouping)
    std::vector<float> r0((size_t)blocks * 1024), r1((size_t)blocks * 1024);
    hipLaunchKernelGGL(loop<0>, dim3(blocks), dim3(1024), lds, 0, d, 8, o); hipMemcpy(r0.data(), o, r0.size() * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(loop<1>, dim3(blocks), dim3(1024), lds, 0, d, 8, o); hipMemcpy(r1.data(), o, r1.size() * 4, hipMemcpyDeviceToHost);
    double s0 = 0, s1 = 0; for (size_t i = 0; i < r0.size(); ++i) { s0 += r0[i]; s1 += r1[i]; }
    printf("checksum 16x16x32 %.6e  32x32x16 %.6e  (per-block tile sums must agree to fp32 summation order)\n", s0, s1);
    for (int warm = 0; warm < 20; ++warm) { hipLaunchKernelGGL(loop<0>, dim3(blocks), dim3(1024), lds, 0, d, ksteps, o); hipLaunchKernelGGL(loop<1>, dim3(blocks), dim3(1024), lds, 0, d, ksteps, o); }
    hipDeviceSynchronize();
    for (int round = 0; round < 6; ++round) {
        float ms[2];
        for (int s = 0; s < 2; ++s) {
            hipEventRecord(e0);
            for (int it = 0; it < 5; ++it) { if (s == 0) hipLaunchKernelGGL(loop<0>, dim3(blocks), dim3(1024), lds, 0, d, ksteps, o); else hipLaunchKernelGGL(loop<1>, dim3(blocks), dim3(1024), lds, 0, d, ksteps, o); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[s], e0, e1); ms[s] /= 5;
        }
        printf("round %d: 16x16x32 %.3f ms = %.0f TFLOP/s | 32x32x16 %.3f ms = %.0f TFLOP/s | ratio (16x16 / 32x32 time) %.3f\n", round, ms[0], flop / ms[0] / 1e9, ms[1], flop / ms[1] / 1e9, ms[0] / ms[1]);
    }
    return 0;
}
