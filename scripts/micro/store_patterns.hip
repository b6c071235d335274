// Micro-benchmark: write-only kernels with different store shapes over a [P][C] bf16 tensor (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// (a) fully coalesced: lane writes 16 B, wave writes 1 KiB contiguous
__global__ __launch_bounds__(256) void st_linear(uint4* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = make_uint4(1, 2, 3, 4);
}
// (b) conv-epilogue shape: block = 128 px x 128 ch tile of a [P][C] tensor; 4 waves (2x2), per wave 64x64;
//     lane: pixel = lane&15 (+16j), channels (lane>>4)*4 (+16 i): 8-byte stores
__global__ __launch_bounds__(256) void st_tile8(uint16_t* out, int P, int C, int tiles_m) {
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int mb = wm * 64 + (lane >> 4) * 4, nb = wn * 64 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = tn * 128 + nb + j * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tm * 128 + mb + i * 16;
            *reinterpret_cast<uint2*>(out + (size_t)p * C + c) = make_uint2(p, c);
        }
    }
}
// (c) same tile, 16-byte stores: 16 lanes cover one pixel's 256 B, wave covers 4 pixels per instruction
__global__ __launch_bounds__(256) void st_tile16(uint16_t* out, int P, int C, int tiles_m) {
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    for (int t = threadIdx.x; t < 128 * 16; t += 256) {
        const int px = t >> 4, ch = t & 15;
        *reinterpret_cast<uint4*>(out + (size_t)(tn * 128 + px) * C + tm * 128 + ch * 8) = make_uint4(t, 2, 3, 4);
    }
}
// (d) like (c) but a block owns all C channels of 128*128/C pixels... (full rows, contiguous region per block)
__global__ __launch_bounds__(256) void st_rows16(uint16_t* out, int P, int C) {
    const size_t base = (size_t)blockIdx.x * 128 * 128;      // elements
    for (int t = threadIdx.x; t < 128 * 16; t += 256) *reinterpret_cast<uint4*>(out + base + (size_t)t * 8) = make_uint4(t, 2, 3, 4);
}

template <typename F> static float time_us(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) { (void)hipEventRecord(e0, 0); f(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
    return best * 1e3f;
}
int main() {
    const int P = 524288;
    uint16_t* out; (void)hipMalloc(&out, (size_t)P * 2048 * 2);
    for (int C : {256, 512, 2048}) {
        const int Pc = (int)((size_t)524288 * 256 / C);
        const size_t bytes = (size_t)Pc * C * 2;
        const int tiles_m = C / 128, tiles_n = Pc / 128;
        float a = time_us([&] { hipLaunchKernelGGL(st_linear, dim3(4096), dim3(256), 0, 0, (uint4*)out, bytes / 16); });
        float b = time_us([&] { hipLaunchKernelGGL(st_tile8, dim3(tiles_m * tiles_n), dim3(256), 0, 0, out, Pc, C, tiles_m); });
        float c = time_us([&] { hipLaunchKernelGGL(st_tile16, dim3(tiles_m * tiles_n), dim3(256), 0, 0, out, Pc, C, tiles_m); });
        float d = time_us([&] { hipLaunchKernelGGL(st_rows16, dim3(tiles_m * tiles_n), dim3(256), 0, 0, out, Pc, C); });
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&st_tile8), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&st_tile16), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
        float b3 = time_us([&] { hipLaunchKernelGGL(st_tile8, dim3(tiles_m * tiles_n), dim3(256), 49152, 0, out, Pc, C, tiles_m); });
        float c3 = time_us([&] { hipLaunchKernelGGL(st_tile16, dim3(tiles_m * tiles_n), dim3(256), 49152, 0, out, Pc, C, tiles_m); });
        printf("   with 48 KiB LDS per block (3 blocks/CU): tile 8B %.1f us (%.2f TB/s) | tile 16B %.1f us (%.2f)\n", b3, bytes / b3 / 1e6, c3, bytes / c3 / 1e6);
        printf("C=%4d (%.0f MB): linear %.1f us (%.2f TB/s) | tile 8B %.1f us (%.2f) | tile 16B %.1f us (%.2f) | rows 16B %.1f us (%.2f)\n", C, bytes / 1e6,
               a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, bytes / d / 1e6);
    }
    return 0;
}
