// Does a wave see its own LDS writes in its next LDS read without waiting for them?  (The distance kernel's store strips rely on it: each
// wave writes a 2 KiB strip in the accumulator layout with ds_write_b128 and reads it back transposed with ds_read_b128, no s_waitcnt between.)
// Each wave owns a strip; per iteration every lane writes values that encode (iteration, writer lane, register), then every lane reads the
// chunk another lane wrote and checks it.  Variants: FULL = all 64 lanes write (the distance kernel's pattern, pitch 128 B, chunk ^ (row & 7));
// HALF = only the lanes of one 8-lane half of every 16 write, alternating per iteration (the pattern of a pass that stages half of an accumulator
// block; pitch 256 B, chunk ^ (row << 1)); WAIT = s_waitcnt lgkmcnt(0) between the writes and the reads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_order scripts/micro/lds_order.hip && /tmp/lds_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool HALF, bool WAIT>
__global__ __launch_bounds__(512) void probe(unsigned long long* bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* strip = lds + wave * 2048;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        if (!HALF) {
            // 16 rows x 128 B; lane writes row lane & 15, logical chunks (lane >> 4) and 4 + (lane >> 4), physical = logical ^ (row & 7)
            char* wr = strip + (lane & 15) * 128;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int c = ii * 4 + (lane >> 4);
                const float tag = (float)(it * 4096 + (lane & 15) * 64 + c * 4);
                *reinterpret_cast<f32x4_t*>(wr + ((c ^ (lane & 7)) << 4)) = f32x4_t{tag, tag + 1.f, tag + 2.f, tag + 3.f};
            }
            if (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // read: row (lane >> 3) + 8 t, logical chunk lane & 7
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = (lane >> 3) + 8 * t, c = lane & 7;
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(strip + row * 128 + ((c ^ (row & 7)) << 4));
                const float tag = (float)(it * 4096 + row * 64 + c * 4);
                nbad += (v[0] != tag) + (v[1] != tag + 1.f) + (v[2] != tag + 2.f) + (v[3] != tag + 3.f);
            }
        } else {
            // 8 rows x 256 B; the lanes whose bit 3 equals it & 1 write row lane & 7, chunks i * 4 + (lane >> 4), physical = logical ^ (row << 1)
            const int h = it & 1, p8 = lane & 7;
            char* wr = strip + p8 * 256;
            if (((lane >> 3) & 1) == h) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = i * 4 + (lane >> 4);
                    const float tag = (float)(it * 4096 + p8 * 64 + c * 4);
                    *reinterpret_cast<f32x4_t*>(wr + ((c ^ (p8 << 1)) << 4)) = f32x4_t{tag, tag + 1.f, tag + 2.f, tag + 3.f};
                }
            }
            if (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int row = lane >> 3, kq = lane & 7;
            const char* rd = strip + row * 256 + ((kq ^ row) << 5);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(rd + 16 * t);
                const float tag = (float)(it * 4096 + row * 64 + (2 * kq + t) * 4);
                nbad += (v[0] != tag) + (v[1] != tag + 1.f) + (v[2] != tag + 2.f) + (v[3] != tag + 3.f);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads are done before the next iteration overwrites the strip
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <bool HALF, bool WAIT>
static void run(const char* name) {
    unsigned long long* bad;
    CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    const int iters = 4000, blocks = 2048;
    hipLaunchKernelGGL((probe<HALF, WAIT>), dim3(blocks), dim3(512), 8 * 2048, 0, bad, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("%-44s %llu wrong values of %.3g checked\n", name, h, (double)blocks * 512 * iters * 8);
    CK(hipFree(bad));
}

int main() {
    run<false, false>("full-exec writes, no wait (distance strips)");
    run<false, true>("full-exec writes, lgkmcnt(0)");
    run<true, false>("half-exec writes, no wait");
    run<true, true>("half-exec writes, lgkmcnt(0)");
    return 0;
}
