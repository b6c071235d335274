// common.h -- shared host/device helpers for libdaliid_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include "../../include/daliid.h"

namespace dali {

void set_error(const char* fmt, ...);

}  // namespace dali

struct dali_ctx {
    int device;
    int num_cus;
    void* ws;          // grow-only device workspace
    size_t ws_bytes;
    void* comm;        // RCCL communicator (ncclComm_t) of this process' rank, or null (dali_ctx_comm_init)
    int comm_rank, comm_world;
};

namespace dali {
// A/B switches read from the environment: cached per call site, re-read after dali_debug_reload_env() (so that one process can time
// variants interleaved on the same box: boxes differ by +-5 %, and a single kernel by more between two processes).
int env_int_cached(const char* name, int def, int* cache, int* epoch_seen);
}  // namespace dali
#define DALI_ENV_INT(NAME, DEF) ([]() -> int { static int v_ = 0, e_ = -1; return dali::env_int_cached(NAME, DEF, &v_, &e_); }())

namespace dali {
// Returns a workspace pointer of at least `bytes` (256-B aligned) or nullptr (+ error set).
void* workspace(dali_ctx* ctx, size_t bytes);
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
}  // namespace dali

#define DALI_HIP(call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            dali::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return DALI_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

#define DALI_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            dali::set_error(__VA_ARGS__);  \
            return DALI_ERR_INVALID;       \
        }                                  \
    } while (0)

#define DALI_LAUNCH_CHECK() DALI_HIP(hipGetLastError())

namespace dali {
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: the opt-in blocks run once per device (of the calling
// thread's current device), under a lock, and are retried if they failed.
struct DeviceOnce {
    std::mutex mu;
    unsigned long long done_mask = 0;
    struct Guard {
        DeviceOnce& o; unsigned long long bit = 1;
        explicit Guard(DeviceOnce& once) : o(once) { o.mu.lock(); int dev = 0; if (hipGetDevice(&dev) == hipSuccess) bit = 1ull << (dev & 63); }
        ~Guard() { o.mu.unlock(); }
        bool first() const { return !(o.done_mask & bit); }
        void done() { o.done_mask |= bit; }
    };
};
}  // namespace dali
#define DALI_ONCE_PER_DEVICE(...)                         \
    do {                                                  \
        static dali::DeviceOnce once_;                    \
        dali::DeviceOnce::Guard guard_(once_);            \
        if (guard_.first()) { __VA_ARGS__; guard_.done(); } \
    } while (0)

// ---- device helpers -------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// round-to-nearest-even f32 -> bf16 bits via the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
// two values in ONE v_cvt_pk_bf16_f32 (gfx950): converted one by one and merged with shift + or it compiled to four instructions per pair
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
