// Per-column fp64 sums of a [rows][cols] fp32 partial slab, finished in the SAME launch.
//
// Level 1 (all blocks): grid (column groups, S); block (g, s) adds rows s, s+S, s+2S, ... of its <= 256 columns in fp64 and stores one
// row of scratch[S][cols].  Level 2 (the LAST block of a column group to arrive, found with one atomic per block): adds the S rows of
// its columns in a fixed order and hands every channel's NV sums to the finish functor (BatchNorm coefficients, gamma / beta gradients,
// plain sums ...).  The order of both levels is fixed, so results do not depend on which block came last.  This replaces the
// reduce_partials + finalize launch pairs (two ~5 us dependent launches, ~140 pairs per ResNet step) by one.
//
// The arrival counters live in a module-scope array (zero at load); the last block puts its counter back to zero, and the host hands out
// counter slots round-robin, so concurrent launches on different streams use different counters unless > RF_SLOTS of them are in flight.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "kernels.h"

namespace dali {

constexpr int RF_SLOTS = 128, RF_GROUPS = 64;            // (ctx.hip sizes the array with the same two numbers)
// The counter array, the slot dispenser and the per-device address table are defined ONCE, in ctx.hip (this header is included by several
// translation units; the kernels get the counters as a pointer argument, so they need no access to the symbol).
// Counters return to zero when the last block of a group leaves.  A launch that faults leaves its slot dirty, but a faulted launch is a
// sticky HIP error for the whole process, so nothing can run on the slot afterwards; launches on one stream are serialised and may share a
// slot, and slots only collide when more than RF_SLOTS launches on DIFFERENT streams are in flight at once (the library uses two streams).
unsigned rf_next_slot();
int rf_counter_base(unsigned int** out);               // the counter array's address on the current device

template <int NV, class Fin>
__global__ __launch_bounds__(256) void reduce_finish_kernel(const float* __restrict__ in, int rows, int cols, int S, double* scratch,
                                                            unsigned int* __restrict__ ctr, Fin fin) {
    constexpr int CG = (256 / NV) * NV;                  // columns per group: whole channels
    __shared__ double s_col[256];
    __shared__ int s_last;
    const int tid = threadIdx.x, s = blockIdx.y;
    const int col = blockIdx.x * CG + tid;
    const bool live = tid < CG && col < cols;
    if (live) {
        double acc[8];                                   // 8 independent chains keep 8 loads in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.0;
        int r = s;
        for (; r + 7 * S < rows; r += 8 * S) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[(size_t)(r + u * S) * cols + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += (double)v[u];
        }
        for (; r < rows; r += S) acc[0] += (double)in[(size_t)r * cols + col];
        // device-scope (sc1) store: written through to where the other XCDs' loads see it -- no fence, see below
        __hip_atomic_store(&scratch[(size_t)s * cols + col], ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Ordering by hand instead of __threadfence(): an agent-scope fence on gfx950 is buffer_wbl2 + buffer_inv, a write-back / invalidate of
    // the XCD's whole L2 (measured: 22-35 us per launch here, against 5 us for the launch pair this kernel replaces).  Every wave waits for
    // the acknowledgement of its own sc1 stores, the barrier collects the waves, then one relaxed device-scope atomic counts the arrival.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&ctr[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == (unsigned)(S - 1);
        if (s_last) __hip_atomic_store(&ctr[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the slot's next launch
    }
    __syncthreads();
    if (!s_last) return;
    double tot = 0.0;
    if (live) {
        const double* p = scratch + col;                 // device-scope (sc1) loads: the other blocks' rows, not a stale line of this XCD's L2
        double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int t = 0;
        for (; t + 32 <= S; t += 32) {                   // 32 loads in flight: S = 64 rows are two round trips (same slice order as below)
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = __hip_atomic_load(p + (size_t)(t + u) * cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 32; ++u) a[u & 7] += v[u];
        }
        for (; t + 16 <= S; t += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = __hip_atomic_load(p + (size_t)(t + u) * cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u & 7] += v[u];
        }
        for (; t < S; ++t) a[t & 7] += __hip_atomic_load(p + (size_t)t * cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += a[u];         // slice sums (rows t = u mod 8) in a fixed order
    }
    s_col[tid] = tot;
    __syncthreads();
    const int c = blockIdx.x * (CG / NV) + tid;
    if (tid < CG / NV && c * NV < cols) fin(c, &s_col[tid * NV]);
}

// out_k[c] = k-th of the NV sums of channel c (null outputs are skipped)
struct FinStore {
    float* out[4];
    int nv;
    __device__ void operator()(int c, const double* v) const {
        for (int k = 0; k < nv; ++k)
            if (out[k]) out[k][c] = (float)v[k];
    }
};

// rows of the first level, as reduce_partials chose them: >= 8 partial rows per thread, at most REDUCE_SMAX
inline int rf_levels(int rows) {
    int S = rows / 8;
    if (S < 1) S = 1;
    if (S > REDUCE_SMAX) S = REDUCE_SMAX;
    return S;
}

template <int NV, class Fin>
inline int launch_reduce_finish(hipStream_t st, const float* partial, int rows, int C, double* scratch, const Fin& fin) {
    constexpr int CG = (256 / NV) * NV;
    const int cols = C * NV, groups = (cols + CG - 1) / CG, S = rf_levels(rows);
    if (groups > RF_GROUPS) { set_error("reduce_finish: more than %d column groups", RF_GROUPS); return DALI_ERR_LIMIT; }
    unsigned int* ctr = nullptr;
    if (int rc = rf_counter_base(&ctr)) return rc;
    ctr += (size_t)rf_next_slot() * RF_GROUPS;
    hipLaunchKernelGGL((reduce_finish_kernel<NV, Fin>), dim3(groups, S), dim3(256), 0, st, partial, rows, cols, S, scratch, ctr, fin);
    DALI_LAUNCH_CHECK();
    return DALI_OK;
}

}  // namespace dali
