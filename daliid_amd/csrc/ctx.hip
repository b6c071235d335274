// ctx.hip -- context, error string and workspace of libdaliid_hip.
#include "common.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <string_view>

namespace dali {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void* workspace(dali_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return ctx->ws;
    // grow-only; the old block may still be in use by enqueued kernels -> drain first
    if (hipDeviceSynchronize() != hipSuccess) { set_error("workspace: device synchronize failed"); return nullptr; }
    if (ctx->ws) { (void)hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    const size_t want = align_up(bytes + bytes / 8, 1 << 20);
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        set_error("workspace: hipMalloc(%zu) failed", want);
        return nullptr;
    }
    ctx->ws = p;
    ctx->ws_bytes = want;
    return p;
}

static std::atomic<int> g_env_epoch{0};
void env_reload();
int env_int_cached(const char* name, int def, int* cache, int* epoch_seen) {
    const int e = g_env_epoch.load(std::memory_order_acquire);
    if (*epoch_seen != e) { const char* v = getenv(name); *cache = v ? atoi(v) : def; *epoch_seen = e; }
    return *cache;
}
void env_reload() { g_env_epoch.fetch_add(1, std::memory_order_acq_rel); }

// ---- arrival counters of reduce_finish_kernel (reduce_finish.h): one definition for the whole library ----
constexpr int RF_SLOTS_ = 128, RF_GROUPS_ = 64;             // = RF_SLOTS, RF_GROUPS of reduce_finish.h (checked there)
static __device__ unsigned int g_rf_counters[RF_SLOTS_ * RF_GROUPS_];

unsigned rf_next_slot() {
    static std::atomic<unsigned> n{0};
    return n.fetch_add(1) % RF_SLOTS_;
}
int rf_counter_base(unsigned int** out) {
    static std::mutex mu;
    static unsigned int* bases[64] = {};
    int dev = 0;
    DALI_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("reduce_finish: device index %d", dev); return DALI_ERR_LIMIT; }
    std::lock_guard<std::mutex> lock(mu);
    if (!bases[dev]) { DALI_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&bases[dev]), HIP_SYMBOL(g_rf_counters))); }
    *out = bases[dev];
    return DALI_OK;
}

}  // namespace dali

extern "C" int dali_version(void) { return 100; }

// Diagnostic (include/daliid_debug.h): the DALI_* A/B switches are re-read from the environment at their next use.
extern "C" int dali_debug_reload_env(void) { dali::env_reload(); return DALI_OK; }

extern "C" const char* dali_last_error(void) { return dali::g_err; }

extern "C" int dali_ctx_create(int device, dali_ctx** out) {
    DALI_REQUIRE(out != nullptr, "dali_ctx_create: out is null");
    int ndev = 0;
    DALI_HIP(hipGetDeviceCount(&ndev));
    DALI_REQUIRE(device >= 0 && device < ndev, "dali_ctx_create: device %d not in [0,%d)", device, ndev);
    hipDeviceProp_t prop;
    DALI_HIP(hipGetDeviceProperties(&prop, device));          // (no hipSetDevice: the caller's current device is left alone)
    if (std::string_view(prop.gcnArchName).substr(0, 6) != "gfx950") {
        dali::set_error("dali_ctx_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return DALI_ERR_UNSUPPORTED;
    }
    dali_ctx* c = new (std::nothrow) dali_ctx();
    if (!c) { dali::set_error("dali_ctx_create: out of host memory"); return DALI_ERR_NOMEM; }
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    c->ws = nullptr;
    c->ws_bytes = 0;
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
    *out = c;
    return DALI_OK;
}

extern "C" int dali_ctx_comm_destroy(dali_ctx* ctx);
extern "C" int dali_ctx_destroy(dali_ctx* ctx) {
    if (!ctx) return DALI_OK;
    (void)dali_ctx_comm_destroy(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    delete ctx;
    return DALI_OK;
}

extern "C" int dali_ctx_reserve(dali_ctx* ctx, size_t bytes) {
    DALI_REQUIRE(ctx != nullptr, "dali_ctx_reserve: null ctx");
    return dali::workspace(ctx, bytes) ? DALI_OK : DALI_ERR_NOMEM;
}
