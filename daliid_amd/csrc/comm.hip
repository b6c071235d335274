// comm.hip -- the gradient all-reduce of the data-parallel train step under the C ABI: one RCCL communicator per context (= per
// process = per GPU), SUM all-reduce of a slice of the flat fp32 gradient buffer on a caller-supplied HIP stream.
//
// Replaces nn.DataParallel's per-step gradient reduce to GPU 0 (Encoders.py:39-40).  RCCL is bound at run time (dlopen of the librccl
// that is already in the process -- PyTorch-ROCm ships one -- else /opt/rocm/lib/librccl.so), so libdaliid_hip.so loads and every
// other entry point works on a box without RCCL.  The unique id is created by rank 0 (dali_comm_unique_id) and handed to the other
// ranks by the host program (daliid_amd/parallel.py broadcasts it through torch.distributed's store); nothing else of the exchange
// touches Python.  xGMI is a point-to-point mesh: buckets that divide by the world size go as reduce-scatter + all-gather on the
// caller's stream (the two halves of a ring all-reduce, each rank owning count / world elements in between), the rest as ncclAllReduce.
#include "common.h"
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>

namespace {

typedef int (*fn_get_unique_id)(void*);
struct dali_nccl_id { char internal[128]; };         // ncclUniqueId (passed by value)
typedef int (*fn_comm_init_rank_t)(void**, int, dali_nccl_id, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank_t comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_reduce_scatter reduce_scatter = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
};
constexpr int NCCL_FLOAT = 7, NCCL_SUM = 0;          // ncclFloat32 / ncclSum (rccl.h enums)

Rccl* rccl() {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (r.lib) return &r;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }     // the copy already in the process (PyTorch's) first
    for (const char* n : names) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!h) { dali::set_error("comm: librccl.so not found (%s)", dlerror()); return nullptr; }
    r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank_t)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    r.reduce_scatter = (fn_reduce_scatter)dlsym(h, "ncclReduceScatter");
    r.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
    r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.reduce_scatter || !r.all_gather) {
        dali::set_error("comm: librccl.so lacks an expected symbol");
        return nullptr;
    }
    r.lib = h;
    return &r;
}

int nccl_check(Rccl* r, int rc, const char* what) {
    if (rc == 0) return DALI_OK;
    dali::set_error("comm: %s failed: %s", what, r->error_string ? r->error_string(rc) : "rccl error");
    return DALI_ERR_HIP;
}

}  // namespace

extern "C" int dali_comm_unique_id(void* id128) {
    DALI_REQUIRE(id128 != nullptr, "dali_comm_unique_id: null argument");
    Rccl* r = rccl();
    if (!r) return DALI_ERR_UNSUPPORTED;
    return nccl_check(r, r->get_unique_id(id128), "ncclGetUniqueId");
}

extern "C" int dali_ctx_comm_init(dali_ctx* ctx, const void* id128, int rank, int world) {
    DALI_REQUIRE(ctx && id128 && world >= 1 && rank >= 0 && rank < world, "dali_ctx_comm_init: bad argument (rank %d of %d)", rank, world);
    DALI_REQUIRE(ctx->comm == nullptr, "dali_ctx_comm_init: this context already has a communicator");
    Rccl* r = rccl();
    if (!r) return DALI_ERR_UNSUPPORTED;
    int dev = -1;
    DALI_HIP(hipGetDevice(&dev));
    DALI_REQUIRE(dev == ctx->device, "dali_ctx_comm_init: current device %d is not the context's device %d", dev, ctx->device);
    dali_nccl_id id;
    memcpy(id.internal, id128, 128);
    void* comm = nullptr;
    const int rc = nccl_check(r, r->comm_init_rank(&comm, world, id, rank), "ncclCommInitRank");
    if (rc) return rc;
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return DALI_OK;
}

extern "C" int dali_ctx_comm_destroy(dali_ctx* ctx) {
    if (!ctx || !ctx->comm) return DALI_OK;
    Rccl* r = rccl();
    if (r) (void)r->comm_destroy(ctx->comm);
    ctx->comm = nullptr; ctx->comm_world = 1; ctx->comm_rank = 0;
    return DALI_OK;
}

extern "C" int dali_allreduce_bucket(dali_ctx* ctx, void* stream, float* buf, int64_t count) {
    DALI_REQUIRE(ctx && buf && count >= 0, "dali_allreduce_bucket: bad argument");
    DALI_REQUIRE(ctx->comm != nullptr, "dali_allreduce_bucket: the context has no communicator (dali_ctx_comm_init)");
    if (count == 0) return DALI_OK;
    Rccl* r = rccl();
    if (!r) return DALI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int world = ctx->comm_world;
    // DALIID_COMM_RSAG=1: reduce-scatter + all-gather as two calls (the form SURVEY 8e names for the 7-link xGMI mesh).  Off by default: that
    // branch has only ever run at world size 1 (no multi-GPU box was available to the builder), and a rank-offset mistake in it would
    // corrupt every gradient bucket silently; ncclAllReduce is what torch.distributed runs for the same buffers.  Compare the two bit for
    // bit on a 2-GPU box (tests/test_gpu_dp.py::test_abi_allreduce_matches_torch_distributed covers whichever is selected) before switching.
    static const bool rsag = getenv("DALIID_COMM_RSAG") && atoi(getenv("DALIID_COMM_RSAG")) != 0;
    if (rsag && world > 1 && count % world == 0 && ((count / world) * sizeof(float)) % 256 == 0) {
        // in place: rank r keeps the sum of chunk r after the reduce-scatter, then every rank gathers all chunks
        const size_t chunk = (size_t)(count / world);
        float* mine = buf + (size_t)ctx->comm_rank * chunk;
        int rc = nccl_check(r, r->reduce_scatter(buf, mine, chunk, NCCL_FLOAT, NCCL_SUM, ctx->comm, st), "ncclReduceScatter");
        if (rc) return rc;
        return nccl_check(r, r->all_gather(mine, buf, chunk, NCCL_FLOAT, ctx->comm, st), "ncclAllGather");
    }
    return nccl_check(r, r->all_reduce(buf, buf, (size_t)count, NCCL_FLOAT, NCCL_SUM, ctx->comm, st), "ncclAllReduce");
}
