// vit_plan.hip -- native executor of the TransReID ViT encoder + BN neck:
//   make_models.build_transformer.forward (make_models.py:184-205) over vit_pytorch.TransReID.forward_features
//   (vit_pytorch.py:375-408; camera = view = 0, local_feature = False; Dropout rate 0).  DropPath (vit_pytorch.py:45-62, wired at
//   :338 and :178-179) is applied in training when the caller hands over the per-sample branch scales (dali_vit_set_drop_path).
// Same contract as resnet_plan.hip: the plan owns topology + launch order, PyTorch owns flat fp32 params / grads /
// buffers and one byte arena.  Parameter names / order follow the reference state_dict ("base.cls_token",
// "base.pos_embed", "base.patch_embed.proj.weight", "base.blocks.N....", "base.norm.*", "base.fc.*", "bottleneck.*").
// `base.fc` (vit_pytorch.py:349) is never used by forward(): it is carried as parameters with zero gradient.
#include "kernels.h"
#include <cstdio>
#include <new>
#include <string>
#include <vector>

using namespace dali;

namespace {
struct TInfo { std::string name; int64_t offset, numel; int shape[4]; int ndim; };
struct Lin { int K, N; int64_t w_off, b_off; uint16_t *w, *wt; };
struct Ln { int64_t g_off, b_off; float *mean, *rstd; };
struct VBlock {
    Ln n1, n2; Lin qkv, proj, fc1, fc2;
    uint16_t *x_in, *h1, *qkv_o, *att, *x_mid, *h2, *pre1, *act1, *x_out;
    float* lse;
};
struct VArena { size_t used = 0; size_t take(size_t b) { size_t o = used; used = align_up(used + b, 256); return o; } };
}  // namespace

struct dali_vit {
    dali_ctx* ctx;
    dali_vit_cfg cfg;
    int B, T, np, C, H, rows;
    std::vector<TInfo> params, buffers;
    int64_t param_elems = 0, buffer_elems = 0;
    int64_t cls_off, pos_off, fcw_off, fcb_off;
    Lin patch; Ln fin; std::vector<VBlock> blocks;
    int64_t neck_g, neck_b, neck_rm, neck_rv;
    size_t arena_bytes = 0;
    std::vector<std::pair<void**, size_t>> fixups;
    uint16_t *wbf16 = nullptr, *patches = nullptr, *pe = nullptr, *x0 = nullptr, *cls_rows = nullptr, *dcls_rows = nullptr, *dgf16 = nullptr;
    float *gf = nullptr, *dgf = nullptr, *neck_mean = nullptr, *neck_invstd = nullptr, *slab = nullptr, *partial = nullptr;
    float *fin_mean = nullptr, *fin_rstd = nullptr, *dp_rows = nullptr;        // dp_rows [2*depth][rows]: DropPath factors per output row
    double* scratch = nullptr;
    uint16_t* gbuf[4] = {nullptr, nullptr, nullptr, nullptr};
    float *P = nullptr, *G = nullptr, *Bf = nullptr;
    char* arena = nullptr;
    bool fwd_training = false;
    const float* dp_scale = nullptr;      // device [2*depth][B]: row 2i = attention branch of block i, 2i+1 = its MLP branch; null = no DropPath
    const float* dp_used = nullptr;       // what the last training forward applied (the backward mirrors it)
    int n_stages = 1;
    std::vector<int> stage_first_block;   // backward stage s runs blocks [stage_first_block[s+1], stage_first_block[s]) downwards
};

namespace {
int64_t addt(std::vector<TInfo>& v, int64_t& total, const std::string& name, std::initializer_list<int> shape) {
    TInfo t; t.name = name; t.ndim = (int)shape.size(); t.numel = 1;
    int i = 0;
    for (int s : shape) { t.shape[i++] = s; t.numel *= s; }
    for (; i < 4; ++i) t.shape[i] = 1;
    t.offset = total; total += (t.numel + 63) / 64 * 64;
    v.push_back(t);
    return t.offset;
}
void add_lin(dali_vit* n, Lin& l, const std::string& name, int K, int N) {
    l.K = K; l.N = N;
    l.w_off = addt(n->params, n->param_elems, name + ".weight", {N, K});
    l.b_off = addt(n->params, n->param_elems, name + ".bias", {N});
    l.w = l.wt = nullptr;
}
void add_ln(dali_vit* n, Ln& l, const std::string& name, int C) {
    l.g_off = addt(n->params, n->param_elems, name + ".weight", {C});
    l.b_off = addt(n->params, n->param_elems, name + ".bias", {C});
}
template <class T> void rsv(dali_vit* n, VArena& a, T*& p, size_t bytes) { n->fixups.emplace_back(reinterpret_cast<void**>(&p), a.take(bytes)); }
}  // namespace

extern "C" int dali_vit_create(dali_ctx* ctx, const dali_vit_cfg* cfg, dali_vit** out) {
    DALI_REQUIRE(ctx && cfg && out, "dali_vit_create: null argument");
    DALI_REQUIRE(cfg->batch > 0 && cfg->patch % 8 == 0 && cfg->stride > 0 && cfg->height >= cfg->patch && cfg->width >= cfg->patch,
                 "dali_vit_create: bad geometry");
    DALI_REQUIRE(cfg->dim % 64 == 0 && cfg->heads * 64 == cfg->dim && cfg->dim <= 2048, "dali_vit_create: head_dim must be 64 (dim=%d heads=%d)", cfg->dim, cfg->heads);
    DALI_REQUIRE(cfg->depth >= 1 && cfg->mlp_hidden % 32 == 0 && cfg->mlp_hidden > 0, "dali_vit_create: bad depth / mlp_hidden");
    dali_vit* n = new (std::nothrow) dali_vit();
    if (!n) { set_error("dali_vit_create: out of host memory"); return DALI_ERR_NOMEM; }
    n->ctx = ctx; n->cfg = *cfg;
    const int ny = (cfg->height - cfg->patch) / cfg->stride + 1, nx = (cfg->width - cfg->patch) / cfg->stride + 1;
    n->B = cfg->batch; n->np = ny * nx; n->T = n->np + 1; n->C = cfg->dim; n->H = cfg->heads; n->rows = n->B * n->T;
    if (n->T > 256) { set_error("dali_vit_create: %d tokens exceed the attention kernel's limit of 256", n->T); delete n; return DALI_ERR_LIMIT; }
    const int C = n->C, Kp = 3 * cfg->patch * cfg->patch;
    n->cls_off = addt(n->params, n->param_elems, "base.cls_token", {1, 1, C});
    n->pos_off = addt(n->params, n->param_elems, "base.pos_embed", {1, n->T, C});
    n->patch.K = Kp; n->patch.N = C;
    n->patch.w_off = addt(n->params, n->param_elems, "base.patch_embed.proj.weight", {C, 3, cfg->patch, cfg->patch});
    n->patch.b_off = addt(n->params, n->param_elems, "base.patch_embed.proj.bias", {C});
    n->blocks.resize(cfg->depth);
    for (int i = 0; i < cfg->depth; ++i) {
        VBlock& b = n->blocks[i];
        const std::string pre = "base.blocks." + std::to_string(i);
        add_ln(n, b.n1, pre + ".norm1", C);
        add_lin(n, b.qkv, pre + ".attn.qkv", C, 3 * C);
        add_lin(n, b.proj, pre + ".attn.proj", C, C);
        add_ln(n, b.n2, pre + ".norm2", C);
        add_lin(n, b.fc1, pre + ".mlp.fc1", C, cfg->mlp_hidden);
        add_lin(n, b.fc2, pre + ".mlp.fc2", cfg->mlp_hidden, C);
    }
    add_ln(n, n->fin, "base.norm", C);
    n->fcw_off = addt(n->params, n->param_elems, "base.fc.weight", {cfg->num_classes, C});
    n->fcb_off = addt(n->params, n->param_elems, "base.fc.bias", {cfg->num_classes});
    n->neck_g = addt(n->params, n->param_elems, "bottleneck.weight", {C});
    n->neck_b = addt(n->params, n->param_elems, "bottleneck.bias", {C});
    n->neck_rm = addt(n->buffers, n->buffer_elems, "bottleneck.running_mean", {C});
    n->neck_rv = addt(n->buffers, n->buffer_elems, "bottleneck.running_var", {C});
    // backward stages = gradient buckets of the data-parallel reducer: up to 4 groups of consecutive blocks, last blocks first
    n->n_stages = cfg->depth >= 4 ? 4 : cfg->depth;
    n->stage_first_block.resize(n->n_stages + 1);
    for (int s = 0; s <= n->n_stages; ++s) n->stage_first_block[s] = cfg->depth - (int)((int64_t)cfg->depth * s / n->n_stages);

    VArena a;
    const size_t rows = n->rows, Hd = cfg->mlp_hidden;
    rsv(n, a, n->wbf16, (size_t)n->param_elems * 2);
    rsv(n, a, n->patches, (size_t)n->B * n->np * Kp * 2);
    rsv(n, a, n->pe, (size_t)n->B * n->np * C * 2);
    rsv(n, a, n->x0, rows * C * 2);
    rsv(n, a, n->patch.wt, (size_t)Kp * C * 2);          // unused (images need no gradient) but keeps Lin uniform
    size_t slab = linear_wgrad_slab_bytes(n->B * n->np, Kp, C);
    size_t part = std::max(layernorm_bwd_partial_floats((int)rows, C), colsum_partial_floats((int)rows, (int)std::max<size_t>(Hd, 3 * C))) * 4;
    for (auto& b : n->blocks) {
        rsv(n, a, b.h1, rows * C * 2); rsv(n, a, b.qkv_o, rows * 3 * C * 2); rsv(n, a, b.att, rows * C * 2);
        rsv(n, a, b.x_mid, rows * C * 2); rsv(n, a, b.h2, rows * C * 2); rsv(n, a, b.pre1, rows * Hd * 2);
        rsv(n, a, b.act1, rows * Hd * 2); rsv(n, a, b.x_out, rows * C * 2);
        rsv(n, a, b.lse, (size_t)n->B * n->H * n->T * 4);
        rsv(n, a, b.n1.mean, rows * 4); rsv(n, a, b.n1.rstd, rows * 4); rsv(n, a, b.n2.mean, rows * 4); rsv(n, a, b.n2.rstd, rows * 4);
        Lin* ls[4] = {&b.qkv, &b.proj, &b.fc1, &b.fc2};
        for (Lin* l : ls) {
            rsv(n, a, l->wt, (size_t)l->K * l->N * 2);
            slab = std::max(slab, linear_wgrad_slab_bytes((int)rows, l->K, l->N));
        }
    }
    rsv(n, a, n->cls_rows, (size_t)n->B * C * 2); rsv(n, a, n->dcls_rows, (size_t)n->B * C * 2); rsv(n, a, n->dgf16, (size_t)n->B * C * 2);
    rsv(n, a, n->gf, (size_t)n->B * C * 4); rsv(n, a, n->dgf, (size_t)n->B * C * 4);
    rsv(n, a, n->neck_mean, C * 4); rsv(n, a, n->neck_invstd, C * 4);
    rsv(n, a, n->fin_mean, n->B * 4); rsv(n, a, n->fin_rstd, n->B * 4);
    rsv(n, a, n->slab, slab); rsv(n, a, n->partial, part);
    rsv(n, a, n->dp_rows, (size_t)2 * cfg->depth * rows * 4);
    rsv(n, a, n->scratch, reduce_scratch_bytes((int)std::max<size_t>(Hd, 3 * C), 2));
    for (int i = 0; i < 4; ++i) rsv(n, a, n->gbuf[i], rows * std::max<size_t>(Hd, 3 * C) * 2);
    n->arena_bytes = a.used;
    *out = n;
    return DALI_OK;
}

extern "C" int dali_vit_destroy(dali_vit* n) { delete n; return DALI_OK; }

extern "C" int dali_vit_sizes(const dali_vit* n, int64_t* param_elems, int64_t* buffer_elems, int64_t* arena_bytes, int* feat_dim,
                              int* n_params, int* n_buffers) {
    DALI_REQUIRE(n, "dali_vit_sizes: null net");
    if (param_elems) *param_elems = n->param_elems;
    if (buffer_elems) *buffer_elems = n->buffer_elems;
    if (arena_bytes) *arena_bytes = (int64_t)n->arena_bytes;
    if (feat_dim) *feat_dim = n->C;
    if (n_params) *n_params = (int)n->params.size();
    if (n_buffers) *n_buffers = (int)n->buffers.size();
    return DALI_OK;
}
extern "C" int dali_vit_tensor_info(const dali_vit* n, int kind, int index, char* name, int name_cap, int64_t* offset, int64_t* numel,
                                    int* shape4, int* ndim) {
    DALI_REQUIRE(n && name && offset && numel && shape4 && ndim, "dali_vit_tensor_info: null argument");
    const auto& v = kind == 0 ? n->params : n->buffers;
    DALI_REQUIRE(index >= 0 && index < (int)v.size(), "dali_vit_tensor_info: index %d out of range", index);
    snprintf(name, name_cap, "%s", v[index].name.c_str());
    *offset = v[index].offset; *numel = v[index].numel; *ndim = v[index].ndim;
    for (int i = 0; i < 4; ++i) shape4[i] = v[index].shape[i];
    return DALI_OK;
}
extern "C" int dali_vit_bind(dali_vit* n, float* params, float* grads, float* buffers, void* arena, size_t arena_bytes) {
    DALI_REQUIRE(n && params && buffers && arena, "dali_vit_bind: null argument");
    DALI_REQUIRE(arena_bytes >= n->arena_bytes, "dali_vit_bind: arena too small (%zu < %zu)", arena_bytes, n->arena_bytes);
    n->P = params; n->G = grads; n->Bf = buffers; n->arena = static_cast<char*>(arena);
    for (auto& f : n->fixups) *f.first = n->arena + f.second;
    n->patch.w = n->wbf16 + n->patch.w_off;
    for (auto& b : n->blocks) { b.qkv.w = n->wbf16 + b.qkv.w_off; b.proj.w = n->wbf16 + b.proj.w_off; b.fc1.w = n->wbf16 + b.fc1.w_off; b.fc2.w = n->wbf16 + b.fc2.w_off; }
    return DALI_OK;
}
extern "C" int dali_vit_refresh_weights(dali_vit* n, void* stream) {
    DALI_REQUIRE(n && n->P, "dali_vit_refresh_weights: net not bound");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_cast_bf16(st, n->P, (size_t)n->param_elems, n->wbf16);
    if (rc) return rc;
    std::vector<TransposeJob> jobs;                      // every linear's dgrad image [K][N], batched launches (48 single launches cost 0.3 ms)
    for (auto& b : n->blocks) {
        Lin* ls[4] = {&b.qkv, &b.proj, &b.fc1, &b.fc2};
        for (Lin* l : ls) jobs.push_back(TransposeJob{l->w, l->wt, l->N, 1, l->K, 0});          // [N][K] -> [K][N]
    }
    return launch_weight_transpose_batched(st, jobs.data(), (int)jobs.size());
}

extern "C" int dali_vit_forward(dali_vit* n, void* stream, const float* images, int training, float* feat, float* global_feat) {
    DALI_REQUIRE(n && n->P && images && feat, "dali_vit_forward: null argument or net not bound");
    hipStream_t st = (hipStream_t)stream;
    const int C = n->C, rows = n->rows, Hd = n->cfg.mlp_hidden;
    const float eps = 1e-6f, scale = 0.125f;           // head_dim 64 -> 64^-0.5
    n->fwd_training = training != 0;
    int rc;
    if ((rc = launch_patchify(st, images, n->B, n->cfg.height, n->cfg.width, n->cfg.patch, n->cfg.stride, n->patches))) return rc;
    if ((rc = launch_linear_fwd(st, n->patches, n->patch.w, n->P + n->patch.b_off, 0, nullptr, n->pe, nullptr, nullptr, n->B * n->np, n->patch.K, C))) return rc;
    if ((rc = launch_assemble_tokens(st, n->pe, n->P + n->cls_off, n->P + n->pos_off, n->B, n->T, C, n->x0))) return rc;
    const uint16_t* x = n->x0;
    const float* dp = training ? n->dp_scale : nullptr;        // DropPath is the identity in eval mode (vit_pytorch.py:58)
    n->dp_used = dp;
    if (dp && (rc = launch_expand_rowscale(st, dp, 2 * (int)n->blocks.size(), n->B, n->T, n->dp_rows))) return rc;
    for (int bi = 0; bi < (int)n->blocks.size(); ++bi) {
        VBlock& b = n->blocks[bi];
        b.x_in = const_cast<uint16_t*>(x);
        if ((rc = launch_layernorm_fwd(st, x, n->P + b.n1.g_off, n->P + b.n1.b_off, rows, C, eps, b.h1, b.n1.mean, b.n1.rstd, nullptr))) return rc;
        if ((rc = launch_linear_fwd(st, b.h1, b.qkv.w, n->P + b.qkv.b_off, 0, nullptr, b.qkv_o, nullptr, nullptr, rows, C, 3 * C))) return rc;
        if ((rc = launch_attention_fwd(st, b.qkv_o, n->B, n->T, n->H, scale, b.att, b.lse))) return rc;
        // x_mid = x + s[b] * proj(...): the per-row factor rides in the GEMM's epilogue (IGemmArgs::row_scale)
        if ((rc = launch_linear_fwd(st, b.att, b.proj.w, n->P + b.proj.b_off, 0, x, b.x_mid, nullptr, nullptr, rows, C, C,
                                    dp ? n->dp_rows + (size_t)(2 * bi) * rows : nullptr))) return rc;
        if ((rc = launch_layernorm_fwd(st, b.x_mid, n->P + b.n2.g_off, n->P + b.n2.b_off, rows, C, eps, b.h2, b.n2.mean, b.n2.rstd, nullptr))) return rc;
        if ((rc = launch_linear_fwd(st, b.h2, b.fc1.w, n->P + b.fc1.b_off, 1, nullptr, b.act1, b.pre1, nullptr, rows, C, Hd))) return rc;
        if ((rc = launch_linear_fwd(st, b.act1, b.fc2.w, n->P + b.fc2.b_off, 0, b.x_mid, b.x_out, nullptr, nullptr, rows, Hd, C,
                                    dp ? n->dp_rows + (size_t)(2 * bi + 1) * rows : nullptr))) return rc;
        x = b.x_out;
    }
    // final LayerNorm on the cls rows only (x[:, 0], vit_pytorch.py:401-403), then the BN neck (make_models.py:187)
    DALI_HIP(hipMemcpy2DAsync(n->cls_rows, (size_t)C * 2, x, (size_t)n->T * C * 2, (size_t)C * 2, n->B, hipMemcpyDeviceToDevice, st));
    if ((rc = launch_layernorm_fwd(st, n->cls_rows, n->P + n->fin.g_off, n->P + n->fin.b_off, n->B, C, eps, nullptr, n->fin_mean, n->fin_rstd, n->gf))) return rc;
    if (global_feat) DALI_HIP(hipMemcpyAsync(global_feat, n->gf, (size_t)n->B * C * 4, hipMemcpyDeviceToDevice, st));
    return launch_bn1d_fwd(st, n->gf, n->B, C, n->P + n->neck_g, n->P + n->neck_b, n->Bf + n->neck_rm, n->Bf + n->neck_rv, training ? 1 : 0, 0.1f,
                           1e-5f, feat, n->neck_mean, n->neck_invstd);
}

namespace {
int lin_bwd(dali_vit* n, hipStream_t st, const Lin& l, const uint16_t* x, const uint16_t* dy, const uint16_t* gelu_pre, uint16_t* dx, int rows) {
    int rc;
    bool bias_done = false;                                       // the bias gradient rides on the weight-gradient GEMM where the kernel supports it
    if ((rc = launch_linear_wgrad(st, x, dy, n->G + l.w_off, rows, l.K, l.N, n->slab, n->G + l.b_off, n->partial, &bias_done))) return rc;
    if (!bias_done && (rc = launch_colsum(st, dy, rows, l.N, n->G + l.b_off, n->partial, n->scratch))) return rc;
    if (dx) return launch_linear_fwd(st, dy, l.wt, nullptr, 0, nullptr, dx, nullptr, gelu_pre, rows, l.N, l.K);
    return DALI_OK;
}
}  // namespace

extern "C" int dali_vit_set_drop_path(dali_vit* n, const float* scales) {
    DALI_REQUIRE(n, "dali_vit_set_drop_path: null net");
    n->dp_scale = scales;
    return DALI_OK;
}

extern "C" int dali_vit_num_stages(const dali_vit* n) { return n ? n->n_stages : 0; }

// Flat-parameter element range whose gradients are complete once dali_vit_backward_stages has run through `stage`:
// stage 0 = final norm + fc + neck + the last group of blocks, ..., the last stage also holds cls / pos / patch embedding.
extern "C" int dali_vit_stage_param_range(const dali_vit* n, int stage, int64_t* begin, int64_t* end) {
    DALI_REQUIRE(n && begin && end && stage >= 0 && stage < n->n_stages, "dali_vit_stage_param_range: bad argument");
    const int lo = n->stage_first_block[stage + 1], hi = n->stage_first_block[stage];
    *begin = (stage == n->n_stages - 1) ? 0 : n->blocks[lo].n1.g_off;
    *end = (stage == 0) ? n->param_elems : n->blocks[hi].n1.g_off;
    return DALI_OK;
}

extern "C" int dali_vit_backward_stages(dali_vit* n, void* stream, const float* d_feat, int stage_begin, int stage_end) {
    DALI_REQUIRE(n && n->P && n->G, "dali_vit_backward: null argument or net not bound");
    DALI_REQUIRE(n->fwd_training, "dali_vit_backward: the last forward was not in training mode");
    DALI_REQUIRE(stage_begin >= 0 && stage_end < n->n_stages && stage_begin <= stage_end, "dali_vit_backward: bad stage range %d..%d", stage_begin, stage_end);
    hipStream_t st = (hipStream_t)stream;
    const int C = n->C, rows = n->rows;
    const float* dp = n->dp_used;
    int rc;
    uint16_t* dx = n->gbuf[0];
    for (int stage = stage_begin; stage <= stage_end; ++stage) {
        if (stage == 0) {
            DALI_REQUIRE(d_feat != nullptr, "dali_vit_backward: d_feat is null");
            // neck (bias frozen, make_models.py:181: no dbeta) + final LayerNorm (cls rows)
            if ((rc = launch_bn1d_bwd(st, n->gf, d_feat, n->B, C, n->P + n->neck_g, n->neck_mean, n->neck_invstd, n->dgf, n->G + n->neck_g, nullptr))) return rc;
            if ((rc = launch_layernorm_bwd(st, nullptr, n->cls_rows, n->P + n->fin.g_off, n->fin_mean, n->fin_rstd, nullptr, n->B, C, n->dcls_rows,
                                           n->G + n->fin.g_off, n->G + n->fin.b_off, n->partial, n->scratch, n->dgf))) return rc;
            DALI_HIP(hipMemsetAsync(dx, 0, (size_t)rows * C * 2, st));
            DALI_HIP(hipMemcpy2DAsync(dx, (size_t)n->T * C * 2, n->dcls_rows, (size_t)C * 2, (size_t)C * 2, n->B, hipMemcpyDeviceToDevice, st));
        }
        for (int i = n->stage_first_block[stage] - 1; i >= n->stage_first_block[stage + 1]; --i) {
            VBlock& b = n->blocks[i];
            uint16_t* t1 = n->gbuf[1]; uint16_t* t2 = n->gbuf[2]; uint16_t* t3 = n->gbuf[3];
            const float* s_att = dp ? dp + (size_t)(2 * i) * n->B : nullptr;
            const float* s_mlp = dp ? dp + (size_t)(2 * i + 1) * n->B : nullptr;
            // x_out = x_mid + s_mlp * fc2(gelu(fc1(LN2(x_mid))))
            const uint16_t* d_branch = dx;
            if (dp) { if ((rc = launch_rowscale_add(st, dx, s_mlp, n->B, n->T, C, nullptr, t2))) return rc; d_branch = t2; }
            if ((rc = lin_bwd(n, st, b.fc2, b.act1, d_branch, b.pre1, t1, rows))) return rc;           // t1 = d_pre1 (GELU' fused)
            if ((rc = lin_bwd(n, st, b.fc1, b.h2, t1, nullptr, t2, rows))) return rc;                  // t2 = d_h2
            if ((rc = launch_layernorm_bwd(st, t2, b.x_mid, n->P + b.n2.g_off, b.n2.mean, b.n2.rstd, dx, rows, C, t3, n->G + b.n2.g_off,
                                           n->G + b.n2.b_off, n->partial, n->scratch))) return rc;     // t3 = dx_mid
            // x_mid = x_in + s_att * proj(attn(qkv(LN1(x_in))))
            d_branch = t3;
            if (dp) { if ((rc = launch_rowscale_add(st, t3, s_att, n->B, n->T, C, nullptr, t2))) return rc; d_branch = t2; }
            if ((rc = lin_bwd(n, st, b.proj, b.att, d_branch, nullptr, t1, rows))) return rc;          // t1 = d_att
            if ((rc = launch_attention_bwd(st, b.qkv_o, b.att, t1, b.lse, n->B, n->T, n->H, 0.125f, t2))) return rc;   // t2 = d_qkv
            if ((rc = lin_bwd(n, st, b.qkv, b.h1, t2, nullptr, t1, rows))) return rc;                  // t1 = d_h1
            if ((rc = launch_layernorm_bwd(st, t1, b.x_in, n->P + b.n1.g_off, b.n1.mean, b.n1.rstd, t3, rows, C, dx, n->G + b.n1.g_off,
                                           n->G + b.n1.b_off, n->partial, n->scratch))) return rc;     // dx = dx_in
        }
        if (stage == n->n_stages - 1) {
            // tokens: d pos_embed, d cls_token, d patch embedding
            if ((rc = launch_assemble_tokens_bwd(st, dx, n->B, n->T, C, n->G + n->pos_off, n->G + n->cls_off, n->gbuf[1]))) return rc;
            if ((rc = lin_bwd(n, st, n->patch, n->patches, n->gbuf[1], nullptr, nullptr, n->B * n->np))) return rc;
        }
    }
    return DALI_OK;
}

extern "C" int dali_vit_backward(dali_vit* n, void* stream, const float* d_feat) {
    DALI_REQUIRE(n && d_feat, "dali_vit_backward: null argument");
    return dali_vit_backward_stages(n, stream, d_feat, 0, n->n_stages - 1);
}
