// resnet_plan.hip -- native executor of the ResNet-50-ReID embedding network
// (Encoders.ResNet50ReID, Encoders.py:306-351, over torchvision's ResNet-50 v1.5 bottleneck trunk).
//
// The plan owns the topology and the launch sequence (forward, backward per stage); PyTorch owns the storage:
// one flat fp32 parameter buffer, one flat fp32 gradient buffer, one flat fp32 buffer of BatchNorm running
// statistics and one byte arena for activations / bf16 weight images / scratch.  Tensor order and names follow
// torchvision's state_dict so the Python mirror exposes the reference's keys.
//
// ReID edits reproduced (Encoders.py:321-322, :334, :341-350): no ReLU after the stem BN, layer4 at stride 1,
// head = global avg-pool + global max-pool -> BatchNorm1d.
#include "kernels.h"
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace dali;

namespace {

struct TensorInfo { std::string name; int64_t offset, numel; int shape[4]; int ndim; };

struct Conv {
    int cin, cout, r, s, stride, pad, hin, win, hout, wout;
    int64_t w_off;            // element offset of the fp32 [cout][r][s][cin] weight in the flat params
    uint16_t* w_bf16;         // [cout][r*s*cin] (arena)
    uint16_t* wt_bf16;        // [cin][r*s*cout] (arena), dgrad image
};
struct Bn {
    int C;
    int64_t g_off, b_off;     // flat params
    int64_t rm_off, rv_off;   // flat buffers
    float *scale, *shift, *mean, *invstd, *coef;   // arena, [C] each (coef [C][3])
};
struct Block {
    Conv c1, c2, c3, cd;
    Bn b1, b2, b3, bd;
    bool has_ds;
    int hin, win, hout, wout, cin, width, cout;
    uint8_t* ybits = nullptr;                                  // ReLU mask of y, one bit per element (dali_bn_act mask_out)
    uint16_t *x, *raw1, *a1, *raw2, *a2, *raw3, *rawd, *y;    // a = relu(bn(raw)), materialised once (fused
                                                                // apply-on-load repeats the VALU work per tap and per M-tile)
    // bn3 behind conv3 through the moments of a2 (bnlin.hip): blocks without a downsample branch never store raw3
    bool lin3 = false;
    float *gram = nullptr, *m2 = nullptr, *sdz = nullptr, *bvec = nullptr, *qk = nullptr;      // [w][w], [w], [cout], [w], [2][cout]
    float *ut = nullptr, *dot = nullptr;                       // (W3 gram)^T [w][cout] (forward -> backward), [w/32][cout] scratch
    uint16_t* wd1 = nullptr;                                   // data-gradient image [w][cout + w]: (A.W3)^T beside -(W3^T diag(Q) W3)
    // the downsample BatchNorm's backward through the moments of the block input x (same scheme, backward only: rawd is still what the forward
    // stores and conv3's epilogue reads): no reduce / apply passes over (dz, rawd), d_rawd is never formed
    bool lin_ds = false;
    float *gram_d = nullptr, *m2_d = nullptr, *ut_d = nullptr, *bvec_d = nullptr, *qk_d = nullptr;   // [cin][cin], [cin], [cin][cout], [cin], [2][cout]
    uint16_t* wdd = nullptr;                                   // [cin][cout + cin]: (A_d.Wd)^T beside -(Wd^T diag(Q_d) Wd)
    // inference: conv3 + the (stride-1) downsample convolution as one GEMM over [a2 | x] (dali_conv1x1_cat_act): the folded weight image and shift
    bool cat_eval = false;
    uint16_t* wcat = nullptr;
    float* shcat = nullptr;
    uint16_t* dbg_d_raw1 = nullptr;                            // where the last backward of this block left d(raw1) (scratch: valid until the next block runs)
};

struct Arena {
    size_t used = 0;
    size_t take(size_t bytes) { size_t o = used; used = align_up(used + bytes, 256); return o; }
};

}  // namespace

struct dali_resnet {
    dali_ctx* ctx;
    dali_resnet_cfg cfg;
    int N, H, W;
    std::vector<TensorInfo> params, buffers;
    int64_t param_elems = 0, buffer_elems = 0;
    Conv stem; Bn stem_bn; Bn neck;
    int stem_h, stem_w, pool_h, pool_w, feat_dim, head_hw;
    std::vector<Block> blocks;
    int stage_first[4], stage_last[4];       // block index ranges per layer
    // arena offsets resolved at bind()
    size_t arena_bytes = 0;
    std::vector<std::pair<void**, size_t>> fixups;
    // arena pointers
    uint16_t *ximg = nullptr, *raw0 = nullptr, *pool0 = nullptr, *w_stem = nullptr;
    uint8_t* pool_arg = nullptr;
    float *feat = nullptr, *emb_in = nullptr, *stat_partial = nullptr, *bwd_partial = nullptr, *wgrad_slab = nullptr, *stem_dw_pad = nullptr;
    float *dfeat = nullptr, *neck_mean = nullptr, *neck_invstd = nullptr, *cs_partial = nullptr;
    double* red_scratch = nullptr;
    int16_t* head_arg = nullptr;
    uint16_t* gbuf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint16_t* wbf16_flat = nullptr;          // bf16 cast of the whole flat parameter buffer
    // bound storages
    float *P = nullptr, *G = nullptr, *B = nullptr;
    char* arena = nullptr;
    // backward state
    uint16_t* cur_dy = nullptr;
    int64_t cur_dy_bytes = 0;
    bool fwd_training = false;
    int feature_mode = DALI_FEATURE_BOTH;
};

namespace {

int64_t add_tensor(std::vector<TensorInfo>& v, int64_t& total, const std::string& name, std::initializer_list<int> shape) {
    TensorInfo t;
    t.name = name;
    t.ndim = (int)shape.size();
    t.numel = 1;
    int i = 0;
    for (int s : shape) { t.shape[i++] = s; t.numel *= s; }
    for (; i < 4; ++i) t.shape[i] = 1;
    t.offset = total;
    total += (t.numel + 63) / 64 * 64;          // 256-byte aligned segments
    v.push_back(t);
    return t.offset;
}

void add_conv(dali_resnet* net, Conv& c, const std::string& name, int cin, int cout, int k, int stride, int pad, int hin, int win) {
    c.cin = cin; c.cout = cout; c.r = k; c.s = k; c.stride = stride; c.pad = pad; c.hin = hin; c.win = win;
    c.hout = (hin + 2 * pad - k) / stride + 1;
    c.wout = (win + 2 * pad - k) / stride + 1;
    c.w_off = add_tensor(net->params, net->param_elems, name + ".weight", {cout, cin, k, k});   // logical OIHW, stored OHWI
    c.w_bf16 = nullptr; c.wt_bf16 = nullptr;
}
void add_bn(dali_resnet* net, Bn& b, const std::string& name, int C) {
    b.C = C;
    b.g_off = add_tensor(net->params, net->param_elems, name + ".weight", {C});
    b.b_off = add_tensor(net->params, net->param_elems, name + ".bias", {C});
    b.rm_off = add_tensor(net->buffers, net->buffer_elems, name + ".running_mean", {C});
    b.rv_off = add_tensor(net->buffers, net->buffer_elems, name + ".running_var", {C});
}

template <class T>
void reserve(dali_resnet* net, Arena& a, T*& ptr, size_t bytes) {
    net->fixups.emplace_back(reinterpret_cast<void**>(&ptr), a.take(bytes));
}
void reserve_bn(dali_resnet* net, Arena& a, Bn& b) {
    reserve(net, a, b.scale, b.C * 4); reserve(net, a, b.shift, b.C * 4);
    reserve(net, a, b.mean, b.C * 4); reserve(net, a, b.invstd, b.C * 4);
    reserve(net, a, b.coef, b.C * 12);
}

// output width if conv c is a 3x3 / stride 1 / pad 1 convolution on a power-of-two pixel grid (the halo wgrad kernel), else 0
int conv_halo_w(const Conv& c) {
    const bool pow2 = (c.wout & (c.wout - 1)) == 0 && ((c.hout * c.wout) & (c.hout * c.wout - 1)) == 0;
    return (c.r == 3 && c.s == 3 && c.stride == 1 && c.pad == 1 && pow2 && c.hout * c.wout >= 128) ? c.wout : 0;
}
GatherGeom conv_geom(const Conv& c, int mode) {
    GatherGeom g{};
    if (mode == 0) {
        g.Hout = c.hout; g.Wout = c.wout; g.Hin = c.hin; g.Win = c.win; g.Ck = c.cin;
    } else {
        g.Hout = c.hin; g.Wout = c.win; g.Hin = c.hout; g.Win = c.wout; g.Ck = c.cout;
    }
    g.R = c.r; g.S = c.s; g.stride = c.stride; g.pad = c.pad; g.mode = mode;
    g.pix_pitch = g.Ck; g.row_pitch = g.Win * g.Ck; g.img_pitch = (long long)g.Hin * g.Win * g.Ck;
    g.lw = g.lhw = -1;
    return g;
}
// bn3 behind conv3 through the moments of conv3's input (bnlin.hip).  The scheme trades passes over [P][4w] tensors for products of size
// w^2 and ~7 small launches per block.  Until round 4 it paid for w <= 128 only (layer1 / layer2); with the block's two data-gradient GEMMs
// merged into one launch over [dz | a2] (IGemmArgs::X2) it pays for every block of ResNet-50 at batch 256: 15.71 (w <= 128) / 15.61 (<= 256) /
// 15.47 ms (all), same box.  DALI_BNLIN_MAXW moves the limit (0: every block keeps the materialised form).
int bnlin_max_width() {
    static int v = -1;
    if (v == -1) { const char* e = getenv("DALI_BNLIN_MAXW"); v = e ? atoi(e) : 512; }
    return v;
}
// inference forward with BatchNorm + ReLU folded into the convolutions' output stages (default); DALI_EVAL_FUSED=0 keeps the training dataflow
// (raw output, then a bn_act pass) for A/B measurements and for tests of the unfused rounding points
bool eval_fused() { return DALI_ENV_INT("DALI_EVAL_FUSED", 1) != 0; }
// a cin = cout = w 1x1 convolution on the grid of conv3: the shape of the Gram GEMM a2^T a2 and of the second data-gradient GEMM
Conv square_conv(const Conv& c3) {
    Conv q = c3;
    q.cin = c3.cin; q.cout = c3.cin; q.w_off = 0; q.w_bf16 = nullptr; q.wt_bf16 = nullptr;
    return q;
}
// the stem reads the packed [N][H+6][W+8][4] image: one tap row (8 taps x 4 ch) per k-tile
GatherGeom stem_geom(const dali_resnet* net) {
    GatherGeom g{};
    g.Hout = net->stem_h; g.Wout = net->stem_w;
    g.Hin = net->H + 6; g.Win = net->W + 8;
    g.Ck = 32; g.R = 7; g.S = 1; g.stride = 2; g.pad = 0; g.mode = 0;
    g.pix_pitch = 4; g.row_pitch = (net->W + 8) * 4; g.img_pitch = (long long)(net->H + 6) * (net->W + 8) * 4;
    g.lw = g.lhw = -1;
    return g;
}

}  // namespace

extern "C" int dali_resnet_create(dali_ctx* ctx, const dali_resnet_cfg* cfg, dali_resnet** out) {
    DALI_REQUIRE(ctx && cfg && out, "dali_resnet_create: null argument");
    DALI_REQUIRE(cfg->batch > 0 && cfg->height >= 32 && cfg->width >= 32 && cfg->height % 32 == 0 && cfg->width % 16 == 0,
                 "dali_resnet_create: bad input shape %dx%dx%d (height %% 32, width %% 16)", cfg->batch, cfg->height, cfg->width);
    DALI_REQUIRE(cfg->width_base >= 32 && cfg->width_base % 32 == 0 && cfg->width_base <= 64,
                 "dali_resnet_create: width_base must be 32 or 64 (got %d)", cfg->width_base);
    for (int i = 0; i < 4; ++i) DALI_REQUIRE(cfg->layers[i] >= 1, "dali_resnet_create: layers[%d] < 1", i);
    dali_resnet* net = new (std::nothrow) dali_resnet();
    if (!net) { set_error("dali_resnet_create: out of host memory"); return DALI_ERR_NOMEM; }
    net->ctx = ctx; net->cfg = *cfg; net->N = cfg->batch; net->H = cfg->height; net->W = cfg->width;
    const int wb = cfg->width_base;
    // ---- topology + parameter table (torchvision order) ----
    add_conv(net, net->stem, "conv1", 3, wb, 7, 2, 3, net->H, net->W);
    add_bn(net, net->stem_bn, "bn1", wb);
    net->stem_h = net->stem.hout; net->stem_w = net->stem.wout;
    net->pool_h = (net->stem_h + 2 - 3) / 2 + 1; net->pool_w = (net->stem_w + 2 - 3) / 2 + 1;
    int h = net->pool_h, w = net->pool_w, inpl = wb;
    const int strides[4] = {1, 2, 2, 1};              // layer4 at stride 1 (Encoders.py:321-322)
    for (int li = 0; li < 4; ++li) {
        const int planes = wb << li;
        net->stage_first[li] = (int)net->blocks.size();
        for (int bi = 0; bi < cfg->layers[li]; ++bi) {
            Block b{};
            const int st = bi == 0 ? strides[li] : 1;
            const std::string pre = "layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            b.hin = h; b.win = w; b.cin = inpl; b.width = planes; b.cout = planes * 4;
            add_conv(net, b.c1, pre + ".conv1", inpl, planes, 1, 1, 0, h, w);
            add_bn(net, b.b1, pre + ".bn1", planes);
            add_conv(net, b.c2, pre + ".conv2", planes, planes, 3, st, 1, h, w);
            add_bn(net, b.b2, pre + ".bn2", planes);
            add_conv(net, b.c3, pre + ".conv3", planes, planes * 4, 1, 1, 0, b.c2.hout, b.c2.wout);
            add_bn(net, b.b3, pre + ".bn3", planes * 4);
            b.has_ds = (bi == 0);                     // torchvision: stride != 1 or inplanes != planes*4; true for every first block
            if (b.has_ds) {
                add_conv(net, b.cd, pre + ".downsample.0", inpl, planes * 4, 1, st, 0, h, w);
                add_bn(net, b.bd, pre + ".downsample.1", planes * 4);
            }
            b.hout = b.c2.hout; b.wout = b.c2.wout;
            // bn3 through the moments of a2 (bnlin.hip); in a block with a downsample branch the identity enters conv3's epilogue as
            // scale_d * rawd + shift_d (res_scale / bias) and the downsample BatchNorm keeps its own two-pass backward
            b.lin3 = planes % 32 == 0 && planes <= bnlin_max_width();
            // ... except where the branch is a stride-1 convolution of a narrow input (ResNet-50: layer1's, 64 -> 256 over 524288 pixels at batch
            // 256), whose Gram matrix costs less than the two passes: there its backward goes through the moments of x as well (block_backward)
            b.lin_ds = b.has_ds && b.lin3 && st == 1 && inpl % 32 == 0 && inpl <= 128 && net->blocks.empty();   // (first block of the net: no mask on dx)
            h = b.hout; w = b.wout; inpl = planes * 4;
            net->blocks.push_back(b);
        }
        net->stage_last[li] = (int)net->blocks.size() - 1;
    }
    net->feat_dim = inpl;
    net->head_hw = h * w;
    DALI_REQUIRE(net->head_hw < 32768, "dali_resnet_create: final map too large");
    add_bn(net, net->neck, "last_bn", net->feat_dim);

    // ---- arena layout ----
    Arena a;
    const size_t N = net->N;
    reserve(net, a, net->wbf16_flat, (size_t)net->param_elems * 2);
    reserve(net, a, net->w_stem, (size_t)wb * 224 * 2);
    reserve(net, a, net->ximg, N * (net->H + 6) * (net->W + 8) * 4 * 2);
    const size_t raw0_bytes = N * net->stem_h * net->stem_w * wb * 2;
    reserve(net, a, net->raw0, raw0_bytes);
    reserve(net, a, net->pool0, N * net->pool_h * net->pool_w * wb * 2);
    reserve(net, a, net->pool_arg, N * net->pool_h * net->pool_w * wb);
    reserve_bn(net, a, net->stem_bn);
    size_t max_act = raw0_bytes, max_stat = (size_t)igemm_conv_stat_tiles(wb, (int)N * net->stem_h * net->stem_w, 224) * wb * 2 * 4;
    size_t max_bwd_partial = bn_bwd_partial_floats((int)N * net->stem_h * net->stem_w, wb, false) * 4;
    size_t max_slab = 0, max_cs = 256;
    {
        int sp, pps; size_t wsb;
        wgrad_plan(wb, 224, (int)N * net->stem_h * net->stem_w, 512, &sp, &pps, &wsb, 7);
        max_slab = wsb;
    }
    for (auto& b : net->blocks) {
        const size_t pin = N * b.hin * b.win, pout = N * b.hout * b.wout;
        reserve(net, a, b.raw1, pin * b.width * 2);
        reserve(net, a, b.a1, pin * b.width * 2);
        reserve(net, a, b.raw2, pout * b.width * 2);
        reserve(net, a, b.a2, pout * b.width * 2);
        if (!b.lin3) reserve(net, a, b.raw3, pout * b.cout * 2);
        else {
            reserve(net, a, b.gram, (size_t)b.width * b.width * 4); reserve(net, a, b.m2, (size_t)b.width * 4);
            reserve(net, a, b.sdz, (size_t)b.cout * 4); reserve(net, a, b.bvec, (size_t)b.width * 4); reserve(net, a, b.qk, (size_t)b.cout * 8);
            reserve(net, a, b.wd1, (size_t)b.width * (b.cout + b.width) * 2);          // [w][cout + w]: (A.W3)^T | -(W3^T diag(Q) W3), one data-gradient image
            reserve(net, a, b.ut, (size_t)b.width * b.cout * 4); reserve(net, a, b.dot, (size_t)(b.width / 32) * b.cout * 12);
            max_cs = std::max(max_cs, std::max(colsum_partial_floats((int)pout, b.cout), colsum_partial_floats((int)pout, b.width)) * 4);
            int sp, pps; size_t wsb;
            wgrad_plan(b.width, b.width, (int)pout, 512, &sp, &pps, &wsb, 1, 0);
            max_slab = std::max(max_slab, wsb);
            max_cs = std::max(max_cs, (size_t)wgrad_colsum_rows(b.width, b.width, 1, (int)pout, sp) * b.width * 4);     // per-split column sums riding on the GEMMs
            wgrad_plan(b.cout, b.width, (int)pout, 512, &sp, &pps, &wsb, 1, 0);
            max_cs = std::max(max_cs, (size_t)wgrad_colsum_rows(b.cout, b.width, 1, (int)pout, sp) * b.cout * 4);
        }
        reserve(net, a, b.y, pout * b.cout * 2);
        reserve(net, a, b.ybits, pout * b.cout / 8);
        if (b.has_ds) reserve(net, a, b.rawd, pout * b.cout * 2);
        if (b.lin_ds) {
            reserve(net, a, b.gram_d, (size_t)b.cin * b.cin * 4); reserve(net, a, b.m2_d, (size_t)b.cin * 4);
            reserve(net, a, b.ut_d, (size_t)b.cin * b.cout * 4); reserve(net, a, b.bvec_d, (size_t)b.cin * 4); reserve(net, a, b.qk_d, (size_t)b.cout * 8);
            reserve(net, a, b.wdd, (size_t)b.cin * (b.cout + b.cin) * 2);
            int sp, pps; size_t wsb;
            wgrad_plan(b.cin, b.cin, (int)pin, 512, &sp, &pps, &wsb, 1, 0);
            max_slab = std::max(max_slab, wsb);
            max_cs = std::max(max_cs, std::max(colsum_partial_floats((int)pin, b.cin), (size_t)wgrad_colsum_rows(b.cin, b.cin, 1, (int)pin, sp) * b.cin) * 4);
        }
        // (split weight image, 2 parts: folding the scales into ONE bf16 image is a coherent 2^-9 perturbation of the weights, which moved the mAP of
        //  separated identities by 1.1e-3 against the oracle; hi + lo images are fp32-grade.  K = 2 (w + cin) <= 256: layer1's first block.)
        b.cat_eval = b.has_ds && b.cd.stride == 1 && b.lin3 && conv_cat_act_supported(b.cout, b.width, b.cin, (int)pout, 2);
        if (b.cat_eval) { reserve(net, a, b.wcat, (size_t)b.cout * 2 * (b.width + b.cin) * 2); reserve(net, a, b.shcat, (size_t)b.cout * 4); }
        reserve_bn(net, a, b.b1); reserve_bn(net, a, b.b2); reserve_bn(net, a, b.b3);
        if (b.has_ds) reserve_bn(net, a, b.bd);
        Conv* cs[4] = {&b.c1, &b.c2, &b.c3, b.has_ds ? &b.cd : nullptr};
        for (Conv* c : cs) {
            if (!c) continue;
            reserve(net, a, c->wt_bf16, (size_t)c->cin * c->r * c->s * c->cout * 2);
            const int P = (int)N * c->hout * c->wout;
            max_stat = std::max(max_stat, (size_t)igemm_conv_stat_tiles(c->cout, P, c->r * c->s * c->cin) * c->cout * 2 * 4);
            int sp, pps; size_t wsb;
            wgrad_plan(c->cout, c->r * c->s * c->cin, P, 512, &sp, &pps, &wsb, c->r * c->s, conv_halo_w(*c));
            max_slab = std::max(max_slab, wsb);
            max_act = std::max(max_act, (size_t)P * c->cout * 2);
            max_act = std::max(max_act, N * c->hin * c->win * c->cin * 2);
            max_bwd_partial = std::max(max_bwd_partial, bn_bwd_partial_floats(P, c->cout, true) * 4);
        }
    }
    reserve(net, a, net->feat, N * net->feat_dim * 4);
    reserve(net, a, net->dfeat, N * net->feat_dim * 4);
    reserve(net, a, net->head_arg, N * net->feat_dim * 2);
    reserve(net, a, net->neck_mean, net->feat_dim * 4);
    reserve(net, a, net->neck_invstd, net->feat_dim * 4);
    reserve(net, a, net->stat_partial, max_stat);
    reserve(net, a, net->bwd_partial, max_bwd_partial);
    reserve(net, a, net->wgrad_slab, max_slab);
    reserve(net, a, net->cs_partial, max_cs);
    reserve(net, a, net->stem_dw_pad, (size_t)wb * 224 * 4);
    reserve(net, a, net->red_scratch, reduce_scratch_bytes(net->feat_dim, 3));
    for (int i = 0; i < 6; ++i) reserve(net, a, net->gbuf[i], max_act);
    net->arena_bytes = a.used;
    *out = net;
    return DALI_OK;
}

extern "C" int dali_resnet_destroy(dali_resnet* net) { delete net; return DALI_OK; }

extern "C" int dali_resnet_sizes(const dali_resnet* net, int64_t* param_elems, int64_t* buffer_elems, int64_t* arena_bytes,
                                 int* feat_dim, int* n_params, int* n_buffers) {
    DALI_REQUIRE(net, "dali_resnet_sizes: null net");
    if (param_elems) *param_elems = net->param_elems;
    if (buffer_elems) *buffer_elems = net->buffer_elems;
    if (arena_bytes) *arena_bytes = (int64_t)net->arena_bytes;
    if (feat_dim) *feat_dim = net->feat_dim;
    if (n_params) *n_params = (int)net->params.size();
    if (n_buffers) *n_buffers = (int)net->buffers.size();
    return DALI_OK;
}

extern "C" int dali_resnet_tensor_info(const dali_resnet* net, int kind, int index, char* name, int name_cap, int64_t* offset,
                                       int64_t* numel, int* shape4, int* ndim) {
    DALI_REQUIRE(net && name && offset && numel && shape4 && ndim, "dali_resnet_tensor_info: null argument");
    const auto& v = kind == 0 ? net->params : net->buffers;
    DALI_REQUIRE(index >= 0 && index < (int)v.size(), "dali_resnet_tensor_info: index %d out of range", index);
    const TensorInfo& t = v[index];
    snprintf(name, name_cap, "%s", t.name.c_str());
    *offset = t.offset; *numel = t.numel; *ndim = t.ndim;
    for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
    return DALI_OK;
}

// Stage s (0..3) covers layer(4-s); returns the flat-parameter element range whose gradients are complete once
// dali_resnet_backward has run through that stage (stage 0 also holds last_bn, stage 3 also the stem).
extern "C" int dali_resnet_stage_param_range(const dali_resnet* net, int stage, int64_t* begin, int64_t* end) {
    DALI_REQUIRE(net && begin && end && stage >= 0 && stage < 4, "dali_resnet_stage_param_range: bad argument");
    const int li = 3 - stage;
    const Block& first = net->blocks[net->stage_first[li]];
    *begin = (li == 0) ? 0 : first.c1.w_off;
    if (li == 3) *end = net->param_elems;
    else *end = net->blocks[net->stage_first[li + 1]].c1.w_off;
    return DALI_OK;
}

extern "C" int dali_resnet_bind(dali_resnet* net, float* params, float* grads, float* buffers, void* arena, size_t arena_bytes) {
    DALI_REQUIRE(net && params && buffers && arena, "dali_resnet_bind: null argument");
    DALI_REQUIRE(arena_bytes >= net->arena_bytes, "dali_resnet_bind: arena too small (%zu < %zu)", arena_bytes, net->arena_bytes);
    DALI_REQUIRE((reinterpret_cast<uintptr_t>(arena) & 255) == 0 && (reinterpret_cast<uintptr_t>(params) & 255) == 0 &&
                 (reinterpret_cast<uintptr_t>(grads) & 255) == 0 && (reinterpret_cast<uintptr_t>(buffers) & 255) == 0,
                 "dali_resnet_bind: storages must be 256-byte aligned");
    net->P = params; net->G = grads; net->B = buffers; net->arena = static_cast<char*>(arena);
    for (auto& f : net->fixups) *f.first = net->arena + f.second;
    // bf16 weight images alias the flat bf16 cast (same offsets as the fp32 flat buffer)
    auto bind_conv = [&](Conv& c) { c.w_bf16 = net->wbf16_flat + c.w_off; };
    for (auto& b : net->blocks) { bind_conv(b.c1); bind_conv(b.c2); bind_conv(b.c3); if (b.has_ds) bind_conv(b.cd); }
    return DALI_OK;
}

// fp32 master weights -> bf16 operand images (forward layout = flat cast; dgrad layout = per-tap transpose;
// stem = padded [64][7][8][4]); eval-mode BN coefficients from the running statistics.
extern "C" int dali_resnet_refresh_weights(dali_resnet* net, void* stream) {
    DALI_REQUIRE(net && net->P, "dali_resnet_refresh_weights: net not bound");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_cast_bf16(st, net->P, (size_t)net->param_elems, net->wbf16_flat);
    if (rc) return rc;
    rc = launch_stem_pack_weight(st, net->P + net->stem.w_off, net->stem.cout, net->w_stem);
    if (rc) return rc;
    std::vector<TransposeJob> jobs;                      // every conv's dgrad image [cin][r*s][cout], one launch
    for (auto& b : net->blocks) {
        Conv* cs[4] = {&b.c1, &b.c2, &b.c3, b.has_ds ? &b.cd : nullptr};
        for (Conv* c : cs)
            if (c) jobs.push_back(TransposeJob{c->w_bf16, c->wt_bf16, c->cout, c->r * c->s, c->cin, 0});
    }
    return launch_weight_transpose_batched(st, jobs.data(), (int)jobs.size());
}

namespace {

int bn_train(dali_resnet* net, hipStream_t st, Bn& b, int tiles, double count) {
    return launch_bn_finalize(st, net->stat_partial, tiles, b.C, count, net->P + b.g_off, net->P + b.b_off, net->B + b.rm_off,
                              net->B + b.rv_off, 0.1f, 1e-5f, b.scale, b.shift, b.mean, b.invstd, net->red_scratch);
}

// conv forward: y(raw) = conv(x [optionally bn+relu on load]); BN of the OUTPUT finalised right after
int conv_bn_fwd(dali_resnet* net, hipStream_t st, const Conv& c, Bn& out_bn, const uint16_t* x, const Bn* in_bn, uint16_t* raw, bool training) {
    IGemmArgs a{};
    a.W = c.w_bf16; a.X = x; a.O = raw; a.Res = nullptr;
    a.in_scale = in_bn ? in_bn->scale : nullptr; a.in_shift = in_bn ? in_bn->shift : nullptr; a.in_relu = in_bn ? 1 : 0;
    a.stats = training ? net->stat_partial : nullptr;
    a.Cm = c.cout; a.P = net->N * c.hout * c.wout;
    a.g = conv_geom(c, 0);
    int rc = launch_igemm_conv(st, a);
    if (rc) return rc;
    if (training) return bn_train(net, st, out_bn, igemm_conv_stat_tiles(a.Cm, a.P, a.g.R * a.g.S * a.g.Ck), (double)a.P);
    return DALI_OK;                                          // inference: the coefficients were formed at the top of the forward (one batched launch)
}

// inference: act = relu(bn(conv(x))) leaves the GEMM directly -- the running-statistics coefficients are known before the convolution runs, so
// they ride in the fused output stage (IGemmArgs::out_scale / out_shift / out_relu) and neither the raw output nor a bn_act pass exists
// (getFeatures.py:56-67 forwards the whole train set this way every epoch, train_encodersKIT.py:104-110; every validate, validateModels.py:38-39)
int conv_bn_relu_eval(dali_resnet* net, hipStream_t st, const Conv& c, const Bn& bn, const uint16_t* x, uint16_t* act) {
    IGemmArgs a{};
    a.W = c.w_bf16; a.X = x; a.O = act;
    a.out_scale = bn.scale; a.out_shift = bn.shift; a.out_relu = 1;
    a.Cm = c.cout; a.P = net->N * c.hout * c.wout;
    a.g = conv_geom(c, 0);
    return launch_igemm_conv(st, a);
}

int conv_wgrad(dali_resnet* net, hipStream_t st, const Conv& c, const uint16_t* x, const Bn* in_bn, const uint16_t* dy) {
    WGradArgs a{};
    a.dY = dy; a.X = x; a.partial = net->wgrad_slab;
    a.in_scale = in_bn ? in_bn->scale : nullptr; a.in_shift = in_bn ? in_bn->shift : nullptr; a.in_relu = in_bn ? 1 : 0;
    a.Cm = c.cout; a.P = net->N * c.hout * c.wout; a.Ntot = c.r * c.s * c.cin;
    a.g = conv_geom(c, 0);
    size_t wsb;
    wgrad_plan(a.Cm, a.Ntot, a.P, 512, &a.splits, &a.pix_per_split, &wsb, c.r * c.s, in_bn ? 0 : conv_halo_w(c));
    return launch_igemm_wgrad(st, a, net->G + c.w_off, 0);
}

// out_mask: ReLU mask of the block output this gradient belongs to (the masked gradient dz = dy * (y > 0) is what every block stores)
int conv_dgrad(dali_resnet* net, hipStream_t st, const Conv& c, const uint16_t* dy, const uint16_t* residual, uint16_t* dx,
               const uint8_t* out_mask = nullptr, const uint16_t* w_image = nullptr, const float* bias = nullptr) {
    IGemmArgs a{};
    a.W = w_image ? w_image : c.wt_bf16; a.X = dy; a.O = dx; a.Res = residual; a.out_mask = out_mask; a.bias = bias;
    a.in_scale = nullptr; a.in_shift = nullptr; a.in_relu = 0; a.stats = nullptr;
    a.Cm = c.cin; a.P = net->N * c.hin * c.win;
    a.g = conv_geom(c, 1);
    return launch_igemm_conv(st, a);
}

uint16_t* next_gbuf(dali_resnet* net, const uint16_t* avoid0, const uint16_t* avoid1 = nullptr, const uint16_t* avoid2 = nullptr,
                    const uint16_t* avoid3 = nullptr) {
    for (int i = 0; i < 6; ++i) {
        uint16_t* g = net->gbuf[i];
        if (g != avoid0 && g != avoid1 && g != avoid2 && g != avoid3) return g;
    }
    return nullptr;
}

}  // namespace

extern "C" int dali_resnet_set_feature(dali_resnet* net, int mode) {
    DALI_REQUIRE(net, "dali_resnet_set_feature: null net");
    DALI_REQUIRE(mode >= DALI_FEATURE_BOTH && mode <= DALI_FEATURE_GMP, "dali_resnet_set_feature: bad feature mode %d", mode);
    net->feature_mode = mode;
    return DALI_OK;
}

extern "C" int dali_resnet_forward(dali_resnet* net, void* stream, const float* images, int training, float* emb) {
    DALI_REQUIRE(net && net->P && images && emb, "dali_resnet_forward: null argument or net not bound");
    hipStream_t st = (hipStream_t)stream;
    const bool tr = training != 0;
    net->fwd_training = tr;
    int rc;
    if (!tr) {                                               // inference: every BatchNorm's scale / shift from the running statistics, one launch
        std::vector<BnEvalJob> jobs;
        auto add = [&](Bn& b) { jobs.push_back(BnEvalJob{net->P + b.g_off, net->P + b.b_off, net->B + b.rm_off, net->B + b.rv_off, b.scale, b.shift, b.C}); };
        add(net->stem_bn);
        for (auto& b : net->blocks) { add(b.b1); add(b.b2); add(b.b3); if (b.has_ds) add(b.bd); }
        if ((rc = launch_bn_eval_coeffs_batched(st, jobs.data(), (int)jobs.size(), 1e-5f))) return rc;
    }
    // ---- stem: conv1 -> bn1 -> (no ReLU) -> maxpool ----
    if ((rc = launch_stem_pack_image(st, images, net->N, net->H, net->W, net->ximg))) return rc;
    if (!tr && eval_fused() && stem_fused_supported(net->N, net->H, net->W, net->stem.cout)) {
        // inference: one launch, the convolution's output never stored (stem.hip); bn1 acts on the fp32 accumulators
        if ((rc = launch_stem_conv_bn_pool(st, net->ximg, net->w_stem, net->stem_bn.scale, net->stem_bn.shift, net->N, net->H, net->W, net->pool0))) return rc;
    } else {
    if (tr && stem_train_supported(net->N, net->H, net->W, net->stem.cout)) {
        // training: raw0 + the statistics' partial sums from the patch kernel (stem.hip), 256-pixel tiles as the implicit-GEMM kernel's slab
        if ((rc = launch_stem_conv_stats(st, net->ximg, net->w_stem, net->N, net->H, net->W, net->raw0, net->stat_partial))) return rc;
        if ((rc = bn_train(net, st, net->stem_bn, stem_train_tiles(net->N, net->H), (double)net->N * net->stem_h * net->stem_w))) return rc;
    } else {
        IGemmArgs a{};
        a.W = net->w_stem; a.X = net->ximg; a.O = net->raw0; a.Res = nullptr; a.in_scale = nullptr; a.in_shift = nullptr; a.in_relu = 0;
        a.stats = tr ? net->stat_partial : nullptr;
        a.Cm = net->stem.cout; a.P = net->N * net->stem_h * net->stem_w;
        a.g = stem_geom(net);
        if ((rc = launch_igemm_conv(st, a))) return rc;
        if (tr && (rc = bn_train(net, st, net->stem_bn, igemm_conv_stat_tiles(a.Cm, a.P, 224), (double)a.P))) return rc;
    }
    if ((rc = launch_maxpool_bn_fwd(st, net->raw0, net->stem_bn.scale, net->stem_bn.shift, net->N, net->stem_h, net->stem_w, net->stem.cout,
                                    net->pool0, net->pool_arg))) return rc;
    }
    // ---- bottleneck trunk ----
    const uint16_t* x = net->pool0;
    for (auto& b : net->blocks) {
        b.x = const_cast<uint16_t*>(x);
        const size_t e1 = (size_t)net->N * b.hin * b.win * b.width, e2 = (size_t)net->N * b.hout * b.wout * b.width;
        if (!tr && eval_fused()) {
            if ((rc = conv_bn_relu_eval(net, st, b.c1, b.b1, x, b.a1))) return rc;
            if ((rc = conv_bn_relu_eval(net, st, b.c2, b.b2, b.a1, b.a2))) return rc;
        } else {
            if ((rc = conv_bn_fwd(net, st, b.c1, b.b1, x, nullptr, b.raw1, tr))) return rc;
            if ((rc = launch_bn_act(st, b.raw1, b.b1.scale, b.b1.shift, nullptr, nullptr, nullptr, nullptr, 1, e1, b.width, b.a1, nullptr))) return rc;
            if ((rc = conv_bn_fwd(net, st, b.c2, b.b2, b.a1, nullptr, b.raw2, tr))) return rc;
            if ((rc = launch_bn_act(st, b.raw2, b.b2.scale, b.b2.shift, nullptr, nullptr, nullptr, nullptr, 1, e2, b.width, b.a2, nullptr))) return rc;
        }
        const size_t elems = (size_t)net->N * b.hout * b.wout * b.cout;
        if (b.lin3) {
            // bn3's batch statistics from the moments of a2 (bnlin.hip), then conv3 writes y = relu(bn3(conv3(a2)) + x) and its ReLU mask itself
            const int Pout = net->N * b.hout * b.wout;
            if (tr) {
                const Conv sq = square_conv(b.c3);
                WGradArgs wa{};
                wa.dY = b.a2; wa.X = b.a2; wa.partial = net->wgrad_slab; wa.Cm = b.width; wa.P = Pout; wa.Ntot = b.width;
                wa.g = conv_geom(sq, 0);
                size_t wsb;
                wgrad_plan(wa.Cm, wa.Ntot, wa.P, 512, &wa.splits, &wa.pix_per_split, &wsb, 1, 0);
                const bool fused_cs = wgrad_colsum_supported(wa.Cm, wa.Ntot, 1, wa.P);      // m2 = colsum(a2) rides on the Gram GEMM
                if (fused_cs) wa.colsum = net->cs_partial;
                if ((rc = launch_igemm_wgrad(st, wa, b.gram, 0, fused_cs ? b.m2 : nullptr, fused_cs ? wgrad_colsum_rows(wa.Cm, wa.Ntot, 1, wa.P, wa.splits) : 0))) return rc;
                if (!fused_cs && (rc = launch_colsum(st, b.a2, Pout, b.width, b.m2, net->cs_partial, net->red_scratch))) return rc;
                if ((rc = launch_bnlin_stats(st, b.c3.w_bf16, b.c3.wt_bf16, b.gram, b.m2, b.cout, b.width, (double)Pout, net->P + b.b3.g_off, net->P + b.b3.b_off,
                                             net->B + b.b3.rm_off, net->B + b.b3.rv_off, 0.1f, 1e-5f, b.ut, b.dot, b.b3.scale, b.b3.shift, b.b3.mean,
                                             b.b3.invstd))) return rc;
            }
            if (!tr && b.cat_eval && eval_fused() && DALI_ENV_INT("DALI_EVAL_CAT", 1) != 0) {
                // inference, stride-1 downsample branch: y = relu([a2 | x] [s3.W3 | sd.Wd]^T + shift3 + shift_d) in one launch: no raw branch
                // output, no residual read (the scales ride in a split hi + lo weight image: fp32-grade weights, mirrored by the twin)
                if ((rc = launch_fold_cat_weights(st, net->P + b.c3.w_off, net->P + b.cd.w_off, b.b3.scale, b.bd.scale, b.b3.shift, b.bd.shift, b.cout, b.width,
                                                  b.cin, 2, b.wcat, b.shcat))) return rc;
                IGemmArgs ca{};
                ca.W = b.wcat; ca.X = b.a2; ca.X2 = x; ca.Ck1 = b.width; ca.x_rep = 2; ca.O = b.y; ca.out_shift = b.shcat; ca.out_relu = 1;
                ca.Cm = b.cout; ca.P = Pout;
                Conv cat = b.c3;
                cat.cin = 2 * (b.width + b.cin);
                ca.g = conv_geom(cat, 0);
                if ((rc = launch_igemm_conv(st, ca))) return rc;
                x = b.y;
                continue;
            }
            IGemmArgs a{};
            a.W = b.c3.w_bf16; a.X = b.a2; a.O = b.y; a.Res = x; a.out_scale = b.b3.scale; a.out_shift = b.b3.shift; a.out_relu = 1;
            if (b.has_ds) {                               // identity = bnd(convd(x)): raw output + its BatchNorm as residual scale / extra shift
                if ((rc = conv_bn_fwd(net, st, b.cd, b.bd, x, nullptr, b.rawd, tr))) return rc;
                a.Res = b.rawd; a.res_scale = b.bd.scale; a.bias = b.bd.shift;
                if (b.lin_ds && tr) {                         // moments of x for the branch's backward: G_x = x^T x, m2 = colsum(x), Ut = (Wd G_x)^T
                    Conv sq = b.cd;
                    sq.cout = b.cd.cin;
                    WGradArgs wa{};
                    wa.dY = x; wa.X = x; wa.partial = net->wgrad_slab; wa.Cm = b.cin; wa.P = Pout; wa.Ntot = b.cin;
                    wa.g = conv_geom(sq, 0);
                    size_t wsb;
                    wgrad_plan(wa.Cm, wa.Ntot, wa.P, 512, &wa.splits, &wa.pix_per_split, &wsb, 1, 0);
                    const bool fused_cs = wgrad_colsum_supported(wa.Cm, wa.Ntot, 1, wa.P);
                    if (fused_cs) wa.colsum = net->cs_partial;
                    if ((rc = launch_igemm_wgrad(st, wa, b.gram_d, 0, fused_cs ? b.m2_d : nullptr, fused_cs ? wgrad_colsum_rows(wa.Cm, wa.Ntot, 1, wa.P, wa.splits) : 0))) return rc;
                    if (!fused_cs && (rc = launch_colsum(st, x, Pout, b.cin, b.m2_d, net->cs_partial, net->red_scratch))) return rc;
                    if ((rc = launch_bnlin_ut(st, b.cd.w_bf16, b.gram_d, b.cout, b.cin, b.ut_d))) return rc;
                }
            }
            a.bits_out = tr ? b.ybits : nullptr;
            a.Cm = b.cout; a.P = Pout; a.g = conv_geom(b.c3, 0);
            if ((rc = launch_igemm_conv(st, a))) return rc;
            x = b.y;
            continue;
        }
        if ((rc = conv_bn_fwd(net, st, b.c3, b.b3, b.a2, nullptr, b.raw3, tr))) return rc;
        if (b.has_ds) {
            if ((rc = conv_bn_fwd(net, st, b.cd, b.bd, x, nullptr, b.rawd, tr))) return rc;
            rc = launch_bn_act(st, b.raw3, b.b3.scale, b.b3.shift, nullptr, b.rawd, b.bd.scale, b.bd.shift, 1, elems, b.cout, b.y, tr ? b.ybits : nullptr);
        } else {
            rc = launch_bn_act(st, b.raw3, b.b3.scale, b.b3.shift, x, nullptr, nullptr, nullptr, 1, elems, b.cout, b.y, tr ? b.ybits : nullptr);
        }
        if (rc) return rc;
        x = b.y;
    }
    // ---- head: avg-pool + max-pool, BatchNorm1d ----
    if ((rc = launch_head_pool_fwd(st, x, net->N, net->head_hw, net->feat_dim, net->feature_mode, net->feat, net->head_arg))) return rc;
    return launch_bn1d_fwd(st, net->feat, net->N, net->feat_dim, net->P + net->neck.g_off, net->P + net->neck.b_off, net->B + net->neck.rm_off,
                           net->B + net->neck.rv_off, tr ? 1 : 0, 0.1f, 1e-5f, emb, net->neck_mean, net->neck_invstd);
}

// Every block receives and hands on the MASKED gradient dz = dL/dy * (y > 0) (the gradient in front of the block's final ReLU): the
// producer (the data-gradient epilogue of the next block's conv1, or the head pooling) applies the mask bits, so that neither the
// BatchNorm backward nor the identity path has to.  prev_bits = ReLU mask of the previous block's output (null for the first block).
static int block_backward(dali_resnet* net, hipStream_t st, Block& b, const uint8_t* prev_bits) {
    int rc;
    uint16_t* dz = net->cur_dy;                                   // masked grad wrt block output y
    const int Pout = net->N * b.hout * b.wout, Pin = net->N * b.hin * b.win;
    uint16_t *d_a2, *d_rawd = nullptr, *scratch_a;
    if (b.lin3) {
        // bn3 + conv3 through the moments of a2 (bnlin.hip): no reduce / apply passes over [P][cout] tensors, raw3 does not exist
        WGradArgs wa{};
        wa.dY = dz; wa.X = b.a2; wa.partial = net->wgrad_slab; wa.Cm = b.cout; wa.P = Pout; wa.Ntot = b.width;
        wa.g = conv_geom(b.c3, 0);
        size_t wsb;
        wgrad_plan(wa.Cm, wa.Ntot, wa.P, 512, &wa.splits, &wa.pix_per_split, &wsb, 1, 0);
        const bool fused_cs = wgrad_colsum_supported(wa.Cm, wa.Ntot, 1, wa.P);              // s = colsum(dz) rides on the weight-gradient GEMM
        if (fused_cs) wa.colsum = net->cs_partial;
        if ((rc = launch_igemm_wgrad(st, wa, net->G + b.c3.w_off, 0, fused_cs ? b.sdz : nullptr,
                                     fused_cs ? wgrad_colsum_rows(wa.Cm, wa.Ntot, 1, wa.P, wa.splits) : 0))) return rc;           // G0 = dz^T a2 into the gradient slot; finished in place below
        if (!fused_cs && (rc = launch_colsum(st, dz, Pout, b.cout, b.sdz, net->cs_partial, net->red_scratch))) return rc;
        const int ldw = b.cout + b.width;
        if ((rc = launch_bnlin_bwd(st, b.c3.w_bf16, b.c3.wt_bf16, b.ut, b.m2, b.sdz, b.cout, b.width, (double)Pout, b.b3.scale, b.b3.mean,
                                   b.b3.invstd, net->G + b.c3.w_off, net->G + b.b3.g_off, net->G + b.b3.b_off, b.wd1, b.wd1 + b.cout, b.bvec, b.qk, ldw, ldw))) return rc;
        d_a2 = next_gbuf(net, dz);
        scratch_a = next_gbuf(net, dz, d_a2);
        {   // d_a2 = dz (A.W3) + W3^T Kc - a2 (W3^T diag(Q) W3) as ONE GEMM over K = cout + w: [dz | a2] against the concatenated image
            // (two launches with the partial result stored and re-read before: 190 -> see docs/experiments.md)
            Conv cat = b.c3;
            cat.cout = ldw;
            IGemmArgs ga{};
            ga.W = b.wd1; ga.X = dz; ga.X2 = b.a2; ga.Ck1 = b.cout; ga.O = d_a2; ga.bias = b.bvec;
            ga.Cm = b.width; ga.P = Pout;
            ga.g = conv_geom(cat, 1);
            if ((rc = launch_igemm_conv(st, ga))) return rc;
        }
        if (b.lin_ds) {
            // downsample branch through the moments of x: G0d = dz^T x into the gradient slot (s = colsum(dz) is bn3's), finished in place by the row
            // kernel together with dgamma / dbeta and the two data-gradient images; the data gradient itself follows conv1's below
            WGradArgs wd{};
            wd.dY = dz; wd.X = b.x; wd.partial = net->wgrad_slab; wd.Cm = b.cout; wd.P = Pout; wd.Ntot = b.cin;
            wd.g = conv_geom(b.cd, 0);
            size_t wsb2;
            wgrad_plan(wd.Cm, wd.Ntot, wd.P, 512, &wd.splits, &wd.pix_per_split, &wsb2, 1, 0);
            if ((rc = launch_igemm_wgrad(st, wd, net->G + b.cd.w_off, 0))) return rc;
            const int ldd = b.cout + b.cin;
            if ((rc = launch_bnlin_bwd(st, b.cd.w_bf16, b.cd.wt_bf16, b.ut_d, b.m2_d, b.sdz, b.cout, b.cin, (double)Pout, b.bd.scale, b.bd.mean, b.bd.invstd,
                                       net->G + b.cd.w_off, net->G + b.bd.g_off, net->G + b.bd.b_off, b.wdd, b.wdd + b.cout, b.bvec_d, b.qk_d, ldd, ldd))) return rc;
        } else if (b.has_ds) {                                    // the downsample BatchNorm: its own two passes over (dz, rawd)
            d_rawd = next_gbuf(net, dz, d_a2, scratch_a);
            BnBwdSide sd{b.rawd, b.bd.mean, b.bd.invstd, b.bd.scale, b.bd.shift};
            if ((rc = launch_bn_bwd(st, dz, nullptr, nullptr, sd, nullptr, 0, Pout, b.cout, net->bwd_partial, b.bd.coef, nullptr, net->G + b.bd.g_off,
                                    net->G + b.bd.b_off, nullptr, nullptr, d_rawd, nullptr, nullptr, net->red_scratch))) return rc;
        }
    } else {
        uint16_t* d_raw3 = next_gbuf(net, dz);
        d_rawd = b.has_ds ? next_gbuf(net, dz, d_raw3) : nullptr;
        BnBwdSide s3{b.raw3, b.b3.mean, b.b3.invstd, b.b3.scale, b.b3.shift};
        BnBwdSide sd{b.rawd, b.bd.mean, b.bd.invstd, b.bd.scale, b.bd.shift};
        rc = launch_bn_bwd(st, dz, nullptr, nullptr, s3, b.has_ds ? &sd : nullptr, 0, Pout, b.cout, net->bwd_partial, b.b3.coef, b.has_ds ? b.bd.coef : nullptr,
                           net->G + b.b3.g_off, net->G + b.b3.b_off, b.has_ds ? net->G + b.bd.g_off : nullptr, b.has_ds ? net->G + b.bd.b_off : nullptr,
                           d_raw3, d_rawd, nullptr, net->red_scratch);
        if (rc) return rc;
        if ((rc = conv_wgrad(net, st, b.c3, b.a2, nullptr, d_raw3))) return rc;
        d_a2 = next_gbuf(net, dz, d_raw3, d_rawd);
        if ((rc = conv_dgrad(net, st, b.c3, d_raw3, nullptr, d_a2))) return rc;
        scratch_a = d_raw3;                                       // d_raw3 is dead from here on
    }
    // bn2 + relu, in place.  The ReLU mask is recomputed from raw2 (raw*scale+shift > 0 <=> a2 > 0: bf16 rounding cannot
    // flush a positive fp32 to zero) instead of reading a2: one tensor read less in each of the two passes.
    BnBwdSide s2{b.raw2, b.b2.mean, b.b2.invstd, b.b2.scale, b.b2.shift};
    if ((rc = launch_bn_bwd(st, d_a2, nullptr, nullptr, s2, nullptr, 1, Pout, b.width, net->bwd_partial, b.b2.coef, nullptr, net->G + b.b2.g_off,
                            net->G + b.b2.b_off, nullptr, nullptr, d_a2, nullptr, nullptr, net->red_scratch))) return rc;
    // conv2
    if ((rc = conv_wgrad(net, st, b.c2, b.a1, nullptr, d_a2))) return rc;
    uint16_t* d_a1 = scratch_a;
    if ((rc = conv_dgrad(net, st, b.c2, d_a2, nullptr, d_a1))) return rc;
    BnBwdSide s1{b.raw1, b.b1.mean, b.b1.invstd, b.b1.scale, b.b1.shift};
    if ((rc = launch_bn_bwd(st, d_a1, nullptr, nullptr, s1, nullptr, 1, Pin, b.width, net->bwd_partial, b.b1.coef, nullptr, net->G + b.b1.g_off,
                            net->G + b.b1.b_off, nullptr, nullptr, d_a1, nullptr, nullptr, net->red_scratch))) return rc;
    b.dbg_d_raw1 = d_a1;
    // conv1 (+ identity / downsample branch); the result is masked with the previous block's ReLU bits
    if ((rc = conv_wgrad(net, st, b.c1, b.x, nullptr, d_a1))) return rc;
    uint16_t* dx = d_a2;                                          // d_a2 is dead now
    if (b.lin_ds) {
        // dx = d_a1 W1 + [dz | x] [(A_d.Wd)^T | -(Wd^T diag(Q_d) Wd)] + Wd^T Kc: conv1's data gradient, then the branch's two-operand GEMM on top of it
        if ((rc = conv_dgrad(net, st, b.c1, d_a1, nullptr, dx, prev_bits))) return rc;
        Conv cat = b.cd;
        cat.cout = b.cout + b.cin;
        IGemmArgs ga{};
        ga.W = b.wdd; ga.X = dz; ga.X2 = b.x; ga.Ck1 = b.cout; ga.O = dx; ga.Res = dx; ga.bias = b.bvec_d;
        ga.Cm = b.cin; ga.P = Pin;
        ga.g = conv_geom(cat, 1);
        if ((rc = launch_igemm_conv(st, ga))) return rc;
    } else if (b.has_ds) {
        if ((rc = conv_wgrad(net, st, b.cd, b.x, nullptr, d_rawd))) return rc;
        // conv1's data gradient first, then the downsample branch accumulates into it in place: with stride 2 only the
        // even-even quarter of the positions receives a contribution (launch_igemm_conv's parity split).  (a + b) * m = a * m + b * m:
        // both launches apply the mask.
        if ((rc = conv_dgrad(net, st, b.c1, d_a1, nullptr, dx, prev_bits))) return rc;
        if ((rc = conv_dgrad(net, st, b.cd, d_rawd, dx, dx, prev_bits))) return rc;
    } else {
        if ((rc = conv_dgrad(net, st, b.c1, d_a1, dz, dx, prev_bits))) return rc;             // identity path: + dz
    }
    net->cur_dy = dx;
    net->cur_dy_bytes = (int64_t)Pin * b.cin * 2;
    return DALI_OK;
}

// maxpool + stem BN backward, then the stem weight gradient (no data gradient: images need none)
static int stem_backward(dali_resnet* net, hipStream_t st) {
    int rc;
    uint16_t* d_raw0 = next_gbuf(net, net->cur_dy);
    if ((rc = launch_maxpool_bn_bwd(st, net->cur_dy, net->pool_arg, net->raw0, net->stem_bn.mean, net->stem_bn.invstd, net->stem_bn.scale,
                                    net->N, net->stem_h, net->stem_w, net->stem.cout, net->bwd_partial, net->stem_bn.coef,
                                    net->G + net->stem_bn.g_off, net->G + net->stem_bn.b_off, d_raw0, net->red_scratch))) return rc;
    WGradArgs a{};
    a.dY = d_raw0; a.X = net->ximg; a.partial = net->wgrad_slab; a.in_scale = nullptr; a.in_shift = nullptr; a.in_relu = 0;
    a.Cm = net->stem.cout; a.P = net->N * net->stem_h * net->stem_w; a.Ntot = 224;
    a.g = stem_geom(net);
    size_t wsb;
    wgrad_plan(a.Cm, a.Ntot, a.P, 512, &a.splits, &a.pix_per_split, &wsb, 7);
    if ((rc = launch_igemm_wgrad(st, a, net->stem_dw_pad, 0))) return rc;
    return launch_stem_unpack_wgrad(st, net->stem_dw_pad, net->stem.cout, net->G + net->stem.w_off);
}

// Runs stages [stage_begin, stage_end] of the backward pass (0: neck+head+layer4, 1: layer3, 2: layer2,
// 3: layer1+stem).  Stage 0 consumes d_emb; later stages continue from the gradient the previous call left.
extern "C" int dali_resnet_backward(dali_resnet* net, void* stream, const float* d_emb, int stage_begin, int stage_end) {
    DALI_REQUIRE(net && net->P && net->G, "dali_resnet_backward: net not bound (grads required)");
    DALI_REQUIRE(net->fwd_training, "dali_resnet_backward: the last forward was not in training mode");
    DALI_REQUIRE(stage_begin >= 0 && stage_end <= 3 && stage_begin <= stage_end, "dali_resnet_backward: bad stage range %d..%d", stage_begin, stage_end);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    for (int stage = stage_begin; stage <= stage_end; ++stage) {
        const int li = 3 - stage;
        if (stage == 0) {
            DALI_REQUIRE(d_emb != nullptr, "dali_resnet_backward: d_emb is null");
            if ((rc = launch_bn1d_bwd(st, net->feat, d_emb, net->N, net->feat_dim, net->P + net->neck.g_off, net->neck_mean, net->neck_invstd,
                                      net->dfeat, net->G + net->neck.g_off, net->G + net->neck.b_off))) return rc;
            net->cur_dy = net->gbuf[0];
            if ((rc = launch_head_pool_bwd(st, net->dfeat, net->head_arg, net->N, net->head_hw, net->feat_dim, net->feature_mode, net->cur_dy,
                                           net->blocks.back().ybits))) return rc;
            net->cur_dy_bytes = (int64_t)net->N * net->head_hw * net->feat_dim * 2;
        }
        for (int bi = net->stage_last[li]; bi >= net->stage_first[li]; --bi)
            if ((rc = block_backward(net, st, net->blocks[bi], bi > 0 ? net->blocks[bi - 1].ybits : nullptr))) return rc;
        if (stage == 3 && (rc = stem_backward(net, st))) return rc;
    }
    return DALI_OK;
}

// Diagnostic (include/daliid_debug.h): the backward pass ONE bottleneck at a time, so that a test can read the gradient entering every block
// (`grad_cur` of dali_resnet_debug_tensor) instead of one per stage.  Call with block = last block first (that call also runs the neck + head
// backward from d_emb), then block - 1, ... down to 0, then -1 (the stem); same launches, same order as dali_resnet_backward.
extern "C" int dali_debug_resnet_backward_block(dali_resnet* net, void* stream, const float* d_emb, int block) {
    DALI_REQUIRE(net && net->P && net->G, "dali_debug_resnet_backward_block: net not bound (grads required)");
    DALI_REQUIRE(net->fwd_training, "dali_debug_resnet_backward_block: the last forward was not in training mode");
    const int nb = (int)net->blocks.size();
    if (block == -1) return stem_backward(net, (hipStream_t)stream);      // after block 0: the stem (its own call, so that block 0's scratch tensors can be read first)
    DALI_REQUIRE(block >= 0 && block < nb, "dali_debug_resnet_backward_block: block %d out of range", block);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (block == nb - 1) {
        DALI_REQUIRE(d_emb != nullptr, "dali_debug_resnet_backward_block: d_emb is null");
        if ((rc = launch_bn1d_bwd(st, net->feat, d_emb, net->N, net->feat_dim, net->P + net->neck.g_off, net->neck_mean, net->neck_invstd, net->dfeat,
                                  net->G + net->neck.g_off, net->G + net->neck.b_off))) return rc;
        net->cur_dy = net->gbuf[0];
        if ((rc = launch_head_pool_bwd(st, net->dfeat, net->head_arg, net->N, net->head_hw, net->feat_dim, net->feature_mode, net->cur_dy,
                                       net->blocks.back().ybits))) return rc;
        net->cur_dy_bytes = (int64_t)net->N * net->head_hw * net->feat_dim * 2;
    }
    if ((rc = block_backward(net, st, net->blocks[block], block > 0 ? net->blocks[block - 1].ybits : nullptr))) return rc;
    return DALI_OK;
}

// Debug/inspection: device pointer + byte size of a named intermediate (valid after a forward).
// Names: raw0, pool0, feat, grad_cur (gradient wrt the input of the last block processed by backward), block<i>.{raw1,raw2,raw3,rawd,y}, block<i>.bn{1,2,3,d}.{scale,shift,mean,invstd}, bn1.{...}
extern "C" int dali_resnet_debug_tensor(dali_resnet* net, const char* name, void** ptr, int64_t* bytes) {
    DALI_REQUIRE(net && name && ptr && bytes && net->arena, "dali_resnet_debug_tensor: bad argument / net not bound");
    const std::string n(name);
    const size_t N = net->N;
    auto bn_field = [&](Bn& b, const std::string& f) -> bool {
        float* p = f == "scale" ? b.scale : f == "shift" ? b.shift : f == "mean" ? b.mean : f == "invstd" ? b.invstd : nullptr;
        if (!p) return false;
        *ptr = p; *bytes = (int64_t)b.C * 4;
        return true;
    };
    if (n == "raw0") { *ptr = net->raw0; *bytes = (int64_t)(N * net->stem_h * net->stem_w * net->stem.cout * 2); return DALI_OK; }
    if (n == "pool0") { *ptr = net->pool0; *bytes = (int64_t)(N * net->pool_h * net->pool_w * net->stem.cout * 2); return DALI_OK; }
    if (n == "grad_cur" && net->cur_dy) { *ptr = net->cur_dy; *bytes = net->cur_dy_bytes; return DALI_OK; }
    if (n == "feat") { *ptr = net->feat; *bytes = (int64_t)(N * net->feat_dim * 4); return DALI_OK; }
    if (n.rfind("bn1.", 0) == 0 && bn_field(net->stem_bn, n.substr(4))) return DALI_OK;
    if (n.rfind("block", 0) == 0) {
        const size_t dot = n.find('.');
        if (dot != std::string::npos) {
            const int bi = atoi(n.substr(5, dot - 5).c_str());
            if (bi >= 0 && bi < (int)net->blocks.size()) {
                Block& b = net->blocks[bi];
                const std::string f = n.substr(dot + 1);
                const size_t pin = N * b.hin * b.win, pout = N * b.hout * b.wout;
                if (f == "raw1") { *ptr = b.raw1; *bytes = (int64_t)(pin * b.width * 2); return DALI_OK; }
                if (f == "raw2") { *ptr = b.raw2; *bytes = (int64_t)(pout * b.width * 2); return DALI_OK; }
                if (f == "raw3" && !b.lin3) { *ptr = b.raw3; *bytes = (int64_t)(pout * b.cout * 2); return DALI_OK; }
                if (f == "rawd" && b.has_ds) { *ptr = b.rawd; *bytes = (int64_t)(pout * b.cout * 2); return DALI_OK; }
                if (f == "y") { *ptr = b.y; *bytes = (int64_t)(pout * b.cout * 2); return DALI_OK; }
                if (f == "d_raw1" && b.dbg_d_raw1) { *ptr = b.dbg_d_raw1; *bytes = (int64_t)(pin * b.width * 2); return DALI_OK; }
                if (f.rfind("bn", 0) == 0 && f.size() > 4) {
                    Bn* bn = f[2] == '1' ? &b.b1 : f[2] == '2' ? &b.b2 : f[2] == '3' ? &b.b3 : (f[2] == 'd' && b.has_ds) ? &b.bd : nullptr;
                    if (bn && bn_field(*bn, f.substr(4))) return DALI_OK;
                }
            }
        }
    }
    set_error("dali_resnet_debug_tensor: unknown tensor '%s'", name);
    return DALI_ERR_INVALID;
}
