/* daliid.h -- C ABI of libdaliid_hip.so: the MI355X (gfx950) kernels under the DaliID Person-ReID
 * hot path (embedding train step + gallery distance / CMC-mAP evaluation).
 *
 * The reference (Gabrielcb/DaliID) is pure Python and has NO FFI of its own; its "plugin interface"
 * for this path is the Python module surface mainKIT.py consumes.  Each entry point
 * below cites the reference call it stands under (paths relative to Person-ReID/).  The Python mirror
 * in daliid_amd/ binds these with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (DALI_OK) or a negative dali_status; dali_last_error() gives the
 *     message of the calling thread's last failure.  Nothing throws across the boundary.
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller (PyTorch) owns every
 *     buffer.  The library only owns the opaque dali_ctx (a grow-only device workspace) and net plans.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued there, nothing synchronises.
 *   - bf16 tensors are passed as uint16_t* (raw bits).  Activations are NHWC.
 */
#ifndef DALIID_H
#define DALIID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DALI_OK = 0,
    DALI_ERR_INVALID = -1,   /* null pointer, bad shape / alignment / enum */
    DALI_ERR_HIP = -2,       /* a HIP runtime call failed */
    DALI_ERR_NOMEM = -3,     /* workspace allocation failed */
    DALI_ERR_LIMIT = -4,     /* a documented capacity limit was exceeded */
    DALI_ERR_UNSUPPORTED = -5
} dali_status;

typedef struct dali_ctx dali_ctx;

/* ---- context ---------------------------------------------------------------------------------- */
int dali_version(void);
const char* dali_last_error(void);
/* One context per process / GPU and per stream that may run beside another: `device` is the HIP device ordinal.  A context owns one
 * grow-only workspace block that its calls use from offset 0 (Adam's partial sums, the resize's intermediate image, the distance
 * pre-pass, the class targets), so two calls on the SAME context must not overlap in time: enqueue them on one stream, or give the
 * second stream its own context (the Python side does: `_lib.ctx(device, lane="side")` for the input pipeline's side stream). */
int dali_ctx_create(int device, dali_ctx** out);
int dali_ctx_destroy(dali_ctx* ctx);
/* Pre-size the workspace (hipMalloc happens here, not inside later calls / graph captures). */
int dali_ctx_reserve(dali_ctx* ctx, size_t bytes);

/* ---- evaluation path: validateModels.validate (validateModels.py:35-58) ------------------------- */
typedef enum { DALI_METRIC_COSINE = 0,   /* 1 - q.g         validateModels.py:47, evaluate.py:291 */
               DALI_METRIC_L2SQ = 1,     /* |q|^2+|g|^2-2q.g  (square of the commented cdist, :45) */
               DALI_METRIC_DOT = 2       /* q.g : the similarity GEMMs of the loss heads (losses.py:62, :277) */
} dali_metric;
typedef enum { DALI_PREC_BF16X3 = 0,     /* split-bf16 (hi*hi+hi*lo+lo*hi), fp32 accumulate: ~1e-6 abs */
               DALI_PREC_BF16 = 1        /* single bf16 product, fp32 accumulate: ~1e-4 abs on unit rows */
} dali_precision;
/* pooling of the embedding head: the `feature` attribute eval scripts set through model.module
 * (evaluateCleanATModels.py:249-256, :335-340); training always uses BOTH (Encoders.py:341-345). */
typedef enum { DALI_FEATURE_BOTH = 0, DALI_FEATURE_GAP = 1, DALI_FEATURE_GMP = 2 } dali_feature;

/* y = x / (|x|_2 + eps), rows of an fp32 [n,d] matrix.
 * validateModels.py:41-42 (eps = 0) and train_encodersKIT.py:198 (eps = 1e-9).
 * y may alias x.  norms (nullable) receives |x|_2 per row. */
int dali_l2norm_rows(dali_ctx* ctx, void* stream, const float* x, int n, int d, float eps,
                     float* y, float* norms);
/* Backward of the above: dx = dy/(n+eps) - x*(x.dy)/(n*(n+eps)^2); needs the forward INPUT x. */
int dali_l2norm_rows_bwd(dali_ctx* ctx, void* stream, const float* x, const float* dy, int n, int d,
                         float eps, float* dx);

/* out[nq,ng] (fp32, row-major) = metric(Q[nq,d], G[ng,d]); fp32 inputs.
 * normalize != 0 fuses the row normalisation of validateModels.py:41-42 into the operand pre-pass
 * (the reference sequence normalise -> 1 - q@g.T becomes one call).
 * Workspace: the two operand images (dali_pairdist_operand_bytes) + (nq+ng)*4.
 * DALI_METRIC_DOT at DALI_PREC_BF16X3 without normalisation on a problem that would fill at most a quarter of the CUs with 128 x 256 tiles
 * (the loss heads' similarity GEMMs, losses.py:62, :277) runs on the fp32 matrix cores instead: exact fp32 products, no operand images, no
 * workspace; Q and G need 4-byte alignment only (any d). */
int dali_pairdist(dali_ctx* ctx, void* stream, const float* Q, const float* G, int nq, int ng, int d,
                  int metric, int precision, int normalize, float* out);

/* The two halves of dali_pairdist, for callers that reuse a prepared gallery against many query sets:
 * prepare: X[n,d] fp32 -> the bf16 operand image the distance kernel reads (dali_pairdist_operand_bytes(n, d, precision) bytes,
 * 16-byte aligned; opaque: DALI_PREC_BF16X3 interleaves the bf16 hi part and the bf16 residual per 32 columns so that one
 * k-step of a row is one 128-byte cache line, DALI_PREC_BF16 pads the row to 64 columns) plus |row|^2 (after the optional
 * normalisation).  Both sides of dali_pairdist_prepared must have been prepared with the same precision. */
size_t dali_pairdist_operand_bytes(int n, int d, int precision);
int dali_pairdist_prepare(dali_ctx* ctx, void* stream, const float* X, int n, int d, int normalize, int precision,
                          void* image, float* sq);
int dali_pairdist_prepared(dali_ctx* ctx, void* stream, const void* q_image, const float* q_sq, const void* g_image,
                           const float* g_sq, int nq, int ng, int d, int metric, int precision, float* out);

/* Two-model distance fusion (evaluateCleanATModels.py:103-160): with inout holding the first model's distmat d1
 * (dali_pairdist), computes the second model's d2 = 1 - q.g on the fly and stores
 *   inout = (w1*d1 + w2*d2) / (w1 + w2),  w_m[q,g] = max(q_mag_m[q], g_mag_m[g])            (:154-157)
 * where the magnitudes are the embedding norms under the chosen pooling (getWeightsByMagnitude :249-256);
 * q_mag_prev/g_mag_prev belong to the first model, q_mag/g_mag to this one.  All four null: w = 1, i.e. the
 * "simple ensemble" (d1 + d2)/2 (:126).  One extra read of the matrix instead of a second matrix + a blend pass. */
int dali_pairdist_blend(dali_ctx* ctx, void* stream, const float* Q, const float* G, int nq, int ng, int d,
                        int precision, int normalize, const float* q_mag_prev, const float* g_mag_prev,
                        const float* q_mag, const float* g_mag, float* inout);

/* market1501-protocol CMC / mAP: the arithmetic of torchreid.metrics.evaluate_rank(distmat, q_pids,
 * g_pids, q_camids, g_camids, use_metric_cuhk03=False) called at validateModels.py:68.
 * ids are int32 codes (the Python mirror factorises the reference's string columns).
 * Exact distance ties are ordered by ascending gallery index.
 * Outputs (device): cmc[max_rank] fp32, mAP[1] fp32 (+ fp64 copy in map64[1], nullable),
 * num_valid[1] int32 (queries with at least one match after junk removal),
 * per-query ap[nq] fp32 and first_rank[nq] int32 (-1 = invalid query) -- both nullable.
 * The gallery is indexed by identity on the device first (counting sort), so the ids are read once per evaluation, not once per
 * query.  Limits, reported through status[0] (device): 1 = a query's identity has more than 4096 gallery entries; 2 = the gallery
 * identity codes span more than 2^20 values (pass dense codes, as the mirror's factorize_ids does). */
int dali_rank_eval(dali_ctx* ctx, void* stream, const float* distmat, const int32_t* q_pids,
                   const int32_t* g_pids, const int32_t* q_camids, const int32_t* g_camids, int nq, int ng,
                   int max_rank, float* cmc, float* mAP, double* map64, int32_t* num_valid,
                   float* ap, int32_t* first_rank, int32_t* status);

/* Gallery-sharded evaluation over N ranks (validateModels.py:41-47,61-69 with the gallery split across GPUs): each rank holds
 * dist_shard [nq][ng] = the distances of ALL queries to ITS gallery slice, whose first entry has global index g_offset.  Three calls with
 * two collectives of the caller's between them (the mirror: ops_eval.rank_eval_sharded over torch.distributed):
 *   dali_rank_shard_matches -> keys [nq][cap] int64 ((orderable distance bits << 32) | global gallery index; unused slots ~0) and counts [nq]
 *                              of the query's matches inside this shard;            [all-gather keys and counts: rank-major]
 *   dali_rank_shard_bins    -> bins [nq][bins_cap + 1] int32: this shard's kept entries binned by the number of matches (of all shards) with a
 *                              smaller key;                                         [all-reduce SUM of bins]
 *   dali_rank_shard_finish  -> cmc / mAP / per-query ap + first_rank from the summed bins; bit-identical to dali_rank_eval on the whole matrix.
 * cap: an upper bound of a query's matches inside one shard, the same on every rank; bins_cap >= a query's matches over all shards
 * (<= 4096).  status[0] = 1 when either bound is exceeded, 2 as for dali_rank_eval. */
int dali_rank_shard_matches(dali_ctx* ctx, void* stream, const float* dist_shard, const int32_t* q_pids, const int32_t* g_pids,
                            const int32_t* q_camids, const int32_t* g_camids, int nq, int ng, int g_offset, int cap,
                            int64_t* keys, int32_t* counts, int32_t* status);
int dali_rank_shard_bins(dali_ctx* ctx, void* stream, const float* dist_shard, const int32_t* q_pids, const int32_t* g_pids,
                         const int32_t* q_camids, const int32_t* g_camids, int nq, int ng, int g_offset, const int64_t* keys_all,
                         const int32_t* counts_all, int world, int cap, int32_t* bins, int bins_cap, int32_t* status);
int dali_rank_shard_finish(dali_ctx* ctx, void* stream, const int32_t* bins, const int32_t* counts_all, int world, int nq, int bins_cap,
                           int max_rank, float* cmc, float* mAP, double* map64, int32_t* num_valid, float* ap, int32_t* first_rank);

/* ---- training path: Encoders.ResNet50ReID trunk (Encoders.py:330-339) ----------------------------- *
 * Single-op entry points (the parity tests call these; the net plan below chains the same kernels).
 * Layouts: activations NHWC bf16; forward weights [cout][r][s][cin] bf16; dgrad weights
 * [cin][r][s][cout] bf16; weight gradients [cout][r][s][cin] fp32.  stride in {1,2}.
 * These stand under torch.nn.Conv2d forward/backward of torchvision's resnet50 as wrapped by
 * Encoders.py:312-322. */

/* y[n,ho,wo,cout] = conv(x).  If in_scale/in_shift (fp32 [cin]) are given, x is read as
 * relu?(x*in_scale+in_shift) (the preceding BatchNorm(+ReLU) fused into the operand load; padding stays 0).
 * stats (nullable): fp32 [dali_conv2d_stat_tiles(...)][cout][2] per-tile partial (sum, sum of squares) of the
 * fp32 results -- the batch statistics training-mode BatchNorm needs.  cin % 32 == 0, cout % 4 == 0. */
int dali_conv2d_fwd(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, uint16_t* y,
                    int n, int h, int wd, int cin, int cout, int r, int s, int stride, int pad,
                    const float* in_scale, const float* in_shift, int in_relu, float* stats);
/* y = relu?( conv(x) * out_scale[c] + out_shift[c] ): the convolution with the BatchNorm (+ ReLU) that follows it folded into the GEMM's output
 * stage -- the fp32 accumulators are scaled, shifted, clamped and rounded to bf16 once; no raw output, no separate BatchNorm pass.  The inference
 * forward of torchvision's Bottleneck conv1 / conv2 (under Encoders.py:336-339) with scale = gamma / sqrt(running_var + eps),
 * shift = beta - running_mean * scale (getFeatures.py:56-67 forwards in eval mode).  cin % 32 == 0, cout % 8 == 0, stride 1 or 2. */
int dali_conv2d_bn_act(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, uint16_t* y, int n, int h, int wd, int cin, int cout,
                       int r, int s, int stride, int pad, const float* out_scale, const float* out_shift, int out_relu);
/* The inference stem in one launch: y = maxpool3x3/2( conv7x7/2(images) * scale[c] + shift[c] ), images fp32 NCHW [n,3,h,w], weight fp32
 * [64][7][7][3] (the net's storage order of the logical OIHW tensor), scale / shift fp32 [64] (bn1 by running statistics; NO ReLU between them and
 * the pool: Encoders.py:321-322, :334), y bf16 NHWC [n, h/4, w/4, 64].  torchvision's conv1 / bn1 / maxpool under Encoders.py:33,36 as
 * getFeatures.py:56-67 forwards them (eval mode): the convolution's output is never stored; rounding points as if it were (bf16 of the
 * convolution's output, fp32 affine, bf16 of the pooled value), so the result equals conv -> bf16 -> affine -> max-pool -> bf16 bit for bit.
 * h % 32 == 0, w in {32, 64, 128} (dali_stem_fused_supported; DALI_ERR_INVALID otherwise -- the net plan then runs the three-launch form). */
int dali_stem_fused_supported(int n, int h, int w);
int dali_stem_conv_bn_maxpool(dali_ctx* ctx, void* stream, const float* images, int n, int h, int w, const float* weight, const float* scale,
                              const float* shift, uint16_t* y);
/* 1x1 convolution (a [pixels][cin] x [cout][cin]^T GEMM) with the fused output stage:
 *   y = gate_{out_mask}( relu?( acc * out_scale[c] + out_shift[c] + bias[c] + res_scale[c] * residual ) ),  bits_out = (y > 0), 1 bit per
 *   element (res_scale: the downsample branch's BatchNorm scale when the residual is its raw convolution output).
 * Forward use: bn3 + identity + ReLU of a bottleneck inside conv3 (Encoders.py:330-339 over torchvision's Bottleneck), with the batch
 * statistics from dali_bnlin_fwd.  Backward use: the masked gradient dz = dy * (y > 0) leaves the data-gradient GEMM directly.
 * Every pointer after `cout` is nullable; cin % 32 == 0, cout % 8 == 0. */
int dali_conv1x1_fused(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, uint16_t* y, int pixels, int cin, int cout,
                       const float* out_scale, const float* out_shift, const float* bias, const uint16_t* residual, int out_relu,
                       uint8_t* bits_out, const uint8_t* out_mask, const float* res_scale);
/* y [pixels][cout] = [x1 | x2] @ w^T (+ bias): a 1x1 GEMM whose reduction runs over the c1 channels of x1 followed by the c2 channels of a SECOND
 * tensor x2 (both plain [pixels][c] bf16; w [cout][c1 + c2]).  Two chained data gradients that accumulate into one output become one launch and the
 * partial result is never stored: the conv3 backward of a Gram-scheme bottleneck, d_a2 = dz (A.W3) + W3^T Kc - a2 (W3^T diag(Q) W3)
 * (autograd of torchvision's Bottleneck tail under Encoders.py:330-339).  c1 % 32 == 0, c2 % 32 == 0, cout % 8 == 0; bias nullable. */
int dali_conv1x1_cat(dali_ctx* ctx, void* stream, const uint16_t* x1, int c1, const uint16_t* x2, int c2, const uint16_t* w, const float* bias,
                     uint16_t* y, int pixels, int cout);
/* The same two-operand GEMM with the scale / shift / ReLU output stage: y = relu?( ([x1 | x2] @ w^T) * out_scale[c] + out_shift[c] ) (out_scale
 * nullable = 1).  The inference forward of a bottleneck with a stride-1 downsample branch as ONE launch: with the two BatchNorms' scales folded
 * into the weight image w = [s3.W3 | sd.Wd] and out_shift = shift3 + shift_d this is relu(bn3(conv3(a2)) + bnd(convd(x))) (torchvision Bottleneck
 * under Encoders.py:336-339, eval mode: getFeatures.py:56-67).  c1 % 64 == 0, c2 % 64 == 0, cout % 128 == 0; c1 + c2 <= 256 (any pixel count
 * with at least two 128 x 128 tiles per CU) or c1 + c2 >= 1024 with cout >= 512 and >= 16384 pixels.
 * weight_parts = 2: w is [cout][2 (c1 + c2)] = [W1 hi | W1 lo | W2 hi | W2 lo] with hi = bf16(v), lo = bf16(v - hi): folded fp32 weights to ~2^-17
 * (a single bf16 image of scale-folded weights is a coherent 2^-9 perturbation that the ranking notices); 2 (c1 + c2) <= 256 only. */
int dali_conv1x1_cat_act(dali_ctx* ctx, void* stream, const uint16_t* x1, int c1, const uint16_t* x2, int c2, const uint16_t* w, int weight_parts,
                         const float* out_scale, const float* out_shift, int out_relu, uint16_t* y, int pixels, int cout);
/* Training-mode BatchNorm behind a 1x1 convolution WITHOUT the convolution's output (csrc/bnlin.hip): raw = a W^T is linear in
 * a [P][w] (bf16), so its batch statistics follow from gram = a^T a [w][w] and m2 = colsum(a) [w] (returned, fp32):
 * mean = W m2 / P, E[raw^2] = diag(W gram W^T) / P; ut = (W gram)^T [w][C] (fp32) is returned for the backward, with m2.  Outputs scale = gamma*invstd, shift = beta - mean*scale,
 * mean, invstd [C]; running statistics updated like torch (nullable).  W bf16 [C][w]; C % 8 == 0, w % 32 == 0. */
int dali_bnlin_fwd(dali_ctx* ctx, void* stream, const uint16_t* a, const uint16_t* W, int P, int C, int w, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* gram, float* m2,
                   float* ut, float* scale, float* shift, float* mean, float* invstd);
/* Its backward from the gradient dz [P][C] in front of the BatchNorm output: dW [C][w], dgamma, dbeta, and the two weight images of the
 * data gradient  d_a = dz wd1^T + a wd2^T + bvec  (wd1 = (scale.W)^T [w][C], wd2 = -(W^T diag(Q) W) [w][w], bvec = W^T Kc [w];
 * d_raw = scale dz + Kc - Q raw is nnops.hip's folded form of the BatchNorm backward). */
int dali_bnlin_bwd(dali_ctx* ctx, void* stream, const uint16_t* dz, const uint16_t* a, const uint16_t* W, int P, int C, int w,
                   const float* ut, const float* m2, const float* scale, const float* mean, const float* invstd, float* dW,
                   float* dgamma, float* dbeta, uint16_t* wd1, uint16_t* wd2, float* bvec);
/* rows of the `stats` buffer for a given problem (depends on the tile configuration the launcher will pick). */
int dali_conv2d_stat_tiles(int cout, int cin, int r, int s, int stride, int pad, int n, int ho, int wo, int fused_operand);
/* dx[n,h,w,cin] = conv_transpose(dy, w) (+ residual[n,h,w,cin] if given).  cout % 32 == 0, cin % 4 == 0.
 * residual_mask (nullable; needs cin % 8 == 0): 1 bit per residual element, bit e & 7 of byte e >> 3 over the flat
 * [n,h,w,cin] index, as dali_bn_act writes it for a block output: the residual is added only where the bit is set, i.e.
 * the identity path's dy * (y > 0) of a bottleneck (Encoders.py:330-351 autograd) is formed here, not stored. */
int dali_conv2d_dgrad(dali_ctx* ctx, void* stream, const uint16_t* dy, const uint16_t* wt, uint16_t* dx,
                      const uint16_t* residual, const uint8_t* residual_mask, int n, int h, int wd, int cin, int cout, int r, int s,
                      int stride, int pad);
/* dw[cout][r][s][cin] (fp32) = (accumulate ? dw : 0) + sum_p dy[p][cout] * x_gathered[p][r,s,cin]; x may carry
 * the same fused affine(+ReLU) as the forward.  Deterministic split-K (fixed-order slab reduction). */
int dali_conv2d_wgrad(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* dy, float* dw,
                      int n, int h, int wd, int cin, int cout, int r, int s, int stride, int pad,
                      const float* in_scale, const float* in_shift, int in_relu, int accumulate);

/* BatchNorm2d pieces (torch.nn.BatchNorm2d inside torchvision's Bottleneck, and Encoders.py:333).  Training-mode
 * statistics come from the conv epilogue partials; finalize turns them into scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale and updates the running statistics exactly like torch (momentum, unbiased var). */
int dali_bn_finalize(dali_ctx* ctx, void* stream, const float* partial, int tiles, int C, double count,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, float eps, float* scale, float* shift, float* mean, float* invstd);
/* y = relu?(raw*scale+shift + identity) with identity = identity tensor | raw2*scale2+shift2 | nothing (bottleneck tail).
 * mask_out (nullable, [pixels*C/8] bytes): bit t of byte i = (y[8i+t] > 0), the ReLU mask the backward needs, at 1/16 of
 * y's bytes. */
int dali_bn_act(dali_ctx* ctx, void* stream, const uint16_t* raw, const float* scale, const float* shift,
                const uint16_t* identity, const uint16_t* raw2, const float* scale2, const float* shift2, int relu,
                int64_t pixels, int C, uint16_t* y, uint8_t* mask_out);
/* Backward of z = bn(raw) followed by an optional ReLU whose mask is the bit mask ybits (dali_bn_act's mask_out) or
 * (ymask > 0) if one of them is given (they are exclusive), else (raw*scale+shift > 0).  g = gradient after the ReLU.  Writes d(raw) (bf16), dgamma, dbeta; with a second side
 * (raw_b...) the same masked gradient also flows through a second BatchNorm (the downsample branch).  dz_out
 * (nullable, may alias g) receives the masked gradient.  draw_a may alias g when dz_out is null. */
int dali_bn_bwd(dali_ctx* ctx, void* stream, const uint16_t* g, const uint16_t* ymask, const uint8_t* ybits, int relu,
                int64_t pixels, int C,
                const uint16_t* raw_a, const float* mean_a, const float* invstd_a, const float* scale_a, const float* shift_a,
                const uint16_t* raw_b, const float* mean_b, const float* invstd_b, const float* scale_b,
                float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, uint16_t* draw_a, uint16_t* draw_b,
                uint16_t* dz_out);
/* Stem tail (Encoders.py:333-335): out = maxpool3x3/2,pad1( raw*scale+shift ), no ReLU; arg = winning tap 0..8. */
int dali_maxpool_bn_fwd(dali_ctx* ctx, void* stream, const uint16_t* raw, const float* scale, const float* shift, int n, int h,
                        int w, int C, uint16_t* out, uint8_t* arg);
int dali_maxpool_bn_bwd(dali_ctx* ctx, void* stream, const uint16_t* dpool, const uint8_t* arg, const uint16_t* raw,
                        const float* mean, const float* invstd, const float* scale, int n, int h, int w, int C,
                        float* dgamma, float* dbeta, uint16_t* draw);
/* Head (Encoders.py:341-345): f[n,c] = mean_hw x + max_hw x (fp32), and its backward; mode = dali_feature
 * (GAP: mean only, GMP: max only -- evaluateCleanATModels.py:335-340). */
int dali_head_pool_fwd(dali_ctx* ctx, void* stream, const uint16_t* x, int n, int hw, int C, int mode, float* f, int16_t* arg);
int dali_head_pool_bwd(dali_ctx* ctx, void* stream, const float* df, const int16_t* arg, int n, int hw, int C, int mode,
                       uint16_t* dx);
/* BatchNorm1d neck (Encoders.py:350) on fp32 [n,C]. */
int dali_bn1d_fwd(dali_ctx* ctx, void* stream, const float* x, int n, int C, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, int training, float momentum, float eps, float* y, float* mean,
                  float* invstd);
int dali_bn1d_bwd(dali_ctx* ctx, void* stream, const float* x, const float* dy, int n, int C, const float* gamma,
                  const float* mean, const float* invstd, float* dx, float* dgamma, float* dbeta);

/* ---- TransReID ViT encoder pieces (vit_pytorch.py:120-184, 251-288, 375-408; make_models.py:184-205) -------- *
 * Tokens are [rows = B*T][C] bf16.  Linear layers (nn.Linear / the patch-embedding conv) run on the implicit-GEMM
 * engine: weights bf16 [N][K] (torch layout), the dgrad image is the transpose [K][N]. */
/* y = act(x @ w^T + bias) (+ residual); act 0 none / 1 exact-erf GELU (Mlp, vit_pytorch.py:120-136); pre (nullable)
 * receives the pre-activation.  K % 32 == 0, N % 4 == 0. */
int dali_linear_fwd(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, const float* bias, int act,
                    const uint16_t* residual, uint16_t* y, uint16_t* pre, int rows, int K, int N);
/* y = row_scale[row] * act(x @ w^T + bias) (+ residual): a residual branch under DropPath (vit_pytorch.py:45-62, applied at :338),
 * row_scale = the per-sample keep / (1 - p) factors expanded to token rows. */
int dali_linear_fwd_scaled(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* w, const float* bias, int act,
                           const uint16_t* residual, const float* row_scale, uint16_t* y, int rows, int K, int N);
/* dx = (dy @ wt^T) * gelu'(gelu_pre) (+ residual). */
int dali_linear_dgrad(dali_ctx* ctx, void* stream, const uint16_t* dy, const uint16_t* wt, const uint16_t* gelu_pre,
                      const uint16_t* residual, uint16_t* dx, int rows, int K, int N);
/* dw[N][K] fp32 = dy^T @ x; dbias[N] fp32 (nullable) = column sums of dy. */
int dali_linear_wgrad(dali_ctx* ctx, void* stream, const uint16_t* x, const uint16_t* dy, float* dw, float* dbias, int rows,
                      int K, int N);
/* PatchEmbed_overlap (vit_pytorch.py:251-288): images fp32 NCHW -> patches bf16 [B*ny*nx][3*patch*patch] in (c,r,s) order. */
int dali_vit_patchify(dali_ctx* ctx, void* stream, const float* img, int B, int H, int W, int patch, int stride, uint16_t* out);
/* x[b,0] = cls + pos[0]; x[b,1+i] = patch_emb[b,i] + pos[1+i] (vit_pytorch.py:379-391, camera = view = 0), and backward. */
int dali_vit_assemble_tokens(dali_ctx* ctx, void* stream, const uint16_t* patch_emb, const float* cls, const float* pos, int B, int T,
                             int C, uint16_t* x);
int dali_vit_assemble_tokens_bwd(dali_ctx* ctx, void* stream, const uint16_t* dx, int B, int T, int C, float* dpos, float* dcls,
                                 uint16_t* dpatch_emb);
/* nn.LayerNorm over C (eps 1e-6 in TransReID); backward optionally adds the residual-stream gradient `add`. */
int dali_layernorm_fwd(dali_ctx* ctx, void* stream, const uint16_t* x, const float* gamma, const float* beta, int rows, int C,
                       float eps, uint16_t* y, float* mean, float* rstd);
int dali_layernorm_bwd(dali_ctx* ctx, void* stream, const uint16_t* g, const uint16_t* x, const float* gamma, const float* mean,
                       const float* rstd, const uint16_t* add, int rows, int C, uint16_t* dx, float* dgamma, float* dbeta);
/* Attention (vit_pytorch.py:152-164): qkv [B*T][3*H*64] -> out [B*T][H*64] = softmax(q k^T * scale) v per head, scores on chip;
 * lse [B*H][T] (row log-sum-exp) feeds the backward, which recomputes the probabilities.  T <= 208, head_dim 64. */
int dali_attention_fwd(dali_ctx* ctx, void* stream, const uint16_t* qkv, int B, int T, int H, int head_dim, float scale,
                       uint16_t* out, float* lse);
int dali_attention_bwd(dali_ctx* ctx, void* stream, const uint16_t* qkv, const uint16_t* out, const uint16_t* d_out, const float* lse,
                       int B, int T, int H, int head_dim, float scale, uint16_t* dqkv);

/* ---- loss heads (train_encodersKIT.py:200-208), fp32 --------------------------------------------------- *
 * S is the similarity matrix fn @ C^T (dali_pairdist with DALI_METRIC_DOT).  labels are int32 codes shared between
 * the batch and the center / proxy label arrays; w[i] is the distortion weight table[samples_distortion[i]]
 * (losses.py:42-52).  Both losses are  sum_i num_i / sum_i den_i  with batch-global sums: *_fwd returns the local
 * sums[2] = {sum num, sum den} (all-reduce them across data-parallel ranks), *_bwd takes the global denominator. */

/* BatchWeightedCenterLoss (losses.py:39-88).  rowstat[nb][4] = {num_i, den_i, argmax_j, max_j softmax}. */
int dali_center_loss_fwd(dali_ctx* ctx, void* stream, const float* S, const int32_t* labels, const int32_t* center_labels,
                         const float* w, float tau, int nb, int NC, float* rowstat, float* sums);
/* dS[nb][NC] = gscale * d(sum num / denom)/dS. */
int dali_center_loss_bwd(dali_ctx* ctx, void* stream, const float* S, const int32_t* labels, const int32_t* center_labels,
                         const float* w, float tau, int nb, int NC, const float* denom, float gscale, float* dS);
/* BatchWeightedProxyLoss (losses.py:273-341).  rowstat[nb][2] = {num_i, den_i}; sel_idx / sel_coef [nb][2*K] hold the
 * selected proxies (K positives then K negatives, -1 padded, K = dali_proxy_kmax()) and d num_i / d S_ij.
 * status[0] != 0 if some row had more than K positives (they are truncated: treat as an error). */
int dali_proxy_loss_fwd(dali_ctx* ctx, void* stream, const float* S, const int32_t* labels, const int32_t* proxy_labels,
                        const float* w, float tau, int nb, int NP, float* rowstat, float* sums, int32_t* sel_idx,
                        float* sel_coef, int32_t* status);
/* dfn[nb][D] (= or +=) gscale/denom * sum_sel coef * proxies[sel][:]. */
int dali_proxy_loss_bwd(dali_ctx* ctx, void* stream, const int32_t* sel_idx, const float* sel_coef, const float* proxies, int nb,
                        int D, const float* denom, float gscale, int accumulate, float* dfn);
int dali_proxy_kmax(void);

/* ---- data parallel: gradient all-reduce over RCCL (xGMI), one communicator per context = per process = per GPU --------------------
 * Replaces nn.DataParallel's reduce of all gradients to GPU 0 (Encoders.py:39-40).  Rank 0 creates the 128-byte id
 * (dali_comm_unique_id) and the host program hands it to the other ranks; every rank then calls dali_ctx_comm_init with the
 * context's device current.  dali_allreduce_bucket: in-place SUM over all ranks of buf[0..count) fp32, enqueued on `stream`
 * (reduce-scatter + all-gather when count divides by the world size, else ncclAllReduce).  RCCL is bound at run time: without it
 * these return DALI_ERR_UNSUPPORTED and everything else works. */
int dali_comm_unique_id(void* id128);
int dali_ctx_comm_init(dali_ctx* ctx, const void* id128, int rank, int world);
int dali_ctx_comm_destroy(dali_ctx* ctx);
int dali_allreduce_bucket(dali_ctx* ctx, void* stream, float* buf, int64_t count);

/* ---- optimizer side of the hot loop, on the flat fp32 storages ------------------------------------------- */
/* torch.optim.Adam step (L2 weight decay added to the gradient; mainKIT.py:99, train_encodersKIT.py:214-216):
 * g' = grad_scale*g + wd*p; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  weights_sqsum (nullable, device) receives sum p^2 of the
 * updated parameters (the trainer's "weights_sum", train_encodersKIT.py:229-231). */
int dali_adam_step(dali_ctx* ctx, void* stream, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                   float* weights_sqsum);
/* momentum = beta*momentum + (1-beta)*online  (train_encodersKIT.py:218-226), flat. */
int dali_ema_update(dali_ctx* ctx, void* stream, float* momentum, const float* online, int64_t n, float beta);

/* ---- optional in-batch triplet head: BatchWeightedSoftmaxTripletLoss (losses.py:607-654) ------------------- *
 * S [nb,nb] = fn @ fn^T (dali_pairdist, DOT); labels int32 codes; w [nb] = the 13-entry distortion table
 * (losses.py:613-627) gathered per row.  Forward: rowstat [nb,2] = {w_i*softplus((s_neg - s_pos)/tau), w_i},
 * sums[2] = {numerator, denominator} (loss = sums[0]/sums[1]), sel_idx [nb,2] = {hardest positive, hardest
 * negative}, sel_coef [nb] = d row/d s_neg (= -d row/d s_pos); status[0] = 1 when some row has no negative (the
 * reference's topk raises there; such rows are skipped).  Backward: dS_sym [nb,nb] = (dS + dS^T) * gscale/denom,
 * so that d loss / d fn = dS_sym @ fn. */
int dali_triplet_loss_fwd(dali_ctx* ctx, void* stream, const float* S, const int32_t* labels, const float* w, float tau, int nb,
                          float* rowstat, float* sums, int32_t* sel_idx, float* sel_coef, int32_t* status);
int dali_triplet_loss_bwd(dali_ctx* ctx, void* stream, const int32_t* sel_idx, const float* sel_coef, int nb, const float* denom,
                          float gscale, float* dS_sym);

/* ---- epoch targets: class centers + farthest-point proxies (train_encodersKIT.py:113-156, :252-284) -------- *
 * fvs [n,d] fp32 un-normalised embeddings.  order [n]: row indices sorted by identity; bounds [n_classes+1]:
 * identity c owns order[bounds[c] .. bounds[c+1]).  first_pick [n_classes]: position (0 .. n_c-1, within the
 * identity's slice) of the first proxy -- the reference draws it with np.random.choice(n) (:257), so the host
 * keeps that stream.  Outputs: centers [n_classes,d] = L2-normalised class means (:131-137); proxies
 * [n_classes*num_proxies,d] = the chosen rows, L2-normalised (:139-141), zero rows where an identity has fewer than
 * num_proxies images; proxy_rows [n_classes*num_proxies] = chosen row of fvs or -1; max_dist [n_classes] = largest
 * pairwise distance among an identity's chosen rows (:278).  All pointers are device pointers.  Ties in the
 * arg-max go to the highest row position (the reference's argsort(...)[-1] leaves them unspecified). */
int dali_class_targets(dali_ctx* ctx, void* stream, const float* fvs, int n, int d, const int32_t* order,
                       const int32_t* bounds, int n_classes, const int32_t* first_pick, int num_proxies,
                       float* centers, float* proxies, int32_t* proxy_rows, float* max_dist);

/* ---- input pipeline on decoded uint8 images (decode stays on the host) -------------------------- *
 * Resize((H,W), bicubic) of getFeatures.py:18 / train_encodersKIT.py:313 = Pillow's two-pass ImagingResample on 8-bit
 * data, reproduced bit for bit: src = n packed RGB images [in_h][in_w][3] at byte offsets src_off; table_of[n] selects
 * the coefficient table of the image's size; bounds_* [tables][out][2] = (first input index, tap count) and coefs_*
 * [tables][out][ksize] = 22-bit fixed-point taps, built by the host as Pillow's precompute_coeffs does
 * (daliid_amd/transforms.py).  out = [n][out_h][out_w][3] uint8.  All pointers are device pointers. */
int dali_resize_bicubic_u8(dali_ctx* ctx, void* stream, const uint8_t* src, const int64_t* src_off, const int32_t* in_h,
                           const int32_t* in_w, const int32_t* table_of, int n, int max_in_h, const int32_t* bounds_h,
                           const int32_t* coefs_h, int ksize_h, const int32_t* bounds_v, const int32_t* coefs_v, int ksize_v,
                           int out_h, int out_w, uint8_t* out);
/* RandomCrop(padding) -> RandomHorizontalFlip -> ColorJitter(brightness, contrast, saturation) -> ToTensor ->
 * RandomErasing(value 0) -> Normalize (train_encodersKIT.py:313-320) of n resized uint8 images [n][h][w][3] with the
 * per-image parameters drawn by the host in torchvision's order: params [n][16] int32 words = crop top, crop left,
 * flip, 4 x op order (0 brightness, 1 contrast, 2 saturation, 3 hue = no-op), erase i, j, h, w (h = 0: none), the
 * three factors as float bits, crop padding, augment flag (0 = eval path: ToTensor + Normalize only,
 * getFeatures.py:18-19).  The enhancers follow PIL.ImageEnhance on 8-bit data exactly.  mean3 / std3 are HOST pointers
 * to 3 floats; out = fp32 [n][3][h][w].  One workgroup per image, the image lives in LDS (h*w*3 <= 150 KiB). */
int dali_augment_batch(dali_ctx* ctx, void* stream, const uint8_t* images, const int32_t* params, int n, int h, int w,
                       const float* mean3, const float* std3, float* out);

/* ---- measurement aid (bench.py roofline leg; no reference counterpart) ------------------------------- *
 * Between _begin and _end every MFMA GEMM kernel launch of the conv / linear path (class 0: igemm_conv_*
 * forward + dgrad, class 1: igemm_wgrad_*) is bracketed by two HIP events on the stream it is launched on.
 * _end waits for them and returns, per class, the summed kernel duration [ms], the summed 2*M*N*K FLOPs
 * and the launch count (arrays of 2).  Process-wide, not re-entrant; launches beyond max_launches are
 * not recorded. */
int dali_gemm_profile_begin(dali_ctx* ctx, int max_launches);
int dali_gemm_profile_end(dali_ctx* ctx, double* total_ms, double* total_flops, long long* launches);

/* ---- net plan: Encoders.ResNet50ReID forward / backward (Encoders.py:306-351) ------------------------ *
 * The plan owns topology + launch order; the caller owns storage: flat fp32 params / grads / BN running
 * statistics, and one byte arena (activations, bf16 weight images, scratch).  Tensor names and order follow
 * torchvision's state_dict ("conv1.weight", "bn1.weight", "layer1.0.conv1.weight", ..., "last_bn.bias"),
 * conv weights are stored [cout][r][s][cin] (the channels_last storage of a logical OIHW tensor). */
typedef struct dali_resnet dali_resnet;
typedef struct {
    int batch, height, width;   /* images: fp32 NCHW [batch,3,height,width]; height % 32 == 0, width % 16 == 0 */
    int layers[4];              /* bottlenecks per stage: {3,4,6,3} for ResNet-50 */
    int width_base;             /* 64 for ResNet-50 (32 allowed for test-size nets) */
} dali_resnet_cfg;

int dali_resnet_create(dali_ctx* ctx, const dali_resnet_cfg* cfg, dali_resnet** out);
int dali_resnet_destroy(dali_resnet* net);
int dali_resnet_sizes(const dali_resnet* net, int64_t* param_elems, int64_t* buffer_elems, int64_t* arena_bytes,
                      int* feat_dim, int* n_params, int* n_buffers);
/* kind 0: parameter, 1: buffer (running_mean / running_var).  offset/numel in fp32 elements of the flat storage. */
int dali_resnet_tensor_info(const dali_resnet* net, int kind, int index, char* name, int name_cap, int64_t* offset,
                            int64_t* numel, int* shape4, int* ndim);
int dali_resnet_stage_param_range(const dali_resnet* net, int stage, int64_t* begin, int64_t* end);
/* All storages 256-byte aligned; grads may be null for inference-only use. */
int dali_resnet_bind(dali_resnet* net, float* params, float* grads, float* buffers, void* arena, size_t arena_bytes);
/* Rebuild the bf16 operand images from the fp32 master weights (call after every optimizer step / state load). */
int dali_resnet_refresh_weights(dali_resnet* net, void* stream);
/* Pooling of the embedding head for the next forward/backward calls (dali_feature; default BOTH).  Mirrors
 * `model.module.feature = pooling` of evaluateCleanATModels.getWeightsByMagnitude (:249-256). */
int dali_resnet_set_feature(dali_resnet* net, int mode);
/* images fp32 NCHW -> emb fp32 [batch, feat_dim].  training != 0: batch statistics + running-stat update
 * (model.train()); else running statistics (model.eval()).  Stands under `model_online(batch_imgs)`
 * (train_encodersKIT.py:197) and `model(batch_gpu)` (getFeatures.py:61). */
int dali_resnet_forward(dali_resnet* net, void* stream, const float* images, int training, float* emb);
/* Backward of the last training forward, stages stage_begin..stage_end (0: neck+head+layer4, 1: layer3,
 * 2: layer2, 3: layer1+stem); fills the flat gradient buffer (overwrites, no accumulation).
 * Stands under `batch_loss.backward()` (train_encodersKIT.py:215). */
int dali_resnet_backward(dali_resnet* net, void* stream, const float* d_emb, int stage_begin, int stage_end);

int dali_resnet_debug_tensor(dali_resnet* net, const char* name, void** ptr, int64_t* bytes);

/* ---- net plan: TransReID ViT + BN neck (make_models.build_transformer.forward, make_models.py:184-205) ----- *
 * Same storage contract as dali_resnet_*.  forward: images fp32 NCHW -> feat fp32 [batch, dim] (after the
 * BatchNorm1d neck); global_feat (nullable) receives the pre-neck cls feature.  DropPath / Dropout are identity
 * (rate 0).  Limits: head_dim 64, at most 208 tokens. */
typedef struct dali_vit dali_vit;
typedef struct {
    int batch, height, width;   /* images fp32 NCHW [batch,3,height,width] */
    int patch, stride;          /* 16, 16 for ViT-B/16; stride < patch = overlapping patches (PatchEmbed_overlap) */
    int dim, depth, heads;      /* 768, 12, 12 */
    int mlp_hidden;             /* 3072 */
    int num_classes;            /* size of the unused `base.fc` head kept for state_dict compatibility (1000) */
} dali_vit_cfg;
int dali_vit_create(dali_ctx* ctx, const dali_vit_cfg* cfg, dali_vit** out);
int dali_vit_destroy(dali_vit* net);
int dali_vit_sizes(const dali_vit* net, int64_t* param_elems, int64_t* buffer_elems, int64_t* arena_bytes, int* feat_dim,
                   int* n_params, int* n_buffers);
int dali_vit_tensor_info(const dali_vit* net, int kind, int index, char* name, int name_cap, int64_t* offset, int64_t* numel,
                         int* shape4, int* ndim);
int dali_vit_bind(dali_vit* net, float* params, float* grads, float* buffers, void* arena, size_t arena_bytes);
int dali_vit_refresh_weights(dali_vit* net, void* stream);
int dali_vit_forward(dali_vit* net, void* stream, const float* images, int training, float* feat, float* global_feat);
int dali_vit_backward(dali_vit* net, void* stream, const float* d_feat);
/* Backward in stages = gradient buckets of the data-parallel reducer (daliid_amd/parallel.py): stage 0 = neck + final norm + the
 * last group of blocks (consumes d_feat), ..., the last stage also holds cls / pos / patch embedding.  Replaces the autograd
 * graph nn.DataParallel reduces as a whole (Encoders.py:39-40). */
int dali_vit_num_stages(const dali_vit* net);
int dali_vit_stage_param_range(const dali_vit* net, int stage, int64_t* begin, int64_t* end);
int dali_vit_backward_stages(dali_vit* net, void* stream, const float* d_feat, int stage_begin, int stage_end);
/* DropPath (vit_pytorch.py:45-62; per block rate linspace(0, drop_path_rate, depth), :338): scales = device fp32 [2*depth][batch],
 * row 2i = attention branch of block i, row 2i+1 = its MLP branch, each entry floor(keep + u)/keep in {0, 1/keep}.  Applied by
 * training-mode forwards (and mirrored by the backward) until reset with NULL; the buffer must outlive the backward. */
int dali_vit_set_drop_path(dali_vit* net, const float* scales);

#ifdef __cplusplus
}
#endif
#endif /* DALIID_H */
