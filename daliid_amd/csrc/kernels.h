// kernels.h -- host-side launchers shared between the single-op C ABI and the net plan.
#pragma once
#include "common.h"

namespace dali {

struct GatherGeom {
    int Hout, Wout;            // pixel grid the GEMM N dimension enumerates (n, ho, wo)
    int Hin, Win;              // grid of the gathered tensor (bounds)
    int Ck;                    // channels per tap of the gathered tensor (multiple of 32)
    int R, S, stride, pad;
    int mode;                  // 0: hi = ho*stride - pad + r ; 1 (dgrad): hi = (ho + pad - r) / stride if divisible
    long long img_pitch;       // elements between images of the gathered tensor
    int row_pitch, pix_pitch;  // elements between rows / pixels
    int lw, lhw;               // log2(Wout), log2(Hout*Wout) or -1
    // stride-2 dgrad by output parity (set by launch_igemm_conv; LDS-DMA kernels only): the GEMM enumerates the sub-grid
    // (n, ho', wo') of output positions (2ho'+oph, 2wo'+opw) of a Hfull x Wfull tensor; only the taps whose parity
    // matches contribute: kr = r0, r0+2, ... (nr of them), ks = s0, s0+2, ... (ns).  sub == 0: r0 = s0 = 0, all taps.
    // sub == 2: all four parity classes in ONE launch of 4 * class_grid workgroups; a workgroup derives its class from its block index
    // (parity_block, conv.hip: the 4-tap class first) and then runs as sub == 1
    int sub, oph, opw, Hfull, Wfull;
    int r0, rstep, nr, s0, sstep, ns;
    int class_grid;
};

struct IGemmArgs {
    const uint16_t* W;        // [Cm][R*S*Ck]
    const uint16_t* X;        // gathered tensor
    uint16_t* O;              // [P][Cm]
    const uint16_t* Res;      // optional residual [P][Cm], added in the epilogue
    const uint8_t* res_mask;  // optional 1-bit gate of Res (bit e & 7 of byte e >> 3, e = p*Cm + c): the ReLU mask of a block output,
                              // so that dz = dy * (y > 0) is formed here instead of being written by the BatchNorm backward
    const float* in_scale;    // optional [Ck] affine (+ReLU) applied to X on load
    const float* in_shift;
    float* stats;             // optional [tiles_n][Cm][2] partial sum / sumsq of the fp32 results
    const float* bias;        // optional [Cm] added before the activation (linear layers)
    uint16_t* O2;             // optional second output: the value BEFORE the activation (kept for the backward)
    const uint16_t* dact_pre; // optional [P][Cm]: multiply the result by gelu'(dact_pre) (backward through GELU)
    int act;                  // 0 none, 1 exact-erf GELU (vit_pytorch.py:120-136 nn.GELU)
    int Cm, P, in_relu;
    GatherGeom g;
    unsigned long long* stamps;   // diagnostic (dali_debug_set_conv_stamps): [block][12] s_memrealtime stamps (see scripts/conv_block_timeline.py)
    // ---- fused output stage (the EPI = 3 kernel instantiations; all optional) ----
    // value = acc * out_scale[c] + out_shift[c] (+ Res) -> ReLU if out_relu -> gated by out_mask -> O; bits_out gets (value > 0).
    // Forward: the BatchNorm of a 1x1 convolution whose batch statistics were derived from the Gram matrix of its input BEFORE the
    // GEMM ran (bnlin.hip), so that y = relu(bn(conv(x)) + identity) leaves the GEMM directly and the raw conv output is never stored.
    // Backward: out_mask = ReLU mask of the block output this gradient belongs to, so that dz = dy * (y > 0) is what is stored.
    const float* row_scale;   // optional [P] (linear layers, LIN instantiations): value = row_scale[p] * (acc + bias) before the residual is added:
                              // DropPath of a residual branch, vit_pytorch.py:45-62 (the caller expands the per-sample factors to rows)
    const float* res_scale;   // optional [Cm] (fused output stage): the residual enters as res_scale[c] * Res (the downsample branch's BatchNorm scale;
                              // its shift rides in `bias`)
    const float* out_scale;   // [Cm]
    const float* out_shift;   // [Cm]
    int out_relu;
    uint8_t* bits_out;        // 1 bit per output element, bit e & 7 of byte e >> 3, e = p*Cm + c (the layout of res_mask)
    const uint8_t* out_mask;  // same layout
    // ---- second operand tensor (the SRC2 instantiations of igemm_conv_dma_kernel; 1x1 / stride 1 only) ----
    // The GEMM's K dimension is the concatenation of X's Ck1 channels and X2's g.Ck - Ck1 channels, both [P][channels] bf16 (plain rows):
    // O = [X | X2] W^T with W [Cm][g.Ck].  Two chained data gradients that accumulate into the same output (bnlin.hip: dz (A.W3) and
    // -a2 (W3^T diag(Q) W3)) become one launch and the partial result is never stored.
    const uint16_t* X2;
    int Ck1;
    // x_rep (0 = 1; the persistent streaming kernel only): K = x_rep * (Ck1 + c2): each tensor's channels appear x_rep times in a row, against a
    // split weight image [W1 hi | W1 lo | W2 hi | W2 lo] -- fp32-grade weights (BatchNorm scales folded in) on the bf16 matrix pipe
    int x_rep;
};

struct WGradArgs {
    const uint16_t* dY;       // [P][Cm]
    const uint16_t* X;        // gathered tensor
    float* partial;           // [splits][Cm][Ntot]
    const float* in_scale; const float* in_shift;
    int Cm, P, Ntot, in_relu;
    int splits, pix_per_split;    // pix_per_split multiple of 32
    GatherGeom g;
    unsigned long long* stamps;   // diagnostic, as in IGemmArgs
    int ablate;                   // diagnostic (DALI_WGRAD_ABLATE, igemm_wgrad3x3_kernel): 1 = no operand requests after the ring's prologue (the k-steps keep
                                  // their reads, MFMAs and barriers), 2 = requests and barriers only (no fragment reads, no MFMAs); results are wrong
    float* colsum;                // optional [splits][Cm]: per-split column sums over the pixels of dY, by one extra MFMA per fragment against a
                                  // ones operand in the n-tile-0 workgroups (128 x 128 kernel only: wgrad_colsum_supported); bnlin.hip's s / m2
};


// one [Co][T][Ci] -> [Ci][T][Co] bf16 weight transpose (dgrad image); first_block is filled by the launcher
struct TransposeJob { const uint16_t* w; uint16_t* wt; int Co, T, Ci, first_block; };
int launch_weight_transpose_batched(hipStream_t st, const TransposeJob* jobs, int n);

struct BnBwdSide { const uint16_t* raw; const float* mean; const float* invstd; const float* scale; const float* shift; };

// conv.hip
int launch_igemm_conv(hipStream_t st, const IGemmArgs& a);
// per-launch event bracket of the GEMM profile steps (dali_gemm_profile_*), for GEMM launches outside conv.hip; cls 0 = forward / data gradient
int gemm_profile_begin(hipStream_t st, int cls, double flops, const char* what);
void gemm_profile_end(hipStream_t st, int slot);
int igemm_conv_stat_tiles(int Cm, int P, int K);
bool conv_cat_act_supported(int Cm, int c1, int c2, int P, int parts = 1);      // [X | X2] GEMM with the scale / shift / ReLU output stage at this size?
// taps = R*S of the convolution (1 for 1x1 convolutions and linear layers): selects the tile shape
// halo_w = output width of a 3x3 / stride 1 / pad 1 convolution whose H*W is a power of two (0 otherwise): enables the halo kernel
void wgrad_plan(int Cm, int Ntot, int P, int target_blocks, int* splits, int* pix_per_split, size_t* ws_bytes, int taps = 1, int halo_w = 0);
// out == nullptr: leave the split-K slabs in a.partial ([splits][Cm][Ntot]) for the caller to reduce
// colsum_out (nullable): where the column sums that rode on the GEMM (a.colsum, colsum_rows partial rows) are reduced to, in the slab reduce's launch
int launch_igemm_wgrad(hipStream_t st, const WGradArgs& a, float* out, int accumulate, float* colsum_out = nullptr, int colsum_rows = 0);
int launch_splitk_reduce2(hipStream_t st, const float* partial, float* out, size_t elems, int splits, int accumulate,
                          const float* partial2, float* out2, size_t elems2, int splits2);
bool wgrad_colsum_supported(int Cm, int Ntot, int taps, int P);
// out[e] (+)= sum_k partial[k][e], fixed order
int launch_splitk_reduce(hipStream_t st, const float* partial, float* out, size_t elems, int splits, int accumulate);

int launch_linear_fwd(hipStream_t st, const uint16_t* x, const uint16_t* w, const float* bias, int act, const uint16_t* residual, uint16_t* y,
                      uint16_t* pre, const uint16_t* dact_pre, int rows, int K, int N, const float* row_scale = nullptr);
int launch_linear_wgrad(hipStream_t st, const uint16_t* x, const uint16_t* dy, float* dw, int rows, int K, int N, float* slab,
                        float* dbias = nullptr, float* cs_partial = nullptr, bool* bias_done = nullptr);
size_t linear_wgrad_colsum_floats(int rows, int K, int N);
int wgrad_colsum_rows(int Cm, int Ntot, int taps, int P, int splits);
size_t linear_wgrad_slab_bytes(int rows, int K, int N);

// bnlin.hip: BatchNorm behind a 1x1 convolution from the moments of the convolution's INPUT (no raw conv output is stored)
//   forward : scale / shift / mean / invstd of raw = a W^T from gram = a^T a [w][w] and m2 = colsum(a) [w]; W [C][w] and Wt = W^T [w][C] bf16;
//             leaves ut = (W gram)^T [w][C] for the backward, dot is scratch of (w/32) * C * 12 bytes (fp32 quadratic-form + fp64 mean partials per 32-row tile)
//   backward: split-K slabs of G0 = dz^T a (+ s_dz = colsum(dz)) -> dW [C][w], dgamma, dbeta, and the two data-gradient weight images
//             wd1 = (A.W)^T [w][C], wd2 = -(W^T diag(Q) W) [w][w] with bvec = W^T Kc [w]; qk [2][C] is scratch
int launch_bnlin_stats(hipStream_t st, const uint16_t* W, const uint16_t* Wt, const float* gram, const float* m2, int C, int w, double count, const float* gamma,
                       const float* beta, float* rm, float* rv, float momentum, float eps, float* ut, float* dot, float* scale, float* shift,
                       float* mean, float* invstd);
// dW holds G0 = dz^T a (the reduced weight-gradient GEMM) on entry and the weight gradient on return; s_dz = colsum(dz) [C]
// ld1 / ld2: row pitch (elements) of wd1 / wd2 (the net plan writes both into one [w][C + w] image: ld1 = ld2 = C + w, wd2 = wd1 + C)
int launch_bnlin_ut(hipStream_t st, const uint16_t* W, const float* gram, int C, int w, float* ut);
int launch_bnlin_bwd(hipStream_t st, const uint16_t* W, const uint16_t* Wt, const float* ut, const float* m2, const float* s_dz, int C,
                     int w, double count, const float* scale, const float* mean, const float* invstd, float* dW, float* dgamma, float* dbeta,
                     uint16_t* wd1, uint16_t* wd2, float* bvec, float* qk, int ld1 = 0, int ld2 = 0);

// nnops.hip
constexpr int REDUCE_SMAX = 64;        // rows of the fp64 second-level scratch
inline size_t reduce_scratch_bytes(int C, int NV) { return (size_t)REDUCE_SMAX * C * NV * sizeof(double); }
int launch_bn_finalize(hipStream_t st, const float* partial, int tiles, int C, double count, const float* gamma, const float* beta,
                       float* rm, float* rv, float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                       double* scratch);
int launch_bn_eval_coeffs(hipStream_t st, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                          float* scale, float* shift);
// many bn_eval_coeffs in one launch (kernel arguments hold the job table: at most BnEvalJobs::MAX jobs per launch)
struct BnEvalJob { const float *gamma, *beta, *rm, *rv; float *scale, *shift; int C; };
struct BnEvalJobs { static constexpr int MAX = 56; BnEvalJob job[MAX]; };
int launch_bn_eval_coeffs_batched(hipStream_t st, const BnEvalJob* jobs, int n, float eps);
int launch_fold_cat_weights(hipStream_t st, const float* w3, const float* wd, const float* s3, const float* sd, const float* h3, const float* hd, int C, int w,
                            int cin, int parts, uint16_t* wcat, float* shcat);
int launch_bn_act(hipStream_t st, const uint16_t* raw, const float* scale, const float* shift, const uint16_t* idn, const uint16_t* raw2,
                  const float* scale2, const float* shift2, int relu, size_t elems, int C, uint16_t* y, uint8_t* mask_out);
int bn_bwd_blocks(int P, int C, int* rows_per_block);
size_t bn_bwd_partial_floats(int P, int C, bool dual);
int launch_bn_bwd(hipStream_t st, const uint16_t* g, const uint16_t* ymask, const uint8_t* ybits, const BnBwdSide& a, const BnBwdSide* b, int relu, int P, int C,
                  float* partial, float* coef_a, float* coef_b, float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b,
                  uint16_t* draw_a, uint16_t* draw_b, uint16_t* dz_out, double* scratch);
int launch_stem_pack_image(hipStream_t st, const float* img, int N, int H, int W, uint16_t* out);
int launch_stem_pack_weight(hipStream_t st, const float* w, int Cout, uint16_t* out);
// stem.hip: the inference stem (conv1 -> bn1 by running statistics -> 3x3 / 2 max-pool) in one launch, packed image -> pooled [N][H/4][W/4][64]
bool stem_fused_supported(int N, int H, int W, int C);
// the training stem convolution (raw0 + per-tile BatchNorm partial sums [stem_train_tiles][64][2]) on the same patch machinery
bool stem_train_supported(int N, int H, int W, int C);
int stem_train_tiles(int N, int H);
int launch_stem_conv_stats(hipStream_t st, const uint16_t* ximg, const uint16_t* w_packed, int N, int H, int W, uint16_t* raw, float* stats);
int launch_stem_conv_bn_pool(hipStream_t st, const uint16_t* ximg, const uint16_t* w_packed, const float* scale, const float* shift, int N, int H, int W,
                             uint16_t* out);
int launch_stem_unpack_wgrad(hipStream_t st, const float* padded, int Cout, float* dw);
int launch_maxpool_bn_fwd(hipStream_t st, const uint16_t* raw, const float* scale, const float* shift, int N, int H, int W, int C,
                          uint16_t* out, uint8_t* arg);
int launch_maxpool_bn_bwd(hipStream_t st, const uint16_t* dp, const uint8_t* arg, const uint16_t* raw, const float* mean, const float* invstd,
                          const float* scale, int N, int H, int W, int C, float* partial, float* coef, float* dgamma, float* dbeta,
                          uint16_t* draw, double* scratch);
int launch_head_pool_fwd(hipStream_t st, const uint16_t* x, int N, int HW, int C, int mode, float* f, int16_t* arg);
// ybits (nullable): ReLU mask of the tensor the gradient belongs to; the masked gradient dz = dy * (y > 0) is what is written
int launch_head_pool_bwd(hipStream_t st, const float* df, const int16_t* arg, int N, int HW, int C, int mode, uint16_t* dx, const uint8_t* ybits = nullptr);
int launch_bn1d_fwd(hipStream_t st, const float* x, int N, int C, const float* gamma, const float* beta, float* rm, float* rv, int training,
                    float momentum, float eps, float* y, float* mean, float* invstd);
int launch_bn1d_bwd(hipStream_t st, const float* x, const float* dy, int N, int C, const float* gamma, const float* mean, const float* invstd,
                    float* dx, float* dgamma, float* dbeta);
int launch_cast_bf16(hipStream_t st, const float* x, size_t n, uint16_t* y);
int launch_weight_transpose(hipStream_t st, const uint16_t* w, int Co, int T, int Ci, uint16_t* wt);


// vit_ops.hip
int launch_patchify(hipStream_t st, const float* img, int B, int H, int W, int ps, int stride, uint16_t* out);
int launch_assemble_tokens(hipStream_t st, const uint16_t* pe, const float* cls, const float* pos, int B, int T, int C, uint16_t* x);
int launch_assemble_tokens_bwd(hipStream_t st, const uint16_t* dx, int B, int T, int C, float* dpos, float* dcls, uint16_t* dpe);
int launch_layernorm_fwd(hipStream_t st, const uint16_t* x, const float* gamma, const float* beta, int rows, int C, float eps,
                         uint16_t* y, float* mean, float* rstd, float* y32);
size_t layernorm_bwd_partial_floats(int rows, int C);
int launch_layernorm_bwd(hipStream_t st, const uint16_t* g, const uint16_t* x, const float* gamma, const float* mean, const float* rstd,
                         const uint16_t* add, int rows, int C, uint16_t* dx, float* dgamma, float* dbeta, float* partial, double* scratch,
                         const float* g32 = nullptr);
size_t colsum_partial_floats(int rows, int C);
int launch_colsum(hipStream_t st, const uint16_t* y, int rows, int C, float* out, float* partial, double* scratch);
int launch_colsum_partials(hipStream_t st, const uint16_t* y, int rows, int C, float* partial, int* n_rows);
// out = (res or 0) + scale[row / rows_per_sample] * branch   (DropPath per sample; C % 8 == 0)
int launch_rowscale_add(hipStream_t st, const uint16_t* branch, const float* scale, int samples, int rows_per_sample, int C, const uint16_t* res,
                        uint16_t* out);
int launch_expand_rowscale(hipStream_t st, const float* scale, int n_vec, int B, int T, float* out);
int launch_attention_fwd(hipStream_t st, const uint16_t* qkv, int B, int T, int H, float scale, uint16_t* out, float* lse);
int launch_attention_bwd(hipStream_t st, const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, int B, int T, int H,
                         float scale, uint16_t* dqkv);

}  // namespace dali
