/* daliid_debug.h -- diagnostic entry points of libdaliid_hip.so.  NOT part of the drop-in surface (include/daliid.h):
 * nothing under daliid_amd/ binds them; scripts/ (A/B timing, in-kernel stamps) and two tests do, through ctypes.
 * Declared here so that the shared library exports nothing that no header names (tests/test_abi.py compares the
 * two headers with `nm -D` of the library).
 */
#ifndef DALIID_DEBUG_H
#define DALIID_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* the DALI_* A/B switches (daliid_amd/csrc/common.h) are re-read from the environment at their next use, so that one
 * process can time two variants back to back on the same box */
int dali_debug_reload_env(void);

/* device buffer of 12 x uint64 per workgroup that the convolution / weight-gradient kernels fill with s_memrealtime
 * stamps (100 MHz); null switches the stamps off (scripts/conv_block_timeline.py) */
int dali_debug_set_conv_stamps(void* dev_ptr);

/* the split count the weight-gradient plan picks for a layer (returned as the status value, >= 1) */
int dali_debug_wgrad_splits(int Cm, int Ntot, int P, int taps, int halo_w);

/* the split-K reduce alone: out[e] (+)= sum_k partial[k][e], fixed order (tests/test_gpu_conv.py, scripts/bench_reduce.py) */
int dali_debug_splitk_reduce(void* stream, const float* partial, float* out, long long elems, int splits, int accumulate);

/* the ResNet plan's backward ONE bottleneck at a time (last block first: that call also runs the neck + head backward from d_emb; after
 * block 0, block = -1 runs the stem), so that a test can read the gradient entering every block (`grad_cur` of dali_resnet_debug_tensor); same launches
 * in the same order as dali_resnet_backward (tests/test_gpu_resnet_blocks.py, the batch-256 composition check) */
struct dali_resnet;
int dali_debug_resnet_backward_block(struct dali_resnet* net, void* stream, const float* d_emb, int block);

#ifdef __cplusplus
}
#endif
#endif
