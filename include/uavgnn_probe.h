/* libuavgnn - PROBE-ONLY entry points (exported by the same shared library, NOT part of the drop-in boundary of include/uavgnn.h).
 *
 * Building blocks and ablation hooks of the measurement tools under tools/ (cell_probe.py, cell_ablate.py, msg_probe.py): wired into no
 * shipped path.  Kept exported so that the probes run against the library that ships; declared apart so that nobody mistakes them for
 * components (round-5 review: csrc/gru_x3p.hip "is tested but wired into nothing"). */
#ifndef UAVGNN_PROBE_H_
#define UAVGNN_PROBE_H_

#include "uavgnn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The same cell from PREPARED operand planes (csrc/gru_x3p.hip): bit-identical results to uavgnn_gru_cell_fwd_x3 with no operand
 * split inside the kernel - staging a K slice is a linear LDS-DMA copy.  `planes`: the operand [inp || h] of the N rows as bf16
 * plane tiles, written by uavgnn_tarmac_msg_fwd (planes_out; K_in = H + M there); `h`: the same hidden state in fp32 (read by the
 * convex update); `tiles`: [W_ih | W_hh] as bf16 plane tiles per (64-unit column block, K slice), built once per weight version by
 * uavgnn_gru_split_weight_tiles (uavgnn_gru_weight_tiles_bytes(K_in, H) bytes, 16-byte aligned).  K_in % 32 == 0, H % 64 == 0. */
long long uavgnn_gru_weight_tiles_bytes(int K_in, int H);
int uavgnn_gru_split_weight_tiles(const float* W_ih, int K_in, const float* W_hh, int H, void* tiles, uavgnn_stream_t stream);
int uavgnn_gru_cell_fwd_planes(const void* planes, int K_in, const float* h, int N, int H, const void* tiles, const float* b_ih,
                               const float* b_hh, float* h_out, float* pre_save, uavgnn_stream_t stream);
/* ... with a per-call variant word `opt` (0 = uavgnn_gru_cell_fwd_planes; the schedules tools/cell_probe.py compares - bit 0:
 * second-half fragment reads behind the first MFMAs, bit 1: activation DMA two slices ahead (three LDS buffers), bit 2: the
 * epilogue's h tile requested inside the last slice, bit 3 (not with bit 1): the DMA of slice t + 2 issued behind the barrier of
 * slice t; results are bit-identical for every value; UAVGNN_EINVAL outside 0 .. 9, 12, 13). */
int uavgnn_gru_cell_fwd_planes_opts(const void* planes, int K_in, const float* h, int N, int H, const void* tiles,
                                    const float* b_ih, const float* b_hh, float* h_out, float* pre_save, int opt,
                                    uavgnn_stream_t stream);

/* ... with a variant word: bits 0-3 are timing ablations of tools/msg_probe.py (parts of the GEMM loop skipped: the outputs are then
 * WRONG); bit 4 (16) selects the one-wavefront-per-row-tile kernel where uavgnn_tarmac_msg_fwd runs the wavefront-pair kernel (no
 * planes_out, M + 2K <= 96) - correct results, the A/B reference; the two kernels sum the x and h halves of the projection in
 * different orders and may differ in the last bit */
int uavgnn_tarmac_msg_fwd_dbg(const float* x, int ld_x, const float* h, int ld_h, int N, int H, int n_ag, const void* tiles,
                              const float* bias, int M, int K, const int32_t* talk_off, const int32_t* talk_src, float scale,
                              float* c_out, int ld_c, float* a_save, float* proj_out, int ld_p, float* x_copy, int ld_xc,
                              void* planes_out, int dbg, uavgnn_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* UAVGNN_PROBE_H_ */
