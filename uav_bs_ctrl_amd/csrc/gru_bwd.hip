// Backward of the fused GRU cell as ONE C-ABI entry, and the single workspace query of the library ("what a
// C-ABI replacement must export": uavgnn_gru_cell_{fwd,bwd}, uavgnn_workspace_bytes(kind, sizes...)).
//
// uavgnn_gru_cell_bwd = autograd of nn.GRUCell (/root/reference/algos/madrqn/agents/gnn_agents.py:246,:270 under
// algos/madrqn/learner.py:157) for the INPUT side of the cell, in the order the Python host issues it (ops._TarmacStep.backward):
//   1. gate gradients from the saved pre-activation sets      d_gi, d_gh [N, 3H], d_h = d_hout * z      (gru_fused.hip)
//   2. d_inp [N, K_in]  = d_gi W_ih                                                                     (gemm_x3.hip, bf16x3)
//   3. d_h   [N, H]    += d_gh W_hh                                                                     (gemm_x3.hip, bf16x3)
// The weight / bias gradients are NOT formed here: their operands (d_gi, d_gh; the caller's inp, h) stay in the caller's buffers,
// because a BPTT caller reduces them once per sequence, not once per step (ops.WeightGradSink.end_sequence) - a one-step caller
// finishes with uavgnn_gemm_tn_x3 (dW_ih = d_gi^T inp, dW_hh = d_gh^T h) and uavgnn_colsum_acc (db_ih = colsum d_gi, db_hh =
// [colsum d_gi[:, :2H] | colsum d_gh[:, 2H:]]).  A non-Python binder does not have to replay the host sequence.
#include "common.h"

extern "C" long long uavgnn_gru_cell_bwd_workspace_bytes(int K_in, int H) {
  if (K_in <= 0 || H <= 0) return 0;
  return 6LL * 3 * H * (static_cast<long long>(K_in) + H);   // bf16 planes [3][K_in][3H] of W_ih^T, then [3][H][3H] of W_hh^T
}

extern "C" int uavgnn_gru_split_weights_bwd(const float* W_ih, int K_in, const float* W_hh, int H, void* planes,
                                            uavgnn_stream_t stream) {
  if (!W_ih || !W_hh || !planes || K_in <= 0 || H <= 0) return UAVGNN_EINVAL;
  int rc = uavgnn_split_bf16x3(W_ih, K_in, 3 * H, K_in, 1, planes, stream);
  if (rc) return rc;
  return uavgnn_split_bf16x3(W_hh, H, 3 * H, H, 1, static_cast<char*>(planes) + 6LL * 3 * H * K_in, stream);
}

extern "C" int uavgnn_gru_cell_bwd(const float* pre, const float* h, const float* d_hout, int N, int K_in, int H,
                                   const void* planes_bwd, float* d_gi, float* d_gh, float* d_inp, int ld_dinp, float* d_h,
                                   uavgnn_stream_t stream) {
  if (N < 0 || !pre || !h || !d_hout || !planes_bwd || !d_gi || !d_gh || !d_inp || !d_h || ld_dinp < K_in || K_in <= 0 || H <= 0)
    return UAVGNN_EINVAL;
  if (N == 0) return 0;
  if (!uavgnn_gemm_x3_supported(N, K_in, 3 * H) || !uavgnn_gemm_x3_supported(N, H, 3 * H)) return UAVGNN_EUNSUPPORTED;
  int rc = uavgnn_gru_gates_bwd_fused(pre, h, d_hout, N, H, d_gi, d_gh, d_h, stream);
  if (rc) return rc;
  rc = uavgnn_gemm_nt_x3(d_gi, 3 * H, N, 3 * H, planes_bwd, K_in, nullptr, d_inp, ld_dinp, 0, stream);
  if (rc) return rc;
  return uavgnn_gemm_nt_x3(d_gh, 3 * H, N, 3 * H, static_cast<const char*>(planes_bwd) + 6LL * 3 * H * K_in, H, nullptr, d_h, H,
                           UAVGNN_GEMM_ACCUMULATE, stream);
}

// One query for every caller-provided scratch / plane buffer of the library (bytes; 0 = unknown kind or bad sizes).
extern "C" long long uavgnn_workspace_bytes(int kind, long long a, long long b, long long c) {
  switch (kind) {
    case UAVGNN_WS_GATV2_BWD: return static_cast<long long>(uavgnn_gatv2_bwd_workspace_bytes(static_cast<int>(a), static_cast<int>(b)));
    case UAVGNN_WS_DEGREE_ORDER: return static_cast<long long>(uavgnn_degree_order_workspace_bytes(static_cast<int>(a)));
    case UAVGNN_WS_CSC_TRANSPOSE: return static_cast<long long>(uavgnn_csc_transpose_workspace_bytes(static_cast<int>(a)));
    case UAVGNN_WS_GRU_PLANES: return uavgnn_gru_cell_x3_workspace_bytes(static_cast<int>(a), static_cast<int>(b));
    case UAVGNN_WS_GRU_BWD_PLANES: return uavgnn_gru_cell_bwd_workspace_bytes(static_cast<int>(a), static_cast<int>(b));
    case UAVGNN_WS_GEMM_PLANES: return (a > 0 && b > 0) ? 6LL * a * b : 0;
    case UAVGNN_WS_GEMM_TN_PARTIALS: {
      const int S = uavgnn_gemm_tn_x3_chunks(a, static_cast<int>(b), static_cast<int>(c));
      return 4LL * S * b * c;
    }
    case UAVGNN_WS_K1_IMAGE: return static_cast<long long>(uavgnn_gatv2_hetero_image_bytes());
    default: return 0;
  }
}
