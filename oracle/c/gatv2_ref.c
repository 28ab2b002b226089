/*
 * Plain-C scalar restatement of the GATv2 relation of the hot path - TEST INFRASTRUCTURE ONLY.
 * Third, independent formulation next to oracle/restatement.py (segment ops) and oracle/dense_fp64.py (dense masks):
 * straight loops in double precision, one destination at a time.  Follows the reference call
 * dglnn.GATv2Conv((F_src,2), D, nh, residual=True, allow_zero_in_degree=True, activation=ReLU)
 * at /root/reference/algos/madrqn/agents/gnn_agents.py:93-96,:103-104 (DGL 0.9.0 semantics restated; parity unpinned,
 * see oracle/restatement.py).  Built by oracle/c/Makefile into oracle/c/libgatv2_ref.so; used by tests/ only.
 */
#include <math.h>
#include <stdlib.h>

/* x_src[E,Fs] x_dst[N,Fd] seg_off[N+1]; W_s[H,Fs] b_s[H] W_d[H,Fd] b_d[H] attn[H] W_r[H,Fd] b_r[H]; out[N,H] */
int gatv2_ref_forward(const double* x_src, int Fs, const double* x_dst, int Fd, const int* seg_off, int N,
                      const double* W_s, const double* b_s, const double* W_d, const double* b_d, const double* attn,
                      const double* W_r, const double* b_r, int nh, int D, double slope, double* out) {
  const int H = nh * D;
  for (int v = 0; v < N; ++v) {
    const int e0 = seg_off[v], deg = seg_off[v + 1] - e0;
    double* o = out + (size_t)v * H;
    double* score = (double*)malloc(sizeof(double) * (deg > 0 ? deg : 1));
    for (int k = 0; k < nh; ++k) {
      double mx = -INFINITY;
      for (int u = 0; u < deg; ++u) {            /* e = attn . lrelu(el + er) for this head */
        double e = 0.0;
        for (int d = 0; d < D; ++d) {
          const int n = k * D + d;
          double z = b_s[n] + b_d[n];
          for (int f = 0; f < Fs; ++f) z += W_s[n * Fs + f] * x_src[(size_t)(e0 + u) * Fs + f];
          for (int f = 0; f < Fd; ++f) z += W_d[n * Fd + f] * x_dst[(size_t)v * Fd + f];
          e += attn[n] * (z > 0 ? z : slope * z);
        }
        score[u] = e;
        if (e > mx) mx = e;
      }
      double den = 0.0;
      for (int u = 0; u < deg; ++u) { score[u] = exp(score[u] - mx); den += score[u]; }
      for (int d = 0; d < D; ++d) {              /* sum_u a el[u] + residual, ReLU */
        const int n = k * D + d;
        double acc = 0.0;
        for (int u = 0; u < deg; ++u) {
          double el = b_s[n];
          for (int f = 0; f < Fs; ++f) el += W_s[n * Fs + f] * x_src[(size_t)(e0 + u) * Fs + f];
          acc += score[u] / den * el;
        }
        double r = b_r ? b_r[n] : 0.0;
        for (int f = 0; f < Fd; ++f) r += W_r[n * Fd + f] * x_dst[(size_t)v * Fd + f];
        const double t = acc + r;
        o[n] = t > 0 ? t : 0.0;
      }
    }
    free(score);
  }
  return 0;
}
