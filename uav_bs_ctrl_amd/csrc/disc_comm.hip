// K5: DiscreteComm message passing (forward / backward).
//
// Replaces the DGL UDF path of /root/reference/algos/madrqn/agents/gnn_agents.py:166-178,:189
//   msg_func:  F.gumbel_softmax(f_enc([x_u || h_u]).view(-1, msg, 2), tau=0.5, hard=True).flatten(1)   per edge
//   aggr_func: nodes.mailbox['m'].max(1)[0]                                                            per destination
// The encoder logits depend on the SOURCE node only, so they arrive per node [N, 2*msg]; the Gumbel noise is per edge
// ([E, msg, 2] in CSC order, an explicit input - SURVEY "hard parts").  One wavefront per destination, lane <-> bit
// pair; every in-edge contributes the straight-through value (y_hard - y_soft) + y_soft and the element-wise max is
// taken ("OR" of one-hot bits).
// Gradient routing: torch.max(dim) sends the gradient of a channel to ONE in-edge, the first maximal entry in mailbox
// order.  The candidates are 1 or 0 up to one rounding of (1 - s) + s, so which edge wins a tie in the reference is
// decided by rounding noise; this kernel uses the exact-arithmetic rule - the first in-edge (CSC order = edge-id order)
// whose hard bit is set, else the first in-edge - and records it in `sel` for the backward.
// Backward is a gather over the transposed CSC (per source), deterministic, no atomics.
//
// Noise.  The reference draws gumbels = -log(Exponential(1)) per (edge, channel, class) from torch's global generator
// (F.gumbel_softmax).  With `gumbel == NULL` the kernel draws them itself: counter-based Philox4x32-10 keyed by a 64-bit
// seed, counter = (CSC position of the edge, channel, step, 0) - one call per (edge, channel) yields both classes' uniforms,
// g = -log(-log(u)) - so the [E, msg, 2] noise tensor (512 B per edge at msg = 64: written by exponential_() + log and read
// back) never exists, any launch is reproducible from (seed, step), and nothing depends on launch geometry.  seed / step are
// read from DEVICE memory (a captured hipGraph replays with the current step).  The injected-noise path stays for fixtures
// that carry the reference's own draws; uavgnn_gumbel_noise materialises the same stream for tests.
#include "common.h"

namespace uavgnn {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter c[4], key k[2]
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// the two Gumbel draws of (edge position e, channel i) at `step`: u = (23 random bits + 1/2) 2^-23 in (0, 1)
__device__ __forceinline__ float2 gumbel_pair(unsigned long long seed, unsigned long long step, int e, int i) {
  uint32_t c[4] = {static_cast<uint32_t>(e), static_cast<uint32_t>(i), static_cast<uint32_t>(step),
                   static_cast<uint32_t>(step >> 32)};
  philox4x32_10(c, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  const float u0 = (static_cast<float>(c[0] >> 9) + 0.5f) * 1.1920928955078125e-7f;   // k + 1/2 is exact in fp32 for k < 2^23
  const float u1 = (static_cast<float>(c[1] >> 9) + 0.5f) * 1.1920928955078125e-7f;
  return make_float2(-logf(-logf(u0)), -logf(-logf(u1)));
}

__global__ __launch_bounds__(256) void gumbel_noise_kernel(const long long* __restrict__ rng, long long E, int M,
                                                           float* __restrict__ out) {
  const unsigned long long seed = static_cast<unsigned long long>(rng[0]), step = static_cast<unsigned long long>(rng[1]);
  for (long long q = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; q < E * M; q += static_cast<long long>(gridDim.x) * 256) {
    const int e = static_cast<int>(q / M), i = static_cast<int>(q - static_cast<long long>(e) * M);
    *reinterpret_cast<float2*>(out + 2 * q) = gumbel_pair(seed, step, e, i);
  }
}

__global__ __launch_bounds__(kThreads) void disc_comm_fwd_kernel(
    const float* __restrict__ logits, int ld, const float* __restrict__ gumbel, const long long* __restrict__ rng, int M,
    const int32_t* __restrict__ talk_off, const int32_t* __restrict__ talk_src, int N, float inv_tau,
    float* __restrict__ c, int ld_c, float* __restrict__ y0_save, int32_t* __restrict__ sel) {
  unsigned long long seed = 0, step = 0;
  if (gumbel == nullptr) {
    seed = static_cast<unsigned long long>(rng[0]);
    step = static_cast<unsigned long long>(rng[1]);
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int d = blockIdx.x * kWavesPerBlock + wave; d < N; d += gridDim.x * kWavesPerBlock) {
    const int e0 = talk_off[d], e1 = talk_off[d + 1];
    for (int i = lane; i < M; i += kWave) {
      float best0 = 0.f, best1 = 0.f;
      int s0 = -1, s1 = -1;
      bool hit0 = false, hit1 = false;
      for (int e = e0; e < e1; ++e) {
        const int u = talk_src[e];
        const float2 l = *reinterpret_cast<const float2*>(logits + static_cast<size_t>(u) * ld + 2 * i);
        const float2 gn = gumbel != nullptr ? *reinterpret_cast<const float2*>(gumbel + (static_cast<size_t>(e) * M + i) * 2)
                                            : gumbel_pair(seed, step, e, i);
        const float t0 = (l.x + gn.x) * inv_tau, t1 = (l.y + gn.y) * inv_tau;
        const float mx = fmaxf(t0, t1);
        const float x0 = expf(t0 - mx), x1 = expf(t1 - mx);
        const float den = x0 + x1;
        const float y0 = x0 / den, y1 = x1 / den;
        y0_save[static_cast<size_t>(e) * M + i] = y0;
        const bool h0 = y0 >= y1;                       // argmax, ties -> index 0 (torch.max semantics)
        const float m0 = ((h0 ? 1.f : 0.f) - y0) + y0;  // straight-through value, same expression as the reference
        const float m1 = ((h0 ? 0.f : 1.f) - y1) + y1;
        if (e == e0) { best0 = m0; best1 = m1; s0 = e; s1 = e; hit0 = h0; hit1 = !h0; }
        if (h0 && !hit0) { best0 = m0; s0 = e; hit0 = true; }
        if (!h0 && !hit1) { best1 = m1; s1 = e; hit1 = true; }
      }
      c[static_cast<size_t>(d) * ld_c + 2 * i] = best0;       // zero for destinations without in-edges
      c[static_cast<size_t>(d) * ld_c + 2 * i + 1] = best1;
      sel[static_cast<size_t>(d) * 2 * M + 2 * i] = s0;
      sel[static_cast<size_t>(d) * 2 * M + 2 * i + 1] = s1;
    }
  }
}

// per source u: d_logits[u, i, :] = sum over out-edges e=(u->v) of the pair-softmax backward of the routed gradient
__global__ __launch_bounds__(kThreads) void disc_comm_bwd_kernel(
    const float* __restrict__ d_c, int ld_dc, const float* __restrict__ y0_save, const int32_t* __restrict__ sel, int M,
    const int32_t* __restrict__ t_off, const int32_t* __restrict__ t_dst, const int32_t* __restrict__ t_pos, int N,
    float inv_tau, float* __restrict__ d_logits, int ld_dl) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int u = blockIdx.x * kWavesPerBlock + wave; u < N; u += gridDim.x * kWavesPerBlock) {
    const int t0 = t_off[u], t1 = t_off[u + 1];
    for (int i = lane; i < M; i += kWave) {
      float a0 = 0.f, a1 = 0.f;
      for (int t = t0; t < t1; ++t) {
        const int v = t_dst[t];
        const int p = t_pos[t];
        const int2 sl = *reinterpret_cast<const int2*>(sel + static_cast<size_t>(v) * 2 * M + 2 * i);
        const float g0 = (sl.x == p) ? d_c[static_cast<size_t>(v) * ld_dc + 2 * i] : 0.f;
        const float g1 = (sl.y == p) ? d_c[static_cast<size_t>(v) * ld_dc + 2 * i + 1] : 0.f;
        const float y0 = y0_save[static_cast<size_t>(p) * M + i];
        const float y1 = 1.f - y0;
        const float dot = g0 * y0 + g1 * y1;
        a0 = fmaf(y0 * (g0 - dot), inv_tau, a0);
        a1 = fmaf(y1 * (g1 - dot), inv_tau, a1);
      }
      d_logits[static_cast<size_t>(u) * ld_dl + 2 * i] = a0;
      d_logits[static_cast<size_t>(u) * ld_dl + 2 * i + 1] = a1;
    }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_disc_comm_fwd(const float* logits, int ld, const float* gumbel, const long long* rng, int msg,
                                    const int32_t* talk_off, const int32_t* talk_src, int N, float inv_tau, float* c,
                                    int ld_c, float* y0_save, int32_t* sel, uavgnn_stream_t stream) {
  if (N < 0 || msg < 1 || !logits || !talk_off || !c || !y0_save || !sel || ld < 2 * msg || ld_c < 2 * msg ||
      (!gumbel && !rng))
    return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipLaunchKernelGGL(disc_comm_fwd_kernel, dim3(capped_grid(N, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), logits, ld, gumbel, rng, msg, talk_off, talk_src, N, inv_tau, c, ld_c,
                     y0_save, sel);
  return launch_status();
}

extern "C" int uavgnn_gumbel_noise(const long long* rng, long long E, int msg, float* out, uavgnn_stream_t stream) {
  if (!rng || !out || E < 0 || msg < 1) return UAVGNN_EINVAL;
  if (E == 0) return 0;
  if (reinterpret_cast<uintptr_t>(out) & 7) return UAVGNN_EUNSUPPORTED;
  hipLaunchKernelGGL(gumbel_noise_kernel, dim3(capped_grid(E * msg, 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), rng, E, msg, out);
  return launch_status();
}

extern "C" int uavgnn_disc_comm_bwd(const float* d_c, int ld_dc, const float* y0_save, const int32_t* sel, int msg,
                                    const int32_t* t_off, const int32_t* t_dst, const int32_t* t_pos, int N,
                                    float inv_tau, float* d_logits, int ld_dl, uavgnn_stream_t stream) {
  if (N < 0 || msg < 1 || !d_c || !y0_save || !sel || !t_off || !d_logits) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipLaunchKernelGGL(disc_comm_bwd_kernel, dim3(capped_grid(N, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), d_c, ld_dc, y0_save, sel, msg, t_off, t_dst, t_pos, N, inv_tau,
                     d_logits, ld_dl);
  return launch_status();
}
