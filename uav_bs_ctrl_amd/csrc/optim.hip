// Tail of MultiAgentQLearner.update in ONE launch over flat parameter buffers (reference:
// /root/reference/algos/madrqn/learner.py:157-166): nn.utils.clip_grad_value_(policy_net.parameters(), 1) on the
// first n_clip elements (the mixer's gradients are not clipped, learner.py:159), torch.optim.AdamW's step (decoupled
// weight decay, bias-corrected moments; learner.py:49,:160) and the polyak update of the target network
// (p_targ <- polyak p_targ + (1 - polyak) p, learner.py:163-166).  Replaces ~15 multi-tensor launches; pure streaming,
// 7 arrays x 4 B per element (2.5 MB of parameters for exp3-TarMAC: latency, not bandwidth).
// lr and the step count are read from DEVICE memory (hyper[0] = lr, hyper[1] = t >= 1 of THIS step) so that a captured
// hipGraph of the update replays with the current values.
#include "common.h"

namespace uavgnn {
namespace {

__global__ __launch_bounds__(256) void adamw_polyak_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ p_targ, long long n, long long n_clip,
                                                           const float* __restrict__ hyper, double beta1_d, double beta2_d,
                                                           float eps, float weight_decay, float clip, float polyak) {
  // The scalars are formed in DOUBLE, as torch.optim.AdamW forms them on the host (Python floats): in fp32,
  // 1 - 0.999f^t is off by ~1.3e-5 relative for the first ~1000 steps (float32(0.999) != 0.999 and the subtraction
  // cancels), which would scale every early step by ~6e-6 - above fp32 round-off.  Once per thread: free.
  const float lr = hyper[0], t = hyper[1];
  const double bc1 = 1.0 - pow(beta1_d, static_cast<double>(t));
  const float sqrt_bc2 = static_cast<float>(sqrt(1.0 - pow(beta2_d, static_cast<double>(t))));
  const float step_size = static_cast<float>(static_cast<double>(lr) / bc1);
  const float decay = static_cast<float>(1.0 - static_cast<double>(lr) * static_cast<double>(weight_decay));
  const float beta2 = static_cast<float>(beta2_d);
  const float w1 = static_cast<float>(1.0 - beta1_d), w2 = static_cast<float>(1.0 - beta2_d);
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x * 4;
  for (long long i = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      float4 g4 = *reinterpret_cast<float4*>(g + i);
      if (clip > 0.f && i + 4 <= n_clip) {
        g4.x = fminf(fmaxf(g4.x, -clip), clip); g4.y = fminf(fmaxf(g4.y, -clip), clip);
        g4.z = fminf(fmaxf(g4.z, -clip), clip); g4.w = fminf(fmaxf(g4.w, -clip), clip);
        *reinterpret_cast<float4*>(g + i) = g4;
      } else if (clip > 0.f && i < n_clip) {
        float* gs = reinterpret_cast<float*>(&g4);
        for (int r = 0; r < 4; ++r)
          if (i + r < n_clip) gs[r] = fminf(fmaxf(gs[r], -clip), clip);
        *reinterpret_cast<float4*>(g + i) = g4;
      }
      float4 p4 = *reinterpret_cast<float4*>(p + i), m4 = *reinterpret_cast<float4*>(m + i),
             v4 = *reinterpret_cast<float4*>(v + i);
      float* ps = reinterpret_cast<float*>(&p4);
      float* ms = reinterpret_cast<float*>(&m4);
      float* vs = reinterpret_cast<float*>(&v4);
      const float* gs = reinterpret_cast<const float*>(&g4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ps[r] *= decay;
        ms[r] = ms[r] + w1 * (gs[r] - ms[r]);                 // exp_avg.lerp_(grad, 1 - beta1)
        vs[r] = beta2 * vs[r] + w2 * gs[r] * gs[r];           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vs[r]) / sqrt_bc2 + eps;
        ps[r] -= step_size * (ms[r] / denom);
      }
      *reinterpret_cast<float4*>(p + i) = p4;
      *reinterpret_cast<float4*>(m + i) = m4;
      *reinterpret_cast<float4*>(v + i) = v4;
      if (p_targ != nullptr) {
        float4 t4 = *reinterpret_cast<float4*>(p_targ + i);
        t4.x = polyak * t4.x + (1.f - polyak) * p4.x; t4.y = polyak * t4.y + (1.f - polyak) * p4.y;
        t4.z = polyak * t4.z + (1.f - polyak) * p4.z; t4.w = polyak * t4.w + (1.f - polyak) * p4.w;
        *reinterpret_cast<float4*>(p_targ + i) = t4;
      }
    } else {
      for (long long q = i; q < n; ++q) {
        float gq = g[q];
        if (clip > 0.f && q < n_clip) {
          gq = fminf(fmaxf(gq, -clip), clip);
          g[q] = gq;
        }
        float pq = p[q] * decay;
        const float mq = m[q] + w1 * (gq - m[q]);
        const float vq = beta2 * v[q] + w2 * gq * gq;
        pq -= step_size * (mq / (sqrtf(vq) / sqrt_bc2 + eps));
        p[q] = pq; m[q] = mq; v[q] = vq;
        if (p_targ != nullptr) p_targ[q] = polyak * p_targ[q] + (1.f - polyak) * pq;
      }
    }
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_adamw_polyak(float* p, float* g, float* exp_avg, float* exp_avg_sq, float* p_targ, long long n,
                                   long long n_clip, const float* hyper, double beta1, double beta2, float eps,
                                   float weight_decay, float clip, float polyak, uavgnn_stream_t stream) {
  if (n < 0 || n_clip < 0 || n_clip > n || !p || !g || !exp_avg || !exp_avg_sq || !hyper) return UAVGNN_EINVAL;
  for (const float* q : {p, g, exp_avg, exp_avg_sq, p_targ})
    if (reinterpret_cast<uintptr_t>(q) & 15) return UAVGNN_EUNSUPPORTED;   // 16-byte accesses
  if (n == 0) return 0;
  const int grid = capped_grid((n + 3) / 4, 256, 1024);
  hipLaunchKernelGGL(adamw_polyak_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, exp_avg,
                     exp_avg_sq, p_targ, n, n_clip, hyper, beta1, beta2, eps, weight_decay, clip, polyak);
  return launch_status();
}
