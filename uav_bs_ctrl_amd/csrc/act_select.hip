// Epsilon-greedy action selection of the rollout step, one launch.
//
// Replaces the tail of MultiAgentQLearner.act (/root/reference/algos/madrqn/learner.py:73-80): greedy = argmax_a Q,
// one exploration draw PER TEAM (learner.py:75-78), a uniform random action for every agent of an exploring team.  The
// uniforms come from the caller (PyTorch's generator, so runs stay reproducible under torch.manual_seed); argmax, the
// comparison and the select are one pass over Q instead of six tiny launches - at rollout batch sizes each of those
// costs more in dispatch latency than in work.
#include "common.h"

namespace uavgnn {
namespace {

__global__ void eps_greedy_kernel(const float* __restrict__ q, int ld_q, int N, int A, int n_agents,
                                  const float* __restrict__ u_team, const float* __restrict__ u_agent, float eps,
                                  const float* __restrict__ eps_dev, long long* __restrict__ acts) {
  if (eps_dev != nullptr) eps = *eps_dev;   // exploration rate from device memory: a captured graph replays with the current one
  for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < N; a += gridDim.x * blockDim.x) {
    const float* __restrict__ row = q + static_cast<size_t>(a) * ld_q;
    int best = 0;
    float bv = row[0];
    for (int j = 1; j < A; ++j) {
      const float v = row[j];
      if (v > bv) {     // first maximum wins, NaN never wins (torch.argmax would propagate it: Q is finite here)
        bv = v;
        best = j;
      }
    }
    const bool explore = u_team[a / n_agents] <= eps;
    int r = static_cast<int>(u_agent[a] * static_cast<float>(A));
    r = r < A ? r : A - 1;
    acts[a] = explore ? r : best;
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_eps_greedy(const float* q, int ld_q, int N, int A, int n_agents, const float* u_team,
                                 const float* u_agent, float eps, long long* acts, uavgnn_stream_t stream) {
  if (N < 0 || A < 1 || n_agents < 1 || ld_q < A || (N > 0 && (!q || !u_team || !u_agent || !acts))) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipLaunchKernelGGL(eps_greedy_kernel, dim3(capped_grid(N, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), q,
                     ld_q, N, A, n_agents, u_team, u_agent, eps, static_cast<const float*>(nullptr), acts);
  return launch_status();
}

extern "C" int uavgnn_eps_greedy_dev(const float* q, int ld_q, int N, int A, int n_agents, const float* u_team,
                                     const float* u_agent, const float* eps_dev, long long* acts,
                                     uavgnn_stream_t stream) {
  if (N < 0 || A < 1 || n_agents < 1 || ld_q < A || !eps_dev || (N > 0 && (!q || !u_team || !u_agent || !acts)))
    return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipLaunchKernelGGL(eps_greedy_kernel, dim3(capped_grid(N, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), q,
                     ld_q, N, A, n_agents, u_team, u_agent, 0.f, eps_dev, acts);
  return launch_status();
}
