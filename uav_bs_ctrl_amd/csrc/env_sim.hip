// Batched simulator step of the multi-UBS coverage environment: B independent environments per launch, one wavefront
// each.  Restates, operation by operation and dtype by dtype (NumPy 2 promotion rules, as the
// reference executes here), /root/reference/envs/mubs_cov/mubs_cov.py:
//   step :104-129 (move, clip), _transmit_data :131-210 (distances, collisions, A2G channel gain - envs/common.py:49-59 -,
//   greedy per-GT scheduling onto resource blocks in priority order with the nearest-UBS-first rule and the
//   least-interfered idle RB, SINR rates, running averages, Jain index - envs/common.py:19-25 -, utilities, next
//   priorities), _get_reward :324-341, get_obs :212-242 (padded per-agent observations, column 0 = visibility flag),
//   get_state :278-296.
// The reference runs O(n M) Python loops per environment step (~130 steps/s at 8 x 80 on one core); the scheduling loop
// is inherently sequential per environment (M iterations) but environments are independent, so a wavefront walks its
// environment's GTs in priority order with the UBS / RB dimension on lanes, and everything else is lane <-> GT.
// Per-environment working set (distances, interference weights, RB table) lives in LDS.
//
// Tie rules: np.argsort over the <= 16 UBS distances is NumPy's small-array insertion sort, i.e. stable - reproduced.
// np.argsort over the M average rates (next priorities) is NOT stable in NumPy (and depends on its SIMD dispatch); this
// kernel uses the stable order (ties -> lower GT index first), see tests/test_env_sim.py for how the fixture pins it.
#include "common.h"

namespace uavgnn {
namespace {

struct EnvConsts {
  int n, M, R, A, episode_limit, fair_service, avoid_collision, state_dim;
  double range_pos, r_cov, r_sns, r_comm, dt, h_ubs, p_tx, n0, bw, fc, a, b, eta_los, eta_nlos, safe_dist, penalty,
      rew_scale, max_rate;
};

// NumPy's pairwise float32 summation (numpy/_core/src/umath/loops_utils.h: pairwise_sum) over a contiguous run.
__device__ float np_sum_f32(const float* a, int n) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_sum_f32(a, n2) + np_sum_f32(a + n2, n - n2);
}

__global__ __launch_bounds__(kWave) void env_step_kernel(
    EnvConsts c, int B, const long long* __restrict__ actions, const double* __restrict__ avail_moves,
    double* __restrict__ pos_ubs, const float* __restrict__ pos_gts, int32_t* __restrict__ prior,
    float* __restrict__ avg_rate, int32_t* __restrict__ t_io, float* __restrict__ run_f32, double* __restrict__ n_colls,
    float* __restrict__ d_u2g_out, float* __restrict__ d_u2u_out, int32_t* __restrict__ gt_ubs_out,
    int32_t* __restrict__ gt_rb_out, float* __restrict__ rate_out, double* __restrict__ rate_ubs_out,
    int32_t* __restrict__ coll_out, double* __restrict__ reward_out, float* __restrict__ done_out,
    float* __restrict__ obs_gt, float* __restrict__ obs_ubs, float* __restrict__ obs_agent,
    float* __restrict__ state_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = c.n, M = c.M, R = c.R;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (b >= B) return;
  // ---- LDS carve-up ------------------------------------------------------------------------------------------------
  float* DUG = reinterpret_cast<float*>(smem);            // [n][M] UBS-GT distances (float32, as the reference stores them)
  float* W = DUG + n * M;                                  // [n][M] float32(p_tx g [d <= r_cov]): interference weights
  // the float64 arrays come first among the odd-sized ones: G starts at byte 8 n M, PU right behind it - both 8-byte
  // aligned for every (n, M), odd M included (behind the three [M] float arrays PU would sit at 4 mod 8)
  double* G = reinterpret_cast<double*>(W + n * M);        // [n][M] p_tx * g (float64 numerator of the SINR)
  double* PU = G + n * M;                                  // [n][2]
  float* AVG = reinterpret_cast<float*>(PU + 2 * n);       // [M]
  float* RATE = AVG + M;                                   // [M]
  float* SCR = RATE + M;                                   // [M] scratch (clipped averages, squares)
  float* ITF = SCR + M;                                    // [16]
  int* PR = reinterpret_cast<int*>(ITF + 16);              // [M] priorities in use
  int* GU = PR + M;                                        // [M] serving UBS per GT (-1)
  int* GR = GU + M;                                        // [M] RB per GT (-1)
  int* ASG = GR + M;                                       // [n][R] GT on RB c of UBS i (-1 = idle)
  int* CNT = ASG + n * R;                                  // [n]
  int* COL = CNT + n;                                      // [n]
  float* DUU = reinterpret_cast<float*>(COL + n);          // [n][n]

  int t = t_io[b];
  const bool moving = actions != nullptr;
  if (moving) t += 1;                                                                       // mubs_cov.py:105
  // ---- positions (mubs_cov.py:108-109) -------------------------------------------------------------------------------
  if (lane < n) {
    double x = pos_ubs[(static_cast<size_t>(b) * n + lane) * 2 + 0], y = pos_ubs[(static_cast<size_t>(b) * n + lane) * 2 + 1];
    if (moving) {
      long long a = actions[static_cast<size_t>(b) * n + lane];
      a = a < 0 ? 0 : (a >= c.A ? c.A - 1 : a);
      x = fmin(fmax(x + avail_moves[2 * a + 0], 0.0), c.range_pos);
      y = fmin(fmax(y + avail_moves[2 * a + 1], 0.0), c.range_pos);
      pos_ubs[(static_cast<size_t>(b) * n + lane) * 2 + 0] = x;
      pos_ubs[(static_cast<size_t>(b) * n + lane) * 2 + 1] = y;
    }
    PU[2 * lane] = x;
    PU[2 * lane + 1] = y;
  }
  for (int m = lane; m < M; m += kWave) {
    PR[m] = prior[static_cast<size_t>(b) * M + m];
    AVG[m] = avg_rate[static_cast<size_t>(b) * M + m];
    GU[m] = -1;
    GR[m] = -1;
  }
  for (int i = lane; i < n * R; i += kWave) ASG[i] = -1;
  if (lane < n) CNT[lane] = 0;
  __syncthreads();
  const float* __restrict__ pg = pos_gts + static_cast<size_t>(b) * M * 2;
  const double k_los = pow(10.0, c.eta_los / 20.0), k_nlos = pow(10.0, c.eta_nlos / 20.0);   // loop-invariant excess-loss factors
  // ---- step 1: distances (:135-141), float64 norm of (float32 GT position - float64 UBS position), stored float32 ----
  for (int k = lane; k < n * M; k += kWave) {
    const int i = k / M, m = k - i * M;
    const double dx = static_cast<double>(pg[2 * m]) - PU[2 * i], dy = static_cast<double>(pg[2 * m + 1]) - PU[2 * i + 1];
    const float d = static_cast<float>(sqrt(dx * dx + dy * dy));
    DUG[k] = d;
    d_u2g_out[static_cast<size_t>(b) * n * M + k] = d;
    // A2G channel gain (envs/common.py:49-59): arctan / exp in float32, path loss in float64 (np.square(h) is a float64)
    const float ang = atanf(static_cast<float>(c.h_ubs) / (d + 1e-5f));
    const float p_los = 1.f / (1.f + static_cast<float>(c.a) * expf(-static_cast<float>(c.b) * (ang - static_cast<float>(c.a))));
    const double dd = sqrt(static_cast<double>(d * d) + c.h_ubs * c.h_ubs);
    const double q = 4.0 * 3.141592653589793 * c.fc * dd / 3e8;
    const double fspl = q * q;
    const double pl = static_cast<double>(p_los) * fspl * k_los + static_cast<double>(1.f - p_los) * fspl * k_nlos;
    const double ptxg = c.p_tx * (1.0 / pl);
    G[k] = ptxg;
    W[k] = d <= static_cast<float>(c.r_cov) ? static_cast<float>(ptxg) : 0.f;     // p_itf rows are float32 (:169,:187)
  }
  for (int k = lane; k < n * n; k += kWave) {
    const int i = k / n, jx = k - i * n;
    const double dx = PU[2 * jx] - PU[2 * i], dy = PU[2 * jx + 1] - PU[2 * i + 1];
    const float d = static_cast<float>(sqrt(dx * dx + dy * dy));
    DUU[k] = d;
    d_u2u_out[static_cast<size_t>(b) * n * n + k] = d;
  }
  __syncthreads();
  if (lane < n) {                                                                           // collisions (:144)
    int col = 0;
    for (int jx = 0; jx < n; ++jx)
      col |= (static_cast<double>(DUU[lane * n + jx]) + (jx == lane ? 99999.0 : 0.0)) < c.safe_dist;
    COL[lane] = col;
    coll_out[static_cast<size_t>(b) * n + lane] = col;
  }
  __syncthreads();
  // ---- step 2: greedy scheduling in priority order (:171-189) ---------------------------------------------------------
  for (int idx = 0; idx < M; ++idx) {
    const int m = PR[idx];
    const float di = lane < n ? DUG[lane * M + m] : INFINITY;
    int rank = 0;                                            // stable ascending order of the n distances
    for (int jx = 0; jx < n; ++jx) {
      const float dj = DUG[jx * M + m];
      rank += (dj < di) || (dj == di && jx < lane);
    }
    int chosen = -1;
    for (int r = 0; r < n; ++r) {
      const unsigned long long bal = __ballot(lane < n && rank == r);
      const int i = __ffsll(static_cast<long long>(bal)) - 1;
      const float dcand = DUG[i * M + m];
      if (!(dcand <= static_cast<float>(c.r_cov))) break;    // nearest first: nobody farther can cover either
      if (CNT[i] < R) {
        chosen = i;
        break;
      }
    }
    if (chosen >= 0) {
      if (lane < R) {                                        // interference seen by GT m on RB `lane` (float32, UBS order)
        float itf = 0.f;
        for (int i2 = 0; i2 < n; ++i2) itf += (ASG[i2 * R + lane] >= 0) ? W[i2 * M + m] : 0.f;
        ITF[lane] = (ASG[chosen * R + lane] >= 0) ? NAN : itf;   // occupied RBs of the serving UBS are excluded (:181-183)
      }
      __syncthreads();
      if (lane == 0) {
        int best = -1;
        float bv = 0.f;
        for (int cc = 0; cc < R; ++cc) {                     // np.nanargmin: first minimum among the non-NaN entries
          const float v = ITF[cc];
          if (v == v && (best < 0 || v < bv)) {
            best = cc;
            bv = v;
          }
        }
        ASG[chosen * R + best] = m;
        CNT[chosen] += 1;
        GU[m] = chosen;
        GR[m] = best;
      }
      __syncthreads();
    }
  }
  // ---- rates (:192-198): float64 SINR over the float32 interference sum (NumPy pairwise order), stored float32 --------
  for (int m = lane; m < M; m += kWave) {
    float rate = 0.f;
    const int i = GU[m];
    if (i >= 0) {
      const int cc = GR[m];
      float col[16];
      for (int i2 = 0; i2 < n; ++i2) {
        const int a2 = ASG[i2 * R + cc];
        col[i2] = (a2 >= 0 && a2 != m) ? W[i2 * M + m] : 0.f;
      }
      const float itf = np_sum_f32(col, n);
      const double sinr = G[i * M + m] / (static_cast<double>(itf) + c.bw * c.n0);
      rate = static_cast<float>(c.bw * log2(1.0 + sinr) * 1e-6);
    }
    RATE[m] = rate;
    rate_out[static_cast<size_t>(b) * M + m] = rate;
    gt_ubs_out[static_cast<size_t>(b) * M + m] = i;
    gt_rb_out[static_cast<size_t>(b) * M + m] = i >= 0 ? GR[m] : -1;
    // step 3 (:203): average data rate, float32
    const float av = (AVG[m] * static_cast<float>(t) + rate) / static_cast<float>(t + 1);
    AVG[m] = av;
    avg_rate[static_cast<size_t>(b) * M + m] = av;
    SCR[m] = fmaxf(av, 1e-6f);                               // np.clip(x, 1e-6, inf) for the Jain index
  }
  __syncthreads();
  double rpu = 0.0;
  if (lane < n) {                                            // rate offered by each UBS (:199), float64
    for (int m = 0; m < M; ++m)
      if (GU[m] == lane) rpu += static_cast<double>(RATE[m]);
    rate_ubs_out[static_cast<size_t>(b) * n + lane] = rpu;
  }
  // ---- scalars by one lane, in NumPy's evaluation order (:204-208, envs/common.py:19-25) -------------------------------
  __shared__ float s_gu;
  if (lane == 0) {
    float* rf = run_f32 + static_cast<size_t>(b) * 4;        // {total_throughput, avg_global_util, fair_idx, global_util}
    const float rsum = np_sum_f32(RATE, M);
    rf[0] = rf[0] + rsum * static_cast<float>(c.dt) / 1e3f;
    const float xs = np_sum_f32(SCR, M);
    for (int m = 0; m < M; ++m) SCR[m] = SCR[m] * SCR[m];
    const float fair = (xs * xs) / (static_cast<float>(M) * np_sum_f32(SCR, M));
    const float gu = fair * (rsum / static_cast<float>(M));
    rf[1] = (rf[1] * static_cast<float>(t) + gu) / static_cast<float>(t + 1);
    rf[2] = fair;
    rf[3] = gu;
    s_gu = gu;
    int ncol = 0;
    for (int i = 0; i < n; ++i) ncol += COL[i];
    n_colls[b] += static_cast<double>(ncol) / 2.0;           // :145
    t_io[b] = t;
    done_out[b] = (t == c.episode_limit) ? 1.f : 0.f;        // :343-345
  }
  __syncthreads();
  // next priorities (:209): STABLE ascending order of the averages
  for (int m = lane; m < M; m += kWave) {
    const float am = AVG[m];
    int rank = 0;
    for (int m2 = 0; m2 < M; ++m2) {
      const float a2 = AVG[m2];
      rank += (a2 < am) || (a2 == am && m2 < m);
    }
    prior[static_cast<size_t>(b) * M + rank] = m;
  }
  // ---- reward (:324-341), float64 from the float32 utility -----------------------------------------------------------
  if (lane < n) {
    float base;
    if (c.fair_service) {
      base = s_gu;
    } else {
      base = np_sum_f32(RATE, M) / static_cast<float>(M);
    }
    double r = static_cast<double>(static_cast<float>(c.rew_scale) * base) / c.max_rate;
    r = r * (rpu == 0.0 ? 0.0 : 1.0);
    if (c.avoid_collision) r = (1 - COL[lane]) * r - COL[lane] * c.penalty;
    reward_out[static_cast<size_t>(b) * n + lane] = r;
  }
  // ---- observations (:215-242) and global state (:278-296) -----------------------------------------------------------
  const int Sg = c.fair_service ? 5 : 4;
  const double norm_s = fmin(c.range_pos, c.r_sns), norm_c = fmin(c.range_pos, c.r_comm);
  for (int k = lane; k < n * M; k += kWave) {
    const int i = k / M, m = k - i * M;
    float* o = obs_gt + (static_cast<size_t>(b) * n * M + k) * Sg;
    const bool vis = DUG[k] <= static_cast<float>(c.r_sns);
    o[0] = vis ? 1.f : 0.f;
    o[1] = vis ? static_cast<float>((static_cast<double>(pg[2 * m]) - PU[2 * i]) / norm_s) : 0.f;
    o[2] = vis ? static_cast<float>((static_cast<double>(pg[2 * m + 1]) - PU[2 * i + 1]) / norm_s) : 0.f;
    o[3] = vis ? static_cast<float>(static_cast<double>(RATE[m]) / c.max_rate) : 0.f;
    if (c.fair_service)
      o[4] = vis ? static_cast<float>(static_cast<double>(AVG[m]) / c.max_rate * M / (n * R)) : 0.f;
  }
  for (int k = lane; k < n * (n - 1); k += kWave) {
    const int i = k / (n - 1), jj = k - i * (n - 1);
    const int other = jj < i ? jj : jj + 1;
    float* o = obs_ubs + (static_cast<size_t>(b) * n * (n - 1) + k) * 3;
    const bool vis = DUU[i * n + other] <= static_cast<float>(c.r_comm);
    o[0] = vis ? 1.f : 0.f;
    o[1] = vis ? static_cast<float>((PU[2 * other] - PU[2 * i]) / norm_c) : 0.f;
    o[2] = vis ? static_cast<float>((PU[2 * other + 1] - PU[2 * i + 1]) / norm_c) : 0.f;
  }
  if (lane < n) {
    obs_agent[(static_cast<size_t>(b) * n + lane) * 2 + 0] = static_cast<float>(PU[2 * lane] / c.range_pos);
    obs_agent[(static_cast<size_t>(b) * n + lane) * 2 + 1] = static_cast<float>(PU[2 * lane + 1] / c.range_pos);
  }
  if (state_out != nullptr) {
    float* st = state_out + static_cast<size_t>(b) * c.state_dim;
    const int Ss = c.fair_service ? 4 : 3;
    if (lane < n) {
      st[2 * lane] = static_cast<float>(PU[2 * lane] / c.range_pos);
      st[2 * lane + 1] = static_cast<float>(PU[2 * lane + 1] / c.range_pos);
    }
    for (int m = lane; m < M; m += kWave) {
      float* g = st + 2 * n + m * Ss;
      g[0] = pg[2 * m] / static_cast<float>(c.range_pos);
      g[1] = pg[2 * m + 1] / static_cast<float>(c.range_pos);
      g[2] = static_cast<float>(static_cast<double>(RATE[m]) / c.max_rate);
      if (c.fair_service) g[3] = static_cast<float>(static_cast<double>(AVG[m]) / c.max_rate * M / (n * R));
    }
  }
}

inline size_t env_lds_bytes(int n, int M, int R) {
  size_t f = static_cast<size_t>(n) * M * 2 + 3 * static_cast<size_t>(M) + 16 + static_cast<size_t>(n) * n;   // floats
  size_t d = static_cast<size_t>(n) * M + 2 * static_cast<size_t>(n);                                         // doubles
  size_t i = 3 * static_cast<size_t>(M) + static_cast<size_t>(n) * R + 2 * static_cast<size_t>(n);            // ints
  return f * 4 + d * 8 + i * 4 + 64;
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_env_state_dim(int n_ubs, int n_gts, int fair_service) {
  return 2 * n_ubs + n_gts * (fair_service ? 4 : 3);
}

// int_consts: {n_ubs, n_gts, n_rbs, n_actions, episode_limit, fair_service, avoid_collision}
// f64_consts: {range_pos, r_cov, r_sns, r_comm, dt, h_ubs, p_tx, n0, bw, fc, a, b, eta_los, eta_nlos, safe_dist, penalty,
//              reward_scale_rate, max_rate}
extern "C" int uavgnn_env_step(const int32_t* int_consts, const double* f64_consts, int B, const long long* actions,
                               const double* avail_moves, double* pos_ubs, const float* pos_gts, int32_t* prior,
                               float* avg_rate, int32_t* t, float* run_f32, double* n_colls, float* d_u2g, float* d_u2u,
                               int32_t* gt_ubs, int32_t* gt_rb, float* rate_per_gt, double* rate_per_ubs,
                               int32_t* mask_collision, double* reward, float* done, float* obs_gt, float* obs_ubs,
                               float* obs_agent, float* state, uavgnn_stream_t stream) {
  if (!int_consts || !f64_consts || B < 0 || !pos_ubs || !pos_gts || !prior || !avg_rate || !t || !run_f32 || !n_colls ||
      !d_u2g || !d_u2u || !gt_ubs || !gt_rb || !rate_per_gt || !rate_per_ubs || !mask_collision || !reward || !done ||
      !obs_gt || !obs_ubs || !obs_agent || (actions && !avail_moves))
    return UAVGNN_EINVAL;
  EnvConsts c;
  c.n = int_consts[0]; c.M = int_consts[1]; c.R = int_consts[2]; c.A = int_consts[3]; c.episode_limit = int_consts[4];
  c.fair_service = int_consts[5]; c.avoid_collision = int_consts[6];
  c.state_dim = uavgnn_env_state_dim(c.n, c.M, c.fair_service);
  c.range_pos = f64_consts[0]; c.r_cov = f64_consts[1]; c.r_sns = f64_consts[2]; c.r_comm = f64_consts[3];
  c.dt = f64_consts[4]; c.h_ubs = f64_consts[5]; c.p_tx = f64_consts[6]; c.n0 = f64_consts[7]; c.bw = f64_consts[8];
  c.fc = f64_consts[9]; c.a = f64_consts[10]; c.b = f64_consts[11]; c.eta_los = f64_consts[12];
  c.eta_nlos = f64_consts[13]; c.safe_dist = f64_consts[14]; c.penalty = f64_consts[15]; c.rew_scale = f64_consts[16];
  c.max_rate = f64_consts[17];
  if (c.n < 1 || c.n > 16 || c.M < 1 || c.M > 1024 || c.R < 1 || c.R > 16 || c.A < 1) return UAVGNN_EUNSUPPORTED;
  const size_t lds = env_lds_bytes(c.n, c.M, c.R);
  if (lds > 64 * 1024) return UAVGNN_EUNSUPPORTED;
  if (B == 0) return 0;
  hipLaunchKernelGGL(env_step_kernel, dim3(B), dim3(kWave), lds, static_cast<hipStream_t>(stream), c, B, actions,
                     avail_moves, pos_ubs, pos_gts, prior, avg_rate, t, run_f32, n_colls, d_u2g, d_u2u, gt_ubs, gt_rb,
                     rate_per_gt, rate_per_ubs, mask_collision, reward, done, obs_gt, obs_ubs, obs_agent, state);
  return launch_status();
}
