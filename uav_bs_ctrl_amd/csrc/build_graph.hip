// f1: device-side observation -> graph construction.
//
// Replaces the per-agent Python loops of /root/reference/algos/madrqn/utils/env_wrappers.py:65-89 (build_obs_graph: keep
// the rows whose visibility flag is 1, drop the flag column), :139-154 (build_comm_graph: edge i->j iff
// d_u2u[i,j] <= r_comm, self loops included) and dgl.batch / dgl.merge (:67,:137) for a whole batch of environments:
// padded observation tensors  gt [B,n,M,1+Fg]  ubs [B,n,U,1+Fu]  (column 0 = flag, mubs_cov.py:215-242) and the
// pairwise UBS distances d_u2u [B,n,n] go in, the segment layout of HeteroBatch comes out, without leaving the GPU.
// Pass 1 counts (one wavefront per agent row, ballot + popcount); the caller turns counts into offsets with a prefix
// sum; pass 2 compacts (rank of a kept row = popcount of lower lanes' ballot bits), so the order of the kept rows is
// the reference's (ascending m).  Pure byte/index work: results are bit-identical to the host builder.
#include "common.h"

namespace uavgnn {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;

__device__ __forceinline__ int lanes_below(unsigned long long mask, int lane) {
  return __popcll(mask & ((1ull << lane) - 1ull));
}

// deg_seen[a], deg_near[a] for agent row a = b*n + i
__global__ __launch_bounds__(kThreads) void obs_degrees_kernel(const float* __restrict__ gt, int M, int Sg,
                                                               const float* __restrict__ ubs, int U, int Su, int N,
                                                               int32_t* __restrict__ deg_seen,
                                                               int32_t* __restrict__ deg_near) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int a = blockIdx.x * kWavesPerBlock + wave; a < N; a += gridDim.x * kWavesPerBlock) {
    int cs = 0, cn = 0;
    for (int m0 = 0; m0 < M; m0 += kWave) {
      const int m = m0 + lane;
      const bool keep = m < M && gt[(static_cast<size_t>(a) * M + m) * Sg] == 1.f;
      cs += __popcll(__ballot(keep));
    }
    for (int m0 = 0; m0 < U; m0 += kWave) {
      const int m = m0 + lane;
      const bool keep = m < U && ubs[(static_cast<size_t>(a) * U + m) * Su] == 1.f;
      cn += __popcll(__ballot(keep));
    }
    if (lane == 0) {
      deg_seen[a] = cs;
      deg_near[a] = cn;
    }
  }
}

template <int F>
__device__ __forceinline__ void compact_rows(const float* __restrict__ src, int rows, int S, size_t a, int lane,
                                             float* __restrict__ dst, int base) {
  for (int m0 = 0; m0 < rows; m0 += kWave) {
    const int m = m0 + lane;
    const float* r = src + (a * rows + m) * S;
    const bool keep = m < rows && r[0] == 1.f;
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      float* o = dst + static_cast<size_t>(base + lanes_below(mask, lane)) * F;
#pragma unroll
      for (int f = 0; f < F; ++f) o[f] = r[1 + f];
    }
    base += __popcll(mask);
  }
}

__global__ __launch_bounds__(kThreads) void obs_compact_kernel(const float* __restrict__ gt, int M, int Fg,
                                                               const float* __restrict__ ubs, int U, int Fu, int N,
                                                               const int32_t* __restrict__ seen_off,
                                                               const int32_t* __restrict__ near_off,
                                                               float* __restrict__ x_gt, float* __restrict__ x_ubs) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int a = blockIdx.x * kWavesPerBlock + wave; a < N; a += gridDim.x * kWavesPerBlock) {
    if (Fg == 4) compact_rows<4>(gt, M, 5, a, lane, x_gt, seen_off[a]);
    if (Fu == 2) compact_rows<2>(ubs, U, 3, a, lane, x_ubs, near_off[a]);
  }
}

// talk relation: one wavefront per environment, lane <-> destination j (n <= 64).
// deg_in[b*n+j] = #{i : d[b,i,j] <= r};  row_cnt[b] = #edges of env b (for the reference edge ids)
__global__ __launch_bounds__(kThreads) void talk_degrees_kernel(const float* __restrict__ d_u2u, int n, int B, float r,
                                                                int32_t* __restrict__ deg_in,
                                                                int32_t* __restrict__ env_edges) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int b = blockIdx.x * kWavesPerBlock + wave; b < B; b += gridDim.x * kWavesPerBlock) {
    const float* d = d_u2u + static_cast<size_t>(b) * n * n;
    int cnt = 0, tot = 0;
    for (int i = 0; i < n; ++i) {
      const bool e = lane < n && d[i * n + lane] <= r;
      cnt += e ? 1 : 0;
      tot += __popcll(__ballot(e));
    }
    if (lane < n) deg_in[b * n + lane] = cnt;
    if (lane == 0) env_edges[b] = tot;
  }
}

__global__ __launch_bounds__(kThreads) void talk_compact_kernel(const float* __restrict__ d_u2u, int n, int B, float r,
                                                                const int32_t* __restrict__ talk_off,
                                                                const int32_t* __restrict__ env_base,
                                                                int32_t* __restrict__ talk_src,
                                                                int32_t* __restrict__ talk_eid) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int b = blockIdx.x * kWavesPerBlock + wave; b < B; b += gridDim.x * kWavesPerBlock) {
    const float* d = d_u2u + static_cast<size_t>(b) * n * n;
    int pos = lane < n ? talk_off[b * n + lane] : 0;   // next free CSC slot of destination j = lane
    int eid = env_base[b];                             // reference edge ids run i-major inside an environment
    for (int i = 0; i < n; ++i) {
      const bool e = lane < n && d[i * n + lane] <= r;
      const unsigned long long mask = __ballot(e);
      if (e) {
        talk_src[pos] = b * n + i;
        talk_eid[pos] = eid + lanes_below(mask, lane);
        ++pos;
      }
      eid += __popcll(mask);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The whole construction in ONE launch for small batches (N <= 4096 agents: a rollout step of a few environments, the
// per-step graphs of a 32-sequence update): one workgroup counts (wave per agent / per environment), scans the three
// degree arrays in LDS and compacts.  At these sizes the multi-launch path above is 17 launches of ~3 us each around a
// few hundred bytes of work (what bounds `act` at the reference's own operating point).  Results are
// bit-identical to the multi-launch path; edge arrays are written at their true length (the caller allocates capacity).
constexpr int kSmallThreads = 1024;
constexpr int kSmallMaxAgents = 4096;

// exclusive scan of v[0..n) in LDS by the whole workgroup (n <= 4 * blockDim), total returned to every thread
__device__ int block_exclusive_scan(int* __restrict__ v, int n, int* __restrict__ wave_tot) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  int x[4], s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = tid * 4 + q;
    x[q] = i < n ? v[i] : 0;
    s += x[q];
  }
  int incl = s;                                 // inclusive scan of the per-thread sums inside the wave
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == kWave - 1) wave_tot[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < (kSmallThreads >> 6); ++w) {
    const int t = wave_tot[w];
    if (w < wave) base += t;
    total += t;
  }
  int run = base + incl - s;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = tid * 4 + q;
    if (i < n) v[i] = run;
    run += x[q];
  }
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(kSmallThreads) void build_graph_small_kernel(
    const float* __restrict__ gt, int M, const float* __restrict__ ubs, int U, const float* __restrict__ d_u2u, int n, int B,
    float r, int32_t* __restrict__ seen_off, int32_t* __restrict__ near_off, int32_t* __restrict__ talk_off,
    float* __restrict__ x_gt, float* __restrict__ x_ubs, int32_t* __restrict__ talk_src, int32_t* __restrict__ talk_eid,
    int32_t* __restrict__ graph_off) {
  __shared__ int sS[kSmallMaxAgents], sN[kSmallMaxAgents], sT[kSmallMaxAgents];
  __shared__ int sWave[kSmallThreads >> 6];
  const int N = B * n;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int W = kSmallThreads >> 6;
  for (int a = wave; a < N; a += W) {                      // pass 1: degrees (as obs_degrees_kernel)
    int cs = 0, cn = 0;
    for (int m0 = 0; m0 < M; m0 += kWave) {
      const int m = m0 + lane;
      cs += __popcll(__ballot(m < M && gt[(static_cast<size_t>(a) * M + m) * 5] == 1.f));
    }
    for (int m0 = 0; m0 < U; m0 += kWave) {
      const int m = m0 + lane;
      cn += __popcll(__ballot(m < U && ubs[(static_cast<size_t>(a) * U + m) * 3] == 1.f));
    }
    if (lane == 0) {
      sS[a] = cs;
      sN[a] = cn;
    }
  }
  if (d_u2u != nullptr) {
    for (int b = wave; b < B; b += W) {                    // as talk_degrees_kernel (n <= 64)
      const float* d = d_u2u + static_cast<size_t>(b) * n * n;
      int cnt = 0;
      for (int i = 0; i < n; ++i) cnt += (lane < n && d[i * n + lane] <= r) ? 1 : 0;
      if (lane < n) sT[b * n + lane] = cnt;
    }
  }
  __syncthreads();
  const int Es = block_exclusive_scan(sS, N, sWave);
  const int En = block_exclusive_scan(sN, N, sWave);
  const int Et = d_u2u != nullptr ? block_exclusive_scan(sT, N, sWave) : 0;
  for (int a = tid; a < N; a += kSmallThreads) {
    seen_off[a] = sS[a];
    near_off[a] = sN[a];
    if (talk_off != nullptr) talk_off[a] = d_u2u != nullptr ? sT[a] : 0;
  }
  if (tid == 0) {
    seen_off[N] = Es;
    near_off[N] = En;
    if (talk_off != nullptr) talk_off[N] = Et;
  }
  for (int b = tid; b <= B; b += kSmallThreads) graph_off[b] = b * n;
  for (int a = wave; a < N; a += W) {                      // pass 2: compaction (as obs_compact_kernel)
    compact_rows<4>(gt, M, 5, a, lane, x_gt, sS[a]);
    compact_rows<2>(ubs, U, 3, a, lane, x_ubs, sN[a]);
  }
  if (d_u2u != nullptr) {
    for (int b = wave; b < B; b += W) {                    // as talk_compact_kernel; env_base[b] = talk_off[b * n]
      const float* d = d_u2u + static_cast<size_t>(b) * n * n;
      int pos = lane < n ? sT[b * n + lane] : 0;
      int eid = sT[b * n];
      for (int i = 0; i < n; ++i) {
        const bool e = lane < n && d[i * n + lane] <= r;
        const unsigned long long mask = __ballot(e);
        if (e) {
          talk_src[pos] = b * n + i;
          talk_eid[pos] = eid + lanes_below(mask, lane);
          ++pos;
        }
        eid += __popcll(mask);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Derived indexes of a batch: the K1 hand-out order and the transpose of the `talk` CSC.  The reference pays the
// equivalent (DGL materialises CSR/CSC formats lazily inside the first message-passing call of every new graph); here
// they are integer passes of a few microseconds, deterministic (no float atomics, integer atomics only where the
// result does not depend on their order).

constexpr int kOrderBins = 256;      // degree buckets: bin = 255 - min(deg, 255)  (ascending bin = descending degree)
constexpr int kOrderThreads = 256;
constexpr int kOrderMaxBlocks = 256;
constexpr int kScanThreads = 1024;
constexpr int kScanPer = 4;
constexpr int kScanTile = kScanThreads * kScanPer;   // elements per block of the scan

__device__ __forceinline__ int order_bin(const int32_t* __restrict__ seg_off, int v) {
  const int deg = seg_off[v + 1] - seg_off[v];
  return kOrderBins - 1 - (deg < kOrderBins - 1 ? deg : kOrderBins - 1);
}

// hist[bin * G + b] = destinations of block b's contiguous chunk that fall into `bin`
__global__ __launch_bounds__(kOrderThreads) void order_hist_kernel(const int32_t* __restrict__ seg_off, int N, int chunk,
                                                                   int32_t* __restrict__ hist) {
  __shared__ int h[kOrderBins];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int lo = blockIdx.x * chunk, hi = min(N, lo + chunk);
  for (int v = lo + threadIdx.x; v < hi; v += kOrderThreads) atomicAdd(&h[order_bin(seg_off, v)], 1);
  __syncthreads();
  hist[threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter: block b walks its chunk in index order; the slot of destination v is
// (prefix of its (bin, block) cell) + (destinations of the same bin earlier in the chunk).
__global__ __launch_bounds__(kOrderThreads) void order_scatter_kernel(const int32_t* __restrict__ seg_off, int N,
                                                                      int chunk, const int32_t* __restrict__ cell_incl,
                                                                      int32_t* __restrict__ order) {
  __shared__ int base[kOrderBins];
  __shared__ int bins[kOrderThreads];
  const int t = threadIdx.x;
  const int cell = t * gridDim.x + blockIdx.x;   // cell_incl = inclusive scan over (bin-major, block-minor) cells
  base[t] = cell > 0 ? cell_incl[cell - 1] : 0;
  const int lo = blockIdx.x * chunk, hi = min(N, lo + chunk);
  for (int tile = lo; tile < hi; tile += kOrderThreads) {
    const int v = tile + t;
    const int bin = v < hi ? order_bin(seg_off, v) : -1;
    bins[t] = bin;
    __syncthreads();
    int rank = 0;
    if (bin >= 0)
      for (int j = 0; j < t; ++j) rank += bins[j] == bin;
    const int pos = bin >= 0 ? base[bin] + rank : 0;
    __syncthreads();
    if (bin >= 0) {
      order[pos] = v;
      atomicAdd(&base[bin], 1);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int wave_scan_inclusive(int v, int lane) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  return v;
}

// Inclusive scan of one tile of kScanTile elements, in place: coalesced load into LDS, every thread scans its
// kScanPer consecutive elements, wave + block prefix, coalesced store.  sums (optional) receives the tile total.
__global__ __launch_bounds__(kScanThreads) void scan_tile_kernel(int32_t* __restrict__ data, int n,
                                                                 int32_t* __restrict__ sums) {
  __shared__ int buf[kScanTile];
  __shared__ int wave_tot[kScanThreads / kWave];
  const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
  const int lo = blockIdx.x * kScanTile;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    const int i = lo + k * kScanThreads + t;
    buf[k * kScanThreads + t] = i < n ? data[i] : 0;
  }
  __syncthreads();
  int v[kScanPer];
  int s = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    s += buf[t * kScanPer + k];
    v[k] = s;
  }
  const int inc = wave_scan_inclusive(s, lane);
  if (lane == kWave - 1) wave_tot[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const int w = lane < kScanThreads / kWave ? wave_tot[lane] : 0;
    const int ws = wave_scan_inclusive(w, lane);
    if (lane < kScanThreads / kWave) wave_tot[lane] = ws - w;   // exclusive prefix of the waves
    if (sums != nullptr && lane == kScanThreads / kWave - 1) sums[blockIdx.x] = ws;
  }
  __syncthreads();
  const int base = wave_tot[wave] + inc - s;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) buf[t * kScanPer + k] = base + v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    const int i = lo + k * kScanThreads + t;
    if (i < n) data[i] = buf[k * kScanThreads + t];
  }
}

// Adds the carry of the preceding tiles to tile blockIdx.x + 1.  scanned = false: sums holds raw tile totals and the
// block reduces sums[0 .. b) itself (b <= kScanThreads; saves the launch that would scan them); scanned = true: sums
// holds their inclusive scan.
__global__ __launch_bounds__(kScanThreads) void scan_carry_kernel(int32_t* __restrict__ data, int n,
                                                                  const int32_t* __restrict__ sums, bool scanned) {
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & (kWave - 1);
  const int b = blockIdx.x + 1;   // tile 0 needs no carry
  if (t == 0) carry_s = scanned ? sums[b - 1] : 0;
  __syncthreads();
  if (!scanned) {
    int c = t < b ? sums[t] : 0;
    c = wave_scan_inclusive(c, lane);
    if (lane == kWave - 1 && c != 0) atomicAdd(&carry_s, c);   // integer: order-independent
    __syncthreads();
  }
  const int carry = carry_s;
  const int lo = b * kScanTile;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    const int i = lo + k * kScanThreads + t;
    if (i < n) data[i] += carry;
  }
}

// ints of scratch scan_inclusive_i32 needs for n elements (tile totals of every level)
inline size_t scan_scratch_elems(long long n) {
  size_t tot = 0;
  while (n > kScanTile) {
    n = (n + kScanTile - 1) / kScanTile;
    tot += static_cast<size_t>(n);
  }
  return tot + 1;
}

// data[0..n) <- inclusive scan, in place.
int scan_inclusive_i32(int32_t* data, int n, int32_t* scratch, hipStream_t st) {
  if (n <= 0) return 0;
  const int tiles = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(scan_tile_kernel, dim3(tiles), dim3(kScanThreads), 0, st, data, n,
                     tiles > 1 ? scratch : static_cast<int32_t*>(nullptr));
  if (tiles > 1) {
    const bool scanned = tiles > kScanThreads;
    if (scanned) {
      const int rc = scan_inclusive_i32(scratch, tiles, scratch + tiles, st);
      if (rc != 0) return rc;
    }
    hipLaunchKernelGGL(scan_carry_kernel, dim3(tiles - 1), dim3(kScanThreads), 0, st, data, n, scratch, scanned);
  }
  return launch_status();
}

// t_off[1 + u] += 1 for every edge leaving u (integer atomics: the counts do not depend on their order)
__global__ void csc_out_degrees_kernel(const int32_t* __restrict__ talk_src, int E, int32_t* __restrict__ t_off) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x)
    atomicAdd(&t_off[1 + talk_src[e]], 1);
}

// Every in-edge (position e in the CSC, destination d) takes a free slot of its source's out-list (8 edges in flight
// per lane so that the returning atomics overlap) ...
__global__ void csc_transpose_fill_kernel(const int32_t* __restrict__ talk_off, const int32_t* __restrict__ talk_src,
                                          int N, const int32_t* __restrict__ t_off, int32_t* __restrict__ cursor,
                                          int32_t* __restrict__ t_dst, int32_t* __restrict__ t_pos) {
  for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < N; d += gridDim.x * blockDim.x) {
    const int e1 = talk_off[d + 1];
    for (int e0 = talk_off[d]; e0 < e1; e0 += 8) {
      int u[8], k[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) u[i] = e0 + i < e1 ? talk_src[e0 + i] : -1;
#pragma unroll
      for (int i = 0; i < 8; ++i) k[i] = u[i] >= 0 ? t_off[u[i]] + atomicAdd(&cursor[u[i]], 1) : 0;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (u[i] >= 0) {
          t_pos[k[i]] = e0 + i;
          t_dst[k[i]] = d;
        }
    }
  }
}

// ... and every source then orders its out-list by CSC position (= by destination), which makes the result
// independent of the slot race above.  Out-lists are short here (<= n_agents - 1); insertion sort by one lane.
__global__ void csc_transpose_sort_kernel(int N, const int32_t* __restrict__ t_off, int32_t* __restrict__ t_dst,
                                          int32_t* __restrict__ t_pos) {
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < N; u += gridDim.x * blockDim.x) {
    const int k0 = t_off[u], k1 = t_off[u + 1];
    for (int k = k0 + 1; k < k1; ++k) {
      const int p = t_pos[k], d = t_dst[k];
      int j = k - 1;
      while (j >= k0 && t_pos[j] > p) {
        t_pos[j + 1] = t_pos[j];
        t_dst[j + 1] = t_dst[j];
        --j;
      }
      t_pos[j + 1] = p;
      t_dst[j + 1] = d;
    }
  }
}

// Transpose for batches of small graphs whose talk edges never leave their graph (every batch built by
// batch() / from_padded_obs / from_obs_dicts): one wavefront per graph, lane = source.  Pass 1 counts the out-edges
// of every source (the graph's in-edges are scanned once, wave-uniform loads); the graph's out-slots start at the
// CSC position of its first in-edge, so no global scan is needed; pass 2 walks the destinations in order and appends,
// which leaves every out-list sorted by CSC position.  No atomics, no workspace, one launch.
__global__ __launch_bounds__(kThreads) void csc_transpose_env_kernel(const int32_t* __restrict__ talk_off,
                                                                     const int32_t* __restrict__ talk_src,
                                                                     const int32_t* __restrict__ graph_off, int B,
                                                                     int32_t* __restrict__ t_off,
                                                                     int32_t* __restrict__ t_dst,
                                                                     int32_t* __restrict__ t_pos) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int b = blockIdx.x * kWavesPerBlock + wave; b < B; b += gridDim.x * kWavesPerBlock) {
    const int a0 = graph_off[b], a1 = graph_off[b + 1];
    const int e_lo = talk_off[a0], e_hi = talk_off[a1];
    int base = e_lo;
    for (int u0 = a0; u0 < a1; u0 += kWave) {      // one trip when the graph has <= 64 agents
      const int u = u0 + lane;
      int cnt = 0;
      for (int e = e_lo; e < e_hi; ++e) cnt += talk_src[e] == u;
      const int inc = wave_scan_inclusive(cnt, lane);
      int pos = base + inc - cnt;
      if (u < a1) t_off[u] = pos;
      for (int d = a0; d < a1; ++d) {
        const int d1 = talk_off[d + 1];
        for (int e = talk_off[d]; e < d1; ++e)
          if (talk_src[e] == u) {
            t_pos[pos] = e;
            t_dst[pos] = d;
            ++pos;
          }
      }
      base += __shfl(inc, kWave - 1);
    }
    if (lane == 0) t_off[a1] = e_hi;   // == the next graph's first slot; the last graph closes the array
  }
}

inline int order_blocks(int N) {
  // one block per 256 destinations up to 256 blocks: the stable in-block ranking of the scatter pass is O(256) per
  // destination and tile, so few large blocks are slow (16 blocks at N = 32768: 26 us vs 5 us with 128)
  const int b = (N + kOrderThreads - 1) / kOrderThreads;
  return b < 1 ? 1 : (b > kOrderMaxBlocks ? kOrderMaxBlocks : b);
}


// ---- the "prefix sum (caller)" of the builder: up to four exclusive scans in ONE launch ------------------------------------
// out_k[0] = 0, out_k[i + 1] = in_k[0] + ... + in_k[i]: seen / near / talk offsets over the N agents and the per-environment
// edge bases over the B environments.  One 1024-thread workgroup per array walks it in rounds of 8192 elements (thread ->
// 8 consecutive elements, wave shuffles + a 16-entry LDS scan, running carry): 32 768 agents take four rounds.  Replaces four
// torch.cumsum calls + zero fills + slice copies (14 launches) per graph: the device builder of a large batch is launch-bound.
struct ScanArgs {
  const int32_t* in[4];
  int32_t* out[4];
  int n[4];
};

__global__ __launch_bounds__(1024) void offsets_scan4_kernel(ScanArgs a) {
  constexpr int IT = 8;
  const int32_t* __restrict__ in = a.in[blockIdx.x];
  int32_t* __restrict__ out = a.out[blockIdx.x];
  const int n = a.n[blockIdx.x];
  if (out == nullptr) return;
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) out[0] = 0;
  int carry = 0;
  for (int base = 0; base < n; base += 1024 * IT) {
    const int i0 = base + tid * IT;
    int v[IT], s = 0;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      v[k] = (i0 + k < n) ? in[i0 + k] : 0;
      s += v[k];
    }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int wprefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int t = wsum[w];
      if (w < wave) wprefix += t;
      total += t;
    }
    int run = carry + wprefix + incl - s;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      run += v[k];
      if (i0 + k < n) out[i0 + k + 1] = run;
    }
    carry += total;
    __syncthreads();
  }
}

}  // namespace
}  // namespace uavgnn

using namespace uavgnn;

extern "C" int uavgnn_obs_degrees(const float* gt, int M, int Fg, const float* ubs, int U, int Fu, int N,
                                  int32_t* deg_seen, int32_t* deg_near, uavgnn_stream_t stream) {
  if (N < 0 || (!gt && M > 0) || (!ubs && U > 0) || !deg_seen || !deg_near || M < 0 || U < 0) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  hipLaunchKernelGGL(obs_degrees_kernel, dim3(capped_grid(N, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), gt, M, Fg + 1, ubs, U, Fu + 1, N, deg_seen, deg_near);
  return launch_status();
}

extern "C" int uavgnn_obs_compact(const float* gt, int M, int Fg, const float* ubs, int U, int Fu, int N,
                                  const int32_t* seen_off, const int32_t* near_off, float* x_gt, float* x_ubs,
                                  uavgnn_stream_t stream) {
  if (N < 0 || (!gt && M > 0) || (!ubs && U > 0) || !seen_off || !near_off) return UAVGNN_EINVAL;
  if (Fg != 4 || Fu != 2) return UAVGNN_EUNSUPPORTED;
  if (N == 0) return 0;
  hipLaunchKernelGGL(obs_compact_kernel, dim3(capped_grid(N, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), gt, M, Fg, ubs, U, Fu, N, seen_off, near_off, x_gt, x_ubs);
  return launch_status();
}

extern "C" int uavgnn_talk_degrees(const float* d_u2u, int n, int B, float r_comm, int32_t* deg_in,
                                   int32_t* env_edges, uavgnn_stream_t stream) {
  if (B < 0 || !d_u2u || !deg_in || !env_edges) return UAVGNN_EINVAL;
  if (n < 1 || n > kWave) return UAVGNN_EUNSUPPORTED;
  if (B == 0) return 0;
  hipLaunchKernelGGL(talk_degrees_kernel, dim3(capped_grid(B, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), d_u2u, n, B, r_comm, deg_in, env_edges);
  return launch_status();
}

extern "C" int uavgnn_talk_compact(const float* d_u2u, int n, int B, float r_comm, const int32_t* talk_off,
                                   const int32_t* env_base, int32_t* talk_src, int32_t* talk_eid,
                                   uavgnn_stream_t stream) {
  if (B < 0 || !d_u2u || !talk_off || !env_base || !talk_src || !talk_eid) return UAVGNN_EINVAL;
  if (n < 1 || n > kWave) return UAVGNN_EUNSUPPORTED;
  if (B == 0) return 0;
  hipLaunchKernelGGL(talk_compact_kernel, dim3(capped_grid(B, kWavesPerBlock, 4096)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), d_u2u, n, B, r_comm, talk_off, env_base, talk_src, talk_eid);
  return launch_status();
}

extern "C" size_t uavgnn_degree_order_workspace_bytes(int N) {   // the (bin, block) cells + the scan's tile totals
  const size_t cells = static_cast<size_t>(kOrderBins) * order_blocks(N);
  return (cells + scan_scratch_elems(static_cast<long long>(cells))) * sizeof(int32_t);
}

extern "C" int uavgnn_degree_order(const int32_t* seg_off, int N, int32_t* order, void* workspace,
                                   size_t workspace_bytes, uavgnn_stream_t stream) {
  if (N < 0 || !seg_off || (N > 0 && (!order || !workspace))) return UAVGNN_EINVAL;
  if (N == 0) return 0;
  if (workspace_bytes < uavgnn_degree_order_workspace_bytes(N)) return UAVGNN_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int G = order_blocks(N);
  int chunk = (N + G - 1) / G;
  chunk = (chunk + kOrderThreads - 1) / kOrderThreads * kOrderThreads;
  int32_t* cells = static_cast<int32_t*>(workspace);
  hipLaunchKernelGGL(order_hist_kernel, dim3(G), dim3(kOrderThreads), 0, st, seg_off, N, chunk, cells);
  int rc = scan_inclusive_i32(cells, kOrderBins * G, cells + kOrderBins * G, st);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(order_scatter_kernel, dim3(G), dim3(kOrderThreads), 0, st, seg_off, N, chunk, cells, order);
  return launch_status();
}

// Zero fill as a KERNEL (not hipMemsetAsync): inside a captured graph a memset becomes a memset node, and the one place where this
// build met a memset node ordered in front of a kernel that depends on it - torch's semaphore-based full reduction in a ~4000-node
// replayed graph - the dependent kernel did not always see it.  The counters below feed atomics.
namespace uavgnn {
namespace {
__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t* __restrict__ p, long long n) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) p[i] = 0;
}
inline void zero_i32(int32_t* p, long long n, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(zero_i32_kernel, dim3(capped_grid(n, 256)), dim3(256), 0, st, p, n);
}
}  // namespace
}  // namespace uavgnn

extern "C" size_t uavgnn_csc_transpose_workspace_bytes(int N) {   // slot cursors + the scan's tile totals
  const size_t n = static_cast<size_t>(N > 0 ? N : 0);
  return (n + scan_scratch_elems(static_cast<long long>(n))) * sizeof(int32_t);
}

extern "C" int uavgnn_csc_transpose(const int32_t* talk_off, const int32_t* talk_src, int N, int E, int32_t* t_off,
                                    int32_t* t_dst, int32_t* t_pos, void* workspace, size_t workspace_bytes,
                                    uavgnn_stream_t stream) {
  if (N < 0 || E < 0 || !talk_off || !t_off || (E > 0 && (!talk_src || !t_dst || !t_pos)) || (N > 0 && !workspace))
    return UAVGNN_EINVAL;
  if (workspace_bytes < uavgnn_csc_transpose_workspace_bytes(N)) return UAVGNN_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  zero_i32(t_off, static_cast<long long>(N) + 1, st);
  if (N == 0 || E == 0) return launch_status();
  int32_t* cursor = static_cast<int32_t*>(workspace);
  int32_t* sums = cursor + N;
  zero_i32(cursor, N, st);
  hipLaunchKernelGGL(csc_out_degrees_kernel, dim3(capped_grid(E, 256)), dim3(256), 0, st, talk_src, E, t_off);
  int rc = scan_inclusive_i32(t_off + 1, N, sums, st);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(csc_transpose_fill_kernel, dim3(capped_grid(N, 256)), dim3(256), 0, st, talk_off, talk_src, N,
                     t_off, cursor, t_dst, t_pos);
  hipLaunchKernelGGL(csc_transpose_sort_kernel, dim3(capped_grid(N, 256)), dim3(256), 0, st, N, t_off, t_dst, t_pos);
  return launch_status();
}

extern "C" int uavgnn_csc_transpose_env(const int32_t* talk_off, const int32_t* talk_src, const int32_t* graph_off,
                                        int B, int N, int32_t* t_off, int32_t* t_dst, int32_t* t_pos,
                                        uavgnn_stream_t stream) {
  if (B < 0 || N < 0 || !talk_off || !graph_off || !t_off) return UAVGNN_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (B == 0 || N == 0) {
    zero_i32(t_off, static_cast<long long>(N) + 1, st);
    return launch_status();
  }
  hipLaunchKernelGGL(csc_transpose_env_kernel, dim3(capped_grid(B, kWavesPerBlock, 4096)), dim3(kThreads), 0, st,
                     talk_off, talk_src, graph_off, B, t_off, t_dst, t_pos);
  return launch_status();
}

extern "C" int uavgnn_build_graph_small_max_agents(void) { return kSmallMaxAgents; }

extern "C" int uavgnn_build_graph_small(const float* gt, int M, int Fg, const float* ubs, int U, int Fu, const float* d_u2u,
                                        int n, int B, float r_comm, int32_t* seen_off, int32_t* near_off, int32_t* talk_off,
                                        float* x_gt, float* x_ubs, int32_t* talk_src, int32_t* talk_eid, int32_t* graph_off,
                                        uavgnn_stream_t stream) {
  if (B < 0 || n < 1 || M < 0 || U < 0 || !seen_off || !near_off || !graph_off || (B > 0 && M > 0 && (!gt || !x_gt)) ||
      (B > 0 && U > 0 && (!ubs || !x_ubs)) || (d_u2u && (!talk_off || !talk_src || !talk_eid)))
    return UAVGNN_EINVAL;
  if (Fg != 4 || Fu != 2 || n > kWave || static_cast<long long>(B) * n > kSmallMaxAgents) return UAVGNN_EUNSUPPORTED;
  hipLaunchKernelGGL(build_graph_small_kernel, dim3(1), dim3(kSmallThreads), 0, static_cast<hipStream_t>(stream), gt, M, ubs,
                     U, d_u2u, n, B, r_comm, seen_off, near_off, talk_off, x_gt, x_ubs, talk_src, talk_eid, graph_off);
  return launch_status();
}

extern "C" int uavgnn_offsets_scan4(const int32_t* d0, int n0, int32_t* o0, const int32_t* d1, int n1, int32_t* o1,
                                    const int32_t* d2, int n2, int32_t* o2, const int32_t* d3, int n3, int32_t* o3,
                                    uavgnn_stream_t stream) {
  ScanArgs a;
  const int32_t* in[4] = {d0, d1, d2, d3};
  int32_t* out[4] = {o0, o1, o2, o3};
  const int n[4] = {n0, n1, n2, n3};
  int used = 0;
  for (int k = 0; k < 4; ++k) {
    if (n[k] < 0 || (out[k] != nullptr && n[k] > 0 && in[k] == nullptr)) return UAVGNN_EINVAL;
    a.in[k] = in[k];
    a.out[k] = out[k];
    a.n[k] = n[k];
    if (out[k] != nullptr) used = k + 1;
  }
  if (used == 0) return 0;
  hipLaunchKernelGGL(offsets_scan4_kernel, dim3(used), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
  return launch_status();
}
