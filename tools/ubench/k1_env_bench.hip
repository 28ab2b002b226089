// Stand-alone timing of the fused K1 forward (csrc/gatv2_hetero.hip) at C3 size on SURVEY's degree distributions, without
// Python: back-to-back launches between two events, whole kernel and phase ablations, + a checksum of the output so that
// variants can be compared for equality.  argv[1]: dist (env | dense | zero), argv[2]: B (default 4096).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iuav_bs_ctrl_amd/csrc [-DK1_OLD] [-D...] tools/ubench/k1_env_bench.hip -o tools/ubench/bin/k1_env_bench[_pair]
#define K1_ABLATE 1
#define K1_STANDALONE 1
#if defined(K1_OLD)   // the rounds 1-3 kernel (pairs of destinations per row tile in phase N): -DK1_OLD [-DK1_BF16Z=0]
#include "../../uav_bs_ctrl_amd/csrc/gatv2_hetero_pair.inc"
#else
#include "../../uav_bs_ctrl_amd/csrc/gatv2_hetero.hip"
#endif
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

__global__ void empty_kernel(float* out) {
  if (out == nullptr) out[threadIdx.x] = 0.f;
}
// calibration: a pure streaming write of the [N, 512] output (float4 per lane, grid-stride), plain and non-temporal
template <bool NT>
__global__ __launch_bounds__(256) void write_rows_kernel(uavgnn::f32x4* __restrict__ out, size_t n4) {
  const uavgnn::f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += size_t(gridDim.x) * 256) {
    if (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}

// shader clock probe: s_memtime ticks (= shader cycles, MI355X_MICROARCH.md) across a dependent FMA chain of known length
__global__ void clock_probe(unsigned long long* ticks, float* sink, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) a = fmaf(a, b, 1e-7f);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  if (a == 12345.f) sink[0] = a;
}

template <class T>
static T* dev(const std::vector<T>& v) {
  T* p;
  hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T));
  hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

// K1_IMAGE=1: launches go through uavgnn_gatv2_hetero_fwd_image with an image prepared once (the rollout path of the learner)
static void* g_image = nullptr;
static int k1_call(const float* xg, int Es, const int32_t* so, const int32_t* ord, const float* xu, int En, const int32_t* no,
                   const float* xa, int N, const float* const* ps, const float* const* pn, float* out, int ld, float* sv_s,
                   float* sv_n, int ph) {
#if !defined(K1_OLD)
  if (g_image != nullptr)
    return uavgnn_gatv2_hetero_fwd_image(xg, Es, so, ord, xu, En, no, xa, N, ps, pn, 4, 64, 0.2f, g_image, out, ld, sv_s, sv_n, ph, nullptr);
#endif
  return uavgnn_gatv2_hetero_fwd_phases(xg, Es, so, ord, xu, En, no, xa, N, ps, pn, 4, 64, 0.2f, out, ld, sv_s, sv_n, ph, nullptr);
}

int main(int argc, char** argv) {
  const std::string dist = argc > 1 ? argv[1] : "env";
  const int B = argc > 2 ? atoi(argv[2]) : 4096, n = 8, M = 80, N = B * n, H = 256;
  const int reps = argc > 3 ? atoi(argv[3]) : 50;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f), U01(0.f, 1.f);
  std::vector<int32_t> so(N + 1, 0), no(N + 1, 0);
  for (int v = 0; v < N; ++v) {
    int d = M;
    if (dist == "env") d = U01(rng) < 0.94f ? 0 : 1 + int(U01(rng) * 52) % 52;
    if (dist == "zero") d = 0;
    so[v + 1] = so[v] + d;
    no[v + 1] = no[v] + (n - 1);
  }
  const int Es = so[N], En = no[N];
  std::vector<float> xg(size_t(Es) * 4), xu(size_t(En) * 2), xa(size_t(N) * 2);
  for (auto& f : xg) f = U(rng);
  for (auto& f : xu) f = U(rng);
  for (auto& f : xa) f = U01(rng);
  std::vector<int32_t> order(N);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return so[a + 1] - so[a] > so[b + 1] - so[b]; });
  auto mk = [&](size_t k, float amp) { std::vector<float> v(k); for (auto& f : v) f = amp * U(rng); return dev(v); };
  // W_s, b_s, W_d, b_d, attn, W_r, b_r per relation
  const float* ps[7] = {mk(H * 4, .3f), mk(H, .1f), mk(H * 2, .3f), mk(H, .1f), mk(H, .3f), mk(H * 2, .3f), mk(H, .1f)};
  const float* pn[7] = {mk(H * 2, .3f), mk(H, .1f), mk(H * 2, .3f), mk(H, .1f), mk(H, .3f), mk(H * 2, .3f), mk(H, .1f)};
  float *d_xg = dev(xg), *d_xu = dev(xu), *d_xa = dev(xa);
  int32_t *d_so = dev(so), *d_no = dev(no), *d_ord = dev(order);
  float* out;
  hipMalloc(&out, size_t(N) * 2 * H * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double bytes = 16.0 * Es + 8.0 * En + double(N) * (8 + 8 + 2048);
  printf("dist %s  N %d  E_seen %d  E_near %d  algorithmic bytes %.1f MB\n", dist.c_str(), N, Es, En, bytes / 1e6);
  std::vector<float> host(size_t(N) * 2 * H);
  auto time_us = [&](auto&& fn) {
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
  };
  printf("empty kernel, 512 x 256 threads:            %7.2f us per launch (back to back)\n",
         time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(256), 0, 0, out); }));
  {
    unsigned long long* d_t;
    hipMalloc(&d_t, 8);
    const int iters = 200000;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(clock_probe, dim3(1024), dim3(256), 0, 0, d_t, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long t;
      hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
      printf("clock probe: %d dependent v_fma in %.1f us, %llu s_memtime ticks -> %.0f MHz tick rate, %.2f ticks / fma, %.1f ns / fma\n", iters,
             ms * 1e3, t, t / (ms * 1e3), double(t) / iters, ms * 1e6 / iters);
    }
  }
  const size_t n4 = size_t(N) * 2 * H / 4;
  for (int g : {512, 1024, 2048, 4096}) {
    const double t0 = time_us([&] { hipLaunchKernelGGL(write_rows_kernel<false>, dim3(g), dim3(256), 0, 0, reinterpret_cast<uavgnn::f32x4*>(out), n4); });
    const double t1 = time_us([&] { hipLaunchKernelGGL(write_rows_kernel<true>, dim3(g), dim3(256), 0, 0, reinterpret_cast<uavgnn::f32x4*>(out), n4); });
    printf("streaming write of %.1f MB, %4d blocks: plain %6.2f us (%.2f TB/s)  nontemporal %6.2f us (%.2f TB/s)\n", n4 * 16 / 1e6, g,
           t0, n4 * 16 / t0 / 1e6, t1, n4 * 16 / t1 / 1e6);
  }
#if !defined(K1_OLD)
  if (getenv("K1_IMAGE")) {
    hipMalloc(&g_image, uavgnn_gatv2_hetero_image_bytes());
    const int rc = uavgnn_gatv2_hetero_prepare(ps, pn, 4, 64, 0.2f, g_image, nullptr);
    hipDeviceSynchronize();
    printf("prepared image: %zu bytes, rc %d\n", uavgnn_gatv2_hetero_image_bytes(), rc);
  }
#endif
  for (int ph : {3, 3 | 4096, 3 | 8192, 3 | 12288, 2, 1, 0, 2 | 32, 2 | 64, 2 | 128, 2 | 32 | 64 | 128}) {
    auto run = [&] {
      int rc = k1_call(d_xg, Es, d_so, d_ord, d_xu, En, d_no, d_xa, N, ps, pn, out, 2 * H, nullptr, nullptr, ph);
      if (rc) { printf("rc %d\n", rc); exit(1); }
    };
    hipMemset(out, getenv("K1_POISON") ? 0xff : 0, size_t(N) * 2 * H * 4);   // K1_POISON: unwritten elements stay NaN
    for (int i = 0; i < 3; ++i) run();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) run();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost);
    if (getenv("K1_POISON")) {
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < host.size(); ++i) if (host[i] != host[i]) { if (!bad) first = i; ++bad; }
      printf("phases %3d: %zu unwritten elements (first: row %zu col %zu)\n", ph, bad, first / (2 * H), first % (2 * H));
    }
    double cs = 0, ca = 0;
    for (size_t i = 0; i < host.size(); ++i) { cs += host[i] * double((i % 977) + 1); ca += fabs(host[i]); }
    if (const char* dp = getenv("K1_DUMP")) {   // raw output of the first listed phase set, for element-wise comparison of two builds
      if (ph == 3) { FILE* f = fopen(dp, "wb"); fwrite(host.data(), 4, host.size(), f); fclose(f); }
    }
    const double us = ms * 1e3 / reps;
    printf("phases %3d: %8.2f us  %7.1f GB/s alg (%.3f of 8 TB/s)   checksum %.9e  abs %.9e\n", ph, us, bytes / us / 1e3,
           bytes / us / 1e3 / 8000.0, cs, ca);
  }
#if !defined(K1_OLD)
  {  // time line of the wavefronts of ONE launch (phases bit 10): 100 MHz stamps at entry, after the workgroup prologue, after the
     // phase-N operand loads, after the residual-only rows / the score tiles / the `near` rows of the first block were issued,
     // after the wave's stores have drained, after phase S
    const int waves = 256 * 8;
    unsigned long long* d_dbg;
    hipMalloc(&d_dbg, size_t(waves) * 8 * 8);
    std::vector<unsigned long long> st(size_t(waves) * 8);
    for (int ph : {3}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d_dbg, 0, size_t(waves) * 64);
        k1_call(d_xg, Es, d_so, d_ord, d_xu, En, d_no, d_xa, N, ps, pn, out, 2 * H, reinterpret_cast<float*>(d_dbg), nullptr, ph | 1024);
        hipDeviceSynchronize();
      }
      hipMemcpy(st.data(), d_dbg, st.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull;
      for (int w = 0; w < waves; ++w) if (st[size_t(w) * 8]) t0 = std::min(t0, st[size_t(w) * 8]);
      printf("time line, phases %d (us after the first wavefront's entry; min / mean / max over %d wavefronts):\n", ph, waves);
      const char* names[8] = {"entry", "prologue done", "N operands loaded", "seen rows issued", "tiles + softmax done", "near rows issued", "stores drained", "phase S done"};
      for (int k = 0; k < 8; ++k) {
        double mn = 1e30, mx = 0, sum = 0; int cnt = 0;
        for (int w = 0; w < waves; ++w) { const unsigned long long v = st[size_t(w) * 8 + k]; if (!v) continue; const double us = (v - t0) * 0.01; mn = std::min(mn, us); mx = std::max(mx, us); sum += us; ++cnt; }
        if (cnt) printf("  %-22s %6.2f / %6.2f / %6.2f   (%d)\n", names[k], mn, sum / cnt, mx, cnt);
      }
    }
  }
#endif
  return 0;
}
