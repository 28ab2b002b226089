// Stand-alone timing of the bf16x3 GRU cell (csrc/gru_x3.hip) at C3 size with parts of the kernel compiled out
// (-DUAVGNN_X3_DBG=1: no LDS fragment reads / MFMA, 2: no global loads inside the slice loop, 3: no split / LDS writes).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iuav_bs_ctrl_amd/csrc [-DUAVGNN_X3_DBG=n] tools/ubench/gru_x3_bench.hip -o ...
#include "../../uav_bs_ctrl_amd/csrc/gru_x3.hip"
#include <cstdio>
#include <vector>
int main() {
  const int N = 32768, K = 320, H = 256;
  float *inp, *h, *Wih, *Whh, *bih, *bhh, *out;
  void* planes;
  hipMalloc(&inp, 4ll * N * K); hipMalloc(&h, 4ll * N * H); hipMalloc(&out, 4ll * N * H);
  hipMalloc(&Wih, 4ll * 3 * H * K); hipMalloc(&Whh, 4ll * 3 * H * H); hipMalloc(&bih, 4 * 3 * H); hipMalloc(&bhh, 4 * 3 * H);
  hipMalloc(&planes, uavgnn_gru_cell_x3_workspace_bytes(K, H));
  std::vector<float> v(size_t(N) * K);
  for (size_t i = 0; i < v.size(); ++i) v[i] = 0.001f * float((i * 7919) % 2003) - 1.f;
  hipMemcpy(inp, v.data(), 4ll * N * K, hipMemcpyHostToDevice); hipMemcpy(h, v.data(), 4ll * N * H, hipMemcpyHostToDevice);
  hipMemcpy(Wih, v.data(), 4ll * 3 * H * K, hipMemcpyHostToDevice); hipMemcpy(Whh, v.data(), 4ll * 3 * H * H, hipMemcpyHostToDevice);
  hipMemset(bih, 0, 4 * 3 * H); hipMemset(bhh, 0, 4 * 3 * H);
  uavgnn_gru_split_weights(Wih, K, Whh, H, planes, nullptr);
  auto run = [&] { uavgnn_gru_cell_fwd_x3(inp, K, K, h, N, H, planes, bih, bhh, out, nullptr, nullptr); };
  for (int i = 0; i < 3; ++i) run();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
#ifndef UAVGNN_X3_DBG
#define UAVGNN_X3_DBG 0
#endif
  printf("gru_cell_fwd_x3 DBG=%d: %.1f us per launch\n", UAVGNN_X3_DBG, ms * 1000.f / 20);
  return 0;
}
