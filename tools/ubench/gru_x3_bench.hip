// Stand-alone timing of the bf16x3 GRU cell (csrc/gru_x3.hip) at C3 size with parts of the kernel compiled out
// (-DUAVGNN_X3_DBG=1: no global loads inside the slice loop, 2: no staging at all, 3: prologue + epilogue only;
// -DUAVGNN_X3_DBG_LD=1 / 2: only the activation / only the weight-plane loads).  argv[1]: 1 = interleaved staging (default), 0 = blocks.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iuav_bs_ctrl_amd/csrc [-DUAVGNN_X3_DBG=n] tools/ubench/gru_x3_bench.hip -o ...
#include "../../uav_bs_ctrl_amd/csrc/gru_x3.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
  const int il = argc > 1 ? atoi(argv[1]) : 1;   // 1 = staging interleaved with the MFMAs (default of the library), 0 = staging in blocks
  const int flags = il ? 0 : UAVGNN_GRU_STAGING_BLOCKS;
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
  auto run = [&] { uavgnn_gru_cell_fwd_x3_opts(inp, K, K, nullptr, 0, 0, h, N, H, planes, bih, bhh, out, nullptr, flags, nullptr); };
  for (int i = 0; i < 3; ++i) run();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
#ifndef UAVGNN_X3_DBG
#define UAVGNN_X3_DBG 0
#endif
  std::vector<float> o(size_t(N) * H);
  hipMemcpy(o.data(), out, 4ll * N * H, hipMemcpyDeviceToHost);
  double cs = 0;
  for (size_t i = 0; i < o.size(); ++i) cs += o[i] * double((i % 977) + 1);
  printf("gru_cell_fwd_x3 DBG=%d interleave=%d: %.1f us per launch   checksum %.9e\n", UAVGNN_X3_DBG, il, ms * 1000.f / 200, cs);
  return 0;
}
