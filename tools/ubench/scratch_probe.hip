// Probe: does per-wave scratch (private segment) stay private when two workgroups share a CU?
// Each lane writes a pattern derived from (block, thread, slot) into a dynamically indexed private array (forces scratch),
// spins, reads it back.  Optional LDS allocation mirrors the K1 backward kernel (66 KB: two workgroups per CU).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/scratch_probe.hip -o tools/ubench/bin/scratch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void probe(unsigned* bad, int spin, int rot) {
  __shared__ unsigned char lds[LDS_BYTES];
  volatile unsigned priv[41];   // 164 bytes, like the kernel under suspicion
  const unsigned tag = blockIdx.x * 1024u + threadIdx.x;
  for (int i = 0; i < 41; ++i) priv[(i + rot) % 41] = tag * 64u + static_cast<unsigned>((i + rot) % 41);
  reinterpret_cast<volatile unsigned*>(lds)[threadIdx.x] = tag;
  unsigned miss = 0;
  for (int s = 0; s < spin; ++s) {
    for (int i = 0; i < 41; ++i) {
      const int j = (i + rot + s) % 41;
      miss += (priv[j] != tag * 64u + static_cast<unsigned>(j)) ? 1u : 0u;
    }
    miss += (reinterpret_cast<volatile unsigned*>(lds)[threadIdx.x] != tag) ? 1u : 0u;
  }
  if (miss) atomicAdd(&bad[blockIdx.x < 256 ? 0 : 1], miss);
}

int main() {
  unsigned* bad; HIP_OK(hipMalloc(&bad, 8));
  for (int grid : {256, 300, 512, 2048}) {
    HIP_OK(hipMemset(bad, 0, 8));
    probe<66304><<<grid, 256>>>(bad, 200, 3);
    HIP_OK(hipDeviceSynchronize());
    unsigned h[2]; HIP_OK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    printf("lds 66304 grid %4d: misses in blocks < 256: %u, in blocks >= 256: %u\n", grid, h[0], h[1]);
    HIP_OK(hipMemset(bad, 0, 8));
    probe<1024><<<grid, 256>>>(bad, 200, 3);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
    printf("lds  1024 grid %4d: misses in blocks < 256: %u, in blocks >= 256: %u\n", grid, h[0], h[1]);
  }
  return 0;
}
