// What exactly do v_permlane16_swap / v_permlane32_swap exchange?  Prints, per 16-lane row, where each result came from.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;        // a: "vdst" operand, b: "src" operand
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"permlane16_swap r[0]", "permlane16_swap r[1]", "permlane32_swap r[0]", "permlane32_swap r[1]"};
  for (int t = 0; t < 4; ++t) {
    printf("%s: ", nm[t]);
    for (int row = 0; row < 4; ++row) printf("row%d<-%s row%d  ", row, h[64 * t + 16 * row] >= 100 ? "b" : "a", (h[64 * t + 16 * row] % 100) / 16);
    printf("\n");
  }
  return 0;
}
