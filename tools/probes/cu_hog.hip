// Contention probe (round 4): a kernel that HOLDS n compute units for a given time -- what RCCL's channel
// kernels do to the data-parallel backward.  One workgroup per held CU: 256 threads and 96 KiB of LDS, so neither a GEMM
// workgroup (144 KiB) nor a second hog fits beside it; it spins on the 100-MHz wall clock with s_sleep and touches no memory.
// Build: hipcc --offload-arch=gfx950 -shared -fPIC -O2 tools/r4/cu_hog.hip -o tools/r4/_cu_hog.so
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void cu_hog_kernel(uint64_t ticks, unsigned* sink) {
  extern __shared__ char lds[];
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(64); ++spins; }
  if (sink && spins == 0xffffffffu) { lds[threadIdx.x] = 1; *sink = lds[0]; }      // keeps the LDS allocation alive
}

extern "C" int cu_hog_launch(int n_cu, double milliseconds, void* stream) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cu_hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr = true; }
  if (n_cu <= 0) return 0;
  hipLaunchKernelGGL(cu_hog_kernel, dim3(n_cu), dim3(256), 96 * 1024, reinterpret_cast<hipStream_t>(stream),
                     (uint64_t)(milliseconds * 1e5), (unsigned*)nullptr);
  return (int)hipGetLastError();
}
