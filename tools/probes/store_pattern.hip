// Store-pattern probe for the GEMM epilogue: a wave writes a 128x128 bf16 sub-tile of a row-major [M][N] matrix with
// 16 B per lane, (A) 8 lanes per 128-B row segment, 8 rows per instruction (today's LDS-strip epilogue) or
// (B) 4 lanes per 64-B row segment, 16 rows per instruction (what a permlane-swap epilogue would emit).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_pattern.hip -o gpurun_out/store_pattern && gpurun_out/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned short* C, int M, int N, int tiles_n, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int m0 = (t / tiles_n) * 256 + (wave >> 1) * 128, n0 = (t % tiles_n) * 256 + (wave & 1) * 128;
    const u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
    if (MODE == 0) {
      for (int p = 0; p < 32; ++p) {           // 16 passes of 8 rows x 128 B for each 64-column half
        const int half = p >> 4, r = (p & 15) * 8 + (lane >> 3), c = half * 64 + (lane & 7) * 8;
        *reinterpret_cast<u32x4*>(C + (size_t)(m0 + r) * N + n0 + c) = v;
      }
    } else {
      for (int i = 0; i < 8; ++i)              // 8 row blocks of 16 rows, 4 instructions of 16 rows x 64 B each
        for (int jp = 0; jp < 4; ++jp) {
          const int kb = lane >> 4, r = 16 * i + (lane & 15), c = 32 * jp + 16 * (kb & 1) + 8 * (kb >> 1);
          *reinterpret_cast<u32x4*>(C + (size_t)(m0 + r) * N + n0 + c) = v;
        }
    }
  }
}
int main() {
  const int M = 26112, N = 10240, tiles_n = N / 256, ntiles = (M / 256) * tiles_n;
  unsigned short* C; hipMalloc(&C, (size_t)M * N * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      hipEventRecord(a);
      for (int it = 0; it < 5; ++it) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, C, M, N, tiles_n, ntiles);
        else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, C, M, N, tiles_n, ntiles);
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("mode %d (%s): %.1f us per pass, %.2f TB/s\n", mode, mode ? "16 rows x 64 B" : "8 rows x 128 B", ms * 200.f, (double)M * N * 2 * 5 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
