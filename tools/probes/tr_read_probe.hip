#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const int* addr_in, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)lds + (uint32_t)addr_in[threadIdx.x];
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = r[0] & 0xffff; out[threadIdx.x * 4 + 1] = r[0] >> 16;
  out[threadIdx.x * 4 + 2] = r[1] & 0xffff; out[threadIdx.x * 4 + 3] = r[1] >> 16;
}
int main() {
  int h_addr[64]; uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int exp = 0; exp < 3; ++exp) {
    for (int l = 0; l < 64; ++l) {
      if (exp == 0) h_addr[l] = l * 8;                                   // contiguous: lane l -> elements 4l..4l+3
      if (exp == 1) h_addr[l] = ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 32;   // 4 rows of stride 128 elements, 16 cols per group
      if (exp == 2) h_addr[l] = (l & 15) * 64 + (l >> 4) * 8;            // every lane its own row (stride 32 elements)
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("exp %d (lane: addr_elem -> 4 results)\n", exp);
    for (int l = 0; l < 64; ++l) printf("  l%02d a%4d -> %4d %4d %4d %4d%s", l, h_addr[l] / 2, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3], (l % 2) ? "\n" : " |");
  }
  return 0;
}
