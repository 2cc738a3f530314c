// Occupancy the runtime grants the dense attention kernels of the train step (DROP == 2 instantiations) at their dynamic LDS
// sizes, for ring depths 2 and 3 -- the question behind profiles/r05_attention_stages_ab.log (did a two-stage ring really put
// more workgroups on a CU?).  Build and run ON THE GPU BOX:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Icogview_amd/csrc [-DCOGV_ATTN_STAGES=3] tools/probes/attn_occupancy.hip -o /tmp/attn_occ && /tmp/attn_occ
#include "../../cogview_amd/csrc/attention.hip"
#include <cstdio>

template <typename K> static void report(const char* name, K kernel, int dyn) {
  int blocks = -1;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, NT, (size_t)dyn);
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel));
  printf("%-28s stages %d dynamic LDS %6d B static %5zu B regs %3d  -> %d workgroups of %d threads per CU (%s)\n", name, NSTG, dyn,
         a.sharedSizeBytes, a.numRegs, blocks, NT, hipGetErrorString(e));
}

int main() {
  report("attn_fwd<f16, dense, bits>", &attn_fwd_kernel<f16_t, false, 2>, NSTG * 2 * TILE);
  report("attn_bwd_dq<f16, dense, bits>", &attn_bwd_dq_kernel<f16_t, false, 2>, ring_bytes(2 * TILE + 1024, true));
  report("attn_bwd_dkdv<f16, dense, bits>", &attn_bwd_dkdv_kernel<f16_t, false, 2>, ring_bytes(2 * TILE + 1536, true));
  return 0;
}
