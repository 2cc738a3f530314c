// LDS images, addressing and fragment reads shared by the LDS-DMA generations (2, 3, 4), plus pin_s.
// Part of the GEMM family of csrc/gemm.hip (included there, in this order: common, gen1, lds, gen2, gen3, gen4, gemv_gen1);
// not a stand-alone header.
#pragma once

namespace {

// =====================================================================================================
// Generation-2 kernel (all operand layouts): direct-to-LDS loads (global_load_lds_dwordx4: no VGPR staging,
// no ds_write pass) into a 3-stage LDS ring, prefetch distance TWO k-tiles, counted s_waitcnt vmcnt(N)
// (never 0 inside the loop), raw s_barrier -- one barrier per k-tile -- and the DMA issue spread behind the
// MFMA groups.  The generation-1 kernel above drains its loads after ONE tile of compute (~512 MFMA cycles
// per wave), which does not cover global latency under load (measured 0.6 PF).
//
// LDS images (the DMA writes linearly: wave-uniform base + lane*16 B, so every permutation is applied to the
// per-lane SOURCE address and undone by the read):
//   K-contiguous operand ("natural"):  [rows = output index][64 k]  128-B rows, 16-B chunk c stored at
//       c ^ swz(row); fragments by one ds_read_b128 per MFMA operand.
//   contraction-strided operand (dgrad's W, both wgrad operands): DMA'd in its NATURAL global layout
//       [64 k rows][output index], row = TB*2 bytes, chunk c stored at c ^ ((k&3)<<2); fragments by two
//       ds_read_b64_tr_b16 -- the LDS transposing read (lane c of a 16-lane group receives, for j = 0..3,
//       element (c&3) of the 8 bytes addressed by lane 4j + (c>>2): verified on hardware by
//       tools/probes/tr_read_probe.hip).  The XOR term sends the 4 k-rows of one transpose block to the 4
//       different 64-B quarters of the 256-B bank row.  No register transposes, no transposed copies in HBM.
//   Both kinds label MFMA k-slot (g, e) of k-step ks as contraction index 16 ks + 8 g + e, so they mix freely.
// Tile (WM*64) x (WN*64) x 64, WM*WN waves, each wave a 64x64 sub-tile (2x2 MFMA 32x32x16).
// Requirements (checked by the dispatcher): K % 64 == 0.  Output rows/columns beyond M / N are clamped on the
// load side (their products land in rows/columns that are never stored).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef short s16x4_t __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- LDS addressing of the DMA kernel, parameterised by the k-tile depth BKT (64 or 32 halves per row)
//   natural region : [rows][BKT] , row = 2*BKT bytes, NC = BKT/8 16-B chunks per row
//       BKT = 64: chunk c at c ^ swz(row)            (rows r, r+1 share a 256-B bank row)
//       BKT = 32: chunk c at c ^ ((row >> 2) & 3)    (4 rows share a bank row; rows r, r+4, r+8, r+12 get
//                                                      different chunks, so any 16 distinct rows are conflict free)
template <int BKT> __device__ __forceinline__ int nswz(int row) {
  return BKT == 64 ? swz(row) : ((row >> 2) & 3);
}
template <int BKT> __device__ __forceinline__ uint32_t nat_off(int row, int chunk) {
  return (uint32_t)(row * (2 * BKT) + ((chunk ^ nswz<BKT>(row)) << 4));
}

// fragment of a transposed region ([BKT k][TB cols], ROWB bytes per k-row): 32-wide column block at col0.
// Issued through inline asm: with the __builtin_amdgcn_ds_read_tr16_b64 form hipcc (ROCm 7.2) orders the read
// against the in-flight LDS-DMA and emits s_waitcnt vmcnt(0) in front of it, draining the prefetch ring every
// k-step (measured: 70 % of wave cycles parked).  The asm reads are invisible to the compiler's counters, so
// the matching wait is explicit (tr_wait2) and carries the destination registers as in/out operands.
struct TrRaw { u32x2 lo, hi; };
template <int ROWB>
__device__ __forceinline__ uint32_t tr_addr(const char* reg, int col0, int lane) {
  const int G = lane >> 4, cb = G & 1, g = G >> 1, r = (lane & 15) >> 2, qq = lane & 3;
  const int c = (col0 >> 3) + 2 * cb + (qq >> 1);
  const int pc = c ^ (r << 2);
  return (uint32_t)(uintptr_t)(reg) + (8 * g + r) * ROWB + pc * 16 + (qq & 1) * 8;
}
template <int ROWB>
__device__ __forceinline__ void tr_issue(uint32_t addr0, int ks, TrRaw& o) {
  const uint32_t a = addr0 + ks * 16 * ROWB;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
               : "=&v"(o.lo), "=&v"(o.hi) : "v"(a), "n"(4 * ROWB) : "memory");
}
// ---- the same for v_mfma_f32_16x16x32 operands: 16-wide column block at col0; lane group G = lane >> 4 is the
//      k-block (8 contraction rows 8G .. 8G+7, two reads of 4 rows).  All four groups read the same 32 bytes of
//      a k-row, so rows k and k + 8 (same k & 3) must not share banks: the chunk XOR also takes bit 3 of k.
__device__ __forceinline__ int trswz16(int k) { return ((k & 3) << 2) | (((k >> 3) & 1) << 1); }
template <int ROWB>
__device__ __forceinline__ uint32_t tr_addr16(const char* reg, int col0, int lane) {
  const int G = lane >> 4, r = (lane & 15) >> 2, qq = lane & 3;
  const int k = 8 * G + r;
  const int c = (col0 >> 3) + (qq >> 1);
  return (uint32_t)(uintptr_t)(reg) + k * ROWB + ((c ^ trswz16(k)) << 4) + (qq & 1) * 8;
}
template <int ROWB>
__device__ __forceinline__ void tr_issue16(uint32_t addr0, int ks, TrRaw& o) {
  const uint32_t a = addr0 + ks * 32 * ROWB;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
               : "=&v"(o.lo), "=&v"(o.hi) : "v"(a), "n"(4 * ROWB) : "memory");
}
// natural-region fragment through asm as well (used only in kernels that also have asm transposing reads, so that
// no compiler-generated lgkmcnt wait -- which cannot see the asm reads queued behind its own -- lands between
// the read issue and the MFMA group)
__device__ __forceinline__ void nat_issue(uint32_t addr, u32x4& o) {
  asm volatile("ds_read_b128 %0, %1" : "=&v"(o) : "v"(addr) : "memory");
}
__device__ __forceinline__ void nat_wait2(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
}
__device__ __forceinline__ void tr_wait2(TrRaw& a, TrRaw& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : : "memory");
}
template <typename T>
__device__ __forceinline__ typename HT<T>::v8 tr_pack(const TrRaw& r) {
  typename HT<T>::v8 out;
  __builtin_memcpy(&out, &r.lo, 8);
  __builtin_memcpy(reinterpret_cast<char*>(&out) + 8, &r.hi, 8);
  return out;
}

// pin a wave-uniform value in scalar registers: opaque to the optimiser, so it cannot be rematerialised by re-reading the
// kernel-argument segment at every use (an s_load + lgkmcnt(0) inside each of the epilogue's 32 passes otherwise)
template <typename V> __device__ __forceinline__ void pin_s(V& x) { asm volatile("" : "+s"(x)); }

}  // namespace
