// MFMA GEMM family for the CogView GPT hot path (gfx950).
//
//   C[M,N] = epilogue( A_op[M,K] * B_op[N,K]^T )         fp16/bf16 inputs, fp32 accumulate
//
// Replaces the cuBLAS calls reached through F.linear in the reference:
//   forward  Y = X W^T + b         mpu/layers.py:243 (ColumnParallelLinear), :319 (RowParallelLinear),
//                                  model/gpt2_modeling.py:117 (tied logits)          -> transA=0, transB=0
//   dgrad    dX = dY W             autograd of the above                            -> transA=0, transB=1
//   wgrad    dW = dY^T X           autograd of the above                            -> transA=1, transB=1
//
// "trans" means the operand is stored with the contraction index as the SLOW dimension (A stored [K][M], B stored
// [K][N]): dgrad's W and both weight-gradient operands.  No transposed copies exist in HBM.
//
// Four tile kernels, newest first, plus gemv_kernel for M <= 8 (decode steps: a pure HBM stream of the weights);
// dispatch: launch_gemm:
//   generation 4  gemm_w4_kernel     256x256x64 tiles, 4 waves of 128x128 (accumulators fill the AGPR file), software-
//                                    pipelined quarter-steps, LDS-DMA granule ring, persistent with per-XCD work queues,
//                                    up to 16 problems per launch.  Default for M, N >= 256.  Its 2 x 4 instantiations
//                                    compile as separate translation units (-DCOGV_W4_TU=k, build.py).
//   generation 3  gemm_pp64_kernel   the same tile, ring and queues with 8 waves of 128x64 in a ping-pong schedule
//                                    (kernel_variant 9; 5-15 % slower than generation 4: 1.5x the LDS fragment bytes).
//   generation 2  gemm_glds_kernel   256x128x32 tiles, 4 waves, 3-stage LDS-DMA ring, 2 workgroups per CU.
//                                    For M or N < 256 and operands >= 4 GiB.
//   generation 1  gemm_kernel        128x128x64 tiles, register-staged with register transposes.  For K % 64 != 0 and
//                                    other unaligned shapes.
// All share the fused epilogue (epilogue8): bias, GeLU (+ stored pre-activation), dGeLU, dropout, += C, abs-max for
// Sandwich-LN, and (generations 3, 4) the bias-gradient column sums of the output.
#include "common.cuh"
#include "cogview_hip.h"
#include "gemm_shared.cuh"

#include <cstdlib>
#include <type_traits>

#ifndef COGV_EXP
#define COGV_EXP 0     // schedule experiments of tools/probes/{gemm_exp,w4_dev}.py (bit 0: no DMA, 1: no reads, 2: no MFMA, 3: DMA re-reads k-tiles 0..3, 4: clock probe, 6: no epilogue math/stores, 11: no epilogue at all, 12: no barriers, 13: no DMA waits)
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;


// Up to MAX_GROUP independent problems of one layout in ONE persistent launch.  The four weight gradients of a
// transformer layer are 300 + 100 + 400 + 400 tiles of 256x256 = 4.7 rounds of 256 CUs (each one alone leaves its
// last round half empty or needs split-K slabs); four layers' worth is 18.75 rounds, so the partial last round
// costs 1.3 % instead of 6 %.
constexpr int MAX_GROUP = 16;
struct GroupArgs {
  GemmArgs g[MAX_GROUP];
  int item_start[MAX_GROUP + 1];     // prefix sums of tiles_m * tiles_n * splitk
  int count;
  int* sched;                        // [0..7] per-XCD item counters, [8] finished workgroups: 0 at launch, re-armed by the last workgroup
  int group_m;                       // generation 4: tile rows per raster group (the 32 CUs of an XCD work on group_m x 32/group_m tiles)
  // generation 4, cross-item prefetch (round 4): one problem, no split-K, an even number (>= 4) of k-tiles, and the three
  // divisions of the tile order replaced by multiplications (w4_tile_fast) that the host verified against w4_tile_slow for
  // every item of this geometry.  xp_ok = 0: every item boundary takes the set-up + prologue path.
  int xp_ok;
  uint32_t xp_magic_ig, xp_magic_gfull, xp_magic_gtail;
};

// Tile order of the generation-3 / 4 kernels: item (position in the launch's work list) -> tile row / column.  Workgroup ids
// are dealt to the 8 XCDs round robin; inside an XCD the tiles run in raster groups of group_m tile rows x all tile columns,
// row fastest (the 32 CUs of an XCD work on group_m x 32 / group_m neighbouring tiles: shared operand panels in one L2).
__host__ __device__ inline void w4_tile_slow(uint32_t bid, uint32_t tiles_m, uint32_t tiles_n, uint32_t group_m, uint32_t& tm, uint32_t& tn) {
  const uint32_t nwg = tiles_m * tiles_n, q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const uint32_t wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  const uint32_t in_group = group_m * tiles_n, group_id = wgid / in_group, first_m = group_id * group_m;
  const uint32_t gsz = tiles_m - first_m < group_m ? tiles_m - first_m : group_m;
  tm = first_m + (wgid % in_group) % gsz;
  tn = (wgid % in_group) / gsz;
}
// x / d as the high word of x * ceil(2^32 / d): exact while x * d < 2^32 (d = 1: magic 0, handled by the caller)
inline uint32_t w4_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d); }
__host__ __device__ inline uint32_t w4_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((unsigned long long)a * b) >> 32); }
__host__ __device__ inline void w4_tile_fast(uint32_t bid, uint32_t tiles_m, uint32_t tiles_n, uint32_t group_m, uint32_t magic_ig,
                                             uint32_t magic_gfull, uint32_t magic_gtail, uint32_t& tm, uint32_t& tn) {
  const uint32_t nwg = tiles_m * tiles_n, q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const uint32_t wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  const uint32_t in_group = group_m * tiles_n;
  const uint32_t group_id = in_group == 1 ? wgid : w4_mulhi(wgid, magic_ig);
  const uint32_t rem = wgid - group_id * in_group, first_m = group_id * group_m;
  const bool tail = tiles_m - first_m < group_m;
  const uint32_t gsz = tail ? tiles_m - first_m : group_m, mg = tail ? magic_gtail : magic_gfull;
  tn = gsz == 1 ? rem : w4_mulhi(rem, mg);
  tm = first_m + rem - tn * gsz;
}

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }

// ---- staging: K-contiguous operand.  Tile rows = output index (m or n), 64 k per row.
template <typename T>
__device__ __forceinline__ void load_nat(const T* __restrict__ base, int ld, int row0, int nrows, int k0, int K,
                                         u32x4 (&r)[4]) {
  const int t = threadIdx.x;
  const int chunk = t & 7;
  const int kk = k0 + chunk * 8;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (t >> 3) + 32 * p;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row0 + row < nrows && kk < K)
      v = *reinterpret_cast<const u32x4*>(base + (size_t)(row0 + row) * ld + kk);
    r[p] = v;
  }
}
__device__ __forceinline__ void store_nat(char* lds, const u32x4 (&r)[4]) {
  const int t = threadIdx.x;
  const int chunk = t & 7;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (t >> 3) + 32 * p;
    *reinterpret_cast<u32x4*>(lds + row * 128 + ((chunk ^ swz(row)) << 4)) = r[p];
  }
}
// ---- staging: K-strided operand stored [K][rows]; tile = 64 k-rows x 128 columns.
template <typename T>
__device__ __forceinline__ void load_tr(const T* __restrict__ base, int ld, int col0, int ncols, int k0, int K,
                                        u32x4 (&r)[4]) {
  const int t = threadIdx.x;
  const int c = t & 15;          // 8-column chunk
  const int kr = (t >> 4) * 4;   // first of 4 k-rows
  const int col = col0 + c * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (k0 + kr + i < K && col < ncols)
      v = *reinterpret_cast<const u32x4*>(base + (size_t)(k0 + kr + i) * ld + col);
    r[i] = v;
  }
}
__device__ __forceinline__ void store_tr(char* lds, const u32x4 (&r)[4]) {
  const int t = threadIdx.x;
  const int c = t & 15;
  const int kr = (t >> 4) * 4;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    // even column 8c+2w : low halves ; odd column 8c+2w+1 : high halves
    u32x2 lo, hi;
    lo[0] = (r[0][w] & 0xffffu) | (r[1][w] << 16);
    lo[1] = (r[2][w] & 0xffffu) | (r[3][w] << 16);
    hi[0] = (r[0][w] >> 16) | (r[1][w] & 0xffff0000u);
    hi[1] = (r[2][w] >> 16) | (r[3][w] & 0xffff0000u);
    const int row_e = c * 8 + 2 * w, row_o = row_e + 1;
    *reinterpret_cast<u32x2*>(lds + row_e * 128 + ((((kr >> 3)) ^ swz(row_e)) << 4) + ((kr & 4) << 1)) = lo;
    *reinterpret_cast<u32x2*>(lds + row_o * 128 + ((((kr >> 3)) ^ swz(row_o)) << 4) + ((kr & 4) << 1)) = hi;
  }
}

template <typename T>
__device__ __forceinline__ typename HT<T>::v8 read_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const typename HT<T>::v8*>(lds + row * 128 + ((chunk ^ swz(row)) << 4));
}


template <typename T, bool AT, bool BT>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // stage s: A at smem + s*32768, B at smem + s*32768 + 16384 (no static LDS: keeps the base 16-B aligned)

  // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous run of tiles,
  //      ordered in groups of 8 tile-rows so that neighbours share A row-panels / B column-panels in L2.
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  constexpr int GROUP_M = 8;
  const int in_group = GROUP_M * p.tiles_n;
  const int group_id = wgid / in_group;
  const int first_m = group_id * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tile_m = first_m + (wgid % in_group) % gsz;
  const int tile_n = (wgid % in_group) / gsz;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = blockIdx.y * p.ktiles_per_split;
  const int kt_end = min(nk_total, kt_begin + p.ktiles_per_split);

  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int fr = lane & 31, fg = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 ra[4], rb[4];
  auto g_load = [&](int kt) {
    const int k0 = kt * BK;
    if (AT) load_tr<T>(A, p.lda, m0, p.M, k0, p.K, ra); else load_nat<T>(A, p.lda, m0, p.M, k0, p.K, ra);
    if (BT) load_tr<T>(B, p.ldb, n0, p.N, k0, p.K, rb); else load_nat<T>(B, p.ldb, n0, p.N, k0, p.K, rb);
  };
  auto l_store = [&](int s) {
    char* la = smem + s * 32768; char* lb = la + 16384;
    if (AT) store_tr(la, ra); else store_nat(la, ra);
    if (BT) store_tr(lb, rb); else store_nat(lb, rb);
  };

  if (kt_begin < kt_end) {
    g_load(kt_begin);
    l_store(0);
  }
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool more = (kt + 1 < kt_end);
    if (more) g_load(kt + 1);
    const char* la = smem + cur * 32768; const char* lb = la + 16384;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      typename HT<T>::v8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = read_frag<T>(la, wm + 32 * i + fr, 2 * ks + fg);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = read_frag<T>(lb, wn + 32 * j + fr, 2 * ks + fg);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = HT<T>::mfma32(fa[i], fb[j], acc[i][j]);
    }
    if (more) l_store(cur ^ 1);
    __syncthreads();
  }

  // ---- stage the fp32 C tile in LDS ([128][128] floats, 64 KiB) for a coalesced epilogue
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * fg;
        const int col = wn + 32 * j + fr;
        ct[row * BN + col] = acc[i][j][e];
      }
  __syncthreads();

  uint32_t amax_pk = 0u;
  const int cchunk = (threadIdx.x & 15) * 8;
#pragma unroll 1
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 16 + (threadIdx.x >> 4);
    const int m = m0 + row, n = n0 + cchunk;
    if (m < p.M && n < p.N) {
      float v[8];
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(ct + row * BN + cchunk);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(ct + row * BN + cchunk + 4);
      v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3];
      v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
      if (p.splitk > 1) {
        float* w = p.ws + ((size_t)blockIdx.y * p.M + m) * p.N + n;
        *reinterpret_cast<f32x4*>(w) = x0;
        *reinterpret_cast<f32x4*>(w + 4) = x1;
      } else {
        amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n, v));
      }
    }
  }
  if ((p.flags & COGV_EPI_ABSMAX) && p.splitk <= 1) {
    // fmaxf drops NaNs, so a NaN anywhere in the tile is carried by a flag and published as a quiet-NaN
    // bit pattern (larger than every finite value under the unsigned ordering used by the atomic)
    const float bm = absmax_pk_block<T>(amax_pk, reinterpret_cast<uint32_t*>(smem));
    if (threadIdx.x == 0) atomic_max_nonneg(p.absmax, bm);
  }
}

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

// ---- epilogue shared by the LDS-DMA kernels: the fp32 C tile goes through the (now idle) ring in NH row slabs,
//      then out with coalesced 16-byte accesses through the fused epilogue8.  Call after a workgroup barrier.
template <typename T, int NW, int TBM, int TBN, int MI, int NJ, int RING, typename ACC>
__device__ __forceinline__ void store_c_tile_impl(const GemmArgs& p, ACC (&acc)[MI][NJ], char* smem, int m0, int n0,
                                                  int wm, int wn, int lane, int ksplit) {
  constexpr bool B32 = sizeof(ACC) == 64;            // 32x32 blocks (f32x16) or 16x16 blocks (f32x4)
  constexpr int BLK = B32 ? 32 : 16;
  const int fr = lane & 31, fg = lane >> 5;
  // ---- epilogue: the fp32 C tile goes through the ring's LDS in NH row slabs of SLAB rows (the ring of the
  //      BKT = 32 variants is smaller than the full C tile), then out with coalesced 16-byte accesses
  constexpr int NH0 = (TBM * TBN * 4 + RING - 1) / RING;
  constexpr int NH = NH0 <= 1 ? 1 : NH0 <= 2 ? 2 : 4;         // power of two so that SLAB divides the tile
  constexpr int SLAB = TBM / NH;
  static_assert(SLAB * TBN * 4 <= RING && SLAB % 32 == 0 && MI * BLK * NW * NJ * BLK == TBM * TBN, "C slab does not fit the ring");
  float* ct = reinterpret_cast<float*>(smem);
  uint32_t amax_pk = 0u;
  constexpr int CPR = TBN / 8;                       // 8-column chunks per row
  constexpr int RPP = NW * 64 / CPR;                 // rows per pass
  const int cchunk = (threadIdx.x % CPR) * 8;
#pragma unroll 1
  for (int hs = 0; hs < NH; ++hs) {
    if (hs > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if ((wm + BLK * i) / SLAB == hs) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (B32) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              ct[((wm + 32 * i) % SLAB + (e & 3) + 8 * (e >> 2) + 4 * fg) * TBN + wn + 32 * j + fr] = acc[i][j][e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              ct[((wm + 16 * i) % SLAB + 4 * (lane >> 4) + e) * TBN + wn + 16 * j + (lane & 15)] = acc[i][j][e];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int row = threadIdx.x / CPR; row < SLAB; row += RPP) {
      const int m = m0 + hs * SLAB + row, n = n0 + cchunk;
      if (m < p.M && n < p.N) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(ct + row * TBN + cchunk);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(ct + row * TBN + cchunk + 4);
        if (p.splitk > 1) {
          float* w = p.ws + ((size_t)ksplit * p.M + m) * p.N + n;
          *reinterpret_cast<f32x4*>(w) = x0;
          *reinterpret_cast<f32x4*>(w + 4) = x1;
        } else {
          float v[8];
          v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3];
          v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
          amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n, v));
        }
      }
    }
  }
  if ((p.flags & COGV_EPI_ABSMAX) && p.splitk <= 1) {
    __syncthreads();
    const float bm = absmax_pk_block<T>(amax_pk, reinterpret_cast<uint32_t*>(smem));
    if (threadIdx.x == 0) atomic_max_nonneg(p.absmax, bm);
  }
}

template <typename T, int NW, int TBM, int TBN, int MI, int NJ, int RING>
__device__ __forceinline__ void store_c_tile(const GemmArgs& p, f32x16 (&acc)[MI][NJ], char* smem, int m0, int n0,
                                             int wm, int wn, int fr, int fg, int ksplit) {
  store_c_tile_impl<T, NW, TBM, TBN, MI, NJ, RING>(p, acc, smem, m0, n0, wm, wn, fr + 32 * fg, ksplit);
}
template <typename T, int NW, int TBM, int TBN, int MI, int NJ, int RING>
__device__ __forceinline__ void store_c_tile(const GemmArgs& p, f32x4 (&acc)[MI][NJ], char* smem, int m0, int n0,
                                             int wm, int wn, int lane, int ksplit) {
  store_c_tile_impl<T, NW, TBM, TBN, MI, NJ, RING>(p, acc, smem, m0, n0, wm, wn, lane, ksplit);
}

// waves per SIMD the register allocation must allow: BKT = 64 -> one 8-wave workgroup per CU (2);
// BKT = 32 -> two 8-wave workgroups (4) or three 4-wave workgroups (3) per CU
// Workgroups per CU the register allocation must allow (expressed as waves per SIMD):
//   ring <= 80 KiB (BKT = 32) -> two workgroups per CU (three for the small 128x128 tile), else one.
constexpr int glds_min_waves(int nw, int ring_bytes) {
  return (ring_bytes <= 53 * 1024 ? 3 : ring_bytes <= 80 * 1024 ? 2 : 1) * nw / 4;
}

// per-wave tile = (32*MI) x (32*NJ); workgroup tile = (WM*32*MI) x (WN*32*NJ); WM*WN waves
template <typename T, bool AT, bool BT, int WM, int WN, int MI, int NJ, int BKT>
__global__ __launch_bounds__(WM * WN * 64, glds_min_waves(WM * WN, 3 * (WM * 32 * MI + WN * 32 * NJ) * 2 * BKT))
void gemm_glds_kernel(const GemmArgs p) {
  constexpr int NW = WM * WN, TBM = WM * 32 * MI, TBN = WN * 32 * NJ, NST = 3;
  constexpr int A_BYTES = TBM * 2 * BKT, B_BYTES = TBN * 2 * BKT, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PER = A_BYTES / 1024 / NW, B_PER = B_BYTES / 1024 / NW;      // 1-KiB DMA pieces per wave per k-tile
  constexpr int LPT = A_PER + B_PER;
  constexpr int KS = BKT / 16;                                                 // MFMA k-steps per k-tile
  constexpr int RPP_N = 1024 / (2 * BKT);                                      // natural rows per DMA piece
  static_assert(A_PER >= 1 && B_PER >= 1, "tile too small for the wave count");
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // NST * STAGE

  uint64_t exp_t0 = 0, exp_r0 = 0;
  if (COGV_EXP & 16) { exp_t0 = __builtin_readcyclecounter(); exp_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  constexpr int GROUP_M = 4;
  const int in_group = GROUP_M * p.tiles_n;
  const int group_id = wgid / in_group;
  const int first_m = group_id * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int tile_m = first_m + (wgid % in_group) % gsz;
  const int tile_n = (wgid % in_group) / gsz;
  const int m0 = tile_m * TBM, n0 = tile_n * TBN;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (wave / WN) * (32 * MI), wn = (wave % WN) * (32 * NJ);
  const int fr = lane & 31, fg = lane >> 5;
  const int nk_total = p.K / BKT;
  const int kt0 = blockIdx.y * p.ktiles_per_split * (BK / BKT);
  const int nk = min(nk_total, kt0 + p.ktiles_per_split * (BK / BKT)) - kt0;   // >= 1 by construction

  // per-lane DMA source pointers for k-tile 0 of this split, and the per-k-tile byte stride
  const char* srcA[A_PER];
  const char* srcB[B_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int piece = i * NW + wave;
    if (!AT) {
      const int row = piece * RPP_N + lane / (BKT / 8);
      const int c = (lane % (BKT / 8)) ^ nswz<BKT>(row);
      const int gm = min(m0 + row, p.M - 1);
      srcA[i] = reinterpret_cast<const char*>(p.A) + ((size_t)gm * p.lda + (size_t)kt0 * BKT + c * 8) * 2;
    } else {
      constexpr int ROWB = TBM * 2;
      const int off = piece * 1024 + lane * 16;
      const int krow = off / ROWB, pc = (off % ROWB) >> 4;
      const int c = pc ^ ((krow & 3) << 2);
      const int col = min(m0 + c * 8, p.M - 8);
      srcA[i] = reinterpret_cast<const char*>(p.A) + ((size_t)((size_t)kt0 * BKT + krow) * p.lda + col) * 2;
    }
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int piece = i * NW + wave;
    if (!BT) {
      const int row = piece * RPP_N + lane / (BKT / 8);
      const int c = (lane % (BKT / 8)) ^ nswz<BKT>(row);
      const int gn = min(n0 + row, p.N - 1);
      srcB[i] = reinterpret_cast<const char*>(p.B) + ((size_t)gn * p.ldb + (size_t)kt0 * BKT + c * 8) * 2;
    } else {
      constexpr int ROWB = TBN * 2;
      const int off = piece * 1024 + lane * 16;
      const int krow = off / ROWB, pc = (off % ROWB) >> 4;
      const int c = pc ^ ((krow & 3) << 2);
      const int col = min(n0 + c * 8, p.N - 8);
      srcB[i] = reinterpret_cast<const char*>(p.B) + ((size_t)((size_t)kt0 * BKT + krow) * p.ldb + col) * 2;
    }
  }
  const size_t kstrideA = AT ? (size_t)BKT * p.lda * 2 : (size_t)BKT * 2;
  const size_t kstrideB = BT ? (size_t)BKT * p.ldb * 2 : (size_t)BKT * 2;
  // one LDS-DMA instruction: piece idx in [0, LPT): first the A pieces of this wave, then the B pieces
  auto issue_piece = [&](int kt, int st, int idx) {
    char* la = smem + st * STAGE;
    char* lb = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      if (idx == i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcA[i] + kt * kstrideA), (lds_void_t*)(la + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      if (idx == A_PER + i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcB[i] + kt * kstrideB), (lds_void_t*)(lb + (i * NW + wave) * 1024), 16, 0, 0);
  };
  auto issue = [&](int kt, int st) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) issue_piece(kt, st, i);
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // LDS byte addresses (stage 0) of this lane's transposing reads
  uint32_t trA[MI], trB[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) trA[i] = AT ? tr_addr<TBM * 2>(smem, wm + 32 * i, lane) : 0u;
#pragma unroll
  for (int j = 0; j < NJ; ++j) trB[j] = BT ? tr_addr<TBN * 2>(smem + A_BYTES, wn + 32 * j, lane) : 0u;

  issue(0, 0);
  issue(nk > 1 ? 1 : 0, 1);
  int st = 0;
  constexpr int PPS = (LPT + KS - 1) / KS;    // DMA pieces issued behind each MFMA group
  constexpr int DSR = (AT || BT) ? 0 : MI + NJ;     // compiler-visible LDS reads per k-step
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<LPT>();                         // tile kt has landed (the batch issued last iteration may be in flight)
    __builtin_amdgcn_s_barrier();              // ... for every wave; stage (kt+2)%3 is free again
    // Branch-free body: past the end the prefetch re-reads the last tile into a stage nobody reads any more
    // (keeps the vmcnt bookkeeping uniform and lets the compiler software-pipeline ds_read against MFMA).
    const int kpf = min(kt + 2, nk - 1);
    const int pst = st == 0 ? 2 : st - 1;
    const char* la = smem + st * STAGE;
    const char* lb = la + A_BYTES;
    const uint32_t soff = (uint32_t)(st * STAGE);
    typename HT<T>::v8 fa[2][MI], fb[2][NJ];
    TrRaw ta[MI], tb[NJ];
    u32x4 na[MI], nb[NJ];
    constexpr bool ASM_ALL = AT || BT;         // mixed kernels: every fragment read is asm-issued
    auto fetch = [&](int ks, int buf) {        // fragments of k-step ks -> fa[buf], fb[buf] (asm reads stay raw)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm + 32 * i + fr;
        if (AT) tr_issue<TBM * 2>(trA[i] + soff, ks, ta[i]);
        else if (ASM_ALL) nat_issue((uint32_t)(uintptr_t)la + nat_off<BKT>(row, 2 * ks + fg), na[i]);
        else fa[buf][i] = *reinterpret_cast<const typename HT<T>::v8*>(la + nat_off<BKT>(row, 2 * ks + fg));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wn + 32 * j + fr;
        if (BT) tr_issue<TBN * 2>(trB[j] + soff, ks, tb[j]);
        else if (ASM_ALL) nat_issue((uint32_t)(uintptr_t)lb + nat_off<BKT>(row, 2 * ks + fg), nb[j]);
        else fb[buf][j] = *reinterpret_cast<const typename HT<T>::v8*>(lb + nat_off<BKT>(row, 2 * ks + fg));
      }
    };
    auto land = [&](int buf) {                 // explicit wait + pack for the asm-issued reads
      if (ASM_ALL) {
        // one wait for everything issued by fetch(); every raw register is an in/out operand of some wait statement
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (AT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[i].lo), "+v"(ta[i].hi) : : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na[i]) : : "memory");
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (BT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[j].lo), "+v"(tb[j].hi) : : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nb[j]) : : "memory");
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (AT) fa[buf][i] = tr_pack<T>(ta[i]); else __builtin_memcpy(&fa[buf][i], &na[i], 16);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (BT) fb[buf][j] = tr_pack<T>(tb[j]); else __builtin_memcpy(&fb[buf][j], &nb[j], 16);
        }
      }
    };
    fetch(0, 0);
    land(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < KS - 1) fetch(ks + 1, nxt);
      if (AT || BT) __builtin_amdgcn_sched_barrier(0);     // keep the read issue ahead of the MFMA group
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = HT<T>::mfma32(fa[cur][i], fb[cur][j], acc[i][j]);
#pragma unroll
      for (int q2 = 0; q2 < PPS; ++q2)
        if (ks * PPS + q2 < LPT) issue_piece(kpf, pst, ks * PPS + q2);
      // pin the issue order inside this group: next fragments first (their LDS latency hides under the
      // MFMAs), DMA pieces between MFMAs.  Masks: 0x100 DS read, 0x008 MFMA, 0x010 VMEM.
      if (ks < KS - 1 && DSR > 0) {
        __builtin_amdgcn_sched_group_barrier(0x100, DSR / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, DSR - DSR / 2, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
#pragma unroll
      for (int q2 = 0; q2 < MI * NJ - 1; ++q2) {
        if (ks * PPS + q2 < LPT && q2 < PPS) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      if (ks < KS - 1 && (AT || BT)) { __builtin_amdgcn_sched_barrier(0); land(nxt); }
    }
    st = (st == 2) ? 0 : st + 1;
  }
  wait_vmcnt<0>();   // drain the (redundant) tail prefetches before the ring is reused
  __syncthreads();   // every wave is done reading the ring: reuse it for the fp32 C tile

  store_c_tile<T, NW, TBM, TBN, MI, NJ, NST * STAGE>(p, acc, smem, m0, n0, wm, wn, fr, fg, blockIdx.y);
  if ((COGV_EXP & 16) && p.out_f32 && threadIdx.x == 0 && (bid == 0 || bid == nwg - 1)) {
    const uint64_t dt = __builtin_readcyclecounter() - exp_t0, dr = __builtin_amdgcn_s_memrealtime() - exp_r0;
    __syncthreads();
    reinterpret_cast<float*>(p.C)[(size_t)m0 * p.ldc + n0] = 100.f * (float)dt / (float)dr;
  }
}

// ---- epilogue of the generation-3 kernel for one wave's 128x64 sub-tile (16x16 accumulator blocks, lane = row
//      l & 15, 4 columns at 4 (l >> 4)): transpose 8 rows at a time through the wave's private 2-KiB LDS strip into
//      "8 lanes x 16 bytes = one 128-byte line per row" order, then epilogue8.  F: compile-time flag mask
//      (-1: runtime flags / fp32 output, -2: split-K partial slab).
// pin a wave-uniform value in scalar registers: opaque to the optimiser, so it cannot be rematerialised by re-reading the
// kernel-argument segment at every use (an s_load + lgkmcnt(0) inside each of the epilogue's 32 passes otherwise)
template <typename V> __device__ __forceinline__ void pin_s(V& x) { asm volatile("" : "+s"(x)); }

template <typename T, int F>
__device__ __forceinline__ void pp64_epilogue(const GemmArgs& pg, f32x4 (&acc)[8][4], float* strip, int m_base, int n_base,
                                              int ksplit, int lane, uint32_t& amax_pk, int colsum_row,
                                              bool land_dma_first = false) {
  const int l15 = lane & 15, kb = lane >> 4;
  if ((COGV_EXP & 2048) && acc[0][0][0] != 12345.f) return;      // probe: no strip transposition either
  // the problem descriptor lives in the kernel-argument segment behind a run-time index: copy what this instance
  // uses into pinned scalar registers once
  GemmArgs p = pg;
  pin_s(p.C); pin_s(p.M); pin_s(p.N); pin_s(p.ldc);
  if (F < 0 || (F & (COGV_EPI_GELU | COGV_EPI_DGELU | COGV_EPI_MULAUX))) { pin_s(p.aux); pin_s(p.ldaux); }
  if (F < 0 || (F & COGV_EPI_DROPOUT)) { pin_s(p.seed); pin_s(p.stream_id); pin_s(p.thr16); pin_s(p.keep_scale); }
  if (F < 0) { pin_s(p.flags); pin_s(p.out_f32); pin_s(p.bias); }
  if (F == -2) pin_s(p.ws);
  const bool want_cs = (F == -1) ? ((p.flags & COGV_EPI_COLSUM) != 0 && !p.out_f32) : (F >= 0 && (F & COGV_EPI_COLSUM));
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int sr = lane >> 3, sc = lane & 7;           // read side: strip row, 8-column group
  // operands of the element-wise pipeline that live in global memory: bias once (the column group of a lane is
  // the same in every pass), the dGeLU pre-activations / the accumulate target for all 16 passes up front
  constexpr bool PRE_BIAS = F >= 0 && (F & COGV_EPI_BIAS), PRE_AUX = F >= 0 && (F & (COGV_EPI_DGELU | COGV_EPI_MULAUX)), PRE_C = F >= 0 && (F & COGV_EPI_ACCUM);
  u32x4 bias_v = {0u, 0u, 0u, 0u}, aux_v[PRE_AUX ? 16 : 1], c_v[PRE_C ? 16 : 1];
  const int n = n_base + 8 * sc;
  {
    if (PRE_BIAS && n < p.N) bias_v = gload16(reinterpret_cast<const T*>(pg.bias) + n);
    if (PRE_AUX || PRE_C) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int m = m_base + 8 * t + sr;
        const bool ok = m < p.M && n < p.N;
        if (PRE_AUX) aux_v[t] = ok ? gload16(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldaux + n) : u32x4{0u, 0u, 0u, 0u};
        if (PRE_C) c_v[t] = ok ? gload16(reinterpret_cast<const T*>(p.C) + (size_t)m * p.ldc + n) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  // Pass t moves the 8 rows 16 (t >> 1) + 8 (t & 1) .. +7 through the strip.  The LDS unit executes one wave's
  // instructions in order, so the writes of pass t + 1 may be issued right behind the reads of pass t: the strip is
  // software-pipelined one pass deep (reads of t + 1 in flight while pass t runs its element-wise chain and store).
  auto put = [&](int t) {
    if ((l15 >> 3) == (t & 1)) {
      const int r = l15 & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(strip + r * 64 + (((4 * j + kb) ^ r) << 2)) = acc[t >> 1][j];
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto get = [&](f32x4& x0, f32x4& x1) {
    x0 = *reinterpret_cast<const f32x4*>(strip + sr * 64 + (((2 * sc) ^ sr) << 2));
    x1 = *reinterpret_cast<const f32x4*>(strip + sr * 64 + (((2 * sc + 1) ^ sr) << 2));
    __builtin_amdgcn_wave_barrier();
  };
  f32x4 xq[2][2];
  put(0);
  get(xq[0][0], xq[0][1]);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    if (t + 1 < 16) {
      put(t + 1);
      get(xq[(t + 1) & 1][0], xq[(t + 1) & 1][1]);
    }
    const f32x4 x0 = xq[t & 1][0], x1 = xq[t & 1][1];
    int m = m_base + 8 * t + sr;
    if ((COGV_EXP & 64) && x0[0] != 12345.f) continue;      // probe: no epilogue
    if (COGV_EXP & 128) m &= 255;                           // probe: all tiles store to the same L2-resident rows
    // (generation-4 kernel) the next item's prologue DMAs, issued in front of this epilogue, are waited for
    // in front of its FIRST store: behind it a vmcnt wait would also have to wait for stores
    if (t == 0 && land_dma_first) wait_vmcnt<0>();
    if (m < p.M && n < p.N) {
      if (F == -2) {                                        // split-K partial: raw fp32 slab
        float* w = p.ws + ((size_t)ksplit * p.M + m) * p.N + n;
        gstore16(w, x0);
        gstore16(w + 4, x1);
      } else {
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        float rv[8];
        amax_pk = absmax_pk(amax_pk, epilogue8<T, (F < 0 ? -1 : F)>(p, m, n, v, PRE_BIAS ? &bias_v : nullptr,
                                                                     PRE_AUX ? &aux_v[t] : nullptr,
                                                                     PRE_C ? &c_v[t] : nullptr, want_cs ? rv : nullptr));
        if (want_cs) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[e] += rv[e];
        }
      }
    }
  }
  if (want_cs) {     // lanes with the same (lane & 7) hold the same 8 columns: fold the 8 strip rows, lanes 0..7 write
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = cs[e];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
      cs[e] = t;
    }
    if (sr == 0 && n < p.N) {
      float* w = pg.colsum_ws + (size_t)colsum_row * p.N + n;
      gstore16(w, f32x4{cs[0], cs[1], cs[2], cs[3]});
      gstore16(w + 4, f32x4{cs[4], cs[5], cs[6], cs[7]});
    }
  }
}

// =====================================================================================================
// Generation-3 kernel: 256x256 tile, 64-deep k-tiles, 8 waves (2 x 4, 128x64 each), ping-pong schedule,
// v_mfma_f32_16x16x32, persistent over (tile, k-split) items of up to four problems.
//
// Why this shape (measured with tools/probes/gemm_exp.py, which compiles the loop with parts removed):
//  * the chip is POWER limited in a dense GEMM: the shader clock falls from 2.4 GHz to 1.3-1.8 GHz as soon as
//    MFMA, LDS reads and LDS-DMA run together, so throughput follows energy per flop.  The 16x16x32 MFMA moves
//    half the accumulator bytes per flop of the 32x32x16 one and measured +15 % on the whole loop;
//  * a K-contiguous operand must arrive as 128-byte row segments (64-deep k-tiles): with 32-deep tiles every
//    L2 request is half a line and the global->LDS stream alone cannot keep up (11.9 vs 18.3 TB/s chip-wide).
// A 64-deep 256x256 k-tile is 64 KiB, so only two fit in LDS -- a whole-tile ring would have prefetch distance
// one.  Instead the k-tile is cut into three granules with different deadlines and the two half-steps of a k-tile
// read DIFFERENT data (quadrant order), so every granule is resident for exactly one READ phase and the prefetch
// distance is three half-steps for all of them:
//
//   granule   content (64 k deep)                      bytes   read in      re-issued (for tile)   needed
//   B         all 256 B rows                           32 KiB  R(2T)        R(2T+1)  (T+2)          R(2T+4)
//   A01       A rows [0,64) u [128,192)                16 KiB  R(2T)        R(2T+1)  (T+2)          R(2T+4)
//   A23       A rows [64,128) u [192,256)              16 KiB  R(2T+1)      R(2T+2)  (T+2)          R(2T+5)
//
//   half-step 2T  : READ  B fragments of the whole k-tile (kept in registers for both half-steps) + A01 fragments,
//                   issue A23(T+1);          MFMA acc[0..3][*] += A01 x B   (32 MFMAs, 2 k-steps of 32)
//   half-step 2T+1: READ  A23 fragments, issue B(T+2), A01(T+2);  MFMA acc[4..7][*] += A23 x B
//
// LDS: 2 buffers x (B 32 KiB | A01 16 KiB | A23 16 KiB) = 128 KiB.  Two barriers per half-step; waves 0-3 and
// 4-7 (one of each per SIMD) run one barrier apart, so one wave of every SIMD is in its MFMA phase while the
// other reads/issues.  Ordering: a granule issued in R(h) replaces data whose last reads were retired
// (lgkmcnt(0)) before every wave's B2(h-1); a granule needed in R(h+1) is certified by every wave's counted vmcnt
// before its B2(h).  Each wave always has exactly 8 DMA instructions issued after the ones it must certify
// (6 + 2), so the wait is vmcnt(8) in both half-steps.
//
// The MFMA operands are SWAPPED (D = B_frag x A_frag), so a lane ends up with 4 consecutive COLUMNS of one
// output row (one ds_write_b128 per 16x16 block), and the epilogue transposes through a private 2-KiB LDS strip
// per wave without any workgroup barrier.  That leaves the ring free after the last READ phase: the NEXT item's
// first 1.75 k-tiles are issued before the epilogue of the current one, so their latency and the draining C
// stores overlap.
template <typename T, bool AT, bool BT>
__global__ __launch_bounds__(512, 2)
void gemm_pp64_kernel(const GroupArgs ga) {
  constexpr int NW = 8, TBM = 256, TBN = 256, KT = 64, KS = 2;
  constexpr int B_OFF = 0, A01_OFF = 32768, A23_OFF = 49152, BUF = 65536;
  constexpr int ROWB_A = 256, ROWB_B = 512;            // k-row bytes of a contraction-strided granule
  extern __shared__ __attribute__((aligned(1024))) char smem[];     // 2 * BUF

  uint64_t exp_t0 = 0, exp_r0 = 0;
  if (COGV_EXP & 16) { exp_t0 = __builtin_readcyclecounter(); exp_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int wm = wr * 128, wn = wc * 64;
  const int l15 = lane & 15, kb = lane >> 4;
  const int nitems = ga.item_start[ga.count];

  // ---- everything that depends on the work item: which problem, which tile, which k range, DMA sources
  struct Item {
    int pi, m0, n0, ksplit, kt0, nk;
    uint32_t offB[4], offA[2][2];      // per-lane byte offsets against a wave-uniform base (SGPR-base DMA form)
  };
  auto setup = [&](int item, Item& it) {
    int pi = 0;
#pragma unroll
    for (int t = 1; t < MAX_GROUP; ++t) pi += (t < ga.count && item >= ga.item_start[t]) ? 1 : 0;
    const GemmArgs& p = ga.g[pi];
    const int local = item - ga.item_start[pi];
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = local % nwg;
    it.pi = pi; it.ksplit = local / nwg;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    constexpr int GROUP_M = 4;
    const int in_group = GROUP_M * p.tiles_n;
    const int group_id = wgid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (wgid % in_group) % gsz;
    const int tile_n = (wgid % in_group) / gsz;
    const int m0 = tile_m * TBM, n0 = tile_n * TBN;
    it.m0 = m0; it.n0 = n0;
    it.kt0 = it.ksplit * p.ktiles_per_split;
    it.nk = min(p.K / KT, it.kt0 + p.ktiles_per_split) - it.kt0;       // >= 1 by construction
    // Piece = one 1-KiB LDS-DMA instruction; wave w owns pieces i*8 + w.  B: 32 pieces (4 per wave); A01, A23: 16 each.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = i * NW + wave;
      if (!BT) {
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(row);
        const int gn = min(n0 + row, p.N - 1);
        it.offB[i] = (uint32_t)(((size_t)gn * p.ldb + c * 8) * 2);
      } else {
        const int off = piece * 1024 + lane * 16;
        const int krow = off / ROWB_B, pc = (off % ROWB_B) >> 4;
        const int c = pc ^ trswz16(krow);
        const int col = min(n0 + c * 8, p.N - 8);
        it.offB[i] = (uint32_t)(((size_t)krow * p.ldb + col) * 2);
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int piece = i * NW + wave;
        if (!AT) {
          const int row = piece * 8 + (lane >> 3);                 // granule row 0..127
          const int c = (lane & 7) ^ swz(row);
          const int tr = (row & 63) + 128 * (row >> 6) + 64 * g;   // tile row
          const int gm = min(m0 + tr, p.M - 1);
          it.offA[g][i] = (uint32_t)(((size_t)gm * p.lda + c * 8) * 2);
        } else {
          const int off = piece * 1024 + lane * 16;
          const int krow = off / ROWB_A, pc = (off % ROWB_A) >> 4;
          const int c = pc ^ trswz16(krow);
          const int gc = c * 8;                                    // granule column 0..127
          const int tcol = (gc & 63) + 128 * (gc >> 6) + 64 * g;   // tile row (= column of the stored A)
          const int col = min(m0 + tcol, p.M - 8);
          it.offA[g][i] = (uint32_t)(((size_t)krow * p.lda + col) * 2);
        }
      }
  };
  auto issue_B = [&](const Item& it, int kt, int buf) {
    if (COGV_EXP & 1) return;
    if (COGV_EXP & 8) kt &= 3;              // re-read the first k-tiles: every request an L2 hit
    const GemmArgs& p = ga.g[it.pi];
    const size_t kstride = BT ? (size_t)KT * p.ldb * 2 : (size_t)KT * 2;
    const char* g = reinterpret_cast<const char*>(p.B) + (size_t)(it.kt0 + kt) * kstride;
    char* l = smem + buf * BUF + B_OFF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (COGV_EXP & 256) {        // probe: the same request stream into VGPRs (discarded) instead of LDS
        u32x4 t; const char* a = g + it.offB[i];
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(a) : "memory");
        continue;
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(g + it.offB[i]), (lds_void_t*)(l + (i * NW + wave) * 1024), 16, 0, 0);
    }
  };
  auto issue_A = [&](const Item& it, int gi, int kt, int buf) {
    if (COGV_EXP & 1) return;
    if (COGV_EXP & 8) kt &= 3;
    const GemmArgs& p = ga.g[it.pi];
    const size_t kstride = AT ? (size_t)KT * p.lda * 2 : (size_t)KT * 2;
    const char* g = reinterpret_cast<const char*>(p.A) + (size_t)(it.kt0 + kt) * kstride;
    char* l = smem + buf * BUF + (gi ? A23_OFF : A01_OFF);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (COGV_EXP & 256) {
        u32x4 t; const char* a = g + it.offA[gi][i];
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(a) : "memory");
        continue;
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(g + it.offA[gi][i]), (lds_void_t*)(l + (i * NW + wave) * 1024), 16, 0, 0);
    }
  };
  // tile 0 complete + B, A01 of tile 1: 14 DMA instructions per wave, in the order the k-loop certifies them
  auto prologue = [&](const Item& it) {
    issue_B(it, 0, 0); issue_A(it, 0, 0, 0);
    issue_A(it, 1, 0, 0);
    const int t1 = min(1, it.nk - 1);
    issue_B(it, t1, 1); issue_A(it, 0, t1, 1);
  };

  // ---- per-lane fragment read addresses (buffer 0, k-step 0).  v_mfma_f32_16x16x32: lane l supplies row
  //      (l & 15) of a 16-row block and the 8 contraction slots of k-block (l >> 4).
  uint32_t adB[4], adA[4];           // adA is relative to the A granule (A01 and A23 share the in-granule layout)
#pragma unroll
  for (int j = 0; j < 4; ++j)
    adB[j] = BT ? tr_addr16<ROWB_B>(smem + B_OFF, wn + 16 * j, lane)
                : (uint32_t)(uintptr_t)(smem + B_OFF) + (uint32_t)((wn + 16 * j + l15) * 128 + ((kb ^ swz(wn + 16 * j + l15)) << 4));
#pragma unroll
  for (int i = 0; i < 4; ++i)
    adA[i] = AT ? tr_addr16<ROWB_A>(smem, wr * 64 + 16 * i, lane)
                : (uint32_t)(uintptr_t)smem + (uint32_t)((wr * 64 + 16 * i + l15) * 128 + ((kb ^ swz(wr * 64 + 16 * i + l15)) << 4));

  // Work distribution: every item, the first one included, comes from an atomic counter.  A static assignment
  // would make the launch as slow as its unluckiest workgroup: this kernel needs a whole CU (512 threads x 256
  // registers), so when other kernels hold CUs -- RCCL's all-reduce channels during the data-parallel backward --
  // some workgroups start late; with the queue they take fewer items, or none and exit at once.
  // One queue per XCD (workgroup b runs on XCD b & 7): item i stays on XCD i & 7, which is what the tile order
  // inside setup() assumes for L2 reuse (one shared queue measured 10-15 % slower).
  __shared__ int s_next;
  const int xq = blockIdx.x & 7;
  if (threadIdx.x == 0) s_next = xq + 8 * atomicAdd(ga.sched + xq, 1);
  __syncthreads();
  Item cur;
  int item = s_next;
  if (item < nitems) { setup(item, cur); prologue(cur); }
#pragma unroll 1
  while (item < nitems) {
    const GemmArgs& p = ga.g[cur.pi];
    const int nk = cur.nk;
    f32x4 acc[8][4];                   // 16x16 blocks of this wave's 128x64: acc[row block][column block]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Outstanding per wave, oldest first: [C stores of the previous item] [B(0) A01(0): 6] [A23(0): 2] [B(1) A01(1): 6].
    // Loads retire in order among loads, so "at most 8 outstanding" means the first 6 have landed (and every store).
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();             // the stagger
    int grabbed = 0;                                       // the item after this one: asked for now, used after the k-loop
    if (threadIdx.x == 0) grabbed = atomicAdd(ga.sched + xq, 1);

    TrRaw tb[KS][4], ta[KS][4];          // KS = 2 k-steps of 32 per k-tile
    u32x4 nb[KS][4], na[KS][4];
    auto read_B = [&](uint32_t boff) {
      if (COGV_EXP & 2) return;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BT) tr_issue16<ROWB_B>(adB[j] + boff, ks, tb[ks][j]);
          else nat_issue((adB[j] + boff) ^ (uint32_t)(ks << 6), nb[ks][j]);
        }
    };
    auto read_A = [&](uint32_t goff) {
      if (COGV_EXP & 2) return;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AT) tr_issue16<ROWB_A>(adA[i] + goff, ks, ta[ks][i]);
          else nat_issue((adA[i] + goff) ^ (uint32_t)(ks << 6), na[ks][i]);
        }
    };
    auto land_A = [&]() {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[ks][i].lo), "+v"(ta[ks][i].hi) : : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na[ks][i]) : : "memory");
        }
    };
    auto land_B = [&]() {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[ks][j].lo), "+v"(tb[ks][j].hi) : : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nb[ks][j]) : : "memory");
        }
    };
    auto mma = [&](int half) {
      if (!(COGV_EXP & 512)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        typename HT<T>::v8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AT) fa[i] = tr_pack<T>(ta[ks][i]); else __builtin_memcpy(&fa[i], &na[ks][i], 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BT) fb[j] = tr_pack<T>(tb[ks][j]); else __builtin_memcpy(&fb[j], &nb[ks][j], 16);
        }
        if (COGV_EXP & 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[i]));
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(fb[j]));
          continue;
        }
        // operands swapped: D[n][m] -> lane (m = l & 15) holds columns n = 4 (l >> 4) .. +3 of its row
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[4 * half + i][j] = HT<T>::mfma16(fb[j], fa[i], acc[4 * half + i][j]);
      }
      if (!(COGV_EXP & 512)) __builtin_amdgcn_s_setprio(0);
    };

    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const uint32_t boff = (uint32_t)(buf * BUF);
      // ---------------- half-step 2T
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_B(boff);
      read_A(boff + A01_OFF);
      issue_A(cur, 1, min(kt + 1, nk - 1), buf ^ 1);
      land_B(); land_A();
      wait_vmcnt<8>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      // ---------------- half-step 2T + 1
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_A(boff + A23_OFF);
      { const int t2 = min(kt + 2, nk - 1); issue_B(cur, t2, buf); issue_A(cur, 0, t2, buf); }
      land_A();
      wait_vmcnt<8>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 0) __builtin_amdgcn_s_barrier();              // even out the barrier count
    // Every wave has passed its last READ phase here (group 1's final B2 is the barrier above): the ring is free.
    wait_vmcnt<0>();                                        // the (redundant) tail prefetches of this item
    __builtin_amdgcn_sched_barrier(0);

    // ---- next item's prologue goes out BEFORE this item's epilogue
    if (threadIdx.x == 0) s_next = xq + 8 * grabbed;       // the grabbed-th item of this XCD's list {x, x + 8, ...}
    __syncthreads();
    const int next = s_next;
    const Item done = cur;
    if (next < nitems) { setup(next, cur); prologue(cur); }

    // ---- epilogue.  The accumulators hold, per lane, 4 consecutive columns of row (l & 15) of each 16x16 block.
    //      Each wave transposes its own 128x64 sub-tile through a PRIVATE 2-KiB strip of LDS (8 rows x 64 fp32
    //      columns at a time; the A23 slot of buffer 1, which the next item's prologue does not touch) into
    //      "8 lanes x 16 bytes = one 128-byte line per row" order for the fused epilogue8: no workgroup barrier,
    //      full-line stores.  (Storing straight from the MFMA layout -- 8 bytes per lane, 32-byte row segments --
    //      measured 4x slower than this: 16 us per tile.)  16-byte chunk c of strip row r sits at chunk c ^ r.
    uint32_t amax_pk = 0u;
    float* strip = reinterpret_cast<float*>(smem + BUF + A23_OFF + wave * 2048);
    // One instance per hot flag combination (compile-time mask): the passes below are fully unrolled (the
    // accumulators need static register indices), so a single runtime-flag body is ~100 KB of code per kernel
    // and every item would stream it through the instruction cache.
    constexpr int F_FWD_DROP = COGV_EPI_BIAS | COGV_EPI_DROPOUT | COGV_EPI_ABSMAX, F_FWD_GELU = COGV_EPI_BIAS | COGV_EPI_GELU;
    if (p.splitk > 1) pp64_epilogue<T, -2>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.out_f32) pp64_epilogue<T, -1>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == 0) pp64_epilogue<T, 0>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == COGV_EPI_BIAS) pp64_epilogue<T, COGV_EPI_BIAS>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == COGV_EPI_ACCUM) pp64_epilogue<T, COGV_EPI_ACCUM>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == F_FWD_DROP) pp64_epilogue<T, F_FWD_DROP>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == F_FWD_GELU) pp64_epilogue<T, F_FWD_GELU>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == (COGV_EPI_DGELU | COGV_EPI_COLSUM)) pp64_epilogue<T, COGV_EPI_DGELU | COGV_EPI_COLSUM>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == COGV_EPI_DGELU) pp64_epilogue<T, COGV_EPI_DGELU>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == (F_FWD_GELU | COGV_EPI_GELU_DAUX)) pp64_epilogue<T, F_FWD_GELU | COGV_EPI_GELU_DAUX>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == (COGV_EPI_MULAUX | COGV_EPI_COLSUM)) pp64_epilogue<T, COGV_EPI_MULAUX | COGV_EPI_COLSUM>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else pp64_epilogue<T, -1>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    if ((p.flags & COGV_EPI_ABSMAX) && p.splitk <= 1) {
      uint32_t wv = max(amax_pk & 0xffffu, amax_pk >> 16);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wv = max(wv, (uint32_t)__shfl_xor((int)wv, o, 64));
      if (lane == 0) atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
    if ((COGV_EXP & 16) && p.out_f32 && threadIdx.x == 0) {
      // shader clock in MHz over this workgroup's lifetime so far (s_memrealtime ticks at 100 MHz)
      const uint64_t dt = __builtin_readcyclecounter() - exp_t0, dr = __builtin_amdgcn_s_memrealtime() - exp_r0;
      reinterpret_cast<float*>(p.C)[(size_t)done.m0 * p.ldc + done.n0] = 100.f * (float)dt / (float)dr;
    }
    item = next;
  }
  // the last workgroup to leave re-arms the queue for the next launch (every workgroup has made its last grab by then)
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ga.sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t) ga.sched[t] = 0;
      __threadfence();
    }
  }
}

// =====================================================================================================
// Generation-4 kernel: the same 256x256x64 tile, persistent queues, LDS-DMA granule ring and epilogue as
// generation 3, but FOUR waves of 128x128 (one per SIMD, accumulators in the 256 AGPRs) instead of eight of
// 128x64.  A 128x128 wave tile reads 256 B of fragments per MFMA instead of 384 B: the GEMM is power limited
// and the probe (tools/probes/gemm_exp.py) puts the fragment reads at ~19 % of the loop's cost, so fewer LDS
// bytes per flop is the lever -- at the price of no second wave per SIMD to hide latency: the next fragments
// are read between the MFMAs of the same wave (software pipeline, order pinned by sched_barrier).
//
// k-tile = 4 granules of 16 KiB (128 rows or columns x 64 k): A01 = tile rows [0,64) u [128,192), A23 = the
// rest, B01 / B23 likewise over the tile's columns; wave (wr, wc) owns granule rows wr*64.. of the A granules
// and wc*64.. of the B granules.  A k-tile is four quarter-steps of 32 MFMAs, each against one A half and one
// B half, ordered so that only ONE half changes between consecutive quarter-steps; its 8 fragments are read
// during the previous quarter-step into the register set that just died (4 sets of 32 registers):
//
//   q0  A01 x B01   reads B23(T)          | half-step 2T  : DMA A01(T+2), B01(T+2)
//   q1  A01 x B23   reads A23(T)          |
//   q2  A23 x B23   reads A01(T+1)        | half-step 2T+1: DMA B23(T+2), A23(T+2)
//   q3  A23 x B01   reads B01(T+1)        |
//
// One barrier per half-step.  The granules read during half-step h were issued in half-step h-3 and are
// certified by every wave's vmcnt(16) (8 DMA instructions per wave per half-step) before the barrier that opens
// h; the granules read during h-1 are free from that barrier on and are re-issued (for two k-tiles later) in h.
// LDS: A granules in [0, 64 KiB) at buffer * 32 KiB + half * 16 KiB, B granules likewise in [64, 128 KiB), then 4 private
// 2-KiB epilogue strips.  Every fragment read is "per-lane register + immediate" (the buffer / granule / k-step part
// fits the 16-bit offset field), so the loop holds 16 address registers instead of one per (buffer, granule, block).
template <int OFF>
__device__ __forceinline__ void nat_issue_o(uint32_t addr, u32x4& o) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF, int ROWB>
__device__ __forceinline__ void tr_issue16_o(uint32_t addr, TrRaw& o) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(o.lo), "=&v"(o.hi) : "v"(addr), "n"(OFF), "n"(OFF + 4 * ROWB) : "memory");
}
// MFMA with the accumulator pinned to the AGPR file and updated in place.  With all 256 AGPRs holding accumulators the
// register allocator has no slack: left to itself (builtin form) it parks parts of the loop-carried accumulators in
// VGPRs, picks untied destination registers and copies / spills around every MFMA.
template <typename T>
__device__ __forceinline__ void mfma16_inplace(f32x4& c, const typename HT<T>::v8& a, const typename HT<T>::v8& b) {
  if constexpr (std::is_same<T, bf16_t>::value)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// LDS-DMA of 16 bytes per lane in the "SGPR base + 32-bit lane offset" form, LDS destination (wave-uniform) through M0
__device__ __forceinline__ void dma16(const void* base, uint32_t lane_off, uint32_t lds_addr) {
  const uint64_t b64 = (uint64_t)(uintptr_t)base;
  const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
  base = reinterpret_cast<const void*>((uintptr_t)bu);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(base), "s"(lds_addr) : "memory", "m0");
}
template <int V> using IC = std::integral_constant<int, V>;
template <typename F, int... R>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, R...>) { (f(IC<R>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- epilogue of the generation-4 kernel (round 4): the same 8-rows-per-store strip transposition and epilogue8 chain as
//      pp64_epilogue, but with the LDS traffic issued BY HAND as a pipeline.  The compiler-scheduled form above waits, in
//      every pass, for the pass's own strip writes and for the previous pass's reads before it issues the next reads
//      (s_waitcnt lgkmcnt(3) x 2 in front of the two ds_read_b128, lgkmcnt(3 / 2) in front of the arithmetic): two exposed LDS
//      round trips per 8-row pass, ~310 cycles for ~22 instructions, whatever the shader clock -- measured with in-kernel
//      timestamps (profiles/r04_gemm_item_phase_probe_v1.log): 2.5-3.9 us per 128 x 64 half in the plain / bias forms, 9 % of a
//      K = 2560 item with the matrix pipe idle.  Here a "super-pass" moves a whole 16-row block: ALL lanes write their four
//      16 x 16 blocks (rows 0-7 into the wave's strip A, rows 8-15 into strip B: 4 KiB per wave, no exec-masked half), the
//      four reads of the block follow at once, and the writes + reads of block T + 1 are issued BEFORE block T's registers are
//      awaited with a counted lgkmcnt(8) -- the LDS unit executes a wave's instructions in order, so the reads of T see T's
//      data and the writes of T + 1 cannot overtake them.  Accumulators are written straight from the AGPRs.
__device__ __forceinline__ void w4_strip_put(const uint32_t (&wa)[4], f32x4 (&blk)[4]) {       // four 16 x 16 blocks, AGPR -> LDS
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("ds_write_b128 %0, %1" : : "v"(wa[j]), "a"(blk[j]) : "memory");
}
__device__ __forceinline__ void w4_strip_get(uint32_t ra0, uint32_t ra1, f32x4 (&x)[4]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %5 offset:2048"
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(ra0), "v"(ra1) : "memory");
}
template <int N>
__device__ __forceinline__ void w4_strip_land(f32x4 (&x)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N) : "memory");
}
template <typename T, int F>
__device__ __forceinline__ void w4_epilogue(const GemmArgs& pg, f32x4 (&acc)[8][4], char* strip, int m_base, int n_base,
                                            int ksplit, int lane, uint32_t& amax_pk, int colsum_row, bool land_dma_first) {
  const int l15 = lane & 15, kb = lane >> 4;
  if ((COGV_EXP & 2048) && lane == 65) return;                    // probe: no epilogue at all
  GemmArgs p = pg;
  pin_s(p.C); pin_s(p.M); pin_s(p.N); pin_s(p.ldc);
  if (F < 0 || (F & (COGV_EPI_GELU | COGV_EPI_DGELU | COGV_EPI_MULAUX))) { pin_s(p.aux); pin_s(p.ldaux); }
  if (F < 0 || (F & COGV_EPI_DROPOUT)) { pin_s(p.seed); pin_s(p.stream_id); pin_s(p.thr16); pin_s(p.keep_scale); }
  if (F < 0) { pin_s(p.flags); pin_s(p.out_f32); pin_s(p.bias); }
  if (F == -2) pin_s(p.ws);
  const bool want_cs = (F == -1) ? ((p.flags & COGV_EPI_COLSUM) != 0 && !p.out_f32) : (F >= 0 && (F & COGV_EPI_COLSUM));
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int sr = lane >> 3, sc = lane & 7;           // read side: strip row, 8-column group
  constexpr bool PRE_BIAS = F >= 0 && (F & COGV_EPI_BIAS), PRE_AUX = F >= 0 && (F & (COGV_EPI_DGELU | COGV_EPI_MULAUX)), PRE_C = F >= 0 && (F & COGV_EPI_ACCUM);
  u32x4 bias_v = {0u, 0u, 0u, 0u}, aux_v[PRE_AUX ? 16 : 1], c_v[PRE_C ? 16 : 1];
  const int n = n_base + 8 * sc;
  {
    if (PRE_BIAS && n < p.N) bias_v = gload16(reinterpret_cast<const T*>(pg.bias) + n);
    if (PRE_AUX || PRE_C) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int m = m_base + 8 * t + sr;
        const bool ok = m < p.M && n < p.N;
        if (PRE_AUX) aux_v[t] = ok ? gload16(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldaux + n) : u32x4{0u, 0u, 0u, 0u};
        if (PRE_C) c_v[t] = ok ? gload16(reinterpret_cast<const T*>(p.C) + (size_t)m * p.ldc + n) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  // write side: lane (row l15, column quad kb) of block j -> strip (l15 >> 3), row r = l15 & 7, 16-byte chunk (4 j + kb) ^ r
  const uint32_t sbase = (uint32_t)(uintptr_t)strip;
  uint32_t wa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wa[j] = sbase + (uint32_t)((l15 >> 3) * 2048 + (l15 & 7) * 256 + ((((4 * j + kb) ^ (l15 & 7))) << 4));
  const uint32_t ra0 = sbase + (uint32_t)(sr * 256 + (((2 * sc) ^ sr) << 4)), ra1 = sbase + (uint32_t)(sr * 256 + (((2 * sc + 1) ^ sr) << 4));
  f32x4 xq[2][4];                                    // [register set][strip * 2 + {columns 0-3, 4-7}]
  // (F == -1, the run-time-flag instance -- fp32 output, rare flag combinations: its element-wise chain keeps a small
  //  array in scratch, and the build refuses scratch traffic while asm-issued loads are in flight: no read-ahead there)
  constexpr bool AHEAD = F != -1;
  if (AHEAD) {
    w4_strip_put(wa, acc[0]);
    w4_strip_get(ra0, ra1, xq[0]);
  }
#pragma unroll
  for (int TT = 0; TT < 8; ++TT) {
    if (!AHEAD) {
      w4_strip_put(wa, acc[TT]);
      w4_strip_get(ra0, ra1, xq[TT & 1]);
      w4_strip_land<0>(xq[TT & 1]);
    } else if (TT + 1 < 8) {
      w4_strip_put(wa, acc[TT + 1]);
      w4_strip_get(ra0, ra1, xq[(TT + 1) & 1]);
      w4_strip_land<8>(xq[TT & 1]);
    } else {
      w4_strip_land<0>(xq[TT & 1]);
    }
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
      const int t = 2 * TT + hs;
      const f32x4 x0 = xq[TT & 1][2 * hs], x1 = xq[TT & 1][2 * hs + 1];
      int m = m_base + 8 * t + sr;
      if ((COGV_EXP & 64) && x0[0] != 12345.f) continue;      // probe: no epilogue arithmetic / stores
      if (COGV_EXP & 128) m &= 255;                           // probe: all tiles store to the same L2-resident rows
      // the next item's prologue DMAs, issued in front of this epilogue, are waited for in front of its FIRST store: behind
      // it a vmcnt wait would also have to wait for stores
      if (t == 0 && land_dma_first) wait_vmcnt<0>();
      if (m < p.M && n < p.N) {
        if (F == -2) {                                        // split-K partial: raw fp32 slab
          float* w = p.ws + ((size_t)ksplit * p.M + m) * p.N + n;
          gstore16(w, x0);
          gstore16(w + 4, x1);
        } else {
          float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          float rv[8];
          amax_pk = absmax_pk(amax_pk, epilogue8<T, (F < 0 ? -1 : F)>(p, m, n, v, PRE_BIAS ? &bias_v : nullptr,
                                                                       PRE_AUX ? &aux_v[t] : nullptr,
                                                                       PRE_C ? &c_v[t] : nullptr, want_cs ? rv : nullptr));
          if (want_cs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] += rv[e];
          }
        }
      }
    }
  }
  if (want_cs) {     // lanes with the same (lane & 7) hold the same 8 columns: fold the 8 strip rows, lanes 0..7 write
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = cs[e];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
      cs[e] = t;
    }
    if (sr == 0 && n < p.N) {
      float* w = pg.colsum_ws + (size_t)colsum_row * p.N + n;
      gstore16(w, f32x4{cs[0], cs[1], cs[2], cs[3]});
      gstore16(w + 4, f32x4{cs[4], cs[5], cs[6], cs[7]});
    }
  }
}

// Probe builds only (-DCOGV_W4_TS, tools/probes/w4_ts.py): per-wave wall time (s_memrealtime, 100 MHz) of the phases of an
// item, summed over the wave's items and written over the first 32 KiB of problem 0's C when the workgroup exits -- the
// output of such a build is garbage there by design.  Phases: 0 first wait + barrier, 1 pre-step (queue atomic, first
// fragments), 2 k-loop, 3 drain wait + item hand-over barrier, 4 next item's setup + prologue issue, 5 / 6 the two epilogue
// halves; slot 7 counts items, slots 8 / 9 are the kernel's total in s_memrealtime / s_memtime ticks (-> shader clock).
#if defined(COGV_W4_TS)
#define W4_TS(k_) do { const uint64_t t_ = __builtin_amdgcn_s_memrealtime(); ts_acc[k_] += (uint32_t)(t_ - ts_last); ts_last = t_; } while (0)
#else
#define W4_TS(k_) do { } while (0)
#endif

template <typename T, bool AT, bool BT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_w4_kernel(const GroupArgs ga) {
  constexpr int NW = 4, TBM = 256, TBN = 256, KT = 64;
  constexpr int GRAN = 16384, BUF = 32768, BREG = 65536, ROWB = 256;   // BUF: buffer stride inside an operand's region
  constexpr int G_A01 = 0, G_A23 = 1, G_B01 = 2, G_B23 = 3;
  extern __shared__ __attribute__((aligned(1024))) char smem[];     // 2 * BREG + NW * 4096

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, kb = lane >> 4;
  const int nitems = ga.item_start[ga.count];
  const uint32_t smem_u32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)smem);

  struct Item {
    int pi, m0, n0, ksplit, kt0, nk;
    const char* baseA; const char* baseB;   // wave-uniform: operand + tile origin + the item's first k-tile
    uint32_t off[4][4];                // [granule][piece]: per-lane byte offsets RELATIVE to the tile origin (clamped at the
                                       // matrix edge; every interior tile of a problem has the same 16 values)
  };
  // tile origin and first k-tile of an item (64-bit scalar arithmetic)
  auto set_bases = [&](const GemmArgs& p, Item& it) {
    const size_t ksA = AT ? (size_t)KT * p.lda * 2 : (size_t)KT * 2, ksB = BT ? (size_t)KT * p.ldb * 2 : (size_t)KT * 2;
    it.baseA = reinterpret_cast<const char*>(p.A) + (AT ? (size_t)it.m0 * 2 : (size_t)it.m0 * p.lda * 2) + (size_t)it.kt0 * ksA;
    it.baseB = reinterpret_cast<const char*>(p.B) + (BT ? (size_t)it.n0 * 2 : (size_t)it.n0 * p.ldb * 2) + (size_t)it.kt0 * ksB;
  };
  auto setup = [&](int item, Item& it) {
    int pi = 0;
    if (ga.count > 1) {
#pragma unroll
      for (int t = 1; t < MAX_GROUP; ++t) pi += (t < ga.count && item >= ga.item_start[t]) ? 1 : 0;
    }
    const GemmArgs& p = ga.g[pi];
    const int local = item - ga.item_start[pi];
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = local % nwg;
    it.pi = pi; it.ksplit = local / nwg;
    uint32_t tile_m, tile_n;
    w4_tile_slow((uint32_t)bid, (uint32_t)p.tiles_m, (uint32_t)p.tiles_n, (uint32_t)ga.group_m, tile_m, tile_n);
    // (the integer divisions above run on the VALU: bring the wave-uniform results back to SGPRs, so that the k-loop
    //  counter, the DMA base pointers and the LDS destinations stay scalar)
    it.ksplit = __builtin_amdgcn_readfirstlane(it.ksplit);
    it.m0 = __builtin_amdgcn_readfirstlane((int)tile_m * TBM); it.n0 = __builtin_amdgcn_readfirstlane((int)tile_n * TBN);
    it.kt0 = it.ksplit * p.ktiles_per_split;
    it.nk = min(p.K / KT, it.kt0 + p.ktiles_per_split) - it.kt0;       // >= 1 by construction
    set_bases(p, it);
    // Piece = one 1-KiB LDS-DMA instruction; a granule is 16 pieces, wave w owns pieces i*4 + w.
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const bool isB = gi >= 2;
      const bool trn = isB ? BT : AT;
      const int g = gi & 1;
      const int t0 = isB ? it.n0 : it.m0, lim = isB ? p.N : p.M, ld = isB ? p.ldb : p.lda;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int piece = i * NW + wave;
        if (!trn) {
          const int row = piece * 8 + (lane >> 3);                 // granule row 0..127
          const int c = (lane & 7) ^ swz(row);
          const int tr = (row & 63) + 128 * (row >> 6) + 64 * g;   // tile row / column
          const int gm = min(tr, lim - 1 - t0);                    // relative to the tile origin
          it.off[gi][i] = (uint32_t)(((size_t)gm * ld + c * 8) * 2);
        } else {
          const int off = piece * 1024 + lane * 16;
          const int krow = off / ROWB, pc = (off % ROWB) >> 4;
          const int c = pc ^ trswz16(krow);
          const int gc = c * 8;                                    // granule column 0..127
          const int tcol = (gc & 63) + 128 * (gc >> 6) + 64 * g;
          const int col = min(tcol, lim - 8 - t0);                 // relative to the tile origin
          it.off[gi][i] = (uint32_t)(((size_t)krow * ld + col) * 2);
        }
      }
    }
  };
  // one piece of granule gi from k-tile source `src` (an operand base of an item, already advanced to the k-tile)
  auto dma_from = [&](const char* src, const Item& it, int gi, int i, int buf) {
    if (COGV_EXP & 1) return;
    const bool isB = gi >= 2;
    const uint32_t l = smem_u32 + (uint32_t)((isB ? BREG : 0) + buf * BUF + (gi & 1) * GRAN + (i * NW + wave) * 1024);
    dma16(src, it.off[gi][i], l);
  };
  auto dma = [&](const Item& it, int gi, int i, int kt, int buf) {
    const GemmArgs& p = ga.g[it.pi];
    const bool isB = gi >= 2;
    const bool trn = isB ? BT : AT;
    const int ld = isB ? p.ldb : p.lda;
    const size_t kstride = trn ? (size_t)KT * ld * 2 : (size_t)KT * 2;
    dma_from((isB ? it.baseB : it.baseA) + (size_t)kt * kstride, it, gi, i, buf);
  };
  // two whole k-tiles, in the order the loop certifies them: [A01 B01](0) [B23 A23](0) [A01 B01](1) [B23 A23](1)
  auto prologue = [&](const Item& it) {
    const int t1 = min(1, it.nk - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_A01, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_B01, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_B23, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_A23, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_A01, i, t1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_B01, i, t1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_B23, i, t1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_A23, i, t1, 1);
  };

  // per-lane fragment addresses inside a granule (k-step 0): v_mfma_f32_16x16x32, lane l supplies row (l & 15)
  // of a 16-row block and the 8 contraction slots of k-block (l >> 4)
  uint32_t adA[2][4], adB[2][4];       // [k-step][block] (a natural region's k-step flips address bit 6: not an add)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wr * 64 + 16 * i, rb = wc * 64 + 16 * i;
    adA[0][i] = AT ? tr_addr16<ROWB>(smem, ra, lane)
                   : (uint32_t)(uintptr_t)smem + (uint32_t)((ra + l15) * 128 + ((kb ^ swz(ra + l15)) << 4));
    adB[0][i] = BT ? tr_addr16<ROWB>(smem + BREG, rb, lane)
                   : (uint32_t)(uintptr_t)(smem + BREG) + (uint32_t)((rb + l15) * 128 + ((kb ^ swz(rb + l15)) << 4));
    adA[1][i] = adA[0][i] ^ 64u; adB[1][i] = adB[0][i] ^ 64u;
  }

  struct Frag { u32x4 n[2][4]; TrRaw t[2][4]; };     // [k-step][16-row block]; one of the two forms is live
  // one fragment (k-step ks, block blk) of granule gi in buffer buf
  auto read1 = [&](Frag& f, auto gic, auto bufc, auto ksc, auto blkc) {
    if (COGV_EXP & 2) return;
    constexpr int gi = decltype(gic)::value, buf = decltype(bufc)::value, ks = decltype(ksc)::value, blk = decltype(blkc)::value;
    constexpr bool isB = gi >= 2;
    constexpr int off = buf * BUF + (gi & 1) * GRAN;
    if (isB ? BT : AT) tr_issue16_o<off + ks * 32 * ROWB, ROWB>(isB ? adB[0][blk] : adA[0][blk], f.t[ks][blk]);
    else nat_issue_o<off>(isB ? adB[ks][blk] : adA[ks][blk], f.n[ks][blk]);
  };
  auto land = [&](Frag& f, bool trn) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (trn) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.t[ks][b].lo), "+v"(f.t[ks][b].hi) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.n[ks][b]) : : "memory");
      }
  };

  // Work items are taken from the XCD's queue TWO ahead (round 4): while item i is computed the index of item i + 1 is
  // already known, so that its first two k-tiles can ride in the DMA slots of item i's last two k-tiles (cross-item
  // prefetch, below); the queue position asked for in item i's pre-step is item i + 2.
  __shared__ int s_next[2];
  const int xq = blockIdx.x & 7;
  if (threadIdx.x == 0) {
    s_next[0] = xq + 8 * atomicAdd(ga.sched + xq, 1);
    s_next[1] = xq + 8 * atomicAdd(ga.sched + xq, 1);
  }
  __syncthreads();
  Item cur;
  bool certified = false;
  int item = __builtin_amdgcn_readfirstlane(s_next[0]);  // wave-uniform by construction: keeps the DMA bases in SGPRs
  int nxt = __builtin_amdgcn_readfirstlane(s_next[1]);
  if (item < nitems) { setup(item, cur); prologue(cur); }
#if defined(COGV_W4_TS)
  uint32_t ts_acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  const uint64_t ts_begin = __builtin_amdgcn_s_memrealtime(), ts_clk0 = __builtin_readcyclecounter();
  uint64_t ts_last = ts_begin;
#endif
#pragma unroll 1
  while (item < nitems) {
    const GemmArgs& p = ga.g[cur.pi];
    const int nk = cur.nk;
    f32x4 acc[2][8][4];                // [column half][16-row block 4a+i][16-column block j]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Frag fA, fI, fB0, fB1;
    // pre-step: the first A01 / B01 fragments (the only exposed LDS latency of the item)
    // `certified`: this item's 32 prologue DMAs were waited for (vmcnt(0)) in front of the previous epilogue's first C store
    // (pp64_epilogue, land_dma_first) -- behind it a vmcnt wait would also wait for stores, the counter retires in order.
    // The first three half-steps then run without vmcnt waits and the stores drain behind their MFMAs.
    // MEASURED (profiles/r03_gemm_peel_ab_v2.log, three alternating runs): +1.2 to +2.1 % on the epilogues without a bias
    // (plain dgrad, dGeLU + column sums), 0 +- 0.5 % on the bias epilogues -- there the compiler's own wait for the bias
    // load (vmcnt(0) in front of the first store) already does the same thing; -1.4 .. +1 % at K = 1024.
    // (Also measured in round 3, removed: issuing the work-queue atomic by hand and reading it after the k-loop -- no
    //  change, profiles/r03_gemm_async_grab_peel_ab_v1.log.)
    if (!certified) wait_vmcnt<24>();
    __builtin_amdgcn_s_barrier();
    W4_TS(0);
    int grabbed = 0;                                       // the item after the next: asked for now, used after the k-loop
    if (threadIdx.x == 0) grabbed = atomicAdd(ga.sched + xq, 1);
    // Cross-item prefetch: the DMA slots of this item's last two k-tiles -- which would re-fetch its own last k-tile --
    // carry the first two k-tiles of the NEXT item instead, and the next item starts with both resident: no set-up
    // arithmetic, no 32-DMA prologue and no exposed first fetch between the items (2.1-2.7 us of a 60-85 us item with the
    // matrix pipe idle, profiles/r04_gemm_item_phase_probe_v1.log), and 2 of 40 k-tiles less L2 traffic at K = 2560.
    // Taken when the launch allows it (ga.xp_ok) and both tiles are interior: every interior tile has the SAME per-lane
    // offsets against its origin, so only two scalar base pointers change -- selected without a branch inside the k-loop.
    Item nx = cur;                                         // scalar fields only (off unused)
    bool xp = false;
    if (ga.xp_ok && nxt < nitems) {
      uint32_t tm, tn;
      w4_tile_fast((uint32_t)nxt, (uint32_t)p.tiles_m, (uint32_t)p.tiles_n, (uint32_t)ga.group_m, ga.xp_magic_ig, ga.xp_magic_gfull,
                   ga.xp_magic_gtail, tm, tn);
      nx.pi = 0; nx.ksplit = 0; nx.kt0 = 0; nx.nk = nk;
      nx.m0 = (int)tm * TBM; nx.n0 = (int)tn * TBN;
      set_bases(p, nx);
      xp = cur.m0 + TBM <= p.M && cur.n0 + TBN <= p.N && nx.m0 + TBM <= p.M && nx.n0 + TBN <= p.N;
    }
    const char* const xA = xp ? nx.baseA : cur.baseA;      // source of the k-tiles past this item's end
    const char* const xB = xp ? nx.baseB : cur.baseB;
    const size_t ksA = AT ? (size_t)KT * p.lda * 2 : (size_t)KT * 2, ksB = BT ? (size_t)KT * p.ldb * 2 : (size_t)KT * 2;
    static_for<8>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      read1(fA, IC<G_A01>{}, IC<0>{}, IC<(r >> 2)>{}, IC<(r & 3)>{});
      read1(fB0, IC<G_B01>{}, IC<0>{}, IC<(r >> 2)>{}, IC<(r & 3)>{});
    });
    land(fA, AT); land(fB0, BT);
    W4_TS(1);

    // one quarter-step: acc[bh][4 ah + i][j] += A(fa) x B(fb) while granule gin of buffer bin is read into fin and
    // granule gd of k-tile ktd is DMA'd into buffer bd
    auto quarter = [&](auto ahc, auto bhc, Frag& fa, Frag& fb, Frag& fin, auto ginc, auto binc, int gd, const char* gsrc, int bd) {
      constexpr int ah = decltype(ahc)::value, bh = decltype(bhc)::value, gin = decltype(ginc)::value;
      static_for<8>([&](auto rc) {
          constexpr int r = decltype(rc)::value, ks = r >> 2, i = r & 3;
          // the 8 incoming fragments go out in front of the first four MFMA groups (two each: >= 256 MFMA cycles
          // before land()), the 4 DMA pieces in front of the last four
          if constexpr (r < 4) {
            read1(fin, ginc, binc, IC<((2 * r) >> 2)>{}, IC<((2 * r) & 3)>{});
            read1(fin, ginc, binc, IC<((2 * r + 1) >> 2)>{}, IC<((2 * r + 1) & 3)>{});
          } else {
            dma_from(gsrc, cur, gd, r - 4, bd);
          }
          typename HT<T>::v8 va;
          if (AT) va = tr_pack<T>(fa.t[ks][i]); else __builtin_memcpy(&va, &fa.n[ks][i], 16);
          if (COGV_EXP & 4) asm volatile("" :: "v"(va));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            typename HT<T>::v8 vb;
            if (BT) vb = tr_pack<T>(fb.t[ks][j]); else __builtin_memcpy(&vb, &fb.n[ks][j], 16);
            if (COGV_EXP & 4) { asm volatile("" :: "v"(vb)); continue; }
            // operands swapped: D[n][m] -> lane (m = l & 15) holds columns n = 4 (l >> 4) .. +3 of its row
            mfma16_inplace<T>(acc[bh][4 * ah + i][j], vb, va);
          }
          __builtin_amdgcn_sched_barrier(0);
      });
      land(fin, gin >= 2 ? BT : AT);
    };
    // one k-tile; fb01 holds B01(kt), fbx is free and ends up holding B01(kt + 1)
    auto tile = [&](int kt, auto bufc, Frag& fb01, Frag& fbx, auto w0c, auto w1c) {
      constexpr int buf = decltype(bufc)::value;
      constexpr bool W0 = decltype(w0c)::value != 0, W1 = decltype(w1c)::value != 0;   // vmcnt waits of the two half-steps
      // k-tile kt + 2 of this item; past its end: k-tile kt + 2 - nk of the next item (cross-item prefetch) or, without it,
      // this item's last k-tile once more.  Scalar selects, no branch.
      const bool past = kt + 2 >= nk;
      const int ktd = past ? (xp ? kt + 2 - nk : nk - 1) : kt + 2;
      const char* const gA = (past ? xA : cur.baseA) + (size_t)ktd * ksA;
      const char* const gB = (past ? xB : cur.baseB) + (size_t)ktd * ksB;
      if (W0 && !(COGV_EXP & 8192)) wait_vmcnt<16>();
      __builtin_amdgcn_sched_barrier(0);
      if (!(COGV_EXP & 4096)) __builtin_amdgcn_s_barrier();      // probes: results are garbage without them
      __builtin_amdgcn_sched_barrier(0);
      quarter(IC<0>{}, IC<0>{}, fA, fb01, fbx, IC<G_B23>{}, IC<buf>{}, G_A01, gA, buf);
      quarter(IC<0>{}, IC<1>{}, fA, fbx, fI, IC<G_A23>{}, IC<buf>{}, G_B01, gB, buf);
      if (W1 && !(COGV_EXP & 8192)) wait_vmcnt<16>();
      __builtin_amdgcn_sched_barrier(0);
      if (!(COGV_EXP & 4096)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      quarter(IC<1>{}, IC<1>{}, fI, fbx, fA, IC<G_A01>{}, IC<(buf ^ 1)>{}, G_B23, gB, buf);
      quarter(IC<1>{}, IC<0>{}, fI, fb01, fbx, IC<G_B01>{}, IC<(buf ^ 1)>{}, G_A23, gA, buf);
    };
    int kt = 0;
    // half-steps 0..2 read the prologue's granules (all certified above); half-step 3 reads what half-step 0 issued:
    // its vmcnt(16) is the first wait that also covers the previous item's C stores, >= 2 us of MFMA work after the last one
    if (certified && nk >= 2) {
      tile(0, IC<0>{}, fB0, fB1, IC<0>{}, IC<0>{});
      tile(1, IC<1>{}, fB1, fB0, IC<0>{}, IC<1>{});
      kt = 2;
    }
    for (; kt + 1 < nk; kt += 2) {                         // two k-tiles per trip: the B register sets swap roles
      tile(kt, IC<0>{}, fB0, fB1, IC<1>{}, IC<1>{});
      tile(kt + 1, IC<1>{}, fB1, fB0, IC<1>{}, IC<1>{});
    }
    if (nk & 1) tile(nk - 1, IC<0>{}, fB0, fB1, IC<1>{}, IC<1>{});
    __builtin_amdgcn_sched_barrier(0);
    W4_TS(2);
    wait_vmcnt<0>();                                        // the tail prefetches: the next item's first two k-tiles (or redundant)

    // ---- without the cross-item prefetch the next item's prologue goes out BEFORE this item's epilogue (the barrier also
    //      retires every wave's last reads)
    if (threadIdx.x == 0) s_next[0] = xq + 8 * (int)grabbed;
    __syncthreads();
    const int after = __builtin_amdgcn_readfirstlane(s_next[0]);
    const int next = nxt;
    W4_TS(3);
    const Item done = cur;
    bool land_first = false;
    if (xp) {              // both k-tiles resident (vmcnt(0) above, barrier): only the scalar fields change
      cur.m0 = nx.m0; cur.n0 = nx.n0; cur.baseA = nx.baseA; cur.baseB = nx.baseB;
      certified = true;
    } else {
      if (next < nitems) { setup(next, cur); prologue(cur); }
      certified = next < nitems;
      land_first = certified;
    }
    W4_TS(4);

    uint32_t amax_pk = 0u;
    char* strip = smem + 2 * BREG + wave * 4096;
    constexpr int F_FWD_DROP = COGV_EPI_BIAS | COGV_EPI_DROPOUT | COGV_EPI_ABSMAX, F_FWD_GELU = COGV_EPI_BIAS | COGV_EPI_GELU;
    const int mb = done.m0 + wr * 128, nb = done.n0 + wc * 128, csr = (done.m0 >> 7) + wr;
    // transposition through the wave's LDS strip.  (Round 3 also measured a register-exchange form -- v_permlane16_swap,
    // no LDS: equal on plain / bias epilogues, +1.5 % GeLU, -2.3 % column sums, profiles/r03_gemm_swap_epilogue_ab.log; removed.)
#define W4_EPI(F_)                                                                                   \
  do {                                                                                               \
    w4_epilogue<T, F_>(p, acc[0], strip, mb, nb, done.ksplit, lane, amax_pk, csr, land_first);       \
    W4_TS(5);                                                                                        \
    w4_epilogue<T, F_>(p, acc[1], strip, mb, nb + 64, done.ksplit, lane, amax_pk, csr, false);       \
    W4_TS(6);                                                                                        \
  } while (0)
    if (p.splitk > 1) W4_EPI(-2);
    else if (p.out_f32) W4_EPI(-1);
    else if (p.flags == 0) W4_EPI(0);
    else if (p.flags == COGV_EPI_BIAS) W4_EPI(COGV_EPI_BIAS);
    else if (p.flags == COGV_EPI_ACCUM) W4_EPI(COGV_EPI_ACCUM);
    else if (p.flags == F_FWD_DROP) W4_EPI(F_FWD_DROP);
    else if (p.flags == F_FWD_GELU) W4_EPI(F_FWD_GELU);
    else if (p.flags == (COGV_EPI_DGELU | COGV_EPI_COLSUM)) W4_EPI(COGV_EPI_DGELU | COGV_EPI_COLSUM);
    else if (p.flags == COGV_EPI_DGELU) W4_EPI(COGV_EPI_DGELU);
    else if (p.flags == (F_FWD_GELU | COGV_EPI_GELU_DAUX)) W4_EPI(F_FWD_GELU | COGV_EPI_GELU_DAUX);
    else if (p.flags == (COGV_EPI_MULAUX | COGV_EPI_COLSUM)) W4_EPI(COGV_EPI_MULAUX | COGV_EPI_COLSUM);
    else W4_EPI(-1);
#undef W4_EPI
    if ((p.flags & COGV_EPI_ABSMAX) && p.splitk <= 1) {
      uint32_t wv = max(amax_pk & 0xffffu, amax_pk >> 16);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wv = max(wv, (uint32_t)__shfl_xor((int)wv, o, 64));
      if (lane == 0) atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
    item = next;
    nxt = after;
#if defined(COGV_W4_TS)
    ts_acc[7] += 1u;
#endif
  }
#if defined(COGV_W4_TS)
  if (lane == 0) {
    uint32_t* o = reinterpret_cast<uint32_t*>(ga.g[0].C) + ((size_t)blockIdx.x * NW + wave) * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = ts_acc[k];
    o[8] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - ts_begin);
    o[9] = (uint32_t)(__builtin_readcyclecounter() - ts_clk0);
  }
#endif
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ga.sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t) ga.sched[t] = 0;
      __threadfence();
    }
  }
}

// =====================================================================================================
// Skinny-M kernel (M <= 8: incremental decoding, one row per beam): C[M,N] = epilogue(A[M,K] B[N,K]^T) is a matrix-VECTOR
// product per row -- every weight byte is used M times, so the kernel is a pure HBM stream of B (the 4B model reads its
// 7.9 GB of weights once per generated token).  A workgroup owns 8 output columns (8 rows of B); its 4 waves take the
// 512-element chunks of K round robin (16 bytes per lane per row: 8 independent 1-KiB loads per chunk in flight), each
// reduces its partial dot products with DPP, the four partials meet in LDS and lane 0 of wave 0 runs the shared fused
// epilogue on its 8 consecutive columns.  The MFMA tile kernels spend the same traffic on 128 rows of which one is real.
template <typename T>
__global__ __launch_bounds__(256) void gemv_kernel(const GemmArgs p) {
  __shared__ float part[4][GEMV_MAX_M][8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * 8;
  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B) + (size_t)n0 * p.ldb;
  float acc[GEMV_MAX_M][8];
#pragma unroll
  for (int m = 0; m < GEMV_MAX_M; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
  const int nchunk = p.K >> 9;
  for (int c = wave; c < nchunk; c += 4) {
    const int k = (c << 9) + lane * 8;
    u32x4 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(B + (size_t)j * p.ldb + k);
#pragma unroll
    for (int m = 0; m < GEMV_MAX_M; ++m) {
      if (m < p.M) {
        float x[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(A + (size_t)m * p.lda + k), x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float wf[8];
          unpack8<T>(w[j], wf);
          float t = acc[m][j];
#pragma unroll
          for (int e = 0; e < 8; ++e) t = fmaf(x[e], wf[e], t);
          acc[m][j] = t;
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < GEMV_MAX_M; ++m)
    if (m < p.M) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = wave_sum_uniform(acc[m][j]);
        if (lane == 0) part[wave][m][j] = t;
      }
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t amax_pk = 0u;
    for (int m = 0; m < p.M; ++m) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[0][m][j] + part[1][m][j] + part[2][m][j] + part[3][m][j];
      amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n0, v));
    }
    if (p.flags & COGV_EPI_ABSMAX) {
      const uint32_t wv = max(amax_pk & 0xffffu, amax_pk >> 16);
      atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
  }
}

// Decode step, attention-output projection: the skinny-M kernel above with the COMBINE of the decode attention's key
// splits as its prologue (round 3: one launch per layer less in a captured decode step).  attn_decode_kernel leaves per
// (row, head, split) a partial (max m, sum l, 64 unnormalised outputs o) -- 66 floats; the attention output element the GEMV
// needs, att[row][head * 64 + d] = sum_s 2^(m_s - M) o_s[d] / sum_s 2^(m_s - M) l_s, is a few hundred bytes of L2-resident
// partials per lane, so every workgroup recombines the 8-element slices it multiplies instead of a separate combine launch
// writing att and this one reading it (attn_decode_combine_kernel; same arithmetic, splits in order, att rounded to the
// storage type before the product as the two-launch form stores it).  The first weight rows are requested before the
// prologue, so the HBM latency overlaps it.
template <typename T>
__global__ __launch_bounds__(256) void gemv_attn_kernel(const GemmArgs p, const float* __restrict__ part_ws, int H, int nsplit) {
  __shared__ float part[4][GEMV_MAX_M][8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * 8;
  const T* B = reinterpret_cast<const T*>(p.B) + (size_t)n0 * p.ldb;
  float acc[GEMV_MAX_M][8];
#pragma unroll
  for (int m = 0; m < GEMV_MAX_M; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
  const int nchunk = p.K >> 9;
  for (int c = wave; c < nchunk; c += 4) {
    const int k = (c << 9) + lane * 8;
    u32x4 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(B + (size_t)j * p.ldb + k);
    const int head = k >> 6, dd = k & 63;
#pragma unroll
    for (int m = 0; m < GEMV_MAX_M; ++m) {
      if (m < p.M) {
        const float* base = part_ws + ((size_t)m * H + head) * nsplit * 66;
        float mx = -INFINITY;
        for (int sp = 0; sp < nsplit; ++sp) mx = fmaxf(mx, base[sp * 66]);
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float tl[32];                                         // l_s 2^(m_s - M) per split (nsplit <= 32), zero beyond
#pragma unroll
        for (int i = 0; i < 32; ++i) tl[i] = 0.f;
#pragma unroll
        for (int sp = 0; sp < 32; ++sp) {
          if (sp < nsplit) {
            const float mi = base[sp * 66];
            const float wgt = (mi == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mi - mx);
            tl[sp] = base[sp * 66 + 1] * wgt;
            const float* po = base + sp * 66 + 2 + dd;        // 8-byte aligned (66 floats per partial)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(po[e], wgt, o[e]);
          }
        }
        // the sum in the association of attn_decode_combine_kernel's xor-butterfly (lanes >= nsplit hold zeros there too),
        // so that both forms of the step produce the same bits
#pragma unroll
        for (int w2 = 16; w2 > 0; w2 >>= 1)
#pragma unroll
          for (int i = 0; i < w2; ++i) tl[i] += tl[i + w2];
        const float L = tl[0];
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = o[e] / L;
        const u32x4 xr = pack8<T>(x);                         // the attention output in its storage type
        unpack8<T>(xr, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float wf[8];
          unpack8<T>(w[j], wf);
          float t = acc[m][j];
#pragma unroll
          for (int e = 0; e < 8; ++e) t = fmaf(x[e], wf[e], t);
          acc[m][j] = t;
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < GEMV_MAX_M; ++m)
    if (m < p.M) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = wave_sum_uniform(acc[m][j]);
        if (lane == 0) part[wave][m][j] = t;
      }
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t amax_pk = 0u;
    for (int m = 0; m < p.M; ++m) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[0][m][j] + part[1][m][j] + part[2][m][j] + part[3][m][j];
      amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n0, v));
    }
    if (p.flags & COGV_EPI_ABSMAX) {
      const uint32_t wv = max(amax_pk & 0xffffu, amax_pk >> 16);
      atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Matrix-vector kernel with the layer's LayerNorms as its PROLOGUE (decode steps, M <= 8 rows, K = hidden size <= 4096).
// A decode step of one token spends more time in its four per-layer Sandwich-LN launches (7 us each: launch latency plus a
// dependent chain on one row) than the LayerNorms cost in bytes, so the chain
//     z --[post-LN gamma_p, beta_p, Sandwich scale |z|max]--> + residual --> t --[pre-LN gamma, beta, Sandwich scale |t|max]--> x_in
//     y = epilogue(x_in . W^T + b)
// (mpu/sparse_transformer.py:314-342: t = x + LN3(attn) feeding LN2, or t = y + LN4(mlp) feeding the next layer's LN1 /
// the final LayerNorm) runs inside EVERY workgroup of the GEMV that consumes x_in: the vectors are M x K 16-bit values, a
// few KB, and recomputing them 320-1280 times is cheaper than one more launch.  Workgroup 0 also stores t (the
// residual stream) once.  Rounding points are those of ln_fwd_kernel (LayerNorm output rounded to the storage type
// before the residual add, t rounded, x_in rounded).  |t|max is taken over all M rows, as x.abs().max() does.
// SF: the residual stream is fp32 -- `res`, `t_out` and (without a post-LN) `z` are fp32 rows, t is formed and
// normalised without an intermediate rounding (the decode counterpart of ln_fwd_kernel's STREAM modes).
template <typename T, int MT, bool SF>   // MT: compile-time bound of the row count (1, 2, 4, 8): registers follow the real batch
__global__ __launch_bounds__(256) void gemv_ln_kernel(const GemvLnArgs q) {
  typedef Row8<T, SF> SR;                // a stream row slice
  extern __shared__ __attribute__((aligned(16))) char xs_raw[];           // x_in [M][K] as T
  __shared__ float part[4][MT][8];
  __shared__ float red[8];
  __shared__ uint32_t redm[16];
  __shared__ float s_amax;
  const GemmArgs& p = q.g;
  T* xs = reinterpret_cast<T*>(xs_raw);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = p.K, nvec = K >> 3;                // K % 512 == 0: every thread owns whole 8-element vectors v = tid, tid + 256
  const float inv_k = 1.0f / (float)K;
  const bool has_post = q.gamma_p != nullptr;
  const int v0 = threadIdx.x, v1 = threadIdx.x + 256;
  const bool ok1 = v1 < nvec;                      // v0 < nvec always (K >= 2048 is not required: guard below)
  const bool ok0 = v0 < nvec;
  // ---- everything that does not depend on the prologue is requested first: this wave's first weight chunk (the HBM
  //      stream: its latency now overlaps the LayerNorm arithmetic), the input rows, the residual and the four affine vectors
  const int n0 = blockIdx.x * 8;
  const T* B = reinterpret_cast<const T*>(p.B) + (size_t)n0 * p.ldb;
  const int nchunk = K >> 9;
  u32x4 w[8];
  {
    const int k = (min(wave, nchunk - 1) << 9) + lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(B + (size_t)j * p.ldb + k);
  }
  const T* Z = reinterpret_cast<const T*>(q.z);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 zr[MT][2], gpr[2], bpr[2], gnr[2], bnr[2];
  typename SR::raw rr[MT][2];            // the stream rows: the residual (post-LN form) or z itself (plain-input form)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int v = u ? v1 : v0; const bool ok = u ? ok1 : ok0;
    gnr[u] = ok ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(q.gamma) + v * 8) : zero4;
    bnr[u] = ok ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(q.beta) + v * 8) : zero4;
    gpr[u] = (ok && has_post) ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(q.gamma_p) + v * 8) : zero4;
    bpr[u] = (ok && has_post) ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(q.beta_p) + v * 8) : zero4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      zr[m][u] = (ok && m < p.M && (has_post || !SF)) ? *reinterpret_cast<const u32x4*>(Z + (size_t)m * K + v * 8) : zero4;
      rr[m][u] = (ok && m < p.M && (has_post || SF)) ? SR::ld(has_post ? q.res : q.z, (size_t)m * K + v * 8) : SR::zero();
    }
  }
  float zamax = q.z_absmax ? *q.z_absmax : 0.f;
  // sums over the workgroup of up to 2 * MT values at once (one LDS round for all rows)
  auto block_sums = [&](float (&a)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = wave_sum_uniform(a[m]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m) part[wave][m][0] = a[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = (part[0][m][0] + part[1][m][0]) + (part[2][m][0] + part[3][m][0]);
  };
  float tv[MT][2][8];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (SF && !has_post) SR::to_f(rr[m][u], tv[m][u]);         // the plain input IS the fp32 stream
      else unpack8<T>(zr[m][u], tv[m][u]);
    }
  if (has_post) {                 // t = residual + LN_post(z), rounded where ln_fwd_kernel rounds
    if (!q.z_absmax) {            // max |z| over all rows taken here (see gemv2_ln_kernel)
      uint32_t zpk = 0u;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) zpk = absmax_pk8(zpk, zr[m][u]);
      zamax = absmax_pk_block<T>(zpk, redm);
    }
    const float c = zamax * 0.125f;
    const float eps_p = q.eps * c * c;
    float s[MT], qq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      s[m] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[m] += tv[m][u][i];            // vectors past K are zero
    }
    block_sums(s);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k;
      qq[m] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (u ? ok1 : ok0)
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = tv[m][u][i] - mean; qq[m] += d * d; }
    }
    block_sums(qq);
    float gp[2][8], bp[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) { unpack8<T>(gpr[u], gp[u]); unpack8<T>(bpr[u], bp[u]); }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k, rstd = 1.0f / sqrtf(qq[m] * inv_k + eps_p);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float r[8], o[8];
        SR::to_f(rr[m][u], r);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (tv[m][u][i] - mean) * rstd * gp[u][i] + bp[u][i];
        if (!SF) { u32x4 lo = pack8<T>(o); unpack8<T>(lo, o); }   // all-T form: LayerNorm output rounded before the residual add
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += r[i];
        if (!SF) { const u32x4 ov = pack8<T>(o); unpack8<T>(ov, o); }   // t rounded to its storage type
#pragma unroll
        for (int i = 0; i < 8; ++i) tv[m][u][i] = o[i];
        if (!(u ? ok1 : ok0)) {
#pragma unroll
          for (int i = 0; i < 8; ++i) tv[m][u][i] = 0.f;
        } else if (blockIdx.x == 0 && q.t_out && m < p.M)
          (void)SR::st(q.t_out, (size_t)m * K + (u ? v1 : v0) * 8, o, 0u);
      }
    }
  }
  // pre-LN: Sandwich scale = max |t| over all rows (x.abs().max(), mpu/sparse_transformer.py:40-44) -- the published
  // abs-max when t is the plain input, else taken here -- then mean / variance per row
  float amax;
  if (!has_post && q.z_absmax) {
    amax = zamax;
  } else {
    uint32_t amax_pk = 0u;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int u = 0; u < 2; ++u) {                                 // rows >= M and vectors past K are zero
        if (SF) {
#pragma unroll
          for (int i = 0; i < 8; ++i) amax_pk = max(amax_pk, __float_as_uint(tv[m][u][i]) & 0x7fffffffu);
        } else amax_pk = absmax_pk8(amax_pk, pack8<T>(tv[m][u]));
      }
    __syncthreads();
    const float a = SF ? absmax_f32_block(amax_pk, redm) : absmax_pk_block<T>(amax_pk, redm);
    if (threadIdx.x == 0) s_amax = a;
    __syncthreads();
    amax = s_amax;
  }
  {
    const float c = amax * 0.125f;
    const float eps_n = q.eps * c * c;
    float s[MT], qq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      s[m] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[m] += tv[m][u][i];
    }
    block_sums(s);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k;
      qq[m] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (u ? ok1 : ok0)
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = tv[m][u][i] - mean; qq[m] += d * d; }
    }
    block_sums(qq);
    float gn[2][8], bn[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) { unpack8<T>(gnr[u], gn[u]); unpack8<T>(bnr[u], bn[u]); }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k, rstd = 1.0f / sqrtf(qq[m] * inv_k + eps_n);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u ? ok1 : ok0) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (tv[m][u][i] - mean) * rstd * gn[u][i] + bn[u][i];
          *reinterpret_cast<u32x4*>(xs + (size_t)m * K + (u ? v1 : v0) * 8) = pack8<T>(o);
        }
      }
    }
  }
  __syncthreads();
  // ---- the matrix-vector product proper (as gemv_kernel, x_in read from LDS; the first chunk is already in registers)
  float acc[MT][8];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
  for (int c = wave; c < nchunk; c += 4) {
    const int k = (c << 9) + lane * 8;
    if (c != wave) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(B + (size_t)j * p.ldb + k);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float x[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(xs + (size_t)m * K + k), x);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float wf[8];
        unpack8<T>(w[j], wf);
        float t = acc[m][j];
#pragma unroll
        for (int e = 0; e < 8; ++e) t = fmaf(x[e], wf[e], t);
        acc[m][j] = t;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = wave_sum_uniform(acc[m][j]);
      if (lane == 0) part[wave][m][j] = t;
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t am = 0u;
    for (int m = 0; m < p.M; ++m) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[0][m][j] + part[1][m][j] + part[2][m][j] + part[3][m][j];
      am = absmax_pk(am, epilogue8<T>(p, m, n0, v));
    }
    if (p.flags & COGV_EPI_ABSMAX) {
      const uint32_t wv = max(am & 0xffffu, am >> 16);
      atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
  __shared__ float red[16];
  const size_t nvec = (size_t)p.M * (p.N / 8);
  uint32_t amax_pk = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / (p.N / 8));
    const int n = (int)(i % (p.N / 8)) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < p.splitk; ++s) {
      const float* w = p.ws + ((size_t)s * p.M + m) * p.N + n;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(w);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(w + 4);
      v[0] += x0[0]; v[1] += x0[1]; v[2] += x0[2]; v[3] += x0[3];
      v[4] += x1[0]; v[5] += x1[1]; v[6] += x1[2]; v[7] += x1[3];
    }
    amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n, v));
  }
  if (p.flags & COGV_EPI_ABSMAX) {
    const float bm = absmax_pk_block<T>(amax_pk, reinterpret_cast<uint32_t*>(red));
    if (threadIdx.x == 0) atomic_max_nonneg(p.absmax, bm);
  }
}

template <typename T, bool AT, bool BT, int WM, int WN, int MI, int NJ, int BKT>
void launch_glds(GemmArgs& a, hipStream_t st) {
  constexpr int TBM = WM * 32 * MI, TBN = WN * 32 * NJ;
  constexpr int shmem = 3 * (TBM + TBN) * 2 * BKT;
  a.tiles_m = (a.M + TBM - 1) / TBM; a.tiles_n = (a.N + TBN - 1) / TBN;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<T, AT, BT, WM, WN, MI, NJ, BKT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_glds_kernel<T, AT, BT, WM, WN, MI, NJ, BKT>), dim3(a.tiles_m * a.tiles_n, a.splitk),
                     dim3(WM * WN * 64), shmem, st, a);
}
template <typename T, int WM, int WN, int MI, int NJ, int BKT>
void launch_glds_layout(const cogv_gemm_desc* d, GemmArgs& a, hipStream_t st) {
  if (!d->trans_a && !d->trans_b) launch_glds<T, false, false, WM, WN, MI, NJ, BKT>(a, st);
  else if (!d->trans_a && d->trans_b) launch_glds<T, false, true, WM, WN, MI, NJ, BKT>(a, st);
  else if (d->trans_a && d->trans_b) launch_glds<T, true, true, WM, WN, MI, NJ, BKT>(a, st);
  else launch_glds<T, true, false, WM, WN, MI, NJ, BKT>(a, st);
}

int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}

// Work-queue counters of the persistent kernel: 64 zero-initialised slots of 16 ints per device (8 per-XCD item
// counters + the finished-workgroup count), used round robin (a launch re-arms its slot when it finishes; launches
// on one stream are ordered anyway).  The only device memory this library allocates itself: 4 KiB per GPU, on first use.
int* sched_slot() {
  constexpr int MAX_DEV = 16, SLOTS = 64;
  static int* pool[MAX_DEV] = {};
  static unsigned turn[MAX_DEV] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAX_DEV) return nullptr;
  if (!pool[dev]) {
    if (hipMalloc(reinterpret_cast<void**>(&pool[dev]), SLOTS * 16 * sizeof(int)) != hipSuccess) return nullptr;
    (void)hipMemset(pool[dev], 0, SLOTS * 16 * sizeof(int));
    (void)hipDeviceSynchronize();
  }
  return pool[dev] + 16 * (turn[dev]++ % SLOTS);
}

#ifndef COGV_W4_TU
// The generation-4 kernel's eight instantiations (2 dtypes x 4 layouts) are compiled from this same file in eight
// separate translation units (-DCOGV_W4_TU=k, see build.py: they build in parallel); unit k exports cogv_w4_launch_k.
#define W4_DECL(k) extern "C" __attribute__((visibility("hidden"))) int cogv_w4_launch_##k(const void* ga, int grid, void* stream);
W4_DECL(0) W4_DECL(1) W4_DECL(2) W4_DECL(3) W4_DECL(4) W4_DECL(5) W4_DECL(6) W4_DECL(7)
#undef W4_DECL
typedef int (*w4_launch_fn)(const void*, int, void*);
const w4_launch_fn W4_LAUNCH[8] = {cogv_w4_launch_0, cogv_w4_launch_1, cogv_w4_launch_2, cogv_w4_launch_3,
                                   cogv_w4_launch_4, cogv_w4_launch_5, cogv_w4_launch_6, cogv_w4_launch_7};
// unit index: bit 2 = fp16 (else bf16), bits 1..0 = layout: 0 NT (forward), 1 NN (dgrad: B stored [K, N]),
// 2 TN (wgrad: both stored contraction-major), 3 A stored [K, M] only
template <typename T, bool AT, bool BT> constexpr int w4_unit() {
  return (std::is_same<T, f16_t>::value ? 4 : 0) + (AT ? (BT ? 2 : 3) : (BT ? 1 : 0));
}

template <typename T, bool AT, bool BT, int GEN = 3>
int launch_pp64(GroupArgs& ga, hipStream_t st) {
  constexpr int shmem = 2 * 65536;
  ga.sched = sched_slot();
  if (!ga.sched) return COGV_ERR_LAUNCH;
  {   // raster group height (experiments: COGV_GEMM_GROUP_M); 4 rows x 8 columns of tiles per XCD by default
    static const int gm = [] { const char* e = getenv("COGV_GEMM_GROUP_M"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 32 ? v : 4; }();
    ga.group_m = gm;
  }
  ga.item_start[0] = 0;
  for (int i = 0; i < ga.count; ++i) {
    GemmArgs& a = ga.g[i];
    a.tiles_m = (a.M + 255) / 256; a.tiles_n = (a.N + 255) / 256;
    ga.item_start[i + 1] = ga.item_start[i] + a.tiles_m * a.tiles_n * a.splitk;
  }
  for (int i = ga.count; i < MAX_GROUP; ++i) ga.item_start[i + 1] = ga.item_start[ga.count];
  const int num_cu = num_cus();
  const int items = ga.item_start[ga.count];
  if constexpr (GEN == 4) {
    ga.xp_ok = 0; ga.xp_magic_ig = ga.xp_magic_gfull = ga.xp_magic_gtail = 0u;
    const char* xp_env = getenv("COGV_GEMM_XP");          // read per launch: tests and A/B runs switch it inside one process
    const bool xp_enabled = !xp_env || atoi(xp_env) != 0;
    const GemmArgs& a0 = ga.g[0];
    const int nkt = a0.K / 64;
    if (xp_enabled && ga.count == 1 && a0.splitk == 1 && nkt >= 4 && (nkt & 1) == 0 && items > num_cu) {
      // multiplicative forms of the three divisions of the tile order, verified against the dividing form for EVERY item of
      // this geometry (cached: a training step launches a handful of distinct geometries)
      struct Geo { int tm, tn, gm, ok; uint32_t mig, mgf, mgt; };
      static Geo cache[32];
      static int ncache = 0;
      const Geo* hit = nullptr;
      for (int i = 0; i < ncache; ++i)
        if (cache[i].tm == a0.tiles_m && cache[i].tn == a0.tiles_n && cache[i].gm == ga.group_m) { hit = &cache[i]; break; }
      Geo g;
      if (!hit) {
        g.tm = a0.tiles_m; g.tn = a0.tiles_n; g.gm = ga.group_m;
        const uint32_t tail = (uint32_t)(a0.tiles_m % ga.group_m);
        g.mig = w4_magic((uint32_t)(ga.group_m * a0.tiles_n)); g.mgf = w4_magic((uint32_t)ga.group_m); g.mgt = w4_magic(tail);
        g.ok = 1;
        for (uint32_t b = 0; b < (uint32_t)items && g.ok; ++b) {
          uint32_t m1, n1, m2, n2;
          w4_tile_slow(b, (uint32_t)g.tm, (uint32_t)g.tn, (uint32_t)g.gm, m1, n1);
          w4_tile_fast(b, (uint32_t)g.tm, (uint32_t)g.tn, (uint32_t)g.gm, g.mig, g.mgf, g.mgt, m2, n2);
          if (m1 != m2 || n1 != n2) g.ok = 0;
        }
        if (ncache < 32) cache[ncache++] = g;
        hit = &g;
      }
      ga.xp_ok = hit->ok; ga.xp_magic_ig = hit->mig; ga.xp_magic_gfull = hit->mgf; ga.xp_magic_gtail = hit->mgt;
    }
    return W4_LAUNCH[w4_unit<T, AT, BT>()](&ga, items < num_cu ? items : num_cu, st);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp64_kernel<T, AT, BT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_pp64_kernel<T, AT, BT>), dim3(items < num_cu ? items : num_cu), dim3(512), shmem, st, ga);
    return COGV_OK;
  }
}
template <typename T, int GEN = 3>
int launch_pp64_layout(int trans_a, int trans_b, GroupArgs& ga, hipStream_t st) {
  if (!trans_a && !trans_b) return launch_pp64<T, false, false, GEN>(ga, st);
  if (!trans_a && trans_b) return launch_pp64<T, false, true, GEN>(ga, st);
  if (trans_a && trans_b) return launch_pp64<T, true, true, GEN>(ga, st);
  return launch_pp64<T, true, false, GEN>(ga, st);
}
template <typename T>
void launch_splitk_reduce(GemmArgs& a, hipStream_t st) {
  const size_t nvec = (size_t)a.M * (a.N / 8);
  int blocks = (int)((nvec + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, a);
}

// Second-generation skinny-M kernels (gemv.hip, one translation unit per dtype: 0 bf16, 1 fp16).  COGV_GEMV2=0 keeps the
// first-generation kernels below (A/B runs); a launcher returns COGV_ERR_UNSUPPORTED for a shape it does not take.
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_launch_0(const void* args, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_launch_1(const void* args, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_attn_launch_0(const void* args, const float* partials, int heads, int nsplit, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_attn_launch_1(const void* args, const float* partials, int heads, int nsplit, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_ln_launch_0(const void* args, int stream_f32, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_ln_launch_1(const void* args, int stream_f32, void* stream);
inline bool gemv2_enabled() {
  static const bool on = [] { const char* e = getenv("COGV_GEMV2"); return !e || atoi(e) != 0; }();
  return on;
}

template <typename T>
int launch_gemm(const cogv_gemm_desc* d, GemmArgs& a, hipStream_t st) {
  // skinny M (decode steps): the HBM-streaming matrix-vector kernel; takes every fused epilogue except the column sums
  if (a.M <= GEMV_MAX_M && !d->trans_a && !d->trans_b && (a.K & 511) == 0 && (a.N & 7) == 0 && !(d->flags & COGV_EPI_COLSUM) &&
      (a.lda & 7) == 0 && (a.ldb & 7) == 0 && d->kernel_variant == 0) {
    a.splitk = 1;
    if (gemv2_enabled()) {
      const int rc2 = std::is_same<T, f16_t>::value ? cogv_gemv2_launch_1(&a, st) : cogv_gemv2_launch_0(&a, st);
      if (rc2 != COGV_ERR_UNSUPPORTED) return rc2 != COGV_OK ? rc2 : cogv_check_launch();
    }
    hipLaunchKernelGGL((gemv_kernel<T>), dim3(a.N / 8), dim3(256), 0, st, a);
    return cogv_check_launch();
  }
  const bool glds_ok = (a.K % BK) == 0 && a.M >= 64 && a.N >= 64 && (!d->trans_a || (a.M & 7) == 0) &&
                       (!d->trans_b || (a.N & 7) == 0) && d->kernel_variant != 1;
  if ((d->flags & COGV_EPI_COLSUM) && !glds_ok) return COGV_ERR_UNSUPPORTED;
  if (glds_ok) {
    // variant 3: generation 2 -- 256x128x32, 4 waves (128x64 each), two workgroups per CU (M or N < 256, huge operands)
    // variant 9: generation 3 -- 256x256x64 ping-pong (8 waves), persistent, 16x16x32 MFMAs
    // variant 10: generation 4 -- the same tile and ring with 4 waves of 128x128 (a third fewer LDS fragment bytes per
    //            flop: +5-7 % forward / dgrad, +15 % wgrad over generation 3): default whenever the 256 tile slots per
    //            round are filled about as well as variant 3's 512
    // (the intermediate designs -- 256x128x64 / 128x128x32 / 256x256x32 rings, ping-pong on 32-deep tiles -- measured
    //  within +-5 % of variant 3 and were removed; DESIGN.md section 4 keeps the numbers)
    int variant = d->kernel_variant;
    const size_t a_span = (size_t)(d->trans_a ? a.K : a.M) * a.lda * 2, b_span = (size_t)(d->trans_b ? a.K : a.N) * a.ldb * 2;
    const bool v9_ok = a.M >= 256 && a.N >= 256 && a_span < (1ull << 32) && b_span < (1ull << 32);   // 32-bit DMA offsets
    if (variant != 0 && variant != 3 && variant != 9 && variant != 10) variant = 0;
    if ((d->flags & COGV_EPI_COLSUM) && (!v9_ok || (variant != 0 && variant != 9 && variant != 10))) return COGV_ERR_UNSUPPORTED;
    if ((d->flags & COGV_EPI_COLSUM) && variant != 9) variant = 10;
    if ((variant == 9 || variant == 10) && !v9_ok) variant = 0;
    if (variant == 0) {
      variant = 3;
      if (v9_ok) {
        const int cu = num_cus();
        const int i3 = ((a.M + 255) / 256) * ((a.N + 127) / 128) * a.splitk, i9 = ((a.M + 255) / 256) * ((a.N + 255) / 256) * a.splitk;
        const float e3 = (float)i3 / (float)(((i3 + 2 * cu - 1) / (2 * cu)) * 2 * cu);
        const float e9 = (float)i9 / (float)(((i9 + cu - 1) / cu) * cu);
        if (e9 * 1.1f >= e3) variant = 10;     // measured 1.1-1.4x at equal fill (16x16x32 MFMAs: less power per flop)
      }
    }
    if (variant == 3) launch_glds_layout<T, 2, 2, 4, 2, 32>(d, a, st);
    else if (variant == 9 || variant == 10) {                                           // generations 3 / 4 (operands < 4 GiB)
      GroupArgs ga; ga.count = 1; ga.g[0] = a;
      const int rc = variant == 10 ? launch_pp64_layout<T, 4>(d->trans_a, d->trans_b, ga, st)
                                   : launch_pp64_layout<T, 3>(d->trans_a, d->trans_b, ga, st);
      if (rc != COGV_OK) return rc;
      a.tiles_m = ga.g[0].tiles_m; a.tiles_n = ga.g[0].tiles_n;
    }
    else launch_glds_layout<T, 4, 2, 2, 2, 64>(d, a, st);
    if (a.splitk > 1) {
      const size_t nvec = (size_t)a.M * (a.N / 8);
      int blocks = (int)((nvec + 255) / 256); if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, a);
    }
    return cogv_check_launch();
  }
  dim3 grid(a.tiles_m * a.tiles_n, a.splitk), block(NTHREADS);
  const size_t shmem = 65536;
#define LAUNCH(AT_, BT_)                                                                                 \
  do {                                                                                                   \
    static bool attr_set = false;                                                                        \
    if (!attr_set) {                                                                                     \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, AT_, BT_>),                      \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                       \
      attr_set = true;                                                                                   \
    }                                                                                                    \
    hipLaunchKernelGGL((gemm_kernel<T, AT_, BT_>), grid, block, shmem, st, a);                           \
  } while (0)
  if (!d->trans_a && !d->trans_b) LAUNCH(false, false);
  else if (!d->trans_a && d->trans_b) LAUNCH(false, true);
  else if (d->trans_a && d->trans_b) LAUNCH(true, true);
  else LAUNCH(true, false);
#undef LAUNCH
  if (a.splitk > 1) {
    const size_t nvec = (size_t)a.M * (a.N / 8);
    int blocks = (int)((nvec + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, a);
  }
  return cogv_check_launch();
}

}  // namespace

extern "C" size_t cogv_gemm_workspace_bytes(const cogv_gemm_desc* d) {
  if (!d || d->splitk <= 1) return 0;
  return (size_t)d->splitk * (size_t)d->M * (size_t)d->N * sizeof(float);
}

// Heuristic used by the host side: split the contraction when the output has too few tiles to fill 256 CUs
// (weight-gradient GEMMs of the 336M config: 64..256 tiles, contraction = b*1088 tokens).
extern "C" int cogv_gemm_colsum_rows(int M) { return 2 * ((M + 255) / 256); }

extern "C" int cogv_gemm_pick_splitk(int M, int N, int K) {
  // 256x256 tiles on one persistent workgroup per CU.  Pick the split that fills whole rounds best; every split
  // costs an fp32 slab write + read (8 M N bytes at ~4 TB/s) against 2 M N K flops at ~1.1 PFLOP/s: 1100 / K each.
  return cogv_gemm_pick_splitk_tiles(((M + 255) / 256) * ((N + 255) / 256), K);
}

// the same for `tiles` 256x256 output tiles in total (a grouped launch): one split count for all problems
extern "C" int cogv_gemm_pick_splitk_tiles(int tiles, int K) {
  const int nk = (K + BK - 1) / BK;
  const int slots = 256;
  if (tiles >= 4 * slots || nk < 16) return 1;
  const float cost = 1100.f / (float)K;
  int best = 1; float best_score = -1.f;
  for (int s = 1; s <= 16 && nk / s >= 8; ++s) {
    const int items = tiles * s;
    const int rounds = (items + slots - 1) / slots;
    const float eff = (float)items / (float)(rounds * slots);
    const float score = eff / (1.f + (s > 1 ? cost * s : 0.f));
    if (score > best_score) { best_score = score; best = s; }
  }
  return best;
}

static int build_gemm_args(const cogv_gemm_desc* d, GemmArgs& a) {
  if (!d) return COGV_ERR_ARG;
  if (d->dtype != COGV_F16 && d->dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return COGV_ERR_ARG;
  if ((d->N & 7) || (d->ldc & 7)) return COGV_ERR_ARG;
  if ((d->lda & 7) || (d->ldb & 7)) return COGV_ERR_ARG;
  if (!d->trans_a && (d->K & 7)) return COGV_ERR_ARG;   // K-contiguous operands are read in 16-byte chunks
  if (!d->trans_b && (d->K & 7)) return COGV_ERR_ARG;
  if (d->trans_a && (d->M & 7)) return COGV_ERR_ARG;
  if (((uintptr_t)d->A | (uintptr_t)d->B | (uintptr_t)d->C) & 15) return COGV_ERR_ARG;
  if ((d->flags & COGV_EPI_BIAS) && (!d->bias || ((uintptr_t)d->bias & 15))) return COGV_ERR_ARG;
  if ((d->flags & (COGV_EPI_DGELU | COGV_EPI_MULAUX)) && !d->aux) return COGV_ERR_ARG;
  if ((d->flags & COGV_EPI_DGELU) && (d->flags & COGV_EPI_MULAUX)) return COGV_ERR_ARG;      // one aux operand
  if ((d->flags & COGV_EPI_GELU_DAUX) && !(d->flags & COGV_EPI_GELU)) return COGV_ERR_ARG;
  if ((d->flags & (COGV_EPI_DGELU | COGV_EPI_GELU | COGV_EPI_MULAUX)) && d->aux && ((d->ldaux & 7) || ((uintptr_t)d->aux & 15)))
    return COGV_ERR_ARG;
  if ((d->flags & COGV_EPI_ABSMAX) && !d->absmax) return COGV_ERR_ARG;
  if ((d->flags & COGV_EPI_DROPOUT) && !(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return COGV_ERR_ARG;

  a.A = d->A; a.B = d->B; a.C = d->C;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.bias = d->bias; a.aux = d->aux; a.ldaux = d->ldaux; a.absmax = d->absmax;
  a.flags = d->flags; a.out_f32 = d->out_f32;
  a.seed = d->seed; a.stream_id = d->stream_id;
  a.thr16 = (d->flags & COGV_EPI_DROPOUT) ? (uint32_t)(d->dropout_p * 65536.0f + 0.5f) : 0u;
  a.keep_scale = 65536.0f / (65536.0f - (float)a.thr16);
  a.tiles_m = (d->M + BM - 1) / BM; a.tiles_n = (d->N + BN - 1) / BN;
  const int nk = (d->K + BK - 1) / BK;
  a.splitk = d->splitk > 1 ? d->splitk : 1;
  if (a.splitk > nk) a.splitk = nk;
  a.ktiles_per_split = (nk + a.splitk - 1) / a.splitk;
  a.splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;   // no empty splits
  a.colsum_ws = d->colsum_partial;
  if ((d->flags & COGV_EPI_COLSUM) && (!d->colsum_partial || ((uintptr_t)d->colsum_partial & 15) || d->splitk > 1 || d->out_f32)) return COGV_ERR_ARG;
  a.ws = reinterpret_cast<float*>(d->workspace);
  if (a.splitk > 1) {
    if (!a.ws || d->workspace_bytes < (size_t)a.splitk * a.M * a.N * sizeof(float)) return COGV_ERR_ARG;
    if ((uintptr_t)a.ws & 15) return COGV_ERR_ARG;
  }
  return COGV_OK;
}

extern "C" int cogv_gemm(const cogv_gemm_desc* d, void* stream) {
  GemmArgs a;
  const int rc = build_gemm_args(d, a);
  if (rc != COGV_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == COGV_F16) return launch_gemm<f16_t>(d, a, st);
  return launch_gemm<bf16_t>(d, a, st);
}

// y = epilogue(LN_pre([res + LN_post(z)]) . B^T): the decode step's GEMV with its LayerNorms as prologue (gemv_ln_kernel).
extern "C" int cogv_gemv_ln(const cogv_gemm_desc* d, const cogv_ln_prologue* ln, void* stream) {
  if (!d || !ln) return COGV_ERR_ARG;
  GemvLnArgs a;
  const int rc = build_gemm_args(d, a.g);
  if (rc != COGV_OK) return rc;
  if (d->trans_a || d->trans_b || a.g.M > GEMV_MAX_M || (a.g.K & 511) || a.g.K > 4096 || (a.g.N & 7) || (a.g.ldb & 7)) return COGV_ERR_UNSUPPORTED;
  if (d->flags & (COGV_EPI_COLSUM | COGV_EPI_ACCUM | COGV_EPI_DGELU | COGV_EPI_MULAUX | COGV_EPI_DROPOUT) || d->out_f32 || d->splitk > 1) return COGV_ERR_UNSUPPORTED;
  if (!ln->z || !ln->gamma || !ln->beta) return COGV_ERR_ARG;
  if (ln->gamma_post && (!ln->beta_post || !ln->residual)) return COGV_ERR_ARG;
  if (((uintptr_t)ln->z | (uintptr_t)ln->gamma | (uintptr_t)ln->beta | (uintptr_t)ln->gamma_post | (uintptr_t)ln->beta_post |
       (uintptr_t)ln->residual | (uintptr_t)ln->t_out) & 15) return COGV_ERR_ARG;
  a.z = ln->z; a.z_absmax = ln->z_absmax; a.gamma_p = ln->gamma_post; a.beta_p = ln->beta_post; a.res = ln->residual;
  a.t_out = ln->t_out; a.gamma = ln->gamma; a.beta = ln->beta; a.eps = ln->eps;
  const bool sf = ln->stream_f32 != 0;
  a.g.splitk = 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (gemv2_enabled()) {
    const int rc2 = d->dtype == COGV_F16 ? cogv_gemv2_ln_launch_1(&a, sf ? 1 : 0, st) : cogv_gemv2_ln_launch_0(&a, sf ? 1 : 0, st);
    if (rc2 != COGV_ERR_UNSUPPORTED) return rc2 != COGV_OK ? rc2 : cogv_check_launch();
  }
  const int mt = a.g.M <= 1 ? 1 : a.g.M <= 2 ? 2 : a.g.M <= 4 ? 4 : 8;
  const int shmem = mt * a.g.K * 2;                 // rows [M, mt) are written as zeros-normalised junk nobody reads
  const dim3 grid(a.g.N / 8), block(256);
#define GEMV_LN_LAUNCH(T_, MT_)                                                                                          \
  do {                                                                                                                   \
    static bool attr = false;                                                                                            \
    if (!attr) {                                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_ln_kernel<T_, MT_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_ln_kernel<T_, MT_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);  \
      attr = true;                                                                                                       \
    }                                                                                                                    \
    if (sf) hipLaunchKernelGGL((gemv_ln_kernel<T_, MT_, true>), grid, block, shmem, st, a);                               \
    else hipLaunchKernelGGL((gemv_ln_kernel<T_, MT_, false>), grid, block, shmem, st, a);                                 \
  } while (0)
  if (d->dtype == COGV_F16) { if (mt == 1) GEMV_LN_LAUNCH(f16_t, 1); else if (mt == 2) GEMV_LN_LAUNCH(f16_t, 2); else if (mt == 4) GEMV_LN_LAUNCH(f16_t, 4); else GEMV_LN_LAUNCH(f16_t, 8); }
  else { if (mt == 1) GEMV_LN_LAUNCH(bf16_t, 1); else if (mt == 2) GEMV_LN_LAUNCH(bf16_t, 2); else if (mt == 4) GEMV_LN_LAUNCH(bf16_t, 4); else GEMV_LN_LAUNCH(bf16_t, 8); }
#undef GEMV_LN_LAUNCH
  return cogv_check_launch();
}

// y = epilogue(att . B^T), att = the combination of cogv_attention_decode's split partials (skip_combine form): gemv_attn_kernel.
extern "C" int cogv_gemv_attn(const cogv_gemm_desc* d, const void* partials, int heads, int capacity, void* stream) {
  if (!d || !partials || heads <= 0 || capacity <= 0 || capacity > 4096 || ((uintptr_t)partials & 15)) return COGV_ERR_ARG;
  cogv_gemm_desc dd = *d;
  dd.A = dd.B;                    // the A operand does not exist: keep build_gemm_args' pointer checks happy
  dd.lda = dd.K;
  GemmArgs a;
  const int rc = build_gemm_args(&dd, a);
  if (rc != COGV_OK) return rc;
  if (d->trans_a || d->trans_b || a.M > GEMV_MAX_M || a.K != heads * 64 || (a.K & 511) || (a.N & 7) || (a.ldb & 7)) return COGV_ERR_UNSUPPORTED;
  if (d->flags & (COGV_EPI_COLSUM | COGV_EPI_ACCUM | COGV_EPI_DGELU | COGV_EPI_MULAUX | COGV_EPI_DROPOUT | COGV_EPI_GELU) || d->out_f32 || d->splitk > 1) return COGV_ERR_UNSUPPORTED;
  a.splitk = 1;
  const int nsplit = (capacity + 127) / 128;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (gemv2_enabled()) {
    const float* pw = reinterpret_cast<const float*>(partials);
    const int rc2 = d->dtype == COGV_F16 ? cogv_gemv2_attn_launch_1(&a, pw, heads, nsplit, st) : cogv_gemv2_attn_launch_0(&a, pw, heads, nsplit, st);
    if (rc2 != COGV_ERR_UNSUPPORTED) return rc2 != COGV_OK ? rc2 : cogv_check_launch();
  }
  if (d->dtype == COGV_F16) hipLaunchKernelGGL((gemv_attn_kernel<f16_t>), dim3(a.N / 8), dim3(256), 0, st, a, reinterpret_cast<const float*>(partials), heads, nsplit);
  else hipLaunchKernelGGL((gemv_attn_kernel<bf16_t>), dim3(a.N / 8), dim3(256), 0, st, a, reinterpret_cast<const float*>(partials), heads, nsplit);
  return cogv_check_launch();
}

// Several GEMMs of the same dtype and layout in one persistent launch of the generation-3 kernel (see GroupArgs).
// Every problem must satisfy that kernel's requirements (M, N >= 256, K % 64 == 0, operands < 4 GiB); otherwise
// COGV_ERR_UNSUPPORTED and the caller issues them one by one.
extern "C" int cogv_gemm_grouped(const cogv_gemm_desc* descs, int count, void* stream) {
  if (!descs || count < 1 || count > MAX_GROUP) return COGV_ERR_ARG;
  GroupArgs ga; ga.count = count;
  for (int i = 0; i < count; ++i) {
    const cogv_gemm_desc* d = descs + i;
    const int rc = build_gemm_args(d, ga.g[i]);
    if (rc != COGV_OK) return rc;
    if (d->dtype != descs[0].dtype || d->trans_a != descs[0].trans_a || d->trans_b != descs[0].trans_b) return COGV_ERR_ARG;
    const GemmArgs& a = ga.g[i];
    const size_t a_span = (size_t)(d->trans_a ? a.K : a.M) * a.lda * 2, b_span = (size_t)(d->trans_b ? a.K : a.N) * a.ldb * 2;
    if ((a.K % BK) || a.M < 256 || a.N < 256 || a_span >= (1ull << 32) || b_span >= (1ull << 32)) return COGV_ERR_UNSUPPORTED;
    if (d->trans_a && (a.M & 7)) return COGV_ERR_UNSUPPORTED;
    if (d->trans_b && (a.N & 7)) return COGV_ERR_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int lrc = descs[0].dtype == COGV_F16 ? launch_pp64_layout<f16_t, 4>(descs[0].trans_a, descs[0].trans_b, ga, st)
                                             : launch_pp64_layout<bf16_t, 4>(descs[0].trans_a, descs[0].trans_b, ga, st);
  if (lrc != COGV_OK) return lrc;
  for (int i = 0; i < count; ++i)
    if (ga.g[i].splitk > 1) {
      if (descs[0].dtype == COGV_F16) launch_splitk_reduce<f16_t>(ga.g[i], st);
      else launch_splitk_reduce<bf16_t>(ga.g[i], st);
    }
  return cogv_check_launch();
}
#else   // COGV_W4_TU: one instantiation of the generation-4 kernel and its launcher
}  // namespace

#define W4_CAT2(a, b) a##b
#define W4_CAT(a, b) W4_CAT2(a, b)
extern "C" __attribute__((visibility("hidden"))) int W4_CAT(cogv_w4_launch_, COGV_W4_TU)(const void* ga, int grid, void* stream) {
  using T = std::conditional<(COGV_W4_TU & 4) != 0, f16_t, bf16_t>::type;
  constexpr int L = COGV_W4_TU & 3;
  constexpr bool AT = L >= 2, BT = L == 1 || L == 2;
  constexpr int shmem = 2 * 65536 + 4 * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<T, AT, BT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_w4_kernel<T, AT, BT>), dim3(grid), dim3(256), shmem, reinterpret_cast<hipStream_t>(stream),
                     *reinterpret_cast<const GroupArgs*>(ga));
  return COGV_OK;
}
#endif
