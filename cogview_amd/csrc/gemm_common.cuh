// Shared by every GEMM generation: tile constants, the grouped-launch argument block, the XCD-aware tile order.
// Part of the GEMM family of csrc/gemm.hip (included there, in this order: common, gen1, lds, gen2, gen3, gen4, gemv_gen1);
// not a stand-alone header.
#pragma once

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

}  // namespace
