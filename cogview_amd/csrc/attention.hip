// Fused attention for CogView's standard_attention (reference mpu/sparse_transformer.py:652-673), head dim 64.
//
//   S = (Q / sqrt(d)) K^T ;  S = S*M - 10000*(1-M) ;  P = softmax(S) ;  P = dropout(P) ;  O = P V
//
// with M the left-to-right mask, optionally with a fully visible prefix ("sep" form built at
// mpu/sparse_transformer.py:477-489):  M[i][j] = (j <= i + (s_k - s_q)) || (j < sep_k).
// The s x s score / probability tensors of the reference (4 materialised copies, 94.7 MB*b per layer at 4B)
// never exist here: flash-style streaming softmax forward, recompute-based backward.
//
// MFMA formulation (v_mfma_f32_32x32x16, wave64).  All kernels keep ONE attention row/column per lane:
//   forward / dQ :  S^T[key][query] = K . Q^T      -> lane = query, the 16 accumulator registers = keys
//   dK/dV        :  S  [query][key] = Q . K^T      -> lane = key,   the 16 accumulator registers = queries
// so row statistics (max, sum, LSE, D) are lane-local scalars, and the probabilities sitting in the
// accumulator registers are *already* a valid B-operand fragment for the next MFMA (O^T = V^T P^T etc.),
// because the hardware's k-slot <-> (lane-group g, element e) map may be relabelled freely as long as the
// A operand uses the same relabelling:   slot(g, e) of k-step t  <->  index 16t + 8(e>>2) + 4g + (e&3).
//
// Data movement: every K / V / Q / dO tile (64 rows x 64 halves) is DMA'd ONCE, in its natural layout, straight
// into a 3-stage LDS ring (global_load_lds_dwordx4, prefetch distance two blocks, counted vmcnt, raw s_barrier).
// Operands whose contraction index runs along the tile ROWS (V^T, K^T, Q^T, dO^T) are read with the LDS
// transposing read ds_read_b64_tr_b16 (semantics verified by tools/probes/tr_read_probe.hip); operands whose
// contraction index is d are read with ds_read_b128 from the SAME tile.  One XOR swizzle of the 16-B chunk
// index, aswz(row) = (b1<<2)|(b3<<1)|b2, keeps both read patterns bank-conflict free; it is applied to the
// per-lane DMA source address (the DMA writes LDS linearly).  No staging registers, no register transposes.
//
// q/k/v are addressed as base + b*batch_stride + row*row_stride + head*64, i.e. straight out of the
// [b, s, 3*h/p] QKV GEMM output -- the reference's _transpose_for_scores permute copies
// (mpu/sparse_transformer.py:112-120,159) are folded into the addressing.
#include "common.cuh"
#include "cogview_hip.h"

#include <type_traits>

namespace {

constexpr int HD = 64;          // head dim
constexpr int NT = 256;         // threads per block (4 waves)
constexpr float MASKED = -10000.0f;
constexpr int TILE = 8192;      // one 64 x 64 16-bit tile
// Stages of the K|V (forward, dQ) / Q|dO (dK.dV) LDS ring = prefetch distance + 1.  Three stages keep two blocks in flight per
// workgroup (48-53 KiB), two stages one (32-36 KiB: it still has a whole block's compute, ~2 us, to land).  Round 5 built the
// two-stage form expecting more workgroups per CU -- the kernels are parked 29-35 % of their wave cycles
// (profiles/r05_attention_pmc.txt) -- but the occupancy of these kernels is set by REGISTERS, not LDS: 143 / 168 / 226 per lane
// (forward / dQ / dK.dV with stored keep bits) = 3 / 3 / 2 waves per SIMD with either ring depth (tools/probes/attn_occupancy.hip,
// profiles/r05_attention_occupancy_probe.log; the counters read 2.4 / 2.3 / 1.7 resident waves per SIMD for both).  Asking the
// compiler for one more wave (__launch_bounds__(256, 4 / 4 / 3)) spills 52 / 120 / 224 bytes per lane inside the block loop,
// between the asm-issued LDS reads and their waits (refused by build.py's hazard scan).  Two stages still measured 1 % faster per
// kernel in the 4B step and 2 % on the 336M shape (profiles/r05_attention_stages_ab.log), so they ship.
#ifndef COGV_ATTN_STAGES
#define COGV_ATTN_STAGES 2
#endif
constexpr int NSTG = COGV_ATTN_STAGES;
// dK.dV register diet (round 5 experiment, profiles/r05_attention_dkdv_diet_ab.log): the training instantiation needs 226
// registers = two waves per SIMD.  COGV_DKDV_AHEAD=0 drops the ahead-of-time request of the transposed dO / Q fragments (-32),
// COGV_DKDV_VLDS=1 keeps the V fragments of the wave's own keys in LDS behind the ring instead of in registers (-16; +16 KiB
// per workgroup: three still fit a CU with the two-stage ring), COGV_DKDV_WAVES=3 asks the allocator for three waves per SIMD
// (168 registers, 24 B of spill outside the asm-read windows).  The runtime then grants three workgroups per CU -- and the
// kernel is NOT faster: backward 1018-1031 us against 1000-1013 with two waves (same call), the 4B step unchanged.  A third
// wave per SIMD does not cover the parked cycles; the defaults (two waves, fragments requested ahead) stay.
#ifndef COGV_DKDV_KW_PITCH
#define COGV_DKDV_KW_PITCH 272
#endif
#ifndef COGV_DKDV_AHEAD
#define COGV_DKDV_AHEAD 1
#endif
#ifndef COGV_DKDV_VLDS
#define COGV_DKDV_VLDS 0
#endif
#ifndef COGV_DKDV_WAVES
#define COGV_DKDV_WAVES 2
#endif
constexpr int DKDV_OWN_V = COGV_DKDV_VLDS ? 4 * 4 * 64 * 16 : 0;      // bytes of the waves' own V fragments behind the ring
static_assert(NSTG == 2 || NSTG == 3, "ring of two or three stages");
constexpr int COLSUM_SMEM = (128 * 68 + 256) * 4;      // tile_colsum's scratch (the ring is free by then)
constexpr int ring_bytes(int stage, bool colsum) { return (colsum && NSTG * stage < COLSUM_SMEM) ? COLSUM_SMEM : NSTG * stage; }

struct AttnArgs {
  const void* q; const void* k; const void* v; void* o;        // forward
  const void* dout; void* dq; void* dk; void* dv;              // backward
  float* lse; float* dvec;                                     // [b][H][s_q]; dvec: [2][b][H][s_q] (-D; dropout row keys, or -LSE log2 e with stored keep bits)
  float* colsum_ws;                                            // optional [B * nblk][3 * H * 64]: sums of dq | dk | dv
  // optional (dense kernels, DROP == 2): the dropout keep bits of the forward pass, one 32-bit word per (query, 64-key block,
  // key half fg): word [(b * H + head)][kb][fg][q], bit 31 - n <-> key 64 kb + 32 (n >> 4) + 8 ((n & 15) >> 2) + 4 fg + (n & 3).
  // The forward kernel writes them (its lane = query layout holds exactly these 32 keys per lane and block), the backward
  // kernels read them instead of regenerating the draws (4 to 4.5 of ~14.5 VALU slots per score element in each of them).
  uint32_t* keepbits;
  // optional (flexible instantiations, IDX = true): an arbitrary mask M [B or 1][s_q][s_k] in the storage type, applied as the
  // reference does -- scaled score * M - 10000 * (1 - M), mpu/sparse_transformer.py:661-663 -- for ANY real M (binary or not);
  // the left-to-right rule is off then (sep_k = s_k).  mask_bs = 0 broadcasts one mask over the batch.
  const void* mask; long long mask_bs;
  const int* kv_index; long long kv_index_bs;                  // optional: key slot j reads K/V row kv_index[b][j] & 0x7fffffff
  // sparse TRAINING form in slot space (sp_w > 0): one index row per query block g = q / sp_w (kv_index_gs apart);
  // bit 31 of an entry = slot masked (-10000); the first sp_npiv slots are pivots and take sp_bias (added to the
  // scaled score); the left-to-right rule runs on slots, the block's own sp_w queries being the last sp_w slots
  long long kv_index_gs; int sp_w, sp_npiv; float sp_bias;
  long long q_bs, k_bs, v_bs, o_bs, do_bs, dq_bs, dk_bs, dv_bs; // batch strides (elements)
  int q_rs, k_rs, v_rs, o_rs, do_rs, dq_rs, dk_rs, dv_rs;       // row strides (elements)
  int B, H, s_q, s_k, sep_k;   // sep_k: keys [0, sep_k) visible to every query
  float scale;
  uint32_t thr16; float keep_scale; uint32_t rng_key;
};

// Workgroup -> (block along the sequence, head, batch / plane).  The grid is ONE-dimensional and XCD-aware: hardware deals
// consecutive workgroup ids to the 8 XCDs round-robin and each XCD has its own L2, so with a plain (blocks, H, B) grid the
// sequence blocks of one (batch, head) -- which all read the same K / V (forward, dQ) or Q / dO (dK/dV) tiles -- sat on
// 8 different XCDs and every tile was fetched through the fabric up to 8 times.  Here XCD x owns the (batch, head) units
// u = x (mod 8) and walks the sequence blocks of one unit consecutively (heaviest block of the causal triangle first).
struct BlockId { int bx, head, z; bool ok; };
__device__ __forceinline__ BlockId xcd_block_id(int nx, int H, int Z, bool heavy_last) {
  const int L = blockIdx.x, x = L & 7, j = L >> 3;
  const int ul = j / nx, k = j - ul * nx;
  const int u = ul * 8 + x;
  BlockId id;
  id.ok = u < H * Z;
  id.z = u / H; id.head = u - id.z * H;
  id.bx = heavy_last ? nx - 1 - k : k;
  return id;
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// chunk swizzle of a 64-row tile: row bit 1 -> chunk bit 2 (separates the rows r, r+2 of a transpose block),
// row bits 3,2 -> chunk bits 1,0 (with bit 1: a bijection of (row>>1)&7, which makes ds_read_b128 conflict free)
__device__ __forceinline__ int aswz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// DMA of one tile: 8 pieces of 1 KiB (8 rows each); wave w issues pieces w and w+4.  Rows beyond `nrows` are
// clamped to the last valid row (finite data; every use of such a row is masked out by the kernels).
template <typename T>
__device__ __forceinline__ void dma_tile(const T* base, long long rs, int row0, int nrows, char* lds, int wave, int lane,
                                         const int* lds_index = nullptr) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = wave + 4 * i;
    const int row = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ aswz(row);
    int gr = min(row0 + row, nrows - 1);
    if (lds_index) {      // gathered keys: slot -> row through the index table staged in LDS (asm read: no vmcnt drain)
      const uint32_t a = (uint32_t)(uintptr_t)(lds_index + gr);
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(gr) : "v"(a) : "memory");
      gr &= 0x7fffffff;
    }
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + (long long)gr * rs + c * 8), (lds_void_t*)(lds + piece * 1024), 16, 0, 0);
  }
}
// 64 floats (one per row of the current block) -> LDS; every wave issues it (same data) to keep vmcnt uniform
__device__ __forceinline__ void dma_stat(const float* base, int row0, int nrows, char* lds, int lane) {
  const int gr = min(row0 + lane, nrows - 1);
  __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + gr), (lds_void_t*)lds, 4, 0, 0);
}

template <typename T>
__device__ __forceinline__ typename HT<T>::v8 nat_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const typename HT<T>::v8*>(lds + row * 128 + ((chunk ^ aswz(row)) << 4));
}

// Transposed A-operand fragment: output row i = d = 32*dblk + (lane&31); contraction slots (g, e) <-> tile row
// src0 + 8(e>>2) + 4g + (e&3).  Two ds_read_b64_tr_b16 (rows src0+4g.. and src0+8+4g..), issued through inline asm
// (the builtin form makes hipcc drain vmcnt(0) in front of every read while LDS-DMA is in flight).
struct TrRaw { u32x2 lo, hi; };
__device__ __forceinline__ uint32_t tr_lane_off(int dblk, int lane) {
  // byte offset inside a tile for src0 = 0, WITHOUT the row-dependent swizzle of rows >= 4 (added per call)
  const int G = lane >> 4, cb = G & 1, g = G >> 1, r = (lane & 15) >> 2, qq = lane & 3;
  const int c = 4 * dblk + 2 * cb + (qq >> 1);
  return (uint32_t)((4 * g + r) * 128 + ((c ^ ((r >> 1) << 2)) << 4) + (qq & 1) * 8);
}
// rows src0 + 4g + r: bits 3,2 of the row come from (src0 + 4g) -> chunk bits 1,0; src0 is a multiple of 8, so
// bit 2 = g (already lane-dependent): fold g's contribution here, src0's (bit 3 and up) at the call.
__device__ __forceinline__ uint32_t tr_lane_fix(int lane) { return (uint32_t)((((lane >> 5) & 1)) << 4); }   // chunk bit 0 <- row bit 2 = g
template <int SRC0>
__device__ __forceinline__ void tr_issue(uint32_t tile_addr, uint32_t lane_off, TrRaw& o) {
  // row = SRC0 + 4g + r (+8 for the second read).  chunk ^= (b3<<1) where b3 = bit 3 of the row:
  //   first read : b3 = (SRC0 >> 3) & 1 ; second read: b3 = ((SRC0 + 8) >> 3) & 1
  constexpr int X0 = (((SRC0 >> 3) & 1) << 1) << 4, X1 = ((((SRC0 + 8) >> 3) & 1) << 1) << 4;
  const uint32_t a0 = tile_addr + SRC0 * 128 + (lane_off ^ X0);        // lane_off < 1024: the XOR stays inside
  const uint32_t a1 = tile_addr + (SRC0 + 8) * 128 + (lane_off ^ X1);  // the 128-B row of this lane
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(o.lo), "=&v"(o.hi) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ void tr_wait(TrRaw& a, TrRaw& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : : "memory");
}
template <typename T>
__device__ __forceinline__ typename HT<T>::v8 tr_pack(const TrRaw& r) {
  typename HT<T>::v8 out;
  __builtin_memcpy(&out, &r.lo, 8);
  __builtin_memcpy(reinterpret_cast<char*>(&out) + 8, &r.hi, 8);
  return out;
}

template <typename T>
__device__ __forceinline__ typename HT<T>::v8 cvt8(const float* p) {
  u32x4 w = pack8<T>(p);
  typename HT<T>::v8 out; __builtin_memcpy(&out, &w, 16);
  return out;
}
template <typename T>
__device__ __forceinline__ typename HT<T>::v8 load_frag_global(const T* rowptr, bool valid) {
  u32x4 w = {0u, 0u, 0u, 0u};
  if (valid) w = *reinterpret_cast<const u32x4*>(rowptr);
  typename HT<T>::v8 out; __builtin_memcpy(&out, &w, 16);
  return out;
}

__device__ __forceinline__ bool visible(int q, int key, int off, int sep_k) { return key <= q + off || key < sep_k; }

// raw v_exp_f32 (2^x): no denormal-range fix-up sequence (softmax terms that small are zero anyway)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Masking of one 32-element accumulator fragment whose element e has index base + (e&3) + 8*(e>>2) along the
// masked axis.  `lim` = largest visible index relative to `base` (visible iff rel <= lim) and `bound` = number of
// valid indices relative to `base`; both are per-lane scalars, the per-element offsets are compile-time constants.
// Only called for blocks that touch the diagonal / the sequence end (wave-uniform branch at the call site).
__device__ __forceinline__ void mask_frag(f32x16& a, int lim, int lim_sep, int bound, float masked_raw) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int rel = (e & 3) + 8 * (e >> 2);
    float v = a[e];
    v = (rel <= lim || rel < lim_sep) ? v : masked_raw;     // reference: score*M - 10000*(1-M)
    v = (rel < bound) ? v : -INFINITY;                       // padding beyond the sequence: not a key at all
    a[e] = v;
  }
}

// Attention dropout bits.  One 64-bit draw per (attention row arow = (b*H+head)*s_q + q, group g = key >> 2 of 4
// consecutive keys); element i = key & 3 takes bits 16 (i & 1) .. +15 of word i >> 1:
//     rk   = pcg32((lo32(arow) ^ key) + hi32(arow) * 0x85EBCA6B)            per row, once per kernel
//     x    = rk + g * 0x9E3779B9;  x ^= x >> 15;  x *= 0x2C1B3C6D;  x ^= x >> 12          word 0
//     y    = (x ^ 0x68E31DA4) * 0x297A2D39;  y ^= y >> 15                                   word 1
// The row hash carries the quality (PCG); the per-draw part is a Weyl step (an ADD per draw when g advances by a
// constant, as it does along a lane's keys in the forward and dQ kernels) and one multiply-xorshift round per word --
// 16 instruction slots per draw where the former generator (PCG + xorshift32 on a 64-bit counter with carry) took 28,
// in kernels that are bound by VALU issue.  Neighbour correlations of the keep mask measured at the noise level
// (oracle/cogview_oracle.py restates it; tests compare every kernel with the oracle's masks).
struct RowKey { uint32_t rk; };
__device__ __forceinline__ RowKey row_key(uint32_t key, unsigned long long arow) {
  return RowKey{pcg32(((uint32_t)arow ^ key) + (uint32_t)(arow >> 32) * 0x85EBCA6Bu)};
}
// (Round 6 measured the same round on 24-BIT multiplies -- v_mul_u32_u24, on the theory that the two v_mul_lo_u32 are quarter-rate
//  and half of the draw's issue slots: forward 349.6 vs 349.6 and 348.4 vs 346.1 us in alternating processes at the bench shape,
//  profiles/r06_attention_mul24_generator_ab.log.  The multiplies are not what bounds the forward kernel; the 32-bit round, whose
//  draws keep 32 bits of state, stays.)
__device__ __forceinline__ u32x2 attn_bits_w(uint32_t w) {          // w = rk + g * 0x9E3779B9
  u32x2 o;
  uint32_t x = w ^ (w >> 15);
  x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  uint32_t y = (x ^ 0x68E31DA4u) * 0x297A2D39u;
  y ^= y >> 15;
  o[0] = x; o[1] = y;
  return o;
}
__device__ __forceinline__ u32x2 attn_bits_at(const RowKey& r, uint32_t g) { return attn_bits_w(r.rk + g * 0x9E3779B9u); }
__device__ __forceinline__ uint32_t bits_of(const u32x2& r, int i) { return (r[i >> 1] >> (16 * (i & 1))) & 0xffffu; }
// keep test on element i of a draw without extracting the 16-bit field: high halves compare the whole word against
// thr << 16 (the low half only adds less than one unit), low halves compare the low 16 bits
__device__ __forceinline__ bool keep_of(const u32x2& r, int i, uint32_t thr16) {
  return (i & 1) ? (r[i >> 1] >= (thr16 << 16)) : ((r[i >> 1] & 0xffffu) >= thr16);
}

// 4 transposed fragments (2 d-blocks x k-steps t = 0,1 of a 32-row sub-block SB) of one tile
template <typename T, int SB>
__device__ __forceinline__ void tr_frags_issue(uint32_t tile_addr, const uint32_t (&loff)[2], TrRaw (&r)[2][2]) {
#pragma unroll
  for (int d = 0; d < 2; ++d) { tr_issue<SB * 32>(tile_addr, loff[d], r[0][d]); tr_issue<SB * 32 + 16>(tile_addr, loff[d], r[1][d]); }
}

// Synchronous variant (reads + wait in ONE asm statement): the destination registers are valid when the
// statement ends, so it stays correct even if the compiler spills them.  Used by the dK/dV kernel, whose
// register pressure is at the limit; the asynchronous form above must only be used where the compiler does not spill
// around it (cogview_amd/csrc/build.py's scan_asm_hazards fails the build on any scratch access or register read while
// an asm-issued load is still in flight).
template <typename T, int SB>
__device__ __forceinline__ void tr_frags_sync(uint32_t tile_addr, const uint32_t (&loff)[2], TrRaw (&r)[2][2]) {
  constexpr int S0 = SB * 32, S1 = SB * 32 + 16;
  constexpr int XA = (((S0 >> 3) & 1) << 1) << 4, XB = ((((S0 + 8) >> 3) & 1) << 1) << 4;
  constexpr int XC = (((S1 >> 3) & 1) << 1) << 4, XD = ((((S1 + 8) >> 3) & 1) << 1) << 4;
  uint32_t a[2][4];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    a[d][0] = tile_addr + S0 * 128 + (loff[d] ^ XA);
    a[d][1] = tile_addr + (S0 + 8) * 128 + (loff[d] ^ XB);
    a[d][2] = tile_addr + S1 * 128 + (loff[d] ^ XC);
    a[d][3] = tile_addr + (S1 + 8) * 128 + (loff[d] ^ XD);
  }
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
      "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\tds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r[0][0].lo), "=&v"(r[0][0].hi), "=&v"(r[1][0].lo), "=&v"(r[1][0].hi),
        "=&v"(r[0][1].lo), "=&v"(r[0][1].hi), "=&v"(r[1][1].lo), "=&v"(r[1][1].hi)
      : "v"(a[0][0]), "v"(a[0][1]), "v"(a[0][2]), "v"(a[0][3]), "v"(a[1][0]), "v"(a[1][1]), "v"(a[1][2]), "v"(a[1][3])
      : "memory");
}

// Column sums of one workgroup's 128 rows x 64 head columns of a gradient that sits in the accumulators as
// vals[d][e] <-> column 32 d + 8 (e >> 2) + 4 fg + (e & 3) of the lane's row (already rounded to the storage type,
// zero for invalid rows).  Through LDS (the ring is free at the end of the kernels; needs 128 x 68 + 256 floats):
// every lane stores its row (272-byte row pitch: conflict-free b128 stores), then thread t adds 32 rows of column
// t & 63 and the four row slices are combined.  Deterministic.  (A butterfly of 160 dependent cross-lane shuffles
// per wave cost 25 us per workgroup here.)
__device__ __forceinline__ void tile_colsum(const float (&vals)[2][16], float* lds, float* dst, int lane, int wave) {
  constexpr int PITCH = 68;
  const int fr = lane & 31, fg = lane >> 5;
  float* row = lds + (wave * 32 + fr) * PITCH;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      *reinterpret_cast<f32x4*>(row + 32 * d + 8 * gq + 4 * fg) =
          f32x4{vals[d][4 * gq], vals[d][4 * gq + 1], vals[d][4 * gq + 2], vals[d][4 * gq + 3]};
  __syncthreads();
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int r = 0; r < 32; r += 2) {
    s0 += lds[(slice * 32 + r) * PITCH + col];
    s1 += lds[(slice * 32 + r + 1) * PITCH + col];
  }
  float* red = lds + 128 * PITCH;
  red[slice * 64 + col] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < 64) dst[threadIdx.x] = red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
  __syncthreads();
}

// =====================================================================================================
// forward: grid (ceil(s_q/128), H, B); wave w owns queries q0 + 32w .. +31.  Ring stage = K tile | V tile.
// =====================================================================================================
template <typename T, bool IDX, int DROP>       // DROP: 1 / 0 = dropout on / off at compile time, -1 = decided by p.thr16,
                                                //       2 = on, and the keep bits are stored in p.keepbits for the backward pass
__global__ __launch_bounds__(NT, 2) void attn_fwd_kernel(const AttnArgs p) {
  const bool drop = DROP < 0 ? (p.thr16 != 0u) : (DROP != 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 3 stages x 16 KiB
  constexpr int STAGE = 2 * TILE, LPT = 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 31, fg = lane >> 5;
  const int nblk_x = (p.s_q + 127) >> 7;
  const BlockId bid = xcd_block_id(nblk_x, p.H, p.B, true);
  if (!bid.ok) return;
  const int b = bid.z, head = bid.head;
  const int q0 = bid.bx * 128, q0w = q0 + wave * 32;
  // IDX (compile time): gathered keys / the sparse training form; the dense instantiation carries none of that code
  const bool spw = IDX && p.sp_w > 0;               // sparse training form (slot space)
  const int gblk = spw ? q0 / p.sp_w : 0;
  const int off = spw ? (p.s_k - p.sp_w - gblk * p.sp_w) : (p.s_k - p.s_q);
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.q_bs + head * HD;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.k_bs + head * HD;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.v_bs + head * HD;
  const int myq = q0w + fr;
  const bool wave_active = q0w < p.s_q;

  typename HT<T>::v8 qf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qf[t] = load_frag_global<T>(Q + (long long)myq * p.q_rs + 16 * t + 8 * fg, myq < p.s_q);

  const int q_last = min(p.s_q, q0 + 128) - 1;
  const int kend_blk = min(p.s_k, max(q_last + off + 1, p.sep_k));
  const int nkb = (kend_blk + 63) >> 6;
  const int qw_last = min(p.s_q, q0w + 32) - 1;
  const int kend_w = wave_active ? min(p.s_k, max(qw_last + off + 1, p.sep_k)) : 0;

  f32x16 oacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2 = p.scale * 1.4426950408889634f;   // raw score -> log2 domain
  const float masked_raw = MASKED / p.scale;         // raw value whose scaled score is exactly -10000
  const long long arow = ((long long)b * p.H + head) * p.s_q + myq;
  const RowKey cb = row_key(p.rng_key, (unsigned long long)arow);
  // dropout scale 1 / (1 - p) folded into the exponent: the probabilities (and their running sum) carry it, the
  // final normalisation takes it back out -- no multiply per kept element
  const float kofs = drop ? __builtin_amdgcn_logf(p.keep_scale) : 0.f;     // v_log_f32 = log2
  const uint32_t loff[2] = {tr_lane_off(0, lane) ^ tr_lane_fix(lane), tr_lane_off(1, lane) ^ tr_lane_fix(lane)};
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;

  // gathered form (sparse_attention_inference, mpu/sparse_transformer.py:727-750): key slot j is row kv_index[b][j] of
  // K and V; the table (<= 4096 slots) is staged once behind the ring
  int* lidx = nullptr;
  if (IDX && p.kv_index) {
    lidx = reinterpret_cast<int*>(smem + NSTG * STAGE);
    const int* gi = p.kv_index + (long long)b * p.kv_index_bs + (long long)gblk * p.kv_index_gs;
    for (int i = threadIdx.x; i < p.s_k; i += NT) lidx[i] = gi[i];
    __syncthreads();
  }
  const float sp_bias_raw = spw ? p.sp_bias / p.scale : 0.f;      // added to RAW scores of pivot slots
  auto issue = [&](int kb, int st) {
    dma_tile<T>(K, p.k_rs, kb * 64, p.s_k, smem + st * STAGE, wave, lane, lidx);
    dma_tile<T>(V, p.v_rs, kb * 64, p.s_k, smem + st * STAGE + TILE, wave, lane, lidx);
  };
  // DROP == 2: this lane's keep word of key block kb goes to kwp[kb * 2 * s_q]
  uint32_t* const kwp = DROP == 2 ? p.keepbits + ((((long long)b * p.H + head) * ((p.s_k + 63) >> 6)) * 2 + fg) * p.s_q + min(myq, p.s_q - 1) : nullptr;
  if (nkb > 0) { issue(0, 0); if (NSTG > 2) issue(nkb > 1 ? 1 : 0, 1); }
  int st = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    wait_vmcnt<(NSTG - 2) * LPT>();
    __builtin_amdgcn_s_barrier();
    issue(min(kb + NSTG - 1, nkb - 1), st == 0 ? NSTG - 1 : st - 1);
    if (kb * 64 < kend_w) {
      const char* lk = smem + st * STAGE;
      const uint32_t lv = smem_addr + st * STAGE + TILE;
      f32x16 sacc[2];
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[sb][e] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          sacc[sb] = HT<T>::mfma32(nat_frag<T>(lk, sb * 32 + fr, 2 * t + fg), qf[t], sacc[sb]);
      }
      // V^T fragments of the first 32 keys: issued now, consumed after the softmax
      TrRaw vr[2][2];
      tr_frags_issue<T, 0>(lv, loff, vr);
      if (IDX && lidx) {
        // slot attributes of this block, one slot per lane: masked flag (bit 31 of the index entry; honoured in both
        // gathered forms -- a decode step over a fixed-capacity key/value cache flags the slots not written yet) and
        // "is a pivot" (training form only: sp_npiv = 0 otherwise)
        int raw;
        const uint32_t ia = (uint32_t)(uintptr_t)(lidx + min(kb * 64 + lane, p.s_k - 1));
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(raw) : "v"(ia) : "memory");
        const unsigned long long mflag = __ballot(raw < 0), mpiv = __ballot(kb * 64 + lane < p.sp_npiv);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          const uint32_t fm = (uint32_t)(mflag >> (32 * sb)) >> (4 * fg), pm = (uint32_t)(mpiv >> (32 * sb)) >> (4 * fg);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int bit = (e & 3) + 8 * (e >> 2);
            float v = sacc[sb][e];
            v = ((fm >> bit) & 1u) ? masked_raw : v;
            v += ((pm >> bit) & 1u) ? sp_bias_raw : 0.f;
            sacc[sb][e] = v;
          }
        }
      }
      const int kfirst = kb * 64;
      if (IDX && p.mask) {      // arbitrary mask tensor: raw' = raw * M + (-10000 / scale) * (1 - M)
        const T* mrow = reinterpret_cast<const T*>(p.mask) + (long long)b * p.mask_bs + (long long)min(myq, p.s_q - 1) * p.s_k;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = kfirst + sb * 32 + 4 * fg + (e & 3) + 8 * (e >> 2);
            const float m = HT<T>::to_f(mrow[min(key, p.s_k - 1)]);
            sacc[sb][e] = fmaf(sacc[sb][e], m, masked_raw * (1.f - m));
          }
      }
      const bool all_visible = (kfirst + 63 <= q0w + off) || (kfirst + 63 < p.sep_k);
      if (!all_visible || kfirst + 64 > p.s_k) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          const int base = kfirst + sb * 32 + 4 * fg;
          mask_frag(sacc[sb], myq + off - base, p.sep_k - base, p.s_k - base, masked_raw);
        }
      }
      // row maximum of the 32 scores: v_max3_f32 (the maxnum of fmaxf costs an extra canonicalising v_max per operand)
      float mb = sacc[0][0];
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mb) : "v"(mb), "v"(sacc[1][0]), "v"(sacc[0][1]));
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mb) : "v"(mb), "v"(sacc[1][1]), "v"(sacc[0][2]));
#pragma unroll
      for (int e = 2; e < 16; e += 1) {
        if (e < 15) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mb) : "v"(mb), "v"(sacc[1][e]), "v"(sacc[0][e + 1]));
        else asm("v_max_f32 %0, %1, %2" : "=v"(mb) : "v"(mb), "v"(sacc[1][e]));
      }
      {
        const float other = __shfl_xor(mb, 32, 64);
        asm("v_max_f32 %0, %1, %2" : "=v"(mb) : "v"(mb), "v"(other));
      }
      mb *= sl2;
      if (!__all(mb <= m_run)) {               // running max grew for some row of this wave: rescale (rare later on)
        const float m_new = fmaxf(m_run, mb);
        const float alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
      }
      float ls = 0.f;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float pv = fast_exp2(fmaf(sacc[sb][e], sl2, kofs - m_run)); sacc[sb][e] = pv; ls += pv; }
      l_run += ls;
      if (drop) {
        uint32_t kw = 0u;                          // DROP == 2: the 32 keep bits of this lane's keys, first element in bit 31
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const u32x2 r = attn_bits_at(cb, (uint32_t)((kb * 64 + sb * 32 + 8 * gq + 4 * fg) >> 2));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool kp = keep_of(r, i, p.thr16);
              sacc[sb][4 * gq + i] = kp ? sacc[sb][4 * gq + i] : 0.f;
              if (DROP == 2) {       // kw = 2 kw + keep as ONE v_addc_co_u32: the compare's lane mask is the carry-in (left to
                                     // itself the compiler builds the word with two selects, an or and a shift per pair)
                const unsigned long long cm = __ballot(kp);
                asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(kw) : "s"(cm) : "vcc");
              }
            }
          }
        if (DROP == 2 && myq < p.s_q) kwp[(long long)kb * 2 * p.s_q] = kw;
      }
      // O^T[d][query] += V^T[d][key] . P^T[key][query]
      TrRaw vr2[2][2];
      tr_wait(vr[0][0], vr[0][1]); tr_wait(vr[1][0], vr[1][1]);
      tr_frags_issue<T, 1>(lv, loff, vr2);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float pe[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pe[e] = sacc[0][8 * t + e];
        const typename HT<T>::v8 pb = cvt8<T>(pe);
#pragma unroll
        for (int d = 0; d < 2; ++d) oacc[d] = HT<T>::mfma32(tr_pack<T>(vr[t][d]), pb, oacc[d]);
      }
      tr_wait(vr2[0][0], vr2[0][1]); tr_wait(vr2[1][0], vr2[1][1]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float pe[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pe[e] = sacc[1][8 * t + e];
        const typename HT<T>::v8 pb = cvt8<T>(pe);
#pragma unroll
        for (int d = 0; d < 2; ++d) oacc[d] = HT<T>::mfma32(tr_pack<T>(vr2[t][d]), pb, oacc[d]);
      }
    }
    st = (st == NSTG - 1) ? 0 : st + 1;
  }
  wait_vmcnt<0>();
  if (wave_active && myq < p.s_q) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);       // = keep_scale * sum of probabilities
    const float inv = (drop ? p.keep_scale : 1.0f) / l_tot;
    if (fg == 0 && p.lse) p.lse[((long long)b * p.H + head) * p.s_q + myq] = (m_run + __builtin_amdgcn_logf(l_tot) - kofs) * 0.6931471805599453f;
    T* O = reinterpret_cast<T*>(p.o) + b * p.o_bs + (long long)myq * p.o_rs + head * HD;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        u32x2 w;
        w[0] = pack2<T>(oacc[d][4 * gq] * inv, oacc[d][4 * gq + 1] * inv);
        w[1] = pack2<T>(oacc[d][4 * gq + 2] * inv, oacc[d][4 * gq + 3] * inv);
        *reinterpret_cast<u32x2*>(O + d * 32 + 8 * gq + 4 * fg) = w;
      }
  }
}

// =====================================================================================================
// dQ: grid (ceil(s_q/128), H, B); lane = query.   dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q]
// Ring stage = K tile | V tile (K serves both S^T (natural read) and dQ^T (transposing read)).
// =====================================================================================================
// DROP == 2: the keep bits come from p.keepbits (written by the forward kernel): every wave DMAs the 64 words of its 32
// queries x 2 key halves for the stage's key block into 256 bytes behind the stage's tiles (one more LDS-DMA per stage and wave,
// certified by the same counted wait) and each lane reads its own word back.
template <typename T, bool IDX, int DROP>
__global__ __launch_bounds__(NT, 2) void attn_bwd_dq_kernel(const AttnArgs p) {
  const bool drop = DROP < 0 ? (p.thr16 != 0u) : (DROP != 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool KB = DROP == 2;
  constexpr int STAGE = 2 * TILE + (KB ? 1024 : 0), LPT = KB ? 5 : 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 31, fg = lane >> 5;
  const int nblk_x = (p.s_q + 127) >> 7;
  const BlockId bid = xcd_block_id(nblk_x, p.H, p.B, true);
  if (!bid.ok) return;
  const int b = bid.z, head = bid.head;
  const int q0 = bid.bx * 128, q0w = q0 + wave * 32;
  const bool spw = IDX && p.sp_w > 0;               // sparse training form (slot space), see attn_fwd_kernel
  const int gblk = spw ? q0 / p.sp_w : 0;
  const int off = spw ? (p.s_k - p.sp_w - gblk * p.sp_w) : (p.s_k - p.s_q);
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.q_bs + head * HD;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.k_bs + head * HD;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.v_bs + head * HD;
  const T* DO = reinterpret_cast<const T*>(p.dout) + b * p.do_bs + head * HD;
  const int myq = q0w + fr;
  const bool qvalid = myq < p.s_q;
  const bool wave_active = q0w < p.s_q;

  typename HT<T>::v8 qf[4], dof[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    qf[t] = load_frag_global<T>(Q + (long long)myq * p.q_rs + 16 * t + 8 * fg, qvalid);
    dof[t] = load_frag_global<T>(DO + (long long)myq * p.do_rs + 16 * t + 8 * fg, qvalid);
  }
  const long long arow = ((long long)b * p.H + head) * p.s_q + myq;
  const RowKey cb = row_key(p.rng_key, (unsigned long long)arow);
  const float kscale = drop ? p.keep_scale : 1.0f;
  const float lse2 = qvalid ? p.lse[arow] * 1.4426950408889634f : 0.f;
  // D[q] = sum_d dO[q][d] O[q][d] (the softmax-backward row term): computed here from the dO fragments this lane
  // already holds (+ the matching O fragments), published for the dK/dV kernel that runs next -- no separate pass
  float dv = 0.f;
  {
    const T* Op = reinterpret_cast<const T*>(p.o) + b * p.o_bs + head * HD;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const typename HT<T>::v8 of = load_frag_global<T>(Op + (long long)myq * p.o_rs + 16 * t + 8 * fg, qvalid);
#pragma unroll
      for (int e = 0; e < 8; ++e) dv = fmaf((float)dof[t][e], (float)of[e], dv);
    }
    dv += __shfl_xor(dv, 32, 64);
    if (qvalid && fg == 0) {
      // published NEGATED (round 6): the dK/dV kernel forms dS = Pd dPd + P (-D) without a sign flip per element
      p.dvec[arow] = -dv;
      // second plane of the workspace: the row's dropout key, so the dK/dV kernel (lane = key, 16 query rows per lane)
      // reads it instead of re-hashing the row for every draw.  With stored keep bits that plane is free and carries
      // -LSE * log2(e) instead: the exponent's addend as the dK/dV kernel needs it (one multiply per score element less there)
      if (!KB) reinterpret_cast<uint32_t*>(p.dvec)[(long long)p.B * p.H * p.s_q + arow] = cb.rk;
      else p.dvec[(long long)p.B * p.H * p.s_q + arow] = -lse2;
    }
  }
  const float sl2 = p.scale * 1.4426950408889634f;
  const float masked_raw = MASKED / p.scale;

  const int q_last = min(p.s_q, q0 + 128) - 1;
  const int kend_blk = min(p.s_k, max(q_last + off + 1, p.sep_k));
  const int nkb = (kend_blk + 63) >> 6;
  const int qw_last = min(p.s_q, q0w + 32) - 1;
  const int kend_w = wave_active ? min(p.s_k, max(qw_last + off + 1, p.sep_k)) : 0;

  f32x16 dqacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[d][e] = 0.f;
  const uint32_t loff[2] = {tr_lane_off(0, lane) ^ tr_lane_fix(lane), tr_lane_off(1, lane) ^ tr_lane_fix(lane)};
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;

  int* lidx = nullptr;
  if (IDX && p.kv_index) {
    lidx = reinterpret_cast<int*>(smem + NSTG * STAGE);
    const int* gi = p.kv_index + (long long)b * p.kv_index_bs + (long long)gblk * p.kv_index_gs;
    for (int i = threadIdx.x; i < p.s_k; i += NT) lidx[i] = gi[i];
    __syncthreads();
  }
  const float sp_bias_raw = spw ? p.sp_bias / p.scale : 0.f;
  const uint32_t* KWQ = KB ? p.keepbits + ((long long)b * p.H + head) * ((p.s_k + 63) >> 6) * 2 * p.s_q + (long long)fg * p.s_q + min(myq, p.s_q - 1)
                           : nullptr;
  auto issue = [&](int kb, int st) {
    dma_tile<T>(K, p.k_rs, kb * 64, p.s_k, smem + st * STAGE, wave, lane, lidx);
    dma_tile<T>(V, p.v_rs, kb * 64, p.s_k, smem + st * STAGE + TILE, wave, lane, lidx);
    if (KB) __builtin_amdgcn_global_load_lds((gbl_void_t*)(KWQ + (long long)kb * 2 * p.s_q),
                                             (lds_void_t*)(smem + st * STAGE + 2 * TILE + wave * 256), 4, 0, 0);
  };
  if (nkb > 0) { issue(0, 0); if (NSTG > 2) issue(nkb > 1 ? 1 : 0, 1); }
  int st = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    wait_vmcnt<(NSTG - 2) * LPT>();
    __builtin_amdgcn_s_barrier();
    issue(min(kb + NSTG - 1, nkb - 1), st == 0 ? NSTG - 1 : st - 1);
    if (kb * 64 < kend_w) {
      const char* lk = smem + st * STAGE; const char* lv = lk + TILE;
      const uint32_t lkt = smem_addr + st * STAGE;
      uint32_t kwv = 0u;                                    // this lane's 32 keep bits of the block (forward's order)
      if (KB) {
        const uint32_t ka = lkt + 2 * TILE + wave * 256 + lane * 4;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(kwv) : "v"(ka) : "memory");
      }
      unsigned long long mflag = 0ull, mpiv = 0ull;         // slot attributes of this block (sparse training form)
      if (spw) {
        int raw;
        const uint32_t ia = (uint32_t)(uintptr_t)(lidx + min(kb * 64 + lane, p.s_k - 1));
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(raw) : "v"(ia) : "memory");
        mflag = __ballot(raw < 0); mpiv = __ballot(kb * 64 + lane < p.sp_npiv);
      }
      auto half = [&](auto SBc) {
        constexpr int sb = decltype(SBc)::value;
        f32x16 sacc, pacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sacc = HT<T>::mfma32(nat_frag<T>(lk, sb * 32 + fr, 2 * t + fg), qf[t], sacc);     // S^T
          pacc = HT<T>::mfma32(nat_frag<T>(lv, sb * 32 + fr, 2 * t + fg), dof[t], pacc);    // dP^T = V dO^T
        }
        TrRaw kr[2][2];
        tr_frags_issue<T, sb>(lkt, loff, kr);
        if (spw) {
          const uint32_t fm = (uint32_t)(mflag >> (32 * sb)) >> (4 * fg), pm = (uint32_t)(mpiv >> (32 * sb)) >> (4 * fg);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int bit = (e & 3) + 8 * (e >> 2);
            float v = sacc[e];
            v = ((fm >> bit) & 1u) ? masked_raw : v;
            v += ((pm >> bit) & 1u) ? sp_bias_raw : 0.f;
            sacc[e] = v;
          }
        }
        const int kfirst = kb * 64 + sb * 32;
        float mk[16];                                          // arbitrary mask tensor (see attn_fwd_kernel): d raw = d raw' * M
        if (IDX && p.mask) {
          const T* mrow = reinterpret_cast<const T*>(p.mask) + (long long)b * p.mask_bs + (long long)min(myq, p.s_q - 1) * p.s_k;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = kfirst + 4 * fg + (e & 3) + 8 * (e >> 2);
            mk[e] = HT<T>::to_f(mrow[min(key, p.s_k - 1)]);
            sacc[e] = fmaf(sacc[e], mk[e], masked_raw * (1.f - mk[e]));
          }
        }
        const bool all_visible = (kfirst + 31 <= q0w + off) || (kfirst + 31 < p.sep_k);
        if (!all_visible || kfirst + 32 > p.s_k) {
          const int base = kfirst + 4 * fg;
          mask_frag(sacc, myq + off - base, p.sep_k - base, p.s_k - base, masked_raw);
        }
        float ds[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          u32x2 r = {0u, 0u};
          if (drop && !KB) r = attn_bits_at(cb, (uint32_t)((kfirst + 8 * gq + 4 * fg) >> 2));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 4 * gq + i;
            const float pr = fast_exp2(fmaf(sacc[e], sl2, -lse2));
            float dp = pacc[e];
            if (KB) {         // keep bit sign-extended over the float's bit pattern: v_bfe_i32 + v_and (through asm: with a
                              // constant position the compiler prefers and + compare + select, three instructions and a VCC chain)
              uint32_t sx;
              asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sx) : "v"(kwv), "n"(31 - (sb * 16 + e)));
              dp = __uint_as_float(sx & __float_as_uint(dp));
            }
            else if (drop) dp = keep_of(r, i, p.thr16) ? dp : 0.f;
            ds[e] = pr * fmaf(dp, kscale, -dv);                  // kscale = 1 / (1 - p) (1 without dropout)
            if (IDX && p.mask) ds[e] *= mk[e];
          }
        }
        tr_wait(kr[0][0], kr[0][1]); tr_wait(kr[1][0], kr[1][1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const typename HT<T>::v8 dsb = cvt8<T>(ds + 8 * t);
#pragma unroll
          for (int d = 0; d < 2; ++d) dqacc[d] = HT<T>::mfma32(tr_pack<T>(kr[t][d]), dsb, dqacc[d]);
        }
      };
      half(std::integral_constant<int, 0>{});
      half(std::integral_constant<int, 1>{});
    }
    st = (st == NSTG - 1) ? 0 : st + 1;
  }
  wait_vmcnt<0>();
  float rq[2][16];
  {
    T* DQ = reinterpret_cast<T*>(p.dq) + b * p.dq_bs + (long long)myq * p.dq_rs + head * HD;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        u32x2 w;
        w[0] = pack2<T>(dqacc[d][4 * gq] * p.scale, dqacc[d][4 * gq + 1] * p.scale);
        w[1] = pack2<T>(dqacc[d][4 * gq + 2] * p.scale, dqacc[d][4 * gq + 3] * p.scale);
        if (qvalid) *reinterpret_cast<u32x2*>(DQ + d * 32 + 8 * gq + 4 * fg) = w;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          rq[d][4 * gq + i] = qvalid ? bits_to_f<T>((uint16_t)(w[i >> 1] >> (16 * (i & 1)))) : 0.f;
      }
  }
  if (p.colsum_ws) {     // bias gradient of the QKV projection, q section
    __syncthreads();     // every wave is done with the ring
    float* dst = p.colsum_ws + ((size_t)b * nblk_x + bid.bx) * (size_t)(3 * p.H * HD) + head * HD;
    tile_colsum(rq, reinterpret_cast<float*>(smem), dst, lane, wave);
  }
}

// =====================================================================================================
// dK/dV: grid (ceil(s_k/128), H, B); lane = key.
//   dV^T[d][key] = sum_q dO^T[d][q] Pd[q][key]        dK^T[d][key] = scale * sum_q Q^T[d][q] dS[q][key]
// Ring stage = Q tile | dO tile | LSE[64] | D[64]  (64 queries per stage; Q and dO each serve a natural and a
// transposing read).
// =====================================================================================================
// DROP == 2: keep bits from p.keepbits.  The workgroup's 128 keys are two 64-key blocks x two key halves = four segments of
// 64 words (one per query of the stage); wave w DMAs segment w in place of the row-key statistics row (same count of
// DMAs per stage), and a lane (= key) picks ITS bit out of the word of each of its 16 queries: four ds_read_b128 +
// 16 x (v_bfe_i32, v_and) per 32 x 32 tile where the regenerating form hashes four draws and shares them through DPP.
// Probe builds only (-DCOGV_ATTN_TS, tools/probes/attn_ts.py): per-wave shader-clock time of the phases of the dK.dV kernel, summed
// over all waves of the launch into g_attn_ts: 0 DMA wait + barrier, 1 DMA issue, 2 the stage's compute (both halves), 3 epilogue
// (stores, column sums), 4 prologue, 5 whole wave, 6 wave-iterations, 7 waves.
#if defined(COGV_ATTN_TS)
__device__ unsigned long long g_attn_ts[1024 * 8];      // 1024 slots of 8 counters (one hot line would serialise 276k atomics per launch)
#define ATS(k_) do { const unsigned long long t_ = __builtin_readcyclecounter(); ats[k_] += t_ - ats_last; ats_last = t_; } while (0)
#else
#define ATS(k_) do { } while (0)
#endif
template <typename T, bool IDX, int DROP>
__global__ __launch_bounds__(NT, (!IDX && DROP == 2) ? COGV_DKDV_WAVES : 2) void attn_bwd_dkdv_kernel(const AttnArgs p) {
  const bool drop = DROP < 0 ? (p.thr16 != 0u) : (DROP != 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool KB = DROP == 2;
  // KWP: pitch (bytes) of the four keep-word segments of a stage.  272 instead of 256 (round 6): a lane group reads two segments
  // in one ds_read_b128 (its key's bit 2 selects which), and at a 256-byte pitch the two hit the same banks -- the 12 % LDS
  // bank-conflict cycles of this kernel alone among the three (profiles/r05_attention_pmc.txt)
  constexpr int KWP = COGV_DKDV_KW_PITCH;
  constexpr int STAGE = 2 * TILE + (KB ? 512 + 4 * KWP : 768), LPT = 7;
#if defined(COGV_ATTN_TS)
  unsigned long long ats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long ats_begin = __builtin_readcyclecounter();
  unsigned long long ats_last = ats_begin;
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 31, fg = lane >> 5;
  // sparse training form (sp_w > 0): grid z = (b, query block g); keys are the block's slots, queries its sp_w rows,
  // dK / dV go to slot-space buffers [z][slot] which cogv_sparse_slot_reduce folds back onto the keys
  const bool spw = IDX && p.sp_w > 0;
  const int nblk = spw ? p.s_q / p.sp_w : 1;
  const int nblk_x = (p.s_k + 127) >> 7;
  const BlockId bid = xcd_block_id(nblk_x, p.H, p.B * nblk, false);       // key block 0 has the most queries: first
  if (!bid.ok) return;
  const int zb = bid.z, head = bid.head;
  const int b = spw ? zb / nblk : zb, gblk = spw ? zb - b * nblk : 0;
  const int k0 = bid.bx * 128, k0w = k0 + wave * 32;
  const int off = spw ? (p.s_k - p.sp_w - gblk * p.sp_w) : (p.s_k - p.s_q);
  const int qlo = gblk * p.sp_w, qhi = spw ? qlo + p.sp_w : p.s_q;
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.q_bs + head * HD;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.k_bs + head * HD;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.v_bs + head * HD;
  const T* DO = reinterpret_cast<const T*>(p.dout) + b * p.do_bs + head * HD;
  const float* LSE = p.lse + ((long long)b * p.H + head) * p.s_q;
  const float* DV = p.dvec + ((long long)b * p.H + head) * p.s_q;
  const float* RK = DV + (long long)p.B * p.H * p.s_q;            // row keys of the dropout generator (written by the dQ kernel)
  const int mykey = k0w + fr;
  const bool kvalid = mykey < p.s_k;
  const bool wave_active = k0w < p.s_k;

  int krow = mykey;
  bool kflag = false;
  if (IDX && p.kv_index) {
    const int raw = kvalid ? p.kv_index[(long long)b * p.kv_index_bs + (long long)gblk * p.kv_index_gs + mykey] : 0;
    kflag = raw < 0; krow = raw & 0x7fffffff;
  }
  typename HT<T>::v8 kf[4], vf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    kf[t] = load_frag_global<T>(K + (long long)krow * p.k_rs + 16 * t + 8 * fg, kvalid);
    vf[t] = load_frag_global<T>(V + (long long)krow * p.v_rs + 16 * t + 8 * fg, kvalid);
  }
  // (diet) the V fragments parked in LDS behind the ring, one 16-byte slot per (wave, t, lane): lane-private, so the in-order LDS
  // queue is all the ordering the write and the later reads need
  constexpr bool VLDS = KB && !IDX && (COGV_DKDV_VLDS != 0);
  typename HT<T>::v8* const own_v = reinterpret_cast<typename HT<T>::v8*>(smem + ring_bytes(STAGE, true)) + (wave * 4) * 64 + lane;
  if (VLDS) {
#pragma unroll
    for (int t = 0; t < 4; ++t) own_v[t * 64] = vf[t];
  }
  const int qbeg_blk = (k0 < p.sep_k) ? qlo : max(qlo, k0 - off);
  const int qbeg_w = (k0w < p.sep_k) ? qlo : max(qlo, k0w - off);
  const int qb0 = qbeg_blk >> 6;
  const int nqb = (qhi + 63) >> 6;
  const float sp_add = (spw && mykey < p.sp_npiv) ? p.sp_bias / p.scale : 0.f;
  const float sl2 = p.scale * 1.4426950408889634f;
  const float l2e = 1.4426950408889634f;
  const float masked_raw = MASKED / p.scale;
  const long long arow0 = ((long long)b * p.H + head) * p.s_q;
  const float kscale = drop ? p.keep_scale : 1.0f;
  const uint32_t gweyl = (uint32_t)(mykey >> 2) * 0x9E3779B9u;       // the lane's key group, as the generator's Weyl offset

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[d][e] = 0.f; dvacc[d][e] = 0.f; }
  const uint32_t loff[2] = {tr_lane_off(0, lane) ^ tr_lane_fix(lane), tr_lane_off(1, lane) ^ tr_lane_fix(lane)};
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;

  // keep-bit segment of this wave: key block k0 / 64 + (wave >> 1), key half wave & 1 (clamped: words of key blocks past
  // the end are never used)
  const int nkbt = (p.s_k + 63) >> 6;
  const uint32_t* KWS = KB ? p.keepbits + ((((long long)b * p.H + head) * nkbt + min((k0 >> 6) + (wave >> 1), nkbt - 1)) * 2 + (wave & 1)) * p.s_q
                           : nullptr;
  auto issue = [&](int qb, int st) {
    char* base = smem + st * STAGE;
    dma_tile<T>(Q, p.q_rs, qb * 64, p.s_q, base, wave, lane);
    dma_tile<T>(DO, p.do_rs, qb * 64, p.s_q, base + TILE, wave, lane);
    dma_stat(KB ? RK : LSE, qb * 64, p.s_q, base + 2 * TILE, lane);      // KB: plane 1 holds -LSE * log2(e) (dQ kernel)
    dma_stat(DV, qb * 64, p.s_q, base + 2 * TILE + 256, lane);
    if (KB) dma_stat(reinterpret_cast<const float*>(KWS), qb * 64, p.s_q, base + 2 * TILE + 512 + wave * KWP, lane);
    else dma_stat(RK, qb * 64, p.s_q, base + 2 * TILE + 512, lane);
  };
  // the lane's key inside its 64-key block: half (wave & 1 -- the wave's 32 keys), key half fg' = bit 2, element 4 (r >> 3) + (r & 3)
  const int kr = mykey & 31;
  const uint32_t kw_seg = (uint32_t)(((wave >> 1) * 2 + ((kr >> 2) & 1)) * (KWP / 4));        // word offset of the lane's segment
  const uint32_t kw_bit = 31u - (uint32_t)((wave & 1) * 16 + 4 * (kr >> 3) + (kr & 3));
  if (qb0 < nqb) { issue(qb0, 0); if (NSTG > 2) issue(min(qb0 + 1, nqb - 1), 1); }
  int st = 0;
  ATS(4);
  for (int qb = qb0; qb < nqb; ++qb) {
    wait_vmcnt<(NSTG - 2) * LPT>();
    __builtin_amdgcn_s_barrier();
    ATS(0);
    issue(min(qb + NSTG - 1, nqb - 1), st == 0 ? NSTG - 1 : st - 1);
    ATS(1);
#if defined(COGV_ATTN_TS)
    ats[6] += 1;
#endif
    if (wave_active && qb * 64 + 63 >= qbeg_w) {
      const char* lq = smem + st * STAGE; const char* ldo = lq + TILE;
      const uint32_t lqt = smem_addr + st * STAGE, ldot = lqt + TILE;
      const float* stat = reinterpret_cast<const float*>(lq + 2 * TILE);
      auto half = [&](auto SBc) {
        constexpr int sb = decltype(SBc)::value;
        f32x16 sacc, pacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
        // KB (stored keep bits: ~60 registers below the limit): the transposed dO / Q fragments of the dV / dK products are
        // requested NOW and land behind the S / dP products and the element-wise chain; the regenerating form (register
        // pressure at the limit) reads them synchronously where they are used -- two exposed LDS round trips per half
        TrRaw dor_a[2][2], qr_a[2][2];
        if (KB && COGV_DKDV_AHEAD) { tr_frags_issue<T, sb>(ldot, loff, dor_a); tr_frags_issue<T, sb>(lqt, loff, qr_a); }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sacc = HT<T>::mfma32(nat_frag<T>(lq, sb * 32 + fr, 2 * t + fg), kf[t], sacc);     // S = Q K^T
          pacc = HT<T>::mfma32(nat_frag<T>(ldo, sb * 32 + fr, 2 * t + fg), VLDS ? own_v[t * 64] : vf[t], pacc);    // dPd = dO V^T
        }
        // element e <-> query qfirst + 4fg + (e&3) + 8(e>>2); key fixed per lane
        const int qfirst = qb * 64 + sb * 32;
        const bool all_visible = (qfirst >= k0w + 31 - off) || (k0w + 31 < p.sep_k);
        const bool tail = qfirst + 32 > p.s_q;
        // Dropout bits: the generator's group is (query row, 4 consecutive keys).  The 4 lanes of a quad hold the 4
        // keys of one group, so lane c of the quad hashes only the queries with (e & 3) == c, reduces each 64-bit
        // draw to a 4-bit keep mask (bit f <-> key 4m+f) and the quad shares it by a broadcast DPP move:
        // 4 hashes per lane per 32x32 tile instead of 16.
        // masking only in blocks that touch the diagonal / the sequence end (wave-uniform branch): masked scores become
        // the raw value whose scaled score is -10000, queries past the end -inf (probability exactly 0)
        if (!all_visible || tail) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int q = qfirst + 4 * fg + (e & 3) + 8 * (e >> 2);
            float a = sacc[e];
            if (!visible(q, mykey, off, p.sep_k)) a = masked_raw;
            if (q >= p.s_q) a = -INFINITY;
            sacc[e] = a;
          }
        }
        if (spw) {
#pragma unroll
          for (int e = 0; e < 16; ++e) sacc[e] = (kflag ? masked_raw : sacc[e]) + sp_add;
        }
        float mq[16];                                          // arbitrary mask tensor: column `mykey` of M, the lane's 16 queries
        if (IDX && p.mask) {
          const T* mcol = reinterpret_cast<const T*>(p.mask) + (long long)b * p.mask_bs + min(mykey, p.s_k - 1);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int q = qfirst + 4 * fg + (e & 3) + 8 * (e >> 2);
            mq[e] = HT<T>::to_f(mcol[(long long)min(q, p.s_q - 1) * p.s_k]);
            if (q < p.s_q) sacc[e] = fmaf(sacc[e], mq[e], masked_raw * (1.f - mq[e]));
          }
        }
        uint32_t kmask[4] = {0u, 0u, 0u, 0u};
        if (drop && !KB) {
          const int c = lane & 3;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            // row q = qb * 64 + sb * 32 + 8 gq + 4 fg + c of this (batch, head): its key comes from the stage's third
            // statistics row; group = the lane's keys
            const uint32_t rkq = reinterpret_cast<const uint32_t*>(stat)[128 + sb * 32 + 8 * gq + 4 * fg + c];
            const u32x2 r = attn_bits_w(rkq + gweyl);
            uint32_t m4 = 0;
#pragma unroll
            for (int f = 0; f < 4; ++f) m4 |= (keep_of(r, f, p.thr16) ? 1u : 0u) << f;
            kmask[gq] = m4;
          }
        }
        const int kbit = mykey & 3;
        float pd[16], ds[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int ql = sb * 32 + 8 * gq + 4 * fg;                // local query of element i = 0
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(stat + ql);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(stat + 64 + ql);
          uint32_t km[4];
          if (KB) {                                             // the words of the 4 queries of this group: one b128 read
            const u32x4 w4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint32_t*>(stat) + 128 + kw_seg + ql);
            km[0] = w4[0]; km[1] = w4[1]; km[2] = w4[2]; km[3] = w4[3];
          } else if (drop) {                                    // quad_perm(i,i,i,i): value held by quad lane i
            km[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)kmask[gq], 0x00, 0xf, 0xf, true);
            km[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)kmask[gq], 0x55, 0xf, 0xf, true);
            km[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)kmask[gq], 0xaa, 0xf, 0xf, true);
            km[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)kmask[gq], 0xff, 0xf, 0xf, true);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 4 * gq + i;
            // (the exponent's argument of two neighbouring elements as ONE packed fma measured neutral to slightly slower,
            //  profiles/r06_attention_dkdv_valu_diet_ab.log: not kept)
            const float pr = fast_exp2(fmaf(sacc[e], sl2, KB ? l4[i] : -l4[i] * l2e));
            // dropped probability Pd = keep ? P / (1 - p) : 0 and dS = Pd dPd - P D.  The keep bit becomes the
            // multiplier 1/(1-p) or 0 by sign-extending it over the float's bit pattern (v_bfe_i32 + v_and): two
            // instructions where compare + two selects (on Pd and dS) took five
            float keepf = kscale;
            if (drop) keepf = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe((int)km[i], KB ? kw_bit : (uint32_t)kbit, 1u) & __float_as_uint(kscale));
            pd[e] = pr * keepf;
            ds[e] = fmaf(pd[e], pacc[e], pr * d4[i]);                 // d4 = -D (published negated by the dQ kernel)
            if (IDX && p.mask) ds[e] *= mq[e];
          }
        }
        if (!kvalid) {
#pragma unroll
          for (int e = 0; e < 16; ++e) { pd[e] = 0.f; ds[e] = 0.f; }
        }
        typename HT<T>::v8 pb[2], dsb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { pb[t] = cvt8<T>(pd + 8 * t); dsb[t] = cvt8<T>(ds + 8 * t); }
        if (KB && COGV_DKDV_AHEAD) {
          tr_wait(dor_a[0][0], dor_a[0][1]); tr_wait(dor_a[1][0], dor_a[1][1]);
          tr_wait(qr_a[0][0], qr_a[0][1]); tr_wait(qr_a[1][0], qr_a[1][1]);
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int d = 0; d < 2; ++d) dvacc[d] = HT<T>::mfma32(tr_pack<T>(dor_a[t][d]), pb[t], dvacc[d]);
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int d = 0; d < 2; ++d) dkacc[d] = HT<T>::mfma32(tr_pack<T>(qr_a[t][d]), dsb[t], dkacc[d]);
        } else {
          {
            TrRaw dor[2][2];
            tr_frags_sync<T, sb>(ldot, loff, dor);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int d = 0; d < 2; ++d) dvacc[d] = HT<T>::mfma32(tr_pack<T>(dor[t][d]), pb[t], dvacc[d]);
          }
          {
            TrRaw qr[2][2];
            tr_frags_sync<T, sb>(lqt, loff, qr);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int d = 0; d < 2; ++d) dkacc[d] = HT<T>::mfma32(tr_pack<T>(qr[t][d]), dsb[t], dkacc[d]);
          }
        }
      };
      // (Round 6 measured the two halves as a software pipeline INSIDE the wave -- the S / dP products of half 1 issued between the
      //  element-wise instructions of half 0, the dV / dK products of half 0 between those of half 1, one MFMA per ~20 VALU
      //  instructions pinned by sched_barrier, 249 registers, no spill, same results: backward 992-1029 us against 987-1006
      //  (profiles/r06_attention_dkdv_intra_wave_pipeline_ab.log).  In-kernel timestamps (profiles/r06_attention_dkdv_phase_probe.log)
      //  say why: a wave spends 57 % of its life in this compute phase and two waves share the SIMD, so the MFMAs of one wave
      //  already run under the VALU work of the other; the SIMD is bound by VALU issue.  Removed.)
      half(std::integral_constant<int, 0>{});
      half(std::integral_constant<int, 1>{});
    }
    ATS(2);
    st = (st == NSTG - 1) ? 0 : st + 1;
  }
  wait_vmcnt<0>();
  float rk[2][16], rv[2][16];
  {
    T* DK = reinterpret_cast<T*>(p.dk) + zb * p.dk_bs + (long long)mykey * p.dk_rs + head * HD;
    T* DVp = reinterpret_cast<T*>(p.dv) + zb * p.dv_bs + (long long)mykey * p.dv_rs + head * HD;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        u32x2 w, w2;
        w[0] = pack2<T>(dkacc[d][4 * gq] * p.scale, dkacc[d][4 * gq + 1] * p.scale);
        w[1] = pack2<T>(dkacc[d][4 * gq + 2] * p.scale, dkacc[d][4 * gq + 3] * p.scale);
        if (kvalid) *reinterpret_cast<u32x2*>(DK + d * 32 + 8 * gq + 4 * fg) = w;
        w2[0] = pack2<T>(dvacc[d][4 * gq], dvacc[d][4 * gq + 1]);
        w2[1] = pack2<T>(dvacc[d][4 * gq + 2], dvacc[d][4 * gq + 3]);
        if (kvalid) *reinterpret_cast<u32x2*>(DVp + d * 32 + 8 * gq + 4 * fg) = w2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rk[d][4 * gq + i] = kvalid ? bits_to_f<T>((uint16_t)(w[i >> 1] >> (16 * (i & 1)))) : 0.f;
          rv[d][4 * gq + i] = kvalid ? bits_to_f<T>((uint16_t)(w2[i >> 1] >> (16 * (i & 1)))) : 0.f;
        }
      }
  }
  if (p.colsum_ws) {     // bias gradient of the QKV projection, k and v sections
    __syncthreads();
    float* dst = p.colsum_ws + ((size_t)zb * nblk_x + bid.bx) * (size_t)(3 * p.H * HD) + (size_t)p.H * HD + head * HD;
    tile_colsum(rk, reinterpret_cast<float*>(smem), dst, lane, wave);
    tile_colsum(rv, reinterpret_cast<float*>(smem), dst + (size_t)p.H * HD, lane, wave);
  }
#if defined(COGV_ATTN_TS)
  ATS(3);
  if (lane == 0) {
    ats[5] = __builtin_readcyclecounter() - ats_begin; ats[7] = 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_attn_ts[(((blockIdx.x << 2) + wave) & 1023) * 8 + k], ats[k]);
  }
#endif
}

// =====================================================================================================
// Decode step: ONE query token per batch row against a fixed-capacity key/value cache (generation/sampling.py:139-148,
// one model call per generated token).  The 128-query tile kernels above spend a whole workgroup per (batch, head) on
// one row (40 workgroups on 256 CUs, 34 us per layer at 1152 slots); this kernel splits the KEYS: grid (capacity / 128,
// H, B), 8 lanes per key (16 bytes of the 128-byte row each: coalesced), a partial (max, sum, output) per split; a
// second small launch combines the partials in split order (deterministic).  The new token's key / value come straight from the QKV GEMM's
// output row and are written into the cache slot *pos by the split that owns it: the cat + index_copy_ launches of the
// generic cache append disappear.  Valid slots are [0, *pos] (device data: no launch parameter depends on the length).
struct DecodeArgs {
  const void* qkv; void* cache; void* out; const long long* pos; float* ws; int* tickets;
  long long qkv_bs, cache_bs, out_bs; int cache_rs;
  int B, H, cap, nsplit; float scale_l2e;
};
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const DecodeArgs p) {
  __shared__ float red_o[32][64];
  __shared__ float red_m[4], red_l[32];
  const int t = threadIdx.x, dch = t & 7, kg = t >> 3, lane = t & 63, wave = t >> 6;
  const int split = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int hp = p.H * HD;
  const T* qrow = reinterpret_cast<const T*>(p.qkv) + b * p.qkv_bs + head * HD + dch * 8;
  float q8[8], kn[8], vn[8];
  unpack8<T>(*reinterpret_cast<const u32x4*>(qrow), q8);
  const u32x4 knew = *reinterpret_cast<const u32x4*>(qrow + hp), vnew = *reinterpret_cast<const u32x4*>(qrow + 2 * hp);
  T* crow = reinterpret_cast<T*>(p.cache) + b * p.cache_bs + head * HD + dch * 8;
  float sc[4]; u32x4 v8[4], kc8[4];
  float m_loc = -INFINITY;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const long long key = (long long)split * 128 + ps * 32 + kg;
    // the cache rows are requested whatever *pos says (slots past it hold stale or no data and are replaced below; a key past
    // the capacity re-reads the last row): the position is a device scalar, and loads that waited for it would start one
    // memory latency late -- the launch is a chain of latencies, not of bytes
    const long long krow = key < p.cap ? key : p.cap - 1;
    // (streamed once per step: non-temporal policy, like the decode step's weight rows -- gemm_shared.cuh: COGV_DECODE_NT)
#if !defined(COGV_DECODE_NT) || COGV_DECODE_NT
    kc8[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(crow + krow * p.cache_rs));
    v8[ps] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(crow + krow * p.cache_rs + hp));
#else
    kc8[ps] = *reinterpret_cast<const u32x4*>(crow + krow * p.cache_rs);
    v8[ps] = *reinterpret_cast<const u32x4*>(crow + krow * p.cache_rs + hp);
#endif
  }
  const long long pos = *p.pos;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const long long key = (long long)split * 128 + ps * 32 + kg;
    const bool valid = key <= pos && key < p.cap;
    const bool cached = valid && key != pos;
    // (bit masks, not a select: the compiler turns a select of a loaded value into a branch and sinks the load into it)
    const uint32_t mk = cached ? 0xffffffffu : 0u;
    const u32x4 k8 = (kc8[ps] & mk) | (knew & ~mk);
    v8[ps] = (v8[ps] & mk) | (vnew & ~mk);
    if (valid && !cached) {                  // the new token's own slot: store it for the steps to come
      *reinterpret_cast<u32x4*>(crow + key * p.cache_rs) = knew;
      *reinterpret_cast<u32x4*>(crow + key * p.cache_rs + hp) = vnew;
    }
    float kf[8]; unpack8<T>(k8, kf);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d = fmaf(q8[e], kf[e], d);
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);     // the key's 8 lanes
    sc[ps] = valid ? d * p.scale_l2e : -INFINITY;
    m_loc = fmaxf(m_loc, sc[ps]);
  }
  m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 8, 64)); m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16, 64)); m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32, 64));
  if (lane == 0) red_m[wave] = m_loc;
  __syncthreads();
  const float m = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
  float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, l = 0.f;
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const float pr = (sc[ps] == -INFINITY) ? 0.f : fast_exp2(sc[ps] - m);
    float vf[8]; unpack8<T>(v8[ps], vf);
    l += pr;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = fmaf(pr, vf[e], o8[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red_o[kg][dch * 8 + e] = o8[e];
  if (dch == 0) red_l[kg] = l;
  __syncthreads();
  float* part = p.ws + (((size_t)b * p.H + head) * p.nsplit + split) * 66;
  if (t < 64) {
    float o = 0.f;
#pragma unroll 8
    for (int g = 0; g < 32; ++g) o += red_o[g][t];
    part[2 + t] = o;
    if (t == 0) {
      float ls = 0.f;
      for (int g = 0; g < 32; ++g) ls += red_l[g];
      part[0] = m; part[1] = ls;
    }
  }
}
// second launch of a decode step: combine the splits' partial (max, sum, output) in split order.  (A device-scope
// release fence + arrival ticket inside the first kernel costs an L2 write-back per workgroup on this part -- measured
// 50 us per layer; the kernel boundary publishes the partials for free.)
template <typename T>
__global__ __launch_bounds__(64) void attn_decode_combine_kernel(const DecodeArgs p) {
  __shared__ float wgt[64];
  const int t = threadIdx.x, head = blockIdx.x, b = blockIdx.y;
  const float* base = p.ws + ((size_t)b * p.H + head) * p.nsplit * 66;
  // lane i holds split i's (max, sum): one independent load each instead of a chain of dependent ones (nsplit <= 32)
  const float mi = t < p.nsplit ? base[t * 66] : -INFINITY;
  const float li = t < p.nsplit ? base[t * 66 + 1] : 0.f;
  // the partial outputs of up to 16 splits are requested together with the statistics (clamped index: unconditional), so the
  // launch pays one memory round trip instead of two
  float ov[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ov[i] = base[(i < p.nsplit ? i : p.nsplit - 1) * 66 + 2 + t];
  float M = mi;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
  const float w = (mi == -INFINITY) ? 0.f : fast_exp2(mi - M);
  float L = li * w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) L += __shfl_xor(L, o, 64);
  wgt[t] = w;
  __syncthreads();
  float O = 0.f;
  if (p.nsplit <= 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) O = i < p.nsplit ? fmaf(ov[i], wgt[i], O) : O;         // same order as the loop below
  } else {
#pragma unroll 8
    for (int i = 0; i < p.nsplit; ++i) O = fmaf(base[i * 66 + 2 + t], wgt[i], O);
  }
  reinterpret_cast<T*>(p.out)[b * p.out_bs + head * HD + t] = HT<T>::from_f(O / L);
}

// Sparse training form: fold the slot-space gradients [B][G][n_slots][H*64] back onto the keys.  Key r is a window
// slot of the blocks g = r/w ... r/w + times - 1 (slot n_piv + r - (g - times + 1) w) and, when it is pivot j
// (pivot_inv[b][r] = j, else -1), slot j of every block whose window starts after it (g >= r/w + times; the blocks
// before that hold it as a masked slot, whose gradient is exactly 0).  fp32 sums in a fixed order: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void sparse_slot_reduce_kernel(const T* __restrict__ dks, const T* __restrict__ dvs,
                                                                 const int* __restrict__ pivot_inv, T* __restrict__ dk,
                                                                 T* __restrict__ dv, long long dk_bs, int dk_rs,
                                                                 long long dv_bs, int dv_rs, int B, int s, int C8,
                                                                 int w, int times, int n_piv) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * s * C8;
  if (tid >= total) return;
  const int c = (int)(tid % C8);
  const long long br = tid / C8;
  const int r = (int)(br % s), b = (int)(br / s);
  const int G = s / w, n_slots = n_piv + times * w, g0 = r / w;
  const long long slot_row = (long long)C8 * 8;
  const int pj = pivot_inv[(long long)b * s + r];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const T* src = (which ? dvs : dks) + (long long)b * G * n_slots * slot_row + c * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    auto add = [&](int g, int slot) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(src + ((long long)g * n_slots + slot) * slot_row);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += bits_to_f<T>((uint16_t)(raw[i >> 1] >> (16 * (i & 1))));
    };
    const int gend = min(G - 1, g0 + times - 1);
    for (int g = g0; g <= gend; ++g) add(g, n_piv + r - (g - times + 1) * w);
    if (pj >= 0)
      for (int g = g0 + times; g < G; ++g) add(g, pj);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack2<T>(acc[2 * i], acc[2 * i + 1]);
    T* dst = which ? dv + b * dv_bs + (long long)r * dv_rs + c * 8 : dk + b * dk_bs + (long long)r * dk_rs + c * 8;
    *reinterpret_cast<u32x4*>(dst) = o;
  }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int fill_args(const cogv_attn_desc* d, AttnArgs& a) {
  if (!d) return COGV_ERR_ARG;
  if (d->dtype != COGV_F16 && d->dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (d->head_dim != HD) return COGV_ERR_UNSUPPORTED;
  if (d->B <= 0 || d->H <= 0 || d->s_q <= 0 || d->s_k <= 0) return COGV_ERR_ARG;
  if (d->s_k < d->s_q && !(d->kv_index && d->sparse_window > 0)) return COGV_ERR_ARG;   // slot space: s_k = slots per block
  if (!(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return COGV_ERR_ARG;
  a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o; a.dout = d->dout; a.dq = d->dq; a.dk = d->dk; a.dv = d->dv;
  a.lse = d->lse; a.dvec = d->dvec; a.colsum_ws = nullptr; a.kv_index = nullptr; a.kv_index_bs = 0;
  a.keepbits = nullptr; a.mask = nullptr; a.mask_bs = 0;
  if (d->mask) {            // arbitrary mask tensor (sep_k is set below: every key is a candidate, the tensor decides)
    if (((uintptr_t)d->mask & 1) || d->kv_index || d->keep_bits) return COGV_ERR_ARG;
    a.mask = d->mask; a.mask_bs = d->mask_bs;
  }
  a.kv_index_gs = 0; a.sp_w = 0; a.sp_npiv = 0; a.sp_bias = 0.f;
  a.q_bs = d->q_bs; a.k_bs = d->k_bs; a.v_bs = d->v_bs; a.o_bs = d->o_bs; a.do_bs = d->do_bs;
  a.dq_bs = d->dq_bs; a.dk_bs = d->dk_bs; a.dv_bs = d->dv_bs;
  a.q_rs = d->q_rs; a.k_rs = d->k_rs; a.v_rs = d->v_rs; a.o_rs = d->o_rs; a.do_rs = d->do_rs;
  a.dq_rs = d->dq_rs; a.dk_rs = d->dk_rs; a.dv_rs = d->dv_rs;
  a.B = d->B; a.H = d->H; a.s_q = d->s_q; a.s_k = d->s_k;
  int sep = d->sep; if (sep < 0) sep = 0;
  a.sep_k = sep > 0 ? sep + (d->s_k - d->s_q) : 0;
  if (a.mask) a.sep_k = d->s_k;
  a.scale = d->scale;
  a.thr16 = (uint32_t)(d->dropout_p * 65536.0f + 0.5f);
  a.keep_scale = 65536.0f / (65536.0f - (float)a.thr16);
  a.rng_key = rng_key(d->seed, d->stream_id);
  return COGV_OK;
}

// gathered keys (kv_index) and the sparse training form's slot attributes
int index_args(const cogv_attn_desc* d, AttnArgs& a) {
  if (d->kv_index) {
    if (a.s_k > 4096) return COGV_ERR_UNSUPPORTED;
    a.kv_index = d->kv_index; a.kv_index_bs = d->kv_index_bs;
    if (d->sparse_window > 0) {          // training form: s_k = slots per query block, queries in blocks of sparse_window
      if ((d->sparse_window % 128) || (a.s_q % d->sparse_window) || d->sparse_pivots < 0 || d->sparse_pivots > a.s_k ||
          a.s_k < d->sparse_window || a.sep_k != 0) return COGV_ERR_ARG;
      a.kv_index_gs = d->kv_index_gs; a.sp_w = d->sparse_window; a.sp_npiv = d->sparse_pivots; a.sp_bias = d->sparse_pivot_bias;
    }
  } else if (d->sparse_window > 0) {
    return COGV_ERR_ARG;
  }
  return COGV_OK;
}

template <typename K>
void set_smem(K kernel, int bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

extern "C" int cogv_attention_fwd(const cogv_attn_desc* d, void* stream) {
  AttnArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  if (!a.q || !a.k || !a.v || !a.o) return COGV_ERR_ARG;
  if (!aligned16(a.q) || !aligned16(a.k) || !aligned16(a.v) || !aligned16(a.o)) return COGV_ERR_ARG;
  if ((a.q_rs | a.k_rs | a.v_rs | a.o_rs) & 7) return COGV_ERR_ARG;
  if ((a.q_bs | a.k_bs | a.v_bs | a.o_bs) & 7) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // one-dimensional XCD-aware grid (xcd_block_id): 8 x ceil(units / 8) x blocks, units = H * B
  dim3 grid(8u * (unsigned)((a.H * a.B + 7) / 8) * (unsigned)((a.s_q + 127) / 128));
  int sh = NSTG * 2 * TILE;
  if ((rc = index_args(d, a))) return rc;
  if (a.kv_index) sh += ((a.s_k * 4 + 15) / 16) * 16;
  // dense kernels: dropout on / off are separate instantiations (no wave-uniform branches and register copies at their
  // joins inside the softmax); the gathered / sparse forms decide at run time
  const bool drop = a.thr16 != 0u;
  if (d->keep_bits && drop && !a.kv_index) {      // dense, dropout on, the caller keeps the bits for the backward pass
    if ((uintptr_t)d->keep_bits & 3) return COGV_ERR_ARG;
    a.keepbits = reinterpret_cast<uint32_t*>(d->keep_bits);
    if (d->dtype == COGV_F16) hipLaunchKernelGGL((attn_fwd_kernel<f16_t, false, 2>), grid, dim3(NT), sh, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, false, 2>), grid, dim3(NT), sh, st, a);
    return cogv_check_launch();
  }
  if (a.kv_index || a.mask) {         // the flexible instantiation: gathered / sparse forms, arbitrary mask tensors
    if (d->dtype == COGV_F16) hipLaunchKernelGGL((attn_fwd_kernel<f16_t, true, -1>), grid, dim3(NT), sh, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, true, -1>), grid, dim3(NT), sh, st, a);
  } else if (d->dtype == COGV_F16) {
    if (drop) hipLaunchKernelGGL((attn_fwd_kernel<f16_t, false, 1>), grid, dim3(NT), sh, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<f16_t, false, 0>), grid, dim3(NT), sh, st, a);
  } else {
    if (drop) hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, false, 1>), grid, dim3(NT), sh, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, false, 0>), grid, dim3(NT), sh, st, a);
  }
  return cogv_check_launch();
}

extern "C" int cogv_attention_bwd(const cogv_attn_desc* d, void* stream) {
  AttnArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  if (!a.q || !a.k || !a.v || !a.o || !a.dout || !a.dq || !a.dk || !a.dv || !a.lse || !a.dvec) return COGV_ERR_ARG;
  if (d->kv_index && d->sparse_window <= 0) return COGV_ERR_UNSUPPORTED;      // the plain gathered form is inference only
  if ((rc = index_args(d, a))) return rc;
  if (a.sp_w > 0 && d->colsum_partial) return COGV_ERR_UNSUPPORTED;
  if (d->colsum_partial) {
    if (d->s_q != d->s_k || ((uintptr_t)d->colsum_partial & 15)) return COGV_ERR_ARG;
    a.colsum_ws = d->colsum_partial;
  }
  if (!aligned16(a.q) || !aligned16(a.k) || !aligned16(a.v) || !aligned16(a.o) || !aligned16(a.dout) ||
      !aligned16(a.dq) || !aligned16(a.dk) || !aligned16(a.dv)) return COGV_ERR_ARG;
  if ((a.q_rs | a.k_rs | a.v_rs | a.o_rs | a.do_rs | a.dq_rs | a.dk_rs | a.dv_rs) & 7) return COGV_ERR_ARG;
  if ((a.q_bs | a.k_bs | a.v_bs | a.o_bs | a.do_bs | a.dq_bs | a.dk_bs | a.dv_bs) & 7) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // sparse training form: dK / dV are slot-space buffers, one [s_k] plane per (batch, query block): dk_bs / dv_bs is
  // the plane stride and grid z runs over B * (s_q / sparse_window) planes
  const int planes = a.sp_w > 0 ? a.B * (a.s_q / a.sp_w) : a.B;
  dim3 gq(8u * (unsigned)((a.H * a.B + 7) / 8) * (unsigned)((a.s_q + 127) / 128));
  dim3 gk(8u * (unsigned)((a.H * planes + 7) / 8) * (unsigned)((a.s_k + 127) / 128));
  const int sh_q = ring_bytes(2 * TILE, true) + (a.kv_index ? ((a.s_k * 4 + 15) / 16) * 16 : 0), sh_k = ring_bytes(2 * TILE + 768, true);
  if (sh_q > 160 * 1024) return COGV_ERR_UNSUPPORTED;
  static int attr_q = 0;
  static bool attr = false;
  if (!attr) {
    set_smem(&attn_bwd_dkdv_kernel<f16_t, false, 0>, sh_k); set_smem(&attn_bwd_dkdv_kernel<bf16_t, false, 0>, sh_k);
    set_smem(&attn_bwd_dkdv_kernel<f16_t, false, 1>, sh_k); set_smem(&attn_bwd_dkdv_kernel<bf16_t, false, 1>, sh_k);
    set_smem(&attn_bwd_dkdv_kernel<f16_t, true, -1>, sh_k); set_smem(&attn_bwd_dkdv_kernel<bf16_t, true, -1>, sh_k);
    set_smem(&attn_bwd_dq_kernel<f16_t, false, 0>, ring_bytes(2 * TILE, true)); set_smem(&attn_bwd_dq_kernel<bf16_t, false, 0>, ring_bytes(2 * TILE, true));
    set_smem(&attn_bwd_dq_kernel<f16_t, false, 1>, ring_bytes(2 * TILE, true)); set_smem(&attn_bwd_dq_kernel<bf16_t, false, 1>, ring_bytes(2 * TILE, true));
    set_smem(&attn_bwd_dq_kernel<f16_t, false, 2>, ring_bytes(2 * TILE + 1024, true)); set_smem(&attn_bwd_dq_kernel<bf16_t, false, 2>, ring_bytes(2 * TILE + 1024, true));
    set_smem(&attn_bwd_dkdv_kernel<f16_t, false, 2>, ring_bytes(2 * TILE + 512 + 4 * COGV_DKDV_KW_PITCH, true) + DKDV_OWN_V); set_smem(&attn_bwd_dkdv_kernel<bf16_t, false, 2>, ring_bytes(2 * TILE + 512 + 4 * COGV_DKDV_KW_PITCH, true) + DKDV_OWN_V);
    attr = true;
  }
  // the flexible dQ instantiation (gathered / sparse keys: ring + index table; arbitrary mask tensors: ring only) shares ONE
  // attribute, which only ever grows (a smaller later request must not lower it under a launch that still needs more)
  if ((a.kv_index || a.mask) && sh_q > attr_q) {
    set_smem(&attn_bwd_dq_kernel<f16_t, true, -1>, sh_q); set_smem(&attn_bwd_dq_kernel<bf16_t, true, -1>, sh_q);
    attr_q = sh_q;
  }
  const bool drop = a.thr16 != 0u;
  if (d->keep_bits && drop && !a.kv_index) {      // the keep bits the forward call stored (same dropout_p / seed / stream)
    if ((uintptr_t)d->keep_bits & 3) return COGV_ERR_ARG;
    a.keepbits = reinterpret_cast<uint32_t*>(d->keep_bits);
    const int shq2 = ring_bytes(2 * TILE + 1024, true), shk2 = ring_bytes(2 * TILE + 512 + 4 * COGV_DKDV_KW_PITCH, true) + DKDV_OWN_V;
    if (d->dtype == COGV_F16) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<f16_t, false, 2>), gq, dim3(NT), shq2, st, a);
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<f16_t, false, 2>), gk, dim3(NT), shk2, st, a);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t, false, 2>), gq, dim3(NT), shq2, st, a);
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<bf16_t, false, 2>), gk, dim3(NT), shk2, st, a);
    }
    return cogv_check_launch();
  }
#define ATTN_BWD_LAUNCH(T_, IDX_, DROP_)                                                             \
  do {                                                                                               \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T_, IDX_, DROP_>), gq, dim3(NT), sh_q, st, a);            \
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T_, IDX_, DROP_>), gk, dim3(NT), sh_k, st, a);          \
  } while (0)
  if (a.kv_index || a.mask) {   // sparse training form / arbitrary mask tensor: the instantiation with the gather, the slot attributes, the mask
    if (d->dtype == COGV_F16) ATTN_BWD_LAUNCH(f16_t, true, -1); else ATTN_BWD_LAUNCH(bf16_t, true, -1);
  } else if (d->dtype == COGV_F16) {
    if (drop) ATTN_BWD_LAUNCH(f16_t, false, 1); else ATTN_BWD_LAUNCH(f16_t, false, 0);
  } else {
    if (drop) ATTN_BWD_LAUNCH(bf16_t, false, 1); else ATTN_BWD_LAUNCH(bf16_t, false, 0);
  }
#undef ATTN_BWD_LAUNCH
  return cogv_check_launch();
}

extern "C" size_t cogv_attention_keep_bits_bytes(int B, int H, int s_q, int s_k) {
  if (B <= 0 || H <= 0 || s_q <= 0 || s_k <= 0) return 0;
  return (size_t)B * H * ((size_t)(s_k + 63) / 64) * 2 * (size_t)s_q * sizeof(uint32_t);
}

extern "C" size_t cogv_attention_decode_workspace_bytes(int B, int H, int capacity) {
  if (B <= 0 || H <= 0 || capacity <= 0) return 0;
  const size_t nsplit = (size_t)(capacity + 127) / 128;
  return (size_t)B * H * nsplit * 66 * sizeof(float);
}

extern "C" int cogv_attention_decode(const cogv_attn_decode_desc* d, void* stream) {
  if (!d || (d->dtype != COGV_F16 && d->dtype != COGV_BF16)) return COGV_ERR_UNSUPPORTED;
  if (d->B <= 0 || d->H <= 0 || d->capacity <= 0 || d->capacity > 4096 || d->head_dim != HD) return COGV_ERR_ARG;
  if (!d->qkv || !d->cache || (!d->out && !d->skip_combine) || !d->pos || !d->workspace) return COGV_ERR_ARG;
  if (!aligned16(d->qkv) || !aligned16(d->cache) || ((d->qkv_bs | d->cache_bs | d->cache_rs) & 7)) return COGV_ERR_ARG;
  if (d->workspace_bytes < cogv_attention_decode_workspace_bytes(d->B, d->H, d->capacity) || ((uintptr_t)d->workspace & 15)) return COGV_ERR_ARG;
  DecodeArgs a;
  a.qkv = d->qkv; a.cache = d->cache; a.out = d->out; a.pos = d->pos;
  a.tickets = nullptr;
  a.ws = reinterpret_cast<float*>(d->workspace);
  a.qkv_bs = d->qkv_bs; a.cache_bs = d->cache_bs; a.out_bs = d->out_bs; a.cache_rs = d->cache_rs;
  a.B = d->B; a.H = d->H; a.cap = d->capacity; a.nsplit = (d->capacity + 127) / 128;
  a.scale_l2e = d->scale * 1.4426950408889634f;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(a.nsplit, a.H, a.B);
  // skip_combine: the consumer (cogv_gemv_attn: the attention-output projection of a decode step) recombines the partials itself
  if (d->dtype == COGV_F16) {
    hipLaunchKernelGGL((attn_decode_kernel<f16_t>), grid, dim3(256), 0, st, a);
    if (!d->skip_combine) hipLaunchKernelGGL((attn_decode_combine_kernel<f16_t>), dim3(a.H, a.B), dim3(64), 0, st, a);
  } else {
    hipLaunchKernelGGL((attn_decode_kernel<bf16_t>), grid, dim3(256), 0, st, a);
    if (!d->skip_combine) hipLaunchKernelGGL((attn_decode_combine_kernel<bf16_t>), dim3(a.H, a.B), dim3(64), 0, st, a);
  }
  return cogv_check_launch();
}

extern "C" int cogv_sparse_slot_reduce(int dtype, const void* dk_slots, const void* dv_slots, const int* pivot_inv,
                                       void* dk, void* dv, long long dk_bs, int dk_rs, long long dv_bs, int dv_rs,
                                       int B, int s, int H, int window, int times, int n_pivots, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (!dk_slots || !dv_slots || !pivot_inv || !dk || !dv) return COGV_ERR_ARG;
  if (B <= 0 || s <= 0 || H <= 0 || window <= 0 || times <= 0 || n_pivots < 0 || (s % window)) return COGV_ERR_ARG;
  if (!aligned16(dk_slots) || !aligned16(dv_slots) || !aligned16(dk) || !aligned16(dv)) return COGV_ERR_ARG;
  if ((dk_bs | dv_bs | dk_rs | dv_rs) & 7) return COGV_ERR_ARG;
  const int C8 = H * HD / 8;
  const long long total = (long long)B * s * C8;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COGV_F16)
    hipLaunchKernelGGL((sparse_slot_reduce_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, (const f16_t*)dk_slots,
                       (const f16_t*)dv_slots, pivot_inv, (f16_t*)dk, (f16_t*)dv, dk_bs, dk_rs, dv_bs, dv_rs, B, s, C8,
                       window, times, n_pivots);
  else
    hipLaunchKernelGGL((sparse_slot_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)dk_slots,
                       (const bf16_t*)dv_slots, pivot_inv, (bf16_t*)dk, (bf16_t*)dv, dk_bs, dk_rs, dv_bs, dv_rs, B, s, C8,
                       window, times, n_pivots);
  return cogv_check_launch();
}

#if defined(COGV_ATTN_TS)
extern "C" int cogv_debug_attn_ts(unsigned long long* host_out, int reset) {
  static unsigned long long h[1024 * 8];
  if (host_out) {
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_attn_ts), sizeof(h)) != hipSuccess) return COGV_ERR_LAUNCH;
    for (int k = 0; k < 8; ++k) { host_out[k] = 0; for (int i = 0; i < 1024; ++i) host_out[k] += h[i * 8 + k]; }
  }
  if (reset) {
    for (int i = 0; i < 1024 * 8; ++i) h[i] = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_ts), h, sizeof(h)) != hipSuccess) return COGV_ERR_LAUNCH;
  }
  return COGV_OK;
}
#endif
