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
// The A operands that need the contraction index contiguous (V^T, K^T, Q^T, dO^T) are produced by a
// register transpose while staging HBM -> LDS (4 rows x 8 columns per thread, ds_write_b64 granules).
//
// q/k/v are addressed as base + b*batch_stride + row*row_stride + head*64, i.e. straight out of the
// [b, s, 3*h/p] QKV GEMM output -- the reference's _transpose_for_scores permute copies
// (mpu/sparse_transformer.py:112-120,159) are folded into the addressing.
#include "common.cuh"
#include "cogview_hip.h"

namespace {

constexpr int HD = 64;          // head dim
constexpr int NT = 256;         // threads per block (4 waves)
constexpr float MASKED = -10000.0f;

struct AttnArgs {
  const void* q; const void* k; const void* v; void* o;        // forward
  const void* dout; void* dq; void* dk; void* dv;              // backward
  float* lse; float* dvec;                                     // [b][H][s_q]
  long long q_bs, k_bs, v_bs, o_bs, do_bs, dq_bs, dk_bs, dv_bs; // batch strides (elements)
  int q_rs, k_rs, v_rs, o_rs, do_rs, dq_rs, dk_rs, dv_rs;       // row strides (elements)
  int B, H, s_q, s_k, sep_k;   // sep_k: keys [0, sep_k) visible to every query
  float scale;
  uint32_t thr16; float keep_scale; uint32_t rng_key;
};

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }

// ---- natural tile: ROWS x 64 halves, 128 B per row, 16-B chunk swizzle (read with ds_read_b128)
template <typename T, int ROWS>
struct NatStage {
  static constexpr int PER = ROWS * 8 / NT;   // 16-B chunks per thread
  u32x4 r[PER > 0 ? PER : 1];
  __device__ __forceinline__ void load(const T* base, long long rs, int row0, int nrows) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int idx = threadIdx.x + p * NT;
      const int row = idx >> 3, chunk = idx & 7;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row0 + row < nrows) v = *reinterpret_cast<const u32x4*>(base + (long long)(row0 + row) * rs + chunk * 8);
      r[p] = v;
    }
  }
  __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int idx = threadIdx.x + p * NT;
      const int row = idx >> 3, chunk = idx & 7;
      *reinterpret_cast<u32x4*>(lds + row * 128 + ((chunk ^ swz(row)) << 4)) = r[p];
    }
  }
};
template <typename T>
__device__ __forceinline__ typename HT<T>::v8 nat_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const typename HT<T>::v8*>(lds + row * 128 + ((chunk ^ swz(row)) << 4));
}

// ---- transposed tile: source ROWS x 64 (row-major) -> LDS [64 cols][ROWS] halves, row = ROWS*2 bytes,
//      8-byte granules (4 source rows) XOR-swizzled by ((col >> 1) & (G-1)), G = granules per LDS row.
//      Staged by a 128-thread half of the block (ROWS == 64: one item each; ROWS == 32: threads 0..63 of it).
template <typename T, int ROWS>
struct TrStage {
  u32x4 r[4];
  // tid: 0..127 index inside the staging half
  __device__ __forceinline__ void load(const T* base, long long rs, int row0, int nrows, int tid) {
    const int c = tid & 7, rg = tid >> 3;        // 8-column chunk, 4-row group
    if (rg * 4 < ROWS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        const int row = row0 + rg * 4 + i;
        if (row < nrows) v = *reinterpret_cast<const u32x4*>(base + (long long)row * rs + c * 8);
        r[i] = v;
      }
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
    constexpr int G = ROWS / 4;                  // granules per LDS row
    const int c = tid & 7, rg = tid >> 3;
    if (rg * 4 < ROWS) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        u32x2 lo, hi;
        lo[0] = (r[0][w] & 0xffffu) | (r[1][w] << 16);
        lo[1] = (r[2][w] & 0xffffu) | (r[3][w] << 16);
        hi[0] = (r[0][w] >> 16) | (r[1][w] & 0xffff0000u);
        hi[1] = (r[2][w] >> 16) | (r[3][w] & 0xffff0000u);
        const int ce = c * 8 + 2 * w, co = ce + 1;
        *reinterpret_cast<u32x2*>(lds + ce * (ROWS * 2) + ((rg ^ ((ce >> 1) & (G - 1))) << 3)) = lo;
        *reinterpret_cast<u32x2*>(lds + co * (ROWS * 2) + ((rg ^ ((co >> 1) & (G - 1))) << 3)) = hi;
      }
    }
  }
};
// A-operand fragment from a transposed tile: row = output index (d), contraction slots of k-step `t16`
// (16 source rows starting at src0): elements e -> source row src0 + 8(e>>2) + 4g + (e&3)
template <typename T, int ROWS>
__device__ __forceinline__ typename HT<T>::v8 tr_frag(const char* lds, int col, int src0, int g) {
  constexpr int G = ROWS / 4;
  const int g0 = (src0 >> 2) + g;       // granule holding rows src0+4g .. +3
  const int g1 = g0 + 2;                // rows src0+8+4g .. +3
  const int sw = (col >> 1) & (G - 1);
  const u32x2 a = *reinterpret_cast<const u32x2*>(lds + col * (ROWS * 2) + ((g0 ^ sw) << 3));
  const u32x2 b = *reinterpret_cast<const u32x2*>(lds + col * (ROWS * 2) + ((g1 ^ sw) << 3));
  u32x4 w; w[0] = a[0]; w[1] = a[1]; w[2] = b[0]; w[3] = b[1];
  typename HT<T>::v8 out; __builtin_memcpy(&out, &w, 16);
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

// dropout bits for (attention row `arow` = (b*H+head)*s_q + q, keys key0..key0+3, key0 % 4 == 0)
__device__ __forceinline__ u32x2 attn_bits(uint32_t key, long long arow, int ngrp, int key0) {
  return Philox::gen64_k(key, (uint64_t)(arow * ngrp + (key0 >> 2)));
}
__device__ __forceinline__ uint32_t bits_of(const u32x2& r, int i) { return (r[i >> 1] >> (16 * (i & 1))) & 0xffffu; }

// =====================================================================================================
// forward: grid (ceil(s_q/128), H, B); wave w owns queries q0 + 32w .. +31
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(NT) void attn_fwd_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (K 8 KiB + V^T 8 KiB)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int q0 = blockIdx.x * 128, q0w = q0 + wave * 32;
  const int off = p.s_k - p.s_q;
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.q_bs + head * HD;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.k_bs + head * HD;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.v_bs + head * HD;
  const int myq = q0w + fr;
  const bool wave_active = q0w < p.s_q;

  typename HT<T>::v8 qf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) qf[t] = load_frag_global<T>(Q + (long long)myq * p.q_rs + 16 * t + 8 * fg, myq < p.s_q);

  // key range: the block needs keys up to the last row's horizon (or the visible prefix)
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
  const float sl2 = p.scale * 1.4426950408889634f;   // scores are kept in the log2 domain
  const float masked_l2 = MASKED * 1.4426950408889634f;
  const long long arow = ((long long)b * p.H + head) * p.s_q + myq;
  const int ngrp = (p.s_k + 3) >> 2;

  NatStage<T, 64> ks; TrStage<T, 64> vs;
  const bool vstager = threadIdx.x < 128;
  auto g_load = [&](int kb) {
    ks.load(K, p.k_rs, kb * 64, p.s_k);
    if (vstager) vs.load(V, p.v_rs, kb * 64, p.s_k, threadIdx.x);
  };
  auto l_store = [&](int s) {
    ks.store(smem + s * 16384);
    if (vstager) vs.store(smem + s * 16384 + 8192, threadIdx.x);
  };
  if (nkb > 0) { g_load(0); l_store(0); }
  __syncthreads();
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nkb) g_load(kb + 1);
    if (kb * 64 < kend_w) {
      const char* lk = smem + cur * 16384; const char* lv = lk + 8192;
      f32x16 sacc[2];
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[sb][e] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          sacc[sb] = HT<T>::mfma32(nat_frag<T>(lk, sb * 32 + fr, 2 * t + fg), qf[t], sacc[sb]);
      }
      // mask + scale (log2 domain), block max
      float mb = -INFINITY;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kb * 64 + sb * 32 + (e & 3) + 8 * (e >> 2) + 4 * fg;
          float s = sacc[sb][e] * sl2;
          if (!visible(myq, key, off, p.sep_k)) s = masked_l2;
          if (key >= p.s_k) s = -INFINITY;
          sacc[sb][e] = s;
          mb = fmaxf(mb, s);
        }
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float m_new = fmaxf(m_run, mb);
      const float alpha = exp2f(m_run - m_new);
      m_run = m_new;
      float ls = 0.f;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float pv = exp2f(sacc[sb][e] - m_new); sacc[sb][e] = pv; ls += pv; }
      l_run = l_run * alpha + ls;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
      if (p.thr16) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int key0 = kb * 64 + sb * 32 + 8 * gq + 4 * fg;
            const u32x2 r = attn_bits(p.rng_key, arow, ngrp, key0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              sacc[sb][4 * gq + i] = (bits_of(r, i) >= p.thr16) ? sacc[sb][4 * gq + i] * p.keep_scale : 0.f;
          }
      }
      // O^T[d][query] += V^T[d][key] . P^T[key][query]
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float pe[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pe[e] = sacc[sb][8 * t + e];
          const typename HT<T>::v8 pb = cvt8<T>(pe);
#pragma unroll
          for (int d = 0; d < 2; ++d)
            oacc[d] = HT<T>::mfma32(tr_frag<T, 64>(lv, d * 32 + fr, sb * 32 + 16 * t, fg), pb, oacc[d]);
        }
    }
    if (kb + 1 < nkb) l_store(cur ^ 1);
    __syncthreads();
  }
  if (wave_active && myq < p.s_q) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (fg == 0 && p.lse) p.lse[((long long)b * p.H + head) * p.s_q + myq] = (m_run + log2f(l_tot)) * 0.6931471805599453f;
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
// D[b][h][q] = sum_d dO[q][d] * O[q][d]     (8 lanes per row)
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void attn_dvec_kernel(const AttnArgs p) {
  const long long nrow = (long long)p.B * p.H * p.s_q;
  const long long rid = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  float s = 0.f;
  if (rid < nrow) {
    const int q = (int)(rid % p.s_q); const long long bh = rid / p.s_q;
    const int head = (int)(bh % p.H); const int b = (int)(bh / p.H);
    float a[8], c[8];
    unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.dout) + b * p.do_bs + (long long)q * p.do_rs + head * HD + sub * 8), a);
    unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.o) + b * p.o_bs + (long long)q * p.o_rs + head * HD + sub * 8), c);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] * c[i];
  }
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  if (rid < nrow && sub == 0) p.dvec[rid] = s;
}

// =====================================================================================================
// dQ: grid (ceil(s_q/128), H, B); lane = query.   dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q]
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(NT) void attn_bwd_dq_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (K 8K + V 8K + K^T 8K)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int q0 = blockIdx.x * 128, q0w = q0 + wave * 32;
  const int off = p.s_k - p.s_q;
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
  const float lse2 = qvalid ? p.lse[arow] * 1.4426950408889634f : 0.f;
  const float dv = qvalid ? p.dvec[arow] : 0.f;
  const int ngrp = (p.s_k + 3) >> 2;
  const float sl2 = p.scale * 1.4426950408889634f;
  const float masked_l2 = MASKED * 1.4426950408889634f;

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

  NatStage<T, 64> ks, vs; TrStage<T, 64> kts;
  const bool tstager = threadIdx.x < 128;
  auto g_load = [&](int kb) {
    ks.load(K, p.k_rs, kb * 64, p.s_k);
    vs.load(V, p.v_rs, kb * 64, p.s_k);
    if (tstager) kts.load(K, p.k_rs, kb * 64, p.s_k, threadIdx.x);
  };
  auto l_store = [&](int s) {
    ks.store(smem + s * 24576);
    vs.store(smem + s * 24576 + 8192);
    if (tstager) kts.store(smem + s * 24576 + 16384, threadIdx.x);
  };
  if (nkb > 0) { g_load(0); l_store(0); }
  __syncthreads();
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nkb) g_load(kb + 1);
    if (kb * 64 < kend_w) {
      const char* lk = smem + cur * 24576; const char* lv = lk + 8192; const char* lkt = lk + 16384;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        f32x16 sacc, pacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sacc = HT<T>::mfma32(nat_frag<T>(lk, sb * 32 + fr, 2 * t + fg), qf[t], sacc);     // S^T
          pacc = HT<T>::mfma32(nat_frag<T>(lv, sb * 32 + fr, 2 * t + fg), dof[t], pacc);    // dP^T = V dO^T
        }
        float ds[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int key0 = kb * 64 + sb * 32 + 8 * gq + 4 * fg;
          u32x2 r = {0u, 0u};
          if (p.thr16) r = attn_bits(p.rng_key, arow, ngrp, key0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 4 * gq + i, key = key0 + i;
            float s = sacc[e] * sl2;
            if (!visible(myq, key, off, p.sep_k)) s = masked_l2;
            float pr = exp2f(s - lse2);
            if (key >= p.s_k) pr = 0.f;
            float dp = pacc[e];
            if (p.thr16) dp = (bits_of(r, i) >= p.thr16) ? dp * p.keep_scale : 0.f;
            ds[e] = pr * (dp - dv);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const typename HT<T>::v8 dsb = cvt8<T>(ds + 8 * t);
#pragma unroll
          for (int d = 0; d < 2; ++d)
            dqacc[d] = HT<T>::mfma32(tr_frag<T, 64>(lkt, d * 32 + fr, sb * 32 + 16 * t, fg), dsb, dqacc[d]);
        }
      }
    }
    if (kb + 1 < nkb) l_store(cur ^ 1);
    __syncthreads();
  }
  if (qvalid) {
    T* DQ = reinterpret_cast<T*>(p.dq) + b * p.dq_bs + (long long)myq * p.dq_rs + head * HD;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        u32x2 w;
        w[0] = pack2<T>(dqacc[d][4 * gq] * p.scale, dqacc[d][4 * gq + 1] * p.scale);
        w[1] = pack2<T>(dqacc[d][4 * gq + 2] * p.scale, dqacc[d][4 * gq + 3] * p.scale);
        *reinterpret_cast<u32x2*>(DQ + d * 32 + 8 * gq + 4 * fg) = w;
      }
  }
}

// =====================================================================================================
// dK/dV: grid (ceil(s_k/128), H, B); lane = key.
//   dV^T[d][key] = sum_q dO^T[d][q] Pd[q][key]        dK^T[d][key] = scale * sum_q Q^T[d][q] dS[q][key]
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(NT) void attn_bwd_dkdv_kernel(const AttnArgs p) {
  // per stage: Q 8K | dO 8K | Q^T 8K | dO^T 8K | lse 256 B | dvec 256 B   (64 queries per stage)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 4 * 8192 + 512;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int k0 = blockIdx.x * 128, k0w = k0 + wave * 32;
  const int off = p.s_k - p.s_q;
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.q_bs + head * HD;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.k_bs + head * HD;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.v_bs + head * HD;
  const T* DO = reinterpret_cast<const T*>(p.dout) + b * p.do_bs + head * HD;
  const float* LSE = p.lse + ((long long)b * p.H + head) * p.s_q;
  const float* DV = p.dvec + ((long long)b * p.H + head) * p.s_q;
  const int mykey = k0w + fr;
  const bool kvalid = mykey < p.s_k;
  const bool wave_active = k0w < p.s_k;

  typename HT<T>::v8 kf[4], vf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    kf[t] = load_frag_global<T>(K + (long long)mykey * p.k_rs + 16 * t + 8 * fg, kvalid);
    vf[t] = load_frag_global<T>(V + (long long)mykey * p.v_rs + 16 * t + 8 * fg, kvalid);
  }
  // first query that can see any key of this block / wave (keys < sep_k are seen by every query)
  const int qbeg_blk = (k0 < p.sep_k) ? 0 : max(0, k0 - off);
  const int qbeg_w = (k0w < p.sep_k) ? 0 : max(0, k0w - off);
  const int qb0 = qbeg_blk >> 6;
  const int nqb = (p.s_q + 63) >> 6;
  const float sl2 = p.scale * 1.4426950408889634f;
  const float masked_l2 = MASKED * 1.4426950408889634f;
  const int ngrp = (p.s_k + 3) >> 2;
  const long long arow0 = ((long long)b * p.H + head) * p.s_q;

  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[d][e] = 0.f; dvacc[d][e] = 0.f; }

  NatStage<T, 64> qs, dos; TrStage<T, 64> ts;   // waves 0,1 transpose Q ; waves 2,3 transpose dO
  float st_l = 0.f, st_d = 0.f;
  auto g_load = [&](int qb) {
    qs.load(Q, p.q_rs, qb * 64, p.s_q);
    dos.load(DO, p.do_rs, qb * 64, p.s_q);
    if (threadIdx.x < 128) ts.load(Q, p.q_rs, qb * 64, p.s_q, threadIdx.x);
    else ts.load(DO, p.do_rs, qb * 64, p.s_q, threadIdx.x - 128);
    if (threadIdx.x < 64) { const int q = qb * 64 + threadIdx.x; st_l = q < p.s_q ? LSE[q] * 1.4426950408889634f : 0.f; }
    else if (threadIdx.x < 128) { const int q = qb * 64 + threadIdx.x - 64; st_d = q < p.s_q ? DV[q] : 0.f; }
  };
  auto l_store = [&](int s) {
    char* base = smem + s * STAGE;
    qs.store(base); dos.store(base + 8192);
    if (threadIdx.x < 128) ts.store(base + 16384, threadIdx.x); else ts.store(base + 24576, threadIdx.x - 128);
    float* stat = reinterpret_cast<float*>(base + 32768);
    if (threadIdx.x < 64) stat[threadIdx.x] = st_l; else if (threadIdx.x < 128) stat[threadIdx.x] = st_d;
  };
  if (qb0 < nqb) { g_load(qb0); l_store(0); }
  __syncthreads();
  for (int qb = qb0; qb < nqb; ++qb) {
    const int cur = (qb - qb0) & 1;
    if (qb + 1 < nqb) g_load(qb + 1);
    if (wave_active && qb * 64 + 63 >= qbeg_w) {
      const char* lq = smem + cur * STAGE; const char* ldo = lq + 8192;
      const char* lqt = lq + 16384; const char* ldot = lq + 24576;
      const float* stat = reinterpret_cast<const float*>(lq + 32768);
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        f32x16 sacc, pacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; pacc[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sacc = HT<T>::mfma32(nat_frag<T>(lq, sb * 32 + fr, 2 * t + fg), kf[t], sacc);     // S = Q K^T
          pacc = HT<T>::mfma32(nat_frag<T>(ldo, sb * 32 + fr, 2 * t + fg), vf[t], pacc);    // dPd = dO V^T
        }
        float pd[16], ds[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int ql = sb * 32 + 8 * gq + 4 * fg;                // local query of element i = 0
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(stat + ql);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(stat + 64 + ql);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 4 * gq + i, q = qb * 64 + ql + i;
            float s = sacc[e] * sl2;
            if (!visible(q, mykey, off, p.sep_k)) s = masked_l2;
            float pr = exp2f(s - l4[i]);
            if (q >= p.s_q || !kvalid) pr = 0.f;
            float keep = 1.f;
            if (p.thr16) {
              const u32x2 r = attn_bits(p.rng_key, arow0 + q, ngrp, mykey & ~3);
              keep = (bits_of(r, mykey & 3) >= p.thr16) ? p.keep_scale : 0.f;
            }
            pd[e] = pr * keep;
            ds[e] = pr * (pacc[e] * keep - d4[i]);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const typename HT<T>::v8 pb = cvt8<T>(pd + 8 * t);
          const typename HT<T>::v8 dsb = cvt8<T>(ds + 8 * t);
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            dvacc[d] = HT<T>::mfma32(tr_frag<T, 64>(ldot, d * 32 + fr, sb * 32 + 16 * t, fg), pb, dvacc[d]);
            dkacc[d] = HT<T>::mfma32(tr_frag<T, 64>(lqt, d * 32 + fr, sb * 32 + 16 * t, fg), dsb, dkacc[d]);
          }
        }
      }
    }
    if (qb + 1 < nqb) l_store(cur ^ 1);
    __syncthreads();
  }
  if (kvalid) {
    T* DK = reinterpret_cast<T*>(p.dk) + b * p.dk_bs + (long long)mykey * p.dk_rs + head * HD;
    T* DVp = reinterpret_cast<T*>(p.dv) + b * p.dv_bs + (long long)mykey * p.dv_rs + head * HD;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        u32x2 w;
        w[0] = pack2<T>(dkacc[d][4 * gq] * p.scale, dkacc[d][4 * gq + 1] * p.scale);
        w[1] = pack2<T>(dkacc[d][4 * gq + 2] * p.scale, dkacc[d][4 * gq + 3] * p.scale);
        *reinterpret_cast<u32x2*>(DK + d * 32 + 8 * gq + 4 * fg) = w;
        w[0] = pack2<T>(dvacc[d][4 * gq], dvacc[d][4 * gq + 1]);
        w[1] = pack2<T>(dvacc[d][4 * gq + 2], dvacc[d][4 * gq + 3]);
        *reinterpret_cast<u32x2*>(DVp + d * 32 + 8 * gq + 4 * fg) = w;
      }
  }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int fill_args(const cogv_attn_desc* d, AttnArgs& a) {
  if (!d) return COGV_ERR_ARG;
  if (d->dtype != COGV_F16 && d->dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (d->head_dim != HD) return COGV_ERR_UNSUPPORTED;
  if (d->B <= 0 || d->H <= 0 || d->s_q <= 0 || d->s_k <= 0 || d->s_k < d->s_q) return COGV_ERR_ARG;
  if (!(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return COGV_ERR_ARG;
  a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o; a.dout = d->dout; a.dq = d->dq; a.dk = d->dk; a.dv = d->dv;
  a.lse = d->lse; a.dvec = d->dvec;
  a.q_bs = d->q_bs; a.k_bs = d->k_bs; a.v_bs = d->v_bs; a.o_bs = d->o_bs; a.do_bs = d->do_bs;
  a.dq_bs = d->dq_bs; a.dk_bs = d->dk_bs; a.dv_bs = d->dv_bs;
  a.q_rs = d->q_rs; a.k_rs = d->k_rs; a.v_rs = d->v_rs; a.o_rs = d->o_rs; a.do_rs = d->do_rs;
  a.dq_rs = d->dq_rs; a.dk_rs = d->dk_rs; a.dv_rs = d->dv_rs;
  a.B = d->B; a.H = d->H; a.s_q = d->s_q; a.s_k = d->s_k;
  int sep = d->sep; if (sep < 0) sep = 0;
  a.sep_k = sep > 0 ? sep + (d->s_k - d->s_q) : 0;
  a.scale = d->scale;
  a.thr16 = (uint32_t)(d->dropout_p * 65536.0f + 0.5f);
  a.keep_scale = 65536.0f / (65536.0f - (float)a.thr16);
  a.rng_key = rng_key(d->seed, d->stream_id);
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
  dim3 grid((a.s_q + 127) / 128, a.H, a.B);
  const int sh = 2 * 16384;
  if (d->dtype == COGV_F16) hipLaunchKernelGGL((attn_fwd_kernel<f16_t>), grid, dim3(NT), sh, st, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), grid, dim3(NT), sh, st, a);
  return cogv_check_launch();
}

extern "C" int cogv_attention_bwd(const cogv_attn_desc* d, void* stream) {
  AttnArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  if (!a.q || !a.k || !a.v || !a.o || !a.dout || !a.dq || !a.dk || !a.dv || !a.lse || !a.dvec) return COGV_ERR_ARG;
  if (!aligned16(a.q) || !aligned16(a.k) || !aligned16(a.v) || !aligned16(a.o) || !aligned16(a.dout) ||
      !aligned16(a.dq) || !aligned16(a.dk) || !aligned16(a.dv)) return COGV_ERR_ARG;
  if ((a.q_rs | a.k_rs | a.v_rs | a.o_rs | a.do_rs | a.dq_rs | a.dk_rs | a.dv_rs) & 7) return COGV_ERR_ARG;
  if ((a.q_bs | a.k_bs | a.v_bs | a.o_bs | a.do_bs | a.dq_bs | a.dk_bs | a.dv_bs) & 7) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long nrow = (long long)a.B * a.H * a.s_q;
  const int gD = (int)((nrow * 8 + 255) / 256);
  dim3 gq((a.s_q + 127) / 128, a.H, a.B), gk((a.s_k + 127) / 128, a.H, a.B);
  const int sh_q = 2 * 24576, sh_k = 2 * (4 * 8192 + 512);
  static bool attr = false;
  if (!attr) {
    set_smem(&attn_bwd_dkdv_kernel<f16_t>, sh_k); set_smem(&attn_bwd_dkdv_kernel<bf16_t>, sh_k);
    set_smem(&attn_bwd_dq_kernel<f16_t>, sh_q); set_smem(&attn_bwd_dq_kernel<bf16_t>, sh_q);
    attr = true;
  }
  if (d->dtype == COGV_F16) {
    hipLaunchKernelGGL((attn_dvec_kernel<f16_t>), dim3(gD), dim3(256), 0, st, a);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<f16_t>), gq, dim3(NT), sh_q, st, a);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<f16_t>), gk, dim3(NT), sh_k, st, a);
  } else {
    hipLaunchKernelGGL((attn_dvec_kernel<bf16_t>), dim3(gD), dim3(256), 0, st, a);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t>), gq, dim3(NT), sh_q, st, a);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<bf16_t>), gk, dim3(NT), sh_k, st, a);
  }
  return cogv_check_launch();
}
