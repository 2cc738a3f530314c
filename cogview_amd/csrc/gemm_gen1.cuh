// Generation 1 (gemm_kernel): 128x128x64 tiles, register-staged operands with register transposes.  The fallback for K % 64 != 0 and other unaligned shapes.
// Part of the GEMM family of csrc/gemm.hip (included there, in this order: common, gen1, lds, gen2, gen3, gen4, gemv_gen1);
// not a stand-alone header.
#pragma once

namespace {

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

}  // namespace
