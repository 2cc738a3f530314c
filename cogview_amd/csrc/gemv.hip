// Skinny-M (M <= 8) matrix-vector kernels of the decode step, second generation (round 4).
//
//   C[M,N] = epilogue( X[M,K] * B[N,K]^T ),   X = A | LayerNorm prologue of a stream row | combine of the decode attention
//
// A decode step of the 4B model reads its 7.9 GB of weights once per generated token: every launch is a pure HBM stream of
// B, 13-52 MB long, i.e. 2-7 us at the memory's speed -- the same order as a launch's fixed costs.  The first generation
// (gemm.hip: gemv_kernel, gemv_ln_kernel, gemv_attn_kernel) gave a workgroup 8 columns and dealt the 512-element chunks of K
// to its four waves round robin: at K = 2560 wave 0 fetched two chunks ONE AFTER THE OTHER (two exposed HBM latencies), at
// K = 10240 every wave five; a wave never had more than 8 KB in flight, and the LayerNorm prologue's own small loads queued
// BEHIND the first weight chunk (the vector-memory counter retires in order).  Measured in the captured 4B step: 2.6 TB/s
// inside the kernels (profiles/r02_decode_kernel_stats_head.csv).
//
// Here a WAVE owns J whole columns (all of K), J * K / 512 <= 20 (40 at K = 10240) independent 16-byte loads per lane,
// all requested at the top of the kernel -- 20 KB in flight per wave, every byte of B requested once, immediately:
//     [the prologue's own inputs]  [ALL weight loads]  [prologue -> x in LDS]  barrier  [dot products as the chunks land]
// The small loads go first so that the prologue starts as soon as THEY arrive and runs while the weights stream in.
// The per-lane accumulation runs over the chunks in ascending order (fp32 fmaf chain), then one DPP wave sum per
// (row, column): no cross-wave reduction.  Lane 0 of the first NW * J / 8 waves runs the shared fused epilogue on 8 columns.
// That is the form for ONE row (FormV: fp32 fmaf chains on the vector ALU).  With 2 .. 8 rows its accumulators, unpacked x rows
// and weight slots no longer fit next to each other: the kernels fell to one or two waves per SIMD, a second round of
// workgroups, and in places spilled INSIDE the request phase (profiles/r04_decode_batch_*: 2.77 ms per token at one row, 3.48 /
// 4.88 / 9.29 at 2 / 4 / 8).  Rows 2 .. 8 therefore take FormM: the same weight stream, but the products go to the matrix
// core -- v_mfma_f32_16x16x32 with the WEIGHTS as the A operand (a lane's 16-byte load IS its fragment: column n0 + lane % 16,
// contraction slots 8 (lane / 16) ..), the x rows as B (from LDS, rows past M are don't-care columns of the result) and a
// 16 x 16 accumulator of 4 registers per lane whatever the row count.  A workgroup owns 16 columns; its waves split the 32-deep
// contraction steps (in pairs: one 128-byte line of a weight row) round robin and their partial tiles meet in LDS in wave order.
// The three kernels use the same association for the same K, so the combine-prologue form and the two-launch form of the
// attention-output projection still agree bit for bit (tests/test_kernels_gpu.py).
//
// Compiled as two translation units (-DCOGV_GEMV_TU=0: bf16, 1: fp16; build.py); without the macro this file is empty.
#include "gemm_shared.cuh"

#include <cstdlib>
#include <type_traits>

#ifdef COGV_GEMV_TU

namespace {

// a zero the compiler cannot see through, in a vector register: added to a wave-uniform address it keeps the load on the
// vector memory path (see gv2_bias)
__device__ __forceinline__ int gv2_vzero() {
  int zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
  return zero;
}

// bias of the 8 columns n .. n + 7 a finishing lane will need, requested ahead of its use.  Unconditional: without a bias (or
// in a lane that finishes nothing) a valid stand-in address is read and the value ignored.
// Through the VECTOR memory path (an offset the compiler cannot see through): as a wave-uniform address this would become a
// scalar load, and the scalar counter has to reach zero -- for the kernel arguments -- before the first weight load can be
// issued: the weight stream would start one memory latency late.
template <typename T>
__device__ __forceinline__ u32x4 gv2_bias(const GemmArgs& p, int n) {
  const T* src = (p.flags & COGV_EPI_BIAS) ? reinterpret_cast<const T*>(p.bias) : reinterpret_cast<const T*>(p.B);
  return gload16(src + (n < p.N ? n : p.N - 8) + gv2_vzero());
}

// =====================================================================================================================
// FormV: one row (MT = 1; the row-count template parameter is kept general).  A wave owns J whole columns.
template <typename T, int J, int KCMAX, bool GUARD_, int MT, int NW_>
struct FormV {
  static constexpr int NW = NW_, COLS = NW_ * J, XPAD = 0, NS = KCMAX * J, XCHUNKS = KCMAX;
  static constexpr bool GUARD = GUARD_;
  struct Regs { u32x4 w[NS]; };
  struct Shared { float outp[MT][NW_ * J]; };
  // slot s of a wave = (chunk c = s / J, column j = s % J): chunk-major, the order the dot products consume them.  Columns
  // past N (a wave at the ragged end of the matrix) re-read the last row; their results are never stored.
  template <int S0, int S1>
  static __device__ __forceinline__ void issue(Regs& r, const GemmArgs& p, int n0, int K, int wave, int lane) {
    const T* B = reinterpret_cast<const T*>(p.B);
    const int nw = n0 + wave * J, kc = K >> 9;
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      const int c = s / J, j = s % J;
      const int n = nw + j < p.N ? nw + j : p.N - 1;
      if (!GUARD || c < kc) r.w[s] = gload16_stream(B + (size_t)n * p.ldb + c * 512 + lane * 8);
    }
  }
  static __device__ __forceinline__ u32x4 bias(const GemmArgs& p, int n0, int wave, int lane) { return gv2_bias<T>(p, n0 + wave * 8); }
  // acc[m][j] += sum over this lane's 8 elements of every chunk (ascending): x rows from LDS, 16 bytes per lane and row; then
  // one DPP wave sum per (row, column), LDS, and lane 0 of wave g runs the fused epilogue on columns n0 + 8g .. + 7.
  // The tail of these launches is a chain of dependent latencies on one lane, so the bias it needs is requested ahead.  A
  // requested abs-max (COGV_EPI_ABSMAX) still costs the tail a memory-side read + atomic per finishing lane; the decode chain
  // does not ask for it any more (cogv_ln_prologue.z_absmax = NULL).  Measured with one returning-nothing atomic per lane
  // instead of atomic_max_nonneg's "read first": the 320 workgroups of the 4h -> h launch finish together and their
  // same-address atomics serialise, 14.6 -> 16.2 us (profiles/r04_decode_gemv2_kernel_stats*.csv).
  static __device__ __forceinline__ void product(const Regs& r, const GemmArgs& p, const T* xs, int XS, int K, Shared& sh, int n0, int wave,
                                                 int lane, const u32x4& bias_pre) {
    const int kc = K >> 9;
    float acc[MT][J];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < J; ++j) acc[m][j] = 0.f;
#pragma unroll
    for (int c = 0; c < KCMAX; ++c) {
      if (!GUARD || c < kc) {
        float x[MT][8];
#pragma unroll
        for (int m = 0; m < MT; ++m) unpack8<T>(*reinterpret_cast<const u32x4*>(xs + (size_t)m * XS + c * 512 + lane * 8), x[m]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
          float wf[8];
          unpack8<T>(r.w[c * J + j], wf);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            float t = acc[m][j];
#pragma unroll
            for (int e = 0; e < 8; ++e) t = fmaf(x[m][e], wf[e], t);
            acc[m][j] = t;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float t = wave_sum_uniform(acc[m][j]);
        if (lane == 0) sh.outp[m][wave * J + j] = t;
      }
    __syncthreads();
    constexpr int NG = (NW_ * J) / 8;
    if (lane == 0 && wave < NG) {
      const int n = n0 + wave * 8;
      if (n < p.N) {
        uint32_t am = 0u;
        for (int m = 0; m < p.M && m < MT; ++m) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = sh.outp[m][wave * 8 + i];
          am = absmax_pk(am, epilogue8<T>(p, m, n, v, &bias_pre));
        }
        if (p.flags & COGV_EPI_ABSMAX) {
          const uint32_t wv = max(am & 0xffffu, am >> 16);
          atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
        }
      }
    }
  }
};

// =====================================================================================================================
// FormM: 2 .. 8 rows on the matrix core.  A workgroup owns TW tiles of 16 columns, NWK waves per tile: wave kw of a tile takes the
// contraction-step PAIRS q = kw, kw + NWK, .. (steps 2q and 2q + 1: 64 consecutive elements = one 128-byte line of each of the
// 16 weight rows), L = K / 32 / NWK loads of 16 bytes per lane, all requested at the top like FormV's slots.
//   A operand = weights: lane l supplies row (l & 15) -> column n0 + (l & 15), contraction slots 8 (l >> 4) .. + 7 of the step
//   B operand = x:       lane l supplies column (l & 15) -> row m = l & 15 (lanes past the row count re-read the last row: those
//                        result columns are never looked at), the same contraction slots, from LDS (rows padded by 16 bytes
//                        so that the rows of one step fall into different banks)
//   D: lane l holds C[m = l & 15][n0 + 4 (l >> 4) + i], i = 0 .. 3.
// The waves' partial tiles are summed in wave order (deterministic), regrouped through LDS into 8 consecutive columns per
// (row, half) and finished by lanes 0 .. 2M - 1 of the tile's first wave in parallel.
// TW = 2 (eight waves, 32 columns) serves the LayerNorm-prologue kernel at 4 and 8 rows: one 8-element vector per thread and row
// in the prologue instead of two, and half as many workgroups recomputing it.
template <typename T, int NWK, int TW, int LMAX, bool GUARD_, int MT>
struct FormM {
  static constexpr int NW = NWK * TW, COLS = 16 * TW, XPAD = 8, NS = LMAX, XCHUNKS = (LMAX * NWK * 32 + 511) / 512;
  static constexpr bool GUARD = GUARD_;
  struct Regs { u32x4 w[LMAX]; };
  struct Shared { f32x4 red[NW][64]; float outp[TW][16][16]; };
  static __device__ __forceinline__ int kstep(int kw, int i) { return 2 * (kw + NWK * (i >> 1)) + (i & 1); }
  template <int S0, int S1>
  static __device__ __forceinline__ void issue(Regs& r, const GemmArgs& p, int n0, int K, int wave, int lane) {
    const int L = K / (32 * NWK), kw = wave % NWK, nt = n0 + (wave / NWK) * 16 + (lane & 15);
    const int n = nt < p.N ? nt : p.N - 1;
    const T* row = reinterpret_cast<const T*>(p.B) + (size_t)n * p.ldb + (lane >> 4) * 8;
#pragma unroll
    for (int i = S0; i < S1; ++i)
      if (!GUARD || i < L) r.w[i] = gload16(row + 32 * kstep(kw, i));     // (default policy: the non-temporal form measured 12-14 % SLOWER here, profiles/r05_decode_nt_ab.log)
  }
  // lanes 0 .. 15 of a tile's first wave finish: lane t -> row t >> 1, columns (tile) + 8 (t & 1) ..
  static __device__ __forceinline__ u32x4 bias(const GemmArgs& p, int n0, int wave, int lane) {
    return gv2_bias<T>(p, n0 + (wave / NWK) * 16 + (lane & 1) * 8);
  }
  // steps I0 .. I1 - 1 of this wave; xs holds the x rows from contraction index kofs on
  template <int I0, int I1>
  static __device__ __forceinline__ void accumulate(const Regs& r, const T* xs, int XS, int K, int kofs, int wave, int lane, f32x4& acc) {
    typedef typename HT<T>::v8 v8;
    const int L = K / (32 * NWK), kw = wave % NWK;
    const int mrow = (lane & 15) < MT ? (lane & 15) : MT - 1;
    const T* xrow = xs + (size_t)mrow * XS + (lane >> 4) * 8 - kofs;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      if (!GUARD || i < L) {
        const v8 b = *reinterpret_cast<const v8*>(xrow + 32 * kstep(kw, i));
        acc = HT<T>::mfma16(__builtin_bit_cast(v8, r.w[i]), b, acc);
      }
    }
  }
  static __device__ __forceinline__ void finish(f32x4 acc, const GemmArgs& p, Shared& sh, int n0, int wave, int lane, const u32x4& bias_pre) {
    const int tile = wave / NWK, kw = wave % NWK;
    sh.red[wave][lane] = acc;
    __syncthreads();
    if (kw == 0) {
      f32x4 s = sh.red[tile * NWK][lane];
#pragma unroll
      for (int w = 1; w < NWK; ++w) s += sh.red[tile * NWK + w][lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) sh.outp[tile][lane & 15][(lane >> 4) * 4 + i] = s[i];
    }
    __syncthreads();
    if (kw == 0) {
      const int m = lane >> 1, n = n0 + tile * 16 + (lane & 1) * 8;
      uint32_t am = 0u;
      if (m < p.M && m < MT && n < p.N) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = sh.outp[tile][m][(lane & 1) * 8 + i];
        am = epilogue8<T>(p, m, n, v, &bias_pre);
      }
      if (p.flags & COGV_EPI_ABSMAX) {
        uint32_t wv = max(am & 0xffffu, am >> 16);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wv = max(wv, (uint32_t)__shfl_xor((int)wv, o, 64));
        if (lane == 0) atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
      }
    }
  }
  static __device__ __forceinline__ void product(const Regs& r, const GemmArgs& p, const T* xs, int XS, int K, Shared& sh, int n0, int wave,
                                                 int lane, const u32x4& bias_pre) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    accumulate<0, LMAX>(r, xs, XS, K, 0, wave, lane, acc);
    finish(acc, p, sh, n0, wave, lane, bias_pre);
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// plain form: X = A (rows of the storage type, a few KB, L2-resident)
template <typename T, typename F, int MT>
__global__ __launch_bounds__(F::NW * 64) void gemv2_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char gv2_smem[];      // x [MT][K + pad] as T
  __shared__ typename F::Shared sh;
  T* xs = reinterpret_cast<T*>(gv2_smem);
  constexpr int NT = F::NW * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = F::GUARD ? p.K : F::XCHUNKS * 512, nvec = K >> 3, XS = K + F::XPAD;
  const int n0 = blockIdx.x * F::COLS;
  // x rows first: the copy to LDS below waits for these loads only
  constexpr int XV = (F::XCHUNKS * 64 + NT - 1) / NT;
  u32x4 xr[MT][XV];
  const T* A = reinterpret_cast<const T*>(p.A);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row = m < p.M ? m : p.M - 1;
#pragma unroll
    for (int u = 0; u < XV; ++u) {
      const int v = threadIdx.x + NT * u;
      if (v < nvec) xr[m][u] = gload16(A + (size_t)row * p.lda + v * 8);
    }
  }
  const u32x4 bias_pre = F::bias(p, n0, wave, lane);
  typename F::Regs w;
  F::template issue<0, F::NS>(w, p, n0, K, wave, lane);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int u = 0; u < XV; ++u) {
      const int v = threadIdx.x + NT * u;
      if (v < nvec) *reinterpret_cast<u32x4*>(xs + (size_t)m * XS + v * 8) = xr[m][u];
    }
  __syncthreads();
  F::product(w, p, xs, XS, K, sh, n0, wave, lane, bias_pre);
}

// plain form in TWO contraction halves (FormM, exact-K classes): x rows of K = 10240 with 4 / 8 rows are 80 / 160 KB -- more LDS
// than a workgroup can have next to anything else.  A wave's loads 0 .. L/2 - 1 lie in the first half of K and the others in
// the second (its step pairs ascend with the load index), so the rows are staged half by half: all weights are requested at
// the top as ever; the second half of x follows them in the memory queue, i.e. arrives right behind the last weight.
template <typename T, typename F, int MT>
__global__ __launch_bounds__(F::NW * 64) void gemv2_k2_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char gv2_smem[];      // x [MT][K / 2 + pad] as T
  __shared__ typename F::Shared sh;
  T* xs = reinterpret_cast<T*>(gv2_smem);
  constexpr int NT = F::NW * 64, K = F::XCHUNKS * 512, KH = K / 2, XS = KH + F::XPAD, NVEC = KH >> 3, XV = (NVEC + NT - 1) / NT;
  static_assert(!F::GUARD && F::NS % 4 == 0, "exact classes with an even number of step pairs per wave");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * F::COLS;
  const T* A = reinterpret_cast<const T*>(p.A);
  u32x4 xr[MT][XV];
  auto fetch = [&](int kofs) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = m < p.M ? m : p.M - 1;
#pragma unroll
      for (int u = 0; u < XV; ++u) {
        const int v = threadIdx.x + NT * u;
        if (v < NVEC) xr[m][u] = gload16(A + (size_t)row * p.lda + kofs + v * 8);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int u = 0; u < XV; ++u) {
        const int v = threadIdx.x + NT * u;
        if (v < NVEC) *reinterpret_cast<u32x4*>(xs + (size_t)m * XS + v * 8) = xr[m][u];
      }
  };
  fetch(0);
  const u32x4 bias_pre = F::bias(p, n0, wave, lane);
  typename F::Regs w;
  F::template issue<0, F::NS>(w, p, n0, K, wave, lane);
  stage();
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  F::template accumulate<0, F::NS / 2>(w, xs, XS, K, 0, wave, lane, acc);
  fetch(KH);
  __syncthreads();                       // every wave is done with the first half of x
  stage();
  __syncthreads();
  F::template accumulate<F::NS / 2, F::NS>(w, xs, XS, K, KH, wave, lane, acc);
  F::finish(acc, p, sh, n0, wave, lane, bias_pre);
}

// ---------------------------------------------------------------------------------------------------------------------
// attention-output projection with the COMBINE of the decode attention's key splits as prologue (see gemv_attn_kernel in
// gemm.hip for the partials' layout and the arithmetic: same association, att rounded to the storage type).  The waves
// share the combine (chunk c belongs to wave c % NW) and hand the combined vector over in LDS.  Only the first weight slots
// are requested in front of the combine: its reads of the partials would queue behind every load issued before them.
template <typename T, typename F, int MT>
__global__ __launch_bounds__(F::NW * 64) void gemv2_attn_kernel(const GemmArgs p, const float* __restrict__ part_ws, int H, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char gv2_smem[];
  __shared__ typename F::Shared sh;
  T* xs = reinterpret_cast<T*>(gv2_smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = F::GUARD ? p.K : F::XCHUNKS * 512, kc = K >> 9, XS = K + F::XPAD;
  const int n0 = blockIdx.x * F::COLS;
  constexpr int PRE = F::NS < 8 ? F::NS : 8;
  typename F::Regs w;
  F::template issue<0, PRE>(w, p, n0, K, wave, lane);
  for (int c = wave; c < kc; c += F::NW) {
    const int k = (c << 9) + lane * 8;
    const int head = k >> 6, dd = k & 63;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = m < p.M ? m : p.M - 1;
      const float* base = part_ws + ((size_t)row * H + head) * nsplit * 66;
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
      // the sum in the association of attn_decode_combine_kernel's xor-butterfly (lanes >= nsplit hold zeros there too)
#pragma unroll
      for (int w2 = 16; w2 > 0; w2 >>= 1)
#pragma unroll
        for (int i = 0; i < w2; ++i) tl[i] += tl[i + w2];
      const float L = tl[0];
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = o[e] / L;
      *reinterpret_cast<u32x4*>(xs + (size_t)m * XS + k) = pack8<T>(x);     // the attention output in its storage type
    }
  }
  F::template issue<PRE, F::NS>(w, p, n0, K, wave, lane);
  const u32x4 bias_pre = F::bias(p, n0, wave, lane);
  __syncthreads();
  F::product(w, p, xs, XS, K, sh, n0, wave, lane, bias_pre);
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm-prologue form (see GemvLnArgs in gemm_shared.cuh and gemv_ln_kernel in gemm.hip: the chain
//     z --[post-LN, Sandwich scale |z|max]--> + residual --> t --[pre-LN, Sandwich scale |t|max]--> x_in ;  y = epilogue(x_in W^T + b)
// of mpu/sparse_transformer.py:314-342 inside every workgroup; same arithmetic and rounding points).  K <= 4096: a thread owns
// NV = 512 / threads 8-element vectors (two with 4 waves, one with 8).  With up to two rows the whole weight stream of the wave
// is requested in front of the prologue; with more rows the prologue's registers leave room for the first 8 slots only.
// (one row: three waves per SIMD -- 168 registers -- so that the 640 workgroups of the h -> 4h launch are resident at once)
template <typename T, typename F, int MT, bool SF>
__global__ __launch_bounds__(F::NW * 64) __attribute__((amdgpu_waves_per_eu(MT == 1 ? 3 : 1)))
void gemv2_ln_kernel(const GemvLnArgs q) {
  typedef Row8<T, SF> SR;                // a stream row slice
  constexpr int NW = F::NW;
  extern __shared__ __attribute__((aligned(16))) char gv2_smem[];           // x_in [MT][K + pad] as T
  __shared__ float part[NW][MT][2];
  __shared__ typename F::Shared sh;
  __shared__ uint32_t redm[16];
  __shared__ float s_amax;
  constexpr int NT = NW * 64, NV = 512 / NT;
  static_assert(NW == 4 || NW == 8, "the prologue's block sums pair 4 or 8 wave partials");
  const GemmArgs& p = q.g;
  T* xs = reinterpret_cast<T*>(gv2_smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = F::GUARD ? p.K : F::XCHUNKS * 512, nvec = K >> 3, XS = K + F::XPAD;
  const float inv_k = 1.0f / (float)K;
  const bool has_post = q.gamma_p != nullptr;
  int vv[NV]; bool okv[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) { vv[u] = threadIdx.x + NT * u; okv[u] = vv[u] < nvec; }
  // ---- the prologue's own inputs FIRST (a few KB, L2-resident), then the weight stream.  The small loads are UNCONDITIONAL
  //      (clamped vector / row indices, a stand-in pointer where an operand is absent; the unwanted values are replaced by
  //      zeros below): a conditional load ends in a control-flow join at which the compiler drains the memory counter, and
  //      the weight loads behind it would start one memory latency late.
  const T* Z = reinterpret_cast<const T*>(q.z);
  const T* gpp = reinterpret_cast<const T*>(has_post ? q.gamma_p : q.gamma);
  const T* bpp = reinterpret_cast<const T*>(has_post ? q.beta_p : q.beta);
  const void* rsrc = has_post ? q.res : q.z;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 zr[MT][NV], gpr[NV], bpr[NV], gnr[NV], bnr[NV];
  typename SR::raw rr[MT][NV];           // the stream rows: the residual (post-LN form) or z itself (plain-input form)
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int v = okv[u] ? vv[u] : 0;
    gnr[u] = gload16(reinterpret_cast<const T*>(q.gamma) + v * 8);
    bnr[u] = gload16(reinterpret_cast<const T*>(q.beta) + v * 8);
    gpr[u] = gload16(gpp + v * 8);
    bpr[u] = gload16(bpp + v * 8);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = m < p.M ? m : p.M - 1;
      zr[m][u] = gload16(Z + (size_t)row * K + v * 8);        // (fp32 plain input: in bounds of the wider rows, discarded)
      rr[m][u] = SR::ld(rsrc, (size_t)row * K + v * 8);
    }
  }
  // (the published abs-max of z: a vector load as well, for gv2_bias's reason)
  const float zamax_raw = *((const COGV_GLOBAL float*)(q.z_absmax ? q.z_absmax : reinterpret_cast<const float*>(q.gamma)) + gv2_vzero());
  const int n0 = blockIdx.x * F::COLS;
  constexpr int PRE = MT <= 2 ? F::NS : (F::NS < 8 ? F::NS : 8);
  typename F::Regs w;
  F::template issue<0, PRE>(w, p, n0, K, wave, lane);
  float zamax = q.z_absmax ? zamax_raw : 0.f;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const bool ok = okv[u];
    gnr[u] = ok ? gnr[u] : zero4;
    bnr[u] = ok ? bnr[u] : zero4;
    gpr[u] = (ok && has_post) ? gpr[u] : zero4;
    bpr[u] = (ok && has_post) ? bpr[u] : zero4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      zr[m][u] = (ok && m < p.M && (has_post || !SF)) ? zr[m][u] : zero4;
      rr[m][u] = (ok && m < p.M && (has_post || SF)) ? rr[m][u] : SR::zero();
    }
  }
  // sums over the workgroup of MT values at once (one LDS round for all rows); pairwise over the waves
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
    for (int m = 0; m < MT; ++m) {
      float s = (part[0][m][0] + part[1][m][0]) + (part[2][m][0] + part[3][m][0]);
      if (NW == 8) s += (part[4 % NW][m][0] + part[5 % NW][m][0]) + (part[6 % NW][m][0] + part[7 % NW][m][0]);
      a[m] = s;
    }
  };
  float tv[MT][NV][8];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      if (SF && !has_post) SR::to_f(rr[m][u], tv[m][u]);         // the plain input IS the fp32 stream
      else unpack8<T>(zr[m][u], tv[m][u]);
    }
  if (has_post) {                 // t = residual + LN_post(z), rounded where ln_fwd_kernel rounds
    if (!q.z_absmax) {
      // max |z| over all rows taken HERE instead of published by z's producer: the vectors are in registers anyway, and the
      // producer (a skinny-M launch whose workgroups all finish together) is spared a burst of same-address atomics in its tail
      uint32_t zpk = 0u;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < NV; ++u) zpk = absmax_pk8(zpk, zr[m][u]);       // rows >= M and vectors past K are zero
      zamax = absmax_pk_block<T>(zpk, redm);
    }
    const float c = zamax * 0.125f;
    const float eps_p = q.eps * c * c;
    float s[MT], qq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      s[m] = 0.f;
#pragma unroll
      for (int u = 0; u < NV; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[m] += tv[m][u][i];            // vectors past K are zero
    }
    block_sums(s);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k;
      qq[m] = 0.f;
#pragma unroll
      for (int u = 0; u < NV; ++u)
        if (okv[u])
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = tv[m][u][i] - mean; qq[m] += d * d; }
    }
    block_sums(qq);
    float gp[NV][8], bp[NV][8];
#pragma unroll
    for (int u = 0; u < NV; ++u) { unpack8<T>(gpr[u], gp[u]); unpack8<T>(bpr[u], bp[u]); }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k, rstd = 1.0f / sqrtf(qq[m] * inv_k + eps_p);
#pragma unroll
      for (int u = 0; u < NV; ++u) {
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
        if (!okv[u]) {
#pragma unroll
          for (int i = 0; i < 8; ++i) tv[m][u][i] = 0.f;
        } else if (blockIdx.x == 0 && q.t_out && m < p.M)
          (void)SR::st(q.t_out, (size_t)m * K + vv[u] * 8, o, 0u);
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
      for (int u = 0; u < NV; ++u) {                                // rows >= M and vectors past K are zero
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
      for (int u = 0; u < NV; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[m] += tv[m][u][i];
    }
    block_sums(s);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k;
      qq[m] = 0.f;
#pragma unroll
      for (int u = 0; u < NV; ++u)
        if (okv[u])
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = tv[m][u][i] - mean; qq[m] += d * d; }
    }
    block_sums(qq);
    float gn[NV][8], bn[NV][8];
#pragma unroll
    for (int u = 0; u < NV; ++u) { unpack8<T>(gnr[u], gn[u]); unpack8<T>(bnr[u], bn[u]); }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k, rstd = 1.0f / sqrtf(qq[m] * inv_k + eps_n);
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        if (okv[u]) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (tv[m][u][i] - mean) * rstd * gn[u][i] + bn[u][i];
          *reinterpret_cast<u32x4*>(xs + (size_t)m * XS + vv[u] * 8) = pack8<T>(o);
        }
      }
    }
  }
  F::template issue<PRE, F::NS>(w, p, n0, K, wave, lane);
  // the epilogue's bias: requested here, behind the weights (it is needed after the last of them; the prologue's registers are
  // free again), still far ahead of its use
  const u32x4 bias_pre = F::bias(p, n0, wave, lane);
  __syncthreads();
  // ---- the matrix-vector product proper
  F::product(w, p, xs, XS, K, sh, n0, wave, lane, bias_pre);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side.  One row: FormV, (J, chunks) classes -- exact for the widths of the model family (h = 1024: K = 1024 / 4096;
// h = 2560: K = 2560 / 10240), any other multiple of 512 takes the guarded two-column form.  2 .. 8 rows: FormM, the waves of a
// workgroup by K (4 up to 2560, 8 up to 5120, 16 up to 10240): at most 20 loads per lane.
using TT = std::conditional<COGV_GEMV_TU != 0, f16_t, bf16_t>::type;

inline int gv2_mt(int M) { return M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : 8; }
// x rows in LDS (dynamic) + the kernels' static arrays stay inside the 64 KB a workgroup gets without an attribute; larger
// row blocks (K = 10240 with more than two rows) go to the first-generation kernels
constexpr size_t GV2_MAX_SHMEM = 56 * 1024;

template <int J, int KCMAX, bool G> using FV = FormV<TT, J, KCMAX, G, 1, 4>;
template <int NWK, int LMAX, bool G, int MT, int TW = 1> using FM = FormM<TT, NWK, TW, LMAX, G, MT>;

// CALL(F, MT) for 2 / 4 / 8 rows with FormM<NWK, TW, LMAX, G, MT>: TW4 = tiles per workgroup at 4 and 8 rows
#define GV2_M_SWITCH(mt, CALL, NWK_, LMAX_, G_, TW4_)                       \
  do {                                                                      \
    if ((mt) == 2) { CALL((FM<NWK_, LMAX_, G_, 2>), 2); }                   \
    else if ((mt) == 4) { CALL((FM<NWK_, LMAX_, G_, 4, TW4_>), 4); }        \
    else { CALL((FM<NWK_, LMAX_, G_, 8, TW4_>), 8); }                       \
  } while (0)

}  // namespace

#define GV2_CAT2(a, b) a##b
#define GV2_CAT(a, b) GV2_CAT2(a, b)
#define GV2_UNWRAP(...) __VA_ARGS__

// C = epilogue(A B^T), M <= 8.  COGV_ERR_UNSUPPORTED: the caller falls back to the first-generation kernel.
extern "C" __attribute__((visibility("hidden"))) int GV2_CAT(cogv_gemv2_launch_, COGV_GEMV_TU)(const void* args, void* stream) {
  const GemmArgs& a = *reinterpret_cast<const GemmArgs*>(args);
  const int mt = gv2_mt(a.M), kc = a.K >> 9;
  if (a.M < 1 || a.M > GEMV_MAX_M || (a.K & 511) || a.K > 10240 || (a.N & 7)) return COGV_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GV2_PLAIN(F_, MT_)                                                                                                   \
  do {                                                                                                                       \
    typedef GV2_UNWRAP F_ FF;                                                                                                \
    const size_t shmem = (size_t)MT_ * (a.K + FF::XPAD) * 2;                                                                 \
    if (shmem > GV2_MAX_SHMEM) return COGV_ERR_UNSUPPORTED;                                                                  \
    hipLaunchKernelGGL((gemv2_kernel<TT, FF, MT_>), dim3((a.N + FF::COLS - 1) / FF::COLS), dim3(FF::NW * 64), shmem, st, a); \
  } while (0)
  // the two-halves kernel: LDS for half the row length; above 64 KB in all (8 rows of K = 10240: 82 KB + 17 KB static) by attribute
#define GV2_PLAIN_K2(F_, MT_)                                                                                                \
  do {                                                                                                                       \
    typedef GV2_UNWRAP F_ FF;                                                                                                \
    const size_t shmem = (size_t)MT_ * (a.K / 2 + FF::XPAD) * 2;                                                             \
    static bool attr = false;                                                                                                \
    if (!attr) {                                                                                                             \
      if (shmem > GV2_MAX_SHMEM &&                                                                                           \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv2_k2_kernel<TT, FF, MT_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)shmem) != hipSuccess)                                                                    \
        return COGV_ERR_UNSUPPORTED;                                                                                         \
      attr = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL((gemv2_k2_kernel<TT, FF, MT_>), dim3((a.N + FF::COLS - 1) / FF::COLS), dim3(FF::NW * 64), shmem, st, a); \
  } while (0)
  if (mt == 1) {
    if (a.K == 1024) GV2_PLAIN((FV<8, 2, false>), 1);
    else if (a.K == 2560) GV2_PLAIN((FV<4, 5, false>), 1);
    else if (a.K == 4096) GV2_PLAIN((FV<2, 8, false>), 1);
    else if (a.K == 10240) GV2_PLAIN((FV<2, 20, false>), 1);
    else GV2_PLAIN((FV<2, 20, true>), 1);
  } else {
    if (a.K == 1024) GV2_M_SWITCH(mt, GV2_PLAIN, 4, 8, false, 1);
    else if (a.K == 2560) GV2_M_SWITCH(mt, GV2_PLAIN, 4, 20, false, 1);
    else if (a.K == 4096) GV2_M_SWITCH(mt, GV2_PLAIN, 8, 16, false, 1);
    else if (a.K == 10240) {
      if (mt == 2) GV2_PLAIN((FM<16, 20, false, 2>), 2);
      else if (mt == 4) GV2_PLAIN_K2((FM<16, 20, false, 4>), 4);
      else GV2_PLAIN_K2((FM<16, 20, false, 8>), 8);
    }
    else if (kc <= 5) GV2_M_SWITCH(mt, GV2_PLAIN, 4, 20, true, 1);
    else if (kc <= 10) GV2_M_SWITCH(mt, GV2_PLAIN, 8, 20, true, 1);
    else return COGV_ERR_UNSUPPORTED;
  }
#undef GV2_PLAIN
#undef GV2_PLAIN_K2
  return COGV_OK;
}

extern "C" __attribute__((visibility("hidden"))) int GV2_CAT(cogv_gemv2_attn_launch_, COGV_GEMV_TU)(const void* args, const float* partials, int heads,
                                                                                                  int nsplit, void* stream) {
  const GemmArgs& a = *reinterpret_cast<const GemmArgs*>(args);
  const int mt = gv2_mt(a.M), kc = a.K >> 9;
  if (a.M < 1 || a.M > GEMV_MAX_M || (a.K & 511) || a.K > 10240 || (a.N & 7) || nsplit > 32) return COGV_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GV2_ATTN(F_, MT_)                                                                                                         \
  do {                                                                                                                            \
    typedef GV2_UNWRAP F_ FF;                                                                                                     \
    const size_t shmem = (size_t)MT_ * (a.K + FF::XPAD) * 2;                                                                      \
    if (shmem > GV2_MAX_SHMEM) return COGV_ERR_UNSUPPORTED;                                                                       \
    hipLaunchKernelGGL((gemv2_attn_kernel<TT, FF, MT_>), dim3((a.N + FF::COLS - 1) / FF::COLS), dim3(FF::NW * 64), shmem, st, a,  \
                       partials, heads, nsplit);                                                                                  \
  } while (0)
  // (the same classes as cogv_gemv2_launch for every (K, M): the combine-prologue form and the two-launch form of the projection
  //  agree bit for bit -- in FormV a column's arithmetic does not even depend on the class)
  if (mt == 1) {
    if (a.K == 1024) GV2_ATTN((FV<8, 2, false>), 1);
    else if (a.K == 2560) GV2_ATTN((FV<4, 5, false>), 1);
    else GV2_ATTN((FV<2, 20, true>), 1);
  } else {
    if (a.K == 1024) GV2_M_SWITCH(mt, GV2_ATTN, 4, 8, false, 1);
    else if (a.K == 2560) GV2_M_SWITCH(mt, GV2_ATTN, 4, 20, false, 1);
    else if (a.K == 4096 || a.K == 10240) return COGV_ERR_UNSUPPORTED;     // (exact FormM classes of the plain form: not instantiated here)
    else if (kc <= 5) GV2_M_SWITCH(mt, GV2_ATTN, 4, 20, true, 1);
    else if (kc <= 10) GV2_M_SWITCH(mt, GV2_ATTN, 8, 20, true, 1);
    else return COGV_ERR_UNSUPPORTED;
  }
#undef GV2_ATTN
  return COGV_OK;
}

extern "C" __attribute__((visibility("hidden"))) int GV2_CAT(cogv_gemv2_ln_launch_, COGV_GEMV_TU)(const void* args, int stream_f32, void* stream) {
  const GemvLnArgs& a = *reinterpret_cast<const GemvLnArgs*>(args);
  const int mt = gv2_mt(a.g.M), kc = a.g.K >> 9;
  if (a.g.M < 1 || a.g.M > GEMV_MAX_M || (a.g.K & 511) || a.g.K > 4096 || (a.g.N & 7)) return COGV_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GV2_LN(F_, MT_)                                                                                                          \
  do {                                                                                                                           \
    typedef GV2_UNWRAP F_ FF;                                                                                                    \
    const size_t shmem = (size_t)MT_ * (a.g.K + FF::XPAD) * 2;                                                                   \
    if (shmem > GV2_MAX_SHMEM) return COGV_ERR_UNSUPPORTED;                                                                      \
    const dim3 grid((a.g.N + FF::COLS - 1) / FF::COLS), block(FF::NW * 64);                                                      \
    if (stream_f32) hipLaunchKernelGGL((gemv2_ln_kernel<TT, FF, MT_, true>), grid, block, shmem, st, a);                         \
    else hipLaunchKernelGGL((gemv2_ln_kernel<TT, FF, MT_, false>), grid, block, shmem, st, a);                                   \
  } while (0)
  if (mt == 1) {
    if (a.g.K == 1024) GV2_LN((FV<8, 2, false>), 1);
    else if (a.g.K == 2560) GV2_LN((FV<4, 5, false>), 1);
    else GV2_LN((FormV<TT, 2, 8, true, 1, 4>), 1);
  } else {
    if (a.g.K == 1024) GV2_M_SWITCH(mt, GV2_LN, 4, 8, false, 2);
    else if (a.g.K == 2560) GV2_M_SWITCH(mt, GV2_LN, 4, 20, false, 2);
    else if (kc <= 5) GV2_M_SWITCH(mt, GV2_LN, 4, 20, true, 2);
    else GV2_M_SWITCH(mt, GV2_LN, 8, 20, true, 1);
  }
#undef GV2_LN
  return COGV_OK;
}

#endif  // COGV_GEMV_TU
