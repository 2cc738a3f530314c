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
// One row (M = 1): workgroups of NW = 4 waves; 2 .. 8 rows: NW = 8 waves -- the LayerNorm prologue then holds ONE 8-element
// vector per thread and row instead of two, so that its registers leave room for the weight requests and a workgroup per CU
// (the whole launch) is resident; with 4 waves the two-row form already fell to two waves per SIMD and a second round of
// workgroups (profiles/r04_decode_batch_2_4_8_gemv2_ab.log: 2.78 ms per token at one row, 4.34 at two).
// The three kernels use the same association for the same K, so the combine-prologue form and the two-launch form of the
// attention-output projection still agree bit for bit (tests/test_kernels_gpu.py).
//
// Compiled as two translation units (-DCOGV_GEMV_TU=0: bf16, 1: fp16; build.py); without the macro this file is empty.
#include "gemm_shared.cuh"

#include <cstdlib>
#include <type_traits>

#ifdef COGV_GEMV_TU

namespace {

// slot s of a wave = (chunk c = s / J, column j = s % J): chunk-major, the order the dot products consume them.  Columns past N
// (a wave at the ragged end of the matrix) re-read the last row; their results are never stored.
template <typename T, int J, int KCMAX, int S0, int S1, bool GUARD>
__device__ __forceinline__ void gv2_issue(u32x4 (&w)[KCMAX * J], const T* B, size_t ldb, int nw, int N, int kc, int lane) {
#pragma unroll
  for (int s = S0; s < S1; ++s) {
    const int c = s / J, j = s % J;
    const int n = nw + j < N ? nw + j : N - 1;
    if (!GUARD || c < kc) w[s] = gload16(B + (size_t)n * ldb + c * 512 + lane * 8);
  }
}

// acc[m][j] += sum over this lane's 8 elements of every chunk: x rows from LDS (16 bytes per lane and row, conflict-free).
// Rows in groups of four (the unpacked x of eight rows next to the weight slots would not fit the register file of a
// 512-thread workgroup); a column's sum is the same fmaf chain whatever the grouping.
template <typename T, int J, int KCMAX, int MT, bool GUARD>
__device__ __forceinline__ void gv2_compute(const u32x4 (&w)[KCMAX * J], const T* xs, int K, int kc, int lane, float (&acc)[MT][J]) {
  constexpr int MG = MT < 4 ? MT : 4;
#pragma unroll
  for (int c = 0; c < KCMAX; ++c) {
    if (!GUARD || c < kc) {
#pragma unroll
      for (int m0 = 0; m0 < MT; m0 += MG) {
        float x[MG][8];
#pragma unroll
        for (int m = 0; m < MG; ++m) unpack8<T>(*reinterpret_cast<const u32x4*>(xs + (size_t)(m0 + m) * K + c * 512 + lane * 8), x[m]);
#pragma unroll
        for (int j = 0; j < J; ++j) {
          float wf[8];
          unpack8<T>(w[c * J + j], wf);
#pragma unroll
          for (int m = 0; m < MG; ++m) {
            float t = acc[m0 + m][j];
#pragma unroll
            for (int e = 0; e < 8; ++e) t = fmaf(x[m][e], wf[e], t);
            acc[m0 + m][j] = t;
          }
        }
        // (four and eight rows: keep the scheduler from hoisting every chunk's LDS reads to the top -- 160 registers of x)
        if (MT >= 4) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// a zero the compiler cannot see through, in a vector register: added to a wave-uniform address it keeps the load on the
// vector memory path (see gv2_bias)
__device__ __forceinline__ int gv2_vzero() {
  int zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
  return zero;
}

// bias of the 8 columns lane 0 of this wave finishes (gv2_finish), requested ahead of its use.  Unconditional: without a bias
// (or in a wave that finishes nothing) a valid stand-in address is read and the value ignored.
template <typename T>
__device__ __forceinline__ u32x4 gv2_bias(const GemmArgs& p, int n0, int wave) {
  const int n = n0 + wave * 8;
  const T* src = (p.flags & COGV_EPI_BIAS) ? reinterpret_cast<const T*>(p.bias) : reinterpret_cast<const T*>(p.B);
  // through the VECTOR memory path (an offset the compiler cannot see through): as a wave-uniform address this would become a
  // scalar load, and the scalar counter has to reach zero -- for the kernel arguments -- before the first weight load can be
  // issued: the weight stream would start one memory latency late
  return gload16(src + (n < p.N ? n : p.N - 8) + gv2_vzero());
}

// wave sums -> LDS -> lane 0 of wave g runs the fused epilogue on columns n0 + 8g .. + 7 (g < NW * J / 8), rows in order.
// The tail of these launches is a chain of dependent latencies on one lane, so the bias it needs is requested ahead
// (bias_pre: gv2_bias).  A requested abs-max (COGV_EPI_ABSMAX) still costs the tail a memory-side read + atomic per finishing
// lane; the decode chain does not ask for it any more (the consuming launch's LayerNorm prologue takes max|z| itself:
// cogv_ln_prologue.z_absmax = NULL).  Measured with one returning-nothing atomic per lane instead of atomic_max_nonneg's
// "read first": the 320 workgroups of the 4h -> h launch finish together and their same-address atomics serialise,
// 14.6 -> 16.2 us (profiles/r04_decode_gemv2_kernel_stats*.csv).
template <typename T, int J, int MT, int NW>
__device__ __forceinline__ void gv2_finish(const GemmArgs& p, float (&acc)[MT][J], float (*outp)[NW * J], int n0, int lane, int wave,
                                           const u32x4& bias_pre) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float t = wave_sum_uniform(acc[m][j]);
      if (lane == 0) outp[m][wave * J + j] = t;
    }
  __syncthreads();
  constexpr int NG = (NW * J) / 8;
  if (lane == 0 && wave < NG) {
    const int n = n0 + wave * 8;
    if (n < p.N) {
      uint32_t am = 0u;
      for (int m = 0; m < p.M && m < MT; ++m) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = outp[m][wave * 8 + i];
        am = absmax_pk(am, epilogue8<T>(p, m, n, v, &bias_pre));
      }
      if (p.flags & COGV_EPI_ABSMAX) {
        const uint32_t wv = max(am & 0xffffu, am >> 16);
        atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// plain form: X = A (rows of the storage type, a few KB, L2-resident)
template <typename T, int J, int KCMAX, bool GUARD, int MT, int NW>
__global__ __launch_bounds__(NW * 64) void gemv2_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char gv2_smem[];      // x [MT][K] as T
  __shared__ float outp[MT][NW * J];
  T* xs = reinterpret_cast<T*>(gv2_smem);
  constexpr int NT = NW * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = GUARD ? p.K : KCMAX * 512, kc = K >> 9, nvec = K >> 3;
  const int n0 = blockIdx.x * NW * J;
  // x rows first: the copy to LDS below waits for these loads only
  constexpr int XV = (KCMAX * 64 + NT - 1) / NT;
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
  const u32x4 bias_pre = gv2_bias<T>(p, n0, wave);
  u32x4 w[KCMAX * J];
  gv2_issue<T, J, KCMAX, 0, KCMAX * J, GUARD>(w, reinterpret_cast<const T*>(p.B), (size_t)p.ldb, n0 + wave * J, p.N, kc, lane);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int u = 0; u < XV; ++u) {
      const int v = threadIdx.x + NT * u;
      if (v < nvec) *reinterpret_cast<u32x4*>(xs + (size_t)m * K + v * 8) = xr[m][u];
    }
  __syncthreads();
  float acc[MT][J];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < J; ++j) acc[m][j] = 0.f;
  gv2_compute<T, J, KCMAX, MT, GUARD>(w, xs, K, kc, lane, acc);
  gv2_finish<T, J, MT, NW>(p, acc, outp, n0, lane, wave, bias_pre);
}

// ---------------------------------------------------------------------------------------------------------------------
// attention-output projection with the COMBINE of the decode attention's key splits as prologue (see gemv_attn_kernel in
// gemm.hip for the partials' layout and the arithmetic: same association, att rounded to the storage type).  The waves
// share the combine (chunk c belongs to wave c % NW) and hand the combined vector over in LDS.  Only the first weight slots
// are requested in front of the combine: its reads of the partials would queue behind every load issued before them.
template <typename T, int J, int KCMAX, bool GUARD, int MT, int NW>
__global__ __launch_bounds__(NW * 64) void gemv2_attn_kernel(const GemmArgs p, const float* __restrict__ part_ws, int H, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char gv2_smem[];
  __shared__ float outp[MT][NW * J];
  T* xs = reinterpret_cast<T*>(gv2_smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = GUARD ? p.K : KCMAX * 512, kc = K >> 9;
  const int n0 = blockIdx.x * NW * J;
  const T* B = reinterpret_cast<const T*>(p.B);
  constexpr int NS = KCMAX * J, PRE = NS < 8 ? NS : 8;
  u32x4 w[NS];
  gv2_issue<T, J, KCMAX, 0, PRE, GUARD>(w, B, (size_t)p.ldb, n0 + wave * J, p.N, kc, lane);
  for (int c = wave; c < kc; c += NW) {
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
      *reinterpret_cast<u32x4*>(xs + (size_t)m * K + k) = pack8<T>(x);      // the attention output in its storage type
    }
  }
  gv2_issue<T, J, KCMAX, PRE, NS, GUARD>(w, B, (size_t)p.ldb, n0 + wave * J, p.N, kc, lane);
  const u32x4 bias_pre = gv2_bias<T>(p, n0, wave);
  __syncthreads();
  float acc[MT][J];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < J; ++j) acc[m][j] = 0.f;
  gv2_compute<T, J, KCMAX, MT, GUARD>(w, xs, K, kc, lane, acc);
  gv2_finish<T, J, MT, NW>(p, acc, outp, n0, lane, wave, bias_pre);
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm-prologue form (see GemvLnArgs in gemm_shared.cuh and gemv_ln_kernel in gemm.hip: the chain
//     z --[post-LN, Sandwich scale |z|max]--> + residual --> t --[pre-LN, Sandwich scale |t|max]--> x_in ;  y = epilogue(x_in W^T + b)
// of mpu/sparse_transformer.py:314-342 inside every workgroup; same arithmetic and rounding points).  K <= 4096: a thread owns
// NV = 512 / threads 8-element vectors (two with 4 waves, one with 8).  With up to two rows the whole weight stream of the wave
// is requested in front of the prologue; with more rows the prologue's registers leave room for the first 8 slots only.
// (one row: three waves per SIMD -- 168 registers -- so that the 640 workgroups of the h -> 4h launch are resident at once)
template <typename T, int MT, bool SF, int J, int KCMAX, bool GUARD, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(MT == 1 ? 3 : 1)))
void gemv2_ln_kernel(const GemvLnArgs q) {
  typedef Row8<T, SF> SR;                // a stream row slice
  extern __shared__ __attribute__((aligned(16))) char gv2_smem[];           // x_in [MT][K] as T
  __shared__ float part[NW][MT][2];
  __shared__ float outp[MT][NW * J];
  __shared__ uint32_t redm[16];
  __shared__ float s_amax;
  constexpr int NT = NW * 64, NV = 512 / NT;
  const GemmArgs& p = q.g;
  T* xs = reinterpret_cast<T*>(gv2_smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = GUARD ? p.K : KCMAX * 512, kc = K >> 9, nvec = K >> 3;
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
  const int n0 = blockIdx.x * NW * J;
  const T* B = reinterpret_cast<const T*>(p.B);
  constexpr int NS = KCMAX * J, PRE = MT <= 2 ? NS : (NS < 8 ? NS : 8);
  u32x4 w[NS];
  gv2_issue<T, J, KCMAX, 0, PRE, GUARD>(w, B, (size_t)p.ldb, n0 + wave * J, p.N, kc, lane);
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
          *reinterpret_cast<u32x4*>(xs + (size_t)m * K + vv[u] * 8) = pack8<T>(o);
        }
      }
    }
  }
  gv2_issue<T, J, KCMAX, PRE, NS, GUARD>(w, B, (size_t)p.ldb, n0 + wave * J, p.N, kc, lane);
  // the epilogue's bias: requested here, behind the weights (it is needed after the last of them; the prologue's registers are
  // free again), still far ahead of its use
  const u32x4 bias_pre = gv2_bias<T>(p, n0, wave);
  __syncthreads();
  // ---- the matrix-vector product proper
  float acc[MT][J];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < J; ++j) acc[m][j] = 0.f;
  gv2_compute<T, J, KCMAX, MT, GUARD>(w, xs, K, kc, lane, acc);
  gv2_finish<T, J, MT, NW>(p, acc, outp, n0, lane, wave, bias_pre);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side: the (J, chunks) class of a contraction length.  Exact classes for the widths of the model family (h = 1024:
// K = 1024 / 4096; h = 2560: K = 2560 / 10240); any other multiple of 512 takes the guarded two-column form.  One row runs in
// workgroups of 4 waves, 2 .. 8 rows in workgroups of 8 (GV2_NW).
using TT = std::conditional<COGV_GEMV_TU != 0, f16_t, bf16_t>::type;

inline int gv2_mt(int M) { return M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : 8; }
// x rows in LDS (dynamic) + the kernels' static arrays stay inside the 64 KB a workgroup gets without an attribute; larger
// row blocks (K = 10240 with more than two rows, K = 4096 with eight) go to the first-generation kernels
constexpr size_t GV2_MAX_SHMEM = 60 * 1024;
// waves per workgroup by row count, as measured on the captured 4B step (profiles/r04_decode_batch_*): 2 rows 4.34 ms per token
// with 4 waves / 4.68 with 8; 4 rows 5.63 / 4.90; 8 rows 9.36 / 18.97 (8 waves: 256 registers per lane, spills)
#define GV2_MT_SWITCH(mt, CALL)  \
  do {                           \
    if ((mt) == 1) { CALL(1, 4); }  \
    else if ((mt) == 2) { CALL(2, 4); } \
    else if ((mt) == 4) { CALL(4, 8); } \
    else { CALL(8, 4); }            \
  } while (0)

}  // namespace

#define GV2_CAT2(a, b) a##b
#define GV2_CAT(a, b) GV2_CAT2(a, b)

// C = epilogue(A B^T), M <= 8.  COGV_ERR_UNSUPPORTED: the caller falls back to the first-generation kernel.
extern "C" __attribute__((visibility("hidden"))) int GV2_CAT(cogv_gemv2_launch_, COGV_GEMV_TU)(const void* args, void* stream) {
  const GemmArgs& a = *reinterpret_cast<const GemmArgs*>(args);
  const int mt = gv2_mt(a.M);
  const size_t shmem = (size_t)mt * a.K * 2;
  if (a.M < 1 || a.M > GEMV_MAX_M || (a.K & 511) || a.K > 10240 || (a.N & 7) || shmem > GV2_MAX_SHMEM) return COGV_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GV2_PLAIN(J_, KC_, G_, MT_, NW_)                                                                                   \
  hipLaunchKernelGGL((gemv2_kernel<TT, J_, KC_, G_, MT_, NW_>), dim3((a.N + NW_ * J_ - 1) / (NW_ * J_)), dim3(NW_ * 64), shmem, st, a)
#define C1024(MT_, NW_) GV2_PLAIN(8, 2, false, MT_, NW_)
#define C2560(MT_, NW_) GV2_PLAIN(4, 5, false, MT_, NW_)
// (K = 4096 / 10240: 16 / 40 weight slots per wave next to the accumulators -- 4 waves per workgroup whatever the row count, so
//  that the register file of the workgroup can take what 512-thread workgroups could only spill)
#define C4096(MT_, NW_) GV2_PLAIN(2, 8, false, MT_, 4)
#define C10240(MT_, NW_) GV2_PLAIN(2, 20, false, MT_, 4)
// K = 10240 with 2 rows: ONE column per wave (20 slots = 80 registers), 8 waves -- the two-column form needed 160 registers
// of weights next to the row staging and spilled INSIDE the request phase (a spilled slot is waited for, stored, reloaded: the
// 4h -> h launch of a two-row step took 34.8 us against 14.1 with one row).  Four rows (80 KB of x, and 640 B of spills per
// lane even in this form) stay with the first generation.
#define C10240W(MT_) GV2_PLAIN(1, 20, false, MT_, 8)
#define CGEN(MT_, NW_) GV2_PLAIN(2, 20, true, MT_, NW_)
  if (a.K == 1024) GV2_MT_SWITCH(mt, C1024);
  else if (a.K == 2560) GV2_MT_SWITCH(mt, C2560);
  else if (a.K == 4096) GV2_MT_SWITCH(mt, C4096);
  else if (a.K == 10240) {
    if (mt == 1) { C10240(1, 4); }
    else { C10240W(2); }                 // mt <= 2 by the LDS bound
  }
  else GV2_MT_SWITCH(mt, CGEN);
#undef C1024
#undef C2560
#undef C4096
#undef C10240
#undef C10240W
#undef CGEN
#undef GV2_PLAIN
  return COGV_OK;
}

extern "C" __attribute__((visibility("hidden"))) int GV2_CAT(cogv_gemv2_attn_launch_, COGV_GEMV_TU)(const void* args, const float* partials, int heads,
                                                                                                  int nsplit, void* stream) {
  const GemmArgs& a = *reinterpret_cast<const GemmArgs*>(args);
  const int mt = gv2_mt(a.M);
  const size_t shmem = (size_t)mt * a.K * 2;
  if (a.M < 1 || a.M > GEMV_MAX_M || (a.K & 511) || a.K > 10240 || (a.N & 7) || shmem > GV2_MAX_SHMEM || nsplit > 32) return COGV_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GV2_ATTN(J_, KC_, G_, MT_, NW_)                                                                                    \
  hipLaunchKernelGGL((gemv2_attn_kernel<TT, J_, KC_, G_, MT_, NW_>), dim3((a.N + NW_ * J_ - 1) / (NW_ * J_)), dim3(NW_ * 64), shmem, st, \
                     a, partials, heads, nsplit)
#define C1024(MT_, NW_) GV2_ATTN(8, 2, false, MT_, NW_)
#define C2560(MT_, NW_) GV2_ATTN(4, 5, false, MT_, NW_)
#define CGEN(MT_, NW_) GV2_ATTN(2, 20, true, MT_, NW_)
  // (a column's arithmetic does not depend on the class: chunks in ascending order per lane, one wave sum -- so this form and
  //  cogv_gemv2_launch agree bit for bit whatever class either takes)
  if (a.K == 1024) GV2_MT_SWITCH(mt, C1024);
  else if (a.K == 2560) GV2_MT_SWITCH(mt, C2560);
  else GV2_MT_SWITCH(mt, CGEN);
#undef C1024
#undef C2560
#undef CGEN
#undef GV2_ATTN
  return COGV_OK;
}

extern "C" __attribute__((visibility("hidden"))) int GV2_CAT(cogv_gemv2_ln_launch_, COGV_GEMV_TU)(const void* args, int stream_f32, void* stream) {
  const GemvLnArgs& a = *reinterpret_cast<const GemvLnArgs*>(args);
  const int mt = gv2_mt(a.g.M);
  const size_t shmem = (size_t)mt * a.g.K * 2;
  if (a.g.M < 1 || a.g.M > GEMV_MAX_M || (a.g.K & 511) || a.g.K > 4096 || (a.g.N & 7) || shmem > GV2_MAX_SHMEM) return COGV_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define GV2_LN(J_, KC_, G_, MT_, NW_)                                                                                          \
  do {                                                                                                                           \
    const dim3 grid((a.g.N + NW_ * J_ - 1) / (NW_ * J_)), block(NW_ * 64);                                                       \
    if (stream_f32) hipLaunchKernelGGL((gemv2_ln_kernel<TT, MT_, true, J_, KC_, G_, NW_>), grid, block, shmem, st, a);           \
    else hipLaunchKernelGGL((gemv2_ln_kernel<TT, MT_, false, J_, KC_, G_, NW_>), grid, block, shmem, st, a);                     \
  } while (0)
#define C1024(MT_, NW_) GV2_LN(8, 2, false, MT_, NW_)
#define C2560(MT_, NW_) GV2_LN(4, 5, false, MT_, NW_)
#define CGEN(MT_, NW_) GV2_LN(2, 8, true, MT_, NW_)
  if (a.g.K == 1024) GV2_MT_SWITCH(mt, C1024);
  else if (a.g.K == 2560) GV2_MT_SWITCH(mt, C2560);
  else GV2_MT_SWITCH(mt, CGEN);
#undef C1024
#undef C2560
#undef CGEN
#undef GV2_LN
  return COGV_OK;
}

#endif  // COGV_GEMV_TU
