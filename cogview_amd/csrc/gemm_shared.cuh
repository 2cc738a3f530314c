// Pieces of the GEMM family shared by its translation units (gemm.hip: the MFMA tile kernels; gemv.hip: the skinny-M
// matrix-vector kernels of the decode step): the argument block, the fused epilogue on 8 consecutive columns, the
// LayerNorm-prologue arguments.  Everything has internal linkage (anonymous namespace): launchers cross the unit boundary
// with a const void* to these structs, as the generation-4 units do.
#pragma once
#include "common.cuh"
#include "cogview_hip.h"

namespace {

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  int lda, ldb, ldc;
  const void* bias;     // [N] (T)
  void* aux;            // GELU: optional pre-activation out [M,ldaux]; DGELU: pre-activation in
  int ldaux;
  float* absmax;        // device scalar
  int flags;
  int out_f32;          // C is float (used by nothing but tests / future)
  uint64_t seed, stream_id;
  uint64_t drop_c0;     // dropout group counter of element (0, 0): the call computes rows [row0, row0 + M) of a larger tensor
  uint32_t thr16;       // dropout threshold (0 => keep all)
  float keep_scale;
  float* colsum_ws;     // COGV_EPI_COLSUM partial sums [2 * tiles_m(256)][N] fp32
  float* ws;            // split-K slabs [S][M][N] fp32
  int splitk;
  int ktiles_per_split;
  int tiles_m, tiles_n;
};

// ---- fused epilogue on 8 consecutive columns of one row (fp32 in registers)
//      F >= 0: the flag mask is a compile-time constant and the output is 16-bit (the generation-3 kernel dispatches
//      the hot combinations to such instances so that one item's epilogue is a few KB of code, not all paths).
// 16-byte accesses through EXPLICIT global-address-space pointers: the epilogues take their pointers from a descriptor
// copy pinned in scalar registers (pp64_epilogue), which hides the pointers' origin from address-space inference --
// a generic pointer would compile to flat_load / flat_store, which also count on the LDS counter
#define COGV_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ u32x4 gload16(const void* q) { return *(const COGV_GLOBAL u32x4*)q; }
// streamed-once data (the weight rows of a decode step: every byte is read by ONE workgroup, once per token): the non-temporal
// policy (global_load_dwordx4 ... nt) keeps the stream from displacing the step's small reused vectors in the caches.  Measured
// in the captured 4B decode step (profiles/r05_decode_nt_ab.log): one row (FormV, a wave streams whole weight rows) 2.78 -> 2.69
// ms per token; 2 / 4 rows (FormM: the 16 lanes of a 16 x 16 fragment read 16 different rows, 16 bytes each per step) 3.06 ->
// 3.50 and 3.74 -> 4.23 ms -- so FormV and the decode attention's cache rows take it, FormM keeps the default policy.
// COGV_DECODE_NT=0 builds default-policy loads everywhere.
#ifndef COGV_DECODE_NT
#define COGV_DECODE_NT 1
#endif
__device__ __forceinline__ u32x4 gload16_stream(const void* q) {
#if COGV_DECODE_NT
  return __builtin_nontemporal_load((const COGV_GLOBAL u32x4*)q);
#else
  return *(const COGV_GLOBAL u32x4*)q;
#endif
}
// the aux operand of the stored-tensor epilogues (MULAUX / DGELU: [M][N], every element read exactly once per launch): non-temporal,
// so that the 535-MB stream of the 4B dGeLU-side dgrad does not take L2 lines from the operand panels -- 1107 -> 1090 us in
// alternating processes (profiles/r06_gemm_aux_nt_probe.log).  What the aux read costs beyond that is its memory side: with an
// L2-resident aux (row stride 0, same instruction stream) the launch takes 85-110 us less (profiles/r06_gemm_aux_l2_probe.log).
#ifndef COGV_AUX_NT
#define COGV_AUX_NT 1
#endif
__device__ __forceinline__ u32x4 gload16_aux(const void* q) {
#if COGV_AUX_NT
  return __builtin_nontemporal_load((const COGV_GLOBAL u32x4*)q);
#else
  return *(const COGV_GLOBAL u32x4*)q;
#endif
}
__device__ __forceinline__ void gstore16(void* q, u32x4 v) { *(COGV_GLOBAL u32x4*)q = v; }
__device__ __forceinline__ void gstore16(float* q, f32x4 v) { *(COGV_GLOBAL f32x4*)q = v; }
// 16-bit C / aux stores of the non-accumulating epilogues carry the NON-TEMPORAL hint (round 4): a 32-CU XCD writes 4 MiB of C per
// round of tiles -- the size of its L2 -- through a write-allocating cache that also holds the operand panels.  Measured against
// plain stores in alternating processes (profiles/r04_gemm_nt_store_ab.log): QKV forward +1.7 %, GeLU + stored gelu' +2.5 % (+3-4 % at
// K = 1024), the other forward / dgrad launches 0 .. +1 %; the accumulating weight gradient, which reads C back, -0.4 % (kept plain).
#ifndef COGV_EXP
#define COGV_EXP 0        // probe builds only (gemm.hip documents the bits); 32768: the epilogue computes but does not store C / aux
#endif
template <bool NT>
__device__ __forceinline__ void gstore16c(void* q, u32x4 v) {
  if (COGV_EXP & 32768) { asm volatile("" : : "v"(v), "v"(q)); return; }
  if (NT) __builtin_nontemporal_store(v, (COGV_GLOBAL u32x4*)q); else *(COGV_GLOBAL u32x4*)q = v;
}

template <typename T, int F = -1>
//      bias_pre / aux_pre / c_pre: values the caller already loaded (the generation-3 epilogue issues all of a
//      sub-tile's dGeLU / accumulate reads up front instead of one exposed global-load latency per pass).
__device__ __forceinline__ uint32_t epilogue8(const GemmArgs& p, int m, int n, float (&v)[8], const u32x4* bias_pre = nullptr,
                                           const u32x4* aux_pre = nullptr, const u32x4* c_pre = nullptr, float* rounded = nullptr) {
  const int flags = F >= 0 ? F : p.flags;
  const bool out_f32 = F >= 0 ? false : (p.out_f32 != 0);
  constexpr bool NT = F >= 0 && !(F & COGV_EPI_ACCUM);
  if (flags & COGV_EPI_BIAS) {
    u32x4 bv = bias_pre ? *bias_pre : gload16(reinterpret_cast<const T*>(p.bias) + n);
    float b[8]; unpack8<T>(bv, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (flags & COGV_EPI_GELU) {
    if (flags & COGV_EPI_GELU_DAUX) {
      // aux receives gelu'(pre-activation) instead of the pre-activation: same bytes, and the backward GEMM's
      // epilogue (COGV_EPI_MULAUX) becomes one multiply per element.  gelu and gelu' are both taken from the SAME fp32
      // pre-activation here, so forward and backward stay consistent without rounding it to the storage type first
      // (the round trip below exists for the stored-pre-activation form, whose backward re-reads the rounded value);
      // 12 conversion instructions per 8 elements less in the most VALU-bound epilogue of the step.
      float gd[8];
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        f32x2_t g2, d2;
        gelu_and_grad_f2(f32x2_t{v[i], v[i + 1]}, g2, d2);
        v[i] = g2[0]; v[i + 1] = g2[1]; gd[i] = d2[0]; gd[i + 1] = d2[1];
      }
      if (p.aux) gstore16c<NT>(reinterpret_cast<T*>(p.aux) + (size_t)m * p.ldaux + n, pack8<T>(gd));
    } else {
      // when the pre-activation is STORED, the activation is evaluated on its rounded value -- exactly what the backward
      // pass (COGV_EPI_DGELU) will read; with nothing stored (inference) it is taken from the fp32 value like above
      if (p.aux) {
        const u32x4 rv = pack8<T>(v);
        unpack8<T>(rv, v);
        gstore16c<NT>(reinterpret_cast<T*>(p.aux) + (size_t)m * p.ldaux + n, rv);
      }
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const f32x2_t g2 = gelu_f2(f32x2_t{v[i], v[i + 1]});
        v[i] = g2[0]; v[i + 1] = g2[1];
      }
    }
  }
  if (flags & COGV_EPI_DGELU) {
    u32x4 uv = aux_pre ? *aux_pre : gload16_aux(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldaux + n);
    float u[8]; unpack8<T>(uv, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= gelu_grad_f(u[i]);
  }
  if (flags & COGV_EPI_MULAUX) {
    u32x4 uv = aux_pre ? *aux_pre : gload16_aux(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldaux + n);
    float u[8]; unpack8<T>(uv, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= u[i];
  }
  if ((flags & COGV_EPI_DROPOUT) && p.thr16) {
    const uint64_t e = (uint64_t)m * (uint64_t)p.N + (uint64_t)n;   // n % 8 == 0
    const u32x4 r = Philox::gen(p.seed, p.stream_id, (e >> 3) + p.drop_c0);
    // MARKED ZEROS (round 6): a dropped element is written as -0.0 and a kept element is never -0.0 -- a kept value that the
    // 16-bit rounding would turn into a zero of either sign is written as +0.0.  Numerically nothing changes (-0 == +0 in
    // every consumer), but the output now CARRIES its own keep mask: "16-bit pattern == 0x8000" <=> dropped, and the
    // Sandwich-LN backward that follows (cogv_sandwich_ln_bwd_marked) reads the mask from x instead of re-hashing it.
    // (DropTiny: |value| at or below it rounds to zero in T; bf16 uses FLT_MIN, i.e. also flushes fp32 denormals -- whatever
    //  the conversion does with them, no kept element can come out as -0.)
    constexpr float tiny = std::is_same<T, f16_t>::value ? 2.98023223876953125e-8f /* 2^-25 */ : 1.17549435e-38f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float kept = v[i] * p.keep_scale;
      v[i] = (drop_bits16(r, i) >= p.thr16) ? ((out_f32 || fabsf(kept) > tiny) ? kept : 0.f) : -0.f;
    }
  }
  if (flags & COGV_EPI_ACCUM) {
    if (out_f32) {
      const COGV_GLOBAL float* c = (const COGV_GLOBAL float*)(reinterpret_cast<const float*>(p.C) + (size_t)m * p.ldc + n);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += c[i];
    } else {
      u32x4 cv = c_pre ? *c_pre : gload16(reinterpret_cast<const T*>(p.C) + (size_t)m * p.ldc + n);
      float c[8]; unpack8<T>(cv, c);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += c[i];
    }
  }
  // returns the pair-wise max of the outputs' |value| bit patterns (see absmax_pk); 0 unless COGV_EPI_ABSMAX
  uint32_t amax_pk = 0u;
  if (out_f32) {
    float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
    gstore16(c, f32x4{v[0], v[1], v[2], v[3]});
    gstore16(c + 4, f32x4{v[4], v[5], v[6], v[7]});
    if (rounded) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rounded[i] = v[i];
    }
    if (flags & COGV_EPI_ABSMAX) amax_pk = absmax_pk8(0u, pack8<T>(v));      // fp32 output: max taken on the 16-bit rounding
  } else {
    u32x4 o = pack8<T>(v);
    gstore16c<NT>(reinterpret_cast<T*>(p.C) + (size_t)m * p.ldc + n, o);
    if (rounded) unpack8<T>(o, rounded);
    if (flags & COGV_EPI_ABSMAX) amax_pk = absmax_pk8(0u, o);
  }
  return amax_pk;
}

constexpr int GEMV_MAX_M = 8;

struct GemvLnArgs {
  GemmArgs g;                                   // B, bias, C, M, N, K, ldb, ldc, flags, absmax (of the output)
  const void* z; const float* z_absmax;         // [M][K] input; its abs-max (device scalar) -- required with a post-LN
  const void* gamma_p; const void* beta_p;      // post-LN affine (nullptr: no post-LN, t = z)
  const void* res; void* t_out;                 // residual [M][K] (with the post-LN); t_out [M][K] written by workgroup 0 (may be null)
  const void* gamma; const void* beta;          // pre-LN affine
  float eps;
};

}  // namespace
