// Common device helpers for the CogView MI355X (gfx950 / CDNA4) hot-path library.
// Wave = 64 lanes everywhere.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#define COGV_OK 0
#define COGV_ERR_ARG 1      // bad argument (shape/alignment/dtype)
#define COGV_ERR_LAUNCH 2   // hipGetLastError() after launch
#define COGV_ERR_UNSUPPORTED 3

#define COGV_F16 0
#define COGV_BF16 1
#define COGV_F32 2

typedef _Float16 f16_t;
typedef __bf16 bf16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// ---------------------------------------------------------------- half traits
template <typename T> struct HT;
template <> struct HT<f16_t> {
  typedef f16x8 v8;
  static __device__ __forceinline__ float to_f(f16_t x) { return (float)x; }
  static __device__ __forceinline__ f16_t from_f(float x) { return (f16_t)x; }
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct HT<bf16_t> {
  typedef bf16x8 v8;
  static __device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
  static __device__ __forceinline__ bf16_t from_f(float x) { return (bf16_t)x; }
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// 16-bit payload <-> float via raw bits (used on packed u32 words)
template <typename T> __device__ __forceinline__ float bits_to_f(uint16_t b);
template <> __device__ __forceinline__ float bits_to_f<f16_t>(uint16_t b) {
  f16_t h; __builtin_memcpy(&h, &b, 2); return (float)h;
}
template <> __device__ __forceinline__ float bits_to_f<bf16_t>(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
template <typename T> __device__ __forceinline__ uint16_t f_to_bits(float x) {
  T h = (T)x; uint16_t b; __builtin_memcpy(&b, &h, 2); return b;
}
// two floats -> one packed 16-bit pair, round-to-nearest-even.  Through a 2-vector conversion: gfx950 has
// v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, but two scalar casts compile to two converts + shift + or (4 instructions).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) {
  const bf16x2_t h = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
  uint32_t u; __builtin_memcpy(&u, &h, 4); return u;
}
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) {
  const f16x2_t h = __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t);
  uint32_t u; __builtin_memcpy(&u, &h, 4); return u;
}
template <typename T> __device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bits_to_f<T>((uint16_t)(v[i] & 0xffffu));
    f[2 * i + 1] = bits_to_f<T>((uint16_t)(v[i] >> 16));
  }
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2<T>(f[2 * i], f[2 * i + 1]);
  return v;
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Wave-wide sum through DPP only (no LDS crossbar): quad butterflies, half-row and row mirrors, then the two
// cross-row broadcasts; the total lands in lane 63 and is returned as a wave-uniform value.
__device__ __forceinline__ float wave_sum_uniform(float v) {
  auto dpp_add = [](float x, auto ctrl, auto row_mask) {
    constexpr int C = decltype(ctrl)::value, RM = decltype(row_mask)::value;
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), C, RM, 0xf, false);
    return x + __builtin_bit_cast(float, y);
  };
  using IC = std::integral_constant<int, 0>;
  (void)sizeof(IC);
  v = dpp_add(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
  v = dpp_add(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
  v = dpp_add(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});   // row_half_mirror
  v = dpp_add(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});   // row_mirror
  v = dpp_add(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast15 -> rows 1, 3
  v = dpp_add(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block reductions (blockDim.x multiple of 64, <= 1024); smem >= 16 floats
__device__ __forceinline__ float block_sum(float v, float* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = smem[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, smem[i]);
  return r;
}

// Abs-max of 16-bit floats on their BIT PATTERNS: with the sign cleared, integer order is magnitude order for fp16 and
// bf16 alike, and every NaN pattern sorts above infinity -- so one v_pk_max_u16 per pair replaces unpack + fabs +
// max + NaN test, and NaN propagates by itself (the reference's x.abs().max() semantics).
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t absmax_pk(uint32_t m, uint32_t packed) {
  const uint32_t x = packed & 0x7fff7fffu;
  u16x2_t a, b;
  __builtin_memcpy(&a, &m, 4); __builtin_memcpy(&b, &x, 4);
  a = __builtin_elementwise_max(a, b);
  __builtin_memcpy(&m, &a, 4);
  return m;
}
__device__ __forceinline__ uint32_t absmax_pk8(uint32_t m, const u32x4& v) {
  return absmax_pk(absmax_pk(absmax_pk(absmax_pk(m, v[0]), v[1]), v[2]), v[3]);
}
// block-wide max of the running pair, returned as the fp32 value of the largest pattern (NaN stays NaN)
template <typename T>
__device__ __forceinline__ float absmax_pk_block(uint32_t m, uint32_t* smem) {
  uint32_t v = max(m & 0xffffu, m >> 16);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  uint32_t r = smem[0];
  for (int i = 1; i < nw; ++i) r = max(r, smem[i]);
  return bits_to_f<T>((uint16_t)r);
}

// ---------------------------------------------------------------- the fp32 residual stream
// Round 3: the residual stream of the transformer (embedding output, y = x + LN3(.), out = y + LN4(.), and its
// gradient) is held in fp32; everything that feeds a GEMM stays in the 16-bit storage type.  Rounding the stream to
// 16 bits twice per layer was THE depth-dependent error of the forward pass: logits rel-L2 against the fp32 reference
// grew from 5.6e-4 after one layer to 1.8e-3 after 48 (fp16; bf16 4.5e-3 -> 1.4e-2), all but 6 % of the per-layer
// growth being those two roundings (tools/r3/depth_probe.py, profiles/r03_depth_parity.log).
// Row8<T, F32>: the 8 consecutive elements of a row one lane owns -- one 16-byte access in the storage type T, two for
// the fp32 stream.
struct raw8_f32 { u32x4 a, b; };
template <typename T, bool F32> struct Row8;
template <typename T> struct Row8<T, false> {
  typedef u32x4 raw;
  static __device__ __forceinline__ raw zero() { return u32x4{0u, 0u, 0u, 0u}; }
  static __device__ __forceinline__ raw ld(const void* base, size_t idx) {
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(base) + idx);
  }
  static __device__ __forceinline__ void to_f(const raw& r, float* f) { unpack8<T>(r, f); }
  // stores the values rounded to T and hands the rounded values back in f
  static __device__ __forceinline__ uint32_t st(void* base, size_t idx, float* f, uint32_t amax) {
    const u32x4 v = pack8<T>(f);
    *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(base) + idx) = v;
    unpack8<T>(v, f);
    return absmax_pk8(amax, v);
  }
};
template <typename T> struct Row8<T, true> {
  typedef raw8_f32 raw;
  static __device__ __forceinline__ raw zero() { return raw8_f32{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}}; }
  static __device__ __forceinline__ raw ld(const void* base, size_t idx) {
    const u32x4* p = reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(base) + idx);
    return raw8_f32{p[0], p[1]};
  }
  static __device__ __forceinline__ void to_f(const raw& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = __uint_as_float(r.a[i]); f[4 + i] = __uint_as_float(r.b[i]); }
  }
  // amax: running integer max of the |value| bit patterns (NaN patterns sort above infinity, as in absmax_pk)
  static __device__ __forceinline__ uint32_t st(void* base, size_t idx, float* f, uint32_t amax) {
    u32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = __float_as_uint(f[i]); b[i] = __float_as_uint(f[4 + i]); }
    u32x4* p = reinterpret_cast<u32x4*>(reinterpret_cast<float*>(base) + idx);
    p[0] = a; p[1] = b;
#pragma unroll
    for (int i = 0; i < 4; ++i) amax = max(amax, max(a[i] & 0x7fffffffu, b[i] & 0x7fffffffu));
    return amax;
  }
};
// block-wide max of fp32 |value| bit patterns, returned as the float with that pattern
__device__ __forceinline__ float absmax_f32_block(uint32_t v, uint32_t* smem) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  uint32_t r = smem[0];
  for (int i = 1; i < nw; ++i) r = max(r, smem[i]);
  return __uint_as_float(r);
}

// non-negative float atomic max through the integer ordering of IEEE-754.
// Same-address atomics serialise at L2 (~90 per microsecond on MI355X), and a launch has thousands of
// workgroups, so the current value is read first (relaxed, agent scope -> served by L2) and the atomic is
// issued only by workgroups that would raise it.  A stale read can only be SMALLER than the true value
// (the cell is monotonic within a launch), i.e. it costs a redundant atomic, never a missed update.
// NaN is published as the quiet-NaN pattern 0x7fc00000, which is larger than every finite pattern, so the
// consumer's LayerNorm goes NaN exactly as the reference's x.abs().max() would make it.
__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
  const unsigned int vi = __float_as_uint(v);
  unsigned int* a = reinterpret_cast<unsigned int*>(addr);
  const unsigned int cur = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (vi > cur) atomicMax(a, vi);
}

// ---------------------------------------------------------------- counter-based dropout RNG
// Dropout masks are a pure function of (seed, stream, element index) so that backward kernels (and
// activation-checkpoint recompute) regenerate them instead of storing them.  Philox4x32-10 costs forty
// quarter-rate 32-bit multiplies per call on CDNA; here one PCG output-permutation hash (2 multiplies) of the
// group counter is expanded with xorshift32 steps (full-rate ops only).  Bit-parity with torch's generator
// is impossible either way (reference mpu/random.py forks the CUDA Philox state); tests check statistics
// and forward/backward mask consistency.
__host__ __device__ __forceinline__ uint32_t pcg32(uint32_t x) {
  const uint32_t state = x * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
__host__ __device__ __forceinline__ uint32_t xorshift32(uint32_t x) {
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;
  return x;
}
__host__ __device__ __forceinline__ uint32_t rng_key(uint64_t seed, uint64_t stream) {
  uint32_t k = pcg32((uint32_t)(stream >> 32) + 0x9E3779B9u);
  k = pcg32((uint32_t)stream ^ k);
  k = pcg32((uint32_t)(seed >> 32) ^ k);
  k = pcg32((uint32_t)seed ^ k);
  return k;
}
struct Philox {   // name kept for call sites; see comment above
  // 128 random bits for group counter `ctr`
  static __device__ __forceinline__ u32x4 gen(uint64_t seed, uint64_t stream, uint64_t ctr) {
    const uint32_t key = rng_key(seed, stream);
    return gen_k(key, ctr);
  }
  static __device__ __forceinline__ u32x4 gen_k(uint32_t key, uint64_t ctr) {
    u32x4 o;
    o[0] = pcg32(((uint32_t)ctr ^ key) + (uint32_t)(ctr >> 32) * 0x85EBCA6Bu);
    o[1] = xorshift32(o[0] ^ 0x68E31DA4u);
    o[2] = xorshift32(o[1]);
    o[3] = xorshift32(o[2]);
    return o;
  }
  // 64 random bits (attention: one group = 1 query x 4 consecutive keys)
  static __device__ __forceinline__ u32x2 gen64_k(uint32_t key, uint64_t ctr) {
    u32x2 o;
    o[0] = pcg32(((uint32_t)ctr ^ key) + (uint32_t)(ctr >> 32) * 0x85EBCA6Bu);
    o[1] = xorshift32(o[0] ^ 0x68E31DA4u);
    return o;
  }
};
// Dropout convention used by every element-wise kernel / epilogue (forward and replay in backward):
// element with linear index e belongs to group G = e >> 3; r = gen(seed, stream, G);
// its 16 random bits are (r[(e&7)>>1] >> (16*(e&1))) & 0xffff; the element is KEPT iff bits >= thr16
// where thr16 = round(p * 65536).  Kept elements are scaled by 65536 / (65536 - thr16).
__device__ __forceinline__ uint32_t drop_bits16(const u32x4& r, int j) {
  return (r[j >> 1] >> (16 * (j & 1))) & 0xffffu;
}

// ---------------------------------------------------------------- misc
// OpenAI tanh approximation, reference mpu/sparse_transformer.py:172-176:
//     gelu(x) = 0.5 x (1 + tanh(u)),  u = 0.79788456 x (1 + 0.044715 x^2)
// evaluated through the identity 0.5 (1 + tanh(u)) = sigmoid(2u) = 1 / (1 + 2^(-2 u log2 e)): one v_exp_f32 and one
// v_rcp_f32 (1 ulp each) instead of libm's tanhf (~40 instructions); absolute error < 2e-7 |x|, far below the
// 16-bit output rounding.  Saturates correctly (exp2 -> inf => 0, exp2 -> 0 => x).  These run inside the GEMM
// epilogues of the h -> 4h layer, where 128 evaluations per lane per tile are not hidden behind MFMA work.
__device__ __forceinline__ float gelu_sigmoid_f(float x, float x2) {
  constexpr float K0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f;      // -2 log2(e) sqrt(2/pi)
  constexpr float K1 = K0 * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(K1, x2, K0));
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_f(float x) {
  return x * gelu_sigmoid_f(x, x * x);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  // d/dx [x s(x)] = s + x s (1 - s) * 2 u'(x),   2 u' = 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2)
  const float x2 = x * x;
  const float s = gelu_sigmoid_f(x, x2);
  constexpr float C0 = 2.0f * 0.7978845608028654f, C1 = C0 * 3.0f * 0.044715f;
  return fmaf(x * fmaf(-s, s, s), fmaf(C1, x2, C0), s);
}

// gelu(x) and gelu'(x) from ONE sigmoid evaluation (the forward epilogue stores gelu' for the backward GEMM, so the
// backward epilogue is a multiply instead of a second exp2 + rcp per element)
__device__ __forceinline__ void gelu_and_grad_f(float x, float& g, float& gd) {
  const float x2 = x * x;
  const float s = gelu_sigmoid_f(x, x2);
  constexpr float C0 = 2.0f * 0.7978845608028654f, C1 = C0 * 3.0f * 0.044715f;
  g = x * s;
  gd = fmaf(x * fmaf(-s, s, s), fmaf(C1, x2, C0), s);
}

// The same for TWO elements on the packed-fp32 VALU forms (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two IEEE operations per
// issue slot, bit-identical to the scalar chain above -- only exp2 and rcp stay per element).  The GEMM epilogue that calls this
// runs with ONE wave per SIMD (gemm_w4_kernel), i.e. it is bound by instruction ISSUE (one slot per ~4 cycles per wave), not by
// the VALU's width: left to the SLP vectoriser only the tail of the chain was packed (x^2, the polynomial and the argument were
// 24 scalar issues per 8 elements).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_and_grad_f2(f32x2_t x, f32x2_t& g, f32x2_t& gd) {
  constexpr float K0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, K1 = K0 * 0.044715f;
  constexpr float C0 = 2.0f * 0.7978845608028654f, C1 = C0 * 3.0f * 0.044715f;
  const f32x2_t x2 = x * x;
  const f32x2_t arg = x * __builtin_elementwise_fma(f32x2_t{K1, K1}, x2, f32x2_t{K0, K0});
  const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} + f32x2_t{1.0f, 1.0f};
  const f32x2_t s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  g = x * s;
  gd = __builtin_elementwise_fma(x * __builtin_elementwise_fma(-s, s, s), __builtin_elementwise_fma(f32x2_t{C1, C1}, x2, f32x2_t{C0, C0}), s);
}
__device__ __forceinline__ f32x2_t gelu_f2(f32x2_t x) {
  constexpr float K0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, K1 = K0 * 0.044715f;
  const f32x2_t arg = x * __builtin_elementwise_fma(f32x2_t{K1, K1}, x * x, f32x2_t{K0, K0});
  const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} + f32x2_t{1.0f, 1.0f};
  return x * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

static inline int cogv_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? COGV_OK : COGV_ERR_LAUNCH;
}
