// Vocab-parallel cross entropy and the fused mixed-precision optimizer step (gfx950).  All HBM-bound.
#include "common.cuh"
#include "cogview_hip.h"

namespace {

// ---------------------------------------------------------------------------------- loads of 8 logits
template <typename LT> struct LogitIO;
template <> struct LogitIO<float> {
  static __device__ __forceinline__ void load8(const float* p, float* v) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
  static __device__ __forceinline__ void store8(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};
template <> struct LogitIO<f16_t> {
  static __device__ __forceinline__ void load8(const f16_t* p, float* v) { unpack8<f16_t>(*reinterpret_cast<const u32x4*>(p), v); }
  static __device__ __forceinline__ void store8(f16_t* p, const float* v) { *reinterpret_cast<u32x4*>(p) = pack8<f16_t>(v); }
};
template <> struct LogitIO<bf16_t> {
  static __device__ __forceinline__ void load8(const bf16_t* p, float* v) { unpack8<bf16_t>(*reinterpret_cast<const u32x4*>(p), v); }
  static __device__ __forceinline__ void store8(bf16_t* p, const float* v) { *reinterpret_cast<u32x4*>(p) = pack8<bf16_t>(v); }
};

// one 256-thread block per row; online (max, sum) per thread, then block combine
template <typename LT>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const LT* logits, const int64_t* target, int64_t vocab_start,
                                                    int v_local, float* rowmax, float* sumexp, float* predicted,
                                                    float* loss) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const LT* L = logits + (size_t)row * v_local;
  float m = -INFINITY, s = 0.f;
  const int nvec = v_local >> 3;
  for (int i = threadIdx.x; i < nvec; i += 256) {
    float v[8]; LogitIO<LT>::load8(L + i * 8, v);
    float bm = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) bm = fmaxf(bm, v[k]);
    if (bm > m) { s *= __expf(m - bm); m = bm; }
#pragma unroll
    for (int k = 0; k < 8; ++k) s += __expf(v[k] - m);
  }
  const float gm = block_max(m, red);
  const float sc = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
  const float gs = block_sum(sc, red);
  if (threadIdx.x == 0) {
    const int64_t t = target[row] - vocab_start;
    const bool in = (t >= 0 && t < v_local);
    const float pl = in ? (float)L[t] : 0.f;
    rowmax[row] = gm; sumexp[row] = gs; predicted[row] = pl;
    if (loss) loss[row] = logf(gs) + gm - pl;
  }
}

template <typename LT>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const LT* logits, const int64_t* target, int64_t vocab_start,
                                                    int v_local, const float* gmax, const float* gsum,
                                                    const float* grad, LT* dlogits) {
  const int row = blockIdx.x;
  const LT* L = logits + (size_t)row * v_local;
  LT* D = dlogits + (size_t)row * v_local;
  const float gm = gmax[row], inv = 1.0f / gsum[row], g = grad[row];
  const int64_t t = target[row] - vocab_start;
  const int nvec = v_local >> 3;
  for (int i = threadIdx.x; i < nvec; i += 256) {
    float v[8]; LogitIO<LT>::load8(L + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float sm = __expf(v[k] - gm) * inv;
      if ((int64_t)(i * 8 + k) == t) sm -= 1.0f;
      v[k] = sm * g;
    }
    LogitIO<LT>::store8(D + i * 8, v);
  }
}

// ---------------------------------------------------------------------------------- grad statistics
// eight consecutive gradients as floats: one 16-byte load of a 16-bit buffer, two of an fp32 buffer (the master gradients of
// FP16_Optimizer's generic path and of fp32 models, mpu/grads.py:62-84)
template <typename T> struct GradIO {
  static __device__ __forceinline__ void load8(const T* p, float* v) { unpack8<T>(*reinterpret_cast<const u32x4*>(p), v); }
  static __device__ __forceinline__ float load1(const T* p) { return HT<T>::to_f(*p); }
};
template <> struct GradIO<float> {
  static __device__ __forceinline__ void load8(const float* p, float* v) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
  }
  static __device__ __forceinline__ float load1(const float* p) { return *p; }
};
template <typename T>
__global__ __launch_bounds__(256) void grad_stats_kernel(const T* grads, const int64_t* chunk_start,
                                                        const int32_t* chunk_len, const uint8_t* chunk_norm,
                                                        int nchunks, double* stats, double* partial) {
  __shared__ float red[16];
  float sq = 0.f; bool bad = false;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const T* g = grads + chunk_start[c];
    const int len = chunk_len[c];
    const bool counted = chunk_norm[c] != 0;
    const int nvec = len >> 3;
    float csq = 0.f;
    for (int i = threadIdx.x; i < nvec; i += 256) {
      float v[8]; GradIO<T>::load8(g + i * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) { csq += v[k] * v[k]; if (!(fabsf(v[k]) <= 3.0e38f)) bad = true; }
    }
    for (int i = (nvec << 3) + threadIdx.x; i < len; i += 256) {
      const float v = GradIO<T>::load1(g + i); csq += v * v; if (!(fabsf(v) <= 3.0e38f)) bad = true;
    }
    if (counted) sq += csq;
  }
  const float bs = block_sum(sq, red);
  const bool any_bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    // per-workgroup partial, summed in workgroup order by grad_stats_final_kernel: the global norm (and with it the clip
    // coefficient of the fused AdamW) is the same bits run after run.  (Rounds 1-2 used atomicAdd on the double: the last
    // floating-point atomic of the training step.)
    partial[blockIdx.x] = (bs == bs) ? (double)bs : 0.0;
    if (any_bad || bs != bs) stats[1] = 1.0;
  }
}
// stats[0] += sum of the workgroups' partials in a fixed order (one wave: lane i takes partials i, i + 64, ...; DPP-free
// butterfly over the 64 lane sums in double)
__global__ __launch_bounds__(64) void grad_stats_final_kernel(const double* partial, int n, double* stats) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) stats[0] += s;
}

// ---------------------------------------------------------------------------------- fused AdamW step
struct AdamArgs {
  void* params; const void* grads; float* master; float* m; float* v;
  const int64_t* chunk_start; const int32_t* chunk_len; const uint8_t* chunk_group; int nchunks;
  float lr[8]; float wd[8];
  float beta1, beta2, omb1, omb2, eps, bc1, bc2; int adam_w_mode;
  float inv_scale, max_norm;
  const double* stats; const double* sumsq_override;
};
template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs p) {
  if (p.stats && p.stats[1] != 0.0) return;                       // overflow: skip the whole step
  float gscale = p.inv_scale;
  if (p.max_norm > 0.f && p.stats) {
    const double ss = p.sumsq_override ? *p.sumsq_override : p.stats[0];
    const float norm = sqrtf((float)ss) * p.inv_scale;             // norm of the UNscaled gradients
    const float coef = p.max_norm / (norm + 1.0e-6f);              // mpu/grads.py:69-72
    if (coef < 1.f) gscale *= coef;
  }
  const float inv_bc1 = 1.0f / p.bc1, inv_sqrt_bc2 = 1.0f / sqrtf(p.bc2);
  T* P = reinterpret_cast<T*>(p.params);
  const T* G = reinterpret_cast<const T*>(p.grads);
  for (int c = blockIdx.x; c < p.nchunks; c += gridDim.x) {
    const int64_t base = p.chunk_start[c];
    const int len = p.chunk_len[c];
    const int grp = p.chunk_group[c] & 7;
    const float lr = p.lr[grp], wd = p.wd[grp];
    const int nvec = len >> 3;
    for (int i = threadIdx.x; i < nvec + ((len & 7) ? 1 : 0); i += 256) {
      const int64_t o = base + (int64_t)i * 8;
      const int cnt = (i < nvec) ? 8 : (len & 7);
      float g[8], w[8], m[8], v[8];
      if (cnt == 8) {
        unpack8<T>(*reinterpret_cast<const u32x4*>(G + o), g);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.master + o), w1 = *reinterpret_cast<const f32x4*>(p.master + o + 4);
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(p.m + o), m1 = *reinterpret_cast<const f32x4*>(p.m + o + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p.v + o), v1 = *reinterpret_cast<const f32x4*>(p.v + o + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { w[k] = w0[k]; w[k + 4] = w1[k]; m[k] = m0[k]; m[k + 4] = m1[k]; v[k] = v0[k]; v[k + 4] = v1[k]; }
      } else {
        for (int k = 0; k < 8; ++k) {
          const bool ok = k < cnt;
          g[k] = ok ? HT<T>::to_f(G[o + k]) : 0.f; w[k] = ok ? p.master[o + k] : 0.f;
          m[k] = ok ? p.m[o + k] : 0.f; v[k] = ok ? p.v[o + k] : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float gr = g[k] * gscale;
        if (!p.adam_w_mode) gr += wd * w[k];                       // L2 mode
        m[k] = p.beta1 * m[k] + p.omb1 * gr;
        v[k] = p.beta2 * v[k] + p.omb2 * gr * gr;
        const float denom = sqrtf(v[k]) * inv_sqrt_bc2 + p.eps;
        float upd = (m[k] * inv_bc1) / denom;
        if (p.adam_w_mode) upd += wd * w[k];                       // decoupled weight decay (apex adam_w_mode=1)
        w[k] -= lr * upd;
      }
      if (cnt == 8) {
        *reinterpret_cast<f32x4*>(p.master + o) = f32x4{w[0], w[1], w[2], w[3]};
        *reinterpret_cast<f32x4*>(p.master + o + 4) = f32x4{w[4], w[5], w[6], w[7]};
        *reinterpret_cast<f32x4*>(p.m + o) = f32x4{m[0], m[1], m[2], m[3]};
        *reinterpret_cast<f32x4*>(p.m + o + 4) = f32x4{m[4], m[5], m[6], m[7]};
        *reinterpret_cast<f32x4*>(p.v + o) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p.v + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        *reinterpret_cast<u32x4*>(P + o) = pack8<T>(w);
      } else {
        for (int k = 0; k < cnt; ++k) { p.master[o + k] = w[k]; p.m[o + k] = m[k]; p.v[o + k] = v[k]; P[o + k] = HT<T>::from_f(w[k]); }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_up_kernel(const T* src, float* dst, size_t n) {
  const size_t nvec = n >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float v[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(src + i * 8), v);
    *reinterpret_cast<f32x4*>(dst + i * 8) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + i * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
  if (blockIdx.x == 0) for (size_t j = (nvec << 3) + threadIdx.x; j < n; j += 256) dst[j] = HT<T>::to_f(src[j]);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_down_kernel(const float* src, T* dst, size_t n) {
  const size_t nvec = n >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i * 8), b = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *reinterpret_cast<u32x4*>(dst + i * 8) = pack8<T>(v);
  }
  if (blockIdx.x == 0) for (size_t j = (nvec << 3) + threadIdx.x; j < n; j += 256) dst[j] = HT<T>::from_f(src[j]);
}

inline int grid_for(size_t nvec) { size_t b = (nvec + 255) / 256; if (b > 2048) b = 2048; if (b < 1) b = 1; return (int)b; }

}  // namespace

extern "C" int cogv_ce_fwd(int logits_dtype, const void* logits, const int64_t* target, int64_t vocab_start, int rows,
                           int v_local, float* rowmax, float* sumexp, float* predicted, float* loss, void* stream) {
  if (rows <= 0 || v_local <= 0 || (v_local & 7)) return COGV_ERR_ARG;
  if (!logits || !target || !rowmax || !sumexp || !predicted || ((uintptr_t)logits & 15)) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (logits_dtype) {
    case COGV_F32: hipLaunchKernelGGL((ce_fwd_kernel<float>), dim3(rows), dim3(256), 0, st, (const float*)logits, target, vocab_start, v_local, rowmax, sumexp, predicted, loss); break;
    case COGV_F16: hipLaunchKernelGGL((ce_fwd_kernel<f16_t>), dim3(rows), dim3(256), 0, st, (const f16_t*)logits, target, vocab_start, v_local, rowmax, sumexp, predicted, loss); break;
    case COGV_BF16: hipLaunchKernelGGL((ce_fwd_kernel<bf16_t>), dim3(rows), dim3(256), 0, st, (const bf16_t*)logits, target, vocab_start, v_local, rowmax, sumexp, predicted, loss); break;
    default: return COGV_ERR_UNSUPPORTED;
  }
  return cogv_check_launch();
}

extern "C" int cogv_ce_bwd(int logits_dtype, const void* logits, const int64_t* target, int64_t vocab_start, int rows,
                           int v_local, const float* gmax, const float* gsum, const float* grad, void* dlogits,
                           void* stream) {
  if (rows <= 0 || v_local <= 0 || (v_local & 7)) return COGV_ERR_ARG;
  if (!logits || !target || !gmax || !gsum || !grad || !dlogits) return COGV_ERR_ARG;
  if (((uintptr_t)logits | (uintptr_t)dlogits) & 15) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (logits_dtype) {
    case COGV_F32: hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3(rows), dim3(256), 0, st, (const float*)logits, target, vocab_start, v_local, gmax, gsum, grad, (float*)dlogits); break;
    case COGV_F16: hipLaunchKernelGGL((ce_bwd_kernel<f16_t>), dim3(rows), dim3(256), 0, st, (const f16_t*)logits, target, vocab_start, v_local, gmax, gsum, grad, (f16_t*)dlogits); break;
    case COGV_BF16: hipLaunchKernelGGL((ce_bwd_kernel<bf16_t>), dim3(rows), dim3(256), 0, st, (const bf16_t*)logits, target, vocab_start, v_local, gmax, gsum, grad, (bf16_t*)dlogits); break;
    default: return COGV_ERR_UNSUPPORTED;
  }
  return cogv_check_launch();
}

extern "C" size_t cogv_grad_stats_workspace_bytes(void) { return 2048 * sizeof(double); }

extern "C" int cogv_grad_stats(int dtype, const void* grads, const int64_t* chunk_start, const int32_t* chunk_len,
                               const uint8_t* chunk_norm, int nchunks, double* stats, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16 && dtype != COGV_F32) return COGV_ERR_UNSUPPORTED;
  if (!grads || !chunk_start || !chunk_len || !chunk_norm || nchunks <= 0 || !stats) return COGV_ERR_ARG;
  if ((uintptr_t)grads & 15) return COGV_ERR_ARG;
  // per-workgroup partial sums live in the CALLER's workspace (round 3 kept them in a lazily allocated static buffer: a race
  // between two streams / host threads, and a hipMalloc inside a graph capture)
  if (!workspace || ((uintptr_t)workspace & 7) || workspace_bytes < cogv_grad_stats_workspace_bytes()) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = nchunks < 2048 ? nchunks : 2048;
  double* partial = reinterpret_cast<double*>(workspace);
  if (dtype == COGV_F16) hipLaunchKernelGGL((grad_stats_kernel<f16_t>), dim3(g), dim3(256), 0, st, (const f16_t*)grads, chunk_start, chunk_len, chunk_norm, nchunks, stats, partial);
  else if (dtype == COGV_BF16) hipLaunchKernelGGL((grad_stats_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)grads, chunk_start, chunk_len, chunk_norm, nchunks, stats, partial);
  else hipLaunchKernelGGL((grad_stats_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)grads, chunk_start, chunk_len, chunk_norm, nchunks, stats, partial);
  hipLaunchKernelGGL(grad_stats_final_kernel, dim3(1), dim3(64), 0, st, partial, g, stats);
  return cogv_check_launch();
}

extern "C" int cogv_adamw_step(const cogv_adam_desc* d, void* stream) {
  if (!d) return COGV_ERR_ARG;
  if (d->dtype != COGV_F16 && d->dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (!d->params || !d->grads || !d->master || !d->exp_avg || !d->exp_avg_sq) return COGV_ERR_ARG;
  if (!d->chunk_start || !d->chunk_len || !d->chunk_group || d->nchunks <= 0 || d->step < 1) return COGV_ERR_ARG;
  if (((uintptr_t)d->params | (uintptr_t)d->grads | (uintptr_t)d->master | (uintptr_t)d->exp_avg | (uintptr_t)d->exp_avg_sq) & 15) return COGV_ERR_ARG;
  AdamArgs a;
  a.params = d->params; a.grads = d->grads; a.master = d->master; a.m = d->exp_avg; a.v = d->exp_avg_sq;
  a.chunk_start = d->chunk_start; a.chunk_len = d->chunk_len; a.chunk_group = d->chunk_group; a.nchunks = d->nchunks;
  for (int i = 0; i < 8; ++i) { a.lr[i] = d->lr[i]; a.wd[i] = d->weight_decay[i]; }
  a.beta1 = d->beta1; a.beta2 = d->beta2; a.eps = d->eps; a.adam_w_mode = d->adam_w_mode;
  // 1-beta evaluated in double then rounded once (what torch.optim.AdamW / the oracle do)
  a.omb1 = (float)(1.0 - (double)d->beta1_d); a.omb2 = (float)(1.0 - (double)d->beta2_d);
  if (d->bias_correction) {
    a.bc1 = (float)(1.0 - pow(d->beta1_d, (double)d->step));
    a.bc2 = (float)(1.0 - pow(d->beta2_d, (double)d->step));
  } else { a.bc1 = 1.f; a.bc2 = 1.f; }
  a.inv_scale = d->inv_loss_scale; a.max_norm = d->max_grad_norm;
  a.stats = d->stats; a.sumsq_override = d->norm_sumsq_override;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = d->nchunks < 4096 ? d->nchunks : 4096;
  if (d->dtype == COGV_F16) hipLaunchKernelGGL((adamw_kernel<f16_t>), dim3(g), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((adamw_kernel<bf16_t>), dim3(g), dim3(256), 0, st, a);
  return cogv_check_launch();
}

extern "C" int cogv_cast_flat(int dtype, const void* src_half, float* dst_f32, size_t n, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (!src_half || !dst_f32 || n == 0 || (((uintptr_t)src_half | (uintptr_t)dst_f32) & 15)) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COGV_F16) hipLaunchKernelGGL((cast_up_kernel<f16_t>), dim3(grid_for(n >> 3)), dim3(256), 0, st, (const f16_t*)src_half, dst_f32, n);
  else hipLaunchKernelGGL((cast_up_kernel<bf16_t>), dim3(grid_for(n >> 3)), dim3(256), 0, st, (const bf16_t*)src_half, dst_f32, n);
  return cogv_check_launch();
}
extern "C" int cogv_cast_flat_back(int dtype, const float* src_f32, void* dst_half, size_t n, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (!src_f32 || !dst_half || n == 0 || (((uintptr_t)src_f32 | (uintptr_t)dst_half) & 15)) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COGV_F16) hipLaunchKernelGGL((cast_down_kernel<f16_t>), dim3(grid_for(n >> 3)), dim3(256), 0, st, src_f32, (f16_t*)dst_half, n);
  else hipLaunchKernelGGL((cast_down_kernel<bf16_t>), dim3(grid_for(n >> 3)), dim3(256), 0, st, src_f32, (bf16_t*)dst_half, n);
  return cogv_check_launch();
}

extern "C" int cogv_version(void) { return 1; }
extern "C" const char* cogv_arch(void) { return "gfx950"; }
