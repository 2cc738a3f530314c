// HBM-bound element-wise / gather / reduction kernels of the CogView GPT hot path (gfx950).
// All tensors are accessed as 16-byte vectors (8 x fp16/bf16 per lane), grid-stride, <= 2048 blocks.
#include "common.cuh"
#include "cogview_hip.h"

namespace {

inline int grid_for(size_t nvec) {
  size_t b = (nvec + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------ embedding (+pos, +dropout, +abs-max)
// reference: VocabParallelEmbedding.forward mpu/layers.py:117-133 (shard-masked gather) and
// GPT2ParallelTransformer.forward mpu/sparse_transformer.py:522-524 (position add + dropout)
struct EmbArgs {
  const int64_t* ids; const void* table; int64_t vocab_start, vocab_end;
  const void* x_in;            // used when ids == NULL
  const int64_t* pos_ids; const void* pos_table; int64_t n_pos;
  void* out; float* absmax_out;
  int64_t n_tok; int h;
  uint64_t seed, stream_id; uint32_t thr16; float keep_scale;
};
// OUT32: `out` is the fp32 residual stream -- word + position are added in fp32 (no rounding of the sum), the abs-max
// is taken over the fp32 values.  Otherwise everything is T and the sum is rounded once (a T + T add in the reference).
template <typename T, bool OUT32>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const EmbArgs p) {
  __shared__ float red[16];
  const int hv = p.h >> 3;
  const size_t nvec = (size_t)p.n_tok * hv;
  float amax = 0.f; bool nan = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const int64_t tok = (int64_t)(i / hv);
    const int col = (int)(i % hv) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.ids) {
      const int64_t id = p.ids[tok];
      if (id >= p.vocab_start && id < p.vocab_end)
        unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.table) + (size_t)(id - p.vocab_start) * p.h + col), v);
    } else {
      unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.x_in) + (size_t)tok * p.h + col), v);
    }
    if (p.pos_table) {
      int64_t pid = p.pos_ids[tok];
      pid = pid < 0 ? 0 : (pid >= p.n_pos ? p.n_pos - 1 : pid);
      float pe[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.pos_table) + (size_t)pid * p.h + col), pe);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += pe[k];
      if (!OUT32) { const u32x4 s = pack8<T>(v); unpack8<T>(s, v); }     // a T + T add in the reference: round once
    }
    if (p.thr16) {
      const uint64_t e = (uint64_t)tok * (uint64_t)p.h + (uint64_t)col;
      const u32x4 r = Philox::gen(p.seed, p.stream_id, e >> 3);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (drop_bits16(r, k) >= p.thr16) ? v[k] * p.keep_scale : 0.f;
    }
    (void)Row8<T, OUT32>::st(p.out, (size_t)tok * p.h + col, v, 0u);      // v <- the stored values
    if (p.absmax_out) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { if (v[k] != v[k]) nan = true; else amax = fmaxf(amax, fabsf(v[k])); }
    }
  }
  if (p.absmax_out) {
    const float bm = block_max(amax, red);
    const bool any_nan = __syncthreads_or(nan);
    if (threadIdx.x == 0) atomic_max_nonneg(p.absmax_out, any_nan ? __uint_as_float(0x7fc00000u) : bm);
  }
}

// backward: d(table)[id] += sum over the tokens with that id of mask(dout[tok]);  same for the position table.
// DETERMINISTIC and rounded ONCE (the reference's embedding_dense_backward accumulates in fp32 and rounds once; the
// round-1/2 kernel scatter-added with packed 16-bit atomics: order-dependent, one rounding per add -- 24 per position
// row at b = 24).  No atomics on floating-point data:
//   pass 1 (emb_key_stats_kernel): per key, the number of tokens carrying it and its FIRST token (integer atomics on a
//          zeroed 2 x int32 table per key: order-independent results);
//   pass 2 (emb_segsum_kernel): one workgroup per (token, 2048-column slab).  Only a key's first token works: it sums
//          the masked gradient rows of all tokens with the key in ASCENDING token order in fp32 (the other occurrences
//          are found by scanning the ids behind it 256 at a time, wave ballots give the order) and adds the sum to the
//          table row with one rounding.  Keys that occur once (most word ids) skip the scan.
struct EmbBwdArgs {
  const void* dout; void* dx; int64_t n_tok; int h;
  uint64_t seed, stream_id; uint32_t thr16; float keep_scale;
};
template <typename T, bool D32>        // D32: dout (and dx) are the fp32 gradient of the residual stream
__device__ __forceinline__ void emb_masked_row(const EmbBwdArgs& p, int64_t tok, int col, float* v) {
  Row8<T, D32>::to_f(Row8<T, D32>::ld(p.dout, (size_t)tok * p.h + col), v);
  if (p.thr16) {
    const uint64_t e = (uint64_t)tok * (uint64_t)p.h + (uint64_t)col;
    const u32x4 r = Philox::gen(p.seed, p.stream_id, e >> 3);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (drop_bits16(r, k) >= p.thr16) ? v[k] * p.keep_scale : 0.f;
  }
}
// dx = mask(dout) (only the tests ask for it)
template <typename T, bool D32>
__global__ __launch_bounds__(256) void embedding_dx_kernel(const EmbBwdArgs p) {
  const int hv = p.h >> 3;
  const size_t nvec = (size_t)p.n_tok * hv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const int64_t tok = (int64_t)(i / hv);
    const int col = (int)(i % hv) * 8;
    float v[8];
    emb_masked_row<T, D32>(p, tok, col, v);
    (void)Row8<T, D32>::st(p.dx, (size_t)tok * p.h + col, v, 0u);
  }
}

struct EmbKeys {           // word ids: keys outside [lo, hi) take no part; position ids: clamped into [0, hi)
  const int64_t* keys; int64_t lo, hi; int clamp;
  int32_t* stats;          // [hi - lo][2]: {count, n_tok - first token} (zeroed before pass 1)
};
__device__ __forceinline__ int64_t emb_key(const EmbKeys& k, int64_t tok) {     // -1: not in this table
  int64_t v = k.keys[tok];
  if (k.clamp) return v < 0 ? 0 : (v >= k.hi ? k.hi - 1 : v);
  return (v >= k.lo && v < k.hi) ? v - k.lo : -1;
}
__global__ __launch_bounds__(256) void emb_key_stats_kernel(const EmbKeys k, int64_t n_tok) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n_tok; t += (int64_t)gridDim.x * 256) {
    const int64_t key = emb_key(k, t);
    if (key < 0) continue;
    atomicAdd(&k.stats[2 * key], 1);
    atomicMax(&k.stats[2 * key + 1], (int32_t)(n_tok - t));
  }
}
template <typename T, bool D32>
__global__ __launch_bounds__(256) void emb_segsum_kernel(const EmbBwdArgs p, const EmbKeys k, void* dtable) {
  __shared__ unsigned long long masks[4];
  const int64_t t = blockIdx.x;
  const int64_t key = emb_key(k, t);
  if (key < 0) return;
  if ((int64_t)k.stats[2 * key + 1] != p.n_tok - t) return;          // not the first token of its key
  int remaining = k.stats[2 * key] - 1;
  const int hv = p.h >> 3;
  const int cv = blockIdx.y * 256 + threadIdx.x;
  const bool active = cv < hv;
  const int col = cv * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (active) emb_masked_row<T, D32>(p, t, col, acc);
  const int wave = threadIdx.x >> 6;
  for (int64_t w0 = t + 1; remaining > 0 && w0 < p.n_tok; w0 += 256) {
    const int64_t j = w0 + threadIdx.x;
    const bool m = j < p.n_tok && emb_key(k, j) == key;
    const unsigned long long bal = __ballot(m);
    if ((threadIdx.x & 63) == 0) masks[wave] = bal;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      unsigned long long mm = masks[w];
      while (mm) {
        const int bit = __builtin_ctzll(mm);
        mm &= mm - 1;
        --remaining;
        if (active) {
          float v[8];
          emb_masked_row<T, D32>(p, w0 + w * 64 + bit, col, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
      }
    }
    __syncthreads();
  }
  if (active) {
    u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<T*>(dtable) + (size_t)key * p.h + col);
    float old[8];
    unpack8<T>(*dst, old);
#pragma unroll
    for (int e = 0; e < 8; ++e) old[e] += acc[e];
    *dst = pack8<T>(old);
  }
}

// ------------------------------------------------------------------ generic unary/binary element-wise
enum { OP_GELU_FWD = 0, OP_GELU_BWD = 1, OP_DROPOUT = 2, OP_ADD = 3, OP_SCALE = 4 };
struct EwArgs {
  const void* a; const void* b; void* out; size_t n; float scale; float* absmax_out;
  uint64_t seed, stream_id; uint32_t thr16; float keep_scale;
};
template <typename T, int OP>
__global__ __launch_bounds__(256) void ew_kernel(const EwArgs p) {
  __shared__ float red[16];
  const size_t nvec = p.n >> 3;
  float amax = 0.f; bool nan = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float a[8], b[8], o[8];
    unpack8<T>(reinterpret_cast<const u32x4*>(p.a)[i], a);
    if (OP == OP_GELU_BWD || OP == OP_ADD) unpack8<T>(reinterpret_cast<const u32x4*>(p.b)[i], b);
    u32x4 r;
    if (OP == OP_DROPOUT) r = Philox::gen(p.seed, p.stream_id, i);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (OP == OP_GELU_FWD) o[k] = gelu_f(a[k]);
      else if (OP == OP_GELU_BWD) o[k] = a[k] * gelu_grad_f(b[k]);      // a = dy, b = x
      else if (OP == OP_DROPOUT) o[k] = (drop_bits16(r, k) >= p.thr16) ? a[k] * p.keep_scale : 0.f;
      else if (OP == OP_ADD) o[k] = a[k] + b[k];
      else o[k] = a[k] * p.scale;
    }
    const u32x4 ov = pack8<T>(o);
    reinterpret_cast<u32x4*>(p.out)[i] = ov;
    if (p.absmax_out) {
      float rr[8]; unpack8<T>(ov, rr);
#pragma unroll
      for (int k = 0; k < 8; ++k) { if (rr[k] != rr[k]) nan = true; else amax = fmaxf(amax, fabsf(rr[k])); }
    }
  }
  if (p.absmax_out) {
    const float bm = block_max(amax, red);
    const bool any_nan = __syncthreads_or(nan);
    if (threadIdx.x == 0) atomic_max_nonneg(p.absmax_out, any_nan ? __uint_as_float(0x7fc00000u) : bm);
  }
}

// ------------------------------------------------------------------ abs-max of a whole tensor
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const void* x, size_t n, float* out) {
  __shared__ float red[16];
  const size_t nvec = n >> 3;
  float amax = 0.f; bool nan = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float a[8]; unpack8<T>(reinterpret_cast<const u32x4*>(x)[i], a);
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (a[k] != a[k]) nan = true; else amax = fmaxf(amax, fabsf(a[k])); }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // scalar tail (n % 8)
    const T* t = reinterpret_cast<const T*>(x);
    for (size_t j = nvec << 3; j < n; ++j) { const float a = HT<T>::to_f(t[j]); if (a != a) nan = true; else amax = fmaxf(amax, fabsf(a)); }
  }
  const float bm = block_max(amax, red);
  const bool any_nan = __syncthreads_or(nan);
  if (threadIdx.x == 0) atomic_max_nonneg(out, any_nan ? __uint_as_float(0x7fc00000u) : bm);
}

// abs-max of an fp32 tensor (the residual stream when no producer epilogue published it); n % 4 == 0
__global__ __launch_bounds__(256) void absmax_f32_kernel(const float* x, size_t n, float* out) {
  __shared__ uint32_t red[16];
  const size_t nvec = n >> 2;
  uint32_t m = 0u;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) m = max(m, v[k] & 0x7fffffffu);
  }
  const float bm = absmax_f32_block(m, red);
  if (threadIdx.x == 0) atomic_max_nonneg(out, bm);
}

// out(fp32) = a(fp32) + b(T): a branch output joins the fp32 residual stream (op-by-op composition of the layer)
template <typename T>
__global__ __launch_bounds__(256) void add_stream_kernel(const float* a, const void* b, float* out, size_t n, float* absmax_out) {
  __shared__ uint32_t red[16];
  const size_t nvec = n >> 3;
  uint32_t m = 0u;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float x[8], y[8];
    Row8<T, true>::to_f(Row8<T, true>::ld(a, i * 8), x);
    unpack8<T>(reinterpret_cast<const u32x4*>(b)[i], y);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] += y[k];
    m = Row8<T, true>::st(out, i * 8, x, m);
  }
  if (absmax_out) {
    const float bm = absmax_f32_block(m, red);
    if (threadIdx.x == 0) atomic_max_nonneg(absmax_out, bm);
  }
}

// ------------------------------------------------------------------ column sum  db[n] = sum_m dY[m][n]
// stage 1: block (bx, by) sums rows [by*rows_per, ...) of an 8*64-column slab into partial[by][N]
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* dy, int M, int N, int ld, int rows_per,
                                                            float* partial) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 8;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    for (int r = r0 + w; r < r1; r += 4) {
      float a[8]; unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(dy) + (size_t)r * ld + col), a);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += a[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[w][lane * 8 + k] = s[k];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    const int gc = blockIdx.x * 512 + c;
    if (gc < N) partial[(size_t)blockIdx.y * N + gc] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
  }
}
// out[c] (+)= sum_b partial[b][c]: 64 columns x 4 row slices per workgroup (a thread-per-column loop over
// hundreds of partial rows took 75 us at N = 10240: only 40 workgroups, each a serial chain of dependent loads)
template <typename T>
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int nslab, int N, void* out, int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // independent sums: all loads of a thread in flight together
  if (c < N) {
    int b = part;
    for (; b + 7 * 4 < nslab; b += 8 * 4) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += partial[(size_t)(b + 4 * u) * N + c];
    }
    for (; b < nslab; b += 4) t[0] += partial[(size_t)b * N + c];
  }
  red[part][cl] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  __syncthreads();
  if (part == 0 && c < N) {
    float s = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    T* o = reinterpret_cast<T*>(out);
    if (accumulate) s += HT<T>::to_f(o[c]);
    o[c] = HT<T>::from_f(s);
  }
}

template <typename T, int OP>
int launch_ew(const EwArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((ew_kernel<T, OP>), dim3(grid_for(a.n >> 3)), dim3(256), 0, st, a);
  return cogv_check_launch();
}
template <int OP>
int dispatch_ew(int dtype, const void* a, const void* b, void* out, size_t n, float scale, float* absmax_out,
                float p, uint64_t seed, uint64_t stream_id, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if ((n & 7) || !a || !out) return COGV_ERR_ARG;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return COGV_ERR_ARG;
  EwArgs e{a, b, out, n, scale, absmax_out, seed, stream_id, 0u, 1.f};
  if (OP == OP_DROPOUT) {
    if (!(p >= 0.f && p < 1.f)) return COGV_ERR_ARG;
    e.thr16 = (uint32_t)(p * 65536.0f + 0.5f);
    e.keep_scale = 65536.0f / (65536.0f - (float)e.thr16);
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return dtype == COGV_F16 ? launch_ew<f16_t, OP>(e, st) : launch_ew<bf16_t, OP>(e, st);
}

}  // namespace

extern "C" int cogv_embedding_fwd(int dtype, const int64_t* ids, const void* table, int64_t vocab_start,
                                  int64_t vocab_end, const void* x_in, const int64_t* pos_ids, const void* pos_table,
                                  int64_t n_pos, void* out, float* absmax_out, int64_t n_tok, int h, float dropout_p,
                                  uint64_t seed, uint64_t stream_id, int out_f32, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (n_tok <= 0 || h <= 0 || (h & 7) || !out) return COGV_ERR_ARG;
  if (!ids && !x_in) return COGV_ERR_ARG;
  if (ids && !table) return COGV_ERR_ARG;
  if (pos_table && (!pos_ids || n_pos <= 0)) return COGV_ERR_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return COGV_ERR_ARG;
  if (((uintptr_t)table | (uintptr_t)x_in | (uintptr_t)pos_table | (uintptr_t)out) & 15) return COGV_ERR_ARG;
  EmbArgs a;
  a.ids = ids; a.table = table; a.vocab_start = vocab_start; a.vocab_end = vocab_end; a.x_in = x_in;
  a.pos_ids = pos_ids; a.pos_table = pos_table; a.n_pos = n_pos; a.out = out; a.absmax_out = absmax_out;
  a.n_tok = n_tok; a.h = h; a.seed = seed; a.stream_id = stream_id;
  a.thr16 = (uint32_t)(dropout_p * 65536.0f + 0.5f);
  a.keep_scale = 65536.0f / (65536.0f - (float)a.thr16);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for((size_t)n_tok * (h >> 3));
  if (out_f32) {
    if (dtype == COGV_F16) hipLaunchKernelGGL((embedding_fwd_kernel<f16_t, true>), dim3(g), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((embedding_fwd_kernel<bf16_t, true>), dim3(g), dim3(256), 0, st, a);
  } else {
    if (dtype == COGV_F16) hipLaunchKernelGGL((embedding_fwd_kernel<f16_t, false>), dim3(g), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((embedding_fwd_kernel<bf16_t, false>), dim3(g), dim3(256), 0, st, a);
  }
  return cogv_check_launch();
}

extern "C" size_t cogv_embedding_bwd_workspace_bytes(int64_t table_rows, int64_t n_pos) {
  return (size_t)(table_rows > 0 ? table_rows : 0) * 8 + (size_t)(n_pos > 0 ? n_pos : 0) * 8;
}

extern "C" int cogv_embedding_bwd(int dtype, const void* dout, const int64_t* ids, void* dtable, int64_t vocab_start,
                                  int64_t vocab_end, const int64_t* pos_ids, void* dpos, int64_t n_pos, void* dx,
                                  int64_t n_tok, int h, float dropout_p, uint64_t seed, uint64_t stream_id,
                                  void* workspace, size_t workspace_bytes, int dout_f32, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (n_tok <= 0 || n_tok >= (int64_t)0x7fffffff || h <= 0 || (h & 7) || !dout) return COGV_ERR_ARG;
  if (dtable && (!ids || vocab_end <= vocab_start)) return COGV_ERR_ARG;
  if (dpos && (!pos_ids || n_pos <= 0)) return COGV_ERR_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return COGV_ERR_ARG;
  if (((uintptr_t)dout | (uintptr_t)dtable | (uintptr_t)dpos | (uintptr_t)dx | (uintptr_t)workspace) & 15) return COGV_ERR_ARG;
  const int64_t rows = dtable ? vocab_end - vocab_start : 0;
  const size_t need = cogv_embedding_bwd_workspace_bytes(rows, dpos ? n_pos : 0);
  if (need && (!workspace || workspace_bytes < need)) return COGV_ERR_ARG;
  EmbBwdArgs a;
  a.dout = dout; a.dx = dx; a.n_tok = n_tok; a.h = h;
  a.seed = seed; a.stream_id = stream_id;
  a.thr16 = (uint32_t)(dropout_p * 65536.0f + 0.5f);
  a.keep_scale = 65536.0f / (65536.0f - (float)a.thr16);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define EMB_LAUNCH(KERNEL, GRID, ...)                                                                       \
  do {                                                                                                     \
    if (dtype == COGV_F16) {                                                                               \
      if (dout_f32) hipLaunchKernelGGL((KERNEL<f16_t, true>), GRID, dim3(256), 0, st, __VA_ARGS__);        \
      else hipLaunchKernelGGL((KERNEL<f16_t, false>), GRID, dim3(256), 0, st, __VA_ARGS__);                \
    } else {                                                                                               \
      if (dout_f32) hipLaunchKernelGGL((KERNEL<bf16_t, true>), GRID, dim3(256), 0, st, __VA_ARGS__);       \
      else hipLaunchKernelGGL((KERNEL<bf16_t, false>), GRID, dim3(256), 0, st, __VA_ARGS__);               \
    }                                                                                                      \
  } while (0)
  if (dx) {
    const int g = grid_for((size_t)n_tok * (h >> 3));
    EMB_LAUNCH(embedding_dx_kernel, dim3(g), a);
  }
  if (need) {
    if (hipMemsetAsync(workspace, 0, need, st) != hipSuccess) return COGV_ERR_LAUNCH;
    const int gk = (int)((n_tok + 255) / 256 > 1024 ? 1024 : (n_tok + 255) / 256);
    const dim3 gs((unsigned)n_tok, (unsigned)(((h >> 3) + 255) / 256));
    EmbKeys kw{ids, vocab_start, vocab_end, 0, reinterpret_cast<int32_t*>(workspace)};
    EmbKeys kp{pos_ids, 0, n_pos, 1, reinterpret_cast<int32_t*>(workspace) + 2 * rows};
    if (dtable) hipLaunchKernelGGL(emb_key_stats_kernel, dim3(gk), dim3(256), 0, st, kw, n_tok);
    if (dpos) hipLaunchKernelGGL(emb_key_stats_kernel, dim3(gk), dim3(256), 0, st, kp, n_tok);
    if (dtable) EMB_LAUNCH(emb_segsum_kernel, gs, a, kw, dtable);
    if (dpos) EMB_LAUNCH(emb_segsum_kernel, gs, a, kp, dpos);
  }
#undef EMB_LAUNCH
  return cogv_check_launch();
}

extern "C" int cogv_gelu_fwd(int dtype, const void* x, void* y, size_t n, void* stream) {
  return dispatch_ew<OP_GELU_FWD>(dtype, x, nullptr, y, n, 1.f, nullptr, 0.f, 0, 0, stream);
}
extern "C" int cogv_gelu_bwd(int dtype, const void* dy, const void* x, void* dx, size_t n, void* stream) {
  if (!x) return COGV_ERR_ARG;
  return dispatch_ew<OP_GELU_BWD>(dtype, dy, x, dx, n, 1.f, nullptr, 0.f, 0, 0, stream);
}
// forward and backward of dropout are the same map (mask * 1/(1-p)) applied to x resp. dy
extern "C" int cogv_dropout(int dtype, const void* x, void* y, size_t n, float p, uint64_t seed, uint64_t stream_id,
                            float* absmax_out, void* stream) {
  return dispatch_ew<OP_DROPOUT>(dtype, x, nullptr, y, n, 1.f, absmax_out, p, seed, stream_id, stream);
}
extern "C" int cogv_add(int dtype, const void* a, const void* b, void* out, size_t n, float* absmax_out, void* stream) {
  if (!b) return COGV_ERR_ARG;
  return dispatch_ew<OP_ADD>(dtype, a, b, out, n, 1.f, absmax_out, 0.f, 0, 0, stream);
}
extern "C" int cogv_scale(int dtype, const void* x, void* y, size_t n, float scale, void* stream) {
  return dispatch_ew<OP_SCALE>(dtype, x, nullptr, y, n, scale, nullptr, 0.f, 0, 0, stream);
}

extern "C" int cogv_add_stream(int dtype, const float* a, const void* b, float* out, size_t n, float* absmax_out, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if ((n & 7) || !a || !b || !out) return COGV_ERR_ARG;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int g = grid_for(n >> 3);
  if (dtype == COGV_F16) hipLaunchKernelGGL((add_stream_kernel<f16_t>), dim3(g), dim3(256), 0, st, a, b, out, n, absmax_out);
  else hipLaunchKernelGGL((add_stream_kernel<bf16_t>), dim3(g), dim3(256), 0, st, a, b, out, n, absmax_out);
  return cogv_check_launch();
}

extern "C" int cogv_absmax(int dtype, const void* x, size_t n, float* out, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16 && dtype != COGV_F32) return COGV_ERR_UNSUPPORTED;
  if (!x || !out || n == 0 || ((uintptr_t)x & 15)) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COGV_F32) {
    if (n & 3) return COGV_ERR_ARG;
    hipLaunchKernelGGL(absmax_f32_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, st, reinterpret_cast<const float*>(x), n, out);
    return cogv_check_launch();
  }
  const int g = grid_for(n >> 3);
  if (dtype == COGV_F16) hipLaunchKernelGGL((absmax_kernel<f16_t>), dim3(g), dim3(256), 0, st, x, n, out);
  else hipLaunchKernelGGL((absmax_kernel<bf16_t>), dim3(g), dim3(256), 0, st, x, n, out);
  return cogv_check_launch();
}

extern "C" int cogv_colsum_finalize(int dtype, const float* partial, int rows, int N, void* out, int accumulate, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (!partial || !out || rows <= 0 || N <= 0) return COGV_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COGV_F16) hipLaunchKernelGGL((colsum_final_kernel<f16_t>), dim3((N + 63) / 64), dim3(256), 0, st, partial, rows, N, out, accumulate);
  else hipLaunchKernelGGL((colsum_final_kernel<bf16_t>), dim3((N + 63) / 64), dim3(256), 0, st, partial, rows, N, out, accumulate);
  return cogv_check_launch();
}

extern "C" size_t cogv_colsum_workspace_bytes(int M, int N) {
  int nslab = (M + 255) / 256; if (nslab > 64) nslab = 64; if (nslab < 1) nslab = 1;
  return (size_t)nslab * (size_t)N * sizeof(float);
}
extern "C" int cogv_colsum(int dtype, const void* dy, int M, int N, int ld, void* out, int accumulate, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (M <= 0 || N <= 0 || (N & 7) || (ld & 7) || !dy || !out || !workspace) return COGV_ERR_ARG;
  if (workspace_bytes < cogv_colsum_workspace_bytes(M, N)) return COGV_ERR_ARG;
  if ((uintptr_t)dy & 15) return COGV_ERR_ARG;
  int nslab = (M + 255) / 256; if (nslab > 64) nslab = 64; if (nslab < 1) nslab = 1;
  const int rows_per = (M + nslab - 1) / nslab;
  nslab = (M + rows_per - 1) / rows_per;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 g1((N + 511) / 512, nslab);
  float* part = reinterpret_cast<float*>(workspace);
  if (dtype == COGV_F16) {
    hipLaunchKernelGGL((colsum_partial_kernel<f16_t>), g1, dim3(256), 0, st, dy, M, N, ld, rows_per, part);
    hipLaunchKernelGGL((colsum_final_kernel<f16_t>), dim3((N + 63) / 64), dim3(256), 0, st, part, nslab, N, out, accumulate);
  } else {
    hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), g1, dim3(256), 0, st, dy, M, N, ld, rows_per, part);
    hipLaunchKernelGGL((colsum_final_kernel<bf16_t>), dim3((N + 63) / 64), dim3(256), 0, st, part, nslab, N, out, accumulate);
  }
  return cogv_check_launch();
}
