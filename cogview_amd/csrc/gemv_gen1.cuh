// First-generation skinny-M kernels (gemv_kernel, gemv_attn_kernel, gemv_ln_kernel): the fallback of csrc/gemv.hip for shapes outside its classes and for COGV_GEMV2=0.
// Part of the GEMM family of csrc/gemm.hip (included there, in this order: common, gen1, lds, gen2, gen3, gen4, gemv_gen1);
// not a stand-alone header.
#pragma once

namespace {

// =====================================================================================================
// Skinny-M kernel (M <= 8: incremental decoding, one row per beam): C[M,N] = epilogue(A[M,K] B[N,K]^T) is a matrix-VECTOR
// product per row -- every weight byte is used M times, so the kernel is a pure HBM stream of B (the 4B model reads its
// 7.9 GB of weights once per generated token).  A workgroup owns 8 output columns (8 rows of B); its 4 waves take the
// 512-element chunks of K round robin (16 bytes per lane per row: 8 independent 1-KiB loads per chunk in flight), each
// reduces its partial dot products with DPP, the four partials meet in LDS and lane 0 of wave 0 runs the shared fused
// epilogue on its 8 consecutive columns.  The MFMA tile kernels spend the same traffic on 128 rows of which one is real.
template <typename T>
__global__ __launch_bounds__(256) void gemv_kernel(const GemmArgs p) {
  __shared__ float part[4][GEMV_MAX_M][8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * 8;
  const T* A = reinterpret_cast<const T*>(p.A);
  const T* B = reinterpret_cast<const T*>(p.B) + (size_t)n0 * p.ldb;
  float acc[GEMV_MAX_M][8];
#pragma unroll
  for (int m = 0; m < GEMV_MAX_M; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
  const int nchunk = p.K >> 9;
  for (int c = wave; c < nchunk; c += 4) {
    const int k = (c << 9) + lane * 8;
    u32x4 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(B + (size_t)j * p.ldb + k);
#pragma unroll
    for (int m = 0; m < GEMV_MAX_M; ++m) {
      if (m < p.M) {
        float x[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(A + (size_t)m * p.lda + k), x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float wf[8];
          unpack8<T>(w[j], wf);
          float t = acc[m][j];
#pragma unroll
          for (int e = 0; e < 8; ++e) t = fmaf(x[e], wf[e], t);
          acc[m][j] = t;
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < GEMV_MAX_M; ++m)
    if (m < p.M) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = wave_sum_uniform(acc[m][j]);
        if (lane == 0) part[wave][m][j] = t;
      }
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t amax_pk = 0u;
    for (int m = 0; m < p.M; ++m) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[0][m][j] + part[1][m][j] + part[2][m][j] + part[3][m][j];
      amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n0, v));
    }
    if (p.flags & COGV_EPI_ABSMAX) {
      const uint32_t wv = max(amax_pk & 0xffffu, amax_pk >> 16);
      atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
  }
}

// Decode step, attention-output projection: the skinny-M kernel above with the COMBINE of the decode attention's key
// splits as its prologue (round 3: one launch per layer less in a captured decode step).  attn_decode_kernel leaves per
// (row, head, split) a partial (max m, sum l, 64 unnormalised outputs o) -- 66 floats; the attention output element the GEMV
// needs, att[row][head * 64 + d] = sum_s 2^(m_s - M) o_s[d] / sum_s 2^(m_s - M) l_s, is a few hundred bytes of L2-resident
// partials per lane, so every workgroup recombines the 8-element slices it multiplies instead of a separate combine launch
// writing att and this one reading it (attn_decode_combine_kernel; same arithmetic, splits in order, att rounded to the
// storage type before the product as the two-launch form stores it).  The first weight rows are requested before the
// prologue, so the HBM latency overlaps it.
template <typename T>
__global__ __launch_bounds__(256) void gemv_attn_kernel(const GemmArgs p, const float* __restrict__ part_ws, int H, int nsplit) {
  __shared__ float part[4][GEMV_MAX_M][8];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * 8;
  const T* B = reinterpret_cast<const T*>(p.B) + (size_t)n0 * p.ldb;
  float acc[GEMV_MAX_M][8];
#pragma unroll
  for (int m = 0; m < GEMV_MAX_M; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
  const int nchunk = p.K >> 9;
  for (int c = wave; c < nchunk; c += 4) {
    const int k = (c << 9) + lane * 8;
    u32x4 w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(B + (size_t)j * p.ldb + k);
    const int head = k >> 6, dd = k & 63;
#pragma unroll
    for (int m = 0; m < GEMV_MAX_M; ++m) {
      if (m < p.M) {
        const float* base = part_ws + ((size_t)m * H + head) * nsplit * 66;
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
        // the sum in the association of attn_decode_combine_kernel's xor-butterfly (lanes >= nsplit hold zeros there too),
        // so that both forms of the step produce the same bits
#pragma unroll
        for (int w2 = 16; w2 > 0; w2 >>= 1)
#pragma unroll
          for (int i = 0; i < w2; ++i) tl[i] += tl[i + w2];
        const float L = tl[0];
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = o[e] / L;
        const u32x4 xr = pack8<T>(x);                         // the attention output in its storage type
        unpack8<T>(xr, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float wf[8];
          unpack8<T>(w[j], wf);
          float t = acc[m][j];
#pragma unroll
          for (int e = 0; e < 8; ++e) t = fmaf(x[e], wf[e], t);
          acc[m][j] = t;
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < GEMV_MAX_M; ++m)
    if (m < p.M) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = wave_sum_uniform(acc[m][j]);
        if (lane == 0) part[wave][m][j] = t;
      }
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t amax_pk = 0u;
    for (int m = 0; m < p.M; ++m) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[0][m][j] + part[1][m][j] + part[2][m][j] + part[3][m][j];
      amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n0, v));
    }
    if (p.flags & COGV_EPI_ABSMAX) {
      const uint32_t wv = max(amax_pk & 0xffffu, amax_pk >> 16);
      atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Matrix-vector kernel with the layer's LayerNorms as its PROLOGUE (decode steps, M <= 8 rows, K = hidden size <= 4096).
// A decode step of one token spends more time in its four per-layer Sandwich-LN launches (7 us each: launch latency plus a
// dependent chain on one row) than the LayerNorms cost in bytes, so the chain
//     z --[post-LN gamma_p, beta_p, Sandwich scale |z|max]--> + residual --> t --[pre-LN gamma, beta, Sandwich scale |t|max]--> x_in
//     y = epilogue(x_in . W^T + b)
// (mpu/sparse_transformer.py:314-342: t = x + LN3(attn) feeding LN2, or t = y + LN4(mlp) feeding the next layer's LN1 /
// the final LayerNorm) runs inside EVERY workgroup of the GEMV that consumes x_in: the vectors are M x K 16-bit values, a
// few KB, and recomputing them 320-1280 times is cheaper than one more launch.  Workgroup 0 also stores t (the
// residual stream) once.  Rounding points are those of ln_fwd_kernel (LayerNorm output rounded to the storage type
// before the residual add, t rounded, x_in rounded).  |t|max is taken over all M rows, as x.abs().max() does.
// SF: the residual stream is fp32 -- `res`, `t_out` and (without a post-LN) `z` are fp32 rows, t is formed and
// normalised without an intermediate rounding (the decode counterpart of ln_fwd_kernel's STREAM modes).
template <typename T, int MT, bool SF>   // MT: compile-time bound of the row count (1, 2, 4, 8): registers follow the real batch
__global__ __launch_bounds__(256) void gemv_ln_kernel(const GemvLnArgs q) {
  typedef Row8<T, SF> SR;                // a stream row slice
  extern __shared__ __attribute__((aligned(16))) char xs_raw[];           // x_in [M][K] as T
  __shared__ float part[4][MT][8];
  __shared__ float red[8];
  __shared__ uint32_t redm[16];
  __shared__ float s_amax;
  const GemmArgs& p = q.g;
  T* xs = reinterpret_cast<T*>(xs_raw);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = p.K, nvec = K >> 3;                // K % 512 == 0: every thread owns whole 8-element vectors v = tid, tid + 256
  const float inv_k = 1.0f / (float)K;
  const bool has_post = q.gamma_p != nullptr;
  const int v0 = threadIdx.x, v1 = threadIdx.x + 256;
  const bool ok1 = v1 < nvec;                      // v0 < nvec always (K >= 2048 is not required: guard below)
  const bool ok0 = v0 < nvec;
  // ---- everything that does not depend on the prologue is requested first: this wave's first weight chunk (the HBM
  //      stream: its latency now overlaps the LayerNorm arithmetic), the input rows, the residual and the four affine vectors
  const int n0 = blockIdx.x * 8;
  const T* B = reinterpret_cast<const T*>(p.B) + (size_t)n0 * p.ldb;
  const int nchunk = K >> 9;
  u32x4 w[8];
  {
    const int k = (min(wave, nchunk - 1) << 9) + lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(B + (size_t)j * p.ldb + k);
  }
  const T* Z = reinterpret_cast<const T*>(q.z);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 zr[MT][2], gpr[2], bpr[2], gnr[2], bnr[2];
  typename SR::raw rr[MT][2];            // the stream rows: the residual (post-LN form) or z itself (plain-input form)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int v = u ? v1 : v0; const bool ok = u ? ok1 : ok0;
    gnr[u] = ok ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(q.gamma) + v * 8) : zero4;
    bnr[u] = ok ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(q.beta) + v * 8) : zero4;
    gpr[u] = (ok && has_post) ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(q.gamma_p) + v * 8) : zero4;
    bpr[u] = (ok && has_post) ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(q.beta_p) + v * 8) : zero4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      zr[m][u] = (ok && m < p.M && (has_post || !SF)) ? *reinterpret_cast<const u32x4*>(Z + (size_t)m * K + v * 8) : zero4;
      rr[m][u] = (ok && m < p.M && (has_post || SF)) ? SR::ld(has_post ? q.res : q.z, (size_t)m * K + v * 8) : SR::zero();
    }
  }
  float zamax = q.z_absmax ? *q.z_absmax : 0.f;
  // sums over the workgroup of up to 2 * MT values at once (one LDS round for all rows)
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
    for (int m = 0; m < MT; ++m) a[m] = (part[0][m][0] + part[1][m][0]) + (part[2][m][0] + part[3][m][0]);
  };
  float tv[MT][2][8];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (SF && !has_post) SR::to_f(rr[m][u], tv[m][u]);         // the plain input IS the fp32 stream
      else unpack8<T>(zr[m][u], tv[m][u]);
    }
  if (has_post) {                 // t = residual + LN_post(z), rounded where ln_fwd_kernel rounds
    if (!q.z_absmax) {            // max |z| over all rows taken here (see gemv2_ln_kernel)
      uint32_t zpk = 0u;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) zpk = absmax_pk8(zpk, zr[m][u]);
      zamax = absmax_pk_block<T>(zpk, redm);
    }
    const float c = zamax * 0.125f;
    const float eps_p = q.eps * c * c;
    float s[MT], qq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      s[m] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[m] += tv[m][u][i];            // vectors past K are zero
    }
    block_sums(s);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k;
      qq[m] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (u ? ok1 : ok0)
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = tv[m][u][i] - mean; qq[m] += d * d; }
    }
    block_sums(qq);
    float gp[2][8], bp[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) { unpack8<T>(gpr[u], gp[u]); unpack8<T>(bpr[u], bp[u]); }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k, rstd = 1.0f / sqrtf(qq[m] * inv_k + eps_p);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
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
        if (!(u ? ok1 : ok0)) {
#pragma unroll
          for (int i = 0; i < 8; ++i) tv[m][u][i] = 0.f;
        } else if (blockIdx.x == 0 && q.t_out && m < p.M)
          (void)SR::st(q.t_out, (size_t)m * K + (u ? v1 : v0) * 8, o, 0u);
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
      for (int u = 0; u < 2; ++u) {                                 // rows >= M and vectors past K are zero
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
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s[m] += tv[m][u][i];
    }
    block_sums(s);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k;
      qq[m] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (u ? ok1 : ok0)
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = tv[m][u][i] - mean; qq[m] += d * d; }
    }
    block_sums(qq);
    float gn[2][8], bn[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) { unpack8<T>(gnr[u], gn[u]); unpack8<T>(bnr[u], bn[u]); }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float mean = s[m] * inv_k, rstd = 1.0f / sqrtf(qq[m] * inv_k + eps_n);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u ? ok1 : ok0) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (tv[m][u][i] - mean) * rstd * gn[u][i] + bn[u][i];
          *reinterpret_cast<u32x4*>(xs + (size_t)m * K + (u ? v1 : v0) * 8) = pack8<T>(o);
        }
      }
    }
  }
  __syncthreads();
  // ---- the matrix-vector product proper (as gemv_kernel, x_in read from LDS; the first chunk is already in registers)
  float acc[MT][8];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
  for (int c = wave; c < nchunk; c += 4) {
    const int k = (c << 9) + lane * 8;
    if (c != wave) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const u32x4*>(B + (size_t)j * p.ldb + k);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float x[8];
      unpack8<T>(*reinterpret_cast<const u32x4*>(xs + (size_t)m * K + k), x);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float wf[8];
        unpack8<T>(w[j], wf);
        float t = acc[m][j];
#pragma unroll
        for (int e = 0; e < 8; ++e) t = fmaf(x[e], wf[e], t);
        acc[m][j] = t;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = wave_sum_uniform(acc[m][j]);
      if (lane == 0) part[wave][m][j] = t;
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t am = 0u;
    for (int m = 0; m < p.M; ++m) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[0][m][j] + part[1][m][j] + part[2][m][j] + part[3][m][j];
      am = absmax_pk(am, epilogue8<T>(p, m, n0, v));
    }
    if (p.flags & COGV_EPI_ABSMAX) {
      const uint32_t wv = max(am & 0xffffu, am >> 16);
      atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
  }
}

}  // namespace
