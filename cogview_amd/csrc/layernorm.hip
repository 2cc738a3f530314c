// Sandwich-LN primitive of CogView (reference mpu/sparse_transformer.py:40-44):
//     LayerNorm(x) := FusedLayerNorm(x / (max|x| / 8)),   max over the WHOLE tensor, detached
// Identity used here:  LN_eps(x / c) == (x - mean) / sqrt(var + eps * c^2) * gamma + beta,   c = max|x| / 8,
// so the kernel never divides the tensor: it reads the global abs-max scalar that the PRODUCER of x left
// behind (GEMM epilogue / residual kernels / cogv_absmax) and folds it into epsilon.
//
// Forward: one wave (64 lanes) per row, row cached in registers, wave-shuffle reductions (no LDS, no barriers).
// Backward: one workgroup of ceil(h/512) waves per row, 8 columns per lane (see ln_bwd_kernel).
// HBM-bound: forward moves 2*h*2 B per row (+2*h*2 with the fused residual), backward 3*h*2 B.
#include "common.cuh"
#include "cogview_hip.h"

#include <cstdlib>

// rows in flight of the wide STREAM_IN backward (fp32 x, add_in, dx: 20 prefetch registers per row instead of 12).
// h = 2560, 26112 rows (tools/r3/mb_ln_stream.py): two rows at 162 registers 171.6 us, four rows at 256 registers
// (+12 B of scratch) 184.7 us.  COGV_LN_BWD_ROWS overrides at run time.
#ifndef COGV_LN_BWD_STREAM_IN_ROWS
#define COGV_LN_BWD_STREAM_IN_ROWS 2
#endif

namespace {

struct LnFwdArgs {
  const void* x; const void* gamma; const void* beta; const void* res;
  void* y; float* mean; float* rstd;
  const float* absmax_in; float* absmax_out;
  int rows, h; float eps;
};

// MODE (cogview_hip.h COGV_LN_*): 0 = every tensor in the storage type T; 1 (STREAM_IN) = x is the fp32 residual
// stream, y is T (LN1, LN2, final LN); 2 (STREAM_OUT) = x is T, residual and y are the fp32 stream (LN3, LN4: the
// LayerNorm output is added to the stream in fp32, no rounding in between), abs-max of y over fp32 values.
template <typename T, int NV, int MODE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs p) {
  typedef Row8<T, MODE == 1> XR;
  typedef Row8<T, MODE == 2> YR;
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  float eps = p.eps;
  if (p.absmax_in) { const float c = *p.absmax_in * 0.125f; eps = p.eps * c * c; }
  const float inv_h = 1.0f / (float)p.h;

  float g[NV][8], b[NV][8];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * 64 + lane) * 8;
    if (col < p.h) {
      unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.gamma) + col), g[v]);
      unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.beta) + col), b[v]);
    }
  }
  uint32_t amax = 0u;                  // running max of |output| bit patterns (absmax_pk / fp32 patterns)
  for (int row = wave_global; row < p.rows; row += nwaves) {
    float x[NV][8];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      if (col < p.h) {
        XR::to_f(XR::ld(p.x, (size_t)row * p.h + col), x[v]);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[v][i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[v][i] = 0.f;
      }
    }
    const float mean = wave_sum_uniform(s) * inv_h;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      if (col < p.h) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = x[v][i] - mean; q += d * d; }
      }
    }
    const float var = wave_sum_uniform(q) * inv_h;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) { if (p.mean) p.mean[row] = mean; if (p.rstd) p.rstd[row] = rstd; }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * 64 + lane) * 8;
      if (col < p.h) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (x[v][i] - mean) * rstd * g[v][i] + b[v][i];
        if (p.res) {
          float r[8];
          if (MODE != 2) {
            // all-T form (stand-alone modules): the reference rounds LN's output to the storage type before the
            // residual add (mpu/sparse_transformer.py:326-329, :337-340); that rounding point is kept here
            u32x4 lo = pack8<T>(o); unpack8<T>(lo, o);
          }
          YR::to_f(YR::ld(p.res, (size_t)row * p.h + col), r);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += r[i];
        }
        amax = YR::st(p.y, (size_t)row * p.h + col, o, amax);
      }
    }
  }
  if (p.absmax_out) {
    __shared__ uint32_t red[16];
    const float bm = MODE == 2 ? absmax_f32_block(amax, red) : absmax_pk_block<T>(amax, red);
    if (threadIdx.x == 0) atomic_max_nonneg(p.absmax_out, bm);
  }
}

struct LnBwdArgs {
  const void* dy; const void* x; const void* gamma; const float* mean; const float* rstd;
  const void* add_in;      // optional [rows,h] (T): dx_out = add_in + dx
  void* dx;
  float* partial;          // [gridDim.x][3][h] fp32: dgamma, dbeta, colsum(dx_out)
  int rows, h;
  int want_colsum;
  uint64_t seed, stream_id; uint32_t thr16; float keep_scale;
  int marked;              // the dropout mask is read from x: 16-bit pattern 0x8000 (-0.0) <=> dropped (cogv_sandwich_ln_bwd_marked)
};

// "dropped" test of element i of 8 raw 16-bit values (marked zeros: the GEMM dropout epilogue writes a dropped element as -0.0
// and no kept element as -0.0, gemm_shared.cuh epilogue8)
__device__ __forceinline__ bool marked_dropped(const u32x4& raw, int i) {
  const uint32_t w = raw[i >> 1];
  return (i & 1) ? ((w >> 16) == 0x8000u) : ((w & 0xffffu) == 0x8000u);
}

// Backward.  One workgroup of ceil(h / 512) waves covers a row: every lane owns 8 columns for the whole kernel, so
// the three per-column accumulators (dgamma, dbeta, column sum of the output) are 24 registers per lane and need
// no cross-wave reduction -- the wave-per-row form kept 3 * h / 64 of them per lane (120 at h = 2560), which capped
// occupancy at two waves per SIMD and ran at 2 TB/s.  R rows are in flight per iteration (R * 2..3 16-byte
// loads per lane); the two row statistics go through a double-buffered LDS exchange, one barrier per R rows.
// MODE as in ln_fwd_kernel, seen from the backward side: 1 (STREAM_IN: LN1, LN2) = x, add_in and dx are the fp32
// stream / its gradient, dy is T;  2 (STREAM_OUT: LN3, LN4) = dy is the fp32 stream gradient, x, add_in, dx are T.
// LEAN (round 5; the dropout-replay forms LN3' / LN4': MODE 2, no add_in): no add_in registers at all (compile time), and the
// 16-bit x of the rows in flight is kept RAW (4 registers per row) and re-normalised in the second phase instead of holding
// x-hat in fp32 (8 per row) across the barrier -- 146 -> <= 128 registers, i.e. four waves per SIMD = three 5-wave workgroups
// per CU instead of two.  The replay's hash and the column sums put ~170 VALU instructions on every row of this form (the
// plain forms: ~90), so it needs more waves to keep the memory pipe busy; same arithmetic, bit-identical results.
// MARK (round 6; 16-bit x only): the keep mask is read from x's marked zeros instead of being re-hashed -- the hash, the
// 64-bit element index and the 16-bit field compares leave the row's instruction stream (the raw x of the rows in flight is
// kept for it, as in the lean form); bit-identical to the replay when x came from the marking GEMM epilogue.
template <typename T, int R, int MODE, bool LEAN = false, bool MARK = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(LEAN ? 4 : (R == 2 ? 3 : 2))))
void ln_bwd_kernel(const LnBwdArgs p) {
  typedef Row8<T, MODE == 2> DYR;
  typedef Row8<T, MODE == 1> XR;       // x, add_in, dx
  static_assert(!LEAN || MODE == 2, "the lean form keeps a 16-bit x raw");
  static_assert(!MARK || MODE != 1, "marked zeros live in a 16-bit x");
  constexpr bool KEEP_RAW = LEAN || MARK;
  __shared__ float red[2][R][8][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int col = threadIdx.x * 8;
  const bool act = col < p.h;
  const bool has_add = !LEAN && p.add_in != nullptr;
  const float inv_h = 1.0f / (float)p.h;

  float g[8], dg[8], db[8], cs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { g[i] = 0.f; dg[i] = 0.f; db[i] = 0.f; cs[i] = 0.f; }
  u32x4 graw = {0u, 0u, 0u, 0u};       // LEAN: gamma stays in its 16-bit form (4 registers) and is widened where it is used
  if (act) graw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.gamma) + col);
  if (!LEAN) unpack8<T>(graw, g);
  int buf = 0;
  typename DYR::raw dyn[R];            // the NEXT iteration's rows: loaded before this iteration's barrier and stores
  typename XR::raw xn_[R], adn[R];
  float meann[R], rstdn[R];
  auto fetch = [&](int row0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      const bool ok = act && row < p.rows;
      dyn[r] = ok ? DYR::ld(p.dy, (size_t)row * p.h + col) : DYR::zero();
      xn_[r] = ok ? XR::ld(p.x, (size_t)row * p.h + col) : XR::zero();
      if (has_add) adn[r] = ok ? XR::ld(p.add_in, (size_t)row * p.h + col) : XR::zero();
      meann[r] = row < p.rows ? p.mean[row] : 0.f;
      rstdn[r] = row < p.rows ? p.rstd[row] : 0.f;
    }
  };
  fetch(blockIdx.x * R);
  for (int row0 = blockIdx.x * R; row0 < p.rows; row0 += gridDim.x * R) {
    typename XR::raw adv[R], xraw[KEEP_RAW ? R : 1];
    float rstd[R], s1[R], s2[R], mrk[R];
    float xhk[LEAN ? 1 : R][8], gyk[R][8];   // normalised input and gamma * dy of the rows in flight (kept for phase 2)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float dy[8], xh1[8];
      float (&xh)[8] = LEAN ? xh1 : xhk[LEAN ? 0 : r];
      if (LEAN) unpack8<T>(graw, g);
      DYR::to_f(dyn[r], dy); XR::to_f(xn_[r], xh);
      if (KEEP_RAW) xraw[r] = xn_[r];
      if (!LEAN) adv[r] = adn[r];
      rstd[r] = rstdn[r];
      const float mr = meann[r] * rstdn[r];
      mrk[r] = mr;
      float a1 = 0.f, a2 = 0.f;
      if (act && row0 + r < p.rows) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          xh[i] = fmaf(xh[i], rstd[r], -mr);
          gyk[r][i] = dy[i] * g[i];
          // (opaque: the ROUNDED product is what both phases use -- left visible, the compiler may re-derive it in phase 2 and
          //  fuse it there with the subtraction of the row mean, one rounding fewer in some instantiations than in others)
          asm volatile("" : "+v"(gyk[r][i]));
          a1 += gyk[r][i];
          a2 = fmaf(gyk[r][i], xh[i], a2);
          dg[i] = fmaf(dy[i], xh[i], dg[i]);
          db[i] += dy[i];
        }
      }
      s1[r] = wave_sum_uniform(a1); s2[r] = wave_sum_uniform(a2);
    }
    fetch(row0 + gridDim.x * R);       // next iteration's rows (rows past the end load nothing)
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) { red[buf][r][wave][0] = s1[r]; red[buf][r][wave][1] = s2[r]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      float m1 = 0.f, m2 = 0.f;
      for (int w = 0; w < nw; ++w) { m1 += red[buf][r][w][0]; m2 += red[buf][r][w][1]; }
      m1 *= inv_h; m2 *= inv_h;
      if (act && row < p.rows) {
        float o[8], xh2[8];
        if (LEAN) {                         // the same fused multiply-add as in phase 1: bit-identical x-hat
          XR::to_f(xraw[r], xh2);
#pragma unroll
          for (int i = 0; i < 8; ++i) xh2[i] = fmaf(xh2[i], rstd[r], -mrk[r]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)        // (explicit fused multiply-add: the same contraction in every instantiation)
          o[i] = rstd[r] * fmaf(-(LEAN ? xh2[i] : xhk[LEAN ? 0 : r][i]), m2, gyk[r][i] - m1);
        if (MARK) {
          if constexpr (MARK) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = marked_dropped(xraw[r], i) ? 0.f : o[i] * p.keep_scale;
          }
        } else if (p.thr16) {
          const uint64_t e = (uint64_t)row * (uint64_t)p.h + (uint64_t)col;
          const u32x4 rn = Philox::gen(p.seed, p.stream_id, e >> 3);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (drop_bits16(rn, i) >= p.thr16) ? o[i] * p.keep_scale : 0.f;
        }
        if (has_add) {
          float a[8]; XR::to_f(adv[r], a);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += a[i];
        }
        (void)XR::st(p.dx, (size_t)row * p.h + col, o, 0u);        // o <- the stored (rounded) values
        if (p.want_colsum) {
#pragma unroll
          for (int i = 0; i < 8; ++i) cs[i] += o[i];
        }
      }
    }
    buf ^= 1;
  }
  if (act) {
    float* out = p.partial + (size_t)blockIdx.x * 3 * p.h + col;
#pragma unroll
    for (int i = 0; i < 8; ++i) { out[i] = dg[i]; out[p.h + i] = db[i]; if (p.want_colsum) out[2 * p.h + i] = cs[i]; }
  }
}

// ---- LN2' and LN3' of a transformer layer in ONE pass over their rows (round 6).  In backward the two are neighbours with no GEMM
// between them (mpu/sparse_transformer.py:326-341 read backwards: y = x + LN3(ao) feeds LN2 and the second residual):
//     dy   = dout + LN2'(dc ; y)                 (STREAM_IN form: dc is T, y / dout / dy are the fp32 stream)
//     d_ao = mask( LN3'(dy ; ao) )               (STREAM_OUT form with marked zeros: ao and d_ao are T)
// As two launches dy is written (it is also LN1's add_in later) and read back at once: 4 of the pair's 22 bytes per element.
// Here a row's dy stays in registers between the two: same arithmetic, expression for expression, as ln_bwd_kernel's MODE 1
// and MODE 2 + MARK forms -- dy and d_ao are bit-identical to the two launches; the five column reductions differ from them
// only in how rows are dealt to workgroups (fp32 summation order).  Two barriers per R rows (two row statistics each).
#ifndef COGV_LN_PAIR_OPAQUE_MEANS
#define COGV_LN_PAIR_OPAQUE_MEANS 1
#endif
struct LnPairArgs {
  const void* dc; const void* y; const void* gamma2; const float* mean2; const float* rstd2; const void* dout; void* dy;
  const void* ao; const void* gamma3; const float* mean3; const float* rstd3; void* d_ao;
  float* partial2;         // [gridDim.x][3][h]: dgamma2, dbeta2, (unused)
  float* partial3;         // [gridDim.x][3][h]: dgamma3, dbeta3, colsum(d_ao)
  int rows, h; int marked; float keep_scale;
};
template <typename T, int R>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
void ln_bwd_pair_kernel(const LnPairArgs p) {
  typedef Row8<T, false> TR;           // dc, ao, d_ao
  typedef Row8<T, true> FR;            // y, dout, dy
  __shared__ float red[2][R][8][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int col = threadIdx.x * 8;
  const bool act = col < p.h;
  const float inv_h = 1.0f / (float)p.h;
  float g2[8], g3[8], dg2[8], db2[8], dg3[8], db3[8], cs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { g2[i] = 0.f; g3[i] = 0.f; dg2[i] = 0.f; db2[i] = 0.f; dg3[i] = 0.f; db3[i] = 0.f; cs[i] = 0.f; }
  if (act) {
    unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.gamma2) + col), g2);
    unpack8<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.gamma3) + col), g3);
  }
  typename TR::raw dcn[R], aon[R];
  typename FR::raw yn[R], don[R];
  float m2n[R], r2n[R], m3n[R], r3n[R];
  auto fetch = [&](int row0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      const bool ok = act && row < p.rows;
      const size_t at = (size_t)row * p.h + col;
      dcn[r] = ok ? TR::ld(p.dc, at) : TR::zero();
      yn[r] = ok ? FR::ld(p.y, at) : FR::zero();
      don[r] = ok ? FR::ld(p.dout, at) : FR::zero();
      aon[r] = ok ? TR::ld(p.ao, at) : TR::zero();
      const bool rv = row < p.rows;
      m2n[r] = rv ? p.mean2[row] : 0.f; r2n[r] = rv ? p.rstd2[row] : 0.f;
      m3n[r] = rv ? p.mean3[row] : 0.f; r3n[r] = rv ? p.rstd3[row] : 0.f;
    }
  };
  fetch(blockIdx.x * R);
  for (int row0 = blockIdx.x * R; row0 < p.rows; row0 += gridDim.x * R) {
    typename TR::raw aor[R];
    typename FR::raw dor[R];
    float rstd2[R], rstd3[R], mr3[R], s1[R], s2[R];
    float xh2[R][8], gy2[R][8];
    // ---- LN2', first phase: the row's two statistics (MODE 1 of ln_bwd_kernel)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float dc[8];
      TR::to_f(dcn[r], dc); FR::to_f(yn[r], xh2[r]);
      aor[r] = aon[r]; dor[r] = don[r];
      rstd2[r] = r2n[r]; rstd3[r] = r3n[r];
      const float mr2 = m2n[r] * r2n[r];
      mr3[r] = m3n[r] * r3n[r];
      float a1 = 0.f, a2 = 0.f;
      if (act && row0 + r < p.rows) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          xh2[r][i] = fmaf(xh2[r][i], rstd2[r], -mr2);
          gy2[r][i] = dc[i] * g2[i];
          asm volatile("" : "+v"(gy2[r][i]));          // the ROUNDED product in both phases (as in ln_bwd_kernel)
          a1 += gy2[r][i];
          a2 = fmaf(gy2[r][i], xh2[r][i], a2);
          dg2[i] = fmaf(dc[i], xh2[r][i], dg2[i]);
          db2[i] += dc[i];
        }
      }
      s1[r] = wave_sum_uniform(a1); s2[r] = wave_sum_uniform(a2);
    }
    fetch(row0 + gridDim.x * R);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) { red[0][r][wave][0] = s1[r]; red[0][r][wave][1] = s2[r]; }
    }
    __syncthreads();
    // ---- LN2', second phase: dy = dout + LN2'(dc) -> memory (fp32) and registers; LN3', first phase on it (MODE 2)
    float dy[R][8], xh3[R][8], gy3[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      float m1 = 0.f, m2 = 0.f;
      for (int w = 0; w < nw; ++w) { m1 += red[0][r][w][0]; m2 += red[0][r][w][1]; }
      m1 *= inv_h; m2 *= inv_h;
#if COGV_LN_PAIR_OPAQUE_MEANS
      asm volatile("" : "+v"(m1), "+v"(m2));
#endif
      float b1 = 0.f, b2 = 0.f;
      if (act && row < p.rows) {
        float a[8];
        FR::to_f(dor[r], a);
        // (the product is ROUNDED before add_in joins it, as in ln_bwd_kernel, where the add sits behind a run-time branch and
        //  cannot contract with it; left visible here the compiler fuses the two and a quarter of dy differs in the last bit)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float t = rstd2[r] * fmaf(-xh2[r][i], m2, gy2[r][i] - m1);
          asm volatile("" : "+v"(t));
          dy[r][i] = t + a[i];
        }
        (void)FR::st(p.dy, (size_t)row * p.h + col, dy[r], 0u);
        TR::to_f(aor[r], xh3[r]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          xh3[r][i] = fmaf(xh3[r][i], rstd3[r], -mr3[r]);
          gy3[r][i] = dy[r][i] * g3[i];
          asm volatile("" : "+v"(gy3[r][i]));
          b1 += gy3[r][i];
          b2 = fmaf(gy3[r][i], xh3[r][i], b2);
          dg3[i] = fmaf(dy[r][i], xh3[r][i], dg3[i]);
          db3[i] += dy[r][i];
        }
      }
      s1[r] = wave_sum_uniform(b1); s2[r] = wave_sum_uniform(b2);
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) { red[1][r][wave][0] = s1[r]; red[1][r][wave][1] = s2[r]; }
    }
    __syncthreads();
    // ---- LN3', second phase: d_ao = mask(LN3'(dy)) -> memory (T), column sums of the stored values
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      float n1 = 0.f, n2 = 0.f;
      for (int w = 0; w < nw; ++w) { n1 += red[1][r][w][0]; n2 += red[1][r][w][1]; }
      n1 *= inv_h; n2 *= inv_h;
#if COGV_LN_PAIR_OPAQUE_MEANS
      asm volatile("" : "+v"(n1), "+v"(n2));
#endif
      if (act && row < p.rows) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rstd3[r] * fmaf(-xh3[r][i], n2, gy3[r][i] - n1);
        if (p.marked) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = marked_dropped(aor[r], i) ? 0.f : o[i] * p.keep_scale;
        }
        (void)TR::st(p.d_ao, (size_t)row * p.h + col, o, 0u);
#pragma unroll
        for (int i = 0; i < 8; ++i) cs[i] += o[i];
      }
    }
    // (the next iteration's first barrier orders this iteration's reads of red[1] against its writes two barriers later)
  }
  if (act) {
    float* o2 = p.partial2 + (size_t)blockIdx.x * 3 * p.h + col;
    float* o3 = p.partial3 + (size_t)blockIdx.x * 3 * p.h + col;
#pragma unroll
    for (int i = 0; i < 8; ++i) { o2[i] = dg2[i]; o2[p.h + i] = db2[i]; o3[i] = dg3[i]; o3[p.h + i] = db3[i]; o3[2 * p.h + i] = cs[i]; }
  }
}

// sums partial[nblk][3][h] over nblk and writes dgamma/dbeta/colsum in T (optionally accumulating)
template <typename T>
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* partial, int nblk, int h, void* dgamma,
                                                            void* dbeta, void* colsum, int accumulate) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;   // 16 row-slices
  const int set = blockIdx.y;
  T* out = reinterpret_cast<T*>(set == 0 ? dgamma : set == 1 ? dbeta : colsum);
  __shared__ float red[16][64];
  float s = 0.f;
  if (c < h && out) {
    // independent partial sums: the loads of one thread are all in flight together (a rolled loop of dependent
    // adds made this a chain of ~32 memory latencies)
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = part;
    for (; b + 7 * 16 < nblk; b += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += partial[((size_t)(b + 16 * u) * 3 + set) * h + c];
    }
    for (; b < nblk; b += 16) t[0] += partial[((size_t)b * 3 + set) * h + c];
    s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && c < h && out) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
    if (accumulate) t += HT<T>::to_f(out[c]);
    out[c] = HT<T>::from_f(t);
  }
}

// Forward grid: ONE resident round of workgroups, each wave looping over its rows (h = 2560: 152 registers = 3
// workgroups per CU = 768; the former 4096 ran 5.3 rounds with a third-full last one: 47.4 -> 45.1 us plain,
// 95.8 -> 84.2 us with the fused residual add; h = 1024: 1536 resident, same time as 4096).  COGV_LN_FWD_BLOCKS overrides.
template <typename T, int NV, int MODE> void launch_fwd_m(const LnFwdArgs& a, int blocks, hipStream_t st) {
  static const int resident = [] {
    const char* e = getenv("COGV_LN_FWD_BLOCKS");
    if (e && atoi(e) > 0) return atoi(e);
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ln_fwd_kernel<T, NV, MODE>, 256, 0) != hipSuccess || per_cu < 1) return 4096;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 4096;
    return per_cu * prop.multiProcessorCount;
  }();
  if (blocks > resident) blocks = resident;
  hipLaunchKernelGGL((ln_fwd_kernel<T, NV, MODE>), dim3(blocks), dim3(256), 0, st, a);
}
template <typename T, int NV> void launch_fwd(const LnFwdArgs& a, int mode, int blocks, hipStream_t st) {
  if (mode == COGV_LN_STREAM_IN) launch_fwd_m<T, NV, 1>(a, blocks, st);
  else if (mode == COGV_LN_STREAM_OUT) launch_fwd_m<T, NV, 2>(a, blocks, st);
  else launch_fwd_m<T, NV, 0>(a, blocks, st);
}
inline int ln_bwd_stream_in_rows() {
  static const int rows_env = [] { const char* e = getenv("COGV_LN_BWD_ROWS"); return e ? atoi(e) : 0; }();
  return rows_env ? rows_env : COGV_LN_BWD_STREAM_IN_ROWS;
}
#ifndef COGV_LN_BWD_LEAN_DEFAULT
#define COGV_LN_BWD_LEAN_DEFAULT 1
#endif
inline bool ln_bwd_lean() {
  const char* e = getenv("COGV_LN_BWD_LEAN");        // read per launch (A/B runs, tests)
  return e ? atoi(e) != 0 : COGV_LN_BWD_LEAN_DEFAULT != 0;
}
// rows in flight of the marked-zeros form at wide rows: 4 (the no-dropout geometry: one workgroup per CU) or 2 (the replay
// form's geometry: lean, three workgroups per CU).  COGV_LN_BWD_MARKED_ROWS overrides.  h = 2560, 26112 rows, LN4' with column
// sums (profiles/r06_ln_marked_zeros_rows_ab.log): replaying form 116.6 us (4.59 TB/s), marked two rows 125.0, marked four rows
// 110.2 (4.85 TB/s), no dropout at all 107.5 -- the hash was never the bound, the rows in flight are.
#ifndef COGV_LN_BWD_MARKED_ROWS_DEFAULT
#define COGV_LN_BWD_MARKED_ROWS_DEFAULT 4
#endif
inline int ln_bwd_marked_rows() {
  const char* e = getenv("COGV_LN_BWD_MARKED_ROWS");
  return e ? atoi(e) : COGV_LN_BWD_MARKED_ROWS_DEFAULT;
}
template <typename T, int MODE> void launch_bwd_m(const LnBwdArgs& a, int blocks, hipStream_t st) {
  const int nw = (a.h + 511) / 512;             // waves per row (h <= 4096 -> <= 8)
  // wide rows with the dropout replay: two rows in flight at 128 registers (two workgroups per CU) beat four rows at
  // 206 (one per CU) -- 112 vs 129 us at h = 2560; without the replay four rows and one workgroup per CU win (109 vs 116)
  if (a.marked && a.thr16 && MODE != 1) {
    if constexpr (MODE != 1) {
      if (nw >= 4 && ln_bwd_marked_rows() == 4) hipLaunchKernelGGL((ln_bwd_kernel<T, 4, MODE, false, true>), dim3(blocks), dim3(nw * 64), 0, st, a);
      else if (MODE == 2 && nw >= 4 && !a.add_in && ln_bwd_lean()) {
        if constexpr (MODE == 2) hipLaunchKernelGGL((ln_bwd_kernel<T, 2, MODE, true, true>), dim3(blocks), dim3(nw * 64), 0, st, a);
      } else if (nw >= 4) hipLaunchKernelGGL((ln_bwd_kernel<T, 2, MODE, false, true>), dim3(blocks), dim3(nw * 64), 0, st, a);
      else hipLaunchKernelGGL((ln_bwd_kernel<T, 4, MODE, false, true>), dim3(blocks), dim3(nw * 64), 0, st, a);
    }
  } else if (MODE == 2 && nw >= 4 && a.thr16 && !a.add_in && ln_bwd_lean()) {
    if constexpr (MODE == 2) hipLaunchKernelGGL((ln_bwd_kernel<T, 2, MODE, true>), dim3(blocks), dim3(nw * 64), 0, st, a);
  } else if (nw >= 4 && (a.thr16 || (MODE == 1 && ln_bwd_stream_in_rows() == 2)))
    hipLaunchKernelGGL((ln_bwd_kernel<T, 2, MODE>), dim3(blocks), dim3(nw * 64), 0, st, a);
  else hipLaunchKernelGGL((ln_bwd_kernel<T, 4, MODE>), dim3(blocks), dim3(nw * 64), 0, st, a);
}
template <typename T> void launch_bwd(const LnBwdArgs& a, int mode, int blocks, hipStream_t st) {
  if (mode == COGV_LN_STREAM_IN) launch_bwd_m<T, 1>(a, blocks, st);
  else if (mode == COGV_LN_STREAM_OUT) launch_bwd_m<T, 2>(a, blocks, st);
  else launch_bwd_m<T, 0>(a, blocks, st);
}

#define NV_SWITCH(FN, T, nv, ...)                          \
  switch (nv) {                                            \
    case 1: FN<T, 1>(__VA_ARGS__); break;                  \
    case 2: FN<T, 2>(__VA_ARGS__); break;                  \
    case 3: FN<T, 3>(__VA_ARGS__); break;                  \
    case 4: FN<T, 4>(__VA_ARGS__); break;                  \
    case 5: FN<T, 5>(__VA_ARGS__); break;                  \
    case 6: FN<T, 6>(__VA_ARGS__); break;                  \
    case 7: FN<T, 7>(__VA_ARGS__); break;                  \
    default: FN<T, 8>(__VA_ARGS__); break;                 \
  }


}  // namespace

// Workgroups of the backward kernel: exactly ONE round of resident workgroups (each loops over its rows) measured best
// at the wide rows -- h = 2560 (5 waves per workgroup): 256 workgroups of the four-row form (one per CU) 96 / 110 us
// (plain / residual-gradient add) vs 101 / 115 with 512, and 512 of the two-row form (two per CU) for the dropout-replay
// variant 112 vs 134 us; a third resident workgroup or a second round is 10-30 % slower.  h = 1024 (2 waves per
// workgroup): 1024 workgroups 44 us vs 59 with 512, 79 with 256.
static int ln_bwd_blocks(int rows, int h, bool dropout_replay /* or any other use of the two-row form */) {
  static const int forced = [] { const char* e = getenv("COGV_LN_BWD_BLOCKS"); return e ? atoi(e) : 0; }();
  const int nw = (h + 511) / 512;
  int cap = nw >= 4 ? (dropout_replay ? 512 : 256) : 2560 / nw;
  if (forced > 0) cap = forced > 1024 ? 1024 : forced;
  else cap = cap < 256 ? 256 : (cap > 1024 ? 1024 : cap);
  int b = (rows + 3) / 4;                       // 4 rows per workgroup iteration
  return b < 1 ? 1 : (b > cap ? cap : b);
}
// upper bound of the workgroup count over all widths (sizes the partial-sum workspace)
extern "C" int cogv_ln_bwd_num_blocks(int rows) {
  const int b = (rows + 3) / 4;
  return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}
extern "C" size_t cogv_ln_bwd_workspace_bytes(int rows, int h) {
  return (size_t)cogv_ln_bwd_num_blocks(rows) * 3 * (size_t)h * sizeof(float);      // sized for the largest block count
}

extern "C" int cogv_sandwich_ln_fwd(int dtype, const void* x, const void* gamma, const void* beta, const void* residual,
                                    void* y, float* mean, float* rstd, const float* absmax_in, float* absmax_out,
                                    int rows, int h, float eps, int stream_mode, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (rows <= 0 || h <= 0 || (h & 7) || h > 4096) return COGV_ERR_ARG;
  if (!x || !gamma || !beta || !y) return COGV_ERR_ARG;
  if (stream_mode < 0 || stream_mode > 2) return COGV_ERR_ARG;
  if (stream_mode == COGV_LN_STREAM_IN && residual) return COGV_ERR_ARG;       // the stream is the input, nothing to add to
  if (stream_mode == COGV_LN_STREAM_OUT && !residual) return COGV_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)residual) & 15) return COGV_ERR_ARG;
  LnFwdArgs a{x, gamma, beta, residual, y, mean, rstd, absmax_in, absmax_out, rows, h, eps};
  const int nv = (h + 511) / 512;
  int blocks = (rows + 3) / 4; if (blocks > 4096) blocks = 4096;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COGV_F16) { NV_SWITCH(launch_fwd, f16_t, nv, a, stream_mode, blocks, st) } else { NV_SWITCH(launch_fwd, bf16_t, nv, a, stream_mode, blocks, st) }
  return cogv_check_launch();
}

static int ln_bwd_impl(int dtype, const void* dy, const void* x, const void* gamma, const float* mean,
                       const float* rstd, const void* add_in, void* dx, void* dgamma, void* dbeta,
                       void* colsum, int accumulate_param_grads, int rows, int h, float dropout_p,
                       uint64_t seed, uint64_t stream_id, void* workspace, size_t workspace_bytes,
                       int stream_mode, void* stream, int marked) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (stream_mode < 0 || stream_mode > 2) return COGV_ERR_ARG;
  if (rows <= 0 || h <= 0 || (h & 7) || h > 4096) return COGV_ERR_ARG;
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !workspace) return COGV_ERR_ARG;
  if (workspace_bytes < cogv_ln_bwd_workspace_bytes(rows, h)) return COGV_ERR_ARG;
  if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)add_in) & 15) return COGV_ERR_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return COGV_ERR_ARG;
  LnBwdArgs a;
  a.dy = dy; a.x = x; a.gamma = gamma; a.mean = mean; a.rstd = rstd; a.add_in = add_in; a.dx = dx;
  a.partial = reinterpret_cast<float*>(workspace); a.rows = rows; a.h = h; a.want_colsum = colsum ? 1 : 0;
  a.seed = seed; a.stream_id = stream_id; a.marked = marked;
  a.thr16 = (uint32_t)(dropout_p * 65536.0f + 0.5f);
  a.keep_scale = 65536.0f / (65536.0f - (float)a.thr16);
  // (the two-row STREAM_IN form without dropout replay keeps the one-workgroup-per-CU cap: 256 workgroups 167 us vs 178
  // with 512 at h = 2560, tools/r3/exp1.sh)
  int blocks = ln_bwd_blocks(rows, h, a.thr16 != 0);
  if (marked && a.thr16 && (h + 511) / 512 >= 4 && ln_bwd_marked_rows() == 4) blocks = ln_bwd_blocks(rows, h, false);
  else if (stream_mode == COGV_LN_STREAM_OUT && a.thr16 && !add_in && (h + 511) / 512 >= 4 && ln_bwd_lean()) {
    // the lean dropout-replay form: three resident workgroups per CU (COGV_LN_BWD_BLOCKS still overrides)
    static const int forced = [] { const char* e = getenv("COGV_LN_BWD_BLOCKS"); return e ? atoi(e) : 0; }();
    if (forced <= 0) {
      const int want = (rows + 1) / 2, cap = cogv_ln_bwd_num_blocks(rows);      // (the partial-sum workspace is sized by `cap`)
      blocks = want < 768 ? want : 768;
      if (blocks > cap) blocks = cap;
    }
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COGV_F16) launch_bwd<f16_t>(a, stream_mode, blocks, st); else launch_bwd<bf16_t>(a, stream_mode, blocks, st);
  if (dgamma || dbeta || colsum) {
    dim3 grid((h + 63) / 64, 3);
    if (dtype == COGV_F16)
      hipLaunchKernelGGL((ln_bwd_reduce_kernel<f16_t>), grid, dim3(1024), 0, st, a.partial, blocks, h, dgamma, dbeta, colsum, accumulate_param_grads);
    else
      hipLaunchKernelGGL((ln_bwd_reduce_kernel<bf16_t>), grid, dim3(1024), 0, st, a.partial, blocks, h, dgamma, dbeta, colsum, accumulate_param_grads);
  }
  return cogv_check_launch();
}

extern "C" int cogv_sandwich_ln_bwd(int dtype, const void* dy, const void* x, const void* gamma, const float* mean,
                                    const float* rstd, const void* add_in, void* dx, void* dgamma, void* dbeta,
                                    void* colsum, int accumulate_param_grads, int rows, int h, float dropout_p,
                                    uint64_t seed, uint64_t stream_id, void* workspace, size_t workspace_bytes,
                                    int stream_mode, void* stream) {
  return ln_bwd_impl(dtype, dy, x, gamma, mean, rstd, add_in, dx, dgamma, dbeta, colsum, accumulate_param_grads, rows, h,
                     dropout_p, seed, stream_id, workspace, workspace_bytes, stream_mode, stream, 0);
}

extern "C" int cogv_sandwich_ln_bwd_marked(int dtype, const void* dy, const void* x, const void* gamma, const float* mean,
                                           const float* rstd, const void* add_in, void* dx, void* dgamma, void* dbeta,
                                           void* colsum, int accumulate_param_grads, int rows, int h, float dropout_p,
                                           void* workspace, size_t workspace_bytes, int stream_mode, void* stream) {
  if (stream_mode == COGV_LN_STREAM_IN) return COGV_ERR_ARG;        // the marks live in a 16-bit x
  return ln_bwd_impl(dtype, dy, x, gamma, mean, rstd, add_in, dx, dgamma, dbeta, colsum, accumulate_param_grads, rows, h,
                     dropout_p, 0, 0, workspace, workspace_bytes, stream_mode, stream, dropout_p > 0.f ? 1 : 0);
}

// LN2' + LN3' of a layer in one pass (ln_bwd_pair_kernel).  dropout_p > 0: ao carries marked zeros (cogv_gemm's dropout epilogue).
extern "C" size_t cogv_ln_bwd_pair_workspace_bytes(int rows, int h) { return 2 * cogv_ln_bwd_workspace_bytes(rows, h); }
extern "C" int cogv_sandwich_ln_bwd_pair(int dtype, const void* dc, const void* y, const void* gamma2, const float* mean2,
                                         const float* rstd2, const void* dout, void* dy, void* dgamma2, void* dbeta2,
                                         const void* ao, const void* gamma3, const float* mean3, const float* rstd3, void* d_ao,
                                         void* dgamma3, void* dbeta3, void* colsum, int accumulate_param_grads, int rows, int h,
                                         float dropout_p, void* workspace, size_t workspace_bytes, void* stream) {
  if (dtype != COGV_F16 && dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (rows <= 0 || h <= 0 || (h & 7) || h > 4096) return COGV_ERR_ARG;
  if (!dc || !y || !gamma2 || !mean2 || !rstd2 || !dout || !dy || !ao || !gamma3 || !mean3 || !rstd3 || !d_ao || !workspace) return COGV_ERR_ARG;
  if (workspace_bytes < cogv_ln_bwd_pair_workspace_bytes(rows, h)) return COGV_ERR_ARG;
  if (((uintptr_t)dc | (uintptr_t)y | (uintptr_t)gamma2 | (uintptr_t)dout | (uintptr_t)dy | (uintptr_t)ao | (uintptr_t)gamma3 | (uintptr_t)d_ao) & 15)
    return COGV_ERR_ARG;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return COGV_ERR_ARG;
  const int nw = (h + 511) / 512;
  if (nw < 4) return COGV_ERR_UNSUPPORTED;        // narrow rows: the two launches (their many-workgroup geometry) stay
  LnPairArgs a;
  a.dc = dc; a.y = y; a.gamma2 = gamma2; a.mean2 = mean2; a.rstd2 = rstd2; a.dout = dout; a.dy = dy;
  a.ao = ao; a.gamma3 = gamma3; a.mean3 = mean3; a.rstd3 = rstd3; a.d_ao = d_ao;
  a.partial2 = reinterpret_cast<float*>(workspace);
  a.partial3 = a.partial2 + (size_t)cogv_ln_bwd_num_blocks(rows) * 3 * (size_t)h;
  a.rows = rows; a.h = h;
  const uint32_t thr16 = (uint32_t)(dropout_p * 65536.0f + 0.5f);
  a.marked = thr16 != 0; a.keep_scale = 65536.0f / (65536.0f - (float)thr16);
  static const int forced = [] { const char* e = getenv("COGV_LN_BWD_PAIR_BLOCKS"); return e ? atoi(e) : 0; }();
  int blocks = forced > 0 ? forced : 512;
  const int want = (rows + 1) / 2, cap = cogv_ln_bwd_num_blocks(rows);
  if (blocks > want) blocks = want;
  if (blocks > cap) blocks = cap;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COGV_F16) hipLaunchKernelGGL((ln_bwd_pair_kernel<f16_t, 2>), dim3(blocks), dim3(nw * 64), 0, st, a);
  else hipLaunchKernelGGL((ln_bwd_pair_kernel<bf16_t, 2>), dim3(blocks), dim3(nw * 64), 0, st, a);
  dim3 grid((h + 63) / 64, 3);
  if (dtype == COGV_F16) {
    if (dgamma2 || dbeta2) hipLaunchKernelGGL((ln_bwd_reduce_kernel<f16_t>), grid, dim3(1024), 0, st, a.partial2, blocks, h, dgamma2, dbeta2, (void*)nullptr, accumulate_param_grads);
    if (dgamma3 || dbeta3 || colsum) hipLaunchKernelGGL((ln_bwd_reduce_kernel<f16_t>), grid, dim3(1024), 0, st, a.partial3, blocks, h, dgamma3, dbeta3, colsum, accumulate_param_grads);
  } else {
    if (dgamma2 || dbeta2) hipLaunchKernelGGL((ln_bwd_reduce_kernel<bf16_t>), grid, dim3(1024), 0, st, a.partial2, blocks, h, dgamma2, dbeta2, (void*)nullptr, accumulate_param_grads);
    if (dgamma3 || dbeta3 || colsum) hipLaunchKernelGGL((ln_bwd_reduce_kernel<bf16_t>), grid, dim3(1024), 0, st, a.partial3, blocks, h, dgamma3, dbeta3, colsum, accumulate_param_grads);
  }
  return cogv_check_launch();
}
