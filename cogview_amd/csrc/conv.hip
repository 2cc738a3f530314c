// VQ-VAE tokenizer kernels (reference vqvae/vqvae_zc.py, production config vqvae/api.py:12-20:
// channel 512, embed_dim 256, n_embed 8192, stride 6 -> three 4x4 stride-2 convs + 1x1, quantise,
// three 4x4 stride-2 transposed convs + 1x1).  gfx950, fp32 end to end.
//
// Why fp32: the reference runs the tokenizer in fp32 and the token id is an argmin over fp32 distances; the
// north star asks for ids that match the CPU path.  CDNA4 has an exact-fp32 MFMA (v_mfma_f32_32x32x2_f32,
// 157 TFLOP/s peak = 16x below bf16 MFMA, bit-for-bit an fmaf chain), so the convolutions stay on the matrix
// cores without giving up fp32 products/accumulation.  Only the summation ORDER differs from the CPU path.
//
// One implicit-GEMM kernel covers every layer through a tap table:
//     out[b, y*om+oy, x*om+ox, co] = act( bias[co] + sum_t sum_ci in[b, y*im+dy[t], x*im+dx[t], ci] * W[co][t][ci] )
//   4x4 stride-2 pad-1 conv   : im=2, om=1, 16 taps (dy,dx) = (ky-1, kx-1)
//   1x1 conv                  : im=1, om=1, 1 tap
//   4x4 stride-2 pad-1 convT  : 4 output parities (blockIdx.z), each a 2x2-tap conv at input resolution
//                               (sub-pixel decomposition), im=1, om=2
// Activations are NHWC fp32 (channels contiguous = the contraction index contiguous), weights are repacked
// once on the host side to [parity][Cout][tap][Cin].  Tile 128 pixels x 128 channels x 32 (k), 4 waves,
// 2x2 MFMA 32x32 per wave, LDS rows of 128 B with the same XOR swizzle as the 16-bit GEMM.
// The k-slot relabelling trick (attention.hip) lets one ds_read_b128 feed four MFMA k-steps.
#include "common.cuh"
#include "cogview_hip.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;   // BK in floats (128 B per LDS row)
constexpr int NT = 256;

struct ConvArgs {
  const float* in; const float* w; const float* bias; float* out;
  int B, IH, IW, Cin;
  int GH, GW;               // iteration grid (pixels per image = GH*GW)
  int OH, OW, Cout;
  int in_mul, out_mul;
  int ntaps;
  int kind;                 // COGV_CONV_*: tap offsets are arithmetic (tap_offset), no table in memory
  const float* rgb_w;       // optional fused 1x1 -> 3 projection of the (bias + ReLU) output: weights [3][Cout] ...
  float* rgb_part;          // ... partial sums [Cout / 128][B * OH * OW][4] instead of the output tensor
  long long w_parity_stride;
  int relu_out;
  int relu_in;              // ReLU applied to the INPUT activations while they are parked in LDS (pre-activation residual blocks)
  const float* residual;    // optional: out += (relu_res ? max(residual, 0) : residual), same NHWC shape as out, after bias / ReLU
  int relu_res;
  int nz;                   // parities (4 for the transposed convolution, else 1)
  int K;                    // ntaps * Cin
  int M;                    // B * GH * GW
};

// input offset (dy, dx) of tap `tap` for output parity z (see the header comment and cogv_conv2d_nhwc_f32):
//   4x4 s2 p1 conv : tap = ky*4 + kx           -> (ky - 1, kx - 1)
//   1x1            : (0, 0)
//   4x4 s2 p1 convT: tap = ty*2 + tx, z = py*2+px -> (py - ty, px - tx)   [py = 0: rows y, y-1;  py = 1: rows y+1, y]
// Pure integer arithmetic on wave-uniform values: a byte table in the kernel arguments costs a vector memory load and a
// full memory latency in front of every k-tile's operand loads (scalar loads have no byte form on gfx950).
__device__ __forceinline__ void tap_offset(int kind, int z, int tap, int& dy, int& dx) {
  if (kind == COGV_CONV_4X4_S2) { dy = (tap >> 2) - 1; dx = (tap & 3) - 1; }
  else if (kind == COGV_CONVT_4X4_S2) { dy = (z >> 1) - (tap >> 1); dx = (z & 1) - (tap & 1); }
  else if (kind == COGV_CONV_3X3_S1) { const int ky = (tap * 11) >> 5; dy = ky - 1; dx = tap - 3 * ky - 1; }   // tap / 3 for tap < 9
  else { dy = 0; dx = 0; }
}

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 7); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Asynchronous 16-byte load: issued through inline asm so that the COMPILER does not track it -- its own vmcnt
// bookkeeping turned "tile kt+1 is needed, tile kt+2 may stay in flight" into vmcnt(0) (it drained both register sets
// before every park in LDS, and half of the older set before issuing the newer one).  The consumer must call
// wait_loads<N>() on the destination registers before the first use (loads return in order: N = number of younger loads
// that may remain outstanding).  Must not be spilled around (cogview_amd/csrc/build.py scan_asm_hazards: no scratch access
// and no read of a destination register while the load is in flight).
__device__ __forceinline__ void ld4_async(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
// all asynchronous loads retired; the operands tie BOTH register sets to the wait, so that the compiler cannot reuse
// a register an in-flight load still targets (found by build.py's asm_load_hazards: the epilogue's address arithmetic
// had been scheduled into those registers in front of a bare s_waitcnt)
__device__ __forceinline__ void drain_loads(f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], f32x4 (&d)[4]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                 "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
               : : "memory");
}
template <int N>
__device__ __forceinline__ void wait_loads(f32x4 (&a)[4], f32x4 (&b)[4]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
               : "n"(N) : "memory");
}

// Operand staging.  thread t -> (tile row t/8 + 32 q for q = 0..3, 16-byte chunk t%8 of the 128-byte k-slice).
// Every load of a k-tile is UNCONDITIONAL, from a clamped (always valid) address, and the rows / taps that fall outside
// the image are zeroed when the registers are parked in LDS: a `v = 0; if (inside) v = load` form makes the compiler
// wait for each load before issuing the next one (the select needs the loaded value), which serialises eight memory
// latencies per k-tile -- measured: MFMA pipe 57 % busy with the waves waiting on vmcnt.
struct RowCtx {            // per thread, fixed for the whole tile: its four pixels (A) / rows (B)
  const float* base[4];    // A: &in[b][0][0][0] of row q's image;   B: &w[n][4 * (t & 7)]
  int y0[4], x0[4];        // A: y * in_mul, x * in_mul
  uint32_t valid;          // bit q: the row exists (m < M / n < N)
};
__device__ __forceinline__ RowCtx make_a_ctx(const ConvArgs& p, int m0) {
  RowCtx c; c.valid = 0u;
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + (t >> 3) + 32 * q;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / (p.GH * p.GW);
    const int rem = mm - b * (p.GH * p.GW);
    const int y = rem / p.GW, x = rem - y * p.GW;
    c.base[q] = p.in + (size_t)b * p.IH * p.IW * p.Cin;
    c.y0[q] = y * p.in_mul; c.x0[q] = x * p.in_mul;
    c.valid |= ok ? (1u << q) : 0u;
  }
  return c;
}
__device__ __forceinline__ RowCtx make_b_ctx(const float* w, int ldw, int n0, int N) {
  RowCtx c; c.valid = 0u;
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n0 + (t >> 3) + 32 * q;
    const bool ok = n < N;
    c.base[q] = w + (size_t)(ok ? n : 0) * ldw;          // row start; the lane's 16-byte chunk is added by load_b
    c.y0[q] = c.x0[q] = 0;
    c.valid |= ok ? (1u << q) : 0u;
  }
  return c;
}
// A tile: 128 pixels x 32 k; chunk = 4 consecutive k = 4 channels of one tap.  Returns the 4-bit keep mask.
// Offsets inside one image are 32-bit (IH * IW * Cin < 2^31, checked by the launcher); the products use the
// full-rate 24-bit multiplier (iy, ix < 2^15 and IW * Cin, Cin < 2^24, checked by the launcher).
template <bool UNIFORM_TAP>
__device__ __forceinline__ uint32_t load_a(const ConvArgs& p, const RowCtx& c, int z, int k0, f32x4 (&r)[4]) {
  const int t = threadIdx.x;
  int dy, dx, ci; bool kok;
  if (UNIFORM_TAP) {           // Cin % 32 == 0: the whole k-tile lies inside one tap -> scalar arithmetic
    const int tap = __builtin_amdgcn_readfirstlane(k0 / p.Cin);
    kok = true;                                        // K % 32 == 0 as well
    tap_offset(p.kind, z, tap, dy, dx);
    ci = k0 - tap * p.Cin + (t & 7) * 4;
  } else {
    const int k = k0 + (t & 7) * 4;
    kok = k < p.K;
    const int tap = kok ? k / p.Cin : 0;
    ci = kok ? k - tap * p.Cin : 0;
    tap_offset(p.kind, z, tap, dy, dx);
  }
  const int row_pitch = p.IW * p.Cin;
  uint32_t keep = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int iy = c.y0[q] + dy, ix = c.x0[q] + dx;
    const bool ok = kok && ((c.valid >> q) & 1u) && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
    const int off = ok ? __mul24(iy, row_pitch) + __mul24(ix, p.Cin) + ci : 0;
    ld4_async(r[q], c.base[q] + off);
    keep |= ok ? (1u << q) : 0u;
  }
  return keep;
}
// B tile: 128 rows x 32 k from a row-major [N][ldw] matrix (weights W[z][co][K]; x / E^T in the nearest-code search)
__device__ __forceinline__ uint32_t load_b(const RowCtx& c, int k0, int K, f32x4 (&r)[4]) {
  // a chunk past the row's K columns (K < 32 or K % 32 != 0) loads the ROW START instead (zeroed when parked): the
  // round-2 form added the chunk offset even then and read up to 112 bytes past the end of the matrix from its last row
  // -- a memory fault whenever that matrix ended a mapped segment (found in round 3: the 64 x 16 codebook of the small
  // golden model aborted the whole GPU suite in one particular allocation order)
  const int kc = k0 + (threadIdx.x & 7) * 4;
  const bool kok = kc < K;
#pragma unroll
  for (int q = 0; q < 4; ++q) ld4_async(r[q], c.base[q] + (kok ? kc : 0));
  return kok ? c.valid : 0u;
}
__device__ __forceinline__ void store_tile(char* lds, const f32x4 (&r)[4], uint32_t keep, bool relu = false) {
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (t >> 3) + 32 * q;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v = ((keep >> q) & 1u) ? r[q] : zero;
    if (relu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    *reinterpret_cast<f32x4*>(lds + row * 128 + (((t & 7) ^ swz(row)) << 4)) = v;
  }
}
__device__ __forceinline__ f32x4 frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const f32x4*>(lds + row * 128 + ((chunk ^ swz(row)) << 4));
}

// one k-tile (32 floats) of MFMAs for a wave's 64x64 sub-tile.  The fragments of k-block kb+1 are read into a second
// register set BEFORE the 16 MFMAs of k-block kb (the compiler's own order reused one set and exposed the LDS latency
// four times per k-tile); the scheduling barriers pin that order.
__device__ __forceinline__ void mma_tile(const char* la, const char* lb, int wm, int wn, int fr, int fg,
                                         f32x16 (&acc)[2][2]) {
  f32x4 fa[2][2], fb[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { fa[0][i] = frag(la, wm + 32 * i + fr, fg); fb[0][i] = frag(lb, wn + 32 * i + fr, fg); }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {          // 8 k per step: lanes g=0 take slots 0..3, g=1 slots 4..7
    const int c = kb & 1;
    if (kb + 1 < 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[c ^ 1][i] = frag(la, wm + 32 * i + fr, 2 * (kb + 1) + fg);
        fb[c ^ 1][i] = frag(lb, wn + 32 * i + fr, 2 * (kb + 1) + fg);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][s], fb[c][j][s], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Workgroup -> (pixel tile, channel tile, parity): the grid is one-dimensional and XCD-aware.  Hardware hands
// consecutive workgroup ids to the 8 XCDs round-robin, each XCD has its own 4-MiB L2, and 64 workgroups run on an XCD at
// a time (2 per CU).  An XCD therefore takes a contiguous chunk of pixel tiles and walks it with (channel tile, parity)
// varying fastest: the 64 concurrent workgroups are 64 / (channel tiles x parities) neighbouring pixel tiles times ALL
// their channel tiles / parities, so every activation line is fetched into that L2 once and hit by the other
// workgroups, and the weight slices are shared by the pixel tiles in flight.  (With a plain 3-D grid the workgroups
// that read the same activations sat on different XCDs or ran far apart in time: 50 % L2 hit rate, the input tensor
// streamed from HBM once per channel tile and parity -- 16 times for the last transposed convolution.)
template <bool UT, int EXP = 0>      // UT: Cin % 32 == 0 (every k-tile lies inside one tap); EXP: timing probes (wrong results)
__global__ __launch_bounds__(NT, 2) void conv_kernel(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages x (A 16K + B 16K) = 64 KiB
  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
  const int per = ntiles * p.nz, chunk = (mtiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int mt = xcd * chunk + local / per, yz = local % per;
  if (mt >= mtiles) return;
  const int z = yz / ntiles;
  const int m0 = mt * BM, n0 = (yz - z * ntiles) * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int fr = lane & 31, fg = lane >> 5;
  const float* W = p.w + (size_t)z * p.w_parity_stride;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  const RowCtx ctx = make_a_ctx(p, m0), bctx = make_b_ctx(W, p.K, n0, p.Cout);
  // two k-tiles in flight in registers: the loads of kt+2 are issued before the MFMAs of kt, the loads of kt+1
  // (issued one iteration earlier) are parked in LDS after them -- two MFMA phases between issue and use
  // The loop body is branch-free around the asynchronous loads: every iteration issues eight loads (the tile index is
  // clamped, so the last two iterations re-read the last tile from L2) and every iteration parks a tile in LDS (the
  // last one parks a copy nobody reads).  With the waits inside `if (more tiles)` branches the compiler copied the
  // in-flight destination registers at the joins BEFORE the s_waitcnt of one branch -- stale data for nk == 1.
  f32x4 ra[2][4], rb[2][4];
  uint32_t ka[2], kb[2];
  ka[0] = load_a<UT>(p, ctx, z, 0, ra[0]); kb[0] = load_b(bctx, 0, p.K, rb[0]);
  ka[1] = load_a<UT>(p, ctx, z, min(1, nk - 1) * BK, ra[1]); kb[1] = load_b(bctx, min(1, nk - 1) * BK, p.K, rb[1]);
  wait_loads<8>(ra[0], rb[0]);
  const bool relu_in = p.relu_in != 0;
  store_tile(smem, ra[0], ka[0], relu_in); store_tile(smem + 16384, rb[0], kb[0]);
  __syncthreads();
  // one iteration: loads of tile kt+2 -> set `l`; MFMAs on LDS buffer `bo`; tile kt+1 (set `s`) -> the other buffer
  auto step = [&](int kt, auto lc, auto sc, int bo) {
    constexpr int l = decltype(lc)::value, sset = decltype(sc)::value;
    if (!(EXP & 1)) {
      const int k2 = min(kt + 2, nk - 1) * BK;
      ka[l] = load_a<UT>(p, ctx, z, k2, ra[l]); kb[l] = load_b(bctx, k2, p.K, rb[l]);
    }
    __builtin_amdgcn_sched_barrier(0);     // nothing that needs tile kt+1's registers may move above the MFMAs
    mma_tile(smem + bo, smem + bo + 16384, wm, wn, fr, fg, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (!(EXP & 2)) {
      if (EXP & 1) wait_loads<0>(ra[sset], rb[sset]); else wait_loads<8>(ra[sset], rb[sset]);    // tile kt+2 stays in flight
      store_tile(smem + (bo ^ 32768), ra[sset], ka[sset], relu_in); store_tile(smem + (bo ^ 32768) + 16384, rb[sset], kb[sset]);
    }
    if (!(EXP & 4)) __syncthreads();
  };
#pragma unroll 1
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, 0);
    if (kt + 1 < nk) step(kt + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, 32768);
  }
  drain_loads(ra[0], rb[0], ra[1], rb[1]);               // the clamped loads of the last iteration
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        ct[(wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * fg) * BN + wn + 32 * j + fr] = acc[i][j][e];
  __syncthreads();
  const int cchunk = (threadIdx.x & 15) * 8;
#pragma unroll 1
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 16 + (threadIdx.x >> 4);
    const int m = m0 + row, n = n0 + cchunk;
    float rgb[3] = {0.f, 0.f, 0.f};
    size_t pix = 0;
    if (m < p.M && n < p.Cout) {
      f32x4 x0 = ld4(ct + row * BN + cchunk), x1 = ld4(ct + row * BN + cchunk + 4);
      if (p.bias) { const f32x4 b0 = ld4(p.bias + n), b1 = ld4(p.bias + n + 4); x0 += b0; x1 += b1; }
      if (p.relu_out) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { x0[i] = fmaxf(x0[i], 0.f); x1[i] = fmaxf(x1[i], 0.f); }
      }
      const int b = m / (p.GH * p.GW);
      const int rem = m - b * (p.GH * p.GW);
      const int y = rem / p.GW, x = rem - y * p.GW;
      const int oy = y * p.out_mul + (z >> 1), ox = x * p.out_mul + (z & 1);   // z = 0 unless transposed
      if (!p.rgb_part) {
        const size_t oidx = (((size_t)b * p.OH + oy) * p.OW + ox) * p.Cout + n;
        if (p.residual) {      // residual block: out = conv(...) + [relu](input)  (vqvae/vqvae_zc.py:110-114, see cogview_hip.h)
          f32x4 r0 = ld4(p.residual + oidx), r1 = ld4(p.residual + oidx + 4);
          if (p.relu_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { r0[i] = fmaxf(r0[i], 0.f); r1[i] = fmaxf(r1[i], 0.f); }
          }
          x0 += r0; x1 += r1;
        }
        float* o = p.out + oidx;
        *reinterpret_cast<f32x4*>(o) = x0;
        *reinterpret_cast<f32x4*>(o + 4) = x1;
      } else {
        // fused final 1x1 convolution (vqvae/vqvae_zc.py:190): this tile's 128 channels of the three output sums; the
        // 134-MB-per-image activation of the last transposed convolution is never written
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const f32x4 w0 = ld4(p.rgb_w + (size_t)c * p.Cout + n), w1 = ld4(p.rgb_w + (size_t)c * p.Cout + n + 4);
          float r = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) { r = fmaf(x0[i], w0[i], r); r = fmaf(x1[i], w1[i], r); }
          rgb[c] = r;
        }
        pix = ((size_t)b * p.OH + oy) * p.OW + ox;
      }
    }
    if (p.rgb_part) {        // the 16 lanes of a row hold its 16 channel groups: fold them, lane 0 of the group stores
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float r = rgb[c];
        r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 8, 64);
        rgb[c] = r;
      }
      if ((threadIdx.x & 15) == 0 && m < p.M)
        *reinterpret_cast<f32x4*>(p.rgb_part + ((size_t)(n0 / BN) * ((size_t)p.B * p.OH * p.OW) + pix) * 4) = f32x4{rgb[0], rgb[1], rgb[2], 0.f};
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Quantize.forward_ (eval branch, vqvae/vqvae_zc.py:41-54): dist = |x|^2 - 2 x E + |E|^2, id = argmax(-dist)
// (first maximum on ties).  GEMM x[M,256] . Et[8192,256]^T with a running (min, index) epilogue: the
// 33.5 MB/image distance map of the reference is never written.  One workgroup = 128 rows x ALL codes.
struct VqArgs { const float* x; const float* et; const float* e2; long long* ids; int M, D, NE; };

__global__ __launch_bounds__(NT) void vq_argmin_kernel(const VqArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float x2s[BM];
  __shared__ float best_v[2][BM];
  __shared__ int best_i[2][BM];
  const int m0 = blockIdx.x * BM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int fr = lane & 31, fg = lane >> 5;
  // |x|^2 per row (2 threads per row, fp32)
  {
    const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
    float s = 0.f;
    if (m0 + row < p.M)
      for (int k = half * (p.D / 2); k < (half + 1) * (p.D / 2); k += 4) {
        const f32x4 v = ld4(p.x + (size_t)(m0 + row) * p.D + k);
        s += v[0] * v[0]; s += v[1] * v[1]; s += v[2] * v[2]; s += v[3] * v[3];
      }
    s += __shfl_xor(s, 1, 64);
    if (half == 0) x2s[row] = s;
  }
  __syncthreads();
  float bv[2][16]; int bi[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) { bv[i][e] = INFINITY; bi[i][e] = 0; }

  const int nk = (p.D + BK - 1) / BK;
  const RowCtx xctx = make_b_ctx(p.x, p.D, m0, p.M);
  for (int n0 = 0; n0 < p.NE; n0 += BN) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 ra[4], rb[4];
    const RowCtx ectx = make_b_ctx(p.et, p.D, n0, p.NE);
    uint32_t ka = load_b(xctx, 0, p.D, ra), kb = load_b(ectx, 0, p.D, rb);
    __syncthreads();
    wait_loads<0>(ra, rb);
    store_tile(smem, ra, ka); store_tile(smem + 16384, rb, kb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      { const int k1 = min(kt + 1, nk - 1) * BK; ka = load_b(xctx, k1, p.D, ra); kb = load_b(ectx, k1, p.D, rb); }   // branch-free, see conv_kernel
      mma_tile(smem + cur * 32768, smem + cur * 32768 + 16384, wm, wn, fr, fg, acc);
      wait_loads<0>(ra, rb);
      store_tile(smem + (cur ^ 1) * 32768, ra, ka); store_tile(smem + (cur ^ 1) * 32768 + 16384, rb, kb);
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + 32 * j + fr;
      const float c2 = col < p.NE ? p.e2[col] : INFINITY;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * fg;
          const float d = (x2s[row] - 2.0f * acc[i][j][e]) + c2;      // same expression order as the reference
          if (d < bv[i][e]) { bv[i][e] = d; bi[i][e] = col; }          // columns visited in increasing order
        }
    }
  }
  // reduce over the 32 lanes that hold different columns of the same row (ties -> smaller index)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = bv[i][e]; int ix = bi[i][e];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64); const int oi = __shfl_xor(ix, o, 64);
        if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
      }
      if (fr == 0) {
        const int row = wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * fg;
        best_v[wave & 1][row] = v; best_i[wave & 1][row] = ix;
      }
    }
  __syncthreads();
  if (threadIdx.x < BM && m0 + threadIdx.x < p.M) {
    const int r = threadIdx.x;
    const float v0 = best_v[0][r], v1 = best_v[1][r];
    const int i0 = best_i[0][r], i1 = best_i[1][r];
    p.ids[m0 + r] = (v1 < v0 || (v1 == v0 && i1 < i0)) ? i1 : i0;
  }
}

// ---------------------------------------------------------------------------------------------------
// layout helpers
// NCHW (c = 3) -> NHWC4 (4th channel zero) : input of encoder conv 1
__global__ void nchw3_to_nhwc4_kernel(const float* in, float* out, int B, int H, int W) {
  const size_t n = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / ((size_t)H * W), px = i % ((size_t)H * W);
    f32x4 v;
    v[0] = in[(b * 3 + 0) * H * W + px]; v[1] = in[(b * 3 + 1) * H * W + px]; v[2] = in[(b * 3 + 2) * H * W + px]; v[3] = 0.f;
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}
// embed_code (vqvae/vqvae_zc.py:95-96): out[pixel][:] = Et[id][:]   (NHWC == the reference's pre-permute layout)
__global__ void embed_code_kernel(const long long* ids, const float* et, float* out, size_t npix, int D, int NE) {
  const int dv = D / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * dv; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i / dv; const int c = (int)(i % dv) * 4;
    long long id = ids[px]; id = id < 0 ? 0 : (id >= NE ? NE - 1 : id);
    *reinterpret_cast<f32x4*>(out + px * D + c) = ld4(et + (size_t)id * D + c);
  }
}
// final 1x1 conv 512 -> 3 (vqvae/vqvae_zc.py:190) on NHWC input, output NCHW, optional de-normalisation
// out = (w . x + b) * scale[c] + shift[c]  (vqvae/api.py:43).  One wave per pixel group; HBM-bound.
__global__ __launch_bounds__(256) void conv1x1_to3_kernel(const float* in, const float* w, const float* bias,
                                                         float* out, size_t npix, int HW, int Cin,
                                                         float s0, float s1, float s2, float t0, float t1, float t2) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t px = wave; px < npix; px += nw) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int c = lane * 4; c < Cin; c += 256) {
      const f32x4 x = ld4(in + px * Cin + c);
      const f32x4 w0 = ld4(w + c), w1 = ld4(w + Cin + c), w2 = ld4(w + 2 * Cin + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) { a0 += x[i] * w0[i]; a1 += x[i] * w1[i]; a2 += x[i] * w2[i]; }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) {
      const size_t b = px / HW, r = px % HW;
      out[(b * 3 + 0) * HW + r] = (a0 + bias[0]) * s0 + t0;
      out[(b * 3 + 1) * HW + r] = (a1 + bias[1]) * s1 + t1;
      out[(b * 3 + 2) * HW + r] = (a2 + bias[2]) * s2 + t2;
    }
  }
}

// out[b][c][pix] = (sum over channel tiles of the fused-projection partial sums + bias[c]) * scale[c] + shift[c]
__global__ __launch_bounds__(256) void rgb_finalize_kernel(const float* part, int ntiles, const float* bias, float* out, size_t npix,
                                                           int HW, float s0, float s1, float s2, float t0, float t1, float t2) {
  for (size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x; px < npix; px += (size_t)gridDim.x * blockDim.x) {
    f32x4 a = ld4(part + px * 4);
    for (int t = 1; t < ntiles; ++t) a += ld4(part + ((size_t)t * npix + px) * 4);
    const size_t b = px / HW, r = px % HW;
    out[(b * 3 + 0) * HW + r] = (a[0] + bias[0]) * s0 + t0;
    out[(b * 3 + 1) * HW + r] = (a[1] + bias[1]) * s1 + t1;
    out[(b * 3 + 2) * HW + r] = (a[2] + bias[2]) * s2 + t2;
  }
}

}  // namespace

extern "C" int cogv_conv2d_nhwc_f32(const cogv_conv_desc* d, void* stream) {
  if (!d || !d->in || !d->w || (!d->out && !d->rgb_partial)) return COGV_ERR_ARG;
  if (d->rgb_partial && (!d->rgb_w || !d->relu || (d->Cout % 128) || (((uintptr_t)d->rgb_w | (uintptr_t)d->rgb_partial) & 15))) return COGV_ERR_ARG;
  if (d->Cin <= 0 || (d->Cin & 3) || (d->Cout & 7) || d->B <= 0) return COGV_ERR_ARG;
  if (((uintptr_t)d->in | (uintptr_t)d->w | (uintptr_t)d->out | (uintptr_t)d->bias) & 15) return COGV_ERR_ARG;
  ConvArgs a;
  a.in = (const float*)d->in; a.w = (const float*)d->w; a.bias = (const float*)d->bias; a.out = (float*)d->out;
  a.B = d->B; a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin; a.Cout = d->Cout; a.relu_out = d->relu;
  a.relu_in = d->relu_in; a.residual = (const float*)d->residual; a.relu_res = d->relu_residual;
  if (a.residual && (d->rgb_partial || ((uintptr_t)a.residual & 15))) return COGV_ERR_ARG;
  a.rgb_w = (const float*)d->rgb_w; a.rgb_part = (float*)d->rgb_partial;
  int nz = 1;
  a.kind = d->kind;
  if (d->kind == COGV_CONV_4X4_S2) {
    if ((d->IH & 1) || (d->IW & 1)) return COGV_ERR_ARG;
    a.GH = a.OH = d->IH / 2; a.GW = a.OW = d->IW / 2; a.in_mul = 2; a.out_mul = 1; a.ntaps = 16;
  } else if (d->kind == COGV_CONV_1X1) {
    a.GH = a.OH = d->IH; a.GW = a.OW = d->IW; a.in_mul = 1; a.out_mul = 1; a.ntaps = 1;
  } else if (d->kind == COGV_CONV_3X3_S1) {
    a.GH = a.OH = d->IH; a.GW = a.OW = d->IW; a.in_mul = 1; a.out_mul = 1; a.ntaps = 9;
  } else if (d->kind == COGV_CONVT_4X4_S2) {
    // out[2y+py] gets taps ky with ky = (py+1) mod 2 (+2):  py=0: ky=1 (iy=y), ky=3 (iy=y-1);  py=1: ky=0 (iy=y+1), ky=2 (iy=y)
    // weights are packed [py*2+px][co][ty*2+tx][ci] with (ty -> ky) = py==0 ? {1,3} : {0,2}, same for x  => dy = py - ty
    a.GH = d->IH; a.GW = d->IW; a.OH = 2 * d->IH; a.OW = 2 * d->IW; a.in_mul = 1; a.out_mul = 2; a.ntaps = 4;
    nz = 4;
  } else return COGV_ERR_UNSUPPORTED;
  a.K = a.ntaps * a.Cin;
  a.M = a.B * a.GH * a.GW;
  a.w_parity_stride = (long long)a.Cout * a.K;
  a.nz = nz;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr = true;
  }
  if ((long long)a.IH * a.IW * a.Cin >= (1ll << 31) || (long long)a.IW * a.Cin >= (1 << 23) || a.IH >= 32768 || a.IW >= 32768) return COGV_ERR_ARG;
  const int mtiles = (a.M + BM - 1) / BM, ntiles = (a.Cout + BN - 1) / BN;
  const long long blocks = 8ll * ((mtiles + 7) / 8) * ntiles * nz;
  if (blocks > 0x7fffffffll) return COGV_ERR_ARG;
  static const int probe = [] { const char* e = getenv("COGV_CONV_EXP"); return e ? atoi(e) : 0; }();
  if (probe && a.Cin % BK == 0) {
    void (*k)(const ConvArgs) = probe == 1 ? conv_kernel<true, 1> : probe == 2 ? conv_kernel<true, 2> : probe == 3 ? conv_kernel<true, 3>
                              : probe == 4 ? conv_kernel<true, 4> : probe == 6 ? conv_kernel<true, 6> : conv_kernel<true, 7>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NT), 65536, reinterpret_cast<hipStream_t>(stream), a);
    return cogv_check_launch();
  }
  if (a.Cin % BK == 0) hipLaunchKernelGGL(conv_kernel<true>, dim3((unsigned)blocks), dim3(NT), 65536, reinterpret_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(conv_kernel<false>, dim3((unsigned)blocks), dim3(NT), 65536, reinterpret_cast<hipStream_t>(stream), a);
  return cogv_check_launch();
}

extern "C" int cogv_rgb_finalize_f32(const float* partial, int ntiles, const float* bias, float* out, int B, int H, int W,
                                     const float* scale3_host, const float* shift3_host, void* stream) {
  if (!partial || !bias || !out || ntiles <= 0 || B <= 0 || H <= 0 || W <= 0 || ((uintptr_t)partial & 15)) return COGV_ERR_ARG;
  const float s[3] = {scale3_host ? scale3_host[0] : 1.f, scale3_host ? scale3_host[1] : 1.f, scale3_host ? scale3_host[2] : 1.f};
  const float t[3] = {shift3_host ? shift3_host[0] : 0.f, shift3_host ? shift3_host[1] : 0.f, shift3_host ? shift3_host[2] : 0.f};
  const size_t npix = (size_t)B * H * W;
  size_t blocks = (npix + 255) / 256; if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(rgb_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partial, ntiles,
                     bias, out, npix, H * W, s[0], s[1], s[2], t[0], t[1], t[2]);
  return cogv_check_launch();
}

extern "C" int cogv_vq_argmin_f32(const float* x, const float* embed_t, const float* embed_sq, int64_t* ids, int M,
                                  int D, int n_embed, void* stream) {
  if (!x || !embed_t || !embed_sq || !ids || M <= 0 || D <= 0 || (D & 7) || n_embed <= 0) return COGV_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)embed_t) & 15) return COGV_ERR_ARG;
  VqArgs a{x, embed_t, embed_sq, reinterpret_cast<long long*>(ids), M, D, n_embed};
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_argmin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr = true; }
  hipLaunchKernelGGL(vq_argmin_kernel, dim3((M + BM - 1) / BM), dim3(NT), 65536, reinterpret_cast<hipStream_t>(stream), a);
  return cogv_check_launch();
}

extern "C" int cogv_nchw3_to_nhwc4_f32(const float* in, float* out, int B, int H, int W, void* stream) {
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || ((uintptr_t)out & 15)) return COGV_ERR_ARG;
  const size_t n = (size_t)B * H * W;
  int g = (int)((n + 255) / 256); if (g > 4096) g = 4096;
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, B, H, W);
  return cogv_check_launch();
}

extern "C" int cogv_embed_code_f32(const int64_t* ids, const float* embed_t, float* out, int64_t npix, int D,
                                   int n_embed, void* stream) {
  if (!ids || !embed_t || !out || npix <= 0 || (D & 3) || (((uintptr_t)embed_t | (uintptr_t)out) & 15)) return COGV_ERR_ARG;
  const size_t n = (size_t)npix * (D / 4);
  int g = (int)((n + 255) / 256); if (g > 4096) g = 4096;
  hipLaunchKernelGGL(embed_code_kernel, dim3(g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(ids), embed_t, out, (size_t)npix, D, n_embed);
  return cogv_check_launch();
}

extern "C" int cogv_conv1x1_to_rgb_f32(const float* in, const float* w, const float* bias, float* out, int B, int H,
                                       int W, int Cin, const float* scale3_host, const float* shift3_host,
                                       void* stream) {
  if (!in || !w || !bias || !out || B <= 0 || (Cin & 3) || (((uintptr_t)in | (uintptr_t)w) & 15)) return COGV_ERR_ARG;
  const float s[3] = {scale3_host ? scale3_host[0] : 1.f, scale3_host ? scale3_host[1] : 1.f, scale3_host ? scale3_host[2] : 1.f};
  const float t[3] = {shift3_host ? shift3_host[0] : 0.f, shift3_host ? shift3_host[1] : 0.f, shift3_host ? shift3_host[2] : 0.f};
  const size_t npix = (size_t)B * H * W;
  size_t blocks = (npix + 3) / 4; if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(conv1x1_to3_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, w,
                     bias, out, npix, H * W, Cin, s[0], s[1], s[2], t[0], t[1], t[2]);
  return cogv_check_launch();
}
