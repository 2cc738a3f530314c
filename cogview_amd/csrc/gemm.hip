// MFMA GEMM family for the CogView GPT hot path (gfx950).
//
//   C[M,N] = epilogue( A_op[M,K] * B_op[N,K]^T )         fp16/bf16 inputs, fp32 accumulate
//
// Replaces the cuBLAS calls reached through F.linear in the reference:
//   forward  Y = X W^T + b         mpu/layers.py:243 (ColumnParallelLinear), :319 (RowParallelLinear),
//                                  model/gpt2_modeling.py:117 (tied logits)          -> transA=0, transB=0
//   dgrad    dX = dY W             autograd of the above                            -> transA=0, transB=1
//   wgrad    dW = dY^T X           autograd of the above                            -> transA=1, transB=1
//
// "trans" means the operand is stored with the contraction index as the SLOW dimension (A stored [K][M], B stored
// [K][N]): dgrad's W and both weight-gradient operands.  No transposed copies exist in HBM.
//
// This file: the host side (dispatch, grouped launch, split-K reduce, C entry points).  The kernels live in the headers
// included below, one per generation -- gemm_gen4.cuh is the one the train step runs; gemm_gen1 / 2 / 3.cuh are the fallbacks
// for shapes it does not take; gemm_lds.cuh holds the LDS images and fragment reads generations 2-4 share; gemv_gen1.cuh the
// first-generation skinny-M kernels (csrc/gemv.hip has the second).
// Four tile kernels, newest first, plus gemv_kernel for M <= 8 (decode steps: a pure HBM stream of the weights);
// dispatch: launch_gemm:
//   generation 4  gemm_w4_kernel     256x256x64 tiles, 4 waves of 128x128 (accumulators fill the AGPR file), software-
//                                    pipelined quarter-steps, LDS-DMA granule ring, persistent with per-XCD work queues,
//                                    up to 16 problems per launch.  Default for M, N >= 256.  Its 2 x 4 instantiations
//                                    compile as separate translation units (-DCOGV_W4_TU=k, build.py).
//   generation 3  gemm_pp64_kernel   the same tile, ring and queues with 8 waves of 128x64 in a ping-pong schedule
//                                    (kernel_variant 9; 5-15 % slower than generation 4: 1.5x the LDS fragment bytes).
//   generation 2  gemm_glds_kernel   256x128x32 tiles, 4 waves, 3-stage LDS-DMA ring, 2 workgroups per CU.
//                                    For M or N < 256 and operands >= 4 GiB.
//   generation 1  gemm_kernel        128x128x64 tiles, register-staged with register transposes.  For K % 64 != 0 and
//                                    other unaligned shapes.
// All share the fused epilogue (epilogue8): bias, GeLU (+ stored pre-activation), dGeLU, dropout, += C, abs-max for
// Sandwich-LN, and (generations 3, 4) the bias-gradient column sums of the output.
#include "common.cuh"
#include "cogview_hip.h"
#include "gemm_shared.cuh"

#include <cstdlib>
#include <type_traits>

#ifndef COGV_EXP
#define COGV_EXP 0     // schedule experiments of tools/probes/{gemm_exp,w4_dev}.py (bit 0: no DMA, 1: no reads, 2: no MFMA, 3: DMA re-reads k-tiles 0..3, 4: clock probe, 6: no epilogue math/stores, 11: no epilogue at all, 12: no barriers, 13: no DMA waits)
#endif

#include "gemm_common.cuh"
#include "gemm_gen1.cuh"
#include "gemm_lds.cuh"
#include "gemm_gen2.cuh"
#include "gemm_gen3.cuh"
#include "gemm_gen4.cuh"
#include "gemv_gen1.cuh"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
  __shared__ float red[16];
  const size_t nvec = (size_t)p.M * (p.N / 8);
  uint32_t amax_pk = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / (p.N / 8));
    const int n = (int)(i % (p.N / 8)) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < p.splitk; ++s) {
      const float* w = p.ws + ((size_t)s * p.M + m) * p.N + n;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(w);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(w + 4);
      v[0] += x0[0]; v[1] += x0[1]; v[2] += x0[2]; v[3] += x0[3];
      v[4] += x1[0]; v[5] += x1[1]; v[6] += x1[2]; v[7] += x1[3];
    }
    amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n, v));
  }
  if (p.flags & COGV_EPI_ABSMAX) {
    const float bm = absmax_pk_block<T>(amax_pk, reinterpret_cast<uint32_t*>(red));
    if (threadIdx.x == 0) atomic_max_nonneg(p.absmax, bm);
  }
}

template <typename T, bool AT, bool BT, int WM, int WN, int MI, int NJ, int BKT>
void launch_glds(GemmArgs& a, hipStream_t st) {
  constexpr int TBM = WM * 32 * MI, TBN = WN * 32 * NJ;
  constexpr int shmem = 3 * (TBM + TBN) * 2 * BKT;
  a.tiles_m = (a.M + TBM - 1) / TBM; a.tiles_n = (a.N + TBN - 1) / TBN;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<T, AT, BT, WM, WN, MI, NJ, BKT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_glds_kernel<T, AT, BT, WM, WN, MI, NJ, BKT>), dim3(a.tiles_m * a.tiles_n, a.splitk),
                     dim3(WM * WN * 64), shmem, st, a);
}
template <typename T, int WM, int WN, int MI, int NJ, int BKT>
void launch_glds_layout(const cogv_gemm_desc* d, GemmArgs& a, hipStream_t st) {
  if (!d->trans_a && !d->trans_b) launch_glds<T, false, false, WM, WN, MI, NJ, BKT>(a, st);
  else if (!d->trans_a && d->trans_b) launch_glds<T, false, true, WM, WN, MI, NJ, BKT>(a, st);
  else if (d->trans_a && d->trans_b) launch_glds<T, true, true, WM, WN, MI, NJ, BKT>(a, st);
  else launch_glds<T, true, false, WM, WN, MI, NJ, BKT>(a, st);
}

int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
}

// Work-queue counters of the persistent kernel: 64 zero-initialised slots of 16 ints per device (8 per-XCD item
// counters + the finished-workgroup count), used round robin (a launch re-arms its slot when it finishes; launches
// on one stream are ordered anyway).  The only device memory this library allocates itself: 4 KiB per GPU, on first use.
int* sched_slot() {
  constexpr int MAX_DEV = 16, SLOTS = 64;
  static int* pool[MAX_DEV] = {};
  static unsigned turn[MAX_DEV] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAX_DEV) return nullptr;
  if (!pool[dev]) {
    if (hipMalloc(reinterpret_cast<void**>(&pool[dev]), SLOTS * 16 * sizeof(int)) != hipSuccess) return nullptr;
    (void)hipMemset(pool[dev], 0, SLOTS * 16 * sizeof(int));
    (void)hipDeviceSynchronize();
  }
  return pool[dev] + 16 * (turn[dev]++ % SLOTS);
}

#ifndef COGV_W4_TU
// CUs the persistent kernels leave free (cogv_gemm_reserve_cus): a collective that runs CONCURRENTLY with a GEMM needs somewhere
// to live -- a generation-3 / 4 workgroup owns its CU's whole register file, so a persistent launch over all CUs keeps RCCL's
// channel workgroups waiting until it ends.
int g_reserved_cus = 0;
int persistent_grid(int items) {
  int n = num_cus() - g_reserved_cus;
  if (n < 8) n = 8;
  return items < n ? items : n;
}

// The generation-4 kernel's eight instantiations (2 dtypes x 4 layouts) are compiled from this same file in eight
// separate translation units (-DCOGV_W4_TU=k, see build.py: they build in parallel); unit k exports cogv_w4_launch_k.
#define W4_DECL(k) extern "C" __attribute__((visibility("hidden"))) int cogv_w4_launch_##k(const void* ga, int grid, void* stream);
W4_DECL(0) W4_DECL(1) W4_DECL(2) W4_DECL(3) W4_DECL(4) W4_DECL(5) W4_DECL(6) W4_DECL(7)
#undef W4_DECL
typedef int (*w4_launch_fn)(const void*, int, void*);
const w4_launch_fn W4_LAUNCH[8] = {cogv_w4_launch_0, cogv_w4_launch_1, cogv_w4_launch_2, cogv_w4_launch_3,
                                   cogv_w4_launch_4, cogv_w4_launch_5, cogv_w4_launch_6, cogv_w4_launch_7};
// unit index: bit 2 = fp16 (else bf16), bits 1..0 = layout: 0 NT (forward), 1 NN (dgrad: B stored [K, N]),
// 2 TN (wgrad: both stored contraction-major), 3 A stored [K, M] only
template <typename T, bool AT, bool BT> constexpr int w4_unit() {
  return (std::is_same<T, f16_t>::value ? 4 : 0) + (AT ? (BT ? 2 : 3) : (BT ? 1 : 0));
}

template <typename T, bool AT, bool BT, int GEN = 3>
int launch_pp64(GroupArgs& ga, hipStream_t st) {
  constexpr int shmem = 2 * 65536;
  ga.sched = sched_slot();
  if (!ga.sched) return COGV_ERR_LAUNCH;
  {   // raster group height (experiments: COGV_GEMM_GROUP_M); 4 rows x 8 columns of tiles per XCD by default
    static const int gm = [] { const char* e = getenv("COGV_GEMM_GROUP_M"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 32 ? v : 4; }();
    ga.group_m = gm;
  }
  ga.item_start[0] = 0;
  for (int i = 0; i < ga.count; ++i) {
    GemmArgs& a = ga.g[i];
    a.tiles_m = (a.M + 255) / 256; a.tiles_n = (a.N + 255) / 256;
    ga.item_start[i + 1] = ga.item_start[i] + a.tiles_m * a.tiles_n * a.splitk;
  }
  for (int i = ga.count; i < MAX_GROUP; ++i) ga.item_start[i + 1] = ga.item_start[ga.count];
  const int num_cu = num_cus();
  const int items = ga.item_start[ga.count];
  if constexpr (GEN == 4) {
    ga.xp_ok = 0; ga.xp_magic_ig = ga.xp_magic_gfull = ga.xp_magic_gtail = 0u;
    const char* xp_env = getenv("COGV_GEMM_XP");          // read per launch: tests and A/B runs switch it inside one process
    const bool xp_enabled = !xp_env || atoi(xp_env) != 0;
    const GemmArgs& a0 = ga.g[0];
    const int nkt = a0.K / 64;
    if (xp_enabled && ga.count == 1 && a0.splitk == 1 && nkt >= 4 && (nkt & 1) == 0 && items > num_cu) {
      // multiplicative forms of the three divisions of the tile order, verified against the dividing form for EVERY item of
      // this geometry (cached: a training step launches a handful of distinct geometries)
      struct Geo { int tm, tn, gm, ok; uint32_t mig, mgf, mgt; };
      static Geo cache[32];
      static int ncache = 0;
      const Geo* hit = nullptr;
      for (int i = 0; i < ncache; ++i)
        if (cache[i].tm == a0.tiles_m && cache[i].tn == a0.tiles_n && cache[i].gm == ga.group_m) { hit = &cache[i]; break; }
      Geo g;
      if (!hit) {
        g.tm = a0.tiles_m; g.tn = a0.tiles_n; g.gm = ga.group_m;
        const uint32_t tail = (uint32_t)(a0.tiles_m % ga.group_m);
        g.mig = w4_magic((uint32_t)(ga.group_m * a0.tiles_n)); g.mgf = w4_magic((uint32_t)ga.group_m); g.mgt = w4_magic(tail);
        g.ok = 1;
        for (uint32_t b = 0; b < (uint32_t)items && g.ok; ++b) {
          uint32_t m1, n1, m2, n2;
          w4_tile_slow(b, (uint32_t)g.tm, (uint32_t)g.tn, (uint32_t)g.gm, m1, n1);
          w4_tile_fast(b, (uint32_t)g.tm, (uint32_t)g.tn, (uint32_t)g.gm, g.mig, g.mgf, g.mgt, m2, n2);
          if (m1 != m2 || n1 != n2) g.ok = 0;
        }
        if (ncache < 32) cache[ncache++] = g;
        hit = &g;
      }
      ga.xp_ok = hit->ok; ga.xp_magic_ig = hit->mig; ga.xp_magic_gfull = hit->mgf; ga.xp_magic_gtail = hit->mgt;
    }
    return W4_LAUNCH[w4_unit<T, AT, BT>()](&ga, persistent_grid(items), st);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp64_kernel<T, AT, BT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_pp64_kernel<T, AT, BT>), dim3(persistent_grid(items)), dim3(512), shmem, st, ga);
    return COGV_OK;
  }
}
template <typename T, int GEN = 3>
int launch_pp64_layout(int trans_a, int trans_b, GroupArgs& ga, hipStream_t st) {
  if (!trans_a && !trans_b) return launch_pp64<T, false, false, GEN>(ga, st);
  if (!trans_a && trans_b) return launch_pp64<T, false, true, GEN>(ga, st);
  if (trans_a && trans_b) return launch_pp64<T, true, true, GEN>(ga, st);
  return launch_pp64<T, true, false, GEN>(ga, st);
}
template <typename T>
void launch_splitk_reduce(GemmArgs& a, hipStream_t st) {
  const size_t nvec = (size_t)a.M * (a.N / 8);
  int blocks = (int)((nvec + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, a);
}

// Second-generation skinny-M kernels (gemv.hip, one translation unit per dtype: 0 bf16, 1 fp16).  COGV_GEMV2=0 keeps the
// first-generation kernels below (A/B runs); a launcher returns COGV_ERR_UNSUPPORTED for a shape it does not take.
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_launch_0(const void* args, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_launch_1(const void* args, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_attn_launch_0(const void* args, const float* partials, int heads, int nsplit, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_attn_launch_1(const void* args, const float* partials, int heads, int nsplit, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_ln_launch_0(const void* args, int stream_f32, void* stream);
extern "C" __attribute__((visibility("hidden"))) int cogv_gemv2_ln_launch_1(const void* args, int stream_f32, void* stream);
inline bool gemv2_enabled() {
  static const bool on = [] { const char* e = getenv("COGV_GEMV2"); return !e || atoi(e) != 0; }();
  return on;
}

template <typename T>
int launch_gemm(const cogv_gemm_desc* d, GemmArgs& a, hipStream_t st) {
  // skinny M (decode steps): the HBM-streaming matrix-vector kernel; takes every fused epilogue except the column sums
  if (a.M <= GEMV_MAX_M && !d->trans_a && !d->trans_b && (a.K & 511) == 0 && (a.N & 7) == 0 && !(d->flags & COGV_EPI_COLSUM) &&
      (a.lda & 7) == 0 && (a.ldb & 7) == 0 && d->kernel_variant == 0) {
    a.splitk = 1;
    if (gemv2_enabled()) {
      const int rc2 = std::is_same<T, f16_t>::value ? cogv_gemv2_launch_1(&a, st) : cogv_gemv2_launch_0(&a, st);
      if (rc2 != COGV_ERR_UNSUPPORTED) return rc2 != COGV_OK ? rc2 : cogv_check_launch();
    }
    hipLaunchKernelGGL((gemv_kernel<T>), dim3(a.N / 8), dim3(256), 0, st, a);
    return cogv_check_launch();
  }
  const bool glds_ok = (a.K % BK) == 0 && a.M >= 64 && a.N >= 64 && (!d->trans_a || (a.M & 7) == 0) &&
                       (!d->trans_b || (a.N & 7) == 0) && d->kernel_variant != 1;
  if ((d->flags & COGV_EPI_COLSUM) && !glds_ok) return COGV_ERR_UNSUPPORTED;
  if (glds_ok) {
    // variant 3: generation 2 -- 256x128x32, 4 waves (128x64 each), two workgroups per CU (M or N < 256, huge operands)
    // variant 9: generation 3 -- 256x256x64 ping-pong (8 waves), persistent, 16x16x32 MFMAs
    // variant 10: generation 4 -- the same tile and ring with 4 waves of 128x128 (a third fewer LDS fragment bytes per
    //            flop: +5-7 % forward / dgrad, +15 % wgrad over generation 3): default whenever the 256 tile slots per
    //            round are filled about as well as variant 3's 512
    // (the intermediate designs -- 256x128x64 / 128x128x32 / 256x256x32 rings, ping-pong on 32-deep tiles -- measured
    //  within +-5 % of variant 3 and were removed; DESIGN.md section 4 keeps the numbers)
    int variant = d->kernel_variant;
    const size_t a_span = (size_t)(d->trans_a ? a.K : a.M) * a.lda * 2, b_span = (size_t)(d->trans_b ? a.K : a.N) * a.ldb * 2;
    const bool v9_ok = a.M >= 256 && a.N >= 256 && a_span < (1ull << 32) && b_span < (1ull << 32);   // 32-bit DMA offsets
    if (variant != 0 && variant != 3 && variant != 9 && variant != 10) variant = 0;
    if ((d->flags & COGV_EPI_COLSUM) && (!v9_ok || (variant != 0 && variant != 9 && variant != 10))) return COGV_ERR_UNSUPPORTED;
    if ((d->flags & COGV_EPI_COLSUM) && variant != 9) variant = 10;
    if ((variant == 9 || variant == 10) && !v9_ok) variant = 0;
    if (variant == 0) {
      variant = 3;
      if (v9_ok) {
        const int cu = num_cus();
        const int i3 = ((a.M + 255) / 256) * ((a.N + 127) / 128) * a.splitk, i9 = ((a.M + 255) / 256) * ((a.N + 255) / 256) * a.splitk;
        const float e3 = (float)i3 / (float)(((i3 + 2 * cu - 1) / (2 * cu)) * 2 * cu);
        const float e9 = (float)i9 / (float)(((i9 + cu - 1) / cu) * cu);
        if (e9 * 1.1f >= e3) variant = 10;     // measured 1.1-1.4x at equal fill (16x16x32 MFMAs: less power per flop)
      }
    }
    if (variant == 3) launch_glds_layout<T, 2, 2, 4, 2, 32>(d, a, st);
    else if (variant == 9 || variant == 10) {                                           // generations 3 / 4 (operands < 4 GiB)
      GroupArgs ga; ga.count = 1; ga.g[0] = a;
      const int rc = variant == 10 ? launch_pp64_layout<T, 4>(d->trans_a, d->trans_b, ga, st)
                                   : launch_pp64_layout<T, 3>(d->trans_a, d->trans_b, ga, st);
      if (rc != COGV_OK) return rc;
      a.tiles_m = ga.g[0].tiles_m; a.tiles_n = ga.g[0].tiles_n;
    }
    else launch_glds_layout<T, 4, 2, 2, 2, 64>(d, a, st);
    if (a.splitk > 1) {
      const size_t nvec = (size_t)a.M * (a.N / 8);
      int blocks = (int)((nvec + 255) / 256); if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, a);
    }
    return cogv_check_launch();
  }
  dim3 grid(a.tiles_m * a.tiles_n, a.splitk), block(NTHREADS);
  const size_t shmem = 65536;
#define LAUNCH(AT_, BT_)                                                                                 \
  do {                                                                                                   \
    static bool attr_set = false;                                                                        \
    if (!attr_set) {                                                                                     \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, AT_, BT_>),                      \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                       \
      attr_set = true;                                                                                   \
    }                                                                                                    \
    hipLaunchKernelGGL((gemm_kernel<T, AT_, BT_>), grid, block, shmem, st, a);                           \
  } while (0)
  if (!d->trans_a && !d->trans_b) LAUNCH(false, false);
  else if (!d->trans_a && d->trans_b) LAUNCH(false, true);
  else if (d->trans_a && d->trans_b) LAUNCH(true, true);
  else LAUNCH(true, false);
#undef LAUNCH
  if (a.splitk > 1) {
    const size_t nvec = (size_t)a.M * (a.N / 8);
    int blocks = (int)((nvec + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, a);
  }
  return cogv_check_launch();
}

}  // namespace

extern "C" size_t cogv_gemm_workspace_bytes(const cogv_gemm_desc* d) {
  if (!d || d->splitk <= 1) return 0;
  return (size_t)d->splitk * (size_t)d->M * (size_t)d->N * sizeof(float);
}

// Heuristic used by the host side: split the contraction when the output has too few tiles to fill 256 CUs
// (weight-gradient GEMMs of the 336M config: 64..256 tiles, contraction = b*1088 tokens).
extern "C" int cogv_gemm_colsum_rows(int M) { return 2 * ((M + 255) / 256); }

extern "C" int cogv_gemm_pick_splitk(int M, int N, int K) {
  // 256x256 tiles on one persistent workgroup per CU.  Pick the split that fills whole rounds best; every split
  // costs an fp32 slab write + read (8 M N bytes at ~4 TB/s) against 2 M N K flops at ~1.1 PFLOP/s: 1100 / K each.
  return cogv_gemm_pick_splitk_tiles(((M + 255) / 256) * ((N + 255) / 256), K);
}

// the same for `tiles` 256x256 output tiles in total (a grouped launch): one split count for all problems
extern "C" int cogv_gemm_pick_splitk_tiles(int tiles, int K) {
  const int nk = (K + BK - 1) / BK;
  const int slots = 256;
  if (tiles >= 4 * slots || nk < 16) return 1;
  const float cost = 1100.f / (float)K;
  int best = 1; float best_score = -1.f;
  for (int s = 1; s <= 16 && nk / s >= 8; ++s) {
    const int items = tiles * s;
    const int rounds = (items + slots - 1) / slots;
    const float eff = (float)items / (float)(rounds * slots);
    const float score = eff / (1.f + (s > 1 ? cost * s : 0.f));
    if (score > best_score) { best_score = score; best = s; }
  }
  return best;
}

static int build_gemm_args(const cogv_gemm_desc* d, GemmArgs& a) {
  if (!d) return COGV_ERR_ARG;
  if (d->dtype != COGV_F16 && d->dtype != COGV_BF16) return COGV_ERR_UNSUPPORTED;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return COGV_ERR_ARG;
  if ((d->N & 7) || (d->ldc & 7)) return COGV_ERR_ARG;
  if ((d->lda & 7) || (d->ldb & 7)) return COGV_ERR_ARG;
  if (!d->trans_a && (d->K & 7)) return COGV_ERR_ARG;   // K-contiguous operands are read in 16-byte chunks
  if (!d->trans_b && (d->K & 7)) return COGV_ERR_ARG;
  if (d->trans_a && (d->M & 7)) return COGV_ERR_ARG;
  if (((uintptr_t)d->A | (uintptr_t)d->B | (uintptr_t)d->C) & 15) return COGV_ERR_ARG;
  if ((d->flags & COGV_EPI_BIAS) && (!d->bias || ((uintptr_t)d->bias & 15))) return COGV_ERR_ARG;
  if ((d->flags & (COGV_EPI_DGELU | COGV_EPI_MULAUX)) && !d->aux) return COGV_ERR_ARG;
  if ((d->flags & COGV_EPI_DGELU) && (d->flags & COGV_EPI_MULAUX)) return COGV_ERR_ARG;      // one aux operand
  if ((d->flags & COGV_EPI_GELU_DAUX) && !(d->flags & COGV_EPI_GELU)) return COGV_ERR_ARG;
  if ((d->flags & (COGV_EPI_DGELU | COGV_EPI_GELU | COGV_EPI_MULAUX)) && d->aux && ((d->ldaux & 7) || ((uintptr_t)d->aux & 15)))
    return COGV_ERR_ARG;
  if ((d->flags & COGV_EPI_ABSMAX) && !d->absmax) return COGV_ERR_ARG;
  if ((d->flags & COGV_EPI_DROPOUT) && !(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return COGV_ERR_ARG;
  if (d->dropout_row0 < 0) return COGV_ERR_ARG;

  a.A = d->A; a.B = d->B; a.C = d->C;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.bias = d->bias; a.aux = d->aux; a.ldaux = d->ldaux; a.absmax = d->absmax;
  a.flags = d->flags; a.out_f32 = d->out_f32;
  a.seed = d->seed; a.stream_id = d->stream_id;
  a.drop_c0 = ((uint64_t)d->dropout_row0 * (uint64_t)d->N) >> 3;       // N % 8 == 0 (checked above)
  a.thr16 = (d->flags & COGV_EPI_DROPOUT) ? (uint32_t)(d->dropout_p * 65536.0f + 0.5f) : 0u;
  a.keep_scale = 65536.0f / (65536.0f - (float)a.thr16);
  a.tiles_m = (d->M + BM - 1) / BM; a.tiles_n = (d->N + BN - 1) / BN;
  const int nk = (d->K + BK - 1) / BK;
  a.splitk = d->splitk > 1 ? d->splitk : 1;
  if (a.splitk > nk) a.splitk = nk;
  a.ktiles_per_split = (nk + a.splitk - 1) / a.splitk;
  a.splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;   // no empty splits
  a.colsum_ws = d->colsum_partial;
  if ((d->flags & COGV_EPI_COLSUM) && (!d->colsum_partial || ((uintptr_t)d->colsum_partial & 15) || d->splitk > 1 || d->out_f32)) return COGV_ERR_ARG;
  a.ws = reinterpret_cast<float*>(d->workspace);
  if (a.splitk > 1) {
    if (!a.ws || d->workspace_bytes < (size_t)a.splitk * a.M * a.N * sizeof(float)) return COGV_ERR_ARG;
    if ((uintptr_t)a.ws & 15) return COGV_ERR_ARG;
  }
  return COGV_OK;
}

extern "C" int cogv_gemm_reserve_cus(int n) {
  const int prev = g_reserved_cus;
  if (n >= 0) g_reserved_cus = n;
  return prev;
}

extern "C" int cogv_gemm(const cogv_gemm_desc* d, void* stream) {
  GemmArgs a;
  const int rc = build_gemm_args(d, a);
  if (rc != COGV_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == COGV_F16) return launch_gemm<f16_t>(d, a, st);
  return launch_gemm<bf16_t>(d, a, st);
}

// y = epilogue(LN_pre([res + LN_post(z)]) . B^T): the decode step's GEMV with its LayerNorms as prologue (gemv_ln_kernel).
extern "C" int cogv_gemv_ln(const cogv_gemm_desc* d, const cogv_ln_prologue* ln, void* stream) {
  if (!d || !ln) return COGV_ERR_ARG;
  GemvLnArgs a;
  const int rc = build_gemm_args(d, a.g);
  if (rc != COGV_OK) return rc;
  if (d->trans_a || d->trans_b || a.g.M > GEMV_MAX_M || (a.g.K & 511) || a.g.K > 4096 || (a.g.N & 7) || (a.g.ldb & 7)) return COGV_ERR_UNSUPPORTED;
  if (d->flags & (COGV_EPI_COLSUM | COGV_EPI_ACCUM | COGV_EPI_DGELU | COGV_EPI_MULAUX | COGV_EPI_DROPOUT) || d->out_f32 || d->splitk > 1) return COGV_ERR_UNSUPPORTED;
  if (!ln->z || !ln->gamma || !ln->beta) return COGV_ERR_ARG;
  if (ln->gamma_post && (!ln->beta_post || !ln->residual)) return COGV_ERR_ARG;
  if (((uintptr_t)ln->z | (uintptr_t)ln->gamma | (uintptr_t)ln->beta | (uintptr_t)ln->gamma_post | (uintptr_t)ln->beta_post |
       (uintptr_t)ln->residual | (uintptr_t)ln->t_out) & 15) return COGV_ERR_ARG;
  a.z = ln->z; a.z_absmax = ln->z_absmax; a.gamma_p = ln->gamma_post; a.beta_p = ln->beta_post; a.res = ln->residual;
  a.t_out = ln->t_out; a.gamma = ln->gamma; a.beta = ln->beta; a.eps = ln->eps;
  const bool sf = ln->stream_f32 != 0;
  a.g.splitk = 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (gemv2_enabled()) {
    const int rc2 = d->dtype == COGV_F16 ? cogv_gemv2_ln_launch_1(&a, sf ? 1 : 0, st) : cogv_gemv2_ln_launch_0(&a, sf ? 1 : 0, st);
    if (rc2 != COGV_ERR_UNSUPPORTED) return rc2 != COGV_OK ? rc2 : cogv_check_launch();
  }
  const int mt = a.g.M <= 1 ? 1 : a.g.M <= 2 ? 2 : a.g.M <= 4 ? 4 : 8;
  const int shmem = mt * a.g.K * 2;                 // rows [M, mt) are written as zeros-normalised junk nobody reads
  const dim3 grid(a.g.N / 8), block(256);
#define GEMV_LN_LAUNCH(T_, MT_)                                                                                          \
  do {                                                                                                                   \
    static bool attr = false;                                                                                            \
    if (!attr) {                                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_ln_kernel<T_, MT_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_ln_kernel<T_, MT_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);  \
      attr = true;                                                                                                       \
    }                                                                                                                    \
    if (sf) hipLaunchKernelGGL((gemv_ln_kernel<T_, MT_, true>), grid, block, shmem, st, a);                               \
    else hipLaunchKernelGGL((gemv_ln_kernel<T_, MT_, false>), grid, block, shmem, st, a);                                 \
  } while (0)
  if (d->dtype == COGV_F16) { if (mt == 1) GEMV_LN_LAUNCH(f16_t, 1); else if (mt == 2) GEMV_LN_LAUNCH(f16_t, 2); else if (mt == 4) GEMV_LN_LAUNCH(f16_t, 4); else GEMV_LN_LAUNCH(f16_t, 8); }
  else { if (mt == 1) GEMV_LN_LAUNCH(bf16_t, 1); else if (mt == 2) GEMV_LN_LAUNCH(bf16_t, 2); else if (mt == 4) GEMV_LN_LAUNCH(bf16_t, 4); else GEMV_LN_LAUNCH(bf16_t, 8); }
#undef GEMV_LN_LAUNCH
  return cogv_check_launch();
}

// y = epilogue(att . B^T), att = the combination of cogv_attention_decode's split partials (skip_combine form): gemv_attn_kernel.
extern "C" int cogv_gemv_attn(const cogv_gemm_desc* d, const void* partials, int heads, int capacity, void* stream) {
  if (!d || !partials || heads <= 0 || capacity <= 0 || capacity > 4096 || ((uintptr_t)partials & 15)) return COGV_ERR_ARG;
  cogv_gemm_desc dd = *d;
  dd.A = dd.B;                    // the A operand does not exist: keep build_gemm_args' pointer checks happy
  dd.lda = dd.K;
  GemmArgs a;
  const int rc = build_gemm_args(&dd, a);
  if (rc != COGV_OK) return rc;
  if (d->trans_a || d->trans_b || a.M > GEMV_MAX_M || a.K != heads * 64 || (a.K & 511) || (a.N & 7) || (a.ldb & 7)) return COGV_ERR_UNSUPPORTED;
  if (d->flags & (COGV_EPI_COLSUM | COGV_EPI_ACCUM | COGV_EPI_DGELU | COGV_EPI_MULAUX | COGV_EPI_DROPOUT | COGV_EPI_GELU) || d->out_f32 || d->splitk > 1) return COGV_ERR_UNSUPPORTED;
  a.splitk = 1;
  const int nsplit = (capacity + 127) / 128;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (gemv2_enabled()) {
    const float* pw = reinterpret_cast<const float*>(partials);
    const int rc2 = d->dtype == COGV_F16 ? cogv_gemv2_attn_launch_1(&a, pw, heads, nsplit, st) : cogv_gemv2_attn_launch_0(&a, pw, heads, nsplit, st);
    if (rc2 != COGV_ERR_UNSUPPORTED) return rc2 != COGV_OK ? rc2 : cogv_check_launch();
  }
  if (d->dtype == COGV_F16) hipLaunchKernelGGL((gemv_attn_kernel<f16_t>), dim3(a.N / 8), dim3(256), 0, st, a, reinterpret_cast<const float*>(partials), heads, nsplit);
  else hipLaunchKernelGGL((gemv_attn_kernel<bf16_t>), dim3(a.N / 8), dim3(256), 0, st, a, reinterpret_cast<const float*>(partials), heads, nsplit);
  return cogv_check_launch();
}

// Several GEMMs of the same dtype and layout in one persistent launch of the generation-3 kernel (see GroupArgs).
// Every problem must satisfy that kernel's requirements (M, N >= 256, K % 64 == 0, operands < 4 GiB); otherwise
// COGV_ERR_UNSUPPORTED and the caller issues them one by one.
extern "C" int cogv_gemm_grouped(const cogv_gemm_desc* descs, int count, void* stream) {
  if (!descs || count < 1 || count > MAX_GROUP) return COGV_ERR_ARG;
  GroupArgs ga; ga.count = count;
  for (int i = 0; i < count; ++i) {
    const cogv_gemm_desc* d = descs + i;
    const int rc = build_gemm_args(d, ga.g[i]);
    if (rc != COGV_OK) return rc;
    if (d->dtype != descs[0].dtype || d->trans_a != descs[0].trans_a || d->trans_b != descs[0].trans_b) return COGV_ERR_ARG;
    const GemmArgs& a = ga.g[i];
    const size_t a_span = (size_t)(d->trans_a ? a.K : a.M) * a.lda * 2, b_span = (size_t)(d->trans_b ? a.K : a.N) * a.ldb * 2;
    if ((a.K % BK) || a.M < 256 || a.N < 256 || a_span >= (1ull << 32) || b_span >= (1ull << 32)) return COGV_ERR_UNSUPPORTED;
    if (d->trans_a && (a.M & 7)) return COGV_ERR_UNSUPPORTED;
    if (d->trans_b && (a.N & 7)) return COGV_ERR_UNSUPPORTED;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int lrc = descs[0].dtype == COGV_F16 ? launch_pp64_layout<f16_t, 4>(descs[0].trans_a, descs[0].trans_b, ga, st)
                                             : launch_pp64_layout<bf16_t, 4>(descs[0].trans_a, descs[0].trans_b, ga, st);
  if (lrc != COGV_OK) return lrc;
  for (int i = 0; i < count; ++i)
    if (ga.g[i].splitk > 1) {
      if (descs[0].dtype == COGV_F16) launch_splitk_reduce<f16_t>(ga.g[i], st);
      else launch_splitk_reduce<bf16_t>(ga.g[i], st);
    }
  return cogv_check_launch();
}
#else   // COGV_W4_TU: one instantiation of the generation-4 kernel and its launcher
}  // namespace

#define W4_CAT2(a, b) a##b
#define W4_CAT(a, b) W4_CAT2(a, b)
extern "C" __attribute__((visibility("hidden"))) int W4_CAT(cogv_w4_launch_, COGV_W4_TU)(const void* ga, int grid, void* stream) {
  using T = std::conditional<(COGV_W4_TU & 4) != 0, f16_t, bf16_t>::type;
  constexpr int L = COGV_W4_TU & 3;
  constexpr bool AT = L >= 2, BT = L == 1 || L == 2;
  constexpr int shmem = 2 * 65536 + 4 * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<T, AT, BT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_w4_kernel<T, AT, BT>), dim3(grid), dim3(256), shmem, reinterpret_cast<hipStream_t>(stream),
                     *reinterpret_cast<const GroupArgs*>(ga));
  return COGV_OK;
}
#endif
