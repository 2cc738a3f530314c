// Generation 3 (gemm_pp64_kernel, kernel_variant 9): 256x256x64 tiles, 8 waves of 128x64 in a ping-pong schedule, persistent work queues, grouped launches.
// Part of the GEMM family of csrc/gemm.hip (included there, in this order: common, gen1, lds, gen2, gen3, gen4, gemv_gen1);
// not a stand-alone header.
#pragma once

namespace {

// ---- epilogue of the generation-3 kernel for one wave's 128x64 sub-tile (16x16 accumulator blocks, lane = row
//      l & 15, 4 columns at 4 (l >> 4)): transpose 8 rows at a time through the wave's private 2-KiB LDS strip into
//      "8 lanes x 16 bytes = one 128-byte line per row" order, then epilogue8.  F: compile-time flag mask
//      (-1: runtime flags / fp32 output, -2: split-K partial slab).

template <typename T, int F>
__device__ __forceinline__ void pp64_epilogue(const GemmArgs& pg, f32x4 (&acc)[8][4], float* strip, int m_base, int n_base,
                                              int ksplit, int lane, uint32_t& amax_pk, int colsum_row,
                                              bool land_dma_first = false) {
  const int l15 = lane & 15, kb = lane >> 4;
  if ((COGV_EXP & 2048) && acc[0][0][0] != 12345.f) return;      // probe: no strip transposition either
  // the problem descriptor lives in the kernel-argument segment behind a run-time index: copy what this instance
  // uses into pinned scalar registers once
  GemmArgs p = pg;
  pin_s(p.C); pin_s(p.M); pin_s(p.N); pin_s(p.ldc);
  if (F < 0 || (F & (COGV_EPI_GELU | COGV_EPI_DGELU | COGV_EPI_MULAUX))) { pin_s(p.aux); pin_s(p.ldaux); }
  if (F < 0 || (F & COGV_EPI_DROPOUT)) { pin_s(p.seed); pin_s(p.stream_id); pin_s(p.drop_c0); pin_s(p.thr16); pin_s(p.keep_scale); }
  if (F < 0) { pin_s(p.flags); pin_s(p.out_f32); pin_s(p.bias); }
  if (F == -2) pin_s(p.ws);
  const bool want_cs = (F == -1) ? ((p.flags & COGV_EPI_COLSUM) != 0 && !p.out_f32) : (F >= 0 && (F & COGV_EPI_COLSUM));
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int sr = lane >> 3, sc = lane & 7;           // read side: strip row, 8-column group
  // operands of the element-wise pipeline that live in global memory: bias once (the column group of a lane is
  // the same in every pass), the dGeLU pre-activations / the accumulate target for all 16 passes up front
  constexpr bool PRE_BIAS = F >= 0 && (F & COGV_EPI_BIAS), PRE_AUX = F >= 0 && (F & (COGV_EPI_DGELU | COGV_EPI_MULAUX)), PRE_C = F >= 0 && (F & COGV_EPI_ACCUM);
  u32x4 bias_v = {0u, 0u, 0u, 0u}, aux_v[PRE_AUX ? 16 : 1], c_v[PRE_C ? 16 : 1];
  const int n = n_base + 8 * sc;
  {
    if (PRE_BIAS && n < p.N) bias_v = gload16(reinterpret_cast<const T*>(pg.bias) + n);
    if (PRE_AUX || PRE_C) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int m = m_base + 8 * t + sr;
        const bool ok = m < p.M && n < p.N;
        if (PRE_AUX) aux_v[t] = ok ? gload16_aux(reinterpret_cast<const T*>(p.aux) + (size_t)m * p.ldaux + n) : u32x4{0u, 0u, 0u, 0u};
        if (PRE_C) c_v[t] = ok ? gload16(reinterpret_cast<const T*>(p.C) + (size_t)m * p.ldc + n) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  // Pass t moves the 8 rows 16 (t >> 1) + 8 (t & 1) .. +7 through the strip.  The LDS unit executes one wave's
  // instructions in order, so the writes of pass t + 1 may be issued right behind the reads of pass t: the strip is
  // software-pipelined one pass deep (reads of t + 1 in flight while pass t runs its element-wise chain and store).
  auto put = [&](int t) {
    if ((l15 >> 3) == (t & 1)) {
      const int r = l15 & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(strip + r * 64 + (((4 * j + kb) ^ r) << 2)) = acc[t >> 1][j];
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto get = [&](f32x4& x0, f32x4& x1) {
    x0 = *reinterpret_cast<const f32x4*>(strip + sr * 64 + (((2 * sc) ^ sr) << 2));
    x1 = *reinterpret_cast<const f32x4*>(strip + sr * 64 + (((2 * sc + 1) ^ sr) << 2));
    __builtin_amdgcn_wave_barrier();
  };
  f32x4 xq[2][2];
  put(0);
  get(xq[0][0], xq[0][1]);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    if (t + 1 < 16) {
      put(t + 1);
      get(xq[(t + 1) & 1][0], xq[(t + 1) & 1][1]);
    }
    const f32x4 x0 = xq[t & 1][0], x1 = xq[t & 1][1];
    int m = m_base + 8 * t + sr;
    if ((COGV_EXP & 64) && x0[0] != 12345.f) continue;      // probe: no epilogue
    if (COGV_EXP & 128) m &= 255;                           // probe: all tiles store to the same L2-resident rows
    // (generation-4 kernel) the next item's prologue DMAs, issued in front of this epilogue, are waited for
    // in front of its FIRST store: behind it a vmcnt wait would also have to wait for stores
    if (t == 0 && land_dma_first) wait_vmcnt<0>();
    if (m < p.M && n < p.N) {
      if (F == -2) {                                        // split-K partial: raw fp32 slab
        float* w = p.ws + ((size_t)ksplit * p.M + m) * p.N + n;
        gstore16(w, x0);
        gstore16(w + 4, x1);
      } else {
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        float rv[8];
        amax_pk = absmax_pk(amax_pk, epilogue8<T, (F < 0 ? -1 : F)>(p, m, n, v, PRE_BIAS ? &bias_v : nullptr,
                                                                     PRE_AUX ? &aux_v[t] : nullptr,
                                                                     PRE_C ? &c_v[t] : nullptr, want_cs ? rv : nullptr));
        if (want_cs) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[e] += rv[e];
        }
      }
    }
  }
  if (want_cs) {     // lanes with the same (lane & 7) hold the same 8 columns: fold the 8 strip rows, lanes 0..7 write
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = cs[e];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
      cs[e] = t;
    }
    if (sr == 0 && n < p.N) {
      float* w = pg.colsum_ws + (size_t)colsum_row * p.N + n;
      gstore16(w, f32x4{cs[0], cs[1], cs[2], cs[3]});
      gstore16(w + 4, f32x4{cs[4], cs[5], cs[6], cs[7]});
    }
  }
}

// =====================================================================================================
// Generation-3 kernel: 256x256 tile, 64-deep k-tiles, 8 waves (2 x 4, 128x64 each), ping-pong schedule,
// v_mfma_f32_16x16x32, persistent over (tile, k-split) items of up to four problems.
//
// Why this shape (measured with tools/probes/gemm_exp.py, which compiles the loop with parts removed):
//  * the chip is POWER limited in a dense GEMM: the shader clock falls from 2.4 GHz to 1.3-1.8 GHz as soon as
//    MFMA, LDS reads and LDS-DMA run together, so throughput follows energy per flop.  The 16x16x32 MFMA moves
//    half the accumulator bytes per flop of the 32x32x16 one and measured +15 % on the whole loop;
//  * a K-contiguous operand must arrive as 128-byte row segments (64-deep k-tiles): with 32-deep tiles every
//    L2 request is half a line and the global->LDS stream alone cannot keep up (11.9 vs 18.3 TB/s chip-wide).
// A 64-deep 256x256 k-tile is 64 KiB, so only two fit in LDS -- a whole-tile ring would have prefetch distance
// one.  Instead the k-tile is cut into three granules with different deadlines and the two half-steps of a k-tile
// read DIFFERENT data (quadrant order), so every granule is resident for exactly one READ phase and the prefetch
// distance is three half-steps for all of them:
//
//   granule   content (64 k deep)                      bytes   read in      re-issued (for tile)   needed
//   B         all 256 B rows                           32 KiB  R(2T)        R(2T+1)  (T+2)          R(2T+4)
//   A01       A rows [0,64) u [128,192)                16 KiB  R(2T)        R(2T+1)  (T+2)          R(2T+4)
//   A23       A rows [64,128) u [192,256)              16 KiB  R(2T+1)      R(2T+2)  (T+2)          R(2T+5)
//
//   half-step 2T  : READ  B fragments of the whole k-tile (kept in registers for both half-steps) + A01 fragments,
//                   issue A23(T+1);          MFMA acc[0..3][*] += A01 x B   (32 MFMAs, 2 k-steps of 32)
//   half-step 2T+1: READ  A23 fragments, issue B(T+2), A01(T+2);  MFMA acc[4..7][*] += A23 x B
//
// LDS: 2 buffers x (B 32 KiB | A01 16 KiB | A23 16 KiB) = 128 KiB.  Two barriers per half-step; waves 0-3 and
// 4-7 (one of each per SIMD) run one barrier apart, so one wave of every SIMD is in its MFMA phase while the
// other reads/issues.  Ordering: a granule issued in R(h) replaces data whose last reads were retired
// (lgkmcnt(0)) before every wave's B2(h-1); a granule needed in R(h+1) is certified by every wave's counted vmcnt
// before its B2(h).  Each wave always has exactly 8 DMA instructions issued after the ones it must certify
// (6 + 2), so the wait is vmcnt(8) in both half-steps.
//
// The MFMA operands are SWAPPED (D = B_frag x A_frag), so a lane ends up with 4 consecutive COLUMNS of one
// output row (one ds_write_b128 per 16x16 block), and the epilogue transposes through a private 2-KiB LDS strip
// per wave without any workgroup barrier.  That leaves the ring free after the last READ phase: the NEXT item's
// first 1.75 k-tiles are issued before the epilogue of the current one, so their latency and the draining C
// stores overlap.
template <typename T, bool AT, bool BT>
__global__ __launch_bounds__(512, 2)
void gemm_pp64_kernel(const GroupArgs ga) {
  constexpr int NW = 8, TBM = 256, TBN = 256, KT = 64, KS = 2;
  constexpr int B_OFF = 0, A01_OFF = 32768, A23_OFF = 49152, BUF = 65536;
  constexpr int ROWB_A = 256, ROWB_B = 512;            // k-row bytes of a contraction-strided granule
  extern __shared__ __attribute__((aligned(1024))) char smem[];     // 2 * BUF

  uint64_t exp_t0 = 0, exp_r0 = 0;
  if (COGV_EXP & 16) { exp_t0 = __builtin_readcyclecounter(); exp_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int wm = wr * 128, wn = wc * 64;
  const int l15 = lane & 15, kb = lane >> 4;
  const int nitems = ga.item_start[ga.count];

  // ---- everything that depends on the work item: which problem, which tile, which k range, DMA sources
  struct Item {
    int pi, m0, n0, ksplit, kt0, nk;
    uint32_t offB[4], offA[2][2];      // per-lane byte offsets against a wave-uniform base (SGPR-base DMA form)
  };
  auto setup = [&](int item, Item& it) {
    int pi = 0;
#pragma unroll
    for (int t = 1; t < MAX_GROUP; ++t) pi += (t < ga.count && item >= ga.item_start[t]) ? 1 : 0;
    const GemmArgs& p = ga.g[pi];
    const int local = item - ga.item_start[pi];
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = local % nwg;
    it.pi = pi; it.ksplit = local / nwg;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    constexpr int GROUP_M = 4;
    const int in_group = GROUP_M * p.tiles_n;
    const int group_id = wgid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (wgid % in_group) % gsz;
    const int tile_n = (wgid % in_group) / gsz;
    const int m0 = tile_m * TBM, n0 = tile_n * TBN;
    it.m0 = m0; it.n0 = n0;
    it.kt0 = it.ksplit * p.ktiles_per_split;
    it.nk = min(p.K / KT, it.kt0 + p.ktiles_per_split) - it.kt0;       // >= 1 by construction
    // Piece = one 1-KiB LDS-DMA instruction; wave w owns pieces i*8 + w.  B: 32 pieces (4 per wave); A01, A23: 16 each.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = i * NW + wave;
      if (!BT) {
        const int row = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz(row);
        const int gn = min(n0 + row, p.N - 1);
        it.offB[i] = (uint32_t)(((size_t)gn * p.ldb + c * 8) * 2);
      } else {
        const int off = piece * 1024 + lane * 16;
        const int krow = off / ROWB_B, pc = (off % ROWB_B) >> 4;
        const int c = pc ^ trswz16(krow);
        const int col = min(n0 + c * 8, p.N - 8);
        it.offB[i] = (uint32_t)(((size_t)krow * p.ldb + col) * 2);
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int piece = i * NW + wave;
        if (!AT) {
          const int row = piece * 8 + (lane >> 3);                 // granule row 0..127
          const int c = (lane & 7) ^ swz(row);
          const int tr = (row & 63) + 128 * (row >> 6) + 64 * g;   // tile row
          const int gm = min(m0 + tr, p.M - 1);
          it.offA[g][i] = (uint32_t)(((size_t)gm * p.lda + c * 8) * 2);
        } else {
          const int off = piece * 1024 + lane * 16;
          const int krow = off / ROWB_A, pc = (off % ROWB_A) >> 4;
          const int c = pc ^ trswz16(krow);
          const int gc = c * 8;                                    // granule column 0..127
          const int tcol = (gc & 63) + 128 * (gc >> 6) + 64 * g;   // tile row (= column of the stored A)
          const int col = min(m0 + tcol, p.M - 8);
          it.offA[g][i] = (uint32_t)(((size_t)krow * p.lda + col) * 2);
        }
      }
  };
  auto issue_B = [&](const Item& it, int kt, int buf) {
    if (COGV_EXP & 1) return;
    if (COGV_EXP & 8) kt &= 3;              // re-read the first k-tiles: every request an L2 hit
    const GemmArgs& p = ga.g[it.pi];
    const size_t kstride = BT ? (size_t)KT * p.ldb * 2 : (size_t)KT * 2;
    const char* g = reinterpret_cast<const char*>(p.B) + (size_t)(it.kt0 + kt) * kstride;
    char* l = smem + buf * BUF + B_OFF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (COGV_EXP & 256) {        // probe: the same request stream into VGPRs (discarded) instead of LDS
        u32x4 t; const char* a = g + it.offB[i];
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(a) : "memory");
        continue;
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(g + it.offB[i]), (lds_void_t*)(l + (i * NW + wave) * 1024), 16, 0, 0);
    }
  };
  auto issue_A = [&](const Item& it, int gi, int kt, int buf) {
    if (COGV_EXP & 1) return;
    if (COGV_EXP & 8) kt &= 3;
    const GemmArgs& p = ga.g[it.pi];
    const size_t kstride = AT ? (size_t)KT * p.lda * 2 : (size_t)KT * 2;
    const char* g = reinterpret_cast<const char*>(p.A) + (size_t)(it.kt0 + kt) * kstride;
    char* l = smem + buf * BUF + (gi ? A23_OFF : A01_OFF);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (COGV_EXP & 256) {
        u32x4 t; const char* a = g + it.offA[gi][i];
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(a) : "memory");
        continue;
      }
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(g + it.offA[gi][i]), (lds_void_t*)(l + (i * NW + wave) * 1024), 16, 0, 0);
    }
  };
  // tile 0 complete + B, A01 of tile 1: 14 DMA instructions per wave, in the order the k-loop certifies them
  auto prologue = [&](const Item& it) {
    issue_B(it, 0, 0); issue_A(it, 0, 0, 0);
    issue_A(it, 1, 0, 0);
    const int t1 = min(1, it.nk - 1);
    issue_B(it, t1, 1); issue_A(it, 0, t1, 1);
  };

  // ---- per-lane fragment read addresses (buffer 0, k-step 0).  v_mfma_f32_16x16x32: lane l supplies row
  //      (l & 15) of a 16-row block and the 8 contraction slots of k-block (l >> 4).
  uint32_t adB[4], adA[4];           // adA is relative to the A granule (A01 and A23 share the in-granule layout)
#pragma unroll
  for (int j = 0; j < 4; ++j)
    adB[j] = BT ? tr_addr16<ROWB_B>(smem + B_OFF, wn + 16 * j, lane)
                : (uint32_t)(uintptr_t)(smem + B_OFF) + (uint32_t)((wn + 16 * j + l15) * 128 + ((kb ^ swz(wn + 16 * j + l15)) << 4));
#pragma unroll
  for (int i = 0; i < 4; ++i)
    adA[i] = AT ? tr_addr16<ROWB_A>(smem, wr * 64 + 16 * i, lane)
                : (uint32_t)(uintptr_t)smem + (uint32_t)((wr * 64 + 16 * i + l15) * 128 + ((kb ^ swz(wr * 64 + 16 * i + l15)) << 4));

  // Work distribution: every item, the first one included, comes from an atomic counter.  A static assignment
  // would make the launch as slow as its unluckiest workgroup: this kernel needs a whole CU (512 threads x 256
  // registers), so when other kernels hold CUs -- RCCL's all-reduce channels during the data-parallel backward --
  // some workgroups start late; with the queue they take fewer items, or none and exit at once.
  // One queue per XCD (workgroup b runs on XCD b & 7): item i stays on XCD i & 7, which is what the tile order
  // inside setup() assumes for L2 reuse (one shared queue measured 10-15 % slower).
  __shared__ int s_next;
  const int xq = blockIdx.x & 7;
  if (threadIdx.x == 0) s_next = xq + 8 * atomicAdd(ga.sched + xq, 1);
  __syncthreads();
  Item cur;
  int item = s_next;
  if (item < nitems) { setup(item, cur); prologue(cur); }
#pragma unroll 1
  while (item < nitems) {
    const GemmArgs& p = ga.g[cur.pi];
    const int nk = cur.nk;
    f32x4 acc[8][4];                   // 16x16 blocks of this wave's 128x64: acc[row block][column block]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Outstanding per wave, oldest first: [C stores of the previous item] [B(0) A01(0): 6] [A23(0): 2] [B(1) A01(1): 6].
    // Loads retire in order among loads, so "at most 8 outstanding" means the first 6 have landed (and every store).
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();             // the stagger
    int grabbed = 0;                                       // the item after this one: asked for now, used after the k-loop
    if (threadIdx.x == 0) grabbed = atomicAdd(ga.sched + xq, 1);

    TrRaw tb[KS][4], ta[KS][4];          // KS = 2 k-steps of 32 per k-tile
    u32x4 nb[KS][4], na[KS][4];
    auto read_B = [&](uint32_t boff) {
      if (COGV_EXP & 2) return;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BT) tr_issue16<ROWB_B>(adB[j] + boff, ks, tb[ks][j]);
          else nat_issue((adB[j] + boff) ^ (uint32_t)(ks << 6), nb[ks][j]);
        }
    };
    auto read_A = [&](uint32_t goff) {
      if (COGV_EXP & 2) return;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AT) tr_issue16<ROWB_A>(adA[i] + goff, ks, ta[ks][i]);
          else nat_issue((adA[i] + goff) ^ (uint32_t)(ks << 6), na[ks][i]);
        }
    };
    auto land_A = [&]() {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[ks][i].lo), "+v"(ta[ks][i].hi) : : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na[ks][i]) : : "memory");
        }
    };
    auto land_B = [&]() {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[ks][j].lo), "+v"(tb[ks][j].hi) : : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nb[ks][j]) : : "memory");
        }
    };
    auto mma = [&](int half) {
      if (!(COGV_EXP & 512)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        typename HT<T>::v8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AT) fa[i] = tr_pack<T>(ta[ks][i]); else __builtin_memcpy(&fa[i], &na[ks][i], 16);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BT) fb[j] = tr_pack<T>(tb[ks][j]); else __builtin_memcpy(&fb[j], &nb[ks][j], 16);
        }
        if (COGV_EXP & 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[i]));
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(fb[j]));
          continue;
        }
        // operands swapped: D[n][m] -> lane (m = l & 15) holds columns n = 4 (l >> 4) .. +3 of its row
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[4 * half + i][j] = HT<T>::mfma16(fb[j], fa[i], acc[4 * half + i][j]);
      }
      if (!(COGV_EXP & 512)) __builtin_amdgcn_s_setprio(0);
    };

    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const uint32_t boff = (uint32_t)(buf * BUF);
      // ---------------- half-step 2T
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_B(boff);
      read_A(boff + A01_OFF);
      issue_A(cur, 1, min(kt + 1, nk - 1), buf ^ 1);
      land_B(); land_A();
      wait_vmcnt<8>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      // ---------------- half-step 2T + 1
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_A(boff + A23_OFF);
      { const int t2 = min(kt + 2, nk - 1); issue_B(cur, t2, buf); issue_A(cur, 0, t2, buf); }
      land_A();
      wait_vmcnt<8>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 0) __builtin_amdgcn_s_barrier();              // even out the barrier count
    // Every wave has passed its last READ phase here (group 1's final B2 is the barrier above): the ring is free.
    wait_vmcnt<0>();                                        // the (redundant) tail prefetches of this item
    __builtin_amdgcn_sched_barrier(0);

    // ---- next item's prologue goes out BEFORE this item's epilogue
    if (threadIdx.x == 0) s_next = xq + 8 * grabbed;       // the grabbed-th item of this XCD's list {x, x + 8, ...}
    __syncthreads();
    const int next = s_next;
    const Item done = cur;
    if (next < nitems) { setup(next, cur); prologue(cur); }

    // ---- epilogue.  The accumulators hold, per lane, 4 consecutive columns of row (l & 15) of each 16x16 block.
    //      Each wave transposes its own 128x64 sub-tile through a PRIVATE 2-KiB strip of LDS (8 rows x 64 fp32
    //      columns at a time; the A23 slot of buffer 1, which the next item's prologue does not touch) into
    //      "8 lanes x 16 bytes = one 128-byte line per row" order for the fused epilogue8: no workgroup barrier,
    //      full-line stores.  (Storing straight from the MFMA layout -- 8 bytes per lane, 32-byte row segments --
    //      measured 4x slower than this: 16 us per tile.)  16-byte chunk c of strip row r sits at chunk c ^ r.
    uint32_t amax_pk = 0u;
    float* strip = reinterpret_cast<float*>(smem + BUF + A23_OFF + wave * 2048);
    // One instance per hot flag combination (compile-time mask): the passes below are fully unrolled (the
    // accumulators need static register indices), so a single runtime-flag body is ~100 KB of code per kernel
    // and every item would stream it through the instruction cache.
    constexpr int F_FWD_DROP = COGV_EPI_BIAS | COGV_EPI_DROPOUT | COGV_EPI_ABSMAX, F_FWD_GELU = COGV_EPI_BIAS | COGV_EPI_GELU;
    if (p.splitk > 1) pp64_epilogue<T, -2>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.out_f32) pp64_epilogue<T, -1>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == 0) pp64_epilogue<T, 0>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == COGV_EPI_BIAS) pp64_epilogue<T, COGV_EPI_BIAS>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == COGV_EPI_ACCUM) pp64_epilogue<T, COGV_EPI_ACCUM>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == F_FWD_DROP) pp64_epilogue<T, F_FWD_DROP>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == F_FWD_GELU) pp64_epilogue<T, F_FWD_GELU>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == (COGV_EPI_DGELU | COGV_EPI_COLSUM)) pp64_epilogue<T, COGV_EPI_DGELU | COGV_EPI_COLSUM>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == COGV_EPI_DGELU) pp64_epilogue<T, COGV_EPI_DGELU>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == (F_FWD_GELU | COGV_EPI_GELU_DAUX)) pp64_epilogue<T, F_FWD_GELU | COGV_EPI_GELU_DAUX>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else if (p.flags == (COGV_EPI_MULAUX | COGV_EPI_COLSUM)) pp64_epilogue<T, COGV_EPI_MULAUX | COGV_EPI_COLSUM>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    else pp64_epilogue<T, -1>(p, acc, strip, done.m0 + wm, done.n0 + wn, done.ksplit, lane, amax_pk, (done.m0 >> 7) + wr);
    if ((p.flags & COGV_EPI_ABSMAX) && p.splitk <= 1) {
      uint32_t wv = max(amax_pk & 0xffffu, amax_pk >> 16);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wv = max(wv, (uint32_t)__shfl_xor((int)wv, o, 64));
      if (lane == 0) atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
    if ((COGV_EXP & 16) && p.out_f32 && threadIdx.x == 0) {
      // shader clock in MHz over this workgroup's lifetime so far (s_memrealtime ticks at 100 MHz)
      const uint64_t dt = __builtin_readcyclecounter() - exp_t0, dr = __builtin_amdgcn_s_memrealtime() - exp_r0;
      reinterpret_cast<float*>(p.C)[(size_t)done.m0 * p.ldc + done.n0] = 100.f * (float)dt / (float)dr;
    }
    item = next;
  }
  // the last workgroup to leave re-arms the queue for the next launch (every workgroup has made its last grab by then)
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ga.sched + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t) ga.sched[t] = 0;
      __threadfence();
    }
  }
}

}  // namespace
