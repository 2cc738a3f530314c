// Generation 4 (gemm_w4_kernel, the default): 256x256x64 tiles, 4 waves of 128x128 with the accumulators in the AGPR file, hand-ordered quarter-steps, pipelined epilogue, cross-item prefetch.
// Part of the GEMM family of csrc/gemm.hip (included there, in this order: common, gen1, lds, gen2, gen3, gen4, gemv_gen1);
// not a stand-alone header.
#pragma once

namespace {

// =====================================================================================================
// Generation-4 kernel: the same 256x256x64 tile, persistent queues, LDS-DMA granule ring and epilogue as
// generation 3, but FOUR waves of 128x128 (one per SIMD, accumulators in the 256 AGPRs) instead of eight of
// 128x64.  A 128x128 wave tile reads 256 B of fragments per MFMA instead of 384 B: the GEMM is power limited
// and the probe (tools/probes/gemm_exp.py) puts the fragment reads at ~19 % of the loop's cost, so fewer LDS
// bytes per flop is the lever -- at the price of no second wave per SIMD to hide latency: the next fragments
// are read between the MFMAs of the same wave (software pipeline, order pinned by sched_barrier).
//
// k-tile = 4 granules of 16 KiB (128 rows or columns x 64 k): A01 = tile rows [0,64) u [128,192), A23 = the
// rest, B01 / B23 likewise over the tile's columns; wave (wr, wc) owns granule rows wr*64.. of the A granules
// and wc*64.. of the B granules.  A k-tile is four quarter-steps of 32 MFMAs, each against one A half and one
// B half, ordered so that only ONE half changes between consecutive quarter-steps; its 8 fragments are read
// during the previous quarter-step into the register set that just died (4 sets of 32 registers):
//
//   q0  A01 x B01   reads B23(T)          | half-step 2T  : DMA A01(T+2), B01(T+2)
//   q1  A01 x B23   reads A23(T)          |
//   q2  A23 x B23   reads A01(T+1)        | half-step 2T+1: DMA B23(T+2), A23(T+2)
//   q3  A23 x B01   reads B01(T+1)        |
//
// One barrier per half-step.  The granules read during half-step h were issued in half-step h-3 and are
// certified by every wave's vmcnt(16) (8 DMA instructions per wave per half-step) before the barrier that opens
// h; the granules read during h-1 are free from that barrier on and are re-issued (for two k-tiles later) in h.
// LDS: A granules in [0, 64 KiB) at buffer * 32 KiB + half * 16 KiB, B granules likewise in [64, 128 KiB), then 4 private
// 2-KiB epilogue strips.  Every fragment read is "per-lane register + immediate" (the buffer / granule / k-step part
// fits the 16-bit offset field), so the loop holds 16 address registers instead of one per (buffer, granule, block).
template <int OFF>
__device__ __forceinline__ void nat_issue_o(uint32_t addr, u32x4& o) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(o) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF, int ROWB>
__device__ __forceinline__ void tr_issue16_o(uint32_t addr, TrRaw& o) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(o.lo), "=&v"(o.hi) : "v"(addr), "n"(OFF), "n"(OFF + 4 * ROWB) : "memory");
}
// MFMA with the accumulator pinned to the AGPR file and updated in place.  With all 256 AGPRs holding accumulators the
// register allocator has no slack: left to itself (builtin form) it parks parts of the loop-carried accumulators in
// VGPRs, picks untied destination registers and copies / spills around every MFMA.
template <typename T>
__device__ __forceinline__ void mfma16_inplace(f32x4& c, const typename HT<T>::v8& a, const typename HT<T>::v8& b) {
  if constexpr (std::is_same<T, bf16_t>::value)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// LDS-DMA of 16 bytes per lane in the "SGPR base + 32-bit lane offset" form, LDS destination (wave-uniform) through M0
__device__ __forceinline__ void dma16(const void* base, uint32_t lane_off, uint32_t lds_addr) {
  const uint64_t b64 = (uint64_t)(uintptr_t)base;
  const uint64_t bu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
  base = reinterpret_cast<const void*>((uintptr_t)bu);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(base), "s"(lds_addr) : "memory", "m0");
}
template <int V> using IC = std::integral_constant<int, V>;
template <typename F, int... R>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, R...>) { (f(IC<R>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- epilogue of the generation-4 kernel (round 4): the same 8-rows-per-store strip transposition and epilogue8 chain as
//      pp64_epilogue, but with the LDS traffic issued BY HAND as a pipeline.  The compiler-scheduled form above waits, in
//      every pass, for the pass's own strip writes and for the previous pass's reads before it issues the next reads
//      (s_waitcnt lgkmcnt(3) x 2 in front of the two ds_read_b128, lgkmcnt(3 / 2) in front of the arithmetic): two exposed LDS
//      round trips per 8-row pass, ~310 cycles for ~22 instructions, whatever the shader clock -- measured with in-kernel
//      timestamps (profiles/r04_gemm_item_phase_probe_v1.log): 2.5-3.9 us per 128 x 64 half in the plain / bias forms, 9 % of a
//      K = 2560 item with the matrix pipe idle.  Here a "super-pass" moves a whole 16-row block: ALL lanes write their four
//      16 x 16 blocks (rows 0-7 into the wave's strip A, rows 8-15 into strip B: 4 KiB per wave, no exec-masked half), the
//      four reads of the block follow at once, and the writes + reads of block T + 1 are issued BEFORE block T's registers are
//      awaited with a counted lgkmcnt(8) -- the LDS unit executes a wave's instructions in order, so the reads of T see T's
//      data and the writes of T + 1 cannot overtake them.  Accumulators are written straight from the AGPRs.
__device__ __forceinline__ void w4_strip_put(const uint32_t (&wa)[4], f32x4 (&blk)[4]) {       // four 16 x 16 blocks, AGPR -> LDS
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("ds_write_b128 %0, %1" : : "v"(wa[j]), "a"(blk[j]) : "memory");
}
__device__ __forceinline__ void w4_strip_get(uint32_t ra0, uint32_t ra1, f32x4 (&x)[4]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %5 offset:2048"
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(ra0), "v"(ra1) : "memory");
}
template <int N>
__device__ __forceinline__ void w4_strip_land(f32x4 (&x)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N) : "memory");
}
template <typename T, int F>
__device__ __forceinline__ void w4_epilogue(const GemmArgs& pg, f32x4 (&acc)[8][4], char* strip, int m_base, int n_base,
                                            int ksplit, int lane, uint32_t& amax_pk, int colsum_row, bool land_dma_first) {
  const int l15 = lane & 15, kb = lane >> 4;
  if ((COGV_EXP & 2048) && acc[0][0][0] != 12345.f) return;      // probe: no epilogue at all (a run-time condition the compiler cannot fold)
  GemmArgs p = pg;
  pin_s(p.C); pin_s(p.M); pin_s(p.N); pin_s(p.ldc);
  if (F < 0 || (F & (COGV_EPI_GELU | COGV_EPI_DGELU | COGV_EPI_MULAUX))) { pin_s(p.aux); pin_s(p.ldaux); }
  if (F < 0 || (F & COGV_EPI_DROPOUT)) { pin_s(p.seed); pin_s(p.stream_id); pin_s(p.drop_c0); pin_s(p.thr16); pin_s(p.keep_scale); }
  if (F < 0) { pin_s(p.flags); pin_s(p.out_f32); pin_s(p.bias); }
  if (F == -2) pin_s(p.ws);
  // (Round 6 measured a DIRECT store from the MFMA layout -- each lane packs the 8 values it holds of a row in two neighbouring
  //  16-column blocks, the four lanes of a row write 64 contiguous bytes, no LDS transposition; it would need the DMA / the
  //  transposing reads to permute the B rows so that those 8 values are consecutive columns.  16 rows x 64 B per store
  //  instruction instead of 8 rows x 128 B is 4-24 % SLOWER on every 4B shape and 19-56 % on the 336M ones
  //  (profiles/r06_gemm_direct_store_probe.log): the strip transposition stays.)
  const bool want_cs = (F == -1) ? ((p.flags & COGV_EPI_COLSUM) != 0 && !p.out_f32) : (F >= 0 && (F & COGV_EPI_COLSUM));
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int sr = lane >> 3, sc = lane & 7;           // read side: strip row, 8-column group
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
  // write side: lane (row l15, column quad kb) of block j -> strip (l15 >> 3), row r = l15 & 7, 16-byte chunk (4 j + kb) ^ r
  const uint32_t sbase = (uint32_t)(uintptr_t)strip;
  uint32_t wa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wa[j] = sbase + (uint32_t)((l15 >> 3) * 2048 + (l15 & 7) * 256 + ((((4 * j + kb) ^ (l15 & 7))) << 4));
  const uint32_t ra0 = sbase + (uint32_t)(sr * 256 + (((2 * sc) ^ sr) << 4)), ra1 = sbase + (uint32_t)(sr * 256 + (((2 * sc + 1) ^ sr) << 4));
  f32x4 xq[2][4];                                    // [register set][strip * 2 + {columns 0-3, 4-7}]
  // (F == -1, the run-time-flag instance -- fp32 output, rare flag combinations: its element-wise chain keeps a small
  //  array in scratch, and the build refuses scratch traffic while asm-issued loads are in flight: no read-ahead there)
  constexpr bool AHEAD = F != -1;
  if (AHEAD) {
    w4_strip_put(wa, acc[0]);
    w4_strip_get(ra0, ra1, xq[0]);
  }
#pragma unroll
  for (int TT = 0; TT < 8; ++TT) {
    if (!AHEAD) {
      w4_strip_put(wa, acc[TT]);
      w4_strip_get(ra0, ra1, xq[TT & 1]);
      w4_strip_land<0>(xq[TT & 1]);
    } else if (TT + 1 < 8) {
      w4_strip_put(wa, acc[TT + 1]);
      w4_strip_get(ra0, ra1, xq[(TT + 1) & 1]);
      w4_strip_land<8>(xq[TT & 1]);
    } else {
      w4_strip_land<0>(xq[TT & 1]);
    }
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
      const int t = 2 * TT + hs;
      const f32x4 x0 = xq[TT & 1][2 * hs], x1 = xq[TT & 1][2 * hs + 1];
      int m = m_base + 8 * t + sr;
      if ((COGV_EXP & 64) && x0[0] != 12345.f) continue;      // probe: no epilogue arithmetic / stores
      if (COGV_EXP & 128) m &= 255;                           // probe: all tiles store to the same L2-resident rows
      // the next item's prologue DMAs, issued in front of this epilogue, are waited for in front of its FIRST store: behind
      // it a vmcnt wait would also have to wait for stores
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

// Probe builds only (-DCOGV_W4_TS, tools/probes/w4_ts.py): per-wave wall time (s_memrealtime, 100 MHz) of the phases of an
// item, summed over the wave's items and written over the first 32 KiB of problem 0's C when the workgroup exits -- the
// output of such a build is garbage there by design.  Phases: 0 first wait + barrier, 1 pre-step (queue atomic, first
// fragments), 2 k-loop, 3 drain wait + item hand-over barrier, 4 next item's setup + prologue issue, 5 / 6 the two epilogue
// halves; slot 7 counts items, slots 8 / 9 are the kernel's total in s_memrealtime / s_memtime ticks (-> shader clock).
#if defined(COGV_W4_TS)
#define W4_TS(k_) do { const uint64_t t_ = __builtin_amdgcn_s_memrealtime(); ts_acc[k_] += (uint32_t)(t_ - ts_last); ts_last = t_; } while (0)
#else
#define W4_TS(k_) do { } while (0)
#endif

template <typename T, bool AT, bool BT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_w4_kernel(const GroupArgs ga) {
  constexpr int NW = 4, TBM = 256, TBN = 256, KT = 64;
  constexpr int GRAN = 16384, BUF = 32768, BREG = 65536, ROWB = 256;   // BUF: buffer stride inside an operand's region
  constexpr int G_A01 = 0, G_A23 = 1, G_B01 = 2, G_B23 = 3;
  extern __shared__ __attribute__((aligned(1024))) char smem[];     // 2 * BREG + NW * 4096

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l15 = lane & 15, kb = lane >> 4;
  const int nitems = ga.item_start[ga.count];
  const uint32_t smem_u32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)smem);

  struct Item {
    int pi, m0, n0, ksplit, kt0, nk;
    const char* baseA; const char* baseB;   // wave-uniform: operand + tile origin + the item's first k-tile
    uint32_t off[4][4];                // [granule][piece]: per-lane byte offsets RELATIVE to the tile origin (clamped at the
                                       // matrix edge; every interior tile of a problem has the same 16 values)
  };
  // tile origin and first k-tile of an item (64-bit scalar arithmetic)
  auto set_bases = [&](const GemmArgs& p, Item& it) {
    const size_t ksA = AT ? (size_t)KT * p.lda * 2 : (size_t)KT * 2, ksB = BT ? (size_t)KT * p.ldb * 2 : (size_t)KT * 2;
    it.baseA = reinterpret_cast<const char*>(p.A) + (AT ? (size_t)it.m0 * 2 : (size_t)it.m0 * p.lda * 2) + (size_t)it.kt0 * ksA;
    it.baseB = reinterpret_cast<const char*>(p.B) + (BT ? (size_t)it.n0 * 2 : (size_t)it.n0 * p.ldb * 2) + (size_t)it.kt0 * ksB;
  };
  auto setup = [&](int item, Item& it) {
    int pi = 0;
    if (ga.count > 1) {
#pragma unroll
      for (int t = 1; t < MAX_GROUP; ++t) pi += (t < ga.count && item >= ga.item_start[t]) ? 1 : 0;
    }
    const GemmArgs& p = ga.g[pi];
    const int local = item - ga.item_start[pi];
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = local % nwg;
    it.pi = pi; it.ksplit = local / nwg;
    uint32_t tile_m, tile_n;
    w4_tile_slow((uint32_t)bid, (uint32_t)p.tiles_m, (uint32_t)p.tiles_n, (uint32_t)ga.group_m, tile_m, tile_n);
    // (the integer divisions above run on the VALU: bring the wave-uniform results back to SGPRs, so that the k-loop
    //  counter, the DMA base pointers and the LDS destinations stay scalar)
    it.ksplit = __builtin_amdgcn_readfirstlane(it.ksplit);
    it.m0 = __builtin_amdgcn_readfirstlane((int)tile_m * TBM); it.n0 = __builtin_amdgcn_readfirstlane((int)tile_n * TBN);
    it.kt0 = it.ksplit * p.ktiles_per_split;
    it.nk = min(p.K / KT, it.kt0 + p.ktiles_per_split) - it.kt0;       // >= 1 by construction
    set_bases(p, it);
    // Piece = one 1-KiB LDS-DMA instruction; a granule is 16 pieces, wave w owns pieces i*4 + w.
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const bool isB = gi >= 2;
      const bool trn = isB ? BT : AT;
      const int g = gi & 1;
      const int t0 = isB ? it.n0 : it.m0, lim = isB ? p.N : p.M, ld = isB ? p.ldb : p.lda;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int piece = i * NW + wave;
        if (!trn) {
          const int row = piece * 8 + (lane >> 3);                 // granule row 0..127
          const int c = (lane & 7) ^ swz(row);
          const int tr = (row & 63) + 128 * (row >> 6) + 64 * g;   // tile row / column
          const int gm = min(tr, lim - 1 - t0);                    // relative to the tile origin
          it.off[gi][i] = (uint32_t)(((size_t)gm * ld + c * 8) * 2);
        } else {
          const int off = piece * 1024 + lane * 16;
          const int krow = off / ROWB, pc = (off % ROWB) >> 4;
          const int c = pc ^ trswz16(krow);
          const int gc = c * 8;                                    // granule column 0..127
          const int tcol = (gc & 63) + 128 * (gc >> 6) + 64 * g;
          const int col = min(tcol, lim - 8 - t0);                 // relative to the tile origin
          it.off[gi][i] = (uint32_t)(((size_t)krow * ld + col) * 2);
        }
      }
    }
  };
  // one piece of granule gi from k-tile source `src` (an operand base of an item, already advanced to the k-tile)
  auto dma_from = [&](const char* src, const Item& it, int gi, int i, int buf) {
    if (COGV_EXP & 1) return;
    const bool isB = gi >= 2;
    const uint32_t l = smem_u32 + (uint32_t)((isB ? BREG : 0) + buf * BUF + (gi & 1) * GRAN + (i * NW + wave) * 1024);
    dma16(src, it.off[gi][i], l);
  };
  auto dma = [&](const Item& it, int gi, int i, int kt, int buf) {
    const GemmArgs& p = ga.g[it.pi];
    const bool isB = gi >= 2;
    const bool trn = isB ? BT : AT;
    const int ld = isB ? p.ldb : p.lda;
    const size_t kstride = trn ? (size_t)KT * ld * 2 : (size_t)KT * 2;
    dma_from((isB ? it.baseB : it.baseA) + (size_t)kt * kstride, it, gi, i, buf);
  };
  // two whole k-tiles, in the order the loop certifies them: [A01 B01](0) [B23 A23](0) [A01 B01](1) [B23 A23](1)
  auto prologue = [&](const Item& it) {
    const int t1 = min(1, it.nk - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_A01, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_B01, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_B23, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_A23, i, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_A01, i, t1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_B01, i, t1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_B23, i, t1, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(it, G_A23, i, t1, 1);
  };

  // per-lane fragment addresses inside a granule (k-step 0): v_mfma_f32_16x16x32, lane l supplies row (l & 15)
  // of a 16-row block and the 8 contraction slots of k-block (l >> 4)
  uint32_t adA[2][4], adB[2][4];       // [k-step][block] (a natural region's k-step flips address bit 6: not an add)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wr * 64 + 16 * i, rb = wc * 64 + 16 * i;
    adA[0][i] = AT ? tr_addr16<ROWB>(smem, ra, lane)
                   : (uint32_t)(uintptr_t)smem + (uint32_t)((ra + l15) * 128 + ((kb ^ swz(ra + l15)) << 4));
    adB[0][i] = BT ? tr_addr16<ROWB>(smem + BREG, rb, lane)
                   : (uint32_t)(uintptr_t)(smem + BREG) + (uint32_t)((rb + l15) * 128 + ((kb ^ swz(rb + l15)) << 4));
    adA[1][i] = adA[0][i] ^ 64u; adB[1][i] = adB[0][i] ^ 64u;
  }

  struct Frag { u32x4 n[2][4]; TrRaw t[2][4]; };     // [k-step][16-row block]; one of the two forms is live
  // one fragment (k-step ks, block blk) of granule gi in buffer buf
  auto read1 = [&](Frag& f, auto gic, auto bufc, auto ksc, auto blkc) {
    if (COGV_EXP & 2) return;
    constexpr int gi = decltype(gic)::value, buf = decltype(bufc)::value, ks = decltype(ksc)::value, blk = decltype(blkc)::value;
    constexpr bool isB = gi >= 2;
    constexpr int off = buf * BUF + (gi & 1) * GRAN;
    if (isB ? BT : AT) tr_issue16_o<off + ks * 32 * ROWB, ROWB>(isB ? adB[0][blk] : adA[0][blk], f.t[ks][blk]);
    else nat_issue_o<off>(isB ? adB[ks][blk] : adA[ks][blk], f.n[ks][blk]);
  };
  auto land = [&](Frag& f, bool trn) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (trn) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.t[ks][b].lo), "+v"(f.t[ks][b].hi) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.n[ks][b]) : : "memory");
      }
  };

  // Work items are taken from the XCD's queue TWO ahead (round 4): while item i is computed the index of item i + 1 is
  // already known, so that its first two k-tiles can ride in the DMA slots of item i's last two k-tiles (cross-item
  // prefetch, below); the queue position asked for in item i's pre-step is item i + 2.
  __shared__ int s_next[2];
  const int xq = blockIdx.x & 7;
  if (threadIdx.x == 0) {
    s_next[0] = xq + 8 * atomicAdd(ga.sched + xq, 1);
    s_next[1] = xq + 8 * atomicAdd(ga.sched + xq, 1);
  }
  __syncthreads();
  Item cur;
  bool certified = false;
  int item = __builtin_amdgcn_readfirstlane(s_next[0]);  // wave-uniform by construction: keeps the DMA bases in SGPRs
  int nxt = __builtin_amdgcn_readfirstlane(s_next[1]);
  if (item < nitems) { setup(item, cur); prologue(cur); }
#if defined(COGV_W4_TS)
  uint32_t ts_acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  const uint64_t ts_begin = __builtin_amdgcn_s_memrealtime(), ts_clk0 = __builtin_readcyclecounter();
  uint64_t ts_last = ts_begin;
#endif
#pragma unroll 1
  while (item < nitems) {
    const GemmArgs& p = ga.g[cur.pi];
    const int nk = cur.nk;
    f32x4 acc[2][8][4];                // [column half][16-row block 4a+i][16-column block j]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Frag fA, fI, fB0, fB1;
    // pre-step: the first A01 / B01 fragments (the only exposed LDS latency of the item)
    // `certified`: this item's 32 prologue DMAs were waited for (vmcnt(0)) in front of the previous epilogue's first C store
    // (pp64_epilogue, land_dma_first) -- behind it a vmcnt wait would also wait for stores, the counter retires in order.
    // The first three half-steps then run without vmcnt waits and the stores drain behind their MFMAs.
    // MEASURED (profiles/r03_gemm_peel_ab_v2.log, three alternating runs): +1.2 to +2.1 % on the epilogues without a bias
    // (plain dgrad, dGeLU + column sums), 0 +- 0.5 % on the bias epilogues -- there the compiler's own wait for the bias
    // load (vmcnt(0) in front of the first store) already does the same thing; -1.4 .. +1 % at K = 1024.
    // (Also measured in round 3, removed: issuing the work-queue atomic by hand and reading it after the k-loop -- no
    //  change, profiles/r03_gemm_async_grab_peel_ab_v1.log.)
    if (!certified) wait_vmcnt<24>();
    __builtin_amdgcn_s_barrier();
    W4_TS(0);
    int grabbed = 0;                                       // the item after the next: asked for now, used after the k-loop
    if (threadIdx.x == 0) grabbed = atomicAdd(ga.sched + xq, 1);
    // Cross-item prefetch: the DMA slots of this item's last two k-tiles -- which would re-fetch its own last k-tile --
    // carry the first two k-tiles of the NEXT item instead, and the next item starts with both resident: no set-up
    // arithmetic, no 32-DMA prologue and no exposed first fetch between the items (2.1-2.7 us of a 60-85 us item with the
    // matrix pipe idle, profiles/r04_gemm_item_phase_probe_v1.log), and 2 of 40 k-tiles less L2 traffic at K = 2560.
    // Taken when the launch allows it (ga.xp_ok) and both tiles are interior: every interior tile has the SAME per-lane
    // offsets against its origin, so only two scalar base pointers change -- selected without a branch inside the k-loop.
    Item nx = cur;                                         // scalar fields only (off unused)
    bool xp = false;
    if (ga.xp_ok && nxt < nitems) {
      uint32_t tm, tn;
      w4_tile_fast((uint32_t)nxt, (uint32_t)p.tiles_m, (uint32_t)p.tiles_n, (uint32_t)ga.group_m, ga.xp_magic_ig, ga.xp_magic_gfull,
                   ga.xp_magic_gtail, tm, tn);
      nx.pi = 0; nx.ksplit = 0; nx.kt0 = 0; nx.nk = nk;
      nx.m0 = (int)tm * TBM; nx.n0 = (int)tn * TBN;
      set_bases(p, nx);
      xp = cur.m0 + TBM <= p.M && cur.n0 + TBN <= p.N && nx.m0 + TBM <= p.M && nx.n0 + TBN <= p.N;
    }
    const char* const xA = xp ? nx.baseA : cur.baseA;      // source of the k-tiles past this item's end
    const char* const xB = xp ? nx.baseB : cur.baseB;
    const size_t ksA = AT ? (size_t)KT * p.lda * 2 : (size_t)KT * 2, ksB = BT ? (size_t)KT * p.ldb * 2 : (size_t)KT * 2;
    static_for<8>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      read1(fA, IC<G_A01>{}, IC<0>{}, IC<(r >> 2)>{}, IC<(r & 3)>{});
      read1(fB0, IC<G_B01>{}, IC<0>{}, IC<(r >> 2)>{}, IC<(r & 3)>{});
    });
    land(fA, AT); land(fB0, BT);
    W4_TS(1);

    // one quarter-step: acc[bh][4 ah + i][j] += A(fa) x B(fb) while granule gin of buffer bin is read into fin and
    // granule gd of k-tile ktd is DMA'd into buffer bd
    auto quarter = [&](auto ahc, auto bhc, Frag& fa, Frag& fb, Frag& fin, auto ginc, auto binc, int gd, const char* gsrc, int bd) {
      constexpr int ah = decltype(ahc)::value, bh = decltype(bhc)::value, gin = decltype(ginc)::value;
      static_for<8>([&](auto rc) {
          constexpr int r = decltype(rc)::value, ks = r >> 2, i = r & 3;
          // the 8 incoming fragments go out in front of the first four MFMA groups (two each: >= 256 MFMA cycles
          // before land()), the 4 DMA pieces in front of the last four
          if constexpr (r < 4) {
            read1(fin, ginc, binc, IC<((2 * r) >> 2)>{}, IC<((2 * r) & 3)>{});
            read1(fin, ginc, binc, IC<((2 * r + 1) >> 2)>{}, IC<((2 * r + 1) & 3)>{});
          } else {
            dma_from(gsrc, cur, gd, r - 4, bd);
          }
          typename HT<T>::v8 va;
          if (AT) va = tr_pack<T>(fa.t[ks][i]); else __builtin_memcpy(&va, &fa.n[ks][i], 16);
          if (COGV_EXP & 4) asm volatile("" :: "v"(va));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            typename HT<T>::v8 vb;
            if (BT) vb = tr_pack<T>(fb.t[ks][j]); else __builtin_memcpy(&vb, &fb.n[ks][j], 16);
            if (COGV_EXP & 4) { asm volatile("" :: "v"(vb)); continue; }
            // operands swapped: D[n][m] -> lane (m = l & 15) holds columns n = 4 (l >> 4) .. +3 of its row
            mfma16_inplace<T>(acc[bh][4 * ah + i][j], vb, va);
          }
          __builtin_amdgcn_sched_barrier(0);
      });
      land(fin, gin >= 2 ? BT : AT);
    };
    // one k-tile; fb01 holds B01(kt), fbx is free and ends up holding B01(kt + 1)
    auto tile = [&](int kt, auto bufc, Frag& fb01, Frag& fbx, auto w0c, auto w1c) {
      constexpr int buf = decltype(bufc)::value;
      constexpr bool W0 = decltype(w0c)::value != 0, W1 = decltype(w1c)::value != 0;   // vmcnt waits of the two half-steps
      // k-tile kt + 2 of this item; past its end: k-tile kt + 2 - nk of the next item (cross-item prefetch) or, without it,
      // this item's last k-tile once more.  Scalar selects, no branch.
      const bool past = kt + 2 >= nk;
      const int ktd = past ? (xp ? kt + 2 - nk : nk - 1) : kt + 2;
      const char* const gA = (past ? xA : cur.baseA) + (size_t)ktd * ksA;
      const char* const gB = (past ? xB : cur.baseB) + (size_t)ktd * ksB;
      if (W0 && !(COGV_EXP & 8192)) wait_vmcnt<16>();
      __builtin_amdgcn_sched_barrier(0);
      if (!(COGV_EXP & 4096)) __builtin_amdgcn_s_barrier();      // probes: results are garbage without them
      __builtin_amdgcn_sched_barrier(0);
      quarter(IC<0>{}, IC<0>{}, fA, fb01, fbx, IC<G_B23>{}, IC<buf>{}, G_A01, gA, buf);
      quarter(IC<0>{}, IC<1>{}, fA, fbx, fI, IC<G_A23>{}, IC<buf>{}, G_B01, gB, buf);
      if (W1 && !(COGV_EXP & 8192)) wait_vmcnt<16>();
      __builtin_amdgcn_sched_barrier(0);
      if (!(COGV_EXP & 4096)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      quarter(IC<1>{}, IC<1>{}, fI, fbx, fA, IC<G_A01>{}, IC<(buf ^ 1)>{}, G_B23, gB, buf);
      quarter(IC<1>{}, IC<0>{}, fI, fb01, fbx, IC<G_B01>{}, IC<(buf ^ 1)>{}, G_A23, gA, buf);
    };
    int kt = 0;
    // half-steps 0..2 read the prologue's granules (all certified above); half-step 3 reads what half-step 0 issued:
    // its vmcnt(16) is the first wait that also covers the previous item's C stores, >= 2 us of MFMA work after the last one
    if (certified && nk >= 2) {
      tile(0, IC<0>{}, fB0, fB1, IC<0>{}, IC<0>{});
      tile(1, IC<1>{}, fB1, fB0, IC<0>{}, IC<1>{});
      kt = 2;
    }
    for (; kt + 1 < nk; kt += 2) {                         // two k-tiles per trip: the B register sets swap roles
      tile(kt, IC<0>{}, fB0, fB1, IC<1>{}, IC<1>{});
      tile(kt + 1, IC<1>{}, fB1, fB0, IC<1>{}, IC<1>{});
    }
    if (nk & 1) tile(nk - 1, IC<0>{}, fB0, fB1, IC<1>{}, IC<1>{});
    __builtin_amdgcn_sched_barrier(0);
    W4_TS(2);
    wait_vmcnt<0>();                                        // the tail prefetches: the next item's first two k-tiles (or redundant)

    // ---- without the cross-item prefetch the next item's prologue goes out BEFORE this item's epilogue (the barrier also
    //      retires every wave's last reads)
    if (threadIdx.x == 0) s_next[0] = xq + 8 * (int)grabbed;
    __syncthreads();
    const int after = __builtin_amdgcn_readfirstlane(s_next[0]);
    const int next = nxt;
    W4_TS(3);
    const Item done = cur;
    bool land_first = false;
    if (xp) {              // both k-tiles resident (vmcnt(0) above, barrier): only the scalar fields change
      cur.m0 = nx.m0; cur.n0 = nx.n0; cur.baseA = nx.baseA; cur.baseB = nx.baseB;
      certified = true;
    } else {
      if (next < nitems) { setup(next, cur); prologue(cur); }
      certified = next < nitems;
      land_first = certified;
    }
    W4_TS(4);

    uint32_t amax_pk = 0u;
    char* strip = smem + 2 * BREG + wave * 4096;
    constexpr int F_FWD_DROP = COGV_EPI_BIAS | COGV_EPI_DROPOUT | COGV_EPI_ABSMAX, F_FWD_GELU = COGV_EPI_BIAS | COGV_EPI_GELU;
    const int mb = done.m0 + wr * 128, nb = done.n0 + wc * 128, csr = (done.m0 >> 7) + wr;
    // transposition through the wave's LDS strip.  (Round 3 also measured a register-exchange form -- v_permlane16_swap,
    // no LDS: equal on plain / bias epilogues, +1.5 % GeLU, -2.3 % column sums, profiles/r03_gemm_swap_epilogue_ab.log; removed.)
#define W4_EPI(F_)                                                                                   \
  do {                                                                                               \
    w4_epilogue<T, F_>(p, acc[0], strip, mb, nb, done.ksplit, lane, amax_pk, csr, land_first);       \
    W4_TS(5);                                                                                        \
    w4_epilogue<T, F_>(p, acc[1], strip, mb, nb + 64, done.ksplit, lane, amax_pk, csr, false);       \
    W4_TS(6);                                                                                        \
  } while (0)
    if (p.splitk > 1) W4_EPI(-2);
    else if (p.out_f32) W4_EPI(-1);
    else if (p.flags == 0) W4_EPI(0);
    else if (p.flags == COGV_EPI_BIAS) W4_EPI(COGV_EPI_BIAS);
    else if (p.flags == COGV_EPI_ACCUM) W4_EPI(COGV_EPI_ACCUM);
    else if (p.flags == F_FWD_DROP) W4_EPI(F_FWD_DROP);
    else if (p.flags == F_FWD_GELU) W4_EPI(F_FWD_GELU);
    else if (p.flags == (COGV_EPI_DGELU | COGV_EPI_COLSUM)) W4_EPI(COGV_EPI_DGELU | COGV_EPI_COLSUM);
    else if (p.flags == COGV_EPI_DGELU) W4_EPI(COGV_EPI_DGELU);
    else if (p.flags == (F_FWD_GELU | COGV_EPI_GELU_DAUX)) W4_EPI(F_FWD_GELU | COGV_EPI_GELU_DAUX);
    else if (p.flags == (COGV_EPI_MULAUX | COGV_EPI_COLSUM)) W4_EPI(COGV_EPI_MULAUX | COGV_EPI_COLSUM);
    else W4_EPI(-1);
#undef W4_EPI
    if ((p.flags & COGV_EPI_ABSMAX) && p.splitk <= 1) {
      uint32_t wv = max(amax_pk & 0xffffu, amax_pk >> 16);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wv = max(wv, (uint32_t)__shfl_xor((int)wv, o, 64));
      if (lane == 0) atomic_max_nonneg(p.absmax, bits_to_f<T>((uint16_t)wv));
    }
    item = next;
    nxt = after;
#if defined(COGV_W4_TS)
    ts_acc[7] += 1u;
#endif
  }
#if defined(COGV_W4_TS)
  if (lane == 0) {
    uint32_t* o = reinterpret_cast<uint32_t*>(ga.g[0].C) + ((size_t)blockIdx.x * NW + wave) * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = ts_acc[k];
    o[8] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - ts_begin);
    o[9] = (uint32_t)(__builtin_readcyclecounter() - ts_clk0);
  }
#endif
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
