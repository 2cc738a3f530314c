// Generation 2 (gemm_glds_kernel): 256x128x32 tiles, 3-stage LDS-DMA ring, two workgroups per CU.  For M or N < 256 and operands >= 4 GiB.
// Part of the GEMM family of csrc/gemm.hip (included there, in this order: common, gen1, lds, gen2, gen3, gen4, gemv_gen1);
// not a stand-alone header.
#pragma once

namespace {

// ---- epilogue shared by the LDS-DMA kernels: the fp32 C tile goes through the (now idle) ring in NH row slabs,
//      then out with coalesced 16-byte accesses through the fused epilogue8.  Call after a workgroup barrier.
template <typename T, int NW, int TBM, int TBN, int MI, int NJ, int RING, typename ACC>
__device__ __forceinline__ void store_c_tile_impl(const GemmArgs& p, ACC (&acc)[MI][NJ], char* smem, int m0, int n0,
                                                  int wm, int wn, int lane, int ksplit) {
  constexpr bool B32 = sizeof(ACC) == 64;            // 32x32 blocks (f32x16) or 16x16 blocks (f32x4)
  constexpr int BLK = B32 ? 32 : 16;
  const int fr = lane & 31, fg = lane >> 5;
  // ---- epilogue: the fp32 C tile goes through the ring's LDS in NH row slabs of SLAB rows (the ring of the
  //      BKT = 32 variants is smaller than the full C tile), then out with coalesced 16-byte accesses
  constexpr int NH0 = (TBM * TBN * 4 + RING - 1) / RING;
  constexpr int NH = NH0 <= 1 ? 1 : NH0 <= 2 ? 2 : 4;         // power of two so that SLAB divides the tile
  constexpr int SLAB = TBM / NH;
  static_assert(SLAB * TBN * 4 <= RING && SLAB % 32 == 0 && MI * BLK * NW * NJ * BLK == TBM * TBN, "C slab does not fit the ring");
  float* ct = reinterpret_cast<float*>(smem);
  uint32_t amax_pk = 0u;
  constexpr int CPR = TBN / 8;                       // 8-column chunks per row
  constexpr int RPP = NW * 64 / CPR;                 // rows per pass
  const int cchunk = (threadIdx.x % CPR) * 8;
#pragma unroll 1
  for (int hs = 0; hs < NH; ++hs) {
    if (hs > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if ((wm + BLK * i) / SLAB == hs) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (B32) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              ct[((wm + 32 * i) % SLAB + (e & 3) + 8 * (e >> 2) + 4 * fg) * TBN + wn + 32 * j + fr] = acc[i][j][e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              ct[((wm + 16 * i) % SLAB + 4 * (lane >> 4) + e) * TBN + wn + 16 * j + (lane & 15)] = acc[i][j][e];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int row = threadIdx.x / CPR; row < SLAB; row += RPP) {
      const int m = m0 + hs * SLAB + row, n = n0 + cchunk;
      if (m < p.M && n < p.N) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(ct + row * TBN + cchunk);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(ct + row * TBN + cchunk + 4);
        if (p.splitk > 1) {
          float* w = p.ws + ((size_t)ksplit * p.M + m) * p.N + n;
          *reinterpret_cast<f32x4*>(w) = x0;
          *reinterpret_cast<f32x4*>(w + 4) = x1;
        } else {
          float v[8];
          v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3];
          v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
          amax_pk = absmax_pk(amax_pk, epilogue8<T>(p, m, n, v));
        }
      }
    }
  }
  if ((p.flags & COGV_EPI_ABSMAX) && p.splitk <= 1) {
    __syncthreads();
    const float bm = absmax_pk_block<T>(amax_pk, reinterpret_cast<uint32_t*>(smem));
    if (threadIdx.x == 0) atomic_max_nonneg(p.absmax, bm);
  }
}

template <typename T, int NW, int TBM, int TBN, int MI, int NJ, int RING>
__device__ __forceinline__ void store_c_tile(const GemmArgs& p, f32x16 (&acc)[MI][NJ], char* smem, int m0, int n0,
                                             int wm, int wn, int fr, int fg, int ksplit) {
  store_c_tile_impl<T, NW, TBM, TBN, MI, NJ, RING>(p, acc, smem, m0, n0, wm, wn, fr + 32 * fg, ksplit);
}
template <typename T, int NW, int TBM, int TBN, int MI, int NJ, int RING>
__device__ __forceinline__ void store_c_tile(const GemmArgs& p, f32x4 (&acc)[MI][NJ], char* smem, int m0, int n0,
                                             int wm, int wn, int lane, int ksplit) {
  store_c_tile_impl<T, NW, TBM, TBN, MI, NJ, RING>(p, acc, smem, m0, n0, wm, wn, lane, ksplit);
}

// waves per SIMD the register allocation must allow: BKT = 64 -> one 8-wave workgroup per CU (2);
// BKT = 32 -> two 8-wave workgroups (4) or three 4-wave workgroups (3) per CU
// Workgroups per CU the register allocation must allow (expressed as waves per SIMD):
//   ring <= 80 KiB (BKT = 32) -> two workgroups per CU (three for the small 128x128 tile), else one.
constexpr int glds_min_waves(int nw, int ring_bytes) {
  return (ring_bytes <= 53 * 1024 ? 3 : ring_bytes <= 80 * 1024 ? 2 : 1) * nw / 4;
}

// per-wave tile = (32*MI) x (32*NJ); workgroup tile = (WM*32*MI) x (WN*32*NJ); WM*WN waves
template <typename T, bool AT, bool BT, int WM, int WN, int MI, int NJ, int BKT>
__global__ __launch_bounds__(WM * WN * 64, glds_min_waves(WM * WN, 3 * (WM * 32 * MI + WN * 32 * NJ) * 2 * BKT))
void gemm_glds_kernel(const GemmArgs p) {
  constexpr int NW = WM * WN, TBM = WM * 32 * MI, TBN = WN * 32 * NJ, NST = 3;
  constexpr int A_BYTES = TBM * 2 * BKT, B_BYTES = TBN * 2 * BKT, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PER = A_BYTES / 1024 / NW, B_PER = B_BYTES / 1024 / NW;      // 1-KiB DMA pieces per wave per k-tile
  constexpr int LPT = A_PER + B_PER;
  constexpr int KS = BKT / 16;                                                 // MFMA k-steps per k-tile
  constexpr int RPP_N = 1024 / (2 * BKT);                                      // natural rows per DMA piece
  static_assert(A_PER >= 1 && B_PER >= 1, "tile too small for the wave count");
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // NST * STAGE

  uint64_t exp_t0 = 0, exp_r0 = 0;
  if (COGV_EXP & 16) { exp_t0 = __builtin_readcyclecounter(); exp_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
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

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (wave / WN) * (32 * MI), wn = (wave % WN) * (32 * NJ);
  const int fr = lane & 31, fg = lane >> 5;
  const int nk_total = p.K / BKT;
  const int kt0 = blockIdx.y * p.ktiles_per_split * (BK / BKT);
  const int nk = min(nk_total, kt0 + p.ktiles_per_split * (BK / BKT)) - kt0;   // >= 1 by construction

  // per-lane DMA source pointers for k-tile 0 of this split, and the per-k-tile byte stride
  const char* srcA[A_PER];
  const char* srcB[B_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int piece = i * NW + wave;
    if (!AT) {
      const int row = piece * RPP_N + lane / (BKT / 8);
      const int c = (lane % (BKT / 8)) ^ nswz<BKT>(row);
      const int gm = min(m0 + row, p.M - 1);
      srcA[i] = reinterpret_cast<const char*>(p.A) + ((size_t)gm * p.lda + (size_t)kt0 * BKT + c * 8) * 2;
    } else {
      constexpr int ROWB = TBM * 2;
      const int off = piece * 1024 + lane * 16;
      const int krow = off / ROWB, pc = (off % ROWB) >> 4;
      const int c = pc ^ ((krow & 3) << 2);
      const int col = min(m0 + c * 8, p.M - 8);
      srcA[i] = reinterpret_cast<const char*>(p.A) + ((size_t)((size_t)kt0 * BKT + krow) * p.lda + col) * 2;
    }
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int piece = i * NW + wave;
    if (!BT) {
      const int row = piece * RPP_N + lane / (BKT / 8);
      const int c = (lane % (BKT / 8)) ^ nswz<BKT>(row);
      const int gn = min(n0 + row, p.N - 1);
      srcB[i] = reinterpret_cast<const char*>(p.B) + ((size_t)gn * p.ldb + (size_t)kt0 * BKT + c * 8) * 2;
    } else {
      constexpr int ROWB = TBN * 2;
      const int off = piece * 1024 + lane * 16;
      const int krow = off / ROWB, pc = (off % ROWB) >> 4;
      const int c = pc ^ ((krow & 3) << 2);
      const int col = min(n0 + c * 8, p.N - 8);
      srcB[i] = reinterpret_cast<const char*>(p.B) + ((size_t)((size_t)kt0 * BKT + krow) * p.ldb + col) * 2;
    }
  }
  const size_t kstrideA = AT ? (size_t)BKT * p.lda * 2 : (size_t)BKT * 2;
  const size_t kstrideB = BT ? (size_t)BKT * p.ldb * 2 : (size_t)BKT * 2;
  // one LDS-DMA instruction: piece idx in [0, LPT): first the A pieces of this wave, then the B pieces
  auto issue_piece = [&](int kt, int st, int idx) {
    char* la = smem + st * STAGE;
    char* lb = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      if (idx == i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcA[i] + kt * kstrideA), (lds_void_t*)(la + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      if (idx == A_PER + i)
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcB[i] + kt * kstrideB), (lds_void_t*)(lb + (i * NW + wave) * 1024), 16, 0, 0);
  };
  auto issue = [&](int kt, int st) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) issue_piece(kt, st, i);
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // LDS byte addresses (stage 0) of this lane's transposing reads
  uint32_t trA[MI], trB[NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i) trA[i] = AT ? tr_addr<TBM * 2>(smem, wm + 32 * i, lane) : 0u;
#pragma unroll
  for (int j = 0; j < NJ; ++j) trB[j] = BT ? tr_addr<TBN * 2>(smem + A_BYTES, wn + 32 * j, lane) : 0u;

  issue(0, 0);
  issue(nk > 1 ? 1 : 0, 1);
  int st = 0;
  constexpr int PPS = (LPT + KS - 1) / KS;    // DMA pieces issued behind each MFMA group
  constexpr int DSR = (AT || BT) ? 0 : MI + NJ;     // compiler-visible LDS reads per k-step
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<LPT>();                         // tile kt has landed (the batch issued last iteration may be in flight)
    __builtin_amdgcn_s_barrier();              // ... for every wave; stage (kt+2)%3 is free again
    // Branch-free body: past the end the prefetch re-reads the last tile into a stage nobody reads any more
    // (keeps the vmcnt bookkeeping uniform and lets the compiler software-pipeline ds_read against MFMA).
    const int kpf = min(kt + 2, nk - 1);
    const int pst = st == 0 ? 2 : st - 1;
    const char* la = smem + st * STAGE;
    const char* lb = la + A_BYTES;
    const uint32_t soff = (uint32_t)(st * STAGE);
    typename HT<T>::v8 fa[2][MI], fb[2][NJ];
    TrRaw ta[MI], tb[NJ];
    u32x4 na[MI], nb[NJ];
    constexpr bool ASM_ALL = AT || BT;         // mixed kernels: every fragment read is asm-issued
    auto fetch = [&](int ks, int buf) {        // fragments of k-step ks -> fa[buf], fb[buf] (asm reads stay raw)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm + 32 * i + fr;
        if (AT) tr_issue<TBM * 2>(trA[i] + soff, ks, ta[i]);
        else if (ASM_ALL) nat_issue((uint32_t)(uintptr_t)la + nat_off<BKT>(row, 2 * ks + fg), na[i]);
        else fa[buf][i] = *reinterpret_cast<const typename HT<T>::v8*>(la + nat_off<BKT>(row, 2 * ks + fg));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = wn + 32 * j + fr;
        if (BT) tr_issue<TBN * 2>(trB[j] + soff, ks, tb[j]);
        else if (ASM_ALL) nat_issue((uint32_t)(uintptr_t)lb + nat_off<BKT>(row, 2 * ks + fg), nb[j]);
        else fb[buf][j] = *reinterpret_cast<const typename HT<T>::v8*>(lb + nat_off<BKT>(row, 2 * ks + fg));
      }
    };
    auto land = [&](int buf) {                 // explicit wait + pack for the asm-issued reads
      if (ASM_ALL) {
        // one wait for everything issued by fetch(); every raw register is an in/out operand of some wait statement
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (AT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[i].lo), "+v"(ta[i].hi) : : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na[i]) : : "memory");
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (BT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[j].lo), "+v"(tb[j].hi) : : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nb[j]) : : "memory");
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (AT) fa[buf][i] = tr_pack<T>(ta[i]); else __builtin_memcpy(&fa[buf][i], &na[i], 16);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (BT) fb[buf][j] = tr_pack<T>(tb[j]); else __builtin_memcpy(&fb[buf][j], &nb[j], 16);
        }
      }
    };
    fetch(0, 0);
    land(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < KS - 1) fetch(ks + 1, nxt);
      if (AT || BT) __builtin_amdgcn_sched_barrier(0);     // keep the read issue ahead of the MFMA group
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = HT<T>::mfma32(fa[cur][i], fb[cur][j], acc[i][j]);
#pragma unroll
      for (int q2 = 0; q2 < PPS; ++q2)
        if (ks * PPS + q2 < LPT) issue_piece(kpf, pst, ks * PPS + q2);
      // pin the issue order inside this group: next fragments first (their LDS latency hides under the
      // MFMAs), DMA pieces between MFMAs.  Masks: 0x100 DS read, 0x008 MFMA, 0x010 VMEM.
      if (ks < KS - 1 && DSR > 0) {
        __builtin_amdgcn_sched_group_barrier(0x100, DSR / 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, DSR - DSR / 2, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
#pragma unroll
      for (int q2 = 0; q2 < MI * NJ - 1; ++q2) {
        if (ks * PPS + q2 < LPT && q2 < PPS) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      if (ks < KS - 1 && (AT || BT)) { __builtin_amdgcn_sched_barrier(0); land(nxt); }
    }
    st = (st == 2) ? 0 : st + 1;
  }
  wait_vmcnt<0>();   // drain the (redundant) tail prefetches before the ring is reused
  __syncthreads();   // every wave is done reading the ring: reuse it for the fp32 C tile

  store_c_tile<T, NW, TBM, TBN, MI, NJ, NST * STAGE>(p, acc, smem, m0, n0, wm, wn, fr, fg, blockIdx.y);
  if ((COGV_EXP & 16) && p.out_f32 && threadIdx.x == 0 && (bid == 0 || bid == nwg - 1)) {
    const uint64_t dt = __builtin_readcyclecounter() - exp_t0, dr = __builtin_amdgcn_s_memrealtime() - exp_r0;
    __syncthreads();
    reinterpret_cast<float*>(p.C)[(size_t)m0 * p.ldc + n0] = 100.f * (float)dt / (float)dr;
  }
}

}  // namespace
