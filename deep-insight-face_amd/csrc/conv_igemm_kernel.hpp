// The implicit-GEMM convolution kernel's definition, included TWICE by conv.hip (no include guard):
//   IGEMM_KERNEL_NAME = conv_igemm_kernel,     IGEMM_BF_MAINLOOP = bf_mainloop_3   (every f32 form + the three-term tier)
//   IGEMM_KERNEL_NAME = conv_igemm_bf2_kernel, IGEMM_BF_MAINLOOP = bf_mainloop_2  (instantiated for the split-bf16 tiles only)
// One kernel holding both split-bf16 mainloops behind a run-time branch shared ONE register allocation: the two-term loop
// (two planes: ~160 live registers) inherited the three-term loop's pressure, its B-fragment prefetch was spilled and every
// reload -- `s_waitcnt vmcnt(0)`, the in-order counter -- waited in the tap loop for the fragments requested behind it.
// (Textual inclusion because hipcc's host pass rejects any compile-time selection of the mainloop inside the kernel's
// `run` lambda: profiles/r04_ablation.txt item 4.)
template <class T, bool PRE, int AM, int BF3 = 0, int LEAN = 0>   // LEAN: 0 general, 1 lean, 2 lean + sub-sampled first output
__global__ __launch_bounds__(T::NT, (BF3 ? 2 : T::MIN_BLOCKS)) void IGEMM_KERNEL_NAME(const ConvArgs a) {
  static_assert(!BF3 || ((AM == 13 || AM == 16) && !PRE), "split-bf16 exists as the B-direct patch kernel only");
  constexpr int AMP = AM % 10;                              // AM >= 10: the B-direct form of patch path AM - 10
  constexpr bool PATCH = (AMP == 3 || AMP == 5 || AMP == 6), BD = AM >= 10;
  static_assert(!BD || PATCH, "B-direct exists for the patch paths only");
  constexpr bool B3P = BF3 && PATCH;                         // split-bf16 patch kernel: gemm_mainloop_patch_bf3
  static_assert(!PATCH || !PRE, "patch path: no pre-activation");
  static_assert(!PATCH || (B3P ? (BD && (AMP == 3 || AMP == 6)) : (T::BM == 64 && T::BN == 64)), "patch path tiles");
  constexpr int WM = T::WM, WN = T::WN;
  constexpr int SLAB = T::BM * T::BN;    // floats per partial-accumulator slab
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_timeout = reinterpret_cast<int*>(smem);   // LDS is free between mainloops; no static __shared__ (G17)
  const int tid = threadIdx.x;
  const int P = gridDim.x;
  const int p = xcd_remap(blockIdx.x, P);
  // iteration unit of the stream-K split: a K-step; a 32-channel slice (nine K-steps) for the split-bf16 patch kernel
  const int KS = (BF3 && AM >= 10) ? a.Kpad / (BK * 9) : a.Kpad / BK;
  const int tiles_n = (a.Cout + T::BN - 1) / T::BN;
  const int tiles_m = (a.M + T::BM - 1) / T::BM;
  const int I = tiles_m * tiles_n * KS;                       // < 2^31 (checked by conv_run)
  // Start of (remapped) block q's share of the iteration space: equal shares (weighting them by the resident slot -- the
  // four co-resident blocks of a CU do not advance at the same rate -- measured no gain); a pure function of the problem.
  auto sk_begin = [&](int q) -> int { return (int)((int64_t)I * q / P); };
  const int beg = sk_begin(p), end = sk_begin(p + 1);

  // development aid: time per phase, summed over the block's tiles (100 MHz ticks)
  unsigned long long tr_main = 0, tr_fix = 0, tr_epi = 0, tr_steps = 0, tr_tiles = 0;
  const unsigned long long tr_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  const unsigned long long tr_c0 = a.trace ? __builtin_amdgcn_s_memtime() : 0;
  int it = beg;
  while (it < end) {
    const unsigned long long tA = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
    int tile, kb, mt, nt;
    a.fd_ks.divmod(it, tile, kb);
    const int left = end - it;
    const int ke = (KS - kb <= left) ? KS : kb + left;
    a.fd_tiles_n.divmod(tile, mt, nt);
    const int m0 = mt * T::BM, n0 = nt * T::BN;

    f32x16 acc[WM][WN];
    zero_acc<T>(acc);
    using ALoadReg = typename std::conditional<AM == 1, ConvPwLoader<T::NA, T::RP, PRE>,
                                               ConvALoader<T::NA, T::RP, PRE, AM == 2 ? 2 : 0>>::type;
    using ALoadGather = ALoadReg;
    using ALoadLin = typename std::conditional<AMP == 3, typename std::conditional<B3P, PatchDma<T, bf3p_emax(T::BM)>, PatchA<T, PATCH_EMAX_S>>::type,   // AM 3 / 5 / 6: halo-resident patch
                                               typename std::conditional<AMP == 5, PatchA<T, PATCH_EMAX_L>, ALoadGather>::type>::type;
    using ALoad2D = typename std::conditional<B3P, PatchDma2D<typename std::conditional<B3P, T, Tile<2, 2, 2, 2>>::type>, PatchA2D<typename std::conditional<B3P, Tile<1, 1>, T>::type>>::type;
    using ALoad = typename std::conditional<AMP == 6, ALoad2D, ALoadLin>::type;
    using BLoadF32 = RowLoader<T::NB, T::RP>;
    using BLoad = typename std::conditional<B3P, NoLoader, BLoadF32>::type;
    ALoad al(a, m0);
    BLoad bl = [&] {
      if constexpr (B3P)
        return BLoad();
      else
        return BLoad(a.w + (int64_t)n0 * a.Kpad, (int64_t)a.Cout - n0, a.Kpad);
    }();
    // this block computes the whole tile: fetch the shortcut tile behind the last K-step
    // (not on the patch path: its prefetch registers leave no room, the shortcut tile would only be spilled)
    const bool whole = !PATCH && kb == 0 && ke == KS;
    EpiRes<T> er;
    auto run = [&](int k0, int k1, bool prefetch_res) {
      if constexpr (B3P)
        IGEMM_BF_MAINLOOP<T>(al, a, n0, k0, k1, reinterpret_cast<char*>(smem), acc);
      else if constexpr (BD)
        gemm_mainloop_patch_bd<T>(al, a, n0, k0, k1, smem, acc, [&] {
          if (prefetch_res && a.res) er.load(a, m0, n0);
        }, [] {}, [] {});
      else if constexpr (PATCH)
        gemm_mainloop_patch<T>(al, bl, k0, k1, smem, acc, [&] {
          if (prefetch_res && a.res) er.load(a, m0, n0);
        });
      else
        gemm_mainloop2<T>(al, bl, k0, k1, smem, acc, [&] {
          if (prefetch_res && a.res) er.load(a, m0, n0);
        });
    };
    run(kb, ke, whole);
    const unsigned long long tB = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
    unsigned long long tC = tB;

    if (kb != 0) {
      // not the owner of this tile: publish the partial accumulators (fragment order, 16 B per lane)
      float* slab = a.sk_slab + (int64_t)p * SLAB;
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
            *reinterpret_cast<f32x4*>(slab + (((m * WN + n) * 4 + q) * T::NT + tid) * 4) = v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.sk_flag + p, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // owner: collect the rest of the K range from the blocks that follow, then finish the tile
      int kdone = ke;
      int q = p;
      while (kdone < KS) {
        ++q;
        const int qb = sk_begin(q), qe = sk_begin(q + 1);   // starts inside this tile
        const int q_kb = qb - tile * KS;
        const int q_len = qe - qb;
        const int q_ke = (KS - q_kb <= q_len) ? KS : q_kb + q_len;
        if (tid == 0) {
          int spins = 0, timeout = a.sk_spin_limit < 0 ? 1 : 0;
          while (!timeout && __hip_atomic_load(a.sk_flag + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.sk_epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > a.sk_spin_limit) {
              timeout = 1;
              break;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          *s_timeout = timeout;
        }
        __syncthreads();
        const int timeout = *s_timeout;
        __syncthreads();
        if (!timeout) {
          const float* slab = a.sk_slab + (int64_t)q * SLAB;
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + (((m * WN + n) * 4 + r4) * T::NT + tid) * 4);
                acc[m][n][4 * r4] += v[0];
                acc[m][n][4 * r4 + 1] += v[1];
                acc[m][n][4 * r4 + 2] += v[2];
                acc[m][n][4 * r4 + 3] += v[3];
              }
        } else {
          // the partner never showed up (not co-resident): compute its K range here instead of
          // waiting for ever -- slower, still correct, and the SAME bits: the range is accumulated
          // from zero, exactly like the partner's slab, and added to this block's partial, which
          // waits in the block's private stash slab meanwhile (sk_slab holds 2 * P slabs)
          float* stash = a.sk_slab + ((int64_t)P + p) * SLAB;
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = {acc[m][n][4 * r4], acc[m][n][4 * r4 + 1], acc[m][n][4 * r4 + 2], acc[m][n][4 * r4 + 3]};
                *reinterpret_cast<f32x4*>(stash + (((m * WN + n) * 4 + r4) * T::NT + tid) * 4) = v;
              }
          zero_acc<T>(acc);
          run(q_kb, q_ke, false);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // own stores are in L2; drop any stale L1 line
#pragma unroll
          for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(stash + (((m * WN + n) * 4 + r4) * T::NT + tid) * 4);
                acc[m][n][4 * r4] = v[0] + acc[m][n][4 * r4];
                acc[m][n][4 * r4 + 1] = v[1] + acc[m][n][4 * r4 + 1];
                acc[m][n][4 * r4 + 2] = v[2] + acc[m][n][4 * r4 + 2];
                acc[m][n][4 * r4 + 3] = v[3] + acc[m][n][4 * r4 + 3];
              }
        }
        kdone = q_ke;
      }
      if (a.trace) tC = __builtin_amdgcn_s_memrealtime();
      // (the split-bf16 kernel's operand rings are dead by now: it has the registers to fetch the shortcut tile at once)
      if constexpr (LEAN == 2)
        conv_epilogue_fast<T, AMP == 6, true>(a, acc, m0, n0, smem, er, whole && a.res != nullptr);
      else if constexpr (B3P || LEAN == 1)                   // (bf3p_applies admits the lean epilogue's cases only)
        conv_epilogue_fast<T, AMP == 6>(a, acc, m0, n0, smem, er, whole && a.res != nullptr);
      else
        // (EpiRes fetches the shortcut rows of a LINEAR tile: the two-sub-tile form fetches them row by row instead)
        conv_epilogue<T, !PATCH || (B3P && AMP != 6), AMP == 6, !BF3>(a, acc, m0, n0, smem, er, whole);
    }
    if (a.trace) {
      const unsigned long long tD = __builtin_amdgcn_s_memrealtime();
      tr_main += tB - tA;
      if (kb != 0) tr_fix += tD - tB; else { tr_fix += tC - tB; tr_epi += tD - tC; }
      tr_steps += ke - kb;
      ++tr_tiles;
    }
    it += ke - kb;
  }
  if (a.trace && tid == 0) {
    unsigned long long* t = a.trace + (size_t)blockIdx.x * 8;
    t[0] = tr_main; t[1] = tr_fix; t[2] = tr_epi; t[3] = tr_steps; t[4] = tr_tiles; t[5] = tr_t0;
    t[6] = __builtin_amdgcn_s_memrealtime();
    // low byte: 1 = conv_igemm_kernel record, 2 = conv_pipe_kernel record; above it: shader-clock cycles of the block
    t[7] = 1 | ((__builtin_amdgcn_s_memtime() - tr_c0) << 8);
  }
}
