// Small-batch convolution: split-K partial GEMM + a reduce / epilogue launch (round 5).  Included by conv.hip.
//
// The reference embeds ONE image per call (predictions.py:152-156) and evaluates at batch 12 (scripts/insight_face.py:112).
// At those sizes a layer has a handful of 64 x 64 output tiles with a long K loop -- IResNet-100's 14 x 14 stage at batch 1:
// 16 tiles of 72 K-steps, its 7 x 7 stage 8 tiles of 144 -- and the chip has 1024 SIMDs.  Round 4 ran them on the persistent
// stream-K grid at P = all resident slots: shares of ~1 K-step, and ONE owner block per tile collecting up to 63 partial slabs
// one after the other (poll, acquire fence, barrier, 16 KB read, add: ~2 us each) -- 35 us per launch for 1.5 us of matrix
// work (profiles/r05_iresnet100_b1_layers_before.txt).  A kernel boundary costs 1.5 us (MI355X_MICROARCH.md, price list
// "boundary"), an in-launch fan-in 3-13 us ("fanin", "splitk-seam"): the cheap seam between "all partials written" and
// "someone sums them" is the boundary.  So:
//   conv_sk_kernel        grid = tiles x S blocks; block (tile, s) accumulates K-steps [KS s / S, KS (s + 1) / S) of its tile
//                         (the general gather or the pointwise loader: any kernel size, stride, padding, pre-activation) and
//                         writes its accumulators, as they stand in the MFMA layout, to slab (tile, s) -- 16 B per lane,
//                         1 KB per wave instruction; no flag, no fence, no wait;
//   conv_sk_reduce_kernel one WAVE per quarter fragment (8 rows x 32 channels of a tile): S x 1 KB reads issued together, the
//                         S partials summed in the FIXED order s = 0 .. S-1 (deterministic: S is a pure function of the
//                         layer's shape and the batch), then the layer's epilogue straight from the MFMA layout -- a lane
//                         holds ONE channel (its constants in six registers) and four rows, every load / store instruction
//                         is two 128-byte segments; same arithmetic, element by element, as conv_epilogue.
// S is chosen on the host (sk2_plan): about two blocks per CU, at least four K-steps per share.
// Blocks that share a weight slice (same column tile and split, different row tiles) get hardware ids congruent mod 8 --
// one XCD, so the slice crosses the fabric once (speed only; at batch 1 the weights are 10x the activations' bytes) -- in
// equal runs per XCD (sk_item).

// ---- operand loaders of the deep-prefetch mainloop below.  They differ from ConvALoader / ConvPwLoader / RowLoader in two
// ways: a K-step's pre-activation constants and bounds mask travel with ITS registers (several K-steps are in flight at
// once), and a load may be issued "invalid" (inv = all ones instead of 0) -- every offset out of the descriptor's range: the
// instruction is issued (so the wave's count of outstanding loads stays a compile-time number and every wait is a counted
// s_waitcnt vmcnt(N)), but it moves no bytes.  A run-time branch around a load group instead makes the compiler wait
// vmcnt(0) at the join -- and it WILL build that branch out of a `valid ? offset : OOB` select on a block-uniform
// condition (the first form of this file: vmcnt(0) in front of every K-step), hence the mask arithmetic.
template <bool PRE>
struct SkPre {
  f32x4 cs, ct;
  unsigned okmask;
};
template <>
struct SkPre<false> {};

template <int N, int RP, bool PRE>
struct SkALoader {                      // multi-tap gather: any kernel size / stride / padding, Cin % 32 == 0, channel-block-major K
  __amdgpu_buffer_rsrc_t rsrc;
  const float* ps;
  const float* pt;
  int pre_act;
  int32_t base[N];
  int32_t hw0[N];
  int H, W, Cin;
  FastDiv fd_kw, fd_taps;

  __device__ __forceinline__ SkALoader(const ConvArgs& a, int m0) {
    const int tid = threadIdx.x;
    H = a.H;
    W = a.W;
    Cin = a.Cin;
    ps = a.pre_scale;
    pt = a.pre_shift;
    pre_act = a.pre_act;
    const int HoWo = a.Ho * a.Wo;
    fd_kw = a.fd_kw;
    fd_taps = a.fd_taps;
    const int n_first = a.fd_howo.div(m0);
    const int64_t img_elems = (int64_t)a.H * a.W * a.Cin;
    const int64_t imgs_left = a.N - n_first;
    int64_t span = (256 + HoWo - 1) / HoWo + 1;           // images a tile can touch
    if (span > imgs_left) span = imgs_left;
    rsrc = make_rsrc(a.x + n_first * img_elems, (uint32_t)(span * img_elems * 4));
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int m = m0 + (tid >> 3) + RP * i;
      if (m < a.M) {
        int n, r, ho, wo;
        a.fd_howo.divmod(m, n, r);
        a.fd_wo.divmod(r, ho, wo);
        const int hi0 = ho * a.stride - a.pad_t;
        const int wi0 = wo * a.stride - a.pad_l;
        base[i] = (int32_t)((((int64_t)(n - n_first) * a.H + hi0) * a.W + wi0) * a.Cin * 4) + (tid & 7) * 16;
        hw0[i] = ((hi0 + 0x4000) << 16) | (wi0 + 0x4000);
      } else {
        base[i] = 0;
        hw0[i] = 0;   // hi0 = wi0 = -0x4000: never inside the image
      }
    }
  }

  // channel-block-major K only (ConvArgs::k_order == 1, Cin % 32 == 0: every multi-tap layer with whole 32-channel slices;
  // sk2_plan admits nothing else): K-step = (32-channel slice, tap), block-uniform, no run-time layout branch in the loop
  __device__ __forceinline__ void load(int kstep, uint32_t inv, f32x4 (&r)[N], SkPre<PRE>& st) const {
    int cblk, tap, kh, kw;
    fd_taps.divmod(kstep, cblk, tap);
    fd_kw.divmod(tap, kh, kw);
    const int ci0 = cblk * BK;
    const int toff = ((kh * W + kw) * Cin + ci0) * 4;
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int hi = (hw0[i] >> 16) - 0x4000 + kh;
      const int wi = (hw0[i] & 0xffff) - 0x4000 + kw;
      const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      r[i] = buf_load4(rsrc, (ok ? (uint32_t)(base[i] + toff) : OOB) | (inv & OOB));
      mask |= ok ? (1u << i) : 0u;
    }
    if constexpr (PRE) {
      st.okmask = mask;                                      // (an invalid step's rows are never read)
      const int cch = (ci0 + (threadIdx.x & 7) * 4) & (int)~inv;      // an invalid step may lie past the last slice: channel 0
      st.cs = *reinterpret_cast<const f32x4*>(ps + cch);
      st.ct = *reinterpret_cast<const f32x4*>(pt + cch);
    }
  }

  __device__ __forceinline__ void finish(f32x4 (&r)[N], const SkPre<PRE>& st) const {
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const bool ok = (st.okmask >> i) & 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = fmaf(r[i][j], st.cs[j], st.ct[j]);
          if (pre_act == ACT_RELU) v = fmaxf(v, 0.f);
          r[i][j] = ok ? v : 0.f;
        }
      }
    }
  }
};

template <int N, int RP, bool PRE>
struct SkPwLoader {                     // 1x1, no padding, Cin % 32 == 0, any stride: a K-step is the next 128 bytes of every row
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t base[N];
  const float* ps;
  const float* pt;
  int pre_act;
  static constexpr uint32_t INVALID = 0x80000000u;

  __device__ __forceinline__ SkPwLoader(const ConvArgs& a, int m0) {
    const int tid = threadIdx.x;
    ps = a.pre_scale;
    pt = a.pre_shift;
    pre_act = a.pre_act;
    const int HoWo = a.Ho * a.Wo;
    const int n_first = a.fd_howo.div(m0);
    const int64_t img_elems = (int64_t)a.H * a.W * a.Cin;
    const int64_t imgs_left = a.N - n_first;
    int64_t span = (256 + HoWo - 1) / HoWo + 1;
    if (span > imgs_left) span = imgs_left;
    rsrc = make_rsrc(a.x + n_first * img_elems, (uint32_t)(span * img_elems * 4));
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int m = m0 + (tid >> 3) + RP * i;
      if (m < a.M) {
        int n, r, ho, wo;
        a.fd_howo.divmod(m, n, r);
        a.fd_wo.divmod(r, ho, wo);
        base[i] = (uint32_t)((((int64_t)(n - n_first) * a.H + ho * a.stride) * a.W + wo * a.stride) * a.Cin * 4) + (tid & 7) * 16;
      } else {
        base[i] = INVALID;
      }
    }
  }

  __device__ __forceinline__ void load(int kstep, uint32_t inv, f32x4 (&r)[N], SkPre<PRE>& st) const {
    const uint32_t o = (uint32_t)kstep * (BK * 4);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = buf_load4(rsrc, (base[i] + o) | (inv & OOB));
    if constexpr (PRE) {
      const int cch = (kstep * BK + (threadIdx.x & 7) * 4) & (int)~inv;
      st.cs = *reinterpret_cast<const f32x4*>(ps + cch);
      st.ct = *reinterpret_cast<const f32x4*>(pt + cch);
      st.okmask = 0;
    }
  }

  __device__ __forceinline__ void finish(f32x4 (&r)[N], const SkPre<PRE>& st) const {
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const bool ok = base[i] != INVALID;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = fmaf(r[i][j], st.cs[j], st.ct[j]);
          if (pre_act == ACT_RELU) v = fmaxf(v, 0.f);
          r[i][j] = ok ? v : 0.f;
        }
      }
    }
  }
};

template <int N, int RP>
struct SkRowLoader {                    // the packed weights [Cout][Kpad]
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t off0, ldb;
  __device__ __forceinline__ SkRowLoader(const float* tile_base, int64_t rows_left, int ld) {
    const int tid = threadIdx.x;
    const int64_t rows = rows_left < RP * N ? rows_left : RP * N;
    rsrc = make_rsrc(tile_base, (uint32_t)(rows * ld * 4));
    ldb = (uint32_t)ld * 4u;
    off0 = (uint32_t)(tid >> 3) * ldb + (uint32_t)(tid & 7) * 16u;
  }
  __device__ __forceinline__ void load(int kstep, uint32_t inv, f32x4 (&r)[N]) const {
    const uint32_t o = off0 + (uint32_t)kstep * (BK * 4);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = buf_load4(rsrc, (o + (uint32_t)i * RP * ldb) | (inv & OOB));
  }
};

// K-steps [kbeg, kend) with the operands requested D K-steps ahead of their use, through a ring of D register sets.
// gemm_mainloop2 fetches ONE K-step ahead: enough at four blocks per CU on a long K loop, where the other blocks' MFMAs
// cover a load's latency -- but a split-K block is alone on its CU (the grid is ~1 block per CU), its share is 4..30
// K-steps, and its weights come from HBM (every launch streams a layer's weights once: 2.4 MB per 14 x 14 layer, nothing
// stays in L2 for the next forward): with one K-step in flight a share of four K-steps paid four exposed memory latencies
// -- 10 us per launch for 1.7 us of MFMAs (profiles/r05_ks_before_*).  Here the first D K-steps are all requested in the
// prologue (a share of <= D K-steps: ONE exposed latency), and slot j is refilled with K-step k + D right behind the LDS
// write that consumed it.  Every load instruction is issued unconditionally (see the loaders): counted waits throughout.
template <class T, int D, class ALoader, class BLoader, class Pre>
__device__ __forceinline__ void gemm_mainloop_deep(const ALoader& al, const BLoader& bl, int kbeg, int kend, float* lds,
                                                   f32x16 (&acc)[1][1]) {
  constexpr int BM = T::BM, BN = T::BN, NA = T::NA, NB = T::NB, RP = T::RP;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  constexpr int BUF = (BM + BN) * LDS_STRIDE;
  constexpr int OFFB = BM * LDS_STRIDE;
  const int st_off = (tid >> 3) * LDS_STRIDE + (tid & 7) * 4;
  const int fr_off = (lane & 31) * LDS_STRIDE + 8 * (lane >> 5);
  auto mfma_step = [&](int cur) {
    const float* pa = lds + cur * BUF + (wr * 32) * LDS_STRIDE + fr_off;
    const float* pb = lds + cur * BUF + OFFB + (wc * 32) * LDS_STRIDE + fr_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f32x4 fa0 = *reinterpret_cast<const f32x4*>(pa + 16 * s), fa1 = *reinterpret_cast<const f32x4*>(pa + 16 * s + 4);
      const f32x4 fb0 = *reinterpret_cast<const f32x4*>(pb + 16 * s), fb1 = *reinterpret_cast<const f32x4*>(pb + 16 * s + 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc[0][0], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc[0][0], 0, 0, 0);
    }
  };
  f32x4 ra[D][NA], rb[D][NB];
  Pre ps[D];
  auto stage = [&](int buf, f32x4 (&xa)[NA], f32x4 (&xb)[NB], const Pre& p) {
    float* wa = lds + buf * BUF + st_off;
    al.finish(xa, p);
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(wa + i * RP * LDS_STRIDE) = xa[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(wa + OFFB + i * RP * LDS_STRIDE) = xb[i];
  };
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const uint32_t inv = ~(uint32_t)((kbeg + j - kend) >> 31);        // 0 while kbeg + j < kend, all ones past the share
    al.load(kbeg + j, inv, ra[j], ps[j]);
    bl.load(kbeg + j, inv, rb[j]);
  }
  stage(0, ra[0], rb[0], ps[0]);
  lds_barrier();
  if (kend - kbeg <= D) {
    // the whole share is in flight already (every batch-1 share): straight-line code, forward exits only
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (kbeg + j >= kend) break;
      mfma_step(j & 1);
      if (j + 1 < D) stage((j & 1) ^ 1, ra[(j + 1) % D], rb[(j + 1) % D], ps[(j + 1) % D]);
      lds_barrier();
    }
    return;
  }
  // Whole ring revolutions first, the share's last r < D K-steps behind the loop.  (One loop with `if (k >= kend) break;`
  // inside the unrolled revolution looks the same and is not: the compiler routes the break through the loop's latch -- a
  // path back to the header on which the later steps of the revolution issued nothing -- and its wait insertion then prices
  // every ring register at the header as that many loads younger than it is: the `s_waitcnt vmcnt(3)` at the top of every
  // revolution of this loop's first form, the ring drained to one K-step.  Read off conv_tnk_kernel's ISA, r05_ablation item 14.)
  const int kfull = kbeg + (kend - kbeg) / D * D;
  for (int k0 = kbeg; k0 < kfull; k0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int k = k0 + j;
      const int cur = (k - kbeg) & 1;
      // slot j was written to LDS one step ago: refill it with K-step k + D
      const uint32_t inv = ~(uint32_t)((k + D - kend) >> 31);
      al.load(k + D, inv, ra[j], ps[j]);
      bl.load(k + D, inv, rb[j]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(cur);
      stage(cur ^ 1, ra[(j + 1) % D], rb[(j + 1) % D], ps[(j + 1) % D]);   // K-step k + 1 (zeros past the end: never read)
      lds_barrier();
    }
  }
#pragma unroll
  for (int j = 0; j < D - 1; ++j) {                          // the tail: everything it needs is in the ring already
    const int k = kfull + j;
    if (k >= kend) break;                                    // a forward exit
    const int cur = (k - kbeg) & 1;
    mfma_step(cur);
    stage(cur ^ 1, ra[j + 1], rb[j + 1], ps[j + 1]);
    lds_barrier();
  }
}

// Block -> work item (c = column tile x split pair, mt = row tile).  Items are numbered c-major and dealt to the eight XCDs
// (hardware block id mod 8) in EQUAL contiguous runs: the blocks that share a weight slice (same c) still sit on one XCD --
// two where a run ends inside a pair -- and every XCD gets ceil(items / 8) blocks.  (Round 5's first mapping gave whole pairs
// to XCDs, c mod 8: 20 pairs of 25 row tiles were 75 blocks on four XCDs and 50 on the other four -- 2.3 against 1.6 blocks
// per CU, the "uneven deal" of the block traces: lives of 12 .. 25 us in one launch.)
__device__ __forceinline__ bool sk_item(int tiles_m, int pairs, int& c, int& mt) {
  const int W = pairs * tiles_m, per = (W + 7) >> 3;
  const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int item = x * per + i;
  if (i >= per || item >= W) return false;
  c = item / tiles_m;
  mt = item - c * tiles_m;
  return true;
}
__host__ inline unsigned sk_grid(int tiles_m, int pairs) { return (unsigned)(((pairs * tiles_m + 7) >> 3) * 8); }

constexpr int SK_DEPTH = 5;             // K-steps in flight per block: 5 x (8 KB + 8 KB) = 80 VGPRs of operands
constexpr int SK_DEPTH_PRE = 3;         // ... with pre-activation constants riding along (8 more VGPRs per K-step): five spilled 12-20 registers

template <class T, bool PRE, int AM>
__global__ __launch_bounds__(T::NT, 2) void conv_sk_kernel(const ConvArgs a, int S, int tiles_m, int tiles_n) {
  static_assert(T::WM == 1 && T::WN == 1 && T::NT == 256, "split-K path: the 64 x 64 tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  int c, mt;                                                  // (column tile, split) pair and row tile: sk_item
  if (!sk_item(tiles_m, tiles_n * S, c, mt)) return;
  const unsigned long long tr_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;   // development aid (net.hip: option dbg = 256)
  const unsigned long long tr_c0 = a.trace ? __builtin_amdgcn_s_memtime() : 0;
  const int nt = c / S, s = c - nt * S;
  const int KS = a.Kpad / BK;
  const int kb = (int)((int64_t)KS * s / S), ke = (int)((int64_t)KS * (s + 1) / S);
  const int m0 = mt * T::BM, n0 = nt * T::BN;
  f32x16 acc[1][1];
  zero_acc<T>(acc);
  using ALoad = typename std::conditional<AM == 1, SkPwLoader<T::NA, T::RP, PRE>, SkALoader<T::NA, T::RP, PRE>>::type;
  using BLoad = SkRowLoader<T::NB, T::RP>;
  const ALoad al(a, m0);
  const BLoad bl(a.w + (int64_t)n0 * a.Kpad, (int64_t)a.Cout - n0, a.Kpad);
  const unsigned long long tr_t1 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  if (ke > kb) gemm_mainloop_deep<T, (PRE ? SK_DEPTH_PRE : SK_DEPTH), ALoad, BLoad, SkPre<PRE>>(al, bl, kb, ke, smem, acc);
  const unsigned long long tr_t2 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  float* slab = a.sk_slab + ((int64_t)(mt * tiles_n + nt) * S + s) * (T::BM * T::BN);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = {acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
    *reinterpret_cast<f32x4*>(slab + (q * T::NT + tid) * 4) = v;
  }
  if (a.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {                                           // conv_igemm_kernel's record: main / fix (here: the slab store) / epi (set-up)
      unsigned long long* t = a.trace + (size_t)blockIdx.x * 8;
      const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
      t[0] = tr_t2 - tr_t1; t[1] = t3 - tr_t2; t[2] = tr_t1 - tr_t0; t[3] = ke - kb; t[4] = 1; t[5] = tr_t0; t[6] = t3;
      t[7] = 1 | ((__builtin_amdgcn_s_memtime() - tr_c0) << 8);
    }
  }
}

// The same split-K block on the B-direct halo-patch mainloop (gemm_mainloop_patch_bd: the tile's pixels + halo in LDS once
// per 32-channel slice, weights in MFMA-fragment order straight from L2): 3x3 / stride 1 / pad 1 layers whose patch fits
// (AMP 3: 128 entries, 5: 168 entries, 6: the 8x8-tile form).  Same slab, same reduce launch.
template <class T, int AMP>
__global__ __launch_bounds__(T::NT, 2) void conv_skp_kernel(const ConvArgs a, int S, int tiles_m, int tiles_n) {
  static_assert(T::WM == 1 && T::WN == 1 && T::NT == 256, "split-K path: the 64 x 64 tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  int c, mt;
  if (!sk_item(tiles_m, tiles_n * S, c, mt)) return;
  const unsigned long long tr_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  const unsigned long long tr_c0 = a.trace ? __builtin_amdgcn_s_memtime() : 0;
  const int nt = c / S, s = c - nt * S;
  const int KS = a.Kpad / BK;
  const int kb = (int)((int64_t)KS * s / S), ke = (int)((int64_t)KS * (s + 1) / S);
  const int m0 = mt * T::BM, n0 = nt * T::BN;
  f32x16 acc[1][1];
  zero_acc<T>(acc);
  using PA = typename std::conditional<AMP == 6, PatchA2D<T>,
                                       typename std::conditional<AMP == 5, PatchA<T, PATCH_EMAX_L>, PatchA<T, PATCH_EMAX_S>>::type>::type;
  const PA pa(a, m0);
  const unsigned long long tr_t1 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  if (ke > kb) gemm_mainloop_patch_bd<T>(pa, a, n0, kb, ke, smem, acc, [] {}, [] {}, [] {});
  const unsigned long long tr_t2 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  float* slab = a.sk_slab + ((int64_t)(mt * tiles_n + nt) * S + s) * (T::BM * T::BN);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = {acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
    *reinterpret_cast<f32x4*>(slab + (q * T::NT + tid) * 4) = v;
  }
  if (a.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      unsigned long long* t = a.trace + (size_t)blockIdx.x * 8;
      const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
      t[0] = tr_t2 - tr_t1; t[1] = t3 - tr_t2; t[2] = tr_t1 - tr_t0; t[3] = ke - kb; t[4] = 1; t[5] = tr_t0; t[6] = t3;
      t[7] = 1 | ((__builtin_amdgcn_s_memtime() - tr_c0) << 8);
    }
  }
}

// The layer's epilogue for FOUR consecutive rows of ONE output channel (what a lane holds of an MFMA accumulator fragment),
// in two halves: prefetch() requests everything but the sums (constants, shortcut values) -- call it before the sums are
// waited for --, finish() applies  act(v scale + shift) (+ shortcut) -> y,  act2(y scale2 + shift2) -> y2  with conv_epilogue's
// arithmetic, element by element, and stores through the output view (channel slice / interior of a larger map / even
// pixels only).
struct SkEpi {
  float sc, sh, al, sc2, sh2, al2;
  float rres[4];
  int img[4], ho[4], wo[4];
  int row0, c;
  bool col_ok, plain_out;
  __device__ __forceinline__ void prefetch(const ConvArgs& a, int row0_, int c_) {
    row0 = row0_;
    c = c_;
    col_ok = c < a.Cout;
    const int cc = col_ok ? c : 0;
    sc = a.scale ? a.scale[cc] : 1.f;
    sh = a.shift ? a.shift[cc] : 0.f;
    al = a.alpha ? a.alpha[cc] : 0.f;
    sc2 = a.scale2 ? a.scale2[cc] : 1.f;
    sh2 = a.shift2 ? a.shift2[cc] : 0.f;
    al2 = a.alpha2 ? a.alpha2[cc] : 0.f;
    plain_out = (a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0);
    const bool strided_res = (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo);
    const bool need_pix = !plain_out || a.y_sub || (a.res && strided_res);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = row0 + e;
      img[e] = ho[e] = wo[e] = 0;
      rres[e] = 0.f;
      if (row < a.M && col_ok) {
        if (need_pix) {
          int rr;
          a.fd_howo.divmod(row, img[e], rr);
          a.fd_wo.divmod(rr, ho[e], wo[e]);
        }
        if (a.res) {
          int64_t ri = row;
          if (strided_res) ri = ((int64_t)img[e] * a.res_H + (int64_t)ho[e] * a.res_stride) * a.res_W + (int64_t)wo[e] * a.res_stride;
          rres[e] = a.res[ri * a.Cout + c];
        }
      }
    }
  }
  __device__ __forceinline__ void finish(const ConvArgs& a, const f32x4& sum) const {
    if (!col_ok) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = row0 + e;
      if (row >= a.M) continue;
      float t = fmaf(sum[e], sc, sh);
      t = apply_act(t, a.act, al);
      if (a.res) t += rres[e];
      const float t2 = apply_act(fmaf(t, sc2, sh2), a.act2, al2);
      int64_t o;
      if (plain_out) o = (int64_t)row * a.y_ld + a.y_coff + c;
      else o = (((int64_t)img[e] * a.y_H + ho[e] + a.y_oy) * a.y_W + wo[e] + a.y_ox) * a.y_ld + a.y_coff + c;
      int64_t oy = o;
      bool y_on = a.y != nullptr;
      if (a.y_sub) {
        y_on = y_on && !((ho[e] | wo[e]) & 1);
        oy = (((int64_t)img[e] * ((a.Ho + 1) >> 1) + (ho[e] >> 1)) * ((a.Wo + 1) >> 1) + (wo[e] >> 1)) * a.Cout + c;
      }
      if (y_on) a.y[oy] = t;
      if (a.y2) a.y2[o] = t2;
    }
  }
};

// one wave per (tile, wave fragment w, quarter q); 64-thread blocks
__global__ __launch_bounds__(64) void conv_sk_reduce_kernel(const ConvArgs a, int S, int tiles_n) {
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int q = item & 3, w = (item >> 2) & 3, tile = item >> 4;
  const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
  const float* slab = a.sk_slab + (int64_t)tile * S * 4096 + (q * 256 + w * 64 + lane) * 4;
  // the fragment's coordinates: register e of quarter q is row 8 q + 4 (lane >> 5) + e, column lane & 31.
  // Everything the epilogue needs besides the partial sums is requested FIRST: one memory latency for the launch, not three
  SkEpi epi;
  epi.prefetch(a, mt * 64 + (w >> 1) * 32 + q * 8 + (lane >> 5) * 4, nt * 64 + (w & 1) * 32 + (lane & 31));
  f32x4 sum;
  {
    constexpr int CH = 16;                                    // partials in flight per lane
    f32x4 v[CH];
    int s0 = 0;
    bool first = true;
    while (s0 < S) {
      const int n = S - s0 < CH ? S - s0 : CH;
#pragma unroll
      for (int i = 0; i < CH; ++i)
        if (i < n) v[i] = *reinterpret_cast<const f32x4*>(slab + (int64_t)(s0 + i) * 4096);
#pragma unroll
      for (int i = 0; i < CH; ++i)
        if (i < n) {
          if (first) {
            sum = v[i];
            first = false;
          } else {
            sum[0] += v[i][0];
            sum[1] += v[i][1];
            sum[2] += v[i][2];
            sum[3] += v[i][3];
          }
        }
      s0 += n;
    }
  }
  epi.finish(a, sum);
}
