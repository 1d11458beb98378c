// Small-batch 3x3 convolution, wide split-K form (round 5, second half).  Included by conv.hip behind conv_tn_kernel and
// conv_splitk.hpp.
//
// conv_skp_kernel cuts a layer into 64 x 64 tiles x S shares -- about TWO blocks per CU -- and a block's life depends on
// how many neighbours the dispatcher gave its CU (profiles/r05_ablation.txt items 2 / 6: 12 .. 25 us inside one launch of
// the 14 x 14 stage at batch 8, for 12 us of matrix work per SIMD).  This kernel cuts the same layer into conv_tn_kernel's
// 64-pixel x 128-CHANNEL tiles x S shares with tiles x S <= the number of CUs: ONE block per CU, one wave per SIMD, every
// wave 32 pixels x 64 channels (an A fragment read from the LDS patch feeds 32 MFMAs).  With nobody else on the SIMD to
// cover a load's latency the wave covers it itself:
//   * B (weights in MFMA-fragment order, ConvArgs::w_frag) comes from L2 / HBM straight into a ring of TNK_D = 3 K-steps
//     of operand registers (96 VGPRs; a wave alone on its SIMD has 512), each half K-step refilled right behind the
//     MFMAs that consumed it: ~2.5 K-steps = 2 us of lookahead.  Every load instruction is issued unconditionally -- a
//     K-step past the share gets out-of-range offsets and moves no bytes -- so every wait is a counted s_waitcnt
//     vmcnt(N) (conv_splitk.hpp tells why a branch around a load is not an option);
//   * A: the tile's pixels + halo of one 32-channel slice lie in LDS (PatchA), TWO buffers: the next slice's patch is
//     requested at ring phase 0 of the first ring group that starts in the current slice (unconditional loads again,
//     out of range when nothing is needed), written to the other buffer two K-steps later (an LDS-only branch), and the
//     block meets at ONE barrier per nine K-steps when it switches buffers.
// The partial sums leave as four 64 x 64-tile slabs' worth of MFMA-layout fragments per block -- conv_sk_kernel's slab
// format, so conv_sk_reduce_kernel (fixed-order sums + the layer's epilogue) is the second launch unchanged.
// K order, products and the order of the partial sums are a pure function of (shape, batch): deterministic.
constexpr int TNK_D = 3;

template <int EMAX>
__device__ __forceinline__ void tnk_mainloop(const PatchA<TileTN, EMAX>& pa, const ConvArgs& a, int n0, int kb, int ke,
                                             float* lds, f32x16 (&acc)[1][2]) {
  using PA = PatchA<TileTN, EMAX>;
  constexpr int D = TNK_D;
  constexpr uint32_t PB = (uint32_t)EMAX * 128u;             // bytes per patch buffer (a multiple of 128: the swizzle XORs bits 4..6)
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  const int KS = a.Kpad / BK;
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w_frag, a.w_frag_bytes);
  uint32_t lane_off[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
    lane_off[n] = (uint32_t)((n0 >> 5) + TileTN::wave_col() * 2 + n) * (uint32_t)KS * 4096u + (uint32_t)lane * 16u;
  // half s of K-step k: pieces (s, 0), (s, 1) of both column fragments; k >= ke: out of range (the vector offset is what the
  // descriptor checks), no bytes move
  auto bload = [&](int k, int s, f32x4 (&b)[2][2]) {
    const uint32_t inv = (uint32_t)((ke - 1 - k) >> 31) & OOB;
    const uint32_t so = (uint32_t)k * 4096u + (uint32_t)s * 2048u;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      b[n][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_off[n] | inv, so, 0));
      b[n][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (lane_off[n] + 1024u) | inv, so, 0));
    }
  };
  auto patch_load = [&](int cblk, uint32_t inv, f32x4 (&r)[PA::NPC]) {
#pragma unroll
    for (int j = 0; j < PA::NPC; ++j)
      r[j] = buf_load4(pa.rsrc, (pa.goff[j] == OOB ? OOB : pa.goff[j] + (uint32_t)cblk * 128u) | inv);
  };
  char* const lds_b = reinterpret_cast<char*>(lds);
  auto patch_store = [&](uint32_t buf, const f32x4 (&r)[PA::NPC]) {
#pragma unroll
    for (int j = 0; j < PA::NPC; ++j) {
      const int slot = tid + TileTN::NT * j;
      if (slot < EMAX * 8) *reinterpret_cast<f32x4*>(lds_b + buf + slot * 16) = r[j];
    }
  };
  auto mfma16 = [&](const f32x4 (&f)[2], const f32x4 (&b)[2][2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[q][t], b[n][q][t], acc[0][n], 0, 0, 0);
  };

  f32x4 br[D][2][2][2];                                      // [ring slot][half s][column fragment n][piece u]
  f32x4 pr[PA::NPC];
  int cb = kb / 9, tap = kb - cb * 9;
  int kw = tap % 3;
  int eoff = (tap / 3) * pa.WP + kw;
  // ---- prologue: the first slice's patch, then the ring's first D K-steps
  patch_load(cb, 0u, pr);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    bload(kb + j, 0, br[j][0]);
    bload(kb + j, 1, br[j][1]);
  }
  patch_store((uint32_t)(cb & 1) * PB, pr);
  int cb_ld = cb;                                            // newest slice whose patch lies in LDS
  if (tap >= 7 && (cb + 1) * 9 < ke) {                       // the first ring group would straddle the slice boundary: its patch now
    patch_load(cb + 1, 0u, pr);
    patch_store((uint32_t)((cb + 1) & 1) * PB, pr);
    cb_ld = cb + 1;
  }
  lds_barrier();
  const uint32_t l0x = (uint32_t)(2 * h);
  auto frag_addr = [&]() -> uint32_t {
    const uint32_t e = (uint32_t)(pa.base[0] + eoff);
    return (uint32_t)(cb & 1) * PB + ((e << 7) | ((l0x ^ ((e >> 1) & 7u)) << 4));
  };
  auto read_frag = [&](uint32_t a0, int s, f32x4 (&f)[2]) {
    f[0] = *reinterpret_cast<const f32x4*>(lds_b + (a0 ^ (uint32_t)(64 * s)));
    f[1] = *reinterpret_cast<const f32x4*>(lds_b + (a0 ^ (uint32_t)(64 * s + 16)));
  };
  f32x4 fa[2], fn[2];
  uint32_t a0 = frag_addr();
  read_frag(a0, 0, fa);
  int need_cb = -1;                                          // the slice whose patch rides in pr, or -1
  for (int k0 = kb; k0 < ke; k0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int k = k0 + j;
      if (k >= ke) break;                                    // block-uniform EXIT (not a join): the counted waits stay exact
      if (j == 0) {
        // next slice's patch wanted: not in LDS yet, and the share reaches it.  Mask arithmetic, no branch around the loads
        const int off_mask = -(cb_ld - cb) | ~(((cb + 1) * 9 - ke) >> 31);      // 0: wanted, -1: not
        need_cb = (cb + 1) | off_mask;
        patch_load(cb + 1, (uint32_t)off_mask & OOB, pr);
      }
      if (j == D - 1 && need_cb >= 0) {                      // (LDS only inside the branch) two K-steps after the request
        patch_store((uint32_t)(need_cb & 1) * PB, pr);
        cb_ld = need_cb;
      }
      read_frag(a0, 1, fn);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(fa, br[j][0]);
      bload(k + D, 0, br[j][0]);
      if (++kw == 3) {
        kw = 0;
        eoff += pa.WP - 2;
      } else {
        ++eoff;
      }
      const bool more = k + 1 < ke;
      if (++tap == 9) {
        tap = 0;
        eoff = 0;
        ++cb;
        if (more) lds_barrier();                             // every wave holds its last fragments of the old patch (fn) in registers
      }
      if (more) {
        a0 = frag_addr();
        read_frag(a0, 0, fa);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma16(fn, br[j][1]);
      bload(k + D, 1, br[j][1]);
    }
  }
}

template <int EMAX>
__global__ __launch_bounds__(256, 1) void conv_tnk_kernel(const ConvArgs a, int S, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float tnk_smem[];      // two patches of EMAX entries
  const int tid = threadIdx.x;
  int c, mt;
  if (!sk_item(tiles_m, tiles_n * S, c, mt)) return;
  const unsigned long long tr_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  const unsigned long long tr_c0 = a.trace ? __builtin_amdgcn_s_memtime() : 0;
  const int nt = c / S, s = c - nt * S;
  const int KS = a.Kpad / BK;
  const int kb = (int)((int64_t)KS * s / S), ke = (int)((int64_t)KS * (s + 1) / S);
  const int m0 = mt * TileTN::BM, n0 = nt * TileTN::BN;
  f32x16 acc[1][2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][n][r] = 0.f;
  unsigned long long tr_t1 = 0;
  {
    const PatchA<TileTN, EMAX> pa(a, m0);
    tr_t1 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
    if (ke > kb) tnk_mainloop<EMAX>(pa, a, n0, kb, ke, tnk_smem, acc);
  }
  const unsigned long long tr_t2 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  // the wave's two 32 x 32 fragments as fragments (wave row, n) of 64 x 64 tile (mt, 2 nt + wave column): conv_sk_kernel's slabs
  const int lane = tid & 63, wr = TileTN::wave_row(), wc = TileTN::wave_col();
  const int tiles_n64 = tiles_n * 2;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    float* slab = a.sk_slab + ((int64_t)(mt * tiles_n64 + nt * 2 + wc) * S + s) * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = {acc[0][n][4 * q], acc[0][n][4 * q + 1], acc[0][n][4 * q + 2], acc[0][n][4 * q + 3]};
      *reinterpret_cast<f32x4*>(slab + (q * 256 + (wr * 2 + n) * 64 + lane) * 4) = v;
    }
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
