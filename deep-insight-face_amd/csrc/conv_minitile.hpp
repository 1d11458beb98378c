// One-image convolution in ONE launch: 16 x 16 output tiles on v_mfma_f32_16x16x4_f32, operands straight from L2 into the
// MFMA operand registers, the K loop split over the waves of the block (round 5).  Included by conv.hip.
//
// Where it runs: the reference's own call shape -- ONE image per call (predictions.py:152-156).  A 14 x 14 x 256 layer is
// then a 196 x 256 x 2304 GEMM: 1.5 us of matrix work for the whole chip, against a fixed ~4.5 us per kernel launch on this
// board (the duration rocprofv3 reports for l2norm_kernel on 512 floats).  The split-K pair of conv_splitk.hpp pays that twice
// per layer (+ the slab round trip): 15 us per layer (profiles/r05_ablation.txt item 1).  One launch per layer needs every CU
// busy WITHOUT a cross-block reduction, i.e. ~200+ independent tiles per layer: 16 x 16 tiles (13 x 16 = 208 for that
// layer; 64 x 64 tiles would be 16).  Such a tile has no operand reuse to speak of (4 flop per byte), so nothing is staged:
//   * a wave owns a run of the tile's K in chunks of 16 k; per chunk a lane loads ONE float4 of A (pixel l & 15, four
//     channels at 4 (l >> 4)) and ONE of B (filter l & 15, the same four k; from a copy of the weights laid out for exactly
//     this read, net.hip: w_f16 -- 1 KB contiguous per wave instruction) -- straight into what four MFMAs consume
//     (A[i][k = l >> 4], B[k = l >> 4][j]: element t of both float4s is the t-th MFMA's operand; the k permutation is the
//     same on both sides).  No LDS, no barrier in the loop; the zero halo and every tail come from out-of-range offsets;
//   * all of a wave's chunks (up to MT_R) are requested before the first MFMA: one exposed memory latency per round;
//   * NW waves split the K run (in-block split-K); their accumulators meet in LDS (NW KB), wave 0 adds them in wave order
//     (deterministic) and applies the layer's epilogue from the MFMA layout (SkEpi: a lane = one channel, four rows);
//   * blocks that share a column tile (the same 16 filters = the same weight bytes) get hardware ids congruent mod 8: one
//     XCD fetches those weights from HBM once (speed only).
// L2 -> CU traffic is tiles x 32 rows x K x 4 bytes (61 MB for the 14 x 14 layer): fine for one image, not for eight --
// mt_plan (conv.hip) admits a layer by that figure; larger batches take the split-K pair or the large-batch kernels.
constexpr int MT_R = 18;                // chunks in flight per wave (8 VGPRs each; 16 with pre-activation constants)
// ... per variant: 16 waves per block = 4 per SIMD = 128 VGPRs (half the chunks), pre-activation constants ride along (half again)
constexpr int mt_round(int nw, bool pre) { return MT_R / ((nw == 16 ? 2 : 1) * (pre ? 2 : 1)); }

template <int NW, bool PRE, bool PW>
__global__ __launch_bounds__(NW * 64, 1) void conv_mt_kernel(const ConvArgs a, int tiles_m, int tiles_n) {
  __shared__ f32x4 red[NW][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int nt, mt;                                                 // column tile (its row tiles on one XCD, equal runs per XCD: sk_item)
  if (!sk_item(tiles_m, tiles_n, nt, mt)) return;
  const int m0 = mt * 16, n0 = nt * 16;
  const int i = lane & 15, kq = lane >> 4;
  // epilogue operands first (wave 0 only uses them): their latency hides under the K loop
  SkEpi epi;
  if (w == 0) epi.prefetch(a, m0 + 4 * kq, n0 + i);
  // A: this lane's output pixel
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, (uint32_t)((int64_t)a.N * a.H * a.W * a.Cin * 4));
  int32_t abase = 0;
  int hi0 = -0x4000, wi0 = -0x4000;                           // never inside the image
  const int m = m0 + i;
  if (m < a.M) {
    int n, r, ho, wo;
    a.fd_howo.divmod(m, n, r);
    a.fd_wo.divmod(r, ho, wo);
    hi0 = ho * a.stride - a.pad_t;
    wi0 = wo * a.stride - a.pad_l;
    abase = (int32_t)((((int64_t)n * a.H + hi0) * a.W + wi0) * a.Cin * 4) + kq * 16;
  }
  // B: the weights in this kernel's fragment order (ConvArgs::w_f16: [column tile][chunk][lane][4], zero filled past Cout):
  // a wave instruction reads ONE contiguous KB -- eight whole lines
  const int nch = a.Kpad / 16;                                // chunks of the tile's K
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.w_f16, a.w_f16_bytes);
  const uint32_t bbase = (uint32_t)nt * (uint32_t)nch * 1024u + (uint32_t)lane * 16u;

  const int c_beg = (int)((int64_t)nch * w / NW), c_end = (int)((int64_t)nch * (w + 1) / NW);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  constexpr int R = mt_round(NW, PRE);
  for (int c0 = c_beg; c0 < c_end; c0 += R) {
    f32x4 av[R], bv[R], cs[PRE ? R : 1], ct[PRE ? R : 1];
    unsigned okm = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int c = c0 + r;
      const uint32_t inv = ~(uint32_t)((c - c_end) >> 31);    // 0 while c < c_end, all ones past the wave's run
      uint32_t aoff;
      int ch;                                                // first channel of this lane's float4
      bool ok = m < a.M;
      if constexpr (PW) {
        aoff = (uint32_t)abase + (uint32_t)c * 64u;
        ch = c * 16 + kq * 4;
      } else {
        int cblk, tap, kh, kw;
        a.fd_taps.divmod(c >> 1, cblk, tap);
        a.fd_kw.divmod(tap, kh, kw);
        const int cc = cblk * 32 + (c & 1) * 16;
        aoff = (uint32_t)(abase + ((kh * a.W + kw) * a.Cin + cc) * 4);
        ch = cc + kq * 4;
        ok = ok && (unsigned)(hi0 + kh) < (unsigned)a.H && (unsigned)(wi0 + kw) < (unsigned)a.W;
      }
      av[r] = buf_load4(xr, (ok ? aoff : OOB) | (inv & OOB));
      bv[r] = buf_load4(wr, (bbase + (uint32_t)c * 1024u) | (inv & OOB));
      if constexpr (PRE) {
        okm |= ok ? (1u << r) : 0u;
        const int chs = ch & (int)~inv;
        cs[r] = *reinterpret_cast<const f32x4*>(a.pre_scale + chs);
        ct[r] = *reinterpret_cast<const f32x4*>(a.pre_shift + chs);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (c0 + r < c_end) {                                  // wave-uniform; no load inside: the counted waits stay exact
        f32x4 v = av[r];
        if constexpr (PRE) {
          const bool ok = (okm >> r) & 1u;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float u = fmaf(v[t], cs[r][t], ct[r][t]);
            if (a.pre_act == ACT_RELU) u = fmaxf(u, 0.f);
            v[t] = ok ? u : 0.f;
          }
        }
        if (r & 1) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t], bv[r][t], acc1, 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t], bv[r][t], acc0, 0, 0, 0);
        }
      }
    }
  }
  f32x4 sum = {acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]};
  red[w][lane] = sum;
  __syncthreads();
  if (w != 0) return;
#pragma unroll
  for (int k = 1; k < NW; ++k) {
    const f32x4 p = red[k][lane];
    sum[0] += p[0];
    sum[1] += p[1];
    sum[2] += p[2];
    sum[3] += p[3];
  }
  epi.finish(a, sum);
}
