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
// one XCD, so the slice crosses the fabric once (speed only; at batch 1 the weights are 10x the activations' bytes).

template <class T, bool PRE, int AM>
__global__ __launch_bounds__(T::NT, T::MIN_BLOCKS) void conv_sk_kernel(const ConvArgs a, int S, int tiles_m, int tiles_n) {
  static_assert(T::WM == 1 && T::WN == 1 && T::NT == 256, "split-K path: the 64 x 64 tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int cq = j / tiles_m, mt = j - cq * tiles_m;
  const int c = cq * 8 + x;                                   // (column tile, split) pair: all its row tiles on one XCD
  if (c >= tiles_n * S) return;
  const int nt = c / S, s = c - nt * S;
  const int KS = a.Kpad / BK;
  const int kb = (int)((int64_t)KS * s / S), ke = (int)((int64_t)KS * (s + 1) / S);
  const int m0 = mt * T::BM, n0 = nt * T::BN;
  f32x16 acc[1][1];
  zero_acc<T>(acc);
  using ALoad = typename std::conditional<AM == 1, ConvPwLoader<T::NA, T::RP, PRE>, ConvALoader<T::NA, T::RP, PRE, 0>>::type;
  using BLoad = RowLoader<T::NB, T::RP>;
  ALoad al(a, m0);
  BLoad bl(a.w + (int64_t)n0 * a.Kpad, (int64_t)a.Cout - n0, a.Kpad);
  if (ke > kb) gemm_mainloop2<T>(al, bl, kb, ke, smem, acc, [] {});
  float* slab = a.sk_slab + ((int64_t)(mt * tiles_n + nt) * S + s) * (T::BM * T::BN);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = {acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
    *reinterpret_cast<f32x4*>(slab + (q * T::NT + tid) * 4) = v;
  }
}

// one wave per (tile, wave fragment w, quarter q); 64-thread blocks
__global__ __launch_bounds__(64) void conv_sk_reduce_kernel(const ConvArgs a, int S, int tiles_n) {
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int q = item & 3, w = (item >> 2) & 3, tile = item >> 4;
  const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
  const float* slab = a.sk_slab + (int64_t)tile * S * 4096 + (q * 256 + w * 64 + lane) * 4;
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
  // the fragment's coordinates: register e of quarter q is row 8 q + 4 (lane >> 5) + e, column lane & 31
  const int c = nt * 64 + (w & 1) * 32 + (lane & 31);
  const int row0 = mt * 64 + (w >> 1) * 32 + q * 8 + (lane >> 5) * 4;
  if (c >= a.Cout) return;
  const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f, al = a.alpha ? a.alpha[c] : 0.f;
  const float sc2 = a.scale2 ? a.scale2[c] : 1.f, sh2 = a.shift2 ? a.shift2[c] : 0.f, al2 = a.alpha2 ? a.alpha2[c] : 0.f;
  const bool plain_out = (a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0);
  const bool strided_res = (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo);
  const bool need_pix = !plain_out || a.y_sub || (a.res && strided_res);
  float rres[4] = {0.f, 0.f, 0.f, 0.f};
  int img[4], ho[4], wo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = row0 + e;
    img[e] = ho[e] = wo[e] = 0;
    if (row < a.M) {
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
