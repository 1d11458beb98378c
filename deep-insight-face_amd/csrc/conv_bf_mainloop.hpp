// Body of the split-bf16 B-direct patch mainloop (conv.hip), included once per number of bf16 terms:
//   BF_NPL_VALUE 3 -> gemm_mainloop_patch_bf3  (hi + mid + lo, six products: "bf16x3")
//   BF_NPL_VALUE 2 -> gemm_mainloop_patch_bf2t (hi + mid, three products: "bf16x2"; the lo plane is neither written nor read)
// Textual inclusion instead of a template parameter: see gemm_mainloop_patch_bf in conv.hip.
template <class T, class PA>
__device__ __forceinline__ void BF_MAINLOOP_NAME(const PA& pa, const ConvArgs& a, int n0, int cbeg, int cend,
                                                        char* lds, f32x16 (&acc)[T::WM][T::WN]) {
  constexpr int NPL = BF_NPL_VALUE;                        // bf16 terms per operand: a LITERAL per inclusion (see conv.hip)
  constexpr int WM = T::WM, WN = T::WN;
  static_assert((WM == 2 && (WN == 1 || WN == 2)) || (WM == 4 && WN == 1), "split-bf16 patch path: wave tiles 64 x 64, 64 x 32, 128 x 32");
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int KS = a.Kpad / BK;
  char* staging = lds + bf3p_planes_b(T::BM);
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w3f, a.w3f_bytes);
  uint32_t boff[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n)
    boff[n] = (uint32_t)((n0 >> 5) + T::wave_col() * WN + n) * (uint32_t)KS * (uint32_t)BF3P_KSTEP_B + (uint32_t)lane * 16u;
  const int klast = 9 * cend - 1;
  // sub-step index u = 2 * K-step + half; past the range the last K-step is fetched again (branch-free, never used)
  auto bload = [&](int u, u32x4 (&b)[WN][3]) {
    // (no run-time condition around these loads: at a join behind a skipped load group the compiler must assume the
    // FEWEST younger loads, i.e. wait for vmcnt(0), and the ring's lookahead is gone)
    int ks = u >> 1;
    ks = ks < klast ? ks : klast;
    const uint32_t so = (uint32_t)ks * (uint32_t)BF3P_KSTEP_B + (uint32_t)(u & 1) * 3072u;
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int p = 0; p < NPL; ++p) b[n][p] = __builtin_amdgcn_raw_buffer_load_b128(wrs, boff[n], so + (uint32_t)p * 1024u, 0);
  };
  uint32_t arow[WM];                                        // byte address of (this lane's pixel, tap row kh, kw = 0), per m
  auto aread = [&](int kw, int s, u32x4 (&f)[WM][3]) {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        f[m][p] = *reinterpret_cast<const u32x4*>(lds + arow[m] + kw * BF3P_EB + p * 64 + s * 32);
  };
  // lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi (small terms first); the four accumulators take turns
  auto mfma24 = [&](const u32x4 (&fa)[WM][3], const u32x4 (&fb)[WN][3]) {
    constexpr int NQ = NPL == 3 ? 6 : 3;
    constexpr int QA[6] = {NPL == 3 ? 2 : 1, 0, NPL == 3 ? 1 : 0, 1, 0, 0}, QB[6] = {0, NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[m][QA[q]]),
                                                              __builtin_bit_cast(bf16x8, fb[n][QB[q]]), acc[m][n], 0, 0, 0);
  };
  auto set_row = [&](int kh) {
#pragma unroll
    for (int m = 0; m < WM; ++m) arow[m] = (uint32_t)(pa.base[m] + kh * pa.WP) * (uint32_t)BF3P_EB + (uint32_t)h * 16u;
  };

  // B fragments: a ring of RING sub-step sets requested RING - 1 sub-steps ahead (18 % RING == 0 keeps the ring's phase fixed
  // per slice); A fragments: FA_SETS sets (2 = read one sub-step ahead, 1 = right before their MFMAs).
  // What the loop is bound by (profiles/r03_ablation.txt): with real operands a K-step takes 2.0-2.1 us against 1.34 us of
  // matrix-pipe time, and NOTHING about the operand streams moves it -- half the B bytes per MFMA (256-row blocks), half the B
  // loads per wave (128 x 32 wave tiles), a ring of six sets (2.5 K-steps ahead), accumulators in AccVGPRs.  With all-zero
  // WEIGHTS -- same instructions, same loads, same bytes -- it runs 20 % faster and the clock held inside the kernels goes from
  // 2.06 to 2.38 GHz: six bf16 MFMAs per product on random data are limited by board power, not by this loop's structure.
  constexpr int RING = WN == 1 ? 6 : 3, LOOK = RING - 1;
  constexpr int FA_SETS = WM == 4 ? 1 : 2;
  static_assert(18 % RING == 0, "ring phase");
  u32x4 bq[RING][WN][3];                                   // (third plane unused, and never allocated, with NPL = 2)
  u32x4 fa[FA_SETS][WM][3];
  pa.issue(cbeg, staging);
#pragma unroll
  for (int i = 0; i < LOOK; ++i) bload(18 * cbeg + i, bq[i]);
  // the patch has landed (the LOOK * 3 * WN B loads behind it may be in flight)
  wait_vmcnt<LOOK * NPL * WN>();
  pa.template convert<NPL>(staging, lds);
  lds_barrier();
  for (int cb = cbeg; cb < cend; ++cb) {
    const int u0 = 18 * cb;
    set_row(0);
    if constexpr (FA_SETS == 2) aread(0, 0, fa[0]);
    // sub-step I of the slice (compile time): tap I / 2, half I % 2
    auto substep = [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int N = FA_SETS == 2 ? I + 1 : I, ntap = N / 2;   // the sub-step whose A fragments are read now
      if constexpr (N < 18) {
        if constexpr (N % 6 == 0 && N > 0) set_row(ntap / 3);     // next tap row
        aread(ntap % 3, N & 1, fa[N % FA_SETS]);
      }
      bload(u0 + I + LOOK, bq[(I + LOOK) % RING]);
      if constexpr (I == 2 * PATCH_PF_TAP) {
        if (cb + 1 < cend) pa.issue(cb + 1, staging);      // lands while the remaining taps run
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma24(fa[I % FA_SETS], bq[I % RING]);
      __builtin_amdgcn_sched_barrier(0);
    };
    substep(std::integral_constant<int, 0>());
    substep(std::integral_constant<int, 1>());
    substep(std::integral_constant<int, 2>());
    substep(std::integral_constant<int, 3>());
    substep(std::integral_constant<int, 4>());
    substep(std::integral_constant<int, 5>());
    substep(std::integral_constant<int, 6>());
    substep(std::integral_constant<int, 7>());
    substep(std::integral_constant<int, 8>());
    substep(std::integral_constant<int, 9>());
    substep(std::integral_constant<int, 10>());
    substep(std::integral_constant<int, 11>());
    substep(std::integral_constant<int, 12>());
    substep(std::integral_constant<int, 13>());
    substep(std::integral_constant<int, 14>());
    substep(std::integral_constant<int, 15>());
    substep(std::integral_constant<int, 16>());
    substep(std::integral_constant<int, 17>());
    if (cb + 1 < cend) {
      // this thread's share of the next patch is in LDS (everything but the ring's youngest loads has landed)
      wait_vmcnt<LOOK * NPL * WN>();
      lds_barrier();                                       // every wave has read its last fragment of the old patch
      pa.template convert<NPL>(staging, lds);
      lds_barrier();
    }
  }
  lds_barrier();
}

