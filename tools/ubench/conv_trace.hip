// Development aid: runs ONE convolution layer through the library's own dif::conv_run with the
// in-kernel trace enabled and prints where the blocks spend their time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deep-insight-face_amd/csrc tools/ubench/conv_trace.hip \
//         -L deep-insight-face_amd/lib -ldif -Wl,-rpath,$PWD/deep-insight-face_amd/lib -o /tmp/conv_trace
//   conv_trace N H W Cin Cout K stride pad res pre korder
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include "ops.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  if (argc < 12) { printf("usage: N H W Cin Cout K stride pad res pre korder\n"); return 1; }
  int N = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
  int K = atoi(argv[6]), stride = atoi(argv[7]), pad = atoi(argv[8]), res = atoi(argv[9]), pre = atoi(argv[10]);
  int korder = atoi(argv[11]);
  int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  int Kdim = K * K * Cin, Kpad = (Kdim + 31) / 32 * 32;
  int64_t M = (int64_t)N * Ho * Wo;
  size_t nx = (size_t)N * H * W * Cin, nw = (size_t)Cout * Kpad, ny = (size_t)M * Cout;
  std::vector<float> hx(nx), hw(nw), hv(std::max(Cin, Cout), 1.0f);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
  float *x, *w, *y, *r, *sc, *sh, *slab;
  unsigned* flag;
  unsigned long long* trace;
  const int maxb = dif::conv_max_blocks();
  CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&r, ny * 4));
  CK(hipMalloc(&sc, hv.size() * 4)); CK(hipMalloc(&sh, hv.size() * 4));
  CK(hipMalloc(&slab, (size_t)maxb * dif::conv_slab_floats() * 4)); CK(hipMalloc(&flag, maxb * 4));
  const int64_t max_tiles = std::max<int64_t>(((M + 63) / 64) * ((Cout + 63) / 64), maxb);
  CK(hipMalloc(&trace, (size_t)max_tiles * 64));
  CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
  CK(hipMemset(r, 0, ny * 4)); CK(hipMemset(flag, 0, maxb * 4));
  CK(hipMemcpy(sc, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(sh, 0, hv.size() * 4));
  dif::ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.use_pipe = 1;
  a.x = x; a.w = w; a.y = y; a.scale = sc; a.shift = sh; a.res = res ? r : nullptr;
  if (pre) { a.pre_scale = sc; a.pre_shift = sh; a.pre_act = dif::ACT_RELU; }
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.KH = a.KW = K; a.stride = stride;
  a.pad_t = a.pad_l = pad; a.Kpad = Kpad; a.k_order = korder; a.M = (int)M; a.act = dif::ACT_RELU;
  a.res_H = Ho; a.res_W = Wo; a.res_stride = 1;
  a.y_ld = Cout; a.y_H = Ho; a.y_W = Wo;
  a.sk_slab = slab; a.sk_flag = flag; a.sk_max_blocks = maxb; a.sk_spin_limit = 1 << 20;

  unsigned epoch = 0;
  hipStream_t st = nullptr;
  for (int i = 0; i < 5; ++i) { a.sk_epoch = ++epoch; if (dif::conv_run(a, st)) { printf("conv_run failed\n"); return 1; } }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) { a.sk_epoch = ++epoch; dif::conv_run(a, st); }
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flops = 2.0 * M * Cout * Kdim;
  printf("M=%lld N=%d K=%d (KS=%d)  %.4f ms  %.1f TFLOP/s  ideal@144TF %.1f us\n", (long long)M, Cout, Kdim, Kpad / 32,
         ms, flops / ms / 1e9, flops / 144e12 * 1e6);
  {
    std::vector<float> hy(ny);
    CK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost));
    double cs = 0, ca = 0;
    for (size_t i = 0; i < ny; ++i) { cs += hy[i]; ca += hy[i] * (double)((i % 97) + 1); }
    printf("checksum %.9g %.9g\n", cs, ca);
  }
  // traced run
  CK(hipMemset(trace, 0, (size_t)max_tiles * 64));
  a.trace = trace; a.sk_epoch = ++epoch;
  dif::conv_run(a, st);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> t((size_t)max_tiles * 8);
  CK(hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost));
  double v[5] = {0, 0, 0, 0, 0}, life = 0, cyc = 0;
  unsigned long long tmin = ~0ull, tmax = 0;
  int nb = 0, kind = 0;
  for (int64_t b = 0; b < max_tiles; ++b) if (t[(size_t)b * 8 + 7]) {
    kind = (int)(t[b * 8 + 7] & 255);
    cyc += (double)(t[b * 8 + 7] >> 8);
    for (int i = 0; i < 5; ++i) v[i] += (double)t[b * 8 + i];
    life += (t[b * 8 + 6] - t[b * 8 + 5]) * 0.01;
    tmin = std::min(tmin, t[b * 8 + 5]); tmax = std::max(tmax, t[b * 8 + 6]);
    ++nb;
  }
  if (!nb) { printf("(no trace records)\n"); return 0; }
  {
    std::vector<double> st, en;
    for (int64_t b = 0; b < max_tiles; ++b) if (t[(size_t)b * 8 + 7]) { st.push_back((t[b * 8 + 5] - tmin) * 0.01); en.push_back((t[b * 8 + 6] - tmin) * 0.01); }
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
    printf("block start (us) p50 %.1f p99 %.1f max %.1f | block end p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f\n", st[nb / 2],
           st[nb * 99 / 100], st[nb - 1], en[nb / 100], en[nb / 10], en[nb / 2], en[nb * 9 / 10], en[nb * 99 / 100], en[nb - 1]);
  }
  if (getenv("SHOW_MAP")) {
    double ex[8] = {0}, es[4] = {0}; int nx[8] = {0}, ns[4] = {0};
    for (int b = 0; b < 1024 && b < max_tiles; ++b) if (t[(size_t)b * 8 + 7]) {
      const double e = (t[b * 8 + 6] - tmin) * 0.01;
      ex[b & 7] += e; nx[b & 7]++; es[(b >> 8) & 3] += e; ns[(b >> 8) & 3]++;
    }
    printf("mean block end by XCD (b %% 8):");
    for (int x = 0; x < 8; ++x) printf(" %.1f", ex[x] / (nx[x] ? nx[x] : 1));
    printf(" | by resident slot (b >> 8):");
    for (int x = 0; x < 4; ++x) printf(" %.1f", es[x] / (ns[x] ? ns[x] : 1));
    printf("\n");
    // per CU within XCD 0: blocks b with b%8==0: CU index j%32 where j=b>>3
    double ec[32] = {0}; int nc[32] = {0};
    for (int b = 0; b < 1024 && b < max_tiles; b += 8) if (t[(size_t)b * 8 + 7]) { const int cu = (b >> 3) & 31; ec[cu] += (t[b * 8 + 6] - tmin) * 0.01; nc[cu]++; }
    printf("XCD 0, mean block end by CU (dispatch order):");
    for (int c = 0; c < 32; ++c) printf(" %.0f", ec[c] / (nc[c] ? nc[c] : 1));
    printf("\n");
  }
  if (cyc > 0) printf("shader clock held over the blocks' lives (s_memtime / s_memrealtime): %.3f GHz\n", cyc / (life * 1e3));
  if (kind == 2)
    printf("pipelined kernel: %d blocks, %.2f tiles/block, span %.1f us, mean block life %.1f us; per tile: set-up %.2f us, first loads -> LDS %.2f us, "
           "K-steps %.2f us, hand-over %.2f us\n", nb, v[4] / nb, (tmax - tmin) * 0.01, life / nb, v[0] * 0.01 / v[4], v[1] * 0.01 / v[4],
           v[2] * 0.01 / v[4], v[3] * 0.01 / v[4]);
  else
    printf("plain kernel: %d blocks, %.2f (part-)tiles/block, %.1f K-steps/block, span %.1f us, mean block life %.1f us; per block: "
           "mainloops %.1f us (%.2f us/K-step incl. prologues), publish/collect partials %.1f us, epilogues %.1f us\n", nb, v[4] / nb, v[3] / nb,
           (tmax - tmin) * 0.01, life / nb, v[0] * 0.01 / nb, v[0] * 0.01 / v[3], v[1] * 0.01 / nb, v[2] * 0.01 / nb);
  return 0;
}
