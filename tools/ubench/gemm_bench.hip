// Microbenchmark of the shared MFMA mainloop: C[M][N] = A[M][K] * B[N][K]^T, f32.
// Development aid (not part of libdif.so): hipcc --offload-arch=gfx950 -O3 -std=c++17 \
//   -I deep-insight-face_amd/csrc tools/ubench/gemm_bench.hip -o tools/ubench/gemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm_core.hpp"

using namespace dif;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class T, int MINW>
__global__ __launch_bounds__(T::NT, MINW) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        float* __restrict__ C, int M, int N, int K) {
  constexpr int WM = T::WM, WN = T::WN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int64_t m0 = (int64_t)blockIdx.x * T::BM;
  const int n0 = blockIdx.y * T::BN;
  f32x16 acc[WM][WN];
  zero_acc<T>(acc);
  RowLoader<T::NA, T::RP> al(A + m0 * K, (int64_t)M - m0, K);
  RowLoader<T::NB, T::RP> bl(B + (int64_t)n0 * K, (int64_t)N - n0, K);
  gemm_mainloop<T>(al, bl, 0, K / BK, smem, acc);
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int c = n0 + (wc * WN + n) * 32 + (lane & 31);
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wr * WM + m) * 32 + frag_row(lane, r);
        if (row < M && c < N) C[row * N + c] = acc[m][n][r];
      }
  }
}

template <class T, int MINW>
double run(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
  auto kern = gemm_kernel<T, MINW>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
  dim3 grid((M + T::BM - 1) / T::BM, (N + T::BN - 1) / T::BN);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

static double checksum(const float* C, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), C, n * 4, hipMemcpyDeviceToHost));
  double s = 0;
  for (size_t i = 0; i < n; i += 97) s += h[i];
  return s;
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; const char* what; };
  Shape shapes[] = {{16384, 4096, 2304, "large (no quantisation)"},
                    {50176, 256, 2304, "iresnet stage3 3x3 @B=256"},
                    {200704, 128, 1152, "iresnet stage2 3x3 @B=256"},
                    {802816, 64, 576, "iresnet stage1 3x3 @B=256"},
                    {12544, 512, 4608, "iresnet stage4 3x3 @B=256"},
                    {200704, 256, 64, "resnet50v2 stage2 _3_conv 1x1 64->256"},
                    {200704, 64, 256, "resnet50v2 stage2 _1_conv 1x1 256->64"},
                    {50176, 512, 128, "resnet50v2 stage3 _3_conv 1x1 128->512"}};
  const int only_shape = argc > 1 ? atoi(argv[1]) : -1;
  const int only_cfg = argc > 2 ? atoi(argv[2]) : -1;
  int shape_idx = -1;
  for (auto& s : shapes) {
    ++shape_idx;
    if (only_shape >= 0 && shape_idx != only_shape) continue;
    int cfg_idx = -1;
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 4));
    CK(hipMalloc(&B, (size_t)s.N * s.K * 4));
    CK(hipMalloc(&C, (size_t)s.M * s.N * 4));
    std::vector<float> h((size_t)s.M * s.K);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)s.N * s.K);
    for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
    CK(hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double fl = 2.0 * s.M * s.N * s.K;
    printf("== %s  M=%d N=%d K=%d  (%.1f GFLOP)\n", s.what, s.M, s.N, s.K, fl / 1e9);
#define RUN(WM, WN, WGM, WGN, MINW)                                                            \
    {                                                                                          \
      using TT = Tile<WM, WN, WGM, WGN>;                                                       \
      ++cfg_idx;                                                                               \
      if (s.N >= TT::BN && (only_cfg < 0 || only_cfg == cfg_idx)) {                            \
        double ms = run<TT, MINW>(A, B, C, s.M, s.N, s.K, 5);                                  \
        printf("  [%2d] tile %3dx%3d waves %dx%d (wave tile %dx%d) minw %d : %8.3f ms  %6.1f TFLOP/s  (chk %.4f)\n", cfg_idx, \
               TT::BM, TT::BN, WGM, WGN, 32 * WM, 32 * WN, MINW, ms, fl / ms / 1e9, checksum(C, (size_t)s.M * s.N)); \
      }                                                                                        \
    }
    RUN(2, 2, 2, 2, 2)   // 128x128, 4 waves
    RUN(2, 1, 2, 2, 2)   // 128x64
    RUN(1, 2, 2, 2, 2)   // 64x128
    RUN(1, 1, 2, 2, 2)   // 64x64
    RUN(2, 1, 2, 4, 2)   // 128x128, 8 waves (wave 64x32)
    RUN(1, 2, 4, 2, 2)   // 128x128, 8 waves (wave 32x64)
    RUN(2, 2, 2, 4, 1)   // 128x256, 8 waves (wave 64x64)
    RUN(2, 2, 4, 2, 1)   // 256x128, 8 waves (wave 64x64)
    RUN(2, 1, 4, 2, 2)   // 256x64, 8 waves
    RUN(1, 1, 4, 2, 2)   // 128x64, 8 waves (wave 32x32)
    RUN(1, 1, 2, 4, 2)   // 64x128, 8 waves (wave 32x32)
    RUN(4, 2, 2, 2, 1)   // 256x128, 4 waves (wave 128x64)
    RUN(2, 4, 2, 2, 1)   // 128x256, 4 waves (wave 64x128)
    CK(hipFree(A));
    CK(hipFree(B));
    CK(hipFree(C));
  }
  return 0;
}
