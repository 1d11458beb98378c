// Microbenchmark 4: v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 at the same block tile
// (does the chip hold a different clock on the two shapes? MI355X_MICROARCH.md DVFS item 7).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm_core.hpp"
using namespace dif;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// 64x64 block, 4 waves (2x2), wave tile 32x32 as 2x2 sub-tiles of 16x16
template <class AL, class BL>
__device__ __forceinline__ void mainloop16(AL& al, BL& bl, int ksteps, float* lds, f32x4 (&acc)[2][2]) {
  using T = Tile<1, 1>;
  constexpr int BM = 64, BN = 64, NA = T::NA, NB = T::NB, RP = T::RP;
  constexpr int BUF = (BM + BN) * LDS_STRIDE, OFFB = BM * LDS_STRIDE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int st_off = (tid >> 3) * LDS_STRIDE + (tid & 7) * 4;
  const int fr_off = (lane & 15) * LDS_STRIDE + 4 * (lane >> 4);
  f32x4 ra[NA], rb[NB];
  al.load(0, ra); bl.load(0, rb);
#pragma unroll
  for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(lds + st_off + i * RP * LDS_STRIDE) = ra[i];
#pragma unroll
  for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(lds + OFFB + st_off + i * RP * LDS_STRIDE) = rb[i];
  __syncthreads();
  for (int ks = 0; ks < ksteps; ++ks) {
    const int cur = ks & 1;
    const bool more = ks + 1 < ksteps;
    if (more) { al.load(ks + 1, ra); bl.load(ks + 1, rb); }
    const float* pa = lds + cur * BUF + (wr * 32) * LDS_STRIDE + fr_off;
    const float* pb = lds + cur * BUF + OFFB + (wc * 32) * LDS_STRIDE + fr_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 fa[2], fb[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) fa[m] = *reinterpret_cast<const f32x4*>(pa + m * 16 * LDS_STRIDE + 16 * s);
#pragma unroll
      for (int n = 0; n < 2; ++n) fb[n] = *reinterpret_cast<const f32x4*>(pb + n * 16 * LDS_STRIDE + 16 * s);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][t], fb[n][t], acc[m][n], 0, 0, 0);
    }
    if (more) {
      float* wa = lds + (cur ^ 1) * BUF + st_off;
      float* wb = lds + (cur ^ 1) * BUF + OFFB + st_off;
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(wa + i * RP * LDS_STRIDE) = ra[i];
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(wb + i * RP * LDS_STRIDE) = rb[i];
    }
    __syncthreads();
  }
}

template <int SHAPE16>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  using T = Tile<1, 1>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tiles_n = (N + 63) / 64;
  const int t = blockIdx.x, m0 = (t / tiles_n) * 64, n0 = (t % tiles_n) * 64;
  const int lane = threadIdx.x & 63, wr = T::wave_row(), wc = T::wave_col();
  RowLoader<T::NA, T::RP> al(A + (int64_t)m0 * K, (int64_t)M - m0, K);
  RowLoader<T::NB, T::RP> bl(B + (int64_t)n0 * K, (int64_t)N - n0, K);
  if (SHAPE16) {
    f32x4 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    mainloop16(al, bl, K / BK, smem, acc);
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 4; ++r) {
      const int row = m0 + wr * 32 + m * 16 + (lane >> 4) * 4 + r, col = n0 + wc * 32 + n * 16 + (lane & 15);
      if (row < M && col < N) C[(int64_t)row * N + col] = acc[m][n][r];
    }
  } else {
    f32x16 acc[1][1];
    zero_acc<T>(acc);
    gemm_mainloop<T>(al, bl, 0, K / BK, smem, acc);
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 32 + frag_row(lane, r), col = n0 + wc * 32 + (lane & 31);
      if (row < M && col < N) C[(int64_t)row * N + col] = acc[0][0][r];
    }
  }
}

template <int SHAPE16>
double run(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
  using T = Tile<1, 1>;
  auto kern = k<SHAPE16>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
  const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), T::LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), T::LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}
static double checksum(const float* C, size_t n) {
  std::vector<float> h(n); CK(hipMemcpy(h.data(), C, n * 4, hipMemcpyDeviceToHost));
  double s = 0; for (size_t i = 0; i < n; i += 97) s += h[i]; return s;
}
int main() {
  const int M = 16384, N = 4096, K = 2304;
  float *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  std::vector<float> h((size_t)M * K); srand(1);
  for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
  CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  h.resize((size_t)N * K);
  for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
  CK(hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const double fl = 2.0 * M * N * K;
  for (int rep = 0; rep < 3; ++rep) {
    double a = run<0>(A, B, C, M, N, K, 20); double ca = checksum(C, (size_t)M * N);
    double b = run<1>(A, B, C, M, N, K, 20); double cb = checksum(C, (size_t)M * N);
    printf("32x32x2: %.3f ms %.1f TF (chk %.3f) | 16x16x4: %.3f ms %.1f TF (chk %.3f)\n", a, fl / a / 1e9, ca, b, fl / b / 1e9, cb);
  }
  return 0;
}
