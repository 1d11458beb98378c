// Microbenchmark 2: where does the per-tile overhead of the conv kernel come from?
// C[M][N] = A[M][K] * B[N][K]^T with (EPI) direct dword stores or LDS-staged float4 stores,
// (PERSIST) one block per tile or a persistent grid looping over tiles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm_core.hpp"
using namespace dif;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <class T, int EPI>
__device__ __forceinline__ void tile(const float* A, const float* B, float* C, int M, int N, int K, int m0, int n0, float* smem) {
  constexpr int WM = T::WM, WN = T::WN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  f32x16 acc[WM][WN];
  zero_acc<T>(acc);
  RowLoader<T::NA, T::RP> al(A + (int64_t)m0 * K, (int64_t)M - m0, K);
  RowLoader<T::NB, T::RP> bl(B + (int64_t)n0 * K, (int64_t)N - n0, K);
  gemm_mainloop<T>(al, bl, 0, K / BK, smem, acc);
  if (EPI == 0) {
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
  } else {
    constexpr int CS = T::BN + 4, CPR = T::BN / 4, RPP = T::NT / CPR, ITER = T::BM / RPP;
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int n = 0; n < WN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          smem[((wr * WM + m) * 32 + frag_row(lane, r)) * CS + (wc * WN + n) * 32 + (lane & 31)] = acc[m][n][r];
    __syncthreads();
    const int c4 = tid % CPR, r0 = tid / CPR, c = n0 + c4 * 4;
    if (c < N) {
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
        const int rl = r0 + i * RPP, row = m0 + rl;
        if (row < M) *reinterpret_cast<f32x4*>(C + (int64_t)row * N + c) = *reinterpret_cast<const f32x4*>(smem + rl * CS + c4 * 4);
      }
    }
    __syncthreads();
  }
}

template <class T, int EPI, int PERSIST>
__global__ __launch_bounds__(T::NT, 2) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tiles_n = (N + T::BN - 1) / T::BN, tiles_m = (M + T::BM - 1) / T::BM;
  if (PERSIST == 0) {
    const int t = blockIdx.x;
    tile<T, EPI>(A, B, C, M, N, K, (t / tiles_n) * T::BM, (t % tiles_n) * T::BN, smem);
  } else {
    const int total = tiles_m * tiles_n;
    const int P = gridDim.x, p = blockIdx.x;
    const int beg = (int)((int64_t)total * p / P), end = (int)((int64_t)total * (p + 1) / P);
    for (int t = beg; t < end; ++t) tile<T, EPI>(A, B, C, M, N, K, (t / tiles_n) * T::BM, (t % tiles_n) * T::BN, smem);
  }
}

template <class T, int EPI, int PERSIST>
double run(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
  auto kern = k<T, EPI, PERSIST>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
  const int tiles = ((M + T::BM - 1) / T::BM) * ((N + T::BN - 1) / T::BN);
  dim3 grid(PERSIST ? (tiles < 512 ? tiles : 512) : tiles);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main() {
  struct Shape { int M, N, K; const char* what; };
  Shape shapes[] = {{200704, 256, 64, "1x1 64->256 @28x28"}, {200704, 64, 256, "1x1 256->64 @28x28"},
                    {50176, 512, 128, "1x1 128->512 @14x14"}, {50176, 128, 512, "1x1 512->128 @14x14"},
                    {12544, 1024, 256, "1x1 256->1024 @7x7"}, {50176, 256, 2304, "3x3-like K=2304"}};
  for (auto& s : shapes) {
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 4)); CK(hipMalloc(&B, (size_t)s.N * s.K * 4)); CK(hipMalloc(&C, (size_t)s.M * s.N * 4));
    std::vector<float> h((size_t)s.M * s.K); srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)s.N * s.K);
    for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
    CK(hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double fl = 2.0 * s.M * s.N * s.K;
    printf("== %s  M=%d N=%d K=%d\n", s.what, s.M, s.N, s.K);
#define RUN(TT, EPI, PERSIST) if (s.N >= TT::BN) { double ms = run<TT, EPI, PERSIST>(A, B, C, s.M, s.N, s.K, 10); \
    printf("  tile %3dx%3d epi=%s %s : %7.3f ms %6.1f TFLOP/s\n", TT::BM, TT::BN, EPI ? "lds " : "dword", PERSIST ? "persistent" : "one-tile  ", ms, fl / ms / 1e9); }
    using T22 = Tile<2, 2>; using T21 = Tile<2, 1>; using T11 = Tile<1, 1>;
    RUN(T22, 0, 0) RUN(T22, 1, 0) RUN(T22, 0, 1) RUN(T22, 1, 1)
    RUN(T21, 0, 0) RUN(T21, 1, 0) RUN(T21, 1, 1)
    RUN(T11, 0, 0) RUN(T11, 1, 0) RUN(T11, 1, 1)
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  }
  return 0;
}
