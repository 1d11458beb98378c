// Microbenchmark 3: register-staged mainloop (product) vs LDS-DMA mainloop
// (buffer_load ... lds, unpadded 128-B rows, XOR chunk swizzle on the source side).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm_core.hpp"
using namespace dif;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N, int RP>
struct DmaRowLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t off0, ldb;
  __device__ __forceinline__ DmaRowLoader(const float* tile_base, int64_t rows_left, int ld) {
    const int tid = threadIdx.x;
    const int64_t rows = rows_left < RP * N ? rows_left : RP * N;
    rsrc = make_rsrc(tile_base, (uint32_t)(rows * ld * 4));
    ldb = (uint32_t)ld * 4u;
    const int r = tid >> 3;
    off0 = (uint32_t)r * ldb + (uint32_t)(((tid & 7) ^ ((r >> 1) & 7)) * 16);
  }
  // LDS destination: rows (wave*8 + RP*i) .. +7 of the tile image, 128 B per row, lane-linear
  __device__ __forceinline__ void issue(int kstep, float* lds_tile) const {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float* dst = lds_tile + (wave * 8 + RP * i) * BK;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, off0 + (uint32_t)kstep * (BK * 4) + (uint32_t)i * RP * ldb, 0, 0, 0);
    }
  }
};

template <class T, class AL, class BL>
__device__ __forceinline__ void mainloop_dma(AL& al, BL& bl, int kbeg, int kend, float* lds, f32x16 (&acc)[T::WM][T::WN]) {
  constexpr int WM = T::WM, WN = T::WN, BM = T::BM, BN = T::BN;
  constexpr int BUF = (BM + BN) * BK, OFFB = BM * BK;
  const int lane = threadIdx.x & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int r = lane & 31, h = lane >> 5;
  const int f = (r >> 1) & 7;
  al.issue(kbeg, lds);
  bl.issue(kbeg, lds + OFFB);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ks = kbeg; ks < kend; ++ks) {
    const int cur = (ks - kbeg) & 1;
    if (ks + 1 < kend) {
      al.issue(ks + 1, lds + (cur ^ 1) * BUF);
      bl.issue(ks + 1, lds + (cur ^ 1) * BUF + OFFB);
    }
    const float* pa = lds + cur * BUF + (wr * WM * 32 + r) * BK;
    const float* pb = lds + cur * BUF + OFFB + (wc * WN * 32 + r) * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c0 = ((2 * h + 4 * s) ^ f) * 4, c1 = ((2 * h + 4 * s + 1) ^ f) * 4;
      f32x4 fa[WM][2], fb[WN][2];
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        fa[m][0] = *reinterpret_cast<const f32x4*>(pa + m * 32 * BK + c0);
        fa[m][1] = *reinterpret_cast<const f32x4*>(pa + m * 32 * BK + c1);
      }
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        fb[n][0] = *reinterpret_cast<const f32x4*>(pb + n * 32 * BK + c0);
        fb[n][1] = *reinterpret_cast<const f32x4*>(pb + n * 32 * BK + c1);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int n = 0; n < WN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][t >> 2][t & 3], fb[n][t >> 2][t & 3], acc[m][n], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

template <class T, int DMA>
__global__ __launch_bounds__(T::NT, 2) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WM = T::WM, WN = T::WN;
  const int tiles_n = (N + T::BN - 1) / T::BN;
  const int t = blockIdx.x, m0 = (t / tiles_n) * T::BM, n0 = (t % tiles_n) * T::BN;
  const int tid = threadIdx.x, lane = tid & 63, wr = T::wave_row(), wc = T::wave_col();
  f32x16 acc[WM][WN];
  zero_acc<T>(acc);
  if (DMA) {
    DmaRowLoader<T::NA, T::RP> al(A + (int64_t)m0 * K, (int64_t)M - m0, K);
    DmaRowLoader<T::NB, T::RP> bl(B + (int64_t)n0 * K, (int64_t)N - n0, K);
    mainloop_dma<T>(al, bl, 0, K / BK, smem, acc);
  } else {
    RowLoader<T::NA, T::RP> al(A + (int64_t)m0 * K, (int64_t)M - m0, K);
    RowLoader<T::NB, T::RP> bl(B + (int64_t)n0 * K, (int64_t)N - n0, K);
    gemm_mainloop<T>(al, bl, 0, K / BK, smem, acc);
  }
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
}

template <class T, int DMA>
double run(const float* A, const float* B, float* C, int M, int N, int K, int iters) {
  auto kern = k<T, DMA>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
  const int tiles = ((M + T::BM - 1) / T::BM) * ((N + T::BN - 1) / T::BN);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(T::NT), T::LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(T::NT), T::LDS_BYTES, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

static double checksum(const float* C, size_t n) {
  std::vector<float> h(n); CK(hipMemcpy(h.data(), C, n * 4, hipMemcpyDeviceToHost));
  double s = 0; for (size_t i = 0; i < n; i += 97) s += h[i]; return s;
}

int main() {
  struct Shape { int M, N, K; const char* what; };
  Shape shapes[] = {{16384, 4096, 2304, "large"}, {50176, 256, 2304, "3x3 stage3"}, {802816, 64, 576, "3x3 stage1"},
                    {200704, 256, 64, "1x1 64->256"}, {50176, 512, 128, "1x1 128->512"}};
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
#define RUN(TT, DMA) if (s.N >= TT::BN) { double ms = run<TT, DMA>(A, B, C, s.M, s.N, s.K, 10); \
    printf("  tile %3dx%3d (%d thr) %s : %7.3f ms %6.1f TFLOP/s chk %.4f\n", TT::BM, TT::BN, TT::NT, DMA ? "lds-dma " : "reg-stage", ms, fl / ms / 1e9, checksum(C, (size_t)s.M * s.N)); }
    using T11 = Tile<1, 1>; using T22 = Tile<2, 2>; using T21w = Tile<2, 1, 2, 4>; using T21 = Tile<2,1>;
    RUN(T11, 0) RUN(T11, 1) RUN(T21, 0) RUN(T21, 1) RUN(T22, 0) RUN(T22, 1) RUN(T21w, 0) RUN(T21w, 1)
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  }
  return 0;
}
