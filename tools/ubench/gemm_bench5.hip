// Microbenchmark 5: where does the K loop of the 64x64 / four-blocks-per-CU mainloop lose its 20 %?
// Same loop as gemm_core.hpp's gemm_mainloop with pieces switched off (results are then wrong on
// purpose; only the time matters):  bit 0 = no barrier, bit 1 = no global loads, bit 2 = no LDS
// fragment reads, bit 3 = no LDS staging writes.   usage: gemm_bench5 [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm_core.hpp"
using namespace dif;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int V>
__global__ __launch_bounds__(256, 4) void k(const float* A, const float* B, float* C, int M, int N, int K) {
  using T = Tile<1, 1>;
  constexpr int NA = T::NA, NB = T::NB, RP = T::RP;
  constexpr int BUF = (64 + 64) * LDS_STRIDE, OFFB = 64 * LDS_STRIDE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int tiles_n = N / 64;
  const int KS = K / 32;
  const int st_off = (tid >> 3) * LDS_STRIDE + (tid & 7) * 4;
  const int fr_off = (lane & 31) * LDS_STRIDE + 8 * (lane >> 5);
  for (int tile = blockIdx.x; tile < (M / 64) * tiles_n; tile += gridDim.x) {
    const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
    RowLoader<NA, RP> al(A + (int64_t)m0 * K, M - m0, K);
    RowLoader<NB, RP> bl(B + (int64_t)n0 * K, N - n0, K);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 ra[NA], rb[NB];
    al.load(0, ra); bl.load(0, rb);
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(lds + st_off + i * RP * LDS_STRIDE) = ra[i];
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(lds + OFFB + st_off + i * RP * LDS_STRIDE) = rb[i];
    __syncthreads();
    f32x4 fa0 = {1.f, 2.f, 3.f, 4.f}, fa1 = fa0, fb0 = fa0, fb1 = fa0;
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1;
      const bool more = ks + 1 < KS;
      if (!(V & 2) && more) { al.load(ks + 1, ra); bl.load(ks + 1, rb); }
      const float* pa = lds + cur * BUF + (wr * 32) * LDS_STRIDE + fr_off;
      const float* pb = lds + cur * BUF + OFFB + (wc * 32) * LDS_STRIDE + fr_off;
      if (V & 16) {
        // all eight fragment reads of the step up front: the second half lands while the first 8 MFMAs run
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa), a1 = *reinterpret_cast<const f32x4*>(pa + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(pb), b1 = *reinterpret_cast<const f32x4*>(pb + 4);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(pa + 16), a3 = *reinterpret_cast<const f32x4*>(pa + 20);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(pb + 16), b3 = *reinterpret_cast<const f32x4*>(pb + 20);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t], b2[t], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[t], b3[t], acc, 0, 0, 0);
      } else
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!(V & 4)) {
          fa0 = *reinterpret_cast<const f32x4*>(pa + 16 * s); fa1 = *reinterpret_cast<const f32x4*>(pa + 16 * s + 4);
          fb0 = *reinterpret_cast<const f32x4*>(pb + 16 * s); fb1 = *reinterpret_cast<const f32x4*>(pb + 16 * s + 4);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc, 0, 0, 0);
      }
      if (!(V & 8) && more) {
        float* wa = lds + (cur ^ 1) * BUF + st_off;
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(wa + i * RP * LDS_STRIDE) = ra[i];
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(wa + OFFB + i * RP * LDS_STRIDE) = rb[i];
      }
      if (!(V & 1)) __syncthreads();
    }
    // minimal epilogue so that nothing is optimised away
    float sacc = 0.f;
    for (int r = 0; r < 16; ++r) sacc += acc[r];
    if (V & 2) sacc += ra[0][0] + rb[0][0];
    C[(int64_t)tile * 256 + tid] = sacc;
    __syncthreads();
  }
}

// LDS-DMA operand staging (no VGPR round trip, no ds_write) with a ring of NST stages of one 32-wide
// K-step each; NST = 2 fits four blocks per CU, NST = 3 (operands fetched two steps ahead) three.
template <int NST, int MINB>
__global__ __launch_bounds__(256, MINB) void kdma(const float* A, const float* B, float* C, int M, int N, int K) {
  using T = Tile<1, 1>;
  constexpr int NA = T::NA, NB = T::NB, RP = T::RP;
  constexpr int BUF = (64 + 64) * BK, OFFB = 64 * BK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int tiles_n = N / 64;
  const int KS = K / 32;
  const int r = lane & 31, h = lane >> 5;
  const int f = dma_swizzle(r);
  for (int tile = blockIdx.x; tile < (M / 64) * tiles_n; tile += gridDim.x) {
    const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
    DmaRowLoader<NA, RP> al(A + (int64_t)m0 * K, M - m0, K);
    DmaRowLoader<NB, RP> bl(B + (int64_t)n0 * K, N - n0, K);
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // prime NST-1 stages
    for (int s = 0; s < NST - 1; ++s)
      if (s < KS) { al.issue(s, lds + s * BUF); bl.issue(s, lds + s * BUF + OFFB); }
    if (NST == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int ks = 0; ks < KS; ++ks) {
      int nxt = cur + (NST - 1); if (nxt >= NST) nxt -= NST;
      if (ks + NST - 1 < KS) { al.issue(ks + NST - 1, lds + nxt * BUF); bl.issue(ks + NST - 1, lds + nxt * BUF + OFFB); }
      const float* pa = lds + cur * BUF + (wr * 32 + r) * BK;
      const float* pb = lds + cur * BUF + OFFB + (wc * 32 + r) * BK;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c0 = ((2 * h + 4 * s) ^ f) * 4, c1 = ((2 * h + 4 * s + 1) ^ f) * 4;
        const f32x4 fa0 = *reinterpret_cast<const f32x4*>(pa + c0), fa1 = *reinterpret_cast<const f32x4*>(pa + c1);
        const f32x4 fb0 = *reinterpret_cast<const f32x4*>(pb + c0), fb1 = *reinterpret_cast<const f32x4*>(pb + c1);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc, 0, 0, 0);
      }
      // the stage the NEXT step reads must have landed: with three stages the youngest four pieces
      // (the stage after it) may still be in flight
      if (NST == 3 && ks + 2 < KS) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur = cur + 1 == NST ? 0 : cur + 1;
    }
    float sacc = 0.f;
    for (int q = 0; q < 16; ++q) sacc += acc[q];
    C[(int64_t)tile * 256 + tid] = sacc;
    __syncthreads();
  }
}

// Hybrid: A (activations: the operand that misses L2) through registers, fetched TWO K-steps ahead with
// two 8-register sets (= the 16 staging registers the plain loop spends on A + B); B (weights,
// L2-resident) through LDS-DMA, one step ahead, no registers.  LDS: A padded 2 x 9 KB, B swizzled 2 x 8 KB.
__global__ __launch_bounds__(256, 4) void khyb(const float* A, const float* B, float* C, int M, int N, int K) {
  using T = Tile<1, 1>;
  constexpr int NA = T::NA, NB = T::NB, RP = T::RP;
  constexpr int ABUF = 64 * LDS_STRIDE, BBUF = 64 * BK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsA = lds;                    // 2 * ABUF floats
  float* ldsB = lds + 2 * ABUF;         // 2 * BBUF floats
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int tiles_n = N / 64;
  const int KS = K / 32;
  const int st_off = (tid >> 3) * LDS_STRIDE + (tid & 7) * 4;
  const int fr_off = (lane & 31) * LDS_STRIDE + 8 * (lane >> 5);
  const int r = lane & 31, h = lane >> 5, f = dma_swizzle(r);
  for (int tile = blockIdx.x; tile < (M / 64) * tiles_n; tile += gridDim.x) {
    const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
    RowLoader<NA, RP> al(A + (int64_t)m0 * K, M - m0, K);
    DmaRowLoader<NB, RP> bl(B + (int64_t)n0 * K, N - n0, K);
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    f32x4 ra0[NA], ra1[NA];
    al.load(0, ra0);
    bl.issue(0, ldsB);
    if (1 < KS) al.load(1, ra1);
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(ldsA + st_off + i * RP * LDS_STRIDE) = ra0[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto step = [&](int ks, int cur, f32x4 (&rfree)[NA], f32x4 (&rnext)[NA]) {
      // rfree: the set whose data (step ks) is already in LDS; rnext holds step ks+1
      if (ks + 1 < KS) bl.issue(ks + 1, ldsB + (cur ^ 1) * BBUF);
      if (ks + 2 < KS) al.load(ks + 2, rfree);
      const float* pa = ldsA + cur * ABUF + (wr * 32) * LDS_STRIDE + fr_off;
      const float* pb = ldsB + cur * BBUF + (wc * 32 + r) * BK;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c0 = ((2 * h + 4 * s) ^ f) * 4, c1 = ((2 * h + 4 * s + 1) ^ f) * 4;
        const f32x4 fa0 = *reinterpret_cast<const f32x4*>(pa + 16 * s), fa1 = *reinterpret_cast<const f32x4*>(pa + 16 * s + 4);
        const f32x4 fb0 = *reinterpret_cast<const f32x4*>(pb + c0), fb1 = *reinterpret_cast<const f32x4*>(pb + c1);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc, 0, 0, 0);
      }
      if (ks + 1 < KS) {
        float* wa = ldsA + (cur ^ 1) * ABUF + st_off;
        // rnext was requested a whole step ago; the loads just issued into rfree (and B's DMA) stay in flight
        if (ks + 2 < KS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(wa + i * RP * LDS_STRIDE) = rnext[i];
      }
      // B's DMA of step ks+1 must have landed before the barrier: it is older than the A loads just issued
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
      if (ks + 2 >= KS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    };
    int ks = 0;
    while (ks < KS) {
      step(ks, 0, ra0, ra1);
      ++ks;
      if (ks >= KS) break;
      step(ks, 1, ra1, ra0);
      ++ks;
    }
    float sacc = 0.f;
    for (int q = 0; q < 16; ++q) sacc += acc[q];
    C[(int64_t)tile * 256 + tid] = sacc;
    __syncthreads();
  }
}

static void run_hyb(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
  const int lds = (2 * 64 * LDS_STRIDE + 2 * 64 * BK) * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(khyb), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(khyb, dim3(1024), dim3(256), lds, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(khyb, dim3(1024), dim3(256), lds, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-46s %8.3f ms  %6.1f TFLOP/s (%.1f %% of 157.3)\n", name, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 1.573);
}

template <int NST, int MINB>
static void run_dma(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
  auto kern = kdma<NST, MINB>;
  const int lds = NST * (64 + 64) * BK * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = MINB * 256;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-46s %8.3f ms  %6.1f TFLOP/s (%.1f %% of 157.3)\n", name, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 1.573);
}

template <int V>
static void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
  auto kern = k<V>;
  const int lds = Tile<1, 1>::LDS_BYTES;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(1024), dim3(256), lds, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(1024), dim3(256), lds, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-46s %8.3f ms  %6.1f TFLOP/s (%.1f %% of 157.3)\n", name, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 1.573);
}

int main(int argc, char** argv) {
  int M = argc > 3 ? atoi(argv[1]) : 16384, N = argc > 3 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 2304;
  float *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)(M / 64) * (N / 64) * 256 * 4));
  std::vector<float> h((size_t)M * K);
  for (auto& v : h) v = (rand() % 200 - 100) * 0.01f;
  CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
  h.resize((size_t)N * K);
  CK(hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
  printf("M=%d N=%d K=%d, 64x64 tile, 1024 persistent blocks (4 per CU)\n", M, N, K);
  for (int pass = 0; pass < 3; ++pass) {
    printf("-- pass %d\n", pass);
    run<0>("full loop (register staging, 4 blocks/CU)", A, B, C, M, N, K);
    run<16>("  + fragment reads up front", A, B, C, M, N, K);
    run_hyb("hybrid: A regs 2 steps ahead + B LDS-DMA", A, B, C, M, N, K);
    run_dma<2, 4>("LDS-DMA, 2 stages, 4 blocks/CU", A, B, C, M, N, K);
    run_dma<2, 3>("LDS-DMA, 2 stages, 3 blocks/CU", A, B, C, M, N, K);
    run_dma<3, 3>("LDS-DMA, 3 stages, 3 blocks/CU", A, B, C, M, N, K);
    run_dma<2, 2>("LDS-DMA, 2 stages, 2 blocks/CU", A, B, C, M, N, K);
    run_dma<3, 2>("LDS-DMA, 3 stages, 2 blocks/CU", A, B, C, M, N, K);
    if (pass == 0) {
      run<1>("no barrier", A, B, C, M, N, K);
      run<2>("no global loads", A, B, C, M, N, K);
      run<4>("no LDS fragment reads", A, B, C, M, N, K);
      run<2 | 8>("no global loads, no staging writes", A, B, C, M, N, K);
      run<1 | 2 | 4 | 8>("MFMA only", A, B, C, M, N, K);
    }
  }
  return 0;
}
