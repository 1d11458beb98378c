// Measurement aid (no reference counterpart): the shader clock the chip HOLDS under sustained f32-MFMA load.
// Rooflines in bench.py price against the 2.4 GHz peak (157.3 TFLOP/s); MI355X boxes settle near 2.0 GHz under the
// convolution kernels, and not all at the same value, so a run records what its own box held.
#include <hip/hip_runtime.h>

#include "../../include/dif.h"
#include "dif_internal.hpp"

namespace dif {

typedef float probe_f32x16 __attribute__((ext_vector_type(16)));

// every wave issues `iters` x 16 v_mfma_f32_32x32x2_f32 (the convolution kernels' instruction) back to back; lane 0 of
// each block reports shader-clock cycles (s_memtime) and 100 MHz ticks (s_memrealtime) over the loop
__global__ __launch_bounds__(256, 4) void mfma_clock_probe_kernel(unsigned long long* out, int iters, float seed) {
  probe_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // sixteen operand pairs of pseudo-random mantissas per lane: matrix-pipe power (and so the clock the chip settles at)
  // depends on how many operand bits toggle between instructions; constant operands read 10 % too high
  float a[16], b[16];
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + (unsigned)seed;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    h = h * 1664525u + 1013904223u;
    a[j] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;          // [-0.25, 0.25)
    h = h * 1664525u + 1013904223u;
    b[j] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (s == 12345.678f) out[0] = 0;     // keeps the accumulators live
}

}  // namespace dif

using namespace dif;

extern "C" int dif_probe_mfma_clock(double* ghz_out, double* tflops_out, void* stream) {
  if (!ghz_out) return set_error("dif_probe_mfma_clock: null output");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int dev = 0;
  DIF_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  DIF_HIP(hipGetDeviceProperties(&prop, dev));
  const int blocks = 4 * prop.multiProcessorCount;         // four waves on every SIMD, as under the convolutions
  const int iters = 10000;                                 // 4 waves x 10000 x 16 MFMAs x 64 cycles = 41 M cycles/SIMD: ~20 ms
  unsigned long long* d = nullptr;
  DIF_HIP(hipMalloc(&d, (size_t)blocks * 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  DIF_HIP(hipEventCreate(&e0));
  DIF_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_clock_probe_kernel, dim3(blocks), dim3(256), 0, st, d, iters, 1.0f);   // lets the clock governor settle
  DIF_HIP(hipEventRecord(e0, st));
  hipLaunchKernelGGL(mfma_clock_probe_kernel, dim3(blocks), dim3(256), 0, st, d, iters, 2.0f);
  DIF_HIP(hipEventRecord(e1, st));
  DIF_HIP(hipGetLastError());
  DIF_HIP(hipStreamSynchronize(st));
  float ms = 0.f;
  DIF_HIP(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long* h = new unsigned long long[(size_t)blocks * 2];
  const hipError_t ce = hipMemcpy(h, d, (size_t)blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (ce != hipSuccess) {
    delete[] h;
    return set_error("dif_probe_mfma_clock: copy failed: %s", hipGetErrorString(ce));
  }
  double cyc = 0, ticks = 0;
  for (int b = 0; b < blocks; ++b) {
    cyc += (double)h[2 * b];
    ticks += (double)h[2 * b + 1];
  }
  delete[] h;
  *ghz_out = ticks > 0 ? cyc / (ticks * 10.0) : 0.0;       // cycles per 10 ns tick -> GHz
  if (tflops_out)                                          // what the probe itself sustained (its ceiling is the f32-MFMA peak)
    *tflops_out = ms > 0 ? (double)blocks * 4 * iters * 16 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12 : 0.0;
  return 0;
}
