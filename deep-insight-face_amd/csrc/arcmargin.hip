// ArcMargin (ArcFace) logits: logits[b][c] = s * cos(theta_bc + m * [c == label_b]).
// Not in the reference (north_star only; Deng et al. 2019).  One f32-MFMA GEMM of the
// embeddings (M side) against the class centres (N side, lanes <-> classes so that a
// wave stores 128-byte runs of a logits row), normalisation and margin in the epilogue.
#include "gemm_core.hpp"
#include "dif_internal.hpp"
#include "arcmargin.hpp"

namespace dif {

// 1/|row| per row (rows of zeros -> 1/sqrt(1e-12), like the oracle's max(sum, 1e-12))
__global__ __launch_bounds__(256) void inv_norm_kernel(const float* __restrict__ rows, int64_t n, int D,
                                                       float* __restrict__ inv) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float x = rows[r * D + k];
    s = fmaf(x, x, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) inv[r] = 1.f / sqrtf(fmaxf(s, 1e-12f));
}

template <class T>
__global__ __launch_bounds__(T::NT, 2) void arcmargin_kernel(const float* __restrict__ emb, int B,
                                                           const float* __restrict__ w, int64_t C, int D,
                                                           const float* __restrict__ einv,
                                                           const float* __restrict__ winv,
                                                           const int64_t* __restrict__ labels, float s, float cm,
                                                           float sm, float th, float mm,
                                                           float* __restrict__ logits) {
  constexpr int WM = T::WM, WN = T::WN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int64_t c0 = (int64_t)blockIdx.x * T::BN;
  const int b0 = blockIdx.y * T::BM;

  f32x16 acc[WM][WN];
  zero_acc<T>(acc);
  RowLoader<T::NA, T::RP> al(emb + (int64_t)b0 * D, (int64_t)B - b0, D);
  RowLoader<T::NB, T::RP> bl(w + c0 * D, C - c0, D);
  gemm_mainloop<T>(al, bl, 0, D / BK, smem, acc);

#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int64_t c = c0 + (wc * WN + n) * 32 + (lane & 31);
    const bool cok = c < C;
    const float wi = cok ? winv[c] : 0.f;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = b0 + (wr * WM + m) * 32 + frag_row(lane, r);
        if (b < B && cok) {
          float cs = acc[m][n][r] * einv[b] * wi;
          if (labels && labels[b] == c) {
            const float sine = sqrtf(fminf(fmaxf(1.f - cs * cs, 0.f), 1.f));
            const float phi = cs * cm - sine * sm;
            cs = (cs > th) ? phi : cs - mm;
          }
          logits[(int64_t)b * C + c] = cs * s;
        }
      }
    }
  }
}

int arcmargin_prepare(ArcMargin* a, hipStream_t st) {
  hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)((a->C + 3) / 4)), dim3(256), 0, st, a->w, a->C, a->d, a->winv);
  DIF_HIP(hipGetLastError());
  return 0;
}

int arcmargin_run(ArcMargin* a, const float* emb, const int64_t* labels, int B, float* logits, hipStream_t st) {
  using T = Tile<2, 2>;
  if (B > a->einv_cap) {
    DIF_HIP(hipStreamSynchronize(st));
    if (a->einv) DIF_HIP(hipFree(a->einv));
    a->einv = nullptr;
    DIF_HIP(hipMalloc(&a->einv, (size_t)B * sizeof(float)));
    a->einv_cap = B;
  }
  hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, emb, (int64_t)B, a->d,
                     a->einv);
  DIF_HIP(hipGetLastError());
  static bool attr_set = false;
  auto kern = arcmargin_kernel<T>;
  if (!attr_set) {
    DIF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                T::LDS_BYTES));
    attr_set = true;
  }
  dim3 grid((unsigned)((a->C + T::BN - 1) / T::BN), (unsigned)((B + T::BM - 1) / T::BM));
  hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, st, emb, B, a->w, a->C, a->d, a->einv, a->winv, labels,
                     a->s, cosf(a->m), sinf(a->m), cosf(3.14159265358979323846f - a->m),
                     sinf(3.14159265358979323846f - a->m) * a->m, logits);
  DIF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dif
