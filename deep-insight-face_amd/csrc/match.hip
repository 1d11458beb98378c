// 1:N gallery match and the small distance kernels.
//
// Semantics (deep_insight_face/evaluation/utility.py:52-66 broadcast over gallery rows
// + np.argmin, see SURVEY.md section 3 "1:N gallery search"):
//   metric 0: d(q, g) = sum((q - g)^2)                      -> argmin
//   metric 1: d(q, g) = arccos(q.g / (|q| |g|)) / pi        -> argmin
// The B x G distance matrix is never materialised.  match_tile_kernel forms
// dot(q, g) tiles on the f32 MFMA (gallery rows on the M side, probes on the lanes),
// turns each dot into a monotone search key in the epilogue
//   metric 1: key = -dot / |g|        (|q| is constant per probe; arccos is decreasing)
//   metric 0: key = |g|^2 - 2 dot     (|q|^2 is constant per probe)
// and keeps a running (key, index) minimum per probe, lowest index on ties (what
// np.argmin returns).  match_finish_kernel reduces the per-block partials and
// recomputes the winner's distance with the reference's own formula from the two
// rows, so the reported distance does not carry the key's cancellation error.
#include "gemm_core.hpp"
#include "dif_internal.hpp"

#include <stdlib.h>

namespace dif {

__device__ __forceinline__ bool better(float k, int i, float bk, int bi) {
  return (k < bk) || (k == bk && i < bi);
}

template <class T>
__global__ __launch_bounds__(T::NT, 2) void match_tile_kernel(const float* __restrict__ gallery, int64_t G,
                                                            const float* __restrict__ probes, int B, int D,
                                                            const float* __restrict__ aux, int metric,
                                                            float* __restrict__ part_key,
                                                            int* __restrict__ part_idx) {
  constexpr int WM = T::WM, WN = T::WN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int p0 = blockIdx.y * T::BN;
  const int ksteps = D / BK;
  const int64_t gtiles = (G + T::BM - 1) / T::BM;

  float bkey[WN];
  int bidx[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    bkey[n] = __builtin_inff();
    bidx[n] = 0x7fffffff;
  }

  for (int64_t gt = blockIdx.x; gt < gtiles; gt += gridDim.x) {
    const int64_t g0 = gt * T::BM;
    f32x16 acc[WM][WN];
    zero_acc<T>(acc);
    RowLoader<T::NA, T::RP> al(gallery + g0 * D, G - g0, D);
    RowLoader<T::NB, T::RP> bl(probes + (int64_t)p0 * D, (int64_t)B - p0, D);
    gemm_mainloop<T>(al, bl, 0, ksteps, smem, acc);

    // aux[g] = -1/|g| (metric 1) or |g|^2 (metric 0); rows past G read 0 through the
    // descriptor and are masked out of the search.
    const int64_t rows_left = G - g0;
    const __amdgpu_buffer_rsrc_t arsrc =
        make_rsrc(aux + g0, (uint32_t)((rows_left < T::BM ? rows_left : T::BM) * 4));
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      const int rbase = (wr * WM + m) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // rows rbase + 8q .. +3 <-> registers 4q .. 4q+3
        const f32x4 ax = buf_load4(arsrc, (uint32_t)(rbase + 8 * q) * 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rl = rbase + 8 * q + j;
          const bool ok = rl < rows_left;
#pragma unroll
          for (int n = 0; n < WN; ++n) {
            const float dot = acc[m][n][4 * q + j];
            float key = (metric == 1) ? dot * ax[j] : fmaf(-2.f, dot, ax[j]);
            key = ok ? key : __builtin_inff();
            if (key < bkey[n]) {   // rows ascend within a lane: strict < keeps the lowest index
              bkey[n] = key;
              bidx[n] = (int)g0 + rl;
            }
          }
        }
      }
    }
  }

  // lanes l and l+32 hold the same probe column
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const float ok = __shfl_xor(bkey[n], 32);
    const int oi = __shfl_xor(bidx[n], 32);
    if (better(ok, oi, bkey[n], bidx[n])) {
      bkey[n] = ok;
      bidx[n] = oi;
    }
  }
  // the WGM waves of a column share probe columns: merge through LDS (the mainloop ended on a barrier)
  float* skey = smem;                                             // [WGM][BN]
  int* sidx = reinterpret_cast<int*>(smem + T::WGM * T::BN);      // [WGM][BN]
  if (lane < 32) {
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int c = (wc * WN + n) * 32 + lane;
      skey[wr * T::BN + c] = bkey[n];
      sidx[wr * T::BN + c] = bidx[n];
    }
  }
  __syncthreads();
  if (wr == 0 && lane < 32) {
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int c = (wc * WN + n) * 32 + lane;
#pragma unroll
      for (int w = 1; w < T::WGM; ++w) {
        const float ok = skey[w * T::BN + c];
        const int oi = sidx[w * T::BN + c];
        if (better(ok, oi, bkey[n], bidx[n])) {
          bkey[n] = ok;
          bidx[n] = oi;
        }
      }
      const int p = p0 + c;
      if (p < B) {
        part_key[(int64_t)blockIdx.x * B + p] = bkey[n];
        part_idx[(int64_t)blockIdx.x * B + p] = bidx[n];
      }
    }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Reference formulas on one pair of rows, evaluated by one wave.
// metric 0: utility.py:54-56; metric 1: utility.py:58-62 (float32 throughout, the
// similarity is clamped to [-1, 1] before arccos -- see DESIGN.md "NaN at s > 1").
__device__ __forceinline__ float pair_distance(const float* a, const float* b, int D, int metric, int lane) {
  if (metric == 0) {
    float s = 0.f;
    for (int k = lane; k < D; k += 64) {
      const float d = a[k] - b[k];
      s = fmaf(d, d, s);
    }
    return wave_sum(s);
  }
  float dot = 0.f, aa = 0.f, bb = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float x = a[k], y = b[k];
    dot = fmaf(x, y, dot);
    aa = fmaf(x, x, aa);
    bb = fmaf(y, y, bb);
  }
  dot = wave_sum(dot);
  aa = wave_sum(aa);
  bb = wave_sum(bb);
  float s = dot / (sqrtf(aa) * sqrtf(bb));
  s = fminf(1.f, fmaxf(-1.f, s));
  return acosf(s) / 3.14159274101257324f;
}

// one wave per probe
__global__ __launch_bounds__(64) void match_finish_kernel(const float* __restrict__ part_key,
                                                          const int* __restrict__ part_idx, int nparts, int B,
                                                          const float* __restrict__ probes,
                                                          const float* __restrict__ gallery, int D, int metric,
                                                          int64_t index_base, int64_t* __restrict__ idx_out,
                                                          float* __restrict__ dist_out,
                                                          float* __restrict__ key_out) {
  const int p = blockIdx.x, lane = threadIdx.x;
  float bk = __builtin_inff();
  int bi = 0x7fffffff;
  for (int t = lane; t < nparts; t += 64) {
    const float k = part_key[(int64_t)t * B + p];
    const int i = part_idx[(int64_t)t * B + p];
    if (better(k, i, bk, bi)) {
      bk = k;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ok = __shfl_xor(bk, o);
    const int oi = __shfl_xor(bi, o);
    if (better(ok, oi, bk, bi)) {
      bk = ok;
      bi = oi;
    }
  }
  const bool found = bi != 0x7fffffff;
  float d = __builtin_nanf("");
  if (found) d = pair_distance(probes + (int64_t)p * D, gallery + (int64_t)bi * D, D, metric, lane);
  if (lane == 0) {
    idx_out[p] = found ? index_base + bi : -1;
    dist_out[p] = d;
    if (key_out) key_out[p] = bk;   // per-probe constants only: comparable across gallery shards
  }
}

// |g|^2 and -1/|g| per gallery row; one wave per row.
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ rows, int64_t n, int D,
                                                        float* __restrict__ sq, float* __restrict__ ninv) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float x = rows[r * D + k];
    s = fmaf(x, x, s);
  }
  s = wave_sum(s);
  if (lane == 0) {
    sq[r] = s;
    ninv[r] = -1.f / sqrtf(s);
  }
}

// Row-paired distance (evaluation/utility.py:52-66) ; rows of e2 broadcast when n2 == 1.
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ e1, const float* __restrict__ e2,
                                                       int64_t n, int64_t n1, int64_t n2, int D, int metric,
                                                       float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const float* a = e1 + (n1 == 1 ? 0 : r) * D;
  const float* b = e2 + (n2 == 1 ? 0 : r) * D;
  const float d = pair_distance(a, b, D, metric, lane);
  if (lane == 0) out[r] = d;
}

// Merge R per-shard results (keys comparable across shards), lowest global index on ties.
__global__ __launch_bounds__(256) void match_merge_kernel(const float* __restrict__ keys,
                                                          const int64_t* __restrict__ idx,
                                                          const float* __restrict__ dist, int R, int B,
                                                          int64_t* __restrict__ idx_out,
                                                          float* __restrict__ dist_out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= B) return;
  float bk = __builtin_inff();
  int64_t bi = -1;
  float bd = __builtin_nanf("");
  for (int r = 0; r < R; ++r) {
    const float k = keys[(int64_t)r * B + p];
    const int64_t i = idx[(int64_t)r * B + p];
    if (i < 0) continue;
    if (bi < 0 || k < bk || (k == bk && i < bi)) {
      bk = k;
      bi = i;
      bd = dist[(int64_t)r * B + p];
    }
  }
  idx_out[p] = bi;
  dist_out[p] = bd;
}

}  // namespace dif

using namespace dif;

namespace dif {

template <class T>
static int launch_match_tiles(const Gallery* g, const float* probes, int B, int metric, int nparts,
                              hipStream_t st) {
  static bool attr_set = false;
  auto kern = match_tile_kernel<T>;
  if (!attr_set) {
    DIF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                T::LDS_BYTES));
    attr_set = true;
  }
  dim3 grid(nparts, (B + T::BN - 1) / T::BN);
  hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, st, g->rows, g->n, probes, B, g->d,
                     metric == 1 ? g->ninv : g->sq, metric, g->part_key, g->part_idx);
  DIF_HIP(hipGetLastError());
  return 0;
}

static int match_tile_kind(int B) {
  // 0: 128 gallery rows x 128 probes, 1: 128 x 64, 3: 64 x 64, 4: 128 x 32 (four waves stacked on M)
  static const int forced = getenv("DIF_MATCH_TILE") ? atoi(getenv("DIF_MATCH_TILE")) : -1;
  if (forced >= 0) return forced;
  // unlike the convolutions, the 128-row tiles win here (the arg-min epilogue and the probe
  // re-reads weigh more on small tiles: 117 vs 99 TF at 256 x 1M, 128 vs 120 TF at 4096 x 125k)
  // B <= 32: the 32-probe tile halves the padded MFMA work, which at 1M rows would otherwise
  // (67 GFLOP for 64 padded probes) cost more than streaming the gallery from HBM
  if (B <= 32) return 4;
  return B <= 64 ? 1 : 0;
}

int match_plan_parts(const Gallery* g, int B) {
  // One gallery tile per block while the partial buffer stays small; beyond that the blocks
  // stride over tiles (the imbalance of a long stride is negligible).
  const int kind = match_tile_kind(B);
  const int BM = kind == 3 ? 64 : 128;   // kinds 0, 1 and 4 all take 128 gallery rows per tile
  const int64_t gtiles = (g->n + BM - 1) / BM;
  int64_t cap = (int64_t)(32 << 20) / ((int64_t)B * 8);   // <= 32 MiB of (key, idx) partials
  if (cap < 1024) cap = 1024;
  int64_t want = gtiles < cap ? gtiles : cap;
  if (want < 1) want = 1;
  return (int)want;
}

int match_run(Gallery* g, const float* probes, int B, int metric, int64_t* idx_out, float* dist_out,
              float* key_out, hipStream_t st) {
  if (g->n <= 0) return set_error("dif_match: gallery is empty");
  if (B <= 0) return 0;
  const int nparts = match_plan_parts(g, B);
  const size_t need = (size_t)nparts * B;
  if (need > g->part_cap) {
    DIF_HIP(hipStreamSynchronize(st));
    if (g->part_key) DIF_HIP(hipFree(g->part_key));
    if (g->part_idx) DIF_HIP(hipFree(g->part_idx));
    g->part_key = nullptr;
    g->part_idx = nullptr;
    DIF_HIP(hipMalloc(&g->part_key, need * sizeof(float)));
    DIF_HIP(hipMalloc(&g->part_idx, need * sizeof(int)));
    g->part_cap = need;
  }
  int rc;
  const int kind = match_tile_kind(B);
  if (kind == 3)
    rc = launch_match_tiles<Tile<1, 1>>(g, probes, B, metric, nparts, st);
  else if (kind == 4 && B <= 32)
    rc = launch_match_tiles<Tile<1, 1, 4, 1>>(g, probes, B, metric, nparts, st);
  else if (kind == 1 || B <= 64)
    rc = launch_match_tiles<Tile<2, 1>>(g, probes, B, metric, nparts, st);
  else
    rc = launch_match_tiles<Tile<2, 2>>(g, probes, B, metric, nparts, st);
  if (rc) return rc;
  hipLaunchKernelGGL(match_finish_kernel, dim3(B), dim3(64), 0, st, g->part_key, g->part_idx, nparts, B, probes,
                     g->rows, g->d, metric, g->index_base, idx_out, dist_out, key_out);
  DIF_HIP(hipGetLastError());
  return 0;
}

int gallery_norms(Gallery* g, hipStream_t st) {
  if (g->n == 0) return 0;
  hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)((g->n + 3) / 4)), dim3(256), 0, st, g->rows, g->n, g->d,
                     g->sq, g->ninv);
  DIF_HIP(hipGetLastError());
  return 0;
}

int pairwise_run(const float* e1, int64_t n1, const float* e2, int64_t n2, int D, int metric, float* out,
                 hipStream_t st) {
  const int64_t n = n1 > n2 ? n1 : n2;
  if (n == 0) return 0;
  hipLaunchKernelGGL(pairwise_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, e1, e2, n, n1, n2, D,
                     metric, out);
  DIF_HIP(hipGetLastError());
  return 0;
}

int match_merge_run(const float* keys, const int64_t* idx, const float* dist, int R, int B, int64_t* idx_out,
                    float* dist_out, hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(match_merge_kernel, dim3((B + 255) / 256), dim3(256), 0, st, keys, idx, dist, R, B, idx_out,
                     dist_out);
  DIF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dif
