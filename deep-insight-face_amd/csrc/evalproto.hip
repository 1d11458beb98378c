// LFW-protocol threshold sweep on the device: the inner loops
// of calculate_roc / calculate_val (deep_insight_face/evaluation/utility.py:104-107,153-161)
// evaluate calculate_accuracy / calculate_val_far for every (fold, threshold) pair -- 10 folds x
// 400 / 4000 thresholds passes over the distance vector in NumPy.  Here one pass per threshold
// block counts, per test fold, the accepted same / different pairs:
//     counts[f][t][0] = #{ i in fold f : dist[i] < thr[t] and     issame[i] }   (true accepts)
//     counts[f][t][1] = #{ i in fold f : dist[i] < thr[t] and not issame[i] }   (false accepts)
// Train-split numbers are totals minus the fold's own; tp/fp/tn/fn and every ratio of the
// reference follow from these two integers and the fold's class sizes (host side, exact).
#include "../../include/dif.h"
#include "dif_internal.hpp"

namespace dif {

constexpr int MAX_FOLDS = 64;

__global__ __launch_bounds__(256) void threshold_counts_kernel(const float* __restrict__ dist,
                                                               const uint8_t* __restrict__ issame,
                                                               const int* __restrict__ fold, int64_t n,
                                                               const double* __restrict__ thr, int T, int nfolds,
                                                               int* __restrict__ counts) {
  __shared__ int acc[MAX_FOLDS * 2];
  const int t = blockIdx.x;
  if (t >= T) return;
  for (int i = threadIdx.x; i < nfolds * 2; i += 256) acc[i] = 0;
  __syncthreads();
  // the reference compares float32 distances with Python-float thresholds: np.less promotes to
  // float64, so compare in double
  const double th = thr[t];
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    if ((double)dist[i] < th) atomicAdd(&acc[fold[i] * 2 + (issame[i] ? 0 : 1)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nfolds * 2; i += 256) {
    const int f = i >> 1, k = i & 1;
    counts[((int64_t)f * T + t) * 2 + k] = acc[i];
  }
}

}  // namespace dif

using namespace dif;

extern "C" int dif_threshold_counts(const float* dist_dev, const uint8_t* issame_dev, const int32_t* fold_dev,
                                    int64_t n, const double* thresholds_dev, int n_thresholds, int n_folds,
                                    int32_t* counts_dev, void* stream) {
  if (n < 0 || n_thresholds < 0) return set_error("dif_threshold_counts: negative size");
  if (n_folds < 1 || n_folds > MAX_FOLDS) return set_error("dif_threshold_counts: n_folds must be in [1, %d]", MAX_FOLDS);
  if (n_thresholds == 0) return 0;
  if (!counts_dev || !thresholds_dev || (n > 0 && (!dist_dev || !issame_dev || !fold_dev)))
    return set_error("dif_threshold_counts: null pointer");
  hipLaunchKernelGGL(threshold_counts_kernel, dim3(n_thresholds), dim3(256), 0, (hipStream_t)stream, dist_dev,
                     issame_dev, fold_dev, n, thresholds_dev, n_thresholds, n_folds, counts_dev);
  DIF_HIP(hipGetLastError());
  return 0;
}
