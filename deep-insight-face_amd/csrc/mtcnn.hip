// hipcc-flags: -ffp-contract=off
// MTCNN cascade glue on the device (BASELINE configs[4] as worded; NOT in the reference: config.py:37 and
// detector/run.py:124 name it in comments only -- the detector the reference ships is YOLOv3-face, detector.hip).
// The three networks run on the convolution kernels (net.hip: build_mtcnn); what is here is the arithmetic BETWEEN them,
// with static shapes so that a batch of frames goes through the whole cascade without a host round trip: every frame has
// a fixed number of slots per stage, an empty slot carries score -1.
//   mtcnn_propose_kernel   P-Net head map -> one proposal per cell (box in frame pixels, P(face) or -1 below the threshold)
//   (dif_nms, detector.hip, does every suppression: scores >= 0 take part, best first, ties by lower index)
//   mtcnn_gather_kernel    kept slot indices -> boxes / scores / regression values of the next stage's slots, optionally
//                          calibrated (regression, squaring, truncation)
//   mtcnn_rescore_kernel   R-Net / O-Net outputs -> slot scores (P(face) where the slot was alive and passes the threshold)
//                          and regression values; optionally the last stage's plain regression of the boxes
// No FMA contraction in this file: the tests compare the boxes with the NumPy restatement's float32 arithmetic bit for bit.
#include "../../include/dif.h"
#include "dif_internal.hpp"

namespace dif {

__device__ __forceinline__ float face_prob(float l0, float l1) { return 1.f / (1.f + expf(l0 - l1)); }

__global__ __launch_bounds__(256) void mtcnn_propose_kernel(const float* __restrict__ head, int n, int gh, int gw, int ld,
                                                            float inv_scale, float thr, float* __restrict__ boxes,
                                                            float* __restrict__ scores) {
  const int64_t total = (int64_t)n * gh * gw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cell = (int)(i % ((int64_t)gh * gw));
    const int gx = cell % gw, gy = cell / gw;
    const float* h = head + i * ld;
    const float p = face_prob(h[0], h[1]);
    float* b = boxes + i * 4;
    b[0] = truncf((2.f * (float)gx + 1.f) * inv_scale);
    b[1] = truncf((2.f * (float)gy + 1.f) * inv_scale);
    b[2] = truncf((2.f * (float)gx + 12.f) * inv_scale);
    b[3] = truncf((2.f * (float)gy + 12.f) * inv_scale);
    scores[i] = p >= thr ? p : -1.f;
  }
}

__device__ __forceinline__ void calibrate_square(const float* b, const float* r, float* o) {
  const float w = b[2] - b[0] + 1.f, h = b[3] - b[1] + 1.f;
  float x1 = b[0] + r[0] * w, y1 = b[1] + r[1] * h;
  const float x2 = b[2] + r[2] * w, y2 = b[3] + r[3] * h;
  const float ww = x2 - x1, hh = y2 - y1;
  const float side = fmaxf(ww, hh);
  x1 = x1 + ww * 0.5f - side * 0.5f;
  y1 = y1 + hh * 0.5f - side * 0.5f;
  o[0] = truncf(x1);
  o[1] = truncf(y1);
  o[2] = truncf(x1 + side);
  o[3] = truncf(y1 + side);
}

// dst slot (f, dst_off + j) <- src slot (f, keep[f][j]); keep < 0: an empty slot (zeros, score -1)
__global__ __launch_bounds__(256) void mtcnn_gather_kernel(const int* __restrict__ keep, int n, int k, const float* __restrict__ sboxes,
                                                           const float* __restrict__ sscores, const float* __restrict__ sreg,
                                                           int sreg_ld, int nsrc, float* __restrict__ dboxes,
                                                           float* __restrict__ dscores, float* __restrict__ dreg, int ndst,
                                                           int dst_off, int calibrate) {
  const int total = n * k;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int f = i / k, j = i - f * k;
    const int src = keep[i];
    const int64_t d = (int64_t)f * ndst + dst_off + j;
    float b[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f}, s = -1.f;
    if (src >= 0) {
      const int64_t q = (int64_t)f * nsrc + src;
      for (int e = 0; e < 4; ++e) b[e] = sboxes[q * 4 + e];
      if (sreg)
        for (int e = 0; e < 4; ++e) r[e] = sreg[q * sreg_ld + e];
      s = sscores[q];
      if (calibrate) {
        float o[4];
        calibrate_square(b, r, o);
        for (int e = 0; e < 4; ++e) b[e] = o[e];
      }
    }
    for (int e = 0; e < 4; ++e) dboxes[d * 4 + e] = b[e];
    if (dreg)
      for (int e = 0; e < 4; ++e) dreg[d * 4 + e] = r[e];
    dscores[d] = s;
  }
}

// out: [slots][ld] network outputs (logits 2 | box 4 | ...); scores in/out per slot; reg out [slots][4];
// plain != 0: boxes are regressed in place (no squaring: the cascade's last step)
__global__ __launch_bounds__(256) void mtcnn_rescore_kernel(const float* __restrict__ out, int slots, int ld, float thr,
                                                            float* __restrict__ scores, float* __restrict__ reg,
                                                            float* __restrict__ boxes, int plain) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < slots; i += gridDim.x * 256) {
    const float* o = out + (int64_t)i * ld;
    const float p = face_prob(o[0], o[1]);
    const float s = scores[i];
    scores[i] = (s >= 0.f && p >= thr) ? p : -1.f;
    for (int e = 0; e < 4; ++e) reg[(int64_t)i * 4 + e] = o[2 + e];
    if (plain) {
      float* b = boxes + (int64_t)i * 4;
      const float w = b[2] - b[0] + 1.f, h = b[3] - b[1] + 1.f;
      const float x1 = b[0] + o[2] * w, y1 = b[1] + o[3] * h, x2 = b[2] + o[4] * w, y2 = b[3] + o[5] * h;
      b[0] = x1;
      b[1] = y1;
      b[2] = x2;
      b[3] = y2;
    }
  }
}

}  // namespace dif

using namespace dif;

extern "C" {

int dif_mtcnn_propose(const float* head_dev, int n, int gh, int gw, int ld, float scale, float threshold, float* boxes_dev,
                      float* scores_dev, void* stream) {
  if (n < 0 || gh < 1 || gw < 1 || ld < 6 || !(scale > 0.f)) return set_error("dif_mtcnn_propose: bad sizes");
  if (n == 0) return 0;
  if (!head_dev || !boxes_dev || !scores_dev) return set_error("dif_mtcnn_propose: null pointer");
  int64_t blocks = ((int64_t)n * gh * gw + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mtcnn_propose_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, head_dev, n, gh, gw, ld,
                     1.f / scale, threshold, boxes_dev, scores_dev);
  DIF_HIP(hipGetLastError());
  return 0;
}

int dif_mtcnn_gather(const int32_t* keep_dev, int n, int k, const float* src_boxes_dev, const float* src_scores_dev,
                     const float* src_reg_dev, int src_reg_ld, int n_src, float* dst_boxes_dev, float* dst_scores_dev,
                     float* dst_reg_dev, int n_dst, int dst_offset, int calibrate, void* stream) {
  if (n < 0 || k < 1 || n_src < 1 || n_dst < k || dst_offset < 0 || dst_offset + k > n_dst)
    return set_error("dif_mtcnn_gather: bad sizes");
  if (n == 0) return 0;
  if (!keep_dev || !src_boxes_dev || !src_scores_dev || !dst_boxes_dev || !dst_scores_dev)
    return set_error("dif_mtcnn_gather: null pointer");
  if (calibrate && !src_reg_dev) return set_error("dif_mtcnn_gather: calibration needs the regression values");
  int blocks = (n * k + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mtcnn_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, keep_dev, n, k, src_boxes_dev,
                     src_scores_dev, src_reg_dev, src_reg_ld, n_src, dst_boxes_dev, dst_scores_dev, dst_reg_dev, n_dst, dst_offset,
                     calibrate);
  DIF_HIP(hipGetLastError());
  return 0;
}

int dif_mtcnn_rescore(const float* out_dev, int slots, int ld, float threshold, float* scores_dev, float* reg_dev,
                      float* boxes_dev, int plain_regression, void* stream) {
  if (slots < 0 || ld < 6) return set_error("dif_mtcnn_rescore: bad sizes");
  if (slots == 0) return 0;
  if (!out_dev || !scores_dev || !reg_dev || (plain_regression && !boxes_dev)) return set_error("dif_mtcnn_rescore: null pointer");
  int blocks = (slots + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mtcnn_rescore_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out_dev, slots, ld, threshold,
                     scores_dev, reg_dev, boxes_dev, plain_regression);
  DIF_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
