// YOLOv3-face post-processing on the device: the box decode of
// deep_insight_face/detector/yolov3.py:36-106 (yolo_head, correct_boxes, boxes_and_scores) and
// the per-class score filter + greedy non-max suppression of :122-172 (get_yolo_output, which
// calls tf.image.non_max_suppression).  Memory-bound / tiny: one thread per anchor box for the
// decode, one block per (image, class) for the suppression.
#include "../../include/dif.h"
#include "dif_internal.hpp"

namespace dif {
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct YoloLayer {
  const float* feats;   // [N][gh][gw][na*(5+C)]
  int gh, gw, offset;   // offset = index of this layer's first box in the concatenated list
  float aw[3], ah[3];   // anchors (pixels of the network input)
};
struct YoloArgs {
  YoloLayer L[3];
  int nlayers, N, C, ntot;
  float in_h, in_w;
  const float* image_shape;   // [N][2] = (height, width) of every original image
  float* boxes;               // [N][ntot][4]  y_min, x_min, y_max, x_max (image pixels)
  float* scores;              // [N][ntot][C]
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void yolo_decode_kernel(const YoloArgs a) {
  const int64_t total = (int64_t)a.N * a.ntot;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / a.ntot);
    int b = (int)(i - (int64_t)n * a.ntot);
    int l = 0;
    while (l + 1 < a.nlayers && b >= a.L[l + 1].offset) ++l;
    const YoloLayer& L = a.L[l];
    b -= L.offset;
    const int an = b % 3;
    const int cell = b / 3;
    const int gx = cell % L.gw, gy = cell / L.gw;
    const int stride = 5 + a.C;
    const float* f = L.feats + (((int64_t)n * L.gh + gy) * L.gw + gx) * (3 * stride) + an * stride;
    // yolo_head (yolov3.py:55-60)
    const float bx = (sigmoidf(f[0]) + (float)gx) / (float)L.gw;
    const float by = (sigmoidf(f[1]) + (float)gy) / (float)L.gh;
    const float bw = expf(f[2]) * L.aw[an] / a.in_w;
    const float bh = expf(f[3]) * L.ah[an] / a.in_h;
    const float conf = sigmoidf(f[4]);
    // correct_boxes (yolov3.py:72-92)
    const float ih = a.image_shape[2 * n], iw = a.image_shape[2 * n + 1];
    const float r = fminf(a.in_h / ih, a.in_w / iw);
    const float nh = rintf(ih * r), nw = rintf(iw * r);
    const float oy = (a.in_h - nh) / 2.f / a.in_h, ox = (a.in_w - nw) / 2.f / a.in_w;
    const float sy = a.in_h / nh, sx = a.in_w / nw;
    const float cy = (by - oy) * sy, cx = (bx - ox) * sx;
    const float hh = bh * sy, ww = bw * sx;
    float* o = a.boxes + i * 4;
    o[0] = (cy - hh / 2.f) * ih;
    o[1] = (cx - ww / 2.f) * iw;
    o[2] = (cy + hh / 2.f) * ih;
    o[3] = (cx + ww / 2.f) * iw;
    for (int c = 0; c < a.C; ++c) a.scores[i * a.C + c] = conf * sigmoidf(f[5 + c]);   // boxes_and_scores :104
  }
}

// IoU with the corner normalisation of tf.image.non_max_suppression
__device__ __forceinline__ float box_iou(const float* p, const float* q) {
  const float py0 = fminf(p[0], p[2]), py1 = fmaxf(p[0], p[2]), px0 = fminf(p[1], p[3]), px1 = fmaxf(p[1], p[3]);
  const float qy0 = fminf(q[0], q[2]), qy1 = fmaxf(q[0], q[2]), qx0 = fminf(q[1], q[3]), qx1 = fmaxf(q[1], q[3]);
  const float ap = (py1 - py0) * (px1 - px0), aq = (qy1 - qy0) * (qx1 - qx0);
  if (ap <= 0.f || aq <= 0.f) return 0.f;
  const float ih = fmaxf(fminf(py1, qy1) - fmaxf(py0, qy0), 0.f);
  const float iw = fmaxf(fminf(px1, qx1) - fmaxf(px0, qx0), 0.f);
  const float inter = ih * iw;
  return inter / (ap + aq - inter);
}

// One block per (image, class).  Greedy: pick the best remaining score (ties: lower index, the
// order of tf.image.non_max_suppression), keep it, drop everything with IoU > threshold.
// NT threads per block: 256 for YOLOv3-face's 10 647 boxes per image; 1024 for MTCNN's dense P-Net grids (26 k cells per frame
// at the first pyramid scale, up to 64 picks: the scans were 39 % of that workload's GPU time at 256 threads)
template <int NT>
__global__ __launch_bounds__(NT) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                 int ntot, int C, int max_boxes, float score_thr, float iou_thr,
                                                 uint8_t* __restrict__ alive, int* __restrict__ keep_idx,
                                                 int* __restrict__ keep_n) {
  constexpr int NW = NT / 64;
  __shared__ float s_score[NW];
  __shared__ int s_idx[NW];
  __shared__ int s_pick;
  const int n = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* bx = boxes + (int64_t)n * ntot * 4;
  const float* sc = scores + (int64_t)n * ntot * C + c;
  uint8_t* al = alive + ((int64_t)n * C + c) * ntot;
  int* out = keep_idx + ((int64_t)n * C + c) * max_boxes;
  for (int i = tid; i < ntot; i += NT) al[i] = sc[(int64_t)i * C] >= score_thr ? 1 : 0;   // mask = box_scores >= thr
  for (int i = tid; i < max_boxes; i += NT) out[i] = -1;
  __syncthreads();
  int kept = 0;
  while (kept < max_boxes) {
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int i = tid; i < ntot; i += NT) {
      if (al[i]) {
        const float s = sc[(int64_t)i * C];
        if (s > best || (s == best && i < bi)) {
          best = s;
          bi = i;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (os > best || (os == best && oi < bi)) {
        best = os;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_score[wave] = best;
      s_idx[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      float b = s_score[0];
      int k = s_idx[0];
      for (int w = 1; w < NW; ++w)
        if (s_score[w] > b || (s_score[w] == b && s_idx[w] < k)) {
          b = s_score[w];
          k = s_idx[w];
        }
      s_pick = k;
    }
    __syncthreads();
    const int pick = s_pick;
    if (pick == 0x7fffffff) break;      // nothing left
    if (tid == 0) out[kept] = pick;
    ++kept;
    float pb[4] = {bx[(int64_t)pick * 4], bx[(int64_t)pick * 4 + 1], bx[(int64_t)pick * 4 + 2], bx[(int64_t)pick * 4 + 3]};
    for (int i = tid; i < ntot; i += NT)
      if (al[i] && (i == pick || box_iou(bx + (int64_t)i * 4, pb) > iou_thr)) al[i] = 0;
    __syncthreads();
  }
  if (tid == 0) keep_n[n * C + c] = kept;
}

// The same greedy suppression for ONE class and up to NT * IPT boxes per image, with every box and score held in REGISTERS:
// a thread owns boxes tid, tid + NT, ... (IPT of them); a pick is a local arg-max over its registers, a block reduction,
// and a suppression pass over its registers -- no memory traffic inside the loop.  nms_kernel scans the alive bytes, the
// scores and the boxes in global memory twice per pick: 1.4 ms per launch on MTCNN's dense 26 k-cell P-Net grid, a
// quarter of that workload's GPU time.  Same picks in the same order (highest score, ties by lower index; IoU > threshold
// suppresses), so the same keep lists bit for bit.
// SLDS: the scores live in LDS instead (4 IPT NT bytes): 26 boxes + 26 scores per thread do not fit the 128 registers a
// 1024-thread block leaves a lane
template <int NT, int IPT, bool SLDS>
__global__ __launch_bounds__(NT) void nms_reg_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int ntot,
                                                     int max_boxes, float score_thr, float iou_thr, int* __restrict__ keep_idx,
                                                     int* __restrict__ keep_n) {
  constexpr int NW = NT / 64;
  __shared__ float s_score[NW];
  __shared__ int s_idx[NW];
  __shared__ float s_box[4];
  __shared__ int s_pick;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* bx = boxes + (int64_t)n * ntot * 4;
  const float* sc = scores + (int64_t)n * ntot;
  int* out = keep_idx + (int64_t)n * max_boxes;
  extern __shared__ float s_lds[];                          // SLDS: [IPT][NT]
  float b[IPT][4], s_reg[SLDS ? 1 : IPT];
  auto S = [&](int j) -> float& { if constexpr (SLDS) return s_lds[j * NT + tid]; else return s_reg[j]; };
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int i = tid + NT * j;
    S(j) = -__builtin_inff();
    b[j][0] = b[j][1] = b[j][2] = b[j][3] = 0.f;
    if (i < ntot) {
      const float v = sc[i];
      if (v >= score_thr) S(j) = v;                        // NaN scores never take part, as in nms_kernel
      const f32x4 q = *reinterpret_cast<const f32x4*>(bx + (int64_t)i * 4);
      b[j][0] = q[0];
      b[j][1] = q[1];
      b[j][2] = q[2];
      b[j][3] = q[3];
    }
  }
  for (int i = tid; i < max_boxes; i += NT) out[i] = -1;
  int kept = 0;
  while (kept < max_boxes) {
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < IPT; ++j)                            // ascending index: the first maximum is the lowest index
      if (S(j) > best) {
        best = S(j);
        bi = tid + NT * j;
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (os > best || (os == best && oi < bi)) {
        best = os;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_score[wave] = best;
      s_idx[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      float bb = s_score[0];
      int k = s_idx[0];
      for (int w = 1; w < NW; ++w)
        if (s_score[w] > bb || (s_score[w] == bb && s_idx[w] < k)) {
          bb = s_score[w];
          k = s_idx[w];
        }
      s_pick = k;
      if (k != 0x7fffffff) {
        out[kept] = k;
        for (int e = 0; e < 4; ++e) s_box[e] = bx[(int64_t)k * 4 + e];
      }
    }
    __syncthreads();
    const int pick = s_pick;
    if (pick == 0x7fffffff) break;                         // nothing left
    ++kept;
    const float pb[4] = {s_box[0], s_box[1], s_box[2], s_box[3]};
#pragma unroll
    for (int j = 0; j < IPT; ++j)
      if (S(j) > -__builtin_inff() && (tid + NT * j == pick || box_iou(b[j], pb) > iou_thr)) S(j) = -__builtin_inff();
    __syncthreads();                                       // s_pick / s_box are rewritten by the next round
  }
  if (tid == 0) keep_n[n] = kept;
}

}  // namespace dif

using namespace dif;

extern "C" {

int dif_yolo_decode(const float* const* feats_dev, const int32_t* grid_hw_host, const float* anchors_host,
                    int n_layers, int n_images, int n_classes, int input_h, int input_w,
                    const float* image_shape_dev, float* boxes_dev, float* scores_dev, void* stream) {
  if (n_layers < 1 || n_layers > 3) return set_error("dif_yolo_decode: 1..3 output layers");
  if (n_images < 0 || n_classes < 1) return set_error("dif_yolo_decode: bad sizes");
  if (n_images == 0) return 0;
  if (!feats_dev || !grid_hw_host || !anchors_host || !image_shape_dev || !boxes_dev || !scores_dev)
    return set_error("dif_yolo_decode: null pointer");
  YoloArgs a;
  a.nlayers = n_layers;
  a.N = n_images;
  a.C = n_classes;
  a.in_h = (float)input_h;
  a.in_w = (float)input_w;
  a.image_shape = image_shape_dev;
  a.boxes = boxes_dev;
  a.scores = scores_dev;
  int off = 0;
  for (int l = 0; l < n_layers; ++l) {
    a.L[l].feats = feats_dev[l];
    a.L[l].gh = grid_hw_host[2 * l];
    a.L[l].gw = grid_hw_host[2 * l + 1];
    a.L[l].offset = off;
    for (int k = 0; k < 3; ++k) {
      a.L[l].aw[k] = anchors_host[(l * 3 + k) * 2];
      a.L[l].ah[k] = anchors_host[(l * 3 + k) * 2 + 1];
    }
    off += a.L[l].gh * a.L[l].gw * 3;
  }
  a.ntot = off;
  const int64_t total = (int64_t)n_images * off;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  DIF_HIP(hipGetLastError());
  return 0;
}

int dif_nms(const float* boxes_dev, const float* scores_dev, int n_images, int n_boxes, int n_classes, int max_boxes,
            float score_threshold, float iou_threshold, uint8_t* alive_ws_dev, int32_t* keep_idx_dev,
            int32_t* keep_count_dev, void* stream) {
  if (n_images < 0 || n_boxes < 0 || n_classes < 1 || max_boxes < 1) return set_error("dif_nms: bad sizes");
  if (n_images == 0) return 0;
  if (!boxes_dev || !scores_dev || !alive_ws_dev || !keep_idx_dev || !keep_count_dev)
    return set_error("dif_nms: null pointer");
  // one class and few enough boxes for the register form (MTCNN's grids and slot lists, YOLOv3-face's 10 647 boxes): the
  // workspace is not used
  if (n_classes == 1 && n_boxes <= 1024 * 26) {
    if (n_boxes <= 1024 * 11)
      hipLaunchKernelGGL((nms_reg_kernel<1024, 11, false>), dim3(n_images), dim3(1024), 0, (hipStream_t)stream, boxes_dev, scores_dev,
                         n_boxes, max_boxes, score_threshold, iou_threshold, keep_idx_dev, keep_count_dev);
    else {
      static bool attr_set[64];
      int dev = 0;
      DIF_HIP(hipGetDevice(&dev));
      if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        DIF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_reg_kernel<1024, 26, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 26 * 1024 * 4));
        attr_set[dev] = true;
      }
      hipLaunchKernelGGL((nms_reg_kernel<1024, 26, true>), dim3(n_images), dim3(1024), 26 * 1024 * 4, (hipStream_t)stream, boxes_dev,
                         scores_dev, n_boxes, max_boxes, score_threshold, iou_threshold, keep_idx_dev, keep_count_dev);
    }
    DIF_HIP(hipGetLastError());
    return 0;
  }
  if (n_boxes > 12288)
    hipLaunchKernelGGL(nms_kernel<1024>, dim3(n_images, n_classes), dim3(1024), 0, (hipStream_t)stream, boxes_dev, scores_dev,
                       n_boxes, n_classes, max_boxes, score_threshold, iou_threshold, alive_ws_dev, keep_idx_dev, keep_count_dev);
  else
    hipLaunchKernelGGL(nms_kernel<256>, dim3(n_images, n_classes), dim3(256), 0, (hipStream_t)stream, boxes_dev, scores_dev,
                       n_boxes, n_classes, max_boxes, score_threshold, iou_threshold, alive_ws_dev, keep_idx_dev, keep_count_dev);
  DIF_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
