// hipcc-flags: -ffp-contract=fast-honor-pragmas
// (build.py compiles with -ffp-contract=fast, which DISREGARDS `#pragma clang fp contract(off)`: area_pixel below restates cv2's
// float32 arithmetic operation by operation and needs the pragma honoured; the other kernels of this file keep contraction)
// Image resampling around the detector (BASELINE configs[4]: raw frames -> detect -> crop -> embed,
// all on the device):
//   letterbox_kernel   detector/yolov3.py:108-119 (letterbox_image): aspect-preserving resize with
//                      PIL's BICUBIC filter (Keys a = -0.5, support widened by the scale factor
//                      when shrinking, i.e. PIL's antialiasing) pasted on a (128,128,128) canvas.
//   crop_resize_kernel detector/run.py:63-87 (filter_bounding_box: margin, clamp, crop) followed by
//                      the resize predictions.py:93,154 applies: cv2.resize(..., interpolation=
//                      Image.BICUBIC) -- PIL's constant 3, which cv2 reads as INTER_AREA (area
//                      coverage resampling).
// uint8 in, uint8 out, one thread per output pixel; memory-bound and tiny next to the networks.
#include "../../include/dif.h"
#include "dif_internal.hpp"

namespace dif {

__device__ __forceinline__ float bicubic_w(float x) {
  const float a = -0.5f;
  x = fabsf(x);
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}

// PIL-style 1-D resampling weights for output coordinate `o`: taps [lo, hi), weights normalised.
__device__ __forceinline__ void pil_taps(int o, float scale, int in_size, int& lo, int& hi, float& center,
                                         float& ww) {
  const float filterscale = scale < 1.f ? 1.f : scale;
  const float support = 2.f * filterscale;
  center = (o + 0.5f) * scale;
  ww = 1.f / filterscale;
  lo = (int)(center - support + 0.5f);
  if (lo < 0) lo = 0;
  hi = (int)(center + support + 0.5f);
  if (hi > in_size) hi = in_size;
}

__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ frames, int N, int H, int W,
                                                        uint8_t* __restrict__ out, int S, int nw, int nh) {
  const int64_t total = (int64_t)N * S * S;
  const int ox0 = (S - nw) / 2, oy0 = (S - nh) / 2;
  const float sx = (float)W / nw, sy = (float)H / nh;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % S);
    const int y = (int)((i / S) % S);
    const int64_t n = i / ((int64_t)S * S);
    uint8_t* o = out + i * 3;
    const int rx = x - ox0, ry = y - oy0;
    if (rx < 0 || rx >= nw || ry < 0 || ry >= nh) {
      o[0] = o[1] = o[2] = 128;
      continue;
    }
    int xl, xh, yl, yh;
    float cx, cy, wx, wy;
    pil_taps(rx, sx, W, xl, xh, cx, wx);
    pil_taps(ry, sy, H, yl, yh, cy, wy);
    // PIL resamples in two passes (horizontal, then vertical), each with its own normalised
    // weights and an 8-bit intermediate; do the same per output pixel
    float wxs = 0.f, wys = 0.f;
    for (int xx = xl; xx < xh; ++xx) wxs += bicubic_w((xx - cx + 0.5f) * wx);
    for (int yy = yl; yy < yh; ++yy) wys += bicubic_w((yy - cy + 0.5f) * wy);
    const float ixs = wxs != 0.f ? 1.f / wxs : 0.f, iys = wys != 0.f ? 1.f / wys : 0.f;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int yy = yl; yy < yh; ++yy) {
      const float wv = bicubic_w((yy - cy + 0.5f) * wy) * iys;
      const uint8_t* row = frames + ((n * H + yy) * W) * 3;
      float r[3] = {0.f, 0.f, 0.f};
      for (int xx = xl; xx < xh; ++xx) {
        const float w = bicubic_w((xx - cx + 0.5f) * wx) * ixs;
        r[0] = fmaf(w, (float)row[xx * 3], r[0]);
        r[1] = fmaf(w, (float)row[xx * 3 + 1], r[1]);
        r[2] = fmaf(w, (float)row[xx * 3 + 2], r[2]);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = fmaf(wv, fminf(fmaxf(floorf(r[c] + 0.5f), 0.f), 255.f), acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)fminf(fmaxf(floorf(acc[c] + 0.5f), 0.f), 255.f);
  }
}

// One destination pixel of cv2.resize(src[ch x cw], (SW, SH), interpolation=INTER_AREA) for uint8 images, operation by
// operation as OpenCV's resize.cpp performs it (the test suite's checker restates the same paths and must be equalled bit for
// bit; PARITY UNPINNED against cv2 itself, which is not installed here):
//   both axes shrink or stay: 2 x 2 blocks (a + b + c + d + 2) >> 2; other integer ratios cvRound(float(sum) * (1.f / area));
//     fractional ratios the DecimateAlpha tables -- per axis (source index, float32 weight) pairs from double arithmetic --
//     with buf = sum_k S * alpha_k per source row and sum = sum_j beta_j * buf_j, every product and addition rounded to
//     float32 in table order (no fused multiply-add), cvRound at the end;
//   an axis enlarges: both axes take the linear path with area-mode coefficients in 11-bit fixed point.
// `src` points at the crop's first pixel, `pitch` = pixels per source row.
struct AreaTaps {                 // computeResizeAreaTab for one destination index: a head, a run of full cells, a tail
  int s_head, s_run0, s_run1, s_tail;       // -1: absent
  float a_head, a_run, a_tail;
};
__device__ __forceinline__ AreaTaps area_taps(int d, int ssize, double scale) {
#pragma clang fp contract(off)      // (HIP's __fmul_rn / __dmul_rn are plain products: without this the compiler fuses them)
  AreaTaps t;
  const double f1 = (double)d * scale;                      // (no fused multiply-add: cv2 rounds the product)
  const double f2 = f1 + scale;
  const double cell = fmin(scale, ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = s2 < ssize - 1 ? s2 : ssize - 1;
  s1 = s1 < s2 ? s1 : s2;
  t.s_head = (s1 - f1 > 1e-3) ? s1 - 1 : -1;
  t.a_head = (float)((s1 - f1) / cell);
  t.s_run0 = s1;
  t.s_run1 = s2;
  t.a_run = (float)(1.0 / cell);
  t.s_tail = (f2 - s2 > 1e-3) ? s2 : -1;
  t.a_tail = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
  return t;
}
__device__ __forceinline__ void linear_taps(int d, int ssize, int dsize, int& s0, int& s1, int& a0, int& a1) {
#pragma clang fp contract(off)
  const double inv = (double)dsize / ssize, scale = 1.0 / inv;
  const double dsc = (double)d * scale;
  int sx = (int)floor(dsc);
  const double back = (double)(sx + 1) * inv;
  float fx = (float)((double)(d + 1) - back);
  fx = fx <= 0.f ? 0.f : fx - floorf(fx);
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
  s0 = sx;
  s1 = sx + 1 < ssize ? sx + 1 : ssize - 1;
  const float c0 = 1.f - fx;
  a0 = (int)rintf(c0 * 2048.f);
  a1 = (int)rintf(fx * 2048.f);
}
__device__ __forceinline__ void area_pixel(const uint8_t* __restrict__ src, int pitch, int cw, int ch, int SW, int SH, int x, int y,
                                           uint8_t* __restrict__ o) {
#pragma clang fp contract(off)      // every product and sum below is rounded by itself, as in OpenCV's scalar code
  const double scale_x = 1.0 / ((double)SW / cw), scale_y = 1.0 / ((double)SH / ch);
  if (scale_x >= 1.0 && scale_y >= 1.0) {
    const int ix = (int)scale_x, iy = (int)scale_y;
    if (fabs(scale_x - ix) < 2.220446049250313e-16 && fabs(scale_y - iy) < 2.220446049250313e-16) {
      int sum[3] = {0, 0, 0};
      for (int yy = 0; yy < iy; ++yy)
        for (int xx = 0; xx < ix; ++xx) {
          const uint8_t* p = src + ((int64_t)(y * iy + yy) * pitch + (x * ix + xx)) * 3;
          sum[0] += p[0];
          sum[1] += p[1];
          sum[2] += p[2];
        }
      const float sc = 1.f / (float)(ix * iy);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        o[c] = (ix == 2 && iy == 2) ? (uint8_t)((sum[c] + 2) >> 2) : (uint8_t)fminf(fmaxf(rintf((float)sum[c] * sc), 0.f), 255.f);
      return;
    }
    const AreaTaps tx = area_taps(x, cw, scale_x), ty = area_taps(y, ch, scale_y);
    float sum[3] = {0.f, 0.f, 0.f};
    bool first = true;
    auto row = [&](int sy, float beta) {
#pragma clang fp contract(off)      // (the pragma of the enclosing function does not reach into a lambda's body)
      const uint8_t* r = src + (int64_t)sy * pitch * 3;
      float buf[3] = {0.f, 0.f, 0.f};
      auto tap = [&](int sx, float alpha) {
#pragma clang fp contract(off)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float prod = (float)r[sx * 3 + c] * alpha;      // plain operators under the pragma (HIP's __fmul_rn / __fadd_rn
          buf[c] = buf[c] + prod;                                // are header functions compiled with contraction ON)
        }
      };
      if (tx.s_head >= 0) tap(tx.s_head, tx.a_head);
      for (int sx = tx.s_run0; sx < tx.s_run1; ++sx) tap(sx, tx.a_run);
      if (tx.s_tail >= 0) tap(tx.s_tail, tx.a_tail);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float prod = beta * buf[c];
        sum[c] = first ? prod : sum[c] + prod;
      }
      first = false;
    };
    if (ty.s_head >= 0) row(ty.s_head, ty.a_head);
    for (int sy = ty.s_run0; sy < ty.s_run1; ++sy) row(sy, ty.a_run);
    if (ty.s_tail >= 0) row(ty.s_tail, ty.a_tail);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)fminf(fmaxf(rintf(sum[c]), 0.f), 255.f);
    return;
  }
  int x0, x1, a0, a1, y0, y1, b0, b1;
  linear_taps(x, cw, SW, x0, x1, a0, a1);
  linear_taps(y, ch, SH, y0, y1, b0, b1);
  const uint8_t* r0 = src + (int64_t)y0 * pitch * 3;
  const uint8_t* r1 = src + (int64_t)y1 * pitch * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1, h1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
    int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    o[c] = (uint8_t)v;
  }
}

// boxes: [N][4] = left, top, right, bottom in frame pixels (as detector/run.py:114 returns them)
// N crops; crop n comes from frame n / K (K boxes per frame: dif_crop_resize_multi; K = 1 otherwise); `valid`: per crop, a
// negative value = an empty slot -> a black crop
__global__ __launch_bounds__(256) void crop_resize_kernel(const uint8_t* __restrict__ frames, int N, int H, int W,
                                                          const float* __restrict__ boxes, float margin,
                                                          uint8_t* __restrict__ out, int SW, int SH, int K = 1,
                                                          const float* __restrict__ valid = nullptr) {
  const int64_t total = (int64_t)N * SW * SH;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % SW);
    const int y = (int)((i / SW) % SH);
    const int64_t n = i / ((int64_t)SW * SH);
    // filter_bounding_box (run.py:76-80): margin/2 on every side, clamped, truncated to int32;
    // boxes == nullptr: the whole image (dif_area_resize)
    const float whole[4] = {0.f, 0.f, (float)W, (float)H};
    const float* b = boxes ? boxes + n * 4 : whole;
    const float mg = boxes ? margin : 0.f;
    int l = (int)fmaxf(b[0] - mg / 2, 0.f), t = (int)fmaxf(b[1] - mg / 2, 0.f);
    int r = (int)fminf(b[2] + mg / 2, (float)W), bt = (int)fminf(b[3] + mg / 2, (float)H);
    uint8_t* o = out + i * 3;
    const int cw = r - l, ch = bt - t;
    const bool nodet = b[0] != b[0] || b[1] != b[1] || b[2] != b[2] || b[3] != b[3] || (valid && valid[n] < 0.f);   // NaN = no detection
    if (nodet || !(cw > 0 && ch > 0)) {
      o[0] = o[1] = o[2] = 0;
      continue;
    }
    const uint8_t* img = frames + (n / K) * (int64_t)H * W * 3;
    area_pixel(img + ((int64_t)t * W + l) * 3, W, cw, ch, SW, SH, x, y, o);
  }
}

}  // namespace dif

using namespace dif;

extern "C" {

int dif_letterbox(const uint8_t* frames_dev, int n, int h, int w, uint8_t* out_dev, int size, void* stream) {
  if (n < 0 || h <= 0 || w <= 0 || size <= 0) return set_error("dif_letterbox: bad sizes");
  if (n == 0) return 0;
  if (!frames_dev || !out_dev) return set_error("dif_letterbox: null pointer");
  const float scale = fminf((float)size / w, (float)size / h);     // yolov3.py:113-115
  const int nw = (int)(w * scale), nh = (int)(h * scale);
  if (nw < 1 || nh < 1) return set_error("dif_letterbox: image too thin");
  int64_t blocks = ((int64_t)n * size * size + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(letterbox_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames_dev, n, h, w,
                     out_dev, size, nw, nh);
  DIF_HIP(hipGetLastError());
  return 0;
}

int dif_crop_resize(const uint8_t* frames_dev, int n, int h, int w, const float* boxes_ltrb_dev, float margin,
                    uint8_t* out_dev, int size, void* stream) {
  if (n < 0 || h <= 0 || w <= 0 || size <= 0) return set_error("dif_crop_resize: bad sizes");
  if (n == 0) return 0;
  if (!frames_dev || !boxes_ltrb_dev || !out_dev) return set_error("dif_crop_resize: null pointer");
  int64_t blocks = ((int64_t)n * size * size + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames_dev, n, h, w,
                     boxes_ltrb_dev, margin, out_dev, size, size);
  DIF_HIP(hipGetLastError());
  return 0;
}

int dif_crop_resize_multi(const uint8_t* frames_dev, int n, int h, int w, const float* boxes_ltrb_dev, const float* valid_dev,
                          int k, float margin, uint8_t* out_dev, int size, void* stream) {
  if (n < 0 || h <= 0 || w <= 0 || size <= 0 || k < 1) return set_error("dif_crop_resize_multi: bad sizes");
  if (n == 0) return 0;
  if (!frames_dev || !boxes_ltrb_dev || !out_dev) return set_error("dif_crop_resize_multi: null pointer");
  int64_t blocks = ((int64_t)n * k * size * size + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames_dev, n * k, h, w,
                     boxes_ltrb_dev, margin, out_dev, size, size, k, valid_dev);
  DIF_HIP(hipGetLastError());
  return 0;
}

int dif_area_resize(const uint8_t* images_dev, int n, int h, int w, uint8_t* out_dev, int out_h, int out_w,
                    void* stream) {
  if (n < 0 || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0) return set_error("dif_area_resize: bad sizes");
  if (n == 0) return 0;
  if (!images_dev || !out_dev) return set_error("dif_area_resize: null pointer");
  int64_t blocks = ((int64_t)n * out_h * out_w + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, images_dev, n, h, w,
                     (const float*)nullptr, 0.f, out_dev, out_w, out_h);
  DIF_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
