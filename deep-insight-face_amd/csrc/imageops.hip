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

// boxes: [N][4] = left, top, right, bottom in frame pixels (as detector/run.py:114 returns them)
__global__ __launch_bounds__(256) void crop_resize_kernel(const uint8_t* __restrict__ frames, int N, int H, int W,
                                                          const float* __restrict__ boxes, float margin,
                                                          uint8_t* __restrict__ out, int SW, int SH) {
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
    const bool nodet = b[0] != b[0] || b[1] != b[1] || b[2] != b[2] || b[3] != b[3];   // NaN = no detection
    if (nodet || !(cw > 0 && ch > 0)) {
      o[0] = o[1] = o[2] = 0;
      continue;
    }
    const uint8_t* img = frames + n * (int64_t)H * W * 3;
    const float sx = (float)cw / SW, sy = (float)ch / SH;
    float acc[3] = {0.f, 0.f, 0.f};
    {
      // cv2 INTER_AREA: the output pixel's footprint in the crop, source pixels weighted by the
      // covered fraction (when enlarging this touches at most 2x2 pixels -- cv2's area-mode
      // linear coefficients are the same coverage fractions)
      const float x0 = x * sx, x1 = (x + 1) * sx, y0 = y * sy, y1 = (y + 1) * sy;
      float wsum = 0.f;
      for (int yy = (int)y0; yy < ch && yy < (int)ceilf(y1); ++yy) {
        const float wy = fminf(y1, yy + 1.f) - fmaxf(y0, (float)yy);
        const uint8_t* row = img + ((int64_t)(t + yy) * W + l) * 3;
        for (int xx = (int)x0; xx < cw && xx < (int)ceilf(x1); ++xx) {
          const float w = wy * (fminf(x1, xx + 1.f) - fmaxf(x0, (float)xx));
          acc[0] = fmaf(w, (float)row[xx * 3], acc[0]);
          acc[1] = fmaf(w, (float)row[xx * 3 + 1], acc[1]);
          acc[2] = fmaf(w, (float)row[xx * 3 + 2], acc[2]);
          wsum += w;
        }
      }
      const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
      acc[0] *= inv; acc[1] *= inv; acc[2] *= inv;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)fminf(fmaxf(rintf(acc[c]), 0.f), 255.f);
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
