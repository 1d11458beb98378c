// Memory-bound layer kernels around the convolutions: input conversion, max pooling,
// the GDC head's full-extent depthwise convolution, row-wise L2 normalisation.
// All NHWC; channel runs are read/written as float4 (16 B per lane).
#include "ops.hpp"
#include "dif_internal.hpp"
#include "../../include/dif.h"

namespace dif {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- input conversion
// predictions.py:94,154: `cv2.resize(...) * rescale`; predictions.py:95: keras vgg16
// preprocess_input (BGR swap + per-channel mean subtraction) for the siamese path.
__global__ __launch_bounds__(256) void input_convert_kernel(const InputArgs a) {
  const int64_t npix = (int64_t)a.N * a.H * a.W;
  const int64_t HW = (int64_t)a.H * a.W;
  const bool bgr = a.bgr & DIF_INPUT_BGR, hflip = a.bgr & DIF_INPUT_HFLIP;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < npix; q += (int64_t)gridDim.x * 256) {
    float v[3];
    int64_t p = q;                                   // source pixel
    if (hflip) {
      const int w = (int)(q % a.W);
      p = q - w + (a.W - 1 - w);
    }
    if (a.layout == DIF_LAYOUT_NHWC) {
      if (a.dtype == DIF_DTYPE_U8) {
        const uint8_t* s = static_cast<const uint8_t*>(a.x) + p * 3;
        v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
      } else {
        const float* s = static_cast<const float*>(a.x) + p * 3;
        v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
      }
    } else {
      const int64_t n = p / HW, r = p - n * HW;
      if (a.dtype == DIF_DTYPE_U8) {
        const uint8_t* s = static_cast<const uint8_t*>(a.x) + n * 3 * HW + r;
        v[0] = s[0]; v[1] = s[HW]; v[2] = s[2 * HW];
      } else {
        const float* s = static_cast<const float*>(a.x) + n * 3 * HW + r;
        v[0] = s[0]; v[1] = s[HW]; v[2] = s[2 * HW];
      }
    }
    f32x4 o;
    o[0] = (bgr ? v[2] : v[0]) * a.scale + a.bias[0];
    o[1] = v[1] * a.scale + a.bias[1];
    o[2] = (bgr ? v[0] : v[2]) * a.scale + a.bias[2];
    o[3] = 0.f;
    *reinterpret_cast<f32x4*>(a.y + q * 4) = o;
  }
}

int input_convert_run(const InputArgs& a, hipStream_t st) {
  const int64_t npix = (int64_t)a.N * a.H * a.W;
  if (npix == 0) return 0;
  int64_t blocks = (npix + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(input_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  DIF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- depthwise KxK convolution
// MobileNetV2's DepthwiseConv2D(3, stride 1 'same' | stride 2 after ZeroPadding2D(correct_pad)) + BN +
// ReLU6, one thread per (pixel, 4 channels): 9 float4 loads + 9 weight float4 (L1/L2-resident) per
// output float4 -- memory-bound, nothing for the matrix pipe here.
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     float* __restrict__ y, int N, int H, int W, int C, int K,
                                                     int stride, int pad_t, int pad_l, int Ho, int Wo, int act) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int64_t n = p / Ho;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < K; ++kh) {
      const int hi = ho * stride - pad_t + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int wi = wo * stride - pad_l + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((n * H + hi) * W + wi) * C + c4 * 4);
        const f32x4 k = *reinterpret_cast<const f32x4*>(w + (int64_t)(kh * K + kw) * C + c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[j], k[j], acc[j]);
      }
    }
    const f32x4 sc = scale ? *reinterpret_cast<const f32x4*>(scale + c4 * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sh = shift ? *reinterpret_cast<const f32x4*>(shift + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = fmaf(acc[j], sc[j], sh[j]);
      if (act == ACT_RELU) t = fmaxf(t, 0.f);
      else if (act == ACT_RELU6) t = fminf(fmaxf(t, 0.f), 6.f);
      o[j] = t;
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

int dwconv_run(const float* x, const float* w, const float* scale, const float* shift, float* y, int N, int H, int W,
               int C, int K, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, hipStream_t st) {
  if (C % 4 != 0) return set_error("dwconv: C must be a multiple of 4 (got %d)", C);
  const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
  if (total == 0) return 0;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(dwconv_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, w, scale, shift, y, N, H, W, C, K, stride,
                     pad_t, pad_l, Ho, Wo, act);
  DIF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- pooling (max / L2 / average)
__global__ __launch_bounds__(256) void maxpool_kernel(const PoolArgs a) {
  const int C4 = a.C / 4;
  const int64_t total = (int64_t)a.N * a.Ho * a.Wo * C4;
  const float inv_kk = 1.f / (float)(a.k * a.k);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int wo = (int)(p % a.Wo);
    p /= a.Wo;
    const int ho = (int)(p % a.Ho);
    const int64_t n = p / a.Ho;
    const float lo = -__builtin_inff();
    f32x4 m = (a.mode == POOL_MAX) ? f32x4{lo, lo, lo, lo} : f32x4{0.f, 0.f, 0.f, 0.f};
    bool padded = false;
    for (int kh = 0; kh < a.k; ++kh) {
      const int hi = ho * a.stride - a.pad_t + kh;
      for (int kw = 0; kw < a.k; ++kw) {
        const int wi = wo * a.stride - a.pad_l + kw;
        if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + ((n * a.H + hi) * a.W + wi) * a.C + c4 * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (a.mode == POOL_MAX)
              m[j] = fmaxf(m[j], v[j]);
            else if (a.mode == POOL_L2)
              m[j] = fmaf(v[j], v[j], m[j]);
            else
              m[j] += v[j];
          }
        } else {
          padded = true;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.mode == POOL_MAX) {
        if (padded && a.zero_pad) m[j] = fmaxf(m[j], 0.f);
      } else if (a.mode == POOL_L2) {
        // x**2 -> AveragePooling2D -> * 9 -> sqrt   (networks/inceptionv3.py:160-163)
        m[j] = sqrtf((m[j] * inv_kk) * (float)(a.k * a.k));
      } else {
        m[j] *= inv_kk;
      }
    }
    const int64_t o = (((n * a.y_H + ho + a.y_oy) * a.y_W + wo + a.y_ox)) * a.y_ld + a.y_coff + c4 * 4;
    *reinterpret_cast<f32x4*>(a.y + o) = m;
    if (a.y2) {
      const int64_t o2 = ((n * a.Ho + ho) * a.Wo + wo) * a.C + c4 * 4;
      const f32x4 s = a.scale2 ? *reinterpret_cast<const f32x4*>(a.scale2 + c4 * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 t = a.shift2 ? *reinterpret_cast<const f32x4*>(a.shift2 + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[j] = fmaf(m[j], s[j], t[j]);
        if (a.act2 == ACT_RELU) q[j] = fmaxf(q[j], 0.f);
      }
      *reinterpret_cast<f32x4*>(a.y2 + o2) = q;
    }
  }
}

int maxpool_run(const PoolArgs& a, hipStream_t st) {
  if (a.C % 4 != 0) return set_error("pool: C must be a multiple of 4 (got %d)", a.C);
  if (a.y_ld % 4 != 0 || a.y_coff % 4 != 0) return set_error("pool: output view must be 16-byte aligned");
  const int64_t total = (int64_t)a.N * a.Ho * a.Wo * (a.C / 4);
  if (total == 0) return 0;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  DIF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- upsample x2 / copy into a concat slice
__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                        int H, int W, int C, int y_ld, int y_coff) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)N * (2 * H) * (2 * W) * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int wo = (int)(p % (2 * W));
    p /= 2 * W;
    const int ho = (int)(p % (2 * H));
    const int64_t n = p / (2 * H);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((n * H + ho / 2) * W + wo / 2) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(y + ((n * 2 * H + ho) * (2 * W) + wo) * (int64_t)y_ld + y_coff + c4 * 4) = v;
  }
}

int upsample2_run(const float* x, float* y, int N, int H, int W, int C, int y_ld, int y_coff, hipStream_t st) {
  if (C % 4 != 0 || y_ld % 4 != 0 || y_coff % 4 != 0) return set_error("upsample: channels must be 16-byte aligned");
  const int64_t total = (int64_t)N * 4 * H * W * (C / 4);
  if (total == 0) return 0;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(upsample2_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, N, H, W, C, y_ld, y_coff);
  DIF_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void copy_to_view_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int64_t npix, int C, int y_ld, int y_coff) {
  const int C4 = C / 4;
  const int64_t total = npix * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t p = i / C4;
    *reinterpret_cast<f32x4*>(y + p * y_ld + y_coff + c4 * 4) = *reinterpret_cast<const f32x4*>(x + p * C + c4 * 4);
  }
}

int copy_to_view_run(const float* x, float* y, int64_t npix, int C, int y_ld, int y_coff, hipStream_t st) {
  if (C % 4 != 0 || y_ld % 4 != 0 || y_coff % 4 != 0) return set_error("copy: channels must be 16-byte aligned");
  const int64_t total = npix * (C / 4);
  if (total == 0) return 0;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(copy_to_view_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, npix, C, y_ld, y_coff);
  DIF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- local response normalisation
// tf.nn.lrn(x, alpha=1e-4, beta=0.75) (networks/inceptionv3.py:95): depth_radius 5, bias 1.
__global__ __launch_bounds__(256) void lrn_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                  int64_t npix, int C, int radius, float bias, float alpha,
                                                  float beta) {
  const int64_t total = npix * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t base = i - c;
    const int lo = c - radius < 0 ? 0 : c - radius;
    const int hi = c + radius >= C ? C - 1 : c + radius;
    float s = 0.f;
    for (int j = lo; j <= hi; ++j) {
      const float v = x[base + j];
      s = fmaf(v, v, s);
    }
    y[i] = x[i] / powf(bias + alpha * s, beta);
  }
}

int lrn_run(const float* x, float* y, int64_t npix, int C, int radius, float bias, float alpha, float beta,
            hipStream_t st) {
  const int64_t total = npix * C;
  if (total == 0) return 0;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(lrn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, npix, C, radius, bias, alpha, beta);
  DIF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- GDC depthwise (kernel = whole map) + BN
// networks/triplet.py:129-130: DepthwiseConv2D(int(nn.shape[1]), depth_multiplier=1, use_bias=False) -> BN
__global__ __launch_bounds__(256) void dwfull_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ y,
                                                     int N, int HW, int C) {
  const int64_t total = (int64_t)N * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t n = i / C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s = fmaf(x[(n * HW + p) * C + c], w[(int64_t)p * C + c], s);
    y[i] = fmaf(s, scale ? scale[c] : 1.f, shift ? shift[c] : 0.f);
  }
}

int dwfull_run(const float* x, const float* w, const float* scale, const float* shift, float* y, int N, int HW,
               int C, hipStream_t st) {
  const int64_t total = (int64_t)N * C;
  if (total == 0) return 0;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dwfull_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, w, scale, shift, y, N, HW, C);
  DIF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- 3-channel 3x3 stem, direct
// The first layer of YOLOv3-face (Conv3x3(32) at 416x416; the same shape as IResNet's conv1, which measured no
// faster this way and stays on the MFMA path): 27 inputs per output value.  As an implicit GEMM its K (36, padded
// to 64) wastes 44 % of the MFMA work and the layer is bound by its output rows anyway.  Direct form: 16 pixels x 4 channel
// groups per wave, the 27 x Cout weights in LDS (broadcast reads), 27 x Cout/4 fma per thread, the same
// epilogue as the convolution kernels: y = act(acc * scale + shift), optional y2 = act2(y * scale2 + shift2).
template <int COUT>
__global__ __launch_bounds__(256, 4) void stem3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ alpha, const float* __restrict__ scale2,
                                                      const float* __restrict__ shift2, const float* __restrict__ alpha2,
                                                      float* __restrict__ y, float* __restrict__ y2, int N, int H, int W,
                                                      int act, int act2) {
  constexpr int CG = COUT / 4;                    // output channels per thread
  __shared__ __attribute__((aligned(16))) float ws[27 * COUT];   // Keras HWIO as it is: [kh][kw][ci < 3][co]
  const int tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float ep[6][COUT];     // epilogue vectors (defaults where absent)
  for (int i = tid; i < 27 * COUT; i += 256) ws[i] = w[i];
  for (int c = tid; c < COUT; c += 256) {
    ep[0][c] = scale ? scale[c] : 1.f;
    ep[1][c] = shift ? shift[c] : 0.f;
    ep[2][c] = alpha ? alpha[c] : 0.f;
    ep[3][c] = scale2 ? scale2[c] : 1.f;
    ep[4][c] = shift2 ? shift2[c] : 0.f;
    ep[5][c] = alpha2 ? alpha2[c] : 0.f;
  }
  __syncthreads();
  const int g = tid & 3;
  const int64_t total = (int64_t)N * H * W;
  const float* wg = ws + g * CG;
  const int c0 = g * CG;
  // the block keeps its weights and walks 64-pixel groups (grid-stride): the 6.9 KB weight fetch is paid once
  for (int64_t pix = (int64_t)blockIdx.x * 64 + (tid >> 2); pix < total; pix += (int64_t)gridDim.x * 64) {
    // the weights are re-read from LDS for every pixel group: left alone, the compiler hoists all 27 x Cout/4 of a
    // thread's weights out of this loop into 256 registers (one wave per SIMD: measured 9x slower)
    asm volatile("" : : "v"(ws), "v"(ep) : "memory");   // (the arrays' addresses escape into the asm: it may have written them)
    const int hw = (int)(pix % ((int64_t)H * W));
    const int64_t n = pix / ((int64_t)H * W);
    const int hq = hw / W, wq = hw - hq * W;
    // a ROLLED loop over the nine taps (pixel load, 12 weight reads from LDS, 3 x CG fma): unrolled, the scheduler
    // hoists the weight reads of all taps (432 registers' worth) and spills them -- measured 3x slower than the MFMA path
    float acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = 0.f;
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
      // the three pixels of one kernel row are requested together (their latencies overlap); the tap loop stays rolled
      const int hi = hq + kh - 1;
      const bool hok = (unsigned)hi < (unsigned)H;
      f32x4 vr[3];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = wq + kw - 1;
        const bool ok = hok && (unsigned)wi < (unsigned)W;
        const f32x4 t = *reinterpret_cast<const f32x4*>(x + ((n * H + (ok ? hi : hq)) * W + (ok ? wi : wq)) * 4);
        vr[kw] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll 1
      for (int kw = 0; kw < 3; ++kw) {
        const f32x4 v = kw == 0 ? vr[0] : (kw == 1 ? vr[1] : vr[2]);
        const float* wt = wg + (kh * 3 + kw) * 3 * COUT;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
          for (int c4 = 0; c4 < CG; c4 += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + ci * COUT + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c4 + j] = fmaf(v[ci], wv[j], acc[c4 + j]);
          }
        }
      }
    }
    float* yo = y ? y + pix * COUT + c0 : nullptr;
    float* y2o = y2 ? y2 + pix * COUT + c0 : nullptr;
#pragma unroll
    for (int c4 = 0; c4 < CG; c4 += 4) {
      f32x4 o, o2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + c4 + j;
        float t1 = fmaf(acc[c4 + j], ep[0][c], ep[1][c]);
        if (act == ACT_RELU) t1 = fmaxf(t1, 0.f);
        else if (act == ACT_PRELU) t1 = t1 >= 0.f ? t1 : t1 * ep[2][c];
        else if (act == ACT_RELU6) t1 = fminf(fmaxf(t1, 0.f), 6.f);
        o[j] = t1;
        float t2 = fmaf(t1, ep[3][c], ep[4][c]);
        if (act2 == ACT_RELU) t2 = fmaxf(t2, 0.f);
        else if (act2 == ACT_PRELU) t2 = t2 >= 0.f ? t2 : t2 * ep[5][c];
        else if (act2 == ACT_RELU6) t2 = fminf(fmaxf(t2, 0.f), 6.f);
        o2[j] = t2;
      }
      if (yo) *reinterpret_cast<f32x4*>(yo + c4) = o;
      if (y2o) *reinterpret_cast<f32x4*>(y2o + c4) = o2;
    }
  }
}

int stem3x3_run(const float* x, const float* w_hwio, const float* scale, const float* shift, const float* alpha,
                const float* scale2, const float* shift2, const float* alpha2, float* y, float* y2, int N, int H, int W,
                int Cout, int act, int act2, hipStream_t st) {
  const int64_t total = (int64_t)N * H * W;
  if (total == 0) return 0;
  int64_t groups = (total + 63) / 64;
  const unsigned blocks = (unsigned)(groups < 4096 ? groups : 4096);     // 16 resident blocks per CU, grid-stride beyond
  if (Cout == 32)
    hipLaunchKernelGGL(stem3x3_kernel<32>, dim3(blocks), dim3(256), 0, st, x, w_hwio, scale, shift, alpha, scale2, shift2,
                       alpha2, y, y2, N, H, W, act, act2);
  else
    return set_error("stem3x3: Cout must be 32 (got %d)", Cout);
  DIF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- GDC tail, fused
// networks/triplet.py:129-138 after the head's first convolution: DepthwiseConv2D(kernel = whole map) -> BN ->
// Conv2D(emd, 1) -> Dropout (identity) -> Flatten -> Dense(emd) -> l2_normalize, two images per block.  About
// 0.5 MMAC per image against 2 MB of weights that stay in L2: bound by launch and latency, so what matters is
// that it is ONE launch (it was four) and that the weight reads are coalesced along the output axis.
// KSPLIT = K slices of the two matrix-vector products = threads / 128.  2 (256 threads) for whole batches: two images per
// block, many blocks.  8 (1024 threads) for one or two images (round 5): the block is alone on the chip and streams the two
// 1 MB weight matrices through ONE CU -- four times the loads in flight (34 -> ~15 us at batch 1); same sums in another
// (fixed) order.
template <int KSPLIT>
__global__ __launch_bounds__(128 * KSPLIT) void gdc_tail_kernel(const float* __restrict__ x, const float* __restrict__ wdw,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ wpw, const float* __restrict__ wd,
                                                       float* __restrict__ y, int N, int HW, int E, float eps) {
  constexpr int C = 512;
  __shared__ float a[2][C];            // depthwise + BN output of the block's two images
  __shared__ float b[2][1024];         // 1x1 convolution output
  __shared__ float part[KSPLIT][2][1024];   // [K slice][image][j]: partial sums of the two matrix-vector products
  __shared__ float red[2][2 * KSPLIT];
  constexpr int NT = 128 * KSPLIT;
  const int tid = threadIdx.x;
  const int64_t n0 = (int64_t)blockIdx.x * 2;
  const bool two = n0 + 1 < N;
  const float* x0 = x + n0 * HW * C;
  const float* x1 = x + (two ? n0 + 1 : n0) * HW * C;
  for (int c = tid; c < C; c += NT) {
    float s0 = 0.f, s1 = 0.f;
    for (int p = 0; p < HW; ++p) {
      const float w = wdw[p * C + c];
      s0 = fmaf(x0[p * C + c], w, s0);
      s1 = fmaf(x1[p * C + c], w, s1);
    }
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    a[0][c] = fmaf(s0, sc, sh);
    a[1][c] = fmaf(s1, sc, sh);
  }
  __syncthreads();
  // y[j] = sum_k v[k] * W[k][j] for both images: a thread owns four adjacent outputs (one 16-byte weight load per k)
  // and one half of the k range; sixteen loads in flight per thread keep the L2 round trip covered
  const int half = tid >> 7, jq = tid & 127;                // `half`: this thread's K slice
  auto gemv = [&](const float* W, int K, const float* v0, const float* v1) {
    const int k0 = half * (K / KSPLIT), k1 = k0 + K / KSPLIT;
    for (int j = 4 * jq; j < E; j += 512) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
      for (int k = k0; k < k1; ++k) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W + (int64_t)k * E + j);
        const float u0 = v0[k], u1 = v1[k];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          s0[q] = fmaf(u0, w[q], s0[q]);
          s1[q] = fmaf(u1, w[q], s1[q]);
        }
      }
      *reinterpret_cast<f32x4*>(&part[half][0][j]) = s0;
      *reinterpret_cast<f32x4*>(&part[half][1][j]) = s1;
    }
  };
  gemv(wpw, C, a[0], a[1]);
  __syncthreads();
  for (int j = tid; j < E; j += NT) {
    float s0 = part[0][0][j], s1 = part[0][1][j];
#pragma unroll
    for (int k = 1; k < KSPLIT; ++k) {
      s0 += part[k][0][j];
      s1 += part[k][1][j];
    }
    b[0][j] = s0;
    b[1][j] = s1;
  }
  __syncthreads();
  gemv(wd, E, b[0], b[1]);
  __syncthreads();
  constexpr int QN = 1024 / NT;                           // outputs per thread (E <= 1024)
  float o0[QN], o1[QN], ss0 = 0.f, ss1 = 0.f;
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int j = tid + NT * q;
    o0[q] = o1[q] = 0.f;
    if (j < E) {
      o0[q] = part[0][0][j];
      o1[q] = part[0][1][j];
#pragma unroll
      for (int k = 1; k < KSPLIT; ++k) {
        o0[q] += part[k][0][j];
        o1[q] += part[k][1][j];
      }
    }
    ss0 = fmaf(o0[q], o0[q], ss0);
    ss1 = fmaf(o1[q], o1[q], ss1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss0 += __shfl_xor(ss0, o);
    ss1 += __shfl_xor(ss1, o);
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = ss0;
    red[1][tid >> 6] = ss1;
  }
  __syncthreads();
  float t0 = red[0][0], t1 = red[1][0];
#pragma unroll
  for (int k = 1; k < 2 * KSPLIT; ++k) {
    t0 += red[0][k];
    t1 += red[1][k];
  }
  const float inv0 = 1.f / sqrtf(fmaxf(t0, eps));
  const float inv1 = 1.f / sqrtf(fmaxf(t1, eps));
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int j = tid + NT * q;
    if (j < E) {
      y[n0 * E + j] = o0[q] * inv0;
      if (two) y[(n0 + 1) * E + j] = o1[q] * inv1;
    }
  }
}

// One or two images (round 5, late): the single block above streams the two 1 MB weight matrices through ONE CU, ~9 us each
// at what a CU draws from L2.  Here the two matrix-vector products are spread over E / 32 blocks each -- a block owns 32
// adjacent outputs (128-byte weight rows, 64 KB of a matrix) --, in two launches: the second needs ALL of the first's outputs,
// and the kernel boundary is the cheap way to hand a vector from every block to every block (1.5 us; a spinning grid barrier
// costs more and can hang).  The normalisation needs all of the second product's outputs: its blocks publish their 32 values
// write-through (sc1), drain, draw a ticket; the block that draws the last one normalises and writes the rows (nobody waits).
//   gdc_tail_a_kernel  depthwise + BN (every block for itself: 3 x 32 KB from L2) -> its 32 columns of the 1x1 convolution
//   gdc_tail_b_kernel  its 32 columns of the dense layer -> ws; last block: l2-normalise
// ws: [2][1024] 1x1 outputs, [2][1024] dense outputs, the ticket (zero between launches: its last taker clears it).
// Summation order: 32 K slices of K / 32 consecutive terms each, added in slice order -- fixed, not the single block's.
constexpr int GT_JB = 32;              // outputs per block
__device__ __forceinline__ void gt_gemv_slice(const float* __restrict__ W, int K, int E, int j0, const float* v0, const float* v1,
                                              float (*part)[2][GT_JB], float* out0, float* out1) {
  // thread = (K slice ks = tid >> 3, four adjacent outputs jq = tid & 7); 256 threads
  const int tid = threadIdx.x, ks = tid >> 3, jq = tid & 7;
  const int kn = K / 32, k0 = ks * kn;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
  for (int k = k0; k < k0 + kn; ++k) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (int64_t)k * E + j0 + 4 * jq);
    const float u0 = v0[k], u1 = v1[k];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s0[q] = fmaf(u0, w[q], s0[q]);
      s1[q] = fmaf(u1, w[q], s1[q]);
    }
  }
  *reinterpret_cast<f32x4*>(&part[ks][0][4 * jq]) = s0;
  *reinterpret_cast<f32x4*>(&part[ks][1][4 * jq]) = s1;
  __syncthreads();
  if (tid < 2 * GT_JB) {
    const int img = tid >> 5, j = tid & 31;
    float t = part[0][img][j];
#pragma unroll
    for (int q = 1; q < 32; ++q) t += part[q][img][j];
    (img ? out1 : out0)[j] = t;
  }
}

__global__ __launch_bounds__(256) void gdc_tail_a_kernel(const float* __restrict__ x, const float* __restrict__ wdw,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ wpw, float* __restrict__ ws, int N, int HW, int E) {
  constexpr int C = 512;
  __shared__ float a[2][C];
  __shared__ float part[32][2][GT_JB];
  const int tid = threadIdx.x;
  const float* x0 = x;
  const float* x1 = x + (N > 1 ? (int64_t)HW * C : 0);
  for (int c = tid; c < C; c += 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int p = 0; p < HW; ++p) {
      const float w = wdw[p * C + c];
      s0 = fmaf(x0[p * C + c], w, s0);
      s1 = fmaf(x1[p * C + c], w, s1);
    }
    const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    a[0][c] = fmaf(s0, sc, sh);
    a[1][c] = fmaf(s1, sc, sh);
  }
  __syncthreads();
  const int j0 = blockIdx.x * GT_JB;
  gt_gemv_slice(wpw, C, E, j0, a[0], a[1], part, ws + j0, ws + 1024 + j0);
}

__global__ __launch_bounds__(256) void gdc_tail_b_kernel(const float* __restrict__ wd, float* __restrict__ ws,
                                                         float* __restrict__ y, int N, int E, float eps) {
  __shared__ float b[2][1024];
  __shared__ float part[32][2][GT_JB];
  __shared__ float o[2][GT_JB];
  __shared__ float red[2][4];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  for (int j = tid; j < E; j += 256) {
    b[0][j] = ws[j];
    b[1][j] = ws[1024 + j];
  }
  __syncthreads();
  const int j0 = blockIdx.x * GT_JB;
  gt_gemv_slice(wd, E, E, j0, b[0], b[1], part, o[0], o[1]);
  // publish: write-through stores, drained, then one ticket per block (MI355X_MICROARCH.md, visibility: the counter form)
  float* wo = ws + 2048;
  unsigned* ticket = reinterpret_cast<unsigned*>(ws + 4096);
  if (tid < 2 * GT_JB) {
    const int img = tid >> 5, j = tid & 31;
    __hip_atomic_store(wo + img * 1024 + j0 + j, o[img][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == gridDim.x - 1;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // the last block: every output of both images (sc1 loads: the other blocks' stores went through to memory)
  constexpr int QN = 4;                                      // E <= 1024
  float v0[QN], v1[QN], ss0 = 0.f, ss1 = 0.f;
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int j = tid + 256 * q;
    v0[q] = v1[q] = 0.f;
    if (j < E) {
      v0[q] = __hip_atomic_load(wo + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v1[q] = __hip_atomic_load(wo + 1024 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ss0 = fmaf(v0[q], v0[q], ss0);
    ss1 = fmaf(v1[q], v1[q], ss1);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    ss0 += __shfl_xor(ss0, d);
    ss1 += __shfl_xor(ss1, d);
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = ss0;
    red[1][tid >> 6] = ss1;
  }
  __syncthreads();
  const float t0 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
  const float t1 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  const float inv0 = 1.f / sqrtf(fmaxf(t0, eps)), inv1 = 1.f / sqrtf(fmaxf(t1, eps));
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int j = tid + 256 * q;
    if (j < E) {
      y[j] = v0[q] * inv0;
      if (N > 1) y[E + j] = v1[q] * inv1;
    }
  }
}

int gdc_tail_run(const float* x, const float* w_dw, const float* scale, const float* shift, const float* w_pw,
                 const float* w_dense, float* y, int N, int HW, int E, float eps, float* ws, hipStream_t st) {
  if (N == 0) return 0;
  if (E > 1024 || E % 8 != 0) return set_error("gdc_tail: emd %d must be a multiple of 8, at most 1024", E);
  if (N <= 2 && ws && E % GT_JB == 0) {
    hipLaunchKernelGGL(gdc_tail_a_kernel, dim3((unsigned)(E / GT_JB)), dim3(256), 0, st, x, w_dw, scale, shift, w_pw, ws, N, HW, E);
    DIF_HIP(hipGetLastError());
    hipLaunchKernelGGL(gdc_tail_b_kernel, dim3((unsigned)(E / GT_JB)), dim3(256), 0, st, w_dense, ws, y, N, E, eps);
    DIF_HIP(hipGetLastError());
    return 0;
  }
  if (N <= 2)
    hipLaunchKernelGGL(gdc_tail_kernel<8>, dim3(1), dim3(1024), 0, st, x, w_dw, scale, shift, w_pw, w_dense, y, N, HW, E, eps);
  else
    hipLaunchKernelGGL(gdc_tail_kernel<2>, dim3((unsigned)((N + 1) / 2)), dim3(256), 0, st, x, w_dw, scale, shift, w_pw, w_dense,
                       y, N, HW, E, eps);
  DIF_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- row-wise L2 normalisation
// tf.nn.l2_normalize(axis=1) (networks/triplet.py:138): x * rsqrt(max(sum(x^2), 1e-12)); one wave per row.
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                     int D, float eps) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= N) return;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float v = x[r * D + k];
    s = fmaf(v, v, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.f / sqrtf(fmaxf(s, eps));
  for (int k = lane; k < D; k += 64) y[r * D + k] = x[r * D + k] * inv;
}

int l2norm_run(const float* x, float* y, int N, int D, float eps, hipStream_t st) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, x, y, N, D, eps);
  DIF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dif
