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

// ---------------------------------------------------------------- GDC tail, fused
// networks/triplet.py:129-138 after the head's first convolution: DepthwiseConv2D(kernel = whole map) -> BN ->
// Conv2D(emd, 1) -> Dropout (identity) -> Flatten -> Dense(emd) -> l2_normalize, two images per block.  About
// 0.5 MMAC per image against 2 MB of weights that stay in L2: bound by launch and latency, so what matters is
// that it is ONE launch (it was four) and that the weight reads are coalesced along the output axis.
__global__ __launch_bounds__(256) void gdc_tail_kernel(const float* __restrict__ x, const float* __restrict__ wdw,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ wpw, const float* __restrict__ wd,
                                                       float* __restrict__ y, int N, int HW, int E, float eps) {
  constexpr int C = 512;
  __shared__ float a[2][C];            // depthwise + BN output of the block's two images
  __shared__ float b[2][1024];         // 1x1 convolution output
  __shared__ float part[2][2][1024];   // [K half][image][j]: partial sums of the two matrix-vector products
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  const int64_t n0 = (int64_t)blockIdx.x * 2;
  const bool two = n0 + 1 < N;
  const float* x0 = x + n0 * HW * C;
  const float* x1 = x + (two ? n0 + 1 : n0) * HW * C;
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
  // y[j] = sum_k v[k] * W[k][j] for both images: a thread owns four adjacent outputs (one 16-byte weight load per k)
  // and one half of the k range; sixteen loads in flight per thread keep the L2 round trip covered
  const int half = tid >> 7, jq = tid & 127;
  auto gemv = [&](const float* W, int K, const float* v0, const float* v1) {
    const int k0 = half * (K / 2), k1 = k0 + K / 2;
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
  for (int j = tid; j < E; j += 256) {
    b[0][j] = part[0][0][j] + part[1][0][j];
    b[1][j] = part[0][1][j] + part[1][1][j];
  }
  __syncthreads();
  gemv(wd, E, b[0], b[1]);
  __syncthreads();
  float o0[4], o1[4], ss0 = 0.f, ss1 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = tid + 256 * q;
    o0[q] = j < E ? part[0][0][j] + part[1][0][j] : 0.f;
    o1[q] = j < E ? part[0][1][j] + part[1][1][j] : 0.f;
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
  const float inv0 = 1.f / sqrtf(fmaxf(red[0][0] + red[0][1] + red[0][2] + red[0][3], eps));
  const float inv1 = 1.f / sqrtf(fmaxf(red[1][0] + red[1][1] + red[1][2] + red[1][3], eps));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = tid + 256 * q;
    if (j < E) {
      y[n0 * E + j] = o0[q] * inv0;
      if (two) y[(n0 + 1) * E + j] = o1[q] * inv1;
    }
  }
}

int gdc_tail_run(const float* x, const float* w_dw, const float* scale, const float* shift, const float* w_pw,
                 const float* w_dense, float* y, int N, int HW, int E, float eps, hipStream_t st) {
  if (N == 0) return 0;
  if (E > 1024 || E % 8 != 0) return set_error("gdc_tail: emd %d must be a multiple of 8, at most 1024", E);
  hipLaunchKernelGGL(gdc_tail_kernel, dim3((unsigned)((N + 1) / 2)), dim3(256), 0, st, x, w_dw, scale, shift, w_pw, w_dense,
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
