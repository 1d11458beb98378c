// First layers of the embedding networks: a KHxKW convolution of a 3-channel image (held as NHWC4) into 64 channels,
// on the f32 MFMA with the input patch of a 2-D output tile resident in LDS.
//
//   IResNet conv1   3x3 / stride 1 / pad 1, 112x112 -> 112x112x64  (K = 27)
//   ResNet50V2 conv1_conv  7x7 / stride 2 / pad 3, 112x112 -> 56x56x64  (K = 147)
//
// As an implicit GEMM on the general kernel (conv.hip) these layers pad every tap to four channels and K to a
// multiple of 32 (36 -> 64: 58 % of the MFMA work wasted; 196 -> 224: 34 %) and gather 16-byte pieces per tap.
// Here K is the true 3 * KH * KW (padded to the MFMA's k = 2), a block owns a TH x TW tile of OUTPUT pixels of one
// image, its input patch ((TH-1)*S + KH rows x (TW-1)*S + KW columns x 3 channel planes) sits in LDS, and an operand
// element of (pixel, k) is one ds_read_b32 at  lane_base(pixel) + offset(k):  the im2col matrix is never formed.
//
// MFMA rows = 32 pixels of the wave (2 rows x 16 columns of the tile, or 4 x 8), columns = output channels (weights from LDS,
// [k][64]).  A lane then holds ONE channel (two: p and 32 + p) of 16 pixels: its epilogue constants are six
// registers per tile, and each accumulator register is stored as it stands -- 32 consecutive channels of two
// pixels per instruction, two 128-byte segments, the full-rate store shape (16-byte pieces of 64 different rows,
// the first version, ran the layer store-bound at 2.5 TB/s; torch.fill_ reaches 7 TB/s on the same buffers).
// Blocks are persistent (grid-stride over tiles); the next tile's patch is fetched into registers before the
// MFMAs of the current one and written to the other LDS patch buffer after its epilogue: one barrier per tile.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "dif_internal.hpp"
#include "ops.hpp"

namespace dif {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// WR x WC = 32: the pixels of one wave (rows x columns); BR x BC waves make the block's tile.  Tiles that do not divide
// the map compute masked pixels, so the shapes follow the maps: 2x16 pixels, 4x1 waves = 8x16 tiles for IResNet's
// 112x112 output; 4x8 pixels, 1x7 waves = 4x56 tiles for ResNet50V2's 56x56 (16x16 tiles wasted 30 % of the MFMA work).
template <int KH_, int KW_, int S_, int PAD_, int WR_, int WC_, int BR_, int BC_>
struct StemCfg {
  static constexpr int KH = KH_, KW = KW_, S = S_, PAD = PAD_, WR = WR_, WC = WC_, BR = BR_, BC = BC_;
  static_assert(WR * WC == 32 && WC >= 8 && WR % 2 == 0, "a wave owns 32 pixels; 4h + i stays inside a row; even tile origins");
  static constexpr int WAVES = BR * BC;
  static constexpr int NT = 64 * WAVES;
  static constexpr int TH = WR * BR, TW = WC * BC;         // output tile of a block
  static constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
  static constexpr int PWP = PW | 1;                       // odd row pitch
  static constexpr int PP = PH * PWP;                      // floats per channel plane
  static constexpr int K = KH * KW * 3, KSTEPS = (K + 1) / 2;
  static constexpr int NPL = (PH * PW + NT - 1) / NT;      // patch pixels per thread
  static constexpr int W_FLOATS = KSTEPS * 2 * 64, PATCH_FLOATS = 3 * PP;
  static constexpr int LDS_BYTES = (W_FLOATS + 2 * PATCH_FLOATS) * 4;
  // patch offset of GEMM index k = (kh*KW + kw)*3 + c  (k >= K: any valid element, its weight row is zero)
  static constexpr int off(int k) {
    const int kk = k < K ? k : K - 1;
    const int tap = kk / 3, c = kk - tap * 3;
    const int kh = tap / KW, kw = tap - kh * KW;
    return c * PP + kh * PWP + kw;
  }
};

struct StemArgs {
  const float* x;   // [N,H,W,4]
  const float* w;   // Keras HWIO [KH][KW][3][64]
  const float *scale, *shift, *alpha, *scale2, *shift2, *alpha2;
  float *y, *y2;    // [N,Ho,Wo,64]
  int N, H, W, Ho, Wo, act, act2;
  int y_sub;        // y keeps the even (ho, wo) pixels only, densely (ConvArgs::y_sub)
  int tiles_h, tiles_w, tiles;
};

__device__ __forceinline__ float stem_act(float v, int act, float alpha) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_PRELU) return v >= 0.f ? v : v * alpha;
  if (act == ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}

template <class C>
__global__ __launch_bounds__(C::NT, 4) void stem_mfma_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wts = smem;                                      // [KSTEPS*2][64]
  float* patch = wts + C::W_FLOATS;                       // [2][3][PP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < C::W_FLOATS; i += C::NT) wts[i] = i < C::K * 64 ? a.w[i] : 0.f;
  const int p = lane & 31, h = lane >> 5;
  const int wave_r = wave / C::BC, wave_c = wave - wave_r * C::BC;
  const int ty = wave_r * C::WR + p / C::WC, tx = wave_c * C::WC + p % C::WC;   // this lane's output pixel inside the tile
  const int lane_base = ty * C::S * C::PWP + tx * C::S;
  // k = 2j + h: offset(2j + 1) - offset(2j) takes three values (next channel / next tap / next kernel row)
  const int base_c = lane_base + h * C::PP;
  const int base_t = lane_base + h * (1 - 2 * C::PP);
  const int base_r = lane_base + h * (C::PWP - (C::KW - 1) - 2 * C::PP);
  const int wlane = h * 64 + p;
  // this lane's two output channels (p and 32 + p) and their epilogue constants
  float sc[2], sh[2], al[2], sc2[2], sh2[2], al2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int c = 32 * t + p;
    sc[t] = a.scale ? a.scale[c] : 1.f;
    sh[t] = a.shift ? a.shift[c] : 0.f;
    al[t] = a.alpha ? a.alpha[c] : 0.f;
    sc2[t] = a.scale2 ? a.scale2[c] : 1.f;
    sh2[t] = a.shift2 ? a.shift2[c] : 0.f;
    al2[t] = a.alpha2 ? a.alpha2[c] : 0.f;
  }

  int tile = blockIdx.x;
  int tn = 0, th0 = 0, tw0 = 0;
  auto decode = [&](int t) {
    const int per_img = a.tiles_h * a.tiles_w;
    tn = t / per_img;
    const int r = t - tn * per_img;
    const int q = r / a.tiles_w;
    th0 = q * C::TH;
    tw0 = (r - q * a.tiles_w) * C::TW;
  };
  f32x4 pr[C::NPL];
  auto fetch = [&]() {
    const int hi0 = th0 * C::S - C::PAD, wi0 = tw0 * C::S - C::PAD;
#pragma unroll
    for (int j = 0; j < C::NPL; ++j) {
      const int idx = tid + C::NT * j;
      const int py = idx / C::PW, px = idx - py * C::PW;
      const int hi = hi0 + py, wi = wi0 + px;
      const bool ok = idx < C::PH * C::PW && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      pr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) pr[j] = *reinterpret_cast<const f32x4*>(a.x + (((int64_t)tn * a.H + hi) * a.W + wi) * 4);
    }
  };
  auto stage = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < C::NPL; ++j) {
      const int idx = tid + C::NT * j;
      const int py = idx / C::PW, px = idx - py * C::PW;
      if (idx < C::PH * C::PW) {
        float* d = dst + py * C::PWP + px;
        d[0] = pr[j][0];
        d[C::PP] = pr[j][1];
        d[2 * C::PP] = pr[j][2];
      }
    }
  };
  if (tile < a.tiles) {
    decode(tile);
    fetch();
    stage(patch);
  }
  __syncthreads();
  int cur = 0;
  while (tile < a.tiles) {
    const int n = tn, th0w = th0, tw0w = tw0;             // this tile (decode() below moves on to the next one)
    const int next = tile + gridDim.x;
    if (next < a.tiles) {
      decode(next);
      fetch();
    }
    const float* pb = patch + cur * C::PATCH_FLOATS;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int j = 0; j < C::KSTEPS; ++j) {
      const int oa = C::off(2 * j);
      const int d = C::off(2 * j + 1) - oa;
      const int base = d == C::PP ? base_c : (d == 1 - 2 * C::PP ? base_t : (d == 0 ? lane_base : base_r));
      const float b = pb[base + oa];
      const float a0 = wts[j * 128 + wlane];
      const float a1 = wts[j * 128 + wlane + 32];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a0, acc0, 0, 0, 0);    // rows = the wave's 32 pixels, columns = channels
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a1, acc1, 0, 0, 0);
    }
    // epilogue in the accumulator layout: acc_t[4q + i] = channel 32t + p of the wave's pixel 8q + 4h + i, i.e. row
    // (8q + i) / WC, column (8q + i) % WC + 4h of the wave's pixels.  One register of the accumulator is stored by
    // one instruction: 32 consecutive channels of two pixels = two 128-byte segments (MI355X_MICROARCH.md: the
    // full-rate shape).
    const int ho0 = th0w + wave_r * C::WR, wo0 = tw0w + wave_c * C::WC + 4 * h;
    const int64_t o0 = (((int64_t)n * a.Ho + ho0) * a.Wo + wo0) * 64 + p;
    const int64_t oy0 = a.y_sub ? (((int64_t)n * ((a.Ho + 1) >> 1) + (ho0 >> 1)) * ((a.Wo + 1) >> 1) + (wo0 >> 1)) * 64 + p : o0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int dy = (8 * q + i) / C::WC, dx = (8 * q + i) % C::WC;
          if (ho0 + dy < a.Ho && wo0 + dx < a.Wo) {
            const float s = t == 0 ? acc0[4 * q + i] : acc1[4 * q + i];
            const float v = stem_act(fmaf(s, sc[t], sh[t]), a.act, al[t]);
            if (a.y) {
              if (!a.y_sub)
                a.y[o0 + ((int64_t)dy * a.Wo + dx) * 64 + 32 * t] = v;
              else if ((dy & 1) == 0 && (dx & 1) == 0)          // tiles and waves start on even rows and columns
                a.y[oy0 + ((int64_t)(dy >> 1) * ((a.Wo + 1) >> 1) + (dx >> 1)) * 64 + 32 * t] = v;
            }
            if (a.y2) a.y2[o0 + ((int64_t)dy * a.Wo + dx) * 64 + 32 * t] = stem_act(fmaf(v, sc2[t], sh2[t]), a.act2, al2[t]);
          }
        }
    if (next < a.tiles) stage(patch + (cur ^ 1) * C::PATCH_FLOATS);
    __syncthreads();
    cur ^= 1;
    tile = next;
  }
}

template <class C>
static int stem_launch(const StemArgs& a0, hipStream_t st) {
  StemArgs a = a0;
  a.tiles_h = (a.Ho + C::TH - 1) / C::TH;
  a.tiles_w = (a.Wo + C::TW - 1) / C::TW;
  const int64_t tiles = (int64_t)a.N * a.tiles_h * a.tiles_w;
  if (tiles == 0) return 0;
  if (tiles > 0x7fffffff) return set_error("stem: %lld tiles exceed the index range", (long long)tiles);
  a.tiles = (int)tiles;
  static int cus[64], bpc[64];
  int dev = 0;
  DIF_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return set_error("stem: device ordinal %d out of range", dev);
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    DIF_HIP(hipGetDeviceProperties(&prop, dev));
    if (C::LDS_BYTES > 48 * 1024)
      DIF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_mfma_kernel<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    int nb = 0;
    DIF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stem_mfma_kernel<C>, C::NT, C::LDS_BYTES));
    bpc[dev] = nb > 0 ? nb : 1;
    cus[dev] = prop.multiProcessorCount;
  }
  const int64_t resident = (int64_t)cus[dev] * bpc[dev];
  const unsigned blocks = (unsigned)(tiles < resident ? tiles : resident);
  hipLaunchKernelGGL(stem_mfma_kernel<C>, dim3(blocks), dim3(C::NT), C::LDS_BYTES, st, a);
  DIF_HIP(hipGetLastError());
  return 0;
}

bool stem_mfma_applies(int KH, int KW, int stride, int pad_t, int pad_l, int Cout) {
  if (Cout != 64 || pad_t != pad_l) return false;
  return (KH == 3 && KW == 3 && stride == 1 && pad_t == 1) || (KH == 7 && KW == 7 && stride == 2 && pad_t == 3);
}

int stem_mfma_run(const float* x, const float* w_hwio, const float* scale, const float* shift, const float* alpha,
                  const float* scale2, const float* shift2, const float* alpha2, float* y, float* y2, int N, int H, int W,
                  int Ho, int Wo, int KH, int stride, int act, int act2, int y_sub, hipStream_t st) {
  StemArgs a;
  a.x = x;
  a.w = w_hwio;
  a.scale = scale;
  a.shift = shift;
  a.alpha = alpha;
  a.scale2 = scale2;
  a.shift2 = shift2;
  a.alpha2 = alpha2;
  a.y = y;
  a.y2 = y2;
  a.N = N;
  a.H = H;
  a.W = W;
  a.Ho = Ho;
  a.Wo = Wo;
  a.act = act;
  a.act2 = act2;
  a.y_sub = y_sub;
  a.tiles_h = a.tiles_w = a.tiles = 0;
  if (KH == 3 && stride == 1) return stem_launch<StemCfg<3, 3, 1, 1, 2, 16, 4, 1>>(a, st);
  if (KH == 7 && stride == 2) {
    // one or two images (round 5: the reference embeds ONE image per call): the 4 x 56 tiles are 14 blocks per image on a
    // chip of 256 CUs -- 4 x 8 tiles of one wave each (98 per image) instead; same arithmetic per output pixel
    if ((int64_t)N * ((Ho + 3) / 4) * ((Wo + 55) / 56) <= 32) return stem_launch<StemCfg<7, 7, 2, 3, 4, 8, 1, 1>>(a, st);
    return stem_launch<StemCfg<7, 7, 2, 3, 4, 8, 1, 7>>(a, st);
  }
  return set_error("stem: no kernel for a %dx%d / stride %d first layer", KH, KH, stride);
}

}  // namespace dif
