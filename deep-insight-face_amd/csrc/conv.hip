// Implicit-GEMM convolution on the f32 MFMA, NHWC, with fused epilogues.
//
//   GEMM view: M = N*Ho*Wo output pixels (A operand, gathered on the fly from the NHWC
//   input: "im2col" never exists in memory), N = Cout (B operand, weights packed
//   [Cout][Kpad] with k = (kh*KW + kw)*Cin + ci), K = KH*KW*Cin.
//
//   Epilogue (per output element, c = output channel):
//       v  = acc * scale[c] + shift[c]          (folded bias and/or inference BatchNorm)
//       v  = act(v)                             (none | ReLU | PReLU(alpha[c]))
//       v += residual[...]                      (shortcut, optionally spatially subsampled)
//       y  = v
//       y2 = act2(v * scale2[c] + shift2[c])    (optional second output: the NEXT block's
//                                                pre-activation BN, so it never needs a pass
//                                                of its own)
//
// The zero halo and every tile tail come from buffer-descriptor range checks (an
// out-of-range offset loads zeros), so the loaders are branch-free.
#include "gemm_core.hpp"
#include "dif_internal.hpp"
#include "ops.hpp"

namespace dif {

// A-operand loader: gathers BM output pixels x 32 k-values per step.
template <int N>
struct ConvALoader {
  __amdgpu_buffer_rsrc_t rsrc;
  int32_t base[N];   // byte offset of (n - n_first, hi0, wi0, 0) relative to the tile's first image (may be < 0)
  int32_t hw0[N];    // hi0 in the high 16 bits, wi0 in the low 16 bits (biased by 0x4000 each)
  int H, W, Cin, KW, taps;
  bool fast;         // Cin % 32 == 0: one (kh, kw) per K-step, block-uniform

  __device__ __forceinline__ ConvALoader(const ConvArgs& a, int64_t m0) {
    const int tid = threadIdx.x;
    H = a.H;
    W = a.W;
    Cin = a.Cin;
    KW = a.KW;
    taps = a.KH * a.KW;
    fast = (a.Cin % BK) == 0;
    const int HoWo = a.Ho * a.Wo;
    const int64_t n_first = (int)m0 / HoWo;
    const int64_t img_elems = (int64_t)a.H * a.W * a.Cin;
    const int64_t imgs_left = a.N - n_first;
    int64_t span = (T_MAX_BM + HoWo - 1) / HoWo + 1;     // images a tile can touch
    if (span > imgs_left) span = imgs_left;
    rsrc = make_rsrc(a.x + n_first * img_elems, (uint32_t)(span * img_elems * 4));
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int m = (int)m0 + (tid >> 3) + 32 * i;
      if (m < a.M) {
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        const int ho = r / a.Wo;
        const int wo = r - ho * a.Wo;
        const int hi0 = ho * a.stride - a.pad_t;
        const int wi0 = wo * a.stride - a.pad_l;
        base[i] = (int32_t)((((int64_t)(n - n_first) * a.H + hi0) * a.W + wi0) * a.Cin * 4) + (tid & 7) * 16;
        hw0[i] = ((hi0 + 0x4000) << 16) | (wi0 + 0x4000);
      } else {
        base[i] = 0;
        hw0[i] = 0;   // hi0 = wi0 = -0x4000: never inside the image
      }
    }
  }
  static constexpr int T_MAX_BM = 256;

  __device__ __forceinline__ void load(int kstep, f32x4 (&r)[N]) const {
    int kh, kw, toff;
    bool tap_ok = true;
    if (fast) {
      const int k0 = kstep * BK;
      const int tap = k0 / Cin;
      const int ci0 = k0 - tap * Cin;
      kh = tap / KW;
      kw = tap - kh * KW;
      toff = ((kh * W + kw) * Cin + ci0) * 4;
    } else {
      const int k = kstep * BK + (threadIdx.x & 7) * 4;
      const int tap = k / Cin;
      const int ci = k - tap * Cin;
      kh = tap / KW;
      kw = tap - kh * KW;
      tap_ok = tap < taps;
      toff = ((kh * W + kw) * Cin + ci) * 4 - (threadIdx.x & 7) * 16;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int hi = (hw0[i] >> 16) - 0x4000 + kh;
      const int wi = (hw0[i] & 0xffff) - 0x4000 + kw;
      const bool ok = tap_ok && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const uint32_t off = ok ? (uint32_t)(base[i] + toff) : OOB;
      r[i] = buf_load4(rsrc, off);
    }
  }
};

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_PRELU) return v >= 0.f ? v : v * alpha;
  return v;
}

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
  using T = Tile<WM, WN>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * T::BM;
  const int n0 = blockIdx.y * T::BN;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  ConvALoader<T::NA> al(a, m0);
  RowLoader<T::NB> bl(a.w + (int64_t)n0 * a.Kpad, (int64_t)a.Cout - n0, a.Kpad);
  gemm_mainloop<WM, WN>(al, bl, a.Kpad / BK, smem, acc);

  const bool strided_res = a.res != nullptr && (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo);
  const int HoWo = a.Ho * a.Wo;
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    const int c = n0 + (wc * WN + n) * 32 + (lane & 31);
    const bool cok = c < a.Cout;
    const int cc = cok ? c : 0;
    const float sc = a.scale ? a.scale[cc] : 1.f;
    const float sh = a.shift ? a.shift[cc] : 0.f;
    const float al1 = a.alpha ? a.alpha[cc] : 0.f;
    const float sc2 = a.scale2 ? a.scale2[cc] : 1.f;
    const float sh2 = a.shift2 ? a.shift2[cc] : 0.f;
    const float al2 = a.alpha2 ? a.alpha2[cc] : 0.f;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wr * WM + m) * 32 + frag_row(lane, r);
        if (row < a.M && cok) {
          float v = fmaf(acc[m][n][r], sc, sh);
          v = apply_act(v, a.act, al1);
          if (a.res) {
            int64_t ri = row;
            if (strided_res) {
              const int64_t img = row / HoWo;
              const int rr = (int)(row - img * HoWo);
              const int ho = rr / a.Wo;
              const int wo = rr - ho * a.Wo;
              ri = (img * a.res_H + (int64_t)ho * a.res_stride) * a.res_W + (int64_t)wo * a.res_stride;
            }
            v += a.res[ri * a.Cout + c];
          }
          if (a.y) a.y[row * a.Cout + c] = v;
          if (a.y2) a.y2[row * a.Cout + c] = apply_act(fmaf(v, sc2, sh2), a.act2, al2);
        }
      }
    }
  }
}

template <int WM, int WN>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  using T = Tile<WM, WN>;
  static bool attr_set = false;
  auto kern = conv_igemm_kernel<WM, WN>;
  if (!attr_set) {
    DIF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                T::LDS_BYTES));
    attr_set = true;
  }
  dim3 grid((unsigned)((a.M + T::BM - 1) / T::BM), (unsigned)((a.Cout + T::BN - 1) / T::BN));
  hipLaunchKernelGGL(kern, grid, dim3(256), T::LDS_BYTES, st, a);
  DIF_HIP(hipGetLastError());
  return 0;
}

int conv_tile_choice(int64_t M, int Cout) {
  // 0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64.  Prefer the largest tile that still
  // yields >= ~2 blocks per CU; small layers take smaller tiles to fill the chip.
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
  const int64_t want = 512;
  if (Cout > 64) {
    if (blocks(128, 128) >= want) return 0;
    if (blocks(64, 128) >= want) return 2;
    return blocks(64, 64) > blocks(64, 128) ? 3 : 2;
  }
  if (blocks(128, 64) >= want) return 1;
  return 3;
}

int conv_run(const ConvArgs& a, int tile, hipStream_t st) {
  if (a.M <= 0) return 0;
  if (a.Cin % 4 != 0) return set_error("conv: Cin must be a multiple of 4 (got %d)", a.Cin);
  if (a.Kpad % BK != 0) return set_error("conv: Kpad must be a multiple of %d", BK);
  if (a.H >= 0x3f00 || a.W >= 0x3f00) return set_error("conv: spatial size too large");
  {
    const int64_t howo = (int64_t)a.Ho * a.Wo;
    const int64_t span = (256 + howo - 1) / howo + 1;
    if (span * a.H * a.W * a.Cin * 4 >= 0x7fffffffLL)
      return set_error("conv: a 256-pixel tile spans more than 2 GiB of input (%dx%dx%d)", a.H, a.W, a.Cin);
  }
  if (tile < 0) tile = conv_tile_choice(a.M, a.Cout);
  switch (tile) {
    case 0: return launch_conv<2, 2>(a, st);
    case 1: return launch_conv<2, 1>(a, st);
    case 2: return launch_conv<1, 2>(a, st);
    default: return launch_conv<1, 1>(a, st);
  }
}

}  // namespace dif
