// Argument blocks and host launchers of the layer kernels (conv.hip, elementwise.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dif {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_PRELU = 2, ACT_RELU6 = 3 };   // RELU6: min(max(x, 0), 6) (MobileNetV2)

// Division of a non-negative int (< 2^31) by a launch-invariant divisor: one mul-hi and a
// shift instead of the ~40-instruction expansion of a runtime integer division.
struct FastDiv {
  uint32_t mul, shift, d;
#if defined(__HIPCC__)
  __device__ __forceinline__ int div(int n) const {
    return d == 1 ? n : (int)(__umulhi((uint32_t)n, mul) >> shift);
  }
  __device__ __forceinline__ void divmod(int n, int& q, int& r) const {
    q = div(n);
    r = n - q * (int)d;
  }
#endif
};
inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = (uint32_t)d;
  if (d <= 1) {
    f.mul = 0;
    f.shift = 0;
    f.d = 1;
    return f;
  }
  int lg = 0;
  while ((1u << lg) < (uint32_t)d) ++lg;        // ceil(log2 d)
  const unsigned p = 31 + lg;
  f.mul = (uint32_t)(((1ull << p) + (uint32_t)d - 1) / (uint32_t)d);
  f.shift = p - 32;
  return f;
}

enum { CONV_OFF_PATCH = 1, CONV_OFF_PATCH2D = 2, CONV_OFF_BD = 4, CONV_OFF_T2 = 16, CONV_OFF_TN = 32, CONV_OFF_SK2 = 64, CONV_OFF_MT = 128 };

struct ConvArgs {
  const float* x;       // [N,H,W,Cin] NHWC
  const float* w;       // packed [Cout][Kpad], zero padded; k order per k_order
  const float* w_frag;  // the same matrix in MFMA-fragment order for the B-direct patch mainloop (conv.hip), or null:
                        // [ceil(Cout/32)][Kpad/32][s 2][u 2][h 2][n 32][t 4] <- w[32 nt + n][32 ks + 16 s + 8 h + 4 u + t]
  uint32_t w_frag_bytes;
  const float* w_f16;   // the same matrix in the one-image kernel's order (conv_minitile.hpp), or null:
                        // [ceil(Cout/16)][Kpad/16][lane 64][t 4] <- w[16 ct + (lane & 15)][16 c + 4 (lane >> 4) + t]
  uint32_t w_f16_bytes;
  const void* w3f;      // split-bf16 mode, 3x3 / stride 1 layers: the three bf16 planes in MFMA-fragment order (conv.hip:
  uint32_t w3f_bytes;   // gemm_mainloop_patch_bf3), [ceil(Cout/32)][Kpad/32][s 2][plane 3][lane 64][8]
  int bf_terms;         // 3 (or 0): all three planes, six products; 2: the hi and mid planes only, three products ("bf16x2")
  int k_order;          // 0: k = (kh*KW + kw)*Cin + ci (tap-major)
                        // 1: k = ((ci/32)*KH*KW + kh*KW + kw)*32 + ci%32 (channel-block-major, Cin % 32 == 0):
                        //    consecutive K-steps sweep the taps of ONE 32-channel slice, i.e. re-read the
                        //    same 128-byte lines of the same pixels -> L2 hits instead of fabric traffic
  float* y;             // [N,Ho,Wo,Cout]
  float* y2;            // optional second output (same shape) or null
  const float* scale;   // per Cout or null (= 1)
  const float* shift;   // per Cout or null (= 0)
  const float* alpha;   // PReLU slopes (act == ACT_PRELU)
  const float* res;     // residual [N,res_H,res_W,Cout] or null
  const float* scale2;
  const float* shift2;
  const float* alpha2;
  // optional pre-activation applied to the INPUT while it is gathered (per input channel):
  // x' = pre_act(x * pre_scale[ci] + pre_shift[ci]); padded taps stay exactly zero
  const float* pre_scale;
  const float* pre_shift;
  int pre_act;
  int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l;
  int Kpad;
  int M;                // N*Ho*Wo
  int act, act2;
  int res_H, res_W, res_stride;
  // Output view: y (and y2) may be a channel slice of a wider tensor (inception concat) and/or
  // sit inside a larger zero-initialised map (ZeroPadding2D after the layer):
  //   y[((n*y_H + ho + y_oy)*y_W + wo + y_ox)*y_ld + y_coff + c]
  // Plain output: y_ld = Cout, y_coff = 0, y_H = Ho, y_W = Wo, y_oy = y_ox = 0.
  int y_ld, y_coff, y_H, y_W, y_oy, y_ox;
  // y_sub = 1: y (not y2) keeps only the pixels with even ho and wo, as a dense [N, (Ho+1)/2, (Wo+1)/2, Cout] tensor --
  // its only reader is a 1x1 / stride 2 convolution (net.hip: finalize), which then runs at stride 1
  int y_sub;
  // stream-K workspace (owned by the caller): one partial-accumulator slab and one flag per
  // persistent block; sk_epoch is unique per launch so flags never need clearing
  float* sk_slab;
  unsigned* sk_flag;
  unsigned sk_epoch;
  int sk_max_blocks;
  int sk_spin_limit;    // polls before the owner computes a missing K range itself; < 0: always (test hook)
  int bdp_mode;         // conv_bdp_kernel for the patch layers: 0 where it pays (conv.hip: conv_bdp_ok), 1 never (set where
                        // launches are short and the lanes run half-chip grids: measured 1-2 % slower there; Net option
                        // "bdp" = 0), 2 wherever its restrictions allow (Net option "bdp" = 2: the parity tests)
  int use_pipe;         // 0: never take the software-pipelined kernel (Net option "pipe"; tests compare both paths)
  int epi_fast;         // the output takes conv.hip's lean epilogue (plain geometry, 32-bit offsets); filled by conv_run
  int dbg;              // development aid (ablation bits of the kernel under work); 0 in production
  int lanes;            // executor lanes launching side by side (0 / 1: this forward has the chip to itself): sk2_plan's wider splits
  unsigned off;         // kernel families switched off (Net options "patch", "patch2d", "bd" = 0): CONV_OFF_* bits
  // development aid (tools/ubench/conv_trace.hip), null in the library: 4 x u64 per hardware block =
  // s_memrealtime (100 MHz) at entry / after the first mainloop / at exit, and the HW_ID register
  unsigned long long* trace;
  // launch-invariant divisors (filled by conv_run)
  FastDiv fd_howo, fd_wo, fd_cin, fd_kw, fd_ks, fd_tiles_n, fd_taps;
  FastDiv fd_wp, fd_rpi;   // halo-patch path (conv.hip: PatchA): padded row width W + 2, padded rows per image H + 1
  FastDiv fd_t2_w, fd_t2_img;   // its 8x8-tile form (PatchA2D): tiles per row W / 8, tiles per image (H / 8) * (W / 8)
};

int conv_max_blocks();        // persistent blocks the kernel may use on this device
size_t conv_slab_floats();    // floats per slab
int conv_run(const ConvArgs& a, hipStream_t st);
const char* conv_last_kernel();   // `family<tile, variant>` of the kernel this thread's last conv_run launched
// which form of the split-bf16 kernel a 3x3 / stride 1 layer on an H x W map with Cout filters takes (0: none: f32 kernels)
int conv_bf3p_form(int H, int W, bool batch_gt1, int Cout);

enum { POOL_MAX = 0, POOL_L2 = 1, POOL_AVG = 2 };
struct PoolArgs {
  const float* x;   // [N,H,W,C]
  float* y;         // [N,Ho,Wo,C] or a view (see ConvArgs)
  float* y2;        // optional relu(y*scale2 + shift2), plain layout
  const float* scale2;
  const float* shift2;
  int N, H, W, C, Ho, Wo, k, stride, pad_t, pad_l;
  int zero_pad;     // 1: padded taps contribute 0 (explicit ZeroPadding2D before a VALID pool)
  int act2;
  int mode;         // POOL_MAX | POOL_L2 (sqrt(mean(x^2) * k*k), networks/inceptionv3.py:160-163) | POOL_AVG
  int y_ld, y_coff, y_H, y_W, y_oy, y_ox;
};
int maxpool_run(const PoolArgs& a, hipStream_t st);
// nearest-neighbour x2 upsampling (keras UpSampling2D(2)) into an output view
int upsample2_run(const float* x, float* y, int N, int H, int W, int C, int y_ld, int y_coff, hipStream_t st);
// dense [N,H,W,C] -> channel slice of a wider tensor (concatenate with an earlier layer)
int copy_to_view_run(const float* x, float* y, int64_t npix, int C, int y_ld, int y_coff, hipStream_t st);
// tf.nn.lrn over the channel axis: y = x / (bias + alpha * sum_{|j-c|<=radius} x_j^2)^beta
int lrn_run(const float* x, float* y, int64_t npix, int C, int radius, float bias, float alpha, float beta,
            hipStream_t st);

// input conversion to the internal NHWC4 float layout
struct InputArgs {
  const void* x;
  float* y;          // [N,H,W,4], channel 3 = 0
  int N, H, W;
  int layout, dtype; // DIF_LAYOUT_*, DIF_DTYPE_*
  float scale;
  float bias[3];
  int bgr;           // DIF_INPUT_* flags
};
int input_convert_run(const InputArgs& a, hipStream_t st);

// depthwise KxK convolution + BN + activation: y[n,ho,wo,c] = act((sum_taps x * w[tap][c]) * scale[c] + shift[c]);
// w is [K*K][C] (= Keras depthwise_kernel [K,K,C,1]); padded taps contribute zero
int dwconv_run(const float* x, const float* w, const float* scale, const float* shift, float* y, int N, int H, int W,
               int C, int K, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, hipStream_t st);
// depthwise conv whose kernel covers the whole map, + BN: y[n,c] = (sum_hw x*w) * scale + shift
int dwfull_run(const float* x, const float* w, const float* scale, const float* shift, float* y, int N, int HW,
               int C, hipStream_t st);
// direct 3x3 / stride 1 / pad 1 convolution of a 3-channel input held as NHWC4, Cout 32; w = Keras HWIO
// [3][3][3][Cout]; epilogue as conv_run's
int stem3x3_run(const float* x, const float* w_hwio, const float* scale, const float* shift, const float* alpha,
                const float* scale2, const float* shift2, const float* alpha2, float* y, float* y2, int N, int H, int W,
                int Cout, int act, int act2, hipStream_t st);
// 3-channel first layers with 64 filters on the MFMA, input patch resident in LDS (stem.hip): 3x3 / stride 1 / pad 1 and
// 7x7 / stride 2 / pad 3; x NHWC4, w = Keras HWIO [KH][KW][3][64], plain [N,Ho,Wo,64] outputs; epilogue as conv_run's
bool stem_mfma_applies(int KH, int KW, int stride, int pad_t, int pad_l, int Cout);
int stem_mfma_run(const float* x, const float* w_hwio, const float* scale, const float* shift, const float* alpha,
                  const float* scale2, const float* shift2, const float* alpha2, float* y, float* y2, int N, int H, int W,
                  int Ho, int Wo, int KH, int stride, int act, int act2, int y_sub, hipStream_t st);
// GDC head tail in one launch (networks/triplet.py:129-138): depthwise over the whole map + BN -> 1x1 conv (512 -> E) ->
// dense (E -> E) -> l2_normalize; x [N][HW][512], w_dw [HW][512], w_pw [512][E], w_dense [E][E], y [N][E]; E <= 1024
// ws: GDC_TAIL_WS_FLOATS floats, zero when first used, or null -- with it one or two images are spread over E / 32 blocks
constexpr int GDC_TAIL_WS_FLOATS = 4096 + 16;
int gdc_tail_run(const float* x, const float* w_dw, const float* scale, const float* shift, const float* w_pw,
                 const float* w_dense, float* y, int N, int HW, int E, float eps, float* ws, hipStream_t st);
// y = x * rsqrt(max(sum(x^2), eps)) per row
int l2norm_run(const float* x, float* y, int N, int D, float eps, hipStream_t st);

}  // namespace dif
