// Argument blocks and host launchers of the layer kernels (conv.hip, elementwise.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dif {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_PRELU = 2 };

struct ConvArgs {
  const float* x;       // [N,H,W,Cin] NHWC
  const float* w;       // packed [Cout][Kpad], k = (kh*KW + kw)*Cin + ci, zero padded
  float* y;             // [N,Ho,Wo,Cout]
  float* y2;            // optional second output (same shape) or null
  const float* scale;   // per Cout or null (= 1)
  const float* shift;   // per Cout or null (= 0)
  const float* alpha;   // PReLU slopes (act == ACT_PRELU)
  const float* res;     // residual [N,res_H,res_W,Cout] or null
  const float* scale2;
  const float* shift2;
  const float* alpha2;
  int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad_t, pad_l;
  int Kpad;
  int M;                // N*Ho*Wo
  int act, act2;
  int res_H, res_W, res_stride;
  // stream-K workspace (owned by the caller): one partial-accumulator slab and one flag per
  // persistent block; sk_epoch is unique per launch so flags never need clearing
  float* sk_slab;
  unsigned* sk_flag;
  unsigned sk_epoch;
  int sk_max_blocks;
  int sk_spin_limit;    // polls before the owner computes a missing K range itself; < 0: always (test hook)
};

int conv_max_blocks();        // persistent blocks the kernel may use on this device
size_t conv_slab_floats();    // floats per slab
int conv_tile_choice(int64_t M, int Cout);
int conv_run(const ConvArgs& a, int tile, hipStream_t st);

struct PoolArgs {
  const float* x;   // [N,H,W,C]
  float* y;         // [N,Ho,Wo,C]
  float* y2;        // optional relu(y*scale2 + shift2)
  const float* scale2;
  const float* shift2;
  int N, H, W, C, Ho, Wo, k, stride, pad_t, pad_l;
  int zero_pad;     // 1: padded taps contribute 0 (explicit ZeroPadding2D before a VALID pool)
  int act2;
};
int maxpool_run(const PoolArgs& a, hipStream_t st);

// input conversion to the internal NHWC4 float layout
struct InputArgs {
  const void* x;
  float* y;          // [N,H,W,4], channel 3 = 0
  int N, H, W;
  int layout, dtype; // DIF_LAYOUT_*, DIF_DTYPE_*
  float scale;
  float bias[3];
  int bgr;
};
int input_convert_run(const InputArgs& a, hipStream_t st);

// depthwise conv whose kernel covers the whole map, + BN: y[n,c] = (sum_hw x*w) * scale + shift
int dwfull_run(const float* x, const float* w, const float* scale, const float* shift, float* y, int N, int HW,
               int C, hipStream_t st);
// y = x * rsqrt(max(sum(x^2), eps)) per row
int l2norm_run(const float* x, float* y, int N, int D, float eps, hipStream_t st);

}  // namespace dif
