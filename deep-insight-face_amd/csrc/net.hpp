// Embedding network: a static list of layer ops over pooled NHWC activation buffers,
// executed as a fixed sequence of kernel launches on the caller's stream.
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "ops.hpp"

namespace dif {

struct Param {
  std::string name;
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool set = false;
  int64_t count() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct BNRef {
  int gamma = -1, beta = -1, mean = -1, var = -1;
  float eps = 0.f;
  bool valid() const { return gamma >= 0; }
};

struct TensorDesc {
  int H = 0, W = 0, C = 0;
  // view into a wider / larger parent tensor (inception concat, ZeroPadding2D after a layer)
  int parent = -1, coff = 0, oy = 0, ox = 0;
  int buf = -1;
  int first_def = -1, last_use = -1;
  int64_t elems() const { return (int64_t)H * W * C; }
};

enum OpKind { OP_INPUT, OP_CONV, OP_MAXPOOL, OP_DWFULL, OP_L2NORM, OP_LRN, OP_ZERO, OP_UPSAMPLE, OP_COPY, OP_DWCONV,
              OP_GDCTAIL };

struct Op {
  OpKind kind = OP_CONV;
  std::string name;
  int x = -1, y = -1, y2 = -1, res = -1;
  // conv / pool geometry
  int KH = 1, KW = 1, stride = 1, pad_t = 0, pad_l = 0;
  int Cin = 0, Cin_true = 0, Cout = 0;
  int res_stride = 1;
  bool y_sub = false;         // output y keeps its even pixels only (finalize: its one reader is a 1x1 / stride 2 convolution)
  int zero_pad = 0;
  int pool_mode = POOL_MAX;
  float const_alpha = 0.f;    // act == ACT_PRELU with one slope for every channel (LeakyReLU)
  bool chw_flatten = false;
  int k_order = 0;            // see ConvArgs::k_order   // dense after an NCHW-order flatten: permute kernel rows at pack time
  // parameters
  int w = -1, bias = -1, alpha = -1, alpha2 = -1;
  int w_pw = -1, w_dense = -1;   // OP_GDCTAIL: the 1x1 convolution and the dense layer behind the depthwise one
  BNRef bn, bn2;
  BNRef pre_bn;               // pre-activation BN applied to the input while it is gathered
  int pre_act = ACT_NONE;
  int act = ACT_NONE, act2 = ACT_NONE;
  // device side (filled by finalize)
  float* d_w = nullptr;
  float* d_w_raw = nullptr;   // 3-channel first layer: the Keras HWIO kernel as it is, for the stem kernels
  bool stem_mfma = false;     // d_w_raw feeds stem.hip's MFMA kernel (64 filters) instead of elementwise.hip's direct one (32)
  float* d_w_pw = nullptr;
  float* d_w_dense = nullptr;
  float* d_w_frag = nullptr;  // the matrix in MFMA-fragment order (ConvArgs::w_frag), 3x3 / stride 1 layers on the f32 path
  uint32_t w_frag_bytes = 0;
  float* d_w_f16 = nullptr;   // ... and in the 16-column fragment order of the one-image kernel (ConvArgs::w_f16), layers it can take
  uint32_t w_f16_bytes = 0;
  void* d_w3f = nullptr;      // compute mode bf16x3: the split-bf16 planes in MFMA-fragment order (ConvArgs::w3f), 3x3 / stride 1 layers
  uint32_t w3f_bytes = 0;
  float* d_scale = nullptr;
  float* d_shift = nullptr;
  float* d_alpha = nullptr;
  float* d_scale2 = nullptr;
  float* d_shift2 = nullptr;
  float* d_alpha2 = nullptr;
  float* d_pre_scale = nullptr;
  float* d_pre_shift = nullptr;
  int Kpad = 0;
  double macs = 0;   // per image
  mutable const char* ran_kernel = nullptr;   // the conv kernel instantiation this op's last launch used (dif_net_op_info)
};

struct Net {
  std::string arch, head;
  int emd = 0, in_h = 0, in_w = 0;
  std::vector<Param> params;
  std::map<std::string, int> pindex;
  std::vector<TensorDesc> tensors;
  std::vector<Op> ops;
  int input_tensor = -1, output_tensor = -1;
  std::vector<int> extra_outputs;   // further outputs after output_tensor (detector heads)
  bool is_output(int t) const {
    return t == output_tensor || std::find(extra_outputs.begin(), extra_outputs.end(), t) != extra_outputs.end();
  }
  int64_t output_offset(int t) const {   // floats per image that precede output t in the caller's buffer
    if (t == output_tensor) return 0;
    int64_t off = tensors[output_tensor].elems();
    for (int e : extra_outputs) {
      if (e == t) return off;
      off += tensors[e].elems();
    }
    return off;
  }
  int64_t output_elems() const { return output_offset(-2); }
  // images from which a forward is worth splitting over two lanes: 64 at 112 x 112, fewer for larger inputs (the detector)
  int lane_min_images() const {
    if (const char* e = getenv("DIF_LANE_MIN_IMAGES")) return atoi(e) > 2 ? atoi(e) : 2;   // development: A/B of the split threshold
    const int64_t px = (int64_t)in_h * in_w;
    const int64_t m = (64LL * 112 * 112 + px - 1) / (px > 0 ? px : 1);
    return m < 2 ? 2 : (m > 64 ? 64 : (int)m);
  }
  float in_scale = 1.f;
  float in_bias[3] = {0.f, 0.f, 0.f};
  int bgr = 0;
  // finalize state
  bool finalized = false;
  int max_batch = 0;
  std::vector<void*> allocs;
  // a lane = one in-flight slice of the batch: its own activation buffers, stream-K workspace
  // and (beyond lane 0, which uses the caller's stream) its own HIP stream
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    int cap = 0;
    int out_n = 0, out_start = 0;   // this forward: images in the whole batch, first image of this lane (where its outputs go)
    std::vector<float*> bufs;
    float* sk_slab = nullptr;
    unsigned* sk_flag = nullptr;
    float* tail_ws = nullptr;         // gdc_tail_run's workspace (one or two images)
    unsigned sk_epoch = 0;
    int sk_max_blocks = 0;
  };
  std::vector<Lane> lanes;
  hipEvent_t ev_start = nullptr;
  bool lane_split = false;          // the lanes' persistent grids take 1 / nl of the resident slots each (short launches)
  bool lanes_active = false;        // this forward runs on all lanes (each with its share of the resident slots)
  std::vector<int64_t> buf_elems;   // per image
  int sk_max_blocks = 0;
  int sk_spin_limit = 1 << 18;
  int bf_terms = 3;                 // option "bf_terms": 3 = bf16x3 (six products), 2 = bf16x2 (hi + mid, three products)
  int compute_bf16x3 = 0;           // option "bf16x3" (set before finalize): convolutions on the split-bf16 MFMA path
  int use_pipe = 1;                 // option "pipe": 0 keeps every convolution on conv_igemm_kernel
  int use_bdp = 1;                  // option "bdp": 0 never conv_bdp_kernel, 1 where it pays, 2 wherever it can run
  int use_stem = 1;                 // option "stem": 0 runs 3-channel first layers on conv_igemm_kernel too
  int use_ysub = 1;                 // option "ysub" (before finalize): 0 keeps outputs read only at stride 2 dense
  int opt_lane_split = -1;          // option "lane_split" (before finalize): -1 by work per launch, 0 / 1 forced
  int opt_lane_prio = 0;            // option "lane_prio" (before finalize): 0 = extra lanes on least-priority streams, 1 = normal
  int conv_dbg = 0;                 // option "dbg": ConvArgs::dbg
  unsigned conv_off = 0;            // options "patch", "patch2d", "bd" = 0: CONV_OFF_* bits handed to every convolution
  int set_option(const char* key, int value);
  // dif_net_set_option; also applied from DIF_OPTIONS="key=value,..." at finalize
  static const char* const* option_names();   // null-terminated list of the keys set_option accepts

  ~Net();
  int build();                       // dispatch on arch/head
  int finalize(int max_batch);
  int embed(const void* x, int n, int layout, int dtype, float* out, hipStream_t st, float* op_ms = nullptr);
  // measurement aid (embed_clock): per-op regions of 8 x u64 records per hardware block, written by the conv kernels
  unsigned long long* trace_buf = nullptr;
  std::vector<size_t> trace_off;
  int embed_clock(const void* x, int n, int layout, int dtype, float* out, hipStream_t st, double* ghz);
  int run_op(const Op& op, Lane& L, const void* x, int n, int layout, int dtype, float* out, hipStream_t st);
  const char* kernel_name(const Op& op, int n) const;
  double flops_per_image() const;
  void release_device();

  // builder helpers
  int P(const std::string& name, std::vector<int64_t> shape);
  BNRef BN(const std::string& prefix, int C, float eps);
  int T(int H, int W, int C);
  int V(int parent, int H, int W, int C, int coff, int oy, int ox);   // view tensor
  void into(int parent, int coff, int oy = 0, int ox = 0);           // the next conv/pool writes into this view
  int out_tensor(int H, int W, int C);
  int pool(const std::string& name, int x, int k, int stride, int pad, int mode, int zero_pad, bool ceil_mode = false);
  int pend_parent = -1, pend_coff = 0, pend_oy = 0, pend_ox = 0;
  int root_of(int t) const { return tensors[t].parent >= 0 ? tensors[t].parent : t; }
  int conv(const std::string& name, int x, int KH, int KW, int stride, int pad, int Cout, bool bias,
           const BNRef& bn, int act, int alpha, int res, int res_stride, bool want_y, const BNRef& bn2, int act2,
           int* y2_out, const std::string& wsuffix = "/kernel", bool same_pad_even = false, int pad_br = -1);
  int build_yolov3();
  int build_mtcnn(int stage);   // 1 P-Net (any input size), 2 R-Net (24 x 24), 3 O-Net (48 x 48)
  int build_resnet50v2();
  void add_input();
  int build_heads(int feat);         // v1 / v2 / v3 head of triplet.py:102-141 on a backbone's feature map
  int build_vgg16();
  int build_mobilenetv2();
  int build_iresnet(const int* layers);
  int build_nn4();
};

}  // namespace dif
