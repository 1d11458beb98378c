// Network construction (layer tables), weight packing and the launch sequence.
//
// Layer tables:
//   "resnet"      keras.applications.ResNet50V2(include_top=False) as selected by
//                 deep_insight_face/networks/triplet.py:90-91 (third-party; restated from
//                 the public definition) followed by a head:
//                   v1  triplet.py:102-117    v2 (GDC + L2-norm)  triplet.py:119-141
//                   v3  triplet.py:143-146 (bare backbone)
//   "iresnet50/100"  ArcFace IResNet (not in the reference).
//
// Fusion plan: every BatchNorm / bias / activation / shortcut add is folded into the
// epilogue of the convolution that produces its input; the pre-activation BN of the NEXT
// residual block is emitted as the producing convolution's second output.  A forward is
// therefore convolutions + one max-pool + the tiny head kernels, nothing else.
#include "net.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>

#include "../../include/dif.h"
#include "dif_internal.hpp"
#include "gemm_core.hpp"

namespace dif {

static const float EPS_RESNET = 1.001e-5f;   // keras.applications.resnet BN epsilon
static const float EPS_KERAS = 1e-3f;        // keras BatchNormalization default (triplet.py:127,130)
static const float EPS_IRESNET = 1e-5f;      // torch BatchNorm default

Net::~Net() { release_device(); }

void Net::release_device() {
  for (Lane& L : lanes) {
    if (L.stream) (void)hipStreamDestroy(L.stream);
    if (L.done) (void)hipEventDestroy(L.done);
  }
  lanes.clear();
  if (ev_start) (void)hipEventDestroy(ev_start);
  ev_start = nullptr;
  for (void* p : allocs) (void)hipFree(p);
  allocs.clear();
  finalized = false;
}

int Net::P(const std::string& name, std::vector<int64_t> shape) {
  Param p;
  p.name = name;
  p.shape = shape;
  params.push_back(p);
  pindex[name] = (int)params.size() - 1;
  return (int)params.size() - 1;
}

BNRef Net::BN(const std::string& prefix, int C, float eps) {
  BNRef r;
  r.gamma = P(prefix + "/gamma", {C});
  r.beta = P(prefix + "/beta", {C});
  r.mean = P(prefix + "/moving_mean", {C});
  r.var = P(prefix + "/moving_variance", {C});
  r.eps = eps;
  return r;
}

int Net::T(int H, int W, int C) {
  TensorDesc t;
  t.H = H;
  t.W = W;
  t.C = C;
  tensors.push_back(t);
  return (int)tensors.size() - 1;
}

int Net::V(int parent, int H, int W, int C, int coff, int oy, int ox) {
  TensorDesc t;
  t.H = H;
  t.W = W;
  t.C = C;
  t.parent = parent;
  t.coff = coff;
  t.oy = oy;
  t.ox = ox;
  tensors.push_back(t);
  return (int)tensors.size() - 1;
}

void Net::into(int parent, int coff, int oy, int ox) {
  pend_parent = parent;
  pend_coff = coff;
  pend_oy = oy;
  pend_ox = ox;
}

int Net::out_tensor(int H, int W, int C) {
  if (pend_parent < 0) return T(H, W, C);
  const int v = V(pend_parent, H, W, C, pend_coff, pend_oy, pend_ox);
  pend_parent = -1;
  return v;
}

int Net::pool(const std::string& name, int x, int k, int stride, int pad, int mode, int zero_pad, bool ceil_mode) {
  const TensorDesc td = tensors[x];
  Op p;
  p.kind = OP_MAXPOOL;
  p.name = name;
  p.x = x;
  p.KH = p.KW = k;
  p.stride = stride;
  p.pad_t = p.pad_l = pad;
  p.zero_pad = zero_pad;
  p.pool_mode = mode;
  p.Cin = p.Cout = td.C;
  // ceil_mode (Caffe's pooling, MTCNN): the last window may hang over the bottom / right edge; the kernel skips taps outside
  const int rnd = ceil_mode ? stride - 1 : 0;
  p.y = out_tensor((td.H + 2 * pad - k + rnd) / stride + 1, (td.W + 2 * pad - k + rnd) / stride + 1, td.C);
  ops.push_back(p);
  return p.y;
}

// Adds a convolution op.  `pad` is symmetric explicit zero padding (VALID on the padded
// map, like Keras ZeroPadding2D + Conv2D); same_pad_even = TensorFlow 'SAME' for an even
// kernel at stride 1 (pad 0 before, k-1 after).  Returns the y tensor (or -1 if !want_y).
int Net::conv(const std::string& name, int x, int KH, int KW, int stride, int pad, int Cout, bool bias,
              const BNRef& bn, int act, int alpha, int res, int res_stride, bool want_y, const BNRef& bn2,
              int act2, int* y2_out, const std::string& wsuffix, bool same_pad_even, int pad_br) {
  const TensorDesc xd = tensors[x];
  Op op;
  op.kind = OP_CONV;
  op.name = name;
  op.x = x;
  op.KH = KH;
  op.KW = KW;
  op.stride = stride;
  op.pad_t = op.pad_l = pad;
  op.Cin = xd.C;
  op.Cin_true = (x == input_tensor) ? 3 : xd.C;
  op.Cout = Cout;
  int Ho, Wo;
  if (same_pad_even) {
    Ho = xd.H;
    Wo = xd.W;
  } else {
    const int pb = pad_br < 0 ? pad : pad_br;      // bottom/right padding (darknet pads top/left only at stride 2)
    Ho = (xd.H + pad + pb - KH) / stride + 1;
    Wo = (xd.W + pad + pb - KW) / stride + 1;
  }
  // (dense layers re-register their kernel with its own 2-D [in, out] shape after this call)
  op.w = P(name + wsuffix, {KH, KW, op.Cin_true, Cout});
  if (bias) op.bias = P(name + "/bias", {Cout});
  op.bn = bn;
  op.act = act;
  op.alpha = alpha;
  op.res = res;
  op.res_stride = res_stride;
  op.bn2 = bn2;
  op.act2 = act2;
  op.macs = (double)Ho * Wo * KH * KW * op.Cin_true * Cout;
  op.y = want_y ? out_tensor(Ho, Wo, Cout) : -1;
  if (y2_out) {
    op.y2 = T(Ho, Wo, Cout);
    *y2_out = op.y2;
  }
  ops.push_back(op);
  return op.y;
}

// ----------------------------------------------------------------------------- ResNet50V2 + heads
int Net::build_resnet50v2() {
  const BNRef none;
  input_tensor = T(in_h, in_w, 4);
  {
    Op in;
    in.kind = OP_INPUT;
    in.name = "input";
    in.y = input_tensor;
    ops.push_back(in);
  }
  // conv1_pad(3) + conv1_conv 7x7/2 (bias); preact network: no BN/ReLU here
  int t = conv("conv1_conv", input_tensor, 7, 7, 2, 3, 64, true, none, ACT_NONE, -1, -1, 1, true, none, ACT_NONE,
               nullptr);
  // pool1_pad(1) + MaxPool 3x3/2
  static const int stacks[4][3] = {{64, 3, 2}, {128, 4, 2}, {256, 6, 2}, {512, 3, 1}};
  int x, pre = -1;
  {
    const TensorDesc td = tensors[t];
    Op p;
    p.kind = OP_MAXPOOL;
    p.name = "pool1_pool";
    p.x = t;
    p.KH = p.KW = 3;
    p.stride = 2;
    p.pad_t = p.pad_l = 1;
    p.zero_pad = 1;
    p.Cin = p.Cout = td.C;
    const int Ho = (td.H + 2 - 3) / 2 + 1, Wo = (td.W + 2 - 3) / 2 + 1;
    p.y = T(Ho, Wo, td.C);
    x = p.y;
    ops.push_back(p);
  }
  // Pre-activation blocks.  relu(preact_bn(x)) feeds only 1x1 convolutions (the block's
  // _1_conv and, on block 1, its _0_conv shortcut), so it is never written to memory: those
  // convolutions apply it to x while gathering their A operand.  Only the block output x
  // itself (needed un-activated by the next identity shortcut) is stored.
  int cin = 64;
  for (int s = 0; s < 4; ++s) {
    const int f = stacks[s][0], nb = stacks[s][1], stride1 = stacks[s][2];
    for (int b = 1; b <= nb; ++b) {
      char nm[64];
      snprintf(nm, sizeof(nm), "conv%d_block%d", s + 2, b);
      const std::string n(nm);
      const bool conv_shortcut = (b == 1);
      const int stride = (b == nb) ? stride1 : 1;
      const BNRef pbn = BN(n + "_preact_bn", cin, EPS_RESNET);
      int sc = -1;
      if (conv_shortcut) {
        sc = conv(n + "_0_conv", x, 1, 1, stride, 0, 4 * f, true, none, ACT_NONE, -1, -1, 1, true, none, ACT_NONE,
                  nullptr);
        ops.back().pre_bn = pbn;
        ops.back().pre_act = ACT_RELU;
      }
      const BNRef bn1 = BN(n + "_1_bn", f, EPS_RESNET);
      int y1 = conv(n + "_1_conv", x, 1, 1, 1, 0, f, false, bn1, ACT_RELU, -1, -1, 1, true, none, ACT_NONE, nullptr);
      ops.back().pre_bn = pbn;
      ops.back().pre_act = ACT_RELU;
      const BNRef bn2 = BN(n + "_2_bn", f, EPS_RESNET);
      int y2 = conv(n + "_2_conv", y1, 3, 3, stride, 1, f, false, bn2, ACT_RELU, -1, -1, 1, true, none, ACT_NONE,
                    nullptr);
      const bool last = (s == 3 && b == nb);
      const int res = conv_shortcut ? sc : x;
      const int res_stride = conv_shortcut ? 1 : stride;   // MaxPooling2D(1, strides=stride) on the shortcut
      if (last) {
        // the network output relu(post_bn(.)) is a real tensor: second output of the last conv
        const BNRef post = BN("post_bn", 4 * f, EPS_RESNET);
        conv(n + "_3_conv", y2, 1, 1, 1, 0, 4 * f, true, none, ACT_NONE, -1, res, res_stride, false, post, ACT_RELU,
             &pre);
      } else {
        x = conv(n + "_3_conv", y2, 1, 1, 1, 0, 4 * f, true, none, ACT_NONE, -1, res, res_stride, true, none,
                 ACT_NONE, nullptr);
      }
      cin = 4 * f;
    }
  }
  (void)cin;
  return build_heads(pre);   // pre = relu(post_bn(.)): [H/32, W/32, 2048]
}

// The embedding heads of bottleneck_network (triplet.py:102-141) on any backbone's feature map.
int Net::build_heads(int feat) {
  const BNRef none;
  const TensorDesc fd = tensors[feat];
  if (head == "v3") {
    output_tensor = feat;
    return 0;
  }
  if (head == "v2") {
    // triplet.py:126-128  Conv2D(512, 1, use_bias=False) -> BN -> PReLU(shared_axes=[1,2])
    const BNRef hb1 = BN("head_bn1", 512, EPS_KERAS);
    const int al = P("head_prelu/alpha", {512});
    int h = conv("head_conv", feat, 1, 1, 1, 0, 512, false, hb1, ACT_PRELU, al, -1, 1, true, none, ACT_NONE,
                 nullptr);
    // triplet.py:129-130  DepthwiseConv2D(kernel = map size) -> BN
    if (fd.H != fd.W) return set_error("GDC head needs a square feature map (got %dx%d)", fd.H, fd.W);
    if (emd <= 1024 && emd % 8 == 0) {
      // triplet.py:129-138 in ONE launch: depthwise over the whole map + BN -> Conv1x1(emd) ->
      // [Dropout = identity] -> Flatten -> Dense(emd) -> l2_normalize.  0.5 MMAC per image: latency, not MFMA work.
      Op g;
      g.kind = OP_GDCTAIL;
      g.name = "head_tail";
      g.x = h;
      g.KH = fd.H;
      g.KW = fd.W;
      g.Cin = 512;
      g.Cout = emd;
      g.w = P("head_dw/depthwise_kernel", {fd.H, fd.W, 512, 1});
      g.bn = BN("head_bn2", 512, EPS_KERAS);
      g.w_pw = P("head_pw/kernel", {1, 1, 512, emd});
      g.w_dense = P("head_dense/kernel", {emd, emd});
      g.y = T(1, 1, emd);
      g.macs = (double)fd.H * fd.W * 512 + 512.0 * emd + (double)emd * emd;
      ops.push_back(g);
      output_tensor = g.y;
      return 0;
    }
    Op d;
    d.kind = OP_DWFULL;
    d.name = "head_dw";
    d.x = h;
    d.KH = fd.H;
    d.KW = fd.W;
    d.Cin = d.Cout = 512;
    d.w = P("head_dw/depthwise_kernel", {fd.H, fd.W, 512, 1});
    d.bn = BN("head_bn2", 512, EPS_KERAS);
    d.y = T(1, 1, 512);
    d.macs = (double)fd.H * fd.W * 512;
    ops.push_back(d);
    // triplet.py:131  Conv2D(emd, 1, use_bias=False); :133-134 Dropout = identity at inference
    int e = conv("head_pw", d.y, 1, 1, 1, 0, emd, false, none, ACT_NONE, -1, -1, 1, true, none, ACT_NONE, nullptr);
    // triplet.py:135-136  Flatten -> Dense(emd, use_bias=False)
    int q = conv("head_dense", e, 1, 1, 1, 0, emd, false, none, ACT_NONE, -1, -1, 1, true, none, ACT_NONE, nullptr);
    params[ops.back().w].shape = {emd, emd};
    // triplet.py:138  l2_normalize(axis=1)
    Op l;
    l.kind = OP_L2NORM;
    l.name = "norm_embedding";
    l.x = q;
    l.Cin = l.Cout = emd;
    l.y = T(1, 1, emd);
    ops.push_back(l);
    output_tensor = l.y;
    return 0;
  }
  if (head == "v1") {
    // triplet.py:105-106  Conv2D(64, 2, 'same', relu) -> MaxPooling2D(2)
    int a = conv("v1_conv1", feat, 2, 2, 1, 0, 64, true, none, ACT_RELU, -1, -1, 1, true, none, ACT_NONE, nullptr,
                 "/kernel", true);
    auto pool2 = [&](int src, const char* nm) {
      const TensorDesc td = tensors[src];
      Op p;
      p.kind = OP_MAXPOOL;
      p.name = nm;
      p.x = src;
      p.KH = p.KW = 2;
      p.stride = 2;
      p.Cin = p.Cout = td.C;
      p.y = T(td.H / 2, td.W / 2, td.C);
      ops.push_back(p);
      return p.y;
    };
    int b = pool2(a, "v1_pool1");
    if (tensors[b].H < 1) return set_error("v1 head: feature map too small");
    // triplet.py:108-109
    int c = conv("v1_conv2", b, 2, 2, 1, 0, 32, true, none, ACT_RELU, -1, -1, 1, true, none, ACT_NONE, nullptr,
                 "/kernel", true);
    int d = pool2(c, "v1_pool2");
    const TensorDesc dd = tensors[d];
    if (dd.H < 1 || dd.W < 1)
      return set_error("v1 head: the second 2x2 pool leaves an empty map (input %dx%d)", in_h, in_w);
    // triplet.py:111-112  Flatten -> Dense(emd): a VALID conv whose kernel is the whole map
    int e = conv("embeddings", d, dd.H, dd.W, 1, 0, emd, true, none, ACT_NONE, -1, -1, 1, true, none, ACT_NONE,
                 nullptr);
    params[ops.back().w].shape = {(int64_t)dd.H * dd.W * dd.C, emd};
    output_tensor = e;
    return 0;
  }
  if (head == "sv2") {
    // networks/siamese.py:107-128 (the siamese builder's v2): Conv1x1(128, bias, relu) -> MaxPooling2D('same') ->
    // Conv1x1(128, bias, relu) -> MaxPooling2D('same') -> BatchNormalization("bn") -> Flatten -> Dropout ->
    // Dense(emd, relu, "norm_embedding").  The BN sits between a pool and the dense layer: it is applied to
    // the dense layer's input while that is gathered (the pre-activation path, without an activation).
    auto pool_same2 = [&](int src, const char* nm) {
      const TensorDesc td = tensors[src];
      Op p;
      p.kind = OP_MAXPOOL;
      p.name = nm;
      p.x = src;
      p.KH = p.KW = 2;
      p.stride = 2;
      p.Cin = p.Cout = td.C;
      p.y = T((td.H + 1) / 2, (td.W + 1) / 2, td.C);     // 'same': ceil; the extra tap is outside the map, ignored by max
      ops.push_back(p);
      return p.y;
    };
    int a = conv("sv2_conv1", feat, 1, 1, 1, 0, 128, true, none, ACT_RELU, -1, -1, 1, true, none, ACT_NONE, nullptr);
    a = pool_same2(a, "sv2_pool1");
    int b = conv("sv2_conv2", a, 1, 1, 1, 0, 128, true, none, ACT_RELU, -1, -1, 1, true, none, ACT_NONE, nullptr);
    b = pool_same2(b, "sv2_pool2");
    const TensorDesc bd = tensors[b];
    const BNRef bn = BN("bn", 128, EPS_KERAS);
    int e = conv("norm_embedding", b, bd.H, bd.W, 1, 0, emd, true, none, ACT_RELU, -1, -1, 1, true, none, ACT_NONE,
                 nullptr);
    ops.back().pre_bn = bn;
    ops.back().pre_act = ACT_NONE;
    params[ops.back().w].shape = {(int64_t)bd.H * bd.W * bd.C, emd};
    output_tensor = e;
    return 0;
  }
  return set_error("unknown head '%s' (v1, v2, v3, sv2)", head.c_str());
}

// ----------------------------------------------------------------------------- VGG16 / MobileNetV2
// The other two backbones bottleneck_network accepts (triplet.py:77,87-93): keras.applications
// VGG16 and MobileNetV2 (alpha 1.0), include_top=False.  Third-party definitions, restated from their
// public layer tables like ResNet50V2; same parameter names as Keras.
void Net::add_input() {
  input_tensor = T(in_h, in_w, 4);
  Op in;
  in.kind = OP_INPUT;
  in.name = "input";
  in.y = input_tensor;
  ops.push_back(in);
}

int Net::build_vgg16() {
  const BNRef none;
  add_input();
  static const int cfg[5][2] = {{64, 2}, {128, 2}, {256, 3}, {512, 3}, {512, 3}};
  int x = input_tensor;
  for (int b = 0; b < 5; ++b) {
    for (int c = 1; c <= cfg[b][1]; ++c) {
      char nm[32];
      snprintf(nm, sizeof(nm), "block%d_conv%d", b + 1, c);
      x = conv(nm, x, 3, 3, 1, 1, cfg[b][0], true, none, ACT_RELU, -1, -1, 1, true, none, ACT_NONE, nullptr);
    }
    char nm[32];
    snprintf(nm, sizeof(nm), "block%d_pool", b + 1);
    x = pool(nm, x, 2, 2, 0, POOL_MAX, 0);
    if (tensors[x].H < 1 || tensors[x].W < 1) return set_error("vgg16: input %dx%d is too small", in_h, in_w);
  }
  return build_heads(x);
}

int Net::build_mobilenetv2() {
  const BNRef none;
  add_input();
  const float eps = 1e-3f;    // keras MobileNetV2: BatchNormalization(epsilon=1e-3, momentum=0.999)
  // keras correct_pad for a 3x3 stride-2 layer: ((1 - H%2 ... )) -> even size: 0 before / 1 after; odd: 1 / 1
  auto pad_tl = [&](int size) { return size % 2 == 0 ? 0 : 1; };
  auto dw = [&](const std::string& name, int x, int stride) {
    const TensorDesc xd = tensors[x];
    Op d;
    d.kind = OP_DWCONV;
    d.name = name;
    d.x = x;
    d.KH = d.KW = 3;
    d.stride = stride;
    d.Cin = d.Cout = xd.C;
    int Ho, Wo;
    if (stride == 1) {
      d.pad_t = d.pad_l = 1;      // 'same'
      Ho = xd.H;
      Wo = xd.W;
    } else {
      d.pad_t = pad_tl(xd.H);     // ZeroPadding2D(correct_pad) + 'valid'
      d.pad_l = pad_tl(xd.W);
      Ho = (xd.H + d.pad_t + 1 - 3) / 2 + 1;
      Wo = (xd.W + d.pad_l + 1 - 3) / 2 + 1;
    }
    d.w = P(name + "/depthwise_kernel", {3, 3, xd.C, 1});
    d.bn = BN(name + "_BN", xd.C, eps);
    d.act = ACT_RELU6;
    d.y = T(Ho, Wo, xd.C);
    d.macs = (double)Ho * Wo * 9 * xd.C;
    ops.push_back(d);
    return d.y;
  };
  // Conv1: ZeroPadding2D(correct_pad) + Conv2D(32, 3, strides 2, valid, no bias) + BN + ReLU6
  int x = conv("Conv1", input_tensor, 3, 3, 2, pad_tl(in_h), 32, false, BN("bn_Conv1", 32, eps), ACT_RELU6, -1, -1, 1,
               true, none, ACT_NONE, nullptr, "/kernel", false, 1);
  if (pad_tl(in_h) != pad_tl(in_w)) return set_error("mobilenet: height and width must have the same parity");
  static const int cfg[7][4] = {{1, 16, 1, 1}, {6, 24, 2, 2}, {6, 32, 3, 2}, {6, 64, 4, 2}, {6, 96, 3, 1}, {6, 160, 3, 2},
                                {6, 320, 1, 1}};
  int block_id = 0, cin = 32;
  for (int g = 0; g < 7; ++g) {
    const int t = cfg[g][0], cout = cfg[g][1];
    for (int i = 0; i < cfg[g][2]; ++i, ++block_id) {
      const int stride = i == 0 ? cfg[g][3] : 1;
      const std::string pre = block_id == 0 ? "expanded_conv" : "block_" + std::to_string(block_id);
      int h = x;
      if (block_id != 0)
        h = conv(pre + "_expand", x, 1, 1, 1, 0, t * cin, false, BN(pre + "_expand_BN", t * cin, eps), ACT_RELU6, -1, -1,
                 1, true, none, ACT_NONE, nullptr);
      h = dw(pre + "_depthwise", h, stride);
      const int res = (cin == cout && stride == 1) ? x : -1;
      x = conv(pre + "_project", h, 1, 1, 1, 0, cout, false, BN(pre + "_project_BN", cout, eps), ACT_NONE, -1, res, 1, true,
               none, ACT_NONE, nullptr);
      cin = cout;
    }
  }
  x = conv("Conv_1", x, 1, 1, 1, 0, 1280, false, BN("Conv_1_bn", 1280, eps), ACT_RELU6, -1, -1, 1, true, none, ACT_NONE,
           nullptr);
  return build_heads(x);
}

// ----------------------------------------------------------------------------- IResNet
int Net::build_iresnet(const int* layers) {
  const BNRef none;
  if (in_h != in_w || in_h % 16 != 0) return set_error("iresnet needs a square input divisible by 16");
  input_tensor = T(in_h, in_w, 4);
  {
    Op in;
    in.kind = OP_INPUT;
    in.name = "input";
    in.y = input_tensor;
    ops.push_back(in);
  }
  static const int widths[4] = {64, 128, 256, 512};
  const BNRef bn1 = BN("bn1", 64, EPS_IRESNET);
  const int al0 = P("prelu/alpha", {64});
  BNRef next_bn = BN("layer1_0_bn1", 64, EPS_IRESNET);
  int xb = -1;
  int x = conv("conv1", input_tensor, 3, 3, 1, 1, 64, false, bn1, ACT_PRELU, al0, -1, 1, true, next_bn, ACT_NONE,
               &xb);
  for (int li = 0; li < 4; ++li) {
    const int cout = widths[li];
    for (int b = 0; b < layers[li]; ++b) {
      char nm[64];
      snprintf(nm, sizeof(nm), "layer%d_%d", li + 1, b);
      const std::string n(nm);
      const int stride = (b == 0) ? 2 : 1;
      const BNRef bn2 = BN(n + "_bn2", cout, EPS_IRESNET);
      const int al = P(n + "_prelu/alpha", {cout});
      int t1 = conv(n + "_conv1", xb, 3, 3, 1, 1, cout, false, bn2, ACT_PRELU, al, -1, 1, true, none, ACT_NONE,
                    nullptr);
      int res = x;
      if (b == 0) {
        const BNRef dbn = BN(n + "_down_bn", cout, EPS_IRESNET);
        res = conv(n + "_down_conv", x, 1, 1, stride, 0, cout, false, dbn, ACT_NONE, -1, -1, 1, true, none,
                   ACT_NONE, nullptr);
      }
      const BNRef bn3 = BN(n + "_bn3", cout, EPS_IRESNET);
      const bool last = (li == 3 && b == layers[li] - 1);
      std::string nxt;
      int nc = cout;
      if (last)
        nxt = "bn2";
      else if (b == layers[li] - 1)
        snprintf(nm, sizeof(nm), "layer%d_0_bn1", li + 2), nxt = nm;
      else
        snprintf(nm, sizeof(nm), "layer%d_%d_bn1", li + 1, b + 1), nxt = nm;
      next_bn = BN(nxt, nc, EPS_IRESNET);
      int xbn = -1;
      int xn = conv(n + "_conv2", t1, 3, 3, stride, 1, cout, false, bn3, ACT_NONE, -1, res, 1, !last, next_bn,
                    ACT_NONE, &xbn);
      x = xn;
      xb = xbn;
    }
  }
  const TensorDesc fd = tensors[xb];   // bn2 output, [H/16, W/16, 512]
  const BNRef feat = BN("features", emd, EPS_IRESNET);
  int e = conv("fc", xb, fd.H, fd.W, 1, 0, emd, true, feat, ACT_NONE, -1, -1, 1, true, none, ACT_NONE, nullptr);
  ops.back().chw_flatten = true;       // the original flattens NCHW: rows are c*HW + h*W + w
  params[ops.back().w].shape = {(int64_t)fd.H * fd.W * fd.C, emd};
  Op l;
  l.kind = OP_L2NORM;
  l.name = "norm_embedding";
  l.x = e;
  l.Cin = l.Cout = emd;
  l.y = T(1, 1, emd);
  ops.push_back(l);
  output_tensor = l.y;
  return 0;
}

// ----------------------------------------------------------------------------- NN4.small2 (OpenFace)
// deep_insight_face/networks/inceptionv3.py:93-309 (InceptionNetwork._load_model) and :312-335
// (conv2d_bn).  Every Conv2D has a bias and is followed by BatchNormalization(epsilon=1e-5) and
// ReLU; the inception branches are concatenated on the channel axis, which here means each
// branch's last layer writes straight into its channel slice of the block's output tensor
// (and, where the reference zero-pads a branch AFTER its last layer, into the interior of a
// zero-filled map).
int Net::build_nn4() {
  const BNRef none;
  if (in_h != 96 || in_w != 96)
    return set_error("Invalid Input shape, Shape should be of dimension (96, 96, 3)");   // inceptionv3.py:66
  const float eps = 1e-5f;
  input_tensor = T(in_h, in_w, 4);
  {
    Op in;
    in.kind = OP_INPUT;
    in.name = "input";
    in.y = input_tensor;
    ops.push_back(in);
  }
  auto cbr = [&](const std::string& cname, const std::string& bname, int x, int k, int stride, int pad, int cout) {
    const BNRef bn = BN(bname, cout, eps);
    return conv(cname, x, k, k, stride, pad, cout, true, bn, ACT_RELU, -1, -1, 1, true, none, ACT_NONE, nullptr);
  };
  auto lrn = [&](const std::string& name, int x) {
    const TensorDesc td = tensors[x];
    Op l;
    l.kind = OP_LRN;
    l.name = name;
    l.x = x;
    l.Cin = l.Cout = td.C;
    l.y = T(td.H, td.W, td.C);
    ops.push_back(l);
    return l.y;
  };
  auto zero = [&](int t) {
    Op z;
    z.kind = OP_ZERO;
    z.name = "zero_fill";
    z.y = t;
    ops.push_back(z);
  };
  // two-conv branch of conv2d_bn(): 1x1 -> pad -> kxk (stride s), written into a concat slice
  auto branch2 = [&](const std::string& layer, int x, int c1, int c2, int k, int stride, int pad, int cat, int coff) {
    const int t = cbr(layer + "_conv1", layer + "_bn1", x, 1, 1, 0, c1);
    into(cat, coff);
    return cbr(layer + "_conv2", layer + "_bn2", t, k, stride, pad, c2);
  };

  // stem: inceptionv3.py:96-112
  int x = cbr("conv1", "bn1", input_tensor, 7, 2, 3, 64);                 // 48x48x64
  x = pool("pool1", x, 3, 2, 1, POOL_MAX, 1);                             // 24x24x64
  x = lrn("lrn_1", x);
  x = cbr("conv2", "bn2", x, 1, 1, 0, 64);
  x = cbr("conv3", "bn3", x, 3, 1, 1, 192);
  x = lrn("lrn_2", x);
  x = pool("pool2", x, 3, 2, 1, POOL_MAX, 1);                             // 12x12x192

  // inception 3a: :114-141   concat [3x3 128, 5x5 32, pool 32, 1x1 64] = 256
  {
    const int cat = T(12, 12, 256);
    zero(cat);
    branch2("inception_3a_3x3", x, 96, 128, 3, 1, 1, cat, 0);
    branch2("inception_3a_5x5", x, 16, 32, 5, 1, 2, cat, 128);
    const int p = pool("inception_3a_pool", x, 3, 2, 0, POOL_MAX, 0);     // 5x5x192
    into(cat, 160, 3, 3);                                                 // ZeroPadding2D(((3,4),(3,4)))
    cbr("inception_3a_pool_conv", "inception_3a_pool_bn", p, 1, 1, 0, 32);
    into(cat, 192);
    cbr("inception_3a_1x1_conv", "inception_3a_1x1_bn", x, 1, 1, 0, 64);
    x = cat;
  }
  // inception 3b: :143-174   concat [128, 64, pool 64, 64] = 320
  {
    const int cat = T(12, 12, 320);
    zero(cat);
    branch2("inception_3b_3x3", x, 96, 128, 3, 1, 1, cat, 0);
    branch2("inception_3b_5x5", x, 32, 64, 5, 1, 2, cat, 128);
    const int p = pool("inception_3b_pool", x, 3, 3, 0, POOL_L2, 0);      // 4x4x256
    into(cat, 192, 4, 4);                                                 // ZeroPadding2D((4,4))
    cbr("inception_3b_pool_conv", "inception_3b_pool_bn", p, 1, 1, 0, 64);
    into(cat, 256);
    cbr("inception_3b_1x1_conv", "inception_3b_1x1_bn", x, 1, 1, 0, 64);
    x = cat;
  }
  // inception 3c: :176-199   concat [256, 64, pool 320] = 640 at 6x6
  {
    const int cat = T(6, 6, 640);
    zero(cat);
    branch2("inception_3c_3x3", x, 128, 256, 3, 2, 1, cat, 0);
    branch2("inception_3c_5x5", x, 32, 64, 5, 2, 2, cat, 256);
    into(cat, 320, 0, 0);                                                 // 5x5 map, ZeroPadding2D(((0,1),(0,1)))
    pool("inception_3c_pool", x, 3, 2, 0, POOL_MAX, 0);
    x = cat;
  }
  // inception 4a: :201-232   concat [192, 64, pool 128, 256] = 640
  {
    const int cat = T(6, 6, 640);
    zero(cat);
    branch2("inception_4a_3x3", x, 96, 192, 3, 1, 1, cat, 0);
    branch2("inception_4a_5x5", x, 32, 64, 5, 1, 2, cat, 192);
    const int p = pool("inception_4a_pool", x, 3, 3, 0, POOL_L2, 0);      // 2x2x640
    into(cat, 256, 2, 2);                                                 // padding=(2,2)
    cbr("inception_4a_pool_conv", "inception_4a_pool_bn", p, 1, 1, 0, 128);
    into(cat, 384);
    cbr("inception_4a_1x1_conv", "inception_4a_1x1_bn", x, 1, 1, 0, 256);
    x = cat;
  }
  // inception 4e: :234-255   concat [256, 128, pool 640] = 1024 at 3x3
  {
    const int cat = T(3, 3, 1024);
    zero(cat);
    branch2("inception_4e_3x3", x, 160, 256, 3, 2, 1, cat, 0);
    branch2("inception_4e_5x5", x, 64, 128, 5, 2, 2, cat, 256);
    into(cat, 384, 0, 0);                                                 // 2x2 map, ZeroPadding2D(((0,1),(0,1)))
    pool("inception_4e_pool", x, 3, 2, 0, POOL_MAX, 0);
    x = cat;
  }
  // inception 5a: :257-283   concat [384, pool 96, 256] = 736
  {
    const int cat = T(3, 3, 736);
    zero(cat);
    branch2("inception_5a_3x3", x, 96, 384, 3, 1, 1, cat, 0);
    const int p = pool("inception_5a_pool", x, 3, 3, 0, POOL_L2, 0);      // 1x1x1024
    into(cat, 384, 1, 1);                                                 // padding=(1,1)
    cbr("inception_5a_pool_conv", "inception_5a_pool_bn", p, 1, 1, 0, 96);
    into(cat, 480);
    cbr("inception_5a_1x1_conv", "inception_5a_1x1_bn", x, 1, 1, 0, 256);
    x = cat;
  }
  // inception 5b: :285-306   concat [384, pool 96, 256] = 736
  {
    const int cat = T(3, 3, 736);
    zero(cat);
    branch2("inception_5b_3x3", x, 96, 384, 3, 1, 1, cat, 0);
    const int p = pool("inception_5b_pool", x, 3, 2, 0, POOL_MAX, 0);     // 1x1x736
    into(cat, 384, 1, 1);                                                 // ZeroPadding2D((1,1))
    cbr("inception_5b_pool_conv", "inception_5b_pool_bn", p, 1, 1, 0, 96);
    into(cat, 480);
    cbr("inception_5b_1x1_conv", "inception_5b_1x1_bn", x, 1, 1, 0, 256);
    x = cat;
  }
  // :308-311  AveragePooling2D(3, strides 1) -> Flatten -> Dense(emd) -> l2_normalize
  x = pool("avg_pool", x, 3, 1, 0, POOL_AVG, 0);                          // 1x1x736
  int d = conv("dense_layer", x, 1, 1, 1, 0, emd, true, none, ACT_NONE, -1, -1, 1, true, none, ACT_NONE, nullptr);
  params[ops.back().w].shape = {736, emd};
  Op l;
  l.kind = OP_L2NORM;
  l.name = "norm_layer";
  l.x = d;
  l.Cin = l.Cout = emd;
  l.y = T(1, 1, emd);
  ops.push_back(l);
  output_tensor = l.y;
  return 0;
}

// ----------------------------------------------------------------------------- YOLOv3-face detector
// The network the reference loads as a converted Keras model (detector/run.py:140; converter
// scripts/yolo_convert_tf.py:60-215 over detector/yolo_cfg/yolov3-face.cfg): Darknet-53
// (residual stages of 1, 2, 8, 8, 4 blocks) + three detection heads, one class -> 18 = 3*(5+1)
// channels per head.  Layer semantics as the converter builds them: Conv2D (stride 1: 'same';
// stride 2: ZeroPadding2D(((1,0),(1,0))) + 'valid') without bias -> BatchNormalization (Keras
// default epsilon 1e-3) -> LeakyReLU(0.1); the three head outputs are linear convolutions with
// bias; shortcut = Add; route = pass-through or channel concatenation; upsample = nearest x2.
// Parameters are named conv_<i> / bn_<i> with i = the convolution's ordinal in the cfg (0..74).
int Net::build_yolov3() {
  const BNRef none;
  if (in_h % 32 != 0 || in_w % 32 != 0) return set_error("yolov3 input must be a multiple of 32 (got %dx%d)", in_h, in_w);
  const int n_out = 3 * (5 + (emd > 0 ? emd : 1));   // emd_size carries the class count for this arch
  input_tensor = T(in_h, in_w, 4);
  {
    Op in;
    in.kind = OP_INPUT;
    in.name = "input";
    in.y = input_tensor;
    ops.push_back(in);
  }
  int ci = 0;
  // conv + BN + leaky; res >= 0 adds a shortcut after the activation (Add([from, conv]))
  auto cbl = [&](int x, int cout, int k, int stride, int res) {
    char nm[32], bm[32];
    snprintf(nm, sizeof(nm), "conv_%d", ci);
    snprintf(bm, sizeof(bm), "bn_%d", ci);
    ++ci;
    const BNRef bn = BN(bm, cout, EPS_KERAS);
    const int pad = (k == 3) ? 1 : 0;
    const int y = conv(nm, x, k, k, stride, pad, cout, false, bn, ACT_PRELU, -1, res, 1, true, none, ACT_NONE, nullptr,
                       "/kernel", false, stride == 2 ? 0 : -1);
    ops.back().const_alpha = 0.1f;
    return y;
  };
  auto linear = [&](int x, int cout) {
    char nm[32];
    snprintf(nm, sizeof(nm), "conv_%d", ci);
    ++ci;
    return conv(nm, x, 1, 1, 1, 0, cout, true, none, ACT_NONE, -1, -1, 1, true, none, ACT_NONE, nullptr);
  };
  auto stage = [&](int x, int cout, int blocks) {
    x = cbl(x, cout, 3, 2, -1);                       // downsample
    for (int b = 0; b < blocks; ++b) {
      const int t = cbl(x, cout / 2, 1, 1, -1);
      x = cbl(t, cout, 3, 1, x);                      // + shortcut
    }
    return x;
  };
  int x = cbl(input_tensor, 32, 3, 1, -1);
  x = stage(x, 64, 1);
  x = stage(x, 128, 2);
  const int r36 = x = stage(x, 256, 8);               // cfg layer 36
  const int r61 = x = stage(x, 512, 8);               // cfg layer 61
  x = stage(x, 1024, 4);
  // head: five alternating 1x1 / 3x3 convs, then 3x3 + linear 1x1 detection conv
  auto head = [&](int x, int c, int* branch) {
    x = cbl(x, c, 1, 1, -1);
    x = cbl(x, 2 * c, 3, 1, -1);
    x = cbl(x, c, 1, 1, -1);
    x = cbl(x, 2 * c, 3, 1, -1);
    x = cbl(x, c, 1, 1, -1);
    *branch = x;                                      // route -4
    x = cbl(x, 2 * c, 3, 1, -1);
    return linear(x, n_out);
  };
  auto up_concat = [&](int x, int c, int skip) {
    x = cbl(x, c, 1, 1, -1);
    const TensorDesc xd = tensors[x];
    const TensorDesc sd = tensors[skip];
    const int cat = T(2 * xd.H, 2 * xd.W, xd.C + sd.C);
    Op u;
    u.kind = OP_UPSAMPLE;
    u.name = "upsample";
    u.x = x;
    u.Cin = u.Cout = xd.C;
    u.y = V(cat, 2 * xd.H, 2 * xd.W, xd.C, 0, 0, 0);
    ops.push_back(u);
    Op c2;
    c2.kind = OP_COPY;
    c2.name = "route_concat";
    c2.x = skip;
    c2.Cin = c2.Cout = sd.C;
    c2.y = V(cat, sd.H, sd.W, sd.C, xd.C, 0, 0);
    ops.push_back(c2);
    return cat;
  };
  int br = -1;
  const int y13 = head(x, 512, &br);
  x = up_concat(br, 256, r61);
  const int y26 = head(x, 256, &br);
  x = up_concat(br, 128, r36);
  const int y52 = head(x, 128, &br);
  output_tensor = y13;
  extra_outputs = {y26, y52};
  return 0;
}

// ----------------------------------------------------------------------------- MTCNN (BASELINE configs[4] as worded)
// NOT IN THE REFERENCE (config.py:37 and detector/run.py:124 only mention it in comments; the detector it ships is
// YOLOv3-face, above).  Layer tables of the public MTCNN (Zhang et al. 2016, "Joint Face Detection and Alignment using
// Multi-task Cascaded Convolutional Networks"), restated from the published definition: every convolution VALID with bias
// and PReLU, Caffe max-pooling (ceil mode).
//   P-Net: conv3x3(10) - pool2/2 - conv3x3(16) - conv3x3(32) - {conv1x1(2) face / not-face logits, conv1x1(4) box regression}
//   R-Net (24 x 24): conv3x3(28) - pool3/2 - conv3x3(48) - pool3/2 - conv2x2(64) - fc(128) - {fc(2), fc(4)}
//   O-Net (48 x 48): conv3x3(32) - pool3/2 - conv3x3(64) - pool3/2 - conv3x3(64) - pool2/2 - conv2x2(128) - fc(256) - {fc(2), fc(4), fc(10)}
// Here: the sibling heads of a network are ONE layer whose filters are their concatenation (`head`: [logits 2 | box 4 |
// (landmarks 10) | zero filters up to a multiple of 4]) -- the same products, one launch; P-Net's 10 first-layer filters are
// held as 12 (two zero filters, whose outputs the next layer's zero weights ignore: channel counts stay multiples of 4); a
// fully connected layer over an h x w x c map is the h x w VALID convolution with kernel [h, w, c, out] (HWC flattening).
// The output is the head's map: [H', W', 8] for P-Net, [1, 1, 8] for R-Net, [1, 1, 16] for O-Net.
int Net::build_mtcnn(int stage) {
  const BNRef none;
  if (stage == 1 && (in_h < 12 || in_w < 12)) return set_error("mtcnn_pnet input must be at least 12x12 (got %dx%d)", in_h, in_w);
  if (stage == 2 && (in_h != 24 || in_w != 24)) return set_error("mtcnn_rnet input is 24x24");
  if (stage == 3 && (in_h != 48 || in_w != 48)) return set_error("mtcnn_onet input is 48x48");
  input_tensor = T(in_h, in_w, 4);
  {
    Op in;
    in.kind = OP_INPUT;
    in.name = "input";
    in.y = input_tensor;
    ops.push_back(in);
  }
  auto cp = [&](const std::string& nm, int x, int k, int cout, bool prelu) {
    const int al = prelu ? P(nm + "_prelu/alpha", {cout}) : -1;
    return conv(nm, x, k, k, 1, 0, cout, true, none, prelu ? ACT_PRELU : ACT_NONE, al, -1, 1, true, none, ACT_NONE, nullptr);
  };
  int x = input_tensor;
  if (stage == 1) {
    x = cp("conv1", x, 3, 12, true);
    x = pool("pool1", x, 2, 2, 0, POOL_MAX, 0, true);
    x = cp("conv2", x, 3, 16, true);
    x = cp("conv3", x, 3, 32, true);
    x = cp("head", x, 1, 8, false);
  } else if (stage == 2) {
    x = cp("conv1", x, 3, 28, true);
    x = pool("pool1", x, 3, 2, 0, POOL_MAX, 0, true);
    x = cp("conv2", x, 3, 48, true);
    x = pool("pool2", x, 3, 2, 0, POOL_MAX, 0, true);
    x = cp("conv3", x, 2, 64, true);
    x = cp("fc1", x, tensors[x].H, 128, true);
    x = cp("head", x, 1, 8, false);
  } else {
    x = cp("conv1", x, 3, 32, true);
    x = pool("pool1", x, 3, 2, 0, POOL_MAX, 0, true);
    x = cp("conv2", x, 3, 64, true);
    x = pool("pool2", x, 3, 2, 0, POOL_MAX, 0, true);
    x = cp("conv3", x, 3, 64, true);
    x = pool("pool3", x, 2, 2, 0, POOL_MAX, 0, true);
    x = cp("conv4", x, 2, 128, true);
    x = cp("fc1", x, tensors[x].H, 256, true);
    x = cp("head", x, 1, 16, false);
  }
  output_tensor = x;
  return 0;
}

int Net::build() {
  if (emd <= 0) return set_error("emd_size must be positive");
  if (arch == "mtcnn_pnet") return build_mtcnn(1);
  if (arch == "mtcnn_rnet") return build_mtcnn(2);
  if (arch == "mtcnn_onet") return build_mtcnn(3);
  if (in_h < 32 || in_w < 32) return set_error("input must be at least 32x32");
  if (arch == "resnet") return build_resnet50v2();
  if (arch == "iresnet50") {
    static const int l[4] = {3, 4, 14, 3};
    return build_iresnet(l);
  }
  if (arch == "nn4") return build_nn4();
  if (arch == "vgg16") return build_vgg16();
  if (arch == "mobilenet") return build_mobilenetv2();
  if (arch == "yolov3") return build_yolov3();
  if (arch == "iresnet100") {
    static const int l[4] = {3, 13, 30, 3};
    return build_iresnet(l);
  }
  return set_error("Invalid bottleneck network '%s' (supported: resnet, vgg16, mobilenet, iresnet50, iresnet100, nn4, "
                   "yolov3, mtcnn_pnet, mtcnn_rnet, mtcnn_onet)", arch.c_str());
}

double Net::flops_per_image() const {
  double m = 0;
  for (const Op& op : ops) m += op.macs;
  return 2.0 * m;
}

// ----------------------------------------------------------------------------- finalize
static int upload(Net* net, const std::vector<float>& host, float** out) {
  float* d = nullptr;
  // +4 floats: per-channel vectors are read 16 bytes at a time, also when Cout % 4 != 0
  DIF_HIP(hipMalloc(&d, (host.size() + 4) * sizeof(float)));
  DIF_HIP(hipMemset(d, 0, (host.size() + 4) * sizeof(float)));
  net->allocs.push_back(d);
  if (!host.empty()) DIF_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = d;
  return 0;
}

static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);   // inf / NaN: truncate
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// scale/shift of "(acc + bias) -> BN":  scale = gamma / sqrt(var + eps), shift = beta - mean*scale (+ bias*scale)
static void fold(const Net* net, const BNRef& bn, int bias, int C, std::vector<float>* scale,
                 std::vector<float>* shift) {
  scale->clear();
  shift->clear();
  if (bn.valid()) {
    const float* g = net->params[bn.gamma].data.data();
    const float* b = net->params[bn.beta].data.data();
    const float* m = net->params[bn.mean].data.data();
    const float* v = net->params[bn.var].data.data();
    scale->resize(C);
    shift->resize(C);
    for (int c = 0; c < C; ++c) {
      const float s = g[c] / sqrtf(v[c] + bn.eps);
      (*scale)[c] = s;
      (*shift)[c] = b[c] - m[c] * s;
    }
    if (bias >= 0) {
      const float* bb = net->params[bias].data.data();
      for (int c = 0; c < C; ++c) (*shift)[c] += bb[c] * (*scale)[c];
    }
  } else if (bias >= 0) {
    *shift = net->params[bias].data;
  }
}

// dif_net_set_option (include/dif.h documents the keys a caller may rely on; the rest are development switches that
// pick between kernel families the parity tests compare)
// every key set_option accepts (dif_net_option_name: include/dif.h documents each one, and a test holds it to that)
const char* const* Net::option_names() {
  static const char* const names[] = {"pipe", "bdp", "stem", "patch", "patch2d", "bd", "t2", "tn", "sk2", "mt", "bf16x3",
                                      "bf_terms", "ysub", "lane_split", "lane_prio", "dbg", nullptr};
  return names;
}

int Net::set_option(const char* key, int value) {
  {
    bool known = false;
    for (const char* const* t = option_names(); *t; ++t) known |= !strcmp(*t, key);
    if (!known) return set_error("dif_net_set_option: unknown option '%s'", key);
  }
  auto flag = [&](unsigned bit) {
    conv_off = value ? (conv_off & ~bit) : (conv_off | bit);
    return 0;
  };
  const bool pre = !finalized;
  if (!strcmp(key, "pipe")) return use_pipe = value != 0, 0;
  if (!strcmp(key, "bdp")) {
    if (value < 0 || value > 2) return set_error("dif_net_set_option: 'bdp' takes 0, 1 or 2");
    return use_bdp = value, 0;
  }
  if (!strcmp(key, "stem")) return use_stem = value != 0, 0;
  if (!strcmp(key, "dbg")) return conv_dbg = value, 0;
  if (!strcmp(key, "patch")) return flag(CONV_OFF_PATCH);
  if (!strcmp(key, "patch2d")) return flag(CONV_OFF_PATCH2D);
  if (!strcmp(key, "bd")) return flag(CONV_OFF_BD);
  if (!strcmp(key, "t2")) return flag(CONV_OFF_T2);
  if (!strcmp(key, "tn")) return flag(CONV_OFF_TN);
  if (!strcmp(key, "sk2")) return flag(CONV_OFF_SK2);
  if (!strcmp(key, "mt")) return flag(CONV_OFF_MT);
  if (!strcmp(key, "bf16x3")) {
    if (!pre && compute_bf16x3 != (value != 0))
      return set_error("dif_net_set_option: 'bf16x3' must be chosen before dif_net_finalize");
    return compute_bf16x3 = value != 0, 0;
  }
  if (!strcmp(key, "bf_terms")) {                          // 3 or 2 bf16 terms per operand in the split-bf16 mode (any time)
    if (value != 2 && value != 3) return set_error("dif_net_set_option: 'bf_terms' takes 2 or 3");
    return bf_terms = value, 0;
  }
  if (!strcmp(key, "lane_prio")) {
    if (!pre) return set_error("dif_net_set_option: 'lane_prio' must be chosen before dif_net_finalize");
    return opt_lane_prio = value, 0;
  }
  if (!strcmp(key, "ysub") || !strcmp(key, "lane_split")) {
    if (!pre) return set_error("dif_net_set_option: '%s' must be chosen before dif_net_finalize", key);
    if (key[0] == 'y') use_ysub = value != 0;
    else opt_lane_split = value < 0 ? -1 : (value != 0);
    return 0;
  }
  return set_error("dif_net_set_option: option '%s' is listed but not handled", key);   // option_names() and this chain out of step
}

int Net::finalize(int mb) {
  if (mb <= 0) return set_error("max_batch must be positive");
  // development hook: DIF_OPTIONS="key=value,key=value" applies dif_net_set_option keys to every net of the process
  // (A/B runs of bench.py and the tools without touching their code); unset in production
  if (const char* env = getenv("DIF_OPTIONS")) {
    std::string all(env);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find(',', pos);
      if (end == std::string::npos) end = all.size();
      const std::string kv = all.substr(pos, end - pos);
      const size_t eq = kv.find('=');
      if (eq == std::string::npos || eq == 0) return set_error("DIF_OPTIONS: expected key=value, got '%s'", kv.c_str());
      if (set_option(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1))) return -1;
      pos = end + 1;
    }
  }
  for (const Param& p : params)
    if (!p.set) return set_error("parameter '%s' was never set", p.name.c_str());
  release_device();
  max_batch = mb;

  // ---- weights and epilogue vectors
  for (Op& op : ops) {
    std::vector<float> scale, shift;
    if (op.kind == OP_CONV) {
      const int taps = op.KH * op.KW;
      const int K = taps * op.Cin;
      op.Kpad = (K + BK - 1) / BK * BK;
      op.k_order = (taps > 1 && op.Cin % BK == 0 && op.Cin_true == op.Cin) ? 1 : 0;
      std::vector<float> packed((size_t)op.Cout * op.Kpad, 0.f);
      const float* w = params[op.w].data.data();   // [taps][Cin_true][Cout] (HWIO) or CHW-flattened dense
      for (int t = 0; t < taps; ++t)
        for (int ci = 0; ci < op.Cin_true; ++ci) {
          const int64_t src_row = op.chw_flatten ? ((int64_t)ci * taps + t) : ((int64_t)t * op.Cin_true + ci);
          const float* src = w + src_row * op.Cout;
          const size_t kk = op.k_order == 1 ? ((size_t)(ci / BK) * taps + t) * BK + ci % BK : (size_t)t * op.Cin + ci;
          for (int co = 0; co < op.Cout; ++co) packed[(size_t)co * op.Kpad + kk] = src[co];
        }
      if (upload(this, packed, &op.d_w)) return -1;
      // 3-channel first layers leave the general kernel.  YOLOv3-face's 3x3 (32 filters) runs as a direct convolution
      // (elementwise.hip): 0.68 ms vs 1.50 ms per 64 frames as an implicit GEMM.  The 64-filter ones (IResNet conv1 3x3,
      // ResNet50V2 conv1_conv 7x7 / 2) run on the MFMA with the true K and the input patch in LDS (stem.hip).
      op.d_w_raw = nullptr;
      op.stem_mfma = false;
      const bool stem_shape = use_stem && op.Cin_true == 3 && op.Cin == 4 && op.res < 0 && !op.pre_bn.valid() &&
                              !op.chw_flatten && tensors[op.y >= 0 ? op.y : op.y2].parent < 0;
      if (stem_shape && op.KH == 3 && op.KW == 3 && op.stride == 1 && op.pad_t == 1 && op.pad_l == 1 && op.Cout == 32) {
        if (upload(this, params[op.w].data, &op.d_w_raw)) return -1;
      } else if (stem_shape && stem_mfma_applies(op.KH, op.KW, op.stride, op.pad_t, op.pad_l, op.Cout)) {
        if (upload(this, params[op.w].data, &op.d_w_raw)) return -1;
        op.stem_mfma = true;
      }
      op.d_w3f = nullptr;
      op.w3f_bytes = 0;
      if (compute_bf16x3 && op.k_order == 1 && op.KH == 3 && op.KW == 3 && op.stride == 1 && op.pad_t == 1 && op.pad_l == 1 &&
          !op.pre_bn.valid() && conv_bf3p_form(tensors[op.x].H, tensors[op.x].W, true, op.Cout) != 0) {
        // split-bf16 mode, fragment order: [Cout/32][KS][s 2][plane 3][lane 64][8 bf16]
        const int KS = op.Kpad / BK, NT32 = (op.Cout + 31) / 32;
        std::vector<uint16_t> wf((size_t)NT32 * KS * 3072, 0);
        size_t o = 0;
        for (int nt = 0; nt < NT32; ++nt)
          for (int ks = 0; ks < KS; ++ks)
            for (int s2 = 0; s2 < 2; ++s2) {
              uint16_t* blk = wf.data() + o;
              o += 3 * 512;
              for (int ln = 0; ln < 64; ++ln)
                for (int t = 0; t < 8; ++t) {
                  const int row = nt * 32 + (ln & 31);
                  float r = row < op.Cout ? packed[(size_t)row * op.Kpad + (size_t)ks * BK + 16 * s2 + 8 * (ln >> 5) + t] : 0.f;
                  for (int p3 = 0; p3 < 3; ++p3) {
                    const uint16_t b = f32_to_bf16_rne(r);
                    blk[p3 * 512 + ln * 8 + t] = b;
                    r -= bf16_to_f32(b);
                  }
                }
            }
        if ((uint64_t)wf.size() * 2 < 0xFFFFFFF0ull) {
          void* d = nullptr;
          DIF_HIP(hipMalloc(&d, wf.size() * sizeof(uint16_t)));
          allocs.push_back(d);
          DIF_HIP(hipMemcpy(d, wf.data(), wf.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
          op.d_w3f = d;
          op.w3f_bytes = (uint32_t)(wf.size() * sizeof(uint16_t));
        }
      }
      // fragment-order copy for the B-direct patch mainloop (conv.hip: gemm_mainloop_patch_bd): every layer the patch paths
      // can take (3x3 / stride 1 / pad 1, whole 32-channel slices)
      op.d_w_frag = nullptr;
      op.w_frag_bytes = 0;
      if (!op.d_w3f && op.k_order == 1 && op.KH == 3 && op.KW == 3 && op.stride == 1 && op.pad_t == 1 && op.pad_l == 1 &&
          !op.pre_bn.valid()) {
        const int KS = op.Kpad / BK, NT32 = (op.Cout + 31) / 32;
        std::vector<float> frag((size_t)NT32 * 32 * op.Kpad, 0.f);
        size_t o = 0;
        for (int nt = 0; nt < NT32; ++nt)
          for (int ks = 0; ks < KS; ++ks)
            for (int s2 = 0; s2 < 2; ++s2)
              for (int u = 0; u < 2; ++u)
                for (int hh = 0; hh < 2; ++hh)
                  for (int nn = 0; nn < 32; ++nn)
                    for (int t = 0; t < 4; ++t, ++o) {
                      const int row = nt * 32 + nn;
                      if (row < op.Cout) frag[o] = packed[(size_t)row * op.Kpad + (size_t)ks * BK + 16 * s2 + 8 * hh + 4 * u + t];
                    }
        if ((uint64_t)frag.size() * 4 < 0xFFFFFFF0ull) {
          if (upload(this, frag, &op.d_w_frag)) return -1;
          op.w_frag_bytes = (uint32_t)(frag.size() * 4);
        }
      }
      // 16-column fragment order for the one-image kernel (conv_minitile.hpp): every layer it can take -- pointwise layers
      // and multi-tap layers with whole 32-channel slices.  A (column tile, 16-k chunk) is 1 KB, lane l at byte 16 l: the
      // kernel's B loads are whole 128-byte lines (from the [Cout][Kpad] matrix a wave instruction touched 16 half lines)
      op.d_w_f16 = nullptr;
      op.w_f16_bytes = 0;
      {
        const bool pw1 = op.KH == 1 && op.KW == 1 && op.pad_t == 0 && op.pad_l == 0 && op.Cin % BK == 0;
        if ((pw1 || (op.k_order == 1 && op.Cin % BK == 0)) && !op.d_w_raw) {
          const int nch = op.Kpad / 16, NT16 = (op.Cout + 15) / 16;
          std::vector<float> f16((size_t)NT16 * nch * 256, 0.f);
          size_t o = 0;
          for (int ct = 0; ct < NT16; ++ct)
            for (int c = 0; c < nch; ++c)
              for (int ln = 0; ln < 64; ++ln)
                for (int t = 0; t < 4; ++t, ++o) {
                  const int row = ct * 16 + (ln & 15);
                  if (row < op.Cout) f16[o] = packed[(size_t)row * op.Kpad + (size_t)c * 16 + 4 * (ln >> 4) + t];
                }
          if ((uint64_t)f16.size() * 4 < 0xFFFFFFF0ull) {
            if (upload(this, f16, &op.d_w_f16)) return -1;
            op.w_f16_bytes = (uint32_t)(f16.size() * 4);
          }
        }
      }
      fold(this, op.bn, op.bias, op.Cout, &scale, &shift);
      if (!scale.empty() && upload(this, scale, &op.d_scale)) return -1;
      if (!shift.empty() && upload(this, shift, &op.d_shift)) return -1;
      if (op.alpha >= 0 && upload(this, params[op.alpha].data, &op.d_alpha)) return -1;
      if (op.alpha < 0 && op.const_alpha != 0.f &&
          upload(this, std::vector<float>((size_t)op.Cout, op.const_alpha), &op.d_alpha))
        return -1;
    } else if (op.kind == OP_GDCTAIL) {
      if (upload(this, params[op.w].data, &op.d_w)) return -1;            // [HW][512]
      if (upload(this, params[op.w_pw].data, &op.d_w_pw)) return -1;      // [512][emd]
      if (upload(this, params[op.w_dense].data, &op.d_w_dense)) return -1;   // [emd][emd]
      fold(this, op.bn, -1, op.Cin, &scale, &shift);
      if (!scale.empty() && upload(this, scale, &op.d_scale)) return -1;
      if (!shift.empty() && upload(this, shift, &op.d_shift)) return -1;
    } else if (op.kind == OP_DWFULL || op.kind == OP_DWCONV) {
      if (upload(this, params[op.w].data, &op.d_w)) return -1;   // [H][W][C][1] == [HW][C]
      fold(this, op.bn, -1, op.Cout, &scale, &shift);
      if (!scale.empty() && upload(this, scale, &op.d_scale)) return -1;
      if (!shift.empty() && upload(this, shift, &op.d_shift)) return -1;
    }
    if (op.pre_bn.valid()) {
      fold(this, op.pre_bn, -1, op.Cin, &scale, &shift);
      if (upload(this, scale, &op.d_pre_scale)) return -1;
      if (upload(this, shift, &op.d_pre_shift)) return -1;
    }
    if (op.y2 >= 0) {
      fold(this, op.bn2, -1, op.Cout, &scale, &shift);
      if (!scale.empty() && upload(this, scale, &op.d_scale2)) return -1;
      if (!shift.empty() && upload(this, shift, &op.d_shift2)) return -1;
      if (op.alpha2 >= 0 && upload(this, params[op.alpha2].data, &op.d_alpha2)) return -1;
    }
  }

  // ---- outputs read only at stride 2.  A two-output convolution whose FIRST output feeds nothing but a 1x1 / stride 2
  // convolution (IResNet: conv1's activation and the last block of a stage feed the next stage's downsample shortcut,
  // while the batch-normalised second output feeds its 3x3) writes that output subsampled -- a quarter of the bytes;
  // the stem's 822 MB output at batch 256 was the layer's bound -- and the reader runs at stride 1 on the dense quarter.
  for (Op& P : ops) {
    if (!use_ysub || P.kind != OP_CONV || P.y < 0 || P.y2 < 0 || P.y_sub || is_output(P.y) || tensors[P.y].parent >= 0 ||
        tensors[P.y2].parent >= 0 || P.Cout % 4 != 0)
      continue;
    bool viewed = false;
    for (const TensorDesc& t : tensors) viewed |= t.parent == P.y;
    Op* reader = nullptr;
    int readers = 0;
    for (Op& D : ops) {
      if (&D == &P) continue;
      const int uses = (D.x == P.y) + (D.res == P.y) + (D.y == P.y) + (D.y2 == P.y);
      if (uses) {
        readers += uses;
        reader = &D;
      }
    }
    if (viewed || readers != 1 || reader->kind != OP_CONV || reader->x != P.y) continue;
    if (reader->KH != 1 || reader->KW != 1 || reader->stride != 2 || reader->pad_t != 0 || reader->pad_l != 0) continue;
    TensorDesc& t = tensors[P.y];
    P.y_sub = true;
    t.H = (t.H + 1) / 2;
    t.W = (t.W + 1) / 2;
    reader->stride = 1;
  }

  // ---- activation buffers: liveness over the op list, first-fit reuse
  for (TensorDesc& t : tensors) t.first_def = t.last_use = t.buf = -1;
  // views share their parent's storage: liveness and buffers are tracked on the root tensor
  auto R = [&](int t) { return t < 0 ? t : root_of(t); };
  for (int i = 0; i < (int)ops.size(); ++i) {
    const Op& op = ops[i];
    for (int t : {R(op.y), R(op.y2)})
      if (t >= 0 && tensors[t].first_def < 0) tensors[t].first_def = i;
    for (int t : {R(op.x), R(op.res), R(op.y), R(op.y2)})
      if (t >= 0) tensors[t].last_use = std::max(tensors[t].last_use, i);
  }
  tensors[output_tensor].last_use = (int)ops.size();   // written straight into the caller's buffer
  for (int e : extra_outputs) tensors[e].last_use = (int)ops.size();
  buf_elems.clear();
  std::vector<int> free_list;
  for (int i = 0; i < (int)ops.size(); ++i) {
    const Op& op = ops[i];
    for (int t : {R(op.y), R(op.y2)}) {
      if (t < 0 || is_output(t) || tensors[t].buf >= 0) continue;
      const int64_t need = tensors[t].elems();
      int best = -1;
      for (int k = 0; k < (int)free_list.size(); ++k) {
        const int b = free_list[k];
        if (buf_elems[b] >= need && (best < 0 || buf_elems[b] < buf_elems[free_list[best]])) best = k;
      }
      if (best >= 0) {
        tensors[t].buf = free_list[best];
        free_list.erase(free_list.begin() + best);
      } else {
        buf_elems.push_back(need);
        tensors[t].buf = (int)buf_elems.size() - 1;
      }
    }
    for (int t : {R(op.x), R(op.res), R(op.y), R(op.y2)})
      if (t >= 0 && !is_output(t) && tensors[t].last_use == i && tensors[t].buf >= 0) {
        if (std::find(free_list.begin(), free_list.end(), tensors[t].buf) == free_list.end())
          free_list.push_back(tensors[t].buf);
      }
  }
  sk_max_blocks = conv_max_blocks();
  if (const char* e = getenv("DIF_SK_SPIN_LIMIT")) sk_spin_limit = atoi(e);   // test hook (tests/test_embed_gpu.py)
  // Lanes: the batch is cut into nl parts that run the launch list on nl streams.
  //  * long launches (IResNet-100 at batch 256: 12 GFLOP per launch per lane): two lanes, each launch sized for the
  //    whole chip -- one lane's tail hides under the other lane's head (+4 %);
  //  * short launches (ResNet50V2: 0.1 ms each, a third of it ramp and a ragged last round of tiles): two lanes whose
  //    persistent grids take HALF the chip's resident slots each, so both lanes' kernels are co-resident all the time
  //    and either one's ramp, tail and epilogue bursts leave the matrix pipes to the other (+3..5 % at batch 128-256;
  //    with whole-chip grids the second lane's blocks only queue behind the first's: -1 %).
  int n_conv = 0;
  for (const Op& op : ops) n_conv += op.kind == OP_CONV;
  const double per_launch = flops_per_image() * (max_batch / 2) / (n_conv > 0 ? n_conv : 1);
  const bool long_launches = per_launch >= 10e9;
  int nl = getenv("DIF_STREAMS") ? atoi(getenv("DIF_STREAMS")) : 2;      // tests force the single-lane executor with it
  lane_split = (opt_lane_split >= 0 ? opt_lane_split : (long_launches ? 0 : 1)) != 0;
  if (nl < 1) nl = 1;
  if (nl > 8) nl = 8;
  if (max_batch < lane_min_images()) nl = 1;               // (a forward of lane_min_images() images is already split: batch 64 13.4 -> 12.7 ms)
  lanes.assign(nl, Lane());
  DIF_HIP(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
  for (int l = 0; l < nl; ++l) {
    Lane& L = lanes[l];
    L.cap = l == 0 ? max_batch : (max_batch + nl - 1) / nl;   // lane 0 also serves unsplit and profiled forwards
    if (l > 0 || (nl > 1 && opt_lane_prio == 0)) {
      // A HIP stream is multiplexed onto one of a few hardware queues per PRIORITY level, the least referenced one.  A host
      // framework that has filled the normal level before this net is finalized -- torch creates its whole stream pool with
      // the first collective -- can leave this lane on the hardware queue of the caller's stream: the lanes then run one
      // after the other and the forward is 5-18 % SLOWER than on one lane (measured: profiles/r04_rccl_lanes.txt).  A
      // multi-lane forward therefore runs ALL its lanes on streams of the LEAST-priority level, which nothing else in the
      // process uses -- lane 0 included, forked from and joined to the caller's stream by events, so the lanes compete as
      // equals (option "lane_prio": 0 = that (default), 1 = lane 0 on the caller's stream, the others on normal-priority
      // streams, as before round 4).
      int prio_least = 0, prio_greatest = 0;
      DIF_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
      if (opt_lane_prio == 0 && prio_least != prio_greatest)
        DIF_HIP(hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, prio_least));
      else
        DIF_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
      DIF_HIP(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    }
    void* d = nullptr;
    DIF_HIP(hipMalloc(&d, (size_t)sk_max_blocks * conv_slab_floats() * sizeof(float)));
    allocs.push_back(d);
    L.sk_slab = static_cast<float*>(d);
    DIF_HIP(hipMalloc(&d, (size_t)sk_max_blocks * sizeof(unsigned)));
    allocs.push_back(d);
    L.sk_flag = static_cast<unsigned*>(d);
    DIF_HIP(hipMemset(L.sk_flag, 0, (size_t)sk_max_blocks * sizeof(unsigned)));
    L.sk_epoch = 0;
    DIF_HIP(hipMalloc(&d, (size_t)GDC_TAIL_WS_FLOATS * sizeof(float)));
    allocs.push_back(d);
    L.tail_ws = static_cast<float*>(d);
    DIF_HIP(hipMemset(L.tail_ws, 0, (size_t)GDC_TAIL_WS_FLOATS * sizeof(float)));
    // persistent-grid size of this lane's stream-K launches.  DIF_SK_LANE_SPLIT=1 gives each lane 1/nl of the
    // chip's resident slots, so the lanes' grids are co-resident instead of queueing behind each other
    L.sk_max_blocks = (lane_split && nl > 1) ? sk_max_blocks / nl : sk_max_blocks;

    L.bufs.assign(buf_elems.size(), nullptr);
    for (size_t b = 0; b < buf_elems.size(); ++b) {
      float* f = nullptr;
      DIF_HIP(hipMalloc(&f, (size_t)buf_elems[b] * L.cap * sizeof(float)));
      allocs.push_back(f);
      L.bufs[b] = f;
    }
  }
  finalized = true;
  return 0;
}

// ----------------------------------------------------------------------------- forward
const char* Net::kernel_name(const Op& op, int n) const {
  switch (op.kind) {
    case OP_INPUT: return "input_convert_kernel";
    case OP_MAXPOOL: return "maxpool_kernel";
    case OP_DWFULL: return "dwfull_kernel";
    case OP_GDCTAIL: return op.ran_kernel && op.ran_kernel[0] ? op.ran_kernel : "gdc_tail_kernel";   // (as the convolutions: what ran last)
    case OP_DWCONV: return "dwconv_kernel";
    case OP_L2NORM: return "l2norm_kernel";
    case OP_LRN: return "lrn_kernel";
    case OP_ZERO: return "memset";
    case OP_UPSAMPLE: return "upsample2_kernel";
    case OP_COPY: return "copy_to_view_kernel";
    case OP_CONV:
      if (op.d_w_raw && use_stem) return op.stem_mfma ? "stem_mfma_kernel" : "stem3x3_kernel";
      // the instantiation the op's last launch took (conv.hip: conv_igemm_kernel / conv_pipe_kernel / conv_bdp_kernel and
      // their operand forms -- chosen per launch from the batch and the lane's grid); before any forward, the family
      return op.ran_kernel && op.ran_kernel[0] ? op.ran_kernel : "conv_igemm_kernel<64x64>";
  }
  return "?";
}

// One layer op of one lane (a lane = its own activation buffers + stream-K workspace).
int Net::run_op(const Op& op, Lane& L, const void* xin, int n, int layout, int dtype, float* out, hipStream_t st) {
  auto ptr = [&](int t) -> float* {
    if (t < 0) return nullptr;
    // outputs are laid out one after another, each for the WHOLE batch: a lane writes its images' part of each
    if (is_output(t)) return out + (int64_t)L.out_n * output_offset(t) + (int64_t)L.out_start * tensors[t].elems();
    return L.bufs[tensors[root_of(t)].buf];
  };
  switch (op.kind) {
    case OP_INPUT: {
      InputArgs a;
      a.x = xin;
      a.y = ptr(op.y);
      a.N = n;
      a.H = in_h;
      a.W = in_w;
      a.layout = layout;
      a.dtype = dtype;
      a.scale = in_scale;
      a.bias[0] = in_bias[0];
      a.bias[1] = in_bias[1];
      a.bias[2] = in_bias[2];
      a.bgr = bgr;
      if (input_convert_run(a, st)) return -1;
      break;
    }
    case OP_CONV: {
      const TensorDesc& xd = tensors[op.x];
      const TensorDesc& yd = tensors[op.y >= 0 && !op.y_sub ? op.y : op.y2];
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      a.x = ptr(op.x);
      a.w = op.d_w;
      a.w3f = op.d_w3f;
      a.w3f_bytes = op.w3f_bytes;
      a.w_frag = op.d_w_frag;
      a.w_frag_bytes = op.w_frag_bytes;
      a.w_f16 = op.d_w_f16;
      a.w_f16_bytes = op.w_f16_bytes;
      a.bf_terms = bf_terms;
      a.y = ptr(op.y);
      a.y2 = ptr(op.y2);
      a.scale = op.d_scale;
      a.shift = op.d_shift;
      a.alpha = op.d_alpha;
      a.res = ptr(op.res);
      a.scale2 = op.d_scale2;
      a.shift2 = op.d_shift2;
      a.alpha2 = op.d_alpha2;
      a.pre_scale = op.d_pre_scale;
      a.pre_shift = op.d_pre_shift;
      a.pre_act = op.pre_act;
      a.N = n;
      a.H = xd.H;
      a.W = xd.W;
      a.Cin = op.Cin;
      a.Ho = yd.H;
      a.Wo = yd.W;
      a.Cout = op.Cout;
      a.KH = op.KH;
      a.KW = op.KW;
      a.stride = op.stride;
      a.pad_t = op.pad_t;
      a.pad_l = op.pad_l;
      a.Kpad = op.Kpad;
      a.k_order = op.k_order;
      a.M = n * yd.H * yd.W;
      a.act = op.act;
      a.act2 = op.act2;
      a.y_sub = op.y_sub;
      if (op.res >= 0) {
        a.res_H = tensors[op.res].H;
        a.res_W = tensors[op.res].W;
        a.res_stride = op.res_stride;
      } else {
        a.res_H = yd.H;
        a.res_W = yd.W;
        a.res_stride = 1;
      }
      {
        const TensorDesc& vd = tensors[op.y >= 0 ? op.y : op.y2];
        if (vd.parent >= 0) {
          const TensorDesc& pd = tensors[vd.parent];
          a.y_ld = pd.C;
          a.y_coff = vd.coff;
          a.y_H = pd.H;
          a.y_W = pd.W;
          a.y_oy = vd.oy;
          a.y_ox = vd.ox;
        } else {
          a.y_ld = op.Cout;
          a.y_coff = 0;
          a.y_H = yd.H;
          a.y_W = yd.W;
          a.y_oy = a.y_ox = 0;
        }
      }
      a.sk_slab = L.sk_slab;
      a.sk_flag = L.sk_flag;
      a.sk_max_blocks = lanes_active ? L.sk_max_blocks : sk_max_blocks;   // one lane alone takes the whole chip
      a.sk_epoch = ++L.sk_epoch;
      a.sk_spin_limit = sk_spin_limit;
      a.use_pipe = use_pipe;
      a.off = conv_off;
      a.dbg = conv_dbg;
      a.lanes = lanes_active ? (int)lanes.size() : 1;
      a.bdp_mode = use_bdp == 2 ? 2 : ((use_bdp == 0 || lane_split) ? 1 : 0);
      a.trace = trace_buf ? trace_buf + trace_off[&op - ops.data()] * 8 : nullptr;
      if (op.d_w_raw && op.stem_mfma && use_stem) {
        if (stem_mfma_run(a.x, op.d_w_raw, a.scale, a.shift, a.alpha, a.scale2, a.shift2, a.alpha2, a.y, a.y2, n, a.H, a.W,
                          a.Ho, a.Wo, a.KH, a.stride, a.act, a.act2, a.y_sub, st))
          return -1;
        break;
      }
      if (op.d_w_raw && use_stem) {
        if (stem3x3_run(a.x, op.d_w_raw, a.scale, a.shift, a.alpha, a.scale2, a.shift2, a.alpha2, a.y, a.y2, n, a.H, a.W, a.Cout,
                        a.act, a.act2, st))
          return -1;
        break;
      }
      if (conv_run(a, st)) return -1;
      op.ran_kernel = conv_last_kernel();
      break;
    }
    case OP_MAXPOOL: {
      const TensorDesc& xd = tensors[op.x];
      const TensorDesc& yd = tensors[op.y];
      PoolArgs a;
      memset(&a, 0, sizeof(a));
      a.x = ptr(op.x);
      a.y = ptr(op.y);
      a.y2 = ptr(op.y2);
      a.scale2 = op.d_scale2;
      a.shift2 = op.d_shift2;
      a.N = n;
      a.H = xd.H;
      a.W = xd.W;
      a.C = xd.C;
      a.Ho = yd.H;
      a.Wo = yd.W;
      a.k = op.KH;
      a.stride = op.stride;
      a.pad_t = op.pad_t;
      a.pad_l = op.pad_l;
      a.zero_pad = op.zero_pad;
      a.act2 = op.act2;
      a.mode = op.pool_mode;
      if (yd.parent >= 0) {
        const TensorDesc& pd = tensors[yd.parent];
        a.y_ld = pd.C;
        a.y_coff = yd.coff;
        a.y_H = pd.H;
        a.y_W = pd.W;
        a.y_oy = yd.oy;
        a.y_ox = yd.ox;
      } else {
        a.y_ld = yd.C;
        a.y_coff = 0;
        a.y_H = yd.H;
        a.y_W = yd.W;
        a.y_oy = a.y_ox = 0;
      }
      if (maxpool_run(a, st)) return -1;
      break;
    }
    case OP_GDCTAIL: {
      const TensorDesc& xd = tensors[op.x];
      if (gdc_tail_run(ptr(op.x), op.d_w, op.d_scale, op.d_shift, op.d_w_pw, op.d_w_dense, ptr(op.y), n, xd.H * xd.W,
                       op.Cout, 1e-12f, L.tail_ws, st))
        return -1;
      op.ran_kernel = n <= 2 && op.Cout % 32 == 0 ? "gdc_tail_a_kernel+gdc_tail_b_kernel" : "gdc_tail_kernel";
      break;
    }
    case OP_DWFULL: {
      const TensorDesc& xd = tensors[op.x];
      if (dwfull_run(ptr(op.x), op.d_w, op.d_scale, op.d_shift, ptr(op.y), n, xd.H * xd.W, xd.C, st)) return -1;
      break;
    }
    case OP_DWCONV: {
      const TensorDesc& xd = tensors[op.x];
      const TensorDesc& yd = tensors[op.y];
      if (dwconv_run(ptr(op.x), op.d_w, op.d_scale, op.d_shift, ptr(op.y), n, xd.H, xd.W, xd.C, op.KH, op.stride, op.pad_t,
                     op.pad_l, yd.H, yd.W, op.act, st))
        return -1;
      break;
    }
    case OP_LRN: {
      const TensorDesc& xd = tensors[op.x];
      // tf.nn.lrn defaults: depth_radius 5, bias 1; alpha / beta from inceptionv3.py:95
      if (lrn_run(ptr(op.x), ptr(op.y), (int64_t)n * xd.H * xd.W, xd.C, 5, 1.f, 1e-4f, 0.75f, st)) return -1;
      break;
    }
    case OP_UPSAMPLE: {
      const TensorDesc& xd = tensors[op.x];
      const TensorDesc& yd = tensors[op.y];
      const int ld = yd.parent >= 0 ? tensors[yd.parent].C : yd.C;
      if (upsample2_run(ptr(op.x), ptr(op.y), n, xd.H, xd.W, xd.C, ld, yd.coff, st)) return -1;
      break;
    }
    case OP_COPY: {
      const TensorDesc& xd = tensors[op.x];
      const TensorDesc& yd = tensors[op.y];
      const int ld = yd.parent >= 0 ? tensors[yd.parent].C : yd.C;
      if (copy_to_view_run(ptr(op.x), ptr(op.y), (int64_t)n * xd.H * xd.W, xd.C, ld, yd.coff, st)) return -1;
      break;
    }
    case OP_ZERO: {
      const TensorDesc& yd = tensors[op.y];
      DIF_HIP(hipMemsetAsync(ptr(op.y), 0, (size_t)n * yd.elems() * sizeof(float), st));
      break;
    }
    case OP_L2NORM:
      if (l2norm_run(ptr(op.x), ptr(op.y), n, op.Cout, 1e-12f, st)) return -1;
      break;
  }
  return 0;
}

// Forward.  Batches of >= lane_min_images() (64 at 112 x 112; 5 frames at 416 x 416) are split over the lanes: lane 0 runs on the caller's
// stream, the others on internal HIP streams, layer by layer in lock-step order of submission.
// Every convolution launch fills the chip, so the lanes mostly alternate; what overlaps is each
// kernel's tail (its last, partly filled round of blocks) with the head of the other lane's
// kernel -- measured +2..3 % on both networks at batch 256 (tools/two_streams.py).
int Net::embed(const void* xin, int n, int layout, int dtype, float* out, hipStream_t st, float* op_ms) {
  if (!finalized) return set_error("dif_net_embed: call dif_net_finalize first");
  if (n < 0 || n > max_batch) return set_error("dif_net_embed: batch %d outside [0, max_batch=%d]", n, max_batch);
  if (n == 0) return 0;
  if (!xin || !out) return set_error("dif_net_embed: null pointer");
  if (layout != DIF_LAYOUT_NHWC && layout != DIF_LAYOUT_NCHW) return set_error("dif_net_embed: bad layout %d", layout);
  if (dtype != DIF_DTYPE_F32 && dtype != DIF_DTYPE_U8) return set_error("dif_net_embed: bad dtype %d", dtype);

  const int nl = (int)lanes.size();
  lanes_active = !(op_ms || nl == 1 || n < lane_min_images());
  lanes[0].out_n = n;
  lanes[0].out_start = 0;
  if (!lanes_active) {
    std::vector<hipEvent_t> ev;
    if (op_ms) {
      ev.resize(ops.size() + 1);
      for (auto& e : ev) DIF_HIP(hipEventCreate(&e));
      DIF_HIP(hipEventRecord(ev[0], st));
    }
    for (size_t i = 0; i < ops.size(); ++i) {
      if (run_op(ops[i], lanes[0], xin, n, layout, dtype, out, st)) return -1;
      if (op_ms) DIF_HIP(hipEventRecord(ev[i + 1], st));
    }
    if (op_ms) {
      DIF_HIP(hipStreamSynchronize(st));
      for (size_t i = 0; i < ops.size(); ++i) DIF_HIP(hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]));
      for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return 0;
  }

  const size_t in_bytes = (size_t)in_h * in_w * 3 * (dtype == DIF_DTYPE_U8 ? 1 : 4);
  std::vector<int> start(nl + 1, 0);
  for (int l = 0; l < nl; ++l) {
    int c = n / nl + (l < n % nl ? 1 : 0);
    if (l > 0 && c > lanes[l].cap) c = lanes[l].cap;     // cannot happen: cap = ceil(max_batch / nl)
    start[l + 1] = start[l] + c;
  }
  const int l_first = lanes[0].stream ? 0 : 1;              // lane 0 has a stream of its own under "lane_prio" 0
  DIF_HIP(hipEventRecord(ev_start, st));
  for (int l = l_first; l < nl; ++l) DIF_HIP(hipStreamWaitEvent(lanes[l].stream, ev_start, 0));
  for (size_t i = 0; i < ops.size(); ++i) {
    for (int l = 0; l < nl; ++l) {
      const int c = start[l + 1] - start[l];
      if (c == 0) continue;
      hipStream_t ls = lanes[l].stream ? lanes[l].stream : st;
      lanes[l].out_n = n;
      lanes[l].out_start = start[l];
      if (run_op(ops[i], lanes[l], static_cast<const char*>(xin) + (size_t)start[l] * in_bytes, c, layout, dtype, out, ls))
        return -1;
    }
  }
  for (int l = l_first; l < nl; ++l) {
    DIF_HIP(hipEventRecord(lanes[l].done, lanes[l].stream));
    DIF_HIP(hipStreamWaitEvent(st, lanes[l].done, 0));
  }
  return 0;
}

// The shader clock held INSIDE the convolution kernels of one forward (single lane): every block of every conv launch
// records its life in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime); the ratio of the sums is the
// life-weighted mean clock.  The rooflines price against the 2.4 GHz peak; this is what the chip sustains under the
// kernels' own mix of MFMA, LDS and memory instructions (a register-only MFMA loop holds more: dif_probe_mfma_clock).
int Net::embed_clock(const void* xin, int n, int layout, int dtype, float* out, hipStream_t st, double* ghz) {
  if (!finalized) return set_error("dif_net_embed_clock: call dif_net_finalize first");
  if (!ghz) return set_error("dif_net_embed_clock: null output");
  if (n < 1 || n > max_batch) return set_error("dif_net_embed_clock: batch %d outside [1, max_batch=%d]", n, max_batch);
  trace_off.assign(ops.size(), 0);
  size_t total = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    trace_off[i] = total;
    if (ops[i].kind != OP_CONV) continue;
    const TensorDesc& yd = tensors[ops[i].y >= 0 && !ops[i].y_sub ? ops[i].y : ops[i].y2];
    const size_t tiles = (((size_t)n * yd.H * yd.W + 63) / 64) * (((size_t)ops[i].Cout + 63) / 64);
    total += tiles > (size_t)sk_max_blocks ? tiles : (size_t)sk_max_blocks;     // a launch is one block per tile or a resident grid
  }
  unsigned long long* d = nullptr;
  DIF_HIP(hipMalloc(&d, total * 8 * sizeof(unsigned long long)));
  DIF_HIP(hipMemsetAsync(d, 0, total * 8 * sizeof(unsigned long long), st));
  trace_buf = d;
  int rc = 0;
  lanes_active = false;
  lanes[0].out_n = n;
  lanes[0].out_start = 0;
  for (size_t i = 0; i < ops.size() && !rc; ++i) rc = run_op(ops[i], lanes[0], xin, n, layout, dtype, out, st);
  trace_buf = nullptr;
  std::vector<unsigned long long> h;
  if (!rc) {
    h.resize(total * 8);
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
      rc = set_error("dif_net_embed_clock: reading the records failed");
  }
  (void)hipFree(d);
  if (rc) return rc;
  if (conv_dbg & 256) {
    // development aid: per convolution, what its blocks did (100 MHz ticks -> us): the launch's span, the blocks' lifetimes,
    // and for conv_igemm_kernel records the time in mainloops / partial-tile hand-over / epilogues
    for (size_t i = 0; i < ops.size(); ++i) {
      if (ops[i].kind != OP_CONV) continue;
      const size_t end = i + 1 < ops.size() ? trace_off[i + 1] : total;
      unsigned long long t_lo = ~0ull, t_hi = 0;
      double life = 0, mn = 1e30, mx = 0, m = 0, f = 0, e = 0, steps = 0, tiles = 0;
      double p_set = 0, p_pro = 0, p_steps = 0, p_hand = 0, p_tiles = 0;
      int nb = 0;
      for (size_t b = trace_off[i]; b < end; ++b) {
        const unsigned long long* t = &h[b * 8];
        if ((t[7] & 0xff) == 0 || t[6] <= t[5]) continue;
        ++nb;
        t_lo = t[5] < t_lo ? t[5] : t_lo;
        t_hi = t[6] > t_hi ? t[6] : t_hi;
        const double l = (double)(t[6] - t[5]);
        life += l;
        mn = l < mn ? l : mn;
        mx = l > mx ? l : mx;
        if ((t[7] & 0xff) == 2) {                          // conv_pipe_kernel: set-up / prologue / K-steps / hand-over, tiles
          p_set += (double)t[0];
          p_pro += (double)t[1];
          p_steps += (double)t[2];
          p_hand += (double)t[3];
          p_tiles += (double)t[4];
        }
        if ((t[7] & 0xff) == 1) {
          m += (double)t[0];
          f += (double)t[1];
          e += (double)t[2];
          steps += (double)t[3];
          tiles += (double)t[4];
        }
      }
      if (nb)
        fprintf(stderr, "trace %-22s blocks %5d span %7.1f us | life mean %7.1f min %7.1f max %7.1f | main %6.1f fix %6.1f epi %6.1f us/block | "
                        "steps/block %.1f tiles/block %.2f\n", ops[i].name.c_str(), nb, (double)(t_hi - t_lo) / 100.0, life / nb / 100.0,
                mn / 100.0, mx / 100.0, m / nb / 100.0, f / nb / 100.0, e / nb / 100.0, steps / nb, tiles / nb);
      if (nb && p_tiles > 0)
        fprintf(stderr, "      pipelined: %.2f tiles/block | per tile: set-up %.2f prologue %.2f K-steps %.2f hand-over %.2f us\n", p_tiles / nb,
                p_set / p_tiles / 100.0, p_pro / p_tiles / 100.0, p_steps / p_tiles / 100.0, p_hand / p_tiles / 100.0);
    }
  }
  if (conv_dbg & 512) {
    // development aid: placement of the blocks of the persistent B-direct launches -- per XCD and per CU, how many blocks ran
    // there, how long they lived and at what clock (records whose t[4] carries HW_ID / XCC_ID: conv_bdp_kernel)
    for (size_t i = 0; i < ops.size(); ++i) {
      if (ops[i].kind != OP_CONV) continue;
      const size_t end = i + 1 < ops.size() ? trace_off[i + 1] : total;
      std::map<unsigned, std::vector<double>> by_xcd, by_cu;     // -> {blocks, sum life, sum cycles, max end}
      unsigned long long t_lo = ~0ull;
      for (size_t b = trace_off[i]; b < end; ++b)
        if ((h[b * 8 + 7] & 0xff) == 1 && h[b * 8 + 4] && h[b * 8 + 5] < t_lo) t_lo = h[b * 8 + 5];
      int nb = 0;
      for (size_t b = trace_off[i]; b < end; ++b) {
        const unsigned long long* t = &h[b * 8];
        if ((t[7] & 0xff) != 1 || !t[4] || t[6] <= t[5]) continue;
        ++nb;
        const unsigned hw = (unsigned)t[4], xcc = (unsigned)(t[4] >> 32) & 0xf;
        const unsigned cu = (xcc << 16) | (hw & 0xff00);          // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        for (auto* mp : {&by_xcd[xcc], &by_cu[cu]}) {
          if (mp->empty()) mp->assign(5, 0.0);
          (*mp)[0] += 1;
          (*mp)[1] += (double)(t[6] - t[5]);
          (*mp)[2] += (double)(t[7] >> 8);
          (*mp)[3] = std::max((*mp)[3], (double)(t[6] - t_lo));
          (*mp)[4] += (double)(t[5] - t_lo);
        }
      }
      if (!nb) continue;
      fprintf(stderr, "place %-20s blocks %d on %zu CUs |", ops[i].name.c_str(), nb, by_cu.size());
      for (auto& kv : by_xcd)
        fprintf(stderr, " xcd%u: %d blk life %.0f us end %.0f us %.2f GHz |", kv.first, (int)kv.second[0], kv.second[1] / kv.second[0] / 100.0,
                kv.second[3] / 100.0, kv.second[2] / (kv.second[1] * 10.0));
      std::map<int, std::vector<double>> hist;                     // blocks per CU -> {CUs, sum mean life, sum end, sum start}
      for (auto& kv : by_cu) {
        auto& hrec = hist[(int)kv.second[0]];
        if (hrec.empty()) hrec.assign(4, 0.0);
        hrec[0] += 1;
        hrec[1] += kv.second[1] / kv.second[0];
        hrec[2] += kv.second[3];
        hrec[3] += kv.second[4] / kv.second[0];
      }
      for (auto& kv : hist)
        fprintf(stderr, " %d/CU: %d CUs life %.0f start %.0f end %.0f us |", kv.first, (int)kv.second[0], kv.second[1] / kv.second[0] / 100.0,
                kv.second[3] / kv.second[0] / 100.0, kv.second[2] / kv.second[0] / 100.0);
      // by dispatch round (block index / CUs: the blocks of one round are the n-th residents of their CUs) and by wave slot
      double rl[8] = {0}, rn[8] = {0}, wl[16] = {0}, wn[16] = {0};
      for (size_t b = trace_off[i]; b < end; ++b) {
        const unsigned long long* t = &h[b * 8];
        if ((t[7] & 0xff) != 1 || !t[4] || t[6] <= t[5]) continue;
        const size_t r = std::min<size_t>((b - trace_off[i]) / by_cu.size(), 7);
        rl[r] += (double)(t[6] - t[5]);
        rn[r] += 1;
        wl[t[4] & 0xf] += (double)(t[6] - t[5]);
        wn[t[4] & 0xf] += 1;
      }
      for (int r = 0; r < 8; ++r)
        if (rn[r] > 0) fprintf(stderr, " round%d: %.0f us (%d) |", r, rl[r] / rn[r] / 100.0, (int)rn[r]);
      for (int r = 0; r < 16; ++r)
        if (wn[r] > 0) fprintf(stderr, " slot%d: %.0f us (%d) |", r, wl[r] / wn[r] / 100.0, (int)wn[r]);
      fprintf(stderr, "\n");
    }
  }
  double cyc = 0, ticks = 0;
  for (size_t b = 0; b < total; ++b) {
    const unsigned long long* t = &h[b * 8];
    const unsigned kind = (unsigned)(t[7] & 0xff);
    if ((kind == 1 || kind == 2) && t[6] > t[5]) {
      cyc += (double)(t[7] >> 8);
      ticks += (double)(t[6] - t[5]);
    }
  }
  *ghz = ticks > 0 ? cyc / (ticks * 10.0) : 0.0;
  return 0;
}

}  // namespace dif
