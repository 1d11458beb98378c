// C ABI entry points of the embedding network and the ArcMargin head (include/dif.h).
#include <new>
#include <string.h>

#include "../../include/dif.h"
#include "arcmargin.hpp"
#include "dif_internal.hpp"
#include "net.hpp"

using namespace dif;

struct dif_net {
  Net net;
};
struct dif_arcmargin {
  ArcMargin a;
};

extern "C" {

int dif_net_create(dif_net** out, const char* arch, const char* head, int emd_size, int in_h, int in_w) {
  if (!out || !arch) return set_error("dif_net_create: null argument");
  dif_net* h = new (std::nothrow) dif_net();
  if (!h) return set_error("dif_net_create: out of host memory");
  h->net.arch = arch;
  h->net.head = head ? head : "v2";
  h->net.emd = emd_size;
  h->net.in_h = in_h;
  h->net.in_w = in_w;
  if (h->net.build()) {
    delete h;
    return -1;
  }
  *out = h;
  return 0;
}

int dif_net_destroy(dif_net* h) {
  delete h;
  return 0;
}

int dif_net_param_count(const dif_net* h) { return h ? (int)h->net.params.size() : 0; }

int dif_net_param_info(const dif_net* h, int i, const char** name, int* ndim, int64_t shape[4]) {
  if (!h || i < 0 || i >= (int)h->net.params.size()) return set_error("dif_net_param_info: index out of range");
  const Param& p = h->net.params[i];
  if (name) *name = p.name.c_str();
  if (ndim) *ndim = (int)p.shape.size();
  if (shape)
    for (size_t k = 0; k < 4; ++k) shape[k] = k < p.shape.size() ? p.shape[k] : 1;
  return 0;
}

int dif_net_set_param(dif_net* h, const char* name, const float* data_host, int64_t count) {
  if (!h || !name || !data_host) return set_error("dif_net_set_param: null argument");
  auto it = h->net.pindex.find(name);
  if (it == h->net.pindex.end()) return set_error("dif_net_set_param: no parameter named '%s'", name);
  Param& p = h->net.params[it->second];
  if (count != p.count())
    return set_error("dif_net_set_param: '%s' has %lld elements, got %lld", name, (long long)p.count(),
                     (long long)count);
  p.data.assign(data_host, data_host + count);
  p.set = true;
  h->net.finalized = false;   // weights changed: finalize again before the next forward
  return 0;
}

int dif_net_get_param(const dif_net* h, const char* name, float* data_host, int64_t count) {
  if (!h || !name || !data_host) return set_error("dif_net_get_param: null argument");
  auto it = h->net.pindex.find(name);
  if (it == h->net.pindex.end()) return set_error("dif_net_get_param: no parameter named '%s'", name);
  const Param& p = h->net.params[it->second];
  if (!p.set) return set_error("dif_net_get_param: '%s' was never set", name);
  if (count != p.count()) return set_error("dif_net_get_param: '%s' size mismatch", name);
  memcpy(data_host, p.data.data(), (size_t)count * sizeof(float));
  return 0;
}

int dif_net_set_input_transform(dif_net* h, float scale, const float bias[3], int flags) {
  if (!h) return set_error("dif_net_set_input_transform: null handle");
  if (flags & ~(DIF_INPUT_BGR | DIF_INPUT_HFLIP)) return set_error("dif_net_set_input_transform: unknown flags");
  h->net.in_scale = scale;
  for (int k = 0; k < 3; ++k) h->net.in_bias[k] = bias ? bias[k] : 0.f;
  h->net.bgr = flags;
  return 0;
}

int dif_net_finalize(dif_net* h, int max_batch) {
  if (!h) return set_error("dif_net_finalize: null handle");
  return h->net.finalize(max_batch);
}

int dif_net_embed_clock(dif_net* h, const void* x_dev, int n, int layout, int dtype, float* out_dev, double* ghz_out,
                        void* stream) {
  if (!h) return set_error("dif_net_embed_clock: null handle");
  return h->net.embed_clock(x_dev, n, layout, dtype, out_dev, static_cast<hipStream_t>(stream), ghz_out);
}

int dif_net_set_option(dif_net* h, const char* key, int value) {
  if (!h || !key) return set_error("dif_net_set_option: null argument");
  return h->net.set_option(key, value);
}

const char* dif_net_option_name(int i) {
  const char* const* t = Net::option_names();
  int n = 0;
  while (t[n]) ++n;
  return i >= 0 && i < n ? t[i] : nullptr;
}

int dif_net_output_dim(const dif_net* h, int64_t shape[3]) {
  if (!h || !shape) return set_error("dif_net_output_dim: null argument");
  const TensorDesc& t = h->net.tensors[h->net.output_tensor];
  shape[0] = t.C;
  shape[1] = t.H;
  shape[2] = t.W;
  return 0;
}

int dif_net_output_count(const dif_net* h) { return h ? 1 + (int)h->net.extra_outputs.size() : 0; }

int dif_net_output_info(const dif_net* h, int i, int64_t shape[3]) {
  if (!h || !shape || i < 0 || i > (int)h->net.extra_outputs.size()) return set_error("dif_net_output_info: bad index");
  const TensorDesc& t = h->net.tensors[i == 0 ? h->net.output_tensor : h->net.extra_outputs[i - 1]];
  shape[0] = t.C;
  shape[1] = t.H;
  shape[2] = t.W;
  return 0;
}

int dif_net_embed(dif_net* h, const void* x_dev, int n, int layout, int dtype, float* out_dev, void* stream) {
  if (!h) return set_error("dif_net_embed: null handle");
  return h->net.embed(x_dev, n, layout, dtype, out_dev, (hipStream_t)stream);
}

double dif_net_flops_per_image(const dif_net* h) { return h ? h->net.flops_per_image() : 0.0; }

int dif_net_launch_count(const dif_net* h) { return h ? (int)h->net.ops.size() : 0; }

int dif_net_op_info(const dif_net* h, int i, const char** name, const char** kernel, double* macs_per_image) {
  if (!h || i < 0 || i >= (int)h->net.ops.size()) return set_error("dif_net_op_info: index out of range");
  const Op& op = h->net.ops[i];
  if (name) *name = op.name.c_str();
  if (kernel) *kernel = h->net.kernel_name(op, h->net.max_batch > 0 ? h->net.max_batch : 1);
  if (macs_per_image) *macs_per_image = op.macs;
  return 0;
}

int dif_net_op_traffic(const dif_net* h, int i, double* act_bytes_per_image, double* param_bytes) {
  if (!h || i < 0 || i >= (int)h->net.ops.size()) return set_error("dif_net_op_traffic: index out of range");
  const Net& net = h->net;
  const Op& op = net.ops[i];
  auto elems = [&](int t) -> double { return t >= 0 ? (double)net.tensors[t].elems() : 0.0; };
  double rd = elems(op.x), wr = elems(op.y) + elems(op.y2), par = 0.0;
  if (op.kind == OP_CONV) {
    // (a sub-sampled first output is stored at a quarter of the size already -- Net::finalize shrank tensors[op.y] --, so the
    // layer's geometry is read from the other output, as Net::run_op does, and elems(op.y) needs no correction: ADVICE r04)
    const TensorDesc& yd = net.tensors[op.y >= 0 && !op.y_sub ? op.y : op.y2];
    // a strided 1x1 layer uses one input pixel per output pixel; every other layer its whole input
    if (op.KH == 1 && op.KW == 1 && op.stride > 1 && op.x >= 0)
      rd = (double)yd.H * yd.W * net.tensors[op.x].C;
    if (op.res >= 0) rd += (double)yd.H * yd.W * yd.C;
    par = (double)op.KH * op.KW * op.Cin_true * op.Cout + 2.0 * op.Cout;
  } else if (op.kind == OP_GDCTAIL || op.kind == OP_DWFULL || op.kind == OP_DWCONV) {
    par = op.macs > 0 && op.kind == OP_GDCTAIL ? op.macs : 0.0;    // its three weight matrices = its MACs per image
  }
  if (act_bytes_per_image) *act_bytes_per_image = 4.0 * (rd + wr);
  if (param_bytes) *param_bytes = 4.0 * par;
  return 0;
}

int dif_net_embed_profile(dif_net* h, const void* x_dev, int n, int layout, int dtype, float* out_dev,
                          void* stream, float* ms_host) {
  if (!h || !ms_host) return set_error("dif_net_embed_profile: null argument");
  return h->net.embed(x_dev, n, layout, dtype, out_dev, (hipStream_t)stream, ms_host);
}

// ------------------------------------------------------------------ ArcMargin
int dif_arcmargin_create(dif_arcmargin** out, int d, int64_t n_classes, float s, float m) {
  if (!out) return set_error("dif_arcmargin_create: null out");
  if (d <= 0 || d % 32 != 0) return set_error("dif_arcmargin_create: d must be a positive multiple of 32");
  if (n_classes <= 0) return set_error("dif_arcmargin_create: n_classes must be positive");
  dif_arcmargin* h = new (std::nothrow) dif_arcmargin();
  if (!h) return set_error("dif_arcmargin_create: out of host memory");
  h->a.d = d;
  h->a.C = n_classes;
  h->a.s = s;
  h->a.m = m;
  if (hipMalloc(&h->a.w, (size_t)n_classes * d * sizeof(float)) != hipSuccess ||
      hipMalloc(&h->a.winv, (size_t)n_classes * sizeof(float)) != hipSuccess) {
    dif_arcmargin_destroy(h);
    return set_error("dif_arcmargin_create: device allocation failed");
  }
  *out = h;
  return 0;
}

int dif_arcmargin_destroy(dif_arcmargin* h) {
  if (!h) return 0;
  if (h->a.w) (void)hipFree(h->a.w);
  if (h->a.winv) (void)hipFree(h->a.winv);
  if (h->a.einv) (void)hipFree(h->a.einv);
  delete h;
  return 0;
}

int dif_arcmargin_set_weight(dif_arcmargin* h, const float* w_dev, void* stream) {
  if (!h || !w_dev) return set_error("dif_arcmargin_set_weight: null argument");
  hipStream_t st = (hipStream_t)stream;
  DIF_HIP(hipMemcpyAsync(h->a.w, w_dev, (size_t)h->a.C * h->a.d * sizeof(float), hipMemcpyDeviceToDevice, st));
  h->a.has_weight = true;
  return arcmargin_prepare(&h->a, st);
}

int dif_arcmargin_logits(dif_arcmargin* h, const float* emb_dev, const int64_t* labels_dev, int n,
                         float* logits_dev, void* stream) {
  if (!h) return set_error("dif_arcmargin_logits: null handle");
  if (!h->a.has_weight) return set_error("dif_arcmargin_logits: class centres were never set");
  if (n < 0) return set_error("dif_arcmargin_logits: negative batch");
  if (n == 0) return 0;
  if (!emb_dev || !logits_dev) return set_error("dif_arcmargin_logits: null pointer");
  return arcmargin_run(&h->a, emb_dev, labels_dev, n, logits_dev, (hipStream_t)stream);
}

}  // extern "C"
