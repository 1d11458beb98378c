#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace dif {
struct ArcMargin {
  int d = 0;
  int64_t C = 0;
  float s = 64.f, m = 0.5f;
  float* w = nullptr;      // [C][d] owned copy
  float* winv = nullptr;   // 1/|w_c|
  float* einv = nullptr;   // 1/|e_b| workspace
  int einv_cap = 0;
  bool has_weight = false;
};
int arcmargin_prepare(ArcMargin* a, hipStream_t st);
int arcmargin_run(ArcMargin* a, const float* emb, const int64_t* labels, int B, float* logits, hipStream_t st);
}  // namespace dif
