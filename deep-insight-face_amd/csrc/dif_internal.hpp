// Internal declarations shared by the translation units of libdif.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace dif {

int set_error(const char* fmt, ...);   // records the message for dif_last_error(), returns -1

#define DIF_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return ::dif::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// ---- gallery (owned copy of the rows + per-row norms + match workspace)
struct Gallery {
  int d = 0;
  int64_t n = 0;
  int64_t cap = 0;
  int64_t index_base = 0;   // global index of row 0 (gallery sharded across ranks)
  float* rows = nullptr;    // [n][d]
  float* rows2 = nullptr;   // the same rows as two bf16 planes per K-step of 32, [n][d/32][hi 32 | mid 32] (the filter's operand)
  uint16_t* rows1 = nullptr;// the same rows rounded to bf16 (the one-term filter's operand: "filter" = 2), capacity rounded up to 64 rows:
                            // [n][d] for match_b1_kernel, or -- rows1_frag -- in MFMA-fragment order for match_g1_kernel
  float* probes2 = nullptr; // this call's probes in the filter's form (probes2_cap rows of d floats)
  size_t probes2_cap = 0;
  float* sq = nullptr;      // |g|^2
  float* ninv = nullptr;    // -1/|g|
  // match workspace (csrc/match.hip): per (block, probe) minimum key, candidate count, candidate rows
  float* part_key = nullptr;
  int* part_cnt = nullptr;
  int* part_idx = nullptr;
  size_t part_cap = 0;
  // per probe: error bound of the search key, packed (key, index) winner, its distance, overflow list
  float* eps = nullptr;
  float* eps32 = nullptr;          // ... and of the f32 re-check the finish stage makes of the filter's candidates
  unsigned long long* best = nullptr;
  float* best_dist = nullptr;
  int* flagged = nullptr;
  size_t probe_cap = 0;
  float* hi = nullptr;             // per probe: key above which a row may be anti-parallel beyond -1 (metric 1)
  int* pcls = nullptr;             // per probe: 1 = |q|^2 outside the filter's range, classified by the finish stage
  int* anti_cnt = nullptr;         // per probe: rows whose key reached hi, and up to KANTI of their indices
  int* anti_idx = nullptr;
  int* nflag = nullptr;            // number of probes sent to the exact search in the current call
  unsigned* sqmax_bits = nullptr;  // bits of max |g|^2 over the rows the metric-0 filter sees
  struct GalleryFlags* flags = nullptr;   // rows the filter cannot rank (csrc/match.hip), found by gallery_norms
  bool rows2_valid = false;        // rows2 holds the split of the CURRENT rows
  bool rows2_refused = false;      // its allocation failed for this capacity: the f32 filter serves, no retry per call
  bool rows1_valid = false, rows1_refused = false;   // the same two states for rows1
  int frag = 1;                    // "frag": rows1 in fragment order -- 1 (default) where the embedding size allows it (match.hip: g1_dims) and the
                                   // gallery is large enough to give every wave of match_g1_kernel a few tiles, 2 whatever its size, 0 never
  bool rows1_frag = false;         // the layout rows1 was last filled in
  bool filter_one = true;          // "filter" = 2 (default): one bf16 term per operand (match_b1_kernel), a wider net re-ranked; 1: two terms
  bool filter_bf2 = true;          // the MFMA filter runs on bf16 operands (dif_gallery_set_option "filter" = 0: f32)
  int bd_fill = 1;                 // "bd_fill": blocks of match_bd_kernel per resident slot (development: no effect measured, r04)
  bool no_bd = false;              // dif_gallery_set_option "bd" = 0: the split-bf16 filter stays on match_tile_kernel for every batch
  bool clamp_nan = false;          // report distance 0 / 1 instead of the reference's NaN (dif_gallery_set_option)
};

int gallery_norms(Gallery* g, const float* src, hipStream_t st);   // src: the caller's rows (copied into g->rows in the same pass), or null
int gallery_split_copy(Gallery* g, hipStream_t st);   // (re)builds rows2 when the option asks for it; never fatal
int gallery_update_rows(Gallery* g, const float* src, int64_t first, int64_t count, int64_t old_n, hipStream_t st);   // rows [first, first+count) <- src
int match_run(Gallery* g, const float* probes, int B, int metric, int64_t* idx_out, float* dist_out,
              float* key_out, hipStream_t st);
int pairwise_run(const float* e1, int64_t n1, const float* e2, int64_t n2, int D, int metric, float* out,
                 hipStream_t st);
int match_merge_run(const void* keys, int64_t key_pitch, const void* idx, int64_t idx_pitch, const void* dist,
                    int64_t dist_pitch, int R, int B, int64_t* idx_out, float* dist_out, hipStream_t st);   // pitches in bytes

}  // namespace dif
