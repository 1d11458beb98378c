// C ABI entry points (include/dif.h): error plumbing, distances, gallery + match.
// The embedding network entry points live in net_api.hip.
#include "../../include/dif.h"
#include "dif_internal.hpp"

#include <stdarg.h>
#include <stdio.h>
#include <new>

namespace dif {

static thread_local char g_err[1024] = "";

int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

static int check_metric(int metric) {
  if (metric != DIF_METRIC_SQL2 && metric != DIF_METRIC_COSINE)
    return set_error("Undefined distance metric %d", metric);   // evaluation/utility.py:64
  return 0;
}

}  // namespace dif

using namespace dif;

template <class P>
static int regrow(P** p, size_t keep_bytes, size_t new_bytes, hipStream_t st) {
  P* q = nullptr;
  DIF_HIP(hipMalloc(&q, new_bytes));
  if (*p && keep_bytes) {
    hipError_t e = hipMemcpyAsync(q, *p, keep_bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      (void)hipFree(q);
      return set_error("dif_gallery_reserve: copy failed: %s", hipGetErrorString(e));
    }
  }
  if (*p) DIF_HIP(hipFree(*p));
  *p = q;
  return 0;
}


struct dif_gallery {
  Gallery g;
};

extern "C" {

int dif_version(void) { return DIF_VERSION; }

const char* dif_last_error(void) { return g_err; }

int dif_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int dif_pairwise(const float* e1_dev, int64_t n1, const float* e2_dev, int64_t n2, int d, int metric,
                 float* out_dev, void* stream) {
  if (metric != DIF_METRIC_SIMILARITY && check_metric(metric)) return -1;
  if (d <= 0) return set_error("dif_pairwise: d must be positive (got %d)", d);
  if (n1 < 0 || n2 < 0) return set_error("dif_pairwise: negative row count");
  if (n1 != n2 && n1 != 1 && n2 != 1)
    return set_error("dif_pairwise: shapes (%lld,%d) and (%lld,%d) do not broadcast", (long long)n1, d,
                     (long long)n2, d);
  if (n1 == 0 || n2 == 0) return 0;
  if (!e1_dev || !e2_dev || !out_dev) return set_error("dif_pairwise: null pointer");
  return pairwise_run(e1_dev, n1, e2_dev, n2, d, metric, out_dev, (hipStream_t)stream);
}

int dif_gallery_create(dif_gallery** out, int d) {
  if (!out) return set_error("dif_gallery_create: null out");
  if (d <= 0 || d % 32 != 0)
    return set_error("dif_gallery_create: embedding size must be a positive multiple of 32 (got %d)", d);
  dif_gallery* h = new (std::nothrow) dif_gallery();
  if (!h) return set_error("dif_gallery_create: out of host memory");
  h->g.d = d;
  *out = h;
  return 0;
}

int dif_gallery_destroy(dif_gallery* h) {
  if (!h) return 0;
  Gallery& g = h->g;
  if (g.rows) (void)hipFree(g.rows);
  if (g.rows2) (void)hipFree(g.rows2);
  if (g.rows1) (void)hipFree(g.rows1);
  if (g.probes2) (void)hipFree(g.probes2);
  if (g.sq) (void)hipFree(g.sq);
  if (g.ninv) (void)hipFree(g.ninv);
  for (void* p : {(void*)g.part_key, (void*)g.part_cnt, (void*)g.part_idx, (void*)g.eps, (void*)g.eps32, (void*)g.best,
                  (void*)g.best_dist, (void*)g.flagged, (void*)g.nflag, (void*)g.sqmax_bits, (void*)g.hi,
                  (void*)g.pcls, (void*)g.anti_cnt, (void*)g.anti_idx, (void*)g.flags})
    if (p) (void)hipFree(p);
  delete h;
  return 0;
}

int dif_gallery_set(dif_gallery* h, const float* rows_dev, int64_t n, int64_t index_base, void* stream) {
  if (!h) return set_error("dif_gallery_set: null handle");
  if (n < 0) return set_error("dif_gallery_set: negative row count");
  if (n > 0x7ffffff0LL) return set_error("dif_gallery_set: at most 2^31-16 rows per shard");
  if (n > 0 && !rows_dev) return set_error("dif_gallery_set: null rows");
  Gallery& g = h->g;
  hipStream_t st = (hipStream_t)stream;
  if (n > g.cap) {
    DIF_HIP(hipStreamSynchronize(st));
    if (g.rows) DIF_HIP(hipFree(g.rows));
    if (g.rows2) DIF_HIP(hipFree(g.rows2));
    if (g.rows1) DIF_HIP(hipFree(g.rows1));
    g.rows1 = nullptr;
    if (g.sq) DIF_HIP(hipFree(g.sq));
    if (g.ninv) DIF_HIP(hipFree(g.ninv));
    g.rows = g.rows2 = g.sq = g.ninv = nullptr;
    g.cap = 0;
    g.n = 0;                                               // an allocation failure below leaves an EMPTY gallery, not dangling rows
    g.rows2_refused = g.rows1_refused = false;
    DIF_HIP(hipMalloc(&g.rows, (size_t)n * g.d * sizeof(float)));
    DIF_HIP(hipMalloc(&g.sq, (size_t)n * sizeof(float)));
    DIF_HIP(hipMalloc(&g.ninv, (size_t)n * sizeof(float)));
    g.cap = n;
    // the split-bf16 copy the filter reads (as large as the rows themselves) is allocated by gallery_split_copy,
    // from gallery_norms below or from the first dif_match after "filter" was switched on: NOT fatal when it does
    // not fit -- the f32 filter needs no copy and gives the same answers
  }
  g.n = n;
  g.index_base = index_base;
  g.rows2_valid = g.rows1_valid = false;
  if (n == 0) return 0;
  return gallery_norms(&g, rows_dev, st);                   // one pass: the copy, the norms and the filter's bf16 copy
}

int dif_gallery_reserve(dif_gallery* h, int64_t capacity, void* stream) {
  if (!h) return set_error("dif_gallery_reserve: null handle");
  if (capacity > 0x7ffffff0LL) return set_error("dif_gallery_reserve: at most 2^31-16 rows per shard");
  Gallery& g = h->g;
  if (capacity <= g.cap) return 0;
  hipStream_t st = (hipStream_t)stream;
  DIF_HIP(hipStreamSynchronize(st));
  const size_t c = (size_t)capacity, n = (size_t)g.n, d = (size_t)g.d;
  if (regrow(&g.rows, n * d * 4, c * d * 4, st)) return -1;
  if (regrow(&g.sq, n * 4, c * 4, st)) return -1;
  if (regrow(&g.ninv, n * 4, c * 4, st)) return -1;
  // the filter's copy: moved when it fits, otherwise dropped (the next dif_match rebuilds it, or the f32 filter serves)
  if (g.rows1) {
    uint16_t* q = nullptr;
    if (hipMalloc(&q, (c + 63) / 64 * 64 * d * 2) == hipSuccess) {   // (whole 64-row tiles: either layout of the first n rows lies inside them)
      if (g.rows1_valid && n) {
        DIF_HIP(hipMemcpyAsync(q, g.rows1, (n + 63) / 64 * 64 * d * 2, hipMemcpyDeviceToDevice, st));
        DIF_HIP(hipStreamSynchronize(st));
      }
    } else {
      (void)hipGetLastError();
      g.rows1_valid = false;
    }
    DIF_HIP(hipFree(g.rows1));
    g.rows1 = q;
  }
  if (g.rows2) {
    float* q = nullptr;
    if (hipMalloc(&q, c * d * 4) == hipSuccess) {
      if (g.rows2_valid && n) {
        DIF_HIP(hipMemcpyAsync(q, g.rows2, n * d * 4, hipMemcpyDeviceToDevice, st));
        DIF_HIP(hipStreamSynchronize(st));
      }
    } else {
      (void)hipGetLastError();
      g.rows2_valid = false;
    }
    DIF_HIP(hipFree(g.rows2));
    g.rows2 = q;
  }
  g.rows1_refused = g.rows2_refused = false;
  g.cap = capacity;
  return 0;
}

int dif_gallery_update(dif_gallery* h, const float* rows_dev, int64_t n, int64_t first_row, void* stream) {
  if (!h) return set_error("dif_gallery_update: null handle");
  if (n < 0 || first_row < 0) return set_error("dif_gallery_update: negative row count or position");
  if (n > 0 && !rows_dev) return set_error("dif_gallery_update: null rows");
  Gallery& g = h->g;
  if (first_row > g.n)
    return set_error("dif_gallery_update: first_row %lld would leave a gap behind the gallery's %lld rows", (long long)first_row,
                     (long long)g.n);
  if (first_row + n > g.cap)
    return set_error("dif_gallery_update: rows [%lld, %lld) exceed the capacity of %lld rows (dif_gallery_reserve grows it)",
                     (long long)first_row, (long long)(first_row + n), (long long)g.cap);
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t old_n = g.n;
  if (first_row + n > g.n) g.n = first_row + n;
  return gallery_update_rows(&g, rows_dev, first_row, n, old_n, st);
}

int64_t dif_gallery_size(const dif_gallery* h) { return h ? h->g.n : 0; }
int64_t dif_gallery_capacity(const dif_gallery* h) { return h ? h->g.cap : 0; }

static const char* const kGalleryOptions[] = {"filter", "frag", "clamp_nan", "bd", "bd_fill", nullptr};

const char* dif_gallery_option_name(int i) {
  int n = 0;
  while (kGalleryOptions[n]) ++n;
  return i >= 0 && i < n ? kGalleryOptions[i] : nullptr;
}

int dif_gallery_set_option(dif_gallery* h, const char* key, int value) {
  if (!h || !key) return set_error("dif_gallery_set_option: null argument");
  {
    bool known = false;
    for (const char* const* t = kGalleryOptions; *t; ++t) known |= std::string(*t) == key;
    if (!known) return set_error("dif_gallery_set_option: unknown key '%s'", key);
  }
  if (std::string(key) == "clamp_nan") {
    h->g.clamp_nan = value != 0;
    return 0;
  }
  if (std::string(key) == "bd_fill") {
    if (value < 1 || value > 16) return set_error("dif_gallery_set_option: 'bd_fill' takes 1..16");
    h->g.bd_fill = value;
    return 0;
  }
  if (std::string(key) == "bd") {
    h->g.no_bd = value == 0;
    return 0;
  }
  if (std::string(key) == "frag") {
    // the one-term filter's copy in MFMA-fragment order (match_g1_kernel) or row-major (match_b1_kernel): same answers.
    // The copy is rewritten in the other layout by the next dif_match (or dif_gallery_set).
    if (value < 0 || value > 2) return set_error("dif_gallery_set_option: 'frag' takes 0, 1 or 2");
    h->g.frag = value;
    return 0;
  }
  if (std::string(key) == "filter") {
    // 0: f32 MFMA on the rows themselves; 1: two-term split-bf16 copy; 2: one-term bf16 copy (a wider net, re-ranked)
    if (value < 0 || value > 2) return set_error("dif_gallery_set_option: 'filter' takes 0, 1 or 2");
    Gallery& g = h->g;
    g.filter_bf2 = value != 0;
    g.filter_one = value == 2;
    const bool drop2 = g.rows2 && value != 1, drop1 = g.rows1 && value != 2;
    if (drop1 || drop2) {                      // give back the copy the chosen filter does not read
      DIF_HIP(hipDeviceSynchronize());         // a dif_match in flight may still read it
      if (drop2) {
        DIF_HIP(hipFree(g.rows2));
        g.rows2 = nullptr;
        g.rows2_valid = false;
      }
      if (drop1) {
        DIF_HIP(hipFree(g.rows1));
        g.rows1 = nullptr;
        g.rows1_valid = false;
      }
    }
    if (value) g.rows2_refused = g.rows1_refused = false;   // switched (back) on: the next dif_gallery_set / dif_match builds the copy
    return 0;
  }
  return set_error("dif_gallery_set_option: unknown key '%s'", key);
}

int dif_gallery_get_stat(dif_gallery* h, const char* key, int64_t* out, void* stream) {
  if (!h || !key || !out) return set_error("dif_gallery_get_stat: null argument");
  Gallery& g = h->g;
  const std::string k(key);
  if (k == "split_copy") {
    *out = ((g.rows2 && g.rows2_valid) || (g.rows1 && g.rows1_valid)) ? 1 : 0;
    return 0;
  }
  if (k == "filter_terms") {                   // what the next dif_match's filter stage runs on: 0 f32 rows, 2 / 1 bf16 terms per operand
    *out = (g.filter_bf2 && g.filter_one && g.rows1 && g.rows1_valid) ? 1 : ((g.filter_bf2 && g.rows2 && g.rows2_valid) ? 2 : 0);
    return 0;
  }
  if (k == "frag_copy") {                      // 1: the one-term copy is held in fragment order (the next dif_match runs match_g1_kernel)
    *out = (g.filter_bf2 && g.filter_one && g.rows1 && g.rows1_valid && g.rows1_frag) ? 1 : 0;
    return 0;
  }
  if (k == "row_bytes") {                      // device bytes held per gallery row
    *out = (int64_t)g.d * 4 * (g.rows2 ? 2 : 1) + (g.rows1 ? (int64_t)g.d * 2 : 0) + 8;
    return 0;
  }
  if (k == "exact_probes") {                   // probes the last dif_match sent to the exact whole-gallery search
    int v = 0;
    if (g.nflag) {
      DIF_HIP(hipMemcpyAsync(&v, g.nflag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
      DIF_HIP(hipStreamSynchronize((hipStream_t)stream));
    }
    *out = v;
    return 0;
  }
  return set_error("dif_gallery_get_stat: unknown key '%s'", key);
}

int dif_match(dif_gallery* h, const float* probes_dev, int n, int metric, int64_t* idx_out_dev,
              float* dist_out_dev, float* key_out_dev, void* stream) {
  if (!h) return set_error("dif_match: null handle");
  if (check_metric(metric)) return -1;
  if (n < 0) return set_error("dif_match: negative probe count");
  if (n == 0) return 0;
  if (!probes_dev || !idx_out_dev || !dist_out_dev) return set_error("dif_match: null pointer");
  return match_run(&h->g, probes_dev, n, metric, idx_out_dev, dist_out_dev, key_out_dev, (hipStream_t)stream);
}

int dif_match_merge(const float* keys_dev, const int64_t* idx_dev, const float* dist_dev, int R, int n,
                    int64_t* idx_out_dev, float* dist_out_dev, void* stream) {
  if (R <= 0 || n < 0) return set_error("dif_match_merge: bad sizes R=%d n=%d", R, n);
  if (n == 0) return 0;
  if (!keys_dev || !idx_dev || !dist_dev || !idx_out_dev || !dist_out_dev)
    return set_error("dif_match_merge: null pointer");
  return match_merge_run(keys_dev, (int64_t)n * 4, idx_dev, (int64_t)n * 8, dist_dev, (int64_t)n * 4, R, n, idx_out_dev,
                         dist_out_dev, (hipStream_t)stream);
}

int dif_match_merge_packed(const void* packed_dev, int R, int n, int64_t* idx_out_dev, float* dist_out_dev,
                           void* stream) {
  if (R <= 0 || n < 0) return set_error("dif_match_merge_packed: bad sizes R=%d n=%d", R, n);
  if (n == 0) return 0;
  if (!packed_dev || !idx_out_dev || !dist_out_dev) return set_error("dif_match_merge_packed: null pointer");
  const char* base = static_cast<const char*>(packed_dev);
  const int64_t pitch = (int64_t)n * 16;     // per rank: key[n] f32 | dist[n] f32 | idx[n] i64
  return match_merge_run(base, pitch, base + (size_t)n * 8, pitch, base + (size_t)n * 4, pitch, R, n, idx_out_dev,
                         dist_out_dev, (hipStream_t)stream);
}

}  // extern "C"
