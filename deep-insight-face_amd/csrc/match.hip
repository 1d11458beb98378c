// 1:N gallery match and the small distance kernels.
// hipcc-flags: -ffp-contract=off
// (no implicit a*b+c fusion anywhere in this file: stage 2 below restates the reference's float32
// operations one by one; the search key uses explicit fmaf where a fused multiply-add is meant)
//
// Semantics (deep_insight_face/evaluation/utility.py:52-66 broadcast over gallery rows
// + np.argmin, see SURVEY.md section 3 "1:N gallery search"):
//   metric 0: d(q, g) = sum((q - g)^2)                      -> argmin
//   metric 1: d(q, g) = arccos(q.g / (|q| |g|)) / pi        -> argmin
// The B x G distance matrix is never materialised.  Three stages:
//
//  1. match_tile_kernel forms dot(q, g) tiles on the f32 MFMA (gallery rows on the M side, probes on
//     the lanes) and turns each dot into a monotone SEARCH KEY in the epilogue
//        metric 1: key = -dot / |g|        (|q| is constant per probe; arccos is decreasing)
//        metric 0: key = |g|^2 - 2 dot     (|q|^2 is constant per probe)
//     The key is only a filter: it carries the MFMA chain's rounding (and, for metric 0, the
//     cancellation of |g|^2 - 2 dot), so near-ties can order differently from the reference's
//     float32 distances.  Every block therefore keeps, per probe, its running minimum key AND the
//     indices of all rows whose key lies within a proven error bound `eps` of it (up to KCAND of
//     them; a list is restarted whenever the minimum drops by more than eps).
//  2. match_finish_kernel takes the global minimum key per probe, gathers the candidates of every
//     block within eps of it and RE-RANKS them with the reference's own arithmetic: the float32
//     operations of utility.py:54-62 in NumPy's evaluation order (products rounded, then NumPy's
//     pairwise summation: 8 interleaved accumulators per block of <= 128, blocks combined by a
//     balanced tree -- restated in np_sum below and pinned bit-for-bit by the golden vectors), so the
//     similarity / squared distance of a candidate is BIT-IDENTICAL to the reference's.  The winner
//     is the lowest index among the minimal float32 distances -- np.argmin -- including NumPy's
//     NaN rule (rounding can push a similarity above 1; arccos then gives NaN and np.argmin returns
//     the first NaN).  arccos itself is evaluated in double and rounded once (NumPy dispatches its
//     float32 arccos to SVML on AVX-512 hosts and to libm elsewhere, <= 2 ulp apart -- measured --
//     so the reference's own tie pattern below that resolution is host-dependent).
//  3. A probe whose candidate list overflowed anywhere (more than KCAND near-minimal rows in one
//     block: only adversarial near-duplicate galleries) is re-searched EXACTLY by
//     match_exact_kernel: the reference distance of every gallery row, packed (key, index) minimum.
//     The kernel is always launched and exits at once when no probe was flagged.
#include "gemm_core.hpp"
#include "dif_internal.hpp"

#include <stdlib.h>

namespace dif {

constexpr int KCAND = 8;        // near-minimal rows remembered per (block, probe)
constexpr int CAND_MAX = 512;   // candidates re-ranked per probe by the finish kernel
constexpr int NP_SCRATCH = 72;  // LDS floats a wave needs for np_sum: 64 leaf sums + the combine stack

// float <-> unsigned with the same ordering (atomicMin on keys)
__device__ __forceinline__ unsigned ord_enc(float f) {
  const unsigned u = __builtin_bit_cast(unsigned, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(unsigned e) {
  const unsigned u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
  return __builtin_bit_cast(float, u);
}

// ---------------------------------------------------------------------------------------------
// NumPy's float32 add.reduce over a contiguous axis (numpy/_core/src/umath/loops_utils.h.src,
// @TYPE@_pairwise_sum; NumPy 2.2.6, the pinned interpreter of this image):
//   n < 8:     res = 0; res += a[i] in order
//   n <= 128:  r[j] = a[j] (j < 8); r[j] += a[i + j] for i = 8, 16, ... < n - n % 8;
//              res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)); then res += a[i] for the rest
//   else:      n2 = n / 2, n2 -= n2 % 8;  sum(a, n2) + sum(a + n2, n - n2)
// The plan flattens the recursion: leaves in order, and after leaf l `pops[l]` additions of the two
// topmost partial sums (a stack machine).  tests/test_oracle_golden.py checks this restatement against
// np.sum bit for bit; tests/test_match_gpu.py checks the device against NumPy.
struct SumPlan {
  int nleaf;
  unsigned short start[64], len[64];
  unsigned char pops[64];
};

static void plan_rec(int off, int n, SumPlan& p) {
  if (n <= 128 || p.nleaf >= 63) {
    p.start[p.nleaf] = (unsigned short)off;
    p.len[p.nleaf] = (unsigned short)n;
    p.pops[p.nleaf] = 0;
    ++p.nleaf;
    return;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  plan_rec(off, n2, p);
  plan_rec(off + n2, n - n2, p);
  ++p.pops[p.nleaf - 1];
}

int make_sum_plan(int d, SumPlan* out) {
  if (d <= 0 || d > 64 * 128) return set_error("distance: embedding size %d outside [1, 8192]", d);
  SumPlan p;
  p.nleaf = 0;
  plan_rec(0, d, p);
  *out = p;
  return 0;
}

// One wave evaluates sum_k term(k) in NumPy's order; every lane returns the result.  Eight lanes
// share a leaf (one accumulator each), so eight leaves run per pass.  `scratch` = NP_SCRATCH floats of
// LDS private to the wave.
template <class F>
__device__ __forceinline__ float np_sum(const SumPlan& plan, float* scratch, int lane, F term) {
#pragma clang fp contract(off)
  const int grp = lane >> 3, j = lane & 7;
  for (int l0 = 0; l0 < plan.nleaf; l0 += 8) {
    const int l = l0 + grp;
    float res = 0.f;
    if (l < plan.nleaf) {
      const int st = plan.start[l], n = plan.len[l];
      if (n < 8) {
        for (int i = 0; i < n; ++i) res = res + term(st + i);
      } else {
        const int body = n - (n % 8);
        float r = term(st + j);
        for (int i = 8; i < body; i += 8) r = r + term(st + i + j);
        // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)): float addition commutes, so the butterfly gives every
        // lane of the group exactly that value
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        res = r;
        for (int i = body; i < n; ++i) res = res + term(st + i);
      }
    }
    if (j == 0 && l < plan.nleaf) scratch[l] = res;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // the stack machine, run by every lane on the same values (the stack lives in scratch[64..71]:
  // depth <= log2(64) + 1; identical stores to one address are benign)
  float* stack = scratch + 64;
  int sp = 0;
  float top = 0.f;
#pragma unroll 1
  for (int l = 0; l < plan.nleaf; ++l) {
    top = scratch[l];
    for (int c = 0; c < plan.pops[l]; ++c) {
      --sp;
      top = stack[sp] + top;
    }
    stack[sp] = top;
    ++sp;
  }
  __builtin_amdgcn_wave_barrier();
  return top;
}

// The reference's float32 distance of one pair of rows (utility.py:54-62), evaluated by one wave.
//   metric 0: np.sum(np.square(np.subtract(a, b)), 1)
//   metric 1: np.arccos(np.sum(a*b, 1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))) / math.pi
// Returns the ranking key: the distance itself, or -inf where the reference's arccos is NaN
// (|similarity| > 1 by rounding, or 0/0): np.argmin ranks NaN before every number.  `dist` receives
// the reported distance: the similarity is clamped to [-1, 1] first (DESIGN.md "NaN at s > 1").
// `sim_out` (optional) receives the similarity of metric 1, bit-identical to the reference's.
__device__ __forceinline__ float ref_distance(const SumPlan& plan, float* scratch, const float* a, const float* b,
                                              int metric, int lane, float* dist, float* sim_out = nullptr) {
#pragma clang fp contract(off)
  if (metric == 0) {
    const float s = np_sum(plan, scratch, lane, [&](int k) {
      const float d = a[k] - b[k];
      return d * d;
    });
    *dist = s;
    return s != s ? -__builtin_inff() : s;
  }
  const float dot = np_sum(plan, scratch, lane, [&](int k) { return a[k] * b[k]; });
  const float aa = np_sum(plan, scratch, lane, [&](int k) { return a[k] * a[k]; });
  const float bb = np_sum(plan, scratch, lane, [&](int k) { return b[k] * b[k]; });
  const float norm = __builtin_sqrtf(aa) * __builtin_sqrtf(bb);
  const float s = dot / norm;
  if (sim_out) *sim_out = s;
  constexpr float PI_F = 3.14159274101257324f;     // float32(math.pi): NumPy divides a float32 array by it in float32
  const bool is_nan = !(s >= -1.f && s <= 1.f);
  const float sc = is_nan ? (s > 1.f ? 1.f : (s < -1.f ? -1.f : s)) : s;
  const float d = (float)acos((double)sc) / PI_F;
  *dist = d;
  return is_nan ? -__builtin_inff() : d;
}

// ---------------------------------------------------------------------------------------------
// Per-probe bound on |search key - exact key| (both directions, plus the width over which the
// reference's float32 distances can collapse): rows outside it cannot be the reference's arg-min.
//   u = 2^-24.  MFMA dot: a 512-term fma chain, |err| <= D u sum|q_k g_k| <= D u |q| |g|.
//   metric 1: key = -dot/|g|            -> E = (D + 8) u |q|;            eps = 2 E + 2.5e-6 |q|
//   metric 0: key = |g|^2 - 2 dot       -> E = u (18 gmax^2 + (2 D + 2) |q| gmax);
//             reference: pairwise sum of (q-g)^2, ties within 24 u (|q| + gmax)^2  -> eps = 2 E + that
// (gmax = the longest gallery row).  Deliberately generous: a wider net only costs re-ranked rows.
__global__ __launch_bounds__(256) void probe_eps_kernel(const float* __restrict__ probes, int B, int D, int metric,
                                                        const unsigned* __restrict__ sqmax_bits,
                                                        float* __restrict__ eps) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= B) return;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float x = probes[(int64_t)p * D + k];
    s = fmaf(x, x, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const float u = 5.9604645e-8f, qn = sqrtf(s) * 1.0001f;
    float e;
    if (metric == 1) {
      e = (2.f * (D + 8) * u + 2.5e-6f) * qn;
    } else {
      const float gmax = sqrtf(__builtin_bit_cast(float, *sqmax_bits)) * 1.0001f;
      e = 2.f * u * (18.f * gmax * gmax + (2.f * D + 2.f) * qn * gmax) + 24.f * u * (qn + gmax) * (qn + gmax);
    }
    eps[p] = e;
  }
}

template <class T>
__global__ __launch_bounds__(T::NT, 2) void match_tile_kernel(const float* __restrict__ gallery, int64_t G,
                                                            const float* __restrict__ probes, int B, int D,
                                                            const float* __restrict__ aux, int metric,
                                                            const float* __restrict__ eps,
                                                            float* __restrict__ part_key,
                                                            int* __restrict__ part_cnt,
                                                            int* __restrict__ part_idx) {
  constexpr int WM = T::WM, WN = T::WN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // candidate state of the block, behind the mainloop's staging buffers
  unsigned* s_min = reinterpret_cast<unsigned*>(smem + T::LDS_FLOATS);   // [BN] running minimum key (ordered bits)
  int* s_cnt = reinterpret_cast<int*>(s_min + T::BN);                     // [BN] rows appended since the last restart
  int* s_idx = s_cnt + T::BN;                                             // [BN][KCAND]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int p0 = blockIdx.y * T::BN;
  const int ksteps = D / BK;
  const int64_t gtiles = (G + T::BM - 1) / T::BM;

  for (int c = tid; c < T::BN; c += T::NT) {
    s_min[c] = ord_enc(__builtin_inff());
    s_cnt[c] = 0;
  }
  float bmin[WN], ep[WN];
  int col[WN];
#pragma unroll
  for (int n = 0; n < WN; ++n) {
    bmin[n] = __builtin_inff();
    col[n] = (wc * WN + n) * 32 + (lane & 31);
    ep[n] = (p0 + col[n] < B) ? eps[p0 + col[n]] : 0.f;
  }
  // (the first use of s_min / s_cnt comes after a mainloop, i.e. after several barriers)

  for (int64_t gt = blockIdx.x; gt < gtiles; gt += gridDim.x) {
    const int64_t g0 = gt * T::BM;
    f32x16 acc[WM][WN];
    zero_acc<T>(acc);
    RowLoader<T::NA, T::RP> al(gallery + g0 * D, G - g0, D);
    RowLoader<T::NB, T::RP> bl(probes + (int64_t)p0 * D, (int64_t)B - p0, D);
    gemm_mainloop<T>(al, bl, 0, ksteps, smem, acc);

    // aux[g] = -1/|g| (metric 1) or |g|^2 (metric 0); rows past G read 0 through the
    // descriptor and are masked out of the search.
    const int64_t rows_left = G - g0;
    const __amdgpu_buffer_rsrc_t arsrc =
        make_rsrc(aux + g0, (uint32_t)((rows_left < T::BM ? rows_left : T::BM) * 4));
    // dots -> keys, in place (rows past G: +inf)
    float tmin[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) tmin[n] = __builtin_inff();
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      const int rbase = (wr * WM + m) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // rows rbase + 8q .. +3 <-> registers 4q .. 4q+3
        const f32x4 ax = buf_load4(arsrc, (uint32_t)(rbase + 8 * q) * 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = rbase + 8 * q + j < rows_left;
#pragma unroll
          for (int n = 0; n < WN; ++n) {
            const float dot = acc[m][n][4 * q + j];
            float key = (metric == 1) ? dot * ax[j] : fmaf(-2.f, dot, ax[j]);
            key = ok ? key : __builtin_inff();
            acc[m][n][4 * q + j] = key;
            tmin[n] = fminf(tmin[n], key);
          }
        }
      }
    }
    // lanes l and l+32 and the WGM waves stacked on M share probe columns: block minimum through LDS
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      tmin[n] = fminf(tmin[n], __shfl_xor(tmin[n], 32));
      if (lane < 32 && tmin[n] < bmin[n]) atomicMin(&s_min[col[n]], ord_enc(tmin[n]));
    }
    __syncthreads();
    float cur[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      cur[n] = ord_dec(s_min[col[n]]);
      // the minimum dropped by more than eps: nothing remembered so far can still be within eps of it
      if (wr == 0 && lane < 32 && cur[n] < bmin[n] - ep[n]) s_cnt[col[n]] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      const int rbase = (wr * WM + m) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = rbase + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int n = 0; n < WN; ++n) {
          if (acc[m][n][r] <= cur[n] + ep[n]) {          // (+inf never passes: cur is finite once a row was seen)
            const int pos = atomicAdd(&s_cnt[col[n]], 1);
            if (pos < KCAND) s_idx[col[n] * KCAND + pos] = (int)g0 + rl;
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < WN; ++n) bmin[n] = cur[n];
  }
  __syncthreads();
  for (int c = tid; c < T::BN; c += T::NT) {
    const int p = p0 + c;
    if (p < B) {
      const int64_t o = (int64_t)blockIdx.x * B + p;
      const int cnt = s_cnt[c];
      part_key[o] = ord_dec(s_min[c]);
      part_cnt[o] = cnt;
      for (int i = 0; i < KCAND && i < cnt; ++i) part_idx[o * KCAND + i] = s_idx[c * KCAND + i];
    }
  }
}

__device__ __forceinline__ unsigned long long pack_key_idx(float key, int idx) {
  return ((unsigned long long)ord_enc(key) << 32) | (unsigned)idx;
}

// one wave per probe: global minimum key, candidates within eps of it, exact re-rank
__global__ __launch_bounds__(64) void match_finish_kernel(const float* __restrict__ part_key,
                                                          const int* __restrict__ part_cnt,
                                                          const int* __restrict__ part_idx, int nparts, int B,
                                                          const float* __restrict__ probes,
                                                          const float* __restrict__ gallery, int D, int metric,
                                                          const float* __restrict__ eps, const SumPlan plan,
                                                          unsigned long long* __restrict__ best,
                                                          float* __restrict__ best_dist,
                                                          int* __restrict__ nflag, int* __restrict__ flagged) {
  __shared__ float scratch[NP_SCRATCH];
  __shared__ int cand[CAND_MAX];
  __shared__ int ncand, overflow;
  const int p = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) {
    ncand = 0;
    overflow = 0;
  }
  float gmin = __builtin_inff();
  for (int t = lane; t < nparts; t += 64) gmin = fminf(gmin, part_key[(int64_t)t * B + p]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gmin = fminf(gmin, __shfl_xor(gmin, o));
  __syncthreads();
  const float lim = gmin + eps[p];
  for (int t = lane; t < nparts; t += 64) {
    const int64_t o = (int64_t)t * B + p;
    if (part_key[o] <= lim) {
      const int cnt = part_cnt[o];
      if (cnt > KCAND) overflow = 1;
      const int take = cnt < KCAND ? cnt : KCAND;
      const int pos = atomicAdd(&ncand, take);
      for (int i = 0; i < take; ++i) {
        if (pos + i < CAND_MAX) cand[pos + i] = part_idx[o * KCAND + i];
        else overflow = 1;
      }
    }
  }
  __syncthreads();
  const int n = ncand < CAND_MAX ? ncand : CAND_MAX;
  unsigned long long bk = ~0ull;
  float bd = __builtin_nanf("");
  const float* q = probes + (int64_t)p * D;
  for (int c = 0; c < n; ++c) {
    const int gi = cand[c];
    float d;
    const float key = ref_distance(plan, scratch, q, gallery + (int64_t)gi * D, metric, lane, &d);
    const unsigned long long pk = pack_key_idx(key, gi);
    if (pk < bk) {
      bk = pk;
      bd = d;
    }
  }
  if (lane == 0) {
    best[p] = bk;
    best_dist[p] = bd;
    if (overflow || n == 0) flagged[atomicAdd(nflag, 1)] = p;
  }
}

// Exact search for the flagged probes: the reference distance of EVERY gallery row (one wave per
// row at a time), packed (key, index) minimum.  Exits at once when nothing was flagged.
__global__ __launch_bounds__(256) void match_exact_kernel(const int* __restrict__ nflag,
                                                          const int* __restrict__ flagged,
                                                          const float* __restrict__ probes,
                                                          const float* __restrict__ gallery, int64_t G, int D,
                                                          int metric, const SumPlan plan,
                                                          unsigned long long* __restrict__ best) {
  __shared__ float scratch[4][NP_SCRATCH];
  const int nf = *nflag;
  if (nf == 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + wave, wstride = (int64_t)gridDim.x * 4;
  for (int f = 0; f < nf; ++f) {
    const int p = flagged[f];
    const float* q = probes + (int64_t)p * D;
    unsigned long long bk = ~0ull;
    for (int64_t g = w0; g < G; g += wstride) {
      float d;
      const float key = ref_distance(plan, scratch[wave], q, gallery + g * D, metric, lane, &d);
      const unsigned long long pk = pack_key_idx(key, (int)g);
      bk = pk < bk ? pk : bk;
    }
    if (lane == 0 && bk != ~0ull) atomicMin(&best[p], bk);
  }
}

// unpack (key, index); the distance of a row the exact search moved to is recomputed here
__global__ __launch_bounds__(64) void match_output_kernel(const unsigned long long* __restrict__ best,
                                                          const float* __restrict__ best_dist,
                                                          const int* __restrict__ nflag, const float* __restrict__ probes,
                                                          const float* __restrict__ gallery, int D, int metric,
                                                          const SumPlan plan, int64_t index_base,
                                                          int64_t* __restrict__ idx_out, float* __restrict__ dist_out,
                                                          float* __restrict__ key_out) {
  __shared__ float scratch[NP_SCRATCH];
  const int p = blockIdx.x, lane = threadIdx.x;
  const unsigned long long bk = best[p];
  const int gi = (int)(unsigned)(bk & 0xffffffffu);
  float d = best_dist[p];
  if (*nflag != 0 && bk != ~0ull)
    (void)ref_distance(plan, scratch, probes + (int64_t)p * D, gallery + (int64_t)gi * D, metric, lane, &d);
  if (lane == 0) {
    const bool found = bk != ~0ull;
    idx_out[p] = found ? index_base + gi : -1;
    dist_out[p] = found ? d : __builtin_nanf("");
    // the reference distance itself (-inf where the reference has NaN): comparable across gallery shards
    if (key_out) key_out[p] = found ? ord_dec((unsigned)(bk >> 32)) : __builtin_inff();
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// |g|^2 and -1/|g| per gallery row (search-key ingredients; fma chain), and the longest row; one wave per row.
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ rows, int64_t n, int D,
                                                        float* __restrict__ sq, float* __restrict__ ninv,
                                                        unsigned* __restrict__ sqmax_bits) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float x = rows[r * D + k];
    s = fmaf(x, x, s);
  }
  s = wave_sum(s);
  if (lane == 0) {
    sq[r] = s;
    ninv[r] = -1.f / sqrtf(s);
    if (s == s) atomicMax(sqmax_bits, __builtin_bit_cast(unsigned, s));   // s >= 0: float order == bit order
  }
}

// Row-paired distance (evaluation/utility.py:52-66); rows of e1 / e2 broadcast when n1 / n2 == 1.
// metric 0 / 1 as the reference (metric 0 bit-identical; metric 1 up to arccos, see ref_distance);
// metric 2 = the cosine similarity itself (common/losses.py:39-40 precedent), bit-identical.
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ e1, const float* __restrict__ e2,
                                                       int64_t n, int64_t n1, int64_t n2, int D, int metric,
                                                       const SumPlan plan, float* __restrict__ out) {
  __shared__ float scratch[4][NP_SCRATCH];
  const int wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const float* a = e1 + (n1 == 1 ? 0 : r) * D;
  const float* b = e2 + (n2 == 1 ? 0 : r) * D;
  float d, sim = 0.f;
  (void)ref_distance(plan, scratch[wave], a, b, metric == 2 ? 1 : metric, lane, &d, &sim);
  if (lane == 0) out[r] = metric == 2 ? sim : d;
}

// Merge R per-shard results: keys are reference distances (comparable across shards), lowest key, then
// lowest global index == np.argmin over the whole gallery.  pk / pi / pd = bytes between two shards'
// rows of each array (the three may be slices of one packed all-gather buffer).
__global__ __launch_bounds__(256) void match_merge_kernel(const char* __restrict__ keys, int64_t pk,
                                                          const char* __restrict__ idx, int64_t pi,
                                                          const char* __restrict__ dist, int64_t pd, int R, int B,
                                                          int64_t* __restrict__ idx_out,
                                                          float* __restrict__ dist_out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= B) return;
  float bk = __builtin_inff();
  int64_t bi = -1;
  float bd = __builtin_nanf("");
  for (int r = 0; r < R; ++r) {
    const float k = reinterpret_cast<const float*>(keys + r * pk)[p];
    const int64_t i = reinterpret_cast<const int64_t*>(idx + r * pi)[p];
    if (i < 0) continue;
    if (bi < 0 || k < bk || (k == bk && i < bi)) {
      bk = k;
      bi = i;
      bd = reinterpret_cast<const float*>(dist + r * pd)[p];
    }
  }
  idx_out[p] = bi;
  dist_out[p] = bd;
}

}  // namespace dif

using namespace dif;

namespace dif {

template <class K>
static int allow_lds(K kern, int bytes) {
  static bool done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!done[dev]) {
    DIF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[dev] = true;
  }
  return 0;
}

template <class T>
static int launch_match_tiles(const Gallery* g, const float* probes, int B, int metric, int nparts,
                              hipStream_t st) {
  auto kern = match_tile_kernel<T>;
  constexpr int lds = T::LDS_BYTES + T::BN * (2 + KCAND) * 4;
  if (allow_lds(kern, lds)) return -1;
  dim3 grid(nparts, (B + T::BN - 1) / T::BN);
  hipLaunchKernelGGL(kern, grid, dim3(T::NT), lds, st, g->rows, g->n, probes, B, g->d,
                     metric == 1 ? g->ninv : g->sq, metric, g->eps, g->part_key, g->part_cnt, g->part_idx);
  DIF_HIP(hipGetLastError());
  return 0;
}

static int match_tile_kind(int B) {
  // 0: 128 gallery rows x 128 probes, 1: 128 x 64, 3: 64 x 64, 4: 128 x 32 (four waves stacked on M)
  static const int forced = getenv("DIF_MATCH_TILE") ? atoi(getenv("DIF_MATCH_TILE")) : -1;
  if (forced >= 0) return forced;
  // unlike the convolutions, the 128-row tiles win here (the arg-min epilogue and the probe
  // re-reads weigh more on small tiles: 117 vs 99 TF at 256 x 1M, 128 vs 120 TF at 4096 x 125k)
  // B <= 32: the 32-probe tile halves the padded MFMA work, which at 1M rows would otherwise
  // (67 GFLOP for 64 padded probes) cost more than streaming the gallery from HBM
  if (B <= 32) return 4;
  return B <= 64 ? 1 : 0;
}

int match_plan_parts(const Gallery* g, int B) {
  // One gallery tile per block while the partial buffers stay small; beyond that the blocks
  // stride over tiles (the imbalance of a long stride is negligible).
  const int kind = match_tile_kind(B);
  const int BM = kind == 3 ? 64 : 128;   // kinds 0, 1 and 4 all take 128 gallery rows per tile
  const int BN = kind == 0 ? 128 : (kind == 4 ? 32 : 64);
  const int64_t gtiles = (g->n + BM - 1) / BM;
  const int64_t colblocks = (B + BN - 1) / BN;
  int64_t cap = (int64_t)(96 << 20) / ((int64_t)B * (8 + 4 * KCAND));   // <= 96 MiB of (key, count, candidates)
  const int64_t fill = (2048 + colblocks - 1) / colblocks;             // enough blocks to fill the chip twice over
  if (cap < fill) cap = fill;
  int64_t want = gtiles < cap ? gtiles : cap;
  if (want < 1) want = 1;
  return (int)want;
}

template <class P>
static int grow(P** p, size_t* cap, size_t need, size_t elem) {
  if (need <= *cap) return 0;
  if (*p) DIF_HIP(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  DIF_HIP(hipMalloc(p, need * elem));
  *cap = need;
  return 0;
}

int match_run(Gallery* g, const float* probes, int B, int metric, int64_t* idx_out, float* dist_out,
              float* key_out, hipStream_t st) {
  if (g->n <= 0) return set_error("dif_match: gallery is empty");
  if (B <= 0) return 0;
  const int nparts = match_plan_parts(g, B);
  const size_t need = (size_t)nparts * B;
  if (need > g->part_cap || (size_t)B > g->probe_cap) {
    DIF_HIP(hipStreamSynchronize(st));
    size_t c1 = g->part_cap, c2 = g->part_cap, c3 = g->part_cap;
    if (grow(&g->part_key, &c1, need, sizeof(float))) return -1;
    if (grow(&g->part_cnt, &c2, need, sizeof(int))) return -1;
    if (grow(&g->part_idx, &c3, need, sizeof(int) * KCAND)) return -1;
    g->part_cap = c1;
    size_t p1 = g->probe_cap, p2 = g->probe_cap, p3 = g->probe_cap, p4 = g->probe_cap;
    if (grow(&g->eps, &p1, (size_t)B, sizeof(float))) return -1;
    if (grow(&g->best, &p2, (size_t)B, sizeof(unsigned long long))) return -1;
    if (grow(&g->best_dist, &p3, (size_t)B, sizeof(float))) return -1;
    if (grow(&g->flagged, &p4, (size_t)B, sizeof(int))) return -1;
    g->probe_cap = p1;
  }
  SumPlan plan;
  if (make_sum_plan(g->d, &plan)) return -1;
  DIF_HIP(hipMemsetAsync(g->nflag, 0, sizeof(int), st));
  hipLaunchKernelGGL(probe_eps_kernel, dim3((B + 3) / 4), dim3(256), 0, st, probes, B, g->d, metric, g->sqmax_bits,
                     g->eps);
  DIF_HIP(hipGetLastError());
  int rc;
  const int kind = match_tile_kind(B);
  if (kind == 3)
    rc = launch_match_tiles<Tile<1, 1>>(g, probes, B, metric, nparts, st);
  else if (kind == 4 && B <= 32)
    rc = launch_match_tiles<Tile<1, 1, 4, 1>>(g, probes, B, metric, nparts, st);
  else if (kind == 1 || B <= 64)
    rc = launch_match_tiles<Tile<2, 1>>(g, probes, B, metric, nparts, st);
  else
    rc = launch_match_tiles<Tile<2, 2>>(g, probes, B, metric, nparts, st);
  if (rc) return rc;
  hipLaunchKernelGGL(match_finish_kernel, dim3(B), dim3(64), 0, st, g->part_key, g->part_cnt, g->part_idx, nparts, B,
                     probes, g->rows, g->d, metric, g->eps, plan, g->best, g->best_dist, g->nflag, g->flagged);
  DIF_HIP(hipGetLastError());
  hipLaunchKernelGGL(match_exact_kernel, dim3(1024), dim3(256), 0, st, g->nflag, g->flagged, probes, g->rows, g->n,
                     g->d, metric, plan, g->best);
  DIF_HIP(hipGetLastError());
  hipLaunchKernelGGL(match_output_kernel, dim3(B), dim3(64), 0, st, g->best, g->best_dist, g->nflag, probes, g->rows,
                     g->d, metric, plan, g->index_base, idx_out, dist_out, key_out);
  DIF_HIP(hipGetLastError());
  return 0;
}

int gallery_norms(Gallery* g, hipStream_t st) {
  if (!g->nflag) {
    DIF_HIP(hipMalloc(&g->nflag, sizeof(int)));
    DIF_HIP(hipMalloc(&g->sqmax_bits, sizeof(unsigned)));
  }
  DIF_HIP(hipMemsetAsync(g->sqmax_bits, 0, sizeof(unsigned), st));
  if (g->n == 0) return 0;
  hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)((g->n + 3) / 4)), dim3(256), 0, st, g->rows, g->n, g->d,
                     g->sq, g->ninv, g->sqmax_bits);
  DIF_HIP(hipGetLastError());
  return 0;
}

int pairwise_run(const float* e1, int64_t n1, const float* e2, int64_t n2, int D, int metric, float* out,
                 hipStream_t st) {
  const int64_t n = n1 > n2 ? n1 : n2;
  if (n == 0) return 0;
  SumPlan plan;
  if (make_sum_plan(D, &plan)) return -1;
  hipLaunchKernelGGL(pairwise_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, e1, e2, n, n1, n2, D,
                     metric, plan, out);
  DIF_HIP(hipGetLastError());
  return 0;
}

int match_merge_run(const void* keys, int64_t key_pitch, const void* idx, int64_t idx_pitch, const void* dist,
                    int64_t dist_pitch, int R, int B, int64_t* idx_out, float* dist_out, hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(match_merge_kernel, dim3((B + 255) / 256), dim3(256), 0, st, static_cast<const char*>(keys),
                     key_pitch, static_cast<const char*>(idx), idx_pitch, static_cast<const char*>(dist), dist_pitch, R,
                     B, idx_out, dist_out);
  DIF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dif
