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
//   The accumulator tile is staged through LDS so that residual loads and both stores are
//   16 bytes per lane along the channel axis (whole 512-byte rows per 32 lanes).
//
//   Scheduling: a persistent "stream-K" grid.  The (tile, K-step) iteration space is cut
//   into P equal contiguous ranges, one per resident block, so every CU gets the same
//   number of MFMA K-steps whatever the tile count (784 tiles of 128x128 over 512 resident
//   blocks would otherwise run 2 rounds for 1.53 rounds of work).  A tile whose K range is
//   cut is finished by the block that holds its first K-step: the other block(s) store
//   their partial accumulators to a per-block slab and raise a flag
//   (agent-scope release / acquire, cdna_hip_programming.md Guideline 16); the split is a
//   pure function of the problem size, so results are run-to-run deterministic.
//
// The zero halo and every tile tail come from buffer-descriptor range checks (an
// out-of-range offset loads zeros), so the loaders are branch-free.
#include "gemm_core.hpp"
#include <string>
#include "dif_internal.hpp"
#include "ops.hpp"

#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

namespace dif {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// A-operand loader: gathers BM output pixels x 32 k-values per step.
// KMODE 0: K layout decided at run time (any Cin % 4 == 0); KMODE 2: compile-time
// specialisation for Cin % 32 == 0 with the channel-block-major K order (the 3x3 layers)
template <int N, int RP, bool PRE, int KMODE = 0>
struct ConvALoader {
  __amdgpu_buffer_rsrc_t rsrc;
  const float* ps;   // pre-activation scale / shift per input channel (PRE only)
  const float* pt;
  int pre_act;
  f32x4 cs, ct;      // this K-step's scale / shift for the thread's 4 channels
  unsigned okmask;   // bit i: row i's chunk of this K-step is inside the image
  int32_t base[N];   // byte offset of (n - n_first, hi0, wi0, 0) relative to the tile's first image (may be < 0)
  int32_t hw0[N];    // hi0 in the high 16 bits, wi0 in the low 16 bits (biased by 0x4000 each)
  int H, W, Cin, KW, taps;
  FastDiv fd_cin, fd_kw, fd_taps;
  int k_order;
  bool fast;         // Cin % 32 == 0: one (kh, kw) per K-step, block-uniform
  static constexpr int T_MAX_BM = 256;

  __device__ __forceinline__ ConvALoader(const ConvArgs& a, int m0) {
    const int tid = threadIdx.x;
    H = a.H;
    W = a.W;
    Cin = a.Cin;
    KW = a.KW;
    taps = a.KH * a.KW;
    fast = (a.Cin % BK) == 0;
    ps = a.pre_scale;
    pt = a.pre_shift;
    pre_act = a.pre_act;
    okmask = 0;
    const int HoWo = a.Ho * a.Wo;
    fd_cin = a.fd_cin;
    fd_kw = a.fd_kw;
    fd_taps = a.fd_taps;
    k_order = a.k_order;
    const int n_first = a.fd_howo.div(m0);
    const int64_t img_elems = (int64_t)a.H * a.W * a.Cin;
    const int64_t imgs_left = a.N - n_first;
    int64_t span = (T_MAX_BM + HoWo - 1) / HoWo + 1;     // images a tile can touch
    if (span > imgs_left) span = imgs_left;
    rsrc = make_rsrc(a.x + n_first * img_elems, (uint32_t)(span * img_elems * 4));
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int m = m0 + (tid >> 3) + RP * i;
      if (m < a.M) {
        int n, r, ho, wo;
        a.fd_howo.divmod(m, n, r);
        a.fd_wo.divmod(r, ho, wo);
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

  __device__ __forceinline__ void load(int kstep, f32x4 (&r)[N]) {
    int kh, kw, toff, cch;
    bool tap_ok = true;
    if (KMODE == 2 || fast) {
      int tap, ci0;
      if (KMODE == 2 || k_order == 1) {
        int cblk;
        fd_taps.divmod(kstep, cblk, tap);
        ci0 = cblk * BK;
      } else {
        fd_cin.divmod(kstep * BK, tap, ci0);
      }
      fd_kw.divmod(tap, kh, kw);
      toff = ((kh * W + kw) * Cin + ci0) * 4;
      cch = ci0 + (threadIdx.x & 7) * 4;
    } else {
      int tap, ci;
      fd_cin.divmod(kstep * BK + (threadIdx.x & 7) * 4, tap, ci);
      fd_kw.divmod(tap, kh, kw);
      tap_ok = tap < taps;
      toff = ((kh * W + kw) * Cin + ci) * 4 - (threadIdx.x & 7) * 16;
      cch = tap_ok ? ci : 0;
    }
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int hi = (hw0[i] >> 16) - 0x4000 + kh;
      const int wi = (hw0[i] & 0xffff) - 0x4000 + kw;
      const bool ok = tap_ok && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const uint32_t off = ok ? (uint32_t)(base[i] + toff) : OOB;
      r[i] = buf_load4(rsrc, off);
      mask |= ok ? (1u << i) : 0u;
    }
    if constexpr (PRE) {
      okmask = mask;
      cs = *reinterpret_cast<const f32x4*>(ps + cch);
      ct = *reinterpret_cast<const f32x4*>(pt + cch);
    }
  }

  // pre-activation of the gathered chunk, applied just before it is written to LDS
  __device__ __forceinline__ void finish(f32x4 (&r)[N]) const {
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const bool ok = (okmask >> i) & 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = fmaf(r[i][j], cs[j], ct[j]);
          if (pre_act == ACT_RELU) v = fmaxf(v, 0.f);
          r[i][j] = ok ? v : 0.f;
        }
      }
    }
  }
};

// Pointwise gather: 1x1 kernel, no padding, Cin % 32 == 0 (any stride).  A row of the GEMM is
// one input pixel's channel vector, so a K-step is just the next 128 bytes of every row: no
// tap arithmetic, no bounds tests in the loop (rows past M get an offset beyond the
// descriptor's range once, in the constructor).
template <int N, int RP, bool PRE>
struct ConvPwLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t base[N];
  const float* ps;
  const float* pt;
  int pre_act;
  f32x4 cs, ct;
  static constexpr uint32_t INVALID = 0x80000000u;   // stays out of range after adding any K offset

  __device__ __forceinline__ ConvPwLoader(const ConvArgs& a, int m0) {
    const int tid = threadIdx.x;
    ps = a.pre_scale;
    pt = a.pre_shift;
    pre_act = a.pre_act;
    const int HoWo = a.Ho * a.Wo;
    const int n_first = a.fd_howo.div(m0);
    const int64_t img_elems = (int64_t)a.H * a.W * a.Cin;
    const int64_t imgs_left = a.N - n_first;
    int64_t span = (256 + HoWo - 1) / HoWo + 1;
    if (span > imgs_left) span = imgs_left;
    rsrc = make_rsrc(a.x + n_first * img_elems, (uint32_t)(span * img_elems * 4));
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int m = m0 + (tid >> 3) + RP * i;
      if (m < a.M) {
        int n, r, ho, wo;
        a.fd_howo.divmod(m, n, r);
        a.fd_wo.divmod(r, ho, wo);
        base[i] = (uint32_t)((((int64_t)(n - n_first) * a.H + ho * a.stride) * a.W + wo * a.stride) * a.Cin * 4) +
                  (tid & 7) * 16;
      } else {
        base[i] = INVALID;
      }
    }
  }

  __device__ __forceinline__ void load(int kstep, f32x4 (&r)[N]) {
    const uint32_t o = (uint32_t)kstep * (BK * 4);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = buf_load4(rsrc, base[i] + o);
    if constexpr (PRE) {
      const int cch = kstep * BK + (threadIdx.x & 7) * 4;
      cs = *reinterpret_cast<const f32x4*>(ps + cch);
      ct = *reinterpret_cast<const f32x4*>(pt + cch);
    }
  }

  __device__ __forceinline__ void finish(f32x4 (&r)[N]) const {
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const bool ok = base[i] != INVALID;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = fmaf(r[i][j], cs[j], ct[j]);
          if (pre_act == ACT_RELU) v = fmaxf(v, 0.f);
          r[i][j] = ok ? v : 0.f;
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------
// Halo-resident A operand for 3x3 / stride 1 / pad 1 layers ("patch" path, AM = 3).
//
// The general gather above re-fetches and re-stages the A tile on EVERY K-step, although the nine taps of a
// 32-channel slice read (shifted copies of) the same input pixels: per slice, 9 x 64 rows x 128 B pass through
// the vector memory path, the staging registers and the LDS write port.  Measured on IResNet-100's 3x3 layers
// (DIF_CONV_DBG=32 ablation, profiles/r02_ablation.txt): that costs 11 % of the layer time, and it is where the
// 2.4x excess of fabric traffic over the compulsory bytes came from.
//
// Here the tile's input pixels -- its own 64 pixels plus the halo above, below and beside them -- are laid in
// LDS ONCE per 32-channel slice, in padded image coordinates
//        P(n, h, w) = (n * (H + 1) + h + 1) * (W + 2) + (w + 1)
// (one zero column each side of a row, one zero row shared by consecutive images), so that the operand row of
// output pixel i for tap (kh, kw) is simply entry  base_i + kh * (W + 2) + kw:  a tap is an ADDRESS OFFSET.
// Entries are 128 bytes (32 floats) without padding; bank conflicts of the fragment reads are removed by the XOR
// swizzle chunk' = chunk ^ ((entry >> 1) & 7) (cdna_hip_programming.md rule 21), applied when an entry is written
// and when it is read.  The B tile uses the same unpadded swizzled rows, so a block needs
// PATCH_EMAX * 128 + 2 * 64 * 128 = 37.9 KB of LDS: four blocks per CU, as before.
// The next slice's patch is fetched into registers while taps 4..8 of the current one run, and written after a
// barrier at the slice boundary (one extra barrier per nine K-steps).
// entries a 64-pixel tile may need (host-checked bound): two instantiations -- 128 entries cover maps up to 14 wide
// (16 prefetch registers, no spills), 168 entries 28-wide maps (24 registers, 7 spilled: still +3.7 % on those layers)
constexpr int PATCH_EMAX_S = 128, PATCH_EMAX_L = 168;
constexpr int PATCH_EMAX_X = 256;   // conv_tn_kernel only (its LDS is set by the epilogue's 33 KB staging tile): maps up to 59 wide
constexpr int patch_lds_bytes(int emax) { return emax * 128 + 2 * 64 * 128; }
constexpr int PATCH_PF_TAP = 4;                       // tap of the current slice at which the next patch is requested

__device__ __forceinline__ int patch_swz(int entry) { return (entry >> 1) & 7; }
// 8x8-tile form (AM = 6): image and first output pixel (h0, w0) of tile m0 / 64; returns that pixel's linear index
__device__ __forceinline__ int tile2d_pix0(const ConvArgs& a, int m0, int& n, int& h0, int& w0) {
  int r, hb, wb;
  a.fd_t2_img.divmod(m0 >> 6, n, r);
  a.fd_t2_w.divmod(r, hb, wb);
  h0 = hb * 8;
  w0 = wb * 8;
  return (n * a.H + h0) * a.W + w0;
}

// T: the tile (BM output pixels, NT threads); EMAX: entries of the LDS patch
template <class T, int EMAX_>
struct PatchA {
  static constexpr int EMAX = EMAX_;
  static constexpr int NPC = (EMAX * 8 + T::NT - 1) / T::NT;   // 16-byte chunks per thread per patch
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t goff[NPC];   // byte offset of this thread's chunk j inside channel slice 0, or OOB
  int base[T::WM];      // entry of this lane's output pixels (row lane & 31 of each of the wave's 32-row tiles), tap (0, 0)
  int WP;
  __device__ __forceinline__ PatchA(const ConvArgs& a, int m0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int HW = a.H * a.W;
    WP = a.W + 2;
    const int RPI = a.H + 1;
    int n_first, r0;
    a.fd_howo.divmod(m0, n_first, r0);
    const int64_t img_elems = (int64_t)HW * a.Cin;
    const int64_t imgs_left = a.N - n_first;
    int64_t span = (T::BM + HW - 1) / HW + 1;
    if (span > imgs_left) span = imgs_left;
    rsrc = make_rsrc(a.x + n_first * img_elems, (uint32_t)(span * img_elems * 4));
    int h0, w0;
    a.fd_wo.divmod(r0, h0, w0);
    const int p_first = (h0 + 1) * WP + (w0 + 1);      // P(m0), images counted from n_first
    const int origin = p_first - WP - 1;                // tap (0, 0) of the first pixel
#pragma unroll
    for (int mi = 0; mi < T::WM; ++mi) {
      int m = m0 + (T::wave_row() * T::WM + mi) * 32 + (lane & 31);
      if (m >= a.M) m = m0;
      int n, r, h, w;
      a.fd_howo.divmod(m, n, r);
      a.fd_wo.divmod(r, h, w);
      base[mi] = ((n - n_first) * RPI + h + 1) * WP + (w + 1) - p_first;
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = tid + T::NT * j;
      const int e = slot >> 3, q = slot & 7;
      const int c = origin + e;                          // padded linear coordinate
      int row, col, img, hp;
      a.fd_wp.divmod(c, row, col);
      a.fd_rpi.divmod(row, img, hp);
      const bool ok = e < EMAX && hp >= 1 && col >= 1 && col <= a.W && (n_first + img) < a.N;
      const int chunk = q ^ patch_swz(e);                // logical 4-channel chunk that lives in this physical slot
      goff[j] = ok ? (uint32_t)(((img * a.H + (hp - 1)) * a.W + (col - 1)) * a.Cin * 4 + chunk * 16) : OOB;
    }
  }
  __device__ __forceinline__ void load(int cblk, f32x4 (&r)[NPC]) const {
#pragma unroll
    for (int j = 0; j < NPC; ++j) r[j] = buf_load4(rsrc, goff[j] == OOB ? OOB : goff[j] + (uint32_t)cblk * 128u);
  }
  // f32 patch: 128-byte entries, slot-linear (the swizzle is on the source side)
  __device__ __forceinline__ void store(float* patch, const f32x4 (&r)[NPC]) const {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = threadIdx.x + T::NT * j;
      if (slot < EMAX * 8) *reinterpret_cast<f32x4*>(patch + slot * 4) = r[j];
    }
  }
};

// The same operand for maps whose height and width are multiples of 8 (the 56x56 and 112x112 stages): the 64 output
// pixels of a tile are an 8x8 SQUARE of one image (AM = 6), so the patch is 10x10 = 100 entries whatever the map's
// width -- a linear 64-pixel run on a 56-wide map would need 245.  Entry (py, px) = input pixel (h0 - 1 + py,
// w0 - 1 + px); tile row r = output pixel (h0 + r / 8, w0 + r % 8); tap (kh, kw) = entry offset kh * 10 + kw.
template <class T>
struct PatchA2D {
  static constexpr int EMAX = 100, SIDE = 10;
  static constexpr int NPC = (EMAX * 8 + T::NT - 1) / T::NT;
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t goff[NPC];
  int base[T::WM];
  int WP;
  // first output pixel (linear index) of tile m0 / 64
  __device__ __forceinline__ PatchA2D(const ConvArgs& a, int m0) {
    static_assert(T::WM == 1 && T::BM == 64, "2-D patch: 64-pixel tile, one 32-row fragment per wave");
    const int tid = threadIdx.x, lane = tid & 63;
    WP = SIDE;
    int n, h0, w0;
    tile2d_pix0(a, m0, n, h0, w0);
    const int64_t img_elems = (int64_t)a.H * a.W * a.Cin;
    rsrc = make_rsrc(a.x + n * img_elems, (uint32_t)(img_elems * 4));
    const int r = T::wave_row() * 32 + (lane & 31);
    base[0] = (r >> 3) * SIDE + (r & 7);
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = tid + T::NT * j;
      const int e = slot >> 3, q = slot & 7;
      const int py = (e * 205) >> 11, px = e - py * SIDE;      // e / 10 for e < 1024
      const int hi = h0 - 1 + py, wi = w0 - 1 + px;
      const bool ok = e < EMAX && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const int chunk = q ^ patch_swz(e);
      goff[j] = ok ? (uint32_t)((hi * a.W + wi) * a.Cin * 4 + chunk * 16) : OOB;
    }
  }
  __device__ __forceinline__ void load(int cblk, f32x4 (&r)[NPC]) const {
#pragma unroll
    for (int j = 0; j < NPC; ++j) r[j] = buf_load4(rsrc, goff[j] == OOB ? OOB : goff[j] + (uint32_t)cblk * 128u);
  }
  __device__ __forceinline__ void store(float* patch, const f32x4 (&r)[NPC]) const {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = threadIdx.x + T::NT * j;
      if (slot < EMAX * 8) *reinterpret_cast<f32x4*>(patch + slot * 4) = r[j];
    }
  }
};

// Mainloop of the patch path: K-step ks = (channel slice ks / 9, tap ks % 9) -- the channel-block-major K order.
// 64x64 tile, 4 waves (Tile<1,1,2,2>).  Ends on a barrier.
template <class T, class PA, class BLoader, class Tail>
__device__ __forceinline__ void gemm_mainloop_patch(const PA& pa, BLoader& bl, int kbeg, int kend, float* lds,
                                                    f32x16 (&acc)[1][1], Tail&& tail) {
  static_assert(T::BM == 64 && T::BN == 64 && T::NT == 256, "patch path: 64x64 tile");
  constexpr int NB = T::NB, RP = T::RP;
  float* patch = lds;
  float* bimg = lds + PA::EMAX * 32;                     // two B images of 64 rows x 32 floats
  const int tid = threadIdx.x, lane = tid & 63;
  const int wc = T::wave_col();
  const int h = lane >> 5;
  // B staging slot (row tid>>3 + RP*i, logical chunk tid&7) and fragment row
  const int rb_row = wc * 32 + (lane & 31);
  const int rb_swz = patch_swz(rb_row);
  const float* pb0 = bimg + rb_row * 32;

  f32x4 pr[PA::NPC], rb[NB];
  int cb = kbeg / 9, tap = kbeg - cb * 9;
  pa.load(cb, pr);
  bl.load(kbeg, rb);
  auto stage_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = (tid >> 3) + RP * i;
      *reinterpret_cast<f32x4*>(bimg + buf * (64 * 32) + row * 32 + (((tid & 7) ^ patch_swz(row)) << 2)) = rb[i];
    }
  };
  pa.store(patch, pr);
  stage_b(0);
  lds_barrier();
  bool pf_issued = false;
  auto mfma_step = [&](int cur) {
    const int kh = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0);
    const int e = pa.base[0] + kh * pa.WP + (tap - 3 * kh);
    const int sa = patch_swz(e);
    const float* pae = patch + e * 32;
    const float* pbe = pb0 + cur * (64 * 32);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int l0 = 2 * h + 4 * s;                        // logical chunks l0, l0 + 1 (k = 16 s + 8 h .. + 7)
      const f32x4 fa0 = *reinterpret_cast<const f32x4*>(pae + ((l0 ^ sa) << 2));
      const f32x4 fa1 = *reinterpret_cast<const f32x4*>(pae + (((l0 + 1) ^ sa) << 2));
      const f32x4 fb0 = *reinterpret_cast<const f32x4*>(pbe + ((l0 ^ rb_swz) << 2));
      const f32x4 fb1 = *reinterpret_cast<const f32x4*>(pbe + (((l0 + 1) ^ rb_swz) << 2));
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc[0][0], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc[0][0], 0, 0, 0);
    }
  };
  int ks = kbeg;
  for (; ks + 1 < kend; ++ks) {
    const int cur = (ks - kbeg) & 1;
    bl.load(ks + 1, rb);
    if (!pf_issued && tap >= PATCH_PF_TAP && (cb + 1) * 9 < kend) {
      pa.load(cb + 1, pr);                                 // lands while taps PF_TAP .. 8 run
      pf_issued = true;
    }
    __builtin_amdgcn_sched_barrier(0);                     // the prefetches stay above the MFMAs
    mfma_step(cur);
    stage_b(cur ^ 1);
    if (++tap == 9) {
      tap = 0;
      ++cb;
      lds_barrier();                                     // every wave has read its last fragment of the old patch
      pa.store(patch, pr);     // pf_issued holds here: the slice had a successor, so its taps >= PF_TAP requested it
      pf_issued = false;
    }
    lds_barrier();
  }
  tail();
  mfma_step((ks - kbeg) & 1);
  lds_barrier();
}

// ------------------------------------------------------------------------------------------
// Patch mainloop with the B operand straight from L2 into the MFMA registers ("B-direct", AM = 13 / 15 / 16).
//
// In gemm_mainloop_patch the weights still go global -> registers -> LDS -> registers every K-step, and that LDS
// round trip is what ties a block's four waves to one barrier per K-step (the K-loop ablation of round 1: barrier,
// staging writes and fragment reads cost 7-11 points each).  The weight matrix is launch-invariant, so net.hip lays
// it out once in FRAGMENT order (ConvArgs::w_frag): for a 32-column tile and a K-step, the four 16-byte pieces a
// lane feeds to the 16 MFMAs are four contiguous KB, lane l at byte 16 l -- a wave fetches its B fragments with four
// fully coalesced buffer_load_dwordx4, half a K-step (eight MFMAs) ahead, into registers that go to the MFMAs as they
// are (a whole step ahead costs 16 more registers: 120 spilled).  LDS then
// holds the A patch only, and the block synchronises only when the patch is replaced: two barriers per NINE K-steps
// instead of ten.  (Both waves of a column pair fetch the same KB: the second hits L1/L2; 16 KB per block per step.)
// `pre()` runs before the MFMAs of every K-step, `post()` after them: conv_bdp_kernel retires the previous tile there.
template <class T, class PA, class Tail, class Pre, class Post>
__device__ __forceinline__ void gemm_mainloop_patch_bd(const PA& pa, const ConvArgs& a, int n0, int kbeg, int kend,
                                                       float* lds, f32x16 (&acc)[1][1], Tail&& tail, Pre&& pre, Post&& post) {
  static_assert(T::BM == 64 && T::BN == 64 && T::NT == 256, "patch path: 64x64 tile");
  float* patch = lds;
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  const int KS = a.Kpad / BK;
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w_frag, a.w_frag_bytes);
  // byte offset of this lane's piece (s, u) = q of K-step ks:  ((tile * KS + ks) * 4 + q) * 1024 + 16 lane.  The tile
  // and lane part is the (bounds-checked) vector offset -- a column tile past ceil(Cout / 32) lies beyond the
  // descriptor and reads zero --, the K-step part a scalar offset: no vector ALU work per step for B
  const uint32_t lane_off = (uint32_t)((n0 >> 5) + T::wave_col()) * (uint32_t)KS * 4096u + (uint32_t)lane * 16u;
  // half a K-step (s = 0 or 1: sixteen k, eight MFMAs) of fragments: pieces (s, 0) and (s, 1)
  auto bload = [&](int ks, int s, f32x4 (&b)[2]) {
    const uint32_t so = (uint32_t)ks * 4096u + (uint32_t)s * 2048u;
    b[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_off, so, 0));
    b[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_off + 1024u, so, 0));
  };

  f32x4 pr[PA::NPC], bA[2], bB[2];                         // bA: half s = 0 of the coming step, bB: half s = 1
  int cb = kbeg / 9, tap = kbeg - cb * 9;
  // entry offset of the tap inside the patch, kh * WP + kw, kept as a scalar and stepped (no per-lane tap arithmetic)
  int kw = tap % 3;
  int eoff = (tap / 3) * pa.WP + kw;
  pa.load(cb, pr);
  bload(kbeg, 0, bA);
  pa.store(patch, pr);
  lds_barrier();
  bool pf_issued = false;
  const char* patch_b = reinterpret_cast<const char*>(patch);
  const uint32_t l0x = (uint32_t)(2 * h);                 // logical chunk of half 0, piece 0; the others are l0x | 1, | 4, | 5
  // Byte address of chunk l0x of this lane's entry; chunk (l0x | d) sits at that address XOR 16 d: the swizzle is an XOR
  // of the chunk index, and d touches other bits than l0x.
  auto frag_addr = [&]() -> uint32_t {
    const uint32_t e = (uint32_t)(pa.base[0] + eoff);
    return (e << 7) | ((l0x ^ ((e >> 1) & 7u)) << 4);
  };
  // A fragments are read one half-step AHEAD of their MFMAs, into two register sets: fa = half 0 of the coming step,
  // fn = half 1 of the current one (the fragment of a step's first half is requested under the previous step's second half)
  auto read_frag = [&](uint32_t a0, int s, f32x4 (&f)[2]) {
    f[0] = *reinterpret_cast<const f32x4*>(patch_b + (a0 ^ (uint32_t)(64 * s)));
    f[1] = *reinterpret_cast<const f32x4*>(patch_b + (a0 ^ (uint32_t)(64 * s + 16)));
  };
  auto mfma8 = [&](const f32x4 (&f)[2], const f32x4 (&b)[2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0][t], b[0][t], acc[0][0], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[1][t], b[1][t], acc[0][0], 0, 0, 0);
  };
  f32x4 fa[2], fn[2];
  uint32_t a0 = frag_addr();
  read_frag(a0, 0, fa);
  int ks = kbeg;
  for (; ks + 1 < kend; ++ks) {
    bload(ks, 1, bB);                                      // lands under the first half's MFMAs
    if (!pf_issued && tap >= PATCH_PF_TAP && (cb + 1) * 9 < kend) {
      pa.load(cb + 1, pr);                                 // the next slice's patch: lands while taps PF_TAP .. 8 run
      pf_issued = true;
    }
    pre();
    read_frag(a0, 1, fn);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(fa, bA);
    bload(ks + 1, 0, bA);                                  // lands under the second half's MFMAs
    if (++kw == 3) {
      kw = 0;
      eoff += pa.WP - 2;
    } else {
      ++eoff;
    }
    const bool swap = ++tap == 9;
    if (!swap) {                                           // the next step reads the same patch: its first half now
      a0 = frag_addr();
      read_frag(a0, 0, fa);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma8(fn, bB);
    post();
    if (swap) {
      tap = 0;
      eoff = 0;
      ++cb;
      lds_barrier();                                     // every wave has read its last fragment of the old patch
      pa.store(patch, pr);
      pf_issued = false;
      lds_barrier();
      a0 = frag_addr();
      read_frag(a0, 0, fa);
    }
  }
  bload(ks, 1, bB);
  tail();
  pre();
  read_frag(a0, 1, fn);
  mfma8(fa, bA);
  mfma8(fn, bB);
  post();
  lds_barrier();
}

// ------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") form of the B-direct patch mainloop: the throughput mode's kernel for 3x3 / stride 1 layers.
//
// Arithmetic as gemm_core.hpp's gemm_mainloop_bf3 (every f32 operand = hi + mid + lo in bf16, six products per pair on
// v_mfma_f32_32x32x16_bf16, f32 accumulation, the same product order), data movement as gemm_mainloop_patch_bd:
//   * A: the tile's pixels + halo lie in LDS once per 32-channel slice, ALREADY SPLIT: an entry is [3 planes][32 bf16]
//     (+ 16 bytes: 208-byte entries, so the 16 lanes of a ds_read_b128 group touch 16 distinct 16-byte bank groups).
//     The split costs 22 VALU per four values once per slice (nine K-steps), not once per K-step as in the gather
//     kernel, and a tap is an address offset;
//   * B: the weights, split once at finalize and stored in FRAGMENT order (ConvArgs::w3f:
//     [Cout/32][Kpad/32][s 2][plane 3][lane 64][8 bf16] <- plane(w[32 nt + (lane & 31)][32 ks + 16 s + 8 (lane >> 5) + t])),
//     go from L2 straight into the MFMA operand registers, one 16-k sub-step ahead; no LDS, no barrier per K-step.
// Block = 4 waves, each 64 x 64 (2 x 2 fragments: 64 accumulator registers) -> 128 pixels x 128 channels; two blocks
// per CU = two waves per SIMD with the full 256-register budget.  That budget is NOT enough for the whole kernel
// (profiles/r03_resource_usage.txt: the 128-column forms carry 126-141 spilled VGPRs / 508-568 B of scratch per lane at 256
// VGPRs, the 64-column forms 44-60): the spills sit in set-up, the epilogue and the stream-K fallback copy of the loop; the
// straight-line slice -- the blocks holding MFMAs -- contains no scratch access, which is what matters, because a scratch
// reload counts on the in-order vmcnt and would drain the B ring (the round-2 gather kernel at 128 registers spilled 200-300
// inside its loop).
// Per wave and K-step: 48 MFMAs (1536 matrix-pipe cycles) against 12 ds_read_b128 and 12 coalesced 1 KB loads.
constexpr int BF3P_EB = 208;          // bytes per patch entry
constexpr int BF3P_KSTEP_B = 6144;    // bytes of w3f per (32-column tile, K-step)
// entries a BM-pixel linear tile can need on maps 7 .. 28 wide (host-checked bound): 232 for 128 pixels, 408 for 256
constexpr int bf3p_emax(int bm) { return bm == 128 ? 232 : 408; }
constexpr int bf3p_planes_b(int bm) { return bf3p_emax(bm) * BF3P_EB; }      // the split patch ...
constexpr int bf3p_lds_b(int bm) { return bf3p_emax(bm) * (BF3P_EB + 128); } // ... and, behind it, the next slice's f32 patch as it arrives

// The f32 patch of one 32-channel slice, fetched global -> LDS by LDS-DMA (buffer_load ... lds: no staging registers --
// held in registers across the five taps it is in flight the prefetch cost 32 VGPRs, and every scratch reload of a spilled
// value waits, through the in-order vmcnt, for the B fragments requested before it).  Slot s = (entry s >> 3, 4-channel
// chunk s & 7) is fetched by thread s % 256 into staging byte 16 s (lane-linear per wave instruction); halo and tail slots
// get an out-of-range offset and land as zeros.  Source offsets are recomputed per slice (two mul-hi divisions per slot)
// instead of being kept: eight more registers, or eight scratch reloads in the loop.
template <class T, int EMAX_>
struct PatchDma {
  static constexpr int EMAX = EMAX_;
  static constexpr int NPC = (EMAX * 8 + T::NT - 1) / T::NT;
  __amdgpu_buffer_rsrc_t rsrc;
  int base[T::WM];      // entry of this lane's output pixels, tap (0, 0)
  int WP, origin, H, W, Cin4, imgs;
  FastDiv fd_wp, fd_rpi;
  __device__ __forceinline__ PatchDma(const ConvArgs& a, int m0) {
    const int lane = threadIdx.x & 63;
    const int HW = a.H * a.W;
    WP = a.W + 2;
    H = a.H;
    W = a.W;
    Cin4 = a.Cin * 4;
    fd_wp = a.fd_wp;
    fd_rpi = a.fd_rpi;
    const int RPI = a.H + 1;
    int n_first, r0;
    a.fd_howo.divmod(m0, n_first, r0);
    const int64_t img_elems = (int64_t)HW * a.Cin;
    const int64_t imgs_left = a.N - n_first;
    int64_t span = (T::BM + HW - 1) / HW + 1;
    if (span > imgs_left) span = imgs_left;
    imgs = (int)span;
    rsrc = make_rsrc(a.x + n_first * img_elems, (uint32_t)(span * img_elems * 4));
    int h0, w0;
    a.fd_wo.divmod(r0, h0, w0);
    const int p_first = (h0 + 1) * WP + (w0 + 1);
    origin = p_first - WP - 1;
#pragma unroll
    for (int mi = 0; mi < T::WM; ++mi) {
      int m = m0 + (T::wave_row() * T::WM + mi) * 32 + (lane & 31);
      if (m >= a.M) m = m0;
      int n, r, h, w;
      a.fd_howo.divmod(m, n, r);
      a.fd_wo.divmod(r, h, w);
      base[mi] = ((n - n_first) * RPI + h + 1) * WP + (w + 1) - p_first;
    }
  }
  __device__ __forceinline__ void issue(int cblk, char* staging) const {
    // (scalar: the DMA's LDS addresses go to M0 -- as a vector value the eight destinations were hoisted out of the slice loop,
    // spilled, and every reload, `s_waitcnt vmcnt(0)` behind it, waited for the pieces and B fragments already in flight)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                          // keeps the offsets from being hoisted out of the K loop (and spilled)
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = tid + T::NT * j;
      const int e = slot >> 3, q = slot & 7;
      int row, col, img, hp;
      fd_wp.divmod(origin + e, row, col);
      fd_rpi.divmod(row, img, hp);
      const bool ok = e < EMAX && hp >= 1 && col >= 1 && col <= W && img < imgs;
      const uint32_t off = ok ? (uint32_t)(((img * H + (hp - 1)) * W + (col - 1)) * Cin4 + q * 16 + cblk * 128) : OOB;
      if ((wave * 64 + T::NT * j) < EMAX * 8)             // wave-uniform: whole instructions beyond the patch are skipped
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds_ptr_of(staging + (wave * 64 + T::NT * j) * 16), 16, off, 0, 0, 0);
    }
  }
  // staging (f32, slot-linear) -> the split planes; every thread converts the slots it fetched itself
  template <int NPL = 3>
  __device__ __forceinline__ void convert(const char* staging, char* planes) const {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                          // as in issue(): recompute the slot addresses, do not keep them
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = tid + T::NT * j;
      if (slot < EMAX * 8) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(staging + slot * 16);
        u32x2 hi, mid, lo;
        bf3_split(v, hi, mid, lo);
        char* w = planes + (slot >> 3) * BF3P_EB + (slot & 7) * 8;
        *reinterpret_cast<u32x2*>(w) = hi;
        *reinterpret_cast<u32x2*>(w + 64) = mid;
        if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(w + 128) = lo;
      }
    }
  }
};

// The same operand for maps whose sides are multiples of 8 and too wide for the linear patch (the 56 x 56 and 112 x 112
// stages): the 128-pixel tile is TWO consecutive 8 x 8 tiles of the 8x8-tile order (tile2d_pix0; they need not be neighbours,
// the second may lie in the next row of tiles or in the next image), each with its own 10 x 10 patch: entries 0..99 and
// 100..199.  Wave row wr works on sub-tile wr; tap (kh, kw) = entry offset 10 kh + kw.
template <class T>
struct PatchDma2D {
  static constexpr int EMAX = 200, SIDE = 10;
  static constexpr int NPC = (EMAX * 8 + T::NT - 1) / T::NT;
  static_assert(T::BM == 128 && T::WGM == 2 && T::WM == 2, "two 8x8 sub-tiles, one per wave row");
  __amdgpu_buffer_rsrc_t rsrc;
  int base[T::WM];
  int WP, H, W, Cin4;
  int hw0[2], ioff[2];      // per sub-tile: (h0 - 1) << 16 | (w0 - 1) & 0xffff, and its image's offset in pixels from the first image
  __device__ __forceinline__ PatchDma2D(const ConvArgs& a, int m0) {
    const int lane = threadIdx.x & 63;
    WP = SIDE;
    H = a.H;
    W = a.W;
    Cin4 = a.Cin * 4;
    int n0, h0, w0, n1, h1, w1;
    tile2d_pix0(a, m0, n0, h0, w0);
    const bool second = m0 + 64 < a.M;
    tile2d_pix0(a, second ? m0 + 64 : m0, n1, h1, w1);
    const int64_t img_elems = (int64_t)a.H * a.W * a.Cin;
    rsrc = make_rsrc(a.x + n0 * img_elems, (uint32_t)((n1 - n0 + 1) * img_elems * 4));
    hw0[0] = ((h0 - 1) << 16) | ((w0 - 1) & 0xffff);
    hw0[1] = second ? (((h1 - 1) << 16) | ((w1 - 1) & 0xffff)) : (int)0xc000c000;     // far outside: every entry reads zero
    ioff[0] = 0;
    ioff[1] = (n1 - n0) * a.H * a.W;
#pragma unroll
    for (int mi = 0; mi < T::WM; ++mi) {
      const int r = mi * 32 + (lane & 31);                 // row inside the wave row's sub-tile
      base[mi] = T::wave_row() * 100 + (r >> 3) * SIDE + (r & 7);
    }
  }
  __device__ __forceinline__ void issue(int cblk, char* staging) const {
    // (scalar: the DMA's LDS addresses go to M0 -- as a vector value the eight destinations were hoisted out of the slice loop,
    // spilled, and every reload, `s_waitcnt vmcnt(0)` behind it, waited for the pieces and B fragments already in flight)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                          // recompute the offsets per slice (see PatchDma)
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = tid + T::NT * j;
      const int e = slot >> 3, q = slot & 7;
      const int sub = e >= 100 ? 1 : 0, el = e - 100 * sub;
      const int py = (el * 205) >> 11, px = el - py * SIDE;          // el / 10 for el < 1024
      const int hi = (hw0[sub] >> 16) + py, wi = (int)(short)(hw0[sub] & 0xffff) + px;
      const bool ok = e < EMAX && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const uint32_t off = ok ? (uint32_t)((ioff[sub] + hi * W + wi) * Cin4 + q * 16 + cblk * 128) : OOB;
      if ((wave * 64 + T::NT * j) < EMAX * 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds_ptr_of(staging + (wave * 64 + T::NT * j) * 16), 16, off, 0, 0, 0);
    }
  }
  template <int NPL = 3>
  __device__ __forceinline__ void convert(const char* staging, char* planes) const {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = tid + T::NT * j;
      if (slot < EMAX * 8) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(staging + slot * 16);
        u32x2 hi, mid, lo;
        bf3_split(v, hi, mid, lo);
        char* w = planes + (slot >> 3) * BF3P_EB + (slot & 7) * 8;
        *reinterpret_cast<u32x2*>(w) = hi;
        *reinterpret_cast<u32x2*>(w + 64) = mid;
        if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(w + 128) = lo;
      }
    }
  }
};

// K-steps [9 cbeg, 9 cend): whole 32-channel slices (the kernel cuts its stream-K shares at slice boundaries).  One slice =
// nine taps = 18 sub-steps of 24 MFMAs, written out as straight-line code: the tap is a compile-time constant (an immediate
// address offset), nothing is carried through a branch, and three rings turn at fixed phase --
//   B fragments: three sub-step sets, requested TWO sub-steps (one K-step, >= 1536 matrix cycles) ahead: an L2 miss served
//                by the MALL takes longer than one sub-step under load;
//   A fragments: two sets, read one sub-step ahead;
//   the patch:   the next slice's f32 patch arrives by LDS-DMA under taps 4..8 and is split into the planes at the slice
//                boundary, between two LDS-only barriers (the B ring stays in flight).
// NPL = 3: six products (hi + mid + lo, "bf16x3"); NPL = 2: the two-term tier ("bf16x2": hi + mid, three products mid.hi,
// hi.mid, hi.hi -- half the MFMAs and two thirds of the B bytes; the lo plane is neither written nor read)
// The mainloop itself lives in conv_bf_mainloop.hpp, included once per number of bf16 terms (3: six products, "bf16x3";
// 2: hi + mid only, three products -- half the MFMAs, two thirds of the B bytes -- "bf16x2").
#define BF_NPL_VALUE 3
#define BF_MAINLOOP_NAME gemm_mainloop_patch_bf3
#include "conv_bf_mainloop.hpp"
#undef BF_NPL_VALUE
#undef BF_MAINLOOP_NAME
#define BF_NPL_VALUE 2
#define BF_MAINLOOP_NAME gemm_mainloop_patch_bf2t
#include "conv_bf_mainloop.hpp"
#undef BF_NPL_VALUE
#undef BF_MAINLOOP_NAME
// Each has a kernel of its own: conv_igemm_kernel.hpp is included once per mainloop (conv_igemm_kernel: three terms,
// conv_igemm_bf2_kernel: two), the launcher picks by ConvArgs::bf_terms.  (Compile-time selection INSIDE one kernel was tried in
// four forms -- a template parameter of the function, of a wrapping class, a tag argument, a property of the loader type,
// `if constexpr` at the call: the HOST pass of this compiler rejected each with a reasonless "substitution failure" at the
// call inside the kernel's `run` lambda, while the device pass accepted all of them.  A run-time branch inside one kernel
// -- round 4's first form -- shared one register allocation between the two loops: conv_igemm_kernel.hpp.)

// (plain forwarding functions: the host pass accepts a call to these from the kernel's `run` lambda, not one to the second
// inclusion's mainloop itself)
template <class T, class PA>
__device__ __forceinline__ void bf_mainloop_3(const PA& pa, const ConvArgs& a, int n0, int cbeg, int cend, char* lds,
                                              f32x16 (&acc)[T::WM][T::WN]) {
  gemm_mainloop_patch_bf3<T>(pa, a, n0, cbeg, cend, lds, acc);
}
template <class T, class PA>
__device__ __forceinline__ void bf_mainloop_2(const PA& pa, const ConvArgs& a, int n0, int cbeg, int cend, char* lds,
                                              f32x16 (&acc)[T::WM][T::WN]) {
  gemm_mainloop_patch_bf2t<T>(pa, a, n0, cbeg, cend, lds, acc);
}

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_PRELU) return v >= 0.f ? v : v * alpha;
  if (act == ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}

__device__ __forceinline__ f32x4 load4_or(const float* p, int c, float dflt) {
  if (p) return *reinterpret_cast<const f32x4*>(p + c);
  return f32x4{dflt, dflt, dflt, dflt};
}

// The shortcut tile of one output tile, in the epilogue's thread mapping (16 bytes per lane along
// the channel axis).  Loaded either from the mainloop's tail hook (latency hidden behind the last
// K-step) or at the start of the epilogue.
template <class T>
struct EpiRes {
  static constexpr int CPR = T::BN / 4;          // float4 chunks per tile row
  static constexpr int RPP = T::NT / CPR;        // rows per pass
  static constexpr int ITER = T::BM / RPP;
  f32x4 rv[ITER];
  __device__ __forceinline__ void load(const ConvArgs& a, int m0, int n0) {
    const int tid = threadIdx.x;
    const int c = n0 + (tid % CPR) * 4;
    const int r0 = tid / CPR;
    const bool strided_res = (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo);
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int row = m0 + r0 + i * RPP;
      rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < a.M && c < a.Cout) {
        int64_t ri = row;
        if (strided_res) {
          const int img = row / HoWo;
          const int rr = row - img * HoWo;
          const int ho = rr / a.Wo;
          const int wo = rr - ho * a.Wo;
          ri = ((int64_t)img * a.res_H + (int64_t)ho * a.res_stride) * a.res_W + (int64_t)wo * a.res_stride;
        }
        rv[i] = *reinterpret_cast<const f32x4*>(a.res + ri * a.Cout + c);
      }
    }
  }
};

// Stages the accumulator tile in LDS, then applies the epilogue with 16-byte accesses
// along the channel axis.  Ends on a barrier (LDS is free afterwards).  `er` already holds the
// shortcut tile when `res_loaded`.
// PRELOAD = false (the patch path, which never prefetches the shortcut tile: its registers hold the next patch): the
// shortcut is fetched row by row inside the loop instead of all at once -- that kernel then compiles without spills.
// TILE2D (AM = 6): tile row r is output pixel (h0 + r / 8, w0 + r % 8) of one image instead of pixel m0 + r.
// YSUB = false (the split-bf16 kernels, whose register budget is spent): ConvArgs::y_sub is not honoured (conv_run keeps
// such layers on the f32 kernels).
template <class T, bool PRELOAD = true, bool TILE2D = false, bool YSUB = true>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[T::WM][T::WN], int m0, int n0,
                                              float* smem, EpiRes<T>& er, bool res_loaded) {
  constexpr int WM = T::WM, WN = T::WN;
  constexpr int CS = T::BN + 4;                 // LDS row pitch in floats (16-B aligned, rows shifted by 4 banks)
  constexpr int CPR = T::BN / 4;                // float4 chunks per tile row
  constexpr int RPP = T::NT / CPR;              // rows per pass
  constexpr int ITER = T::BM / RPP;
  static_assert(T::BM * CS <= T::LDS_FLOATS || T::BM > 128, "accumulator tile must fit in the staging LDS (the 256-row split-bf16 tile sizes its LDS itself)");
  static_assert(T::NT % CPR == 0 && T::BM % RPP == 0, "epilogue mapping");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        smem[((wr * WM + m) * 32 + frag_row(lane, r)) * CS + (wc * WN + n) * 32 + (lane & 31)] = acc[m][n][r];
  lds_barrier();

  const int c4 = tid % CPR;
  const int r0 = tid / CPR;
  const int c = n0 + c4 * 4;
  if (c < a.Cout) {
    const bool plain_out = (a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0);
    const f32x4 sc = load4_or(a.scale, c, 1.f), sh = load4_or(a.shift, c, 0.f), al = load4_or(a.alpha, c, 0.f);
    const f32x4 sc2 = load4_or(a.scale2, c, 1.f), sh2 = load4_or(a.shift2, c, 0.f), al2 = load4_or(a.alpha2, c, 0.f);
    if constexpr (PRELOAD) {
      if (a.res && !res_loaded) er.load(a, m0, n0);
    }
    const bool strided_res = (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo);
    int t2_pix0 = 0, t2_pix1 = 0;
    bool t2_second = true;
    if constexpr (TILE2D) {
      int n, h0, w0;
      t2_pix0 = tile2d_pix0(a, m0, n, h0, w0);
      if constexpr (T::BM > 64) {                          // two 8x8 sub-tiles (PatchDma2D)
        t2_second = m0 + 64 < a.M;
        t2_pix1 = tile2d_pix0(a, t2_second ? m0 + 64 : m0, n, h0, w0);
      }
    }
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int rl = r0 + i * RPP;
      const int row = TILE2D ? (rl < 64 ? t2_pix0 : t2_pix1) + ((rl & 63) >> 3) * a.W + (rl & 7) : m0 + rl;
      if (TILE2D ? (rl < 64 || t2_second) : row < a.M) {
        f32x4 rres = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PRELOAD) {
          rres = er.rv[i];
        } else if (a.res) {
          int64_t ri = row;
          if (strided_res) {
            int img, rr, ho, wo;
            a.fd_howo.divmod(row, img, rr);
            a.fd_wo.divmod(rr, ho, wo);
            ri = ((int64_t)img * a.res_H + (int64_t)ho * a.res_stride) * a.res_W + (int64_t)wo * a.res_stride;
          }
          rres = *reinterpret_cast<const f32x4*>(a.res + ri * a.Cout + c);
        }
        const f32x4 av = *reinterpret_cast<const f32x4*>(smem + rl * CS + c4 * 4);
        f32x4 v, v2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = fmaf(av[j], sc[j], sh[j]);
          t = apply_act(t, a.act, al[j]);
          if (a.res) t += rres[j];
          v[j] = t;
          v2[j] = apply_act(fmaf(t, sc2[j], sh2[j]), a.act2, al2[j]);
        }
        int64_t o;
        if (plain_out) {
          o = (int64_t)row * a.y_ld + a.y_coff + c;
        } else {
          int img, rr, ho, wo;
          a.fd_howo.divmod(row, img, rr);
          a.fd_wo.divmod(rr, ho, wo);
          o = (((int64_t)img * a.y_H + ho + a.y_oy) * a.y_W + wo + a.y_ox) * a.y_ld + a.y_coff + c;
        }
        int64_t oy = o;
        bool y_on = a.y != nullptr;
        if (YSUB && a.y_sub) {
          int img, rr, ho, wo;
          a.fd_howo.divmod(row, img, rr);
          a.fd_wo.divmod(rr, ho, wo);
          y_on = y_on && !((ho | wo) & 1);
          oy = (((int64_t)img * ((a.Ho + 1) >> 1) + (ho >> 1)) * ((a.Wo + 1) >> 1) + (wo >> 1)) * a.Cout + c;
        }
        if (a.Cout % 4 == 0) {
          if (y_on) *reinterpret_cast<f32x4*>(a.y + oy) = v;
          if (a.y2) *reinterpret_cast<f32x4*>(a.y2 + o) = v2;
        } else {
          // narrow heads whose channel count is not a multiple of 4 (e.g. 18 = 3*(5+1) YOLO
          // outputs): rows are not 16-byte aligned, store the valid channels one by one
          for (int j = 0; j < 4; ++j)
            if (c + j < a.Cout) {
              if (a.y) a.y[o + j] = v[j];
              if (a.y2) a.y2[o + j] = v2[j];
            }
        }
      }
    }
  }
  lds_barrier();
}

// The common case of conv_epilogue, written lean: plain output geometry (y_ld == Cout, no view, no sub-sampling), Cout % 4 == 0,
// unit-stride shortcut, 32-bit byte offsets (ConvArgs::epi_fast, decided on the host).  Same arithmetic, element by element.
// Why it exists: conv_epilogue serves every layer shape through run-time flags; compiled into the persistent kernels it came
// out as a forest of uniform branches with 64-bit address arithmetic, SGPR spills restored by v_readlane in front of every
// store, and -- fatal -- a VGPR spill reloaded from scratch at the end of every row group: `scratch_load; s_waitcnt vmcnt(0)`
// waits, through the in-order vmcnt, for the stores just issued to be acknowledged, 2-3 us per group of rows.  Block traces
// (net.hip: option dbg = 256) showed 13-17 us of epilogue per 64 x 64 tile against 18 us of mainloop on the 64-channel layers.
// Here every access is a buffer access with a 32-bit offset (out-of-range = dropped / zero replaces each predicate, a null
// tensor gets an empty descriptor), the shortcut rows of a group are requested together, and nothing is carried but the offsets.
template <class T, bool TILE2D, bool YSUB = false>
__device__ __forceinline__ void conv_epilogue_fast(const ConvArgs& a, f32x16 (&acc)[T::WM][T::WN], int m0, int n0,
                                                   float* smem, const EpiRes<T>& er, bool res_loaded) {
  constexpr int WM = T::WM, WN = T::WN;
  constexpr int CS = T::BN + 4, CPR = T::BN / 4, RPP = T::NT / CPR, ITER = T::BM / RPP;
  constexpr int CH = ITER < 4 ? ITER : 4;                   // rows per group
  static_assert(ITER % CH == 0, "epilogue row groups");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        smem[((wr * WM + m) * 32 + frag_row(lane, r)) * CS + (wc * WN + n) * 32 + (lane & 31)] = acc[m][n][r];
  lds_barrier();
  const int c4 = tid % CPR, r0 = tid / CPR;
  const int c = n0 + c4 * 4;
  const bool col_ok = c < a.Cout;
  const uint32_t bytes = (uint32_t)a.M * (uint32_t)a.Cout * 4u;
  const __amdgpu_buffer_rsrc_t y_rsrc = make_rsrc(a.y, (a.y && !YSUB) ? bytes : 0u);
  const __amdgpu_buffer_rsrc_t y2_rsrc = make_rsrc(a.y2, a.y2 ? bytes : 0u);
  const __amdgpu_buffer_rsrc_t res_rsrc = make_rsrc(a.res, a.res ? bytes : 0u);
  const int cc = col_ok ? c : 0;
  const f32x4 sc = load4_or(a.scale, cc, 1.f), sh = load4_or(a.shift, cc, 0.f), al = load4_or(a.alpha, cc, 0.f);
  const f32x4 sc2 = load4_or(a.scale2, cc, 1.f), sh2 = load4_or(a.shift2, cc, 0.f), al2 = load4_or(a.alpha2, cc, 0.f);
  const int act = a.act, act2 = a.act2;
  int t2_pix0 = 0, t2_pix1 = 0;
  bool t2_second = true;
  if constexpr (TILE2D) {
    int n, h0, w0;
    t2_pix0 = tile2d_pix0(a, m0, n, h0, w0);
    if constexpr (T::BM > 64) {                            // the split-bf16 kernel's tile: two 8x8 sub-tiles (PatchDma2D)
      t2_second = m0 + 64 < a.M;
      t2_pix1 = tile2d_pix0(a, t2_second ? m0 + 64 : m0, n, h0, w0);
    }
  }
  // YSUB: the first output keeps its even pixels only, densely (ConvArgs::y_sub); the second output and the shortcut are whole
  const int hs = (a.Ho + 1) >> 1, ws = (a.Wo + 1) >> 1;
  const __amdgpu_buffer_rsrc_t ys_rsrc = make_rsrc(a.y, (a.y && YSUB) ? (uint32_t)a.N * (uint32_t)(hs * ws) * (uint32_t)a.Cout * 4u : 0u);
  const bool has_res = a.res != nullptr;
  const float* srow = smem + r0 * CS + c4 * 4;
#pragma unroll
  for (int i0 = 0; i0 < ITER; i0 += CH) {
    uint32_t voff[CH];
    int prow[CH];
    f32x4 rv[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int rl = r0 + (i0 + j) * RPP;
      const int row = TILE2D ? (rl < 64 ? t2_pix0 : t2_pix1) + ((rl & 63) >> 3) * a.W + (rl & 7) : m0 + rl;
      const bool row_ok = TILE2D ? (rl < 64 || t2_second) : row < a.M;
      voff[j] = (col_ok && row_ok) ? ((uint32_t)row * (uint32_t)a.Cout + (uint32_t)c) * 4u : OOB;
      prow[j] = row;
      rv[j] = res_loaded ? er.rv[i0 + j] : (has_res ? buf_load4(res_rsrc, voff[j]) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(srow + (i0 + j) * RPP * CS);
      f32x4 v, v2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = fmaf(av[e], sc[e], sh[e]);
        const float tr = fmaxf(t, 0.f), tp = t >= 0.f ? t : t * al[e];
        t = act == ACT_RELU ? tr : (act == ACT_PRELU ? tp : (act == ACT_RELU6 ? fminf(tr, 6.f) : t));
        if (has_res) t += rv[j][e];
        v[e] = t;
        float u = fmaf(t, sc2[e], sh2[e]);
        const float ur = fmaxf(u, 0.f), up = u >= 0.f ? u : u * al2[e];
        v2[e] = act2 == ACT_RELU ? ur : (act2 == ACT_PRELU ? up : (act2 == ACT_RELU6 ? fminf(ur, 6.f) : u));
      }
      if constexpr (YSUB) {
        int img, rr, ho, wo;
        a.fd_howo.divmod(voff[j] == OOB ? 0 : prow[j], img, rr);
        a.fd_wo.divmod(rr, ho, wo);
        const uint32_t vs = (voff[j] != OOB && !((ho | wo) & 1))
                                ? ((uint32_t)((img * hs + (ho >> 1)) * ws + (wo >> 1)) * (uint32_t)a.Cout + (uint32_t)c) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ys_rsrc, vs, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rsrc, voff[j], 0, 0);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v2), y2_rsrc, voff[j], 0, 0);
    }
  }
  lds_barrier();
}

// XCD-aware remap of the hardware block id: blocks b, b+8, b+16, ... share an XCD (and its
// L2), so give each XCD one contiguous chunk of the iteration space (speed only).
__device__ __forceinline__ int xcd_remap(int b, int P) {
  const int q = P >> 3, r = P & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

struct NoLoader {};   // B operand of the B-direct kernels: fetched by the mainloop itself

// AM (A-operand gather mode): 0 general, 1 pointwise (1x1, no padding, Cin % 32 == 0),
// 2 multi-tap with Cin % 32 == 0 and channel-block-major K
// BF3 (with AM = 13): the split-bf16 patch kernel, gemm_mainloop_patch_bf3 on the fragment-order split weights a.w3f
// LEAN: the tile is finished by conv_epilogue_fast (the launcher checked its case), otherwise by the general conv_epilogue --
// one of the two per instantiation, not both behind a run-time flag
#define IGEMM_KERNEL_NAME conv_igemm_kernel
#define IGEMM_BF_MAINLOOP bf_mainloop_3
#include "conv_igemm_kernel.hpp"
#undef IGEMM_KERNEL_NAME
#undef IGEMM_BF_MAINLOOP
#define IGEMM_KERNEL_NAME conv_igemm_bf2_kernel
#define IGEMM_BF_MAINLOOP bf_mainloop_2
#include "conv_igemm_kernel.hpp"
#undef IGEMM_KERNEL_NAME
#undef IGEMM_BF_MAINLOOP

#include "conv_splitk.hpp"
#include "conv_minitile.hpp"

// ------------------------------------------------------------------------------------------
// Software-pipelined variant for layers with a SHORT K loop and several tiles per resident block
// (the 1x1 layers of the bottleneck networks, 3x3 layers with 64 input channels).
//
// In conv_igemm_kernel a block's life is prologue -> mainloop -> epilogue; with 2..18 K-steps the
// epilogue (shortcut read + output write, a bandwidth burst when every block reaches it together)
// is a third of that life and the matrix pipes idle through it.  Here a block is persistent, walks
// whole tiles of its XCD's share of the tile space round-robin (no tile is ever split, so results
// do not depend on the schedule; a dynamic queue was tried: on this ISA the atomic's return sits in
// the same in-order vmcnt queue as the operand loads and delays them), keeps the finished tile's
// accumulators in a second register set and
// retires them in eight chunks of two registers DURING the first eight K-steps of the next tile:
// shortcut loads are issued before a step's MFMAs, scale/shift/activation/add/stores after them.
// The epilogue works straight from the MFMA D layout (for register r a lane holds one channel of
// row (r&3) + 8(r>>2) + 4(lane>>5): 32 consecutive channels = 128 contiguous bytes per lane half),
// so it needs no LDS staging and no barrier, and any output channel count or channel-slice view.
// Restrictions (checked by the launcher, which otherwise uses conv_igemm_kernel): 64x64 tile, plain
// spatial output (no padded interior), unit-stride shortcut.
// CPS = chunks of the previous tile retired per K-step (1 or 4).
template <class T, bool PRE, int AM, int CPS>
__global__ __launch_bounds__(T::NT, T::MIN_BLOCKS) void conv_pipe_kernel(const ConvArgs a) {
  static_assert(T::WM == 1 && T::WN == 1 && T::WGM == 2 && T::WGN == 2, "pipelined kernel: 64x64 tile only");
  constexpr int NA = T::NA, NB = T::NB, RP = T::RP;
  constexpr int BUF = (T::BM + T::BN) * LDS_STRIDE, OFFB = T::BM * LDS_STRIDE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  const int KS = a.Kpad / BK;
  const int tiles_n = (a.Cout + T::BN - 1) / T::BN;
  const int ntiles = ((a.M + T::BM - 1) / T::BM) * tiles_n;
  const int st_off = (tid >> 3) * LDS_STRIDE + (tid & 7) * 4;
  const int fr_off = (lane & 31) * LDS_STRIDE + 8 * (lane >> 5);

  // this XCD's share of the tile space (hardware block b runs on XCD b % 8: speed only)
  const int P = gridDim.x, xcd = blockIdx.x & 7;
  const int chunk_beg = (int)((int64_t)ntiles * xcd / 8), chunk_end = (int)((int64_t)ntiles * (xcd + 1) / 8);
  const int nblk = (P - xcd + 7) >> 3;                               // blocks on this XCD
  int tile = chunk_beg + (blockIdx.x >> 3);
  if (tile >= chunk_end) return;

  using ALoad = typename std::conditional<AM == 1, ConvPwLoader<NA, RP, PRE>, ConvALoader<NA, RP, PRE, 0>>::type;
  using BLoad = RowLoader<NB, RP>;

  // ---- state of the tile being retired
  f32x16 accp;
#pragma unroll
  for (int r = 0; r < 16; ++r) accp[r] = 0.f;
  const int ecol_l = wc * 32 + (lane & 31);
  const int erow_l = wr * 32 + 4 * (lane >> 5);
  int prow0 = 0, pcol = 0, p_n0 = -1;
  float sc = 1.f, sh = 0.f, sc2 = 1.f, sh2 = 0.f;
  float rres[2 * CPS];

  // Branch-free on purpose: with branches around the loads the compiler can no longer tell which
  // vector-memory results are outstanding and drains the queue (vmcnt(0)) in the middle of a K-step,
  // right after the next step's operand loads were issued.  Buffer accesses with an out-of-range
  // offset read zero / are dropped, which replaces every lane predicate; a null tensor gets an empty
  // descriptor.
  const __amdgpu_buffer_rsrc_t res_rsrc = make_rsrc(a.res, a.res ? (uint32_t)((int64_t)a.M * a.Cout * 4) : 0u);
  const __amdgpu_buffer_rsrc_t y_rsrc = make_rsrc(a.y, a.y ? (uint32_t)((int64_t)a.M * a.y_ld * 4) : 0u);
  const __amdgpu_buffer_rsrc_t y2_rsrc = make_rsrc(a.y2, a.y2 ? (uint32_t)((int64_t)a.M * a.y_ld * 4) : 0u);
  // No block-uniform conditions inside the steps either (the compiler would branch on them): "no tile
  // to retire yet" and "column beyond Cout" are folded into row_lim (rows below it are stored), "no
  // shortcut" into an empty descriptor that reads 0.0, the activation into a per-lane slope for
  // negative inputs (1 = linear, 0 = ReLU, alpha = PReLU / LeakyReLU).
  int row_lim = 0;
  float sl = 1.f, sl2 = 1.f;
  // chunk J retires registers 2J and 2J+1 of accp
  auto epi_pre = [&](auto jc) {
    constexpr int J = decltype(jc)::value;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 2 * J + q;       // compile-time after unrolling
      const int row = prow0 + (r & 3) + 8 * (r >> 2);
      const uint32_t off = (uint32_t)(row * a.Cout + pcol) * 4u;
      rres[(J % CPS) * 2 + q] =
          __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res_rsrc, row < row_lim ? off : OOB, 0, 0));
    }
  };
  auto epi_post = [&](auto jc) {
    constexpr int J = decltype(jc)::value;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 2 * J + q;
      const int row = prow0 + (r & 3) + 8 * (r >> 2);
      const float v = fmaf(accp[r], sc, sh);
      const float t = (v >= 0.f ? v : v * sl) + rres[(J % CPS) * 2 + q];
      const float u = fmaf(t, sc2, sh2);
      const uint32_t off = (uint32_t)(row * a.y_ld + a.y_coff + pcol) * 4u;
      const uint32_t o = row < row_lim ? off : OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, t), y_rsrc, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, u >= 0.f ? u : u * sl2), y2_rsrc, o, 0, 0);
    }
  };

  unsigned long long tr_setup = 0, tr_pro = 0, tr_steps = 0, tr_hand = 0, tr_n = 0;
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long tr_c0 = a.trace ? __builtin_amdgcn_s_memtime() : 0;
  while (true) {
    const unsigned long long tA = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
    int mt, nt;
    a.fd_tiles_n.divmod(tile, mt, nt);
    const int m0 = mt * T::BM, n0 = nt * T::BN;
    ALoad ald(a, m0);
    BLoad bld(a.w + (int64_t)n0 * a.Kpad, (int64_t)a.Cout - n0, a.Kpad);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    f32x4 ra[NA], rb[NB];
    auto stage = [&](int buf) {
      float* wa = smem + buf * BUF + st_off;
      ald.finish(ra);
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(wa + i * RP * LDS_STRIDE) = ra[i];
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(wa + OFFB + i * RP * LDS_STRIDE) = rb[i];
    };
    auto mfma_step = [&](int cur) {
      const float* pa = smem + cur * BUF + (wr * 32) * LDS_STRIDE + fr_off;
      const float* pb = smem + cur * BUF + OFFB + (wc * 32) * LDS_STRIDE + fr_off;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const f32x4 fa0 = *reinterpret_cast<const f32x4*>(pa + 16 * s2), fa1 = *reinterpret_cast<const f32x4*>(pa + 16 * s2 + 4);
        const f32x4 fb0 = *reinterpret_cast<const f32x4*>(pb + 16 * s2), fb1 = *reinterpret_cast<const f32x4*>(pb + 16 * s2 + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[t], fb0[t], acc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[t], fb1[t], acc, 0, 0, 0);
      }
    };
    // one K-step; `jc` selects the chunk of the previous tile retired alongside it (8 = none)
    auto kstep = [&](int ks, auto jc) {
      constexpr int J = decltype(jc)::value;
      const bool more = ks + 1 < KS;
      if (more) {
        ald.load(ks + 1, ra);
        bld.load(ks + 1, rb);
      }
      __builtin_amdgcn_sched_barrier(0);   // the prefetch stays above the MFMAs (see gemm_mainloop2)
      if constexpr (J * CPS < 8) {
        epi_pre(std::integral_constant<int, J * CPS>());
        if constexpr (CPS > 1) epi_pre(std::integral_constant<int, J * CPS + 1>());
        if constexpr (CPS > 2) epi_pre(std::integral_constant<int, J * CPS + 2>());
        if constexpr (CPS > 3) epi_pre(std::integral_constant<int, J * CPS + 3>());
      }
      mfma_step(ks & 1);
      if constexpr (J * CPS < 8) {
        epi_post(std::integral_constant<int, J * CPS>());
        if constexpr (CPS > 1) epi_post(std::integral_constant<int, J * CPS + 1>());
        if constexpr (CPS > 2) epi_post(std::integral_constant<int, J * CPS + 2>());
        if constexpr (CPS > 3) epi_post(std::integral_constant<int, J * CPS + 3>());
      }
      if (more) stage((ks & 1) ^ 1);
      lds_barrier();
    };

    const unsigned long long tB = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
    ald.load(0, ra);
    bld.load(0, rb);
    stage(0);
    lds_barrier();
    const unsigned long long tC = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
    // the first eight steps carry the previous tile's epilogue, unrolled so that its accumulator
    // registers are addressed statically
    if (0 < KS) kstep(0, std::integral_constant<int, 0>());
    if (1 < KS) kstep(1, std::integral_constant<int, 1>());
    if (2 < KS) kstep(2, std::integral_constant<int, 2>());
    if (3 < KS) kstep(3, std::integral_constant<int, 3>());
    if (4 < KS) kstep(4, std::integral_constant<int, 4>());
    if (5 < KS) kstep(5, std::integral_constant<int, 5>());
    if (6 < KS) kstep(6, std::integral_constant<int, 6>());
    if (7 < KS) kstep(7, std::integral_constant<int, 7>());
    for (int ks = 8; ks < KS; ++ks) kstep(ks, std::integral_constant<int, 8>());
    // steps that could not carry the whole previous tile: retire the rest now
    {
      const int done = KS * CPS;      // chunks already retired inside the steps
      if (done <= 1) { epi_pre(std::integral_constant<int, 1>()); epi_post(std::integral_constant<int, 1>()); }
      if (done <= 2) { epi_pre(std::integral_constant<int, 2>()); epi_post(std::integral_constant<int, 2>()); }
      if (done <= 3) { epi_pre(std::integral_constant<int, 3>()); epi_post(std::integral_constant<int, 3>()); }
      if (done <= 4) { epi_pre(std::integral_constant<int, 4>()); epi_post(std::integral_constant<int, 4>()); }
      if (done <= 5) { epi_pre(std::integral_constant<int, 5>()); epi_post(std::integral_constant<int, 5>()); }
      if (done <= 6) { epi_pre(std::integral_constant<int, 6>()); epi_post(std::integral_constant<int, 6>()); }
      if (done <= 7) { epi_pre(std::integral_constant<int, 7>()); epi_post(std::integral_constant<int, 7>()); }
    }

    const unsigned long long tD = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
    // hand the finished tile over to the retiring set
    accp = acc;
    prow0 = m0 + erow_l;
    // The per-channel constants depend on the column tile only, and a block's tiles are nblk apart: whenever nblk is a
    // multiple of the column-tile count (every power-of-two channel count) they all share ONE column tile and the six
    // dependent global loads -- 0.7-1.3 us per tile in the block traces, 5-10 % of a 2..8-step tile -- happen once per block.
    if (n0 != p_n0) {
      p_n0 = n0;
      pcol = n0 + ecol_l;
      if (pcol < a.Cout) {
        sc = a.scale ? a.scale[pcol] : 1.f;
        sh = a.shift ? a.shift[pcol] : 0.f;
        sc2 = a.scale2 ? a.scale2[pcol] : 1.f;
        sh2 = a.shift2 ? a.shift2[pcol] : 0.f;
        sl = a.act == ACT_RELU ? 0.f : (a.act == ACT_PRELU ? (a.alpha ? a.alpha[pcol] : 0.f) : 1.f);
        sl2 = a.act2 == ACT_RELU ? 0.f : (a.act2 == ACT_PRELU ? (a.alpha2 ? a.alpha2[pcol] : 0.f) : 1.f);
      }
    }
    row_lim = pcol < a.Cout ? a.M : 0;
    const int nxt = tile + nblk;                        // static round-robin inside the XCD's chunk
    if (a.trace) {
      const unsigned long long tE = __builtin_amdgcn_s_memrealtime();
      tr_setup += tB - tA; tr_pro += tC - tB; tr_steps += tD - tC; tr_hand += tE - tD; ++tr_n;
    }
    if (nxt >= chunk_end) break;
    tile = nxt;
  }
  if (a.trace && tid == 0) {
    unsigned long long* t = a.trace + (size_t)blockIdx.x * 8;
    t[0] = tr_setup; t[1] = tr_pro; t[2] = tr_steps; t[3] = tr_hand; t[4] = tr_n; t[5] = tr_t0;
    t[6] = __builtin_amdgcn_s_memrealtime();
    t[7] = 2 | ((__builtin_amdgcn_s_memtime() - tr_c0) << 8);
  }
  // drain: the last tile has no successor to hide behind
  epi_pre(std::integral_constant<int, 0>()); epi_post(std::integral_constant<int, 0>());
  epi_pre(std::integral_constant<int, 1>()); epi_post(std::integral_constant<int, 1>());
  epi_pre(std::integral_constant<int, 2>()); epi_post(std::integral_constant<int, 2>());
  epi_pre(std::integral_constant<int, 3>()); epi_post(std::integral_constant<int, 3>());
  epi_pre(std::integral_constant<int, 4>()); epi_post(std::integral_constant<int, 4>());
  epi_pre(std::integral_constant<int, 5>()); epi_post(std::integral_constant<int, 5>());
  epi_pre(std::integral_constant<int, 6>()); epi_post(std::integral_constant<int, 6>());
  epi_pre(std::integral_constant<int, 7>()); epi_post(std::integral_constant<int, 7>());
}

// ------------------------------------------------------------------------------------------
// B-direct patch kernel with the epilogue taken out of the block's critical path (AMP = patch form 3 / 5 / 6).
//
// In conv_igemm_kernel a block stops issuing MFMAs for ~10 us per tile: the staged epilogue (shortcut rows fetched one
// after the other) and the next tile's set-up.  That is 7 % of a 72-step tile, 14 % of a 36-step one (28x28 stage) and
// 24 % of an 18-step one (64-channel layers).  Here, as in conv_pipe_kernel, a finished tile's accumulators move to a
// second register set and are retired two registers per K-step during the NEXT part's first eight K-steps: shortcut
// values requested before a step's MFMAs, scale / shift / activation / add / stores after them, straight from the MFMA
// layout (32 channels x two rows per instruction = two 128-byte segments), no LDS, no barrier.  The grid is always the
// persistent stream-K one (also for short K loops: the partial-tile hand-off costs less than a non-persistent block's
// set-up per tile).  Same arithmetic per output element as conv_epilogue, in the same order.
// Launcher restrictions: plain output geometry, unit-stride shortcut, no RELU6, 32-bit byte offsets (conv_bdp_ok).
template <class T, int AMP>
__global__ __launch_bounds__(T::NT, T::MIN_BLOCKS) void conv_bdp_kernel(const ConvArgs a) {
  static_assert(T::WM == 1 && T::WN == 1 && T::BM == 64 && T::BN == 64, "64x64 tile, one 32x32 fragment per wave");
  using PA = typename std::conditional<AMP == 3, PatchA<T, PATCH_EMAX_S>,
                                       typename std::conditional<AMP == 5, PatchA<T, PATCH_EMAX_L>, PatchA2D<T>>::type>::type;
  constexpr bool TILE2D = AMP == 6;
  constexpr int SLAB = T::BM * T::BN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // behind the patch: the retiring tile's accumulators, register r of thread t at [r][t] (lane-private: no barrier, no
  // bank conflict), then the stream-K time-out word
  float* accp = smem + PA::EMAX * 32 + threadIdx.x;
  int* s_timeout = reinterpret_cast<int*>(smem + PA::EMAX * 32 + 16 * T::NT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int P = gridDim.x;
  const int p = xcd_remap(blockIdx.x, P);
  const int KS = a.Kpad / BK;
  const int tiles_n = (a.Cout + T::BN - 1) / T::BN;
  const int tiles_m = (a.M + T::BM - 1) / T::BM;
  const int I = tiles_m * tiles_n * KS;
  auto sk_begin = [&](int q) -> int { return (int)((int64_t)I * q / P); };
  const int beg = sk_begin(p), end = sk_begin(p + 1);
  const unsigned long long tr_t0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0;
  const unsigned long long tr_c0 = a.trace ? __builtin_amdgcn_s_memtime() : 0;

  // ---- the tile being retired (registers 2J, 2J + 1 of accp are chunk J; p_next = next chunk, 8 = nothing pending)
  int p_next = 8, p_row0 = 0, pcol = 0, row_lim = 0;
  float sc = 1.f, sh = 0.f, sl = 1.f, sc2 = 1.f, sh2 = 0.f, sl2 = 1.f;
  float rres[2], av[2];
  const int erow_l = T::wave_row() * 32 + 4 * (lane >> 5);
  const int ecol_l = T::wave_col() * 32 + (lane & 31);
  // branch-free accesses: an out-of-range offset reads zero / is dropped, a null tensor gets an empty descriptor
  const __amdgpu_buffer_rsrc_t res_rsrc = make_rsrc(a.res, a.res ? (uint32_t)((int64_t)a.M * a.Cout * 4) : 0u);
  const __amdgpu_buffer_rsrc_t y_rsrc = make_rsrc(a.y, a.y ? (uint32_t)((int64_t)a.M * a.Cout * 4) : 0u);
  const __amdgpu_buffer_rsrc_t y2_rsrc = make_rsrc(a.y2, a.y2 ? (uint32_t)((int64_t)a.M * a.Cout * 4) : 0u);
  // Accumulator register r of a lane is tile row erow_l + (r & 3) + 8 (r >> 2): relative to the lane's register 0 that is
  // a block-uniform number of output rows, so a register's byte offset is the lane's base offset (set at hand-over, OOB
  // for a column beyond Cout) plus a scalar -- one vector add and one select per access.
  uint32_t p_off = OOB, voff[2];
  auto reg_rows = [&](int r) -> int { return TILE2D ? (r >> 2) * a.W + (r & 3) : 8 * (r >> 2) + (r & 3); };
  // Chunk j = accumulator registers 2j, 2j + 1, kept in LDS (in registers the set costs 16 VGPRs the K loop does not have:
  // 44-64 spilled; a switch over static register indices made the compiler clone the K loop): pre() requests the two
  // values and the two shortcut values before the step's MFMAs, post() uses them after.
  auto retire_pre = [&](int j) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int dr = reg_rows(2 * j + q);
      const bool ok = p_off != OOB && (TILE2D || p_row0 + dr < row_lim);      // p_row0: the lane's register-0 row
      voff[q] = ok ? p_off + (uint32_t)(dr * a.Cout) * 4u : OOB;
      rres[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res_rsrc, voff[q], 0, 0));
      av[q] = accp[(2 * j + q) * T::NT];
    }
  };
  auto retire_post = [&](int j) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float v = fmaf(av[q], sc, sh);
      const float t = (v >= 0.f ? v : v * sl) + rres[q];
      const float u = fmaf(t, sc2, sh2);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, t), y_rsrc, voff[q], 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, u >= 0.f ? u : u * sl2), y2_rsrc, voff[q], 0, 0);
    }
  };
  auto hook_pre = [&]() {
    if (p_next < 8) retire_pre(p_next);
  };
  auto hook_post = [&]() {
    if (p_next < 8) {
      retire_post(p_next);
      ++p_next;
    }
  };
  auto flush = [&]() {                                     // whatever the K-steps could not carry
    while (p_next < 8) {
      retire_pre(p_next);
      retire_post(p_next);
      ++p_next;
    }
  };

  int it = beg;
  while (it < end) {
    int tile, kb, mt, nt;
    a.fd_ks.divmod(it, tile, kb);
    const int left = end - it;
    const int ke = (KS - kb <= left) ? KS : kb + left;
    a.fd_tiles_n.divmod(tile, mt, nt);
    const int m0 = mt * T::BM, n0 = nt * T::BN;
    f32x16 acc[1][1];
    zero_acc<T>(acc);
    PA al(a, m0);
    gemm_mainloop_patch_bd<T>(al, a, n0, kb, ke, smem, acc, [] {}, hook_pre, hook_post);

    if (kb != 0) {
      // not the owner of this tile: publish the partial accumulators (fragment order, 16 B per lane)
      float* slab = a.sk_slab + (int64_t)p * SLAB;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
        *reinterpret_cast<f32x4*>(slab + (q * T::NT + tid) * 4) = v;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.sk_flag + p, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // owner: collect the rest of the K range from the blocks that follow
      int kdone = ke;
      int q = p;
      while (kdone < KS) {
        ++q;
        const int qb = sk_begin(q), qe = sk_begin(q + 1);
        const int q_kb = qb - tile * KS;
        const int q_len = qe - qb;
        const int q_ke = (KS - q_kb <= q_len) ? KS : q_kb + q_len;
        if (tid == 0) {
          int spins = 0, timeout = a.sk_spin_limit < 0 ? 1 : 0;
          while (!timeout && __hip_atomic_load(a.sk_flag + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.sk_epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > a.sk_spin_limit) {
              timeout = 1;
              break;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          *s_timeout = timeout;
        }
        __syncthreads();
        const int timeout = *s_timeout;
        __syncthreads();
        if (!timeout) {
          const float* slab = a.sk_slab + (int64_t)q * SLAB;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(slab + (r4 * T::NT + tid) * 4);
            acc[0][0][4 * r4] += v[0];
            acc[0][0][4 * r4 + 1] += v[1];
            acc[0][0][4 * r4 + 2] += v[2];
            acc[0][0][4 * r4 + 3] += v[3];
          }
        } else {
          // the partner is not co-resident: compute its K range here, from zero like its slab, and add (same bits)
          float* stash = a.sk_slab + ((int64_t)P + p) * SLAB;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            f32x4 v = {acc[0][0][4 * r4], acc[0][0][4 * r4 + 1], acc[0][0][4 * r4 + 2], acc[0][0][4 * r4 + 3]};
            *reinterpret_cast<f32x4*>(stash + (r4 * T::NT + tid) * 4) = v;
          }
          zero_acc<T>(acc);
          gemm_mainloop_patch_bd<T>(al, a, n0, q_kb, q_ke, smem, acc, [] {}, hook_pre, hook_post);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(stash + (r4 * T::NT + tid) * 4);
            acc[0][0][4 * r4] = v[0] + acc[0][0][4 * r4];
            acc[0][0][4 * r4 + 1] = v[1] + acc[0][0][4 * r4 + 1];
            acc[0][0][4 * r4 + 2] = v[2] + acc[0][0][4 * r4 + 2];
            acc[0][0][4 * r4 + 3] = v[3] + acc[0][0][4 * r4 + 3];
          }
        }
        kdone = q_ke;
      }
      // hand the finished tile over to the retiring set
      flush();
#pragma unroll
      for (int r = 0; r < 16; ++r) accp[r * T::NT] = acc[0][0][r];
      pcol = n0 + ecol_l;
      if constexpr (TILE2D) {
        int n, h0, w0;
        const int pix0 = tile2d_pix0(a, m0, n, h0, w0);
        p_row0 = pix0 + (erow_l >> 3) * a.W + (erow_l & 7);      // erow_l = 32 wr + 4 h: the lane's register-0 pixel
      } else {
        p_row0 = m0 + erow_l;
      }
      row_lim = a.M;
      if (pcol < a.Cout) {
        p_off = (uint32_t)(p_row0 * a.Cout + pcol) * 4u;
        sc = a.scale ? a.scale[pcol] : 1.f;
        sh = a.shift ? a.shift[pcol] : 0.f;
        sc2 = a.scale2 ? a.scale2[pcol] : 1.f;
        sh2 = a.shift2 ? a.shift2[pcol] : 0.f;
        sl = a.act == ACT_RELU ? 0.f : (a.act == ACT_PRELU ? (a.alpha ? a.alpha[pcol] : 0.f) : 1.f);
        sl2 = a.act2 == ACT_RELU ? 0.f : (a.act2 == ACT_PRELU ? (a.alpha2 ? a.alpha2[pcol] : 0.f) : 1.f);
      } else {
        p_off = OOB;
      }
      p_next = 0;
    }
    it += ke - kb;
  }
  flush();                                                 // the last tile has no successor to hide behind
  if (a.trace && tid == 0) {
    unsigned long long* t = a.trace + (size_t)blockIdx.x * 8;
    t[0] = t[1] = t[2] = t[3] = 0;
    // where the block ran: HW_ID (wave / SIMD / CU / SH / SE fields) and XCC_ID, for the placement statistics of dbg = 512
    t[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    t[5] = tr_t0;
    t[6] = __builtin_amdgcn_s_memrealtime();
    t[7] = 1 | ((__builtin_amdgcn_s_memtime() - tr_c0) << 8);
  }
}

// Per-device launch state.  One process drives one GPU by convention, but nothing here depends on it:
// the CU count and the "dynamic LDS limit raised" bit of every kernel are cached per device ordinal.
constexpr int kMaxDevices = 64;

static int cur_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}

static int num_cus() {
  static std::atomic<int> cus[kMaxDevices];
  const int dev = cur_device();
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    hipDeviceProp_t prop;
    c = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    if (c <= 0) c = 256;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

// hipFuncAttributeMaxDynamicSharedMemorySize, once per (device, kernel).  Keyed on the kernel's ADDRESS: every
// convolution kernel has the same function type, so a per-type flag would be shared by all of them.
static int allow_dynamic_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  const int dev = cur_device();
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, kern})) return 0;
  DIF_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, kern});
  return 0;
}

int conv_max_blocks() { return 4 * num_cus(); }

// The kernel instantiation the last conv_run of this thread launched, as `family<tile, variant>` (dif_net_op_info:
// the bench line and the per-layer tables name the kernel that ran, not the family's generic name).
static thread_local const char* g_last_kernel = "";
const char* conv_last_kernel() { return g_last_kernel; }
static const char* am_form(int am) {
  switch (am) {
    case 0: return "gather";
    case 1: return "pointwise";
    case 3: return "patch128";
    case 5: return "patch168";
    case 6: return "patch8x8";
    case 13: return "patch128+Bdirect";
    case 15: return "patch168+Bdirect";
    case 16: return "patch8x8+Bdirect";
  }
  return "?";
}
template <class T>
static std::string kernel_label(const char* family, int am, const char* extra = "") {
  return std::string(family) + "<" + std::to_string(T::BM) + "x" + std::to_string(T::BN) + "," + am_form(am) + extra + ">";
}
size_t conv_slab_floats() { return 128 * 128; }   // per resident slot (sk_max_blocks of them): published partial + fallback stash of
                                                   // either shipped tile -- 1024 x 2 x 64x64 or 512 x 2 x 128x128 floats, 64 MiB per lane

template <class T, bool PRE, int AM, int BF3 = 0>
static int launch_conv_pre(const ConvArgs& a, hipStream_t st);

// Short K loop, several whole tiles per resident block, plain output, unit-stride shortcut: the
// software-pipelined kernel.  Returns 1 when it does not apply (the caller falls through).
template <class T, bool PRE, int AM, int CPS = 1>
static int launch_conv_pipe(const ConvArgs& a, hipStream_t st) {
  auto kern = conv_pipe_kernel<T, PRE, AM, CPS>;
  constexpr int lds = T::LDS_BYTES;
  if (allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return -1;
  const int64_t tiles = ((a.M + T::BM - 1) / T::BM) * (int64_t)((a.Cout + T::BN - 1) / T::BN);
  // The grid is sized for the WHOLE chip even when the lanes split it (ConvArgs::sk_max_blocks = half the resident slots
  // for the stream-K kernels, whose blocks wait for each other): this kernel's blocks are independent, so the half that does
  // not fit beside the other lane's launch starts as slots come free -- from either lane -- and fills the staggered ends
  // in which one or two resident blocks per CU leave the matrix pipes half idle (r03_ablation.txt item 21 e: ResNet-50V2
  // +2.9 % at batch 256; 3x and 4x the lane's slots gain less, 8x loses; alone on the chip 2x the slots loses 1 %).
  const int64_t slots = 4 * (int64_t)num_cus();
  const int64_t P = tiles < slots ? tiles : slots;
  ConvArgs b = a;
  b.fd_howo = make_fastdiv(a.Ho * a.Wo);
  b.fd_wo = make_fastdiv(a.Wo);
  b.fd_cin = make_fastdiv(a.Cin);
  b.fd_kw = make_fastdiv(a.KW);
  b.fd_ks = make_fastdiv(a.Kpad / BK);
  b.fd_taps = make_fastdiv(a.KH * a.KW);
  b.fd_tiles_n = make_fastdiv((a.Cout + T::BN - 1) / T::BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)P), dim3(T::NT), lds, st, b);
  DIF_HIP(hipGetLastError());
  static const std::string label = kernel_label<T>("conv_pipe_kernel", AM, PRE ? ",preact" : "");
  g_last_kernel = label.c_str();
  return 0;
}

// conv_bdp_kernel's restrictions (otherwise the launcher keeps conv_igemm_kernel's B-direct form)
static bool conv_bdp_ok(const ConvArgs& a) {
  // 64-channel layers (18 K-steps per tile) measured 2 % faster one tile per block than on the persistent grid
  constexpr int min_ks = 32;
  if (a.bdp_mode == 1 || !a.w_frag || a.y_sub) return false;
  if (a.bdp_mode != 2 && a.Kpad / BK < min_ks) return false;
  if (!(a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0 && a.y_ld == a.Cout && a.y_coff == 0)) return false;
  if (a.res && (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo)) return false;
  if (a.act == ACT_RELU6 || a.act2 == ACT_RELU6) return false;
  if ((int64_t)a.M * a.Cout * 4 >= 0xFFFFFFF0LL) return false;
  // at least 1.5 tiles per resident block, or there is no next tile to retire the previous one under (ResNet50V2's 3x3
  // layers on half-chip lane grids: 0.8-1.5 tiles per block, 1 % slower here than on conv_igemm_kernel)
  const int64_t tiles = ((a.M + 63) / 64) * (int64_t)((a.Cout + 63) / 64);
  int64_t slots = 4 * (int64_t)num_cus();
  if (slots > a.sk_max_blocks) slots = a.sk_max_blocks;
  return a.bdp_mode == 2 || 2 * tiles >= 3 * slots;
}

template <class T, int AMP>
static int launch_conv_bdp(const ConvArgs& a, hipStream_t st) {
  auto kern = conv_bdp_kernel<T, AMP>;
  constexpr int emax = AMP == 3 ? PATCH_EMAX_S : (AMP == 5 ? PATCH_EMAX_L : 100);
  constexpr int lds_bytes = emax * 128 + 16 * T::NT * 4 + 16;      // patch + retiring accumulators + time-out word
  if (allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes)) return -1;
  const int64_t tiles = ((a.M + T::BM - 1) / T::BM) * (int64_t)((a.Cout + T::BN - 1) / T::BN);
  const int KS = a.Kpad / BK;
  const int64_t I = tiles * KS;
  if (I >= 0x7fffffffLL) return set_error("conv: iteration space too large");
  int64_t P = T::BLOCKS_PER_CU * (int64_t)num_cus();
  if (P > a.sk_max_blocks) P = a.sk_max_blocks;
  // (two or three times as many blocks as resident slots, so that the dispatcher evens out the 20-35 % spread of the blocks'
  // lifetimes, measured no gain on one lane and -2..-7 % on two: profiles/r03_ablation.txt)
  if (P > (I + 3) / 4) P = (I + 3) / 4;
  if (P < 1) P = 1;
  ConvArgs b = a;
  b.fd_howo = make_fastdiv(a.Ho * a.Wo);
  b.fd_wo = make_fastdiv(a.Wo);
  b.fd_cin = make_fastdiv(a.Cin);
  b.fd_kw = make_fastdiv(a.KW);
  b.fd_ks = make_fastdiv(KS);
  b.fd_taps = make_fastdiv(a.KH * a.KW);
  b.fd_wp = make_fastdiv(a.W + 2);
  b.fd_rpi = make_fastdiv(a.H + 1);
  b.fd_t2_w = make_fastdiv(a.W / 8 > 0 ? a.W / 8 : 1);
  b.fd_t2_img = make_fastdiv((a.H / 8) * (a.W / 8) > 0 ? (a.H / 8) * (a.W / 8) : 1);
  b.fd_tiles_n = make_fastdiv((a.Cout + T::BN - 1) / T::BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)P), dim3(T::NT), lds_bytes, st, b);
  DIF_HIP(hipGetLastError());
  static const std::string label = kernel_label<T>("conv_bdp_kernel", AMP + 10);
  g_last_kernel = label.c_str();
  return 0;
}

// 3x3 / stride 1 / pad 1, whole 32-channel slices in channel-block-major K order, plain output geometry, and a
// 64-pixel tile's halo patch bounded by PATCH_EMAX entries (row wraps add 2 entries each, an image boundary adds
// one padded row).  IResNet's 28x28, 14x14 and 7x7 stages qualify; 56x56 and up keep the per-K-step gather.
// returns 0 (no), PATCH_EMAX_S or PATCH_EMAX_L: the smallest patch size that covers every tile of the layer
static bool patch_shape(const ConvArgs& a) {
  if ((a.off & CONV_OFF_PATCH) || a.pre_scale || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad_t != 1 || a.pad_l != 1) return false;
  return a.Cin % BK == 0 && a.k_order == 1 && a.Ho == a.H && a.Wo == a.W;
}
// entries the halo patch of a BM-pixel linear tile can need
static int patch_entry_bound(const ConvArgs& a, int BM) {
  const int HW = a.H * a.W, WP = a.W + 2;
  const int row_wraps = (BM - 2) / a.W + 1, img_wraps = a.N > 1 ? (BM - 2) / HW + 1 : 0;
  return (BM - 1) + 2 * row_wraps + WP * img_wraps + 2 * WP + 4;
}
static int patch_applies(const ConvArgs& a) {
  if (!patch_shape(a)) return 0;
  const int e_bound = patch_entry_bound(a, 64);
  return e_bound <= PATCH_EMAX_S ? PATCH_EMAX_S : (e_bound <= PATCH_EMAX_L ? PATCH_EMAX_L : 0);
}
// The split-bf16 patch kernel (128-pixel tile, 128 or 64 columns): 1 = linear patch (maps up to 28 wide), 2 = two 8x8
// sub-tiles (sides multiples of 8), 0 = the layer stays on the f32 kernels.  net.hip asks the same question at finalize
// (conv_bf3p_form) to decide which weight layout a layer needs.
int conv_bf3p_form(int H, int W, bool batch_gt1, int Cout) {
  if (Cout < 64) return 0;
  const int HW = H * W, WP = W + 2, BM = 128;
  const int row_wraps = (BM - 2) / W + 1, img_wraps = batch_gt1 ? (BM - 2) / HW + 1 : 0;
  if ((BM - 1) + 2 * row_wraps + WP * img_wraps + 2 * WP + 4 <= bf3p_emax(128)) return 1;
  return (H % 8 == 0 && W % 8 == 0 && (int64_t)H * W * Cout < (1 << 28)) ? 2 : 0;
}
static int bf3p_applies(const ConvArgs& a) {
  if (!a.w3f || !patch_shape(a)) return 0;                 // (a sub-sampled first output is fine: epilogue variant 2)
  if (!(a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0 && a.y_ld == a.Cout && a.y_coff == 0)) return 0;
  if (a.Cout % 4 != 0 || (a.res && (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo))) return 0;   // the lean epilogue's case
  if ((int64_t)a.M * a.Cout * 4 >= 0xFFFFFFF0LL || (int64_t)2 * a.H * a.W * a.Cin * 4 >= 0x7fffffffLL) return 0;
  return conv_bf3p_form(a.H, a.W, true, a.Cout);
}

// the 8x8-tile form of the patch path (AM = 6): the same layers on maps whose sides are multiples of 8, where the
// linear patch would not fit (IResNet's 112x112 and 56x56 layers, VGG16's, the detector's 208 / 104 stages)
static bool patch2d_applies(const ConvArgs& a) {
  if ((a.off & CONV_OFF_PATCH2D) || a.pre_scale || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad_t != 1 || a.pad_l != 1) return false;
  if (a.Cin % BK != 0 || a.k_order != 1 || a.Ho != a.H || a.Wo != a.W) return false;
  if ((int64_t)a.H * a.W * a.Cin * 4 >= 0x7fffffffLL) return false;      // one image per buffer descriptor
  return a.H % 8 == 0 && a.W % 8 == 0 && patch_applies(a) == 0;
}

constexpr int SK_MIN_KS = 32;   // K-steps per tile from which few-tile layers take the stream-K grid (measured threshold)

static bool pipe_applies(const ConvArgs& a, int64_t tiles, int KS, int64_t slots) {
  constexpr int sk_min_ks = SK_MIN_KS;
  if (!a.use_pipe) return false;                                      // dif_net_set_option("pipe", 0)
  if (a.y_sub) return false;                                          // its stores know one output geometry
  if (slots < 8) return false;                                        // the kernel deals tiles out per XCD (8 of them)
  if (KS >= sk_min_ks && tiles < 8 * slots) return false;           // long K, few tiles: stream-K's case
  if (tiles < slots + slots / 2) return false;                        // fewer than ~1.5 tiles per block: nothing to overlap
  if (a.act == ACT_RELU6 || a.act2 == ACT_RELU6) return false;        // its epilogue knows slopes, not clamps

  if (!(a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0)) return false;
  if (a.res && (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo)) return false;
  if ((int64_t)a.M * (a.y_ld > a.Cout ? a.y_ld : a.Cout) * 4 >= 0xFFFFFFF0LL) return false;   // 32-bit buffer offsets
  return true;
}

// ------------------------------------------------------------------------------------------
// conv_t2_kernel (round 4): 3x3 / stride 1 layers on maps whose sides are multiples of 8, with a
// 128-pixel x 64-channel tile per block -- TWO consecutive 8x8 tiles of the 8x8-tile order, one per wave row, each with its
// own 10 x 10 halo patch (200 entries, 25.6 KB) -- instead of one: a wave owns 64 pixels x 32 channels (two row fragments),
// so a B fragment fetched from L2 feeds 32 MFMAs instead of 16 and a tile's set-up and epilogue are paid once per 128
// pixels.  Meant for IResNet's 64-channel 112 x 112 / 56 x 56 layers (18 K-steps per tile: 18 us of mainloop beside
// 10-12 us of set-up and epilogue on the 64 x 64 kernel).  Same B-direct mainloop as gemm_mainloop_patch_bd, same
// accumulation order per output element (K-step, half, t): bit-identical.  One whole tile per block (no stream-K).
using TileT2 = Tile<2, 1, 2, 2>;

struct PatchA2D2 {
  static constexpr int EMAX = 200, SIDE = 10, NPC = (EMAX * 8 + 255) / 256;
  __amdgpu_buffer_rsrc_t rsrc;
  int base[2];              // entry of this lane's output pixels (row fragment mi of the wave row's sub-tile), tap (0, 0)
  int hw0[2], ioff[2];      // per sub-tile: (h0 - 1) << 16 | (w0 - 1) & 0xffff; its image's offset in pixels from the first image
  int H, W, Cin4;
  __device__ __forceinline__ PatchA2D2(const ConvArgs& a, int m0) {
    const int lane = threadIdx.x & 63;
    H = a.H;
    W = a.W;
    Cin4 = a.Cin * 4;
    int n0, h0, w0, n1, h1, w1;
    tile2d_pix0(a, m0, n0, h0, w0);
    const bool second = m0 + 64 < a.M;
    tile2d_pix0(a, second ? m0 + 64 : m0, n1, h1, w1);
    const int64_t img_elems = (int64_t)a.H * a.W * a.Cin;
    rsrc = make_rsrc(a.x + n0 * img_elems, (uint32_t)((n1 - n0 + 1) * img_elems * 4));
    hw0[0] = ((h0 - 1) << 16) | ((w0 - 1) & 0xffff);
    hw0[1] = second ? (((h1 - 1) << 16) | ((w1 - 1) & 0xffff)) : (int)0xc000c000;     // far outside: every entry reads zero
    ioff[0] = 0;
    ioff[1] = (n1 - n0) * a.H * a.W;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int r = mi * 32 + (lane & 31);
      base[mi] = TileT2::wave_row() * 100 + (r >> 3) * SIDE + (r & 7);
    }
  }
  __device__ __forceinline__ void load(int cblk, f32x4 (&pr)[NPC]) const {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                          // offsets recomputed per slice, not kept (two slices per tile here)
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = tid + 256 * j;
      const int e = slot >> 3, q = slot & 7;
      const int sub = e >= 100 ? 1 : 0, el = e - 100 * sub;
      const int py = (el * 205) >> 11, px = el - py * SIDE;          // el / 10 for el < 1024
      const int hi = (hw0[sub] >> 16) + py, wi = (int)(short)(hw0[sub] & 0xffff) + px;
      const bool ok = e < EMAX && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const int chunk = q ^ patch_swz(e);
      pr[j] = buf_load4(rsrc, ok ? (uint32_t)((ioff[sub] + hi * W + wi) * Cin4 + chunk * 16 + cblk * 128) : OOB);
    }
  }
  __device__ __forceinline__ void store(float* patch, const f32x4 (&pr)[NPC]) const {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = threadIdx.x + 256 * j;
      if (slot < EMAX * 8) *reinterpret_cast<f32x4*>(patch + slot * 4) = pr[j];
    }
  }
};

__device__ __forceinline__ void t2_mainloop(const PatchA2D2& pa, const ConvArgs& a, int n0, float* patch, f32x16 (&acc)[2][1]) {
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int KS = a.Kpad / BK;
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w_frag, a.w_frag_bytes);
  const uint32_t lane_off = (uint32_t)((n0 >> 5) + TileT2::wave_col()) * (uint32_t)KS * 4096u + (uint32_t)lane * 16u;
  auto bload = [&](int ks, int s, f32x4 (&b)[2]) {
    const uint32_t so = (uint32_t)ks * 4096u + (uint32_t)s * 2048u;
    b[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_off, so, 0));
    b[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_off + 1024u, so, 0));
  };
  f32x4 pr[PatchA2D2::NPC], bA[2], bB[2];
  int cb = 0, tap = 0, kw = 0, eoff = 0;
  pa.load(0, pr);
  bload(0, 0, bA);
  pa.store(patch, pr);
  lds_barrier();
  bool pf_issued = false;
  const char* patch_b = reinterpret_cast<const char*>(patch);
  const uint32_t l0x = (uint32_t)(2 * h);
  auto frag_addr = [&](int m) -> uint32_t {
    const uint32_t e = (uint32_t)(pa.base[m] + eoff);
    return (e << 7) | ((l0x ^ ((e >> 1) & 7u)) << 4);
  };
  auto read_frag = [&](uint32_t a0, int s, f32x4 (&f)[2]) {
    f[0] = *reinterpret_cast<const f32x4*>(patch_b + (a0 ^ (uint32_t)(64 * s)));
    f[1] = *reinterpret_cast<const f32x4*>(patch_b + (a0 ^ (uint32_t)(64 * s + 16)));
  };
  auto mfma16 = [&](const f32x4 (&f)[2][2], const f32x4 (&b)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[m][q][t], b[q][t], acc[m][0], 0, 0, 0);
  };
  f32x4 fa[2][2], fn[2][2];
  uint32_t a0[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    a0[m] = frag_addr(m);
    read_frag(a0[m], 0, fa[m]);
  }
  for (int ks = 0; ks < KS; ++ks) {
    const bool more = ks + 1 < KS;
    bload(ks, 1, bB);
    if (!pf_issued && tap >= PATCH_PF_TAP && (cb + 1) * 9 < KS) {
      pa.load(cb + 1, pr);
      pf_issued = true;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) read_frag(a0[m], 1, fn[m]);
    __builtin_amdgcn_sched_barrier(0);
    mfma16(fa, bA);
    bload(more ? ks + 1 : ks, 0, bA);                      // (past the last step: a harmless refetch, no branch around loads)
    if (++kw == 3) {
      kw = 0;
      eoff += PatchA2D2::SIDE - 2;
    } else {
      ++eoff;
    }
    const bool swap = ++tap == 9;
    if (!swap && more) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a0[m] = frag_addr(m);
        read_frag(a0[m], 0, fa[m]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma16(fn, bB);
    if (swap && more) {
      tap = 0;
      eoff = 0;
      ++cb;
      lds_barrier();
      pa.store(patch, pr);
      pf_issued = false;
      lds_barrier();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a0[m] = frag_addr(m);
        read_frag(a0[m], 0, fa[m]);
      }
    }
  }
  lds_barrier();
}

template <bool YSUB>      // YSUB: the first output keeps its even pixels only (ConvArgs::y_sub)
__global__ __launch_bounds__(256, 3) void conv_t2_kernel(const ConvArgs a, int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float t2_smem[];
  const int tile = xcd_remap((int)blockIdx.x, ntiles);
  const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
  const int m0 = mt * TileT2::BM, n0 = nt * TileT2::BN;
  f32x16 acc[2][1];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;
  {
    const PatchA2D2 pa(a, m0);
    t2_mainloop(pa, a, n0, t2_smem, acc);
  }
  EpiRes<TileT2> er;
  conv_epilogue_fast<TileT2, true, YSUB>(a, acc, m0, n0, t2_smem, er, false);
}

// ------------------------------------------------------------------------------------------
// conv_tn_kernel (round 4): the linear-patch 3x3 / stride 1 layers with a 64-pixel x 128-CHANNEL tile per block -- a wave
// owns 32 pixels x 64 channels (two column fragments) -- one whole tile per block.  The same idea as conv_t2_kernel turned
// by ninety degrees: an A fragment read from the LDS patch feeds 32 MFMAs instead of 16, and the patch (its fill, its swaps,
// the tile's set-up) is paid once per 128 channels instead of once per 64 -- on IResNet's 28 x 28 stage (128 channels)
// conv_bdp_kernel loads every patch twice, once per column tile.  B-direct mainloop, bit-identical accumulation order.
using TileTN = Tile<1, 2, 2, 2>;

template <int EMAX>
__device__ __forceinline__ void tn_mainloop(const PatchA<TileTN, EMAX>& pa, const ConvArgs& a, int n0, float* patch,
                                            f32x16 (&acc)[1][2]) {
  using PA = PatchA<TileTN, EMAX>;
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int KS = a.Kpad / BK;
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w_frag, a.w_frag_bytes);
  uint32_t lane_off[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
    lane_off[n] = (uint32_t)((n0 >> 5) + TileTN::wave_col() * 2 + n) * (uint32_t)KS * 4096u + (uint32_t)lane * 16u;
  auto bload = [&](int ks, int s, f32x4 (&b)[2][2]) {
    const uint32_t so = (uint32_t)ks * 4096u + (uint32_t)s * 2048u;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      b[n][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_off[n], so, 0));
      b[n][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_off[n] + 1024u, so, 0));
    }
  };
  f32x4 pr[PA::NPC], bA[2][2], bB[2][2];
  int cb = 0, tap = 0, kw = 0, eoff = 0;
  pa.load(0, pr);
  bload(0, 0, bA);
  pa.store(patch, pr);
  lds_barrier();
  bool pf_issued = false;
  const char* patch_b = reinterpret_cast<const char*>(patch);
  const uint32_t l0x = (uint32_t)(2 * h);
  auto frag_addr = [&]() -> uint32_t {
    const uint32_t e = (uint32_t)(pa.base[0] + eoff);
    return (e << 7) | ((l0x ^ ((e >> 1) & 7u)) << 4);
  };
  auto read_frag = [&](uint32_t a0, int s, f32x4 (&f)[2]) {
    f[0] = *reinterpret_cast<const f32x4*>(patch_b + (a0 ^ (uint32_t)(64 * s)));
    f[1] = *reinterpret_cast<const f32x4*>(patch_b + (a0 ^ (uint32_t)(64 * s + 16)));
  };
  auto mfma16 = [&](const f32x4 (&f)[2], const f32x4 (&b)[2][2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[q][t], b[n][q][t], acc[0][n], 0, 0, 0);
  };
  f32x4 fa[2], fn[2];
  uint32_t a0 = frag_addr();
  read_frag(a0, 0, fa);
  for (int ks = 0; ks < KS; ++ks) {
    const bool more = ks + 1 < KS;
    bload(ks, 1, bB);
    if (!pf_issued && tap >= PATCH_PF_TAP && (cb + 1) * 9 < KS) {
      pa.load(cb + 1, pr);
      pf_issued = true;
    }
    read_frag(a0, 1, fn);
    __builtin_amdgcn_sched_barrier(0);
    mfma16(fa, bA);
    bload(more ? ks + 1 : ks, 0, bA);
    if (++kw == 3) {
      kw = 0;
      eoff += pa.WP - 2;
    } else {
      ++eoff;
    }
    const bool swap = ++tap == 9;
    if (!swap && more) {
      a0 = frag_addr();
      read_frag(a0, 0, fa);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma16(fn, bB);
    if (swap && more) {
      tap = 0;
      eoff = 0;
      ++cb;
      lds_barrier();
      pa.store(patch, pr);
      pf_issued = false;
      lds_barrier();
      a0 = frag_addr();
      read_frag(a0, 0, fa);
    }
  }
  lds_barrier();
}

template <int EMAX, bool YSUB>
__global__ __launch_bounds__(256, 4) void conv_tn_kernel(const ConvArgs a, int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float tn_smem[];
  const int tile = xcd_remap((int)blockIdx.x, ntiles);
  const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
  const int m0 = mt * TileTN::BM, n0 = nt * TileTN::BN;
  f32x16 acc[1][2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][n][r] = 0.f;
  {
    const PatchA<TileTN, EMAX> pa(a, m0);
    tn_mainloop<EMAX>(pa, a, n0, tn_smem, acc);
  }
  EpiRes<TileTN> er;
  conv_epilogue_fast<TileTN, false, YSUB>(a, acc, m0, n0, tn_smem, er, false);
}

static int patch_applies(const ConvArgs& a);
static bool tn_applies(const ConvArgs& a, int emax) {
  if ((a.off & CONV_OFF_TN) || emax == 0 || !a.w_frag || (a.off & CONV_OFF_BD) || a.Cout % 128 != 0 || a.pre_scale) return false;
  if (emax == PATCH_EMAX_S && (a.dbg & 256)) return false;          // (dbg bit 256: the 128-entry-patch layers stay on conv_bdp_kernel: A/B)
  if ((a.dbg & 1024) || !(a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0 && a.y_ld == a.Cout && a.y_coff == 0)) return false;
  if (a.res && (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo)) return false;
  return (int64_t)a.M * a.Cout * 4 < 0xFFFFFFF0LL;
}

template <int EMAX>
static int launch_conv_tn(const ConvArgs& a, hipStream_t st) {
  constexpr int epi = TileTN::BM * (TileTN::BN + 4) * 4;
  constexpr int lds = epi > EMAX * 128 ? epi : EMAX * 128;
  if (allow_dynamic_lds(reinterpret_cast<const void*>(conv_tn_kernel<EMAX, false>), lds)) return -1;
  if (allow_dynamic_lds(reinterpret_cast<const void*>(conv_tn_kernel<EMAX, true>), lds)) return -1;
  const int tiles_m = (a.M + TileTN::BM - 1) / TileTN::BM, tiles_n = a.Cout / TileTN::BN;
  const int64_t ntiles = (int64_t)tiles_m * tiles_n;
  if (ntiles >= 0x7fffffffLL) return set_error("conv: too many tiles");
  ConvArgs b = a;
  b.fd_howo = make_fastdiv(a.Ho * a.Wo);
  b.fd_wo = make_fastdiv(a.Wo);
  b.fd_wp = make_fastdiv(a.W + 2);
  b.fd_rpi = make_fastdiv(a.H + 1);
  b.epi_fast = 1;
  if (a.y_sub) hipLaunchKernelGGL((conv_tn_kernel<EMAX, true>), dim3((unsigned)ntiles), dim3(256), lds, st, b, tiles_n, (int)ntiles);
  else hipLaunchKernelGGL((conv_tn_kernel<EMAX, false>), dim3((unsigned)ntiles), dim3(256), lds, st, b, tiles_n, (int)ntiles);
  DIF_HIP(hipGetLastError());
  g_last_kernel = EMAX == PATCH_EMAX_X ? "conv_tn_kernel<64x128,patch256+Bdirect>"
                                       : (EMAX == PATCH_EMAX_L ? "conv_tn_kernel<64x128,patch168+Bdirect>" : "conv_tn_kernel<64x128,patch128+Bdirect>");
  return 0;
}

// (conv_tnk_kernel -- round 5: the small-batch 3x3 layers on 64 x 128 tiles x S shares, ONE block per CU, a three-K-step register
// ring of weights per wave, two LDS patches, conv_sk_reduce_kernel behind it; bit-identical to the 64 x 64 pair up to the
// split -- was built, measured and removed: its blocks live 18 us for 12 us of matrix work, the same as conv_skp_kernel's once
// that kernel's blocks are dealt evenly (sk_item), and it loses where tiles x S misses the CU count (batch 12: 4.37 vs 4.06 ms).
// profiles/r05_ablation.txt item 14 keeps the numbers and what its ISA taught about `break` in an unrolled ring loop.)

constexpr int SK2_MIN_KS = 4;          // fewest K-steps a split-K share may hold (sk2_plan)

// conv_t2_kernel's layers: the 8x8-tile patch layers with a SHORT K loop (the long ones have conv_bdp_kernel), weights in
// fragment order, whole 64-channel column tiles, the lean epilogue's plain geometry (no sub-sampled first output)
static bool t2_applies(const ConvArgs& a) {
  if (a.off & CONV_OFF_T2) return false;                    // Net option "t2" = 0
  if (!patch2d_applies(a) || !a.w_frag || (a.off & CONV_OFF_BD) || a.Cout % 64 != 0 || a.Kpad / BK >= 32) return false;
  if ((a.dbg & 1024) || !(a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0 && a.y_ld == a.Cout && a.y_coff == 0)) return false;
  if (a.res && (a.res_stride != 1 || a.res_H != a.Ho || a.res_W != a.Wo)) return false;
  if ((int64_t)a.M * a.Cout * 4 >= 0xFFFFFFF0LL || (int64_t)2 * a.H * a.W * a.Cin * 4 >= 0x7fffffffLL) return false;
  return a.M % 64 == 0;
}

static int launch_conv_t2(const ConvArgs& a, hipStream_t st) {
  constexpr int lds = TileT2::BM * (TileT2::BN + 4) * 4 > PatchA2D2::EMAX * 128 ? TileT2::BM * (TileT2::BN + 4) * 4 : PatchA2D2::EMAX * 128;
  if (allow_dynamic_lds(reinterpret_cast<const void*>(conv_t2_kernel<false>), lds)) return -1;
  if (allow_dynamic_lds(reinterpret_cast<const void*>(conv_t2_kernel<true>), lds)) return -1;
  const int tiles_m = (a.M + TileT2::BM - 1) / TileT2::BM, tiles_n = a.Cout / TileT2::BN;
  const int64_t ntiles = (int64_t)tiles_m * tiles_n;
  if (ntiles >= 0x7fffffffLL) return set_error("conv: too many tiles");
  ConvArgs b = a;
  b.fd_t2_w = make_fastdiv(a.W / 8);
  b.fd_t2_img = make_fastdiv((a.H / 8) * (a.W / 8));
  b.fd_howo = make_fastdiv(a.Ho * a.Wo);
  b.fd_wo = make_fastdiv(a.Wo);
  b.epi_fast = 1;
  if (a.y_sub) hipLaunchKernelGGL(conv_t2_kernel<true>, dim3((unsigned)ntiles), dim3(256), lds, st, b, tiles_n, (int)ntiles);
  else hipLaunchKernelGGL(conv_t2_kernel<false>, dim3((unsigned)ntiles), dim3(256), lds, st, b, tiles_n, (int)ntiles);
  DIF_HIP(hipGetLastError());
  g_last_kernel = "conv_t2_kernel<128x64,2x(patch8x8)+Bdirect>";
  return 0;
}

// (conv_pw_kernel -- round 4's barrier-free pointwise GEMM: wave-private LDS-DMA ring, weights in fragment order, bit-identical,
// +2 % on one lane and -8..10 % inside the two-lane executor, never the default -- was removed in round 5: its measurements
// stay in profiles/r04_ablation.txt item 3 and DESIGN.md section 5; the per-step barrier is not what the pointwise layers wait for.)

// ---- the small-batch split-K path (conv_splitk.hpp)
// Few tiles with a long K loop: S splits per tile so that the grid is about two blocks per CU, each share at least
// SK2_MIN_KS K-steps; 0 = the path does not apply.  A pure function of the layer's shape and the batch (deterministic sums).
static int sk2_plan(const ConvArgs& a) {
  if ((a.off & CONV_OFF_SK2) || (a.trace && !(a.dbg & 256))) return 0;   // (the clock measurement keeps round 4's kernels; the block trace, dbg 256, sees this path)
  const bool pw1 = a.KH == 1 && a.KW == 1 && a.pad_t == 0 && a.pad_l == 0 && a.Cin % BK == 0;
  if (!pw1 && !(a.k_order == 1 && a.Cin % BK == 0)) return 0;          // the loaders' two K layouts (conv_splitk.hpp)
  const int64_t tiles = ((a.M + 63) / 64) * (int64_t)((a.Cout + 63) / 64);
  const int KS = a.Kpad / BK;
  const int64_t cus = num_cus();
  if (tiles >= 2 * cus || KS < 2 * SK2_MIN_KS) return 0;
  if (const int force = (a.dbg >> 16) & 63) return force == 1 ? 0 : (force <= KS / SK2_MIN_KS ? force : 0);   // (dbg bits 16..21: S for every candidate layer, 1 = no split: the S sweep)
  const int64_t cap = (int64_t)a.sk_max_blocks * (int64_t)conv_slab_floats() / 4096;   // slabs the workspace holds
  // S minimising  rounds(S) x share(S) + 0.1 S  in K-steps: rounds = blocks on the busiest CU (blocks are dealt evenly --
  // sk_item -- and co-resident blocks share the CU's matrix pipes, so their K-steps add up), share = K-steps per block, and
  // every split costs the reduce launch one more slab to read; at most four blocks per CU (what a CU holds at once).
  int64_t S = 0;
  double best = 1e30;
  for (int64_t s = 2; s <= KS / SK2_MIN_KS && tiles * s <= 4 * cus && tiles * s <= cap; ++s) {
    const double cost = (double)((tiles * s + cus - 1) / cus) * (double)((KS + s - 1) / s) + 0.1 * (double)s;
    if (cost < best) {
      best = cost;
      S = s;
    }
  }
  // (against no split at all: rounds(1) x KS; a tie goes to no split)
  if (S && (double)((tiles + cus - 1) / cus) * (double)KS <= best) S = 0;
  // Where that says "no split" with more than half the CUs busy, and the chip is this forward's alone (one lane), a second
  // look with two things the first model leaves out:
  //   * between one and two tiles per CU no split within four blocks per CU beats the whole tiles on rounds x share (392 tiles --
  //     the 14 x 14 stage at batch 32, the 28 x 28 stage at batch 16: two rounds either way) -- and yet half the CUs idle through
  //     the second round.  Up to eight blocks per CU here; a grid beyond what the CUs hold at once runs in waves, each charged
  //     6 K-steps' time for its ramp and drain.  Three shares of those layers: batch 16 5.23 -> 4.77 ms, batch 32 8.25 -> 7.86;
  //   * just under one tile per CU (248 tiles: the 14 x 14 stage at batch 20) every block is alone on its CU and hides no load
  //     latency: priced at 1.15 x its K-steps, two shares win: batch 20 6.82 -> 5.71 ms (IResNet-50 3.40 -> 2.90).
  // (IResNet-100; r05_ablation item 14 has the sweeps and the wider rules that lost.)
  if (S == 0 && 2 * tiles > cus && a.lanes <= 1 && !(a.dbg & 32768)) {         // (dbg bit 32768: off; bit 4194304: without the `alone` factor; A/B)
    auto cost_of = [&](int64_t s) {
      const int64_t blocks = tiles * s;
      // a block with no neighbour on its CU hides no latency (priced on long K loops only: ResNet-50V2's short ones lose 4 % split)
      const double alone = (blocks <= cus && KS >= 32 && !(a.dbg & 4194304)) ? 1.15 : 1.0;
      return alone * (double)((blocks + cus - 1) / cus) * (double)((KS + s - 1) / s) + 6.0 * (double)((blocks + 4 * cus - 1) / (4 * cus)) +
             (s > 1 ? 0.1 * (double)s : 0.0);
    };
    best = cost_of(1);
    for (int64_t s = 2; s <= KS / SK2_MIN_KS && tiles * s <= 8 * cus && tiles * s <= cap; ++s) {
      const double cost = cost_of(s);
      if (cost < best) {
        best = cost;
        S = s;
      }
    }
  }
  return S >= 2 ? (int)S : 0;
}

// (Requesting so much LDS per block that no CU can hold more than the plan's rounds = ceil(blocks / CUs) of them -- to make
// the dispatcher deal a ~2-blocks-per-CU grid evenly: block lives of one 500-block launch range 12 .. 25 us -- was measured:
// the longest life drops (25 -> 20 us) but blocks then WAIT for a slot (span 25 -> 29 us), and batch 1 loses 10 %:
// profiles/r05_ablation.txt item 2.)
template <bool PRE, int AM>
static int launch_conv_sk(const ConvArgs& a, int S, hipStream_t st) {
  using T = Tile<1, 1>;
  void (*kern)(const ConvArgs, int, int, int) = conv_sk_kernel<T, PRE, AM>;
  if (allow_dynamic_lds(reinterpret_cast<const void*>(kern), T::LDS_BYTES)) return -1;
  const int tiles_m = (a.M + 63) / 64, tiles_n = (a.Cout + 63) / 64;
  ConvArgs b = a;
  b.fd_howo = make_fastdiv(a.Ho * a.Wo);
  b.fd_wo = make_fastdiv(a.Wo);
  b.fd_cin = make_fastdiv(a.Cin);
  b.fd_kw = make_fastdiv(a.KW);
  b.fd_taps = make_fastdiv(a.KH * a.KW);
  if (a.k_order == 1 && a.Cin % BK != 0) return set_error("conv: channel-block-major K order needs Cin %% 32 == 0");
  const unsigned grid = sk_grid(tiles_m, tiles_n * S);
  if ((int64_t)grid > a.sk_max_blocks) b.trace = nullptr;     // (the trace buffer holds that many records per launch)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(T::NT), T::LDS_BYTES, st, b, S, tiles_m, tiles_n);
  DIF_HIP(hipGetLastError());
  hipLaunchKernelGGL(conv_sk_reduce_kernel, dim3((unsigned)(tiles_m * tiles_n * 16)), dim3(64), 0, st, b, S, tiles_n);
  DIF_HIP(hipGetLastError());
  {                                                         // interned: Op::ran_kernel keeps the pointer
    static std::mutex mu;
    static std::set<std::string> names;
    std::lock_guard<std::mutex> lock(mu);
    g_last_kernel = names.insert(std::string("conv_sk_kernel<64x64,") + am_form(AM) + (PRE ? ",preact" : "") + ",S=" +
                                 std::to_string(S) + ">+reduce").first->c_str();
  }
  return 0;
}

// ---- the one-image path (conv_minitile.hpp): waves per block, or 0 when the layer is not for it
static int mt_plan(const ConvArgs& a) {
  if ((a.off & CONV_OFF_MT) || a.trace || !a.w_f16) return 0;
  const bool pw1 = a.KH == 1 && a.KW == 1 && a.pad_t == 0 && a.pad_l == 0 && a.Cin % BK == 0;
  if (!pw1 && !(a.k_order == 1 && a.Cin % BK == 0)) return 0;
  if ((int64_t)a.N * a.H * a.W * a.Cin * 4 >= 0x7fffffffLL || (int64_t)16 * a.Kpad * 4 >= 0x7fffffffLL) return 0;   // 32-bit offsets
  const int64_t tiles = ((a.M + 15) / 16) * (int64_t)((a.Cout + 15) / 16);
  const int64_t traffic = tiles * 32 * (int64_t)a.Kpad * 4;          // L2 -> CU bytes: no operand is reused inside a tile
  // (the 784 tiles of a 56 x 56 x 64 3x3 layer: split-K is faster; the 784 of ResNet-50V2's 28 x 28 stage's 1x1 layers, K = 64:
  // this kernel is, 0.460 -> 0.455 ms per forward -- r05_ablation item 11)
  const int64_t tile_cap = (pw1 && a.Kpad <= 256 ? 4 : 2) * (int64_t)num_cus();
  if (tiles > tile_cap || traffic > (96ll << 20)) return 0;
  const int nch = a.Kpad / 16;
  const bool pre = a.pre_scale != nullptr;
  int nw = nch <= 4 * mt_round(4, pre) ? 4 : 8;                      // a wave's run in one round of loads where eight waves allow it
  while (nw < 16 && tiles * nw < 2 * (int64_t)num_cus() && nch >= 4 * nw) nw *= 2;   // few tiles: more waves per tile
  return nw;
}

template <int NW, bool PRE, bool PW>
static int launch_conv_mt_t(const ConvArgs& a, hipStream_t st) {
  const int tiles_m = (a.M + 15) / 16, tiles_n = (a.Cout + 15) / 16;
  ConvArgs b = a;
  b.fd_howo = make_fastdiv(a.Ho * a.Wo);
  b.fd_wo = make_fastdiv(a.Wo);
  b.fd_kw = make_fastdiv(a.KW);
  b.fd_taps = make_fastdiv(a.KH * a.KW);
  const unsigned grid = sk_grid(tiles_m, tiles_n);
  // (asking for more than half a CU's LDS, so that no CU is dealt two blocks while another idles: no effect, r05_ablation item 5)
  hipLaunchKernelGGL((conv_mt_kernel<NW, PRE, PW>), dim3(grid), dim3(NW * 64), 0, st, b, tiles_m, tiles_n);
  DIF_HIP(hipGetLastError());
  static const std::string label = std::string("conv_mt_kernel<16x16,") + (PW ? "pointwise" : "gather") + (PRE ? ",preact" : "") +
                                   "," + std::to_string(NW) + " waves>";
  g_last_kernel = label.c_str();
  return 0;
}
template <int NW>
static int launch_conv_mt_n(const ConvArgs& a, hipStream_t st) {
  const bool pw1 = a.KH == 1 && a.KW == 1 && a.pad_t == 0 && a.pad_l == 0 && a.Cin % BK == 0;
  if (a.pre_scale) return pw1 ? launch_conv_mt_t<NW, true, true>(a, st) : launch_conv_mt_t<NW, true, false>(a, st);
  return pw1 ? launch_conv_mt_t<NW, false, true>(a, st) : launch_conv_mt_t<NW, false, false>(a, st);
}
static int launch_conv_mt(const ConvArgs& a, int nw, hipStream_t st) {
  return nw == 4 ? launch_conv_mt_n<4>(a, st) : (nw == 8 ? launch_conv_mt_n<8>(a, st) : launch_conv_mt_n<16>(a, st));
}

template <int AMP>
static int launch_conv_skp(const ConvArgs& a, int S, hipStream_t st) {
  using T = Tile<1, 1>;
  void (*kern)(const ConvArgs, int, int, int) = conv_skp_kernel<T, AMP>;
  constexpr int lds_bytes = (AMP == 5 ? PATCH_EMAX_L : PATCH_EMAX_S) * 128;
  if (allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes)) return -1;
  const int tiles_m = (a.M + 63) / 64, tiles_n = (a.Cout + 63) / 64;
  ConvArgs b = a;
  b.fd_howo = make_fastdiv(a.Ho * a.Wo);
  b.fd_wo = make_fastdiv(a.Wo);
  b.fd_wp = make_fastdiv(a.W + 2);
  b.fd_rpi = make_fastdiv(a.H + 1);
  const unsigned grid = sk_grid(tiles_m, tiles_n * S);
  if ((int64_t)grid > a.sk_max_blocks) b.trace = nullptr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(T::NT), lds_bytes, st, b, S, tiles_m, tiles_n);
  DIF_HIP(hipGetLastError());
  hipLaunchKernelGGL(conv_sk_reduce_kernel, dim3((unsigned)(tiles_m * tiles_n * 16)), dim3(64), 0, st, b, S, tiles_n);
  DIF_HIP(hipGetLastError());
  {
    static std::mutex mu;
    static std::set<std::string> names;
    std::lock_guard<std::mutex> lock(mu);
    g_last_kernel = names.insert(std::string("conv_skp_kernel<64x64,") + am_form(AMP + 10) + ",S=" + std::to_string(S) + ">+reduce").first->c_str();
  }
  return 0;
}

template <class T>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  if constexpr (T::BM == 64 && T::BN == 64) {
    if (const int nw = mt_plan(a)) return launch_conv_mt(a, nw, st);
    if (const int S = sk2_plan(a)) {
      // 3x3 / stride 1 layers whose linear halo patch fits: the B-direct patch mainloop (dbg bit 16384 keeps the gather: A/B)
      const int emax = (!(a.off & CONV_OFF_BD) && a.w_frag && !(a.dbg & 16384)) ? patch_applies(a) : 0;
      if (emax == PATCH_EMAX_S) return launch_conv_skp<3>(a, S, st);
      if (emax == PATCH_EMAX_L) return launch_conv_skp<5>(a, S, st);
      const bool pw1 = a.KH == 1 && a.KW == 1 && a.pad_t == 0 && a.pad_l == 0 && a.Cin % BK == 0;
      if (pw1) return a.pre_scale ? launch_conv_sk<true, 1>(a, S, st) : launch_conv_sk<false, 1>(a, S, st);
      return a.pre_scale ? launch_conv_sk<true, 0>(a, S, st) : launch_conv_sk<false, 0>(a, S, st);
    }
  }
  // (LDS-DMA operand staging was +1..5 % on the plain GEMM microbenchmark but -1.2 % inside this kernel -- interleaved
  // A/B, both networks -- so operands are staged through registers.)
  // 1x1 / no padding / whole 32-channel K-steps: the pointwise loader
  constexpr bool kDefaultTile = (T::BM == 64 && T::BN == 64);
  const bool pw = a.KH == 1 && a.KW == 1 && a.pad_t == 0 && a.pad_l == 0 && a.Cin % BK == 0;
  // (a compile-time specialisation of the 3x3 gather, KMODE 2, was measured 5 % SLOWER than the
  // run-time-selected path on IResNet-100 -- hipcc schedules the loop differently -- so the
  // multi-tap layers stay on the general loader)
  if constexpr (kDefaultTile) {
    const int64_t tiles = ((a.M + T::BM - 1) / T::BM) * (int64_t)((a.Cout + T::BN - 1) / T::BN);
    int64_t slots = 4 * (int64_t)num_cus();
    if (slots > a.sk_max_blocks) slots = a.sk_max_blocks;
    // (the 8x8-tile patch form ahead of the pipelined kernel, which took the short-K 3x3 layers -- 64 input channels -- before)
    const bool bd = !(a.off & CONV_OFF_BD) && a.w_frag != nullptr;
    if (patch2d_applies(a)) {
      if (t2_applies(a)) return launch_conv_t2(a, st);
      if (bd && conv_bdp_ok(a)) return launch_conv_bdp<T, 6>(a, st);
      return bd ? launch_conv_pre<T, false, 16>(a, st) : launch_conv_pre<T, false, 6>(a, st);
    }
    // the deferred-epilogue patch kernel before the pipelined (gather) one: with very many tiles (batch 512 on one lane)
    // the 128-channel 28x28 layers qualify for both
    if (bd && !pw) {
      // conv_tn_kernel where conv_bdp_kernel would have run (several tiles per resident block): with fewer tiles the 64 x 64
      // one-tile-per-block kernel has twice as many blocks to balance (ResNet-50V2's 3x3 layers: 4.93 -> 5.42 ms with it)
      int emax_tn = patch_applies(a);
      // maps 29 .. 59 wide (YOLOv3's 52 x 52 stage): a 64-pixel tile's patch needs up to 71 + 3 (W + 2) entries -- more than the
      // other patch kernels' LDS budget holds, inside this one's (the epilogue's staging tile is larger than the patch)
      if (emax_tn == 0 && patch_shape(a) && patch_entry_bound(a, 64) <= PATCH_EMAX_X && !(a.dbg & 8192)) emax_tn = PATCH_EMAX_X;
      if (tn_applies(a, emax_tn) && (conv_bdp_ok(a) || (a.dbg & 512)))
        return emax_tn == PATCH_EMAX_X ? launch_conv_tn<PATCH_EMAX_X>(a, st)
                                       : (emax_tn == PATCH_EMAX_L ? launch_conv_tn<PATCH_EMAX_L>(a, st) : launch_conv_tn<PATCH_EMAX_S>(a, st));
    }
    if (bd && !pw && conv_bdp_ok(a)) {
      const int emax = patch_applies(a);
      if (emax == PATCH_EMAX_S) return launch_conv_bdp<T, 3>(a, st);
      if (emax == PATCH_EMAX_L) return launch_conv_bdp<T, 5>(a, st);
    }
    if (a.Cin % 4 == 0 && pipe_applies(a, tiles, a.Kpad / BK, slots)) {
      // pointwise layers retire the previous tile four chunks per K-step, i.e. within the first two
      // steps (measured best for every K: 2 steps +16 %, 8 steps +5 % over one chunk per step); the
      // multi-tap layers (18 steps) one chunk per step
      if (pw && a.pre_scale) return launch_conv_pipe<T, true, 1, 4>(a, st);
      if (pw) return launch_conv_pipe<T, false, 1, 4>(a, st);
      if (!a.pre_scale) return launch_conv_pipe<T, false, 0, 1>(a, st);
    }
  }
  if (pw && a.pre_scale) return launch_conv_pre<T, true, 1>(a, st);
  if (pw) return launch_conv_pre<T, false, 1>(a, st);
  if constexpr (kDefaultTile) {
    {
      const bool bd2 = !(a.off & CONV_OFF_BD) && a.w_frag != nullptr;
      const int emax = patch_applies(a);
      if (bd2 && conv_bdp_ok(a)) {
        if (emax == PATCH_EMAX_S) return launch_conv_bdp<T, 3>(a, st);
        if (emax == PATCH_EMAX_L) return launch_conv_bdp<T, 5>(a, st);
        if (patch2d_applies(a)) return launch_conv_bdp<T, 6>(a, st);
      }
      if (emax == PATCH_EMAX_S) return bd2 ? launch_conv_pre<T, false, 13>(a, st) : launch_conv_pre<T, false, 3>(a, st);
      if (emax == PATCH_EMAX_L) return bd2 ? launch_conv_pre<T, false, 15>(a, st) : launch_conv_pre<T, false, 5>(a, st);
      if (patch2d_applies(a)) return bd2 ? launch_conv_pre<T, false, 16>(a, st) : launch_conv_pre<T, false, 6>(a, st);
    }
  }
  if (a.pre_scale) return launch_conv_pre<T, true, 0>(a, st);
  return launch_conv_pre<T, false, 0>(a, st);
}

template <class T, bool PRE, int AM, int BF3, int LEAN>
static int launch_conv_pre_impl(const ConvArgs& a, hipStream_t st);

template <class T, bool PRE, int AM, int BF3>
static int launch_conv_pre(const ConvArgs& a, hipStream_t st) {
  // the lean epilogue's case (conv_epilogue_fast)
  const bool lean = !(a.dbg & 1024) && a.y_H == a.Ho && a.y_W == a.Wo && a.y_oy == 0 && a.y_ox == 0 && a.y_ld == a.Cout &&
                    a.y_coff == 0 && !a.y_sub && a.Cout % 4 == 0 &&
                    (!a.res || (a.res_stride == 1 && a.res_H == a.Ho && a.res_W == a.Wo)) && (int64_t)a.M * a.Cout * 4 < 0xFFFFFFF0LL;
  if constexpr (BF3) return a.y_sub ? launch_conv_pre_impl<T, PRE, AM, BF3, 2>(a, st) : launch_conv_pre_impl<T, PRE, AM, BF3, 1>(a, st);
  else return lean ? launch_conv_pre_impl<T, PRE, AM, BF3, 1>(a, st) : launch_conv_pre_impl<T, PRE, AM, BF3, 0>(a, st);
}

template <class T, bool PRE, int AM, int BF3, int LEAN>
static int launch_conv_pre_impl(const ConvArgs& a, hipStream_t st) {
  // (the two-term split-bf16 tier is a kernel of its own: conv_igemm_kernel.hpp)
  void (*kern)(const ConvArgs) = conv_igemm_kernel<T, PRE, AM, BF3, LEAN>;
  if constexpr (BF3 != 0) {
    if (a.bf_terms == 2) kern = conv_igemm_bf2_kernel<T, PRE, AM, BF3, LEAN>;
  }
  constexpr int AMP = AM % 10;
  constexpr int emax = AMP == 3 ? PATCH_EMAX_S : (AMP == 5 ? PATCH_EMAX_L : 100);
  constexpr int epi_bytes = T::BM * (T::BN + 4) * 4;       // the epilogue's staging tile
  // B-direct: the patch alone (or the epilogue's staging tile if that is larger)
  constexpr int bd_bytes = emax * 128 > epi_bytes ? emax * 128 : epi_bytes;
  constexpr int b3p_bytes = bf3p_lds_b(T::BM) > epi_bytes ? bf3p_lds_b(T::BM) : epi_bytes;
  constexpr int lds_bytes = BF3 ? b3p_bytes
                                : (AM >= 10 ? bd_bytes : ((AMP == 3 || AMP == 5 || AMP == 6) ? patch_lds_bytes(emax) : T::LDS_BYTES));
  if (allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes)) return -1;
  const int64_t tiles = ((a.M + T::BM - 1) / T::BM) * (int64_t)((a.Cout + T::BN - 1) / T::BN);
  const int KS = (BF3 && AM >= 10) ? a.Kpad / (BK * 9) : a.Kpad / BK;     // stream-K units (the kernel's `KS`)
  const int64_t I = tiles * KS;
  if (I >= 0x7fffffffLL) return set_error("conv: iteration space too large");
  // Resident blocks for this tile shape (LDS-limited: 2 per CU, 4 for the 64x64 tile).
  int64_t slots = (BF3 ? 512 / T::NT : T::BLOCKS_PER_CU) * (int64_t)num_cus();     // split-bf16: eight waves per CU
  if (slots > a.sk_max_blocks) slots = a.sk_max_blocks;
  // Many tiles, or a short K loop (< 32 steps: a split tile's slab hand-off would cost more
  // than the imbalance it removes -- measured): one whole tile per block, the hardware
  // dispatcher balances them and no tile is ever split.  Few long tiles: stream-K, one equal
  // K-step range per resident block.
  int64_t P;
  // (the two-term split-bf16 tier: whole tiles from one round of blocks up -- as for conv_tn_kernel, one-tile blocks of the two
  // lanes interleave where persistent grids do not: 47.17 -> 46.52 ms at batch 512, 26.91 -> 26.60 at 256; the three-term tier,
  // power-limited, gains at 512 and loses at 256 and keeps stream-K; dbg bit 2048 forces it for the A/B)
  if (tiles >= 8 * slots || a.Kpad / BK < SK_MIN_KS || (((BF3 && a.bf_terms == 2) || (a.dbg & 2048)) && tiles >= slots) || (a.dbg & 4096)) {
    P = tiles;
  } else {
    P = slots;
    if (P > (I + 3) / 4) P = (I + 3) / 4;
  }
  if (P < 1) P = 1;
  ConvArgs b = a;
  b.fd_howo = make_fastdiv(a.Ho * a.Wo);
  b.fd_wo = make_fastdiv(a.Wo);
  b.fd_cin = make_fastdiv(a.Cin);
  b.fd_kw = make_fastdiv(a.KW);
  b.fd_ks = make_fastdiv(KS);
  b.fd_taps = make_fastdiv(a.KH * a.KW);
  b.fd_wp = make_fastdiv(a.W + 2);
  b.fd_rpi = make_fastdiv(a.H + 1);
  b.fd_t2_w = make_fastdiv(a.W / 8 > 0 ? a.W / 8 : 1);
  b.fd_t2_img = make_fastdiv((a.H / 8) * (a.W / 8) > 0 ? (a.H / 8) * (a.W / 8) : 1);
  if (a.k_order == 1 && a.Cin % BK != 0) return set_error("conv: channel-block-major K order needs Cin %% 32 == 0");
  b.fd_tiles_n = make_fastdiv((a.Cout + T::BN - 1) / T::BN);
  b.epi_fast = LEAN != 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)P), dim3(T::NT), lds_bytes, st, b);
  DIF_HIP(hipGetLastError());
  static const std::string label = kernel_label<T>(BF3 ? "conv_igemm_kernel[split-bf16]" : "conv_igemm_kernel", AM, PRE ? ",preact" : "");
  g_last_kernel = label.c_str();
  return 0;
}

// The general f32 kernels ship the 64x64 tile (four blocks per CU) -- and, for layers of at most 32 output channels, 128 x 32
// (the end of conv_run).  64x64 was measured per layer over both networks against
// 128x128 / 128x64 / 64x128 at four waves (round 1: it wins or ties everywhere) and against 128x128 at eight waves
// (round 2: within 1 % either way, profiles/r02_ablation.txt), so the other f32 instantiations were dropped.
int conv_run(const ConvArgs& a, hipStream_t st) {
  if (a.M <= 0) return 0;
  if (a.Cin % 4 != 0) return set_error("conv: Cin must be a multiple of 4 (got %d)", a.Cin);
  if (a.Cout % 4 != 0) {
    if (a.res) return set_error("conv: a shortcut needs Cout %% 4 == 0 (got %d)", a.Cout);
  } else if (a.y_ld % 4 != 0 || a.y_coff % 4 != 0) {
    return set_error("conv: output view must be 16-byte aligned");
  }
  if (a.y_ld < a.y_coff + a.Cout || a.y_H < a.Ho + a.y_oy || a.y_W < a.Wo + a.y_ox)
    return set_error("conv: output view does not fit its parent tensor");
  if (a.y_sub && (a.Cout % 4 != 0 || a.y_ld != a.Cout || a.y_H != a.Ho || a.y_W != a.Wo))
    return set_error("conv: a subsampled first output needs a plain output geometry");
  if (a.Kpad % BK != 0) return set_error("conv: Kpad must be a multiple of %d", BK);
  if (a.H >= 0x3f00 || a.W >= 0x3f00) return set_error("conv: spatial size too large");
  if ((a.pre_scale == nullptr) != (a.pre_shift == nullptr)) return set_error("conv: pre_scale and pre_shift go together");
  if (!a.sk_slab || !a.sk_flag || a.sk_max_blocks < 1) return set_error("conv: stream-K workspace missing");
  {
    const int64_t howo = (int64_t)a.Ho * a.Wo;
    const int64_t span = (256 + howo - 1) / howo + 1;
    if (span * a.H * a.W * a.Cin * 4 >= 0x7fffffffLL)
      return set_error("conv: a 256-pixel tile spans more than 2 GiB of input (%dx%dx%d)", a.H, a.W, a.Cin);
  }
  // (four waves of 128 x 32 with a six-set B ring, and eight waves on a 256 x 128 tile -- Tile<4, 1, 1, 4>, Tile<2, 2, 4, 2>: the
  // mainloop takes either -- measured the same as this one within 2 %)
  // (three or two bf16 terms per operand: ConvArgs::bf_terms, picked inside the kernel)
  const int form = bf3p_applies(a) * 2 + (a.Cout <= 64 ? 1 : 0);
  // two-term tier: waves of 128 x 32 (four row fragments share a B fragment: half the B loads per MFMA, which with half the
  // MFMAs per product is what the loop waits for; 47.6 -> 46.9 ms at batch 512, r04_ablation.txt item 7; three terms: +-0.4 %)
  if (form == 2 && (a.bf_terms == 2 || (a.dbg & 128))) return launch_conv_pre<Tile<4, 1, 1, 4>, false, 13, 1>(a, st);
  switch (form) {
    case 2: return launch_conv_pre<Tile<2, 2, 2, 2>, false, 13, 1>(a, st);     // linear patch, 128 columns
    case 3: return launch_conv_pre<Tile<2, 1, 2, 2>, false, 13, 1>(a, st);     // linear patch, 64 columns
    case 4: return launch_conv_pre<Tile<2, 2, 2, 2>, false, 16, 1>(a, st);     // two 8x8 sub-tiles, 128 columns
    case 5: return launch_conv_pre<Tile<2, 1, 2, 2>, false, 16, 1>(a, st);     // two 8x8 sub-tiles, 64 columns
    default: break;
  }
  // Layers of at most 32 output channels (MTCNN's P-Net: 10 / 16 / 32 channels over 1.7 M pixels per launch; heads of 8): a
  // 128-pixel x 32-channel tile -- four waves stacked over the pixels -- instead of 64 x 64 with half or three quarters of the
  // MFMA columns multiplying padding.  [dbg bit 8388608: the 64 x 64 tile, A/B]
  if (a.Cout <= 32 && !(a.dbg & 8388608)) return launch_conv<Tile<1, 1, 4, 1>>(a, st);
  return launch_conv<Tile<1, 1>>(a, st);
}

}  // namespace dif
