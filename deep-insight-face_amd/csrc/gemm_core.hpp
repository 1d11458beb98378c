// fp32 MFMA tile mainloop for gfx950 (MI355X), shared by the implicit-GEMM
// convolution, the embedding x gallery match and the ArcMargin logits kernels.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, bit-exact f32 fma
// chain (MI355X_MICROARCH.md "Matrix cores"); peak 157.3 TFLOP/s.  This is the
// parity path: the reference computes in float32 throughout.
//
// Geometry: WGM x WGN waves per block; each wave owns WM x WN MFMA tiles of 32x32, so
// the block tile is BM = 32*WM*WGM rows x BN = 32*WN*WGN columns and the block has
// 64*WGM*WGN threads.  K advances in steps of BK = 32 floats.  Operand tiles are staged
// global -> registers -> LDS (16-byte chunks, so the A loader can synthesise the
// zero halo of a convolution) into two LDS buffers: the loads of step k+1 are
// issued before the MFMAs of step k and written to the other buffer after them,
// one barrier per step.
//
// LDS image: [row][BK + 4] floats (144-byte rows).  A wave's ds_read_b128 then
// touches 16 distinct 16-byte slots per 16-lane group (rows distinct mod 16) --
// conflict-free (SQ_LDS_BANK_CONFLICT = 0 measured); ds_write_b128 writes whole
// 128-byte rows per 8-lane group.
//
// Operand maps (cdna_hip_programming.md section 3): A lane l holds A[i = l&31][k = l>>5],
// B lane l holds B[k = l>>5][j = l&31]; D lane l, reg r holds
// D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].  Any permutation of k applied to A
// and B alike leaves the product unchanged, so lane-half h consumes k = 16*s + 8*h + t
// (t = 0..7) in sub-step s: 32 contiguous bytes per lane, two ds_read_b128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dif {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;               // floats per K-step
constexpr int LDS_STRIDE = BK + 4;   // floats per LDS row (144 B)

template <int WM_, int WN_, int WGM_ = 2, int WGN_ = 2>
struct Tile {
  static constexpr int WM = WM_, WN = WN_, WGM = WGM_, WGN = WGN_;
  static constexpr int NT = 64 * WGM * WGN;       // threads per block
  static constexpr int BM = 32 * WM * WGM;
  static constexpr int BN = 32 * WN * WGN;
  static constexpr int RP = NT / 8;               // rows staged per pass (8 chunks of 16 B per row)
  static constexpr int NA = BM / RP;              // 16-B chunks of A per thread per K-step
  static constexpr int NB = BN / RP;
  static constexpr int LDS_FLOATS = 2 * (BM + BN) * LDS_STRIDE;
  static constexpr int LDS_BYTES = LDS_FLOATS * 4;
  // blocks per CU the kernels are built for: LDS-limited (160 KiB)
  static constexpr int BLOCKS_PER_CU = LDS_BYTES * 4 <= 160 * 1024 ? 4 : 2;
  // ... and the waves per SIMD that makes (4 blocks of 4 waves, or 2 blocks of 8 waves = 4 waves per SIMD,
  // i.e. at most 128 VGPRs): the second argument of __launch_bounds__, which on HIP is waves per
  // execution unit (MI355X_MICROARCH.md "Terms"), so the compiler holds that register line
  static constexpr int WAVES_PER_EU = BLOCKS_PER_CU * WGM * WGN / 4;
  static constexpr int MIN_BLOCKS = WAVES_PER_EU;
  static_assert(BM % RP == 0 && BN % RP == 0, "tile rows must be a multiple of the staging pass");
  __device__ static __forceinline__ int wave_row() { return (threadIdx.x >> 6) / WGN; }
  __device__ static __forceinline__ int wave_col() { return (threadIdx.x >> 6) % WGN; }
};

// D-fragment coordinates inside one 32x32 MFMA tile.
__device__ __forceinline__ int frag_row(int lane, int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int frag_col(int lane) { return lane & 31; }

// LDS-only workgroup barrier: the fences name the local address space, so the compiler waits for this wave's LDS
// operations (lgkmcnt) but leaves its global loads -- the B fragments requested ahead -- in flight across the barrier
// (__syncthreads() drains vmcnt as well -- the B prefetch at every patch swap, and every global STORE in flight: the
// epilogue's output tile, the pipelined kernels' retiring stores -- although all the kernels ever share is LDS)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Accumulates K-steps [kbeg, kend) into acc.  Ends on a barrier (LDS is free afterwards).
// Loader concept:
//   struct L { __device__ void load(int kstep, f32x4 (&r)[N]);    // issue the loads of one K-step
//              __device__ void finish(f32x4 (&r)[N]); };          // A side only: run just before the
//                                                                 // LDS write (pre-activation transform)
// Row i of the thread's share is tile row (tid>>3) + RP*i, chunk (tid&7) of the
// K-step (floats 4*(tid&7) .. +3).
template <class T, class ALoader, class BLoader>
__device__ __forceinline__ void gemm_mainloop(ALoader& al, BLoader& bl, int kbeg, int kend, float* lds,
                                              f32x16 (&acc)[T::WM][T::WN]) {
  constexpr int WM = T::WM, WN = T::WN, BM = T::BM, BN = T::BN, NA = T::NA, NB = T::NB, RP = T::RP;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();

  constexpr int BUF = (BM + BN) * LDS_STRIDE;   // floats per LDS buffer: A rows then B rows
  constexpr int OFFB = BM * LDS_STRIDE;

  // staging coordinates
  const int st_off = (tid >> 3) * LDS_STRIDE + (tid & 7) * 4;
  // fragment read coordinates
  const int fr_off = (lane & 31) * LDS_STRIDE + 8 * (lane >> 5);

  f32x4 ra[NA], rb[NB];
  al.load(kbeg, ra);
  bl.load(kbeg, rb);
  al.finish(ra);
#pragma unroll
  for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(lds + st_off + i * RP * LDS_STRIDE) = ra[i];
#pragma unroll
  for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(lds + OFFB + st_off + i * RP * LDS_STRIDE) = rb[i];
  lds_barrier();

  for (int ks = kbeg; ks < kend; ++ks) {
    const int cur = (ks - kbeg) & 1;
    const bool more = (ks + 1 < kend);
    if (more) {
      al.load(ks + 1, ra);
      bl.load(ks + 1, rb);
    }
    __builtin_amdgcn_sched_barrier(0);   // the prefetch stays above the MFMAs (see gemm_mainloop2)
    const float* pa = lds + cur * BUF + (wr * WM * 32) * LDS_STRIDE + fr_off;
    const float* pb = lds + cur * BUF + OFFB + (wc * WN * 32) * LDS_STRIDE + fr_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 fa[WM][2], fb[WN][2];
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        fa[m][0] = *reinterpret_cast<const f32x4*>(pa + m * 32 * LDS_STRIDE + 16 * s);
        fa[m][1] = *reinterpret_cast<const f32x4*>(pa + m * 32 * LDS_STRIDE + 16 * s + 4);
      }
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        fb[n][0] = *reinterpret_cast<const f32x4*>(pb + n * 32 * LDS_STRIDE + 16 * s);
        fb[n][1] = *reinterpret_cast<const f32x4*>(pb + n * 32 * LDS_STRIDE + 16 * s + 4);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int m = 0; m < WM; ++m) {
#pragma unroll
          for (int n = 0; n < WN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][t >> 2][t & 3], fb[n][t >> 2][t & 3],
                                                             acc[m][n], 0, 0, 0);
          }
        }
      }
    }
    if (more) {
      float* wa = lds + (cur ^ 1) * BUF + st_off;
      float* wb = lds + (cur ^ 1) * BUF + OFFB + st_off;
      al.finish(ra);
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(wa + i * RP * LDS_STRIDE) = ra[i];
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(wb + i * RP * LDS_STRIDE) = rb[i];
    }
    lds_barrier();
  }
}

// Variant used by the convolution kernel.  Same data movement, plus
//  * the last K-step is peeled out of the loop and preceded by `tail()`: the caller issues the
//    loads its epilogue will need (the shortcut tile) there, so their latency hides behind the
//    last 16 MFMAs -- the operand staging registers are dead by then, so this costs no VGPRs;
// (Fetching operands TWO K-steps ahead through a second set of staging registers was measured: at the
// 128-VGPR budget of four waves per SIMD it spills, and loses 5-35 % on every layer class.)
template <class T, class ALoader, class BLoader, class Tail>
__device__ __forceinline__ void gemm_mainloop2(ALoader& al, BLoader& bl, int kbeg, int kend, float* lds,
                                               f32x16 (&acc)[T::WM][T::WN], Tail&& tail) {
  constexpr int WM = T::WM, WN = T::WN, BM = T::BM, BN = T::BN, NA = T::NA, NB = T::NB, RP = T::RP;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  constexpr int BUF = (BM + BN) * LDS_STRIDE;
  constexpr int OFFB = BM * LDS_STRIDE;
  const int st_off = (tid >> 3) * LDS_STRIDE + (tid & 7) * 4;
  const int fr_off = (lane & 31) * LDS_STRIDE + 8 * (lane >> 5);

  auto mfma_step = [&](int cur) {
    const float* pa = lds + cur * BUF + (wr * WM * 32) * LDS_STRIDE + fr_off;
    const float* pb = lds + cur * BUF + OFFB + (wc * WN * 32) * LDS_STRIDE + fr_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 fa[WM][2], fb[WN][2];
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        fa[m][0] = *reinterpret_cast<const f32x4*>(pa + m * 32 * LDS_STRIDE + 16 * s);
        fa[m][1] = *reinterpret_cast<const f32x4*>(pa + m * 32 * LDS_STRIDE + 16 * s + 4);
      }
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        fb[n][0] = *reinterpret_cast<const f32x4*>(pb + n * 32 * LDS_STRIDE + 16 * s);
        fb[n][1] = *reinterpret_cast<const f32x4*>(pb + n * 32 * LDS_STRIDE + 16 * s + 4);
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int n = 0; n < WN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][t >> 2][t & 3], fb[n][t >> 2][t & 3],
                                                             acc[m][n], 0, 0, 0);
    }
  };
  auto stage = [&](int buf, f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
    float* wa = lds + buf * BUF + st_off;
    al.finish(ra);
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(wa + i * RP * LDS_STRIDE) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(wa + OFFB + i * RP * LDS_STRIDE) = rb[i];
  };

  f32x4 ra[NA], rb[NB];
  al.load(kbeg, ra);
  bl.load(kbeg, rb);
  stage(0, ra, rb);
  lds_barrier();
  int ks = kbeg;
  for (; ks + 1 < kend; ++ks) {
    const int cur = (ks - kbeg) & 1;
    al.load(ks + 1, ra);
    bl.load(ks + 1, rb);
    // keep the prefetch where it is written: left alone, the scheduler sinks these loads below the
    // MFMAs, right in front of the LDS writes that consume them, and the loop stops overlapping
    // global-load latency with matrix work at all
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(cur);
    stage(cur ^ 1, ra, rb);
    lds_barrier();
  }
  tail();
  mfma_step((ks - kbeg) & 1);
  lds_barrier();
}


template <class T>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[T::WM][T::WN]) {
#pragma unroll
  for (int m = 0; m < T::WM; ++m)
#pragma unroll
    for (int n = 0; n < T::WN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
}

// Buffer resource over [base, base + bytes): loads past the end return zeros, which is
// how tile tails and the zero halo of a convolution are produced without branches.
// Built from wave-uniform values only (kernel arguments / blockIdx), so hipcc keeps the
// descriptor in SGPRs (cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
constexpr uint32_t OOB = 0xFFFFFFF0u;   // byte offset guaranteed past any descriptor's range

// ------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") arithmetic: the throughput mode behind dif_net_set_option("bf16x3").
//
// Every f32 operand is written as hi + mid + lo, three bf16 terms obtained by repeated
// round-to-nearest (x -> hi; x - hi -> mid; x - hi - mid -> lo: 3 x 8 significand bits, |error| <= 2^-24 |x|),
// and a . b is accumulated in f32 from the six products that matter
//     lo.hi + hi.lo + mid.mid + mid.hi + hi.mid + hi.hi        (dropped: mid.lo, lo.mid, lo.lo <= 2^-24 |a||b|)
// on v_mfma_f32_32x32x16_bf16 (16 K-values per instruction at half the cycles of the f32 MFMA's 2): 16/6 = 2.67x
// the f32-MFMA ceiling for the same algorithmic work, at f32-level accuracy (tests: the same 1e-5 cosine gate).
// Not bit-identical to an f32 fma chain, which is why it is a mode and not the default.  The kernel is conv.hip's
// gemm_mainloop_patch_bf3 (3x3 / stride 1 layers; every other layer stays on the f32 kernels).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bf3_cvt2(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf3_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf3_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// four f32 -> three planes of four bf16 (8 bytes each)
__device__ __forceinline__ void bf3_split(const f32x4& v, u32x2& hi, u32x2& mid, u32x2& lo) {
  hi = u32x2{bf3_cvt2(v[0], v[1]), bf3_cvt2(v[2], v[3])};
  float r0 = v[0] - bf3_lo(hi[0]), r1 = v[1] - bf3_hi(hi[0]), r2 = v[2] - bf3_lo(hi[1]), r3 = v[3] - bf3_hi(hi[1]);
  mid = u32x2{bf3_cvt2(r0, r1), bf3_cvt2(r2, r3)};
  r0 -= bf3_lo(mid[0]);
  r1 -= bf3_hi(mid[0]);
  r2 -= bf3_lo(mid[1]);
  r3 -= bf3_hi(mid[1]);
  lo = u32x2{bf3_cvt2(r0, r1), bf3_cvt2(r2, r3)};
}


// ------------------------------------------------------------------------------------------
// Two-term split-bf16 mainloop ("bf16x2"): the search-key FILTER of the gallery match (match.hip), nowhere else.
// Both operands lie in memory ALREADY split, in the byte layout of the f32 matrix they stand for: per row and K-step of
// 32 values, [hi: 32 bf16][mid: 32 bf16] = the same 128 bytes, so the f32 row loaders and the 144-byte LDS rows serve
// unchanged.  x = hi + mid + O(2^-18 |x|); a . b ~ mid.hi + hi.mid + hi.hi (dropped: mid.mid <= 2^-18 |a||b|) on
// v_mfma_f32_32x32x16_bf16, three MFMAs of 32 cycles per 16 k against eight of 64 cycles on the f32 MFMA: 5.3x less
// matrix time for a dot product good to ~1e-5 |a||b| -- which is all a filter with a proven error bound needs.
template <class T, class ALoader, class BLoader>
__device__ __forceinline__ void gemm_mainloop_bf2(ALoader& al, BLoader& bl, int kbeg, int kend, float* lds,
                                                  f32x16 (&acc)[T::WM][T::WN]) {
  constexpr int WM = T::WM, WN = T::WN, BM = T::BM, BN = T::BN, NA = T::NA, NB = T::NB, RP = T::RP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = T::wave_row(), wc = T::wave_col();
  constexpr int BUF = (BM + BN) * LDS_STRIDE, OFFB = BM * LDS_STRIDE;
  const int st_off = (tid >> 3) * LDS_STRIDE + (tid & 7) * 4;
  const int fr_off = (lane & 31) * LDS_STRIDE + 4 * (lane >> 5);     // floats: row lane & 31, byte 16 (lane >> 5) of a plane's 16-k half
  f32x4 ra[NA], rb[NB];
  al.load(kbeg, ra);
  bl.load(kbeg, rb);
#pragma unroll
  for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(lds + st_off + i * RP * LDS_STRIDE) = ra[i];
#pragma unroll
  for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(lds + OFFB + st_off + i * RP * LDS_STRIDE) = rb[i];
  lds_barrier();
  for (int ks = kbeg; ks < kend; ++ks) {
    const int cur = (ks - kbeg) & 1;
    const bool more = ks + 1 < kend;
    if (more) {
      al.load(ks + 1, ra);
      bl.load(ks + 1, rb);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float* pa = lds + cur * BUF + (wr * WM * 32) * LDS_STRIDE + fr_off;
    const float* pb = lds + cur * BUF + OFFB + (wc * WN * 32) * LDS_STRIDE + fr_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[WM], am[WM], bh[WN], bm[WN];
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        ah[m] = *reinterpret_cast<const bf16x8*>(pa + m * 32 * LDS_STRIDE + 8 * s);
        am[m] = *reinterpret_cast<const bf16x8*>(pa + m * 32 * LDS_STRIDE + 16 + 8 * s);
      }
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        bh[n] = *reinterpret_cast<const bf16x8*>(pb + n * 32 * LDS_STRIDE + 8 * s);
        bm[n] = *reinterpret_cast<const bf16x8*>(pb + n * 32 * LDS_STRIDE + 16 + 8 * s);
      }
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bm[n], acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
    }
    if (more) {
      float* wa = lds + (cur ^ 1) * BUF + st_off;
      float* wb = lds + (cur ^ 1) * BUF + OFFB + st_off;
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(wa + i * RP * LDS_STRIDE) = ra[i];
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(wb + i * RP * LDS_STRIDE) = rb[i];
    }
    lds_barrier();
  }
}

// Row-major [rows][ld] matrix, K contiguous; rows >= nrows read as zero.  Used for packed
// convolution weights ([Cout][Kpad]), the gallery, the probes and the ArcMargin class
// centres.  `base` points at the tile's first row (block-uniform), so byte offsets stay
// below 2^32 for any tile; ld is a multiple of 4 and K is padded to a multiple of BK.
template <int N, int RP>
struct RowLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t off0;   // byte offset of (row = tid>>3, k = 4*(tid&7)) inside the tile
  uint32_t ldb;    // row pitch in bytes
  __device__ __forceinline__ RowLoader(const float* tile_base, int64_t rows_left, int ld) {
    const int tid = threadIdx.x;
    const int64_t rows = rows_left < RP * N ? rows_left : RP * N;
    rsrc = make_rsrc(tile_base, (uint32_t)(rows * ld * 4));
    ldb = (uint32_t)ld * 4u;
    off0 = (uint32_t)(tid >> 3) * ldb + (uint32_t)(tid & 7) * 16u;
  }
  __device__ __forceinline__ void finish(f32x4 (&)[N]) const {}
  __device__ __forceinline__ void load(int kstep, f32x4 (&r)[N]) const {
    const uint32_t o = off0 + (uint32_t)kstep * (BK * 4);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = buf_load4(rsrc, o + (uint32_t)i * RP * ldb);
  }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;   // destination of buffer_load ... lds (conv.hip: PatchDma)
// LDS pointer of a generic pointer KNOWN to point into LDS: its low 32 bits.  (A plain address-space cast carries a null
// check -- select on the generic pointer's high half, src_shared_base -- which this compiler folds, for some kernel
// shapes, into `v_cmp_ne_u32 vcc, 0, src_shared_base`: "Illegal instruction detected: Operand has incorrect register class".)
__device__ __forceinline__ lds_ptr_t lds_ptr_of(const void* p) {
  return (lds_ptr_t)(uintptr_t)(uint32_t)(uintptr_t)p;
}

}  // namespace dif
