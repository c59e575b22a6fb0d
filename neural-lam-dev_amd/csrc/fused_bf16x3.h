// Split-bf16 ("bf16x3") emulation of the fp32 GEMMs on the bf16 matrix cores.
//   x = hi + lo (+ 2^-18 |x|),  hi = bf16(x), lo = bf16(x - hi)
//   a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi           (dropped: a_lo b_lo ~ 2^-18 |a b|)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (32 cycles for K = 16, against 8 x 64
// cycles of v_mfma_f32_32x32x2_f32): 3/16 of the fp32 MFMA time at ~2^-16 relative error per
// product -- tighter than the TF32 the reference itself trains with on GPUs
// (train_model.py:246-248).  The accumulator layout is the fp32 one, so the row-on-lane
// kernels keep their structure.
//
// Weight image in LDS: two bf16 planes (hi, lo), natural [n][k] order, row pitch K + 4
// elements (K/2 + 2 dwords = 2 x odd: the 8-byte row reads of a 32-lane half are
// conflict-free).  One image serves W . x (row reads, 2 x ds_read_b64 per fragment) and
// W^T . g (gfx950 transposed reads, 2 x ds_read_b64_tr_b16 per fragment).
#pragma once
#include "fused_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct B3Image {
  __bf16* hi;
  __bf16* lo;
  int pitch;   // elements
  int swz;     // rows with bit 3 set start `swz` elements later (16-row kernels: fused16.h); 0: none
};
// element offset of (row, col) in either plane of an image
__device__ __forceinline__ int b3_at(const B3Image& im, int row, int col) {
  return row * im.pitch + col + im.swz * ((row >> 3) & 1);
}
__host__ __device__ constexpr int b3_pitch(int k_pad32) { return k_pad32 + 4; }
// bytes of one image with n_pad rows
__host__ __device__ constexpr size_t b3_image_bytes(int n_pad, int k_pad32) {
  return (size_t)2 * n_pad * b3_pitch(k_pad32) * sizeof(__bf16);
}
__device__ __forceinline__ B3Image b3_image(void* base, int n_pad, int k_pad32) {
  B3Image im;
  im.pitch = b3_pitch(k_pad32);
  im.swz = 0;
  im.hi = reinterpret_cast<__bf16*>(base);
  im.lo = im.hi + n_pad * im.pitch;
  return im;
}

__device__ __forceinline__ void b3_split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)v[j];
    lo[j] = (__bf16)(v[j] - (float)hi[j]);
  }
}

// W (n_out x k_in fp32, row-major) -> image rows [row0, row0 + n_pad), zero padded to
// k_pad32 columns.  float4 path needs k_in % 4 == 0 and 16-byte aligned rows.
__device__ __forceinline__ void load_weight_lds_b3(const B3Image& im, int row0,
                                                   const float* __restrict__ W, int64_t ldW,
                                                   int n_out, int k_in, int n_pad, int k_pad32,
                                                   int tid, int nthreads) {
  const bool vec = (k_in % 4 == 0) && (ldW % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0);
  if (vec) {
    const int cpr = k_pad32 >> 2;
    const int total = n_pad * cpr;
    // 16 loads in flight per thread: a 128 x 128 image (16 float4 per thread of a 256-thread
    // workgroup) arrives in ONE global round trip (the prologue is a fixed cost of every launch,
    // and most launches of the hierarchical models are small)
    constexpr int NB = 16;
    for (int base = 0; base < total; base += NB * nthreads) {
      // every load is unconditional on a clamped address and masked afterwards: a load inside
      // `ok ? load : 0` sits in its own basic block, and the register allocator then parked one
      // loaded value with `s_waitcnt vmcnt(0); v_mov` in the middle of the batch -- a full global
      // round trip of stall in every prologue (seen in the ISA of tail_fwd_kernel<128, ...>)
      f32x4 v[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int idx = base + u * nthreads + tid;
        const int i = idx / cpr, c = idx - i * cpr;
        const bool ok = idx < total && i < n_out && 4 * c < k_in;
        v[u] = *reinterpret_cast<const f32x4*>(W + (ok ? (int64_t)i * ldW + 4 * c : 0));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int idx = base + u * nthreads + tid;
        if (idx < total) {
          const int i = idx / cpr, c = idx - i * cpr;
          if (!(i < n_out && 4 * c < k_in)) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          bf16x4 hi, lo;
          b3_split4(v[u], hi, lo);
          *reinterpret_cast<bf16x4*>(im.hi + b3_at(im, row0 + i, 4 * c)) = hi;
          *reinterpret_cast<bf16x4*>(im.lo + b3_at(im, row0 + i, 4 * c)) = lo;
        }
      }
    }
  } else {
    // narrow / unaligned rows (the 2-3 wide static features of the embedders): eight loads in
    // flight per thread, then the stores -- not one global round trip per element
    const int total = n_pad * k_pad32;
    for (int base = 0; base < total; base += 8 * nthreads) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * nthreads + tid;
        const int i = idx / k_pad32, k = idx - i * k_pad32;
        const bool ok = idx < total && i < n_out && k < k_in;
        v[u] = W[ok ? (int64_t)i * ldW + k : 0];
        if (!ok) v[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * nthreads + tid;
        if (idx < total) {
          const int i = idx / k_pad32, k = idx - i * k_pad32;
          const __bf16 h = (__bf16)v[u];
          im.hi[b3_at(im, row0 + i, k)] = h;
          im.lo[b3_at(im, row0 + i, k)] = (__bf16)(v[u] - (float)h);
        }
      }
    }
  }
}

// ---- batched prologue loads ---------------------------------------------------------------
// The prologue (weights -> split -> LDS) is a fixed cost of every launch, and most launches of the
// node side are one tile per wavefront: measured 5.3 us of a 35 us nlam_node_bwd for three images
// loaded one after the other (tools/node_timeline.py) -- one global round trip each.  Batched
// form: w16_issue() for EVERY image (and v16_issue() for the vectors) first, then the commits, so
// that all global loads of the prologue are in flight together.  NU = float4 loads per thread =
// ceil(n_pad * k_pad32 / 4 / nthreads).  Unaligned / narrow weights fall back to
// load_weight_lds_b3 at commit time.
template <int NU>
struct WLoad16 {
  f32x4 v[NU];
  bool vec;
};
template <int NU>
__device__ __forceinline__ void w16_issue(WLoad16<NU>& w, const float* __restrict__ W, int64_t ldW,
                                          int n_out, int k_in, int n_pad, int k_pad32, int tid,
                                          int nthreads) {
  w.vec = (k_in % 4 == 0) && (ldW % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0) &&
          (int64_t)NU * nthreads * 4 >= (int64_t)n_pad * k_pad32;
  if (w.vec) {
    const int cpr = k_pad32 >> 2;
    const int total = n_pad * cpr;
#pragma unroll
    for (int u = 0; u < NU; ++u) {   // unconditional, clamped; masked in w16_commit (see load_weight_lds_b3)
      const int idx = u * nthreads + tid;
      const int i = idx / cpr, c = idx - i * cpr;
      const bool ok = idx < total && i < n_out && 4 * c < k_in;
      w.v[u] = *reinterpret_cast<const f32x4*>(W + (ok ? (int64_t)i * ldW + 4 * c : 0));
    }
  }
}
template <int NU>
__device__ __forceinline__ void w16_commit(const WLoad16<NU>& w, const B3Image& im, int row0,
                                           const float* __restrict__ W, int64_t ldW, int n_out,
                                           int k_in, int n_pad, int k_pad32, int tid, int nthreads) {
  if (w.vec) {
    const int cpr = k_pad32 >> 2;
    const int total = n_pad * cpr;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = u * nthreads + tid;
      if (idx < total) {
        const int i = idx / cpr, c = idx - i * cpr;
        const f32x4 x = (i < n_out && 4 * c < k_in) ? w.v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x4 hi, lo;
        b3_split4(x, hi, lo);
        *reinterpret_cast<bf16x4*>(im.hi + b3_at(im, row0 + i, 4 * c)) = hi;
        *reinterpret_cast<bf16x4*>(im.lo + b3_at(im, row0 + i, 4 * c)) = lo;
      }
    }
  } else {
    load_weight_lds_b3(im, row0, W, ldW, n_out, k_in, n_pad, k_pad32, tid, nthreads);
  }
}
// up to 8 per-feature vectors of <= 64 entries (vector tid >> 6, entry tid & 63: 512 threads cover
// 8 vectors, 256 threads 4) in ONE load instruction; a NULL vector reads as zeros.  dst: 64-float
// slots, one per vector.
struct VLoad16 {
  float v;
};
__device__ __forceinline__ void v16_issue(VLoad16& l, const float* const (&vec)[8], const int (&len)[8],
                                          int tid) {
  const int j = tid >> 6, i = tid & 63;
  const float* p = vec[0];
  int n = len[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    p = j == k ? vec[k] : p;
    n = j == k ? len[k] : n;
  }
  l.v = (j < 8 && p != nullptr && i < n) ? p[i] : 0.f;
}
__device__ __forceinline__ void v16_commit(const VLoad16& l, float* __restrict__ dst, int nvec, int tid) {
  if ((tid >> 6) < nvec) dst[tid] = l.v;
}

// eight consecutive accumulator registers -> hi / lo bf16 fragments (the B operand of the
// K = 16 step that covers features 16 s .. 16 s + 15 of a 32-feature block: lane half h
// holds features 16 s + 4 h + {0..3} and 16 s + 8 + 4 h + {0..3} in registers 8 s .. 8 s + 7)
__device__ __forceinline__ void b3_split(const f32x16& x, int r0, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = x[r0 + i];
    hi[i] = (__bf16)v;
    lo[i] = (__bf16)(v - (float)hi[i]);
  }
}
__device__ __forceinline__ void b3_hi(const f32x16& x, int r0, bf16x8& hi) {
#pragma unroll
  for (int i = 0; i < 8; ++i) hi[i] = (__bf16)x[r0 + i];
}
__device__ __forceinline__ bf16x8 b3_join(const bf16x4& a, const bf16x4& b) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
  return r;
}
// A fragment of W . x: row `row`, the eight k of step (kb, s) for lane half h
__device__ __forceinline__ bf16x8 b3_row_frag(const __bf16* __restrict__ plane, int pitch, int row,
                                              int kb, int s, int h) {
  const __bf16* p = plane + row * pitch + 32 * kb + 16 * s + 4 * h;
  return b3_join(*reinterpret_cast<const bf16x4*>(p), *reinterpret_cast<const bf16x4*>(p + 8));
}
#define B3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// out[nb] += W[rows 32 nb .. +31][cols k0 + 32 KB features] . IN, IN in accumulator layout.
// TERMS = 3: split-bf16 (hi hi + hi lo + lo hi); TERMS = 1: plain bf16 products (hi hi only, the
// "bf16" arithmetic mode: bf16 operands, fp32 accumulate -- the lo planes are never read).
template <int NB, int KB, int TERMS = 3>
__device__ __forceinline__ void gemm_acc_b3(f32x16 (&out)[NB], const B3Image& W, int kb0,
                                            const f32x16 (&in)[KB], int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 bh, bl;
      if constexpr (TERMS == 3) b3_split(in[kb], 8 * s, bh, bl);
      else b3_hi(in[kb], 8 * s, bh);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8 ah = b3_row_frag(W.hi, W.pitch, 32 * nb + t, kb0 + kb, s, h);
        out[nb] = B3_MFMA(ah, bh, out[nb]);
        if constexpr (TERMS == 3) {
          const bf16x8 al = b3_row_frag(W.lo, W.pitch, 32 * nb + t, kb0 + kb, s, h);
          out[nb] = B3_MFMA(ah, bl, out[nb]);
          out[nb] = B3_MFMA(al, bh, out[nb]);
        }
      }
      // wide shapes: keep the scheduler from hoisting every fragment read of the unrolled
      // product to the front (16 x 2 x 8 registers at d = 128: spills)
      if constexpr (NB * KB >= 8) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---- transposed fragments (gfx950 ds_read_b64_tr_b16) --------------------------------
// Per 16-lane group the instruction reads a 4-row x 16-column block of 16-bit elements and
// hands lane i of the group column i (rows in elements 0..3).  Lane 4 q + p of the group
// supplies the address of row q, columns 4 p .. 4 p + 3.  EXEC must be all ones.
// b3_tr_frag returns, for lane (c = lane & 31, kg = lane >> 5), the eight elements
//   plane[row0 + 4 kg + {0..3}][col0 + c],  plane[row0 + 8 + 4 kg + {0..3}][col0 + c]
// i.e. the A (or B) fragment of a product that contracts over the image's ROW index in the
// order the accumulator layout uses (rows 16 s + 8 (i >> 2) + 4 kg + (i & 3), i = 0..7).
__device__ __forceinline__ bf16x8 b3_tr_frag(const __bf16* __restrict__ plane, int pitch, int row0,
                                             int col0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const __bf16* a = plane + (row0 + 4 * (g >> 1) + q) * pitch + col0 + 16 * (g & 1) + 4 * p;
  const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 8 * pitch));
  return b3_join(v0, v1);
}
// contraction over 16 CONSECUTIVE rows (row0 + 8 kg + {0..7}): the outer products, where
// both operands come from row tiles and any common row order is valid
__device__ __forceinline__ bf16x8 b3_tr_frag_rows(const __bf16* __restrict__ plane, int pitch,
                                                  int row0, int col0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const __bf16* a = plane + (row0 + 8 * (g >> 1) + q) * pitch + col0 + 16 * (g & 1) + 4 * p;
  const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a);
  const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * pitch));
  return b3_join(v0, v1);
}

// out[kb] += W[rows 32 NB][cols 32 (kb0 + kb) ..]^T . G   (gx = W^T gy), G in accumulator
// layout; the same image as gemm_acc_b3, read transposed.
template <int KBO, int NB, int TERMS = 3>
__device__ __forceinline__ void gemm_acc_wt_b3(f32x16 (&out)[KBO], const B3Image& W, int kb0,
                                               const f32x16 (&g)[NB], int lane) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 bh, bl;
      if constexpr (TERMS == 3) b3_split(g[nb], 8 * s, bh, bl);
      else b3_hi(g[nb], 8 * s, bh);
#pragma unroll
      for (int kb = 0; kb < KBO; ++kb) {
        const bf16x8 ah = b3_tr_frag(W.hi, W.pitch, 32 * nb + 16 * s, 32 * (kb0 + kb), lane);
        out[kb] = B3_MFMA(ah, bh, out[kb]);
        if constexpr (TERMS == 3) {
          const bf16x8 al = b3_tr_frag(W.lo, W.pitch, 32 * nb + 16 * s, 32 * (kb0 + kb), lane);
          out[kb] = B3_MFMA(ah, bl, out[kb]);
          out[kb] = B3_MFMA(al, bh, out[kb]);
        }
      }
      if constexpr (NB * KBO >= 8) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---- row tiles as bf16 planes (operands of the weight-gradient outer products) --------
// A 32-row tile of `width` features as two planes [32][pitch] (pitch = width32 + 4): the
// bytes of the fp32 tile [32][width32 + 4], so it can take an fp32 tile's place.
struct B3Tile {
  __bf16* hi;
  __bf16* lo;
  int pitch;
};
__device__ __forceinline__ B3Tile b3_tile(float* tile_base, int width32) {
  B3Tile t;
  t.pitch = width32 + 4;
  t.hi = reinterpret_cast<__bf16*>(tile_base);
  t.lo = t.hi + NLAM_TILE * t.pitch;
  return t;
}
// accumulator layout (lane = row t, half h) -> planes, feature block col0 / 32 onwards
template <int NB>
__device__ __forceinline__ void acc_to_tile_b3(const f32x16 (&acc)[NB], const B3Tile& T, int col0,
                                               int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[nb][4 * q + j];
      bf16x4 hi, lo;
      b3_split4(v, hi, lo);
      const int off = t * T.pitch + col0 + 32 * nb + 8 * q + 4 * h;
      *reinterpret_cast<bf16x4*>(T.hi + off) = hi;
      *reinterpret_cast<bf16x4*>(T.lo + off) = lo;
    }
  }
}
// two-phase staged float4 rows (load_rows_v) -> planes; rows >= nrows are zeroed
template <int NV>
__device__ __forceinline__ void put_rows_v_b3(const B3Tile& T, int col0, int width, int nrows,
                                              int lane, const f32x4 (&v)[NV]) {
  const int lpr = width >> 2;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr, c4 = lane - sub * lpr;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int t = sub + k * rpi;
    if (sub < rpi && t < NLAM_TILE) {
      f32x4 x = v[k];
      if (t >= nrows) x = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x4 hi, lo;
      b3_split4(x, hi, lo);
      const int off = t * T.pitch + col0 + 4 * c4;
      *reinterpret_cast<bf16x4*>(T.hi + off) = hi;
      *reinterpret_cast<bf16x4*>(T.lo + off) = lo;
    }
  }
}

// out[nb] += W[rows 32 nb ..][cols 32 (kb0 + kb) ..] . X^T with X a bf16-plane row tile: the
// B fragments are plain 8-byte row reads of the planes (no register transpose, no
// conversion: the rows were split when they were staged).
template <int NB, int KB, int TERMS = 3>
__device__ __forceinline__ void gemm_tile_b3(f32x16 (&out)[NB], const B3Image& W, int kb0,
                                             const B3Tile& X, int xcol0, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int xo = t * X.pitch + xcol0 + 32 * kb + 16 * s + 4 * h;
      const bf16x8 bh = b3_join(*reinterpret_cast<const bf16x4*>(X.hi + xo),
                                *reinterpret_cast<const bf16x4*>(X.hi + xo + 8));
      bf16x8 bl;
      if constexpr (TERMS == 3)
        bl = b3_join(*reinterpret_cast<const bf16x4*>(X.lo + xo),
                     *reinterpret_cast<const bf16x4*>(X.lo + xo + 8));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8 ah = b3_row_frag(W.hi, W.pitch, 32 * nb + t, kb0 + kb, s, h);
        out[nb] = B3_MFMA(ah, bh, out[nb]);
        if constexpr (TERMS == 3) {
          const bf16x8 al = b3_row_frag(W.lo, W.pitch, 32 * nb + t, kb0 + kb, s, h);
          out[nb] = B3_MFMA(ah, bl, out[nb]);
          out[nb] = B3_MFMA(al, bh, out[nb]);
        }
      }
    }
  }
}

// acc[j] (lanes = features 64 j + lane) += sum over the 32 tile rows of X[t][xcol0 + 64 j + lane]
// -- the per-feature (bias / gamma / beta) gradient of one tile -- as a ones-vector product on
// the matrix cores: A = all-ones fragment (exact in bf16, no LDS read), B = transposed plane
// fragments, 2 MFMAs per 32-column block and K step.  Every row of the 32x32 result block
// holds the column sums; rows >= nrows of the planes must be zero.  Replaces a 32-step LDS
// loop (2^-18 relative error per element from the hi/lo split).
template <int NV, int TERMS = 3>
__device__ __forceinline__ void tile_colsum_b3(float (&acc)[NV], const B3Tile& X, int xcol0,
                                               int lane) {
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float v[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {      // one 32-column block at a time: 16 live registers
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 bh = b3_tr_frag_rows(X.hi, X.pitch, 16 * u, xcol0 + 64 * j + 32 * blk, lane);
        c = B3_MFMA(ones, bh, c);
        if constexpr (TERMS == 3) {
          const bf16x8 bl = b3_tr_frag_rows(X.lo, X.pitch, 16 * u, xcol0 + 64 * j + 32 * blk, lane);
          c = B3_MFMA(ones, bl, c);
        }
      }
      v[blk] = c[0];
    }
    acc[j] += (lane < 32) ? v[0] : v[1];
  }
}

// dW[ib][jb] += sum_t G[t][gcol0 + 32 ib + .] (x) X[t][xcol0 + 32 jb + .] over the 32 tile
// rows; G and X are bf16-plane tiles.  Result block layout as outer_accum (fused_common.h).
template <int NI, int NJ, int TERMS = 3>
__device__ __forceinline__ void outer_accum_b3(f32x16 (&dW)[NI][NJ], const B3Tile& G, int gcol0,
                                               const B3Tile& X, int xcol0, int lane) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    bf16x8 ah[NI], al[NI];
#pragma unroll
    for (int ib = 0; ib < NI; ++ib) {
      ah[ib] = b3_tr_frag_rows(G.hi, G.pitch, 16 * u, gcol0 + 32 * ib, lane);
      if constexpr (TERMS == 3)
        al[ib] = b3_tr_frag_rows(G.lo, G.pitch, 16 * u, gcol0 + 32 * ib, lane);
    }
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      const bf16x8 bh = b3_tr_frag_rows(X.hi, X.pitch, 16 * u, xcol0 + 32 * jb, lane);
      bf16x8 bl;
      if constexpr (TERMS == 3)
        bl = b3_tr_frag_rows(X.lo, X.pitch, 16 * u, xcol0 + 32 * jb, lane);
#pragma unroll
      for (int ib = 0; ib < NI; ++ib) {
        dW[ib][jb] = B3_MFMA(ah[ib], bh, dW[ib][jb]);
        if constexpr (TERMS == 3) {
          dW[ib][jb] = B3_MFMA(ah[ib], bl, dW[ib][jb]);
          dW[ib][jb] = B3_MFMA(al[ib], bh, dW[ib][jb]);
        }
      }
    }
  }
}

// out[kb] += W[rows 32 NB][cols 32 (kb0 + kb) ..]^T . G with G a bf16-plane row tile
// (gx = W^T gy straight from the staged planes: no register copy of gy at all).
template <int KBO, int NB, int TERMS = 3>
__device__ __forceinline__ void gemm_tile_wt_b3(f32x16 (&out)[KBO], const B3Image& W, int kb0,
                                                const B3Tile& G, int gcol0, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int go = t * G.pitch + gcol0 + 32 * nb + 16 * s + 4 * h;
      const bf16x8 bh = b3_join(*reinterpret_cast<const bf16x4*>(G.hi + go),
                                *reinterpret_cast<const bf16x4*>(G.hi + go + 8));
      bf16x8 bl;
      if constexpr (TERMS == 3)
        bl = b3_join(*reinterpret_cast<const bf16x4*>(G.lo + go),
                     *reinterpret_cast<const bf16x4*>(G.lo + go + 8));
#pragma unroll
      for (int kb = 0; kb < KBO; ++kb) {
        const bf16x8 ah = b3_tr_frag(W.hi, W.pitch, 32 * nb + 16 * s, 32 * (kb0 + kb), lane);
        out[kb] = B3_MFMA(ah, bh, out[kb]);
        if constexpr (TERMS == 3) {
          const bf16x8 al = b3_tr_frag(W.lo, W.pitch, 32 * nb + 16 * s, 32 * (kb0 + kb), lane);
          out[kb] = B3_MFMA(ah, bl, out[kb]);
          out[kb] = B3_MFMA(al, bh, out[kb]);
        }
      }
    }
  }
}
