// 16-row ("acc16") building blocks of the hidden-64 kernels that run TWO wavefronts per SIMD
// (512-thread workgroups, <= 256 registers per lane) -- csrc/fused16_*.hip.
//
// Why: the 32-row kernels of fused_edge.hip / fused_mlp.hip hold ~500 registers and 139 KB of
// LDS per 4-wave workgroup, i.e. one wave per SIMD.  One wave alone issues a vector instruction
// every 4 cycles (two co-resident waves: every 2), and nothing overlaps its VALU / LDS / MFMA /
// global-memory phases: those kernels sat at 0.04-0.27 of their HBM roof with the matrix cores
// 12 % busy.
//
// Layout: a wavefront owns a tile of 16 rows (edges or nodes).  Lane l = (t, g) with t = l & 15
// the row and g = l >> 4.  A row tensor of D features is NF = D / 16 registers-quads:
//     a[fb][reg]  <->  feature 16 fb + 4 g + reg            (fb < NF, reg < 4)
// which is the C/D map of v_mfma_f32_16x16x32_bf16 for the transposed product
// D^T[f][t] = sum_k W[f][k] X[t][k].  Consequences:
//   * a row tensor costs D / 4 bytes of registers per lane (16 VGPRs at D = 64, half of the
//     32-row form), and a result is directly the B operand of the next product: K step s
//     takes blocks 2 s and 2 s + 1, i.e. k slot (g, j) <-> feature 32 s + 16 (j >> 2) + 4 g +
//     (j & 3); the weight fragments are read in the same order (two 8-byte LDS reads);
//   * rows move between global memory and registers DIRECTLY in this layout (lane (t, g)
//     reads / writes the 16-byte chunks 16 fb + 4 g of row t: 64 contiguous bytes per row and
//     instruction): no LDS staging, no tile transposes, no per-row index tables;
//   * LayerNorm statistics are a per-lane sum + two cross-lane adds (xor 16, xor 32);
//   * only what contracts over the ROW index goes through LDS: bf16 hi/lo planes [16][pitch]
//     for the weight-gradient outer products and the per-feature sums (gfx950 transposed reads
//     + 32x32x16 MFMAs, exactly 16 rows = one K step), and one fp32 tile for the
//     receiver-aligned segmented sums.
// tools/sim16.py is the CPU model these index formulas were checked against.
#pragma once
#include "fused_bf16x3.h"

#define NLAM_T16 16
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// ---- weight images -------------------------------------------------------------------
// Two bf16 planes (hi, lo) in natural [n][k] order like fused_bf16x3.h.  Two layouts:
//   plain    (forward-only kernels): row pitch K + 8 elements -- with 16 rows x 2 k-groups per
//            32-lane half the 8-byte ROW reads of W . x are conflict-free at K/2 + 4 dwords per row;
//            the transposed reads of W^T . g are 2-way conflicted there (rows 0 and 7 of a half
//            wrap onto the same banks);
//   swizzled (kernels that also read the image transposed: every backward): row pitch K + 16 and
//            rows with bit 3 set start 8 elements later -- row reads AND transposed reads are both
//            conflict-free (tools/sim16.py bank_report; SQ_LDS_BANK_CONFLICT was 0.36-0.45 of the
//            LDS cycles of the 16-row backward kernels with the plain layout: VERDICT r4 2c).
// The helpers take either: B3Image carries pitch and swizzle.
__host__ __device__ constexpr int w16_pitch(int k32, bool swz = false) { return k32 + (swz ? 16 : 8); }
__host__ __device__ constexpr size_t w16_image_bytes(int n_pad, int k32, bool swz = false) {
  return (size_t)2 * (n_pad * w16_pitch(k32, swz) + (swz ? 8 : 0)) * sizeof(__bf16);
}
__device__ __forceinline__ B3Image w16_image(void* base, int n_pad, int k32, bool swz = false) {
  B3Image im;
  im.pitch = w16_pitch(k32, swz);
  im.swz = swz ? 8 : 0;
  im.hi = reinterpret_cast<__bf16*>(base);
  im.lo = im.hi + n_pad * im.pitch + im.swz;
  return im;
}

// ---- 16-row bf16 planes ----------------------------------------------------------------
__host__ __device__ constexpr int p16_pitch(int width) { return width + 4; }
__host__ __device__ constexpr size_t p16_bytes(int width) {
  return (size_t)2 * NLAM_T16 * p16_pitch(width) * sizeof(__bf16);
}
__device__ __forceinline__ B3Tile p16_tile(void* base, int width) {
  B3Tile t;
  t.pitch = p16_pitch(width);
  t.hi = reinterpret_cast<__bf16*>(base);
  t.lo = t.hi + NLAM_T16 * t.pitch;
  return t;
}

// (batched prologue loads -- w16_issue / w16_commit / v16_issue / v16_commit: fused_bf16x3.h)

// two register quads (feature blocks 2 s, 2 s + 1) -> the hi / lo B fragments of K step s
__device__ __forceinline__ void split16(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)a[j];
    lo[j] = (__bf16)(a[j] - (float)hi[j]);
    hi[4 + j] = (__bf16)b[j];
    lo[4 + j] = (__bf16)(b[j] - (float)hi[4 + j]);
  }
}
__device__ __forceinline__ void hi16(const f32x4& a, const f32x4& b, bf16x8& hi) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)a[j];
    hi[4 + j] = (__bf16)b[j];
  }
}

// An index the compiler must treat as new at this point: row addresses formed from it late in a
// tile are recomputed there (a few VALU) instead of being kept -- or spilled -- as 64-bit
// pointers from the top of the tile.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ int64_t opaque(int64_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <int N>
__device__ __forceinline__ void zero16(f32x4 (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// out[fb] += W[row0 + 16 fb + .][col0 + 32 KS features] . IN   (fb < NO), IN = 2 KS quads.
// TERMS = 3: split-bf16 (hi hi + hi lo + lo hi); 1: plain bf16 products.
template <int NO, int KS, int TERMS = 3>
__device__ __forceinline__ void gemm_acc16(f32x4 (&out)[NO], const B3Image& W, int row0, int col0,
                                           const f32x4* __restrict__ in, int lane) {
  const int i = lane & 15, g = lane >> 4;
  bf16x8 bh[KS], bl[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if constexpr (TERMS == 3) split16(in[2 * s], in[2 * s + 1], bh[s], bl[s]);
    else hi16(in[2 * s], in[2 * s + 1], bh[s]);
  }
  const int base = (row0 + i) * W.pitch + col0 + 4 * g + W.swz * ((i >> 3) & 1);   // (row0 % 16 == 0)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int fb = 0; fb < NO; ++fb) {
      const int o = base + 16 * fb * W.pitch + 32 * s;
      const bf16x8 ah = b3_join(*reinterpret_cast<const bf16x4*>(W.hi + o),
                                *reinterpret_cast<const bf16x4*>(W.hi + o + 16));
      out[fb] = MFMA16(ah, bh[s], out[fb]);
      if constexpr (TERMS == 3) {
        const bf16x8 al = b3_join(*reinterpret_cast<const bf16x4*>(W.lo + o),
                                  *reinterpret_cast<const bf16x4*>(W.lo + o + 16));
        out[fb] = MFMA16(ah, bl[s], out[fb]);
        out[fb] = MFMA16(al, bh[s], out[fb]);
      }
    }
    // deep products: keep the scheduler from hoisting every fragment read of the unrolled
    // product to the front (4 K steps x 4 blocks x 8 registers: spills)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// out[kb] += W[row0 + 32 NS features][col0 + 16 kb + .]^T . G   (kb < KO): gx = W^T gy from
// the SAME image through the transposed LDS read (per 16-lane group: 4 rows x 16 columns).
template <int KO, int NS, int TERMS = 3>
__device__ __forceinline__ void gemm_acc16_wt(f32x4 (&out)[KO], const B3Image& W, int row0,
                                              int col0, const f32x4* __restrict__ gin, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  bf16x8 bh[NS], bl[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if constexpr (TERMS == 3) split16(gin[2 * s], gin[2 * s + 1], bh[s], bl[s]);
    else hi16(gin[2 * s], gin[2 * s + 1], bh[s]);
  }
  const int base = (row0 + 4 * g + q) * W.pitch + col0 + 4 * p + W.swz * (g >> 1);   // (row0 % 16 == 0)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int kb = 0; kb < KO; ++kb) {
      const int o = base + 32 * s * W.pitch + 16 * kb;
      const bf16x8 ah =
          b3_join(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(W.hi + o)),
                  __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(W.hi + o + 16 * W.pitch)));
      out[kb] = MFMA16(ah, bh[s], out[kb]);
      if constexpr (TERMS == 3) {
        const bf16x8 al =
            b3_join(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(W.lo + o)),
                    __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(W.lo + o + 16 * W.pitch)));
        out[kb] = MFMA16(ah, bl[s], out[kb]);
        out[kb] = MFMA16(al, bh[s], out[kb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // (bounds the hoisting of fragment reads)
  }
}

// ---- split once, use twice ------------------------------------------------------------
// The hi / lo B fragments of a 32 KS-feature row tensor.  The same bf16 values are what the
// row-contracting products read from the LDS planes, so a tensor is split ONCE and the
// fragments serve the GEMM (registers) and the planes (frag16_to_planes): the split is ~3 VALU
// per element and these kernels are VALU-issue bound.
template <int KS>
struct Frag16 {
  bf16x8 h[KS], l[KS];
};
template <int KS, int TERMS = 3>
__device__ __forceinline__ void make_frag16(Frag16<KS>& f, const f32x4* __restrict__ in) {
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if constexpr (TERMS == 3) split16(in[2 * s], in[2 * s + 1], f.h[s], f.l[s]);
    else hi16(in[2 * s], in[2 * s + 1], f.h[s]);
  }
}
template <int NO, int KS, int TERMS = 3>
__device__ __forceinline__ void gemm_frag16(f32x4 (&out)[NO], const B3Image& W, int row0, int col0,
                                            const Frag16<KS>& f, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int base = (row0 + i) * W.pitch + col0 + 4 * g + W.swz * ((i >> 3) & 1);   // (row0 % 16 == 0)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
#pragma unroll
    for (int fb = 0; fb < NO; ++fb) {
      const int o = base + 16 * fb * W.pitch + 32 * s;
      const bf16x8 ah = b3_join(*reinterpret_cast<const bf16x4*>(W.hi + o),
                                *reinterpret_cast<const bf16x4*>(W.hi + o + 16));
      out[fb] = MFMA16(ah, f.h[s], out[fb]);
      if constexpr (TERMS == 3) {
        const bf16x8 al = b3_join(*reinterpret_cast<const bf16x4*>(W.lo + o),
                                  *reinterpret_cast<const bf16x4*>(W.lo + o + 16));
        out[fb] = MFMA16(ah, f.l[s], out[fb]);
        out[fb] = MFMA16(al, f.h[s], out[fb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // (bounds the hoisting of fragment reads)
  }
}
template <int KO, int NS, int TERMS = 3>
__device__ __forceinline__ void gemm_frag16_wt(f32x4 (&out)[KO], const B3Image& W, int row0,
                                               int col0, const Frag16<NS>& f, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int base = (row0 + 4 * g + q) * W.pitch + col0 + 4 * p + W.swz * (g >> 1);   // (row0 % 16 == 0)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int kb = 0; kb < KO; ++kb) {
      const int o = base + 32 * s * W.pitch + 16 * kb;
      const bf16x8 ah =
          b3_join(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(W.hi + o)),
                  __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(W.hi + o + 16 * W.pitch)));
      out[kb] = MFMA16(ah, f.h[s], out[kb]);
      if constexpr (TERMS == 3) {
        const bf16x8 al =
            b3_join(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(W.lo + o)),
                    __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(W.lo + o + 16 * W.pitch)));
        out[kb] = MFMA16(ah, f.l[s], out[kb]);
        out[kb] = MFMA16(al, f.h[s], out[kb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // (bounds the hoisting of fragment reads)
  }
}
// fragments -> planes: elements 0..3 of step s are features 32 s + 4 g + {0..3}, elements 4..7
// features 32 s + 16 + 4 g + {0..3} of row t
template <int KS, int TERMS = 3>
__device__ __forceinline__ void frag16_to_planes(const Frag16<KS>& f, const B3Tile& T, int col0,
                                                 int lane) {
  const int t = lane & 15, g = lane >> 4;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int off = t * T.pitch + col0 + 32 * s + 4 * g;
    *reinterpret_cast<bf16x4*>(T.hi + off) = __builtin_shufflevector(f.h[s], f.h[s], 0, 1, 2, 3);
    *reinterpret_cast<bf16x4*>(T.hi + off + 16) = __builtin_shufflevector(f.h[s], f.h[s], 4, 5, 6, 7);
    if constexpr (TERMS == 3) {
      *reinterpret_cast<bf16x4*>(T.lo + off) = __builtin_shufflevector(f.l[s], f.l[s], 0, 1, 2, 3);
      *reinterpret_cast<bf16x4*>(T.lo + off + 16) = __builtin_shufflevector(f.l[s], f.l[s], 4, 5, 6, 7);
    }
  }
}

// silu and its derivative from ONE sigmoid (the two transcendentals are the expensive part)
__device__ __forceinline__ void silu_both(float x, float& s, float& ds) {
  const float sg = nlam_sigmoid(x);
  s = x * sg;
  ds = sg * (1.0f + x * (1.0f - sg));
}

// per-feature vector (bias / gamma / beta, fp32 in LDS) in acc16 layout
template <int NF>
__device__ __forceinline__ void vec_to_acc16(f32x4 (&v)[NF], const float* __restrict__ p, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb) v[fb] = *reinterpret_cast<const f32x4*>(p + 16 * fb + 4 * g);
}

// row sums over the 4 lanes (t, 0..3) that share a row
__device__ __forceinline__ float row_sum16(float s) {
  s = lane_xor16_sum(s);
  s = lane_xor32_sum(s);
  return s;
}
template <int NF>
__device__ __forceinline__ void ln16_stats(const f32x4 (&z)[NF], float& mean, float& rstd) {
  constexpr float inv_d = 1.0f / (16.0f * NF);
  float s = 0.f;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += z[fb][r];
  mean = row_sum16(s) * inv_d;
  float v = 0.f;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = z[fb][r] - mean;
      v += d * d;
    }
  rstd = rsqrtf(row_sum16(v) * inv_d + 1e-5f);
}
// y = (z - mean) rstd gamma + beta in place (gamma / beta: fp32 vectors in LDS)
template <int NF>
__device__ __forceinline__ void ln16_apply(f32x4 (&z)[NF], const float* __restrict__ gamma,
                                           const float* __restrict__ beta, int lane) {
  float mean, rstd;
  ln16_stats<NF>(z, mean, rstd);
  const int g = lane >> 4;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * fb + 4 * g);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * fb + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) z[fb][r] = (z[fb][r] - mean) * rstd * gm[r] + bt[r];
  }
}
// LayerNorm backward in place: z (pre-norm) -> xhat, g (dL/dy) -> dL/dz.  The dgamma summand
// g * xhat goes straight to the bf16 planes P (no register copy); the dbeta summand is g
// itself, taken by the caller before the call.
template <int NF, int TERMS = 3, bool WRITE_P = true>
__device__ __forceinline__ void ln16_bwd(f32x4 (&z)[NF], f32x4 (&g)[NF], const B3Tile& P,
                                         const float* __restrict__ gamma, int lane) {
  constexpr float inv_d = 1.0f / (16.0f * NF);
  float mean, rstd;
  ln16_stats<NF>(z, mean, rstd);
  const int t = lane & 15, gq = lane >> 4;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * fb + 4 * gq);
    f32x4 prod;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xh = (z[fb][r] - mean) * rstd;
      z[fb][r] = xh;
      prod[r] = g[fb][r] * xh;
      const float gv = g[fb][r] * gm[r];
      g[fb][r] = gv;
      s1 += gv;
      s2 += gv * xh;
    }
    if constexpr (WRITE_P) {
      bf16x4 hi, lo;
      b3_split4(prod, hi, lo);
      const int off = t * P.pitch + 16 * fb + 4 * gq;
      *reinterpret_cast<bf16x4*>(P.hi + off) = hi;
      if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(P.lo + off) = lo;
    }
  }
  const float m1 = row_sum16(s1) * inv_d, m2 = row_sum16(s2) * inv_d;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
#pragma unroll
    for (int r = 0; r < 4; ++r) g[fb][r] = rstd * (g[fb][r] - m1 - z[fb][r] * m2);
}

// ---- global rows <-> registers ---------------------------------------------------------
// `row` points at this lane's row (already offset by the batch item); columns [c0, c0 + 16 NF)
template <int NF>
__device__ __forceinline__ void load_row16(f32x4* __restrict__ a, const float* __restrict__ row,
                                           int lane) {
  const float* p = row + 4 * (lane >> 4);
#pragma unroll
  for (int fb = 0; fb < NF; ++fb) a[fb] = *reinterpret_cast<const f32x4*>(p + 16 * fb);
}
template <int NF>
__device__ __forceinline__ void store_row16(float* __restrict__ row, const f32x4* __restrict__ a,
                                            int lane) {
  float* p = row + 4 * (lane >> 4);
#pragma unroll
  for (int fb = 0; fb < NF; ++fb) *reinterpret_cast<f32x4*>(p + 16 * fb) = a[fb];
}
// narrow or unaligned rows (the 2-3 wide static features, the 17-wide output map): element
// loads / stores, columns >= width read as zero
template <int NF>
__device__ __forceinline__ void load_row16_s(f32x4* __restrict__ a, const float* __restrict__ row,
                                             int width, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * fb + 4 * g + r;
      a[fb][r] = f < width ? row[f] : 0.f;
    }
}
template <int NF>
__device__ __forceinline__ void store_row16_s(float* __restrict__ row, const f32x4* __restrict__ a,
                                              int width, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * fb + 4 * g + r;
      if (f < width) row[f] = a[fb][r];
    }
}
template <int NF>
__device__ __forceinline__ void mask16(f32x4* __restrict__ a, bool keep) {
  if (!keep) {
#pragma unroll
    for (int fb = 0; fb < NF; ++fb) a[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// ---- sender-side gather ---------------------------------------------------------------------
// Sum of the rows gh[eid[p]] over this lane's node's out-edges, in sender-list order (fixed =>
// deterministic).  A tile is one latency chain per wavefront (26 k rows = 1.6 tiles per SIMD), so
// the dependent loads are batched: the four lanes of a row fetch 32 edge ids in ONE round trip
// (lane g takes positions g, g + 4, ...), then the rows of four edges are in flight together.
// Lanes past their own degree read edge 0's row and add nothing.
__device__ __forceinline__ void gather_sender_sum16(f32x4* __restrict__ acc, const float* __restrict__ gh,
                                                    const int32_t* __restrict__ colptr,
                                                    const int32_t* __restrict__ eid, int64_t node,
                                                    int n_send, bool valid, int lane) {
  const int t = lane & 15, g = lane >> 4;
  int p0 = 0, deg = 0;
  if (valid && node < n_send) {
    p0 = colptr[node];
    deg = colptr[node + 1] - p0;
  }
  int dmax = deg;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) dmax = max(dmax, __shfl_xor(dmax, o, 64));
  dmax = __builtin_amdgcn_readfirstlane(dmax);   // the four lanes of a row agree; rows: xor 1..8
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) acc[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < dmax; c0 += 32) {
    int ev[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = c0 + 4 * u + g;
      ev[u] = eid[j < deg ? p0 + j : 0];
    }
#pragma unroll
    for (int j4 = 0; j4 < 32; j4 += 4) {
      if (c0 + j4 < dmax) {   // wave-uniform
        f32x4 v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = __shfl(ev[j4 >> 2], t + 16 * k, 64);   // position c0 + j4 + k of row t
          load_row16<4>(v[k], gh + (int64_t)e * 64, lane);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c0 + j4 + k < deg) {
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) acc[fb] += v[k][fb];
          }
      }
    }
  }
}

// ---- whole-row ("row-shaped") global access -------------------------------------------------
// Lane (r4 = l >> 4, c = l & 15) moves the 16-byte chunk c of row 4 k + r4 (k = 0..3): one
// wave-instruction = 4 whole 256-byte rows.  The fragment-shaped form above (16 rows x 64 bytes per
// instruction) makes 2.8 x the L1 accesses per instruction (TCP_TOTAL_CACHE_ACCESSES, profiles/
// r03_pmc_*): the gathered / scattered edge and sender rows therefore move row-shaped and change
// shape through this wave's fp32 LDS tile (4 x ds_write_b128 + 4 x ds_read_b128 per tensor).
// ridx[k] = row index of slot 4 k + r4: from the slot-arranged index (lanes 0..15) by 4 shuffles.
__device__ __forceinline__ void rs_index(int (&ridx)[4], int idx_t, int lane) {
#pragma unroll
  for (int k = 0; k < 4; ++k) ridx[k] = __shfl(idx_t, 4 * k + (lane >> 4), 64);
}
__device__ __forceinline__ void rs_load(f32x4 (&v)[4], const float* __restrict__ base, int64_t ld,
                                        const int (&ridx)[4], int lane) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] = *reinterpret_cast<const f32x4*>(base + (int64_t)ridx[k] * ld + 4 * (lane & 15));
}
__device__ __forceinline__ void rs_store(float* __restrict__ base, int64_t ld, const int (&ridx)[4],
                                         const f32x4 (&v)[4], int nrows, int lane) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (4 * k + (lane >> 4) < nrows)
      *reinterpret_cast<f32x4*>(base + (int64_t)ridx[k] * ld + 4 * (lane & 15)) = v[k];
}
__device__ __forceinline__ void rs_to_tile(const f32x4 (&v)[4], float* __restrict__ tile, int ld,
                                           int lane) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    *reinterpret_cast<f32x4*>(tile + (4 * k + (lane >> 4)) * ld + 4 * (lane & 15)) = v[k];
}
__device__ __forceinline__ void tile_to_rs(f32x4 (&v)[4], const float* __restrict__ tile, int ld,
                                           int lane) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    v[k] = *reinterpret_cast<const f32x4*>(tile + (4 * k + (lane >> 4)) * ld + 4 * (lane & 15));
}
template <int NF>
__device__ __forceinline__ void tile_to_acc16(f32x4* __restrict__ a, const float* __restrict__ tile,
                                              int ld, int lane) {
  const int t = lane & 15, g = lane >> 4;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
    a[fb] = *reinterpret_cast<const f32x4*>(tile + t * ld + 16 * fb + 4 * g);
}

// row-shaped registers -> accumulator layout through this wave's fp32 tile (DS operations of one
// wave execute in order, so back-to-back conversions through the same tile need no waits)
__device__ __forceinline__ void rs_to_acc16(f32x4 (&a)[4], const f32x4 (&v)[4],
                                            float* __restrict__ tile, int ld, int lane) {
  rs_to_tile(v, tile, ld, lane);
  wave_sync();
  tile_to_acc16<4>(a, tile, ld, lane);
  wave_sync();
}
// tile (accumulator-layout rows already written) -> whole-row stores at base + idx[slot] * gld
__device__ __forceinline__ void tile_store_rs(const float* __restrict__ tile, int ld,
                                              float* __restrict__ base, int64_t gld, int idx_t,
                                              int nrows, int lane) {
  f32x4 v[4];
  int r[4];
  tile_to_rs(v, tile, ld, lane);
  rs_index(r, idx_t, lane);
  rs_store(base, gld, r, v, nrows, lane);
}

// ---- registers -> LDS ------------------------------------------------------------------
template <int NF, int TERMS = 3>
__device__ __forceinline__ void acc16_to_planes(const f32x4* __restrict__ a, const B3Tile& T,
                                                int col0, int lane) {
  const int t = lane & 15, g = lane >> 4;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb) {
    bf16x4 hi, lo;
    b3_split4(a[fb], hi, lo);
    const int off = t * T.pitch + col0 + 16 * fb + 4 * g;
    *reinterpret_cast<bf16x4*>(T.hi + off) = hi;
    if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(T.lo + off) = lo;
  }
}
// fp32 tile [16][ld] (ld = width + 4: conflict-free 16-byte writes, tools/sim16.py)
template <int NF>
__device__ __forceinline__ void acc16_to_tile(const f32x4* __restrict__ a, float* __restrict__ tile,
                                              int ld, int lane) {
  const int t = lane & 15, g = lane >> 4;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
    *reinterpret_cast<f32x4*>(tile + t * ld + 16 * fb + 4 * g) = a[fb];
}

// ---- products that contract over the 16 tile rows ---------------------------------------
// dW[ib][jb] += sum_t G[t][gcol0 + 32 ib + .] (x) X[t][xcol0 + 32 jb + .]; result block layout
// as outer_accum (fused_common.h): row i = 8 (r >> 2) + 4 h + (r & 3), column j = lane & 31.
template <int NI, int NJ, int TERMS = 3>
__device__ __forceinline__ void outer_accum16(f32x16 (&dW)[NI][NJ], const B3Tile& G, int gcol0,
                                              const B3Tile& X, int xcol0, int lane) {
  bf16x8 ah[NI], al[NI];
#pragma unroll
  for (int ib = 0; ib < NI; ++ib) {
    ah[ib] = b3_tr_frag_rows(G.hi, G.pitch, 0, gcol0 + 32 * ib, lane);
    if constexpr (TERMS == 3) al[ib] = b3_tr_frag_rows(G.lo, G.pitch, 0, gcol0 + 32 * ib, lane);
  }
#pragma unroll
  for (int jb = 0; jb < NJ; ++jb) {
    const bf16x8 bh = b3_tr_frag_rows(X.hi, X.pitch, 0, xcol0 + 32 * jb, lane);
    bf16x8 bl;
    if constexpr (TERMS == 3) bl = b3_tr_frag_rows(X.lo, X.pitch, 0, xcol0 + 32 * jb, lane);
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
// The same products plus the column sums of G (db = colsum(G): the bias gradient that goes with
// dW = G^T X) from the A fragments the outer product loads anyway: D = ones (32 x 16) . G (16 x 32)
// holds the 32 column sums of a block in every row, so lane l & 31 reads them in register 0 --
// lanes = features, the layout of the per-lane accumulators -- with no plane store, no transposed
// read and no sync of its own (colsum16 cost 16 transposed reads + 8 MFMAs per call; profiles/
// r05_experiments.txt item 8: the row-contracting products are 30-43 % of an MLP backward).
template <int NI, int NJ, int TERMS = 3>
__device__ __forceinline__ void outer_accum16_cs(f32x16 (&dW)[NI][NJ], float (&csum)[NI / 2],
                                                 const B3Tile& G, int gcol0, const B3Tile& X, int xcol0,
                                                 int lane) {
  static_assert(NI == 2 || NI == 4, "64 or 128 gradient columns");
  bf16x8 ah[NI], al[NI];
#pragma unroll
  for (int ib = 0; ib < NI; ++ib) {
    ah[ib] = b3_tr_frag_rows(G.hi, G.pitch, 0, gcol0 + 32 * ib, lane);
    if constexpr (TERMS == 3) al[ib] = b3_tr_frag_rows(G.lo, G.pitch, 0, gcol0 + 32 * ib, lane);
  }
  {
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
#pragma unroll
    for (int j = 0; j < NI / 2; ++j) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = B3_MFMA(ones, ah[2 * j + h], c);
        if constexpr (TERMS == 3) c = B3_MFMA(ones, al[2 * j + h], c);
        v[h] = c[0];
      }
      csum[j] += (lane < 32) ? v[0] : v[1];
    }
  }
#pragma unroll
  for (int jb = 0; jb < NJ; ++jb) {
    const bf16x8 bh = b3_tr_frag_rows(X.hi, X.pitch, 0, xcol0 + 32 * jb, lane);
    bf16x8 bl;
    if constexpr (TERMS == 3) bl = b3_tr_frag_rows(X.lo, X.pitch, 0, xcol0 + 32 * jb, lane);
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
// column sums of a 64-wide plane pair on the 32 x 32 x 16 product (K = the 16 tile rows exactly:
// 4 transposed reads + 2 MFMAs per 32 columns and plane; colsum16 below pads K to 32)
template <int TERMS = 3>
__device__ __forceinline__ void colsum16_64(float& acc, const B3Tile& X, int xcol0, int lane) {
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
  float v[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = B3_MFMA(ones, b3_tr_frag_rows(X.hi, X.pitch, 0, xcol0 + 32 * blk, lane), c);
    if constexpr (TERMS == 3) c = B3_MFMA(ones, b3_tr_frag_rows(X.lo, X.pitch, 0, xcol0 + 32 * blk, lane), c);
    v[blk] = c[0];
  }
  acc += (lane < 32) ? v[0] : v[1];
}
// acc[j] (lanes = features xcol0 + 64 j + lane) += column sums of the 16 plane rows (rows of
// padded slots must hold zeros), as ones-vector products on the 16x16x32 MFMA: B[k = row][n =
// feature] by transposed reads (k slots 16..31 are zero fragments), A = ones, so every result
// row holds the 16 column sums of the block; lane l keeps the block l >> 4 (its feature is
// 16 (l >> 4) + (l & 15) = l).  4-register temporaries.
__device__ __forceinline__ float colsum16_blockset(const B3Tile& X, int col0, int nblk, int lane,
                                                   bool lo_too) {
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
  const int kg = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3, gsel = lane >> 4;
  const bool live = lane < 32;
  const int base = (8 * kg + q) * X.pitch + col0 + 4 * p;
  float res = 0.f;
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    if (blk < nblk) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      const int o = base + 16 * blk;
      bf16x8 bh = b3_join(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(X.hi + o)),
                          __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(X.hi + o + 4 * X.pitch)));
      if (!live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) bh[i] = (__bf16)0.0f;
      }
      c = MFMA16(ones, bh, c);
      if (lo_too) {
        bf16x8 bl = b3_join(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(X.lo + o)),
                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(X.lo + o + 4 * X.pitch)));
        if (!live) {
#pragma unroll
          for (int i = 0; i < 8; ++i) bl[i] = (__bf16)0.0f;
        }
        c = MFMA16(ones, bl, c);
      }
      res = (gsel == blk) ? c[0] : res;
    }
  }
  return res;
}
template <int NV, int TERMS = 3>
__device__ __forceinline__ void colsum16(float (&acc)[NV], const B3Tile& X, int xcol0, int lane) {
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] += colsum16_blockset(X, xcol0 + 64 * j, 4, lane, TERMS == 3);
}
// one 32-column block only (narrow outputs: the 17-wide output map); lanes >= 32 add zero
template <int TERMS = 3>
__device__ __forceinline__ void colsum16_32(float& acc, const B3Tile& X, int xcol0, int lane) {
  acc += colsum16_blockset(X, xcol0, 2, lane, TERMS == 3);
}

// ---- workgroup-level fold of the per-wave weight-gradient blocks (8 waves) -----------------
// Every wave writes its blocks to its own LDS image (all NW in parallel), then all threads add
// the images in wave order (fixed => deterministic) straight into the slab.  img: NW * 32 NI *
// ldimg floats of dead LDS.  Call from all threads after a __syncthreads().
template <int NI, int NJ, int LDJ, int NW>
__device__ __forceinline__ void fold_blocks_to_slab16(const f32x16* __restrict__ dW /* [NI][LDJ] */,
                                                      float* __restrict__ img, int ldimg,
                                                      float* __restrict__ slab, int tid, int wave,
                                                      int lane) {
  const int h = lane >> 5, j = lane & 31;
  const int n = 32 * NI * ldimg;
  float* mine = img + wave * n;
#pragma unroll
  for (int ib = 0; ib < NI; ++ib)
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 32 * ib + 8 * (r >> 2) + 4 * h + (r & 3);
        mine[i * ldimg + 32 * jb + j] = dW[ib * LDJ + jb][r];
      }
  __syncthreads();
  for (int i = tid; i < n; i += 64 * NW) {
    float s = img[i];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += img[w * n + i];
    slab[i] = s;
  }
  __syncthreads();
}
// per-feature vectors held as lanes = features (NV values per lane): wave order, one pass
template <int NV, int NW>
__device__ __forceinline__ void fold_vec_to_slab16(const float (&v)[NV], float* __restrict__ img,
                                                   float* __restrict__ slab, int n_out, int tid,
                                                   int wave, int lane) {
#pragma unroll
  for (int j = 0; j < NV; ++j) img[wave * (64 * NV) + 64 * j + lane] = v[j];
  __syncthreads();
  for (int i = tid; i < n_out; i += 64 * NW) {
    float s = img[i];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += img[w * (64 * NV) + i];
    slab[i] = s;
  }
  __syncthreads();
}

// ---- receiver-aligned segmented sums over a 16-row half tile ----------------------------------
// Rows [0, nh) of the fp32 tile are consecutive CSR positions; `ends` bit s = row s closes its
// receiver's segment, `rcv_of` (lanes 0..15) its receiver.  Lanes = features (D = 64).  The
// running sum is carried in `carry` from one half to the next (a segment may straddle the two
// halves of a 32-edge tile; it never leaves the tile), so the order of the additions -- and the
// result, bit for bit -- is that of the 32-row kernels.
template <typename Emit>
__device__ __forceinline__ void half_segment_sums(const float* __restrict__ tile, int ld, int nh,
                                                  unsigned ends, int rcv_of, int lane, float& carry,
                                                  Emit emit) {
  float v[NLAM_T16];
#pragma unroll
  for (int s = 0; s < NLAM_T16; ++s) v[s] = tile[s * ld + lane];
  float acc = carry;
#pragma unroll
  for (int s = 0; s < NLAM_T16; ++s) {
    if (s < nh) {   // wave-uniform
      acc += v[s];
      if ((ends >> s) & 1u) {
        emit(__builtin_amdgcn_readlane(rcv_of, s), acc);
        acc = 0.f;
      }
    }
  }
  carry = acc;
}
