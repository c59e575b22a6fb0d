// "Feature-split" fused kernels (gfx950) for hidden widths whose weight matrices do not fit the
// LDS next to row tiles: hidden_dim 256 in bf16-mixed arithmetic (BASELINE configs[4]; reference
// interaction_net.py:63-131, utils.py:191-214 under train_model.py's `--precision bf16-mixed`).
//
// One 256 x 256 bf16 image is 131 KB of the 160 KB LDS, so the weights leave LDS altogether:
//   * a workgroup has D / 32 wavefronts; wave w keeps ITS 32 output features of the weight
//     matrix as MFMA A-fragments in registers (32 x D bf16 = 64 registers per lane) for the
//     life of the persistent workgroup;
//   * the workgroup shares a tile of R = 64 rows (two 32-row blocks) as bf16 planes in LDS
//     (the B operand, read by every wave; 50 % of the LDS bandwidth at full MFMA rate);
//   * every wave produces the 32 x 64 block (its features, all rows) in accumulator layout;
//     LayerNorm statistics are exchanged between the waves through LDS (two floats per row and
//     wave); outputs go through an fp32 LDS tile so that global stores, row scatters and the
//     receiver-aligned segment sums stay whole-row and coalesced.
// Same entry points and semantics as fused_wide.hip (the host sequence in wide.py is shared);
// arithmetic: TERMS = 1 (NLAM_MFMA=bf16): plain bf16 products, fp32 accumulate, fp32 LayerNorm /
// residuals / aggregates, bf16 storage of the Linear outputs; TERMS = 3 (the default mode):
// split-bf16 operands (hi + lo planes, three products), every stored row fp32 -- fp32-grade
// results from the same kernels at about a third of the MFMA rate and twice the plane bytes.
#include <stdlib.h>

#include "fused_bf16x3.h"
#include "fused_common.h"
#include "fused_fs.h"

#define FS_R 64   // rows per workgroup tile

// The arithmetic of a call: 1 (NLAM_MFMA=bf16) or 3 (split bf16, the default mode); bf16-stored
// rows exist in the bf16 mode only.
#define FS_TERMS(what, any_bf16_rows)                                                          \
  const int terms_ = nlam_mfma_terms();                                                        \
  NLAM_REQUIRE(terms_ == 1 || terms_ == 3, what ": hidden 256 runs on the MFMA modes (bf16x3, bf16)"); \
  NLAM_REQUIRE(terms_ == 1 || !(any_bf16_rows), what ": bf16 rows exist under NLAM_MFMA=bf16 only")

template <int K, int TERMS>
struct FsW {   // register-resident slice of a weight matrix: A fragments of all K / 16 steps
  bf16x8 hi[K / 16];
  bf16x8 lo[TERMS == 3 ? K / 16 : 1];
};

__device__ __forceinline__ void fs_cvt8(const f32x4& v0, const f32x4& v1, bf16x8& hi, bf16x8& lo,
                                        bool want_lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)v0[j];
    hi[4 + j] = (__bf16)v1[j];
    if (want_lo) {
      lo[j] = (__bf16)(v0[j] - (float)hi[j]);
      lo[4 + j] = (__bf16)(v1[j] - (float)hi[4 + j]);
    }
  }
}

// rows [row0, row0 + 32) of W (n_out x D): the A operand of W . x for output features row0 ..
// (fragment order of b3_row_frag: lane (n, h) holds k = 16 s + 4 h + {0..3}, 16 s + 8 + 4 h + {0..3})
// k_in <= K columns are real (the rest of the fragment is zero); rows need not be aligned
template <int K, int TERMS>
__device__ __forceinline__ void fs_load_w_rows(FsW<K, TERMS>& A, const float* __restrict__ W,
                                               int64_t ldW, int row0, int n_out, int k_in,
                                               int lane) {
  const int n = lane & 31, h = lane >> 5;
  const bool ok = row0 + n < n_out;
  const float* wr = W + (int64_t)(ok ? row0 + n : 0) * ldW;
  const bool vec = (k_in == K) && (ldW % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0);
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    f32x4 v0, v1;
    if (vec) {
      v0 = *reinterpret_cast<const f32x4*>(wr + 16 * s + 4 * h);
      v1 = *reinterpret_cast<const f32x4*>(wr + 16 * s + 8 + 4 * h);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k0 = 16 * s + 4 * h + j, k1 = k0 + 8;
        v0[j] = k0 < k_in ? wr[k0] : 0.f;
        v1[j] = k1 < k_in ? wr[k1] : 0.f;
      }
    }
    if (!ok) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
    fs_cvt8(v0, v1, A.hi[s], A.lo[TERMS == 3 ? s : 0], TERMS == 3);
  }
}
// columns [col0, col0 + 32) of W (n_rows x >= col0 + 32): the A operand of W^T . g for output
// (input-gradient) features col0 ..: lane (i, h) holds W[k][col0 + i] for the fragment's eight k
template <int K, int TERMS>
__device__ __forceinline__ void fs_load_w_cols(FsW<K, TERMS>& A, const float* __restrict__ W,
                                               int64_t ldW, int col0, int n_rows, int lane) {
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < K / 16; ++s) {
    f32x4 v0, v1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k0 = 16 * s + 4 * h + j, k1 = k0 + 8;
      v0[j] = k0 < n_rows ? W[(int64_t)k0 * ldW + col0 + i] : 0.f;
      v1[j] = k1 < n_rows ? W[(int64_t)k1 * ldW + col0 + i] : 0.f;
    }
    fs_cvt8(v0, v1, A.hi[s], A.lo[TERMS == 3 ? s : 0], TERMS == 3);
  }
}

// The same slice, loaded by the whole workgroup through LDS: 64 rows of W at a time are read
// with coalesced float4 loads, written TRANSPOSED as bf16 (T[col][k], pitch 68) and picked up as
// fragments.  (The per-lane column walk above issues 128 strided dword loads per lane: ~50 us of
// prologue, which is the whole run time on the small mesh levels of Hi-LAM.)  scratch: >= 256 *
// 68 bf16 (twice that for the hi + lo images of TERMS = 3); every thread of the 512-thread
// workgroup must call; ends with a barrier.
template <int K, int TERMS>
__device__ __forceinline__ void fs_load_w_cols_lds(FsW<K, TERMS>& A, const float* __restrict__ W,
                                                   int64_t ldW, int col0, int n_rows,
                                                   void* scratch, int tid) {
  static_assert(K == 256, "fs_load_w_cols_lds: 256-wide slices only");
  constexpr int TP = 68;
  __bf16* T = reinterpret_cast<__bf16*>(scratch);
  __bf16* TL = T + 256 * TP;   // (TERMS == 3: the lo parts)
  const int lane = tid & 63;
  const int i = lane & 31, h = lane >> 5;
  const int c4 = tid & 63, rg = tid >> 6;   // 64 float4 chunk columns x 8 row groups
  const bool vec = (ldW % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0);
  // all K / 64 chunks' rows are requested up front (one round trip, not four); a thread takes 8
  // CONSECUTIVE rows of its 16-byte column chunk, so that the transposed image gets 8-byte
  // stores (8 per chunk and thread) instead of 2-byte ones (32)
  f32x4 v[K / 64][8];
#pragma unroll
  for (int c = 0; c < K / 64; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int row = 64 * c + 8 * rg + k;
      const float* wr = W + (int64_t)(row < n_rows ? row : 0) * ldW + 4 * c4;
      if (vec) {
        v[c][k] = *reinterpret_cast<const f32x4*>(wr);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[c][k][j] = wr[j];
      }
      if (row >= n_rows) v[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
  for (int c = 0; c < K / 64; ++c) {   // (unrolled: A.hi must stay in registers)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 r03, r47;   // rows 8 rg + 0..3 / + 4..7 of column 4 c4 + j
      bf16x4 l03, l47;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r03[k] = (__bf16)v[c][k][j];
        r47[k] = (__bf16)v[c][4 + k][j];
        if constexpr (TERMS == 3) {
          l03[k] = (__bf16)(v[c][k][j] - (float)r03[k]);
          l47[k] = (__bf16)(v[c][4 + k][j] - (float)r47[k]);
        }
      }
      const int off = (4 * c4 + j) * TP + 8 * rg;
      *reinterpret_cast<bf16x4*>(T + off) = r03;
      *reinterpret_cast<bf16x4*>(T + off + 4) = r47;
      if constexpr (TERMS == 3) {
        *reinterpret_cast<bf16x4*>(TL + off) = l03;
        *reinterpret_cast<bf16x4*>(TL + off + 4) = l47;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int off = (col0 + i) * TP + 16 * u + 4 * h;
      A.hi[4 * c + u] = b3_join(*reinterpret_cast<const bf16x4*>(T + off),
                                *reinterpret_cast<const bf16x4*>(T + off + 8));
      if constexpr (TERMS == 3)
        A.lo[4 * c + u] = b3_join(*reinterpret_cast<const bf16x4*>(TL + off),
                                  *reinterpret_cast<const bf16x4*>(TL + off + 8));
    }
    __syncthreads();
  }
}

// shared row tile as bf16 planes: hi[R][P] (| lo[R][P]), P = K + 4 elements
// PAD = 4 (pitch = 2 banks mod 64): conflict-free for the row-fragment reads of fs_gemm (lane t
// reads row t).  The weight-gradient pass reads its planes TRANSPOSED (ds_read_b64_tr_b16: the
// 16 lanes of a group touch 4 rows x 4 eight-byte chunks), where that pitch put 69 % of the LDS
// cycles into bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE); PAD = 16 (pitch = 8
// banks mod 64) makes those 16 accesses hit 16 distinct bank pairs.
template <int K, int TERMS, int PAD = 4>
struct FsPlanes {
  __bf16* hi;
  __bf16* lo;
  static constexpr int P = K + PAD;
  static constexpr size_t bytes = (size_t)(TERMS == 3 ? 2 : 1) * FS_R * (K + PAD) * sizeof(__bf16);
  __device__ __forceinline__ void init(void* base) {
    hi = reinterpret_cast<__bf16*>(base);
    lo = hi + (TERMS == 3 ? FS_R * P : 0);
  }
};

// acc[rb] += A . X^T for the two 32-row blocks of the shared tile
// (KSTEPS < K / 16: only the first 16 KSTEPS columns of X are non-zero -- the narrow head's backward)
template <int K, int TERMS, int KSTEPS = K / 16>
__device__ __forceinline__ void fs_gemm(f32x16 (&acc)[2], const FsW<K, TERMS>& A,
                                        const FsPlanes<K, TERMS>& X, int lane) {
  const int t = lane & 31, h = lane >> 5;
  constexpr int P = FsPlanes<K, TERMS>::P;
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int xo = (32 * rb + t) * P + 16 * s + 4 * h;
      const bf16x8 bh = b3_join(*reinterpret_cast<const bf16x4*>(X.hi + xo),
                                *reinterpret_cast<const bf16x4*>(X.hi + xo + 8));
      acc[rb] = B3_MFMA(A.hi[s], bh, acc[rb]);
      if constexpr (TERMS == 3) {
        const bf16x8 bl = b3_join(*reinterpret_cast<const bf16x4*>(X.lo + xo),
                                  *reinterpret_cast<const bf16x4*>(X.lo + xo + 8));
        acc[rb] = B3_MFMA(A.hi[s], bl, acc[rb]);
        acc[rb] = B3_MFMA(A.lo[s], bh, acc[rb]);
      }
    }
    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the fragment prefetch depth
  }
}

// ---- cooperative row staging -------------------------------------------------------------
// The workgroup (NT = 2 D threads) moves R = 64 rows of D floats as float4 chunks: thread `tid`
// owns chunk column c4 = tid % (D / 4) of rows rg + 8 k, rg = tid / (D / 4), k = 0..7.
template <int D>
struct FsMap {
  static constexpr int CPR = D / 4;       // float4 chunks per row
  int c4, rg;
  __device__ __forceinline__ FsMap(int tid) : c4(tid % CPR), rg(tid / CPR) {}
  __device__ __forceinline__ int row(int k) const { return rg + 8 * k; }
};

// sub-tile context (one 32-row block): batch item, first position, rows, receivers
struct FsSub {
  int64_t b;
  int p0, ne, r0, nr;
};
struct FsTiling {
  const int32_t* tiles;      // edge mode: (ntiles, 4); NULL = row mode
  int64_t ntiles;            // 32-row sub-tiles per batch item
  int64_t rows;              // positions per batch item
  const int32_t* csr_rec;
  const int32_t* csr_rowptr;
  int B;
};
__device__ __forceinline__ FsSub fs_sub(const FsTiling& tl, int64_t id) {
  FsSub s;
  const int64_t total = tl.ntiles * tl.B;
  if (id >= total) { s.b = 0; s.p0 = 0; s.ne = 0; s.r0 = 0; s.nr = 0; return s; }
  s.b = id / tl.ntiles;
  const int64_t k = id - s.b * tl.ntiles;
  if (tl.tiles != nullptr) {
    // The tile table is graph data no kernel writes, and the tile id of a workgroup is uniform:
    // read it through the constant address space, i.e. as a SCALAR load (own counter, scalar
    // cache).  As plain global loads the two headers of a 64-row tile were two serialized vector
    // round trips at the top of EVERY tile (the kernels' stores may alias, so the compiler
    // could not make them scalar itself).
    typedef int fs_i32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) fs_i32x4* const_i32x4_ptr;
    const fs_i32x4 hdr = ((const_i32x4_ptr)(uintptr_t)tl.tiles)[k];
    s.p0 = hdr.x; s.ne = hdr.y - hdr.x; s.r0 = hdr.z; s.nr = hdr.w - hdr.z;
  } else {
    s.p0 = (int)(k * NLAM_TILE);
    const int64_t left = tl.rows - (int64_t)s.p0;
    s.ne = (int)(left < NLAM_TILE ? left : NLAM_TILE);
    s.r0 = 0; s.nr = 0;
  }
  return s;
}

// =========================================================================== lin_fwd ===
// out = x W^T + bias, W: n_out (<= 256) x k_in (<= K), output features padded to 256.
// bf16 STORAGE of row operands (hidden 256; the dtype the reference's autocast gives Linear
// outputs and their gradients).  A lane moves 8 elements = 16 bytes, so a 256-wide row is 32
// lanes and the 512 threads of a workgroup cover 16 rows per pass (4 passes per 64-row tile):
// HALF the vector-memory instructions of the fp32 form -- which is what these kernels are bound
// by (an 8-byte-per-lane form with the fp32 thread map was slower than fp32 storage).
// Pointers to such operands are bf16 pointers passed as float*; pitches count bf16 elements.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 fs_bf16x8;
__device__ __forceinline__ void fs_cvt8f(const fs_bf16x8& v, float (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (float)v[j];
}
__device__ __forceinline__ fs_bf16x8 fs_cvt8b(const float (&x)[8]) {
  fs_bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)x[j];
  return v;
}
__device__ __forceinline__ fs_bf16x8 fs_ld8(const float* ptr, int64_t idx, int c8) {
  return reinterpret_cast<const fs_bf16x8*>(reinterpret_cast<const __bf16*>(ptr) + idx)[c8];
}
__device__ __forceinline__ void fs_st8(float* ptr, int64_t idx, int c8, const fs_bf16x8& v) {
  reinterpret_cast<fs_bf16x8*>(reinterpret_cast<__bf16*>(ptr) + idx)[c8] = v;
}
// 8 bf16 -> a plane row (two 8-byte LDS writes: plane rows are only 8-byte aligned)
__device__ __forceinline__ void fs_plane_put8(__bf16* plane, int off, const fs_bf16x8& v) {
  const bf16x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
  *reinterpret_cast<bf16x4*>(plane + off) = lo;
  *reinterpret_cast<bf16x4*>(plane + off + 4) = hi;
}

struct FsLinParams {
  RowView x;                 // width = k_in
  const float* W; int64_t ldW; const float* bias; int n_out;
  float* out; int64_t out_bstride; int64_t out_ld;
  const float* add; int64_t add_bstride; int64_t add_ld;   // optional addend of out (rows like out)
  int64_t rows; int B;
  int x_vec;                 // x rows float4-loadable and k_in == K
  int out_bf16;              // out rows stored as bf16 (n_out == D; pitches in bf16 elements)
};

// rows of the workgroup tile -> bf16 planes (zero beyond nrows / k_in); float4 or scalar source
template <int D, int K, int TERMS>
__device__ __forceinline__ void fs_stage_x(const FsPlanes<K, TERMS>& X, const RowView& x, int64_t b,
                                           int64_t r0, int nrows, bool vec, int tid) {
  const float* xb = x.ptr + b * x.bstride + r0 * x.ld;
  if (vec) {
    // 2 D threads over K / 4 chunk columns: K == D -> 8 rows per pass, 8 passes
    constexpr int CPR = K / 4;
    constexpr int RPP = (2 * D) / CPR;          // rows per pass
    constexpr int NP = FS_R / RPP;
    const int c4 = tid % CPR, rg = tid / CPR;
    f32x4 v[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int r = rg + RPP * k;
      v[k] = reinterpret_cast<const f32x4*>(xb + (int64_t)(r < nrows ? r : nrows - 1) * x.ld)[c4];
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int r = rg + RPP * k;
      f32x4 xx = v[k];
      if (r >= nrows) xx = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x4 hi, lo;
      b3_split4(xx, hi, lo);
      *reinterpret_cast<bf16x4*>(X.hi + r * X.P + 4 * c4) = hi;
      if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(X.lo + r * X.P + 4 * c4) = lo;
    }
  } else {
    for (int idx = tid; idx < FS_R * K; idx += 2 * D) {
      const int r = idx / K, c = idx - r * K;
      const float v = (r < nrows && c < x.width) ? xb[(int64_t)r * x.ld + c] : 0.f;
      const __bf16 hi = (__bf16)v;
      X.hi[r * X.P + c] = hi;
      if constexpr (TERMS == 3) X.lo[r * X.P + c] = (__bf16)(v - (float)hi);
    }
  }
}

// accumulator blocks of this wave (features 32 wave .., rows of both blocks) -> fp32 tile
template <int LDO>
__device__ __forceinline__ void fs_acc_to_tile(const f32x16 (&acc)[2], float* __restrict__ otile,
                                               int wave, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {acc[rb][4 * q], acc[rb][4 * q + 1], acc[rb][4 * q + 2], acc[rb][4 * q + 3]};
      *reinterpret_cast<f32x4*>(otile + (32 * rb + t) * LDO + 32 * wave + 8 * q + 4 * h) = o;
    }
}
template <int LDO>
__device__ __forceinline__ void fs_tile_to_acc(f32x16 (&acc)[2], const float* __restrict__ otile,
                                               int wave, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(otile + (32 * rb + t) * LDO + 32 * wave +
                                                       8 * q + 4 * h);
      acc[rb][4 * q] = o[0]; acc[rb][4 * q + 1] = o[1];
      acc[rb][4 * q + 2] = o[2]; acc[rb][4 * q + 3] = o[3];
    }
}
// per-feature vector (bias / gamma / beta) of this wave's block, zero beyond n
__device__ __forceinline__ f32x16 fs_vec_block(const float* __restrict__ p, int n, int wave, int lane) {
  const int h = lane >> 5;
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 32 * wave + 8 * q + 4 * h + j;
      v[4 * q + j] = (p != nullptr && f < n) ? p[f] : 0.f;
    }
  return v;
}

// TRANS: out = x W (+ add) with W: k_in x n_out -- the data gradient gx = gy W of a Linear
template <int D, int K, int TERMS, bool TRANS>
__device__ __forceinline__ void fs_lin_fwd_body(const FsLinParams& p, const int bid, const int gdim,
                                                float* smem) {
  constexpr int LDO = D + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tile / batch address math goes to the scalar unit
  FsPlanes<K, TERMS> X;
  X.init(smem);
  float* otile = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + FsPlanes<K, TERMS>::bytes);
  FsW<K, TERMS> A;
  // (TERMS == 3: hi + lo transposed images = 69.6 KB, over the planes and the output tile, which
  //  are contiguous from smem and unused until the first tile)
  if constexpr (TRANS && K == 256) fs_load_w_cols_lds<K, TERMS>(A, p.W, p.ldW, 32 * wave, p.x.width, TERMS == 1 ? (void*)otile : (void*)smem, tid);
  else if constexpr (TRANS) fs_load_w_cols<K, TERMS>(A, p.W, p.ldW, 32 * wave, p.x.width, lane);
  else fs_load_w_rows<K, TERMS>(A, p.W, p.ldW, 32 * wave, p.n_out, p.x.width, lane);
  const f32x16 bias = fs_vec_block(p.bias, p.n_out, wave, lane);
  constexpr int CPR = D / 4;
  const int oc4 = tid % CPR, org = tid / CPR;    // output chunk column / row group (8 per pass)
  const int64_t tiles_per_b = (p.rows + FS_R - 1) / FS_R;
  const int64_t ntiles = tiles_per_b * p.B;
  // float4 rows: the NEXT tile's rows (and addend rows) are in flight during this tile's GEMM
  constexpr int XCPR = K / 4, XRPP = (2 * D) / XCPR, XNP = FS_R / XRPP;
  const int xc4 = tid % XCPR, xrg = tid / XCPR;
  const bool x_vec = p.x_vec != 0;
  f32x4 vx[XNP];
  auto tile_of = [&](int64_t tt, int64_t& b, int64_t& r0, int& nrows) {
    b = tt / tiles_per_b;
    r0 = (tt - b * tiles_per_b) * FS_R;
    nrows = (int)((p.rows - r0) < FS_R ? (p.rows - r0) : FS_R);
  };
  auto issue = [&](int64_t tt) {
    int64_t b, r0; int nrows;
    tile_of(tt, b, r0, nrows);
    const float* xb = p.x.ptr + b * p.x.bstride + r0 * p.x.ld;
#pragma unroll
    for (int k = 0; k < XNP; ++k) {
      const int r = xrg + XRPP * k;
      vx[k] = reinterpret_cast<const f32x4*>(xb + (int64_t)(r < nrows ? r : nrows - 1) * p.x.ld)[xc4];
    }
  };
  if (x_vec && (int64_t)bid < ntiles) issue(bid);
  for (int64_t tt = bid; tt < ntiles; tt += gdim) {
    int64_t b, r0; int nrows;
    tile_of(tt, b, r0, nrows);
    if (x_vec) {
#pragma unroll
      for (int k = 0; k < XNP; ++k) {
        const int r = xrg + XRPP * k;
        f32x4 xx = vx[k];
        if (r >= nrows) xx = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x4 hi, lo;
        b3_split4(xx, hi, lo);
        *reinterpret_cast<bf16x4*>(X.hi + r * X.P + 4 * xc4) = hi;
        if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(X.lo + r * X.P + 4 * xc4) = lo;
      }
    } else {
      fs_stage_x<D, K, TERMS>(X, p.x, b, r0, nrows, false, tid);
    }
    __syncthreads();
    if (x_vec && tt + gdim < ntiles) issue(tt + gdim);
    const int nc4 = p.n_out >> 2;   // (n_out % 4 == 0, checked on the host)
    f32x4 av[8];
    if (p.add != nullptr) {
      const float* ab = p.add + b * p.add_bstride + r0 * p.add_ld;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = org + 8 * k;
        av[k] = reinterpret_cast<const f32x4*>(ab + (int64_t)(r < nrows ? r : nrows - 1) * p.add_ld)
            [oc4 < nc4 ? oc4 : 0];
      }
    }
    f32x16 acc[2] = {bias, bias};
    fs_gemm<K, TERMS>(acc, A, X, lane);
    fs_acc_to_tile<LDO>(acc, otile, wave, lane);
    __syncthreads();
    {
      float* ob = p.out + b * p.out_bstride + r0 * p.out_ld;
      if (p.add != nullptr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = org + 8 * k;
          if (r < nrows && oc4 < nc4)
            reinterpret_cast<f32x4*>(ob + (int64_t)r * p.out_ld)[oc4] =
                *reinterpret_cast<const f32x4*>(otile + r * LDO + 4 * oc4) + av[k];
        }
      } else if (p.out_bf16) {
        const int oc8 = tid % (D / 8), orh = tid / (D / 8);     // 16 rows per pass
        const int64_t o0 = b * p.out_bstride + r0 * p.out_ld;
#pragma unroll
        for (int k = 0; k < FS_R / 16; ++k) {
          const int r = orh + 16 * k;
          if (r < nrows) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(otile + r * LDO + 8 * oc8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(otile + r * LDO + 8 * oc8 + 4);
            const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            fs_st8(p.out, o0 + (int64_t)r * p.out_ld, oc8, fs_cvt8b(x));
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = org + 8 * k;
          if (r < nrows && oc4 < nc4)
            reinterpret_cast<f32x4*>(ob + (int64_t)r * p.out_ld)[oc4] =
                *reinterpret_cast<const f32x4*>(otile + r * LDO + 4 * oc4);
        }
      }
    }
    // (the next staging overwrites the planes only after every wave passed the barrier behind
    //  its fragment reads; the otile reads are ordered before the next tile's otile writes by
    //  the barrier after the next staging)
  }
}

// one launch, up to NLAM_WIDE_MAXP independent problems (fused_common.h: WideMulti)
template <int D, int K, int TERMS, bool TRANS = false>
__global__ __launch_bounds__(2 * D) void fs_lin_fwd_kernel(WideMulti<FsLinParams> m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = wide_multi_find(m, blockIdx.x);
  fs_lin_fwd_body<D, K, TERMS, TRANS>(m.p[k], blockIdx.x - m.first[k], m.first[k + 1] - m.first[k],
                                      smem);
}

static unsigned fs_grid(int64_t ntiles) {
  int64_t g = ntiles;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// grid shares of a multi launch: fs_grid() each, scaled down to <= 256 workgroups in all (one
// 512-thread workgroup is resident per CU)
template <typename P, typename TilesOf>
static unsigned fs_multi_grid(WideMulti<P>& m, TilesOf tiles_of) {
  // shares proportional to work, one round of the device in all (fused_common.h)
  int64_t rounds[NLAM_WIDE_MAXP], g[NLAM_WIDE_MAXP];
  for (int k = 0; k < m.n; ++k) rounds[k] = (tiles_of(m.p[k]) + 1 - 1) / 1;
  nlam_multi_shares(m.n, rounds, g, 256);
  m.first[0] = 0;
  for (int k = 0; k < m.n; ++k) m.first[k + 1] = m.first[k] + (int)g[k];
  for (int k = m.n; k < NLAM_WIDE_MAXP; ++k) m.first[k + 1] = m.first[m.n];
  return (unsigned)m.first[m.n];
}

template <int D, int K, int TERMS, bool TRANS = false>
static int launch_fs_lin_fwd(WideMulti<FsLinParams>& m, hipStream_t s) {
  const size_t lds = FsPlanes<K, TERMS>::bytes + (size_t)FS_R * (D + 4) * sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "fs_lin_fwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = fs_lin_fwd_kernel<D, K, TERMS, TRANS>;
  NLAM_BIG_LDS(kern, "fs_lin_fwd_kernel");
  const unsigned grid = fs_multi_grid(m, [](const FsLinParams& p) {
    return ((p.rows + FS_R - 1) / FS_R) * p.B;
  });
  kern<<<grid, 2 * D, lds, s>>>(m);
  NLAM_CHECK_LAUNCH("fs_lin_fwd_kernel");
  return 0;
}
template <int D, int K, int TERMS, bool TRANS = false>
static int launch_fs_lin_fwd(const FsLinParams& p, hipStream_t s) {
  WideMulti<FsLinParams> m;
  m.n = 1;
  m.p[0] = p;
  return launch_fs_lin_fwd<D, K, TERMS, TRANS>(m, s);
}

// out = x W^T + bias for x: (B, rows, k_in <= 256) fp32, W: 256 x k_in; NLAM_MFMA=bf16.
int nlam_fs_lin_fwd_256(const float* x, int64_t x_bstride, int64_t x_ld, int k_in, const float* W,
                        int64_t ldW, const float* bias, int n_out, float* out, int64_t out_bstride,
                        int64_t out_ld, int64_t B, int64_t rows, int out_bf16, void* stream) {
  NLAM_REQUIRE(!out_bf16 || (n_out == 256 && out_ld % 8 == 0 && out_bstride % 8 == 0),
               "fs_lin_fwd: bf16 output rows are 256 wide with pitches %% 8 == 0");
  if (B <= 0 || rows <= 0) return 0;
  FS_TERMS("nlam_lin_fwd", out_bf16);
  NLAM_REQUIRE(n_out >= 4 && n_out <= 256 && n_out % 4 == 0, "fs_lin_fwd: n_out %d unsupported", n_out);
  NLAM_REQUIRE(k_in >= 1 && k_in <= 256, "fs_lin_fwd: k_in %d out of range", k_in);
  NLAM_REQUIRE(view_vec_ok(out, out_bstride, out_ld, n_out),
               "fs_lin_fwd: output rows must be 16-byte aligned with pitch %% 4 == 0");
  FsLinParams p;
  p.x = RowView{x, x_bstride, x_ld, k_in};
  p.W = W; p.ldW = ldW; p.bias = bias; p.n_out = n_out;
  p.out = out; p.out_bstride = out_bstride; p.out_ld = out_ld;
  p.add = nullptr; p.add_bstride = 0; p.add_ld = 0;
  p.rows = rows; p.B = (int)B; p.out_bf16 = out_bf16;
  hipStream_t s = (hipStream_t)stream;
  if (k_in <= 32) { p.x_vec = (k_in == 32 && view_vec_ok(x, x_bstride, x_ld, 32)); return terms_ == 3 ? launch_fs_lin_fwd<256, 32, 3>(p, s) : launch_fs_lin_fwd<256, 32, 1>(p, s); }
  if (k_in <= 64) { p.x_vec = (k_in == 64 && view_vec_ok(x, x_bstride, x_ld, 64)); return terms_ == 3 ? launch_fs_lin_fwd<256, 64, 3>(p, s) : launch_fs_lin_fwd<256, 64, 1>(p, s); }
  NLAM_REQUIRE(k_in == 256 || k_in <= 256, "fs_lin_fwd: k_in %d", k_in);
  p.x_vec = (k_in == 256 && view_vec_ok(x, x_bstride, x_ld, 256));
  return terms_ == 3 ? launch_fs_lin_fwd<256, 256, 3>(p, s) : launch_fs_lin_fwd<256, 256, 1>(p, s);
}

// gx = gy W (+ gx_add), W: 256 x 256 (data gradient of a Linear); NLAM_MFMA=bf16.
int nlam_fs_lin_bwd_data_256(const float* gy, int64_t gy_bstride, int64_t gy_ld, const float* W,
                             int64_t ldW, float* gx, int64_t gx_bstride, int64_t gx_ld,
                             const float* gx_add, int64_t ga_bstride, int64_t ga_ld, int64_t B,
                             int64_t rows, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  FS_TERMS("nlam_lin_bwd_data", 0);
  NLAM_REQUIRE(view_vec_ok(gy, gy_bstride, gy_ld, 256) && view_vec_ok(gx, gx_bstride, gx_ld, 256) &&
                   (gx_add == nullptr || view_vec_ok(gx_add, ga_bstride, ga_ld, 256)),
               "fs_lin_bwd_data: operand rows must be 16-byte aligned with pitch %% 4 == 0");
  FsLinParams p;
  p.x = RowView{gy, gy_bstride, gy_ld, 256};
  p.W = W; p.ldW = ldW; p.bias = nullptr; p.n_out = 256;
  p.out = gx; p.out_bstride = gx_bstride; p.out_ld = gx_ld;
  p.add = gx_add; p.add_bstride = ga_bstride; p.add_ld = ga_ld;
  p.rows = rows; p.B = (int)B; p.x_vec = 1; p.out_bf16 = 0;
  return terms_ == 3 ? launch_fs_lin_fwd<256, 256, 3, true>(p, (hipStream_t)stream)
                     : launch_fs_lin_fwd<256, 256, 1, true>(p, (hipStream_t)stream);
}

// accumulator blocks -> bf16 planes (hi only / hi + lo)
template <int D, int TERMS>
__device__ __forceinline__ void fs_acc_to_planes(const f32x16 (&acc)[2], const FsPlanes<D, TERMS>& X,
                                                 int wave, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[rb][4 * q], acc[rb][4 * q + 1], acc[rb][4 * q + 2], acc[rb][4 * q + 3]};
      bf16x4 hi, lo;
      b3_split4(v, hi, lo);
      const int off = (32 * rb + t) * X.P + 32 * wave + 8 * q + 4 * h;
      *reinterpret_cast<bf16x4*>(X.hi + off) = hi;
      if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(X.lo + off) = lo;
    }
}

// n <= NLAM_WIDE_MAXP independent 256 -> 256 problems in one launch (aligned 256-wide rows)
int nlam_fs_lin_fwd_multi_256(int n, const float* const* x, const int64_t* x_bstride,
                              const int64_t* x_ld, const float* const* W, const int64_t* ldW,
                              const float* const* bias, float* const* out,
                              const int64_t* out_bstride, const int64_t* out_ld, const int64_t* B,
                              const int64_t* rows, int out_bf16_mask, void* stream) {
  FS_TERMS("nlam_lin_fwd_multi", out_bf16_mask);
  WideMulti<FsLinParams> m;
  m.n = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    NLAM_REQUIRE(view_vec_ok(x[k], x_bstride[k], x_ld[k], 256) &&
                     view_vec_ok(out[k], out_bstride[k], out_ld[k], 256),
                 "nlam_lin_fwd_multi: operand rows must be 16-byte aligned, width 256");
    FsLinParams& p = m.p[m.n++];
    p.x = RowView{x[k], x_bstride[k], x_ld[k], 256};
    p.W = W[k]; p.ldW = ldW[k]; p.bias = bias[k]; p.n_out = 256;
    p.out = out[k]; p.out_bstride = out_bstride[k]; p.out_ld = out_ld[k];
    p.add = nullptr; p.add_bstride = 0; p.add_ld = 0;
    p.rows = rows[k]; p.B = (int)B[k]; p.x_vec = 1; p.out_bf16 = (out_bf16_mask >> k) & 1;
    NLAM_REQUIRE(!p.out_bf16 || (out_ld[k] % 8 == 0 && out_bstride[k] % 8 == 0),
                 "nlam_lin_fwd_multi: bf16 output pitches must be multiples of 8");
  }
  if (m.n == 0) return 0;
  return terms_ == 3 ? launch_fs_lin_fwd<256, 256, 3>(m, (hipStream_t)stream)
                     : launch_fs_lin_fwd<256, 256, 1>(m, (hipStream_t)stream);
}

int nlam_fs_lin_bwd_data_multi_256(int n, const float* const* gy, const int64_t* gy_bstride,
                                   const int64_t* gy_ld, const float* const* W, const int64_t* ldW,
                                   float* const* gx, const int64_t* gx_bstride, const int64_t* gx_ld,
                                   const float* const* gx_add, const int64_t* ga_bstride,
                                   const int64_t* ga_ld, const int64_t* B, const int64_t* rows,
                                   void* stream) {
  FS_TERMS("nlam_lin_bwd_data_multi", 0);
  WideMulti<FsLinParams> m;
  m.n = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    NLAM_REQUIRE(view_vec_ok(gy[k], gy_bstride[k], gy_ld[k], 256) &&
                     view_vec_ok(gx[k], gx_bstride[k], gx_ld[k], 256) &&
                     (gx_add[k] == nullptr || view_vec_ok(gx_add[k], ga_bstride[k], ga_ld[k], 256)),
                 "nlam_lin_bwd_data_multi: operand rows must be 16-byte aligned, width 256");
    FsLinParams& p = m.p[m.n++];
    p.x = RowView{gy[k], gy_bstride[k], gy_ld[k], 256};
    p.W = W[k]; p.ldW = ldW[k]; p.bias = nullptr; p.n_out = 256;
    p.out = gx[k]; p.out_bstride = gx_bstride[k]; p.out_ld = gx_ld[k];
    p.add = gx_add[k]; p.add_bstride = ga_bstride[k]; p.add_ld = ga_ld[k];
    p.rows = rows[k]; p.B = (int)B[k]; p.x_vec = 1; p.out_bf16 = 0;
  }
  if (m.n == 0) return 0;
  return terms_ == 3 ? launch_fs_lin_fwd<256, 256, 3, true>(m, (hipStream_t)stream)
                     : launch_fs_lin_fwd<256, 256, 1, true>(m, (hipStream_t)stream);
}

// Diagnostic (NLAM_STAMP=1): wave 0 of every workgroup of fs_tail_fwd adds the s_memtime cycles
// of each phase of each tile to g_fs_stamps (tools/stamp_fs.py).
__device__ unsigned long long g_fs_stamps[16];
int nlam_wide_stamps(unsigned long long* out, int reset);   // fused_wide.hip (NLAM_STAMP_WIDE=1)
extern "C" int nlam_debug_fs_stamps(unsigned long long* out, int reset) {
  if (getenv("NLAM_STAMP_WIDE") != nullptr) return nlam_wide_stamps(out, reset);
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fs_stamps), sizeof(unsigned long long) * 16) != hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fs_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#define FSSTAMP(k)                                                  \
  if (stamp) {                                                      \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    __builtin_amdgcn_s_waitcnt(0xC07F);                             \
    __builtin_amdgcn_sched_barrier(0);                              \
    fst[k] += now_ - fprev;                                         \
    fprev = now_;                                                   \
  }
static int fs_stamp_flag() {
  static const int f = getenv("NLAM_STAMP") != nullptr ? 1 : 0;
  return f;
}

// ========================================================================= tail forward ===
struct FsTailFwdParams {
  FsTiling tl;
  RowView a; const int32_t* idx_a;
  RowView b; const int32_t* idx_b;       // optional
  RowView c; const int32_t* idx_c;       // optional
  const float* W2; int64_t ldW2; const float* b2; const float* gamma; const float* beta;
  int n_out;
  float* h_out; int64_t h_bstride;
  void* z_keep; int64_t z_bstride;       // optional (HAS_LN): the pre-LayerNorm rows, position order: bf16 (TERMS 1) / fp32 (TERMS 3)
  float* y; int64_t y_bstride; int64_t y_ld; const int32_t* idx_y;
  RowView res;
  float* agg; int64_t agg_bstride; int64_t agg_ld; const float* inv_deg;
  int vec_y;
  int stamp;
};

// slot r of the workgroup tile (sub-tile r >> 5, slot r & 31): its position, clamped to a valid one
// (the two sub-tile contexts are separate variables selected by value: an indexed array of
//  structs would live in scratch)
struct FsSub2 {
  FsSub s0, s1;
  __device__ __forceinline__ int64_t b(int rb) const { return rb ? s1.b : s0.b; }
  __device__ __forceinline__ int p0(int rb) const { return rb ? s1.p0 : s0.p0; }
  __device__ __forceinline__ int ne(int rb) const { return rb ? s1.ne : s0.ne; }
};
__device__ __forceinline__ int fs_slot_pos(const FsSub2& sub, int r) {
  const int rb = r >> 5, t = r & 31;
  const int ne = sub.ne(rb);
  const int p0 = sub.p0(rb);
  return ne > 0 ? p0 + (t < ne ? t : ne - 1) : (p0 > 0 ? p0 - 1 : 0);   // (empty sub-tile: p0 may be M)
}

// tile-local segmented sums of one sub-tile over a 64-feature column chunk (lanes = features):
// fast path when every receiver of the tile has in-edges, row-pointer loop otherwise
template <typename Emit>
__device__ __forceinline__ void fs_segment_sums(const float* __restrict__ tile, int ld,
                                                const FsSub& sb, const FsTiling& tl, int lane,
                                                Emit emit) {
  if (sb.ne <= 0) {   // a sub-tile of receivers without in-edges: their sums are zero
    for (int i = 0; i < sb.nr; ++i) emit(sb.r0 + i, 0.f);
    return;
  }
  const int t = lane & 31;
  const int rcv = tl.csr_rec[sb.p0 + (t < sb.ne ? t : sb.ne - 1)];
  const int ri = sb.r0 + (lane < sb.nr ? lane : sb.nr);
  const int rp = tl.csr_rowptr[ri] - sb.p0;
  const int rpn = __shfl_down(rp, 1, 64);
  const bool dense = __all((lane >= sb.nr) || (rpn > rp));
  if (dense) {
    tile_segment_sums<64>(tile, ld, sb.ne, rcv, lane,
                          [&](int r, int f0, float acc) { emit(r, acc); (void)f0; });
  } else {
    for (int i = 0; i < sb.nr; ++i) {
      const int beg = __shfl(rp, i, 64), end = __shfl(rp, i + 1, 64);
      float acc = 0.f;
      for (int sidx = beg; sidx < end; ++sidx) acc += tile[sidx * ld + lane];
      emit(sb.r0 + i, acc);
    }
  }
}

// IO16: the sources a / b / c and h_out are bf16 rows (16-byte lanes, 16 rows per pass); h is then
// rounded to bf16 before the SiLU (it IS a bf16 tensor: the backward reads the stored value).
template <int D, bool HAS_LN, int TERMS, bool IO16 = false>
__global__ __launch_bounds__(2 * D) void fs_tail_fwd_kernel(FsTailFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 2 * D, NW = D / 32, LDO = D + 4, CPR = D / 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tile / batch address math goes to the scalar unit
  const int t = lane & 31, h = lane >> 5;
  FsPlanes<D, TERMS> S;
  S.init(smem);
  float* mtile = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + FsPlanes<D, TERMS>::bytes);
  float* red = mtile + FS_R * LDO;                 // [2][FS_R][NW]
  int* itab0 = reinterpret_cast<int*>(red + 2 * FS_R * NW);   // [2][4][FS_R]: a, b, c, y
  FsW<D, TERMS> A;
  const unsigned long long fstart = p.stamp ? __builtin_amdgcn_s_memtime() : 0;
  const int c4 = tid % CPR, rg = tid / CPR;        // staging map: rows rg + 8 k
  const int64_t nsub = p.tl.ntiles * p.tl.B;
  const int64_t ntiles = (nsub + 1) / 2;
  // slot -> row tables, double buffered: the NEXT tile's indices are fetched while this tile
  // computes (a dependent global load per tile would otherwise sit on the critical path of the
  // lock-stepped workgroup)
  int nidx[4];
  auto fetch_idx = [&](int64_t tt) {
    const FsSub2 sub = {fs_sub(p.tl, 2 * tt), fs_sub(p.tl, 2 * tt + 1)};
    const int pos = fs_slot_pos(sub, tid & (FS_R - 1));
    nidx[0] = p.idx_a ? p.idx_a[pos] : pos;
    nidx[1] = (p.b.ptr && p.idx_b) ? p.idx_b[pos] : pos;
    nidx[2] = (p.c.ptr && p.idx_c) ? p.idx_c[pos] : pos;
    nidx[3] = p.idx_y ? p.idx_y[pos] : pos;
  };
  auto put_idx = [&](int* tab) {
#pragma unroll
    for (int k = 0; k < 4; ++k) tab[k * FS_R + tid] = nidx[k];
  };
  int par = 0;
  const bool stamp = p.stamp != 0;
  unsigned long long fst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  fs_load_w_rows<D, TERMS>(A, p.W2, p.ldW2, 32 * wave, p.n_out, D, lane);
  unsigned long long fprev = stamp ? __builtin_amdgcn_s_memtime() : 0;
  fst[7] = fprev - fstart;   // weight slice
  // (requesting the indices BEFORE the weight slice moved their 4 k cycles into the weight phase
  // and saved nothing: tools/stamp_fs.py, N = 81)
  if (tid < FS_R && (int64_t)blockIdx.x < ntiles) {
    fetch_idx(blockIdx.x);
    put_idx(itab0);
  }
  __syncthreads();
  FSSTAMP(6)   // first tile's header -> indices -> tables + barrier
  for (int64_t tt = blockIdx.x; tt < ntiles; tt += gridDim.x, par ^= 1) {
    const FsSub2 sub = {fs_sub(p.tl, 2 * tt), fs_sub(p.tl, 2 * tt + 1)};
    const int* itab = itab0 + par * 4 * FS_R;
    const bool more = tt + gridDim.x < ntiles;
    if (tid < FS_R && more) fetch_idx(tt + gridDim.x);
    // ---- h = a + b + c (row layout), keep h, s = silu(h) -> planes
    if constexpr (IO16) {
      const int c8 = tid % (D / 8), rh = tid / (D / 8);
      fs_bf16x8 va[4], vb[4], vc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = rh + 16 * k, rb = r >> 5;
        va[k] = fs_ld8(p.a.ptr, sub.b(rb) * p.a.bstride + (int64_t)itab[r] * p.a.ld, c8);
        if (p.b.ptr)
          vb[k] = fs_ld8(p.b.ptr, sub.b(rb) * p.b.bstride + (int64_t)itab[FS_R + r] * p.b.ld, c8);
        if (p.c.ptr)
          vc[k] = fs_ld8(p.c.ptr, sub.b(rb) * p.c.bstride + (int64_t)itab[2 * FS_R + r] * p.c.ld, c8);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = rh + 16 * k, rb = r >> 5;
        const bool valid = (r & 31) < sub.ne(rb);
        float x[8], y[8];
        fs_cvt8f(va[k], x);
        if (p.b.ptr) {
          fs_cvt8f(vb[k], y);
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] += y[j];
        }
        if (p.c.ptr) {
          fs_cvt8f(vc[k], y);
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] += y[j];
        }
        const fs_bf16x8 hb = fs_cvt8b(x);          // h as stored
        if (valid && p.h_out != nullptr)
          fs_st8(p.h_out, sub.b(rb) * p.h_bstride + (int64_t)(sub.p0(rb) + (r & 31)) * D, c8, hb);
        fs_cvt8f(hb, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = valid ? nlam_silu(x[j]) : 0.f;
        fs_plane_put8(S.hi, r * S.P + 8 * c8, fs_cvt8b(x));
        __builtin_amdgcn_sched_barrier(0);   // (one row block's temporaries at a time)
      }
    } else
    {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = rg + 8 * k;
        const float* ab = p.a.ptr + sub.b(r >> 5) * p.a.bstride;
        v[k] = reinterpret_cast<const f32x4*>(ab + (int64_t)itab[r] * p.a.ld)[c4];
      }
      if (p.b.ptr) {
        f32x4 u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = rg + 8 * k;
          const float* bb = p.b.ptr + sub.b(r >> 5) * p.b.bstride;
          u[k] = reinterpret_cast<const f32x4*>(bb + (int64_t)itab[FS_R + r] * p.b.ld)[c4];
        }
        if (p.c.ptr) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int r = rg + 8 * k;
            const float* cb = p.c.ptr + sub.b(r >> 5) * p.c.bstride;
            u[k] += reinterpret_cast<const f32x4*>(cb + (int64_t)itab[2 * FS_R + r] * p.c.ld)[c4];
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += u[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = rg + 8 * k;
        const int rb = r >> 5;
        const bool valid = (r & 31) < sub.ne(rb);
        f32x4 x = v[k];
        if (valid && p.h_out != nullptr)
          reinterpret_cast<f32x4*>(p.h_out + sub.b(rb) * p.h_bstride +
                                   (int64_t)(sub.p0(rb) + (r & 31)) * D)[c4] = x;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = valid ? nlam_silu(x[j]) : 0.f;
        bf16x4 hi, lo;
        b3_split4(x, hi, lo);
        *reinterpret_cast<bf16x4*>(S.hi + r * S.P + 4 * c4) = hi;
        if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(S.lo + r * S.P + 4 * c4) = lo;
      }
    }
    FSSTAMP(0)   // issue + landing of the gathered rows, h store, silu -> planes
    __syncthreads();
    FSSTAMP(1)   // barrier (the slowest wave's rows)
    // residual rows: in flight during the GEMM and the LayerNorm exchange
    f32x4 rv[8];
    const bool res_vec = p.y != nullptr && p.vec_y && p.res.ptr != nullptr;
    if (res_vec) {
      const int cc = c4 < (p.n_out >> 2) ? c4 : 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = rg + 8 * k;
        rv[k] = reinterpret_cast<const f32x4*>(p.res.ptr + sub.b(r >> 5) * p.res.bstride +
                                               (int64_t)itab[3 * FS_R + r] * p.res.ld)[cc];
      }
    }
    f32x16 z[2];
    z[0] = fs_vec_block(p.b2, p.n_out, wave, lane);
    z[1] = z[0];
    // (narrow head without LayerNorm: the waves whose 32 output features lie past n_out have no
    // product to form -- seven of the eight on the 17-wide output map)
    if (HAS_LN || 32 * wave < p.n_out) fs_gemm<D, TERMS>(z, A, S, lane);
    FSSTAMP(2)   // residual issue + GEMM
    if (HAS_LN) {
      // The Linear output is a bf16 tensor, as under the reference's autocast (LayerNorm then
      // works in fp32 on those bf16 values): the backward reads the kept bf16 rows instead of
      // repeating this GEMM, and sees exactly the values normalised here.
      // (split-bf16 mode: z stays fp32 and is kept as fp32 rows, hi + lo of the planes)
      if constexpr (TERMS == 1) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) z[rb][r] = (float)(__bf16)z[rb][r];
      }
      // LayerNorm over the D features of a row = over the NW waves: exchange through LDS
      const float inv_n = 1.0f / (float)p.n_out;
      float mean[2], rstd[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float sm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += z[rb][r];
        sm = lane_xor32_sum(sm);
        if (h == 0) red[(32 * rb + t) * NW + wave] = sm;
      }
      __syncthreads();
      if (p.z_keep != nullptr) fs_acc_to_planes<D, TERMS>(z, S, wave, lane);   // (every GEMM is done)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sm += red[(32 * rb + t) * NW + w];
        mean[rb] = sm * inv_n;
        float vs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dlt = z[rb][r] - mean[rb];
          vs += dlt * dlt;
        }
        vs = lane_xor32_sum(vs);
        if (h == 0) red[FS_R * NW + (32 * rb + t) * NW + wave] = vs;
      }
      __syncthreads();
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float vs = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) vs += red[FS_R * NW + (32 * rb + t) * NW + w];
        rstd[rb] = rsqrtf(vs * inv_n + 1e-5f);
      }
      if (p.z_keep != nullptr) {   // whole bf16 rows, coalesced (the planes are complete: barrier above)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = rg + 8 * k;
          const int rb = r >> 5;
          if ((r & 31) < sub.ne(rb)) {
            const int64_t zoff = sub.b(rb) * p.z_bstride + (int64_t)(sub.p0(rb) + (r & 31)) * D;
            const bf16x4 zh = *reinterpret_cast<const bf16x4*>(S.hi + r * S.P + 4 * c4);
            if constexpr (TERMS == 3) {
              const bf16x4 zl = *reinterpret_cast<const bf16x4*>(S.lo + r * S.P + 4 * c4);
              f32x4 zf;
#pragma unroll
              for (int j = 0; j < 4; ++j) zf[j] = (float)zh[j] + (float)zl[j];
              reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.z_keep) + zoff)[c4] = zf;
            } else {
              reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.z_keep) + zoff)[c4] = zh;
            }
          }
        }
      }
      const f32x16 gav = fs_vec_block(p.gamma, p.n_out, wave, lane);
      const f32x16 bev = fs_vec_block(p.beta, p.n_out, wave, lane);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[rb][r] = (z[rb][r] - mean[rb]) * rstd[rb] * gav[r] + bev[r];
    }
    FSSTAMP(3)   // LayerNorm: statistics exchange (2 barriers), z_keep store, normalise
    fs_acc_to_tile<LDO>(z, mtile, wave, lane);
    if (tid < FS_R && more) put_idx(itab0 + (par ^ 1) * 4 * FS_R);
    __syncthreads();
    FSSTAMP(4)   // output tile + next tables + barrier
    // ---- outputs from the fp32 tile: receiver aggregation, whole-row stores
    if (p.agg != nullptr) {
      const int rb = wave / (NW / 2), fc = wave % (NW / 2);
      const FsSub sb = rb ? sub.s1 : sub.s0;
      float* aggb = p.agg + sb.b * p.agg_bstride;
      fs_segment_sums(mtile + 32 * rb * LDO + 64 * fc, LDO, sb, p.tl, lane, [&](int r, float acc) {
        const float sc = p.inv_deg ? p.inv_deg[r] : 1.0f;
        aggb[(int64_t)r * p.agg_ld + 64 * fc + lane] = acc * sc;
      });
    }
    if (p.y != nullptr) {
      if (p.vec_y) {
        const int nc4 = p.n_out >> 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = rg + 8 * k;
          const int rb = r >> 5;
          if ((r & 31) < sub.ne(rb) && c4 < nc4) {
            const int64_t row = itab[3 * FS_R + r];
            f32x4 o = *reinterpret_cast<const f32x4*>(mtile + r * LDO + 4 * c4);
            if (p.res.ptr != nullptr) o += rv[k];
            reinterpret_cast<f32x4*>(p.y + sub.b(rb) * p.y_bstride + row * p.y_ld)[c4] = o;
          }
        }
      } else {
        for (int idx = tid; idx < FS_R * p.n_out; idx += NT) {
          const int r = idx / p.n_out, cc = idx - r * p.n_out;
          const int rb = r >> 5;
          if ((r & 31) < sub.ne(rb)) {
            const int64_t row = itab[3 * FS_R + r];
            float o = mtile[r * LDO + cc];
            if (p.res.ptr != nullptr) o += p.res.ptr[sub.b(rb) * p.res.bstride + row * p.res.ld + cc];
            p.y[sub.b(rb) * p.y_bstride + row * p.y_ld + cc] = o;
          }
        }
      }
    }
    // (no barrier here: the next iteration writes the S planes -- free since every wave passed
    //  the barriers behind its GEMM -- and touches the output tile, the LayerNorm exchange
    //  and this tile's index table only behind its own first barrier)
    FSSTAMP(5)   // aggregation + row stores
  }
  if (stamp && tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_fs_stamps[k], fst[k]);
  }
}

template <int D, bool HAS_LN, int TERMS, bool IO16 = false>
static int launch_fs_tail_fwd(const FsTailFwdParams& p, hipStream_t s) {
  const size_t lds = FsPlanes<D, TERMS>::bytes + (size_t)FS_R * (D + 4) * sizeof(float) +
                     (size_t)2 * FS_R * (D / 32) * sizeof(float) + (size_t)8 * FS_R * sizeof(int);
  NLAM_REQUIRE(lds <= 160 * 1024, "fs_tail_fwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = fs_tail_fwd_kernel<D, HAS_LN, TERMS, IO16>;
  NLAM_BIG_LDS(kern, "fs_tail_fwd_kernel");
  const int64_t ntiles = (p.tl.ntiles * p.tl.B + 1) / 2;
  kern<<<fs_grid(ntiles), 2 * D, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("fs_tail_fwd_kernel");
  return 0;
}

int nlam_fs_tail_fwd_256(
    const int32_t* tiles, int64_t ntiles, int64_t rows, const int32_t* csr_rec,
    const int32_t* csr_rowptr,
    const float* a, int64_t a_bstride, int64_t a_ld, const int32_t* idx_a,
    const float* b, int64_t b_bstride, int64_t b_ld, const int32_t* idx_b,
    const float* c, int64_t c_bstride, int64_t c_ld, const int32_t* idx_c,
    const float* W2, int64_t ldW2, const float* b2, const float* gamma, const float* beta,
    int n_out, float* h_out, int64_t h_bstride, void* z_keep, int64_t z_bstride,
    float* y, int64_t y_bstride, int64_t y_ld, const int32_t* idx_y,
    const float* res, int64_t res_bstride, int64_t res_ld,
    float* agg, int64_t agg_bstride, int64_t agg_ld, const float* inv_deg,
    int64_t B, int io_bf16, void* stream) {
  constexpr int d = 256;
  NLAM_REQUIRE(!io_bf16 || (a_ld % 8 == 0 && a_bstride % 8 == 0 &&
                            (!b || (b_ld % 8 == 0 && b_bstride % 8 == 0)) &&
                            (!c || (c_ld % 8 == 0 && c_bstride % 8 == 0)) && h_bstride % 8 == 0),
               "nlam_tail_fwd: bf16 rows need pitches %% 8 == 0");
  NLAM_REQUIRE(z_keep == nullptr || (gamma != nullptr && (reinterpret_cast<uintptr_t>(z_keep) & 7u) == 0 &&
                                     z_bstride % 4 == 0),
               "nlam_tail_fwd: z_keep needs the LayerNorm form and 8-byte aligned rows");
  FS_TERMS("nlam_tail_fwd", io_bf16);
  NLAM_REQUIRE(terms_ == 1 || z_keep == nullptr || (nlam_aligned16(z_keep) && z_bstride % 4 == 0),
               "nlam_tail_fwd: fp32 z_keep rows must be 16-byte aligned");
  NLAM_REQUIRE(n_out >= 1 && n_out <= d, "nlam_tail_fwd: n_out %d out of range", n_out);
  NLAM_REQUIRE(gamma == nullptr || n_out == d, "nlam_tail_fwd: LayerNorm needs n_out == d");
  NLAM_REQUIRE(view_vec_ok(a, a_bstride, a_ld, d) && (!b || view_vec_ok(b, b_bstride, b_ld, d)) &&
                   (!c || view_vec_ok(c, c_bstride, c_ld, d)) && (!c || b),
               "nlam_tail_fwd: sources must be 16-byte aligned rows of width d");
  NLAM_REQUIRE(h_out == nullptr || (nlam_aligned16(h_out) && h_bstride % 4 == 0),
               "nlam_tail_fwd: h_out misaligned");
  NLAM_REQUIRE(agg == nullptr || (tiles != nullptr && csr_rec != nullptr && csr_rowptr != nullptr &&
                                  agg_ld >= n_out && n_out == d),
               "nlam_tail_fwd: aggregation needs edge tiles and n_out == d");
  FsTailFwdParams p;
  p.tl = FsTiling{tiles, ntiles, rows, csr_rec, csr_rowptr, (int)B};
  p.a = RowView{a, a_bstride, a_ld, d}; p.idx_a = idx_a;
  p.b = RowView{b, b_bstride, b_ld, d}; p.idx_b = idx_b;
  p.c = RowView{c, c_bstride, c_ld, d}; p.idx_c = idx_c;
  p.W2 = W2; p.ldW2 = ldW2; p.b2 = b2; p.gamma = gamma; p.beta = beta; p.n_out = n_out;
  p.h_out = h_out; p.h_bstride = h_bstride;
  p.z_keep = z_keep; p.z_bstride = z_bstride;
  p.y = y; p.y_bstride = y_bstride; p.y_ld = y_ld; p.idx_y = idx_y;
  p.res = RowView{res, res_bstride, res_ld, n_out};
  p.agg = agg; p.agg_bstride = agg_bstride; p.agg_ld = agg_ld; p.inv_deg = inv_deg;
  p.vec_y = (y != nullptr && view_vec_ok(y, y_bstride, y_ld, n_out) &&
             (res == nullptr || view_vec_ok(res, res_bstride, res_ld, n_out))) ? 1 : 0;
  p.stamp = fs_stamp_flag();
  hipStream_t s = (hipStream_t)stream;
  if (io_bf16)
    return gamma != nullptr ? launch_fs_tail_fwd<256, true, 1, true>(p, s)
                            : launch_fs_tail_fwd<256, false, 1, true>(p, s);
  if (terms_ == 3)
    return gamma != nullptr ? launch_fs_tail_fwd<256, true, 3>(p, s)
                            : launch_fs_tail_fwd<256, false, 3>(p, s);
  return gamma != nullptr ? launch_fs_tail_fwd<256, true, 1>(p, s)
                          : launch_fs_tail_fwd<256, false, 1>(p, s);
}


template <int LDO>
__device__ __forceinline__ void fs_acc_to_tile1(const f32x16& acc, float* __restrict__ otile, int rb,
                                                int wave, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 o = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    *reinterpret_cast<f32x4*>(otile + (32 * rb + t) * LDO + 32 * wave + 8 * q + 4 * h) = o;
  }
}
template <int LDO>
__device__ __forceinline__ void fs_tile_to_acc1(f32x16& acc, const float* __restrict__ otile, int rb,
                                                int wave, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(otile + (32 * rb + t) * LDO + 32 * wave + 8 * q + 4 * h);
    acc[4 * q] = o[0]; acc[4 * q + 1] = o[1]; acc[4 * q + 2] = o[2]; acc[4 * q + 3] = o[3];
  }
}
template <int D, int TERMS>
__device__ __forceinline__ void fs_acc_to_planes1(const f32x16& acc, const FsPlanes<D, TERMS>& X, int rb,
                                                  int wave, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    bf16x4 hi, lo;
    b3_split4(v, hi, lo);
    const int off = (32 * rb + t) * X.P + 32 * wave + 8 * q + 4 * h;
    *reinterpret_cast<bf16x4*>(X.hi + off) = hi;
    if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(X.lo + off) = lo;
  }
}

// ======================================================================== tail backward ===
// Slab per workgroup: [dgamma (D) | dbeta (D)]  (HAS_LN only).
struct FsTailBwdParams {
  FsTiling tl;
  const float* h; int64_t h_bstride;
  const void* z_keep; int64_t z_bstride;     // HAS_LN: the pre-LayerNorm rows kept by the forward: bf16 (TERMS 1) / fp32 (TERMS 3)
  RowView g1; const int32_t* idx_g1; const float* scale1;
  RowView g2; const int32_t* idx_g2;
  const float* W2; int64_t ldW2; const float* b2; const float* gamma; int n_out;
  float* gz_out; int64_t gz_bstride;
  float* gh; int64_t gh_bstride; int64_t gh_ld; const int32_t* idx_gh;
  float* gpr; int64_t gpr_bstride; int64_t gpr_ld;
  float* slab; int64_t slab_stride;
  int vec_g;
};

// IO16: h and gz_out are bf16 rows (16-byte lanes; z_keep always is)
template <int D, bool HAS_LN, int TERMS, bool IO16 = false>
__global__ __launch_bounds__(2 * D) void fs_tail_bwd_kernel(FsTailBwdParams q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 2 * D, NW = D / 32, LDO = D + 4, CPR = D / 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tile / batch address math goes to the scalar unit
  const int t = lane & 31, h = lane >> 5;
  FsPlanes<D, TERMS> S;      // kept z rows (bf16), later gz
  S.init(smem);
  __bf16* DS = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(smem) + FsPlanes<D, TERMS>::bytes);
  constexpr int PD = D + 4;  // silu'(h) as one bf16 plane (2^-9 relative: inside the bf16-mixed budget)
  static_assert(!(IO16 && TERMS == 3), "bf16 rows exist in the bf16 mode only");
  float* gtile = reinterpret_cast<float*>(reinterpret_cast<char*>(DS) +
                                          (TERMS == 1 ? (size_t)FS_R * PD * sizeof(__bf16) : 0));
  float* red = gtile + FS_R * LDO;                            // [2][FS_R][NW]
  int* itab0 = reinterpret_cast<int*>(red + 2 * FS_R * NW);   // [2][4][FS_R]: g1, g2, gh, scale
  FsW<D, TERMS> A2;
  // (TERMS == 3: hi + lo transposed images over the S planes and the g tile, contiguous from smem)
  fs_load_w_cols_lds<D, TERMS>(A2, q.W2, q.ldW2, 32 * wave, q.n_out, TERMS == 1 ? (void*)gtile : (void*)smem, tid);
  f32x16 dgam, dbet;
#pragma unroll
  for (int r = 0; r < 16; ++r) dgam[r] = dbet[r] = 0.f;
  const int c4 = tid % CPR, rg = tid / CPR;
  const int NO = (q.n_out + 31) & ~31;
  const int64_t nsub = q.tl.ntiles * q.tl.B;
  const int64_t ntiles = (nsub + 1) / 2;
  // double-buffered slot tables (see fs_tail_fwd_kernel): next tile's indices fetched early, its
  // row scales (a dependent load) once the indices have arrived
  int nidx[3];
  auto fetch_idx = [&](int64_t tt) {
    const FsSub2 sub = {fs_sub(q.tl, 2 * tt), fs_sub(q.tl, 2 * tt + 1)};
    const int pos = fs_slot_pos(sub, tid & (FS_R - 1));
    nidx[0] = q.idx_g1 ? q.idx_g1[pos] : pos;
    nidx[1] = (q.g2.ptr && q.idx_g2) ? q.idx_g2[pos] : pos;
    nidx[2] = q.idx_gh ? q.idx_gh[pos] : pos;
  };
  auto put_idx = [&](int* tab) {
#pragma unroll
    for (int k = 0; k < 3; ++k) tab[k * FS_R + tid] = nidx[k];
    reinterpret_cast<float*>(tab)[3 * FS_R + tid] = q.scale1 ? q.scale1[nidx[0]] : 1.0f;
  };
  int par = 0;
  if (tid < FS_R && (int64_t)blockIdx.x < ntiles) {
    fetch_idx(blockIdx.x);
    put_idx(itab0);
  }
  __syncthreads();
  for (int64_t tt = blockIdx.x; tt < ntiles; tt += gridDim.x, par ^= 1) {
    const FsSub2 sub = {fs_sub(q.tl, 2 * tt), fs_sub(q.tl, 2 * tt + 1)};
    const int* itab = itab0 + par * 4 * FS_R;
    const float* stab = reinterpret_cast<const float*>(itab + 3 * FS_R);
    const bool more = tt + gridDim.x < ntiles;
    if (tid < FS_R && more) fetch_idx(tt + gridDim.x);
    // ---- stage: silu'(h) as a bf16 plane, the kept z rows (already bf16) into the S planes,
    //      g = scale * g1[idx] + g2[idx] as fp32 rows
    {
      if constexpr (IO16) {
        const int c8 = tid % (D / 8), rh = tid / (D / 8);
        fs_bf16x8 vh[4], vz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = rh + 16 * k;
          const int pos = fs_slot_pos(sub, r);
          vh[k] = fs_ld8(q.h, sub.b(r >> 5) * q.h_bstride + (int64_t)pos * D, c8);
          if (HAS_LN)
            vz[k] = reinterpret_cast<const fs_bf16x8*>(reinterpret_cast<const __bf16*>(q.z_keep) +
                                                       sub.b(r >> 5) * q.z_bstride + (int64_t)pos * D)[c8];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = rh + 16 * k;
          const bool valid = (r & 31) < sub.ne(r >> 5);
          float x[8];
          fs_cvt8f(vh[k], x);
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = valid ? nlam_silu_grad(x[j]) : 0.f;
          fs_plane_put8(DS, r * PD + 8 * c8, fs_cvt8b(x));
          if (HAS_LN) fs_plane_put8(S.hi, r * S.P + 8 * c8, vz[k]);
        }
      } else {
      if constexpr (TERMS == 3) {
        // split-bf16 mode: silu'(h) is taken from the fp32 h rows where it is applied (no bf16
        // plane); the kept z rows are fp32 and go to the hi + lo planes
        if (HAS_LN) {
          f32x4 vz[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int r = rg + 8 * k;
            const int pos = fs_slot_pos(sub, r);
            vz[k] = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(q.z_keep) +
                                                   sub.b(r >> 5) * q.z_bstride + (int64_t)pos * D)[c4];
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int r = rg + 8 * k;
            bf16x4 hi, lo;
            b3_split4(vz[k], hi, lo);
            *reinterpret_cast<bf16x4*>(S.hi + r * S.P + 4 * c4) = hi;
            *reinterpret_cast<bf16x4*>(S.lo + r * S.P + 4 * c4) = lo;
          }
        }
      } else {
      f32x4 vh[8];
      bf16x4 vz[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = rg + 8 * k;
        const int pos = fs_slot_pos(sub, r);
        vh[k] = reinterpret_cast<const f32x4*>(q.h + sub.b(r >> 5) * q.h_bstride + (int64_t)pos * D)[c4];
        if (HAS_LN)
          vz[k] = reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(q.z_keep) +
                                                  sub.b(r >> 5) * q.z_bstride + (int64_t)pos * D)[c4];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = rg + 8 * k;
        const bool valid = (r & 31) < sub.ne(r >> 5);
        bf16x4 dh;
#pragma unroll
        for (int j = 0; j < 4; ++j) dh[j] = (__bf16)(valid ? nlam_silu_grad(vh[k][j]) : 0.f);
        *reinterpret_cast<bf16x4*>(DS + r * PD + 4 * c4) = dh;
        if (HAS_LN) *reinterpret_cast<bf16x4*>(S.hi + r * S.P + 4 * c4) = vz[k];
      }
      }
      }
      if (q.vec_g) {
#pragma unroll 1
        for (int k0 = 0; k0 < 8; k0 += 4) {
          f32x4 vg[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = rg + 8 * (k0 + k);
            const float* gb = q.g1.ptr + sub.b(r >> 5) * q.g1.bstride;
            vg[k] = reinterpret_cast<const f32x4*>(gb + (int64_t)itab[r] * q.g1.ld)[c4];
          }
          if (q.g2.ptr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int r = rg + 8 * (k0 + k);
              const float* gb = q.g2.ptr + sub.b(r >> 5) * q.g2.bstride;
              const f32x4 v2 = reinterpret_cast<const f32x4*>(gb + (int64_t)itab[FS_R + r] * q.g2.ld)[c4];
              vg[k] = vg[k] * stab[r] + v2;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) vg[k] *= stab[rg + 8 * (k0 + k)];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = rg + 8 * (k0 + k);
            f32x4 x = vg[k];
            if ((r & 31) >= sub.ne(r >> 5)) x = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(gtile + r * LDO + 4 * c4) = x;
          }
        }
      } else {
        // (scalar path = the narrow head: only its first 32 columns are ever read again)
        constexpr int CW = HAS_LN ? D : 32;
        for (int idx = tid; idx < FS_R * CW; idx += NT) {
          const int r = idx / CW, cc = idx - r * CW;
          const int rb = r >> 5;
          float v = 0.f;
          if ((r & 31) < sub.ne(rb) && cc < q.n_out) {
            v = q.g1.ptr[sub.b(rb) * q.g1.bstride + (int64_t)itab[r] * q.g1.ld + cc] * stab[r];
            if (q.g2.ptr)
              v += q.g2.ptr[sub.b(rb) * q.g2.bstride + (int64_t)itab[FS_R + r] * q.g2.ld + cc];
          }
          gtile[r * LDO + cc] = v;
        }
      }
    }
    __syncthreads();
    // Every wave reads and writes only its own 32 columns of the g tile, so g -> gamma*g -> gz
    // is done in place there, one row block at a time; likewise z -> gz in the S planes.
    if (HAS_LN) {
      // z: the forward's bf16 rows (its LayerNorm input), this wave's 32 features
      f32x16 z[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int zo = (32 * rb + t) * S.P + 32 * wave + 8 * qq + 4 * h;
          const bf16x4 zv = *reinterpret_cast<const bf16x4*>(S.hi + zo);
#pragma unroll
          for (int j = 0; j < 4; ++j) z[rb][4 * qq + j] = (float)zv[j];
          if constexpr (TERMS == 3) {
            const bf16x4 zl = *reinterpret_cast<const bf16x4*>(S.lo + zo);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[rb][4 * qq + j] += (float)zl[j];
          }
        }
      const float inv_n = 1.0f / (float)q.n_out;
      float mean[2], rstd[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float sm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += z[rb][r];
        sm = lane_xor32_sum(sm);
        if (h == 0) red[(32 * rb + t) * NW + wave] = sm;
      }
      __syncthreads();
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sm += red[(32 * rb + t) * NW + w];
        mean[rb] = sm * inv_n;
        float vs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dlt = z[rb][r] - mean[rb];
          vs += dlt * dlt;
        }
        vs = lane_xor32_sum(vs);
        if (h == 0) red[FS_R * NW + (32 * rb + t) * NW + wave] = vs;
      }
      __syncthreads();
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float vs = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) vs += red[FS_R * NW + (32 * rb + t) * NW + w];
        rstd[rb] = rsqrtf(vs * inv_n + 1e-5f);
      }
      __syncthreads();   // (every wave has read the statistics: red is reused for s1 / s2)
      {
        const f32x16 gav = fs_vec_block(q.gamma, q.n_out, wave, lane);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          f32x16 g;
          fs_tile_to_acc1<LDO>(g, gtile, rb, wave, lane);
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float xh = (z[rb][r] - mean[rb]) * rstd[rb];
            z[rb][r] = xh;
            dbet[r] += g[r];
            dgam[r] += g[r] * xh;
            const float gv = g[r] * gav[r];
            g[r] = gv;
            s1 += gv;
            s2 += gv * xh;
          }
          fs_acc_to_tile1<LDO>(g, gtile, rb, wave, lane);
          s1 = lane_xor32_sum(s1);
          s2 = lane_xor32_sum(s2);
          if (h == 0) {
            red[(32 * rb + t) * NW + wave] = s1;
            red[FS_R * NW + (32 * rb + t) * NW + wave] = s2;
          }
        }
      }
      __syncthreads();   // s1 / s2 complete
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          s1 += red[(32 * rb + t) * NW + w];
          s2 += red[FS_R * NW + (32 * rb + t) * NW + w];
        }
        const float m1 = s1 * inv_n, m2 = s2 * inv_n;
        f32x16 g;
        fs_tile_to_acc1<LDO>(g, gtile, rb, wave, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = rstd[rb] * (g[r] - m1 - z[rb][r] * m2);
        fs_acc_to_tile1<LDO>(g, gtile, rb, wave, lane);
        fs_acc_to_planes1<D, TERMS>(g, S, rb, wave, lane);
      }
    } else {
      // gz = g: the tile already holds it; planes for W2^T gz
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        f32x16 g;
        fs_tile_to_acc1<LDO>(g, gtile, rb, wave, lane);
        fs_acc_to_planes1<D, TERMS>(g, S, rb, wave, lane);
      }
    }
    __syncthreads();
    if constexpr (IO16 && HAS_LN) {   // (NO == D here: checked on the host; the narrow head's gz stays fp32)
      const int c8 = tid % (D / 8), rh = tid / (D / 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = rh + 16 * k;
        const int rb = r >> 5;
        if ((r & 31) < sub.ne(rb)) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(gtile + r * LDO + 8 * c8);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(gtile + r * LDO + 8 * c8 + 4);
          const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fs_st8(q.gz_out, sub.b(rb) * q.gz_bstride + (int64_t)(sub.p0(rb) + (r & 31)) * D, c8, fs_cvt8b(x));
        }
      }
    } else {
      const int nc4 = NO >> 2;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = rg + 8 * k;
        const int rb = r >> 5;
        if ((r & 31) < sub.ne(rb) && c4 < nc4)
          reinterpret_cast<f32x4*>(q.gz_out + sub.b(rb) * q.gz_bstride +
                                   (int64_t)(sub.p0(rb) + (r & 31)) * NO)[c4] =
              *reinterpret_cast<const f32x4*>(gtile + r * LDO + 4 * c4);
      }
    }
    // gh = (W2^T gz) * silu'(h)
    f32x16 gh[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) gh[rb][r] = 0.f;
    // (no LayerNorm = the narrow head, n_out <= 32: gz has two non-zero 16-column steps, not D / 16)
    if constexpr (HAS_LN) fs_gemm<D, TERMS>(gh, A2, S, lane);
    else fs_gemm<D, TERMS, 2>(gh, A2, S, lane);
    if constexpr (TERMS == 3) {
      f32x4 hv[2][4];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int pos = fs_slot_pos(sub, 32 * rb + t);
        const float* hb = q.h + sub.b(rb) * q.h_bstride + (int64_t)pos * D + 32 * wave + 4 * h;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) hv[rb][qq] = *reinterpret_cast<const f32x4*>(hb + 8 * qq);
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const bool valid = t < sub.ne(rb);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
          for (int j = 0; j < 4; ++j) gh[rb][4 * qq + j] *= valid ? nlam_silu_grad(hv[rb][qq][j]) : 0.f;
      }
    } else {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const bf16x4 dv = *reinterpret_cast<const bf16x4*>(DS + (32 * rb + t) * PD + 32 * wave + 8 * qq + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) gh[rb][4 * qq + j] *= (float)dv[j];
      }
    }
    if (tid < FS_R && more) put_idx(itab0 + (par ^ 1) * 4 * FS_R);
    __syncthreads();   // the gz rows have been read from the tile
    fs_acc_to_tile<LDO>(gh, gtile, wave, lane);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = rg + 8 * k;
      const int rb = r >> 5;
      if ((r & 31) < sub.ne(rb))
        reinterpret_cast<f32x4*>(q.gh + sub.b(rb) * q.gh_bstride + (int64_t)itab[2 * FS_R + r] * q.gh_ld)[c4] =
            *reinterpret_cast<const f32x4*>(gtile + r * LDO + 4 * c4);
    }
    if (q.gpr != nullptr) {
      const int rb = wave / (NW / 2), fc = wave % (NW / 2);
      const FsSub sb = rb ? sub.s1 : sub.s0;
      float* gb = q.gpr + sb.b * q.gpr_bstride;
      fs_segment_sums(gtile + 32 * rb * LDO + 64 * fc, LDO, sb, q.tl, lane, [&](int r, float acc) {
        gb[(int64_t)r * q.gpr_ld + 64 * fc + lane] = acc;
      });
    }
    __syncthreads();
  }
  if (HAS_LN) {
    // per-lane (row slot) partials -> per-feature sums over the 32 slots of each lane half
    float* slab = q.slab + (int64_t)blockIdx.x * q.slab_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = dgam[r], bsum = dbet[r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        bsum += __shfl_xor(bsum, o, 64);
      }
      if (t == 0) {
        const int f = 32 * wave + 8 * (r >> 2) + 4 * h + (r & 3);
        slab[f] = a;
        slab[D + f] = bsum;
      }
    }
  }
}

template <int D, bool HAS_LN, int TERMS, bool IO16 = false>
static int launch_fs_tail_bwd(const FsTailBwdParams& q, hipStream_t s, unsigned grid) {
  const size_t lds = FsPlanes<D, TERMS>::bytes + (TERMS == 1 ? (size_t)FS_R * (D + 4) * sizeof(__bf16) : 0) +
                     (size_t)FS_R * (D + 4) * sizeof(float) +
                     (size_t)2 * FS_R * (D / 32) * sizeof(float) + (size_t)8 * FS_R * sizeof(int);
  NLAM_REQUIRE(lds <= 160 * 1024, "fs_tail_bwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = fs_tail_bwd_kernel<D, HAS_LN, TERMS, IO16>;
  NLAM_BIG_LDS(kern, "fs_tail_bwd_kernel");
  kern<<<grid, 2 * D, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("fs_tail_bwd_kernel");
  return 0;
}

// grid = the slab count the host sized its buffer for (nlam_bwd_grid(B * ntiles))
int nlam_fs_tail_bwd_256(
    const int32_t* tiles, int64_t ntiles, int64_t rows, const int32_t* csr_rec,
    const int32_t* csr_rowptr, const float* h, int64_t h_bstride,
    const void* z_keep, int64_t z_bstride,
    const float* g1, int64_t g1_bstride, int64_t g1_ld, const int32_t* idx_g1, const float* scale1,
    const float* g2, int64_t g2_bstride, int64_t g2_ld, const int32_t* idx_g2,
    const float* W2, int64_t ldW2, const float* b2, const float* gamma, int n_out,
    float* gz_out, int64_t gz_bstride,
    float* gh, int64_t gh_bstride, int64_t gh_ld, const int32_t* idx_gh,
    float* gpr, int64_t gpr_bstride, int64_t gpr_ld,
    float* slab, int64_t slab_stride, int64_t B, unsigned grid, int io_bf16, void* stream) {
  constexpr int d = 256;
  NLAM_REQUIRE(!io_bf16 || (h_bstride % 8 == 0 && (gamma == nullptr || (n_out == d && gz_bstride % 8 == 0))),
               "nlam_tail_bwd: bf16 h rows (and, with LayerNorm, n_out == d bf16 gz rows): pitches %% 8 == 0");
  NLAM_REQUIRE(gamma == nullptr || (z_keep != nullptr && (reinterpret_cast<uintptr_t>(z_keep) & 7u) == 0 &&
                                    z_bstride % 4 == 0),
               "nlam_tail_bwd: hidden 256 with LayerNorm needs the z_keep rows of nlam_tail_fwd");
  FS_TERMS("nlam_tail_bwd", io_bf16);
  NLAM_REQUIRE(terms_ == 1 || z_keep == nullptr || (nlam_aligned16(z_keep) && z_bstride % 4 == 0),
               "nlam_tail_bwd: fp32 z_keep rows must be 16-byte aligned");
  NLAM_REQUIRE(gamma == nullptr || n_out == d, "nlam_tail_bwd: LayerNorm needs n_out == d");
  NLAM_REQUIRE(h != nullptr && nlam_aligned16(h) && h_bstride % 4 == 0, "nlam_tail_bwd: bad h");
  NLAM_REQUIRE(g1 != nullptr && gz_out != nullptr && nlam_aligned16(gz_out) && gz_bstride % 4 == 0,
               "nlam_tail_bwd: g1 / gz_out missing or misaligned");
  NLAM_REQUIRE(gh != nullptr && view_vec_ok(gh, gh_bstride, gh_ld, d), "nlam_tail_bwd: bad gh view");
  NLAM_REQUIRE(gamma == nullptr || (slab != nullptr && slab_stride >= 2 * d), "nlam_tail_bwd: slab too small");
  FsTailBwdParams q;
  q.tl = FsTiling{tiles, ntiles, rows, csr_rec, csr_rowptr, (int)B};
  q.h = h; q.h_bstride = h_bstride;
  q.z_keep = z_keep; q.z_bstride = z_bstride;
  q.g1 = RowView{g1, g1_bstride, g1_ld, n_out}; q.idx_g1 = idx_g1; q.scale1 = scale1;
  q.g2 = RowView{g2, g2_bstride, g2_ld, n_out}; q.idx_g2 = idx_g2;
  q.W2 = W2; q.ldW2 = ldW2; q.b2 = b2; q.gamma = gamma; q.n_out = n_out;
  q.gz_out = gz_out; q.gz_bstride = gz_bstride;
  q.gh = gh; q.gh_bstride = gh_bstride; q.gh_ld = gh_ld; q.idx_gh = idx_gh;
  q.gpr = gpr; q.gpr_bstride = gpr_bstride; q.gpr_ld = gpr_ld;
  q.slab = slab; q.slab_stride = slab_stride;
  q.vec_g = (n_out == d && view_vec_ok(g1, g1_bstride, g1_ld, n_out) &&
             (g2 == nullptr || view_vec_ok(g2, g2_bstride, g2_ld, n_out))) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (io_bf16)
    return gamma != nullptr ? launch_fs_tail_bwd<256, true, 1, true>(q, s, grid)
                            : launch_fs_tail_bwd<256, false, 1, true>(q, s, grid);
  if (terms_ == 3)
    return gamma != nullptr ? launch_fs_tail_bwd<256, true, 3>(q, s, grid)
                            : launch_fs_tail_bwd<256, false, 3>(q, s, grid);
  return gamma != nullptr ? launch_fs_tail_bwd<256, true, 1>(q, s, grid)
                          : launch_fs_tail_bwd<256, false, 1>(q, s, grid);
}

// ==================================================================== weight gradients ===
// dW (ng x nx) = sum_rows G[r]^T (x) f(X[r]),  db = colsum(G);  f = silu when silu_x (X is then
// the kept pre-activation h).  Every wave owns a 32-feature slice of dW for the whole launch:
//   GW == 256: wave w holds dW[32 w .. 32 w + 31][0 .. 32 NXB)   (8 NXB accumulator registers x 16)
//   GW == 32 : wave w holds dW[0 .. 31][32 w .. 32 w + 31]        (nx = 256; the n_out <= 32 heads)
// Slab per workgroup as nlam_wide_outer: [dW (ng x 32 NXB) | db (ng)].
struct FsOuterParams {
  RowView g;        // (B, rows, GW)
  RowView x;        // (B, rows, nx); batch-invariant x (bstride 0) allowed
  float* slab; int64_t slab_stride;
  FsTiling tl;      // row mode
  int silu_x;
  int x_vec;
  int io_bf16;      // bit 0: g rows are bf16, bit 1: x rows are bf16 (256-wide operands only)
};

template <int GW, int NXB, int TERMS>
__device__ __forceinline__ void fs_outer_body(const FsOuterParams& q, const int bid, const int gdim,
                                              float* smem) {
  constexpr int NX = 32 * NXB, NT = 512;
  constexpr int NJ = GW == 256 ? NXB : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tile / batch address math goes to the scalar unit
  FsPlanes<GW, TERMS, 16> G;
  G.init(smem);
  FsPlanes<NX, TERMS, 16> X;
  X.init(reinterpret_cast<char*>(smem) + FsPlanes<GW, TERMS, 16>::bytes);
  f32x16 dW[1][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dW[0][j][r] = 0.f;
  float db = 0.f;
  // staging maps: G rows (GW / 4 chunks per row), X rows (NX / 4 chunks per row, wide form)
  constexpr int GCPR = GW / 4, GRPP = NT / GCPR, GNV = FS_R / GRPP;   // 256: 64, 8, 8;  32: 8, 64, 1
  constexpr int XCPR = NX / 4, XRPP = NT / XCPR, XNV = (FS_R + XRPP - 1) / XRPP;
  const int gc4 = tid % GCPR, grg = tid / GCPR;
  const int xc4 = tid % XCPR, xrg = tid / XCPR;
  const int64_t nsub = q.tl.ntiles * q.tl.B;
  const int64_t ntiles = (nsub + 1) / 2;
  const bool silu_x = q.silu_x != 0;
  const bool x_vec = NXB >= 8 ? true : (q.x_vec != 0);
  const bool g16 = GW == 256 && (q.io_bf16 & 1), x16 = NXB == 8 && (q.io_bf16 & 2);
  const int c8 = tid % 32, rh = tid / 32;     // 16-byte-lane map of 256-wide bf16 rows
  f32x4 vg[GNV], vx[XNV];
  auto issue = [&](int64_t tt) {
    const FsSub2 sub = {fs_sub(q.tl, 2 * tt), fs_sub(q.tl, 2 * tt + 1)};
    if (g16) {   // bf16 rows: 16-byte lanes, 16 rows per pass; the 16 bytes ride in vg[k] as they are
      if constexpr (GW == 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = rh + 16 * k;
          const int pos = fs_slot_pos(sub, r);
          fs_bf16x8 v = fs_ld8(q.g.ptr, sub.b(r >> 5) * q.g.bstride + (int64_t)pos * q.g.ld, c8);
          if ((r & 31) >= sub.ne(r >> 5)) v = fs_bf16x8{};
          vg[k] = __builtin_bit_cast(f32x4, v);
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < GNV; ++k) {
      const int r = grg + GRPP * k;
      const int pos = fs_slot_pos(sub, r);
      vg[k] = reinterpret_cast<const f32x4*>(q.g.ptr + sub.b(r >> 5) * q.g.bstride + (int64_t)pos * q.g.ld)[gc4];
      if ((r & 31) >= sub.ne(r >> 5)) vg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    }
    if (x16) {
      if constexpr (NXB == 8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = rh + 16 * k;
          const int pos = fs_slot_pos(sub, r);
          fs_bf16x8 v = fs_ld8(q.x.ptr, sub.b(r >> 5) * q.x.bstride + (int64_t)pos * q.x.ld, c8);
          if ((r & 31) >= sub.ne(r >> 5)) v = fs_bf16x8{};
          vx[k] = __builtin_bit_cast(f32x4, v);
        }
      }
    } else if (x_vec) {
#pragma unroll
      for (int k = 0; k < XNV; ++k) {
        const int r = (xrg + XRPP * k) & (FS_R - 1);
        const int pos = fs_slot_pos(sub, r);
        const int cc = (4 * xc4 < q.x.width) ? xc4 : 0;
        vx[k] = reinterpret_cast<const f32x4*>(q.x.ptr + sub.b(r >> 5) * q.x.bstride + (int64_t)pos * q.x.ld)[cc];
        if ((r & 31) >= sub.ne(r >> 5) || 4 * xc4 >= q.x.width) vx[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto put = [&](int64_t tt) {
    if (g16) {
      if constexpr (GW == 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          fs_plane_put8(G.hi, (rh + 16 * k) * G.P + 8 * c8, __builtin_bit_cast(fs_bf16x8, vg[k]));
      }
    } else {
#pragma unroll
    for (int k = 0; k < GNV; ++k) {
      const int r = grg + GRPP * k;
      bf16x4 hi, lo;
      b3_split4(vg[k], hi, lo);
      *reinterpret_cast<bf16x4*>(G.hi + r * G.P + 4 * gc4) = hi;
      if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(G.lo + r * G.P + 4 * gc4) = lo;
    }
    }
    if (x16) {
      if constexpr (NXB == 8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          fs_bf16x8 v = __builtin_bit_cast(fs_bf16x8, vx[k]);
          if (silu_x) {
            float x[8];
            fs_cvt8f(v, x);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = nlam_silu(x[j]);
            v = fs_cvt8b(x);
          }
          fs_plane_put8(X.hi, (rh + 16 * k) * X.P + 8 * c8, v);
        }
      }
    } else if (x_vec) {
#pragma unroll
      for (int k = 0; k < XNV; ++k) {
        const int r = xrg + XRPP * k;
        if (r < FS_R) {
          f32x4 v = vx[k];
          if (silu_x) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = nlam_silu(v[j]);
          }
          bf16x4 hi, lo;
          b3_split4(v, hi, lo);
          *reinterpret_cast<bf16x4*>(X.hi + r * X.P + 4 * xc4) = hi;
          if constexpr (TERMS == 3) *reinterpret_cast<bf16x4*>(X.lo + r * X.P + 4 * xc4) = lo;
        }
      }
    } else {
      // narrow or unaligned x rows (static features of the embedders): scalar staging
      const FsSub2 sub = {fs_sub(q.tl, 2 * tt), fs_sub(q.tl, 2 * tt + 1)};
      for (int idx = tid; idx < FS_R * NX; idx += NT) {
        const int r = idx / NX, cc = idx - r * NX;
        const int rb = r >> 5;
        float v = 0.f;
        if ((r & 31) < sub.ne(rb) && cc < q.x.width)
          v = q.x.ptr[sub.b(rb) * q.x.bstride + (int64_t)(sub.p0(rb) + (r & 31)) * q.x.ld + cc];
        if (silu_x) v = nlam_silu(v);
        const __bf16 hi = (__bf16)v;
        X.hi[r * X.P + cc] = hi;
        if constexpr (TERMS == 3) X.lo[r * X.P + cc] = (__bf16)(v - (float)hi);
      }
    }
  };
  int64_t tt = bid;
  if (tt < ntiles) issue(tt);
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  for (; tt < ntiles; tt += gdim) {
    put(tt);
    __syncthreads();
    if (tt + gdim < ntiles) issue(tt + gdim);
    const bool active = GW == 256 || true;
    if (active) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const B3Tile Gt = {G.hi + 32 * rb * G.P, G.lo + 32 * rb * G.P, G.P};
        const B3Tile Xt = {X.hi + 32 * rb * X.P, X.lo + 32 * rb * X.P, X.P};
        if constexpr (GW == 256) {
          outer_accum_b3<1, NJ, TERMS>(dW, Gt, 32 * wave, Xt, 0, lane);
        } else {
          outer_accum_b3<1, 1, TERMS>(dW, Gt, 0, Xt, 32 * wave, lane);
        }
        if (GW == 256 || wave == 0) {
          const int gcol = GW == 256 ? 32 * wave : 0;
          f32x16 c;
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const bf16x8 bh = b3_tr_frag_rows(Gt.hi, Gt.pitch, 16 * u, gcol, lane);
            c = B3_MFMA(ones, bh, c);
            if constexpr (TERMS == 3) {
              const bf16x8 bl = b3_tr_frag_rows(Gt.lo, Gt.pitch, 16 * u, gcol, lane);
              c = B3_MFMA(ones, bl, c);
            }
          }
          db += c[0];
        }
      }
    }
    __syncthreads();
  }
  float* slab = q.slab + (int64_t)bid * q.slab_stride;
  const int h = lane >> 5, j = lane & 31;
  if constexpr (GW == 256) {
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r >> 2) + 4 * h + (r & 3);
        slab[(int64_t)(32 * wave + i) * NX + 32 * jb + j] = dW[0][jb][r];
      }
    if (lane < 32) slab[(int64_t)GW * NX + 32 * wave + lane] = db;
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 8 * (r >> 2) + 4 * h + (r & 3);
      slab[(int64_t)i * NX + 32 * wave + j] = dW[0][0][r];
    }
    if (wave == 0 && lane < 32) slab[(int64_t)GW * NX + lane] = db;
  }
}

template <int GW, int NXB, int TERMS>
__global__ __launch_bounds__(512) void fs_outer_kernel(WideMulti<FsOuterParams, NLAM_WIDE_MAXP_OUTER> m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = wide_multi_find(m, blockIdx.x);
  fs_outer_body<GW, NXB, TERMS>(m.p[k], blockIdx.x - m.first[k], m.first[k + 1] - m.first[k], smem);
}

// m.first[] = the slab counts the caller sized its buffers for (one workgroup per slab)
template <int GW, int NXB, int TERMS>
static int launch_fs_outer(const WideMulti<FsOuterParams, NLAM_WIDE_MAXP_OUTER>& m, hipStream_t s) {
  const size_t lds = FsPlanes<GW, TERMS, 16>::bytes + FsPlanes<32 * NXB, TERMS, 16>::bytes;
  auto kern = fs_outer_kernel<GW, NXB, TERMS>;
  NLAM_BIG_LDS(kern, "fs_outer_kernel");
  kern<<<(unsigned)m.first[m.n], 512, lds, s>>>(m);
  NLAM_CHECK_LAUNCH("fs_outer_kernel");
  return 0;
}
template <int GW, int NXB, int TERMS>
static int launch_fs_outer(const FsOuterParams& q, hipStream_t s, unsigned grid) {
  WideMulti<FsOuterParams, NLAM_WIDE_MAXP_OUTER> m;
  m.n = 1;
  m.p[0] = q;
  m.first[0] = 0;
  for (int k = 0; k < NLAM_WIDE_MAXP_OUTER; ++k) m.first[k + 1] = (int)grid;
  return launch_fs_outer<GW, NXB, TERMS>(m, s);
}

// (ng, nx) in {(256, 256), (32, 256), (256, <= 64)}; grid = the slab count of the caller
static int fs_outer_fill(FsOuterParams& q, const float* g, int64_t g_bstride, int64_t g_ld, int ng,
                         const float* x, int64_t x_bstride, int64_t x_ld, int nx, int silu_x,
                         float* slab, int64_t slab_stride, int64_t B, int64_t rows, int io_bf16) {
  q.io_bf16 = io_bf16;
  NLAM_REQUIRE(!(io_bf16 & 1) || (ng == 256 && g_ld % 8 == 0 && g_bstride % 8 == 0),
               "nlam_wide_outer: bf16 g rows must be 256 wide, pitches %% 8 == 0");
  NLAM_REQUIRE(!(io_bf16 & 2) || (nx == 256 && x_ld % 8 == 0 && x_bstride % 8 == 0),
               "nlam_wide_outer: bf16 x rows must be 256 wide, pitches %% 8 == 0");
  NLAM_REQUIRE((ng == 256 || ng == 32) && nx >= 1 && (nx <= 64 || nx == 256) && (ng == 256 || nx == 256),
               "nlam_wide_outer: shape %d x %d unsupported", ng, nx);
  NLAM_REQUIRE(view_vec_ok(g, g_bstride, g_ld, ng) && x != nullptr && x_ld >= nx,
               "nlam_wide_outer: g rows must be 16-byte aligned with pitch %% 4 == 0");
  const int nxp = (nx + 31) & ~31;
  NLAM_REQUIRE(slab != nullptr && slab_stride >= (int64_t)ng * nxp + ng, "nlam_wide_outer: slab too small");
  q.g = RowView{g, g_bstride, g_ld, ng};
  q.x = RowView{x, x_bstride, x_ld, nx};
  q.slab = slab; q.slab_stride = slab_stride;
  q.tl = FsTiling{nullptr, (rows + NLAM_TILE - 1) / NLAM_TILE, rows, nullptr, nullptr, (int)B};
  q.silu_x = silu_x;
  q.x_vec = view_vec_ok(x, x_bstride, x_ld, nx) ? 1 : 0;
  NLAM_REQUIRE(nx < 256 || q.x_vec, "nlam_wide_outer: 256-wide x rows must be 16-byte aligned");
  return 0;
}

int nlam_fs_outer_256(const float* g, int64_t g_bstride, int64_t g_ld, int ng,
                      const float* x, int64_t x_bstride, int64_t x_ld, int nx, int silu_x,
                      float* slab, int64_t slab_stride, int64_t B, int64_t rows, unsigned grid,
                      int io_bf16, void* stream) {
  FS_TERMS("nlam_wide_outer", io_bf16);
  FsOuterParams q;
  if (fs_outer_fill(q, g, g_bstride, g_ld, ng, x, x_bstride, x_ld, nx, silu_x, slab, slab_stride, B, rows,
                    io_bf16))
    return 1;
  hipStream_t s = (hipStream_t)stream;
  if (terms_ == 3) {
    if (ng == 32) return launch_fs_outer<32, 8, 3>(q, s, grid);
    if (nx <= 32) return launch_fs_outer<256, 1, 3>(q, s, grid);
    if (nx <= 64) return launch_fs_outer<256, 2, 3>(q, s, grid);
    return launch_fs_outer<256, 8, 3>(q, s, grid);
  }
  if (ng == 32) return launch_fs_outer<32, 8, 1>(q, s, grid);
  if (nx <= 32) return launch_fs_outer<256, 1, 1>(q, s, grid);
  if (nx <= 64) return launch_fs_outer<256, 2, 1>(q, s, grid);
  return launch_fs_outer<256, 8, 1>(q, s, grid);
}

// all problems 256 x nx[k] with nx[k] = 256 for all k, or nx[k] <= 64 for all k (nx == NULL: 256);
// grid[k] = slab count of problem k
int nlam_fs_outer_multi_256(int n, const float* const* g, const int64_t* g_bstride,
                            const int64_t* g_ld, const float* const* x, const int64_t* x_bstride,
                            const int64_t* x_ld, const int32_t* silu_x, float* const* slab,
                            const int64_t* slab_stride, const int64_t* B, const int64_t* rows,
                            const unsigned* grid, const int32_t* io_bf16, void* stream,
                            const int32_t* nx) {
  int any16 = 0;
  for (int k = 0; k < n && io_bf16; ++k) any16 |= io_bf16[k];
  FS_TERMS("nlam_wide_outer_multi", any16);
  WideMulti<FsOuterParams, NLAM_WIDE_MAXP_OUTER> m;
  m.n = 0;
  m.first[0] = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    if (fs_outer_fill(m.p[m.n], g[k], g_bstride[k], g_ld[k], 256, x[k], x_bstride[k], x_ld[k],
                      nx ? nx[k] : 256, silu_x[k], slab[k], slab_stride[k], B[k], rows[k],
                      io_bf16 ? io_bf16[k] : 0))
      return 1;
    m.first[m.n + 1] = m.first[m.n] + (int)grid[k];
    ++m.n;
  }
  if (m.n == 0) return 0;
  for (int k = m.n; k < NLAM_WIDE_MAXP_OUTER; ++k) m.first[k + 1] = m.first[m.n];
  int nxmax = 0;
  for (int k = 0; k < n; ++k) nxmax = nx ? (nx[k] > nxmax ? nx[k] : nxmax) : 256;
  hipStream_t s = (hipStream_t)stream;
  if (nxmax <= 32) return terms_ == 3 ? launch_fs_outer<256, 1, 3>(m, s) : launch_fs_outer<256, 1, 1>(m, s);
  if (nxmax <= 64) return terms_ == 3 ? launch_fs_outer<256, 2, 3>(m, s) : launch_fs_outer<256, 2, 1>(m, s);
  for (int k = 0; k < n; ++k)
    NLAM_REQUIRE(!nx || nx[k] == 256, "nlam_wide_outer_multi: x widths of one launch: all 256, or all <= 64");
  return terms_ == 3 ? launch_fs_outer<256, 8, 3>(m, s) : launch_fs_outer<256, 8, 1>(m, s);
}
