// Wave-level building blocks of the fused gfx950 kernels.
//
// Layout convention ("row-on-lane"): a wavefront owns a tile of 32 rows (edges
// or nodes).  Lane l = (t, h) with t = l & 31 the row and h = l >> 5.  A row's
// D features live in accumulator blocks acc[nb][r] (nb = feature block of 32,
// r in [0,16)):   feature f = 32 nb + 8 (r >> 2) + 4 h + (r & 3).
// This is exactly the C/D map of v_mfma_f32_32x32x2_f32 when the GEMM is
// computed transposed (D^T[f][t] = W[f][:] . X[t][:]), so
//   * the result of one GEMM is directly the B operand of the next one
//     (no LDS round trip, no lane movement), and
//   * LayerNorm over the features is a per-lane sum plus one xor-32 shuffle.
// The k index of a K=2 MFMA step is permuted consistently on both operands:
// step (c, j) multiplies k_a = 8c + j (lanes h = 0) and k_b = 8c + 4 + j
// (lanes h = 1); any pairing is valid as long as A and B agree.
//
// LDS row tiles are [32][LD] floats with LD = K + 4 (K a multiple of 8): the
// pad makes the float4 operand reads (16-lane groups, distinct rows) and the
// float4 tile writes conflict-free.
#pragma once
#include "nlam_common.h"

#define NLAM_TILE 32

struct RowView {        // (B, rows, width) fp32 with unit column stride
  const float* ptr;
  int64_t bstride;      // elements between batch items (0 = batch-invariant)
  int64_t ld;           // elements between rows
  int width;            // columns taken from this view
};

static inline bool view_vec_ok(const float* ptr, int64_t bstride, int64_t ld, int width) {
  return ptr != nullptr && nlam_aligned16(ptr) && (bstride % 4 == 0) && (ld % 4 == 0) &&
         (width % 4 == 0) && width >= 4 && width <= 256;
}

static inline unsigned persistent_grid(int64_t ntiles, size_t lds_bytes) {
  // 256 CUs; as many workgroups per CU as the LDS footprint admits (<= 2)
  int per_cu = lds_bytes * 2 <= 160 * 1024 ? 2 : 1;
  int64_t g = (ntiles + 3) / 4;
  const int64_t cap = 256 * per_cu;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}


// Orders this wavefront's LDS traffic: tile data written by some lanes is read by
// other lanes of the SAME wave (DS ops of one wave execute in order; the fence
// keeps the compiler from reordering across it).  No workgroup barrier needed.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- weights global -> LDS ---------------------------------------------------
// W is (n_out x k_in) row-major in global memory; the LDS image is
// [n_pad][k_pad + 4] with zero fill (n_pad multiple of 32, k_pad multiple of 8).
__device__ __forceinline__ void load_weight_lds(float* __restrict__ Ws, const float* __restrict__ W,
                                                int64_t ldW, int n_out, int k_in, int n_pad,
                                                int k_pad, int tid, int nthreads) {
  const int ld = k_pad + 4;
  const bool vec = (k_in % 4 == 0) && (ldW % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0);
  if (vec) {
    // float4 chunks, 8 loads in flight per thread (this prologue dominates the
    // small mesh-sized launches, so it must not be a load->store chain)
    const int cpr = k_pad >> 2;              // chunks per row
    const int total = n_pad * cpr;
    for (int base = 0; base < total; base += 8 * nthreads) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * nthreads + tid;
        const int i = idx / cpr, c = idx - i * cpr;
        const bool ok = idx < total && i < n_out && 4 * c < k_in;
        v[u] = ok ? *reinterpret_cast<const f32x4*>(W + (int64_t)i * ldW + 4 * c)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * nthreads + tid;
        if (idx < total) {
          const int i = idx / cpr, c = idx - i * cpr;
          *reinterpret_cast<f32x4*>(Ws + i * ld + 4 * c) = v[u];
        }
      }
    }
  } else {
    for (int idx = tid; idx < n_pad * k_pad; idx += nthreads) {
      const int i = idx / k_pad, k = idx - i * k_pad;
      Ws[i * ld + k] = (i < n_out && k < k_in) ? W[(int64_t)i * ldW + k] : 0.f;
    }
  }
}
// per-feature vector -> LDS, zero padded to n_pad
__device__ __forceinline__ void load_vec_lds(float* __restrict__ vs, const float* __restrict__ v,
                                             int n, int n_pad, int tid, int nthreads) {
  for (int i = tid; i < n_pad; i += nthreads) vs[i] = (v != nullptr && i < n) ? v[i] : 0.f;
}

// Several per-feature vectors (n_pad <= nthreads entries each) with their global loads issued
// together -- and BEFORE the weight image's loads when vecs_issue() is called first: the prologue
// of a launch is then one global round trip instead of one per vector (it is a fixed cost of
// every launch, and most launches of the hierarchical models are small).
template <int NV>
struct VecLoads {
  float v[NV];
};
template <int NV>
__device__ __forceinline__ void vecs_issue(VecLoads<NV>& l, const float* const (&src)[NV], int n,
                                           int tid) {
#pragma unroll
  for (int j = 0; j < NV; ++j) l.v[j] = (src[j] != nullptr && tid < n) ? src[j][tid] : 0.f;
}
template <int NV>
__device__ __forceinline__ void vecs_commit(const VecLoads<NV>& l, float* const (&dst)[NV], int n_pad,
                                            int tid) {
  if (tid < n_pad) {
#pragma unroll
    for (int j = 0; j < NV; ++j) dst[j][tid] = l.v[j];
  }
}

// ---- row staging: global rows -> LDS tile ------------------------------------
// Tile row t (< nrows) comes from src + row_off(t); `width` floats are copied to
// columns [col0, col0 + width); vectorised when width % 4 == 0 and the source
// rows are 16-byte aligned (VEC = true), scalar otherwise.  With ADD the values
// are added to what the tile already holds.  Rows >= nrows are zero-filled
// (ADD = false) so that padded slots contribute nothing downstream.
template <bool VEC, bool ADD, typename RowPtr>
__device__ __forceinline__ void stage_rows(float* __restrict__ tile, int ld, int col0, int width,
                                           int nrows, int lane, RowPtr row_ptr) {
  if (VEC) {
    const int lpr = width >> 2;                 // lanes per row (float4 each)
    const int rpi = 64 / lpr;                   // rows per wave-instruction
    const int sub = lane / lpr, c4 = lane - sub * lpr;
    // row_ptr may shuffle indices across lanes (gathers): it is evaluated by all
    // lanes, outside the divergent parts (lanes 0..31 hold the slot indices and
    // are never masked here because rpi * lpr > 32).
    for (int k = 0; k * rpi < NLAM_TILE; ++k) {
      const int t = sub + k * rpi;
      const float* src = row_ptr(t < NLAM_TILE ? t : 0);
      if (sub < rpi && t < NLAM_TILE) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < nrows) v = reinterpret_cast<const f32x4*>(src)[c4];
        f32x4* dst = reinterpret_cast<f32x4*>(tile + t * ld + col0) + c4;
        if (ADD) v += *dst;
        *dst = v;
      }
    }
  } else {
    for (int idx = lane; idx < NLAM_TILE * width; idx += 64) {
      const int t = idx / width, c = idx - t * width;
      float v = (t < nrows) ? row_ptr(t)[c] : 0.f;
      float* dst = tile + t * ld + col0 + c;
      if (ADD) v += *dst;
      *dst = v;
    }
  }
}

// ---- workgroups per problem of a multi-problem launch (host side) ----------------
// Proportional to the problems' WORK (rounds = tiles a workgroup takes per trip round), at most
// `cap` in all -- one workgroup of these kernels is resident per CU (LDS), so a workgroup beyond
// the 256th waits for a whole share to finish -- never more than a problem has rounds, at least
// one each; what rounding down leaves over goes to the largest problem.  (Equal shares of capped
// wants gave an edge problem and two node problems a third of the device each although the edge
// rows are 8 x the node rows, and more than 256 workgroups queued the node problems behind the
// edge problem: Hi-LAM-256 34.4 -> 32.9 ms when this replaced them.)
static inline int64_t nlam_multi_shares(int n, const int64_t* rounds_in, int64_t* g, int64_t cap) {
  int64_t sum = 0, used = 0, tot = 0;
  int big = 0;
  for (int k = 0; k < n; ++k) {
    const int64_t r = rounds_in[k] < 1 ? 1 : rounds_in[k];
    sum += r;
    if (r > (rounds_in[big] < 1 ? 1 : rounds_in[big])) big = k;
  }
  for (int k = 0; k < n; ++k) {
    const int64_t r = rounds_in[k] < 1 ? 1 : rounds_in[k];
    g[k] = sum > cap ? (r * cap) / sum : r;
    if (g[k] < 1) g[k] = 1;
    used += g[k];
  }
  if (n > 0) {
    const int64_t rb = rounds_in[big] < 1 ? 1 : rounds_in[big];
    if (sum > cap && used < cap) g[big] += cap - used;
    // (the at-least-one bump of small problems can push the total past `cap`: the largest
    // problem gives the excess back, so no workgroup queues behind a full device round)
    if (sum > cap && used > cap) g[big] = g[big] - (used - cap) < 1 ? 1 : g[big] - (used - cap);
    if (g[big] > rb) g[big] = rb;
  }
  for (int k = 0; k < n; ++k) tot += g[k];
  return tot;
}

// ---- two-phase staging: issue every row load of a source first (they stay in
// flight together), write the LDS tile afterwards.  Loads are unconditional --
// slots >= nrows read a valid row (their index is clamped by the caller's
// row_ptr) and are zeroed when written -- so the compiler emits no branches and
// no per-load waits.  NV = number of float4 (or floats) per lane that covers the
// 32 rows: NV >= ceil(32 / (64 / (width/4))) for the vector form.
template <int NV, typename RowPtr>
__device__ __forceinline__ void load_rows_v(f32x4 (&v)[NV], int width, int lane, RowPtr row_ptr) {
  const int lpr = width >> 2;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr, c4 = lane - sub * lpr;
  const bool on = sub < rpi;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (k * rpi < NLAM_TILE) {   // wave-uniform
      const int t = sub + k * rpi;
      const float* src = row_ptr((on && t < NLAM_TILE) ? t : 0);
      v[k] = reinterpret_cast<const f32x4*>(src)[on ? c4 : 0];
    }
  }
}
template <int NV, bool ADD>
__device__ __forceinline__ void put_rows_v(float* __restrict__ tile, int ld, int col0, int width,
                                           int nrows, int lane, const f32x4 (&v)[NV]) {
  const int lpr = width >> 2;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr, c4 = lane - sub * lpr;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int t = sub + k * rpi;
    if (sub < rpi && t < NLAM_TILE) {
      f32x4 x = v[k];
      if (t >= nrows) x = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4* dst = reinterpret_cast<f32x4*>(tile + t * ld + col0) + c4;
      if (ADD) x += *dst;
      *dst = x;
    }
  }
}
// ---- gathers by per-lane row index --------------------------------------------------
// The lambda / __shfl form above costs one ds_bpermute per row load, and under register
// pressure the compiler serialises them (shuffle -> wait -> address -> load): in-kernel stamps
// of edge_bwd showed 5.6 k cycles to ISSUE the 40 row loads of a tile.  Fetching each lane's
// indices from the global tables instead exposes a dependent global latency (measured: worse).
// So the slot indices a tile already holds (lanes 0..31, prefetched one tile ahead) are stashed
// in a 32-entry LDS table per index kind, and every lane reads the indices of ITS rows (slot
// sub + k * rpi) with plain, batched ds_reads; the row loads follow without cross-lane traffic.
// Padded slots hold a valid (clamped) index and are zeroed by put_rows_v.
__device__ __forceinline__ void stash_slot_index(int* __restrict__ tab, int value, int lane) {
  if (lane < NLAM_TILE) tab[lane] = value;
}
template <int NV>
__device__ __forceinline__ void lane_row_index(int (&idx)[NV], const int* __restrict__ tab,
                                               int width, int lane) {
  const int lpr = width >> 2;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int t = sub + k * rpi;
    idx[k] = tab[(sub < rpi && t < NLAM_TILE) ? t : 0];
  }
}
template <int NV>
__device__ __forceinline__ void load_rows_i(f32x4 (&v)[NV], const float* __restrict__ base,
                                            int64_t ld, const int (&idx)[NV], int width, int lane) {
  const int lpr = width >> 2;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr;
  const int c4 = sub < rpi ? lane - sub * lpr : 0;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (k * rpi < NLAM_TILE)   // wave-uniform
      v[k] = reinterpret_cast<const f32x4*>(base + (int64_t)idx[k] * ld)[c4];
  }
}
// tile rows -> global rows base + idx[k] * ld (scatter by per-lane row index), optionally + res
template <int NV, bool RES>
__device__ __forceinline__ void store_rows_i(const float* __restrict__ tile, int ldt, int col0,
                                             int width, int nrows, int lane,
                                             float* __restrict__ base, int64_t ld,
                                             const int (&idx)[NV],
                                             const float* __restrict__ rbase = nullptr,
                                             int64_t rld = 0) {
  const int lpr = width >> 2;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr, c4 = lane - sub * lpr;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int t = sub + k * rpi;
    if (sub < rpi && t < nrows) {
      f32x4 x = *(reinterpret_cast<const f32x4*>(tile + t * ldt + col0) + c4);
      if (RES) x += reinterpret_cast<const f32x4*>(rbase + (int64_t)idx[k] * rld)[c4];
      reinterpret_cast<f32x4*>(base + (int64_t)idx[k] * ld)[c4] = x;
    }
  }
}

// The receivers of a receiver-aligned tile are CONSECUTIVE rows r0 .. r0 + nr - 1 (nr <= 32, ~4-9
// on the mesh graphs): their rows are loaded once (ceil(nr / rows-per-instruction) wave-wide
// loads instead of the 32-row gather that fetched every receiver row deg times) and expanded to
// the edge slots from LDS.  A CU issues a wave-wide 16-byte access every ~40 cycles whatever it
// hits, so the gathers are priced by their NUMBER.  NV loads cover NV * rows-per-instruction
// receivers (d = 64: 4 per load); callers prefetch NV = 4 and fetch the rest, if a tile has
// more than 16 receivers, when they stage.
template <int NV>
__device__ __forceinline__ void load_rows_c(f32x4 (&v)[NV], const float* __restrict__ base,
                                            int64_t ld, int r0, int nr, int width, int lane) {
  const int lpr = width >> 2;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr;
  const int c4 = sub < rpi ? lane - sub * lpr : 0;
  const int last = nr - 1;
#pragma unroll
  for (int k = 0; k < NV; ++k) {   // unconditional (slots past nr re-read the last row): no branches
    const int t = sub + k * rpi;
    v[k] = reinterpret_cast<const f32x4*>(base + (int64_t)(r0 + (t < last ? t : last)) * ld)[c4];
  }
}
// accumulator layout (lane = slot t, half h) <- tile row `row` of that slot (its receiver's row)
template <int NB, bool ADD>
__device__ __forceinline__ void tile_rows_to_acc(f32x16 (&acc)[NB], const float* __restrict__ tile,
                                                 int ld, int row, bool valid, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * ld + 32 * nb + 8 * q + 4 * h);
      if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ADD) acc[nb][4 * q + j] += v[j];
        else acc[nb][4 * q + j] = v[j];
      }
    }
  }
}

// scalar form for narrow / unaligned sources (width <= 2 NS)
template <int NS, typename RowPtr>
__device__ __forceinline__ void load_rows_s(float (&v)[NS], int width, int lane, RowPtr row_ptr) {
  const int total = NLAM_TILE * width;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int idx = lane + 64 * k;
    const bool on = idx < total;
    const int t = on ? idx / width : 0, c = on ? idx - t * width : 0;
    v[k] = row_ptr(t)[c];
  }
}
template <int NS>
__device__ __forceinline__ void put_rows_s(float* __restrict__ tile, int ld, int col0, int width,
                                           int nrows, int lane, const float (&v)[NS]) {
  const int total = NLAM_TILE * width;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int idx = lane + 64 * k;
    if (idx < total) {
      const int t = idx / width, c = idx - t * width;
      tile[t * ld + col0 + c] = (t < nrows) ? v[k] : 0.f;
    }
  }
}

// Contiguous-row views (rows r0 .. r0+nrows of batch item b).  Vector-capable
// views are staged in two phases (view_load_v: issue all loads; view_put_v: write
// the tile) so that several sources overlap their latency; narrow / unaligned
// views (feature embedders, 17-wide outputs) use the simple scalar stager.
template <int NV>
__device__ __forceinline__ void view_load_v(f32x4 (&v)[NV], const RowView& src, int64_t b,
                                            int64_t r0, int nrows, int lane) {
  const float* base = src.ptr + b * src.bstride + r0 * src.ld;
  const int last = nrows - 1;
  auto rp = [&](int t) { return base + (int64_t)(t < last ? t : last) * src.ld; };
  load_rows_v<NV>(v, src.width, lane, rp);
}
// Narrow / unaligned rows (static features of the embedders, the 17-wide output map): dword
// loads over the 32 x width elements of the tile, in TWO phases per chunk of 12 per lane -- all
// loads of the chunk are issued, then written to LDS.  (The plain load -> store loop paid one
// global round trip per 64 elements: 9 serialized round trips = 15.6 k cycles, 49 % of the tile
// time, for the 17-wide gy of the output map's backward.)  The (row, column) of an element
// advances incrementally: no per-element integer division by the run-time width.
__device__ __forceinline__ void view_stage_s(float* __restrict__ tile, int ld, int col0,
                                             const RowView& src, int64_t b, int64_t r0, int nrows,
                                             int lane) {
  const float* base = src.ptr + b * src.bstride + r0 * src.ld;
  const int width = src.width;
  const int total = NLAM_TILE * width;
  const int dq = 64 / width, dr = 64 - dq * width;   // (row, column) step of idx += 64
  constexpr int CH = 12;
  int t = lane / width, c = lane - t * width;
  const int last = nrows - 1;
  for (int i0 = lane; i0 < total; i0 += 64 * CH) {
    float v[CH];
    int tt = t, cc = c;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int tr = tt < last ? tt : last;          // (slots past the tile read a valid row)
      v[k] = base[(int64_t)tr * src.ld + cc];
      cc += dr; tt += dq;
      if (cc >= width) { cc -= width; tt += 1; }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (i0 + 64 * k < total) tile[t * ld + col0 + c] = (t < nrows) ? v[k] : 0.f;
      c += dr; t += dq;
      if (c >= width) { c -= width; t += 1; }
    }
  }
}

// The same in two separate calls (register prefetch of a narrow source one tile ahead):
// NS >= ceil(32 * width / 64) values per lane.
template <int NS>
__device__ __forceinline__ void view_load_s(float (&v)[NS], const RowView& src, int64_t b,
                                            int64_t r0, int nrows, int lane) {
  const float* base = src.ptr + b * src.bstride + r0 * src.ld;
  const int width = src.width;
  const int dq = 64 / width, dr = 64 - dq * width;
  int t = lane / width, c = lane - t * width;
  const int last = nrows - 1;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int tr = t < last ? t : last;
    v[k] = base[(int64_t)tr * src.ld + c];
    c += dr; t += dq;
    if (c >= width) { c -= width; t += 1; }
  }
}
template <int NS>
__device__ __forceinline__ void view_put_s(float* __restrict__ tile, int ld, int col0, int width,
                                           int nrows, int lane, const float (&v)[NS]) {
  const int total = NLAM_TILE * width;
  const int dq = 64 / width, dr = 64 - dq * width;
  int t = lane / width, c = lane - t * width;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    if (lane + 64 * k < total) tile[t * ld + col0 + c] = (t < nrows) ? v[k] : 0.f;
    c += dr; t += dq;
    if (c >= width) { c -= width; t += 1; }
  }
}

// zero columns [col0, col0+width) of the tile (K padding)
__device__ __forceinline__ void zero_cols(float* __restrict__ tile, int ld, int col0, int width,
                                          int lane) {
  for (int idx = lane; idx < NLAM_TILE * width; idx += 64) {
    const int t = idx / width, c = idx - t * width;
    tile[t * ld + col0 + c] = 0.f;
  }
}

// ---- LDS tile <-> accumulator layout -----------------------------------------
// acc[nb][4q..4q+3] <-> tile[t][32 nb + 8 q + 4 h .. +3]
template <int NB>
__device__ __forceinline__ void tile_to_acc(f32x16 (&acc)[NB], const float* __restrict__ tile,
                                            int ld, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + t * ld + 32 * nb + 8 * q + 4 * h);
      acc[nb][4 * q + 0] = v[0];
      acc[nb][4 * q + 1] = v[1];
      acc[nb][4 * q + 2] = v[2];
      acc[nb][4 * q + 3] = v[3];
    }
  }
}
template <int NB>
__device__ __forceinline__ void acc_to_tile(const f32x16 (&acc)[NB], float* __restrict__ tile,
                                            int ld, int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[nb][4 * q + 0], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
      *reinterpret_cast<f32x4*>(tile + t * ld + 32 * nb + 8 * q + 4 * h) = v;
    }
  }
}

// per-feature vector (bias / gamma / beta) in accumulator layout: v[nb][r] = p[f(nb,r,h)]
template <int NB>
__device__ __forceinline__ void vec_to_acc(f32x16 (&v)[NB], const float* __restrict__ p, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(p + 32 * nb + 8 * q + 4 * h);
      v[nb][4 * q + 0] = x[0];
      v[nb][4 * q + 1] = x[1];
      v[nb][4 * q + 2] = x[2];
      v[nb][4 * q + 3] = x[3];
    }
  }
}

// ---- GEMMs -------------------------------------------------------------------
// out[nb] += W[32 nb .. +31][0 .. 8*kc) . X^T, X read from an LDS row tile.
// Ws: LDS weight image, row stride ldw (= k_pad + 4), rows = output features.
template <int NB>
__device__ __forceinline__ void gemm_tile(f32x16 (&out)[NB], const float* __restrict__ Ws, int ldw,
                                          const float* __restrict__ tile, int ld, int kc,
                                          int lane) {
  const int t = lane & 31, h = lane >> 5;
  for (int c = 0; c < kc; ++c) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(tile + t * ld + 8 * c + 4 * h);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Ws + (32 * nb + t) * ldw + 8 * c + 4 * h);
      out[nb] = mfma32(a[0], b[0], out[nb]);
      out[nb] = mfma32(a[1], b[1], out[nb]);
      out[nb] = mfma32(a[2], b[2], out[nb]);
      out[nb] = mfma32(a[3], b[3], out[nb]);
    }
  }
}

// out[nb] += W[32 nb ..][k0 .. k0 + 32 KB) . IN, with IN (KB feature blocks) in
// accumulator layout used directly as the B operand.
template <int NB, int KB>
__device__ __forceinline__ void gemm_acc(f32x16 (&out)[NB], const float* __restrict__ Ws, int ldw,
                                         int k0, const f32x16 (&in)[KB], int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(Ws + (32 * nb + t) * ldw + k0 + 32 * kb +
                                                        8 * q + 4 * h);
        out[nb] = mfma32(a[0], in[kb][4 * q + 0], out[nb]);
        out[nb] = mfma32(a[1], in[kb][4 * q + 1], out[nb]);
        out[nb] = mfma32(a[2], in[kb][4 * q + 2], out[nb]);
        out[nb] = mfma32(a[3], in[kb][4 * q + 3], out[nb]);
      }
    }
  }
}

// out[nb] += W^T . IN : out feature i = column i of W, summed over W's rows k
// (the data-gradient GEMM g_in = W^T g_out).  Ws rows = k (KB blocks of 32),
// columns = i.  A operand element (i, k): Ws[k][i] -> 4 scalar LDS reads per
// 4 MFMAs, consecutive lanes consecutive i (conflict-free).
template <int NB, int KB>
__device__ __forceinline__ void gemm_acc_wt(f32x16 (&out)[NB], const float* __restrict__ Ws,
                                            int ldw, int i0, const f32x16 (&in)[KB], int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kbase = 32 * kb + 8 * q + 4 * h;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float* col = Ws + i0 + 32 * nb + t;
        out[nb] = mfma32(col[(kbase + 0) * ldw], in[kb][4 * q + 0], out[nb]);
        out[nb] = mfma32(col[(kbase + 1) * ldw], in[kb][4 * q + 1], out[nb]);
        out[nb] = mfma32(col[(kbase + 2) * ldw], in[kb][4 * q + 2], out[nb]);
        out[nb] = mfma32(col[(kbase + 3) * ldw], in[kb][4 * q + 3], out[nb]);
      }
    }
  }
}

// dW[ib][jb] += sum_t G[t][32 ib + .] (x) X[t][32 jb + .] over the 32 tile rows:
// the weight-gradient outer product.  G and X are LDS row tiles.  Result block
// layout: row i = 32 ib + 8 (r >> 2) + 4 h + (r & 3), column j = 32 jb + (lane & 31).
template <int NI, int NJ>
__device__ __forceinline__ void outer_accum(f32x16 (&dW)[NI][NJ], const float* __restrict__ G,
                                            int ldg, int gcol0, const float* __restrict__ X,
                                            int ldx, int xcol0, int lane) {
  const int i = lane & 31, h = lane >> 5;
#pragma unroll 4
  for (int s = 0; s < NLAM_TILE / 2; ++s) {
    const int t = 2 * s + h;
    float a[NI], b[NJ];
#pragma unroll
    for (int ib = 0; ib < NI; ++ib) a[ib] = G[t * ldg + gcol0 + 32 * ib + i];
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) b[jb] = X[t * ldx + xcol0 + 32 * jb + i];
#pragma unroll
    for (int ib = 0; ib < NI; ++ib)
#pragma unroll
      for (int jb = 0; jb < NJ; ++jb) dW[ib][jb] = mfma32(a[ib], b[jb], dW[ib][jb]);
  }
}

// ---- LayerNorm in accumulator layout -----------------------------------------
template <int NB>
__device__ __forceinline__ void ln_stats(const f32x16 (&z)[NB], float& mean, float& rstd) {
  constexpr float inv_d = 1.0f / (32.0f * NB);
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += z[nb][r];
  s = lane_xor32_sum(s);
  mean = s * inv_d;
  float v = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = z[nb][r] - mean;
      v += d * d;
    }
  v = lane_xor32_sum(v);
  rstd = rsqrtf(v * inv_d + 1e-5f);
}

// y = (z - mean) rstd gamma + beta, in place
template <int NB>
__device__ __forceinline__ void ln_apply(f32x16 (&z)[NB], const float* __restrict__ gamma,
                                         const float* __restrict__ beta, int lane) {
  float mean, rstd;
  ln_stats<NB>(z, mean, rstd);
  const int h = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 32 * nb + 8 * q + 4 * h);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 32 * nb + 8 * q + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        z[nb][4 * q + j] = (z[nb][4 * q + j] - mean) * rstd * g[j] + b[j];
    }
  }
}

// ---- coalesced tile -> global row stores -------------------------------------
template <bool VEC, typename RowPtr>
__device__ __forceinline__ void store_rows(const float* __restrict__ tile, int ld, int col0,
                                           int width, int nrows, int lane, RowPtr row_ptr) {
  if (VEC) {
    const int lpr = width >> 2;
    const int rpi = 64 / lpr;
    const int sub = lane / lpr, c4 = lane - sub * lpr;
    for (int k = 0; k * rpi < NLAM_TILE; ++k) {   // uniform trip count (see stage_rows)
      const int t = sub + k * rpi;
      float* dst = row_ptr(t < NLAM_TILE ? t : 0);
      if (sub < rpi && t < nrows)
        reinterpret_cast<f32x4*>(dst)[c4] =
            *(reinterpret_cast<const f32x4*>(tile + t * ld + col0) + c4);
    }
  } else {
    for (int idx = lane; idx < nrows * width; idx += 64) {
      const int t = idx / width, c = idx - t * width;
      row_ptr(t)[c] = tile[t * ld + col0 + c];
    }
  }
}

// out[t][c] = tile[t][c] + res[t][c].  The residual rows are loaded four wave-instructions at a
// time (unconditionally, from a clamped valid row) BEFORE their adds and stores: the plain
// load -> add -> store loop exposed one global latency per iteration (stamps of mlp_bwd: 11 k
// of a 40 k-cycle tile in the residual store of the 64-wide input gradient).
template <bool VEC, int BATCH = 4, typename RowPtr, typename ResPtr>
__device__ __forceinline__ void store_rows_res(const float* __restrict__ tile, int ld, int col0,
                                               int width, int nrows, int lane, RowPtr row_ptr,
                                               ResPtr res_ptr) {
  if (VEC) {
    const int lpr = width >> 2;
    const int rpi = 64 / lpr;
    const int sub = lane / lpr, c4 = lane - sub * lpr;
    const bool on = sub < rpi;
    const int last = nrows - 1;
    for (int k0 = 0; k0 * rpi < NLAM_TILE; k0 += BATCH) {
      f32x4 r[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int t = sub + (k0 + u) * rpi;
        const float* rsrc = res_ptr((on && t < last) ? t : last);
        r[u] = reinterpret_cast<const f32x4*>(rsrc)[on ? c4 : 0];
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int t = sub + (k0 + u) * rpi;
        float* dst = row_ptr(t < NLAM_TILE ? t : 0);
        if (on && t < nrows)
          reinterpret_cast<f32x4*>(dst)[c4] =
              *(reinterpret_cast<const f32x4*>(tile + t * ld + col0) + c4) + r[u];
      }
    }
  } else {
    for (int idx = lane; idx < nrows * width; idx += 64) {
      const int t = idx / width, c = idx - t * width;
      row_ptr(t)[c] = tile[t * ld + col0 + c] + res_ptr(t)[c];
    }
  }
}

// ---- tile-local segmented sums -------------------------------------------------
// Rows [0, ne) of the tile are grouped by receiver (lane t < 32 holds rcv of row t, rows of
// one receiver are consecutive).  Lanes = features.  All 32 row reads are issued up front
// (no dependent LDS round trip per row); the running sum is emitted at each segment end
// (wave-uniform bit test), in row order -- bit-identical to a per-receiver loop.
// emit(receiver, f0, value) stores one 64-float slice.  Returns the number of segments.
template <int D, typename Emit>
__device__ __forceinline__ int tile_segment_sums(const float* __restrict__ tile, int ld, int ne,
                                                 int rcv, int lane, Emit emit) {
  const int t = lane & 31;
  const int rnext = __shfl_down(rcv, 1, 64);
  const bool is_end = (lane < 32) && (t < ne) && (t == ne - 1 || rnext != rcv);
  const unsigned ends = (unsigned)(__ballot(is_end) & 0xffffffffull);
#pragma unroll
  for (int f0 = 0; f0 < D; f0 += 64) {
    float v[NLAM_TILE];
#pragma unroll
    for (int s = 0; s < NLAM_TILE; ++s) v[s] = tile[s * ld + f0 + lane];
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < NLAM_TILE; ++s) {
      acc += v[s];
      if ((ends >> s) & 1u) {
        emit(__builtin_amdgcn_readlane(rcv, s), f0, acc);
        acc = 0.f;
      }
    }
  }
  return __popc(ends);
}

// ---- gradient helpers ----------------------------------------------------------
// lanes = features: acc[j] += sum_t tile[t][col0 + 64 j + lane]  (j < NV), the
// per-feature (bias / gamma / beta) gradient contribution of one tile.
template <int NV>
__device__ __forceinline__ void tile_colsum(float (&acc)[NV], const float* __restrict__ tile,
                                            int ld, int col0, int nrows, int lane) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float s = 0.f;
    for (int t = 0; t < nrows; ++t) s += tile[t * ld + col0 + 64 * j + lane];
    acc[j] += s;
  }
}

// Same over ALL 32 tile rows (the caller guarantees that rows >= nrows hold zeros): a fixed trip
// count lets four row reads fly per wait (eight spill in edge_bwd) instead of one dependent LDS round trip per row
// (stamps of edge_bwd: the three runtime-length column sums were ~5 k of a 41 k-cycle tile).
template <int NV>
__device__ __forceinline__ void tile_colsum_all(float (&acc)[NV], const float* __restrict__ tile,
                                                int ld, int col0, int lane) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float s = 0.f;
#pragma unroll 1
    for (int t0 = 0; t0 < NLAM_TILE; t0 += 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = tile[(t0 + u) * ld + col0 + 64 * j + lane];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u];
    }
    acc[j] += s;
  }
}

// Workgroup-level fold of per-wave weight-gradient blocks: every wave writes its
// blocks to its own LDS image (all four in parallel), then all 256 threads add the
// four images in wave order (fixed => deterministic) straight into the slab.
// img must hold 4 * (32 NI) * ldimg floats of dead LDS.  Call from all threads,
// after a __syncthreads() that retires the tiles / weights.
template <int NI, int NJ>
__device__ __forceinline__ void fold_blocks_to_slab(const f32x16 (&dW)[NI][NJ],
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
        mine[i * ldimg + 32 * jb + j] = dW[ib][jb][r];
      }
  __syncthreads();
  for (int i = tid; i < n; i += 256)
    slab[i] = ((img[i] + img[n + i]) + img[2 * n + i]) + img[3 * n + i];
  __syncthreads();
}
// same for per-feature vectors held as lanes = features (NV values per lane)
template <int NV>
__device__ __forceinline__ void fold_vec_lds(const float (&v)[NV], float* __restrict__ img, int wave,
                                             int lane) {
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float* dst = img + 64 * j + lane;
        *dst = (w == 0) ? v[j] : (*dst + v[j]);
      }
    }
    __syncthreads();
  }
}

// Up to NLAM_WIDE_MAXP independent problems of one kernel type in ONE launch: Hi-LAM's small
// mesh levels are bound by the latency of their ~24 launches per InteractionNet, not by work.
// Workgroup b serves problem k with first[k] <= b < first[k + 1] as block b - first[k] of
// first[k + 1] - first[k].
#define NLAM_WIDE_MAXP 8
// (the weight-gradient launches take more: the small layers' problems of a whole AR step are merged
// at the end of its backward -- they feed nothing in between; the argument block stays under 4 KB)
#define NLAM_WIDE_MAXP_OUTER 24
template <typename P, int N = NLAM_WIDE_MAXP>
struct WideMulti {
  int n;
  int first[N + 1];
  P p[N];
};
template <typename P, int N>
__device__ __forceinline__ int wide_multi_find(const WideMulti<P, N>& m, int b) {
  int k = 0;
  while (k + 1 < m.n && b >= m.first[k + 1]) ++k;
  return k;
}
