// Wide fused kernels (gfx950): the hidden_dim = 128 path of the InteractionNet / make_mlp
// blocks (reference interaction_net.py:63-131, utils.py:191-214).
//
// At d = 128 one split-bf16 weight matrix (two bf16 planes, 67.6 KB) plus one 32-row tile per
// wave is what fits next to each other in the 160 KB LDS of a CU, and a d x d weight-gradient
// accumulator (256 registers per lane) is all a wave can hold.  The layer is therefore cut at
// the Linear boundaries -- ONE weight matrix per kernel -- and every weight gradient is a lean
// streaming pass of its own:
//
//   nlam_lin_fwd        first Linear of every MLP / the three projections of edge_mlp.0
//   nlam_tail_fwd       h = a + b[idx] + c[idx];  y = [res +] [LN](W2 silu(h) + b2)
//                       [+ receiver aggregation over receiver-aligned tiles]; keeps h (training)
//   nlam_tail_bwd       from h and the incoming gradient: gz = LN'(z; g), gh = (W2^T gz) silu'(h)
//                       [+ receiver-side segment sums of gh], gamma / beta gradients
//   nlam_lin_bwd_data   gx = gy W [+ gx_add]
//   nlam_wide_outer     dW = G^T f(X) (f = identity | silu), db = colsum(G)
//
// Row-on-lane layout and split-bf16 ("bf16x3") / plain bf16 MFMA arithmetic as in
// fused_common.h / fused_bf16x3.h (TERMS = 3 | 1).  h is kept by the forward (HBM is 288 GB;
// the rows are written once, coalesced) so that the backward neither repeats the first GEMM
// nor gathers; edge rows h / gz live in receiver-sorted (CSR) POSITION order, which is the
// order the tiles walk, so they are contiguous.
#include <stdlib.h>

#include "fused_bf16x3.h"
#include <type_traits>
#include "fused_common.h"
#include "fused_fs.h"

struct WideTiling {
  const int32_t* tiles;      // edge mode: (ntiles, 4) = p0, p1, r0, r1; NULL = row mode
  int64_t ntiles;            // tiles per batch item
  int64_t rows;              // positions per batch item (M, or the row count)
  const int32_t* csr_rec;    // edge mode: receiver of a position
  const int32_t* csr_rowptr; // edge mode: n_rec + 1
};

struct WTile {
  int p0, ne, r0, nr;        // wave-uniform
};
// the header load alone / its decoding: a prologue issues the load, then its weight loads, and
// decodes (= waits for the header) only after those are in flight
__device__ __forceinline__ int4 wide_tile_raw(const WideTiling& tl, int64_t k) {
  return tl.tiles != nullptr ? reinterpret_cast<const int4*>(tl.tiles)[k] : int4{0, 0, 0, 0};
}
__device__ __forceinline__ WTile wide_tile_decode(const WideTiling& tl, int64_t k, const int4& hdr) {
  WTile w;
  if (tl.tiles != nullptr) {
    w.p0 = hdr.x; w.ne = hdr.y - hdr.x; w.r0 = hdr.z; w.nr = hdr.w - hdr.z;
  } else {
    w.p0 = (int)(k * NLAM_TILE);
    const int64_t left = tl.rows - (int64_t)w.p0;
    w.ne = (int)(left < NLAM_TILE ? left : NLAM_TILE);
    w.r0 = 0; w.nr = 0;
  }
  return w;
}
__device__ __forceinline__ WTile wide_tile(const WideTiling& tl, int64_t k) {
  WTile w;
  if (tl.tiles != nullptr) {
    const int4 hdr = reinterpret_cast<const int4*>(tl.tiles)[k];
    w.p0 = hdr.x; w.ne = hdr.y - hdr.x; w.r0 = hdr.z; w.nr = hdr.w - hdr.z;
  } else {
    w.p0 = (int)(k * NLAM_TILE);
    const int64_t left = tl.rows - (int64_t)w.p0;
    w.ne = (int)(left < NLAM_TILE ? left : NLAM_TILE);
    w.r0 = 0; w.nr = 0;
  }
  return w;
}
// index of tile slot t (lanes 0..31; clamped to the last valid position so every load is legal;
// a tile of receivers without in-edges -- ne == 0, p0 possibly == M -- reads the position before)
__device__ __forceinline__ int wide_index(const int32_t* idx, const WTile& w, int lane) {
  const int t = lane & 31;
  const int pos = w.ne > 0 ? w.p0 + (t < w.ne ? t : w.ne - 1) : (w.p0 > 0 ? w.p0 - 1 : 0);
  return idx ? idx[pos] : pos;
}

// float4 row registers (load_rows_v layout) -> contiguous global rows base + t * ld
template <int NV>
__device__ __forceinline__ void store_rows_regs(float* __restrict__ base, int64_t ld, int width,
                                                int nrows, int lane, const f32x4 (&v)[NV]) {
  const int lpr = width >> 2;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr, c4 = lane - sub * lpr;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int t = sub + k * rpi;
    if (sub < rpi && t < nrows) reinterpret_cast<f32x4*>(base + (int64_t)t * ld)[c4] = v[k];
  }
}

// per-wave fold of accumulator blocks into the slab, ONE block row at a time (the d = 128
// images of all block rows together would not fit in LDS).  img: 4 * 32 * ldimg floats.
template <int NI, int NJ>
__device__ __forceinline__ void fold_block_rows_to_slab(const f32x16 (&dW)[NI][NJ],
                                                        float* __restrict__ img, int ldimg,
                                                        float* __restrict__ slab, int tid, int wave,
                                                        int lane) {
  const int h = lane >> 5, j = lane & 31;
  const int n = 32 * ldimg;
  float* mine = img + wave * n;
#pragma unroll
  for (int ib = 0; ib < NI; ++ib) {
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r >> 2) + 4 * h + (r & 3);
        mine[i * ldimg + 32 * jb + j] = dW[ib][jb][r];
      }
    __syncthreads();
    for (int i = tid; i < n; i += 256)
      slab[ib * n + i] = ((img[i] + img[n + i]) + img[2 * n + i]) + img[3 * n + i];
    __syncthreads();
  }
}

// ============================================================== tail forward ===
// Diagnostic (NLAM_STAMP_WIDE=1): per-phase s_memtime cycle sums of the hidden-128 tail kernels over
// all waves (tools/stamp_wide.py): [0..7] tail_fwd, [8..15] tail_bwd.  Read through
// nlam_debug_fs_stamps() (fused_fs.hip delegates when the variable is set).
__device__ unsigned long long g_wide_stamps[16];
int nlam_wide_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_stamps), sizeof(unsigned long long) * 16) != hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wide_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
static int wide_stamp_flag() {
  static const int f = getenv("NLAM_STAMP_WIDE") != nullptr ? 1 : 0;
  return f;
}
#define WSTAMP(k)                                                   \
  if constexpr (STAMP) {                                                      \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    __builtin_amdgcn_s_waitcnt(0xC07F);                             \
    __builtin_amdgcn_sched_barrier(0);                              \
    wst[k] += now_ - wprev;                                         \
    wprev = now_;                                                   \
  }

struct TailFwdParams {
  WideTiling tl;
  RowView a; const int32_t* idx_a;
  RowView b; const int32_t* idx_b;       // optional (b.ptr == NULL)
  RowView c; const int32_t* idx_c;       // optional
  const float* W2; int64_t ldW2; const float* b2; const float* gamma; const float* beta;
  int n_out;
  float* h_out; int64_t h_bstride;       // optional: (B, rows, D), pitch D, position order
  float* y; int64_t y_bstride; int64_t y_ld; const int32_t* idx_y;   // optional row output
  RowView res;                           // optional residual (rows addressed like y)
  float* agg; int64_t agg_bstride; int64_t agg_ld; const float* inv_deg;   // optional (edge mode)
  int B;
  int vec_y;                             // y / res rows 16-byte aligned, n_out % 4 == 0
  int stamp;                             // NLAM_STAMP_WIDE=1
  // optional projection in front of the tail (node update: h = a + pre . preW^T, preW (D, D)):
  // one-tile-per-wave problems only -- the workgroup holds preW's image first and W2's after it
  RowView pre; const float* preW; int64_t ldpreW;
};

template <int D, int NOUTB, bool HAS_LN, int TERMS, bool STAMP = false, bool PRE = false>
__global__ __launch_bounds__(256) void tail_fwd_kernel(TailFwdParams p) {
#define TAIL_BID blockIdx.x
#define TAIL_NBLK gridDim.x
#include "tail_fwd_body.h"
#undef TAIL_BID
#undef TAIL_NBLK
}

// Several independent MLP tails in one launch (the static-feature embedders of a model: ten small
// problems at the start of a Hi-LAM step, latency each): workgroup b serves problem k with
// first[k] <= b < first[k + 1] as block b - first[k] of that share.  The same body text.
constexpr int TAIL_MAXP = 8;   // (kernel argument block: 8 x ~390 B)
template <int D, int NOUTB, bool HAS_LN, int TERMS>
__global__ __launch_bounds__(256) void tail_fwd_multi_kernel(WideMulti<TailFwdParams, TAIL_MAXP> m) {
  constexpr bool STAMP = false, PRE = false;
  const int k_ = wide_multi_find(m, (int)blockIdx.x);
  const TailFwdParams& p = m.p[k_];
#define TAIL_BID (blockIdx.x - (unsigned)m.first[k_])
#define TAIL_NBLK ((unsigned)(m.first[k_ + 1] - m.first[k_]))
#include "tail_fwd_body.h"
#undef TAIL_BID
#undef TAIL_NBLK
}

static unsigned wide_grid(int64_t total_tiles) {
  int64_t g = (total_tiles + 3) / 4;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return (unsigned)g;
}

template <int D, int NOUTB, bool HAS_LN, int TERMS>
static int launch_tail_fwd(const TailFwdParams& p, hipStream_t s) {
  const size_t lds = b3_image_bytes(32 * NOUTB, D) + (size_t)3 * 32 * NOUTB * sizeof(float) +
                     (size_t)4 * NLAM_TILE * (D + 4) * sizeof(float) +
                     (size_t)4 * 4 * NLAM_TILE * sizeof(int);
  NLAM_REQUIRE(lds <= 160 * 1024, "tail_fwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = tail_fwd_kernel<D, NOUTB, HAS_LN, TERMS>;
  if (HAS_LN && TERMS == 3 && p.stamp) kern = tail_fwd_kernel<D, NOUTB, HAS_LN, TERMS, HAS_LN && TERMS == 3>;
  if constexpr (HAS_LN && 32 * NOUTB == D) {
    if (p.preW != nullptr) kern = tail_fwd_kernel<D, NOUTB, HAS_LN, TERMS, false, true>;
  }
  NLAM_BIG_LDS(kern, "tail_fwd_kernel");
  kern<<<wide_grid(p.tl.ntiles * p.B), 256, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("tail_fwd_kernel");
  return 0;
}

static bool wide_view_ok(const float* ptr, int64_t bstride, int64_t ld, int d) {
  return view_vec_ok(ptr, bstride, ld, d);
}

// the node update's second projection inside the tail: one-tile-per-wave problems at hidden 128
extern "C" int nlam_tail_fwd_pre_supported(int d, int64_t B, int64_t rows) {
  return d == 128 && nlam_mfma_terms() != 0 && B >= 1 && rows >= 1 &&
         B * ((rows + NLAM_TILE - 1) / NLAM_TILE) <= 4 * 256;
}

static int tail_fwd_impl(
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
    int64_t B, int d, int io_bf16, void* stream,
    const float* pre, int64_t pre_bstride, int64_t pre_ld, const float* preW, int64_t ldpreW);

extern "C" int nlam_tail_fwd(
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
    int64_t B, int d, int io_bf16, void* stream) {
  return tail_fwd_impl(tiles, ntiles, rows, csr_rec, csr_rowptr, a, a_bstride, a_ld, idx_a, b, b_bstride,
                       b_ld, idx_b, c, c_bstride, c_ld, idx_c, W2, ldW2, b2, gamma, beta, n_out, h_out,
                       h_bstride, z_keep, z_bstride, y, y_bstride, y_ld, idx_y, res, res_bstride, res_ld,
                       agg, agg_bstride, agg_ld, inv_deg, B, d, io_bf16, stream, nullptr, 0, 0, nullptr, 0);
}

extern "C" int nlam_tail_fwd_pre(
    int64_t rows, const float* a, int64_t a_bstride, int64_t a_ld,
    const float* pre, int64_t pre_bstride, int64_t pre_ld, const float* preW, int64_t ldpreW,
    const float* W2, int64_t ldW2, const float* b2, const float* gamma, const float* beta,
    float* h_out, int64_t h_bstride, float* y, int64_t y_bstride, int64_t y_ld,
    const float* res, int64_t res_bstride, int64_t res_ld, int64_t B, int d, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(nlam_tail_fwd_pre_supported(d, B, rows),
               "nlam_tail_fwd_pre: hidden 128 and at most 1024 row tiles (B * ceil(rows / 32))");
  NLAM_REQUIRE(pre != nullptr && preW != nullptr && gamma != nullptr && beta != nullptr,
               "nlam_tail_fwd_pre: null operand");
  NLAM_REQUIRE(view_vec_ok(pre, pre_bstride, pre_ld, d) && nlam_aligned16(preW) && ldpreW % 4 == 0,
               "nlam_tail_fwd_pre: projected rows / weights must be 16-byte aligned, pitch %% 4 == 0");
  return tail_fwd_impl(nullptr, (rows + NLAM_TILE - 1) / NLAM_TILE, rows, nullptr, nullptr, a, a_bstride,
                       a_ld, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr, W2, ldW2, b2, gamma,
                       beta, d, h_out, h_bstride, nullptr, 0, y, y_bstride, y_ld, nullptr, res, res_bstride,
                       res_ld, nullptr, 0, 0, nullptr, B, d, 0, stream, pre, pre_bstride, pre_ld, preW,
                       ldpreW);
}

static int tail_fwd_impl(
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
    int64_t B, int d, int io_bf16, void* stream,
    const float* pre, int64_t pre_bstride, int64_t pre_ld, const float* preW, int64_t ldpreW) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(io_bf16 == 0 || d == 256, "nlam_tail_fwd: bf16 rows exist at hidden 256 only");
  if (d == 256)
    return nlam_fs_tail_fwd_256(tiles, ntiles, rows, csr_rec, csr_rowptr, a, a_bstride, a_ld, idx_a,
                                b, b_bstride, b_ld, idx_b, c, c_bstride, c_ld, idx_c, W2, ldW2, b2,
                                gamma, beta, n_out, h_out, h_bstride, z_keep, z_bstride, y, y_bstride,
                                y_ld, idx_y, res, res_bstride, res_ld, agg, agg_bstride, agg_ld,
                                inv_deg, B, io_bf16, stream);
  (void)z_keep; (void)z_bstride;   // hidden 128 repeats the GEMM in its backward instead
  NLAM_REQUIRE(d == 128, "nlam_tail_fwd: hidden width %d unsupported (128, 256)", d);
  NLAM_REQUIRE(nlam_mfma_terms() != 0, "nlam_tail_fwd: needs NLAM_MFMA=bf16x3|bf16");
  NLAM_REQUIRE(n_out >= 1 && n_out <= d, "nlam_tail_fwd: n_out %d out of range", n_out);
  NLAM_REQUIRE((gamma == nullptr) == (beta == nullptr), "nlam_tail_fwd: gamma/beta mismatch");
  NLAM_REQUIRE(gamma == nullptr || n_out == d, "nlam_tail_fwd: LayerNorm needs n_out == d");
  NLAM_REQUIRE(gamma != nullptr || n_out <= 32, "nlam_tail_fwd: no-LN form supports n_out <= 32");
  NLAM_REQUIRE(wide_view_ok(a, a_bstride, a_ld, d) && (!b || wide_view_ok(b, b_bstride, b_ld, d)) &&
                   (!c || wide_view_ok(c, c_bstride, c_ld, d)) && (!c || b),
               "nlam_tail_fwd: sources must be 16-byte aligned rows of width d");
  NLAM_REQUIRE(h_out == nullptr || (nlam_aligned16(h_out) && h_bstride % 4 == 0),
               "nlam_tail_fwd: h_out misaligned");
  NLAM_REQUIRE(agg == nullptr || (tiles != nullptr && csr_rec != nullptr && csr_rowptr != nullptr &&
                                  agg_ld >= n_out && n_out == d),
               "nlam_tail_fwd: aggregation needs edge tiles and n_out == d");
  NLAM_REQUIRE(tiles != nullptr || ntiles == (rows + NLAM_TILE - 1) / NLAM_TILE,
               "nlam_tail_fwd: row mode expects ntiles == ceil(rows / 32)");
  TailFwdParams p;
  p.tl = WideTiling{tiles, ntiles, rows, csr_rec, csr_rowptr};
  p.a = RowView{a, a_bstride, a_ld, d}; p.idx_a = idx_a;
  p.b = RowView{b, b_bstride, b_ld, d}; p.idx_b = idx_b;
  p.c = RowView{c, c_bstride, c_ld, d}; p.idx_c = idx_c;
  p.W2 = W2; p.ldW2 = ldW2; p.b2 = b2; p.gamma = gamma; p.beta = beta; p.n_out = n_out;
  p.h_out = h_out; p.h_bstride = h_bstride;
  p.y = y; p.y_bstride = y_bstride; p.y_ld = y_ld; p.idx_y = idx_y;
  p.res = RowView{res, res_bstride, res_ld, n_out};
  p.agg = agg; p.agg_bstride = agg_bstride; p.agg_ld = agg_ld; p.inv_deg = inv_deg;
  p.B = (int)B;
  p.vec_y = (y != nullptr && view_vec_ok(y, y_bstride, y_ld, n_out) &&
             (res == nullptr || view_vec_ok(res, res_bstride, res_ld, n_out))) ? 1 : 0;
  p.stamp = wide_stamp_flag();
  p.pre = RowView{pre, pre_bstride, pre_ld, d}; p.preW = preW; p.ldpreW = ldpreW;
  NLAM_REQUIRE(preW == nullptr || (d == 128 && tiles == nullptr && ntiles * B <= 4 * 256),
               "nlam_tail_fwd: the fused projection needs one row tile per wave at hidden 128");
  NLAM_REQUIRE(idx_y == nullptr || y == nullptr || (p.vec_y && n_out == d),
               "nlam_tail_fwd: scattered output rows must be 16-byte aligned and d wide");
  hipStream_t s = (hipStream_t)stream;
  const bool t3 = nlam_mfma_terms() == 3;
  if (gamma != nullptr)
    return t3 ? launch_tail_fwd<128, 4, true, 3>(p, s) : launch_tail_fwd<128, 4, true, 1>(p, s);
  return t3 ? launch_tail_fwd<128, 1, false, 3>(p, s) : launch_tail_fwd<128, 1, false, 1>(p, s);
}

// ============================================================= tail backward ===
// Slab per workgroup: [dgamma (32 NOUTB) | dbeta (32 NOUTB)]  (HAS_LN only).
struct TailBwdParams {
  WideTiling tl;
  const float* h; int64_t h_bstride;                     // (B, rows, D), pitch D
  RowView g1; const int32_t* idx_g1; const float* scale1; // incoming gradient rows (scaled)
  RowView g2; const int32_t* idx_g2;                      // optional addend
  const float* W2; int64_t ldW2; const float* b2; const float* gamma; int n_out;
  float* gz_out; int64_t gz_bstride;                     // (B, rows, 32 NOUTB), position order
  float* gh; int64_t gh_bstride; int64_t gh_ld; const int32_t* idx_gh;   // (B, rows, D)
  float* gpr; int64_t gpr_bstride; int64_t gpr_ld;       // optional receiver-side sums of gh
  float* slab; int64_t slab_stride;
  int B;
  int vec_g;                                             // g1 / g2 rows float4-loadable
  int stamp;                                             // NLAM_STAMP_WIDE=1
};

template <int D, int NOUTB, bool HAS_LN, int TERMS, bool STAMP = false>
__global__ __launch_bounds__(256) void tail_bwd_kernel(TailBwdParams q) {
#define TAIL_BID blockIdx.x
#define TAIL_NBLK gridDim.x
#include "tail_bwd_body.h"
#undef TAIL_BID
#undef TAIL_NBLK
}

template <int D, int NOUTB, bool HAS_LN, int TERMS>
__global__ __launch_bounds__(256) void tail_bwd_multi_kernel(WideMulti<TailBwdParams, TAIL_MAXP> m) {
  constexpr bool STAMP = false;
  const int k_ = wide_multi_find(m, (int)blockIdx.x);
  const TailBwdParams& q = m.p[k_];
#define TAIL_BID (blockIdx.x - (unsigned)m.first[k_])
#define TAIL_NBLK ((unsigned)(m.first[k_ + 1] - m.first[k_]))
#include "tail_bwd_body.h"
#undef TAIL_BID
#undef TAIL_NBLK
}

template <int D, int NOUTB, bool HAS_LN, int TERMS>
static int launch_tail_bwd(const TailBwdParams& q, hipStream_t s) {
  const size_t lds = b3_image_bytes(32 * NOUTB, D) + (size_t)2 * 32 * NOUTB * sizeof(float) +
                     (size_t)4 * NLAM_TILE * (D + 4) * sizeof(float) +
                     (size_t)4 * 4 * NLAM_TILE * sizeof(int);
  NLAM_REQUIRE(lds <= 160 * 1024, "tail_bwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = tail_bwd_kernel<D, NOUTB, HAS_LN, TERMS>;
  if (HAS_LN && TERMS == 3 && q.stamp) kern = tail_bwd_kernel<D, NOUTB, HAS_LN, TERMS, HAS_LN && TERMS == 3>;
  NLAM_BIG_LDS(kern, "tail_bwd_kernel");
  kern<<<wide_grid(q.tl.ntiles * q.B), 256, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("tail_bwd_kernel");
  return 0;
}

extern "C" int64_t nlam_tail_bwd_slab_stride(int n_out) { return 2 * (int64_t)((n_out + 31) & ~31); }

extern "C" int nlam_tail_bwd(
    const int32_t* tiles, int64_t ntiles, int64_t rows, const int32_t* csr_rec,
    const int32_t* csr_rowptr, const float* h, int64_t h_bstride,
    const void* z_keep, int64_t z_bstride,
    const float* g1, int64_t g1_bstride, int64_t g1_ld, const int32_t* idx_g1, const float* scale1,
    const float* g2, int64_t g2_bstride, int64_t g2_ld, const int32_t* idx_g2,
    const float* W2, int64_t ldW2, const float* b2, const float* gamma, int n_out,
    float* gz_out, int64_t gz_bstride,
    float* gh, int64_t gh_bstride, int64_t gh_ld, const int32_t* idx_gh,
    float* gpr, int64_t gpr_bstride, int64_t gpr_ld,
    float* slab, int64_t slab_stride, int64_t B, int d, int io_bf16, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(io_bf16 == 0 || d == 256, "nlam_tail_bwd: bf16 rows exist at hidden 256 only");
  if (d == 256) {
    NLAM_REQUIRE(gpr == nullptr || (tiles != nullptr && csr_rec != nullptr && csr_rowptr != nullptr &&
                                    gpr_ld >= d), "nlam_tail_bwd: gpr needs edge tiles");
    NLAM_REQUIRE(tiles != nullptr || ntiles == (rows + NLAM_TILE - 1) / NLAM_TILE,
                 "nlam_tail_bwd: row mode expects ntiles == ceil(rows / 32)");
    NLAM_REQUIRE(n_out >= 1 && n_out <= d && (gamma != nullptr || n_out <= 32),
                 "nlam_tail_bwd: n_out %d unsupported", n_out);
    return nlam_fs_tail_bwd_256(tiles, ntiles, rows, csr_rec, csr_rowptr, h, h_bstride, z_keep,
                                z_bstride, g1, g1_bstride, g1_ld, idx_g1, scale1, g2, g2_bstride, g2_ld, idx_g2, W2, ldW2, b2,
                                gamma, n_out, gz_out, gz_bstride, gh, gh_bstride, gh_ld, idx_gh, gpr,
                                gpr_bstride, gpr_ld, slab, slab_stride, B, wide_grid(ntiles * B), io_bf16,
                                stream);
  }
  NLAM_REQUIRE(d == 128, "nlam_tail_bwd: hidden width %d unsupported (128, 256)", d);
  NLAM_REQUIRE(nlam_mfma_terms() != 0, "nlam_tail_bwd: needs NLAM_MFMA=bf16x3|bf16");
  NLAM_REQUIRE(n_out >= 1 && n_out <= d, "nlam_tail_bwd: n_out out of range");
  NLAM_REQUIRE(gamma == nullptr || n_out == d, "nlam_tail_bwd: LayerNorm needs n_out == d");
  NLAM_REQUIRE(gamma != nullptr || n_out <= 32, "nlam_tail_bwd: no-LN form supports n_out <= 32");
  NLAM_REQUIRE(h != nullptr && nlam_aligned16(h) && h_bstride % 4 == 0, "nlam_tail_bwd: bad h");
  NLAM_REQUIRE(g1 != nullptr && gz_out != nullptr && nlam_aligned16(gz_out) && gz_bstride % 4 == 0,
               "nlam_tail_bwd: g1 / gz_out missing or misaligned");
  NLAM_REQUIRE(gh != nullptr && view_vec_ok(gh, gh_bstride, gh_ld, d), "nlam_tail_bwd: bad gh view");
  NLAM_REQUIRE(gpr == nullptr || (tiles != nullptr && csr_rec != nullptr && csr_rowptr != nullptr &&
                                  gpr_ld >= d), "nlam_tail_bwd: gpr needs edge tiles");
  NLAM_REQUIRE(gamma == nullptr || (slab != nullptr && slab_stride >= nlam_tail_bwd_slab_stride(n_out)),
               "nlam_tail_bwd: slab too small");
  NLAM_REQUIRE(tiles != nullptr || ntiles == (rows + NLAM_TILE - 1) / NLAM_TILE,
               "nlam_tail_bwd: row mode expects ntiles == ceil(rows / 32)");
  TailBwdParams q;
  q.tl = WideTiling{tiles, ntiles, rows, csr_rec, csr_rowptr};
  q.h = h; q.h_bstride = h_bstride;
  q.g1 = RowView{g1, g1_bstride, g1_ld, n_out}; q.idx_g1 = idx_g1; q.scale1 = scale1;
  q.g2 = RowView{g2, g2_bstride, g2_ld, n_out}; q.idx_g2 = idx_g2;
  q.W2 = W2; q.ldW2 = ldW2; q.b2 = b2; q.gamma = gamma; q.n_out = n_out;
  q.gz_out = gz_out; q.gz_bstride = gz_bstride;
  q.gh = gh; q.gh_bstride = gh_bstride; q.gh_ld = gh_ld; q.idx_gh = idx_gh;
  q.gpr = gpr; q.gpr_bstride = gpr_bstride; q.gpr_ld = gpr_ld;
  q.slab = slab; q.slab_stride = slab_stride; q.B = (int)B;
  const int no = (n_out + 31) & ~31;
  q.vec_g = (n_out == no && view_vec_ok(g1, g1_bstride, g1_ld, n_out) &&
             (g2 == nullptr || view_vec_ok(g2, g2_bstride, g2_ld, n_out))) ? 1 : 0;
  q.stamp = wide_stamp_flag();
  hipStream_t s = (hipStream_t)stream;
  const bool t3 = nlam_mfma_terms() == 3;
  if (gamma != nullptr)
    return t3 ? launch_tail_bwd<128, 4, true, 3>(q, s) : launch_tail_bwd<128, 4, true, 1>(q, s);
  return t3 ? launch_tail_bwd<128, 1, false, 3>(q, s) : launch_tail_bwd<128, 1, false, 1>(q, s);
}

// ---- several MLP tails (row mode, LayerNorm, n_out = d = 128, no indices / residual) per launch
// shares[k] = workgroups of problem k (= the slabs its backward writes): proportional to the tiles,
// one round of the device in all
extern "C" int nlam_mlp_tail_multi_shares(int n, const int64_t* B, const int64_t* rows, int32_t* shares) {
  NLAM_REQUIRE(n >= 1 && n <= TAIL_MAXP && B && rows && shares, "nlam_mlp_tail_multi_shares: n %d out of [1, %d]",
               n, TAIL_MAXP);
  int64_t rounds[TAIL_MAXP], g[TAIL_MAXP];
  for (int k = 0; k < n; ++k) {
    NLAM_REQUIRE(B[k] >= 1 && rows[k] >= 1, "nlam_mlp_tail_multi_shares: empty problem %d", k);
    rounds[k] = (((rows[k] + NLAM_TILE - 1) / NLAM_TILE) * B[k] + 3) / 4;
  }
  nlam_multi_shares(n, rounds, g, 256);
  for (int k = 0; k < n; ++k) shares[k] = (int32_t)g[k];
  return 0;
}

extern "C" int nlam_mlp_tail_fwd_multi(int n, int d, const float* const* h, const float* const* W2,
                                       const int64_t* ldW2, const float* const* b2,
                                       const float* const* gamma, const float* const* beta,
                                       float* const* y, const int64_t* B, const int64_t* rows,
                                       void* stream) {
  NLAM_REQUIRE(n >= 1 && n <= TAIL_MAXP && d == 128 && nlam_mfma_terms() != 0,
               "nlam_mlp_tail_fwd_multi: n %d out of [1, %d], hidden 128, an MFMA mode", n, TAIL_MAXP);
  WideMulti<TailFwdParams, TAIL_MAXP> m;
  int32_t shares[TAIL_MAXP];
  if (nlam_mlp_tail_multi_shares(n, B, rows, shares)) return 1;
  m.n = n;
  m.first[0] = 0;
  for (int k = 0; k < n; ++k) {
    NLAM_REQUIRE(h[k] && W2[k] && b2[k] && gamma[k] && beta[k] && y[k] && nlam_aligned16(h[k]) &&
                     nlam_aligned16(y[k]), "nlam_mlp_tail_fwd_multi: operand of problem %d", k);
    TailFwdParams& p = m.p[k];
    p.tl = WideTiling{nullptr, (rows[k] + NLAM_TILE - 1) / NLAM_TILE, rows[k], nullptr, nullptr};
    p.a = RowView{h[k], rows[k] * d, d, d}; p.idx_a = nullptr;
    p.b = RowView{nullptr, 0, 0, d}; p.idx_b = nullptr;
    p.c = RowView{nullptr, 0, 0, d}; p.idx_c = nullptr;
    p.W2 = W2[k]; p.ldW2 = ldW2[k]; p.b2 = b2[k]; p.gamma = gamma[k]; p.beta = beta[k]; p.n_out = d;
    p.h_out = nullptr; p.h_bstride = 0;
    p.y = y[k]; p.y_bstride = rows[k] * d; p.y_ld = d; p.idx_y = nullptr;
    p.res = RowView{nullptr, 0, 0, d};
    p.agg = nullptr; p.agg_bstride = 0; p.agg_ld = 0; p.inv_deg = nullptr;
    p.B = (int)B[k]; p.vec_y = 1; p.stamp = 0;
    p.pre = RowView{nullptr, 0, 0, d}; p.preW = nullptr; p.ldpreW = 0;
    m.first[k + 1] = m.first[k] + shares[k];
  }
  for (int k = n; k < TAIL_MAXP; ++k) m.first[k + 1] = m.first[n];
  const size_t lds = b3_image_bytes(128, 128) + (size_t)3 * 128 * sizeof(float) +
                     (size_t)4 * NLAM_TILE * (128 + 4) * sizeof(float) + (size_t)4 * 4 * NLAM_TILE * sizeof(int);
  hipStream_t s = (hipStream_t)stream;
  if (nlam_mfma_terms() == 3) {
    auto kern = tail_fwd_multi_kernel<128, 4, true, 3>;
    NLAM_BIG_LDS(kern, "tail_fwd_multi_kernel");
    kern<<<(unsigned)m.first[n], 256, lds, s>>>(m);
  } else {
    auto kern = tail_fwd_multi_kernel<128, 4, true, 1>;
    NLAM_BIG_LDS(kern, "tail_fwd_multi_kernel");
    kern<<<(unsigned)m.first[n], 256, lds, s>>>(m);
  }
  NLAM_CHECK_LAUNCH("tail_fwd_multi_kernel");
  return 0;
}

// gz_out[k] (B, rows, d), gh[k] (B, rows, d) = the gradient of h; slab[k]: nslabs[k] =
// nlam_mlp_tail_multi_shares slabs of nlam_tail_bwd_slab_stride(d) floats [dgamma | dbeta]
extern "C" int nlam_mlp_tail_bwd_multi(int n, int d, const float* const* h, const float* const* gy,
                                       const float* const* W2, const int64_t* ldW2,
                                       const float* const* b2, const float* const* gamma,
                                       float* const* gz_out, float* const* gh, float* const* slab,
                                       const int32_t* nslabs, const int64_t* B, const int64_t* rows,
                                       void* stream) {
  NLAM_REQUIRE(n >= 1 && n <= TAIL_MAXP && d == 128 && nlam_mfma_terms() != 0,
               "nlam_mlp_tail_bwd_multi: n %d out of [1, %d], hidden 128, an MFMA mode", n, TAIL_MAXP);
  WideMulti<TailBwdParams, TAIL_MAXP> m;
  int32_t shares[TAIL_MAXP];
  if (nlam_mlp_tail_multi_shares(n, B, rows, shares)) return 1;
  m.n = n;
  m.first[0] = 0;
  for (int k = 0; k < n; ++k) {
    NLAM_REQUIRE(h[k] && gy[k] && W2[k] && b2[k] && gamma[k] && gz_out[k] && gh[k] && slab[k] &&
                     nlam_aligned16(h[k]) && nlam_aligned16(gy[k]) && nlam_aligned16(gz_out[k]) &&
                     nlam_aligned16(gh[k]) && nslabs[k] == shares[k],
                 "nlam_mlp_tail_bwd_multi: operand / slab count of problem %d", k);
    TailBwdParams& q = m.p[k];
    q.tl = WideTiling{nullptr, (rows[k] + NLAM_TILE - 1) / NLAM_TILE, rows[k], nullptr, nullptr};
    q.h = h[k]; q.h_bstride = rows[k] * d;
    q.g1 = RowView{gy[k], rows[k] * d, d, d}; q.idx_g1 = nullptr; q.scale1 = nullptr;
    q.g2 = RowView{nullptr, 0, 0, d}; q.idx_g2 = nullptr;
    q.W2 = W2[k]; q.ldW2 = ldW2[k]; q.b2 = b2[k]; q.gamma = gamma[k]; q.n_out = d;
    q.gz_out = gz_out[k]; q.gz_bstride = rows[k] * d;
    q.gh = gh[k]; q.gh_bstride = rows[k] * d; q.gh_ld = d; q.idx_gh = nullptr;
    q.gpr = nullptr; q.gpr_bstride = 0; q.gpr_ld = 0;
    q.slab = slab[k]; q.slab_stride = nlam_tail_bwd_slab_stride(d); q.B = (int)B[k];
    q.vec_g = 1; q.stamp = 0;
    m.first[k + 1] = m.first[k] + shares[k];
  }
  for (int k = n; k < TAIL_MAXP; ++k) m.first[k + 1] = m.first[n];
  const size_t lds = b3_image_bytes(128, 128) + (size_t)2 * 128 * sizeof(float) +
                     (size_t)4 * NLAM_TILE * (128 + 4) * sizeof(float) + (size_t)4 * 4 * NLAM_TILE * sizeof(int);
  hipStream_t s = (hipStream_t)stream;
  if (nlam_mfma_terms() == 3) {
    auto kern = tail_bwd_multi_kernel<128, 4, true, 3>;
    NLAM_BIG_LDS(kern, "tail_bwd_multi_kernel");
    kern<<<(unsigned)m.first[n], 256, lds, s>>>(m);
  } else {
    auto kern = tail_bwd_multi_kernel<128, 4, true, 1>;
    NLAM_BIG_LDS(kern, "tail_bwd_multi_kernel");
    kern<<<(unsigned)m.first[n], 256, lds, s>>>(m);
  }
  NLAM_CHECK_LAUNCH("tail_bwd_multi_kernel");
  return 0;
}

// ====================================================== data gradient of a Linear ===
// gx (B, rows, 32 KB) = gy (B, rows, 32 NOUTB) . W (32 NOUTB x 32 KB) [+ gx_add]
struct LinBwdDataParams {
  RowView gy;
  const float* W; int64_t ldW; int n_out; int k_in;
  float* gx; int64_t gx_bstride; int64_t gx_ld;
  const float* gx_add; int64_t ga_bstride; int64_t ga_ld;
  int64_t rows; int B;
};

template <int NOUTB, int KB, int TERMS>
__device__ __forceinline__ void lin_bwd_data_body(const LinBwdDataParams& q, int bid, int gdim,
                                                  float* smem) {
  constexpr int NO = 32 * NOUTB, K = 32 * KB;
  constexpr int LDT = (NO > K ? NO : K) + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the weight image is read TRANSPOSED only here (gx = W^T gy): pitch K + 16 (8 banks mod 64)
  // instead of K + 4 -- SQ counters: 50 % of this kernel's LDS cycles were bank conflicts
  constexpr int WP = K + 16;
  B3Image Wim;
  Wim.pitch = WP;
  Wim.swz = 0;
  Wim.hi = reinterpret_cast<__bf16*>(smem);
  Wim.lo = Wim.hi + NO * WP;
  float* tile = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) +
                                         (size_t)2 * NO * WP * sizeof(__bf16)) +
                wave * (NLAM_TILE * LDT);
  const int64_t tiles_per_b = (q.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * q.B;
  const int64_t tt0 = (int64_t)bid * 4 + wave;
  const int64_t tstride = (int64_t)gdim * 4;
  static_assert(NO == 128 && K == 128, "row mappings below are the 128-wide ones");
  constexpr int NV = K / 8;   // float4 per lane that cover a 32 x 128 tile (row t = sub + 2 k)
  const int sub = lane >> 5, c4 = lane & 31;
  // The gradient rows of tile n + 1 are requested as soon as tile n's registers are staged (their
  // round trip rides under tile n's MFMAs); past the end the last tile is requested again.
  f32x4 vg[4 * NOUTB];
  auto request = [&](int64_t task) {
    const int64_t tq = task < ntiles ? task : ntiles - 1;
    const int64_t b = tq / tiles_per_b;
    const int64_t r0 = (tq - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((q.rows - r0) < NLAM_TILE ? (q.rows - r0) : NLAM_TILE);
    RowView gv = q.gy;
    gv.width = NO;   // (compile-time row mapping: no per-load branches)
    view_load_v<4 * NOUTB>(vg, gv, b, r0, nrows, lane);
  };
  {
    // prologue: the weight loads and the FIRST tile's rows are in flight together (most launches
    // of the hierarchical models are one tile per wave), then the LDS image
    WLoad16<(NO * K / 4 + 255) / 256> lw;
    w16_issue(lw, q.W, q.ldW, q.n_out, q.k_in, NO, K, tid, 256);
    request(tt0);
    __builtin_amdgcn_sched_barrier(0);
    w16_commit(lw, Wim, 0, q.W, q.ldW, q.n_out, q.k_in, NO, K, tid, 256);
  }
  __syncthreads();
  // (nothing in flight at loop entry: the loop-top wait then counts this tile's stores as younger
  // than the rows it waits for on BOTH paths into the loop header, and does not drain them)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // (the loop exists twice, with and without addend: a load under a branch inside the loop would
  // make the compiler's wait for it a full drain)
  auto run = [&](auto has_add_t) {
  constexpr bool has_add = decltype(has_add_t)::value;
  for (int64_t tt = tt0; tt < ntiles; tt += tstride) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((q.rows - r0) < NLAM_TILE ? (q.rows - r0) : NLAM_TILE);
    const B3Tile Gp = b3_tile(tile, NO);
    put_rows_v_b3<4 * NOUTB>(Gp, 0, NO, nrows, lane, vg);
    // addend rows of THIS tile (used at its end), then the next tile's gradient rows: memory
    // operations retire in order, so waiting for the addend does not wait for the younger rows
    f32x4 va[NV];
    if constexpr (has_add) {
      RowView av{q.gx_add, q.ga_bstride, q.ga_ld, K};
      view_load_v<NV>(va, av, b, r0, nrows, lane);
    }
    // (no-addend loop: a wave's last tile requests nothing -- loads in flight at the end hold the
    // wave for a round trip; with an addend the request stays unconditional, or the wait for the
    // addend rows would cover it)
    if (has_add || tt + tstride < ntiles) request(tt + tstride);
    wave_sync();
    f32x16 gx[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) gx[kb][r] = 0.f;
    gemm_tile_wt_b3<KB, NOUTB, TERMS>(gx, Wim, 0, Gp, 0, lane);
    wave_sync();
    acc_to_tile<KB>(gx, tile, LDT, lane);
    wave_sync();
    // UNCONDITIONAL stores (padded slots repeat the tile's last row: same value, same address):
    // a store under a branch is not counted by the compiler's wait bookkeeping, and the next
    // loop-top wait would then drain every store of this tile
    float* ob = q.gx + b * q.gx_bstride + r0 * q.gx_ld;
    const int last = nrows - 1;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int t = sub + 2 * k;
      const int tc = t < last ? t : last;
      f32x4 x = *(reinterpret_cast<const f32x4*>(tile + tc * LDT) + c4);
      if constexpr (has_add) x += va[k];
      reinterpret_cast<f32x4*>(ob + (int64_t)tc * q.gx_ld)[c4] = x;
    }
    wave_sync();
  }
  };
  if (q.gx_add != nullptr) run(std::true_type{});
  else run(std::false_type{});
}

template <int NOUTB, int KB, int TERMS>
__global__ __launch_bounds__(256) void lin_bwd_data_kernel(WideMulti<LinBwdDataParams> m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = wide_multi_find(m, blockIdx.x);
  lin_bwd_data_body<NOUTB, KB, TERMS>(m.p[k], blockIdx.x - m.first[k], m.first[k + 1] - m.first[k],
                                      smem);
}

template <typename P, typename TilesOf>
static unsigned wide_multi_grid(WideMulti<P>& m, TilesOf tiles_of) {
  // shares proportional to work, one round of the device in all (fused_common.h)
  int64_t rounds[NLAM_WIDE_MAXP], g[NLAM_WIDE_MAXP];
  for (int k = 0; k < m.n; ++k) rounds[k] = (tiles_of(m.p[k]) + 4 - 1) / 4;
  nlam_multi_shares(m.n, rounds, g, 256);
  m.first[0] = 0;
  for (int k = 0; k < m.n; ++k) m.first[k + 1] = m.first[k] + (int)g[k];
  for (int k = m.n; k < NLAM_WIDE_MAXP; ++k) m.first[k + 1] = m.first[m.n];
  return (unsigned)m.first[m.n];
}

template <int NOUTB, int KB, int TERMS>
static int launch_lin_bwd_data(WideMulti<LinBwdDataParams>& m, hipStream_t s) {
  constexpr int NO = 32 * NOUTB, K = 32 * KB;
  constexpr int LDT = (NO > K ? NO : K) + 4;
  const size_t lds = (size_t)2 * NO * (K + 16) * sizeof(__bf16) + (size_t)4 * NLAM_TILE * LDT * sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "lin_bwd_data: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = lin_bwd_data_kernel<NOUTB, KB, TERMS>;
  NLAM_BIG_LDS(kern, "lin_bwd_data_kernel");
  const unsigned grid = wide_multi_grid(m, [](const LinBwdDataParams& q) {
    return ((q.rows + NLAM_TILE - 1) / NLAM_TILE) * q.B;
  });
  kern<<<grid, 256, lds, s>>>(m);
  NLAM_CHECK_LAUNCH("lin_bwd_data_kernel");
  return 0;
}

static int lin_bwd_data_fill(LinBwdDataParams& q, const float* gy, int64_t gy_bstride, int64_t gy_ld,
                             int n_out, const float* W, int64_t ldW, int k_in, float* gx,
                             int64_t gx_bstride, int64_t gx_ld, const float* gx_add,
                             int64_t ga_bstride, int64_t ga_ld, int64_t B, int64_t rows) {
  NLAM_REQUIRE(n_out == 128 && k_in == 128, "nlam_lin_bwd_data: shape %d x %d unsupported (128 x 128)",
               n_out, k_in);
  NLAM_REQUIRE(view_vec_ok(gy, gy_bstride, gy_ld, n_out) && view_vec_ok(gx, gx_bstride, gx_ld, k_in) &&
                   (gx_add == nullptr || view_vec_ok(gx_add, ga_bstride, ga_ld, k_in)),
               "nlam_lin_bwd_data: operand rows must be 16-byte aligned with pitch %% 4 == 0");
  q.gy = RowView{gy, gy_bstride, gy_ld, n_out};
  q.W = W; q.ldW = ldW; q.n_out = n_out; q.k_in = k_in;
  q.gx = gx; q.gx_bstride = gx_bstride; q.gx_ld = gx_ld;
  q.gx_add = gx_add; q.ga_bstride = ga_bstride; q.ga_ld = ga_ld;
  q.rows = rows; q.B = (int)B;
  return 0;
}

// the share rule of the multi-problem launches, callable without a device (tests/test_host_logic.py)
extern "C" int nlam_debug_multi_shares(int n, const int64_t* rounds, int64_t cap, int64_t* out) {
  NLAM_REQUIRE(n >= 1 && n <= NLAM_WIDE_MAXP && rounds != nullptr && out != nullptr && cap >= 1,
               "nlam_debug_multi_shares: n %d out of [1, %d]", n, NLAM_WIDE_MAXP);
  nlam_multi_shares(n, rounds, out, cap);
  return 0;
}

extern "C" int nlam_lin_bwd_data(const float* gy, int64_t gy_bstride, int64_t gy_ld, int n_out,
                                 const float* W, int64_t ldW, int k_in,
                                 float* gx, int64_t gx_bstride, int64_t gx_ld,
                                 const float* gx_add, int64_t ga_bstride, int64_t ga_ld,
                                 int64_t B, int64_t rows, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(nlam_mfma_terms() != 0, "nlam_lin_bwd_data: needs NLAM_MFMA=bf16x3|bf16");
  if (n_out == 256 && k_in == 256)
    return nlam_fs_lin_bwd_data_256(gy, gy_bstride, gy_ld, W, ldW, gx, gx_bstride, gx_ld, gx_add,
                                    ga_bstride, ga_ld, B, rows, stream);
  WideMulti<LinBwdDataParams> m;
  m.n = 1;
  if (lin_bwd_data_fill(m.p[0], gy, gy_bstride, gy_ld, n_out, W, ldW, k_in, gx, gx_bstride, gx_ld,
                        gx_add, ga_bstride, ga_ld, B, rows))
    return 1;
  hipStream_t s = (hipStream_t)stream;
  return nlam_mfma_terms() == 3 ? launch_lin_bwd_data<4, 4, 3>(m, s)
                                : launch_lin_bwd_data<4, 4, 1>(m, s);
}

extern "C" int nlam_lin_bwd_data_multi(int n, int d, const float* const* gy, const int64_t* gy_bstride,
                                       const int64_t* gy_ld, const float* const* W,
                                       const int64_t* ldW, float* const* gx,
                                       const int64_t* gx_bstride, const int64_t* gx_ld,
                                       const float* const* gx_add, const int64_t* ga_bstride,
                                       const int64_t* ga_ld, const int64_t* B, const int64_t* rows,
                                       void* stream) {
  NLAM_REQUIRE(n >= 1 && n <= NLAM_WIDE_MAXP, "nlam_lin_bwd_data_multi: n %d out of [1, %d]", n,
               NLAM_WIDE_MAXP);
  NLAM_REQUIRE(nlam_mfma_terms() != 0, "nlam_lin_bwd_data_multi: needs NLAM_MFMA=bf16x3|bf16");
  if (d == 256)
    return nlam_fs_lin_bwd_data_multi_256(n, gy, gy_bstride, gy_ld, W, ldW, gx, gx_bstride, gx_ld,
                                          gx_add, ga_bstride, ga_ld, B, rows, stream);
  NLAM_REQUIRE(d == 128, "nlam_lin_bwd_data_multi: width %d unsupported (128, 256)", d);
  WideMulti<LinBwdDataParams> m;
  m.n = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    if (lin_bwd_data_fill(m.p[m.n], gy[k], gy_bstride[k], gy_ld[k], 128, W[k], ldW[k], 128, gx[k],
                          gx_bstride[k], gx_ld[k], gx_add[k], ga_bstride[k], ga_ld[k], B[k], rows[k]))
      return 1;
    ++m.n;
  }
  if (m.n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  return nlam_mfma_terms() == 3 ? launch_lin_bwd_data<4, 4, 3>(m, s)
                                : launch_lin_bwd_data<4, 4, 1>(m, s);
}

// ================================================= streaming weight gradients ===
// dW (32 NGB x 32 NXB) = sum_rows G[r]^T (x) f(X[r]),  db = colsum(G);  f = silu when SILU_X
// (X is then the kept pre-activation h).  Slab per workgroup: [dW | db (32 NGB)].
struct WideOuterParams {
  RowView g;        // (B, rows, 32 NGB)
  RowView x;        // (B, rows, 32 NXB); batch-invariant x (bstride 0) allowed
  float* slab; int64_t slab_stride;
  int64_t rows; int B;
  int silu_x;
  int x_vec;        // x rows float4-loadable (else scalar staging: narrow static features)
};

template <int NGB, int NXB, int TERMS>
__device__ __forceinline__ void wide_outer_body(const WideOuterParams& q, int bid, int gdim,
                                                float* smem) {
  constexpr int NG = 32 * NGB, NX = 32 * NXB;
  constexpr int NV = (NG + 63) / 64;
  // plane pitch + 16 (8 banks mod 64): both operands are read TRANSPOSED here; with the usual
  // + 4 the 16 lanes of a ds_read_b64_tr_b16 group collide (fused_fs.hip, FsPlanes)
  constexpr int ldg = NG + 16, ldx = NX + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* TG = smem + wave * (NLAM_TILE * (ldg + ldx));
  float* TX = TG + NLAM_TILE * ldg;
  f32x16 dW[NGB][NXB];
#pragma unroll
  for (int i = 0; i < NGB; ++i)
#pragma unroll
    for (int j = 0; j < NXB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW[i][j][r] = 0.f;
  float db[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) db[j] = 0.f;
  const int64_t tiles_per_b = (q.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * q.B;
  B3Tile TGp, TXp;
  TGp.pitch = ldg; TGp.hi = reinterpret_cast<__bf16*>(TG); TGp.lo = TGp.hi + NLAM_TILE * ldg;
  TXp.pitch = ldx; TXp.hi = reinterpret_cast<__bf16*>(TX); TXp.lo = TXp.hi + NLAM_TILE * ldx;
  const bool silu_x = q.silu_x != 0;
  constexpr int NVX = NXB >= 4 ? 4 * NXB : 8;   // float4 per lane that cover 32 rows of x
  bool x_vec = true;   // (128-wide x: always float4 rows, checked on the host)
  if constexpr (NXB < 4) x_vec = q.x_vec != 0;
  RowView xv = q.x;
  if constexpr (NXB >= 4) xv.width = NX;   // compile-time row mapping for the 128-wide form
  // The rows of tile n + 1 are requested as soon as tile n's registers are staged, so their
  // round trip rides under tile n's MFMAs (one wave per SIMD: nobody else hides it).
  f32x4 vg[4 * NGB];
  f32x4 vx[NVX];
  auto request = [&](int64_t task) {
    const int64_t tq = task < ntiles ? task : ntiles - 1;
    const int64_t b = tq / tiles_per_b;
    const int64_t r0 = (tq - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((q.rows - r0) < NLAM_TILE ? (q.rows - r0) : NLAM_TILE);
    view_load_v<4 * NGB>(vg, q.g, b, r0, nrows, lane);
    if (x_vec) view_load_v<NVX>(vx, xv, b, r0, nrows, lane);
  };
  const int64_t tstride = (int64_t)gdim * 4;
  request((int64_t)bid * 4 + wave);
  for (int64_t tt = (int64_t)bid * 4 + wave; tt < ntiles; tt += tstride) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((q.rows - r0) < NLAM_TILE ? (q.rows - r0) : NLAM_TILE);
    put_rows_v_b3<4 * NGB>(TGp, 0, NG, nrows, lane, vg);
    if (x_vec) {
      if (silu_x) {
#pragma unroll
        for (int k = 0; k < NVX; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) vx[k][j] = nlam_silu(vx[k][j]);
      }
      put_rows_v_b3<NVX>(TXp, 0, xv.width, nrows, lane, vx);
      if (NXB < 4 && q.x.width < NX) {   // zero the K padding (dropped columns, kept finite)
        const int padw = NX - q.x.width;
        for (int idx = lane; idx < NLAM_TILE * padw; idx += 64) {
          const int tr = idx / padw, cc = q.x.width + idx - tr * padw;
          TXp.hi[tr * TXp.pitch + cc] = (__bf16)0.f;
          TXp.lo[tr * TXp.pitch + cc] = (__bf16)0.f;
        }
      }
    } else if constexpr (NXB < 4) {
      const float* xb = q.x.ptr + b * q.x.bstride + r0 * q.x.ld;
      for (int idx = lane; idx < NLAM_TILE * NX; idx += 64) {
        const int tr = idx / NX, cc = idx - tr * NX;
        float v = (tr < nrows && cc < q.x.width) ? xb[(int64_t)tr * q.x.ld + cc] : 0.f;
        if (silu_x) v = nlam_silu(v);
        const __bf16 hi = (__bf16)v;
        TXp.hi[tr * TXp.pitch + cc] = hi;
        TXp.lo[tr * TXp.pitch + cc] = (__bf16)(v - (float)hi);
      }
    }
    if (tt + tstride < ntiles) request(tt + tstride);   // (nothing for a wave's last tile)
    wave_sync();
    tile_colsum_b3<NV, TERMS>(db, TGp, 0, lane);
    outer_accum_b3<NGB, NXB, TERMS>(dW, TGp, 0, TXp, 0, lane);
    wave_sync();
  }
  __syncthreads();
  float* slab = q.slab + (int64_t)bid * q.slab_stride;
  fold_block_rows_to_slab<NGB, NXB>(dW, smem, NX, slab, tid, wave, lane);
  fold_vec_lds<NV>(db, smem, wave, lane);
  for (int i = tid; i < NG; i += 256) slab[NG * NX + i] = smem[i];
}

template <int NGB, int NXB, int TERMS>
__global__ __launch_bounds__(256) void wide_outer_kernel(WideMulti<WideOuterParams, NLAM_WIDE_MAXP_OUTER> m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = wide_multi_find(m, blockIdx.x);
  wide_outer_body<NGB, NXB, TERMS>(m.p[k], blockIdx.x - m.first[k], m.first[k + 1] - m.first[k],
                                   smem);
}

// counts: slabs (= workgroups) per problem; NULL = nlam_bwd_grid(tiles) each (single launches)
template <int NGB, int NXB, int TERMS>
static int launch_wide_outer(WideMulti<WideOuterParams, NLAM_WIDE_MAXP_OUTER>& m, hipStream_t s,
                             const int* counts = nullptr) {
  constexpr int NG = 32 * NGB, NX = 32 * NXB;
  size_t lds = (size_t)4 * NLAM_TILE * (NG + 16 + NX + 16) * sizeof(float);
  const size_t fold = (size_t)4 * 32 * NX * sizeof(float);
  if (fold > lds) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "wide_outer: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = wide_outer_kernel<NGB, NXB, TERMS>;
  NLAM_BIG_LDS(kern, "wide_outer_kernel");
  // every problem's slab count is what the host sized its slab buffer for
  m.first[0] = 0;
  for (int k = 0; k < m.n; ++k)
    m.first[k + 1] = m.first[k] + (counts ? counts[k] : (int)wide_grid(((m.p[k].rows + NLAM_TILE - 1) / NLAM_TILE) * m.p[k].B));
  for (int k = m.n; k < NLAM_WIDE_MAXP_OUTER; ++k) m.first[k + 1] = m.first[m.n];
  kern<<<(unsigned)m.first[m.n], 256, lds, s>>>(m);
  NLAM_CHECK_LAUNCH("wide_outer_kernel");
  return 0;
}

static int wide_outer_fill(WideOuterParams& q, const float* g, int64_t g_bstride, int64_t g_ld, int ng,
                           const float* x, int64_t x_bstride, int64_t x_ld, int nx, int silu_x,
                           float* slab, int64_t slab_stride, int64_t B, int64_t rows) {
  NLAM_REQUIRE((ng == 128 || ng == 32) && nx >= 1 && (nx <= 64 || nx == 128) && (ng == 128 || nx == 128),
               "nlam_wide_outer: shape %d x %d unsupported", ng, nx);
  NLAM_REQUIRE(view_vec_ok(g, g_bstride, g_ld, ng) && x != nullptr && x_ld >= nx,
               "nlam_wide_outer: g rows must be 16-byte aligned with pitch %% 4 == 0");
  const int nxp = (nx + 31) & ~31;
  NLAM_REQUIRE(slab != nullptr && slab_stride >= (int64_t)ng * nxp + ng, "nlam_wide_outer: slab too small");
  q.g = RowView{g, g_bstride, g_ld, ng};
  q.x = RowView{x, x_bstride, x_ld, nx};
  q.slab = slab; q.slab_stride = slab_stride; q.rows = rows; q.B = (int)B; q.silu_x = silu_x;
  q.x_vec = view_vec_ok(x, x_bstride, x_ld, nx) ? 1 : 0;
  NLAM_REQUIRE(nx < 128 || q.x_vec, "nlam_wide_outer: 128-wide x rows must be 16-byte aligned");
  return 0;
}

extern "C" int nlam_wide_outer(const float* g, int64_t g_bstride, int64_t g_ld, int ng,
                               const float* x, int64_t x_bstride, int64_t x_ld, int nx, int silu_x,
                               float* slab, int64_t slab_stride, int64_t B, int64_t rows,
                               int io_bf16, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(nlam_mfma_terms() != 0, "nlam_wide_outer: needs NLAM_MFMA=bf16x3|bf16");
  if (ng == 256 || nx == 256)
    return nlam_fs_outer_256(g, g_bstride, g_ld, ng, x, x_bstride, x_ld, nx, silu_x, slab, slab_stride,
                             B, rows, wide_grid(((rows + NLAM_TILE - 1) / NLAM_TILE) * B), io_bf16,
                             stream);
  NLAM_REQUIRE(io_bf16 == 0, "nlam_wide_outer: bf16 rows exist at hidden 256 only");
  WideMulti<WideOuterParams, NLAM_WIDE_MAXP_OUTER> m;
  m.n = 1;
  if (wide_outer_fill(m.p[0], g, g_bstride, g_ld, ng, x, x_bstride, x_ld, nx, silu_x, slab,
                      slab_stride, B, rows))
    return 1;
  hipStream_t s = (hipStream_t)stream;
  const bool t3 = nlam_mfma_terms() == 3;
  if (ng == 32) return t3 ? launch_wide_outer<1, 4, 3>(m, s) : launch_wide_outer<1, 4, 1>(m, s);
  if (nx <= 32) return t3 ? launch_wide_outer<4, 1, 3>(m, s) : launch_wide_outer<4, 1, 1>(m, s);
  if (nx <= 64) return t3 ? launch_wide_outer<4, 2, 3>(m, s) : launch_wide_outer<4, 2, 1>(m, s);
  return t3 ? launch_wide_outer<4, 4, 3>(m, s) : launch_wide_outer<4, 4, 1>(m, s);
}

static int wide_outer_multi_impl(int n, int d, const float* const* g, const int64_t* g_bstride,
                                 const int64_t* g_ld, const float* const* x,
                                 const int64_t* x_bstride, const int64_t* x_ld,
                                 const int32_t* silu_x, float* const* slab,
                                 const int64_t* slab_stride, const int64_t* B,
                                 const int64_t* rows, const int32_t* nslabs,
                                 const int32_t* io_bf16, void* stream, const int32_t* nx);

// all problems d x d
extern "C" int nlam_wide_outer_multi(int n, int d, const float* const* g, const int64_t* g_bstride,
                                     const int64_t* g_ld, const float* const* x,
                                     const int64_t* x_bstride, const int64_t* x_ld,
                                     const int32_t* silu_x, float* const* slab,
                                     const int64_t* slab_stride, const int64_t* B,
                                     const int64_t* rows, const int32_t* nslabs,
                                     const int32_t* io_bf16, void* stream) {
  return wide_outer_multi_impl(n, d, g, g_bstride, g_ld, x, x_bstride, x_ld, silu_x, slab, slab_stride, B,
                               rows, nslabs, io_bf16, stream, nullptr);
}
// all problems d x nx[k], nx[k] <= 64 (narrow first Linears: static-feature embedders, the grid
// embedder): every slab row has 32 ceil(max_k nx[k] / 32) columns
extern "C" int nlam_wide_outer_multi_nx(int n, int d, const float* const* g, const int64_t* g_bstride,
                                        const int64_t* g_ld, const float* const* x,
                                        const int64_t* x_bstride, const int64_t* x_ld,
                                        const int32_t* nx, float* const* slab,
                                        const int64_t* slab_stride, const int64_t* B,
                                        const int64_t* rows, const int32_t* nslabs, void* stream) {
  NLAM_REQUIRE(nx != nullptr && n >= 1 && n <= NLAM_WIDE_MAXP_OUTER, "nlam_wide_outer_multi_nx: bad n / nx");
  int32_t zeros[NLAM_WIDE_MAXP_OUTER] = {0};
  for (int k = 0; k < n; ++k)
    NLAM_REQUIRE(nx[k] >= 1 && nx[k] <= 64, "nlam_wide_outer_multi_nx: nx[%d] = %d out of [1, 64]", k, nx[k]);
  return wide_outer_multi_impl(n, d, g, g_bstride, g_ld, x, x_bstride, x_ld, zeros, slab, slab_stride, B,
                               rows, nslabs, nullptr, stream, nx);
}

static int wide_outer_multi_impl(int n, int d, const float* const* g, const int64_t* g_bstride,
                                 const int64_t* g_ld, const float* const* x,
                                 const int64_t* x_bstride, const int64_t* x_ld,
                                 const int32_t* silu_x, float* const* slab,
                                 const int64_t* slab_stride, const int64_t* B,
                                 const int64_t* rows, const int32_t* nslabs,
                                 const int32_t* io_bf16, void* stream, const int32_t* nx) {
  NLAM_REQUIRE(n >= 1 && n <= NLAM_WIDE_MAXP_OUTER, "nlam_wide_outer_multi: n %d out of [1, %d]", n,
               NLAM_WIDE_MAXP_OUTER);
  NLAM_REQUIRE(nlam_mfma_terms() != 0, "nlam_wide_outer_multi: needs NLAM_MFMA=bf16x3|bf16");
  for (int k = 0; k < n; ++k)
    NLAM_REQUIRE(nslabs != nullptr && nslabs[k] >= 1 && nslabs[k] <= 1024,
                 "nlam_wide_outer_multi: nslabs[%d] out of [1, 1024]", k);
  if (d == 256) {
    unsigned grid[NLAM_WIDE_MAXP_OUTER];
    for (int k = 0; k < n; ++k) grid[k] = (unsigned)nslabs[k];
    return nlam_fs_outer_multi_256(n, g, g_bstride, g_ld, x, x_bstride, x_ld, silu_x, slab,
                                   slab_stride, B, rows, grid, io_bf16, stream, nx);
  }
  for (int k = 0; k < n; ++k)
    NLAM_REQUIRE(io_bf16 == nullptr || io_bf16[k] == 0,
                 "nlam_wide_outer_multi: bf16 rows exist at hidden 256 only");
  NLAM_REQUIRE(d == 128, "nlam_wide_outer_multi: width %d unsupported (128, 256)", d);
  WideMulti<WideOuterParams, NLAM_WIDE_MAXP_OUTER> m;
  int counts[NLAM_WIDE_MAXP_OUTER];
  m.n = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    if (wide_outer_fill(m.p[m.n], g[k], g_bstride[k], g_ld[k], 128, x[k], x_bstride[k], x_ld[k],
                        nx ? nx[k] : 128, silu_x[k], slab[k], slab_stride[k], B[k], rows[k]))
      return 1;
    counts[m.n] = nslabs[k];
    ++m.n;
  }
  if (m.n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool t3 = nlam_mfma_terms() == 3;
  if (nx != nullptr) {
    int nxmax = 0;
    for (int k = 0; k < n; ++k) nxmax = nx[k] > nxmax ? nx[k] : nxmax;
    // (wide_outer_fill checked every slab against ITS nx; the kernel writes 32 NXB columns per row)
    const int nxp = nxmax <= 32 ? 32 : 64;
    for (int k = 0; k < n; ++k)
      NLAM_REQUIRE(slab_stride[k] >= (int64_t)128 * nxp + 128, "nlam_wide_outer_multi_nx: slab %d too small", k);
    if (nxmax <= 32) return t3 ? launch_wide_outer<4, 1, 3>(m, s, counts) : launch_wide_outer<4, 1, 1>(m, s, counts);
    return t3 ? launch_wide_outer<4, 2, 3>(m, s, counts) : launch_wide_outer<4, 2, 1>(m, s, counts);
  }
  return t3 ? launch_wide_outer<4, 4, 3>(m, s, counts) : launch_wide_outer<4, 4, 1>(m, s, counts);
}

// ===================================== projections (first Linear), several per launch ===
// out = x W^T + b with W 128 x 128 (the split-bf16 image of nlam_lin_fwd's wide form)
struct WideLinParams {
  RowView x;
  const float* W; int64_t ldW; const float* bias;
  float* out; int64_t out_bstride; int64_t out_ld;
  int64_t rows; int B;
};

template <int TERMS>
__device__ __forceinline__ void wide_lin_fwd_body(const WideLinParams& p, int bid, int gdim,
                                                  float* smem) {
  constexpr int K = 128, NO = 128, KB = 4, NOUTB = 4;
  constexpr int ldt = K + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const B3Image W = b3_image(smem, NO, K);
  float* bs = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + b3_image_bytes(NO, K));
  float* tile = bs + NO + wave * (NLAM_TILE * ldt);
  const int64_t tiles_per_b = (p.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * p.B;
  const int64_t tt0 = (int64_t)bid * 4 + wave;
  const int64_t tstride = (int64_t)gdim * 4;
  constexpr int NV = NO / 8;
  const int sub = lane >> 5, c4 = lane & 31;
  // rows of tile n + 1 requested as soon as tile n's registers are staged (see lin_bwd_data_body)
  f32x4 vx[4 * KB];
  auto request = [&](int64_t task) {
    const int64_t tq = task < ntiles ? task : ntiles - 1;
    const int64_t b = tq / tiles_per_b;
    const int64_t r0 = (tq - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((p.rows - r0) < NLAM_TILE ? (p.rows - r0) : NLAM_TILE);
    RowView xv = p.x;
    xv.width = K;   // (compile-time row mapping: no per-load branches)
    view_load_v<4 * KB>(vx, xv, b, r0, nrows, lane);
  };
  {   // weights, bias and the first tile's rows in flight together, then the LDS image
    static_assert(NO <= 256, "one vector entry per thread");
    VecLoads<1> lv;
    const float* const vsrc[1] = {p.bias};
    float* const vdst[1] = {bs};
    vecs_issue(lv, vsrc, NO, tid);
    WLoad16<(NO * K / 4 + 255) / 256> lw;
    w16_issue(lw, p.W, p.ldW, NO, K, NO, K, tid, 256);
    request(tt0);
    __builtin_amdgcn_sched_barrier(0);
    w16_commit(lw, W, 0, p.W, p.ldW, NO, K, NO, K, tid, 256);
    vecs_commit(lv, vdst, NO, tid);
  }
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);   // (nothing in flight at loop entry: lin_bwd_data_body)
  for (int64_t tt = tt0; tt < ntiles; tt += tstride) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((p.rows - r0) < NLAM_TILE ? (p.rows - r0) : NLAM_TILE);
    const B3Tile Xp = b3_tile(tile, K);
    put_rows_v_b3<4 * KB>(Xp, 0, K, nrows, lane, vx);
    if (tt + tstride < ntiles) request(tt + tstride);   // (nothing for a wave's last tile)
    wave_sync();
    f32x16 a[NOUTB];
    vec_to_acc<NOUTB>(a, bs, lane);
    gemm_tile_b3<NOUTB, KB, TERMS>(a, W, 0, Xp, 0, lane);
    wave_sync();
    acc_to_tile<NOUTB>(a, tile, ldt, lane);
    wave_sync();
    // unconditional stores; padded slots repeat the tile's last row (lin_bwd_data_body)
    float* ob = p.out + b * p.out_bstride + r0 * p.out_ld;
    const int last = nrows - 1;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int t = sub + 2 * k;
      const int tc = t < last ? t : last;
      reinterpret_cast<f32x4*>(ob + (int64_t)tc * p.out_ld)[c4] =
          *(reinterpret_cast<const f32x4*>(tile + tc * ldt) + c4);
    }
    wave_sync();
  }
}

template <int TERMS>
__global__ __launch_bounds__(256) void wide_lin_fwd_kernel(WideMulti<WideLinParams> m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = wide_multi_find(m, blockIdx.x);
  wide_lin_fwd_body<TERMS>(m.p[k], blockIdx.x - m.first[k], m.first[k + 1] - m.first[k], smem);
}

template <int TERMS>
static int launch_wide_lin_fwd(WideMulti<WideLinParams>& m, hipStream_t s) {
  const size_t lds = b3_image_bytes(128, 128) + 128 * sizeof(float) +
                     (size_t)4 * NLAM_TILE * 132 * sizeof(float);
  auto kern = wide_lin_fwd_kernel<TERMS>;
  NLAM_BIG_LDS(kern, "wide_lin_fwd_kernel");
  const unsigned grid = wide_multi_grid(m, [](const WideLinParams& q) {
    return ((q.rows + NLAM_TILE - 1) / NLAM_TILE) * q.B;
  });
  kern<<<grid, 256, lds, s>>>(m);
  NLAM_CHECK_LAUNCH("wide_lin_fwd_kernel");
  return 0;
}

int nlam_k16_lin_fwd_multi_c(int n, const float* const* x, const int64_t* x_bstride,
                             const int64_t* x_ld, const float* const* W, const int64_t* ldW,
                             const float* const* bias, float* const* out,
                             const int64_t* out_bstride, const int64_t* out_ld, const int64_t* B,
                             const int64_t* rows, void* stream);

extern "C" int nlam_lin_fwd_multi(int n, int d, const float* const* x, const int64_t* x_bstride,
                                  const int64_t* x_ld, const float* const* W, const int64_t* ldW,
                                  const float* const* bias, float* const* out,
                                  const int64_t* out_bstride, const int64_t* out_ld,
                                  const int64_t* B, const int64_t* rows, int out_bf16_mask,
                                  void* stream) {
  NLAM_REQUIRE(n >= 1 && n <= NLAM_WIDE_MAXP, "nlam_lin_fwd_multi: n %d out of [1, %d]", n,
               NLAM_WIDE_MAXP);
  NLAM_REQUIRE(out_bf16_mask == 0 || d == 256, "nlam_lin_fwd_multi: bf16 rows exist at hidden 256 only");
  NLAM_REQUIRE(nlam_mfma_terms() != 0, "nlam_lin_fwd_multi: needs NLAM_MFMA=bf16x3|bf16");
  if (d == 64)   // the 16-row hidden-64 kernels (fused16_mlp.hip)
    return nlam_k16_lin_fwd_multi_c(n, x, x_bstride, x_ld, W, ldW, bias, out, out_bstride, out_ld, B,
                                    rows, stream);
  if (d == 256)
    return nlam_fs_lin_fwd_multi_256(n, x, x_bstride, x_ld, W, ldW, bias, out, out_bstride, out_ld, B,
                                     rows, out_bf16_mask, stream);
  NLAM_REQUIRE(d == 128, "nlam_lin_fwd_multi: width %d unsupported (64, 128, 256)", d);
  WideMulti<WideLinParams> m;
  m.n = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    NLAM_REQUIRE(view_vec_ok(x[k], x_bstride[k], x_ld[k], 128) &&
                     view_vec_ok(out[k], out_bstride[k], out_ld[k], 128),
                 "nlam_lin_fwd_multi: operand rows must be 16-byte aligned, width 128");
    WideLinParams& q = m.p[m.n++];
    q.x = RowView{x[k], x_bstride[k], x_ld[k], 128};
    q.W = W[k]; q.ldW = ldW[k]; q.bias = bias[k];
    q.out = out[k]; q.out_bstride = out_bstride[k]; q.out_ld = out_ld[k];
    q.rows = rows[k]; q.B = (int)B[k];
  }
  if (m.n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  return nlam_mfma_terms() == 3 ? launch_wide_lin_fwd<3>(m, s) : launch_wide_lin_fwd<1>(m, s);
}
