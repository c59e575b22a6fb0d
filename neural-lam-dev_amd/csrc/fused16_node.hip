// Node-side kernels of a CHAIN of InteractionNets that share their mesh nodes (the reference's
// processor: models/graph_lam.py:51-57,88 chains processor_layers m2m InteractionNets on the same
// 6,561 rows; interaction_net.py:86-131 is one link).  Between two edge passes the reference runs,
// per node row, the aggregation MLP of layer l-1 (x' = x + LN(V2 silu(V1 [x | agg] + c1) + c2),
// interaction_net.py:112-115, utils.py:191-214) and then the sender / receiver halves of layer l's
// first edge-MLP Linear (interaction_net.py:121).  On 26 k rows every one of those launches is a
// single 16-row tile per wavefront -- a fixed launch + prologue + latency cost -- so the chain is
// fused per row here:
//   node_fwd16   x' and, from the same registers, P = [x' W1s^T | x' W1r^T + b1] of the NEXT layer;
//   node_bwd16   backward data pass of the same pair in reverse: gPs = sum over the node's
//                out-edges of the edge-MLP hidden gradient (the sender scatter of the reference's
//                autograd, as a gather over the sender lists), G = [gPs | gPr] [W1s; W1r] + g_res
//                (gradient on x_l), then the aggregation MLP of layer l-1 backward with G as its
//                output gradient -> g_res', g_agg', ga; second-layer / LayerNorm gradients to a slab;
//   node_outer16 every 128-wide weight gradient of the pair in one pass over stored rows:
//                workgroups of role A form dV1 = ga^T [x | agg] (layer l-1), role B
//                dWp = [gPs | gPr]^T x_l (layer l).
// hidden width 64, split-bf16 (fp32-grade) products; building blocks: fused16.h.
#include <cstdlib>

#include "fused16.h"
#include "fused_params.h"

#define K16_NW 8
#define K16_THREADS 512

// ======================================================================= forward
template <bool HAS_PROJ, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void node_fwd16_kernel(NodeFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  char* cur = smem16;
  const B3Image V1im = w16_image(cur, D, 2 * D);
  cur += w16_image_bytes(D, 2 * D);
  const B3Image V2im = w16_image(cur, D, D);
  cur += w16_image_bytes(D, D);
  const B3Image Wpim = w16_image(cur, 2 * D, D);
  if (HAS_PROJ) cur += w16_image_bytes(2 * D, D);
  float* c1s = reinterpret_cast<float*>(cur);
  float* c2s = c1s + D;
  float* gs = c2s + D;
  float* bs = gs + D;
  float* bps = bs + D;   // 2 D
  {   // every global load of the prologue in flight together (fused16.h, batched prologue loads)
    VLoad16 lv;
    const float* const vecs[8] = {p.c1, p.c2, p.gamma, p.beta, HAS_PROJ ? p.bA : nullptr,
                                  HAS_PROJ ? p.bB : nullptr, nullptr, nullptr};
    const int lens[8] = {D, D, D, D, D, D, 0, 0};
    v16_issue(lv, vecs, lens, tid);
    WLoad16<4> l1;
    WLoad16<2> l2, la, lb;
    w16_issue(l1, p.V1, p.ldV1, D, 2 * D, D, 2 * D, tid, K16_THREADS);
    w16_issue(l2, p.V2, p.ldV2, D, D, D, D, tid, K16_THREADS);
    if constexpr (HAS_PROJ) {
      w16_issue(la, p.WA, p.ldWA, D, D, D, D, tid, K16_THREADS);
      w16_issue(lb, p.WB, p.ldWB, D, D, D, D, tid, K16_THREADS);
    }
    v16_commit(lv, c1s, 6, tid);
    w16_commit(l1, V1im, 0, p.V1, p.ldV1, D, 2 * D, D, 2 * D, tid, K16_THREADS);
    w16_commit(l2, V2im, 0, p.V2, p.ldV2, D, D, D, D, tid, K16_THREADS);
    if constexpr (HAS_PROJ) {
      w16_commit(la, Wpim, 0, p.WA, p.ldWA, D, D, D, D, tid, K16_THREADS);
      w16_commit(lb, Wpim, D, p.WB, p.ldWB, D, D, D, D, tid, K16_THREADS);
    }
  }
  __syncthreads();
  const int64_t tiles_per_b = (p.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * p.B;
  for (int64_t tt = (int64_t)blockIdx.x * K16_NW + wave; tt < ntiles;
       tt += (int64_t)gridDim.x * K16_NW) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int nrows = (int)((p.rows - r0) < NLAM_T16 ? (p.rows - r0) : NLAM_T16);
    const bool valid = t < nrows;
    const int64_t row = r0 + (valid ? t : nrows - 1);
    f32x4 y[4];
    {
      f32x4 x[8], h[4];
      load_row16<4>(x, p.x.ptr + b * p.x.bstride + row * p.x.ld, lane);
      load_row16<4>(x + 4, p.agg.ptr + b * p.agg.bstride + row * p.agg.ld, lane);
      vec_to_acc16<4>(h, c1s, lane);
      gemm_acc16<4, 4, TERMS>(h, V1im, 0, 0, x, lane);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[fb][r] = nlam_silu(h[fb][r]);
      vec_to_acc16<4>(y, c2s, lane);
      gemm_acc16<4, 2, TERMS>(y, V2im, 0, 0, h, lane);
      ln16_apply<4>(y, gs, bs, lane);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) y[fb] += x[fb];
    }
    if (valid) store_row16<4>(p.xout + b * p.xo_bstride + (r0 + t) * p.xo_ld, y, lane);
    if constexpr (HAS_PROJ) {
      f32x4 pr[8];
      vec_to_acc16<8>(pr, bps, lane);
      gemm_acc16<8, 2, TERMS>(pr, Wpim, 0, 0, y, lane);
      if (valid) store_row16<8>(p.P + b * p.p_bstride + (r0 + t) * p.p_ld, pr, lane);
    }
  }
}

static int launch_node_fwd16(const NodeFwdParams& p, hipStream_t s) {
  constexpr int D = 64;
  const bool proj = p.P != nullptr;
  const size_t lds = w16_image_bytes(D, 2 * D) + w16_image_bytes(D, D) +
                     (proj ? w16_image_bytes(2 * D, D) : 0) + 6 * D * sizeof(float);
  const int64_t ntiles = ((p.rows + NLAM_T16 - 1) / NLAM_T16) * p.B;
  int64_t g = (ntiles + K16_NW - 1) / K16_NW;
  if (g > 256) g = 256;
  if (proj) {
    auto kern = node_fwd16_kernel<true, 3>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)g, K16_THREADS, lds, s>>>(p);
  } else {
    auto kern = node_fwd16_kernel<false, 3>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)g, K16_THREADS, lds, s>>>(p);
  }
  NLAM_CHECK_LAUNCH("node_fwd16_kernel");
  return 0;
}

// ============================================================= backward, data pass
// Diagnostic (NLAM_TIMELINE_NODE=1): s_memrealtime (100 MHz) of workgroup phases of the last
// nlam_node_bwd launch with a node update: start, weights in LDS, first tile's gather done, G
// formed, tile loop done, slab written -- per workgroup (first 256).
__device__ unsigned long long g_node_tl[256 * 8];
extern "C" int nlam_debug_node_timeline(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_node_tl), sizeof(unsigned long long) * 256 * 8) ==
                 hipSuccess ? 0 : 1;
}
#define NODE_TL(k)                                                                        \
  if constexpr (TL) {                                                                     \
    if (tid == 0 && blockIdx.x < 256) g_node_tl[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  }

template <bool HAS_A, int TERMS, bool TL = false>
__global__ __launch_bounds__(K16_THREADS, 2) void node_bwd16_kernel(NodeBwdParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  char* cur = smem16;
  // swizzled (transposed reads conflict-free, fused16.h) where the 160 KiB allow it: with the node
  // update below (HAS_A) the three images + planes leave 3 KB, enough for V2's 2 KB only
  constexpr bool WP_SWZ = !HAS_A;
  const B3Image Wpim = w16_image(cur, 2 * D, D, WP_SWZ);
  cur += w16_image_bytes(2 * D, D, WP_SWZ);
  const B3Image V1im = w16_image(cur, D, 2 * D);
  const B3Image V2im = w16_image(cur + w16_image_bytes(D, 2 * D), D, D, true);
  if (HAS_A) cur += w16_image_bytes(D, 2 * D) + w16_image_bytes(D, D, true);
  float* c1s = reinterpret_cast<float*>(cur);
  float* c2s = c1s + D;
  float* gs = c2s + D;
  cur += 3 * D * sizeof(float);
  char* mine = cur + wave * (2 * p16_bytes(D));
  const B3Tile Ts = p16_tile(mine, D), Tz = p16_tile(mine + p16_bytes(D), D);
  NODE_TL(0)

  {   // every global load of the prologue in flight together
    VLoad16 lv;
    const float* const vecs[8] = {HAS_A ? q.c1 : nullptr, HAS_A ? q.c2 : nullptr,
                                  HAS_A ? q.gamma : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int lens[8] = {D, D, D, 0, 0, 0, 0, 0};
    if constexpr (HAS_A) v16_issue(lv, vecs, lens, tid);
    WLoad16<4> l1;
    WLoad16<2> l2, la, lb;
    w16_issue(la, q.WA, q.ldWA, D, D, D, D, tid, K16_THREADS);
    w16_issue(lb, q.WB, q.ldWB, D, D, D, D, tid, K16_THREADS);
    if constexpr (HAS_A) {
      w16_issue(l1, q.V1, q.ldV1, D, 2 * D, D, 2 * D, tid, K16_THREADS);
      w16_issue(l2, q.V2, q.ldV2, D, D, D, D, tid, K16_THREADS);
      v16_commit(lv, c1s, 3, tid);
    }
    w16_commit(la, Wpim, 0, q.WA, q.ldWA, D, D, D, D, tid, K16_THREADS);
    w16_commit(lb, Wpim, D, q.WB, q.ldWB, D, D, D, D, tid, K16_THREADS);
    if constexpr (HAS_A) {
      w16_commit(l1, V1im, 0, q.V1, q.ldV1, D, 2 * D, D, 2 * D, tid, K16_THREADS);
      w16_commit(l2, V2im, 0, q.V2, q.ldV2, D, D, D, D, tid, K16_THREADS);
    }
  }
  __syncthreads();
  NODE_TL(1)

  f32x16 dV2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dV2[i][j][r] = 0.f;
  float dc2[1] = {0.f}, dgam[1] = {0.f}, dbet[1] = {0.f};

  const int64_t tiles_per_b = (q.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * q.B;
  for (int64_t tt = (int64_t)blockIdx.x * K16_NW + wave; tt < ntiles;
       tt += (int64_t)gridDim.x * K16_NW) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int nrows = (int)((q.rows - r0) < NLAM_T16 ? (q.rows - r0) : NLAM_T16);
    const bool valid = t < nrows;
    const int64_t row = r0 + (valid ? t : nrows - 1);
    // ---- layer l: gPs (sender gather), G = [gPs | gPr] [W1s; W1r] + g_res
    f32x4 G[4];
    {
      f32x4 gp[8];
      gather_sender_sum16(gp, q.gh + b * q.gh_bstride, q.csc_colptr, q.csc_eid, row, q.n_send,
                          valid, lane);
      NODE_TL(2)
      float* prow = q.gP + b * q.gp_bstride + opaque(row) * q.gp_ld;
      load_row16<4>(gp + 4, prow + D, lane);
      if (valid) store_row16<4>(prow, gp, lane);
      zero16<4>(G);
      gemm_acc16_wt<4, 4, TERMS>(G, Wpim, 0, 0, gp, lane);
      f32x4 ad[4];
      load_row16<4>(ad, q.g_res + b * q.gr_bstride + opaque(row) * q.gr_ld, lane);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) G[fb] += ad[fb];
    }
    __builtin_amdgcn_sched_barrier(0);
    NODE_TL(3)
    if constexpr (!HAS_A) {
      if (valid) store_row16<4>(q.gx_out + b * q.gx_bstride + (r0 + t) * q.gx_ld, G, lane);
    } else {
      mask16<4>(G, valid);   // padded rows: zero gradient, every sum below ignores them
      // ---- layer l-1: recompute h, s, z; LayerNorm backward; second-layer gradients
      f32x4 hkeep[4], g[4];
      {
        f32x4 hpre[4];
        {
          f32x4 x[8];
          const int64_t rw = opaque(row);
          load_row16<4>(x, q.x.ptr + b * q.x.bstride + rw * q.x.ld, lane);
          load_row16<4>(x + 4, q.agg.ptr + b * q.agg.bstride + rw * q.agg.ld, lane);
          vec_to_acc16<4>(hpre, c1s, lane);
          gemm_acc16<4, 4, TERMS>(hpre, V1im, 0, 0, x, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 sact[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {   // silu and silu' from one sigmoid; hkeep = silu'(h)
            float sv, dv;
            silu_both(hpre[fb][r], sv, dv);
            sact[fb][r] = sv;
            hkeep[fb][r] = dv;
          }
        acc16_to_planes<4, TERMS>(sact, Ts, 0, lane);
        f32x4 z[4];
        vec_to_acc16<4>(z, c2s, lane);
        gemm_acc16<4, 2, TERMS>(z, V2im, 0, 0, sact, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) g[fb] = G[fb];
        acc16_to_planes<4, TERMS>(g, Tz, 0, lane);
        wave_sync();
        colsum16_64<TERMS>(dbet[0], Tz, 0, lane);
        wave_sync();
        ln16_bwd<4, TERMS>(z, g, Tz, gs, lane);
        wave_sync();
        colsum16_64<TERMS>(dgam[0], Tz, 0, lane);
        wave_sync();
      }
      __builtin_amdgcn_sched_barrier(0);
      acc16_to_planes<4, TERMS>(g, Tz, 0, lane);
      wave_sync();
      outer_accum16_cs<2, 2, TERMS>(dV2, dc2, Tz, 0, Ts, 0, lane);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 ga[4];
      zero16<4>(ga);
      gemm_acc16_wt<4, 2, TERMS>(ga, V2im, 0, 0, g, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[fb][r] *= hkeep[fb][r];
      if (valid) store_row16<4>(q.ga_out + (b * q.rows + r0 + t) * D, ga, lane);
      f32x4 gx[8];
      zero16<8>(gx);
      gemm_acc16_wt<8, 2, TERMS>(gx, V1im, 0, 0, ga, lane);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) gx[fb] += G[fb];   // the residual x' = x + ...
      if (valid) {
        store_row16<4>(q.gx_out + b * q.gx_bstride + (r0 + t) * q.gx_ld, gx, lane);
        store_row16<4>(q.gagg_out + b * q.gagg_bstride + (r0 + t) * q.gagg_ld, gx + 4, lane);
      }
      wave_sync();   // the planes are rewritten by the next tile
    }
  }
  NODE_TL(4)
  if constexpr (HAS_A) {
    __syncthreads();
    float* img = reinterpret_cast<float*>(smem16);   // weights and planes are dead
    float* slab = q.slab + (int64_t)blockIdx.x * q.slab_stride;
    fold_blocks_to_slab16<2, 2, 2, K16_NW>(&dV2[0][0], img, D, slab, tid, wave, lane);
    const float v3[3] = {dc2[0], dgam[0], dbet[0]};   // contiguous in the slab: one pass
    fold_vec_to_slab16<3, K16_NW>(v3, img, slab + D * D, 3 * D, tid, wave, lane);
  }
  NODE_TL(5)
}

// slab floats per workgroup: [dV2 64 x 64 | dc2 | dgamma | dbeta]
extern "C" int64_t nlam_node_bwd_slab_stride(void) { return 64 * 64 + 3 * 64; }
// number of slabs of nlam_node_bwd (= its workgroups: one per 8 tiles, at most one per CU) and of
// nlam_node_outer (two workgroups -- one per role -- share a slab) for B x rows node rows
static int64_t node_grid(int64_t B, int64_t rows, int64_t cap) {
  const int64_t ntiles = ((rows + NLAM_T16 - 1) / NLAM_T16) * B;
  int64_t g = (ntiles + K16_NW - 1) / K16_NW;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return g;
}
extern "C" int64_t nlam_node_bwd_grid(int64_t B, int64_t rows) { return node_grid(B, rows, 256); }
extern "C" int64_t nlam_node_outer_grid(int64_t B, int64_t rows) { return node_grid(B, rows, 128); }

static int launch_node_bwd16(const NodeBwdParams& q, hipStream_t s) {
  constexpr int D = 64;
  const bool has_a = q.x.ptr != nullptr;
  size_t lds = w16_image_bytes(2 * D, D, !has_a) + 3 * D * sizeof(float) + (size_t)K16_NW * 2 * p16_bytes(D);
  if (has_a) lds += w16_image_bytes(D, 2 * D) + w16_image_bytes(D, D, true);
  const size_t fold = (size_t)K16_NW * D * D * sizeof(float);
  if (has_a && fold > lds) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "node_bwd16: LDS footprint %zu B exceeds 160 KiB", lds);
  const int64_t g = nlam_node_bwd_grid(q.B, q.rows);
  static const bool tl = getenv("NLAM_TIMELINE_NODE") != nullptr;
  if (has_a && tl) {
    auto kern = node_bwd16_kernel<true, 3, true>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)g, K16_THREADS, lds, s>>>(q);
  } else if (has_a) {
    auto kern = node_bwd16_kernel<true, 3>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)g, K16_THREADS, lds, s>>>(q);
  } else {
    auto kern = node_bwd16_kernel<false, 3>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)g, K16_THREADS, lds, s>>>(q);
  }
  NLAM_CHECK_LAUNCH("node_bwd16_kernel");
  return 0;
}

// ====================================================== backward, weight-gradient pass
// Role A (even workgroups when has_a): dV1 (64 x 128) = ga^T [x | agg], dc1 = colsum ga.
// Role B: dWp (128 x 64) = [gPs | gPr]^T x_l, dbp (128) = colsum [gPs | gPr].
// Slab (per role-pair index): [dV1 64 x 128 | dc1 64 | dWp 128 x 64 | dbp 128].
template <int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void node_outer16_kernel(NodeOuterParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  const int nrole = q.has_a ? 2 : 1;
  const int role = q.has_a ? (int)(blockIdx.x & 1) : 1;
  const int wg = (int)blockIdx.x / nrole, nwg = (int)gridDim.x / nrole;
  char* mine = smem16 + wave * (p16_bytes(D) + p16_bytes(2 * D));
  float* img = reinterpret_cast<float*>(smem16);
  float* slab = q.slab + (int64_t)wg * q.slab_stride;
  const int64_t tiles_per_b = (q.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * q.B;
  if (role == 0) {
    const B3Tile Tg = p16_tile(mine, D), Tx = p16_tile(mine + p16_bytes(D), 2 * D);
    f32x16 dW[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[i][j][r] = 0.f;
    float db[1] = {0.f};
    for (int64_t tt = (int64_t)wg * K16_NW + wave; tt < ntiles; tt += (int64_t)nwg * K16_NW) {
      const int64_t b = tt / tiles_per_b;
      const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
      const int nrows = (int)((q.rows - r0) < NLAM_T16 ? (q.rows - r0) : NLAM_T16);
      const bool valid = t < nrows;
      const int64_t row = r0 + (valid ? t : nrows - 1);
      f32x4 g[4], x[8];
      load_row16<4>(g, q.ga.ptr + b * q.ga.bstride + row * q.ga.ld, lane);
      load_row16<4>(x, q.xa.ptr + b * q.xa.bstride + row * q.xa.ld, lane);
      load_row16<4>(x + 4, q.xb.ptr + b * q.xb.bstride + row * q.xb.ld, lane);
      mask16<4>(g, valid);
      acc16_to_planes<4, TERMS>(g, Tg, 0, lane);
      acc16_to_planes<8, TERMS>(x, Tx, 0, lane);
      wave_sync();
      outer_accum16_cs<2, 4, TERMS>(dW, db, Tg, 0, Tx, 0, lane);
      wave_sync();
    }
    __syncthreads();
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
      fold_blocks_to_slab16<1, 4, 4, K16_NW>(&dW[ib][0], img, 2 * D, slab + 32 * ib * 2 * D, tid, wave, lane);
    fold_vec_to_slab16<1, K16_NW>(db, img, slab + D * 2 * D, D, tid, wave, lane);
  } else {
    const B3Tile Tg = p16_tile(mine + p16_bytes(D), 2 * D), Tx = p16_tile(mine, D);
    f32x16 dW[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[i][j][r] = 0.f;
    float db[2] = {0.f, 0.f};
    for (int64_t tt = (int64_t)wg * K16_NW + wave; tt < ntiles; tt += (int64_t)nwg * K16_NW) {
      const int64_t b = tt / tiles_per_b;
      const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
      const int nrows = (int)((q.rows - r0) < NLAM_T16 ? (q.rows - r0) : NLAM_T16);
      const bool valid = t < nrows;
      const int64_t row = r0 + (valid ? t : nrows - 1);
      f32x4 g[8], x[4];
      load_row16<8>(g, q.gP.ptr + b * q.gP.bstride + row * q.gP.ld, lane);
      load_row16<4>(x, q.xl.ptr + b * q.xl.bstride + row * q.xl.ld, lane);
      mask16<8>(g, valid);
      acc16_to_planes<8, TERMS>(g, Tg, 0, lane);
      acc16_to_planes<4, TERMS>(x, Tx, 0, lane);
      wave_sync();
      outer_accum16_cs<4, 2, TERMS>(dW, db, Tg, 0, Tx, 0, lane);
      wave_sync();
    }
    __syncthreads();
    float* sb = slab + D * 2 * D + D;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
      fold_blocks_to_slab16<2, 2, 2, K16_NW>(&dW[2 * h2][0], img, D, sb + 64 * h2 * D, tid, wave, lane);
    fold_vec_to_slab16<2, K16_NW>(db, img, sb + 2 * D * D, 2 * D, tid, wave, lane);
  }
}

extern "C" int64_t nlam_node_outer_slab_stride(void) { return 64 * 128 + 64 + 128 * 64 + 128; }

static int launch_node_outer16(const NodeOuterParams& q, hipStream_t s) {
  constexpr int D = 64;
  size_t lds = (size_t)K16_NW * (p16_bytes(D) + p16_bytes(2 * D));
  const size_t fold = (size_t)K16_NW * 64 * 64 * sizeof(float);    // role B: 64 rows x 64; A: 32 x 128
  if (fold > lds) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "node_outer16: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = node_outer16_kernel<3>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t g = nlam_node_outer_grid(q.B, q.rows) * (q.has_a ? 2 : 1);
  kern<<<(unsigned)g, K16_THREADS, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("node_outer16_kernel");
  return 0;
}

// ==================================================================== C entry points
static bool node_view_ok(const float* p, int64_t bstride, int64_t ld) {
  return view_vec_ok(p, bstride, ld, 64);
}

extern "C" int nlam_node_chain_supported(void) {
  return nlam_mfma_b3() && nlam_k16_on(K16_NODE_CHAIN) ? 1 : 0;
}

extern "C" int nlam_node_fwd(const float* x, int64_t x_bstride, int64_t x_ld, const float* agg,
                             int64_t agg_bstride, int64_t agg_ld, const float* V1, int64_t ldV1,
                             const float* c1, const float* V2, int64_t ldV2, const float* c2,
                             const float* gamma, const float* beta, float* xout, int64_t xo_bstride,
                             int64_t xo_ld, const float* WA, int64_t ldWA, const float* bA,
                             const float* WB, int64_t ldWB, const float* bB, float* P,
                             int64_t p_bstride, int64_t p_ld, int64_t B, int64_t rows, void* stream) {
  NLAM_REQUIRE(nlam_node_chain_supported(), "nlam_node_fwd: needs the split-bf16 MFMA mode");
  NLAM_REQUIRE(x && agg && V1 && V2 && c1 && c2 && gamma && beta && xout, "nlam_node_fwd: NULL operand");
  NLAM_REQUIRE(B >= 1 && rows >= 1, "nlam_node_fwd: empty input");
  NLAM_REQUIRE(node_view_ok(x, x_bstride, x_ld) && node_view_ok(agg, agg_bstride, agg_ld) &&
                   node_view_ok(xout, xo_bstride, xo_ld),
               "nlam_node_fwd: rows must be 16-byte aligned, 64 wide");
  NLAM_REQUIRE(P == nullptr || (WA && WB && view_vec_ok(P, p_bstride, p_ld, 128)),
               "nlam_node_fwd: projection operands");
  NodeFwdParams p;
  p.x = RowView{x, x_bstride, x_ld, 64};
  p.agg = RowView{agg, agg_bstride, agg_ld, 64};
  p.V1 = V1; p.ldV1 = ldV1; p.c1 = c1; p.V2 = V2; p.ldV2 = ldV2; p.c2 = c2;
  p.gamma = gamma; p.beta = beta;
  p.xout = xout; p.xo_bstride = xo_bstride; p.xo_ld = xo_ld;
  p.WA = WA; p.ldWA = ldWA; p.bA = bA; p.WB = WB; p.ldWB = ldWB; p.bB = bB;
  p.P = P; p.p_bstride = p_bstride; p.p_ld = p_ld;
  p.rows = rows; p.B = (int)B;
  return launch_node_fwd16(p, (hipStream_t)stream);
}

extern "C" int nlam_node_bwd(const float* gh, int64_t gh_bstride, const int32_t* csc_colptr,
                             const int32_t* csc_eid, int64_t n_send, float* gP, int64_t gp_bstride,
                             int64_t gp_ld, const float* g_res, int64_t gr_bstride, int64_t gr_ld,
                             const float* WA, int64_t ldWA, const float* WB, int64_t ldWB,
                             const float* x, int64_t x_bstride, int64_t x_ld, const float* agg,
                             int64_t agg_bstride, int64_t agg_ld, const float* V1, int64_t ldV1,
                             const float* c1, const float* V2, int64_t ldV2, const float* c2,
                             const float* gamma, float* gx_out, int64_t gx_bstride, int64_t gx_ld,
                             float* gagg_out, int64_t gagg_bstride, int64_t gagg_ld, float* ga_out,
                             float* slab, int64_t slab_stride, int64_t B, int64_t rows, void* stream) {
  NLAM_REQUIRE(nlam_node_chain_supported(), "nlam_node_bwd: needs the split-bf16 MFMA mode");
  NLAM_REQUIRE(gh && csc_colptr && csc_eid && gP && g_res && WA && WB && gx_out,
               "nlam_node_bwd: NULL operand");
  NLAM_REQUIRE(B >= 1 && rows >= 1 && n_send >= 0 && n_send <= rows, "nlam_node_bwd: bad sizes");
  NLAM_REQUIRE(nlam_aligned16(gh) && gh_bstride % 4 == 0 && view_vec_ok(gP, gp_bstride, gp_ld, 128) &&
                   node_view_ok(g_res, gr_bstride, gr_ld) && node_view_ok(gx_out, gx_bstride, gx_ld),
               "nlam_node_bwd: rows must be 16-byte aligned");
  NodeBwdParams q;
  q.gh = gh; q.gh_bstride = gh_bstride; q.csc_colptr = csc_colptr; q.csc_eid = csc_eid;
  q.n_send = (int)n_send;
  q.gP = gP; q.gp_bstride = gp_bstride; q.gp_ld = gp_ld;
  q.g_res = g_res; q.gr_bstride = gr_bstride; q.gr_ld = gr_ld;
  q.WA = WA; q.ldWA = ldWA; q.WB = WB; q.ldWB = ldWB;
  q.x = RowView{x, x_bstride, x_ld, 64};
  q.agg = RowView{agg, agg_bstride, agg_ld, 64};
  q.V1 = V1; q.ldV1 = ldV1; q.c1 = c1; q.V2 = V2; q.ldV2 = ldV2; q.c2 = c2; q.gamma = gamma;
  q.gx_out = gx_out; q.gx_bstride = gx_bstride; q.gx_ld = gx_ld;
  q.gagg_out = gagg_out; q.gagg_bstride = gagg_bstride; q.gagg_ld = gagg_ld;
  q.ga_out = ga_out; q.slab = slab; q.slab_stride = slab_stride;
  q.rows = rows; q.B = (int)B;
  if (x != nullptr) {
    NLAM_REQUIRE(agg && V1 && c1 && V2 && c2 && gamma && gagg_out && ga_out && slab,
                 "nlam_node_bwd: NULL operand of the node update");
    NLAM_REQUIRE(node_view_ok(x, x_bstride, x_ld) && node_view_ok(agg, agg_bstride, agg_ld) &&
                     node_view_ok(gagg_out, gagg_bstride, gagg_ld) && nlam_aligned16(ga_out),
                 "nlam_node_bwd: rows must be 16-byte aligned");
    NLAM_REQUIRE(slab_stride >= nlam_node_bwd_slab_stride(), "nlam_node_bwd: slab too small");
  }
  return launch_node_bwd16(q, (hipStream_t)stream);
}

extern "C" int nlam_node_outer(const float* ga, const float* xa, int64_t xa_bstride, int64_t xa_ld,
                               const float* xb, int64_t xb_bstride, int64_t xb_ld, const float* gP,
                               int64_t gp_bstride, int64_t gp_ld, const float* xl,
                               int64_t xl_bstride, int64_t xl_ld, float* slab, int64_t slab_stride,
                               int64_t B, int64_t rows, void* stream) {
  NLAM_REQUIRE(nlam_node_chain_supported(), "nlam_node_outer: needs the split-bf16 MFMA mode");
  NLAM_REQUIRE(gP && xl && slab && B >= 1 && rows >= 1, "nlam_node_outer: NULL operand");
  NLAM_REQUIRE(view_vec_ok(gP, gp_bstride, gp_ld, 128) && node_view_ok(xl, xl_bstride, xl_ld),
               "nlam_node_outer: rows must be 16-byte aligned");
  NLAM_REQUIRE(slab_stride >= nlam_node_outer_slab_stride(), "nlam_node_outer: slab too small");
  NodeOuterParams q;
  q.has_a = ga != nullptr;
  if (q.has_a) {
    NLAM_REQUIRE(xa && xb && nlam_aligned16(ga) && node_view_ok(xa, xa_bstride, xa_ld) &&
                     node_view_ok(xb, xb_bstride, xb_ld),
                 "nlam_node_outer: operands of the node update");
  }
  q.ga = RowView{ga, rows * 64, 64, 64};
  q.xa = RowView{xa, xa_bstride, xa_ld, 64};
  q.xb = RowView{xb, xb_bstride, xb_ld, 64};
  q.gP = RowView{gP, gp_bstride, gp_ld, 128};
  q.xl = RowView{xl, xl_bstride, xl_ld, 64};
  q.slab = slab; q.slab_stride = slab_stride; q.rows = rows; q.B = (int)B;
  return launch_node_outer16(q, (hipStream_t)stream);
}
