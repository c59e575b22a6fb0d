// Fused edge kernels (gfx950): per-edge MLP + LayerNorm + receiver aggregation
// of one InteractionNet layer (interaction_net.py:102-131) over receiver-sorted
// (CSR) edge tiles.
//
//   h_k   = [W1e e_k  or  Pe_k] + Ps[send(k)] + Pr[rec(k)]        (Pr carries b1)
//   m_k   = LN(W2 silu(h_k) + b2)
//   agg_i = scale_i * sum_{k: rec(k) = i} m_k                      (sum / mean)
//   e'_k  = e_k + m_k                                              (update_edges)
//
// Ps / Pr are the node-side projections of the first edge-MLP layer (computed
// once per node by nlam_lin_fwd), so per edge only the d x d GEMMs remain.  A
// tile holds <= 32 edges that are whole in-edge segments of consecutive
// receivers (nlam_graph_tiles_host): the segmented reduction is tile-local, has
// a fixed order and needs no atomics.  Rows are gathered / scattered by index
// but always moved as whole rows (coalesced 16 B per lane).
#include "fused_common.h"

struct EdgeFwdParams {
  // graph tables (device)
  const int32_t* tiles;       // (ntiles, 4): p0, p1, r0, r1
  int64_t ntiles;
  const int32_t* csr_rowptr;  // n_rec + 1
  const int32_t* csr_eid;     // original edge id at CSR position
  const int32_t* csr_send;
  const int32_t* csr_rec;
  const float* inv_deg;       // n_rec or NULL
  // operands
  RowView e;                  // HAS_EGEMM: edge reps (B, M, d) in original order;
                              // else: Pe (1 or B, M, d) in original order
  RowView ps;                 // (B, N_s, d)
  RowView pr;                 // (B, N_r, d)
  const float* W1e; int64_t ldW1e;
  const float* W2; int64_t ldW2; const float* b2;
  const float* gamma; const float* beta;
  float* agg; int64_t agg_bstride; int64_t agg_ld;
  float* e_out; int64_t eo_bstride; int64_t eo_ld;   // HAS_EGEMM only
  int B;
};

template <int D, bool HAS_EGEMM>
__global__ __launch_bounds__(256) void edge_fwd_kernel(EdgeFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NB = D / 32;
  constexpr int LDW = D + 4, LDT = D + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* W1s = smem;                                   // HAS_EGEMM only
  float* W2s = W1s + (HAS_EGEMM ? D * LDW : 0);
  float* b2s = W2s + D * LDW;
  float* gs = b2s + D;
  float* bs = gs + D;
  float* tile = bs + D + wave * (NLAM_TILE * LDT);
  if (HAS_EGEMM) load_weight_lds(W1s, p.W1e, p.ldW1e, D, D, D, D, tid, 256);
  load_weight_lds(W2s, p.W2, p.ldW2, D, D, D, D, tid, 256);
  load_vec_lds(b2s, p.b2, D, D, tid, 256);
  load_vec_lds(gs, p.gamma, D, D, tid, 256);
  load_vec_lds(bs, p.beta, D, D, tid, 256);
  __syncthreads();

  const int64_t total = p.ntiles * p.B;
  const int t = lane & 31;
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < total; tt += (int64_t)gridDim.x * 4) {
    const int64_t b = tt / p.ntiles;
    const int64_t ti = tt - b * p.ntiles;
    const int4 tl = reinterpret_cast<const int4*>(p.tiles)[ti];
    const int p0 = tl.x, ne = tl.y - tl.x, r0 = tl.z, nr = tl.w - tl.z;
    // per-slot indices: lane (t, *) holds those of slot t
    int eid = 0, snd = 0, rcv = 0;
    if (t < ne) {
      eid = p.csr_eid[p0 + t];
      snd = p.csr_send[p0 + t];
      rcv = p.csr_rec[p0 + t];
    }
    const float* eb = p.e.ptr + b * p.e.bstride;
    const float* psb = p.ps.ptr + b * p.ps.bstride;
    const float* prb = p.pr.ptr + b * p.pr.bstride;
    auto e_row = [&](int s) { return eb + (int64_t)__shfl(eid, s, 64) * p.e.ld; };
    auto ps_row = [&](int s) { return psb + (int64_t)__shfl(snd, s, 64) * p.ps.ld; };
    auto pr_row = [&](int s) { return prb + (int64_t)__shfl(rcv, s, 64) * p.pr.ld; };

    f32x16 a1[NB], ebuf[NB];
    if (HAS_EGEMM) {
      stage_rows<true, false>(tile, LDT, 0, D, ne, lane, e_row);
      wave_sync();
      tile_to_acc<NB>(ebuf, tile, LDT, lane);
      wave_sync();
      stage_rows<true, false>(tile, LDT, 0, D, ne, lane, ps_row);
    } else {
      stage_rows<true, false>(tile, LDT, 0, D, ne, lane, e_row);   // Pe rows
      wave_sync();
      stage_rows<true, true>(tile, LDT, 0, D, ne, lane, ps_row);
    }
    wave_sync();
    stage_rows<true, true>(tile, LDT, 0, D, ne, lane, pr_row);
    wave_sync();
    tile_to_acc<NB>(a1, tile, LDT, lane);
    if (HAS_EGEMM) gemm_acc<NB, NB>(a1, W1s, LDW, 0, ebuf, lane);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[nb][r] = nlam_silu(a1[nb][r]);
    f32x16 m[NB];
    vec_to_acc<NB>(m, b2s, lane);
    gemm_acc<NB, NB>(m, W2s, LDW, 0, a1, lane);
    ln_apply<NB>(m, gs, bs, lane);

    // messages -> LDS; segmented reduction with lanes = features
    wave_sync();
    acc_to_tile<NB>(m, tile, LDT, lane);
    wave_sync();
    {
      int rp = 0;
      if (lane <= nr) rp = p.csr_rowptr[r0 + lane] - p0;
      float* ab = p.agg + b * p.agg_bstride;
      for (int i = 0; i < nr; ++i) {
        const int beg = __shfl(rp, i, 64), end = __shfl(rp, i + 1, 64);
        const float sc = p.inv_deg ? p.inv_deg[r0 + i] : 1.0f;
#pragma unroll
        for (int f0 = 0; f0 < D; f0 += 64) {
          float acc = 0.f;
          for (int s = beg; s < end; ++s) acc += tile[s * LDT + f0 + lane];
          ab[(int64_t)(r0 + i) * p.agg_ld + f0 + lane] = acc * sc;
        }
      }
    }
    if (HAS_EGEMM) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) ebuf[nb][r] += m[nb][r];
      wave_sync();
      acc_to_tile<NB>(ebuf, tile, LDT, lane);
      wave_sync();
      float* ob = p.e_out + b * p.eo_bstride;
      auto o_row = [&](int s) { return ob + (int64_t)__shfl(eid, s, 64) * p.eo_ld; };
      store_rows<true>(tile, LDT, 0, D, ne, lane, o_row);
    }
    wave_sync();
  }
}

template <int D, bool HAS_EGEMM>
static int launch_edge_fwd(const EdgeFwdParams& p, hipStream_t s) {
  const size_t lds = ((size_t)(HAS_EGEMM ? 2 : 1) * D * (D + 4) + 3 * D +
                      (size_t)4 * NLAM_TILE * (D + 4)) * sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "edge_fwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = edge_fwd_kernel<D, HAS_EGEMM>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  kern<<<persistent_grid(p.ntiles * p.B, lds), 256, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("edge_fwd_kernel");
  return 0;
}

static bool rows_vec_ok(const float* ptr, int64_t bstride, int64_t ld, int d) {
  return view_vec_ok(ptr, bstride, ld, d);
}

extern "C" int nlam_edge_fwd(
    const int32_t* tiles, int64_t ntiles, const int32_t* csr_rowptr, const int32_t* csr_eid,
    const int32_t* csr_send, const int32_t* csr_rec, const float* inv_deg,
    const float* e, int64_t e_bstride, int64_t e_ld, int has_egemm,
    const float* ps, int64_t ps_bstride, int64_t ps_ld,
    const float* pr, int64_t pr_bstride, int64_t pr_ld,
    const float* W1e, int64_t ldW1e, const float* W2, int64_t ldW2, const float* b2,
    const float* gamma, const float* beta,
    float* agg, int64_t agg_bstride, int64_t agg_ld,
    float* e_out, int64_t eo_bstride, int64_t eo_ld,
    int64_t B, int d, void* stream) {
  if (B <= 0 || ntiles <= 0) return 0;
  NLAM_REQUIRE(d == 64 || d == 128, "nlam_edge_fwd: d=%d not in {64,128}", d);
  NLAM_REQUIRE(rows_vec_ok(e, e_bstride, e_ld, d) && rows_vec_ok(ps, ps_bstride, ps_ld, d) &&
                   rows_vec_ok(pr, pr_bstride, pr_ld, d),
               "nlam_edge_fwd: operand rows must be 16-byte aligned with pitch %% 4 == 0");
  NLAM_REQUIRE(agg != nullptr && agg_ld >= d, "nlam_edge_fwd: bad agg view");
  if (has_egemm) {
    NLAM_REQUIRE(W1e != nullptr && e_out != nullptr && rows_vec_ok(e_out, eo_bstride, eo_ld, d),
                 "nlam_edge_fwd: update_edges needs W1e and a 16-byte aligned e_out");
  }
  EdgeFwdParams p;
  p.tiles = tiles; p.ntiles = ntiles; p.csr_rowptr = csr_rowptr; p.csr_eid = csr_eid;
  p.csr_send = csr_send; p.csr_rec = csr_rec; p.inv_deg = inv_deg;
  p.e = RowView{e, e_bstride, e_ld, d};
  p.ps = RowView{ps, ps_bstride, ps_ld, d};
  p.pr = RowView{pr, pr_bstride, pr_ld, d};
  p.W1e = W1e; p.ldW1e = ldW1e; p.W2 = W2; p.ldW2 = ldW2; p.b2 = b2;
  p.gamma = gamma; p.beta = beta;
  p.agg = agg; p.agg_bstride = agg_bstride; p.agg_ld = agg_ld;
  p.e_out = e_out; p.eo_bstride = eo_bstride; p.eo_ld = eo_ld;
  p.B = (int)B;
  hipStream_t s = (hipStream_t)stream;
  if (d == 64) return has_egemm ? launch_edge_fwd<64, true>(p, s) : launch_edge_fwd<64, false>(p, s);
  return has_egemm ? launch_edge_fwd<128, true>(p, s) : launch_edge_fwd<128, false>(p, s);
}
