// 16-row, two-waves-per-SIMD forms of the fused edge kernels (gfx950): per-edge MLP +
// LayerNorm + receiver aggregation of one InteractionNet layer (interaction_net.py:102-131)
// and its backward, over the receiver-aligned 32-edge tiles of nlam_graph_tiles_host.
//
//   h_k   = [W1e e_k  or  Pe_k] + Ps[send(k)] + Pr[rec(k)]        (Pr carries b1)
//   m_k   = LN(W2 silu(h_k) + b2)
//   agg_i = scale_i * sum_{k: rec(k) = i} m_k                      (sum / mean)
//   e'_k  = e_k + m_k                                              (update_edges)
//
// Same parameter blocks, slab layout, grid and C entry points as fused_edge.hip; building blocks
// and register layout: fused16.h.  A wavefront takes a whole 32-edge tile as two 16-row halves:
// every row gather (e, Ps[send], Pr[rec], g_agg[rec], g_e') lands directly in accumulator
// layout (lane (t, g) loads the 16-byte chunks 16 fb + 4 g of ITS row: no LDS staging, no slot
// tables; the receivers of a half are 2-3 distinct rows, so their "gather" is a few cache
// lines per instruction), results leave the same way, and the receiver-side sums run over a
// 16-row fp32 LDS tile with the running sum carried from the first half to the second (the
// order of the additions is that of the 32-row kernel: bit-identical aggregates).
//
// Memory latency is hidden by a register prefetch, not by occupancy: the rows of the NEXT half
// that stream from HBM / MALL or scatter over the sender set (e, Ps[send], g_e') are requested
// at the top of the current half and land under its ~5 k cycles of work; the receiver rows
// (Pr, g_agg: 2-3 distinct L2-resident rows per half) are loaded at use.  (First version, every
// row loaded where it was consumed: 45 k cycles per half and wave, 5 k of them work --
// profiles/r03_*.)  The update-edges backward leaves dW1e = gh^T e to a streaming pass of
// nlam_outer_bwd over gh_out and e (both in original edge order: linear reads): two 64 x 64
// accumulators plus the prefetch do not fit in 256 registers.
#include <stdlib.h>

#include "fused16.h"
#include "fused_params.h"

#define K16_NW 8
#define K16_THREADS 512

// tasks (tile, batch item) of this wavefront: first, first + stride, ... < end.  With a grid
// that is a multiple of 8 the tasks are dealt in 8 contiguous chunks, one per group of
// workgroups that share blockIdx % 8 (= an XCD under the observed round-robin placement): the
// node rows a chunk gathers (Ps / Pr / g_agg of a mesh region) then stay in ONE 4 MB L2 instead
// of passing through all eight.  Placement only changes speed, never results.
struct TaskRange {
  unsigned first, stride, end;
};
__device__ __forceinline__ TaskRange k16_tasks(unsigned total, int wave) {
  TaskRange r;
  const unsigned G = gridDim.x, b = blockIdx.x;
  if ((G & 7u) == 0 && total >= 8 * K16_NW * 8) {
    const unsigned xcd = b & 7u, k = b >> 3, per = G >> 3;
    const unsigned chunk = (total + 7u) / 8u;
    const unsigned lo = xcd * chunk;
    unsigned hi = lo + chunk;
    if (hi > total) hi = total;
    r.first = lo + k * K16_NW + wave;
    r.stride = per * K16_NW;
    r.end = lo < total ? hi : 0;
    if (r.end == 0) r.first = 1;   // (empty chunk: no task)
  } else {
    r.first = b * K16_NW + wave;
    r.stride = G * K16_NW;
    r.end = total;
  }
  return r;
}

// slot indices of one 16-row half of a tile (lane (t, g): slot 16 hf + t)
struct HalfIdx {
  int eid, snd, rcv, rnx;   // rnx: receiver of the next slot (-1 past the tile)
};
__device__ __forceinline__ HalfIdx load_half_idx(const EdgeFwdParams& p, int p0, int ne, int hf,
                                                 int lane) {
  const int slot = 16 * hf + (lane & 15);
  const int pos = slot < ne ? p0 + slot : 0;          // padded slots read position 0 (exists)
  const int posn = slot + 1 < ne ? p0 + slot + 1 : 0;
  HalfIdx h;
  h.eid = p.csr_eid[pos];
  h.snd = p.csr_send[pos];
  h.rcv = p.csr_rec[pos];
  const int rn = p.csr_rec[posn];
  h.rnx = slot + 1 < ne ? rn : -1;
  return h;
}
__device__ __forceinline__ int4 load_hdr16(const EdgeFwdParams& p, unsigned tt, const TaskRange& tr) {
  const unsigned q = tt < tr.end ? tt : tr.end - 1;
  const int4 v = reinterpret_cast<const int4*>(p.tiles)[q % (unsigned)p.ntiles];
  int4 r;   // the header is the same in every lane: keep it in scalar registers
  r.x = __builtin_amdgcn_readfirstlane(v.x);
  r.y = __builtin_amdgcn_readfirstlane(v.y);
  r.z = __builtin_amdgcn_readfirstlane(v.z);
  r.w = __builtin_amdgcn_readfirstlane(v.w);
  return r;
}

// rows of one half that are requested a half ahead
template <bool GEO>
struct Pre16 {
  f32x4 E[4], ps[4], geo[GEO ? 4 : 1];
};
template <bool GEO>
__device__ __forceinline__ void issue_pre16(Pre16<GEO>& n, const HalfIdx& ix, const float* eb,
                                            int64_t e_ld, const float* psb, int64_t ps_ld,
                                            const float* gob, int64_t go_ld, bool has_geo, int lane) {
  load_row16<4>(n.E, eb + (int64_t)ix.eid * e_ld, lane);
  load_row16<4>(n.ps, psb + (int64_t)ix.snd * ps_ld, lane);
  if constexpr (GEO) {
    if (has_geo) load_row16<4>(n.geo, gob + (int64_t)ix.eid * go_ld, lane);
  }
}
__device__ __forceinline__ HalfIdx pick_idx(bool first, const HalfIdx& a, const HalfIdx& b) {
  HalfIdx r;
  r.eid = first ? a.eid : b.eid;
  r.snd = first ? a.snd : b.snd;
  r.rcv = first ? a.rcv : b.rcv;
  r.rnx = first ? a.rnx : b.rnx;
  return r;
}

// ================================================================== forward
template <bool HAS_EGEMM, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void edge_fwd16_kernel(EdgeFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64, NF = 4, LDT = D + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tasks, batch
  // item and every row base stay in scalar registers
  const int t = lane & 15;
  char* cur = smem16;
  const B3Image W1im = w16_image(cur, D, D);
  if (HAS_EGEMM) cur += w16_image_bytes(D, D);
  const B3Image W2im = w16_image(cur, D, D);
  cur += w16_image_bytes(D, D);
  float* b2s = reinterpret_cast<float*>(cur);
  float* gs = b2s + D;
  float* bs = gs + D;
  float* tile = bs + D + wave * (NLAM_T16 * LDT);
  if (HAS_EGEMM) load_weight_lds_b3(W1im, 0, p.W1e, p.ldW1e, D, D, D, D, tid, K16_THREADS);
  load_weight_lds_b3(W2im, 0, p.W2, p.ldW2, D, D, D, D, tid, K16_THREADS);
  load_vec_lds(b2s, p.b2, D, D, tid, K16_THREADS);
  load_vec_lds(gs, p.gamma, D, D, tid, K16_THREADS);
  load_vec_lds(bs, p.beta, D, D, tid, K16_THREADS);
  __syncthreads();

  const TaskRange tr = k16_tasks((unsigned)(p.ntiles * p.B), wave);
  if (tr.first >= tr.end) return;
  // indices run one task ahead of the rows, headers two (dependent loads off the critical path)
  int4 hdr = load_hdr16(p, tr.first, tr);
  HalfIdx i0 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 0, lane);
  HalfIdx i1 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 1, lane);
  int4 hdr_n = load_hdr16(p, tr.first + tr.stride, tr);
  Pre16<false> nx;
  {
    const unsigned b0 = tr.first / (unsigned)p.ntiles;
    issue_pre16<false>(nx, i0, p.e.ptr + (int64_t)b0 * p.e.bstride, p.e.ld,
                       p.ps.ptr + (int64_t)b0 * p.ps.bstride, p.ps.ld, nullptr, 0, false, lane);
  }
  for (unsigned tt = tr.first; tt < tr.end; tt += tr.stride) {
    const unsigned b = tt / (unsigned)p.ntiles;
    const unsigned tn = tt + tr.stride < tr.end ? tt + tr.stride : tr.end - 1;
    const unsigned bn = tn / (unsigned)p.ntiles;
    const int ne = hdr.y - hdr.x, r0 = hdr.z, nr = hdr.w - hdr.z;
    const HalfIdx n0 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 0, lane);
    const HalfIdx n1 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 1, lane);
    const int4 hdr_nn = load_hdr16(p, tt + 2 * tr.stride, tr);
    const float* eb = p.e.ptr + (int64_t)b * p.e.bstride;
    const float* psb = p.ps.ptr + (int64_t)b * p.ps.bstride;
    const float* prb = p.pr.ptr + (int64_t)b * p.pr.bstride;
    float* ab = p.agg + (int64_t)b * p.agg_bstride;
    float carry = 0.f;
    int nseg = 0;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int nh = ne - 16 * hf < NLAM_T16 ? ne - 16 * hf : NLAM_T16;
      if (nh <= 0) break;   // wave-uniform
      const HalfIdx ix = hf == 0 ? i0 : i1;
      const bool valid = t < nh;
      const bool is_end = valid && (ix.rnx != ix.rcv);
      const unsigned ends = (unsigned)(__ballot(is_end) & 0xffffull);
      nseg += __popc(ends);
      // this half's rows: e and Ps were requested a half ago; Pr (L2) now
      f32x4 E[NF], h[NF];
      {
        f32x4 c[NF];
        load_row16<NF>(c, prb + (int64_t)ix.rcv * p.pr.ld, lane);
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) {
          E[fb] = nx.E[fb];
          h[fb] = nx.ps[fb] + c[fb];
        }
      }
      {   // the next half's e / Ps rows (second half of this tile, or the next tile's first half)
        const bool more = hf == 0 && ne > NLAM_T16;
        const HalfIdx nix = pick_idx(more, i1, n0);
        issue_pre16<false>(nx, nix, more ? eb : p.e.ptr + (int64_t)bn * p.e.bstride, p.e.ld,
                           more ? psb : p.ps.ptr + (int64_t)bn * p.ps.bstride, p.ps.ld, nullptr, 0,
                           false, lane);
      }
      Frag16<2> fr;
      if constexpr (HAS_EGEMM) {
        make_frag16<2, TERMS>(fr, E);
        gemm_frag16<NF, 2, TERMS>(h, W1im, 0, 0, fr, lane);
      } else {
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) h[fb] += E[fb];      // Pe + Ps + Pr
      }
#pragma unroll
      for (int fb = 0; fb < NF; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[fb][r] = nlam_silu(h[fb][r]);
      f32x4 m[NF];
      vec_to_acc16<NF>(m, b2s, lane);
      make_frag16<2, TERMS>(fr, h);
      gemm_frag16<NF, 2, TERMS>(m, W2im, 0, 0, fr, lane);
      ln16_apply<NF>(m, gs, bs, lane);
      // messages -> LDS; receiver-side sums with lanes = features
      acc16_to_tile<NF>(m, tile, LDT, lane);
      if constexpr (HAS_EGEMM) {
        if (valid) {
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) E[fb] += m[fb];
          store_row16<NF>(p.e_out + (int64_t)b * p.eo_bstride + (int64_t)ix.eid * p.eo_ld, E, lane);
        }
      }
      wave_sync();
      half_segment_sums(tile, LDT, nh, ends, ix.rcv, lane, carry, [&](int r, float acc) {
        const float sc = p.inv_deg ? p.inv_deg[r] : 1.0f;
        ab[(int64_t)r * p.agg_ld + lane] = acc * sc;
      });
      wave_sync();
    }
    if (nseg != nr) {   // (rare) receivers without in-edges inside the tile: their aggregate is 0
      for (int i = 0; i < nr; ++i) {
        const int beg = p.csr_rowptr[r0 + i], end = p.csr_rowptr[r0 + i + 1];
        if (beg == end) ab[(int64_t)(r0 + i) * p.agg_ld + lane] = 0.f;
      }
    }
    hdr = hdr_n;
    hdr_n = hdr_nn;
    i0 = n0;
    i1 = n1;
  }
}

template <bool HAS_EGEMM>
static int launch_edge_fwd16(const EdgeFwdParams& p, hipStream_t s) {
  constexpr int D = 64;
  const size_t lds = (HAS_EGEMM ? 2 : 1) * w16_image_bytes(D, D) + 3 * D * sizeof(float) +
                     (size_t)K16_NW * NLAM_T16 * (D + 4) * sizeof(float);
  auto kern = edge_fwd16_kernel<HAS_EGEMM, 3>;
  NLAM_BIG_LDS(kern, __func__);
  int64_t g = (p.ntiles * p.B + K16_NW - 1) / K16_NW;
  if (g > 256) g = 256;
  if (g > 8) g &= ~(int64_t)7;      // multiple of 8: XCD-chunked tasks (k16_tasks)
  kern<<<(unsigned)g, K16_THREADS, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("edge_fwd16_kernel");
  return 0;
}

int nlam_k16_edge_fwd(const EdgeFwdParams& p, int has_egemm, hipStream_t s) {
  if (!nlam_k16_on(K16_EDGE_FWD) || !nlam_mfma_b3()) return -1;
  if (p.e.width != 64) return -1;
  return has_egemm ? launch_edge_fwd16<true>(p, s) : launch_edge_fwd16<false>(p, s);
}

// ================================================================= backward
// Recomputes h, s = silu(h), z = W2 s + b2 from the inputs, then
//   gm_k = scale * g_agg[rec(k)] + g_eout_k
//   gz   = LN'(z; gm),   dW2 += gz (x) s,  db2 += gz,  dgamma, dbeta
//   gh   = (W2^T gz) * silu'(h)       -> gh_out (original edge order), gPr_i = sum_{rec=i} gh
//   g_e  = g_eout + W1e^T gh                                     (has_egemm)
// Slab per workgroup: [dW1e slot (D x D, NOT written: nlam_edge_bwd_defers_dw1e) | dW2 (D x D) |
// db2 | dgamma | dbeta].
template <bool HAS_EGEMM, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void edge_bwd16_kernel(EdgeBwdParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64, NF = 4, LDT = D + 4;
  const EdgeFwdParams& p = q.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  char* cur = smem16;
  const B3Image W1im = w16_image(cur, D, D);
  if (HAS_EGEMM) cur += w16_image_bytes(D, D);
  const B3Image W2im = w16_image(cur, D, D);
  cur += w16_image_bytes(D, D);
  float* b2s = reinterpret_cast<float*>(cur);
  float* gs = b2s + D;
  cur += 2 * D * sizeof(float);
  // per wave: TA (S planes) | TB (g / g xhat / GZ planes) | HS (fp32 GH tile of the
  // receiver-side sums)
  char* mine = cur + wave * (3 * p16_bytes(D));
  static_assert(p16_bytes(D) == (size_t)NLAM_T16 * LDT * sizeof(float), "fp32 tile = plane pair");
  const B3Tile TA = p16_tile(mine, D), TB = p16_tile(mine + p16_bytes(D), D);
  float* HS = reinterpret_cast<float*>(mine + 2 * p16_bytes(D));
  if (HAS_EGEMM) load_weight_lds_b3(W1im, 0, p.W1e, p.ldW1e, D, D, D, D, tid, K16_THREADS);
  load_weight_lds_b3(W2im, 0, p.W2, p.ldW2, D, D, D, D, tid, K16_THREADS);
  load_vec_lds(b2s, p.b2, D, D, tid, K16_THREADS);
  load_vec_lds(gs, p.gamma, D, D, tid, K16_THREADS);
  __syncthreads();

  f32x16 dW2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW2[i][j][r] = 0.f;
  float db2[1] = {0.f}, dgam[1] = {0.f}, dbet[1] = {0.f};

  const bool has_geo = HAS_EGEMM && q.g_eout != nullptr;
  const TaskRange tr = k16_tasks((unsigned)(p.ntiles * p.B), wave);
  int4 hdr = {0, 0, 0, 0}, hdr_n = {0, 0, 0, 0};
  HalfIdx i0 = {0, 0, 0, -1}, i1 = {0, 0, 0, -1};
  Pre16<HAS_EGEMM> nx;
  if (tr.first < tr.end) {
    hdr = load_hdr16(p, tr.first, tr);
    i0 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 0, lane);
    i1 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 1, lane);
    hdr_n = load_hdr16(p, tr.first + tr.stride, tr);
    const unsigned b0 = tr.first / (unsigned)p.ntiles;
    issue_pre16<HAS_EGEMM>(nx, i0, p.e.ptr + (int64_t)b0 * p.e.bstride, p.e.ld,
                           p.ps.ptr + (int64_t)b0 * p.ps.bstride, p.ps.ld,
                           has_geo ? q.g_eout + (int64_t)b0 * q.geo_bstride : nullptr, q.geo_ld,
                           has_geo, lane);
  }
  for (unsigned tt = tr.first; tt < tr.end; tt += tr.stride) {
    const unsigned b = tt / (unsigned)p.ntiles;
    const unsigned tn = tt + tr.stride < tr.end ? tt + tr.stride : tr.end - 1;
    const unsigned bn = tn / (unsigned)p.ntiles;
    const int ne = hdr.y - hdr.x, r0 = hdr.z, nr = hdr.w - hdr.z;
    const HalfIdx n0 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 0, lane);
    const HalfIdx n1 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 1, lane);
    const int4 hdr_nn = load_hdr16(p, tt + 2 * tr.stride, tr);
    const float* eb = p.e.ptr + (int64_t)b * p.e.bstride;
    const float* psb = p.ps.ptr + (int64_t)b * p.ps.bstride;
    const float* prb = p.pr.ptr + (int64_t)b * p.pr.bstride;
    const float* gab = q.g_agg.ptr + (int64_t)b * q.g_agg.bstride;
    const float* gob = has_geo ? q.g_eout + (int64_t)b * q.geo_bstride : nullptr;
    float* gb = q.gpr + (int64_t)b * q.gpr_bstride;
    float carry = 0.f;
    int nseg = 0;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int nh = ne - 16 * hf < NLAM_T16 ? ne - 16 * hf : NLAM_T16;
      if (nh <= 0) break;   // wave-uniform
      const HalfIdx ix = hf == 0 ? i0 : i1;
      const bool valid = t < nh;
      const bool is_end = valid && (ix.rnx != ix.rcv);
      const unsigned ends = (unsigned)(__ballot(is_end) & 0xffffull);
      nseg += __popc(ends);
      // ---- this half's rows: e, Ps (and g_e') were requested a half ago; Pr, g_agg (L2) now
      f32x4 h[NF], g[NF], geo[HAS_EGEMM ? NF : 1];
      Frag16<2> fr;
      {
        f32x4 c[NF];
        load_row16<NF>(c, prb + (int64_t)ix.rcv * p.pr.ld, lane);
        load_row16<NF>(g, gab + (int64_t)ix.rcv * q.g_agg.ld, lane);
        if constexpr (HAS_EGEMM) {
          make_frag16<2, TERMS>(fr, nx.E);
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) {
            h[fb] = nx.ps[fb] + c[fb];
            geo[fb] = nx.geo[fb];
          }
        } else {
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) h[fb] = (nx.ps[fb] + c[fb]) + nx.E[fb];    // Pe + Ps + Pr
        }
      }
      {   // the next half's rows (second half of this tile, or the next tile's first half)
        const bool more = hf == 0 && ne > NLAM_T16;
        const HalfIdx nix = pick_idx(more, i1, n0);
        issue_pre16<HAS_EGEMM>(nx, nix, more ? eb : p.e.ptr + (int64_t)bn * p.e.bstride, p.e.ld,
                               more ? psb : p.ps.ptr + (int64_t)bn * p.ps.bstride, p.ps.ld,
                               has_geo ? (more ? gob : q.g_eout + (int64_t)bn * q.geo_bstride) : nullptr,
                               q.geo_ld, has_geo, lane);
      }
      // ---- recompute the forward: h, s = silu(h) (and silu'(h) from the same sigmoid), z
      if constexpr (HAS_EGEMM) gemm_frag16<NF, 2, TERMS>(h, W1im, 0, 0, fr, lane);
      {
        f32x4 ds[NF];   // silu'(h): parked in LDS (HS is free until the GH tile) for the gh product
#pragma unroll
        for (int fb = 0; fb < NF; ++fb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sv, dv;
            silu_both(h[fb][r], sv, dv);
            h[fb][r] = sv;
            ds[fb][r] = dv;
          }
        acc16_to_tile<NF>(ds, HS, LDT, lane);
      }
      make_frag16<2, TERMS>(fr, h);
      frag16_to_planes<2, TERMS>(fr, TA, 0, lane);             // S stays in TA until dW2 is formed
      {
        f32x4 z[NF];
        vec_to_acc16<NF>(z, b2s, lane);
        gemm_frag16<NF, 2, TERMS>(z, W2im, 0, 0, fr, lane);
        __builtin_amdgcn_sched_barrier(0);
        // ---- incoming gradient of the messages: scale * g_agg[rec] (+ g_e')
        if (p.inv_deg != nullptr) {
          const float sc = p.inv_deg[ix.rcv];
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) g[fb] *= sc;
        }
        if constexpr (HAS_EGEMM) {
          if (has_geo) {
#pragma unroll
            for (int fb = 0; fb < NF; ++fb) g[fb] += geo[fb];
          }
        }
        mask16<NF>(g, valid);   // padded slots carry a zero gradient: every sum below ignores them
        acc16_to_planes<NF, TERMS>(g, TB, 0, lane);             // dbeta summand
        wave_sync();
        colsum16<1, TERMS>(dbet, TB, 0, lane);
        wave_sync();
        ln16_bwd<NF, TERMS>(z, g, TB, gs, lane);                // g -> gz; g * xhat -> planes
        wave_sync();
        colsum16<1, TERMS>(dgam, TB, 0, lane);
        wave_sync();
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- dW2 += gz (x) s, db2 += gz
      make_frag16<2, TERMS>(fr, g);
      frag16_to_planes<2, TERMS>(fr, TB, 0, lane);
      wave_sync();
      colsum16<1, TERMS>(db2, TB, 0, lane);
      outer_accum16<2, 2, TERMS>(dW2, TB, 0, TA, 0, lane);
      __builtin_amdgcn_sched_barrier(0);
      // ---- gh = (W2^T gz) * silu'(h)
      f32x4 gh[NF];
      zero16<NF>(gh);
      gemm_frag16_wt<NF, 2, TERMS>(gh, W2im, 0, 0, fr, lane);
#pragma unroll
      for (int fb = 0; fb < NF; ++fb)
        gh[fb] *= *reinterpret_cast<const f32x4*>(HS + t * LDT + 16 * fb + 4 * (lane >> 4));
      if (valid)
        store_row16<NF>(q.gh_out + (int64_t)b * q.gh_bstride + (int64_t)ix.eid * D, gh, lane);
      // receiver-side sum of gh (segments are tile-local)
      acc16_to_tile<NF>(gh, HS, LDT, lane);
      wave_sync();
      half_segment_sums(HS, LDT, nh, ends, ix.rcv, lane, carry, [&](int r, float acc) {
        gb[(int64_t)r * q.gpr_ld + lane] = acc;
      });
      if constexpr (HAS_EGEMM) {
        // ---- g_e = g_e' + W1e^T gh
        __builtin_amdgcn_sched_barrier(0);
        make_frag16<2, TERMS>(fr, gh);
        if (!has_geo) zero16<NF>(geo);
        gemm_frag16_wt<NF, 2, TERMS>(geo, W1im, 0, 0, fr, lane);
        if (valid)
          store_row16<NF>(q.g_e + (int64_t)b * q.ge_bstride + (int64_t)ix.eid * q.ge_ld, geo, lane);
      }
      wave_sync();
    }
    if (nseg != nr) {   // (rare) receivers without in-edges inside the tile
      for (int i = 0; i < nr; ++i) {
        const int beg = p.csr_rowptr[r0 + i], end = p.csr_rowptr[r0 + i + 1];
        if (beg == end) gb[(int64_t)(r0 + i) * q.gpr_ld + lane] = 0.f;
      }
    }
    hdr = hdr_n;
    hdr_n = hdr_nn;
    i0 = n0;
    i1 = n1;
  }

  __syncthreads();
  float* img = reinterpret_cast<float*>(smem16);
  float* slab = q.slab + (int64_t)blockIdx.x * q.slab_stride;
  constexpr int nW = D * D;
  fold_blocks_to_slab16<2, 2, 2, K16_NW>(&dW2[0][0], img, D, slab + nW, tid, wave, lane);
  fold_vec_to_slab16<1, K16_NW>(db2, img, slab + 2 * nW, D, tid, wave, lane);
  fold_vec_to_slab16<1, K16_NW>(dgam, img, slab + 2 * nW + D, D, tid, wave, lane);
  fold_vec_to_slab16<1, K16_NW>(dbet, img, slab + 2 * nW + 2 * D, D, tid, wave, lane);
}

template <bool HAS_EGEMM>
static int launch_edge_bwd16(const EdgeBwdParams& q, hipStream_t s) {
  constexpr int D = 64;
  size_t lds = (HAS_EGEMM ? 2 : 1) * w16_image_bytes(D, D) + 2 * D * sizeof(float) +
               (size_t)K16_NW * 3 * p16_bytes(D);
  const size_t fold = (size_t)K16_NW * D * D * sizeof(float);
  if (fold > lds) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "edge_bwd16: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = edge_bwd16_kernel<HAS_EGEMM, 3>;
  NLAM_BIG_LDS(kern, __func__);
  // one slab per workgroup: the grid is what the host side sized the slabs for
  kern<<<(unsigned)nlam_bwd_grid(q.f.ntiles * q.f.B), K16_THREADS, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("edge_bwd16_kernel");
  return 0;
}

static bool edge_bwd16_applies(int d, int has_egemm) {
  return d == 64 && nlam_k16_on(has_egemm ? K16_EDGE_BWD_UPD : K16_EDGE_BWD) && nlam_mfma_b3() &&
         getenv("NLAM_STAMP") == nullptr;
}
// 1: nlam_edge_bwd (update_edges form) leaves the dW1e slot of its slabs unwritten; the caller
// forms dW1e = gh_out^T e with nlam_outer_bwd (both operands in original edge order)
extern "C" int nlam_edge_bwd_defers_dw1e(int d) { return edge_bwd16_applies(d, 1) ? 1 : 0; }

int nlam_k16_edge_bwd(const EdgeBwdParams& q, int has_egemm, hipStream_t s) {
  if (!edge_bwd16_applies(q.f.e.width, has_egemm)) return -1;
  return has_egemm ? launch_edge_bwd16<true>(q, s) : launch_edge_bwd16<false>(q, s);
}
