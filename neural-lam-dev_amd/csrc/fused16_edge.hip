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
// and register layout: fused16.h.  A wavefront takes a whole 32-edge tile as two 16-row halves;
// the receiver-side sums run over a 16-row fp32 LDS tile with the running sum carried from the
// first half to the second (the order of the additions is that of the 32-row kernel:
// bit-identical aggregates).
//
// What bounds these kernels, measured (profiles/r03_stamp16_*, r03_pmc_*): the CU's vector-memory
// pipe.  One wave-wide 16-byte access costs 45-140 cycles of it per CU whatever it hits, and a
// first version that loaded every row where it was consumed, per edge, and spilled to scratch
// issued ~60 of them per 16 edges: 84 % of a wave's time went to issuing loads into a full queue
// and waiting for them.  So the NUMBER of vector-memory instructions is what is minimised:
//   * whole rows: the gathered edge / sender rows and the scattered results move as 4 whole
//     256-byte rows per instruction and change shape through the wave's fp32 LDS tile;
//   * receiver rows (Pr, g_agg): the receivers of a half are 1-5 CONSECUTIVE rows, loaded once
//     (one instruction per tensor instead of one row per edge) and expanded to the edge slots
//     from LDS;
//   * three index loads per half (edge id, sender, receiver); the tile headers are scalar loads;
//   * the edge rows of the next half (e, g_e': HBM / MALL) are requested a half ahead; nothing
//     is loaded twice and nothing spills.
// The update-edges backward shares its two weight-gradient products over the workgroup
// (edge_bwd16c_kernel) so that all of this fits in 256 registers.
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

// tile header (p0, p1, r0, r1): a SCALAR load (wave-uniform address in the constant address
// space; the tile table is never written by these kernels)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) i32x4 const_i32x4;
__device__ __forceinline__ i32x4 tile_hdr(const EdgeFwdParams& p, unsigned tt, unsigned end) {
  const unsigned q = tt < end ? tt : end - 1;
  return *reinterpret_cast<const_i32x4*>(
      reinterpret_cast<uintptr_t>(p.tiles + 4 * (size_t)(q % (unsigned)p.ntiles)));
}

// slot indices of one 16-row half of a tile (lane (t, g): slot 16 hf + t)
struct HalfIdx {
  int eid, snd, rcv;
};
__device__ __forceinline__ HalfIdx load_half_idx(const EdgeFwdParams& p, int p0, int ne, int hf,
                                                 int lane) {
  const int slot = 16 * hf + (lane & 15);
  const int pos = slot < ne ? p0 + slot : 0;          // padded slots read position 0 (exists)
  HalfIdx h;
  h.eid = p.csr_eid[pos];
  h.snd = p.csr_send[pos];
  h.rcv = p.csr_rec[pos];
  return h;
}
__device__ __forceinline__ HalfIdx pick_idx(bool first, const HalfIdx& a, const HalfIdx& b) {
  HalfIdx r;
  r.eid = first ? a.eid : b.eid;
  r.snd = first ? a.snd : b.snd;
  r.rcv = first ? a.rcv : b.rcv;
  return r;
}
// bit s: slot s of this half closes its receiver's segment (its successor -- the next lane, or
// the first slot of the second half -- has another receiver, or it is the tile's last edge)
__device__ __forceinline__ unsigned half_ends(int rcv, int next_first_rcv, int hf, int ne, int lane) {
  const int t = lane & 15, slot = 16 * hf + t;
  int rn = __shfl_down(rcv, 1, 64);
  if (t == 15) rn = next_first_rcv;
  const bool is_end = slot < ne && (slot + 1 >= ne || rn != rcv);
  return (unsigned)(__ballot(is_end) & 0xffffull);
}

// the edge rows of one half (e, g_e'), requested a half ahead; row-shaped (fused16.h)
template <bool GEO>
struct Pre16 {
  f32x4 E[4], geo[GEO ? 4 : 1];
};
template <bool GEO>
__device__ __forceinline__ void issue_pre16(Pre16<GEO>& n, int eid_t, const float* eb, int64_t e_ld,
                                            const float* gob, int64_t go_ld, bool has_geo, int lane) {
  int re[4];
  rs_index(re, eid_t, lane);
  rs_load(n.E, eb, e_ld, re, lane);
  if constexpr (GEO) {
    if (has_geo) rs_load(n.geo, gob, go_ld, re, lane);
  }
}
__device__ __forceinline__ void load_ps_rs(f32x4 (&v)[4], const float* psb, int64_t ps_ld, int snd_t,
                                           int lane) {
  int rs[4];
  rs_index(rs, snd_t, lane);
  rs_load(v, psb, ps_ld, rs, lane);
}

// ---- receiver rows, once per distinct receiver -------------------------------------------------
// The receivers of a half are the consecutive rows ra .. ra + nrecv - 1 (receiver-sorted edges;
// 1-5 on the neural-lam graphs).  They are fetched row-shaped in chunks of 8 (lane (r4, c): row
// ra + 4 k + r4, clamped), parked in 8 rows of the fp32 tile and read back by every edge slot of
// that chunk.  The first chunk is requested early (recv_issue) and lands under the first GEMM;
// further chunks (more than 8 receivers in 16 edges: none of the reference's graphs) are requested
// where they are expanded.
struct RecvRows {
  f32x4 v[2];
};
__device__ __forceinline__ void recv_issue(RecvRows& r, const float* __restrict__ base, int64_t ld,
                                           int ra, int nrecv, int lane) {
  const int r4 = lane >> 4, c = lane & 15, last = nrecv - 1;
  const int a0 = r4 < last ? r4 : last, a1 = 4 + r4 < last ? 4 + r4 : last;
  r.v[0] = *reinterpret_cast<const f32x4*>(base + (int64_t)(ra + a0) * ld + 4 * c);
  if (nrecv > 4) r.v[1] = *reinterpret_cast<const f32x4*>(base + (int64_t)(ra + a1) * ld + 4 * c);
}
// rows -> tile rows row0 .. row0 + 7
__device__ __forceinline__ void recv_to_tile(const RecvRows& r, float* __restrict__ tile, int ld,
                                             int row0, int nrecv, int lane) {
  const int r4 = lane >> 4, c = lane & 15;
  *reinterpret_cast<f32x4*>(tile + (row0 + r4) * ld + 4 * c) = r.v[0];
  if (nrecv > 4) *reinterpret_cast<f32x4*>(tile + (row0 + 4 + r4) * ld + 4 * c) = r.v[1];
}
// accumulator layout <- tile row `row0 + ro` for the slots whose receiver is in this chunk
__device__ __forceinline__ void recv_expand(f32x4 (&a)[4], const float* __restrict__ tile, int ld,
                                            int row0, int ro, int lane) {
  const int g = lane >> 4;
  const bool mine = ro >= 0 && ro < 8;
  const int row = row0 + (mine ? ro : 0);
#pragma unroll
  for (int fb = 0; fb < 4; ++fb) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * ld + 16 * fb + 4 * g);
    if (mine) a[fb] = v;
  }
}
// ONE tensor (Pr) / TWO tensors (Pr, g_agg): first chunk already requested into pv (/ gv)
template <bool TWO>
__device__ __forceinline__ void recv_gather(f32x4 (&c)[4], f32x4 (&g)[4], RecvRows& pv, RecvRows& gv,
                                            const float* __restrict__ prb, int64_t pr_ld,
                                            const float* __restrict__ gab, int64_t ga_ld, int ra,
                                            int nrecv, int rcv_t, float* __restrict__ tile, int ld,
                                            int lane) {
  for (int rb0 = 0;;) {
    const int nrem = nrecv - rb0 < 8 ? nrecv - rb0 : 8;
    recv_to_tile(pv, tile, ld, 0, nrem, lane);
    if constexpr (TWO) recv_to_tile(gv, tile, ld, 8, nrem, lane);
    wave_sync();
    recv_expand(c, tile, ld, 0, rcv_t - ra - rb0, lane);
    if constexpr (TWO) recv_expand(g, tile, ld, 8, rcv_t - ra - rb0, lane);
    wave_sync();
    rb0 += 8;
    if (rb0 >= nrecv) break;   // wave-uniform
    recv_issue(pv, prb, pr_ld, ra + rb0, nrecv - rb0 < 8 ? nrecv - rb0 : 8, lane);
    if constexpr (TWO) recv_issue(gv, gab, ga_ld, ra + rb0, nrecv - rb0 < 8 ? nrecv - rb0 : 8, lane);
  }
}

// ================================================================== forward
// (128 registers and <= 73 KB of LDS: two workgroups per CU, four waves per SIMD)
template <bool HAS_EGEMM, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 4) void edge_fwd16_kernel(EdgeFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64, NF = 4, LDT = D + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tasks, batch
  // item and every row base stay in scalar registers
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
  {   // (batched prologue loads: one global round trip)
    VLoad16 lv;
    const float* const vecs[8] = {p.b2, p.gamma, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int lens[8] = {D, D, 0, 0, 0, 0, 0, 0};
    v16_issue(lv, vecs, lens, tid);
    WLoad16<2> l2;
    w16_issue(l2, p.W2, p.ldW2, D, D, D, D, tid, K16_THREADS);
    v16_commit(lv, b2s, 2, tid);
    w16_commit(l2, W2im, 0, p.W2, p.ldW2, D, D, D, D, tid, K16_THREADS);
  }
  load_vec_lds(bs, p.beta, D, D, tid, K16_THREADS);
  __syncthreads();

  const TaskRange tr = k16_tasks((unsigned)(p.ntiles * p.B), wave);
  if (tr.first >= tr.end) return;
  // indices run one task ahead of the rows, headers two (dependent loads off the critical path)
  i32x4 hdr = tile_hdr(p, tr.first, tr.end);
  HalfIdx i0 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 0, lane);
  HalfIdx i1 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 1, lane);
  i32x4 hdr_n = tile_hdr(p, tr.first + tr.stride, tr.end);
  Pre16<false> nx;
  {
    const unsigned b0 = tr.first / (unsigned)p.ntiles;
    issue_pre16<false>(nx, i0.eid, p.e.ptr + (int64_t)b0 * p.e.bstride, p.e.ld, nullptr, 0, false, lane);
  }
  for (unsigned tt = tr.first; tt < tr.end; tt += tr.stride) {
    const unsigned b = tt / (unsigned)p.ntiles;
    const unsigned tn = tt + tr.stride < tr.end ? tt + tr.stride : tr.end - 1;
    const unsigned bn = tn / (unsigned)p.ntiles;
    const int ne = hdr.y - hdr.x, r0 = hdr.z, nr = hdr.w - hdr.z;
    const HalfIdx n0 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 0, lane);
    const HalfIdx n1 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 1, lane);
    const i32x4 hdr_nn = tile_hdr(p, tt + 2 * tr.stride, tr.end);
    const float* eb = p.e.ptr + (int64_t)b * p.e.bstride;
    const float* psb = p.ps.ptr + (int64_t)b * p.ps.bstride;
    const float* prb = p.pr.ptr + (int64_t)b * p.pr.bstride;
    float* ab = p.agg + (int64_t)b * p.agg_bstride;
    float carry = 0.f;
    int nseg = 0;
#pragma nounroll
    for (int hf = 0; hf < 2; ++hf) {
      const int nh = ne - 16 * hf < NLAM_T16 ? ne - 16 * hf : NLAM_T16;
      if (nh <= 0) break;   // wave-uniform
      const HalfIdx ix = pick_idx(hf == 0, i0, i1);
      const int nfirst = hf == 0 ? __builtin_amdgcn_readfirstlane(i1.rcv) : -1;
      const unsigned ends = half_ends(ix.rcv, nfirst, hf, ne, lane);
      nseg += __popc(ends);
      const int ra = __builtin_amdgcn_readfirstlane(ix.rcv);
      const int nrecv = __builtin_amdgcn_readlane(ix.rcv, nh - 1) - ra + 1;
      // ---- this half's rows: e was requested a half ago; Ps (whole rows) and Pr (once per
      // receiver) now, landing under the first GEMM
      f32x4 E[NF], h[NF];
      {
        f32x4 psv[4], c[NF];
        RecvRows prv;
        zero16<NF>(c);
        load_ps_rs(psv, psb, p.ps.ld, ix.snd, lane);
        recv_issue(prv, prb, p.pr.ld, ra, nrecv < 8 ? nrecv : 8, lane);
        rs_to_acc16(E, nx.E, tile, LDT, lane);
        if constexpr (HAS_EGEMM) {
          Frag16<2> fr;
          zero16<NF>(h);
          make_frag16<2, TERMS>(fr, E);
          gemm_frag16<NF, 2, TERMS>(h, W1im, 0, 0, fr, lane);
        } else {
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) h[fb] = E[fb];      // Pe
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          f32x4 pa[NF];
          rs_to_acc16(pa, psv, tile, LDT, lane);
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) h[fb] += pa[fb];
        }
        __builtin_amdgcn_sched_barrier(0);
        recv_gather<false>(c, c, prv, prv, prb, p.pr.ld, nullptr, 0, ra, nrecv, ix.rcv, tile, LDT, lane);
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) h[fb] += c[fb];
        __builtin_amdgcn_sched_barrier(0);
        {   // the next half's e rows (second half of this tile, or the next tile's first half)
          const bool more = hf == 0 && ne > NLAM_T16;
          issue_pre16<false>(nx, more ? i1.eid : n0.eid, more ? eb : p.e.ptr + (int64_t)bn * p.e.bstride,
                             p.e.ld, nullptr, 0, false, lane);
        }
      }
#pragma unroll
      for (int fb = 0; fb < NF; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[fb][r] = nlam_silu(h[fb][r]);
      f32x4 m[NF];
      vec_to_acc16<NF>(m, b2s, lane);
      {
        Frag16<2> fr;
        make_frag16<2, TERMS>(fr, h);
        gemm_frag16<NF, 2, TERMS>(m, W2im, 0, 0, fr, lane);
      }
      ln16_apply<NF>(m, gs, bs, lane);
      // messages -> LDS; receiver-side sums with lanes = features
      acc16_to_tile<NF>(m, tile, LDT, lane);
      wave_sync();
      half_segment_sums(tile, LDT, nh, ends, ix.rcv, lane, carry, [&](int r, float acc) {
        const float sc = p.inv_deg ? p.inv_deg[r] : 1.0f;
        ab[(int64_t)r * p.agg_ld + lane] = acc * sc;
      });
      wave_sync();
      if constexpr (HAS_EGEMM) {   // e' = e + m leaves as whole rows through the same tile
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) E[fb] += m[fb];
        acc16_to_tile<NF>(E, tile, LDT, lane);
        wave_sync();
        tile_store_rs(tile, LDT, p.e_out + (int64_t)b * p.eo_bstride, p.eo_ld, ix.eid, nh, lane);
        wave_sync();
      }
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
  const int64_t cap = 512;   // two workgroups per CU (<= 128 registers, <= 73 KB of LDS)
  if (g > cap) g = cap;
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

// Diagnostic build only (NLAM_STAMP16=1): per-segment cycle sums of the per-wave backward tile loop,
// summed over all waves (s_memtime stamps perturb the schedule: read the SHARES).
__device__ unsigned long long g_k16_stamps[16];
extern "C" int nlam_debug_k16_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k16_stamps), sizeof(unsigned long long) * 16) != hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_k16_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#define STAMP16(k)                                                \
  if constexpr (STAMP) {                                          \
    __builtin_amdgcn_sched_barrier(0);                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);                           \
    __builtin_amdgcn_sched_barrier(0);                            \
    st[k] += now_ - tprev;                                        \
    tprev = now_;                                                 \
  }

// ================================================================= backward
// Recomputes h, s = silu(h), z = W2 s + b2 from the inputs, then
//   gm_k = scale * g_agg[rec(k)] + g_eout_k
//   gz   = LN'(z; gm),   dW2 += gz (x) s,  db2 += gz,  dgamma, dbeta
//   gh   = (W2^T gz) * silu'(h)       -> gh_out (original edge order), gPr_i = sum_{rec=i} gh
//   dW1e += gh (x) e ;  g_e = g_eout + W1e^T gh                 (update_edges: edge_bwd16c_kernel)
// Slab per workgroup: [dW1e (D x D) | dW2 (D x D) | db2 | dgamma | dbeta].
//
// ---- without edge update (g2m, m2g: e is the projected Pe): dW2 per wave ---------------------
// ABL (diagnostic, NLAM_ABL16): 1 = no global row traffic (rows synthesised in registers, result
// stores dropped), 2 = no matrix / LayerNorm / column-sum arithmetic (rows still move): which floor
// the kernel sits on.  Results are wrong in both; only the time is read.
template <int TERMS, bool STAMP = false, int ABL = 0>
__global__ __launch_bounds__(K16_THREADS, 2) void edge_bwd16_kernel(EdgeBwdParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64, NF = 4, LDT = D + 4;
  const EdgeFwdParams& p = q.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  char* cur = smem16;
  const B3Image W2im = w16_image(cur, D, D);
  cur += w16_image_bytes(D, D);
  float* b2s = reinterpret_cast<float*>(cur);
  float* gs = b2s + D;
  cur += 2 * D * sizeof(float);
  // per wave: TA (S planes) | TB (g / g xhat / GZ planes) | HS (fp32 tile: row transposes,
  // receiver rows, silu'(h), then the GH tile of the receiver-side sums)
  char* mine = cur + wave * (3 * p16_bytes(D));
  static_assert(p16_bytes(D) == (size_t)NLAM_T16 * LDT * sizeof(float), "fp32 tile = plane pair");
  const B3Tile TA = p16_tile(mine, D), TB = p16_tile(mine + p16_bytes(D), D);
  float* HS = reinterpret_cast<float*>(mine + 2 * p16_bytes(D));
  {   // (batched prologue loads: one global round trip)
    VLoad16 lv;
    const float* const vecs[8] = {p.b2, p.gamma, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int lens[8] = {D, D, 0, 0, 0, 0, 0, 0};
    v16_issue(lv, vecs, lens, tid);
    WLoad16<2> l2;
    w16_issue(l2, p.W2, p.ldW2, D, D, D, D, tid, K16_THREADS);
    v16_commit(lv, b2s, 2, tid);
    w16_commit(l2, W2im, 0, p.W2, p.ldW2, D, D, D, D, tid, K16_THREADS);
  }
  __syncthreads();

  f32x16 dW2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW2[i][j][r] = 0.f;
  float db2[1] = {0.f}, dgam[1] = {0.f}, dbet[1] = {0.f};

  const TaskRange tr = k16_tasks((unsigned)(p.ntiles * p.B), wave);
  i32x4 hdr = {0, 0, 0, 0}, hdr_n = {0, 0, 0, 0};
  HalfIdx i0 = {0, 0, 0}, i1 = {0, 0, 0};
  Pre16<false> nx;
  if (tr.first < tr.end) {
    hdr = tile_hdr(p, tr.first, tr.end);
    i0 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 0, lane);
    i1 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 1, lane);
    hdr_n = tile_hdr(p, tr.first + tr.stride, tr.end);
    const unsigned b0 = tr.first / (unsigned)p.ntiles;
    issue_pre16<false>(nx, i0.eid, p.e.ptr + (int64_t)b0 * p.e.bstride, p.e.ld, nullptr, 0, false, lane);
  }
  unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = STAMP ? __builtin_amdgcn_s_memtime() : 0;
  for (unsigned tt = tr.first; tt < tr.end; tt += tr.stride) {
    const unsigned b = tt / (unsigned)p.ntiles;
    const unsigned tn = tt + tr.stride < tr.end ? tt + tr.stride : tr.end - 1;
    const unsigned bn = tn / (unsigned)p.ntiles;
    const int ne = hdr.y - hdr.x, r0 = hdr.z, nr = hdr.w - hdr.z;
    const HalfIdx n0 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 0, lane);
    const HalfIdx n1 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 1, lane);
    const i32x4 hdr_nn = tile_hdr(p, tt + 2 * tr.stride, tr.end);
    const float* eb = p.e.ptr + (int64_t)b * p.e.bstride;
    const float* psb = p.ps.ptr + (int64_t)b * p.ps.bstride;
    const float* prb = p.pr.ptr + (int64_t)b * p.pr.bstride;
    const float* gab = q.g_agg.ptr + (int64_t)b * q.g_agg.bstride;
    float* gb = q.gpr + (int64_t)b * q.gpr_bstride;
    float carry = 0.f;
    int nseg = 0;
#pragma nounroll
    for (int hf = 0; hf < 2; ++hf) {
      const int nh = ne - 16 * hf < NLAM_T16 ? ne - 16 * hf : NLAM_T16;
      if (nh <= 0) break;   // wave-uniform
      STAMP16(9)    // loop overhead, index loads of the next tile
      const HalfIdx ix = pick_idx(hf == 0, i0, i1);
      const bool valid = t < nh;
      const int nfirst = hf == 0 ? __builtin_amdgcn_readfirstlane(i1.rcv) : -1;
      const unsigned ends = half_ends(ix.rcv, nfirst, hf, ne, lane);
      nseg += __popc(ends);
      const int ra = __builtin_amdgcn_readfirstlane(ix.rcv);
      const int nrecv = __builtin_amdgcn_readlane(ix.rcv, nh - 1) - ra + 1;
      // ---- this half's rows: Pe was requested a half ago; Ps (whole rows), Pr and g_agg (once
      // per receiver) now
      f32x4 h[NF], g[NF];
      {
        f32x4 psv[4], c[NF];
        RecvRows prv, gav;
        zero16<NF>(c);
        zero16<NF>(g);
        if constexpr (ABL == 1) {
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) psv[k2] = f32x4{(float)ix.snd, 1.f, 2.f, (float)lane};
          prv.v[0] = prv.v[1] = gav.v[0] = gav.v[1] = f32x4{(float)ra, 0.5f, 0.25f, (float)lane};
        } else {
          load_ps_rs(psv, psb, p.ps.ld, ix.snd, lane);
          recv_issue(prv, prb, p.pr.ld, ra, nrecv < 8 ? nrecv : 8, lane);
          recv_issue(gav, gab, q.g_agg.ld, ra, nrecv < 8 ? nrecv : 8, lane);
        }
        rs_to_acc16(h, nx.E, HS, LDT, lane);                                       // Pe
        STAMP16(0)   // prefetched Pe rows landed + transposed
        __builtin_amdgcn_sched_barrier(0);
        {
          f32x4 pa[NF];
          rs_to_acc16(pa, psv, HS, LDT, lane);
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) h[fb] += pa[fb];
        }
        __builtin_amdgcn_sched_barrier(0);
        recv_gather<true>(c, g, prv, gav, prb, p.pr.ld, gab, q.g_agg.ld, ra, nrecv, ix.rcv, HS, LDT, lane);
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) h[fb] += c[fb];                            // Pe + Ps + Pr
        __builtin_amdgcn_sched_barrier(0);
        STAMP16(1)   // Ps / Pr / g_agg rows landed, h complete
        {   // the next half's Pe rows (second half of this tile, or the next tile's first half)
          const bool more = hf == 0 && ne > NLAM_T16;
          if constexpr (ABL == 1) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) nx.E[k2] = f32x4{(float)(more ? i1.eid : n0.eid), 1.f, 2.f, 3.f};
          } else {
            issue_pre16<false>(nx, more ? i1.eid : n0.eid, more ? eb : p.e.ptr + (int64_t)bn * p.e.bstride,
                               p.e.ld, nullptr, 0, false, lane);
          }
        }
      }
      STAMP16(2)   // next prefetch issued
      // ---- s = silu(h) (and silu'(h) from the same sigmoid), z
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
      Frag16<2> fr;
      make_frag16<2, TERMS>(fr, h);
      frag16_to_planes<2, TERMS>(fr, TA, 0, lane);             // S stays in TA until dW2 is formed
      {
        f32x4 z[NF];
        vec_to_acc16<NF>(z, b2s, lane);
        if constexpr (ABL != 2) gemm_frag16<NF, 2, TERMS>(z, W2im, 0, 0, fr, lane);
        __builtin_amdgcn_sched_barrier(0);
        STAMP16(3)   // silu, S planes, second GEMM
        // ---- incoming gradient of the messages: scale * g_agg[rec]
        if (p.inv_deg != nullptr) {
          const float sc = p.inv_deg[ix.rcv];
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) g[fb] *= sc;
        }
        mask16<NF>(g, valid);   // padded slots carry a zero gradient: every sum below ignores them
        if constexpr (ABL != 2) {
          acc16_to_planes<NF, TERMS>(g, TB, 0, lane);             // dbeta summand
          wave_sync();
          colsum16<1, TERMS>(dbet, TB, 0, lane);
          wave_sync();
          ln16_bwd<NF, TERMS>(z, g, TB, gs, lane);                // g -> gz; g * xhat -> planes
          wave_sync();
          colsum16<1, TERMS>(dgam, TB, 0, lane);
          wave_sync();
        } else {
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) g[fb] += z[fb];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP16(4)   // g assembled, dbeta, LayerNorm backward, dgamma
      // ---- dW2 += gz (x) s, db2 += gz
      make_frag16<2, TERMS>(fr, g);
      frag16_to_planes<2, TERMS>(fr, TB, 0, lane);
      wave_sync();
      if constexpr (ABL != 2) {
        colsum16<1, TERMS>(db2, TB, 0, lane);
        outer_accum16<2, 2, TERMS>(dW2, TB, 0, TA, 0, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP16(5)   // GZ planes, db2, dW2 outer product
      // ---- gh = (W2^T gz) * silu'(h)
      f32x4 gh[NF];
      zero16<NF>(gh);
      if constexpr (ABL != 2) gemm_frag16_wt<NF, 2, TERMS>(gh, W2im, 0, 0, fr, lane);
      else {
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) gh[fb] = g[fb];
      }
#pragma unroll
      for (int fb = 0; fb < NF; ++fb)
        gh[fb] *= *reinterpret_cast<const f32x4*>(HS + t * LDT + 16 * fb + 4 * (lane >> 4));
      STAMP16(6)   // W2^T gz, silu'
      // gh -> the fp32 tile: whole-row stores to gh_out and the receiver-side sums (tile-local segments)
      wave_sync();
      acc16_to_tile<NF>(gh, HS, LDT, lane);
      wave_sync();
      if constexpr (ABL != 1)
        tile_store_rs(HS, LDT, q.gh_out + (int64_t)b * q.gh_bstride, D, ix.eid, nh, lane);
      half_segment_sums(HS, LDT, nh, ends, ix.rcv, lane, carry, [&](int r, float acc) {
        if constexpr (ABL != 1) gb[(int64_t)r * q.gpr_ld + lane] = acc;
        else asm volatile("" ::"v"(acc));
      });
      wave_sync();
      STAMP16(7)   // gh tile, gh_out stores, receiver-side sums
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
  if (STAMP && lane == 0) {
    STAMP16(10)   // (tail: waits of the last stores)
#pragma unroll
    for (int k = 0; k < 12; ++k) atomicAdd(&g_k16_stamps[k], st[k]);
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

// ---------------------------------------------------- update_edges backward, cooperative dW
// Two 64 x 64 weight-gradient accumulators (dW1e, dW2: 128 registers) do not fit beside the
// working set of a two-waves-per-SIMD kernel, and spilling them or forming dW1e in a second pass
// over gh_out / e costs vector-memory instructions, the resource this kernel is bound by
// (measured both ways).  So the two row-contracting products are shared by the WORKGROUP: per
// phase every wave writes the bf16 planes of ITS 16 rows into two 128-row plane pairs (P: S,
// later GH; Q: GZ, later E), and after a barrier wave w accumulates ONE 32 x 32 block ((w >> 1) &
// 1, w & 1) of dW2 (then of dW1e) over the 64 rows of its half ((w >> 2) * 64 ...): 2 x 16
// accumulator registers per wave instead of 128.  Four workgroup barriers per phase; every wave
// runs every phase (a wave without a task or without a second half runs it on clamped, valid
// rows with a zero gradient, which contributes exact zeros).
template <int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void edge_bwd16c_kernel(EdgeBwdParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64, NF = 4, LDT = D + 4, GR = NLAM_T16 * K16_NW;   // 128 rows per phase
  const EdgeFwdParams& p = q.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  char* cur = smem16;
  const B3Image W1im = w16_image(cur, D, D);
  cur += w16_image_bytes(D, D);
  const B3Image W2im = w16_image(cur, D, D);
  cur += w16_image_bytes(D, D);
  float* b2s = reinterpret_cast<float*>(cur);
  float* gs = b2s + D;
  cur += 2 * D * sizeof(float);
  constexpr int PP = p16_pitch(D);
  __bf16* Phi = reinterpret_cast<__bf16*>(cur);
  __bf16* Plo = Phi + GR * PP;
  __bf16* Qhi = Plo + GR * PP;
  __bf16* Qlo = Qhi + GR * PP;
  cur += (size_t)4 * GR * PP * sizeof(__bf16);
  float* HS = reinterpret_cast<float*>(cur) + wave * (NLAM_T16 * LDT);
  B3Tile Ps, Qs;                                  // this wave's 16 rows of the shared planes
  Ps.pitch = Qs.pitch = PP;
  Ps.hi = Phi + NLAM_T16 * wave * PP; Ps.lo = Plo + NLAM_T16 * wave * PP;
  Qs.hi = Qhi + NLAM_T16 * wave * PP; Qs.lo = Qlo + NLAM_T16 * wave * PP;
  load_weight_lds_b3(W1im, 0, p.W1e, p.ldW1e, D, D, D, D, tid, K16_THREADS);
  load_weight_lds_b3(W2im, 0, p.W2, p.ldW2, D, D, D, D, tid, K16_THREADS);
  load_vec_lds(b2s, p.b2, D, D, tid, K16_THREADS);
  load_vec_lds(gs, p.gamma, D, D, tid, K16_THREADS);
  __syncthreads();

  f32x16 dWa, dWb;   // this wave's block of dW2 / dW1e over its half of the rows
#pragma unroll
  for (int r = 0; r < 16; ++r) dWa[r] = dWb[r] = 0.f;
  float db2[1] = {0.f}, dgam[1] = {0.f}, dbet[1] = {0.f};
  const int ob_i = 32 * ((wave >> 1) & 1), ob_j = 32 * (wave & 1), orow = 64 * (wave >> 2);

  const bool has_geo = q.g_eout != nullptr;
  const TaskRange tr = k16_tasks((unsigned)(p.ntiles * p.B), wave);
  // workgroup-uniform trip count (wave 0's view of the chunk)
  const unsigned base = tr.first - (unsigned)wave;
  const unsigned niter = tr.end > base ? (tr.end - base + tr.stride - 1) / tr.stride : 0;
  i32x4 hdr = {0, 0, 0, 0}, hdr_n = {0, 0, 0, 0};
  HalfIdx i0 = {0, 0, 0}, i1 = {0, 0, 0};
  Pre16<true> nx;
  if (niter > 0) {
    hdr = tile_hdr(p, tr.first, tr.end);
    i0 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 0, lane);
    i1 = load_half_idx(p, hdr.x, hdr.y - hdr.x, 1, lane);
    hdr_n = tile_hdr(p, tr.first + tr.stride, tr.end);
    const unsigned t0 = tr.first < tr.end ? tr.first : tr.end - 1;
    const unsigned b0 = t0 / (unsigned)p.ntiles;
    issue_pre16<true>(nx, i0.eid, p.e.ptr + (int64_t)b0 * p.e.bstride, p.e.ld,
                      has_geo ? q.g_eout + (int64_t)b0 * q.geo_bstride : nullptr, q.geo_ld, has_geo,
                      lane);
  }
  for (unsigned it = 0; it < niter; ++it) {
    const unsigned tt = tr.first + it * tr.stride;
    const bool active = tt < tr.end;
    const unsigned tc = active ? tt : tr.end - 1;                 // (clamped: valid rows, zero gradient)
    const unsigned b = tc / (unsigned)p.ntiles;
    const unsigned tn = tt + tr.stride < tr.end ? tt + tr.stride : tr.end - 1;
    const unsigned bn = tn / (unsigned)p.ntiles;
    const int ne = active ? hdr.y - hdr.x : 0, r0 = hdr.z, nr = hdr.w - hdr.z;
    const HalfIdx n0 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 0, lane);
    const HalfIdx n1 = load_half_idx(p, hdr_n.x, hdr_n.y - hdr_n.x, 1, lane);
    const i32x4 hdr_nn = tile_hdr(p, tt + 2 * tr.stride, tr.end);
    const float* eb = p.e.ptr + (int64_t)b * p.e.bstride;
    const float* psb = p.ps.ptr + (int64_t)b * p.ps.bstride;
    const float* prb = p.pr.ptr + (int64_t)b * p.pr.bstride;
    const float* gab = q.g_agg.ptr + (int64_t)b * q.g_agg.bstride;
    const float* gob = has_geo ? q.g_eout + (int64_t)b * q.geo_bstride : nullptr;
    float* gb = q.gpr + (int64_t)b * q.gpr_bstride;
    float carry = 0.f;
    int nseg = 0;
#pragma nounroll
    for (int hf = 0; hf < 2; ++hf) {
      int nh = ne - 16 * hf < NLAM_T16 ? ne - 16 * hf : NLAM_T16;
      if (nh < 0) nh = 0;
      const HalfIdx ix = pick_idx(hf == 0, i0, i1);
      const bool valid = t < nh;
      const int nfirst = hf == 0 ? __builtin_amdgcn_readfirstlane(i1.rcv) : -1;
      const unsigned ends = half_ends(ix.rcv, nfirst, hf, ne, lane);
      nseg += __popc(ends);
      const int ra = __builtin_amdgcn_readfirstlane(ix.rcv);
      const int nrecv = nh > 0 ? __builtin_amdgcn_readlane(ix.rcv, nh - 1) - ra + 1 : 1;
      __syncthreads();   // every wave is done with the previous phase's reads of P / Q
      // ---- this phase's rows: e and g_e' were requested a phase ago; Ps (whole rows), Pr and
      // g_agg (once per receiver) now, landing under the first GEMM
      f32x4 h[NF], g[NF], geo[NF];
      Frag16<2> Ef;
      {
        f32x4 psv[4], c[NF];
        RecvRows prv, gav;
        zero16<NF>(c);
        zero16<NF>(g);
        load_ps_rs(psv, psb, p.ps.ld, ix.snd, lane);
        recv_issue(prv, prb, p.pr.ld, ra, nrecv < 8 ? nrecv : 8, lane);
        recv_issue(gav, gab, q.g_agg.ld, ra, nrecv < 8 ? nrecv : 8, lane);
        rs_to_acc16(h, nx.E, HS, LDT, lane);
        if (has_geo) rs_to_acc16(geo, nx.geo, HS, LDT, lane);
        else zero16<NF>(geo);
        make_frag16<2, TERMS>(Ef, h);
        zero16<NF>(h);
        // ---- recompute the forward: h (W1e e first: the node rows land meanwhile)
        gemm_frag16<NF, 2, TERMS>(h, W1im, 0, 0, Ef, lane);
        __builtin_amdgcn_sched_barrier(0);
        {
          f32x4 pa[NF];
          rs_to_acc16(pa, psv, HS, LDT, lane);
#pragma unroll
          for (int fb = 0; fb < NF; ++fb) h[fb] += pa[fb];
        }
        __builtin_amdgcn_sched_barrier(0);
        recv_gather<true>(c, g, prv, gav, prb, p.pr.ld, gab, q.g_agg.ld, ra, nrecv, ix.rcv, HS, LDT, lane);
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) h[fb] += c[fb];
        __builtin_amdgcn_sched_barrier(0);
        {   // the next phase's edge rows (second half of this tile, or the next tile's first half)
          const bool more = hf == 0;
          issue_pre16<true>(nx, more ? i1.eid : n0.eid, more ? eb : p.e.ptr + (int64_t)bn * p.e.bstride,
                            p.e.ld, has_geo ? (more ? gob : q.g_eout + (int64_t)bn * q.geo_bstride) : nullptr,
                            q.geo_ld, has_geo, lane);
        }
      }
      // ---- incoming gradient of the messages: scale * g_agg[rec] + g_e'
      if (p.inv_deg != nullptr) {
        const float sc = p.inv_deg[ix.rcv];
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) g[fb] *= sc;
      }
#pragma unroll
      for (int fb = 0; fb < NF; ++fb) g[fb] += geo[fb];
      mask16<NF>(g, valid);   // padded / absent rows carry a zero gradient
      // ---- s = silu(h) (silu'(h) from the same sigmoid), z
      {
        f32x4 ds[NF];   // silu'(h): parked in this wave's fp32 tile until the gh product
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
      Frag16<2> gzf;
      {
        f32x4 z[NF];
        vec_to_acc16<NF>(z, b2s, lane);
        {
          Frag16<2> fr;
          make_frag16<2, TERMS>(fr, h);                        // S
          gemm_frag16<NF, 2, TERMS>(z, W2im, 0, 0, fr, lane);
          frag16_to_planes<2, TERMS>(fr, Ps, 0, lane);         // (P / Q: free since the barrier)
        }
        __builtin_amdgcn_sched_barrier(0);
        acc16_to_planes<NF, TERMS>(g, Qs, 0, lane);            // dbeta summand
        wave_sync();
        colsum16<1, TERMS>(dbet, Qs, 0, lane);
        wave_sync();
        ln16_bwd<NF, TERMS>(z, g, Qs, gs, lane);               // g -> gz; g * xhat -> planes
        wave_sync();
        colsum16<1, TERMS>(dgam, Qs, 0, lane);
        wave_sync();
        make_frag16<2, TERMS>(gzf, g);                         // GZ
        frag16_to_planes<2, TERMS>(gzf, Qs, 0, lane);
      }
      __syncthreads();
      // ---- dW2 block += GZ^T S over this wave's 64 rows; db2 from its own 16
      colsum16<1, TERMS>(db2, Qs, 0, lane);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rw = orow + 16 * k;
        const bf16x8 ah = b3_tr_frag_rows(Qhi, PP, rw, ob_i, lane);
        const bf16x8 bh = b3_tr_frag_rows(Phi, PP, rw, ob_j, lane);
        dWa = B3_MFMA(ah, bh, dWa);
        if constexpr (TERMS == 3) {
          const bf16x8 al = b3_tr_frag_rows(Qlo, PP, rw, ob_i, lane);
          const bf16x8 bl = b3_tr_frag_rows(Plo, PP, rw, ob_j, lane);
          dWa = B3_MFMA(ah, bl, dWa);
          dWa = B3_MFMA(al, bh, dWa);
        }
      }
      // ---- gh = (W2^T gz) * silu'(h)
      f32x4 gh[NF];
      zero16<NF>(gh);
      gemm_frag16_wt<NF, 2, TERMS>(gh, W2im, 0, 0, gzf, lane);
#pragma unroll
      for (int fb = 0; fb < NF; ++fb)
        gh[fb] *= *reinterpret_cast<const f32x4*>(HS + t * LDT + 16 * fb + 4 * (lane >> 4));
      // gh -> the fp32 tile (in silu'(h)'s place): whole-row stores to gh_out and the
      // receiver-side sums (tile-local segments)
      wave_sync();
      acc16_to_tile<NF>(gh, HS, LDT, lane);
      wave_sync();
      tile_store_rs(HS, LDT, q.gh_out + (int64_t)b * q.gh_bstride, D, ix.eid, nh, lane);
      half_segment_sums(HS, LDT, nh, ends, ix.rcv, lane, carry, [&](int r, float acc) {
        gb[(int64_t)r * q.gpr_ld + lane] = acc;
      });
      wave_sync();
      Frag16<2> fr;
      make_frag16<2, TERMS>(fr, gh);                           // GH
      __syncthreads();   // every wave has formed its dW2 block from P / Q
      frag16_to_planes<2, TERMS>(fr, Ps, 0, lane);
      frag16_to_planes<2, TERMS>(Ef, Qs, 0, lane);
      __syncthreads();
      // ---- dW1e block += GH^T E
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rw = orow + 16 * k;
        const bf16x8 ah = b3_tr_frag_rows(Phi, PP, rw, ob_i, lane);
        const bf16x8 bh = b3_tr_frag_rows(Qhi, PP, rw, ob_j, lane);
        dWb = B3_MFMA(ah, bh, dWb);
        if constexpr (TERMS == 3) {
          const bf16x8 al = b3_tr_frag_rows(Plo, PP, rw, ob_i, lane);
          const bf16x8 bl = b3_tr_frag_rows(Qlo, PP, rw, ob_j, lane);
          dWb = B3_MFMA(ah, bl, dWb);
          dWb = B3_MFMA(al, bh, dWb);
        }
      }
      // ---- g_e = g_e' + W1e^T gh, as whole rows through the tile
      gemm_frag16_wt<NF, 2, TERMS>(geo, W1im, 0, 0, fr, lane);
      acc16_to_tile<NF>(geo, HS, LDT, lane);
      wave_sync();
      tile_store_rs(HS, LDT, q.g_e + (int64_t)b * q.ge_bstride, q.ge_ld, ix.eid, nh, lane);
      wave_sync();
    }
    if (active && nseg != nr) {   // (rare) receivers without in-edges inside the tile
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

  // ---- fold: the two row halves of every block (fixed order), then the per-feature sums
  __syncthreads();
  float* img = reinterpret_cast<float*>(smem16);   // [dW1e | dW2][row half][64][64]
  float* slab = q.slab + (int64_t)blockIdx.x * q.slab_stride;
  constexpr int nW = D * D;
  {
    const int hh = lane >> 5, j = lane & 31;
    float* i1p = img + (wave >> 2) * nW;
    float* i2p = img + 2 * nW + (wave >> 2) * nW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = ob_i + 8 * (r >> 2) + 4 * hh + (r & 3);
      i1p[i * D + ob_j + j] = dWb[r];
      i2p[i * D + ob_j + j] = dWa[r];
    }
  }
  __syncthreads();
  for (int i = tid; i < nW; i += K16_THREADS) {
    slab[i] = img[i] + img[nW + i];
    slab[nW + i] = img[2 * nW + i] + img[3 * nW + i];
  }
  __syncthreads();
  fold_vec_to_slab16<1, K16_NW>(db2, img, slab + 2 * nW, D, tid, wave, lane);
  fold_vec_to_slab16<1, K16_NW>(dgam, img, slab + 2 * nW + D, D, tid, wave, lane);
  fold_vec_to_slab16<1, K16_NW>(dbet, img, slab + 2 * nW + 2 * D, D, tid, wave, lane);
}

static int launch_edge_bwd16c(const EdgeBwdParams& q, hipStream_t s) {
  constexpr int D = 64;
  const size_t lds = 2 * w16_image_bytes(D, D) + 2 * D * sizeof(float) +
                     (size_t)4 * NLAM_T16 * K16_NW * p16_pitch(D) * sizeof(__bf16) +
                     (size_t)K16_NW * NLAM_T16 * (D + 4) * sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "edge_bwd16c: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = edge_bwd16c_kernel<3>;
  NLAM_BIG_LDS(kern, __func__);
  kern<<<(unsigned)nlam_bwd_grid(q.f.ntiles * q.f.B), K16_THREADS, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("edge_bwd16c_kernel");
  return 0;
}

static int launch_edge_bwd16(const EdgeBwdParams& q, hipStream_t s) {
  constexpr int D = 64;
  size_t lds = w16_image_bytes(D, D) + 2 * D * sizeof(float) + (size_t)K16_NW * 3 * p16_bytes(D);
  const size_t fold = (size_t)K16_NW * D * D * sizeof(float);
  if (fold > lds) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "edge_bwd16: LDS footprint %zu B exceeds 160 KiB", lds);
  static const bool stamp = getenv("NLAM_STAMP16") != nullptr;
  static const int abl = getenv("NLAM_ABL16") ? atoi(getenv("NLAM_ABL16")) : 0;
  auto kern = stamp ? edge_bwd16_kernel<3, true> : edge_bwd16_kernel<3, false>;
  if (abl == 1) kern = edge_bwd16_kernel<3, false, 1>;
  if (abl == 2) kern = edge_bwd16_kernel<3, false, 2>;
  NLAM_BIG_LDS(kern, __func__);
  // one slab per workgroup: the grid is what the host side sized the slabs for
  kern<<<(unsigned)nlam_bwd_grid(q.f.ntiles * q.f.B), K16_THREADS, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("edge_bwd16_kernel");
  return 0;
}

int nlam_k16_edge_bwd(const EdgeBwdParams& q, int has_egemm, hipStream_t s) {
  if (q.f.e.width != 64 || !nlam_mfma_b3() || getenv("NLAM_STAMP") != nullptr) return -1;
  if (!nlam_k16_on(has_egemm ? K16_EDGE_BWD_UPD : K16_EDGE_BWD)) return -1;
  return has_egemm ? launch_edge_bwd16c(q, s) : launch_edge_bwd16(q, s);
}
