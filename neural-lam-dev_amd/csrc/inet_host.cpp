// Host sequencer: ONE C call per InteractionNet forward / backward (hidden width 64).
//
// Replaces the body of the reference's InteractionNet.forward (interaction_net.py:86-115:
// propagate -> message :117-121 -> aggregate :124-131 -> aggr_mlp + residual :106-113) and its
// autograd, for the configurations every BASELINE model uses at hidden 64: hidden_layers = 1,
// plain (non-split) MLPs, in-degree <= 32, shared or separate sender / receiver nodes,
// update_edges on / off, batch-invariant operands (the reference's expand_to_batch views,
// ar_model.py:204-209).  The launches are the C entry points of this same library in the order
// neural_lam_amd/fused.py issues them one ctypes call at a time; a Python caller pays ~40 us of
// interpreter time per launch, which makes eager (non-graph) steps host-bound, so the sequence
// lives here.  Stateless: the caller allocates every output and the workspace.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/nlam_hip.h"
#include "nlam_common.h"

namespace {

constexpr int D = 64;

struct Carve {   // bump allocator over the caller's workspace (256-byte aligned blocks)
  float* base;
  int64_t used = 0, cap;
  Carve(float* b, int64_t c) : base(b), cap(c) {}
  float* take(int64_t n) {
    const int64_t a = (n + 63) & ~int64_t(63);
    float* p = base ? base + used : nullptr;
    used += a;
    return p;
  }
};

struct Segs {   // pending slab reductions of the layer: one nlam_reduce_slabs_batch at the end
  static constexpr int MAXN = 64;
  int n = 0;
  const float* slab[MAXN];
  int64_t nslabs[MAXN], stride[MAXN], src_off[MAXN], src_ld[MAXN], dst_ld[MAXN];
  int32_t rows[MAXN], cols[MAXN];
  float* dst[MAXN];
  void add(const float* s, int64_t ns, int64_t st, int64_t off, int r, int c, int64_t sld, float* d,
           int64_t dld) {
    if (d == nullptr) return;
    slab[n] = s; nslabs[n] = ns; stride[n] = st; src_off[n] = off; rows[n] = r; cols[n] = c;
    src_ld[n] = sld; dst[n] = d; dst_ld[n] = dld;
    ++n;
  }
};

int64_t ntiles32(int64_t B, int64_t rows) { return B * ((rows + 31) / 32); }

// which of the two supported shapes (0 = unsupported)
int inet_case(const nlam_inet_args* a) {
  if (a == nullptr || a->d != D || a->B < 1) return 0;
  if (!nlam_lin_multi_supported() || !nlam_node_chain_supported()) return 0;
  if (a->g.ntiles <= 0 || a->g.M < 1) return 0;
  const bool same = a->rec.ptr == nullptr;
  if (same) {
    if (a->send.B != a->B || a->n_send_rows != a->g.n_rec || a->g.n_send > a->g.n_rec) return 0;
    return 1;
  }
  return 2;
}

}  // namespace

extern "C" int64_t nlam_sizeof_inet_args(void) { return (int64_t)sizeof(nlam_inet_args); }
extern "C" int64_t nlam_sizeof_inet_grads(void) { return (int64_t)sizeof(nlam_inet_grads); }
extern "C" int nlam_inet_supported(const nlam_inet_args* a) { return inet_case(a) != 0 ? 1 : 0; }

extern "C" int nlam_inet_fwd(const nlam_inet_args* a, void* stream) {
  const int kind = inet_case(a);
  NLAM_REQUIRE(kind != 0, "nlam_inet_fwd: configuration not covered (nlam_inet_supported())");
  const nlam_inet_graph& g = a->g;
  const nlam_inet_weights& w = a->w;
  const bool same = kind == 1, upd = a->update_edges != 0;
  const int64_t B = a->B, N_s = a->n_send_rows, N_r = g.n_rec, M = g.M;
  const float* W1e = w.W1;
  const float* W1s = w.W1 + D;
  const float* W1r = w.W1 + 2 * D;
  const float* inv_deg = a->mean ? g.inv_deg : nullptr;
  NLAM_REQUIRE(a->P && a->agg && a->rec_out && (same || a->Pr) && (upd ? a->e_out != nullptr : a->Pe != nullptr),
               "nlam_inet_fwd: NULL output buffer");
  int rc;
  const float *ps, *pr;
  int64_t ps_bs, ps_ld, pr_bs, pr_ld;
  if (same) {
    // P = [x W1s^T | x W1r^T + b1]
    rc = nlam_lin_fwd(a->send.ptr, a->send.bstride, a->send.ld, D, W1s, w.ldW1, nullptr, D, W1r, w.ldW1,
                      w.b1, D, a->P, N_s * 2 * D, 2 * D, a->send.B, N_s, 0, stream);
    if (rc) return rc;
    ps = a->P; pr = a->P + D; ps_bs = pr_bs = N_s * 2 * D; ps_ld = pr_ld = 2 * D;
  } else {
    const float* x[3] = {a->send.ptr, a->rec.ptr, a->edge.ptr};
    const int64_t xbs[3] = {a->send.bstride, a->rec.bstride, a->edge.bstride};
    const int64_t xld[3] = {a->send.ld, a->rec.ld, a->edge.ld};
    const float* Wk[3] = {W1s, W1r, W1e};
    const int64_t ldw[3] = {w.ldW1, w.ldW1, w.ldW1};
    const float* bias[3] = {nullptr, w.b1, nullptr};
    float* out[3] = {a->P, a->Pr, a->Pe};
    const int64_t Bk[3] = {a->send.B, a->rec.B, a->edge.B};
    // (a problem with zero rows is skipped by the multi-problem launch: a projection the caller
    // already holds -- nlam_grid_encode_fwd wrote it -- is not recomputed)
    const int64_t rows[3] = {a->ps_given ? 0 : N_s, a->pr_given ? 0 : N_r, M};
    const int64_t obs[3] = {Bk[0] > 1 ? N_s * D : 0, Bk[1] > 1 ? N_r * D : 0, Bk[2] > 1 ? M * D : 0};
    const int64_t old_[3] = {D, D, D};
    rc = nlam_lin_fwd_multi(upd ? 2 : 3, D, x, xbs, xld, Wk, ldw, bias, out, obs, old_, Bk, rows, 0, stream);
    if (rc) return rc;
    ps = a->P; pr = a->Pr;
    ps_bs = a->send.B > 1 ? N_s * D : 0; pr_bs = a->rec.B > 1 ? N_r * D : 0;
    ps_ld = pr_ld = D;
  }
  if (same && !upd) {
    rc = nlam_lin_fwd(a->edge.ptr, a->edge.bstride, a->edge.ld, D, W1e, w.ldW1, nullptr, D, nullptr, 0,
                      nullptr, 0, a->Pe, a->edge.B > 1 ? M * D : 0, D, a->edge.B, M, 0, stream);
    if (rc) return rc;
  }
  if (upd) {
    rc = nlam_edge_fwd(g.tiles, g.ntiles, g.csr_rowptr, g.csr_eid, g.csr_send, g.csr_rec, inv_deg,
                       a->edge.ptr, a->edge.bstride, a->edge.ld, 1, ps, ps_bs, ps_ld, pr, pr_bs, pr_ld,
                       W1e, w.ldW1, w.W2, w.ldW2, w.b2, w.gam, w.bet, a->agg, N_r * D, D, a->e_out,
                       M * D, D, B, D, stream);
  } else {
    rc = nlam_edge_fwd(g.tiles, g.ntiles, g.csr_rowptr, g.csr_eid, g.csr_send, g.csr_rec, inv_deg,
                       a->Pe, a->edge.B > 1 ? M * D : 0, D, 0, ps, ps_bs, ps_ld, pr, pr_bs, pr_ld,
                       nullptr, 0, w.W2, w.ldW2, w.b2, w.gam, w.bet, a->agg, N_r * D, D, nullptr, 0, 0,
                       B, D, stream);
  }
  if (rc) return rc;
  const nlam_inet_view& r = same ? a->send : a->rec;
  return nlam_mlp_fwd(r.ptr, r.bstride, r.ld, D, a->agg, N_r * D, D, D, w.V1, w.ldV1, w.c1, w.V2,
                      w.ldV2, w.c2, w.gam2, w.bet2, r.ptr, r.bstride, r.ld, a->rec_out, N_r * D, D, B,
                      N_r, D, D, stream);
}

// Workspace floats nlam_inet_bwd needs for this configuration (0 = not covered).
static int64_t inet_bwd_plan(const nlam_inet_args* a, const nlam_inet_grads* gr, float* ws, int64_t cap,
                             void* stream, bool run);

extern "C" int64_t nlam_inet_bwd_workspace(const nlam_inet_args* a) {
  if (inet_case(a) == 0) return 0;
  return inet_bwd_plan(a, nullptr, nullptr, 0, nullptr, false);
}

extern "C" int nlam_inet_bwd(const nlam_inet_args* a, const nlam_inet_grads* gr, float* ws,
                             int64_t ws_floats, void* stream) {
  NLAM_REQUIRE(inet_case(a) != 0, "nlam_inet_bwd: configuration not covered (nlam_inet_supported())");
  NLAM_REQUIRE(gr != nullptr && ws != nullptr, "nlam_inet_bwd: NULL gradients / workspace");
  NLAM_REQUIRE(ws_floats >= inet_bwd_plan(a, nullptr, nullptr, 0, nullptr, false),
               "nlam_inet_bwd: workspace too small (nlam_inet_bwd_workspace())");
  const int64_t rc = inet_bwd_plan(a, gr, ws, ws_floats, stream, true);
  return rc < 0 ? (int)-rc : 0;
}

// One routine carves the workspace and (run = true) issues the launches, so that the size query
// and the execution cannot disagree.  Returns the floats used, or -(error code) when running.
static int64_t inet_bwd_plan(const nlam_inet_args* a, const nlam_inet_grads* gr, float* ws, int64_t cap,
                             void* stream, bool run) {
  const int kind = inet_case(a);
  const nlam_inet_graph& g = a->g;
  const nlam_inet_weights& w = a->w;
  const bool same = kind == 1, upd = a->update_edges != 0;
  const int64_t B = a->B, N_s = a->n_send_rows, N_r = g.n_rec, M = g.M;
  const int64_t Bs = a->send.B, Br = same ? a->send.B : a->rec.B, Be = a->edge.B;
  const float* W1e = w.W1;
  const float* W1s = w.W1 + D;
  const float* W1r = w.W1 + 2 * D;
  const float* inv_deg = a->mean ? g.inv_deg : nullptr;
  const nlam_inet_view& rv = same ? a->send : a->rec;
  Carve c(ws, cap);
  Segs segs;
  int rc = 0;
#define RUN(call)                 \
  do {                            \
    if (run) {                    \
      rc = (call);                \
      if (rc) return -(int64_t)rc; \
    }                             \
  } while (0)

  // 1. node update backward: gradient on x_r (residual included) and on the aggregate
  float* g_rec_tmp = c.take(B * N_r * D);
  float* g_agg = c.take(B * N_r * D);
  float* ga = c.take(B * N_r * D);
  const int64_t st1 = nlam_mlp_bwd_slab_stride(2 * D, D, D), ns1 = nlam_bwd_grid(ntiles32(B, N_r));
  float* slab1 = c.take(ns1 * st1);
  const float* g_rec_out = run ? gr->g_rec_out : nullptr;
  if (run && g_rec_out == nullptr) return -4;   // (the caller passes zeros when no gradient arrived)
  RUN(nlam_mlp_bwd(rv.ptr, rv.bstride, rv.ld, D, a->agg, N_r * D, D, D, w.V1, w.ldV1, w.c1, w.V2, w.ldV2,
                   w.c2, w.gam2, g_rec_out, N_r * D, D, g_rec_tmp, N_r * D, D, g_agg, N_r * D, D, 1,
                   slab1, st1, ga, B, N_r, D, D, stream));
  {
    const int64_t o2 = (int64_t)D * 2 * D + D, ov = o2 + D * D;
    segs.add(slab1, ns1, st1, o2, D, D, D, run ? gr->dV2 : (float*)1, D);
    segs.add(slab1, ns1, st1, ov, 1, D, D, run ? gr->dc2 : (float*)1, D);
    segs.add(slab1, ns1, st1, ov + D, 1, D, D, run ? gr->dgam2 : (float*)1, D);
    segs.add(slab1, ns1, st1, ov + 2 * D, 1, D, D, run ? gr->dbet2 : (float*)1, D);
  }
  const float* g_res = g_rec_tmp;   // gradient already on x_r, per sample or summed over the batch
  int64_t g_res_bs = N_r * D;
  if (Br == 1 && B > 1) {
    float* t3 = c.take(N_r * D);
    RUN(nlam_sum_batch(g_rec_tmp, N_r * D, t3, B, N_r * D, stream));
    g_res = t3;
    g_res_bs = 0;
  }

  // 2. edge backward
  // (grid-side nets with a batch-invariant edge term: per-tile sender partials instead of gh rows)
  const bool parts = !same && !upd && Be == 1 && B > 1 && g.part_slot != nullptr &&
                     g.pcsc_colptr != nullptr && g.pcsc_rows != nullptr &&
                     nlam_edge_bwd_parts_supported(g.ntiles, B, D) != 0;
  float* gh = parts ? c.take(B * g.ntiles * 16 * D) : c.take(B * M * D);
  const int64_t gh_bs = parts ? g.ntiles * 16 * D : M * D;
  float* gP = c.take(B * N_r * (same ? 2 * D : D));     // same: [gPs | gPr]; else gPr
  float* gpr = same ? gP + D : gP;
  const int64_t gpr_bs = N_r * (same ? 2 * D : D), gpr_ld = same ? 2 * D : D;
  float* g_e = nullptr;
  if (upd) g_e = (Be == 1 && B > 1) ? c.take(B * M * D) : (run ? gr->g_edge : nullptr);
  const bool bsum = !upd && Be == 1 && B > 1 && nlam_edge_bwd_forms_batch_sum(g.ntiles, B, D) != 0;
  float* dPe1 = bsum ? c.take(M * D) : nullptr;
  const int64_t st2 = nlam_edge_bwd_slab_stride(D), ns2 = nlam_bwd_grid(B * g.ntiles);
  float* slab2 = c.take(ns2 * st2);
  const float *ps, *pr;
  int64_t ps_bs, ps_ld, pr_bs, pr_ld;
  if (same) {
    ps = a->P; pr = a->P + D; ps_bs = pr_bs = N_s * 2 * D; ps_ld = pr_ld = 2 * D;
  } else {
    ps = a->P; pr = a->Pr; ps_bs = Bs > 1 ? N_s * D : 0; pr_bs = Br > 1 ? N_r * D : 0; ps_ld = pr_ld = D;
  }
  if (upd) {
    RUN(nlam_edge_bwd(g.tiles, g.ntiles, g.csr_rowptr, g.csr_eid, g.csr_send, g.csr_rec, inv_deg,
                      a->edge.ptr, a->edge.bstride, a->edge.ld, 1, ps, ps_bs, ps_ld, pr, pr_bs, pr_ld, W1e,
                      w.ldW1, w.W2, w.ldW2, w.b2, w.gam, g_agg, N_r * D, D, gr->g_edge_out, M * D, D, gh,
                      M * D, gpr, gpr_bs, gpr_ld, g_e, M * D, D, slab2, st2, B, D, stream));
  } else if (parts) {
    RUN(nlam_edge_bwd_parts(g.tiles, g.ntiles, g.csr_rowptr, g.csr_eid, g.csr_send, g.csr_rec, inv_deg, a->Pe,
                            D, ps, ps_bs, ps_ld, pr, pr_bs, pr_ld, w.W2, w.ldW2, w.b2, w.gam, g_agg, N_r * D, D,
                            g.part_slot, gh, gh_bs, gpr, gpr_bs, gpr_ld, dPe1, D, slab2, st2, B, D, stream));
  } else {
    // batch-invariant edge term: dPe = sum_b gh[b] comes out of the edge kernel (nlam_hip.h)
    RUN(nlam_edge_bwd(g.tiles, g.ntiles, g.csr_rowptr, g.csr_eid, g.csr_send, g.csr_rec, inv_deg, a->Pe,
                      Be > 1 ? M * D : 0, D, 0, ps, ps_bs, ps_ld, pr, pr_bs, pr_ld, nullptr, 0, w.W2, w.ldW2,
                      w.b2, w.gam, g_agg, N_r * D, D, nullptr, 0, 0, gh, M * D, gpr, gpr_bs, gpr_ld, dPe1,
                      0, D, slab2, st2, B, D, stream));
  }
  {
    // slab layout of nlam_edge_bwd: [dW1e d x d | dW2 d x d | db2 | dgamma | dbeta]
    float* dW1 = run ? gr->dW1 : (float*)1;
    if (upd) segs.add(slab2, ns2, st2, 0, D, D, D, dW1, 3 * D);
    segs.add(slab2, ns2, st2, D * D, D, D, D, run ? gr->dW2 : (float*)1, D);
    segs.add(slab2, ns2, st2, 2 * D * D, 1, D, D, run ? gr->db2 : (float*)1, D);
    segs.add(slab2, ns2, st2, 2 * D * D + D, 1, D, D, run ? gr->dgam : (float*)1, D);
    segs.add(slab2, ns2, st2, 2 * D * D + 2 * D, 1, D, D, run ? gr->dbet : (float*)1, D);
  }

  // 3. node side: sender gather + projections backward + the 128-wide weight gradients
  if (same) {
    RUN(nlam_node_bwd(gh, M * D, g.csc_colptr, g.csc_eid, g.n_send, gP, N_r * 2 * D, 2 * D, g_res, g_res_bs,
                      D, W1s, w.ldW1, W1r, w.ldW1, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr,
                      0, nullptr, nullptr, gr->g_send, N_s * D, D, nullptr, 0, 0, nullptr, nullptr, 0, B, N_s,
                      stream));
    const int64_t st3 = nlam_node_outer_slab_stride(), ns3 = nlam_node_outer_grid(B, N_s);
    float* slab3 = c.take(ns3 * st3);
    RUN(nlam_node_outer(ga, a->send.ptr, a->send.bstride, a->send.ld, a->agg, N_r * D, D, gP, N_r * 2 * D,
                        2 * D, a->send.ptr, a->send.bstride, a->send.ld, slab3, st3, B, N_s, stream));
    float* dW1 = run ? gr->dW1 : (float*)1;
    const int64_t o = 2 * D * D + D;
    segs.add(slab3, ns3, st3, 0, D, 2 * D, 2 * D, run ? gr->dV1 : (float*)1, 2 * D);
    segs.add(slab3, ns3, st3, 2 * D * D, 1, D, D, run ? gr->dc1 : (float*)1, D);
    segs.add(slab3, ns3, st3, o, D, D, D, run ? dW1 + D : (float*)1, 3 * D);
    segs.add(slab3, ns3, st3, o + D * D, D, D, D, run ? dW1 + 2 * D : (float*)1, 3 * D);
    segs.add(slab3, ns3, st3, o + 2 * D * D + D, 1, D, D, run ? gr->db1 : (float*)1, D);
  }
  // multi-problem projection backward: sender (gather), receiver, edge (update_edges off), and
  // the deferred dV1 of the node update (separate nodes); the edge third alone when nodes are shared
  {
    const float *x[4], *xb[4], *gy[4], *Wk[4], *gxa[4], *ghp[4];
    float *gx[4], *slab[4];
    int64_t xbs[4], xld[4], xbbs[4], xbld[4], gybs[4], gyld[4], ldw[4], gxbs[4], gxld[4], gabs[4], gald[4],
        nsum[4], sstr[4], ghbs[4], nsend[4], sst[4], Bk[4], rows[4];
    const int32_t *colp[4], *eidp[4];
    int n = 0;
    auto push = [&](const float* x_, int64_t xbs_, int64_t xld_, const float* xb_, int64_t xbbs_,
                    const float* gy_, int64_t gybs_, const float* W_, float* gx_, int64_t gxbs_,
                    const float* ga_, int64_t gabs_, int64_t nsum_, int64_t sstr_, const float* gh_,
                    int64_t Bk_, int64_t rows_, float* dW, int64_t dW_ld, float* db, int kcols) {
      const int64_t st = (int64_t)D * kcols + D, ns = nlam_bwd_grid(ntiles32(Bk_, rows_));
      float* sl = c.take(ns * st);
      x[n] = x_; xbs[n] = xbs_; xld[n] = xld_; xb[n] = xb_; xbbs[n] = xbbs_; xbld[n] = D;
      gy[n] = gy_; gybs[n] = gybs_; gyld[n] = D; Wk[n] = W_; ldw[n] = w.ldW1;
      gx[n] = gx_; gxbs[n] = gxbs_; gxld[n] = D; gxa[n] = ga_; gabs[n] = gabs_; gald[n] = D;
      nsum[n] = nsum_; sstr[n] = sstr_; ghp[n] = gh_; ghbs[n] = gh_bs;
      colp[n] = gh_ ? (parts ? g.pcsc_colptr : g.csc_colptr) : nullptr;
      eidp[n] = gh_ ? (parts ? g.pcsc_rows : g.csc_eid) : nullptr; nsend[n] = gh_ ? g.n_send : 0;
      slab[n] = sl; sst[n] = st; Bk[n] = Bk_; rows[n] = rows_;
      segs.add(sl, ns, st, 0, D, kcols, kcols, dW, dW_ld);
      segs.add(sl, ns, st, (int64_t)D * kcols, 1, D, D, db, D);
      ++n;
    };
    float* dW1 = run ? gr->dW1 : (float*)1;
    if (!same) {
      const bool fs = Bs == 1 && B > 1, fr = Br == 1 && B > 1;
      push(a->send.ptr, a->send.bstride, a->send.ld, nullptr, 0, nullptr, 0, W1s, run ? gr->g_send : nullptr,
           Bs > 1 ? N_s * D : 0, run ? gr->g_send_add : nullptr, Bs > 1 ? N_s * D : 0, fs ? B : 1, 0, gh,
           fs ? 1 : B, N_s, run ? dW1 + D : (float*)1, 3 * D, nullptr, D);
      push(a->rec.ptr, a->rec.bstride, a->rec.ld, nullptr, 0, gpr, fr ? 0 : N_r * D, W1r,
           run ? gr->g_rec : nullptr, Br > 1 ? N_r * D : 0, g_res, g_res_bs, fr ? B : 1, fr ? N_r * D : 0,
           nullptr, fr ? 1 : B, N_r, run ? dW1 + 2 * D : (float*)1, 3 * D, run ? gr->db1 : (float*)1, D);
    }
    if (!upd) {
      const bool fe = Be == 1 && B > 1;   // (bsum: gy = dPe1, already summed over the batch)
      push(a->edge.ptr, a->edge.bstride, a->edge.ld, nullptr, 0, bsum ? dPe1 : gh, fe ? 0 : M * D, W1e,
           run ? gr->g_edge : nullptr, Be > 1 ? M * D : 0, nullptr, 0, (fe && !bsum) ? B : 1,
           (fe && !bsum) ? M * D : 0, nullptr, fe ? 1 : B, M, dW1, 3 * D, nullptr, D);
    }
    if (!same)   // deferred dV1 = ga^T [x_r | agg], dc1 = colsum ga
      push(a->rec.ptr, a->rec.bstride, a->rec.ld, a->agg, N_r * D, ga, N_r * D, nullptr, nullptr, 0, nullptr, 0,
           1, 0, nullptr, B, N_r, run ? gr->dV1 : (float*)1, 2 * D, run ? gr->dc1 : (float*)1, 2 * D);
    // (separate nodes: the batch sum of an edge-updating layer's g_e is issued before the
    // projections, the order of neural_lam_amd/fused.py -- tools/site_stats.py walks rocprof
    // traces against that order)
    if (!same && upd && Be == 1 && B > 1) RUN(nlam_sum_batch(g_e, M * D, gr->g_edge, B, M * D, stream));
    if (n > 0)
      RUN(nlam_lin_bwd_multi(n, D, x, xbs, xld, xb, xbbs, xbld, gy, gybs, gyld, Wk, ldw, gx, gxbs, gxld, gxa,
                             gabs, gald, nsum, sstr, ghp, ghbs, colp, eidp, nsend, slab, sst, Bk, rows,
                             stream));
  }
  // 4. batch-invariant edge input of an edge-updating layer: its gradient sums over the batch
  if (same && upd && Be == 1 && B > 1) RUN(nlam_sum_batch(g_e, M * D, gr->g_edge, B, M * D, stream));
  // 5. every parameter gradient of the layer: one reduction launch
  if (run && segs.n > 0)
    RUN(nlam_reduce_slabs_batch(segs.n, segs.slab, segs.nslabs, segs.stride, segs.src_off, segs.rows,
                                segs.cols, segs.src_ld, segs.dst, segs.dst_ld, stream));
#undef RUN
  return c.used;
}
