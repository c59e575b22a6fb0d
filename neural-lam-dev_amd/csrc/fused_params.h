// Parameter blocks of the fused hidden-64 kernels, shared by the 32-row kernels
// (fused_mlp.hip, fused_edge.hip) and the 16-row, two-waves-per-SIMD kernels (fused16_*.hip).
#pragma once
#include "fused_common.h"

struct MlpParams {
  RowView src[2];
  int nsrc;
  int k_in;             // sum of source widths (W1 is hid x k_in)
  int k_pad;            // k_in rounded up to a multiple of 8
  int n_out;            // true output width (<= 32 * NOUTB)
  const float* W1; int64_t ldW1; const float* b1;
  const float* W2; int64_t ldW2; const float* b2;
  const float* gamma; const float* beta;
  const float* res; int64_t res_bstride; int64_t res_ld;  // optional residual
  float* out; int64_t out_bstride; int64_t out_ld;
  int64_t rows;         // rows per batch item
  int B;
  int vec_mask;         // bit s: source s may use float4 loads; bit 2: res; bit 3: out
};

struct LinParams {
  RowView x;
  int k_pad;
  const float* WA; int64_t ldWA; const float* bA; int nA;
  const float* WB; int64_t ldWB; const float* bB; int nB;
  float* out; int64_t out_bstride; int64_t out_ld;
  int64_t rows; int B; int vec_mask;  // bit0: x, bit3: out
  int timeline;                       // NLAM_TIMELINE=1: per-workgroup 100 MHz stamps
};

struct MlpBwdParams {
  MlpParams f;                 // forward operands (out/res unused)
  RowView gy;                  // (B, rows, n_out)
  float* gxa; int64_t gxa_bstride; int64_t gxa_ld;   // optional grads of the sources
  float* gxb; int64_t gxb_bstride; int64_t gxb_ld;
  int add_gy_to_gxa;           // residual taken from source a: gxa += gy
  float* slab; int64_t slab_stride;
  float* ga_out;               // DEFER_DW1: (B, rows, HID) gradient of the hidden pre-activation
  int vec_gy, vec_gxa, vec_gxb;
  int stamp;                   // NLAM_STAMP=1: per-phase s_memtime sums (diagnostic)
};

struct LinBwdParams {
  RowView x;                   // (B, rows, k_in)
  RowView gy;                  // (B, rows, n_out)
  const float* WA; int64_t ldWA; int nA;
  const float* WB; int64_t ldWB; int nB;
  float* gx; int64_t gx_bstride; int64_t gx_ld;   // optional
  const float* gx_add; int64_t ga_bstride; int64_t ga_ld;   // optional addend of gx
  float* slab; int64_t slab_stride;
  int64_t rows; int B;
  int gy_nsum; int64_t gy_sum_stride;   // gy[b] := sum_{s < gy_nsum} gy[b][s * gy_sum_stride + ...]
  int vec_x, vec_gy, vec_gx;
  // 16-row kernels only: columns [0, 64) of gy are NOT read but formed as the sum of the rows
  // gh[eid] over the row's sender list (the sender-side scatter of the edge gradient as a gather;
  // gh: (B, M, 64) in edge order, batch pitch gh_bstride -- also the pitch of the gy_nsum slices)
  const float* gh; int64_t gh_bstride;
  const int32_t* csc_colptr; const int32_t* csc_eid; int n_send;
};

struct OuterParams {
  RowView g;                   // (B, rows, 32 NGB)
  RowView xa, xb;              // widths sum to <= 32 NXB; xb.ptr may be NULL
  const int32_t* x_index;      // optional: x row of (batch-local) row r is x_index[r]
  float* slab; int64_t slab_stride;
  int64_t rows; int B;
};

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
  int wave_major;             // edge_fwd: task numbering of the persistent waves (see the kernel)
};

struct EdgeBwdParams {
  EdgeFwdParams f;              // forward operands (agg / e_out unused)
  RowView g_agg;                // (B, N_r, d)
  const float* g_eout; int64_t geo_bstride; int64_t geo_ld;   // (B, M, d) original order, may be NULL
  float* gh_out; int64_t gh_bstride;                          // (B, M, d) original edge order, pitch d
  float* gpr; int64_t gpr_bstride; int64_t gpr_ld;            // (B, N_r, d)
  float* g_e; int64_t ge_bstride; int64_t ge_ld;              // (B, M, d) original order (has_egemm)
  float* slab; int64_t slab_stride;
  // optional (batch-sum form only): per-tile sender partial sums of gh INSTEAD of the gh rows.
  // part_slot[pos] = rank of the slot's sender among the tile's distinct senders (< 16, CSR
  // position order) | number of distinct senders of the tile << 8; row 16 * tile + slot of gpart (B, 16 * ntiles, d) receives the sum of the
  // tile's gh rows of that sender.  The sender-side reduction then reads one row per (tile, sender)
  // pair instead of one per edge (m2g: 76 k instead of 255 k per sample).
  const int32_t* part_slot; float* gpart; int64_t gpart_bstride;
};

// ---- node-side kernels of a chain of InteractionNets on shared nodes (fused16_node.hip)
struct NodeFwdParams {
  RowView x, agg;               // (B, N, 64): layer input, aggregated messages
  const float* V1; int64_t ldV1; const float* c1;    // aggregation MLP (64 x 128, 64 x 64)
  const float* V2; int64_t ldV2; const float* c2;
  const float* gamma; const float* beta;
  float* xout; int64_t xo_bstride; int64_t xo_ld;    // x' = x + LN(...)
  const float* WA; int64_t ldWA; const float* bA;    // NEXT layer's sender / receiver projections
  const float* WB; int64_t ldWB; const float* bB;
  float* P; int64_t p_bstride; int64_t p_ld;         // (B, N, 128) = [x' WA^T + bA | x' WB^T + bB]; NULL: none
  int64_t rows; int B;
};

struct NodeBwdParams {
  const float* gh; int64_t gh_bstride;               // (B, M, 64) edge-MLP hidden gradients, edge order
  const int32_t* csc_colptr; const int32_t* csc_eid; // sender lists (n_send + 1, M)
  int n_send;
  float* gP; int64_t gp_bstride; int64_t gp_ld;      // (B, N, 128): [gPs written here | gPr read]
  const float* g_res; int64_t gr_bstride; int64_t gr_ld;   // gradient already on x_l
  const float* WA; int64_t ldWA; const float* WB; int64_t ldWB;
  RowView x, agg;                                    // layer l-1 (x.ptr == NULL: no node update below)
  const float* V1; int64_t ldV1; const float* c1;
  const float* V2; int64_t ldV2; const float* c2;
  const float* gamma;
  float* gx_out; int64_t gx_bstride; int64_t gx_ld;        // gradient on x_{l-1} (or on x_l without update)
  float* gagg_out; int64_t gagg_bstride; int64_t gagg_ld;  // gradient on agg_{l-1}
  float* ga_out;                                     // (B, N, 64) hidden gradient of the node update
  float* slab; int64_t slab_stride;
  int64_t rows; int B;
};

struct NodeOuterParams {
  RowView ga, xa, xb;           // role A: dV1 = ga^T [xa | xb]
  RowView gP, xl;               // role B: dWp = gP^T xl
  float* slab; int64_t slab_stride;
  int64_t rows; int B; int has_a;
};

// ---- 16-row kernels (fused16_*.hip): same parameter blocks, same slab layouts and grids.
// Each returns -1 when the shape / alignment is not one it handles (the caller then continues
// with the 32-row kernel), 0 on success, > 0 on error.
int nlam_k16_mlp_bwd(const MlpBwdParams& q, hipStream_t s);
int nlam_k16_lin_bwd(const LinBwdParams& q, hipStream_t s);
// several independent projections (n_out = 64 each) in one launch: q[k].slab holds
// nlam_bwd_grid(ceil(rows_k / 32) * B_k) slabs; 0 ok, -1 = a shape the 16-row kernels do not take
int nlam_k16_lin_bwd_multi(const LinBwdParams* q, const OuterParams* o, const int* kind, int n,
                           hipStream_t s);
int nlam_k16_lin_fwd_multi(const LinParams* p, int n, hipStream_t s);
int nlam_k16_mlp_fwd_multi(const MlpParams* p, int n, hipStream_t s);
int nlam_k16_mlp_bwd_multi(const MlpBwdParams* q, int n, hipStream_t s);
extern "C" int64_t nlam_mlp_bwd_slab_stride(int k_in, int hid, int n_out);
int nlam_k16_outer_bwd(const OuterParams& q, hipStream_t s);
int nlam_k16_mlp_fwd(const MlpParams& p, hipStream_t s);
int nlam_k16_lin_fwd(const LinParams& p, hipStream_t s);
int nlam_edge_bwd2(const EdgeBwdParams& q, int has_egemm, hipStream_t s);   // fused_edge2.hip
// bit mask of the kernel families that take the 16-row form (NLAM_K16 in the environment,
// default all; nlam_set_k16 changes it at run time for A/B timing in one process)
// (bits 8, 64 and 128 selected the 16-row forms of the edge kernels: measured slower than the
// 32-row edge forward and than the round-4 edge backward, and removed in round 4)
enum { K16_MLP_BWD = 1, K16_LIN_BWD = 2, K16_OUTER_BWD = 4, K16_MLP_FWD = 16,
       K16_LIN_FWD = 32, K16_NODE_CHAIN = 256,
       // round 4: the pipelined, mask-free 32-row edge backward (fused_edge2.hip) for layers
       // without / with an edge update; it takes precedence over the 16-row forms above
       K16_EDGE_BWD2 = 512, K16_EDGE_BWD2_UPD = 1024 };
// default: the families whose 16-row form is the faster one on MI355X (profiles/r03_*)
#define K16_DEFAULT (K16_MLP_BWD | K16_LIN_BWD | K16_OUTER_BWD | K16_MLP_FWD | \
                     K16_LIN_FWD | K16_NODE_CHAIN | K16_EDGE_BWD2 | K16_EDGE_BWD2_UPD)
bool nlam_k16_on(int family);
