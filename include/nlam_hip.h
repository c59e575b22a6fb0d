/*
 * nlam_hip.h -- C ABI of libnlam_hip.so, the MI355X (gfx950) message-passing
 * core that sits behind neural_lam.interaction_net.InteractionNet and the
 * GraphLAM / Hi-LAM encode-process-decode stack.
 *
 * Conventions
 *   - every device pointer is a raw fp32 (float*) or int32 (int32_t*) HBM
 *     address owned by the caller (the Python host allocates through torch);
 *     nothing is allocated, freed or retained by the library;
 *   - all matrices are row-major; `ld*` is the row pitch in elements;
 *     `*_bstride` is the pitch between batch items in elements and may be 0
 *     (the reference's stride-0 `expand_to_batch` views, ar_model.py:204-209);
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream);
 *   - return value 0 = success, otherwise an error code; the message is
 *     available from nlam_last_error() (thread-local);
 *   - thread-compatible: concurrent calls must use different streams/buffers.
 *
 * Each entry point names the reference interface (file:line under
 * neural_lam/) it replaces.  The reference has no FFI of its own (it is pure
 * Python over torch / torch_geometric); INTEGRATION.md shows the ctypes
 * binding a maintainer adds.
 */
#ifndef NLAM_HIP_H
#define NLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* nlam_last_error(void);
/* ABI version of this header; bumped on any signature change (2: nlam_inet_grads.g_send_add,
 * nlam_set_k16 returns the previous mask; 3: round-5 entry points; 4: nlam_inet_graph carries the
 * sender-partial tables of nlam_edge_bwd_parts).  A binding compares it with
 * the version it was written against and refuses a stale prebuilt library (_lib.py). */
#define NLAM_ABI_VERSION 4
int nlam_abi_version(void);
/* GEMM arithmetic of the fused kernels: a property of a run, as the reference's `--precision`
 * (train_model.py:72-77,285).  Initial value: NLAM_MFMA in the environment (fp32 | bf16x3 | bf16,
 * default bf16x3); nlam_set_mfma_mode() changes it between runs of one process (every launch
 * reads it; a forward and its backward must run in the same mode).
 * 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = split-bf16 ("bf16x3": every fp32 operand
 * as bf16 hi + lo, three v_mfma_f32_32x32x16_bf16 products, fp32 accumulate; ~2^-16
 * relative error per product, tighter than the TF32 of train_model.py:246-248). */
int nlam_mfma_mode(void);   /* 0 = fp32, 1 = bf16x3 (default), 2 = bf16 (NLAM_MFMA) */
int nlam_set_mfma_mode(int mode);   /* 0 / 1 / 2 as above; 2 = the arithmetic of `--precision bf16-mixed` */

/* ---------------------------------------------------------------- graph --
 * Host-side preprocessing, run once per InteractionNet at construction.
 * Replaces the implicit edge bookkeeping of torch_geometric's
 * MessagePassing.propagate as configured by interaction_net.py:56-62.
 *
 * Input: local edge ids (senders in [0,n_send), receivers in [0,n_rec)),
 * host memory, original edge order e = 0..M-1.
 * Output (host, int32):
 *   csr_rowptr[n_rec+1], csr_eid[M]  : edges grouped by receiver, stable
 *       (csr_eid[p] = original edge id at CSR position p)
 *   csr_send[M], csr_rec[M]          : sender / receiver of the edge at p
 *   csc_colptr[n_send+1], csc_pos[M] : edges grouped by sender, stable; values
 *       are CSR positions p (so a sender-side reduction reads CSR-ordered rows)
 *   csc_eid[M]                       : the same lists as original edge ids
 *   inv_deg[n_rec] (float)           : 1 / max(in_degree, 1)  ("mean" aggr)
 */
int nlam_graph_build_host(const int64_t* send, const int64_t* rec, int64_t M,
                          int64_t n_send, int64_t n_rec, int32_t* csr_rowptr,
                          int32_t* csr_eid, int32_t* csr_send, int32_t* csr_rec,
                          int32_t* csc_colptr, int32_t* csc_pos, int32_t* csc_eid,
                          float* inv_deg);

/* Receiver-aligned edge tiles for the fused edge kernels: tile = CSR positions
 * [p0, p1), p1 - p0 <= max_edges, covering the complete in-edge segments of
 * receivers [r0, r1), r1 - r0 <= max_recs.  tiles: host int32[4 * capacity].
 * Returns the tile count, -1 if a receiver has more than max_edges in-edges,
 * -2 if capacity is too small. */
int64_t nlam_graph_tiles_host(const int32_t* csr_rowptr, int64_t n_rec,
                              int32_t max_edges, int32_t max_recs,
                              int32_t* tiles, int64_t capacity);

/* ------------------------------------------------------------ generic ops --
 * Shape-generic fp32 kernels (any hidden_dim, any number of MLP layers).
 * They cover every configuration reachable through the reference's public
 * constructors; the fused kernels further down take over for the shapes the
 * BASELINE configs use.
 */

/* C[M x N] = A[M x K] * B[K x N] (+ bias[N]) (+ C if accumulate).
 * A(i,k) = A[i*sa_i + k*sa_k], B(k,j) = B[k*sb_k + j*sb_j]  (element strides),
 * C row-major with pitch ldc.  `splitk` > 1 splits K over that many
 * workgroups per tile; partial tiles go to `workspace`
 * (splitk*M*N floats) and are summed in a fixed order (deterministic).
 * Replaces torch.nn.Linear forward / backward GEMMs inside utils.make_mlp
 * (utils.py:191-214). */
int nlam_gemm(int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_i,
              int64_t sa_k, const float* B, int64_t sb_k, int64_t sb_j,
              const float* bias, float* C, int64_t ldc, int accumulate,
              int splitk, float* workspace, void* stream);

/* y = x * sigmoid(x) over n elements;  gx = gy * d silu(x)/dx. (utils.py:207) */
int nlam_silu_fwd(const float* x, float* y, int64_t n, void* stream);
int nlam_silu_bwd(const float* x, const float* gy, float* gx, int64_t n,
                  void* stream);

/* Row LayerNorm (eps 1e-5, affine) with optional fused residual:
 *   y[r] = (res ? res[r] : 0) + gamma * (z[r]-mean)/sqrt(var+eps) + beta.
 * (utils.py:210-212; residual = interaction_net.py:109,112) */
int nlam_layernorm_fwd(const float* z, int64_t ldz, const float* gamma,
                       const float* beta, const float* res, int64_t ldres,
                       float* y, int64_t ldy, int64_t rows, int64_t d,
                       void* stream);
/* gz from gy; per-block partial dgamma/dbeta are written to `partial`
 * (2 * nblocks * d floats, nblocks = nlam_layernorm_bwd_blocks(rows)) and
 * reduced in order into dgamma/dbeta (accumulate=1 adds to them). */
int64_t nlam_layernorm_bwd_blocks(int64_t rows);
int nlam_layernorm_bwd(const float* z, int64_t ldz, const float* gamma,
                       const float* gy, int64_t ldgy, float* gz, int64_t ldgz,
                       float* dgamma, float* dbeta, int accumulate,
                       float* partial, int64_t rows, int64_t d, void* stream);

/* out[r] += / = sum over rows of x (column sums; bias gradients). `partial`:
 * nlam_colsum_blocks(rows) * d floats. */
int64_t nlam_colsum_blocks(int64_t rows);
int nlam_colsum(const float* x, int64_t ldx, float* out, int accumulate,
                float* partial, int64_t rows, int64_t d, void* stream);

/* out[b][k][0:d] = s * x[b][idx[k]][0:d]  (row gather along dim -2), with
 * s = row_scale ? row_scale[idx[k]] : 1 (the 1/deg factor of "mean" backward).
 * Replaces MessagePassing.__lift__ / index_select (interaction_net.py:103). */
int nlam_gather_rows(const float* x, int64_t x_bstride, int64_t ldx,
                     const int32_t* idx, const float* row_scale, float* out,
                     int64_t out_bstride, int64_t ldout, int64_t B,
                     int64_t n_out, int64_t d, void* stream);

/* out[b][i][0:d] = scale_i * sum_{p in [rowptr[i], rowptr[i+1])}
 *                       src[b][pos ? pos[p] : p][0:d]
 * (+ out if accumulate).  One wavefront walks one segment: no atomics,
 * fixed summation order.  scale_i = scale ? scale[i] : 1.
 * Replaces MessagePassing.aggregate / scatter_add_ (interaction_net.py:124-131)
 * and the index_add_ backward of the gathers. */
int nlam_segment_sum(const float* src, int64_t src_bstride, int64_t ldsrc,
                     const int32_t* rowptr, const int32_t* pos,
                     const float* scale, float* out, int64_t out_bstride,
                     int64_t ldout, int accumulate, int64_t B, int64_t n_out,
                     int64_t d, void* stream);

/* nlam_segment_sum (scale = 1, no accumulate) and, from the same pass over src, the batch sum of
 * every listed row:  bsum[pos[p]][0:d] = sum_b src[b][pos[p]][0:d]  for every p of every segment
 * (rows listed in no segment are not written; a row listed twice gets the same value twice).  The
 * backward of an InteractionNet whose first-layer edge term is batch-invariant needs both the
 * per-sender sums of the edge gradients and their sum over the batch: every edge is in exactly
 * one sender's list, so one walk over the lists serves both (reference interaction_net.py:121
 * under expand_to_batch, base_graph_model.py:139,152).  d in {64, 128, 256}, B * d <= 1024,
 * 16-byte aligned rows.  Summation order: list order per sample, sample order per row. */
int nlam_segment_sum_bsum(const float* src, int64_t src_bstride, int64_t ldsrc,
                          const int32_t* rowptr, const int32_t* pos, float* out,
                          int64_t out_bstride, int64_t ldout, float* bsum, int64_t ldbsum,
                          int64_t B, int64_t n_out, int64_t d, void* stream);

/* out[r][0:d] = a[r][0:d] + b[r][0:d]  (strided 2-D add; cat-slice grads). */
int nlam_add_rows(const float* a, int64_t lda, const float* b, int64_t ldb,
                  float* out, int64_t ldout, int64_t rows, int64_t d,
                  void* stream);
/* out[r][0:d] = x[r][0:d]  (strided copy into a column slice of a wider row).*/
int nlam_copy_rows(const float* x, int64_t x_bstride, int64_t ldx, float* out,
                   int64_t out_bstride, int64_t ldout, int64_t B, int64_t rows,
                   int64_t d, void* stream);
/* out[0:n] = sum_b x[b*bstride + 0:n]  (gradient of a stride-0 expand). */
int nlam_sum_batch(const float* x, int64_t bstride, float* out, int64_t B,
                   int64_t n, void* stream);

/* ------------------------------------------------------------- fused ops --
 * gfx950 kernels for hidden_layers == 1 and hidden width in {64, 128}: one
 * wavefront owns 32 rows, features live in MFMA accumulator layout, weights in
 * LDS, workgroups persistent.  Forward saves nothing (backward recomputes).
 */

/* y = [res +] [LayerNorm](W2 silu(W1 [xa | xb] + b1) + b2)
 * = utils.make_mlp([k_in, hid, n_out]) applied to rows (utils.py:191-214);
 * the two-source form is the node update aggr_mlp([x_r, agg]) with residual
 * (interaction_net.py:106-109).  xb may be NULL.  gamma/beta NULL = no LN.
 * W1: (hid x k_in) pitch ldW1, W2: (n_out x hid) pitch ldW2. */
int nlam_mlp_fwd(const float* xa, int64_t xa_bstride, int64_t xa_ld, int xa_width,
                 const float* xb, int64_t xb_bstride, int64_t xb_ld, int xb_width,
                 const float* W1, int64_t ldW1, const float* b1,
                 const float* W2, int64_t ldW2, const float* b2,
                 const float* gamma, const float* beta,
                 const float* res, int64_t res_bstride, int64_t res_ld,
                 float* out, int64_t out_bstride, int64_t out_ld,
                 int64_t B, int64_t rows, int hid, int n_out, void* stream);

/* Backward of nlam_mlp_fwd with recomputation (nothing is saved in forward).
 * gy: gradient of the output (B, rows, n_out).  gxa / gxb (optional) receive
 * the gradients of the two sources; add_gy_to_gxa adds gy to gxa (the residual
 * was source a).  Parameter gradients are written as per-workgroup partial
 * slabs  [dW1 (hid x KP32) | db1 | dW2 (NO32 x hid) | db2 | dgamma | dbeta]
 * (KP32 / NO32 = k_in / n_out rounded up to 32) with pitch slab_stride >=
 * nlam_mlp_bwd_slab_stride(); the number of slabs is nlam_bwd_grid(B *
 * ceil(rows/32)); nlam_reduce_slabs sums them in a fixed order. */
int64_t nlam_bwd_grid(int64_t ntiles);
int64_t nlam_mlp_bwd_slab_stride(int k_in, int hid, int n_out);
int nlam_mlp_bwd(const float* xa, int64_t xa_bstride, int64_t xa_ld, int xa_width,
                 const float* xb, int64_t xb_bstride, int64_t xb_ld, int xb_width,
                 const float* W1, int64_t ldW1, const float* b1,
                 const float* W2, int64_t ldW2, const float* b2, const float* gamma,
                 const float* gy, int64_t gy_bstride, int64_t gy_ld,
                 float* gxa, int64_t gxa_bstride, int64_t gxa_ld,
                 float* gxb, int64_t gxb_bstride, int64_t gxb_ld, int add_gy_to_gxa,
                 float* slab, int64_t slab_stride, float* ga_out,
                 int64_t B, int64_t rows, int hid, int n_out, void* stream);
/* Deferred weight gradient of a first layer: with ga_out != NULL nlam_mlp_bwd
 * stores ga = dL/d(pre-activation) (B, rows, hid; contiguous) instead of forming
 * dW1 / db1 (their slab ranges are left untouched); this pass computes
 *   dW (ng x KX32) = sum_rows g[r]^T (x) [xa | xb][x_index ? x_index[r] : r],
 *   db (ng) = column sums of g
 * as per-workgroup slabs [dW | db] (count nlam_bwd_grid(B*ceil(rows/32)), pitch >=
 * nlam_outer_bwd_slab_stride).  Also usable for W1e's gradient from (gh, e). */
int64_t nlam_outer_bwd_slab_stride(int ng, int kx);
int nlam_outer_bwd(const float* g, int64_t g_bstride, int64_t g_ld, int ng,
                   const float* xa, int64_t xa_bstride, int64_t xa_ld, int xa_width,
                   const float* xb, int64_t xb_bstride, int64_t xb_ld, int xb_width,
                   const int32_t* x_index, float* slab, int64_t slab_stride,
                   int64_t B, int64_t rows, void* stream);
/* ---- node side of a CHAIN of InteractionNets on shared nodes (hidden width 64) ----------
 * Replaces, between two edge passes of consecutive layers of the reference's processor
 * (models/graph_lam.py:51-57,88: pyg Sequential of m2m InteractionNets on the same mesh
 * nodes), the per-node work of interaction_net.py:112-115 (aggregation MLP + residual of
 * layer l-1) followed by the sender / receiver halves of interaction_net.py:121 (first
 * edge-MLP Linear of layer l), and their autograd.  All row views: 16-byte aligned, 64
 * columns (P / gP: 128).  1 = the process's arithmetic mode has these kernels (split-bf16). */
int nlam_node_chain_supported(void);
/* xout = x + LN(V2 silu(V1 [x | agg] + c1) + c2); P (may be NULL) = [xout WA^T + bA |
 * xout WB^T + bB] with WA / WB (64 x 64) the NEXT layer's W1[:, d:2d] / W1[:, 2d:3d]. */
int nlam_node_fwd(const float* x, int64_t x_bstride, int64_t x_ld,
                  const float* agg, int64_t agg_bstride, int64_t agg_ld,
                  const float* V1, int64_t ldV1, const float* c1,
                  const float* V2, int64_t ldV2, const float* c2,
                  const float* gamma, const float* beta,
                  float* xout, int64_t xo_bstride, int64_t xo_ld,
                  const float* WA, int64_t ldWA, const float* bA,
                  const float* WB, int64_t ldWB, const float* bB,
                  float* P, int64_t p_bstride, int64_t p_ld,
                  int64_t B, int64_t rows, void* stream);
/* Backward data pass.  Layer l: gP[:, :, 0:64] = sum over the node's out-edges (sender lists
 * csc_colptr / csc_eid, nodes >= n_send have none) of gh (B, M, 64; the hidden gradient
 * nlam_edge_bwd leaves in edge order); gP[:, :, 64:128] is read (nlam_edge_bwd's gpr);
 * G = gP [WA; WB] + g_res.  With x == NULL: gx_out = G.  Otherwise the aggregation MLP of
 * layer l-1 (x, agg, V1 .. gamma) is differentiated with output gradient G:
 * gx_out = dL/dx (residual included), gagg_out = dL/dagg, ga_out (B, rows, 64 contiguous) =
 * hidden gradient for nlam_node_outer, and per-workgroup slabs [dV2 64x64 | dc2 | dgamma |
 * dbeta] (count nlam_node_bwd_grid(B, rows), pitch >= nlam_node_bwd_slab_stride()). */
int64_t nlam_node_bwd_slab_stride(void);
int64_t nlam_node_bwd_grid(int64_t B, int64_t rows);
int nlam_node_bwd(const float* gh, int64_t gh_bstride,
                  const int32_t* csc_colptr, const int32_t* csc_eid, int64_t n_send,
                  float* gP, int64_t gp_bstride, int64_t gp_ld,
                  const float* g_res, int64_t gr_bstride, int64_t gr_ld,
                  const float* WA, int64_t ldWA, const float* WB, int64_t ldWB,
                  const float* x, int64_t x_bstride, int64_t x_ld,
                  const float* agg, int64_t agg_bstride, int64_t agg_ld,
                  const float* V1, int64_t ldV1, const float* c1,
                  const float* V2, int64_t ldV2, const float* c2, const float* gamma,
                  float* gx_out, int64_t gx_bstride, int64_t gx_ld,
                  float* gagg_out, int64_t gagg_bstride, int64_t gagg_ld,
                  float* ga_out, float* slab, int64_t slab_stride,
                  int64_t B, int64_t rows, void* stream);
/* Weight-gradient pass of the same pair: slabs [dV1 64x128 | dc1 64 | dWp 128x64 | dbp 128]
 * with dV1 = ga^T [xa | xb], dc1 = colsum ga (skipped when ga == NULL) and dWp = gP^T xl
 * (rows 0..63 = dWA, 64..127 = dWB), dbp = colsum gP.  Slab count nlam_node_outer_grid(B, rows),
 * pitch >= nlam_node_outer_slab_stride(). */
int64_t nlam_node_outer_slab_stride(void);
int64_t nlam_node_outer_grid(int64_t B, int64_t rows);
int nlam_node_outer(const float* ga,
                    const float* xa, int64_t xa_bstride, int64_t xa_ld,
                    const float* xb, int64_t xb_bstride, int64_t xb_ld,
                    const float* gP, int64_t gp_bstride, int64_t gp_ld,
                    const float* xl, int64_t xl_bstride, int64_t xl_ld,
                    float* slab, int64_t slab_stride, int64_t B, int64_t rows, void* stream);

/* ---- one call per InteractionNet (hidden width 64) ---------------------------------------
 * nlam_inet_fwd / nlam_inet_bwd run the WHOLE forward / backward of the reference's
 * InteractionNet.forward (interaction_net.py:86-131) -- hidden_layers = 1, plain MLPs, in-degree
 * <= 32 -- as the launch sequence documented in neural_lam_amd/fused.py, from one host call: a
 * Python caller otherwise pays one interpreter round trip per launch.  All row buffers the caller
 * allocates are contiguous (rows x 64 fp32, batch-major); the views describe the INPUTS, which
 * may be batch-invariant (B = 1, bstride = 0: the reference's expand_to_batch) or strided.
 * Supported (nlam_inet_supported() = 1): split-bf16 arithmetic mode with the 16-row kernels on;
 * shared nodes (rec.ptr == NULL) with per-sample node rows, or separate sender / receiver rows. */
typedef struct nlam_inet_graph {     /* device tables: nlam_graph_build_host + nlam_graph_tiles_host */
  const int32_t* tiles; int64_t ntiles;
  const int32_t* csr_rowptr; const int32_t* csr_eid; const int32_t* csr_send; const int32_t* csr_rec;
  const float* inv_deg;
  const int32_t* csc_colptr; const int32_t* csc_eid;
  int64_t n_send, n_rec, M;
  /* optional (all three or none; nlam_edge_bwd_parts below): per-tile sender partial sums */
  const int32_t* part_slot; const int32_t* pcsc_colptr; const int32_t* pcsc_rows;
} nlam_inet_graph;
typedef struct nlam_inet_view {      /* (B, rows, 64) fp32 input; B = 1: batch-invariant */
  const float* ptr; int64_t B, bstride, ld;
} nlam_inet_view;
typedef struct nlam_inet_weights {   /* edge_mlp.{0,2,3}, aggr_mlp.{0,2,3} (utils.py:191-214) */
  const float* W1; int64_t ldW1; const float* b1;      /* (64, 192): [edge | sender | receiver] */
  const float* W2; int64_t ldW2; const float* b2; const float* gam; const float* bet;
  const float* V1; int64_t ldV1; const float* c1;      /* (64, 128): [x_r | agg] */
  const float* V2; int64_t ldV2; const float* c2; const float* gam2; const float* bet2;
} nlam_inet_weights;
typedef struct nlam_inet_args {
  nlam_inet_graph g;
  nlam_inet_weights w;
  nlam_inet_view send, rec, edge;    /* rec.ptr == NULL: receivers are the senders (shared nodes) */
  int64_t n_send_rows;               /* rows of send (>= g.n_send) */
  int64_t B;                         /* batch size of the outputs */
  int d, update_edges, mean;         /* d = 64 */
  /* forward outputs and saved intermediates (caller-allocated, contiguous): */
  float* P;       /* shared: (B, N, 128) = [Ps | Pr]; else Ps (send.B, n_send_rows, 64) */
  float* Pr;      /* separate nodes: (rec.B, n_rec, 64) */
  float* Pe;      /* update_edges == 0: (edge.B, M, 64) */
  float* agg;     /* (B, n_rec, 64) */
  float* e_out;   /* update_edges: (B, M, 64) */
  float* rec_out; /* (B, n_rec, 64) */
  /* separate nodes: P (ps_given) / Pr (pr_given) already hold the sender / receiver projection
   * of edge_mlp.0 -- written by nlam_grid_encode_fwd in the same pass as the rows they project --
   * and nlam_inet_fwd skips that third of the projection launch */
  int ps_given, pr_given;
} nlam_inet_args;
typedef struct nlam_inet_grads {
  const float* g_rec_out;            /* (B, n_rec, 64) contiguous, required */
  const float* g_edge_out;           /* (B, M, 64) contiguous or NULL */
  float* g_send;                     /* (send.B, n_send_rows, 64): total gradient when nodes are shared */
  float* g_rec;                      /* (rec.B, n_rec, 64); unused when nodes are shared */
  float* g_edge;                     /* (edge.B, M, 64) */
  float* dW1; float* db1; float* dW2; float* db2; float* dgam; float* dbet;     /* contiguous */
  float* dV1; float* dc1; float* dV2; float* dc2; float* dgam2; float* dbet2;
  /* optional addend of g_send (same shape; separate sender / receiver nodes only): a gradient
   * that reached send_rep through another consumer, added inside the projection backward */
  const float* g_send_add;
} nlam_inet_grads;
/* sizeof(nlam_inet_args) / sizeof(nlam_inet_grads) as this library was built: a binding that
 * mirrors the structs (ctypes, cgo, JNI) checks its own layout against them at load time */
int64_t nlam_sizeof_inet_args(void);
int64_t nlam_sizeof_inet_grads(void);
int nlam_inet_supported(const nlam_inet_args* a);
int nlam_inet_fwd(const nlam_inet_args* a, void* stream);
/* workspace (fp32 elements) of nlam_inet_bwd for this configuration; the workspace holds the
 * intermediates and the per-workgroup weight-gradient slabs, reduced by ONE launch at the end */
int64_t nlam_inet_bwd_workspace(const nlam_inet_args* a);
int nlam_inet_bwd(const nlam_inet_args* a, const nlam_inet_grads* gr, float* ws, int64_t ws_floats,
                  void* stream);

/* out[i] (+)= sum_s slab[s * stride + i], i < n (deterministic order). */
int nlam_reduce_slabs(const float* slab, int64_t nslabs, int64_t stride, int64_t n,
                      float* out, int accumulate, void* stream);

/* Sums up to 64 matrix segments of the per-workgroup slabs into (strided)
 * destinations in ONE launch:  dst_k[r*dst_ld_k + c] = sum_s slab[s*stride +
 * src_off_k + r*src_ld_k + c], r < rows_k, c < cols_k.  Array arguments are HOST
 * arrays of length nseg (<= 64).  Fixed summation order (deterministic). */
int nlam_reduce_slabs_multi(const float* slab, int64_t nslabs, int64_t stride, int nseg,
                            const int64_t* src_off, const int32_t* rows,
                            const int32_t* cols, const int64_t* src_ld,
                            float* const* dst, const int64_t* dst_ld, void* stream);
/* Same, but every segment names its own slab buffer (slab_k, nslabs_k, stride_k): one
 * launch finishes all parameter gradients of an InteractionNet layer's backward
 * (interaction_net.py:86-131; the reductions are launch-latency bound). */
int nlam_reduce_slabs_batch(int nseg, const float* const* slab, const int64_t* nslabs,
                            const int64_t* stride, const int64_t* src_off,
                            const int32_t* rows, const int32_t* cols,
                            const int64_t* src_ld, float* const* dst,
                            const int64_t* dst_ld, void* stream);

/* out[:, 0:nA] = x WA^T + bA, out[:, nA:nA+nB] = x WB^T + bB (WB may be NULL):
 * the node-side projections Ps = W1s x_s, Pr = W1r x_r + b1 of the edge MLP's
 * first Linear (edge_mlp.0, interaction_net.py:65,121) after splitting
 * W1 [e; x_s; x_r] = W1e e + W1s x_s + W1r x_r.  nA, nB multiples of 32. */
int nlam_lin_fwd(const float* x, int64_t x_bstride, int64_t x_ld, int k_in,
                 const float* WA, int64_t ldWA, const float* bA, int nA,
                 const float* WB, int64_t ldWB, const float* bB, int nB,
                 float* out, int64_t out_bstride, int64_t out_ld,
                 int64_t B, int64_t rows, int out_bf16, void* stream);

/* Fused edge update + aggregation of one InteractionNet layer
 * (interaction_net.py:102-105,117-131):
 *   h_k = (has_egemm ? W1e e_k : e_k) + ps[send(k)] + pr[rec(k)]
 *   m_k = LN(W2 silu(h_k) + b2);  agg_i = inv_deg_i * sum_{rec(k)=i} m_k;
 *   e_out_k = e_k + m_k (has_egemm only).
 * With has_egemm = 0, `e` holds the pre-projected edge term Pe = W1e e (+0),
 * typically batch-invariant (e_bstride = 0).  Edge rows are addressed in the
 * ORIGINAL edge order through csr_eid; tiles from nlam_graph_tiles_host. */
int nlam_edge_fwd(const int32_t* tiles, int64_t ntiles,
                  const int32_t* csr_rowptr, const int32_t* csr_eid,
                  const int32_t* csr_send, const int32_t* csr_rec,
                  const float* inv_deg,
                  const float* e, int64_t e_bstride, int64_t e_ld, int has_egemm,
                  const float* ps, int64_t ps_bstride, int64_t ps_ld,
                  const float* pr, int64_t pr_bstride, int64_t pr_ld,
                  const float* W1e, int64_t ldW1e,
                  const float* W2, int64_t ldW2, const float* b2,
                  const float* gamma, const float* beta,
                  float* agg, int64_t agg_bstride, int64_t agg_ld,
                  float* e_out, int64_t eo_bstride, int64_t eo_ld,
                  int64_t B, int d, void* stream);

/* Backward of nlam_lin_fwd: gx = gy [WA; WB] [+ gx_add] (optional; gx_add folds the
 * autograd accumulation of a second gradient of the same input).  gy_nsum > 1: x is
 * batch-invariant and gy is read as sum_{s < gy_nsum} gy[s * gy_sum_stride + ...] (the
 * expand_to_batch backward of ar_model.py:204-209 folded into the load).  Per-workgroup slabs
 * [dW ((nA+nB) x KP32) | db (nA+nB)], KP32 = k_in rounded up to 32; number of
 * slabs = nlam_bwd_grid(B * ceil(rows/32)). */
int64_t nlam_lin_bwd_slab_stride(int k_in, int n_out);
int nlam_lin_bwd(const float* x, int64_t x_bstride, int64_t x_ld, int k_in,
                 const float* gy, int64_t gy_bstride, int64_t gy_ld,
                 const float* WA, int64_t ldWA, int nA,
                 const float* WB, int64_t ldWB, int nB,
                 float* gx, int64_t gx_bstride, int64_t gx_ld,
                 const float* gx_add, int64_t ga_bstride, int64_t ga_ld,
                 int64_t gy_nsum, int64_t gy_sum_stride,
                 float* slab, int64_t slab_stride, int64_t B, int64_t rows,
                 void* stream);

/* Backward of nlam_edge_fwd with recomputation.  Inputs as in forward plus
 *   g_agg (B, N_r, d): gradient of the aggregate; g_eout (B, M, d, original
 *   order, may be NULL): gradient of e_out (has_egemm).
 * Outputs: gh_out (B, M, d) = gradient of h in the original edge order (pitch d) --
 * the caller reduces it per sender (nlam_segment_sum over csc_eid) and, for a
 * batch-invariant Pe, over the batch; gpr (B, N_r, d) = per-receiver sum of gh;
 * g_e (B, M, d, original order; has_egemm) = g_eout + W1e^T gh; per-workgroup
 * slabs [dW1e (d x d) | dW2 (d x d) | db2 | dgamma | dbeta], count =
 * nlam_bwd_grid(B * ntiles).
 * Without an edge update (has_egemm = 0) a non-NULL g_e, (M, d) with pitch ge_ld, receives
 * dPe = sum_b gh_out[b]: the gradient of a batch-invariant Pe (e_bstride = 0), formed in the
 * kernel's registers when the split-bf16 round-4 kernel runs (a wave then takes whole tiles and
 * runs their batch items back to back), by one nlam_sum_batch launch otherwise.
 * nlam_edge_bwd_forms_batch_sum(): 1 if the in-kernel form would run for this tile count / batch
 * (it hands out whole tiles and is only taken when that costs at most ~6 % in rounds); callers
 * that get 0 keep the batch sum folded into the projection backward's load (gy_nsum). */
int nlam_edge_bwd_forms_batch_sum(int64_t ntiles, int64_t B, int d);
int64_t nlam_edge_bwd_slab_stride(int d);
int nlam_edge_bwd(const int32_t* tiles, int64_t ntiles,
                  const int32_t* csr_rowptr, const int32_t* csr_eid,
                  const int32_t* csr_send, const int32_t* csr_rec,
                  const float* inv_deg,
                  const float* e, int64_t e_bstride, int64_t e_ld, int has_egemm,
                  const float* ps, int64_t ps_bstride, int64_t ps_ld,
                  const float* pr, int64_t pr_bstride, int64_t pr_ld,
                  const float* W1e, int64_t ldW1e,
                  const float* W2, int64_t ldW2, const float* b2, const float* gamma,
                  const float* g_agg, int64_t gagg_bstride, int64_t gagg_ld,
                  const float* g_eout, int64_t geo_bstride, int64_t geo_ld,
                  float* gh_out, int64_t gh_bstride,
                  float* gpr, int64_t gpr_bstride, int64_t gpr_ld,
                  float* g_e, int64_t ge_bstride, int64_t ge_ld,
                  float* slab, int64_t slab_stride, int64_t B, int d, void* stream);

/* nlam_edge_bwd without an edge update and with a batch-invariant first-layer edge term pe (M, d)
 * (the grid-side nets: reference base_graph_model.py:139,152 with expand_to_batch), B > 1, when the
 * gh rows are wanted ONLY for the sender-side sums of the projection backward: the kernel writes,
 * instead of gh (B, M, d), the sums of each tile's gh rows per distinct sender.
 *   part_slot (M, CSR position order): bits 0-7 = rank of the position's sender among the distinct
 *     senders of its tile (every tile must have at most 16), bits 8-15 = that number of distinct
 *     senders (the same on every position of a tile);
 *   gpart (B, 16 ntiles, d): row 16 t + s = sum of the gh rows of tile t whose part_slot is s (rows
 *     of unused slots are not written);
 *   dpe (M, d) = sum_b gh[b] as nlam_edge_bwd's g_e in this form; gpr as there.
 * The sender sums are then  gPs[b][n] = sum of gpart[b][r] over the rows r listed for sender n
 * (pcsc_colptr / pcsc_rows, the CSC of the (tile, sender) pairs; nlam_lin_bwd_multi's gather takes
 * them in place of csc_colptr / csc_eid): m2g of the MEPS graph has 76 k pairs for 255 k edges.
 * Arithmetic: the indicator product that forms the receiver sums (fp32 accumulation of the split
 * hi + lo bf16 planes of gh).  nlam_edge_bwd_parts_supported(): hidden 64, split-bf16 mode, and
 * nlam_edge_bwd_forms_batch_sum(ntiles, B, d) = 1. */
int nlam_edge_bwd_parts_supported(int64_t ntiles, int64_t B, int d);
int nlam_edge_bwd_parts(const int32_t* tiles, int64_t ntiles,
                        const int32_t* csr_rowptr, const int32_t* csr_eid,
                        const int32_t* csr_send, const int32_t* csr_rec, const float* inv_deg,
                        const float* pe, int64_t pe_ld,
                        const float* ps, int64_t ps_bstride, int64_t ps_ld,
                        const float* pr, int64_t pr_bstride, int64_t pr_ld,
                        const float* W2, int64_t ldW2, const float* b2, const float* gamma,
                        const float* g_agg, int64_t gagg_bstride, int64_t gagg_ld,
                        const int32_t* part_slot, float* gpart, int64_t gpart_bstride,
                        float* gpr, int64_t gpr_bstride, int64_t gpr_ld, float* dpe, int64_t dpe_ld,
                        float* slab, int64_t slab_stride, int64_t B, int d, void* stream);

/* One AdamW step over a flat fp32 parameter buffer (decoupled weight decay,
 * bias correction; torch.optim.AdamW semantics, ar_model.py:191-195).
 * g is multiplied by grad_scale first (1/world after a SUM all-reduce).
 * `step` is the 1-based step count. */
int nlam_adamw_step(float* p, const float* g, float* m, float* v, int64_t n,
                    float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, float grad_scale,
                    void* stream);

/* out = src[0] + ... + src[n - 1] over numel fp32 elements (n <= 8, fixed order, 16-byte aligned
 * contiguous operands): the per-chunk aggregates of a SplitMLPs InteractionNet
 * (interaction_net.py:134-163; hi_lam_parallel.py:26-53) in one pass. */
int nlam_sum_many(int n, const float* const* src, float* out, int64_t numel, void* stream);

/* Gradient packing for the flat-buffer all-reduce (the reference leaves this to DDP's bucket
 * copies, train_model.py:265-274 `strategy="ddp"`): n gradient tensors -> their slices of one
 * fp32 buffer in ONE launch.  table (device memory, int64): n triples [source address | element
 * offset in dst | numel], then first[n + 1] = prefix sums of ceil(numel / nlam_pack_chunk());
 * nchunks = first[n].  A null source address writes zeros.  Slices must not overlap. */
int64_t nlam_pack_chunk(void);
int nlam_pack_segments(const int64_t* table, int n, int64_t nchunks, float* dst, void* stream);

/* ------------------------------------------------------------ rollout glue --
 * y = a + x * scale[f] + shift[f] over (rows, F): the state residual
 * prev_state + net_out * diff_std + diff_mean (base_graph_model.py:174-177). */
int nlam_affine_residual(const float* a, const float* x, const float* scale,
                         const float* shift, float* y, int64_t rows, int F,
                         void* stream);

/* One rollout step's state update in one pass -- the residual above followed by the boundary
 * overwrite (ar_model.py:244-247: boundary_mask * true_state + interior_mask * pred_state):
 *   out[b][n][f] = mask[n] ? truth[b][n][f] : prev[b][n][f] + net_out[b][n][f] * scale[f] + shift[f]
 * prev / truth are (B, N, F) views with their own batch pitch (slices of the batch tensors),
 * net_out / out are contiguous; mask (N) in {0, 1}.  Backward: gx = (1 - mask) g scale[f],
 * gprev = (1 - mask) g (optional, null to skip). */
int nlam_state_step(const float* prev, int64_t prev_bstride, const float* net_out,
                    const float* truth, int64_t truth_bstride, const float* mask,
                    const float* scale, const float* shift, float* out, int64_t B, int64_t N,
                    int F, void* stream);
int nlam_state_step_bwd(const float* g, const float* mask, const float* scale, float* gx,
                        float* gprev, int64_t B, int64_t N, int F, void* stream);
/* gx = g * scale[f] (its backward w.r.t. x). */
int nlam_scale_cols(const float* g, const float* scale, float* gx, int64_t rows,
                    int F, void* stream);
/* out[b][n][:] = mask[n] * truth[b][n][:] + (1 - mask[n]) * pred[b][n][:], the
 * boundary overwrite of ar_model.py:244-247; truth == NULL gives the backward
 * g_pred = (1 - mask) * g_out. */
int nlam_boundary_mix(const float* pred, const float* truth, const float* mask,
                      float* out, int64_t B, int64_t N, int F, void* stream);
/* out[0] = scale * sum_{r,f} keep[r % N] * w[f] * (pred - target)^2 over
 * (rows, F): torch.mean over (B,T) of the masked, grid-averaged, variable-summed
 * wmse/mse (metrics.py:21-108, ar_model.py:294-298) with keep = interior mask,
 * w = 1/std^2, scale = 1/(n_interior * B * T).  partial: nlam_wmse_blocks()
 * floats.  Deterministic two-stage reduction. */
int64_t nlam_wmse_blocks(void);
int nlam_wmse_fwd(const float* pred, const float* target, const float* keep,
                  const float* w, float* partial, float* out, int64_t rows,
                  int64_t N, int F, float scale, void* stream);
/* g_pred = gloss[0] * 2 * scale * keep * w * (pred - target). */
int nlam_wmse_bwd(const float* pred, const float* target, const float* keep,
                  const float* w, const float* gloss, float scale, float* g_pred,
                  int64_t rows, int64_t N, int F, void* stream);

/* ---- hidden_dim 128 ("wide") path: one weight matrix per kernel (csrc/fused_wide.hip) -------
 * The same reference operators as above -- make_mlp blocks (utils.py:191-214) and the
 * InteractionNet message / aggregate / update (interaction_net.py:86-131) -- cut at the Linear
 * boundaries, because two split-bf16 128 x 128 weight images do not fit the 160 KB LDS.
 * Requires NLAM_MFMA=bf16x3 (default) or bf16.  Index arrays are int32; NULL = identity.
 *
 * nlam_tail_fwd: for every position p (< rows; tile k covers positions [32 k, 32 k + 32) in
 * row mode, or tiles[k] = (p0, p1, r0, r1) in edge mode -- the receiver-aligned tiles of
 * nlam_graph_tiles_host over receiver-sorted positions):
 *   h[p]  = a[idx_a[p]] + b[idx_b[p]] + c[idx_c[p]]            (b, c optional; width d)
 *   m[p]  = [LayerNorm](W2 silu(h[p]) + b2)                     (W2: n_out x d)
 *   h_out[p] = h[p]                       (optional; (B, rows, d) contiguous rows, kept for backward)
 *   y[idx_y[p]] = m[p] (+ res[idx_y[p]])  (optional)
 *   agg[i] = inv_deg[i] * sum_{p: csr_rec[p] = i} m[p]          (optional, edge mode)
 * i.e. edge_mlp / aggr_mlp / embedder blocks after their first Linear (nlam_lin_fwd).
 * d = 256 (csrc/fused_fs.hip), NLAM_MFMA=bf16: W2 silu(h) + b2 is rounded to bf16 before the
 * LayerNorm -- the Linear output dtype under the reference's `--precision bf16-mixed` autocast
 * (train_model.py:73-76) -- and z_keep (optional; (B, rows, d) bf16 rows, position order, batch
 * pitch z_bstride elements) receives those rows for nlam_tail_bwd, which then needs no second
 * GEMM.  d = 256 in the default split-bf16 mode: no rounding, z_keep holds fp32 rows (16-byte
 * aligned, pitch in fp32 elements).  d = 128 ignores z_keep (its backward repeats the GEMM from h).
 *
 * bf16 STORAGE of intermediates (hidden 256 only; the dtype the reference's autocast gives the
 * outputs of nn.Linear and their gradients): `out_bf16` / `io_bf16` say which row operands are
 * bf16 instead of fp32 -- the pointer is then a bf16 pointer passed as float*, its pitches count
 * bf16 elements (multiples of 8: a lane moves 16 bytes) and the rows are 256 wide.
 *   nlam_lin_fwd: out_bf16 (nlam_lin_fwd_multi: bit k = problem k);
 *   nlam_tail_fwd: io_bf16 != 0 = a, b, c AND h_out are bf16 (h is rounded to bf16 before the SiLU);
 *   nlam_tail_bwd: io_bf16 != 0 = h AND (LayerNorm form) gz_out are bf16; without LayerNorm (the
 *     narrow heads, n_out <= 32) gz_out stays fp32, 32 wide;
 *   nlam_wide_outer[_multi]: bit 0 = g, bit 1 = x.
 * 0 everywhere = all fp32 (the only form at hidden 64 / 128). */
int nlam_tail_fwd(const int32_t* tiles, int64_t ntiles, int64_t rows, const int32_t* csr_rec,
                  const int32_t* csr_rowptr,
                  const float* a, int64_t a_bstride, int64_t a_ld, const int32_t* idx_a,
                  const float* b, int64_t b_bstride, int64_t b_ld, const int32_t* idx_b,
                  const float* c, int64_t c_bstride, int64_t c_ld, const int32_t* idx_c,
                  const float* W2, int64_t ldW2, const float* b2, const float* gamma,
                  const float* beta, int n_out, float* h_out, int64_t h_bstride,
                  void* z_keep, int64_t z_bstride,
                  float* y, int64_t y_bstride, int64_t y_ld, const int32_t* idx_y,
                  const float* res, int64_t res_bstride, int64_t res_ld,
                  float* agg, int64_t agg_bstride, int64_t agg_ld, const float* inv_deg,
                  int64_t B, int d, int io_bf16, void* stream);
/* The node update with its aggregate projection inside the tail (reference interaction_net.py:
 * 127-131: aggr_mlp(cat(rec_rep, edge_rep_aggr)) + residual):  h = a + pre . preW^T  (preW (d, d),
 * e.g. the aggregate columns V1[:, d:] of the first Linear; `a` holds the receiver part and the
 * bias), y = res + LN(W2 silu(h) + b2), h kept in h_out.  The same products as nlam_lin_fwd(pre,
 * preW) followed by nlam_tail_fwd(a, b = that product), accumulated ONTO a instead of summed
 * beside it (fp32 rounding-order differences only); one launch and no round trip of the product.
 * Hidden 128, contiguous rows, and
 * B * ceil(rows / 32) <= 1024 (one row tile per wave: the workgroup holds preW's split-bf16 image
 * first and W2's after it); nlam_tail_fwd_pre_supported says whether a shape qualifies. */
int nlam_tail_fwd_pre_supported(int d, int64_t B, int64_t rows);
int nlam_tail_fwd_pre(int64_t rows, const float* a, int64_t a_bstride, int64_t a_ld,
                      const float* pre, int64_t pre_bstride, int64_t pre_ld, const float* preW,
                      int64_t ldpreW, const float* W2, int64_t ldW2, const float* b2,
                      const float* gamma, const float* beta, float* h_out, int64_t h_bstride,
                      float* y, int64_t y_bstride, int64_t y_ld, const float* res,
                      int64_t res_bstride, int64_t res_ld, int64_t B, int d, void* stream);
/* Several independent MLP tails in one launch (n <= 8): y_k = LN(W2_k silu(h_k) + b2_k) for
 * contiguous h_k, y_k (B_k, rows_k, 128) -- the static-feature embedders of a model (reference
 * base_graph_model.py:127-130, base_hi_graph_model.py:137-166: ten small, mutually independent MLPs
 * at the start of every Hi-LAM step); backward: gz_k = LN'(z_k; gy_k) and gh_k = (W2_k^T gz_k) *
 * silu'(h_k) as nlam_tail_bwd, dgamma / dbeta partials in slab_k (nslabs[k] =
 * nlam_mlp_tail_multi_shares slabs of nlam_tail_bwd_slab_stride(128) floats).  The kernels are the
 * bodies of nlam_tail_fwd / nlam_tail_bwd compiled a second time around a per-problem block
 * range (csrc/tail_fwd_body.h, tail_bwd_body.h): the same arithmetic in the same order. */
int nlam_mlp_tail_multi_shares(int n, const int64_t* B, const int64_t* rows, int32_t* shares);
int nlam_mlp_tail_fwd_multi(int n, int d, const float* const* h, const float* const* W2,
                            const int64_t* ldW2, const float* const* b2, const float* const* gamma,
                            const float* const* beta, float* const* y, const int64_t* B,
                            const int64_t* rows, void* stream);
int nlam_mlp_tail_bwd_multi(int n, int d, const float* const* h, const float* const* gy,
                            const float* const* W2, const int64_t* ldW2, const float* const* b2,
                            const float* const* gamma, float* const* gz_out, float* const* gh,
                            float* const* slab, const int32_t* nslabs, const int64_t* B,
                            const int64_t* rows, void* stream);
/* Backward of nlam_tail_fwd from the kept h:  g[p] = scale1[idx_g1[p]] * g1[idx_g1[p]] +
 * g2[idx_g2[p]] is the gradient of m[p];  gz = LN'(z; g) (= g without LayerNorm) is written to
 * gz_out (B, rows, ceil32(n_out)) for the weight-gradient pass;  gh[idx_gh[p]] =
 * (W2^T gz[p]) * silu'(h[p]);  gpr[i] = sum_{p: csr_rec[p] = i} gh[p] (optional, edge mode);
 * dgamma / dbeta as per-workgroup slabs [dgamma | dbeta] (count nlam_bwd_grid(B * ntiles),
 * pitch >= nlam_tail_bwd_slab_stride).  dW2 = gz^T silu(h), db2 = colsum(gz): nlam_wide_outer.
 * d = 256 with LayerNorm: z_keep = the rows nlam_tail_fwd kept (required). */
int64_t nlam_tail_bwd_slab_stride(int n_out);
int nlam_tail_bwd(const int32_t* tiles, int64_t ntiles, int64_t rows, const int32_t* csr_rec,
                  const int32_t* csr_rowptr, const float* h, int64_t h_bstride,
                  const void* z_keep, int64_t z_bstride,
                  const float* g1, int64_t g1_bstride, int64_t g1_ld, const int32_t* idx_g1,
                  const float* scale1,
                  const float* g2, int64_t g2_bstride, int64_t g2_ld, const int32_t* idx_g2,
                  const float* W2, int64_t ldW2, const float* b2, const float* gamma, int n_out,
                  float* gz_out, int64_t gz_bstride,
                  float* gh, int64_t gh_bstride, int64_t gh_ld, const int32_t* idx_gh,
                  float* gpr, int64_t gpr_bstride, int64_t gpr_ld,
                  float* slab, int64_t slab_stride, int64_t B, int d, int io_bf16, void* stream);
/* gx = gy W [+ gx_add]: data gradient of a Linear (W: n_out x k_in = 128 x 128). */
int nlam_lin_bwd_data(const float* gy, int64_t gy_bstride, int64_t gy_ld, int n_out,
                      const float* W, int64_t ldW, int k_in,
                      float* gx, int64_t gx_bstride, int64_t gx_ld,
                      const float* gx_add, int64_t ga_bstride, int64_t ga_ld,
                      int64_t B, int64_t rows, void* stream);
/* Weight gradient of a Linear as a streaming pass:  dW (ng x nx) = sum_rows g[r]^T (x) f(x[r]),
 * db (ng) = colsum(g), f = silu if silu_x (x is then the kept pre-activation h) else identity;
 * per-workgroup slabs [dW | db] (count nlam_bwd_grid(B * ceil(rows / 32)), pitch >= ng nx + ng).
 * (ng, nx) in {(128, 128), (32, 128), (128, <= 64)}; pitch >= ng * ceil32(nx) + ng; narrow or
 * unaligned x rows (static features of the embedders) are staged by scalar loads. */
int nlam_wide_outer(const float* g, int64_t g_bstride, int64_t g_ld, int ng,
                    const float* x, int64_t x_bstride, int64_t x_ld, int nx, int silu_x,
                    float* slab, int64_t slab_stride, int64_t B, int64_t rows, int io_bf16,
                    void* stream);

/* Several INDEPENDENT problems of one kind in one launch (n <= 8, nlam_wide_outer_multi: n <= 24;
 * arrays of n entries): the small
 * mesh levels of Hi-LAM (reference hi_lam.py:82-207: 10 InteractionNets per processor layer on
 * 81 ... 6,561-node levels) are bound by the latency of their launches, not by their work.
 * nlam_lin_fwd_multi: out_k = x_k W_k^T + bias_k (bias_k may be NULL), all d -> d; d = 64 (n <= 4,
 * needs nlam_lin_multi_supported()), 128, or 256.
 * nlam_lin_bwd_data_multi / nlam_wide_outer_multi: as the single forms, all d x d.
 * d = 128, or 256 (NLAM_MFMA=bf16 only).  nlam_wide_outer_multi: nslabs[k] = slabs (= workgroups)
 * of problem k, chosen by the caller -- proportional to the problems' row counts with ~512 in
 * all, so that the slab traffic of a layer's seven weight gradients stays a fraction of what
 * 7 x nlam_bwd_grid() slabs would be; slab[k] holds nslabs[k] * slab_stride[k] floats. */
int nlam_lin_fwd_multi(int n, int d, const float* const* x, const int64_t* x_bstride, const int64_t* x_ld,
                       const float* const* W, const int64_t* ldW, const float* const* bias,
                       float* const* out, const int64_t* out_bstride, const int64_t* out_ld,
                       const int64_t* B, const int64_t* rows, int out_bf16_mask, void* stream);
/* Several independent narrow-input MLP blocks in ONE launch (n <= 8): the static-feature embedders
 * of a model -- mesh nodes and every edge set, `utils.make_mlp([k, 64, 64])` with LayerNorm
 * (utils.py:191-214; base_graph_model.py:55-60, graph_lam.py:37-42, base_hi_graph_model.py:51-74)
 * applied to 2-3 wide (<= 32) feature rows.  Arguments as nlam_mlp_fwd / nlam_mlp_bwd, one array
 * entry per problem; one source, no residual; per-problem slabs as nlam_mlp_bwd
 * (nlam_bwd_grid(B_k * ceil(rows_k / 32)) slabs of pitch >= nlam_mlp_bwd_slab_stride(k, 64, 64)). */
int nlam_mlp_multi_supported(void);
int nlam_mlp_fwd_multi(int n, const float* const* x, const int64_t* x_bstride, const int64_t* x_ld,
                       const int32_t* x_width, const float* const* W1, const int64_t* ldW1,
                       const float* const* b1, const float* const* W2, const int64_t* ldW2,
                       const float* const* b2, const float* const* gamma, const float* const* beta,
                       float* const* out, const int64_t* out_bstride, const int64_t* out_ld,
                       const int64_t* B, const int64_t* rows, int hid, int n_out, void* stream);
int nlam_mlp_bwd_multi(int n, const float* const* x, const int64_t* x_bstride, const int64_t* x_ld,
                       const int32_t* x_width, const float* const* W1, const int64_t* ldW1,
                       const float* const* b1, const float* const* W2, const int64_t* ldW2,
                       const float* const* b2, const float* const* gamma,
                       const float* const* gy, const int64_t* gy_bstride, const int64_t* gy_ld,
                       float* const* gx, const int64_t* gx_bstride, const int64_t* gx_ld,
                       float* const* slab, const int64_t* slab_stride,
                       const int64_t* B, const int64_t* rows, int hid, int n_out, void* stream);

/* Hidden width 64: the sender / receiver / edge thirds of an InteractionNet's first edge-MLP Linear
 * (interaction_net.py:121) differentiated in ONE launch (n <= 4 independent problems, d = 64):
 *   gx_k = gy_k W_k (+ gx_add_k)   (gx_k may be NULL), per-workgroup slabs [dW_k 64x64 | db_k 64]
 *   with dW_k = gy_k^T x_k, db_k = colsum gy_k: nlam_bwd_grid(B_k * ceil(rows_k / 32)) slabs of pitch
 *   slab_stride_k >= 64*64 + 64 each.
 * gy_nsum_k > 1: x_k is batch-invariant (B_k = 1) and gy_k is summed over gy_nsum_k slices
 * gy_sum_stride_k apart while it is read.  gh_k != NULL: gy_k is not read but FORMED as the sum of
 * the rows gh_k[eid] (B, M, 64; batch / slice pitch gh_bstride_k) over the row's sender list
 * (csc_colptr_k / csc_eid_k, rows >= n_send_k have none) -- the sender-side scatter of the
 * reference's autograd for edge_index[0], as a gather.
 * xb_k != NULL: problem k is instead the DEFERRED first-layer weight gradient of a node update
 * (what nlam_mlp_bwd leaves to nlam_outer_bwd): W_k = gx_k = gh_k = NULL, gy_k = the stored hidden
 * gradient, slabs [dW_k 64x128 | db_k 64] with dW_k = gy_k^T [x_k | xb_k] (pitch >= 64*128 + 64).
 * nlam_lin_multi_supported(): 1 = available in this process's arithmetic mode. */
int nlam_lin_multi_supported(void);
int nlam_lin_bwd_multi(int n, int d, const float* const* x, const int64_t* x_bstride, const int64_t* x_ld,
                       const float* const* xb, const int64_t* xb_bstride, const int64_t* xb_ld,
                       const float* const* gy, const int64_t* gy_bstride, const int64_t* gy_ld,
                       const float* const* W, const int64_t* ldW,
                       float* const* gx, const int64_t* gx_bstride, const int64_t* gx_ld,
                       const float* const* gx_add, const int64_t* ga_bstride, const int64_t* ga_ld,
                       const int64_t* gy_nsum, const int64_t* gy_sum_stride,
                       const float* const* gh, const int64_t* gh_bstride,
                       const int32_t* const* csc_colptr, const int32_t* const* csc_eid,
                       const int64_t* n_send, float* const* slab, const int64_t* slab_stride,
                       const int64_t* B, const int64_t* rows, void* stream);
int nlam_lin_bwd_data_multi(int n, int d, const float* const* gy, const int64_t* gy_bstride,
                            const int64_t* gy_ld, const float* const* W, const int64_t* ldW,
                            float* const* gx, const int64_t* gx_bstride, const int64_t* gx_ld,
                            const float* const* gx_add, const int64_t* ga_bstride,
                            const int64_t* ga_ld, const int64_t* B, const int64_t* rows,
                            void* stream);
int nlam_wide_outer_multi(int n, int d, const float* const* g, const int64_t* g_bstride,
                          const int64_t* g_ld, const float* const* x, const int64_t* x_bstride,
                          const int64_t* x_ld, const int32_t* silu_x, float* const* slab,
                          const int64_t* slab_stride, const int64_t* B, const int64_t* rows,
                          const int32_t* nslabs, const int32_t* io_bf16,
                          void* stream);
/* The same for narrow x operands (first Linears of static-feature embedders and of the grid
 * embedder: dW (d x nx[k]) = g^T x, db = colsum g): every problem d x nx[k] with nx[k] <= 64; all
 * slabs of one launch have rows of 32 ceil(max_k nx[k] / 32) columns: [dW (d x that) | db (d)]. */
int nlam_wide_outer_multi_nx(int n, int d, const float* const* g, const int64_t* g_bstride,
                             const int64_t* g_ld, const float* const* x, const int64_t* x_bstride,
                             const int64_t* x_ld, const int32_t* nx, float* const* slab,
                             const int64_t* slab_stride, const int64_t* B, const int64_t* rows,
                             const int32_t* nslabs, void* stream);

/* Grid feature rows of predict_step (reference base_graph_model.py:116-124): out (B, N, sum w) =
 * concatenation of up to four (B | 1, N, w_k) sources along the feature axis (bstride 0 =
 * batch-invariant static features). */
int nlam_concat_rows(int nsrc, const float* const* src, const int64_t* bstride, const int64_t* ld,
                     const int32_t* width, float* out, int64_t B, int64_t N, void* stream);

/* Grid-side encoder chain of predict_step in one pass over the grid rows (hidden 64; reference
 * base_graph_model.py:116-143,157 and the grid-row thirds of interaction_net.py:121):
 *   feat = cat(src_0 .. src_{nsrc-1})                 (B, rows, k_in <= 64)   [optional]
 *   emb  = LayerNorm(W2 silu(W1 feat + b1) + b2)      grid_embedder
 *   ps   = emb Ws^T                                   sender third of g2m_gnn.edge_mlp.0
 *   rep  = emb + LayerNorm(E2 silu(E1 emb + e1) + e2) encoding_grid_mlp + residual
 *   pr   = rep Wr^T + br                              receiver third of m2g_gnn.edge_mlp.0
 * Sources as nlam_concat_rows (any alignment, bstride 0 = batch-invariant); outputs contiguous
 * (B, rows, 64), 16-byte aligned.  Bitwise equal to nlam_concat_rows -> nlam_mlp_fwd ->
 * nlam_lin_fwd -> nlam_mlp_fwd -> nlam_lin_fwd in the split-bf16 mode (the only mode it runs in:
 * nlam_grid_encode_supported()). */
int nlam_grid_encode_supported(void);
int nlam_grid_encode_fwd(int nsrc, const float* const* src, const int64_t* src_bstride,
                         const int64_t* src_ld, const int32_t* src_width, const float* W1,
                         int64_t ldW1, const float* b1, const float* W2, int64_t ldW2,
                         const float* b2, const float* gamma, const float* beta, const float* Ws,
                         int64_t ldWs, const float* E1, int64_t ldE1, const float* e1,
                         const float* E2, int64_t ldE2, const float* e2, const float* egamma,
                         const float* ebeta, const float* Wr, int64_t ldWr, const float* br,
                         float* feat, float* emb, float* ps, float* rep, float* pr, int64_t B,
                         int64_t rows, void* stream);

/* State step AND the training loss term of the same AR step in one pass (the loss target of step
 * t is the boundary truth of step t; reference ar_model.py:244-247,294-298, base_graph_model.py:
 * 174-177, metrics.py:21-84): out = mask ? truth : prev + net_out * scale + shift (as
 * nlam_state_step) and loss[0] = lscale * sum_{b,n,f} keep[n] w[f] (out - truth)^2 (as nlam_wmse_fwd;
 * partial: nlam_state_step_wmse_blocks() floats of scratch; N * F < 2^31).  Backward: g = g_state (NULL = 0) + 2 lscale
 * gloss[0] keep w (pred - truth);  gx = (1 - mask) g scale[f];  gprev (optional) = (1 - mask) g. */
int nlam_state_step_wmse_blocks(void);
int nlam_state_step_wmse_fwd(const float* prev, int64_t prev_bstride, const float* net_out,
                             const float* truth, int64_t truth_bstride, const float* mask,
                             const float* scale, const float* shift, const float* keep, const float* w,
                             float* out, float* partial, float* loss, float lscale, int64_t B,
                             int64_t N, int F, void* stream);
int nlam_state_step_wmse_bwd(const float* pred, const float* truth, int64_t truth_bstride,
                             const float* mask, const float* scale, const float* keep, const float* w,
                             const float* gloss, float lscale, const float* g_state, float* gx,
                             float* gprev, int64_t B, int64_t N, int F, void* stream);

/* output_std head (reference base_graph_model.py:161-177 with args.output_std): net_out is
 * (rows, 2F); state = prev + net_out[:, :F] * scale + shift, pred_std = softplus(net_out[:, F:])
 * (beta 1, threshold 20, as torch.nn.functional.softplus).  The backward takes either incoming
 * gradient as NULL (= zero) and writes g_out (rows, 2F). */
int nlam_std_head_fwd(const float* net_out, const float* prev, const float* scale,
                      const float* shift, float* state, float* pred_std, int64_t rows, int F,
                      void* stream);
int nlam_std_head_bwd(const float* net_out, const float* g_state, const float* g_std,
                      const float* scale, float* g_out, int64_t rows, int F, void* stream);
/* Masked Gaussian NLL training loss (reference metrics.py:166-190 reduced as
 * ar_model.py:294-298): out[0] = scale * sum keep[r % N] * (0.5 z^2 + log std + 0.5 log 2 pi),
 * z = (target - pred) / std; partial: nlam_wmse_blocks() floats of workspace; deterministic. */
int nlam_nll_fwd(const float* pred, const float* target, const float* pred_std,
                 const float* keep, float* partial, float* out, int64_t rows, int64_t N, int F,
                 float scale, void* stream);
int nlam_nll_bwd(const float* pred, const float* target, const float* pred_std,
                 const float* keep, const float* gloss, float scale, float* g_pred, float* g_std,
                 int64_t rows, int64_t N, int F, void* stream);

/* Diagnostic only: with NLAM_STAMP=1 in the environment nlam_edge_bwd (update_edges
 * form) runs an instrumented build that sums s_memtime cycles per phase of its tile
 * loop over all waves: out[0..5] = staging, recompute, LN-backward + column sums,
 * dW2 + W2^T gz, gh store + receiver reduce, dW1e + W1e^T gh + store.  reset != 0
 * zeroes the counters after reading. */
int nlam_debug_edge_bwd_stamps(unsigned long long* out, int reset);
/* Same for nlam_mlp_bwd: out[0..6] = staging, GEMM1 + silu, GEMM2 + LN backward, planes + dW2,
 * W2^T gz, X again + dW1 (or the ga store), W1^T ga + stores. */
int nlam_debug_mlp_bwd_stamps(unsigned long long* out, int reset);
/* The same for the hidden-256 tail kernels (16 values: forward phases 0..7, backward 8..15). */
int nlam_debug_fs_stamps(unsigned long long* out, int reset);
/* Host logic, no device needed: the workgroups-per-problem rule of the multi-problem launches
 * (nlam_lin_fwd_multi / nlam_lin_bwd_data_multi at hidden 128 / 256).  rounds[k] = trip rounds of
 * problem k (tiles / tiles per workgroup round); out[k] = its workgroups: proportional to the
 * rounds, sum <= cap unless n > cap, 1 <= out[k] <= rounds[k]. */
int nlam_debug_multi_shares(int n, const int64_t* rounds, int64_t cap, int64_t* out);
/* Diagnostic (NLAM_TIMELINE=1 in the environment of the process): nlam_lin_fwd records
 * s_memrealtime (100 MHz) per workgroup at start / after the weight prologue / at exit;
 * out: host array of 3 * 1024 values (first 1024 workgroups of the last launch). */
int nlam_debug_lin_fwd_timeline(unsigned long long* out);

/* Diagnostic (NLAM_TIMELINE_NODE=1): s_memrealtime (100 MHz) stamps of the last nlam_node_bwd launch
 * with a node update, 8 per workgroup (first 256): start, weights in LDS, gather done, G formed,
 * tile loop done, slab written.  out: host array of 256 * 8 values. */
int nlam_debug_node_timeline(unsigned long long* out);

/* Debug / self-test: verifies the MFMA fp32 32x32x2 operand and accumulator
 * lane maps the fused kernels rely on.  out: 32*32 floats = A(32x64) * B(64x32)
 * for the integer test pattern documented in csrc/mfma_probe.hip. */
int nlam_mfma_probe(float* out, void* stream);

/* Tuning hook: which hidden-64 kernel families run in their newer form.  Bit mask: 1 nlam_mlp_bwd,
 * 2 nlam_lin_bwd, 4 nlam_outer_bwd, 16 nlam_mlp_fwd, 32 nlam_lin_fwd (16-row, two-waves-per-SIMD
 * kernels, csrc/fused16_*.hip, instead of the 32-row ones), 256 the nlam_node_* chain kernels
 * (off: nlam_node_chain_supported() returns 0), 512 / 1024 nlam_edge_bwd without / with an edge
 * update in its round-4 pipelined form (csrc/fused_edge2.hip) instead of the round-2 kernel.
 * Bits 8, 64 and 128 (16-row edge kernels) are retired.  Default: all of the above (or NLAM_K16 in
 * the environment).  Same entry points, slab layouts and results (to rounding) either way; used
 * to time both forms in one process.  Returns the mask that was in force before the call; a
 * negative mask changes nothing (query). */
int nlam_set_k16(int mask);

#ifdef __cplusplus
}
#endif
#endif /* NLAM_HIP_H */
