// Grid-side encoder chain of predict_step in ONE pass over the grid rows (gfx950, hidden 64).
//
// The reference's predict_step (models/base_graph_model.py:116-143,157) runs, per grid node and
// sample, five row-local stages before / around the encoder GNN:
//   grid_features = cat(prev_state, prev_prev_state, forcing, static)           :116-124
//   grid_emb      = grid_embedder(grid_features)            make_mlp, LayerNorm :127
//   Ps(g2m)       = grid_emb W1s^T      sender third of g2m_gnn.edge_mlp.0      interaction_net.py:121
//   grid_rep      = grid_emb + encoding_grid_mlp(grid_emb)                      :141-143
//   Pr(m2g)       = grid_rep W1r^T + b1 receiver third of m2g_gnn.edge_mlp.0    interaction_net.py:121
// Launch by launch that was concat_rows -> mlp_fwd -> lin_fwd_multi(g2m) -> mlp_fwd ->
// lin_fwd_multi(m2g): each streams a (B x 63,784)-row tensor in and out again (~0.8 GB, 200 us of
// a 2.5 ms GraphLAM-64 step).  Here a 16-row tile stays in registers from the source rows to the
// last projection: the sources are read once (coalesced, through a per-wave LDS tile), and only
// what a later kernel needs is written -- grid_features (the embedder's backward input), grid_emb,
// Ps, grid_rep, Pr: 57 MB in, 317 MB out.
//
// Same building blocks, same arithmetic and the same order of operations as mlp_fwd16_kernel /
// lin_fwd16_kernel (fused16_mlp.hip), so every output is BITWISE what the launch-by-launch path
// produces (tests/test_gpu_grid.py).  Workgroup = 512 threads = 8 wavefronts sharing one copy of
// the six weight images (108 KB of LDS); the next tile's source rows are requested before this
// tile's GEMMs (two waves per SIMD: nothing else hides that round trip).
#include <cstdlib>

#include "fused16.h"
#include "fused_params.h"

#define G16_NW 8
#define G16_THREADS 512
#define G16_MAXSRC 4
#define G16_XLD 68      // fp32 staging tile [16][68]: 64 columns + 4 (conflict-free 16-byte access)
#define G16_SLOTS 16    // dword loads per lane that cover 16 rows x <= 64 columns

struct GridFwdParams {
  RowView src[G16_MAXSRC];    // (B | 1, rows, width_k); widths sum to k_in <= 64
  int nsrc, k_in;
  const float* W1; int64_t ldW1; const float* b1;      // grid_embedder: 64 x k_in, 64 x 64, LN
  const float* W2; int64_t ldW2; const float* b2;
  const float* gamma; const float* beta;
  const float* Ws; int64_t ldWs;                        // Ps = emb Ws^T            (64 x 64)
  const float* E1; int64_t ldE1; const float* e1;      // encoding MLP: 64 x 64, 64 x 64, LN, residual
  const float* E2; int64_t ldE2; const float* e2;
  const float* egamma; const float* ebeta;
  const float* Wr; int64_t ldWr; const float* br;      // Pr = rep Wr^T + br       (64 x 64)
  float* feat;    // (B, rows, k_in) contiguous; may be NULL
  float* emb;     // (B, rows, 64) contiguous, every one of the four
  float* ps;
  float* rep;
  float* pr;
  int64_t rows;
  int B;
};

// element e of a tile's concatenated rows (row-major [16][k_in]) -> where it comes from
struct G16Slot {
  int src;     // source index
  int row;     // tile row 0..15
  int c;       // column inside the source
  int col;     // column of the concatenated row
  bool live;
};
__device__ __forceinline__ G16Slot g16_slot(const GridFwdParams& p, int e) {
  G16Slot s;
  s.live = e < NLAM_T16 * p.k_in;
  const int ee = s.live ? e : 0;
  s.row = ee / p.k_in;
  s.col = ee - s.row * p.k_in;
  int c = s.col, k = 0;
#pragma unroll
  for (int j = 0; j + 1 < G16_MAXSRC; ++j) {
    const bool next = k == j && j + 1 < p.nsrc && c >= p.src[j].width;
    c = next ? c - p.src[j].width : c;
    k = next ? j + 1 : k;
  }
  s.src = k;
  s.c = c;
  return s;
}

template <int TERMS>
__global__ __launch_bounds__(G16_THREADS, 2) void grid_fwd16_kernel(GridFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  constexpr int D = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  char* cur = smem16;
  const B3Image W1im = w16_image(cur, D, D);  cur += w16_image_bytes(D, D);
  const B3Image W2im = w16_image(cur, D, D);  cur += w16_image_bytes(D, D);
  const B3Image Wsim = w16_image(cur, D, D);  cur += w16_image_bytes(D, D);
  const B3Image E1im = w16_image(cur, D, D);  cur += w16_image_bytes(D, D);
  const B3Image E2im = w16_image(cur, D, D);  cur += w16_image_bytes(D, D);
  const B3Image Wrim = w16_image(cur, D, D);  cur += w16_image_bytes(D, D);
  float* vec = reinterpret_cast<float*>(cur);           // 9 per-feature vectors of 64
  float* b1s = vec, *b2s = vec + D, *gs = vec + 2 * D, *bs = vec + 3 * D;
  float* e1s = vec + 4 * D, *e2s = vec + 5 * D, *egs = vec + 6 * D, *ebs = vec + 7 * D;
  float* brs = vec + 8 * D;
  cur += 9 * D * sizeof(float);
  float* XT = reinterpret_cast<float*>(cur) + wave * (NLAM_T16 * G16_XLD);

  {   // every global load of the prologue in flight together (fused_bf16x3.h)
    VLoad16 lv;
    const float* const vecs[8] = {p.b1, p.b2, p.gamma, p.beta, p.e1, p.e2, p.egamma, p.ebeta};
    const int lens[8] = {D, D, D, D, D, D, D, D};
    v16_issue(lv, vecs, lens, tid);
    const float brv = (tid < D && p.br != nullptr) ? p.br[tid] : 0.f;
    WLoad16<2> l1, l2, ls, l3, l4, lr;
    w16_issue(l1, p.W1, p.ldW1, D, p.k_in, D, D, tid, G16_THREADS);
    w16_issue(l2, p.W2, p.ldW2, D, D, D, D, tid, G16_THREADS);
    w16_issue(ls, p.Ws, p.ldWs, D, D, D, D, tid, G16_THREADS);
    w16_issue(l3, p.E1, p.ldE1, D, D, D, D, tid, G16_THREADS);
    w16_issue(l4, p.E2, p.ldE2, D, D, D, D, tid, G16_THREADS);
    w16_issue(lr, p.Wr, p.ldWr, D, D, D, D, tid, G16_THREADS);
    v16_commit(lv, vec, 8, tid);
    if (tid < D) brs[tid] = brv;
    w16_commit(l1, W1im, 0, p.W1, p.ldW1, D, p.k_in, D, D, tid, G16_THREADS);
    w16_commit(l2, W2im, 0, p.W2, p.ldW2, D, D, D, D, tid, G16_THREADS);
    w16_commit(ls, Wsim, 0, p.Ws, p.ldWs, D, D, D, D, tid, G16_THREADS);
    w16_commit(l3, E1im, 0, p.E1, p.ldE1, D, D, D, D, tid, G16_THREADS);
    w16_commit(l4, E2im, 0, p.E2, p.ldE2, D, D, D, D, tid, G16_THREADS);
    w16_commit(lr, Wrim, 0, p.Wr, p.ldWr, D, D, D, D, tid, G16_THREADS);
  }
  // the staging tile's columns >= k_in stay zero for the life of the kernel
  for (int i = lane; i < NLAM_T16 * G16_XLD; i += 64) XT[i] = 0.f;
  __syncthreads();

  // per lane, once: where each of its G16_SLOTS elements of a tile comes from and goes to.
  // desc = src | row << 2 | c << 6 | live << 12; dst = LDS float offset in the staging tile
  int desc[G16_SLOTS], dst[G16_SLOTS];
#pragma unroll
  for (int j = 0; j < G16_SLOTS; ++j) {
    const G16Slot s = g16_slot(p, lane + 64 * j);
    desc[j] = s.src | (s.row << 2) | (s.c << 6) | ((s.live ? 1 : 0) << 12);
    dst[j] = s.row * G16_XLD + s.col;
  }
  const int nslots = (NLAM_T16 * p.k_in + 63) >> 6;   // wave-uniform

  const int64_t tiles_per_b = (p.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * p.B;
  const int64_t stride = (int64_t)gridDim.x * G16_NW;

  // request the source elements of tile tt (rows past the end re-read the tile's last valid row)
  auto issue = [&](float (&v)[G16_SLOTS], int64_t tt) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int last = (int)((p.rows - r0) < NLAM_T16 ? (p.rows - r0) : NLAM_T16) - 1;
    const float* base[G16_MAXSRC];
    int64_t ld[G16_MAXSRC];
#pragma unroll
    for (int k = 0; k < G16_MAXSRC; ++k) {
      const int kk = k < p.nsrc ? k : 0;
      base[k] = p.src[kk].ptr + b * p.src[kk].bstride + r0 * p.src[kk].ld;
      ld[k] = p.src[kk].ld;
    }
#pragma unroll
    for (int j = 0; j < G16_SLOTS; ++j) {
      if (j < nslots) {   // wave-uniform
        const int sidx = desc[j] & 3;
        int row = (desc[j] >> 2) & 15;
        row = row < last ? row : last;
        const int c = (desc[j] >> 6) & 63;
        const float* bp = sidx == 0 ? base[0] : (sidx == 1 ? base[1] : (sidx == 2 ? base[2] : base[3]));
        const int64_t l = sidx == 0 ? ld[0] : (sidx == 1 ? ld[1] : (sidx == 2 ? ld[2] : ld[3]));
        v[j] = bp[row * l + c];
      }
    }
  };

  float nx[G16_SLOTS];
  int64_t tt = (int64_t)blockIdx.x * G16_NW + wave;
  if (tt < ntiles) issue(nx, tt);
  for (; tt < ntiles; tt += stride) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int nrows = (int)((p.rows - r0) < NLAM_T16 ? (p.rows - r0) : NLAM_T16);
    const bool valid = t < nrows;
    const int64_t orow = b * p.rows + r0;    // first output row of the tile (outputs are contiguous)

    // ---- stage: registers -> this wave's fp32 tile, then accumulator layout
    wave_sync();   // (the previous tile's reads of XT are done)
#pragma unroll
    for (int j = 0; j < G16_SLOTS; ++j)
      if (j < nslots && ((desc[j] >> 12) & 1)) XT[dst[j]] = nx[j];
    wave_sync();
    if (tt + stride < ntiles) issue(nx, tt + stride);   // next tile's rows ride under this tile
    f32x4 x[4];
    tile_to_acc16<4>(x, XT, G16_XLD, lane);
    if (p.feat != nullptr) {
      // grid_features rows: 16 x k_in floats, contiguous in the output; float4 chunks (k_in % 4 == 0
      // checked by the host side, else element stores)
      float* fo = p.feat + orow * p.k_in;
      const int cpr = p.k_in >> 2;
      if ((p.k_in & 3) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = lane + 64 * j;
          const int r = q / cpr, c4 = q - r * cpr;
          if (q < nrows * cpr)
            *reinterpret_cast<f32x4*>(fo + 4 * q) = *reinterpret_cast<const f32x4*>(XT + r * G16_XLD + 4 * c4);
        }
      } else {
#pragma unroll
        for (int j = 0; j < G16_SLOTS; ++j) {
          const int e = lane + 64 * j;
          if (j < nslots && e < nrows * p.k_in) fo[e] = XT[dst[j]];
        }
      }
    }

    // ---- grid_embedder: emb = LN(W2 silu(W1 x + b1) + b2)
    f32x4 emb[4];
    {
      f32x4 h[4];
      vec_to_acc16<4>(h, b1s, lane);
      gemm_acc16<4, 2, TERMS>(h, W1im, 0, 0, x, lane);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[fb][r] = nlam_silu(h[fb][r]);
      vec_to_acc16<4>(emb, b2s, lane);
      gemm_acc16<4, 2, TERMS>(emb, W2im, 0, 0, h, lane);
      ln16_apply<4>(emb, gs, bs, lane);
    }
    if (valid) store_row16<4>(p.emb + (orow + t) * D, emb, lane);
    // ---- sender projection of the encoder GNN
    {
      f32x4 y[4];
      zero16<4>(y);
      gemm_acc16<4, 2, TERMS>(y, Wsim, 0, 0, emb, lane);
      if (valid) store_row16<4>(p.ps + (orow + t) * D, y, lane);
    }
    // ---- encoding MLP with residual: rep = emb + LN(E2 silu(E1 emb + e1) + e2)
    f32x4 rep[4];
    {
      f32x4 h[4];
      vec_to_acc16<4>(h, e1s, lane);
      gemm_acc16<4, 2, TERMS>(h, E1im, 0, 0, emb, lane);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[fb][r] = nlam_silu(h[fb][r]);
      vec_to_acc16<4>(rep, e2s, lane);
      gemm_acc16<4, 2, TERMS>(rep, E2im, 0, 0, h, lane);
      ln16_apply<4>(rep, egs, ebs, lane);
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) rep[fb] += emb[fb];
    }
    if (valid) store_row16<4>(p.rep + (orow + t) * D, rep, lane);
    // ---- receiver projection of the decoder GNN
    {
      f32x4 y[4];
      vec_to_acc16<4>(y, brs, lane);
      gemm_acc16<4, 2, TERMS>(y, Wrim, 0, 0, rep, lane);
      if (valid) store_row16<4>(p.pr + (orow + t) * D, y, lane);
    }
  }
}

extern "C" int nlam_grid_encode_supported(void) {
  return nlam_mfma_b3() && nlam_k16_on(K16_MLP_FWD) && nlam_k16_on(K16_LIN_FWD) ? 1 : 0;
}

extern "C" int nlam_grid_encode_fwd(
    int nsrc, const float* const* src, const int64_t* src_bstride, const int64_t* src_ld,
    const int32_t* src_width, const float* W1, int64_t ldW1, const float* b1, const float* W2,
    int64_t ldW2, const float* b2, const float* gamma, const float* beta, const float* Ws,
    int64_t ldWs, const float* E1, int64_t ldE1, const float* e1, const float* E2, int64_t ldE2,
    const float* e2, const float* egamma, const float* ebeta, const float* Wr, int64_t ldWr,
    const float* br, float* feat, float* emb, float* ps, float* rep, float* pr, int64_t B,
    int64_t rows, void* stream) {
  NLAM_REQUIRE(nlam_grid_encode_supported(), "nlam_grid_encode_fwd: needs the split-bf16 16-row kernels");
  NLAM_REQUIRE(nsrc >= 1 && nsrc <= G16_MAXSRC, "nlam_grid_encode_fwd: nsrc %d out of [1, %d]", nsrc, G16_MAXSRC);
  NLAM_REQUIRE(W1 && b1 && W2 && b2 && gamma && beta && Ws && E1 && e1 && E2 && e2 && egamma && ebeta &&
                   Wr && emb && ps && rep && pr,
               "nlam_grid_encode_fwd: NULL operand");
  if (B <= 0 || rows <= 0) return 0;
  GridFwdParams p;
  p.nsrc = nsrc;
  p.k_in = 0;
  for (int k = 0; k < G16_MAXSRC; ++k) {
    if (k < nsrc) {
      NLAM_REQUIRE(src[k] != nullptr && src_width[k] >= 1 && src_ld[k] >= src_width[k] && src_ld[k] < (1 << 20),
                   "nlam_grid_encode_fwd: bad source %d", k);
      p.src[k] = RowView{src[k], src_bstride[k], src_ld[k], src_width[k]};
      p.k_in += src_width[k];
    } else {
      p.src[k] = RowView{nullptr, 0, 0, 0};
    }
  }
  NLAM_REQUIRE(p.k_in <= 64, "nlam_grid_encode_fwd: %d input columns (max 64)", p.k_in);
  NLAM_REQUIRE(nlam_aligned16(emb) && nlam_aligned16(ps) && nlam_aligned16(rep) && nlam_aligned16(pr) &&
                   (feat == nullptr || nlam_aligned16(feat)),
               "nlam_grid_encode_fwd: outputs must be 16-byte aligned");
  p.W1 = W1; p.ldW1 = ldW1; p.b1 = b1; p.W2 = W2; p.ldW2 = ldW2; p.b2 = b2;
  p.gamma = gamma; p.beta = beta; p.Ws = Ws; p.ldWs = ldWs;
  p.E1 = E1; p.ldE1 = ldE1; p.e1 = e1; p.E2 = E2; p.ldE2 = ldE2; p.e2 = e2;
  p.egamma = egamma; p.ebeta = ebeta; p.Wr = Wr; p.ldWr = ldWr; p.br = br;
  p.feat = feat; p.emb = emb; p.ps = ps; p.rep = rep; p.pr = pr;
  p.rows = rows; p.B = (int)B;
  const size_t lds = 6 * w16_image_bytes(64, 64) + 9 * 64 * sizeof(float) +
                     (size_t)G16_NW * NLAM_T16 * G16_XLD * sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "grid_fwd16: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = grid_fwd16_kernel<3>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles = ((rows + NLAM_T16 - 1) / NLAM_T16) * B;
  int64_t g = (ntiles + G16_NW - 1) / G16_NW;
  if (g > 256) g = 256;   // one 8-wave workgroup per CU (LDS)
  kern<<<(unsigned)g, G16_THREADS, lds, (hipStream_t)stream>>>(p);
  NLAM_CHECK_LAUNCH("grid_fwd16_kernel");
  return 0;
}
