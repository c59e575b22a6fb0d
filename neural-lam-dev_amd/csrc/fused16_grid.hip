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

// ---- source staging -----------------------------------------------------------------------
// A tile's source elements arrive by COALESCED dword loads and change shape through the wave's
// fp32 tile: source k's 16 x w_k block of a tile is one contiguous run of 16 w_k floats when its
// rows are dense (ld == w_k: the state / forcing slices), so lane l of a source's slot u takes
// element 64 u + l of that run.  Every source gets the same compile-time number S of slots
// (S = ceil(max_k w_k / 4); slot (k, u) is statically source k), so the per-tile part of an
// address is ONE scalar base per source (saddr form of global_load: no per-lane pointer
// arithmetic, no selects) and the per-lane part -- dword offset inside the source's tile and LDS
// offset, packed into one register per slot -- is computed once before the tile loop.  No
// branches, no predicated loads or stores in the tile loop: lanes past a run's end re-read its
// element 0 and park it in the tile's padding columns, and the last tile of a batch item is moved
// back so that it is a full 16 rows (rows it shares with its neighbour are computed twice,
// bit-identically).
template <int S>
struct G16Map {
  int desc[G16_MAXSRC * S];    // per lane: dword offset | LDS offset << 20
};
template <int S>
__device__ __forceinline__ void g16_make_map(G16Map<S>& m, const GridFwdParams& p, int lane) {
  int col0 = 0;
#pragma unroll
  for (int k = 0; k < G16_MAXSRC; ++k) {
    const int w = k < p.nsrc ? p.src[k].width : 0;
    const int wd = w > 0 ? w : 1;
#pragma unroll
    for (int u = 0; u < S; ++u) {
      const int e = 64 * u + lane;
      const bool live = e < NLAM_T16 * w;
      const int row = live ? e / wd : (lane & 15);
      const int c = live ? e - row * w : 0;
      const int off = live ? row * (int)p.src[k].ld + c : 0;
      const int dst = live ? row * G16_XLD + col0 + c : row * G16_XLD + 64 + ((lane >> 4) & 3);
      m.desc[k * S + u] = off | (dst << 20);
    }
    col0 += w;
  }
}

template <int TERMS, int S, bool FEAT>
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
  // the staging tile's columns >= k_in stay zero for the life of the kernel (its padding columns
  // 64..67 take the parked elements and are never read)
  for (int i = lane; i < NLAM_T16 * G16_XLD; i += 64) XT[i] = 0.f;
  constexpr int NSL = G16_MAXSRC * S;
  G16Map<S> map;
  g16_make_map<S>(map, p, lane);
  __syncthreads();

  const int64_t tiles_per_b = (p.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * p.B;
  const int64_t stride = (int64_t)gridDim.x * G16_NW;

  // request the source elements of tile tt
  auto issue = [&](float (&v)[NSL], int64_t tt) {
    const int64_t b = tt / tiles_per_b;
    int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    r0 = r0 + NLAM_T16 <= p.rows ? r0 : p.rows - NLAM_T16;   // the last tile is moved back: full
#pragma unroll
    for (int k = 0; k < G16_MAXSRC; ++k) {
      // (unused sources repeat source 0 in the parameter block: valid addresses, parked values)
      const float* bp = p.src[k].ptr + b * p.src[k].bstride + r0 * p.src[k].ld;   // scalar
#pragma unroll
      for (int u = 0; u < S; ++u) v[k * S + u] = bp[map.desc[k * S + u] & 0xFFFFF];
    }
  };

  float nx[NSL];
  int64_t tt = (int64_t)blockIdx.x * G16_NW + wave;
  if (tt < ntiles) issue(nx, tt);
  // the loop is entered with nothing in flight: hipcc merges its wait counts over the loop's two
  // entries, and with the first tile's loads still pending at the header every wait at the top of
  // a tile also drained the previous tile's 20 row stores (fused_edge2.hip, same remedy)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0); lgkmcnt / expcnt untouched
  for (; tt < ntiles; tt += stride) {
    const int64_t b = tt / tiles_per_b;
    int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    r0 = r0 + NLAM_T16 <= p.rows ? r0 : p.rows - NLAM_T16;
    const int64_t orow = b * p.rows + r0;    // first output row of the tile (outputs are contiguous)

    // ---- stage: registers -> this wave's fp32 tile, then accumulator layout
    wave_sync();   // (the previous tile's reads of XT are done)
#pragma unroll
    for (int j = 0; j < NSL; ++j) XT[(unsigned)map.desc[j] >> 20] = nx[j];
    wave_sync();
    // next tile's rows ride under this tile (a wave's last tile re-requests itself: no branch
    // around loads, the values are never used)
    issue(nx, tt + stride < ntiles ? tt + stride : tt);
    f32x4 x[4];
    tile_to_acc16<4>(x, XT, G16_XLD, lane);
    if constexpr (FEAT) {
      // grid_features rows: 16 x k_in floats, contiguous in the output (k_in % 4 == 0: float4)
      float* fo = p.feat + orow * p.k_in;
      const int cpr = p.k_in >> 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int q = lane + 64 * j;
        q = q < NLAM_T16 * cpr ? q : NLAM_T16 * cpr - 1;   // (clamped: duplicates store the same bytes)
        const int r = q / cpr, c4 = q - r * cpr;
        *reinterpret_cast<f32x4*>(fo + 4 * q) = *reinterpret_cast<const f32x4*>(XT + r * G16_XLD + 4 * c4);
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
    store_row16<4>(p.emb + (orow + t) * D, emb, lane);
    // ---- sender projection of the encoder GNN
    {
      f32x4 y[4];
      zero16<4>(y);
      gemm_acc16<4, 2, TERMS>(y, Wsim, 0, 0, emb, lane);
      store_row16<4>(p.ps + (orow + t) * D, y, lane);
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
    store_row16<4>(p.rep + (orow + t) * D, rep, lane);
    // ---- receiver projection of the decoder GNN
    {
      f32x4 y[4];
      vec_to_acc16<4>(y, brs, lane);
      gemm_acc16<4, 2, TERMS>(y, Wrim, 0, 0, rep, lane);
      store_row16<4>(p.pr + (orow + t) * D, y, lane);
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
  NLAM_REQUIRE(p.k_in <= 64 && p.k_in % 4 == 0, "nlam_grid_encode_fwd: %d input columns (a multiple of 4, <= 64)", p.k_in);
  NLAM_REQUIRE(rows >= NLAM_T16, "nlam_grid_encode_fwd: %lld rows (>= 16)", (long long)rows);
  int wmax = 0;
  for (int k = 0; k < nsrc; ++k) {
    wmax = src_width[k] > wmax ? src_width[k] : wmax;
    NLAM_REQUIRE(NLAM_T16 * src_ld[k] < (1 << 20), "nlam_grid_encode_fwd: source %d row pitch too large", k);
  }
  NLAM_REQUIRE(wmax <= 32, "nlam_grid_encode_fwd: a source of %d columns (max 32 per source)", wmax);
  for (int k = nsrc; k < G16_MAXSRC; ++k) p.src[k] = RowView{src[0], src_bstride[0], src_ld[0], 0};
  p.W1 = W1; p.ldW1 = ldW1; p.b1 = b1; p.W2 = W2; p.ldW2 = ldW2; p.b2 = b2;
  p.gamma = gamma; p.beta = beta; p.Ws = Ws; p.ldWs = ldWs;
  p.E1 = E1; p.ldE1 = ldE1; p.e1 = e1; p.E2 = E2; p.ldE2 = ldE2; p.e2 = e2;
  p.egamma = egamma; p.ebeta = ebeta; p.Wr = Wr; p.ldWr = ldWr; p.br = br;
  p.feat = feat; p.emb = emb; p.ps = ps; p.rep = rep; p.pr = pr;
  p.rows = rows; p.B = (int)B;
  const size_t lds = 6 * w16_image_bytes(64, 64) + 9 * 64 * sizeof(float) +
                     (size_t)G16_NW * NLAM_T16 * G16_XLD * sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "grid_fwd16: LDS footprint %zu B exceeds 160 KiB", lds);
  const int64_t ntiles = ((rows + NLAM_T16 - 1) / NLAM_T16) * B;
  int64_t g = (ntiles + G16_NW - 1) / G16_NW;
  if (g > 256) g = 256;   // one 8-wave workgroup per CU (LDS)
  // slots per source: ceil(wmax / 4), rounded up to an instantiated count
#define G16_LAUNCH(SS)                                                              \
  do {                                                                              \
    if (feat != nullptr) {                                                          \
      auto kern = grid_fwd16_kernel<3, SS, true>;                                   \
      NLAM_BIG_LDS(kern, __func__);                                                 \
      kern<<<(unsigned)g, G16_THREADS, lds, (hipStream_t)stream>>>(p);              \
    } else {                                                                        \
      auto kern = grid_fwd16_kernel<3, SS, false>;                                  \
      NLAM_BIG_LDS(kern, __func__);                                                 \
      kern<<<(unsigned)g, G16_THREADS, lds, (hipStream_t)stream>>>(p);              \
    }                                                                               \
  } while (0)
  const int need = (wmax + 3) / 4;
  if (need <= 3) G16_LAUNCH(3);
  else if (need <= 5) G16_LAUNCH(5);
  else G16_LAUNCH(8);
#undef G16_LAUNCH
  NLAM_CHECK_LAUNCH("grid_fwd16_kernel");
  return 0;
}
