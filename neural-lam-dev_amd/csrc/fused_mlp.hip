// Fused row-MLP kernels (gfx950): the make_mlp blocks of the reference
// (utils.py:191-214) with hidden_layers == 1,
//     y = [res +] [LayerNorm]( W2 silu(W1 [x_a | x_b] + b1) + b2 ),
// and the single-Linear "projection" used by the algebraic split of the edge
// MLP's first layer (W1 [e; x_s; x_r] = W1e e + W1s x_s + W1r x_r).
//
// One wavefront owns a tile of 32 rows (fused_common.h); a 256-thread workgroup
// keeps the weights in LDS and walks tiles persistently.  Forward saves nothing;
// backward recomputes the hidden activations from the inputs.
#include <type_traits>
#include <cstdlib>
#include "fused_common.h"
#include "fused_bf16x3.h"
#include "fused_fs.h"
#include "fused_params.h"


// LDS layout (floats): W1s[HID][k_pad+4] | W2s[32*NOUTB][HID+4] | b1s[HID] | b2s | gs | bs |
//                      tiles[4][32][ldt],  ldt = max(k_pad, HID, 32*NOUTB) + 4
template <int HID, int NOUTB>
__device__ __forceinline__ int mlp_ldt(int k_pad) {
  int m = k_pad;
  if (HID > m) m = HID;
  if (32 * NOUTB > m) m = 32 * NOUTB;
  return m + 4;
}

// B3KB > 0: split-bf16 GEMMs (fused_bf16x3.h) with K padded to 32 * B3KB columns.
template <int HID, int NOUTB, bool HAS_LN, int B3KB = 0>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NBH = HID / 32;
  constexpr int NVS = HID / 8;   // float4 per lane per source (source width <= HID)
  constexpr bool B3 = B3KB > 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool fast_stage = (p.vec_mask & 1) && (p.nsrc == 1 || ((p.vec_mask >> 1) & 1));
  const int kp = B3 ? 32 * B3KB : p.k_pad;     // staged / zero-padded K
  const int ldw1 = kp + 4, ldw2 = HID + 4;
  const int ldt = mlp_ldt<HID, NOUTB>(kp);
  float* W1s = smem;
  float* W2s = W1s + HID * ldw1;               // (the bf16 images have the fp32 byte sizes)
  float* b1s = W2s + 32 * NOUTB * ldw2;
  float* b2s = b1s + HID;
  float* gs = b2s + 32 * NOUTB;
  float* bs = gs + 32 * NOUTB;
  float* tile = bs + 32 * NOUTB + wave * (NLAM_TILE * ldt);
  const B3Image W1im = b3_image(W1s, HID, B3 ? 32 * B3KB : 32);
  const B3Image W2im = b3_image(W2s, 32 * NOUTB, HID);

  if (B3) {
    load_weight_lds_b3(W1im, 0, p.W1, p.ldW1, HID, p.k_in, HID, 32 * B3KB, tid, 256);
    load_weight_lds_b3(W2im, 0, p.W2, p.ldW2, p.n_out, HID, 32 * NOUTB, HID, tid, 256);
  } else {
    load_weight_lds(W1s, p.W1, p.ldW1, HID, p.k_in, HID, p.k_pad, tid, 256);
    load_weight_lds(W2s, p.W2, p.ldW2, p.n_out, HID, 32 * NOUTB, HID, tid, 256);
  }
  load_vec_lds(b1s, p.b1, HID, HID, tid, 256);
  load_vec_lds(b2s, p.b2, p.n_out, 32 * NOUTB, tid, 256);
  load_vec_lds(gs, p.gamma, p.n_out, 32 * NOUTB, tid, 256);
  load_vec_lds(bs, p.beta, p.n_out, 32 * NOUTB, tid, 256);
  __syncthreads();

  const int64_t tiles_per_b = (p.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * p.B;
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < ntiles; tt += (int64_t)gridDim.x * 4) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((p.rows - r0) < NLAM_TILE ? (p.rows - r0) : NLAM_TILE);
    // stage [x_a | x_b] rows: both sources' loads in flight before the LDS writes
    if (fast_stage) {
      f32x4 va[NVS], vb[NVS];
      view_load_v<NVS>(va, p.src[0], b, r0, nrows, lane);
      if (p.nsrc > 1) view_load_v<NVS>(vb, p.src[1], b, r0, nrows, lane);
      put_rows_v<NVS, false>(tile, ldt, 0, p.src[0].width, nrows, lane, va);
      if (p.nsrc > 1)
        put_rows_v<NVS, false>(tile, ldt, p.src[0].width, p.src[1].width, nrows, lane, vb);
    } else {
      view_stage_s(tile, ldt, 0, p.src[0], b, r0, nrows, lane);
      if (p.nsrc > 1) view_stage_s(tile, ldt, p.src[0].width, p.src[1], b, r0, nrows, lane);
    }
    if (kp > p.k_in) zero_cols(tile, ldt, p.k_in, kp - p.k_in, lane);
    wave_sync();

    f32x16 a1[NBH];
    vec_to_acc<NBH>(a1, b1s, lane);
    if constexpr (B3) {
      f32x16 xin[B3KB];
      tile_to_acc<B3KB>(xin, tile, ldt, lane);
      gemm_acc_b3<NBH, B3KB>(a1, W1im, 0, xin, lane);
    } else {
      gemm_tile<NBH>(a1, W1s, ldw1, tile, ldt, p.k_pad >> 3, lane);
    }
#pragma unroll
    for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[nb][r] = nlam_silu(a1[nb][r]);
    f32x16 a2[NOUTB];
    vec_to_acc<NOUTB>(a2, b2s, lane);
    if constexpr (B3) gemm_acc_b3<NOUTB, NBH>(a2, W2im, 0, a1, lane);
    else gemm_acc<NOUTB, NBH>(a2, W2s, ldw2, 0, a1, lane);
    if (HAS_LN) ln_apply<NOUTB>(a2, gs, bs, lane);

    wave_sync();  // all operand reads of the tile are done
    acc_to_tile<NOUTB>(a2, tile, ldt, lane);
    wave_sync();
    float* ob = p.out + b * p.out_bstride + r0 * p.out_ld;
    auto op = [&](int t) { return ob + (int64_t)t * p.out_ld; };
    if (p.res != nullptr) {
      const float* rb = p.res + b * p.res_bstride + r0 * p.res_ld;
      auto rp = [&](int t) { return rb + (int64_t)t * p.res_ld; };
      if (((p.vec_mask >> 2) & 1) && ((p.vec_mask >> 3) & 1))
        store_rows_res<true, 1>(tile, ldt, 0, p.n_out, nrows, lane, op, rp);   // (2 waves / SIMD:
        // larger batches of residual loads spill here and cost more than they hide)
      else
        store_rows_res<false>(tile, ldt, 0, p.n_out, nrows, lane, op, rp);
    } else {
      if ((p.vec_mask >> 3) & 1)
        store_rows<true>(tile, ldt, 0, p.n_out, nrows, lane, op);
      else
        store_rows<false>(tile, ldt, 0, p.n_out, nrows, lane, op);
    }
    wave_sync();  // stores read the tile; next iteration overwrites it
  }
}

template <int HID, int NOUTB>
static size_t mlp_lds_bytes(int k_pad) {
  int m = k_pad;
  if (HID > m) m = HID;
  if (32 * NOUTB > m) m = 32 * NOUTB;
  const int ldt = m + 4;
  size_t fl = (size_t)HID * (k_pad + 4) + (size_t)32 * NOUTB * (HID + 4) + HID + 3 * 32 * NOUTB +
              (size_t)4 * NLAM_TILE * ldt;
  return fl * sizeof(float);
}

template <int HID, int NOUTB, bool HAS_LN, int B3KB = 0>
static int launch_mlp_fwd(const MlpParams& p, hipStream_t s) {
  const size_t lds = mlp_lds_bytes<HID, NOUTB>(B3KB > 0 ? 32 * B3KB : p.k_pad);
  NLAM_REQUIRE(lds <= 160 * 1024, "mlp_fwd: LDS footprint %zu B exceeds 160 KiB (k_in=%d)", lds,
               p.k_in);
  auto kern = mlp_fwd_kernel<HID, NOUTB, HAS_LN, B3KB>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles = ((p.rows + NLAM_TILE - 1) / NLAM_TILE) * p.B;
  kern<<<persistent_grid(ntiles, lds), 256, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("mlp_fwd_kernel");
  return 0;
}

// C ABI -----------------------------------------------------------------------
extern "C" int nlam_mlp_fwd(
    const float* xa, int64_t xa_bstride, int64_t xa_ld, int xa_width,
    const float* xb, int64_t xb_bstride, int64_t xb_ld, int xb_width,
    const float* W1, int64_t ldW1, const float* b1, const float* W2, int64_t ldW2,
    const float* b2, const float* gamma, const float* beta,
    const float* res, int64_t res_bstride, int64_t res_ld,
    float* out, int64_t out_bstride, int64_t out_ld,
    int64_t B, int64_t rows, int hid, int n_out, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(hid == 64 || hid == 128, "nlam_mlp_fwd: hidden width %d not in {64,128}", hid);
  NLAM_REQUIRE(n_out >= 1 && n_out <= hid, "nlam_mlp_fwd: n_out %d out of range", n_out);
  NLAM_REQUIRE(xa != nullptr && xa_width >= 1, "nlam_mlp_fwd: first source missing");
  NLAM_REQUIRE((gamma == nullptr) == (beta == nullptr), "nlam_mlp_fwd: gamma/beta mismatch");
  NLAM_REQUIRE(gamma == nullptr || n_out == hid, "nlam_mlp_fwd: LayerNorm needs n_out == hid");
  MlpParams p;
  p.src[0] = RowView{xa, xa_bstride, xa_ld, xa_width};
  p.src[1] = RowView{xb, xb_bstride, xb_ld, xb ? xb_width : 0};
  p.nsrc = xb ? 2 : 1;
  p.k_in = xa_width + (xb ? xb_width : 0);
  p.k_pad = (p.k_in + 7) & ~7;
  p.n_out = n_out;
  p.W1 = W1; p.ldW1 = ldW1; p.b1 = b1; p.W2 = W2; p.ldW2 = ldW2; p.b2 = b2;
  p.gamma = gamma; p.beta = beta;
  p.res = res; p.res_bstride = res_bstride; p.res_ld = res_ld;
  p.out = out; p.out_bstride = out_bstride; p.out_ld = out_ld;
  p.rows = rows; p.B = (int)B;
  p.vec_mask = 0;
  if (view_vec_ok(xa, xa_bstride, xa_ld, xa_width)) p.vec_mask |= 1;
  // the second source lands at column xa_width of the LDS tile: keep it 16-B aligned
  if (xb && view_vec_ok(xb, xb_bstride, xb_ld, xb_width) && (xa_width % 4 == 0)) p.vec_mask |= 2;
  if (res && view_vec_ok(res, res_bstride, res_ld, n_out)) p.vec_mask |= 4;
  if (view_vec_ok(out, out_bstride, out_ld, n_out)) p.vec_mask |= 8;
  hipStream_t s = (hipStream_t)stream;
  if (hid == 64) {   // 16-row form (fused16_mlp.hip) where it applies
    const int r16 = nlam_k16_mlp_fwd(p, s);
    if (r16 >= 0) return r16;
  }
  const bool ln = gamma != nullptr;
  const int noutb = (n_out + 31) / 32;
  if (hid == 64 && nlam_mfma_b3()) {
    const int kb = (p.k_in + 31) / 32;
    if (ln && kb == 1) return launch_mlp_fwd<64, 2, true, 1>(p, s);
    if (ln && kb == 2) return launch_mlp_fwd<64, 2, true, 2>(p, s);
    if (ln && kb == 4) return launch_mlp_fwd<64, 2, true, 4>(p, s);
    if (!ln && noutb == 1 && kb == 2) return launch_mlp_fwd<64, 1, false, 2>(p, s);
  }
  if (hid == 64) {
    if (ln) return launch_mlp_fwd<64, 2, true>(p, s);
    if (noutb == 1) return launch_mlp_fwd<64, 1, false>(p, s);
    return launch_mlp_fwd<64, 2, false>(p, s);
  }
  if (ln) return launch_mlp_fwd<128, 4, true>(p, s);
  if (noutb == 1) return launch_mlp_fwd<128, 1, false>(p, s);
  NLAM_REQUIRE(noutb == 4, "nlam_mlp_fwd: hid 128 supports n_out <= 32 or == 128 without LN");
  return launch_mlp_fwd<128, 4, false>(p, s);
}

// ------------------------------------------------------------- projection
// out[:, 0:nA] = x WA^T + bA ; out[:, nA:nA+nB] = x WB^T + bB  (WB optional).

// Diagnostic (NLAM_TIMELINE=1): s_memrealtime (100 MHz, chip-wide) at workgroup start,
// after the weight prologue and after the tile loop, for up to 1024 workgroups.
__device__ unsigned long long g_lin_fwd_timeline[3 * 1024];
extern "C" int nlam_debug_lin_fwd_timeline(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lin_fwd_timeline),
                             sizeof(unsigned long long) * 3 * 1024) == hipSuccess ? 0 : 1;
}

template <int NOUTB>
__global__ __launch_bounds__(256, 2) void lin_fwd_kernel(LinParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool tl = p.timeline && tid == 0 && blockIdx.x < 1024;
  if (tl) g_lin_fwd_timeline[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  const int ldw = p.k_pad + 4;
  const int n_out = p.nA + p.nB;
  const int ldt = (p.k_pad > 32 * NOUTB ? p.k_pad : 32 * NOUTB) + 4;
  float* Ws = smem;
  float* bs = Ws + 32 * NOUTB * ldw;
  float* tile = bs + 32 * NOUTB + wave * (NLAM_TILE * ldt);
  load_weight_lds(Ws, p.WA, p.ldWA, p.nA, p.x.width, p.nA, p.k_pad, tid, 256);
  load_vec_lds(bs, p.bA, p.nA, p.nA, tid, 256);
  if (p.nB > 0) {
    load_weight_lds(Ws + p.nA * ldw, p.WB, p.ldWB, p.nB, p.x.width, 32 * NOUTB - p.nA, p.k_pad,
                    tid, 256);
    load_vec_lds(bs + p.nA, p.bB, p.nB, 32 * NOUTB - p.nA, tid, 256);
  }
  __syncthreads();
  if (tl) g_lin_fwd_timeline[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  const int64_t tiles_per_b = (p.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * p.B;
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < ntiles; tt += (int64_t)gridDim.x * 4) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((p.rows - r0) < NLAM_TILE ? (p.rows - r0) : NLAM_TILE);
    if (p.vec_mask & 1) {
      f32x4 vx[16];
      view_load_v<16>(vx, p.x, b, r0, nrows, lane);
      put_rows_v<16, false>(tile, ldt, 0, p.x.width, nrows, lane, vx);
    } else {
      view_stage_s(tile, ldt, 0, p.x, b, r0, nrows, lane);
    }
    if (p.k_pad > p.x.width) zero_cols(tile, ldt, p.x.width, p.k_pad - p.x.width, lane);
    wave_sync();
    f32x16 a[NOUTB];
    vec_to_acc<NOUTB>(a, bs, lane);
    gemm_tile<NOUTB>(a, Ws, ldw, tile, ldt, p.k_pad >> 3, lane);
    wave_sync();
    acc_to_tile<NOUTB>(a, tile, ldt, lane);
    wave_sync();
    float* ob = p.out + b * p.out_bstride + r0 * p.out_ld;
    auto op = [&](int t) { return ob + (int64_t)t * p.out_ld; };
    if ((p.vec_mask >> 3) & 1)
      store_rows<true>(tile, ldt, 0, n_out, nrows, lane, op);
    else
      store_rows<false>(tile, ldt, 0, n_out, nrows, lane, op);
    wave_sync();
  }
  if (tl) {
    __builtin_amdgcn_s_waitcnt(0);
    g_lin_fwd_timeline[3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
  }
}

// Split-bf16 form of the projection (fused_bf16x3.h): k_in % 32 == 0, float4 views.
template <int NOUTB, int KB, int TERMS = 3, int MINW = 2>
__global__ __launch_bounds__(256, MINW) void lin_fwd_b3_kernel(LinParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int K = 32 * KB, NO = 32 * NOUTB;
  constexpr int ldt = (K > NO ? K : NO) + 4;
  const B3Image W = b3_image(smem, NO, K);
  float* bs = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + b3_image_bytes(NO, K));
  float* tile = bs + NO + wave * (NLAM_TILE * ldt);
  load_weight_lds_b3(W, 0, p.WA, p.ldWA, p.nA, K, p.nA, K, tid, 256);
  load_vec_lds(bs, p.bA, p.nA, p.nA, tid, 256);
  if (p.nB > 0) {
    load_weight_lds_b3(W, p.nA, p.WB, p.ldWB, p.nB, K, NO - p.nA, K, tid, 256);
    load_vec_lds(bs + p.nA, p.bB, p.nB, NO - p.nA, tid, 256);
  }
  __syncthreads();
  const int n_out = p.nA + p.nB;
  const int64_t tiles_per_b = (p.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * p.B;
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < ntiles; tt += (int64_t)gridDim.x * 4) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((p.rows - r0) < NLAM_TILE ? (p.rows - r0) : NLAM_TILE);
    f32x4 vx[8 * KB];
    view_load_v<8 * KB>(vx, p.x, b, r0, nrows, lane);
    const B3Tile Xp = b3_tile(tile, K);
    put_rows_v_b3<8 * KB>(Xp, 0, K, nrows, lane, vx);
    wave_sync();
    f32x16 a[NOUTB];
    vec_to_acc<NOUTB>(a, bs, lane);
    gemm_tile_b3<NOUTB, KB, TERMS>(a, W, 0, Xp, 0, lane);
    wave_sync();
    acc_to_tile<NOUTB>(a, tile, ldt, lane);
    wave_sync();
    float* ob = p.out + b * p.out_bstride + r0 * p.out_ld;
    auto op = [&](int t) { return ob + (int64_t)t * p.out_ld; };
    store_rows<true>(tile, ldt, 0, n_out, nrows, lane, op);
    wave_sync();
  }
}

template <int NOUTB, int KB, int TERMS = 3, int MINW = 2>
static int launch_lin_fwd_b3(const LinParams& p, hipStream_t s) {
  constexpr int K = 32 * KB, NO = 32 * NOUTB;
  constexpr int ldt = (K > NO ? K : NO) + 4;
  const size_t lds = b3_image_bytes(NO, K) + NO * 4 + (size_t)4 * NLAM_TILE * ldt * 4;
  NLAM_REQUIRE(lds <= 160 * 1024, "lin_fwd_b3: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = lin_fwd_b3_kernel<NOUTB, KB, TERMS, MINW>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles = ((p.rows + NLAM_TILE - 1) / NLAM_TILE) * p.B;
  kern<<<persistent_grid(ntiles, lds), 256, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("lin_fwd_b3_kernel");
  return 0;
}

template <int NOUTB>
static int launch_lin_fwd(const LinParams& p, hipStream_t s) {
  const int ldt = (p.k_pad > 32 * NOUTB ? p.k_pad : 32 * NOUTB) + 4;
  const size_t lds = ((size_t)32 * NOUTB * (p.k_pad + 4) + 32 * NOUTB + (size_t)4 * NLAM_TILE * ldt) *
                     sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "lin_fwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = lin_fwd_kernel<NOUTB>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles = ((p.rows + NLAM_TILE - 1) / NLAM_TILE) * p.B;
  kern<<<persistent_grid(ntiles, lds), 256, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("lin_fwd_kernel");
  return 0;
}

extern "C" int nlam_lin_fwd(const float* x, int64_t x_bstride, int64_t x_ld, int k_in,
                            const float* WA, int64_t ldWA, const float* bA, int nA,
                            const float* WB, int64_t ldWB, const float* bB, int nB,
                            float* out, int64_t out_bstride, int64_t out_ld, int64_t B,
                            int64_t rows, int out_bf16, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(!out_bf16 || (nlam_mfma_terms() == 1 && nA == 256 && (WB == nullptr || nB == 0)),
               "nlam_lin_fwd: bf16 output rows exist for the hidden-256 bf16 path only");
  NLAM_REQUIRE(nA > 0 && nA % 32 == 0 && nB >= 0 && nB % 32 == 0,
               "nlam_lin_fwd: output block widths must be multiples of 32 (got %d, %d)", nA, nB);
  NLAM_REQUIRE(k_in >= 1 && k_in <= 256, "nlam_lin_fwd: k_in %d out of range", k_in);
  LinParams p;
  p.x = RowView{x, x_bstride, x_ld, k_in};
  p.k_pad = (k_in + 7) & ~7;
  p.WA = WA; p.ldWA = ldWA; p.bA = bA; p.nA = nA;
  p.WB = WB; p.ldWB = ldWB; p.bB = bB; p.nB = WB ? nB : 0;
  p.out = out; p.out_bstride = out_bstride; p.out_ld = out_ld;
  p.rows = rows; p.B = (int)B;
  p.vec_mask = 0;
  static const bool timeline = getenv("NLAM_TIMELINE") && getenv("NLAM_TIMELINE")[0] == '1';
  p.timeline = timeline ? 1 : 0;
  if (view_vec_ok(x, x_bstride, x_ld, k_in)) p.vec_mask |= 1;
  if (view_vec_ok(out, out_bstride, out_ld, p.nA + p.nB)) p.vec_mask |= 8;
  hipStream_t s = (hipStream_t)stream;
  // hidden 256 (fused_fs.hip): register-stationary weight slices, bf16 or split-bf16 operands
  if (nlam_mfma_terms() != 0 && nA == 256 && p.nB == 0 && (p.vec_mask & 8))
    return nlam_fs_lin_fwd_256(x, x_bstride, x_ld, k_in, WA, ldWA, bA, nA, out, out_bstride, out_ld,
                               B, rows, out_bf16, stream);
  NLAM_REQUIRE(!out_bf16, "nlam_lin_fwd: bf16 output rows need 16-byte aligned rows, pitch %% 8 == 0");
  {
    const int r16 = nlam_k16_lin_fwd(p, s);
    if (r16 >= 0) return r16;
  }
  if (nlam_mfma_b3() && k_in == 64 && (p.vec_mask & 1) && (p.vec_mask & 8)) {
    if ((p.nA + p.nB) == 64) return launch_lin_fwd_b3<2, 2>(p, s);
    if ((p.nA + p.nB) == 128) return launch_lin_fwd_b3<4, 2>(p, s);
  }
  // hidden 128 (fused_wide.hip): one 128 x 128 image, one wave per SIMD
  if (nlam_mfma_terms() != 0 && k_in == 128 && (p.nA + p.nB) == 128 && (p.vec_mask & 1) &&
      (p.vec_mask & 8))
    return nlam_mfma_terms() == 3 ? launch_lin_fwd_b3<4, 4, 3, 1>(p, s)
                                  : launch_lin_fwd_b3<4, 4, 1, 1>(p, s);
  switch ((p.nA + p.nB) / 32) {
    case 2: return launch_lin_fwd<2>(p, s);
    case 4: return launch_lin_fwd<4>(p, s);
    case 8: return launch_lin_fwd<8>(p, s);
    default:
      nlam_set_error("nlam_lin_fwd: total output width %d unsupported", p.nA + p.nB);
      return 1;
  }
}

// =============================================================== backward ===
// Per-workgroup partial parameter gradients go to a slab
//   [dW1 (HID x KP32) | db1 (HID) | dW2 (32 NOUTB x HID) | db2 | dgamma | dbeta]
// (KP32 = k_in rounded up to 32); nlam_reduce_slabs sums the slabs in a fixed
// order.  Activations are recomputed from the inputs (nothing saved in forward).

__device__ unsigned long long g_mlp_bwd_stamps[8];
extern "C" int nlam_debug_mlp_bwd_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_bwd_stamps), sizeof(unsigned long long) * 8) !=
      hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_bwd_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#define MSTAMP(k)                                                   \
  if (q.stamp) {                                                    \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    __builtin_amdgcn_s_waitcnt(0xC07F);                             \
    __builtin_amdgcn_sched_barrier(0);                              \
    mst[k] += now_ - mprev;                                         \
    mprev = now_;                                                   \
  }

// DEFER_DW1: the first layer's weight gradient (HID x KP32 = up to 128 accumulator
// registers per wave) is not formed here; the kernel stores ga = dL/d(pre-activation)
// and nlam_outer_bwd computes dW1 = ga^T [x_a | x_b], db1 = colsum(ga) in a lean second
// pass.  Used for the K = 128 node update, where the in-kernel form spilled.
// B3: every GEMM / outer product as split-bf16 MFMAs (fused_bf16x3.h).
// XP (B3, K <= 64, float4 views): the input rows are staged ONCE, as bf16 planes in a third
// tile that stays put for the whole tile (B operand of the first GEMM and of the dW1 outer
// product), and the x / gy rows of the NEXT tile are loaded into registers in the middle of
// the current one.  Stamps of the plain form: 25 % of a tile waiting for its rows, another
// 10 % re-staging X for dW1.
template <int HID, int NOUTB, int KB, bool HAS_LN, bool DEFER_DW1, bool B3 = false, bool XP = false>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(MlpBwdParams q) {
  static_assert(!XP || (B3 && !DEFER_DW1 && KB <= 2), "XP: B3, in-kernel dW1, K <= 64");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NBH = HID / 32;
  constexpr int KP32 = 32 * KB;
  constexpr int NV_H = (HID + 63) / 64, NV_O = (32 * NOUTB + 63) / 64;
  constexpr int NVS = HID / 8;
  const MlpParams& p = q.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int ldw1 = KP32 + 4, ldw2 = HID + 4;
  constexpr int ldt0 = (KP32 > HID ? KP32 : HID) + 4;              // X / S / GX tile
  constexpr int ldt1 = (HID > 32 * NOUTB ? HID : 32 * NOUTB) + 4;  // GY / GZ / GA tile
  constexpr int WSTRIDE = NLAM_TILE * (ldt0 + ldt1);
  float* W1s = smem;
  float* W2s = W1s + HID * ldw1;
  float* b1s = W2s + 32 * NOUTB * ldw2;
  float* b2s = b1s + HID;
  float* gs = b2s + 32 * NOUTB;
  float* T0base = gs + 32 * NOUTB;
  float* T1base = T0base + NLAM_TILE * ldt0;
  float* T0 = T0base + wave * WSTRIDE;
  float* T1 = T1base + wave * WSTRIDE;
  float* T2 = T0base + 4 * WSTRIDE + wave * (NLAM_TILE * (KP32 + 4));   // XP: X planes
  const B3Tile T2x = b3_tile(T2, KP32);

  const B3Image W1im = b3_image(W1s, HID, KP32), W2im = b3_image(W2s, 32 * NOUTB, HID);
  if (B3) {
    load_weight_lds_b3(W1im, 0, p.W1, p.ldW1, HID, p.k_in, HID, KP32, tid, 256);
    load_weight_lds_b3(W2im, 0, p.W2, p.ldW2, p.n_out, HID, 32 * NOUTB, HID, tid, 256);
  } else {
    load_weight_lds(W1s, p.W1, p.ldW1, HID, p.k_in, HID, KP32, tid, 256);
    load_weight_lds(W2s, p.W2, p.ldW2, p.n_out, HID, 32 * NOUTB, HID, tid, 256);
  }
  load_vec_lds(b1s, p.b1, HID, HID, tid, 256);
  load_vec_lds(b2s, p.b2, p.n_out, 32 * NOUTB, tid, 256);
  load_vec_lds(gs, p.gamma, p.n_out, 32 * NOUTB, tid, 256);
  __syncthreads();
  // bf16-plane views of the two tiles (operands of the outer products)
  const B3Tile T0s = b3_tile(T0, HID), T0x = b3_tile(T0, KP32);
  const B3Tile T1o = b3_tile(T1, 32 * NOUTB), T1h = b3_tile(T1, HID);

  constexpr int KBA = DEFER_DW1 ? 1 : KB;   // no dW1 accumulators when deferred
  f32x16 dW1[NBH][KBA], dW2[NOUTB][NBH];
#pragma unroll
  for (int i = 0; i < NBH; ++i)
#pragma unroll
    for (int j = 0; j < KBA; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW1[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < NOUTB; ++i)
#pragma unroll
    for (int j = 0; j < NBH; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW2[i][j][r] = 0.f;
  float db1[NV_H], db2[NV_O], dgam[NV_O], dbet[NV_O];
#pragma unroll
  for (int j = 0; j < NV_H; ++j) db1[j] = 0.f;
#pragma unroll
  for (int j = 0; j < NV_O; ++j) db2[j] = dgam[j] = dbet[j] = 0.f;

  const bool fast_stage = (p.vec_mask & 1) && (p.nsrc == 1 || ((p.vec_mask >> 1) & 1));
  auto stage_x = [&](int64_t b, int64_t r0, int nrows) {
    if (fast_stage) {
      f32x4 va[NVS], vb[NVS];
      view_load_v<NVS>(va, p.src[0], b, r0, nrows, lane);
      if (p.nsrc > 1) view_load_v<NVS>(vb, p.src[1], b, r0, nrows, lane);
      put_rows_v<NVS, false>(T0, ldt0, 0, p.src[0].width, nrows, lane, va);
      if (p.nsrc > 1)
        put_rows_v<NVS, false>(T0, ldt0, p.src[0].width, p.src[1].width, nrows, lane, vb);
    } else {
      view_stage_s(T0, ldt0, 0, p.src[0], b, r0, nrows, lane);
      if (p.nsrc > 1) view_stage_s(T0, ldt0, p.src[0].width, p.src[1], b, r0, nrows, lane);
    }
    if (KP32 > p.k_in) zero_cols(T0, ldt0, p.k_in, KP32 - p.k_in, lane);
  };

  const int64_t tiles_per_b = (p.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * p.B;
  unsigned long long mst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long mprev = q.stamp ? __builtin_amdgcn_s_memtime() : 0;
  // XP: rows of the next tile, in flight from the middle of the current one
  f32x4 pva[XP ? NVS : 1], pvg[XP ? NVS : 1];   // (XP takes single-source inputs only)
  // narrow gy (the 17-wide output map, n_out <= 24): dword prefetch, 12 values per lane
  constexpr int NGS = 12;
  constexpr bool GYS = XP && NOUTB == 1 && !HAS_LN;
  float pgs[GYS ? NGS : 1];
  auto issue_next = [&](int64_t task) {
    if constexpr (XP) {
      const int64_t tq = task < ntiles ? task : ntiles - 1;
      const int64_t bq = tq / tiles_per_b;
      const int64_t rq = (tq - bq * tiles_per_b) * NLAM_TILE;
      const int nq = (int)((p.rows - rq) < NLAM_TILE ? (p.rows - rq) : NLAM_TILE);
      view_load_v<NVS>(pva, p.src[0], bq, rq, nq, lane);
      if constexpr (GYS) view_load_s<NGS>(pgs, q.gy, bq, rq, nq, lane);
      else view_load_v<NVS>(pvg, q.gy, bq, rq, nq, lane);
    }
  };
  if (XP && (int64_t)blockIdx.x * 4 + wave < ntiles) issue_next((int64_t)blockIdx.x * 4 + wave);
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < ntiles; tt += (int64_t)gridDim.x * 4) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nfull = (int)((p.rows - r0) < NLAM_TILE ? (p.rows - r0) : NLAM_TILE);
    const int nrows = nfull;
    if constexpr (XP) {
      // ---- the prefetched rows: X as bf16 planes (T2, kept), gy as fp32 (T1)
      put_rows_v_b3<NVS>(T2x, 0, p.src[0].width, nrows, lane, pva);
      if (KP32 > p.k_in) {
        const int padw = KP32 - p.k_in;
        for (int idx = lane; idx < NLAM_TILE * padw; idx += 64) {
          const int tr = idx / padw, cc = p.k_in + idx - tr * padw;
          T2x.hi[tr * T2x.pitch + cc] = (__bf16)0.f;
          T2x.lo[tr * T2x.pitch + cc] = (__bf16)0.f;
        }
      }
      if constexpr (GYS) view_put_s<NGS>(T1, ldt1, 0, p.n_out, nrows, lane, pgs);
      else put_rows_v<NVS, false>(T1, ldt1, 0, p.n_out, nrows, lane, pvg);
    } else {
    // ---- recompute forward; gy rows go to T1 right away (latencies overlap)
    stage_x(b, r0, nrows);
    if (q.vec_gy) {
      f32x4 vg[NVS];
      view_load_v<NVS>(vg, q.gy, b, r0, nfull, lane);
      put_rows_v<NVS, false>(T1, ldt1, 0, p.n_out, nrows, lane, vg);
    } else {
      view_stage_s(T1, ldt1, 0, q.gy, b, r0, nrows, lane);
    }
    }
    if (32 * NOUTB > p.n_out) zero_cols(T1, ldt1, p.n_out, 32 * NOUTB - p.n_out, lane);
    wave_sync();
    MSTAMP(0)   // x and gy rows landed + staged
    f32x16 hpre[NBH];
    vec_to_acc<NBH>(hpre, b1s, lane);
    if constexpr (XP) {
      gemm_tile_b3<NBH, KB>(hpre, W1im, 0, T2x, 0, lane);   // B operand straight from the planes
    } else if constexpr (B3) {
      constexpr int XB = KB >= 2 ? 2 : 1;      // two input blocks at a time (registers)
#pragma unroll
      for (int kb0 = 0; kb0 < KB; kb0 += XB) {
        f32x16 xin[XB];
        tile_to_acc<XB>(xin, T0 + 32 * kb0, ldt0, lane);
        gemm_acc_b3<NBH, XB>(hpre, W1im, kb0, xin, lane);
      }
    } else {
      gemm_tile<NBH>(hpre, W1s, ldw1, T0, ldt0, p.k_pad >> 3, lane);
    }
    f32x16 sact[NBH];
#pragma unroll
    for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sact[nb][r] = nlam_silu(hpre[nb][r]);
    MSTAMP(1)   // GEMM1 + silu
    f32x16 g[NOUTB];
    tile_to_acc<NOUTB>(g, T1, ldt1, lane);
    if (HAS_LN) {
      tile_colsum_all<NV_O>(dbet, T1, ldt1, 0, lane);
      f32x16 z[NOUTB];
      vec_to_acc<NOUTB>(z, b2s, lane);
      if constexpr (B3) gemm_acc_b3<NOUTB, NBH>(z, W2im, 0, sact, lane);
      else gemm_acc<NOUTB, NBH>(z, W2s, ldw2, 0, sact, lane);
      // S goes to T0 right away (X was consumed by the first GEMM): sact's
      // registers are free during the LayerNorm backward
      wave_sync();
      if constexpr (B3) acc_to_tile_b3<NBH>(sact, T0s, 0, lane);
      else acc_to_tile<NBH>(sact, T0, ldt0, lane);
      // LayerNorm backward: z -> xhat in place; g (= gy) -> gz
      constexpr float inv_d = 1.0f / (32.0f * NOUTB);
      float mean, rstd;
      ln_stats<NOUTB>(z, mean, rstd);
      const int hh = lane >> 5;
      float s1 = 0.f, s2 = 0.f;
      wave_sync();   // (the gy tile has been read: T1 takes gy * xhat block by block)
#pragma unroll
      for (int nb = 0; nb < NOUTB; ++nb) {
        f32x16 prod[1];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(gs + 32 * nb + 8 * qq + 4 * hh);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * qq + j;
            const float xh = (z[nb][r] - mean) * rstd;
            z[nb][r] = xh;
            prod[0][r] = g[nb][r] * xh;         // gy * xhat -> dgamma
            const float gv = g[nb][r] * gm[j];
            g[nb][r] = gv;
            s1 += gv;
            s2 += gv * xh;
          }
        }
        acc_to_tile<1>(prod, T1 + 32 * nb, ldt1, lane);
      }
      s1 = lane_xor32_sum(s1);
      s2 = lane_xor32_sum(s2);
      const float m1 = s1 * inv_d, m2 = s2 * inv_d;
      wave_sync();
      tile_colsum_all<NV_O>(dgam, T1, ldt1, 0, lane);
#pragma unroll
      for (int nb = 0; nb < NOUTB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[nb][r] = rstd * (g[nb][r] - m1 - z[nb][r] * m2);
    }
    // g is gz (zero on padded rows).  Publish GZ (T1); S is in T0.
    MSTAMP(2)   // GEMM2 + LayerNorm backward (+ two column sums)
    wave_sync();
    // column sums on the matrix cores where the register budget allows (the K = 64 form
    // with both weight gradients in registers spills with it: measured 145 -> 198 us)
    constexpr bool MCS = B3 && !(KB == 2 && HAS_LN && !DEFER_DW1);
    if constexpr (MCS) {
      // GZ straight to bf16 planes; its column sums (db2) on the matrix cores
      acc_to_tile_b3<NOUTB>(g, T1o, 0, lane);
      if (!HAS_LN) acc_to_tile_b3<NBH>(sact, T0s, 0, lane);
      wave_sync();
      tile_colsum_b3<NV_O>(db2, T1o, 0, lane);
      outer_accum_b3<NOUTB, NBH>(dW2, T1o, 0, T0s, 0, lane);
    } else if constexpr (B3) {
      acc_to_tile<NOUTB>(g, T1, ldt1, lane);
      if (!HAS_LN) acc_to_tile_b3<NBH>(sact, T0s, 0, lane);
      wave_sync();
      tile_colsum_all<NV_O>(db2, T1, ldt1, 0, lane);
      wave_sync();
      acc_to_tile_b3<NOUTB>(g, T1o, 0, lane);   // GZ as bf16 planes over the fp32 copy
      wave_sync();
      outer_accum_b3<NOUTB, NBH>(dW2, T1o, 0, T0s, 0, lane);
    } else {
      acc_to_tile<NOUTB>(g, T1, ldt1, lane);
      if (!HAS_LN) acc_to_tile<NBH>(sact, T0, ldt0, lane);
      wave_sync();
      tile_colsum_all<NV_O>(db2, T1, ldt1, 0, lane);
      outer_accum<NOUTB, NBH>(dW2, T1, ldt1, 0, T0, ldt0, 0, lane);
    }
    MSTAMP(3)   // GZ / S planes, db2, dW2 outer product
    // ga = (W2^T gz) * silu'(h)   (registers + weights only)
    f32x16 ga[NBH];
#pragma unroll
    for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[nb][r] = 0.f;
    if constexpr (B3) gemm_acc_wt_b3<NBH, NOUTB>(ga, W2im, 0, g, lane);
    else gemm_acc_wt<NBH, NOUTB>(ga, W2s, ldw2, 0, g, lane);
#pragma unroll
    for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ga[nb][r] *= nlam_silu_grad(hpre[nb][r]);
    wave_sync();
    acc_to_tile<NBH>(ga, T1, ldt1, lane);           // GA
    MSTAMP(4)   // W2^T gz, silu', GA tile
    // XP: next tile's rows fly from here on.  (One phase earlier is SLOWER, 32.5 k -> 36.6 k
    // cycles per tile: vmcnt retires in order, so the waits of the later phases -- X again,
    // residual rows -- then also wait for these rows.)
    issue_next(tt + (int64_t)gridDim.x * 4);
    if constexpr (DEFER_DW1) {
      wave_sync();
      float* gb = q.ga_out + (b * p.rows + r0) * HID;
      auto gp = [&](int t) { return gb + (int64_t)t * HID; };
      store_rows<true>(T1, ldt1, 0, HID, nrows, lane, gp);
    } else if constexpr (XP) {
      wave_sync();
      tile_colsum_all<NV_H>(db1, T1, ldt1, 0, lane);
      wave_sync();
      acc_to_tile_b3<NBH>(ga, T1h, 0, lane);        // GA planes over its fp32 copy
      wave_sync();
      outer_accum_b3<NBH, KB>(dW1, T1h, 0, T2x, 0, lane);   // X planes are still in T2
    } else {
      stage_x(b, r0, nrows);                        // X again
      wave_sync();
      tile_colsum_all<NV_H>(db1, T1, ldt1, 0, lane);
      if constexpr (B3) {
        // X (fp32 tile) -> registers -> planes in place; GA planes over its fp32 copy
        static_assert(KB <= 2, "B3 in-kernel dW1 supports k_in <= 64");
        f32x16 xin[KB];
        tile_to_acc<KB>(xin, T0, ldt0, lane);
        wave_sync();
        acc_to_tile_b3<KB>(xin, T0x, 0, lane);
        acc_to_tile_b3<NBH>(ga, T1h, 0, lane);
        wave_sync();
        outer_accum_b3<NBH, KB>(dW1, T1h, 0, T0x, 0, lane);
      } else {
        outer_accum<NBH, KB>(dW1, T1, ldt1, 0, T0, ldt0, 0, lane);
      }
    }
    MSTAMP(5)   // X again + db1 + dW1 outer product (or the ga store)
    const bool want_gx = q.gxa != nullptr || q.gxb != nullptr;
    if (want_gx) {
      // gx = W1^T ga, 64 input columns (2 blocks) at a time to bound the registers
      wave_sync();
      constexpr int GXB = KB >= 2 ? 2 : 1;
#pragma unroll
      for (int kb0 = 0; kb0 < KB; kb0 += GXB) {
        f32x16 gx[GXB];
#pragma unroll
        for (int kb = 0; kb < GXB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) gx[kb][r] = 0.f;
        if constexpr (B3) gemm_acc_wt_b3<GXB, NBH>(gx, W1im, kb0, ga, lane);
        else gemm_acc_wt<GXB, NBH>(gx, W1s, ldw1, 32 * kb0, ga, lane);
        acc_to_tile<GXB>(gx, T0 + 32 * kb0, ldt0, lane);
      }
      wave_sync();
      const float* gyb = q.gy.ptr + b * q.gy.bstride + r0 * q.gy.ld;
      auto gyp = [&](int t) { return gyb + (int64_t)t * q.gy.ld; };
      if (q.gxa != nullptr) {
        float* ob = q.gxa + b * q.gxa_bstride + r0 * q.gxa_ld;
        auto op = [&](int t) { return ob + (int64_t)t * q.gxa_ld; };
        const int w = p.src[0].width;
        if (q.add_gy_to_gxa) {
          if (q.vec_gxa && q.vec_gy)
            store_rows_res<true>(T0, ldt0, 0, w, nrows, lane, op, gyp);
          else
            store_rows_res<false>(T0, ldt0, 0, w, nrows, lane, op, gyp);
        } else {
          if (q.vec_gxa)
            store_rows<true>(T0, ldt0, 0, w, nrows, lane, op);
          else
            store_rows<false>(T0, ldt0, 0, w, nrows, lane, op);
        }
      }
      if (q.gxb != nullptr) {
        float* ob = q.gxb + b * q.gxb_bstride + r0 * q.gxb_ld;
        auto op = [&](int t) { return ob + (int64_t)t * q.gxb_ld; };
        if (q.vec_gxb)
          store_rows<true>(T0, ldt0, p.src[0].width, p.src[1].width, nrows, lane, op);
        else
          store_rows<false>(T0, ldt0, p.src[0].width, p.src[1].width, nrows, lane, op);
      }
      wave_sync();
    }
    MSTAMP(6)   // gx = W1^T ga + stores
  }
  if (q.stamp && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_mlp_bwd_stamps[k], mst[k]);
  }

  // ---- fold the waves' partials in LDS (fixed order) and write the slab
  __syncthreads();
  float* img = smem;  // weights are dead: reuse the front of LDS
  float* slab = q.slab + (int64_t)blockIdx.x * q.slab_stride;
  constexpr int n1 = HID * KP32, n2 = 32 * NOUTB * HID;
  if constexpr (!DEFER_DW1) {
    fold_blocks_to_slab<NBH, KBA>(dW1, img, KP32, slab, tid, wave, lane);
    fold_vec_lds<NV_H>(db1, img, wave, lane);
    for (int i = tid; i < HID; i += 256) slab[n1 + i] = img[i];
    __syncthreads();
  }
  fold_blocks_to_slab<NOUTB, NBH>(dW2, img, HID, slab + n1 + HID, tid, wave, lane);
  float* vbase = slab + n1 + HID + n2;
  fold_vec_lds<NV_O>(db2, img, wave, lane);
  for (int i = tid; i < 32 * NOUTB; i += 256) vbase[i] = img[i];
  __syncthreads();
  fold_vec_lds<NV_O>(dgam, img, wave, lane);
  for (int i = tid; i < 32 * NOUTB; i += 256) vbase[32 * NOUTB + i] = img[i];
  __syncthreads();
  fold_vec_lds<NV_O>(dbet, img, wave, lane);
  for (int i = tid; i < 32 * NOUTB; i += 256) vbase[2 * 32 * NOUTB + i] = img[i];
}

extern "C" int64_t nlam_bwd_grid(int64_t ntiles) {
  int64_t g = (ntiles + 3) / 4;
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return g;
}

// out[i] (+)= sum_s slab[s * stride + i], i < n: a workgroup owns 32 consecutive
// elements and splits the slabs over its 32 sub-groups (1024 threads keep enough
// loads in flight for these latency-bound 10-30 MB reductions); partial sums are
// combined in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ slab,
                                                            int64_t nslabs, int64_t stride,
                                                            int64_t n, float* __restrict__ out,
                                                            int accumulate) {
  __shared__ float red[32][33];
  const int e = threadIdx.x & 31, gsub = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + e;
  float s = 0.f;
  if (i < n) {
    for (int64_t k = gsub; k < nslabs; k += 32 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t ku = k + 32 * u;
        v[u] = slab[(ku < nslabs ? ku : k) * stride + i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k + 32 * u < nslabs) s += v[u];
    }
  }
  red[gsub][e] = s;
  __syncthreads();
  if (gsub == 0 && i < n) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) v += red[g][e];
    if (accumulate) v += out[i];
    out[i] = v;
  }
}
extern "C" int nlam_reduce_slabs(const float* slab, int64_t nslabs, int64_t stride, int64_t n,
                                 float* out, int accumulate, void* stream) {
  if (n <= 0) return 0;
  reduce_slabs_kernel<<<(unsigned)((n + 31) / 32), 1024, 0, (hipStream_t)stream>>>(
      slab, nslabs, stride, n, out, accumulate);
  NLAM_CHECK_LAUNCH("reduce_slabs");
  return 0;
}

template <int HID, int NOUTB, int KB, bool HAS_LN, bool DEFER_DW1 = false, bool B3 = false,
          bool XP = false>
static int launch_mlp_bwd(const MlpBwdParams& q, hipStream_t s) {
  constexpr int KP32 = 32 * KB;
  constexpr int ldt0 = (KP32 > HID ? KP32 : HID) + 4;
  constexpr int ldt1 = (HID > 32 * NOUTB ? HID : 32 * NOUTB) + 4;
  const size_t lds = ((size_t)HID * (KP32 + 4) + (size_t)32 * NOUTB * (HID + 4) + HID +
                      2 * 32 * NOUTB + (size_t)4 * NLAM_TILE * (ldt0 + ldt1) +
                      (XP ? (size_t)4 * NLAM_TILE * (KP32 + 4) : 0)) * sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "mlp_bwd: LDS footprint %zu B exceeds 160 KiB", lds);
  static_assert((size_t)4 * HID * KP32 * 4 <= 160 * 1024, "fold images exceed LDS");
  const size_t fold_bytes = (size_t)4 * HID * KP32 * sizeof(float);
  const size_t lds_alloc = lds > fold_bytes ? lds : fold_bytes;
  auto kern = mlp_bwd_kernel<HID, NOUTB, KB, HAS_LN, DEFER_DW1, B3, XP>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles = ((q.f.rows + NLAM_TILE - 1) / NLAM_TILE) * q.f.B;
  kern<<<(unsigned)nlam_bwd_grid(ntiles), 256, lds_alloc, s>>>(q);
  NLAM_CHECK_LAUNCH("mlp_bwd_kernel");
  return 0;
}

/* slab floats per workgroup for nlam_mlp_bwd */
extern "C" int64_t nlam_mlp_bwd_slab_stride(int k_in, int hid, int n_out) {
  const int kp32 = (k_in + 31) & ~31, no32 = (n_out + 31) & ~31;
  return (int64_t)hid * kp32 + hid + (int64_t)no32 * hid + 3 * no32;
}

extern "C" int nlam_mlp_bwd(
    const float* xa, int64_t xa_bstride, int64_t xa_ld, int xa_width,
    const float* xb, int64_t xb_bstride, int64_t xb_ld, int xb_width,
    const float* W1, int64_t ldW1, const float* b1, const float* W2, int64_t ldW2,
    const float* b2, const float* gamma,
    const float* gy, int64_t gy_bstride, int64_t gy_ld,
    float* gxa, int64_t gxa_bstride, int64_t gxa_ld,
    float* gxb, int64_t gxb_bstride, int64_t gxb_ld, int add_gy_to_gxa,
    float* slab, int64_t slab_stride, float* ga_out,
    int64_t B, int64_t rows, int hid, int n_out, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(hid == 64 || hid == 128, "nlam_mlp_bwd: hidden width %d not in {64,128}", hid);
  NLAM_REQUIRE(n_out >= 1 && n_out <= hid, "nlam_mlp_bwd: n_out out of range");
  NLAM_REQUIRE(gamma == nullptr || n_out == hid, "nlam_mlp_bwd: LayerNorm needs n_out == hid");
  NLAM_REQUIRE(slab != nullptr && slab_stride >= nlam_mlp_bwd_slab_stride(
                   xa_width + (xb ? xb_width : 0), hid, n_out), "nlam_mlp_bwd: slab too small");
  NLAM_REQUIRE(!add_gy_to_gxa || xa_width == n_out, "nlam_mlp_bwd: residual width mismatch");
  MlpBwdParams q;
  MlpParams& p = q.f;
  p.src[0] = RowView{xa, xa_bstride, xa_ld, xa_width};
  p.src[1] = RowView{xb, xb_bstride, xb_ld, xb ? xb_width : 0};
  p.nsrc = xb ? 2 : 1;
  p.k_in = xa_width + (xb ? xb_width : 0);
  p.k_pad = (p.k_in + 7) & ~7;
  p.n_out = n_out;
  p.W1 = W1; p.ldW1 = ldW1; p.b1 = b1; p.W2 = W2; p.ldW2 = ldW2; p.b2 = b2;
  p.gamma = gamma; p.beta = nullptr;
  p.res = nullptr; p.res_bstride = 0; p.res_ld = 0; p.out = nullptr; p.out_bstride = 0; p.out_ld = 0;
  p.rows = rows; p.B = (int)B;
  p.vec_mask = 0;
  if (view_vec_ok(xa, xa_bstride, xa_ld, xa_width)) p.vec_mask |= 1;
  if (xb && view_vec_ok(xb, xb_bstride, xb_ld, xb_width) && (xa_width % 4 == 0)) p.vec_mask |= 2;
  q.gy = RowView{gy, gy_bstride, gy_ld, n_out};
  q.gxa = gxa; q.gxa_bstride = gxa_bstride; q.gxa_ld = gxa_ld;
  q.gxb = gxb; q.gxb_bstride = gxb_bstride; q.gxb_ld = gxb_ld;
  q.add_gy_to_gxa = add_gy_to_gxa;
  q.slab = slab; q.slab_stride = slab_stride;
  q.ga_out = ga_out;
  NLAM_REQUIRE(ga_out == nullptr || nlam_aligned16(ga_out), "nlam_mlp_bwd: ga_out misaligned");
  q.vec_gy = view_vec_ok(gy, gy_bstride, gy_ld, n_out);
  q.vec_gxa = gxa && view_vec_ok(gxa, gxa_bstride, gxa_ld, xa_width);
  q.vec_gxb = gxb && view_vec_ok(gxb, gxb_bstride, gxb_ld, xb_width) && (xa_width % 4 == 0);
  static const bool mstamp = getenv("NLAM_STAMP") != nullptr;
  q.stamp = mstamp ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (hid == 64) {   // 16-row, two-waves-per-SIMD form (fused16_mlp.hip) where it applies
    const int r16 = nlam_k16_mlp_bwd(q, s);
    if (r16 >= 0) return r16;
  }
  const bool ln = gamma != nullptr;
  const int kb = (p.k_in + 31) / 32;
  const int noutb = (n_out + 31) / 32;
#define MLP_BWD_CASE(H, NO, K, L) return launch_mlp_bwd<H, NO, K, L>(q, s)
  if (hid == 64 && nlam_mfma_b3()) {
    // float4 views of x and gy: X staged once as planes + next-tile register prefetch
    const bool xp = (p.vec_mask & 1) && p.nsrc == 1 && q.vec_gy && n_out == 32 * noutb;
    if (xp && ln && kb == 1) return launch_mlp_bwd<64, 2, 1, true, false, true, true>(q, s);
    if (xp && ln && kb == 2) return launch_mlp_bwd<64, 2, 2, true, false, true, true>(q, s);
    if (ln && kb == 1) return launch_mlp_bwd<64, 2, 1, true, false, true>(q, s);
    if (ln && kb == 2) return launch_mlp_bwd<64, 2, 2, true, false, true>(q, s);
    if (ln && kb == 4 && ga_out != nullptr) return launch_mlp_bwd<64, 2, 4, true, true, true>(q, s);
    // the 17-wide output map: X planes + prefetch as above, gy by dword prefetch
    if (!ln && noutb == 1 && kb == 2 && (p.vec_mask & 1) && p.nsrc == 1 && n_out <= 24)
      return launch_mlp_bwd<64, 1, 2, false, false, true, true>(q, s);
    if (!ln && noutb == 1 && kb == 2) return launch_mlp_bwd<64, 1, 2, false, false, true>(q, s);
  }
  if (hid == 64) {
    if (ln) {
      if (kb == 1) MLP_BWD_CASE(64, 2, 1, true);
      if (kb == 2) MLP_BWD_CASE(64, 2, 2, true);
      if (kb == 4 && ga_out != nullptr) return launch_mlp_bwd<64, 2, 4, true, true>(q, s);
      if (kb == 4) MLP_BWD_CASE(64, 2, 4, true);
    } else if (noutb == 1 && kb == 2) {
      MLP_BWD_CASE(64, 1, 2, false);
    }
  }
#undef MLP_BWD_CASE
  nlam_set_error("nlam_mlp_bwd: unsupported shape hid=%d k_in=%d n_out=%d ln=%d", hid, p.k_in,
                 n_out, (int)ln);
  return 1;
}

// ---------------------------------------------------- projection backward
// y = x [WA; WB]^T + [bA; bB]:  gx = gy [WA; WB],  dW = gy^T x,  db = colsum(gy).
// Slab per workgroup: [dW (32 NOUTB x KP32) | db (32 NOUTB)].

// B3: split-bf16 MFMAs (needs the float4 views and k_in == 32 KB, n_out == 32 NOUTB).
template <int NOUTB, int KB, bool SUMGY = false, bool B3 = false>
__global__ __launch_bounds__(256) void lin_bwd_kernel(LinBwdParams q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KP32 = 32 * KB, NO = 32 * NOUTB;
  constexpr int NV = (NO + 63) / 64;
  constexpr int ldw = KP32 + 4, ldt0 = KP32 + 4, ldt1 = NO + 4;
  constexpr int WSTRIDE = NLAM_TILE * (ldt0 + ldt1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Ws = smem;
  float* T0base = Ws + NO * ldw;
  float* T1base = T0base + NLAM_TILE * ldt0;
  float* T0 = T0base + wave * WSTRIDE;
  float* T1 = T1base + wave * WSTRIDE;
  const B3Image Wim = b3_image(Ws, NO, KP32);      // same bytes as the fp32 image
  if (B3) {
    load_weight_lds_b3(Wim, 0, q.WA, q.ldWA, q.nA, q.x.width, q.nA, KP32, tid, 256);
    if (q.nB > 0)
      load_weight_lds_b3(Wim, q.nA, q.WB, q.ldWB, q.nB, q.x.width, NO - q.nA, KP32, tid, 256);
  } else {
    load_weight_lds(Ws, q.WA, q.ldWA, q.nA, q.x.width, q.nA, KP32, tid, 256);
    if (q.nB > 0)
      load_weight_lds(Ws + q.nA * ldw, q.WB, q.ldWB, q.nB, q.x.width, NO - q.nA, KP32, tid, 256);
  }
  __syncthreads();
  const B3Tile T0p = b3_tile(T0, KP32), T1p = b3_tile(T1, NO);
  f32x16 dW[NOUTB][KB];
#pragma unroll
  for (int i = 0; i < NOUTB; ++i)
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW[i][j][r] = 0.f;
  float db[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) db[j] = 0.f;

  const int n_out = q.nA + q.nB;
  const int64_t tiles_per_b = (q.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * q.B;
  // B3 without the batch sum: the x / gy rows of the NEXT tile are requested as soon as the
  // current ones have been written to LDS (same registers), and land under this tile's MFMAs
  constexpr bool PF = B3 && !SUMGY;
  f32x4 pvx[PF ? 4 * KB : 1], pvg[PF ? 4 * NOUTB : 1];
  auto issue_next = [&](int64_t task) {
    if constexpr (PF) {
      const int64_t tq = task < ntiles ? task : ntiles - 1;
      const int64_t bq = tq / tiles_per_b;
      const int64_t rq = (tq - bq * tiles_per_b) * NLAM_TILE;
      const int nq = (int)((q.rows - rq) < NLAM_TILE ? (q.rows - rq) : NLAM_TILE);
      view_load_v<4 * KB>(pvx, q.x, bq, rq, nq, lane);
      view_load_v<4 * NOUTB>(pvg, q.gy, bq, rq, nq, lane);
    }
  };
  const bool pf = PF && q.vec_x && q.vec_gy;
  if (pf && (int64_t)blockIdx.x * 4 + wave < ntiles) issue_next((int64_t)blockIdx.x * 4 + wave);
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < ntiles; tt += (int64_t)gridDim.x * 4) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nfull = (int)((q.rows - r0) < NLAM_TILE ? (q.rows - r0) : NLAM_TILE);
    const int nrows = nfull;
    if (pf) {
      if constexpr (PF) {
        put_rows_v_b3<4 * KB>(T0p, 0, q.x.width, nrows, lane, pvx);
        put_rows_v_b3<4 * NOUTB>(T1p, 0, n_out, nrows, lane, pvg);
        issue_next(tt + (int64_t)gridDim.x * 4);
      }
    } else if (q.vec_x && q.vec_gy) {
      f32x4 vx[4 * KB], vg[4 * NOUTB];
      view_load_v<4 * KB>(vx, q.x, b, r0, nfull, lane);
      view_load_v<4 * NOUTB>(vg, q.gy, b, r0, nfull, lane);
      // batch-invariant x: the per-sample gradients are summed while loading (fixed
      // order s = 0, 1, 2, ...; FL slices in flight, bounded by the register budget)
      if (SUMGY) {
        constexpr int FL = NOUTB <= 2 ? 3 : 1;
        const int last = q.gy_nsum - 1;
        for (int s0 = 1; s0 <= last; s0 += FL) {
          f32x4 tv[FL][4 * NOUTB];
#pragma unroll
          for (int f = 0; f < FL; ++f) {
            const int sf = s0 + f < last ? s0 + f : last;
            RowView gv = q.gy;
            gv.ptr = q.gy.ptr + sf * q.gy_sum_stride;
            view_load_v<4 * NOUTB>(tv[f], gv, b, r0, nfull, lane);
          }
#pragma unroll
          for (int f = 0; f < FL; ++f) {
            if (s0 + f <= last) {
#pragma unroll
              for (int k = 0; k < 4 * NOUTB; ++k) vg[k] += tv[f][k];
            }
          }
        }
      }
      if (B3) {
        put_rows_v_b3<4 * KB>(T0p, 0, q.x.width, nrows, lane, vx);
        put_rows_v_b3<4 * NOUTB>(T1p, 0, n_out, nrows, lane, vg);
      } else {
        put_rows_v<4 * KB, false>(T0, ldt0, 0, q.x.width, nrows, lane, vx);
        put_rows_v<4 * NOUTB, false>(T1, ldt1, 0, n_out, nrows, lane, vg);
      }
    } else {
      view_stage_s(T0, ldt0, 0, q.x, b, r0, nrows, lane);
      view_stage_s(T1, ldt1, 0, q.gy, b, r0, nrows, lane);
    }
    if (KP32 > q.x.width) zero_cols(T0, ldt0, q.x.width, KP32 - q.x.width, lane);
    if (NO > n_out) zero_cols(T1, ldt1, n_out, NO - n_out, lane);
    wave_sync();
    f32x16 gx[KB];
    if constexpr (B3) {
      // x and gy were staged as bf16 planes: column sums, dW and gx all read them directly
      tile_colsum_b3<NV>(db, T1p, 0, lane);
      outer_accum_b3<NOUTB, KB>(dW, T1p, 0, T0p, 0, lane);
      if (q.gx != nullptr) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) gx[kb][r] = 0.f;
        gemm_tile_wt_b3<KB, NOUTB>(gx, Wim, 0, T1p, 0, lane);
      }
    } else {
      tile_colsum_all<NV>(db, T1, ldt1, 0, lane);
      outer_accum<NOUTB, KB>(dW, T1, ldt1, 0, T0, ldt0, 0, lane);
      if (q.gx != nullptr) {
        f32x16 g[NOUTB];
        tile_to_acc<NOUTB>(g, T1, ldt1, lane);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) gx[kb][r] = 0.f;
        gemm_acc_wt<KB, NOUTB>(gx, Ws, ldw, 0, g, lane);
      }
    }
    if (q.gx != nullptr) {
      wave_sync();
      acc_to_tile<KB>(gx, T0, ldt0, lane);
      wave_sync();
      float* ob = q.gx + b * q.gx_bstride + r0 * q.gx_ld;
      auto op = [&](int t) { return ob + (int64_t)t * q.gx_ld; };
      if (q.gx_add != nullptr) {
        const float* ab = q.gx_add + b * q.ga_bstride + r0 * q.ga_ld;
        auto ap = [&](int t) { return ab + (int64_t)t * q.ga_ld; };
        if (q.vec_gx)
          store_rows_res<true>(T0, ldt0, 0, q.x.width, nrows, lane, op, ap);
        else
          store_rows_res<false>(T0, ldt0, 0, q.x.width, nrows, lane, op, ap);
      } else if (q.vec_gx) {
        store_rows<true>(T0, ldt0, 0, q.x.width, nrows, lane, op);
      } else {
        store_rows<false>(T0, ldt0, 0, q.x.width, nrows, lane, op);
      }
      wave_sync();
    }
  }
  __syncthreads();
  float* img = smem;
  float* slab = q.slab + (int64_t)blockIdx.x * q.slab_stride;
  fold_blocks_to_slab<NOUTB, KB>(dW, img, KP32, slab, tid, wave, lane);
  fold_vec_lds<NV>(db, img, wave, lane);
  for (int i = tid; i < NO; i += 256) slab[NO * KP32 + i] = img[i];
}

template <int NOUTB, int KB, bool SUMGY = false, bool B3 = false>
static int launch_lin_bwd(const LinBwdParams& q, hipStream_t s) {
  constexpr int KP32 = 32 * KB, NO = 32 * NOUTB;
  const size_t lds = ((size_t)NO * (KP32 + 4) + (size_t)4 * NLAM_TILE * (KP32 + 4 + NO + 4)) *
                     sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "lin_bwd: LDS footprint %zu B exceeds 160 KiB", lds);
  const size_t fold_bytes = (size_t)4 * NO * KP32 * sizeof(float);
  NLAM_REQUIRE(fold_bytes <= 160 * 1024, "lin_bwd: fold images exceed LDS");
  const size_t lds_alloc = lds > fold_bytes ? lds : fold_bytes;
  auto kern = lin_bwd_kernel<NOUTB, KB, SUMGY, B3>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles = ((q.rows + NLAM_TILE - 1) / NLAM_TILE) * q.B;
  kern<<<(unsigned)nlam_bwd_grid(ntiles), 256, lds_alloc, s>>>(q);
  NLAM_CHECK_LAUNCH("lin_bwd_kernel");
  return 0;
}

extern "C" int64_t nlam_lin_bwd_slab_stride(int k_in, int n_out) {
  const int kp32 = (k_in + 31) & ~31;
  return (int64_t)n_out * kp32 + n_out;
}

extern "C" int nlam_lin_bwd(const float* x, int64_t x_bstride, int64_t x_ld, int k_in,
                            const float* gy, int64_t gy_bstride, int64_t gy_ld,
                            const float* WA, int64_t ldWA, int nA,
                            const float* WB, int64_t ldWB, int nB,
                            float* gx, int64_t gx_bstride, int64_t gx_ld,
                            const float* gx_add, int64_t ga_bstride, int64_t ga_ld,
                            int64_t gy_nsum, int64_t gy_sum_stride,
                            float* slab, int64_t slab_stride, int64_t B, int64_t rows,
                            void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(nA > 0 && nA % 32 == 0 && nB >= 0 && nB % 32 == 0,
               "nlam_lin_bwd: output block widths must be multiples of 32");
  NLAM_REQUIRE(gx_add == nullptr || gx != nullptr, "nlam_lin_bwd: gx_add without gx");
  LinBwdParams q;
  q.x = RowView{x, x_bstride, x_ld, k_in};
  q.nA = nA; q.nB = WB ? nB : 0;
  q.gy = RowView{gy, gy_bstride, gy_ld, q.nA + q.nB};
  q.WA = WA; q.ldWA = ldWA; q.WB = WB; q.ldWB = ldWB;
  q.gx = gx; q.gx_bstride = gx_bstride; q.gx_ld = gx_ld;
  q.gx_add = gx_add; q.ga_bstride = ga_bstride; q.ga_ld = ga_ld;
  q.slab = slab; q.slab_stride = slab_stride;
  q.rows = rows; q.B = (int)B;
  NLAM_REQUIRE(slab != nullptr && slab_stride >= nlam_lin_bwd_slab_stride(k_in, q.nA + q.nB),
               "nlam_lin_bwd: slab too small");
  q.vec_x = view_vec_ok(x, x_bstride, x_ld, k_in);
  q.vec_gy = view_vec_ok(gy, gy_bstride, gy_ld, q.nA + q.nB);
  q.vec_gx = gx && view_vec_ok(gx, gx_bstride, gx_ld, k_in) &&
             (gx_add == nullptr || view_vec_ok(gx_add, ga_bstride, ga_ld, k_in));
  q.gy_nsum = gy_nsum > 1 ? (int)gy_nsum : 1;
  q.gy_sum_stride = gy_sum_stride;
  q.gh = nullptr; q.gh_bstride = 0; q.csc_colptr = nullptr; q.csc_eid = nullptr; q.n_send = 0;
  NLAM_REQUIRE(q.gy_nsum == 1 || (q.vec_x && q.vec_gy && gy_sum_stride % 4 == 0),
               "nlam_lin_bwd: gy_nsum > 1 needs 16-byte aligned x / gy rows and slices");
  hipStream_t s = (hipStream_t)stream;
  {
    const int r16 = nlam_k16_lin_bwd(q, s);
    if (r16 >= 0) return r16;
  }
  const int noutb = (q.nA + q.nB) / 32, kb = (k_in + 31) / 32;
  const bool b3 = nlam_mfma_b3() && q.vec_x && q.vec_gy && k_in == 32 * kb &&
                  (q.nA + q.nB) == 32 * noutb;
  if (b3) {
    if (q.gy_nsum > 1 && noutb == 2 && kb == 2) return launch_lin_bwd<2, 2, true, true>(q, s);
    if (q.gy_nsum <= 1 && noutb == 2 && kb == 2) return launch_lin_bwd<2, 2, false, true>(q, s);
    if (q.gy_nsum <= 1 && noutb == 4 && kb == 2) return launch_lin_bwd<4, 2, false, true>(q, s);
  }
  if (q.gy_nsum > 1) {
    if (noutb == 2 && kb == 2) return launch_lin_bwd<2, 2, true>(q, s);
    nlam_set_error("nlam_lin_bwd: gy_nsum > 1 unsupported for k_in=%d n_out=%d", k_in,
                   q.nA + q.nB);
    return 1;
  }
  if (noutb == 2 && kb == 2) return launch_lin_bwd<2, 2>(q, s);
  if (noutb == 4 && kb == 2) return launch_lin_bwd<4, 2>(q, s);
  if (noutb == 2 && kb == 1) return launch_lin_bwd<2, 1>(q, s);
  if (noutb == 4 && kb == 4) return launch_lin_bwd<4, 4>(q, s);
  nlam_set_error("nlam_lin_bwd: unsupported shape k_in=%d n_out=%d", k_in, q.nA + q.nB);
  return 1;
}


// ------------------------------------------------ deferred weight gradients
// dW (32 NGB x 32 NXB) = sum_rows G[r]^T (x) [xa | xb][r],  db = colsum(G).
// A lean streaming pass: no weights in LDS, every register for the accumulators.
// Slab per workgroup: [dW (32 NGB x 32 NXB) | db (32 NGB)].

template <int NGB, int NXB, bool B3 = false>
__global__ __launch_bounds__(256) void outer_bwd_kernel(OuterParams q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NG = 32 * NGB, NX = 32 * NXB;
  constexpr int NV = (NG + 63) / 64;
  constexpr int ldg = NG + 4, ldx = NX + 4;
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
  const int kx = q.xa.width + (q.xb.ptr ? q.xb.width : 0);
  const int64_t tiles_per_b = (q.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * q.B;
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < ntiles; tt += (int64_t)gridDim.x * 4) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((q.rows - r0) < NLAM_TILE ? (q.rows - r0) : NLAM_TILE);
    const int t = lane & 31;
    int xi = 0;
    if (q.x_index) xi = q.x_index[r0 + (t < nrows ? t : nrows - 1)];
    const float* gb = q.g.ptr + b * q.g.bstride + r0 * q.g.ld;
    const float* xab = q.xa.ptr + b * q.xa.bstride + (q.x_index ? 0 : r0 * q.xa.ld);
    const float* xbb = q.xb.ptr ? q.xb.ptr + b * q.xb.bstride + (q.x_index ? 0 : r0 * q.xb.ld)
                                : nullptr;
    const int last = nrows - 1;
    auto g_row = [&](int s) { return gb + (int64_t)(s < last ? s : last) * q.g.ld; };
    auto xa_row = [&](int s) {
      const int64_t r = q.x_index ? (int64_t)__shfl(xi, s, 64) : (int64_t)(s < last ? s : last);
      return xab + r * q.xa.ld;
    };
    auto xb_row = [&](int s) {
      const int64_t r = q.x_index ? (int64_t)__shfl(xi, s, 64) : (int64_t)(s < last ? s : last);
      return xbb + r * q.xb.ld;
    };
    f32x4 vg[4 * NGB], va[8], vb[8];
    load_rows_v<4 * NGB>(vg, NG, lane, g_row);
    load_rows_v<8>(va, q.xa.width, lane, xa_row);
    if (xbb) load_rows_v<8>(vb, q.xb.width, lane, xb_row);
    if constexpr (B3) {
      // G and X straight to bf16 planes; the column sums of G run on the matrix cores
      const B3Tile TGp = b3_tile(TG, NG), TXp = b3_tile(TX, NX);
      put_rows_v_b3<4 * NGB>(TGp, 0, NG, nrows, lane, vg);
      put_rows_v_b3<8>(TXp, 0, q.xa.width, nrows, lane, va);
      if (xbb) put_rows_v_b3<8>(TXp, q.xa.width, q.xb.width, nrows, lane, vb);
      wave_sync();
      tile_colsum_b3<NV>(db, TGp, 0, lane);
      outer_accum_b3<NGB, NXB>(dW, TGp, 0, TXp, 0, lane);
    } else {
      put_rows_v<4 * NGB, false>(TG, ldg, 0, NG, nrows, lane, vg);
      put_rows_v<8, false>(TX, ldx, 0, q.xa.width, nrows, lane, va);
      if (xbb) put_rows_v<8, false>(TX, ldx, q.xa.width, q.xb.width, nrows, lane, vb);
      if (NX > kx) zero_cols(TX, ldx, kx, NX - kx, lane);
      wave_sync();
      tile_colsum_all<NV>(db, TG, ldg, 0, lane);
      outer_accum<NGB, NXB>(dW, TG, ldg, 0, TX, ldx, 0, lane);
    }
    wave_sync();
  }
  __syncthreads();
  float* slab = q.slab + (int64_t)blockIdx.x * q.slab_stride;
  fold_blocks_to_slab<NGB, NXB>(dW, smem, NX, slab, tid, wave, lane);
  fold_vec_lds<NV>(db, smem, wave, lane);
  for (int i = tid; i < NG; i += 256) slab[NG * NX + i] = smem[i];
}

template <int NGB, int NXB, bool B3 = false>
static int launch_outer_bwd(const OuterParams& q, hipStream_t s) {
  constexpr int NG = 32 * NGB, NX = 32 * NXB;
  size_t lds = (size_t)4 * NLAM_TILE * (NG + 4 + NX + 4) * sizeof(float);
  const size_t fold = (size_t)4 * NG * NX * sizeof(float);
  if (fold > lds) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "outer_bwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = outer_bwd_kernel<NGB, NXB, B3>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles = ((q.rows + NLAM_TILE - 1) / NLAM_TILE) * q.B;
  kern<<<(unsigned)nlam_bwd_grid(ntiles), 256, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("outer_bwd_kernel");
  return 0;
}

extern "C" int64_t nlam_outer_bwd_slab_stride(int ng, int kx) {
  const int nx = (kx + 31) & ~31;
  return (int64_t)ng * nx + ng;
}

extern "C" int nlam_outer_bwd(const float* g, int64_t g_bstride, int64_t g_ld, int ng,
                              const float* xa, int64_t xa_bstride, int64_t xa_ld, int xa_width,
                              const float* xb, int64_t xb_bstride, int64_t xb_ld, int xb_width,
                              const int32_t* x_index, float* slab, int64_t slab_stride,
                              int64_t B, int64_t rows, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
  NLAM_REQUIRE(ng == 64, "nlam_outer_bwd: G width %d unsupported (64)", ng);
  const int kx = xa_width + (xb ? xb_width : 0);
  NLAM_REQUIRE(view_vec_ok(g, g_bstride, g_ld, ng) && view_vec_ok(xa, xa_bstride, xa_ld, xa_width) &&
                   xa_width <= 64 && (!xb || (view_vec_ok(xb, xb_bstride, xb_ld, xb_width) &&
                                              xb_width <= 64 && xa_width % 4 == 0)),
               "nlam_outer_bwd: operands must be 16-byte aligned rows of width <= 64, %% 4 == 0");
  NLAM_REQUIRE(slab != nullptr && slab_stride >= nlam_outer_bwd_slab_stride(ng, kx),
               "nlam_outer_bwd: slab too small");
  OuterParams q;
  q.g = RowView{g, g_bstride, g_ld, ng};
  q.xa = RowView{xa, xa_bstride, xa_ld, xa_width};
  q.xb = RowView{xb, xb_bstride, xb_ld, xb ? xb_width : 0};
  q.x_index = x_index; q.slab = slab; q.slab_stride = slab_stride; q.rows = rows; q.B = (int)B;
  hipStream_t s = (hipStream_t)stream;
  {
    const int r16 = nlam_k16_outer_bwd(q, s);
    if (r16 >= 0) return r16;
  }
  const int nxb = (kx + 31) / 32;
  if (nlam_mfma_b3() && kx == 32 * nxb) {
    if (nxb == 2) return launch_outer_bwd<2, 2, true>(q, s);
    if (nxb == 4) return launch_outer_bwd<2, 4, true>(q, s);
  }
  if (nxb <= 2) return launch_outer_bwd<2, 2>(q, s);
  if (nxb <= 4) return launch_outer_bwd<2, 4>(q, s);
  nlam_set_error("nlam_outer_bwd: X width %d unsupported", kx);
  return 1;
}


// ------------------------------------------------------- multi-segment reduce
// One launch sums up to NLAM_MAX_SEGS matrix segments of per-workgroup slabs straight
// into their (possibly strided) destinations, e.g. the three column blocks of
// edge_mlp.0.weight's gradient that come from three different kernels:
//   dst_k[r * dst_ld_k + c] = sum_{s < nslabs_k} slab_k[s * stride_k + src_off_k + r * src_ld_k + c]
// Every segment names its own slab buffer, so ONE launch finishes all weight gradients
// of an InteractionNet layer's backward (these reductions are launch-latency bound).
// (64 segments x 44 bytes of 32-bit fields: the block travels as a kernel argument, < 4 KB)
#define NLAM_MAX_SEGS 64
struct ReduceSegs {
  int nseg;
  const float* slab[NLAM_MAX_SEGS];
  float* dst[NLAM_MAX_SEGS];
  int32_t nslabs[NLAM_MAX_SEGS];
  int32_t stride[NLAM_MAX_SEGS];
  int32_t src_off[NLAM_MAX_SEGS];
  int32_t cols[NLAM_MAX_SEGS];
  int32_t src_ld[NLAM_MAX_SEGS], dst_ld[NLAM_MAX_SEGS];
  int32_t first[NLAM_MAX_SEGS + 1];   // prefix sums of rows*cols
};
static inline bool reduce_seg_fits(int64_t nslabs, int64_t stride, int64_t src_off, int64_t src_ld,
                                   int64_t dst_ld, int64_t total) {
  const int64_t lim = 0x7fffffff;
  return nslabs <= lim && stride <= lim && src_off <= lim && src_ld <= lim && dst_ld <= lim &&
         total <= lim && stride >= 0 && src_off >= 0 && src_ld >= 0 && dst_ld >= 0;
}

// G sub-groups per workgroup split the slabs (G = 16 for the 100-256 slabs of a large layer; 4 or
// 1 when a small layer has a handful: a 256 x 256 gradient is 1,024 workgroups per matrix, and
// 16-wave workgroups of which 15 waves had no slab made the reduction of an 81-node level of
// Hi-LAM-256 take 57 us).
// V4: every segment is float4-addressable (host-checked): a lane owns FOUR consecutive outputs, a
// wavefront reads 1 KB of every slab per load instead of 256 B (slabs are 64 KB - 256 KB apart:
// with 4-byte lanes every wave-load opened a new DRAM page for 256 bytes, 2.0 TB/s on the 120 MB
// of a hidden-256 layer).
template <int G, bool V4>
__global__ __launch_bounds__(64 * G) void reduce_slabs_multi_kernel(ReduceSegs q) {
  // 64 (V4: 256) consecutive outputs per workgroup, G sub-groups that split the slabs; eight slab
  // loads in flight per thread (clamped + masked, so they issue back to back); the summation
  // order is fixed by the launch shape
  constexpr int W = V4 ? 4 : 1;
  typedef typename std::conditional<V4, f32x4, float>::type vec_t;
  __shared__ vec_t red[G][65];
  const int e = threadIdx.x & 63, gsub = threadIdx.x >> 6;
  const int64_t i = ((int64_t)blockIdx.x * 64 + e) * W;
  const int64_t n = q.first[q.nseg];
  // segment of this workgroup's first output by a wave-uniform binary search (scalar loads of the
  // kernel argument block; a lane-by-lane linear walk was up to 64 dependent loads), then the
  // few steps a lane needs when the workgroup's outputs straddle segment boundaries
  int k = 0;
  {
    const int64_t i0 = (int64_t)blockIdx.x * 64 * W;
    int lo = 0, hi = q.nseg;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)q.first[mid] <= i0) lo = mid; else hi = mid;
    }
    k = lo;
  }
  while (k + 1 < q.nseg && i >= q.first[k + 1]) ++k;
  const int64_t local = i - q.first[k];
  const int cols = q.cols[k] > 0 ? q.cols[k] : 1;
  const int64_t r = local / cols, c = local - r * cols;
  const float* __restrict__ slab = q.slab[k] + (int64_t)q.src_off[k] + r * (int64_t)q.src_ld[k] + c;
  const int64_t nslabs = q.nslabs[k], stride = q.stride[k];
  vec_t s = vec_t{};
  if (i < n) {
    for (int64_t sl = gsub; sl < nslabs; sl += G * 8) {
      vec_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t su = sl + G * u;
        v[u] = *reinterpret_cast<const vec_t*>(slab + (su < nslabs ? su : sl) * stride);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (sl + G * u < nslabs) s += v[u];
    }
  }
  float* const out = q.dst[k] + r * (int64_t)q.dst_ld[k] + c;
  if constexpr (G == 1) {
    if (i < n) *reinterpret_cast<vec_t*>(out) = s;
  } else {
    red[gsub][e] = s;
    __syncthreads();
    if (gsub == 0 && i < n) {
      vec_t v = vec_t{};
#pragma unroll
      for (int g = 0; g < G; ++g) v += red[g][e];
      *reinterpret_cast<vec_t*>(out) = v;
    }
  }
}

// every segment float4-addressable: boundaries, row widths, pitches, offsets and base pointers
static bool reduce_segs_vec4(const ReduceSegs& q) {
  for (int k = 0; k < q.nseg; ++k) {
    if ((q.first[k] | q.first[k + 1] | q.cols[k] | q.src_off[k] | q.src_ld[k] | q.dst_ld[k] | q.stride[k]) & 3)
      return false;
    if (((uintptr_t)q.slab[k] | (uintptr_t)q.dst[k]) & 15) return false;
  }
  return true;
}

static int launch_reduce_segs(ReduceSegs& q, hipStream_t s) {
  for (int k = q.nseg; k < NLAM_MAX_SEGS; ++k) {
    q.slab[k] = nullptr; q.nslabs[k] = 0; q.stride[k] = 0; q.src_off[k] = 0; q.cols[k] = 0;
    q.src_ld[k] = 0; q.dst_ld[k] = 0; q.dst[k] = nullptr; q.first[k + 1] = q.first[q.nseg];
  }
  const int64_t n = q.first[q.nseg];
  if (n <= 0) return 0;
  int maxn = 1;
  for (int k = 0; k < q.nseg; ++k) maxn = q.nslabs[k] > maxn ? q.nslabs[k] : maxn;
  // (float4 lanes need enough outputs to fill the device: at hidden 64 a layer's gradients are
  // ~30 k elements = 120 workgroups of 256 outputs, and the 4-byte form was 0.2-0.4 % faster)
  if (n >= 65536 && reduce_segs_vec4(q)) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (maxn <= 8) reduce_slabs_multi_kernel<1, true><<<grid, 64, 0, s>>>(q);
    else if (maxn <= 32) reduce_slabs_multi_kernel<4, true><<<grid, 256, 0, s>>>(q);
    else reduce_slabs_multi_kernel<16, true><<<grid, 1024, 0, s>>>(q);
  } else {
    const unsigned grid = (unsigned)((n + 63) / 64);
    if (maxn <= 8) reduce_slabs_multi_kernel<1, false><<<grid, 64, 0, s>>>(q);
    else if (maxn <= 32) reduce_slabs_multi_kernel<4, false><<<grid, 256, 0, s>>>(q);
    else reduce_slabs_multi_kernel<16, false><<<grid, 1024, 0, s>>>(q);
  }
  NLAM_CHECK_LAUNCH("reduce_slabs_multi");
  return 0;
}

extern "C" int nlam_reduce_slabs_multi(const float* slab, int64_t nslabs, int64_t stride, int nseg,
                                       const int64_t* src_off, const int32_t* rows,
                                       const int32_t* cols, const int64_t* src_ld,
                                       float* const* dst, const int64_t* dst_ld, void* stream) {
  NLAM_REQUIRE(nseg >= 1 && nseg <= NLAM_MAX_SEGS, "reduce_slabs_multi: nseg %d out of [1,%d]",
               nseg, NLAM_MAX_SEGS);
  NLAM_REQUIRE(slab != nullptr && nslabs >= 1, "reduce_slabs_multi: no slabs");
  ReduceSegs q;
  q.nseg = nseg;
  q.first[0] = 0;
  for (int k = 0; k < nseg; ++k) {
    NLAM_REQUIRE(rows[k] >= 1 && cols[k] >= 1 && dst[k] != nullptr, "reduce_slabs_multi: bad segment");
    const int64_t tot = (int64_t)q.first[k] + (int64_t)rows[k] * cols[k];
    NLAM_REQUIRE(reduce_seg_fits(nslabs, stride, src_off[k], src_ld[k], dst_ld[k], tot),
                 "reduce_slabs_multi: segment %d exceeds the 32-bit field range", k);
    q.slab[k] = slab; q.nslabs[k] = (int32_t)nslabs; q.stride[k] = (int32_t)stride;
    q.src_off[k] = (int32_t)src_off[k]; q.cols[k] = cols[k];
    q.src_ld[k] = (int32_t)src_ld[k]; q.dst_ld[k] = (int32_t)dst_ld[k]; q.dst[k] = dst[k];
    q.first[k + 1] = (int32_t)tot;
  }
  return launch_reduce_segs(q, (hipStream_t)stream);
}

extern "C" int nlam_reduce_slabs_batch(int nseg, const float* const* slab, const int64_t* nslabs,
                                       const int64_t* stride, const int64_t* src_off,
                                       const int32_t* rows, const int32_t* cols,
                                       const int64_t* src_ld, float* const* dst,
                                       const int64_t* dst_ld, void* stream) {
  NLAM_REQUIRE(nseg >= 1 && nseg <= NLAM_MAX_SEGS, "reduce_slabs_batch: nseg %d out of [1,%d]",
               nseg, NLAM_MAX_SEGS);
  ReduceSegs q;
  q.nseg = nseg;
  q.first[0] = 0;
  for (int k = 0; k < nseg; ++k) {
    NLAM_REQUIRE(slab[k] != nullptr && nslabs[k] >= 1 && rows[k] >= 1 && cols[k] >= 1 &&
                     dst[k] != nullptr,
                 "reduce_slabs_batch: bad segment %d", k);
    const int64_t tot = (int64_t)q.first[k] + (int64_t)rows[k] * cols[k];
    NLAM_REQUIRE(reduce_seg_fits(nslabs[k], stride[k], src_off[k], src_ld[k], dst_ld[k], tot),
                 "reduce_slabs_batch: segment %d exceeds the 32-bit field range", k);
    q.slab[k] = slab[k]; q.nslabs[k] = (int32_t)nslabs[k]; q.stride[k] = (int32_t)stride[k];
    q.src_off[k] = (int32_t)src_off[k]; q.cols[k] = cols[k];
    q.src_ld[k] = (int32_t)src_ld[k]; q.dst_ld[k] = (int32_t)dst_ld[k]; q.dst[k] = dst[k];
    q.first[k + 1] = (int32_t)tot;
  }
  return launch_reduce_segs(q, (hipStream_t)stream);
}
