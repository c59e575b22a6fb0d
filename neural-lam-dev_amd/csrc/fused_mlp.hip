// Fused row-MLP kernels (gfx950): the make_mlp blocks of the reference
// (utils.py:191-214) with hidden_layers == 1,
//     y = [res +] [LayerNorm]( W2 silu(W1 [x_a | x_b] + b1) + b2 ),
// and the single-Linear "projection" used by the algebraic split of the edge
// MLP's first layer (W1 [e; x_s; x_r] = W1e e + W1s x_s + W1r x_r).
//
// One wavefront owns a tile of 32 rows (fused_common.h); a 256-thread workgroup
// keeps the weights in LDS and walks tiles persistently.  Forward saves nothing;
// backward recomputes the hidden activations from the inputs.
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

// LDS layout (floats): W1s[HID][k_pad+4] | W2s[32*NOUTB][HID+4] | b1s[HID] | b2s | gs | bs |
//                      tiles[4][32][ldt],  ldt = max(k_pad, HID, 32*NOUTB) + 4
template <int HID, int NOUTB>
__device__ __forceinline__ int mlp_ldt(int k_pad) {
  int m = k_pad;
  if (HID > m) m = HID;
  if (32 * NOUTB > m) m = 32 * NOUTB;
  return m + 4;
}

template <int HID, int NOUTB, bool HAS_LN>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NBH = HID / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldw1 = p.k_pad + 4, ldw2 = HID + 4;
  const int ldt = mlp_ldt<HID, NOUTB>(p.k_pad);
  float* W1s = smem;
  float* W2s = W1s + HID * ldw1;
  float* b1s = W2s + 32 * NOUTB * ldw2;
  float* b2s = b1s + HID;
  float* gs = b2s + 32 * NOUTB;
  float* bs = gs + 32 * NOUTB;
  float* tile = bs + 32 * NOUTB + wave * (NLAM_TILE * ldt);

  load_weight_lds(W1s, p.W1, p.ldW1, HID, p.k_in, HID, p.k_pad, tid, 256);
  load_weight_lds(W2s, p.W2, p.ldW2, p.n_out, HID, 32 * NOUTB, HID, tid, 256);
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
    // stage [x_a | x_b] rows
    int col = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s < p.nsrc) {
        const RowView v = p.src[s];
        const float* base = v.ptr + b * v.bstride + r0 * v.ld;
        auto rp = [&](int t) { return base + (int64_t)t * v.ld; };
        if ((p.vec_mask >> s) & 1)
          stage_rows<true, false>(tile, ldt, col, v.width, nrows, lane, rp);
        else
          stage_rows<false, false>(tile, ldt, col, v.width, nrows, lane, rp);
        col += v.width;
      }
    }
    if (p.k_pad > col) zero_cols(tile, ldt, col, p.k_pad - col, lane);
    wave_sync();

    f32x16 a1[NBH];
    vec_to_acc<NBH>(a1, b1s, lane);
    gemm_tile<NBH>(a1, W1s, ldw1, tile, ldt, p.k_pad >> 3, lane);
#pragma unroll
    for (int nb = 0; nb < NBH; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[nb][r] = nlam_silu(a1[nb][r]);
    f32x16 a2[NOUTB];
    vec_to_acc<NOUTB>(a2, b2s, lane);
    gemm_acc<NOUTB, NBH>(a2, W2s, ldw2, 0, a1, lane);
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
        store_rows_res<true>(tile, ldt, 0, p.n_out, nrows, lane, op, rp);
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

template <int HID, int NOUTB, bool HAS_LN>
static int launch_mlp_fwd(const MlpParams& p, hipStream_t s) {
  const size_t lds = mlp_lds_bytes<HID, NOUTB>(p.k_pad);
  NLAM_REQUIRE(lds <= 160 * 1024, "mlp_fwd: LDS footprint %zu B exceeds 160 KiB (k_in=%d)", lds,
               p.k_in);
  auto kern = mlp_fwd_kernel<HID, NOUTB, HAS_LN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
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
  const bool ln = gamma != nullptr;
  const int noutb = (n_out + 31) / 32;
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
struct LinParams {
  RowView x;
  int k_pad;
  const float* WA; int64_t ldWA; const float* bA; int nA;
  const float* WB; int64_t ldWB; const float* bB; int nB;
  float* out; int64_t out_bstride; int64_t out_ld;
  int64_t rows; int B; int vec_mask;  // bit0: x, bit3: out
};

template <int NOUTB>
__global__ __launch_bounds__(256) void lin_fwd_kernel(LinParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const int64_t tiles_per_b = (p.rows + NLAM_TILE - 1) / NLAM_TILE;
  const int64_t ntiles = tiles_per_b * p.B;
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < ntiles; tt += (int64_t)gridDim.x * 4) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_TILE;
    const int nrows = (int)((p.rows - r0) < NLAM_TILE ? (p.rows - r0) : NLAM_TILE);
    const float* base = p.x.ptr + b * p.x.bstride + r0 * p.x.ld;
    auto rp = [&](int t) { return base + (int64_t)t * p.x.ld; };
    if (p.vec_mask & 1)
      stage_rows<true, false>(tile, ldt, 0, p.x.width, nrows, lane, rp);
    else
      stage_rows<false, false>(tile, ldt, 0, p.x.width, nrows, lane, rp);
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
}

template <int NOUTB>
static int launch_lin_fwd(const LinParams& p, hipStream_t s) {
  const int ldt = (p.k_pad > 32 * NOUTB ? p.k_pad : 32 * NOUTB) + 4;
  const size_t lds = ((size_t)32 * NOUTB * (p.k_pad + 4) + 32 * NOUTB + (size_t)4 * NLAM_TILE * ldt) *
                     sizeof(float);
  NLAM_REQUIRE(lds <= 160 * 1024, "lin_fwd: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = lin_fwd_kernel<NOUTB>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const int64_t ntiles = ((p.rows + NLAM_TILE - 1) / NLAM_TILE) * p.B;
  kern<<<persistent_grid(ntiles, lds), 256, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("lin_fwd_kernel");
  return 0;
}

extern "C" int nlam_lin_fwd(const float* x, int64_t x_bstride, int64_t x_ld, int k_in,
                            const float* WA, int64_t ldWA, const float* bA, int nA,
                            const float* WB, int64_t ldWB, const float* bB, int nB,
                            float* out, int64_t out_bstride, int64_t out_ld, int64_t B,
                            int64_t rows, void* stream) {
  if (B <= 0 || rows <= 0) return 0;
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
  if (view_vec_ok(x, x_bstride, x_ld, k_in)) p.vec_mask |= 1;
  if (view_vec_ok(out, out_bstride, out_ld, p.nA + p.nB)) p.vec_mask |= 8;
  hipStream_t s = (hipStream_t)stream;
  switch ((p.nA + p.nB) / 32) {
    case 2: return launch_lin_fwd<2>(p, s);
    case 4: return launch_lin_fwd<4>(p, s);
    case 8: return launch_lin_fwd<8>(p, s);
    default:
      nlam_set_error("nlam_lin_fwd: total output width %d unsupported", p.nA + p.nB);
      return 1;
  }
}
