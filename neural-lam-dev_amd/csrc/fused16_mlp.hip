// 16-row, two-waves-per-SIMD forms of the row-MLP backward kernels (gfx950): the make_mlp
// blocks of the reference (utils.py:191-214, hidden_layers == 1, hidden width 64), the
// projection Linear of the split first edge-MLP layer (interaction_net.py:121) and the deferred
// first-layer weight gradient.  Same parameter blocks, slab layouts, grids and C entry points
// as the 32-row kernels of fused_mlp.hip (which remain for exact-fp32 arithmetic and for the
// shapes listed as unsupported below); building blocks and layout: fused16.h.
//
// Workgroup = 512 threads = 8 wavefronts (two per SIMD) that share ONE copy of the weight
// images in LDS; every wavefront walks its own 16-row tiles.  Rows move between global memory
// and registers directly in accumulator layout; LDS carries only the weight images and the
// bf16 planes of the products that contract over rows (weight / bias / LayerNorm gradients).
#include <cstdlib>

#include "fused16.h"
#include "fused_params.h"

#define K16_NW 8
#define K16_THREADS 512
#ifndef MLP_BWD16_NODW
#define MLP_BWD16_NODW 0   // diagnostic build (-DMLP_BWD16_NODW=1): the data path only, no weight / bias / LayerNorm gradients
#endif
constexpr bool NODW = MLP_BWD16_NODW != 0;

static int g_k16_mask = -2;
bool nlam_k16_on(int family) {
  if (g_k16_mask == -2) {
    const char* e = getenv("NLAM_K16");
    g_k16_mask = e ? atoi(e) : K16_DEFAULT;
  }
  return (g_k16_mask & family) != 0;
}
extern "C" int nlam_set_k16(int mask) {
  (void)nlam_k16_on(0);   // (resolve the default / NLAM_K16 first)
  const int prev = g_k16_mask;
  if (mask >= 0) g_k16_mask = mask;
  return prev;
}

// this lane's slice of a row [xa | xb] (column c0 = 16 fb + 4 g): unconditional loads from a
// clamped valid address, zeroed past the end (no branches around loads)
template <int KF>
__device__ __forceinline__ void load_cat16(f32x4* __restrict__ x, const float* __restrict__ ra,
                                           int wa, const float* __restrict__ rb, int wb, int lane) {
  const int g = lane >> 4;
  // all KF loads first, the zeroing of the columns past the end afterwards: with the select
  // next to each load the compiler re-used one register quad and waited for every load in turn
  // (four serial global round trips per tile in the .s)
  bool live[KF];
#pragma unroll
  for (int fb = 0; fb < KF; ++fb) {
    const int c0 = 16 * fb + 4 * g;
    const bool in_a = c0 < wa, in_b = !in_a && (c0 - wa) < wb;
    const float* p = in_a ? ra + c0 : (in_b ? rb + (c0 - wa) : ra);
    live[fb] = in_a || in_b;
    x[fb] = *reinterpret_cast<const f32x4*>(p);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int fb = 0; fb < KF; ++fb)
    if (!live[fb]) x[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
}
template <int KF>
__device__ __forceinline__ void store_cols16(float* __restrict__ row, const f32x4* __restrict__ a,
                                             int fb0, int width, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int fb = 0; fb < KF; ++fb) {
    const int c0 = 16 * fb + 4 * g;
    if (c0 < width) *reinterpret_cast<f32x4*>(row + c0) = a[fb0 + fb];
  }
}
// narrow / unaligned rows: element loads from clamped addresses -- all loads first, the zeroing
// of the columns past the end afterwards (a select next to each load serialises the loads)
// N4: width <= 4 (the 2-3 wide static features of the embedders): only the first quad of the lanes
// g == 0 holds live columns, so FOUR element loads per lane cover the row instead of 4 NF (twelve
// of the embedders' sixteen loads per tile and row pass fetched a clamped column to zero it)
template <int NF, bool N4 = false>
__device__ __forceinline__ void load_narrow16(f32x4* __restrict__ a, const float* __restrict__ row,
                                              int width, int lane) {
  const int g = lane >> 4;
  if constexpr (N4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a[0][r] = row[r < width ? r : 0];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r >= width || g != 0) a[0][r] = 0.f;
#pragma unroll
    for (int fb = 1; fb < NF; ++fb) a[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * fb + 4 * g + r;
      a[fb][r] = row[f < width ? f : 0];
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int fb = 0; fb < NF; ++fb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * fb + 4 * g + r;
      if (f >= width) a[fb][r] = 0.f;
    }
}

// ================================================================ MLP backward
// KB: input blocks of 32 (k_in <= 32 KB); NOB: output blocks of 32; DEFER: the first layer's
// weight gradient is left to nlam_outer_bwd (ga_out written) -- the K = 128 node update.
// Sources: KB == 1: one source, any width <= 32 and alignment (element loads);
//          KB == 2: one source, 16-byte aligned rows, width % 4 == 0;
//          KB == 4: two sources of 64 columns each, 16-byte aligned.
// gy: NOB == 2: 64 columns, 16-byte aligned; NOB == 1: any width <= 32 (element loads).
template <int KB, int NOB, bool HAS_LN, bool DEFER, int TERMS, bool N4 = false>
__device__ __forceinline__ void mlp_bwd16_body(const MlpBwdParams& q, int wg, int nwg, char* smem16) {
  static_assert(!N4 || KB == 1, "N4 is a form of the narrow-input (KB = 1) kernels");
  constexpr int HID = 64, NFH = 4, KF = 2 * KB, NFO = 2 * NOB, NO = 32 * NOB, KP32 = 32 * KB;
  constexpr int KBA = DEFER ? 1 : KB;
  const MlpParams& p = q.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int t = lane & 15;
  char* cur = smem16;
  // (both images are also read transposed: swizzled layout, fused16.h)
  const B3Image W1im = w16_image(cur, HID, KP32, true);
  cur += w16_image_bytes(HID, KP32, true);
  const B3Image W2im = w16_image(cur, NO, HID, true);
  cur += w16_image_bytes(NO, HID, true);
  float* b1s = reinterpret_cast<float*>(cur);
  float* b2s = b1s + HID;
  float* gs = b2s + NO;
  cur += (HID + 2 * NO) * sizeof(float);
  constexpr int PW0 = (!DEFER && KP32 > HID) ? KP32 : HID;   // S planes, later X planes
  constexpr int PW1 = HID > NO ? HID : NO;                   // g / prod / GZ planes, later GA
  constexpr size_t HST = DEFER ? 0 : (size_t)NLAM_T16 * (HID + 4) * sizeof(float);   // h stash
  constexpr size_t PL0 = NODW ? 0 : p16_bytes(PW0), PL1 = NODW ? 0 : p16_bytes(PW1);   // (NODW: no planes)
  char* mine = cur + wave * (PL0 + PL1 + HST);
  void* R0 = mine;
  void* R1 = mine + PL0;
  float* HS = reinterpret_cast<float*>(mine + PL0 + PL1);

  {   // every global load of the prologue in flight together (fused16.h, batched prologue loads)
    VLoad16 lv;
    const float* const vecs[8] = {p.b1, p.b2, p.gamma, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int lens[8] = {HID, p.n_out, p.n_out, 0, 0, 0, 0, 0};
    static_assert(NO <= 64, "vector slots are 64 floats");
    WLoad16<KB> l1;
    WLoad16<NOB> l2;
    if constexpr (NO == 64) v16_issue(lv, vecs, lens, tid);
    w16_issue(l1, p.W1, p.ldW1, HID, p.k_in, HID, KP32, tid, K16_THREADS);
    w16_issue(l2, p.W2, p.ldW2, p.n_out, HID, NO, HID, tid, K16_THREADS);
    if constexpr (NO == 64) {
      v16_commit(lv, b1s, 3, tid);
    } else {
      load_vec_lds(b1s, p.b1, HID, HID, tid, K16_THREADS);
      load_vec_lds(b2s, p.b2, p.n_out, NO, tid, K16_THREADS);
      load_vec_lds(gs, p.gamma, p.n_out, NO, tid, K16_THREADS);
    }
    w16_commit(l1, W1im, 0, p.W1, p.ldW1, HID, p.k_in, HID, KP32, tid, K16_THREADS);
    w16_commit(l2, W2im, 0, p.W2, p.ldW2, p.n_out, HID, NO, HID, tid, K16_THREADS);
  }
  __syncthreads();

  f32x16 dW1[2][KBA], dW2[NOB][2];
  if constexpr (!DEFER) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < KBA; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW1[i][j][r] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < NOB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW2[i][j][r] = 0.f;
  float db1[1] = {0.f}, db2[1] = {0.f}, dgam[1] = {0.f}, dbet[1] = {0.f};

  const int64_t tiles_per_b = (p.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * p.B;
  const bool want_gx = q.gxa != nullptr || q.gxb != nullptr;
  for (int64_t tt = (int64_t)wg * K16_NW + wave; tt < ntiles; tt += (int64_t)nwg * K16_NW) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int nrows = (int)((p.rows - r0) < NLAM_T16 ? (p.rows - r0) : NLAM_T16);
    const bool valid = t < nrows;
    const int64_t row = r0 + (valid ? t : nrows - 1);   // padded slots re-read a valid row

    // ---- this lane's slices of the x rows; recompute the forward: h, s = silu(h)
    auto load_x = [&](f32x4* x) {
      const int64_t rw = opaque(row);
      const float* ra = p.src[0].ptr + b * p.src[0].bstride + rw * p.src[0].ld;
      const float* rb = KB == 4 ? p.src[1].ptr + b * p.src[1].bstride + rw * p.src[1].ld : ra;
      if constexpr (KB == 1) load_narrow16<KF, N4>(x, ra, p.src[0].width, lane);
      else if constexpr (KB == 4) load_cat16<KF>(x, ra, 64, rb, 64, lane);
      else load_cat16<KF>(x, ra, p.src[0].width, ra, 0, lane);
    };
    const B3Tile Ts = p16_tile(R0, HID), Tz = p16_tile(R1, NO);
    f32x4 hkeep[DEFER ? NFH : 1];   // DEFER: h stays in registers (no weight-gradient blocks of W1)
    f32x4 g[NFO];
    const float* rg = q.gy.ptr + b * q.gy.bstride + opaque(row) * q.gy.ld;
    {
      f32x4 hpre[NFH];
      {
        f32x4 x[KF];
        load_x(x);
        // the upstream gradient rows are requested with the x rows: their latency rides under
        // the forward recompute (they used to be requested where they are first needed, after the
        // second GEMM, behind a scheduling fence: one exposed HBM round trip per tile)
        if constexpr (NOB == 2) load_row16<NFO>(g, rg, lane);
        vec_to_acc16<NFH>(hpre, b1s, lane);
        gemm_acc16<NFH, KB, TERMS>(hpre, W1im, 0, 0, x, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      // silu and silu' from ONE sigmoid (v_exp + v_rcp are quarter-rate); silu'(h) is what is
      // kept (registers / LDS stash) for the ga product, not h
      f32x4 sact[NFH];
#pragma unroll
      for (int fb = 0; fb < NFH; ++fb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sv, dv;
          silu_both(hpre[fb][r], sv, dv);
          sact[fb][r] = sv;
          hpre[fb][r] = dv;
        }
      if constexpr (DEFER) {
#pragma unroll
        for (int fb = 0; fb < NFH; ++fb) hkeep[fb] = hpre[fb];
      } else {
        acc16_to_tile<NFH>(hpre, HS, HID + 4, lane);          // silu'(h): back from LDS later
      }
      if constexpr (!NODW) acc16_to_planes<NFH, TERMS>(sact, Ts, 0, lane);         // S stays in R0 until dW2 is formed
      if constexpr (HAS_LN) {
        f32x4 z[NFO];
        vec_to_acc16<NFO>(z, b2s, lane);
        gemm_acc16<NFO, 2, TERMS>(z, W2im, 0, 0, sact, lane);
        __builtin_amdgcn_sched_barrier(0);
        mask16<NFO>(g, valid);   // padded rows carry a zero gradient: every sum below ignores them
        if constexpr (!NODW) {
          acc16_to_planes<NFO, TERMS>(g, Tz, 0, lane);          // dbeta summand: gy
          wave_sync();
          colsum16_64<TERMS>(dbet[0], Tz, 0, lane);
          wave_sync();
        }
        ln16_bwd<NFO, TERMS, !NODW>(z, g, Tz, gs, lane);      // g: gy -> gz; gy * xhat -> planes
        if constexpr (!NODW) {
          wave_sync();
          colsum16_64<TERMS>(dgam[0], Tz, 0, lane);
          wave_sync();
        }
      } else {
        if constexpr (NOB == 1) load_narrow16<NFO>(g, rg, p.n_out, lane);
        mask16<NFO>(g, valid);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- second layer's weight gradient: dW2 += gz (x) s, db2 += gz
    if constexpr (!NODW) {
      acc16_to_planes<NFO, TERMS>(g, Tz, 0, lane);
      wave_sync();
      if constexpr (NOB == 1) {
        colsum16_32<TERMS>(db2[0], Tz, 0, lane);
        outer_accum16<NOB, 2, TERMS>(dW2, Tz, 0, Ts, 0, lane);
      } else {
        outer_accum16_cs<2, 2, TERMS>(dW2, db2, Tz, 0, Ts, 0, lane);   // db2 from dW2's own A fragments
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- ga = (W2^T gz) * silu'(h)
    // (the x rows of the first layer's weight gradient and the residual's gy rows are requested
    // here: L2-hot re-reads whose round trip rides under the next two GEMMs instead of being
    // exposed where the values are first used)
    constexpr bool X2_EARLY = !DEFER && KB == 1 && !NODW;   // (KB = 2: 16 more live registers spill)
    f32x4 x2[DEFER ? 1 : KF];
    if constexpr (X2_EARLY) load_x(x2);
    // (the residual's rows early only where the registers allow it: the K = 128 node update, which
    // keeps no x rows here; the K = 64 blocks spilled with both sets live)
    constexpr bool GY2_EARLY = DEFER;
    f32x4 gy2[4];
    const bool add_gy = want_gx && q.gxa != nullptr && q.add_gy_to_gxa;
    if (GY2_EARLY && add_gy)
      load_row16<4>(gy2, q.gy.ptr + b * q.gy.bstride + opaque(row) * q.gy.ld, lane);
    f32x4 ga[NFH];
    zero16<NFH>(ga);
    gemm_acc16_wt<NFH, NOB, TERMS>(ga, W2im, 0, 0, g, lane);
    __builtin_amdgcn_sched_barrier(0);
    {
      const int gq4 = lane >> 4;
#pragma unroll
      for (int fb = 0; fb < NFH; ++fb) {
        f32x4 h4;
        if constexpr (DEFER) h4 = hkeep[fb];
        else h4 = *reinterpret_cast<const f32x4*>(HS + t * (HID + 4) + 16 * fb + 4 * gq4);
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[fb][r] *= h4[r];   // (h4 holds silu'(h))
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DEFER) {
      if (valid) store_row16<NFH>(q.ga_out + (b * p.rows + r0 + t) * HID, ga, lane);
    } else {
      // ---- first layer's weight gradient: dW1 += ga (x) x, db1 += ga (x re-read: L2-hot)
      wave_sync();
      const B3Tile Ta = p16_tile(R1, HID), Tx = p16_tile(R0, KP32);
      if constexpr (!NODW) {
        acc16_to_planes<NFH, TERMS>(ga, Ta, 0, lane);
        if constexpr (!X2_EARLY) load_x(x2);
        acc16_to_planes<KF, TERMS>(x2, Tx, 0, lane);
        wave_sync();
        outer_accum16_cs<2, KB, TERMS>(dW1, db1, Ta, 0, Tx, 0, lane);   // db1 from dW1's own A fragments
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- input gradient gx = W1^T ga (+ gy: the residual taken from source a)
    if (want_gx) {
      // (the residual's gy rows -- an L2-hot re-read -- are requested in front of the W1^T product
      // instead of behind it: x2 is dead here, so the four quads fit, and the round trip rides
      // under the 24 MFMAs)
      if (!GY2_EARLY && add_gy)
        load_row16<4>(gy2, q.gy.ptr + b * q.gy.bstride + opaque(row) * q.gy.ld, lane);
      f32x4 gx[KF];
      zero16<KF>(gx);
      gemm_acc16_wt<KF, 2, TERMS>(gx, W1im, 0, 0, ga, lane);
      const int wa = p.src[0].width;
      if (q.gxa != nullptr) {
        if (q.add_gy_to_gxa) {   // n_out == wa == 64 (checked by the host side)
#pragma unroll
          for (int fb = 0; fb < 4 && fb < KF; ++fb) gx[fb] += gy2[fb];
        }
        float* o = q.gxa + b * q.gxa_bstride + (r0 + t) * q.gxa_ld;
        if (valid) {
          if constexpr (KB == 1) store_row16_s<KF>(o, gx, wa, lane);
          else if constexpr (KB == 4) store_cols16<4>(o, gx, 0, 64, lane);
          else store_cols16<KF>(o, gx, 0, wa, lane);
        }
      }
      if constexpr (KB == 4) {
        if (q.gxb != nullptr && valid)
          store_cols16<4>(q.gxb + b * q.gxb_bstride + (r0 + t) * q.gxb_ld, gx, 4, 64, lane);
      }
    }
    wave_sync();   // the planes are rewritten by the next tile
  }

  if constexpr (NODW) return;   // (diagnostic build: nothing was accumulated)
  // ---- fold the eight waves' partials in LDS (fixed order) and write the slab
  __syncthreads();
  float* img = reinterpret_cast<float*>(smem16);   // weights and planes are dead
  float* slab = q.slab + (int64_t)wg * q.slab_stride;
  constexpr int n1 = HID * KP32, n2 = NO * HID;
  if constexpr (!DEFER) {
    fold_blocks_to_slab16<2, KBA, KBA, K16_NW>(&dW1[0][0], img, KP32, slab, tid, wave, lane);
    fold_vec_to_slab16<1, K16_NW>(db1, img, slab + n1, HID, tid, wave, lane);
  }
  fold_blocks_to_slab16<NOB, 2, 2, K16_NW>(&dW2[0][0], img, HID, slab + n1 + HID, tid, wave, lane);
  float* vbase = slab + n1 + HID + n2;
  fold_vec_to_slab16<1, K16_NW>(db2, img, vbase, NO, tid, wave, lane);
  if constexpr (HAS_LN) {
    fold_vec_to_slab16<1, K16_NW>(dgam, img, vbase + NO, NO, tid, wave, lane);
    fold_vec_to_slab16<1, K16_NW>(dbet, img, vbase + 2 * NO, NO, tid, wave, lane);
  }
}

template <int KB, int NOB, bool HAS_LN, bool DEFER, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void mlp_bwd16_kernel(MlpBwdParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  mlp_bwd16_body<KB, NOB, HAS_LN, DEFER, TERMS>(q, (int)blockIdx.x, (int)gridDim.x, smem16);
}

// several independent narrow-input MLPs (the static-feature embedders of a model: mesh nodes and
// every edge set, utils.py:191-214 applied to 2-3 wide rows) in one launch: KB = 1, LayerNorm
#define K16_MAXM 8
struct MlpBwdMulti {
  MlpBwdParams q[K16_MAXM];
  int n;
  int wg0[K16_MAXM + 1];
};
template <int TERMS, bool N4>
__global__ __launch_bounds__(K16_THREADS, 2) void mlp_bwd16_multi_kernel(MlpBwdMulti m) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  int k = 0;
  while (k + 1 < m.n && (int)blockIdx.x >= m.wg0[k + 1]) ++k;
  mlp_bwd16_body<1, 2, true, false, TERMS, N4>(m.q[k], (int)blockIdx.x - m.wg0[k],
                                               m.wg0[k + 1] - m.wg0[k], smem16);
}

template <int KB, int NOB, bool HAS_LN, bool DEFER>
static int launch_mlp_bwd16(const MlpBwdParams& q, hipStream_t s) {
  constexpr int HID = 64, NO = 32 * NOB, KP32 = 32 * KB;
  constexpr int PW0 = (!DEFER && KP32 > HID) ? KP32 : HID;
  constexpr int PW1 = HID > NO ? HID : NO;
  size_t lds = w16_image_bytes(HID, KP32, true) + w16_image_bytes(NO, HID, true) +
               (HID + 2 * NO) * sizeof(float) +
               K16_NW * ((NODW ? 0 : p16_bytes(PW0) + p16_bytes(PW1)) +
                         (DEFER ? 0 : (size_t)NLAM_T16 * (HID + 4) * sizeof(float)));
  size_t fold = (size_t)K16_NW * HID * HID * sizeof(float);          // dW2 images
  if (!DEFER && (size_t)K16_NW * HID * KP32 * sizeof(float) > fold)
    fold = (size_t)K16_NW * HID * KP32 * sizeof(float);
  if (fold > lds && !NODW) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "mlp_bwd16: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = mlp_bwd16_kernel<KB, NOB, HAS_LN, DEFER, 3>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles32 = ((q.f.rows + 31) / 32) * q.f.B;
  unsigned grid = (unsigned)nlam_bwd_grid(ntiles32);
  if (NODW) {   // diagnostic: NLAM_NODW_WGS workgroups per CU (the slab fold then overruns: timing only)
    static const int per = getenv("NLAM_NODW_WGS") ? atoi(getenv("NLAM_NODW_WGS")) : 1;
    grid = grid * (per > 1 ? per : 1);
    if (grid > 256u * per) grid = 256u * per;
  }
  kern<<<grid, K16_THREADS, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("mlp_bwd16_kernel");
  return 0;
}

int nlam_k16_mlp_bwd(const MlpBwdParams& q, hipStream_t s) {
  if (!nlam_k16_on(K16_MLP_BWD) || !nlam_mfma_b3() || q.stamp) return -1;
  const MlpParams& p = q.f;
  const bool ln = p.gamma != nullptr;
  const int kb = (p.k_in + 31) / 32;
  const bool a_vec = (p.vec_mask & 1) != 0;
  // gy as 64 aligned columns / anything narrow
  const bool gy64 = p.n_out == 64 && q.vec_gy;
  if (ln && !gy64) return -1;
  if (q.gxa != nullptr && !(q.vec_gxa || kb == 1)) return -1;
  if (q.gxb != nullptr && !q.vec_gxb) return -1;
  if (q.add_gy_to_gxa && !(gy64 && p.src[0].width == 64)) return -1;
  if (ln && p.nsrc == 1 && kb == 1) return launch_mlp_bwd16<1, 2, true, false>(q, s);
  if (ln && p.nsrc == 1 && kb == 2 && a_vec) return launch_mlp_bwd16<2, 2, true, false>(q, s);
  if (ln && p.nsrc == 2 && kb == 4 && q.ga_out != nullptr && a_vec && (p.vec_mask & 2) &&
      p.src[0].width == 64 && p.src[1].width == 64)
    return launch_mlp_bwd16<4, 2, true, true>(q, s);
  if (!ln && p.nsrc == 1 && kb == 2 && a_vec && p.n_out <= 32)
    return launch_mlp_bwd16<2, 1, false, false>(q, s);
  return -1;
}

// ====================================================== projection backward
// y = x [WA; WB]^T (+ b): gx = gy [WA; WB] (+ gx_add), dW = gy^T x, db = colsum(gy); k_in = 64,
// n_out = 32 NOB in {64, 128}, 16-byte aligned views.  gy_nsum > 1: x is batch-invariant and gy
// is summed over its gy_nsum batch slices while it is loaded (fixed order).  gh != NULL: columns
// [0, 64) of gy are the sums of the edge-gradient rows over the row's sender list (fused16.h,
// gather_sender_sum16) instead of a stored tensor.  One workgroup `wg` of `nwg` of a problem.
template <int NOB, int TERMS, bool GATHER>
__device__ __forceinline__ void lin_bwd16_body(const LinBwdParams& q, int wg, int nwg, char* smem16) {
  constexpr int K = 64, KF = 4, NO = 32 * NOB, NFO = 2 * NOB, NV = NOB / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int t = lane & 15;
  const B3Image Wim = w16_image(smem16, NO, K, true);   // read transposed: swizzled layout
  char* mine = smem16 + w16_image_bytes(NO, K, true) + wave * (p16_bytes(NO) + p16_bytes(K));
  const B3Tile Tg = p16_tile(mine, NO), Tx = p16_tile(mine + p16_bytes(NO), K);
  {
    WLoad16<NOB> la, lb;   // (NOB x 512 float4 cover up to 32 NOB rows of 64)
    w16_issue(la, q.WA, q.ldWA, q.nA, K, q.nA, K, tid, K16_THREADS);
    if (q.nB > 0) w16_issue(lb, q.WB, q.ldWB, q.nB, K, NO - q.nA, K, tid, K16_THREADS);
    w16_commit(la, Wim, 0, q.WA, q.ldWA, q.nA, K, q.nA, K, tid, K16_THREADS);
    if (q.nB > 0) w16_commit(lb, Wim, q.nA, q.WB, q.ldWB, q.nB, K, NO - q.nA, K, tid, K16_THREADS);
  }
  __syncthreads();
  f32x16 dW[NOB][2];
#pragma unroll
  for (int i = 0; i < NOB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW[i][j][r] = 0.f;
  float db[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) db[j] = 0.f;

  const int64_t tiles_per_b = (q.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * q.B;
  for (int64_t tt = (int64_t)wg * K16_NW + wave; tt < ntiles; tt += (int64_t)nwg * K16_NW) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int nrows = (int)((q.rows - r0) < NLAM_T16 ? (q.rows - r0) : NLAM_T16);
    const bool valid = t < nrows;
    const int64_t row = r0 + (valid ? t : nrows - 1);
    f32x4 x[KF], g[NFO];
    load_row16<KF>(x, q.x.ptr + b * q.x.bstride + row * q.x.ld, lane);
    const float* rg = q.gy.ptr + b * q.gy.bstride + row * q.gy.ld;
    constexpr int F0 = 4;   // quads of the gathered sender block
    if (GATHER && q.gh != nullptr) {
      gather_sender_sum16(g, q.gh + b * q.gh_bstride, q.csc_colptr, q.csc_eid, row, q.n_send, valid,
                          lane);
      for (int sl = 1; sl < q.gy_nsum; ++sl) {
        f32x4 tv[F0];
        gather_sender_sum16(tv, q.gh + (b + sl) * q.gh_bstride, q.csc_colptr, q.csc_eid, row, q.n_send,
                            valid, lane);
#pragma unroll
        for (int fb = 0; fb < F0; ++fb) g[fb] += tv[fb];
      }
      if constexpr (NFO > F0) {
        load_row16<NFO - F0>(g + F0, rg + 16 * F0, lane);
        for (int sl = 1; sl < q.gy_nsum; ++sl) {
          f32x4 tv[NFO - F0];
          load_row16<NFO - F0>(tv, rg + 16 * F0 + (int64_t)sl * q.gy_sum_stride, lane);
#pragma unroll
          for (int fb = 0; fb < NFO - F0; ++fb) g[F0 + fb] += tv[fb];
        }
      }
    } else {
      load_row16<NFO>(g, rg, lane);
      for (int sl = 1; sl < q.gy_nsum; ++sl) {
        f32x4 tv[NFO];
        load_row16<NFO>(tv, rg + (int64_t)sl * q.gy_sum_stride, lane);
#pragma unroll
        for (int fb = 0; fb < NFO; ++fb) g[fb] += tv[fb];
      }
    }
    mask16<NFO>(g, valid);
    acc16_to_planes<NFO, TERMS>(g, Tg, 0, lane);
    acc16_to_planes<KF, TERMS>(x, Tx, 0, lane);
    wave_sync();
    outer_accum16_cs<NOB, 2, TERMS>(dW, db, Tg, 0, Tx, 0, lane);   // (db from dW's own A fragments)
    if (q.gx != nullptr) {
      f32x4 gx[KF];
      zero16<KF>(gx);
      gemm_acc16_wt<KF, NOB, TERMS>(gx, Wim, 0, 0, g, lane);
      if (q.gx_add != nullptr) {
        f32x4 ad[KF];
        load_row16<KF>(ad, q.gx_add + b * q.ga_bstride + row * q.ga_ld, lane);
#pragma unroll
        for (int fb = 0; fb < KF; ++fb) gx[fb] += ad[fb];
      }
      if (valid) store_row16<KF>(q.gx + b * q.gx_bstride + (r0 + t) * q.gx_ld, gx, lane);
    }
    wave_sync();
  }
  __syncthreads();
  float* img = reinterpret_cast<float*>(smem16);
  float* slab = q.slab + (int64_t)wg * q.slab_stride;
  // 64 output rows at a time (8 x 64 x 64 floats = 128 KB of images)
#pragma unroll
  for (int h2 = 0; h2 < NOB / 2; ++h2)
    fold_blocks_to_slab16<2, 2, 2, K16_NW>(&dW[2 * h2][0], img, K, slab + 64 * h2 * K, tid, wave, lane);
  fold_vec_to_slab16<NV, K16_NW>(db, img, slab + NO * K, NO, tid, wave, lane);
}

template <int NOB, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void lin_bwd16_kernel(LinBwdParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  lin_bwd16_body<NOB, TERMS, false>(q, (int)blockIdx.x, (int)gridDim.x, smem16);
}

// several independent projections in one launch (the sender / receiver / edge thirds of an
// InteractionNet's first edge-MLP Linear, interaction_net.py:121: three row sets, three weights)
#define K16_MAXP 4
template <int NXB, int TERMS>
__device__ __forceinline__ void outer_bwd16_body(const OuterParams& q, int wg, int nwg, char* smem16);
// kind 0: a projection (q); kind 1: a deferred 64 x 128 first-layer weight gradient (o)
struct LinBwdMulti {
  LinBwdParams q[K16_MAXP];
  OuterParams o[K16_MAXP];
  int kind[K16_MAXP];
  int n;
  int wg0[K16_MAXP + 1];   // first workgroup of problem k
};
template <int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void lin_bwd16_multi_kernel(LinBwdMulti m) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  int k = 0;
  while (k + 1 < m.n && (int)blockIdx.x >= m.wg0[k + 1]) ++k;
  const int wg = (int)blockIdx.x - m.wg0[k], nwg = m.wg0[k + 1] - m.wg0[k];
  if (m.kind[k] == 1) outer_bwd16_body<4, TERMS>(m.o[k], wg, nwg, smem16);
  else lin_bwd16_body<2, TERMS, true>(m.q[k], wg, nwg, smem16);
}

template <int NOB>
static size_t lin_bwd16_lds() {
  constexpr int K = 64, NO = 32 * NOB;
  size_t lds = w16_image_bytes(NO, K, true) + K16_NW * (p16_bytes(NO) + p16_bytes(K));
  const size_t fold = (size_t)K16_NW * 64 * K * sizeof(float);
  return fold > lds ? fold : lds;
}

template <int NOB>
static int launch_lin_bwd16(const LinBwdParams& q, hipStream_t s) {
  const size_t lds = lin_bwd16_lds<NOB>();
  NLAM_REQUIRE(lds <= 160 * 1024, "lin_bwd16: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = lin_bwd16_kernel<NOB, 3>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles32 = ((q.rows + 31) / 32) * q.B;
  kern<<<(unsigned)nlam_bwd_grid(ntiles32), K16_THREADS, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("lin_bwd16_kernel");
  return 0;
}

static bool lin_bwd16_ok(const LinBwdParams& q) {
  if (q.x.width != 64 || !q.vec_x || !q.vec_gy) return false;
  if (q.gx != nullptr && !q.vec_gx) return false;
  return true;
}

int nlam_k16_lin_bwd(const LinBwdParams& q, hipStream_t s) {
  if (!nlam_k16_on(K16_LIN_BWD) || !nlam_mfma_b3()) return -1;
  const int n_out = q.nA + q.nB;
  if (!lin_bwd16_ok(q) || q.gh != nullptr) return -1;   // (the gather exists in the multi form)
  if (n_out == 64) return launch_lin_bwd16<2>(q, s);
  if (n_out == 128) return launch_lin_bwd16<4>(q, s);
  return -1;
}

int nlam_k16_lin_bwd_multi(const LinBwdParams* q, const OuterParams* o, const int* kind, int n,
                           hipStream_t s) {
  if (!nlam_k16_on(K16_LIN_BWD) || !nlam_k16_on(K16_OUTER_BWD) || !nlam_mfma_b3() || n < 1 ||
      n > K16_MAXP)
    return -1;
  LinBwdMulti m;
  m.n = n;
  m.wg0[0] = 0;
  size_t lds = lin_bwd16_lds<2>();
  for (int k = 0; k < n; ++k) {
    m.kind[k] = kind[k];
    m.q[k] = q[k];
    m.o[k] = o[k];
    int64_t rows, B;
    if (kind[k] == 1) {
      if (o[k].g.width != 64 || o[k].xa.width != 64 || o[k].xb.ptr == nullptr || o[k].xb.width != 64 ||
          o[k].x_index != nullptr)
        return -1;
      rows = o[k].rows; B = o[k].B;
      const size_t need = (size_t)K16_NW * (p16_bytes(64) + p16_bytes(128));
      const size_t fold = (size_t)K16_NW * 32 * 128 * sizeof(float);
      if (need > lds) lds = need;
      if (fold > lds) lds = fold;
    } else {
      if (!lin_bwd16_ok(q[k]) || q[k].nA + q[k].nB != 64) return -1;
      rows = q[k].rows; B = q[k].B;
    }
    m.wg0[k + 1] = m.wg0[k] + (int)nlam_bwd_grid(((rows + 31) / 32) * B);
  }
  for (int k = n; k < K16_MAXP; ++k) {
    m.wg0[k + 1] = m.wg0[n];
    m.kind[k] = 0;
  }
  NLAM_REQUIRE(lds <= 160 * 1024, "lin_bwd16_multi: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = lin_bwd16_multi_kernel<3>;
  NLAM_BIG_LDS(kern, __func__);
  kern<<<(unsigned)m.wg0[n], K16_THREADS, lds, s>>>(m);
  NLAM_CHECK_LAUNCH("lin_bwd16_multi_kernel");
  return 0;
}

// ================================================ deferred weight gradients
// dW (64 x 32 NXB) = sum_rows G[r]^T (x) [xa | xb][r], db = colsum(G); x rows optionally
// gathered by x_index.  Slab: [dW | db] as nlam_outer_bwd.
template <int NXB, int TERMS>
__device__ __forceinline__ void outer_bwd16_body(const OuterParams& q, int wg, int nwg, char* smem16) {
  constexpr int NG = 64, NX = 32 * NXB, KF = 2 * NXB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int t = lane & 15;
  char* mine = smem16 + wave * (p16_bytes(NG) + p16_bytes(NX));
  const B3Tile Tg = p16_tile(mine, NG), Tx = p16_tile(mine + p16_bytes(NG), NX);
  f32x16 dW[2][NXB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NXB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW[i][j][r] = 0.f;
  float db[1] = {0.f};
  const int wa = q.xa.width, wb = q.xb.ptr ? q.xb.width : 0;
  const int64_t tiles_per_b = (q.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * q.B;
  for (int64_t tt = (int64_t)wg * K16_NW + wave; tt < ntiles; tt += (int64_t)nwg * K16_NW) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int nrows = (int)((q.rows - r0) < NLAM_T16 ? (q.rows - r0) : NLAM_T16);
    const bool valid = t < nrows;
    const int64_t row = r0 + (valid ? t : nrows - 1);
    const int64_t xrow = q.x_index ? (int64_t)q.x_index[row] : row;
    f32x4 g[4], x[KF];
    load_row16<4>(g, q.g.ptr + b * q.g.bstride + row * q.g.ld, lane);
    const float* ra = q.xa.ptr + b * q.xa.bstride + xrow * q.xa.ld;
    const float* rb = q.xb.ptr ? q.xb.ptr + b * q.xb.bstride + xrow * q.xb.ld : ra;
    load_cat16<KF>(x, ra, wa, rb, wb, lane);
    mask16<4>(g, valid);
    acc16_to_planes<4, TERMS>(g, Tg, 0, lane);
    acc16_to_planes<KF, TERMS>(x, Tx, 0, lane);
    wave_sync();
    outer_accum16_cs<2, NXB, TERMS>(dW, db, Tg, 0, Tx, 0, lane);
    wave_sync();
  }
  __syncthreads();
  float* img = reinterpret_cast<float*>(smem16);
  float* slab = q.slab + (int64_t)wg * q.slab_stride;
  // one 32-row block of dW at a time: 8 x 32 x NX floats of images (<= 128 KB)
#pragma unroll
  for (int ib = 0; ib < 2; ++ib)
    fold_blocks_to_slab16<1, NXB, NXB, K16_NW>(&dW[ib][0], img, NX, slab + 32 * ib * NX, tid, wave, lane);
  fold_vec_to_slab16<1, K16_NW>(db, img, slab + NG * NX, NG, tid, wave, lane);
}

template <int NXB, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void outer_bwd16_kernel(OuterParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  outer_bwd16_body<NXB, TERMS>(q, (int)blockIdx.x, (int)gridDim.x, smem16);
}

template <int NXB>
static int launch_outer_bwd16(const OuterParams& q, hipStream_t s) {
  constexpr int NG = 64, NX = 32 * NXB;
  size_t lds = K16_NW * (p16_bytes(NG) + p16_bytes(NX));
  const size_t fold = (size_t)K16_NW * 32 * NX * sizeof(float);
  if (fold > lds) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "outer_bwd16: LDS footprint %zu B exceeds 160 KiB", lds);
  auto kern = outer_bwd16_kernel<NXB, 3>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles32 = ((q.rows + 31) / 32) * q.B;
  kern<<<(unsigned)nlam_bwd_grid(ntiles32), K16_THREADS, lds, s>>>(q);
  NLAM_CHECK_LAUNCH("outer_bwd16_kernel");
  return 0;
}

int nlam_k16_outer_bwd(const OuterParams& q, hipStream_t s) {
  if (!nlam_k16_on(K16_OUTER_BWD) || !nlam_mfma_b3()) return -1;
  const int kx = q.xa.width + (q.xb.ptr ? q.xb.width : 0);
  if (q.g.width != 64 || q.xa.width % 4 != 0) return -1;
  if (kx == 64) return launch_outer_bwd16<2>(q, s);
  if (kx == 128) return launch_outer_bwd16<4>(q, s);
  return -1;
}

// ================================================================== forward
// y = [res +] [LayerNorm](W2 silu(W1 [x_a | x_b] + b1) + b2): no row-contracting product, so no
// LDS besides the weight images and ~100 registers: four or more waves per SIMD.  Sources as in
// mlp_bwd16_kernel; NOB == 2: 64 outputs, 16-byte aligned (res, out); NOB == 1: any width <= 32.
template <int KB, int NOB, bool HAS_LN, int TERMS, bool N4 = false>
__device__ __forceinline__ void mlp_fwd16_body(const MlpParams& p, int wg, int nwg, char* smem16) {
  constexpr int HID = 64, NFH = 4, KF = 2 * KB, NFO = 2 * NOB, NO = 32 * NOB, KP32 = 32 * KB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  char* cur = smem16;
  const B3Image W1im = w16_image(cur, HID, KP32);
  cur += w16_image_bytes(HID, KP32);
  const B3Image W2im = w16_image(cur, NO, HID);
  cur += w16_image_bytes(NO, HID);
  float* b1s = reinterpret_cast<float*>(cur);
  float* b2s = b1s + HID;
  float* gs = b2s + NO;
  float* bs = gs + NO;
  {   // every global load of the prologue in flight together (fused16.h, batched prologue loads)
    VLoad16 lv;
    const float* const vecs[8] = {p.b1, p.b2, p.gamma, p.beta, nullptr, nullptr, nullptr, nullptr};
    const int lens[8] = {HID, p.n_out, p.n_out, p.n_out, 0, 0, 0, 0};
    WLoad16<KB> l1;
    WLoad16<NOB> l2;
    if constexpr (NO == 64) v16_issue(lv, vecs, lens, tid);
    w16_issue(l1, p.W1, p.ldW1, HID, p.k_in, HID, KP32, tid, K16_THREADS);
    w16_issue(l2, p.W2, p.ldW2, p.n_out, HID, NO, HID, tid, K16_THREADS);
    if constexpr (NO == 64) {
      v16_commit(lv, b1s, 4, tid);
    } else {
      load_vec_lds(b1s, p.b1, HID, HID, tid, K16_THREADS);
      load_vec_lds(b2s, p.b2, p.n_out, NO, tid, K16_THREADS);
      load_vec_lds(gs, p.gamma, p.n_out, NO, tid, K16_THREADS);
      load_vec_lds(bs, p.beta, p.n_out, NO, tid, K16_THREADS);
    }
    w16_commit(l1, W1im, 0, p.W1, p.ldW1, HID, p.k_in, HID, KP32, tid, K16_THREADS);
    w16_commit(l2, W2im, 0, p.W2, p.ldW2, p.n_out, HID, NO, HID, tid, K16_THREADS);
  }
  __syncthreads();
  const int64_t tiles_per_b = (p.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * p.B;
  for (int64_t tt = (int64_t)wg * K16_NW + wave; tt < ntiles; tt += (int64_t)nwg * K16_NW) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int nrows = (int)((p.rows - r0) < NLAM_T16 ? (p.rows - r0) : NLAM_T16);
    const bool valid = t < nrows;
    const int64_t row = r0 + (valid ? t : nrows - 1);
    f32x4 h[NFH];
    {
      f32x4 x[KF];
      const float* ra = p.src[0].ptr + b * p.src[0].bstride + row * p.src[0].ld;
      if constexpr (KB == 1) {
        load_narrow16<KF, N4>(x, ra, p.src[0].width, lane);
      } else if constexpr (KB == 4) {
        const float* rb = p.src[1].ptr + b * p.src[1].bstride + row * p.src[1].ld;
        load_cat16<KF>(x, ra, 64, rb, 64, lane);
      } else {
        load_cat16<KF>(x, ra, p.src[0].width, ra, 0, lane);
      }
      vec_to_acc16<NFH>(h, b1s, lane);
      gemm_acc16<NFH, KB, TERMS>(h, W1im, 0, 0, x, lane);
    }
#pragma unroll
    for (int fb = 0; fb < NFH; ++fb)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[fb][r] = nlam_silu(h[fb][r]);
    f32x4 y[NFO];
    vec_to_acc16<NFO>(y, b2s, lane);
    gemm_acc16<NFO, 2, TERMS>(y, W2im, 0, 0, h, lane);
    if constexpr (HAS_LN) ln16_apply<NFO>(y, gs, bs, lane);
    float* orow = p.out + b * p.out_bstride + (r0 + t) * p.out_ld;
    if constexpr (NOB == 2) {
      if (p.res != nullptr) {
        f32x4 rv[NFO];
        load_row16<NFO>(rv, p.res + b * p.res_bstride + row * p.res_ld, lane);
#pragma unroll
        for (int fb = 0; fb < NFO; ++fb) y[fb] += rv[fb];
      }
      if (valid) store_row16<NFO>(orow, y, lane);
    } else {
      if (p.res != nullptr) {
        f32x4 rv[NFO];
        load_narrow16<NFO>(rv, p.res + b * p.res_bstride + row * p.res_ld, p.n_out, lane);
#pragma unroll
        for (int fb = 0; fb < NFO; ++fb) y[fb] += rv[fb];
      }
      if (valid) store_row16_s<NFO>(orow, y, p.n_out, lane);
    }
  }
}

template <int KB, int NOB, bool HAS_LN, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void mlp_fwd16_kernel(MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  mlp_fwd16_body<KB, NOB, HAS_LN, TERMS>(p, (int)blockIdx.x, (int)gridDim.x, smem16);
}

struct MlpFwdMulti {
  MlpParams p[K16_MAXM];
  int n;
  int wg0[K16_MAXM + 1];
};
template <int TERMS, bool N4>
__global__ __launch_bounds__(K16_THREADS, 2) void mlp_fwd16_multi_kernel(MlpFwdMulti m) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  int k = 0;
  while (k + 1 < m.n && (int)blockIdx.x >= m.wg0[k + 1]) ++k;
  mlp_fwd16_body<1, 2, true, TERMS, N4>(m.p[k], (int)blockIdx.x - m.wg0[k], m.wg0[k + 1] - m.wg0[k],
                                        smem16);
}

template <int KB, int NOB, bool HAS_LN>
static int launch_mlp_fwd16(const MlpParams& p, hipStream_t s) {
  constexpr int HID = 64, NO = 32 * NOB, KP32 = 32 * KB;
  const size_t lds = w16_image_bytes(HID, KP32) + w16_image_bytes(NO, HID) + (HID + 3 * NO) * sizeof(float);
  auto kern = mlp_fwd16_kernel<KB, NOB, HAS_LN, 3>;
  NLAM_BIG_LDS(kern, __func__);
  const int64_t ntiles = ((p.rows + NLAM_T16 - 1) / NLAM_T16) * p.B;
  int64_t g = (ntiles + K16_NW - 1) / K16_NW;
  if (g > 512) g = 512;   // two 512-thread workgroups per CU (<= 128 registers, <= 54 KB of LDS)
  kern<<<(unsigned)g, K16_THREADS, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("mlp_fwd16_kernel");
  return 0;
}

int nlam_k16_mlp_fwd(const MlpParams& p, hipStream_t s) {
  if (!nlam_k16_on(K16_MLP_FWD) || !nlam_mfma_b3()) return -1;
  const bool ln = p.gamma != nullptr;
  const int kb = (p.k_in + 31) / 32;
  const bool a_vec = (p.vec_mask & 1) != 0, out_vec = (p.vec_mask & 8) != 0;
  const bool res_ok = p.res == nullptr || (p.vec_mask & 4) != 0;
  if (ln) {
    if (p.n_out != 64 || !out_vec || !res_ok) return -1;
    if (p.nsrc == 1 && kb == 1) return launch_mlp_fwd16<1, 2, true>(p, s);
    if (p.nsrc == 1 && kb == 2 && a_vec) return launch_mlp_fwd16<2, 2, true>(p, s);
    if (p.nsrc == 2 && kb == 4 && a_vec && (p.vec_mask & 2) && p.src[0].width == 64 &&
        p.src[1].width == 64)
      return launch_mlp_fwd16<4, 2, true>(p, s);
    return -1;
  }
  if (p.nsrc == 1 && kb == 2 && a_vec && p.n_out <= 32) return launch_mlp_fwd16<2, 1, false>(p, s);
  return -1;
}

// ================================================================ projection
// out[:, 0:nA] = x WA^T + bA ; out[:, nA:nA+nB] = x WB^T + bB: k_in = 64, n_out = 32 NOB in {64, 128}
template <int NOB, int TERMS>
__device__ __forceinline__ void lin_fwd16_body(const LinParams& p, int wg, int nwg, char* smem16) {
  constexpr int K = 64, KF = 4, NO = 32 * NOB, NFO = 2 * NOB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = lane & 15;
  const B3Image Wim = w16_image(smem16, NO, K);
  float* bs = reinterpret_cast<float*>(smem16 + w16_image_bytes(NO, K));
  {
    WLoad16<NOB> la, lb;
    w16_issue(la, p.WA, p.ldWA, p.nA, K, p.nA, K, tid, K16_THREADS);
    if (p.nB > 0) w16_issue(lb, p.WB, p.ldWB, p.nB, K, NO - p.nA, K, tid, K16_THREADS);
    load_vec_lds(bs, p.bA, p.nA, p.nA, tid, K16_THREADS);
    if (p.nB > 0) load_vec_lds(bs + p.nA, p.bB, p.nB, NO - p.nA, tid, K16_THREADS);
    w16_commit(la, Wim, 0, p.WA, p.ldWA, p.nA, K, p.nA, K, tid, K16_THREADS);
    if (p.nB > 0) w16_commit(lb, Wim, p.nA, p.WB, p.ldWB, p.nB, K, NO - p.nA, K, tid, K16_THREADS);
  }
  __syncthreads();
  const int64_t tiles_per_b = (p.rows + NLAM_T16 - 1) / NLAM_T16;
  const int64_t ntiles = tiles_per_b * p.B;
  for (int64_t tt = (int64_t)wg * K16_NW + wave; tt < ntiles; tt += (int64_t)nwg * K16_NW) {
    const int64_t b = tt / tiles_per_b;
    const int64_t r0 = (tt - b * tiles_per_b) * NLAM_T16;
    const int nrows = (int)((p.rows - r0) < NLAM_T16 ? (p.rows - r0) : NLAM_T16);
    const bool valid = t < nrows;
    const int64_t row = r0 + (valid ? t : nrows - 1);
    f32x4 x[KF], y[NFO];
    load_row16<KF>(x, p.x.ptr + b * p.x.bstride + row * p.x.ld, lane);
    vec_to_acc16<NFO>(y, bs, lane);
    gemm_acc16<NFO, 2, TERMS>(y, Wim, 0, 0, x, lane);
    if (valid) store_row16<NFO>(p.out + b * p.out_bstride + (r0 + t) * p.out_ld, y, lane);
  }
}

template <int NOB, int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void lin_fwd16_kernel(LinParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  lin_fwd16_body<NOB, TERMS>(p, (int)blockIdx.x, (int)gridDim.x, smem16);
}

struct LinFwdMulti {
  LinParams p[K16_MAXP];
  int n;
  int wg0[K16_MAXP + 1];
};
template <int TERMS>
__global__ __launch_bounds__(K16_THREADS, 2) void lin_fwd16_multi_kernel(LinFwdMulti m) {
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  int k = 0;
  while (k + 1 < m.n && (int)blockIdx.x >= m.wg0[k + 1]) ++k;
  lin_fwd16_body<2, TERMS>(m.p[k], (int)blockIdx.x - m.wg0[k], m.wg0[k + 1] - m.wg0[k], smem16);
}

static int64_t lin_fwd16_grid(const LinParams& p) {
  const int64_t ntiles = ((p.rows + NLAM_T16 - 1) / NLAM_T16) * p.B;
  int64_t g = (ntiles + K16_NW - 1) / K16_NW;
  if (g > 512) g = 512;
  return g < 1 ? 1 : g;
}

template <int NOB>
static int launch_lin_fwd16(const LinParams& p, hipStream_t s) {
  constexpr int K = 64, NO = 32 * NOB;
  const size_t lds = w16_image_bytes(NO, K) + NO * sizeof(float);
  auto kern = lin_fwd16_kernel<NOB, 3>;
  NLAM_BIG_LDS(kern, __func__);
  kern<<<(unsigned)lin_fwd16_grid(p), K16_THREADS, lds, s>>>(p);
  NLAM_CHECK_LAUNCH("lin_fwd16_kernel");
  return 0;
}

static bool lin_fwd16_ok(const LinParams& p) {
  return p.x.width == 64 && (p.vec_mask & 1) && (p.vec_mask & 8);
}

int nlam_k16_lin_fwd(const LinParams& p, hipStream_t s) {
  if (!nlam_k16_on(K16_LIN_FWD) || !nlam_mfma_b3()) return -1;
  if (!lin_fwd16_ok(p)) return -1;
  const int n_out = p.nA + p.nB;
  if (n_out == 64) return launch_lin_fwd16<2>(p, s);
  if (n_out == 128) return launch_lin_fwd16<4>(p, s);
  return -1;
}

int nlam_k16_lin_fwd_multi(const LinParams* p, int n, hipStream_t s) {
  if (!nlam_k16_on(K16_LIN_FWD) || !nlam_mfma_b3() || n < 1 || n > K16_MAXP) return -1;
  LinFwdMulti m;
  m.n = n;
  m.wg0[0] = 0;
  for (int k = 0; k < n; ++k) {
    if (!lin_fwd16_ok(p[k]) || p[k].nA + p[k].nB != 64) return -1;
    m.p[k] = p[k];
    m.wg0[k + 1] = m.wg0[k] + (int)lin_fwd16_grid(p[k]);
  }
  for (int k = n; k < K16_MAXP; ++k) m.wg0[k + 1] = m.wg0[n];
  const size_t lds = w16_image_bytes(64, 64) + 64 * sizeof(float);
  auto kern = lin_fwd16_multi_kernel<3>;
  NLAM_BIG_LDS(kern, __func__);
  kern<<<(unsigned)m.wg0[n], K16_THREADS, lds, s>>>(m);
  NLAM_CHECK_LAUNCH("lin_fwd16_multi_kernel");
  return 0;
}

// ---- multi-problem embedder MLPs (KB = 1: one source of <= 32 columns, hidden = n_out = 64, LayerNorm)
static bool mlp16_embedder_shape(const MlpParams& p) {
  return p.nsrc == 1 && p.k_in >= 1 && p.k_in <= 32 && p.n_out == 64 && p.gamma != nullptr &&
         p.res == nullptr && (p.vec_mask & 8) != 0;
}

int nlam_k16_mlp_fwd_multi(const MlpParams* p, int n, hipStream_t s) {
  if (!nlam_k16_on(K16_MLP_FWD) || !nlam_mfma_b3() || n < 1 || n > K16_MAXM) return -1;
  MlpFwdMulti m;
  m.n = n;
  m.wg0[0] = 0;
  for (int k = 0; k < n; ++k) {
    if (!mlp16_embedder_shape(p[k])) return -1;
    m.p[k] = p[k];
    const int64_t ntiles = ((p[k].rows + NLAM_T16 - 1) / NLAM_T16) * p[k].B;
    int64_t g = (ntiles + K16_NW - 1) / K16_NW;
    if (g > 512) g = 512;
    m.wg0[k + 1] = m.wg0[k] + (int)(g < 1 ? 1 : g);
  }
  for (int k = n; k < K16_MAXM; ++k) m.wg0[k + 1] = m.wg0[n];
  const size_t lds = w16_image_bytes(64, 32) + w16_image_bytes(64, 64) + (64 + 3 * 64) * sizeof(float);
  bool n4 = true;   // every problem at most 4 columns wide (the static features are 2-3)
  for (int k = 0; k < n; ++k) n4 = n4 && p[k].src[0].width <= 4;
  if (n4) {
    auto kern = mlp_fwd16_multi_kernel<3, true>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)m.wg0[n], K16_THREADS, lds, s>>>(m);
  } else {
    auto kern = mlp_fwd16_multi_kernel<3, false>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)m.wg0[n], K16_THREADS, lds, s>>>(m);
  }
  NLAM_CHECK_LAUNCH("mlp_fwd16_multi_kernel");
  return 0;
}

int nlam_k16_mlp_bwd_multi(const MlpBwdParams* q, int n, hipStream_t s) {
  if (!nlam_k16_on(K16_MLP_BWD) || !nlam_mfma_b3() || n < 1 || n > K16_MAXM) return -1;
  MlpBwdMulti m;
  m.n = n;
  m.wg0[0] = 0;
  for (int k = 0; k < n; ++k) {
    if (!mlp16_embedder_shape(q[k].f) || !q[k].vec_gy || q[k].gxb != nullptr || q[k].ga_out != nullptr ||
        q[k].add_gy_to_gxa || q[k].stamp)
      return -1;
    m.q[k] = q[k];
    m.wg0[k + 1] = m.wg0[k] + (int)nlam_bwd_grid(((q[k].f.rows + 31) / 32) * q[k].f.B);
  }
  for (int k = n; k < K16_MAXM; ++k) m.wg0[k + 1] = m.wg0[n];
  constexpr int HID = 64, NO = 64, KP32 = 32;
  size_t lds = w16_image_bytes(HID, KP32, true) + w16_image_bytes(NO, HID, true) +
               (HID + 2 * NO) * sizeof(float) +
               K16_NW * (p16_bytes(HID) + p16_bytes(HID) + (size_t)NLAM_T16 * (HID + 4) * sizeof(float));
  const size_t fold = (size_t)K16_NW * HID * HID * sizeof(float);
  if (fold > lds) lds = fold;
  NLAM_REQUIRE(lds <= 160 * 1024, "mlp_bwd16_multi: LDS footprint %zu B exceeds 160 KiB", lds);
  bool n4 = true;
  for (int k = 0; k < n; ++k) n4 = n4 && q[k].f.src[0].width <= 4;
  if (n4) {
    auto kern = mlp_bwd16_multi_kernel<3, true>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)m.wg0[n], K16_THREADS, lds, s>>>(m);
  } else {
    auto kern = mlp_bwd16_multi_kernel<3, false>;
    NLAM_BIG_LDS(kern, __func__);
    kern<<<(unsigned)m.wg0[n], K16_THREADS, lds, s>>>(m);
  }
  NLAM_CHECK_LAUNCH("mlp_bwd16_multi_kernel");
  return 0;
}

// ==================================================================== C entry points
extern "C" int nlam_lin_multi_supported(void) {
  return nlam_mfma_b3() && nlam_k16_on(K16_LIN_FWD) && nlam_k16_on(K16_LIN_BWD) &&
                 nlam_k16_on(K16_OUTER_BWD) ? 1 : 0;
}

// the d = 64 branch of nlam_lin_fwd_multi (fused_wide.hip)
int nlam_k16_lin_fwd_multi_c(int n, const float* const* x, const int64_t* x_bstride,
                             const int64_t* x_ld, const float* const* W, const int64_t* ldW,
                             const float* const* bias, float* const* out,
                             const int64_t* out_bstride, const int64_t* out_ld, const int64_t* B,
                             const int64_t* rows, void* stream) {
  NLAM_REQUIRE(n >= 1 && n <= K16_MAXP, "nlam_lin_fwd_multi: n %d out of [1, %d] at width 64", n, K16_MAXP);
  LinParams p[K16_MAXP];
  int m = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    NLAM_REQUIRE(view_vec_ok(x[k], x_bstride[k], x_ld[k], 64) &&
                     view_vec_ok(out[k], out_bstride[k], out_ld[k], 64),
                 "nlam_lin_fwd_multi: operand rows must be 16-byte aligned, width 64");
    LinParams& q = p[m++];
    q.x = RowView{x[k], x_bstride[k], x_ld[k], 64};
    q.k_pad = 64;
    q.WA = W[k]; q.ldWA = ldW[k]; q.bA = bias[k]; q.nA = 64;
    q.WB = nullptr; q.ldWB = 0; q.bB = nullptr; q.nB = 0;
    q.out = out[k]; q.out_bstride = out_bstride[k]; q.out_ld = out_ld[k];
    q.rows = rows[k]; q.B = (int)B[k]; q.vec_mask = 9; q.timeline = 0;
  }
  if (m == 0) return 0;
  const int r = nlam_k16_lin_fwd_multi(p, m, (hipStream_t)stream);
  NLAM_REQUIRE(r >= 0, "nlam_lin_fwd_multi: width 64 needs the split-bf16 16-row kernels "
                       "(nlam_lin_multi_supported())");
  return r;
}

extern "C" int nlam_lin_bwd_multi(
    int n, int d, const float* const* x, const int64_t* x_bstride, const int64_t* x_ld,
    const float* const* xb, const int64_t* xb_bstride, const int64_t* xb_ld,
    const float* const* gy, const int64_t* gy_bstride, const int64_t* gy_ld, const float* const* W,
    const int64_t* ldW, float* const* gx, const int64_t* gx_bstride, const int64_t* gx_ld,
    const float* const* gx_add, const int64_t* ga_bstride, const int64_t* ga_ld,
    const int64_t* gy_nsum, const int64_t* gy_sum_stride, const float* const* gh,
    const int64_t* gh_bstride, const int32_t* const* csc_colptr, const int32_t* const* csc_eid,
    const int64_t* n_send, float* const* slab, const int64_t* slab_stride, const int64_t* B,
    const int64_t* rows, void* stream) {
  NLAM_REQUIRE(d == 64, "nlam_lin_bwd_multi: width %d unsupported (64)", d);
  NLAM_REQUIRE(n >= 1 && n <= K16_MAXP, "nlam_lin_bwd_multi: n %d out of [1, %d]", n, K16_MAXP);
  LinBwdParams q[K16_MAXP];
  OuterParams o[K16_MAXP];
  int kind[K16_MAXP];
  int m = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    LinBwdParams& r = q[m];
    OuterParams& ou = o[m];
    kind[m] = xb[k] != nullptr ? 1 : 0;
    ++m;
    NLAM_REQUIRE(x[k] && slab[k] && (gy[k] || gh[k]), "nlam_lin_bwd_multi: NULL operand %d", k);
    if (xb[k] != nullptr) {   // deferred first-layer weight gradient: dW (64 x 128) = gy^T [x | xb]
      NLAM_REQUIRE(W[k] == nullptr && gx[k] == nullptr && gh[k] == nullptr && gy_nsum[k] <= 1,
                   "nlam_lin_bwd_multi: problem %d with xb forms weight gradients only", k);
      NLAM_REQUIRE(slab_stride[k] >= 64 * 128 + 64, "nlam_lin_bwd_multi: slab %d too small", k);
      NLAM_REQUIRE(view_vec_ok(x[k], x_bstride[k], x_ld[k], 64) &&
                       view_vec_ok(xb[k], xb_bstride[k], xb_ld[k], 64) &&
                       view_vec_ok(gy[k], gy_bstride[k], gy_ld[k], 64),
                   "nlam_lin_bwd_multi: operand rows of problem %d must be 16-byte aligned, width 64", k);
      ou.g = RowView{gy[k], gy_bstride[k], gy_ld[k], 64};
      ou.xa = RowView{x[k], x_bstride[k], x_ld[k], 64};
      ou.xb = RowView{xb[k], xb_bstride[k], xb_ld[k], 64};
      ou.x_index = nullptr; ou.slab = slab[k]; ou.slab_stride = slab_stride[k];
      ou.rows = rows[k]; ou.B = (int)B[k];
      r = LinBwdParams{};
      continue;
    }
    ou = OuterParams{};
    NLAM_REQUIRE(W[k] != nullptr, "nlam_lin_bwd_multi: NULL weight %d", k);
    NLAM_REQUIRE(slab_stride[k] >= 64 * 64 + 64, "nlam_lin_bwd_multi: slab %d too small", k);
    r.x = RowView{x[k], x_bstride[k], x_ld[k], 64};
    r.gy = RowView{gy[k], gy_bstride[k], gy_ld[k], 64};
    r.WA = W[k]; r.ldWA = ldW[k]; r.nA = 64; r.WB = nullptr; r.ldWB = 0; r.nB = 0;
    r.gx = gx[k]; r.gx_bstride = gx_bstride[k]; r.gx_ld = gx_ld[k];
    r.gx_add = gx_add[k]; r.ga_bstride = ga_bstride[k]; r.ga_ld = ga_ld[k];
    r.slab = slab[k]; r.slab_stride = slab_stride[k];
    r.rows = rows[k]; r.B = (int)B[k];
    r.gy_nsum = gy_nsum[k] > 1 ? (int)gy_nsum[k] : 1;
    r.gy_sum_stride = gy_sum_stride[k];
    r.gh = gh[k]; r.gh_bstride = gh_bstride[k];
    r.csc_colptr = csc_colptr[k]; r.csc_eid = csc_eid[k]; r.n_send = (int)n_send[k];
    r.vec_x = view_vec_ok(x[k], x_bstride[k], x_ld[k], 64);
    r.vec_gy = gh[k] ? 1 : view_vec_ok(gy[k], gy_bstride[k], gy_ld[k], 64);
    r.vec_gx = gx[k] == nullptr ||
               (view_vec_ok(gx[k], gx_bstride[k], gx_ld[k], 64) &&
                (gx_add[k] == nullptr || view_vec_ok(gx_add[k], ga_bstride[k], ga_ld[k], 64)));
    NLAM_REQUIRE(r.vec_x && r.vec_gy && r.vec_gx && (r.gy_nsum == 1 || gy_sum_stride[k] % 4 == 0 || gh[k]),
                 "nlam_lin_bwd_multi: operand rows of problem %d must be 16-byte aligned, width 64", k);
    if (gh[k]) {
      NLAM_REQUIRE(csc_colptr[k] && csc_eid[k] && nlam_aligned16(gh[k]) && gh_bstride[k] % 4 == 0 &&
                       n_send[k] >= 0 && n_send[k] <= rows[k],
                   "nlam_lin_bwd_multi: gather operands of problem %d", k);
    }
    NLAM_REQUIRE(gx_add[k] == nullptr || gx[k] != nullptr, "nlam_lin_bwd_multi: gx_add without gx");
  }
  if (m == 0) return 0;
  const int r = nlam_k16_lin_bwd_multi(q, o, kind, m, (hipStream_t)stream);
  NLAM_REQUIRE(r >= 0, "nlam_lin_bwd_multi: needs the split-bf16 16-row kernels (nlam_lin_multi_supported())");
  return r;
}

extern "C" int nlam_mlp_multi_supported(void) {
  return nlam_mfma_b3() && nlam_k16_on(K16_MLP_FWD) && nlam_k16_on(K16_MLP_BWD) ? 1 : 0;
}

extern "C" int nlam_mlp_fwd_multi(int n, const float* const* x, const int64_t* x_bstride,
                                  const int64_t* x_ld, const int32_t* x_width, const float* const* W1,
                                  const int64_t* ldW1, const float* const* b1, const float* const* W2,
                                  const int64_t* ldW2, const float* const* b2,
                                  const float* const* gamma, const float* const* beta,
                                  float* const* out, const int64_t* out_bstride, const int64_t* out_ld,
                                  const int64_t* B, const int64_t* rows, int hid, int n_out,
                                  void* stream) {
  NLAM_REQUIRE(hid == 64 && n_out == 64, "nlam_mlp_fwd_multi: hidden / output width 64 only");
  NLAM_REQUIRE(n >= 1 && n <= K16_MAXM, "nlam_mlp_fwd_multi: n %d out of [1, %d]", n, K16_MAXM);
  MlpParams p[K16_MAXM];
  int m = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    NLAM_REQUIRE(x[k] && W1[k] && W2[k] && gamma[k] && beta[k] && out[k], "nlam_mlp_fwd_multi: NULL operand %d", k);
    NLAM_REQUIRE(x_width[k] >= 1 && x_width[k] <= 32, "nlam_mlp_fwd_multi: input width %d out of [1, 32]",
                 x_width[k]);
    MlpParams& q = p[m++];
    q.src[0] = RowView{x[k], x_bstride[k], x_ld[k], x_width[k]};
    q.src[1] = RowView{nullptr, 0, 0, 0};
    q.nsrc = 1; q.k_in = x_width[k]; q.k_pad = (x_width[k] + 7) & ~7; q.n_out = 64;
    q.W1 = W1[k]; q.ldW1 = ldW1[k]; q.b1 = b1[k]; q.W2 = W2[k]; q.ldW2 = ldW2[k]; q.b2 = b2[k];
    q.gamma = gamma[k]; q.beta = beta[k];
    q.res = nullptr; q.res_bstride = 0; q.res_ld = 0;
    q.out = out[k]; q.out_bstride = out_bstride[k]; q.out_ld = out_ld[k];
    q.rows = rows[k]; q.B = (int)B[k];
    q.vec_mask = view_vec_ok(out[k], out_bstride[k], out_ld[k], 64) ? 8 : 0;
    NLAM_REQUIRE(q.vec_mask & 8, "nlam_mlp_fwd_multi: output rows of problem %d must be 16-byte aligned", k);
  }
  if (m == 0) return 0;
  const int r = nlam_k16_mlp_fwd_multi(p, m, (hipStream_t)stream);
  NLAM_REQUIRE(r >= 0, "nlam_mlp_fwd_multi: needs the split-bf16 16-row kernels (nlam_mlp_multi_supported())");
  return r;
}

extern "C" int nlam_mlp_bwd_multi(int n, const float* const* x, const int64_t* x_bstride,
                                  const int64_t* x_ld, const int32_t* x_width, const float* const* W1,
                                  const int64_t* ldW1, const float* const* b1, const float* const* W2,
                                  const int64_t* ldW2, const float* const* b2,
                                  const float* const* gamma, const float* const* gy,
                                  const int64_t* gy_bstride, const int64_t* gy_ld, float* const* gx,
                                  const int64_t* gx_bstride, const int64_t* gx_ld, float* const* slab,
                                  const int64_t* slab_stride, const int64_t* B, const int64_t* rows,
                                  int hid, int n_out, void* stream) {
  NLAM_REQUIRE(hid == 64 && n_out == 64, "nlam_mlp_bwd_multi: hidden / output width 64 only");
  NLAM_REQUIRE(n >= 1 && n <= K16_MAXM, "nlam_mlp_bwd_multi: n %d out of [1, %d]", n, K16_MAXM);
  MlpBwdParams q[K16_MAXM];
  int m = 0;
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0 || rows[k] <= 0) continue;
    NLAM_REQUIRE(x[k] && W1[k] && W2[k] && gamma[k] && gy[k] && slab[k], "nlam_mlp_bwd_multi: NULL operand %d", k);
    NLAM_REQUIRE(x_width[k] >= 1 && x_width[k] <= 32, "nlam_mlp_bwd_multi: input width %d out of [1, 32]",
                 x_width[k]);
    NLAM_REQUIRE(slab_stride[k] >= nlam_mlp_bwd_slab_stride(x_width[k], 64, 64),
                 "nlam_mlp_bwd_multi: slab %d too small", k);
    MlpBwdParams& r = q[m++];
    MlpParams& p = r.f;
    p.src[0] = RowView{x[k], x_bstride[k], x_ld[k], x_width[k]};
    p.src[1] = RowView{nullptr, 0, 0, 0};
    p.nsrc = 1; p.k_in = x_width[k]; p.k_pad = (x_width[k] + 7) & ~7; p.n_out = 64;
    p.W1 = W1[k]; p.ldW1 = ldW1[k]; p.b1 = b1[k]; p.W2 = W2[k]; p.ldW2 = ldW2[k]; p.b2 = b2[k];
    p.gamma = gamma[k]; p.beta = nullptr;
    p.res = nullptr; p.res_bstride = 0; p.res_ld = 0; p.out = nullptr; p.out_bstride = 0; p.out_ld = 0;
    p.rows = rows[k]; p.B = (int)B[k]; p.vec_mask = 8;
    r.gy = RowView{gy[k], gy_bstride[k], gy_ld[k], 64};
    r.gxa = gx[k]; r.gxa_bstride = gx_bstride[k]; r.gxa_ld = gx_ld[k];
    r.gxb = nullptr; r.gxb_bstride = 0; r.gxb_ld = 0; r.add_gy_to_gxa = 0;
    r.slab = slab[k]; r.slab_stride = slab_stride[k]; r.ga_out = nullptr;
    r.vec_gy = view_vec_ok(gy[k], gy_bstride[k], gy_ld[k], 64);
    r.vec_gxa = 0; r.vec_gxb = 0; r.stamp = 0;
    NLAM_REQUIRE(r.vec_gy, "nlam_mlp_bwd_multi: gy rows of problem %d must be 16-byte aligned", k);
  }
  if (m == 0) return 0;
  const int r = nlam_k16_mlp_bwd_multi(q, m, (hipStream_t)stream);
  NLAM_REQUIRE(r >= 0, "nlam_mlp_bwd_multi: needs the split-bf16 16-row kernels (nlam_mlp_multi_supported())");
  return r;
}
