// Edge backward of one InteractionNet layer at hidden 64, split-bf16 arithmetic: the round-4
// rewrite of edge_bwd_kernel<64, *, *, true> (fused_edge.hip; reference semantics
// interaction_net.py:117-131, backward of the edge MLP + LayerNorm + receiver aggregation).
//
// Same data flow, layout (32 rows per wavefront, row-on-lane accumulators, fused_common.h), LDS
// images, slab layout and grid as the kernel it replaces; what changed is the instruction stream
// (the old kernel was bound by instruction issue along one dependent chain per tile, with one
// wave per SIMD -- DESIGN.md 4.7):
//   * MASK-FREE tiles: padded slots (a tile holds <= 32 edges) replicate the tile's LAST edge --
//     same gathered rows, and stores that write the same bytes to the same address -- and only
//     the upstream gradient gm is zeroed for them (one multiply by a 0/1 lane constant that also
//     carries the 1/deg scale).  gm = 0 makes gz, gh and every weight-gradient contribution of a
//     padded slot exactly zero, so no other select is needed (the old kernel: 270 v_cndmask and
//     ~30 exec-mask regions per tile).
//   * 32-bit row offsets on scalar bases: one v_mad_u32_u24 per row access instead of a 64-bit
//     multiply-add chain (the old kernel: ~350 integer VALU per tile, a third of them quarter rate).
//   * the row gathers of tile n+1 (and the row stores of tile n) are issued INSIDE the MFMA
//     phases of tile n, two per K step, instead of one burst of 40 loads at the end of the tile
//     (20 % of the old tile time was spent issuing that burst, 10 % waiting for it); the slot
//     indices run two tiles ahead so that no address depends on a load of the same tile.
//   * gamma / beta gradients: row-group partial sums (8 ds_read_b128 + 16 v_pk_add per tile)
//     into four per-lane accumulators instead of a 32-step column walk; db2 as a ones-vector
//     MFMA over the gz planes that the weight-gradient product needs anyway.
//   * silu and silu' share one sigmoid; sg = silu'(h) is kept instead of h.
//
// Registers.  A wave of this kernel owns the whole 512-entry file, but VALU operands must sit in
// the 256 architectural VGPRs.  With the compiler's default choice every MFMA of a 512-register
// kernel writes AGPRs, so each GEMM result cost 32 v_accvgpr_read before the first VALU touched
// it, and the allocator shuffled ~400 values per tile between the halves (old kernel: 398
// v_accvgpr_* of 2,183 VALU per tile).  Here the two persistent weight-gradient accumulators
// (128 registers, touched by nothing but their MFMAs) are pinned to AGPRs through tied inline-asm
// MFMAs, and everything else is compiled in the VGPR form (the flag below, read by build.py).
// NLAM_HIPCC_FLAGS: -mllvm -amdgpu-mfma-vgpr-form
#include <stdlib.h>

#include "fused_common.h"
#include "fused_bf16x3.h"
#include "fused_params.h"

namespace {

constexpr int D = 64, NB = 2, LDT = D + 4;
constexpr int NVR = 8;   // wave-wide 16-byte loads per 32 x 64 row tile (4 rows each)
constexpr int NVC = 4;   // ... per 16 compact receiver rows

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) i32x4* const_i32x4_ptr;

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// global row access: scalar base + 32-bit byte offset (global_load_dwordx4 v, v_off, s[base])
__device__ __forceinline__ f32x4 ldg_off(const float* base, uint32_t off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ void stg_off(float* base, uint32_t off, const f32x4& v) {
  *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + off) = v;
}
__device__ __forceinline__ uint32_t row_off(int idx, uint32_t ld4, uint32_t col) {
  return __umul24((uint32_t)idx, ld4) + col;
}

struct Hdr { int p0, ne, r0, nr; };   // wave-uniform

// Slot indices of one tile (slot t = lane & 31; padded slots replicate the last edge) and the
// scale of its upstream aggregate gradient (1/deg for mean aggregation, 0 on padded slots).
struct Ctx {
  Hdr h;
  int b;            // batch item (wave-uniform)
  int eid, snd, rcv;
  float sc;
  int slot;         // PART: rank of the slot's sender among the tile's distinct senders
};

__device__ __forceinline__ Hdr load_hdr(const EdgeFwdParams& p, unsigned tile) {
  // (through the constant address space: one s_load_dwordx4; the kernel's stores could alias the
  // table as far as the compiler knows, so it would not use the scalar cache by itself)
  const i32x4 v = ((const_i32x4_ptr)(uintptr_t)p.tiles)[tile];
  Hdr h;
  h.p0 = v.x; h.ne = v.y - v.x; h.r0 = v.z; h.nr = v.w - v.z;
  return h;
}
template <bool PART = false>
__device__ __forceinline__ Ctx load_ctx(const EdgeFwdParams& p, const Hdr& h, int b, int lane,
                                        const int32_t* __restrict__ part_slot = nullptr) {
  Ctx c;
  c.h = h; c.b = b;
  const int t = lane & 31;
  // (a tile of receivers without in-edges has ne == 0: every slot then points at CSR position 0,
  // whose rows exist; gm = 0 on every slot and its row stores go to a scratch row, see the loop)
  const int last = h.ne > 0 ? h.ne - 1 : 0;
  const int pos = (h.ne > 0 ? h.p0 : 0) + (t < last ? t : last);
  c.eid = p.csr_eid[pos];
  c.snd = p.csr_send[pos];
  c.rcv = p.csr_rec[pos];
  c.sc = 1.0f;   // (mean aggregation: 1 / deg follows one tile later, load_scale)
  c.slot = 0;
  if constexpr (PART) c.slot = part_slot[pos];
  return c;
}
// 1 / deg of the slots' receivers, requested a tile after the receiver ids themselves so that no
// load of the tile loop depends on another load of the same tile
__device__ __forceinline__ float load_scale(const EdgeFwdParams& p, const Ctx& c) {
  return p.inv_deg ? p.inv_deg[c.rcv] : 1.0f;
}

// ---- LDS tiles ------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void put_rows(float* __restrict__ tile, const f32x4 (&v)[NV], int sub, int c4) {
#pragma unroll
  for (int k = 0; k < NV; ++k)
    *reinterpret_cast<f32x4*>(tile + (sub + 4 * k) * LDT + 4 * c4) = v[k];
}
template <int NV>
__device__ __forceinline__ void put_rows_planes(const B3Tile& T, const f32x4 (&v)[NV], int sub, int c4) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    bf16x4 hi, lo;
    b3_split4(v[k], hi, lo);
    const int off = (sub + 4 * k) * T.pitch + 4 * c4;
    *reinterpret_cast<bf16x4*>(T.hi + off) = hi;
    *reinterpret_cast<bf16x4*>(T.lo + off) = lo;
  }
}
// accumulator layout <- tile row `row` (this lane's slot row, or its receiver's compact row)
template <bool ADD>
__device__ __forceinline__ void row_to_acc(f32x16 (&acc)[NB], const float* __restrict__ tile, int row,
                                           int hh) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * LDT + 32 * nb + 8 * q + 4 * hh);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ADD) acc[nb][4 * q + j] += v[j];
        else acc[nb][4 * q + j] = v[j];
      }
    }
}
// acc4 += sum over this lane's row group (rows sub, sub + 4, ...) of tile[row][4 c4 .. 4 c4 + 3]
__device__ __forceinline__ void colsum4(f32x4& acc4, const float* __restrict__ tile, int sub, int c4) {
#pragma unroll
  for (int k0 = 0; k0 < NVR; k0 += 4) {   // (four rows in flight per wait: 16 transient registers)
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v[k] = *reinterpret_cast<const f32x4*>(tile + (sub + 4 * (k0 + k)) * LDT + 4 * c4);
    acc4 += (v[0] + v[1]) + (v[2] + v[3]);
  }
}

// ---- GEMMs with a per-K-step hook ------------------------------------------------------------
// 24 MFMAs in four K steps of six; `between(step)` is called after the MFMAs of each step: the
// row traffic of the neighbouring tiles rides in their shadow.  The LDS fragments of step i + 1
// are requested BEFORE the MFMAs of step i are issued (two fragment sets in registers): with one
// wave per SIMD nothing else hides the ~130-cycle LDS latency, and a step that first reads and
// then multiplies took ~500 cycles for 192 cycles of matrix work.
struct FragsAB {
  bf16x8 ah[NB], al[NB], bh, bl;
};
__device__ __forceinline__ void mfma6(f32x16 (&out)[NB], const FragsAB& f) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    out[nb] = B3_MFMA(f.ah[nb], f.bh, out[nb]);
    out[nb] = B3_MFMA(f.ah[nb], f.bl, out[nb]);
    out[nb] = B3_MFMA(f.al[nb], f.bh, out[nb]);
  }
}
// out[nb] += W[rows 32 nb ..] . X^T, X a plane tile (K = 64: steps (kb, s))
template <typename F>
__device__ __forceinline__ void gemm_tile_cb(f32x16 (&out)[NB], const B3Image& W, const B3Tile& X,
                                             int lane, F between) {
  const int t = lane & 31, h = lane >> 5;
  auto load = [&](FragsAB& f, int i) {
    const int kb = i >> 1, s = i & 1;
    const int xo = t * X.pitch + 32 * kb + 16 * s + 4 * h;
    f.bh = b3_join(*reinterpret_cast<const bf16x4*>(X.hi + xo), *reinterpret_cast<const bf16x4*>(X.hi + xo + 8));
    f.bl = b3_join(*reinterpret_cast<const bf16x4*>(X.lo + xo), *reinterpret_cast<const bf16x4*>(X.lo + xo + 8));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      f.ah[nb] = b3_row_frag(W.hi, W.pitch, 32 * nb + t, kb, s, h);
      f.al[nb] = b3_row_frag(W.lo, W.pitch, 32 * nb + t, kb, s, h);
    }
  };
  FragsAB f[2];
  load(f[0], 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i + 1 < 4) load(f[(i + 1) & 1], i + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(out, f[i & 1]);
    between(i);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// out[kb] += W[:, 32 kb ..]^T . G, G a plane tile (gx = W^T gy; steps (nb, s) over W's rows)
template <typename F>
__device__ __forceinline__ void gemm_tile_wt_cb(f32x16 (&out)[NB], const B3Image& W, const B3Tile& G,
                                                int lane, F between) {
  const int t = lane & 31, h = lane >> 5;
  auto load = [&](FragsAB& f, int i) {
    const int nb = i >> 1, s = i & 1;
    const int go = t * G.pitch + 32 * nb + 16 * s + 4 * h;
    f.bh = b3_join(*reinterpret_cast<const bf16x4*>(G.hi + go), *reinterpret_cast<const bf16x4*>(G.hi + go + 8));
    f.bl = b3_join(*reinterpret_cast<const bf16x4*>(G.lo + go), *reinterpret_cast<const bf16x4*>(G.lo + go + 8));
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      f.ah[kb] = b3_tr_frag(W.hi, W.pitch, 32 * nb + 16 * s, 32 * kb, lane);
      f.al[kb] = b3_tr_frag(W.lo, W.pitch, 32 * nb + 16 * s, 32 * kb, lane);
    }
  };
  FragsAB f[2];
  load(f[0], 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i + 1 < 4) load(f[(i + 1) & 1], i + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma6(out, f[i & 1]);
    between(i);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// dW[ib][jb] += sum_rows G[:, 32 ib ..] (x) X[:, 32 jb ..]   (steps (u, jb): 16 rows, one X block)
template <typename F>
__device__ __forceinline__ void outer_cb(f32x16 (&dW)[NB][NB], const B3Tile& G, const B3Tile& X,
                                         int lane, F between) {
  auto load = [&](FragsAB& f, int i) {
    const int u = i >> 1, jb = i & 1;
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {   // (re-read for both jb of a u: two LDS reads cheaper than 16 live registers)
      f.ah[ib] = b3_tr_frag_rows(G.hi, G.pitch, 16 * u, 32 * ib, lane);
      f.al[ib] = b3_tr_frag_rows(G.lo, G.pitch, 16 * u, 32 * ib, lane);
    }
    f.bh = b3_tr_frag_rows(X.hi, X.pitch, 16 * u, 32 * jb, lane);
    f.bl = b3_tr_frag_rows(X.lo, X.pitch, 16 * u, 32 * jb, lane);
  };
  FragsAB f[2];
  load(f[0], 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i + 1 < 4) load(f[(i + 1) & 1], i + 1);
    __builtin_amdgcn_sched_barrier(0);
    const int jb = i & 1;
    const FragsAB& c = f[i & 1];
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
      // accumulate in place in AGPRs (srcC = vDst exactly: back-to-back issue is interlocked
      // by the hardware; the operands come from LDS reads the compiler waits for)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dW[ib][jb]) : "v"(c.ah[ib]), "v"(c.bh));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dW[ib][jb]) : "v"(c.ah[ib]), "v"(c.bl));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dW[ib][jb]) : "v"(c.al[ib]), "v"(c.bh));
    }
    between(i);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- receiver-side sums of gh on the matrix cores ----------------------------------------------
// gPr[i] = sum over the slots k of receiver i of gh[k] = (Ind . GH)[i], Ind[i][k] = (rec(k) == r0 + i):
// a 32 x 32 0/1 matrix (exact in bf16) times the GH planes (hi + lo), 8 MFMAs per tile.  Receivers
// without in-edges get exact zeros, padded slots contribute their (zero) rows; the order of the
// additions is fixed by the MFMA.  Replaces a 32-step walk over the rows with a scalar branch
// per row (1.8 k of 25 k cycles per tile on the m2m graph, 8 segments per tile on m2g).
constexpr int IND_PITCH = 40;   // bf16 elements (80-byte rows: the 16-byte fragment reads stay aligned)
__device__ __forceinline__ void ind_build(__bf16* __restrict__ ind, int roff, int lane) {
  // zero 32 x 40 bf16 = 2560 B: 160 x 16 B
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4* p4 = reinterpret_cast<f32x4*>(ind);
  p4[lane] = z4;
  p4[64 + lane] = z4;
  if (lane < 32) p4[128 + lane] = z4;
  wave_sync();
  if (lane < 32) ind[roff * IND_PITCH + lane] = (__bf16)1.0f;
}
__device__ __forceinline__ void seg_mfma(f32x16 (&seg)[NB], const __bf16* __restrict__ ind, const B3Tile& G,
                                         int lane) {
  const int i = lane & 31, kg = lane >> 5;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(ind + i * IND_PITCH + 16 * u + 8 * kg);
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
      const bf16x8 bh = b3_tr_frag_rows(G.hi, G.pitch, 16 * u, 32 * ib, lane);
      const bf16x8 bl = b3_tr_frag_rows(G.lo, G.pitch, 16 * u, 32 * ib, lane);
      if (u == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        seg[ib] = B3_MFMA(a, bh, z);
      } else {
        seg[ib] = B3_MFMA(a, bh, seg[ib]);
      }
      seg[ib] = B3_MFMA(a, bl, seg[ib]);
    }
  }
}
// seg (receiver i in the registers, feature in the lanes) -> rows of a [receiver][feature] fp32
// tile -> 256-byte row stores of gPr[r0 .. r0 + nr)
__device__ __forceinline__ void seg_store(const f32x16 (&seg)[NB], float* __restrict__ tile, int nr,
                                          float* __restrict__ gb, uint32_t row0_off, uint32_t ld4,
                                          int lane) {
  const int j = lane & 31, hh = lane >> 5, sub = lane >> 4, c4 = lane & 15;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i0 = 8 * (r >> 2) + (r & 3);
    if (i0 < nr) {   // wave-uniform (the h = 1 row i0 + 4 may lie past nr: harmless in LDS)
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) tile[(i0 + 4 * hh) * LDT + 32 * ib + j] = seg[ib][r];
    }
  }
  wave_sync();
  for (int k0 = 0; k0 < nr; k0 += 4) {   // wave-uniform trip count (1 on most tiles)
    const int i = k0 + sub;
    if (i < nr) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + i * LDT + 4 * c4);
      stg_off(gb, row0_off + (uint32_t)i * ld4 + 16u * (uint32_t)c4, v);
    }
  }
}

struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

}  // namespace

// Slab per workgroup (as edge_bwd_kernel): [dW1e (D x D) | dW2 (D x D) | db2 | dgamma | dbeta].
// Diagnostic build only (NLAM_STAMP2=1): per-phase cycle sums of the tile loop over all waves
// (s_memtime stamps; they perturb the schedule, read the SHARES).
__device__ unsigned long long g_edge_bwd2_stamps[8];
int nlam_edge_bwd2_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_edge_bwd2_stamps), sizeof(unsigned long long) * 8) != hipSuccess)
    return 1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_edge_bwd2_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#define STAMP2(k)                                                  \
  if (STAMP) {                                                     \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    __builtin_amdgcn_s_waitcnt(0xC07F);                            \
    __builtin_amdgcn_sched_barrier(0);                             \
    st[k] += now_ - tprev;                                         \
    tprev = now_;                                                  \
  }

// HAS_GEO: an upstream gradient on e' exists (every edge-updating layer but the last of a chain).
// ABL (diagnostic, NLAM_ABL2=1): every row access goes to row 0 / a scratch row -- the same
// instruction stream without HBM traffic (results are meaningless).
// BSUM (no edge update, batch-invariant first-layer edge term Pe, B > 1): a wave takes whole tiles
// and runs their B batch items back to back, sums gh over the batch in registers and writes
// dPe = sum_b gh[b] (1, M, d) to q.g_e -- the gradient of the batch-invariant operand, which the
// projection backward otherwise forms by reading gh B times (m2g: 261 MB of its 1,154 MB).
// PART (with BSUM): the gh rows are not written; each tile emits the sums of its gh rows per distinct
// sender instead (a second indicator product on the matrix cores, <= 16 rows per tile): what the
// sender-side reduction of the projection backward needs, in a third of the rows.
template <bool HAS_EGEMM, bool HAS_GEO, bool STAMP = false, bool ABL = false, bool BSUM = false,
          bool PART = false>
__global__ __launch_bounds__(256) void edge_bwd2_kernel(EdgeBwdParams q, int flags) {
  const int xcd_remap = flags & 1;
  const bool edge_binner_on = (flags & 2) != 0;
  static_assert(!BSUM || !HAS_EGEMM, "the batch sum is the no-edge-update form's extra output");
  static_assert(!PART || BSUM, "sender partials exist in the batch-sum form");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDW = D + 4;
  constexpr int WSTRIDE = 3 * NLAM_TILE * LDT;
  const EdgeFwdParams& p = q.f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  float* W1s = smem;
  float* W2s = W1s + (HAS_EGEMM ? D * LDW : 0);
  float* b2s = W2s + D * LDW;
  float* gs = b2s + D;
  float* T0base = gs + D;
  float* T0 = T0base + wave * WSTRIDE;
  float* T1 = T0 + NLAM_TILE * LDT;
  float* T2 = T1 + NLAM_TILE * LDT;
  // per-wave slot tables, double-buffered over tiles: [2][eid | send][32]
  int* itab = reinterpret_cast<int*>(T0base + 4 * WSTRIDE) + wave * (4 * NLAM_TILE);
  __bf16* ind = reinterpret_cast<__bf16*>(reinterpret_cast<int*>(T0base + 4 * WSTRIDE) + 4 * (4 * NLAM_TILE)) +
                wave * (NLAM_TILE * IND_PITCH);
  const B3Image W1im = b3_image(W1s, D, D), W2im = b3_image(W2s, D, D);
  const unsigned ntiles = (unsigned)p.ntiles;
  const unsigned total = ntiles * (unsigned)p.B;
  const unsigned stride = gridDim.x * 4;
  // XCD-aware order: workgroups are dispatched round-robin over the 8 XCDs (each with its own
  // L2), and neighbouring tiles share sender / receiver rows -- so XCD x takes a CONTIGUOUS eighth
  // of every round's tiles instead of every eighth tile (node rows are then fetched by one L2,
  // not by all eight)
  const unsigned G = gridDim.x;
  const unsigned wg = (xcd_remap && (G & 7u) == 0) ? (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  // Tasks of this wave, numbered j = 0, 1, ...: strided over the (batch item, tile) pairs, or --
  // BSUM -- a contiguous range of tiles with the batch items innermost.
  // (strided form) wave-major numbering: the waves that own a task of the last, partial round are
  // then wave 0 of as many workgroups -- one per CU -- instead of all four waves of a few CUs
  const unsigned Bu = (unsigned)p.B;
  // strided form with a batch-invariant edge operand (g2m): batch-inner task numbering
  const bool binner = !BSUM && !HAS_EGEMM && p.e.bstride == 0 && Bu > 1 && edge_binner_on;
  const unsigned wid = (!BSUM && !binner && (flags & 8)) ? (unsigned)wave * G + wg : wg * 4 + wave;
  unsigned tile0 = 0, niter;
  if (BSUM) {
    const unsigned per = (ntiles + stride - 1) / stride;          // tiles per wave
    tile0 = wid * per;
    const unsigned tile1 = tile0 + per < ntiles ? tile0 + per : ntiles;
    niter = tile1 > tile0 ? (tile1 - tile0) * Bu : 0;
  } else {
    niter = total > wid ? (total - wid + stride - 1) / stride : 0;
  }
  auto task_hdr = [&](unsigned j, int& b) {   // header + batch item of task j (clamped to the last)
    if (BSUM) {
      const unsigned jc = niter == 0 ? 0 : (j < niter ? j : niter - 1);
      const unsigned tl = jc / Bu;
      b = (int)(jc - tl * Bu);
      const unsigned tile = tile0 + tl;
      return load_hdr(p, tile < ntiles ? tile : ntiles - 1);
    }
    const unsigned task = wid + j * stride;
    const unsigned tq = task < total ? task : total - 1;
    if (binner) {   // batch-invariant Pe, strided form: the B tasks of a tile are consecutive
      const unsigned tl = tq / Bu;   // (their Pe rows / index tables are fetched once: fused_edge.hip)
      b = (int)(tq - tl * Bu);
      return load_hdr(p, tl);
    }
    const unsigned bb = tq / ntiles;
    b = (int)bb;
    return load_hdr(p, tq - bb * ntiles);
  };
  // ---- prologue: weights, vectors and the first tiles' indices in one global round trip
  int b0, b1, b2i;
  const Hdr h0 = task_hdr(0, b0);
  const Hdr h1 = task_hdr(1, b1);
  Hdr hdr2 = task_hdr(2, b2i);
  __builtin_amdgcn_sched_barrier(0);
  VLoad16 lv;
  const float* const vecs[8] = {p.b2, p.gamma, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const int lens[8] = {D, D, 0, 0, 0, 0, 0, 0};
  v16_issue(lv, vecs, lens, tid);
  WLoad16<D * D / 4 / 256> l1, l2;
  if (HAS_EGEMM) w16_issue(l1, p.W1e, p.ldW1e, D, D, D, D, tid, 256);
  w16_issue(l2, p.W2, p.ldW2, D, D, D, D, tid, 256);
  __builtin_amdgcn_sched_barrier(0);
  Ctx cur = load_ctx<PART>(p, h0, b0, lane, q.part_slot);
  Ctx nxt = load_ctx<PART>(p, h1, b1, lane, q.part_slot);
  cur.sc = load_scale(p, cur);
  nxt.sc = load_scale(p, nxt);
  v16_commit(lv, b2s, 2, tid);
  if (HAS_EGEMM) w16_commit(l1, W1im, 0, p.W1e, p.ldW1e, D, D, D, D, tid, 256);
  w16_commit(l2, W2im, 0, p.W2, p.ldW2, D, D, D, D, tid, 256);
  __syncthreads();
  const B3Tile T0p = b3_tile(T0, D), T1p = b3_tile(T1, D), T2p = b3_tile(T2, D);

  f32x16 dW1[NB][NB], dW2[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW1[i][j][r] = dW2[i][j][r] = 0.f;
  float db2[1] = {0.f};
  f32x4 dgam4 = {0.f, 0.f, 0.f, 0.f}, dbet4 = {0.f, 0.f, 0.f, 0.f};

  const int t = lane & 31, hh = lane >> 5;
  const int sub = lane >> 4, c4 = lane & 15;
  const uint32_t col16 = 16u * (uint32_t)c4;
  constexpr bool has_geo = HAS_GEO;
  static_assert(HAS_EGEMM || !HAS_GEO, "g_eout only exists with an edge update");
  const uint32_t ldE = 4u * (uint32_t)p.e.ld, ldR = 4u * (uint32_t)p.pr.ld;
  const uint32_t ldG = 4u * (uint32_t)q.g_agg.ld, ldO = 4u * (uint32_t)q.geo_ld;
  const uint32_t ldGE = 4u * (uint32_t)q.ge_ld, ldGP = 4u * (uint32_t)q.gpr_ld;

  // Row traffic.  Prefetched ONE TILE AHEAD (issued under this tile's MFMAs, consumed at the
  // top of the next tile): e [NVR wave-wide 16-byte loads], ps[snd] [NVR], pr of the tile's
  // (consecutive) receivers [NVC: 16 rows; the rare tile with more fetches the rest on use].
  // Requested at the START of their own tile and used from its LayerNorm backward on: g_eout
  // [NVR] -- kept in row shape until the g_e store, where it is added back -- and g_agg [NVC].
  f32x4 vE[NVR], vS[NVR], vR[NVC], vO[NVR], vG[NVC];
  const float *ebN, *psN, *prN;    // batch bases of the tile being prefetched (wave-uniform)
  Hdr hN;
  const int* tabN;
  auto fetch_begin = [&](const Ctx& c, int parN) {
    int* tab = itab + parN * (2 * NLAM_TILE);
    // (a tile without edges stores its rows to a scratch row: index 0 of a redirected base)
    if (lane < NLAM_TILE) {
      tab[lane] = (c.h.ne > 0 && !ABL) ? c.eid : 0;
      tab[NLAM_TILE + lane] = ABL ? 0 : c.snd;
    }
    wave_sync();
    const int64_t b = c.b;
    ebN = p.e.ptr + b * p.e.bstride;
    psN = p.ps.ptr + b * p.ps.bstride;
    prN = p.pr.ptr + b * p.pr.bstride;
    hN = c.h;
    tabN = tab;
  };
  auto fetch_E = [&](int k) { vE[k] = ldg_off(ebN, row_off(tabN[sub + 4 * k], ldE, col16)); };
  // (sender rows: ids are not bounded by the tile count, so these eight keep 64-bit addresses)
  auto fetch_S = [&](int k) {
    vS[k] = *reinterpret_cast<const f32x4*>(psN + (int64_t)tabN[NLAM_TILE + sub + 4 * k] * p.ps.ld + 4 * c4);
  };
  auto crow = [&](const Hdr& h, int k) {   // compact receiver row of load k (clamped to the last)
    const int tr = sub + 4 * k;
    return h.r0 + (tr < h.nr - 1 ? tr : h.nr - 1);
  };
  auto fetch_R = [&](int k) { vR[k] = ldg_off(prN, row_off(crow(hN, k), ldR, col16)); };

  int par = 0;
  if (niter > 0) {
    fetch_begin(cur, 0);
#pragma unroll
    for (int k = 0; k < NVR; ++k) { fetch_E(k); fetch_S(k); }
#pragma unroll
    for (int k = 0; k < NVC; ++k) fetch_R(k);
  }
  // Enter the loop with nothing in flight: the compiler merges its wait counts over the loop's
  // two entries, and the prologue's short load sequence would otherwise make every wait at the
  // top of a tile as strict as if the tile's 17 row stores did not exist (measured: 1.6 k cycles
  // per tile waiting for store acknowledgements).  vmcnt(0), lgkmcnt / expcnt untouched.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = STAMP ? __builtin_amdgcn_s_memtime() : 0;
  f32x16 gsum[BSUM ? NB : 1];   // BSUM: sum of gh over the batch items of the current tile
#pragma unroll
  for (int nb = 0; nb < (BSUM ? NB : 1); ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) gsum[nb][r] = 0.f;
  for (unsigned j = 0; j < niter; ++j) {
    const Hdr H = cur.h;
    const int ne = H.ne, nr = H.nr, r0 = H.r0;
    const int rcv = cur.rcv;
    const int64_t b = cur.b;
    const int* tabC = itab + par * (2 * NLAM_TILE);
    const int lastrow = ne > 0 ? ne - 1 : 0;
    const float* gaC = q.g_agg.ptr + b * q.g_agg.bstride;
    const float* goC = has_geo ? q.g_eout + b * q.geo_bstride : nullptr;
    auto fetch_O = [&](int k) {
      if constexpr (has_geo) vO[k] = ldg_off(goC, row_off(tabC[sub + 4 * k], ldO, col16));
    };
    auto fetch_G = [&](int k) { vG[k] = ldg_off(gaC, row_off(crow(H, k), ldG, col16)); };
    // ================================================================ P0: stage the rows
    if (HAS_EGEMM) put_rows_planes<NVR>(T0p, vE, sub, c4);          // E stays in T0 (planes)
    else {
#pragma unroll
      for (int k = 0; k < NVR; ++k) vS[k] += vE[k];                 // Pe + Ps
    }
    put_rows<NVR>(T1, vS, sub, c4);
    put_rows<NVC>(T2, vR, sub, c4);                                  // Pr rows of the tile's receivers
    if (nr > 16) {   // (rare) the receivers past the 16 prefetched ones
      f32x4 xr[NVC];
#pragma unroll
      for (int k = 0; k < NVC; ++k) {
        const int tr = 16 + sub + 4 * k;
        const int row = r0 + (tr < nr - 1 ? tr : nr - 1);
        xr[k] = ldg_off(p.pr.ptr + b * p.pr.bstride, row_off(row, ldR, col16));
      }
      put_rows<NVC>(T2 + 16 * LDT, xr, sub, c4);
    }
    wave_sync();
    // this slot's receiver row in the compact tile (a tile without edges: row 0 -- its slots point
    // at CSR position 0, whose receiver is not one of the tile's)
    const int roff = ne > 0 ? rcv - r0 : 0;
    f32x16 hpre[NB];
    row_to_acc<false>(hpre, T1, t, hh);
    row_to_acc<true>(hpre, T2, roff, hh);
    __builtin_amdgcn_sched_barrier(0);
    STAMP2(0)   // rows landed + staged, h assembled
    // ================================================================ P1: recompute h, s, z
    // (under the MFMAs: this tile's gradient rows, then the next tile's e rows)
    fetch_begin(nxt, par ^ 1);
    if (HAS_EGEMM) {
      gemm_tile_cb(hpre, W1im, T0p, lane, [&](int st_) { fetch_O(2 * st_); fetch_O(2 * st_ + 1); fetch_G(st_); });
    }
    f32x16 sact[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = hpre[nb][r];
        const float sg = nlam_sigmoid(x);
        const float s = x * sg;
        sact[nb][r] = s;
        hpre[nb][r] = fmaf(s, 1.0f - sg, sg);     // silu'(h) = sg + s (1 - sg); h itself is dead
      }
    wave_sync();   // (the Pr rows of T2 are in registers)
    acc_to_tile_b3<NB>(sact, T2p, 0, lane);
    f32x16 z[NB];
    vec_to_acc<NB>(z, b2s, lane);
    wave_sync();
    if (HAS_EGEMM) {
      gemm_tile_cb(z, W2im, T2p, lane, NoHook());
    } else {
      gemm_tile_cb(z, W2im, T2p, lane, [&](int st_) { fetch_G(st_); });
    }
    float mean, rstd;
    ln_stats<NB>(z, mean, rstd);
    __builtin_amdgcn_sched_barrier(0);
    STAMP2(1)   // GEMM1, silu / silu', S planes, GEMM2, stats
    // ================================================================ P2: LayerNorm backward
    // gm = sc * g_agg[rec] (+ g_eout); padded slots: exactly zero
    const float msk = (t < ne) ? 1.0f : 0.0f;
    const float sc = cur.sc * msk;
    f32x16 g[NB];
    wave_sync();
    if constexpr (has_geo) {
      put_rows<NVR>(T1, vO, sub, c4);
      wave_sync();
      row_to_acc<false>(g, T1, t, hh);
      wave_sync();
    }
    put_rows<NVC>(T1, vG, sub, c4);               // g_agg rows of the receivers (compact)
    if (nr > 16) {
      f32x4 xg[NVC];
#pragma unroll
      for (int k = 0; k < NVC; ++k) {
        const int tr = 16 + sub + 4 * k;
        const int row = r0 + (tr < nr - 1 ? tr : nr - 1);
        xg[k] = ldg_off(gaC, row_off(row, ldG, col16));
      }
      put_rows<NVC>(T1 + 16 * LDT, xg, sub, c4);
    }
    wave_sync();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(T1 + roff * LDT + 32 * nb + 8 * qq + 4 * hh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * qq + j;
          g[nb][r] = has_geo ? fmaf(g[nb][r], msk, v[j] * sc) : v[j] * sc;
        }
      }
    wave_sync();
    acc_to_tile<NB>(g, T1, LDT, lane);             // gm rows -> dbeta partial sums
    wave_sync();
    colsum4(dbet4, T1, sub, c4);
    constexpr float inv_d = 1.0f / (float)D;
    float s1 = 0.f, s2 = 0.f;
    const float nmr = -mean * rstd;
    wave_sync();   // (the gm rows of T1 are read; DS operations of a wave execute in order)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gs + 32 * nb + 8 * qq + 4 * hh);
        f32x4 prod;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * qq + j;
          const float xh = fmaf(z[nb][r], rstd, nmr);
          prod[j] = g[nb][r] * xh;                   // gm * xhat -> dgamma
          const float gv = g[nb][r] * gm[j];
          g[nb][r] = gv;
          s1 += gv;
          s2 = fmaf(gv, xh, s2);
        }
        *reinterpret_cast<f32x4*>(T1 + t * LDT + 32 * nb + 8 * qq + 4 * hh) = prod;
      }
    wave_sync();
    colsum4(dgam4, T1, sub, c4);
    s1 = lane_xor32_sum(s1);
    s2 = lane_xor32_sum(s2);
    {
      const float m1r = -s1 * inv_d * rstd, m2r = -s2 * inv_d * rstd;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float xh = fmaf(z[nb][r], rstd, nmr);
          g[nb][r] = fmaf(xh, m2r, fmaf(g[nb][r], rstd, m1r));   // gz
        }
    }
    wave_sync();
    acc_to_tile_b3<NB>(g, T1p, 0, lane);           // GZ planes (over the gm * xhat rows, read by now)
    wave_sync();
    __builtin_amdgcn_sched_barrier(0);
    STAMP2(2)   // gm assembled, LayerNorm backward, dgamma / dbeta partial sums, GZ planes
    // ================================================================ P3: dW2, db2, gh
    // indices of the tile after next (consumed one tile from now: no wait in this tile)
    Ctx nn = load_ctx<PART>(p, hdr2, b2i, lane, q.part_slot);
    nxt.sc = load_scale(p, nxt);   // (nxt.rcv landed a tile ago; the prologue's value is the same)
    hdr2 = task_hdr(j + 3, b2i);
    outer_cb(dW2, T1p, T2p, lane, [&](int st_) { fetch_S(2 * st_); fetch_S(2 * st_ + 1); });
    tile_colsum_b3<1>(db2, T1p, 0, lane);
    f32x16 gh[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) gh[nb][r] = 0.f;
    // (BSUM: the batch items of a tile run back to back and Pe is batch-invariant -- its rows are
    // only fetched when the next task starts a new tile)
    const bool new_e = !BSUM || nxt.b == 0;
    gemm_tile_wt_cb(gh, W2im, T1p, lane, [&](int st_) {
      if (new_e) { fetch_E(2 * st_); fetch_E(2 * st_ + 1); }
      fetch_R(st_);
    });
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) gh[nb][r] *= hpre[nb][r];
    __builtin_amdgcn_sched_barrier(0);
    STAMP2(3)   // next indices, dW2, db2, gh = W2^T gz * silu'
    // ================================================================ P4: gh rows, receiver sums
    wave_sync();
    acc_to_tile<NB>(gh, T2, LDT, lane);            // GH fp32 (S planes are dead)
    acc_to_tile_b3<NB>(gh, T1p, 0, lane);          // GH planes (GZ planes are dead)
    wave_sync();
    // row stores: padded slots re-store the last edge's row (same address, same bytes)
    // (unconditional: the compiler's s_waitcnt bookkeeping only counts memory operations that are
    // issued on every path, and the next tile's gathers are waited for past these stores.  A tile
    // of receivers without in-edges -- never on the neural-lam graphs -- stores to a scratch row
    // in this workgroup's slab, which the epilogue overwrites.)
    float* scratch = q.slab + (int64_t)blockIdx.x * q.slab_stride;
    float* ghb = (ne > 0 && !ABL) ? q.gh_out + b * q.gh_bstride : scratch;
    auto store_gh = [&](int k) {
      const int tr = sub + 4 * k;
      const int trc = tr < lastrow ? tr : lastrow;
      const f32x4 v = *reinterpret_cast<const f32x4*>(T2 + trc * LDT + 4 * c4);
      stg_off(ghb, row_off(tabC[sub + 4 * k], 4u * D, col16), v);
    };
    float* gb = q.gpr + b * q.gpr_bstride;
    ind_build(ind, roff, lane);
    if (HAS_EGEMM) {
      f32x16 ge[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) ge[nb][r] = 0.f;
      STAMP2(4)   // GH tile + planes
      gemm_tile_wt_cb(ge, W1im, T1p, lane, [&](int st_) { store_gh(2 * st_); store_gh(2 * st_ + 1); });
      STAMP2(5)   // W1e^T gh with the gh row stores
      // ============================================================== P5: g_e rows, dW1e
      wave_sync();
      acc_to_tile<NB>(ge, T2, LDT, lane);
      wave_sync();
      float* ob = (ne > 0 && !ABL) ? q.g_e + b * q.ge_bstride : scratch;
      outer_cb(dW1, T1p, T0p, lane, [&](int st_) {
#pragma unroll
        for (int k = 2 * st_; k < 2 * st_ + 2; ++k) {
          const int tr = sub + 4 * k;
          const int trc = tr < lastrow ? tr : lastrow;
          f32x4 v = *reinterpret_cast<const f32x4*>(T2 + trc * LDT + 4 * c4);
          if constexpr (has_geo) v += vO[k];      // g_e = g_eout + W1e^T gh (the rows are still in row shape)
          stg_off(ob, row_off(tabC[sub + 4 * k], ldGE, col16), v);
        }
      });
      STAMP2(6)   // dW1e with the g_e row stores
      f32x16 seg[NB];
      seg_mfma(seg, ind, T1p, lane);
      wave_sync();   // (the g_e rows of T2 are read)
      seg_store(seg, T2, nr, gb, (uint32_t)r0 * ldGP, ldGP, lane);
    } else {
      f32x16 seg[NB];
      seg_mfma(seg, ind, T1p, lane);
      if constexpr (!PART) {
#pragma unroll
        for (int k = 0; k < NVR; ++k) store_gh(k);
      }
      if constexpr (BSUM) {
        // dPe = sum over the batch items of this tile (b runs innermost; first item: plain copy)
        const float keep = b == 0 ? 0.0f : 1.0f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) gsum[nb][r] = fmaf(gsum[nb][r], keep, gh[nb][r]);
        if (b == (int64_t)p.B - 1 && ne > 0) {   // (wave-uniform) the tile's last batch item: emit the rows
          acc_to_tile<NB>(gsum, T0, LDT, lane);   // (T0 is free without an edge GEMM)
          wave_sync();
#pragma unroll
          for (int k = 0; k < NVR; ++k) {
            const int tr = sub + 4 * k;
            const int trc = tr < lastrow ? tr : lastrow;
            const f32x4 v = *reinterpret_cast<const f32x4*>(T0 + trc * LDT + 4 * c4);
            stg_off(q.g_e, row_off(tabC[sub + 4 * k], ldGE, col16), v);
          }
        }
      }
      wave_sync();   // (the gh rows of T2 are read)
      seg_store(seg, T2, nr, gb, (uint32_t)r0 * ldGP, ldGP, lane);
      if constexpr (PART) {
        // sums of the tile's gh rows per distinct sender: Ind2[s][k] = (slot(k) == s)
        // (part_slot packs the slot [bits 0-7] and the tile's number of distinct senders [bits 8-15])
        const int ns = ne > 0 ? (__builtin_amdgcn_readfirstlane(cur.slot) >> 8) : 0;
        wave_sync();
        ind_build(ind, cur.slot & 255, lane);
        wave_sync();
        seg_mfma(seg, ind, T1p, lane);
        wave_sync();
        const unsigned tile_id = tile0 + j / Bu;   // (BSUM task numbering: batch items innermost)
        seg_store(seg, T2, ns, q.gpart + b * q.gpart_bstride, 4u * D * (16u * tile_id), 4u * D, lane);
      }
    }
    wave_sync();
    __builtin_amdgcn_sched_barrier(0);
    STAMP2(7)   // receiver sums (no edge update: + the gh row stores)
    cur = nxt;
    nxt = nn;
    par ^= 1;
  }
  if (STAMP && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_edge_bwd2_stamps[k], st[k]);
  }

  // ---- fold the per-wave gradient blocks into this workgroup's slab (fixed order)
  // (the asm MFMAs are opaque to the compiler's hazard recogniser: let the last one drain
  // before its AGPRs are read -- 16 passes of 4 cycles)
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  __syncthreads();
  float* img = smem;
  float* slab = q.slab + (int64_t)blockIdx.x * q.slab_stride;
  constexpr int nW = D * D;
  if (HAS_EGEMM) fold_blocks_to_slab<NB, NB>(dW1, img, D, slab, tid, wave, lane);
  fold_blocks_to_slab<NB, NB>(dW2, img, D, slab + nW, tid, wave, lane);
  fold_vec_lds<1>(db2, img, wave, lane);
  for (int i = tid; i < D; i += 256) slab[2 * nW + i] = img[i];
  __syncthreads();
  // dgamma / dbeta: 16 row-group partials (4 waves x 4 groups) per feature, summed in a fixed order
  float* part = img;   // [2][16][64]
  *reinterpret_cast<f32x4*>(part + (wave * 4 + sub) * D + 4 * c4) = dgam4;
  *reinterpret_cast<f32x4*>(part + 16 * D + (wave * 4 + sub) * D + 4 * c4) = dbet4;
  __syncthreads();
  if (tid < 2 * D) {
    const float* src = part + (tid >> 6) * 16 * D + (tid & 63);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += src[i * D];
    slab[2 * nW + D + tid] = a;   // [dgamma | dbeta] are consecutive
  }
}

template <bool HAS_EGEMM, bool HAS_GEO, bool STAMP = false, bool ABL = false, bool BSUM = false,
          bool PART = false>
static int launch_edge_bwd2(const EdgeBwdParams& q, hipStream_t s) {
  const size_t lds = ((size_t)(HAS_EGEMM ? 2 : 1) * D * (D + 4) + 2 * D +
                      (size_t)4 * 3 * NLAM_TILE * LDT + 4 * 4 * NLAM_TILE) * sizeof(float) +
                     (size_t)4 * NLAM_TILE * IND_PITCH * sizeof(__bf16);
  auto kern = edge_bwd2_kernel<HAS_EGEMM, HAS_GEO, STAMP, ABL, BSUM, PART>;
  NLAM_BIG_LDS(kern, __func__);
  static const int xcd = getenv("NLAM_NO_XCD_ORDER") == nullptr;
  static const int binner = getenv("NLAM_EDGE_BINNER") == nullptr || atoi(getenv("NLAM_EDGE_BINNER")) != 0;
  // (NLAM_EDGE_WAVE_MAJOR: bit 0 = this kernel, bit 1 = edge_fwd; default 3)
  static const int wmajor = getenv("NLAM_EDGE_WAVE_MAJOR") == nullptr || (atoi(getenv("NLAM_EDGE_WAVE_MAJOR")) & 1) != 0;
  kern<<<(unsigned)nlam_bwd_grid(q.f.ntiles * q.f.B), 256, lds, s>>>(q, xcd | (binner ? 2 : 0) | (wmajor ? 8 : 0));
  NLAM_CHECK_LAUNCH("edge_bwd2_kernel");
  return 0;
}

// Whether nlam_edge_bwd (no edge update, batch-invariant Pe) forms dPe = sum_b gh[b] inside the
// kernel: callers ask before they request it, and keep the batch sum folded into the projection
// backward's load otherwise.
extern "C" int nlam_edge_bwd_forms_batch_sum(int64_t ntiles, int64_t B, int d) {
  if (d != D || B <= 1 || ntiles <= 0 || !nlam_mfma_b3() || !nlam_k16_on(K16_EDGE_BWD2)) return 0;
  // (the diagnostic builds -- phase stamps, the no-traffic ablation -- run the strided form and
  // never write dPe: callers must not allocate it and feed it on)
  if (getenv("NLAM_STAMP") != nullptr || getenv("NLAM_STAMP2") != nullptr ||
      getenv("NLAM_ABL2") != nullptr)
    return 0;
  const int64_t nw = 4 * nlam_bwd_grid(ntiles * B);
  const int64_t rounds_tiles = ((ntiles + nw - 1) / nw) * B;
  const double rounds_tasks = (double)ntiles * (double)B / (double)nw;
  return (double)rounds_tiles <= 1.06 * rounds_tasks + 0.5 ? 1 : 0;
}

// -1: a shape this kernel does not take (the caller continues with the older kernels); -2: the
// batch sum dPe was requested but the batch-inner form would be unbalanced (the caller runs this
// kernel again without g_e and forms the sum with nlam_sum_batch)
int nlam_edge_bwd2(const EdgeBwdParams& q, int has_egemm, hipStream_t s) {
  const EdgeFwdParams& p = q.f;
  if (p.e.width != D || !nlam_mfma_b3() || getenv("NLAM_STAMP") != nullptr) return -1;
  if (!nlam_k16_on(has_egemm ? K16_EDGE_BWD2_UPD : K16_EDGE_BWD2)) return -1;
  // 32-bit row offsets (v_mad_u32_u24) on the edge- and receiver-indexed operands: their row ids
  // are below 32 * ntiles (a tile holds <= 32 edges of <= 32 receivers); ids < 2^24, pitches < 2^22
  // floats and every batch item of an operand below 4 GiB
  const int64_t M = (int64_t)p.ntiles * 32;
  auto ok = [](int64_t rows, int64_t ld) { return rows < (1 << 24) && ld < (1 << 22) && rows * ld * 4 < (1ll << 32); };
  if (!ok(M, p.e.ld) || !ok(M, D) || (q.g_eout && !ok(M, q.geo_ld)) || (has_egemm && !ok(M, q.ge_ld)) ||
      !ok(M, p.pr.ld) || !ok(M, q.g_agg.ld) || !ok(M, q.gpr_ld))
    return -1;
  static const bool stamp = getenv("NLAM_STAMP2") != nullptr;
  static const bool abl = getenv("NLAM_ABL2") != nullptr;
  if (abl && has_egemm && q.g_eout) return launch_edge_bwd2<true, true, false, true>(q, s);
  if (stamp && has_egemm && q.g_eout) return launch_edge_bwd2<true, true, true>(q, s);
  if (stamp && !has_egemm) return launch_edge_bwd2<false, false, true>(q, s);
  if (!has_egemm) {
    // dPe = sum_b gh[b] requested (g_e != NULL): only meaningful for a batch-invariant Pe.  The
    // batch-inner form hands out whole tiles: taken when that costs at most ~6 % in rounds
    // (m2g: 7,973 tiles on 1,024 waves = 8 tiles each, 32 rounds against 31.1; g2m: 3,191 tiles
    // = 4 each, 16 rounds against 12.5 -- there the strided form runs and the caller sums)
    if (q.g_e != nullptr) {
      if (!(p.e.bstride == 0 && p.B > 1 && ok(M, q.ge_ld))) return -1;
      if (nlam_edge_bwd_forms_batch_sum(p.ntiles, p.B, D)) {
        if (q.gpart != nullptr) return launch_edge_bwd2<false, false, false, false, true, true>(q, s);
        return launch_edge_bwd2<false, false, false, false, true>(q, s);
      }
      return -2;   // (this kernel WITHOUT the batch sum: see nlam_edge_bwd)
    }
    return launch_edge_bwd2<false, false>(q, s);
  }
  return q.g_eout ? launch_edge_bwd2<true, true>(q, s) : launch_edge_bwd2<true, false>(q, s);
}
