// Body of tail_fwd_kernel (csrc/fused_wide.hip), included once per kernel: the single-problem
// kernel (TAIL_BID = blockIdx.x, TAIL_NBLK = gridDim.x: the text the compiler sees is the kernel as it
// was) and the multi-problem kernel (a problem's share of the grid).  Not a stand-alone header.
  static_assert(!PRE || (HAS_LN && 32 * NOUTB == D), "PRE: the node update (n_out == D, LayerNorm)");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NB = D / 32, NV = D / 8, NO = 32 * NOUTB;
  constexpr int LDT = D + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* W2s = smem;
  float* b2s = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + b3_image_bytes(NO, D));
  float* gs = b2s + NO;
  float* bs = gs + NO;
  float* tile = bs + NO + wave * (NLAM_TILE * LDT);
  // per-wave slot-index tables [a | b | c | y] (see lane_row_index)
  int* itab = reinterpret_cast<int*>(bs + NO + 4 * (NLAM_TILE * LDT)) + wave * (4 * NLAM_TILE);
  const B3Image W2im = b3_image(W2s, NO, D);
  // slot indices of a tile (lanes 0..31), fetched ONE TILE AHEAD so that the row gathers do
  // not wait for an index load
  struct Ctx { WTile w; int ia, ib, ic, iy, rcv; };
  auto load_hdr = [&](int64_t task, int64_t total) {
    const int64_t tq = task < total ? task : total - 1;
    const int64_t bq = tq / p.tl.ntiles;
    return wide_tile(p.tl, tq - bq * p.tl.ntiles);
  };
  auto load_idx = [&](const WTile& w) {
    Ctx c;
    c.w = w;
    c.rcv = p.tl.csr_rec ? wide_index(p.tl.csr_rec, c.w, lane) : 0;
    c.ia = wide_index(p.idx_a, c.w, lane);
    c.ib = p.b.ptr ? wide_index(p.idx_b, c.w, lane) : 0;
    c.ic = p.c.ptr ? wide_index(p.idx_c, c.w, lane) : 0;
    c.iy = wide_index(p.idx_y, c.w, lane);
    return c;
  };
  auto load_ctx = [&](int64_t task, int64_t total) { return load_idx(load_hdr(task, total)); };
  const int64_t total = p.tl.ntiles * p.B;
  const int64_t tstride = (int64_t)TAIL_NBLK * 4;
  int64_t tt = (int64_t)TAIL_BID * 4 + wave;
  // Prologue as ONE chain of overlapping round trips (most launches of the hierarchical models
  // are one tile per wave, i.e. all prologue): first tile's header, then the weights and vectors
  // (in flight), then the slot indices as soon as the header is there, then the LDS images.
  Ctx cur;
  {
    static_assert(NO <= 256, "one vector entry per thread");
    const int64_t tq0 = tt < total ? tt : total - 1;
    // (wave-uniform by construction; readfirstlane lets the header come through the scalar
    // cache, on its own counter, so waiting for it does not wait for the weight loads)
    const int64_t k0 = __builtin_amdgcn_readfirstlane((int)(tq0 - (tq0 / p.tl.ntiles) * p.tl.ntiles));
    const int4 hdr0 = wide_tile_raw(p.tl, k0);
    __builtin_amdgcn_sched_barrier(0);
    VecLoads<3> lv;
    const float* const vsrc[3] = {p.b2, p.gamma, p.beta};
    float* const vdst[3] = {b2s, gs, bs};
    vecs_issue(lv, vsrc, p.n_out, tid);
    WLoad16<16> lw;
    const float* const W0 = PRE ? p.preW : p.W2;
    const int64_t ldW0 = PRE ? p.ldpreW : p.ldW2;
    w16_issue(lw, W0, ldW0, p.n_out, D, NO, D, tid, 256);
    __builtin_amdgcn_sched_barrier(0);
    cur = load_idx(wide_tile_decode(p.tl, k0, hdr0));
    w16_commit(lw, W2im, 0, W0, ldW0, p.n_out, D, NO, D, tid, 256);
    vecs_commit(lv, vdst, NO, tid);
  }
  __syncthreads();
  // PRE: every wave of the workgroup swaps the weight image once (between the two barriers)
  auto swap_to_W2 = [&]() {
    __syncthreads();
    WLoad16<16> lw;
    w16_issue(lw, p.W2, p.ldW2, p.n_out, D, NO, D, tid, 256);
    w16_commit(lw, W2im, 0, p.W2, p.ldW2, p.n_out, D, NO, D, tid, 256);
    __syncthreads();
  };
  if (tt >= total) {
    if constexpr (PRE) swap_to_W2();
    return;
  }
  unsigned long long wst[STAMP ? 8 : 1] = {0};
  unsigned long long wprev = STAMP ? __builtin_amdgcn_s_memtime() : 0;
  for (; tt < total; tt += tstride) {
    const int64_t b = tt / p.tl.ntiles;
    const WTile w = cur.w;
    const int ne = w.ne;
    const int rcv = cur.rcv;
    stash_slot_index(itab, cur.ia, lane);
    stash_slot_index(itab + NLAM_TILE, cur.ib, lane);
    stash_slot_index(itab + 2 * NLAM_TILE, cur.ic, lane);
    stash_slot_index(itab + 3 * NLAM_TILE, cur.iy, lane);
    wave_sync();
    int iy[NV];
    lane_row_index<NV>(iy, itab + 3 * NLAM_TILE, D, lane);
    f32x4 vA[NV];
    {
      int ia[NV];
      lane_row_index<NV>(ia, itab, D, lane);
      load_rows_i<NV>(vA, p.a.ptr + b * p.a.bstride, p.a.ld, ia, D, lane);
    }
    // residual rows of the output (e' = e + m): requested with the gathers, used at the very end
    // of the tile -- their HBM round trip rides under the whole tile.  (Inside the store loop
    // each load was its own basic block with a full wait: 16 serialized round trips per tile,
    // half of the kernel's time: tools/stamp_wide.py.)  Padded slots carry the clamped index
    // of the tile's last row.
    const bool res_rows = p.y != nullptr && p.res.ptr != nullptr && p.vec_y && NO == D;
    f32x4 vRes[NV];
    const Ctx nxt = load_ctx(tt + tstride, total);   // (lands during this tile's work)
    if (p.b.ptr) {
      f32x4 vB[NV];
      {
        int ib[NV];
        lane_row_index<NV>(ib, itab + NLAM_TILE, D, lane);
        load_rows_i<NV>(vB, p.b.ptr + b * p.b.bstride, p.b.ld, ib, D, lane);
      }
      if (p.c.ptr) {
        f32x4 vC[NV];
        int ic[NV];
        lane_row_index<NV>(ic, itab + 2 * NLAM_TILE, D, lane);
        load_rows_i<NV>(vC, p.c.ptr + b * p.c.bstride, p.c.ld, ic, D, lane);
#pragma unroll
        for (int k = 0; k < NV; ++k) vB[k] += vC[k];
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) vA[k] += vB[k];
    }
    // (requested AFTER the gathers were waited for: memory operations retire in order, and the
    // residual rows are not needed before the end of the tile)
    if (res_rows) load_rows_i<NV>(vRes, p.res.ptr + b * p.res.bstride, p.res.ld, iy, D, lane);
    f32x4 vP[PRE ? NV : 1];
    if constexpr (PRE) {   // (contiguous rows p0 .. p0 + ne - 1 of the projected operand)
      const float* pb = p.pre.ptr + b * p.pre.bstride + (int64_t)w.p0 * p.pre.ld;
      const int last = ne > 0 ? ne - 1 : 0;
      load_rows_v<NV>(vP, D, lane, [&](int s2) { return pb + (int64_t)(s2 < last ? s2 : last) * p.pre.ld; });
    }
    if (!PRE && p.h_out != nullptr)
      store_rows_regs<NV>(p.h_out + b * p.h_bstride + (int64_t)w.p0 * D, D, D, ne, lane, vA);
    WSTAMP(0)   // slot tables, row gathers a / b / c issued + landed + summed, h rows stored
    put_rows_v<NV, false>(tile, LDT, 0, D, ne, lane, vA);
    wave_sync();
    f32x16 a1[NB];
    tile_to_acc<NB>(a1, tile, LDT, lane);
    if constexpr (PRE) {
      // h = a + pre . preW^T with preW's image in LDS, then W2's image takes its place
      wave_sync();
      put_rows_v<NV, false>(tile, LDT, 0, D, ne, lane, vP);
      wave_sync();
      {
        f32x16 xa[NB];
        tile_to_acc<NB>(xa, tile, LDT, lane);
        gemm_acc_b3<NB, NB, TERMS>(a1, W2im, 0, xa, lane);
      }
      swap_to_W2();
      if (p.h_out != nullptr) {
        acc_to_tile<NB>(a1, tile, LDT, lane);
        wave_sync();
        float* hb = p.h_out + b * p.h_bstride + (int64_t)w.p0 * D;
        store_rows<true>(tile, LDT, 0, D, ne, lane, [&](int s2) { return hb + (int64_t)s2 * D; });
        wave_sync();
      }
    }
    WSTAMP(1)   // h tile staged + back in accumulator layout
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[nb][r] = nlam_silu(a1[nb][r]);
    f32x16 m[NOUTB];
    vec_to_acc<NOUTB>(m, b2s, lane);
    WSTAMP(2)   // silu
    gemm_acc_b3<NOUTB, NB, TERMS>(m, W2im, 0, a1, lane);
    WSTAMP(3)   // GEMM (W2 silu(h) + b2)
    if (HAS_LN) ln_apply<NOUTB>(m, gs, bs, lane);
    wave_sync();
    acc_to_tile<NOUTB>(m, tile, LDT, lane);
    wave_sync();
    WSTAMP(4)   // LayerNorm + message tile
    if (p.agg != nullptr) {
      float* aggb = p.agg + b * p.agg_bstride;
      const int t = lane & 31;
      // lane i <= nr: segment boundaries of the tile's receivers (edge mode only)
      const int ri = w.r0 + (lane < w.nr ? lane : w.nr);
      const int rp = p.tl.csr_rowptr[ri] - w.p0;
      const float invd = p.inv_deg ? p.inv_deg[w.r0 + (lane < w.nr ? lane : 0)] : 1.0f;
      const int rpn = __shfl_down(rp, 1, 64);
      const bool dense = __all((lane >= w.nr) || (rpn > rp));
      (void)t;
      if (dense) {
        tile_segment_sums<NO>(tile, LDT, ne, rcv, lane, [&](int r, int f0, float acc) {
          const float sc = __shfl(invd, r - w.r0, 64);
          aggb[(int64_t)r * p.agg_ld + f0 + lane] = acc * sc;
        });
      } else {
        for (int i = 0; i < w.nr; ++i) {
          const int beg = __shfl(rp, i, 64), end = __shfl(rp, i + 1, 64);
          const float sc = __shfl(invd, i, 64);
#pragma unroll
          for (int f0 = 0; f0 < NO; f0 += 64) {
            float acc = 0.f;
            for (int s = beg; s < end; ++s) acc += tile[s * LDT + f0 + lane];
            aggb[(int64_t)(w.r0 + i) * p.agg_ld + f0 + lane] = acc * sc;
          }
        }
      }
    }
    WSTAMP(5)   // receiver sums
    if (p.y != nullptr) {
      // scattered rows (idx_y) are float4-only (checked on the host); narrow outputs are
      // contiguous rows and take the scalar path
      float* yb = p.y + b * p.y_bstride;
      const float* rb = p.res.ptr ? p.res.ptr + b * p.res.bstride : nullptr;
      if (p.vec_y && NO == D) {
        if (rb) {
          // (residual rows: requested at the top of the tile)
          constexpr int lpr = D >> 2, rpi = 64 / lpr;
          const int sub = lane / lpr, c4 = lane - sub * lpr;
#pragma unroll
          for (int k = 0; k < NV; ++k) {
            const int tr = sub + k * rpi;
            if (tr < ne) {
              const f32x4 x = *(reinterpret_cast<const f32x4*>(tile + tr * LDT) + c4) + vRes[k];
              reinterpret_cast<f32x4*>(yb + (int64_t)iy[k] * p.y_ld)[c4] = x;
            }
          }
        } else {
          store_rows_i<NV, false>(tile, LDT, 0, D, ne, lane, yb, p.y_ld, iy);
        }
      } else {
        auto y_row = [&](int s) { return yb + (int64_t)(w.p0 + s) * p.y_ld; };
        auto r_row = [&](int s) { return rb + (int64_t)(w.p0 + s) * p.res.ld; };
        if (rb) {
          if (p.vec_y) store_rows_res<true>(tile, LDT, 0, p.n_out, ne, lane, y_row, r_row);
          else store_rows_res<false>(tile, LDT, 0, p.n_out, ne, lane, y_row, r_row);
        } else {
          if (p.vec_y) store_rows<true>(tile, LDT, 0, p.n_out, ne, lane, y_row);
          else store_rows<false>(tile, LDT, 0, p.n_out, ne, lane, y_row);
        }
      }
    }
    wave_sync();
    WSTAMP(6)   // row stores (+ residual)
    cur = nxt;
  }
  if constexpr (STAMP) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(&g_wide_stamps[k], wst[k]);
    }
  }
