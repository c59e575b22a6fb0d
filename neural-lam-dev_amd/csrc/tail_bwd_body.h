// Body of tail_bwd_kernel (csrc/fused_wide.hip), included once per kernel: the single-problem
// kernel (TAIL_BID = blockIdx.x, TAIL_NBLK = gridDim.x: the text the compiler sees is the kernel as it
// was) and the multi-problem kernel (a problem's share of the grid).  Not a stand-alone header.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NB = D / 32, NV = D / 8, NO = 32 * NOUTB, NVG = NO / 8;
  constexpr int NV_O = (NO + 63) / 64;
  constexpr int LDT = D + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 31, hh = lane >> 5;
  float* W2s = smem;
  float* b2s = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + b3_image_bytes(NO, D));
  float* gs = b2s + NO;
  float* tile = gs + NO + wave * (NLAM_TILE * LDT);
  int* itab = reinterpret_cast<int*>(gs + NO + 4 * (NLAM_TILE * LDT)) + wave * (4 * NLAM_TILE);
  const B3Image W2im = b3_image(W2s, NO, D);
  // slot indices / row scales of a tile (lanes 0..31), fetched one tile ahead
  struct Ctx { WTile w; int i1, i2, igh, rcv; float sc1; };
  auto load_idx = [&](const WTile& w) {
    Ctx c;
    c.w = w;
    c.rcv = q.tl.csr_rec ? wide_index(q.tl.csr_rec, c.w, lane) : 0;
    c.i1 = wide_index(q.idx_g1, c.w, lane);
    c.i2 = q.g2.ptr ? wide_index(q.idx_g2, c.w, lane) : 0;
    c.igh = wide_index(q.idx_gh, c.w, lane);
    c.sc1 = q.scale1 ? q.scale1[c.i1] : 1.0f;   // (dependent load, hidden by the prefetch)
    return c;
  };
  auto load_ctx = [&](int64_t task, int64_t total) {
    const int64_t tq = task < total ? task : total - 1;
    const int64_t bq = tq / q.tl.ntiles;
    return load_idx(wide_tile(q.tl, tq - bq * q.tl.ntiles));
  };
  const int64_t total = q.tl.ntiles * q.B;
  const int64_t tstride = (int64_t)TAIL_NBLK * 4;
  int64_t tt = (int64_t)TAIL_BID * 4 + wave;
  // prologue: first tile's header (scalar load), weights + vectors in flight, then the slot
  // indices, then the LDS images (see tail_fwd_kernel)
  Ctx cur;
  {
    static_assert(NO <= 256, "one vector entry per thread");
    const int64_t tot1 = total > 0 ? total : 1;
    const int64_t tq0 = tt < tot1 ? tt : tot1 - 1;
    const int64_t k0 = __builtin_amdgcn_readfirstlane((int)(tq0 - (tq0 / q.tl.ntiles) * q.tl.ntiles));
    const int4 hdr0 = wide_tile_raw(q.tl, k0);
    __builtin_amdgcn_sched_barrier(0);
    VecLoads<2> lv;
    const float* const vsrc[2] = {q.b2, q.gamma};
    float* const vdst[2] = {b2s, gs};
    vecs_issue(lv, vsrc, q.n_out, tid);
    WLoad16<16> lw;
    w16_issue(lw, q.W2, q.ldW2, q.n_out, D, NO, D, tid, 256);
    __builtin_amdgcn_sched_barrier(0);
    cur = load_idx(wide_tile_decode(q.tl, k0, hdr0));
    w16_commit(lw, W2im, 0, q.W2, q.ldW2, q.n_out, D, NO, D, tid, 256);
    vecs_commit(lv, vdst, NO, tid);
  }
  __syncthreads();

  // per-feature partial sums (lanes = features 64 j + lane), accumulated over the tiles
  float dgam[NV_O], dbet[NV_O];
#pragma unroll
  for (int j = 0; j < NV_O; ++j) dgam[j] = dbet[j] = 0.f;
  const B3Tile Tp = b3_tile(tile, NO);   // bf16-plane view of the tile (column sums)
  unsigned long long wst[STAMP ? 8 : 1] = {0};
  unsigned long long wprev = STAMP ? __builtin_amdgcn_s_memtime() : 0;
  for (; tt < total; tt += tstride) {
    const int64_t b = tt / q.tl.ntiles;
    const WTile w = cur.w;
    const int ne = w.ne;
    const int i1 = cur.i1, i2 = cur.i2;
    const int rcv = cur.rcv;
    const float sc1 = cur.sc1;
    // per-wave slot tables [g1 | g2 | gh | scale bits] -> per-lane row indices
    stash_slot_index(itab, cur.i1, lane);
    stash_slot_index(itab + NLAM_TILE, cur.i2, lane);
    stash_slot_index(itab + 2 * NLAM_TILE, cur.igh, lane);
    stash_slot_index(itab + 3 * NLAM_TILE, __float_as_int(cur.sc1), lane);
    wave_sync();
    const Ctx nxt = load_ctx(tt + tstride, total);
    const float* g1b = q.g1.ptr + b * q.g1.bstride;
    const float* g2b = q.g2.ptr ? q.g2.ptr + b * q.g2.bstride : nullptr;
    auto g1_row = [&](int s) { return g1b + (int64_t)__shfl(i1, s, 64) * q.g1.ld; };
    auto g2_row = [&](int s) { return g2b + (int64_t)__shfl(i2, s, 64) * q.g2.ld; };
    // ---- h rows (contiguous) -> tile -> accumulator layout.  h is staged twice (here for
    // s = silu(h), below for silu'(h)) instead of living in 64 registers across the whole
    // tile: at d = 128 that is the difference between spilling and not (L2-hot re-read).
    const float* hb = q.h + b * q.h_bstride + (int64_t)(ne > 0 ? w.p0 : (w.p0 > 0 ? w.p0 - 1 : 0)) * D;
    const int last = ne > 0 ? ne - 1 : 0;
    auto h_row = [&](int s) { return hb + (int64_t)(s < last ? s : last) * D; };
    {
      f32x4 vH[NV];
      load_rows_v<NV>(vH, D, lane, h_row);
      put_rows_v<NV, false>(tile, LDT, 0, D, ne, lane, vH);
      wave_sync();
    }
    WSTAMP(0)   // slot tables, h rows requested + landed + staged
    // ---- incoming gradient rows -> tile (their loads fly under the GEMM below)
    f32x4 vG[NVG];
    if (q.vec_g) {
      // per-lane row indices (and row scales) straight from the tables: no shuffles
      int ig[NVG];
      lane_row_index<NVG>(ig, itab, NO, lane);
      load_rows_i<NVG>(vG, g1b, q.g1.ld, ig, NO, lane);
      if (g2b) {
        f32x4 vO[NVG];
        lane_row_index<NVG>(ig, itab + NLAM_TILE, NO, lane);
        load_rows_i<NVG>(vO, g2b, q.g2.ld, ig, NO, lane);
        if (q.scale1 != nullptr) {
          lane_row_index<NVG>(ig, itab + 3 * NLAM_TILE, NO, lane);
#pragma unroll
          for (int k = 0; k < NVG; ++k) vG[k] *= __int_as_float(ig[k]);
        }
#pragma unroll
        for (int k = 0; k < NVG; ++k) vG[k] += vO[k];
      } else if (q.scale1 != nullptr) {
        lane_row_index<NVG>(ig, itab + 3 * NLAM_TILE, NO, lane);
#pragma unroll
        for (int k = 0; k < NVG; ++k) vG[k] *= __int_as_float(ig[k]);
      }
    }
    f32x16 z[NOUTB];
    if (HAS_LN) {
      f32x16 sact[NB];
      tile_to_acc<NB>(sact, tile, LDT, lane);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sact[nb][r] = nlam_silu(sact[nb][r]);
      vec_to_acc<NOUTB>(z, b2s, lane);
      gemm_acc_b3<NOUTB, NB, TERMS>(z, W2im, 0, sact, lane);
    }
    WSTAMP(1)   // gradient rows issued, silu, GEMM (z recomputed)
    wave_sync();
    if (q.vec_g) {
      put_rows_v<NVG, false>(tile, LDT, 0, NO, ne, lane, vG);
    } else {
      // narrow / unaligned gradient rows (e.g. the 17-wide output map): scalar staging
      // (uniform trip count; the index shuffles stay outside the divergent part)
      for (int idx = lane; idx < NLAM_TILE * NO; idx += 64) {
        const int tr = idx / NO, cc = idx - tr * NO;
        const float* r1 = g1_row(tr);
        const float* r2 = g2b ? g2_row(tr) : nullptr;
        const float sc = __shfl(sc1, tr, 64);
        float v = 0.f;
        if (tr < ne && cc < q.n_out) {
          v = r1[cc] * sc;
          if (r2) v += r2[cc];
        }
        tile[tr * LDT + cc] = v;
      }
    }
    wave_sync();
    f32x16 g[NOUTB];
    tile_to_acc<NOUTB>(g, tile, LDT, lane);
    if (HAS_LN) {
      constexpr float inv_d = 1.0f / (float)NO;
      float mean, rstd;
      ln_stats<NOUTB>(z, mean, rstd);
      // dbeta: column sums of the incoming gradient, on the matrix cores from bf16 planes
      wave_sync();
      acc_to_tile_b3<NOUTB>(g, Tp, 0, lane);
      wave_sync();
      tile_colsum_b3<NV_O, TERMS>(dbet, Tp, 0, lane);
      wave_sync();
      float s1 = 0.f, s2 = 0.f;
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
            prod[0][r] = g[nb][r] * xh;          // gy * xhat -> dgamma
            const float gv = g[nb][r] * gm[j];
            g[nb][r] = gv;
            s1 += gv;
            s2 += gv * xh;
          }
        }
        acc_to_tile_b3<1>(prod, Tp, 32 * nb, lane);
      }
      wave_sync();
      tile_colsum_b3<NV_O, TERMS>(dgam, Tp, 0, lane);
      s1 = lane_xor32_sum(s1);
      s2 = lane_xor32_sum(s2);
      const float m1 = s1 * inv_d, m2 = s2 * inv_d;
#pragma unroll
      for (int nb = 0; nb < NOUTB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[nb][r] = rstd * (g[nb][r] - m1 - z[nb][r] * m2);
    }
    WSTAMP(2)   // gradient rows staged, LayerNorm backward, dbeta / dgamma column sums
    // g = gz (zero on padded slots / columns): publish it for the weight-gradient pass
    wave_sync();
    acc_to_tile<NOUTB>(g, tile, LDT, lane);
    wave_sync();
    {
      float* gzb = q.gz_out + b * q.gz_bstride + (int64_t)w.p0 * NO;
      auto gz_row = [&](int s) { return gzb + (int64_t)s * NO; };
      store_rows<true>(tile, LDT, 0, NO, ne, lane, gz_row);
    }
    WSTAMP(3)   // gz tile + row stores
    // gh = (W2^T gz) * silu'(h)
    f32x16 gh[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) gh[nb][r] = 0.f;
    {
      // h again (its loads fly under the GEMM)
      f32x4 vH[NV];
      load_rows_v<NV>(vH, D, lane, h_row);
      gemm_acc_wt_b3<NB, NOUTB, TERMS>(gh, W2im, 0, g, lane);
      wave_sync();   // (the gz rows of the tile are stored)
      put_rows_v<NV, false>(tile, LDT, 0, D, ne, lane, vH);
      wave_sync();
    }
    WSTAMP(4)   // h rows again + GEMM (W2^T gz)
    {
      f32x16 hpre[NB];
      tile_to_acc<NB>(hpre, tile, LDT, lane);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) gh[nb][r] *= nlam_silu_grad(hpre[nb][r]);
    }
    wave_sync();
    acc_to_tile<NB>(gh, tile, LDT, lane);
    wave_sync();
    {
      float* ghb = q.gh + b * q.gh_bstride;
      int igh[NV];
      lane_row_index<NV>(igh, itab + 2 * NLAM_TILE, D, lane);
      store_rows_i<NV, false>(tile, LDT, 0, D, ne, lane, ghb, q.gh_ld, igh);
    }
    WSTAMP(5)   // silu', gh tile + row stores
    if (q.gpr != nullptr) {
      float* gb = q.gpr + b * q.gpr_bstride;
      const int ri = w.r0 + (lane < w.nr ? lane : w.nr);
      const int rp = q.tl.csr_rowptr[ri] - w.p0;
      const int rpn = __shfl_down(rp, 1, 64);
      const bool dense = __all((lane >= w.nr) || (rpn > rp));
      if (dense) {
        tile_segment_sums<D>(tile, LDT, ne, rcv, lane, [&](int r, int f0, float acc) {
          gb[(int64_t)r * q.gpr_ld + f0 + lane] = acc;
        });
      } else {
        for (int i = 0; i < w.nr; ++i) {
          const int beg = __shfl(rp, i, 64), end = __shfl(rp, i + 1, 64);
#pragma unroll
          for (int f0 = 0; f0 < D; f0 += 64) {
            float acc = 0.f;
            for (int s = beg; s < end; ++s) acc += tile[s * LDT + f0 + lane];
            gb[(int64_t)(w.r0 + i) * q.gpr_ld + f0 + lane] = acc;
          }
        }
      }
    }
    wave_sync();
    WSTAMP(6)   // receiver sums
    cur = nxt;
  }
  (void)t;
  if constexpr (STAMP) {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(&g_wide_stamps[8 + k], wst[k]);
    }
  }
  if (HAS_LN) {
    __syncthreads();
    float* img = smem;
    float* slab = q.slab + (int64_t)TAIL_BID * q.slab_stride;
    fold_vec_lds<NV_O>(dgam, img, wave, lane);
    for (int i = tid; i < NO; i += 256) slab[i] = img[i];
    __syncthreads();
    fold_vec_lds<NV_O>(dbet, img, wave, lane);
    for (int i = tid; i < NO; i += 256) slab[NO + i] = img[i];
  }
