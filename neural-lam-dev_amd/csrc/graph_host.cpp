// Host-side graph preprocessing for libnlam_hip.so: receiver-sorted (CSR) and
// sender-sorted (CSC) edge tables, built once per InteractionNet.
#include <vector>

#include "nlam_common.h"

extern "C" int nlam_graph_build_host(const int64_t* send, const int64_t* rec, int64_t M,
                                     int64_t n_send, int64_t n_rec, int32_t* csr_rowptr,
                                     int32_t* csr_eid, int32_t* csr_send, int32_t* csr_rec,
                                     int32_t* csc_colptr, int32_t* csc_pos, int32_t* csc_eid,
                                     float* inv_deg) {
  NLAM_REQUIRE(M >= 0 && n_send > 0 && n_rec > 0, "graph_build: bad sizes");
  NLAM_REQUIRE(M < (int64_t)0x7fffffff && n_send < (int64_t)0x7fffffff &&
                   n_rec < (int64_t)0x7fffffff,
               "graph_build: sizes exceed int32");
  for (int64_t e = 0; e < M; ++e) {
    NLAM_REQUIRE(send[e] >= 0 && send[e] < n_send, "graph_build: sender %ld of edge %ld out of [0,%ld)",
                 (long)send[e], (long)e, (long)n_send);
    NLAM_REQUIRE(rec[e] >= 0 && rec[e] < n_rec, "graph_build: receiver %ld of edge %ld out of [0,%ld)",
                 (long)rec[e], (long)e, (long)n_rec);
  }
  // counting sort by receiver (stable in original edge order)
  for (int64_t i = 0; i <= n_rec; ++i) csr_rowptr[i] = 0;
  for (int64_t e = 0; e < M; ++e) csr_rowptr[rec[e] + 1]++;
  for (int64_t i = 0; i < n_rec; ++i) {
    const int32_t deg = csr_rowptr[i + 1];
    if (inv_deg) inv_deg[i] = 1.0f / (float)(deg > 1 ? deg : 1);
    csr_rowptr[i + 1] += csr_rowptr[i];
  }
  std::vector<int32_t> cursor(csr_rowptr, csr_rowptr + n_rec);
  std::vector<int32_t> pos_of_edge((size_t)M);
  for (int64_t e = 0; e < M; ++e) {
    const int32_t p = cursor[rec[e]]++;
    csr_eid[p] = (int32_t)e;
    csr_send[p] = (int32_t)send[e];
    csr_rec[p] = (int32_t)rec[e];
    pos_of_edge[(size_t)e] = p;
  }
  // counting sort by sender; entries are CSR positions, ascending per sender
  for (int64_t j = 0; j <= n_send; ++j) csc_colptr[j] = 0;
  for (int64_t e = 0; e < M; ++e) csc_colptr[send[e] + 1]++;
  for (int64_t j = 0; j < n_send; ++j) csc_colptr[j + 1] += csc_colptr[j];
  std::vector<int32_t> cur2(csc_colptr, csc_colptr + n_send);
  for (int64_t p = 0; p < M; ++p) {
    const int32_t j = csr_send[p];
    const int32_t q = cur2[j]++;
    csc_pos[q] = (int32_t)p;
    csc_eid[q] = csr_eid[p];
  }
  return 0;
}

// Receiver-aligned edge tiles for the fused edge kernels: consecutive CSR
// positions [p0, p1) with p1 - p0 <= max_edges covering the whole in-edge
// segments of consecutive receivers [r0, r1), r1 - r0 <= max_recs.  A tile never
// splits a receiver's segment, so per-tile reductions need no cross-tile carry
// (deterministic, no atomics).  tiles: int32[4 * capacity] = (p0, p1, r0, r1).
// Returns the number of tiles, or -1 (a receiver has more than max_edges
// in-edges) / -2 (capacity too small).
extern "C" int64_t nlam_graph_tiles_host(const int32_t* csr_rowptr, int64_t n_rec,
                                         int32_t max_edges, int32_t max_recs, int32_t* tiles,
                                         int64_t capacity) {
  int64_t nt = 0;
  int64_t r = 0;
  while (r < n_rec) {
    const int32_t p0 = csr_rowptr[r];
    int64_t r1 = r;
    while (r1 < n_rec && (r1 - r) < max_recs && (csr_rowptr[r1 + 1] - p0) <= max_edges) ++r1;
    if (r1 == r) {
      nlam_set_error("graph_tiles: receiver %ld has %d in-edges (> %d)", (long)r,
                     csr_rowptr[r + 1] - p0, max_edges);
      return -1;
    }
    if (nt >= capacity) {
      nlam_set_error("graph_tiles: capacity %ld too small", (long)capacity);
      return -2;
    }
    tiles[4 * nt + 0] = p0;
    tiles[4 * nt + 1] = csr_rowptr[r1];
    tiles[4 * nt + 2] = (int32_t)r;
    tiles[4 * nt + 3] = (int32_t)r1;
    ++nt;
    r = r1;
  }
  return nt;
}
