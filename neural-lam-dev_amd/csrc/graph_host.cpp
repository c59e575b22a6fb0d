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
