// Micro-benchmark: what does the SHAPE of a wave-wide row access cost on gfx950?
//   shape 0 ("rows"):      one instruction = 4 whole 256-byte rows  (lane = (row l >> 4, chunk l & 15))
//   shape 1 ("fragment"):  one instruction = 16 rows x 64 bytes     (lane = (row l & 15, chunk 4 fb + (l >> 4)))
// Both move 16 rows x 256 B per 4 instructions.  Modes: gather (indexed loads, summed), scatter
// (indexed stores), copy (gather + scatter).  Build: hipcc --offload-arch=gfx950 -O3 -o
// tools/rowshape_probe tools/rowshape_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int MODE, int DEPTH>   // DEPTH: 16-row groups in flight per wave
__global__ __launch_bounds__(512) void probe(const float* __restrict__ tab, const int* __restrict__ idx,
                                             float* __restrict__ out, const int* __restrict__ oidx,
                                             long ngroups) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long g0 = wave * DEPTH; g0 < ngroups; g0 += nwaves * DEPTH) {
    f32x4 v[DEPTH][4];
    int orow[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long grp = g0 + d < ngroups ? g0 + d : ngroups - 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int r, c;
        if (SHAPE == 0) { r = (lane >> 4) + 4 * k; c = lane & 15; }
        else { r = lane & 15; c = 4 * k + (lane >> 4); }
        const int row = idx[grp * 16 + r];
        orow[d][k] = oidx[grp * 16 + r];
        if (MODE != 1) v[d][k] = *reinterpret_cast<const f32x4*>(tab + (long)row * 64 + 4 * c);
        else v[d][k] = f32x4{(float)row, 1.f, 2.f, 3.f};
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int c;
        if (SHAPE == 0) c = lane & 15; else c = 4 * k + (lane >> 4);
        if (MODE == 0) acc += v[d][k];
        else if (g0 + d < ngroups)
          *reinterpret_cast<f32x4*>(out + (long)orow[d][k] * 64 + 4 * c) = v[d][k];
      }
    }
  }
  if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[0] = acc[0];
}

template <int SHAPE, int MODE, int DEPTH>
static float run(const float* tab, const int* idx, float* out, const int* oidx, long ngroups, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) probe<SHAPE, MODE, DEPTH><<<grid, 512>>>(tab, idx, out, oidx, ngroups);
  hipEventRecord(a);
  const int it = 10;
  for (int i = 0; i < it; ++i) probe<SHAPE, MODE, DEPTH><<<grid, 512>>>(tab, idx, out, oidx, ngroups);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / it;
}

int main() {
  const long nrows_small = 26244, nrows_big = 1 << 20;     // 6.7 MB (L2 / MALL) and 268 MB (HBM)
  const long M = 230464;                                  // rows moved per launch (m2m edges x 4)
  const long ngroups = M / 16;
  float *tab, *out; int *idx, *oidx;
  hipMalloc(&tab, nrows_big * 256); hipMalloc(&out, nrows_big * 256);
  hipMalloc(&idx, M * 4); hipMalloc(&oidx, M * 4);
  hipMemset(tab, 0, nrows_big * 256);
  std::vector<int> h(M), ho(M);
  for (int cfg = 0; cfg < 3; ++cfg) {
    // cfg 0: node rows (small table, mesh-local random: 16 consecutive slots hit ~3 receivers)
    // cfg 1: node rows, one distinct sender per slot within a local window
    // cfg 2: edge rows (big table, scattered within a 4096-row neighbourhood: CSR -> edge id)
    srand(1);
    for (long i = 0; i < M; ++i) {
      if (cfg == 0) h[i] = (int)(((i / 9) + 0) % nrows_small);
      else if (cfg == 1) h[i] = (int)(((i / 9) + (rand() % 200)) % nrows_small);
      else h[i] = (int)(((i / 4096) * 4096 + (rand() % 4096)) % nrows_big);
      ho[i] = (int)(((i / 4096) * 4096 + (rand() % 4096)) % nrows_big);
    }
    hipMemcpy(idx, h.data(), M * 4, hipMemcpyHostToDevice);
    hipMemcpy(oidx, ho.data(), M * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"receiver rows (dup, L2)", "sender rows (local, L2)", "edge rows (scattered, 268 MB)"};
    for (int grid = 256; grid <= 512; grid += 256) {
      const double gb = M * 256.0 / 1e9;
      float t;
#define RUN(S, Mo, D, label) t = run<S, Mo, D>(tab, idx, out, oidx, ngroups, grid); \
      printf("%-34s grid %3d %-28s %7.1f us %7.2f TB/s\n", names[cfg], grid, label, t * 1e3, gb / t);
      RUN(0, 0, 1, "gather rows d1")
      RUN(1, 0, 1, "gather frag d1")
      RUN(0, 0, 2, "gather rows d2")
      RUN(1, 0, 2, "gather frag d2")
      RUN(0, 0, 4, "gather rows d4")
      RUN(1, 0, 4, "gather frag d4")
      if (cfg == 2) {
        RUN(0, 1, 1, "scatter rows d1")
        RUN(1, 1, 1, "scatter frag d1")
        RUN(0, 2, 2, "copy rows d2")
        RUN(1, 2, 2, "copy frag d2")
      }
    }
  }
  return 0;
}
