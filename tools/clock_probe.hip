// Shader clock under MFMA load: s_memtime (core clock) vs s_memrealtime (100 MHz), and the
// cycles a chain of dependent v_mfma_f32_32x32x2_f32 takes.  Build:
//   hipcc --offload-arch=gfx950 -O3 -o tools/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters, float* sink) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const float a = threadIdx.x * 1e-3f, b = 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}
int main() {
  const int grids[] = {1, 256, 1024};
  unsigned long long* d; float* sink;
  hipMalloc(&d, 2 * 1024 * sizeof(unsigned long long)); hipMalloc(&sink, 4);
  for (int gi = 0; gi < 3; ++gi) {
    for (int iters : {2000, 20000}) {
      const int g = grids[gi];
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      probe<<<g, 256>>>(d, iters, sink);  // warm
      hipDeviceSynchronize();
      hipEventRecord(e0);
      probe<<<g, 256>>>(d, iters, sink);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2048]; hipMemcpy(h, d, 2 * g * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double ct = 0, rt = 0; for (int i = 0; i < g; ++i) { ct += h[2 * i]; rt += h[2 * i + 1]; }
      ct /= g; rt /= g;
      const double mf = 32.0 * iters;   // MFMAs per wave
      printf("grid %4d iters %6d: kernel %.3f ms | s_memtime %.0f ticks, s_memrealtime %.0f ticks (100 MHz -> %.1f us)"
             " | memtime ticks per MFMA %.2f | memtime MHz %.0f | TFLOP/s (4 waves/CU x grid) %.1f\n",
             g, iters, ms, ct, rt, rt / 100.0, ct / mf, ct / (rt / 100.0),
             (double)g * 4 * mf * 4096.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
