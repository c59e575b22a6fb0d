// Probe: xor-16 / xor-32 lane sums through gfx950's v_permlane16_swap / v_permlane32_swap (VALU)
// against __shfl_xor (ds_bpermute: LDS crossbar).  Build: hipcc --offload-arch=gfx950 -O3 -o
// tools/permlane_probe tools/permlane_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
// (the builtins fold their two results into one register when both operands hold the same value
// -- hipcc 7.2 emits `v_add v, r0, r0` -- so the swaps are inline asm; `s_nop 1` covers the
// "VALU write -> v_permlane read" hazard the assembler does not see inside an asm string)
__device__ __forceinline__ float swap_sum16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float swap_sum32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(float* out) {
  const float x = (float)(threadIdx.x * threadIdx.x + 1);
  float s = swap_sum32(swap_sum16(x));
  float t = x;
  t += __shfl_xor(t, 16, 64);
  t += __shfl_xor(t, 32, 64);
  out[threadIdx.x] = s;
  out[64 + threadIdx.x] = t;
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  float h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) if (h[i] != h[64 + i]) ++bad;
  printf("permlane swap sums vs shfl_xor: %d mismatches (lane 5: %g vs %g)\n", bad, h[5], h[69]);
  return bad != 0;
}
