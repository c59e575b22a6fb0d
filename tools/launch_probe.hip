// Fixed cost of a kernel launch in a stream of dependent launches on this box: null kernels,
// large dynamic LDS, many workgroups, and kernels that leave dirty lines in L2.
//   hipcc --offload-arch=gfx950 -O3 -o tools/launch_probe tools/launch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_null(float* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_lds(float* p) {
  extern __shared__ float sm[];
  if (p == nullptr && threadIdx.x == 9999) { sm[0] = 1.f; p[0] = sm[1]; }
}
__global__ void k_write(float4* p, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    p[i] = float4{1.f, 2.f, 3.f, 4.f};
}
__global__ void k_read(const float4* p, long n4, float* out) {
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    s += p[i].x;
  if (s == 123.456f) out[0] = s;
}
template <typename F> static double time_us(F f, int n) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / n;
}
int main() {
  float* d; hipMalloc(&d, 256 << 20);
  hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int N = 300;
  printf("null <<<1,64>>>              : %.2f us/launch\n", time_us([&] { k_null<<<1, 64>>>(d); }, N));
  printf("null <<<206,256>>>           : %.2f us/launch\n", time_us([&] { k_null<<<206, 256>>>(d); }, N));
  printf("null <<<1024,256>>>          : %.2f us/launch\n", time_us([&] { k_null<<<1024, 256>>>(d); }, N));
  printf("lds 70 KB <<<206,256>>>      : %.2f us/launch\n", time_us([&] { k_lds<<<206, 256, 70 * 1024>>>(d); }, N));
  printf("lds 140 KB <<<256,256>>>     : %.2f us/launch\n", time_us([&] { k_lds<<<256, 256, 140 * 1024>>>(d); }, N));
  for (long mb : {1L, 13L, 64L}) {
    const long n4 = mb * (1 << 20) / 16;
    printf("write %3ld MB <<<1024,256>>>  : %.2f us/launch\n", mb, time_us([&] { k_write<<<1024, 256>>>((float4*)d, n4); }, N));
    printf("read  %3ld MB <<<1024,256>>>  : %.2f us/launch\n", mb, time_us([&] { k_read<<<1024, 256>>>((const float4*)d, n4, d + (200 << 18)); }, N));
    printf("write+read %3ld MB alternating: %.2f us/pair\n", mb, time_us([&] {
      k_write<<<1024, 256>>>((float4*)d, n4); k_read<<<1024, 256>>>((const float4*)d, n4, d + (200 << 18)); }, N));
  }
  return 0;
}
