// Training-step glue kernels: flat AdamW update over the packed parameter buffer.
#include "nlam_common.h"

// torch.optim.AdamW semantics (decoupled weight decay, bias correction), the
// optimiser the reference configures at ar_model.py:191-195.  g is scaled by
// grad_scale first (1/world_size after a SUM all-reduce).
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                             float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                             float beta1, float beta2, float eps, float wd, float bc1,
                             float bc2_sqrt, float grad_scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}

extern "C" int nlam_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay,
                               int64_t step, float grad_scale, void* stream) {
  if (n <= 0) return 0;
  NLAM_REQUIRE(step >= 1, "adamw: step must be >= 1");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  adamw_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(
      p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
  NLAM_CHECK_LAUNCH("adamw");
  return 0;
}
