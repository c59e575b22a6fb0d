// Shared helpers for libnlam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nlam_hip.h"

void nlam_set_error(const char* fmt, ...);

#define NLAM_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      nlam_set_error(__VA_ARGS__);     \
      return 1;                        \
    }                                  \
  } while (0)

#define NLAM_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      nlam_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return 2;                                                              \
    }                                                                        \
  } while (0)

// Raise a kernel's dynamic-LDS limit to the full 160 KiB of a gfx950 CU.  The attribute is
// per device: remembered per (kernel, device), thread-safe, and the HIP status is checked.
int nlam_enable_big_lds(const void* kern, const char* name);
#define NLAM_BIG_LDS(kern, name)                                        \
  do {                                                                  \
    if (nlam_enable_big_lds((const void*)(kern), name) != 0) return 2;  \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// GEMM arithmetic of the fused kernels: NLAM_MFMA = fp32 | bf16x3 (default) | bf16 in the
// environment, read once per process; any other value aborts the process with a message (the
// Python binding raises before that, _lib.py).  nlam_mfma_b3(): one of the two bf16 forms
// (fused_bf16x3.h) rather than the exact fp32 MFMA.  The generic kernels (generic_ops.hip)
// always use the exact fp32 MFMA.
bool nlam_mfma_b3();
// bf16 MFMA terms per fp32 product: 0 = exact fp32 MFMA, 3 = split-bf16 (NLAM_MFMA=bf16x3, the
// default), 1 = plain bf16 products with fp32 accumulate (NLAM_MFMA=bf16: the reference's
// `--precision bf16-mixed` arithmetic; fp32 storage, LayerNorm, residuals and aggregates).
// The d = 64 kernels run their bf16x3 form in both bf16 modes.
int nlam_mfma_terms();

static inline bool nlam_aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// sigmoid via v_exp_f32 + v_rcp_f32 (1 ulp each) instead of an IEEE division
// sequence (~10 VALU instructions): the silu / silu' evaluations are the bulk of the
// non-MFMA work of the fused kernels.  Relative error ~2e-7, far inside the 1e-4 parity bar.
__device__ __forceinline__ float nlam_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float nlam_silu(float x) { return x * nlam_sigmoid(x); }
// d/dx [x sigmoid(x)] = s (1 + x (1 - s))
__device__ __forceinline__ float nlam_silu_grad(float x) {
  float s = nlam_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}

// x[l] + x[l ^ 16] / x[l] + x[l ^ 32] in every lane through gfx950's v_permlane16_swap /
// v_permlane32_swap (a VALU half-exchange: odd 16-lane rows of the first operand <-> even rows of
// the second; upper 32 lanes of the first <-> lower 32 of the second) instead of __shfl_xor, which
// compiles to ds_bpermute_b32 -- a round trip through the LDS crossbar on the critical path of
// every LayerNorm (tools/permlane_probe.hip: same sums, bit for bit).  Inline asm: with both
// operands holding the same value hipcc 7.2 folds the builtin's two results into one register;
// `s_nop 1` covers the "VALU write -> v_permlane read" hazard the assembler cannot see.
__device__ __forceinline__ float lane_xor16_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float lane_xor32_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
