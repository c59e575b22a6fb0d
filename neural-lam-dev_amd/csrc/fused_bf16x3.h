// Split-bf16 ("bf16x3") emulation of the fp32 GEMMs on the bf16 matrix cores.
//   x = hi + lo (+ 2^-18 |x|),  hi = bf16(x), lo = bf16(x - hi)
//   a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi           (dropped: a_lo b_lo ~ 2^-18 |a b|)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (32 cycles for K = 16, against 8 x 64
// cycles of v_mfma_f32_32x32x2_f32): 3/16 of the fp32 MFMA time at ~2^-16 relative error per
// product -- tighter than the TF32 the reference itself trains with on GPUs
// (train_model.py:246-248).  The accumulator layout is the fp32 one, so the row-on-lane
// kernels keep their structure.
//
// Weight image in LDS: two bf16 planes (hi, lo), natural [n][k] order, row pitch K + 4
// elements (K/2 + 2 dwords = 2 x odd: the 8-byte row reads of a 32-lane half are
// conflict-free).  One image serves W . x (row reads, 2 x ds_read_b64 per fragment) and
// W^T . g (gfx950 transposed reads, 2 x ds_read_b64_tr_b16 per fragment).
#pragma once
#include "fused_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct B3Image {
  __bf16* hi;
  __bf16* lo;
  int pitch;   // elements
};
__host__ __device__ constexpr int b3_pitch(int k_pad32) { return k_pad32 + 4; }
// bytes of one image with n_pad rows
__host__ __device__ constexpr size_t b3_image_bytes(int n_pad, int k_pad32) {
  return (size_t)2 * n_pad * b3_pitch(k_pad32) * sizeof(__bf16);
}
__device__ __forceinline__ B3Image b3_image(void* base, int n_pad, int k_pad32) {
  B3Image im;
  im.pitch = b3_pitch(k_pad32);
  im.hi = reinterpret_cast<__bf16*>(base);
  im.lo = im.hi + n_pad * im.pitch;
  return im;
}

__device__ __forceinline__ void b3_split4(const f32x4& v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)v[j];
    lo[j] = (__bf16)(v[j] - (float)hi[j]);
  }
}

// W (n_out x k_in fp32, row-major) -> image rows [row0, row0 + n_pad), zero padded to
// k_pad32 columns.  float4 path needs k_in % 4 == 0 and 16-byte aligned rows.
__device__ __forceinline__ void load_weight_lds_b3(const B3Image& im, int row0,
                                                   const float* __restrict__ W, int64_t ldW,
                                                   int n_out, int k_in, int n_pad, int k_pad32,
                                                   int tid, int nthreads) {
  const bool vec = (k_in % 4 == 0) && (ldW % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0);
  if (vec) {
    const int cpr = k_pad32 >> 2;
    const int total = n_pad * cpr;
    for (int base = 0; base < total; base += 8 * nthreads) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * nthreads + tid;
        const int i = idx / cpr, c = idx - i * cpr;
        const bool ok = idx < total && i < n_out && 4 * c < k_in;
        v[u] = ok ? *reinterpret_cast<const f32x4*>(W + (int64_t)i * ldW + 4 * c)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * nthreads + tid;
        if (idx < total) {
          const int i = idx / cpr, c = idx - i * cpr;
          bf16x4 hi, lo;
          b3_split4(v[u], hi, lo);
          *reinterpret_cast<bf16x4*>(im.hi + (row0 + i) * im.pitch + 4 * c) = hi;
          *reinterpret_cast<bf16x4*>(im.lo + (row0 + i) * im.pitch + 4 * c) = lo;
        }
      }
    }
  } else {
    for (int idx = tid; idx < n_pad * k_pad32; idx += nthreads) {
      const int i = idx / k_pad32, k = idx - i * k_pad32;
      const float v = (i < n_out && k < k_in) ? W[(int64_t)i * ldW + k] : 0.f;
      const __bf16 h = (__bf16)v;
      im.hi[(row0 + i) * im.pitch + k] = h;
      im.lo[(row0 + i) * im.pitch + k] = (__bf16)(v - (float)h);
    }
  }
}

// eight consecutive accumulator registers -> hi / lo bf16 fragments (the B operand of the
// K = 16 step that covers features 16 s .. 16 s + 15 of a 32-feature block: lane half h
// holds features 16 s + 4 h + {0..3} and 16 s + 8 + 4 h + {0..3} in registers 8 s .. 8 s + 7)
__device__ __forceinline__ void b3_split(const f32x16& x, int r0, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float v = x[r0 + i];
    hi[i] = (__bf16)v;
    lo[i] = (__bf16)(v - (float)hi[i]);
  }
}
__device__ __forceinline__ bf16x8 b3_join(const bf16x4& a, const bf16x4& b) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
  return r;
}
// A fragment of W . x: row `row`, the eight k of step (kb, s) for lane half h
__device__ __forceinline__ bf16x8 b3_row_frag(const __bf16* __restrict__ plane, int pitch, int row,
                                              int kb, int s, int h) {
  const __bf16* p = plane + row * pitch + 32 * kb + 16 * s + 4 * h;
  return b3_join(*reinterpret_cast<const bf16x4*>(p), *reinterpret_cast<const bf16x4*>(p + 8));
}
#define B3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// out[nb] += W[rows 32 nb .. +31][cols k0 + 32 KB features] . IN, IN in accumulator layout.
template <int NB, int KB>
__device__ __forceinline__ void gemm_acc_b3(f32x16 (&out)[NB], const B3Image& W, int kb0,
                                            const f32x16 (&in)[KB], int lane) {
  const int t = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 bh, bl;
      b3_split(in[kb], 8 * s, bh, bl);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8 ah = b3_row_frag(W.hi, W.pitch, 32 * nb + t, kb0 + kb, s, h);
        const bf16x8 al = b3_row_frag(W.lo, W.pitch, 32 * nb + t, kb0 + kb, s, h);
        out[nb] = B3_MFMA(ah, bh, out[nb]);
        out[nb] = B3_MFMA(ah, bl, out[nb]);
        out[nb] = B3_MFMA(al, bh, out[nb]);
      }
    }
  }
}
