// Shape-generic fp32 kernels of libnlam_hip.so (any hidden_dim / MLP depth).
// These are the building blocks of the generic InteractionNet / MLP path; the
// fused kernels (fused_*.hip) replace them for the BASELINE shapes.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "nlam_common.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void nlam_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* nlam_last_error(void) { return g_err; }

#include <cctype>
#include <atomic>
#include <string>
#include <mutex>
#include <utility>
#include <vector>
int nlam_enable_big_lds(const void* kern, const char* name) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& kd : done)
    if (kd.first == kern && kd.second == dev) return 0;
  const hipError_t e =
      hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    nlam_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize, 160 KiB) failed on "
                   "device %d: %s", name, dev, hipGetErrorString(e));
    return 2;
  }
  done.emplace_back(kern, dev);
  return 0;
}
extern "C" int nlam_abi_version(void) { return NLAM_ABI_VERSION; }

// GEMM arithmetic of the fused kernels (see fused_bf16x3.h): NLAM_MFMA=fp32 | bf16x3
#define NLAM_MFMA_DEFAULT_MODE 1
static int mfma_mode_from_env() {
  const char* e = getenv("NLAM_MFMA");
  if (e == nullptr || e[0] == 0) return NLAM_MFMA_DEFAULT_MODE;
  std::string v(e);
  for (char& c : v) c = (char)tolower((unsigned char)c);
  if (v == "fp32") return 0;
  if (v == "bf16x3" || v == "b3") return 1;
  if (v == "bf16") return 2;
  // a typo must not silently select an arithmetic (it used to fall through to fp32 / bf16)
  fprintf(stderr, "libnlam_hip: NLAM_MFMA=\"%s\" is not one of fp32 | bf16x3 | bf16\n", e);
  abort();
  return -1;
}
// The arithmetic is a property of a RUN (the reference takes `--precision` per run,
// train_model.py:72-77,285): the environment gives the initial value, nlam_set_mfma_mode()
// changes it between runs of one process.  Every launcher reads it at launch time.
static std::atomic<int>& mfma_mode_cell() {
  static std::atomic<int> mode{mfma_mode_from_env()};
  return mode;
}
static int nlam_mfma_mode_value() { return mfma_mode_cell().load(std::memory_order_relaxed); }
extern "C" int nlam_set_mfma_mode(int mode) {
  NLAM_REQUIRE(mode >= 0 && mode <= 2, "nlam_set_mfma_mode: mode %d not in {0 fp32, 1 bf16x3, 2 bf16}", mode);
  mfma_mode_cell().store(mode, std::memory_order_relaxed);
  return 0;
}
bool nlam_mfma_b3() { return nlam_mfma_mode_value() != 0; }
int nlam_mfma_terms() {
  const int m = nlam_mfma_mode_value();
  return m == 0 ? 0 : (m == 1 ? 3 : 1);
}
extern "C" int nlam_mfma_mode(void) { return nlam_mfma_mode_value(); }

// ------------------------------------------------------------------- GEMM
// 64x64 output tile per 256-thread workgroup, 4 waves each owning a 32x32
// sub-tile computed with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
// Operands may have arbitrary element strides; tiles are staged through LDS
// with zero fill at the edges.
#define G_TM 64
#define G_TN 64
#define G_TK 16

typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
// BF16 (NLAM_MFMA=bf16, the bf16-mixed arithmetic of the reference's `--precision bf16-mixed`):
// operands rounded to bf16 when the fragments are read from LDS, one
// v_mfma_f32_32x32x16_bf16 per 16-deep K tile, fp32 accumulate; storage stays fp32.
template <bool BF16>
__global__ __launch_bounds__(256) void gemm_kernel(
    int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t sa_i,
    int64_t sa_k, const float* __restrict__ B, int64_t sb_k, int64_t sb_j,
    const float* __restrict__ bias, float* __restrict__ C, int64_t ldc,
    int accumulate, int splitk, int64_t kchunk, float* __restrict__ ws) {
  __shared__ float As[G_TK][G_TM + 1];
  __shared__ float Bs[G_TK][G_TN + 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.x * G_TM;
  const int64_t j0 = (int64_t)blockIdx.y * G_TN;
  const int z = blockIdx.z;
  const int64_t kbeg = (int64_t)z * kchunk;
  int64_t kend = kbeg + kchunk;
  if (kend > K) kend = K;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int64_t kt = kbeg; kt < kend; kt += G_TK) {
    // A tile: 64 (i) x 16 (k)
    if (sa_k == 1) {
      const int k = tid & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = (tid >> 4) + 16 * r;
        const int64_t gi = i0 + i, gk = kt + k;
        As[k][i] = (gi < M && gk < kend) ? A[gi * sa_i + gk] : 0.f;
      }
    } else {
      const int i = tid & 63;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = (tid >> 6) + 4 * r;
        const int64_t gi = i0 + i, gk = kt + k;
        As[k][i] = (gi < M && gk < kend) ? A[gi * sa_i + gk * sa_k] : 0.f;
      }
    }
    // B tile: 16 (k) x 64 (j)
    if (sb_k == 1 && sb_j != 1) {
      const int k = tid & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = (tid >> 4) + 16 * r;
        const int64_t gj = j0 + j, gk = kt + k;
        Bs[k][j] = (gj < N && gk < kend) ? B[gk + gj * sb_j] : 0.f;
      }
    } else {
      const int j = tid & 63;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = (tid >> 6) + 4 * r;
        const int64_t gj = j0 + j, gk = kt + k;
        Bs[k][j] = (gj < N && gk < kend) ? B[gk * sb_k + gj * sb_j] : 0.f;
      }
    }
    __syncthreads();
    if constexpr (BF16) {
      g_bf16x8 a, b;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = (__bf16)As[8 * (lane >> 5) + u][wr * 32 + (lane & 31)];
        b[u] = (__bf16)Bs[8 * (lane >> 5) + u][wc * 32 + (lane & 31)];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < G_TK / 2; ++kk) {
        const float a = As[kk * 2 + (lane >> 5)][wr * 32 + (lane & 31)];
        const float b = Bs[kk * 2 + (lane >> 5)][wc * 32 + (lane & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const int64_t j = j0 + wc * 32 + (lane & 31);
  if (j >= N) return;
  const float bj = (bias != nullptr && splitk == 1) ? bias[j] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t i = i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (i < M) {
      if (splitk == 1) {
        float v = acc[r] + bj;
        if (accumulate) v += C[i * ldc + j];
        C[i * ldc + j] = v;
      } else {
        ws[((int64_t)z * M + i) * N + j] = acc[r];
      }
    }
  }
}

// Larger-tile variant for the big shapes of the generic path (hidden_dim 128/256,
// SplitMLPs): 128x128 output tile, 4 waves each owning 64x64 (2x2 MFMA blocks), K
// tiles of 16 prefetched into registers while the previous tile is multiplied.
#define H_TM 128
#define H_TN 128
#define H_TK 16
#define H_LD (H_TM + 4)

template <bool BF16>
__global__ __launch_bounds__(256) void gemm128_kernel(
    int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t sa_i, int64_t sa_k,
    const float* __restrict__ B, int64_t sb_k, int64_t sb_j, const float* __restrict__ bias,
    float* __restrict__ C, int64_t ldc, int accumulate, int splitk, int64_t kchunk,
    float* __restrict__ ws) {
  __shared__ float As[H_TK][H_LD];
  __shared__ float Bs[H_TK][H_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.x * H_TM, j0 = (int64_t)blockIdx.y * H_TN;
  const int z = blockIdx.z;
  const int64_t kbeg = (int64_t)z * kchunk;
  int64_t kend = kbeg + kchunk;
  if (kend > K) kend = K;
  // loader maps: 8 elements per thread per operand, along the contiguous dimension
  const bool a_kc = sa_k == 1;   // A rows contiguous in k
  const bool b_kc = sb_k == 1 && sb_j != 1;
  const int a_i = a_kc ? (tid >> 1) : ((tid & 15) * 8), a_k = a_kc ? ((tid & 1) * 8) : (tid >> 4);
  const int b_j = b_kc ? (tid >> 1) : ((tid & 15) * 8), b_k = b_kc ? ((tid & 1) * 8) : (tid >> 4);
  float ra[8], rb[8];
  // interior chunks whose 8 elements are contiguous and 16-byte aligned are two float4
  // loads; edges and odd strides fall back to guarded scalar loads
  auto load8 = [&](float (&r)[8], const float* __restrict__ P, int64_t s_row, int64_t s_k,
                   bool kc, int64_t row, int64_t k0, int64_t row_lim) {
    const int64_t rlast = row + (kc ? 0 : 7), klast = k0 + (kc ? 7 : 0);
    const float* ptr = P + row * s_row + k0 * s_k;
    if ((kc ? s_k : s_row) == 1 && rlast < row_lim && klast < kend &&
        (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0) {
      const f32x4 v0 = reinterpret_cast<const f32x4*>(ptr)[0];
      const f32x4 v1 = reinterpret_cast<const f32x4*>(ptr)[1];
#pragma unroll
      for (int u = 0; u < 4; ++u) { r[u] = v0[u]; r[4 + u] = v1[u]; }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t gr = row + (kc ? 0 : u), gk = k0 + (kc ? u : 0);
        r[u] = (gr < row_lim && gk < kend) ? P[gr * s_row + gk * s_k] : 0.f;
      }
    }
  };
  auto load_tile = [&](int64_t kt) {
    // A: element (i, k) at A[i*sa_i + k*sa_k]; contiguous along k (a_kc) or along i
    load8(ra, A, sa_i, sa_k, a_kc, i0 + a_i, kt + a_k, M);
    // B: element (k, j) at B[k*sb_k + j*sb_j]; contiguous along k (b_kc) or along j
    load8(rb, B, sb_j, sb_k, b_kc, j0 + b_j, kt + b_k, N);
  };
  auto put_tile = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      As[a_k + (a_kc ? u : 0)][a_i + (a_kc ? 0 : u)] = ra[u];
      Bs[b_k + (b_kc ? u : 0)][b_j + (b_kc ? 0 : u)] = rb[u];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  if (kbeg < kend) load_tile(kbeg);
  for (int64_t kt = kbeg; kt < kend; kt += H_TK) {
    __syncthreads();           // previous tile fully consumed
    put_tile();
    __syncthreads();
    if (kt + H_TK < kend) load_tile(kt + H_TK);   // in flight during the MFMAs below
    if constexpr (BF16) {
      const int kg = lane >> 5, c = lane & 31;
      g_bf16x8 a0, a1, b0, b1;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0[u] = (__bf16)As[8 * kg + u][wr * 64 + c];
        a1[u] = (__bf16)As[8 * kg + u][wr * 64 + 32 + c];
        b0[u] = (__bf16)Bs[8 * kg + u][wc * 64 + c];
        b1[u] = (__bf16)Bs[8 * kg + u][wc * 64 + 32 + c];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < H_TK / 2; ++kk) {
        const int kr = kk * 2 + (lane >> 5), c = lane & 31;
        const float a0 = As[kr][wr * 64 + c], a1 = As[kr][wr * 64 + 32 + c];
        const float b0 = Bs[kr][wc * 64 + c], b1 = Bs[kr][wc * 64 + 32 + c];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int bj = 0; bj < 2; ++bj) {
    const int64_t j = j0 + wc * 64 + bj * 32 + (lane & 31);
    if (j >= N) continue;
    const float bv = (bias != nullptr && splitk == 1) ? bias[j] : 0.f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t i = i0 + wr * 64 + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (i < M) {
          if (splitk == 1) {
            float v = acc[bi][bj][r] + bv;
            if (accumulate) v += C[i * ldc + j];
            C[i * ldc + j] = v;
          } else {
            ws[((int64_t)z * M + i) * N + j] = acc[bi][bj][r];
          }
        }
      }
  }
}

__global__ void gemm_splitk_reduce(int64_t M, int64_t N, const float* __restrict__ ws,
                                   int splitk, const float* __restrict__ bias,
                                   float* __restrict__ C, int64_t ldc, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  const int64_t i = idx / N, j = idx % N;
  float s = 0.f;
  for (int z = 0; z < splitk; ++z) s += ws[(int64_t)z * M * N + idx];
  if (bias) s += bias[j];
  if (accumulate) s += C[i * ldc + j];
  C[i * ldc + j] = s;
}

extern "C" int nlam_gemm(int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_i,
                         int64_t sa_k, const float* B, int64_t sb_k, int64_t sb_j,
                         const float* bias, float* C, int64_t ldc, int accumulate,
                         int splitk, float* workspace, void* stream) {
  NLAM_REQUIRE(M > 0 && N > 0 && K >= 0, "nlam_gemm: bad shape %ld %ld %ld", (long)M,
               (long)N, (long)K);
  NLAM_REQUIRE(ldc >= N, "nlam_gemm: ldc < N");
  if (splitk < 1) splitk = 1;
  NLAM_REQUIRE(splitk == 1 || workspace != nullptr, "nlam_gemm: splitk needs workspace");
  int64_t kchunk = (K + splitk - 1) / splitk;
  kchunk = ((kchunk + G_TK - 1) / G_TK) * G_TK;
  if (kchunk == 0) kchunk = G_TK;
  hipStream_t s = (hipStream_t)stream;
  NLAM_REQUIRE(splitk <= 65535, "nlam_gemm: grid too large");
  const bool bf16 = nlam_mfma_terms() == 1;   // NLAM_MFMA=bf16: bf16 products, fp32 accumulate
  if (M >= 96 && N >= 96) {   // big shapes: 128x128 tiles with register prefetch
    const int64_t gx = (M + H_TM - 1) / H_TM, gy = (N + H_TN - 1) / H_TN;
    NLAM_REQUIRE(gy <= 65535, "nlam_gemm: grid too large");
    if (bf16)
      gemm128_kernel<true><<<dim3((unsigned)gx, (unsigned)gy, (unsigned)splitk), 256, 0, s>>>(
          M, N, K, A, sa_i, sa_k, B, sb_k, sb_j, bias, C, ldc, accumulate, splitk, kchunk,
          workspace);
    else
      gemm128_kernel<false><<<dim3((unsigned)gx, (unsigned)gy, (unsigned)splitk), 256, 0, s>>>(
          M, N, K, A, sa_i, sa_k, B, sb_k, sb_j, bias, C, ldc, accumulate, splitk, kchunk,
          workspace);
    NLAM_CHECK_LAUNCH("gemm128_kernel");
  } else {
    const int64_t gx = (M + G_TM - 1) / G_TM, gy = (N + G_TN - 1) / G_TN;
    NLAM_REQUIRE(gy <= 65535, "nlam_gemm: grid too large");
    if (bf16)
      gemm_kernel<true><<<dim3((unsigned)gx, (unsigned)gy, (unsigned)splitk), 256, 0, s>>>(
          M, N, K, A, sa_i, sa_k, B, sb_k, sb_j, bias, C, ldc, accumulate, splitk, kchunk,
          workspace);
    else
      gemm_kernel<false><<<dim3((unsigned)gx, (unsigned)gy, (unsigned)splitk), 256, 0, s>>>(
          M, N, K, A, sa_i, sa_k, B, sb_k, sb_j, bias, C, ldc, accumulate, splitk, kchunk,
          workspace);
    NLAM_CHECK_LAUNCH("gemm_kernel");
  }
  if (splitk > 1) {
    const int64_t n = M * N;
    gemm_splitk_reduce<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(M, N, workspace, splitk,
                                                                   bias, C, ldc, accumulate);
    NLAM_CHECK_LAUNCH("gemm_splitk_reduce");
  }
  return 0;
}

// ------------------------------------------------------------------- SiLU
__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4,
                                int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = nlam_silu(v[c]);
    reinterpret_cast<f32x4*>(y)[i] = o;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = nlam_silu(x[i]);
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                float* __restrict__ gx, int64_t n4, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 g = reinterpret_cast<const f32x4*>(gy)[i];
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = g[c] * nlam_silu_grad(v[c]);
    reinterpret_cast<f32x4*>(gx)[i] = o;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    gx[i] = gy[i] * nlam_silu_grad(x[i]);
}
static inline unsigned ew_grid(int64_t work) {
  int64_t b = (work + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}
extern "C" int nlam_silu_fwd(const float* x, float* y, int64_t n, void* stream) {
  if (n <= 0) return 0;
  const bool v = nlam_aligned16(x) && nlam_aligned16(y);
  const int64_t n4 = v ? n / 4 : 0;
  silu_fwd_kernel<<<ew_grid(v ? n4 + 3 : n), 256, 0, (hipStream_t)stream>>>(x, y, n4, n);
  NLAM_CHECK_LAUNCH("silu_fwd");
  return 0;
}
extern "C" int nlam_silu_bwd(const float* x, const float* gy, float* gx, int64_t n,
                             void* stream) {
  if (n <= 0) return 0;
  const bool v = nlam_aligned16(x) && nlam_aligned16(gy) && nlam_aligned16(gx);
  const int64_t n4 = v ? n / 4 : 0;
  silu_bwd_kernel<<<ew_grid(v ? n4 + 3 : n), 256, 0, (hipStream_t)stream>>>(x, gy, gx, n4, n);
  NLAM_CHECK_LAUNCH("silu_bwd");
  return 0;
}

// -------------------------------------------------------------- LayerNorm
// One wavefront per row, lanes stride over the d columns.
#define LN_EPS 1e-5f
#define LN_MAXT 16  // columns per lane: d <= 64 * LN_MAXT

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const float* __restrict__ z, int64_t ldz, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ res, int64_t ldres,
    float* __restrict__ y, int64_t ldy, int64_t rows, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float inv_d = 1.0f / (float)d;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    const float* zr = z + r * ldz;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += zr[c];
    const float mean = wave_sum(s) * inv_d;
    float v = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float t = zr[c] - mean;
      v += t * t;
    }
    const float rstd = rsqrtf(wave_sum(v) * inv_d + LN_EPS);
    for (int c = lane; c < d; c += 64) {
      float o = (zr[c] - mean) * rstd * gamma[c] + beta[c];
      if (res) o += res[r * ldres + c];
      y[r * ldy + c] = o;
    }
  }
}

extern "C" int nlam_layernorm_fwd(const float* z, int64_t ldz, const float* gamma,
                                  const float* beta, const float* res, int64_t ldres, float* y,
                                  int64_t ldy, int64_t rows, int64_t d, void* stream) {
  if (rows <= 0) return 0;
  NLAM_REQUIRE(d >= 1 && d <= 64 * LN_MAXT, "layernorm: d=%ld unsupported", (long)d);
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  layernorm_fwd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(
      z, ldz, gamma, beta, res, ldres, y, ldy, rows, (int)d);
  NLAM_CHECK_LAUNCH("layernorm_fwd");
  return 0;
}

extern "C" int64_t nlam_layernorm_bwd_blocks(int64_t rows) {
  int64_t b = (rows + 63) / 64;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return b;
}

// Block b owns rows [b*rpb, (b+1)*rpb); its 4 waves take rows round-robin and
// keep per-column dgamma/dbeta partials in registers, merged through LDS.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ z, int64_t ldz, const float* __restrict__ gamma,
    const float* __restrict__ gy, int64_t ldgy, float* __restrict__ gz, int64_t ldgz,
    float* __restrict__ partial, int64_t rows, int d, int64_t rpb) {
  __shared__ float red[2][4][64 * LN_MAXT / 4];  // sized for d <= 256 per pass; see loop
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t rbeg = (int64_t)blockIdx.x * rpb;
  int64_t rend = rbeg + rpb;
  if (rend > rows) rend = rows;
  const float inv_d = 1.0f / (float)d;
  float dg[LN_MAXT], db[LN_MAXT];
#pragma unroll
  for (int t = 0; t < LN_MAXT; ++t) dg[t] = db[t] = 0.f;
  for (int64_t r = rbeg + wave; r < rend; r += 4) {
    const float* zr = z + r * ldz;
    const float* gr = gy + r * ldgy;
    // the row of z and of gy is read ONCE, both in flight together, and kept in registers
    float zv[LN_MAXT], gv[LN_MAXT], gmv[LN_MAXT];
#pragma unroll
    for (int t = 0; t < LN_MAXT; ++t) {
      const int c = lane + 64 * t;
      const bool on = c < d;
      zv[t] = on ? zr[c] : 0.f;
      gv[t] = on ? gr[c] : 0.f;
      gmv[t] = on ? gamma[c] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXT; ++t) s += zv[t];
    const float mean = wave_sum(s) * inv_d;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXT; ++t) {
      const float dt = (lane + 64 * t < d) ? zv[t] - mean : 0.f;
      v += dt * dt;
    }
    const float rstd = rsqrtf(wave_sum(v) * inv_d + LN_EPS);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < LN_MAXT; ++t) {
      const float xh = (zv[t] - mean) * rstd;
      const float g = gv[t] * gmv[t];       // 0 beyond d
      s1 += g;
      s2 += g * xh;
    }
    const float m1 = wave_sum(s1) * inv_d;
    const float m2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int t = 0; t < LN_MAXT; ++t) {
      const int c = lane + 64 * t;
      if (c < d) {
        const float xh = (zv[t] - mean) * rstd;
        const float go = gv[t];
        gz[r * ldgz + c] = rstd * (go * gmv[t] - m1 - xh * m2);
        dg[t] += go * xh;
        db[t] += go;
      }
    }
  }
  // merge the 4 waves, 256 columns at a time
  float* pg = partial + (int64_t)blockIdx.x * 2 * d;
#pragma unroll
  for (int t0 = 0; t0 < LN_MAXT; t0 += 4) {
    if (t0 * 64 >= d) break;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      red[0][wave][t * 64 + lane] = dg[t0 + t];
      red[1][wave][t * 64 + lane] = db[t0 + t];
    }
    __syncthreads();
    const int c = t0 * 64 + threadIdx.x;
    if (c < d) {
      const int cl = threadIdx.x;
      pg[c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
      pg[d + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    }
  }
}

// out[c] (+)= sum_b partial[b*stride + c], fixed order
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int64_t nblocks,
                                       int64_t stride, float* __restrict__ out, int n,
                                       int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float s = 0.f;
  for (int64_t b = 0; b < nblocks; ++b) s += partial[b * stride + c];
  if (accumulate) s += out[c];
  out[c] = s;
}

extern "C" int nlam_layernorm_bwd(const float* z, int64_t ldz, const float* gamma,
                                  const float* gy, int64_t ldgy, float* gz, int64_t ldgz,
                                  float* dgamma, float* dbeta, int accumulate, float* partial,
                                  int64_t rows, int64_t d, void* stream) {
  if (rows <= 0) return 0;
  NLAM_REQUIRE(d >= 1 && d <= 64 * LN_MAXT, "layernorm_bwd: d=%ld unsupported", (long)d);
  NLAM_REQUIRE(partial != nullptr, "layernorm_bwd: partial workspace is null");
  const int64_t nb = nlam_layernorm_bwd_blocks(rows);
  const int64_t rpb = (rows + nb - 1) / nb;
  hipStream_t s = (hipStream_t)stream;
  layernorm_bwd_kernel<<<(unsigned)nb, 256, 0, s>>>(z, ldz, gamma, gy, ldgy, gz, ldgz, partial,
                                                    rows, (int)d, rpb);
  NLAM_CHECK_LAUNCH("layernorm_bwd");
  // fixed-order parallel reduction of the per-block partials (fused_mlp.hip)
  int rc = nlam_reduce_slabs(partial, nb, 2 * d, d, dgamma, accumulate, stream);
  if (rc) return rc;
  return nlam_reduce_slabs(partial + d, nb, 2 * d, d, dbeta, accumulate, stream);
}

// ----------------------------------------------------------------- colsum
extern "C" int64_t nlam_colsum_blocks(int64_t rows) {
  int64_t b = (rows + 255) / 256;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return b;
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ldx,
                                                     float* __restrict__ partial, int64_t rows,
                                                     int d, int64_t rpb) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t rbeg = (int64_t)blockIdx.x * rpb;
  int64_t rend = rbeg + rpb;
  if (rend > rows) rend = rows;
  float acc[LN_MAXT];
#pragma unroll
  for (int t = 0; t < LN_MAXT; ++t) acc[t] = 0.f;
  // eight rows per trip: their loads are issued together (clamped + masked), the adds keep
  // the row order
  for (int64_t r = rbeg + wave; r < rend; r += 4 * 8) {
#pragma unroll
    for (int t = 0; t < LN_MAXT; ++t) {
      const int c = lane + 64 * t;
      if (c < d) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t ru = r + 4 * u;
          v[u] = x[(ru < rend ? ru : r) * ldx + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (r + 4 * u < rend) acc[t] += v[u];
      }
    }
  }
  float* pg = partial + (int64_t)blockIdx.x * d;
#pragma unroll
  for (int t0 = 0; t0 < LN_MAXT; t0 += 4) {
    if (t0 * 64 >= d) break;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) red[wave][t * 64 + lane] = acc[t0 + t];
    __syncthreads();
    const int c = t0 * 64 + threadIdx.x;
    if (c < d) {
      const int cl = threadIdx.x;
      pg[c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    }
  }
}
extern "C" int nlam_colsum(const float* x, int64_t ldx, float* out, int accumulate,
                           float* partial, int64_t rows, int64_t d, void* stream) {
  NLAM_REQUIRE(d >= 1 && d <= 64 * LN_MAXT, "colsum: d=%ld unsupported", (long)d);
  NLAM_REQUIRE(partial != nullptr, "colsum: partial workspace is null");
  if (rows <= 0) return 0;
  const int64_t nb = nlam_colsum_blocks(rows);
  const int64_t rpb = (rows + nb - 1) / nb;
  hipStream_t s = (hipStream_t)stream;
  colsum_kernel<<<(unsigned)nb, 256, 0, s>>>(x, ldx, partial, rows, (int)d, rpb);
  NLAM_CHECK_LAUNCH("colsum");
  return nlam_reduce_slabs(partial, nb, d, d, out, accumulate, stream);
}

// ----------------------------------------------------------- gather / copy
// VEC = 4: one thread moves a float4; VEC = 1: scalar fallback.
template <int VEC>
__global__ void gather_rows_kernel(const float* __restrict__ x, int64_t x_bstride, int64_t ldx,
                                   const int32_t* __restrict__ idx,
                                   const float* __restrict__ row_scale, float* __restrict__ out,
                                   int64_t out_bstride, int64_t ldout, int64_t B, int64_t n_out,
                                   int dv) {
  const int64_t total = B * n_out * dv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    const int c = (int)(g % dv);
    const int64_t row = g / dv;
    const int64_t k = row % n_out, b = row / n_out;
    const int64_t src = idx ? (int64_t)idx[k] : k;
    const float sc = row_scale ? row_scale[src] : 1.0f;
    if (VEC == 4) {
      reinterpret_cast<f32x4*>(out + b * out_bstride + k * ldout)[c] =
          reinterpret_cast<const f32x4*>(x + b * x_bstride + src * ldx)[c] * sc;
    } else {
      out[b * out_bstride + k * ldout + c] = x[b * x_bstride + src * ldx + c] * sc;
    }
  }
}

static int launch_gather(const float* x, int64_t x_bstride, int64_t ldx, const int32_t* idx,
                         const float* row_scale, float* out, int64_t out_bstride, int64_t ldout, int64_t B,
                         int64_t n_out, int64_t d, void* stream, const char* name) {
  if (B <= 0 || n_out <= 0 || d <= 0) return 0;
  const bool vec = (d % 4 == 0) && (ldx % 4 == 0) && (ldout % 4 == 0) && (x_bstride % 4 == 0) &&
                   (out_bstride % 4 == 0) && nlam_aligned16(x) && nlam_aligned16(out);
  hipStream_t s = (hipStream_t)stream;
  if (vec) {
    const int dv = (int)(d / 4);
    gather_rows_kernel<4><<<ew_grid(B * n_out * dv), 256, 0, s>>>(
        x, x_bstride, ldx, idx, row_scale, out, out_bstride, ldout, B, n_out, dv);
  } else {
    gather_rows_kernel<1><<<ew_grid(B * n_out * d), 256, 0, s>>>(
        x, x_bstride, ldx, idx, row_scale, out, out_bstride, ldout, B, n_out, (int)d);
  }
  NLAM_CHECK_LAUNCH(name);
  return 0;
}

extern "C" int nlam_gather_rows(const float* x, int64_t x_bstride, int64_t ldx,
                                const int32_t* idx, const float* row_scale, float* out,
                                int64_t out_bstride,
                                int64_t ldout, int64_t B, int64_t n_out, int64_t d,
                                void* stream) {
  NLAM_REQUIRE(idx != nullptr, "gather_rows: idx is null");
  return launch_gather(x, x_bstride, ldx, idx, row_scale, out, out_bstride, ldout, B, n_out, d,
                       stream, "gather_rows");
}
extern "C" int nlam_copy_rows(const float* x, int64_t x_bstride, int64_t ldx, float* out,
                              int64_t out_bstride, int64_t ldout, int64_t B, int64_t rows,
                              int64_t d, void* stream) {
  return launch_gather(x, x_bstride, ldx, nullptr, nullptr, out, out_bstride, ldout, B, rows, d,
                       stream, "copy_rows");
}

// ------------------------------------------------------------ segment sum
// One wavefront per (batch, output row).  With float4 columns a row needs
// LPR = d/4 lanes, so R = 64/LPR source rows are in flight per load
// instruction; sub-group partials are combined with xor-shuffles (fixed order).
template <int LPR>
__global__ __launch_bounds__(256) void segment_sum_vec_kernel(
    const float* __restrict__ src, int64_t src_bstride, int64_t ldsrc,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ pos,
    const float* __restrict__ scale, float* __restrict__ out, int64_t out_bstride,
    int64_t ldout, int accumulate, int64_t B, int64_t n_out) {
  constexpr int R = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, c4 = lane % LPR;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= B * n_out) return;
  const int64_t b = w / n_out, i = w % n_out;
  const int beg = rowptr[i], end = rowptr[i + 1];
  const float* sb = src + b * src_bstride;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // 4 row loads in flight per sub-group (16 rows per wave at d = 64: one trip covers
  // a typical in-degree); rows past the segment end are clamped to a valid row and
  // masked, so the loads are unconditional and issue back to back
  for (int p = beg + sub; p < end; p += 4 * R) {
    int64_t r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = p + u * R < end ? p + u * R : p;
      r[u] = pos ? (int64_t)pos[pp] : pp;
    }
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const f32x4*>(sb + r[u] * ldsrc)[c4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p + u * R < end) acc += v[u];
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
  }
  if (sub == 0) {
    const float sc = scale ? scale[i] : 1.0f;
    f32x4* o = reinterpret_cast<f32x4*>(out + b * out_bstride + i * ldout) + c4;
    f32x4 v = acc * sc;
    if (accumulate) v += *o;
    *o = v;
  }
}

// Batch-folded form (B >= R): one wavefront per output row serves every sample.  The
// segment bounds and the position list are read once (wave-uniform) and each of the
// R = 64/LPR sub-groups walks the list for its own sample with 8 row loads in flight --
// a quarter of the waves and index fetches of the per-(b, row) form.  Rows are summed in
// list order (deterministic).
template <int LPR>
__global__ __launch_bounds__(256) void segment_sum_bfold_kernel(
    const float* __restrict__ src, int64_t src_bstride, int64_t ldsrc,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ pos,
    const float* __restrict__ scale, float* __restrict__ out, int64_t out_bstride,
    int64_t ldout, int accumulate, int64_t B, int64_t n_out) {
  constexpr int R = 64 / LPR;
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, c4 = lane % LPR;
  const int64_t i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (i >= n_out) return;
  const int beg = rowptr[i], end = rowptr[i + 1];
  const float sc = scale ? scale[i] : 1.0f;
  for (int64_t b0 = 0; b0 < B; b0 += R) {
    const int64_t b = b0 + sub;
    const bool valid = b < B;
    const float* sb = src + (valid ? b : B - 1) * src_bstride;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = beg; p < end; p += U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pp = p + u < end ? p + u : end - 1;
        const int64_t r = pos ? (int64_t)pos[pp] : pp;
        v[u] = reinterpret_cast<const f32x4*>(sb + r * ldsrc)[c4];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p + u < end) acc += v[u];
    }
    if (valid) {
      f32x4* o = reinterpret_cast<f32x4*>(out + b * out_bstride + i * ldout) + c4;
      f32x4 v = acc * sc;
      if (accumulate) v += *o;
      *o = v;
    }
  }
}

// Segment sums of every sample AND the batch sum of every listed row in one pass over src (the
// sender-side sums gPs[b][n] = sum of gh[b][e] over the out-edges of n, and dPe[e] = sum_b gh[b][e],
// the gradient of a batch-invariant edge term: every edge is in exactly one sender's list, so the
// walk over the lists reads each row of each sample once).  One wavefront per output row; its
// R = 64 / LPR sub-groups take the samples b = sub, sub + R, ...; eight list entries per trip.
// Sums in list order per sample, in sample order per row: deterministic.
template <int LPR, int NB>   // NB = ceil(B / R) <= 4
__global__ __launch_bounds__(256) void segment_sum_bsum_kernel(
    const float* __restrict__ src, int64_t src_bstride, int64_t ldsrc,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ pos,
    float* __restrict__ out, int64_t out_bstride, int64_t ldout,
    float* __restrict__ bsum, int64_t ldbsum, int64_t B, int64_t n_out) {
  constexpr int R = 64 / LPR;
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, c4 = lane % LPR;
  const int64_t i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (i >= n_out) return;
  const int beg = rowptr[i], end = rowptr[i + 1];
  f32x4 acc[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int p = beg; p < end; p += U) {
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = p + u < end ? p + u : end - 1;
      r[u] = pos ? (int64_t)pos[pp] : pp;
    }
    f32x4 es[U];
#pragma unroll
    for (int u = 0; u < U; ++u) es[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int64_t b = (int64_t)k * R + sub;
      const bool valid = b < B;
      const float* sb = src + (valid ? b : B - 1) * src_bstride;
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = reinterpret_cast<const f32x4*>(sb + r[u] * ldsrc)[c4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (valid && p + u < end) {
          acc[k] += v[u];
          es[u] += v[u];
        }
      }
    }
    // samples of the other sub-groups (fixed order: xor LPR, 2 LPR, ...)
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) es[u][c] += __shfl_xor(es[u][c], o, 64);
    if (sub == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p + u < end) reinterpret_cast<f32x4*>(bsum + r[u] * ldbsum)[c4] = es[u];
    }
  }
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int64_t b = (int64_t)k * R + sub;
    if (b < B) reinterpret_cast<f32x4*>(out + b * out_bstride + i * ldout)[c4] = acc[k];
  }
}

extern "C" int nlam_segment_sum_bsum(const float* src, int64_t src_bstride, int64_t ldsrc,
                                     const int32_t* rowptr, const int32_t* pos, float* out,
                                     int64_t out_bstride, int64_t ldout, float* bsum, int64_t ldbsum,
                                     int64_t B, int64_t n_out, int64_t d, void* stream) {
  if (B <= 0 || n_out <= 0) return 0;
  NLAM_REQUIRE(rowptr != nullptr && src != nullptr && out != nullptr && bsum != nullptr,
               "segment_sum_bsum: null operand");
  NLAM_REQUIRE((d == 64 || d == 128 || d == 256) && B * d <= 4 * 256,
               "segment_sum_bsum: d %ld (64, 128, 256) with B %ld (B * d <= 1024)", (long)d, (long)B);
  NLAM_REQUIRE((ldsrc % 4 == 0) && (ldout % 4 == 0) && (ldbsum % 4 == 0) && (src_bstride % 4 == 0) &&
                   (out_bstride % 4 == 0) && nlam_aligned16(src) && nlam_aligned16(out) &&
                   nlam_aligned16(bsum),
               "segment_sum_bsum: rows must be 16-byte aligned, pitches %% 4 == 0");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((n_out + 3) / 4);
#define SEG_BSUM(LPR, NB)                                                                          \
  segment_sum_bsum_kernel<LPR, NB><<<grid, 256, 0, s>>>(src, src_bstride, ldsrc, rowptr, pos, out, \
                                                        out_bstride, ldout, bsum, ldbsum, B, n_out)
  const int R = (int)(256 / d), nb = (int)((B + R - 1) / R);
  if (d == 64) {
    if (nb <= 1) SEG_BSUM(16, 1); else if (nb == 2) SEG_BSUM(16, 2); else SEG_BSUM(16, 4);
  } else if (d == 128) {
    if (nb <= 1) SEG_BSUM(32, 1); else if (nb == 2) SEG_BSUM(32, 2); else SEG_BSUM(32, 4);
  } else {
    if (nb <= 1) SEG_BSUM(64, 1); else if (nb == 2) SEG_BSUM(64, 2); else SEG_BSUM(64, 4);
  }
#undef SEG_BSUM
  NLAM_CHECK_LAUNCH("segment_sum_bsum");
  return 0;
}

__global__ __launch_bounds__(256) void segment_sum_scalar_kernel(
    const float* __restrict__ src, int64_t src_bstride, int64_t ldsrc,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ pos,
    const float* __restrict__ scale, float* __restrict__ out, int64_t out_bstride,
    int64_t ldout, int accumulate, int64_t B, int64_t n_out, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= B * n_out) return;
  const int64_t b = w / n_out, i = w % n_out;
  const int beg = rowptr[i], end = rowptr[i + 1];
  const float sc = scale ? scale[i] : 1.0f;
  for (int c = lane; c < d; c += 64) {
    float acc = 0.f;
    for (int p = beg; p < end; ++p) {
      const int64_t r = pos ? (int64_t)pos[p] : p;
      acc += src[b * src_bstride + r * ldsrc + c];
    }
    float* o = out + b * out_bstride + i * ldout + c;
    *o = accumulate ? (*o + acc * sc) : acc * sc;
  }
}

extern "C" int nlam_segment_sum(const float* src, int64_t src_bstride, int64_t ldsrc,
                                const int32_t* rowptr, const int32_t* pos, const float* scale,
                                float* out, int64_t out_bstride, int64_t ldout, int accumulate,
                                int64_t B, int64_t n_out, int64_t d, void* stream) {
  if (B <= 0 || n_out <= 0 || d <= 0) return 0;
  NLAM_REQUIRE(rowptr != nullptr, "segment_sum: rowptr is null");
  const int64_t waves = B * n_out;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (ldsrc % 4 == 0) && (ldout % 4 == 0) && (src_bstride % 4 == 0) &&
                   (out_bstride % 4 == 0) && nlam_aligned16(src) && nlam_aligned16(out);
#define SEG_LAUNCH(LPR)                                                                     \
  segment_sum_vec_kernel<LPR><<<grid, 256, 0, s>>>(src, src_bstride, ldsrc, rowptr, pos,   \
                                                   scale, out, out_bstride, ldout,         \
                                                   accumulate, B, n_out)
#define SEG_FOLD(LPR)                                                                       \
  segment_sum_bfold_kernel<LPR><<<(unsigned)((n_out + 3) / 4), 256, 0, s>>>(                \
      src, src_bstride, ldsrc, rowptr, pos, scale, out, out_bstride, ldout, accumulate, B, n_out)
  if (vec && d == 64 && B >= 4) {
    SEG_FOLD(16);
  } else if (vec && d == 128 && B >= 2) {
    SEG_FOLD(32);
  } else if (vec && d == 64) {
    SEG_LAUNCH(16);
  } else if (vec && d == 128) {
    SEG_LAUNCH(32);
  } else if (vec && d == 256) {
    SEG_LAUNCH(64);
  } else if (vec && d == 32) {
    SEG_LAUNCH(8);
  } else if (vec && d == 16) {
    SEG_LAUNCH(4);
  } else {
    segment_sum_scalar_kernel<<<grid, 256, 0, s>>>(src, src_bstride, ldsrc, rowptr, pos, scale,
                                                   out, out_bstride, ldout, accumulate, B,
                                                   n_out, (int)d);
  }
#undef SEG_LAUNCH
#undef SEG_FOLD
  NLAM_CHECK_LAUNCH("segment_sum");
  return 0;
}

// ------------------------------------------------------------ small helpers
__global__ void add_rows_kernel(const float* __restrict__ a, int64_t lda,
                                const float* __restrict__ b, int64_t ldb,
                                float* __restrict__ out, int64_t ldout, int64_t rows, int d) {
  const int64_t total = rows * d;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / d;
    const int c = (int)(g % d);
    out[r * ldout + c] = a[r * lda + c] + b[r * ldb + c];
  }
}
extern "C" int nlam_add_rows(const float* a, int64_t lda, const float* b, int64_t ldb,
                             float* out, int64_t ldout, int64_t rows, int64_t d, void* stream) {
  if (rows <= 0 || d <= 0) return 0;
  add_rows_kernel<<<ew_grid(rows * d), 256, 0, (hipStream_t)stream>>>(a, lda, b, ldb, out, ldout,
                                                                     rows, (int)d);
  NLAM_CHECK_LAUNCH("add_rows");
  return 0;
}

__global__ void sum_batch_kernel(const float* __restrict__ x, int64_t bstride,
                                 float* __restrict__ out, int64_t B, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int64_t b = 0; b < B; ++b) s += x[b * bstride + i];
    out[i] = s;
  }
}
__global__ void sum_batch_vec_kernel(const f32x4* __restrict__ x, int64_t bstride4,
                                     f32x4* __restrict__ out, int64_t B, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 s = x[i];
    for (int64_t b = 1; b < B; ++b) s += x[b * bstride4 + i];
    out[i] = s;
  }
}
extern "C" int nlam_sum_batch(const float* x, int64_t bstride, float* out, int64_t B, int64_t n,
                              void* stream) {
  if (n <= 0) return 0;
  if (B > 0 && n % 4 == 0 && bstride % 4 == 0 && nlam_aligned16(x) && nlam_aligned16(out)) {
    sum_batch_vec_kernel<<<ew_grid(n / 4), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const f32x4*>(x), bstride / 4, reinterpret_cast<f32x4*>(out), B, n / 4);
  } else {
    sum_batch_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(x, bstride, out, B, n);
  }
  NLAM_CHECK_LAUNCH("sum_batch");
  return 0;
}

// ------------------------------------------------------------- MFMA probe
// A[i][k] = (i + 1) + 100 * (k % 7), B[k][j] = (j - 3) * ((k % 5) + 1), both exactly
// representable; the host checks out against the integer product.  A wrong lane map
// (operand or accumulator) changes the result.
__global__ __launch_bounds__(64) void mfma_probe_kernel(float* __restrict__ out) {
  const int lane = threadIdx.x;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < 64; k0 += 2) {
    const int k = k0 + (lane >> 5);
    const int i = lane & 31, j = lane & 31;
    const float a = (float)((i + 1) + 100 * (k % 7));
    const float b = (float)((j - 3) * ((k % 5) + 1));
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    out[row * 32 + (lane & 31)] = acc[r];
  }
}
extern "C" int nlam_mfma_probe(float* out, void* stream) {
  mfma_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out);
  NLAM_CHECK_LAUNCH("mfma_probe");
  return 0;
}
