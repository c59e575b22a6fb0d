// Hidden 256 (NLAM_MFMA=bf16) forms of the wide-path entry points: feature-split kernels with
// register-stationary weights (fused_fs.hip).  Internal C++ linkage inside libnlam_hip.so --
// the C ABI (include/nlam_hip.h) reaches them through nlam_lin_fwd / nlam_tail_fwd /
// nlam_tail_bwd / nlam_lin_bwd_data / nlam_wide_outer when the width is 256.
#pragma once
#include <cstdint>

int nlam_fs_lin_fwd_256(const float* x, int64_t x_bstride, int64_t x_ld, int k_in, const float* W,
                        int64_t ldW, const float* bias, int n_out, float* out, int64_t out_bstride,
                        int64_t out_ld, int64_t B, int64_t rows, int out_bf16, void* stream);
int nlam_fs_lin_bwd_data_256(const float* gy, int64_t gy_bstride, int64_t gy_ld, const float* W,
                             int64_t ldW, float* gx, int64_t gx_bstride, int64_t gx_ld,
                             const float* gx_add, int64_t ga_bstride, int64_t ga_ld, int64_t B,
                             int64_t rows, void* stream);
int nlam_fs_tail_fwd_256(
    const int32_t* tiles, int64_t ntiles, int64_t rows, const int32_t* csr_rec,
    const int32_t* csr_rowptr,
    const float* a, int64_t a_bstride, int64_t a_ld, const int32_t* idx_a,
    const float* b, int64_t b_bstride, int64_t b_ld, const int32_t* idx_b,
    const float* c, int64_t c_bstride, int64_t c_ld, const int32_t* idx_c,
    const float* W2, int64_t ldW2, const float* b2, const float* gamma, const float* beta,
    int n_out, float* h_out, int64_t h_bstride, void* z_keep, int64_t z_bstride,
    float* y, int64_t y_bstride, int64_t y_ld, const int32_t* idx_y,
    const float* res, int64_t res_bstride, int64_t res_ld,
    float* agg, int64_t agg_bstride, int64_t agg_ld, const float* inv_deg,
    int64_t B, int io_bf16, void* stream);
int nlam_fs_tail_bwd_256(
    const int32_t* tiles, int64_t ntiles, int64_t rows, const int32_t* csr_rec,
    const int32_t* csr_rowptr, const float* h, int64_t h_bstride,
    const void* z_keep, int64_t z_bstride,
    const float* g1, int64_t g1_bstride, int64_t g1_ld, const int32_t* idx_g1, const float* scale1,
    const float* g2, int64_t g2_bstride, int64_t g2_ld, const int32_t* idx_g2,
    const float* W2, int64_t ldW2, const float* b2, const float* gamma, int n_out,
    float* gz_out, int64_t gz_bstride,
    float* gh, int64_t gh_bstride, int64_t gh_ld, const int32_t* idx_gh,
    float* gpr, int64_t gpr_bstride, int64_t gpr_ld,
    float* slab, int64_t slab_stride, int64_t B, unsigned grid, int io_bf16, void* stream);
int nlam_fs_outer_256(const float* g, int64_t g_bstride, int64_t g_ld, int ng,
                      const float* x, int64_t x_bstride, int64_t x_ld, int nx, int silu_x,
                      float* slab, int64_t slab_stride, int64_t B, int64_t rows, unsigned grid,
                      int io_bf16, void* stream);
int nlam_fs_lin_fwd_multi_256(int n, const float* const* x, const int64_t* x_bstride,
                              const int64_t* x_ld, const float* const* W, const int64_t* ldW,
                              const float* const* bias, float* const* out,
                              const int64_t* out_bstride, const int64_t* out_ld, const int64_t* B,
                              const int64_t* rows, int out_bf16_mask, void* stream);
int nlam_fs_lin_bwd_data_multi_256(int n, const float* const* gy, const int64_t* gy_bstride,
                                   const int64_t* gy_ld, const float* const* W, const int64_t* ldW,
                                   float* const* gx, const int64_t* gx_bstride, const int64_t* gx_ld,
                                   const float* const* gx_add, const int64_t* ga_bstride,
                                   const int64_t* ga_ld, const int64_t* B, const int64_t* rows,
                                   void* stream);
int nlam_fs_outer_multi_256(int n, const float* const* g, const int64_t* g_bstride,
                            const int64_t* g_ld, const float* const* x, const int64_t* x_bstride,
                            const int64_t* x_ld, const int32_t* silu_x, float* const* slab,
                            const int64_t* slab_stride, const int64_t* B, const int64_t* rows,
                            const unsigned* grid, const int32_t* io_bf16, void* stream, const int32_t* nx);
