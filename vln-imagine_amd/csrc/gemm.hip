// MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] * B[N,K]^T )      ("NT": both operands K-contiguous)
//
// One kernel template serves every dense contraction of the HAMT/DUET hot path:
//   forward   Y  = X  * W^T          (A = X [rows,in],  B = W  [out,in])
//   dgrad     dX = dY * (W^T)^T      (A = dY [rows,out], B = W^T [in,out], a transposed shadow copy)
//   wgrad     dW = dY^T * X          (A = dY^T [out,rows], B = X^T [in,rows]; split over rows, f32 atomics)
// replacing the nn.Linear / torch.matmul calls of
//   VLN-HAMT/finetune_src/models/vilmodel_cmt.py:101-103,145,174,187,327-329 and their autograd.
//
// Base geometry (64-wide waves): 128x128 output tile, 4 waves in 2x2, each wave a 64x64 sub-tile = 2x2 MFMA 32x32
// accumulators (64 acc VGPRs). A k-tile is 128 BYTES of K per row for both dtypes (bf16: BK=64 ->
// v_mfma_f32_32x32x16_bf16; f32: BK=32 -> v_mfma_f32_32x32x2_f32, the exact-fp32 parity path), staged in 16-B chunks with an
// XOR swizzle (chunk ^ ((row>>1)&7)) so the ds_read_b128 fragment reads of 16 different rows hit 16 different 16-B slots of
// the 256-B bank row. Tiles are dealt to XCDs in contiguous chunks of a grouped order (bijective remap, tile_origin) so the
// tiles an XCD runs together share operand panels in its L2.
//
// Kernels in this file (variant numbers of vlni_gemm_nt_v; all bit-identical):
//   gemm_nt_kernel        (1)      register-staged, one 32-KiB LDS stage, 4 blocks per CU
//   gemm_nt_glds_kernel   (2-5)    LDS-DMA (global_load_lds_dwordx4) into 2 or 3 stages, counted vmcnt + raw s_barrier, 4 or 8 waves
//   gemm_nt_big_kernel    (6-13)   the same pipeline for any WM x WN wave grid and MI x NJ accumulators per wave:
//                                  256x128 / 256x256 / 128x256 (one block per CU), 64x128 / 128x64 / 64x64 (under-filled launches),
//                                  192x128 / 128x192 (two 40-KiB stages: two blocks per CU with 17 % fewer operand bytes per flop)
//   gemm_nn_glds_kernel   (+16)    B given as [K,N]: transposing LDS reads on the weight itself (bf16 dgrad without a W^T copy)
//   gemm_tn_*_kernel               weight gradients dY^T X straight from row-major dY and X (ds_read_b64_tr_b16), grouped segments
// What bounds them on the K = 768 shapes of this workload is the CU's vector-memory pipe (~66 GB/s per CU), see DESIGN.md sections 6-7.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#include <algorithm>

// ---- the kernels, once per 16-bit compute type -------------------------------------------------
#define VLNI_NS k_bf16
#define VLNI_H16 __bf16
#define VLNI_H16_ID VLNI_BF16
#define VLNI_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define VLNI_MFMA16S(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#include "gemm_impl.inc"
#undef VLNI_NS
#undef VLNI_H16
#undef VLNI_H16_ID
#undef VLNI_MFMA16
#undef VLNI_MFMA16S
#define VLNI_NS k_f16
#define VLNI_H16 _Float16
#define VLNI_H16_ID VLNI_F16
#define VLNI_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define VLNI_MFMA16S(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#include "gemm_impl.inc"
#undef VLNI_NS
#undef VLNI_H16
#undef VLNI_H16_ID
#undef VLNI_MFMA16
#undef VLNI_MFMA16S

// ---- C-ABI: dtype 0 (float32) and 1 (bfloat16) go to the bfloat16 instance, 2 (float16) to the float16 instance -----------------
extern "C" int vlni_gemm_nt_v(int dtype, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                              const float* bias, int act, const void* residual, long ldr, void* preact, long ldp, const void* dact_src,
                              long ldd, int dact, float alpha, int split_k, int atomic_f32, int variant, float drop_p, unsigned drop_seed,
                              void* stream) {
  return dtype == VLNI_F16 ? k_f16::vlni_gemm_nt_v(dtype, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, preact, ldp, dact_src, ldd, dact,
                                                   alpha, split_k, atomic_f32, variant, drop_p, drop_seed, stream)
                           : k_bf16::vlni_gemm_nt_v(dtype, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, preact, ldp, dact_src, ldd, dact,
                                                    alpha, split_k, atomic_f32, variant, drop_p, drop_seed, stream);
}
extern "C" int vlni_gemm_nt_dual(int dtype, const void* const* A, const long* lda, const void* const* B, const long* ldb, void* const* C,
                                 const long* ldc, const int* M, int N, int K, const float* const* bias, int act, const void* const* residual,
                                 const long* ldr, void* const* preact, const long* ldp, const void* const* dact_src, const long* ldd, int dact,
                                 int variant, float drop_p, const unsigned* drop_seed, void* stream) {
  return dtype == VLNI_F16 ? k_f16::vlni_gemm_nt_dual(dtype, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, preact, ldp, dact_src, ldd, dact,
                                                      variant, drop_p, drop_seed, stream)
                           : k_bf16::vlni_gemm_nt_dual(dtype, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, preact, ldp, dact_src, ldd, dact,
                                                       variant, drop_p, drop_seed, stream);
}
extern "C" int vlni_gemm_nt_multi(int dtype, int nprob, const void* const* A, const long* lda, const void* const* B, const long* ldb, void* const* C,
                                  const long* ldc, const int* M, int N, int K, const float* const* bias, int act, const void* const* residual,
                                  const long* ldr, void* const* preact, const long* ldp, const void* const* dact_src, const long* ldd, int dact,
                                  int variant, float drop_p, const unsigned* drop_seed, void* stream) {
  return dtype == VLNI_F16 ? k_f16::vlni_gemm_nt_multi(dtype, nprob, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, preact, ldp, dact_src, ldd,
                                                       dact, variant, drop_p, drop_seed, stream)
                           : k_bf16::vlni_gemm_nt_multi(dtype, nprob, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, preact, ldp, dact_src, ldd,
                                                        dact, variant, drop_p, drop_seed, stream);
}
extern "C" int vlni_gemm_nt(int dtype, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                            const float* bias, int act, const void* residual, long ldr, void* preact, long ldp, const void* dact_src,
                            long ldd, int dact, float alpha, int split_k, int atomic_f32, void* stream) {
  return vlni_gemm_nt_v(dtype, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, preact, ldp, dact_src, ldd, dact, alpha, split_k,
                        atomic_f32, 0, 0.f, 0u, stream);
}
#ifdef VLNI_DIAG
extern "C" int vlni_debug_pk_stamps(void* host_dst, int bytes) { return k_bf16::vlni_debug_pk_stamps(host_dst, bytes); }
#endif
extern "C" int vlni_reduce_parts(const void* table, int n_entries, int n_blocks, void* stream) {
  return k_bf16::vlni_reduce_parts(table, n_entries, n_blocks, stream);
}
extern "C" int vlni_reduce_parts_sq(const void* table, int n_entries, int n_blocks, float* sumsq, int nslots, void* stream) {
  return k_bf16::vlni_reduce_parts_sq(table, n_entries, n_blocks, sumsq, nslots, stream);
}
// weight gradients: the `_bf16` entry points keep their names; `_h16` take the 16-bit dtype (1 bfloat16, 2 float16) in front
extern "C" int vlni_gemm_tn_bf16_grouped_v(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb, float* C,
                                           long ldc, int N, int K, float* colsum, int split, int variant, void* stream) {
  return k_bf16::vlni_gemm_tn_bf16_grouped_v(nseg, A, B, M, lda, ldb, C, ldc, N, K, colsum, split, variant, stream);
}
extern "C" int vlni_gemm_tn_bf16_grouped_part(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb,
                                              float* part, long part_stride, int N, int K, float* colsum_part, int split, int variant,
                                              void* stream) {
  return k_bf16::vlni_gemm_tn_bf16_grouped_part(nseg, A, B, M, lda, ldb, part, part_stride, N, K, colsum_part, split, variant, stream);
}
extern "C" int vlni_gemm_tn_bf16_grouped(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb, float* C,
                                         long ldc, int N, int K, float* colsum, int split, void* stream) {
  return k_bf16::vlni_gemm_tn_bf16_grouped(nseg, A, B, M, lda, ldb, C, ldc, N, K, colsum, split, stream);
}
extern "C" int vlni_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K, float* colsum,
                                 int split, void* stream) {
  return k_bf16::vlni_gemm_tn_bf16(A, lda, B, ldb, C, ldc, M, N, K, colsum, split, stream);
}
extern "C" int vlni_gemm_tn_h16_grouped_v(int dtype, int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb,
                                          float* C, long ldc, int N, int K, float* colsum, int split, int variant, void* stream) {
  VLNI_CHECK(dtype == VLNI_BF16 || dtype == VLNI_F16, VLNI_EINVAL, "gemm_tn_h16: dtype %d (1 bfloat16, 2 float16)", dtype);
  return dtype == VLNI_F16 ? k_f16::vlni_gemm_tn_bf16_grouped_v(nseg, A, B, M, lda, ldb, C, ldc, N, K, colsum, split, variant, stream)
                           : k_bf16::vlni_gemm_tn_bf16_grouped_v(nseg, A, B, M, lda, ldb, C, ldc, N, K, colsum, split, variant, stream);
}
extern "C" int vlni_gemm_tn_h16_grouped_part(int dtype, int nseg, const void* const* A, const void* const* B, const int* M, long lda,
                                             long ldb, float* part, long part_stride, int N, int K, float* colsum_part, int split,
                                             int variant, void* stream) {
  VLNI_CHECK(dtype == VLNI_BF16 || dtype == VLNI_F16, VLNI_EINVAL, "gemm_tn_h16: dtype %d (1 bfloat16, 2 float16)", dtype);
  return dtype == VLNI_F16 ? k_f16::vlni_gemm_tn_bf16_grouped_part(nseg, A, B, M, lda, ldb, part, part_stride, N, K, colsum_part, split, variant, stream)
                           : k_bf16::vlni_gemm_tn_bf16_grouped_part(nseg, A, B, M, lda, ldb, part, part_stride, N, K, colsum_part, split, variant, stream);
}
