// Fused masked multi-head attention, forward and backward, head dim 64; Sk <= 256 keys (bf16) / 128 (fp32 parity path).
//   forward : O = softmax(Q K^T * scale + kmask[b,key] + bias[b,q,key]) V,  LSE saved
//   backward: dQ, dK, dV (and dbias) by recomputing P from LSE (no S x S tensor ever reaches HBM)
// restating BertSelfAttention / BertOutAttention
//   (VLN-HAMT/finetune_src/models/vilmodel_cmt.py:100-134, :326-353) and, with `bias`, the
//   graph_sprels self-attention of VLN-DUET/map_nav_src/models/vilmodel.py:384-399.
// Mask semantics are the reference's: ADDITIVE (1-m)*-10000 (finite), so a fully padded query row still
// yields a finite softmax; only the tile padding (key >= Sk) is excluded with -inf.
//
// v1 data path (both dtypes): Q/K/V/dO tiles are staged through LDS as f32 (row stride 65 dwords:
// conflict-free for both "row on lane" and "column on lane" reads) and every contraction runs on the
// exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32. The score tile is produced TRANSPOSED
// (S^T = K Q^T) in the forward pass so each lane owns one query column: the row max / row sum are
// 16*NKT in-register ops + one 32-lane wavefront shuffle, and P^T is already the B operand of the
// P V product (no LDS round trip for P). In the backward pass keys sit on the lanes (S = Q K^T), so
// P and dS feed dV^T and dK^T straight from registers and only dS crosses LDS once, for dQ.
#include <stdlib.h>

#include "common.h"


#define VLNI_NS k_bf16
#define VLNI_H16 __bf16
#define VLNI_H16_ID VLNI_BF16
#define VLNI_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#include "attention_impl.inc"
#undef VLNI_NS
#undef VLNI_H16
#undef VLNI_H16_ID
#undef VLNI_MFMA16
#define VLNI_NS k_f16
#define VLNI_H16 _Float16
#define VLNI_H16_ID VLNI_F16
#define VLNI_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#include "attention_impl.inc"
#undef VLNI_NS
#undef VLNI_H16
#undef VLNI_H16_ID
#undef VLNI_MFMA16

// C-ABI: dtype 0 (float32) / 1 (bfloat16) -> the bfloat16 instance, 2 (float16) -> the float16 instance
extern "C" int vlni_attn_fwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const float* kmask, const float* bias, void* out, long ldo, float* lse, int B, int nh, int Sq, int Sk, float scale, float drop_p, unsigned drop_seed, void* stream) {
  return dtype == VLNI_F16 ? k_f16::vlni_attn_fwd(dtype, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, lse, B, nh, Sq, Sk, scale, drop_p, drop_seed, stream) : k_bf16::vlni_attn_fwd(dtype, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, lse, B, nh, Sq, Sk, scale, drop_p, drop_seed, stream);
}
extern "C" int vlni_attn_bwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const float* kmask, const float* bias, const void* out, long ldo, const void* dout, long lddo, const float* lse, void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, float* dbias, int B, int nh, int Sq, int Sk, float scale, float drop_p, unsigned drop_seed, void* stream) {
  return dtype == VLNI_F16 ? k_f16::vlni_attn_bwd(dtype, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, dout, lddo, lse, dq, lddq, dk, lddk, dv, lddv, dbias, B, nh, Sq, Sk, scale, drop_p, drop_seed, stream) : k_bf16::vlni_attn_bwd(dtype, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, dout, lddo, lse, dq, lddq, dk, lddk, dv, lddv, dbias, B, nh, Sq, Sk, scale, drop_p, drop_seed, stream);
}
extern "C" int vlni_attn_fwd_dual(int dtype, const void* const* q, const long* ldq, const void* const* k, const long* ldk, const void* const* v, const long* ldv, const float* const* kmask, const float* const* bias, void* const* out, const long* ldo, float* const* lse, int B, int nh, const int* Sq, const int* Sk, float scale, float drop_p, const unsigned* drop_seed, void* stream) {
  return dtype == VLNI_F16 ? k_f16::vlni_attn_fwd_dual(dtype, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, lse, B, nh, Sq, Sk, scale, drop_p, drop_seed, stream) : k_bf16::vlni_attn_fwd_dual(dtype, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, lse, B, nh, Sq, Sk, scale, drop_p, drop_seed, stream);
}
extern "C" int vlni_attn_bwd_dual(int dtype, const void* const* q, const long* ldq, const void* const* k, const long* ldk, const void* const* v, const long* ldv, const float* const* kmask, const float* const* bias, const void* const* out, const long* ldo, const void* const* dout, const long* lddo, const float* const* lse, void* const* dq, const long* lddq, void* const* dk, const long* lddk, void* const* dv, const long* lddv, float* dbias0, int B, int nh, const int* Sq, const int* Sk, float scale, float drop_p, const unsigned* drop_seed, void* stream) {
  return dtype == VLNI_F16 ? k_f16::vlni_attn_bwd_dual(dtype, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, dout, lddo, lse, dq, lddq, dk, lddk, dv, lddv, dbias0, B, nh, Sq, Sk, scale, drop_p, drop_seed, stream) : k_bf16::vlni_attn_bwd_dual(dtype, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, dout, lddo, lse, dq, lddq, dk, lddk, dv, lddv, dbias0, B, nh, Sq, Sk, scale, drop_p, drop_seed, stream);
}
extern "C" int vlni_attn_probs(int dtype, const void* q, long ldq, const void* k, long ldk, const float* kmask, const float* bias, float* probs, int B, int nh, int Sq, int Sk, float scale, void* stream) {
  return dtype == VLNI_F16 ? k_f16::vlni_attn_probs(dtype, q, ldq, k, ldk, kmask, bias, probs, B, nh, Sq, Sk, scale, stream) : k_bf16::vlni_attn_probs(dtype, q, ldq, k, ldk, kmask, bias, probs, B, nh, Sq, Sk, scale, stream);
}
