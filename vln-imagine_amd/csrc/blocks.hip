// Block-level entry points of the C-ABI (round 5): one call = one transformer SUBLAYER (attention block, FFN block) of one or two streams,
// forward or backward. The function issues the sublayer's 4-7 kernel launches itself - the very launches ops.py issues one by one
// (vlni_gemm_nt_multi, vlni_attn_*, vlni_layernorm_*), in the same order on the same stream - so that an EAGER caller (an unchanged reference
// agent on the drop-in modules: no captured graph hides the host) crosses Python -> C once per sublayer and direction instead of 4-7 times.
// No allocation, no state: the caller passes every activation buffer. Kernel choices (GEMM pipeline ids) are the caller's (its autotune).
//
// What the blocks compute (reference: VLN-HAMT/finetune_src/models/vilmodel_cmt.py):
//   self-attention block  y = LN(dense(attn(x Wq, x Wk, x Wv)) + x)          BertAttention :151-161 (BertSelfAttention :100-134, BertSelfOutput :144-148)
//   FFN block             y = LN(W2 gelu(W1 x + b1) + b2 + x)               BertIntermediate :173-176 + BertOutput :186-190
// with hidden dropout between a dense output and its residual add and attention-probability dropout inside the attention kernel, both as
// counter-based masks the backward regenerates from the seed. n = 1: one stream (text encoder, history panorama encoder); n = 2: the language
// and vision streams of a cross-modal layer (:399-421), or DUET's global-map and local-viewpoint branches, as dual-problem launches.
#include <stdint.h>

#include "../../include/vlni.h"

extern "C" {

#define BLK_TRY(call)            \
  do {                           \
    int rc_ = (call);            \
    if (rc_ != VLNI_OK) return rc_; \
  } while (0)

static inline const char* at(const void* p, long off_elems, int es) { return (const char*)p + off_elems * es; }
// The attention kernels are built for 64-wide heads (H = 64 nh: every released HAMT / DUET checkpoint); another head size takes the caller's
// launch-by-launch path (VLNI_EUNSUP), it is not scored with the wrong scale.
#define BLK_HEADS(a) do { if ((a)->nh < 1 || (a)->H != 64 * (a)->nh) return VLNI_EUNSUP; } while (0)

int vlni_self_att_block_fwd(const VlniBlockArgs* a, void* stream) {
  if (!a || a->n < 1 || a->n > 2) return VLNI_EINVAL;
  BLK_HEADS(a);
  const float scale = 0.125f;                                  // 1 / sqrt(64)
  const int n = a->n, H = a->H, es = a->dtype == VLNI_F32 ? 4 : 2;
  const void* A[2]; long lda[2]; const void* W[2]; long ldw[2]; void* C[2]; long ldc[2]; int M[2]; const float* bias[2];
  for (int i = 0; i < n; ++i) {
    const VlniBlockSide& s = a->s[i];
    A[i] = s.x; lda[i] = s.ldx; W[i] = s.w_in; ldw[i] = H; C[i] = s.mid; ldc[i] = 3L * H; M[i] = s.B * s.S; bias[i] = s.b_in;
  }
  BLK_TRY(vlni_gemm_nt_multi(a->dtype, n, A, lda, W, ldw, C, ldc, M, 3 * H, H, bias, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                             a->v_in, 0.f, nullptr, stream));
  // attention: one dual launch where the kernels allow it (16-bit, aligned, same batch), else one launch per stream
  int done = 0;
  if (n == 2 && a->dtype != VLNI_F32 && a->s[0].B == a->s[1].B) {
    const void* q[2]; const void* k[2]; const void* v[2]; long ld3[2]; const float* km[2]; const float* bs[2] = {a->bias0, nullptr};
    void* o[2]; long ldo[2]; float* lse[2]; int Sq[2]; unsigned seeds[2];
    for (int i = 0; i < 2; ++i) {
      const VlniBlockSide& s = a->s[i];
      q[i] = s.mid; k[i] = at(s.mid, H, es); v[i] = at(s.mid, 2L * H, es); ld3[i] = 3L * H; km[i] = s.kmask;
      o[i] = s.aux; ldo[i] = H; lse[i] = s.lse; Sq[i] = s.S; seeds[i] = s.seed_attn;
    }
    const int rc = vlni_attn_fwd_dual(a->dtype, q, ld3, k, ld3, v, ld3, km, bs, o, ldo, lse, a->s[0].B, a->nh, Sq, Sq, scale, a->p_attn, seeds, stream);
    if (rc == VLNI_OK) done = 1;
    else if (rc != VLNI_EUNSUP) return rc;
  }
  if (!done)
    for (int i = 0; i < n; ++i) {
      const VlniBlockSide& s = a->s[i];
      BLK_TRY(vlni_attn_fwd(a->dtype, s.mid, 3L * H, at(s.mid, H, es), 3L * H, at(s.mid, 2L * H, es), 3L * H, s.kmask, i == 0 ? a->bias0 : nullptr,
                            s.aux, H, s.lse, s.B, a->nh, s.S, s.S, scale, a->p_attn, s.seed_attn, stream));
    }
  const void* res[2]; long ldr[2]; unsigned dseed[2];
  for (int i = 0; i < n; ++i) {
    const VlniBlockSide& s = a->s[i];
    A[i] = s.aux; lda[i] = H; W[i] = s.w_out; ldw[i] = H; C[i] = s.pre; ldc[i] = H; bias[i] = s.b_out; res[i] = s.x; ldr[i] = s.ldx;
    dseed[i] = s.seed_dense;
  }
  BLK_TRY(vlni_gemm_nt_multi(a->dtype, n, A, lda, W, ldw, C, ldc, M, H, H, bias, 0, res, ldr, nullptr, nullptr, nullptr, nullptr, 0, a->v_out,
                             a->p_hidden, dseed, stream));
  if (n == 2) {
    const void* x2[2] = {a->s[0].pre, a->s[1].pre}; long ld2[2] = {H, H}; const float* g2[2] = {a->s[0].gamma, a->s[1].gamma};
    const float* b2[2] = {a->s[0].beta, a->s[1].beta}; void* y2[2] = {a->s[0].y, a->s[1].y}; float* m2[2] = {a->s[0].mean, a->s[1].mean};
    float* r2[2] = {a->s[0].rstd, a->s[1].rstd};
    return vlni_layernorm_fwd_dual(a->dtype, x2, ld2, g2, b2, a->eps, y2, ld2, m2, r2, M, H, stream);
  }
  const VlniBlockSide& s = a->s[0];
  return vlni_layernorm_fwd(a->dtype, s.pre, H, s.gamma, s.beta, a->eps, s.y, H, s.mean, s.rstd, M[0], H, stream);
}

// LayerNorm backward of one or two streams: dpre (+ the dropped copy when p_hidden > 0), dgamma / dbeta accumulated
static int blk_ln_bwd(const VlniBlockArgs* a, const int* M, void* stream) {
  const int n = a->n, H = a->H;
  const bool drop = a->p_hidden > 0.f;
  if (n == 2) {
    const void* dy[2]; long lddy[2]; const void* x[2]; long ld[2]; const float* g[2]; const float* mu[2]; const float* rs[2]; void* dx[2];
    float* dg[2]; float* db[2]; void* dxd[2]; unsigned seeds[2];
    for (int i = 0; i < 2; ++i) {
      const VlniBlockSide& s = a->s[i];
      dy[i] = s.dy; lddy[i] = s.lddy; x[i] = s.pre; ld[i] = H; g[i] = s.gamma; mu[i] = s.mean; rs[i] = s.rstd; dx[i] = s.dpre; dg[i] = s.dgamma;
      db[i] = s.dbeta; dxd[i] = s.dmid_drop; seeds[i] = s.seed_dense;
    }
    return vlni_layernorm_bwd_dual(a->dtype, dy, lddy, x, ld, g, mu, rs, dx, ld, dg, db, M, H, nullptr, nullptr, drop ? dxd : nullptr, drop ? ld : nullptr,
                                   drop ? a->p_hidden : 0.f, drop ? seeds : nullptr, stream);
  }
  const VlniBlockSide& s = a->s[0];
  return vlni_layernorm_bwd(a->dtype, s.dy, s.lddy, s.pre, H, s.gamma, s.mean, s.rstd, s.dpre, H, s.dgamma, s.dbeta, M[0], H, nullptr, 0,
                            drop ? s.dmid_drop : nullptr, drop ? H : 0, drop ? a->p_hidden : 0.f, drop ? s.seed_dense : 0u, stream);
}

// activation gradients of the self-attention block; the weight / bias gradients (dWo = dmid_drop^T ctx, dWqkv = dqkv^T x) are the caller's
int vlni_self_att_block_bwd(const VlniBlockArgs* a, void* stream) {
  if (!a || a->n < 1 || a->n > 2) return VLNI_EINVAL;
  BLK_HEADS(a);
  const float scale = 0.125f;                                  // 1 / sqrt(64)
  const int n = a->n, H = a->H, es = a->dtype == VLNI_F32 ? 4 : 2;
  int M[2];
  for (int i = 0; i < n; ++i) M[i] = a->s[i].B * a->s[i].S;
  BLK_TRY(blk_ln_bwd(a, M, stream));
  const void* A[2]; long lda[2]; const void* W[2]; long ldw[2]; void* C[2]; long ldc[2];
  for (int i = 0; i < n; ++i) {
    const VlniBlockSide& s = a->s[i];
    A[i] = s.dmid_drop; lda[i] = H; W[i] = s.wt_out; ldw[i] = H; C[i] = s.daux; ldc[i] = H;
  }
  BLK_TRY(vlni_gemm_nt_multi(a->dtype, n, A, lda, W, ldw, C, ldc, M, H, H, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, a->v_out,
                             0.f, nullptr, stream));
  int done = 0;
  if (n == 2 && a->dtype != VLNI_F32 && a->s[0].B == a->s[1].B) {
    const void* q[2]; const void* k[2]; const void* v[2]; long ld3[2]; const float* km[2]; const float* bs[2] = {a->bias0, nullptr};
    const void* o[2]; const void* dO[2]; long ldo[2]; const float* lse[2]; void* dq[2]; void* dk[2]; void* dv[2]; int Sq[2]; unsigned seeds[2];
    for (int i = 0; i < 2; ++i) {
      const VlniBlockSide& s = a->s[i];
      q[i] = s.mid; k[i] = at(s.mid, H, es); v[i] = at(s.mid, 2L * H, es); ld3[i] = 3L * H; km[i] = s.kmask; o[i] = s.aux; dO[i] = s.daux; ldo[i] = H;
      lse[i] = s.lse; dq[i] = s.dmid; dk[i] = (void*)at(s.dmid, H, es); dv[i] = (void*)at(s.dmid, 2L * H, es); Sq[i] = s.S; seeds[i] = s.seed_attn;
    }
    const int rc = vlni_attn_bwd_dual(a->dtype, q, ld3, k, ld3, v, ld3, km, bs, o, ldo, dO, ldo, lse, dq, ld3, dk, ld3, dv, ld3, a->dbias0, a->s[0].B,
                                      a->nh, Sq, Sq, scale, a->p_attn, seeds, stream);
    if (rc == VLNI_OK) done = 1;
    else if (rc != VLNI_EUNSUP) return rc;
  }
  if (!done)
    for (int i = 0; i < n; ++i) {
      const VlniBlockSide& s = a->s[i];
      BLK_TRY(vlni_attn_bwd(a->dtype, s.mid, 3L * H, at(s.mid, H, es), 3L * H, at(s.mid, 2L * H, es), 3L * H, s.kmask, i == 0 ? a->bias0 : nullptr, s.aux, H,
                            s.daux, H, s.lse, s.dmid, 3L * H, (void*)at(s.dmid, H, es), 3L * H, (void*)at(s.dmid, 2L * H, es), 3L * H,
                            i == 0 ? a->dbias0 : nullptr, s.B, a->nh, s.S, s.S, scale, a->p_attn, s.seed_attn, stream));
    }
  // dx = dqkv Wqkv (+ dpre: the residual path); streams whose input needs no gradient are left out
  const void* res[2]; long ldr[2]; int Md[2], m = 0;
  for (int i = 0; i < n; ++i) {
    const VlniBlockSide& s = a->s[i];
    if (!s.dx) continue;
    A[m] = s.dmid; lda[m] = 3L * H; W[m] = s.wt_in; ldw[m] = 3L * H; C[m] = s.dx; ldc[m] = H; res[m] = s.dpre; ldr[m] = H; Md[m] = M[i];
    ++m;
  }
  if (m == 0) return VLNI_OK;
  return vlni_gemm_nt_multi(a->dtype, m, A, lda, W, ldw, C, ldc, Md, H, 3 * H, nullptr, 0, res, ldr, nullptr, nullptr, nullptr, nullptr, 0, a->v_in, 0.f,
                            nullptr, stream);
}

int vlni_ffn_block_fwd(const VlniBlockArgs* a, void* stream) {
  if (!a || a->n < 1 || a->n > 2) return VLNI_EINVAL;
  const int n = a->n, H = a->H, FF = a->FF;
  const void* A[2]; long lda[2]; const void* W[2]; long ldw[2]; void* C[2]; long ldc[2]; int M[2]; const float* bias[2]; void* pre[2]; long ldp[2];
  for (int i = 0; i < n; ++i) {
    const VlniBlockSide& s = a->s[i];
    A[i] = s.x; lda[i] = s.ldx; W[i] = s.w_in; ldw[i] = H; C[i] = s.mid; ldc[i] = FF; M[i] = s.B * s.S; bias[i] = s.b_in; pre[i] = s.aux; ldp[i] = FF;
  }
  BLK_TRY(vlni_gemm_nt_multi(a->dtype, n, A, lda, W, ldw, C, ldc, M, FF, H, bias, a->act, nullptr, nullptr, pre, ldp, nullptr, nullptr, 0, a->v_in, 0.f,
                             nullptr, stream));
  const void* res[2]; long ldr[2]; unsigned dseed[2];
  for (int i = 0; i < n; ++i) {
    const VlniBlockSide& s = a->s[i];
    A[i] = s.mid; lda[i] = FF; W[i] = s.w_out; ldw[i] = FF; C[i] = s.pre; ldc[i] = H; bias[i] = s.b_out; res[i] = s.x; ldr[i] = s.ldx;
    dseed[i] = s.seed_dense;
  }
  BLK_TRY(vlni_gemm_nt_multi(a->dtype, n, A, lda, W, ldw, C, ldc, M, H, FF, bias, 0, res, ldr, nullptr, nullptr, nullptr, nullptr, 0, a->v_out,
                             a->p_hidden, dseed, stream));
  if (n == 2) {
    const void* x2[2] = {a->s[0].pre, a->s[1].pre}; long ld2[2] = {H, H}; const float* g2[2] = {a->s[0].gamma, a->s[1].gamma};
    const float* b2[2] = {a->s[0].beta, a->s[1].beta}; void* y2[2] = {a->s[0].y, a->s[1].y}; float* m2[2] = {a->s[0].mean, a->s[1].mean};
    float* r2[2] = {a->s[0].rstd, a->s[1].rstd};
    return vlni_layernorm_fwd_dual(a->dtype, x2, ld2, g2, b2, a->eps, y2, ld2, m2, r2, M, H, stream);
  }
  const VlniBlockSide& s = a->s[0];
  return vlni_layernorm_fwd(a->dtype, s.pre, H, s.gamma, s.beta, a->eps, s.y, H, s.mean, s.rstd, M[0], H, stream);
}

// activation gradients of the FFN block; dW2 = dmid_drop^T gelu(z), dW1 = dz^T x are the caller's
int vlni_ffn_block_bwd(const VlniBlockArgs* a, void* stream) {
  if (!a || a->n < 1 || a->n > 2) return VLNI_EINVAL;
  const int n = a->n, H = a->H, FF = a->FF;
  int M[2];
  for (int i = 0; i < n; ++i) M[i] = a->s[i].B * a->s[i].S;
  BLK_TRY(blk_ln_bwd(a, M, stream));
  const void* A[2]; long lda[2]; const void* W[2]; long ldw[2]; void* C[2]; long ldc[2]; const void* ds[2]; long ldd[2];
  for (int i = 0; i < n; ++i) {
    const VlniBlockSide& s = a->s[i];
    A[i] = s.dmid_drop; lda[i] = H; W[i] = s.wt_out; ldw[i] = H; C[i] = s.daux; ldc[i] = FF; ds[i] = s.aux; ldd[i] = FF;
  }
  // dz = (dmid W2) * GELU'(z): the derivative fused in the dgrad epilogue (dact 1: from z, 3: the stored derivative)
  BLK_TRY(vlni_gemm_nt_multi(a->dtype, n, A, lda, W, ldw, C, ldc, M, FF, H, nullptr, 0, nullptr, nullptr, nullptr, nullptr, ds, ldd, a->dact, a->v_out,
                             0.f, nullptr, stream));
  const void* res[2]; long ldr[2]; int Md[2], m = 0;
  for (int i = 0; i < n; ++i) {
    const VlniBlockSide& s = a->s[i];
    if (!s.dx) continue;
    A[m] = s.daux; lda[m] = FF; W[m] = s.wt_in; ldw[m] = FF; C[m] = s.dx; ldc[m] = H; res[m] = s.dpre; ldr[m] = H; Md[m] = M[i];
    ++m;
  }
  if (m == 0) return VLNI_OK;
  return vlni_gemm_nt_multi(a->dtype, m, A, lda, W, ldw, C, ldc, Md, H, FF, nullptr, 0, res, ldr, nullptr, nullptr, nullptr, nullptr, 0, a->v_in, 0.f,
                            nullptr, stream);
}

}  // extern "C"
