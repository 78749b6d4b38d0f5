/* vlni.h -- C-ABI of the MI355X (gfx950) operator library under the HAMT/DUET drop-in modules.
 *
 * The reference (akhilperincherry/VLN-Imagine) is pure PyTorch: it has no FFI for this path; the
 * boundary the drop-in keeps is the Python module API (models.vilmodel_cmt.NavCMT,
 * models.vilmodel.GlocalTextPathNavCMT, ...). This header is the operator layer those modules call:
 * plain device pointers + explicit shapes/strides + a hipStream_t (as void*), no torch types, no
 * allocation inside, no global state except the thread-local error string and the optional dropout seed base.
 * Every function returns 0 (VLNI_OK) or a negative code; the message is vlni_last_error().
 * Row strides (`ld*`) are in ELEMENTS of the tensor's dtype. dtype: 0 = float32, 1 = bfloat16, 2 = float16 (every entry point
 * that takes a dtype; float16 runs the bfloat16 kernels with the f16 MFMA opcode: csrc/*_impl.inc are compiled once per type).
 * All reductions accumulate in float32. "R:" = VLN-HAMT/finetune_src/models/vilmodel_cmt.py,
 * "D:" = VLN-DUET/map_nav_src/models/vilmodel.py, "T:" = VLN-DUET/map_nav_src/models/transformer.py.
 */
#ifndef VLNI_H
#define VLNI_H
#ifdef __cplusplus
extern "C" {
#endif

#define VLNI_F32 0
#define VLNI_BF16 1
#define VLNI_F16 2
#define VLNI_OK 0
#define VLNI_EINVAL (-1)
#define VLNI_ELAUNCH (-2)
#define VLNI_EUNSUP (-3)

const char* vlni_last_error(void);
int vlni_version(void);
/* Optional: a device-resident unsigned added to every dropout seed below (0 / never called = seeds used as given). Lets a
   captured hipGraph draw new masks on every replay: a node of the graph advances *device_ptr, forward and backward nodes of the
   same replay read the same value. The one piece of process-wide state besides the error string. */
int vlni_set_dropout_seed_base(const unsigned* device_ptr);
/* Optional: a device-resident int that counts rows whose table index was out of range. The entry points that take caller-supplied row
   indices (vlni_embed_combine_fwd, vlni_scatter_add_rows, vlni_scatter_add_rows_small) never touch memory outside the table: such a row
   adds nothing, and - once a counter is registered - is counted, so the host can raise the way nn.Embedding (R:535-544, R:596-618) would,
   when it chooses to look (no per-call device-to-host sync). NULL / never called = skipped silently. */
int vlni_set_index_error_counter(int* device_ptr);
/* Host -> device copy of a small launch table on `stream` (hipMemcpyAsync). From pinned host memory it may be recorded into a hipGraph
   capture (a memcpy node; the host bytes must stay alive and unchanged for the graph's life). */
int vlni_upload(void* dst, const void* src_host, long bytes, void* stream);

/* C[M,N] = epi(alpha * A[M,K] * B[N,K]^T): replaces nn.Linear forward (R:101-103,145,174,187,327-329,
 * 536-537, 956-960; D:598-655 MLP), its dgrad (B = transposed weight shadow) and, with
 * atomic_f32 + split_k, its wgrad (A = dY^T, B = X^T; C float32, accumulated with atomics).
 * epi: + bias[N] -> (store preact) -> (* act'(dact_src): 1 gelu', 2 relu'; 3: * dact_src itself) -> act (1 gelu-erf R:27-33,
 * 2 relu, 3 gelu-erf with GELU'(pre) stored to `preact` instead of pre: the FFN forward of the 16-bit paths, whose dgrad then
 * runs dact = 3, one multiply instead of three transcendentals per element) -> + residual -> C.
 * K, lda, ldb multiples of 16 bytes; A, B 16-byte aligned. */
int vlni_gemm_nt(int dtype, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                 const float* bias, int act, const void* residual, long ldr, void* preact, long ldp,
                 const void* dact_src, long ldd, int dact, float alpha, int split_k, int atomic_f32, void* stream);

/* Same with an explicit kernel variant: 0 auto, 1 register-staged 32 KiB, 2 / 3 LDS-DMA 2- / 3-stage, 4 / 5 the same with 8 waves,
   6 / 7 / 8 LDS-DMA large tiles 256x128 / 256x256 / 128x256, 9 / 10 / 11 small tiles 64x128 / 128x64 / 64x64, 12 / 13 tiles 192x128 / 128x192
   (two blocks per CU), 14 persistent 128x128 (register epilogue): results identical bit for bit. 15 = persistent 256x256 "8-phase"
   kernel on v_mfma_f32_16x16x32 for long launches (16-bit types, K % 128 == 0; other cases run variant 14): same products summed in
   another tree, i.e. equal to the others within float32 rounding of the accumulator. 32 (= id 16; ids above 15 are passed as 32 + id - 16
   because "+ 16" is the layout flag below) = 256x128 tiles, three LDS buffers, four dedicated loader waves, same MFMA as 15: step-long
   launches and N = 768 projections (16-bit types, K % 64 == 0, K >= 128).
   variant + 16: B is given as [K,N] row-major with ldb >= N, i.e. the forward weight W[out,in] itself as the dgrad operand
   (dX = dY * W) - no transposed weight copy; bf16 only, K % 64 == 0, K >= 192, N % 8 == 0, pipelines 2..5. */
int vlni_gemm_nt_v(int dtype, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                   const float* bias, int act, const void* residual, long ldr, void* preact, long ldp,
                   const void* dact_src, long ldd, int dact, float alpha, int split_k, int atomic_f32, int variant,
                   float drop_p, unsigned drop_seed /* dropout after act, before residual; mask = f(seed, row*N+col) */,
                   void* stream);

#ifdef VLNI_DIAG
/* Diagnostic builds only (VLNI_DIAG=1 python -m vln_imagine_amd.build; not in the default library): per-block cycle sums (8 x uint64 x
   <= 768 blocks: DMA wait, barrier, DMA issue, reads + MFMA, epilogue, k-steps, kernel, cold start) of the last launch of the persistent
   GEMM's stamped build (environment VLNI_PK_HACK=4), tools/gemm_stamps.py */
int vlni_debug_pk_stamps(void* host_dst, int bytes);
#endif

/* Two problems (same N, K, epilogue kind; arrays of 2) in ONE launch: the language / vision streams of a cross-modal layer. */
int vlni_gemm_nt_dual(int dtype, const void* const* A, const long* lda, const void* const* B, const long* ldb, void* const* C,
                      const long* ldc, const int* M, int N, int K, const float* const* bias, int act,
                      const void* const* residual, const long* ldr, void* const* preact, const long* ldp,
                      const void* const* dact_src, const long* ldd, int dact, int variant, float drop_p,
                      const unsigned* drop_seed, void* stream);

/* nprob (1..4) problems of the same N, K and epilogue kind in ONE launch (arrays of nprob); vlni_gemm_nt_dual is nprob = 2. Round 5: the
 * history panorama encoder's BertLayer (R:216-239 under HistoryEmbeddings R:603-614) rides as a third problem beside the language / vision
 * streams of the cross-modal layer with the same (N, K, epilogue) (R:399-421). */
int vlni_gemm_nt_multi(int dtype, int nprob, const void* const* A, const long* lda, const void* const* B, const long* ldb, void* const* C,
                       const long* ldc, const int* M, int N, int K, const float* const* bias, int act,
                       const void* const* residual, const long* ldr, void* const* preact, const long* ldp,
                       const void* const* dact_src, const long* ldd, int dact, int variant, float drop_p,
                       const unsigned* drop_seed, void* stream);

/* Weight gradient without transposes: C[N,K] += A[M,N]^T B[M,K] (bf16 in, float32 atomics out, split over M rows);
 * colsum[N] (optional) += column sums of A = bias gradient. Autograd of nn.Linear (R:101-103,...). */
int vlni_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                      float* colsum, int split, void* stream);

/* Grouped form: up to 16 row segments (A_s [M_s,N], B_s [M_s,K], common lda/ldb) reduced into the same C: the deferred
 * weight gradient of one parameter over all T steps of an episode in ONE launch. */
int vlni_gemm_tn_bf16_grouped(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb, float* C,
                              long ldc, int N, int K, float* colsum, int split, void* stream);

int vlni_gemm_tn_bf16_grouped_v(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb, float* C,
                                long ldc, int N, int K, float* colsum, int split, int variant /* 0/1 register-staged, 2..5 LDS-DMA 128 x 128, 6 = 256 x 256 two-stage, 7 / 8 = 256 x 256 ring (8: wave rows one barrier apart) */,
                                void* stream);
/* The same grouped launch in "partials" mode: row split z stores its share with PLAIN stores to part + z * part_stride (dense [N][K]
   float32) and its column sums to colsum_part + z * N (may be NULL); nothing is accumulated. Float atomics run at ~1.3 TB/s on MI355X
   against ~5 TB/s for stores and were 30-50 % of a weight-gradient launch. Splits actually written: ceil(T / ceil(T / split)) with
   T = sum ceil(M_s / 64). Needs an LDS-DMA variant (2..8) and at least 3 row tiles per split (else -3).
   vlni_reduce_parts then adds the partials into the gradients, MANY tensors per launch: table = device array of
   { float* dst; const float* part; long n4; long stride4; int split; int blk0; } (n4 = float4 per tensor, stride4 = float4 between
   splits, blk0 = first block of the entry, a block covers 1024 float4; entries sorted by blk0), n_blocks = total blocks. */
int vlni_gemm_tn_bf16_grouped_part(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb, float* part,
                                   long part_stride, int N, int K, float* colsum_part, int split, int variant, void* stream);
int vlni_reduce_parts(const void* table, int n_entries, int n_blocks, void* stream);
/* The same launch with per-entry modes in the upper bits of `split` and an optional sum-of-squares accumulator (float[1], caller zeroes it):
   a training step then needs no zero fill of the gradient arena under the reduced tensors, no read of dst in the reduction and no separate
   pass for the gradient norm of clip_grad_norm_ (r2r/agent_cmt.py:829). */
#define VLNI_PART_STORE   (1 << 16)   /* dst = sum of the partials (dst is not read); with split 0: dst = 0 */
#define VLNI_PART_NOWRITE (1 << 17)   /* dst is left as it is (with SUMSQ and split 0: only its sum of squares is taken) */
#define VLNI_PART_SUMSQ   (1 << 18)   /* the sum of the squares of the entry's final values is added to the accumulator */
/* sumsq: `nslots` accumulators 32 floats (128 bytes) apart - block b adds to slot b % nslots, so that the tens of thousands of block sums
   of one launch do not queue on one address; vlni_sumsq_fold adds the slots into the single float the optimizer kernels read. */
int vlni_reduce_parts_sq(const void* table, int n_entries, int n_blocks, float* sumsq, int nslots, void* stream);
int vlni_sumsq_fold(const float* slots, int nslots, float* out, void* stream);
/* The grouped weight-gradient launches for either 16-bit dtype (1 bfloat16, 2 float16); arguments as the `_bf16_` forms above */
int vlni_gemm_tn_h16_grouped_v(int dtype, int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb,
                               float* C, long ldc, int N, int K, float* colsum, int split, int variant, void* stream);
int vlni_gemm_tn_h16_grouped_part(int dtype, int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb,
                                  float* part, long part_stride, int N, int K, float* colsum_part, int split, int variant,
                                  void* stream);

/* Fused masked attention, head dim 64, heads packed along the row (head h at column h*64), Sk <= 128.
 * kmask [B,Sk] additive float32 ((1-m)*-10000, R:1010-1012) or NULL; bias [B,Sq,Sk] additive float32
 * shared by all heads (graph_sprels, D:1145-1147) or NULL; lse [B,nh,Sq] float32.
 * Replaces BertSelfAttention.forward R:100-134 and BertOutAttention.forward R:326-353. */
int vlni_attn_fwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                  const float* kmask, const float* bias, void* out, long ldo, float* lse, int B, int nh, int Sq, int Sk,
                  float scale, float drop_p, unsigned drop_seed /* attention-prob dropout, R:120 */, void* stream);
int vlni_attn_bwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                  const float* kmask, const float* bias, const void* out, long ldo, const void* dout, long lddo,
                  const float* lse, void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, float* dbias, int B,
                  int nh, int Sq, int Sk, float scale, float drop_p, unsigned drop_seed, void* stream);
/* Two attention problems with the same batch, heads and scale in ONE launch (arrays of 2): the language / vision stream of a
   cross-modal layer or the two directions of its bidirectional cross-attention (R:385-407). bfloat16, q/k/v (dout) 16-byte
   aligned, strides multiples of 8, <= 256 keys; additive bias (and dbias0) on problem 0 only; otherwise VLNI_EUNSUP. */
int vlni_attn_fwd_dual(int dtype, const void* const* q, const long* ldq, const void* const* k, const long* ldk,
                       const void* const* v, const long* ldv, const float* const* kmask, const float* const* bias,
                       void* const* out, const long* ldo, float* const* lse, int B, int nh, const int* Sq, const int* Sk,
                       float scale, float drop_p, const unsigned* drop_seed, void* stream);
int vlni_attn_bwd_dual(int dtype, const void* const* q, const long* ldq, const void* const* k, const long* ldk,
                       const void* const* v, const long* ldv, const float* const* kmask, const float* const* bias,
                       const void* const* out, const long* ldo, const void* const* dout, const long* lddo,
                       const float* const* lse, void* const* dq, const long* lddq, void* const* dk, const long* lddk,
                       void* const* dv, const long* lddv, float* dbias0, int B, int nh, const int* Sq, const int* Sk,
                       float scale, float drop_p, const unsigned* drop_seed, void* stream);
/* softmax(q k^T * scale + kmask + bias) MATERIALISED as float32 [B][nh][Sq][Sk] (Sk <= 512) - only for the visualisation outputs of
   NavCMT.forward(..., return_cross_attention_probs=True) (reference vilmodel_cmt.py:391,393,438,439); same q / k layout as above */
int vlni_attn_probs(int dtype, const void* q, long ldq, const void* k, long ldk, const float* kmask, const float* bias, float* probs,
                    int B, int nh, int Sq, int Sk, float scale, void* stream);

/* LayerNorm over the last dim H in {256,512,768} (BertLayerNorm R:22; eps 1e-12, T:170-182 eps 1e-5).
 * bwd accumulates (+=) into dgamma/dbeta (float32; both NULL to skip). */
int vlni_layernorm_fwd(int dtype, const void* x, long ldx, const float* gamma, const float* beta, float eps, void* y,
                       long ldy, float* mean, float* rstd, int rows, int H, void* stream);
int vlni_layernorm_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, const float* gamma,
                       const float* mean, const float* rstd, void* dx, long lddx, float* dgamma, float* dbeta, int rows,
                       int H, const void* dres /* optional: dx += dres (pre-norm residual path) */, long lddres,
                       void* dx_drop /* optional second output dx*mask/(1-p): grad of the dropped dense output (R:146-147) */,
                       long lddxd, float drop_p, unsigned drop_seed, void* stream);
/* Two LayerNorm problems in one launch - the language and the vision stream of a cross-modal layer (R:144-148,186-190 on both
   streams of R:366-445): every pointer / stride / row-count argument is an array of 2; same dtype, H and eps. dres / dx_drop /
   drop_seed may be NULL. */
int vlni_layernorm_fwd_dual(int dtype, const void* const* x, const long* ldx, const float* const* gamma, const float* const* beta,
                            float eps, void* const* y, const long* ldy, float* const* mean, float* const* rstd, const int* rows, int H,
                            void* stream);
int vlni_layernorm_bwd_dual(int dtype, const void* const* dy, const long* lddy, const void* const* x, const long* ldx,
                            const float* const* gamma, const float* const* mean, const float* const* rstd, void* const* dx,
                            const long* lddx, float* const* dgamma, float* const* dbeta, const int* rows, int H,
                            const void* const* dres, const long* lddres, void* const* dx_drop, const long* lddxd, float drop_p,
                            const unsigned* drop_seed, void* stream);
/* y = LayerNorm(sum_k src_k), 1..4 sources, each dense / broadcast row (ld 0) / gathered by int64 row index,
 * float32 (parameter tables) or activation dtype; xsum (optional) keeps the pre-norm sum for backward.
 * Replaces BertEmbeddings R:58-73, ImageEmbeddings R:535-544, HistoryEmbeddings R:576-618, D:1087-1131. */
int vlni_sum_layernorm_fwd(int dtype, int n, const void* const* src, const long* src_ld, const long* const* src_idx,
                           const int* src_is_f32, const float* gamma, const float* beta, float eps, void* y, long ldy,
                           void* xsum, long ldxs, float* mean, float* rstd, int rows, int H, void* stream);

/* y = LayerNorm(x + bias + residual): the BertSelfOutput / BertOutput tail (R:144-148,186-190) for a dense output made elsewhere (this
 * library's own GEMM adds bias and residual in its epilogue and is followed by vlni_layernorm_fwd). bias / residual / xsum may be NULL;
 * xsum keeps the pre-norm sum for the backward. bwd: dx is the gradient of x AND of residual; dgamma / dbeta / dbias accumulate (+=),
 * dbias may be NULL. */
int vlni_bias_residual_layernorm_fwd(int dtype, const void* x, long ldx, const float* bias, const void* residual, long ldr,
                                     const float* gamma, const float* beta, float eps, void* y, long ldy, void* xsum, long ldxs, float* mean,
                                     float* rstd, int rows, int H, void* stream);
int vlni_bias_residual_layernorm_bwd(int dtype, const void* dy, long lddy, const void* xsum, long ldxs, const float* gamma,
                                     const float* mean, const float* rstd, void* dx, long lddx, float* dgamma, float* dbeta, float* dbias,
                                     int rows, int H, void* stream);

int vlni_cast(int src_dtype, int dst_dtype, const void* src, void* dst, long n, void* stream);
/* dst[c][r] = src[r][c] (r < R), zero for R <= r < Rpad: wgrad operands and transposed weight shadows */
int vlni_transpose(int src_dtype, int dst_dtype, const void* src, long lds, void* dst, long ldd, int R, int C, int Rpad,
                   void* stream);
/* Many transposes in one launch (all transposed weight shadows after an optimizer step). table_dev: n entries of 56 bytes in
   DEVICE memory {const void* src; void* dst; long lds; long ldd; int R, C, Rpad, tile0, tiles_c, 0}, tile0 = running sum of
   ceil(C/64)*ceil(Rpad/64) over the entries before it; total_tiles = the grand total. dst (bfloat16) [c][r] = src[r][c]. */
int vlni_transpose_batched(int src_dtype, const void* table_dev, int n, int total_tiles, void* stream);
/* 16-bit shadows of many float32 parameters in one launch: table = device array of
   { const float* src; void* dst; long lds, ldd; int R, C, mode, tile0, tiles_c, pad; } - mode 0: dst[r][c] = src[r][c], mode 1: dst[c][r] =
   src[r][c]; tile0 = first 64 x 64 tile (= block) of the entry, tiles_c = tiles per row of tiles; entries sorted by tile0. The weights an
   optimizer other than vlni_adamw_step_groups has just stepped (torch.optim.AdamW, r2r/agent_cmt.py:98) are re-cast in one launch. */
int vlni_shadow_refresh(int dst_dtype, const void* table_dev, int n, int total_tiles, void* stream);
/* out[n] += sum_r x[r][n]: bias gradients */
int vlni_colsum(int dtype, const void* x, long ldx, int rows, int N, float* out, void* stream);
/* y = x W^T + b with K <= 16 float32 features (angle 4-d R:537,599; DUET 7-/14-d position D:1093,1140-1150) */
int vlni_smallk_linear_fwd(int dtype, const float* x, long ldx, const float* W, const float* b, void* y, long ldy, int rows,
                           int N, int K, void* stream);
int vlni_smallk_linear_bwd(int dtype, const void* dy, long lddy, const float* x, long ldx, float* dW, float* db, int rows,
                           int N, int K, void* stream);
/* table_grad[idx[r]] += src[r] (nn.Embedding backward) for idx[r] in [0, table_rows); idx NULL: all rows into row 0 */
int vlni_scatter_add_rows(int dtype, const void* src, long lds, const long* idx, float* table_grad, int rows, int H,
                          int table_rows, void* stream);
/* the same for a table of <= 8 rows whose indices lie in [0, table_rows) (navigation-type / token-type embeddings): register
   accumulators per table row, block totals as atomics - thousands of rows on 2-3 table rows serialise the per-element atomics */
int vlni_scatter_add_rows_small(int dtype, const void* src, long lds_, const long* idx, float* table_grad, int rows, int H,
                                int table_rows, void* stream);
/* out[b] = mean_s x[b][s] (torch.mean(pano_embeddings, 1) R:612) */
/* lens (int64 [B]) non-NULL: mean over the first lens[b] rows only (DUET's masked panorama mean, r2r/agent.py:159-161) */
int vlni_seqmean_fwd(int dtype, const void* x, void* out, const long* lens, int B, int S, int H, void* stream);
int vlni_seqmean_bwd(int dtype, const void* dout, void* dx, const long* lens, int B, int S, int H, void* stream);
/* f[b][j][:] = visn[b][r0+j][:] * lang[b][0][:]  (action-head input `ob_embeds * txt_embeds[:, :1]`, R:1192; visn [B,Sv,H], lang [B,Sl,H], f [B,n,H],
   all contiguous) and its backward: dvisn / dlang are written in full (zero outside the gated rows / row 0); either may be NULL */
int vlni_gate_rows_fwd(int dtype, const void* visn, const void* lang, void* f, int B, int Sv, int Sl, int r0, int n, int H, void* stream);
int vlni_gate_rows_bwd(int dtype, const void* df, const void* visn, const void* lang, void* dvisn, void* dlang, int B, int Sv, int Sl,
                       int r0, int n, int H, void* stream);
/* logits[r] = mask[r] ? -inf : <h[r], w> + bias  (NextActionPrediction.net.4 R:960 + masked_fill_ R:1200) */
int vlni_rowdot_fwd(int dtype, const void* h, long ldh, const float* w, const float* bias, const unsigned char* mask,
                    float* out, int rows, int H, void* stream);
int vlni_rowdot_bwd(int dtype, const float* dl, const void* h, long ldh, const float* w, const unsigned char* mask,
                    void* dh, long lddh, float* dw, float* dbias, int rows, int H, void* stream);
/* CrossEntropyLoss(ignore_index, reduction='sum') (r2r/agent_cmt.py:105,547): loss_sum += ..., dlogits = softmax - onehot */
int vlni_cross_entropy(const float* logits, long ld, const long* target, long ignore_index, float* loss_sum,
                       float* dlogits, long lddl, int rows, int V, void* stream);
/* aux head (AlignWithContrastiveLoss R:737-790): noun-phrase token mean (CSR: seg_off[nseg+1], rowidx) ... */
int vlni_segment_mean_fwd(int dtype, const void* x, long ldx, const int* seg_off, const int* rowidx, void* out, int nseg,
                          int H, void* stream);
int vlni_segment_mean_bwd(int dtype, const void* dout, const int* seg_off, const int* rowidx, float* dx32, int nseg, int H,
                          void* stream);
/* ... and F.cosine_similarity(eps) per row (R:782) */
int vlni_cosine_fwd(int dtype, const void* x, const void* y, float eps, float* cosv, float* nx, float* ny, int rows, int H,
                    void* stream);
int vlni_cosine_bwd(int dtype, const void* x, const void* y, const float* gcos, const float* cosv, const float* nx,
                    const float* ny, void* dx, void* dy, int rows, int H, void* stream);

/* Pairwise dot products of two small float32 row sets: out[na, nb] = a[na, H] b[nb, H]^T (similarity matrix of the InfoNCE / margin
   alignment losses against in-batch negatives, vilmodel_cmt.py:793-856) and its backward (da / db may be NULL). */
int vlni_pairdot_fwd(const float* a, const float* b, float* out, int na, int nb, int H, void* stream);
int vlni_pairdot_bwd(const float* g, const float* a, const float* b, float* da, float* db, int na, int nb, int H, void* stream);
/* y = x * mask/(1-p), counter-based mask f(seed, linear index); x NULL writes the scaled mask */
int vlni_dropout(int dtype, const void* x, void* y, long n, float p, unsigned seed, void* stream);
/* dz = da * act'(z), act 1 gelu-erf / 2 relu (n multiple of 4) */
int vlni_act_bwd(int dtype, int act, const void* da, const void* z, void* dz, long n, void* stream);
/* Observation / panorama input tensors built on the device from a resident view-feature table [n_viewpoints][36][D] (float32 or
   bfloat16): replaces the host-side numpy assembly + per-step host->device copy of r2r/agent_cmt.py:130-176 (HAMT
   _cand_pano_feature_variable) and map_nav_src/r2r/agent.py:67-97 (DUET _panorama_feature_variable). Per slot (b, v):
   view[b,v] < 0 -> zeros ([STOP], padding); else image = table[vp_row[b]][view[b,v]], angle = is_cand ? [sin h, cos h, sin e, cos e]*
   of cand_he[b,v,:] : angle_table[base_view[b]][view[b,v]] (data_utils.py:481-534). out_img [B,V,D], out_ang [B,V,A] float32. */
int vlni_build_views(int table_dtype, const void* table, const long* vp_row, const int* view, const float* cand_he,
                     const unsigned char* is_cand, const int* base_view, const float* angle_table, float* out_img, float* out_ang,
                     int B, int V, int D, int A, void* stream);
/* DUET's topological map resident on the device, batched over the B episodes of a rank; replaces the per-episode python
   dict graph of VLN-DUET/map_nav_src/models/graph_utils.py:43-161 and the python double loops of r2r/agent.py:98-207.
   Per episode: pos [G][3] float64, dis [G][G] float64 (95959595 = nothing known), via [G][G] int32 (intermediate node of a relaxed
   pair, -1 = direct edge / nothing), seen [G] (FloydGraph.visited). G <= 256 node slots; the host assigns slots to viewpoint names.
   vlni_graph_init      fills dis / via / seen.
   vlni_graph_observe   GraphMap.update_graph (graph_utils.py:107-113): cur[B] = slot of the current viewpoint (-1: episode ended, no
                        change), cand[B][C] candidate slots (-1 pad) with positions; edge kept if shorter, then every pair is relaxed
                        through cur; n_nodes[B] = slots in use after this observation. cand_dist[B][C] (optional) = edge lengths
                        computed by the caller: the reference squares with python `**`, i.e. libm pow, which is one ulp off
                        x*x for ~0.1 % of the inputs; NULL = exact float64 sqrt(dx*dx + dy*dy + dz*dz) on the device.
   vlni_graph_pos_fts   GraphMap.get_pos_fts (graph_utils.py:130-153) for nodes[B][N] (-1 = the [stop] token -> 0 1 0 1 0 0 0,
                        -2 = padding row -> zeros): A angle features of (heading, elevation) relative to heading[b] / elevation[b]
                        (angles cast to float32 first, as the reference does) + line distance / 30, shortest distance / 30, hops / 10;
                        hops = len(FloydGraph.path), expanded lazily from `via`.  Row r of episode b goes to out + b*ld_batch + r*ld_row
                        (so the 14-wide vp_pos_fts of r2r/agent.py:178-195 is two calls).  status[0] becomes 1 + b if a hop-count walk of
                        episode b did not terminate (inconsistent map); the caller zeroes it.
   vlni_graph_pair_dists  gmap_pair_dists of r2r/agent.py:135-139,157-160: out[B][N][N] float32, 0 on the diagonal, for [stop] and padding.
   vlni_gather_rows_or_zero  out[i] = rows[i] >= 0 ? table[rows[i]][0:D] : 0 (float32 out, float32 / bfloat16 table): the slot fill of
                        _create_diffusion_imaginations_v2 (VLN-HAMT/finetune_src/r2r/agent_cmt.py:286-309) from a resident table. */
int vlni_graph_init(double* dis, int* via, unsigned char* seen, int B, int G, void* stream);
int vlni_graph_observe(double* pos, double* dis, int* via, unsigned char* seen, const int* cur, const int* cand,
                       const double* cur_pos, const double* cand_pos, const double* cand_dist, const int* n_nodes, int B, int G, int C,
                       void* stream);
int vlni_graph_pos_fts(const double* pos, const double* dis, const int* via, const int* cur, const int* nodes, const double* heading,
                       const double* elevation, float* out, long ld_row, long ld_batch, int* status, int B, int G, int N, int A,
                       void* stream);
int vlni_graph_pair_dists(const double* dis, const int* nodes, float* out, int B, int G, int N, void* stream);
int vlni_gather_rows_or_zero(int dtype, const void* table, long ld, const long* rows, float* out, int n, int D, void* stream);
/* DUET global/local logit fusion, replaces the per-sample python loop of VLN-DUET/map_nav_src/models/vilmodel.py:1198-1217:
   src[B,G]: local candidate index that IS map node g (>= 0), -2 = unvisited node without a candidate (takes the summed local
   logits of the visited candidates), -1 = nothing to add; bw[B,V]: candidate j is an already-visited viewpoint.
   fwd: out = gl + gathered ll (float32 logits, may hold -inf); bwd: dll from dout (dgl = dout) */
int vlni_duet_fuse_fwd(const float* gl, const float* ll, const int* src, const unsigned char* bw, float* out, int B, int G, int V,
                       void* stream);
int vlni_duet_fuse_bwd(const float* dout, const int* src, const unsigned char* bw, float* dll, int B, int G, int V, void* stream);
/* The whole logit tail of a DUET navigation call in one launch (VLN-DUET/map_nav_src/models/vilmodel.py:1185-1217):
   w = sigmoid(f[b]) (f NULL: 0.5); gl = (visited | !gmask) ? -inf : graw * w; ll = nav ? lraw * (1 - w) : -inf; fused = duet_fuse(gl, ll).
   graw [B,G] / lraw [B,V]: outputs of the two ClsPrediction heads (float32); f [B]: pre-sigmoid output of sap_fuse_linear;
   visited, gmask [B,G] and nav [B,V]: 0/1 bytes. bwd: any of d_gl / d_ll / d_fused may be NULL; writes dgraw, dlraw and (f given) df. */
int vlni_duet_heads_fwd(const float* graw, const float* lraw, const float* f, const unsigned char* visited, const unsigned char* gmask,
                        const unsigned char* nav, const int* src, const unsigned char* bw, float* gl, float* ll, float* fused, int B,
                        int G, int V, void* stream);
int vlni_duet_heads_bwd(const float* d_gl, const float* d_ll, const float* d_fused, const float* graw, const float* lraw, const float* f,
                        const unsigned char* visited, const unsigned char* gmask, const unsigned char* nav, const int* src,
                        const unsigned char* bw, float* dgraw, float* dlraw, float* df, int B, int G, int V, void* stream);
/* optimizer side of the measured step (r2r/agent_cmt.py:827-832): clip_grad_norm_ + AdamW over a flat arena */
int vlni_adamw_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, const float* clip_coef, void* stream);
int vlni_sumsq(const float* g, long n, float* sumsq, void* stream);
int vlni_clip_coef(const float* sumsq, float max_norm, float* coef, void* stream);
/* the same step with every per-step quantity on the device (replayable inside a captured hipGraph): state = 4 floats
   [clip factor, 1-beta1^t, 1-beta2^t, t], zeroed once by the caller; vlni_optim_prepare advances t and fills the rest */
int vlni_optim_prepare(const float* sumsq, float max_norm, float beta1, float beta2, float* state, void* stream);
int vlni_adamw_step_dev(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, const float* lr_dev,
                        float beta1, float beta2, float eps, float weight_decay, const float* state, void* stream);
/* Parameter-group form (the optimizer of the shipped "variant4" warm-up: r2r/agent_cmt.py:82-96, r2r/main.py:202-255): the
   arena is laid out group by group (ngrp <= 8, grp_end[k] = first element past group k, device int64); grp_lr (device,
   2*ngrp floats) = learning rate per group, then a trainable flag per group (0 = the group is skipped altogether, like
   torch.optim skips parameters without a gradient); gstate (device, 4*ngrp floats, zeroed once) = per-group step count and
   bias corrections; state (device, 8 floats, zeroed once): [0] clip factor / loss scale, [3] calls, [4] loss scale S the caller
   multiplied the loss by (0 = 1), [5] 1 = this step was skipped, [6] good steps since S changed, [7] unscaled gradient norm.
   growth_interval > 0 = dynamic loss scaling (torch.cuda.amp.GradScaler, VLN-DUET/pretrain_src/train_r2r.py:201-234): a
   non-finite gradient sum skips the step and halves S, growth_interval good steps double it. shadow_dtype: dtype of the
   compute-dtype mirror of the parameters kept current by the step (1 bf16, 2 f16), shadow may be null. */
int vlni_optim_prepare_groups(const float* sumsq, float max_norm, float beta1, float beta2, float* state, float* gstate,
                              const float* grp_lr, int ngrp, int growth_interval, void* stream);
int vlni_adamw_step_groups(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, long n,
                           const long* grp_end, const float* grp_lr, const float* gstate, int ngrp, float beta1, float beta2,
                           float eps, float weight_decay, const float* state, void* stream);
/* dst[i] = (out dtype)(src[i] * scale), float32 <-> bfloat16: packs / unpacks the gradient all-reduce payload (DDP's gradient
   averaging with bf16 compression, r2r/agent_cmt.py:61-63); src 16-byte, dst 8-byte aligned */
int vlni_scale_cast(int dt_in, int dt_out, const void* src, void* dst, long n, float scale, void* stream);

/* Fused embed-and-combine (round 5): y = dropout([LN_o]([LN_a](a) + LN_b(f W_b^T + b_b) + row + table[idx] + table2[idx2] + extra)) in ONE launch -
 * the observation / history / panorama / map-node embeddings (ImageEmbeddings R:535-544, HistoryEmbeddings R:596-618, D:1087-1131, D:1140-1156) that were
 * LayerNorm + small-K linear + LayerNorm + sum-LayerNorm + dropout. a [rows, H] is the image linear's output; f [rows, K <= 16] float32 angle / position
 * features; optional parts NULL. Writes what the backward needs: linb (the small linear's output), xsum (pre-LN_o sum), the LayerNorms' statistics.
 * Backward = the existing operators (vlni_dropout, vlni_layernorm_bwd x3, vlni_smallk_linear_bwd, vlni_colsum, vlni_scatter_add_rows). */
int vlni_embed_combine_fwd(int dtype, const void* a, long lda, const float* ga, const float* ba, const float* f, long ldf, int K,
                           const float* Wb, const float* bb, const float* gb, const float* beb, const void* extra, long lde,
                           const float* row, const float* table, const long* idx, int table_rows, const float* table2, const long* idx2,
                           int table2_rows, const float* go, const float* bo, float eps, void* linb, void* xsum, void* y, long ldy, float* mean_a,
                           float* rstd_a, float* mean_b, float* rstd_b, float* mean_o, float* rstd_o, float drop_p, unsigned drop_seed,
                           int rows, int H, void* stream);

/* Prediction-head tail in one launch (round 5): out[r] = mask[r] ? -inf : <dropout(LayerNorm(x[r])), w> + bias - NextActionPrediction R:953-963 + R:1200 and
 * ClsPrediction D:1009-1020 after their first Linear + ReLU. hd [rows, H] (the dropped LayerNorm output) and mean / rstd are kept for the backward
 * (vlni_rowdot_bwd, the dropout mask, vlni_layernorm_bwd). */
int vlni_ln_rowdot_fwd(int dtype, const void* x, long ldx, const float* gamma, const float* beta, float eps, void* hd, float* mean, float* rstd,
                       const float* w, const float* bias, const unsigned char* mask, float* out, float drop_p, unsigned drop_seed, int rows, int H,
                       void* stream);

/* ---- block-level entry points (round 5; csrc/blocks.hip) -------------------------------------------------------------------------
 * One call = one transformer SUBLAYER of one or two streams, forward or backward: the function issues the sublayer's own 4-7 launches
 * (vlni_gemm_nt_multi, vlni_attn_*, vlni_layernorm_*) on `stream`, so an eager caller crosses the boundary once per sublayer and direction.
 *   self-attention block  y = LN(dense(attn(x Wq, x Wk, x Wv)) + x)    BertAttention R:151-161 (R:100-134, R:144-148)
 *   FFN block             y = LN(W2 gelu(W1 x + b1) + b2 + x)         BertIntermediate R:173-176 + BertOutput R:186-190
 * n = 1: one stream (text encoder, history panorama encoder); n = 2: the two streams of a cross-modal layer (R:399-421) / DUET's global
 * and local branches as dual-problem launches. Caller-owned buffers throughout; weight / bias gradients stay separate calls
 * (vlni_gemm_tn_h16_grouped_*: dW_out = dmid_drop^T aux|mid, dW_in = dmid|daux^T x). */
typedef struct VlniBlockSide {
  int B, S;                                   /* samples, tokens per sample */
  const void* x; long ldx;                    /* input [B*S, H] */
  const float* kmask;                         /* attention: additive key mask [B, S] or NULL */
  const void* w_in; const float* b_in;        /* attention: packed Wqkv [3H, H] / bqkv; FFN: W1 [FF, H] / b1 (weights in the compute dtype) */
  const void* w_out; const float* b_out;      /* attention: Wo [H, H] / bo; FFN: W2 [H, FF] / b2 */
  const void* wt_in; const void* wt_out;      /* backward: transposed weights (dgrad operands): Wqkv^T [H, 3H] / W1^T [H, FF]; Wo^T / W2^T [FF, H] */
  const float* gamma; const float* beta;
  unsigned seed_attn, seed_dense;             /* dropout seeds: attention probabilities / dense output */
  void* mid;                                  /* attention: packed qkv [B*S, 3H]; FFN: gelu(z) [B*S, FF] */
  void* aux;                                  /* attention: context [B*S, H]; FFN: z (act 1) or GELU'(z) (act 3) [B*S, FF] */
  float* lse;                                 /* attention: [B, nh, S] */
  void* pre;                                  /* pre-LayerNorm sum [B*S, H] */
  void* y; float* mean; float* rstd;
  const void* dy; long lddy;
  void* dpre;                                 /* d(pre) */
  void* dmid_drop;                            /* d(dense output) = d(pre) * mask / (1 - p); == dpre when p_hidden == 0 */
  void* daux;                                 /* attention: d(context); FFN: d(z) */
  void* dmid;                                 /* attention: d(qkv) */
  void* dx;                                   /* or NULL: the input needs no gradient */
  float* dgamma; float* dbeta;                /* accumulated into, or NULL */
} VlniBlockSide;
typedef struct VlniBlockArgs {
  int dtype, n, H, FF, nh;
  float eps, p_attn, p_hidden;
  int act, dact;                              /* FFN epilogue codes of vlni_gemm_nt (1 / 1 or 3 / 3) */
  int v_in, v_out;                            /* GEMM pipeline ids (`variant` of vlni_gemm_nt_v) of the in / out projections of this direction */
  const float* bias0; float* dbias0;          /* attention, stream 0: additive [B, S, S] score bias (D:1145-1147) and its gradient */
  VlniBlockSide s[2];
} VlniBlockArgs;
int vlni_self_att_block_fwd(const VlniBlockArgs* a, void* stream);
int vlni_self_att_block_bwd(const VlniBlockArgs* a, void* stream);
int vlni_ffn_block_fwd(const VlniBlockArgs* a, void* stream);
int vlni_ffn_block_bwd(const VlniBlockArgs* a, void* stream);

/* SURVEY.md section 8(b) lists a minimum operator set by name; four of those names are these entry points (same signatures):
 *   vlni_gemm_bias_act_fwd / _bwd        -> vlni_gemm_nt (forward: bias + act epilogue; backward: the dgrad launch with dact / dact_src)
 *   vlni_embed_sum_layernorm_fwd / _bwd  -> vlni_sum_layernorm_fwd / vlni_layernorm_bwd (+ vlni_scatter_add_rows for gathered tables)
 *   vlni_cosine_loss_fwd / _bwd          -> vlni_cosine_fwd / vlni_cosine_bwd (the alignment head's 1 - cos terms, R:737-790)
 *   vlni_masked_fill_head_fwd / _bwd     -> vlni_rowdot_fwd / vlni_rowdot_bwd (Linear(768 -> 1) + masked_fill(-inf), R:953-963,1200)
 * vlni_attn_fwd/bwd, vlni_bias_residual_layernorm_fwd/bwd and vlni_segment_mean_fwd/bwd carry the survey's names already. */
#define vlni_gemm_bias_act_fwd vlni_gemm_nt
#define vlni_gemm_bias_act_bwd vlni_gemm_nt
#define vlni_embed_sum_layernorm_fwd vlni_sum_layernorm_fwd
#define vlni_embed_sum_layernorm_bwd vlni_layernorm_bwd
#define vlni_cosine_loss_fwd vlni_cosine_fwd
#define vlni_cosine_loss_bwd vlni_cosine_bwd
#define vlni_masked_fill_head_fwd vlni_rowdot_fwd
#define vlni_masked_fill_head_bwd vlni_rowdot_bwd

#ifdef __cplusplus
}
#endif
#endif
