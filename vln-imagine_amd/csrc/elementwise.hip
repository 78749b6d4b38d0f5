// HBM-bound row/element kernels of the hot path: dtype casts, tile transposes (wgrad operands and
// transposed weight shadows), column sums (bias gradients), the tiny-K feature projections
// (angle / position features, K <= 16), embedding-gradient scatter, sequence mean, the action-head
// row dot + nav-type mask, masked cross-entropy, and the imagination-grounding auxiliary pieces
// (noun-phrase segment mean, cosine loss). All coalesced 16-B/8-B accesses; reductions through
// wavefront shuffles; f32 accumulation everywhere.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) DT<D>::st4(d + i, DT<S>::ld4(s + i));
  if (i < n && i + 3 >= n)
    for (long j = i; j < n; ++j) DT<D>::st(d + j, DT<S>::ld(s + j));
}

// dst[c][r] = src[r][c] for r < R, zero for R <= r < Rpad.   64x64 tiles through LDS.
template <typename S, typename D>
__global__ __launch_bounds__(256) void transpose_kernel(const S* __restrict__ src, long lds_, D* __restrict__ dst, long ldd,
                                                        int R, int C, int Rpad) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? DT<S>::ld(src + (long)r * lds_ + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) DT<D>::st(dst + (long)c * ldd + r, tile[tx][i]);
  }
}

// The same 64x64 tile transpose for MANY matrices in one launch (all transposed weight shadows after an optimizer step):
// block -> entry by binary search over the entries' first-tile indices. dst has the dtype of src (the compute-dtype mirror).
struct TrEntry { const void* src; void* dst; long lds, ldd; int R, C, Rpad, tile0, tiles_c, pad_; };
template <typename S>
__global__ __launch_bounds__(256) void transpose_batched_kernel(const TrEntry* __restrict__ tab, int n) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const TrEntry e = tab[lo];
  const int t = blockIdx.x - e.tile0;
  const int r0 = (t / e.tiles_c) * 64, c0 = (t % e.tiles_c) * 64;
  const S* src = (const S*)e.src;
  S* dst = (S*)e.dst;
  if constexpr (sizeof(S) == 2) {
    // full 16-bit tiles with 16-byte aligned rows: 16-byte global accesses on both sides (the scalar path moves 2 bytes per lane: 2.2 TB/s
    // over all weight shadows of a step), 2-byte LDS accesses in between (row pitch 66 elements = 33 words: the 8 rows a lane gathers and
    // the 8 columns of a wave fall in different banks)
    const bool vec = r0 + 64 <= e.R && r0 + 64 <= e.Rpad && c0 + 64 <= e.C && (e.lds & 7) == 0 && (e.ldd & 7) == 0
                     && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    if (vec) {
      typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
      unsigned short (*t16)[66] = (unsigned short (*)[66])tile;
      const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;            // 8 lanes per 128-byte row piece, 32 rows per pass
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = rr + h * 32;
        const u16x8 v = *(const u16x8*)((const unsigned short*)src + (long)(r0 + r) * e.lds + c0 + q * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) t16[r][q * 8 + j] = v[j];
      }
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = rr + h * 32;
        u16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = t16[q * 8 + j][c];
        *(u16x8*)((unsigned short*)dst + (long)(c0 + c) * e.ldd + r0 + q * 8) = v;
      }
      return;
    }
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < e.R && c < e.C) ? DT<S>::ld(src + (long)r * e.lds + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < e.C && r < e.Rpad) DT<S>::st(dst + (long)c * e.ldd + r, tile[tx][i]);
  }
}

// 16-bit shadows of MANY float32 parameters in one launch (what an optimizer that is not the library's own leaves stale after its step):
// entry = one parameter -> its place in a shadow; mode 0: dst[r][c] = src[r][c] (biases, LayerNorm vectors: R = 1), mode 1: dst[c][r] = src[r][c]
// (the dgrad operand W^T). A block moves one 64 x 64 tile; row-packed shadows (Q | K | V) are one entry per parameter.
struct ShEntry { const float* src; void* dst; long lds, ldd; int R, C, mode, tile0, tiles_c, pad_; };
template <typename D>
__global__ __launch_bounds__(256) void shadow_refresh_kernel(const ShEntry* __restrict__ tab, int n) {
  __shared__ float tile[64][65];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ShEntry e = tab[lo];
  const int t = blockIdx.x - e.tile0;
  const int r0 = (t / e.tiles_c) * 64, c0 = (t % e.tiles_c) * 64;
  D* dst = (D*)e.dst;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if (e.mode == 0) {
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i, c = c0 + tx;
      if (r < e.R && c < e.C) DT<D>::st(dst + (long)r * e.ldd + c, e.src[(long)r * e.lds + c]);
    }
    return;
  }
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < e.R && c < e.C) ? e.src[(long)r * e.lds + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < e.C && r < e.R) DT<D>::st(dst + (long)c * e.ldd + r, tile[tx][i]);
  }
}
extern "C" int vlni_shadow_refresh(int dst_dtype, const void* table_dev, int n, int total_tiles, void* stream) {
  VLNI_CHECK(table_dev && n > 0 && total_tiles > 0, VLNI_EINVAL, "shadow_refresh: n=%d tiles=%d", n, total_tiles);
  const ShEntry* tab = (const ShEntry*)table_dev;
  if (dst_dtype == VLNI_BF16) hipLaunchKernelGGL((shadow_refresh_kernel<__bf16>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, tab, n);
  else if (dst_dtype == VLNI_F16) hipLaunchKernelGGL((shadow_refresh_kernel<_Float16>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, tab, n);
  else if (dst_dtype == VLNI_F32) hipLaunchKernelGGL((shadow_refresh_kernel<float>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, tab, n);   // packed float32 copies (Q | K | V biases)
  else { vlni_set_error("shadow_refresh: bad dtype %d", dst_dtype); return VLNI_EINVAL; }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// out[n] += sum_r x[r][n]      (bias gradients).  grid.x over column groups of 256, grid.y over row slabs.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long ldx, int rows, int N,
                                                     float* __restrict__ out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  const int per = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float a = 0.f;
  for (int r = r0; r < r1; ++r) a += DT<T>::ld(x + (long)r * ldx + col);
  atomicAdd(out + col, a);
}

// y[r][n] = b[n] + sum_k x[r][k] W[n][k],  K <= 16, x f32 features, W/b f32 parameters.
template <typename T>
__global__ __launch_bounds__(256) void smallk_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ W,
                                                         const float* __restrict__ b, T* __restrict__ y, long ldy, int rows,
                                                         int N, int K) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float w[16];
  for (int k = 0; k < K; ++k) w[k] = W[(long)n * K + k];
  const float bv = b ? b[n] : 0.f;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    float a = bv;
    for (int k = 0; k < K; ++k) a += x[(long)r * ldx + k] * w[k];
    DT<T>::st(y + (long)r * ldy + n, a);
  }
}
// dW[n][k] += sum_r dy[r][n] x[r][k];  db[n] += sum_r dy[r][n].  grid (ceil(N / 256), row slabs): a block owns rows
// [slab * rpb, (slab + 1) * rpb); their x rows (<= 64 x 16 floats) go to LDS first so that the row loop holds nothing but independent
// dy loads, eight in flight (the loop is latency-bound: 24 us -> see tools/kernel_calls.py).
template <typename T>
__global__ __launch_bounds__(256) void smallk_bwd_kernel(const T* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                         long ldx, float* __restrict__ dW, float* __restrict__ db, int rows,
                                                         int N, int K, int rpb) {
  // block = 64 output columns x one slab of <= 256 rows; its 4 waves take every 4th row of the slab, their partial sums meet in LDS
  // and only 64 x (K + 1) block totals go out as float atomics (one thread per column and slab of 8-64 rows, as before, made
  // (rows / 8) x N x (K + 1) contended atomics: 45-90 us for DUET's 1184 x 14 position features)
  __shared__ float xs[256 * 16];
  __shared__ float red[4][17][64];
  const int r0 = blockIdx.y * rpb, nr = min(rpb, rows - r0);
  for (int i = threadIdx.x; i < nr * K; i += 256) xs[i] = x[(long)(r0 + i / K) * ldx + i % K];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  float a[16], s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 0.f;
  if (n < N) {
    const T* dp = dy + (long)r0 * lddy + n;
    int r = w;
    for (; r + 28 < nr; r += 32) {                           // eight of this wave's rows in flight (the loop is latency-bound)
      float d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = DT<T>::ld(dp + (long)(r + 4 * u) * lddy);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += d[u];
#pragma unroll
        for (int k = 0; k < 16; ++k)                         // full unroll + uniform guard: a[] stays in registers (K is a run-time value)
          if (k < K) a[k] += d[u] * xs[(r + 4 * u) * K + k];
      }
    }
    for (; r < nr; r += 4) {
      const float d = DT<T>::ld(dp + (long)r * lddy);
      s += d;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < K) a[k] += d * xs[r * K + k];
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) red[w][k][lane] = a[k];
  red[w][16][lane] = s;
  __syncthreads();
  // the block's 64 x K gradient entries are CONTIGUOUS in dW ([N][K] row-major): consecutive threads add to consecutive addresses
  const int ncol = min(64, N - blockIdx.x * 64);
  float* const dWb = dW + (long)blockIdx.x * 64 * K;
  for (int i = threadIdx.x; i < ncol * K; i += 256) {
    const int c = i / K, k = i - c * K;
    atomicAdd(dWb + i, (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]));
  }
  if (db && threadIdx.x < ncol) {
    const int c = threadIdx.x;
    atomicAdd(db + blockIdx.x * 64 + c, (red[0][16][c] + red[1][16][c]) + (red[2][16][c] + red[3][16][c]));
  }
}

// table_grad[idx[r]][:] += src[r][:]   (embedding backward; idx == null: every row goes to row 0)
template <typename T>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const T* __restrict__ src, long lds_,
                                                               const long* __restrict__ idx, float* __restrict__ tab, int rows,
                                                               int H, int table_rows, int* __restrict__ ierr) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const long t = idx ? idx[r] : 0;
    if (t < 0 || t >= table_rows) {                           // an index outside the table: nothing is added, the row is counted
      if (ierr && threadIdx.x == 0) atomicAdd(ierr, 1);
      continue;
    }
    for (int c = threadIdx.x; c < H; c += 256) atomicAdd(tab + t * H + c, DT<T>::ld(src + (long)r * lds_ + c));
  }
}

// The same for a table of at most 8 rows (navigation-type / token-type embeddings: thousands of source rows land on 2-3 table rows, so
// per-element global atomics serialise ~800-deep on every address: 68 us per call). A block takes 32 source rows x 256 columns,
// every thread keeps one register accumulator per table row for its column and only the block totals go out as atomics.
template <typename T>
__global__ __launch_bounds__(256) void scatter_add_rows_small_kernel(const T* __restrict__ src, long lds_, const long* __restrict__ idx,
                                                                     float* __restrict__ tab, int rows, int H, int TR, int* __restrict__ ierr) {
  constexpr int RPB = 32;
  const int c = blockIdx.y * 256 + threadIdx.x, r0 = blockIdx.x * RPB;
  if (c >= H) return;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const int r1 = min(rows, r0 + RPB);
  for (int r = r0; r < r1; ++r) {
    const int t = (int)idx[r];                                // uniform over the block
    if ((t < 0 || t >= TR) && ierr && c == 0) atomicAdd(ierr, 1);      // matches no accumulator below: nothing is added
    const float v = DT<T>::ld(src + (long)r * lds_ + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += (t == k) ? v : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < TR && acc[k] != 0.f) atomicAdd(tab + (long)k * H + c, acc[k]);
}

// out[b][:] = mean over the first n rows of x[b] (n = lens[b], or S when lens is null) and its backward dx[b][s][:] = s < n ? dout[b][:] / n : 0
// (lens: the masked panorama mean of VLN-DUET/map_nav_src/r2r/agent.py:159-161)
template <typename T>
__global__ __launch_bounds__(256) void seqmean_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, const long* __restrict__ lens, int B,
                                                          int S, int H) {
  const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;      // grid (B, ceil(H / 256)): 3x the blocks of one-per-sample
  if (c >= H) return;
  const int n = lens ? min(max((int)lens[b], 1), S) : S;             // clamped: see seqmean_bwd_kernel
  const T* xp = x + (long)b * S * H + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 3 < n; s += 4) {                                         // four rows in flight (the loop is latency-bound)
    a0 += DT<T>::ld(xp + (long)s * H); a1 += DT<T>::ld(xp + (long)(s + 1) * H);
    a2 += DT<T>::ld(xp + (long)(s + 2) * H); a3 += DT<T>::ld(xp + (long)(s + 3) * H);
  }
  for (; s < n; ++s) a0 += DT<T>::ld(xp + (long)s * H);
  DT<T>::st(out + (long)b * H + c, ((a0 + a1) + (a2 + a3)) / n);
}
template <typename T>
__global__ __launch_bounds__(256) void seqmean_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, const long* __restrict__ lens, int B,
                                                          int S, int H) {
  const int b = blockIdx.x;
  const int n = lens ? min(max((int)lens[b], 1), S) : S;             // a device-side length is not validated by the host: 0 would divide by zero, > S read past the rows
  const float inv = 1.0f / n;
  for (int c = threadIdx.x; c < H; c += 256) {
    const float g = DT<T>::ld(dout + (long)b * H + c) * inv;
    for (int s = 0; s < S; ++s) DT<T>::st(dx + ((long)b * S + s) * H + c, s < n ? g : 0.f);
  }
}

// f[b][j][:] = visn[b][r0 + j][:] * lang[b][0][:]: the action head's input for act_pred_token == 'ob_txt'
// (VLN-HAMT/finetune_src/models/vilmodel_cmt.py:1192, `ob_embeds * txt_embeds[:, :1]`), and its backward, which writes BOTH full-size
// gradients in one launch: dvisn rows [r0, r0 + n) = df * lang[b][0], its other rows zero; dlang row 0 = sum_j df[b][j] * visn[b][r0 + j],
// its other rows zero. One block per output row (grid (B, Sv + Sl)); torch needed ten kernels for it (two zero fills, two slices'
// copies, casts, a multiply, a reduction).
template <typename T>
__global__ __launch_bounds__(256) void gate_rows_fwd_kernel(const T* __restrict__ visn, const T* __restrict__ lang, T* __restrict__ f,
                                                            int Sv, int Sl, int r0, int n, int H) {
  const int b = blockIdx.x, j = blockIdx.y, c = threadIdx.x * 4;
  if (c >= H) return;
  const f32x4 v = DT<T>::ld4(visn + ((long)b * Sv + r0 + j) * H + c), g = DT<T>::ld4(lang + (long)b * Sl * H + c);
  DT<T>::st4(f + ((long)b * n + j) * H + c, v * g);
}
template <typename T>
__global__ __launch_bounds__(256) void gate_rows_bwd_kernel(const T* __restrict__ df, const T* __restrict__ visn, const T* __restrict__ lang,
                                                            T* __restrict__ dvisn, T* __restrict__ dlang, int Sv, int Sl, int r0, int n, int H) {
  const int b = blockIdx.x, y = blockIdx.y, c = threadIdx.x * 4;
  if (c >= H) return;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (y < Sv) {                                   // one row of dvisn
    if (dvisn == nullptr) return;
    f32x4 o = zero;
    if (y >= r0 && y < r0 + n) o = DT<T>::ld4(df + ((long)b * n + (y - r0)) * H + c) * DT<T>::ld4(lang + (long)b * Sl * H + c);
    DT<T>::st4(dvisn + ((long)b * Sv + y) * H + c, o);
  } else if (dlang != nullptr) {                  // one row of dlang
    const int row = y - Sv;
    f32x4 a0 = zero, a1 = zero;
    if (row == 0) {
      const T* dp = df + (long)b * n * H + c;
      const T* vp = visn + ((long)b * Sv + r0) * H + c;
      int j = 0;
      for (; j + 1 < n; j += 2) {                 // two rows in flight
        a0 += DT<T>::ld4(dp + (long)j * H) * DT<T>::ld4(vp + (long)j * H);
        a1 += DT<T>::ld4(dp + (long)(j + 1) * H) * DT<T>::ld4(vp + (long)(j + 1) * H);
      }
      if (j < n) a0 += DT<T>::ld4(dp + (long)j * H) * DT<T>::ld4(vp + (long)j * H);
    }
    DT<T>::st4(dlang + ((long)b * Sl + row) * H + c, a0 + a1);
  }
}

// logits[r] = mask[r] ? -inf : (bias + sum_c h[r][c] w[c])      (NextActionPrediction.net.4 + masked_fill)
template <typename T>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const T* __restrict__ h, long ldh, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const unsigned char* __restrict__ mask,
                                                         float* __restrict__ out, int rows, int H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    float a = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
      const f32x4 v = DT<T>::ld4(h + (long)r * ldh + c), ww = *(const f32x4*)(w + c);
      a += v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3];
    }
    a = wave_sum(a);
    if (lane == 0) out[r] = (mask && mask[r]) ? -INFINITY : a + (bias ? bias[0] : 0.f);
  }
}
// dh[r][c] = dl[r] * w[c] (0 where masked); dw[c] += sum_r dl[r] h[r][c]; dbias += sum_r dl[r]
template <typename T>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dl, const T* __restrict__ h, long ldh,
                                                         const float* __restrict__ w, const unsigned char* __restrict__ mask,
                                                         T* __restrict__ dh, long lddh, float* __restrict__ dw,
                                                         float* __restrict__ dbias, int rows, int H) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const float wc = c < H ? w[c] : 0.f;
  float aw = 0.f, ab = 0.f;
  int r = blockIdx.y;
  for (; r < rows; r += gridDim.y) {
    const float g = (mask && mask[r]) ? 0.f : dl[r];
    if (c < H) {
      DT<T>::st(dh + (long)r * lddh + c, g * wc);
      aw += g * DT<T>::ld(h + (long)r * ldh + c);
    }
    ab += g;
  }
  if (c < H) atomicAdd(dw + c, aw);
  if (c == 0 && dbias) atomicAdd(dbias, ab);
}

// Masked cross-entropy, reduction = sum, ignore_index: one wave per row.
// loss += sum_r [t_r != ignore] (logsumexp(l_r) - l_r[t_r]);  dlogits[r] = softmax(l_r) - onehot(t_r) (0 if ignored)
__global__ __launch_bounds__(64) void ce_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ target,
                                                long ignore, float* __restrict__ loss, float* __restrict__ dlogits, long lddl,
                                                int rows, int V) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const long t = target[r];
  if (t == ignore) {
    for (int c = lane; c < V; c += 64) dlogits[(long)r * lddl + c] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int c = lane; c < V; c += 64) m = fmaxf(m, logits[(long)r * ld + c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += __expf(logits[(long)r * ld + c] - m);
  s = wave_sum(s);
  const float lse = m + __logf(s);
  for (int c = lane; c < V; c += 64) {
    const float p = __expf(logits[(long)r * ld + c] - lse);
    dlogits[(long)r * lddl + c] = p - (c == t ? 1.f : 0.f);
  }
  if (lane == 0) atomicAdd(loss, lse - logits[(long)r * ld + t]);
}

// Noun-phrase segment mean (aux head): out[s][:] = mean over tok in [off[s], off[s+1]) of x[rowidx[tok]][:]
template <typename T>
__global__ __launch_bounds__(256) void segmean_fwd_kernel(const T* __restrict__ x, long ldx, const int* __restrict__ off,
                                                          const int* __restrict__ rowidx, T* __restrict__ out, int H) {
  const int s = blockIdx.x, a = off[s], b = off[s + 1];
  const float inv = b > a ? 1.0f / (float)(b - a) : 0.f;       // an empty segment (padding of a fixed-capacity plan) is a zero row
  for (int c = threadIdx.x; c < H; c += 256) {
    float acc = 0.f;
    for (int t = a; t < b; ++t) acc += DT<T>::ld(x + (long)rowidx[t] * ldx + c);
    DT<T>::st(out + (long)s * H + c, acc * inv);
  }
}
// dx32[rowidx[tok]][:] += dout[s][:] / len_s      (f32 scratch, atomics: a token may sit in several phrases)
template <typename T>
__global__ __launch_bounds__(256) void segmean_bwd_kernel(const T* __restrict__ dout, const int* __restrict__ off,
                                                          const int* __restrict__ rowidx, float* __restrict__ dx32, int H) {
  const int s = blockIdx.x, a = off[s], b = off[s + 1];
  const float inv = b > a ? 1.0f / (float)(b - a) : 0.f;
  for (int c = threadIdx.x; c < H; c += 256) {
    const float g = DT<T>::ld(dout + (long)s * H + c) * inv;
    for (int t = a; t < b; ++t) atomicAdd(dx32 + (long)rowidx[t] * H + c, g);
  }
}

// cos[r] = <x/max(|x|,eps), y/max(|y|,eps)>   (torch.nn.functional.cosine_similarity, eps 1e-8), one wave per row.
// Saves nx, ny for backward.
template <typename T>
__global__ __launch_bounds__(64) void cosine_fwd_kernel(const T* __restrict__ x, const T* __restrict__ y, float eps,
                                                        float* __restrict__ cosv, float* __restrict__ nx, float* __restrict__ ny,
                                                        int H) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float xy = 0.f, xx = 0.f, yy = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float a = DT<T>::ld(x + (long)r * H + c), b = DT<T>::ld(y + (long)r * H + c);
    xy += a * b; xx += a * a; yy += b * b;
  }
  xy = wave_sum(xy); xx = wave_sum(xx); yy = wave_sum(yy);
  const float nxx = fmaxf(sqrtf(xx), eps), nyy = fmaxf(sqrtf(yy), eps);
  if (lane == 0) { cosv[r] = xy / (nxx * nyy); nx[r] = nxx; ny[r] = nyy; }
}
// dx = g * (y/(nx ny) - cos * x / nx^2),  dy symmetric   (norms above eps)
template <typename T>
__global__ __launch_bounds__(64) void cosine_bwd_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                        const float* __restrict__ gcos, const float* __restrict__ cosv,
                                                        const float* __restrict__ nx, const float* __restrict__ ny,
                                                        T* __restrict__ dx, T* __restrict__ dy, int H) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float g = gcos[r], cv = cosv[r], a = nx[r], b = ny[r];
  for (int c = lane; c < H; c += 64) {
    const float xv = DT<T>::ld(x + (long)r * H + c), yv = DT<T>::ld(y + (long)r * H + c);
    if (dx) DT<T>::st(dx + (long)r * H + c, g * (yv / (a * b) - cv * xv / (a * a)));
    if (dy) DT<T>::st(dy + (long)r * H + c, g * (xv / (a * b) - cv * yv / (b * b)));
  }
}

// dz = da * act'(z)   (act: 1 gelu-erf, 2 relu); used where the activation derivative cannot ride a GEMM epilogue
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(int act, const T* __restrict__ da, const T* __restrict__ z,
                                                      T* __restrict__ dz, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    const f32x4 g = DT<T>::ld4(da + i), zz = DT<T>::ld4(z + i);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] * (act == 1 ? gelu_erf_grad(zz[j]) : (zz[j] > 0.f ? 1.f : 0.f));
    DT<T>::st4(dz + i, o);
  }
}

// y = x * keep(seed, idx)/(1-p)  (x == null: writes the mask itself, for tests); idx = linear element index
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n, unsigned thr,
                                                      unsigned seed0, float inv, const unsigned* sbase) {
  const unsigned seed = eff_seed(seed0, sbase);
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float m = drop_scale((unsigned)i, seed, thr, inv);
    DT<T>::st(y + i, x ? DT<T>::ld(x + i) * m : m);
  }
}

// Fused AdamW over a flat parameter arena (torch.optim.AdamW semantics, decoupled weight decay), plus
// global-norm clipping folded in: g <- g * clip_coef[0].   Optionally refreshes a bf16 shadow of p.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, __bf16* __restrict__ shadow, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    const float* __restrict__ clip_coef) {
  const float cc = clip_coef ? clip_coef[0] : 1.f;
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    f32x4 pv = *(f32x4*)(p + i), gv = *(const f32x4*)(g + i), mv = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv[j] * cc;
      pv[j] *= (1.f - lr * wd);
      mv[j] = b1 * mv[j] + (1.f - b1) * gg;
      vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv[j]) / sqrtf(bc2) + eps;
      pv[j] -= (lr / bc1) * mv[j] / denom;
    }
    *(f32x4*)(p + i) = pv; *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv;
    if (shadow) {
      bf16x4 s = {(__bf16)pv[0], (__bf16)pv[1], (__bf16)pv[2], (__bf16)pv[3]};
      *(bf16x4*)(shadow + i) = s;
    }
  }
}

// sumsq[0] += sum g^2
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float a = 0.f;
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 * stride < n; i += 4 * stride) {              // four independent 16-byte loads in flight per lane
    const f32x4 v0 = *(const f32x4*)(g + i), v1 = *(const f32x4*)(g + i + stride), v2 = *(const f32x4*)(g + i + 2 * stride),
                v3 = *(const f32x4*)(g + i + 3 * stride);
    a += (v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3]) + (v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3]) +
         (v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2] + v2[3] * v2[3]) + (v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2] + v3[3] * v3[3]);
  }
  for (; i < n; i += stride) {
    const f32x4 v = *(const f32x4*)(g + i);
    a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
// coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))      (torch.nn.utils.clip_grad_norm_)
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float* coef) {
  coef[0] = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
}

// Device-resident optimizer state so that a captured hipGraph replays a CORRECT step: state = [clip coef, 1-b1^t, 1-b2^t, t].
__global__ void optim_prepare_kernel(const float* sumsq, float max_norm, float b1, float b2, float* state) {
  const float t = state[3] + 1.f;
  state[3] = t;
  state[0] = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
  state[1] = 1.f - powf(b1, t);
  state[2] = 1.f - powf(b2, t);
}

__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, __bf16* __restrict__ shadow, long n,
                                                        const float* __restrict__ lr_dev, float b1, float b2, float eps, float wd,
                                                        const float* __restrict__ state) {
  const float cc = state[0], bc1 = state[1], rbc2 = 1.f / sqrtf(state[2]), lr = lr_dev[0];
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    f32x4 pv = *(f32x4*)(p + i), gv = *(const f32x4*)(g + i), mv = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv[j] * cc;
      pv[j] *= (1.f - lr * wd);
      mv[j] = b1 * mv[j] + (1.f - b1) * gg;
      vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(vv[j]) * rbc2 + eps;
      pv[j] -= (lr / bc1) * mv[j] / denom;
    }
    *(f32x4*)(p + i) = pv; *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv;
    if (shadow) {
      bf16x4 s = {(__bf16)pv[0], (__bf16)pv[1], (__bf16)pv[2], (__bf16)pv[3]};
      *(bf16x4*)(shadow + i) = s;
    }
  }
}

// ---- parameter groups (the optimizer the shipped "variant4" warm-up builds: VLN-HAMT/finetune_src/r2r/agent_cmt.py:82-96,
// r2r/main.py:202-255): the arena is laid out group by group, grp_end[k] = first element past group k; per group the learning
// rate (grp_lr[k]) and a trainable flag (grp_lr[ngrp + k]) live in device memory, so a stage change is a small device write and a
// captured step replays it. A group that is off is skipped altogether (torch.optim skips parameters without a gradient: no weight
// decay, no moment update, no step count). gstate[4k..4k+2] = (t, 1 - beta1^t, 1 - beta2^t) of group k.
// state (8 floats): [0] clip factor x 1/loss-scale, [3] calls so far, [4] loss scale S (1 unless set), [5] 1 = step skipped
// (non-finite gradients under dynamic loss scaling), [6] good steps since the last change of S, [7] unscaled gradient norm.
__global__ void optim_prepare_groups_kernel(const float* sumsq, float max_norm, float b1, float b2, float* state, float* gstate,
                                            const float* grp_lr, int ngrp, int growth_interval) {
  const int k = threadIdx.x;
  const float S = state[4] > 0.f ? state[4] : 1.f;
  const float ss = sumsq[0];
  const bool finite = ss == ss && ss < 3.0e38f;
  const bool skip = growth_interval > 0 && !finite;             // GradScaler semantics: drop the step, halve the scale
  if (k < ngrp && !skip && grp_lr[ngrp + k] != 0.f) {
    const float t = gstate[4 * k] + 1.f;
    gstate[4 * k] = t;
    gstate[4 * k + 1] = 1.f - powf(b1, t);
    gstate[4 * k + 2] = 1.f - powf(b2, t);
  }
  if (k == 0) {
    const float norm = sqrtf(ss) / S;
    state[3] += 1.f;
    state[7] = norm;
    state[5] = skip ? 1.f : 0.f;
    state[0] = skip ? 0.f : fminf(1.f, max_norm / (norm + 1e-6f)) / S;
    if (growth_interval > 0) {
      if (skip) { state[4] = fmaxf(S * 0.5f, 1.f); state[6] = 0.f; }
      else if (state[6] + 1.f >= (float)growth_interval) { state[4] = fminf(S * 2.f, 16777216.f); state[6] = 0.f; }
      else state[6] += 1.f;
    }
  }
}

template <typename SH>        // SH: type of the compute-dtype mirror of the parameters (__bf16 or _Float16)
__global__ __launch_bounds__(256) void adamw_groups_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, SH* __restrict__ shadow, long n,
                                                           const long* __restrict__ grp_end, const float* __restrict__ grp_lr,
                                                           const float* __restrict__ gstate, int ngrp, float b1, float b2,
                                                           float eps, float wd, const float* __restrict__ state) {
  if (state[5] != 0.f) return;                                   // skipped step: parameters, moments and mirror stay
  const float cc = state[0];
  // Two float4 groups per thread and iteration, every load of both issued before the first use (8 x 16 B per lane in flight), streaming
  // (non-temporal) accesses: the four arrays are 4 x 430 MB per step and nothing is re-read before the next step (the mirror is: plain store).
  const long T = (long)gridDim.x * blockDim.x * 4;
  long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (; i0 < n; i0 += 2 * T) {
    long idx[2] = {i0, i0 + T};
    bool on[2];
    float lr[2], bc1[2], rbc2[2];
    f32x4 pv[2], gv[2], mv[2], vv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = idx[u];
      on[u] = i < n;
      if (!on[u]) continue;
      int k = 0;
      while (k + 1 < ngrp && i >= grp_end[k]) ++k;               // groups start on 8-element boundaries: a float4 never straddles
      on[u] = grp_lr[ngrp + k] != 0.f;
      if (!on[u]) continue;
      lr[u] = grp_lr[k]; bc1[u] = gstate[4 * k + 1]; rbc2[u] = 1.f / sqrtf(gstate[4 * k + 2]);
      pv[u] = __builtin_nontemporal_load((const f32x4*)(p + i));
      gv[u] = __builtin_nontemporal_load((const f32x4*)(g + i));
      mv[u] = __builtin_nontemporal_load((const f32x4*)(m + i));
      vv[u] = __builtin_nontemporal_load((const f32x4*)(v + i));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!on[u]) continue;
      const long i = idx[u];
      // An element that has never received a gradient (g, m and v all exactly zero: a parameter behind a detach() - the embeddings under
      // --fix_lang_embedding - or an unused head) is left alone, weight decay included: torch.optim.AdamW skips parameters whose
      // .grad is None the same way (r2r/agent_cmt.py:98 builds it over ALL parameters). A float4 of such elements is not even stored.
      bool any = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = gv[u][j] != 0.f || mv[u][j] != 0.f || vv[u][j] != 0.f;
        any |= live;
        const float gg = gv[u][j] * cc;
        const float p0 = pv[u][j] * (1.f - lr[u] * wd);
        const float m1 = b1 * mv[u][j] + (1.f - b1) * gg;
        const float v1 = b2 * vv[u][j] + (1.f - b2) * gg * gg;
        const float denom = sqrtf(v1) * rbc2[u] + eps;
        pv[u][j] = live ? p0 - (lr[u] / bc1[u]) * m1 / denom : pv[u][j];
        mv[u][j] = m1;                                        // (0 where nothing was ever added)
        vv[u][j] = v1;
      }
      if (!any) continue;
      __builtin_nontemporal_store(pv[u], (f32x4*)(p + i));
      __builtin_nontemporal_store(mv[u], (f32x4*)(m + i));
      __builtin_nontemporal_store(vv[u], (f32x4*)(v + i));
      if (shadow) {
        typedef SH sh4 __attribute__((ext_vector_type(4)));
        sh4 s = {(SH)pv[u][0], (SH)pv[u][1], (SH)pv[u][2], (SH)pv[u][3]};
        *(sh4*)(shadow + i) = s;
      }
    }
  }
}

// dst = (OUT)(src * scale): the gradient all-reduce payload (float32 arena -> bf16 with the 1/world pre-division, and back)
template <typename IN, typename OUT>
__global__ __launch_bounds__(256) void scale_cast_kernel(const IN* __restrict__ src, OUT* __restrict__ dst, long n, float scale) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    f32x4 x = DT<IN>::ld4(src + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] *= scale;
    DT<OUT>::st4(dst + i, x);
  }
  if (i < n && i + 3 >= n)
    for (long j = i; j < n; ++j) DT<OUT>::st(dst + j, DT<IN>::ld(src + j) * scale);
}

// DUET global/local logit fusion (VLN-DUET/map_nav_src/models/vilmodel.py:1198-1217) as one gather / one scatter:
//   fused[b,0] = gl[b,0] + ll[b,0];  fused[b,g] = gl[b,g] + (src[b,g] >= 0 ? ll[b,src] : src == -2 ? sum_{j: bw[b,j]} ll[b,j] : 0)
// src / bw come from the host-side plan (which map node is which local candidate; which candidates are visited).
__global__ __launch_bounds__(64) void duet_fuse_fwd_kernel(const float* __restrict__ gl, const float* __restrict__ ll,
                                                           const int* __restrict__ src, const unsigned char* __restrict__ bw,
                                                           float* __restrict__ out, int G, int V) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int j = lane; j < V; j += 64)
    if (bw[b * V + j]) s += ll[b * V + j];
  s = wave_sum(s);
  for (int g = lane; g < G; g += 64) {
    const int c = src[b * G + g];
    float v = gl[b * G + g];
    if (g == 0) v += ll[b * V];
    else if (c >= 0) v += ll[b * V + c];
    else if (c == -2) v += s;
    out[b * G + g] = v;
  }
}
__global__ __launch_bounds__(64) void duet_fuse_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ src,
                                                           const unsigned char* __restrict__ bw, float* __restrict__ dll, int G,
                                                           int V) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float sb = 0.f;
  for (int g = lane; g < G; g += 64)
    if (g > 0 && src[b * G + g] == -2) sb += dout[b * G + g];
  sb = wave_sum(sb);
  for (int j = lane; j < V; j += 64) {
    float d = (j == 0) ? dout[b * G] : 0.f;
    for (int g = 1; g < G; ++g)
      if (src[b * G + g] == j) d += dout[b * G + g];
    if (bw[b * V + j]) d += sb;
    dll[b * V + j] = d;
  }
}

// The whole tail of VLN-DUET/map_nav_src/models/vilmodel.py:1185-1217 in one launch per direction (one wave per sample):
//   w = sigmoid(f[b]) (f == nullptr: 0.5)          global[g] = (visited[g] | !gmask[g]) ? -inf : graw[g] * w
//   local[j] = nav[j] ? lraw[j] * (1 - w) : -inf    fused = duet_fuse(global, local)   (kernel above)
// Replaces sigmoid, two scalings, 1 - w, two mask expressions, two masked_fills and the fuse launch (12 launches forward, as many backward).
__device__ __forceinline__ float duet_w(const float* f, int b) { return f ? 1.f / (1.f + __expf(-f[b])) : 0.5f; }
__global__ __launch_bounds__(64) void duet_heads_fwd_kernel(const float* __restrict__ graw, const float* __restrict__ lraw,
                                                            const float* __restrict__ f, const unsigned char* __restrict__ visited,
                                                            const unsigned char* __restrict__ gmask, const unsigned char* __restrict__ nav,
                                                            const int* __restrict__ src, const unsigned char* __restrict__ bw,
                                                            float* __restrict__ gl, float* __restrict__ ll, float* __restrict__ fused,
                                                            int G, int V) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float w = duet_w(f, b), ninf = -__builtin_inff();
  auto local = [&](int j) { return nav[b * V + j] ? lraw[b * V + j] * (1.f - w) : ninf; };
  float s = 0.f;
  for (int j = lane; j < V; j += 64) {
    const float v = local(j);
    ll[b * V + j] = v;
    if (bw[b * V + j]) s += v;
  }
  s = wave_sum(s);
  for (int g = lane; g < G; g += 64) {
    const float v = (visited[b * G + g] || !gmask[b * G + g]) ? ninf : graw[b * G + g] * w;
    gl[b * G + g] = v;
    const int c = src[b * G + g];
    float o = v;
    if (g == 0) o += local(0);
    else if (c >= 0) o += local(c);
    else if (c == -2) o += s;
    fused[b * G + g] = o;
  }
}
// d_gl / d_ll / d_fused may each be null (output unused). dgraw, dlraw always written; df when f is given.
__global__ __launch_bounds__(64) void duet_heads_bwd_kernel(const float* __restrict__ d_gl, const float* __restrict__ d_ll,
                                                            const float* __restrict__ d_fu, const float* __restrict__ graw,
                                                            const float* __restrict__ lraw, const float* __restrict__ f,
                                                            const unsigned char* __restrict__ visited, const unsigned char* __restrict__ gmask,
                                                            const unsigned char* __restrict__ nav, const int* __restrict__ src,
                                                            const unsigned char* __restrict__ bw, float* __restrict__ dgraw,
                                                            float* __restrict__ dlraw, float* __restrict__ df, int G, int V) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float w = duet_w(f, b);
  float sb = 0.f, dw = 0.f;
  if (d_fu)
    for (int g = lane; g < G; g += 64)
      if (g > 0 && src[b * G + g] == -2) sb += d_fu[b * G + g];
  sb = wave_sum(sb);
  for (int g = lane; g < G; g += 64) {
    const bool dead = visited[b * G + g] || !gmask[b * G + g];
    float d = 0.f;
    if (!dead) {
      d = (d_gl ? d_gl[b * G + g] : 0.f) + (d_fu ? d_fu[b * G + g] : 0.f);
      dw += d * graw[b * G + g];
    }
    dgraw[b * G + g] = d * w;
  }
  for (int j = lane; j < V; j += 64) {
    float d = 0.f;
    if (nav[b * V + j]) {
      d = d_ll ? d_ll[b * V + j] : 0.f;
      if (d_fu) {
        if (j == 0) d += d_fu[b * G];
        for (int g = 1; g < G; ++g)
          if (src[b * G + g] == j) d += d_fu[b * G + g];
        if (bw[b * V + j]) d += sb;
      }
      dw -= d * lraw[b * V + j];
    }
    dlraw[b * V + j] = d * (1.f - w);
  }
  dw = wave_sum(dw);
  if (df && lane == 0) df[b] = dw * w * (1.f - w);
}

// Observation / panorama tensors assembled ON THE DEVICE from a resident feature table (replaces the numpy
// concatenations + 7 MB host->device copy per step of VLN-HAMT/finetune_src/r2r/agent_cmt.py:130-176 and
// VLN-DUET/map_nav_src/r2r/agent.py:67-97). One block per output slot (b, v):
//   view[b,v] >= 0: image row = table[vp_row[b]][view[b,v]][:]; angle = (cand_he given ? angle_feature(h, e) : angle_table[base_view[b]][view])
//   view[b,v] <  0: zeros ([STOP] slot and padding)
template <typename S>
__global__ __launch_bounds__(256) void build_views_kernel(const S* __restrict__ table, const long* __restrict__ vp_row,
                                                          const int* __restrict__ view, const float* __restrict__ cand_he,
                                                          const unsigned char* __restrict__ is_cand, const int* __restrict__ base_view,
                                                          const float* __restrict__ angle_table, float* __restrict__ out_img,
                                                          float* __restrict__ out_ang, int V, int D, int A) {
  const int slot = blockIdx.x, b = slot / V;
  const int vw = view[slot];
  float* oi = out_img + (long)slot * D;
  if (vw < 0) {
    for (int c = threadIdx.x; c < D; c += 256) oi[c] = 0.f;
    if (threadIdx.x < A) out_ang[(long)slot * A + threadIdx.x] = 0.f;
    return;
  }
  const S* src = table + ((long)vp_row[b] * 36 + vw) * D;
  for (int c = threadIdx.x; c < D; c += 256) oi[c] = DT<S>::ld(src + c);
  if (threadIdx.x < A) {
    float v;
    if (is_cand[slot]) {
      const float hd = cand_he[2 * slot], el = cand_he[2 * slot + 1];
      const int q = threadIdx.x & 3;                     // [sin h, cos h, sin e, cos e] repeated A/4 times
      v = q == 0 ? sinf(hd) : q == 1 ? cosf(hd) : q == 2 ? sinf(el) : cosf(el);
    } else {
      v = angle_table[((long)base_view[b] * 36 + vw) * A + threadIdx.x];
    }
    out_ang[(long)slot * A + threadIdx.x] = v;
  }
}

}  // namespace

#define BY_DTYPE(dtype, CALL_F32, CALL_BF16, CALL_F16)  \
  do {                                              \
    if ((dtype) == VLNI_F32) { CALL_F32; }          \
    else if ((dtype) == VLNI_BF16) { CALL_BF16; }   \
    else if ((dtype) == VLNI_F16) { CALL_F16; }     \
    else { vlni_set_error("bad dtype %d", (dtype)); return VLNI_EINVAL; } \
  } while (0)

extern "C" int vlni_cast(int src_dtype, int dst_dtype, const void* src, void* dst, long n, void* stream) {
  VLNI_CHECK(n > 0, VLNI_EINVAL, "cast: n=%ld", n);
  dim3 grid((unsigned)std::min<long>(2048, (n / 4 + 255) / 256 + 1)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == VLNI_F32 && dst_dtype == VLNI_BF16)
    hipLaunchKernelGGL((cast_kernel<float, __bf16>), grid, block, 0, st, (const float*)src, (__bf16*)dst, n);
  else if (src_dtype == VLNI_BF16 && dst_dtype == VLNI_F32)
    hipLaunchKernelGGL((cast_kernel<__bf16, float>), grid, block, 0, st, (const __bf16*)src, (float*)dst, n);
  else if (src_dtype == VLNI_F32 && dst_dtype == VLNI_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, st, (const float*)src, (float*)dst, n);
  else if (src_dtype == VLNI_BF16 && dst_dtype == VLNI_BF16)
    hipLaunchKernelGGL((cast_kernel<__bf16, __bf16>), grid, block, 0, st, (const __bf16*)src, (__bf16*)dst, n);
  else if (src_dtype == VLNI_F32 && dst_dtype == VLNI_F16)
    hipLaunchKernelGGL((cast_kernel<float, _Float16>), grid, block, 0, st, (const float*)src, (_Float16*)dst, n);
  else if (src_dtype == VLNI_F16 && dst_dtype == VLNI_F32)
    hipLaunchKernelGGL((cast_kernel<_Float16, float>), grid, block, 0, st, (const _Float16*)src, (float*)dst, n);
  else if (src_dtype == VLNI_F16 && dst_dtype == VLNI_F16)
    hipLaunchKernelGGL((cast_kernel<_Float16, _Float16>), grid, block, 0, st, (const _Float16*)src, (_Float16*)dst, n);
  else { vlni_set_error("cast: bad dtypes %d %d", src_dtype, dst_dtype); return VLNI_EINVAL; }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_transpose(int src_dtype, int dst_dtype, const void* src, long lds_, void* dst, long ldd, int R, int C,
                              int Rpad, void* stream) {
  VLNI_CHECK(R > 0 && C > 0 && Rpad >= R && ldd >= Rpad && lds_ >= C, VLNI_EINVAL, "transpose: R=%d C=%d Rpad=%d", R, C, Rpad);
  dim3 grid(cdiv(C, 64), cdiv(Rpad, 64)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == VLNI_F32 && dst_dtype == VLNI_BF16)
    hipLaunchKernelGGL((transpose_kernel<float, __bf16>), grid, block, 0, st, (const float*)src, lds_, (__bf16*)dst, ldd, R, C, Rpad);
  else if (src_dtype == VLNI_BF16 && dst_dtype == VLNI_BF16)
    hipLaunchKernelGGL((transpose_kernel<__bf16, __bf16>), grid, block, 0, st, (const __bf16*)src, lds_, (__bf16*)dst, ldd, R, C, Rpad);
  else if (src_dtype == VLNI_F32 && dst_dtype == VLNI_F16)
    hipLaunchKernelGGL((transpose_kernel<float, _Float16>), grid, block, 0, st, (const float*)src, lds_, (_Float16*)dst, ldd, R, C, Rpad);
  else if (src_dtype == VLNI_F16 && dst_dtype == VLNI_F16)
    hipLaunchKernelGGL((transpose_kernel<_Float16, _Float16>), grid, block, 0, st, (const _Float16*)src, lds_, (_Float16*)dst, ldd, R, C, Rpad);
  else if (src_dtype == VLNI_F32 && dst_dtype == VLNI_F32)
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, block, 0, st, (const float*)src, lds_, (float*)dst, ldd, R, C, Rpad);
  else { vlni_set_error("transpose: bad dtypes %d %d", src_dtype, dst_dtype); return VLNI_EINVAL; }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// table: n entries of 56 bytes in DEVICE memory {src*, dst*, long lds, long ldd, int R, C, Rpad, tile0, tiles_c, 0}, tile0 =
// running sum of ceil(C/64)*ceil(Rpad/64), entries sorted by tile0; total_tiles = the grand total. dst[c][r] = src[r][c] (bf16 out).
extern "C" int vlni_transpose_batched(int src_dtype, const void* table_dev, int n, int total_tiles, void* stream) {
  VLNI_CHECK(table_dev && n > 0 && total_tiles > 0, VLNI_EINVAL, "transpose_batched: n=%d tiles=%d", n, total_tiles);
  const TrEntry* tab = (const TrEntry*)table_dev;
  if (src_dtype == VLNI_F32) hipLaunchKernelGGL((transpose_batched_kernel<float>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, tab, n);
  else if (src_dtype == VLNI_BF16) hipLaunchKernelGGL((transpose_batched_kernel<__bf16>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, tab, n);
  else if (src_dtype == VLNI_F16) hipLaunchKernelGGL((transpose_batched_kernel<_Float16>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, tab, n);
  else { vlni_set_error("transpose_batched: bad dtype %d", src_dtype); return VLNI_EINVAL; }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_colsum(int dtype, const void* x, long ldx, int rows, int N, float* out, void* stream) {
  VLNI_CHECK(rows > 0 && N > 0, VLNI_EINVAL, "colsum: rows=%d N=%d", rows, N);
  dim3 grid(cdiv(N, 256), std::max(1, std::min(rows / 32, 256))), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, ldx, rows, N, out),
           hipLaunchKernelGGL((colsum_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, (const __bf16*)x, ldx, rows, N, out),
          hipLaunchKernelGGL((colsum_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, (const _Float16*)x, ldx, rows, N, out));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_smallk_linear_fwd(int dtype, const float* x, long ldx, const float* W, const float* b, void* y, long ldy,
                                      int rows, int N, int K, void* stream) {
  VLNI_CHECK(K >= 1 && K <= 16 && rows > 0 && N > 0, VLNI_EINVAL, "smallk_fwd: rows=%d N=%d K=%d", rows, N, K);
  dim3 grid(cdiv(N, 256), std::min(rows, 1024)), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((smallk_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, x, ldx, W, b, (float*)y, ldy, rows, N, K),
           hipLaunchKernelGGL((smallk_fwd_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, x, ldx, W, b, (__bf16*)y, ldy, rows, N, K),
          hipLaunchKernelGGL((smallk_fwd_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, x, ldx, W, b, (_Float16*)y, ldy, rows, N, K));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_smallk_linear_bwd(int dtype, const void* dy, long lddy, const float* x, long ldx, float* dW, float* db,
                                      int rows, int N, int K, void* stream) {
  VLNI_CHECK(K >= 1 && K <= 16 && rows > 0 && N > 0, VLNI_EINVAL, "smallk_bwd: rows=%d N=%d K=%d", rows, N, K);
  static const int rpb_env = getenv("VLNI_SMALLK_RPB") ? atoi(getenv("VLNI_SMALLK_RPB")) : 0;
  const int rpb = std::min(256, std::max(16, rpb_env > 0 ? rpb_env : rows <= 4096 ? 64 : 128));   // <= 256 rows of x in LDS (probe: tools/scratch/smallk_probe.py)
  dim3 grid(cdiv(N, 64), cdiv(rows, rpb)), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((smallk_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)dy, lddy, x, ldx, dW, db, rows, N, K, rpb),
           hipLaunchKernelGGL((smallk_bwd_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, (const __bf16*)dy, lddy, x, ldx, dW, db, rows, N, K, rpb),
          hipLaunchKernelGGL((smallk_bwd_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, (const _Float16*)dy, lddy, x, ldx, dW, db, rows, N, K, rpb));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_scatter_add_rows(int dtype, const void* src, long lds_, const long* idx, float* table_grad, int rows,
                                     int H, int table_rows, void* stream) {
  VLNI_CHECK(rows > 0 && H > 0 && table_rows > 0, VLNI_EINVAL, "scatter_add_rows: rows=%d H=%d table_rows=%d", rows, H, table_rows);
  int* ierr = vlni_index_error_counter();
  dim3 grid(std::min(rows, 2048)), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((scatter_add_rows_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)src, lds_, idx, table_grad, rows, H, table_rows, ierr),
           hipLaunchKernelGGL((scatter_add_rows_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, (const __bf16*)src, lds_, idx, table_grad, rows, H, table_rows, ierr),
          hipLaunchKernelGGL((scatter_add_rows_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, (const _Float16*)src, lds_, idx, table_grad, rows, H, table_rows, ierr));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_scatter_add_rows_small(int dtype, const void* src, long lds_, const long* idx, float* table_grad, int rows,
                                           int H, int table_rows, void* stream) {
  VLNI_CHECK(rows > 0 && H > 0 && table_rows >= 1 && table_rows <= 8, VLNI_EINVAL, "scatter_add_rows_small: rows=%d H=%d table_rows=%d (1..8)",
             rows, H, table_rows);
  VLNI_CHECK(src && idx && table_grad, VLNI_EINVAL, "scatter_add_rows_small: null pointer");
  dim3 grid(cdiv(rows, 32), cdiv(H, 256)), block(256);
  BY_DTYPE(dtype,
           hipLaunchKernelGGL((scatter_add_rows_small_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)src, lds_, idx, table_grad, rows, H, table_rows, vlni_index_error_counter()),
           hipLaunchKernelGGL((scatter_add_rows_small_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, (const __bf16*)src, lds_, idx, table_grad, rows, H, table_rows, vlni_index_error_counter()),
          hipLaunchKernelGGL((scatter_add_rows_small_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, (const _Float16*)src, lds_, idx, table_grad, rows, H, table_rows, vlni_index_error_counter()));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_seqmean_fwd(int dtype, const void* x, void* out, const long* lens, int B, int S, int H, void* stream) {
  VLNI_CHECK(B > 0 && S > 0 && H > 0, VLNI_EINVAL, "seqmean_fwd: %d %d %d", B, S, H);
  BY_DTYPE(dtype, hipLaunchKernelGGL((seqmean_fwd_kernel<float>), dim3(B, cdiv(H, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, lens, B, S, H),
           hipLaunchKernelGGL((seqmean_fwd_kernel<__bf16>), dim3(B, cdiv(H, 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)out, lens, B, S, H),
          hipLaunchKernelGGL((seqmean_fwd_kernel<_Float16>), dim3(B, cdiv(H, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)out, lens, B, S, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_seqmean_bwd(int dtype, const void* dout, void* dx, const long* lens, int B, int S, int H, void* stream) {
  VLNI_CHECK(B > 0 && S > 0 && H > 0, VLNI_EINVAL, "seqmean_bwd: %d %d %d", B, S, H);
  BY_DTYPE(dtype, hipLaunchKernelGGL((seqmean_bwd_kernel<float>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)dout, (float*)dx, lens, B, S, H),
           hipLaunchKernelGGL((seqmean_bwd_kernel<__bf16>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dout, (__bf16*)dx, lens, B, S, H),
          hipLaunchKernelGGL((seqmean_bwd_kernel<_Float16>), dim3(B), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dout, (_Float16*)dx, lens, B, S, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_gate_rows_fwd(int dtype, const void* visn, const void* lang, void* f, int B, int Sv, int Sl, int r0, int n, int H,
                                  void* stream) {
  VLNI_CHECK(B > 0 && n > 0 && r0 >= 0 && r0 + n <= Sv && Sl > 0 && H % 4 == 0 && H <= 1024, VLNI_EINVAL,
             "gate_rows_fwd: B=%d Sv=%d Sl=%d r0=%d n=%d H=%d", B, Sv, Sl, r0, n, H);
  dim3 grid(B, n), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((gate_rows_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)visn, (const float*)lang, (float*)f, Sv, Sl, r0, n, H),
           hipLaunchKernelGGL((gate_rows_fwd_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, (const __bf16*)visn, (const __bf16*)lang, (__bf16*)f, Sv, Sl, r0, n, H),
          hipLaunchKernelGGL((gate_rows_fwd_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, (const _Float16*)visn, (const _Float16*)lang, (_Float16*)f, Sv, Sl, r0, n, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_gate_rows_bwd(int dtype, const void* df, const void* visn, const void* lang, void* dvisn, void* dlang, int B, int Sv,
                                  int Sl, int r0, int n, int H, void* stream) {
  VLNI_CHECK(B > 0 && n > 0 && r0 >= 0 && r0 + n <= Sv && Sl > 0 && H % 4 == 0 && H <= 1024 && (dvisn || dlang), VLNI_EINVAL,
             "gate_rows_bwd: B=%d Sv=%d Sl=%d r0=%d n=%d H=%d", B, Sv, Sl, r0, n, H);
  dim3 grid(B, Sv + Sl), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((gate_rows_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)df, (const float*)visn, (const float*)lang, (float*)dvisn, (float*)dlang, Sv, Sl, r0, n, H),
           hipLaunchKernelGGL((gate_rows_bwd_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, (const __bf16*)df, (const __bf16*)visn, (const __bf16*)lang, (__bf16*)dvisn, (__bf16*)dlang, Sv, Sl, r0, n, H),
          hipLaunchKernelGGL((gate_rows_bwd_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, (const _Float16*)df, (const _Float16*)visn, (const _Float16*)lang, (_Float16*)dvisn, (_Float16*)dlang, Sv, Sl, r0, n, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_rowdot_fwd(int dtype, const void* h, long ldh, const float* w, const float* bias,
                               const unsigned char* mask, float* out, int rows, int H, void* stream) {
  VLNI_CHECK(rows > 0 && H % 4 == 0 && ldh % 4 == 0, VLNI_EINVAL, "rowdot_fwd: rows=%d H=%d", rows, H);
  dim3 grid(std::min(cdiv(rows, 4), 1024)), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((rowdot_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)h, ldh, w, bias, mask, out, rows, H),
           hipLaunchKernelGGL((rowdot_fwd_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, (const __bf16*)h, ldh, w, bias, mask, out, rows, H),
          hipLaunchKernelGGL((rowdot_fwd_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, (const _Float16*)h, ldh, w, bias, mask, out, rows, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_rowdot_bwd(int dtype, const float* dl, const void* h, long ldh, const float* w, const unsigned char* mask,
                               void* dh, long lddh, float* dw, float* dbias, int rows, int H, void* stream) {
  VLNI_CHECK(rows > 0 && H > 0, VLNI_EINVAL, "rowdot_bwd: rows=%d H=%d", rows, H);
  // rows strided over grid.y: each dh row written once. A thread's loop is a chain of dependent loads (one row per ~0.8 us): 512 row groups keep the
  // episode's 14 k candidate rows at ~28 trips per thread (128 groups: 111 trips, 90 us) for 4 x the atomics of the column sums
  static const int gy = [] { const char* e = getenv("VLNI_ROWDOT_BWD_GROUPS"); return e ? atoi(e) : 512; }();
  dim3 grid(cdiv(H, 256), std::max(1, std::min(rows / 4, gy))), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((rowdot_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, dl, (const float*)h, ldh, w, mask, (float*)dh, lddh, dw, dbias, rows, H),
           hipLaunchKernelGGL((rowdot_bwd_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, dl, (const __bf16*)h, ldh, w, mask, (__bf16*)dh, lddh, dw, dbias, rows, H),
          hipLaunchKernelGGL((rowdot_bwd_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, dl, (const _Float16*)h, ldh, w, mask, (_Float16*)dh, lddh, dw, dbias, rows, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_cross_entropy(const float* logits, long ld, const long* target, long ignore_index, float* loss_sum,
                                  float* dlogits, long lddl, int rows, int V, void* stream) {
  VLNI_CHECK(rows > 0 && V > 0, VLNI_EINVAL, "cross_entropy: rows=%d V=%d", rows, V);
  hipLaunchKernelGGL(ce_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, logits, ld, target, ignore_index, loss_sum, dlogits, lddl, rows, V);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_segment_mean_fwd(int dtype, const void* x, long ldx, const int* seg_off, const int* rowidx, void* out,
                                     int nseg, int H, void* stream) {
  VLNI_CHECK(nseg > 0 && H > 0, VLNI_EINVAL, "segment_mean_fwd: nseg=%d H=%d", nseg, H);
  BY_DTYPE(dtype, hipLaunchKernelGGL((segmean_fwd_kernel<float>), dim3(nseg), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, seg_off, rowidx, (float*)out, H),
           hipLaunchKernelGGL((segmean_fwd_kernel<__bf16>), dim3(nseg), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, ldx, seg_off, rowidx, (__bf16*)out, H),
          hipLaunchKernelGGL((segmean_fwd_kernel<_Float16>), dim3(nseg), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, ldx, seg_off, rowidx, (_Float16*)out, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_segment_mean_bwd(int dtype, const void* dout, const int* seg_off, const int* rowidx, float* dx32, int nseg,
                                     int H, void* stream) {
  VLNI_CHECK(nseg > 0 && H > 0, VLNI_EINVAL, "segment_mean_bwd: nseg=%d H=%d", nseg, H);
  BY_DTYPE(dtype, hipLaunchKernelGGL((segmean_bwd_kernel<float>), dim3(nseg), dim3(256), 0, (hipStream_t)stream, (const float*)dout, seg_off, rowidx, dx32, H),
           hipLaunchKernelGGL((segmean_bwd_kernel<__bf16>), dim3(nseg), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dout, seg_off, rowidx, dx32, H),
          hipLaunchKernelGGL((segmean_bwd_kernel<_Float16>), dim3(nseg), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dout, seg_off, rowidx, dx32, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_cosine_fwd(int dtype, const void* x, const void* y, float eps, float* cosv, float* nx, float* ny, int rows,
                               int H, void* stream) {
  VLNI_CHECK(rows > 0 && H > 0, VLNI_EINVAL, "cosine_fwd: rows=%d H=%d", rows, H);
  BY_DTYPE(dtype, hipLaunchKernelGGL((cosine_fwd_kernel<float>), dim3(rows), dim3(64), 0, (hipStream_t)stream, (const float*)x, (const float*)y, eps, cosv, nx, ny, H),
           hipLaunchKernelGGL((cosine_fwd_kernel<__bf16>), dim3(rows), dim3(64), 0, (hipStream_t)stream, (const __bf16*)x, (const __bf16*)y, eps, cosv, nx, ny, H),
          hipLaunchKernelGGL((cosine_fwd_kernel<_Float16>), dim3(rows), dim3(64), 0, (hipStream_t)stream, (const _Float16*)x, (const _Float16*)y, eps, cosv, nx, ny, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_cosine_bwd(int dtype, const void* x, const void* y, const float* gcos, const float* cosv, const float* nx,
                               const float* ny, void* dx, void* dy, int rows, int H, void* stream) {
  VLNI_CHECK(rows > 0 && H > 0, VLNI_EINVAL, "cosine_bwd: rows=%d H=%d", rows, H);
  BY_DTYPE(dtype, hipLaunchKernelGGL((cosine_bwd_kernel<float>), dim3(rows), dim3(64), 0, (hipStream_t)stream, (const float*)x, (const float*)y, gcos, cosv, nx, ny, (float*)dx, (float*)dy, H),
           hipLaunchKernelGGL((cosine_bwd_kernel<__bf16>), dim3(rows), dim3(64), 0, (hipStream_t)stream, (const __bf16*)x, (const __bf16*)y, gcos, cosv, nx, ny, (__bf16*)dx, (__bf16*)dy, H),
          hipLaunchKernelGGL((cosine_bwd_kernel<_Float16>), dim3(rows), dim3(64), 0, (hipStream_t)stream, (const _Float16*)x, (const _Float16*)y, gcos, cosv, nx, ny, (_Float16*)dx, (_Float16*)dy, H));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// Pairwise dot products of two small float32 row sets (the in-batch-negatives similarity matrix of the InfoNCE / margin alignment losses,
// vilmodel_cmt.py:793-856: a few hundred projected imaginations x every other sample's noun phrases): out[i, j] = <a_i, b_j>.
// One wave per output; the backward kernels give dA[i] = sum_j g[i, j] b_j and dB[j] = sum_i g[i, j] a_i with one wave per row.
__global__ __launch_bounds__(64) void pairdot_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                         int nb, int H) {
  const int i = blockIdx.y, j = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int c = lane; c < H; c += 64) s += a[(long)i * H + c] * b[(long)j * H + c];
  s = wave_sum(s);
  if (lane == 0) out[(long)i * nb + j] = s;
}
// d_rows[r, :] = sum_q g(r, q) * other[q, :]; g is [na, nb] row-major, `trans` selects g[q, r] (the dB side)
__global__ __launch_bounds__(256) void pairdot_bwd_kernel(const float* __restrict__ g, const float* __restrict__ other, float* __restrict__ d,
                                                          int nr, int nq, int H, int trans) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += 256) {
    float s = 0.f;
    for (int q = 0; q < nq; ++q) s += (trans ? g[(long)q * nr + r] : g[(long)r * nq + q]) * other[(long)q * H + c];
    d[(long)r * H + c] = s;
  }
}
extern "C" int vlni_pairdot_fwd(const float* a, const float* b, float* out, int na, int nb, int H, void* stream) {
  VLNI_CHECK(na > 0 && nb > 0 && H > 0 && na <= 65535, VLNI_EINVAL, "pairdot_fwd: na=%d nb=%d H=%d", na, nb, H);
  hipLaunchKernelGGL(pairdot_fwd_kernel, dim3(nb, na), dim3(64), 0, (hipStream_t)stream, a, b, out, nb, H);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_pairdot_bwd(const float* g, const float* a, const float* b, float* da, float* db, int na, int nb, int H, void* stream) {
  VLNI_CHECK(na > 0 && nb > 0 && H > 0, VLNI_EINVAL, "pairdot_bwd: na=%d nb=%d H=%d", na, nb, H);
  if (da) hipLaunchKernelGGL(pairdot_bwd_kernel, dim3(na), dim3(256), 0, (hipStream_t)stream, g, b, da, na, nb, H, 0);
  if (db) hipLaunchKernelGGL(pairdot_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, a, db, nb, na, H, 1);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_act_bwd(int dtype, int act, const void* da, const void* z, void* dz, long n, void* stream) {
  VLNI_CHECK(n > 0 && n % 4 == 0 && (act == 1 || act == 2), VLNI_EINVAL, "act_bwd: n=%ld act=%d", n, act);
  dim3 grid((unsigned)std::min<long>(2048, (n / 4 + 255) / 256)), block(256);
  BY_DTYPE(dtype, hipLaunchKernelGGL((act_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, act, (const float*)da, (const float*)z, (float*)dz, n),
           hipLaunchKernelGGL((act_bwd_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, act, (const __bf16*)da, (const __bf16*)z, (__bf16*)dz, n),
          hipLaunchKernelGGL((act_bwd_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, act, (const _Float16*)da, (const _Float16*)z, (_Float16*)dz, n));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// y[i] = x[i] * mask(i) / (1-p) with the library's counter-based mask (x NULL: y = the scaled mask). Used where a dropout
// cannot ride a GEMM / LayerNorm / attention kernel, and by the tests to build exact torch references.
extern "C" int vlni_dropout(int dtype, const void* x, void* y, long n, float p, unsigned seed, void* stream) {
  VLNI_CHECK(n > 0 && p >= 0.f && p < 1.f, VLNI_EINVAL, "dropout: n=%ld p=%f", n, p);
  dim3 grid((unsigned)std::min<long>(2048, (n + 255) / 256)), block(256);
  const unsigned thr = drop_thr(p);
  const float inv = 1.0f / (1.0f - p);
  BY_DTYPE(dtype, hipLaunchKernelGGL((dropout_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, (float*)y, n, thr, seed, inv, vlni_seed_base()),
           hipLaunchKernelGGL((dropout_kernel<__bf16>), grid, block, 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y, n, thr, seed, inv, vlni_seed_base()),
          hipLaunchKernelGGL((dropout_kernel<_Float16>), grid, block, 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, n, thr, seed, inv, vlni_seed_base()));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_adamw_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, const float* clip_coef, void* stream) {
  VLNI_CHECK(n > 0 && n % 4 == 0 && step >= 1, VLNI_EINVAL, "adamw_step: n=%ld (multiple of 4) step=%d", n, step);
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  dim3 grid((unsigned)std::min<long>(4096, (n / 4 + 255) / 256)), block(256);
  hipLaunchKernelGGL(adamw_kernel, grid, block, 0, (hipStream_t)stream, p, g, m, v, (__bf16*)bf16_shadow, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, clip_coef);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// out[0] += sum_i slots[32 i]: folds vlni_reduce_parts_sq's spread accumulators into the one the optimizer kernels read
__global__ __launch_bounds__(64) void sumsq_fold_kernel(const float* __restrict__ slots, int nslots, float* __restrict__ out) {
  float a = 0.f;
  for (int i = threadIdx.x; i < nslots; i += 64) a += slots[i * 32];
  a = wave_sum(a);
  if (threadIdx.x == 0) out[0] += a;
}
extern "C" int vlni_sumsq_fold(const float* slots, int nslots, float* out, void* stream) {
  VLNI_CHECK(slots && out && nslots >= 1, VLNI_EINVAL, "sumsq_fold: nslots=%d", nslots);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slots, nslots, out);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
// sumsq[0] += sum(g^2) (caller zeroes sumsq); then vlni_clip_coef turns it into the clip_grad_norm_ factor.
extern "C" int vlni_sumsq(const float* g, long n, float* sumsq, void* stream) {
  VLNI_CHECK(n > 0 && n % 4 == 0, VLNI_EINVAL, "sumsq: n=%ld", n);
  dim3 grid((unsigned)std::min<long>(2048, (n / 4 + 255) / 256)), block(256);
  hipLaunchKernelGGL(sumsq_kernel, grid, block, 0, (hipStream_t)stream, g, n, sumsq);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_build_views(int table_dtype, const void* table, const long* vp_row, const int* view, const float* cand_he,
                                const unsigned char* is_cand, const int* base_view, const float* angle_table, float* out_img,
                                float* out_ang, int B, int V, int D, int A, void* stream) {
  VLNI_CHECK(B > 0 && V > 0 && D > 0 && A > 0 && A <= 256 && A % 4 == 0, VLNI_EINVAL, "build_views: B=%d V=%d D=%d A=%d", B, V, D, A);
  VLNI_CHECK(table && vp_row && view && cand_he && is_cand && base_view && angle_table && out_img && out_ang, VLNI_EINVAL,
             "build_views: null pointer");
  BY_DTYPE(table_dtype,
           hipLaunchKernelGGL((build_views_kernel<float>), dim3(B * V), dim3(256), 0, (hipStream_t)stream, (const float*)table, vp_row, view,
                              cand_he, is_cand, base_view, angle_table, out_img, out_ang, V, D, A),
           hipLaunchKernelGGL((build_views_kernel<__bf16>), dim3(B * V), dim3(256), 0, (hipStream_t)stream, (const __bf16*)table, vp_row, view,
                              cand_he, is_cand, base_view, angle_table, out_img, out_ang, V, D, A),
          hipLaunchKernelGGL((build_views_kernel<_Float16>), dim3(B * V), dim3(256), 0, (hipStream_t)stream, (const _Float16*)table, vp_row, view,
                              cand_he, is_cand, base_view, angle_table, out_img, out_ang, V, D, A));
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_duet_fuse_fwd(const float* gl, const float* ll, const int* src, const unsigned char* bw, float* out, int B,
                                  int G, int V, void* stream) {
  VLNI_CHECK(B > 0 && G > 0 && V > 0 && gl && ll && src && bw && out, VLNI_EINVAL, "duet_fuse_fwd: B=%d G=%d V=%d", B, G, V);
  hipLaunchKernelGGL(duet_fuse_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, gl, ll, src, bw, out, G, V);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_duet_fuse_bwd(const float* dout, const int* src, const unsigned char* bw, float* dll, int B, int G, int V,
                                  void* stream) {
  VLNI_CHECK(B > 0 && G > 0 && V > 0 && dout && src && bw && dll, VLNI_EINVAL, "duet_fuse_bwd: B=%d G=%d V=%d", B, G, V);
  hipLaunchKernelGGL(duet_fuse_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, dout, src, bw, dll, G, V);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_duet_heads_fwd(const float* graw, const float* lraw, const float* f, const unsigned char* visited,
                                   const unsigned char* gmask, const unsigned char* nav, const int* src, const unsigned char* bw,
                                   float* gl, float* ll, float* fused, int B, int G, int V, void* stream) {
  VLNI_CHECK(B > 0 && G > 0 && V > 0 && graw && lraw && visited && gmask && nav && src && bw && gl && ll && fused, VLNI_EINVAL,
             "duet_heads_fwd: B=%d G=%d V=%d", B, G, V);
  hipLaunchKernelGGL(duet_heads_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, graw, lraw, f, visited, gmask, nav, src, bw, gl, ll,
                     fused, G, V);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_duet_heads_bwd(const float* d_gl, const float* d_ll, const float* d_fused, const float* graw, const float* lraw,
                                   const float* f, const unsigned char* visited, const unsigned char* gmask, const unsigned char* nav,
                                   const int* src, const unsigned char* bw, float* dgraw, float* dlraw, float* df, int B, int G, int V,
                                   void* stream) {
  VLNI_CHECK(B > 0 && G > 0 && V > 0 && graw && lraw && visited && gmask && nav && src && bw && dgraw && dlraw && (df || !f), VLNI_EINVAL,
             "duet_heads_bwd: B=%d G=%d V=%d", B, G, V);
  hipLaunchKernelGGL(duet_heads_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, d_gl, d_ll, d_fused, graw, lraw, f, visited, gmask,
                     nav, src, bw, dgraw, dlraw, df, G, V);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// Graph-replayable form of clip + AdamW: the step count, bias corrections and clip factor live in `state` (4 floats on the
// device, zero-initialised by the caller), the learning rate in lr_dev[0]; nothing that changes per step is a launch argument.
extern "C" int vlni_optim_prepare(const float* sumsq, float max_norm, float beta1, float beta2, float* state, void* stream) {
  VLNI_CHECK(sumsq && state, VLNI_EINVAL, "optim_prepare: null pointer");
  hipLaunchKernelGGL(optim_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, beta1, beta2, state);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_adamw_step_dev(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, const float* lr_dev,
                                   float beta1, float beta2, float eps, float weight_decay, const float* state, void* stream) {
  VLNI_CHECK(n > 0 && n % 4 == 0 && lr_dev && state, VLNI_EINVAL, "adamw_step_dev: n=%ld (multiple of 4), lr/state non-null", n);
  dim3 grid((unsigned)std::min<long>(4096, (n / 4 + 255) / 256)), block(256);
  hipLaunchKernelGGL(adamw_dev_kernel, grid, block, 0, (hipStream_t)stream, p, g, m, v, (__bf16*)bf16_shadow, n, lr_dev, beta1,
                     beta2, eps, weight_decay, state);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
// Parameter-group form of the two calls above (see optim_prepare_groups_kernel): ngrp <= 8 contiguous groups of the arena.
// growth_interval > 0 switches dynamic loss scaling on (torch.cuda.amp.GradScaler: VLN-DUET/pretrain_src/train_r2r.py:201-234):
// state[4] holds the scale the caller multiplied the loss by; non-finite gradients skip the step and halve it.
extern "C" int vlni_optim_prepare_groups(const float* sumsq, float max_norm, float beta1, float beta2, float* state, float* gstate,
                                         const float* grp_lr, int ngrp, int growth_interval, void* stream) {
  VLNI_CHECK(sumsq && state && gstate && grp_lr && ngrp >= 1 && ngrp <= 8, VLNI_EINVAL, "optim_prepare_groups: ngrp=%d (1..8)", ngrp);
  hipLaunchKernelGGL(optim_prepare_groups_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq, max_norm, beta1, beta2, state, gstate,
                     grp_lr, ngrp, growth_interval);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_adamw_step_groups(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, long n,
                                      const long* grp_end, const float* grp_lr, const float* gstate, int ngrp, float beta1,
                                      float beta2, float eps, float weight_decay, const float* state, void* stream) {
  VLNI_CHECK(n > 0 && n % 4 == 0 && grp_end && grp_lr && gstate && state && ngrp >= 1 && ngrp <= 8, VLNI_EINVAL,
             "adamw_step_groups: n=%ld (multiple of 4), ngrp=%d (1..8), tables non-null", n, ngrp);
  VLNI_CHECK(shadow == nullptr || shadow_dtype == VLNI_BF16 || shadow_dtype == VLNI_F16, VLNI_EINVAL,
             "adamw_step_groups: mirror dtype %d (bf16 or f16)", shadow_dtype);
  dim3 grid((unsigned)std::min<long>(4096, (n / 4 + 255) / 256)), block(256);
  if (shadow_dtype == VLNI_F16)
    hipLaunchKernelGGL(adamw_groups_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, p, g, m, v, (_Float16*)shadow, n, grp_end,
                       grp_lr, gstate, ngrp, beta1, beta2, eps, weight_decay, state);
  else
    hipLaunchKernelGGL(adamw_groups_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, p, g, m, v, (__bf16*)shadow, n, grp_end,
                       grp_lr, gstate, ngrp, beta1, beta2, eps, weight_decay, state);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
// dst[i] = (out dtype)(src[i] * scale); float32 <-> bfloat16 (the bf16 gradient all-reduce payload, pre-divided by the world size)
extern "C" int vlni_scale_cast(int dt_in, int dt_out, const void* src, void* dst, long n, float scale, void* stream) {
  VLNI_CHECK(n > 0 && src && dst && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, VLNI_EINVAL, "scale_cast: n=%ld / alignment", n);
  dim3 grid((unsigned)std::min<long>(4096, (n / 4 + 255) / 256)), block(256);
  if (dt_in == VLNI_F32 && dt_out == VLNI_BF16)
    hipLaunchKernelGGL((scale_cast_kernel<float, __bf16>), grid, block, 0, (hipStream_t)stream, (const float*)src, (__bf16*)dst, n, scale);
  else if (dt_in == VLNI_BF16 && dt_out == VLNI_F32)
    hipLaunchKernelGGL((scale_cast_kernel<__bf16, float>), grid, block, 0, (hipStream_t)stream, (const __bf16*)src, (float*)dst, n, scale);
  else if (dt_in == VLNI_F32 && dt_out == VLNI_F32)
    hipLaunchKernelGGL((scale_cast_kernel<float, float>), grid, block, 0, (hipStream_t)stream, (const float*)src, (float*)dst, n, scale);
  else VLNI_CHECK(false, VLNI_EUNSUP, "scale_cast: dtypes %d -> %d", dt_in, dt_out);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_clip_coef(const float* sumsq, float max_norm, float* coef, void* stream) {
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, coef);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
