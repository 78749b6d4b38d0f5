// Row LayerNorm forward/backward (+ the multi-source "sum then LayerNorm" used by every embedding
// block), HBM-bound: one 64-lane wave per row, the whole row in registers (16-B/8-B vector loads,
// two-pass variance), wavefront-shuffle reductions, no LDS on the forward path.
// Restates BertLayerNorm = torch.nn.LayerNorm at
//   VLN-HAMT/finetune_src/models/vilmodel_cmt.py:22,71,147,189,536-542,593-616 (eps 1e-12)
//   VLN-DUET/map_nav_src/models/transformer.py:170-182 (eps 1e-5).
#include "common.h"

namespace {

constexpr int WAVES = 4;

template <typename T, int NC>     // H = 256 * NC; blk / nblk: this block's index and the block count of ITS problem (dual launches)
__device__ __forceinline__ void ln_fwd_body(const T* __restrict__ x, long ldx, const float* __restrict__ g, const float* __restrict__ b,
                                            float eps, T* __restrict__ y, long ldy, float* __restrict__ mean, float* __restrict__ rstd,
                                            int rows, int blk, int nblk) {
  constexpr int H = 256 * NC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gv[NC], bv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    gv[c] = *(const f32x4*)(g + c * 256 + lane * 4);
    bv[c] = *(const f32x4*)(b + c * 256 + lane * 4);
  }
  for (int row = blk * WAVES + wave; row < rows; row += nblk * WAVES) {
    f32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      v[c] = DT<T>::ld4(x + (long)row * ldx + c * 256 + lane * 4);
      s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mu = wave_sum(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[c][j] - mu;
        q += d * d;
      }
    const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / H) + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mu) * rs * gv[c][j] + bv[c][j];
      DT<T>::st4(y + (long)row * ldy + c * 256 + lane * 4, o);
    }
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
  }
}
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ g,
                                                     const float* __restrict__ b, float eps, T* __restrict__ y, long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows) {
  ln_fwd_body<T, NC>(x, ldx, g, b, eps, y, ldy, mean, rstd, rows, blockIdx.x, gridDim.x);
}
// Two LayerNorms in one launch (the language and the vision stream of a cross-modal layer: different rows, different parameters):
// blocks [0, nb0) take problem 0, the rest problem 1
struct LnFwd2 {
  const void* x[2]; long ldx[2]; const float* g[2]; const float* b[2]; void* y[2]; long ldy[2]; float* mean[2]; float* rstd[2];
  int rows[2]; int nb0;
};
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_fwd_dual_kernel(LnFwd2 p, float eps) {
  const int i = (int)blockIdx.x >= p.nb0 ? 1 : 0;
  ln_fwd_body<T, NC>((const T*)p.x[i], p.ldx[i], p.g[i], p.b[i], eps, (T*)p.y[i], p.ldy[i], p.mean[i], p.rstd[i], p.rows[i],
                     i ? blockIdx.x - p.nb0 : blockIdx.x, i ? gridDim.x - p.nb0 : p.nb0);
}

// dx = rstd * (g*dy - mean_H(g*dy) - xhat * mean_H(g*dy*xhat)); dgamma += sum_rows dy*xhat; dbeta += sum_rows dy
// backward: 16 waves per block so that the 2*H float atomics per block (dgamma / dbeta, every block into the same 6 KiB)
// are issued by <= 128 blocks instead of 512
constexpr int BWAVES = 16;
template <typename T, int NC>
__device__ __forceinline__ void ln_bwd_body(const T* __restrict__ dy, long lddy, const T* __restrict__ x, long ldx,
                                            const float* __restrict__ g, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, T* __restrict__ dx, long lddx,
                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                            const T* __restrict__ dres, long lddres, T* __restrict__ dxd,
                                            long lddxd, unsigned dthr, unsigned dseed0, float dinv, const unsigned* sbase,
                                            int blk, int nblk, float (*red)[2][256 * NC]) {
  const unsigned dseed = dxd ? eff_seed(dseed0, sbase) : 0u;
  constexpr int H = 256 * NC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gv[NC], pg[NC], pb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    gv[c] = *(const f32x4*)(g + c * 256 + lane * 4);
    pg[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    pb[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int row = blk * BWAVES + wave; row < rows; row += nblk * BWAVES) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NC], d[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const f32x4 xv = DT<T>::ld4(x + (long)row * ldx + c * 256 + lane * 4);
      d[c] = DT<T>::ld4(dy + (long)row * lddy + c * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[c][j] = (xv[j] - mu) * rs;
        pg[c][j] += d[c][j] * xh[c][j];
        pb[c][j] += d[c][j];
        const float dg = d[c][j] * gv[c][j];
        s1 += dg;
        s2 += dg * xh[c][j];
      }
    }
    s1 = wave_sum(s1) * (1.0f / H);
    s2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rs * (d[c][j] * gv[c][j] - s1 - xh[c][j] * s2);
      if (dxd) {   // gradient w.r.t. the dropped dense output that was added to the residual before this LayerNorm
        f32x4 od;
#pragma unroll
        for (int j = 0; j < 4; ++j) od[j] = o[j] * drop_scale((unsigned)row * H + c * 256 + lane * 4 + j, dseed, dthr, dinv);
        DT<T>::st4(dxd + (long)row * lddxd + c * 256 + lane * 4, od);
      }
      if (dres) o += DT<T>::ld4(dres + (long)row * lddres + c * 256 + lane * 4);   // pre-norm blocks: + residual-path gradient
      DT<T>::st4(dx + (long)row * lddx + c * 256 + lane * 4, o);
    }
  }
  if (dgamma == nullptr) return;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[wave][0][c * 256 + lane * 4 + j] = pg[c][j];
      red[wave][1][c * 256 + lane * 4 + j] = pb[c][j];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += BWAVES * 64) {
    float a = 0.f, bsum = 0.f;
#pragma unroll
    for (int w = 0; w < BWAVES; ++w) {
      a += red[w][0][i];
      bsum += red[w][1][i];
    }
    atomicAdd(dgamma + i, a);
    atomicAdd(dbeta + i, bsum);
  }
}
template <typename T, int NC>
__global__ __launch_bounds__(BWAVES * 64) void ln_bwd_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ x, long ldx,
                                                     const float* __restrict__ g, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx, long lddx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                     const T* __restrict__ dres, long lddres, T* __restrict__ dxd,
                                                     long lddxd, unsigned dthr, unsigned dseed0, float dinv, const unsigned* sbase) {
  __shared__ float red[BWAVES][2][256 * NC];
  ln_bwd_body<T, NC>(dy, lddy, x, ldx, g, mean, rstd, dx, lddx, dgamma, dbeta, rows, dres, lddres, dxd, lddxd, dthr, dseed0, dinv, sbase,
                     blockIdx.x, gridDim.x, red);
}
struct LnBwd2 {
  const void* dy[2]; long lddy[2]; const void* x[2]; long ldx[2]; const float* g[2]; const float* mean[2]; const float* rstd[2];
  void* dx[2]; long lddx[2]; float* dgamma[2]; float* dbeta[2]; int rows[2]; const void* dres[2]; long lddres[2]; void* dxd[2]; long lddxd[2];
  unsigned dseed[2]; int nb0;
};
template <typename T, int NC>
__global__ __launch_bounds__(BWAVES * 64) void ln_bwd_dual_kernel(LnBwd2 p, unsigned dthr, float dinv, const unsigned* sbase) {
  __shared__ float red[BWAVES][2][256 * NC];
  const int i = (int)blockIdx.x >= p.nb0 ? 1 : 0;
  ln_bwd_body<T, NC>((const T*)p.dy[i], p.lddy[i], (const T*)p.x[i], p.ldx[i], p.g[i], p.mean[i], p.rstd[i], (T*)p.dx[i], p.lddx[i],
                     p.dgamma[i], p.dbeta[i], p.rows[i], (const T*)p.dres[i], p.lddres[i], (T*)p.dxd[i], p.lddxd[i], dthr, p.dseed[i], dinv,
                     sbase, i ? blockIdx.x - p.nb0 : blockIdx.x, i ? gridDim.x - p.nb0 : p.nb0, red);
}

// ---- sum of up to 4 row sources, then LayerNorm --------------------------------------------------
struct SumSrc {
  const void* p;        // base
  long ld;              // row stride in elements (0 = one broadcast row)
  const long* idx;      // optional row gather: row r reads p + idx[r]*ld
  int is_f32;           // 1: float source (parameter tables); 0: activation dtype T
};
struct SumP {
  SumSrc s[4];
  int n;
};

template <typename T, int NC>
__global__ __launch_bounds__(256) void sum_ln_fwd_kernel(SumP sp, const float* __restrict__ g, const float* __restrict__ b,
                                                         float eps, T* __restrict__ y, long ldy, T* __restrict__ xsum,
                                                         long ldxs, float* __restrict__ mean, float* __restrict__ rstd,
                                                         int rows) {
  constexpr int H = 256 * NC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * WAVES + wave; row < rows; row += gridDim.x * WAVES) {
    f32x4 v[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < sp.n; ++k) {
      const SumSrc& s = sp.s[k];
      const long r = s.idx ? s.idx[row] : (long)row;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const long off = r * s.ld + c * 256 + lane * 4;
        const f32x4 t = s.is_f32 ? *(const f32x4*)((const float*)s.p + off) : DT<T>::ld4((const T*)s.p + off);
        v[c] += t;
      }
    }
    if (xsum) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        DT<T>::st4(xsum + (long)row * ldxs + c * 256 + lane * 4, v[c]);
        v[c] = DT<T>::ld4(xsum + (long)row * ldxs + c * 256 + lane * 4);   // normalise exactly what backward will re-read
      }
    }
    float s0 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) s0 += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    const float mu = wave_sum(s0) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[c][j] - mu;
        q += d * d;
      }
    const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / H) + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const f32x4 gv = *(const f32x4*)(g + c * 256 + lane * 4), bv = *(const f32x4*)(b + c * 256 + lane * 4);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mu) * rs * gv[j] + bv[j];
      DT<T>::st4(y + (long)row * ldy + c * 256 + lane * 4, o);
    }
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
  }
}

// ---- fused "embed and combine" (round 5): the observation / history / panorama / map-node embeddings as ONE launch ----
//   y = dropout( [LN_o]( [LN_a](a) + LN_b(f W_b^T + b_b) + extra + row + table[idx] + table2[idx2] ) )
// a: the image linear's output [rows, H] (a GEMM of its own); f: K <= 16 float32 angle / position features per row; row: one broadcast
// float32 row; table / table2: float32 embedding tables gathered by int64 row indices; extra: one more dense [rows, H] activation
// (HistoryEmbeddings: the mean of the encoded panorama). Replaces, per call, LayerNorm + small-K linear + LayerNorm + sum-LayerNorm +
// dropout (ImageEmbeddings R:535-544, HistoryEmbeddings R:596-618, D:1087-1131, D:1140-1156) - five launches of 5-8 us on the critical path in
// front of the first cross-modal layer. Every intermediate the unfused operators round to the activation type is rounded here the same
// way (LN_a / LN_b outputs, the small linear's output, the pre-LN sum), so the results are the unfused ones bit for bit; what the
// backward needs is written out: the small linear's output, the pre-LN sum and the three LayerNorms' statistics.
struct EmbedP {
  const void* a; long lda; const float* ga; const float* ba;
  const float* f; long ldf; int K; const float* Wb; const float* bb; const float* gb; const float* beb;
  const void* extra; long lde;
  const float* row; const float* table; const long* idx; const float* table2; const long* idx2;
  const float* go; const float* bo; float eps;
  void* linb; void* xsum; void* y; long ldy;
  float* mean_a; float* rstd_a; float* mean_b; float* rstd_b; float* mean_o; float* rstd_o;
  unsigned drop_thr, drop_seed; float drop_inv; const unsigned* sbase;
  int rows;
  int trows, trows2; int* ierr;      // rows of table / table2 (an index outside adds nothing and is counted in *ierr when set)
};
template <typename T> __device__ __forceinline__ float rt(float x) { return (float)(T)x; }       // round to the activation type and back
template <> __device__ __forceinline__ float rt<float>(float x) { return x; }

template <typename T, int NC>
__device__ __forceinline__ void ln_inplace(f32x4 (&v)[NC], const float* __restrict__ g, const float* __restrict__ b, float eps, int lane,
                                           float& mu, float& rs) {
  constexpr int H = 256 * NC;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
  mu = wave_sum(s) * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = v[c][j] - mu;
      q += d * d;
    }
  rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / H) + eps);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const f32x4 gv = *(const f32x4*)(g + c * 256 + lane * 4), bv = *(const f32x4*)(b + c * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[c][j] = rt<T>((v[c][j] - mu) * rs * gv[j] + bv[j]);     // what the unfused LayerNorm stores
  }
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void embed_combine_kernel(EmbedP p) {
  constexpr int H = 256 * NC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned seed = p.drop_thr ? eff_seed(p.drop_seed, p.sbase) : 0u;
  for (int row = blockIdx.x * WAVES + wave; row < p.rows; row += gridDim.x * WAVES) {
    f32x4 v[NC], t[NC];
    float mu, rs;
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = DT<T>::ld4((const T*)p.a + (long)row * p.lda + c * 256 + lane * 4);
    if (p.ga) {
      ln_inplace<T, NC>(v, p.ga, p.ba, p.eps, lane, mu, rs);
      if (lane == 0) { p.mean_a[row] = mu; p.rstd_a[row] = rs; }
    }
    if (p.f) {
      // K == 4 (the angle features, vilmodel_cmt.py:537,599): the row's features and each column's four weights are one 16-byte load each
      // (the generic loop below is 16 predicated dword loads per output element: 170 us on the episode's 14 k history rows, 6 x the
      // launch's HBM time). Same products in the same order either way.
      const bool k4 = p.K == 4 && (p.ldf & 3) == 0 && (((uintptr_t)p.f | (uintptr_t)p.Wb) & 15) == 0;
      float fr[16];
      if (k4) {
        const f32x4 f4 = *(const f32x4*)(p.f + (long)row * p.ldf);
        fr[0] = f4[0]; fr[1] = f4[1]; fr[2] = f4[2]; fr[3] = f4[3];
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) fr[k] = k < p.K ? p.f[(long)row * p.ldf + k] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bb) b4 = *(const f32x4*)(p.bb + c * 256 + lane * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = c * 256 + lane * 4 + j;
          float acc = b4[j];
          if (k4) {
            const f32x4 w4 = *(const f32x4*)(p.Wb + (long)col * 4);
            acc += fr[0] * w4[0]; acc += fr[1] * w4[1]; acc += fr[2] * w4[2]; acc += fr[3] * w4[3];
          } else {
#pragma unroll
            for (int k = 0; k < 16; ++k)
              if (k < p.K) acc += fr[k] * p.Wb[(long)col * p.K + k];
          }
          t[c][j] = acc;
        }
        if (p.linb) DT<T>::st4((T*)p.linb + (long)row * H + c * 256 + lane * 4, t[c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[c][j] = rt<T>(t[c][j]);                               // normalise what the backward will re-read
      }
      ln_inplace<T, NC>(t, p.gb, p.beb, p.eps, lane, mu, rs);
      if (lane == 0) { p.mean_b[row] = mu; p.rstd_b[row] = rs; }
#pragma unroll
      for (int c = 0; c < NC; ++c) v[c] += t[c];
    }
    if (p.row) {
#pragma unroll
      for (int c = 0; c < NC; ++c) v[c] += *(const f32x4*)(p.row + c * 256 + lane * 4);
    }
    if (p.table) {
      const long r = p.idx[row];
      if (r >= 0 && r < p.trows) {
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] += *(const f32x4*)(p.table + r * H + c * 256 + lane * 4);
      } else if (p.ierr && lane == 0) atomicAdd(p.ierr, 1);
    }
    if (p.table2) {
      const long r = p.idx2[row];
      if (r >= 0 && r < p.trows2) {
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] += *(const f32x4*)(p.table2 + r * H + c * 256 + lane * 4);
      } else if (p.ierr && lane == 0) atomicAdd(p.ierr, 1);
    }
    if (p.extra) {
#pragma unroll
      for (int c = 0; c < NC; ++c) v[c] += DT<T>::ld4((const T*)p.extra + (long)row * p.lde + c * 256 + lane * 4);
    }
    if (p.go) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        DT<T>::st4((T*)p.xsum + (long)row * H + c * 256 + lane * 4, v[c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[c][j] = rt<T>(v[c][j]);
      }
      ln_inplace<T, NC>(v, p.go, p.bo, p.eps, lane, mu, rs);
      if (lane == 0) { p.mean_o[row] = mu; p.rstd_o[row] = rs; }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (p.drop_thr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[c][j] *= drop_scale((unsigned)((long)row * H + c * 256 + lane * 4 + j), seed, p.drop_thr, p.drop_inv);
      }
      DT<T>::st4((T*)p.y + (long)row * p.ldy + c * 256 + lane * 4, v[c]);
    }
  }
}

// ---- fused prediction-head tail (round 5): logits[r] = mask[r] ? -inf : <dropout(LN(x[r])), w> + bias ----
// NextActionPrediction R:953-963 (LayerNorm, Dropout, Linear(768 -> 1), then masked_fill R:1200) and DUET's ClsPrediction D:1009-1020 after their
// first Linear + ReLU: LayerNorm + dropout + row dot were three launches of ~5 us at the end of every step. hd (the dropped LayerNorm output, what
// the row dot's weight gradient needs) and the statistics are written for the backward, which stays vlni_rowdot_bwd + the mask + vlni_layernorm_bwd.
// Same roundings and the same summation order as the three kernels: identical results.
template <typename T, int NC>
__global__ __launch_bounds__(256) void ln_rowdot_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ g, const float* __restrict__ b,
                                                        float eps, T* __restrict__ hd, float* __restrict__ mean, float* __restrict__ rstd,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const unsigned char* __restrict__ mask, float* __restrict__ out, unsigned thr,
                                                        unsigned seed0, float inv, const unsigned* sbase, int rows) {
  constexpr int H = 256 * NC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned seed = thr ? eff_seed(seed0, sbase) : 0u;
  for (int row = blockIdx.x * WAVES + wave; row < rows; row += gridDim.x * WAVES) {
    f32x4 v[NC];
    float mu, rs;
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = DT<T>::ld4(x + (long)row * ldx + c * 256 + lane * 4);
    ln_inplace<T, NC>(v, g, b, eps, lane, mu, rs);
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (thr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[c][j] = rt<T>(v[c][j] * drop_scale((unsigned)((long)row * H + c * 256 + lane * 4 + j), seed, thr, inv));
      }
      DT<T>::st4(hd + (long)row * H + c * 256 + lane * 4, v[c]);
      const f32x4 ww = *(const f32x4*)(w + c * 256 + lane * 4);
      a += v[c][0] * ww[0] + v[c][1] * ww[1] + v[c][2] * ww[2] + v[c][3] * ww[3];
    }
    a = wave_sum(a);
    if (lane == 0) {
      mean[row] = mu; rstd[row] = rs;
      out[row] = (mask && mask[row]) ? -INFINITY : a + (bias ? bias[0] : 0.f);
    }
  }
}

int ln_grid(int rows) { return max(1, min(cdiv(rows, WAVES), 2048)); }
int ln_bwd_grid(int rows) { return max(1, min(cdiv(rows, BWAVES), 128)); }

}  // namespace

#define LN_DISPATCH(KERNEL, T, ...) LN_DISPATCH_GB(ln_grid(rows), 256, KERNEL, T, __VA_ARGS__)
#define LN_DISPATCH_G(GRID, KERNEL, T, ...) LN_DISPATCH_GB(GRID, BWAVES * 64, KERNEL, T, __VA_ARGS__)
#define LN_DISPATCH_GB(GRID, BLOCK, KERNEL, T, ...)                                                              \
  do {                                                                                                   \
    const int nc = H / 256;                                                                              \
    dim3 grid(GRID), block(BLOCK);                                                                       \
    hipStream_t st = (hipStream_t)stream;                                                                \
    if (nc == 3) hipLaunchKernelGGL((KERNEL<T, 3>), grid, block, 0, st, __VA_ARGS__);                    \
    else if (nc == 2) hipLaunchKernelGGL((KERNEL<T, 2>), grid, block, 0, st, __VA_ARGS__);               \
    else hipLaunchKernelGGL((KERNEL<T, 1>), grid, block, 0, st, __VA_ARGS__);                            \
  } while (0)

static int ln_check(const char* who, int dtype, int rows, int H, long ld) {
  VLNI_CHECK(dtype == VLNI_F32 || dtype == VLNI_BF16 || dtype == VLNI_F16, VLNI_EINVAL, "%s: bad dtype %d", who, dtype);
  VLNI_CHECK(H == 256 || H == 512 || H == 768, VLNI_EUNSUP, "%s: H=%d not in {256,512,768}", who, H);
  VLNI_CHECK(rows > 0 && ld >= H && ld % 4 == 0, VLNI_EINVAL, "%s: rows=%d ld=%ld", who, rows, ld);
  return VLNI_OK;
}

extern "C" int vlni_layernorm_fwd(int dtype, const void* x, long ldx, const float* gamma, const float* beta, float eps,
                                  void* y, long ldy, float* mean, float* rstd, int rows, int H, void* stream) {
  int rc = ln_check("layernorm_fwd", dtype, rows, H, ldx < ldy ? ldx : ldy);
  if (rc) return rc;
  if (dtype == VLNI_F32) {
    using TT = float;
    const float* xx = (const float*)x; float* yy = (float*)y;
    LN_DISPATCH(ln_fwd_kernel, TT, xx, ldx, gamma, beta, eps, yy, ldy, mean, rstd, rows);
  } else if (dtype == VLNI_BF16) {
    using TT = __bf16;
    const __bf16* xx = (const __bf16*)x; __bf16* yy = (__bf16*)y;
    LN_DISPATCH(ln_fwd_kernel, TT, xx, ldx, gamma, beta, eps, yy, ldy, mean, rstd, rows);
  } else {
    using TT = _Float16;
    const _Float16* xx = (const _Float16*)x; _Float16* yy = (_Float16*)y;
    LN_DISPATCH(ln_fwd_kernel, TT, xx, ldx, gamma, beta, eps, yy, ldy, mean, rstd, rows);
  }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_layernorm_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, const float* gamma,
                                  const float* mean, const float* rstd, void* dx, long lddx, float* dgamma, float* dbeta,
                                  int rows, int H, const void* dres, long lddres, void* dx_drop, long lddxd, float drop_p,
                                  unsigned drop_seed, void* stream) {
  int rc = ln_check("layernorm_bwd", dtype, rows, H, ldx < lddx ? (ldx < lddy ? ldx : lddy) : (lddx < lddy ? lddx : lddy));
  if (rc) return rc;
  VLNI_CHECK((dgamma == nullptr) == (dbeta == nullptr), VLNI_EINVAL, "layernorm_bwd: dgamma/dbeta both or neither");
  if (dtype == VLNI_F32) {
    using TT = float;
    const float* d = (const float*)dy; const float* xx = (const float*)x; float* o = (float*)dx; const float* dr = (const float*)dres; float* dd = (float*)dx_drop;
    LN_DISPATCH_G(ln_bwd_grid(rows), ln_bwd_kernel, TT, d, lddy, xx, ldx, gamma, mean, rstd, o, lddx, dgamma, dbeta, rows, dr, lddres, dd, lddxd, drop_thr(drop_p), drop_seed, 1.0f / (1.0f - drop_p), vlni_seed_base());
  } else if (dtype == VLNI_BF16) {
    using TT = __bf16;
    const __bf16* d = (const __bf16*)dy; const __bf16* xx = (const __bf16*)x; __bf16* o = (__bf16*)dx; const __bf16* dr = (const __bf16*)dres; __bf16* dd = (__bf16*)dx_drop;
    LN_DISPATCH_G(ln_bwd_grid(rows), ln_bwd_kernel, TT, d, lddy, xx, ldx, gamma, mean, rstd, o, lddx, dgamma, dbeta, rows, dr, lddres, dd, lddxd, drop_thr(drop_p), drop_seed, 1.0f / (1.0f - drop_p), vlni_seed_base());
  } else {
    using TT = _Float16;
    const _Float16* d = (const _Float16*)dy; const _Float16* xx = (const _Float16*)x; _Float16* o = (_Float16*)dx; const _Float16* dr = (const _Float16*)dres; _Float16* dd = (_Float16*)dx_drop;
    LN_DISPATCH_G(ln_bwd_grid(rows), ln_bwd_kernel, TT, d, lddy, xx, ldx, gamma, mean, rstd, o, lddx, dgamma, dbeta, rows, dr, lddres, dd, lddxd, drop_thr(drop_p), drop_seed, 1.0f / (1.0f - drop_p), vlni_seed_base());
  }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// Two LayerNorm problems in one launch (arrays of 2; same dtype, H and eps; block ranges [0, nb0) and [nb0, nb0 + nb1))
extern "C" int vlni_layernorm_fwd_dual(int dtype, const void* const* x, const long* ldx, const float* const* gamma,
                                       const float* const* beta, float eps, void* const* y, const long* ldy, float* const* mean,
                                       float* const* rstd, const int* rows, int H, void* stream) {
  LnFwd2 p;
  for (int i = 0; i < 2; ++i) {
    int rc = ln_check("layernorm_fwd_dual", dtype, rows[i], H, ldx[i] < ldy[i] ? ldx[i] : ldy[i]);
    if (rc) return rc;
    p.x[i] = x[i]; p.ldx[i] = ldx[i]; p.g[i] = gamma[i]; p.b[i] = beta[i]; p.y[i] = y[i]; p.ldy[i] = ldy[i];
    p.mean[i] = mean[i]; p.rstd[i] = rstd[i]; p.rows[i] = rows[i];
  }
  p.nb0 = ln_grid(rows[0]);
  const int nblk = p.nb0 + ln_grid(rows[1]);
  if (dtype == VLNI_F32) { using TT = float; LN_DISPATCH_GB(nblk, 256, ln_fwd_dual_kernel, TT, p, eps); }
  else if (dtype == VLNI_BF16) { using TT = __bf16; LN_DISPATCH_GB(nblk, 256, ln_fwd_dual_kernel, TT, p, eps); }
  else { using TT = _Float16; LN_DISPATCH_GB(nblk, 256, ln_fwd_dual_kernel, TT, p, eps); }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
extern "C" int vlni_layernorm_bwd_dual(int dtype, const void* const* dy, const long* lddy, const void* const* x, const long* ldx,
                                       const float* const* gamma, const float* const* mean, const float* const* rstd, void* const* dx,
                                       const long* lddx, float* const* dgamma, float* const* dbeta, const int* rows, int H,
                                       const void* const* dres, const long* lddres, void* const* dx_drop, const long* lddxd, float drop_p,
                                       const unsigned* drop_seed, void* stream) {
  LnBwd2 p;
  for (int i = 0; i < 2; ++i) {
    int rc = ln_check("layernorm_bwd_dual", dtype, rows[i], H,
                      ldx[i] < lddx[i] ? (ldx[i] < lddy[i] ? ldx[i] : lddy[i]) : (lddx[i] < lddy[i] ? lddx[i] : lddy[i]));
    if (rc) return rc;
    VLNI_CHECK((dgamma[i] == nullptr) == (dbeta[i] == nullptr), VLNI_EINVAL, "layernorm_bwd_dual: dgamma/dbeta both or neither");
    p.dy[i] = dy[i]; p.lddy[i] = lddy[i]; p.x[i] = x[i]; p.ldx[i] = ldx[i]; p.g[i] = gamma[i]; p.mean[i] = mean[i]; p.rstd[i] = rstd[i];
    p.dx[i] = dx[i]; p.lddx[i] = lddx[i]; p.dgamma[i] = dgamma[i]; p.dbeta[i] = dbeta[i]; p.rows[i] = rows[i];
    p.dres[i] = dres ? dres[i] : nullptr; p.lddres[i] = dres && lddres ? lddres[i] : 0;
    p.dxd[i] = dx_drop ? dx_drop[i] : nullptr; p.lddxd[i] = dx_drop && lddxd ? lddxd[i] : 0;
    p.dseed[i] = drop_seed ? drop_seed[i] : 0u;
  }
  p.nb0 = ln_bwd_grid(rows[0]);
  const int nblk = p.nb0 + ln_bwd_grid(rows[1]);
  const unsigned thr = drop_thr(drop_p);
  const float inv = 1.0f / (1.0f - drop_p);
  if (dtype == VLNI_F32) { using TT = float; LN_DISPATCH_GB(nblk, BWAVES * 64, ln_bwd_dual_kernel, TT, p, thr, inv, vlni_seed_base()); }
  else if (dtype == VLNI_BF16) { using TT = __bf16; LN_DISPATCH_GB(nblk, BWAVES * 64, ln_bwd_dual_kernel, TT, p, thr, inv, vlni_seed_base()); }
  else { using TT = _Float16; LN_DISPATCH_GB(nblk, BWAVES * 64, ln_bwd_dual_kernel, TT, p, thr, inv, vlni_seed_base()); }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// y = LayerNorm(sum_k src_k); xsum (optional) receives the pre-norm sum for the backward pass.
// src arrays have n (1..4) entries: base pointer, row stride, optional int64 row-gather index, f32 flag.
extern "C" int vlni_sum_layernorm_fwd(int dtype, int n, const void* const* src, const long* src_ld,
                                      const long* const* src_idx, const int* src_is_f32, const float* gamma,
                                      const float* beta, float eps, void* y, long ldy, void* xsum, long ldxs, float* mean,
                                      float* rstd, int rows, int H, void* stream) {
  int rc = ln_check("sum_layernorm_fwd", dtype, rows, H, ldy);
  if (rc) return rc;
  VLNI_CHECK(n >= 1 && n <= 4, VLNI_EINVAL, "sum_layernorm_fwd: n=%d", n);
  SumP sp;
  sp.n = n;
  for (int i = 0; i < n; ++i) {
    sp.s[i].p = src[i]; sp.s[i].ld = src_ld[i]; sp.s[i].idx = src_idx ? src_idx[i] : nullptr;
    sp.s[i].is_f32 = src_is_f32[i];
    VLNI_CHECK(src_ld[i] % 4 == 0, VLNI_EINVAL, "sum_layernorm_fwd: src %d ld %ld", i, src_ld[i]);
  }
  if (dtype == VLNI_F32) {
    using TT = float;
    float* yy = (float*)y; float* xs = (float*)xsum;
    LN_DISPATCH(sum_ln_fwd_kernel, TT, sp, gamma, beta, eps, yy, ldy, xs, ldxs, mean, rstd, rows);
  } else if (dtype == VLNI_BF16) {
    using TT = __bf16;
    __bf16* yy = (__bf16*)y; __bf16* xs = (__bf16*)xsum;
    LN_DISPATCH(sum_ln_fwd_kernel, TT, sp, gamma, beta, eps, yy, ldy, xs, ldxs, mean, rstd, rows);
  } else {
    using TT = _Float16;
    _Float16* yy = (_Float16*)y; _Float16* xs = (_Float16*)xsum;
    LN_DISPATCH(sum_ln_fwd_kernel, TT, sp, gamma, beta, eps, yy, ldy, xs, ldxs, mean, rstd, rows);
  }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// logits = masked(<dropout(LayerNorm(x)), w> + bias) in one launch (see ln_rowdot_kernel). hd [rows, H], mean / rstd [rows], out [rows] float32.
extern "C" int vlni_ln_rowdot_fwd(int dtype, const void* x, long ldx, const float* gamma, const float* beta, float eps, void* hd, float* mean,
                                  float* rstd, const float* w, const float* bias, const unsigned char* mask, float* out, float drop_p,
                                  unsigned drop_seed, int rows, int H, void* stream) {
  int rc = ln_check("ln_rowdot_fwd", dtype, rows, H, ldx);
  if (rc) return rc;
  VLNI_CHECK(x && gamma && beta && hd && mean && rstd && w && out, VLNI_EINVAL, "ln_rowdot_fwd: null pointer");
  VLNI_CHECK(drop_p >= 0.f && drop_p < 1.f, VLNI_EINVAL, "ln_rowdot_fwd: dropout p=%f", drop_p);
  const unsigned thr = drop_thr(drop_p);
  const float inv = 1.0f / (1.0f - drop_p);
  if (dtype == VLNI_F32) { using TT = float; LN_DISPATCH(ln_rowdot_kernel, TT, (const float*)x, ldx, gamma, beta, eps, (float*)hd, mean, rstd, w, bias, mask, out, thr, drop_seed, inv, vlni_seed_base(), rows); }
  else if (dtype == VLNI_BF16) { using TT = __bf16; LN_DISPATCH(ln_rowdot_kernel, TT, (const __bf16*)x, ldx, gamma, beta, eps, (__bf16*)hd, mean, rstd, w, bias, mask, out, thr, drop_seed, inv, vlni_seed_base(), rows); }
  else { using TT = _Float16; LN_DISPATCH(ln_rowdot_kernel, TT, (const _Float16*)x, ldx, gamma, beta, eps, (_Float16*)hd, mean, rstd, w, bias, mask, out, thr, drop_seed, inv, vlni_seed_base(), rows); }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// Fused embed-and-combine forward (see embed_combine_kernel). Optional parts are NULL: ga/ba (no LayerNorm on a), f (no small-K branch), extra,
// row, table/idx, table2/idx2, go/bo (no outer LayerNorm: xsum / mean_o / rstd_o unused), drop_p 0. linb [rows, H], xsum [rows, H] dense.
extern "C" int vlni_embed_combine_fwd(int dtype, const void* a, long lda, const float* ga, const float* ba, const float* f, long ldf, int K,
                                      const float* Wb, const float* bb, const float* gb, const float* beb, const void* extra, long lde,
                                      const float* row, const float* table, const long* idx, int table_rows, const float* table2, const long* idx2,
                                      int table2_rows, const float* go, const float* bo, float eps, void* linb, void* xsum, void* y, long ldy, float* mean_a,
                                      float* rstd_a, float* mean_b, float* rstd_b, float* mean_o, float* rstd_o, float drop_p, unsigned drop_seed,
                                      int rows, int H, void* stream) {
  int rc = ln_check("embed_combine_fwd", dtype, rows, H, lda < ldy ? lda : ldy);
  if (rc) return rc;
  VLNI_CHECK(a && y, VLNI_EINVAL, "embed_combine_fwd: null a / y");
  VLNI_CHECK((ga == nullptr) == (ba == nullptr) && (ga == nullptr || (mean_a && rstd_a)), VLNI_EINVAL, "embed_combine_fwd: LN_a parameters / statistics");
  VLNI_CHECK(f == nullptr || (K >= 1 && K <= 16 && Wb && gb && beb && mean_b && rstd_b), VLNI_EINVAL, "embed_combine_fwd: small-K branch (K=%d)", K);
  VLNI_CHECK((go == nullptr) == (bo == nullptr) && (go == nullptr || (xsum && mean_o && rstd_o)), VLNI_EINVAL, "embed_combine_fwd: outer LayerNorm");
  VLNI_CHECK((table == nullptr) == (idx == nullptr) && (table2 == nullptr) == (idx2 == nullptr), VLNI_EINVAL, "embed_combine_fwd: table without index");
  VLNI_CHECK((table == nullptr || table_rows > 0) && (table2 == nullptr || table2_rows > 0), VLNI_EINVAL, "embed_combine_fwd: table rows %d / %d", table_rows,
             table2_rows);
  VLNI_CHECK(extra == nullptr || lde % 4 == 0, VLNI_EINVAL, "embed_combine_fwd: extra stride %ld", lde);
  VLNI_CHECK(drop_p >= 0.f && drop_p < 1.f, VLNI_EINVAL, "embed_combine_fwd: dropout p=%f", drop_p);
  EmbedP p;
  p.a = a; p.lda = lda; p.ga = ga; p.ba = ba; p.f = f; p.ldf = ldf; p.K = K; p.Wb = Wb; p.bb = bb; p.gb = gb; p.beb = beb; p.extra = extra; p.lde = lde;
  p.row = row; p.table = table; p.idx = idx; p.table2 = table2; p.idx2 = idx2; p.go = go; p.bo = bo; p.eps = eps; p.linb = linb; p.xsum = xsum; p.y = y;
  p.ldy = ldy; p.mean_a = mean_a; p.rstd_a = rstd_a; p.mean_b = mean_b; p.rstd_b = rstd_b; p.mean_o = mean_o; p.rstd_o = rstd_o;
  p.drop_thr = drop_thr(drop_p); p.drop_seed = drop_seed; p.drop_inv = 1.0f / (1.0f - drop_p); p.sbase = vlni_seed_base(); p.rows = rows;
  p.trows = table_rows; p.trows2 = table2_rows; p.ierr = vlni_index_error_counter();
  if (dtype == VLNI_F32) { using TT = float; LN_DISPATCH(embed_combine_kernel, TT, p); }
  else if (dtype == VLNI_BF16) { using TT = __bf16; LN_DISPATCH(embed_combine_kernel, TT, p); }
  else { using TT = _Float16; LN_DISPATCH(embed_combine_kernel, TT, p); }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// BertSelfOutput / BertOutput tail (VLN-HAMT/finetune_src/models/vilmodel_cmt.py:144-148,186-190) as ONE launch per direction for a dense
// output that does not come from this library's GEMM (whose epilogue adds the bias and the residual itself, leaving plain LayerNorm):
//   fwd  y = LayerNorm(x + bias + residual); xsum (optional) keeps the pre-norm sum - the `x` the backward needs
//   bwd  dx = d(x) = d(residual) (+= nothing), dgamma / dbeta += ..., dbias += column sums of dx
// Built from the sum-of-sources LayerNorm and the LayerNorm backward above (SURVEY 8b names these two entry points).
extern "C" int vlni_colsum(int dtype, const void* x, long ldx, int rows, int N, float* out, void* stream);
extern "C" int vlni_bias_residual_layernorm_fwd(int dtype, const void* x, long ldx, const float* bias, const void* residual, long ldr,
                                                const float* gamma, const float* beta, float eps, void* y, long ldy, void* xsum, long ldxs,
                                                float* mean, float* rstd, int rows, int H, void* stream) {
  VLNI_CHECK(x && gamma && beta && y && mean && rstd, VLNI_EINVAL, "bias_residual_layernorm_fwd: null pointer");
  const void* src[3];
  long ld[3];
  int f32[3], n = 0;
  src[n] = x; ld[n] = ldx; f32[n++] = 0;
  if (residual) { src[n] = residual; ld[n] = ldr; f32[n++] = 0; }
  if (bias) { src[n] = bias; ld[n] = 0; f32[n++] = 1; }
  return vlni_sum_layernorm_fwd(dtype, n, src, ld, nullptr, f32, gamma, beta, eps, y, ldy, xsum, ldxs, mean, rstd, rows, H, stream);
}
extern "C" int vlni_bias_residual_layernorm_bwd(int dtype, const void* dy, long lddy, const void* xsum, long ldxs, const float* gamma,
                                                const float* mean, const float* rstd, void* dx, long lddx, float* dgamma, float* dbeta,
                                                float* dbias, int rows, int H, void* stream) {
  int rc = vlni_layernorm_bwd(dtype, dy, lddy, xsum, ldxs, gamma, mean, rstd, dx, lddx, dgamma, dbeta, rows, H, nullptr, 0, nullptr, 0, 0.f, 0u,
                              stream);
  if (rc || !dbias) return rc;
  return vlni_colsum(dtype, dx, lddx, rows, H, dbias, stream);
}
