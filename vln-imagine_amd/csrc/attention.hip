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

namespace {

constexpr int LD = 65;   // LDS row stride (dwords) of the 64-wide head tiles

template <typename T>
__device__ __forceinline__ void stage_rows(float* dst, const T* src, long ld, int row0, int nrows_valid, int nrows_tile,
                                           int tid, int nthreads) {
  // dst[row][0..63] = src[(row0+row)*ld + 0..63] for row < nrows_valid, zero otherwise
  for (int i = tid; i < nrows_tile * 16; i += nthreads) {
    const int row = i >> 4, c4 = (i & 15) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows_valid) v = DT<T>::ld4(src + (long)(row0 + row) * ld + c4);
    float* d = dst + row * LD + c4;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
}

__device__ __forceinline__ int acc_row(int x, int hh) { return (x & 3) + 8 * (x >> 2) + 4 * hh; }

struct AttnP {
  const void *q, *k, *v;
  long ldq, ldk, ldv;
  const float* kmask;   // [B, Sk] additive or null
  const float* bias;    // [B, Sq, Sk] additive or null
  void* out; long ldo;
  float* lse;           // [B, nh, Sq]
  int B, nh, Sq, Sk;
  float scale;
  const unsigned* seed_base;                     // effective seed = drop_seed + *seed_base (vlni_set_dropout_seed_base)
  unsigned drop_thr, drop_seed; float drop_inv;   // attention-probability dropout (idx = ((b*nh+h)*Sq+q)*Sk+key)
  // backward only
  const void* dout; long lddo;
  void *dq, *dk, *dv;
  long lddq, lddk, lddv;
  float* dbias;         // [B, Sq, Sk] accumulated over heads (atomics) or null
};

// ------------------------------------------------------------------------------------------------
// forward: block = 2 waves, each wave 32 query rows; grid = (ceil(Sq/64), B*nh)
template <typename T, int NKT>
__global__ __launch_bounds__(128) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int SKP = NKT * 32;
  float* Ks = smem;
  float* Vs = Ks + SKP * LD;
  float* Qs = Vs + SKP * LD;
  const int bh = blockIdx.y, b = bh / p.nh, hd = bh % p.nh;
  const int q0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const unsigned dseed = p.drop_thr ? eff_seed(p.drop_seed, p.seed_base) : 0u;

  stage_rows<T>(Ks, (const T*)p.k + (long)b * p.Sk * p.ldk + hd * 64, p.ldk, 0, p.Sk, SKP, tid, 128);
  stage_rows<T>(Vs, (const T*)p.v + (long)b * p.Sk * p.ldv + hd * 64, p.ldv, 0, p.Sk, SKP, tid, 128);
  stage_rows<T>(Qs, (const T*)p.q + (long)b * p.Sq * p.ldq + hd * 64, p.ldq, q0, min(64, p.Sq - q0), 64, tid, 128);
  __syncthreads();

  const int qloc = wave * 32 + r;         // this lane's query row inside the block
  const int qg = q0 + qloc;               // global query index
  float qreg[32];
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) qreg[kk] = Qs[qloc * LD + 2 * kk + hh];

  f32x16 s[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int x = 0; x < 16; ++x) s[kt][x] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk)
      s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(kt * 32 + r) * LD + 2 * kk + hh], qreg[kk], s[kt], 0, 0, 0);
  }
  // s[kt][x] = <Q[q=r], K[key]>, key = kt*32 + acc_row(x, hh)
  const float* km = p.kmask ? p.kmask + (long)b * p.Sk : nullptr;
  const float* bs = (p.bias && qg < p.Sq) ? p.bias + ((long)b * p.Sq + qg) * p.Sk : nullptr;
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int key = kt * 32 + acc_row(x, hh);
      float v = s[kt][x] * p.scale;
      if (key < p.Sk) {
        if (km) v += km[key];
        if (bs) v += bs[key];
      } else {
        v = -INFINITY;
      }
      s[kt][x] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const float e = __expf(s[kt][x] - m);
      s[kt][x] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  if (p.drop_thr) {                                   // dropout on the probabilities (the row sum above stays un-dropped)
    const unsigned base = (unsigned)((b * p.nh + hd) * p.Sq + qg) * (unsigned)p.Sk;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int x = 0; x < 16; ++x) s[kt][x] *= drop_scale(base + kt * 32 + acc_row(x, hh), dseed, p.drop_thr, p.drop_inv);
  }

  T* out = (T*)p.out + ((long)b * p.Sq + qg) * p.ldo + hd * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    f32x16 o;
#pragma unroll
    for (int x = 0; x < 16; ++x) o[x] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int x = 0; x < 16; ++x)
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(kt * 32 + acc_row(x, hh)) * LD + dt * 32 + r], s[kt][x], o, 0, 0, 0);
    // o[y] = O^T[d = dt*32 + acc_row(y, hh)][q = r]
    if (qg < p.Sq) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
        DT<T>::st4(out + dt * 32 + 8 * g + 4 * hh, v);
      }
    }
  }
  if (p.lse && hh == 0 && qg < p.Sq) p.lse[((long)b * p.nh + hd) * p.Sq + qg] = m + __logf(l);
}

// ------------------------------------------------------------------------------------------------
// backward: block = 4 waves; wave w owns key tile w (32 keys); grid = B*nh; query rows in chunks of 64
template <typename T, int NKT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int SKP = NKT * 32, LDS_S = SKP + 1;
  float* Ks = smem;
  float* Vs = Ks + SKP * LD;
  float* Qs = Vs + SKP * LD;
  float* dOs = Qs + 64 * LD;
  float* dSs = dOs + 64 * LD;             // [64][SKP+1]
  float* lse_s = dSs + 64 * LDS_S;        // [64]
  float* del_s = lse_s + 64;              // [64]
  const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const unsigned dseed = p.drop_thr ? eff_seed(p.drop_seed, p.seed_base) : 0u;

  stage_rows<T>(Ks, (const T*)p.k + (long)b * p.Sk * p.ldk + hd * 64, p.ldk, 0, p.Sk, SKP, tid, 256);
  stage_rows<T>(Vs, (const T*)p.v + (long)b * p.Sk * p.ldv + hd * 64, p.ldv, 0, p.Sk, SKP, tid, 256);
  __syncthreads();

  const bool owner = wave < NKT;
  const int key = wave * 32 + r;           // this lane's key (phase 1)
  float kreg[32], vreg[32];
  if (owner) {
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      kreg[kk] = Ks[key * LD + 2 * kk + hh];
      vreg[kk] = Vs[key * LD + 2 * kk + hh];
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int x = 0; x < 16; ++x) { dk[dt][x] = 0.f; dv[dt][x] = 0.f; }
  const float kmv = (p.kmask && key < p.Sk) ? p.kmask[(long)b * p.Sk + key] : 0.f;

  for (int q0 = 0; q0 < p.Sq; q0 += 64) {
    const int nq = min(64, p.Sq - q0);
    stage_rows<T>(Qs, (const T*)p.q + (long)b * p.Sq * p.ldq + hd * 64, p.ldq, q0, nq, 64, tid, 256);
    stage_rows<T>(dOs, (const T*)p.dout + (long)b * p.Sq * p.lddo + hd * 64, p.lddo, q0, nq, 64, tid, 256);
    // delta[q] = sum_d dO[q][d] * O[q][d]; one wave per row, lane = d
    for (int row = wave; row < 64; row += 4) {
      float dl = 0.f;
      if (row < nq) {
        const long qrow = (long)b * p.Sq + q0 + row;
        dl = DT<T>::ld((const T*)p.dout + qrow * p.lddo + hd * 64 + lane) *
             DT<T>::ld((const T*)p.out + qrow * p.ldo + hd * 64 + lane);
      }
      dl = wave_sum(dl);
      if (lane == 0) {
        del_s[row] = dl;
        lse_s[row] = row < nq ? p.lse[((long)b * p.nh + hd) * p.Sq + q0 + row] : 0.f;
      }
    }
    __syncthreads();

    if (owner) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x16 s, dp;
#pragma unroll
        for (int x = 0; x < 16; ++x) { s[x] = 0.f; dp[x] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[(qt * 32 + r) * LD + 2 * kk + hh], kreg[kk], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[(qt * 32 + r) * LD + 2 * kk + hh], vreg[kk], dp, 0, 0, 0);
        }
        // s[x] / dp[x]: q = qt*32 + acc_row(x, hh), key = this lane's key
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          const int ql = qt * 32 + acc_row(x, hh);
          float pv = 0.f, ds = 0.f, pdrop = 0.f;
          if (key < p.Sk && ql < nq) {
            float sv = s[x] * p.scale + kmv;
            if (p.bias) sv += p.bias[((long)b * p.Sq + q0 + ql) * p.Sk + key];
            pv = __expf(sv - lse_s[ql]);
            float dpx = dp[x];
            if (p.drop_thr) {
              const float ms = drop_scale((unsigned)((b * p.nh + hd) * p.Sq + q0 + ql) * (unsigned)p.Sk + key, dseed,
                                          p.drop_thr, p.drop_inv);
              dpx *= ms;                      // dP = dP_dropped * mask / keep
              pdrop = pv * ms;                // dV uses the dropped probabilities
            } else {
              pdrop = pv;
            }
            ds = pv * (dpx - del_s[ql]);
            if (p.dbias) atomicAdd(p.dbias + ((long)b * p.Sq + q0 + ql) * p.Sk + key, ds);
          }
          s[x] = pdrop;
          dp[x] = ds * p.scale;
          dSs[ql * LDS_S + key] = dp[x];
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int x = 0; x < 16; ++x) {
            const int ql = qt * 32 + acc_row(x, hh);
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[ql * LD + dt * 32 + r], s[x], dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[ql * LD + dt * 32 + r], dp[x], dk[dt], 0, 0, 0);
          }
      }
    }
    __syncthreads();

    // phase 2: dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]; 4 tiles (qt, dt), one per wave
    {
      const int qt = wave >> 1, dt = wave & 1;
      f32x16 dq;
#pragma unroll
      for (int x = 0; x < 16; ++x) dq[x] = 0.f;
#pragma unroll 8
      for (int kk = 0; kk < SKP / 2; ++kk)
        dq = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * kk + hh) * LD + dt * 32 + r], dSs[(qt * 32 + r) * LDS_S + 2 * kk + hh], dq,
                                                  0, 0, 0);
      const int ql = qt * 32 + r;
      if (ql < nq) {
        T* o = (T*)p.dq + ((long)b * p.Sq + q0 + ql) * p.lddq + hd * 64 + dt * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {dq[4 * g], dq[4 * g + 1], dq[4 * g + 2], dq[4 * g + 3]};
          DT<T>::st4(o + 8 * g + 4 * hh, v);
        }
      }
    }
    __syncthreads();
  }

  if (owner && key < p.Sk) {
    T* ok = (T*)p.dk + ((long)b * p.Sk + key) * p.lddk + hd * 64;
    T* ov = (T*)p.dv + ((long)b * p.Sk + key) * p.lddv + hd * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a = {dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]};
        f32x4 c = {dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]};
        DT<T>::st4(ok + dt * 32 + 8 * g + 4 * hh, a);
        DT<T>::st4(ov + dt * 32 + 8 * g + 4 * hh, c);
      }
  }
}

// ================================================================================================
// bf16 data path (throughput): same algorithms on v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate).
// K, V, Q, dO tiles sit in LDS as bf16 rows of 128 bytes, 16-byte chunks XOR-swizzled with
//     f(row) = (((row>>1)&1)<<2) | ((row>>2)&3)
// which makes BOTH access kinds of the one image conflict-free: ds_read_b128 row fragments (32 lanes = 32 rows,
// same chunk) and ds_read_b64_tr_b16 transposed fragments (4 rows x 64 bytes per 32-lane half).
// Forward: S^T = K Q^T (lane owns a query row), softmax in registers, P^T converted pairwise to bf16 is already the
// B operand of O^T = V^T P^T (k order 16s + 8(j>>2) + 4h + (j&3), matched by the transposed V reads).
// Backward: keys on lanes; P and dS feed dV^T / dK^T from registers, dS crosses LDS once (bf16) for dQ.
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int boff(int row, int ch) { return row * 128 + ((ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4); }
__device__ __forceinline__ bf16x8 ld_row(const char* tile, int row, int ch) { return *(const bf16x8*)(tile + boff(row, ch)); }
// transposed fragment: rows r0..r0+3 and r1..r1+3 (this lane addresses row +qq), 16 columns starting at chunk ch (+half8)
__device__ __forceinline__ bf16x8 ld_tr(const char* tile, int r0, int r1, int ch, int half8) {
  using lds_ptr = __attribute__((address_space(3))) s16x4v*;
  const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + boff(r0, ch) + half8));
  const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + boff(r1, ch) + half8));
  const s16x8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int base) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)a[base + j];
  return v;
}
// rows [row0, row0+nvalid) of a [*, ld] bf16 matrix (64 columns at `src`) -> swizzled LDS tile of ntile rows, zero padded
__device__ __forceinline__ void stage_bf16(char* dst, const __bf16* src, long ld, int row0, int nvalid, int ntile, int tid,
                                           int nthreads) {
  for (int i = tid; i < ntile * 8; i += nthreads) {
    const int row = i >> 3, ch = i & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nvalid) v = *(const uint4*)(src + (long)(row0 + row) * ld + ch * 8);
    *(uint4*)(dst + boff(row, ch)) = v;
  }
}

// The same staging with compile-time trip counts, split into "issue every global load" and "write LDS": all loads of a phase
// are in flight together (one exposed latency per phase instead of one per loop iteration), and the loads of the NEXT query
// chunk are issued before the MFMA phase of the current one.
template <int NTILE, int NTH>
__device__ __forceinline__ void tile_load(uint4 (&v)[NTILE * 8 / NTH], const __bf16* src, long ld, int row0, int nvalid, int tid) {
#pragma unroll
  for (int i = 0; i < NTILE * 8 / NTH; ++i) {
    const int idx = tid + i * NTH, row = idx >> 3, ch = idx & 7;
    v[i] = row < nvalid ? *(const uint4*)(src + (long)(row0 + row) * ld + ch * 8) : make_uint4(0, 0, 0, 0);
  }
}
template <int NTILE, int NTH>
__device__ __forceinline__ void tile_store(char* dst, const uint4 (&v)[NTILE * 8 / NTH], int tid) {
#pragma unroll
  for (int i = 0; i < NTILE * 8 / NTH; ++i) {
    const int idx = tid + i * NTH;
    *(uint4*)(dst + boff(idx >> 3, idx & 7)) = v[i];
  }
}
__device__ __forceinline__ float dot8_bf16(const uint4& a, const uint4& b) {
  const bf16x8 x = __builtin_bit_cast(bf16x8, a), y = __builtin_bit_cast(bf16x8, b);
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) t += (float)x[j] * (float)y[j];
  return t;
}

// forward: block = 4 waves x 32 query rows; grid = (ceil(Sq/128), B*nh)
template <int NKT>
__device__ __forceinline__ void attn_fwd_bf16_body(const AttnP& p, int qblk, int bh) {
  constexpr int SKP = NKT * 32;
  __shared__ __attribute__((aligned(16))) char smem[2 * SKP * 128];
  char* Ks = smem;
  char* Vs = smem + SKP * 128;
  const int b = bh / p.nh, hd = bh % p.nh;
  const int q0 = qblk * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const unsigned dseed = p.drop_thr ? eff_seed(p.drop_seed, p.seed_base) : 0u;
  const int qg = q0 + wave * 32 + r;
  bf16x8 qf[4];
  {
    uint4 kreg[NKT], vreg[NKT];
    tile_load<SKP, 256>(kreg, (const __bf16*)p.k + (long)b * p.Sk * p.ldk + hd * 64, p.ldk, 0, p.Sk, tid);
    tile_load<SKP, 256>(vreg, (const __bf16*)p.v + (long)b * p.Sk * p.ldv + hd * 64, p.ldv, 0, p.Sk, tid);
    const __bf16* qp = (const __bf16*)p.q + ((long)b * p.Sq + min(qg, p.Sq - 1)) * p.ldq + hd * 64 + 8 * hh;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
    tile_store<SKP, 256>(Ks, kreg, tid);
    tile_store<SKP, 256>(Vs, vreg, tid);
  }
  __syncthreads();
  if (q0 + wave * 32 >= p.Sq) return;                 // whole wave beyond the last query row (no barrier after this point)

  f32x16 s[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int x = 0; x < 16; ++x) s[kt][x] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Ks, kt * 32 + r, 2 * kk + hh), qf[kk], s[kt], 0, 0, 0);
  }
  const float* km = p.kmask ? p.kmask + (long)b * p.Sk : nullptr;
  const float* bs = (p.bias && qg < p.Sq) ? p.bias + ((long)b * p.Sq + qg) * p.Sk : nullptr;
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int key = kt * 32 + acc_row(x, hh);
      float v = s[kt][x] * p.scale;
      if (key < p.Sk) {
        if (km) v += km[key];
        if (bs) v += bs[key];
      } else {
        v = -INFINITY;
      }
      s[kt][x] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const float e = __expf(s[kt][x] - m);
      s[kt][x] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  if (p.drop_thr) {                                   // dropout on the probabilities (the row sum above stays un-dropped)
    const unsigned base = (unsigned)((b * p.nh + hd) * p.Sq + qg) * (unsigned)p.Sk;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int x = 0; x < 16; ++x) s[kt][x] *= drop_scale(base + kt * 32 + acc_row(x, hh), dseed, p.drop_thr, p.drop_inv);
  }

  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1;
  __bf16* out = (__bf16*)p.out + ((long)b * p.Sq + qg) * p.ldo + hd * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    f32x16 o;
#pragma unroll
    for (int x = 0; x < 16; ++x) o[x] = 0.f;
    const int ch = dt * 4 + 2 * cb + (pp >> 1), half8 = 8 * (pp & 1);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int r0 = kt * 32 + 16 * s2 + 4 * hh + qq;
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(Vs, r0, r0 + 8, ch, half8), pack8(s[kt], 8 * s2), o, 0, 0, 0);
      }
    if (qg < p.Sq) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
        DT<__bf16>::st4(out + dt * 32 + 8 * g + 4 * hh, v);
      }
    }
  }
  if (p.lse && hh == 0 && qg < p.Sq) p.lse[((long)b * p.nh + hd) * p.Sq + qg] = m + __logf(l);
}
template <int NKT>
__global__ __launch_bounds__(256) void attn_fwd_bf16_kernel(AttnP p) { attn_fwd_bf16_body<NKT>(p, blockIdx.x, blockIdx.y); }

// Two attention problems in ONE launch (the language and the vision stream of a cross-modal layer, or the two directions of its
// bidirectional cross-attention: vilmodel_cmt.py:385-407): each alone is 768 latency-bound blocks that use about a third of the wave
// slots of the chip. Blocks [0, blocks_a) run problem a, the rest problem b; both take the key-tile count of the longer context.
struct AttnP2 { AttnP a, b; int blocks_a; };
template <int NKT>
__global__ __launch_bounds__(256) void attn_fwd_bf16_dual_kernel(AttnP2 pp) {
  const bool second = (int)blockIdx.x >= pp.blocks_a;                    // block-uniform
  const AttnP& p = second ? pp.b : pp.a;
  const int id = (int)blockIdx.x - (second ? pp.blocks_a : 0), nq = (p.Sq + 127) / 128;
  attn_fwd_bf16_body<NKT>(p, id % nq, id / nq);
}

// backward: block = NW waves (4, or 8 for 129..256 keys); wave w owns key tile w; grid = B*nh; query rows in chunks of 64
// BIAS / DROP are compile-time: the per-element bias / dbias addresses and the dropout hash of 16 x 2 elements otherwise stay
// live next to the accumulators and push the kernel to ~470 registers (one 4-wave block per CU).
template <int NKT> constexpr int attn_bwd_lds() { return 2 * NKT * 32 * 128 + 2 * 64 * 128 + 64 * (NKT * 32 * 2 + 16) + 2 * 64 * 4; }
template <int NKT, int NW, bool BIAS, bool DROP>
__device__ __forceinline__ void attn_bwd_bf16_body(const AttnP& p, int bh, char* smem) {
  constexpr int NTH = NW * 64;
  constexpr int SKP = NKT * 32, DSS = SKP * 2 + 16;       // dS row stride in bytes (odd number of 16-B slots)
  char* Ks = smem;
  char* Vs = Ks + SKP * 128;
  char* Qs = Vs + SKP * 128;
  char* dOs = Qs + 64 * 128;
  char* dSs = dOs + 64 * 128;
  float* lse_s = (float*)(dSs + 64 * DSS);
  float* del_s = lse_s + 64;
  const int b = bh / p.nh, hd = bh % p.nh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
  const unsigned dseed = p.drop_thr ? eff_seed(p.drop_seed, p.seed_base) : 0u;
  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1, half8 = 8 * (pp & 1);
  constexpr int QPT = 64 * 8 / NTH;                       // 16-B pieces of a 64-row chunk per thread
  const __bf16* const qsrc = (const __bf16*)p.q + (long)b * p.Sq * p.ldq + hd * 64;
  const __bf16* const dosrc = (const __bf16*)p.dout + (long)b * p.Sq * p.lddo + hd * 64;
  const __bf16* const osrc = (const __bf16*)p.out + (long)b * p.Sq * p.ldo + hd * 64;
  const float* const lsesrc = p.lse + ((long)b * p.nh + hd) * p.Sq;
  uint4 qreg[QPT], doreg[QPT], oreg[QPT];
  float lsereg = 0.f;
  auto chunk_load = [&](int q0) {                          // every global load of one 64-query chunk, nothing waits here
    const int nq = min(64, p.Sq - q0);
    tile_load<64, NTH>(qreg, qsrc, p.ldq, q0, nq, tid);
    tile_load<64, NTH>(doreg, dosrc, p.lddo, q0, nq, tid);
    tile_load<64, NTH>(oreg, osrc, p.ldo, q0, nq, tid);
    lsereg = (tid < nq) ? lsesrc[q0 + tid] : 0.f;
  };
  {
    uint4 kreg[SKP * 8 / NTH], vreg[SKP * 8 / NTH];
    tile_load<SKP, NTH>(kreg, (const __bf16*)p.k + (long)b * p.Sk * p.ldk + hd * 64, p.ldk, 0, p.Sk, tid);
    tile_load<SKP, NTH>(vreg, (const __bf16*)p.v + (long)b * p.Sk * p.ldv + hd * 64, p.ldv, 0, p.Sk, tid);
    chunk_load(0);
    tile_store<SKP, NTH>(Ks, kreg, tid);
    tile_store<SKP, NTH>(Vs, vreg, tid);
  }
  __syncthreads();
  const bool owner = wave < NKT;
  const int key = wave * 32 + r;
  bf16x8 kf[4], vf[4];
  if (owner) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[kk] = ld_row(Ks, key, 2 * kk + hh);
      vf[kk] = ld_row(Vs, key, 2 * kk + hh);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int x = 0; x < 16; ++x) { dk[dt][x] = 0.f; dv[dt][x] = 0.f; }
  const float kmv = (p.kmask && key < p.Sk) ? p.kmask[(long)b * p.Sk + key] : 0.f;

  for (int q0 = 0; q0 < p.Sq; q0 += 64) {
    const int nq = min(64, p.Sq - q0);
    tile_store<64, NTH>(Qs, qreg, tid);
    tile_store<64, NTH>(dOs, doreg, tid);
#pragma unroll
    for (int i = 0; i < QPT; ++i) {                        // delta[q] = <dO[q], O[q]>: the 8 pieces of a row sit in 8 adjacent lanes
      float t = dot8_bf16(doreg[i], oreg[i]);
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      if ((tid & 7) == 0) del_s[(tid + i * NTH) >> 3] = t;   // rows >= nq were loaded as zeros
    }
    if (tid < 64) lse_s[tid] = lsereg;
    __syncthreads();
    if (q0 + 64 < p.Sq) chunk_load(q0 + 64);               // in flight under the MFMA phases below
    if (owner) {
      // one base per chunk + 32-bit row offsets (64-bit per-element addresses cost ~80 registers)
      const float* const bias_b = BIAS ? p.bias + ((long)b * p.Sq + q0) * p.Sk + key : nullptr;
      float* const dbias_b = (BIAS && p.dbias) ? p.dbias + ((long)b * p.Sq + q0) * p.Sk + key : nullptr;
#pragma unroll 1
      for (int qt = 0; qt < 2; ++qt) {
        f32x16 s, dp;
#pragma unroll
        for (int x = 0; x < 16; ++x) { s[x] = 0.f; dp[x] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Qs, qt * 32 + r, 2 * kk + hh), kf[kk], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(dOs, qt * 32 + r, 2 * kk + hh), vf[kk], dp, 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          const int ql = qt * 32 + acc_row(x, hh);
          float pv = 0.f, ds = 0.f, pdrop = 0.f;
          if (key < p.Sk && ql < nq) {
            float sv = s[x] * p.scale + kmv;
            if (BIAS) sv += bias_b[ql * p.Sk];
            pv = __expf(sv - lse_s[ql]);
            float dpx = dp[x];
            if (DROP) {
              const float ms = drop_scale((unsigned)((b * p.nh + hd) * p.Sq + q0 + ql) * (unsigned)p.Sk + key, dseed,
                                          p.drop_thr, p.drop_inv);
              dpx *= ms;                      // dP = dP_dropped * mask / keep
              pdrop = pv * ms;                // dV uses the dropped probabilities
            } else {
              pdrop = pv;
            }
            ds = pv * (dpx - del_s[ql]);
            if (BIAS && dbias_b) atomicAdd(dbias_b + ql * p.Sk, ds);
          }
          s[x] = pdrop;
          dp[x] = ds * p.scale;
          *(__bf16*)(dSs + ql * DSS + key * 2) = (__bf16)dp[x];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pb = pack8(s, 8 * s2), db = pack8(dp, 8 * s2);
          const int r0 = qt * 32 + 16 * s2 + 4 * hh + qq;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int ch = dt * 4 + 2 * cb + (pp >> 1);
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(dOs, r0, r0 + 8, ch, half8), pb, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(Qs, r0, r0 + 8, ch, half8), db, dk[dt], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    if (wave < 4) {
      const int qt = wave >> 1, dt = wave & 1;
      const int ch = dt * 4 + 2 * cb + (pp >> 1);
      f32x16 dq;
#pragma unroll
      for (int x = 0; x < 16; ++x) dq[x] = 0.f;
#pragma unroll
      for (int kk = 0; kk < SKP / 16; ++kk) {
        const int r0 = 16 * kk + 8 * hh + qq;
        const bf16x8 bfrag = *(const bf16x8*)(dSs + (qt * 32 + r) * DSS + (16 * kk + 8 * hh) * 2);
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(Ks, r0, r0 + 4, ch, half8), bfrag, dq, 0, 0, 0);
      }
      const int ql = qt * 32 + r;
      if (ql < nq) {
        __bf16* o = (__bf16*)p.dq + ((long)b * p.Sq + q0 + ql) * p.lddq + hd * 64 + dt * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {dq[4 * g], dq[4 * g + 1], dq[4 * g + 2], dq[4 * g + 3]};
          DT<__bf16>::st4(o + 8 * g + 4 * hh, v);
        }
      }
    }
    __syncthreads();
  }
  if (owner && key < p.Sk) {
    __bf16* ok = (__bf16*)p.dk + ((long)b * p.Sk + key) * p.lddk + hd * 64;
    __bf16* ov = (__bf16*)p.dv + ((long)b * p.Sk + key) * p.lddv + hd * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a = {dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]};
        f32x4 c = {dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]};
        DT<__bf16>::st4(ok + dt * 32 + 8 * g + 4 * hh, a);
        DT<__bf16>::st4(ov + dt * 32 + 8 * g + 4 * hh, c);
      }
  }
}
template <int NKT, int NW, bool BIAS, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_bf16_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) char smem[attn_bwd_lds<NKT>()];
  attn_bwd_bf16_body<NKT, NW, BIAS, DROP>(p, blockIdx.x, smem);
}
// dual form (see attn_fwd_bf16_dual_kernel); an additive bias is supported on problem a only (DUET's graph_sprels stream)
template <int NKT, int NW, bool BIAS_A, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_bf16_dual_kernel(AttnP2 pp) {
  __shared__ __attribute__((aligned(16))) char smem[attn_bwd_lds<NKT>()];      // ONE array for both bodies
  if ((int)blockIdx.x < pp.blocks_a) attn_bwd_bf16_body<NKT, NW, BIAS_A, DROP>(pp.a, blockIdx.x, smem);
  else attn_bwd_bf16_body<NKT, NW, false, DROP>(pp.b, (int)blockIdx.x - pp.blocks_a, smem);
}

template <int NKT>
int launch_bf16_dual(const AttnP2& pp, bool bwd, hipStream_t st) {
  constexpr int NW = NKT <= 4 ? 4 : 8;
  if (bwd) {
    const bool bias = pp.a.bias != nullptr, drop = pp.a.drop_thr != 0;
    const int blocks = pp.blocks_a + pp.b.B * pp.b.nh;
    if (bias && drop) hipLaunchKernelGGL((attn_bwd_bf16_dual_kernel<NKT, NW, true, true>), dim3(blocks), dim3(NW * 64), 0, st, pp);
    else if (bias) hipLaunchKernelGGL((attn_bwd_bf16_dual_kernel<NKT, NW, true, false>), dim3(blocks), dim3(NW * 64), 0, st, pp);
    else if (drop) hipLaunchKernelGGL((attn_bwd_bf16_dual_kernel<NKT, NW, false, true>), dim3(blocks), dim3(NW * 64), 0, st, pp);
    else hipLaunchKernelGGL((attn_bwd_bf16_dual_kernel<NKT, NW, false, false>), dim3(blocks), dim3(NW * 64), 0, st, pp);
  } else {
    const int blocks = pp.blocks_a + cdiv(pp.b.Sq, 128) * pp.b.B * pp.b.nh;
    hipLaunchKernelGGL((attn_fwd_bf16_dual_kernel<NKT>), dim3(blocks), dim3(256), 0, st, pp);
  }
  return 0;
}
int dispatch_bf16_dual(const AttnP2& pp, bool bwd, hipStream_t st) {
  switch (cdiv(pp.a.Sk > pp.b.Sk ? pp.a.Sk : pp.b.Sk, 32)) {
    case 1: return launch_bf16_dual<1>(pp, bwd, st);
    case 2: return launch_bf16_dual<2>(pp, bwd, st);
    case 3: return launch_bf16_dual<3>(pp, bwd, st);
    case 4: return launch_bf16_dual<4>(pp, bwd, st);
    case 5: case 6: return launch_bf16_dual<6>(pp, bwd, st);
    case 7: case 8: return launch_bf16_dual<8>(pp, bwd, st);
  }
  return -1;
}

template <int NKT>
int launch_bf16(const AttnP& p, bool bwd, hipStream_t st) {
  constexpr int NW = NKT <= 4 ? 4 : 8;
  if (bwd) {
    const bool bias = p.bias != nullptr || p.dbias != nullptr, drop = p.drop_thr != 0;
    if (bias && p.bias == nullptr) return -1;
    if (bias && drop) hipLaunchKernelGGL((attn_bwd_bf16_kernel<NKT, NW, true, true>), dim3(p.B * p.nh), dim3(NW * 64), 0, st, p);
    else if (bias) hipLaunchKernelGGL((attn_bwd_bf16_kernel<NKT, NW, true, false>), dim3(p.B * p.nh), dim3(NW * 64), 0, st, p);
    else if (drop) hipLaunchKernelGGL((attn_bwd_bf16_kernel<NKT, NW, false, true>), dim3(p.B * p.nh), dim3(NW * 64), 0, st, p);
    else hipLaunchKernelGGL((attn_bwd_bf16_kernel<NKT, NW, false, false>), dim3(p.B * p.nh), dim3(NW * 64), 0, st, p);
  }
  else hipLaunchKernelGGL((attn_fwd_bf16_kernel<NKT>), dim3(cdiv(p.Sq, 128), p.B * p.nh), dim3(256), 0, st, p);
  return 0;
}
int dispatch_bf16(const AttnP& p, bool bwd, hipStream_t st) {
  switch (cdiv(p.Sk, 32)) {
    case 1: return launch_bf16<1>(p, bwd, st);
    case 2: return launch_bf16<2>(p, bwd, st);
    case 3: return launch_bf16<3>(p, bwd, st);
    case 4: return launch_bf16<4>(p, bwd, st);
    case 5: case 6: return launch_bf16<6>(p, bwd, st);      // 129..192 keys (8 waves in the backward pass)
    case 7: case 8: return launch_bf16<8>(p, bwd, st);      // 193..256 keys
  }
  return -1;
}

template <typename T, int NKT>
int launch_fwd(const AttnP& p, hipStream_t st) {
  const size_t lds = (size_t)(2 * NKT * 32 * LD + 64 * LD) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<T, NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_fwd_kernel<T, NKT>), dim3(cdiv(p.Sq, 64), p.B * p.nh), dim3(128), lds, st, p);
  return 0;
}
template <typename T, int NKT>
int launch_bwd(const AttnP& p, hipStream_t st) {
  const size_t lds = (size_t)(2 * NKT * 32 * LD + 2 * 64 * LD + 64 * (NKT * 32 + 1) + 128) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<T, NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_bwd_kernel<T, NKT>), dim3(p.B * p.nh), dim3(256), lds, st, p);
  return 0;
}

template <typename T>
int dispatch(const AttnP& p, bool bwd, hipStream_t st) {
  const int nkt = cdiv(p.Sk, 32);
  switch (nkt) {
    case 1: return bwd ? launch_bwd<T, 1>(p, st) : launch_fwd<T, 1>(p, st);
    case 2: return bwd ? launch_bwd<T, 2>(p, st) : launch_fwd<T, 2>(p, st);
    case 3: return bwd ? launch_bwd<T, 3>(p, st) : launch_fwd<T, 3>(p, st);
    case 4: return bwd ? launch_bwd<T, 4>(p, st) : launch_fwd<T, 4>(p, st);
  }
  return -1;
}

int check_common(const char* who, int dtype, const AttnP& p) {
  VLNI_CHECK(dtype == VLNI_F32 || dtype == VLNI_BF16, VLNI_EINVAL, "%s: bad dtype %d", who, dtype);
  VLNI_CHECK(p.B > 0 && p.nh > 0 && p.Sq > 0 && p.Sk > 0, VLNI_EINVAL, "%s: empty problem", who);
  VLNI_CHECK(p.Sk <= (dtype == VLNI_BF16 ? 256 : 128), VLNI_EUNSUP, "%s: Sk=%d keys not covered (bf16 <= 256, fp32 <= 128)", who, p.Sk);
  VLNI_CHECK(p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldv % 4 == 0 && p.ldo % 4 == 0, VLNI_EINVAL,
             "%s: row strides must be multiples of 4 elements", who);
  VLNI_CHECK(p.ldq >= p.nh * 64 && p.ldk >= p.nh * 64 && p.ldv >= p.nh * 64 && p.ldo >= p.nh * 64, VLNI_EINVAL,
             "%s: row strides smaller than nh*64", who);
  return VLNI_OK;
}

}  // namespace

extern "C" int vlni_attn_fwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                             const float* kmask, const float* bias, void* out, long ldo, float* lse, int B, int nh, int Sq,
                             int Sk, float scale, float drop_p, unsigned drop_seed, void* stream) {
  AttnP p = {};
  p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.kmask = kmask; p.bias = bias;
  p.out = out; p.ldo = ldo; p.lse = lse; p.B = B; p.nh = nh; p.Sq = Sq; p.Sk = Sk; p.scale = scale;
  p.drop_thr = drop_thr(drop_p); p.drop_seed = drop_seed; p.drop_inv = 1.0f / (1.0f - drop_p); p.seed_base = vlni_seed_base();
  int rc = check_common("attn_fwd", dtype, p);
  if (rc) return rc;
  static const bool f32mfma = getenv("VLNI_ATTN_F32MFMA") != nullptr;
  const bool fast = dtype == VLNI_BF16 && !f32mfma && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 &&
                    (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0;
  rc = fast ? dispatch_bf16(p, false, (hipStream_t)stream)
            : (dtype == VLNI_F32 ? dispatch<float>(p, false, (hipStream_t)stream) : dispatch<__bf16>(p, false, (hipStream_t)stream));
  VLNI_CHECK(rc == 0, VLNI_EUNSUP, "attn_fwd: no kernel for Sk=%d", Sk);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_attn_bwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                             const float* kmask, const float* bias, const void* out, long ldo, const void* dout, long lddo,
                             const float* lse, void* dq, long lddq, void* dk, long lddk, void* dv, long lddv, float* dbias,
                             int B, int nh, int Sq, int Sk, float scale, float drop_p, unsigned drop_seed, void* stream) {
  AttnP p = {};
  p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.kmask = kmask; p.bias = bias;
  p.out = (void*)out; p.ldo = ldo; p.lse = (float*)lse; p.B = B; p.nh = nh; p.Sq = Sq; p.Sk = Sk; p.scale = scale;
  p.dout = dout; p.lddo = lddo; p.dq = dq; p.dk = dk; p.dv = dv; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv; p.dbias = dbias;
  p.drop_thr = drop_thr(drop_p); p.drop_seed = drop_seed; p.drop_inv = 1.0f / (1.0f - drop_p); p.seed_base = vlni_seed_base();
  int rc = check_common("attn_bwd", dtype, p);
  if (rc) return rc;
  VLNI_CHECK(lddo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0, VLNI_EINVAL, "attn_bwd: grad strides");
  VLNI_CHECK(lse != nullptr, VLNI_EINVAL, "attn_bwd: lse required");
  static const bool f32mfma = getenv("VLNI_ATTN_F32MFMA") != nullptr;
  const bool fast = dtype == VLNI_BF16 && !f32mfma && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 &&
                    (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout) & 15) == 0;
  rc = fast ? dispatch_bf16(p, true, (hipStream_t)stream)
            : (dtype == VLNI_F32 ? dispatch<float>(p, true, (hipStream_t)stream) : dispatch<__bf16>(p, true, (hipStream_t)stream));
  VLNI_CHECK(rc == 0, VLNI_EUNSUP, "attn_bwd: no kernel for Sk=%d", Sk);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// Two attention problems (arrays of 2; same batch, heads and scale) in one launch: bfloat16 only, 16-byte aligned q/k/v (and dout),
// row strides multiples of 8, <= 256 keys each, an additive bias on problem 0 only. VLNI_EUNSUP otherwise (the caller launches twice).
static int attn_fill(AttnP& p, int i, const void* const* q, const long* ldq, const void* const* k, const long* ldk, const void* const* v,
                     const long* ldv, const float* const* kmask, const float* const* bias, void* const* out, const long* ldo,
                     float* const* lse, int B, int nh, const int* Sq, const int* Sk, float scale, float drop_p, const unsigned* drop_seed) {
  p = AttnP{};
  p.q = q[i]; p.k = k[i]; p.v = v[i]; p.ldq = ldq[i]; p.ldk = ldk[i]; p.ldv = ldv[i];
  p.kmask = kmask ? kmask[i] : nullptr; p.bias = bias ? bias[i] : nullptr;
  p.out = out[i]; p.ldo = ldo[i]; p.lse = lse[i]; p.B = B; p.nh = nh; p.Sq = Sq[i]; p.Sk = Sk[i]; p.scale = scale;
  p.drop_thr = drop_thr(drop_p); p.drop_seed = drop_seed ? drop_seed[i] : 0; p.drop_inv = 1.0f / (1.0f - drop_p); p.seed_base = vlni_seed_base();
  return check_common("attn_dual", VLNI_BF16, p);
}
static bool attn_fast_ok(const AttnP& p, bool bwd) {
  return p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && (!bwd || p.lddo % 8 == 0) &&
         (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (bwd ? (uintptr_t)p.dout : 0)) & 15) == 0;
}

extern "C" int vlni_attn_fwd_dual(int dtype, const void* const* q, const long* ldq, const void* const* k, const long* ldk,
                                  const void* const* v, const long* ldv, const float* const* kmask, const float* const* bias,
                                  void* const* out, const long* ldo, float* const* lse, int B, int nh, const int* Sq, const int* Sk,
                                  float scale, float drop_p, const unsigned* drop_seed, void* stream) {
  VLNI_CHECK(dtype == VLNI_BF16, VLNI_EUNSUP, "attn_fwd_dual: bfloat16 only (dtype %d)", dtype);
  AttnP2 pp;
  for (int i = 0; i < 2; ++i) {
    int rc = attn_fill(i ? pp.b : pp.a, i, q, ldq, k, ldk, v, ldv, kmask, bias, out, ldo, lse, B, nh, Sq, Sk, scale, drop_p, drop_seed);
    if (rc) return rc;
  }
  VLNI_CHECK(pp.b.bias == nullptr, VLNI_EUNSUP, "attn_fwd_dual: additive bias on problem 0 only");
  VLNI_CHECK(attn_fast_ok(pp.a, false) && attn_fast_ok(pp.b, false), VLNI_EUNSUP, "attn_fwd_dual: operands not 16-byte aligned / strides not multiples of 8");
  pp.blocks_a = cdiv(pp.a.Sq, 128) * B * nh;
  VLNI_CHECK(dispatch_bf16_dual(pp, false, (hipStream_t)stream) == 0, VLNI_EUNSUP, "attn_fwd_dual: no kernel for Sk=%d/%d", Sk[0], Sk[1]);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_attn_bwd_dual(int dtype, const void* const* q, const long* ldq, const void* const* k, const long* ldk,
                                  const void* const* v, const long* ldv, const float* const* kmask, const float* const* bias,
                                  const void* const* out, const long* ldo, const void* const* dout, const long* lddo,
                                  const float* const* lse, void* const* dq, const long* lddq, void* const* dk, const long* lddk,
                                  void* const* dv, const long* lddv, float* dbias0, int B, int nh, const int* Sq, const int* Sk,
                                  float scale, float drop_p, const unsigned* drop_seed, void* stream) {
  VLNI_CHECK(dtype == VLNI_BF16, VLNI_EUNSUP, "attn_bwd_dual: bfloat16 only (dtype %d)", dtype);
  AttnP2 pp;
  for (int i = 0; i < 2; ++i) {
    AttnP& p = i ? pp.b : pp.a;
    int rc = attn_fill(p, i, q, ldq, k, ldk, v, ldv, kmask, bias, (void* const*)out, ldo, (float* const*)lse, B, nh, Sq, Sk, scale, drop_p,
                       drop_seed);
    if (rc) return rc;
    p.dout = dout[i]; p.lddo = lddo[i]; p.dq = dq[i]; p.dk = dk[i]; p.dv = dv[i]; p.lddq = lddq[i]; p.lddk = lddk[i]; p.lddv = lddv[i];
    VLNI_CHECK(p.lddo % 4 == 0 && p.lddq % 4 == 0 && p.lddk % 4 == 0 && p.lddv % 4 == 0 && p.lse != nullptr, VLNI_EINVAL, "attn_bwd_dual: grad strides / lse");
  }
  pp.a.dbias = dbias0;
  VLNI_CHECK(pp.b.bias == nullptr && (dbias0 == nullptr || pp.a.bias != nullptr), VLNI_EUNSUP, "attn_bwd_dual: additive bias on problem 0 only");
  VLNI_CHECK(attn_fast_ok(pp.a, true) && attn_fast_ok(pp.b, true), VLNI_EUNSUP, "attn_bwd_dual: operands not 16-byte aligned / strides not multiples of 8");
  pp.blocks_a = B * nh;
  VLNI_CHECK(dispatch_bf16_dual(pp, true, (hipStream_t)stream) == 0, VLNI_EUNSUP, "attn_bwd_dual: no kernel for Sk=%d/%d", Sk[0], Sk[1]);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

// ------------------------------------------------------------------------------------------------
// Attention PROBABILITIES materialised for visualisation (NavCMT.forward(..., return_cross_attention_probs=True): the reference's
// LXRTXLayer returns softmax(raw scores) of its four attentions, vilmodel_cmt.py:391,393,438,439). Never on the training path (the
// fused kernels above keep the scores on chip); one wave per (batch, head, query row), keys strided over the lanes.
namespace {
template <typename T>
__global__ __launch_bounds__(64) void attn_probs_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ k, long ldk,
                                                        const float* __restrict__ kmask, const float* __restrict__ bias,
                                                        float* __restrict__ P, int nh, int Sq, int Sk, float scale) {
  __shared__ float qs[64];
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  qs[lane] = DT<T>::ld(q + ((long)b * Sq + i) * ldq + h * 64 + lane);
  __syncthreads();
  float sc[8];                                              // Sk <= 512
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = lane + 64 * u;
    sc[u] = -INFINITY;
    if (j < Sk) {
      const T* kr = k + ((long)b * Sk + j) * ldk + h * 64;
      float a = 0.f;
      for (int d = 0; d < 64; d += 4) {
        const f32x4 kv = DT<T>::ld4(kr + d);
        a += qs[d] * kv[0] + qs[d + 1] * kv[1] + qs[d + 2] * kv[2] + qs[d + 3] * kv[3];
      }
      a = a * scale + (kmask ? kmask[(long)b * Sk + j] : 0.f) + (bias ? bias[((long)b * Sq + i) * Sk + j] : 0.f);
      sc[u] = a;
      mx = fmaxf(mx, a);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (lane + 64 * u < Sk) { sc[u] = expf(sc[u] - mx); sum += sc[u]; }
  }
  sum = wave_sum(sum);
  float* o = P + (((long)b * nh + h) * Sq + i) * Sk;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (lane + 64 * u < Sk) o[lane + 64 * u] = sc[u] / sum;
}
}  // namespace

extern "C" int vlni_attn_probs(int dtype, const void* q, long ldq, const void* k, long ldk, const float* kmask, const float* bias,
                               float* probs, int B, int nh, int Sq, int Sk, float scale, void* stream) {
  VLNI_CHECK(q && k && probs && B > 0 && nh > 0 && Sq > 0 && Sk > 0 && Sk <= 512, VLNI_EINVAL, "attn_probs: B=%d nh=%d Sq=%d Sk=%d (Sk <= 512)",
             B, nh, Sq, Sk);
  VLNI_CHECK(ldq >= nh * 64 && ldk >= nh * 64 && ldk % 4 == 0 && ((uintptr_t)k & 15) == 0, VLNI_EINVAL, "attn_probs: strides / alignment");
  VLNI_CHECK(dtype == VLNI_F32 || dtype == VLNI_BF16, VLNI_EINVAL, "attn_probs: dtype %d", dtype);
  dim3 grid(Sq, nh, B);
  if (dtype == VLNI_F32)
    hipLaunchKernelGGL((attn_probs_kernel<float>), grid, dim3(64), 0, (hipStream_t)stream, (const float*)q, ldq, (const float*)k, ldk, kmask,
                       bias, probs, nh, Sq, Sk, scale);
  else
    hipLaunchKernelGGL((attn_probs_kernel<__bf16>), grid, dim3(64), 0, (hipStream_t)stream, (const __bf16*)q, ldq, (const __bf16*)k, ldk,
                       kmask, bias, probs, nh, Sq, Sk, scale);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
