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
// What bounds them on the K = 768 shapes of this workload is the CU's vector-memory pipe (~66 GB/s per CU), see DESIGN.md section 6.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;     // ROWB: bytes of K per tile row
constexpr int NT = 256;

struct GemmP {
  const char* A; long lda;      // byte pointers; ld in ELEMENTS
  const char* B; long ldb;
  char* C; long ldc;
  int M, N, K;
  const float* bias;            // [N] or null
  int act;                      // 0 none, 1 gelu(erf), 2 relu
  const char* residual; long ldr;   // added after act
  char* preact; long ldp;       // value before act (after bias), same dtype as C
  const char* dact_src; long ldd; int dact;   // multiply by act'(src) (1 gelu', 2 relu')
  float alpha;
  unsigned drop_thr, drop_seed; float drop_inv; const unsigned* seed_base;   // effective seed = drop_seed + *seed_base   // dropout on the epilogue value (after act / act'), before the residual
  int atomic_f32;               // C is float, accumulate with atomics (split-K wgrad)
  int kt_per_split;
  int vec_ok;                   // every epilogue tensor allows 4-element vector accesses
  int group_m;                  // tile rasterisation: m-tiles per group (see tile_origin)
  // optional SECOND problem with the same N, K and epilogue kind (the language / vision streams of a cross-modal
  // layer): tiles [0, tiles0) belong to problem 0, the rest to problem 1 -> one launch fills the chip instead of two tails
  int tiles0;
  const char* A1; long lda1;
  const char* B1; long ldb1;
  char* C1; long ldc1;
  int M1;
  const float* bias1;
  const char* residual1; long ldr1;
  char* preact1; long ldp1;
  const char* dact_src1; long ldd1;
  unsigned drop_seed1;
};

// tile -> (problem, tile inside the problem); rewrites the by-value parameter block for problem 1 (block-uniform)
__device__ __forceinline__ void select_problem(GemmP& p, int& wgid) {
  if (wgid >= p.tiles0) {
    wgid -= p.tiles0;
    p.A = p.A1; p.lda = p.lda1; p.B = p.B1; p.ldb = p.ldb1; p.C = p.C1; p.ldc = p.ldc1; p.M = p.M1;
    p.bias = p.bias1; p.residual = p.residual1; p.ldr = p.ldr1; p.preact = p.preact1; p.ldp = p.ldp1;
    p.dact_src = p.dact_src1; p.ldd = p.ldd1; p.drop_seed = p.drop_seed1;
  }
}

// Tile order inside one problem: groups of `group_m` m-tiles; inside a group the m-tile index runs fastest, then the n-tile.
// The 64 tiles an XCD works on at one time (32 CUs x 2 blocks) then span ~8 A row-panels x ~8 B row-panels (~3 MB) instead of
// 3 A panels x EVERY B panel (> the 4 MiB L2 at N = 3072): operands are re-read from the XCD's L2, not from the Infinity Cache.
__device__ __forceinline__ void tile_origin(const GemmP& p, int wgid, int tbm, int tbn, int& m0, int& n0) {
  const int ntn = (p.N + tbn - 1) / tbn;
  if (p.group_m <= 1) { m0 = (wgid / ntn) * tbm; n0 = (wgid % ntn) * tbn; return; }
  const int ntm = (p.M + tbm - 1) / tbm;
  const int per = p.group_m * ntn;
  const int g = wgid / per, idx = wgid - g * per;
  const int first = g * p.group_m, gsz = min(ntm - first, p.group_m);
  m0 = (first + idx % gsz) * tbm;
  n0 = (idx / gsz) * tbn;
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T> struct Mma;
template <> struct Mma<__bf16> {
  static constexpr int KSTEPS = 4;   // 4 x k16 per 128-byte tile row
  template <int NJ>
  __device__ static __forceinline__ void step(const char* As, const char* Bs, int kk, int rowA0, int rowB0, int r, int h,
                                              f32x16 (&acc)[2][NJ]) {
    const int chunk = kk * 2 + h;
    bf16x8 a[2], b[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8*)(As + lds_off(rowA0 + i * 32 + r, chunk));
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = *(const bf16x8*)(Bs + lds_off(rowB0 + j * 32 + r, chunk));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KSTEPS = 16;  // 16 x k2 per 128-byte tile row (BK = 32 floats)
  template <int NJ>
  __device__ static __forceinline__ void step(const char* As, const char* Bs, int kk, int rowA0, int rowB0, int r, int h,
                                              f32x16 (&acc)[2][NJ]) {
    const int k = kk * 2 + h;
    const int chunk = k >> 2, within = (k & 3) * 4;
    float a[2], b[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *(const float*)(As + lds_off(rowA0 + i * 32 + r, chunk) + within);
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = *(const float*)(Bs + lds_off(rowB0 + j * 32 + r, chunk) + within);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};

// LDS-only workgroup barrier for the epilogues: __syncthreads() fences, and in a kernel that uses LDS-DMA hipcc turns that fence into
// s_waitcnt vmcnt(0) - i.e. every barrier would also wait for the round trip of the global STORES issued just before it.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// One 64-row slab of the tile, staged in LDS as float32 [64][TBN_]: every thread finishes 4 consecutive columns of NPASS rows.
// Per group of <= 4 passes the order is: all global loads (GELU' source, residual) -> all LDS reads -> math -> all stores, so no
// store ever sits between a load and its use and no LDS read waits for a store round trip (measured: 8.4 k -> ~3 k cycles per tile).
template <typename T, int TBN_, int NTH, int GPMAX = 4>
__device__ __forceinline__ void epilogue_slab(const GemmP& p, const float* e, int row0, int n0, int tid, const f32x4& bv, unsigned dseed) {
  constexpr int TPR = TBN_ / 4;                                  // threads per row
  constexpr int RPP0 = NTH / TPR, RPP = RPP0 >= 16 ? 16 : RPP0 >= 8 ? 8 : RPP0 >= 4 ? 4 : 2;    // rows per pass: a power of two dividing 64
  constexpr int NPASS = 64 / RPP, GP = NPASS < GPMAX ? NPASS : GPMAX;
  const int c4 = (tid % TPR) * 4, gcol = n0 + c4, rl0 = tid / TPR;
  if (gcol >= p.N || rl0 >= RPP) return;                         // (192-column tiles leave 512 - 48 * 8 threads without a row)
  if (p.vec_ok) {
#pragma unroll
    for (int g = 0; g < NPASS; g += GP) {
      f32x4 z[GP], rs[GP], v[GP];
      if (p.dact) {
#pragma unroll
        for (int k = 0; k < GP; ++k) {
          const int grow = min(row0 + rl0 + RPP * (g + k), p.M - 1);
          z[k] = DT<T>::ld4((const T*)p.dact_src + (long)grow * p.ldd + gcol);
        }
      }
      if (p.residual) {
#pragma unroll
        for (int k = 0; k < GP; ++k) {
          const int grow = min(row0 + rl0 + RPP * (g + k), p.M - 1);
          rs[k] = DT<T>::ld4((const T*)p.residual + (long)grow * p.ldr + gcol);
        }
      }
#pragma unroll
      for (int k = 0; k < GP; ++k) v[k] = *(const f32x4*)(e + (rl0 + RPP * (g + k)) * TBN_ + c4);
      // all the math first (it consumes every loaded register), then nothing but stores: no wait of any kind between the stores
      f32x4 pre[GP];
#pragma unroll
      for (int k = 0; k < GP; ++k) {
        const int grow = row0 + rl0 + RPP * (g + k);
        f32x4 w = v[k];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = w[u] * p.alpha + bv[u];
        pre[k] = w;
        if (p.dact) {
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] *= (p.dact == 1) ? gelu_grad_t<T>(z[k][u]) : (z[k][u] > 0.f ? 1.f : 0.f);
        }
        if (p.act == 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = gelu_t<T>(w[u]);
        } else if (p.act == 2) {
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = fmaxf(w[u], 0.f);
        }
        if (p.drop_thr) {
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] *= drop_scale((unsigned)grow * (unsigned)p.N + gcol + u, dseed, p.drop_thr, p.drop_inv);
        }
        if (p.residual) w += rs[k];
        v[k] = w;
      }
#pragma unroll
      for (int k = 0; k < GP; ++k) {
        const int grow = row0 + rl0 + RPP * (g + k);
        if (grow >= p.M) continue;
        if (p.preact) DT<T>::st4((T*)p.preact + (long)grow * p.ldp + gcol, pre[k]);
        DT<T>::st4((T*)p.C + (long)grow * p.ldc + gcol, v[k]);
      }
    }
  } else {
    for (int k = 0; k < NPASS; ++k) {
      const int rl = rl0 + RPP * k, grow = row0 + rl;
      if (grow >= p.M) continue;
      const f32x4 v = *(const f32x4*)(e + rl * TBN_ + c4);
      for (int u = 0; u < 4 && gcol + u < p.N; ++u) {
        float w = v[u] * p.alpha + bv[u];
        const long col = gcol + u;
        if (p.preact) DT<T>::st((T*)p.preact + (long)grow * p.ldp + col, w);
        if (p.dact) {
          const float zz = DT<T>::ld((const T*)p.dact_src + (long)grow * p.ldd + col);
          w *= (p.dact == 1) ? gelu_grad_t<T>(zz) : (zz > 0.f ? 1.f : 0.f);
        }
        if (p.act == 1) w = gelu_t<T>(w);
        else if (p.act == 2) w = fmaxf(w, 0.f);
        if (p.drop_thr) w *= drop_scale((unsigned)grow * (unsigned)p.N + (unsigned)col, dseed, p.drop_thr, p.drop_inv);
        if (p.residual) w += DT<T>::ld((const T*)p.residual + (long)grow * p.ldr + col);
        DT<T>::st((T*)p.C + (long)grow * p.ldc + col, w);
      }
    }
  }
}

template <typename T, int NW>      // NW waves: 2 (M) x NW/2 (N); a wave owns 64 rows x (256/NW) columns
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, char* smem, f32x16 (&acc)[2][NW == 4 ? 2 : 1], int m0, int n0,
                                              int tid, int wr, int wc, int r, int h) {
  constexpr int NJ = NW == 4 ? 2 : 1, WCOLS = 32 * NJ, RPP = NW * 2, NPASS = 64 / RPP;   // rows per read pass, passes per half
  // ---- split-K / wgrad: float32 atomics straight from the accumulators (128 contiguous bytes per half-wave) ----
  if (p.atomic_f32) {
    float* Cf = (float*)p.C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wc * WCOLS + j * 32 + r;
      if (col >= p.N) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          const int row = m0 + wr * 64 + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
          if (row < p.M) atomicAdd(&Cf[(long)row * p.ldc + col], acc[i][j][x] * p.alpha);
        }
    }
    return;
  }

  // ---- epilogue through LDS: accumulators (col = lane&31, row = (x&3)+8*(x>>2)+4*(lane>>5)) are laid out as a
  //      [64][128] float32 half-tile, then every thread handles 4 consecutive columns of a row: bias / act' / act /
  //      residual with 16-B (8-B bf16) global accesses, 32 threads per 128-column row (full cache lines).
  float* const e = (float*)smem;
  const unsigned dseed = p.drop_thr ? eff_seed(p.drop_seed, p.seed_base) : 0u;
  const int gcol = n0 + (tid & 31) * 4;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && gcol < p.N) {
    if (p.vec_ok) bv = *(const f32x4*)(p.bias + gcol);
    else
      for (int u = 0; u < 4; ++u) bv[u] = (gcol + u < p.N) ? p.bias[gcol + u] : 0.f;
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wr == half) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int x = 0; x < 16; ++x)
            e[(i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h) * BN + wc * WCOLS + j * 32 + r] = acc[i][j][x];
    }
    lds_barrier();
    epilogue_slab<T, BN, NW * 64>(p, e, m0 + half * 64, n0, tid, bv, dseed);
    lds_barrier();
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void gemm_nt_kernel(GemmP p) {
  constexpr int ES = sizeof(T);
  constexpr int BK = ROWB / ES;        // elements of K per tile
  constexpr int EPC = 16 / ES;         // elements per 16-B chunk
  // ONE 32-KiB stage (A tile then B tile); the next tile waits in registers. 32 KiB/block + <=128 VGPRs
  // keeps 4 blocks (16 waves) resident per CU so that blocks hide each other's load latency and epilogues.
  __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * ROWB];
  char* const As = smem;
  char* const Bs = smem + BM * ROWB;

  // ---- tile assignment: contiguous chunk of the tile list per XCD (bijective for any grid) ----
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  select_problem(p, wgid);
  int m0, n0;
  tile_origin(p, wgid, BM, BN, m0, n0);

  const int nkt = (p.K + BK - 1) / BK;
  const int kt0 = blockIdx.z * p.kt_per_split;
  const int kt1 = min(nkt, kt0 + p.kt_per_split);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int lc = tid & 7, lr = tid >> 3;       // staging: chunk column, first row

  // per-thread global row pointers (rows clamped: out-of-range rows load valid memory, never stored)
  const char* ga[4];
  const char* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = min(m0 + lr + 32 * i, p.M - 1), rb = min(n0 + lr + 32 * i, p.N - 1);
    ga[i] = p.A + ((long)ra * p.lda + lc * EPC) * ES;
    gb[i] = p.B + ((long)rb * p.ldb + lc * EPC) * ES;
  }
  uint4 sa[4], sb[4];
  auto gload = [&](int kt) {
    const long koff = (long)kt * BK;
    const bool ok = (koff + lc * EPC) < p.K;   // K is a multiple of EPC: a chunk is all-in or all-out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sa[i] = ok ? *(const uint4*)(ga[i] + koff * ES) : make_uint4(0, 0, 0, 0);
      sb[i] = ok ? *(const uint4*)(gb[i] + koff * ES) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(As + lds_off(lr + 32 * i, lc)) = sa[i];
      *(uint4*)(Bs + lds_off(lr + 32 * i, lc)) = sb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

  if (kt0 < kt1) {
    gload(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
      lstore();                                      // registers -> LDS (tile kt)
      __syncthreads();
      if (kt + 1 < kt1) gload(kt + 1);               // issue early: in flight under the MFMAs below
#pragma unroll
      for (int kk = 0; kk < Mma<T>::KSTEPS; ++kk) Mma<T>::template step<2>(As, Bs, kk, wr * 64, wc * 64, r, h, acc);
      __syncthreads();                               // every wave done reading before the stage is rewritten
    }
  }

  gemm_epilogue<T, 4>(p, smem, acc, m0, n0, tid, wr, wc, r, h);
}

// ------------------------------------------------------------------------------------------------
// Same contraction with a DEEPER pipeline: tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR
// staging, no ds_write), three 32-KiB stages, two k-tiles in flight under the MFMAs of the current one, ONE raw
// s_barrier per k-tile behind a counted s_waitcnt vmcnt(8) (8 = LDS-DMA instructions per thread per tile; never
// __syncthreads(), which would drain the queue). The LDS image is lane-linear per wave-instruction (1 KiB = 8 rows
// of 128 B), so the XOR swizzle is applied to the per-lane SOURCE chunk; reads use the same lds_off().
// Used when K is a multiple of the k-tile and rows need no zero fill (M/N edges are clamped, never stored).
template <typename T, int NST, int NW>     // NST stages = NST-1 k-tiles in flight (3: 96 KiB, 2: 64 KiB); NW = 4 or 8 waves
__global__ __launch_bounds__(NW * 64) void gemm_nt_glds_kernel(GemmP p) {
  constexpr int ES = sizeof(T);
  constexpr int WC = NW / 2, NJ = NW == 4 ? 2 : 1, WCOLS = 32 * NJ, IPW = 16 / NW;   // LDS-DMA instructions per operand per wave
  constexpr int STAGE = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  select_problem(p, wgid);
  int m0, n0;
  tile_origin(p, wgid, BM, BN, m0, n0);
  constexpr int BK = ROWB / ES;
  const int nkt = p.K / BK;
  const int kt0 = blockIdx.z * p.kt_per_split;
  const int nk = min(nkt, kt0 + p.kt_per_split) - kt0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC, r = lane & 31, h = lane >> 5;

  const char* ga[IPW];
  const char* gb[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);                       // logical chunk this LDS slot must hold
    ga[i] = p.A + ((long)min(m0 + row, p.M - 1) * p.lda) * ES + c * 16 + (long)kt0 * ROWB;
    gb[i] = p.B + ((long)min(n0 + row, p.N - 1) * p.ldb) * ES + c * 16 + (long)kt0 * ROWB;
  }
  using gptr = const __attribute__((address_space(1))) void*;
  using lptr = __attribute__((address_space(3))) void*;
  auto issue = [&](int t, int stage) {
    char* sa = dsmem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      __builtin_amdgcn_global_load_lds((gptr)(ga[i] + (long)t * ROWB), (lptr)(sa + i * NW * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr)(gb[i] + (long)t * ROWB), (lptr)(sa + BM * ROWB + i * NW * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

  if (nk > 0) {
    issue(0, 0);
    if (NST == 3 && nk > 1) issue(1, 1);
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
      if (NST == 3 && t + 1 < nk) {                  // tile t landed (this wave's part); tile t+1 may still fly
        if (NW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                                       // every wave's part of tile t landed; stage (t-1)%NST is free
      if (t + NST - 1 < nk) issue(t + NST - 1, stage == 0 ? NST - 1 : stage - 1);   // (t+NST-1)%NST == (t-1)%NST
      const char* As = dsmem + stage * STAGE;
#pragma unroll
      for (int kk = 0; kk < Mma<T>::KSTEPS; ++kk) Mma<T>::template step<NJ>(As, As + BM * ROWB, kk, wr * 64, wc * WCOLS, r, h, acc);
      stage = stage == NST - 1 ? 0 : stage + 1;
    }
  }
  __syncthreads();                                                        // all reads done before the epilogue reuses stage 0
  gemm_epilogue<T, NW>(p, dsmem, acc, m0, n0, tid, wr, wc, r, h);
}

// ------------------------------------------------------------------------------------------------
// LARGE-TILE LDS-DMA kernels. With 128x128 tiles every output element pulls (128+128)*K*2/16384 bytes through the
// CU's vector-memory path; at K = 768 that path (L2 -> LDS, ~45-70 GB/s per CU sustained), not the MFMA, sets the
// tile time. 256x128 / 128x256 tiles move 0.75x and 256x256 tiles 0.5x the bytes per output. 8 waves as WM x WN, a
// wave owns (32*MI) x (32*NJ) outputs = MI*NJ accumulators; LDS stage = (TBM+TBN) rows of 128 B, NST stages
// (256x256: 2 x 64 KiB; 256x128: 3 x 48 KiB), one raw s_barrier per k-tile behind a counted vmcnt. One block per CU.
template <typename T> struct MmaG;
template <> struct MmaG<__bf16> {
  static constexpr int KSTEPS = 4;
  template <int MI, int NJ>
  __device__ static __forceinline__ void step(const char* As, const char* Bs, int kk, int rowA0, int rowB0, int r, int h,
                                              f32x16 (&acc)[MI][NJ]) {
    const int chunk = kk * 2 + h;
    bf16x8 a[MI], b[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = *(const bf16x8*)(Bs + lds_off(rowB0 + j * 32 + r, chunk));
#pragma unroll
    for (int i = 0; i < MI; ++i) a[i] = *(const bf16x8*)(As + lds_off(rowA0 + i * 32 + r, chunk));
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};
template <> struct MmaG<float> {
  static constexpr int KSTEPS = 16;
  template <int MI, int NJ>
  __device__ static __forceinline__ void step(const char* As, const char* Bs, int kk, int rowA0, int rowB0, int r, int h,
                                              f32x16 (&acc)[MI][NJ]) {
    const int k = kk * 2 + h;
    const int chunk = k >> 2, within = (k & 3) * 4;
    float a[MI], b[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = *(const float*)(Bs + lds_off(rowB0 + j * 32 + r, chunk) + within);
#pragma unroll
    for (int i = 0; i < MI; ++i) a[i] = *(const float*)(As + lds_off(rowA0 + i * 32 + r, chunk) + within);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};

template <typename T, int NST, int WM, int WN, int MI, int NJ>
__global__ __launch_bounds__(WM * WN * 64, (NST * (WM * MI + WN * NJ) * 32 * ROWB <= 80 * 1024) ? (WM * WN / 2) : 1) void gemm_nt_big_kernel(GemmP p) {
  constexpr int ES = sizeof(T), NW = WM * WN, NTH = NW * 64;
  constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32, WROWS = MI * 32, WCOLS = NJ * 32;
  constexpr int STAGE = (TBM + TBN) * ROWB;
  constexpr int IA = TBM / 8 / NW, IB = TBN / 8 / NW;        // LDS-DMA instructions per wave per stage
  static_assert(TBM % (8 * NW) == 0 && TBN % (8 * NW) == 0, "tile rows must split evenly over the waves");
  static_assert(64 * TBN * 4 <= NST * STAGE, "epilogue scratch");
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  select_problem(p, wgid);
  int m0, n0;
  tile_origin(p, wgid, TBM, TBN, m0, n0);
  constexpr int BK = ROWB / ES;
  const int nk = p.K / BK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN, r = lane & 31, h = lane >> 5;

  const char* ga[IA];
  const char* gb[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    ga[i] = p.A + ((long)min(m0 + row, p.M - 1) * p.lda) * ES + ((lane & 7) ^ ((row >> 1) & 7)) * 16;
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    gb[i] = p.B + ((long)min(n0 + row, p.N - 1) * p.ldb) * ES + ((lane & 7) ^ ((row >> 1) & 7)) * 16;
  }
  using gptr = const __attribute__((address_space(1))) void*;
  using lptr = __attribute__((address_space(3))) void*;
  auto issue = [&](int t, int stage) {
    char* sa = dsmem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < IA; ++i) __builtin_amdgcn_global_load_lds((gptr)(ga[i] + (long)t * ROWB), (lptr)(sa + i * NW * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < IB; ++i)
      __builtin_amdgcn_global_load_lds((gptr)(gb[i] + (long)t * ROWB), (lptr)(sa + TBM * ROWB + i * NW * 1024), 16, 0, 0);
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

  issue(0, 0);
  if (NST == 3 && nk > 1) issue(1, 1);
  int stage = 0;
  for (int t = 0; t < nk; ++t) {
    if (NST == 3 && t + 1 < nk) {
      if constexpr (IA + IB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (IA + IB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (IA + IB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                          // tile t landed everywhere; stage (t-1)%NST is free
    if (t + NST - 1 < nk) issue(t + NST - 1, stage == 0 ? NST - 1 : stage - 1);
    const char* As = dsmem + stage * STAGE;
#pragma unroll
    for (int kk = 0; kk < MmaG<T>::KSTEPS; ++kk)
      MmaG<T>::template step<MI, NJ>(As, As + TBM * ROWB, kk, wr * WROWS, wc * WCOLS, r, h, acc);
    stage = stage == NST - 1 ? 0 : stage + 1;
  }
  __syncthreads();

  // ---- epilogue: 64-row slabs of the tile go through LDS as [64][TBN] float32, then row-wise 4-column vectors ----
  float* const e = (float*)dsmem;
  const unsigned dseed = p.drop_thr ? eff_seed(p.drop_seed, p.seed_base) : 0u;
  const int gcol = n0 + (tid % (TBN / 4)) * 4;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && gcol < p.N) {
    if (p.vec_ok) bv = *(const f32x4*)(p.bias + gcol);
    else
      for (int u = 0; u < 4; ++u) bv[u] = (gcol + u < p.N) ? p.bias[gcol + u] : 0.f;
  }
#pragma unroll
  for (int c = 0; c < TBM / 64; ++c) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int blk = wr * MI + i;                          // 32-row block of the tile held in acc[i][*]
      if ((blk >> 1) == c) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int x = 0; x < 16; ++x)
            e[((blk & 1) * 32 + (x & 3) + 8 * (x >> 2) + 4 * h) * TBN + wc * WCOLS + j * 32 + r] = acc[i][j][x];
      }
    }
    lds_barrier();
    epilogue_slab<T, TBN, NTH, 2>(p, e, m0 + c * 64, n0, tid, bv, dseed);
    lds_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient without transposes:  C[N,K] += A[M,N]^T * B[M,K]   (bf16 in, float32 atomics out)
//   A = dY (rows = tokens, N contiguous), B = X (rows = tokens, K contiguous); the reduction runs over ROWS,
//   so both MFMA operands are needed "k-major". The tiles are staged exactly as they lie in memory
//   ([64 rows][128 cols] = 256-byte rows, 16-byte chunks XOR-swizzled) and the fragments are fetched with the
//   gfx950 transposing LDS read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, column-major to the
//   lanes): two reads give a lane the 8 consecutive reduction indices of its output row/column.
//   Swizzle = guide image (b): chunk ^ (((row&3)<<2) | ((row>>2)&3)) -> conflict-free for the 32x32x16 tr reads.
//   Optionally the k-tile-0 blocks also reduce the columns of A (bias gradient), so no separate colsum pass.
constexpr int TN_MAXSEG = 16;
struct TnP {
  // up to 16 row segments (A_s [M_s,N], B_s [M_s,K]) reduced into the same C: the deferred weight gradient of one
  // parameter over all steps of an episode is ONE launch with a long reduction instead of one short launch per step
  const __bf16* A[TN_MAXSEG];
  const __bf16* B[TN_MAXSEG];
  int segM[TN_MAXSEG];
  int mt_start[TN_MAXSEG + 1];    // prefix sums of 64-row tiles
  int nseg;
  long lda, ldb;
  float* C; long ldc;
  int N, K;
  int mt_per_split;
  float* colsum;
  // "partials" mode (part != null): split z writes its share of C with plain stores to part + z * part_stride (dense [N][K]) and its
  // column sums to colsum + z * N; vlni_reduce_parts adds them into the gradient afterwards. Float atomics run at ~1.3 TB/s on this
  // chip against ~5 TB/s for stores, and were 30-50 % of a weight-gradient launch (tools/tn_probe.py).
  float* part; long part_stride;
};
__device__ __forceinline__ void tn_out(const TnP& p, int row, int col, float v) {
  if (p.part) p.part[(long)blockIdx.z * p.part_stride + (long)row * p.K + col] = v;
  else atomicAdd(&p.C[(long)row * p.ldc + col], v);
}
__device__ __forceinline__ void tn_cs(const TnP& p, int row, float v) {
  if (p.part) p.colsum[(long)blockIdx.z * p.N + row] = v;
  else atomicAdd(p.colsum + row, v);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tr_off(int row, int ch) { return row * 256 + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
// LDS-DMA issued as inline asm in the kernels that read their tiles with the transposing LDS read: hipcc's waitcnt pass treats the
// ds_read_tr BUILTIN as a read that may alias an outstanding global_load_lds BUILTIN and puts s_waitcnt vmcnt(0) in front of the
// first one - which serialises every stage's global -> LDS copy with the MFMAs it was meant to overlap (measured on the 256 x 256
// weight-gradient tile: 4.3 k cycles per 64-row step instead of 3.3 k). As asm the copy is invisible to that pass (its completion is
// awaited by the explicit s_waitcnt vmcnt(N) + s_barrier these kernels carry anyway) while the reads stay builtins, so the compiler
// still counts lgkmcnt for them. lds_addr must be wave-uniform: lane l's 16 bytes land at lds_addr + 16 * l. M0 is a reserved register
// (hipcc rejects it on the clobber list); nothing else in these kernels uses it.
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_u32(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row_lo, int ch, int half8) {
  using lds_ptr = __attribute__((address_space(3))) s16x4*;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + tr_off(row_lo, ch) + half8));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + tr_off(row_lo + 4, ch) + half8));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(NT) void gemm_tn_bf16_kernel(TnP p) {
  constexpr int BR = 64;                                   // reduction rows per stage
  __shared__ __attribute__((aligned(16))) char smem[2 * BR * 256];
  char* const As = smem;
  char* const Bs = smem + BR * 256;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, rr = nwg & 7;
  const int wgid = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int ntk = (p.K + BN - 1) / BN;
  const int n0 = (wgid / ntk) * BM, k0 = (wgid % ntk) * BN;
  const int nmt = p.mt_start[p.nseg];
  const int mt0 = blockIdx.z * p.mt_per_split, mt1 = min(nmt, mt0 + p.mt_per_split);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int ch = tid & 15, lr = tid >> 4;                  // staging: 16 chunks per 256-byte row, 16 rows per pass
  const bool a_ok = (n0 + ch * 8) < p.N, b_ok = (k0 + ch * 8) < p.K;   // N, K multiples of 8
  uint4 sa[4], sb[4];
  int seg = 0;
  auto gload = [&](int mt) {
    while (mt >= p.mt_start[seg + 1]) ++seg;                 // tiles are visited in increasing order
    const __bf16* ga = p.A[seg] + n0 + ch * 8;
    const __bf16* gb = p.B[seg] + k0 + ch * 8;
    const int segM = p.segM[seg], r0 = (mt - p.mt_start[seg]) * BR;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + lr + 16 * i;
      const bool ok = row < segM;
      sa[i] = (ok && a_ok) ? *(const uint4*)(ga + (long)row * p.lda) : make_uint4(0, 0, 0, 0);
      sb[i] = (ok && b_ok) ? *(const uint4*)(gb + (long)row * p.ldb) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(As + tr_off(lr + 16 * i, ch)) = sa[i];
      *(uint4*)(Bs + tr_off(lr + 16 * i, ch)) = sb[i];
    }
  };
  // fragment addressing: 16-lane group g = lane>>4 -> (h = g>>1, column block cb = g&1); lane 4q+p of the group
  // addresses row q, columns 4p..4p+3 of the 4x16 block
  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1;
  const int frow = 8 * h + qq;                             // + 16*kk (+4 for the second read)
  const int half8 = 8 * (pp & 1);
  const int cha = (wr * 64 + 16 * cb) / 8 + (pp >> 1);     // + 4*i  (32 columns = 4 chunks)
  const int chb = (wc * 64 + 16 * cb) / 8 + (pp >> 1);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;
  // bias gradient (column sums of A) for the k-tile-0 blocks: accumulated from the staging REGISTERS (the 8 columns of
  // this thread's chunk over its 4 rows per stage), so the MFMA loop sees no extra LDS traffic
  const bool do_cs = p.colsum != nullptr && k0 == 0;
  float csr[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csr[e] = 0.f;

  if (mt0 < mt1) {
    gload(mt0);
    for (int mt = mt0; mt < mt1; ++mt) {
      lstore();
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8 v = __builtin_bit_cast(bf16x8, sa[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) csr[e] += (float)v[e];
        }
      }
      __syncthreads();
      if (mt + 1 < mt1) gload(mt + 1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = tr_frag(As, 16 * kk + frow, cha + 4 * i, half8);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = tr_frag(Bs, 16 * kk + frow, chb + 4 * j, half8);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = k0 + wc * 64 + j * 32 + r;
    if (col >= p.K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int row = n0 + wr * 64 + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
        if (row < p.N) tn_out(p, row, col, acc[i][j][x]);
      }
  }
  if (do_cs) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {                        // lanes l, l+16, l+32, l+48 hold the same chunk: fold them first
      csr[e] += __shfl_xor(csr[e], 16, 64);
      csr[e] += __shfl_xor(csr[e], 32, 64);
    }
    if (lane < 16 && a_ok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(p.colsum + n0 + ch * 8 + e, csr[e]);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// The same weight-gradient contraction with the LDS-DMA pipeline of gemm_nt_glds_kernel: row tiles go global -> LDS with
// global_load_lds_dwordx4 (1 KiB = 4 rows of 256 B per wave-instruction, the image-(b) swizzle applied to the per-lane SOURCE
// chunk), NST stages, NW = 4 or 8 waves. Rows past a segment's end and columns past N / K read a 256-byte page of zeros
// (LDS-DMA cannot zero-fill). The bias gradient rides the matrix pipe: colsum(A) = A^T * 1, one extra MFMA per k-step with an
// all-ones B fragment in the k-tile-0 blocks (no VALU / LDS work in the loop).
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

template <int NST, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_tn_glds_kernel(TnP p) {
  constexpr int BR = 64, STAGE = 2 * BR * 256, WC = NW / 2, NJ = NW == 4 ? 2 : 1, WCOLS = 32 * NJ, IPW = 16 / NW;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, rr = nwg & 7;
  const int wgid = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int ntk = (p.K + BN - 1) / BN;
  const int n0 = (wgid / ntk) * BM, k0 = (wgid % ntk) * BN;
  const int nmt = p.mt_start[p.nseg];
  const int mt0 = blockIdx.z * p.mt_per_split;
  const int nk = min(nmt, mt0 + p.mt_per_split) - mt0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC, r = lane & 31, h = lane >> 5;

  // per-lane source geometry of this wave's IPW LDS-DMA instructions per operand
  int srow[IPW], acol[IPW], bcol[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int row = (i * NW + wave) * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));     // logical chunk this LDS slot must hold
    srow[i] = row;
    acol[i] = (n0 + c * 8 < p.N) ? n0 + c * 8 : -1;
    bcol[i] = (k0 + c * 8 < p.K) ? k0 + c * 8 : -1;
  }
  int seg = 0;
  const char* zp = (const char*)g_zero_page;            // pinned in scalar registers: otherwise re-loaded (GOT) inside the loop
  asm volatile("" : "+s"(zp));
  const unsigned lds0 = lds_u32(dsmem);
  auto issue = [&](int t, int stage) {
    const int mt = mt0 + t;
    while (mt >= p.mt_start[seg + 1]) ++seg;                 // tiles are issued in increasing order
    const int segM = p.segM[seg], r0 = (mt - p.mt_start[seg]) * BR;
    const __bf16* A = p.A[seg];
    const __bf16* B = p.B[seg];
    const unsigned sa = lds0 + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int row = r0 + srow[i];
      const bool rok = row < segM;
      const void* pa = (rok && acol[i] >= 0) ? (const void*)(A + (long)row * p.lda + acol[i]) : (const void*)zp;
      const void* pb = (rok && bcol[i] >= 0) ? (const void*)(B + (long)row * p.ldb + bcol[i]) : (const void*)zp;
      lds_dma16(pa, sa + i * NW * 1024);
      lds_dma16(pb, sa + BR * 256 + i * NW * 1024);
    }
  };
  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1;
  const int frow = 8 * h + qq, half8 = 8 * (pp & 1);
  const int cha = (wr * 64 + 16 * cb) / 8 + (pp >> 1);
  const int chb = (wc * WCOLS + 16 * cb) / 8 + (pp >> 1);

  f32x16 acc[2][NJ], acs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int x = 0; x < 16; ++x) acs[i][x] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;
  }
  const bool do_cs = p.colsum != nullptr && k0 == 0 && wc == 0;      // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  auto frags = [&](const char* As, const char* Bs, int kk, bf16x8 (&a)[2], bf16x8 (&b)[NJ]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = tr_frag(As, 16 * kk + frow, cha + 4 * i, half8);
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = tr_frag(Bs, 16 * kk + frow, chb + 4 * j, half8);
  };
  auto mma = [&](const bf16x8 (&a)[2], const bf16x8 (&b)[NJ]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], ones, acs[i], 0, 0, 0);
    }
  };

  if (nk > 0) {
    issue(0, 0);
    if (NST == 3 && nk > 1) issue(1, 1);
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
      if (NST == 3 && t + 1 < nk) {
        if (NW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (t + NST - 1 < nk) issue(t + NST - 1, stage == 0 ? NST - 1 : stage - 1);
      // k-step kk + 1's fragments are requested before k-step kk's MFMAs issue (two fragment sets)
      const char* As = dsmem + stage * STAGE;
      const char* Bs = As + BR * 256;
      bf16x8 a0[2], b0[NJ], a1[2], b1[NJ];
      frags(As, Bs, 0, a0, b0);
      frags(As, Bs, 1, a1, b1);
      mma(a0, b0);
      frags(As, Bs, 2, a0, b0);
      mma(a1, b1);
      frags(As, Bs, 3, a1, b1);
      mma(a0, b0);
      mma(a1, b1);
      stage = stage == NST - 1 ? 0 : stage + 1;
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = k0 + wc * WCOLS + j * 32 + r;
    if (col >= p.K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int row = n0 + wr * 64 + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
        if (row < p.N) tn_out(p, row, col, acc[i][j][x]);
      }
  }
  if (do_cs && r == 0) {                                   // column 0 of the ones-product holds the column sums of A
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int row = n0 + wr * 64 + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
        if (row < p.N) tn_cs(p, row, acs[i][x]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// "NN" form of the same contraction for bf16 dgrad:  C[M,N] = epi(A[M,K] * B[K,N])  with B = W exactly as the forward pass
// stores it ([out, in] row-major): no transposed weight copy has to be rebuilt after every optimizer step. A is staged and
// read like in gemm_nt_glds_kernel; the B tile is staged as it lies in memory ([64 k-rows][128 columns], 256-byte rows,
// tr_off swizzle on the per-lane SOURCE chunk) and its fragments come from the transposing LDS read (ds_read_b64_tr_b16),
// like the X operand of the weight-gradient kernel. Same stage size (32 KiB), same instruction counts, same epilogue.
template <int NST, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_nn_glds_kernel(GemmP p) {
  using T = __bf16;
  constexpr int ES = 2, BK = 64;
  constexpr int WC = NW / 2, NJ = NW == 4 ? 2 : 1, WCOLS = 32 * NJ, IPW = 16 / NW;
  constexpr int STAGE = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  select_problem(p, wgid);
  int m0, n0;
  tile_origin(p, wgid, BM, BN, m0, n0);
  const int nk = p.K / BK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC, r = lane & 31, h = lane >> 5;

  const char* ga[IPW];
  const char* gb[IPW];
  bool bok[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    ga[i] = p.A + ((long)min(m0 + row, p.M - 1) * p.lda) * ES + ((lane & 7) ^ ((row >> 1) & 7)) * 16;
    const int krow = (i * NW + wave) * 4 + (lane >> 4);                    // 4 k-rows of 256 B per LDS-DMA instruction
    const int c = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));   // logical 16-B chunk this LDS slot must hold
    bok[i] = n0 + c * 8 < p.N;                                             // N is a multiple of 8: a chunk is all-in or all-out
    gb[i] = p.B + ((long)krow * p.ldb + n0 + c * 8) * ES;
  }
  const char* zp = (const char*)g_zero_page;            // pinned in scalar registers: otherwise re-loaded (GOT) inside the loop
  asm volatile("" : "+s"(zp));
  const unsigned lds0 = lds_u32(dsmem);
  auto issue = [&](int t, int stage) {
    const unsigned sa = lds0 + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      lds_dma16(ga[i] + (long)t * ROWB, sa + i * NW * 1024);
      const void* pb = bok[i] ? (const void*)(gb[i] + (long)t * BK * p.ldb * ES) : (const void*)zp;
      lds_dma16(pb, sa + BM * ROWB + i * NW * 1024);
    }
  };
  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1;
  const int frow = 8 * h + qq, half8 = 8 * (pp & 1);
  const int chb = (wc * WCOLS + 16 * cb) / 8 + (pp >> 1);

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

  if (nk > 0) {
    issue(0, 0);
    if (NST == 3 && nk > 1) issue(1, 1);
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
      if (NST == 3 && t + 1 < nk) {
        if (NW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (t + NST - 1 < nk) issue(t + NST - 1, stage == 0 ? NST - 1 : stage - 1);
      const char* As = dsmem + stage * STAGE;
      const char* Bs = As + BM * ROWB;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 a[2], b[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8*)(As + lds_off(wr * 64 + i * 32 + r, kk * 2 + h));
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = tr_frag(Bs, 16 * kk + frow, chb + 4 * j, half8);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      stage = stage == NST - 1 ? 0 : stage + 1;
    }
  }
  __syncthreads();
  gemm_epilogue<T, NW>(p, dsmem, acc, m0, n0, tid, wr, wc, r, h);
}

// 256 x 256 output tile of the same weight-gradient contraction: 8 waves as 2 (N) x 4 (K), a wave owns 128 x 64 outputs
// (8 accumulators), half the L2->LDS bytes and half the LDS-read bytes per MFMA of the 128 x 128 tile. A stage holds four
// [64 rows][128 cols] sub-tiles (A left/right, B left/right) in the layout of the kernel above; 2 stages = 128 KiB.
// The LDS-DMA is issued as asm (lds_dma16 above) so the next stage's copy stays in flight under the MFMAs; k-step kk + 1's
// twelve fragment reads are requested before k-step kk's eight MFMAs. The bias gradient (column sums of A) costs ONE extra
// MFMA per k-step and wave: the four waves that share an A half each take one of its four 32-row blocks.
template <int NST>
__global__ __launch_bounds__(512) void gemm_tn_big_kernel(TnP p) {
  constexpr int BR = 64, SUB = BR * 256, STAGE = 4 * SUB, NW = 8;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, rr = nwg & 7;
  const int wgid = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int ntk = (p.K + 255) / 256;
  const int n0 = (wgid / ntk) * 256, k0 = (wgid % ntk) * 256;
  const int nmt = p.mt_start[p.nseg];
  const int mt0 = blockIdx.z * p.mt_per_split;
  const int nk = min(nmt, mt0 + p.mt_per_split) - mt0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, r = lane & 31, h = lane >> 5;

  // a sub-tile is 16 LDS-DMA instructions (4 rows x 256 B each); 8 waves x 2 instructions per sub-tile.
  // per-lane element offsets inside a 64-row tile of A / B (-1: column past N / K -> the zero page)
  int srow[2];
  long aoff[2][2], boff[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (i * NW + wave) * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    srow[i] = row;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      aoff[u][i] = (n0 + u * 128 + c * 8 < p.N) ? (long)row * p.lda + n0 + u * 128 + c * 8 : -1;
      boff[u][i] = (k0 + u * 128 + c * 8 < p.K) ? (long)row * p.ldb + k0 + u * 128 + c * 8 : -1;
    }
  }
  // the segment a tile lies in changes a handful of times per block: its descriptors stay in scalar registers and are re-read
  // from the kernel arguments only on a change (reading them every step put ~1 k cycles of scalar-load latency between the
  // barrier and the first MFMA)
  int seg = 0;
  while (mt0 >= p.mt_start[seg + 1] && seg + 1 < p.nseg) ++seg;
  const __bf16* curA = p.A[seg];
  const __bf16* curB = p.B[seg];
  int curM = p.segM[seg], seg_lo = p.mt_start[seg], seg_hi = p.mt_start[seg + 1];
  const char* zp = (const char*)g_zero_page;            // pinned in scalar registers: otherwise re-loaded (GOT) inside the loop
  asm volatile("" : "+s"(zp));
  const unsigned lds0 = lds_u32(dsmem);
  auto issue = [&](int t, int stage) {
    const int mt = mt0 + t;
    if (mt >= seg_hi) {                                      // tiles are issued in increasing order
      do { ++seg; } while (mt >= p.mt_start[seg + 1]);
      curA = p.A[seg]; curB = p.B[seg]; curM = p.segM[seg]; seg_lo = p.mt_start[seg]; seg_hi = p.mt_start[seg + 1];
    }
    const int r0 = (mt - seg_lo) * BR;
    const __bf16* At = curA + (long)r0 * p.lda;
    const __bf16* Bt = curB + (long)r0 * p.ldb;
    const unsigned sa = lds0 + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool rok = r0 + srow[i] < curM;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const void* pa = (rok && aoff[u][i] >= 0) ? (const void*)(At + aoff[u][i]) : (const void*)zp;
        const void* pb = (rok && boff[u][i] >= 0) ? (const void*)(Bt + boff[u][i]) : (const void*)zp;
        lds_dma16(pa, sa + u * SUB + i * NW * 1024);
        lds_dma16(pb, sa + (2 + u) * SUB + i * NW * 1024);
      }
    }
  };
  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1;
  const int frow = 8 * h + qq, half8 = 8 * (pp & 1);
  const int cha = (16 * cb) / 8 + (pp >> 1);                         // + 4 * i: 32-column block i of A sub-tile wr
  const int chb = ((wc & 1) * 64 + 16 * cb) / 8 + (pp >> 1);         // + 4 * j inside B sub-tile wc >> 1
  f32x16 acc[4][2], acs;
#pragma unroll
  for (int x = 0; x < 16; ++x) acs[x] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;
  const bool do_cs = p.colsum != nullptr && k0 == 0;                 // block-uniform; wave (wr, wc) sums A block wc of half wr
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

  // fragments of k-step kk of the stage at `st`: A blocks 0..3 of sub-tile wr, B blocks 0..1 of sub-tile 2 + (wc >> 1)
  auto frags = [&](const char* st, int kk, bf16x8 (&a)[4], bf16x8 (&b)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = tr_frag(st + (2 + (wc >> 1)) * SUB, 16 * kk + frow, chb + 4 * j, half8);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = tr_frag(st + wr * SUB, 16 * kk + frow, cha + 4 * i, half8);
  };
  auto mma = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (do_cs) {
      const bf16x8 mine = wc == 0 ? a[0] : wc == 1 ? a[1] : wc == 2 ? a[2] : a[3];
      acs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mine, ones, acs, 0, 0, 0);
    }
  };

  if (nk > 0) {
    issue(0, 0);
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 1 < nk) issue(t + 1, stage ^ 1);
      const char* st = dsmem + stage * STAGE;
      bf16x8 a0[4], b0[2], a1[4], b1[2];
      frags(st, 0, a0, b0);
      frags(st, 1, a1, b1);
      mma(a0, b0);
      frags(st, 2, a0, b0);
      mma(a1, b1);
      frags(st, 3, a1, b1);
      mma(a0, b0);
      mma(a1, b1);
      stage ^= 1;
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = k0 + wc * 64 + j * 32 + r;
    if (col >= p.K) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int row = n0 + wr * 128 + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
        if (row < p.N) tn_out(p, row, col, acc[i][j][x]);
      }
  }
  if (do_cs && r == 0) {                                   // column 0 of the ones-product holds the column sums of A
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int row = n0 + wr * 128 + wc * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
      if (row < p.N) tn_cs(p, row, acs[x]);
    }
  }
}

// The same 256 x 256 tile as a 4-slot ring of 32-row half-steps (4 x 32 KiB): a half-step's rows are requested three barriers
// before they are read (two half-steps = 64 KiB in flight under the MFMAs of a third), and the barrier at the top of half-step s
// certifies half-step s + 1, so the first fragments of s + 1 are already being read from LDS while the last MFMAs of s issue -
// no LDS latency and no global latency is exposed after a barrier. 64-row reduction tiles of the host (mt_start, mt_per_split)
// are walked as two half-steps each. (Tried and dropped: the same ring with four waves of 128 x 128 - a third less LDS-read
// traffic, but one wave per SIMD hides nothing: 383 vs 251 us on the 16.5 k-row x 3072 x 768 probe at split 2.)
__global__ __launch_bounds__(512) void gemm_tn_ring_kernel(TnP p) {
  constexpr int BR = 32, SUB = BR * 256, STAGE = 4 * SUB, NSLOT = 4;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, rr = nwg & 7;
  const int wgid = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (bid >> 3);
  const int ntk = (p.K + 255) / 256;
  const int n0 = (wgid / ntk) * 256, k0 = (wgid % ntk) * 256;
  const int nmt = p.mt_start[p.nseg];
  const int mt0 = blockIdx.z * p.mt_per_split;
  const int S = 2 * (min(nmt, mt0 + p.mt_per_split) - mt0);          // half-steps of this block

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, r = lane & 31, h = lane >> 5;

  // a 32-row sub-tile is 8 LDS-DMA instructions (4 rows x 256 B each): one per wave and sub-tile.
  // Per lane: four running source pointers (A left/right, B left/right) that advance by 32 rows per half-step; a lane whose 16-byte
  // chunk lies past N / K points at the zero page with stride 0. Only a segment change or a segment's ragged last rows recompute
  // anything (address arithmetic every half-step cost ~480 cycles of the ~1 k-cycle MFMA budget).
  const int srow = wave * 4 + (lane >> 4);
  const int c16 = (lane & 15) ^ (((srow & 3) << 2) | ((srow >> 2) & 3));
  bool cok[4];
  long coff[4];                                               // element offset of the lane's chunk inside a 32-row slab
  unsigned step[4];                                           // bytes per half-step (0 for zero-page lanes)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    cok[u] = n0 + u * 128 + c16 * 8 < p.N;
    cok[2 + u] = k0 + u * 128 + c16 * 8 < p.K;
    coff[u] = (long)srow * p.lda + n0 + u * 128 + c16 * 8;
    coff[2 + u] = (long)srow * p.ldb + k0 + u * 128 + c16 * 8;
    step[u] = cok[u] ? (unsigned)(BR * p.lda * 2) : 0u;
    step[2 + u] = cok[2 + u] ? (unsigned)(BR * p.ldb * 2) : 0u;
  }
  int seg = 0;
  while (mt0 >= p.mt_start[seg + 1] && seg + 1 < p.nseg) ++seg;
  int curM = p.segM[seg], seg_lo = p.mt_start[seg], seg_hi = p.mt_start[seg + 1];
  const char* zp = (const char*)g_zero_page;
  asm volatile("" : "+s"(zp));
  const unsigned lds0 = lds_u32(dsmem);
  const char* ptr[4];
  auto point_at = [&](int r0) {                               // pointers of the half-step that starts at row r0 of segment `seg`
    const __bf16* A = p.A[seg] + (long)r0 * p.lda;
    const __bf16* B = p.B[seg] + (long)r0 * p.ldb;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ptr[u] = cok[u] ? (const char*)(A + coff[u]) : zp;
      ptr[2 + u] = cok[2 + u] ? (const char*)(B + coff[2 + u]) : zp;
    }
  };
  point_at((mt0 - seg_lo) * 64);
  auto issue = [&](int q) {                                   // half-step q -> slot q % 4; half-steps are issued in increasing order
    const int mt = mt0 + (q >> 1);
    if (mt >= seg_hi) {
      do { ++seg; } while (mt >= p.mt_start[seg + 1]);
      curM = p.segM[seg]; seg_lo = p.mt_start[seg]; seg_hi = p.mt_start[seg + 1];
      point_at(0);
    }
    const int r0 = (mt - seg_lo) * 64 + (q & 1) * BR;
    const unsigned sa = lds0 + (q & (NSLOT - 1)) * STAGE + wave * 1024;
    if (r0 + BR <= curM) {
#pragma unroll
      for (int u = 0; u < 4; ++u) lds_dma16(ptr[u], sa + u * SUB);
    } else {                                                  // ragged end of a segment: rows past it read zeros
      const bool rok = r0 + srow < curM;
#pragma unroll
      for (int u = 0; u < 4; ++u) lds_dma16(rok ? ptr[u] : zp, sa + u * SUB);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) ptr[u] += step[u];
  };
  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1;
  const int frow = 8 * h + qq, half8 = 8 * (pp & 1);
  const int cha = (16 * cb) / 8 + (pp >> 1);
  const int chb = ((wc & 1) * 64 + 16 * cb) / 8 + (pp >> 1);
  f32x16 acc[4][2], acs;
#pragma unroll
  for (int x = 0; x < 16; ++x) acs[x] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;
  const bool do_cs = p.colsum != nullptr && k0 == 0;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  // fragments of k-step kk of the stage at `st`: A blocks 0..3 of sub-tile wr, B blocks 0..1 of sub-tile 2 + (wc >> 1)
  auto frags = [&](const char* st, int kk, bf16x8 (&a)[4], bf16x8 (&b)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = tr_frag(st + (2 + (wc >> 1)) * SUB, 16 * kk + frow, chb + 4 * j, half8);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = tr_frag(st + wr * SUB, 16 * kk + frow, cha + 4 * i, half8);
  };
  // one half-step = two groups of [8 (+1) MFMAs of one fragment set | the 12 reads (24 ds_read_tr) of the other set], interleaved
  // instead of 24 reads up front (right after a barrier both waves of a SIMD are in the same phase, so reads-first leaves the matrix
  // pipe idle). Same probe, split 2: no hint 247 us, sched_group_barrier (1 MFMA, 3 reads) x 8: 249 us, iglp_opt(0): 233 us;
  // iglp_opt(1) sends hipcc out of memory on this file. On the 2-stage kernels above the hint changes nothing.
  auto half = [&](auto CS, const char* rd, int rd_kk, bf16x8 (&an)[4], bf16x8 (&bn)[2], const bf16x8 (&a)[4], const bf16x8 (&b)[2]) {
    frags(rd, rd_kk, an, bn);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if constexpr (decltype(CS)::value) {
      const bf16x8 mine = wc == 0 ? a[0] : wc == 1 ? a[1] : wc == 2 ? a[2] : a[3];
      acs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mine, ones, acs, 0, 0, 0);
    }
    __builtin_amdgcn_iglp_opt(0);                             // the compiler's MFMA / DS-read interleave for small GEMM loops
  };
  auto run = [&](auto CS) {
    bf16x8 a0[4], b0[2], a1[4], b1[2];
    frags(dsmem, 0, a0, b0);
    for (int s = 0; s < S; ++s) {
      // own share of half-step s + 1 landed (s + 2 may still be in flight), then everybody's; every wave is also done reading s - 1
      if (s + 2 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 3 < S) issue(s + 3);
      half(CS, dsmem + (s & (NSLOT - 1)) * STAGE, 1, a1, b1, a0, b0);
      // after the last half-step this reads a slot nobody needs (unused values) - keeps the loop body free of branches
      half(CS, dsmem + ((s + 1) & (NSLOT - 1)) * STAGE, 0, a0, b0, a1, b1);
    }
  };

  if (S > 0) {
    issue(0);
    if (S > 1) issue(1);
    if (S > 2) issue(2);
    // half-step 0 has to be there before its first fragments are read
    if (S > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (S > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (do_cs) run(std::true_type{});
    else run(std::false_type{});
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = k0 + wc * 64 + j * 32 + r;
    if (col >= p.K) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int row = n0 + wr * 128 + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
        if (row < p.N) tn_out(p, row, col, acc[i][j][x]);
      }
  }
  if (do_cs && r == 0) {
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int row = n0 + wr * 128 + wc * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
      if (row < p.N) tn_cs(p, row, acs[x]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT form of the 128 x 128 LDS-DMA kernel (variant 14, variant 22 = B as [K,N]): <= 512 blocks (two per CU, 8 waves each)
// walk the tile list; the k-tiles of consecutive tiles form ONE stream through the two LDS stages, so only a block's first tile has
// a prologue, and the epilogue never touches LDS: the MFMA operands are swapped (D^T = B A^T), which leaves a lane with one output
// ROW and 8 x 4 consecutive COLUMNS in its accumulators; four v_permlane32_swap per 16-column group turn that into 8 consecutive
// columns per lane, so bias / GELU / GELU' / dropout / residual run in registers with 16-byte global accesses while the LDS-DMA of
// the next tile's first k-tile is already in flight (issued before the last k-tile's MFMAs). In the per-launch kernels above
// prologue + epilogue are ~35 % of a 128 x 128 x 768 tile (2.8 k + 7.6 k of 29.5 k cycles) and fully exposed.
// Wave grid 4 (M) x 2 (N), a wave owns 32 rows x 64 columns = 2 accumulators: the four stores of a wave cover whole 128-byte lines.
// Same k order and the same products as every other variant -> bit-identical results.
struct PkSide { const char* A; long lda; const char* B; long ldb; int M, m0, n0; };

// Diagnostic build only (HACK & 4, VLNI_PK_HACK=4): per block, wave 0's cycle sums of the phases of its k-steps; read back with
// vlni_debug_pk_stamps. [0] waits for its LDS-DMA, [1] barrier, [2] LDS-DMA issue, [3] fragment reads + MFMA issue, [4] epilogue,
// [5] k-steps, [6] whole kernel, [7] first wait + barrier (cold start). Never executed by the product kernels.
__device__ unsigned long long g_pk_stamps[768 * 8];
__device__ __forceinline__ unsigned long long pk_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

__device__ __forceinline__ void pk_locate(const GemmP& p, int tile, PkSide& s) {
  int w = tile;
  if (w >= p.tiles0) { w -= p.tiles0; s.A = p.A1; s.lda = p.lda1; s.B = p.B1; s.ldb = p.ldb1; s.M = p.M1; }
  else { s.A = p.A; s.lda = p.lda; s.B = p.B; s.ldb = p.ldb; s.M = p.M; }
  const int ntn = (p.N + BN - 1) / BN;
  if (p.group_m <= 1) { s.m0 = (w / ntn) * BM; s.n0 = (w % ntn) * BN; return; }
  const int ntm = (s.M + BM - 1) / BM;
  const int per = p.group_m * ntn;
  const int g = w / per, idx = w - g * per;
  const int first = g * p.group_m, gsz = min(ntm - first, p.group_m);
  s.m0 = (first + idx % gsz) * BM;
  s.n0 = (idx / gsz) * BN;
}

// Epilogue of one 128 x 128 tile straight from the (transposed) accumulators of the persistent kernels: wave (wr, wc) of the 4 x 2 grid owns
// rows wr*32.. and columns wc*64.., lane = row, registers = columns; zeroes the accumulators for the next tile.
__device__ __forceinline__ void pk_epilogue(const GemmP& p, int tile, const PkSide& cur, f32x16 (&acc)[2], int wr, int wc, int r, int h,
                                            unsigned dseed0, unsigned dseed1) {
  using T = __bf16;
  const bool p1 = tile >= p.tiles0;
  const float* bias = p1 ? p.bias1 : p.bias;
  const T* res = (const T*)(p1 ? p.residual1 : p.residual); const long ldr = p1 ? p.ldr1 : p.ldr;
  T* pre = (T*)(p1 ? p.preact1 : p.preact); const long ldp = p1 ? p.ldp1 : p.ldp;
  const T* dsrc = (const T*)(p1 ? p.dact_src1 : p.dact_src); const long ldd = p1 ? p.ldd1 : p.ldd;
  T* C = (T*)(p1 ? p.C1 : p.C); const long ldc = p1 ? p.ldc1 : p.ldc;
  const unsigned dseed = p1 ? dseed1 : dseed0;
  const int m = cur.m0 + wr * 32 + r;
  const bool mok = m < cur.M;
  const int mc = min(m, cur.M - 1);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int col = cur.n0 + wc * 64 + j * 32 + 16 * q + 8 * h;          // this lane's 8 consecutive columns after the swaps
      float v[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // scalar copies + __float_as_uint: __builtin_bit_cast applied to an ext_vector ELEMENT makes clang (ROCm 7.2) pass the
        // same value for both operands of every swap (one v_permlane32_swap for all eight) - checked in the .ll / .s
        const float lo = acc[j][8 * q + u], hi = acc[j][8 * q + 4 + u];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
        v[u] = __uint_as_float(sw[0]);
        v[4 + u] = __uint_as_float(sw[1]);
      }
      if (col < p.N) {                                                     // N % 8 == 0: a lane's 8 columns are all-in or all-out
        if (bias) {
          const f32x4 b0 = *(const f32x4*)(bias + col), b1 = *(const f32x4*)(bias + col + 4);
#pragma unroll
          for (int u = 0; u < 4; ++u) { v[u] = v[u] * p.alpha + b0[u]; v[4 + u] = v[4 + u] * p.alpha + b1[u]; }
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] *= p.alpha;
        }
        if (pre && mok) {
          bf16x8 t;
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = (T)v[u];
          *(bf16x8*)(pre + (long)m * ldp + col) = t;
        }
        if (p.dact) {
          const bf16x8 z = *(const bf16x8*)(dsrc + (long)mc * ldd + col);
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] *= (p.dact == 1) ? gelu_grad_t<T>((float)z[u]) : ((float)z[u] > 0.f ? 1.f : 0.f);
        }
        if (p.act == 1) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = gelu_t<T>(v[u]);
        } else if (p.act == 2) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = fmaxf(v[u], 0.f);
        }
        if (p.drop_thr) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] *= drop_scale((unsigned)m * (unsigned)p.N + col + u, dseed, p.drop_thr, p.drop_inv);
        }
        if (res) {
          const bf16x8 rs = *(const bf16x8*)(res + (long)mc * ldr + col);
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] += (float)rs[u];
        }
        if (mok) {
          bf16x8 t;
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = (T)v[u];
          *(bf16x8*)(C + (long)m * ldc + col) = t;
        }
      }
    }
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[j][x] = 0.f;
  }
}

// HACK != 0: timing-only builds that give WRONG results (tools/gemm_step_probe.py, VLNI_PK_HACK): 1 = the B operand's LDS-DMA is issued for a
// tile's first k-tile only (half the L2 -> LDS bytes), 2 = fragments are read from LDS for the first k16 step of a k-tile only (a quarter
// of the LDS reads), 3 = both. They price the vector-memory pipe and the LDS read bandwidth against the MFMA time of this loop.
template <bool NN, int HACK = 0>
__global__ __launch_bounds__(512, 4) void gemm_pk_kernel(GemmP p, int ntiles) {
  using T = __bf16;
  constexpr int BK = 64, NW = 8, STAGE = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  // this block's tiles: the XCD's contiguous chunk of the (grouped) tile list, walked with the XCD's block count as stride, so the
  // tiles an XCD works on at any time are neighbours that share operand panels in its L2
  const int bid = blockIdx.x, xcd = bid & 7, gx = gridDim.x >> 3;
  const int q8 = ntiles >> 3, rr = ntiles & 7;
  const int c0 = xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8;
  const int c1 = c0 + q8 + (xcd < rr ? 1 : 0);
  int tile = c0 + (bid >> 3);
  if (tile >= c1) return;
  const int nk = p.K / BK;

  // ---- issue side: per-lane source pointers of this wave's 2 + 2 LDS-DMA instructions per k-tile ----
  const char* zp = (const char*)g_zero_page;
  asm volatile("" : "+s"(zp));
  const unsigned lds0 = lds_u32(dsmem);
  const char* ga[2];
  const char* gb[2];
  unsigned bstep[2];                                       // bytes per k-tile of the B pointers (0 for zero-page lanes)
  auto point_at = [&](const PkSide& s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * NW + wave) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      ga[i] = s.A + ((long)min(s.m0 + row, s.M - 1) * s.lda) * 2 + c * 16;
      if constexpr (NN) {
        const int krow = (i * NW + wave) * 4 + (lane >> 4);                    // 4 k-rows of 256 B per LDS-DMA instruction
        const int cc = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
        const bool ok = s.n0 + cc * 8 < p.N;
        gb[i] = ok ? s.B + ((long)krow * s.ldb + s.n0 + cc * 8) * 2 : zp;
        bstep[i] = ok ? (unsigned)(BK * s.ldb * 2) : 0u;
      } else {
        gb[i] = s.B + ((long)min(s.n0 + row, p.N - 1) * s.ldb) * 2 + c * 16;
        bstep[i] = ROWB;
      }
    }
  };
  // one of the wave's four LDS-DMA instructions of a k-tile (piece 0 / 2: A rows, 1 / 3: B rows); pointers advance with the piece
  auto issue_piece = [&](int stage, int piece, bool with_b = true) {
    const unsigned sa = lds0 + stage * STAGE + wave * 1024;
    const int i = piece >> 1;
    if (piece & 1) {
      if (!(HACK & 1) || with_b) lds_dma16(gb[i], sa + BM * ROWB + i * NW * 1024);
      gb[i] += bstep[i];
    } else {
      lds_dma16(ga[i], sa + i * NW * 1024);
      ga[i] += ROWB;
    }
  };
  auto issue = [&](int stage, bool with_b = true) {
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) issue_piece(stage, pc, with_b);
  };
  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1;
  const int frow = 8 * h + qq, half8 = 8 * (pp & 1);
  const int chb = (wc * 64 + 16 * cb) / 8 + (pp >> 1);

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[j][x] = 0.f;
  const unsigned dseed0 = p.drop_thr ? eff_seed(p.drop_seed, p.seed_base) : 0u;
  const unsigned dseed1 = p.drop_thr ? eff_seed(p.drop_seed1, p.seed_base) : 0u;

  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t3 = 0, st_begin = 0;
  bool first_step = true;
  if constexpr (HACK & 4) st_begin = pk_stamp();
  PkSide cur, nxt;
  pk_locate(p, tile, cur);
  point_at(cur);
  issue(0);
  int stage = 0;
  while (true) {
    const int ntile = tile + gx;
    const bool more = ntile < c1;
    for (int kt = 0; kt < nk; ++kt) {
      unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      if constexpr (HACK & 4) s0 = pk_stamp();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of k-tile kt landed (and its last epilogue's stores)
      if constexpr (HACK & 4) s1 = pk_stamp();
      __builtin_amdgcn_s_barrier();                        // everybody's share landed; everybody finished reading the other stage
      if constexpr (HACK & 4) s2 = pk_stamp();
      // the next k-tile's LDS-DMA (this tile's, or the first of the block's next tile) goes out right behind the barrier. (Issuing it
      // piece by piece behind the MFMAs of the k16 steps below was slower: 47.6 vs 43.4 us on the N = 2304 dual launch of a step.)
      const bool in_tile = kt + 1 < nk;
      if (!in_tile && more) {
        pk_locate(p, ntile, nxt);
        point_at(nxt);
      }
      if (in_tile || more) issue(stage ^ 1, !in_tile);
      if constexpr (HACK & 4) {
        s3 = pk_stamp();
        st_acc[0] += s1 - s0; st_acc[1] += s2 - s1; st_acc[2] += s3 - s2; st_acc[5] += 1;
        if (first_step) { st_acc[7] = s2 - s0; first_step = false; }
        st_t3 = s3;
      }
      const char* As = dsmem + stage * STAGE;
      const char* Bs = As + BM * ROWB;
      bf16x8 a, b[2];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (!(HACK & 2) || kk == 0) {
          a = *(const bf16x8*)(As + lds_off(wr * 32 + r, kk * 2 + h));
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (NN) b[j] = tr_frag(Bs, 16 * kk + frow, chb + 4 * j, half8);
            else b[j] = *(const bf16x8*)(Bs + lds_off(wc * 64 + j * 32 + r, kk * 2 + h));
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a, acc[j], 0, 0, 0);   // D^T: lane = row, regs = columns
      }
      if constexpr (HACK & 4) st_acc[3] += pk_stamp() - st_t3;
      stage ^= 1;
    }
    unsigned long long e0 = 0;
    if constexpr (HACK & 4) e0 = pk_stamp();
    pk_epilogue(p, tile, cur, acc, wr, wc, r, h, dseed0, dseed1);
    if constexpr (HACK & 4) st_acc[4] += pk_stamp() - e0;
    if (!more) break;
    tile = ntile;
    cur = nxt;
  }
  if constexpr (HACK & 4) {
    st_acc[6] = pk_stamp() - st_begin;
    if (tid == 0 && blockIdx.x < 768)
      for (int i = 0; i < 8; ++i) g_pk_stamps[blockIdx.x * 8 + i] = st_acc[i];
  }
}

// ------------------------------------------------------------------------------------------------
// The persistent kernel with a DEEPER prefetch at the same two blocks per CU (variants 15 / 32, + 16 for [K,N] weights): the k-tile
// stream is cut into 32-deep half-steps that travel through a ring of NSLOT 16-KiB slots (4 slots = 64 KiB, 5 = 80 KiB per block), and a
// half-step is requested NSLOT - 1 barriers before it is read (counted vmcnt, never 0 in steady state). The 2-stage kernels request a
// 64-deep k-tile ONE barrier ahead, so every step lasts at least one L2 / Infinity-Cache round trip: with the B operand's DMA and three
// quarters of the LDS reads removed (VLNI_PK_HACK) gemm_pk_kernel still runs ~2 k cycles per k-tile where its MFMAs need 1 k - it is
// latency-bound per block, not LDS- or MFMA-bound. Here up to (NSLOT - 1) x 32 k of both operands are in flight per block.
// 64-byte LDS rows: 16-byte chunk c of row r sits in slot c ^ ((r >> 2) & 3), which makes the 16 rows of a ds_read_b128 lane group
// hit 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int lds_off32(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <bool NN, int NSLOT>
__global__ __launch_bounds__(512, 4) void gemm_pkr_kernel(GemmP p, int ntiles) {
  constexpr int HK = 32, NW = 8, SLOT = (BM + BN) * 64, D = NSLOT - 1;       // D half-steps requested ahead
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int bid = blockIdx.x, xcd = bid & 7, gx = gridDim.x >> 3;
  const int q8 = ntiles >> 3, rr = ntiles & 7;
  const int c0 = xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8;
  const int c1 = c0 + q8 + (xcd < rr ? 1 : 0);
  int tile = c0 + (bid >> 3);
  if (tile >= c1) return;
  const int S = p.K / HK;                                  // half-steps per tile
  const int total = ((c1 - 1 - tile) / gx + 1) * S;        // half-steps of this block's whole stream

  // ---- issue side: ONE LDS-DMA instruction per operand, wave and half-step (16 rows x 64 B, or 4 k-rows x 256 B of a [K,N] weight) ----
  const char* zp = (const char*)g_zero_page;
  asm volatile("" : "+s"(zp));
  const unsigned lds0 = lds_u32(dsmem);
  const char* ga;
  const char* gb;
  unsigned bstep;
  auto point_at = [&](const PkSide& s) {
    const int row = wave * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    ga = s.A + ((long)min(s.m0 + row, s.M - 1) * s.lda) * 2 + c * 16;
    if constexpr (NN) {
      const int krow = wave * 4 + (lane >> 4);
      const int cc = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
      const bool ok = s.n0 + cc * 8 < p.N;
      gb = ok ? s.B + ((long)krow * s.ldb + s.n0 + cc * 8) * 2 : zp;
      bstep = ok ? (unsigned)(HK * s.ldb * 2) : 0u;
    } else {
      gb = s.B + ((long)min(s.n0 + row, p.N - 1) * s.ldb) * 2 + c * 16;
      bstep = 64;
    }
  };
  int islot = 0;                                           // slot of the next half-step to request
  auto issue = [&]() {
    const unsigned sa = lds0 + islot * SLOT + wave * 1024;
    lds_dma16(ga, sa);
    lds_dma16(gb, sa + BM * 64);
    ga += 64;
    gb += bstep;
    islot = islot == NSLOT - 1 ? 0 : islot + 1;
  };
  const int qq = (lane & 15) >> 2, pp = lane & 3, cb = (lane >> 4) & 1;
  const int frow = 8 * h + qq, half8 = 8 * (pp & 1);
  const int chb = (wc * 64 + 16 * cb) / 8 + (pp >> 1);

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[j][x] = 0.f;
  const unsigned dseed0 = p.drop_thr ? eff_seed(p.drop_seed, p.seed_base) : 0u;
  const unsigned dseed1 = p.drop_thr ? eff_seed(p.drop_seed1, p.seed_base) : 0u;

  // the issue side runs D half-steps ahead of the compute side and crosses tile borders on its own
  PkSide cur, isd;
  pk_locate(p, tile, cur);
  isd = cur;
  point_at(isd);
  int itile = tile, is = 0, issued = 0;                    // issue-side tile, its next half-step, half-steps requested so far
  auto issue_next = [&]() {
    if (is == S) {                                         // next tile of this block (exists: issued < total)
      itile += gx;
      pk_locate(p, itile, isd);
      point_at(isd);
      is = 0;
    }
    issue();
    ++is;
    ++issued;
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (issued < total) issue_next();

  int cslot = 0, g = 0;                                    // slot and stream index of the half-step being computed
  while (true) {
    for (int s = 0; s < S; ++s, ++g) {
      // half-step g of the stream has landed when at most min(D - 1, total - 1 - g) younger half-steps (2 DMAs each) are in flight
      const int younger = total - 1 - g;
      if (younger >= D - 1) wait_vm<2 * (D - 1)>();
      else if (younger == 0) wait_vm<0>();
      else if (younger == 1) wait_vm<2>();
      else if (younger == 2) wait_vm<4>();
      else wait_vm<6>();
      __builtin_amdgcn_s_barrier();                        // everybody's share landed; everybody finished reading the slot requested next
      if (issued < total) issue_next();
      const char* As = dsmem + cslot * SLOT;
      const char* Bs = As + BM * 64;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 b[2];
        const bf16x8 a = *(const bf16x8*)(As + lds_off32(wr * 32 + r, kk * 2 + h));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (NN) b[j] = tr_frag(Bs, 16 * kk + frow, chb + 4 * j, half8);
          else b[j] = *(const bf16x8*)(Bs + lds_off32(wc * 64 + j * 32 + r, kk * 2 + h));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a, acc[j], 0, 0, 0);
      }
      cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
    }
    pk_epilogue(p, tile, cur, acc, wr, wc, r, h, dseed0, dseed1);
    if (g >= total) break;
    tile += gx;
    pk_locate(p, tile, cur);
  }
}

}  // namespace

static int gemm_check_one(int es, const void* A, long lda, const void* B, long ldb, long ldc, int M, int N, int K, bool nn = false) {
  const int epc = 16 / es;
  VLNI_CHECK(M > 0 && N > 0 && K > 0, VLNI_EINVAL, "gemm_nt: empty problem %d %d %d", M, N, K);
  VLNI_CHECK(K % epc == 0 && lda % epc == 0 && ldb % epc == 0, VLNI_EINVAL,
             "gemm_nt: K/lda/ldb (%d/%ld/%ld) must be multiples of %d", K, lda, ldb, epc);
  VLNI_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, VLNI_EINVAL, "gemm_nt: A/B must be 16-B aligned");
  VLNI_CHECK(lda >= K && ldb >= (nn ? N : K) && ldc >= N, VLNI_EINVAL, "gemm_nt: leading dims too small");
  VLNI_CHECK(!nn || (es == 2 && K % 64 == 0 && K >= 192 && N % 8 == 0), VLNI_EUNSUP,
             "gemm_nt: the [K,N] weight layout needs bf16, K %% 64 == 0, K >= 192, N %% 8 == 0 (K=%d N=%d)", K, N);
  return VLNI_OK;
}

static int gemm_group_m(int tbm) {
  static const int env = getenv("VLNI_GROUP_M") ? atoi(getenv("VLNI_GROUP_M")) : -1;
  return env >= 0 ? env : 1024 / tbm;
}

template <typename T, int NST, int WM, int WN, int MI, int NJ>
static void gemm_big_go(GemmP& p, hipStream_t st) {
  constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32, LDS = NST * (TBM + TBN) * ROWB;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<T, NST, WM, WN, MI, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr = true;
  }
  const int ntn = cdiv(p.N, TBN);
  p.tiles0 = cdiv(p.M, TBM) * ntn;
  const int tiles = p.tiles0 + (p.A1 ? cdiv(p.M1, TBM) * ntn : 0);
  p.group_m = gemm_group_m(TBM);
  hipLaunchKernelGGL((gemm_nt_big_kernel<T, NST, WM, WN, MI, NJ>), dim3(tiles), dim3(WM * WN * 64), LDS, st, p);
}

static bool gemm_pk_vec_ok(const GemmP& p) {        // every epilogue tensor allows 16-byte (8 x bf16) accesses
  auto ok = [](const void* q, long ld) { return q == nullptr || ((((uintptr_t)q) & 15) == 0 && ld % 8 == 0); };
  auto okb = [](const float* b) { return b == nullptr || (((uintptr_t)b) & 15) == 0; };
  return ok(p.C, p.ldc) && ok(p.residual, p.ldr) && ok(p.preact, p.ldp) && ok(p.dact_src, p.ldd) && okb(p.bias) &&
         (p.A1 == nullptr || (ok(p.C1, p.ldc1) && ok(p.residual1, p.ldr1) && ok(p.preact1, p.ldp1) && ok(p.dact_src1, p.ldd1) && okb(p.bias1)));
}

static int gemm_launch(int dtype, GemmP& p, int split_k, int variant, void* stream) {
  const bool nn = (variant & 16) != 0;           // B given as [K,N] (the forward weight itself): bf16 dgrad
  variant = (variant & 15) + ((variant >> 5) << 4);   // ids above 15 are passed as 32 + (id - 16), so that "+ 16" stays the layout flag
  const int es = dtype == VLNI_F32 ? 4 : 2, bk = ROWB / es;
  const int nkt = cdiv(p.K, bk);
  p.kt_per_split = cdiv(nkt, split_k);
  const int splits = cdiv(nkt, p.kt_per_split);
  const int ntn = cdiv(p.N, BN);
  p.tiles0 = cdiv(p.M, BM) * ntn;
  const int tiles = p.tiles0 + (p.A1 ? cdiv(p.M1, BM) * ntn : 0);
  dim3 grid(tiles, 1, splits);
  p.group_m = gemm_group_m(BM);
  static const bool no_glds = getenv("VLNI_NO_GLDS") != nullptr;
  const bool glds_ok = !no_glds && (p.K % bk == 0) && p.kt_per_split >= 3;
  if (variant == 0) variant = (glds_ok && (long)grid.x * grid.z <= 256) ? 3 : 1;
  if (!glds_ok) variant = 1;
  if (variant >= 6 && (splits > 1 || p.atomic_f32)) variant = 1;       // large tiles: whole-K, plain stores only
  hipStream_t st = (hipStream_t)stream;
  // persistent 128 x 128 kernel (NT: variant 14, [K,N] weights: variant 6 + 16): bf16, whole K, 8-element vector epilogue
  const bool pk_ok = dtype == VLNI_BF16 && splits == 1 && !p.atomic_f32 && p.K % 64 == 0 && p.K >= 128 && p.N % 8 == 0 && gemm_pk_vec_ok(p);
  const int ring = (!nn && variant == 15) || (nn && variant == 7) ? 4 : variant == 16 ? 5 : 0;     // persistent ring kernels: slots
  if ((variant == 14 && !nn) || (variant == 6 && nn) || ring) {
    if (pk_ok) {
      constexpr int LDS = 2 * (BM + BN) * ROWB;
      static bool attr_pk = false;
      if (!attr_pk) {
        (void)hipFuncSetAttribute((const void*)gemm_pk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_pk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_pkr_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
        (void)hipFuncSetAttribute((const void*)gemm_pkr_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
        (void)hipFuncSetAttribute((const void*)gemm_pkr_kernel<false, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16384);
        (void)hipFuncSetAttribute((const void*)gemm_pkr_kernel<true, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16384);
        attr_pk = true;
      }
      const int G = 8 * std::min(64, cdiv(tiles, 8));
      static const int hack = getenv("VLNI_PK_HACK") ? atoi(getenv("VLNI_PK_HACK")) : 0;
      if (ring == 4) {
        if (nn) hipLaunchKernelGGL((gemm_pkr_kernel<true, 4>), dim3(G), dim3(512), 4 * 16384, st, p, tiles);
        else hipLaunchKernelGGL((gemm_pkr_kernel<false, 4>), dim3(G), dim3(512), 4 * 16384, st, p, tiles);
      } else if (ring == 5) {
        if (nn) hipLaunchKernelGGL((gemm_pkr_kernel<true, 5>), dim3(G), dim3(512), 5 * 16384, st, p, tiles);
        else hipLaunchKernelGGL((gemm_pkr_kernel<false, 5>), dim3(G), dim3(512), 5 * 16384, st, p, tiles);
      } else if (hack && !nn) {
        static bool attr_h = false;
        if (!attr_h) {
          (void)hipFuncSetAttribute((const void*)gemm_pk_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
          (void)hipFuncSetAttribute((const void*)gemm_pk_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
          (void)hipFuncSetAttribute((const void*)gemm_pk_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
          (void)hipFuncSetAttribute((const void*)gemm_pk_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
          attr_h = true;
        }
        if (hack == 1) hipLaunchKernelGGL((gemm_pk_kernel<false, 1>), dim3(G), dim3(512), LDS, st, p, tiles);
        else if (hack == 2) hipLaunchKernelGGL((gemm_pk_kernel<false, 2>), dim3(G), dim3(512), LDS, st, p, tiles);
        else if (hack == 4) hipLaunchKernelGGL((gemm_pk_kernel<false, 4>), dim3(G), dim3(512), LDS, st, p, tiles);
        else hipLaunchKernelGGL((gemm_pk_kernel<false, 3>), dim3(G), dim3(512), LDS, st, p, tiles);
      } else if (nn) hipLaunchKernelGGL((gemm_pk_kernel<true>), dim3(G), dim3(512), LDS, st, p, tiles);
      else hipLaunchKernelGGL((gemm_pk_kernel<false>), dim3(G), dim3(512), LDS, st, p, tiles);
      VLNI_LAUNCH_CHECK();
      return VLNI_OK;
    }
    variant = nn ? 5 : 0;                            // not eligible: the per-launch pipelines
    if (!nn) variant = (glds_ok && (long)grid.x * grid.z <= 256) ? 3 : 1;
  }
  if (nn) {
    VLNI_CHECK(splits == 1 && !p.atomic_f32, VLNI_EUNSUP, "gemm_nt: [K,N] weight layout takes no split-K");
    if (variant < 2 || variant > 5) variant = 5;
    constexpr int ST = (BM + BN) * ROWB;
    static bool attr_nn = false;
    if (!attr_nn) {
      (void)hipFuncSetAttribute((const void*)gemm_nn_glds_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ST);
      (void)hipFuncSetAttribute((const void*)gemm_nn_glds_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST);
      (void)hipFuncSetAttribute((const void*)gemm_nn_glds_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ST);
      (void)hipFuncSetAttribute((const void*)gemm_nn_glds_kernel<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST);
      attr_nn = true;
    }
    if (variant == 2) hipLaunchKernelGGL((gemm_nn_glds_kernel<2, 4>), grid, dim3(256), 2 * ST, st, p);
    else if (variant == 3) hipLaunchKernelGGL((gemm_nn_glds_kernel<3, 4>), grid, dim3(256), 3 * ST, st, p);
    else if (variant == 4) hipLaunchKernelGGL((gemm_nn_glds_kernel<3, 8>), grid, dim3(512), 3 * ST, st, p);
    else hipLaunchKernelGGL((gemm_nn_glds_kernel<2, 8>), grid, dim3(512), 2 * ST, st, p);
  } else if (variant >= 6) {
    // 6: 256x128 tile, 3 stages; 7: 256x256 tile, 2 stages; 8: 128x256 tile, 3 stages (8 waves each)
    // 12 / 13: 192x128 / 128x192 tiles, 2 stages = 80 KiB -> TWO blocks fill the CU's 160 KiB of LDS: 17 % fewer operand bytes per
    // flop than 128x128 through the CU's vector-memory pipe (the bound here) while keeping the 2-blocks-per-CU overlap of epilogues
    // 9 / 10 / 11: SMALL tiles 64x128 / 128x64 / 64x64 (4 waves, 3 stages, 2-3 blocks per CU) for launches that do not fill one
    // round of 128x128 tiles: their time is one tile's latency, so more and shorter tiles win (history encoder, M = 2.3 k rows)
    if (dtype == VLNI_F32) {
      if (variant == 6) gemm_big_go<float, 3, 4, 2, 2, 2>(p, st);
      else if (variant == 7) gemm_big_go<float, 2, 2, 4, 4, 2>(p, st);
      else if (variant == 8) gemm_big_go<float, 3, 2, 4, 2, 2>(p, st);
      else if (variant == 9) gemm_big_go<float, 3, 2, 2, 1, 2>(p, st);
      else if (variant == 10) gemm_big_go<float, 3, 2, 2, 2, 1>(p, st);
      else if (variant == 12) gemm_big_go<float, 2, 2, 4, 3, 1>(p, st);
      else if (variant == 13) gemm_big_go<float, 2, 4, 2, 1, 3>(p, st);
      else gemm_big_go<float, 3, 2, 2, 1, 1>(p, st);
    } else {
      if (variant == 6) gemm_big_go<__bf16, 3, 4, 2, 2, 2>(p, st);
      else if (variant == 7) gemm_big_go<__bf16, 2, 2, 4, 4, 2>(p, st);
      else if (variant == 8) gemm_big_go<__bf16, 3, 2, 4, 2, 2>(p, st);
      else if (variant == 9) gemm_big_go<__bf16, 3, 2, 2, 1, 2>(p, st);
      else if (variant == 10) gemm_big_go<__bf16, 3, 2, 2, 2, 1>(p, st);
      else if (variant == 12) gemm_big_go<__bf16, 2, 2, 4, 3, 1>(p, st);
      else if (variant == 13) gemm_big_go<__bf16, 2, 4, 2, 1, 3>(p, st);
      else gemm_big_go<__bf16, 3, 2, 2, 1, 1>(p, st);
    }
  } else if (variant >= 2) {
    constexpr int ST = (BM + BN) * ROWB;
    const bool deep = variant == 3 || variant == 4, wide = variant >= 4;
    static bool attr = false;
    if (!attr) {
#define VLNI_ATTR(TT, NS, W) (void)hipFuncSetAttribute((const void*)gemm_nt_glds_kernel<TT, NS, W>, hipFuncAttributeMaxDynamicSharedMemorySize, NS * ST)
      VLNI_ATTR(float, 3, 4); VLNI_ATTR(__bf16, 3, 4); VLNI_ATTR(float, 2, 4); VLNI_ATTR(__bf16, 2, 4);
      VLNI_ATTR(float, 3, 8); VLNI_ATTR(__bf16, 3, 8); VLNI_ATTR(float, 2, 8); VLNI_ATTR(__bf16, 2, 8);
#undef VLNI_ATTR
      attr = true;
    }
#define VLNI_GO(NS, W)                                                                                       \
    do {                                                                                                     \
      if (dtype == VLNI_F32) hipLaunchKernelGGL((gemm_nt_glds_kernel<float, NS, W>), grid, dim3(W * 64), NS * ST, st, p);  \
      else hipLaunchKernelGGL((gemm_nt_glds_kernel<__bf16, NS, W>), grid, dim3(W * 64), NS * ST, st, p);       \
    } while (0)
    if (deep && !wide) VLNI_GO(3, 4);
    else if (!deep && !wide) VLNI_GO(2, 4);
    else if (deep) VLNI_GO(3, 8);
    else VLNI_GO(2, 8);
#undef VLNI_GO
  } else if (dtype == VLNI_F32) {
    hipLaunchKernelGGL(gemm_nt_kernel<float>, grid, dim3(NT), 0, st, p);
  } else {
    hipLaunchKernelGGL(gemm_nt_kernel<__bf16>, grid, dim3(NT), 0, st, p);
  }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

static bool gemm_vec_ok(int es, int N, const void* C, long ldc, const void* residual, long ldr, const void* preact, long ldp,
                        const void* dact_src, long ldd, const float* bias) {
  const uintptr_t am = (uintptr_t)(4 * es - 1);     // 4 elements: 16 B (f32) / 8 B (bf16)
  auto okp = [&](const void* q, long ld) { return q == nullptr || ((((uintptr_t)q) & am) == 0 && ld % 4 == 0); };
  return (N % 4 == 0) && okp(C, ldc) && okp(residual, ldr) && okp(preact, ldp) && okp(dact_src, ldd) &&
         (bias == nullptr || (((uintptr_t)bias) & 15) == 0);
}

// variant: 0 = choose by shape, 1 = register-staged 32-KiB kernel (4 blocks/CU), 2 = LDS-DMA 2-stage (64 KiB),
// 3 = LDS-DMA 3-stage (96 KiB), 4 / 5 = the 3- / 2-stage kernels with 8 waves per tile (2 waves per SIMD), 6 / 7 / 8 = large
// tiles 256x128 / 256x256 / 128x256 (one block per CU). All variants compute the same result; the host side may time them once per shape.
// variant + 16: B is given as [K,N] row-major (ldb >= N) - the forward weight itself, so bf16 dgrad needs no transposed weight
// copy (pipelines 2..5; bf16, K % 64 == 0, K >= 192, N % 8 == 0).
extern "C" int vlni_gemm_nt_v(int dtype, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N,
                              int K, const float* bias, int act, const void* residual, long ldr, void* preact, long ldp,
                              const void* dact_src, long ldd, int dact, float alpha, int split_k, int atomic_f32,
                              int variant, float drop_p, unsigned drop_seed, void* stream) {
  VLNI_CHECK(dtype == VLNI_F32 || dtype == VLNI_BF16, VLNI_EINVAL, "gemm_nt: bad dtype %d", dtype);
  const int es = dtype == VLNI_F32 ? 4 : 2;
  int rc = gemm_check_one(es, A, lda, B, ldb, ldc, M, N, K, (variant & 16) != 0);
  if (rc) return rc;
  VLNI_CHECK(!(atomic_f32 && (bias || act || residual || preact || dact)), VLNI_EINVAL, "gemm_nt: atomic output takes no epilogue");
  VLNI_CHECK(split_k >= 1 && (split_k == 1 || atomic_f32), VLNI_EINVAL, "gemm_nt: split_k needs atomic_f32");
  VLNI_CHECK(drop_p >= 0.f && drop_p < 1.f && !(atomic_f32 && drop_p > 0.f), VLNI_EINVAL, "gemm_nt: dropout p=%f", drop_p);
  GemmP p = {};
  p.A = (const char*)A; p.lda = lda; p.B = (const char*)B; p.ldb = ldb; p.C = (char*)C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.bias = bias; p.act = act; p.residual = (const char*)residual; p.ldr = ldr;
  p.preact = (char*)preact; p.ldp = ldp; p.dact_src = (const char*)dact_src; p.ldd = ldd; p.dact = dact;
  p.alpha = alpha; p.atomic_f32 = atomic_f32;
  p.drop_thr = drop_thr(drop_p); p.drop_seed = drop_seed; p.drop_inv = 1.0f / (1.0f - drop_p); p.seed_base = vlni_seed_base();
  p.vec_ok = gemm_vec_ok(es, N, C, ldc, residual, ldr, preact, ldp, dact_src, ldd, bias);
  return gemm_launch(dtype, p, split_k, variant, stream);
}

// Two problems of the same N, K and epilogue kind in ONE launch (arrays of 2: operands, rows, bias, residual, preact,
// dact_src, dropout seed). The language and vision streams of a cross-modal layer never fill 256 CUs on their own.
extern "C" int vlni_gemm_nt_dual(int dtype, const void* const* A, const long* lda, const void* const* B, const long* ldb,
                                 void* const* C, const long* ldc, const int* M, int N, int K, const float* const* bias, int act,
                                 const void* const* residual, const long* ldr, void* const* preact, const long* ldp,
                                 const void* const* dact_src, const long* ldd, int dact, int variant, float drop_p,
                                 const unsigned* drop_seed, void* stream) {
  VLNI_CHECK(dtype == VLNI_F32 || dtype == VLNI_BF16, VLNI_EINVAL, "gemm_nt_dual: bad dtype %d", dtype);
  const int es = dtype == VLNI_F32 ? 4 : 2;
  for (int i = 0; i < 2; ++i) {
    int rc = gemm_check_one(es, A[i], lda[i], B[i], ldb[i], ldc[i], M[i], N, K, (variant & 16) != 0);
    if (rc) return rc;
  }
  VLNI_CHECK(drop_p >= 0.f && drop_p < 1.f, VLNI_EINVAL, "gemm_nt_dual: dropout p=%f", drop_p);
  auto at = [](const void* const* arr, int i) -> const void* { return arr ? arr[i] : nullptr; };
  auto al = [](const long* arr, int i) -> long { return arr ? arr[i] : 0; };
  GemmP p = {};
  p.A = (const char*)A[0]; p.lda = lda[0]; p.B = (const char*)B[0]; p.ldb = ldb[0]; p.C = (char*)C[0]; p.ldc = ldc[0];
  p.M = M[0]; p.N = N; p.K = K; p.bias = bias ? bias[0] : nullptr; p.act = act;
  p.residual = (const char*)at(residual, 0); p.ldr = al(ldr, 0);
  p.preact = (char*)at((const void* const*)preact, 0); p.ldp = al(ldp, 0);
  p.dact_src = (const char*)at(dact_src, 0); p.ldd = al(ldd, 0); p.dact = dact; p.alpha = 1.f; p.atomic_f32 = 0;
  p.drop_thr = drop_thr(drop_p); p.drop_seed = drop_seed ? drop_seed[0] : 0; p.drop_inv = 1.0f / (1.0f - drop_p); p.seed_base = vlni_seed_base();
  p.A1 = (const char*)A[1]; p.lda1 = lda[1]; p.B1 = (const char*)B[1]; p.ldb1 = ldb[1]; p.C1 = (char*)C[1]; p.ldc1 = ldc[1];
  p.M1 = M[1]; p.bias1 = bias ? bias[1] : nullptr;
  p.residual1 = (const char*)at(residual, 1); p.ldr1 = al(ldr, 1);
  p.preact1 = (char*)at((const void* const*)preact, 1); p.ldp1 = al(ldp, 1);
  p.dact_src1 = (const char*)at(dact_src, 1); p.ldd1 = al(ldd, 1); p.drop_seed1 = drop_seed ? drop_seed[1] : 0;
  p.vec_ok = gemm_vec_ok(es, N, p.C, p.ldc, p.residual, p.ldr, p.preact, p.ldp, p.dact_src, p.ldd, p.bias) &&
             gemm_vec_ok(es, N, p.C1, p.ldc1, p.residual1, p.ldr1, p.preact1, p.ldp1, p.dact_src1, p.ldd1, p.bias1);
  return gemm_launch(dtype, p, 1, variant, stream);
}

// Diagnostic: copies the phase-cycle sums of the last VLNI_PK_HACK=4 launch of the persistent GEMM (8 x u64 per block, 768 blocks) to the host.
extern "C" int vlni_debug_pk_stamps(void* host_dst, int bytes) {
  VLNI_CHECK(host_dst && bytes > 0 && bytes <= (int)sizeof(unsigned long long) * 768 * 8, VLNI_EINVAL, "debug_pk_stamps: bytes=%d", bytes);
  VLNI_CHECK(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_pk_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess, VLNI_ELAUNCH,
             "debug_pk_stamps: hipMemcpyFromSymbol failed");
  return VLNI_OK;
}

extern "C" int vlni_gemm_nt(int dtype, const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N,
                            int K, const float* bias, int act, const void* residual, long ldr, void* preact, long ldp,
                            const void* dact_src, long ldd, int dact, float alpha, int split_k, int atomic_f32,
                            void* stream) {
  return vlni_gemm_nt_v(dtype, A, lda, B, ldb, C, ldc, M, N, K, bias, act, residual, ldr, preact, ldp, dact_src, ldd, dact,
                        alpha, split_k, atomic_f32, 0, 0.f, 0u, stream);
}

// C[N,K] += sum_s A_s[M_s,N]^T B_s[M_s,K] (bf16 operands as they lie in memory, float32 atomic accumulation, split over
// rows); colsum (optional, [N]) += column sums of all A_s (the bias gradient). nseg <= 16 row segments share lda/ldb.
// Replaces autograd's weight-gradient matmuls (one launch per parameter per episode when the segments are the T steps).
// variant: 0/1 = register-staged kernel, 2 = LDS-DMA 2-stage, 3 = 3-stage, 4 / 5 = 3- / 2-stage with 8 waves, 6 = 256 x 256 tiles (identical sums up
// to float atomics order).
static int tn_grouped_go(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb, float* C, long ldc,
                         int N, int K, float* colsum, float* part, long part_stride, int split, int variant, void* stream) {
  VLNI_CHECK(nseg >= 1 && nseg <= TN_MAXSEG, VLNI_EINVAL, "gemm_tn: nseg=%d not in 1..%d", nseg, TN_MAXSEG);
  VLNI_CHECK(N > 0 && K > 0 && split >= 1, VLNI_EINVAL, "gemm_tn: bad problem N=%d K=%d split=%d", N, K, split);
  VLNI_CHECK(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, VLNI_EINVAL, "gemm_tn: N/K/lda/ldb multiples of 8");
  TnP p;
  p.nseg = nseg; p.lda = lda; p.ldb = ldb; p.C = C; p.ldc = ldc; p.N = N; p.K = K; p.colsum = colsum;
  p.part = part; p.part_stride = part_stride;
  p.mt_start[0] = 0;
  for (int s = 0; s < nseg; ++s) {
    VLNI_CHECK(M[s] > 0, VLNI_EINVAL, "gemm_tn: empty segment %d", s);
    VLNI_CHECK(((uintptr_t)A[s] & 15) == 0 && ((uintptr_t)B[s] & 15) == 0, VLNI_EINVAL, "gemm_tn: A/B must be 16-B aligned");
    p.A[s] = (const __bf16*)A[s]; p.B[s] = (const __bf16*)B[s]; p.segM[s] = M[s];
    p.mt_start[s + 1] = p.mt_start[s] + cdiv(M[s], 64);
  }
  for (int s = nseg; s < TN_MAXSEG; ++s) { p.A[s] = nullptr; p.B[s] = nullptr; p.segM[s] = 0; p.mt_start[s + 1] = p.mt_start[nseg]; }
  const int nmt = p.mt_start[nseg];
  p.mt_per_split = cdiv(nmt, split);
  dim3 grid(cdiv(N, BM) * cdiv(K, BN), 1, cdiv(nmt, p.mt_per_split));
  hipStream_t st = (hipStream_t)stream;
  if (variant == 7) {                       // 256 x 256 tiles, ring of four 32-row half-steps
    constexpr int LDS = 4 * 4 * 32 * 256;
    static bool attr7 = false;
    if (!attr7) {
      (void)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      attr7 = true;
    }
    hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3(cdiv(N, 256) * cdiv(K, 256), 1, grid.z), dim3(512), LDS, st, p);
  } else if (variant == 6) {                // 256 x 256 tiles, 2 x 64 KiB stages
    constexpr int LDS = 2 * 4 * 64 * 256;
    static bool attr6 = false;
    if (!attr6) {
      (void)hipFuncSetAttribute((const void*)gemm_tn_big_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      attr6 = true;
    }
    hipLaunchKernelGGL((gemm_tn_big_kernel<2>), dim3(cdiv(N, 256) * cdiv(K, 256), 1, grid.z), dim3(512), LDS, st, p);
  } else if (variant >= 2 && p.mt_per_split >= 3) {
    constexpr int ST = 2 * 64 * 256;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ST);
      (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST);
      (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ST);
      (void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST);
      attr = true;
    }
    if (variant == 2) hipLaunchKernelGGL((gemm_tn_glds_kernel<2, 4>), grid, dim3(256), 2 * ST, st, p);
    else if (variant == 3) hipLaunchKernelGGL((gemm_tn_glds_kernel<3, 4>), grid, dim3(256), 3 * ST, st, p);
    else if (variant == 4) hipLaunchKernelGGL((gemm_tn_glds_kernel<3, 8>), grid, dim3(512), 3 * ST, st, p);
    else hipLaunchKernelGGL((gemm_tn_glds_kernel<2, 8>), grid, dim3(512), 2 * ST, st, p);
  } else {
    VLNI_CHECK(part == nullptr, VLNI_EUNSUP, "gemm_tn partials: needs an LDS-DMA variant (2..7) and >= 3 row tiles per split (variant=%d, %d)",
               variant, p.mt_per_split);
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, grid, dim3(NT), 0, st, p);
  }
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_gemm_tn_bf16_grouped_v(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb,
                                           float* C, long ldc, int N, int K, float* colsum, int split, int variant, void* stream) {
  return tn_grouped_go(nseg, A, B, M, lda, ldb, C, ldc, N, K, colsum, nullptr, 0, split, variant, stream);
}

extern "C" int vlni_gemm_tn_bf16_grouped_part(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb,
                                              float* part, long part_stride, int N, int K, float* colsum_part, int split, int variant,
                                              void* stream) {
  VLNI_CHECK(part && part_stride >= (long)N * K && part_stride % 4 == 0, VLNI_EINVAL, "gemm_tn partials: part_stride=%ld < N*K", part_stride);
  return tn_grouped_go(nseg, A, B, M, lda, ldb, nullptr, 0, N, K, colsum_part, part, part_stride, split, variant, stream);
}

namespace {
struct PartEntry { float* dst; const float* part; long n4, stride4; int split, blk0; };
// dst[i] += sum_z part[z][i], many (dst, part) pairs per launch; block -> entry by binary search over the entries' first blocks;
// a block covers 1024 float4 of one entry
__global__ __launch_bounds__(256) void reduce_parts_kernel(const PartEntry* __restrict__ tab, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PartEntry e = tab[lo];
  const long base = (long)(blockIdx.x - e.blk0) * 1024 + threadIdx.x;
  f32x4* dst = (f32x4*)e.dst;
  const f32x4* part = (const f32x4*)e.part;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = base + u * 256;
    if (i >= e.n4) break;
    f32x4 a = dst[i];
    for (int z = 0; z < e.split; ++z) a += part[(long)z * e.stride4 + i];
    dst[i] = a;
  }
}
}  // namespace

extern "C" int vlni_reduce_parts(const void* table, int n_entries, int n_blocks, void* stream) {
  VLNI_CHECK(table && n_entries > 0 && n_blocks > 0, VLNI_EINVAL, "reduce_parts: n_entries=%d n_blocks=%d", n_entries, n_blocks);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, (const PartEntry*)table, n_entries);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_gemm_tn_bf16_grouped(int nseg, const void* const* A, const void* const* B, const int* M, long lda, long ldb,
                                         float* C, long ldc, int N, int K, float* colsum, int split, void* stream) {
  return vlni_gemm_tn_bf16_grouped_v(nseg, A, B, M, lda, ldb, C, ldc, N, K, colsum, split, 0, stream);
}

extern "C" int vlni_gemm_tn_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                                 float* colsum, int split, void* stream) {
  return vlni_gemm_tn_bf16_grouped(1, &A, &B, &M, lda, ldb, C, ldc, N, K, colsum, split, stream);
}
