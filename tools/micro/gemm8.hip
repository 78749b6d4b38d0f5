// Stand-alone bench of the 256 x 256 x 64 "8-phase" bf16 NT GEMM (C[M,N] = A[M,K] W[N,K]^T) before it moves into csrc/gemm_impl.inc.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm8 gemm8.hip ; run: ./gemm8
//
// Geometry: 8 waves as 2 (M) x 4 (N), a wave owns 128 rows x 64 columns = acc[8][4] of v_mfma_f32_16x16x32_bf16; two 64-KiB LDS buffers,
// each = four 16-KiB half-tiles (A-h0, A-h1, B-h0, B-h1: 128 rows x 128 B, 16-B chunks XOR-swizzled with (row >> 1) & 7).
// A k-tile is 4 phases (one 64 x 32 quadrant of the wave's tile each: 16 MFMAs), every phase = {fragment reads, ONE half-tile of
// LDS-DMA, barrier, MFMAs, barrier}; the two wave groups (wr = 0 / 1) run one barrier apart, so one group's MFMAs always sit beside
// the other group's reads and DMA issue. vmcnt is counted (6 = three half-tiles stay in flight), never 0 inside the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

typedef __bf16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct P8 {
  const char* A; long lda; const char* B; long ldb; char* C; long ldc;
  int M, N, K; const float* bias; int ntiles, group_m;
};
struct Side { int m0, n0; };

__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_u32(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ void locate(const P8& p, int w, Side& s) {
  constexpr int TM = 256, TN = 256;
  const int ntn = (p.N + TN - 1) / TN;
  if (p.group_m <= 1) { s.m0 = (w / ntn) * TM; s.n0 = (w % ntn) * TN; return; }
  const int ntm = (p.M + TM - 1) / TM;
  const int per = p.group_m * ntn;
  const int g = w / per, idx = w - g * per;
  const int first = g * p.group_m, gsz = min(ntm - first, p.group_m);
  s.m0 = (first + idx % gsz) * TM;
  s.n0 = (idx / gsz) * TN;
}

#define SB() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { SB(); __builtin_amdgcn_s_barrier(); SB(); } while (0)

template <int STAMP>
__global__ __launch_bounds__(512) void gemm8_kernel(P8 p, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fq = lane >> 4;
  const int bid = blockIdx.x, xcd = bid & 7, gx = gridDim.x >> 3;
  const int q8 = p.ntiles >> 3, rr = p.ntiles & 7;
  const int c0 = xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8;
  const int c1 = c0 + q8 + (xcd < rr ? 1 : 0);
  int tile = c0 + (bid >> 3);
  if (tile >= c1) return;
  const int nk = p.K / 64;                                   // even, >= 2

  // ---- LDS-DMA side: half-tile X in {A0, A1, B0, B1}, instruction i in {0, 1}: LDS rows (i * 8 + wave) * 8 + (lane >> 3) ----
  const unsigned lds0 = lds_u32(dsmem);
  const char* pA[2][2];
  const char* pB[2][2];
  auto point = [&](const Side& s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rho = (i * 8 + wave) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((rho >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ra = s.m0 + (rho >> 6) * 128 + h * 64 + (rho & 63);
        const int rb = s.n0 + (rho >> 5) * 64 + h * 32 + (rho & 31);
        pA[h][i] = p.A + (long)min(ra, p.M - 1) * p.lda * 2 + c * 16;
        pB[h][i] = p.B + (long)min(rb, p.N - 1) * p.ldb * 2 + c * 16;
      }
    }
  };
  // half-tile offsets inside a 64-KiB buffer: A-h0 0, A-h1 16 KiB, B-h0 32 KiB, B-h1 48 KiB
#define STAGE(buf, ARR, h, OFF)                                              \
  do {                                                                       \
    const unsigned dst_ = lds0 + (buf) * 65536 + (OFF) + wave * 1024;        \
    lds_dma16(ARR[h][0], dst_);                                              \
    lds_dma16(ARR[h][1], dst_ + 8192);                                       \
    ARR[h][0] += 128;                                                        \
    ARR[h][1] += 128;                                                        \
  } while (0)
#define STAGE_A0(buf) STAGE(buf, pA, 0, 0)
#define STAGE_A1(buf) STAGE(buf, pA, 1, 16384)
#define STAGE_B0(buf) STAGE(buf, pB, 0, 32768)
#define STAGE_B1(buf) STAGE(buf, pB, 1, 49152)

  // ---- fragment side ----
  const int sw = (fr >> 1) & 7;
  const unsigned aoff = (wr * 64 + fr) * 128 + ((fq ^ sw) << 4);             // + i * 16384 + mb * 2048, ^ 64 for the second k32
  const unsigned boff = 32768 + (wc * 32 + fr) * 128 + ((fq ^ sw) << 4);     // + j * 16384 + nb * 2048
  h16x8 a[4][2], b0[2][2], b1[2][2];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto rd_a = [&](int buf, int i) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) a[mb][kk] = *(const h16x8*)(dsmem + buf * 65536 + i * 16384 + mb * 2048 + (aoff ^ (kk * 64)));
  };
  auto rd_b = [&](int buf, int j, h16x8 (&b)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) b[nb][kk] = *(const h16x8*)(dsmem + buf * 65536 + j * 16384 + nb * 2048 + (boff ^ (kk * 64)));
  };
  auto mma = [&](int i, int j, h16x8 (&b)[2][2]) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[i * 4 + mb][j * 2 + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nb][kk], a[mb][kk], acc[i * 4 + mb][j * 2 + nb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  unsigned long long t_begin = 0, t_loop = 0, t_epi = 0;
  if constexpr (STAMP) t_begin = __builtin_amdgcn_s_memtime();

  Side cur, nxt;
  locate(p, tile, cur);
  point(cur);
  // prologue: k-tile 0 whole (buffer 0) and k-tile 1 without its A-h1 (buffer 1); the A-h1 pointer stays one k-tile behind the others
  STAGE_B0(0); STAGE_A0(0); STAGE_B1(0); STAGE_A1(0);
  STAGE_B0(1); STAGE_A0(1); STAGE_B1(1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  BAR();

  // one k-tile = 4 phases. d: buffer of this k-tile. SWITCH: behind phase 1's DMA the pointers move to the next output tile.
#define KTILE(d, SWITCH)                                                                                       \
  do {                                                                                                         \
    /* phase 1: quadrant (0,0) */                                                                              \
    rd_b(d, 0, b0); SB(); rd_a(d, 0); SB();                                                                    \
    STAGE_A1((d) ^ 1);                                                                                        \
    if (SWITCH) { if (more) { locate(p, ntile, nxt); point(nxt); } else point(cur); }                          \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                         \
    BAR(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB();                                            \
    mma(0, 0, b0); BAR();                                                                                      \
    /* phase 2: quadrant (0,1) */                                                                              \
    rd_b(d, 1, b1); SB();                                                                                      \
    STAGE_B0(d);                                                                                              \
    BAR(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB();                                            \
    mma(0, 1, b1); BAR();                                                                                      \
    /* phase 3: quadrant (1,1) */                                                                              \
    rd_a(d, 1); SB();                                                                                          \
    STAGE_A0(d);                                                                                              \
    BAR(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB();                                            \
    mma(1, 1, b1); BAR();                                                                                      \
    /* phase 4: quadrant (1,0), registers only; the wait certifies the NEXT k-tile (read one phase later) */   \
    STAGE_B1(d);                                                                                              \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                           \
    BAR(); SB();                                                                                               \
    mma(1, 0, b0); BAR();                                                                                      \
  } while (0)

  while (true) {
    const int ntile = tile + gx;
    const bool more = ntile < c1;
    unsigned long long tl = 0;
    if constexpr (STAMP) tl = __builtin_amdgcn_s_memtime();
    if (wr == 1) BAR();                                      // the two wave groups run one barrier apart
    for (int kt = 0; kt < nk; kt += 2) {
      const bool last = kt + 2 >= nk;
      KTILE(0, last);
      KTILE(1, false);
    }
    if (wr == 0) BAR();                                      // together again for the epilogue
    unsigned long long te = 0;
    if constexpr (STAMP) { te = __builtin_amdgcn_s_memtime(); t_loop += te - tl; }

    // ---- epilogue from registers: lane (fr, fq) holds C[row mb*16 + fr][cols nb*16 + fq*4 .. +4]; v_permlane16_swap pairs the
    //      16-lane rows fq = 2a / 2a+1 so that a lane owns 8 consecutive columns (16-byte stores, 64 contiguous bytes per row and instruction)
    {
      __bf16* C = (__bf16*)p.C;
      f32x4 bv[2][2];
      int colj[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        colj[j] = cur.n0 + wc * 64 + j * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
        const int cc = min(colj[j], p.N - 8);
        bv[j][0] = p.bias ? *(const f32x4*)(p.bias + cc) : f32x4{0.f, 0.f, 0.f, 0.f};
        bv[j][1] = p.bias ? *(const f32x4*)(p.bias + cc + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int im = 0; im < 8; ++im) {
        const int row = cur.m0 + wr * 128 + (im >> 2) * 64 + (im & 3) * 16 + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = colj[j];
          float v[8];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float x = acc[im][j * 2][u], y = acc[im][j * 2 + 1][u];
            const auto s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
            v[u] = __uint_as_float(s2[0]) + bv[j][0][u];
            v[4 + u] = __uint_as_float(s2[1]) + bv[j][1][u];
          }
          h16x8 o;
#pragma unroll
          for (int u = 0; u < 8; ++u) o[u] = (__bf16)v[u];
          if (col < p.N && row < p.M) *(h16x8*)(C + (long)row * p.ldc + col) = o;
          acc[im][j * 2] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[im][j * 2 + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if constexpr (STAMP) t_epi += __builtin_amdgcn_s_memtime() - te;
    if (!more) break;
    tile = ntile;
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (STAMP) {
    if (lane == 0) {
      unsigned long long* o = stamps + ((long)blockIdx.x * 8 + wave) * 4;
      o[0] = __builtin_amdgcn_s_memtime() - t_begin; o[1] = t_loop; o[2] = t_epi; o[3] = 0;
    }
  }
}


// ======================================================================================================================
// 256 x 128 x 64 tile, THREE 48-KiB LDS buffers (k-tile s in buffer s % 3): 8 waves as 4 (M) x 2 (N), a wave owns 64 x 64 = acc[4][4];
// a k-tile is 2 phases of 16 MFMAs (rows 0-31 / 32-63 of the wave's tile); the two wave groups (waves 0-3 / 4-7) run one barrier apart.
// k-tile s + 2 is requested during k-tile s (3 + 3 LDS-DMA instructions per wave); s_waitcnt vmcnt(6) in phase 2 certifies k-tile s + 1.
// The buffer of s + 2 was last read in k-tile s - 1: no ordering subtleties beyond retiring phase 2's reads before its first barrier.
template <int STAMP, int NLX>     // NLX: dedicated loader waves (0 = the 8 computing waves request their own data); 5 = 4 loaders (A) + computing waves (B)
__global__ __launch_bounds__(512 + 64 * (NLX == 5 ? 4 : NLX)) void gemm8h_kernel(P8 p, unsigned long long* stamps) {
  constexpr int NL = NLX == 5 ? 4 : NLX;
  constexpr bool HYB = NLX == 5;
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  constexpr int TM = 256, TN = 128, BUF = (TM + TN) * 128;     // 49152
  constexpr int NI = HYB ? 8 : (NL ? 48 / NL : 6);             // LDS-DMA instructions per requesting wave and k-tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = NL > 0 && wave >= 8;
  const int wr = (wave & 7) >> 1, wc = wave & 1, grp = (wave & 7) >> 2, fr = lane & 15, fq = lane >> 4;
  const int bid = blockIdx.x, xcd = bid & 7, gx = gridDim.x >> 3;
  const int q8 = p.ntiles >> 3, rr = p.ntiles & 7;
  const int c0 = xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8;
  const int c1 = c0 + q8 + (xcd < rr ? 1 : 0);
  int tile = c0 + (bid >> 3);
  if (tile >= c1) return;
  const int nk = p.K / 64;

  auto locate = [&](int w, Side& s) __attribute__((always_inline)) {
    const int ntn = (p.N + TN - 1) / TN;
    if (p.group_m <= 1) { s.m0 = (w / ntn) * TM; s.n0 = (w % ntn) * TN; return; }
    const int ntm = (p.M + TM - 1) / TM;
    const int per = p.group_m * ntn;
    const int g = w / per, idx = w - g * per;
    const int first = g * p.group_m, gsz = min(ntm - first, p.group_m);
    s.m0 = (first + idx % gsz) * TM;
    s.n0 = (idx / gsz) * TN;
  };
  // request side. The k-tile's 48 KiB are 48 pieces of 1 KiB = 8 LDS rows: piece q (0..31 A, 32..47 B) = rows (q % 32 or q - 32) * 8 ...
  // Requesting wave v (0..NV-1, NV = 8 or NL) owns pieces v + NV * j, j = 0 .. NI-1.
  constexpr int NV = NL ? NL : 8;
  const int v = NL ? (wave - 8) : wave;
  const unsigned lds0 = lds_u32(dsmem);
  const char* pg[NI];
  auto point = [&](const Side& s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int q = v + NV * j;
      const int rho = (q & 31) * 8 + (lane >> 3);                 // (B pieces: q - 32 < 16, so q & 31 = q - 32)
      const int c = (lane & 7) ^ ((rho >> 1) & 7);
      if (NV * j < 32 && q < 32) pg[j] = p.A + (long)min(s.m0 + rho, p.M - 1) * p.lda * 2 + c * 16;
      else pg[j] = p.B + (long)min(s.n0 + rho, p.N - 1) * p.ldb * 2 + c * 16;
    }
  };
#define H_DMA(j, boff)                                                                                          \
  do {                                                                                                          \
    asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"                           \
                 : : "v"(pg[j]), "s"(lds0 + (boff) + v * 1024), "n"(NV * (j) * 1024) : "memory", "scc");        \
    pg[j] += 128;                                                                                               \
  } while (0)

  unsigned long long t_begin = 0, t_loop = 0, t_epi = 0;
  if constexpr (STAMP) t_begin = __builtin_amdgcn_s_memtime();
  Side cur, nxt;
  locate(tile, cur);
  unsigned rbuf = 0, wbuf = 2 * BUF;
  int left = nk - 2;

  if (loader || NL == 0) {
    point(cur);
#pragma unroll
    for (int j = 0; j < NI; ++j) H_DMA(j, 0);
#pragma unroll
    for (int j = 0; j < NI; ++j) H_DMA(j, BUF);
    if constexpr (NI == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (NI == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  }
  const char* pgb[2];
  auto pointb = [&](const Side& s2) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rho = (wave + 8 * j) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((rho >> 1) & 7);
      pgb[j] = p.B + (long)min(s2.n0 + rho, p.N - 1) * p.ldb * 2 + c * 16;
    }
  };
#define HB_DMA(j, boff)                                                                                         \
  do {                                                                                                          \
    asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"                           \
                 : : "v"(pgb[j]), "s"(lds0 + (boff) + wave * 1024), "n"(32768 + (j) * 8192) : "memory", "scc");    \
    pgb[j] += 128;                                                                                              \
  } while (0)
  if (HYB && !loader) {
    pointb(cur);
    HB_DMA(0, 0); HB_DMA(1, 0); HB_DMA(0, BUF); HB_DMA(1, BUF);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  BAR();

  if (loader) {
    // ---- loader waves: the request stream only, in step with wave group 0's barriers ----
    while (true) {
      const int ntile = tile + gx;
      const bool more = ntile < c1;
      for (int kt = 0; kt < nk; ++kt) {
        if (left == 0) {
          int nt_ = more ? ntile : tile;
          asm volatile("" : "+s"(nt_));
          locate(nt_, nxt);
          point(nxt);
          left = nk;
        }
        --left;
#pragma unroll
        for (int j = 0; j < NI / 2; ++j) H_DMA(j, wbuf);
        BAR(); BAR();
#pragma unroll
        for (int j = NI / 2; j < NI; ++j) H_DMA(j, wbuf);
        if constexpr (NI == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (NI == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        BAR(); BAR();
        wbuf = wbuf == 2 * BUF ? 0 : wbuf + BUF;
      }
      BAR();                                                   // (group 0's line-up barrier)
      if (!more) break;
      tile = ntile;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  const int sw = (fr >> 1) & 7;
  const unsigned aoff = (wr * 64 + fr) * 128 + ((fq ^ sw) << 4);             // + mb * 2048, ^ 64 for the second k32
  const unsigned boffr = 32768 + (wc * 64 + fr) * 128 + ((fq ^ sw) << 4);    // + nb * 2048
  h16x8 a[2][2], b[4][2];
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  while (true) {
    const int ntile = tile + gx;
    const bool more = ntile < c1;
    unsigned long long tl = 0;
    if constexpr (STAMP) tl = __builtin_amdgcn_s_memtime();
    if (grp == 1) BAR();
    for (int kt = 0; kt < nk; ++kt) {
      if constexpr (NL == 0) {
        if (left == 0) {
          int nt_ = more ? ntile : tile;
          asm volatile("" : "+s"(nt_));
          locate(nt_, nxt);
          point(nxt);
          left = nk;
        }
        --left;
      }
      if constexpr (HYB) {
        if (left == 0) {
          int nt_ = more ? ntile : tile;
          asm volatile("" : "+s"(nt_));
          locate(nt_, nxt);
          pointb(nxt);
          left = nk;
        }
        --left;
      }
      const char* rb = dsmem + rbuf;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b[nb][kk] = *(const h16x8*)(rb + nb * 2048 + (boffr ^ (kk * 64)));
      SB();
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a[mb][kk] = *(const h16x8*)(rb + mb * 2048 + (aoff ^ (kk * 64)));
      SB();
      if constexpr (NL == 0) { H_DMA(0, wbuf); H_DMA(1, wbuf); H_DMA(2, wbuf); }
      if constexpr (HYB) HB_DMA(0, wbuf);
      BAR(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nb][kk], a[mb][kk], acc[mb][nb], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      BAR();
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a[mb][kk] = *(const h16x8*)(rb + (mb + 2) * 2048 + (aoff ^ (kk * 64)));
      SB();
      if constexpr (NL == 0) {
        H_DMA(3, wbuf); H_DMA(4, wbuf); H_DMA(5, wbuf);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
      if constexpr (HYB) { HB_DMA(1, wbuf); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BAR(); SB();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) acc[mb + 2][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nb][kk], a[mb][kk], acc[mb + 2][nb], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      BAR();
      rbuf = rbuf == 2 * BUF ? 0 : rbuf + BUF;
      wbuf = wbuf == 2 * BUF ? 0 : wbuf + BUF;
    }
    if (grp == 0) BAR();
    unsigned long long te = 0;
    if constexpr (STAMP) { te = __builtin_amdgcn_s_memtime(); t_loop += te - tl; }
    {
      __bf16* C = (__bf16*)p.C;
      f32x4 bv[2][2];
      int colj[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        colj[j] = cur.n0 + wc * 64 + j * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
        const int cc = min(colj[j], p.N - 8);
        bv[j][0] = p.bias ? *(const f32x4*)(p.bias + cc) : f32x4{0.f, 0.f, 0.f, 0.f};
        bv[j][1] = p.bias ? *(const f32x4*)(p.bias + cc + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int im = 0; im < 4; ++im) {
        const int row = cur.m0 + wr * 64 + im * 16 + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = colj[j];
          float v8[8];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float x = acc[im][j * 2][u], y = acc[im][j * 2 + 1][u];
            const auto s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
            v8[u] = __uint_as_float(s2[0]) + bv[j][0][u];
            v8[4 + u] = __uint_as_float(s2[1]) + bv[j][1][u];
          }
          h16x8 o;
#pragma unroll
          for (int u = 0; u < 8; ++u) o[u] = (__bf16)v8[u];
          if (col < p.N && row < p.M) *(h16x8*)(C + (long)row * p.ldc + col) = o;
          acc[im][j * 2] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[im][j * 2 + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if constexpr (STAMP) t_epi += __builtin_amdgcn_s_memtime() - te;
    if (!more) break;
    tile = ntile;
    if constexpr (NL == 0) cur = nxt;
    else locate(tile, cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (STAMP) {
    if (lane == 0) {
      unsigned long long* o = stamps + ((long)blockIdx.x * 8 + wave) * 4;
      o[0] = __builtin_amdgcn_s_memtime() - t_begin; o[1] = t_loop; o[2] = t_epi; o[3] = 0;
    }
  }
}

// ---- reference: one thread per output, float accumulation in k order ----
__global__ void ref_kernel(const __bf16* A, const __bf16* B, float* C, int M, int N, int K, const float* bias) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * N) return;
  const int m = idx / N, n = idx % N;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(long)m * K + k] * (float)B[(long)n * K + k];
  C[idx] = s + (bias ? bias[n] : 0.f);
}
__global__ void fill_kernel(__bf16* x, long n, unsigned seed) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 0x9E3779B1u + seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  x[i] = (__bf16)(((float)(h & 0xFFFF) / 32768.0f) - 1.0f);
}

static int g_kind = 0;          // 0: 256 x 256, 1..4: 256 x 128 forms
static int tiles_of(int M, int N) { return ((M + 255) / 256) * ((N + (g_kind ? 127 : 255)) / (g_kind ? 128 : 256)); }
static void launch(const P8& p0, hipStream_t st, bool stamp, unsigned long long* stamps) {
  P8 p = p0;
  p.ntiles = tiles_of(p.M, p.N);
  static const int gm = getenv("GROUP_M") ? atoi(getenv("GROUP_M")) : 4;
  p.group_m = gm;
  const int G = 8 * std::min(32, (p.ntiles + 7) / 8);
  if (g_kind == 1) {
    if (stamp) hipLaunchKernelGGL((gemm8h_kernel<1, 0>), dim3(G), dim3(512), 147456, st, p, stamps);
    else hipLaunchKernelGGL((gemm8h_kernel<0, 0>), dim3(G), dim3(512), 147456, st, p, stamps);
    return;
  }
  if (g_kind == 2) {
    if (stamp) hipLaunchKernelGGL((gemm8h_kernel<1, 4>), dim3(G), dim3(768), 147456, st, p, stamps);
    else hipLaunchKernelGGL((gemm8h_kernel<0, 4>), dim3(G), dim3(768), 147456, st, p, stamps);
    return;
  }
  if (g_kind == 4) {
    if (stamp) hipLaunchKernelGGL((gemm8h_kernel<1, 5>), dim3(G), dim3(768), 147456, st, p, stamps);
    else hipLaunchKernelGGL((gemm8h_kernel<0, 5>), dim3(G), dim3(768), 147456, st, p, stamps);
    return;
  }
  if (g_kind == 3) {
    if (stamp) hipLaunchKernelGGL((gemm8h_kernel<1, 8>), dim3(G), dim3(1024), 147456, st, p, stamps);
    else hipLaunchKernelGGL((gemm8h_kernel<0, 8>), dim3(G), dim3(1024), 147456, st, p, stamps);
    return;
  }
  if (stamp) hipLaunchKernelGGL(gemm8_kernel<1>, dim3(G), dim3(512), 131072, st, p, stamps);
  else hipLaunchKernelGGL(gemm8_kernel<0>, dim3(G), dim3(512), 131072, st, p, stamps);
}

int main(int argc, char** argv) {
  hipFuncSetAttribute((const void*)gemm8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)gemm8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)gemm8h_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void*)gemm8h_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void*)gemm8h_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void*)gemm8h_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void*)gemm8h_kernel<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void*)gemm8h_kernel<0, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void*)gemm8h_kernel<1, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void*)gemm8h_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  g_kind = argc > 1 ? atoi(argv[1]) : 0;
  const char* names[] = {"256 x 256, 2 buffers", "256 x 128, 3 buffers", "256 x 128, 3 buffers, 4 loader waves", "256 x 128, 3 buffers, 8 loader waves", "256 x 128, 3 buffers, 4 loader waves (A) + computing waves request B"};
  printf("kernel: %s\n", names[g_kind]);
  struct Shape { int M, N, K; bool check; };
  std::vector<Shape> shapes = {
      {1000, 520, 256, true}, {256, 256, 192, true}, {4096, 4096, 4096, true}, {8192, 8192, 8192, false},
      {48384, 768, 768, true}, {48384, 2304, 768, false}, {48384, 3072, 768, false}, {48384, 768, 3072, false},
      {8256, 768, 768, true}, {8256, 2304, 768, false}, {8256, 3072, 768, false}, {8256, 768, 3072, false},
      {5120, 768, 768, false}, {5120, 2304, 768, false}, {5120, 3072, 768, false}, {5120, 768, 3072, false},
      {2304, 768, 768, false}, {2304, 2304, 768, false}, {2304, 3072, 768, false}, {2304, 768, 3072, false}};
  unsigned long long* stamps;
  hipMalloc(&stamps, 256 * 8 * 4 * 8);
  for (const Shape& s : shapes) {
    __bf16 *A, *B, *C;
    float *R, *bias;
    hipMalloc(&A, (long)s.M * s.K * 2);
    hipMalloc(&B, (long)s.N * s.K * 2);
    hipMalloc(&C, (long)s.M * s.N * 2);
    hipMalloc(&bias, s.N * 4);
    fill_kernel<<<(int)(((long)s.M * s.K + 255) / 256), 256>>>(A, (long)s.M * s.K, 1u);
    fill_kernel<<<(int)(((long)s.N * s.K + 255) / 256), 256>>>(B, (long)s.N * s.K, 2u);
    std::vector<float> hb(s.N);
    for (int i = 0; i < s.N; ++i) hb[i] = 0.25f * (float)((i * 37) % 17 - 8);
    hipMemcpy(bias, hb.data(), s.N * 4, hipMemcpyHostToDevice);
    hipMemset(C, 0xff, (long)s.M * s.N * 2);
    P8 p = {(const char*)A, s.K, (const char*)B, s.K, (char*)C, s.N, s.M, s.N, s.K, bias, 0, 0};
    launch(p, 0, false, stamps);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 1; }
    double maxerr = 0, maxref = 0;
    long bad = 0;
    if (s.check) {
      hipMalloc(&R, (long)s.M * s.N * 4);
      ref_kernel<<<(int)(((long)s.M * s.N + 255) / 256), 256>>>(A, B, R, s.M, s.N, s.K, bias);
      std::vector<float> hr((long)s.M * s.N);
      std::vector<__bf16> hc((long)s.M * s.N);
      hipMemcpy(hr.data(), R, hr.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < hr.size(); ++i) {
        const double d = fabs((double)(float)hc[i] - hr[i]);
        maxerr = std::max(maxerr, d);
        maxref = std::max(maxref, (double)fabs(hr[i]));
        if (!(d <= 0.01 * fabs(hr[i]) + 0.02)) ++bad;
      }
      hipFree(R);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = s.M >= 8192 && s.N >= 8192 ? 5 : 20;
    for (int i = 0; i < 3; ++i) launch(p, 0, false, stamps);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch(p, 0, false, stamps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / iters, tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
    // stamped run: cycles in the main loop / the epilogue, per block
    launch(p, 0, true, stamps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hs(256 * 8 * 4);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    const int ntiles = tiles_of(s.M, s.N), G = 8 * std::min(32, (ntiles + 7) / 8);
    double tot = 0, lp = 0, ep = 0;
    for (int b = 0; b < G; ++b) { tot += hs[(b * 8) * 4]; lp += hs[(b * 8) * 4 + 1]; ep += hs[(b * 8) * 4 + 2]; }
    const double tiles_per_block = (double)ntiles / G;
    printf("M %6d N %5d K %5d: %8.1f us %7.1f TF/s | tiles %5d (%.2f/block) | per block: total %.0f cyc, loop %.0f (%.0f per k-tile), epilogue %.0f per tile",
           s.M, s.N, s.K, us, tf, ntiles, tiles_per_block, tot / G, lp / G, lp / G / tiles_per_block / (s.K / 64), ep / G / tiles_per_block);
    if (s.check) printf(" | check: max err %.4f (max |ref| %.1f) bad %ld", maxerr, maxref, bad);
    printf("\n");
    fflush(stdout);
    hipFree(A); hipFree(B); hipFree(C); hipFree(bias);
  }
  return 0;
}
