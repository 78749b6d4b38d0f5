// LDS-DMA (global_load_lds_dwordx4) acceptance rate of one CU versus the shape of the 1-KiB piece an instruction moves:
// 64 lanes x 16 B as  4 rows x 256 B,  8 rows x 128 B,  16 rows x 64 B  or  64 rows x 16 B  of an L2-resident buffer.
// Every wave of every block issues ITERS instructions into its own LDS KiB; cycles per instruction and bytes/clk/CU are printed.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip ; run: ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory");
}

template <int ROWB>   // bytes of one row piece: 256, 128, 64 or 16
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ buf, long row_stride, int rows_total, int iters,
                                                  unsigned long long* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = ROWB / 16;                       // lanes per row piece
  const int r_in = lane / LPR, c = lane % LPR;
  const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * 1024;
  const int rows_per_instr = 64 / LPR;
  unsigned long long t0 = 0, t1 = 0;
  int row = (blockIdx.x * 8 + wave) * rows_per_instr % rows_total;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; ++i) {
    const char* src = buf + (long)((row + r_in) % rows_total) * row_stride + c * 16;
    lds_dma16(src, dst);
    row += 977;                                        // walk the buffer (stays L2-resident: rows_total * row_stride <= 2 MiB)
    if (row >= rows_total) row -= rows_total;
    if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int ROWB>
static void run(const char* buf, int waves, int blocks_per_cu) {
  const int iters = 2048, rows_total = 2048;           // 2048 rows x 1 KiB stride = 2 MiB: inside one XCD's 4-MiB L2
  const int nblk = 256 * blocks_per_cu;
  unsigned long long* out;
  hipMalloc(&out, sizeof(unsigned long long) * nblk * 8);
  hipFuncSetAttribute((const void*)dma_kernel<ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep)
    hipLaunchKernelGGL(dma_kernel<ROWB>, dim3(nblk), dim3(waves * 64), 8192 * 8 / blocks_per_cu > 65536 ? 65536 : 65536 / blocks_per_cu, 0, buf, 1024L, rows_total, iters, out);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nblk * 8);
  hipMemcpy(h.data(), out, sizeof(unsigned long long) * nblk * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  int n = 0;
  for (int b = 0; b < nblk; ++b)
    for (int w = 0; w < waves; ++w) { sum += (double)h[b * 8 + w]; ++n; }
  const double cyc = sum / n;                          // cycles one wave needed for its `iters` instructions
  const double per_cu_instr = (double)iters * waves * blocks_per_cu;
  printf("piece %3d B x %2d rows, %d waves/block, %d blocks/CU: %.1f cycles per instruction per wave, CU accepts 1 KiB per %.1f cycles = %.1f B/clk\n",
         ROWB, 1024 / ROWB, waves, blocks_per_cu, cyc / iters, cyc / per_cu_instr, 1024.0 * per_cu_instr / cyc);
  hipFree(out);
}

int main() {
  char* buf;
  hipMalloc(&buf, 4 << 20);
  hipMemset(buf, 1, 4 << 20);
  for (int waves : {4, 8})
    for (int bpc : {1, 2}) {
      run<256>(buf, waves, bpc);
      run<128>(buf, waves, bpc);
      run<64>(buf, waves, bpc);
      run<16>(buf, waves, bpc);
    }
  return 0;
}
