"""Micro-benchmark of vlni_gemm_nt over the hot-path shapes (HIP-event timing, 20 reps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_imagine_amd import ops

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us

dt = torch.bfloat16
shapes = [(4096, 4096, 4096), (8192, 8192, 1024), (5504, 768, 768), (5504, 2304, 768), (5504, 3072, 768), (5504, 768, 3072),
          (2432, 768, 768), (2432, 3072, 768), (2432, 768, 3072), (384, 512, 768), (128, 128, 768), (128, 128, 64), (128,128,6144)]
for (M, N, K) in shapes:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt); z = torch.empty_like(out)
    us0 = t(lambda: ops.gemm_nt(a, b, out=out))
    us1 = t(lambda: ops.gemm_nt(a, b, out=out, bias=bias, residual=res))
    us2 = t(lambda: ops.gemm_nt(a, b, out=out, bias=bias, act=1, preact=z))
    us3 = t(lambda: torch.matmul(a, b.t()))
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d}  plain {us0:8.1f}us {fl/us0/1e6:7.1f}TF | +bias+res {us1:8.1f}us | +gelu+preact {us2:8.1f}us | torch(hipblaslt) {us3:8.1f}us {fl/us3/1e6:7.1f}TF")
# wgrad-like: split-K atomics
for (N, K, M) in [(768, 768, 5504), (3072, 768, 5504), (768, 3072, 5504), (2304, 768, 5504)]:
    a = (torch.randn(N, M, device="cuda") * 0.1).to(dt); b = (torch.randn(K, M, device="cuda") * 0.5).to(dt)
    out = torch.zeros(N, K, device="cuda")
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    for split in (1, 2, 4, max(1, min(M // 64, 512 // tiles))):
        us = t(lambda: ops.gemm_nt(a, b, out=out, split_k=split, atomic=True))
        print(f"wgrad N={N} K={K} M={M} split={split:3d}: {us:8.1f}us {2.0*N*K*M/us/1e6:7.1f}TF")
