"""Per-variant timing of vlni_gemm_nt_v on the hot-path shapes (HIP events, back-to-back launches; variants interleaved in rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_imagine_amd import ops

dt = torch.bfloat16
shapes = [(5120, 3072, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 768, 768), (5120, 3072, 768), (5120, 768, 3072),
          (2304, 768, 768), (2304, 3072, 768), (2304, 768, 3072), (30720, 3072, 768), (30720, 768, 3072), (30720, 768, 768), (4096, 4096, 4096)]
variants = tuple(int(v) for v in os.environ.get("VARIANTS", "1,2,3,4,5,6,7,8,9,10,11,12,13").split(","))
for (M, N, K) in shapes:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=dt); z = torch.empty_like(out)
    for mode in ("plain", "gelu+preact"):
        best = {v: 1e9 for v in variants}
        for rnd in range(3):
            for v in variants:
                args = (v, a, b, out, bias, 1, None, z, None, 0, 1.0, 1, False, M, N, K) if mode != "plain" else \
                       (v, a, b, out, None, 0, None, None, None, 0, 1.0, 1, False, M, N, K)
                for _ in range(2): ops._gemm_call(*args)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops._gemm_call(*args)
                e1.record(); torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / 10 * 1e3)
        fl = 2.0 * M * N * K
        us3 = None
        if mode == "plain":
            for _ in range(3): torch.matmul(a, b.t())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): torch.matmul(a, b.t())
            e1.record(); torch.cuda.synchronize(); us3 = e0.elapsed_time(e1) / 10 * 1e3
        print(f"M={M:5d} N={N:4d} K={K:4d} {mode:12s} " + " ".join(f"v{v}:{best[v]:6.1f}us/{fl/best[v]/1e6:4.0f}TF" for v in variants)
              + (f" | hipblaslt {us3:6.1f}us/{fl/us3/1e6:4.0f}TF" if us3 else ""), flush=True)
