"""Would grouping the language-side (M=5504) and vision-side (M=2432) GEMMs into one launch pay? sum of two vs one launch of 7936 rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_imagine_amd import ops

def t(fn, reps=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

dt = torch.bfloat16
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    res = {}
    for M in (5504, 2432, 7936):
        a = (torch.randn(M, K, device="cuda") * 0.5).to(dt); out = torch.empty(M, N, device="cuda", dtype=dt)
        res[M] = t(lambda: ops.gemm_nt(a, b, out=out))
    print(f"N={N} K={K}: 5504 {res[5504]:.1f}us + 2432 {res[2432]:.1f}us = {res[5504]+res[2432]:.1f}us   vs 7936 rows {res[7936]:.1f}us", flush=True)
