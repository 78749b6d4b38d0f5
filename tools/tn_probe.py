"""Micro-benchmark of the grouped transposing-read wgrad kernels in partials mode (plain-store row splits, the mode the product
runs them in; the batched reduction is not included): TF/s vs variant x split for the episode-level shapes."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_imagine_amd import _lib

def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

st = torch.cuda.current_stream().cuda_stream
for (N, K, M, nseg) in [(3072, 768, 2752, 6), (768, 3072, 2752, 6), (2304, 768, 2752, 6)]:
    dys = [(torch.randn(M, N, device="cuda") * 0.1).bfloat16() for _ in range(nseg)]
    xs = [(torch.randn(M, K, device="cuda") * 0.5).bfloat16() for _ in range(nseg)]
    out = torch.zeros(N, K, device="cuda"); cs = torch.zeros(N, device="cuda")
    pa = (ctypes.c_void_p * nseg)(*[d.data_ptr() for d in dys]); pb = (ctypes.c_void_p * nseg)(*[x.data_ptr() for x in xs])
    pm = (ctypes.c_int * nseg)(*[M] * nseg)
    fl = 2.0 * N * K * M * nseg
    res = []
    for variant, splits in ((5, (4, 8, 12)), (6, (7, 9)), (7, (2, 4, 7, 9, 14))):
        for split in splits:
            part = torch.empty(16 * (N * K + N), device="cuda")
            us = t(lambda: _lib.call("vlni_gemm_tn_bf16_grouped_part", nseg, pa, pb, pm, N, K, part.data_ptr(), N * K, N, K, part.data_ptr() + 4 * 16 * N * K, split, variant, st))
            res.append(f"v{variant}s{split}:{us:4.0f}us/{fl/us/1e6:3.0f}TF")
    print(f"N={N:5d} K={K:5d} M={M}x{nseg}: " + "  ".join(res), flush=True)
