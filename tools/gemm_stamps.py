"""Where a k-step of the persistent GEMM spends its cycles: runs the stamped diagnostic build (VLNI_PK_HACK=4, s_memtime around the
phases of wave 0 of every block; cdna_hip_programming.md section 7 'In-kernel stamps') on the dual-problem launches of a step and prints
per-phase cycles per k-step (median over blocks). The stamped build's own run time is not a measurement (its fences forbid overlaps).
Needs the diagnostic library: VLNI_DIAG=1 python -m vln_imagine_amd.build --force
usage: VLNI_PK_HACK=4 python tools/gemm_stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vln_imagine_amd import _lib, ops  # noqa: E402

assert os.environ.get("VLNI_PK_HACK") == "4", "run with VLNI_PK_HACK=4"
dt = torch.bfloat16
M0, M1 = 64 * 86, 64 * 40
r = lambda *s, sc=0.5: (torch.randn(*s, device="cuda") * sc).to(dt)
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a = (r(M0, K), r(M1, K))
    w = (r(N, K, sc=0.05), r(N, K, sc=0.05))
    bias = (torch.randn(N, device="cuda"), torch.randn(N, device="cuda"))
    ops.GEMM_VARIANTS = (14,)
    ops._GEMM_BEST.clear()
    for _ in range(5):
        ops.gemm_nt2(a, w, bias=bias)
    torch.cuda.synchronize()
    buf = np.zeros((768, 8), np.uint64)
    _lib.call("vlni_debug_pk_stamps", buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    tiles = ((M0 + 127) // 128 + (M1 + 127) // 128) * (N // 128)
    nb = 8 * min(64, -(-tiles // 8))
    st = buf[:nb].astype(np.float64)
    st = st[st[:, 5] > 0]
    per = st[:, :4] / st[:, 5:6]
    med = np.median(per, 0)
    print(f"N={N:4d} K={K:4d} tiles {tiles:4d} blocks {nb}: per k-step (cycles, median over blocks): dma-wait {med[0]:6.0f}  barrier {med[1]:6.0f}  "
          f"dma-issue {med[2]:6.0f}  reads+mfma {med[3]:6.0f}  | sum {med.sum():6.0f};  epilogue/tile {np.median(st[:, 4] / (st[:, 5] / (K // 64))):7.0f}  "
          f"cold start {np.median(st[:, 7]):6.0f}  kernel {np.median(st[:, 6]):8.0f} cycles, k-steps/block {np.median(st[:, 5]):.0f}")
