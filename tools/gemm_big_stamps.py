"""Phase cycles per k-step of the large-tile NT kernels (stamped diagnostic builds, variants 36-38 = 256x128 / 256x256 / 192x128 tiles):
wave 0 of every block, median over blocks. The stamped build's own run time is not a measurement. Needs the diagnostic library:
VLNI_DIAG=1 python -m vln_imagine_amd.build --force ; usage: python tools/gemm_big_stamps.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vln_imagine_amd import _lib, ops  # noqa: E402

dt = torch.bfloat16
r = lambda *s, sc=0.5: (torch.randn(*s, device="cuda") * sc).to(dt)
for (M, N, K) in ((4096, 4096, 4096), (8192, 768, 3072), (8192, 3072, 768), (30720, 3072, 768)):
    a, w = r(M, K), r(N, K, sc=0.05)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    for v, name, tm, tn in ((36, "256x128", 256, 128), (37, "256x256", 256, 256), (38, "192x128", 192, 128)):
        for _ in range(3):
            ops._gemm_call(v, a, w, out, None, 0, None, None, None, 0, 1.0, 1, False, M, N, K)
        torch.cuda.synchronize()
        buf = np.zeros((768, 8), np.uint64)
        _lib.call("vlni_debug_pk_stamps", buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
        tiles = -(-M // tm) * -(-N // tn)
        st = buf[:min(tiles, 768)].astype(np.float64)
        st = st[st[:, 5] > 0]
        med = np.median(st[:, :4] / st[:, 5:6], 0)
        print(f"M={M:5d} N={N:4d} K={K:4d} {name} tiles {tiles:5d}: per k-step (cycles): dma-wait {med[0]:6.0f}  barrier {med[1]:6.0f}  dma-issue {med[2]:6.0f}  "
              f"reads+mfma {med[3]:6.0f} | sum {med.sum():6.0f};  epilogue {np.median(st[:, 4]):7.0f}  kernel {np.median(st[:, 6]):8.0f} cycles", flush=True)
