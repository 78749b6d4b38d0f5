"""Round 6: in-kernel cycle stamps of the 256 x 256 8-phase GEMM (variant 15) on the episode-long shapes of the batched backward, per
epilogue kind: cycles per 64-deep k-tile, epilogue cycles per tile, share of the epilogue in a tile - wave 0 (wave row 0) and wave 4
(wave row 1) of every block, medians over blocks. Needs a build with -DVLNI_DIAG -DVLNI_P8_STAMP (tools/build_variant.sh); the
stamped build's own run time is not a measurement. usage: VLNI_LIB_PATH=... python tools/p8_stamps.py [tag]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vln_imagine_amd import _lib, ops  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "p8"
dt = torch.bfloat16
r = lambda *s, sc=0.5: (torch.randn(*s, device="cuda") * sc).to(dt)
M = int(os.environ.get("M", 49536))
for (N, K, kind) in ((768, 768, "plain"), (768, 768, "res"), (768, 2304, "res"), (768, 3072, "res"), (3072, 768, "dact3"), (3072, 768, "gelu3"),
                     (2304, 768, "plain"), (768, 768, "resdrop")):
    a, w = r(M, K), r(N, K, sc=0.05)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    bias = torch.randn(N, device="cuda")
    res, z = r(M, N), r(M, N, sc=1.0)
    kw = {"plain": (bias, 0, None, None, None, 0, None), "res": (None, 0, res, None, None, 0, None), "dact3": (None, 0, None, None, z, 3, None),
          "gelu3": (bias, 3, None, z, None, 0, None), "resdrop": (bias, 0, res, None, None, 0, (0.1, 1234))}[kind]
    bias_, act, res_, pre, dsrc, dact, drop = kw

    def go():
        ops._gemm_call(15, a, w, out, bias_, act, res_, pre, dsrc, dact, 1.0, 1, False, M, N, K, drop)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    buf = np.zeros((768, 8), np.uint64)
    _lib.call("vlni_debug_pk_stamps", buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    tiles = -(-M // 256) * -(-N // 256)
    msg = []
    for g in (0, 1):
        sq = buf[g::2][:min(256, tiles)].astype(np.float64)
        sq = sq[sq[:, 5] > 0]
        kt = np.median(sq[:, 0] / sq[:, 5])
        ep = np.median(sq[:, 4] / sq[:, 3])
        msg.append(f"wave row {g}: k-tile {kt:5.0f} cyc, epilogue {ep:6.0f} cyc/tile = {ep / (ep + kt * K / 64):.2f} of a tile, kernel {np.median(sq[:, 6]):8.0f} cyc "
                   f"({np.median(sq[:, 3]):.0f} tiles/block)")
    print(f"{TAG} M={M} N={N:4d} K={K:4d} {kind:7s} tiles {tiles:4d} ({tiles / 256:.2f} rounds) {us:6.1f} us (stamped build) | " + " | ".join(msg), flush=True)
