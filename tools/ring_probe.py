"""Round 6: the ring weight-gradient kernel (variant 7, partials mode) at the step's episode-long shapes for ONE build of the library
(VLNI_LIB_PATH selects an A/B build made by tools/build_variant.sh with -DVLNI_RING_V=n): us per launch at the row splits around one
round of 256 x 256 tiles, a numerics check against torch on the first shape, and - when the build carries -DVLNI_DIAG -DVLNI_RING_STAMP -
wave 0's cycle sums per half-step (wait for own LDS-DMA / barrier / request issue / fragment reads + MFMAs) and the store epilogue.
usage: VLNI_LIB_PATH=... python tools/ring_probe.py [tag]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vln_imagine_amd import _lib  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("VLNI_LIB_PATH", "product"))
STAMPS = os.environ.get("STAMPS", "0") == "1"


def t(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


st = torch.cuda.current_stream().cuda_stream
SHAPES = [(768, 768, (5504,) * 6), (768, 768, (2752,) * 6), (768, 768, (5504, 2752) * 6), (2304, 768, (5504,) * 6), (2304, 768, (2752,) * 6),
          (3072, 768, (5504,) * 6), (768, 3072, (5504,) * 6), (3072, 768, (2752,) * 6), (768, 3072, (2752,) * 6)]
tot = {}
first = True
for (N, K, segs) in SHAPES:
    nseg = len(segs)
    dys = [(torch.randn(m, N, device="cuda") * 0.1).bfloat16() for m in segs]
    xs = [(torch.randn(m, K, device="cuda") * 0.5).bfloat16() for m in segs]
    pa = (ctypes.c_void_p * nseg)(*[d.data_ptr() for d in dys])
    pb = (ctypes.c_void_p * nseg)(*[x.data_ptr() for x in xs])
    pm = (ctypes.c_int * nseg)(*segs)
    nmt = sum((m + 63) // 64 for m in segs)
    fl = 2.0 * N * K * sum(segs)
    t256 = -(-N // 256) * -(-K // 256)
    s1 = max(1, round(252 / t256))
    res = []
    for split in sorted({s1, max(1, (3 * s1) // 4)}):
        per = -(-nmt // split)
        eff = -(-nmt // per)
        part = torch.empty(eff * (N * K + N), device="cuda")

        def go():
            _lib.call("vlni_gemm_tn_h16_grouped_part", 1, nseg, pa, pb, pm, N, K, part.data_ptr(), N * K, N, K,
                      part.data_ptr() + 4 * eff * N * K, split, 7, st)
        us = t(go)
        res.append(f"s{split}:{us:5.0f}us/{fl / us / 1e6:4.0f}TF")
        key = f"s{'1' if split == s1 else '3/4'}"
        f0, t0 = tot.get(key, (0.0, 0.0))
        tot[key] = (f0 + fl, t0 + us)
        if split == s1:                # numerics of this build: sum of the partials against torch
            got = part[:eff * N * K].view(eff, N, K).sum(0)
            ref = sum(d.float().t() @ x.float() for d, x in zip(dys, xs))
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            res.append(f"relerr {err:.1e}")
            assert err < 2e-3, err
        if STAMPS and split == s1:
            torch.cuda.synchronize()
            buf = np.zeros((768, 8), np.uint64)
            _lib.call("vlni_debug_pk_stamps", buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
            sq = buf[:min(768, t256 * eff)].astype(np.float64)
            sq = sq[sq[:, 5] > 0]
            med = np.median(sq[:, :4] / sq[:, 5:6], 0)
            res.append(f"[per half-step: wait {med[0]:.0f} barrier {med[1]:.0f} issue {med[2]:.0f} reads+mfma {med[3]:.0f} = {med.sum():.0f} cyc; "
                       f"epilogue {np.median(sq[:, 4]):.0f}, kernel {np.median(sq[:, 6]):.0f} cyc over {np.median(sq[:, 5]):.0f} half-steps]")
    print(f"{TAG} N={N:5d} K={K:5d} rows={sum(segs):6d} ({nseg} seg) tiles {t256}: " + "  ".join(res), flush=True)
for k, (f, us) in tot.items():
    print(f"{TAG} TOTAL {k}: {us:7.0f} us  {f / us / 1e6:5.0f} TF/s", flush=True)
