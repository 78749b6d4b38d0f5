"""Round 5: the grouped weight-gradient kernels (partials mode, the mode the step runs them in) at the bench step's episode-long
shapes, per (variant, row split), for the tile -> XCD mapping selected by VLNI_TN_XCD (1 = row splits kept together per XCD,
0 = the mapping of rounds 1-4). The switch is read once per process: run the probe twice.
usage: VLNI_TN_XCD=0|1 python tools/tn_xcd_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vln_imagine_amd import _lib  # noqa: E402


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
print("VLNI_TN_XCD =", os.environ.get("VLNI_TN_XCD", "1"))
# (N, K, rows per segment, segments): language / vision / both streams of a cross-modal layer over T = 6 steps, one text-encoder layer
SHAPES = [(768, 768, (5504,) * 6), (768, 768, (2752,) * 6), (768, 768, (5504, 2752) * 6), (2304, 768, (5504,) * 6), (2304, 768, (2752,) * 6),
          (3072, 768, (5504,) * 6), (768, 3072, (5504,) * 6), (3072, 768, (2752,) * 6), (768, 3072, (2752,) * 6), (2304, 768, (5120,))]
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
    t128 = -(-N // 128) * -(-K // 128)
    s1 = max(1, round(252 / t256))
    res = []
    for variant, splits in ((5, sorted({max(1, round(504 / t128)), max(1, round(252 / t128))})),
                            (6, sorted({s1, max(1, (3 * s1) // 4)})), (7, sorted({s1, max(1, (3 * s1) // 4), max(1, s1 // 2), 2 * s1}))):
        for split in splits:
            if nmt // split < 3:
                continue
            per = -(-nmt // split)
            eff = -(-nmt // per)
            part = torch.empty(eff * (N * K + N), device="cuda")
            us = t(lambda: _lib.call("vlni_gemm_tn_h16_grouped_part", 1, nseg, pa, pb, pm, N, K, part.data_ptr(), N * K, N, K,
                                     part.data_ptr() + 4 * eff * N * K, split, variant, st))
            res.append(f"v{variant}s{split}:{us:4.0f}us/{fl / us / 1e6:4.0f}TF")
    print(f"N={N:5d} K={K:5d} rows={sum(segs):6d} ({nseg} seg): " + "  ".join(res), flush=True)
