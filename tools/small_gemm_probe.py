"""Kernel-duration floor of the step's SMALL forward / dgrad GEMM launches (DUET: 1-3 k rows per launch, K = 768 / 3072): every NT pipeline on
the shapes the autotuner saw, 20 launches inside a captured hipGraph (so the host's launch cost is out of the number), beside the vendor
library's kernel on the same shape. usage: python tools/small_gemm_probe.py [duet|hamt]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vln_imagine_amd import ops  # noqa: E402

fam = sys.argv[1] if len(sys.argv) > 1 else "duet"
args = argparse.Namespace(batch=32 if fam == "duet" else 64, T=6, L=80, V=37, I=6)
dev = torch.device("cuda")
w = bench.Workload(fam, args, False, dev, torch.bfloat16, batch=args.batch, tag="probe")
w.model.train()
from vln_imagine_amd.train import FlatTrainer  # noqa: E402
tr = FlatTrainer(w.model)
for _ in range(2):
    tr.zero_grad()
    w.run(criterion=ops.cross_entropy_sum, mode="taped")["loss"].backward()
    tr.step()
torch.cuda.synchronize()
seen = {}
for key, v in ops._GEMM_BEST.items():
    if len(key) == 10:       # dual: (dt, M0, M1, N, K, act, dact, res, pre, kn)
        M, N, K = key[1] + key[2], key[3], key[4]
    else:
        M, N, K = key[1], key[2], key[3]
    if isinstance(M, int) and M <= 4096:
        seen.setdefault((M, N, K), set()).add(v)
print(f"{len(seen)} small shapes", flush=True)


def graph_time(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


dt = torch.bfloat16
variants = tuple(ops.GEMM_VARIANTS) + (ops.P8H_VARIANT,)
for (M, N, K), picked in sorted(seen.items()):
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    res = {}
    for v in variants:
        try:
            res[v] = graph_time(lambda: ops._gemm_call(v, a, b, out, None, 0, None, None, None, 0, 1.0, 1, False, M, N, K))
        except Exception as e:
            res[v] = float("nan")
    bt = b.t()
    vend = graph_time(lambda: torch.matmul(a, bt, out=out))
    order = sorted((t, v) for v, t in res.items() if t == t)
    print(f"M={M:5d} N={N:4d} K={K:4d} picked {sorted(picked)} | best " + " ".join(f"v{v}:{t:5.1f}" for t, v in order[:4])
          + f" | vendor {vend:5.1f} us", flush=True)
