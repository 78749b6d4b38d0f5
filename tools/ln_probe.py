"""LayerNorm forward / backward launch times at the rows of a step against their byte bound (x, dy read, dx written once).
Usage: python tools/ln_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vln_imagine_amd import ops  # noqa: E402

dt, H = torch.bfloat16, 768


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


def mk(rows):
    x = torch.randn(rows, H, device="cuda").to(dt)
    dy = torch.randn(rows, H, device="cuda").to(dt)
    g, b = torch.randn(H, device="cuda"), torch.randn(H, device="cuda")
    y, m, r = ops.ln_fwd(x, g, b, 1e-12)
    return x, dy, g, b, m, r


for rows in (2368, 5120, 8256, 13824, 16512, 33024, 49536):
    x, dy, g, b, m, r = mk(rows)
    dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    tf = timeit(lambda: ops.ln_fwd(x, g, b, 1e-12))
    tb = timeit(lambda: ops.ln_bwd(dy, x, g, m, r, dgamma=dg, dbeta=db))
    tbd = timeit(lambda: ops.ln_bwd(dy, x, g, m, r, dgamma=dg, dbeta=db, drop=(0.1, 1234)))
    by = rows * H * 2
    print(f"rows {rows:6d}: fwd {tf:6.1f} us {2 * by / tf / 1e6:5.2f} TB/s | bwd {tb:6.1f} us {3 * by / tb / 1e6:5.2f} TB/s | bwd+drop {tbd:6.1f} us "
          f"{4 * by / tbd / 1e6:5.2f} TB/s", flush=True)
for r0, r1 in ((5504, 2752), (33024, 16512)):
    a, c = mk(r0), mk(r1)
    xs, dys, gs, bs, ms, rs = ((a[i], c[i]) for i in (0, 1, 2, 3, 4, 5))
    for t in gs + bs:
        t.grad = None
    tf = timeit(lambda: ops.ln_fwd2(xs, gs, bs, 1e-12)) if hasattr(ops, "ln_fwd2") else float("nan")
    tb = timeit(lambda: ops._ln_bwd_to2(dys, xs, gs, bs, ms, rs, (True, True)))
    tbd = timeit(lambda: ops._ln_bwd_to2(dys, xs, gs, bs, ms, rs, (True, True), drop=(0.1, (11, 12))))
    by = (r0 + r1) * H * 2
    print(f"dual {r0}+{r1}: fwd {tf:6.1f} us {2 * by / tf / 1e6:5.2f} TB/s | bwd {tb:6.1f} us {3 * by / tb / 1e6:5.2f} TB/s | bwd+drop {tbd:6.1f} us "
          f"{4 * by / tbd / 1e6:5.2f} TB/s", flush=True)
