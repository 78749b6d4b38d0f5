"""LayerNorm forward / backward launch times at the step's row counts (single and dual) against their byte bound at 5 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vln_imagine_amd import ops
dt, H = torch.bfloat16, 768
r = lambda n: (torch.randn(n, H, device="cuda")).to(dt)
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g, b = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
for rows in ((5504, 2624), (5120, 0), (2304, 0)):
    xs = tuple(r(n) for n in rows if n)
    dys = tuple(r(n) for n in rows if n)
    if len(xs) == 2:
        (y0, m0, r0), (y1, m1, r1) = ops.ln_fwd2(xs, (g, g), (b, b), 1e-12)
        f = t(lambda: ops.ln_fwd2(xs, (g, g), (b, b), 1e-12))
        bw = t(lambda: ops._ln_bwd_to2(dys, xs, (g, g), (b, b), (m0, m1), (r0, r1), (True, True)))
    else:
        y0, m0, r0 = ops.ln_fwd(xs[0], g, b, 1e-12)
        f = t(lambda: ops.ln_fwd(xs[0], g, b, 1e-12))
        bw = t(lambda: ops.ln_bwd(dys[0], xs[0], g, m0, r0))
    n = sum(rows)
    print(f"rows {rows}: fwd {f:5.1f} us ({n * H * 4 / 5e6 / f:4.0%} of byte bound)  bwd {bw:5.1f} us ({n * H * 6 / 5e6 / bw:4.0%})", flush=True)
