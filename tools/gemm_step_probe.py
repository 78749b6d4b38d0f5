"""Per-variant timing of the dual-problem GEMM launches of one HAMT navigation step at the bench's batch (language stream 64 x 86 rows
+ vision stream 64 x 40 rows), forward (NT) and dgrad (weight as [K, N], transposing reads), with the epilogues the step uses.
HIP events around 10 back-to-back launches, variants interleaved in rounds (cdna_hip_programming.md rule 24); torch.matmul
(hipBLASLt) on the concatenated rows as the vendor reference for the plain contraction. Usage: python tools/gemm_step_probe.py [variants...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vln_imagine_amd import ops  # noqa: E402

dt = torch.bfloat16
T = int(os.environ.get("T", "1"))                       # T > 1: the row counts of T time-batched / episode-batched steps
M0, M1 = int(os.environ.get("M0", T * 64 * 86)), int(os.environ.get("M1", T * 64 * 40))   # DUET step 3: M0=448 (map) M1=1184 (viewpoint)
NT_VARIANTS = tuple(int(v) for v in os.environ.get("NT_VARIANTS", "5,12,13,14,15,32").split(",") if not (int(v) == 15 and M0 + M1 < 6144))
NN_VARIANTS = tuple(int(v) for v in os.environ.get("NN_VARIANTS", "4,5,6").split(","))
ROUNDS = int(os.environ.get("ROUNDS", "3"))


def rnd(*shape, s=0.5):
    return (torch.randn(*shape, device="cuda") * s).to(dt)


GRAPH = os.environ.get("GRAPH", "0") == "1"           # time n launches replayed from ONE hipGraph: short launches (< 20 us) are host-bound otherwise


def time_call(fn, n=10):
    for _ in range(2):
        fn()
    if GRAPH:
        n = 20
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(nn, N, K, kind):
    a = (rnd(M0, K), rnd(M1, K))
    if nn:
        w = (rnd(K, N, s=0.05), rnd(K, N, s=0.05))
        b = (ops.KN(w[0]), ops.KN(w[1]))
    else:
        w = (rnd(N, K, s=0.05), rnd(N, K, s=0.05))
        b = w
    bias = (torch.randn(N, device="cuda"), torch.randn(N, device="cuda"))
    res = (rnd(M0, N), rnd(M1, N))
    z = (rnd(M0, N, s=1.0), rnd(M1, N, s=1.0))
    kw = {"plain": dict(bias=bias), "res": dict(bias=bias, residual=res), "gelu": dict(bias=bias, act=1, preact=z),
          "dact": dict(dact_src=z, dact=1), "dres": dict(residual=res),
          # the kinds a 16-bit training step launches (round 4): GELU + stored GELU', multiply by the stored derivative, residual + dropout
          "gelu3": dict(bias=bias, act=3, preact=z), "dact3": dict(dact_src=z, dact=3), "resdrop": dict(bias=bias, residual=res, drop=(0.1, (11, 12)))}[kind]
    variants = NN_VARIANTS if nn else NT_VARIANTS
    best = {v: 1e9 for v in variants}
    saved = (ops.GEMM_VARIANTS, ops.NN_VARIANTS)
    try:
        for _ in range(ROUNDS):
            for v in variants:
                ops.GEMM_VARIANTS, ops.NN_VARIANTS = (v,), (v,)
                ops._GEMM_BEST.clear()
                best[v] = min(best[v], time_call(lambda: ops.gemm_nt2(a, b, **kw)))
    finally:
        ops.GEMM_VARIANTS, ops.NN_VARIANTS = saved
        ops._GEMM_BEST.clear()
    fl = 2.0 * (M0 + M1) * N * K
    extra = ""
    if nn:                                   # the same dgrad through the NT kernels on a W^T copy (what launches of >= NT_LONG_ROWS rows use)
        wt = (w[0].t().contiguous(), w[1].t().contiguous())
        try:
            for v in (14, 15, 32):
                ops.GEMM_VARIANTS = (v,)
                ops._GEMM_BEST.clear()
                t = min(time_call(lambda: ops.gemm_nt2(a, wt, **kw)) for _ in range(ROUNDS))
                extra += f" | W^T v{v}:{t:6.1f}us/{fl / t / 1e6:4.0f}TF"
        finally:
            ops.GEMM_VARIANTS = saved[0]
            ops._GEMM_BEST.clear()
    acat = torch.cat(a, 0)
    wm = w[0] if nn else w[0].t()
    us_lib = time_call(lambda: torch.matmul(acat, wm))
    tag = "NN" if nn else "NT"
    print(f"{tag} rows {M0}+{M1} N={N:4d} K={K:4d} {kind:7s} " + " ".join(f"v{v}:{best[v]:6.1f}us/{fl / best[v] / 1e6:4.0f}TF" for v in variants)
          + extra + f" | hipblaslt(plain) {us_lib:6.1f}us/{fl / us_lib / 1e6:4.0f}TF", flush=True)
    return fl, min(best.values()), {v: best[v] for v in variants}


def main():
    tot = {}
    if os.environ.get("EPILOGUES"):           # the same contraction under every epilogue kind: what bias / GELU + pre-activation store / residual cost
        N, K = (int(v) for v in os.environ["EPILOGUES"].split("x"))
        for kind in ("plain", "res", "gelu"):
            run(False, N, K, kind)
        for kind in ("plain", "dact", "dres"):
            run(True, N, K, kind)
        return
    K3 = os.environ.get("STEP_KINDS", "0") == "1"             # the epilogue kinds of the 16-bit train-mode step instead of the round-3 ones
    for nn, N, K, kind, count in ((False, 2304, 768, "plain", 2), (False, 768, 768, "resdrop" if K3 else "res", 2), (False, 3072, 768, "gelu3" if K3 else "gelu", 1),
                                  (False, 768, 3072, "resdrop" if K3 else "res", 1),
                                  (True, 3072, 768, "dact3" if K3 else "dact", 1), (True, 768, 3072, "dres", 1), (True, 768, 2304, "dres", 2), (True, 768, 768, "plain", 2)):
        fl, us, per = run(nn, N, K, kind)
        for v, t in per.items():
            key = ("NN" if nn else "NT", v)
            f0, t0 = tot.get(key, (0.0, 0.0))
            tot[key] = (f0 + fl * count, t0 + t * count)
    print("per cross-modal layer (counts as in one layer: 2 x QKV, 2 x O, FFN1, FFN2 and their dgrads):")
    for (tag, v), (f, t) in sorted(tot.items()):
        print(f"  {tag} v{v}: {t:8.1f} us  {f / t / 1e6:6.0f} TF/s  ({f / t / 1e6 / 2500:.3f} of 2.5 PF)")


if __name__ == "__main__":
    main()
