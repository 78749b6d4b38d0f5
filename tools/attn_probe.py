"""Time of the attention launches of one HAMT navigation step at the bench's batch (B = 64, 12 heads of 64): the bidirectional
cross-attention pair and the two streams' self-attention as dual launches, the text encoder's self-attention as a single launch.
Prints microseconds and the fraction of the launch's byte bound (q, k, v read + out written; backward: + out, dout read, dq, dk, dv
written) at 5 TB/s. usage: python tools/attn_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vln_imagine_amd import ops  # noqa: E402

dt, B, H = torch.bfloat16, int(os.environ.get("B", "64")), 768        # B=384: the episode-batched backward of the bench (T x B)
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def dual(name, Sq, Sk):
    q = tuple(r(B * s, 3 * H) for s in Sq)
    kv = tuple(r(B * s, 3 * H) for s in Sk)
    km = tuple(torch.zeros(B, s, device="cuda") for s in Sk)
    qs, ks, vs = tuple(t[:, :H] for t in q), tuple(t[:, H:2 * H] for t in kv), tuple(t[:, 2 * H:] for t in kv)
    (o0, l0), (o1, l1) = ops.attn_fwd2(qs, ks, vs, B, Sq, Sk, km)
    fwd = timeit(lambda: ops.attn_fwd2(qs, ks, vs, B, Sq, Sk, km))
    do = (torch.randn_like(o0), torch.randn_like(o1))
    dq = tuple(torch.empty_like(t) for t in q)
    dkv = tuple(torch.empty_like(t) for t in kv)
    bwd = timeit(lambda: ops.attn_bwd2(qs, ks, vs, (o0, o1), do, (l0, l1), tuple(t[:, :H] for t in dq), tuple(t[:, H:2 * H] for t in dkv),
                                       tuple(t[:, 2 * H:] for t in dkv), B, Sq, Sk, km))
    bf = sum(B * (sq * 2 + sk * 2) * H * 2 for sq, sk in zip(Sq, Sk))
    bb = sum(B * (sq * 4 + sk * 4) * H * 2 for sq, sk in zip(Sq, Sk))
    print(f"{name:34s} Sq {Sq} Sk {Sk}: fwd {fwd:6.1f} us ({bf / 5e6 / fwd:4.0%} of byte bound)  bwd {bwd:6.1f} us ({bb / 5e6 / bwd:4.0%})", flush=True)


def single(name, Sq, Sk):
    q, kv = r(B * Sq, 3 * H), r(B * Sk, 3 * H)
    km = torch.zeros(B, Sk, device="cuda")
    qs, ks, vs = q[:, :H], kv[:, H:2 * H], kv[:, 2 * H:]
    o, l = ops.attn_fwd(qs, ks, vs, B, Sq, Sk, km)
    fwd = timeit(lambda: ops.attn_fwd(qs, ks, vs, B, Sq, Sk, km))
    do, dq, dkv = torch.randn_like(o), torch.empty_like(q), torch.empty_like(kv)
    bwd = timeit(lambda: ops.attn_bwd(qs, ks, vs, o, do, l, dq[:, :H], dkv[:, H:2 * H], dkv[:, 2 * H:], B, Sq, Sk, km))
    bf, bb = B * (Sq * 2 + Sk * 2) * H * 2, B * (Sq * 4 + Sk * 4) * H * 2
    print(f"{name:34s} Sq {Sq} Sk {Sk}: fwd {fwd:6.1f} us ({bf / 5e6 / fwd:4.0%} of byte bound)  bwd {bwd:6.1f} us ({bb / 5e6 / bwd:4.0%})", flush=True)


dual("cross-attention pair (step 3)", (86, 41), (41, 86))
dual("self-attention, both streams", (86, 41), (86, 41))
dual("last layer, [CLS] query + vision", (1, 41), (86, 41))
single("text encoder self-attention", 80, 80)
single("history panorama encoder", 36, 36)


def layout(name, Sq, Sk, Bx):
    """The same work with every head's rows CONTIGUOUS (nh = 1, B x 12 'samples' of 64 columns, row stride 128 B) against the model's
    layout (heads = 128-byte column slices of [rows, 2304] rows, row stride 4608 B): what the strided 128-byte pieces cost."""
    q, kv = r(Bx * Sq, 3 * H), r(Bx * Sk, 3 * H)
    km = torch.zeros(Bx, Sk, device="cuda")
    qs, ks, vs = q[:, :H], kv[:, H:2 * H], kv[:, 2 * H:]
    o, l = ops.attn_fwd(qs, ks, vs, Bx, Sq, Sk, km)
    f1 = timeit(lambda: ops.attn_fwd(qs, ks, vs, Bx, Sq, Sk, km))
    do, dq, dkv = torch.randn_like(o), torch.empty_like(q), torch.empty_like(kv)
    b1 = timeit(lambda: ops.attn_bwd(qs, ks, vs, o, do, l, dq[:, :H], dkv[:, H:2 * H], dkv[:, 2 * H:], Bx, Sq, Sk, km))
    B2 = Bx * 12
    q2, k2, v2 = r(B2 * Sq, 64), r(B2 * Sk, 64), r(B2 * Sk, 64)
    km2 = torch.zeros(B2, Sk, device="cuda")
    o2, l2 = ops.attn_fwd(q2, k2, v2, B2, Sq, Sk, km2, nh=1)
    f2 = timeit(lambda: ops.attn_fwd(q2, k2, v2, B2, Sq, Sk, km2, nh=1))
    do2, dq2, dk2, dv2 = torch.randn_like(o2), torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
    b2 = timeit(lambda: ops.attn_bwd(q2, k2, v2, o2, do2, l2, dq2, dk2, dv2, B2, Sq, Sk, km2, nh=1))
    print(f"{name:28s} B {Bx:4d} Sq {Sq} Sk {Sk}: model layout fwd {f1:6.1f} bwd {b1:6.1f} us | heads contiguous fwd {f2:6.1f} bwd {b2:6.1f} us", flush=True)


layout("text self-attention", 80, 80, 64)
layout("vision queries, lang keys", 43, 86, 64)
layout("lang queries, vision keys", 86, 43, 64)
layout("episode-batched backward", 86, 43, 384)
layout("episode-batched backward", 43, 86, 384)
