"""Episode-long weight-gradient shapes (dW[N,K] = dY[M,N]^T X[M,K], M = T x B x tokens) of the bench step: ops.wgrad called on ONE gradient (float atomics
into the output; the step itself defers and groups them - same-shape gradients in one launch, partial slabs, one batched reduction - and reaches 820-830
TFLOP/s over its 26 launches, profiles/r03_step_breakdown.md) beside the vendor library's TN GEMM on the same operands (bf16 output). Measured: 405-519
against 222-602 TFLOP/s - a 50 k-deep reduction into a 768 x 768 ... 3072 x 768 output is no shape the library is tuned for either.
usage: python tools/wgrad_probe.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vln_imagine_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 49536
dt = torch.bfloat16


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    dy, x = (torch.randn(M, N, device="cuda") * 0.5).to(dt), (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    out = torch.zeros(N, K, device="cuda")
    us = timeit(lambda: ops.wgrad(dy, x, out=out))
    us_lib = timeit(lambda: torch.matmul(dy.t(), x))
    fl = 2.0 * M * N * K
    print(f"wgrad M={M} N={N:4d} K={K:4d}: ops.wgrad {us:7.1f} us / {fl / us / 1e6:5.0f} TF | vendor bf16-out TN {us_lib:7.1f} us / {fl / us_lib / 1e6:5.0f} TF", flush=True)
