"""One dual-problem GEMM shape, one variant, N launches: a target for rocprofv3 (kernel trace / PMC). argv: N K variant [nn] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vln_imagine_amd import ops
N, K, v = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nn = len(sys.argv) > 4 and sys.argv[4] == "nn"
n = int(sys.argv[5]) if len(sys.argv) > 5 else 20
M0, M1 = 64 * 86, 64 * 40
dt = torch.bfloat16
r = lambda *s, sc=0.5: (torch.randn(*s, device="cuda") * sc).to(dt)
a = (r(M0, K), r(M1, K))
w = (r(K, N, sc=0.05), r(K, N, sc=0.05)) if nn else (r(N, K, sc=0.05), r(N, K, sc=0.05))
b = (ops.KN(w[0]), ops.KN(w[1])) if nn else w
bias = (torch.randn(N, device="cuda"), torch.randn(N, device="cuda"))
ops.GEMM_VARIANTS, ops.NN_VARIANTS = (v,), (v,)
for _ in range(n):
    ops.gemm_nt2(a, b, bias=bias)
torch.cuda.synchronize()
