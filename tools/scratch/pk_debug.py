import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vln_imagine_amd import ops
torch.manual_seed(0)
for (M, N, K) in ((128, 128, 128), (128, 128, 192), (128, 128, 256), (128, 128, 384), (128, 128, 768), (128, 128, 768)):
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    o1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); o2 = torch.empty_like(o1)
    ops._gemm_call(1, a, b, o1, None, 0, None, None, None, 0, 1.0, 1, False, M, N, K)
    o2.fill_(7.0)
    ops._gemm_call(14, a, b, o2, None, 0, None, None, None, 0, 1.0, 1, False, M, N, K)
    torch.cuda.synchronize()
    d = (o1.float() - o2.float()).abs()
    bad = d > 0
    print(M, N, K, "max", d.max().item(), "frac bad", bad.float().mean().item())
    if bad.any():
        # partial sums: which k-tiles contribute? compare with reference restricted to k ranges
        af, bf = a.float(), b.float()
        full = af @ bf.t()
        for nkt in range(1, K // 64 + 1):
            part = af[:, :64 * nkt] @ bf[:, :64 * nkt].t()
            print("   first %d k-tiles: max|o2 - partial| = %.4f" % (nkt, (o2.float() - part).abs().max().item()))
        # hypothesis: k-tile t uses B row block start for t>=1
        print("   o2[0,:8]", o2[0, :8].tolist(), "\n   o1[0,:8]", o1[0, :8].tolist())
        print("   o2[5,30:36]", o2[5, 30:36].tolist(), "\n   o1[5,30:36]", o1[5, 30:36].tolist())
