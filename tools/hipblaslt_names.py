"""Which hipBLASLt macro-tiles win on the hot-path shapes? (run under rocprofv3 --kernel-trace; names encode MT/MI/etc.)"""
import torch
dt = torch.bfloat16
for (M, N, K) in [(8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 768, 768), (5120, 3072, 768), (5120, 768, 3072), (2304, 768, 3072), (30720, 3072, 768), (4096, 4096, 4096)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    for _ in range(5): torch.matmul(a, b.t())
    torch.cuda.synchronize()
