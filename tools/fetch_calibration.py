"""Calibrates rocprofv3's FETCH_SIZE on THIS kernel's access pattern (MI355X_MICROARCH.md: 'calibrate on a known byte count in your
own access pattern'): a GEMM with ONE column tile (N = 128) streams A exactly once - 65536 x 768 bf16 = 100.7 MB - and re-reads only
a 196-KB B panel, so FETCH_SIZE x correction must come out at ~101 MB. Run under `rocprofv3 --pmc FETCH_SIZE --output-format csv`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vln_imagine_amd import ops

M, N, K = 65536, 128, 768
a = torch.randn(M, K, device="cuda").bfloat16()
b = torch.randn(N, K, device="cuda").bfloat16()
ops.AUTOTUNE = False
for v in (5, 1):                      # LDS-DMA pipeline (global_load_lds, 16 B per lane) and the register-staged one (global_load_dwordx4)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops._gemm_call(v, a, b, out, None, 0, None, None, None, 0, 1.0, 1, False, M, N, K)
torch.cuda.synchronize()
print("A bytes", M * K * 2, "B bytes", N * K * 2, "C bytes", M * N * 2)
