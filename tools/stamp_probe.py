import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_imagine_amd import ops, _lib
lib = _lib.load()
lib.vlni_debug_set_stamps.argtypes = [ctypes.c_void_p]
dt = torch.bfloat16
for (M, N, K, v) in [(8192, 3072, 768, 5), (8192, 768, 768, 5), (8192, 768, 3072, 5)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt); b = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    st = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    lib.vlni_debug_set_stamps(st.data_ptr())
    for _ in range(3):
        ops._gemm_call(v, a, b, out, None, 0, None, None, None, 0, 1.0, 1, False, M, N, K)
    torch.cuda.synchronize()
    s = st.view(-1, 8).cpu().double()
    s = s[s[:, 5] > 0]
    m = s.mean(0)
    nk = m[5].item()
    print(f"M={M} N={N} K={K} v{v}: blocks {len(s)} per k-tile: store drain after epilogue {m[2]:.0f} | main {m[3]:.0f} epilogue {m[4]:.0f} cycles (100 MHz counter?)")
lib.vlni_debug_set_stamps(None)
