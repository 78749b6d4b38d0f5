"""smallk_bwd time vs rows for the current VLNI_SMALLK_RPB (one process per setting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vln_imagine_amd import _lib, ops
out = []
for rows, K in ((736, 7), (1184, 14), (2368, 4), (2304, 4), (5504, 4), (8064, 7)):
    dy = (torch.randn(rows, 768, device="cuda")).to(torch.bfloat16)
    x = torch.randn(rows, K, device="cuda")
    dW, db = torch.zeros(768, K, device="cuda"), torch.zeros(768, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: _lib.call("vlni_smallk_linear_bwd", ops.BF16, dy.data_ptr(), 768, x.data_ptr(), K, dW.data_ptr(), db.data_ptr(), rows, 768, K, st)
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{rows}x{K}: {e0.elapsed_time(e1) / 20 * 1e3:.1f}us")
print("RPB", os.environ.get("VLNI_SMALLK_RPB", "default"), " ".join(out))
