"""vlni_adamw_step_groups on a 107 M-element arena (the HAMT model's size), 3 groups: microseconds per launch and effective TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vln_imagine_amd import _lib, ops
n = 107_000_000 // 8 * 8
dev = "cuda"
p, g, m, v = (torch.randn(n, device=dev) * 0.02 for _ in range(4))
v.abs_()
sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
G = 3
grp_end = torch.tensor([n // 10 // 8 * 8, n // 5 // 8 * 8, n], dtype=torch.int64, device=dev)
grp_lr = torch.tensor([1e-5, 1e-5, 1e-6, 1.0, 1.0, 1.0], device=dev)          # lr per group, then trainable flags
gstate = torch.tensor([3.0, 1 - 0.9 ** 3, 1 - 0.999 ** 3, 0.0] * G, device=dev)
state = torch.zeros(8, device=dev)
state[0] = 1.0
st = torch.cuda.current_stream().cuda_stream
call = lambda: _lib.call("vlni_adamw_step_groups", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), ops.BF16, n,
                         grp_end.data_ptr(), grp_lr.data_ptr(), gstate.data_ptr(), G, 0.9, 0.999, 1e-8, 0.01, state.data_ptr(), st)
for _ in range(3): call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print(f"adamw {n / 1e6:.0f} M elements: {us:.0f} us, {n * 30 / us / 1e6:.2f} TB/s (30 B per element)")
