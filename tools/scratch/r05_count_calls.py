import sys, os, collections, argparse
sys.path.insert(0, os.getcwd())
import torch, bench
from vln_imagine_amd import dropin, _lib, ops
args = argparse.Namespace(batch=64, T=6, L=80, V=37, I=6)
w = bench.Workload("hamt", args, False, torch.device("cuda"), torch.bfloat16, batch=64, tag="probe"); w.model.train()
tr = dropin.DropInTrainer(dropin.wrap_hamt(w.model, 0.4), w.et, "hamt")
for _ in range(3): tr.step()
cnt = collections.Counter(); orig = _lib.call
def c(name, *a):
    cnt[name] += 1; return orig(name, *a)
_lib.call = c
tr.step(); torch.cuda.synchronize()
_lib.call = orig
for k, v in cnt.most_common(): print(v, k)
print("total", sum(cnt.values()), "BLOCK_CALLS", ops.BLOCK_CALLS)
