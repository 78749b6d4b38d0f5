import cProfile, io, os, pstats, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from vln_imagine_amd import dropin
fam = sys.argv[1] if len(sys.argv) > 1 else "hamt"
args = argparse.Namespace(batch=64 if fam == "hamt" else 32, T=6, L=80, V=37, I=6)
w = bench.Workload(fam, args, False, torch.device("cuda"), torch.bfloat16, batch=args.batch, tag="probe")
w.model.train()
wrap = (dropin.wrap_hamt if fam == "hamt" else dropin.wrap_duet)(w.model, feat_dropout=0.4)
tr = dropin.DropInTrainer(wrap, w.et, fam)
for _ in range(4):
    tr.step()
torch.cuda.synchronize()
lossf = dropin.hamt_agent_loss if fam == "hamt" else dropin.duet_agent_loss
pr = cProfile.Profile()
for _ in range(3):
    tr.opt.zero_grad()
    pr.enable()
    loss, _ = lossf(tr.w, tr.et)
    pr.disable()
    loss.backward(); tr.opt.step()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45)
print(s.getvalue()[:9000])
