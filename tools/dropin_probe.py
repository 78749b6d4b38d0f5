"""The drop-in eager iteration (vln_imagine_amd/dropin.py: what an unchanged reference agent runs) timed at the bench's shapes, and a
cProfile of its host side at a tiny batch (kernels negligible): where do the milliseconds of Python / ctypes / autograd go?
usage: python tools/dropin_probe.py [hamt|duet] [--profile]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vln_imagine_amd import dropin  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("family", nargs="?", default="hamt")
ap.add_argument("--profile", action="store_true")
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--eval", action="store_true")
a = ap.parse_args()
args = argparse.Namespace(batch=a.batch or (64 if a.family == "hamt" else 32), T=6, L=80, V=37, I=6)
dev = torch.device("cuda")


def make(B):
    w = bench.Workload(a.family, args, False, dev, torch.bfloat16, batch=B, tag="probe")
    if not a.eval:
        w.model.train()
    wrap = (dropin.wrap_hamt if a.family == "hamt" else dropin.wrap_duet)(w.model, feat_dropout=0.0 if a.eval else 0.4)
    return dropin.DropInTrainer(wrap, w.et, a.family)


tr = make(args.batch)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    tr.step()
torch.cuda.synchronize()
print(f"{a.family} drop-in eager, batch {args.batch}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms per iteration", flush=True)
# host only: phases of one iteration without waiting for the GPU
torch.cuda.synchronize()
t0 = time.perf_counter(); tr.opt.zero_grad(); loss, _ = (dropin.hamt_agent_loss if a.family == "hamt" else dropin.duet_agent_loss)(tr.w, tr.et)
t1 = time.perf_counter(); loss.backward()
t2 = time.perf_counter(); torch.nn.utils.clip_grad_norm_(tr.params, 40.0); tr.opt.step()
t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
print(f"  host: forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f}, clip + AdamW {1e3 * (t3 - t2):.1f}, GPU drain {1e3 * (t4 - t3):.1f}", flush=True)
if a.profile:
    del tr
    tr = make(2)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    torch.autograd.set_multithreading_enabled(False)     # backward on this thread: its Python (Function.backward bodies) shows in the profile
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(70)
    print(s.getvalue()[:16000])
