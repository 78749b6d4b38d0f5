"""Repro harness for the round-3 anomaly ("not understood"): with the history mask computed INSIDE a captured step
(`arange < lengths[t]`) some capture sequences replayed step 0's mask at steps >= 1. Runs the forward-only stepped graphs several
times per process with the in-graph mask and reports, per (capture, episode, step), whether logits / mask / history buffer agree with
the eager run. Usage: python tools/stale_mask_repro.py [debug_refs(0|1)] [want_states(0|1)]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden.variants import HAMT_C1                      # noqa: E402
from tests.test_hamt_gpu import build_product                  # noqa: E402
from vln_imagine_amd import ops, synth                         # noqa: E402
from vln_imagine_amd.hamt.buckets import EpisodeBuffers, SteppedInferenceGraphs   # noqa: E402
from vln_imagine_amd.hamt.config import HamtConfig             # noqa: E402
from vln_imagine_amd.hamt.episode import run_episode           # noqa: E402

B, I, L, V = 8, 4, 64, 31
dbg_refs = len(sys.argv) > 1 and sys.argv[1] == "1"
want_states = len(sys.argv) > 2 and sys.argv[2] == "1"
model = build_product(HamtConfig(**HAMT_C1))
bad = 0
for cap, T in enumerate((2, 3, 3, 4, 3)):
    eps = [synth.HamtEpisode(tag=f"inf{T}_{i}", B=B, L=L - 5 * i, V=V - 2 * i, I=I, T=T, ragged=True) for i in range(3)]
    bufs = EpisodeBuffers(B, L, V, I, T, "cuda").load(eps[0])
    dbg = {} if dbg_refs else None
    g = SteppedInferenceGraphs(model, bufs, want_states=want_states, mask_in_graph=True, debug=dbg)
    for ei, ep in enumerate(eps):
        with torch.no_grad():
            ref = run_episode(model, EpisodeBuffers(B, L, V, I, T, "cuda").load(ep), use_aux=False, criterion=ops.cross_entropy_sum)
        bufs.load(ep, steps=False)
        g.begin()
        for t in range(T):
            bufs.put_hist_lens(t, ep.hist_lens[t])
            bufs.put_step(t, ep.steps[t], keys=EpisodeBuffers.OBS_KEYS)
            if t > 0:
                bufs.put_step(t - 1, ep.steps[t - 1], keys=EpisodeBuffers.HIST_KEYS)
            g.step(t)
            torch.cuda.synchronize()
            a, b = g.logits(t), ref["logits"][t]
            fin = torch.isfinite(b)
            ok = bool(torch.equal(torch.isfinite(a), fin) and torch.allclose(a[fin], b[fin], atol=2e-5))
            msg = f"capture {cap} T={T} episode {ei} step {t}: logits {'ok' if ok else 'MISMATCH'}"
            if dbg is not None:
                want = torch.arange(T, device="cuda")[None, :] < bufs.hist_lens_dev[t][:, None]
                msg += f" mask {'ok' if torch.equal(dbg[('hm', t)], want) else 'STALE ' + str(dbg[('hm', t)].int().tolist()) + ' want ' + str(want.int().tolist())}"
            if not ok:
                bad += 1
                err = float((a[fin] - b[fin]).abs().max())
                msg += f" max|d|={err:.3e} lens={bufs.hist_lens_dev[t].tolist()}"
                # which stale mask explains it? logits of the eager model with step s's lengths at step t
                g.step(t)
                torch.cuda.synchronize()
                again = g.logits(t)
                msg += f" replay-again {'same' if torch.equal(again[fin], a[fin]) else 'differs'}"
            print(msg, flush=True)
    del g
print("mismatches:", bad)
