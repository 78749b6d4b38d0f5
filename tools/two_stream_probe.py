"""Does a training step get shorter when the batch runs as NSPLIT independent sub-batches on NSPLIT streams (the tail round of one
stream's launch beside the head of another's)? Times the captured step of the bench's HAMT workload at B = 64 as one batch and as
NSPLIT sub-batches (same model, same gradient arena). Usage: NSPLIT=2 python tools/two_stream_probe.py"""
import os
import sys
import gc

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vln_imagine_amd import ops, synth  # noqa: E402
from vln_imagine_amd.train import FlatTrainer  # noqa: E402
from vln_imagine_amd.hamt.config import HamtConfig  # noqa: E402
from vln_imagine_amd.hamt.episode import EpisodeTensors, TapedEpisode  # noqa: E402
from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT  # noqa: E402
from vln_imagine_amd.hamt.spec import param_shapes  # noqa: E402

B, T = int(os.environ.get("B", "64")), 6
NS = [int(v) for v in os.environ.get("NSPLIT", "2,4").split(",")]
cfg = HamtConfig()
model = NavCMT(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()})
model = model.cuda().train().set_compute_dtype(torch.bfloat16)
tr = FlatTrainer(model)


def build(n):
    ets = [EpisodeTensors(synth.HamtEpisode(tag=f"h{n}_{i}", B=B // n, L=80, V=37, I=6, T=T, ragged=False), "cuda") for i in range(n)]
    tapes = [ops.EpisodeTape(T) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)] if n > 1 else [None]

    def on(i):
        return torch.cuda.stream(streams[i]) if streams[i] is not None else torch.cuda.stream(torch.cuda.current_stream())

    def fwd_bwd():
        cur = torch.cuda.current_stream()
        eps = [TapedEpisode(model, ets[i], tape=tapes[i], criterion=ops.cross_entropy_sum, overlap_history=(n == 1 or os.environ.get("OVERLAP") == "1")) for i in range(n)]
        for s in streams:
            if s is not None:
                s.wait_stream(cur)
        for i in range(n):
            with on(i):
                eps[i].begin()
        for t in range(T):
            for i in range(n):
                with on(i):
                    eps[i].step(t)
        outs = []
        for i in range(n):
            with on(i):
                outs.append(eps[i].finish()["loss"])
        for s in streams:
            if s is not None:
                cur.wait_stream(s)
        loss = outs[0]
        for o in outs[1:]:
            loss = loss + o
        loss = loss / n
        loss.backward()
        cap_cur = torch.cuda.is_current_stream_capturing()
        for i, s in enumerate(streams):               # backward nodes ran on the forward's streams: join them (and the tapes' history streams)
            if s is not None:
                side = getattr(tapes[i], "_side", None)
                for x in (side, s):
                    if x is None:
                        continue
                    with torch.cuda.stream(x):
                        live = torch.cuda.is_current_stream_capturing() == cap_cur
                    if live:
                        cur.wait_stream(x)
        return loss
    return fwd_bwd


CAPTURING = [False]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for n in NS:
    fb = build(n)
    for _ in range(2):
        tr.zero_grad()
        loss = fb()
        tr.step()
    torch.cuda.synchronize()
    print(f"nsplit {n}: eager ok, loss {float(loss):.5f}", flush=True)
    loss = None
    gc.collect()
    CAPTURING[0] = True
    g = tr.capture(fb, warmup=1)
    CAPTURING[0] = False
    ms = timeit(g)
    print(f"nsplit {n}: captured step {ms:.3f} ms", flush=True)
    del g
    gc.collect()
