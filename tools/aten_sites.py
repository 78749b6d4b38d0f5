"""Which torch-native ops does one taped step still launch, and from which SOURCE LINE? A TorchDispatchMode that records the innermost
frame inside the package for every non-view aten op of the forward (backward ops run on the autograd thread: listed as 'engine').
Usage: python tools/aten_sites.py [--duet]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from vln_imagine_amd import ops, synth  # noqa: E402
from vln_imagine_amd.train import FlatTrainer  # noqa: E402

DUET = "--duet" in sys.argv
if DUET:
    from vln_imagine_amd.duet.config import DuetConfig
    from vln_imagine_amd.duet.episode import DuetEpisodeTensors as EpisodeTensors, run_episode_taped
    from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT as Net
    from vln_imagine_amd.duet.spec import param_shapes
    cfg = DuetConfig()
    ep = synth.DuetEpisode(tag="hp", B=8, L=80, V=36, I=6, T=6, ragged=False)
else:
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode_taped
    from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT as Net
    from vln_imagine_amd.hamt.spec import param_shapes
    cfg = HamtConfig()
    ep = synth.HamtEpisode(tag="hp", B=8, L=80, V=37, I=6, T=6, ragged=False)
model = Net(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()})
model = model.cuda().train().set_compute_dtype(torch.bfloat16)
et = EpisodeTensors(ep, "cuda")
tr = FlatTrainer(model)
tape = ops.EpisodeTape(6)

VIEWS = {"view", "as_strided", "reshape", "slice", "select", "expand", "detach", "_unsafe_view", "unsqueeze", "squeeze", "transpose", "t", "alias",
         "permute", "unbind", "narrow", "split", "chunk", "flatten", "view_as", "unflatten", "_reshape_alias", "empty", "empty_like", "empty_strided",
         "new_empty", "new_empty_strided", "split_with_sizes", "lift_fresh", "is_nonzero", "_local_scalar_dense", "set_", "resize_", "record_stream"}


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            site = "engine"
            for fr in reversed(traceback.extract_stack()):
                if "imagine_amd" in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:100]}"
                    break
            self.cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def step():
    tr.zero_grad()
    out = run_episode_taped(model, et, tape=tape, criterion=ops.cross_entropy_sum)
    out["loss"].backward()
    tr.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with Sites() as s:
    step()
torch.cuda.synchronize()
tot = 0
for (name, site), n in sorted(s.cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} {name:22s} {site}")
    tot += n
print("total", tot)
