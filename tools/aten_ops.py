"""Which torch-native (aten) ops still run in one bench step, and from where (python source line)? torch.profiler with stacks."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from vln_imagine_amd import ops, synth
from vln_imagine_amd.train import FlatTrainer
if "--duet" in sys.argv:
    from vln_imagine_amd.duet.config import DuetConfig
    from vln_imagine_amd.duet.episode import DuetEpisodeTensors as EpisodeTensors, run_episode, run_episode_taped
    from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT
    from vln_imagine_amd.duet.spec import param_shapes
    cfg = DuetConfig()
    model = GlocalTextPathNavCMT(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()})
    model = model.cuda().train().set_compute_dtype(torch.bfloat16)
    et = EpisodeTensors(synth.DuetEpisode(tag="hp", B=8, L=80, V=36, I=6, T=6, ragged=False), "cuda")
else:
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode, run_episode_taped
    from vln_imagine_amd.hamt.models.vilmodel_cmt import NavCMT
    from vln_imagine_amd.hamt.spec import param_shapes
    cfg = HamtConfig()
    model = NavCMT(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()})
    model = (model.cuda().train() if "--train" in sys.argv else model.cuda().eval()).set_compute_dtype(torch.bfloat16)
    et = EpisodeTensors(synth.HamtEpisode(tag="hp", B=8, L=80, V=37, I=6, T=6, ragged=False), "cuda")
tr = FlatTrainer(model)
TAPE = ops.EpisodeTape(6) if "--taped" in sys.argv else None      # --taped: the episode-tape drivers (what bench.py times)
def step():
    tr.zero_grad()
    if TAPE is not None:
        out = run_episode_taped(model, et, tape=TAPE, criterion=ops.cross_entropy_sum)
    else:
        out = run_episode(model, et, criterion=ops.cross_entropy_sum, keep=False)
    out["loss"].backward()
    tr.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::view", "aten::as_strided", "aten::reshape", "aten::empty_like", "aten::empty_strided",
                                                          "aten::slice", "aten::select", "aten::expand", "aten::detach", "aten::_unsafe_view", "aten::unsqueeze", "aten::squeeze", "aten::transpose", "aten::t", "aten::alias", "aten::result_type", "aten::to", "aten::contiguous", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::permute", "aten::unbind", "aten::narrow", "aten::split", "aten::chunk", "aten::flatten", "aten::view_as", "aten::size", "aten::stride", "aten::numel", "aten::dim", "aten::resize_", "aten::set_", "aten::record_stream", "aten::clone", "aten::ones_like", "aten::zeros_like", "aten::new_zeros", "aten::new_empty", "aten::expand_as", "aten::unflatten", "aten::unsafe_split", "aten::_reshape_alias"):
        st = [s for s in (ev.stack or []) if ("imagine_amd" in s or "bench.py" in s) and "aten_ops" not in s]
        src = st[0] if st else ("backward/engine" if not ev.stack else ev.stack[0])
        cnt[(ev.name, src.split("/")[-1][:90])] += 1
for (name, src), n in cnt.most_common(70):
    print(f"{n:4d} {name:28s} {src}")
