"""cProfile of the host side of one DUET bench step at a tiny batch (kernels negligible)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_imagine_amd import ops, synth
from vln_imagine_amd.duet.config import DuetConfig
from vln_imagine_amd.duet.episode import DuetEpisodeTensors, run_episode
from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT
from vln_imagine_amd.duet.spec import param_shapes
from vln_imagine_amd.train import FlatTrainer
B = int(os.environ.get("B", "2"))
cfg = DuetConfig()
model = GlocalTextPathNavCMT(cfg)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()})
model = model.cuda().eval().set_compute_dtype(torch.bfloat16)
tr = FlatTrainer(model)
et = DuetEpisodeTensors(synth.DuetEpisode(tag="hp", B=B, L=80, V=36, I=6, T=6, ragged=False), "cuda")
def step():
    tr.zero_grad()
    out = run_episode(model, et, criterion=ops.cross_entropy_sum, keep=False)
    out["loss"].backward()
    tr.step()
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 3 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30); print(s.getvalue()[:7000])
