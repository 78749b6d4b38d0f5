"""cProfile of the host side of one bench step at a tiny batch (kernels negligible): where do the ~50 ms of Python go?"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vln_imagine_amd import ops, synth
from vln_imagine_amd.hamt.config import HamtConfig
from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode
from vln_imagine_amd.train import FlatTrainer
import bench
cfg = HamtConfig()
model = bench.make_model(cfg, torch.bfloat16, torch.device("cuda"))
tr = FlatTrainer(model)
et = EpisodeTensors(synth.HamtEpisode(tag="hp", B=2, L=80, V=37, I=6, T=6, ragged=False), "cuda")
def step():
    tr.zero_grad()
    out = run_episode(model, et, criterion=ops.cross_entropy_sum, keep=False)
    out["loss"].backward()
    tr.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
