"""Per-step cost of DUET's navigation-input builders: the reference-style host loops (oracle/graph_oracle.py, CPU) next to the
device-resident map (vln_imagine_amd/graphmap.py), same exploration (B = 64 agents, 15 steps, 200 viewpoints)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import graph_oracle as GO
from vln_imagine_amd import synth
from vln_imagine_amd.graphmap import DeviceGraphMap

B, T = 64, 15
walk = synth.GraphWalk(tag="walk_big", B=B, T=T, n=200, k=5)
cands = [[[c["viewpointId"] for c in ob["candidate"]] for ob in obs] for obs in walk.steps]
V = 36
nav_types = np.zeros((B, V), np.int64)
lens = [V] * B


def host():
    maps = [GO.TopoMap(ob["viewpoint"]) for ob in walk.steps[0]]
    per = []
    for t, obs in enumerate(walk.steps):
        t0 = time.perf_counter()
        for ob, m in zip(obs, maps):
            m.observe(ob)
            m.step_id[ob["viewpoint"]] = t + 1
        g = GO.nav_gmap_variable(obs, maps)
        v = GO.nav_vp_variable(obs, maps, cands[t], lens, nav_types, V)
        dev = [torch.from_numpy(a).cuda() for a in (g["gmap_pos_fts"], g["gmap_pair_dists"], g["gmap_step_ids"], g["gmap_visited_masks"],
                                                    g["gmap_masks"], v["vp_pos_fts"])]            # the agent's .cuda() copies
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t0)
    return per


def device():
    pano = torch.zeros((B, V, 768), device="cuda")
    nt = torch.from_numpy(nav_types).cuda()
    dm = None
    per = []
    for t, obs in enumerate(walk.steps):
        t0 = time.perf_counter()
        if dm is None:
            dm = DeviceGraphMap(obs, cap=128)
        else:
            dm.observe(obs)
        dm.mark_step(obs, t)
        g = dm.nav_gmap_variable(obs)
        v = dm.nav_vp_variable(obs, pano, cands[t], lens, nt)
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t0)
    return per, max(len(n) for n in dm.names)


device()
for name, fn in (("host loops (reference style, 1 core)", host), ("device map", lambda: device()[0])):
    runs = [fn() for _ in range(3)]
    best = np.min(np.array(runs), 0)
    print(f"{name:40s} per step: first {best[0]*1e3:7.2f} ms  last {best[-1]*1e3:7.2f} ms  mean {best.mean()*1e3:7.2f} ms", flush=True)
print("largest map:", device()[1], "nodes; B =", B, "T =", T)
