#!/usr/bin/env python3
"""Generates tests/golden/graph_walk.npz by driving the REFERENCE topological map (VLN-DUET/map_nav_src/models/graph_utils.py,
imported in the build container only) with the seeded exploration `synth.GraphWalk`.

Stored per (step t, agent b), in the reference's own node order (insertion order of `node_positions`):
  names      node names                                   visited    FloydGraph.visited
  dist       GraphMap.graph.distance over all node pairs  hops       len(FloydGraph.path) from the current node to every node
  pos_fts    GraphMap.get_pos_fts(cur, [None] + nodes)    start_fts  get_pos_fts(cur, [start_vp])
The builder loops of r2r/agent.py (not importable here: MatterSim) are restated in oracle/graph_oracle.py on top of these."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/VLN-DUET/map_nav_src")


from vln_imagine_amd import synth  # noqa: E402
from tests.golden.variants import WALK  # noqa: E402



def main():
    from models.graph_utils import GraphMap  # the reference
    w = synth.GraphWalk(**WALK)
    out = {}
    maps = [GraphMap(ob["viewpoint"]) for ob in w.steps[0]]
    for t, obs in enumerate(w.steps):
        for b, (ob, m) in enumerate(zip(obs, maps)):
            m.update_graph(ob)
            names = list(m.node_positions.keys())
            cur = ob["viewpoint"]
            out[f"names_{t}_{b}"] = np.array(names)
            out[f"visited_{t}_{b}"] = np.array([m.graph.visited(k) for k in names])
            out[f"dist_{t}_{b}"] = np.array([[m.graph.distance(x, y) for y in names] for x in names], np.float64)
            out[f"hops_{t}_{b}"] = np.array([len(m.graph.path(cur, y)) for y in names], np.int64)
            out[f"pos_fts_{t}_{b}"] = m.get_pos_fts(cur, [None] + names, ob["heading"], ob["elevation"])
            out[f"start_fts_{t}_{b}"] = m.get_pos_fts(cur, [m.start_vp], ob["heading"], ob["elevation"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_walk.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", sum(len(out[f"names_{WALK['T'] - 1}_{b}"]) for b in range(WALK["B"])), "nodes at the end")


if __name__ == "__main__":
    main()
