"""CPU: oracle/graph_oracle.py against the reference's own topological-map outputs (tests/golden/graph_walk.npz, written by
tests/golden/make_golden_graph.py from VLN-DUET/map_nav_src/models/graph_utils.py) and builder-level properties."""
import os

import numpy as np

from oracle import graph_oracle as GO
from tests.golden.variants import WALK
from vln_imagine_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "graph_walk.npz")


def _replay():
    w = synth.GraphWalk(**WALK)
    maps = [GO.TopoMap(ob["viewpoint"]) for ob in w.steps[0]]
    for t, obs in enumerate(w.steps):
        for ob, m in zip(obs, maps):
            m.observe(ob)
        yield t, obs, maps


def test_topomap_matches_reference_graph_utils():
    g = np.load(GOLD)
    multi_hop = 0
    for t, obs, maps in _replay():
        for b, (ob, m) in enumerate(zip(obs, maps)):
            names = [str(x) for x in g[f"names_{t}_{b}"]]
            assert m.names == names
            n, cur = len(names), m.slot[ob["viewpoint"]]
            assert np.array_equal(m.seen[:n], g[f"visited_{t}_{b}"])
            d = np.array([[m.distance(x, y) for y in range(n)] for x in range(n)])
            assert np.array_equal(d, g[f"dist_{t}_{b}"])                              # float64, bit-exact
            hops = np.array([m.hops(cur, y) for y in range(n)])
            assert np.array_equal(hops, g[f"hops_{t}_{b}"])
            multi_hop += int((hops > 1).sum())
            f = m.pos_fts(ob["viewpoint"], [None] + names, ob["heading"], ob["elevation"])
            assert f.dtype == np.float32 and np.abs(f - g[f"pos_fts_{t}_{b}"]).max() <= 1e-6
            assert np.array_equal(f[:, 4:], g[f"pos_fts_{t}_{b}"][:, 4:])             # distances / hop counts: exact
            assert np.abs(m.pos_fts(ob["viewpoint"], [m.start_vp], ob["heading"], ob["elevation"]) - g[f"start_fts_{t}_{b}"]).max() <= 1e-6
    assert multi_hop > 50                                                              # the walk exercises relaxed pairs


def test_nav_variables_layout():
    for t, obs, maps in _replay():
        for ob, m in zip(obs, maps):
            m.step_id[ob["viewpoint"]] = t + 1
        out = GO.nav_gmap_variable(obs, maps)
        B, G = out["gmap_masks"].shape
        for b in range(B):
            ids = out["gmap_vpids"][b]
            n = len(ids)
            assert ids[0] is None and out["gmap_masks"][b].sum() == n
            assert np.array_equal(out["gmap_pos_fts"][b, 0], np.array([0, 1, 0, 1, 0, 0, 0], np.float32))
            assert not out["gmap_pos_fts"][b, n:].any() and not out["gmap_pair_dists"][b, 0].any()
            vis = out["gmap_visited_masks"][b, :n]
            assert not vis[0] and vis[1:1 + vis.sum()].all()                           # [stop], visited..., frontier...
            assert (out["gmap_step_ids"][b, :n][vis] > 0).all()
            assert np.array_equal(out["gmap_pair_dists"][b], out["gmap_pair_dists"][b].T)
        cand = [[c["viewpointId"] for c in ob["candidate"]] for ob in obs]
        vp = GO.nav_vp_variable(obs, maps, cand, [36] * B, np.zeros((B, 36), np.int64), 36)
        assert vp["vp_pos_fts"].shape == (B, 37, 14)
        for b in range(B):
            assert np.array_equal(vp["vp_pos_fts"][b, 5, :7], vp["vp_pos_fts"][b, 0, :7]) and not vp["vp_pos_fts"][b, 0, 7:].any()
            assert not vp["vp_pos_fts"][b, len(cand[b]) + 1:, 7:].any()


def test_imaginations_v2_slots():
    flags = {"a": ["True", "False", "True"], "b": ["False", "False"], "c": ["False", "True", "False", "True"]}
    feats = {"a": np.arange(2 * 770, dtype=np.float64).reshape(2, 770), "c": -np.arange(2 * 770, dtype=np.float64).reshape(2, 770)}
    f, m = GO.imaginations_v2(["a", "b", "c"], flags, feats)
    assert f.shape == (3, 4, 768) and f.dtype == np.float32
    assert m.tolist() == [[True, False, True, False], [False] * 4, [False, True, False, True]]
    assert np.array_equal(f[0, 2], feats["a"][1, :768].astype(np.float32)) and not f[1].any() and not f[0, 1].any()
    assert np.array_equal(f[2, 3], feats["c"][1, :768].astype(np.float32))


def test_rollout_oracle_matches_reference_golden(golden_dir):
    """Whole chain on CPU: oracle model + oracle builders through duet/rollout.py == reference model + reference GraphMap."""
    import torch
    from oracle.duet_oracle import DuetOracle
    from tests.golden.variants import DUET_C1, rollout_setup
    from tests.test_oracle_hamt import _close
    from vln_imagine_amd.duet.config import DuetConfig
    from vln_imagine_amd.duet.rollout import rollout
    from vln_imagine_amd.duet.spec import param_shapes
    g = np.load(os.path.join(golden_dir, "duet_rollout.npz"))
    walk, feats, keys, ep = rollout_setup()
    cfg = DuetConfig(**DUET_C1)
    sd = {k: torch.from_numpy(v).requires_grad_() for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()}
    torch.set_num_threads(8)
    t = torch.from_numpy
    out = rollout(DuetOracle(cfg, sd), walk, GO.OracleNavBuilders(feats, keys), t(ep.txt_ids), t(ep.txt_masks), t(ep.imagine_feats),
                  t(ep.imagine_masks))
    assert len(out["fused"]) == int(g["steps"])
    _close(out["loss"].item(), g["loss"], what="loss")
    for i, f in enumerate(out["fused"]):
        assert np.array_equal(out["targets"][i], g[f"target{i}"])
        assert ["|".join("" if k is None else k for k in row) for row in out["gmap_vpids"][i]] == g[f"vpids{i}"].tolist()
        _close(f.detach(), g[f"fused{i}"], what=f"fused{i}")
    assert (g["target1"] >= 0).all() and (g[f"target{int(g['steps']) - 1}"] == -100).any()      # an agent that stopped early is ignored later


def test_hamt_rollout_oracle_matches_reference_golden(golden_dir):
    """Whole HAMT chain on CPU: oracle model + oracle builders through hamt/rollout.py == reference NavCMT (tests/golden/hamt_rollout.npz)."""
    import torch
    from oracle.hamt_oracle import HamtOracle
    from tests.golden.variants import HAMT_C1, hamt_rollout_setup
    from tests.test_oracle_hamt import _close
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.rollout import rollout
    from vln_imagine_amd.hamt.spec import param_shapes
    g = np.load(os.path.join(golden_dir, "hamt_rollout.npz"))
    walk, feats, keys, ep, imag, flags = hamt_rollout_setup()
    cfg = HamtConfig(**HAMT_C1)
    sd = {k: torch.from_numpy(v).requires_grad_() for k, v in synth.fill_state_dict(param_shapes(cfg).items()).items()}
    torch.set_num_threads(8)
    t = torch.from_numpy
    out = rollout(HamtOracle(cfg, sd), walk, GO.OracleObsBuilders(feats, keys, imag, flags), t(ep.txt_ids), t(ep.txt_masks),
                  annotations=(ep.sub_instr_segs, ep.sub_instr_imag_flag, ep.noun_phrase_segs))
    assert len(out["logits"]) == int(g["steps"]) and np.array_equal(out["hist_lens"], g["hist_lens"])
    _close(out["loss"].item(), g["loss"], what="loss")
    _close(out["aux"].item(), g["aux"], what="aux")
    for i, f in enumerate(out["logits"]):
        assert np.array_equal(out["targets"][i], g[f"target{i}"])
        _close(f.detach(), g[f"logits{i}"], what=f"logits{i}")
    last = g[f"target{int(g['steps']) - 1}"]
    assert (last == -100).any() and (g["target1"] == 0).any()                     # ended agents ignored; first candidates taken


def test_host_graphmap_module_matches_reference_golden():
    """`models.graph_utils.GraphMap` as shipped in the DUET package (host-side drop-in): same numbers as the reference's."""
    from vln_imagine_amd.duet.models.graph_utils import GraphMap
    g = np.load(GOLD)
    w = synth.GraphWalk(**WALK)
    maps = [GraphMap(ob["viewpoint"]) for ob in w.steps[0]]
    for t, obs in enumerate(w.steps):
        for b, (ob, m) in enumerate(zip(obs, maps)):
            m.update_graph(ob)
            names = list(m.node_positions.keys())
            assert names == [str(x) for x in g[f"names_{t}_{b}"]]
            assert np.array_equal(np.array([[m.graph.distance(x, y) for y in names] for x in names]), g[f"dist_{t}_{b}"])
            assert [len(m.graph.path(ob["viewpoint"], y)) for y in names] == g[f"hops_{t}_{b}"].tolist()
            assert [m.graph.visited(k) for k in names] == g[f"visited_{t}_{b}"].tolist()
            f = m.get_pos_fts(ob["viewpoint"], [None] + names, ob["heading"], ob["elevation"])
            assert np.abs(f - g[f"pos_fts_{t}_{b}"]).max() <= 1e-6 and np.array_equal(f[:, 4:], g[f"pos_fts_{t}_{b}"][:, 4:])
