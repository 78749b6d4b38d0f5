"""GPU: the device-resident topological map and the navigation / imagination builders (csrc/graphmap.hip through the C-ABI)
against oracle/graph_oracle.py and against the reference's own outputs (tests/golden/graph_walk.npz)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.variants import WALK

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "graph_walk.npz")


def _cand_names(obs):
    return [[c["viewpointId"] for c in ob["candidate"]] for ob in obs]


def _drive(walk, ended_at=None, **kw):
    """Both maps through the same exploration; yields after each step's observation."""
    from oracle import graph_oracle as GO
    from vln_imagine_amd.graphmap import DeviceGraphMap
    dm = DeviceGraphMap(walk.steps[0], **kw)
    om = [GO.TopoMap(ob["viewpoint"]) for ob in walk.steps[0]]
    for t, obs in enumerate(walk.steps):
        ended = None if ended_at is None else [t >= e for e in ended_at]
        if ended is not None:                       # an ended agent stays where it stopped (the simulator is not moved any more)
            obs = [walk.steps[min(t, e - 1)][b] for b, e in enumerate(ended_at)]
        if t > 0:
            dm.observe(obs, ended)
        for b, (ob, m) in enumerate(zip(obs, om)):
            if ended is None or not ended[b] or t == 0:
                m.observe(ob)
                m.step_id[ob["viewpoint"]] = t + 1
        dm.mark_step(obs, t, None if t == 0 else ended)
        yield t, obs, dm, om


def _assert_state_equal(dm, om):
    dis, via, seen = dm.dis.cpu().numpy(), dm.via.cpu().numpy(), dm.seen.cpu().numpy()
    for b, m in enumerate(om):
        n = len(m.names)
        assert dm.names[b] == m.names
        off = ~np.eye(n, dtype=bool)
        assert np.array_equal(dis[b, :n, :n][off], m.dis[:n, :n][off])            # float64, bit-exact
        assert np.array_equal(via[b, :n, :n][off], m.via[:n, :n][off])
        assert np.array_equal(seen[b, :n].astype(bool), m.seen[:n])


def test_map_state_and_nav_variables_match_oracle():
    from oracle import graph_oracle as GO
    from vln_imagine_amd import synth
    walk = synth.GraphWalk(**WALK)
    for t, obs, dm, om in _drive(walk):
        _assert_state_equal(dm, om)
        got, ref = dm.nav_gmap_variable(obs), GO.nav_gmap_variable(obs, om)
        assert got["gmap_vpids"] == ref["gmap_vpids"] and got["no_vp_left"] == ref["no_vp_left"]
        for k in ("gmap_step_ids", "gmap_visited_masks", "gmap_masks", "gmap_pair_dists"):
            assert np.array_equal(got[k].cpu().numpy(), ref[k]), k
        p = got["gmap_pos_fts"].cpu().numpy()
        assert np.abs(p - ref["gmap_pos_fts"]).max() <= 1e-6                       # sin / cos of float32 angles
        assert np.array_equal(p[..., 4:], ref["gmap_pos_fts"][..., 4:])            # distances and hop counts: exact
        cand = _cand_names(obs)
        V = 36
        pano = torch.zeros((dm.B, V, 768), device="cuda")
        nav_types = torch.zeros((dm.B, V), dtype=torch.long, device="cuda")
        for b, c in enumerate(cand):
            nav_types[b, :len(c)] = 1
        lens = [30 + b for b in range(dm.B)]
        got = dm.nav_vp_variable(obs, pano, cand, lens, nav_types)
        ref = GO.nav_vp_variable(obs, om, cand, lens, nav_types.cpu().numpy(), V)
        assert np.abs(got["vp_pos_fts"].cpu().numpy() - ref["vp_pos_fts"]).max() <= 1e-6
        assert np.array_equal(got["vp_masks"].cpu().numpy(), ref["vp_masks"])
        assert np.array_equal(got["vp_nav_masks"].cpu().numpy(), ref["vp_nav_masks"])
        assert got["vp_cand_vpids"] == ref["vp_cand_vpids"] and got["vp_img_embeds"].shape == (dm.B, V + 1, 768)
        for b, ob in enumerate(obs):                                               # simulator-side path expansion
            for name in om[b].names[::3]:
                assert dm.path(b, ob["viewpoint"], name) == om[b].path(om[b].slot[ob["viewpoint"]], om[b].slot[name])
        dm.check()


def test_map_matches_reference_golden():
    """Straight against what graph_utils.py produced in the build container (no oracle in between)."""
    from vln_imagine_amd import synth
    g = np.load(GOLD)
    walk = synth.GraphWalk(**WALK)
    for t, obs, dm, _ in _drive(walk):
        dis, seen = dm.dis.cpu().numpy(), dm.seen.cpu().numpy()
        out = dm.nav_gmap_variable(obs)
        for b, ob in enumerate(obs):
            names = [str(x) for x in g[f"names_{t}_{b}"]]
            n = len(names)
            assert dm.names[b] == names
            off = ~np.eye(n, dtype=bool)
            assert np.array_equal(dis[b, :n, :n][off], g[f"dist_{t}_{b}"][off])
            assert np.array_equal(seen[b, :n].astype(bool), g[f"visited_{t}_{b}"])
            # the builder lists [stop], visited, frontier; the golden lists [stop] + insertion order -> compare by name
            row = {k: i for i, k in enumerate(out["gmap_vpids"][b])}
            got = out["gmap_pos_fts"][b].cpu().numpy()[[row[k] for k in [None] + names]]
            assert np.abs(got - g[f"pos_fts_{t}_{b}"]).max() <= 1e-6
            assert np.array_equal(got[:, 4:], g[f"pos_fts_{t}_{b}"][:, 4:])
            assert np.array_equal(np.round(got[1:, 6] * 10).astype(np.int64), g[f"hops_{t}_{b}"])


def test_ended_episodes_are_frozen_and_big_batch():
    """B = 64 agents, 15 steps (the R2R action cap), 200 viewpoints; some episodes end early and must not change any more."""
    from oracle import graph_oracle as GO
    from vln_imagine_amd import synth
    walk = synth.GraphWalk(tag="walk_big", B=64, T=15, n=200, k=5)
    ended_at = [4 + (b * 7) % 12 for b in range(64)]
    last = None
    for t, obs, dm, om in _drive(walk, ended_at=ended_at, cap=128):
        last = (obs, dm, om)
    obs, dm, om = last
    _assert_state_equal(dm, om)
    got, ref = dm.nav_gmap_variable(obs), GO.nav_gmap_variable(obs, om)
    assert np.array_equal(got["gmap_pair_dists"].cpu().numpy(), ref["gmap_pair_dists"])
    p = got["gmap_pos_fts"].cpu().numpy()
    assert np.abs(p - ref["gmap_pos_fts"]).max() <= 1e-6 and np.array_equal(p[..., 4:], ref["gmap_pos_fts"][..., 4:])
    pd = got["gmap_pair_dists"]
    assert torch.equal(pd, pd.transpose(1, 2)) and max(len(n) for n in dm.names) > 30
    dm.check()


def test_capacity_and_argument_errors():
    from vln_imagine_amd import _lib, synth
    from vln_imagine_amd.graphmap import DeviceGraphMap
    walk = synth.GraphWalk(**WALK)
    dm = DeviceGraphMap(walk.steps[0], cap=8)
    with pytest.raises(ValueError):
        for obs in walk.steps[1:]:
            dm.observe(obs)
    with pytest.raises(_lib.VlniError):
        _lib.call("vlni_graph_init", dm.dis.data_ptr(), dm.via.data_ptr(), dm.seen.data_ptr(), 4, 300, None)


def test_node_images_and_gradients():
    """update_node_embed / get_node_embed bookkeeping (graph_utils.py:115-128, agent.py:461-479) with autograd: same values and the
    same gradients on every step's panorama encoding as a list-based restatement."""
    from vln_imagine_amd import synth
    walk = synth.GraphWalk(**WALK)
    B, V, H = WALK["B"], 36, 768
    g = torch.Generator().manual_seed(3)
    panos = [torch.randn((B, V, H), generator=g).cuda().requires_grad_() for _ in walk.steps]
    masks = [(torch.arange(V)[None, :] < torch.tensor([30 + b for b in range(B)])[:, None]).cuda() for _ in walk.steps]
    twins = [p.detach().clone().requires_grad_() for p in panos]
    book = [dict() for _ in range(B)]                                              # name -> [sum, count]
    seen = [set() for _ in range(B)]
    for t, obs, dm, _ in _drive(walk):
        cand = _cand_names(obs)
        dm.update_node_embeds(obs, panos[t], masks[t], cand)
        m = masks[t].float()
        avg = (twins[t] * m.unsqueeze(2)).sum(1) / m.sum(1, keepdim=True)
        for b, ob in enumerate(obs):
            seen[b].add(ob["viewpoint"])
            book[b][ob["viewpoint"]] = [avg[b], 1]
            for j, name in enumerate(cand[b]):
                if name not in seen[b]:
                    if name in book[b]:
                        book[b][name] = [book[b][name][0] + twins[t][b, j], book[b][name][1] + 1]
                    else:
                        book[b][name] = [twins[t][b, j], 1]
        out = dm.nav_gmap_variable(obs)
    emb = out["gmap_img_embeds"]
    ref = torch.zeros_like(emb)
    rows = []
    for b, ids in enumerate(out["gmap_vpids"]):
        rows.append(torch.stack([torch.zeros(H, device="cuda")] + [book[b][k][0] / book[b][k][1] for k in ids[1:]]))
        ref[b, :len(ids)] = rows[-1].detach()
    assert torch.allclose(emb, ref, atol=1e-6)
    w = torch.randn(emb.shape, generator=g).cuda()
    (emb * w).sum().backward()
    sum((r * w[b, :r.shape[0]]).sum() for b, r in enumerate(rows)).backward()
    for p, q in zip(panos, twins):
        assert torch.allclose(p.grad, q.grad, atol=1e-6)
    assert sum(int((p.grad != 0).any()) for p in panos) == len(panos)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_imagination_table(dtype):
    from oracle import graph_oracle as GO
    from vln_imagine_amd.builders import ImaginationTable
    rng = np.random.default_rng(5)
    flags, feats = {}, {}
    for i in range(40):
        fl = ["True" if rng.random() < 0.6 else "False" for _ in range(int(rng.integers(1, 9)))]
        if i % 9 == 0:
            fl = ["False"] * len(fl)
        flags[f"{i}_0"] = fl
        if "True" in fl:
            feats[f"{i}_0"] = rng.standard_normal((fl.count("True"), 772)).astype(np.float32)
    if dtype == torch.bfloat16:                                                    # make the stored values bf16-exact
        feats = {k: torch.from_numpy(v).bfloat16().float().numpy() for k, v in feats.items()}
    table = ImaginationTable(feats, flags, dtype=dtype)
    for ids in (list(flags)[:16], list(flags)[16:], ["0_0", "9_0"] if "True" in flags["9_0"] + flags["0_0"] else ["1_0"]):
        if all(all(f == "False" for f in flags[i]) for i in ids):
            continue
        f, m = table.batch(ids)
        rf, rm = GO.imaginations_v2(ids, flags, feats)
        assert np.array_equal(f.cpu().numpy(), rf) and np.array_equal(m.cpu().numpy(), rm)


def test_rollout_with_device_builders_matches_reference_golden(golden_dir):
    """End to end on the GPU: HIP model (fp32) + resident view features + device topological maps through duet/rollout.py against
    the reference model driven by the reference GraphMap (tests/golden/duet_rollout.npz): logits, loss, gradient norms at 1e-4."""
    from tests.golden.variants import DUET_C1, rollout_setup
    from tests.test_duet_gpu import build_product
    from tests.test_hamt_gpu import _close
    from vln_imagine_amd import ops
    from vln_imagine_amd.builders import ResidentFeatures
    from vln_imagine_amd.duet.config import DuetConfig
    from vln_imagine_amd.duet.rollout import DeviceNavBuilders, rollout
    g = np.load(os.path.join(golden_dir, "duet_rollout.npz"))
    walk, feats, keys, ep = rollout_setup()
    model = build_product(DuetConfig(**DUET_C1))
    t = lambda a: torch.from_numpy(a).cuda()
    builders = DeviceNavBuilders(ResidentFeatures(feats, keys))
    out = rollout(model, walk, builders, t(ep.txt_ids), t(ep.txt_masks), t(ep.imagine_feats), t(ep.imagine_masks),
                  criterion=ops.cross_entropy_sum)
    out["loss"].backward()
    builders.map.check()
    assert len(out["fused"]) == int(g["steps"])
    _close(out["loss"].item(), g["loss"], 1e-4, "loss")
    for i, f in enumerate(out["fused"]):
        assert np.array_equal(out["targets"][i], g[f"target{i}"])
        assert ["|".join("" if k is None else k for k in row) for row in out["gmap_vpids"][i]] == g[f"vpids{i}"].tolist()
        _close(f.detach().float().cpu().numpy(), g[f"fused{i}"], 1e-4, f"fused{i}")
    params = dict(model.named_parameters())
    for n, ref in zip(g["grad_names"].tolist(), g["grad_norms"]):
        if ref < 0:
            assert params[n].grad is None or float(params[n].grad.abs().max()) == 0.0, n
        else:
            nrm = float(params[n].grad.double().norm())
            assert abs(nrm - ref) <= max(2e-4 * ref, 2e-5), (n, nrm, ref)
