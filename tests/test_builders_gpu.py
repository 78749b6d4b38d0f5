"""GPU: the device-side observation builders (vln_imagine_amd/builders.py, vlni_build_views) against literal numpy restatements of the
reference agents' host-side builders (VLN-HAMT/finetune_src/r2r/agent_cmt.py:130-176, VLN-DUET/map_nav_src/r2r/agent.py:67-97)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
D, A = 768, 4


def _angle_feature(h, e):                                     # r2r/data_utils.py:481-484
    return np.array([math.sin(h), math.cos(h), math.sin(e), math.cos(e)] * (A // 4), np.float32)


def _fake_env(n_vp=5, B=4, seed=0):
    """Observation dicts as R2RBatch._get_obs builds them (env.py:290-335): 'feature' = [36, D + A] with the angle table of the
    current view appended, candidates carry the view feature at their pointId + their own angle feature."""
    from vln_imagine_amd.builders import view_angle_table
    rng = np.random.RandomState(seed)
    feats = rng.uniform(-0.5, 0.5, (n_vp, 36, D)).astype(np.float32)
    keys = [f"scan_{i}" for i in range(n_vp)]
    table = view_angle_table(A)
    obs = []
    for b in range(B):
        vp = int(rng.randint(n_vp))
        base = int(rng.randint(36))
        cands = []
        for j in range(int(rng.randint(1, 6))):
            pid = int(rng.randint(36))
            if j == 1 and b % 2 == 0:
                pid = cands[0]["pointId"]                       # two candidates may share a view (then a panorama has > 36 tokens)
            h, e = float(rng.uniform(-math.pi, math.pi)), float(rng.uniform(-0.5, 0.5))
            cands.append({"pointId": pid, "heading": h, "elevation": e, "viewpointId": f"vp{b}_{j}",
                          "feature": np.concatenate([feats[vp, pid], _angle_feature(h, e)])})
        obs.append({"key": keys[vp], "viewIndex": base, "candidate": cands,
                    "feature": np.concatenate([feats[vp], table[base]], 1)})
    return feats, keys, obs


def test_hamt_observation_builder_matches_reference_loop():
    from vln_imagine_amd.builders import ResidentFeatures, ViewBuilder
    feats, keys, obs = _fake_env()
    # ---- agent_cmt.py:130-176, restated
    ob_lens, img_l, ang_l, nav_l = [], [], [], []
    for ob in obs:
        ci, ca, ct = [], [], []
        used = np.zeros((36,), bool)
        for cc in ob["candidate"]:
            ci.append(cc["feature"][:D]); ca.append(cc["feature"][D:]); used[cc["pointId"]] = True; ct.append(1)
        ci.append(np.zeros((D,), np.float32)); ca.append(np.zeros((A,), np.float32)); ct.append(2)
        pano = ob["feature"][~used]
        img_l.append(np.concatenate([np.vstack(ci), pano[:, :D]], 0)); ang_l.append(np.concatenate([np.vstack(ca), pano[:, D:]], 0))
        ct.extend([0] * (36 - used.sum())); nav_l.append(ct); ob_lens.append(len(ct))
    V = max(ob_lens)
    pad = lambda a: np.concatenate([a, np.zeros((V - a.shape[0], a.shape[1]), np.float32)], 0)
    ref_img, ref_ang = np.stack([pad(a) for a in img_l]), np.stack([pad(a) for a in ang_l])
    ref_nav = np.stack([np.array(t + [0] * (V - len(t))) for t in nav_l])
    # ---- device builder
    vb = ViewBuilder(ResidentFeatures(feats, keys), A)
    img, ang, nav, lens, cand_lens = vb.hamt_observation(obs)
    assert lens == ob_lens and cand_lens == [len(ob["candidate"]) + 1 for ob in obs]
    assert np.array_equal(img.cpu().numpy(), ref_img) and np.array_equal(nav.cpu().numpy(), ref_nav)
    assert np.abs(ang.cpu().numpy() - ref_ang).max() < 2e-6          # device sinf / cosf vs libm
    # a bfloat16-resident table returns the rounded features
    vb16 = ViewBuilder(ResidentFeatures(feats, keys, dtype=torch.bfloat16), A)
    img16 = vb16.hamt_observation(obs)[0]
    assert torch.equal(img16.cpu(), torch.from_numpy(ref_img).bfloat16().float())


def test_duet_panorama_builder_matches_reference_loop():
    from vln_imagine_amd.builders import ResidentFeatures, ViewBuilder
    feats, keys, obs = _fake_env(seed=3)
    img_l, loc_l, nav_l, lens = [], [], [], []
    for ob in obs:                                               # map_nav_src/r2r/agent.py:67-97, restated
        vi, va, nt, used = [], [], [], set()
        for cc in ob["candidate"]:
            vi.append(cc["feature"][:D]); va.append(cc["feature"][D:]); nt.append(1); used.add(cc["pointId"])
        vi.extend([x[:D] for k, x in enumerate(ob["feature"]) if k not in used])
        va.extend([x[D:] for k, x in enumerate(ob["feature"]) if k not in used])
        nt.extend([0] * (36 - len(used)))
        vi, va = np.stack(vi, 0), np.stack(va, 0)
        img_l.append(vi); loc_l.append(np.concatenate([va, np.ones((len(vi), 3), np.float32)], 1)); nav_l.append(nt); lens.append(len(vi))
    V = max(lens)
    pad = lambda a: np.concatenate([a, np.zeros((V - a.shape[0], a.shape[1]), np.float32)], 0)
    vb = ViewBuilder(ResidentFeatures(feats, keys), A)
    out = vb.duet_panorama(obs)
    assert out["view_lens"].tolist() == lens
    assert np.array_equal(out["view_img_fts"].cpu().numpy(), np.stack([pad(a) for a in img_l]))
    assert np.abs(out["loc_fts"].cpu().numpy() - np.stack([pad(a) for a in loc_l])).max() < 2e-6
    assert np.array_equal(out["nav_types"].cpu().numpy(), np.stack([np.array(t + [0] * (V - len(t))) for t in nav_l]))
    assert out["cand_vpids"] == [[c["viewpointId"] for c in ob["candidate"]] for ob in obs]


def test_hamt_candidate_and_history_builders_match_reference_loops():
    from vln_imagine_amd.builders import ResidentFeatures, ViewBuilder
    feats, keys, obs = _fake_env(seed=7, B=6)
    B = len(obs)
    vb = ViewBuilder(ResidentFeatures(feats, keys), A)
    # _candidate_variable, agent_cmt.py:178-196 restated
    lens = [len(ob["candidate"]) + 1 for ob in obs]
    ci, ca, cn = np.zeros((B, max(lens), D), np.float32), np.zeros((B, max(lens), A), np.float32), np.zeros((B, max(lens)), np.int64)
    for i, ob in enumerate(obs):
        for j, cc in enumerate(ob["candidate"]):
            ci[i, j], ca[i, j], cn[i, j] = cc["feature"][:D], cc["feature"][D:], 1
        cn[i, lens[i] - 1] = 2
    img, ang, nav, got_lens = vb.hamt_candidates(obs)
    assert got_lens == lens and np.array_equal(img.cpu().numpy(), ci) and np.array_equal(nav.cpu().numpy(), cn)
    assert np.abs(ang.cpu().numpy() - ca).max() < 2e-6
    # _history_variable + prev_act_angle, agent_cmt.py:198-215,589-594 restated
    next_ids = [(-1 if i % 3 == 0 else len(ob["candidate"]) - 1) for i, ob in enumerate(obs)]
    hi = np.stack([ob["feature"][ob["viewIndex"], :D] for ob in obs])
    hp, ha = np.stack([ob["feature"][:, :D] for ob in obs]), np.stack([ob["feature"][:, D:] for ob in obs])
    pa = np.zeros((B, A), np.float32)
    for i, nid in enumerate(next_ids):
        if nid != -1:
            pa[i] = obs[i]["candidate"][nid]["feature"][-A:]
    g_hi, g_hp, g_ha, g_pa = vb.hamt_history(obs, next_ids)
    assert np.array_equal(g_hi.cpu().numpy(), hi) and np.array_equal(g_hp.cpu().numpy(), hp)
    assert np.abs(g_ha.cpu().numpy() - ha).max() < 2e-6 and np.abs(g_pa.cpu().numpy() - pa).max() < 2e-6
    assert not g_pa[0].any()


def test_hamt_rollout_with_device_builders_matches_reference_golden(golden_dir):
    """End to end on the GPU: HIP NavCMT (fp32) + resident view features + resident imagination table through hamt/rollout.py against
    the reference NavCMT driven by the reference-style host loops (tests/golden/hamt_rollout.npz): logits, losses, gradient norms."""
    import os
    from tests.golden.variants import HAMT_C1, hamt_rollout_setup
    from tests.test_hamt_gpu import _close, build_product
    from vln_imagine_amd import ops
    from vln_imagine_amd.builders import ResidentFeatures
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.rollout import DeviceObsBuilders, rollout
    g = np.load(os.path.join(golden_dir, "hamt_rollout.npz"))
    walk, feats, keys, ep, imag, flags = hamt_rollout_setup()
    model = build_product(HamtConfig(**HAMT_C1))
    t = lambda a: torch.from_numpy(a).cuda()
    out = rollout(model, walk, DeviceObsBuilders(ResidentFeatures(feats, keys), imag, flags), t(ep.txt_ids), t(ep.txt_masks),
                  annotations=(ep.sub_instr_segs, ep.sub_instr_imag_flag, ep.noun_phrase_segs), criterion=ops.cross_entropy_sum)
    out["loss"].backward()
    assert len(out["logits"]) == int(g["steps"]) and np.array_equal(out["hist_lens"], g["hist_lens"])
    _close(out["loss"].item(), g["loss"], 1e-4, "loss")
    _close(out["aux"].item(), g["aux"], 1e-4, "aux")
    for i, f in enumerate(out["logits"]):
        assert np.array_equal(out["targets"][i], g[f"target{i}"])
        _close(f.detach().float().cpu().numpy(), g[f"logits{i}"], 1e-4, f"logits{i}")
    params = dict(model.named_parameters())
    for n, ref in zip(g["grad_names"].tolist(), g["grad_norms"]):
        if ref < 0:
            assert params[n].grad is None or float(params[n].grad.abs().max()) == 0.0, n
        else:
            nrm = float(params[n].grad.double().norm())
            assert abs(nrm - ref) <= max(2e-4 * ref, 2e-5), (n, nrm, ref)


@pytest.mark.parametrize("kind", ["npz", "hdf5"])
def test_feature_stores_from_disk_feed_the_resident_tables(tmp_path, kind):
    """formats.load_view_features / load_imagination_table (the reference's HDF5 stores - float64, chunked + gzip for the views - or .npz
    archives with the same keys) -> the device builders give what the in-memory construction gives."""
    from oracle import graph_oracle as GO
    from vln_imagine_amd import formats
    from vln_imagine_amd.builders import ViewBuilder
    from vln_imagine_amd.hdf5_lite import write_store
    feats, keys, obs = _fake_env(seed=11)
    wide = np.concatenate([feats, np.zeros(feats.shape[:2] + (4,), np.float32)], 2)          # stores carry >= D columns
    store = {k: wide[i] for i, k in enumerate(keys)}
    if kind == "npz":
        np.savez(tmp_path / "views.npz", **store)
    else:
        write_store(str(tmp_path / "views.hdf5"), {k: v.astype(np.float64) for k, v in store.items()}, chunks=(9, 256), compress=True)
    table = formats.load_view_features(str(tmp_path / f"views.{kind}"))
    assert sorted(table.index) == sorted(keys)
    img, ang, nav, lens, cl = ViewBuilder(table, A).hamt_observation(obs)
    from vln_imagine_amd.builders import ResidentFeatures
    img2, ang2, nav2, lens2, cl2 = ViewBuilder(ResidentFeatures(feats, keys), A).hamt_observation(obs)
    assert torch.equal(img, img2) and torch.equal(ang, ang2) and torch.equal(nav, nav2) and lens == lens2 and cl == cl2
    flags = {"7_0": ["True", "False", "True"], "7_1": ["False", "False"], "9_2": ["False", "True"]}
    rng = np.random.default_rng(3)
    imag = {"7_0": rng.standard_normal((2, 770)).astype(np.float32), "9_2": rng.standard_normal((1, 770)).astype(np.float32)}
    if kind == "npz":
        np.savez(tmp_path / "imag.npz", **imag)
    else:
        write_store(str(tmp_path / "imag.hdf5"), imag)
    t = formats.load_imagination_table(str(tmp_path / f"imag.{kind}"), flags)
    f, m = t.batch(["9_2", "7_1", "7_0"])
    rf, rm = GO.imaginations_v2(["9_2", "7_1", "7_0"], flags, imag)
    assert np.array_equal(f.cpu().numpy(), rf) and np.array_equal(m.cpu().numpy(), rm)


def test_reverie_panorama_builder_matches_reference_loop():
    """Views + objects (reverie/agent_obj.py:52-107 restated): loc_fts / nav_types interleave each sample's views and objects."""
    from vln_imagine_amd.builders import ResidentFeatures, ResidentObjects, ViewBuilder, reverie_panorama
    feats, keys, obs = _fake_env(seed=5, B=5)
    rng = np.random.RandomState(9)
    objects = {}
    for i, k in enumerate(keys):
        n = [3, 0, 5, 1, 2][i % 5]
        objects[k] = {"obj_img_fts": rng.uniform(-1, 1, (n, D + 2)).astype(np.float32), "obj_ang_fts": rng.uniform(-1, 1, (n, 4)).astype(np.float32),
                      "obj_box_fts": rng.uniform(0, 1, (n, 3)).astype(np.float32), "obj_ids": [f"o{i}_{j}" for j in range(n)]}
    for ob in obs:                                               # what ObjectFeatureDB adds to an observation
        ob.update({k2: objects[ob["key"]][k2] for k2 in ("obj_img_fts", "obj_ang_fts", "obj_box_fts", "obj_ids")})
    vi_l, oi_l, loc_l, nav_l = [], [], [], []
    for ob in obs:
        vi, va, nt, used = [], [], [], set()
        for cc in ob["candidate"]:
            vi.append(cc["feature"][:D]); va.append(cc["feature"][D:]); nt.append(1); used.add(cc["pointId"])
        vi.extend([x[:D] for k, x in enumerate(ob["feature"]) if k not in used])
        va.extend([x[D:] for k, x in enumerate(ob["feature"]) if k not in used])
        nt.extend([0] * (36 - len(used)))
        vloc = np.concatenate([np.stack(va), np.ones((len(vi), 3), np.float32)], 1)
        oloc = np.concatenate([ob["obj_ang_fts"], ob["obj_box_fts"]], 1).reshape(-1, 7)
        nt.extend([2] * len(oloc))
        vi_l.append(np.stack(vi)); oi_l.append(ob["obj_img_fts"][:, :D].reshape(-1, D)); loc_l.append(np.concatenate([vloc, oloc], 0)); nav_l.append(nt)
    pad = lambda arrs: np.stack([np.concatenate([a, np.zeros((max(len(x) for x in arrs) - len(a),) + a.shape[1:], np.float32)], 0) for a in arrs])
    out = reverie_panorama(ViewBuilder(ResidentFeatures(feats, keys), A), ResidentObjects(objects), obs)
    assert np.array_equal(out["view_img_fts"].cpu().numpy(), pad(vi_l)) and np.array_equal(out["obj_img_fts"].cpu().numpy(), pad(oi_l))
    assert np.abs(out["loc_fts"].cpu().numpy() - pad(loc_l)).max() < 2e-6
    L = max(len(n) for n in nav_l)
    assert np.array_equal(out["nav_types"].cpu().numpy(), np.stack([np.array(n + [0] * (L - len(n))) for n in nav_l]))
    assert out["view_lens"].tolist() == [len(v) for v in vi_l] and out["obj_lens"].tolist() == [len(o) for o in oi_l]
    assert out["obj_ids"] == [ob["obj_ids"] for ob in obs]
