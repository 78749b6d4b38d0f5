"""Device-side builders of the per-step model inputs.

The reference agents assemble every step's observation tensors on the host: python loops over candidates, numpy concatenations of
36 x 768 view features per sample, padding, then a ~7 MB host->device copy (VLN-HAMT/finetune_src/r2r/agent_cmt.py:130-176
`_cand_pano_feature_variable`, VLN-DUET/map_nav_src/r2r/agent.py:67-97 `_panorama_feature_variable`). Here the view features of
all viewpoints stay resident in HBM (`ResidentFeatures`, 288 GB hold every R2R scan many times over); per step the host only
decides WHICH view goes into which slot (a few hundred integers) and one kernel (`vlni_build_views`) gathers the rows and fills the
angle features. The slot layout and feature values are exactly the reference's (tests/test_builders_gpu.py restates its loops).
"""
import math

import numpy as np
import torch

from . import _lib, ops


def view_angle_table(angle_feat_size=4):
    """[36 base views][36 views][A]: angle features of the discretised panorama relative to the agent's view, closed form of
    r2r/data_utils.py:506-534 (view ix: heading (ix % 12) * 30 deg, elevation (ix // 12 - 1) * 30 deg; base elevation 0)."""
    t = np.empty((36, 36, angle_feat_size), np.float32)
    for base in range(36):
        bh = (base % 12) * math.radians(30)
        for ix in range(36):
            h, e = (ix % 12) * math.radians(30) - bh, (ix // 12 - 1) * math.radians(30)
            t[base, ix] = np.array([math.sin(h), math.cos(h), math.sin(e), math.cos(e)] * (angle_feat_size // 4), np.float32)
    return t


class ResidentFeatures:
    """View features of every viewpoint, resident on the device: table [n, 36, D] (float32 or bfloat16) + key -> row index.
    Built once from whatever the feature store is (the reference reads an HDF5 file keyed 'scan_viewpoint', r2r/data_utils.py:15-47)."""

    def __init__(self, feats, keys, device="cuda", dtype=torch.float32):
        feats = torch.as_tensor(feats)
        assert feats.dim() == 3 and feats.shape[1] == 36 and len(keys) == feats.shape[0]
        self.table = feats.to(device=device, dtype=dtype).contiguous()
        self.index = {k: i for i, k in enumerate(keys)}
        self.D = feats.shape[2]

    def rows(self, keys):
        return [self.index[k] for k in keys]


class ViewBuilder:
    """Builds [B, V, D] image features, [B, V, A] angle features and [B, V] nav types on the device.

    `obs` is the reference's observation list reduced to what the builders read: per sample a dict with
      'key' (scan_viewpoint), 'viewIndex' (the agent's current discretised view = base view of the angle table),
      'candidate': list of {'pointId', 'heading', 'elevation'} (relative heading / elevation of the candidate, env.py:254,287)."""

    def __init__(self, features, angle_feat_size=4):
        self.f, self.A = features, angle_feat_size
        self.angle_table = torch.from_numpy(view_angle_table(angle_feat_size)).to(features.table.device)

    def _launch(self, rows, view, he, is_cand, base):
        dev = self.f.table.device
        B, V = view.shape
        t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=d, non_blocking=True)
        rows_t, view_t, he_t = t(np.asarray(rows), torch.long), t(view, torch.int32), t(he, torch.float32)
        cand_t, base_t = t(is_cand, torch.uint8), t(np.asarray(base), torch.int32)
        img = torch.empty((B, V, self.f.D), dtype=torch.float32, device=dev)
        ang = torch.empty((B, V, self.A), dtype=torch.float32, device=dev)
        _lib.call("vlni_build_views", ops._dt(self.f.table), self.f.table.data_ptr(), rows_t.data_ptr(), view_t.data_ptr(), he_t.data_ptr(),
                  cand_t.data_ptr(), base_t.data_ptr(), self.angle_table.data_ptr(), img.data_ptr(), ang.data_ptr(), B, V, self.f.D, self.A,
                  ops._st())
        return img, ang

    def hamt_observation(self, obs, views=36):
        """HAMT `_cand_pano_feature_variable` (agent_cmt.py:130-176): candidates, [STOP] (zeros, nav type 2), the non-candidate
        views in view order (nav type 0), zero padding. Returns (ob_img_fts, ob_ang_fts, ob_nav_types, ob_lens, ob_cand_lens)."""
        B = len(obs)
        lens = [len(ob["candidate"]) + 1 + views - len({c["pointId"] for c in ob["candidate"]}) for ob in obs]
        V = max(lens)
        view = np.full((B, V), -1, np.int32)
        he = np.zeros((B, V, 2), np.float32)
        is_cand = np.zeros((B, V), np.uint8)
        nav = np.zeros((B, V), np.int64)
        for i, ob in enumerate(obs):
            n = len(ob["candidate"])
            used = set()
            for j, c in enumerate(ob["candidate"]):
                view[i, j], he[i, j], is_cand[i, j], nav[i, j] = c["pointId"], (c["heading"], c["elevation"]), 1, 1
                used.add(c["pointId"])
            nav[i, n] = 2                                          # [STOP]: zero features
            rest = [v for v in range(views) if v not in used]
            view[i, n + 1:n + 1 + len(rest)] = rest
        img, ang = self._launch(self.f.rows([ob["key"] for ob in obs]), view, he, is_cand, [ob["viewIndex"] for ob in obs])
        return img, ang, torch.from_numpy(nav).to(img.device), lens, [len(ob["candidate"]) + 1 for ob in obs]

    def hamt_candidates(self, obs):
        """HAMT `_candidate_variable` (agent_cmt.py:178-196, `ob_type == 'cand'`): the candidates, then [END] (zeros, nav type 2)."""
        B = len(obs)
        lens = [len(ob["candidate"]) + 1 for ob in obs]
        V = max(lens)
        view = np.full((B, V), -1, np.int32)
        he = np.zeros((B, V, 2), np.float32)
        is_cand = np.zeros((B, V), np.uint8)
        nav = np.zeros((B, V), np.int64)
        for i, ob in enumerate(obs):
            for j, c in enumerate(ob["candidate"]):
                view[i, j], he[i, j], is_cand[i, j], nav[i, j] = c["pointId"], (c["heading"], c["elevation"]), 1, 1
            nav[i, lens[i] - 1] = 2
        img, ang = self._launch(self.f.rows([ob["key"] for ob in obs]), view, he, is_cand, [ob["viewIndex"] for ob in obs])
        return img, ang, torch.from_numpy(nav).to(img.device), lens

    def hamt_history(self, obs, next_ids=None, views=36):
        """HAMT `_history_variable` + the previous-action angle (agent_cmt.py:198-215,589-594), one launch: the view the agent looks
        through, the whole panorama with its angle table, and the angle feature of the candidate it moves to (`next_ids[b]`, -1 or
        None = no move -> zeros). Returns (hist_img_feats [B, D], hist_pano_img_feats [B, 36, D], hist_pano_ang_feats [B, 36, A],
        prev_act_angle [B, A])."""
        B, V = len(obs), views + 2
        view = np.full((B, V), -1, np.int32)
        he = np.zeros((B, V, 2), np.float32)
        is_cand = np.zeros((B, V), np.uint8)
        view[:, :views] = np.arange(views)[None, :]
        for i, ob in enumerate(obs):
            view[i, views] = ob["viewIndex"]
            nid = -1 if next_ids is None else int(next_ids[i])
            if nid >= 0:
                c = ob["candidate"][nid]
                view[i, views + 1], he[i, views + 1], is_cand[i, views + 1] = c["pointId"], (c["heading"], c["elevation"]), 1
        img, ang = self._launch(self.f.rows([ob["key"] for ob in obs]), view, he, is_cand, [ob["viewIndex"] for ob in obs])
        return img[:, views], img[:, :views], ang[:, :views], ang[:, views + 1]

    def duet_panorama(self, obs, views=36):
        """DUET `_panorama_feature_variable` (map_nav_src/r2r/agent.py:67-97): candidate views first (nav type 1), then the
        views no candidate uses (nav type 0); loc_fts = [angle(4), box (1, 1, 1)]. Returns the `panorama` batch entries."""
        B = len(obs)
        W = max(len(ob["candidate"]) + views - len({c["pointId"] for c in ob["candidate"]}) for ob in obs)   # candidates may share a view
        view = np.full((B, W), -1, np.int32)
        he = np.zeros((B, W, 2), np.float32)
        is_cand = np.zeros((B, W), np.uint8)
        nav = np.zeros((B, W), np.int64)
        lens, cand_vpids = [], []
        for i, ob in enumerate(obs):
            used, n = set(), 0
            for c in ob["candidate"]:
                view[i, n], he[i, n], is_cand[i, n], nav[i, n] = c["pointId"], (c["heading"], c["elevation"]), 1, 1
                used.add(c["pointId"])
                n += 1
            rest = [v for v in range(views) if v not in used]
            view[i, n:n + len(rest)] = rest
            lens.append(n + len(rest))
            cand_vpids.append([c.get("viewpointId") for c in ob["candidate"]])
        V = max(lens)
        img, ang = self._launch(self.f.rows([ob["key"] for ob in obs]), view[:, :V], he[:, :V], is_cand[:, :V], [ob["viewIndex"] for ob in obs])
        valid = torch.from_numpy((view[:, :V] >= 0)).to(img.device)
        loc = torch.cat([ang, valid.unsqueeze(2).expand(-1, -1, 3).to(ang.dtype)], 2)            # box features are all ones
        return {"view_img_fts": img, "loc_fts": loc, "nav_types": torch.from_numpy(nav[:, :V]).to(img.device),
                "view_lens": torch.tensor(lens, device=img.device), "cand_vpids": cand_vpids}


class ImaginationTable:
    """Every instruction's imagination features resident on the device + the per-sub-instruction 'True' / 'False' generated flags;
    `batch(instr_ids)` is `_create_diffusion_imaginations_v2` (VLN-HAMT/finetune_src/r2r/agent_cmt.py:247-313, same function in
    VLN-DUET/map_nav_src/r2r/agent.py): slot s of sample b holds the instruction's next stored imagination iff its flag is 'True',
    zeros otherwise; an instruction without any generated imagination counts as length 0. The reference reads each instruction's
    [n_true, >= D] array from an HDF5 file keyed 'pathid_instridx' (r2r/data_utils.py:33-47) and pads on the host every batch."""

    def __init__(self, features, generated_flags, feat_size=768, device="cuda", dtype=torch.float32):
        self.flags, self.D = generated_flags, feat_size
        self.first, rows, n = {}, [], 0
        for iid, fl in generated_flags.items():
            k = sum(f == "True" for f in fl)
            if k:
                a = np.asarray(features[iid])[:, :feat_size].astype(np.float32)
                assert a.shape == (k, feat_size), f"{iid}: {a.shape} stored imaginations for {k} 'True' flags"   # agent_cmt.py:303
                rows.append(a)
            self.first[iid] = n
            n += k
        self.table = torch.from_numpy(np.concatenate(rows, 0) if rows else np.zeros((1, feat_size), np.float32)).to(device=device, dtype=dtype)

    def batch(self, instr_ids):
        flags = [self.flags[i] for i in instr_ids]
        lens = [0 if all(f == "False" for f in fl) else len(fl) for fl in flags]
        B, I = len(instr_ids), max(lens)
        rows = np.full((B, I), -1, np.int64)
        for b, (iid, fl) in enumerate(zip(instr_ids, flags)):
            on = np.flatnonzero([f == "True" for f in fl])
            rows[b, on] = self.first[iid] + np.arange(len(on))
        dev = self.table.device
        out = torch.empty((B, I, self.D), dtype=torch.float32, device=dev)
        rows_t = torch.from_numpy(rows).to(dev)
        _lib.call("vlni_gather_rows_or_zero", ops._dt(self.table), self.table.data_ptr(), self.D, rows_t.data_ptr(), out.data_ptr(), B * I,
                  self.D, ops._st())
        return out, rows_t >= 0


class ResidentObjects:
    """REVERIE / SOON object annotations of every viewpoint, resident on the device (the reference's ObjectFeatureDB reads an HDF5
    file keyed 'scan_viewpoint' per observation, reverie/data_utils.py): per key `n` objects with image features [n, D], angle
    features [n, 4], box features [n, 3] and ids. Rows of all viewpoints are concatenated; loc = [angle | box | 0] in 8 columns."""

    def __init__(self, objects, feat_size=768, device="cuda", dtype=torch.float32):
        """objects: {key: {'obj_img_fts': [n, >= D], 'obj_ang_fts': [n, 4], 'obj_box_fts': [n, 3], 'obj_ids': [n]}} (n may be 0)."""
        self.first, self.count, self.ids, self.D = {}, {}, {}, feat_size
        img, loc, n = [], [], 0
        for k, o in objects.items():
            c = len(o["obj_ids"])
            self.first[k], self.count[k], self.ids[k] = n, c, list(o["obj_ids"])
            if c:
                img.append(np.asarray(o["obj_img_fts"])[:, :feat_size].astype(np.float32))
                loc.append(np.concatenate([np.asarray(o["obj_ang_fts"], np.float32), np.asarray(o["obj_box_fts"], np.float32),
                                           np.zeros((c, 1), np.float32)], 1))
            n += c
        self.img = torch.from_numpy(np.concatenate(img, 0) if img else np.zeros((1, feat_size), np.float32)).to(device=device, dtype=dtype)
        self.loc = torch.from_numpy(np.concatenate(loc, 0) if loc else np.zeros((1, 8), np.float32)).to(device)


def reverie_panorama(view_builder, objects, obs, views=36):
    """REVERIE `_panorama_feature_variable` (VLN-DUET/map_nav_src/reverie/agent_obj.py:52-107): the R2R panorama slots + the
    viewpoint's objects. loc_fts / nav_types hold each sample's views IMMEDIATELY followed by its objects (nav type 2), padded to the
    longest sum; view_img_fts and obj_img_fts are padded separately."""
    out = view_builder.duet_panorama(obs, views)
    dev = out["view_img_fts"].device
    B = len(obs)
    vl = out["view_lens"].tolist()
    ol = [objects.count[ob["key"]] for ob in obs]
    O = max(max(ol), 1)
    rows = np.full((B, O), -1, np.int64)
    for b, ob in enumerate(obs):
        rows[b, :ol[b]] = objects.first[ob["key"]] + np.arange(ol[b])
    rows_t = torch.from_numpy(rows).to(dev)
    obj_img = torch.empty((B, O, objects.D), dtype=torch.float32, device=dev)
    obj_loc = torch.empty((B, O, 8), dtype=torch.float32, device=dev)
    st = ops._st()
    _lib.call("vlni_gather_rows_or_zero", ops._dt(objects.img), objects.img.data_ptr(), objects.D, rows_t.data_ptr(), obj_img.data_ptr(),
              B * O, objects.D, st)
    _lib.call("vlni_gather_rows_or_zero", 0, objects.loc.data_ptr(), 8, rows_t.data_ptr(), obj_loc.data_ptr(), B * O, 8, st)
    # interleave per sample: [views of b | objects of b | padding]
    V = out["loc_fts"].shape[1]
    L = max(v + o for v, o in zip(vl, ol))
    src = np.full((B, L), B * (V + O), np.int64)                  # last row of the pool = zeros
    nav = np.zeros((B, L), np.int64)
    nav_v = out["nav_types"].cpu().numpy()
    for b in range(B):
        src[b, :vl[b]] = b * (V + O) + np.arange(vl[b])
        src[b, vl[b]:vl[b] + ol[b]] = b * (V + O) + V + np.arange(ol[b])
        nav[b, :vl[b]] = nav_v[b, :vl[b]]
        nav[b, vl[b]:vl[b] + ol[b]] = 2
    pool = torch.cat([torch.cat([out["loc_fts"], obj_loc[:, :, :7]], 1).reshape(B * (V + O), 7), torch.zeros((1, 7), device=dev)], 0)
    loc = pool.index_select(0, torch.from_numpy(src.reshape(-1)).to(dev)).view(B, L, 7)
    out.update(obj_img_fts=obj_img[:, :max(ol)] if max(ol) else obj_img[:, :0], loc_fts=loc, nav_types=torch.from_numpy(nav).to(dev),
               obj_lens=torch.tensor(ol, device=dev), obj_ids=[objects.ids[ob["key"]] for ob in obs])
    return out
