"""Synthetic DUET episode driver: GMapNavAgent.rollout reduced to its model calls, the GPU-side map bookkeeping
and the loss assembly (VLN-DUET/map_nav_src/r2r/agent.py:409-500 modes, :468-479 node embeddings, :541 CE sum,
:616-623 loss = ml * train_ml / B + 0.5 * aux). `model(mode, batch)` is anything with the
GlocalTextPathNavCMT.forward contract (models/vilmodel.py:1237-1288)."""
import os

import torch
import torch.nn.functional as F


class DuetEpisodeTensors:
    def __init__(self, ep, device="cpu"):
        dev = torch.device(device)
        t = lambda a: torch.from_numpy(a).to(dev)
        self.ep, self.B, self.T, self.device = ep, ep.B, ep.T, dev
        self.txt_ids, self.txt_masks = t(ep.txt_ids), t(ep.txt_masks)
        self.imagine_feats, self.imagine_masks = t(ep.imagine_feats), t(ep.imagine_masks)
        self.steps = [{k: (t(v) if hasattr(v, "dtype") else v) for k, v in s.items()} for s in ep.steps]
        # map-node sources as gather indices into the bank [zero row | avg_0, pano_0 | avg_1, pano_1 | ...]
        import numpy as np
        widths = [int((s["view_lens"] + s["obj_lens"]).max()) if "obj_lens" in s else int(s["view_lens"].max()) for s in ep.steps]
        base = np.concatenate([[1], 1 + np.cumsum([w + 1 for w in widths])])
        self.pano_widths, self.node_idx = widths, []
        for t_, s in enumerate(ep.steps):
            G = s["gmap_masks"].shape[1]
            idx = np.zeros((ep.B, G), np.int64)
            for b, srcs in enumerate(s["node_src"]):
                for j, src in enumerate(srcs):
                    idx[b, j + 1] = base[src[1]] + (0 if src[0] == "avg" else 1 + src[2])
            S_t = int(base[t_ + 1])                                    # bank rows per sample once step t_'s panorama is in
            self.node_idx.append(t((idx + np.arange(ep.B)[:, None] * S_t).reshape(-1)))


_PANORAMA_UPFRONT = os.environ.get("VLNI_PANORAMA_UPFRONT", "1") == "1"     # A/B switch (round 5): TapedEpisode(upfront_panorama=True)


def ce_sum(logits, target):
    return F.cross_entropy(logits.float(), target, ignore_index=-100, reduction="sum")


def run_episode(model, et, use_aux=True, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum, keep=True):
    ep = et.ep
    out = {"fused": [], "global": [], "local": [], "pano": [], "gmap": [], "vp": []}
    txt = model("language", {"txt_ids": et.txt_ids, "txt_masks": et.txt_masks})
    img = model("imagine", {"imagine_feats": et.imagine_feats, "imagine_masks": et.imagine_masks})
    aux = None
    if use_aux:
        batch = {"align_txt_embeds": txt, "txt_masks": et.txt_masks, "align_imagine_embeds": img,
                 "imagine_masks": et.imagine_masks, "obs_instr_ids": [f"i{b}" for b in range(et.B)]}
        if getattr(ep, "O", 0) == 0:             # REVERIE aligns with the whole instruction: no sub-instruction annotations
            batch.update(sub_instr_segs=ep.sub_instr_segs, sub_instr_imag_flag=ep.sub_instr_imag_flag,
                         noun_phrase_segs=ep.noun_phrase_segs)
        aux, img = model("align_with_contrastive_loss", batch)
    ml_loss, og_loss = 0.0, 0.0
    bank = None
    for t, s in enumerate(et.steps):
        has_obj = "obj_img_fts" in s                                           # REVERIE (reverie/agent_obj.py:381-384)
        pano, pano_masks = model("panorama", {
            "view_img_fts": s["view_img_fts"], "obj_img_fts": s.get("obj_img_fts"), "loc_fts": s["loc_fts"],
            "nav_types": s["nav_types"], "view_lens": s["view_lens"], "obj_lens": s.get("obj_lens")})
        plen = s["view_lens"] + s["obj_lens"] if has_obj else s["view_lens"]
        if pano.is_cuda:                                                                  # masked mean, agent.py:468-469
            from vln_imagine_amd import ops
            avg = ops.seq_mean(pano, plen.contiguous())
        else:                                                                             # the CPU oracle driven through this loop (tests)
            avg = (pano * pano_masks.unsqueeze(2)).sum(1) / plen.to(pano.dtype)[:, None]
        assert pano.shape[1] == et.pano_widths[t]
        if bank is None:
            bank = [torch.zeros_like(avg).unsqueeze(1)]
        bank += [avg.unsqueeze(1), pano]
        # node j of sample b = one row of an earlier step's panorama output (the agent's per-node python lists + pad_tensors_wgrad,
        # agent.py:100-134), here one gather
        rows = torch.cat(bank, 1)
        gmap_img = rows.reshape(-1, rows.shape[2]).index_select(0, et.node_idx[t]).view(et.B, -1, rows.shape[2])
        vp_img = torch.cat([torch.zeros_like(pano[:, :1]), pano], 1)           # agent.py:164-166
        ones = torch.ones(et.B, 1, dtype=torch.bool, device=pano.device)
        vlen1 = plen + 1
        vp_masks = torch.arange(vp_img.shape[1], device=pano.device)[None, :] < vlen1[:, None]
        nav = model("navigation", {
            "txt_embeds": txt, "txt_masks": et.txt_masks, "gmap_img_embeds": gmap_img,
            "gmap_step_ids": s["gmap_step_ids"], "gmap_pos_fts": s["gmap_pos_fts"], "gmap_masks": s["gmap_masks"],
            "gmap_pair_dists": s["gmap_pair_dists"], "gmap_visited_masks": s["gmap_visited_masks"],
            "gmap_vpids": s["gmap_vpids"], "vp_img_embeds": vp_img, "vp_pos_fts": s["vp_pos_fts"],
            "vp_masks": vp_masks, "vp_nav_masks": torch.cat([ones, s["nav_types"] == 1], 1),
            "vp_obj_masks": torch.cat([~ones, s["nav_types"] == 2], 1) if has_obj else None,
            "vp_cand_vpids": s["vp_cand_vpids"],
            "imagine_embeds": img, "imagine_masks": et.imagine_masks})
        ml_loss = ml_loss + criterion(nav["fused_logits"], s["target"])
        if has_obj:                                                            # object grounding CE, agent_obj.py:461-463
            og_loss = og_loss + criterion(nav["obj_logits"], s["obj_target"])
            if keep:
                out.setdefault("obj", []).append(nav["obj_logits"])
        if keep:
            out["fused"].append(nav["fused_logits"]); out["global"].append(nav["global_logits"])
            out["local"].append(nav["local_logits"]); out["pano"].append(pano)
            out["gmap"].append(nav["gmap_embeds"]); out["vp"].append(nav["vp_embeds"])
    loss = ml_loss * train_ml / et.B
    if torch.is_tensor(og_loss):
        loss = loss + og_loss * train_ml / et.B                                # agent_obj.py:544-546
    if use_aux and torch.is_tensor(aux):
        loss = loss + cosine_weight * aux
    out.update(loss=loss, ml_loss=ml_loss, og_loss=og_loss, aux=aux, txt_embeds=txt, imagine_embeds=img)
    return out


def _taped_inputs(et):
    """Step inputs of a non-REVERIE episode padded to ONE map size Gmax and concatenated over the T steps (built once per episode):
    padded map nodes are masked out exactly like a smaller map's padding in a ragged batch (agent.py:98-134 pads to the batch maximum)."""
    hit = getattr(et, "_taped", None)
    if hit is not None:
        return hit
    import numpy as np
    ep, B, T, dev = et.ep, et.B, et.T, et.device
    has_obj = "obj_img_fts" in et.steps[0]             # REVERIE / SOON: object tokens behind each sample's views (reverie/agent_obj.py:52-107, 381-384)
    Gs = [s["gmap_masks"].shape[1] for s in et.steps]
    Gmax, P = max(Gs), et.pano_widths[0]
    assert all(w == P for w in et.pano_widths)
    # bank rows: (t, b, 0) = panorama mean of step t, (t, b, 1 + v) = view v of step t's panorama; one zero row at the very end
    ZERO = T * B * (P + 1)
    pad2 = lambda x, G: torch.cat([x, x.new_zeros((B, Gmax - G) + tuple(x.shape[2:]))], 1) if G < Gmax else x
    steps, off = [], np.full((T, B, Gmax), ZERO, np.int64)
    for t, s in enumerate(et.steps):
        G = Gs[t]
        for b, srcs in enumerate(ep.steps[t]["node_src"]):
            for j, src in enumerate(srcs):
                off[t, b, j + 1] = (src[1] * B + b) * (P + 1) + (0 if src[0] == "avg" else 1 + src[2])
        pd = s["gmap_pair_dists"]
        pdp = pd.new_zeros((B, Gmax, Gmax))
        pdp[:, :G, :G] = pd
        ones = torch.ones(B, 1, dtype=torch.bool, device=dev)
        plen = s["view_lens"] + s["obj_lens"] if has_obj else s["view_lens"]      # tokens of the panorama: views (+ objects)
        extra = {}
        if has_obj:
            extra = dict(obj_img_fts=s["obj_img_fts"], obj_lens=s["obj_lens"], obj_target=s["obj_target"], pano_lens=plen.contiguous(),
                         vp_obj_masks=torch.cat([~ones, s["nav_types"] == 2], 1))                  # agent_obj.py:150-152
        steps.append(dict(
            extra, view_img_fts=s["view_img_fts"], loc_fts=s["loc_fts"], nav_types=s["nav_types"], view_lens=s["view_lens"],
            gmap_step_ids=pad2(s["gmap_step_ids"], G), gmap_pos_fts=pad2(s["gmap_pos_fts"], G), gmap_masks=pad2(s["gmap_masks"], G),
            gmap_pair_dists=pdp, gmap_visited_masks=pad2(s["gmap_visited_masks"], G),
            gmap_vpids=[list(v) + [None] * (Gmax - len(v)) for v in s["gmap_vpids"]],
            vp_pos_fts=s["vp_pos_fts"], vp_masks=torch.arange(P + 1, device=dev)[None, :] < (plen + 1)[:, None],
            pano_masks=torch.arange(P, device=dev)[None, :] < plen[:, None],
            vp_nav_masks=torch.cat([ones, s["nav_types"] == 1], 1), vp_cand_vpids=s["vp_cand_vpids"], target=s["target"]))
    idx = torch.from_numpy(off).to(dev)                                                  # rows of the flattened [T * B * (P + 1) + 1, H] bank
    cat = lambda k: torch.cat([st[k] for st in steps], 0).contiguous()
    full = {k: cat(k) for k in ("view_img_fts", "loc_fts", "nav_types", "view_lens", "gmap_step_ids", "gmap_pos_fts", "gmap_masks",
                                "gmap_pair_dists", "gmap_visited_masks", "vp_pos_fts", "vp_masks", "vp_nav_masks", "pano_masks", "target")
            + (("obj_img_fts", "obj_lens", "obj_target", "pano_lens", "vp_obj_masks") if has_obj else ())}
    full["gmap_vpids"] = [v for st in steps for v in st["gmap_vpids"]]
    full["vp_cand_vpids"] = [v for st in steps for v in st["vp_cand_vpids"]]
    et._taped = (steps, full, idx, Gmax, P, ZERO)
    return et._taped


def _nav_batch(st, gmap_img, vp_img, kvg, kvl, mask):
    return {"txt_embeds": None, "txt_masks": None, "text_kv": (kvg, kvl, mask), "gmap_img_embeds": gmap_img,
            "gmap_step_ids": st["gmap_step_ids"], "gmap_pos_fts": st["gmap_pos_fts"], "gmap_masks": st["gmap_masks"],
            "gmap_pair_dists": st["gmap_pair_dists"], "gmap_visited_masks": st["gmap_visited_masks"], "gmap_vpids": st["gmap_vpids"],
            "vp_img_embeds": vp_img, "vp_pos_fts": st["vp_pos_fts"], "vp_masks": st["vp_masks"], "vp_nav_masks": st["vp_nav_masks"],
            "vp_obj_masks": st.get("vp_obj_masks"), "vp_cand_vpids": st["vp_cand_vpids"], "imagine_embeds": None, "imagine_masks": None,
            "fuse_plan": st.get("fuse_plan"), "masks_add": st.get("masks_add")}


class TapedEpisode:
    """GMapNavAgent.rollout with a step-by-step FORWARD (the caller may pick the next viewpoint from step t's fused logits, agent.py:409-500)
    and ONE episode-batched BACKWARD (vln_imagine_amd.ops.EpisodeTape, see hamt/episode.py:TapedEpisode): the T `panorama` / `navigation`
    calls write their activations into slices of episode-wide buffers, a ghost pass of the same model code over T x B samples records the
    autograd graph, and backward runs on T x longer launches (DUET's per-step launches are 0.2-1.2 k rows: latency-bound). Every step's
    map is padded to the episode's largest (masked like a ragged batch's padding); map-node images are rows of a bank of panorama
    outputs that only ever grows (agent.py:468-479), so the ghost pass gathers every step's nodes from the one full bank. The text-side
    K / V projections are made once per episode (model.project_text). Results equal run_episode's to rounding (tests/test_tape_gpu.py).

    Phases (each may be its own captured graph, duet.buckets): begin() | step(t) -> fused logits [B, Gmax] | finish() -> result dict.
    Step t reads step t's slices of the padded inputs only (`_taped_inputs`: static buffers when `et` is a duet.buckets.DuetEpisodeBuffers, which
    a rollout fills as it goes), finish() all of them."""

    def __init__(self, model, et, tape=None, use_aux=True, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum, ghost_compute=False,
                 feat_dropout=0.0, upfront_panorama=False):
        from vln_imagine_amd import ops
        self.model, self.et, self.B, self.T = model, et, et.B, et.T
        # teacher forcing (the ground-truth path's panoramas are known when the episode starts, agent.py:449-467): the T `panorama` calls as
        # ONE call on T x B samples in begin() (EpisodeTape.record_steps) - 6.9 k-row launches instead of T x 1.2 k-row ones, and the steps
        # are `navigation` calls only. Off for rollouts with the host in the loop (the next viewpoint is the agent's choice).
        self.upfront = bool(upfront_panorama) and _PANORAMA_UPFRONT
        # the wrapper's feature dropout on the panorama image features (VLNBert.drop_env, model.py:20): the drivers call the model itself,
        # so it is applied here with the tape's counter-based masks (step, ghost pass and batched backward see the same mask)
        self.feat_dropout = float(feat_dropout)
        self.tape = tape if tape is not None else ops.EpisodeTape(et.T)
        assert self.tape.T >= et.T
        self.use_aux, self.train_ml, self.cosine_weight, self.criterion, self.ghost_compute = use_aux, train_ml, cosine_weight, criterion, ghost_compute
        self.step_logits = []

    def _drop(self, x):
        from vln_imagine_amd import ops
        return ops.dropout(x, self.feat_dropout, self.model.training) if self.feat_dropout > 0.0 else x

    def _language(self):
        model, et, B = self.model, self.et, self.B
        ep = et.ep
        self.txt = model("language", {"txt_ids": et.txt_ids, "txt_masks": et.txt_masks})
        img = model("imagine", {"imagine_feats": et.imagine_feats, "imagine_masks": et.imagine_masks})
        self.aux = None
        if self.use_aux:
            batch = {"align_txt_embeds": self.txt, "txt_masks": et.txt_masks, "align_imagine_embeds": img, "imagine_masks": et.imagine_masks,
                     "obs_instr_ids": [f"i{b}" for b in range(B)]}
            if getattr(ep, "O", 0) == 0:             # REVERIE aligns with the whole instruction: no sub-instruction annotations (run_episode)
                batch.update(sub_instr_segs=ep.sub_instr_segs, sub_instr_imag_flag=ep.sub_instr_imag_flag, noun_phrase_segs=ep.noun_phrase_segs)
            self.aux, img = model("align_with_contrastive_loss", batch)
        self.img = img
        self.kv_g, self.kv_l, self.lm = model.project_text(self.txt, et.txt_masks, img, et.imagine_masks)   # once per episode, with autograd

    def _pano_batch(self, st):
        """Arguments of a `panorama` call on the step inputs `st` (one step's, or all T x B samples'): views, object tokens where the episode
        has them (each sample's objects follow its views: one row gather inside the model, vilmodel.py:1096-1114)."""
        return {"view_img_fts": self._drop(st["view_img_fts"]), "obj_img_fts": self._drop(st["obj_img_fts"]) if "obj_img_fts" in st else None,
                "loc_fts": st["loc_fts"], "nav_types": st["nav_types"], "view_lens": st["view_lens"], "obj_lens": st.get("obj_lens"),
                "pano_masks": st["pano_masks"]}

    def begin(self):
        et, tape, B, T = self.et, self.tape, self.B, self.T
        self.steps, self.full, self.idx, self.Gmax, self.P, self.ZERO = _taped_inputs(et)
        tape.reset()
        self.step_logits = []
        self._language()
        H, dt, dev, P, ZERO = self.txt.shape[-1], self.kv_g[0].dtype, et.device, self.P, self.ZERO
        # bank [T * B * (P + 1) + 1, H] of panorama outputs (layout: _taped_inputs); rows of later steps and the last row stay zero.
        # vpbuf [T, B, 1 + P, H]: each step's viewpoint tokens, slot 0 = the zero [STOP] embedding (agent.py:164-166)
        bank = getattr(tape, "_bank", None)
        if bank is None or bank.shape != (ZERO + 1, H) or bank.dtype != dt:
            bank = tape._bank = torch.zeros((ZERO + 1, H), dtype=dt, device=dev)
            tape._vpbuf = torch.zeros((T, B, 1 + P, H), dtype=dt, device=dev)
        self.bank, self.bank4, self.vpbuf = bank, bank[:ZERO].view(T, B, P + 1, H), tape._vpbuf
        if self.upfront:
            from vln_imagine_amd import ops
            model, full = self.model, self.full
            # teacher forcing knows the maps of all steps: the additive forms of their masks in two launches instead of two per step
            gm_all, vm_all = ops.additive_mask(full["gmap_masks"]), ops.additive_mask(full["vp_masks"])
            self.full = full = dict(full, masks_add=(gm_all, vm_all))
            self.steps = [dict(st, masks_add=(gm_all[t * B:(t + 1) * B], vm_all[t * B:(t + 1) * B])) for t, st in enumerate(self.steps)]
            with tape.record_steps("panorama", T):               # the call the ghost pass repeats (_batched)
                pano_all, _ = model("panorama", self._pano_batch(full))
            with torch.no_grad():
                pano4 = pano_all.view(T, B, P, H)
                self.bank4[:, :, 0] = ops.seq_mean(pano_all, full.get("pano_lens", full["view_lens"])).view(T, B, H)     # masked mean, agent.py:468-469
                self.bank4[:, :, 1:] = pano4
                self.vpbuf[:, :, 1:] = pano4

    def step(self, t):
        from vln_imagine_amd import ops
        model, tape, B, st = self.model, self.tape, self.B, self.steps[t]
        if not self.upfront:
            with tape.record("panorama", t):
                pano, pmask = model("panorama", self._pano_batch(st))
        with torch.no_grad():
            if not self.upfront:
                self.bank4[t, :, 0] = ops.seq_mean(pano, st.get("pano_lens", st["view_lens"]))             # masked mean, agent.py:468-469
                self.bank4[t, :, 1:] = pano
                self.vpbuf[t][:, 1:] = pano
            gmap_img = self.bank.index_select(0, self.idx[t].reshape(-1)).view(B, self.Gmax, -1)
            vp_img = self.vpbuf[t]
        with tape.record("navigation", t):
            nav = model("navigation", _nav_batch(st, gmap_img, vp_img, self.kv_g, self.kv_l, self.lm))
        self.step_logits.append(nav["fused_logits"])
        return nav["fused_logits"]

    def _batched(self, ctx_pano, ctx_nav):
        """`panorama` and `navigation` on all T x B samples (the ghost pass of the tape; the whole forward under teacher forcing)."""
        from vln_imagine_amd import ops
        model, B, T, full, Gmax, ZERO = self.model, self.B, self.T, self.full, self.Gmax, self.ZERO
        kv_g, kv_l, lm = self.kv_g, self.kv_l, self.lm
        H = self.txt.shape[-1]
        with ctx_pano:
            pano_all, pmask_all = model("panorama", self._pano_batch(full))
        avg_all = ops.seq_mean(pano_all, full.get("pano_lens", full["view_lens"]))              # [T * B, H]
        rows = F.pad(torch.cat([avg_all.unsqueeze(1), pano_all], 1).reshape(ZERO, H), (0, 0, 0, 1))   # the full bank, with autograd
        gmap_all = rows.index_select(0, self.idx.reshape(-1)).view(T * B, Gmax, H)              # step t's nodes only point at steps <= t
        vp_all = F.pad(pano_all, (0, 0, 1, 0))
        rep = lambda x: x.unsqueeze(0).expand((T,) + tuple(x.shape)).reshape((T * x.shape[0],) + tuple(x.shape[1:]))
        Lt = kv_g[0].shape[0] // B
        repkv = lambda kv: kv.view(B, Lt, -1).unsqueeze(0).expand(T, B, Lt, kv.shape[-1]).reshape(T * B * Lt, kv.shape[-1])
        with ctx_nav:
            nav = model("navigation", _nav_batch(full, gmap_all, vp_all, [repkv(k) for k in kv_g], [repkv(k) for k in kv_l], rep(lm)))
        ml_loss = self.criterion(nav["fused_logits"], full["target"])
        loss = ml_loss * self.train_ml / B
        Tn = lambda x: list(x.view((T, B) + tuple(x.shape[1:])))
        og = {}
        if "obj_target" in full:                                                                # object grounding CE, agent_obj.py:461-463, 544-546
            og_loss = self.criterion(nav["obj_logits"], full["obj_target"])
            loss = loss + og_loss * self.train_ml / B
            og = {"og_loss": og_loss, "obj": Tn(nav["obj_logits"])}
        if self.use_aux and torch.is_tensor(self.aux):
            loss = loss + self.cosine_weight * self.aux
        return dict(og, loss=loss, ml_loss=ml_loss, aux=self.aux, fused=Tn(nav["fused_logits"]), **{"global": Tn(nav["global_logits"])},
                    local=Tn(nav["local_logits"]), pano=Tn(pano_all), step_logits=self.step_logits, txt_embeds=self.txt,
                    imagine_embeds=self.img, tape=self.tape, gmax=Gmax)

    def finish(self):
        return self._batched(self.tape.ghost("panorama", compute=self.ghost_compute), self.tape.ghost("navigation", compute=self.ghost_compute))


def run_episode_taped(model, et, tape=None, use_aux=True, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum, on_step=None,
                      ghost_compute=False, feat_dropout=0.0):
    """One episode through TapedEpisode: begin, T steps (`on_step(t, fused_logits)` may pick the next viewpoint), finish."""
    te = TapedEpisode(model, et, tape, use_aux, train_ml, cosine_weight, criterion, ghost_compute, feat_dropout,
                      upfront_panorama=on_step is None)          # nobody picks viewpoints between the steps: teacher forcing
    te.begin()
    for t in range(et.T):
        lg = te.step(t)
        if on_step is not None:
            on_step(t, lg)
    return te.finish()


def run_episode_time_batched(model, et, use_aux=True, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum):
    """Teacher forcing (agent.py:449-467 with the ground-truth path): all T panoramas as ONE `panorama` call on T x B samples and all T
    `navigation` calls as one on maps padded to the episode's largest - forward AND backward on T x longer launches, no tape. The same
    logits, loss and gradients as run_episode to rounding (tests/test_tape_gpu.py); a sampled rollout cannot use this."""
    import contextlib
    te = TapedEpisode(model, et, None, use_aux, train_ml, cosine_weight, criterion)
    te.steps, te.full, te.idx, te.Gmax, te.P, te.ZERO = _taped_inputs(et)
    te._language()
    return te._batched(contextlib.nullcontext(), contextlib.nullcontext())
