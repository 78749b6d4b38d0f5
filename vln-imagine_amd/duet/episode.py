"""Synthetic DUET episode driver: GMapNavAgent.rollout reduced to its model calls, the GPU-side map bookkeeping
and the loss assembly (VLN-DUET/map_nav_src/r2r/agent.py:409-500 modes, :468-479 node embeddings, :541 CE sum,
:616-623 loss = ml * train_ml / B + 0.5 * aux). `model(mode, batch)` is anything with the
GlocalTextPathNavCMT.forward contract (models/vilmodel.py:1237-1288)."""
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
        avg = (pano * pano_masks.unsqueeze(2)).sum(1) / plen.to(pano.dtype)[:, None]      # agent.py:468-469
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
