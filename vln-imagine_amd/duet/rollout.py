"""GMapNavAgent.rollout under teacher forcing with the per-step inputs built where the data lives
(VLN-DUET/map_nav_src/r2r/agent.py:391-623).

`rollout` is the reference's loop: map update -> panorama -> node images -> navigation inputs -> navigation -> imitation target
-> next observation. Everything the reference assembles on the host comes from a `builders` object:
  DeviceNavBuilders   the product: resident view features + `ViewBuilder.duet_panorama`, `DeviceGraphMap` (csrc/graphmap.hip)
  (tests plug the CPU restatement of the reference's loops in here to check the whole chain end to end)
The environment is reduced to a list of observations per step (`synth.GraphWalk`, standing in for env._get_obs()).
"""
import numpy as np
import torch

from ..builders import ViewBuilder
from ..graphmap import DeviceGraphMap
from .episode import ce_sum


class DeviceNavBuilders:
    def __init__(self, features, device="cuda", cap=128):
        self.views, self.dev, self.cap, self.map = ViewBuilder(features), device, cap, None

    def start(self, obs):
        self.map = DeviceGraphMap(obs, cap=self.cap, device=self.dev)

    def observe(self, obs, ended):
        self.map.observe(obs, ended)

    def mark_step(self, obs, t, ended):
        self.map.mark_step(obs, t, ended)

    def panorama(self, obs):
        return self.views.duet_panorama(obs)

    def node_images(self, obs, pano, pano_masks, cand_vpids, ended):
        self.map.update_node_embeds(obs, pano, pano_masks, cand_vpids, ended)

    def navigation(self, obs, pano, pano_inputs):
        nav = self.map.nav_gmap_variable(obs)
        nav.update(self.map.nav_vp_variable(obs, pano, pano_inputs["cand_vpids"], pano_inputs["view_lens"], pano_inputs["nav_types"]))
        return nav

    def targets(self, a):
        return torch.from_numpy(a).to(self.dev)


def teacher_targets(walk, t, gmap_vpids, ended):
    """_teacher_action_r4r, imitation branch (r2r/agent.py:253-262): the map node that is the next ground-truth viewpoint, [stop] at
    the end of the path, ignore index for ended episodes."""
    a = np.zeros((walk.B,), np.int64)
    for b in range(walk.B):
        if ended[b]:
            a[b] = -100
        elif t < walk.length[b] - 1:
            a[b] = gmap_vpids[b].index(walk.steps[t + 1][b]["viewpoint"])
    return a


def rollout(model, walk, builders, txt_ids, txt_masks, imagine_feats=None, imagine_masks=None, train_ml=0.2, criterion=ce_sum):
    """Returns {'loss', 'fused': [per-step fused logits], 'targets': [...], 'gmap_vpids': [...]} ."""
    B = walk.B
    obs = walk.steps[0]
    builders.start(obs)
    txt = model("language", {"txt_ids": txt_ids, "txt_masks": txt_masks})
    img = None
    if imagine_feats is not None:
        img = model("imagine", {"imagine_feats": imagine_feats, "imagine_masks": imagine_masks})
    ended = np.zeros((B,), bool)
    out = {"fused": [], "targets": [], "gmap_vpids": []}
    ml = 0.0
    for t in range(walk.T):
        builders.mark_step(obs, t, ended)
        pin = builders.panorama(obs)
        pano, pano_masks = model("panorama", {"view_img_fts": pin["view_img_fts"], "obj_img_fts": None, "loc_fts": pin["loc_fts"],
                                              "nav_types": pin["nav_types"], "view_lens": pin["view_lens"], "obj_lens": None})
        builders.node_images(obs, pano, pano_masks, pin["cand_vpids"], ended)
        nav_in = builders.navigation(obs, pano, pin)
        nav_in.update(txt_embeds=txt, txt_masks=txt_masks, vp_obj_masks=None,          # (the VLNBert wrapper's defaultdict, model.py:22-24)
                      imagine_embeds=img, imagine_masks=imagine_masks if img is not None else None)
        nav = model("navigation", nav_in)
        a = teacher_targets(walk, t, nav_in["gmap_vpids"], ended)
        ml = ml + criterion(nav["fused_logits"], builders.targets(a))
        out["fused"].append(nav["fused_logits"]); out["targets"].append(a); out["gmap_vpids"].append(nav_in["gmap_vpids"])
        # the episode ends on [stop], when nothing is left to explore, or at the step cap (agent.py:583-589,610)
        just = np.array([(a[b] == 0 and not ended[b]) or nav_in["no_vp_left"][b] or t == walk.T - 1 for b in range(B)])
        if t + 1 < walk.T:
            nxt = walk.steps[t + 1]
            obs = [obs[b] if (ended[b] or just[b]) else nxt[b] for b in range(B)]       # an ended agent is not moved any more
            builders.observe(obs, ended)         # agent.py:604-610: an agent that stopped THIS step still re-observes where it stands
        ended = ended | just
        if ended.all():
            break
    out["loss"] = ml * train_ml / B
    return out
