"""`VLNBert` / `Critic` wrappers (drop-in for VLN-DUET/map_nav_src/models/model.py:12-62)."""
import collections

import torch.nn as nn

from vln_imagine_amd import ops
from .vlnbert_init import get_vlnbert_models


class VLNBert(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.vln_bert = get_vlnbert_models(args, config=None)
        self.drop_env = nn.Dropout(p=args.feat_dropout)
        ops.mark_agent_model(self.vln_bert)       # gradients accumulated directly / grouped at the end of the agent's loss.backward() (ops.GradSession)

    NAV_OUT = ("gmap_embeds", "vp_embeds", "global_logits", "local_logits", "fused_logits", "obj_logits")

    def forward(self, mode, batch):
        from vln_imagine_amd import graphed
        with graphed.of(self.vln_bert).scope():          # every model call of the agent on one stream (graphed.ModeGraphs.scope)
            return self._forward(mode, batch)

    def _forward(self, mode, batch):
        batch = collections.defaultdict(lambda: None, batch)
        m = self.vln_bert
        if mode == "panorama":
            def panorama(**k):
                k = collections.defaultdict(lambda: None, k)
                k["view_img_fts"] = self.drop_env(k["view_img_fts"])
                if k.get("obj_img_fts") is not None:
                    k["obj_img_fts"] = self.drop_env(k["obj_img_fts"])
                return m(mode, k)
            return self._graphed(mode, (), panorama, batch, ("view_img_fts", "obj_img_fts", "loc_fts", "nav_types", "view_lens", "obj_lens"))
        if mode in ("language", "imagine"):
            keys = ("txt_ids", "txt_masks") if mode == "language" else ("imagine_feats", "imagine_masks")
            L0 = None
            if mode == "language" and graphed_ready(m):              # text length up to its bucket (graphed.BUCKETS): pad ids / off masks, sliced back
                from vln_imagine_amd import graphed
                L0 = batch["txt_ids"].shape[1]
                Lb = graphed.bucket(L0, graphed.BUCKETS[0])
                batch = dict(batch, txt_ids=graphed.pad_dim(batch["txt_ids"], 1, Lb, 0), txt_masks=graphed.pad_dim(batch["txt_masks"], 1, Lb, False))
            out = self._graphed(mode, (), lambda **k: m(mode, collections.defaultdict(lambda: None, k)), batch, keys)
            return out if L0 is None else out[:, :L0]
        if mode == "navigation":
            import torch
            keys = ("txt_embeds", "txt_masks", "gmap_img_embeds", "gmap_step_ids", "gmap_pos_fts", "gmap_masks", "gmap_pair_dists",
                    "gmap_visited_masks", "vp_img_embeds", "vp_pos_fts", "vp_masks", "vp_nav_masks", "vp_obj_masks", "imagine_embeds", "imagine_masks")
            if batch.get("text_kv") is not None or batch.get("fuse_plan") is not None or batch.get("masks_add") is not None \
                    or not graphed_ready(m):
                return m(mode, batch)                          # an episode driver's own extras: the plain call
            # the fusion's index plan comes from the caller's vpid LISTS: built (and cached per list identity) out here, it enters the call as two
            # tensors, so nothing inside a captured call depends on host data
            src, bw = m._fuse_plan(batch["gmap_vpids"], batch["gmap_visited_masks"], batch["vp_cand_vpids"], batch["gmap_masks"].shape[1],
                                   batch["vp_img_embeds"].shape[1])
            extra = {"fuse_src": src, "fuse_bw": bw}
            # shape buckets (graphed.BUCKETS): text keys and map nodes padded the way the agent pads a ragged batch (agent.py:98-134) - off masks,
            # zero features / distances, "nothing" (-1) in the fusion plan; the padded nodes' logits are -inf and are sliced away below
            from vln_imagine_amd import graphed
            G0 = batch["gmap_masks"].shape[1]
            Lb, Gb = graphed.bucket(batch["txt_masks"].shape[1], graphed.BUCKETS[0]), graphed.bucket(G0, graphed.BUCKETS[2])
            pd = graphed.pad_dim
            batch = dict(batch, txt_embeds=pd(batch["txt_embeds"], 1, Lb), txt_masks=pd(batch["txt_masks"], 1, Lb, False),
                         gmap_img_embeds=pd(batch["gmap_img_embeds"], 1, Gb), gmap_step_ids=pd(batch["gmap_step_ids"], 1, Gb, 0),
                         gmap_pos_fts=pd(batch["gmap_pos_fts"], 1, Gb), gmap_masks=pd(batch["gmap_masks"], 1, Gb, False),
                         gmap_pair_dists=pd(pd(batch["gmap_pair_dists"], 1, Gb), 2, Gb), gmap_visited_masks=pd(batch["gmap_visited_masks"], 1, Gb, False))
            extra["fuse_src"] = pd(src, 1, Gb, -1)

            def navigation(fuse_src=None, fuse_bw=None, **k):
                k = collections.defaultdict(lambda: None, k)
                k["fuse_plan"] = (fuse_src, fuse_bw)
                if self._graphing():
                    # one captured call = one self-contained autograd graph: the text-side K / V projections are made inside it instead of being
                    # shared between the steps' graphs through the model's per-episode cache
                    k["text_kv"] = m.project_text(k["txt_embeds"], k["txt_masks"], k.get("imagine_embeds"), k.get("imagine_masks"))
                out = m(mode, k)
                return tuple(out[n] for n in self.NAV_OUT if out[n] is not None)
            has_obj = batch.get("vp_obj_masks") is not None
            out = self._graphed(mode, (has_obj,), navigation, dict(batch, **extra), keys + ("fuse_src", "fuse_bw"))
            names = [n for n in self.NAV_OUT if n != "obj_logits" or has_obj]
            res = dict(zip(names, out))
            res.setdefault("obj_logits", None)
            if Gb != G0:
                for n in ("gmap_embeds", "global_logits", "fused_logits"):
                    res[n] = res[n][:, :G0]
            return res
        if mode != "align_with_contrastive_loss":
            raise NotImplementedError("wrong mode: %s" % mode)
        return m(mode, batch)

    def _graphing(self):
        from vln_imagine_amd import graphed
        return graphed.of(self.vln_bert).capturing

    def _graphed(self, mode, consts, fn, batch, keys):
        """One autograd node per call where possible (vln_imagine_amd/graphed.py): the tensors of `batch` under `keys`, in that order."""
        import torch
        from vln_imagine_amd import graphed
        names = tuple(k for k in keys if batch.get(k) is not None)
        if not all(torch.is_tensor(batch[k]) for k in names):
            return fn(**{k: batch[k] for k in names})
        return graphed.of(self.vln_bert).call(mode, (consts, names), lambda *ts: fn(**dict(zip(names, ts))), tuple(batch[k] for k in names))


def graphed_ready(model):
    from vln_imagine_amd import graphed
    return graphed.of(model)._ready() is not None


class Critic(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.state2value = nn.Sequential(nn.Linear(768, 512), nn.ReLU(), nn.Dropout(args.dropout), nn.Linear(512, 1))

    def forward(self, state):
        s = self.state2value
        return ops.row_dot(s[2](ops.linear(state, s[0].weight, s[0].bias, act=2)), s[3].weight, s[3].bias, None).squeeze()
