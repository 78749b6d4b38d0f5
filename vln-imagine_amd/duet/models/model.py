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

    def forward(self, mode, batch):
        batch = collections.defaultdict(lambda: None, batch)
        if mode == "panorama":
            batch["view_img_fts"] = self.drop_env(batch["view_img_fts"])
            if batch.get("obj_img_fts") is not None:
                batch["obj_img_fts"] = self.drop_env(batch["obj_img_fts"])
        elif mode not in ("language", "imagine", "align_with_contrastive_loss", "navigation"):
            raise NotImplementedError("wrong mode: %s" % mode)
        return self.vln_bert(mode, batch)


class Critic(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.state2value = nn.Sequential(nn.Linear(768, 512), nn.ReLU(), nn.Dropout(args.dropout), nn.Linear(512, 1))

    def forward(self, state):
        s = self.state2value
        return ops.row_dot(s[2](ops.linear(state, s[0].weight, s[0].bias, act=2)), s[3].weight, s[3].bias, None).squeeze()
