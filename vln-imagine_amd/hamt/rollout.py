"""Seq2SeqCMTAgent.rollout under teacher forcing with the per-step inputs built where the data lives
(VLN-HAMT/finetune_src/r2r/agent_cmt.py:372-760).

`rollout` is the reference's loop: language -> imaginations -> alignment head -> [observation -> visual -> imitation target ->
history token -> move] per step. Everything the reference assembles on the host comes from a `builders` object:
  DeviceObsBuilders   the product: resident view features (`ViewBuilder.hamt_observation` / `hamt_history`) and a resident
                      imagination table (`ImaginationTable.batch`), one gather launch each
  (tests plug the CPU restatement of the reference's loops in here to check the whole chain end to end)
The environment is reduced to a list of observations per step (`synth.GraphWalk`, standing in for env._get_obs()).
"""
import numpy as np
import torch

from ..builders import ImaginationTable, ViewBuilder
from .episode import ce_sum


class DeviceObsBuilders:
    def __init__(self, features, imag_feats, imag_flags, device="cuda"):
        self.views, self.dev = ViewBuilder(features), device
        self.table = ImaginationTable(imag_feats, imag_flags, device=device)

    def observation(self, obs):
        return self.views.hamt_observation(obs)

    def history(self, obs, next_ids):
        return self.views.hamt_history(obs, next_ids)

    def imaginations(self, instr_ids):
        return self.table.batch(instr_ids)

    def targets(self, a):
        return torch.from_numpy(a).to(self.dev)


def teacher_targets(walk, t, obs, ended):
    """_teacher_action, agent_cmt.py:315-340: the candidate that is the next ground-truth viewpoint, [STOP] (= number of candidates)
    where the path ends, ignore index for ended episodes."""
    a = np.zeros((walk.B,), np.int64)
    for b, ob in enumerate(obs):
        if ended[b]:
            a[b] = -100
        elif t < walk.length[b] - 1:
            nxt = walk.steps[t + 1][b]["viewpoint"]
            a[b] = [c["viewpointId"] for c in ob["candidate"]].index(nxt)
        else:
            a[b] = len(ob["candidate"])
    return a


def rollout(model, walk, builders, txt_ids, txt_masks, annotations=None, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum):
    """`annotations` = (sub_instr_segs, sub_instr_imag_flag, noun_phrase_segs) switches the alignment head on.
    Returns {'loss', 'aux', 'logits': [per step], 'targets': [per step], 'hist_lens'}."""
    B = walk.B
    dev = txt_ids.device
    obs = walk.steps[0]
    txt = model("language", txt_ids=txt_ids, txt_masks=txt_masks)
    imagine_feats, imagine_masks = builders.imaginations([ob["instr_id"] for ob in obs])
    img = model("imagine", imagine_pano_img_feats=imagine_feats, imagine_masks=None)
    aux = None
    if annotations is not None:
        aux, img = model("align_with_contrastive_loss", align_txt_embeds=txt, txt_masks=txt_masks, align_imagine_embeds=img,
                         imagine_masks=imagine_masks, sub_instr_segs=annotations[0], sub_instr_imag_flag=annotations[1],
                         noun_phrase_segs=annotations[2])
    hist = [model("history").expand(B, -1)]
    hist_lens = np.ones((B,), np.int64)
    ended = np.zeros((B,), bool)
    out = {"logits": [], "targets": []}
    ml = 0.0
    for t in range(walk.T):
        ob_img, ob_ang, nav_types, ob_lens, cand_lens = builders.observation(obs)
        ob_masks = torch.arange(ob_img.shape[1], device=dev)[None, :] < torch.as_tensor(ob_lens, device=dev)[:, None]
        hl = torch.as_tensor(hist_lens, device=dev)
        hist_masks = torch.arange(int(hist_lens.max()), device=dev)[None, :] < hl[:, None]
        logits = model("visual", txt_embeds=txt, txt_masks=txt_masks, hist_embeds=torch.stack(hist, 1), hist_masks=hist_masks,
                       ob_img_feats=ob_img, ob_ang_feats=ob_ang, ob_nav_types=nav_types, ob_masks=ob_masks, imagine_embeds=img,
                       imagine_masks=imagine_masks)[0]
        a = teacher_targets(walk, t, obs, ended)
        ml = ml + criterion(logits, builders.targets(a))
        out["logits"].append(logits); out["targets"].append(a)
        move = np.where((a == np.array(cand_lens) - 1) | (a == -100) | ended, -1, a)               # agent_cmt.py:578-581
        if not (ended | (move == -1)).all() and t != walk.T - 1:
            hi, hp, ha, pa = builders.history(obs, move)
            hist.append(model("history", hist_img_feats=hi, hist_ang_feats=pa, hist_pano_img_feats=hp, hist_pano_ang_feats=ha,
                              ob_step_ids=torch.tensor([t], device=dev)))
            hist_lens = hist_lens + (~ended)
        if t + 1 < walk.T:
            nxt = walk.steps[t + 1]
            obs = [obs[b] if move[b] == -1 else nxt[b] for b in range(B)]                          # no move: the agent stays
        ended = ended | (move == -1)
        if ended.all():
            break
    loss = ml * train_ml / B
    if torch.is_tensor(aux):
        loss = loss + cosine_weight * aux
    out.update(loss=loss, aux=aux, hist_lens=hist_lens)
    return out
