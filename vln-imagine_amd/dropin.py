"""The training iteration an UNCHANGED reference agent runs against the drop-in modules - the number a user gets by switching PYTHONPATH and
nothing else: the `VLNBertCMT` / `VLNBert` wrappers called eagerly mode by mode, plain autograd, one
`loss.backward()`, `clip_grad_norm_`, `torch.optim.AdamW`. No episode tape, no FlatTrainer, no captured graph.

HAMT: Seq2SeqCMTAgent.rollout + train (VLN-HAMT/finetune_src/r2r/agent_cmt.py:400-462 language / imagine / align, :492 history [CLS],
:498-606 per-step `visual` (return_states) + `history`, :547 CrossEntropyLoss(ignore_index, sum), :746-752 loss assembly, :827-832 backward /
clip / step). DUET: GMapNavAgent.rollout (VLN-DUET/map_nav_src/r2r/agent.py:409-500) through duet.episode.run_episode, which already speaks
the wrapper's `(mode, batch)` interface; agent_base.py:223-228 backward / clip / step.
Used by bench.py (`extras.drop_in_eager`) and tests/test_wrappers_gpu.py."""
import argparse

import torch
import torch.nn as nn

from . import ops


def wrap_hamt(navcmt, feat_dropout=0.4):
    """A models.model_HAMT.VLNBertCMT around an existing NavCMT (the wrapper's own constructor builds a new model from run arguments)."""
    from vln_imagine_amd.hamt.models.model_HAMT import VLNBertCMT
    w = VLNBertCMT.__new__(VLNBertCMT)
    nn.Module.__init__(w)
    w.args = argparse.Namespace(feat_dropout=feat_dropout, no_lang_ca=bool(navcmt.config.no_lang_ca))
    w.vln_bert = navcmt
    navcmt.visual_lang_rows = "cls"            # what VLNBertCMT.__init__ selects (it reads txt_embeds[:, 0] only)
    w.drop_env = nn.Dropout(p=feat_dropout)
    ops.mark_agent_model(navcmt)                # as VLNBertCMT.__init__
    return w


def wrap_duet(model, feat_dropout=0.4):
    from vln_imagine_amd.duet.models.model import VLNBert
    w = VLNBert.__new__(VLNBert)
    nn.Module.__init__(w)
    w.args = argparse.Namespace(feat_dropout=feat_dropout)
    w.vln_bert = model
    w.drop_env = nn.Dropout(p=feat_dropout)
    ops.mark_agent_model(model)                 # as VLNBert.__init__
    return w


def hamt_agent_loss(w, et, train_ml=0.2, cosine_weight=0.5, use_aux=True, bypass=True, use_imagine=True, keep=None):
    """One teacher-forced rollout through the VLNBertCMT wrapper, the agent's own call sequence; returns (loss, per-step logits).
    bypass / use_imagine: as hamt.episode.run_episode (agent_cmt.py:419-462 guards the imagination calls with args.imagine_enc_pano);
    keep: a dict that receives what the golden tests compare (aux, states, history tokens, text / imagination embeddings)."""
    ep = et.ep
    B = et.B
    criterion = nn.CrossEntropyLoss(ignore_index=-100, reduction="sum")           # agent_cmt.py:105
    txt = w("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks)
    img = w("imagine", imagine_pano_img_feats=et.imagine_feats, imagine_masks=None if bypass else et.imagine_masks) if use_imagine else None
    aux = None
    use_aux = use_aux and use_imagine
    if use_aux:
        aux, img = w("align_with_contrastive_loss", align_txt_embeds=txt, txt_masks=et.txt_masks, align_imagine_embeds=img,
                     imagine_masks=et.imagine_masks, sub_instr_segs=ep.sub_instr_segs, sub_instr_imag_flag=ep.sub_instr_imag_flag,
                     noun_phrase_segs=ep.noun_phrase_segs)
    hist = [w("history").expand(B, -1)]                                            # :492
    hist_lens = [1] * B
    ml_loss, logits = 0.0, []
    for t, s in enumerate(et.steps):
        lg, _states = w("visual", txt_embeds=txt, txt_masks=et.txt_masks, hist_embeds=hist, hist_lens=hist_lens,
                        ob_img_feats=s["ob_img_feats"], ob_ang_feats=s["ob_ang_feats"], ob_nav_types=s["ob_nav_types"], ob_masks=s["ob_masks"],
                        return_states=True, imagine_embeds=img, imagine_masks=et.imagine_masks if use_imagine else None)
        ml_loss = ml_loss + criterion(lg.float(), s["target"])
        logits.append(lg)
        hist.append(w("history", hist_img_feats=s["hist_img_feats"], hist_ang_feats=s["hist_ang_feats"],
                      hist_pano_img_feats=s["hist_pano_img_feats"], hist_pano_ang_feats=s["hist_pano_ang_feats"], ob_step=t))
        hist_lens = [n + 1 for n in hist_lens]
        if keep is not None:
            keep.setdefault("states", []).append(_states)
            keep.setdefault("hist", []).append(hist[-1])
    loss = ml_loss * train_ml / B                                                  # :746-752
    if use_aux and torch.is_tensor(aux):
        loss = loss + cosine_weight * aux
    if keep is not None:
        keep.update(aux=aux, txt_embeds=txt, imagine_embeds=img, hist_cls=hist[0])
    return loss, logits


def duet_agent_loss(w, et):
    from vln_imagine_amd.duet.episode import run_episode
    criterion = nn.CrossEntropyLoss(ignore_index=-100, reduction="sum")           # agent_base.py:162 / agent.py:541
    out = run_episode(w, et, criterion=lambda lg, tgt: criterion(lg.float(), tgt), keep=False)
    return out["loss"], None


class DropInTrainer:
    """optimizer.zero_grad() -> rollout -> loss.backward() -> clip_grad_norm_(40) -> optimizer.step(), as agent_cmt.py:809-832 does it."""

    def __init__(self, wrapper, et, family, lr=1e-5):
        self.w, self.et, self.family = wrapper, et, family
        self.params = [p for p in wrapper.parameters() if p.requires_grad]
        self.opt = torch.optim.AdamW(self.params, lr=lr)                            # agent_cmt.py:98 optimizer(self.vln_bert.parameters(), lr)

    def step(self):
        self.opt.zero_grad()
        loss, _ = (hamt_agent_loss if self.family == "hamt" else duet_agent_loss)(self.w, self.et)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.params, 40.0)
        self.opt.step()
        return loss
