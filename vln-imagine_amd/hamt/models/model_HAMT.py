"""`VLNBertCMT` / `Critic` wrappers (drop-in for VLN-HAMT/finetune_src/models/model_HAMT.py:13-96,289-300):
feature dropout on the caller's inputs, mode dispatch into NavCMT, state = txt[CLS] * hist[CLS]."""
import torch
import torch.nn as nn

from vln_imagine_amd import ops
from .vlnbert_init import get_vlnbert_models


def length2mask(length, size=None, device=None):
    """True where position >= length (utils/misc.py:13-18)."""
    size = int(max(length)) if size is None else size
    ar = torch.arange(size, dtype=torch.int64, device=device)
    return ar[None, :] > (torch.as_tensor(length, dtype=torch.int64, device=device) - 1)[:, None]


class VLNBertCMT(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.vln_bert = get_vlnbert_models(args, config=None)
        # this wrapper reads txt_embeds[:, 0] only (the `states` below, like model_HAMT.py:61-72): the last cross-modal layer need not
        # compute the other language rows (NavCMT.visual_lang_rows; identical logits / states / gradients)
        self.vln_bert.visual_lang_rows = "cls"
        self.drop_env = nn.Dropout(p=args.feat_dropout)
        ops.mark_agent_model(self.vln_bert)       # gradients accumulated directly / grouped at the end of the agent's loss.backward() (ops.GradSession)

    def forward(self, mode, txt_ids=None, txt_masks=None, txt_embeds=None, hist_img_feats=None, hist_ang_feats=None,
                hist_pano_img_feats=None, hist_pano_ang_feats=None, hist_embeds=None, hist_lens=None, ob_step=None,
                ob_img_feats=None, ob_ang_feats=None, ob_nav_types=None, ob_masks=None, imagine_pano_img_feats=None,
                imagine_masks=None, imagine_embeds=None, align_txt_embeds=None, align_imagine_embeds=None,
                sub_instr_segs=None, sub_instr_imag_flag=None, noun_phrase_segs=None, obs_instr_ids=None,
                return_states=False, return_cross_attention_probs=False):
        from vln_imagine_amd import graphed
        with graphed.of(self.vln_bert).scope():
            return self._forward(mode, txt_ids, txt_masks, txt_embeds, hist_img_feats, hist_ang_feats, hist_pano_img_feats, hist_pano_ang_feats,
                                 hist_embeds, hist_lens, ob_step, ob_img_feats, ob_ang_feats, ob_nav_types, ob_masks, imagine_pano_img_feats,
                                 imagine_masks, imagine_embeds, align_txt_embeds, align_imagine_embeds, sub_instr_segs, sub_instr_imag_flag,
                                 noun_phrase_segs, obs_instr_ids, return_states, return_cross_attention_probs)

    def _forward(self, mode, txt_ids, txt_masks, txt_embeds, hist_img_feats, hist_ang_feats, hist_pano_img_feats, hist_pano_ang_feats, hist_embeds,
                 hist_lens, ob_step, ob_img_feats, ob_ang_feats, ob_nav_types, ob_masks, imagine_pano_img_feats, imagine_masks, imagine_embeds,
                 align_txt_embeds, align_imagine_embeds, sub_instr_segs, sub_instr_imag_flag, noun_phrase_segs, obs_instr_ids, return_states,
                 return_cross_attention_probs):
        m = self.vln_bert
        from vln_imagine_amd import graphed
        padding = graphed.of(m)._ready() is not None                       # shape buckets only where the per-call graphs can run (graphed.BUCKETS)
        if mode == "language":
            L0 = txt_ids.shape[1]
            if padding:
                Lb = graphed.bucket(L0, graphed.BUCKETS[0])
                txt_ids, txt_masks = graphed.pad_dim(txt_ids, 1, Lb, 0), graphed.pad_dim(txt_masks, 1, Lb, False)
            out = self._graphed(mode, (), lambda **k: m(mode, **k), txt_ids=txt_ids, txt_masks=txt_masks)
            if isinstance(out, (tuple, list)):                           # no_lang_ca: the per-layer text states (a list, :1022-1030)
                return [o[:, :L0] for o in out]
            return out[:, :L0]
        if mode == "imagine":
            def imagine(imagine_pano_img_feats=None, imagine_masks=None):
                if imagine_pano_img_feats is not None:
                    imagine_pano_img_feats = self.drop_env(imagine_pano_img_feats)
                return m(mode, imagine_pano_img_feats=imagine_pano_img_feats, imagine_masks=imagine_masks)
            return self._graphed(mode, (), imagine, imagine_pano_img_feats=imagine_pano_img_feats, imagine_masks=imagine_masks)
        if mode == "align_with_contrastive_loss":
            return m(mode, align_txt_embeds=align_txt_embeds, txt_masks=txt_masks, align_imagine_embeds=align_imagine_embeds,
                     imagine_masks=imagine_masks, sub_instr_segs=sub_instr_segs, sub_instr_imag_flag=sub_instr_imag_flag,
                     noun_phrase_segs=noun_phrase_segs, obs_instr_ids=obs_instr_ids)
        if mode == "history":
            def history(hist_img_feats=None, hist_ang_feats=None, ob_step_ids=None, hist_pano_img_feats=None, hist_pano_ang_feats=None):
                if hist_img_feats is not None:
                    hist_img_feats = self.drop_env(hist_img_feats)
                if hist_pano_img_feats is not None:
                    hist_pano_img_feats = self.drop_env(hist_pano_img_feats)
                return m(mode, hist_img_feats=hist_img_feats, hist_ang_feats=hist_ang_feats, ob_step_ids=ob_step_ids,
                         hist_pano_img_feats=hist_pano_img_feats, hist_pano_ang_feats=hist_pano_ang_feats)
            step = self._step_id(ob_step, m.device) if ob_step is not None else None
            return self._graphed(mode, (), history, hist_img_feats=hist_img_feats, hist_ang_feats=hist_ang_feats, ob_step_ids=step,
                                 hist_pano_img_feats=hist_pano_img_feats, hist_pano_ang_feats=hist_pano_ang_feats)
        if mode == "visual":
            hist = torch.stack(hist_embeds, 1)
            hist_masks = self._hist_mask(hist_lens, hist.size(1), hist.device)
            no_ca = bool(self.args.no_lang_ca)

            def visual(txt_embeds=None, txt_masks=None, hist_embeds=None, hist_masks=None, ob_img_feats=None, ob_ang_feats=None,
                       ob_nav_types=None, ob_masks=None, imagine_embeds=None, imagine_masks=None, lang_side=None):
                outs = m(
                    mode, txt_embeds=txt_embeds, txt_masks=txt_masks, hist_embeds=hist_embeds, hist_masks=hist_masks,
                    ob_img_feats=self.drop_env(ob_img_feats), ob_ang_feats=ob_ang_feats, ob_nav_types=ob_nav_types,
                    ob_masks=ob_masks, imagine_embeds=imagine_embeds, imagine_masks=imagine_masks,
                    return_cross_attention_probs=return_cross_attention_probs, lang_side=lang_side)
                act_logits, txt_o, hist_o, ob_o = outs[:4]
                extra = tuple(outs[4:])                      # (cross_attn_probs, self_attn_probs) per layer when asked for (:80-95)
                if return_states:
                    states = hist_o[:, 0] if no_ca else txt_o[:, 0] * hist_o[:, 0]
                    return (act_logits, states) + extra
                return (act_logits,) + extra

            if isinstance(txt_embeds, list) or return_cross_attention_probs:      # per-layer text states / probability maps: the plain call
                return visual(txt_embeds, txt_masks, hist, hist_masks, ob_img_feats, ob_ang_feats, ob_nav_types, ob_masks, imagine_embeds,
                              imagine_masks)
            V0 = ob_masks.shape[1]
            if padding:
                # text keys and candidate views up to their buckets: padded keys are masked (additive -10000 like every padded token of a ragged
                # batch), padded candidates have navigation type 0 and an off mask, i.e. a -inf logit that is sliced away below
                Lb, Vb = graphed.bucket(txt_masks.shape[1], graphed.BUCKETS[0]), graphed.bucket(V0, graphed.BUCKETS[1])
                txt_embeds, txt_masks = graphed.pad_dim(txt_embeds, 1, Lb), graphed.pad_dim(txt_masks, 1, Lb, False)
                ob_img_feats, ob_ang_feats = graphed.pad_dim(ob_img_feats, 1, Vb), graphed.pad_dim(ob_ang_feats, 1, Vb)
                ob_nav_types, ob_masks = graphed.pad_dim(ob_nav_types, 1, Vb, 0), graphed.pad_dim(ob_masks, 1, Vb, False)

            def visual_own_language_side(**k):
                # one captured call = one self-contained autograd graph: the episode's language side (concatenation, masks, first-layer Q / K / V)
                # is built inside it instead of being shared between the steps' graphs through NavCMT's per-episode cache
                ls = m.language_side(k["txt_embeds"], k["txt_masks"], k.get("imagine_embeds"), k.get("imagine_masks")) if self._graphing() else None
                return visual(lang_side=ls, **k)
            out = self._graphed(mode, (bool(return_states),), visual_own_language_side, txt_embeds=txt_embeds, txt_masks=txt_masks, hist_embeds=hist,
                                hist_masks=hist_masks, ob_img_feats=ob_img_feats, ob_ang_feats=ob_ang_feats, ob_nav_types=ob_nav_types,
                                ob_masks=ob_masks, imagine_embeds=imagine_embeds, imagine_masks=imagine_masks)
            out = out if isinstance(out, tuple) else (out,)
            return (out[0][:, :V0],) + tuple(out[1:])
        raise NotImplementedError("wrong mode: %s" % mode)

    # A host -> device copy of pageable memory waits for everything queued on the stream: one per `history` / `visual` call stalled the host on
    # the GPU a dozen times per rollout (1.2 ms each at the bench's shapes). The step ids and the history masks of an episode are a handful
    # of distinct small values: kept on the device.
    def _step_id(self, ob_step, device):
        c = self.__dict__.setdefault("_step_ids", {})
        t = c.get((ob_step, device))
        if t is None:
            t = c[(ob_step, device)] = torch.tensor([ob_step], dtype=torch.long, device=device)
        return t

    def _hist_mask(self, hist_lens, size, device):
        c = self.__dict__.setdefault("_hist_masks", {})
        key = (tuple(int(n) for n in hist_lens), size, device)
        t = c.get(key)
        if t is None:
            if len(c) > 256:
                c.clear()
            t = c[key] = length2mask(hist_lens, size=size, device=device).logical_not()
        return t

    def _graphing(self):
        from vln_imagine_amd import graphed
        return graphed.of(self.vln_bert).capturing

    def _graphed(self, mode, consts, fn, **named):
        """One autograd node per call where possible (vln_imagine_amd/graphed.py): the tensors that are given, in a fixed order, are the node's
        inputs; everything else about the call is in `consts` and in which names are present."""
        from vln_imagine_amd import graphed
        names = tuple(k for k, v in named.items() if v is not None)
        if not all(torch.is_tensor(named[k]) for k in names):
            return fn(**{k: named[k] for k in names})
        return graphed.of(self.vln_bert).call(mode, (consts, names), lambda *ts: fn(**dict(zip(names, ts))), tuple(named[k] for k in names))


class Critic(nn.Module):
    """state -> value, Linear(768,512) ReLU Dropout Linear(512,1) (model_HAMT.py:289-300)."""

    def __init__(self, args):
        super().__init__()
        self.state2value = nn.Sequential(nn.Linear(768, 512), nn.ReLU(), nn.Dropout(args.dropout), nn.Linear(512, 1))

    def forward(self, state):
        s = self.state2value
        h = ops.linear(state, s[0].weight, s[0].bias, act=2)
        h = s[2](h)
        return ops.row_dot(h, s[3].weight, s[3].bias, None).squeeze()
