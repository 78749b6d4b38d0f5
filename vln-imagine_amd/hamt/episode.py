"""Synthetic HAMT episode driver: the teacher-forced IL rollout of the reference
agent reduced to its model calls and loss assembly.

Follows VLN-HAMT/finetune_src/r2r/agent_cmt.py:392-462 (language / imagine / align),
:492 (history CLS), :498-606 (per-step visual + history), :547 (CE, reduction sum,
ignore_index -100) and :746-752 (loss = ml * train_ml / B + cosine_weight * aux).
`model` is anything with the NavCMT.forward(mode, **kw) contract
(models/vilmodel_cmt.py:999-1205): the reference itself, the CPU oracle, or the HIP
product model, so all three are driven by the very same code.
"""
import torch
import torch.nn.functional as F


class EpisodeTensors:
    """Device-resident copy of a synth.HamtEpisode."""

    def __init__(self, ep, device="cpu"):
        dev = torch.device(device)
        t = lambda a: torch.from_numpy(a).to(dev)
        self.ep = ep
        self.B, self.T = ep.B, ep.T
        self.txt_ids = t(ep.txt_ids)
        self.txt_masks = t(ep.txt_masks)
        self.imagine_feats = t(ep.imagine_feats)
        self.imagine_masks = t(ep.imagine_masks)
        self.steps = []
        for s in ep.steps:
            self.steps.append({k: (t(v) if hasattr(v, "dtype") else v) for k, v in s.items()})
        self.step_ids = [torch.tensor([i], device=dev) for i in range(ep.T)]
        self.hist_masks = []
        for lens in ep.hist_lens:
            n = max(lens)
            m = torch.arange(n)[None, :] < torch.tensor(lens)[:, None]
            self.hist_masks.append(m.to(dev))


def ce_sum(logits, target):
    return F.cross_entropy(logits.float(), target, ignore_index=-100, reduction="sum")


def run_episode(model, et, bypass=True, use_aux=True, train_ml=0.2, cosine_weight=0.5,
                criterion=ce_sum, keep=True):
    """Runs one episode; returns dict with loss terms and (if keep) per-step outputs."""
    ep = et.ep
    out = {"logits": [], "states": [], "hist": [], "txt_o": [], "ob_o": [], "hist_o": []}
    txt_embeds = model("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks)
    imagine_embeds = model("imagine", imagine_pano_img_feats=et.imagine_feats,
                           imagine_masks=None if bypass else et.imagine_masks)
    aux = None
    if use_aux:
        aux, imagine_embeds = model(
            "align_with_contrastive_loss", align_txt_embeds=txt_embeds, txt_masks=et.txt_masks,
            align_imagine_embeds=imagine_embeds, imagine_masks=et.imagine_masks,
            sub_instr_segs=ep.sub_instr_segs, sub_instr_imag_flag=ep.sub_instr_imag_flag,
            noun_phrase_segs=ep.noun_phrase_segs)
    hist = [model("history").expand(et.B, -1)]
    ml_loss = 0.0
    for t, s in enumerate(et.steps):
        hist_embeds = torch.stack(hist, 1)
        logits, txt_o, hist_o, ob_o = model(
            "visual", txt_embeds=txt_embeds, txt_masks=et.txt_masks, hist_embeds=hist_embeds,
            hist_masks=et.hist_masks[t], ob_img_feats=s["ob_img_feats"],
            ob_ang_feats=s["ob_ang_feats"], ob_nav_types=s["ob_nav_types"],
            ob_masks=s["ob_masks"], imagine_embeds=imagine_embeds, imagine_masks=et.imagine_masks)
        ml_loss = ml_loss + criterion(logits, s["target"])
        h = model("history", hist_img_feats=s["hist_img_feats"], hist_ang_feats=s["hist_ang_feats"],
                  ob_step_ids=et.step_ids[t],
                  hist_pano_img_feats=s["hist_pano_img_feats"],
                  hist_pano_ang_feats=s["hist_pano_ang_feats"])
        hist.append(h)
        if keep:
            out["logits"].append(logits)
            out["states"].append(txt_o[:, 0] * hist_o[:, 0])   # model_HAMT.py:86
            out["hist"].append(h)
            out["txt_o"].append(txt_o); out["ob_o"].append(ob_o); out["hist_o"].append(hist_o)
    loss = ml_loss * train_ml / et.B
    if use_aux and torch.is_tensor(aux):
        loss = loss + cosine_weight * aux
    out.update(loss=loss, ml_loss=ml_loss, aux=aux, txt_embeds=txt_embeds,
               imagine_embeds=imagine_embeds, hist_cls=hist[0])
    return out


def run_episode_time_batched(model, et, bypass=True, use_aux=True, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum):
    """The same teacher-forced episode with all T steps executed as ONE batch of T*B samples.

    Under teacher forcing every step's observation and history INPUTS are known up front (agent_cmt.py:561-562), so the
    T `history` calls collapse into one call on [T*B] rows and the T `visual` calls into one call whose sample (t, b)
    sees the history prefix [CLS, h_0 .. h_{t-1}] (padded to T entries, masked exactly like ended episodes are in the
    reference, model_HAMT.py:62-63). Logits, loss and gradients equal the step-by-step rollout; only the GEMM row
    count (x T) and the launch count (/ T) change. Sampling / RL rollouts cannot use this (actions feed back)."""
    ep, B, T = et.ep, et.B, et.T
    dev = et.txt_ids.device
    txt = model("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks)
    img = model("imagine", imagine_pano_img_feats=et.imagine_feats, imagine_masks=None if bypass else et.imagine_masks)
    aux = None
    if use_aux:
        aux, img = model("align_with_contrastive_loss", align_txt_embeds=txt, txt_masks=et.txt_masks, align_imagine_embeds=img,
                         imagine_masks=et.imagine_masks, sub_instr_segs=ep.sub_instr_segs,
                         sub_instr_imag_flag=ep.sub_instr_imag_flag, noun_phrase_segs=ep.noun_phrase_segs)
    cat = lambda k: torch.cat([s[k] for s in et.steps], 0)
    cls = model("history").expand(B, -1)                                                   # [B, H]
    h_all = model("history", hist_img_feats=cat("hist_img_feats"), hist_ang_feats=cat("hist_ang_feats"),
                  ob_step_ids=torch.arange(T, device=dev).repeat_interleave(B),
                  hist_pano_img_feats=cat("hist_pano_img_feats"), hist_pano_ang_feats=cat("hist_pano_ang_feats"))
    H = h_all.shape[-1]
    hist_steps = h_all.view(T, B, H)
    # sample (t, b): [CLS, h_0 .. h_{T-2}] with the first t+1 entries valid
    prefix = torch.cat([cls.to(h_all.dtype).unsqueeze(0), hist_steps[:T - 1]], 0)          # [T, B, H] entries 0..T-1
    hist = prefix.permute(1, 0, 2).unsqueeze(0).expand(T, B, T, H).reshape(T * B, T, H)
    hist_masks = (torch.arange(T, device=dev)[None, :] <= torch.arange(T, device=dev)[:, None])    # [t, entry]
    hist_masks = hist_masks.unsqueeze(1).expand(T, B, T).reshape(T * B, T)
    rep = lambda x: x.unsqueeze(0).expand((T,) + tuple(x.shape)).reshape((T * x.shape[0],) + tuple(x.shape[1:]))
    logits, txt_o, hist_o, ob_o = model(
        "visual", txt_embeds=rep(txt), txt_masks=rep(et.txt_masks), hist_embeds=hist, hist_masks=hist_masks,
        ob_img_feats=cat("ob_img_feats"), ob_ang_feats=cat("ob_ang_feats"), ob_nav_types=cat("ob_nav_types"),
        ob_masks=cat("ob_masks"), imagine_embeds=rep(img), imagine_masks=rep(et.imagine_masks))
    ml_loss = criterion(logits, cat("target"))
    loss = ml_loss * train_ml / B
    if use_aux and torch.is_tensor(aux):
        loss = loss + cosine_weight * aux
    return {"loss": loss, "ml_loss": ml_loss, "aux": aux, "logits": list(logits.view(T, B, -1)), "txt_embeds": txt,
            "imagine_embeds": img, "hist": list(hist_steps)}
