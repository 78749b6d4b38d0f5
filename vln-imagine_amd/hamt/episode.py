"""Synthetic HAMT episode driver: the teacher-forced IL rollout of the reference
agent reduced to its model calls and loss assembly.

Follows VLN-HAMT/finetune_src/r2r/agent_cmt.py:392-462 (language / imagine / align),
:492 (history CLS), :498-606 (per-step visual + history), :547 (CE, reduction sum,
ignore_index -100) and :746-752 (loss = ml * train_ml / B + cosine_weight * aux).
`model` is anything with the NavCMT.forward(mode, **kw) contract
(models/vilmodel_cmt.py:999-1205): the reference itself, the CPU oracle, or the HIP
product model, so all three are driven by the very same code.
"""
import contextlib
import os

import torch
import torch.nn.functional as F


_MASK_IN_GRAPH = int(os.environ.get("VLNI_MASK_IN_GRAPH", "0"))       # tools/stale_mask_repro.py (round-3 anomaly hunt)
_EPISODE_MASKS = os.environ.get("VLNI_EPISODE_MASKS", "1") != "0"     # A/B switch: mask forms of all steps once per episode
_OVERLAP_HISTORY = os.environ.get("VLNI_OVERLAP_HISTORY", "1") != "0"  # A/B switch: the history encoder of step t on a side stream
# Round 5: in which ORDER the two concurrent calls of a step are issued (and so captured). The kernel trace of the replayed step
# (gpurun_out/r5f/last_step.tsv) shows the `visual` chain idle for ~180 us at the start of every step while the queue that got the history
# branch runs its first ~20 small kernels: with the history call captured first the graph executor continues on that branch and starts the
# other one late. 1 = `visual` first (the critical path), then `history` on the second stream.
_HISTORY_AFTER = os.environ.get("VLNI_HISTORY_AFTER", "1") == "1"
# Round 5: teacher forcing knows every step's history inputs when the episode starts (the drivers already rely on that for the history
# masks of all steps), so the T `history` calls are ONE call on T x B samples in begin() (EpisodeTape.record_steps), on the second stream
# beside the text encoder (5.1 k-row launches: half of the chip idle), and the steps' `visual` calls run alone on the chip with no fork /
# join per step. Why: a replayed hipGraph with parallel branches is fed by the host in CAPTURE order at ~8 us per node (a single-stream
# graph: ~2.3 us; tools/graph_fork_probe.py), so the ~32 small launches of a per-step history call captured in front of the step's `visual`
# call held the critical path back by ~180 us per step (gpurun_out/r5f/last_step.tsv). 0 = one history call beside each step's `visual`
# call (rounds 2-4). Sampled rollouts (lag_history=True) are not affected.
_HISTORY_UPFRONT = os.environ.get("VLNI_HISTORY_UPFRONT", "1") == "1"
# Round 5: step t's `history` call INSIDE its `visual` call's launches (3-problem GEMMs). OFF by default - measured on one box
# (gpurun_out/r5c-r5e, 30 steps x 2): side-stream overlap 27.02 ms, lockstep with the panorama's attention / LayerNorm forked to a second stream
# 27.0-27.2, everything on one stream 26.4 -> 26.6 on another box (+0.5); a timing-only build WITHOUT the panorama's attention / LayerNorm
# launches (the bound for 3-problem attention / LayerNorm kernels) 26.65. The step-long launches are one round of tiles either way, so the
# merged launch costs what the unmerged one did and the side stream was already running the history encoder in the idle quarter of the chip;
# what lockstep adds are dependencies (the merged GEMM waits for both chains). The GEMM family's own time falls by 1.2 ms (0.26 -> 0.28 of peak).
_LOCKSTEP_HISTORY = os.environ.get("VLNI_LOCKSTEP_HISTORY", "0") == "1"


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
        self._full = {}
        # history length before each step, [T, B] (model_HAMT.py:62-63), resident: a captured step must not copy host data
        self.hist_lens_dev = torch.tensor(ep.hist_lens, device=dev)
        self.hist_mask_T = torch.arange(ep.T, device=dev)[None, None, :] < self.hist_lens_dev[:, :, None]      # [T, B, T]

    def put_hist_lens(t_self, t, lens):
        """History length of every sample before step t (model_HAMT.py:62-63) - what a sampled rollout calls once step t - 1's action is
        known. Lengths and BOTH mask forms are rewritten together (the padded [T, B, T] one is what the taped / graphed drivers read)."""
        self = t_self
        lens = torch.as_tensor(lens, dtype=torch.int64)
        dev = self.hist_lens_dev.device
        self.hist_lens_dev[t].copy_(lens)
        self.hist_masks[t] = (torch.arange(t + 1)[None, :] < lens[:, None]).to(dev)
        self.hist_mask_T[t].copy_(torch.arange(self.T)[None, :] < lens[:, None])

    def full(self, k):
        """Step inputs `k` of all T steps as ONE [T*B, ...] tensor (step t = rows [t B, (t + 1) B)), built once."""
        v = self._full.get(k)
        if v is None:
            v = self._full[k] = torch.cat([s[k] for s in self.steps], 0).contiguous()
        return v


def ce_sum(logits, target):
    return F.cross_entropy(logits.float(), target, ignore_index=-100, reduction="sum")


def run_episode(model, et, bypass=True, use_aux=True, train_ml=0.2, cosine_weight=0.5,
                criterion=ce_sum, keep=True, use_imagine=True):
    """Runs one episode; returns dict with loss terms and (if keep) per-step outputs.
    use_imagine=False: a model built with imagine_enc_pano=False (the paper's no-imagination baseline, vilmodel_cmt.py:975-996,1099-1116):
    no `imagine` / alignment call, the `visual` calls get no imagination tokens (agent_cmt.py:419-462 guards them with args.imagine_enc_pano)."""
    ep = et.ep
    out = {"logits": [], "states": [], "hist": [], "txt_o": [], "ob_o": [], "hist_o": []}
    txt_embeds = model("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks)
    imagine_embeds = None
    if use_imagine:
        imagine_embeds = model("imagine", imagine_pano_img_feats=et.imagine_feats,
                               imagine_masks=None if bypass else et.imagine_masks)
    use_aux = use_aux and use_imagine
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
            ob_masks=s["ob_masks"], imagine_embeds=imagine_embeds, imagine_masks=et.imagine_masks if use_imagine else None)
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


def run_episode_time_batched(model, et, bypass=True, use_aux=True, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum, use_imagine=True):
    """The same teacher-forced episode with all T steps executed as ONE batch of T*B samples.

    Under teacher forcing every step's observation and history INPUTS are known up front (agent_cmt.py:561-562), so the
    T `history` calls collapse into one call on [T*B] rows and the T `visual` calls into one call whose sample (t, b)
    sees the history prefix [CLS, h_0 .. h_{t-1}] (padded to T entries, masked exactly like ended episodes are in the
    reference, model_HAMT.py:62-63). Logits, loss and gradients equal the step-by-step rollout; only the GEMM row
    count (x T) and the launch count (/ T) change. Sampling / RL rollouts cannot use this (actions feed back)."""
    ep, B, T = et.ep, et.B, et.T
    dev = et.txt_ids.device
    cat = et.full
    # the history encoder of all T steps (13.8 k-row launches) runs on a second stream beside the text encoder (5.1 k-row launches that
    # fill half of the chip); autograd then runs its backward on that stream too, beside the text encoder's
    side = None
    if dev.type == "cuda":
        for k in ("hist_img_feats", "hist_ang_feats", "hist_pano_img_feats", "hist_pano_ang_feats"):
            cat(k)
        main = torch.cuda.current_stream()
        side = getattr(et, "_side", None)
        if side is None:
            side = et._side = torch.cuda.Stream()
        side.wait_stream(main)
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        h_all = model("history", hist_img_feats=cat("hist_img_feats"), hist_ang_feats=cat("hist_ang_feats"),
                      ob_step_ids=torch.arange(T, device=dev).repeat_interleave(B),
                      hist_pano_img_feats=cat("hist_pano_img_feats"), hist_pano_ang_feats=cat("hist_pano_ang_feats"))
    txt = model("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks)
    img = model("imagine", imagine_pano_img_feats=et.imagine_feats, imagine_masks=None if bypass else et.imagine_masks) if use_imagine else None
    use_aux = use_aux and use_imagine
    im_masks = et.imagine_masks if use_imagine else None
    aux = None
    if use_aux:
        aux, img = model("align_with_contrastive_loss", align_txt_embeds=txt, txt_masks=et.txt_masks, align_imagine_embeds=img,
                         imagine_masks=et.imagine_masks, sub_instr_segs=ep.sub_instr_segs,
                         sub_instr_imag_flag=ep.sub_instr_imag_flag, noun_phrase_segs=ep.noun_phrase_segs)
    cls = model("history").expand(B, -1)                                                   # [B, H]
    if side is not None:
        main.wait_stream(side)
    H = h_all.shape[-1]
    hist_steps = h_all.view(T, B, H)
    # sample (t, b): [CLS, h_0 .. h_{T-2}] with the first t+1 entries valid
    prefix = torch.cat([cls.to(h_all.dtype).unsqueeze(0), hist_steps[:T - 1]], 0)          # [T, B, H] entries 0..T-1
    hist = prefix.permute(1, 0, 2).unsqueeze(0).expand(T, B, T, H).reshape(T * B, T, H)
    hist_masks = (torch.arange(T, device=dev)[None, :] <= torch.arange(T, device=dev)[:, None])    # [t, entry]
    hist_masks = hist_masks.unsqueeze(1).expand(T, B, T).reshape(T * B, T)
    rep1 = lambda x: x.unsqueeze(0).expand((T,) + tuple(x.shape)).reshape((T * x.shape[0],) + tuple(x.shape[1:]))
    rep = lambda x: [rep1(v) for v in x] if isinstance(x, list) else rep1(x)          # no_lang_ca: `language` returns the per-layer text states (:1022-1030)
    kw = {}
    if hasattr(model, "language_side") and not isinstance(txt, list):      # product model: the episode's language side built once (see TapedEpisode)
        kw["lang_side"] = model.language_side(txt, et.txt_masks, img, im_masks).repeat(T)
    logits, txt_o, hist_o, ob_o = model(
        "visual", txt_embeds=rep(txt), txt_masks=rep(et.txt_masks), hist_embeds=hist, hist_masks=hist_masks,
        ob_img_feats=cat("ob_img_feats"), ob_ang_feats=cat("ob_ang_feats"), ob_nav_types=cat("ob_nav_types"),
        ob_masks=cat("ob_masks"), imagine_embeds=rep(img) if use_imagine else None, imagine_masks=rep(im_masks) if use_imagine else None, **kw)
    ml_loss = criterion(logits, cat("target"))
    loss = ml_loss * train_ml / B
    if use_aux and torch.is_tensor(aux):
        loss = loss + cosine_weight * aux
    return {"loss": loss, "ml_loss": ml_loss, "aux": aux, "logits": list(logits.view(T, B, -1)), "txt_embeds": txt,
            "imagine_embeds": img, "hist": list(hist_steps)}


class TapedEpisode:
    """Step-by-step FORWARD - the call pattern a sampled rollout needs: step t + 1's observation may depend on the action chosen from step
    t's logits (r2r/agent_cmt.py:498-606) - and ONE episode-batched BACKWARD (vln_imagine_amd.ops.EpisodeTape): the T `visual` /
    `history` calls write their activations into slices of episode-wide buffers, a ghost pass of the same model code over the T x B
    samples records the autograd graph without launching a kernel, and loss.backward() then runs on T x longer launches.
    Valid because no transformer output of step t enters step t + 1's input: history tokens are re-encoded from features
    (vilmodel_cmt.py:576-618, 1056-1205). Every step sees the history padded to T entries ([CLS, h_0 .. h_{t-1}, ...] with the mask of
    model_HAMT.py:62-63); logits, loss and gradients equal run_episode's to rounding (tests/test_tape_gpu.py).

    Three phases, so that each can be its own captured hipGraph with host code (action choice, simulator) between them
    (train.GraphedStep(stages=...), hamt.buckets.SteppedEpisodeGraphs):
      begin()     language, imaginations, alignment head, history [CLS]
      step(t)     -> (logits [B, V], state [B, H] or None); reads step t's slices of et.full(k)
      finish()    ghost pass + loss -> the dict run_episode_taped returns

    lag_history=False (teacher forcing: the view taken at step t is known up front): step t's `history` call (the panorama encoder: 2.3 k-row
      launches that fill a fifth of the chip) runs on a second stream beside step t's `visual` call - both read features and h_0 .. h_{t-1}
      only - and its ghost pass is recorded on that stream, so autograd runs the history encoder's batched backward there too.
    lag_history=True (sampled rollouts: the history features of step t exist only after the action was chosen from step t's logits):
      `history` of step t - 1 opens step t, the last one opens finish(); step t reads the static mask et.hist_mask_T[t] (and finish() all of
      et.hist_mask_T), so the caller writes step t's history lengths with et.put_hist_lens(t, lens) - never et.hist_lens_dev alone - and step
      t's observation / step t - 1's history features any time before step t.
    feat_dropout: the wrapper's feature dropout (VLNBertCMT.drop_env, model_HAMT.py:20,38,44-47,63) on ob / hist / hist-pano image features.
      The drivers call NavCMT directly, so it is applied HERE, with the tape's counter-based masks: the step, the ghost pass and the batched
      backward see the same mask (an nn.Dropout inside a recorded step would draw another one in the ghost pass).
    ghost_compute=True (tests): the batched pass COMPUTES with the recorded dropout seeds instead of reusing the steps' buffers."""

    def __init__(self, model, et, tape=None, bypass=True, use_aux=True, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum,
                 ghost_compute=False, overlap_history=True, lag_history=False, want_states=False, feat_dropout=0.0, use_imagine=True):
        from vln_imagine_amd import ops
        self.use_imagine = use_imagine
        use_aux = use_aux and use_imagine
        self.model, self.et, self.B, self.T = model, et, et.B, et.T
        self.feat_dropout, self._ops = float(feat_dropout), ops
        self.tape = tape if tape is not None else ops.EpisodeTape(et.T)
        assert self.tape.T >= et.T
        self.bypass, self.use_aux, self.train_ml, self.cosine_weight, self.criterion = bypass, use_aux, train_ml, cosine_weight, criterion
        self.ghost_compute, self.lag, self.want_states = ghost_compute, lag_history, want_states
        self.overlap = overlap_history and not lag_history and _OVERLAP_HISTORY
        # teacher forcing, product model: step t's `history` call runs in lockstep with its `visual` call (3-problem GEMM launches, NavCMT
        # `visual` with hist_step=) instead of beside it on a second stream; the ghost pass and the backward keep the two separate calls
        self.lockstep = (_LOCKSTEP_HISTORY and overlap_history and not lag_history and hasattr(model, "language_side")
                         and getattr(getattr(model, "config", None), "hist_enc_pano", False) and not getattr(model.config, "no_lang_ca", True))
        self.step_logits = []

    def _drop(self, x):
        """drop_env on a feature tensor (inside a tape context: ops.dropout then uses the tape's seeds)."""
        return self._ops.dropout(x, self.feat_dropout, self.model.training) if self.feat_dropout > 0.0 else x

    def _history(self, t):
        f, B = self.et.full, self.B
        sl = slice(t * B, (t + 1) * B)
        with self.tape.record("history", t):
            return self.model("history", hist_img_feats=self._drop(f("hist_img_feats")[sl]), hist_ang_feats=f("hist_ang_feats")[sl],
                              ob_step_ids=self.et.step_ids[t], hist_pano_img_feats=self._drop(f("hist_pano_img_feats")[sl]),
                              hist_pano_ang_feats=f("hist_pano_ang_feats")[sl])

    def begin(self):
        model, et, tape, B, T = self.model, self.et, self.tape, self.B, self.T
        ep = et.ep
        dev = et.txt_ids.device
        tape.reset()
        self.step_logits = []
        self.main = torch.cuda.current_stream() if dev.type == "cuda" else None
        self.side = None
        if self.overlap and self.main is not None:
            self.side = getattr(tape, "_side", None)
            if self.side is None:
                self.side = tape._side = torch.cuda.Stream()
        for k in ("hist_img_feats", "hist_ang_feats", "hist_pano_img_feats", "hist_pano_ang_feats", "ob_img_feats", "ob_ang_feats",
                  "ob_nav_types", "ob_masks", "target"):
            et.full(k)                                         # built (once) on the main stream before any side-stream reader
        self.ar = torch.arange(T, device=dev)
        if not self.lag:
            self.valid = et.hist_mask_T                                                        # [step t, sample b, entry j]: entry j < history length before step t
            self.hm_full = self.valid.reshape(T * B, T).contiguous()
        # teacher forcing, product model: ALL T `history` calls of the episode (and the [CLS] one) on the second stream from here on, beside
        # the text encoder, whose 5.1 k-row launches fill half of the chip. They read features only (vilmodel_cmt.py:576-618); step t waits
        # for the event behind history t - 1. See _HISTORY_UPFRONT.
        self.upfront = (_HISTORY_UPFRONT and dev.type == "cuda" and not self.lag and not self.lockstep and hasattr(model, "language_side"))
        self.h_event = None
        if self.upfront and self.side is not None:
            self.side.wait_stream(self.main)                   # the fork point; the call itself is issued below, AFTER the text side
        self.txt = model("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks)
        imf = et.imagine_feats                                 # outside the tape (one call per episode): torch's own dropout
        if self.feat_dropout > 0.0:
            imf = F.dropout(imf, self.feat_dropout, model.training)
        img = model("imagine", imagine_pano_img_feats=imf, imagine_masks=None if self.bypass else et.imagine_masks) if self.use_imagine else None
        self.im_masks = et.imagine_masks if self.use_imagine else None
        self.aux = None
        if self.use_aux:
            self.aux, img = model("align_with_contrastive_loss", align_txt_embeds=self.txt, txt_masks=et.txt_masks, align_imagine_embeds=img,
                                  imagine_masks=et.imagine_masks, sub_instr_segs=ep.sub_instr_segs,
                                  sub_instr_imag_flag=ep.sub_instr_imag_flag, noun_phrase_segs=ep.noun_phrase_segs)
        self.img = img
        # the language stream of every `visual` call of the episode, incl. the first cross-modal layer's language Q / K / V: once, with autograd
        self.ls = None if isinstance(self.txt, list) else model.language_side(self.txt, et.txt_masks, img, self.im_masks)
        self.lock = self.lockstep and self.ls is not None
        if not self.upfront:
            self._cls_and_history_buffer(dev)
        else:
            with (torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()):
                self._cls_and_history_buffer(dev)
                f = et.full
                with tape.record_steps("history", T):          # the call the ghost pass repeats (finish())
                    h_all = model("history", hist_img_feats=self._drop(f("hist_img_feats")), hist_ang_feats=f("hist_ang_feats"),
                                  ob_step_ids=self.ar.repeat_interleave(B), hist_pano_img_feats=self._drop(f("hist_pano_img_feats")),
                                  hist_pano_ang_feats=f("hist_pano_ang_feats"))
                with torch.no_grad():                          # entry j + 1 of every later step's input = h_j (zero where the history is shorter)
                    prefix = torch.cat([self.cls.detach().to(h_all.dtype).unsqueeze(0), h_all.view(T, B, -1)[:T - 1]], 0)
                    torch.mul(prefix.permute(1, 0, 2).unsqueeze(0), self.valid.to(h_all.dtype).unsqueeze(-1), out=self.hb)
                if self.side is not None:
                    self.h_event = torch.cuda.Event()
                    self.h_event.record(self.side)
        # teacher forcing holds the observation features of all T steps: ONE cast to the compute type per episode instead of one per step
        # (sampled rollouts write step t + 1's observation after step t: they keep the per-step cast inside ops.linear)
        self.ob_feats = et.full("ob_img_feats")
        cdt = getattr(model, "compute_dtype", None)
        if not self.lag and dev.type == "cuda" and cdt in (torch.bfloat16, torch.float16) and self.ob_feats.dtype != cdt:
            self.ob_feats = self._ops.cast(self.ob_feats, cdt)
        # teacher forcing knows the masks of all T steps here: their additive / boolean forms once per episode, not once per step
        self.vm_full = self.nav0_full = None
        if not self.lag and dev.type == "cuda" and _EPISODE_MASKS:
            self.vm_full = self._ops.additive_mask(torch.cat([self.hm_full, et.full("ob_masks")], 1))      # [T B, T + V]
            self.nav0_full = et.full("ob_nav_types") == 0

    def _cls_and_history_buffer(self, dev):
        """The [CLS] history token and the history inputs of all steps: sample (t, b) holds [CLS, h_0 .. h_{t-1}, ...]. Entries beyond a
        sample's history length are masked keys: teacher forcing zeroes them (all lengths are known), a lagging history leaves later entries
        as they are (finite: the buffer is zeroed once per tape). The recorded steps read slices of it, the ghost pass an autograd expression
        with the same values."""
        model, tape, B, T = self.model, self.tape, self.B, self.T
        self.cls = cls = model("history").expand(B, -1)                                        # [B, H]
        H, dt = cls.shape[-1], cls.dtype
        hb = getattr(tape, "_hist_buf", None)
        if hb is None or hb.shape != (T, B, T, H) or hb.dtype != dt:
            hb = tape._hist_buf = torch.zeros((T, B, T, H), dtype=dt, device=dev)
        self.hb = hb
        with torch.no_grad():
            hb[:, :, 0] = cls
        self.validf = self.valid.to(dt) if (not self.lag and dev.type == "cuda" and _EPISODE_MASKS) else None

    def _put_history(self, t, h):
        """History token of step t into entry t + 1 of the later steps' inputs (teacher forcing: zero where the sample's history is shorter)."""
        if t + 1 < self.T:
            hb = self.hb
            with torch.no_grad():
                if self.validf is not None and h.dtype == hb.dtype:
                    torch.mul(h.unsqueeze(0), self.validf[t + 1:, :, t + 1, None], out=hb[t + 1:, :, t + 1])      # one launch
                else:
                    hb[t + 1:, :, t + 1] = h * self.valid[t + 1:, :, t + 1, None].to(h.dtype)

    def step(self, t):
        model, et, tape, B, T, hb = self.model, self.et, self.tape, self.B, self.T, self.hb
        f, main, side = et.full, self.main, self.side
        sl = slice(t * B, (t + 1) * B)
        if self.lag:
            if t > 0:
                h = self._history(t - 1)
                with torch.no_grad():
                    hb[t:, :, t] = h
            # step t's mask over the T padded entries: a static tensor the host wrote with the lengths where the episode lives in static buffers
            if _MASK_IN_GRAPH:                                # (anomaly hunt only, tools/stale_mask_repro.py)
                hm = (torch.arange(T, device=self.ar.device) if _MASK_IN_GRAPH == 2 else self.ar)[None, :] < et.hist_lens_dev[t][:, None]
            else:
                hm = et.hist_mask_T[t]
        else:
            hm = self.hm_full[sl]
            if self.upfront:
                if t == 0 and self.h_event is not None:
                    main.wait_event(self.h_event)          # the history buffer of the whole episode is filled
            elif side is not None and not self.lock:
                side.wait_stream(main)                     # the fork point: step t's history needs nothing of step t's `visual` call
                if not _HISTORY_AFTER:
                    with torch.cuda.stream(side):
                        h = self._history(t)
        lock = self.lock
        with tape.record(("visual", "history") if lock else "visual", t):
            hist_step = None
            if lock:
                with tape.use("history"):                  # the arguments of _history(t), drawn under its key in its order
                    hist_step = ("visual", "history", dict(
                        hist_img_feats=self._drop(f("hist_img_feats")[sl]), hist_ang_feats=f("hist_ang_feats")[sl], ob_step_ids=et.step_ids[t],
                        hist_pano_img_feats=self._drop(f("hist_pano_img_feats")[sl]), hist_pano_ang_feats=f("hist_pano_ang_feats")[sl]), side)
            res = model(
                "visual", txt_embeds=self.txt, txt_masks=et.txt_masks, hist_embeds=hb[t], hist_masks=hm,
                ob_img_feats=self._drop(self.ob_feats[sl]), ob_ang_feats=f("ob_ang_feats")[sl], ob_nav_types=f("ob_nav_types")[sl],
                ob_masks=f("ob_masks")[sl], imagine_embeds=self.img, imagine_masks=self.im_masks, lang_side=self.ls,
                vis_mask_add=self.vm_full[sl] if self.vm_full is not None else None,
                ob_is_nav0=self.nav0_full[sl] if self.nav0_full is not None else None, **({"hist_step": hist_step} if lock else {}))
        lg, txt_o, hist_o, ob_o = res[:4]
        if _HISTORY_AFTER and not self.lag and not lock and side is not None and not self.upfront:
            with torch.cuda.stream(side):                  # issued AFTER the `visual` call, forked where it was before (see _HISTORY_AFTER)
                h = self._history(t)
        self.step_logits.append(lg)
        state = txt_o[:, 0] * hist_o[:, 0] if self.want_states else None                      # model_HAMT.py:86
        if not self.lag and not self.upfront:
            if lock:
                h = res[4]
            elif side is not None:
                main.wait_stream(side)
            else:
                h = self._history(t)
            self._put_history(t, h)
        return lg, state

    def finish(self):
        """The ghost pass: the same two calls on the T x B samples; no kernels, only the autograd graph over the filled buffers."""
        model, et, tape, B, T = self.model, self.et, self.tape, self.B, self.T
        f, main, side, ar = et.full, self.main, self.side, self.ar
        if self.lag:
            self._history(T - 1)                     # consumed by no step; run so that the batched backward reads defined activations
            valid = et.hist_mask_T if not _MASK_IN_GRAPH else ar[None, None, :] < et.hist_lens_dev[:T][:, :, None]
            hm_full = valid.reshape(T * B, T).contiguous()
        else:
            valid, hm_full = self.valid, self.hm_full
        if side is not None:
            side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            with tape.ghost("history", compute=self.ghost_compute):
                h_all = model("history", hist_img_feats=self._drop(f("hist_img_feats")), hist_ang_feats=f("hist_ang_feats"),
                              ob_step_ids=ar.repeat_interleave(B), hist_pano_img_feats=self._drop(f("hist_pano_img_feats")),
                              hist_pano_ang_feats=f("hist_pano_ang_feats"))
        if side is not None:
            main.wait_stream(side)
        H = h_all.shape[-1]
        prefix = torch.cat([self.cls.to(h_all.dtype).unsqueeze(0), h_all.view(T, B, H)[:T - 1]], 0)    # entries 0 .. T-1 as [entry, B, H]
        if self.lag:                                                  # entry j of sample (t, b): written for every t >= j, never masked to zero
            hist = prefix.permute(1, 0, 2).unsqueeze(0).expand(T, B, T, H)
        else:
            hist = prefix.permute(1, 0, 2).unsqueeze(0) * valid.to(h_all.dtype)[:, :, :, None]        # zeros beyond the valid entries
        rep1 = lambda x: x.unsqueeze(0).expand((T,) + tuple(x.shape)).reshape((T * x.shape[0],) + tuple(x.shape[1:]))
        rep = lambda x: [rep1(v) for v in x] if isinstance(x, list) else rep1(x)      # no_lang_ca: the per-layer text states (vilmodel_cmt.py:1022-1030)
        if self.ls is not None and getattr(getattr(model, "config", None), "concat_imagine_with", None) == "language":
            rep = lambda x: x          # the language side is handed over ready-made (lang_side=): `visual` only checks these for presence
        with tape.ghost("visual", compute=self.ghost_compute):
            logits, txt_o, hist_o, ob_o = model(
                "visual", txt_embeds=rep(self.txt), txt_masks=rep(et.txt_masks), hist_embeds=hist.reshape(T * B, T, H), hist_masks=hm_full,
                ob_img_feats=self._drop(self.ob_feats), ob_ang_feats=f("ob_ang_feats"), ob_nav_types=f("ob_nav_types"),
                ob_masks=f("ob_masks"), imagine_embeds=rep(self.img) if self.use_imagine else None,
                imagine_masks=rep(self.im_masks) if self.use_imagine else None,
                lang_side=self.ls.repeat(T) if self.ls is not None else None,
                vis_mask_add=self.vm_full if not self.lag else None, ob_is_nav0=self.nav0_full if not self.lag else None)
        ml_loss = self.criterion(logits, f("target"))
        loss = ml_loss * self.train_ml / B
        if self.use_aux and torch.is_tensor(self.aux):
            loss = loss + self.cosine_weight * self.aux
        return {"loss": loss, "ml_loss": ml_loss, "aux": self.aux, "logits": list(logits.view(T, B, -1)), "step_logits": self.step_logits,
                "txt_embeds": self.txt, "imagine_embeds": self.img, "hist": list(h_all.view(T, B, H)), "tape": tape}


def run_episode_taped(model, et, tape=None, bypass=True, use_aux=True, train_ml=0.2, cosine_weight=0.5, criterion=ce_sum, on_step=None,
                      ghost_compute=False, overlap_history=True, lag_history=False, feat_dropout=0.0, use_imagine=True):
    """One episode through TapedEpisode: begin, T steps (`on_step(t, logits, state)` may choose the action), finish."""
    te = TapedEpisode(model, et, tape, bypass, use_aux, train_ml, cosine_weight, criterion, ghost_compute, overlap_history, lag_history,
                      want_states=on_step is not None, feat_dropout=feat_dropout, use_imagine=use_imagine)
    te.begin()
    for t in range(et.T):
        lg, state = te.step(t)
        if on_step is not None:
            on_step(t, lg, state)
    return te.finish()
