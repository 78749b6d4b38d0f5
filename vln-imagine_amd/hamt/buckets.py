"""Shape-bucketed hipGraph cache for real (ragged) HAMT batches.

In the reference the text length, the number of candidate views, the imagination count and the episode length change from batch to
batch (VLN-HAMT/finetune_src/r2r/agent_cmt.py:130-176 pads a batch to ITS longest instruction / largest candidate set,
:498-606 runs until the longest episode ends). A captured step has fixed shapes and fixed addresses, so batches are padded up to
the next BUCKET (L, V) - with exactly the padding the reference itself uses inside a batch: text pads are id 0 / mask False,
observation pads are zero features / mask False / nav type 0 (their logits are -inf, agent_cmt.py:558) - into one set of static
device buffers per bucket, and each (L, V, T) bucket owns one captured training step (train.GraphedStep). The imagination-grounding
head's index lists (agent_cmt.py:436-459 -> models/vilmodel_cmt.py:755-785) become fixed-capacity device buffers as well
(AlignWithContrastiveLoss.set_static_plan), refilled per batch.

Padded key positions carry the additive mask -10000: their softmax weight is exp(-10000 - max) = 0 in float32, so every real row
of every activation - and every logit - is bit-identical to the unpadded eager run (tests/test_buckets_gpu.py); weight gradients
differ by summation order only (more zero rows)."""
import os

import numpy as np
import torch

from vln_imagine_amd import ops


class EpisodeBuffers:
    """Static device buffers of one bucket; quacks like hamt.episode.EpisodeTensors. `load(ep)` pads a synth.HamtEpisode-shaped
    numpy episode (B and I as the bucket, L <= bucket L, V <= bucket V, T == bucket T) into them."""

    def __init__(self, B, L, V, I, T, device, feat=768, ang=4, pano=36):
        dev = torch.device(device)
        self.B, self.L, self.V, self.I, self.T, self.device = B, L, V, I, T, dev
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
        self.txt_ids, self.txt_masks = z(B, L, dt=torch.int64), z(B, L, dt=torch.bool)
        self.imagine_feats, self.imagine_masks = z(B, I, feat), z(B, I, dt=torch.bool)
        # step inputs live in ONE [T * B, ...] buffer per key (step t = rows [t B, (t + 1) B)): run_episode reads the per-step views,
        # the episode tape (hamt.episode.TapedEpisode) the whole buffers through full(k)
        self._full = dict(ob_img_feats=z(T * B, V, feat), ob_ang_feats=z(T * B, V, ang), ob_nav_types=z(T * B, V, dt=torch.int64),
                          ob_masks=z(T * B, V, dt=torch.bool), target=z(T * B, dt=torch.int64), hist_img_feats=z(T * B, feat),
                          hist_ang_feats=z(T * B, ang), hist_pano_img_feats=z(T * B, pano, feat), hist_pano_ang_feats=z(T * B, pano, ang))
        self.steps = [{k: v[t * B:(t + 1) * B] for k, v in self._full.items()} for t in range(T)]
        self.step_ids = [torch.tensor([i], device=dev) for i in range(T)]
        self.hist_masks = [torch.ones((B, t + 1), dtype=torch.bool, device=dev) for t in range(T)]
        self.hist_lens_dev = (torch.arange(T, device=dev) + 1)[:, None].expand(T, B).contiguous()     # [T, B], model_HAMT.py:62-63
        self.hist_mask_T = torch.arange(T, device=dev)[None, None, :] < self.hist_lens_dev[:, :, None]  # [T, B, T]: step t's mask over T padded entries
        cap = B * I
        self.plan = dict(rows=z(cap, dt=torch.int64), scored=z(cap, dt=torch.int64), seg_off=z(cap + 1, dt=torch.int32),
                         tok_rows=z(B * L, dt=torch.int32), weight=z(cap), count=torch.ones((), dtype=torch.float32, device=dev),
                         target=torch.full((cap,), cap, dtype=torch.int64, device=dev))
        self.ep = self                       # run_episode reads the annotation lists from `.ep`; the static plan replaces them
        self.sub_instr_segs = self.sub_instr_imag_flag = self.noun_phrase_segs = None

    def full(self, k):
        return self._full[k]

    @staticmethod
    def _put(dst, src):
        """dst[:src.shape] = src, the rest zero / False (one H2D copy of a host-padded array)."""
        host = np.zeros(tuple(dst.shape), dtype=src.dtype)
        host[tuple(slice(0, n) for n in src.shape)] = src
        dst.copy_(torch.from_numpy(host), non_blocking=False)

    OBS_KEYS = ("ob_img_feats", "ob_ang_feats", "ob_nav_types", "ob_masks")
    HIST_KEYS = ("hist_img_feats", "hist_ang_feats", "hist_pano_img_feats", "hist_pano_ang_feats")

    def put_step(self, t, src, keys=None):
        """Writes step t's inputs `keys` (default: all of observation, history features, target) from numpy arrays, padded to the bucket."""
        for k in keys or (self.OBS_KEYS + self.HIST_KEYS + ("target",)):
            buf = self.steps[t][k]
            if k == "target":
                buf.copy_(torch.from_numpy(np.asarray(src[k])))
            else:
                self._put(buf, np.asarray(src[k], dtype=np.float32) if buf.dtype == torch.float32 else src[k])

    def put_hist_lens(self, t, lens):
        """History length of every sample before step t (model_HAMT.py:62-63); step t's graph reads it."""
        lens = np.asarray(lens, dtype=np.int64)                 # (numpy on the host: torch CPU ops on 8-element tensors cost more than the copies)
        self.hist_lens_dev[t].copy_(torch.from_numpy(lens))
        self.hist_masks[t].copy_(torch.from_numpy(np.arange(t + 1)[None, :] < lens[:, None]))
        self.hist_mask_T[t].copy_(torch.from_numpy(np.arange(self.T)[None, :] < lens[:, None]))

    def load(self, ep, steps=True):
        """Instruction side (text, imaginations, alignment plan) and - unless steps=False: a rollout fills them as it goes - all T steps."""
        assert ep.B == self.B and ep.I == self.I and ep.T == self.T and ep.L <= self.L and ep.V <= self.V, "episode does not fit the bucket"
        self._put(self.txt_ids, ep.txt_ids)
        self._put(self.txt_masks, ep.txt_masks)
        self._put(self.imagine_feats, ep.imagine_feats.astype(np.float32))
        self._put(self.imagine_masks, ep.imagine_masks)
        if steps:
            for t, src in enumerate(ep.steps):
                self.put_hist_lens(t, ep.hist_lens[t])
                self.put_step(t, src)
        self._load_plan(ep)
        return self

    def _load_plan(self, ep):
        """The alignment head's index lists (the reference's triple loop, vilmodel_cmt.py:755-785, with its assertions) into self.plan;
        `self`: anything with B, L, I and the plan buffers (duet.buckets.DuetEpisodeBuffers shares this)."""
        B, L, I, cap = self.B, self.L, self.I, self.B * self.I
        rows, scored, seg_off, tok = [], [], [0], []
        for b in range(B):
            flags = [x == "True" for x in ep.sub_instr_imag_flag[b]]
            assert len(flags) == len(ep.sub_instr_segs[b]) == len(ep.noun_phrase_segs[b])
            for i, f in enumerate(flags):
                if not f:
                    continue
                assert bool(ep.imagine_masks[b, i]), "Imagine embeds is not valid where embedding addition is being applied."
                s0, s1 = ep.sub_instr_segs[b][i]
                rows.append(b * I + i)
                for (a, z) in ep.noun_phrase_segs[b][i]:
                    assert a >= s0 and z <= s1, "check np indices and sub-instr indices. They seem off."
                    assert bool(ep.txt_masks[b, a:z + 1].all()), "Text_embeds is not valid where embedding addition is being applied."
                    tok.extend(range(b * L + a, b * L + z + 1))          # row index in the PADDED [B, L] layout
                if ep.noun_phrase_segs[b][i]:
                    scored.append(len(rows) - 1)
                    seg_off.append(len(tok))
        n = len(scored)
        pad = lambda v, size, fill: np.asarray(list(v) + [fill] * (size - len(v)))
        p = self.plan
        p["rows"].copy_(torch.from_numpy(pad(rows, cap, 0).astype(np.int64)))
        p["scored"].copy_(torch.from_numpy(pad(scored, cap, 0).astype(np.int64)))
        p["seg_off"].copy_(torch.from_numpy(pad(seg_off, cap + 1, seg_off[-1]).astype(np.int32)))      # empty segments behind the real ones
        p["tok_rows"].copy_(torch.from_numpy(pad(tok, B * L, 0).astype(np.int32)))
        p["weight"].copy_(torch.from_numpy(pad([1.0] * n, cap, 0.0).astype(np.float32)))
        p["target"].copy_(torch.from_numpy(pad([rows[j] for j in scored], cap, cap).astype(np.int64)))  # pads -> the scratch row
        p["count"].fill_(float(max(n, 1)))


class HamtGraphBuckets:
    """One captured training step per (L, V, T) bucket. step(ep): pad the batch into its bucket's buffers and replay that bucket's
    graphs; the first batch of a bucket runs eagerly (lazy initialisation, GEMM autotune at the bucket's row counts) and is captured
    right after. Returns the loss tensor and the per-step logits (static outputs of the bucket's graph)."""

    def __init__(self, trainer, model, B, I, l_buckets=(48, 64, 80), v_buckets=(25, 31, 37), device="cuda"):
        self.trainer, self.model, self.B, self.I, self.device = trainer, model, B, I, device
        self.l_buckets, self.v_buckets = sorted(l_buckets), sorted(v_buckets)
        self.buckets = {}                # (L, V, T) -> [buffers, captured step or None, outputs dict]
        self.head = getattr(model, "contrastive_alignment_model", None)

    def key_for(self, ep):
        L = next((x for x in self.l_buckets if x >= ep.L), None)
        V = next((x for x in self.v_buckets if x >= ep.V), None)
        if L is None or V is None:
            raise ValueError(f"episode with {ep.L} text tokens / {ep.V} observation tokens exceeds the largest bucket "
                             f"(l_buckets {self.l_buckets}, v_buckets {self.v_buckets})")
        return (L, V, ep.T)

    def _fwd_bwd(self, key):
        from vln_imagine_amd.hamt.episode import run_episode
        bufs, _, outs = self.buckets[key]

        def fwd_bwd():
            if self.head is not None:
                self.head.set_static_plan(bufs.plan)
            try:
                out = run_episode(self.model, bufs, criterion=ops.cross_entropy_sum, keep=True)
            finally:
                if self.head is not None:
                    self.head.set_static_plan(None)
            if self.model.compute_dtype == torch.float16:          # the fused step divides the trainer's loss scale out again
                (out["loss"] * self.trainer.loss_scale).backward()
            else:
                out["loss"].backward()
            outs["logits"] = [t.detach() for t in out["logits"]]
            return out["loss"].detach()
        return fwd_bwd

    def step(self, ep):
        key = self.key_for(ep)
        ent = self.buckets.get(key)
        if ent is None:
            ent = self.buckets[key] = [EpisodeBuffers(self.B, key[0], key[1], self.I, key[2], self.device), None, {}]
        bufs = ent[0].load(ep)
        fwd_bwd = self._fwd_bwd(key)
        if ent[1] is None:                       # first batch of this bucket: a real eager step, then the capture (which runs nothing)
            self.trainer.zero_grad()
            loss = fwd_bwd()
            self.trainer.allreduce_grads()
            self.trainer.step()
            logits = [t.clone() for t in ent[2]["logits"]]
            loss = loss.clone()
            ent[2].clear()
            ent[1] = self.trainer.capture(fwd_bwd, warmup=0)
            return loss, logits
        loss = ent[1]()
        return loss, ent[2]["logits"]


class SteppedEpisodeGraphs:
    """Graph replay for rollouts whose next observation depends on the action (sampling / RL, r2r/agent_cmt.py:498-606, :814-827): the
    episode is captured as T + 2 graphs - begin | step 0 | ... | step T-1 | ghost pass + backward + optimizer - over one set of static
    buffers (`bufs`, an EpisodeBuffers); between two replays the caller reads step t's logits (`logits(t)`, a static tensor), picks the
    action, asks its simulator, and writes step t + 1's observation and step t's history features into `bufs` (`put_step`). The history
    call lags by one step (TapedEpisode(lag_history=True)). The backward is still ONE episode-batched pass.

        g = SteppedEpisodeGraphs(trainer, model, bufs)           # one eager warm-up episode on the data in `bufs`, then the capture
        g.begin()                                               # after bufs.load(ep, steps=False) (or load(ep))
        for t in range(T):
            g.step(t); a = g.logits(t).argmax(1)                # host decides; then bufs.put_hist_lens(t + 1, ...), bufs.put_step(t + 1, obs, keys=OBS_KEYS)
                                                                # and bufs.put_step(t, hist, keys=HIST_KEYS) (the view taken at step t)
        loss = g.finish()                                       # targets written into bufs.full('target') before this
    """

    def __init__(self, trainer, model, bufs, tape=None, want_states=False, **episode_kw):
        from vln_imagine_amd.hamt.episode import TapedEpisode
        self.trainer, self.model, self.bufs = trainer, model, bufs
        self.head = getattr(model, "contrastive_alignment_model", None)
        self.ep = TapedEpisode(model, bufs, tape if tape is not None else ops.EpisodeTape(bufs.T), criterion=ops.cross_entropy_sum,
                               lag_history=True, want_states=want_states, **episode_kw)
        self._out, self._steps = {}, {}

        def begin():
            if self.head is not None:
                self.head.set_static_plan(bufs.plan)
            try:
                self.ep.begin()
            finally:
                if self.head is not None:
                    self.head.set_static_plan(None)

        def step(t):
            def run():
                lg, st = self.ep.step(t)
                self._steps[t] = (lg.detach(), st.detach() if st is not None else None)
            return run

        def finish():
            out = self.ep.finish()
            loss = out["loss"]
            if model.compute_dtype == torch.float16:               # the fused step divides the trainer's loss scale out again
                (loss * trainer.loss_scale).backward()
            else:
                loss.backward()
            self._out["logits"] = [t.detach() for t in out["logits"]]
            return loss.detach()

        self.graphs = trainer.capture(finish, warmup=1, stages=[begin] + [step(t) for t in range(bufs.T)])

    def begin(self):
        self.graphs.stage(0)

    def step(self, t):
        self.graphs.stage(1 + t)

    def logits(self, t):
        return self._steps[t][0]

    def state(self, t):
        return self._steps[t][1]

    def finish(self):
        return self.graphs.finish()


class SteppedInferenceGraphs:
    """Forward-only rollouts (validation: Seq2SeqCMTAgent.test -> rollout(train_ml=None, feedback='argmax'), r2r/agent_cmt.py:700-760) from captured
    graphs: begin (language, imaginations, history [CLS]) | step t (history of step t - 1, then visual) under no_grad, no tape, no trainer - nothing
    is kept for a backward. Same static buffers and the same host protocol as SteppedEpisodeGraphs; `logits(t)` is a static tensor.

        g = SteppedInferenceGraphs(model, bufs)                 # one eager warm-up on the data in `bufs`, then the capture
        g.begin(); [g.step(t), g.logits(t) ...]
    """

    def __init__(self, model, bufs, bypass=True, want_states=False, mask_in_graph=False, debug=None):
        self.model, self.bufs, self.T = model, bufs, bufs.T
        self._logits, self._states = {}, {}
        self._ar = torch.arange(bufs.T, device=bufs.device)
        mask_in_graph = mask_in_graph or int(os.environ.get("VLNI_MASK_IN_GRAPH", "0"))
        B, T, dev = bufs.B, bufs.T, bufs.device
        st = {}

        def begin():
            if hasattr(model, "_lang_side"):
                model._lang_side = None
            st["txt"] = model("language", txt_ids=bufs.txt_ids, txt_masks=bufs.txt_masks)
            st["img"] = model("imagine", imagine_pano_img_feats=bufs.imagine_feats, imagine_masks=None if bypass else bufs.imagine_masks)
            cls = model("history").expand(B, -1)
            if "hb" not in st:
                st["hb"] = torch.zeros((B, T, cls.shape[-1]), dtype=cls.dtype, device=dev)       # [CLS, h_0 .. h_{T-2}]; later entries masked
            st["hb"][:, 0] = cls

        def step(t):
            def run():
                s = bufs.steps
                if t > 0:
                    st["hb"][:, t] = model("history", hist_img_feats=s[t - 1]["hist_img_feats"], hist_ang_feats=s[t - 1]["hist_ang_feats"],
                                           ob_step_ids=bufs.step_ids[t - 1], hist_pano_img_feats=s[t - 1]["hist_pano_img_feats"],
                                           hist_pano_ang_feats=s[t - 1]["hist_pano_ang_feats"])
                if mask_in_graph:                                   # the mask from the lengths INSIDE the step's graph (model_HAMT.py:63)
                    ar = torch.arange(T, device=dev) if mask_in_graph == 2 else self._ar
                    hm = ar[None, :] < bufs.hist_lens_dev[t][:, None]
                else:
                    hm = bufs.hist_mask_T[t]                        # [B, T] static, written by the host with the lengths (put_hist_lens)
                if debug is not None:
                    debug[("hm", t)] = hm
                lg, txt_o, hist_o, _ = model("visual", txt_embeds=st["txt"], txt_masks=bufs.txt_masks, hist_embeds=st["hb"], hist_masks=hm,
                                             ob_img_feats=s[t]["ob_img_feats"], ob_ang_feats=s[t]["ob_ang_feats"], ob_nav_types=s[t]["ob_nav_types"],
                                             ob_masks=s[t]["ob_masks"], imagine_embeds=st["img"], imagine_masks=bufs.imagine_masks)
                self._logits[t] = lg
                if want_states:
                    self._states[t] = txt_o[:, 0] * hist_o[:, 0]
            return run

        stages = [begin] + [step(t) for t in range(T)]
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                         # eager warm-up (lazy initialisation, GEMM autotune at these row counts)
                for f in stages:
                    f()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graphs, pool = [], None
            for f in stages:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **(dict(pool=pool) if pool is not None else {})):
                    f()
                pool = g.pool()
                self.graphs.append(g)

    def begin(self):
        self.graphs[0].replay()

    def step(self, t):
        self.graphs[1 + t].replay()

    def logits(self, t):
        return self._logits[t]

    def state(self, t):
        return self._states[t]
