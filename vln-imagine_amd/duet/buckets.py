"""Static episode buffers and captured-graph runners for DUET (the DUET half of hamt/buckets.py).

In the reference every step of every batch has its own shapes: the instruction length, the number of map nodes (it grows with the
exploration, VLN-DUET/map_nav_src/r2r/agent.py:98-134 pads a batch to ITS largest map), the panorama width. A captured step has fixed
shapes and fixed addresses, so an episode is padded up to a BUCKET (L text tokens, Gmax map nodes, T steps) - with the padding the
reference itself uses inside a batch: text pads are id 0 / mask False, padded map nodes are masked out (their logits are -inf), padded
views have mask False / nav type 0 - into one set of device buffers per bucket:

  DuetEpisodeBuffers      the buffers; quacks like duet.episode.DuetEpisodeTensors for run_episode_taped / TapedEpisode. Everything the
                          reference derives from python lists per step - which bank row is which map node (agent.py:468-479), the
                          global / local fusion plan (models/vilmodel.py:1198-1217), the alignment head's index lists - is a device
                          tensor here, written by the host (`load`, `put_step`), so a replayed graph reads the current episode's.
  DuetGraphBuckets        teacher forcing: one captured training step per (L, Gmax, T) bucket, like hamt.buckets.HamtGraphBuckets
  SteppedEpisodeGraphs    sampled rollouts: begin | T step graphs | ghost pass + backward + optimizer, the host between the replays
"""
import numpy as np
import torch

from vln_imagine_amd import ops
from vln_imagine_amd.hamt.buckets import EpisodeBuffers as _HamtBuffers


class DuetEpisodeBuffers:
    PANO_KEYS = ("view_img_fts", "loc_fts", "nav_types", "view_lens")
    MAP_KEYS = ("gmap_step_ids", "gmap_pos_fts", "gmap_masks", "gmap_pair_dists", "gmap_visited_masks", "vp_pos_fts")

    def __init__(self, B, L, I, T, Gmax, device, P=36, feat=768):
        dev = torch.device(device)
        self.B, self.L, self.I, self.T, self.Gmax, self.P, self.device = B, L, I, T, Gmax, P, dev
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
        self.txt_ids, self.txt_masks = z(B, L, dt=torch.int64), z(B, L, dt=torch.bool)
        self.imagine_feats, self.imagine_masks = z(B, I, feat), z(B, I, dt=torch.bool)
        N = T * B
        self.ZERO = N * (P + 1)                                # the bank's zero row (duet.episode._taped_inputs)
        full = dict(view_img_fts=z(N, P, feat), loc_fts=z(N, P, 7), nav_types=z(N, P, dt=torch.int64), view_lens=torch.ones(N, dtype=torch.int64, device=dev),
                    gmap_step_ids=z(N, Gmax, dt=torch.int64), gmap_pos_fts=z(N, Gmax, 7), gmap_masks=z(N, Gmax, dt=torch.bool),
                    gmap_pair_dists=z(N, Gmax, Gmax), gmap_visited_masks=z(N, Gmax, dt=torch.bool), vp_pos_fts=z(N, P + 1, 14),
                    vp_masks=z(N, P + 1, dt=torch.bool), vp_nav_masks=z(N, P + 1, dt=torch.bool), pano_masks=z(N, P, dt=torch.bool), target=torch.full((N,), -100, dtype=torch.int64, device=dev))
        full["gmap_masks"][:, 0] = True                        # a map always holds its [STOP] node: no all-masked softmax row before the host wrote a step
        full["vp_masks"][:, 0] = True
        full["pano_masks"][:, 0] = True
        self.src, self.bw = z(N, Gmax, dt=torch.int32) - 1, z(N, P + 1, dt=torch.uint8)
        self.idx = torch.full((T, B, Gmax), self.ZERO, dtype=torch.int64, device=dev)
        full["gmap_vpids"] = full["vp_cand_vpids"] = None      # the fusion plan is given as tensors (fuse_plan)
        full["fuse_plan"] = (self.src, self.bw)
        self.full = full
        self.steps = []
        for t in range(T):
            sl = slice(t * B, (t + 1) * B)
            st = {k: (v[sl] if torch.is_tensor(v) else v) for k, v in full.items() if k != "fuse_plan"}
            st["fuse_plan"] = (self.src[sl], self.bw[sl])
            self.steps.append(st)
        self._taped = (self.steps, self.full, self.idx, Gmax, P, self.ZERO)
        cap = B * I
        self.plan = dict(rows=z(cap, dt=torch.int64), scored=z(cap, dt=torch.int64), seg_off=z(cap + 1, dt=torch.int32),
                         tok_rows=z(B * L, dt=torch.int32), weight=z(cap), count=torch.ones((), dtype=torch.float32, device=dev),
                         target=torch.full((cap,), cap, dtype=torch.int64, device=dev))
        self.ep = self
        self.sub_instr_segs = self.sub_instr_imag_flag = self.noun_phrase_segs = None

    _put = staticmethod(_HamtBuffers._put)

    def put_step(self, t, s):
        """Step t of a synth.DuetEpisode-shaped numpy step dict (what the agent's _panorama_feature_variable / _nav_gmap_variable /
        _nav_vp_variable produce, agent.py:67-207), padded into the bucket; the python-list parts become the index tensors."""
        from vln_imagine_amd.duet.models.vilmodel import GlocalTextPathNavCMT
        B, P, Gmax, st = self.B, self.P, self.Gmax, self.steps[t]
        G = s["gmap_masks"].shape[1]
        assert G <= Gmax and s["view_img_fts"].shape[1] == P and "obj_img_fts" not in s, "step does not fit the bucket"
        for k in self.PANO_KEYS + self.MAP_KEYS + ("target",):
            self._put(st[k], np.asarray(s[k], dtype=np.float32) if st[k].dtype == torch.float32 else np.asarray(s[k]))
        lens = np.asarray(s["view_lens"])
        self._put(st["vp_masks"], np.arange(P + 1)[None, :] < (lens + 1)[:, None])
        self._put(st["pano_masks"], np.arange(P)[None, :] < lens[:, None])
        self._put(st["vp_nav_masks"], np.concatenate([np.ones((B, 1), bool), np.asarray(s["nav_types"]) == 1], 1))
        off = np.full((B, Gmax), self.ZERO, np.int64)
        for b, srcs in enumerate(s["node_src"]):
            for j, src in enumerate(srcs):
                off[b, j + 1] = (src[1] * B + b) * (P + 1) + (0 if src[0] == "avg" else 1 + src[2])
        self.idx[t].copy_(torch.from_numpy(off))
        vpids = [list(v) + [None] * (Gmax - len(v)) for v in s["gmap_vpids"]]
        vis = np.zeros((B, Gmax), bool)
        vis[:, :G] = s["gmap_visited_masks"]
        src, bw = GlocalTextPathNavCMT.fuse_plan(vpids, vis.tolist(), s["vp_cand_vpids"], Gmax, P + 1)
        st["fuse_plan"][0].copy_(torch.tensor(src, dtype=torch.int32))
        st["fuse_plan"][1].copy_(torch.tensor(bw, dtype=torch.uint8))

    def load(self, ep, steps=True):
        """Instruction side (text, imaginations, alignment plan) and - unless steps=False: a rollout fills them as it goes - all T steps."""
        assert ep.B == self.B and ep.I == self.I and ep.T == self.T and ep.L <= self.L, "episode does not fit the bucket"
        self._put(self.txt_ids, ep.txt_ids)
        self._put(self.txt_masks, ep.txt_masks)
        self._put(self.imagine_feats, ep.imagine_feats.astype(np.float32))
        self._put(self.imagine_masks, ep.imagine_masks)
        if steps:
            for t, s in enumerate(ep.steps):
                self.put_step(t, s)
        _HamtBuffers._load_plan(self, ep)
        return self


def _runner(model, bufs, trainer):
    head = getattr(model, "contrastive_alignment_model", None)

    class _Plan:
        def __enter__(self):
            if head is not None:
                head.set_static_plan(bufs.plan)

        def __exit__(self, *exc):
            if head is not None:
                head.set_static_plan(None)
            return False

    def backward(loss):
        if model.compute_dtype == torch.float16:               # the fused step divides the trainer's loss scale out again
            (loss * trainer.loss_scale).backward()
        else:
            loss.backward()
        return loss.detach()
    return _Plan, backward


class DuetGraphBuckets:
    """One captured training step (step-by-step forward on the episode tape, one batched backward, optimizer) per (L, Gmax, T) bucket.
    step(ep): pad the episode into its bucket's buffers and replay that bucket's graphs; the first episode of a bucket runs eagerly (lazy
    initialisation, GEMM autotune at the bucket's row counts) and is captured right after. Returns (loss, per-step fused logits)."""

    def __init__(self, trainer, model, B, I, l_buckets=(48, 64, 80), g_buckets=(12, 20, 28), device="cuda"):
        self.trainer, self.model, self.B, self.I, self.device = trainer, model, B, I, device
        self.l_buckets, self.g_buckets = sorted(l_buckets), sorted(g_buckets)
        self.buckets = {}

    def key_for(self, ep):
        L = next((x for x in self.l_buckets if x >= ep.L), None)
        g = max(s["gmap_masks"].shape[1] for s in ep.steps)
        G = next((x for x in self.g_buckets if x >= g), None)
        if L is None or G is None:
            raise ValueError(f"episode with {ep.L} text tokens / {g} map nodes exceeds the largest bucket "
                             f"(l_buckets {self.l_buckets}, g_buckets {self.g_buckets})")
        return (L, G, ep.T)

    def _fwd_bwd(self, ent):
        from vln_imagine_amd.duet.episode import run_episode_taped
        bufs, outs = ent[0], ent[2]
        plan, backward = _runner(self.model, bufs, self.trainer)
        if ent[3] is None:
            ent[3] = ops.EpisodeTape(bufs.T)

        def fwd_bwd():
            with plan():
                out = run_episode_taped(self.model, bufs, tape=ent[3], criterion=ops.cross_entropy_sum)
            outs["fused"] = [t.detach() for t in out["fused"]]
            return backward(out["loss"])
        return fwd_bwd

    def step(self, ep):
        key = self.key_for(ep)
        ent = self.buckets.get(key)
        if ent is None:
            ent = self.buckets[key] = [DuetEpisodeBuffers(self.B, key[0], self.I, key[2], key[1], self.device), None, {}, None]
        ent[0].load(ep)
        fwd_bwd = self._fwd_bwd(ent)
        if ent[1] is None:
            self.trainer.zero_grad()
            loss = fwd_bwd()
            self.trainer.allreduce_grads()
            self.trainer.step()
            logits = [t.clone() for t in ent[2]["fused"]]
            loss = loss.clone()
            ent[2].clear()
            ent[1] = self.trainer.capture(fwd_bwd, warmup=0)
            return loss, logits
        loss = ent[1]()
        return loss, ent[2]["fused"]


class SteppedEpisodeGraphs:
    """Graph replay for rollouts whose next step depends on the action taken (sampling / RL, agent.py:409-500): T + 2 graphs - begin |
    step 0 | ... | step T-1 | ghost pass + backward + optimizer - over one DuetEpisodeBuffers; between two replays the caller reads
    step t's fused logits (`logits(t)`), picks the next viewpoint, updates its graph map and writes step t + 1 with bufs.put_step. The
    warm-up is one REAL training step on the episode loaded in `bufs`.

        g = SteppedEpisodeGraphs(trainer, model, bufs)
        bufs.load(ep, steps=False); g.begin()
        for t in range(T):
            bufs.put_step(t, step_t); g.step(t); a = g.logits(t).argmax(1)
        loss = g.finish()
    """

    def __init__(self, trainer, model, bufs, tape=None, **episode_kw):
        from vln_imagine_amd.duet.episode import TapedEpisode
        self.trainer, self.model, self.bufs = trainer, model, bufs
        self.ep = TapedEpisode(model, bufs, tape if tape is not None else ops.EpisodeTape(bufs.T), criterion=ops.cross_entropy_sum, **episode_kw)
        plan, backward = _runner(model, bufs, trainer)
        self._steps, self._out = {}, {}

        def begin():
            with plan():
                self.ep.begin()

        def step(t):
            def run():
                self._steps[t] = self.ep.step(t).detach()
            return run

        def finish():
            out = self.ep.finish()
            self._out["fused"] = [t.detach() for t in out["fused"]]
            return backward(out["loss"])

        self.graphs = trainer.capture(finish, warmup=1, stages=[begin] + [step(t) for t in range(bufs.T)])

    def begin(self):
        self.graphs.stage(0)

    def step(self, t):
        self.graphs.stage(1 + t)

    def logits(self, t):
        return self._steps[t]

    def finish(self):
        return self.graphs.finish()
