"""Deterministic closed-form weights and synthetic episode inputs.

Everything here is a pure function of (name, flat index): a splitmix64 hash
mapped to 24-bit uniforms, exact in float32 and free of libm, so the golden
generator (this container, reference imported), the CPU oracle, the HIP path
and bench.py all see bit-identical tensors without shipping weights.

Shapes follow SURVEY.md section 8(d) (HAMT: 80 text tokens, 37 observation tokens =
candidates + STOP + other views, I imagination slots; reference builders:
VLN-HAMT/finetune_src/r2r/agent_cmt.py:130-176,247-313).
"""
import math

import numpy as np

_C1 = np.uint64(0x9E3779B97F4A7C15)
_C2 = np.uint64(0xBF58476D1CE4E5B9)
_C3 = np.uint64(0x94D049BB133111EB)


def _fnv1a64(name: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def det_u24(name: str, n: int) -> np.ndarray:
    """n deterministic integers in [0, 2^24) keyed by `name`."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = idx * _C1 + _fnv1a64(name)
        z = (z ^ (z >> np.uint64(30))) * _C2
        z = (z ^ (z >> np.uint64(27))) * _C3
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.int64)


def det_uniform(name: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = det_u24(name, n).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def det_randint(name: str, shape, lo: int, hi: int) -> np.ndarray:
    """integers in [lo, hi) (hi exclusive)."""
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + det_u24(name, n) % (hi - lo)).astype(np.int64).reshape(shape)


def init_param(name: str, shape) -> np.ndarray:
    """Closed-form initial value of a parameter from its state_dict key + shape."""
    shape = tuple(shape)
    if len(shape) == 1:
        if name.endswith("weight"):           # LayerNorm gamma
            return det_uniform(name, shape, 0.9, 1.1)
        return det_uniform(name, shape, -0.05, 0.05)  # every bias / LN beta
    if name.endswith("embeddings.weight") or name.endswith("embedding.weight"):
        return det_uniform(name, shape, -0.05, 0.05)  # lookup tables
    fan_in = shape[-1] if len(shape) >= 2 else 1
    a = min(0.5, 0.04 * math.sqrt(768.0 / max(fan_in, 1)))
    return det_uniform(name, shape, -a, a)


def fill_state_dict(named_shapes):
    """{key: np.float32 array} for an iterable of (key, shape)."""
    return {k: init_param(k, s) for k, s in named_shapes}


def angle_feat(name: str, shape_prefix):
    """(sin h, cos h, sin e, cos e) of deterministic heading/elevation."""
    h = det_uniform(name + ".h", shape_prefix, -math.pi, math.pi).astype(np.float64)
    e = det_uniform(name + ".e", shape_prefix, -math.pi / 6, math.pi / 6).astype(np.float64)
    # keep libm out of the fixture: a short Taylor/Bhaskara-free form is overkill;
    # np.sin in float64 then rounded to float32 is stable to far below 1e-7.
    return np.stack([np.sin(h), np.cos(h), np.sin(e), np.cos(e)], -1).astype(np.float32)


def _build_text_imagine(self, tag, B, L, I, ragged, feat=768, vocab=30522, imag_feat=None):
    """Instruction tokens, imagination features and the sub-instruction / noun-phrase annotation shared by the
    HAMT and DUET synthetic episodes (reference builders: r2r/agent_cmt.py:247-313, r2r/data_utils.py:119-450)."""
    k = lambda s: f"{tag}/{s}"
    # ---- text
    if ragged:
        lens = det_randint(k("txt_len"), (B,), max(8, L // 2), L + 1)
        lens[0] = L
    else:
        lens = np.full((B,), L, np.int64)
    self.txt_lens = lens
    ids = det_randint(k("txt_ids"), (B, L), 1, vocab)
    pos = np.arange(L)[None, :]
    self.txt_masks = pos < lens[:, None]
    self.txt_ids = np.where(self.txt_masks, ids, 0).astype(np.int64)
    # ---- imaginations + sub-instruction / noun-phrase annotation
    # imagination features keep their own width (768-d in the released stores, data_utils.py:15-47) when the view features are wider
    self.imagine_feats = det_uniform(k("imag"), (B, I, imag_feat or feat), -0.5, 0.5)
    valid = np.ones((B, I), bool)
    if ragged:
        valid = det_randint(k("imag_valid"), (B, I), 0, 5) > 0   # 80 % valid
        valid[0, :] = True
        if B > 1:
            valid[B - 1, :] = False                                # an all-False sample
    self.imagine_masks = valid
    self.imagine_feats = self.imagine_feats * valid[..., None]
    self.sub_instr_segs, self.sub_instr_imag_flag, self.noun_phrase_segs = [], [], []
    for b in range(B):
        n_tok = int(lens[b]) - 2                # tokens 1..len-2 are partitioned
        cuts = np.linspace(1, 1 + n_tok, I + 1).astype(int)
        segs, flags, nps = [], [], []
        for i in range(I):
            s, e = int(cuts[i]), int(max(cuts[i], cuts[i + 1] - 1))
            segs.append([s, e])
            flags.append("True" if valid[b, i] else "False")
            n_np = int(det_randint(k(f"nnp{b}_{i}"), (1,), 0, 3)[0])
            if ragged and b == 0 and i == 1:
                n_np = 0                        # flag-True slot with no noun phrase
            if b == 0 and i == 0:
                n_np = 2                        # multi-phrase slot
            cur, lst = s, []
            for j in range(n_np):
                ln = int(det_randint(k(f"npl{b}_{i}_{j}"), (1,), 1, 4)[0])
                a = cur
                z = min(e, a + ln - 1)
                if a > e:
                    break
                lst.append([a, z])
                cur = z + 2
            nps.append(lst)
        self.sub_instr_segs.append(segs)
        self.sub_instr_imag_flag.append(flags)
        self.noun_phrase_segs.append(nps)


class HamtEpisode:
    """Synthetic HAMT episode inputs (numpy), SURVEY.md section 8(d).

    tag      : string folded into every hash key (rank / seed)
    B, L, V, I, T: batch, text length (padded), observation tokens, imagination
               slots, number of steps.
    ragged   : ragged text lengths / observation lengths / invalid imagination
               slots (parity cases); False = dense (bench default).
    """

    def __init__(self, tag="ep0", B=4, L=80, V=37, I=4, T=2, ragged=True,
                 feat=768, ang=4, pano=36, vocab=30522, imag_feat=None):
        self.B, self.L, self.V, self.I, self.T = B, L, V, I, T
        k = lambda s: f"{tag}/{s}"
        _build_text_imagine(self, tag, B, L, I, ragged, feat, vocab, imag_feat)
        lens, valid = self.txt_lens, self.imagine_masks
        # ---- per-step observations / history / targets
        self.steps = []
        for t in range(T):
            kk = lambda s: k(f"t{t}/{s}")
            ncand = det_randint(kk("ncand"), (B,), 2, 7)
            if ragged:
                ob_lens = det_randint(kk("oblen"), (B,), V - 12, V + 1)
                ob_lens[0] = V
            else:
                ob_lens = np.full((B,), V, np.int64)
            ob_lens = np.maximum(ob_lens, ncand + 1)
            nav = np.zeros((B, V), np.int64)
            for b in range(B):
                nav[b, :ncand[b]] = 1
                nav[b, ncand[b]] = 2
            ob_masks = np.arange(V)[None, :] < ob_lens[:, None]
            ob_img = det_uniform(kk("ob_img"), (B, V, feat), -0.5, 0.5) * ob_masks[..., None]
            ob_ang = angle_feat(kk("ob_ang"), (B, V)) * ob_masks[..., None]
            # STOP token has zero image/angle feature (agent_cmt.py:157-160)
            for b in range(B):
                ob_img[b, ncand[b]] = 0
                ob_ang[b, ncand[b]] = 0
            target = det_randint(kk("target"), (B,), 0, 1 << 20) % (ncand + 1)
            if ragged and t == T - 1 and B > 2:
                target[2] = -100                    # an ended episode (ignore_index)
            step = dict(
                ob_img_feats=ob_img.astype(np.float32), ob_ang_feats=ob_ang.astype(np.float32),
                ob_nav_types=nav, ob_masks=ob_masks, ob_lens=ob_lens, target=target.astype(np.int64),
                hist_img_feats=det_uniform(kk("h_img"), (B, feat), -0.5, 0.5),
                hist_ang_feats=angle_feat(kk("h_ang"), (B,)),
                hist_pano_img_feats=det_uniform(kk("hp_img"), (B, pano, feat), -0.5, 0.5),
                hist_pano_ang_feats=angle_feat(kk("hp_ang"), (B, pano)),
            )
            self.steps.append(step)
        # history lengths before each step (all episodes alive in the synthetic case)
        self.hist_lens = [[1 + t] * B for t in range(T)]


def probe(x: np.ndarray, n: int = 512):
    """Small fingerprint of a big tensor: strided samples + sum + abs-sum."""
    f = np.asarray(x, np.float64).reshape(-1)
    stride = max(1, f.size // n)
    return dict(samples=f[::stride][:n].astype(np.float32), sum=np.float64(f.sum()),
                asum=np.float64(np.abs(f).sum()), shape=np.array(x.shape, np.int64))


class DuetEpisode:
    """Synthetic DUET episode: per step a 36-view panorama and a growing topological map
    (G_t = 1 [STOP] + (t+1) visited + (3+2t) frontier nodes), restating the host-side builders
    VLN-DUET/map_nav_src/r2r/agent.py:67-207 (_panorama_feature_variable, _nav_gmap_variable, _nav_vp_variable).

    Node images are tied to panorama outputs exactly like the agent does: a visited node = masked mean of that
    step's panorama embeddings (agent.py:468-479), a frontier node = one candidate view embedding of the step that
    first saw it. `node_src[t][b]` lists, per map node j >= 1, ("avg", step) or ("view", step, view_index)."""

    def __init__(self, tag="ep0", B=4, L=80, V=36, I=4, T=2, ragged=True, feat=768, vocab=30522, O=0, obj_feat=None):
        """O > 0: REVERIE-style panoramas, up to O object tokens appended after each sample's views (nav_type 2), restating
        VLN-DUET/map_nav_src/reverie/agent_obj.py:60-130 (_panorama_feature_variable) and :269-285 (_teacher_object)."""
        self.B, self.L, self.V, self.I, self.T, self.O = B, L, V, I, T, O
        _build_text_imagine(self, tag, B, L, I, ragged, feat, vocab)
        if O > 0:        # REVERIE: one imagination per instruction, always present (agent_obj.py:214-235)
            self.imagine_masks = np.ones((B, I), bool)
            self.imagine_feats = det_uniform(f"{tag}/imag", (B, I, feat), -0.5, 0.5)
        k = lambda s: f"{tag}/{s}"
        self.steps = []
        for t in range(T):
            kk = lambda s: k(f"t{t}/{s}")
            ncand = det_randint(kk("ncand"), (B,), 3, 7)                 # >= 3: one backtrack + new frontier nodes
            view_lens = det_randint(kk("vlen"), (B,), V - 8, V + 1) if ragged else np.full((B,), V, np.int64)
            view_lens[0] = V
            vmask = np.arange(V)[None, :] < view_lens[:, None]
            nav = np.zeros((B, V), np.int64)
            for b in range(B):
                nav[b, :ncand[b]] = 1
            G = 1 + (t + 1) + (3 + 2 * t)
            glens = np.full((B,), G, np.int64)
            if ragged and B > 1:
                glens[1] = G - 1                                          # one sample with a smaller map
            gmap_vpids, visited, step_ids, node_src, cand_vpids = [], [], [], [], []
            for b in range(B):
                g = int(glens[b])
                vis = [f"v{s}" for s in range(t + 1)]                     # visited viewpoints, step order
                fro = [f"f{j}" for j in range(g - 1 - len(vis))]          # frontier nodes
                gmap_vpids.append([None] + vis + fro)
                visited.append([0] + [1] * len(vis) + [0] * len(fro))
                step_ids.append([0] + [s + 1 for s in range(t + 1)] + [0] * len(fro))
                src = [("avg", s) for s in range(t + 1)]
                for j in range(len(fro)):
                    s = min(t, j // 2)                                    # the step that first saw this frontier node
                    src.append(("view", s, j % 3))
                node_src.append(src)
                # candidates of the current panorama: first a visited one (backtrack branch) if t > 0, then frontier nodes
                cands = ([vis[-2]] if t > 0 else []) + fro[::-1]
                cand_vpids.append(cands[:int(ncand[b])])
                ncand[b] = len(cand_vpids[-1])
                nav[b] = 0
                nav[b, :ncand[b]] = 1
            gm = np.arange(G)[None, :] < glens[:, None]
            pd = det_uniform(kk("pair"), (B, G, G), 0.0, 30.0)
            pd = np.triu(pd, 1)
            pd = pd + pd.transpose(0, 2, 1)
            pd[:, 0, :] = 0
            pd[:, :, 0] = 0
            pd = pd * (gm[:, :, None] & gm[:, None, :])
            vis_m = np.zeros((B, G), bool)
            sid = np.zeros((B, G), np.int64)
            for b in range(B):
                vis_m[b, :glens[b]] = np.array(visited[b], bool)
                sid[b, :glens[b]] = step_ids[b]
            # teacher action: STOP or an unvisited node of the map
            target = np.zeros((B,), np.int64)
            for b in range(B):
                opts = [0] + [j for j in range(1, int(glens[b])) if not visited[b][j]]
                target[b] = opts[int(det_randint(kk(f"tgt{b}"), (1,), 0, len(opts))[0])]
            if ragged and t == T - 1 and B > 2:
                target[2] = -100
            if O > 0:
                obj_lens = det_randint(kk("olen"), (B,), 1, O + 1) if ragged else np.full((B,), O, np.int64)
                obj_lens[0] = O
                S = V + O                                                 # sample 0 has V views and O objects
                pmask = np.arange(S)[None, :] < (view_lens + obj_lens)[:, None]
                nav_o = np.zeros((B, S), np.int64)
                obj_target = np.full((B,), -100, np.int64)
                for b in range(B):
                    nav_o[b, :view_lens[b]] = nav[b, :view_lens[b]]
                    nav_o[b, view_lens[b]:view_lens[b] + obj_lens[b]] = 2
                    if int(det_randint(kk(f"ogt{b}"), (1,), 0, 3)[0]) > 0:            # 2 of 3 samples stand at a goal viewpoint
                        obj_target[b] = 1 + view_lens[b] + int(det_randint(kk(f"ogi{b}"), (1,), 0, int(obj_lens[b]))[0])
                omask = np.arange(O)[None, :] < obj_lens[:, None]
                self.steps.append(dict(
                    view_img_fts=(det_uniform(kk("view"), (B, V, feat), -0.5, 0.5) * vmask[..., None]).astype(np.float32),
                    obj_img_fts=(det_uniform(kk("obj"), (B, O, obj_feat or feat), -0.5, 0.5) * omask[..., None]).astype(np.float32),
                    loc_fts=(det_uniform(kk("loc"), (B, S, 7), -0.6, 0.6) * pmask[..., None]).astype(np.float32),
                    nav_types=nav_o, view_lens=view_lens, obj_lens=obj_lens, obj_target=obj_target,
                    gmap_vpids=gmap_vpids, gmap_lens=glens, gmap_masks=gm, gmap_step_ids=sid,
                    gmap_pos_fts=(det_uniform(kk("gpos"), (B, G, 7), -0.6, 0.6) * gm[..., None]).astype(np.float32),
                    gmap_pair_dists=pd.astype(np.float32), gmap_visited_masks=vis_m, node_src=node_src,
                    vp_pos_fts=det_uniform(kk("vppos"), (B, S + 1, 14), -0.6, 0.6).astype(np.float32),
                    vp_cand_vpids=[[None] + c for c in cand_vpids], target=target))
                continue
            self.steps.append(dict(
                view_img_fts=(det_uniform(kk("view"), (B, V, feat), -0.5, 0.5) * vmask[..., None]).astype(np.float32),
                loc_fts=(det_uniform(kk("loc"), (B, V, 7), -0.6, 0.6) * vmask[..., None]).astype(np.float32),
                nav_types=nav, view_lens=view_lens,
                gmap_vpids=gmap_vpids, gmap_lens=glens, gmap_masks=gm, gmap_step_ids=sid,
                gmap_pos_fts=(det_uniform(kk("gpos"), (B, G, 7), -0.6, 0.6) * gm[..., None]).astype(np.float32),
                gmap_pair_dists=pd.astype(np.float32), gmap_visited_masks=vis_m, node_src=node_src,
                vp_pos_fts=det_uniform(kk("vppos"), (B, V + 1, 14), -0.6, 0.6).astype(np.float32),
                vp_cand_vpids=[[None] + c for c in cand_vpids], target=target))


class GraphWalk:
    """Seeded exploration of a synthetic building: what DUET's environment hands the agent per step, reduced to what the
    topological map and the navigation builders read (VLN-DUET/map_nav_src/r2r/env.py:213-262 observation fields `viewpoint`,
    `position`, `heading`, `elevation`, `candidate[].{viewpointId, position, pointId, heading, elevation}`).

    World: `n` viewpoints on a jittered grid over two floors, each linked to its `k` nearest neighbours (symmetrised).
    Walk: B agents, T steps; each step moves to a neighbour (unvisited first, so maps grow; sometimes back, so nodes are re-observed
    and shorter edges / new relaxations occur). Positions are python floats like the simulator's."""

    def __init__(self, tag="walk0", B=4, T=6, n=48, k=4, revisit=True):
        """revisit=False: ground-truth-like paths that never return to a seen viewpoint; an agent with no fresh neighbour stops
        there (`length[b]` = observations before its [stop])."""
        self.B, self.T = B, T
        self.length = [T] * B
        side = int(math.ceil(math.sqrt(n / 2)))
        xy = det_uniform(f"{tag}/jit", (n, 2), -0.8, 0.8).astype(np.float64)
        self.pos = []
        for v in range(n):
            f, r = divmod(v, side * side)
            self.pos.append((float(2.5 * (r % side) + xy[v, 0]), float(2.5 * (r // side) + xy[v, 1]), float(3.0 * f + 0.1 * xy[v, 0])))
        P = np.array(self.pos)
        d = np.sqrt(((P[:, None] - P[None]) ** 2).sum(-1))
        np.fill_diagonal(d, np.inf)
        self.adj = [set(np.argsort(d[v])[:k].tolist()) for v in range(n)]
        for v in range(n):
            for u in list(self.adj[v]):
                self.adj[u].add(v)
        self.adj = [sorted(a) for a in self.adj]
        self.name = lambda v: f"vp{v:03d}"
        self.steps = []                                      # steps[t][b] = observation dict
        cur = det_randint(f"{tag}/start", (B,), 0, n).tolist()
        been = [{c} for c in cur]
        for t in range(T):
            obs = []
            for b in range(B):
                v = cur[b]
                heading = float(det_uniform(f"{tag}/h{t}_{b}", (1,), 0.0, 2 * math.pi)[0])
                elevation = float(det_uniform(f"{tag}/e{t}_{b}", (1,), -0.5, 0.5)[0])
                cands = []
                for j, u in enumerate(self.adj[v]):
                    dx, dy, dz = (self.pos[u][a] - self.pos[v][a] for a in range(3))
                    rel_h = math.atan2(dx, dy) - heading
                    rel_e = math.atan2(dz, math.hypot(dx, dy)) - elevation
                    cands.append({"viewpointId": self.name(u), "position": self.pos[u], "heading": rel_h, "elevation": rel_e,
                                  "pointId": int((round(math.degrees(rel_h) / 30) % 12) + 12 * (1 + (rel_e > 0.26) - (rel_e < -0.26)))})
                obs.append({"instr_id": f"{tag}_{b}", "viewpoint": self.name(v), "position": self.pos[v], "heading": heading,
                            "elevation": elevation, "candidate": cands, "viewIndex": int(det_randint(f"{tag}/vi{t}_{b}", (1,), 0, 36)[0]),
                            "key": f"scan_{self.name(v)}"})
            self.steps.append(obs)
            for b in range(B):
                fresh = [u for u in self.adj[cur[b]] if u not in been[b]]
                r = int(det_randint(f"{tag}/mv{t}_{b}", (1,), 0, 1 << 20)[0])
                if not revisit:
                    if not fresh or self.length[b] <= t:
                        self.length[b] = min(self.length[b], t + 1)      # stays where it is from now on
                        continue
                    cur[b] = fresh[r % len(fresh)]
                    been[b].add(cur[b])
                    continue
                pool = fresh if fresh and r % 4 != 0 else self.adj[cur[b]]
                cur[b] = pool[r % len(pool)]
                been[b].add(cur[b])
