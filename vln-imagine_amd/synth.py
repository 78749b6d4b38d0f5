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


class HamtEpisode:
    """Synthetic HAMT episode inputs (numpy), SURVEY.md section 8(d).

    tag      : string folded into every hash key (rank / seed)
    B, L, V, I, T: batch, text length (padded), observation tokens, imagination
               slots, number of steps.
    ragged   : ragged text lengths / observation lengths / invalid imagination
               slots (parity cases); False = dense (bench default).
    """

    def __init__(self, tag="ep0", B=4, L=80, V=37, I=4, T=2, ragged=True,
                 feat=768, ang=4, pano=36, vocab=30522):
        self.B, self.L, self.V, self.I, self.T = B, L, V, I, T
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
        self.imagine_feats = det_uniform(k("imag"), (B, I, feat), -0.5, 0.5)
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
