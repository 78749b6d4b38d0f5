#!/usr/bin/env python3
"""Generates tests/golden/duet_*.npz by running the REFERENCE GlocalTextPathNavCMT on CPU (build container only).
Same harness shims as make_golden_hamt.py; weights / inputs are closed-form (synth)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/VLN-DUET/map_nav_src")

import transformers  # noqa: E402
import models.vilmodel as REF  # noqa: E402  (the reference)

from vln_imagine_amd import synth  # noqa: E402
from vln_imagine_amd.duet.config import duet_config_dict  # noqa: E402
from vln_imagine_amd.duet.episode import DuetEpisodeTensors, run_episode  # noqa: E402
from tests.golden.variants import DUET_VARIANTS, DUET_C1, DUET_EP  # noqa: E402

_orig = REF.GlocalTextPathNavCMT.init_weights


def _guarded(self):
    if not getattr(self, "_graft_pi", False):
        self._graft_pi = True
        self.post_init()
    else:
        _orig(self)


REF.GlocalTextPathNavCMT.init_weights = _guarded


class _CloneIdentity(torch.nn.Module):
    def forward(self, x):
        return x.clone()


def build_reference(cfgd):
    cfg = transformers.BertConfig()
    for k, v in cfgd.items():
        setattr(cfg, k, v)
    m = REF.GlocalTextPathNavCMT(cfg)
    sd = m.state_dict()
    m.load_state_dict({k: torch.from_numpy(synth.init_param(k, v.shape)) if v.dtype.is_floating_point else v
                       for k, v in sd.items()})
    m.eval()
    if hasattr(m, "contrastive_alignment_model"):
        m.contrastive_alignment_model.image_proj.dropout = _CloneIdentity()
    return m


def run_variant(name, over, epkw):
    cfgd = duet_config_dict(**{**DUET_C1, **over})
    model = build_reference(cfgd)
    kw = dict(DUET_EP)
    kw.update(epkw)
    ep = synth.DuetEpisode(**kw)
    out = run_episode(lambda mode, b: model(mode, b), DuetEpisodeTensors(ep, "cpu"))
    out["loss"].backward()
    g = {"loss": out["loss"].detach().numpy(), "ml_loss": out["ml_loss"].detach().numpy(),
         "aux": out["aux"].detach().numpy() if torch.is_tensor(out["aux"]) else np.float32(out["aux"]),
         "imagine_embeds": out["imagine_embeds"].detach().numpy()}
    if "obj" in out:
        g["og_loss"] = out["og_loss"].detach().numpy()
        for t in range(ep.T):
            g[f"obj{t}"] = out["obj"][t].detach().numpy()
    for k, v in synth.probe(out["txt_embeds"].detach().numpy()).items():
        g[f"txt_embeds.{k}"] = v
    for t in range(ep.T):
        for nm in ("fused", "global", "local"):
            g[f"{nm}{t}"] = out[nm][t].detach().numpy()
        for nm in ("pano", "gmap", "vp"):
            for k, v in synth.probe(out[nm][t].detach().numpy()).items():
                g[f"{nm}{t}.{k}"] = v
    names, norms, heads = [], [], []
    for n, p in model.named_parameters():
        names.append(n)
        if p.grad is None:
            norms.append(-1.0); heads.append(np.zeros(8, np.float32))
        else:
            gr = p.grad.detach().double().reshape(-1)
            norms.append(float(gr.norm()))
            h = np.zeros(8, np.float32); h[:min(8, gr.numel())] = gr[:8].float().numpy()
            heads.append(h)
    g["grad_names"], g["grad_norms"], g["grad_heads"] = np.array(names), np.array(norms, np.float64), np.stack(heads)
    g["meta"] = np.array([f"torch={torch.__version__}", f"transformers={transformers.__version__}", f"variant={name}"])
    path = os.path.join(ROOT, "tests", "golden", f"duet_{name}.npz")
    np.savez_compressed(path, **g)
    print(f"{name}: loss={float(g['loss']):.6f} aux={float(g['aux']):.6f} fused0[0,:4]={g['fused0'][0, :4]} "
          f"-> {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, (over, epkw) in DUET_VARIANTS.items():
        if not only or name in only:
            run_variant(name, over, epkw)
