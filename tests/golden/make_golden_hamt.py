#!/usr/bin/env python3
"""Generates tests/golden/hamt_*.npz by running the REFERENCE NavCMT on CPU.

Runs only in the build container (needs /root/reference). The reference module is
imported as-is (never copied); three harness shims from SURVEY.md section 8(c):
  1. NavCMT.init_weights guard (transformers 5.x needs post_init()).
  2. config = transformers.BertConfig() + the attributes vlnbert_init.py:43-76 sets.
  3. image_proj.dropout -> clone-identity so backward through the in-place aux head
     is defined in eval mode (vilmodel_cmt.py:781).
  4. (variant c1_no_lang_ca only) LXRTXLayer.self_att returns `(lang_input,)` under no_lang_ca (:402) and
     LXRTXLayer.forward then indexes element [1] of it for a visualisation softmax (:438): the reference raises
     IndexError for every no_lang_ca model. The shim appends a dummy second element to that tuple; nothing else
     of the reference changes (the softmax of the dummy is discarded, :443-444).
Weights and inputs come from vln_imagine_amd.synth closed forms, so the fixtures hold
only outputs (logits, losses, small embeddings, probes of large tensors, gradient
norms + leading elements).

usage: python tests/golden/make_golden_hamt.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/VLN-HAMT/finetune_src")

import transformers  # noqa: E402
import models.vilmodel_cmt as REF  # noqa: E402  (the reference)

from vln_imagine_amd import synth  # noqa: E402
from vln_imagine_amd.hamt.episode import EpisodeTensors, run_episode  # noqa: E402
from vln_imagine_amd.hamt.config import hamt_config_dict  # noqa: E402
from tests.golden.variants import HAMT_VARIANTS as VARIANTS, HAMT_C1, HAMT_EP, hamt_variant_run_kw  # noqa: E402

_orig_init = REF.NavCMT.init_weights


def _guarded(self):
    if not getattr(self, "_graft_pi", False):
        self._graft_pi = True
        self.post_init()
    else:
        _orig_init(self)


REF.NavCMT.init_weights = _guarded


_orig_self_att = REF.LXRTXLayer.self_att


def _self_att_padded(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask):      # shim 4
    lang_out, visn_out = _orig_self_att(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask)
    if self.no_lang_ca and len(lang_out) == 1:
        lang_out = (lang_out[0], torch.zeros(1))
    return lang_out, visn_out


REF.LXRTXLayer.self_att = _self_att_padded


class _CloneIdentity(torch.nn.Module):
    def forward(self, x):
        return x.clone()


def build_reference(cfgd):
    cfg = transformers.BertConfig()
    for k, v in cfgd.items():
        setattr(cfg, k, v)
    m = REF.NavCMT(cfg)
    sd = m.state_dict()
    new = {k: torch.from_numpy(synth.init_param(k, v.shape)) if v.dtype.is_floating_point else v
           for k, v in sd.items()}
    m.load_state_dict(new)
    m.eval()
    if hasattr(m, "contrastive_alignment_model"):
        m.contrastive_alignment_model.image_proj.dropout = _CloneIdentity()
    return m


def run_variant(name, over, epkw, *_):
    cfgd = hamt_config_dict(**{**HAMT_C1, **over})
    model = build_reference(cfgd)
    kw = dict(HAMT_EP)
    kw.update(epkw)
    ep = synth.HamtEpisode(**kw)
    et = EpisodeTensors(ep, "cpu")
    kink = []                      # pre-activations of the action head's ReLU on rows that reach the loss: a value within float32 rounding of
    h = model.next_action.net[1].register_forward_hook(lambda m, i, o: kink.append(i[0].detach()))     # zero makes the gradient ill-defined
    out = run_episode(model, et, bypass=cfgd["bypass_imag_encoder"], **{"use_aux": True, **hamt_variant_run_kw(name)})
    h.remove()
    margin = min(float(z[torch.isfinite(lg)].abs().min()) for z, lg in zip(kink, out["logits"]))
    assert margin > 2e-6, f"{name}: an action-head ReLU input is {margin:.1e} from zero on a scored row - pick another episode tag (variants.py)"
    out["loss"].backward()
    g = {}
    g["loss"] = out["loss"].detach().numpy()
    g["ml_loss"] = out["ml_loss"].detach().numpy()
    g["aux"] = out["aux"].detach().numpy() if torch.is_tensor(out["aux"]) else np.float32(out["aux"] or 0.0)
    if out["imagine_embeds"] is not None:             # None for imagine_enc_pano=False
        g["imagine_embeds"] = out["imagine_embeds"].detach().numpy()
    g["hist_cls"] = out["hist_cls"].detach().numpy()
    txt_list = out["txt_embeds"] if isinstance(out["txt_embeds"], list) else [out["txt_embeds"]]     # no_lang_ca: one per layer + the input
    for i, te in enumerate(txt_list):
        for k, v in synth.probe(te.detach().numpy()).items():
            g[f"txt_embeds.{k}" if i == 0 else f"txt_embeds{i}.{k}"] = v
    for t in range(ep.T):
        g[f"logits{t}"] = out["logits"][t].detach().numpy()
        g[f"state{t}"] = out["states"][t].detach().numpy()
        g[f"hist{t}"] = out["hist"][t].detach().numpy()
        for nm in ("txt_o", "ob_o", "hist_o"):
            for k, v in synth.probe(out[nm][t].detach().numpy()).items():
                g[f"{nm}{t}.{k}"] = v
    names, norms, heads = [], [], []
    for n, p in model.named_parameters():
        names.append(n)
        if p.grad is None:
            norms.append(-1.0)
            heads.append(np.zeros(8, np.float32))
        else:
            gr = p.grad.detach().double().reshape(-1)
            norms.append(float(gr.norm()))
            h = np.zeros(8, np.float32)
            h[:min(8, gr.numel())] = gr[:8].float().numpy()
            heads.append(h)
    g["grad_names"] = np.array(names)
    g["grad_norms"] = np.array(norms, np.float64)
    g["grad_heads"] = np.stack(heads)
    g["meta"] = np.array([f"torch={torch.__version__}", f"transformers={transformers.__version__}",
                          f"variant={name}", f"cfg={sorted(over.items())}", f"ep={sorted(kw.items())}"])
    path = os.path.join(ROOT, "tests", "golden", f"hamt_{name}.npz")
    np.savez_compressed(path, **g)
    print(f"{name}: relu margin {margin:.1e} loss={float(g['loss']):.6f} aux={float(g['aux']):.6f} "
          f"logit0[0,:3]={g['logits0'][0, :3]} -> {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, v in VARIANTS.items():
        if only and name not in only:
            continue
        run_variant(name, *v)
