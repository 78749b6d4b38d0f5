"""Which parameters receive their gradient through a torch-native autograd node (not accumulated in place by the operators' kernels) in each
wrapper mode: eager torch.autograd.grad(outputs, all parameters) inside an open gradient session - the non-None results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.golden.variants import HAMT_C1, DUET_C1
from vln_imagine_amd import dropin, graphed, ops, synth
graphed.ENABLED = False


def report(tag, m, outs):
    ses = next(m.parameters())._vlni_auto
    names = {id(p): n for n, p in m.named_parameters()}
    prms = [p for p in ses.params if p.requires_grad]
    outs = [o for o in outs if torch.is_tensor(o) and o.requires_grad]
    ses.begin(register=False)
    ses.hold = True
    gi = torch.autograd.grad(outs, prms, [torch.ones_like(o) for o in outs], allow_unused=True)
    ses.hold = False
    ses.end(quiet=True)
    print(tag, "native:", [names[id(p)] for p, g in zip(prms, gi) if g is not None], flush=True)


if os.environ.get("FAMILY", "hamt") == "hamt":
    from tests.test_hamt_gpu import build_product
    from vln_imagine_amd.hamt.config import HamtConfig
    from vln_imagine_amd.hamt.episode import EpisodeTensors
    cfg = HamtConfig(**HAMT_C1)
    et = EpisodeTensors(synth.HamtEpisode(tag="graphed", B=8, L=80, V=37, I=4, T=2, ragged=True), "cuda")
    m = build_product(cfg, torch.bfloat16).train()
    w = dropin.wrap_hamt(m, feat_dropout=0.0)
    txt = w("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks)
    report("language", m, [txt])
    txt = w("language", txt_ids=et.txt_ids, txt_masks=et.txt_masks).detach().requires_grad_(True)
    img = w("imagine", imagine_pano_img_feats=et.imagine_feats, imagine_masks=et.imagine_masks)
    report("imagine", m, [img])
    img = w("imagine", imagine_pano_img_feats=et.imagine_feats, imagine_masks=et.imagine_masks).detach().requires_grad_(True)
    h0 = w("history")
    report("history cls", m, [h0])
    s = et.steps[0]
    h1 = w("history", hist_img_feats=s["hist_img_feats"], hist_ang_feats=s["hist_ang_feats"], hist_pano_img_feats=s["hist_pano_img_feats"],
           hist_pano_ang_feats=s["hist_pano_ang_feats"], ob_step=0)
    report("history", m, [h1])
    hist = [w("history").expand(et.B, -1).detach().requires_grad_(True)]
    lg, st = w("visual", txt_embeds=txt, txt_masks=et.txt_masks, hist_embeds=hist, hist_lens=[1] * et.B, ob_img_feats=s["ob_img_feats"],
               ob_ang_feats=s["ob_ang_feats"], ob_nav_types=s["ob_nav_types"], ob_masks=s["ob_masks"], return_states=True, imagine_embeds=img,
               imagine_masks=et.imagine_masks)
    report("visual", m, [lg, st])
else:
    from tests.test_duet_gpu import build_product
    from vln_imagine_amd.duet.config import DuetConfig
    from vln_imagine_amd.duet.episode import DuetEpisodeTensors, run_episode
    cfg = DuetConfig(**DUET_C1)
    et = DuetEpisodeTensors(synth.DuetEpisode(tag="graphed", B=8, L=80, V=36, I=4, T=2, ragged=True), "cuda")
    m = build_product(cfg, torch.bfloat16).train()
    w = dropin.wrap_duet(m, feat_dropout=0.0)
    seen = {}
    real = w.forward

    def spy(mode, batch):
        out = real(mode, batch)
        if mode not in seen:
            seen[mode] = out
        return out
    w.forward = spy
    out = run_episode(w, et, criterion=ops.cross_entropy_sum, keep=False)
    for mode, o in seen.items():
        flat = []
        def walk(x):
            if torch.is_tensor(x):
                flat.append(x)
            elif isinstance(x, dict):
                for v in x.values():
                    walk(v)
            elif isinstance(x, (list, tuple)):
                for v in x:
                    walk(v)
        walk(o)
        try:
            report("duet " + mode, m, flat)
        except RuntimeError as e:
            print("duet", mode, "error", str(e)[:200])
