"""Bisects the capture_end crash of the backward-graph capture (one variant per process: VARIANT env)."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.golden.variants import HAMT_C1
from tests.test_hamt_gpu import build_product
from vln_imagine_amd import dropin, graphed, ops, synth
from vln_imagine_amd.hamt.config import HamtConfig
from vln_imagine_amd.hamt.episode import EpisodeTensors
V = os.environ.get("VARIANT", "a")
cfg = HamtConfig(**HAMT_C1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pred_head_dropout_prob=0.0)
et = EpisodeTensors(synth.HamtEpisode(tag="graphed", B=8, L=80, V=37, I=4, T=2, ragged=True), "cuda")
graphed.ENABLED = False
m = build_product(cfg, torch.bfloat16).train()
w = dropin.wrap_hamt(m, feat_dropout=0.0)
loss, _ = dropin.hamt_agent_loss(w, et)
loss.backward()                      # eager iteration: autotune, shadows
for p in m.parameters():
    p.grad = None
ses = next(m.parameters())._vlni_auto
ids, masks = et.txt_ids.clone(), et.txt_masks.clone()
pool = torch.cuda.graph_pool_handle()
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, pool=pool):
    with torch.enable_grad():
        out = m("language", txt_ids=ids, txt_masks=masks)
print("fwd captured", out.shape, flush=True)
gout = torch.zeros_like(out)
prms = [p for p in ses.params if p.requires_grad]
if V != "nosession":
    ses.begin(register=False)
    ses.hold, ses.recorder = True, []
torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
print("capturing bwd, variant", V, flush=True)
with torch.cuda.graph(g2, pool=pool):
    with torch.enable_grad():
        if V in ("a", "nosession"):
            gi = torch.autograd.grad([out], prms, [gout], allow_unused=True)
        elif V == "b":
            out.backward(gout)
        elif V == "c":                # only a few parameters as inputs
            gi = torch.autograd.grad([out], prms[:8], [gout], allow_unused=True)
print("bwd captured", flush=True)
if V in ("a", "c", "nosession"):
    print("non-None param grads:", sum(g is not None for g in gi), flush=True)
g1.replay(); g2.replay(); torch.cuda.synchronize()
print("replayed ok; queue", len(ses.queue), flush=True)
