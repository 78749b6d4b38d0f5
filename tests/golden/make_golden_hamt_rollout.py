#!/usr/bin/env python3
"""Generates tests/golden/hamt_rollout.npz: Seq2SeqCMTAgent.rollout under teacher forcing (VLN-HAMT/finetune_src/r2r/agent_cmt.py:372-760)
with the REFERENCE NavCMT (build container only). The agent's builder loops cannot be imported here (MatterSim, h5py), so they come
from oracle/graph_oracle.py (OracleObsBuilders), driven through the same loop the product uses (vln_imagine_amd/hamt/rollout.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.golden.make_golden_hamt import build_reference  # noqa: E402  (puts the reference on sys.path)
from oracle.graph_oracle import OracleObsBuilders  # noqa: E402
from tests.golden.variants import HAMT_C1, hamt_rollout_setup  # noqa: E402
from vln_imagine_amd.hamt.config import hamt_config_dict  # noqa: E402
from vln_imagine_amd.hamt.rollout import rollout  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    walk, feats, keys, ep, imag, flags = hamt_rollout_setup()
    model = build_reference(hamt_config_dict(**HAMT_C1))
    t = torch.from_numpy
    out = rollout(lambda mode, **kw: model(mode, **kw), walk, OracleObsBuilders(feats, keys, imag, flags), t(ep.txt_ids), t(ep.txt_masks),
                  annotations=(ep.sub_instr_segs, ep.sub_instr_imag_flag, ep.noun_phrase_segs))
    out["loss"].backward()
    g = {"loss": out["loss"].detach().numpy(), "aux": out["aux"].detach().numpy(), "steps": np.int64(len(out["logits"])),
         "hist_lens": out["hist_lens"]}
    for i, (f, a) in enumerate(zip(out["logits"], out["targets"])):
        g[f"logits{i}"], g[f"target{i}"] = f.detach().numpy(), a
    names, norms = [], []
    for n, p in model.named_parameters():
        names.append(n)
        norms.append(-1.0 if p.grad is None else float(p.grad.detach().double().norm()))
    g["grad_names"], g["grad_norms"] = np.array(names), np.array(norms)
    path = os.path.join(ROOT, "tests", "golden", "hamt_rollout.npz")
    np.savez_compressed(path, **g)
    print("loss", float(g["loss"]), "aux", float(g["aux"]), "steps", int(g["steps"]), "hist_lens", g["hist_lens"], [a.tolist() for a in out["targets"]],
          os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
