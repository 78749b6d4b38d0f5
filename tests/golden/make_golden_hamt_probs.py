#!/usr/bin/env python3
"""Generates tests/golden/hamt_attention_probs.npz: the visualisation outputs of the REFERENCE NavCMT
(`return_cross_attention_probs=True`, VLN-HAMT/finetune_src/models/vilmodel_cmt.py:1128-1153,1202-1203) for step 0 of the
'c1_shipped' synthetic episode (build container only). Stored per layer and map: shape, row sums' range, a strided sample."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests.golden.make_golden_hamt import build_reference  # noqa: E402
from tests.golden.variants import HAMT_C1, hamt_variant_setup, probs_sample as sample, visual_step0  # noqa: E402
from vln_imagine_amd.hamt.config import hamt_config_dict  # noqa: E402
from vln_imagine_amd.hamt.episode import EpisodeTensors  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg, ep = hamt_variant_setup("c1_shipped")
    model = build_reference(hamt_config_dict(**HAMT_C1))
    with torch.no_grad():
        out = visual_step0(lambda mode, **kw: model(mode, **kw), EpisodeTensors(ep))
    g = {"logits": out[0].numpy(), "layers": np.int64(len(out[4]))}
    for l, ((lq, vq), (ls, vs)) in enumerate(zip(out[4], out[5])):
        for name, p in (("lq", lq), ("vq", vq), ("ls", ls), ("vs", vs)):
            g[f"{name}{l}.shape"] = np.array(p.shape)
            g[f"{name}{l}.sample"] = sample(p)
    path = os.path.join(ROOT, "tests", "golden", "hamt_attention_probs.npz")
    np.savez_compressed(path, **g)
    print({k: v.tolist() for k, v in g.items() if k.endswith("shape")}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
