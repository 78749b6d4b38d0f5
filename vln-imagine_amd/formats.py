"""On-disk formats of the imagination pipeline, read ONCE into the resident tables of builders.py.

What the reference reads at run time, and where:
  * generated-flag JSON   r2r/parser.py:155-176   list of {path_id, instruction, generated_imaginations: ['True' | 'False', ...]}
                                                   -> {'<path_id>_<instruction>': flags}
  * annotation JSON       r2r/main.py:36-61 + r2r/env.py:125-127   `fgr2r_nounphrase_segmentation_data_<split>.json`:
                          list of {instruction_id, instr_segmentation_indices, noun_phrase_indices, ...}
  * feature stores        r2r/data_utils.py:15-47  HDF5, one dataset per key: view features 'scan_viewpoint' -> [36, >= D],
                          imagination features 'pathid_instridx' -> [n_true, >= D]; first D columns, float32
The reference re-opens the HDF5 file per key inside the rollout and pads on the host every batch; here each file is walked once
and its rows live in HBM afterwards. HDF5 goes through h5py where it is installed and through hdf5_lite.py (a pure-Python reader of
the classic HDF5 subset these stores use: symbol-table root group, contiguous or chunked + gzip 2-D datasets) where it is not - it
is not part of this image; the same tables also load from a directory of .npy files or an .npz archive with identical keys."""
import json
import os

import numpy as np


def load_generated_flags(path):
    """{instr_id: ['True' | 'False' per sub-instruction]} (parser.py:155-176; instr_id = '<path_id>_<instruction index>')."""
    with open(path) as f:
        return {f"{d['path_id']}_{d['instruction']}": list(d["generated_imaginations"]) for d in json.load(f)}


def load_annotations(path):
    """(instr_id -> sub-instruction token spans, instr_id -> noun-phrase spans per sub-instruction), env.py:126-127."""
    with open(path) as f:
        data = json.load(f)
    return ({d["instruction_id"]: d["instr_segmentation_indices"] for d in data},
            {d["instruction_id"]: d["noun_phrase_indices"] for d in data})


class AnnotationIndex:
    """What the alignment head needs per batch (agent_cmt.py:436-459): spans, generated flags and noun phrases of the batch's
    instructions, as python lists in batch order."""

    def __init__(self, annotation_json, generated_flag_json):
        self.segs, self.nps = load_annotations(annotation_json)
        self.flags = load_generated_flags(generated_flag_json) if isinstance(generated_flag_json, str) else dict(generated_flag_json)
        for k, fl in self.flags.items():                       # env.py:129 counts the 'True's; a span list must cover every flag
            if k in self.segs and len(self.segs[k]) != len(fl):
                raise ValueError(f"{k}: {len(self.segs[k])} sub-instruction spans for {len(fl)} generated flags")

    def batch(self, instr_ids):
        return ([self.segs[i] for i in instr_ids], [self.flags[i] for i in instr_ids], [self.nps[i] for i in instr_ids])


def _iter_store(path, feat_size):
    """(key, float32 [rows, feat_size]) of an HDF5 file, an .npz archive or a directory of <key>.npy files."""
    if os.path.isdir(path):
        for name in sorted(os.listdir(path)):
            if name.endswith(".npy"):
                yield name[:-4], np.load(os.path.join(path, name))[:, :feat_size].astype(np.float32)
    elif path.endswith(".npz"):
        with np.load(path) as z:
            for k in z.files:
                yield k, z[k][:, :feat_size].astype(np.float32)
    else:
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            with h5py.File(path, "r") as f:
                for k in f.keys():
                    yield k, f[k][...][:, :feat_size].astype(np.float32)
        else:                               # no h5py in this image: the classic-HDF5 subset these stores use, read in pure Python
            from .hdf5_lite import Hdf5File
            f = Hdf5File(path)
            for k in f.keys():
                yield k, f[k][:, :feat_size].astype(np.float32)


def load_view_features(path, feat_size=768, device="cuda", dtype=None):
    """ImageFeaturesDB (data_utils.py:15-30) -> builders.ResidentFeatures: every 'scan_viewpoint' -> [36, feat_size] row block."""
    import torch

    from .builders import ResidentFeatures
    keys, rows = [], []
    for k, a in _iter_store(path, feat_size):
        if a.shape[0] != 36:
            raise ValueError(f"{k}: {a.shape[0]} views, expected 36")
        keys.append(k)
        rows.append(a)
    return ResidentFeatures(np.stack(rows), keys, device=device, dtype=dtype or torch.float32)


def load_imagination_table(path, generated_flags, feat_size=768, device="cuda", dtype=None):
    """ImaginationImageFeaturesDB (data_utils.py:33-47) + generated flags -> builders.ImaginationTable."""
    import torch

    from .builders import ImaginationTable
    return ImaginationTable(dict(_iter_store(path, feat_size)), generated_flags, feat_size=feat_size, device=device, dtype=dtype or torch.float32)
