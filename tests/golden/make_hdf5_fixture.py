"""Writes the tiny HDF5 feature-store fixtures of tests/test_formats_cpu.py with vln_imagine_amd/hdf5_lite.write_store (h5py is not in
the build image): the two shapes the reference's stores have (VLN-HAMT/finetune_src/r2r/data_utils.py:15-47) -
  views_tiny.hdf5  'scan_viewpoint' -> [36, 800] float64, chunked (9 x 250: ragged last column chunk) + gzip: what `create_dataset(key, data.shape,
                   dtype='float', compression='gzip')` of the view-feature extraction scripts produces; 9 keys = two group leaves
  imag_tiny.hdf5   'pathid_instridx' -> [n_true, 768] float32, contiguous
Values are the closed-form hashes of vln_imagine_amd/synth.py, so the tests regenerate the expected arrays instead of storing them.
    python tests/golden/make_hdf5_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vln_imagine_amd import synth  # noqa: E402
from vln_imagine_amd.hdf5_lite import write_store  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def view_arrays():
    """9 keys (two 8-symbol group leaves), [36, 800] float64 on a 1/16 grid (so that gzip keeps the fixture small)."""
    return {f"scan{(i % 3):02d}_vp{i:03d}": (np.round(synth.det_uniform(f"h5/view{i}", (36, 800), -0.5, 0.5) * 16) / 16).astype(np.float64)
            for i in range(9)}


def imag_arrays():
    return {f"{100 + i}_{i % 3}": synth.det_uniform(f"h5/imag{i}", (1 + i % 4, 768), -0.5, 0.5).astype(np.float32) for i in range(5)}


if __name__ == "__main__":
    write_store(os.path.join(HERE, "views_tiny.hdf5"), view_arrays(), chunks=(9, 250), compress=True)
    write_store(os.path.join(HERE, "imag_tiny.hdf5"), imag_arrays(), chunks=None)
    for f in ("views_tiny.hdf5", "imag_tiny.hdf5"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
