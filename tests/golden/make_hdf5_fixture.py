"""Writes the tiny HDF5 feature-store fixtures of tests/test_formats_cpu.py WITH THE HDF5 LIBRARY ITSELF: `h5import` of the HDF5 1.10.6
tools this image carries under /opt/conda/bin (h5py, which the reference uses, is not installed; libhdf5 is what h5py writes through).
The two shapes the reference's stores have (VLN-HAMT/finetune_src/r2r/data_utils.py:15-47) -
  views_tiny.hdf5  'scan_viewpoint' -> [36, 800] float64, chunked (9 x 250: ragged last column chunk) + gzip 4: what `create_dataset(key,
                   data.shape, dtype='float', compression='gzip')` of the view-feature extraction scripts produces; 9 keys = two group leaves
  imag_tiny.hdf5   'pathid_instridx' -> [n_true, 768] float32, contiguous
Values are the closed-form hashes of vln_imagine_amd/synth.py, so the tests regenerate the expected arrays instead of storing them;
the files pin vln_imagine_amd/hdf5_lite.py (the pure-Python reader) to bytes the real library produced.
    python tests/golden/make_hdf5_fixture.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vln_imagine_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
H5TOOLS = os.environ.get("H5TOOLS", "/opt/conda/bin")


def view_arrays():
    """9 keys (two 8-symbol group leaves), [36, 800] float64 on a 1/16 grid (so that gzip keeps the fixture small)."""
    return {f"scan{(i % 3):02d}_vp{i:03d}": (np.round(synth.det_uniform(f"h5/view{i}", (36, 800), -0.5, 0.5) * 16) / 16).astype(np.float64)
            for i in range(9)}


def imag_arrays():
    return {f"{100 + i}_{i % 3}": synth.det_uniform(f"h5/imag{i}", (1 + i % 4, 768), -0.5, 0.5).astype(np.float32) for i in range(5)}


def h5import(path, arrays, chunks=None, gzip=None, append=False):
    """One `h5import` call: every array as a raw little-endian file plus its configuration file (h5import's documented keywords).
    append=True adds the datasets to an existing file (h5import takes at most 30 inputs per call)."""
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [os.path.join(H5TOOLS, "h5import")]
        for i, (k, a) in enumerate(arrays.items()):
            a = np.ascontiguousarray(a)
            bits = a.dtype.itemsize * 8
            a.astype(a.dtype.newbyteorder("<")).tofile(os.path.join(tmp, f"{i}.bin"))
            cfg = [f"PATH {k}", "INPUT-CLASS FP", f"INPUT-SIZE {bits}", "INPUT-BYTE-ORDER LE", f"RANK {a.ndim}",
                   "DIMENSION-SIZES " + " ".join(map(str, a.shape)), "OUTPUT-CLASS FP", f"OUTPUT-SIZE {bits}", "OUTPUT-BYTE-ORDER LE"]
            if chunks:
                cfg.append("CHUNKED-DIMENSION-SIZES " + " ".join(map(str, chunks)))
            if gzip:
                cfg += ["COMPRESSION-TYPE GZIP", f"COMPRESSION-PARAM {gzip}"]
            with open(os.path.join(tmp, f"{i}.cfg"), "w") as f:
                f.write("\n".join(cfg) + "\n")
            cmd += [os.path.join(tmp, f"{i}.bin"), "-c", os.path.join(tmp, f"{i}.cfg")]
        if os.path.exists(path) and not append:
            os.remove(path)
        subprocess.run(cmd + ["-o", path], check=True, capture_output=True)


if __name__ == "__main__":
    h5import(os.path.join(HERE, "views_tiny.hdf5"), view_arrays(), chunks=(9, 250), gzip=4)
    h5import(os.path.join(HERE, "imag_tiny.hdf5"), imag_arrays())
    for f in ("views_tiny.hdf5", "imag_tiny.hdf5"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
