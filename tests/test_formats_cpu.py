"""On-disk formats: the HDF5 branch of formats._iter_store executes here - through hdf5_lite.py, the
pure-Python reader of the classic HDF5 subset the reference's feature stores use (VLN-HAMT/finetune_src/r2r/data_utils.py:15-47) -
on committed fixtures (tests/golden/*.hdf5, written by tests/golden/make_hdf5_fixture.py) and on files written on the fly."""
import os
import struct

import numpy as np
import pytest

from tests.golden.make_hdf5_fixture import imag_arrays, view_arrays
from vln_imagine_amd import formats
from vln_imagine_amd.hdf5_lite import Hdf5File, read_store, write_store

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_view_feature_store_hdf5_chunked_gzip():
    """'scan_viewpoint' -> [36, >= 768] float64, chunked + deflate, keys spread over two group leaves: the first 768 columns as float32,
    exactly as ImageFeaturesDB.get_image_feature slices and casts them (data_utils.py:27)."""
    want = view_arrays()
    got = dict(formats._iter_store(os.path.join(GOLDEN, "views_tiny.hdf5"), 768))
    assert sorted(got) == sorted(want) and len(got) == 9
    for k, a in want.items():
        assert got[k].dtype == np.float32 and got[k].shape == (36, 768)
        assert np.array_equal(got[k], a[:, :768].astype(np.float32)), k


def test_imagination_feature_store_hdf5_contiguous():
    """'pathid_instridx' -> [n_true, 768] float32 (ImaginationImageFeaturesDB, data_utils.py:33-47), ragged row counts."""
    want = imag_arrays()
    got = dict(formats._iter_store(os.path.join(GOLDEN, "imag_tiny.hdf5"), 768))
    assert sorted(got) == sorted(want)
    for k, a in want.items():
        assert np.array_equal(got[k], a), k


@pytest.mark.parametrize("chunks,compress", [(None, False), ((5, 7), True), ((5, 7), False), ((64, 64), True)])
def test_hdf5_lite_round_trip(tmp_path, chunks, compress):
    rng = np.random.default_rng(3)
    arrays = {f"k{i:03d}": rng.standard_normal((3 + i % 5, 11 + i)).astype([np.float32, np.float64][i % 2]) for i in range(21)}   # 3 leaves
    arrays["ints"] = rng.integers(-5, 5, (4, 6)).astype(np.int32)
    path = str(tmp_path / "s.hdf5")
    write_store(path, arrays, chunks=chunks, compress=compress)
    back = read_store(path)
    assert sorted(back) == sorted(arrays)
    for k, a in arrays.items():
        assert back[k].dtype == a.dtype and np.array_equal(back[k], a), k
    f = Hdf5File(path)
    assert "k007" in f and "nope" not in f


def test_hdf5_lite_rejects_what_it_does_not_read(tmp_path):
    path = str(tmp_path / "s.hdf5")
    write_store(path, {"a": np.zeros((2, 2), np.float32)})
    raw = bytearray(open(path, "rb").read())
    raw[8] = 2                                                  # a version-2 superblock (libver='latest' files)
    open(path, "wb").write(bytes(raw))
    with pytest.raises(NotImplementedError, match="superblock version 2"):
        Hdf5File(path)
    open(path, "wb").write(b"not hdf5 at all")
    with pytest.raises(ValueError):
        Hdf5File(path)


def test_superblock_fields_follow_the_specification(tmp_path):
    """The fixture writer's superblock, byte for byte where the format fixes it (HDF5 File Format Specification III.A, version 0)."""
    path = str(tmp_path / "s.hdf5")
    write_store(path, {"a": np.ones((2, 3), np.float32)})
    b = open(path, "rb").read()
    assert b[:8] == b"\x89HDF\r\n\x1a\n" and b[8] == 0 and b[13] == 8 and b[14] == 8
    leaf_k, internal_k = struct.unpack_from("<HH", b, 16)
    assert (leaf_k, internal_k) == (4, 16)
    base, _, eof, _ = struct.unpack_from("<QQQQ", b, 24)
    assert base == 0 and eof == len(b)
    assert struct.unpack_from("<I", b, 56 + 16)[0] == 1           # root entry caches the B-tree / heap addresses


# ---- cross-checks against the HDF5 library's own tools, where the image carries them (HDF5 1.10.6 under /opt/conda/bin) ----
H5TOOLS = os.environ.get("H5TOOLS", "/opt/conda/bin")
needs_h5tools = pytest.mark.skipif(not all(os.path.exists(os.path.join(H5TOOLS, t)) for t in ("h5repack", "h5dump")),
                                   reason="HDF5 command-line tools not installed")


@needs_h5tools
@pytest.mark.parametrize("args", [["-f", "SHUF", "-f", "GZIP=6"], ["-f", "GZIP=1", "-f", "FLET"], ["-l", "CHUNK=36x800", "-f", "GZIP=9"],
                                  ["-l", "CHUNK=5x33"], ["-l", "CONTI", "-f", "NONE"], ["-l", "COMPA", "-f", "NONE"]])
def test_reader_on_library_written_variants(tmp_path, args):
    """h5repack rewrites the view store with other layouts / filter pipelines (shuffle + deflate is h5py's `shuffle=True`; a 5 x 33
    chunking gives a B-tree with more than one level); the reader returns the same arrays."""
    import subprocess
    want = {k: v[:8, :100] if "COMPA" in args else v for k, v in view_arrays().items()}
    src = os.path.join(GOLDEN, "views_tiny.hdf5")
    if "COMPA" in args:                                        # compact data lives in the object header: at most 64 KB
        src = str(tmp_path / "small.hdf5")
        write_store(src, want)
    dst = str(tmp_path / "v.hdf5")
    subprocess.run([os.path.join(H5TOOLS, "h5repack")] + args + [src, dst], check=True, capture_output=True)
    got = read_store(dst)
    assert sorted(got) == sorted(want)
    for k, a in want.items():
        assert got[k].dtype == a.dtype and np.array_equal(got[k], a), k


@needs_h5tools
def test_reader_refuses_the_new_file_format_by_name(tmp_path):
    import subprocess
    dst = str(tmp_path / "latest.hdf5")
    subprocess.run([os.path.join(H5TOOLS, "h5repack"), "-L", os.path.join(GOLDEN, "imag_tiny.hdf5"), dst], check=True, capture_output=True)
    with pytest.raises(NotImplementedError, match="superblock version"):
        Hdf5File(dst)


@needs_h5tools
@pytest.mark.parametrize("chunks,compress", [(None, False), ((9, 250), True), ((7, 64), False)])
def test_library_reads_what_the_fixture_writer_writes(tmp_path, chunks, compress):
    """write_store is test infrastructure; the library's h5dump must read its files back to the same bytes."""
    import subprocess
    rng = np.random.default_rng(5)
    arrays = {f"k{i:02d}": rng.standard_normal((36, 300)).astype([np.float64, np.float32][i % 2]) for i in range(11)}
    path = str(tmp_path / "w.hdf5")
    write_store(path, arrays, chunks=chunks, compress=compress)
    for k, a in arrays.items():
        out = str(tmp_path / "d.bin")
        subprocess.run([os.path.join(H5TOOLS, "h5dump"), "-d", "/" + k, "-b", "LE", "-o", out, path], check=True, capture_output=True)
        assert np.array_equal(np.fromfile(out, a.dtype.newbyteorder("<")).reshape(a.shape), a), k


@pytest.mark.skipif(not os.path.exists(os.path.join(H5TOOLS, "h5import")), reason="HDF5 command-line tools not installed")
def test_reader_walks_a_multi_level_group_btree_written_by_the_library(tmp_path):
    """The real view-feature stores hold ~10 k keys: the root group's B-tree then has internal nodes above its symbol-table leaves and the
    local heap has grown. 300 datasets written by the library's own h5import (> 2 * 16 * 8 + 1 keys force a level-1 tree)."""
    import numpy as np
    from tests.golden.make_hdf5_fixture import h5import
    from vln_imagine_amd.hdf5_lite import Hdf5File
    arrays = {f"scan{(i * 7) % 11:02d}_vp{i:04d}": np.full((2, 3), float(i), np.float32) + np.arange(6, dtype=np.float32).reshape(2, 3) / 8
              for i in range(300)}
    path = str(tmp_path / "many.hdf5")
    keys = list(arrays)
    for c in range(0, len(keys), 25):
        h5import(path, {k: arrays[k] for k in keys[c:c + 25]}, append=c > 0)
    f = Hdf5File(path)
    assert sorted(f.keys()) == sorted(arrays)
    for k in list(arrays)[::37] + [list(arrays)[-1]]:
        assert np.array_equal(f.dataset(k), arrays[k]), k
