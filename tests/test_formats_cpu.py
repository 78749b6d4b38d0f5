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
