"""CPU-only checks of the C ABI library: it loads, exports every symbol include/rsx.h declares, and its
host-side (no-GPU) functions agree bit for bit with the oracle.  No device compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import criteo, hashing, tfrecord

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from recsys_amd import _lib, build
    build.build(verbose=False)
    return _lib.lib()


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "rsx.h")).read()
    declared = set(re.findall(r"\b(rsx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rsx_stream_t"}
    assert len(declared) >= 10
    raw = C.CDLL(os.path.join(ROOT, "recsys_amd", "librsx.so"))
    for name in sorted(declared):
        assert hasattr(raw, name), "librsx.so does not export %s" % name
    assert L.rsx_version() >= 100
    assert L.rsx_strerror(0) == b"ok" and b"invalid" in L.rsx_strerror(-1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from recsys_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RsxError):
        _lib.lib()


def test_no_gpu_means_no_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from recsys_amd import _lib
    from recsys_amd.ops import EmbeddingArena
    with pytest.raises(_lib.RsxError):
        EmbeddingArena(np.array([0, 4]), 16, 8)


def test_fingerprint64_matches_oracle_all_lengths(L):
    rng = np.random.default_rng(0)
    strs = [b"a", b"b", b"c", b"d", b"", b"NULL", b"05db9164"] + \
           [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in list(range(0, 140)) + [200, 255, 256, 257, 1000]]
    buf = np.frombuffer(b"".join(strs), np.uint8) if sum(map(len, strs)) else np.zeros(0, np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in strs])]).astype(np.int64)
    out = np.zeros(len(strs), np.uint64)
    assert L.rsx_hash_fp64_h(buf.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), len(strs),
                             out.ctypes.data_as(C.c_void_p)) == 0
    assert [int(x) for x in out] == [hashing.fingerprint64(s) for s in strs]
    assert int(out[0]) == 12917804110809363939          # Appendix B-1 KAT straight through the C ABI
    # TF documentation example of to_hash_bucket_fast(["Hello", "TensorFlow", "2.x"], 3) -> [0, 2, 2], through the C ABI
    for s_, want in ((b"Hello", 0), (b"TensorFlow", 2), (b"2.x", 2)):
        assert int(L.rsx_fingerprint64_h(s_, len(s_))) % 3 == want


def test_bucketize_and_crc_match_oracle(L):
    rng = np.random.default_rng(1)
    x = np.concatenate([np.floor(np.exp(rng.normal(2, 2, 500))), [0, 1, 2, 6, 20, 1000, 1e6, np.nan, -1, -3.5]]).astype(np.float32)
    for j, b in enumerate(criteo.CONT_BOUNDARIES):
        bd = np.asarray(b, np.float32)
        shift = 4.0 if j == 1 else 1.0
        out = np.zeros(len(x), np.int32)
        assert L.rsx_bucketize_log_h(x.ctypes.data_as(C.c_void_p), len(x), bd.ctypes.data_as(C.c_void_p), len(bd),
                                     shift, out.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(out, criteo.bucketize(x, b, shift))
    for n in [0, 1, 7, 8, 9, 63, 64, 1000]:
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        arr = np.frombuffer(d, np.uint8) if n else np.zeros(0, np.uint8)
        assert L.rsx_crc32c_h(arr.ctypes.data_as(C.c_void_p), n) == tfrecord.crc32c(d)
        assert L.rsx_masked_crc32c_h(arr.ctypes.data_as(C.c_void_p), n) == tfrecord.masked_crc(d)
    s = np.frombuffer(b"123456789", np.uint8)
    assert L.rsx_crc32c_h(s.ctypes.data_as(C.c_void_p), 9) == 0xE3069283


def test_layout_transform_matches_oracle(L):
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    lin, emb = build_feature_columns(16)
    lay = CriteoLayout.from_columns(emb)
    assert np.array_equal(lay.row_off, criteo.row_offsets())
    assert [c.key for c in lay.columns] == [c["src"] for c in criteo.field_table()]
    rng = np.random.default_rng(2)
    B = 33
    cont = np.floor(np.exp(rng.normal(2, 2, (B, 13)))).astype(np.float32)
    cont[:, 1] -= 3                                        # _c2 goes down to -3 (dcn/readme.md:7) -> log(x+4)
    cat = [[(b"NULL" if rng.random() < 0.1 else ("%08x" % rng.integers(0, 1 << 32)).encode()) for _ in range(26)]
           for _ in range(B)]
    flat = [v for row in cat for v in row]
    buf = np.frombuffer(b"".join(flat), np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(v) for v in flat])]).astype(np.int64)
    ids = lay.transform(cont, buf, offs)
    assert np.array_equal(ids, criteo.transform_batch(cont, cat, c2_shift=4.0))


def test_launch_form_registry(monkeypatch):
    """recsys_amd/_lib.py FORMS: the launch-form toggles live behind ONE environment variable; unknown names fail loudly."""
    from recsys_amd import _lib
    monkeypatch.delenv("RSX_FORMS", raising=False)
    assert _lib.form("fm_fuse") == "1" and _lib.form("SORT_RIDE_MAX") == "2048"
    monkeypatch.setenv("RSX_FORMS", "fm_fuse=0, Sort_Ride_Max=0")
    assert _lib.form("fm_fuse") == "0" and _lib.form("sort_ride_max") == "0" and _lib.form("mlp_fuse") == "1"
    monkeypatch.setenv("RSX_FORMS", "no_such_form=1")
    with pytest.raises(_lib.RsxError):
        _lib.form("fm_fuse")
    with pytest.raises(KeyError):
        monkeypatch.delenv("RSX_FORMS")
        _lib.form("no_such_form")
