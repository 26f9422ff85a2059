"""FarmHash Fingerprint64 pinned from outside the repo, and the two in-repo implementations held against each other at
scale.

* `tests/golden/cityhash_abseil_kats.json`: outputs of Google's own CityHash64 (the abseil copy inside pyarrow's
  libarrow_compute.so on this image) for 481 strings of 0..32 bytes -- where CityHash64 v1.1 and farmhashna::Hash64 are
  the same function (shared HashLen0to16 / HashLen17to32).  That range holds every value the reference hashes: 8-hex
  Criteo categoricals (the 8-16-byte branch, pinned by nothing external before), 'NULL', decimal int64 ids.
* live, when that library is present: 10^6 random strings of 0..32 bytes, product C++ vs abseil.
* 10^6 random strings of 0..100 bytes: product C++ (`rsx_hash_fp64_h`) == Python oracle, every one of them, and the sha256
  of the 10^6 outputs equals the committed digest (the 33-64 and > 64-byte branches have NO external vector: they rest on
  these two independently written implementations agreeing; no Criteo / DIN feature value reaches them)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import hashing
from tests.golden.make_hash_kats import abseil_cityhash64

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DIGEST_1M = "596265e9d4acea67978bf031413fd14c22a19b655106af623f9405a812546eae"   # sha256 of the 10^6 uint64 outputs, little endian


@pytest.fixture(scope="module")
def L():
    from recsys_amd import _lib
    return _lib.lib()


def _cxx(L, raw, offs):
    buf = np.frombuffer(raw, np.uint8) if len(raw) else np.zeros(1, np.uint8)
    offs = np.ascontiguousarray(offs, np.int64)
    out = np.zeros(len(offs) - 1, np.uint64)
    assert L.rsx_hash_fp64_h(buf.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), len(out),
                             out.ctypes.data_as(C.c_void_p)) == 0
    return out


def _random_strings(seed, n, max_len):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, n)
    raw = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8).tobytes()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return raw, offs, lens


def test_external_cityhash_vectors_pin_both_implementations(L):
    kat = json.load(open(os.path.join(GOLD, "cityhash_abseil_kats.json")))
    strs = [bytes.fromhex(h) for h in kat["strings_hex"]]
    want = np.array([int(v) for v in kat["hash"]], np.uint64)
    assert max(map(len, strs)) <= 32 and len(strs) >= 400
    assert {len(s) for s in strs} == set(range(33))                              # every length 0..32 is covered
    assert [hashing.fingerprint64(s) for s in strs] == [int(v) for v in want]    # the oracle
    offs = np.concatenate([[0], np.cumsum([len(s) for s in strs])])
    assert np.array_equal(_cxx(L, b"".join(strs), offs), want)                   # the product
    i = kat["strings_hex"].index(b"05db9164".hex())                              # a Criteo value: the 8-16-byte branch
    assert int(want[i]) == 1602218533055279028 == hashing.fingerprint64(b"05db9164")


def test_product_hash_equals_abseil_cityhash_on_a_million_short_strings(L):
    fn, what = abseil_cityhash64()
    if fn is None:
        pytest.skip(what)
    n = 1_000_000
    raw, offs, _ = _random_strings(7, n, 32)
    got = _cxx(L, raw, offs)
    ext = np.fromiter((fn(raw[offs[i]:offs[i + 1]]) for i in range(n)), np.uint64, n)
    assert np.array_equal(got, ext), "first mismatch at %d" % int(np.argmax(got != ext))


def test_cxx_equals_python_oracle_on_a_million_strings_of_0_to_100_bytes(L):
    n = 1_000_000
    raw, offs, lens = _random_strings(20190625, n, 100)
    got = _cxx(L, raw, offs)
    assert hashlib.sha256(got.tobytes()).hexdigest() == DIGEST_1M
    py = np.fromiter((hashing.fingerprint64(raw[offs[i]:offs[i + 1]]) for i in range(n)), np.uint64, n)
    bad = np.flatnonzero(got != py)
    assert bad.size == 0, "first mismatch at %d (len %d)" % (int(bad[0]), int(lens[bad[0]]))
    # all four length branches (0-16, 17-32, 33-64, > 64 bytes) were exercised in volume
    assert min(np.bincount(np.digitize(lens, [17, 33, 65]), minlength=4)) > 100_000
