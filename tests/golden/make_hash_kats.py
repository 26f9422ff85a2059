#!/usr/bin/env python
"""Generates tests/golden/cityhash_abseil_kats.json: known answers for FarmHash Fingerprint64 on strings of 0..32 bytes
from an EXTERNAL implementation -- Google's own CityHash64 as compiled into the abseil copy that pyarrow's
libarrow_compute.so ships on this image (symbol absl::lts_*::hash_internal::CityHash64(const char*, size_t)).

Why that pins Fingerprint64: farmhashna::Hash64 (= farmhash::Fingerprint64, what tf.string_to_hash_bucket_fast calls,
fm/fm.py:89) and CityHash64 v1.1 share HashLen0to16 and HashLen17to32 verbatim -- for inputs of up to 32 bytes the two
functions are the same function; beyond 32 bytes they differ (HashLen33to64 and the > 64-byte loop were redesigned in
FarmHash), so this source says nothing there.  0..32 bytes covers every value the reference's pipelines hash: the
8-hex-character Criteo categoricals (8-16 branch), the 'NULL' default (4-7 branch), and int64 ids formatted as decimal
strings (<= 20 characters: the 0-16 and 17-32 branches; deepfm/deepfm.py:41,46).

The fixture is data (input bytes + expected 64-bit outputs); neither TensorFlow nor the reference is involved.
usage: python tests/golden/make_hash_kats.py"""
import ctypes as C
import glob
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def abseil_cityhash64():
    """-> (callable(bytes) -> int, description) or (None, reason)."""
    try:
        import pyarrow
    except Exception as e:                                            # pragma: no cover
        return None, "pyarrow not importable: %r" % (e,)
    for lib in sorted(glob.glob(os.path.join(os.path.dirname(pyarrow.__file__), "libarrow*.so*"))):
        m = re.search(rb"_ZN4absl\w*13hash_internal10CityHash64EPKcm", open(lib, "rb").read())
        if m is None:
            continue
        try:                                                          # (libarrow.so only IMPORTS the symbol)
            fn = getattr(C.CDLL(lib), m.group(0).decode())
        except (AttributeError, OSError):
            continue
        fn.restype, fn.argtypes = C.c_uint64, [C.c_char_p, C.c_size_t]
        return (lambda s: int(fn(s, len(s)))), "%s in %s (pyarrow %s)" % (m.group(0).decode(), os.path.basename(lib), pyarrow.__version__)
    return None, "no abseil CityHash64 export found next to pyarrow"


def main():
    fn, what = abseil_cityhash64()
    assert fn is not None, what
    rng = np.random.default_rng(20190625)
    strs = [b"", b"a", b"b", b"c", b"d", b"NULL", b"Hello", b"TensorFlow", b"2.x", b"05db9164", b"68fd1e64", b"7e0ccccf"]
    strs += [("%d" % v).encode() for v in (0, 7, 42, 63001, 499999, 2 ** 31 - 1, 2 ** 63 - 1, -1, -2 ** 63)]
    strs += [("%08x" % int(v)).encode() for v in rng.integers(0, 2 ** 32, 64)]          # Criteo-shaped values
    for n in range(0, 33):
        strs += [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for _ in range(12)]
    out = {"source": what,
           "valid_for": "inputs of 0..32 bytes: CityHash64 v1.1 == farmhashna::Hash64 == farmhash::Fingerprint64 there",
           "strings_hex": [s.hex() for s in strs], "hash": [str(fn(s)) for s in strs]}
    path = os.path.join(HERE, "cityhash_abseil_kats.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote %d vectors to %s (%s)" % (len(strs), path, what))


if __name__ == "__main__":
    main()
