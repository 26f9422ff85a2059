"""Oracle: Criteo-39 feature transforms (schema, slot order, bucketize, hashing).

Follows fm/fm.py:39-97 `build_feature_columns` (twins xdeepfm/xdeepfm.py:36-94,
dcn/dcn.py:41-99) and the TF feature_column semantics of SURVEY.md Appendix
A-1 (name-sorted column order), A-2 (hash) and A-3 (bucketize the log value).
Pinned by SURVEY.md Appendix B-3 / B-5 in tests/test_oracle_kats.py.

Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np

from .hashing import hash_bucket

# fm/fm.py:53-67 -- boundaries (raw-value quantiles, applied to the LOG value, A-3)
CONT_BOUNDARIES = [
    [0.0, 1.0, 2.0, 3.0, 5.0, 12.0],
    [0.0, 1.0, 2.0, 4.0, 10.0, 28.0, 76.0, 301.0],
    [1.0, 2.0, 3.0, 5.0, 7.0, 10.0, 16.0, 24.0, 54.0],
    [1.0, 2.0, 3.0, 5.0, 6.0, 9.0, 13.0, 20.0],
    [20.0, 155.0, 1087.0, 1612.0, 2936.0, 5064.0, 8622.0, 16966.0, 39157.0],
    [3.0, 7.0, 13.0, 24.0, 36.0, 53.0, 85.0, 154.0, 411.0],
    [0.0, 1.0, 2.0, 4.0, 6.0, 10.0, 17.0, 43.0],
    [1.0, 2.0, 4.0, 6.0, 8.0, 12.0, 17.0, 25.0, 37.0],
    [4.0, 8.0, 16.0, 28.0, 41.0, 63.0, 109.0, 147.0, 321.0],
    [0.0, 1.0, 2.0],
    [0.0, 1.0, 2.0, 3.0, 4.0, 8.0],
    [0.0, 1.0, 2.0],
    [1.0, 2.0, 3.0, 5.0, 7.0, 10.0, 14.0, 22.0],
]
# fm/fm.py:72-73 -- the capped hash sizes that override :69-70
CAT_BUCKETS = [1460, 583, 100000, 100000, 305, 23, 12517, 633, 3, 93145, 5683, 100000, 3194, 27, 14992, 100000,
               10, 5652, 2172, 3, 100000, 17, 15, 100000, 104, 100000]
CONT_NAMES = ["_c%d" % i for i in range(1, 14)]
CAT_NAMES = ["_c%d" % i for i in range(14, 40)]
NUM_FIELDS = 39


def field_table():
    """The 39 embedding columns in TF's name-sorted slot order (Appendix A-1).

    Returns list of dicts {name, kind, src, rows, log_shift, boundaries}."""
    cols = []
    for j, n in enumerate(CONT_NAMES):
        cols.append(dict(name=n + "_bucketized_embedding", kind="cont", src=n, rows=len(CONT_BOUNDARIES[j]) + 1,
                         boundaries=CONT_BOUNDARIES[j]))
    for j, n in enumerate(CAT_NAMES):
        cols.append(dict(name=n + "_embedding", kind="cat", src=n, rows=CAT_BUCKETS[j], boundaries=None))
    cols.sort(key=lambda c: c["name"])
    return cols


def row_offsets():
    rows = [c["rows"] for c in field_table()]
    return np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)


def bucketize(x, boundaries, log_shift=1.0):
    """fm/fm.py:76-79: v = log(x + shift) in fp32, idx = #boundaries <= v (upper_bound).
    NaN -> len(boundaries) (all comparisons false in TF's std::upper_bound)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        v = np.log(np.asarray(x, np.float32) + np.float32(log_shift)).astype(np.float32)
    b = np.asarray(boundaries, np.float32)
    idx = np.searchsorted(b, v, side="right")
    idx = np.where(np.isnan(v), len(b), idx)
    return idx.astype(np.int64)


def transform_batch(cont, cat, c2_shift=1.0):
    """cont: [B,13] float32 raw values (_c1.._c13); cat: list of B lists of 26 bytes values.
    Returns ids [B,39] int32 (table-local, slot order).  c2_shift=4.0 reproduces fm.py:77-78."""
    cols = field_table()
    B = len(cont)
    ids = np.zeros((B, NUM_FIELDS), np.int32)
    for slot, c in enumerate(cols):
        j = int(c["src"][2:])
        if c["kind"] == "cont":
            shift = c2_shift if c["src"] == "_c2" else 1.0
            ids[:, slot] = bucketize(np.asarray(cont)[:, j - 1], c["boundaries"], shift)
        else:
            ids[:, slot] = [hash_bucket(cat[b][j - 14], c["rows"]) for b in range(B)]
    return ids
