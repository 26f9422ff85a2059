"""Criteo-39 feature columns: the product-side mirror of `build_feature_columns`
(fm/fm.py:47-97, xdeepfm/xdeepfm.py:44-94, dcn/dcn.py:49-99).

The reference builds tf.feature_column objects; here a column is a small descriptor and
`input_layer` semantics (name-sorted slot order) are resolved once into a
`CriteoLayout`: slot order, row offsets into the concatenated table, and the host transform
raw features -> table-local ids (FarmHash % bucket, bucketize(log(x+shift))) done by librsx.so's
host functions.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from ._lib import check, lib

# fm/fm.py:53-67
BUCKETS_CONT = [
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
# fm/fm.py:72-73 (the capped sizes; :69-70 are overridden)
BUCKETS_CAT = [1460, 583, 100000, 100000, 305, 23, 12517, 633, 3, 93145, 5683, 100000, 3194, 27, 14992, 100000,
               10, 5652, 2172, 3, 100000, 17, 15, 100000, 104, 100000]
CONT_FEATURE = ["_c%d" % i for i in range(1, 14)]
CAT_FEATURE = ["_c%d" % i for i in range(14, 40)]
LABEL = "_c0"


@dataclass
class Column:
    name: str                 # TF column name (sort key of input_layer)
    key: str                  # source feature `_cN`
    kind: str                 # 'numeric' | 'bucketized_embedding' | 'hash_embedding' | 'bucketized_indicator' | 'hash_indicator'
    rows: int = 0
    dimension: int = 0
    boundaries: Optional[List[float]] = None
    log_shift: float = 1.0


def build_feature_columns(embedding_size, linear="indicator_all"):
    """Returns (linear_feature_columns, embedding_feature_columns) like fm/fm.py:47.
    linear = 'indicator_all' (fm.py: 39 indicators, :83,91,94), 'numeric+indicator' (xdeepfm.py:82,91:
    13 numerics + 26 indicators) or 'numeric' (dcn.py:86: 13 numerics, unused by its model_fn)."""
    lin, emb = [], []
    for j, key in enumerate(CONT_FEATURE):
        shift = 4.0 if key == "_c2" else 1.0       # fm/fm.py:77-78 (same in xdeepfm.py, dcn.py)
        b = BUCKETS_CONT[j]
        emb.append(Column(key + "_bucketized_embedding", key, "bucketized_embedding", len(b) + 1, embedding_size, b, shift))
        if linear == "indicator_all":
            lin.append(Column(key + "_bucketized_indicator", key, "bucketized_indicator", len(b) + 1, 0, b, shift))
        else:
            lin.append(Column(key, key, "numeric", 1, 1, None, shift))
    for j, key in enumerate(CAT_FEATURE):
        emb.append(Column(key + "_embedding", key, "hash_embedding", BUCKETS_CAT[j], embedding_size))
        if linear != "numeric":
            lin.append(Column(key + "_indicator", key, "hash_indicator", BUCKETS_CAT[j]))
    return lin, emb


def build_model_columns(embedding_size):
    """deepfm/deepfm.py:37-51 AS COMMITTED: two int64 id features hashed into 500 000 / 100 000 buckets
    (`categorical_column_with_hash_bucket(..., dtype=int64)`: the key is hashed as its decimal string), each with an
    indicator (linear) and an embedding column.  input_layer's name order puts i_id before u_id."""
    lin, emb = [], []
    for key, rows in (("u_id", 500000), ("i_id", 100000)):
        lin.append(Column(key + "_indicator", key, "hash_indicator", rows))
        emb.append(Column(key + "_embedding", key, "hash_embedding", rows, embedding_size))
    return lin, emb


@dataclass
class CriteoLayout:
    """input_layer's name-sorted slot order resolved for the concatenated table."""
    columns: List[Column]
    row_off: np.ndarray = field(default=None)

    @classmethod
    def from_columns(cls, embedding_feature_columns):
        cols = sorted(embedding_feature_columns, key=lambda c: c.name)
        off = np.concatenate([[0], np.cumsum([c.rows for c in cols])]).astype(np.int64)
        return cls(cols, off)

    @property
    def F(self):
        return len(self.columns)

    def field_mask(self, keys):
        """bitmask over slots whose source key is in `keys` (first-order indicator fields)."""
        m = 0
        for i, c in enumerate(self.columns):
            if c.key in keys:
                m |= 1 << i
        return m

    def transform_int64(self, features):
        """Host transform for int64-keyed hash columns (deepfm/deepfm.py:41,46): features[key] int64 [B] (or [B,1]) ->
        ids [B,F] int32 in slot order, id = Fingerprint64(str(key)) % rows."""
        L = lib()
        B = int(np.asarray(features[self.columns[0].key]).reshape(-1).shape[0])
        ids = np.empty((B, self.F), np.int32)
        tmp = np.empty(B, np.int32)
        for slot, c in enumerate(self.columns):
            keys = np.ascontiguousarray(np.asarray(features[c.key]).reshape(-1), np.int64)
            check(L.rsx_hash_int64_keys_h(keys.ctypes.data_as(C.c_void_p), B, c.rows, tmp.ctypes.data_as(C.c_void_p)),
                  "rsx_hash_int64_keys_h")
            ids[:, slot] = tmp
        return ids

    def transform(self, cont, cat_bytes, cat_offs):
        """Host transform of one batch.  cont [B,13] float32 raw _c1.._c13; categorical values as one
        concatenated byte buffer with offsets [B*26+1] in (b major, _c14.._c39 minor) order.
        -> ids [B,F] int32 in slot order.  Hash + bucketize run in librsx.so (host functions)."""
        L = lib()
        cont = np.ascontiguousarray(cont, np.float32)
        B = cont.shape[0]
        ids = np.empty((B, self.F), np.int32)
        h = np.empty(B * 26, np.uint64)
        cat_bytes = np.ascontiguousarray(cat_bytes, np.uint8)
        cat_offs = np.ascontiguousarray(cat_offs, np.int64)
        check(L.rsx_hash_fp64_h(cat_bytes.ctypes.data_as(C.c_void_p), cat_offs.ctypes.data_as(C.c_void_p), B * 26,
                                h.ctypes.data_as(C.c_void_p)), "rsx_hash_fp64_h")
        h = h.reshape(B, 26)
        tmp = np.empty(B, np.int32)
        for slot, c in enumerate(self.columns):
            j = int(c.key[2:])
            if c.boundaries is not None:
                x = np.ascontiguousarray(cont[:, j - 1])
                bd = np.asarray(c.boundaries, np.float32)
                check(L.rsx_bucketize_log_h(x.ctypes.data_as(C.c_void_p), B, bd.ctypes.data_as(C.c_void_p), len(bd),
                                            c.log_shift, tmp.ctypes.data_as(C.c_void_p)), "rsx_bucketize_log_h")
                ids[:, slot] = tmp
            else:
                ids[:, slot] = (h[:, j - 14] % np.uint64(c.rows)).astype(np.int32)
        return ids
