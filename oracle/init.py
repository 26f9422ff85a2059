"""Oracle: parameter initialisers with the reference's TF-1.x distributions.

Seeds are numpy's, so values never equal TF's; parity tests share explicit weights instead.
Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np

from .criteo import CAT_BUCKETS, field_table, row_offsets


def trunc_normal(rng, shape, std, dtype):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2
    return (x * std).astype(dtype)


def glorot_uniform(rng, fan_in, fan_out, shape, dtype):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, shape).astype(dtype)


def glorot_normal(rng, fan_in, fan_out, shape, dtype):
    std = np.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    return trunc_normal(rng, shape, std, dtype)


def _tower(P, rng, pre, d_in, layers, dtype, final_unit=True):
    d = d_in
    for i, n in enumerate(layers):
        P[f"{pre}.W{i}"] = glorot_uniform(rng, d, n, (d, n), dtype)
        P[f"{pre}.b{i}"] = np.zeros(n, dtype)
        P[f"{pre}.gamma{i}"] = np.ones(n, dtype)
        P[f"{pre}.beta{i}"] = np.zeros(n, dtype)
        d = n
    if final_unit:
        P[f"{pre}.Wout"] = glorot_uniform(rng, d, 1, (d, 1), dtype)
        P[f"{pre}.bout"] = np.zeros(1, dtype)
    return d


def criteo_tables(rng, D, dtype, rows=None):
    R = int(row_offsets()[-1]) if rows is None else rows
    return trunc_normal(rng, (R, D), 1.0 / np.sqrt(D), dtype)


def deepfm_params(seed=0, D=16, layers=(100, 100), dtype=np.float32, row_off=None, with_dnn=True):
    rng = np.random.default_rng(seed)
    row_off = row_offsets() if row_off is None else row_off
    R, F = int(row_off[-1]), len(row_off) - 1
    P = {"tables": criteo_tables(rng, D, dtype, R),
         "w1": glorot_uniform(rng, R, 1, (R,), dtype), "b1": np.zeros(1, dtype)}
    n_in = 2
    if with_dnn:
        _tower(P, rng, "dnn", F * D, layers, dtype)
        n_in = 3
    P["out.W"] = glorot_uniform(rng, n_in, 1, (n_in, 1), dtype)
    P["out.b"] = np.zeros(1, dtype)
    return P


def dcn_params(seed=0, D=16, layers=(100, 100), cross_layers=3, dtype=np.float32, row_off=None):
    rng = np.random.default_rng(seed)
    row_off = row_offsets() if row_off is None else row_off
    R, F = int(row_off[-1]), len(row_off) - 1
    dim = F * D
    P = {"tables": criteo_tables(rng, D, dtype, R)}
    P["cross.W"] = glorot_normal(rng, dim, dim, (cross_layers, dim), dtype)
    P["cross.b"] = glorot_normal(rng, dim, dim, (cross_layers, dim), dtype)   # dcn/dcn.py:140: bias is glorot too
    last = _tower(P, rng, "dnn", dim, layers, dtype, final_unit=False)
    P["out.W"] = glorot_uniform(rng, last + dim, 1, (last + dim, 1), dtype)
    P["out.b"] = np.zeros(1, dtype)
    return P


def xdeepfm_layout():
    """(cat_slot [26], cat_off [27]) for the linear one-hot blocks of _c14.._c39."""
    cols = field_table()
    slot_of = {c["src"]: i for i, c in enumerate(cols)}
    cat_slot = np.array([slot_of["_c%d" % j] for j in range(14, 40)], np.int64)
    cat_off = np.concatenate([[0], np.cumsum(CAT_BUCKETS)]).astype(np.int64)
    return cat_slot, cat_off


def xdeepfm_params(seed=0, D=16, layers=(100, 100), cin=(128, 128), dtype=np.float32, row_off=None, cat_rows=None):
    rng = np.random.default_rng(seed)
    row_off = row_offsets() if row_off is None else row_off
    R, F = int(row_off[-1]), len(row_off) - 1
    Rc = int(sum(CAT_BUCKETS)) if cat_rows is None else cat_rows
    P = {"tables": criteo_tables(rng, D, dtype, R), "tables2": criteo_tables(rng, D, dtype, R)}
    nlin = 13 + Rc
    P["lin.wnum"] = glorot_uniform(rng, nlin, 1, (13,), dtype)
    P["lin.wcat"] = glorot_uniform(rng, nlin, 1, (Rc,), dtype)
    P["lin.b"] = np.zeros(1, dtype)
    H = F
    for k, n in enumerate(cin):
        P[f"cin.W{k}"] = glorot_uniform(rng, F * H, n, (F * H, n), dtype)
        P[f"cin.c{k}"] = np.zeros(n, dtype)
        H = n
    tot = int(sum(cin))
    P["cin.Wout"] = glorot_uniform(rng, tot, 1, (tot, 1), dtype)
    P["cin.bout"] = np.zeros(1, dtype)
    _tower(P, rng, "dnn", F * D, layers, dtype)
    P["out.W"] = glorot_uniform(rng, 3, 1, (3, 1), dtype)
    P["out.b"] = np.zeros(1, dtype)
    return P


def din_params(seed=0, K=32, n_item=63002, n_cate=802, dtype=np.float32):
    rng = np.random.default_rng(seed)
    P = {"item_bias": np.zeros(n_item, dtype),
         "item_emb": glorot_normal(rng, n_item, K, (n_item, K), dtype),
         "cate_emb": glorot_normal(rng, n_cate, K, (n_cate, K), dtype)}
    for pre in ("att_i", "att_c"):
        d = 4 * K
        for i, n in enumerate((80, 40, 1)):
            P[f"{pre}.W{i}"] = glorot_uniform(rng, d, n, (d, n), dtype)
            P[f"{pre}.b{i}"] = np.zeros(n, dtype)
            d = n
    d = 3 * K
    for i, n in enumerate((100, 50, 20)):
        P[f"mlp.W{i}"] = glorot_uniform(rng, d, n, (d, n), dtype)
        P[f"mlp.b{i}"] = np.zeros(n, dtype)
        d = n
    P["mlp.Wout"] = glorot_uniform(rng, d, 1, (d, 1), dtype)
    P["mlp.bout"] = np.zeros(1, dtype)
    return P
