"""Seeded synthetic Criteo-39 / DIN batches: the only sample shard of the
reference is a missing blob, so benchmarks and end-to-end tests run on data of the same schema:
label ~ Bernoulli(0.25); _c1.._c13 = floor(exp(N(2,2))) clipped to [0,1e6] (fp32, >= 0 so the log is finite;
_c2 additionally shifted down to -3, dcn/readme.md:7); _c14.._c39 = 8-hex-char strings drawn Zipf(1.05)
over the REAL cardinalities (fm/fm.py:69-70), 5 % absent -> the 'NULL' default (fm/fm.py:44).
"""
import numpy as np

SEED = 20190625
# fm/fm.py:69-70: uncapped cardinalities of _c14.._c39
CARDINALITIES = [1460, 583, 10131226, 2202607, 305, 23, 12517, 633, 3, 93145, 5683, 8351592, 3194, 27, 14992, 5461305,
                 10, 5652, 2172, 3, 7046546, 17, 15, 286180, 104, 142571]


def _mix(v):
    v = (v.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    v = (v ^ (v >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    v = (v ^ (v >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return v ^ (v >> np.uint64(31))


def criteo_raw_batch(rng, B, null_frac=0.05, zipf_a=1.05):
    """-> label [B] f32, cont [B,13] f32, cat: list of B lists of 26 bytes objects."""
    label = (rng.random(B) < 0.25).astype(np.float32)
    cont = np.clip(np.floor(np.exp(rng.normal(2.0, 2.0, (B, 13)))), 0, 1e6).astype(np.float32)
    cont[:, 1] = np.maximum(cont[:, 1] - 3.0, -3.0)              # _c2 has negatives down to -3
    cat = []
    vals = np.empty((B, 26), np.uint64)
    for j, card in enumerate(CARDINALITIES):
        z = rng.zipf(zipf_a, B).astype(np.uint64) % np.uint64(card)
        vals[:, j] = _mix(z + np.uint64(j) * np.uint64(1 << 40)) & np.uint64(0xFFFFFFFF)
    null = rng.random((B, 26)) < null_frac
    for b in range(B):
        cat.append([b"NULL" if null[b, j] else b"%08x" % int(vals[b, j]) for j in range(26)])
    return label, cont, cat


def pack_cat(cat):
    flat = [v for row in cat for v in row]
    buf = np.frombuffer(b"".join(flat), np.uint8).copy()
    offs = np.concatenate([[0], np.cumsum([len(v) for v in flat])]).astype(np.int64)
    return buf, offs


def criteo_id_batches(layout, n_batches, B, seed=SEED):
    """Pre-hashed fast path: list of (ids int32 [B,39], labels f32 [B], cont_log f32 [B,13]) on the host,
    produced by the product's own host transform (FarmHash % bucket, bucketize(log))."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_batches):
        label, cont, cat = criteo_raw_batch(rng, B)
        buf, offs = pack_cat(cat)
        ids = layout.transform(cont, buf, offs)
        shift = np.ones(13, np.float32)
        shift[1] = 4.0
        with np.errstate(invalid="ignore", divide="ignore"):
            cont_log = np.log(cont + shift[None, :]).astype(np.float32)
        out.append((ids, label, cont_log))
    return out


def din_batch(rng, B, P=100, n_item=63002, n_cate=802, zipf_a=1.05):
    """DIN batch: i_id ~ Zipf over [1, n_item-1], i_cate = cate_of[i_id]; ragged histories
    left-aligned and zero padded to P (din/din.py:56-57); label ~ Bernoulli(0.5)."""
    cate_of = (_mix(np.arange(n_item, dtype=np.uint64)) % np.uint64(n_cate - 1)).astype(np.int64) + 1
    cate_of[0] = 0
    draw = lambda n: (rng.zipf(zipf_a, n).astype(np.int64) * 2654435761 % (n_item - 1)) + 1
    i_id = draw(B)
    hist = draw(B * P).reshape(B, P)
    lens = rng.integers(1, P + 1, B)
    hist[np.arange(P)[None, :] >= lens[:, None]] = 0
    return dict(i_id=i_id, i_cate=cate_of[i_id], u_iid_seq=hist, u_icat_seq=cate_of[hist],
                label=(rng.random(B) < 0.5).astype(np.int64))
