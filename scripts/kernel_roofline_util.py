"""HIP-event timing over graph-captured repeats, and Criteo-shaped synthetic ids (shared by the probe scripts; nothing here or
in any other script touches oracle/ -- that is test infrastructure)."""
import numpy as np
import torch


def criteo_row_off():
    """Row offsets of the 39 Criteo fields' tables, from the product's own feature columns (fm/fm.py:30-63)."""
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    return np.asarray(CriteoLayout.from_columns(build_feature_columns(16)[1]).row_off, np.int64)


def synth_ids(rng, B, row_off, zipf_a=1.05):
    """Criteo-shaped ids: Zipf over each field's bucket count, table-local, [B,F] int32."""
    F = len(row_off) - 1
    ids = np.zeros((B, F), np.int32)
    for f in range(F):
        n = int(row_off[f + 1] - row_off[f])
        r = rng.zipf(zipf_a, B).astype(np.int64)
        ids[:, f] = ((r * 2654435761) % n).astype(np.int32) if n > 16 else rng.integers(0, n, B)
    return ids


def timeit(fn, reps=20, inner=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (reps * inner) * 1e3      # us
