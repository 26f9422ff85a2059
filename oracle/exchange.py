"""Oracle: what tf.distribute.MirroredStrategy hands the optimizer for an embedding variable, restated in numpy.

Reference: fm/fm.py:184-194 (`MirroredStrategy()`), fm/fm.py:162-163 (`AdamOptimizer(...).minimize`), SURVEY Appendix A-4
(every replica's embedding gradient is an IndexedSlices over the UNIQUE rows of ITS batch), A-12 (the replicas' IndexedSlices
are aggregated by concatenating indices / values in replica order) and A-5 (`_apply_sparse_shared` sums duplicated indices
before the update).  The replicas' lists are each ascending and duplicate-free; their concatenation is de-duplicated into the
global ascending list, and the gradient of global row j is the sum over replicas r = 0 .. N-1 (replica order) of the entry
replica r holds for that row, if it holds one.

These functions are the bit-exact checker of the INDEX work of csrc/uniq_exchange.hip (rsx_uniq_pack / rsx_uniq_merge) and the
value checker of its sums (rsx_segsum_bwd_packed, rsx_merged_adam_rows).  The key-block layout (`pack_keys`) is the product's
own wire format (include/rsx.h "exchange of per-rank UNIQUE-ROW lists") restated independently of the kernels.

Test infrastructure only (see oracle/__init__.py)."""
import numpy as np


def rows_per_part(rows, P):
    """Rows owned by one of the P row-range parts of a field (include/rsx.h rsx_uniq_pack: a multiple of 32)."""
    return ((int(rows) + P - 1) // P + 31) & ~31


def unique_lists(ids, row_off):
    """ids int [b, F] table-local -> per field the replica's ascending unique GLOBAL rows (A-4: `unique(ids)`)."""
    ids = np.asarray(ids, np.int64)
    return [np.unique(ids[:, f] + int(row_off[f])) for f in range(ids.shape[1])]


def goff_caps(row_off, b_local):
    """cap_f = min(b_local, rows_f) bounds a replica's unique rows of field f; goff = exclusive prefix of the caps."""
    rows = np.diff(np.asarray(row_off, np.int64))
    caps = np.minimum(rows, int(b_local))
    return np.concatenate([[0], np.cumsum(caps)]).astype(np.int64)


def key_block_ints(F, P, capT):
    return (F + F * (P + 1) + int(capT) + 3) & ~3


def pack_keys(lists, row_off, goff, P):
    """One replica's key block: [nuniq[F] | rstart[F][P + 1] | rows packed at goff (-1 padded to cap_f) | pad to 4 ints].
    rstart[f][p] = first position of the list whose row lies in part p or above; rstart[f][P] = nuniq[f]."""
    F = len(lists)
    capT = int(goff[-1])
    out = np.zeros(key_block_ints(F, P, capT), np.int32)
    base = F + F * (P + 1)
    for f, L in enumerate(lists):
        L = np.asarray(L, np.int64)
        nu = len(L)
        cap = int(goff[f + 1] - goff[f])
        assert nu <= cap and (nu < 2 or (np.diff(L) > 0).all())
        out[f] = nu
        rpp = rows_per_part(int(row_off[f + 1] - row_off[f]), P)
        part = (L - int(row_off[f])) // rpp
        out[F + f * (P + 1):F + (f + 1) * (P + 1)] = np.searchsorted(part, np.arange(P + 1), side="left")
        out[base + goff[f]:base + goff[f] + cap] = -1
        out[base + goff[f]:base + goff[f] + nu] = L
    return out


def concat_unique(rank_lists, stride, R):
    """rank_lists[r][f]: replica r's ascending unique global rows of field f.  -> (uniq_row [F, stride] (first nuniq[f] valid),
    nuniq [F], slot [R] (f * stride + j for listed rows, -1 elsewhere), src [N, F, stride] (position of global row j in
    replica r's list, -1 if absent; valid for j < nuniq[f]))."""
    N, F = len(rank_lists), len(rank_lists[0])
    uniq_row = np.full((F, stride), -1, np.int64)
    nuniq = np.zeros(F, np.int32)
    slot = np.full(R, -1, np.int32)
    src = np.full((N, F, stride), -1, np.int32)
    for f in range(F):
        cat = np.concatenate([np.asarray(rank_lists[r][f], np.int64) for r in range(N)])     # A-12: concatenated
        U = np.unique(cat)                                                                    # A-5: duplicated indices merged
        nuniq[f] = len(U)
        uniq_row[f, :len(U)] = U
        slot[U] = f * stride + np.arange(len(U))
        for r in range(N):
            L = np.asarray(rank_lists[r][f], np.int64)
            src[r, f, np.searchsorted(U, L)] = np.arange(len(L))
    return uniq_row, nuniq, slot, src


def replica_sums(rank_lists_f, rank_vals_f, dtype=np.float64):
    """One field: replica r holds (rows L_r ascending, values V_r [len(L_r), ...]).  -> (U, G): G[j] = sum over replicas in
    REPLICA ORDER of the value replica r holds for row U[j] (a replica that does not hold the row adds nothing)."""
    U = np.unique(np.concatenate([np.asarray(L, np.int64) for L in rank_lists_f]))
    G = np.zeros((len(U),) + tuple(rank_vals_f[0].shape[1:]), dtype)
    for L, V in zip(rank_lists_f, rank_vals_f):
        j = np.searchsorted(U, np.asarray(L, np.int64))
        G[j] = G[j] + np.asarray(V, dtype)[:len(L)]          # (one entry per row and replica: a plain indexed add is exact)
    return U, G
