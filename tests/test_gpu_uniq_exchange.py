"""Op-level tests of the data-parallel unique-list exchange (csrc/uniq_exchange.hip), through the C ABI:

  rsx_uniq_pack          key blocks                      BIT-EXACT against oracle.exchange.pack_keys
  rsx_uniq_merge         global lists / slot map / src   BIT-EXACT against oracle.exchange.concat_unique (np.unique over the
                                                         concatenated per-replica lists: fm/fm.py:162-163,184-194, A-4 / A-12)
  rsx_segsum_bwd_packed  a replica's (row, sum) block    bit-exact vs the oracle's ascending-pair-order sums on segments of <= 2 entries,
                                                         2e-6 of sum|terms| vs the fp64 sums everywhere
  rsx_merged_adam_rows   the replica sum + touched rows  the SUMMED GRADIENT is read back out of the first-step moments
                                                         (m = (1 - beta1) g from zero state) and held BIT-EXACT against the
                                                         rank-ordered fp32 sum of the blocks and to 2e-6 RELATIVE against the fp64
                                                         sum of the oracle's concatenated IndexedSlices -- a dropped 1/N, a missing
                                                         rank or a mis-strided block fails here (TF-1 Adam's first update
                                                         lr * sign(g) hides gradient MAGNITUDES from trajectory tests)
"""
import numpy as np
import pytest

from oracle import criteo, exchange, models, nn
from tests.parity_util import synth_ids

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _row_off(rows):
    return criteo.row_offsets() if rows is None else np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)


def _arena(row_off, D, cap, rng, with_w1=True):
    from recsys_amd.ops import EmbeddingArena
    R = int(row_off[-1])
    tables = (rng.standard_normal((R, D)) * 0.25).astype(np.float32)
    w1 = (rng.standard_normal(R) * 0.1).astype(np.float32)
    a = EmbeddingArena(row_off, D, cap, "cuda", with_w1=with_w1, tables=tables, w1=w1 if with_w1 else None)
    return a, tables, w1


def _enable(a, N, b, parts, monkeypatch):
    if parts:
        monkeypatch.setenv("RSX_UX_PARTS", str(parts))
    else:
        monkeypatch.delenv("RSX_UX_PARTS", raising=False)
    ux = a.enable_unique_exchange(N, b)
    assert not parts or ux.parts == parts
    goff = exchange.goff_caps(a.row_off_np, b)
    assert np.array_equal(ux.goff_np, goff) and ux.KS == exchange.key_block_ints(a.F, ux.parts, goff[-1])
    return ux, goff


def _crafted_lists(rng, row_off, goff, N, P, flavour):
    """Per replica and field an ascending duplicate-free list of global rows, at most cap_f long; the flavours plant the cases
    the merge has to survive: empty lists, lists AT cap, the first / last row of a field, rows either side of every part
    boundary, one replica holding everything while the others hold nothing."""
    F = len(row_off) - 1
    out = []
    for r in range(N):
        lists = []
        for f in range(F):
            rows = int(row_off[f + 1] - row_off[f])
            cap = int(goff[f + 1] - goff[f])
            rpp = exchange.rows_per_part(rows, P)
            if flavour == "edges":
                mode = (r + f) % 4
                if mode == 0:
                    n = 0                                              # empty list
                elif mode == 1:
                    n = cap                                            # list at cap
                else:
                    n = int(rng.integers(1, cap + 1))
                sel = set(rng.choice(rows, size=n, replace=False).tolist()) if n else set()
                if mode >= 2:
                    must = [0, rows - 1] + [x for p in range(1, P) for x in (p * rpp - 1, p * rpp) if 0 <= x < rows]
                    for x in must:
                        if len(sel) < cap or x in sel:
                            sel.add(x)
                    while len(sel) > cap:
                        sel.pop()
                loc = np.array(sorted(sel), np.int64)
            elif flavour == "one_rank":
                loc = np.sort(rng.choice(rows, size=cap, replace=False)) if r == N - 1 else np.zeros(0, np.int64)
            else:                                                       # skewed overlap: most replicas share the head rows
                n = int(rng.integers(0, cap + 1))
                p = 1.0 / (1.0 + np.arange(rows)) ** 1.05
                loc = np.sort(rng.choice(rows, size=n, replace=False, p=p / p.sum())) if rows < 200000 else \
                    np.unique(np.minimum((rng.pareto(1.05, n) * 3).astype(np.int64), rows - 1))
            lists.append(loc.astype(np.int64) + int(row_off[f]))
        out.append(lists)
    return out


def _check_merge(a, ux, rank_lists, k):
    """All k window positions against np.unique over the concatenated lists -- every index output, bit for bit."""
    F, st, R, N = a.F, a.stride, a.R, ux.world
    bufs = a.window_bufs(k)
    for i in range(k):
        want_u, want_n, want_slot, want_src = exchange.concat_unique([rl for rl in rank_lists[i]], st, R)
        nu = bufs[i]["nuniq"].cpu().numpy()
        assert np.array_equal(nu, want_n), (i, nu, want_n)
        got_u = bufs[i]["uniq_row"].cpu().numpy().reshape(F, st)
        got_src = ux.src[i].cpu().numpy().reshape(N, F, st)
        for f in range(F):
            n = int(want_n[f])
            assert np.array_equal(got_u[f, :n], want_u[f, :n]), (i, f)
            assert np.array_equal(got_src[:, f, :n], want_src[:, f, :n]), (i, f)
        assert np.array_equal(bufs[i]["slot"].cpu().numpy()[:R], want_slot), i


MERGE_CASES = [
    # rows, N, b, parts (0 = the product's own choice), jobs
    (None, 2, 256, 0, 1),                                   # deepfm.py as 2 x 256 (parts 1)
    (None, 8, 256, 0, 3),                                   # 8 x 256, a window of 3
    ((3, 7, 40, 11, 1), 3, 6, 1, 2),                        # every list at cap; a one-row field
    ((5, 1000, 33, 64, 100000), 3, 300, 3, 1),              # world 3, parts 3
    ((5, 1000, 33, 64, 100000), 8, 300, 5, 2),              # parts 5: part boundaries inside 32-row words' neighbours
    ((97, 4000, 31, 32, 640000), 8, 512, 64, 1),            # parts 64 = RSX_UNIQ_MAX_PARTS, the largest field the merge admits
    (None, 8, 4096, 0, 1),                                  # dcn.py as 8 x 4 096 (the product picks 4 parts)
    ((63003, 803), 8, 103424, 0, 1),                        # din.py: ONE 63 003-row field, 8 x (1 024 x 101) entries (62 parts)
]


@pytest.mark.parametrize("rows,N,b,parts,k", MERGE_CASES)
@pytest.mark.parametrize("flavour", ["edges", "zipf", "one_rank"])
def test_uniq_merge_is_exact(rows, N, b, parts, k, flavour, monkeypatch):
    rng = np.random.default_rng(1000 * N + b + (parts or 0))
    row_off = _row_off(rows)
    a, _, _ = _arena(row_off, 16, N * b, rng, with_w1=False)
    ux, goff = _enable(a, N, b, parts, monkeypatch)
    for rep in range(2):                 # the second round reuses the workspaces: stale slot entries must be gone
        rank_lists = [_crafted_lists(rng, row_off, goff, N, ux.parts, flavour if rep == 0 else "zipf") for _ in range(k)]
        keys = np.stack([np.concatenate([exchange.pack_keys(rank_lists[i][r], row_off, goff, ux.parts) for i in range(k)])
                         for r in range(N)])
        a.ux_merge(torch.from_numpy(keys).cuda(), k)
        torch.cuda.synchronize()
        _check_merge(a, ux, rank_lists, k)


@pytest.mark.parametrize("rows,N,b,parts,k", [(None, 2, 256, 0, 1), (None, 8, 256, 0, 8), ((3, 7, 40, 11, 1), 3, 6, 1, 2),
                                              ((5, 1000, 33, 64, 100000), 3, 300, 3, 4), ((97, 4000, 31, 32, 640000), 8, 512, 64, 1),
                                              (None, 4, 4096, 0, 4), ((63003, 803), 2, 20000, 0, 1)])
def test_uniq_pack_is_exact_and_feeds_the_merge(rows, N, b, parts, k, monkeypatch):
    """Real batches: every replica's dedup sort + rsx_uniq_pack against the numpy key block, then the merge of the N GPU-made
    blocks against np.unique -- the ids phase of a step end to end."""
    rng = np.random.default_rng(77 + N + b)
    row_off = _row_off(rows)
    a, _, _ = _arena(row_off, 16, N * b, rng, with_w1=False)
    ux, goff = _enable(a, N, b, parts, monkeypatch)
    ids = [[synth_ids(rng, b, row_off) for _ in range(k)] for _ in range(N)]
    locals_ = [(ux.local, ux.keys)] + [ux.new_local() for _ in range(N - 1)]
    blocks = []
    for r in range(N):
        ux.local, ux.keys = locals_[r]
        got = a.ux_sort_pack([torch.from_numpy(x).cuda() for x in ids[r]]).cpu().numpy().reshape(k, ux.KS)
        for i in range(k):
            want = exchange.pack_keys(exchange.unique_lists(ids[r][i], row_off), row_off, goff, ux.parts)
            assert np.array_equal(got[i], want), (r, i)
        blocks.append(got.reshape(-1))
    a.ux_merge(torch.from_numpy(np.stack(blocks)).cuda(), k)
    torch.cuda.synchronize()
    _check_merge(a, ux, [[exchange.unique_lists(ids[r][i], row_off) for r in range(N)] for i in range(k)], k)


def _rank_block(a, ux, goff, row_off, tables, ids, dX, gy1, gy2, G_out, gw1_out, fm):
    """One replica: sort + pack + (gather) + rsx_segsum_bwd_packed into its block; checked against the oracle's sums.
    -> (key block, lists, fp64 row sums per field, fp64 first-order sums per field, fp64 sum|terms| per field)."""
    b, F, D = ids.shape[0], a.F, a.D
    idt = torch.from_numpy(ids).cuda()
    keys = a.ux_sort_pack([idt]).reshape(-1).clone()
    S = None
    if fm:
        _, S, _, _ = a.gather(idt, fm=True, first_order=True)
    a.ux_segsum_local(b, S, torch.from_numpy(dX).cuda(), None if gy1 is None else torch.from_numpy(gy1).cuda(),
                      None if gy2 is None else torch.from_numpy(gy2).cuda(), G_out, gw1_out)
    torch.cuda.synchronize()
    rws = ids.astype(np.int64) + row_off[None, :-1]
    dE32 = dX.reshape(b, F, D).copy()
    if fm:
        dE32 = models.fm2_bwd(tables[rws], S.cpu().numpy(), gy2) + dE32          # fp32, the kernel's operation order
    r, v = models._pairs_field_major(rws, dE32)
    uniq, G32 = nn.segment_sum_rows(r, v)
    _, G64 = nn.segment_sum_rows(r, v.astype(np.float64))
    _, A64 = nn.segment_sum_rows(r, np.abs(v).astype(np.float64))
    cnt = np.bincount(np.searchsorted(uniq, r), minlength=len(uniq))
    lists = exchange.unique_lists(ids, row_off)
    assert np.array_equal(np.concatenate(lists), uniq)
    got = G_out.cpu().numpy()
    g64, g1_64, a64 = [], [], []
    if gy1 is not None:
        r1, v1 = models._pairs_field_major(rws, np.repeat(gy1[:, None], F, 1))
        _, W32 = nn.segment_sum_rows(r1, v1)
        _, W64 = nn.segment_sum_rows(r1, v1.astype(np.float64))
        gotw = gw1_out.cpu().numpy()
    o = 0
    for f in range(F):
        n = len(lists[f])
        blk = got[goff[f]:goff[f] + n]
        short = cnt[o:o + n] <= 2         # (<= SEG_SHORT = 16 entries in general; a 'spread' wave cooperates on anything > 2)
        assert np.array_equal(blk[short], G32[o:o + n][short]), f          # ascending-pair order, unfused: the oracle's bits
        assert (np.abs(blk - G64[o:o + n]) <= 2e-6 * A64[o:o + n] + 1e-30).all(), f
        g64.append(G64[o:o + n])
        a64.append(A64[o:o + n])
        if gy1 is not None:
            wb = gotw[goff[f]:goff[f] + n]
            assert np.array_equal(wb[short], W32[o:o + n][short]), f
            assert (np.abs(wb - W64[o:o + n]) <= 2e-6 * np.abs(W64[o:o + n]) + 1e-12).all(), f
            g1_64.append(W64[o:o + n])
        o += n
    return keys, lists, g64, g1_64, a64


@pytest.mark.parametrize("rows,N,b,parts,D,second", [
    (None, 2, 256, 0, 16, False), (None, 8, 256, 0, 16, False), (None, 3, 256, 3, 16, True),
    ((3, 7, 40, 11, 1), 3, 6, 1, 4, False), ((5, 1000, 33, 64, 100000), 8, 300, 5, 32, False), (None, 4, 2048, 0, 16, False)])
def test_segsum_packed_and_merged_adam_sum_the_replicas(rows, N, b, parts, D, second, monkeypatch):
    from recsys_amd.ops import AdamTF1, EmbeddingArena
    rng = np.random.default_rng(5 + N * b + D)
    row_off = _row_off(rows)
    F, R = len(row_off) - 1, int(row_off[-1])
    a, tables, w1 = _arena(row_off, D, N * b, rng)
    ux, goff = _enable(a, N, b, parts, monkeypatch)
    capT = int(goff[-1])
    a2 = None
    if second:               # xdeepfm.py: a second table set looked up with the same ids (one dedup serves both)
        t2 = (rng.standard_normal((R, D)) * 0.25).astype(np.float32)
        a2 = EmbeddingArena(row_off, D, N * b, "cuda", tables=t2)
        a2.share_sort_of(a)
        a2.ux = ux
    nG = 2 if second else 1
    Lp = (nG * capT * D + capT + 3) & ~3
    buf = torch.zeros(N, Lp, device="cuda")
    locals_ = [(ux.local, ux.keys)] + [ux.new_local() for _ in range(N - 1)]
    keys, lists, g64, g64b, w64, a64, a64b = [], [], [], [], [], [], []
    fm = not second
    for r in range(N):
        ux.local, ux.keys = locals_[r]
        ids = synth_ids(rng, b, row_off)
        # positive gradients: the first-order sums and the second set's rows have no cancellation ("relative" = relative to the
        # sum); the FM term g (S - E) does cancel: those are held relative to the sum of the terms' magnitudes
        dX = rng.uniform(0.5e-2, 1.5e-2, (b, F * D)).astype(np.float32)
        gy1 = rng.uniform(0.5e-2, 1.5e-2, b).astype(np.float32)
        gy2 = rng.uniform(0.5e-2, 1.5e-2, b).astype(np.float32) if fm else None
        Gv = buf[r, :capT * D].view(capT, D)
        wv = buf[r, nG * capT * D:nG * capT * D + capT]
        k_, l_, g_, w_, a_ = _rank_block(a, ux, goff, row_off, tables, ids, dX, gy1, gy2, Gv, wv, fm)
        keys.append(k_); lists.append(l_); g64.append(g_); w64.append(w_); a64.append(a_)
        if second:
            dX2 = rng.uniform(0.5e-2, 1.5e-2, (b, F * D)).astype(np.float32)
            _, _, g2_, _, a2_ = _rank_block(a2, ux, goff, row_off, t2, ids, dX2, None, None,
                                            buf[r, capT * D:2 * capT * D].view(capT, D), None, False)
            g64b.append(g2_); a64b.append(a2_)
    a.ux_merge(torch.stack(keys).contiguous(), 1)
    a.select(0)
    opt = AdamTF1(lr=1e-3)
    lr, b1, b2, eps = (np.float32(x) for x in opt.hp)
    st = opt.state.cpu().numpy()
    b1p, b2p = np.float32(st[0]), np.float32(st[1])
    a.ux_merged_adam(buf[0, :capT * D].view(capT, D), buf[0, nG * capT * D:nG * capT * D + capT], Lp, opt, [],
                     second=None if not second else (a2, buf[0, capT * D:2 * capT * D].view(capT, D)))
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    omb1, omb2 = np.float32(1) - b1, np.float32(1) - b2
    alpha = np.float32(lr * np.sqrt(np.float32(1) - b2p) / (np.float32(1) - b1p))
    sets = [(a, tables, 0, g64, a64)] + ([(a2, t2, capT * D, g64b, a64b)] if second else [])
    touched = np.zeros(R, bool)
    dropped_seen = 0
    for ar, t0, off, gsets, asets in sets:
        m, v, var = ar.m_t.cpu().numpy(), ar.v_t.cpu().numpy(), ar.tables.cpu().numpy()
        for f in range(F):
            blocks = [host[r, off:off + capT * D].reshape(capT, D)[goff[f]:goff[f] + len(lists[r][f])] for r in range(N)]
            U, G32 = exchange.replica_sums([lists[r][f] for r in range(N)], blocks, np.float32)          # rank order, fp32
            _, G64 = exchange.replica_sums([lists[r][f] for r in range(N)], [gsets[r][f] for r in range(N)])     # the oracle's sum
            _, A64 = exchange.replica_sums([lists[r][f] for r in range(N)], [asets[r][f] for r in range(N)])     # sum of |terms|
            touched[U] = True
            m_want = G32 * omb1
            assert np.array_equal(m[U], m_want), (f, "m = (1 - beta1) * rank-ordered replica sum, bit for bit")
            assert np.array_equal(v[U], (G32 * G32) * omb2), f
            # the summed gradient read back out of m, against the fp64 sum of the concatenated IndexedSlices: 2e-6 RELATIVE (to
            # the sum of the terms' magnitudes = the sum itself where nothing cancels)
            g_back = m[U].astype(np.float64) / float(omb1)
            assert (np.abs(g_back - G64) <= 2e-6 * A64).all(), f
            if not fm:
                assert (np.abs(g_back - G64) <= 2e-6 * np.abs(G64)).all(), f
            if N > 1:        # the test SEES magnitudes: the sum without the last replica's share is out of tolerance on its rows
                _, Gd = exchange.replica_sums([lists[r][f] for r in range(N - 1)] + [lists[N - 1][f][:0]],
                                              [gsets[r][f] for r in range(N - 1)] + [gsets[N - 1][f][:0]])
                Ud = np.unique(np.concatenate([lists[r][f] for r in range(N - 1)]))
                at = np.searchsorted(U, Ud)
                dropped_seen += int((np.abs(G64[at] - Gd) > 2e-6 * A64[at]).any(axis=-1).sum())
            var_want = t0[U] - (alpha * m_want) / (np.sqrt((G32 * G32) * omb2) + eps)
            np.testing.assert_allclose(var[U], var_want, rtol=0, atol=2e-9)
        un = ~touched
        assert np.array_equal(var[un], t0[un]) and not m[un].any() and not v[un].any()      # no other row moved
    assert N == 1 or dropped_seen > 0
    # first order (tf.layers.dense kernel [R, 1]: ApplyAdam's formula on the touched rows)
    mw, vw, wv_ = a.m_w.cpu().numpy(), a.v_w.cpu().numpy(), a.w1.cpu().numpy()
    wo = nG * capT * D
    for f in range(F):
        blocks = [host[r, wo:wo + capT][goff[f]:goff[f] + len(lists[r][f])] for r in range(N)]
        U, W32 = exchange.replica_sums([lists[r][f] for r in range(N)], blocks, np.float32)
        _, W64 = exchange.replica_sums([lists[r][f] for r in range(N)], [w64[r][f] for r in range(N)])
        assert np.array_equal(mw[U], W32 * omb1), f
        assert np.array_equal(vw[U], (W32 * W32) * omb2), f
        assert (np.abs(mw[U].astype(np.float64) / float(omb1) - W64) <= 2e-6 * np.abs(W64)).all(), f
        np.testing.assert_allclose(wv_[U], w1[U] - ((W32 * omb1) * alpha) / (np.sqrt((W32 * W32) * omb2) + eps), rtol=0, atol=2e-9)
    assert not mw[~touched].any() and np.array_equal(wv_[~touched], w1[~touched])
