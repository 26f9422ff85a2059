"""GPU parity tests proper: HIP kernels (through the C ABI) vs the CPU oracle on seeded inputs."""
import numpy as np
import pytest

from oracle import criteo, models, nn
from tests.parity_util import deepfm_parity_run, synth_ids

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _arena(row_off, D, cap, rng, with_w1=True, mask=None):
    from recsys_amd.ops import EmbeddingArena
    R = int(row_off[-1])
    tables = rng.standard_normal((R, D)).astype(np.float32) * 0.25
    w1 = rng.standard_normal(R).astype(np.float32) * 0.1
    a = EmbeddingArena(row_off, D, cap, "cuda", with_w1=with_w1, w1_field_mask=mask, tables=tables, w1=w1 if with_w1 else None)
    return a, tables, w1


@pytest.mark.parametrize("D,B,rows", [(16, 256, None), (16, 1, None), (4, 37, (3, 7, 4, 11, 6)), (8, 130, (5, 1, 9)),
                                      (32, 65, (63, 8)), (64, 9, (3, 3))])
def test_gather_fm_fwd(D, B, rows):
    rng = np.random.default_rng(B + D)
    row_off = criteo.row_offsets() if rows is None else np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    a, tables, w1 = _arena(row_off, D, max(B, 1), rng)
    ids = synth_ids(rng, B, row_off)
    E, S, y1, y2 = a.gather(torch.from_numpy(ids).cuda(), fm=True, first_order=True)
    torch.cuda.synchronize()
    _, Eo, y1o, So, y2o = models.gather_fm_fwd(tables, w1, ids, row_off)
    assert np.array_equal(E.cpu().numpy().reshape(Eo.shape), Eo)           # a gather is bit-exact
    np.testing.assert_allclose(S.cpu().numpy(), So, rtol=1e-5, atol=1e-6)   # tolerance: fp32, 1e-5 (north_star)
    np.testing.assert_allclose(y1.cpu().numpy(), y1o, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y2.cpu().numpy(), y2o, rtol=1e-5, atol=2e-5)


def test_gather_empty_batch_and_masks():
    rng = np.random.default_rng(0)
    row_off = np.array([0, 3, 10, 14], np.int64)
    a, tables, w1 = _arena(row_off, 16, 8, rng, mask=0b101)
    ids = torch.zeros((0, 3), dtype=torch.int32, device="cuda")
    E, S, y1, y2 = a.gather(ids, fm=True, first_order=True)
    assert E.shape == (0, 48)
    ids_np = synth_ids(rng, 8, row_off)
    E, S, y1, y2 = a.gather(torch.from_numpy(ids_np).cuda(), fm=False, first_order=True)
    rows = ids_np + row_off[None, :-1]
    want = w1[rows[:, 0]] + w1[rows[:, 2]]                                   # field 1 masked out of the first-order term
    np.testing.assert_allclose(y1.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    assert S is None and y2 is None


@pytest.mark.parametrize("B,rows", [(256, None), (1, None), (100, (3, 7, 4, 11, 6)), (4096, (3, 1000, 50)), (1000, (2,)),
                                    (600, (3, 1000, 50)), (2048, None), (5000, (1, 65536, 65537, 256, 257)),
                                    (16384, (3, 100000, 257)), (20000, (3, 100000, 257, 1, 70000)), (40000, None),
                                    (65536, (2, 262144, 513))])
def test_field_sort_is_exact(B, rows):
    """Index work is bit-exact: sorted unique rows, segment boundaries, permutation, slot map."""
    rng = np.random.default_rng(B)
    row_off = criteo.row_offsets() if rows is None else np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    a, _, _ = _arena(row_off, 16, B + 3, rng)          # stride != B on purpose
    F = len(row_off) - 1
    for rep in range(2):                               # second call must clear the first call's slot entries
        ids = synth_ids(rng, B, row_off)
        a.field_sort(torch.from_numpy(ids).cuda())
        torch.cuda.synchronize()
        st = a.stride
        perm = a.perm.cpu().numpy().reshape(F, st)
        seg = a.seg_off.cpu().numpy().reshape(F, st + 1)
        uniq = a.uniq_row.cpu().numpy().reshape(F, st)
        nu = a.nuniq.cpu().numpy()
        slot = a.slot.cpu().numpy()[:a.R]
        want_slot = np.full(a.R, -1, np.int64)
        for f in range(F):
            order = np.lexsort((np.arange(B), ids[:, f]))          # by id, then by example index
            assert np.array_equal(perm[f, :B], order)
            u, first = np.unique(ids[order, f], return_index=True)
            assert nu[f] == len(u)
            assert np.array_equal(uniq[f, :len(u)], u + row_off[f])
            assert np.array_equal(seg[f, :len(u)], first) and seg[f, len(u)] == B
            want_slot[u + row_off[f]] = f * st + np.arange(len(u))
        assert np.array_equal(slot, want_slot)


@pytest.mark.parametrize("B,rows", [(4096, None), (1024, (3, 100000, 40)), (3000, (1, 2, 131072, 17)), (8192, (5, 93145, 1460)),
                                    (4096, (100000,) * 3)])
def test_field_sort_split_lists_and_repeat(B, rows):
    """The several-workgroups-per-field sort (sort_device.h field_sort_split_block, 1024 <= B <= 8192): every output exact
    over three calls on fresh ids (slot-map entries of the previous call forgotten, the per-field list counters reset by the
    last range), and the long / huge segment lists of the two-stage segment-sum hold exactly the segments of 17..256 /
    > 256 entries (list ORDER is arrival order: compared as sets)."""
    rng = np.random.default_rng(7 * B)
    row_off = criteo.row_offsets() if rows is None else np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    a, _, _ = _arena(row_off, 16, B + 5, rng)
    F = len(row_off) - 1
    st = a.stride
    for rep in range(3):
        ids = synth_ids(rng, B, row_off)
        if rep == 1:                                   # skew: half of the batch on one id per field
            ids[: B // 2] = ids[0]
        a.field_sort(torch.from_numpy(ids).cuda())
        torch.cuda.synchronize()
        perm = a.perm.cpu().numpy().reshape(F, st)
        seg = a.seg_off.cpu().numpy().reshape(F, st + 1)
        uniq = a.uniq_row.cpu().numpy().reshape(F, st)
        nu = a.nuniq.cpu().numpy()
        slot = a.slot.cpu().numpy()[:a.R]
        sid = a.segid.cpu().numpy()
        nch = (B + 15) // 16
        cnts = sid[F * st: F * st + 2 * F]
        lists = sid[F * st + 2 * F: F * st + 2 * F + F * nch].reshape(F, nch)
        want_slot = np.full(a.R, -1, np.int64)
        for f in range(F):
            order = np.lexsort((np.arange(B), ids[:, f]))
            assert np.array_equal(perm[f, :B], order)
            u, first, counts = np.unique(ids[order, f], return_index=True, return_counts=True)
            assert nu[f] == len(u)
            assert np.array_equal(uniq[f, :len(u)], u + row_off[f])
            assert np.array_equal(seg[f, :len(u)], first) and seg[f, len(u)] == B
            want_slot[u + row_off[f]] = f * st + np.arange(len(u))
            if not a._two_stage(B):                    # (B <= 1024: no two-stage workspace, the sort gets no segid pointer)
                continue
            assert np.array_equal(sid[f * st: f * st + B], np.repeat(np.arange(len(u)), counts))
            long_j = set(np.nonzero((counts > 16) & (counts <= 256))[0].tolist())
            huge_j = set(np.nonzero(counts > 256)[0].tolist())
            assert cnts[f] == len(long_j) and cnts[F + f] == len(huge_j)
            assert set(lists[f, :len(long_j)].tolist()) == long_j
            assert set(lists[f, nch - len(huge_j):].tolist()) == huge_j if huge_j else True
        assert np.array_equal(slot, want_slot)


@pytest.mark.parametrize("B,rows,D", [(256, None, 16), (300, (3, 7, 4, 11, 6), 4), (2048, (2, 500), 16),
                                      (2048, None, 16), (4099, (3, 1000, 50, 100000, 1), 16), (1500, (7, 300), 8),
                                      (16384, (3, 40000), 32), (30000, (3, 100000, 40), 16)])
def test_segsum_bwd(B, rows, D):
    rng = np.random.default_rng(B + 1)
    row_off = criteo.row_offsets() if rows is None else np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    a, tables, w1 = _arena(row_off, D, B, rng)
    F = len(row_off) - 1
    ids = synth_ids(rng, B, row_off)
    dX = rng.standard_normal((B, F * D)).astype(np.float32) * 1e-2
    gy1 = rng.standard_normal(B).astype(np.float32) * 1e-2
    gy2 = rng.standard_normal(B).astype(np.float32) * 1e-2
    idt = torch.from_numpy(ids).cuda()
    a.field_sort(idt)
    E, S, _, _ = a.gather(idt, fm=True, first_order=True)
    a.segsum(B, S, torch.from_numpy(dX).cuda(), torch.from_numpy(gy1).cuda(), torch.from_numpy(gy2).cuda())
    torch.cuda.synchronize()
    rws = ids.astype(np.int64) + row_off[None, :-1]
    So = S.cpu().numpy()                                   # same S on both sides -> the rest is order-exact
    dE = models.fm2_bwd(tables[rws], So, gy2) + dX.reshape(B, F, D)
    r, v = models._pairs_field_major(rws, dE)
    uniq, G = nn.segment_sum_rows(r, v)
    _, g1 = nn.segment_sum_rows(*models._pairs_field_major(rws, np.repeat(gy1[:, None], F, 1)))
    slot = a.slot.cpu().numpy()[:a.R]
    Gg = a.G.cpu().numpy()[slot[uniq]]
    g1g = a.gw1.cpu().numpy()[slot[uniq]]
    assert (slot[uniq] >= 0).all() and (slot >= 0).sum() == len(uniq)
    # short segments (<= 16 examples): sequential ascending-b sums with unfused mul/add = the oracle's operation
    # order -> bit-exact.  Long segments are summed as 16 ordered sub-range partials (deterministic, different
    # rounding): fp32 tolerance.
    cnt = np.bincount(np.searchsorted(uniq, r), minlength=len(uniq))
    short = cnt <= 16
    assert short.any()
    assert np.array_equal(Gg[short], G[short])
    assert np.array_equal(g1g[short], g1[short])
    # (atol: sums of up to thousands of +-1e-2 terms cancel to ~0; a different but fixed association moves them by ulps
    # of the partial sums, not of the result)
    np.testing.assert_allclose(Gg, G, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(g1g, g1, rtol=2e-5, atol=1e-6)
    a.segsum(B, S, torch.from_numpy(dX).cuda(), torch.from_numpy(gy1).cuda(), torch.from_numpy(gy2).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(a.G.cpu().numpy()[slot[uniq]], Gg)             # deterministic run to run


def test_adam_tf1_bit_exact_over_steps():
    """Element-wise optimizer math is reproducible bit for bit (no FMA contraction, IEEE div/sqrt)."""
    from recsys_amd import _lib
    from recsys_amd.ops import AdamTF1, DenseArena
    rng = np.random.default_rng(3)
    row_off = np.array([0, 5, 25, 26], np.int64)
    B, D, F = 16, 16, 3
    a, tables, w1 = _arena(row_off, D, B, rng)
    dense = DenseArena({"w": (7, 3), "b": (5,)}, "cuda")            # 26 floats: exercises the scalar tail
    dvals = {"w": rng.standard_normal((7, 3)).astype(np.float32), "b": rng.standard_normal(5).astype(np.float32)}
    dense.load(dvals)
    opt = AdamTF1(device="cuda")
    oo = nn.AdamTF1(dtype=np.float32)
    T, W = tables.copy(), w1.copy()
    for step in range(4):
        ids = synth_ids(rng, B, row_off)
        dX = (rng.standard_normal((B, F * D)) * 10.0 ** rng.integers(-6, 0)).astype(np.float32)
        gy1 = rng.standard_normal(B).astype(np.float32) * 1e-3
        gd = {k: rng.standard_normal(v.shape).astype(np.float32) * 1e-3 for k, v in dvals.items()}
        idt = torch.from_numpy(ids).cuda()
        a.field_sort(idt)
        a.segsum(B, None, torch.from_numpy(dX).cuda(), torch.from_numpy(gy1).cuda(), None)
        for k in gd:
            dense[k].grad.copy_(torch.from_numpy(gd[k]))
        opt.step(a.adam_segments() + dense.adam_segments())
        torch.cuda.synchronize()
        rws = ids.astype(np.int64) + row_off[None, :-1]
        u, G = nn.segment_sum_rows(*models._pairs_field_major(rws, dX.reshape(B, F, D)))
        _, g1 = nn.segment_sum_rows(*models._pairs_field_major(rws, np.repeat(gy1[:, None], F, 1)))
        oo.apply_sparse("t", T, u, G)
        gw = np.zeros_like(W)
        gw[u] = g1
        oo.apply_dense("w1", W, gw)
        for k in gd:
            oo.apply_dense(k, dvals[k], gd[k])
        oo.finish_step()
        assert np.array_equal(a.tables.cpu().numpy(), T), step
        assert np.array_equal(a.w1.cpu().numpy(), W), step
        for k in gd:
            assert np.array_equal(dense[k].detach().cpu().numpy(), dvals[k]), (step, k)
            assert float(dense[k].grad.abs().max()) == 0.0            # zero_grad after use
    assert opt.global_step == 4


def test_adam_lazy_rows_mode():
    from recsys_amd.ops import AdamTF1
    rng = np.random.default_rng(4)
    row_off = np.array([0, 50, 60], np.int64)
    B, D, F = 8, 16, 2
    a, tables, w1 = _arena(row_off, D, B, rng)
    opt, oo = AdamTF1(device="cuda"), nn.AdamTF1(dtype=np.float32)
    T, W = tables.copy(), w1.copy()
    for step in range(3):
        ids = synth_ids(rng, B, row_off)
        dX = rng.standard_normal((B, F * D)).astype(np.float32) * 1e-3
        gy1 = rng.standard_normal(B).astype(np.float32) * 1e-3
        idt = torch.from_numpy(ids).cuda()
        a.field_sort(idt)
        a.segsum(B, None, torch.from_numpy(dX).cuda(), torch.from_numpy(gy1).cuda(), None)
        opt.step(a.adam_segments(lazy=True))
        rws = ids.astype(np.int64) + row_off[None, :-1]
        u, G = nn.segment_sum_rows(*models._pairs_field_major(rws, dX.reshape(B, F, D)))
        _, g1 = nn.segment_sum_rows(*models._pairs_field_major(rws, np.repeat(gy1[:, None], F, 1)))
        oo.apply_sparse("t", T, u, G, lazy=True)
        oo.apply_sparse("w", W, u, g1, lazy=True)
        oo.finish_step()
    torch.cuda.synchronize()
    assert np.array_equal(a.tables.cpu().numpy(), T) and np.array_equal(a.w1.cpu().numpy(), W)


@pytest.mark.parametrize("tower,dropout,B", [("hip", 0.0, 64), ("torch", 0.0, 64), ("hip", 0.5, 50), ("torch", 0.5, 50)])
def test_deepfm_train_parity_small(tower, dropout, B):
    err, losses, perr = deepfm_parity_run(B=B, steps=4, seed=2, rows=(3, 7, 40, 11, 600), D=16, layers=(32, 16),
                                          return_all=True, tower=tower, dropout=dropout)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5
    assert max(perr.values()) < 1e-5, perr


@pytest.mark.parametrize("tower,dropout", [("hip", 0.0), ("hip", 0.5), ("torch", 0.0)])
def test_deepfm_train_parity_criteo_bs256(tower, dropout):
    """BASELINE config 2: DeepFM, Criteo 39 fields, d=16, DNN 100-100, batch 256 -- oracle vs HIP, tolerance 1e-5."""
    err, losses, perr = deepfm_parity_run(B=256, steps=3, seed=5, return_all=True, tower=tower, dropout=dropout)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5, losses
    # parameters: Adam divides by sqrt(v)+1e-8, so GEMM-rounding noise on ~1e-8 gradients is amplified to a few
    # 1e-6 of the 1e-3 step; outputs (the north_star tolerance) stay within 1e-5, parameters within 5e-5
    assert max(perr.values()) < 5e-5, perr


def test_deepfm_hip_graph_matches_eager():
    e1 = deepfm_parity_run(B=128, steps=6, seed=7, rows=(3, 7, 40, 11, 600), layers=(32, 16), use_graph=True)
    assert e1 < 1e-5, e1


def test_deepfm_lazy_rows_mode_parity():
    err, losses, perr = deepfm_parity_run(B=64, steps=3, seed=8, rows=(3, 7, 40, 11, 600), layers=(32, 16),
                                          adam_mode="lazy_rows", return_all=True)
    assert err < 1e-5 and max(perr.values()) < 1e-5, (err, perr)


def _hash32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def test_tower_rng_dropout_is_the_documented_hash_and_consistent_fwd_bwd():
    """The in-kernel dropout mask (no mask buffer) must be the documented counter hash: training one step with the
    RNG must equal, bit for bit, training the same step with that mask injected explicitly -- which also proves the
    forward (next layer's A-load / head) and backward (d-input epilogue, dW A-load) evaluate the SAME mask."""
    from recsys_amd import deepfm
    from recsys_amd.feature_columns import build_feature_columns  # noqa: F401
    from tests.parity_util import load_oracle_weights, make_estimator, small_columns
    from oracle import init
    rows, D, layers, B, rate, seed = (3, 7, 40, 11, 600), 16, (32, 16), 96, 0.5, 0x5eed
    lin, emb = small_columns(rows, D)
    row_off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    P = init.deepfm_params(3, D, layers, np.float32, row_off)
    rng = np.random.default_rng(9)
    ids = synth_ids(rng, B, row_off)
    y = rng.integers(0, 2, B).astype(np.float32)
    outs = []
    for inject in (False, True):
        params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": D, "learning_rate": 1e-3,
                  "dropout": rate, "deep_layers": "32,16", "max_batch_size": B, "tower": "hip"}
        est = make_estimator(deepfm.model_fn, params)
        f = {"ids": torch.from_numpy(ids).cuda()}
        est._call_model_fn(f, None, "infer")
        load_oracle_weights(est, P)
        if inject:
            step = 1                                              # Adam state word 3 before the first step
            masks = []
            for l, n in enumerate(layers):
                key = _hash32(np.array([(seed ^ ((step * 0x9E3779B9) & 0xFFFFFFFF) ^ ((l * 0x85EBCA6B + 0x27220A95) & 0xFFFFFFFF))]))[0]
                h = _hash32(np.arange(B * n, dtype=np.uint64) ^ key)
                masks.append(torch.from_numpy((h >= np.uint64(int(rate * 2 ** 32))).astype(np.float32).reshape(B, n)).cuda())
            keep = float(torch.cat([m.reshape(-1) for m in masks]).mean())
            assert 0.45 < keep < 0.55                              # ~Bernoulli(1 - rate)
            est.params["_dropout_masks"] = masks
        loss = float(est._train_step(f, torch.from_numpy(y).cuda()))
        a = est.store.embeddings["input_layer"]
        outs.append((loss, a.tables.cpu().numpy().copy(), est.store.dense.flat.cpu().numpy().copy()))
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


# ------------------------------------------------------------------------------------------- FM / DCN
def test_fm_train_parity_criteo_bs256():
    """BASELINE config 1 (fm.py Criteo d=16 bs=256): oracle vs the fused embedding kernels."""
    err, losses, perr = deepfm_parity_run(B=256, steps=3, seed=11, return_all=True, kind="fm")
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5, losses
    assert max(perr.values()) < 5e-5, perr


@pytest.mark.parametrize("B,dim,L", [(33, 624, 3), (256, 624, 4), (5, 64, 1), (70, 1024, 7), (1500, 624, 3), (4099, 624, 2),
                                     (9001, 128, 3)])
def test_cross_layers_op_parity(B, dim, L):
    from recsys_amd.ops import CrossFn, CrossLayers
    rng = np.random.default_rng(B + L)
    x0 = rng.standard_normal((B, dim)).astype(np.float32) * 0.3
    W = rng.standard_normal((L, dim)).astype(np.float32) * 0.05
    Bc = rng.standard_normal((L, dim)).astype(np.float32) * 0.05
    g = rng.standard_normal((B, dim)).astype(np.float32)
    xs, ss = models.cross_fwd(x0, W, Bc)
    dx0, dW, dB = models.cross_bwd(xs, ss, W, g)
    op = CrossLayers(dim, L, B)
    tx, tW, tB = (torch.from_numpy(a).cuda().requires_grad_() for a in (x0, W, Bc))
    out = CrossFn.apply(tx, tW, tB, op)
    out.backward(torch.from_numpy(g).cuda())
    tol = dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(out.detach().cpu().numpy(), xs[-1], **tol)
    np.testing.assert_allclose(tx.grad.cpu().numpy(), dx0, **tol)
    # batch sums of B terms of magnitude ~2 (unit-variance upstream gradient): the fp32 summation-order noise between the
    # kernel's per-wave partials and numpy's pairwise sum grows with B (~7e-8 x |partial sum| per add), so does the tolerance
    bt = 1e-4 * max(1.0, B / 1500.0)
    np.testing.assert_allclose(tW.grad.cpu().numpy(), dW, rtol=1e-4, atol=bt)
    np.testing.assert_allclose(tB.grad.cpu().numpy(), dB, rtol=1e-4, atol=bt)
    assert tuple(models.cross_fwd(np.array([[1.0, 2.0, 0, 0]], np.float32), np.array([[1.0, 1, 0, 0]], np.float32),
                                  np.zeros((1, 4), np.float32))[0][-1][0][:2]) == (4.0, 8.0)   # Appendix B-6 KAT


@pytest.mark.parametrize("tower,dropout", [("hip", 0.0), ("hip", 0.5), ("torch", 0.5)])
def test_dcn_train_parity_small(tower, dropout):
    err, losses, perr = deepfm_parity_run(B=48, steps=4, seed=12, rows=(3, 7, 40, 11, 600), D=16, layers=(32, 16),
                                          return_all=True, tower=tower, dropout=dropout, kind="dcn", cross_layers=3)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5, losses
    assert max(perr.values()) < 5e-5, perr


def test_dcn_train_parity_criteo():
    """BASELINE config 4 shape (dcn.py Criteo d=16, 3 cross layers) at a batch the oracle finishes in seconds."""
    err, losses, perr = deepfm_parity_run(B=512, steps=2, seed=13, return_all=True, kind="dcn", cross_layers=3, dropout=0.5)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 2e-5, losses
    assert max(perr.values()) < 5e-5, perr


def test_large_batch_tower_and_dcn_parity_bs1024():
    """B > 512 takes the fixed-point statistics rows (include/rsx.h RSX_TOWER_FIXED_STATS_MIN_B); B >= 1024 the split-batch dW tiles
    (+ tower_reduce_dw_k) and 4 row tiles per d(input) workgroup: DeepFM and DCN at batch
    1024 / 1100 (ragged last tiles) on a small layout vs the oracle."""
    for kind, B in (("deepfm", 1024), ("dcn", 1100)):
        err, losses, perr = deepfm_parity_run(B=B, steps=2, seed=41, rows=(3, 7, 40, 11, 600), D=16, layers=(32, 16),
                                              return_all=True, dropout=0.5, kind=kind, cross_layers=3)
        assert err < 1e-5, (kind, err)
        for lg, lo in losses:
            assert abs(lg - lo) < 2e-5, (kind, losses)
        assert max(perr.values()) < 5e-5, (kind, perr)


def test_library_loaded_before_torch_still_launches():
    """__graft_entry__.build() followed by smoke() in ONE process: the C-ABI library is dlopen'ed before anything touched
    torch.cuda.  It must still bind to torch's HIP runtime (recsys_amd._lib.lib imports torch first); with /opt/rocm's
    libamdhip64 loaded as a second runtime the first kernel launch fails."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; g.build(); g.smoke(); print('ORDER_OK')" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "ORDER_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_overlapped_optimizer_is_bit_identical_to_the_plain_path():
    """The production step (dedup sort riding in the first forward launch, untouched-row Adam sweep carried by the tower
    launches, scatter fused with the touched-row update, 8-step HIP graphs) against the plain one (stand-alone sort,
    segment-sum, ONE full TF-1 Adam sweep, eager): same weights, same batches, dropout by the counter hash.  The split is
    exact arithmetic, so after hundreds of steps every table row, Adam slot and dense variable must be bit-identical; a
    race between the carried sweep and the touched-row update would show here (scripts/soak_overlap.py runs it longer)."""
    from recsys_amd import deepfm, synthetic
    from recsys_amd.estimator import Estimator, PackedBatch, RunConfig
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    N, B = 320, 256
    lin, emb = build_feature_columns(16, "indicator_all")
    layout = CriteoLayout.from_columns(emb)
    host = synthetic.criteo_id_batches(layout, 16, B, seed=321)
    res = []
    for overlap, graph in ((True, True), (False, False)):
        params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
                  "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": B, "overlap_adam": overlap}
        est = Estimator(deepfm.model_fn, None, params, RunConfig(use_hip_graph=graph, adam_mode="tf1_dense", device="cuda", seed=9))
        feats = [PackedBatch({"ids": i}, y, device="cuda") for i, y, _ in host]
        with torch.no_grad():
            est._call_model_fn(feats[0].views()[0], None, "infer")
        if graph:
            est.train_resident(feats, N, 8)
        else:
            for s_ in range(N):
                est._train_step(feats[s_ % 16])
        torch.cuda.synchronize()
        a = est.store.embeddings["input_layer"]
        assert est.global_step == N
        res.append((a.tables.clone(), a.m_t.clone(), a.v_t.clone(), a.w1.clone(), est.store.dense.flat.clone()))
    for name, x, y in zip(("tables", "m", "v", "w1", "dense"), *res):
        assert torch.isfinite(x).all(), name
        assert torch.equal(x, y), (name, float((x - y).abs().max()))


def test_gather_with_the_sort_riding_along_equals_the_two_separate_launches():
    """rsx_gather_fm_fwd_sort: gather outputs and every sort output identical to rsx_gather_fm_fwd + rsx_field_sort."""
    from oracle import criteo
    from recsys_amd.ops import EmbeddingArena
    from tests.parity_util import synth_ids
    row_off = criteo.row_offsets()
    rng = np.random.default_rng(5)
    for B in (256, 100, 1024):
        a = EmbeddingArena(row_off, 16, 1024, "cuda", with_w1=True, w1_field_mask=(1 << 39) - 1)
        b = EmbeddingArena(row_off, 16, 1024, "cuda", with_w1=True, w1_field_mask=(1 << 39) - 1)
        with torch.no_grad():
            a.tables.normal_(); a.w1.normal_()
            b.tables.copy_(a.tables); b.w1.copy_(a.w1)
        for rep in range(2):                                  # the second round also clears the first round's slot map
            ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
            ra = a.gather(ids, fm=True, first_order=True, sort_job=a.sort_job(ids))
            rb = b.gather(ids, fm=True, first_order=True)
            b.field_sort(ids)
            torch.cuda.synchronize()
            for x, y in zip(ra, rb):
                assert torch.equal(x, y)
            for k in ("perm", "seg_off", "uniq_row", "nuniq", "slot"):
                assert torch.equal(getattr(a, k), getattr(b, k)), (B, rep, k)
