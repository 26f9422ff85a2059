"""fp64 finite-difference checks of the oracle's hand-written backward passes ('Gradient' row)."""
import numpy as np
import pytest

from oracle import init, models, nn


def small_layout(F=5, rows=(3, 7, 4, 11, 6)):
    return np.concatenate([[0], np.cumsum(rows[:F])]).astype(np.int64)


def rand_ids(rng, B, row_off):
    F = len(row_off) - 1
    return np.stack([rng.integers(0, row_off[f + 1] - row_off[f], B) for f in range(F)], 1).astype(np.int32)


def loss_of(model, args, y, kw):
    z = model.forward(*args, train=True, **kw)
    return nn.sigmoid_ce_mean(z, y)[0]


def fd_check(model, args, y, kw, names, rng, n_probe=6, h=1e-6, tol=2e-6):
    P = model.P
    z = model.forward(*args, train=True, **kw)
    _, dz = nn.sigmoid_ce_mean(z, y)
    g, s = model.backward(dz)
    dense = {k: v.reshape(P[k].shape) for k, v in g.items()}
    for k, (rows, vals) in s.items():
        d = np.zeros_like(P[k])
        np.add.at(d, rows, vals)
        dense[k] = d
    assert set(dense) == set(names), (sorted(dense), sorted(names))
    for k in names:
        flat = P[k].reshape(-1)
        gflat = dense[k].reshape(-1)
        nz = np.flatnonzero(gflat)
        cand = nz if len(nz) else np.arange(flat.size)
        for i in rng.choice(cand, min(n_probe, len(cand)), replace=False):
            old = flat[i]
            flat[i] = old + h
            lp = loss_of(model, args, y, kw)
            flat[i] = old - h
            lm = loss_of(model, args, y, kw)
            flat[i] = old
            num = (lp - lm) / (2 * h)
            assert abs(num - gflat[i]) <= tol * max(1.0, abs(num)), (k, i, num, gflat[i])


def _masks(rng, B, layers, rate=0.5):
    return [(rng.random((B, n)) >= rate).astype(np.float64) for n in layers]


def test_fm_and_deepfm_grads():
    rng = np.random.default_rng(1)
    off = small_layout()
    B, D = 8, 4
    ids = rand_ids(rng, B, off)
    y = rng.integers(0, 2, B).astype(np.float64)
    P = init.deepfm_params(3, D, (6, 5), np.float64, off, with_dnn=False)
    P["b1"] += 0.3
    fd_check(models.FM(P, off), (ids,), y, {}, list(P), rng)
    P = init.deepfm_params(4, D, (6, 5), np.float64, off)
    for k in P:
        if k.endswith((".b0", ".b1", "bout", "b1")):
            P[k] += 0.2            # keep relus alive
    m = models.DeepFM(P, off, 2, 0.5)
    fd_check(m, (ids,), y, {"masks": _masks(rng, B, (6, 5))}, list(P), rng)


def test_dcn_grads():
    rng = np.random.default_rng(2)
    off = small_layout()
    B, D = 7, 4
    ids = rand_ids(rng, B, off)
    y = rng.integers(0, 2, B).astype(np.float64)
    P = init.dcn_params(5, D, (6, 5), 3, np.float64, off)
    for k in ("dnn.b0", "dnn.b1"):
        P[k] += 0.2
    fd_check(models.DCN(P, off, 2, 0.5), (ids,), y, {"masks": _masks(rng, B, (6, 5))}, list(P), rng)


def test_xdeepfm_grads():
    rng = np.random.default_rng(3)
    rows = (3, 7, 4, 11, 6)
    off = small_layout(5, rows)
    B, D = 6, 3
    ids = rand_ids(rng, B, off)
    y = rng.integers(0, 2, B).astype(np.float64)
    # synthetic layout: slots 0,2,4 play the role of hashed fields for the linear part
    cat_slot = np.array([0, 2, 4] + [0] * 23)
    cat_rows = [rows[0], rows[2], rows[4]] + [rows[0]] * 23
    cat_off = np.concatenate([[0], np.cumsum(cat_rows)]).astype(np.int64)
    P = init.xdeepfm_params(6, D, (6, 5), (4, 3), np.float64, off, int(cat_off[-1]))
    for k in ("dnn.b0", "dnn.b1", "dnn.bout", "cin.c0", "cin.c1", "cin.bout", "lin.b"):
        P[k] += 0.3
    logx = rng.random((B, 13))
    m = models.XDeepFM(P, off, cat_slot, cat_off, (4, 3), 2, 0.5)
    fd_check(m, (ids, logx), y, {"masks": _masks(rng, B, (6, 5))}, list(P), rng)


def test_din_grads():
    rng = np.random.default_rng(4)
    B, Pn, K = 5, 6, 4
    P = init.din_params(7, K, 20, 9, np.float64)
    for k in P:
        if ".b" in k:
            P[k] += 0.2
    P["item_bias"] += rng.standard_normal(20) * 0.1
    i_id = rng.integers(1, 20, B)
    i_cate = rng.integers(1, 9, B)
    hist_i = rng.integers(1, 20, (B, Pn))
    hist_c = rng.integers(1, 9, (B, Pn))
    for b in range(B):                       # ragged histories, zero padded (din/din.py:56-57,107)
        n = rng.integers(1, Pn + 1)
        hist_i[b, n:] = 0
        hist_c[b, n:] = 0
    y = rng.integers(0, 2, B).astype(np.float64)
    mk = {"att_i": [(rng.random((B * Pn, n)) >= 0.5).astype(np.float64) for n in (80, 40)],
          "att_c": [(rng.random((B * Pn, n)) >= 0.5).astype(np.float64) for n in (80, 40)],
          "mlp": [(rng.random((B, n)) >= 0.5).astype(np.float64) for n in (100, 50, 20)]}
    m = models.DIN(P, 0.5)
    fd_check(m, (i_id, i_cate, hist_i, hist_c), y, {"masks": mk}, list(P), rng)
    # padding rows (id 0) get exactly zero gradient through the masked sum but non-zero through the MLP features:
    z = m.forward(i_id, i_cate, hist_i, hist_c, train=True, masks=mk)
    assert np.isfinite(z).all()


def test_deepfm_matches_torch_autograd():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    off = small_layout()
    B, D = 9, 4
    ids = rand_ids(rng, B, off)
    y = rng.integers(0, 2, B).astype(np.float64)
    P = init.deepfm_params(8, D, (6, 5), np.float64, off)
    m = models.DeepFM(P, off, 2, 0.0)
    z = m.forward(ids, train=True)
    loss, dz = nn.sigmoid_ce_mean(z, y)
    g, s = m.backward(dz)
    T = {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}
    rows = torch.tensor(ids.astype(np.int64) + off[None, :-1])
    E = T["tables"][rows]
    y1 = torch.relu(T["w1"][rows].sum(1) + T["b1"])
    S = E.sum(1)
    y2 = 0.5 * (S * S - (E * E).sum(1)).sum(1)
    h = E.reshape(B, -1)
    for i in range(2):
        a = torch.relu(h @ T[f"dnn.W{i}"] + T[f"dnn.b{i}"])
        h = torch.nn.functional.batch_norm(a, None, None, T[f"dnn.gamma{i}"], T[f"dnn.beta{i}"], True, 0.0, 1e-3)
    yd = torch.relu(h @ T["dnn.Wout"] + T["dnn.bout"])
    zt = (torch.cat([y1[:, None], y2[:, None], yd], 1) @ T["out.W"] + T["out.b"]).reshape(-1)
    lt = torch.nn.functional.binary_cross_entropy_with_logits(zt, torch.tensor(y))
    lt.backward()
    assert abs(lt.item() - loss) < 1e-12
    np.testing.assert_allclose(zt.detach().numpy(), z, atol=1e-12)
    for k, v in g.items():
        np.testing.assert_allclose(T[k].grad.numpy().reshape(-1), v.reshape(-1), atol=1e-10, err_msg=k)
    for k, (r, vals) in s.items():
        d = np.zeros_like(P[k])
        np.add.at(d, r, vals)
        np.testing.assert_allclose(T[k].grad.numpy(), d, atol=1e-10, err_msg=k)


def test_train_step_dp_identity():
    """DP(N, b) == single(N*b) for the sparse+dense update when BN is per-replica-free (no DNN): FM."""
    rng = np.random.default_rng(6)
    off = small_layout()
    B, D = 8, 4
    ids = rand_ids(rng, B, off)
    y = rng.integers(0, 2, B).astype(np.float64)
    P1 = init.deepfm_params(9, D, (), np.float64, off, with_dnn=False)
    P2 = {k: v.copy() for k, v in P1.items()}
    o1, o2 = nn.AdamTF1(dtype=np.float64), nn.AdamTF1(dtype=np.float64)
    models.train_step(models.FM(P1, off), o1, (ids,), y)
    # two replicas of 4: grads of (1/N) * replica mean loss, summed
    m = models.FM(P2, off)
    gs, ss = [], []
    for r in range(2):
        sl = slice(r * 4, r * 4 + 4)
        z = m.forward(ids[sl])
        _, dz = nn.sigmoid_ce_mean(z, y[sl])
        g, s = m.backward(dz / 2)
        gs.append(g)
        ss.append(s)
    for k in gs[0]:
        o2.apply_dense(k, P2[k], (gs[0][k] + gs[1][k]).reshape(P2[k].shape))
    for k in ss[0]:
        rows = np.concatenate([ss[0][k][0], ss[1][k][0]])
        vals = np.concatenate([ss[0][k][1], ss[1][k][1]])
        u, G = nn.segment_sum_rows(rows, vals)
        if k == "w1":
            dense = np.zeros_like(P2[k])
            dense[u] = G
            o2.apply_dense(k, P2[k], dense)
        else:
            o2.apply_sparse(k, P2[k], u, G)
    for k in P1:
        np.testing.assert_allclose(P1[k], P2[k], atol=1e-14, err_msg=k)
