"""Second-opinion check of the numpy oracle (hand-written backward) against `oracle.torch_ref`, a TF-op-literal
PyTorch-CPU restatement whose gradients come from autograd: fp64 logits, loss, EVERY gradient and one TF-1 Adam step
must agree for FM, DeepFM, DCN, xDeepFM (CIN via split / matmul / reshape / transpose / conv1d-as-matmul, i.e. the
(f major, h minor) flattening is derived, not assumed) and DIN (query via tile + reshape)."""
import numpy as np
import pytest
import torch

from oracle import init, models, nn, torch_ref as tr
from tests.test_oracle_grads import _masks, rand_ids, small_layout


def _oracle_dense_grads(model, args, y, kw):
    P = model.P
    z = model.forward(*args, train=True, **kw)
    loss, dz = nn.sigmoid_ce_mean(z, y)
    g, s = model.backward(dz)
    dense = {k: v.reshape(P[k].shape) for k, v in g.items()}
    for k, (rows, vals) in s.items():
        d = np.zeros_like(P[k])
        np.add.at(d, rows, vals)
        dense[k] = d
    return z, loss, dense


def _compare(model, args, y, kw, logits_fn, atol=1e-10):
    P0 = {k: v.copy() for k, v in model.P.items()}
    z, loss, dense = _oracle_dense_grads(model, args, y, kw)
    T = tr.params_to_torch(P0)
    lt, zt, gt = tr.loss_and_grads(logits_fn, T, torch.tensor(y))
    np.testing.assert_allclose(zt.numpy(), z, atol=1e-11)
    assert abs(float(lt) - float(loss)) < 1e-12
    assert set(dense) == set(gt)
    for k in dense:
        np.testing.assert_allclose(gt[k].numpy(), dense[k], atol=atol, err_msg=k)
    # one optimizer step: oracle train_step (sparse / dense TF formulas) vs the torch restatement
    opt = nn.AdamTF1(dtype=np.float64)
    models.train_step(model, opt, args, y, kw)
    topt = tr.AdamTF1()
    topt.step(T, gt)
    for k in model.P:
        np.testing.assert_allclose(T[k].detach().numpy(), model.P[k], atol=1e-12, err_msg="after Adam: " + k)
    # the step moved every touched variable by ~lr
    assert max(np.abs(model.P[k] - P0[k]).max() for k in P0) > 5e-4


def _ids_t(ids):
    return torch.tensor(ids.astype(np.int64))


def test_fm_and_deepfm_match_torch_literal_onehot():
    rng = np.random.default_rng(11)
    off = small_layout()
    B, D = 9, 4
    ids = rand_ids(rng, B, off)
    y = rng.integers(0, 2, B).astype(np.float64)
    P = init.deepfm_params(3, D, (6, 5), np.float64, off, with_dnn=False)
    P["b1"] += 0.3
    for literal in (False, True):
        m = models.FM({k: v.copy() for k, v in P.items()}, off)
        _compare(m, (ids,), y, {}, lambda T: tr.fm_logits(T, _ids_t(ids), off, literal=literal))
    P = init.deepfm_params(4, D, (6, 5), np.float64, off)
    for k in P:
        if k.endswith((".b0", ".b1", "bout", "b1")):
            P[k] += 0.2
    mk = _masks(rng, B, (6, 5))
    m = models.DeepFM(P, off, 2, 0.5)
    _compare(m, (ids,), y, {"masks": mk},
             lambda T: tr.deepfm_logits(T, _ids_t(ids), off, 2, 0.5, mk, True, literal=True))


def test_dcn_matches_torch():
    rng = np.random.default_rng(12)
    off = small_layout()
    B, D = 10, 4
    ids = rand_ids(rng, B, off)
    y = rng.integers(0, 2, B).astype(np.float64)
    P = init.dcn_params(5, D, (6, 5), 3, np.float64, off)
    for k in ("dnn.b0", "dnn.b1"):
        P[k] += 0.2
    mk = _masks(rng, B, (6, 5))
    _compare(models.DCN(P, off, 2, 0.5), (ids,), y, {"masks": mk},
             lambda T: tr.dcn_logits(T, _ids_t(ids), off, 2, 0.5, mk, True))


def test_xdeepfm_matches_torch_literal_cin():
    rng = np.random.default_rng(13)
    rows = (3, 7, 4, 11, 6)
    off = small_layout(5, rows)
    B, D = 7, 3
    ids = rand_ids(rng, B, off)
    y = rng.integers(0, 2, B).astype(np.float64)
    cat_slot = np.array([0, 2, 4] + [0] * 23)
    cat_rows = [rows[0], rows[2], rows[4]] + [rows[0]] * 23
    cat_off = np.concatenate([[0], np.cumsum(cat_rows)]).astype(np.int64)
    P = init.xdeepfm_params(6, D, (6, 5), (4, 3), np.float64, off, int(cat_off[-1]))
    for k in ("dnn.b0", "dnn.b1", "dnn.bout", "cin.c0", "cin.c1", "cin.bout", "lin.b"):
        P[k] += 0.3
    logx = rng.random((B, 13))
    mk = _masks(rng, B, (6, 5))
    m = models.XDeepFM(P, off, cat_slot, cat_off, (4, 3), 2, 0.5)
    _compare(m, (ids, logx), y, {"masks": mk},
             lambda T: tr.xdeepfm_logits(T, _ids_t(ids), torch.tensor(logx), off, cat_slot, cat_off, (4, 3), 2, 0.5,
                                         mk, True))


def test_cin_layer_literal_equals_einsum_forward():
    """The CIN layer alone, non-square F != H, so that an (h major) flattening could not pass."""
    rng = np.random.default_rng(14)
    B, F, H, N, D = 3, 5, 4, 6, 3
    X0, Xk = rng.standard_normal((B, F, D)), rng.standard_normal((B, H, D))
    W, c = rng.standard_normal((F * H, N)), rng.standard_normal(N)
    ref = tr.cin_layer_literal(torch.tensor(X0), torch.tensor(Xk), torch.tensor(W), torch.tensor(c)).numpy()
    np.testing.assert_allclose(models.cin_layer_fwd(X0, Xk, W, c), ref, atol=1e-12)
    # and the h-major reading is measurably different
    Wt = W.reshape(F, H, N).transpose(1, 0, 2).reshape(F * H, N)
    assert np.abs(models.cin_layer_fwd(X0, Xk, Wt, c) - ref).max() > 1e-3


def test_din_matches_torch():
    rng = np.random.default_rng(15)
    B, Pn, K = 6, 7, 4
    P = init.din_params(7, K, 20, 9, np.float64)
    for k in P:
        if ".b" in k:
            P[k] += 0.2
    P["item_bias"] += rng.standard_normal(20) * 0.1
    i_id = rng.integers(1, 20, B)
    i_cate = rng.integers(1, 9, B)
    hist_i = rng.integers(1, 20, (B, Pn))
    hist_c = rng.integers(1, 9, (B, Pn))
    for b in range(B):
        n = rng.integers(1, Pn + 1)
        hist_i[b, n:] = 0
        hist_c[b, n:] = 0
    y = rng.integers(0, 2, B).astype(np.float64)
    mk = {"att_i": [(rng.random((B * Pn, n)) >= 0.5).astype(np.float64) for n in (80, 40)],
          "att_c": [(rng.random((B * Pn, n)) >= 0.5).astype(np.float64) for n in (80, 40)],
          "mlp": [(rng.random((B, n)) >= 0.5).astype(np.float64) for n in (100, 50, 20)]}
    t = lambda a: torch.tensor(np.asarray(a).astype(np.int64))
    _compare(models.DIN(P, 0.5), (i_id, i_cate, hist_i, hist_c), y, {"masks": mk},
             lambda T: tr.din_logits(T, t(i_id), t(i_cate), t(hist_i), t(hist_c), 0.5, mk, True))


def test_cpu_baseline_step_runs_and_matches_oracle_fp32():
    """The timed CPU baseline (bench.py) computes the same DeepFM step as the oracle: loss of the first two steps."""
    rng = np.random.default_rng(16)
    off = small_layout()
    B, D = 32, 4
    P = init.deepfm_params(4, D, (6, 5), np.float32, off)
    ids = [rand_ids(rng, B, off) for _ in range(2)]
    ys = [rng.integers(0, 2, B).astype(np.float32) for _ in range(2)]
    for literal in (False, True):
        cpu = tr.DeepFMCpuBaseline({k: v.copy() for k, v in P.items()}, off, 2, 0.0, literal=literal)
        m = models.DeepFM({k: v.copy() for k, v in P.items()}, off, 2, 0.0)
        opt = nn.AdamTF1(dtype=np.float32)
        for s in range(2):
            l_cpu = cpu.step(ids[s], ys[s])
            l_or, _ = models.train_step(m, opt, (ids[s],), ys[s], {})
            assert abs(l_cpu - float(l_or)) < 2e-6, (literal, s, l_cpu, l_or)
        for k in P:
            np.testing.assert_allclose(cpu.P[k].detach().numpy().reshape(-1), m.P[k].reshape(-1), atol=3e-6, err_msg=k)
