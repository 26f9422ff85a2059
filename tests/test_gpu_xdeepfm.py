"""xDeepFM (BASELINE config 3): CIN layer op parity and full train steps vs the oracle."""
import numpy as np
import pytest

from oracle import criteo, init, models, nn
from tests.parity_util import synth_ids

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("B,F,H,N", [(5, 39, 39, 128), (4, 39, 128, 128), (3, 7, 5, 20), (9, 39, 20, 10), (2, 4, 33, 16)])
def test_cin_layer_fwd_bwd(B, F, H, N):
    from recsys_amd.ops import CinLayerFn
    rng = np.random.default_rng(B * 100 + H)
    D = 16
    X0 = rng.standard_normal((B, F, D)).astype(np.float32) * 0.3
    Xk = rng.standard_normal((B, H, D)).astype(np.float32) * 0.3
    W = rng.standard_normal((F * H, N)).astype(np.float32) * 0.1
    c = rng.standard_normal(N).astype(np.float32) * 0.1
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    out_o = models.cin_layer_fwd(X0.astype(np.float64), Xk.astype(np.float64), W.astype(np.float64), c.astype(np.float64))
    d0_o, dk_o, dW_o, dc_o = models.cin_layer_bwd(X0.astype(np.float64), Xk.astype(np.float64), W.astype(np.float64), out_o,
                                                  g.astype(np.float64))
    t = [torch.from_numpy(a).cuda().requires_grad_() for a in (X0, Xk, W, c)]
    out = CinLayerFn.apply(*t)
    out.backward(torch.from_numpy(g).cuda())
    tol = dict(rtol=2e-5, atol=2e-5)                      # fp32 MFMA vs fp64 oracle
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_o, **tol)
    # relu masks can differ only where |pre| ~ 1e-7; compare gradients where the fp64 pre-activation is not that close to 0
    np.testing.assert_allclose(t[0].grad.cpu().numpy(), d0_o, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(t[1].grad.cpu().numpy(), dk_o, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(t[2].grad.cpu().numpy(), dW_o, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(t[3].grad.cpu().numpy(), dc_o, rtol=1e-4, atol=1e-4)


def test_cin_first_layer_aliasing_x0():
    """Layer 0 passes X0 as both operands (xdeepfm/xdeepfm.py:143-148); autograd must add both gradient roles."""
    from recsys_amd.ops import CinLayerFn
    rng = np.random.default_rng(1)
    B, F, N, D = 6, 39, 32, 16
    X0 = rng.standard_normal((B, F, D)).astype(np.float32) * 0.3
    W = rng.standard_normal((F * F, N)).astype(np.float32) * 0.1
    c = np.zeros(N, np.float32)
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    X64 = X0.astype(np.float64)
    out_o = models.cin_layer_fwd(X64, X64, W.astype(np.float64), c.astype(np.float64))
    d0, dk, _, _ = models.cin_layer_bwd(X64, X64, W.astype(np.float64), out_o, g.astype(np.float64))
    tx = torch.from_numpy(X0).cuda().requires_grad_()
    out = CinLayerFn.apply(tx, tx, torch.from_numpy(W).cuda(), torch.from_numpy(c).cuda())
    out.backward(torch.from_numpy(g).cuda())
    np.testing.assert_allclose(tx.grad.cpu().numpy(), d0 + dk, rtol=1e-4, atol=1e-4)


def _xdeepfm_run(B, steps, seed, cin, layers, dropout, use_graph=False, D=16, cin_split=None):
    from recsys_amd import xdeepfm
    from recsys_amd.estimator import ModeKeys
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    from tests.parity_util import make_estimator
    rng = np.random.default_rng(seed)
    lin, emb = build_feature_columns(D, "numeric+indicator")
    lay = CriteoLayout.from_columns(emb)
    row_off = criteo.row_offsets()
    cat_slot, cat_off = init.xdeepfm_layout()
    P = init.xdeepfm_params(seed, D, layers, cin, np.float32, row_off)
    for k in ("lin.b", "cin.bout", "dnn.bout"):
        P[k] += np.float32(0.05)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": D, "learning_rate": 1e-3,
              "dropout": dropout, "deep_layers": ",".join(map(str, layers)), "cross_layers": ",".join(map(str, cin)),
              "max_batch_size": B, "cin_split": cin_split}
    est = make_estimator(xdeepfm.model_fn, params, use_graph=use_graph)
    batches = []
    for _ in range(steps):
        ids = synth_ids(rng, B, row_off)
        logx = np.log(np.floor(np.exp(rng.normal(2, 1, (B, 13)))) + 1.0).astype(np.float32)
        batches.append((ids, logx, rng.integers(0, 2, B).astype(np.float32)))

    def feats(ids, logx):
        return {"ids": torch.from_numpy(ids).cuda(), "cont_log": torch.from_numpy(logx).cuda()}

    est._call_model_fn(feats(*batches[0][:2]), None, ModeKeys.PREDICT)
    st = est.store
    w1 = np.zeros(int(row_off[-1]), np.float32)                     # oracle lin.wcat -> arena w1 (slot-order rows)
    for j in range(26):
        s = int(cat_slot[j])
        w1[row_off[s]:row_off[s + 1]] = P["lin.wcat"][cat_off[j]:cat_off[j + 1]]
    with torch.no_grad():
        st.embeddings["input_layer"].tables.copy_(torch.from_numpy(P["tables"]))
        st.embeddings["input_layer"].w1.copy_(torch.from_numpy(w1))
        st.embeddings["input_layer_1"].tables.copy_(torch.from_numpy(P["tables2"]))
    st.dense.load({k: v for k, v in P.items() if k in st.dense.params})
    # The CIN contraction sums ~F*H products per output; two fp32 implementations with different summation orders
    # (numpy einsum vs the MFMA k-order) differ by a few 1e-5 on the logits, so the reference here is the fp64 oracle.
    P = {k: v.astype(np.float64) for k, v in P.items()}
    om = models.XDeepFM(P, row_off, cat_slot, cat_off, cin, len(layers), dropout)
    opt = nn.AdamTF1(dtype=np.float64)
    err, losses = 0.0, []
    for ids, logx, y in batches:
        mk = None
        if dropout > 0:
            mk = [(rng.random((B, n)) >= dropout).astype(np.float32) for n in layers]
            est.params["_dropout_masks"] = [torch.from_numpy(m).cuda() for m in mk]
        f = feats(ids, logx)
        with torch.no_grad():
            pg = est._call_model_fn(f, None, ModeKeys.PREDICT).predictions["prob"].cpu().numpy().reshape(-1)
        po = nn.sigmoid(om.forward(ids, logx.astype(np.float64), train=False))
        err = max(err, float(np.abs(pg - po).max()))
        lg = float(est._train_step(f, torch.from_numpy(y).cuda()))
        lo, _ = models.train_step(om, opt, (ids, logx.astype(np.float64)), y.astype(np.float64),
                                  {"masks": [m.astype(np.float64) for m in mk]} if mk else None)
        losses.append((lg, float(lo)))
    a1 = st.embeddings["input_layer"]
    w1g = a1.w1.cpu().numpy()
    wcat = np.concatenate([w1g[row_off[int(cat_slot[j])]:row_off[int(cat_slot[j]) + 1]] for j in range(26)])
    perr = {"tables": float(np.abs(a1.tables.cpu().numpy() - P["tables"]).max()),
            "tables2": float(np.abs(st.embeddings["input_layer_1"].tables.cpu().numpy() - P["tables2"]).max()),
            "lin.wcat": float(np.abs(wcat - P["lin.wcat"]).max())}
    for k, p in st.dense.params.items():
        perr[k] = float(np.abs(p.detach().cpu().numpy() - P[k].reshape(p.shape)).max())
    return err, losses, perr


@pytest.mark.parametrize("cin,dropout,B", [((8, 4), 0.0, 16), ((20, 10, 10), 0.5, 24)])
def test_xdeepfm_train_parity(cin, dropout, B):
    err, losses, perr = _xdeepfm_run(B=B, steps=3, seed=21, cin=cin, layers=(32, 16), dropout=dropout, cin_split=0)   # fp32 MFMA kernels
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5, losses
    assert max(perr.values()) < 5e-5, perr


@pytest.mark.parametrize("D,cin,layers", [(8, (12, 6), (20, 10)), (16, (136, 8), (24, 12)), (32, (10, 10), (30, 18))])
def test_xdeepfm_generic_path_covers_the_flag_envelope(D, cin, layers):
    """--embedding_size != 16, a CIN width above 128 or a tower width that is no multiple of 4 (xdeepfm/xdeepfm.py:12-19 accepts
    them all) train through the generic path -- the same gather / scatter / TF-1 Adam kernels under autograd, the CIN layer and
    the tower as library GEMMs -- with the same parity bar against the fp64 oracle."""
    err, losses, perr = _xdeepfm_run(B=16, steps=3, seed=21 + D, cin=cin, layers=layers, dropout=0.5, D=D)
    assert err < 2e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 2e-5, losses
    assert max(perr.values()) < 1e-4, perr


def test_xdeepfm_train_parity_config3():
    """BASELINE config 3: xdeepfm.py Criteo d=16, CIN [128,128], DNN 100-100 (batch reduced for the numpy oracle)."""
    err, losses, perr = _xdeepfm_run(B=64, steps=2, seed=22, cin=(128, 128), layers=(100, 100), dropout=0.5, cin_split=0)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 2e-5, losses
    assert max(perr.values()) < 5e-5, perr


@pytest.mark.parametrize("cin_split", [3, 4])
@pytest.mark.parametrize("cin,dropout,B,steps", [((8, 4), 0.0, 16, 3), ((20, 10, 10), 0.5, 24, 3), ((128, 128), 0.5, 64, 2)])
def test_xdeepfm_train_parity_split3(cin, dropout, B, steps, cin_split):
    """The same parity bars with the CIN contraction on the 16-bit matrix cores: three bf16 planes per operand
    (csrc/cin_split.hip, cin_split = 3: every product exact to 2^-23) and cin_split = 4 (xdeepfm.py's default: forward / data
    gradients with two scaled fp16 planes per operand, weight gradients on three bf16 planes): predictions 1e-5, losses 1e-5 /
    2e-5, variables 5e-5 against the fp64 oracle -- incl. BASELINE config 3's CIN [128,128], DNN 100-100."""
    err, losses, perr = _xdeepfm_run(B=B, steps=steps, seed=21 if B < 64 else 22, cin=cin, layers=(32, 16) if B < 64 else (100, 100),
                                     dropout=dropout, cin_split=cin_split)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < (1e-5 if B < 64 else 2e-5), losses
    assert max(perr.values()) < 5e-5, perr


@pytest.mark.parametrize("cin_split", [0, 3, 4])
def test_xdeepfm_hip_graph(cin_split):
    err, losses, perr = _xdeepfm_run(B=32, steps=5, seed=23, cin=(16, 16), layers=(32, 16), dropout=0.0, use_graph=True, cin_split=cin_split)
    assert err < 1e-5 and max(perr.values()) < 5e-5, (err, perr)


@pytest.mark.parametrize("B,F,sizes", [(7, 39, (128, 128)), (5, 6, (20, 10, 10)), (3, 9, (33,)), (64, 39, (40, 8))])
def test_cin_net_fused_head(B, F, sizes):
    """CinNet (layers + the concat / reduce_sum / dense(relu) head, xdeepfm.py:135-182) against the fp64 oracle chain."""
    from recsys_amd.ops import CinNet, DenseArena
    rng = np.random.default_rng(B + 10 * len(sizes))
    D = 16
    shapes, H = {}, F
    for k, n in enumerate(sizes):
        shapes[f"cin.W{k}"], shapes[f"cin.c{k}"] = (F * H, n), (n,)
        H = n
    shapes["cin.Wout"], shapes["cin.bout"] = (sum(sizes), 1), (1,)
    P = DenseArena(shapes)
    vals = {k: (rng.standard_normal(s) * (0.3 if k == "cin.Wout" else 0.1)).astype(np.float32) for k, s in shapes.items()}
    vals["cin.bout"] = np.asarray([0.2], np.float32)
    P.load(vals)
    X0 = (rng.standard_normal((B, F, D)) * 0.4).astype(np.float32)
    gy = rng.standard_normal(B).astype(np.float32)
    V = {k: v.astype(np.float64) for k, v in vals.items()}
    X64, Xs = X0.astype(np.float64), []
    Xs.append(X64)
    for k in range(len(sizes)):
        Xs.append(models.cin_layer_fwd(X64, Xs[-1], V[f"cin.W{k}"], V[f"cin.c{k}"]))
    res = np.concatenate(Xs[1:], 1).sum(-1)
    y_o = nn.dense_fwd(res, V["cin.Wout"], V["cin.bout"], relu=True)
    dres, dWout_o, dbout_o = nn.dense_bwd(res, V["cin.Wout"], y_o, gy.astype(np.float64)[:, None], relu=True)
    g_o, dX0_o, off, dnext = {}, np.zeros_like(X64), np.cumsum((0,) + tuple(sizes)), None
    for k in range(len(sizes) - 1, -1, -1):
        dout = np.repeat(dres[:, off[k]:off[k + 1], None], D, 2)
        if dnext is not None:
            dout = dout + dnext
        d0, dnext, g_o[f"cin.W{k}"], g_o[f"cin.c{k}"] = models.cin_layer_bwd(X64, Xs[k], V[f"cin.W{k}"], Xs[k + 1], dout)
        dX0_o += d0
    dX0_o += dnext
    net = CinNet(F, D, sizes, B + 3)
    tx = torch.from_numpy(X0).cuda()
    y = net.forward(tx, P)
    np.testing.assert_allclose(y.cpu().numpy(), y_o.reshape(-1), rtol=2e-5, atol=2e-5)
    dX0 = net.backward(tx, P, torch.from_numpy(gy).cuda())
    tol = dict(rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dX0.cpu().numpy(), dX0_o, **tol)
    np.testing.assert_allclose(P["cin.Wout"].grad.cpu().numpy(), dWout_o, **tol)
    np.testing.assert_allclose(P["cin.bout"].grad.cpu().numpy(), dbout_o.reshape(1), **tol)
    for k in range(len(sizes)):
        np.testing.assert_allclose(P[f"cin.W{k}"].grad.cpu().numpy(), g_o[f"cin.W{k}"], **tol)
        np.testing.assert_allclose(P[f"cin.c{k}"].grad.cpu().numpy(), g_o[f"cin.c{k}"], **tol)


@pytest.mark.parametrize("B", [48, 2500])
def test_two_table_sets_in_one_scatter_launch_is_bit_identical(B):
    """rsx_segsum_adam_rows(second_h): xDeepFM's two table sets (one shared sort) updated by ONE launch == two launches."""
    from recsys_amd.ops import AdamTF1, DenseArena, EmbeddingArena
    rng = np.random.default_rng(B)
    row_off = np.concatenate([[0], np.cumsum(rng.integers(3, 60, 7))]).astype(np.int64)
    F, D, R = 7, 16, int(row_off[-1])
    ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
    t1, t2 = rng.standard_normal((R, D)).astype(np.float32), rng.standard_normal((R, D)).astype(np.float32)
    w1 = rng.standard_normal(R).astype(np.float32)
    dX1 = torch.from_numpy(rng.standard_normal((B, F * D)).astype(np.float32)).cuda()
    dX2 = torch.from_numpy(rng.standard_normal((B, F * D)).astype(np.float32)).cuda()
    g1 = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).cuda()
    res = []
    for merged in (False, True):
        a1 = EmbeddingArena(row_off, D, B, "cuda", with_w1=True, tables=t1.copy(), w1=w1.copy())
        a2 = EmbeddingArena(row_off, D, B, "cuda", with_w1=False, tables=t2.copy())
        a2.share_sort_of(a1)
        dense = DenseArena({"w": (33,)})
        dense.grad.copy_(torch.arange(dense.n, device="cuda") * 0.01)
        opt = AdamTF1(device="cuda")
        for _ in range(3):
            a1.field_sort(ids)
            a2.field_sort(ids)
            if merged:
                a1.segsum_adam(B, None, dX1, g1, None, opt, dense.adam_segments(), None, second=(a2, dX2))
            else:
                a1.segsum_adam(B, None, dX1, g1, None, opt, [], None, advance=False)
                a2.segsum_adam(B, None, dX2, None, None, opt, dense.adam_segments(), None)
            dense.grad.copy_(torch.arange(dense.n, device="cuda") * 0.01)
        res.append([x.cpu().numpy().copy() for x in (a1.tables, a1.m_t, a1.v_t, a1.w1, a2.tables, a2.m_t, a2.v_t, dense.flat, opt.state[:4])])
    for x, y in zip(*res):
        np.testing.assert_array_equal(x, y)
