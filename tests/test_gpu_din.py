"""DIN (BASELINE config 5) parity: pooling kernel op-level, sparse-row segment machinery, full train steps vs the oracle."""
import numpy as np
import pytest

from oracle import init, models, nn

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("B,P,K", [(7, 13, 32), (64, 100, 32), (5, 9, 16), (3, 4, 4), (9, 70, 64)])
def test_din_pool_fwd_bwd(B, P, K):
    from recsys_amd.ops import DinPoolFn
    rng = np.random.default_rng(B + P)
    H = rng.standard_normal((B, P, K)).astype(np.float32)
    w = rng.standard_normal((B, P)).astype(np.float32)
    ids = rng.integers(0, 50, (B, P)).astype(np.int32)
    for b in range(B):
        ids[b, rng.integers(1, P + 1):] = 0                      # ragged, zero padded
    g = rng.standard_normal((B, K)).astype(np.float32)
    mask = (ids > 0).astype(np.float32)
    out_o = (H * w[:, :, None] * mask[:, :, None]).sum(1)
    dH_o = g[:, None, :] * (w * mask)[:, :, None]
    dw_o = (H * g[:, None, :]).sum(2) * mask
    tH, tw = torch.from_numpy(H).cuda().requires_grad_(), torch.from_numpy(w).cuda().requires_grad_()
    out = DinPoolFn.apply(tH, tw, torch.from_numpy(ids).cuda())
    out.backward(torch.from_numpy(g).cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_o, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tH.grad.cpu().numpy(), dH_o, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(tw.grad.cpu().numpy(), dw_o, rtol=1e-5, atol=1e-5)


def test_sparse_table_segments_and_ordered_sums():
    from recsys_amd.ops import SparseTable
    rng = np.random.default_rng(0)
    R, K = 300, 32
    tbl = SparseTable(R, K, 5000, table=rng.standard_normal((R, K)).astype(np.float32))
    for rep in range(2):
        ids1 = rng.integers(0, R, 40).astype(np.int32)
        ids2 = (rng.zipf(1.3, (30, 100)) % R).astype(np.int32)
        t1, t2 = torch.from_numpy(ids1).cuda(), torch.from_numpy(ids2).cuda()
        r1, r2 = tbl.lookup(t1), tbl.lookup(t2)
        assert np.array_equal(r1.detach().cpu().numpy(), tbl.table.cpu().numpy()[ids1])
        g1 = rng.standard_normal((40, K)).astype(np.float32)
        g2 = rng.standard_normal((30, 100, K)).astype(np.float32)
        (r2 * torch.from_numpy(g2).cuda()).sum().backward(retain_graph=False)     # registers second lookup first
        (r1 * torch.from_numpy(g1).cuda()).sum().backward()
        tbl.finalize()
        torch.cuda.synchronize()
        rows = np.concatenate([ids1, ids2.reshape(-1)])
        vals = np.concatenate([g1, g2.reshape(-1, K)])
        uniq, G = nn.segment_sum_rows(rows, vals)
        U = int(tbl.nuniq.item())
        assert U == len(uniq) and np.array_equal(tbl.uniq_row.cpu().numpy()[:U], uniq)
        slot = tbl.slot.cpu().numpy()[:R]
        assert np.array_equal(np.flatnonzero(slot >= 0), uniq) and np.array_equal(slot[uniq], np.arange(U))
        cnt = np.bincount(np.searchsorted(uniq, rows), minlength=U)
        Gg = tbl.G.cpu().numpy()[:U]
        # entry-order sums: bit-exact up to 64/(K/4) = 8 entries (one entry per lane group); longer segments are
        # summed as ordered sub-range partials (deterministic, fp32-tolerance vs the sequential order)
        assert np.array_equal(Gg[cnt <= 8], G[cnt <= 8])
        np.testing.assert_allclose(Gg, G, rtol=2e-5, atol=1e-6)


def _din_run(B, Pn, K, n_item, n_cate, steps, seed, dropout, use_graph=False, zero_targets=0, extra_params=None,
             batch_sizes=None):
    from recsys_amd import din, synthetic
    from recsys_amd.estimator import ModeKeys
    from tests.parity_util import make_estimator
    rng = np.random.default_rng(seed)
    P = init.din_params(seed, K, n_item, n_cate, np.float32)
    P["item_bias"] += (rng.standard_normal(n_item) * 0.01).astype(np.float32)
    params = {"embedding_size": K, "learning_rate": 1e-3, "dropout": dropout, "max_batch_size": B, "n_item": n_item,
              "n_cate": n_cate}
    params.update(extra_params or {})
    est = make_estimator(din.model_fn, params, use_graph=use_graph)
    sizes = list(batch_sizes) if batch_sizes is not None else [B] * steps      # (B = the capacity, sizes[i] <= B)
    batches = [synthetic.din_batch(rng, bs, Pn, n_item, n_cate) for bs in sizes]
    for b in batches:              # target item / category id 0: an ordinary row for tf.gather (din/din.py:96-101)
        b["i_id"][:zero_targets] = 0
        b["i_cate"][:max(zero_targets - 1, 0)] = 0

    def feats(b):
        return {k: torch.from_numpy(b[k]).cuda() for k in ("i_id", "i_cate", "u_iid_seq", "u_icat_seq")}

    est._call_model_fn(feats(batches[0]), None, ModeKeys.PREDICT)
    st = est.store
    with torch.no_grad():
        st.embeddings["i_id"].table.copy_(torch.from_numpy(P["item_emb"]))
        st.embeddings["i_cate"].table.copy_(torch.from_numpy(P["cate_emb"]))
        st.embeddings["i_item"].table[:, 0].copy_(torch.from_numpy(P["item_bias"]))
    st.dense.load({k: v for k, v in P.items() if k in st.dense.params})
    om = models.DIN(P, dropout)
    opt = nn.AdamTF1(dtype=np.float32)
    err, losses = 0.0, []
    for b in batches:
        mk = None
        if dropout > 0:
            nb = len(b["i_id"])
            mk = {"att_i": [(rng.random((nb * Pn, n)) >= dropout).astype(np.float32) for n in (80, 40)],
                  "att_c": [(rng.random((nb * Pn, n)) >= dropout).astype(np.float32) for n in (80, 40)],
                  "mlp": [(rng.random((nb, n)) >= dropout).astype(np.float32) for n in (100, 50, 20)]}
            est.params["_dropout_masks"] = {k: [torch.from_numpy(m).cuda() for m in v] for k, v in mk.items()}
        f = feats(b)
        with torch.no_grad():
            pg = est._call_model_fn(f, None, ModeKeys.PREDICT).predictions["prob"].cpu().numpy()
        args = (b["i_id"], b["i_cate"], b["u_iid_seq"], b["u_icat_seq"])
        po = nn.sigmoid(om.forward(*args, train=False))
        err = max(err, float(np.abs(pg - po).max()))
        lg = float(est._train_step(f, torch.from_numpy(b["label"]).cuda()))
        lo, _ = models.train_step(om, opt, args, b["label"], {"masks": mk} if mk else None)
        losses.append((lg, float(lo)))
    perr = {"item_emb": float(np.abs(st.embeddings["i_id"].table.cpu().numpy() - P["item_emb"]).max()),
            "cate_emb": float(np.abs(st.embeddings["i_cate"].table.cpu().numpy() - P["cate_emb"]).max()),
            "item_bias": float(np.abs(st.embeddings["i_item"].table[:, 0].cpu().numpy() - P["item_bias"]).max())}
    for k, p in st.dense.params.items():
        perr[k] = float(np.abs(p.detach().cpu().numpy() - P[k].reshape(p.shape)).max())
    return err, losses, perr


@pytest.mark.parametrize("dropout", [0.0, 0.5])
def test_din_train_parity_small(dropout):
    err, losses, perr = _din_run(B=24, Pn=12, K=16, n_item=200, n_cate=20, steps=4, seed=3, dropout=dropout)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5, losses
    assert max(perr.values()) < 5e-5, perr


@pytest.mark.parametrize("fused", [True, False])
def test_din_target_id_zero_trains_like_any_other_row(fused):
    """din/din.py:95-107 masks id 0 in the HISTORIES only; a target item / category 0 gets its gradient (both the fused step
    and the autograd path key the history padding to a dummy row instead of skipping row 0)."""
    err, losses, perr = _din_run(B=24, Pn=12, K=16, n_item=200, n_cate=20, steps=4, seed=7, dropout=0.0, zero_targets=5,
                                 extra_params={"fused_step": fused})
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5, losses
    assert max(perr.values()) < 5e-5, perr


@pytest.mark.parametrize("use_graph", [False, True])
def test_din_vocabulary_multiple_of_32_and_shrinking_batches(use_graph):
    """(a) the item bias rides through the item field's scatter indexed by GLOBAL arena rows (the dummy row n_item
    and the category rows included): with n_item % 32 == 0 an n_item-row allocation has no slack and the dummy row's update
    landed in the first-moment slot of bias row 0; (b) a batch smaller than an earlier one (the last batch of an epoch) must
    not pick up the larger batch's bias gradients at its history entries.  Every step is held against the oracle."""
    err, losses, perr = _din_run(B=32, Pn=12, K=16, n_item=256, n_cate=32, steps=6, seed=11, dropout=0.0, use_graph=use_graph,
                                 batch_sizes=[32, 20, 32, 7, 20, 32])
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5, losses
    assert max(perr.values()) < 5e-5, perr


def test_din_train_parity_config5_shape():
    """BASELINE config 5 shape: Amazon-Electronics vocabularies, hist_len 100, K=32 (batch reduced so the numpy
    oracle finishes in seconds; the bs=1024 run is a bench / property test)."""
    err, losses, perr = _din_run(B=128, Pn=100, K=32, n_item=63002, n_cate=802, steps=2, seed=4, dropout=0.5)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 2e-5, losses
    assert max(perr.values()) < 5e-5, perr


def test_din_hip_graph_steps():
    err, losses, perr = _din_run(B=32, Pn=20, K=32, n_item=500, n_cate=30, steps=6, seed=5, dropout=0.0, use_graph=True)
    assert err < 1e-5 and max(perr.values()) < 5e-5, (err, perr)


@pytest.mark.parametrize("B,P,K,rate", [(3, 5, 16, 0.0), (7, 13, 32, 0.5), (64, 100, 32, 0.5), (300, 70, 32, 0.0)])
def test_fused_attention_mlp_matches_reference_expression(B, P, K, rate):
    """rsx_din_attn_fwd / _bwd (one launch per direction) vs the reference expression of din/din.py:111-121 evaluated
    in fp64 by torch autograd: logits and every gradient (history rows, query, 3 weight matrices, 3 biases), with
    injected dropout masks.  Tolerance 1e-5 relative to the largest reference entry (fp32, north_star)."""
    from recsys_amd.ops import DinAttnFn
    g = torch.Generator(device="cuda").manual_seed(B * 131 + P)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    M = B * P
    H, q = rn(B, P, K).requires_grad_(), rn(B, K).requires_grad_()
    W0, b0 = (rn(4 * K, 80) * 0.1).requires_grad_(), (rn(80) * 0.1).requires_grad_()
    W1, b1 = (rn(80, 40) * 0.1).requires_grad_(), (rn(40) * 0.1).requires_grad_()
    W2, b2 = (rn(40, 1) * 0.1).requires_grad_(), rn(1).requires_grad_()
    masks = None
    if rate > 0:
        masks = [(torch.rand(M, n, device="cuda", generator=g) >= rate).float() for n in (80, 40)]
    dw = rn(B, P)
    w = DinAttnFn.apply(H, q, W0, b0, W1, b1, W2, b2, rate, masks, None, 0, 0)
    (w * dw).sum().backward()
    got = [w.detach()] + [t.grad.clone() for t in (H, q, W0, b0, W1, b1, W2, b2)]
    t64 = [t.detach().double().requires_grad_() for t in (H, q, W0, b0, W1, b1, W2, b2)]
    Hd, qd, W0d, b0d, W1d, b1d, W2d, b2d = t64
    hh = Hd.reshape(M, K)
    qq = qd[:, None, :].expand(B, P, K).reshape(M, K)
    x = torch.cat([hh, qq, hh * qq, hh - qq], 1)
    a = torch.relu(x @ W0d + b0d)
    a = a * (masks[0].double() / (1 - rate)) if masks else a
    a = torch.relu(a @ W1d + b1d)
    a = a * (masks[1].double() / (1 - rate)) if masks else a
    wr = (a @ W2d + b2d).reshape(B, P)
    (wr * dw.double()).sum().backward()
    ref = [wr.detach()] + [t.grad for t in t64]
    for name, a_, b_ in zip(("w", "dH", "dq", "dW0", "db0", "dW1", "db1", "dW2", "db2"), got, ref):
        err = float((a_.double() - b_).abs().max() / (b_.abs().max() + 1e-30))
        assert err < 1e-5, (name, err)


def test_fused_attention_rng_dropout_consistent_between_forward_and_backward():
    """Hash-RNG dropout (no mask buffers): the backward pass must regenerate exactly the forward's keep pattern.  With
    W2 = 1, b = 0 and non-negative pre-activations, d w / d b1[n] counts the kept (row, n) pairs scaled by 1/keep, which
    must equal the forward's sum of a2-dropout indicators recovered from w itself."""
    from recsys_amd.ops import DinAttnFn
    B, P, K, rate = 16, 20, 32, 0.5
    g = torch.Generator(device="cuda").manual_seed(5)
    H = torch.rand(B, P, K, device="cuda", generator=g).requires_grad_()
    q = torch.rand(B, K, device="cuda", generator=g).requires_grad_()
    W0 = (torch.rand(4 * K, 80, device="cuda", generator=g) * 0.01).requires_grad_()
    b0 = torch.ones(80, device="cuda").requires_grad_()
    W1 = torch.zeros(80, 40, device="cuda").requires_grad_()
    b1 = torch.ones(40, device="cuda").requires_grad_()          # a2 == 1 everywhere before dropout
    W2 = torch.ones(40, 1, device="cuda").requires_grad_()
    b2 = torch.zeros(1, device="cuda").requires_grad_()
    step = torch.tensor([7], dtype=torch.int32, device="cuda")
    w = DinAttnFn.apply(H, q, W0, b0, W1, b1, W2, b2, rate, None, step, 123, 0)
    kept = w.detach() * (1 - rate)                               # number of kept units of layer 2 per row
    assert float(kept.min()) >= 0 and abs(float(kept.mean()) - 20.0) < 1.5      # ~ Binomial(40, 0.5)
    w.sum().backward()
    # d w / d b1[n] = sum_rows keep2[row, n] / (1 - rate)  ->  summed over n it must reproduce sum_rows w
    assert abs(float(b1.grad.sum()) - float(w.detach().sum())) < 1e-2 * float(w.detach().sum())
    w2 = DinAttnFn.apply(H, q, W0, b0, W1, b1, W2, b2, rate, None, step, 123, 0)
    assert torch.equal(w2.detach(), w.detach())                  # same (seed, step, layer) -> same pattern
    step += 1
    w3 = DinAttnFn.apply(H, q, W0, b0, W1, b1, W2, b2, rate, None, step, 123, 0)
    assert not torch.equal(w3.detach(), w.detach())              # next step -> new pattern


@pytest.mark.parametrize("B,P", [(9, 20), (40, 100), (3, 1)])
def test_valid_rows_list_is_exact(B, P):
    """rsx_din_valid_rows: ascending list of the non-padding positions, their count, w zeroed at the padded ones."""
    import ctypes as C
    from recsys_amd.ops import _ptr, _stream, check, lib
    rng = np.random.default_rng(B * P)
    ids = rng.integers(0, 4, (B, P)).astype(np.int32)          # ~25 % padding anywhere (not only as a suffix)
    ids[0] = 0                                                 # an all-padding example
    t = torch.from_numpy(ids).cuda()
    rows = torch.full((B * P + 2 + (B * P + 1023) // 1024,), -7, dtype=torch.int32, device="cuda")
    w = torch.full((B, P), 3.0, device="cuda")
    check(lib().rsx_din_valid_rows(_ptr(t), B, P, _ptr(rows), _ptr(rows[B * P:]), _ptr(w), _stream()))
    want = np.flatnonzero(ids.reshape(-1) > 0).astype(np.int32)
    r = rows.cpu().numpy()
    assert int(r[B * P]) == len(want)
    np.testing.assert_array_equal(r[:len(want)], want)
    np.testing.assert_array_equal(w.cpu().numpy(), np.where(ids > 0, 3.0, 0.0).astype(np.float32))


def test_attention_over_the_valid_rows_equals_attention_over_all_rows():
    """DinAttnPoolFn (row list: padded positions skipped) == DinAttnFn + DinPoolFn over every position: same pooled output,
    same dH / dq / weight gradients (the skipped rows contribute exact zeros; only the summation grouping differs)."""
    from recsys_amd.ops import DinAttnFn, DinAttnPoolFn, DinPoolFn
    rng = np.random.default_rng(5)
    B, P, K, N1, N2 = 12, 37, 32, 80, 40
    H = (rng.standard_normal((B, P, K)) * 0.5).astype(np.float32)
    q = (rng.standard_normal((B, K)) * 0.5).astype(np.float32)
    lens = rng.integers(1, P + 1, B)
    hist = rng.integers(1, 50, (B, P)).astype(np.int32)
    hist[np.arange(P)[None, :] >= lens[:, None]] = 0
    Ws = [rng.standard_normal(s).astype(np.float32) * 0.2 for s in ((4 * K, N1), (N1,), (N1, N2), (N2,), (N2, 1), (1,))]
    g = rng.standard_normal((B, K)).astype(np.float32)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = []
    for fused in (True, False):
        tH, tq = torch.from_numpy(H).cuda().requires_grad_(), torch.from_numpy(q).cuda().requires_grad_()
        tW = [torch.from_numpy(x).cuda().requires_grad_() for x in Ws]
        th = torch.from_numpy(hist).cuda()
        gout = torch.zeros(sum(x.size for x in Ws), device="cuda")
        if fused:
            out = DinAttnPoolFn.apply(tH, tq, th, *tW, 0.0, None, step, 1, 0, gout)
        else:
            out = DinPoolFn.apply(tH, DinAttnFn.apply(tH, tq, *tW, 0.0, None, step, 1, 0), th)
        out.backward(torch.from_numpy(g).cuda())
        wg = gout.cpu().numpy() if fused else np.concatenate([x.grad.cpu().numpy().reshape(-1) for x in tW])
        res.append((out.detach().cpu().numpy(), tH.grad.cpu().numpy(), tq.grad.cpu().numpy(), wg))
    for a, b, name in zip(res[0], res[1], ("out", "dH", "dq", "weights")):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5, err_msg=name)


def test_gather_rows_multi_equals_indexing():
    """rsx_gather_rows_multi: several tf.gather / embedding_lookup calls (din/din.py:96-105) in one launch, with output row
    strides (a lookup written straight into its column slice of the 'mlp_layer' input)."""
    import ctypes as C
    from recsys_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    t32 = torch.randn(500, 32, device="cuda")
    t16 = torch.randn(70, 16, device="cuda")
    ids_a = torch.from_numpy(rng.integers(0, 500, 130).astype(np.int32)).cuda()
    ids_b = torch.from_numpy(rng.integers(0, 70, 999).astype(np.int32)).cuda()
    ids_c = torch.from_numpy(rng.integers(0, 400, 57).astype(np.int32)).cuda()
    X = torch.full((130, 96), -7.0, device="cuda")
    o_b = torch.empty(999, 16, device="cuda")
    o_c = torch.empty(57, 32, device="cuda")
    t4 = torch.randn(500, 4, device="cuda")                      # a 1-D variable stored as column 0 of a 4-wide table
    o_s = torch.empty(130, device="cuda")
    jobs = (_lib.GatherJob * 4)()
    for j, (tab, ids, out, k, ld, base, ldt) in zip(jobs, ((t32, ids_a, X[:, 32:], 32, 96, 0, 0), (t16, ids_b, o_b, 16, 16, 0, 0),
                                                          (t32, ids_c, o_c, 32, 32, 100, 0), (t4, ids_a, o_s, 1, 1, 0, 4))):
        j.table, j.ids, j.out, j.n, j.K, j.ld_out, j.row_base, j.ld_table = \
            tab.data_ptr(), ids.data_ptr(), out.data_ptr(), ids.shape[0], k, ld, base, ldt
    assert L.rsx_gather_rows_multi(jobs, 4, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    assert torch.equal(o_s, t4[ids_a.long(), 0])
    assert torch.equal(X[:, 32:64], t32[ids_a.long()]) and bool((X[:, :32] == -7).all()) and bool((X[:, 64:] == -7).all())
    assert torch.equal(o_b, t16[ids_b.long()])
    assert torch.equal(o_c, t32[ids_c.long() + 100])


@pytest.mark.parametrize("B,P", [(3, 5), (64, 100), (1024, 100)])
def test_din_prepare_keys_and_row_lists_are_exact(B, P):
    """rsx_din_prepare: the sort keys of both id tables (history padding -> the tables' dummy rows, target ids untouched -- a
    target id 0 stays row 0) and both histories' ascending lists of non-padding positions, against numpy."""
    import ctypes as C
    from recsys_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(B)
    i_id = rng.integers(0, 50, B).astype(np.int32)
    i_cate = rng.integers(0, 9, B).astype(np.int32)
    hi = rng.integers(1, 50, (B, P)).astype(np.int32)
    hc = rng.integers(1, 9, (B, P)).astype(np.int32)
    for b in range(B):
        n = rng.integers(0, P + 1)
        hi[b, n:] = 0
        hc[b, rng.integers(0, P + 1):] = 0          # (the two histories' masks need not agree)
    dev = lambda a: torch.from_numpy(a).cuda()
    d = [dev(x) for x in (i_id, i_cate, hi, hc)]
    M, N = B * P, B * (P + 1)
    keys2 = torch.zeros(N, 2, dtype=torch.int32, device="cuda")
    rows = [torch.zeros(M + 2 + (M + 1023) // 1024, dtype=torch.int32, device="cuda") for _ in range(2)]
    w = [torch.ones(B, P, device="cuda") for _ in range(2)]
    p = lambda t: C.c_void_p(t.data_ptr())
    assert L.rsx_din_prepare(p(d[0]), p(d[1]), p(d[2]), p(d[3]), B, P, 50, 9, p(keys2), p(rows[0]), p(rows[0][M:]), p(w[0]),
                             p(rows[1]), p(rows[1][M:]), p(w[1]), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    want = np.zeros((N, 2), np.int32)
    want[:B, 0], want[:B, 1] = i_id, i_cate
    want[B:, 0] = np.where(hi.reshape(-1) > 0, hi.reshape(-1), 50)
    want[B:, 1] = np.where(hc.reshape(-1) > 0, hc.reshape(-1), 9)
    assert np.array_equal(keys2.cpu().numpy(), want)
    for t, h in enumerate((hi, hc)):
        valid = np.flatnonzero(h.reshape(-1) > 0)
        r = rows[t].cpu().numpy()
        assert r[M] == len(valid) and np.array_equal(r[:len(valid)], valid)
        assert np.array_equal(w[t].cpu().numpy().reshape(-1) == 0, h.reshape(-1) <= 0)
    # rsx_din_prepare2: the same keys FIELD-MAJOR (what rsx_field_sort_large_t takes) + the labels' cast to float32
    stride = N + 5
    keys_t = torch.zeros(2, stride, dtype=torch.int32, device="cuda")
    lab = torch.from_numpy(rng.integers(0, 2, B).astype(np.int64)).cuda()
    lab_f = torch.full((B,), -1.0, device="cuda")
    assert L.rsx_din_prepare2(p(d[0]), p(d[1]), p(d[2]), p(d[3]), B, P, 50, 9, p(keys_t), stride, p(rows[0]), p(rows[0][M:]),
                              p(w[0]), p(rows[1]), p(rows[1][M:]), p(w[1]), p(lab), p(lab_f),
                              C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(keys_t[:, :N].cpu().numpy(), want.T)
    assert np.array_equal(lab_f.cpu().numpy(), lab.cpu().numpy().astype(np.float32))


def test_large_sort_of_field_major_keys_equals_the_transposing_entry():
    """rsx_field_sort_large_t (keys already [F, stride]: no transpose launch, the buffer is clobbered) against
    rsx_field_sort_large on the same keys: every output of the dedup sort identical."""
    from recsys_amd.ops import EmbeddingArena
    rng = np.random.default_rng(3)
    rows_per = (700, 90)
    row_off = np.concatenate([[0], np.cumsum(rows_per)]).astype(np.int32)
    N = 20000
    outs = []
    keys = np.stack([rng.integers(0, r, N) for r in rows_per], 1).astype(np.int32)
    for transposed in (False, True):
        a = EmbeddingArena(row_off, 32, N + 7, "cuda")
        assert N > a.LDS_SORT_MAX_B
        if transposed:
            kt = torch.zeros(2, a.stride, dtype=torch.int32, device="cuda")
            kt[:, :N] = torch.from_numpy(np.ascontiguousarray(keys.T)).cuda()
            a.field_sort_t(kt, N)
        else:
            a.field_sort(torch.from_numpy(keys).cuda())
        torch.cuda.synchronize()
        nu = a.nuniq.cpu().numpy()
        outs.append((nu, a.perm.view(2, -1)[:, :N].cpu().numpy(), a.seg_off.view(2, -1).cpu().numpy(),
                     a.uniq_row.view(2, -1).cpu().numpy(), a.slot.cpu().numpy()))
    nu = outs[0][0]
    assert np.array_equal(nu, outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][4], outs[1][4])
    for f in range(2):
        assert np.array_equal(outs[0][2][f, :nu[f] + 1], outs[1][2][f, :nu[f] + 1])
        assert np.array_equal(outs[0][3][f, :nu[f]], outs[1][3][f, :nu[f]])


def test_din_eval_head_through_the_fused_tower_equals_the_torch_path():
    """EVAL / PREDICT of din.py: the MLP head through FusedTower.infer (dropout off, no batch-norm) against the torch layers."""
    from recsys_amd import din, synthetic
    from recsys_amd.estimator import ModeKeys
    from tests.parity_util import make_estimator
    rng = np.random.default_rng(2)
    B, Pn, K = 48, 20, 32
    out = []
    for fused in (True, False):
        params = {"embedding_size": K, "learning_rate": 1e-2, "dropout": 0.5, "max_batch_size": B, "n_item": 300, "n_cate": 20,
                  "fused_infer": fused}
        est = make_estimator(din.model_fn, params)
        r2 = np.random.default_rng(5)
        bs = [synthetic.din_batch(r2, B, Pn, 300, 20) for _ in range(4)]
        feats = lambda b: {k: torch.from_numpy(b[k]).cuda() for k in ("i_id", "i_cate", "u_iid_seq", "u_icat_seq")}
        for b in bs[:3]:
            est._train_step(feats(b), torch.from_numpy(b["label"]).cuda())
        with torch.no_grad():
            sp = est._call_model_fn(feats(bs[3]), torch.from_numpy(bs[3]["label"]).cuda(), ModeKeys.EVAL)
        out.append((sp.predictions["prob"].cpu().numpy().reshape(-1), float(sp.loss)))
    (p1, l1), (p2, l2) = out
    assert np.abs(p1 - p2).max() < 2e-6 and abs(l1 - l2) < 2e-6, (np.abs(p1 - p2).max(), l1, l2)
