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


def _din_run(B, Pn, K, n_item, n_cate, steps, seed, dropout, use_graph=False):
    from recsys_amd import din, synthetic
    from recsys_amd.estimator import ModeKeys
    from tests.parity_util import make_estimator
    rng = np.random.default_rng(seed)
    P = init.din_params(seed, K, n_item, n_cate, np.float32)
    P["item_bias"] += (rng.standard_normal(n_item) * 0.01).astype(np.float32)
    params = {"embedding_size": K, "learning_rate": 1e-3, "dropout": dropout, "max_batch_size": B, "n_item": n_item,
              "n_cate": n_cate}
    est = make_estimator(din.model_fn, params, use_graph=use_graph)
    batches = [synthetic.din_batch(rng, B, Pn, n_item, n_cate) for _ in range(steps)]

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
            mk = {"att_i": [(rng.random((B * Pn, n)) >= dropout).astype(np.float32) for n in (80, 40)],
                  "att_c": [(rng.random((B * Pn, n)) >= dropout).astype(np.float32) for n in (80, 40)],
                  "mlp": [(rng.random((B, n)) >= dropout).astype(np.float32) for n in (100, 50, 20)]}
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
