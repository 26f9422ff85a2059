"""Data-parallel HIP path with DISTINCT per-rank batches on one GPU.  dist.LoopbackDataParallel runs
the N ranks of a synchronous step one after the other -- N different batches, shared variables, per-rank batch-norm /
dropout, the real send blocks concatenated into the gathered buffers, the optimizer stage once -- and the result is
compared with the ORACLE doing what tf.distribute.MirroredStrategy does (oracle.models.train_step_dp: N backward passes at
dz / N, dense gradients summed, IndexedSlices concatenated in replica order; fm/fm.py:184-194, deepfm/readme.md:24).
Unlike the EmulatedDataParallel tests (one batch tiled N times) a wrong rank stride, block offset or replica sum FAILS here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS = (3, 7, 40, 11, 600, 2500)


def _build(kind, world, B, layers, D, dropout, exchange, monkeypatch, rows=ROWS, window=False):
    import torch
    from oracle import init
    from recsys_amd import dcn, deepfm, fm
    from tests.dp_harness import LoopbackDataParallel
    from tests.parity_util import load_oracle_weights, make_estimator, small_columns
    monkeypatch.setenv("RSX_DP_EXCHANGE", exchange)
    row_off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    lin, emb = small_columns(rows, D)
    mfn = {"deepfm": deepfm.model_fn, "dcn": dcn.model_fn, "fm": fm.model_fn}[kind]
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": D, "learning_rate": 1e-3,
              "dropout": dropout, "deep_layers": ",".join(map(str, layers)), "cross_layers": 2, "max_batch_size": B}
    if kind in ("deepfm", "fm"):
        P = init.deepfm_params(3, D, layers if kind == "deepfm" else (), np.float32, row_off, with_dnn=kind != "fm")
    else:
        P = init.dcn_params(3, D, layers, 2, np.float32, row_off)
    if "b1" in P:
        P["b1"] += np.float32(0.05)
    est = make_estimator(mfn, params)
    est.store.dp = LoopbackDataParallel(world)
    with torch.no_grad():
        est._call_model_fn({"ids": torch.zeros(B, len(rows), dtype=torch.int32, device="cuda")}, None, "infer")
    load_oracle_weights(est, P)
    return est, P, row_off


def _oracle_model(kind, P64, row_off, layers, dropout):
    from oracle import models
    return {"dcn": lambda: models.DCN(P64, row_off, len(layers), dropout), "fm": lambda: models.FM(P64, row_off),
            "deepfm": lambda: models.DeepFM(P64, row_off, len(layers), dropout)}[kind]()


def _param_err(est, P64):
    a = est.store.embeddings["input_layer"]
    perr = {"tables": float(np.abs(a.tables.cpu().numpy() - P64["tables"]).max())}
    if a.with_w1:
        perr["w1"] = float(np.abs(a.w1.cpu().numpy() - P64["w1"]).max())
    for k, p in est.store.dense.params.items():
        perr[k] = float(np.abs(p.detach().cpu().numpy() - P64[k].reshape(p.shape)).max())
    return perr


@pytest.mark.parametrize("exchange", ["examples", "unique"])
@pytest.mark.parametrize("kind,world,B,dropout", [("deepfm", 2, 48, 0.5), ("deepfm", 3, 40, 0.0), ("dcn", 2, 56, 0.5),
                                                  ("fm", 3, 64, 0.0), ("deepfm", 4, 300, 0.0), ("dcn", 4, 600, 0.0),
                                                  ("deepfm", 2, 1100, 0.0), ("dcn", 2, 2200, 0.0)])
def test_loopback_dp_with_distinct_batches_matches_the_oracle(kind, world, B, dropout, exchange, monkeypatch):
    import torch
    from oracle import models, nn
    from tests.dp_harness import loopback_train_step
    from tests.parity_util import synth_ids
    layers, D = ((32, 16), 16) if kind != "fm" else ((), 16)
    est, P, row_off = _build(kind, world, B, layers, D, dropout, exchange, monkeypatch)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    om = _oracle_model(kind, P64, row_off, layers, dropout)
    opt = nn.AdamTF1(dtype=np.float64)
    rng = np.random.default_rng(11)
    for step in range(4):
        ids = [synth_ids(rng, B, row_off) for _ in range(world)]
        ys = [(rng.random(B) < 0.3).astype(np.float32) for _ in range(world)]
        mk = None
        if dropout > 0.0 and len(layers):
            mk = [[(rng.random((B, n)) >= dropout).astype(np.float32) for n in layers] for _ in range(world)]

        def before(r):
            est.params["_dropout_masks"] = None if mk is None else [torch.from_numpy(m).cuda() for m in mk[r]]

        lg = loopback_train_step(est, [{"ids": torch.from_numpy(i).cuda()} for i in ids],
                                 [torch.from_numpy(y).cuda() for y in ys], before_rank=before)
        lo, _ = models.train_step_dp(om, opt, [(i,) for i in ids], [y.astype(np.float64) for y in ys],
                                     None if mk is None else [{"masks": [m.astype(np.float64) for m in mk[r]]} for r in range(world)])
        assert all(abs(a - b) < 1e-5 for a, b in zip(lg, lo)), (step, lg, lo)
    perr = _param_err(est, P64)
    assert max(perr.values()) < 2e-5, perr
    # eval forward of the trained variables on a fresh batch: within 1e-5 of the oracle's
    ids = synth_ids(rng, B, row_off)
    with torch.no_grad():
        zg = est._call_model_fn({"ids": torch.from_numpy(ids).cuda()}, None, "infer").predictions["prob"]
    zo = nn.sigmoid(om.forward(ids, train=False))
    assert float(np.abs(zg.cpu().numpy().reshape(-1) - zo).max()) < 1e-5


@pytest.mark.parametrize("exchange", ["examples", "unique", "unique-parts3", "unique-parts64"])
@pytest.mark.parametrize("kind,world,B,k", [("deepfm", 2, 48, 3), ("fm", 3, 40, 4), ("dcn", 2, 64, 2)])
def test_loopback_dp_inside_optimizer_windows_matches_the_oracle(kind, world, B, k, exchange, monkeypatch):
    """The same through optimizer windows (ONE ids collective + k global dedup results + one untouched-row sweep per window,
    the lazy window pass in every step's optimizer launch): windows of k steps, two windows.
    unique-partsP: the merge of the unique-row lists with P row-range parts per field (count pass + emit pass: what long lists
    -- dcn.py at 8 x 4 096, din.py's item table -- run; RSX_UX_PARTS forces it at these sizes)."""
    if exchange.startswith("unique-parts"):
        monkeypatch.setenv("RSX_UX_PARTS", exchange[len("unique-parts"):])
        exchange = "unique"
    import torch
    from oracle import models, nn
    from tests.dp_harness import loopback_train_step
    from tests.parity_util import synth_ids
    layers, D = ((32, 16), 16) if kind != "fm" else ((), 16)
    est, P, row_off = _build(kind, world, B, layers, D, 0.0, exchange, monkeypatch)
    assert est.store.window_k >= k and getattr(est.store, "window_dp", False)
    P64 = {n: v.astype(np.float64) for n, v in P.items()}
    om = _oracle_model(kind, P64, row_off, layers, 0.0)
    opt = nn.AdamTF1(dtype=np.float64)
    rng = np.random.default_rng(5)
    for w in range(2):
        ids = [[synth_ids(rng, B, row_off) for _ in range(k)] for _ in range(world)]          # [rank][position]
        ys = [[(rng.random(B) < 0.3).astype(np.float32) for _ in range(k)] for _ in range(world)]
        feats = [[{"ids": torch.from_numpy(i).cuda()} for i in ids[r]] for r in range(world)]
        for pos in range(k):
            lg = loopback_train_step(est, [feats[r][pos] for r in range(world)],
                                     [torch.from_numpy(ys[r][pos]).cuda() for r in range(world)], window=(k, pos, feats))
            lo, _ = models.train_step_dp(om, opt, [(ids[r][pos],) for r in range(world)],
                                         [ys[r][pos].astype(np.float64) for r in range(world)])
            assert all(abs(a - b) < 1e-5 for a, b in zip(lg, lo)), (w, pos, lg, lo)
    perr = _param_err(est, P64)
    assert max(perr.values()) < 2e-5, perr


def test_loopback_dp_of_a_batchnorm_free_model_equals_the_single_process_run_on_the_global_batch(monkeypatch):
    """fm.py has no batch-norm: DP(N, b) IS single(N b).  Pre-dedup exchange ('examples'): the optimizer stage sums the global
    batch's entries in the single process's order, so the embedding rows of the FIRST step are the same bits; the unique-list
    exchange adds per-rank partial sums (another association): 2e-6."""
    import torch
    from recsys_amd import fm
    from tests.dp_harness import loopback_train_step
    from tests.parity_util import load_oracle_weights, make_estimator, small_columns, synth_ids
    world, B, D = 3, 64, 16
    res = {}
    for exchange in ("examples", "unique"):
        est, P, row_off = _build("fm", world, B, (), D, 0.0, exchange, monkeypatch)
        lin, emb = small_columns(ROWS, D)
        single = make_estimator(fm.model_fn, {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": D,
                                              "learning_rate": 1e-3, "dropout": 0.0, "deep_layers": "", "max_batch_size": B * world})
        rng = np.random.default_rng(2)
        for step in range(3):
            ids = [synth_ids(rng, B, row_off) for _ in range(world)]
            ys = [(rng.random(B) < 0.3).astype(np.float32) for _ in range(world)]
            loopback_train_step(est, [{"ids": torch.from_numpy(i).cuda()} for i in ids], [torch.from_numpy(y).cuda() for y in ys])
            gi, gy = torch.from_numpy(np.concatenate(ids)).cuda(), torch.from_numpy(np.concatenate(ys)).cuda()
            if not single.store.built:
                with torch.no_grad():
                    single._call_model_fn({"ids": gi}, None, "infer")
                load_oracle_weights(single, P)
            single._train_step({"ids": gi}, gy)
            a, b = est.store.embeddings["input_layer"], single.store.embeddings["input_layer"]
            if step == 0 and exchange == "examples":
                assert torch.equal(a.tables, b.tables) and torch.equal(a.w1, b.w1)
        assert float((a.tables - b.tables).abs().max()) < 2e-6 and float((a.w1 - b.w1).abs().max()) < 2e-6
        assert float((est.store.dense.flat - single.store.dense.flat).abs().max()) < 2e-6


def test_the_loopback_comparison_fails_when_two_rank_blocks_are_swapped(monkeypatch):
    """Sensitivity of the harness itself: hand the optimizer stage the gathered gradient buffer with the blocks of ranks 0 and 1
    SWAPPED (their ids stay in place) -- the class of bug the identical-replica tests cannot see -- and the comparison with the
    oracle must fail by orders of magnitude more than the tolerance."""
    import torch
    from oracle import models, nn
    from tests.dp_harness import LoopbackDataParallel, loopback_train_step
    from tests.parity_util import synth_ids
    world, B, layers, D = 3, 40, (32, 16), 16
    est, P, row_off = _build("deepfm", world, B, layers, D, 0.0, "examples", monkeypatch)
    real = LoopbackDataParallel._gather_from_send

    def swapped(self, x):
        out = real(self, x)
        return torch.cat([out[1:2], out[0:1], out[2:]], 0)

    monkeypatch.setattr(LoopbackDataParallel, "_gather_from_send", swapped)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    om = _oracle_model("deepfm", P64, row_off, layers, 0.0)
    opt = nn.AdamTF1(dtype=np.float64)
    rng = np.random.default_rng(11)
    for step in range(2):
        ids = [synth_ids(rng, B, row_off) for _ in range(world)]
        ys = [(rng.random(B) < 0.3).astype(np.float32) for _ in range(world)]
        loopback_train_step(est, [{"ids": torch.from_numpy(i).cuda()} for i in ids], [torch.from_numpy(y).cuda() for y in ys])
        models.train_step_dp(om, opt, [(i,) for i in ids], [y.astype(np.float64) for y in ys])
    perr = _param_err(est, P64)
    assert perr["tables"] > 1e-4, perr


@pytest.mark.parametrize("exchange", ["examples", "unique"])
@pytest.mark.parametrize("world,B,cin,window", [(2, 24, (8, 4), 1), (3, 16, (20, 10, 10), 1), (2, 16, (8, 4), 3)])
def test_loopback_dp_xdeepfm_matches_the_oracle(world, B, cin, window, exchange, monkeypatch):
    """xdeepfm.py (two table sets behind one dedup, the first-order weights of the indicator columns, 3.3 MB-class dense arena
    through its own all-reduce when large): distinct batches per rank against oracle.models.train_step_dp."""
    import torch
    from oracle import criteo, init, models, nn
    from recsys_amd import xdeepfm
    from tests.dp_harness import LoopbackDataParallel, loopback_train_step
    from recsys_amd.feature_columns import build_feature_columns
    from tests.parity_util import make_estimator, synth_ids
    monkeypatch.setenv("RSX_DP_EXCHANGE", exchange)
    D, layers, seed = 16, (32, 16), 21
    rng = np.random.default_rng(seed)
    lin, emb = build_feature_columns(D, "numeric+indicator")
    row_off = criteo.row_offsets()
    cat_slot, cat_off = init.xdeepfm_layout()
    P = init.xdeepfm_params(seed, D, layers, cin, np.float32, row_off)
    for k in ("lin.b", "cin.bout", "dnn.bout"):
        P[k] += np.float32(0.05)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": D, "learning_rate": 1e-3,
              "dropout": 0.0, "deep_layers": ",".join(map(str, layers)), "cross_layers": ",".join(map(str, cin)),
              "max_batch_size": B}
    est = make_estimator(xdeepfm.model_fn, params)
    est.store.dp = LoopbackDataParallel(world)

    def feats(ids, logx):
        return {"ids": torch.from_numpy(ids).cuda(), "cont_log": torch.from_numpy(logx).cuda()}

    def batch():
        return (synth_ids(rng, B, row_off), np.log(np.floor(np.exp(rng.normal(2, 1, (B, 13)))) + 1.0).astype(np.float32),
                rng.integers(0, 2, B).astype(np.float32))

    b0 = batch()
    with torch.no_grad():
        est._call_model_fn(feats(*b0[:2]), None, "infer")
    st = est.store
    assert st.dp_unique == (exchange == "unique")
    w1 = np.zeros(int(row_off[-1]), np.float32)
    for j in range(26):
        s_ = int(cat_slot[j])
        w1[row_off[s_]:row_off[s_ + 1]] = P["lin.wcat"][cat_off[j]:cat_off[j + 1]]
    with torch.no_grad():
        st.embeddings["input_layer"].tables.copy_(torch.from_numpy(P["tables"]))
        st.embeddings["input_layer"].w1.copy_(torch.from_numpy(w1))
        st.embeddings["input_layer_1"].tables.copy_(torch.from_numpy(P["tables2"]))
    st.dense.load({k: v for k, v in P.items() if k in st.dense.params})
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    om = models.XDeepFM(P64, row_off, cat_slot, cat_off, cin, len(layers), 0.0)
    opt = nn.AdamTF1(dtype=np.float64)
    for w in range(2):
        bs = [[batch() for _ in range(window)] for _ in range(world)]            # [rank][position]
        fs = [[feats(*b[:2]) for b in bs[r]] for r in range(world)]
        for pos in range(window):
            lg = loopback_train_step(est, [fs[r][pos] for r in range(world)],
                                     [torch.from_numpy(bs[r][pos][2]).cuda() for r in range(world)],
                                     window=(window, pos, fs) if window > 1 else None)
            lo, _ = models.train_step_dp(om, opt, [(bs[r][pos][0], bs[r][pos][1].astype(np.float64)) for r in range(world)],
                                         [bs[r][pos][2].astype(np.float64) for r in range(world)])
            assert all(abs(a - b) < 2e-5 for a, b in zip(lg, lo)), (w, pos, lg, lo)
    a1 = st.embeddings["input_layer"]
    w1g = a1.w1.cpu().numpy()
    wcat = np.concatenate([w1g[row_off[int(cat_slot[j])]:row_off[int(cat_slot[j]) + 1]] for j in range(26)])
    perr = {"tables": float(np.abs(a1.tables.cpu().numpy() - P64["tables"]).max()),
            "tables2": float(np.abs(st.embeddings["input_layer_1"].tables.cpu().numpy() - P64["tables2"]).max()),
            "lin.wcat": float(np.abs(wcat - P64["lin.wcat"]).max())}
    for k, p in st.dense.params.items():
        perr[k] = float(np.abs(p.detach().cpu().numpy() - P64[k].reshape(p.shape)).max())
    assert max(perr.values()) < 5e-5, perr


@pytest.mark.parametrize("exchange", ["examples", "unique", "unique-parts5"])
@pytest.mark.parametrize("world,B,Pn", [(2, 24, 12), (3, 16, 20), (2, 96, 100)])
def test_loopback_dp_din_matches_the_oracle(world, B, Pn, exchange, monkeypatch):
    """din.py's fused step (din/din.py:204-206 MirroredStrategy): per-rank batches with ragged histories, target ids 0, the item
    bias riding as the item field's first-order vector, history padding mapped to the dummy rows.  (96, 100): 9 696 entries per
    rank -- the multi-launch sort and the two-stage scatter (global or, under the unique-list exchange, the rank's own)."""
    import torch
    from oracle import init, models, nn
    from recsys_amd import din, synthetic
    from tests.dp_harness import LoopbackDataParallel, loopback_train_step
    from tests.parity_util import make_estimator
    if exchange.startswith("unique-parts"):
        monkeypatch.setenv("RSX_UX_PARTS", exchange[len("unique-parts"):])
        exchange = "unique"
    monkeypatch.setenv("RSX_DP_EXCHANGE", exchange)
    K, n_item, n_cate = 16, 300, 20
    rng = np.random.default_rng(5)
    P = init.din_params(2, K, n_item, n_cate, np.float32)
    P["item_bias"] += (rng.standard_normal(n_item) * 0.01).astype(np.float32)
    est = make_estimator(din.model_fn, {"embedding_size": K, "learning_rate": 1e-3, "dropout": 0.0, "n_item": n_item,
                                        "n_cate": n_cate, "max_batch_size": B})
    est.store.dp = LoopbackDataParallel(world)
    keys = ("i_id", "i_cate", "u_iid_seq", "u_icat_seq")

    def feats(b):
        return {k: torch.from_numpy(b[k]).cuda() for k in keys}

    b0 = synthetic.din_batch(rng, B, Pn, n_item, n_cate)
    with torch.no_grad():
        est._call_model_fn(feats(b0), None, "infer")
    st = est.store
    assert st.din is not None and st.dp_unique == (exchange == "unique")
    with torch.no_grad():
        st.embeddings["i_id"].table.copy_(torch.from_numpy(P["item_emb"]))
        st.embeddings["i_cate"].table.copy_(torch.from_numpy(P["cate_emb"]))
        st.embeddings["i_item"].table[:, 0].copy_(torch.from_numpy(P["item_bias"]))
    st.dense.load({k: v for k, v in P.items() if k in st.dense.params})
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    om = models.DIN(P64, 0.0)
    opt = nn.AdamTF1(dtype=np.float64)
    for step in range(3):
        bs = [synthetic.din_batch(rng, B, Pn, n_item, n_cate) for _ in range(world)]
        bs[0]["i_id"][:2] = 0                                    # target id 0 trains like any row; the histories' 0 is padding
        lg = loopback_train_step(est, [feats(b) for b in bs], [torch.from_numpy(b["label"]).cuda() for b in bs])
        lo, _ = models.train_step_dp(om, opt, [tuple(b[k] for k in keys) for b in bs], [b["label"].astype(np.float64) for b in bs])
        assert all(abs(a - b_) < 1e-5 for a, b_ in zip(lg, lo)), (step, lg, lo)
    perr = {"item_emb": float(np.abs(st.embeddings["i_id"].table.cpu().numpy() - P64["item_emb"]).max()),
            "cate_emb": float(np.abs(st.embeddings["i_cate"].table.cpu().numpy() - P64["cate_emb"]).max()),
            "item_bias": float(np.abs(st.embeddings["i_item"].table[:, 0].cpu().numpy() - P64["item_bias"]).max())}
    for k, p in st.dense.params.items():
        perr[k] = float(np.abs(p.detach().cpu().numpy() - P64[k].reshape(p.shape)).max())
    assert max(perr.values()) < 2e-5, perr
