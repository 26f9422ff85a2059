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
    from recsys_amd.dist import LoopbackDataParallel
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
    from recsys_amd.dist import loopback_train_step
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


@pytest.mark.parametrize("exchange", ["examples", "unique"])
@pytest.mark.parametrize("kind,world,B,k", [("deepfm", 2, 48, 3), ("fm", 3, 40, 4), ("dcn", 2, 64, 2)])
def test_loopback_dp_inside_optimizer_windows_matches_the_oracle(kind, world, B, k, exchange, monkeypatch):
    """The same through optimizer windows (ONE ids collective + k global dedup results + one untouched-row sweep per window,
    the lazy window pass in every step's optimizer launch): windows of k steps, two windows."""
    import torch
    from oracle import models, nn
    from recsys_amd.dist import loopback_train_step
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
    from recsys_amd.dist import loopback_train_step
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
    from recsys_amd.dist import LoopbackDataParallel, loopback_train_step
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
