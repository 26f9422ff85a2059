"""Shared helpers of the GPU parity tests and __graft_entry__.smoke(): run the same seeded inputs
through the CPU oracle and through the HIP path (via the Estimator surface / C ABI) and compare."""
import numpy as np

from oracle import criteo, init, models, nn


def synth_ids(rng, B, row_off, zipf_a=1.05):
    """Criteo-shaped ids: Zipf over each field's bucket count, table-local, [B,F] int32."""
    F = len(row_off) - 1
    ids = np.zeros((B, F), np.int32)
    for f in range(F):
        n = int(row_off[f + 1] - row_off[f])
        r = rng.zipf(zipf_a, B).astype(np.int64)
        ids[:, f] = ((r * 2654435761) % n).astype(np.int32) if n > 16 else rng.integers(0, n, B)
    return ids


def small_columns(rows, D):
    from recsys_amd.feature_columns import Column
    emb = [Column("f%02d_embedding" % i, "f%02d" % i, "hash_embedding", r, D) for i, r in enumerate(rows)]
    lin = [Column("f%02d_indicator" % i, "f%02d" % i, "hash_indicator", r) for i, r in enumerate(rows)]
    return lin, emb


def make_estimator(model_fn, params, adam_mode="tf1_dense", use_graph=False):
    from recsys_amd.estimator import Estimator, RunConfig
    return Estimator(model_fn, None, params, RunConfig(use_hip_graph=use_graph, adam_mode=adam_mode))


def load_oracle_weights(est, P):
    """Copy the oracle's numpy parameters into the Estimator's HBM arenas."""
    import torch
    st = est.store
    with torch.no_grad():
        for name, arena in st.embeddings.items():
            key = "tables" if name == "input_layer" else "tables2"
            arena.tables.copy_(torch.from_numpy(P[key]))
            if arena.with_w1 and "w1" in P:
                arena.w1.copy_(torch.from_numpy(P["w1"]))
    st.dense.load({k: v for k, v in P.items() if k in st.dense.params})


def deepfm_parity_run(B=64, steps=2, seed=0, rows=None, D=16, layers=(100, 100), adam_mode="tf1_dense",
                      use_graph=False, return_all=False, tower="hip", dropout=0.0, kind="deepfm", cross_layers=3,
                      data_parallel=False, oracle_dtype=None):
    """Train `steps` steps of `kind` in {deepfm, fm, dcn} on both sides from identical weights and batches (injected
    dropout masks).  Returns max |prob_gpu - prob_oracle| over all steps (and losses / final parameter errors)."""
    import torch
    from recsys_amd import dcn, deepfm, fm
    from recsys_amd.estimator import ModeKeys
    from recsys_amd.feature_columns import build_feature_columns
    rng = np.random.default_rng(seed)
    if rows is None:
        lin, emb = build_feature_columns(D)
        row_off = criteo.row_offsets()
    else:
        lin, emb = small_columns(rows, D)
        row_off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    if kind == "dcn":
        P = init.dcn_params(seed, D, layers, cross_layers, np.float32, row_off)
        mfn = dcn.model_fn
    elif kind == "fm":
        P = init.deepfm_params(seed, D, (), np.float32, row_off, with_dnn=False)
        mfn, layers = fm.model_fn, ()
    else:
        P = init.deepfm_params(seed, D, layers, np.float32, row_off)
        mfn = deepfm.model_fn
    if "b1" in P:
        P["b1"] += np.float32(0.05)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": D,
              "learning_rate": 1e-3, "dropout": dropout, "deep_layers": ",".join(map(str, layers)), "max_batch_size": B,
              "tower": tower, "cross_layers": cross_layers}
    est = make_estimator(mfn, params, adam_mode, use_graph)
    if data_parallel:        # world_size-1 RCCL: exercises the collective code path; results must not change
        from recsys_amd import dist as rdist
        rdist.init_process_group("nccl")
        est.store.dp = est.dist = rdist.DataParallel()
    batches = [(synth_ids(rng, B, row_off), rng.integers(0, 2, B).astype(np.float32)) for _ in range(steps)]
    ids0 = torch.from_numpy(batches[0][0]).cuda()
    est._call_model_fn({"ids": ids0}, None, ModeKeys.PREDICT)        # creates the variables
    load_oracle_weights(est, P)
    import os
    odt = np.float64 if (oracle_dtype or os.environ.get("RSX_TEST_ORACLE_DTYPE", "f32")) == "f64" else np.float32
    if odt is np.float64:      # fp64 oracle from the same fp32 initial values (what the large-batch cases compare with: an
        P = {k: v.astype(np.float64) for k, v in P.items()}      # fp32 numpy BN / matmul over 4096 rows has its own 1e-5 noise)
    om = {"dcn": lambda: models.DCN(P, row_off, len(layers), dropout), "fm": lambda: models.FM(P, row_off),
          "deepfm": lambda: models.DeepFM(P, row_off, len(layers), dropout)}[kind]()
    opt = nn.AdamTF1(dtype=odt)
    err = 0.0
    losses = []
    for ids, y in batches:
        f = {"ids": torch.from_numpy(ids).cuda()}
        lab = torch.from_numpy(y).cuda()
        with torch.no_grad():
            zg = est._call_model_fn(f, None, ModeKeys.PREDICT).predictions["prob"]
        mk = None
        if dropout > 0.0 and len(layers):      # injected keep-masks, identical on both sides
            mk = [(rng.random((B, n)) >= dropout).astype(np.float32) for n in layers]
            est.params["_dropout_masks"] = [torch.from_numpy(m).cuda() for m in mk]
        loss_g = est._train_step(f, lab)
        zo_eval = nn.sigmoid(om.forward(ids, train=False))
        loss_o, _ = models.train_step(om, opt, (ids,), y.astype(odt), {"masks": [m.astype(odt) for m in mk]} if mk else None,
                                      lazy=(adam_mode == "lazy_rows"))
        err = max(err, float(np.abs(zg.cpu().numpy().reshape(-1) - zo_eval).max()))
        losses.append((float(loss_g), float(loss_o)))
    if not return_all:
        return err
    st = est.store
    a = st.embeddings["input_layer"]
    perr = {"tables": float(np.abs(a.tables.cpu().numpy() - P["tables"]).max())}
    if a.with_w1:
        perr["w1"] = float(np.abs(a.w1.cpu().numpy() - P["w1"]).max())
    for k, p in st.dense.params.items():
        perr[k] = float(np.abs(p.detach().cpu().numpy() - P[k].reshape(p.shape)).max())
    return err, losses, perr
