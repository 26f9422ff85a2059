"""Full-size parity cases: every BASELINE.json config at ITS batch size (batch size selects different kernels: the
radix field sort, the two-stage segment sum, split dW tiles, the large single-field sort of DIN).

Shared by
  * tests/golden/make_golden_fullsize.py  -- runs `make_inputs` + `oracle_run` (fp64 oracle) in the build container and
    commits the expected outputs as tests/golden/full_<name>.npz (small: losses, probabilities, the final dense
    variables, a sample of table rows -- the 54 MB inputs are regenerated from the seed on both sides and pinned by a
    digest);
  * tests/test_gpu_fullsize.py            -- regenerates the inputs on the GPU box, runs `hip_run` through the Estimator
    surface / C ABI and compares with the committed numbers (1e-5 on probabilities and losses).
"""
import hashlib

import numpy as np

from oracle import criteo, init, models, nn

STEPS = 2
CONFIGS = {
    # name: (model, batch, seed)            BASELINE.json configs[1..4]; dropout 0 (no masks to ship); lr 1e-3
    "deepfm_bs256": ("deepfm", 256, 101),
    "xdeepfm_bs256_cin128": ("xdeepfm", 256, 102),
    "dcn_bs4096": ("dcn", 4096, 103),
    "din_bs1024_p100_k32": ("din", 1024, 104),
}


def _synth_ids(rng, B, row_off):
    from tests.parity_util import synth_ids
    return synth_ids(rng, B, row_off)


def make_inputs(name):
    """-> (P fp32 dict, batches list, digest).  Deterministic in the seed (numpy Generator on the same image)."""
    kind, B, seed = CONFIGS[name]
    rng = np.random.default_rng(seed)
    row_off = criteo.row_offsets()
    if kind == "deepfm":
        P = init.deepfm_params(seed, 16, (100, 100), np.float32, row_off)
        P["b1"] += np.float32(0.05)
        batches = [dict(ids=_synth_ids(rng, B, row_off), label=rng.integers(0, 2, B).astype(np.float32)) for _ in range(STEPS)]
    elif kind == "dcn":
        P = init.dcn_params(seed, 16, (100, 100), 3, np.float32, row_off)
        batches = [dict(ids=_synth_ids(rng, B, row_off), label=rng.integers(0, 2, B).astype(np.float32)) for _ in range(STEPS)]
    elif kind == "xdeepfm":
        P = init.xdeepfm_params(seed, 16, (100, 100), (128, 128), np.float32, row_off)
        for k in ("lin.b", "cin.bout", "dnn.bout"):
            P[k] += np.float32(0.05)
        batches = []
        for _ in range(STEPS):
            ids = _synth_ids(rng, B, row_off)
            logx = np.log(np.floor(np.exp(rng.normal(2, 1, (B, 13)))) + 1.0).astype(np.float32)
            batches.append(dict(ids=ids, cont_log=logx, label=rng.integers(0, 2, B).astype(np.float32)))
    else:
        from recsys_amd import synthetic          # numpy-only generator of DIN-shaped batches (ragged histories)
        P = init.din_params(seed, 32, 63002, 802, np.float32)
        P["item_bias"] += (rng.standard_normal(63002) * 0.01).astype(np.float32)
        batches = [synthetic.din_batch(rng, B, 100, 63002, 802) for _ in range(STEPS)]
    h = hashlib.sha256()
    for k in sorted(P):
        h.update(k.encode())
        h.update(np.ascontiguousarray(P[k]).tobytes())
    for b in batches:
        for k in sorted(b):
            h.update(k.encode())
            h.update(np.ascontiguousarray(b[k]).tobytes())
    return P, batches, h.hexdigest()


def _sample_rows(touched, R, rng):
    touched = np.unique(np.asarray(touched).reshape(-1))
    pick_t = touched[:: max(1, len(touched) // 1500)]
    untouched = np.setdiff1d(rng.integers(0, R, 400), touched)[:200]
    return np.concatenate([pick_t, untouched]).astype(np.int64)


def oracle_run(name, P32, batches):
    """fp64 oracle: eval-mode probabilities before each step, train loss of each step, final variables."""
    kind, B, seed = CONFIGS[name]
    P = {k: v.astype(np.float64) for k, v in P32.items()}
    row_off = criteo.row_offsets()
    opt = nn.AdamTF1(dtype=np.float64)
    if kind == "deepfm":
        m = models.DeepFM(P, row_off, 2, 0.0)
        args = lambda b: (b["ids"],)
    elif kind == "dcn":
        m = models.DCN(P, row_off, 2, 0.0)
        args = lambda b: (b["ids"],)
    elif kind == "xdeepfm":
        cat_slot, cat_off = init.xdeepfm_layout()
        m = models.XDeepFM(P, row_off, cat_slot, cat_off, (128, 128), 2, 0.0)
        args = lambda b: (b["ids"], b["cont_log"].astype(np.float64))
    else:
        m = models.DIN(P, 0.0)
        args = lambda b: (b["i_id"], b["i_cate"], b["u_iid_seq"], b["u_icat_seq"])
    probs, losses = [], []
    for b in batches:
        probs.append(nn.sigmoid(m.forward(*args(b), train=False)))
        loss, _ = models.train_step(m, opt, args(b), b["label"].astype(np.float64))
        losses.append(float(loss))
    out = {"probs": np.stack(probs), "losses": np.array(losses)}
    rng = np.random.default_rng(seed + 1)
    sparse = {"deepfm": ("tables", "w1"), "dcn": ("tables",), "xdeepfm": ("tables", "tables2", "lin.wcat"),
              "din": ("item_emb", "cate_emb", "item_bias")}[kind]
    for k in P:
        if k in sparse:
            if kind == "din":
                touched = np.concatenate([np.concatenate([b["i_id"], b["u_iid_seq"].reshape(-1)]) if k != "cate_emb" else
                                          np.concatenate([b["i_cate"], b["u_icat_seq"].reshape(-1)]) for b in batches])
            elif k == "lin.wcat":
                cat_slot, cat_off = init.xdeepfm_layout()
                touched = np.concatenate([(b["ids"][:, cat_slot].astype(np.int64) + cat_off[None, :-1]).reshape(-1) for b in batches])
            else:
                touched = np.concatenate([(b["ids"].astype(np.int64) + row_off[None, :-1]).reshape(-1) for b in batches])
            rows = _sample_rows(touched, P[k].shape[0], rng)
            out["rows." + k] = rows
            out["vals." + k] = P[k][rows].astype(np.float32)
        else:
            out["final." + k] = P[k].astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------ HIP side (GPU box) ---
def hip_setup(name, P, first_batch, use_graph=False, extra_params=None, kind_B=None):
    """The Estimator of config `name` with the oracle's initial variables P loaded -> (est, feats: batch dict -> features)."""
    import torch
    from recsys_amd import dcn, deepfm, din, xdeepfm
    from recsys_amd.estimator import ModeKeys
    from recsys_amd.feature_columns import build_feature_columns
    from tests.parity_util import load_oracle_weights, make_estimator
    kind, B, seed = kind_B or CONFIGS[name]
    row_off = criteo.row_offsets()
    base = {"embedding_size": 16, "learning_rate": 1e-3, "dropout": 0.0, "deep_layers": "100,100", "max_batch_size": B}
    if kind in ("deepfm", "dcn", "xdeepfm"):
        linear = {"deepfm": "indicator_all", "dcn": "numeric", "xdeepfm": "numeric+indicator"}[kind]
        lin, emb = build_feature_columns(16, linear)
        base.update({"linear_feature_columns": lin, "embedding_feature_columns": emb,
                     "cross_layers": {"dcn": 3, "xdeepfm": "128,128"}.get(kind)})
        mfn = {"deepfm": deepfm.model_fn, "dcn": dcn.model_fn, "xdeepfm": xdeepfm.model_fn}[kind]
        feat_keys = ("ids", "cont_log") if kind == "xdeepfm" else ("ids",)
    else:
        base.update({"embedding_size": 32, "n_item": 63002, "n_cate": 802})
        mfn = din.model_fn
        feat_keys = ("i_id", "i_cate", "u_iid_seq", "u_icat_seq")
    base.update(extra_params or {})
    est = make_estimator(mfn, base, use_graph=use_graph)

    def feats(b):
        return {k: torch.from_numpy(np.ascontiguousarray(b[k])).cuda() for k in feat_keys}

    est._call_model_fn(feats(first_batch), None, ModeKeys.PREDICT)
    st = est.store
    if kind == "din":
        with torch.no_grad():
            st.embeddings["i_id"].table.copy_(torch.from_numpy(P["item_emb"]))
            st.embeddings["i_cate"].table.copy_(torch.from_numpy(P["cate_emb"]))
            st.embeddings["i_item"].table[:, 0].copy_(torch.from_numpy(P["item_bias"]))
        st.dense.load({k: v for k, v in P.items() if k in st.dense.params})
    elif kind == "xdeepfm":
        cat_slot, cat_off = init.xdeepfm_layout()
        w1 = np.zeros(int(row_off[-1]), np.float32)
        for j in range(26):
            s = int(cat_slot[j])
            w1[row_off[s]:row_off[s + 1]] = P["lin.wcat"][cat_off[j]:cat_off[j + 1]]
        with torch.no_grad():
            st.embeddings["input_layer"].tables.copy_(torch.from_numpy(P["tables"]))
            st.embeddings["input_layer"].w1.copy_(torch.from_numpy(w1))
            st.embeddings["input_layer_1"].tables.copy_(torch.from_numpy(P["tables2"]))
        st.dense.load({k: v for k, v in P.items() if k in st.dense.params})
    else:
        load_oracle_weights(est, P)
    return est, feats


def hip_run(name, P, batches, use_graph=False, extra_params=None):
    import torch
    from recsys_amd.estimator import ModeKeys
    kind, B, seed = CONFIGS[name]
    row_off = criteo.row_offsets()
    est, feats = hip_setup(name, P, batches[0], use_graph, extra_params)
    st = est.store
    probs, losses = [], []
    for b in batches:
        f = feats(b)
        with torch.no_grad():
            probs.append(est._call_model_fn(f, None, ModeKeys.PREDICT).predictions["prob"].cpu().numpy().reshape(-1))
        lab = torch.from_numpy(b["label"]).cuda()
        losses.append(float(est._train_step(f, lab)))
    got = {"probs": np.stack(probs), "losses": np.array(losses)}
    tabs = {}
    if kind == "din":
        tabs = {"item_emb": st.embeddings["i_id"].table, "cate_emb": st.embeddings["i_cate"].table,
                "item_bias": st.embeddings["i_item"].table[:, 0]}
    else:
        a = st.embeddings["input_layer"]
        tabs["tables"] = a.tables
        if kind == "deepfm":
            tabs["w1"] = a.w1
        if kind == "xdeepfm":
            tabs["tables2"] = st.embeddings["input_layer_1"].tables
            cat_slot, cat_off = init.xdeepfm_layout()
            w1g = a.w1
            tabs["lin.wcat"] = torch.cat([w1g[int(row_off[int(cat_slot[j])]):int(row_off[int(cat_slot[j]) + 1])] for j in range(26)])
    got["tables"] = {k: v.detach().cpu().numpy() for k, v in tabs.items()}
    got["dense"] = {k: p.detach().cpu().numpy() for k, p in st.dense.params.items()}
    return got
