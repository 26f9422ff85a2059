"""Worker of tests/test_gpu_knobs.py: trains one BASELINE-shaped model for a few steps through the product's default launch
schedule (HBM-resident batches, HIP graphs, optimizer windows) under whatever RSX_* knobs the environment carries and prints a
digest of every variable and optimizer slot.  usage: knob_worker.py <model> <batch> <steps> [cin_bf16]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    model, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    bf16 = len(sys.argv) > 4 and sys.argv[4] == "1"
    from recsys_amd import dcn, deepfm, din, fm, synthetic, xdeepfm
    from recsys_amd.estimator import Estimator, PackedBatch, RunConfig
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    linear = {"deepfm": "indicator_all", "fm": "indicator_all", "dcn": "numeric", "xdeepfm": "numeric+indicator"}.get(model)
    lin, emb = build_feature_columns(16, linear) if linear else (None, None)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 32 if model == "din" else 16,
              "learning_rate": 1e-3, "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": B,
              "cross_layers": {"dcn": 3, "xdeepfm": "32,16"}.get(model), "cin_bf16": bf16}
    mfn = {"deepfm": deepfm.model_fn, "fm": fm.model_fn, "dcn": dcn.model_fn, "xdeepfm": xdeepfm.model_fn, "din": din.model_fn}[model]
    est = Estimator(mfn, None, params, RunConfig(use_hip_graph=True, adam_mode="tf1_dense", device="cuda", seed=77))
    if model == "din":                       # din.py's fused TRAIN step (DinFused): histories of 100, K = 32
        rng = np.random.default_rng(5)
        raw = [synthetic.din_batch(rng, B) for _ in range(4)]
        feats = [PackedBatch({k: v.astype(np.int32) for k, v in b.items() if k != "label"}, b["label"], device="cuda") for b in raw]
    else:
        layout = CriteoLayout.from_columns(emb)
        host = synthetic.criteo_id_batches(layout, 4, B, seed=5)
        feats = [PackedBatch({"ids": i, "cont_log": c} if model == "xdeepfm" else {"ids": i}, y, device="cuda") for i, y, c in host]
    with torch.no_grad():
        est._call_model_fn(feats[0].views()[0], None, "infer")
    loss = est.train_resident(feats, steps, 8)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    st = est.store
    for name in sorted(st.embeddings):
        a = st.embeddings[name]
        if hasattr(a, "tables"):             # an arena (Criteo models)
            ts = (a.tables, a.m_t, a.v_t) + ((a.w1, a.m_w, a.v_w) if a.with_w1 else ())
        else:                                # a SparseTable view (din.py: views of the two-field arena and of the bias arena)
            ts = (a.table, a.m, a.v)
        for t in ts:
            h.update(t.cpu().numpy().tobytes())
    for t in (st.dense.flat, st.dense.m, st.dense.v):
        h.update(t.cpu().numpy().tobytes())
    dense = st.dense.flat.cpu().numpy()
    print(json.dumps({"digest": h.hexdigest(), "loss": float(loss), "dense_abs_sum": float(np.abs(dense).sum()),
                      "dense": dense[::97].tolist()}))


if __name__ == "__main__":
    main()
