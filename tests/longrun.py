"""Long-horizon parity: the reference's only published outputs are END-OF-TRAINING AUC / logloss
(fm/fm.py:150-153,221; README.md:22-28), so a few hundred TRAIN steps followed by an EVAL over held-out batches is what "parity"
has to mean beyond the 2-5 step trajectories of the other tests.  200 steps at the BASELINE batch size (256), then
`Estimator.evaluate` semantics over 20 held-out batches: mean batch loss, AUC-200, accuracy (moving batch-norm statistics never
updated: Appendix A-8).

Labels are LEARNABLE -- Bernoulli(sigmoid(planted per-row weights)) -- so that the model moves away from AUC 0.5 and the eval
numbers depend on what the 200 steps did.

Shared by tests/golden/make_golden_long.py (fp64 oracle, offline in the build container -> tests/golden/long_<name>.npz) and
tests/test_gpu_long.py (HIP path on the GPU box, through the optimizer-window schedule the product runs)."""
import hashlib

import numpy as np

from oracle import criteo, init, models, nn
from tests.parity_util import synth_ids

TRAIN_STEPS, EVAL_BATCHES, B = 200, 20, 256
CONFIGS = {"deepfm_bs256": ("deepfm", 201), "xdeepfm_bs256_cin128": ("xdeepfm", 202)}


def make_inputs(name):
    """-> (P fp32 dict, train batches, eval batches, digest); deterministic in the seed."""
    kind, seed = CONFIGS[name]
    rng = np.random.default_rng(seed)
    row_off = criteo.row_offsets()
    F = len(row_off) - 1
    if kind == "deepfm":
        P = init.deepfm_params(seed, 16, (100, 100), np.float32, row_off)
        P["b1"] += np.float32(0.05)
    else:
        P = init.xdeepfm_params(seed, 16, (100, 100), (128, 128), np.float32, row_off)
        for k in ("lin.b", "cin.bout", "dnn.bout"):
            P[k] += np.float32(0.05)
    planted = rng.standard_normal(int(row_off[-1])) * 1.2           # the signal the labels carry

    def batch():
        ids = synth_ids(rng, B, row_off)
        z = planted[ids.astype(np.int64) + row_off[None, :-1]].sum(1) / np.sqrt(F) * 2.0 - 0.8
        b = dict(ids=ids, label=(rng.random(B) < 1.0 / (1.0 + np.exp(-z))).astype(np.float32))
        if kind == "xdeepfm":
            b["cont_log"] = np.log(np.floor(np.exp(rng.normal(2, 1, (B, 13)))) + 1.0).astype(np.float32)
        return b

    train = [batch() for _ in range(TRAIN_STEPS)]
    ev = [batch() for _ in range(EVAL_BATCHES)]
    h = hashlib.sha256()
    for k in sorted(P):
        h.update(k.encode())
        h.update(np.ascontiguousarray(P[k]).tobytes())
    for b in train + ev:
        for k in sorted(b):
            h.update(np.ascontiguousarray(b[k]).tobytes())
    return P, train, ev, h.hexdigest()


def _oracle_model(kind, P):
    row_off = criteo.row_offsets()
    if kind == "deepfm":
        return models.DeepFM(P, row_off, 2, 0.0), (lambda b, dt: (b["ids"],))
    cat_slot, cat_off = init.xdeepfm_layout()
    return models.XDeepFM(P, row_off, cat_slot, cat_off, (128, 128), 2, 0.0), (lambda b, dt: (b["ids"], b["cont_log"].astype(dt)))


def accuracy_hits(labels, prob):
    """tf.metrics.accuracy(labels, round(pred)): round-half-to-even (Appendix A-11)."""
    return int((np.rint(prob) == labels).sum())


def oracle_run(name, P32, train, ev, dtype=np.float64, progress=None):
    """The oracle's 200 TRAIN steps (TF-1 non-lazy Adam) + the EVAL: every step's train loss, and after the last step the
    per-batch eval losses, eval probabilities, AUC-200, accuracy."""
    kind, _ = CONFIGS[name]
    P = {k: v.astype(dtype) for k, v in P32.items()}
    m, args = _oracle_model(kind, P)
    opt = nn.AdamTF1(dtype=dtype)
    losses = []
    for i, b in enumerate(train):
        loss, _ = models.train_step(m, opt, args(b, dtype), b["label"].astype(dtype))
        losses.append(float(loss))
        if progress and (i + 1) % 20 == 0:
            progress(i + 1, losses[-1])
    auc = nn.StreamingAUC()
    ev_losses, probs, hits = [], [], 0
    for b in ev:
        z = m.forward(*args(b, dtype), train=False)
        l, _ = nn.sigmoid_ce_mean(z, b["label"].astype(dtype))
        p = nn.sigmoid(z)
        auc.update(b["label"], p.astype(np.float32))
        hits += accuracy_hits(b["label"], p)
        ev_losses.append(float(l))
        probs.append(p)
    return {"train_losses": np.array(losses), "eval_losses": np.array(ev_losses), "eval_probs": np.stack(probs).astype(np.float64),
            "eval_loss": float(np.mean(ev_losses)), "auc": float(auc.result()), "accuracy": hits / float(EVAL_BATCHES * B),
            "final_dense": {k: P[k].astype(np.float32) for k in P if P[k].ndim <= 2 and P[k].shape[0] < 10000 and P[k].size <= 20000}}


def hip_run(name, P, train, ev, extra_params=None, use_graph=False):
    """The same run through the product: optimizer windows (Estimator._train_window, the schedule train() runs) and
    Estimator.evaluate (device metrics) -> the same dictionary."""
    import torch
    from recsys_amd.estimator import ModeKeys
    from tests import fullsize
    kind, seed = CONFIGS[name]
    est, feats = fullsize.hip_setup(name, P, train[0], use_graph, extra_params, kind_B=(kind, B, seed))
    K = est._window_len()
    losses = []
    dev = [(feats(b), torch.from_numpy(b["label"]).cuda()) for b in train]
    for s in range(0, TRAIN_STEPS, K):
        win = dev[s:s + K]
        fw = [f for f, _ in win]
        for pos, (f, l) in enumerate(win):          # Estimator._train_window's loop, with every step's loss read when it is made
            est.store.window = (len(win), pos, fw) if len(win) > 1 else None      # (the loss tensor is a buffer the next step reuses)
            try:
                losses.append(float(est._train_eager(f, l)))
            finally:
                est.store.window = None
    evd = [(feats(b), torch.from_numpy(b["label"]).cuda()) for b in ev]
    res = est.evaluate(lambda: iter(evd), steps=EVAL_BATCHES)
    probs, ev_losses = [], []
    with torch.no_grad():
        for f, l in evd:
            sp = est._call_model_fn(f, l, ModeKeys.EVAL)
            probs.append(sp.predictions["prob"].cpu().numpy().reshape(-1).astype(np.float64))
            ev_losses.append(float(sp.loss))
    return {"train_losses": np.array(losses), "eval_losses": np.array(ev_losses), "eval_probs": np.stack(probs),
            "eval_loss": float(res["loss"]), "auc": float(res["AUC"]), "accuracy": float(res["Accuracy"]), "window": K,
            "final_dense": {k: p.detach().cpu().numpy() for k, p in est.store.dense.params.items()}}
