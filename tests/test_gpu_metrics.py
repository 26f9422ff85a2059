"""Device eval metrics (rsx_eval_metrics_update, csrc/metrics.hip) against the oracle's tf.metrics.auc /
tf.metrics.accuracy restatement (oracle.nn.StreamingAUC / StreamingAccuracy; fm/fm.py:150-153)."""
import numpy as np
import pytest
import torch

from oracle import nn
from recsys_amd import metrics


def _hist_cpu(th, p, y):
    k = np.searchsorted(th, p, side="left")          # #{i : th[i] < p}
    k[np.isnan(p)] = 0
    T = len(th)
    return (np.bincount(k[y > 0.5], minlength=T + 1), np.bincount(k[y <= 0.5], minlength=T + 1))


def _adversarial_predictions(rng, n):
    th = metrics.auc_thresholds()
    p = rng.random(n).astype(np.float32)
    # values exactly ON thresholds, one ulp either side, exactly 0.5 (round-half-even -> 0), 0, 1
    special = np.concatenate([th[1:-1], np.nextafter(th[1:-1], np.float32(2)), np.nextafter(th[1:-1], np.float32(-1)),
                              np.array([0.5, 0.0, 1.0, np.nextafter(np.float32(0.5), np.float32(1)), 1.5, 2.5], np.float32)])
    special = np.clip(special, 0, 1).astype(np.float32)
    k = min(len(special) * 4, n // 2)                 # small batches: a random half of the batch gets special values
    idx = rng.choice(n, k, replace=False)
    p[idx] = rng.permutation(np.tile(special, 4))[:k] if k < len(special) * 4 else np.tile(special, 4)
    y = (rng.random(n) < 0.3 + 0.4 * p).astype(np.float32)       # informative labels: AUC well away from 0.5
    return p, y


def test_finalize_matches_oracle_cpu():
    """Host half only (no GPU): histogram -> AUC / accuracy equals the oracle's streaming counters."""
    rng = np.random.default_rng(0)
    p, y = _adversarial_predictions(rng, 60000)
    hp, hn = _hist_cpu(metrics.auc_thresholds(), p, y)
    res = metrics.finalize(hp, hn, int((np.round(p) == y).sum()), len(p), 1, 0.25)
    auc, acc = nn.StreamingAUC(), nn.StreamingAccuracy()
    auc.update(y, p)
    acc.update(y, p)
    assert res["AUC"] == auc.result()
    assert res["Accuracy"] == acc.result()
    assert 0.6 < res["AUC"] < 0.9


@pytest.mark.gpu
def test_device_metrics_match_oracle_streaming():
    dev = torch.device("cuda")
    rng = np.random.default_rng(1)
    n_batches, B = 200, 256 + 37                      # evaluate(steps=200) of ragged-size batches
    met = metrics.EvalMetrics(dev)
    auc, acc = nn.StreamingAUC(), nn.StreamingAccuracy()
    losses = []
    for i in range(n_batches):
        p, y = _adversarial_predictions(rng, B if i % 7 else B + 1000)
        loss = np.float32(rng.random())
        losses.append(loss)
        auc.update(y, p)
        acc.update(y, p)
        met.update(torch.from_numpy(y).to(dev).reshape(-1, 1), torch.from_numpy(p).to(dev).reshape(-1, 1),
                   torch.tensor(loss, device=dev))
    res = met.result()
    assert res["examples"] == int(acc.count) and res["examples"] > 50000
    assert res["AUC"] == auc.result()                 # integer counters + the same fp32 formula: exact
    assert res["Accuracy"] == acc.result()
    assert abs(res["loss"] - float(np.mean(np.asarray(losses, np.float64)))) < 1e-12
    # the histogram itself, bin for bin
    T = met.T
    s = met.state.cpu().numpy()
    th = metrics.auc_thresholds()
    rng = np.random.default_rng(1)
    hp, hn = np.zeros(T + 1, np.int64), np.zeros(T + 1, np.int64)
    for i in range(n_batches):
        p, y = _adversarial_predictions(rng, B if i % 7 else B + 1000)
        rng.random()
        a, b = _hist_cpu(th, p, y)
        hp += a
        hn += b
    np.testing.assert_array_equal(s[:T + 1], hp)
    np.testing.assert_array_equal(s[T + 1:2 * T + 2], hn)


@pytest.mark.gpu
def test_device_metrics_nan_and_bounds():
    dev = torch.device("cuda")
    met = metrics.EvalMetrics(dev)
    p = np.array([np.nan, 0.0, 1.0, 0.5, 0.25], np.float32)
    y = np.array([1, 0, 1, 0, 1], np.float32)
    met.update(torch.from_numpy(y).to(dev), torch.from_numpy(p).to(dev))
    auc, acc = nn.StreamingAUC(), nn.StreamingAccuracy()
    auc.update(y, p)
    acc.update(y, p)
    res = met.result()
    assert res["AUC"] == auc.result() and res["Accuracy"] == acc.result()
    assert res["loss"] == 0.0
