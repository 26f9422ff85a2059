"""Streaming eval metrics on device: tf.metrics.auc (200 thresholds,
trapezoidal ROC) and tf.metrics.accuracy(labels, tf.round(pred)) as used at fm/fm.py:150-153."""
import torch


class StreamingAUC:
    def __init__(self, device, num_thresholds=200):
        n, eps = num_thresholds, 1e-7
        th = [0.0 - eps] + [(i + 1) / (n - 1) for i in range(n - 2)] + [1.0 + eps]
        self.th = torch.tensor(th, dtype=torch.float32, device=device)
        z = lambda: torch.zeros(n, dtype=torch.float64, device=device)
        self.tp, self.fp, self.tn, self.fn = z(), z(), z(), z()

    def update(self, labels, pred):
        y = labels.reshape(-1) > 0.5
        p = pred.reshape(-1).to(torch.float32)
        gt = p[None, :] > self.th[:, None]
        yp, yn = y[None, :], ~y[None, :]
        self.tp += (gt & yp).sum(1)
        self.fp += (gt & yn).sum(1)
        self.fn += (~gt & yp).sum(1)
        self.tn += (~gt & yn).sum(1)

    def result(self):
        tp, fp, tn, fn = (a.to(torch.float32) for a in (self.tp, self.fp, self.tn, self.fn))
        e = 1e-6
        tpr = (tp + e) / (tp + fn + e)
        fpr = fp / (fp + tn + e)
        return float(((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2.0).sum())


class StreamingAccuracy:
    def __init__(self, device):
        self.total = torch.zeros((), dtype=torch.float64, device=device)
        self.count = 0

    def update(self, labels, pred):
        y = labels.reshape(-1).to(torch.float32)
        r = torch.round(pred.reshape(-1).to(torch.float32))     # round-half-to-even like tf.round
        self.total += (r == y).sum()
        self.count += y.numel()

    def result(self):
        return float(self.total) / max(self.count, 1)
