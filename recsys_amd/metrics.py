"""Streaming eval metrics on device: tf.metrics.auc (200 thresholds, trapezoidal
ROC), tf.metrics.accuracy(labels, tf.round(pred)) and the Estimator's mean of the batch losses, as the
eval_metric_ops of fm/fm.py:150-153 and Estimator.evaluate(steps=200) fm/fm.py:221 produce them.

One HIP launch per eval batch (`rsx_eval_metrics_update`, csrc/metrics.hip) accumulates everything in a 406-word
device buffer; the host reads it back ONCE in `result()` -- evaluate() never synchronises per batch.  Data-parallel
evaluation sums the integer counters of all ranks before the finalisation (`all_reduce`)."""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib


def auc_thresholds(num_thresholds=200):
    """TF metrics_impl.py: kepsilon = 1e-7; thresholds = [0 - eps] + [(i+1)/(n-1) for i in range(n-2)] + [1 + eps],
    Python doubles handed to a float32 constant."""
    n, eps = num_thresholds, 1e-7
    return np.array([0.0 - eps] + [(i + 1) * 1.0 / (n - 1) for i in range(n - 2)] + [1.0 + eps], np.float32)


class EvalMetrics:
    """AUC + Accuracy + mean loss of one evaluate() call."""

    def __init__(self, device, num_thresholds=200):
        self.T = num_thresholds
        self.device = torch.device(device)
        self.th = torch.from_numpy(auc_thresholds(num_thresholds)).to(self.device)
        words = lib().rsx_eval_metrics_state_words(self.T)
        self.state = torch.zeros(words, dtype=torch.int64, device=self.device)

    def update(self, labels, prob, batch_loss=None):
        """labels / prob: device tensors of B elements (any shape); batch_loss: device scalar or None."""
        y = labels.reshape(-1).to(torch.float32).contiguous()
        p = prob.reshape(-1).to(torch.float32).contiguous()
        if y.numel() != p.numel():
            raise ValueError("labels and predictions differ in size")
        bl = None if batch_loss is None else batch_loss.detach().reshape(-1)[:1].to(torch.float32).contiguous()
        check(lib().rsx_eval_metrics_update(C.c_void_p(p.data_ptr()), C.c_void_p(y.data_ptr()),
                                            C.c_void_p(self.th.data_ptr()), self.T,
                                            C.c_void_p(bl.data_ptr() if bl is not None else None),
                                            C.c_void_p(self.state.data_ptr()), int(p.numel()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rsx_eval_metrics_update")

    def all_reduce(self, dp):
        """Sum the counters of every rank (data-parallel evaluation over disjoint shards)."""
        T = self.T
        loss = self.state[2 * T + 5:2 * T + 6].view(torch.float64).clone()
        ints = self.state.clone()
        ints[2 * T + 5] = 0
        dp.all_reduce_sum(ints)
        dp.all_reduce_sum(loss)
        self.state.copy_(ints)
        self.state[2 * T + 5:2 * T + 6] = loss.view(torch.int64)

    def result(self):
        """-> {'AUC', 'Accuracy', 'loss', 'examples'}; ONE device->host copy."""
        T = self.T
        s = self.state.cpu().numpy()
        return finalize(s[:T + 1], s[T + 1:2 * T + 2], int(s[2 * T + 2]), int(s[2 * T + 3]), int(s[2 * T + 4]),
                        float(s[2 * T + 5:2 * T + 6].view(np.float64)[0]))


def finalize(hist_pos, hist_neg, correct, examples, batches, loss_sum):
    """Counters -> metric values with TF's fp32 formulas: tpr = (tp + 1e-6) / (tp + fn + 1e-6),
    fpr = fp / (fp + tn + 1e-6), AUC = sum((fpr[i] - fpr[i+1]) * (tpr[i] + tpr[i+1]) / 2)."""
    T = len(hist_pos) - 1
    # tp[i] = #(label & pred > t_i) = examples that exceed MORE than i thresholds
    tp = (hist_pos[::-1].cumsum()[::-1])[1:].astype(np.float32)
    fp = (hist_neg[::-1].cumsum()[::-1])[1:].astype(np.float32)
    fn = np.float32(hist_pos.sum()) - tp
    tn = np.float32(hist_neg.sum()) - fp
    assert tp.shape[0] == T
    e = np.float32(1e-6)
    tpr = (tp + e) / (tp + fn + e)
    fpr = fp / (fp + tn + e)
    auc = float(np.sum((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / np.float32(2.0), dtype=np.float32))
    return {"AUC": auc, "Accuracy": correct / max(examples, 1), "loss": loss_sum / max(batches, 1), "examples": examples}
