"""Oracle: numpy layers (forward + hand-written backward), loss, TF-1 Adam, metrics.

Restates the TF-1.x ops the reference's model_fn bodies call (SURVEY.md
section 8c "call sites" row): tf.layers.dense / batch_normalization / dropout
(deepfm/deepfm.py:103-108), tf.nn.sigmoid_cross_entropy_with_logits
(fm/fm.py:146-149), tf.train.AdamOptimizer (fm/fm.py:162-163; Appendix A-5),
tf.metrics.auc / accuracy (fm/fm.py:150-153; Appendix A-11).

Every function takes/returns arrays of one dtype (float32 to mimic TF, float64
for gradient checks).  Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon


# ---------------------------------------------------------------- dense ------
def dense_fwd(x, W, b, relu=False):
    y = x @ W + b
    if relu:
        y = np.maximum(y, 0)
    return y


def dense_bwd(x, W, y, dy, relu=False):
    """Returns dx, dW, db.  `y` is the layer output (for the relu mask)."""
    if relu:
        dy = dy * (y > 0)
    return dy @ W.T, x.T @ dy, dy.sum(0)


# ------------------------------------------------------------ batch-norm -----
def bn_train_fwd(x, gamma, beta):
    """tf.layers.batch_normalization(training=True) on rank-2 input (non-fused path):
    tf.nn.moments + tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma; y = x*inv + (beta-mean*inv).
    Moving statistics are NOT updated (Appendix A-8: no UPDATE_OPS dependency)."""
    dt = x.dtype
    mean = x.mean(0)
    var = ((x - mean) ** 2).mean(0)
    rstd = (1.0 / np.sqrt(var + dt.type(BN_EPS))).astype(dt)
    inv = rstd * gamma
    y = x * inv + (beta - mean * inv)
    return y, (x, mean, rstd, gamma)


def bn_train_bwd(cache, dy):
    x, mean, rstd, gamma = cache
    B = x.shape[0]
    xhat = (x - mean) * rstd
    dgamma = (dy * xhat).sum(0)
    dbeta = dy.sum(0)
    dx = (gamma * rstd / B) * (B * dy - dbeta - xhat * dgamma)
    return dx.astype(x.dtype), dgamma, dbeta


def bn_eval_fwd(x, gamma, beta):
    """EVAL/PREDICT: moving_mean=0, moving_var=1 forever (Appendix A-8)."""
    dt = x.dtype
    inv = (1.0 / np.sqrt(dt.type(1.0) + dt.type(BN_EPS))).astype(dt) * gamma
    return x * inv + beta


# ---------------------------------------------------------------- dropout ----
def dropout_fwd(x, rate, mask):
    """Inverted dropout with an injected 0/1 keep mask (Appendix A-9)."""
    if mask is None or rate == 0.0:
        return x
    return x * mask * x.dtype.type(1.0 / (1.0 - rate))


def dropout_bwd(dy, rate, mask):
    if mask is None or rate == 0.0:
        return dy
    return dy * mask * dy.dtype.type(1.0 / (1.0 - rate))


# ------------------------------------------------------------------- loss ----
def sigmoid(z):
    return (1.0 / (1.0 + np.exp(-z))).astype(z.dtype)


def sigmoid_ce_mean(z, y):
    """fm/fm.py:146-149: mean_b[max(z,0) - z*y + log1p(exp(-|z|))]; returns (loss, dloss/dz)."""
    z = z.reshape(-1)
    y = y.reshape(-1).astype(z.dtype)
    per = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    loss = per.mean(dtype=z.dtype)
    dz = (sigmoid(z) - y) / z.dtype.type(z.shape[0])
    return loss, dz.astype(z.dtype)


# --------------------------------------------------------------- TF-1 Adam ---
class AdamTF1:
    """tf.train.AdamOptimizer(lr), beta1 .9, beta2 .999, eps 1e-8 (fm/fm.py:162).

    * dense grads  -> training_ops ApplyAdam:  m += (g-m)(1-b1); v += (g*g-v)(1-b2);
                      var -= (m*alpha)/(sqrt(v)+eps)
    * sparse grads -> adam.py::_apply_sparse_shared, NON-lazy (Appendix A-5):
                      m = m*b1 (whole var); m[idx] += g(1-b1); v = v*b2; v[idx] += g*g(1-b2);
                      var -= (alpha*m)/(sqrt(v)+eps) (whole var).  Duplicate indices summed first.
    alpha = lr*sqrt(1-b2^t)/(1-b1^t) with the powers kept as running fp products ("epsilon-hat").
    """

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, dtype=np.float32):
        self.dt = np.dtype(dtype).type
        self.lr, self.b1, self.b2, self.eps = map(self.dt, (lr, b1, b2, eps))
        self.b1p, self.b2p = self.b1, self.b2      # beta powers for step t=1
        self.slots = {}

    def alpha(self):
        one = self.dt(1)
        return self.dt(self.lr * np.sqrt(one - self.b2p, dtype=self.dt) / (one - self.b1p))

    def _slot(self, name, var):
        if name not in self.slots:
            self.slots[name] = (np.zeros_like(var), np.zeros_like(var))
        return self.slots[name]

    def apply_dense(self, name, var, g):
        m, v = self._slot(name, var)
        one = self.dt(1)
        a = self.alpha()
        m += (g - m) * (one - self.b1)
        v += (g * g - v) * (one - self.b2)
        var -= (m * a) / (np.sqrt(v) + self.eps)

    def apply_sparse(self, name, var, rows, G, lazy=False):
        """rows: unique int rows [U]; G: [U, ...] summed grads."""
        m, v = self._slot(name, var)
        one = self.dt(1)
        a = self.alpha()
        if lazy:   # the "lazy_rows" throughput mode of this build (NOT TF semantics)
            m[rows] = m[rows] * self.b1 + G * (one - self.b1)
            v[rows] = v[rows] * self.b2 + (G * G) * (one - self.b2)
            var[rows] -= (a * m[rows]) / (np.sqrt(v[rows]) + self.eps)
            return
        m *= self.b1
        m[rows] += G * (one - self.b1)
        v *= self.b2
        v[rows] += (G * G) * (one - self.b2)
        var -= (a * m) / (np.sqrt(v) + self.eps)

    def finish_step(self):
        self.b1p = self.dt(self.b1p * self.b1)
        self.b2p = self.dt(self.b2p * self.b2)


def segment_sum_rows(rows, vals):
    """Dedup + sum duplicates in ascending pair order (TF: unique + unsorted_segment_sum on CPU).
    rows [N] int, vals [N, ...] -> (uniq_sorted [U], G [U, ...])."""
    rows = np.asarray(rows).reshape(-1)
    uniq, inv = np.unique(rows, return_inverse=True)
    G = np.zeros((len(uniq),) + vals.shape[1:], vals.dtype)
    np.add.at(G, inv, vals)     # np.add.at applies in index order -> ascending pair order
    return uniq, G


# ----------------------------------------------------------------- metrics ---
class StreamingAUC:
    """tf.metrics.auc(labels, pred): num_thresholds=200, trapezoidal ROC (Appendix A-11)."""

    def __init__(self, num_thresholds=200):
        n = num_thresholds
        eps = 1e-7
        th = [(i + 1) * 1.0 / (n - 1) for i in range(n - 2)]
        self.th = np.array([0.0 - eps] + th + [1.0 + eps], np.float32)
        self.tp = np.zeros(n, np.float64)
        self.fp = np.zeros(n, np.float64)
        self.tn = np.zeros(n, np.float64)
        self.fn = np.zeros(n, np.float64)

    def update(self, labels, pred):
        y = np.asarray(labels).reshape(-1) > 0.5      # cast to bool
        p = np.asarray(pred, np.float32).reshape(-1)
        gt = p[None, :] > self.th[:, None]
        self.tp += (gt & y).sum(1)
        self.fp += (gt & ~y).sum(1)
        self.fn += (~gt & y).sum(1)
        self.tn += (~gt & ~y).sum(1)

    def result(self):
        tp, fp, tn, fn = (a.astype(np.float32) for a in (self.tp, self.fp, self.tn, self.fn))
        e = np.float32(1e-6)
        tpr = (tp + e) / (tp + fn + e)
        fpr = fp / (fp + tn + e)
        return float(np.sum((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / np.float32(2.0), dtype=np.float32))


class StreamingAccuracy:
    """tf.metrics.accuracy(labels, tf.round(pred)): round-half-to-even."""

    def __init__(self):
        self.total = 0.0
        self.count = 0.0

    def update(self, labels, pred):
        y = np.asarray(labels, np.float32).reshape(-1)
        r = np.round(np.asarray(pred, np.float32).reshape(-1))   # numpy rounds half to even, like tf.round
        self.total += float((r == y).sum())
        self.count += y.shape[0]

    def result(self):
        return self.total / max(self.count, 1.0)
