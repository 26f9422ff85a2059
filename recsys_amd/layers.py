"""Small dense layers of the model_fn bodies (the DNN tower is left to
rocBLAS/hipBLASLt through torch -- 37 MFLOP at B=256 is not a hand-kernel target) with the TF-1.x
semantics the reference relies on."""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # tf.layers.batch_normalization default


def dense(x, W, b, relu=False):
    """tf.layers.dense (deepfm/deepfm.py:104)."""
    y = torch.addmm(b, x, W)
    return torch.relu(y) if relu else y


def batch_norm(x, gamma, beta, training):
    """tf.layers.batch_normalization (deepfm/deepfm.py:106).  TRAIN: batch statistics (biased var).
    EVAL/PREDICT: the moving statistics, which the reference never updates (no UPDATE_OPS dependency,
    Appendix A-8), i.e. mean 0 / var 1 forever."""
    if training:
        return F.batch_norm(x, None, None, gamma, beta, True, 0.0, BN_EPS)
    return x * (gamma * (1.0 / math.sqrt(1.0 + BN_EPS))) + beta


def dropout(x, rate, training, mask=None):
    """tf.layers.dropout (deepfm/deepfm.py:107): inverted dropout, TRAIN only.  `mask` (0/1 keep mask)
    injects a fixed mask for parity tests."""
    if not training or rate == 0.0:
        return x
    if mask is not None:
        return x * mask * (1.0 / (1.0 - rate))
    return F.dropout(x, rate, True)


def tower(x, P, pre, n_layers, training, rate, masks=None):
    """[dense(relu) -> BN -> dropout] * n (deepfm/deepfm.py:103-107, xdeepfm/xdeepfm.py:188-191, dcn/dcn.py:146-149)."""
    h = x
    for i in range(n_layers):
        h = dense(h, P[f"{pre}.W{i}"], P[f"{pre}.b{i}"], relu=True)
        h = batch_norm(h, P[f"{pre}.gamma{i}"], P[f"{pre}.beta{i}"], training)
        h = dropout(h, rate, training, None if masks is None else masks[i])
    return h


def sigmoid_ce_mean(logits, labels):
    """tf.reduce_mean(tf.nn.sigmoid_cross_entropy_with_logits) fm/fm.py:146-149."""
    return F.binary_cross_entropy_with_logits(logits.reshape(-1), labels.reshape(-1).to(logits.dtype))


# ---- initialisers (Appendix A-4, A-7) ---------------------------------------------------------
def trunc_normal_(t, std, gen=None):
    torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=gen)
    return t


def glorot_uniform_(t, fan_in, fan_out, gen=None):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        t.uniform_(-lim, lim, generator=gen)
    return t


def glorot_normal_(t, fan_in, fan_out, gen=None):
    std = math.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    return trunc_normal_(t, std, gen)
