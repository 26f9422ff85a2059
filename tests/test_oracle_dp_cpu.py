"""The oracle's data-parallel TRAIN step (oracle.models.train_step_dp: N replicas on N different batches, MirroredStrategy
semantics -- fm/fm.py:184-194) pinned against the oracle's single-process step: for a model without
batch-norm DP(N, b) IS the step on the concatenated batch of N*b examples (fp64: to rounding), with batch-norm it is not
(statistics per replica) -- both checked, so that the GPU loopback tests (tests/test_gpu_dp_loopback.py) compare the HIP
path with the right thing."""
import copy

import numpy as np

from oracle import init, models, nn


def _setup(seed, rows=(3, 7, 40, 11, 600), D=4):
    off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    rng = np.random.default_rng(seed)
    return off, rng


def _ids(rng, n, off):
    return np.stack([rng.integers(0, off[f + 1] - off[f], n) for f in range(len(off) - 1)], 1).astype(np.int32)


def test_dp_step_of_a_batchnorm_free_model_is_the_step_on_the_concatenated_batch():
    off, rng = _setup(0)
    N, b, D = 3, 7, 4
    P1 = init.deepfm_params(1, D, (), np.float64, off, with_dnn=False)
    P2 = copy.deepcopy(P1)
    o1, o2 = nn.AdamTF1(dtype=np.float64), nn.AdamTF1(dtype=np.float64)
    for step in range(3):
        ids = [_ids(rng, b, off) for _ in range(N)]
        ys = [rng.integers(0, 2, b).astype(np.float64) for _ in range(N)]
        losses, _ = models.train_step_dp(models.FM(P1, off), o1, [(i,) for i in ids], ys)
        loss_g, _ = models.train_step(models.FM(P2, off), o2, (np.concatenate(ids),), np.concatenate(ys))
        assert abs(np.mean(losses) - loss_g) < 1e-14
        for k in P1:
            np.testing.assert_allclose(P1[k], P2[k], rtol=0, atol=1e-13, err_msg=k)


def test_dp_step_keeps_batchnorm_statistics_per_replica():
    """DeepFM: DP(2, b) differs from single(2b) (BN over b rows per replica), equals it when both replicas hold the SAME
    batch (then the per-replica statistics are the global ones and the 1/N-scaled gradients add up to the global ones)."""
    off, rng = _setup(1)
    b, D, layers = 8, 4, (6, 5)
    P0 = init.deepfm_params(2, D, layers, np.float64, off)
    ids_a, ids_b = _ids(rng, b, off), _ids(rng, b, off)
    ya, yb = rng.integers(0, 2, b).astype(np.float64), rng.integers(0, 2, b).astype(np.float64)

    def run_dp(idl, yl):
        P = copy.deepcopy(P0)
        models.train_step_dp(models.DeepFM(P, off, len(layers), 0.0), nn.AdamTF1(dtype=np.float64), [(i,) for i in idl], yl)
        return P

    def run_single(i, y):
        P = copy.deepcopy(P0)
        models.train_step(models.DeepFM(P, off, len(layers), 0.0), nn.AdamTF1(dtype=np.float64), (i,), y)
        return P

    same = run_dp([ids_a, ids_a], [ya, ya])
    ref = run_single(ids_a, ya)
    # (identical replicas: every replica's statistics are those of the batch itself; the 2b-row batch [a; a] has the same
    # mean and biased variance, so single(2b) would match too -- compare with single(b): gradients sum to the same values)
    for k in same:
        np.testing.assert_allclose(same[k], ref[k], rtol=0, atol=1e-12, err_msg=k)
    dp = run_dp([ids_a, ids_b], [ya, yb])
    glob = run_single(np.concatenate([ids_a, ids_b]), np.concatenate([ya, yb]))
    assert max(float(np.abs(dp[k] - glob[k]).max()) for k in dp if k.startswith("dnn.W")) > 1e-6
