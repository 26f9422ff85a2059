"""GPU box: Estimator.evaluate / predict throughput (deepfm.py, batch 256 and 4096) over host batches."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recsys_amd import deepfm, synthetic
from recsys_amd.estimator import Estimator, RunConfig
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns

lin, emb = build_feature_columns(16, "indicator_all")
layout = CriteoLayout.from_columns(emb)
for B in (256, 4096):
    host = synthetic.criteo_id_batches(layout, 32, B, seed=5)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
              "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": B}
    est = Estimator(deepfm.model_fn, None, params, RunConfig(device="cuda", seed=1, log_step_count_steps=1000000))

    def fn(n):
        def gen():
            for s in range(n):
                i, y, _ = host[s % 32]
                yield {"ids": i}, y.reshape(-1, 1)
        return gen
    est.train(fn(40), steps=40)
    est.evaluate(fn(40))
    torch.cuda.synchronize()
    n = 600
    t0 = time.time()
    r = est.evaluate(fn(n))
    dt = time.time() - t0
    print("evaluate B=%d: %.1f us per batch, %.2f M examples/s (AUC %.4f)" % (B, dt / n * 1e6, n * B / dt / 1e6, r["AUC"]), flush=True)
    t0 = time.time()
    k = 0
    for p in est.predict(fn(n)):
        k += 1
    dt = time.time() - t0
    print("predict  B=%d: %.1f us per batch" % (B, dt / n * 1e6), flush=True)
