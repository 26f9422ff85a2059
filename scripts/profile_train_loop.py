"""GPU box: cProfile of Estimator.train over TFRecord shards (deepfm.py, batch 256): where the host time of a step goes."""
import cProfile
import os
import pstats
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recsys_amd import deepfm, synthetic
from recsys_amd import input_pipeline as ip
from recsys_amd.estimator import Estimator, RunConfig
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns

lin, emb = build_feature_columns(16, "indicator_all")
layout = CriteoLayout.from_columns(emb)
with tempfile.TemporaryDirectory() as d:
    rng = np.random.default_rng(0)
    files = []
    for k in range(2):
        label, cont, cat = synthetic.criteo_raw_batch(rng, 150000)
        p = os.path.join(d, "part-r-%05d" % k)
        ip.write_criteo_shard(p, label, cont, cat)
        files.append(p)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
              "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": 256}
    est = Estimator(deepfm.model_fn, None, params, RunConfig(device="cuda", seed=1, log_step_count_steps=1000000))
    fn = lambda: ip.criteo_input_fn(files, 256, num_epochs=-1, need_shuffle=True, layout=layout, num_parallel=32)
    est.train(fn, steps=304)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    est.train(fn, steps=2400)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
