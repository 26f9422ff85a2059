"""GPU box: host time of one hipGraphLaunch of a captured DeepFM optimizer window (8 steps, 58 kernel nodes), GPU idle vs busy."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recsys_amd import deepfm, synthetic
from recsys_amd.estimator import Estimator, PackedBatch, RunConfig
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns

lin, emb = build_feature_columns(16, "indicator_all")
layout = CriteoLayout.from_columns(emb)
params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
          "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": 256}
est = Estimator(deepfm.model_fn, None, params, RunConfig(device="cuda", seed=1, log_step_count_steps=1000000))
host = synthetic.criteo_id_batches(layout, 64, 256, seed=1)
pbs = [PackedBatch({"ids": i}, y, pin=True) for i, y, c in host]
with torch.no_grad():
    est._call_model_fn(pbs[0].to("cuda").views()[0], None, "infer")
K = est._window_len()
for w in range(8):
    est._train_window_packed(pbs[w * K:(w + 1) * K])
torch.cuda.synchronize()
key = [k for k in est._graphs if k[0] == "packedwin"][0]
sets = est._graphs[key]["sets"]
g0, g1 = sets[0]["graph"], sets[1]["graph"]
for mode in ("idle", "busy"):
    ts = []
    for rep in range(20):
        torch.cuda.synchronize()
        if mode == "busy":
            g1.replay()
        t0 = time.perf_counter()
        g0.replay()
        ts.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    print("graph launch, GPU %s: host time median %.0f us (min %.0f, max %.0f)" % (mode, float(np.median(ts)), min(ts), max(ts)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); g0.replay(); e1.record(); torch.cuda.synchronize()
print("GPU time of the window: %.0f us" % (e0.elapsed_time(e1) * 1e3))
