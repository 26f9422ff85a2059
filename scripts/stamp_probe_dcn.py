#!/usr/bin/env python
"""stamp_probe.py for dcn.py at batch 4096 (SPLIT tower backward): phase stamps of the FIRST d(input) workgroup and the first
dW workgroup of each backward layer, plain path (every kernel alone)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSX_LIB_PATH"] = os.path.join(ROOT, "scripts", "_build", "librsx_stamps.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from recsys_amd import _lib, dcn, synthetic  # noqa: E402
from recsys_amd.estimator import Estimator, PackedBatch, RunConfig  # noqa: E402
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns  # noqa: E402
from scripts.stamp_probe import NAMES  # noqa: E402

B = 4096
L = _lib.lib()
fn = C.CDLL(os.environ["RSX_LIB_PATH"]).rsx_dbg_stamps_tower
lin, emb = build_feature_columns(16, "numeric")
layout = CriteoLayout.from_columns(emb)
host = synthetic.criteo_id_batches(layout, 4, B, seed=3)
params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
          "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": B, "overlap_adam": False, "cross_layers": 3}
est = Estimator(dcn.model_fn, None, params, RunConfig(use_hip_graph=True, seed=1))
feats = [PackedBatch({"ids": i}, y, device="cuda") for i, y, c in host]
with torch.no_grad():
    est._call_model_fn(feats[0].views()[0], None, "infer")
acc = np.zeros((64,), np.float64)
reps = 0
for s in range(30):
    est._train_step(feats[s % 4])
    torch.cuda.synchronize()
    if s >= 10:
        buf = (C.c_ulonglong * 64)()
        assert fn(buf) == 0
        t = np.array(list(buf), np.float64)
        acc += np.where(t > 0, t - (t[36] if t[36] > 0 else t[4]), 0)
        reps += 1
t = acc / reps * 0.01
BIG = {36: "fwd0 big: entry", 37: "fwd0 big: input rows in LDS", 38: "fwd0 big: k-loop done", 39: "fwd0 big: end",
       32: "fwd1 big: entry", 33: "fwd1 big: input rows in LDS (BN + dropout applied)", 34: "fwd1 big: k-loop done", 35: "fwd1 big: end",
       48: "bwd1 big dX: entry", 49: "bwd1 big dX: da tile in LDS", 50: "bwd1 big dX: first pass k-loop done", 51: "bwd1 big dX: end",
       52: "bwd1 big dW: entry", 53: "bwd1 big dW: constants", 54: "bwd1 big dW: all stages done",
       40: "bwd0 big dX: entry", 41: "bwd0 big dX: da tile in LDS", 42: "bwd0 big dX: first pass k-loop done", 43: "bwd0 big dX: end",
       44: "bwd0 big dW: entry", 45: "bwd0 big dW: constants", 46: "bwd0 big dW: all stages done"}
NAMES = dict(NAMES)
NAMES.update(BIG)
print("---- dcn bs %d, plain path: us since fwd0's entry; delta to the previous stamp of the same kernel" % B)
prev = None
for k in NAMES:
    d = "" if prev is None or (k % 4 == 0 and k != 12) else "  (+%.2f)" % (t[k] - t[prev])
    print("%-58s %8.2f%s" % (NAMES[k], t[k], d))
    prev = k
