#!/usr/bin/env python
"""Where do the latency-bound tower kernels spend their microseconds?  Runs DeepFM bs-256 training steps on a
-DRSX_STAMPS build of librsx.so (scripts/_build/librsx_stamps.so, built by scripts/build_stamps.sh) and prints the time
between the phase stamps of one workgroup per kernel (100 MHz wall clock), for the plain path (every kernel alone) and for
the overlapped path (sweep slices riding)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSX_LIB_PATH"] = os.path.join(ROOT, "scripts", "_build", "librsx_stamps.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from recsys_amd import _lib, deepfm, synthetic  # noqa: E402
from recsys_amd.estimator import Estimator, PackedBatch, RunConfig  # noqa: E402
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns  # noqa: E402

NAMES = {0: "fwd1: entry", 1: "fwd1: BN stats of the previous layer reduced", 2: "fwd1: k-loop done", 3: "fwd1: end",
         4: "fwd0: entry", 32: "gfwd0: W operand loads issued", 33: "gfwd0: ids + offsets arrived (rows computed)",
         34: "gfwd0: rows arrived", 35: "gfwd0: LDS tile written", 5: "fwd0: (no prologue) / gfwd0: workgroup barrier",
         6: "fwd0: k-loop done", 7: "fwd0: end",
         8: "head: entry", 9: "head: BN stats reduced", 12: "head: activations + dot product reduced (incl. the wait for the row loads)",
         13: "head: loss + dz", 10: "head: dy stored, column partials reduced", 11: "head: end",
         16: "bwd0 dX tile: entry", 17: "bwd0 dX: column constants", 18: "bwd0 dX: k-loop done", 19: "bwd0 dX: end",
         20: "bwd0 dW tile: entry", 21: "bwd0 dW: column constants", 22: "bwd0 dW: k-loop done", 23: "bwd0 dW: end",
         24: "bwd1 dX tile: entry", 25: "bwd1 dX: column constants", 26: "bwd1 dX: k-loop done", 27: "bwd1 dX: end",
         28: "bwd1 dW tile: entry", 29: "bwd1 dW: column constants", 30: "bwd1 dW: k-loop done", 31: "bwd1 dW: end"}


def main():
    L = _lib.lib()
    fn = C.CDLL(os.environ["RSX_LIB_PATH"]).rsx_dbg_stamps_tower
    lin, emb = build_feature_columns(16, "indicator_all")
    layout = CriteoLayout.from_columns(emb)
    host = synthetic.criteo_id_batches(layout, 4, 256, seed=3)
    for overlap in (False, True):
        params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
                  "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": 256, "overlap_adam": overlap}
        est = Estimator(deepfm.model_fn, None, params, RunConfig(use_hip_graph=True, seed=1))
        feats = [PackedBatch({"ids": i}, y, device="cuda") for i, y, c in host]
        with torch.no_grad():
            est._call_model_fn(feats[0].views()[0], None, "infer")
        acc = np.zeros((64,), np.float64)
        reps = 0
        for s in range(40):
            est._train_step(feats[s % 4])
            torch.cuda.synchronize()
            if s >= 10:
                buf = (C.c_ulonglong * 64)()
                assert fn(buf) == 0
                t = np.array(list(buf), np.float64)
                acc += np.where(t > 0, t - t[4], 0)       # relative to the entry of fwd0 (the step's second launch)
                reps += 1
        t = acc / reps * 0.01                              # us
        print("---- overlap_adam=%s: us since fwd0's entry; delta to the previous stamp of the same kernel" % overlap)
        prev = None
        for k in NAMES:
            d = "" if prev is None or (k % 4 == 0 and k not in (12, 32, 36)) else "  (+%.2f)" % (t[k] - t[prev])
            print("%-58s %8.2f%s" % (NAMES[k], t[k], d))
            prev = k


if __name__ == "__main__":
    main()
