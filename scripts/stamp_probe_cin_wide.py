#!/usr/bin/env python
"""Where do the 8-examples-per-workgroup CIN launches (csrc/cin_bf16_wide.hip) spend their microseconds INSIDE a training
step?  xdeepfm.py --cin_bf16 steps on the -DRSX_STAMPS build (scripts/build_stamps.sh), phase stamps of the layer-2 forward
(workgroup (0, 0) and the last one) and of the last layer's data-gradient launch (100 MHz wall clock)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSX_LIB_PATH"] = os.path.join(ROOT, "scripts", "_build", "librsx_stamps.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from recsys_amd import synthetic, xdeepfm  # noqa: E402
from recsys_amd.estimator import Estimator, PackedBatch, RunConfig  # noqa: E402
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns  # noqa: E402

FWD = [(0, "entry"), (1, "loads requested"), (2, "X0 / Xk^T in LDS (loads arrived)"), (3, "barrier"), (4, "field loop done, partials in LDS"),
       (5, "barrier"), (6, "end")]
DX = [(32, "entry"), (33, "loads requested"), (34, "X0 / dpre^T in LDS"), (35, "barrier"), (36, "field loop done"), (37, "barrier"),
      (38, "dX0 partials out, dXk partials in LDS"), (39, "end")]


def main():
    lib = C.CDLL(os.environ["RSX_LIB_PATH"])
    lin, emb = build_feature_columns(16, "numeric+indicator")
    layout = CriteoLayout.from_columns(emb)
    host = synthetic.criteo_id_batches(layout, 8, 256, seed=5)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
              "dropout": 0.5, "deep_layers": "100,100", "cross_layers": "128,128", "max_batch_size": 256, "cin_bf16": True}
    est = Estimator(xdeepfm.model_fn, None, params, RunConfig(use_hip_graph=False, seed=11))
    feats = [PackedBatch({"ids": i, "cont_log": c}, y, device="cuda") for i, y, c in host]
    with torch.no_grad():
        est._call_model_fn(feats[0].views()[0], None, "infer")
    acc, reps = np.zeros(64), 0
    for s in range(48):
        lib.rsx_dbg_stamps_cin_wide_reset()
        est._train_step(*feats[s % 8].views())
        torch.cuda.synchronize()
        if s >= 16:
            buf = (C.c_ulonglong * 64)()
            assert lib.rsx_dbg_stamps_cin_wide(buf) == 0
            t = np.array(list(buf), np.float64)
            r = np.zeros(64)
            r[:32] = np.where(t[:32] > 0, t[:32] - t[0], 0)
            r[32:] = np.where(t[32:] > 0, t[32:] - t[32], 0)
            acc += r
            reps += 1
    t = acc / reps * 0.01
    print("cin_fwd_bf16_wide_k<4> (layer 2), us since workgroup (0,0)'s entry")
    for base, tag in ((0, "workgroup (0,0)"), (8, "workgroup (7, last)")):
        prev = None
        for k, name in FWD:
            v = t[base + k]
            print("  %-20s %-45s %7.2f%s" % (tag, name, v, "" if prev is None else "  (+%.2f)" % (v - prev)))
            prev = v
    print("  latest entry of any workgroup %7.2f   latest end of any workgroup %7.2f" % (t[16], t[17]))
    print("cin_bwd_dx_bf16_wide_k<4> (last layer), us since workgroup (0,0)'s entry")
    prev = None
    for k, name in DX:
        v = t[k]
        print("  %-45s %7.2f%s" % (name, v, "" if prev is None else "  (+%.2f)" % (v - prev)))
        prev = v
    print("  latest entry of any workgroup %7.2f   latest end of any workgroup %7.2f" % (t[48], t[49]))


if __name__ == "__main__":
    main()
