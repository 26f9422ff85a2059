"""GPU box: optimizer windows against single steps (RSX_ADAM_WINDOW=1) over a matrix of models / batch sizes / schedules --
every variable must be bit-identical.  (tests/test_gpu_adam_window.py holds the small cases; this is the wide sweep.)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tests.test_gpu_adam_window import _est_for, _model_state
from recsys_amd import synthetic
from recsys_amd.dist import EmulatedDataParallel
from recsys_amd.estimator import PackedBatch
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns

layout = CriteoLayout.from_columns(build_feature_columns(16, "indicator_all")[1])
CASES = [("deepfm", 100, 1, 45, 8, 16), ("deepfm", 37, 1, 33, 8, 8), ("deepfm", 1024, 1, 40, 8, 8), ("deepfm", 2048, 1, 26, 8, 8),
         ("dcn", 4096, 1, 22, 8, 8), ("dcn", 1500, 1, 19, 4, 8), ("fm", 1024, 2, 35, 8, 8), ("xdeepfm", 256, 1, 37, 8, 16),
         ("deepfm", 512, 4, 30, 8, 8), ("fm", 4096, 1, 21, 8, 8)]
bad = 0
for kind, B, world, steps, spg, nb in CASES:
    host = synthetic.criteo_id_batches(layout, nb, B, seed=B + steps)
    rng = np.random.default_rng(B)
    logx = [np.log(np.floor(np.exp(rng.normal(2, 1, (B, 13)))) + 1.0).astype(np.float32) for _ in host]
    res, ks = [], []
    for win in ("0", "1"):
        if win == "1":
            os.environ["RSX_ADAM_WINDOW"] = "1"
        else:
            os.environ.pop("RSX_ADAM_WINDOW", None)
        est = _est_for(kind, B)
        if world > 1:
            est.store.dp = est.dist = EmulatedDataParallel(world)
        feats = [PackedBatch({"ids": i, "cont_log": lx} if kind == "xdeepfm" else {"ids": i}, y, device="cuda")
                 for (i, y, _), lx in zip(host, logx)]
        with torch.no_grad():
            est._call_model_fn(feats[0].views()[0], None, "infer")
        ks.append(est._window_len())
        est.train_resident(feats, steps, spg)
        assert est.global_step == steps
        res.append(_model_state(est))
        del est
        torch.cuda.empty_cache()
    os.environ.pop("RSX_ADAM_WINDOW", None)
    diff = [n for n in res[0] if not torch.equal(res[0][n], res[1][n])]
    bad += bool(diff)
    print("%-8s B=%-5d world=%d steps=%-3d spg=%-2d windows of %d vs %d: %s" % (kind, B, world, steps, spg, ks[0], ks[1], "bit-identical" if not diff else "DIFF %s" % diff), flush=True)
print("MATRIX_OK" if not bad else "MATRIX_DIFF")
