import sys, os
sys.path.insert(0, ".")
import torch
from tests.test_gpu_adam_window import _deepfm_est, _state
from recsys_amd import synthetic
from recsys_amd.estimator import PackedBatch
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
layout = CriteoLayout.from_columns(build_feature_columns(16, "indicator_all")[1])
host = synthetic.criteo_id_batches(layout, 16, 256, seed=321)
def run(win, steps, spg):
    est = _deepfm_est(max(win, 1), graph=win > 0, overlap=win > 0)
    feats = [PackedBatch({"ids": i}, y, device="cuda") for i, y, _ in host]
    with torch.no_grad():
        est._call_model_fn(feats[0].views()[0], None, "infer")
    if win:
        est.train_resident(feats, steps, spg)
    else:
        for s_ in range(steps):
            est._train_step(feats[s_ % 16])
    return _state(est)
ref = {}
cases = [(8, s, 8) for s in (24, 25, 26, 32, 33, 40)]
for win, steps, spg in cases:
    if steps not in ref:
        ref[steps] = run(0, steps, 0)
    st = run(win, steps, spg)
    bad = {k: int((st[k] != ref[steps][k]).sum()) for k in st if not torch.equal(st[k], ref[steps][k])}
    print("window=%d steps=%d spg=%d:" % (win, steps, spg), "OK" if not bad else "DIFF %s" % bad, flush=True)
