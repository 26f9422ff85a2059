"""GPU box: soak / race check of the overlapped split optimizer.  Trains DeepFM bs256 for N steps twice from identical
weights and batches -- (a) the production path (sort riding in the first forward launch, untouched-row sweep carried by the
tower launches, scatter fused with the touched-row Adam, HIP graphs of 8 steps) and (b) the plain path (stand-alone sort,
segment-sum, ONE full Adam sweep, no graphs) -- and compares every parameter.  The split is exact arithmetic, so any
difference beyond fp32 re-association of long segments (none at this batch size) points at a race."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from recsys_amd import deepfm, synthetic
from recsys_amd.estimator import Estimator, PackedBatch, RunConfig
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
MODEL = sys.argv[2] if len(sys.argv) > 2 else "deepfm"          # deepfm (bs 256) | dcn (bs 4096: two-stage scatter, split dW)
EMU = int(sys.argv[3]) if len(sys.argv) > 3 else 0              # > 0: both runs through the data-parallel step with an emulated
                                                                # world of EMU replicas (send block, eager train_op, replica-sum Adam)
from recsys_amd import dcn
B = 256 if MODEL == "deepfm" else 4096
mfn = deepfm.model_fn if MODEL == "deepfm" else dcn.model_fn
lin, emb = build_feature_columns(16, "indicator_all" if MODEL == "deepfm" else "numeric")
layout = CriteoLayout.from_columns(emb)
host = synthetic.criteo_id_batches(layout, 64, B, seed=123)
# PRE > 0: the first PRE steps run over a DIFFERENT pool of batches -- the rows only that pool touches are never touched again,
# so their first moments decay into the denormals (~900 steps) during the soak: the regime the window sweep's operand guard
# (csrc/adam_device.h adam_win_guard1) has to get right on real state, not only on the synthetic grid of the unit test
PRE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
host_pre = synthetic.criteo_id_batches(layout, 64, B, seed=456) if PRE else None
res = []
for overlap, graph in ((True, True), (False, False)):
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
              "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": B, "overlap_adam": overlap, "cross_layers": 3}
    est = Estimator(mfn, None, params, RunConfig(use_hip_graph=graph, adam_mode="tf1_dense", device="cuda", seed=77))
    if EMU:
        from tests.dp_harness import EmulatedDataParallel
        est.store.dp = est.dist = EmulatedDataParallel(EMU)
    feats = [PackedBatch({"ids": i}, y, device="cuda") for i, y, _ in host]
    with torch.no_grad():
        est._call_model_fn(feats[0].views()[0], None, "infer")
    if PRE:
        feats_pre = [PackedBatch({"ids": i}, y, device="cuda") for i, y, _ in host_pre]
        if graph:
            est.train_resident(feats_pre, PRE, 8)
        else:
            for s in range(PRE):
                est._train_step(feats_pre[s % 64])
    if graph:
        est.train_resident(feats, N, 8)
    else:
        for s in range(N):
            est._train_step(feats[s % 64])
    torch.cuda.synchronize()
    a = est.store.embeddings["input_layer"]
    res.append({"tables": a.tables.clone(), "m": a.m_t.clone(), "v": a.v_t.clone(),
                "w1": a.w1.clone() if a.with_w1 else torch.zeros(1), "dense": est.store.dense.flat.clone(),
                "step": est.global_step})
a, b = res
print("steps", a["step"], b["step"])
bad = 0
for k in ("tables", "m", "v", "w1", "dense"):
    d = (a[k] - b[k]).abs().max().item()
    eq = torch.equal(a[k].view(torch.int32), b[k].view(torch.int32))          # bit patterns: signs of zero included
    den = int(((a[k].abs() > 0) & (a[k].abs() < 1.1754944e-38)).sum())
    print("%-7s bit-identical=%s  max|diff|=%.3e  finite=%s  denormal elements=%d" % (k, eq, d, bool(torch.isfinite(a[k]).all()), den))
    # emulated data-parallel: the plain path sums the replicas' dense gradients with torch.sum, the production path inside
    # the optimizer launch -- the same values in the same order, but allow 1 ulp-level drift there
    bad += (not eq) if not EMU else (d > 1e-5)
print("SOAK_OK" if bad == 0 else "SOAK_DIFF")
