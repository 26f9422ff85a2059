"""Data-parallel code path on ONE GPU (world_size 1 over RCCL): the all-gather / all-reduce plumbing, the global sort +
segment-sum on gathered blocks and HIP-graph capture of the collectives.  With one replica the numbers must equal
the single-process run, i.e. stay within the oracle tolerance.  (N > 1 is covered by tests/test_dist_gloo.py on CPU.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %r)
from tests.parity_util import deepfm_parity_run
# (injected dropout masks are per-step tensors, which a captured graph cannot follow: the graph case runs dropout 0)
for kind, graph, drop in (("deepfm", False, 0.5), ("deepfm", True, 0.0), ("dcn", False, 0.5), ("dcn", True, 0.0)):
    err, losses, perr = deepfm_parity_run(B=64, steps=5, seed=31, rows=(3, 7, 40, 11, 600), layers=(32, 16), return_all=True,
                                          kind=kind, dropout=drop, use_graph=graph, data_parallel=True)
    assert err < 1e-5, (kind, err)
    assert all(abs(a - b) < 1e-5 for a, b in losses), (kind, losses)
    assert max(perr.values()) < 5e-5, (kind, perr)
# din.py: the fused step through RCCL at world 1 (keys all-gather, send block, blocked scatter) against the single-replica step
import numpy as np, torch
from oracle import init
from recsys_amd import din, synthetic, dist as rdist
from tests.parity_util import make_estimator
K, n_item, n_cate, B, Pn = 16, 300, 20, 24, 12
rng = np.random.default_rng(5)
P = init.din_params(2, K, n_item, n_cate, np.float32)
base = {"embedding_size": K, "learning_rate": 1e-3, "dropout": 0.0, "n_item": n_item, "n_cate": n_cate, "max_batch_size": B}
ests = []
for use_dp in (True, False):
    est = make_estimator(din.model_fn, dict(base), use_graph=use_dp)
    if use_dp:
        est.store.dp = est.dist = rdist.DataParallel()
    ests.append(est)
for step in range(3):
    b = synthetic.din_batch(rng, B, Pn, n_item, n_cate)
    losses = []
    for est in ests:
        f = {k: torch.from_numpy(b[k]).cuda() for k in ("i_id", "i_cate", "u_iid_seq", "u_icat_seq")}
        if not est.store.built:
            with torch.no_grad():
                est._call_model_fn(f, None, "infer")
            st = est.store
            with torch.no_grad():
                st.embeddings["i_id"].table.copy_(torch.from_numpy(P["item_emb"]))
                st.embeddings["i_cate"].table.copy_(torch.from_numpy(P["cate_emb"]))
            st.dense.load({k: v for k, v in P.items() if k in st.dense.params})
        losses.append(float(est._train_step(f, torch.from_numpy(b["label"]).cuda())))
    assert abs(losses[0] - losses[1]) < 1e-6, losses
for name in ("i_id", "i_cate", "i_item"):
    assert float((ests[0].store.embeddings[name].table - ests[1].store.embeddings[name].table).abs().max()) < 2e-6, name
assert float((ests[0].store.dense.flat - ests[1].store.dense.flat).abs().max()) < 2e-6
import torch.distributed as dist
dist.destroy_process_group()
print("DP_OK")
""" % ROOT


@pytest.mark.parametrize("capture_collectives,overlap,direct", [("1", "0", "1"), ("0", "0", "1"), ("1", "1", "1"), ("0", "1", "1"),
                                                                ("0", "0", "0"), ("0", "1", "0")])
def test_dp_world1_rccl_matches_oracle(capture_collectives, overlap, direct):
    """direct '1' (default, round 6): the collectives are RCCL calls through the C ABI on the step's own stream (dist.DirectComm);
    '0': torch.distributed (ProcessGroupNCCL), rounds 1-5.  capture '1' (the default with direct): collectives captured into the
    step's HIP graph; '0': graph SEGMENTS with eager collectives between them.
    overlap '1' = RSX_DP_OVERLAP: DeepFM's dense arena all-reduced per tower layer from inside backward -- on a side HIP
    stream forked from the step's (direct), asynchronous ProcessGroupNCCL launches (torch) -- instead of riding in the step's
    all-gather."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               RSX_DP_CAPTURE=capture_collectives, RSX_DP_OVERLAP=overlap, RSX_DP_DIRECT=direct)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert "DP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


COMM_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import torch
from recsys_amd import dist as rdist
rdist.init_process_group("nccl")
dp = rdist.DataParallel()
assert dp.comm is not None and (dp.comm.rank, dp.comm.world) == (0, 1), "RCCL through the C ABI did not come up"
assert rdist.dp_capture(dp)                    # captured collectives are the default with the direct communicator
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(1, 4099, device="cuda", generator=g)
ids = torch.randint(0, 1 << 30, (256, 39), device="cuda", dtype=torch.int32, generator=g)
grad = torch.randn(1000, device="cuda", generator=g)
odd = torch.arange(7, device="cuda", dtype=torch.uint8)            # a block that is not a multiple of 4 bytes
# eager
assert torch.equal(dp.all_gather_rows(ids), ids) and torch.equal(dp.all_gather_rows(x), x) and torch.equal(dp.all_gather_rows(odd), odd)
g0 = grad.clone()
assert torch.equal(dp.all_reduce_sum(grad), g0)
out = torch.empty(1, 4096, device="cuda")
xs = torch.randn(1, 4096, device="cuda", generator=g)
dp._overlapped_allreduce_allgather(grad, out, xs)
assert torch.equal(out, xs) and torch.equal(grad, g0)
cnt = torch.tensor([5, 7], device="cuda", dtype=torch.int64)        # integer counters: the control plane (torch)
assert dp.all_reduce_sum(cnt).tolist() == [5, 7]
# captured into a HIP graph, replayed on new data: the collectives are graph nodes
static_in = torch.zeros(1, 4096, device="cuda")
static_g = torch.zeros(1000, device="cuda")
static_out = torch.empty(1, 4096, device="cuda")
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, capture_error_mode="thread_local"):
    dp._overlapped_allreduce_allgather(static_g, static_out, static_in)
    y = dp.all_gather_rows(static_in * 2.0)
    h = dp.all_reduce_async(static_g)          # the side stream of RSX_DP_OVERLAP, forked and joined inside the capture
    z = static_in + 1.0
    dp.wait_all([h])
    w = static_g * 3.0
for rep in range(3):
    a = torch.randn(1, 4096, device="cuda", generator=g)
    b = torch.randn(1000, device="cuda", generator=g)
    static_in.copy_(a); static_g.copy_(b)
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, a) and torch.equal(y, a * 2.0) and torch.equal(z, a + 1.0) and torch.equal(w, b * 3.0)
dp.comm.close()
import torch.distributed as dist
dist.destroy_process_group()
print("COMM_OK")
""" % ROOT


def test_direct_comm_world1_eager_and_captured():
    """dist.DirectComm = rsx_comm_* / rsx_all_gather / rsx_all_reduce_* (include/rsx.h) over RCCL at world 1: values, the
    non-float control-plane fallback, and HIP-graph capture + replay of the collectives (incl. the side-stream fork / join)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("RSX_DP_DIRECT", None)
    r = subprocess.run([sys.executable, "-c", COMM_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert "COMM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("exchange", ["examples", "unique"])
@pytest.mark.parametrize("kind,B,world,overlap", [("deepfm", 48, 3, 0), ("dcn", 40, 2, 0), ("deepfm", 300, 4, 0),
                                                  ("deepfm", 600, 4, 0), ("fm", 56, 3, 0), ("deepfm", 48, 3, 1)])
def test_blocked_scatter_of_replicated_batch_equals_single_batch(kind, B, world, overlap, exchange, monkeypatch):
    """Multi-block data-parallel compute on one GPU: `world` identical replicas of a batch b (collectives replaced by
    local tiling) must train exactly like ONE process on the batch repeated `world` times -- same BN statistics, same
    mean loss, gradients summed over replicas with the 1/N loss scale.  Exercises the global dedup sort, the scatter
    reading rank blocks in place from the gathered buffer, and (B*world > 1024) its two-stage form.
    overlap = 1: the RSX_DP_OVERLAP exchange (per-layer dense all-reduce, example block alone in the all-gather).
    exchange: "examples" = the pre-dedup per-example block + one global dedup sort (rounds 1-4); "unique" (round 5, default) =
    every replica's own dedup + segment-sum, unique (row, sum) lists exchanged and merged in rank order (the replicas' partial
    sums are added in another association than the single process adds the examples: 2e-6).  Identical replicas cannot catch a
    wrong rank stride -- tests/test_gpu_dp_loopback.py runs DISTINCT batches against the oracle."""
    monkeypatch.setenv("RSX_DP_OVERLAP", str(overlap))
    monkeypatch.setenv("RSX_DP_EXCHANGE", exchange)
    import numpy as np
    import torch
    from oracle import init
    from recsys_amd import dcn, deepfm, fm
    from tests.dp_harness import EmulatedDataParallel
    from recsys_amd.estimator import PackedBatch
    from tests.parity_util import load_oracle_weights, make_estimator, small_columns, synth_ids
    rows, D, layers = (3, 7, 40, 11, 600), 16, (32, 16)
    row_off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    lin, emb = small_columns(rows, D)
    rng = np.random.default_rng(7)
    mfn = {"deepfm": deepfm.model_fn, "dcn": dcn.model_fn, "fm": fm.model_fn}[kind]
    base = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": D, "learning_rate": 1e-3,
            "dropout": 0.0, "deep_layers": ",".join(map(str, layers)), "cross_layers": 2}
    P = init.deepfm_params(3, D, layers, np.float32, row_off, with_dnn=kind != "fm") if kind in ("deepfm", "fm") else \
        init.dcn_params(3, D, layers, 2, np.float32, row_off)
    ests = []
    for w in (world, 1):
        est = make_estimator(mfn, dict(base, max_batch_size=B * (1 if w > 1 else world)))
        if w > 1:
            est.store.dp = EmulatedDataParallel(w)
        ests.append(est)
    ids = [synth_ids(rng, B, row_off) for _ in range(3)]
    ys = [(rng.random(B) < 0.3).astype(np.float32) for _ in range(3)]
    for step in range(3):
        losses = []
        for est, rep in zip(ests, (1, world)):
            i = torch.from_numpy(np.tile(ids[step], (rep, 1))).cuda()
            y = torch.from_numpy(np.tile(ys[step], rep)).cuda()
            if not est.store.built:
                with torch.no_grad():
                    est._call_model_fn({"ids": i}, None, "infer")
                load_oracle_weights(est, P)
            losses.append(float(est._train_step({"ids": i}, y)))
        assert abs(losses[0] - losses[1]) < 1e-6, losses
    a, b = ests
    assert bool(getattr(a.store, "dp_unique", False)) == (exchange == "unique")
    for name in a.store.embeddings:
        ta, tb = a.store.embeddings[name].tables, b.store.embeddings[name].tables
        assert float((ta - tb).abs().max()) < 2e-6, name
        for slot in ("m_t", "v_t"):
            assert float((getattr(a.store.embeddings[name], slot) - getattr(b.store.embeddings[name], slot)).abs().max()) < 2e-6
        if a.store.embeddings[name].with_w1:
            assert float((a.store.embeddings[name].w1 - b.store.embeddings[name].w1).abs().max()) < 2e-6, name
    assert float((a.store.dense.flat - b.store.dense.flat).abs().max()) < 2e-6


def test_xdeepfm_blocked_data_parallel_equals_single_batch():
    """xdeepfm.py's data-parallel step (global sort first, sweep slices in the CIN launches, send block [dX1 | dX2 | g_lin |
    dense], both table sets in one scatter + Adam launch reading rank blocks in place): an emulated world of 2 replicas of
    a batch must train exactly like one process on the batch repeated twice.  Both sides start from the same seeded init."""
    import numpy as np
    import torch
    from recsys_amd import xdeepfm
    from tests.dp_harness import EmulatedDataParallel
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    from tests.parity_util import make_estimator, synth_ids
    B, world = 40, 2
    rng = np.random.default_rng(11)
    lin, emb = build_feature_columns(16, "numeric+indicator")
    row_off = CriteoLayout.from_columns(emb).row_off
    base = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
            "dropout": 0.0, "deep_layers": "32,16", "cross_layers": "16,8"}
    ests = []
    for w in (world, 1):
        est = make_estimator(xdeepfm.model_fn, dict(base, max_batch_size=B * (1 if w > 1 else world)))
        if w > 1:
            est.store.dp = EmulatedDataParallel(w)
        ests.append(est)
    for step in range(3):
        ids = synth_ids(rng, B, np.asarray(row_off))
        logx = np.log(np.floor(np.exp(rng.normal(2, 1, (B, 13)))) + 1.0).astype(np.float32)
        y = (rng.random(B) < 0.3).astype(np.float32)
        losses = []
        for est, rep in zip(ests, (1, world)):
            f = {"ids": torch.from_numpy(np.tile(ids, (rep, 1))).cuda(), "cont_log": torch.from_numpy(np.tile(logx, (rep, 1))).cuda()}
            losses.append(float(est._train_step(f, torch.from_numpy(np.tile(y, rep)).cuda())))
        assert abs(losses[0] - losses[1]) < 1e-6, losses
    a, b = ests
    for name in a.store.embeddings:
        ta, tb = a.store.embeddings[name].tables, b.store.embeddings[name].tables
        assert float((ta - tb).abs().max()) < 2e-6, name
    assert float((a.store.embeddings["input_layer"].w1 - b.store.embeddings["input_layer"].w1).abs().max()) < 2e-6
    assert float((a.store.dense.flat - b.store.dense.flat).abs().max()) < 2e-6


@pytest.mark.parametrize("B,Pn,world,use_graph", [(24, 12, 2, False), (24, 12, 3, True), (96, 100, 2, False)])
def test_din_fused_step_data_parallel_equals_single_batch(B, Pn, world, use_graph):
    """din.py's fused TRAIN step under data parallelism (round 4; din/din.py:204-206 MirroredStrategy): `world` emulated
    replicas of a batch -- keys all-gathered for ONE global dedup sort, [dense arena | value block | bias gradients] exchanged
    straight from the send block, the scatter reading rank blocks in place -- must train like one process on the batch
    repeated `world` times (entries are summed in rank-block order instead of targets-then-histories order: 2e-6).
    (96, 100): 9 696 entries per rank, the multi-launch sort and the two-stage scatter."""
    import numpy as np
    import torch
    from oracle import init
    from recsys_amd import din, synthetic
    from tests.dp_harness import EmulatedDataParallel
    from tests.parity_util import make_estimator
    K, n_item, n_cate = 16, 300, 20
    rng = np.random.default_rng(5)
    P = init.din_params(2, K, n_item, n_cate, np.float32)
    P["item_bias"] += (rng.standard_normal(n_item) * 0.01).astype(np.float32)
    base = {"embedding_size": K, "learning_rate": 1e-3, "dropout": 0.0, "n_item": n_item, "n_cate": n_cate}
    ests = []
    for w in (world, 1):
        est = make_estimator(din.model_fn, dict(base, max_batch_size=B * (1 if w > 1 else world)), use_graph=use_graph)
        if w > 1:
            est.store.dp = EmulatedDataParallel(w)
        ests.append(est)
    keys = ("i_id", "i_cate", "u_iid_seq", "u_icat_seq")
    for step in range(3):
        b = synthetic.din_batch(rng, B, Pn, n_item, n_cate)
        b["i_id"][:2] = 0                                        # target id 0 trains like any row; the histories' 0 is padding
        losses = []
        for est, rep in zip(ests, (1, world)):
            f = {k: torch.from_numpy(np.tile(b[k], (rep,) + (1,) * (b[k].ndim - 1))).cuda() for k in keys}
            y = torch.from_numpy(np.tile(b["label"], rep)).cuda()
            if not est.store.built:
                with torch.no_grad():
                    est._call_model_fn(f, None, "infer")
                st = est.store
                with torch.no_grad():
                    st.embeddings["i_id"].table.copy_(torch.from_numpy(P["item_emb"]))
                    st.embeddings["i_cate"].table.copy_(torch.from_numpy(P["cate_emb"]))
                    st.embeddings["i_item"].table[:, 0].copy_(torch.from_numpy(P["item_bias"]))
                st.dense.load({k: v for k, v in P.items() if k in st.dense.params})
                assert st.din is not None
            losses.append(float(est._train_step(f, y)))
        assert abs(losses[0] - losses[1]) < 1e-6, losses
    a, b2 = ests
    for name in ("i_id", "i_cate", "i_item"):
        ta, tb = a.store.embeddings[name], b2.store.embeddings[name]
        assert float((ta.table - tb.table).abs().max()) < 2e-6, name
        assert float((ta.m - tb.m).abs().max()) < 2e-6 and float((ta.v - tb.v).abs().max()) < 2e-6, name
    assert float((a.store.dense.flat - b2.store.dense.flat).abs().max()) < 2e-6
