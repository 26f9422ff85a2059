"""world_size-2 `gloo` test (CPU) of the data-parallel plumbing in recsys_amd/dist.py:
rank-ordered all-gather of ids and of the packed per-example gradient block, flat dense all-reduce, and the
identity DP(N=2, b) == single(2b) when the gathered data drive the same dedup + segment-sum + TF-1 Adam
(here executed by the oracle, since the HIP kernels need a GPU; SURVEY.md section 4 'Distributed' row)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from oracle import init, models, nn
    from recsys_amd import dist as rdist
    rdist.init_process_group("gloo")
    dp = rdist.DataParallel()
    assert (dp.rank, dp.world) == (rank, world)
    rows = (3, 7, 40, 11)
    off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    F, D, b = 4, 4, 6
    rng = np.random.default_rng(0)                     # same stream on both ranks -> same global batch
    ids_g = np.stack([rng.integers(0, r, world * b) for r in rows], 1).astype(np.int32)
    y_g = rng.integers(0, 2, world * b).astype(np.float64)
    P = init.deepfm_params(1, D, (), np.float64, off, with_dnn=False)     # FM: no BN, so DP == single exactly
    sl = slice(rank * b, (rank + 1) * b)
    m = models.FM(P, off)
    z = m.forward(ids_g[sl])
    _, dz = nn.sigmoid_ce_mean(z, y_g[sl])
    g, s = m.backward(dz / world)                      # MirroredStrategy: loss scaled by 1/N
    # --- product plumbing under test -----------------------------------------------------------
    ids_all = dp.all_gather_rows(torch.from_numpy(ids_g[sl])).numpy()
    assert np.array_equal(ids_all, ids_g)
    rows_local, E, S, y1, cat = m.c
    gy2 = (dz / world)[:, None] @ P["out.W"].T
    dX = np.zeros((b, F * D))
    gy1 = gy2[:, 0] * (y1 > 0)
    dXg, Sg, gy1g, gy2g, ids_pk = dp.gather_example_grads(
        torch.from_numpy(dX.astype(np.float32)), torch.from_numpy(S.astype(np.float32)), torch.from_numpy(gy1.astype(np.float32)),
        torch.from_numpy(np.ascontiguousarray(gy2[:, 1]).astype(np.float32)), ids=torch.from_numpy(ids_g[sl]))
    assert np.array_equal(ids_pk.numpy(), ids_g)                     # int32 ids survive the float32 bit-cast ride
    dXg, Sg, gy1g, gy2g = dp.gather_example_grads(torch.from_numpy(dX), torch.from_numpy(S), torch.from_numpy(gy1),
                                                  torch.from_numpy(np.ascontiguousarray(gy2[:, 1])))
    names = sorted(g)
    flat = torch.from_numpy(np.concatenate([g[k].reshape(-1) for k in names]))
    # the dense arena folded into the same collective (summed in rank order) == the separate all-reduce
    flat2 = flat.clone()
    r2 = dp.gather_example_grads(torch.from_numpy(dX), torch.from_numpy(S), torch.from_numpy(gy1),
                                 torch.from_numpy(np.ascontiguousarray(gy2[:, 1])), dense=flat2)
    # blocked form: no copies after the collective, rank blocks are read in place
    flat3 = flat.clone()
    v = dp.gather_example_grads(torch.from_numpy(dX), torch.from_numpy(S), torch.from_numpy(gy1),
                                torch.from_numpy(np.ascontiguousarray(gy2[:, 1])), dense=flat3, blocked=True)
    bb, stride = v[4]
    buf = dp._keep
    assert bb == b and stride == buf.shape[1] and stride % 4 == 0
    assert v[0].data_ptr() == buf.data_ptr() and torch.equal(buf[:, :dX.size].reshape(world * b, -1), r2[0])
    assert torch.equal(buf[:, dX.size:dX.size + S.size].reshape(world * b, -1), r2[1]) and torch.equal(flat3, flat2)
    dp.all_reduce_sum(flat)
    assert torch.allclose(flat2, flat, rtol=0, atol=1e-15)
    for a_, b_ in zip(r2, (dXg, Sg, gy1g, gy2g)):
        assert torch.equal(a_, b_)
    # --- global update from the gathered blocks (what every rank's HIP kernels would do) -------
    rws = ids_all.astype(np.int64) + off[None, :-1]
    dE = models.fm2_bwd(P["tables"][rws], Sg.numpy(), gy2g.numpy()) + dXg.numpy().reshape(-1, F, D)
    opt = nn.AdamTF1(dtype=np.float64)
    o = 0
    for k in names:
        n = P[k].size
        opt.apply_dense(k, P[k], flat.numpy()[o:o + n].reshape(P[k].shape))
        o += n
    u, G = nn.segment_sum_rows(*models._pairs_field_major(rws, dE))
    opt.apply_sparse("tables", P["tables"], u, G)
    _, g1 = nn.segment_sum_rows(*models._pairs_field_major(rws, np.repeat(gy1g.numpy()[:, None], F, 1)))
    dense = np.zeros_like(P["w1"])
    dense[u] = g1
    opt.apply_dense("w1", P["w1"], dense)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **P)
    dp.barrier()
    dist.destroy_process_group()


def test_dp2_equals_single_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle import init, models, nn
    rows = (3, 7, 40, 11)
    off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    rng = np.random.default_rng(0)
    ids_g = np.stack([rng.integers(0, r, world * 6) for r in rows], 1).astype(np.int32)
    y_g = rng.integers(0, 2, world * 6).astype(np.float64)
    P = init.deepfm_params(1, 4, (), np.float64, off, with_dnn=False)
    models.train_step(models.FM(P, off), nn.AdamTF1(dtype=np.float64), (ids_g,), y_g)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for k in P:
        assert np.array_equal(r0[k], r1[k]), k                       # replicas stay bit-identical
        np.testing.assert_allclose(r0[k], P[k], rtol=0, atol=1e-13, err_msg=k)   # == single batch of N*b


class _FakeDenseArena:
    """The few members of ops.DenseArena that dist.DataParallel's send block touches (the real one needs a GPU)."""

    def __init__(self, n):
        self.n = n
        self.flat, self.m, self.v = torch.zeros(n), torch.zeros(n), torch.zeros(n)
        self.grad = torch.zeros(n)

    def rebind_grad(self, storage):
        g = storage[:self.n]
        g.copy_(self.grad)
        self.grad = g


def _send_block_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from recsys_amd import _lib
    from recsys_amd import dist as rdist
    rdist.init_process_group("gloo")
    dp = rdist.DataParallel()
    n, b_max, widths = 37, 8, [12, 4, 1, 1]                  # dense arena of 37 floats -> the block starts at float 40
    dense = _FakeDenseArena(n)
    send = dp.make_send_block(dense, b_max, widths)
    assert dense.grad.data_ptr() == send.data_ptr()           # the arena's gradient lives at the head of the send block
    for b in (8, 5):                                          # full and ragged last batch
        dense.grad.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
        dX, S, gy2, gy1 = dp.send_views(b)
        assert dX.shape == (b, 12) and S.shape == (b, 4) and gy2.shape == (b,) and gy1.shape == (b,)
        assert dX.data_ptr() == send.data_ptr() + 40 * 4
        base = 1000.0 * (rank + 1)
        dX.copy_(base + torch.arange(b * 12, dtype=torch.float32).view(b, 12))
        S.fill_(base + 0.5)
        gy2.fill_(base + 0.25)
        gy1.fill_(base + 0.125)
        views, (bb, stride), seg = dp.gather_send_block(b, fold_dense=True)
        out = dp._keep
        assert bb == b and stride % 4 == 0 and out.shape == (world, stride) and stride >= 40 + b * 18
        for r in range(world):                                # every rank's block sits `stride` floats after the previous one
            rb = 1000.0 * (r + 1)
            assert torch.equal(out[r, :n], torch.arange(n, dtype=torch.float32) * (r + 1))
            assert torch.equal(out[r, 40:40 + b * 12].view(b, 12), rb + torch.arange(b * 12, dtype=torch.float32).view(b, 12))
            assert torch.equal(out[r, 40 + b * 12:40 + b * 16], torch.full((b * 4,), rb + 0.5))
            assert torch.equal(out[r, 40 + b * 16:40 + b * 17], torch.full((b,), rb + 0.25))
            assert torch.equal(out[r, 40 + b * 17:40 + b * 18], torch.full((b,), rb + 0.125))
        assert views[0].data_ptr() == out.data_ptr() + 40 * 4 and views[0].shape == (b, 12) and views[3].shape == (b,)
        # the optimizer segment that folds the dense all-reduce: B replicas, `stride` floats apart, starting at rank 0's arena
        (sg,) = seg
        assert sg["kind"] == _lib.RSX_ADAM_DENSE and sg["B"] == world and sg["stride"] == stride and sg["n"] == n
        assert sg["g"].data_ptr() == out.data_ptr() and sg["var"] is dense.flat
        # the un-folded form sums the arenas in rank order into the local one
        _, _, seg2 = dp.gather_send_block(b, fold_dense=False)
        assert seg2 is None
        assert torch.equal(dense.grad, torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world)))
    # large arenas (xDeepFM's CIN filters): a true all-reduce of the arena, overlapped with the all-gather of the example
    # block alone -- forced here by dropping the threshold
    os.environ["RSX_DP_ALLREDUCE_MIN_BYTES"] = "64"
    b = 6
    dense.grad.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
    dX, S, gy2, gy1 = dp.send_views(b)
    dX.copy_(1000.0 * (rank + 1) + torch.arange(b * 12, dtype=torch.float32).view(b, 12))
    S.fill_(rank + 0.5)
    gy2.fill_(rank + 0.25)
    gy1.fill_(rank + 0.125)
    views, (bb, stride), seg = dp.gather_send_block(b, fold_dense=True)
    out = dp._keep
    assert seg is None and bb == b and stride % 4 == 0 and stride >= b * 18 and out.shape == (world, stride)
    assert torch.equal(dense.grad, torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world)))   # summed in place
    for r in range(world):
        assert torch.equal(out[r, :b * 12].view(b, 12), 1000.0 * (r + 1) + torch.arange(b * 12, dtype=torch.float32).view(b, 12))
        assert torch.equal(out[r, b * 12:b * 16], torch.full((b * 4,), r + 0.5))
        assert torch.equal(out[r, b * 17:b * 18], torch.full((b,), r + 0.125))
    assert views[0].data_ptr() == out.data_ptr() and views[0].shape == (b, 12)
    del os.environ["RSX_DP_ALLREDUCE_MIN_BYTES"]
    # RSX_DP_OVERLAP: slices of the arena all-reduced asynchronously (per tower layer), then the example block alone
    dense.names = ["b1", "dnn.W0", "dnn.b0", "dnn.W1", "dnn.b1", "dnn.Wout", "out.W"]
    dense.offsets = dict(zip(dense.names, (0, 4, 16, 20, 28, 32, 36)))
    per_layer, rest = rdist.overlap_ranges(dense, [["dnn.W0", "dnn.b0", "dnn.gamma0"], ["dnn.W1", "dnn.b1"]])
    assert per_layer == [(4, 20), (20, 32)] and rest == [(0, 4), (32, 37)]
    try:
        rdist.overlap_ranges(dense, [["dnn.W0", "dnn.W1"], ["dnn.b0"]])
        raise AssertionError("interleaved layers must be refused")
    except _lib.RsxError:
        pass
    dense.grad.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
    dX, S, gy2, gy1 = dp.send_views(b)
    dX.fill_(rank + 1.0)
    hs = [dp.all_reduce_async(dense.grad[lo:hi]) for lo, hi in [per_layer[1]] + rest + [per_layer[0]]]
    dp.wait_all(hs)
    views, (bb, stride), seg = dp.gather_send_block(b, fold_dense=True, dense_done=True)
    assert seg is None and all(h.work is None for h in hs)
    assert torch.equal(dense.grad, torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world)))
    out = dp._keep
    assert out.shape == (world, stride) and views[0].data_ptr() == out.data_ptr()
    for r in range(world):
        assert torch.equal(out[r, :b * 12], torch.full((b * 12,), r + 1.0))
    # the prefetchable ids all-gather: synchronous when nobody issued it, otherwise it waits for the asynchronous launch
    x = torch.full((3, 2), rank, dtype=torch.int32)
    outp = torch.empty(world * 3, 2, dtype=torch.int32)
    op = rdist._PrefetchableAllGather(outp, x, None)
    want = torch.cat([torch.full((3, 2), r, dtype=torch.int32) for r in range(world)])
    op()
    assert torch.equal(outp, want)
    outp.zero_()
    op.issue()
    op.issue()                                                # idempotent while one launch is pending
    op()
    assert torch.equal(outp, want) and op.work is None
    dp.barrier()
    dist.destroy_process_group()


def test_send_block_layout_world2(tmp_path):
    """The zero-copy data-parallel exchange (DataParallel.make_send_block / send_views / gather_send_block) over a real
    2-process gloo group: block layout, rank stride, alignment, the replica-sum optimizer segment."""
    mp.spawn(_send_block_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _window_ids_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from recsys_amd import dist as rdist
    rdist.init_process_group("gloo")
    dp = rdist.DataParallel()
    k, b, F = 4, 5, 3
    # batch j of the window on rank r: ids = 1000*r + 100*j + (example, field)
    local = [torch.arange(b * F, dtype=torch.int32).reshape(b, F) + 1000 * rank + 100 * j for j in range(k)]
    got = rdist.window_global_ids(dp, [{"ids": x} for x in local])
    assert len(got) == k
    for j in range(k):       # global batch j = the ranks' batch j in rank order (what all_gather_rows gives step by step)
        want = torch.cat([torch.arange(b * F, dtype=torch.int32).reshape(b, F) + 1000 * r + 100 * j for r in range(world)])
        assert got[j].shape == (world * b, F) and got[j].is_contiguous() and torch.equal(got[j], want), j
        assert torch.equal(got[j], dp.all_gather_rows(local[j]))
    assert rdist.window_global_ids(None, [{"ids": x} for x in local])[2] is local[2]
    dp.barrier()
    dist.destroy_process_group()


def test_window_ids_all_gather_world2(tmp_path):
    """Optimizer windows under data parallelism: ONE all-gather of the stacked ids of the window's k local batches must give
    the same k global batches (rank order) as k per-step all-gathers."""
    mp.spawn(_window_ids_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _rank_inputs(rank, k, KS, capT, D, n):
    """What rank `rank` contributes to one step of the unique-list exchange (deterministic, distinct per rank)."""
    keys = (torch.arange(k * KS, dtype=torch.int32) + 100000 * (rank + 1)).view(1, k * KS)
    dense = torch.arange(n, dtype=torch.float32) * (rank + 1)
    G = (torch.arange(capT * D, dtype=torch.float32) + 1000.0 * (rank + 1)).view(capT, D)
    gw1 = torch.arange(capT, dtype=torch.float32) + 0.5 * (rank + 1)
    return keys, dense, G, gw1


def _unique_exchange_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from recsys_amd import dist as rdist
    rdist.init_process_group("gloo")
    dp = rdist.DataParallel()
    assert rdist.dp_capture(dp) is False                      # gloo collectives cannot sit in a HIP graph
    k, KS, capT, D, n = 3, 44, 10, 4, 37
    dense = _FakeDenseArena(n)
    send = dp.make_send_block(dense, capT, [D, 1])            # [dense | G [capT, D] | gw1 [capT]]
    keys, dg, G, gw1 = _rank_inputs(rank, k, KS, capT, D, n)
    keys_g = dp.all_gather_keys(keys)
    assert keys_g.shape == (world, k * KS) and keys_g.dtype == torch.int32
    for r in range(world):
        assert torch.equal(keys_g[r:r + 1], _rank_inputs(r, k, KS, capT, D, n)[0])
    Gv, gwv = dp.send_views(capT)
    assert Gv.shape == (capT, D) and gwv.shape == (capT,) and Gv.data_ptr() == send.data_ptr() + 40 * 4
    dense.grad.copy_(dg)
    Gv.copy_(G)
    gwv.copy_(gw1)
    (G0, gw0), (bb, stride), seg = dp.gather_send_block(capT, fold_dense=True)
    out = dp._keep
    assert bb == capT and stride % 4 == 0 and G0.data_ptr() % 16 == 0 and out.shape == (world, stride)
    for r in range(world):                                    # rank r's block: G0 / gw0 + r * stride floats
        _, dr, Gr, gr = _rank_inputs(r, k, KS, capT, D, n)
        base = out.view(-1)
        o = (G0.data_ptr() - out.data_ptr()) // 4 + r * stride
        assert torch.equal(base[o:o + capT * D].view(capT, D), Gr)
        o1 = (gw0.data_ptr() - out.data_ptr()) // 4 + r * stride
        assert torch.equal(base[o1:o1 + capT], gr)
        assert torch.equal(out[r, :n], dr)
    assert seg[0]["B"] == world and seg[0]["stride"] == stride
    torch.save({"keys_g": keys_g, "out": out.clone(), "stride": stride}, os.path.join(out_dir, "rank%d.pt" % rank))
    dp.barrier()
    dist.destroy_process_group()


def test_unique_exchange_plumbing_world2_and_the_loopback_harness_agree(tmp_path):
    """The two collectives of the unique-row-list exchange over a real 2-process gloo group (key blocks in rank order; the send
    block [dense | G | gw1] with rank r's block r * stride floats on), and the single-process LoopbackDataParallel harness the
    GPU parity tests use must hand the optimizer stage the SAME gathered buffers for the same per-rank inputs."""
    world = 2
    mp.spawn(_unique_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from recsys_amd import dist as rdist
    k, KS, capT, D, n = 3, 44, 10, 4, 37

    class _Opt:
        shadow = False

    class _Store:
        opt, embeddings, din = _Opt(), {}, None

    from tests.dp_harness import LoopbackDataParallel
    lb = LoopbackDataParallel(world)
    dense = _FakeDenseArena(n)
    lb.make_send_block(dense, capT, [D, 1])
    store = _Store()
    lb.begin_step()
    for r in range(world):
        lb.enter_rank(r, store)
        assert store.opt.shadow == (r < world - 1)
        keys, dg, G, gw1 = _rank_inputs(r, k, KS, capT, D, n)
        keys_g = lb.all_gather_keys(keys)
        Gv, gwv = lb.send_views(capT)
        dense.grad.copy_(dg)
        Gv.copy_(G)
        gwv.copy_(gw1)
        lb.leave_rank(store)
    (G0, gw0), (bb, stride), seg = lb.gather_send_block(capT, fold_dense=True)
    ref = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    ref1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    assert torch.equal(ref["keys_g"], ref1["keys_g"]) and torch.equal(ref["out"], ref1["out"])
    assert torch.equal(keys_g, ref["keys_g"])                 # (the LAST rank's pass sees every rank's real key block)
    assert stride == ref["stride"] and torch.equal(lb._keep, ref["out"])
    assert rdist.dp_capture(lb) is True and rdist.dp_capture(None) is False
