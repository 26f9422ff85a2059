"""Round-4 launch fusions at the op level: each fused / riding form against the launches it replaces, bit for bit.
  * rsx_gather_cross_fwd == rsx_gather_fm_fwd + rsx_cross_fwd (dcn/dcn.py:125-142)
  * rsx_segsum_partials_ride == rsx_segsum_partials + the riders' own launches (rsx_tower_reduce_dw_jobs layout 1,
    rsx_cross_reduce_run, rsx_vec_reduce_run)
  * rsx_cross_bwd_defer + rsx_cross_reduce_run == rsx_cross_bwd
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import criteo
from tests.test_gpu_embedding import _arena, synth_ids

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,rows,L", [(256, None, 3), (37, None, 1), (4099, None, 3), (130, (3, 7, 4, 11, 6), 8)])
def test_gather_cross_fwd_equals_the_two_launches(B, rows, L):
    from recsys_amd.ops import CrossLayers
    rng = np.random.default_rng(B + L)
    row_off = criteo.row_offsets() if rows is None else np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    a, _, _ = _arena(row_off, 16, B, rng, with_w1=False)
    dim = a.F * 16
    W = torch.from_numpy(rng.standard_normal((L, dim)).astype(np.float32) * 0.05).cuda()
    Bc = torch.from_numpy(rng.standard_normal((L, dim)).astype(np.float32) * 0.05).cuda()
    wout = torch.from_numpy(rng.standard_normal(dim).astype(np.float32) * 0.1).cuda()
    ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
    two, one = CrossLayers(dim, L, B), CrossLayers(dim, L, B)
    assert one.fused_gather_ok(a)
    E2, _, _, _ = a.gather(ids)
    s2, _, cz2 = two.forward(E2, W, Bc, wout=wout)
    E1, s1, cz1 = one.gather_forward(a, ids, W, Bc, wout)
    torch.cuda.synchronize()
    assert torch.equal(E1, E2) and torch.equal(s1, s2) and torch.equal(cz1, cz2)


def test_cross_bwd_defer_and_scatter_riders_equal_their_own_launches():
    """dcn.py's shapes at batch 4096: the cross backward with its reduce deferred, dW partial tiles in the large-batch layout, two
    partial-vector reduces -- once as their own launches, once riding in the scatter's stage A; the scatter's outputs too."""
    from recsys_amd import _lib
    from recsys_amd.ops import CrossLayers, make_scatter_riders, _ptr, _stream
    L_ = _lib.lib()
    rng = np.random.default_rng(4)
    B, dim, Lc = 4096, 624, 3
    row_off = criteo.row_offsets()
    g = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
    x0, W, Bc, wout, gz, dXt = g(B, dim) * 0.3, g(Lc, dim) * 0.05, g(Lc, dim) * 0.05, g(dim) * 0.1, g(B) * 0.01, g(B, dim) * 0.01
    sb, K, N, G, n = 7, 100, 52, 37, 1000
    part, vp = g(sb, 112, 64), [g(G, n), g(G, n)]
    res = {}
    for ride in (False, True):
        a, _, _ = _arena(row_off, 16, B, np.random.default_rng(9), with_w1=False)
        ids = torch.from_numpy(synth_ids(np.random.default_rng(5), B, row_off)).cuda()
        a.field_sort(ids)
        op = CrossLayers(dim, Lc, B)
        op.forward(x0, W, Bc, wout=wout)
        dW, dB, dwo, dX = torch.zeros(Lc, dim).cuda(), torch.zeros(Lc, dim).cuda(), torch.zeros(dim).cuda(), dXt.clone()
        cjob = op.backward(x0, W, Bc, dW, dB, dX, True, gz=gz, wout=wout, dwout=dwo, defer_reduce=True)
        assert cjob.n == (2 * Lc + 1) * dim
        # a dW job in the row-block layout [sb][K + 1 -> 16][N -> 16] and two partial-vector jobs
        odW, odb = torch.zeros(K, N).cuda(), torch.zeros(N).cuda()
        dj = _lib.DwReduceJob(part.data_ptr(), odW.data_ptr(), odb.data_ptr(), sb, K, N, 1)
        vo = [torch.zeros(n).cuda(), torch.zeros(n).cuda()]
        vj = [_lib.VecReduceJob(vp[i].data_ptr(), vo[i].data_ptr(), G, n) for i in range(2)]
        riders = make_scatter_riders([dj], cjob, vj)
        if not ride:
            from recsys_amd.ops import run_scatter_riders
            run_scatter_riders(riders)
            riders = None
        blk = None
        _lib.check(L_.rsx_segsum_partials_ride(_ptr(a.tables), None, _ptr(dX), None, None, _ptr(a.perm), _ptr(a.seg_off),
                                               _ptr(a.uniq_row), C.byref(a.partials), a.w1_mask, B, a.F, a.D, a.stride, -1, blk,
                                               None if riders is None else C.byref(riders), _stream()), "rsx_segsum_partials_ride")
        torch.cuda.synchronize()
        res[ride] = [t.clone() for t in (dW, dB, dwo, dX, odW, odb, vo[0], vo[1], a.G, a.P)]
    for x, y in zip(res[False], res[True]):
        assert torch.equal(x, y)
    # and the deferred cross reduce equals the two-launch entry
    op = CrossLayers(dim, Lc, B)
    op.forward(x0, W, Bc, wout=wout)
    dW, dB, dwo, dX = torch.zeros(Lc, dim).cuda(), torch.zeros(Lc, dim).cuda(), torch.zeros(dim).cuda(), dXt.clone()
    op.backward(x0, W, Bc, dW, dB, dX, True, gz=gz, wout=wout, dwout=dwo)
    torch.cuda.synchronize()
    for x, y in zip(res[True][:4], (dW, dB, dwo, dX)):
        assert torch.equal(x, y)
    # reference values of the riders: plain sums in the documented orders
    np.testing.assert_allclose(res[True][4].cpu().numpy(), part.sum(0)[:K, :N].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(res[True][5].cpu().numpy(), part.sum(0)[K, :N].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(res[True][6].cpu().numpy(), vp[0].sum(0).cpu().numpy(), rtol=1e-5, atol=1e-5)
