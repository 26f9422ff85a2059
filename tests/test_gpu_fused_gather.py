"""Round 4: the input_layer lookup riding in the FIRST tower-forward launch (rsx_gather_tower_fwd0, csrc/tower.hip
tower_gather_fwd_k) against the two launches it replaces (rsx_gather_fm_fwd + rsx_tower_fwd_layer) and the oracle.
The fused launch keeps the stand-alone kernels' lane mapping and summation orders: every output is compared BIT FOR BIT."""
import ctypes as C

import numpy as np
import pytest

from oracle import criteo, models
from tests.parity_util import synth_ids

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _setup(B, rows, N, seed, mask=None):
    from recsys_amd.ops import EmbeddingArena
    rng = np.random.default_rng(seed)
    row_off = criteo.row_offsets() if rows is None else np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    R, F, D = int(row_off[-1]), len(row_off) - 1, 16
    tables = rng.standard_normal((R, D)).astype(np.float32) * 0.25
    w1 = rng.standard_normal(R).astype(np.float32) * 0.1
    a = EmbeddingArena(row_off, D, max(B, 1), "cuda", with_w1=True, w1_field_mask=mask, tables=tables, w1=w1)
    ids = synth_ids(rng, B, row_off)
    W = torch.from_numpy(rng.standard_normal((F * D, N)).astype(np.float32) * 0.05).cuda()
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32) * 0.1).cuda()
    return a, tables, w1, row_off, ids, W, b


def _two_launches(a, ids_t, W, b, B, N):
    from recsys_amd import _lib
    from recsys_amd.ops import _ptr, _stream
    E, S, y1, y2 = a.gather(ids_t, fm=True, first_order=True)
    out = torch.zeros(B, N, device="cuda")
    fstat = torch.zeros((B + 15) // 16, 2, N, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().rsx_tower_fwd_layer(_ptr(E), _ptr(W), _ptr(b), _ptr(out), _ptr(fstat), None, None, None, None, None,
                                              None, 0, 0, 0.0, B, a.F * a.D, N, None, None, None, 0, _stream()), "rsx_tower_fwd_layer")
    return E, S, y1, y2, out, fstat


def _one_launch(a, ids_t, W, b, B, N, sort_job=None, with_s=True):
    from recsys_amd import _lib
    from recsys_amd.ops import _ptr, _stream
    E, S, y1, y2 = a.gather_outputs(B, fm=with_s, first_order=with_s)
    out = torch.zeros(B, N, device="cuda")
    fstat = torch.zeros((B + 15) // 16, 2, N, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().rsx_gather_tower_fwd0(_ptr(a.tables), _ptr(a.w1) if with_s else None, _ptr(a.row_off), _ptr(ids_t),
                                                _ptr(E), _ptr(S), _ptr(y1), _ptr(y2), a.w1_mask, a.F, a.D, _ptr(W), _ptr(b),
                                                _ptr(out), _ptr(fstat), B, N, None if sort_job is None else C.byref(sort_job),
                                                None, None, 0, _stream()), "rsx_gather_tower_fwd0")
    return E, S, y1, y2, out, fstat


@pytest.mark.parametrize("B,rows,N,mask", [(256, None, 100, None), (1, None, 100, None), (37, (3, 7, 4, 11, 6), 36, 0b10101),
                                           (300, None, 100, None), (16, (5,), 16, None), (100, tuple(range(1, 50)), 200, None),
                                           (511, (1000, 3) * 32, 52, (1 << 63) | 5)])
def test_fused_gather_first_layer_is_the_bits_of_the_two_launches(B, rows, N, mask):
    a, tables, w1, row_off, ids, W, b = _setup(B, rows, N, seed=B + N, mask=mask)
    ids_t = torch.from_numpy(ids).cuda()
    want = _two_launches(a, ids_t, W, b, B, N)
    got = _one_launch(a, ids_t, W, b, B, N)
    torch.cuda.synchronize()
    for name, g, w in zip(("E", "S", "y1", "y2", "a0", "fstat"), got, want):
        assert torch.equal(g, w), name
    # and the oracle (fp32 tolerance 1e-5, north_star)
    _, Eo, y1o, So, y2o = models.gather_fm_fwd(tables, w1, ids, row_off)
    assert np.array_equal(got[0].cpu().numpy().reshape(Eo.shape), Eo)
    if mask is None:
        np.testing.assert_allclose(got[2].cpu().numpy(), y1o, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[3].cpu().numpy(), y2o, rtol=1e-5, atol=2e-5)
    a0 = np.maximum(Eo.reshape(B, -1).astype(np.float64) @ W.cpu().numpy().astype(np.float64) + b.cpu().numpy(), 0.0)
    np.testing.assert_allclose(got[4].cpu().numpy(), a0, rtol=1e-5, atol=1e-5)


def test_fused_gather_without_fm_outputs_and_with_a_riding_sort():
    """S / y1 / y2 / w1 NULL (dcn.py's lookup has no first-order / FM term) and the step's dedup sort as extra workgroups of the
    same launch: the sort's outputs equal the stand-alone sort's."""
    B, N = 200, 100
    a, tables, w1, row_off, ids, W, b = _setup(B, None, N, seed=5)
    ids_t = torch.from_numpy(ids).cuda()
    a.field_sort(ids_t)
    torch.cuda.synchronize()
    ref = {k: getattr(a, k).clone() for k in ("perm", "seg_off", "uniq_row", "nuniq")}
    slot_ref = a.slot.clone()
    want = _two_launches(a, ids_t, W, b, B, N)
    job = a.sort_job(ids_t)
    got = _one_launch(a, ids_t, W, b, B, N, sort_job=job, with_s=False)
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[4], want[4]) and torch.equal(got[5], want[5])
    assert got[1] is None and got[2] is None and got[3] is None
    nu = ref["nuniq"].cpu().numpy()
    assert np.array_equal(a.nuniq.cpu().numpy(), nu)
    for f in range(a.F):
        n = int(nu[f])
        assert torch.equal(a.uniq_row[f * a.stride:f * a.stride + n], ref["uniq_row"][f * a.stride:f * a.stride + n])
        assert torch.equal(a.perm[f * a.stride:f * a.stride + B], ref["perm"][f * a.stride:f * a.stride + B])
    assert torch.equal(a.slot, slot_ref)


def test_fused_gather_envelope():
    from recsys_amd import _lib
    L = _lib.lib()
    assert L.rsx_gather_tower_fwd0_supported(256, 39, 16) == 1
    assert L.rsx_gather_tower_fwd0_supported(4096, 39, 16) == 0      # large batches: the wide first layer has its own kernels
    assert L.rsx_gather_tower_fwd0_supported(256, 39, 32) == 0
    d = torch.zeros(64, device="cuda")
    rc = L.rsx_gather_tower_fwd0(d.data_ptr(), None, d.data_ptr(), d.data_ptr(), d.data_ptr(), None, None, None, 0, 2, 32,
                                 d.data_ptr(), d.data_ptr(), d.data_ptr(), None, 4, 4, None, None, None, 0, None)
    assert rc == -3      # RSX_EUNSUPPORTED (include/rsx.h)


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_deepfm_train_parity_with_and_without_the_fused_gather(fuse, monkeypatch):
    """The DeepFM TRAIN step against the oracle through both forms of the first launch (RSX_FORMS fuse_gather)."""
    from tests.parity_util import deepfm_parity_run
    monkeypatch.setenv("RSX_FORMS", "fuse_gather=%s" % fuse)
    err, losses, perr = deepfm_parity_run(B=64, steps=3, seed=21, dropout=0.5, return_all=True)
    assert err < 1e-5, err
    for lg, lo in losses:
        assert abs(lg - lo) < 1e-5, losses
    assert max(perr.values()) < 5e-5, perr


@pytest.mark.parametrize("B", [256, 37, 1])
def test_fm_fused_forward_and_head_is_the_bits_of_the_two_launches(B):
    """rsx_gather_fm_head (fm.py's forward + head in one launch, round 4) against rsx_gather_fm_fwd + rsx_fm_head_terms: S,
    prob, gy1, gy2 and the rows of dense-gradient terms (16 examples each, pre-added in example order) bit for bit; the rows' column
    sums against rsx_fm_head's own fp64 reduction (1e-6: fp32 additions in example order)."""
    from recsys_amd import _lib
    from recsys_amd.ops import _ptr, _stream
    a, tables, w1, row_off, ids, W, b = _setup(B, None, 4, seed=B)
    L = _lib.lib()
    ids_t = torch.from_numpy(ids).cuda()
    rng = np.random.default_rng(B)
    c0 = torch.tensor([0.05], device="cuda")
    wo = torch.from_numpy(rng.standard_normal(2).astype(np.float32)).cuda()
    bo = torch.tensor([-0.1], device="cuda")
    lab = torch.from_numpy((rng.random(B) < 0.4).astype(np.float32)).cuda()
    stride, nd, offs = 16, 12, (0, 4, 8)
    f = lambda *shape: torch.full(shape, 7.0, device="cuda")
    # two launches
    E, S, y1, y2 = a.gather(ids_t, fm=True, first_order=True)
    prob0, g10, g20, terms0 = f(B), f(B), f(B), f((B + 15) // 16, stride)
    _lib.check(L.rsx_fm_head_terms(_ptr(y1), _ptr(y2), _ptr(c0), _ptr(wo), _ptr(bo), _ptr(lab), _ptr(prob0), _ptr(g10), _ptr(g20),
                                   None, None, None, None, _ptr(terms0), stride, nd, *offs, 1.0 / B, B, None, _stream()))
    # the reducing form (reference for the sums)
    prob1, g11, g21 = f(B), f(B), f(B)
    dwo, dbo, dc0, loss = f(2), f(1), f(1), f(1)
    _lib.check(L.rsx_fm_head(_ptr(y1), _ptr(y2), _ptr(c0), _ptr(wo), _ptr(bo), _ptr(lab), _ptr(prob1), _ptr(g11), _ptr(g21),
                             _ptr(dwo), _ptr(dbo), _ptr(dc0), _ptr(loss), 1.0 / B, B, None, _stream()))
    # one launch
    S2, prob2, g12, g22, terms2 = f(B, 16), f(B), f(B), f(B), f((B + 15) // 16, stride)
    _lib.check(L.rsx_gather_fm_head(_ptr(a.tables), _ptr(a.w1), _ptr(a.row_off), _ptr(ids_t), _ptr(S2), a.w1_mask, _ptr(c0), _ptr(wo),
                                    _ptr(bo), _ptr(lab), _ptr(prob2), _ptr(g12), _ptr(g22), _ptr(terms2), stride, nd, *offs, 1.0 / B,
                                    B, a.F, a.D, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(S2, S) and torch.equal(prob2, prob0) and torch.equal(g12, g10) and torch.equal(g22, g20)
    assert torch.equal(prob1, prob0) and torch.equal(g11, g10)
    assert torch.equal(terms2, terms0), (terms2 - terms0).abs().max()
    t = terms0.double().sum(0).cpu().numpy()
    np.testing.assert_allclose([t[4], t[5]], dwo.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(t[8], dbo.item(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(t[0], dc0.item(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(t[12] / B, loss.item(), rtol=1e-6)
    assert float(terms0[:, [1, 2, 3, 6, 7, 9, 10, 11, 13, 14, 15]].abs().max()) == 0.0
