"""Split-operand bf16 MFMA path of the CIN layer (csrc/cin_split.hip): ns planes per contraction operand.
  ns = 3 is the parity path: every product exact to 2^-23, so the kernels are compared with the PLAIN fp64 evaluation
  (the oracle's formula, oracle/models.py cin_layer_fwd / cin_layer_bwd) at fp32-accumulation tolerance;
  ns = 1 is compared with the fp64 evaluation in which the operands are rounded to bf16 first (as tests/test_gpu_cin_bf16.py);
  ns = 2 sits in between (2^-16-grade products);
  ns = 4: forward / data gradients with two scaled fp16 planes per operand (three MFMAs per k-step, products to 2^-22), weight
  gradients on three bf16 planes -- held to the SAME tolerances as ns = 3, plus a test with operands spread over 2^40."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _inputs(B, F, H, N, seed, first_layer):
    rng = np.random.default_rng(seed)
    X0 = (rng.standard_normal((B, F, 16)) * 0.3).astype(np.float32)
    Xk = X0 if first_layer else (np.abs(rng.standard_normal((B, H, 16))) * 0.3).astype(np.float32)
    W = (rng.standard_normal((F * H, N)) * 0.1).astype(np.float32)
    c = (rng.standard_normal(N) * 0.1).astype(np.float32)
    return X0, Xk, W, c


SHAPES = [
    (256, 39, 128, 128, False),     # BASELINE config 3, layer 2
    (256, 39, 39, 128, True),       # BASELINE config 3, layer 1
    (7, 5, 6, 20, False),           # ragged everything
    (33, 39, 100, 50, False),
    (1, 3, 16, 16, False),
    (250, 40, 128, 128, False),
    (19, 38, 72, 96, False),
]
TOL = {4: 2e-6, 3: 2e-6}


@pytest.mark.parametrize("ns", [4, 3])
@pytest.mark.parametrize("B,F,H,N,first", SHAPES)
def test_cin_split_forward(B, F, H, N, first, ns):
    from recsys_amd.ops import _ptr, _stream, check, lib
    if first:
        H = F
    X0, Xk, W, c = _inputs(B, F, H, N, B * 7 + H, first)
    t = lambda a: torch.from_numpy(a).cuda()
    tX0, tW, tc = t(X0), t(W), t(c)
    tXk = tX0 if first else t(Xk)
    w16 = torch.empty(int(lib().rsx_cin_split_weight_elems(F, H, N, ns)), dtype=torch.int16, device="cuda")
    out = torch.full((B, N, 16), float("nan"), device="cuda")
    check(lib().rsx_cin_split_prep((C.c_void_p * 1)(tW.data_ptr()), (C.c_void_p * 1)(w16.data_ptr()), (C.c_int32 * 1)(H),
                                   (C.c_int32 * 1)(N), 1, F, ns, _stream()))
    check(lib().rsx_cin_split_fwd(_ptr(tX0), _ptr(tXk), _ptr(w16), _ptr(tc), _ptr(out), B, F, H, N, 16, ns, _stream()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    f8 = np.float64
    Xr, Wr = (bf16_round(Xk), bf16_round(W)) if ns == 1 else (Xk, W)
    pre = np.einsum("bfd,bhd,fhn->bnd", X0.astype(f8), Xr.astype(f8), Wr.astype(f8).reshape(F, H, N), optimize=True) + c[None, :, None]
    ref = np.maximum(pre, 0)
    assert np.isfinite(got).all()
    err = _rel(got, ref)
    print("ns=%d B=%d F=%d H=%d N=%d: max |err| / max |out| = %.3g" % (ns, B, F, H, N, err))
    assert err < TOL[ns], err


BWD_SHAPES = [
    (256, 39, 128, 128, False, True, False),     # BASELINE config 3, layer 2 (last layer: direct-connect gradient)
    (256, 39, 39, 128, True, False, True),       # BASELINE config 3, layer 1 (X0 in both roles, accumulating)
    (7, 5, 6, 20, False, True, True),
    (33, 39, 100, 50, False, False, False),
    (1, 3, 16, 16, False, True, False),
    (250, 40, 128, 128, False, True, False),
    (19, 38, 72, 96, False, False, True),
    (700, 39, 32, 16, False, True, False),       # a batch whose X0 slab does not fit the weight-gradient launch's LDS
]
BTOL = {4: 3e-6, 3: 3e-6}


@pytest.mark.parametrize("ns", [4, 3])
@pytest.mark.parametrize("B,F,H,N,first,gs,acc", BWD_SHAPES)
def test_cin_split_backward(B, F, H, N, first, gs, acc, ns):
    """dXk, dX0 (tile partials + the reduce launch), dW, dc.  ns = 3 / 2: against the plain fp64 gradients of the oracle's
    formula; ns = 1: against fp64 with the operands (W, dpre, the products X0 * Xk) rounded to bf16 first."""
    from recsys_amd import _lib
    from recsys_amd.ops import _ptr, _stream, check, lib
    if first:
        H = F
    X0, Xk, W, c = _inputs(B, F, H, N, B * 7 + H + 1, first)
    rng = np.random.default_rng(B + 13)
    dout = rng.standard_normal((B, N, 16)).astype(np.float32)
    gsv = rng.standard_normal(B).astype(np.float32) if gs else None
    wout = rng.standard_normal(N).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    tX0, tW, tc, tdout, twout = t(X0), t(W), t(c), t(dout), t(wout)
    tXk = tX0 if first else t(Xk)
    w16 = torch.empty(int(lib().rsx_cin_split_weight_elems(F, H, N, ns)), dtype=torch.int16, device="cuda")
    ws = torch.empty(int(lib().rsx_cin_split_bwd_workspace_bytes(B, N, ns)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, N, 16, device="cuda")
    check(lib().rsx_cin_split_prep((C.c_void_p * 1)(tW.data_ptr()), (C.c_void_p * 1)(w16.data_ptr()), (C.c_int32 * 1)(H),
                                   (C.c_int32 * 1)(N), 1, F, ns, _stream()))
    check(lib().rsx_cin_split_fwd(_ptr(tX0), _ptr(tXk), _ptr(w16), _ptr(tc), _ptr(out), B, F, H, N, 16, ns, _stream()))
    a = 1 if (acc or first) else 0
    dX0 = torch.full((B, F, 16), 0.5 if a else float("nan"), device="cuda")
    dXk = dX0 if first else torch.full((B, H, 16), 0.25 if acc else float("nan"), device="cuda")
    dW, dc = torch.full_like(tW, float("nan")), torch.full_like(tc, float("nan"))
    tgs = t(gsv) if gs else None
    pt = torch.full((int(lib().rsx_cin_bf16_dx0_parts_floats(B, F, H)),), float("nan"), device="cuda")
    check(lib().rsx_cin_split_bwd_dx(_ptr(tX0), _ptr(tXk), _ptr(w16), _ptr(out), _ptr(tdout), _ptr(tgs) if gs else None,
                                     _ptr(twout) if gs else None, _ptr(dXk), a, _ptr(pt), _ptr(ws), B, F, H, N, 16, ns, _stream()))
    check(lib().rsx_cin_dx0_reduce((C.c_void_p * 1)(pt.data_ptr()), (C.c_int32 * 1)((H + 15) // 16), 1, _ptr(dX0), a, B, F, 16,
                                   _stream()))
    job = (_lib.CinDwJob * 1)(_lib.CinDwJob(tXk.data_ptr(), ws.data_ptr(), dW.data_ptr(), dc.data_ptr(), H, N, B))
    check(lib().rsx_cin_split_bwd_dw(_ptr(tX0), job, 1, B, F, 16, ns, _stream()))
    torch.cuda.synchronize()
    got = dict(out=out.cpu().numpy(), dX0=dX0.cpu().numpy(), dXk=dXk.cpu().numpy(), dW=dW.cpu().numpy(), dc=dc.cpu().numpy())
    f8 = np.float64
    rd = bf16_round if ns == 1 else (lambda x: x)
    W3 = rd(W).astype(f8).reshape(F, H, N)
    g = dout.astype(f8) + (gsv[:, None, None] * wout[None, :, None] if gs else 0.0)
    dpre = g * (got["out"] > 0)                        # the kernel masks with ITS forward output
    dpre_r = rd(dpre.astype(np.float32)).astype(f8)
    Z = (X0[:, :, None, :] * Xk[:, None, :, :]).astype(np.float32)          # fp32 product, rounded once
    dW_r = np.einsum("bfhd,bnd->fhn", rd(Z).astype(f8), dpre_r, optimize=True).reshape(F * H, N)
    dc_r = dpre.sum((0, 2))
    dXk_r = np.einsum("bfd,fhn,bnd->bhd", X0.astype(f8), W3, dpre_r, optimize=True)
    dX0_r = np.einsum("bhd,fhn,bnd->bfd", Xk.astype(f8), W3, dpre_r, optimize=True)
    base = 0.5 if a else 0.0
    if first:
        ref = dict(dW=dW_r, dc=dc_r, dX0=base + dXk_r + dX0_r, dXk=base + dXk_r + dX0_r)
    else:
        ref = dict(dW=dW_r, dc=dc_r, dX0=base + dX0_r, dXk=(0.25 if acc else 0.0) + dXk_r)
    for k in ("dc", "dXk", "dX0", "dW"):
        assert np.isfinite(got[k]).all(), k
        err = _rel(got[k], ref[k])
        assert err < BTOL[ns], (k, err)
    # the dX0 reduce riding in the weight-gradient launch (rsx_cin_split_bwd_dw_dx0): the same sums in the same order -> the same bits
    dX0b = torch.full((B, F, 16), 0.5 if a else float("nan"), device="cuda")
    dXkb = dX0b if first else torch.full((B, H, 16), 0.25 if acc else float("nan"), device="cuda")
    dWb, dcb = torch.full_like(tW, float("nan")), torch.full_like(tc, float("nan"))
    pt.fill_(float("nan"))
    check(lib().rsx_cin_split_bwd_dx(_ptr(tX0), _ptr(tXk), _ptr(w16), _ptr(out), _ptr(tdout), _ptr(tgs) if gs else None,
                                     _ptr(twout) if gs else None, _ptr(dXkb), a, _ptr(pt), _ptr(ws), B, F, H, N, 16, ns, _stream()))
    jobb = (_lib.CinDwJob * 1)(_lib.CinDwJob(tXk.data_ptr(), ws.data_ptr(), dWb.data_ptr(), dcb.data_ptr(), H, N, B))
    check(lib().rsx_cin_split_bwd_dw_dx0(_ptr(tX0), jobb, 1, B, F, 16, ns, (C.c_void_p * 1)(pt.data_ptr()),
                                         (C.c_int32 * 1)((H + 15) // 16), 1, _ptr(dX0b), a, _stream()))
    torch.cuda.synchronize()
    assert np.array_equal(dX0b.cpu().numpy(), got["dX0"]) and np.array_equal(dWb.cpu().numpy(), got["dW"])
    assert np.array_equal(dcb.cpu().numpy(), got["dc"]) and np.array_equal(dXkb.cpu().numpy(), got["dXk"])


@pytest.mark.parametrize("ns", [4, 3])
def test_cin_split_forward_wide_dynamic_range(ns):
    """Operands whose magnitudes are spread over many binades (per example and per field by 2^+-20, elements by another 2^8):
    the power-of-two scales of mode 4 are per accumulation chain, so every example / field keeps its own precision."""
    from recsys_amd.ops import _ptr, _stream, check, lib
    B, F, H, N = 64, 39, 128, 128
    rng = np.random.default_rng(77)
    X0 = (rng.standard_normal((B, F, 16)) * 0.3).astype(np.float32)
    Xk = (np.abs(rng.standard_normal((B, H, 16))) * np.exp2(rng.integers(-8, 1, (B, H, 16))) *
          np.exp2(rng.integers(-20, 21, (B, 1, 1)))).astype(np.float32)
    W = (rng.standard_normal((F, H, N)) * np.exp2(rng.integers(-8, 1, (F, H, N))) * np.exp2(rng.integers(-20, 21, (F, 1, 1)))).astype(np.float32)
    c = np.zeros(N, np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    tX0, tXk, tW, tc = t(X0), t(Xk), t(W.reshape(F * H, N)), t(c)
    w16 = torch.empty(int(lib().rsx_cin_split_weight_elems(F, H, N, ns)), dtype=torch.int16, device="cuda")
    out = torch.full((B, N, 16), float("nan"), device="cuda")
    check(lib().rsx_cin_split_prep((C.c_void_p * 1)(tW.data_ptr()), (C.c_void_p * 1)(w16.data_ptr()), (C.c_int32 * 1)(H),
                                   (C.c_int32 * 1)(N), 1, F, ns, _stream()))
    check(lib().rsx_cin_split_fwd(_ptr(tX0), _ptr(tXk), _ptr(w16), _ptr(tc), _ptr(out), B, F, H, N, 16, ns, _stream()))
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    f8 = np.float64
    # per (example, field) term T = Xk W_f, compared per example: err relative to the example's own largest |term sum|
    pre = np.einsum("bfd,bhd,fhn->bnd", X0.astype(f8), Xk.astype(f8), W.astype(f8), optimize=True)
    mag = np.einsum("bfd,bhd,fhn->bnd", np.abs(X0).astype(f8), Xk.astype(f8), np.abs(W).astype(f8), optimize=True)
    ref = np.maximum(pre, 0)
    assert np.isfinite(got).all()
    err = float((np.abs(got - ref) / mag.max(axis=(1, 2), keepdims=True)).max())
    print("ns=%d wide range: max |err| / max_b sum |terms| = %.3g" % (ns, err))
    assert err < 2e-6, err


@pytest.mark.parametrize("B,F,N,gs", [(256, 39, 128, False), (250, 39, 128, True), (7, 5, 20, True), (64, 40, 48, False), (9, 17, 33, True)])
def test_cin_split_first_layer_field_split(B, F, N, gs):
    """Mode 4, first layer (H = F, dXk IS dX0): rsx_cin_split_bwd_dx with acc_dxk = 2 splits the fields over two workgroups per tile
    of h; the second half's dXk arrives as one more tile partial.  Same sums, one association different: against the unsplit
    launch within fp32 rounding of dX0's largest element, and against the fp64 gradients at the backward tolerance."""
    from recsys_amd.ops import _ptr, _stream, check, lib
    ns, H = 4, F
    X0, _, W, c = _inputs(B, F, H, N, B * 3 + F, True)
    rng = np.random.default_rng(B + 5)
    dout = rng.standard_normal((B, N, 16)).astype(np.float32)
    gsv = rng.standard_normal(B).astype(np.float32)
    wout = rng.standard_normal(N).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    tX0, tW, tc, tdout, tgs, twout = t(X0), t(W), t(c), t(dout), t(gsv), t(wout)
    w16 = torch.empty(int(lib().rsx_cin_split_weight_elems(F, H, N, ns)), dtype=torch.int16, device="cuda")
    ws = torch.empty(int(lib().rsx_cin_split_bwd_workspace_bytes(B, N, ns)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, N, 16, device="cuda")
    check(lib().rsx_cin_split_prep((C.c_void_p * 1)(tW.data_ptr()), (C.c_void_p * 1)(w16.data_ptr()), (C.c_int32 * 1)(H),
                                   (C.c_int32 * 1)(N), 1, F, ns, _stream()))
    check(lib().rsx_cin_split_fwd(_ptr(tX0), _ptr(tX0), _ptr(w16), _ptr(tc), _ptr(out), B, F, H, N, 16, ns, _stream()))
    HT = (H + 15) // 16
    res = []
    for mode in (0, 2):
        dX0 = torch.full((B, F, 16), float("nan"), device="cuda")
        pt = torch.full((int(lib().rsx_cin_bf16_dx0_parts_floats(B, F, H)) + B * F * 16,), float("nan"), device="cuda")
        check(lib().rsx_cin_split_bwd_dx(_ptr(tX0), _ptr(tX0), _ptr(w16), _ptr(out), _ptr(tdout), _ptr(tgs) if gs else None,
                                         _ptr(twout) if gs else None, _ptr(dX0), mode, _ptr(pt), _ptr(ws), B, F, H, N, 16, ns, _stream()))
        check(lib().rsx_cin_dx0_reduce((C.c_void_p * 1)(pt.data_ptr()), (C.c_int32 * 1)(HT + (1 if mode == 2 else 0)), 1, _ptr(dX0), 1,
                                       B, F, 16, _stream()))
        torch.cuda.synchronize()
        res.append(dX0.cpu().numpy())
    assert np.isfinite(res[1]).all()
    assert _rel(res[1], res[0]) < 1e-6, _rel(res[1], res[0])
    f8 = np.float64
    g = dout.astype(f8) + (gsv[:, None, None] * wout[None, :, None] if gs else 0.0)
    dpre = g * (out.cpu().numpy() > 0)
    W3 = W.astype(f8).reshape(F, H, N)
    ref = np.einsum("bfd,fhn,bnd->bhd", X0.astype(f8), W3, dpre, optimize=True) + np.einsum("bhd,fhn,bnd->bfd", X0.astype(f8), W3, dpre, optimize=True)
    assert _rel(res[1], ref) < BTOL[4], _rel(res[1], ref)
    # outside the envelope: H != F
    bad = lib().rsx_cin_split_bwd_dx(_ptr(tX0), _ptr(tX0), _ptr(w16), _ptr(out), _ptr(tdout), None, None, _ptr(dX0), 2, _ptr(pt), _ptr(ws),
                                     B, F, H + 1 if H < 128 else H - 1, N, 16, ns, _stream())
    assert bad != 0
