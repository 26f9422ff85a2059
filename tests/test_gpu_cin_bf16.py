"""bf16 MFMA path of the CIN layer (csrc/cin_bf16.hip).  Two kinds of checks:
  1. kernel correctness: against an fp64 evaluation in which exactly the operands the kernel rounds (Xk, W, dpre, the
     products X0*Xk of the weight gradient) are rounded to bf16 first -- what remains is fp32 accumulation order, so the
     tolerance is tight (1e-5 relative to the largest entry);
  2. the cost of bf16 itself: against the fp32 path / the fp64 oracle, with the measured tolerance stated here
     (operands carry 8 significant bits: ~4e-3 relative per product, averaging down over the F*H = 4992-term sums)."""
import ctypes as C

import numpy as np
import pytest

from oracle import models

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def bf16_round(x):
    """numpy fp32 -> nearest-even bf16 -> fp32 (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def _run_layer(B, F, H, N, seed, first_layer=False, with_gs=False, acc=False, parts=False):
    from recsys_amd.ops import _ptr, _stream, check, lib
    rng = np.random.default_rng(seed)
    D = 16
    X0 = (rng.standard_normal((B, F, D)) * 0.3).astype(np.float32)
    Xk = X0 if first_layer else (np.abs(rng.standard_normal((B, H, D))) * 0.3).astype(np.float32)
    W = (rng.standard_normal((F * H, N)) * 0.1).astype(np.float32)
    c = (rng.standard_normal(N) * 0.1).astype(np.float32)
    dout = rng.standard_normal((B, N, D)).astype(np.float32)
    gs = rng.standard_normal(B).astype(np.float32) if with_gs else None
    wout = rng.standard_normal(N).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    tX0, tW, tc, tdout, twout = t(X0), t(W), t(c), t(dout), t(wout)
    tXk = tX0 if first_layer else t(Xk)
    w16 = torch.empty(int(lib().rsx_cin_bf16_weight_elems(F, H, N)), dtype=torch.int16, device="cuda")
    ws = torch.empty(int(lib().rsx_cin_bf16_bwd_workspace_bytes(B, N)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, N, D, device="cuda")
    check(lib().rsx_cin_prep_bf16(_ptr(tW), _ptr(w16), F, H, N, _stream()))
    check(lib().rsx_cin_layer_fwd_bf16(_ptr(tX0), _ptr(tXk), _ptr(w16), _ptr(tc), _ptr(out), B, F, H, N, D, None, _stream()))
    dX0 = torch.full((B, F, D), 0.5 if (acc or first_layer) else float("nan"), device="cuda")
    dXk = dX0 if first_layer else torch.full((B, H, D), 0.25 if acc else float("nan"), device="cuda")
    dW, dc = torch.empty_like(tW), torch.empty_like(tc)
    tgs = t(gs) if with_gs else None
    if parts:
        # eight examples per workgroup (csrc/cin_bf16_wide.hip): dX0 as per-h-tile partials + the reduce launch, the weight
        # gradient launch told that the workspace holds one row of bias-gradient partials per example
        from recsys_amd import _lib
        pt = torch.full((int(lib().rsx_cin_bf16_dx0_parts_floats(B, F, H)),), float("nan"), device="cuda")
        check(lib().rsx_cin_layer_bwd_dx_bf16_parts(_ptr(tX0), _ptr(tXk), _ptr(w16), _ptr(out), _ptr(tdout),
                                                    _ptr(tgs) if with_gs else None, _ptr(twout) if with_gs else None, _ptr(dXk),
                                                    1 if (acc or first_layer) else 0, _ptr(pt), _ptr(ws), B, F, H, N, D, _stream()))
        check(lib().rsx_cin_dx0_reduce((C.c_void_p * 1)(pt.data_ptr()), (C.c_int32 * 1)((H + 15) // 16), 1, _ptr(dX0),
                                       1 if (acc or first_layer) else 0, B, F, D, _stream()))
        job = (_lib.CinDwJob * 1)(_lib.CinDwJob(tXk.data_ptr(), ws.data_ptr(), dW.data_ptr(), dc.data_ptr(), H, N, B))
        check(lib().rsx_cin_bwd_dw_bf16(_ptr(tX0), job, 1, B, F, D, None, _stream()))
    else:
        check(lib().rsx_cin_layer_bwd_bf16(_ptr(tX0), _ptr(tXk), _ptr(w16), _ptr(out), _ptr(tdout),
                                           _ptr(tgs) if with_gs else None, _ptr(twout) if with_gs else None, _ptr(dXk),
                                           1 if (acc or first_layer) else 0, _ptr(dX0), 1 if (acc or first_layer) else 0,
                                           _ptr(dW), _ptr(dc), _ptr(ws), B, F, H, N, D, None, _stream()))
    torch.cuda.synchronize()
    got = dict(out=out.cpu().numpy(), dX0=dX0.cpu().numpy(), dXk=dXk.cpu().numpy(), dW=dW.cpu().numpy(), dc=dc.cpu().numpy())
    # ---- fp64 evaluation with the kernel's roundings ---------------------------------------------------------------
    f8 = np.float64
    Xk_r, W_r = bf16_round(Xk).astype(f8), bf16_round(W).astype(f8)
    W3 = W_r.reshape(F, H, N)
    pre = np.einsum("bfd,bhd,fhn->bnd", X0.astype(f8), Xk_r, W3, optimize=True) + c[None, :, None]
    out_r = np.maximum(pre, 0)
    g = dout.astype(f8) + (gs[:, None, None] * wout[None, :, None] if with_gs else 0.0)
    dpre = g * (got["out"] > 0)                       # the kernel masks with ITS forward output
    dpre_r = bf16_round(dpre.astype(np.float32)).astype(f8)
    Z_r = bf16_round((X0[:, :, None, :] * Xk[:, None, :, :]).astype(np.float32)).astype(f8)      # [B,F,H,D], fp32 product rounded once
    dW_r = np.einsum("bfhd,bnd->fhn", Z_r, dpre_r, optimize=True).reshape(F * H, N)
    dc_r = dpre.sum((0, 2))
    dXk_r = np.einsum("bfd,fhn,bnd->bhd", X0.astype(f8), W3, dpre_r, optimize=True)
    dX0_r = np.einsum("bhd,fhn,bnd->bfd", Xk.astype(f8), W3, dpre_r, optimize=True)
    base = 0.5 if (acc or first_layer) else 0.0
    if first_layer:
        ref = dict(out=out_r, dW=dW_r, dc=dc_r, dX0=base + dXk_r + dX0_r, dXk=base + dXk_r + dX0_r)
    else:
        ref = dict(out=out_r, dW=dW_r, dc=dc_r, dX0=base + dX0_r, dXk=(0.25 if acc else 0.0) + dXk_r)
    # ---- plain fp64 (no rounding): the price of bf16 ------------------------------------------------------------------
    W3f = W.astype(f8).reshape(F, H, N)
    pre_f = np.einsum("bfd,bhd,fhn->bnd", X0.astype(f8), Xk.astype(f8), W3f, optimize=True) + c[None, :, None]
    return got, ref, np.maximum(pre_f, 0)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("B,F,H,N,first,gs,acc", [
    (256, 39, 128, 128, False, True, False),     # BASELINE config 3, layer 2 (last layer: direct-connect gradient)
    (256, 39, 39, 128, True, False, True),       # BASELINE config 3, layer 1 (X0 in both roles, accumulating)
    (7, 5, 6, 20, False, True, True),            # ragged everything: odd batch, H, N not multiples of 16 / 32
    (33, 39, 100, 50, False, False, False),
    (1, 3, 16, 16, False, True, False),
    (250, 40, 128, 128, False, True, False),     # F = 40: every wave of the wide launches has five fields; ragged last group
    (19, 39, 72, 96, False, False, True),        # three k-steps in both directions
])
@pytest.mark.parametrize("parts", [False, True])
def test_cin_bf16_kernels_match_fp64_with_the_same_roundings(B, F, H, N, first, gs, acc, parts):
    if first:
        H = F
    got, ref, _ = _run_layer(B, F, H, N, seed=B * 7 + H, first_layer=first, with_gs=gs, acc=acc, parts=parts)
    for k in ("out", "dc", "dXk", "dX0", "dW"):
        assert np.isfinite(got[k]).all(), k
        assert _rel(got[k], ref[k]) < 2e-5, (k, _rel(got[k], ref[k]))


def test_cin_bf16_forward_error_vs_fp32_semantics_is_bounded():
    """The approximation itself, at the BASELINE shape: relative to the largest activation, bf16 operands cost < 1e-2
    (measured ~2e-3); the fp32 path is at 1e-6 on the same data."""
    got, _, exact = _run_layer(256, 39, 128, 128, seed=99, with_gs=True)
    err = _rel(got["out"], exact)
    assert err < 1e-2, err
    print("bf16 CIN forward, max |err| / max |out| = %.3g" % err)


def test_xdeepfm_bf16_trajectory_stays_close_to_fp32():
    """200-step training trajectories, identical weights / batches / dropout seeds, CIN in fp32 vs bf16: the measured
    drift of logits and loss is what DESIGN.md quotes as this path's tolerance."""
    from recsys_amd import synthetic, xdeepfm
    from recsys_amd.estimator import Estimator, PackedBatch, RunConfig
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    lin, emb = build_feature_columns(16, "numeric+indicator")
    layout = CriteoLayout.from_columns(emb)
    host = synthetic.criteo_id_batches(layout, 8, 256, seed=5)
    res = {}
    for bf in (False, True):
        params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
                  "dropout": 0.5, "deep_layers": "100,100", "cross_layers": "128,128", "max_batch_size": 256, "cin_bf16": bf}
        est = Estimator(xdeepfm.model_fn, None, params, RunConfig(use_hip_graph=False, seed=11))
        feats = [PackedBatch({"ids": i, "cont_log": c}, y, device="cuda") for i, y, c in host]
        with torch.no_grad():
            est._call_model_fn(feats[0].views()[0], None, "infer")
        losses = []
        for s in range(200):
            losses.append(float(est._train_step(*feats[s % 8].views())))
        with torch.no_grad():
            z = est._call_model_fn(feats[0].views()[0], None, "infer").predictions["prob"].cpu().numpy().reshape(-1)
        res[bf] = (np.array(losses), z)
    dl = np.abs(res[True][0] - res[False][0])
    dpv = np.abs(res[True][1] - res[False][1])
    print("bf16 vs fp32 CIN over 200 steps: max |dloss| = %.3g, mean |dloss| = %.3g, first-10-steps max |dloss| = %.3g, "
          "final |dprob| max = %.3g mean = %.3g" % (dl.max(), dl.mean(), dl[:10].max(), dpv.max(), dpv.mean()))
    assert np.isfinite(res[True][0]).all()
    # measured on MI355X (r02): max |dloss| 6e-3, final max |dprob| 0.095 (two trajectories through 200 dropout-0.5 steps
    # drift apart on individual examples; the per-step loss stays within 1e-2 and the first steps within 1e-3)
    assert dl.max() < 3e-2 and dl[:10].max() < 2e-3 and dpv.max() < 0.3 and dpv.mean() < 3e-2
    assert res[True][0][-20:].mean() < res[True][0][:20].mean()          # it trains


@pytest.mark.parametrize("bf16", [False, True])
def test_xdeepfm_sweep_carriers_cover_the_whole_update(bf16):
    """Exact split of the TF-1 Adam update over the carrier launches (CIN forward, tower, the ONE merged weight-gradient
    launch of the bf16 path, scatter): every variable after 12 steps must be BIT-IDENTICAL to the plain path (stand-alone
    sweep) -- a slice that no launch executes, or executes twice, would show here."""
    from recsys_amd import synthetic, xdeepfm
    from recsys_amd.estimator import Estimator, PackedBatch, RunConfig
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    lin, emb = build_feature_columns(16, "numeric+indicator")
    layout = CriteoLayout.from_columns(emb)
    host = synthetic.criteo_id_batches(layout, 4, 128, seed=9)
    states = []
    for overlap in (False, True):
        params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
                  "dropout": 0.5, "deep_layers": "100,100", "cross_layers": "128,128", "max_batch_size": 128, "cin_bf16": bf16,
                  "overlap_adam": overlap}
        est = Estimator(xdeepfm.model_fn, None, params, RunConfig(use_hip_graph=False, seed=3))
        feats = [PackedBatch({"ids": i, "cont_log": c}, y, device="cuda") for i, y, c in host]
        with torch.no_grad():
            est._call_model_fn(feats[0].views()[0], None, "infer")
        for s in range(12):
            est._train_step(*feats[s % 4].views())
        torch.cuda.synchronize()
        st = est.store
        a1, a2 = st.embeddings["input_layer"], st.embeddings["input_layer_1"]
        states.append([t.clone() for t in (a1.tables, a1.m_t, a1.v_t, a1.w1, a1.m_w, a1.v_w, a2.tables, a2.m_t, a2.v_t,
                                          st.dense.flat, st.dense.m, st.dense.v)])
    for name, x, y in zip(("t1", "m1", "v1", "w1", "mw", "vw", "t2", "m2", "v2", "dense", "dm", "dv"), *states):
        assert torch.isfinite(x).all(), name
        if bf16:
            assert torch.equal(x, y), (name, float((x - y).abs().max()))
        else:
            # fp32 path: a weight-gradient launch that carries a sweep slice uses a smaller tile configuration
            # (cin.hip: co-resident sweep workgroups inherit the kernel's footprint), i.e. another summation order of
            # dW -- rounding-level differences, while a missing / doubled slice would be ~1e-3 (one Adam step)
            assert float((x - y).abs().max()) < 2e-6, (name, float((x - y).abs().max()))
