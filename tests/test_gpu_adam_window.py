"""Optimizer windows (include/rsx.h rsx_adam_window): k consecutive TRAIN steps share ONE sweep over the rows of the
optimizer state that none of them touches.  TF-1's Adam is not lazy (every row decays and moves every step, SURVEY Appendix
A-5); for an untouched row step t+1 only needs step t's result, so the k updates are applied back to back in registers.  The
window is exact arithmetic -- the same fp32 operations in the same order -- so everything here is compared BIT FOR BIT with
the step-by-step path."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _adam_np(var, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, dense=False):
    """One untouched-row update of TF-1 Adam at step t (1-based), fp32 op by op as csrc/adam_device.h does it."""
    f = np.float32
    p1, p2 = f(1.0), f(1.0)
    for _ in range(t):
        p1, p2 = f(p1 * f(b1)), f(p2 * f(b2))
    alpha = f(f(lr) * np.sqrt(f(1.0) - p2, dtype=np.float32) / (f(1.0) - p1))
    if dense:       # ApplyAdam with g = 0
        m1 = (m + (f(0.0) - m) * f(f(1.0) - f(b1))).astype(f)
        v1 = (v + (f(0.0) - v) * f(f(1.0) - f(b2))).astype(f)
        var1 = (var - (m1 * alpha) / (np.sqrt(v1, dtype=f) + f(eps))).astype(f)
    else:
        m1 = (m * f(b1)).astype(f)
        v1 = (v * f(b2)).astype(f)
        var1 = (var - (alpha * m1) / (np.sqrt(v1, dtype=f) + f(eps))).astype(f)
    return var1, m1, v1


@pytest.mark.parametrize("k", [2, 3, 4, 7, 8])
def test_window_sweep_equals_k_single_sweeps(k):
    """rsx_adam_seg.slot_w: rows no step of the window touches get k updates in one pass, all others are left alone --
    against k stand-alone single-step sweeps of the same rows (which advance the beta powers between them)."""
    from recsys_amd import _lib
    from recsys_amd.ops import AdamTF1
    R, D = 5000, 16
    g = torch.Generator(device="cpu").manual_seed(k)
    tab0, m0 = torch.randn(R, D, generator=g), torch.randn(R, D, generator=g) * 0.01
    v0 = torch.rand(R, D, generator=g) * 1e-4
    w0, mw0, vw0 = torch.randn(R, generator=g), torch.randn(R, generator=g) * 0.01, torch.rand(R, generator=g) * 1e-4
    # the window's slot maps: equally spaced slices of one allocation (rsx_adam_seg.slot_w)
    slot_all = torch.full((k, R + 4), -1, dtype=torch.int32)
    for i in range(k):
        idx = torch.randperm(R, generator=g)[:400]
        slot_all[i, idx] = torch.arange(400, dtype=torch.int32)
    slot_all = slot_all.cuda()
    slots = [slot_all[i] for i in range(k)]
    touched = torch.stack([s[:R] >= 0 for s in slots]).any(0)

    def run(window):
        t, m, v = tab0.clone().cuda(), m0.clone().cuda(), v0.clone().cuda()
        w, mw, vw = w0.clone().cuda(), mw0.clone().cuda(), vw0.clone().cuda()
        opt = AdamTF1(lr=1e-3, device="cuda")
        if window:
            segs = [dict(kind=_lib.RSX_ADAM_TABLE_TF1_COLD, d=D, n=R, var=t, m=m, v=v, slot=slots[0], slot_w=slots[1:]),
                    dict(kind=_lib.RSX_ADAM_VEC_COLD, n=R, var=w, m=mw, v=vw, slot=slots[0], slot_w=slots[1:])]
            for sl in opt.cold_slices(segs, [1.0, 2.0]):
                opt.run_slice(sl)
        else:
            # the union of the window's maps as ONE map: k single sweeps that skip every row any step touches
            union = torch.where(touched.cuda(), torch.zeros(R, dtype=torch.int32, device="cuda"),
                                torch.full((R,), -1, dtype=torch.int32, device="cuda"))
            union = torch.cat([union, torch.full((4,), -1, dtype=torch.int32, device="cuda")])
            segs = [dict(kind=_lib.RSX_ADAM_TABLE_TF1_COLD, d=D, n=R, var=t, m=m, v=v, slot=union),
                    dict(kind=_lib.RSX_ADAM_VEC_COLD, n=R, var=w, m=mw, v=vw, slot=union)]
            for _ in range(k):
                for sl in opt.cold_slices(segs, [1.0]):
                    opt.run_slice(sl)
                st = opt.state.cpu().numpy().copy()          # COLD slices never advance the powers: do it as the step would
                st[0], st[1] = np.float32(st[0] * np.float32(0.9)), np.float32(st[1] * np.float32(0.999))
                opt.state.copy_(torch.from_numpy(st))
        torch.cuda.synchronize()
        return [x.cpu() for x in (t, m, v, w, mw, vw)]

    a, b = run(True), run(False)
    for name, x, y in zip(("tables", "m", "v", "w1", "m_w", "v_w"), a, b):
        assert torch.equal(x, y), (name, float((x - y).abs().max()))
    # touched rows untouched by the sweep; untouched rows moved; and the numbers are the numpy restatement's
    touched = touched.cpu()
    assert torch.equal(a[0][touched], tab0[touched]) and torch.equal(a[3][touched], w0[touched])
    cold = ~touched
    var, m, v = tab0.numpy()[cold.numpy()], m0.numpy()[cold.numpy()], v0.numpy()[cold.numpy()]
    wv, wm, wvv = w0.numpy()[cold.numpy()], mw0.numpy()[cold.numpy()], vw0.numpy()[cold.numpy()]
    for t_ in range(1, k + 1):
        var, m, v = _adam_np(var, m, v, t_)
        wv, wm, wvv = _adam_np(wv, wm, wvv, t_, dense=True)
    np.testing.assert_allclose(a[0].numpy()[cold.numpy()], var, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(a[3].numpy()[cold.numpy()], wv, rtol=1e-6, atol=1e-9)


def _bits(x):
    return x.contiguous().view(torch.int32)


@pytest.mark.parametrize("k,hyper", [(8, {}), (4, {}), (2, {}), (8, dict(lr=0.05)), (8, dict(eps=1e-7, b2=0.99)),
                                     (8, dict(b1=0.5)), (5, dict(lr=3e-7)), (8, dict(eps=1e-12))])
def test_window_sweep_with_adversarial_state_is_the_bits_of_k_single_sweeps(k, hyper):
    """The window sweep applies its updates with packed square roots / divisions that are correctly rounded on a restricted
    domain (csrc/adam_fast.h) and sends every wave holding an element outside it to the IEEE form.  State built from the
    values that sit on and around the guard's edges -- zeros of both signs, denormals (where the moments of a row end up
    ~900 steps after its last gradient), tiny and huge normals, inf, NaN -- in every combination of (var, m, v), compared BIT
    FOR BIT (signs of zero and NaN payloads included) with k single-step sweeps, which always use the IEEE form.  The
    hyper-parameter sets include ones that switch the packed form off altogether (b1 = 0.5, tiny lr, tiny eps)."""
    from recsys_amd import _lib
    from recsys_amd.ops import AdamTF1
    f = np.float32
    m_vals = [0.0, -0.0, 1e-45, -1e-45, 4e-45, -6e-42, 1e-39, -1.2e-38, 2.0 ** -100, -(2.0 ** -90), 2.0 ** -81, 2.0 ** -80,
              -(2.0 ** -79), 1e-20, -3e-12, 1e-7, -2.5e-3, 0.3, -7.0, 2.0 ** 29, 2.0 ** 30, -(2.0 ** 31), np.inf, -np.inf, np.nan]
    v_vals = [0.0, -0.0, 1e-45, 1e-40, 2.0 ** -126, 2.0 ** -100, 2.0 ** -96, 2.0 ** -87, 2.0 ** -86, 2.0 ** -85, 1e-20, 1e-15,
              3e-9, 1e-4, 0.5, 100.0, 2.0 ** 39, 2.0 ** 40, 2.0 ** 41, 1e30, np.inf, np.nan]
    var_vals = [0.0, -0.0, 1e-45, -1e-40, 2.0 ** -60, -(2.0 ** -25), 2.0 ** -24, -(2.0 ** -23), 1e-5, -0.013, 0.5, -3.0, 1e6,
                np.inf, np.nan]
    with np.errstate(over="ignore"):
        mv, vv, xv = np.array(m_vals, dtype=f), np.array(v_vals, dtype=f), np.array(var_vals, dtype=f)
    rng = np.random.default_rng(k)
    # (1) the full grid, one combination per element; (2) rows of ordinary state with a single odd element in them;
    # (3) ordinary rows (these waves must take the packed form)
    grid = np.stack(np.meshgrid(xv, mv, vv, indexing="ij"), -1).reshape(-1, 3)
    D = 16
    n_grid_rows = (len(grid) + D - 1) // D
    pad = np.zeros((n_grid_rows * D - len(grid), 3), dtype=f)
    grid = np.concatenate([grid, pad]).reshape(n_grid_rows, D, 3)
    n_ord = 1500
    ordn = np.stack([rng.standard_normal((n_ord, D)).astype(f) * f(0.05),
                     rng.standard_normal((n_ord, D)).astype(f) * f(1e-4) * (f(0.9) ** rng.integers(0, 700, (n_ord, D))).astype(f),
                     (rng.random((n_ord, D)).astype(f) * f(1e-6)) ** 2], -1)
    odd = ordn[:600].copy()
    pick = rng.integers(0, len(grid.reshape(-1, 3)), 600)
    odd[np.arange(600), rng.integers(0, D, 600)] = grid.reshape(-1, 3)[pick]
    state = np.concatenate([grid, odd, ordn])
    R = state.shape[0]
    tab0, m0, v0 = (torch.from_numpy(np.ascontiguousarray(state[..., i])) for i in range(3))
    slot_all = torch.full((k, R + 4), -1, dtype=torch.int32)
    g = torch.Generator(device="cpu").manual_seed(k)
    for i in range(k):
        idx = torch.randperm(R, generator=g)[:R // 50]
        slot_all[i, idx] = torch.arange(R // 50, dtype=torch.int32)
    slot_all = slot_all.cuda()
    slots = [slot_all[i] for i in range(k)]
    touched = torch.stack([s[:R] >= 0 for s in slots]).any(0)
    b1, b2 = hyper.get("b1", 0.9), hyper.get("b2", 0.999)

    def run(window):
        t, m, v = tab0.clone().cuda(), m0.clone().cuda(), v0.clone().cuda()
        opt = AdamTF1(lr=hyper.get("lr", 1e-3), beta1=b1, beta2=b2, eps=hyper.get("eps", 1e-8), device="cuda")
        if window:
            segs = [dict(kind=_lib.RSX_ADAM_TABLE_TF1_COLD, d=D, n=R, var=t, m=m, v=v, slot=slots[0], slot_w=slots[1:])]
            for sl in opt.cold_slices(segs, [1.0, 2.0]):
                opt.run_slice(sl)
        else:
            union = torch.where(touched.cuda(), torch.zeros(R, dtype=torch.int32, device="cuda"),
                                torch.full((R,), -1, dtype=torch.int32, device="cuda"))
            union = torch.cat([union, torch.full((4,), -1, dtype=torch.int32, device="cuda")])
            segs = [dict(kind=_lib.RSX_ADAM_TABLE_TF1_COLD, d=D, n=R, var=t, m=m, v=v, slot=union)]
            for _ in range(k):
                for sl in opt.cold_slices(segs, [1.0]):
                    opt.run_slice(sl)
                st = opt.state.cpu().numpy().copy()
                st[0], st[1] = f(st[0] * f(b1)), f(st[1] * f(b2))
                opt.state.copy_(torch.from_numpy(st))
        torch.cuda.synchronize()
        return [x.cpu() for x in (t, m, v)]

    a, b = run(True), run(False)
    for name, x, y in zip(("tables", "m", "v"), a, b):
        diff = _bits(x) != _bits(y)
        assert not bool(diff.any()), (name, int(diff.sum()), state[diff.any(1).numpy()][:3])
    # the numbers are numpy's (IEEE, denormals on) for the ordinary rows no step touches
    cold = (~touched[-n_ord:]).cpu().numpy()
    var, m, v = (state[-n_ord:][cold][..., i] for i in range(3))
    lr, eps = hyper.get("lr", 1e-3), hyper.get("eps", 1e-8)
    for t_ in range(1, k + 1):
        var, m, v = _adam_np(var, m, v, t_, lr=lr, b1=b1, b2=b2, eps=eps)
    assert np.array_equal(a[0].numpy()[-n_ord:][cold].view(np.int32), var.view(np.int32))
    assert np.array_equal(a[1].numpy()[-n_ord:][cold].view(np.int32), m.view(np.int32))


def test_multi_sort_equals_single_sorts():
    from oracle import criteo
    from recsys_amd.ops import EmbeddingArena
    from tests.parity_util import synth_ids
    row_off = criteo.row_offsets()
    rng = np.random.default_rng(11)
    for B in (256, 1000, 2048):
        a = EmbeddingArena(row_off, 16, 2048, "cuda")
        b = EmbeddingArena(row_off, 16, 2048, "cuda")
        for rep in range(2):                                  # the second window also clears the first one's slot maps
            nj = 8 if B == 256 else 4
            ids = [torch.from_numpy(synth_ids(rng, B, row_off)).cuda() for _ in range(nj)]
            a.sort_window(ids)
            for i in range(nj):
                b.select(i)
                b.field_sort(ids[i])
            b.select(0)
            torch.cuda.synchronize()
            for i in range(nj):
                for key in ("perm", "seg_off", "uniq_row", "nuniq", "slot"):
                    x, y = a.sortbufs[i][key], b.sortbufs[i][key]
                    if key in ("perm", "uniq_row"):           # entries beyond the batch / the unique count are workspace
                        n = a.sortbufs[i]["nuniq"].cpu().numpy()
                        for f in range(a.F):
                            lim = B if key == "perm" else int(n[f])
                            assert torch.equal(x[f * a.stride:f * a.stride + lim], y[f * a.stride:f * a.stride + lim]), (B, i, key, f)
                    elif key == "seg_off":
                        n = a.sortbufs[i]["nuniq"].cpu().numpy()
                        for f in range(a.F):
                            o = f * (a.stride + 1)
                            assert torch.equal(x[o:o + int(n[f]) + 1], y[o:o + int(n[f]) + 1]), (B, i, key, f)
                    else:
                        assert torch.equal(x, y), (B, i, key)


def _deepfm_est(window, graph=True, seed=9, overlap=True, B=256):
    from recsys_amd import deepfm
    from recsys_amd.estimator import Estimator, RunConfig
    from recsys_amd.feature_columns import build_feature_columns
    lin, emb = build_feature_columns(16, "indicator_all")
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
              "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": B, "overlap_adam": overlap, "adam_window": window}
    return Estimator(deepfm.model_fn, None, params, RunConfig(use_hip_graph=graph, adam_mode="tf1_dense", device="cuda", seed=seed,
                                                               log_step_count_steps=10))


def _state(est):
    a = est.store.embeddings["input_layer"]
    torch.cuda.synchronize()
    return {"tables": a.tables.clone(), "m": a.m_t.clone(), "v": a.v_t.clone(), "w1": a.w1.clone(), "m_w": a.m_w.clone(),
            "v_w": a.v_w.clone(), "dense": est.store.dense.flat.clone(), "dense_m": est.store.dense.m.clone(),
            "opt": est.store.opt.state[:4].clone()}      # (beta powers, step counter; words 8.. hold a window's step sizes)


@pytest.mark.parametrize("window,steps,spg", [(8, 131, 16), (8, 75, 8), (7, 60, 8), (4, 131, 16), (3, 64, 8), (2, 37, 8)])
def test_windowed_resident_training_is_bit_identical_to_single_steps(window, steps, spg):
    """train_resident with optimizer windows of `window` steps inside its HIP graphs (head / tail graphs give shorter
    windows too) against the plain path: stand-alone sort, segment-sum, ONE full TF-1 sweep per step, eager."""
    from recsys_amd import synthetic
    from recsys_amd.estimator import PackedBatch
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    layout = CriteoLayout.from_columns(build_feature_columns(16, "indicator_all")[1])
    host = synthetic.criteo_id_batches(layout, 16, 256, seed=321)
    res = []
    for win in (window, 0):
        est = _deepfm_est(max(win, 1), graph=win > 0, overlap=win > 0)
        feats = [PackedBatch({"ids": i}, y, device="cuda") for i, y, _ in host]
        with torch.no_grad():
            est._call_model_fn(feats[0].views()[0], None, "infer")
        if win:
            assert est._window_len() == window
            est.train_resident(feats, steps, spg)
        else:
            for s_ in range(steps):
                est._train_step(feats[s_ % 16])
        assert est.global_step == steps
        res.append(_state(est))
    for name in res[0]:
        assert torch.isfinite(res[0][name].float()).all(), name
        assert torch.equal(res[0][name], res[1][name]), (name, float((res[0][name].float() - res[1][name].float()).abs().max()))


def test_windowed_streaming_train_is_bit_identical_and_respects_boundaries(tmp_path):
    """Estimator.train over an input_fn: windows of 4 steps (one graph replay each) cut at every log line, checkpoint and at
    the end of training, against RSX_ADAM_WINDOW=1 (every step on its own)."""
    from recsys_amd import synthetic
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    layout = CriteoLayout.from_columns(build_feature_columns(16, "indicator_all")[1])
    host = synthetic.criteo_id_batches(layout, 16, 256, seed=99)

    def input_fn():
        for s_ in range(1000):
            i, y, _ = host[s_ % 16]
            yield {"ids": i}, y

    res = []
    for win in ("4", "1"):
        os.environ["RSX_ADAM_WINDOW"] = win
        try:
            est = _deepfm_est(8)
            est.train(input_fn, steps=47)
            assert est.global_step == 47
            est.train(input_fn, max_steps=90)          # a second call: continues, stops exactly at max_steps
            assert est.global_step == 90
            if win == "4":
                assert any(k[0] == "packedwin" for k in est._graphs), "no window graph was captured"
        finally:
            os.environ.pop("RSX_ADAM_WINDOW", None)
        res.append(_state(est))
    for name in res[0]:
        assert torch.equal(res[0][name], res[1][name]), (name, float((res[0][name].float() - res[1][name].float()).abs().max()))


def _est_for(kind, B):
    from recsys_amd import dcn, deepfm, fm, xdeepfm
    from recsys_amd.estimator import Estimator, RunConfig
    from recsys_amd.feature_columns import build_feature_columns
    linear = {"deepfm": "indicator_all", "fm": "indicator_all", "dcn": "numeric", "xdeepfm": "numeric+indicator"}[kind]
    lin, emb = build_feature_columns(16, linear)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
              "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": B,
              "cross_layers": {"dcn": 3, "xdeepfm": "32,16"}.get(kind)}
    mfn = {"deepfm": deepfm.model_fn, "fm": fm.model_fn, "dcn": dcn.model_fn, "xdeepfm": xdeepfm.model_fn}[kind]
    return Estimator(mfn, None, params, RunConfig(use_hip_graph=True, adam_mode="tf1_dense", device="cuda", seed=5))


def _model_state(est):
    torch.cuda.synchronize()
    out = {"dense": est.store.dense.flat.clone(), "dense_m": est.store.dense.m.clone(), "opt": est.store.opt.state[:4].clone()}
    for name, a in est.store.embeddings.items():
        for k in ("tables", "m_t", "v_t", "w1", "m_w", "v_w"):
            if getattr(a, k, None) is not None:
                out[name + "." + k] = getattr(a, k).clone()
    return out


@pytest.mark.parametrize("kind,world", [("deepfm", 2), ("deepfm", 4), ("fm", 2), ("dcn", 2), ("xdeepfm", 2),
                                        ("fm", 1), ("dcn", 1), ("xdeepfm", 1)])
def test_windowed_step_of_every_model_is_bit_identical_to_its_single_steps(kind, world):
    """Every model's fused step through optimizer windows against the same step run one by one (RSX_ADAM_WINDOW=1), on one
    GPU (world 1) and as the data-parallel step of an emulated world (ids all-gather -- ONE for the window's 8 local batches --,
    send block, replica-sum Adam): every variable bit-identical."""
    from recsys_amd import synthetic
    from tests.dp_harness import EmulatedDataParallel
    from recsys_amd.estimator import PackedBatch
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    B = 128
    layout = CriteoLayout.from_columns(build_feature_columns(16, "indicator_all")[1])
    host = synthetic.criteo_id_batches(layout, 16, B, seed=77)
    rng = np.random.default_rng(3)
    logx = [np.log(np.floor(np.exp(rng.normal(2, 1, (B, 13)))) + 1.0).astype(np.float32) for _ in host]
    res = []
    for win in ("8", "1"):
        os.environ["RSX_ADAM_WINDOW"] = win
        try:
            est = _est_for(kind, B)
            if world > 1:
                est.store.dp = est.dist = EmulatedDataParallel(world)
            feats = [PackedBatch({"ids": i, "cont_log": lx} if kind == "xdeepfm" else {"ids": i}, y, device="cuda")
                     for (i, y, _), lx in zip(host, logx)]
            with torch.no_grad():
                est._call_model_fn(feats[0].views()[0], None, "infer")
            assert est._window_len() == int(win), (est._window_len(), win)
            est.train_resident(feats, 58, 8)
            assert est.global_step == 58
        finally:
            os.environ.pop("RSX_ADAM_WINDOW", None)
        res.append(_model_state(est))
    for name in res[0]:
        assert torch.isfinite(res[0][name].float()).all(), name
        assert torch.equal(res[0][name], res[1][name]), (name, float((res[0][name].float() - res[1][name].float()).abs().max()))
