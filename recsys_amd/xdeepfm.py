"""xDeepFM on the Criteo 39-field pipeline -- MI355X-native mirror of `xdeepfm/xdeepfm.py`
(model_fn :123-233, build_feature_columns :44-94, flags :12-34; BASELINE config 3: CIN [128,128]).

logits = dense([linear_y, cin_y, dnn_y], 1) (:194-195) with
  linear_y = relu(dense([13 log-values | 26 one-hot blocks], 1))                    (:127,131)
  cin_y    = relu(dense(sum_d concat_k X^k, 1)), X^{k+1} = CIN layer(X^0, X^k)      (:135-182, csrc/cin.hip)
  dnn_y    = relu(dense(tower(E_dnn), 1)) over a SECOND, independent embedding set  (:185-192)
"""
import os

import torch

from . import layers as L
from .deepfm import define_flags as _deepfm_flags
from .deepfm import input_fn, run_main  # noqa: F401
from .estimator import EstimatorSpec, ModeKeys, get_variable_store
from .feature_columns import CriteoLayout, build_feature_columns
from . import _lib
from .ops import CinLayerFn, CinNet, EmbeddingArena, FusedTower, _ptr, _stream


def build_variables(store, params, capacity):
    layout = CriteoLayout.from_columns(params["embedding_feature_columns"])
    D = params["embedding_size"]
    F = layout.F
    cin = list(map(int, params["cross_layers"].split(",")))
    # the fused TRAIN step is the only xDeepFM path: check its envelope before any variable exists, so an unsupported
    # configuration fails here with the reason instead of at the first step
    _layers = list(map(int, params["deep_layers"].split(",")))
    if D not in (4, 8, 16, 32, 64):
        raise _lib.RsxError("xdeepfm: embedding_size must be 4, 8, 16, 32 or 64 (rows are moved as float4 lanes); got %d" % D)
    # The hand-written CIN kernels cover embedding_size 16 and layer widths <= 128 (BASELINE's [128,128], the script's default
    # 20,10,10), the fused tower widths that are multiples of 4.  Every other value the reference's flags accept
    # (xdeepfm/xdeepfm.py:12-19) trains through the GENERIC path: the same gather / scatter / Adam kernels under autograd, the
    # CIN layer and the tower as library GEMMs (`_cin_layer_generic`, layers.py) -- slower, same arithmetic.
    store.generic = bool(params.get("force_generic", False)) or D != 16 or max(cin) > 128 or \
        not FusedTower.supports(F * D, _layers)
    layers = list(map(int, params["deep_layers"].split(",")))
    if store.dp is not None:
        capacity *= store.dp.world
    cat_keys = {c.key for c in params["linear_feature_columns"] if c.kind == "hash_indicator"}
    a1 = EmbeddingArena(layout.row_off, D, capacity, store.device, with_w1=True, w1_field_mask=layout.field_mask(cat_keys))
    a2 = EmbeddingArena(layout.row_off, D, capacity, store.device, with_w1=False)
    a2.share_sort_of(a1)                       # same ids, same layout: one dedup sort per step serves both table sets
    n_lin = 13 + sum(c.rows for c in params["linear_feature_columns"] if c.kind == "hash_indicator")
    with torch.no_grad():
        for a in (a1, a2):
            t = torch.empty(a.R, D)
            L.trunc_normal_(t, 1.0 / D ** 0.5, store.gen)
            a.tables.copy_(t)
        w = torch.empty(a1.R)
        L.glorot_uniform_(w, n_lin, 1, store.gen)
        a1.w1.copy_(w)
    shapes, init = {}, {}
    zeros = lambda t, g: t.zero_()
    ones = lambda t, g: t.fill_(1.0)
    shapes["lin.wnum"], init["lin.wnum"] = (13,), lambda t, g: L.glorot_uniform_(t, n_lin, 1, g)
    shapes["lin.b"], init["lin.b"] = (1,), zeros
    H = F
    for k, n in enumerate(cin):
        # tf.get_variable without initializer -> glorot-uniform over [1, F*H, N] (A-7)
        shapes[f"cin.W{k}"], init[f"cin.W{k}"] = (F * H, n), lambda t, g, fi=F * H, fo=n: L.glorot_uniform_(t, fi, fo, g)
        shapes[f"cin.c{k}"], init[f"cin.c{k}"] = (n,), zeros
        H = n
    tot = sum(cin)
    shapes["cin.Wout"], init["cin.Wout"] = (tot, 1), lambda t, g: L.glorot_uniform_(t, tot, 1, g)
    shapes["cin.bout"], init["cin.bout"] = (1,), zeros
    d = F * D
    for i, n in enumerate(layers):
        shapes[f"dnn.W{i}"], shapes[f"dnn.b{i}"] = (d, n), (n,)
        init[f"dnn.W{i}"] = lambda t, g, fi=d, fo=n: L.glorot_uniform_(t, fi, fo, g)
        init[f"dnn.b{i}"] = zeros
        shapes[f"dnn.gamma{i}"], init[f"dnn.gamma{i}"] = (n,), ones
        shapes[f"dnn.beta{i}"], init[f"dnn.beta{i}"] = (n,), zeros
        d = n
    shapes["dnn.Wout"], shapes["dnn.bout"] = (d, 1), (1,)
    init["dnn.Wout"] = lambda t, g, fi=d: L.glorot_uniform_(t, fi, 1, g)
    init["dnn.bout"] = zeros
    shapes["out.W"], shapes["out.b"] = (3, 1), (1,)
    init["out.W"] = lambda t, g: L.glorot_uniform_(t, 3, 1, g)
    init["out.b"] = zeros
    store.build({"input_layer": a1, "input_layer_1": a2}, shapes, init, params["learning_rate"])
    store.layout = layout
    store.cin_sizes = cin
    store.dp_block = False
    if store.generic:
        store.tower = store.cin = None
        return
    store.tower = FusedTower(store.dense, "dnn", F * D, layers, capacity, store.device)
    # cin_bf16: the CIN contraction on the bf16 MFMA path (csrc/cin_bf16.hip).  Off by default: fp32 is the parity path
    # cin_split = ns (1..3): the contraction on the bf16 matrix cores with ns bf16 planes per operand (csrc/cin_split.hip);
    # ns = 3 keeps every product to 2^-23 -- fp32-grade, held to the same 1e-5 parity tests as the fp32 MFMA kernels
    # cin_split = 4: forward / data gradients with two scaled fp16 planes per operand (half the MFMAs of ns = 3, products to
    # 2^-22: held to the tolerances of ns = 3 in tests/test_gpu_cin_split.py), weight gradients on three bf16 planes.
    # Default (cin_split unset, no cin_bf16): 4 -- the fastest path that holds the parity bar (RSX_CIN_SPLIT_DEFAULT overrides:
    # A/B runs); cin_split = 0 selects the fp32 MFMA kernels of csrc/cin.hip.
    bf16 = bool(params.get("cin_bf16", False))
    split = params.get("cin_split")
    split = (0 if bf16 else default_cin_split()) if split is None else int(split)
    if split and not (F <= 40 and D == 16 and max(cin) <= 128 and len(cin) <= 4):
        split = 0                                    # outside the kernels' envelope: the fp32 MFMA path
    store.cin = CinNet(F, D, cin, capacity, store.device, bf16=bf16 and not split, split=split)
    from .deepfm import dp_unique_wanted
    want_ux = store.dp is not None and dp_unique_wanted(store, params) and \
        EmbeddingArena.unique_exchange_ok(layout.row_off, store.dp.world)
    sort_cap = capacity // store.dp.world if want_ux else capacity      # (unique-list exchange: the ranks sort their own batches)
    if store.adam_mode == "tf1_dense" and bool(params.get("overlap_adam", True)) and sort_cap <= 16384 and \
            (store.dp is None or params.get("dp_send_block", True)):
        store.window_k = _lib.default_adam_window(capacity, want_ux)          # optimizer windows (include/rsx.h rsx_adam_window)
        store.window_dp = True
    store.dp_block = False
    store.dp_unique = False
    if store.dp is not None and params.get("dp_send_block", True):          # zero-copy gradient exchange (see deepfm.py)
        if want_ux:
            # round 5: the ranks exchange unique (row, sum) lists of BOTH table sets (one dedup serves both):
            # send block [dense | G1 [capT, D] | G2 [capT, D] | g_lin sums [capT]]
            ux = a1.enable_unique_exchange(store.dp.world, capacity // store.dp.world)
            a2.ux = ux
            store.dp.make_send_block(store.dense, ux.capT, [D, D, 1])
            store.dp_unique = True
        else:
            store.dp.make_send_block(store.dense, capacity // store.dp.world, [F * D, F * D, 1])
        store.dp_block = True
    store.graph_safe_dp = True      # the fused step issues its collectives outside autograd


def _cin_layer_generic(X0, Xk, W, c):
    """One CIN layer for ANY embedding size / width (xdeepfm/xdeepfm.py:145-172) as library ops under autograd:
    Z[b,d,f*H+h] = X0[b,f,d] * Xk[b,h,d] ((f major, h minor)), out[b,n,d] = relu(Z[b,d,:] . W[:,n] + c[n])."""
    B, F, D = X0.shape
    H = Xk.shape[1]
    Z = torch.einsum("bfd,bhd->bdfh", X0, Xk).reshape(B, D, F * H)
    return torch.relu(Z @ W + c).transpose(1, 2)


def _cin(X0, P, sizes, sweeps=None, generic=False):
    """'cin_net' (:135-182): every layer's full map is both the next hidden state and a direct output.
    sweeps[k]: slice of the untouched-row optimizer sweep carried by layer k's weight-gradient launch."""
    outs, Xk = [], X0
    for k in range(len(sizes)):
        if generic:
            Xk = _cin_layer_generic(X0, Xk, P[f"cin.W{k}"], P[f"cin.c{k}"])
        else:
            Xk = CinLayerFn.apply(X0, Xk, P[f"cin.W{k}"], P[f"cin.c{k}"], None if sweeps is None else sweeps[k])
        outs.append(Xk)
    res = torch.cat(outs, 1).sum(-1)                                        # (:180-181)
    return L.dense(res, P["cin.Wout"], P["cin.bout"], relu=True)            # cin_y [B,1] (:182)


def _train_fused(store, a1, a2, ids, logx, labels, params, masks):
    dp, P = store.dp, store.dense
    B = ids.shape[0]
    sweeps, hot, L = None, None, len(store.cin_sizes)
    tower_sweeps, last_sweep = None, None
    ux = dp is not None and getattr(store, "dp_unique", False)      # exchange of per-rank unique-row lists (deepfm._train_fused)
    zc = dp is not None and getattr(store, "dp_block", False) and not ux
    with torch.no_grad():
        # data-parallel: the optimizer sees the GLOBAL batch -- the dedup sort runs over the all-gathered ids (issued first:
        # they depend on nothing of the step), so the same exact split of the TF-1 update applies as on one GPU
        # optimizer window (deepfm.py, include/rsx.h rsx_adam_window): position 0 sorts the ids of all wk batches and sweeps the
        # untouched rows of BOTH table sets once for the whole window (a launch of its own); the other positions run neither
        wk, wpos, wfeat = store.window_of_step()
        ids_sort = dp.all_gather_id_list([ids], prefetchable=True)[0] if (zc and wk == 1) else ids
        job, ride = None, False
        split = store.adam_mode == "tf1_dense" and bool(params.get("overlap_adam", True))
        if wk > 1 and not (split and (dp is None or zc or ux)):
            raise _lib.RsxError("optimizer windows need the split TF-1 update (and, data-parallel, the send block)")
        if ux and wpos == 0:
            # ids phase of the unique-list exchange: local sorts -> key blocks -> one all-gather -> global lists / slot maps / src
            idl = [f["ids"] for f in wfeat] if wk > 1 else [ids]
            a1.ux_merge(dp.all_gather_keys(a1.ux_sort_pack(idl), a1, idl), wk)
        a1.select(wpos)
        a2.select(wpos)
        if ux:
            a1.last_B = a2.last_B = a1.ux.max_unique
        if wk > 1:
            if wpos == 0:
                if not ux:
                    from .dist import window_global_ids
                    a1.sort_window(window_global_ids(dp, wfeat))       # data-parallel: one all-gather for all wk batches' ids
                c1, _ = a1.adam_split_segments(window_k=wk)
                c2, _ = a2.adam_split_segments(window_k=wk)
                store.opt.window_sweep(c1[::-1] + c2)
            if not ux:
                a1.last_B = a2.last_B = B * (dp.world if dp is not None else 1)
            hot = ()
        elif dp is None or zc or ux:
            if not ux:
                a2.last_B = ids_sort.shape[0]
            # The sort (it serves a2 as well, share_sort_of) rides in the tower's first forward launch when no sweep slice is
            # scheduled before or in that launch (slices read the sort's slot map); otherwise it runs first.
            ride = (not ux and ids_sort.shape[0] <= int(_lib.form("sort_ride_max"))
                    and _lib.form("xdfm_sort_ride") == "1")
            job = a1.sort_job(ids_sort) if not ux else None
            if store.adam_mode == "tf1_dense" and bool(params.get("overlap_adam", True)):
                # exact split of the TF-1 update (see deepfm.py): the sweep over the UNtouched rows of both table sets
                # (700 MB of streaming) rides in the CIN forward and weight-gradient launches (MFMA work, little HBM); the touched rows
                # and the dense variables follow the scatter in one small launch
                c1, h1 = a1.adam_split_segments()
                c2, h2 = a2.adam_split_segments()
                # carriers, in launch order: [CIN fwd_0..fwd_{L-1} | tower fwd_0, fwd_1, head, bwd_1, bwd_0 | CIN backward:
                # bf16 dx_{L-1}..dx_0 then ONE launch with every layer's dW (L + 1 launches); fp32 bwd_{L-1}..bwd_0 | scatter].
                # Default shares: the fp32 CIN launches are long MFMA kernels and take the sweep by their flops; on the
                # bf16 path every backward launch carries a share (weights measured on MI355X; RSX_XDFM_SWEEP_WEIGHTS
                # overrides).
                w = [float(store.cin_sizes[k]) * (a1.F if k == 0 else store.cin_sizes[k - 1]) for k in range(len(store.cin_sizes))]
                tw = sum(w)
                nl = len(store.tower.widths)
                env = os.environ.get("RSX_XDFM_SWEEP_WEIGHTS")
                nd = L + 1 if (store.cin.bf16 or store.cin.split) else L
                if env:
                    wts = [float(x) for x in env.split(",")]
                elif store.cin.split:      # (the split-operand CIN launches carry nothing)
                    wts = [0.0] * L + [0.0] * nl + [1.0] + [2.0] * nl + [0.0] * (L + 1) + [2.0]
                elif store.cin.bf16:
                    wts = [0.0] * L + [0.0] * nl + [1.0] + [2.0] * nl + [0.0] * L + [2.0] + [2.0]
                else:
                    wts = [0.5 * x / tw for x in w] + [0.0] * (2 * nl + 1) + [0.5 * x / tw for x in w][::-1] + [0.0]
                assert len(wts) == L + 2 * nl + 1 + nd + 1, "RSX_XDFM_SWEEP_WEIGHTS: %d weights expected" % (L + 2 * nl + 2 + nd)
                # (first-order vector first: the LAST slice, carried by the scatter launch, may hold table blocks only)
                sl_all = store.opt.cold_slices(c1[::-1] + c2, wts)
                bw = sl_all[L + 2 * nl + 1:L + 2 * nl + 1 + nd]
                # CinNet.forward: fwd_k; CinNet.backward: layer order (bf16: dx_0..dx_{L-1}, then the dW launch)
                sweeps = sl_all[:L] + (bw[:L][::-1] + bw[L:] if (store.cin.bf16 or store.cin.split) else bw[::-1])
                tower_sweeps = sl_all[L:L + 2 * nl + 1]
                last_sweep = sl_all[-1]
                ride = ride and all(x is None for x in sl_all[:L + 1])    # no slice before the launch that carries the sort
                hot = h1 + h2
            if not ride:
                job = None
                if not ux:
                    a1.field_sort(ids_sort)
        dX1v, dX2v, glv = dp.send_views(B) if zc else (None,) * 3          # per-example gradient block, written in place
        # both input_layer calls (:125,185) + the pre-activation of linear_net (one-hot weights + 13 numeric log-values, :127)
        E1 = torch.empty(B, a1.F * a1.D, device=ids.device)
        E2 = torch.empty(B, a1.F * a1.D, device=ids.device)
        lin_pre = torch.empty(B, device=ids.device)
        logx_g = logx.contiguous()
        gjob = None
        if store.cin.gather_ride_ok():    # the lookup rides in the CIN's filter-preparation launch (one launch less on the chain)
            gjob = _lib.GatherTwoJob(a1.tables.data_ptr(), a1.w1.data_ptr(), a2.tables.data_ptr(), a1.row_off.data_ptr(), ids.data_ptr(),
                                     logx_g.data_ptr(), P["lin.wnum"].data_ptr(), E1.data_ptr(), E2.data_ptr(), lin_pre.data_ptr(),
                                     a1.w1_mask, B, a1.F, a1.D, logx.shape[1])
        else:
            _lib.check(_lib.lib().rsx_gather_two_fwd(_ptr(a1.tables), _ptr(a1.w1), _ptr(a2.tables), _ptr(a1.row_off), _ptr(ids),
                                                     _ptr(logx_g), _ptr(P["lin.wnum"]), _ptr(E1), _ptr(E2), _ptr(lin_pre), a1.w1_mask,
                                                     B, a1.F, a1.D, logx.shape[1], _stream()), "rsx_gather_two_fwd")
        X0 = E1.view(B, a1.F, a1.D)
        cin_y = store.cin.forward(X0, P, None if sweeps is None else sweeps[:L], gather_job=gjob)              # 'cin_net' (:135-182)
        loss, prob, dX2, g_lin, g_cin = store.tower.train_step(
            E2, labels.reshape(-1).to(torch.float32), params["dropout"], store.opt.state.view(torch.int32)[3:4],
            s0=lin_pre, c0="lin.b", s1=cin_y, replicas=dp.world if dp is not None else 1, masks=masks,
            seed=0x5eed + (7919 * dp.rank if dp is not None else 0),       # replicas draw independent dropout patterns
            sort_job=job, sort_in_fwd=True, sweeps=tower_sweeps, outs=(dX2v, glv, None) if zc else None)
        logx_c = logx.contiguous()
        dX1 = store.cin.backward(X0, P, g_cin.reshape(-1), None if sweeps is None else sweeps[L:],
                                 dX0_out=dX1v.view(B, a1.F, a1.D) if zc else None,
                                 lin=(logx_c, g_lin, P["lin.wnum"].grad)).view(B, -1)   # cin.* and lin.wnum grads land in the dense arena
        if ux:
            # the rank's own sorted segment-sums of both table sets, written as its block of the send buffer
            G1v, G2v, gw1v = dp.send_views(a1.ux.capT)
            a1.ux_segsum_local(B, None, dX1, g_lin, None, G1v, gw1v, wpos)
            a2.ux_segsum_local(B, None, dX2, None, None, G2v, None, wpos)

    def train_op():
        with torch.no_grad():
            if ux:
                # ONE collective [dense | G1 | G2 | g_lin sums] (the 3.3 MB of CIN filters through an all-reduce beside it), then
                # both table sets' touched-row Adam off the merged lists + the dense update in one launch
                (G10, G20, gw10), blocks, dense_segs = dp.gather_send_block(a1.ux.capT, fold_dense=True)
                a1.select(wpos)
                a2.select(wpos)
                a1.ux_merged_adam(G10, gw10, blocks[1], store.opt, dense_segs or store.dense.adam_segments(), last_sweep,
                                  second=(a2, G20), window=(wk, wpos))
            elif zc:
                # ONE collective straight from the send block [dense arena | dX1 | dX2 | g_lin]; both table sets' scatter +
                # touched-row Adam + the dense update (replica arenas summed in rank order) in one launch
                (dX1g, dX2g, glg), blocks, dense_segs = dp.gather_send_block(B, fold_dense=hot is not None)
                Bg = B * dp.world
                if hot is not None:
                    a1.select(wpos)
                    a2.select(wpos)
                    a1.segsum_adam(Bg, None, dX1g, glg, None, store.opt, dense_segs or store.dense.adam_segments(), last_sweep,
                                   blocks=blocks, second=(a2, dX2g), window=(wk, wpos))
                else:
                    a1.segsum(Bg, None, dX1g, glg, None, blocks=blocks)
                    a2.segsum(Bg, None, dX2g, None, None, blocks=blocks)
                    store.apply_gradients()
            elif dp is not None:
                both = torch.cat([dX1, dX2], 1)          # both table sets' gradients + ids in ONE all-gather
                dg, _, g1, _, idsg = dp.gather_example_grads(both, None, g_lin, None, ids=ids)
                w = dX1.shape[1]
                a1.field_sort(idsg)
                a2.field_sort(idsg)
                a1.segsum(dg.shape[0], None, dg[:, :w].contiguous(), g1, None)
                a2.segsum(dg.shape[0], None, dg[:, w:].contiguous(), None, None)
                dp.all_reduce_sum(store.dense.grad)
                store.apply_gradients()
            elif hot is not None:
                # scatter + touched-row Adam of BOTH table sets (one shared sort) in one launch, which also carries the
                # dense variables and advances the beta powers
                a1.select(wpos)
                a2.select(wpos)
                a1.segsum_adam(B, None, dX1, g_lin, None, store.opt, store.dense.adam_segments(), last_sweep, second=(a2, dX2),
                               window=(wk, wpos))
            else:
                a1.segsum(B, None, dX1, g_lin, None)
                a2.segsum(B, None, dX2, None, None)
                store.apply_gradients()

    return EstimatorSpec(ModeKeys.TRAIN, predictions={"prob": prob}, loss=loss[0], train_op=train_op)


def model_fn(features, labels, mode, params):
    """xdeepfm/xdeepfm.py:123-233.  features: 'ids' int32 [B,39], 'cont_log' float32 [B,13] (log(x+shift) of _c1.._c13)."""
    store = get_variable_store()
    ids, logx = features["ids"], features["cont_log"]
    if not store.built:
        build_variables(store, params, capacity=max(int(params.get("max_batch_size", 0)), ids.shape[0]))
    a1, a2, P = store.embeddings["input_layer"], store.embeddings["input_layer_1"], store.dense
    masks = params.get("_dropout_masks")
    generic = getattr(store, "generic", False)
    if mode == ModeKeys.TRAIN and not generic:
        return _train_fused(store, a1, a2, ids, logx, labels, params, masks)
    B = ids.shape[0]
    if not generic and params.get("fused_infer", True) and not torch.is_grad_enabled() and B <= store.tower.cap:
        # EVAL / PREDICT through the TRAIN step's kernels: both gathers + linear_net pre-activation, CIN forward (fp32 or bf16
        # as configured), FusedTower.infer (see deepfm.py)
        E1 = torch.empty(B, a1.F * a1.D, device=ids.device)
        E2 = torch.empty(B, a1.F * a1.D, device=ids.device)
        lin_pre = torch.empty(B, device=ids.device)
        _lib.check(_lib.lib().rsx_gather_two_fwd(_ptr(a1.tables), _ptr(a1.w1), _ptr(a2.tables), _ptr(a1.row_off), _ptr(ids),
                                                 _ptr(logx.contiguous()), _ptr(P["lin.wnum"]), _ptr(E1), _ptr(E2), _ptr(lin_pre),
                                                 a1.w1_mask, B, a1.F, a1.D, logx.shape[1], _stream()), "rsx_gather_two_fwd")
        cin_y = store.cin.forward(E1.view(B, a1.F, a1.D), P, None)
        lab = None if (labels is None or mode == ModeKeys.PREDICT) else labels.reshape(-1).to(torch.float32)
        prob, loss = store.tower.infer(E2, store.opt.state.view(torch.int32)[3:4], lab, s0=lin_pre, c0="lin.b", s1=cin_y)
        predictions = {"prob": prob}
        if mode == ModeKeys.PREDICT:
            return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
        return EstimatorSpec(mode, predictions=predictions, loss=loss[0], eval_metric_ops={"AUC": None, "Accuracy": None})
    n_layers = len(params["deep_layers"].split(","))
    training = mode == ModeKeys.TRAIN
    if training:                 # generic TRAIN: the gathers are autograd nodes whose backward is the sorted segment-sum
        from .ops import gather_fm
        ids_s = ids if store.dp is None else store.dp.all_gather_rows(ids)     # the optimizer sees the global batch
        a1.field_sort(ids_s)                                     # one dedup sort serves both table sets
        a2.field_sort(ids_s)                                     # (a2 aliases a1's sort outputs: bookkeeping only)
        E1, y1cat = gather_fm(a1, ids, fm=False, first_order=True, dp=store.dp)
        (E2,) = gather_fm(a2, ids, fm=False, first_order=False, dp=store.dp)
    else:
        E1, _, y1cat, _ = a1.gather(ids, first_order=True)
        E2, _, _, _ = a2.gather(ids)
    linear_y = torch.relu(torch.addmv(y1cat, logx, P["lin.wnum"]) + P["lin.b"])                 # (:131)
    cin_y = _cin(E1.view(B, a1.F, a1.D), P, store.cin_sizes, generic=generic)
    dnn_net = L.tower(E2, P, "dnn", n_layers, training, params["dropout"], masks)
    dnn_y = L.dense(dnn_net, P["dnn.Wout"], P["dnn.bout"], relu=True)                          # (:192)
    logits = L.dense(torch.cat([linear_y[:, None], cin_y, dnn_y], -1), P["out.W"], P["out.b"])  # (:194-195) [B,1]
    pred = torch.sigmoid(logits)
    predictions = {"prob": pred}
    if mode == ModeKeys.PREDICT:
        return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
    loss = L.sigmoid_ce_mean(logits, labels)
    if training:
        def train_op():                                            # AdamOptimizer.minimize (:225-226)
            store.minimize(loss)
        return EstimatorSpec(mode, predictions=predictions, loss=loss, train_op=train_op)
    return EstimatorSpec(mode, predictions=predictions, loss=loss, eval_metric_ops={"AUC": None, "Accuracy": None})


def default_cin_split():
    """xdeepfm.py's CIN mode when neither --cin_split nor --cin_bf16 is given (RSX_CIN_SPLIT_DEFAULT overrides: A/B runs)."""
    return int(os.environ.get("RSX_CIN_SPLIT_DEFAULT", "4"))


def define_flags():
    p = _deepfm_flags()
    p.add_argument("--cross_layers", default="20,10,10")       # xdeepfm/xdeepfm.py:19 (BASELINE config 3 uses 128,128)
    p.add_argument("--cin_bf16", type=lambda s: s.lower() in ("1", "true", "yes"), default=False,
                   help="CIN contraction on bf16 MFMA (fp32 accumulate); not the reference-parity path")
    p.add_argument("--cin_split", type=int, default=None,
                   help="CIN contraction on the 16-bit MFMA with split operands: 3 = three bf16 planes per operand (products exact "
                        "to 2^-23), 4 = two scaled fp16 planes forward / data gradients + three bf16 planes weight gradients "
                        "(2^-22-grade); 0 = the fp32 MFMA kernels; unset = the default (see build_variables)")
    p.add_argument("--eval_steps", type=int, default=200)
    p.set_defaults(num_epochs=5, eval_parts=10, log_steps=50, save_checkpoints_steps=2000)
    return p


def make_params(FLAGS):
    lin, emb = build_feature_columns(FLAGS.embedding_size, "numeric+indicator")
    return {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": FLAGS.embedding_size,
            "learning_rate": FLAGS.learning_rate, "dropout": FLAGS.dropout, "deep_layers": FLAGS.deep_layers,
            "cross_layers": FLAGS.cross_layers, "max_batch_size": FLAGS.batch_size, "cin_bf16": FLAGS.cin_bf16, "cin_split": FLAGS.cin_split, **({"adam_window": FLAGS.adam_window} if getattr(FLAGS, "adam_window", 0) else {})}


def main(argv=None):
    FLAGS = define_flags().parse_args(argv)
    FLAGS._argv = argv
    return run_main(model_fn, FLAGS, make_params)


if __name__ == "__main__":
    main()
