"""Deep & Cross Network on the Criteo 39-field pipeline -- MI355X-native mirror of `dcn/dcn.py`
(model_fn :117-190, build_feature_columns :49-99, flags :16-39).

x0 = the 39x16 = 624-wide embedding vector; `cross_layers` cross layers (csrc/cross.hip, all fused);
deep tower [dense(relu) -> BN -> dropout] x n WITHOUT a final 1-unit layer (:144-149);
logits = dense(concat[deep, x_L], 1) (:151-152).  No first-order term (linear columns are built but unused, :96,128).
"""
import ctypes as C
import os

import torch

from . import _lib
from . import layers as L
from .deepfm import define_flags as _deepfm_flags, dp_unique_wanted
from .deepfm import input_fn, run_main  # noqa: F401
from .estimator import EstimatorSpec, ModeKeys, get_variable_store
from .feature_columns import CriteoLayout, build_feature_columns
from .ops import make_scatter_riders, CrossFn, CrossLayers, EmbeddingArena, FusedTower, _stream, gather_fm


def build_variables(store, params, capacity):
    layout = CriteoLayout.from_columns(params["embedding_feature_columns"])
    D = params["embedding_size"]
    dim = layout.F * D
    nL = int(params["cross_layers"])
    if store.dp is not None:
        capacity *= store.dp.world
    arena = EmbeddingArena(layout.row_off, D, capacity, store.device, with_w1=False)
    with torch.no_grad():
        t = torch.empty(arena.R, D)
        L.trunc_normal_(t, 1.0 / D ** 0.5, store.gen)
        arena.tables.copy_(t)
    layers = list(map(int, params["deep_layers"].split(",")))
    shapes, init = {}, {}
    zeros = lambda t, g: t.zero_()
    ones = lambda t, g: t.fill_(1.0)
    # dcn/dcn.py:139-140: weight AND bias are glorot-normal over a 1-D [dim] shape (fan_in = fan_out = dim)
    shapes["cross.W"], shapes["cross.b"] = (nL, dim), (nL, dim)
    init["cross.W"] = lambda t, g: L.glorot_normal_(t, dim, dim, g)
    init["cross.b"] = lambda t, g: L.glorot_normal_(t, dim, dim, g)
    d = dim
    for i, n in enumerate(layers):
        shapes[f"dnn.W{i}"], shapes[f"dnn.b{i}"] = (d, n), (n,)
        init[f"dnn.W{i}"] = lambda t, g, fi=d, fo=n: L.glorot_uniform_(t, fi, fo, g)
        init[f"dnn.b{i}"] = zeros
        shapes[f"dnn.gamma{i}"], init[f"dnn.gamma{i}"] = (n,), ones
        shapes[f"dnn.beta{i}"], init[f"dnn.beta{i}"] = (n,), zeros
        d = n
    shapes["out.W"], shapes["out.b"] = (d + dim, 1), (1,)
    init["out.W"] = lambda t, g, fi=d + dim: L.glorot_uniform_(t, fi, 1, g)
    init["out.b"] = zeros
    store.build({"input_layer": arena}, shapes, init, params["learning_rate"])
    store.layout = layout
    store.cross = CrossLayers(dim, nL, capacity, store.device)
    store.tower = None
    want_hip = params.get("tower", "hip") == "hip"
    if want_hip and not FusedTower.supports(dim, layers):
        print("INFO:deep_layers=%s is outside the fused tower's envelope (widths multiple of 4, last <= 256): using the "
              "autograd tower (tower='torch')" % params["deep_layers"], flush=True)
        want_hip = False
    if want_hip:
        store.tower = FusedTower(store.dense, "dnn", dim, layers, capacity, store.device)
        want_ux = store.dp is not None and params.get("dp_send_block", True) and dp_unique_wanted(store, params) and \
            EmbeddingArena.unique_exchange_ok(layout.row_off, store.dp.world)
        # (unique-list exchange: the window's sorts are the ranks' LOCAL ones -- dcn.py at 8 x 4 096 keeps its windows)
        sort_cap = capacity // store.dp.world if want_ux else capacity
        if store.adam_mode == "tf1_dense" and bool(params.get("overlap_adam", True)) and sort_cap <= 16384:
            store.window_k = _lib.default_adam_window(capacity, want_ux)          # optimizer windows (include/rsx.h rsx_adam_window)
            store.window_dp = True
        store.graph_safe_dp = True      # the fused step issues its collectives outside autograd
        store.dp_block = False
        store.dp_unique = False
        if store.dp is not None and params.get("dp_send_block", True):      # zero-copy gradient exchange (see deepfm.py)
            if want_ux:         # round 5: the ranks exchange unique (row, sum) lists; send block [dense | G [capT, D]]
                ux = arena.enable_unique_exchange(store.dp.world, capacity // store.dp.world)
                store.dp.make_send_block(store.dense, ux.capT, [D])
                store.dp_unique = True
            else:               # RSX_DP_EXCHANGE=examples: [dense | dX of the local batch]
                store.dp.make_send_block(store.dense, capacity // store.dp.world, [dim])
            store.dp_block = True
        # share of the untouched-row Adam sweep carried by [fwd_0.., head, bwd_{L-1}..bwd_0, scatter]
        env = os.environ.get("RSX_SWEEP_WEIGHTS")
        store.sweep_weights = params.get("sweep_weights") or ([float(x) for x in env.split(",")] if env else
                                                              [0.0] * len(layers) + [1.0] + [3.0] * len(layers) + [1.0])


def _train_fused(store, arena, ids, labels, params, masks):
    """TRAIN step, explicit kernel sequence (same structure as recsys_amd/deepfm.py::_train_fused): gather -> cross fwd ->
    tower fwd (first launch carries the dedup sort) / head / bwd (carrying slices of the untouched-row Adam sweep) ->
    cross bwd -> [train_op:] scatter + touched-row Adam + dense Adam in one launch."""
    dp, P = store.dp, store.dense
    nh = store.tower.widths[-1]
    nl = len(store.tower.widths)
    oW, oG = P["out.W"].detach().view(-1), P["out.W"].grad.view(-1)
    with torch.no_grad():
        overlap = store.adam_mode == "tf1_dense" and bool(params.get("overlap_adam", True))
        wk, wpos, wfeat = store.window_of_step()
        ux = dp is not None and store.dp_unique
        ids_sort = dp.all_gather_id_list([ids], prefetchable=True)[0] if (dp is not None and wk == 1 and not ux) else ids   # first: see deepfm._train_fused
        zc = dp is not None and store.dp_block and not ux
        # Round 4: the lookup and the cross layers' forward in ONE launch (rsx_gather_cross_fwd: the gather's lanes already hold the
        # example's row in the cross kernel's layout); RSX_GATHER_CROSS=0: two launches
        gcross = store.cross.fused_gather_ok(arena)
        if gcross:
            x0, _, cz = store.cross.gather_forward(arena, ids, P["cross.W"], P["cross.b"], oW[nh:])
        else:
            x0, _, _, _ = arena.gather(ids)
        job, sweeps, hot, last_sweep = None, None, None, None
        # optimizer window (deepfm.py, include/rsx.h rsx_adam_window): position 0 sorts the ids of all wk batches and sweeps the
        # untouched rows ONCE for the whole window (a launch of its own); the other positions run neither
        if wk > 1 and not overlap:
            raise _lib.RsxError("optimizer windows need the split TF-1 update (adam_mode=tf1_dense, overlap_adam)")
        arena.select(wpos)
        if ux:
            # ids phase of the unique-list exchange (deepfm._train_fused): local sorts -> key blocks -> one all-gather -> merge
            if wpos == 0:
                idl = [f["ids"] for f in wfeat] if wk > 1 else [ids]
                arena.ux_merge(dp.all_gather_keys(arena.ux_sort_pack(idl), arena, idl), wk)
            arena.select(wpos)
            arena.last_B = arena.ux.max_unique
            if wk > 1:
                if wpos == 0:
                    cold, _ = arena.adam_split_segments(window_k=wk)
                    store.opt.window_sweep(cold)
                hot = ()
        elif wk > 1:
            if wpos == 0:
                from .dist import window_global_ids
                arena.sort_window(window_global_ids(dp, wfeat))    # data-parallel: one all-gather for all wk batches' ids
                cold, _ = arena.adam_split_segments(window_k=wk)
                store.opt.window_sweep(cold)
            arena.last_B = ids.shape[0] * (dp.world if dp is not None else 1)
            hot = ()
        elif ids_sort.shape[0] <= 2048:              # the sort rides in the first tower-forward launch; larger ones run stand-alone
            # (a 256-thread carrier workgroup sorts 4096 keys in 55 us, the 1024-thread kernel in 26 us)
            job = arena.sort_job(ids_sort)
        else:
            arena.field_sort(ids_sort)
        if overlap and wk == 1:
            cold, hot = arena.adam_split_segments()
            sweeps = store.opt.cold_slices(cold, store.sweep_weights)
            last_sweep = sweeps[-1] if len(sweeps) == 2 * nl + 2 else None
            sweeps = sweeps[:2 * nl + 1]
            if job is not None:
                assert sweeps[0] is None, "the first forward launch carries the sort: no sweep slice may ride with it"
        # Round 4 (single replica, fused optimizer launch): the tower's dW partial-tile reductions and the cross layers' gradient
        # reduce -- two launches whose results only the optimizer reads -- ride in the scatter's stage-A launch as extra
        # workgroups (rsx_segsum_partials_ride).  Data parallel: the dense gradients go into a collective first, so they stay.
        ride = dp is None and hot is not None and _lib.form("scatter_riders") == "1"
        if not gcross:
            _, _, cz = store.cross.forward(x0, P["cross.W"], P["cross.b"], wout=oW[nh:])
        # Round 6: the cross layers' backward needs only the head's gradient -- it rides in the second tower layer's backward launch
        # and the first layer's launch accumulates onto the dX it wrote.  Data parallel too (dX lives in the send block either
        # way); there the reduce of its gradient partials runs as a launch of its own right behind the tower, because the dense
        # gradients go through a collective before the scatter that would otherwise carry it.
        # (a launch that carries a slice of the optimizer sweep -- the single-step schedule -- keeps its own riders only)
        xride = hot is not None and sweeps is None and store.cross.cross_ride_ok(store.tower, ids.shape[0])
        cr = store.cross.rider_args(x0, P["cross.W"], P["cross.b"], P["cross.W"].grad, P["cross.b"].grad, oW[nh:], oG[nh:]) \
            if xride else None
        loss, prob, dX, gz, _ = store.tower.train_step(
            x0, labels.reshape(-1).to(torch.float32), params["dropout"], store.opt.state.view(torch.int32)[3:4],
            s0=cz, head=((oW[:nh], oG[:nh]), "out.b", None, None), relu0=False, relu2=False,
            replicas=dp.world if dp is not None else 1, masks=masks,
            seed=0x5eed + (7919 * dp.rank if dp is not None else 0),     # replicas draw independent dropout patterns
            sort_job=job, sweeps=sweeps, sort_in_fwd=True, outs=(dp.send_views(ids.shape[0])[0], None, None) if zc else None,
            defer_dw_reduce=ride, cross_rider=cr)
        if xride:
            cross_job = store.tower.cross_job_pending
            if not ride:
                _lib.check(_lib.lib().rsx_cross_reduce_run(C.byref(cross_job), _stream()), "rsx_cross_reduce_run")
                cross_job = None
        else:
            cross_job = store.cross.backward(x0, P["cross.W"], P["cross.b"], P["cross.W"].grad, P["cross.b"].grad, dX, True,
                                             gz=gz, wout=oW[nh:], dwout=oG[nh:], defer_reduce=ride)
        riders = make_scatter_riders(store.tower.dw_jobs_pending, cross_job) if ride else None
        if ux:      # the rank's own sorted segment-sum, written as its block of the send buffer (deepfm._train_fused)
            (Gv,) = dp.send_views(arena.ux.capT)
            arena.ux_segsum_local(dX.shape[0], None, dX, None, None, Gv, None, wpos)

    def train_op():
        with torch.no_grad():
            dXg, blocks, Bg, dense_segs = dX, None, dX.shape[0], None
            if ux:                  # ONE collective [dense | G], then the touched-row Adam off the merged lists
                (G0,), blocks, dense_segs = dp.gather_send_block(arena.ux.capT, fold_dense=True)
                arena.select(wpos)
                arena.ux_merged_adam(G0, None, blocks[1], store.opt, dense_segs or store.dense.adam_segments(), last_sweep,
                                     window=(wk, wpos))
                return
            if zc:                  # ONE collective straight from the send block (dense arena + dX)
                (dXg,), blocks, dense_segs = dp.gather_send_block(dX.shape[0], fold_dense=hot is not None)
                Bg = dX.shape[0] * dp.world
            elif dp is not None:    # ONE collective: per-example gradient block + dense arena (summed in rank order);
                # the scatter then reads every rank's block in place from the gathered buffer
                dXg, _, _, _, blocks = dp.gather_example_grads(dX, dense=store.dense.grad, blocked=True)
                Bg = dX.shape[0] * dp.world
            if hot is not None:
                arena.select(wpos)
                arena.segsum_adam(Bg, None, dXg, None, None, store.opt, dense_segs or store.dense.adam_segments(),
                                  last_sweep, blocks=blocks, window=(wk, wpos), riders=riders)
            else:
                arena.segsum(Bg, None, dXg, None, None, blocks=blocks)
                store.apply_gradients()

    return EstimatorSpec(ModeKeys.TRAIN, predictions={"prob": prob}, loss=loss[0], train_op=train_op)


def model_fn(features, labels, mode, params):
    """dcn/dcn.py:117-190."""
    store = get_variable_store()
    ids = features["ids"]
    if not store.built:
        build_variables(store, params, capacity=max(int(params.get("max_batch_size", 0)), ids.shape[0]))
    arena, P = store.embeddings["input_layer"], store.dense
    training = mode == ModeKeys.TRAIN
    layers = params["deep_layers"].split(",")
    masks = params.get("_dropout_masks")
    if training and store.tower is not None:
        return _train_fused(store, arena, ids, labels, params, masks)
    if not training and store.tower is not None and params.get("fused_infer", True) and not torch.is_grad_enabled() \
            and ids.shape[0] <= store.tower.cap:
        # EVAL / PREDICT through the TRAIN step's kernels (gather, fused cross layers, FusedTower.infer: see deepfm.py)
        nh = store.tower.widths[-1]
        oW = P["out.W"].detach().view(-1)
        x0, _, _, _ = arena.gather(ids)
        _, _, cz = store.cross.forward(x0, P["cross.W"], P["cross.b"], wout=oW[nh:])
        lab = None if (labels is None or mode == ModeKeys.PREDICT) else labels.reshape(-1).to(torch.float32)
        prob, loss = store.tower.infer(x0, store.opt.state.view(torch.int32)[3:4], lab, s0=cz,
                                       head=((oW[:nh], None), "out.b", None, None), relu0=False, relu2=False)
        predictions = {"prob": prob}
        if mode == ModeKeys.PREDICT:
            return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
        return EstimatorSpec(mode, predictions=predictions, loss=loss[0], eval_metric_ops={"AUC": None, "Accuracy": None})
    if training:
        store.sort_ids_for_backward(arena, ids)
    (x0,) = gather_fm(arena, ids, dp=store.dp if training else None)               # embedding_net (:123)
    xl = CrossFn.apply(x0, P["cross.W"], P["cross.b"], store.cross)                # 'cross_layers' (:132-142)
    dnn_net = L.tower(x0, P, "dnn", len(layers), training, params["dropout"], masks)   # 'deep_layers' (:144-149)
    logits = L.dense(torch.cat([dnn_net, xl], -1), P["out.W"], P["out.b"]).reshape(-1)   # (:151-152)
    pred = torch.sigmoid(logits)
    predictions = {"prob": pred}
    if mode == ModeKeys.PREDICT:
        return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
    loss = L.sigmoid_ce_mean(logits, labels)
    if mode == ModeKeys.EVAL:
        return EstimatorSpec(mode, predictions=predictions, loss=loss, eval_metric_ops={"AUC": None, "Accuracy": None})
    return EstimatorSpec(mode, predictions=predictions, loss=loss, train_op=lambda: store.minimize(loss))


def define_flags():
    p = _deepfm_flags()
    p.add_argument("--cross_layers", type=int, default=4)      # dcn/dcn.py:24
    p.set_defaults(save_checkpoints_steps=2000)
    return p


def make_params(FLAGS):
    lin, emb = build_feature_columns(FLAGS.embedding_size, "numeric")
    return {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": FLAGS.embedding_size,
            "learning_rate": FLAGS.learning_rate, "dropout": FLAGS.dropout, "deep_layers": FLAGS.deep_layers,
            "cross_layers": FLAGS.cross_layers, "max_batch_size": FLAGS.batch_size, **({"adam_window": FLAGS.adam_window} if getattr(FLAGS, "adam_window", 0) else {})}


def main(argv=None):
    FLAGS = define_flags().parse_args(argv)
    FLAGS._argv = argv
    return run_main(model_fn, FLAGS, make_params)


if __name__ == "__main__":
    main()
