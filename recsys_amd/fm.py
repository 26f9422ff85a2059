"""Factorization Machine on the Criteo 39-field pipeline -- MI355X-native mirror of `fm/fm.py`
(model_fn :115-170, build_feature_columns :47-97, flags :16-37; BASELINE config 1, the plumbing model).

logits = dense([relu(first_order), fm_second_order], 1).  The whole model is the fused embedding kernels
(gather + first-order + FM forward, sorted segment-sum backward) plus a 2->1 dense head.
"""
import ctypes as C
import os

import torch

from . import _lib
from . import layers as L
from .deepfm import build_variables as _build
from .deepfm import define_flags, dp_unique_wanted, input_fn, make_params, run_main  # noqa: F401
from .estimator import EstimatorSpec, ModeKeys, get_variable_store
from .ops import EmbeddingArena, _ptr, _stream, gather_fm


def model_fn(features, labels, mode, params):
    """fm/fm.py:115-170."""
    store = get_variable_store()
    ids = features["ids"]
    if not store.built:
        cap = max(int(params.get("max_batch_size", 0)), ids.shape[0])
        _build(store, params, capacity=cap, with_dnn=False)
        a0 = store.embeddings["input_layer"]
        want_ux = store.dp is not None and params.get("fused", True) and dp_unique_wanted(store, params) and \
            EmbeddingArena.unique_exchange_ok(a0.row_off_np, store.dp.world)
        if store.adam_mode == "tf1_dense" and params.get("fused", True):
            gcap = cap * (store.dp.world if store.dp is not None else 1)
            if (cap if want_ux else gcap) <= 16384:      # (unique-list exchange: the window's sorts are the ranks' local ones)
                store.window_k = _lib.default_adam_window(gcap, want_ux)     # optimizer windows (include/rsx.h rsx_adam_window)
                store.window_dp = True
        store.dp_block = False
        store.dp_unique = False
        if store.dp is not None and params.get("fused", True):
            # data-parallel: the dense gradient arena + the rank's block of the sparse exchange in one persistent send block --
            # [G [capT, D] | gw1 [capT]] (unique-row lists, round 5: deepfm.py) or, RSX_DP_EXCHANGE=examples, [S | gy2 | gy1] of
            # the local batch
            if want_ux:
                ux = a0.enable_unique_exchange(store.dp.world, cap)
                store.dp.make_send_block(store.dense, ux.capT, [a0.D, 1])
                store.dp_unique = True
            else:
                store.dp.make_send_block(store.dense, cap, [a0.D, 1, 1])
            store.dp_block = True
            store.graph_safe_dp = True       # the fused step issues its collectives outside autograd
    arena, P = store.embeddings["input_layer"], store.dense
    training = mode == ModeKeys.TRAIN
    if training and store.adam_mode == "tf1_dense" and params.get("fused", True) and (store.dp is None or store.dp_block):
        return _train_fused(store, arena, ids, labels)
    if not training and store.adam_mode == "tf1_dense" and params.get("fused", True) and params.get("fused_infer", True) \
            and not torch.is_grad_enabled():
        # EVAL / PREDICT: gather + the FM head kernel of the TRAIN step (its gradients go to scratch)
        B = ids.shape[0]
        dev = ids.device
        _, _, y1p, y2 = arena.gather(ids, fm=True, first_order=True)
        lab = torch.zeros(B, device=dev) if (labels is None or mode == ModeKeys.PREDICT) else labels.reshape(-1).to(torch.float32)
        prob, loss = torch.empty(B, device=dev), torch.empty(1, device=dev)
        scr = torch.empty(2 * B + 4, device=dev)
        _lib.check(_lib.lib().rsx_fm_head(_ptr(y1p), _ptr(y2), _ptr(P["b1"]), _ptr(P["out.W"].detach().view(-1)), _ptr(P["out.b"]),
                                          _ptr(lab), _ptr(prob), _ptr(scr[:B]), _ptr(scr[B:2 * B]), _ptr(scr[2 * B:2 * B + 2]),
                                          _ptr(scr[2 * B + 2:2 * B + 3]), _ptr(scr[2 * B + 3:]), _ptr(loss), 1.0 / B, B, None,
                                          _stream()), "rsx_fm_head")
        predictions = {"prob": prob}
        if mode == ModeKeys.PREDICT:
            return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
        return EstimatorSpec(mode, predictions=predictions, loss=loss[0], eval_metric_ops={"AUC": None, "Accuracy": None})
    if training:
        store.sort_ids_for_backward(arena, ids)
    _, y1p, y2 = gather_fm(arena, ids, fm=True, first_order=True, dp=store.dp if training else None)
    y_1d = torch.relu(y1p + P["b1"])                                               # 'linear_net' (:120-121)
    logits = L.dense(torch.stack([y_1d, y2], -1), P["out.W"], P["out.b"])          # (:131-132)  [B,1]
    pred = torch.sigmoid(logits)
    predictions = {"prob": pred}
    if mode == ModeKeys.PREDICT:
        return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
    loss = L.sigmoid_ce_mean(logits, labels)
    if mode == ModeKeys.EVAL:
        return EstimatorSpec(mode, predictions=predictions, loss=loss, eval_metric_ops={"AUC": None, "Accuracy": None})
    return EstimatorSpec(mode, predictions=predictions, loss=loss, train_op=lambda: store.minimize(loss))


def _train_fused(store, arena, ids, labels):
    """TRAIN step as 4 launches, no autograd: dedup sort -> gather (+ first order + FM) -> FM head with its backward,
    carrying most of the untouched-row Adam sweep as extra workgroups -> segment-sum fused with the touched-row and dense
    Adam (+ the rest of the sweep).  Same exact split of the TF-1 update as deepfm.py."""
    P, dp = store.dense, store.dp
    B = ids.shape[0]
    dev = ids.device
    with torch.no_grad():
        # data-parallel: the optimizer sees the GLOBAL batch -- dedup sort over the all-gathered ids, per-example gradient
        # block [S | gy2 | gy1] + dense arena exchanged by ONE all-gather straight from the send block (see deepfm.py)
        # optimizer window (deepfm.py, include/rsx.h rsx_adam_window): position 0 sorts the ids of all wk batches and sweeps the
        # untouched rows ONCE for the whole window (a launch of its own); the other positions run neither
        wk, wpos, wfeat = store.window_of_step()
        arena.select(wpos)
        sweep = sweep2 = None
        ux = dp is not None and store.dp_unique
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
                    store.opt.window_sweep(cold[::-1])
            else:
                cold, hot = arena.adam_split_segments()
                sweep, sweep2 = store.opt.cold_slices(cold[::-1], [0.7, 0.3])
        elif wk > 1:
            if wpos == 0:
                from .dist import window_global_ids
                arena.sort_window(window_global_ids(dp, wfeat))    # data-parallel: one all-gather for all wk batches' ids
                cold, _ = arena.adam_split_segments(window_k=wk)
                store.opt.window_sweep(cold[::-1])
            arena.last_B = B * (dp.world if dp is not None else 1)
        else:
            arena.field_sort(dp.all_gather_id_list([ids], prefetchable=True)[0] if dp is not None else ids)
            cold, hot = arena.adam_split_segments()
            # 70 % of the untouched-row sweep rides in the head launch, 30 % (table blocks only: the first-order vector goes
            # first) in the scatter + touched-row Adam launch (measured: 76.4 -> 72.2 us per step)
            sweep, sweep2 = store.opt.cold_slices(cold[::-1], [0.7, 0.3])
        Sv, gy2v, gy1v = dp.send_views(B) if (dp is not None and not ux) else (None,) * 3
        prob = torch.empty(B, device=dev)
        gy1, gy2 = (gy1v, gy2v) if (dp is not None and not ux) else (torch.empty(B, device=dev) for _ in range(2))
        oW, oG = P["out.W"].detach().view(-1), P["out.W"].grad.view(-1)
        world = dp.world if dp is not None else 1
        lab = labels.reshape(-1).to(torch.float32)
        # round 4: forward AND head in ONE launch (rsx_gather_fm_head: 3.25 -> 2.25 launches per step).  Every example leaves its
        # contribution to the four dense gradients as a row of `terms` in the dense arena's layout; the optimizer launch adds
        # the rows in example order (an RSX_ADAM_DENSE segment with B = batch "replicas"), the loss is read from the terms when
        # somebody asks (estimator.LazyMeanLoss).  Single replica, windows on (no sweep slice to carry); RSX_FM_FUSE=0: two launches.
        terms_on = dp is None and _lib.form("fm_fuse") != "0" and P.n + 1 <= 64
        fuse = terms_on and sweep is None
        terms, loss = None, None
        if terms_on:
            stride = (P.n + 1 + 3) & ~3
            terms = torch.empty((B + 15) // 16, stride, device=dev)      # rows of 16 examples, pre-added in example order
            from .estimator import LazyMeanLoss
            loss = LazyMeanLoss(terms, P.n, B)
        if fuse:
            S = torch.empty(B, arena.D, device=dev)
            _lib.check(_lib.lib().rsx_gather_fm_head(
                _ptr(arena.tables), _ptr(arena.w1), _ptr(arena.row_off), _ptr(ids), _ptr(S), arena.w1_mask, _ptr(P["b1"]), _ptr(oW),
                _ptr(P["out.b"]), _ptr(lab), _ptr(prob), _ptr(gy1), _ptr(gy2), _ptr(terms), stride, P.n, P.offsets["b1"],
                P.offsets["out.W"], P.offsets["out.b"], 1.0 / (B * world), B, arena.F, arena.D, _stream()), "rsx_gather_fm_head")
        else:
            # two launches: a sweep slice rides in the head launch (every step on its own), or data parallel.  Single replica:
            # the head leaves the same per-example terms as the fused launch, so both schedules train the same bits.
            E, S, y1p, y2 = arena.gather(ids, fm=True, first_order=True, S_out=Sv)
            if terms is None:
                loss = torch.empty(1, device=dev)
            _lib.check(_lib.lib().rsx_fm_head_terms(
                _ptr(y1p), _ptr(y2), _ptr(P["b1"]), _ptr(oW), _ptr(P["out.b"]), _ptr(lab), _ptr(prob), _ptr(gy1), _ptr(gy2),
                _ptr(oG), _ptr(P["out.b"].grad), _ptr(P["b1"].grad), _ptr(loss) if terms is None else None, _ptr(terms),
                stride if terms is not None else 0, P.n, P.offsets["b1"], P.offsets["out.W"], P.offsets["out.b"],
                1.0 / (B * world), B, None if sweep is None else C.byref(sweep), _stream()), "rsx_fm_head_terms")

        if ux:      # the rank's own sorted segment-sum, written as its block of the send buffer (deepfm._train_fused)
            Gv, gw1v = dp.send_views(arena.ux.capT)
            arena.ux_segsum_local(B, S, None, gy1, gy2, Gv, gw1v, wpos)

    def train_op():
        with torch.no_grad():
            if ux:
                (G0, gw10), blocks, dense_segs = dp.gather_send_block(arena.ux.capT, fold_dense=True)
                arena.select(wpos)
                arena.ux_merged_adam(G0, gw10, blocks[1], store.opt, dense_segs or store.dense.adam_segments(), sweep2,
                                     window=(wk, wpos))
            elif dp is not None:
                (Sg, gy2g, gy1g), blocks, dense_segs = dp.gather_send_block(B, fold_dense=True)
                arena.select(wpos)
                arena.segsum_adam(B * world, Sg, None, gy1g, gy2g, store.opt, dense_segs or store.dense.adam_segments(), sweep2,
                                  blocks=blocks, window=(wk, wpos))
            else:
                arena.select(wpos)
                dense_segs = store.dense.adam_segments()
                if terms is not None:       # the head's dense gradients: the examples' rows, summed in example order by the launch
                    dense_segs = [dict(kind=_lib.RSX_ADAM_DENSE, n=P.n, var=P.flat, m=P.m, v=P.v, g=terms, B=terms.shape[0],
                                       stride=terms.shape[1], zero_grad=0)]
                arena.segsum_adam(B, S, None, gy1, gy2, store.opt, dense_segs, sweep2, window=(wk, wpos))

    return EstimatorSpec(ModeKeys.TRAIN, predictions={"prob": prob}, loss=loss if terms is not None else loss[0],
                         train_op=train_op)


def main(argv=None):
    FLAGS = define_flags().parse_args(argv)
    FLAGS._argv = argv
    return run_main(model_fn, FLAGS, make_params)


if __name__ == "__main__":
    main()
