"""DeepFM on the Criteo 39-field pipeline -- MI355X-native mirror of the reference's
`deepfm/deepfm.py:model_fn` (:73-150) married to `fm/fm.py:build_feature_columns` (:47-97), i.e. the
README's "DeepFM on Criteo, d=16, DNN 100-100, bs=256" (BASELINE config 2).

Same surface as the script: FLAGS-compatible argparse names, `input_fn`, `model_fn(features, labels,
mode, params)`, `main`.  The embedding gather + first-order + FM term, the row-wise gradient scatter
and the TF-1 Adam sweep are librsx.so kernels; the 624-100-100-1 tower is rocBLAS via torch.
"""
import argparse
import os

import torch

from . import _lib
from . import layers as L
from .dist import dp_overlap_enabled as dist_overlap_enabled, overlap_ranges as dist_overlap_ranges
from .estimator import Estimator, EstimatorSpec, EvalSpec, ModeKeys, RunConfig, TrainSpec, get_variable_store, \
    train_and_evaluate
from .feature_columns import CAT_FEATURE, CONT_FEATURE, CriteoLayout, build_feature_columns
from .ops import EmbeddingArena, gather_fm


def _dense_specs(F, D, layers, n_inputs_out, with_dnn=True):
    """Variable shapes + initialisers (glorot-uniform kernels, zero biases, BN gamma 1 / beta 0: Appendix A-7)."""
    shapes, init = {}, {}
    zeros = lambda t, g: t.zero_()
    ones = lambda t, g: t.fill_(1.0)

    def add_dense(name_w, name_b, fi, fo):
        shapes[name_w], shapes[name_b] = (fi, fo), (fo,)
        init[name_w] = lambda t, g, fi=fi, fo=fo: L.glorot_uniform_(t, fi, fo, g)
        init[name_b] = zeros

    shapes["b1"], init["b1"] = (1,), zeros
    if with_dnn:
        d = F * D
        for i, n in enumerate(layers):
            add_dense(f"dnn.W{i}", f"dnn.b{i}", d, n)
            shapes[f"dnn.gamma{i}"], init[f"dnn.gamma{i}"] = (n,), ones
            shapes[f"dnn.beta{i}"], init[f"dnn.beta{i}"] = (n,), zeros
            d = n
        add_dense("dnn.Wout", "dnn.bout", d, 1)
    add_dense("out.W", "out.b", n_inputs_out, 1)
    return shapes, init


def build_variables(store, params, capacity, with_dnn=True):
    """Creates what TF creates lazily inside input_layer / tf.layers.* on the first model_fn call."""
    layout = CriteoLayout.from_columns(params["embedding_feature_columns"])
    D = params["embedding_size"]
    lin_keys = {c.key for c in params["linear_feature_columns"] if c.kind.endswith("indicator")}
    if store.dp is not None:
        capacity *= store.dp.world                              # the sort/scatter workspace holds the global batch
    arena = EmbeddingArena(layout.row_off, D, capacity, store.device, with_w1=True,
                           w1_field_mask=layout.field_mask(lin_keys))
    with torch.no_grad():
        t = torch.empty(arena.R, D)
        L.trunc_normal_(t, 1.0 / D ** 0.5, store.gen)          # embedding_column initializer (A-4)
        arena.tables.copy_(t)
        w = torch.empty(arena.R)
        L.glorot_uniform_(w, arena.R, 1, store.gen)             # tf.layers.dense kernel [R,1]
        arena.w1.copy_(w)
    layers = list(map(int, params["deep_layers"].split(","))) if with_dnn else []
    shapes, init = _dense_specs(layout.F, D, layers, 3 if with_dnn else 2, with_dnn)
    store.build({"input_layer": arena}, shapes, init, params["learning_rate"])
    store.layout = layout
    store.tower = None
    from .ops import FusedTower
    want_hip = with_dnn and params.get("tower", "hip") == "hip"
    if want_hip and not FusedTower.supports(layout.F * D, layers):
        print("INFO:deep_layers=%s is outside the fused tower's envelope (widths multiple of 4, last <= 256): using the "
              "autograd tower (tower='torch')" % params["deep_layers"], flush=True)
        want_hip = False
    if want_hip:
        store.tower = FusedTower(store.dense, "dnn", layout.F * D, layers, capacity, store.device)
        # optimizer windows (include/rsx.h rsx_adam_window): up to 8 consecutive steps share ONE sweep over the untouched rows
        # (capacity = the GLOBAL batch under data parallelism; the window's sorts are the ranks' LOCAL ones under the unique-list
        # exchange, so only the local batch has to fit the one-launch multi-sort)
        want_ux = store.dp is not None and params.get("dp_send_block", True) and dp_unique_wanted(store, params) and \
            EmbeddingArena.unique_exchange_ok(layout.row_off, store.dp.world)
        sort_cap = capacity // store.dp.world if want_ux else capacity
        if store.adam_mode == "tf1_dense" and bool(params.get("overlap_adam", True)) and sort_cap <= 16384:
            store.window_k = _lib.default_adam_window(capacity, want_ux)
            store.window_dp = True
        store.graph_safe_dp = True      # the fused step issues its collectives outside autograd
        store.dp_block = False
        store.dp_unique = False
        if store.dp is not None and params.get("dp_send_block", True):
            # zero-copy gradient exchange: the dense gradient arena and the rank's block of the sparse exchange live inside ONE
            # persistent send buffer (no pack launch before the all-gather).
            # Round 5 (default; RSX_DP_EXCHANGE=examples keeps the round-1..4 exchange of the pre-dedup per-example block): every
            # rank de-duplicates and sums ITS batch, the ranks exchange unique (row, sum) lists (EmbeddingArena.enable_unique_exchange,
            # csrc/uniq_exchange.hip) -- the send block is [dense | G [capT, D] | gw1 [capT]]
            b_local = capacity // store.dp.world
            if want_ux:
                ux = arena.enable_unique_exchange(store.dp.world, b_local)
                store.dp.make_send_block(store.dense, ux.capT, [D, 1])
                store.dp_unique = True
            else:
                store.dp.make_send_block(store.dense, b_local, [layout.F * D, D, 1, 1])
            store.dp_block = True
        # share of the untouched-row Adam sweep carried by [fwd_0.., head, bwd_{L-1}..bwd_0, scatter] (measured: the
        # latency-bound scatter + touched-row Adam launch hides a quarter of the sweep; r02 grid over the shares with the faster
        # head kernel: [0,0,2,3,3,2] 93.2 us, [0,0,1,3.5,3.5,2.5] 91.4 us per step)
        env = os.environ.get("RSX_SWEEP_WEIGHTS")
        store.sweep_weights = params.get("sweep_weights") or ([float(x) for x in env.split(",")] if env else
                                                              [0.0] * len(layers) + [1.0] + [3.5] * len(layers) + [2.5])


def dp_unique_wanted(store, params):
    """The data-parallel sparse exchange of this run: unique-row lists (round 5; the default from two ranks on) need the split
    TF-1 update (the optimizer launch that owns the touched rows); RSX_DP_EXCHANGE=examples: the pre-dedup per-example block of
    rounds 1-4 -- also the default at world 1 (RSX_FORCE_DIST), where there is nothing to merge and the rank-local dedup +
    segment-sum are pure overhead (through RCCL at world 1: deepfm.py 0.0790 ms against 0.0875)."""
    world = store.dp.world if store.dp is not None else 1
    default = "unique" if world > 1 else "examples"
    return os.environ.get("RSX_DP_EXCHANGE", default) == "unique" and store.adam_mode == "tf1_dense" and \
        bool(params.get("overlap_adam", True)) and params.get("dp_send_block", True)


def model_fn(features, labels, mode, params):
    """deepfm/deepfm.py:73-150.  features['ids']: int32 [B,39] table-local ids in slot order
    (recsys_amd.input_pipeline produces them; hashing/bucketizing is the input_layer's host half)."""
    store = get_variable_store()
    ids = features["ids"]
    if not store.built:
        build_variables(store, params, capacity=max(int(params.get("max_batch_size", 0)), ids.shape[0]))
    arena, P = store.embeddings["input_layer"], store.dense
    training = mode == ModeKeys.TRAIN
    n_layers = len(params["deep_layers"].split(","))
    masks = params.get("_dropout_masks")

    if training and getattr(store, "tower", None) is not None:
        return _train_fused(store, arena, ids, labels, params, masks)
    if not training and getattr(store, "tower", None) is not None and params.get("fused_infer", True) \
            and not torch.is_grad_enabled() and ids.shape[0] <= store.tower.cap:
        # EVAL / PREDICT through the TRAIN step's kernels (gather + 2 tower launches + head: 4 launches instead of ~70
        # framework ones; BN in inference form, dropout off -- FusedTower.infer)
        E, _, y1p, y2 = arena.gather(ids, fm=True, first_order=True)
        lab = None if (labels is None or mode == ModeKeys.PREDICT) else labels.reshape(-1).to(torch.float32)
        prob, loss = store.tower.infer(E, store.opt.state.view(torch.int32)[3:4], lab, s0=y1p, c0="b1", s1=y2)
        predictions = {"prob": prob}
        if mode == ModeKeys.PREDICT:
            return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
        return EstimatorSpec(mode, predictions=predictions, loss=loss[0], eval_metric_ops={"AUC": None, "Accuracy": None})
    if training:
        store.sort_ids_for_backward(arena, ids)                # dedup for the sparse gradient (ids only)
    E, y1p, y2 = gather_fm(arena, ids, fm=True, first_order=True, dp=store.dp if training else None)
    y_1d = torch.relu(y1p + P["b1"])                           # 'first-order' (:90-91)
    dnn_net = L.tower(E, P, "dnn", n_layers, training, params["dropout"], masks)    # 'dnn' (:100-107)
    y_dnn = L.dense(dnn_net, P["dnn.Wout"], P["dnn.bout"], relu=True)              # (:108)
    logits = torch.cat([y_1d[:, None], y2[:, None], y_dnn], -1)                     # (:110)
    logits = L.dense(logits, P["out.W"], P["out.b"]).reshape(-1)                    # (:111-112)
    pred = torch.sigmoid(logits)
    predictions = {"prob": pred}
    if mode == ModeKeys.PREDICT:
        return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
    loss = L.sigmoid_ce_mean(logits, labels)
    if mode == ModeKeys.EVAL:
        return EstimatorSpec(mode, predictions=predictions, loss=loss, eval_metric_ops={"AUC": None, "Accuracy": None})

    def train_op():                                            # AdamOptimizer.minimize (:142-143)
        store.minimize(loss)

    return EstimatorSpec(mode, predictions=predictions, loss=loss, train_op=train_op)


def _train_fused(store, arena, ids, labels, params, masks):
    """TRAIN step with no autograd, 7 launches: gather_fm -> tower forward x2 (the first carries the dedup sort) -> head +
    loss (+ a slice of the untouched-row Adam sweep) -> tower backward x2 (+ sweep slices) -> [train_op:] sorted
    segment-sum fused with the touched-row and dense-variable Adam.  Data-parallel: two all-gathers (ids; gradient block
    + dense arena) around the same launches."""
    dp = store.dp
    with torch.no_grad():
        # The ids-only dedup sort rides in the first tower-forward launch as extra workgroups.  (A side HIP stream was
        # measured instead: inside a graph the fork/join across HW queues costs ~10 us each way, more than it hides.)
        overlap = store.adam_mode == "tf1_dense" and bool(params.get("overlap_adam", True))
        # data-parallel: the optimizer sees the GLOBAL batch (TF concatenates the replicas' IndexedSlices), so the
        # dedup sort runs over the all-gathered ids -- a 40 KB collective issued FIRST (ids depend on nothing of this step),
        # so that every launch from the gather to the last backward layer is one graph segment
        # Optimizer window (estimator.Window, rsx_adam_window): this step is position wpos of wk consecutive steps whose
        # batches are known; position 0 sorts all of them and carries ONE sweep for the whole window, the others none.
        wk, wpos, wfeat = store.window_of_step()
        if wk > 1 and not overlap:
            raise _lib.RsxError("optimizer windows need the split TF-1 update (adam_mode=tf1_dense, overlap_adam)")
        ux = dp is not None and store.dp_unique      # the exchange of per-rank unique-row lists (round 5)
        ids_sort = dp.all_gather_id_list([ids], prefetchable=True)[0] if (dp is not None and wk == 1 and not ux) else ids
        zc = dp is not None and store.dp_block and not ux     # outputs of the per-example gradient block written in place
        dXv, Sv, gy2v, gy1v = dp.send_views(ids.shape[0]) if zc else (None,) * 4
        job = None
        if ux:
            # ids phase of the unique-list exchange: the rank's OWN dedup sorts (the window's wk batches in one launch) -> key
            # blocks -> ONE all-gather -> the global lists / slot maps / src of all wk positions (rsx_uniq_merge): 3 launches
            # and a collective per WINDOW, no global sort
            if wpos == 0:
                keys_g = dp.all_gather_keys(arena.ux_sort_pack([f["ids"] for f in wfeat] if wk > 1 else [ids]), arena,
                                            [f["ids"] for f in wfeat] if wk > 1 else [ids])
                arena.ux_merge(keys_g, wk)
            arena.select(wpos)
            arena.last_B = arena.ux.max_unique
            if wk > 1 and wpos == 0 and overlap:
                cold, hot = arena.adam_split_segments(window_k=wk)
                store.opt.window_sweep(cold[::-1])
        elif wk > 1:
            arena.select(wpos)
            if wpos == 0:
                from .dist import window_global_ids
                win_ids = window_global_ids(dp, wfeat)   # data-parallel: ONE all-gather for the ids of all wk local batches
                arena.sort_window(win_ids)
            arena.last_B = ids.shape[0] * (dp.world if dp is not None else 1)
        elif ids_sort.shape[0] <= int(_lib.form("sort_ride_max")):   # rides in another launch; larger sorts are faster with 1024 threads of their own
            arena.select(0)
            job = arena.sort_job(ids_sort)
        else:
            arena.select(0)
        # RSX_SORT_IN_GATHER=1: the sort rides in the GATHER launch (the step's first) instead, so that both tower-forward
        # launches may carry sweep slices too.  Measured (r02, MI355X): the same 93.4 us with the forward shares at 0, and
        # 106-108 us with any share given to the forward launches ([1,1,1,3,3,2.5] ...): a forward launch is a pure chain of
        # dependent L2 accesses and stretches by more than the slice it hides.  So the default stays the r01 form.
        in_gather = job is not None and overlap and _lib.form("sort_in_gather") == "1"
        # round 4: the gather itself rides in the first tower-forward launch (E tiles gathered straight into LDS as the MFMA
        # A operand, rsx_gather_tower_fwd0): one launch less on the step's dependent chain
        fuse_gather = not in_gather and store.tower.fused_gather_ok(arena, ids.shape[0])
        if fuse_gather:
            E, S, y1p, y2 = arena.gather_outputs(ids.shape[0], fm=True, first_order=True, S_out=Sv)
        else:
            E, S, y1p, y2 = arena.gather(ids, fm=True, first_order=True, S_out=Sv, sort_job=job if in_gather else None)
        if in_gather:
            job = None
        elif job is None and wk == 1 and not ux:
            arena.field_sort(ids_sort)
        sweeps, hot, last_sweep = None, None, None
        if ux and overlap and wk > 1:
            hot = ()            # (the window's sweep ran with the ids phase above)
        elif overlap and wk > 1 and wpos == 0:
            # ONE sweep for the whole window, as a launch of its own: k updates per row in registers make the slices
            # ALU-heavy, and as riders they inherit their carrier's occupancy (measured, DeepFM bs 256: carried 164 us per
            # 4-step window, stand-alone 73 us = 18 us per step against 53-60 us for a one-step sweep)
            cold, hot = arena.adam_split_segments(window_k=wk)
            store.opt.window_sweep(cold[::-1])
        elif overlap and wpos == 0:
            # Exact TF-1 Adam, split: the sort runs first (its slot map says which rows this step touches); the
            # HBM-bound sweep over the UNtouched rows (old state only) then rides along in the tower launches as extra
            # workgroups, filling the CUs the latency-bound tower leaves idle; touched rows + dense follow the scatter.
            cold, hot = arena.adam_split_segments()
            # (first-order vector first: the optional LAST slice, carried by the scatter launch, may hold table blocks only)
            sweeps = store.opt.cold_slices(cold[::-1], store.sweep_weights)      # [fwd_0.., head, bwd_{L-1}..bwd_0, scatter]
            last_sweep = sweeps[-1] if len(sweeps) == 2 * len(store.tower.widths) + 2 else None
            sweeps = sweeps[:2 * len(store.tower.widths) + 1]
            assert job is None or sweeps[0] is None, "a forward launch that carries the sort cannot carry a sweep slice"
        elif overlap:
            hot = ()            # a later position of the window: the window's sweep already ran
        # RSX_DP_OVERLAP=1 (opt-in): the dense arena is all-reduced per tower layer from inside backward, on RCCL's stream
        pending, layer_done = [], None
        if zc and dist_overlap_enabled():
            nl = len(store.tower.widths)
            per_layer, rest = dist_overlap_ranges(
                store.dense, [[f"dnn.{v}{l}" for v in ("W", "b", "gamma", "beta")] for l in range(nl)])
            g = store.dense.grad

            def layer_done(l):
                for lo, hi in ([per_layer[l]] + (rest if l == nl - 1 else [])):
                    pending.append(dp.all_reduce_async(g[lo:hi]))
        loss, prob, dX, gy1, gy2 = store.tower.train_step(
            E, labels.reshape(-1).to(torch.float32), params["dropout"], store.opt.state.view(torch.int32)[3:4],
            s0=y1p, c0="b1", s1=y2,
            replicas=dp.world if dp is not None else 1, masks=masks,
            seed=0x5eed + (7919 * dp.rank if dp is not None else 0),     # replicas draw independent dropout patterns
            sort_job=job, sweeps=sweeps, sort_in_fwd=overlap, outs=(dXv, gy1v, gy2v) if zc else None,
            layer_done=layer_done, gather=(arena, ids, S, y1p, y2) if fuse_gather else None)
        if ux:
            # the rank's own sorted segment-sum (what a single replica's scatter does), written as its block of the send buffer
            Gv, gw1v = dp.send_views(arena.ux.capT)
            arena.ux_segsum_local(dX.shape[0], S, dX, gy1, gy2, Gv, gw1v, wpos)

    def train_op():
        with torch.no_grad():
            Sg, dXg, gy1g, gy2g, blocks, Bg, dense_segs = S, dX, gy1, gy2, None, dX.shape[0], None
            if ux:
                # ONE collective [dense | G | gw1], then the touched-row Adam off the merged lists: N looked-up rows per global
                # unique row, summed in rank order
                if layer_done is not None:
                    dp.wait_all(pending)
                (G0, gw10), blocks, dense_segs = dp.gather_send_block(arena.ux.capT, fold_dense=True,
                                                                      dense_done=layer_done is not None)
                arena.select(wpos)
                arena.ux_merged_adam(G0, gw10, blocks[1], store.opt, dense_segs or store.dense.adam_segments(), last_sweep,
                                     window=(wk, wpos))
                return
            if zc:                  # ONE collective straight from the send block (dense arena + per-example block)
                if layer_done is not None:
                    dp.wait_all(pending)
                (dXg, Sg, gy2g, gy1g), blocks, dense_segs = dp.gather_send_block(
                    dX.shape[0], fold_dense=hot is not None, dense_done=layer_done is not None)
                Bg = dX.shape[0] * dp.world
            elif dp is not None:    # ONE collective: per-example gradient block + dense arena (summed in rank order);
                # the scatter then reads every rank's block in place from the gathered buffer
                dXg, Sg, gy1g, gy2g, blocks = dp.gather_example_grads(dX, S, gy1, gy2, dense=store.dense.grad, blocked=True)
                Bg = dX.shape[0] * dp.world
            if hot is not None:     # scatter + touched-row Adam + dense Adam in ONE launch; advances the beta powers
                arena.select(wpos)
                arena.segsum_adam(Bg, Sg, dXg, gy1g, gy2g, store.opt, dense_segs or store.dense.adam_segments(),
                                  last_sweep, blocks=blocks, window=(wk, wpos))
            else:
                arena.segsum(Bg, Sg, dXg, gy1g, gy2g, blocks=blocks)
                store.apply_gradients()

    return EstimatorSpec(ModeKeys.TRAIN, predictions={"prob": prob}, loss=loss[0], train_op=train_op)


# ---- driver (deepfm/deepfm.py:153-234 + fm/fm.py flags) -----------------------------------------
def define_flags(p=None):
    p = p or argparse.ArgumentParser()
    p.add_argument("--embedding_size", type=int, default=16)
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--task_type", default="train", help="{train, infer, eval}")
    p.add_argument("--num_epochs", type=int, default=10)
    p.add_argument("--deep_layers", default="100,100")
    p.add_argument("--train_path", default="/home/wangrc/criteo_data/train/")
    p.add_argument("--train_parts", type=int, default=150)
    p.add_argument("--eval_parts", type=int, default=5)
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--log_steps", type=int, default=100)
    p.add_argument("--save_checkpoints_steps", type=int, default=1000)
    p.add_argument("--num_parallel", type=int, default=8)
    p.add_argument("--mirror", type=lambda s: s.lower() in ("1", "true", "yes"), default=True)
    p.add_argument("--model_dir", default="./model/")
    p.add_argument("--adam_mode", default="tf1_dense")
    p.add_argument("--adam_window", type=int, default=0,
                   help="steps per optimizer window (include/rsx.h rsx_adam_window; bit-identical to single steps): 0 = the "
                        "model's default (8 up to batch 1024, 4 above), 1 = every step on its own")
    p.add_argument("--feature_set", default="criteo", choices=["criteo", "uid_iid"],
                   help="criteo: the 39-field pipeline of fm.py (BASELINE configs); uid_iid: deepfm.py as committed "
                        "(int64 u_id / i_id hashed into 500000 / 100000 buckets, int64 label)")
    return p


def input_fn(filenames, batch_size, num_epochs=-1, need_shuffle=False, num_parallel=8, layout=None, shard=None,
             shard_tail=False):
    if layout is not None and layout.columns[0].key in ("i_id", "u_id"):     # --feature_set uid_iid (the script as committed)
        from .input_pipeline import uid_iid_input_fn
        assert shard is None or shard[1] == 1, "--feature_set uid_iid: single replica only"
        return uid_iid_input_fn(filenames, batch_size, num_epochs, need_shuffle, layout)
    from .input_pipeline import criteo_input_fn
    return criteo_input_fn(filenames, batch_size, num_epochs, need_shuffle, num_parallel, layout, shard=shard,
                           shard_tail=shard_tail)


def make_params(FLAGS, linear="indicator_all"):
    if getattr(FLAGS, "feature_set", "criteo") == "uid_iid":
        # deepfm/deepfm.py AS COMMITTED (:28-51): two int64 id features u_id / i_id, hashed as decimal strings
        from .feature_columns import build_model_columns
        lin, emb = build_model_columns(FLAGS.embedding_size)
    else:
        lin, emb = build_feature_columns(FLAGS.embedding_size, linear)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": FLAGS.embedding_size,
              "learning_rate": FLAGS.learning_rate, "dropout": FLAGS.dropout, "deep_layers": FLAGS.deep_layers,
              "max_batch_size": FLAGS.batch_size}
    if getattr(FLAGS, "adam_window", 0):
        params["adam_window"] = FLAGS.adam_window
    return params


def run_main(model_fn, FLAGS, make_params_fn):
    """The `main(_)` driver shared by the Criteo scripts (fm/fm.py:173-224, deepfm/deepfm.py:153-234, ...)."""
    if FLAGS.mirror:
        # MirroredStrategy() = every GPU of the host (fm/fm.py:184-186): on a multi-GPU box this process becomes the launcher
        from . import dist
        rc = dist.maybe_spawn_mirror(FLAGS, model_fn.__module__, getattr(FLAGS, "_argv", None))
        if rc is not None:
            if rc != 0:
                raise SystemExit(rc)
            return None
    files = [FLAGS.train_path + "part-r-{:0>5}".format(i) for i in range(FLAGS.train_parts)]
    train_files, eval_files = files[:-FLAGS.eval_parts], files[-FLAGS.eval_parts:]
    params = make_params_fn(FLAGS)
    config = RunConfig(save_checkpoints_steps=FLAGS.save_checkpoints_steps, keep_checkpoint_max=5,
                       log_step_count_steps=FLAGS.log_steps, adam_mode=FLAGS.adam_mode)
    est = Estimator(model_fn, FLAGS.model_dir, params, config)
    shard = None
    if FLAGS.mirror:
        from . import dist
        dp = dist.attach_if_distributed(est)
        if dp is not None:
            # MirroredStrategy (fm/fm.py:184-194): ONE batch stream, successive batches go to successive replicas -- rank r
            # trains on batches r, r+N, ... (disjoint records, same step count on every rank: csrc/tfrecord_reader.cpp)
            # and evaluates its own share of the eval stream; the metric counters are summed in Estimator.evaluate.
            shard = (dp.rank, dp.world)
    layout = CriteoLayout.from_columns(params["embedding_feature_columns"])
    if FLAGS.task_type == "train":
        tr = TrainSpec(lambda: input_fn(train_files, FLAGS.batch_size, FLAGS.num_epochs, True, FLAGS.num_parallel, layout, shard))
        ev = EvalSpec(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.num_parallel, layout, shard, True), steps=200)
        return train_and_evaluate(est, tr, ev)
    if FLAGS.task_type == "eval":
        return est.evaluate(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.num_parallel, layout, shard, True), steps=200)
    if FLAGS.task_type == "infer":
        out = []
        for i, p in enumerate(est.predict(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.num_parallel, layout))):
            if est._is_chief():
                print(p)
            out.append(p)
            if i >= 9:
                break
        return out
    raise SystemExit("unknown --task_type %r" % FLAGS.task_type)


def main(argv=None):
    FLAGS = define_flags().parse_args(argv)
    FLAGS._argv = argv
    return run_main(model_fn, FLAGS, make_params)


if __name__ == "__main__":
    main()
