"""Deep Interest Network on Amazon-Electronics style data -- MI355X-native mirror of `din/din.py`
(model_fn :83-180, feature_description :44-50, flags :12-40; BASELINE config 5).

Variables (din/din.py:88-90): `i_item` bias [63002] (zeros), `i_id` [63002, K] and `i_cate` [802, K] (glorot-normal).
Two attention blocks with separate MLP weights (hard-coded 80-40, :85) pool the two id histories against the target
item / category (:103-128); MLP 100-50-20 on [q_item, pooled_item, pooled_cate] (:130-138); logits += i_item[i_id] (:139).
Hand kernels: row gathers, the attention-weighted masked history sum (fwd/bwd), the sparse-row dedup/segment-sum and
the non-lazy TF-1 Adam sweep.  The attention / head MLP GEMMs (M = B*P rows) are library GEMMs via torch.
"""
import argparse

import torch

from . import layers as L
from .estimator import Estimator, EstimatorSpec, EvalSpec, ModeKeys, RunConfig, TrainSpec, get_variable_store, \
    train_and_evaluate
from .ops import DinAttnFn, DinAttnPoolFn, DinPoolFn, FusedTower, SparseTable

ATTENTION_LAYERS = [80, 40]      # din/din.py:85 (the din_layers flag is ignored by the reference)
MLP_LAYERS = [100, 50, 20]       # din/din.py:86 (the deep_layers flag is ignored by the reference)
N_ITEM, N_CATE = 63002, 802      # din/din.py:88-90


def _prefer_rocblas():
    """The attention MLP's weight gradients are [K, B*P] x [B*P, N] GEMMs with B*P = 102 400: torch's default fp32 path
    (hipBLASLt) runs them in 270-360 us each, rocBLAS in 38-54 us (scripts/gemm_probe.py; 6 of them per step)."""
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.backends.cuda.preferred_blas_library("cublas")      # "cublas" names rocBLAS on ROCm
    except Exception:                                                  # knob absent in this torch: keep the default
        pass


def build_variables(store, params, B, P):
    _prefer_rocblas()
    K = params["embedding_size"]
    n_item, n_cate = params.get("n_item", N_ITEM), params.get("n_cate", N_CATE)
    world = store.dp.world if store.dp is not None else 1
    cap = (B * (P + 1)) * world
    # id 0 is the history padding (din/din.py:56-57,107): its gradient entries are exactly zero (masked sum), so the
    # sparse path may skip that one huge segment; direct lookups never use id 0
    item = SparseTable(n_item, K, cap, store.device, null_row=0)
    cate = SparseTable(n_cate, K, cap, store.device, null_row=0)
    bias = SparseTable(n_item, 4, B * world, store.device)     # i_item [n_item] stored as column 0 of a 4-wide table
    with torch.no_grad():
        for tbl, rows in ((item, n_item), (cate, n_cate)):
            t = torch.empty(rows, K)
            L.glorot_normal_(t, rows, K, store.gen)             # tf.glorot_normal_initializer over [rows, K]
            tbl.table.copy_(t)
    shapes, init = {}, {}
    zeros = lambda t, g: t.zero_()

    def add(pre, d, widths):
        for i, n in enumerate(widths):
            shapes[f"{pre}.W{i}"], shapes[f"{pre}.b{i}"] = (d, n), (n,)
            init[f"{pre}.W{i}"] = lambda t, g, fi=d, fo=n: L.glorot_uniform_(t, fi, fo, g)
            init[f"{pre}.b{i}"] = zeros
            d = n
        return d

    add("att_i", 4 * K, ATTENTION_LAYERS + [1])
    add("att_c", 4 * K, ATTENTION_LAYERS + [1])
    d = add("mlp", 3 * K, MLP_LAYERS)
    shapes["mlp.Wout"], shapes["mlp.bout"] = (d, 1), (1,)
    init["mlp.Wout"] = lambda t, g, fi=d: L.glorot_uniform_(t, fi, 1, g)
    init["mlp.bout"] = zeros
    # The fused tower reads activation rows as float4: layer widths are STORED padded to a multiple of 4 (50 -> 52); the
    # variables keep their reference shapes as leading slices, the pad units have zero weights, zero bias, zero gradient.
    pad4 = lambda n: (n + 3) & ~3
    widths = [pad4(n) for n in MLP_LAYERS]
    storage, dprev = {}, 3 * K
    for i, n in enumerate(widths):
        storage[f"mlp.W{i}"], storage[f"mlp.b{i}"] = (dprev, n), (n,)
        dprev = n
    storage["mlp.Wout"] = (dprev, 1)
    store.build({"i_id": item, "i_cate": cate, "i_item": bias}, shapes, init, params["learning_rate"], storage)
    store.tower = None
    if (3 * K) % 4 == 0 and widths[-1] <= 256:
        store.tower = FusedTower(store.dense, "mlp", 3 * K, widths, B, store.device, batch_norm=False)


class DinHeadFn(torch.autograd.Function):
    """'mlp_layer' + logits + loss of the TRAIN step (din/din.py:131-147) through the fused tower kernels (csrc/tower.hip,
    no batch-norm): forward AND backward run inside forward(); backward() hands the stored input gradients to autograd
    (the attention / pooling / lookups upstream stay autograd Functions).  The loss already carries the 1/(B*replicas)
    scale, so backward() expects an incoming gradient of 1."""

    @staticmethod
    def forward(ctx, X, i_b, labels, store, rate, masks, replicas):
        t = store.tower
        if masks is not None:       # injected keep masks (parity tests) arrive with the reference widths: pad to storage
            masks = [torch.nn.functional.pad(m, (0, w - m.shape[1]), value=1.0) for m, w in zip(masks, t.widths)]
        loss, prob, dX, gs0, _ = t.train_step(
            X.contiguous(), labels.reshape(-1).to(torch.float32), rate, store.opt.state.view(torch.int32)[3:4],
            s0=i_b.contiguous(), head=("mlp.Wout", "mlp.bout", None, None), relu0=False, relu2=False, replicas=replicas,
            masks=masks, seed=0xD1AD)
        ctx.save_for_backward(dX, gs0)
        ctx.mark_non_differentiable(prob)
        return loss[0], prob

    @staticmethod
    def backward(ctx, g_loss, g_prob):
        dX, gs0 = ctx.saved_tensors
        return dX, gs0, None, None, None, None, None


def _attention(tbl, hist, q, P_, pre, training, rate, masks, store=None, layer0=0):
    """din/din.py:103-125."""
    B, Pn = hist.shape
    K = tbl.K
    H = tbl.lookup(hist)                                                   # dense_emb [B,P,K] (:105)
    n1, n2 = P_[f"{pre}.W0"].shape[1], P_[f"{pre}.W1"].shape[1]
    if len(ATTENTION_LAYERS) == 2 and DinAttnFn.supported(K, n1, n2) and store is not None:
        # fused MFMA kernel: the [B*P, 4K] concat, the tiled query and the layer outputs never round-trip through HBM;
        # dropout = the counter hash of the fused tower (keyed by the optimizer's device-side step counter)
        r = rate if training else 0.0
        mk2 = masks if (training and r > 0.0) else None
        names = [f"{pre}.{v}{i}" for i in range(3) for v in ("W", "b")]
        gout = P_.packed_grad(names) if training else None
        Ws = [P_[n] for n in names]
        step = store.opt.state.view(torch.int32)[3:4]
        if gout is not None:        # TRAIN: attention + pooling as one node, weight gradients straight into the arena
            return DinAttnPoolFn.apply(H, q, hist, *Ws, r, mk2, step, 0xD1A77, layer0, gout)
        w = DinAttnFn.apply(H, q, *Ws, r, mk2, step, 0xD1A77, layer0)
        return DinPoolFn.apply(H, w, hist)                                  # masked weighted sum (:122-124)
    hist_emb = H.reshape(B * Pn, K)
    query_emb = q[:, None, :].expand(B, Pn, K).reshape(B * Pn, K)          # tile + reshape (:111)
    att = torch.cat([hist_emb, query_emb, hist_emb * query_emb, hist_emb - query_emb], 1)     # (:114)
    for i in range(len(ATTENTION_LAYERS)):
        att = L.dense(att, P_[f"{pre}.W{i}"], P_[f"{pre}.b{i}"], relu=True)
        att = L.dropout(att, rate, training, None if masks is None else masks[i])
    w = L.dense(att, P_[f"{pre}.W2"], P_[f"{pre}.b2"]).reshape(B, Pn)        # att_wgt (:120-121)
    return DinPoolFn.apply(H, w, hist)                                      # masked weighted sum (:122-124)


def model_fn(features, labels, mode, params):
    """din/din.py:83-180.  features: i_id, i_cate int [B]; u_iid_seq, u_icat_seq int [B,P] (zero padded)."""
    store = get_variable_store()
    i_id = features["i_id"].to(torch.int32)
    i_cate = features["i_cate"].to(torch.int32)
    hist_i = features["u_iid_seq"].to(torch.int32).contiguous()
    hist_c = features["u_icat_seq"].to(torch.int32).contiguous()
    if not store.built:
        build_variables(store, params, max(int(params.get("max_batch_size", 0)), i_id.shape[0]), hist_i.shape[1])
    item, cate, bias = (store.embeddings[k] for k in ("i_id", "i_cate", "i_item"))
    P_ = store.dense
    training = mode == ModeKeys.TRAIN
    rate = params["dropout"]
    mk = params.get("_dropout_masks") or {}

    i_b = bias.lookup(i_id)[:, 0]                                          # tf.gather(pkg_w, i_id) (:96)
    pkg_emb = item.lookup(i_id)                                            # (:100)
    pkgc_emb = cate.lookup(i_cate)                                         # (:101)
    fused = store if params.get("fused_attention", True) else None
    pkg_emb_h = _attention(item, hist_i, pkg_emb, P_, "att_i", training, rate, mk.get("att_i"), fused, 0)
    pkgc_emb_h = _attention(cate, hist_c, pkgc_emb, P_, "att_c", training, rate, mk.get("att_c"), fused, 2)
    net = torch.cat([pkg_emb, pkg_emb_h, pkgc_emb_h], 1)                   # 'mlp_layer' (:131)
    fused_head = training and store.tower is not None and params.get("fused_head", True)
    if fused_head:
        dp = store.dp
        loss, pred = DinHeadFn.apply(net, i_b, labels, store, rate, mk.get("mlp"), dp.world if dp is not None else 1)

        def train_op_fused():                                              # AdamOptimizer.minimize (:172-173)
            loss.backward()                                                # (1/world is inside the fused head's loss scale)
            with torch.no_grad():
                for tbl in (item, cate, bias):
                    tbl.finalize(dp)
                if dp is not None:
                    dp.all_reduce_sum(store.dense.grad)
                store.apply_gradients()

        return EstimatorSpec(mode, predictions={"prob": pred}, loss=loss, train_op=train_op_fused)
    for i in range(len(MLP_LAYERS)):
        net = L.dense(net, P_[f"mlp.W{i}"], P_[f"mlp.b{i}"], relu=True)
        net = L.dropout(net, rate, training, mk["mlp"][i] if "mlp" in mk else None)
    logits = L.dense(net, P_["mlp.Wout"], P_["mlp.bout"]).reshape(-1) + i_b   # (:138-139)
    pred = torch.sigmoid(logits)
    predictions = {"prob": pred}
    if mode == ModeKeys.PREDICT:
        return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
    loss = L.sigmoid_ce_mean(logits, labels)
    if mode == ModeKeys.EVAL:
        return EstimatorSpec(mode, predictions=predictions, loss=loss, eval_metric_ops={"AUC": None, "Accuracy": None})
    dp = store.dp
    if dp is not None:          # MirroredStrategy scales the replica loss by 1/N (the fused head does the same inside)
        loss = loss / dp.world

    def train_op():                                                        # AdamOptimizer.minimize (:172-173)
        loss.backward()
        with torch.no_grad():
            for tbl in (item, cate, bias):
                tbl.finalize(dp)
            if dp is not None:
                dp.all_reduce_sum(store.dense.grad)
            store.apply_gradients()

    return EstimatorSpec(mode, predictions=predictions, loss=loss, train_op=train_op)


def define_flags():
    p = argparse.ArgumentParser()
    p.add_argument("--embedding_size", type=int, default=32)
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--task_type", default="train")
    p.add_argument("--num_epochs", type=int, default=5)
    p.add_argument("--deep_layers", default="100,100")        # accepted, ignored like the reference (:19,85-86)
    p.add_argument("--din_layers", default="80,40")
    p.add_argument("--train_path", default="/home/wangrc/din_dataset/")
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--log_steps", type=int, default=100)
    p.add_argument("--eval_steps", type=int, default=200)
    p.add_argument("--save_checkpoints_steps", type=int, default=1000)
    p.add_argument("--mirror", type=lambda s: s.lower() in ("1", "true", "yes"), default=True)
    p.add_argument("--model_dir", default="./din_model/")
    p.add_argument("--hist_len", type=int, default=100)
    return p


def input_fn(filenames, batch_size, num_epochs=-1, need_shuffle=False, hist_len=100, shard=None):
    from .input_pipeline import din_input_fn
    return din_input_fn(filenames, batch_size, num_epochs, need_shuffle, hist_len=hist_len, ids_int32=True, shard=shard)


def main(argv=None):
    FLAGS = define_flags().parse_args(argv)
    train_files, eval_files = [FLAGS.train_path + "train2"], [FLAGS.train_path + "valid2"]      # din/din.py:197-198
    params = {"embedding_size": FLAGS.embedding_size, "learning_rate": FLAGS.learning_rate, "dropout": FLAGS.dropout,
              "max_batch_size": FLAGS.batch_size, "hist_len": FLAGS.hist_len}
    cfg = RunConfig(save_checkpoints_steps=FLAGS.save_checkpoints_steps, keep_checkpoint_max=5,
                    log_step_count_steps=FLAGS.log_steps)
    est = Estimator(model_fn, FLAGS.model_dir, params, cfg)
    shard = None
    if FLAGS.mirror:
        from . import dist
        dp = dist.attach_if_distributed(est)
        if dp is not None:          # every replica takes its own batches of the one stream (MirroredStrategy, din/din.py:187-190)
            shard = (dp.rank, dp.world)
    if FLAGS.task_type == "train":
        tr = TrainSpec(lambda: input_fn(train_files, FLAGS.batch_size, FLAGS.num_epochs, True, FLAGS.hist_len, shard))
        ev = EvalSpec(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.hist_len, shard), steps=FLAGS.eval_steps)
        return train_and_evaluate(est, tr, ev)
    if FLAGS.task_type == "eval":
        return est.evaluate(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.hist_len, shard), steps=FLAGS.eval_steps)
    return list(zip(range(10), est.predict(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.hist_len))))


if __name__ == "__main__":
    main()
