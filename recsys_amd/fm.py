"""Factorization Machine on the Criteo 39-field pipeline -- MI355X-native mirror of `fm/fm.py`
(model_fn :115-170, build_feature_columns :47-97, flags :16-37; BASELINE config 1, the plumbing model).

logits = dense([relu(first_order), fm_second_order], 1).  The whole model is the fused embedding kernels
(gather + first-order + FM forward, sorted segment-sum backward) plus a 2->1 dense head.
"""
import torch

from . import layers as L
from .deepfm import build_variables as _build
from .deepfm import define_flags, input_fn, make_params, run_main  # noqa: F401
from .estimator import EstimatorSpec, ModeKeys, get_variable_store
from .ops import gather_fm


def model_fn(features, labels, mode, params):
    """fm/fm.py:115-170."""
    store = get_variable_store()
    ids = features["ids"]
    if not store.built:
        _build(store, params, capacity=max(int(params.get("max_batch_size", 0)), ids.shape[0]), with_dnn=False)
    arena, P = store.embeddings["input_layer"], store.dense
    training = mode == ModeKeys.TRAIN
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


def main(argv=None):
    return run_main(model_fn, define_flags().parse_args(argv), make_params)


if __name__ == "__main__":
    main()
