"""Pins the committed full-size fixtures (tests/golden/full_*.npz, produced by the numpy oracle with its HAND-WRITTEN
backward passes) with the independent restatement: `oracle.torch_ref` -- the reference scripts op by op on PyTorch-CPU
fp64, every gradient from autograd, a literal TF-1 Adam -- regenerates every committed quantity (eval probabilities
before each step, the train losses, every final dense variable, the sampled touched / untouched table rows) from the
seeded inputs and must agree: 1e-9 on the fp64 quantities, fp32 rounding on the ones stored as fp32.

The fixtures are what the GPU parity tests at BASELINE.json's batch sizes compare against (tests/test_gpu_fullsize.py), so
an error of understanding shared by the oracle's forward and its hand-written backward could otherwise hide in them.
(The reference itself holds no golden vectors and TensorFlow cannot run here: parity stays "unpinned" -- this makes the
same-author chain two independent derivations deep at full size, not only at toy sizes.)"""
import os

import numpy as np
import pytest
import torch

from oracle import criteo, init, torch_ref as tr
from tests import fullsize

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _logits_fn(kind, b, row_off, train):
    t = lambda a: torch.tensor(np.asarray(a).astype(np.int64))
    if kind == "deepfm":
        return lambda T: tr.deepfm_logits(T, t(b["ids"]), row_off, 2, 0.0, None, train)
    if kind == "dcn":
        return lambda T: tr.dcn_logits(T, t(b["ids"]), row_off, 2, 0.0, None, train)
    if kind == "xdeepfm":
        cat_slot, cat_off = init.xdeepfm_layout()
        logx = torch.tensor(b["cont_log"].astype(np.float64))
        return lambda T: tr.xdeepfm_logits(T, t(b["ids"]), logx, row_off, cat_slot, cat_off, (128, 128), 2, 0.0, None, train)
    return lambda T: tr.din_logits(T, t(b["i_id"]), t(b["i_cate"]), t(b["u_iid_seq"]), t(b["u_icat_seq"]), 0.0, None, train)


@pytest.mark.parametrize("name", list(fullsize.CONFIGS))
def test_torch_autograd_regenerates_the_fullsize_fixture(name):
    kind, B, seed = fullsize.CONFIGS[name]
    gold = np.load(os.path.join(GOLD, "full_%s.npz" % name))
    P32, batches, digest = fullsize.make_inputs(name)
    assert digest == str(gold["digest"]), "inputs differ from the ones the fixture was generated from"
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    row_off = criteo.row_offsets()
    T = tr.params_to_torch({k: v.astype(np.float64) for k, v in P32.items()})
    opt = tr.AdamTF1()
    probs, losses = [], []
    for b in batches:
        with torch.no_grad():
            probs.append(torch.sigmoid(_logits_fn(kind, b, row_off, False)(T)).numpy().reshape(-1))
        loss, _, grads = tr.loss_and_grads(_logits_fn(kind, b, row_off, True), T, torch.tensor(b["label"].astype(np.float64)))
        losses.append(float(loss))
        opt.step(T, grads)
    np.testing.assert_allclose(np.stack(probs), gold["probs"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(losses), gold["losses"], rtol=0, atol=1e-9)
    checked = 0
    for key in gold.files:
        if key.startswith("final."):
            got = T[key[6:]].detach().numpy()
            np.testing.assert_allclose(got.astype(np.float32), gold[key], rtol=2e-7, atol=1e-9, err_msg=key)
            checked += 1
        elif key.startswith("rows."):
            k = key[5:]
            got = T[k].detach().numpy()[gold[key]]
            np.testing.assert_allclose(got.astype(np.float32), gold["vals." + k], rtol=2e-7, atol=1e-9, err_msg=k)
            checked += 1
    assert checked == len(P32)          # every variable of the model is pinned (dense: whole, tables: the sampled rows)
