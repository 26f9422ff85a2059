"""The oracle and the product's host functions against the COMMITTED golden fixtures (tests/golden/)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import criteo, hashing, init, models, nn, tfrecord

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_host_kats_oracle_and_product():
    from recsys_amd import _lib, build
    build.build(verbose=False)
    L = _lib.lib()
    k = json.load(open(os.path.join(G, "host_kats.json")))
    for hx, want in zip(k["strings_hex"], k["fingerprint64"]):
        s = bytes.fromhex(hx)
        assert str(hashing.fingerprint64(s)) == want
        arr = np.frombuffer(s, np.uint8) if s else np.zeros(0, np.uint8)
        assert str(L.rsx_fingerprint64_h(arr.ctypes.data_as(C.c_void_p), len(s))) == want
    for s, want in k["tf_kats"].items():                       # upstream TF values
        assert str(hashing.fingerprint64(s.encode())) == want
    assert tfrecord.crc32c(b"123456789") == int(k["crc32c_123456789"], 16)
    assert tfrecord.masked_crc(b"123456789") == int(k["masked_crc_123456789"], 16)
    assert tfrecord.frame(b"abc").hex() == k["frame_abc_hex"]
    b = k["bucketize_c1"]
    assert criteo.bucketize(np.array(b["x"], np.float32), criteo.CONT_BOUNDARIES[0]).tolist() == b["idx"] == [1, 1, 2, 2, 4, 5, 6]


def test_golden_shard_parses_to_golden_ids():
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    from recsys_amd.input_pipeline import criteo_input_fn
    exp = np.load(os.path.join(G, "criteo_24_expected.npz"))
    lay = CriteoLayout.from_columns(build_feature_columns(16)[1])
    got = list(criteo_input_fn([os.path.join(G, "criteo_24.tfrecord")], 24, num_epochs=1, layout=lay))
    assert len(got) == 1
    assert np.array_equal(got[0][0]["ids"], exp["ids"]) and np.array_equal(got[0][1].reshape(-1), exp["label"])
    recs = list(tfrecord.unframe(open(os.path.join(G, "criteo_24.tfrecord"), "rb").read()))
    assert len(recs) == 24 and tfrecord.decode_example(recs[0])["_c0"] == [float(exp["label"][0])]


def test_oracle_reproduces_golden_ops_and_trajectory():
    c = np.load(os.path.join(G, "cin_layer.npz"))
    out = models.cin_layer_fwd(c["X0"], c["Xk"], c["W"], c["c"])
    np.testing.assert_allclose(out, c["out"], rtol=0, atol=1e-14)
    d0, dk, dW, dc = models.cin_layer_bwd(c["X0"], c["Xk"], c["W"], out, c["g"])
    for a, b in ((d0, c["dX0"]), (dk, c["dXk"]), (dW, c["dW"]), (dc, c["dc"])):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)
    x = np.load(os.path.join(G, "cross_layers.npz"))
    xs, ss = models.cross_fwd(x["x0"], x["W"], x["Bc"])
    np.testing.assert_allclose(xs[-1], x["xL"], rtol=0, atol=1e-13)
    t = np.load(os.path.join(G, "deepfm_trajectory.npz"))
    rows = tuple(int(r) for r in t["rows"])
    off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    P = {k[5:]: t[k].astype(np.float64) for k in t.files if k.startswith("init.")}
    m, opt = models.DeepFM(P, off, 2, 0.0), nn.AdamTF1(dtype=np.float64)
    for s in range(t["ids"].shape[0]):
        loss, _ = models.train_step(m, opt, (t["ids"][s],), t["labels"][s])
        assert abs(float(loss) - float(t["losses"][s])) < 1e-6      # init.* is stored in fp32
    for k in P:
        np.testing.assert_allclose(P[k], t["final." + k], rtol=0, atol=2e-6, err_msg=k)


def test_oracle_reproduces_golden_dcn_and_din_trajectories():
    t = np.load(os.path.join(G, "dcn_trajectory.npz"))
    rows = tuple(int(r) for r in t["rows"])
    off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    P = {k[5:]: t[k].astype(np.float64) for k in t.files if k.startswith("init.")}
    m, opt = models.DCN(P, off, 2, 0.0), nn.AdamTF1(dtype=np.float64)
    for s in range(t["ids"].shape[0]):
        loss, _ = models.train_step(m, opt, (t["ids"][s],), t["labels"][s])
        assert abs(float(loss) - float(t["losses"][s])) < 1e-6
    for k in P:
        np.testing.assert_allclose(P[k], t["final." + k], rtol=0, atol=2e-6, err_msg=k)
    d = np.load(os.path.join(G, "din_trajectory.npz"))
    P = {k[5:]: d[k].astype(np.float64) for k in d.files if k.startswith("init.")}
    m, opt = models.DIN(P, 0.0), nn.AdamTF1(dtype=np.float64)
    for s in range(d["losses"].shape[0]):
        args = tuple(d["batch." + k][s] for k in ("i_id", "i_cate", "u_iid_seq", "u_icat_seq"))
        loss, _ = models.train_step(m, opt, args, d["batch.label"][s].astype(np.float64))
        assert abs(float(loss) - float(d["losses"][s])) < 1e-6
    for k in P:
        np.testing.assert_allclose(P[k], d["final." + k], rtol=0, atol=2e-6, err_msg=k)


def test_oracle_reproduces_golden_fm_trajectory():
    t = np.load(os.path.join(G, "fm_trajectory.npz"))
    rows = tuple(int(r) for r in t["rows"])
    off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    P = {k[5:]: t[k].astype(np.float64) for k in t.files if k.startswith("init.")}
    m, opt = models.FM(P, off), nn.AdamTF1(dtype=np.float64)
    for s in range(t["ids"].shape[0]):
        loss, _ = models.train_step(m, opt, (t["ids"][s],), t["labels"][s])
        assert abs(float(loss) - float(t["losses"][s])) < 1e-6
    for k in P:
        np.testing.assert_allclose(P[k], t["final." + k], rtol=0, atol=2e-6, err_msg=k)
