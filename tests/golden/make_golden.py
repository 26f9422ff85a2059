#!/usr/bin/env python
"""Generates the committed golden fixtures of tests/golden/ from the oracle.

The reference (TensorFlow-1.x scripts) cannot be executed here or on the GPU box, and holds no fixtures of its own
(SURVEY.md section 8c), so these vectors are produced by the repository's fp64 oracle -- itself pinned by the
Appendix-B known answers in tests/test_oracle_kats.py -- and committed so that (a) the oracle cannot drift silently
and (b) the GPU box compares the HIP path against fixed numbers.  Fixtures are data only (inputs + expected outputs).

usage: python tests/golden/make_golden.py        (rewrites tests/golden/*.npz / *.json / *.tfrecord)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import criteo, hashing, init, models, nn, tfrecord  # noqa: E402


def main():
    rng = np.random.default_rng(20190625)
    # 1. hashing / bucketize / framing ------------------------------------------------------------------------
    strs = [b"a", b"b", b"c", b"d", b"", b"NULL", b"05db9164", b"68fd1e64", b"7e0ccccf"] + \
           [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 300)]
    hk = {"strings_hex": [s.hex() for s in strs], "fingerprint64": [str(hashing.fingerprint64(s)) for s in strs],
          "tf_kats": {"a": "12917804110809363939", "b": "11795596070477164822", "c": "11430444447143000872",
                      "d": "4470636696479570465"},
          "crc32c_123456789": "0xE3069283", "masked_crc_123456789": "0xC78AB0E5",
          "frame_abc_hex": tfrecord.frame(b"abc").hex()}
    x = [0, 1, 2, 6, 20, 1000, 1e6]
    hk["bucketize_c1"] = {"x": x, "idx": criteo.bucketize(np.array(x, np.float32), criteo.CONT_BOUNDARIES[0]).tolist()}
    json.dump(hk, open(os.path.join(HERE, "host_kats.json"), "w"), indent=1)

    # 2. a tiny Criteo TFRecord shard written by the oracle codec + the ids it must parse to ---------------------
    n = 24
    cont = np.clip(np.floor(np.exp(rng.normal(2, 2, (n, 13)))), 0, 1e6).astype(np.float32)
    cont[:, 1] -= 3
    label = (rng.random(n) < 0.25).astype(np.float32)
    cat = [[(b"NULL" if rng.random() < 0.1 else ("%08x" % rng.integers(0, 1 << 32)).encode()) for _ in range(26)] for _ in range(n)]
    blob = b""
    for r in range(n):
        ex = {"_c0": [float(label[r])]}
        ex.update({"_c%d" % j: [float(cont[r, j - 1])] for j in range(1, 14)})
        ex.update({"_c%d" % j: [cat[r][j - 14]] for j in range(14, 40) if cat[r][j - 14] != b"NULL"})
        blob += tfrecord.frame(tfrecord.encode_example(ex))
    open(os.path.join(HERE, "criteo_24.tfrecord"), "wb").write(blob)
    np.savez_compressed(os.path.join(HERE, "criteo_24_expected.npz"), ids=criteo.transform_batch(cont, cat, 4.0), label=label, cont=cont)

    # 3. op-level vectors (fp64 expected outputs) -----------------------------------------------------------------
    B, F, H, N, D = 3, 5, 6, 20, 16
    X0 = rng.standard_normal((B, F, D)) * 0.3
    Xk = rng.standard_normal((B, H, D)) * 0.3
    W = rng.standard_normal((F * H, N)) * 0.1
    c = rng.standard_normal(N) * 0.1
    g = rng.standard_normal((B, N, D))
    out = models.cin_layer_fwd(X0, Xk, W, c)
    d0, dk, dW, dc = models.cin_layer_bwd(X0, Xk, W, out, g)
    np.savez_compressed(os.path.join(HERE, "cin_layer.npz"), X0=X0, Xk=Xk, W=W, c=c, g=g, out=out, dX0=d0, dXk=dk, dW=dW, dc=dc)
    x0 = rng.standard_normal((7, 624)) * 0.3
    Wc, Bc = rng.standard_normal((3, 624)) * 0.05, rng.standard_normal((3, 624)) * 0.05
    gx = rng.standard_normal((7, 624))
    xs, ss = models.cross_fwd(x0, Wc, Bc)
    dx0, dWc, dBc = models.cross_bwd(xs, ss, Wc, gx)
    np.savez_compressed(os.path.join(HERE, "cross_layers.npz"), x0=x0, W=Wc, Bc=Bc, g=gx, xL=xs[-1], dx0=dx0, dW=dWc, dB=dBc)

    # 4. a DeepFM training trajectory on a small layout: inputs are seeds + ids/labels, outputs after 3 steps ------
    rows = (3, 7, 40, 11, 600)
    off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    Bt, steps = 32, 3
    P = init.deepfm_params(7, 16, (32, 16), np.float64, off)
    ids = np.stack([np.stack([rng.integers(0, r, Bt) for r in rows], 1) for _ in range(steps)]).astype(np.int32)
    y = rng.integers(0, 2, (steps, Bt)).astype(np.float64)
    P0 = {k: v.copy() for k, v in P.items()}
    m, opt = models.DeepFM(P, off, 2, 0.0), nn.AdamTF1(dtype=np.float64)
    losses, probs = [], []
    for s in range(steps):
        probs.append(nn.sigmoid(m.forward(ids[s], train=False)))
        loss, _ = models.train_step(m, opt, (ids[s],), y[s])
        losses.append(float(loss))
    np.savez_compressed(os.path.join(HERE, "deepfm_trajectory.npz"), rows=np.array(rows), ids=ids, labels=y, losses=np.array(losses),
                        probs=np.stack(probs), **{"init." + k: v.astype(np.float32) for k, v in P0.items()},
                        **{"final." + k: v for k, v in P.items()})
    # 5. a DCN trajectory (3 cross layers) on the same small layout -----------------------------------------------
    Pd = init.dcn_params(11, 16, (32, 16), 3, np.float64, off)
    Pd0 = {k: v.copy() for k, v in Pd.items()}
    md, optd = models.DCN(Pd, off, 2, 0.0), nn.AdamTF1(dtype=np.float64)
    ids_d = np.stack([np.stack([rng.integers(0, r, Bt) for r in rows], 1) for _ in range(steps)]).astype(np.int32)
    y_d = rng.integers(0, 2, (steps, Bt)).astype(np.float64)
    losses, probs = [], []
    for s in range(steps):
        probs.append(nn.sigmoid(md.forward(ids_d[s], train=False)))
        loss, _ = models.train_step(md, optd, (ids_d[s],), y_d[s])
        losses.append(float(loss))
    np.savez_compressed(os.path.join(HERE, "dcn_trajectory.npz"), rows=np.array(rows), ids=ids_d, labels=y_d,
                        losses=np.array(losses), probs=np.stack(probs),
                        **{"init." + k: v.astype(np.float32) for k, v in Pd0.items()}, **{"final." + k: v for k, v in Pd.items()})

    # 6. a DIN trajectory: tiny vocabularies, ragged zero-padded histories (padding id 0), K = 16 -------------------
    n_item, n_cate, Bd, Pn, Kd = 50, 7, 6, 5, 16
    Pn_ = init.din_params(5, Kd, n_item, n_cate, np.float64)
    Pn_["item_bias"] += rng.standard_normal(n_item) * 0.01
    Pn0 = {k: v.copy() for k, v in Pn_.items()}
    mdin, optn = models.DIN(Pn_, 0.0), nn.AdamTF1(dtype=np.float64)
    cate_of = rng.integers(1, n_cate, n_item)
    cate_of[0] = 0
    bat, losses, probs = [], [], []
    for s in range(steps):
        i_id = rng.integers(1, n_item, Bd)
        hist = rng.integers(1, n_item, (Bd, Pn))
        hist[np.arange(Pn)[None, :] >= rng.integers(1, Pn + 1, Bd)[:, None]] = 0
        b = dict(i_id=i_id, i_cate=cate_of[i_id], u_iid_seq=hist, u_icat_seq=cate_of[hist], label=rng.integers(0, 2, Bd))
        args = (b["i_id"], b["i_cate"], b["u_iid_seq"], b["u_icat_seq"])
        probs.append(nn.sigmoid(mdin.forward(*args, train=False)))
        loss, _ = models.train_step(mdin, optn, args, b["label"].astype(np.float64))
        losses.append(float(loss))
        bat.append(b)
    np.savez_compressed(os.path.join(HERE, "din_trajectory.npz"), n_item=n_item, n_cate=n_cate, K=Kd,
                        losses=np.array(losses), probs=np.stack(probs),
                        **{"batch.%s" % k: np.stack([b[k] for b in bat]).astype(np.int64) for k in bat[0]},
                        **{"init." + k: v.astype(np.float32) for k, v in Pn0.items()}, **{"final." + k: v for k, v in Pn_.items()})
    # 7. an FM trajectory (fm/fm.py: first order + FM second order + 2->1 head) on the small layout ------------------
    Pf = init.deepfm_params(13, 16, (), np.float64, off, with_dnn=False)
    Pf["b1"] += 0.05
    Pf0 = {k: v.copy() for k, v in Pf.items()}
    mf, optf = models.FM(Pf, off), nn.AdamTF1(dtype=np.float64)
    ids_f = np.stack([np.stack([rng.integers(0, r, Bt) for r in rows], 1) for _ in range(steps)]).astype(np.int32)
    y_f = rng.integers(0, 2, (steps, Bt)).astype(np.float64)
    losses, probs = [], []
    for s in range(steps):
        probs.append(nn.sigmoid(mf.forward(ids_f[s], train=False)))
        loss, _ = models.train_step(mf, optf, (ids_f[s],), y_f[s])
        losses.append(float(loss))
    np.savez_compressed(os.path.join(HERE, "fm_trajectory.npz"), rows=np.array(rows), ids=ids_f, labels=y_f,
                        losses=np.array(losses), probs=np.stack(probs),
                        **{"init." + k: v.astype(np.float32) for k, v in Pf0.items()}, **{"final." + k: v for k, v in Pf.items()})
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
