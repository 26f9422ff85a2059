"""CPU tests of the host ingest (librsx.so host functions + recsys_amd.input_pipeline) against the oracle codec."""
import numpy as np
import pytest

from oracle import criteo, tfrecord


@pytest.fixture(scope="module")
def layout():
    from recsys_amd import build
    build.build(verbose=False)
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    return CriteoLayout.from_columns(build_feature_columns(16)[1])


def _raw(n, seed=0):
    from recsys_amd import synthetic
    return synthetic.criteo_raw_batch(np.random.default_rng(seed), n)


def test_writer_output_decodes_with_the_oracle_codec(tmp_path, layout):
    from recsys_amd.input_pipeline import write_criteo_shard
    label, cont, cat = _raw(37)
    p = tmp_path / "part-r-00000"
    write_criteo_shard(str(p), label, cont, cat)
    recs = list(tfrecord.unframe(p.read_bytes()))            # framing + crc checked by the oracle
    assert len(recs) == 37
    for r, rec in enumerate(recs):
        ex = tfrecord.decode_example(rec)
        assert ex["_c0"] == [float(label[r])]
        for j in range(1, 14):
            assert ex["_c%d" % j] == [float(cont[r, j - 1])]
        for j in range(14, 40):
            v = cat[r][j - 14]
            if v == b"NULL":
                assert "_c%d" % j not in ex                   # nulls are omitted (Spark connector), parse restores 'NULL'
            else:
                assert ex["_c%d" % j] == [v]


def test_parse_of_oracle_encoded_records_matches_oracle_transform(tmp_path, layout):
    """Shard written by the ORACLE's encoder (unpacked + packed variants), read by the product's C++ parser."""
    from recsys_amd.input_pipeline import criteo_input_fn
    label, cont, cat = _raw(50, seed=3)
    blob = b""
    for r in range(50):
        ex = {"_c0": [float(label[r])]}
        ex.update({"_c%d" % j: [float(cont[r, j - 1])] for j in range(1, 14)})
        ex.update({"_c%d" % j: [cat[r][j - 14]] for j in range(14, 40) if cat[r][j - 14] != b"NULL"})
        blob += tfrecord.frame(tfrecord.encode_example(ex))
    p = tmp_path / "part-r-00000"
    p.write_bytes(blob)
    got = list(criteo_input_fn([str(p)], batch_size=16, num_epochs=1, layout=layout))
    assert [b[1].shape[0] for b in got] == [16, 16, 16, 2]       # final partial batch is kept (A-10)
    ids = np.concatenate([b[0]["ids"] for b in got])
    lab = np.concatenate([b[1] for b in got])
    logx = np.concatenate([b[0]["cont_log"] for b in got])
    assert np.array_equal(ids, criteo.transform_batch(cont, cat, c2_shift=4.0))
    assert np.array_equal(lab.reshape(-1), label)
    shift = np.ones(13, np.float32)
    shift[1] = 4.0
    with np.errstate(invalid="ignore", divide="ignore"):
        # libm logf vs numpy's SIMD log differ by <= 1 ulp (TF's Eigen plog is a third implementation); the bucket ids
        # above are exact because integer Criteo values never put log(x+shift) within an ulp of a boundary
        np.testing.assert_allclose(logx, np.log(cont + shift), rtol=3e-7, atol=0, equal_nan=True)


def test_corrupt_and_truncated_shards_fail_loudly(tmp_path, layout):
    from recsys_amd._lib import RsxError
    from recsys_amd.input_pipeline import read_shard, write_criteo_shard
    label, cont, cat = _raw(5)
    p = tmp_path / "s"
    write_criteo_shard(str(p), label, cont, cat)
    raw = bytearray(p.read_bytes())
    raw[40] ^= 0x01
    (tmp_path / "bad").write_bytes(bytes(raw))
    with pytest.raises(RsxError):
        read_shard(str(tmp_path / "bad"))
    (tmp_path / "trunc").write_bytes(bytes(p.read_bytes()[:-3]))
    with pytest.raises(RsxError):
        read_shard(str(tmp_path / "trunc"))
    (tmp_path / "empty").write_bytes(b"")
    assert len(read_shard(str(tmp_path / "empty"))[1]) == 0


def test_input_fn_epochs_shuffle_and_multi_file(tmp_path, layout):
    from recsys_amd.input_pipeline import criteo_input_fn, write_criteo_shard
    files = []
    for k in range(3):
        label, cont, cat = _raw(40, seed=10 + k)
        label[:] = np.arange(40) + 100 * k                      # tag records through the label
        f = tmp_path / ("part-r-%05d" % k)
        write_criteo_shard(str(f), label, cont, cat)
        files.append(str(f))
    plain = list(criteo_input_fn(files, 32, num_epochs=2, layout=layout))
    labs = np.concatenate([b[1].reshape(-1) for b in plain])
    one = np.concatenate([np.arange(40) + 100 * k for k in range(3)])
    assert np.array_equal(labs, np.concatenate([one, one]))     # batches cross file boundaries; repeat(2)
    assert [b[1].shape[0] for b in plain] == [32, 32, 32, 24] * 2
    shuf = list(criteo_input_fn(files, 8, num_epochs=1, need_shuffle=True, layout=layout, shuffle_buffer=4, seed=1))
    firsts = [int(b[1][0, 0]) for b in shuf]
    assert sorted(firsts) == sorted(int(x) for x in one[::8]) and firsts != sorted(firsts)   # whole batches permuted
    for b in shuf:                                             # ... and never mixed inside
        v = b[1].reshape(-1)
        assert np.array_equal(v, v[0] + np.arange(len(v))) or len(set((v // 100).tolist())) > 1
    it = criteo_input_fn(files, 64, num_epochs=-1, layout=layout)
    assert sum(next(it)[1].shape[0] for _ in range(5)) > 120    # infinite repeat


def test_din_roundtrip(tmp_path):
    from recsys_amd import synthetic
    from recsys_amd.input_pipeline import din_input_fn, write_din_shard
    b = synthetic.din_batch(np.random.default_rng(0), 33, P=20, n_item=500, n_cate=30)
    p = tmp_path / "train2"
    write_din_shard(str(p), b)
    recs = list(tfrecord.unframe(p.read_bytes()))
    ex = tfrecord.decode_example(recs[0])
    n0 = int((b["u_iid_seq"][0] > 0).sum())
    assert ex["i_id"] == [int(b["i_id"][0])] and ex["u_iid_seq"] == b["u_iid_seq"][0, :n0].tolist()   # ragged on disk
    got = list(din_input_fn([str(p)], 16, num_epochs=1, hist_len=20))
    assert [g[1].shape[0] for g in got] == [16, 16, 1]
    for k in ("i_id", "i_cate", "u_iid_seq", "u_icat_seq"):
        assert np.array_equal(np.concatenate([g[0][k] for g in got]), b[k])                            # zero padded back to P
    assert np.array_equal(np.concatenate([g[1] for g in got]), b["label"])


def _tagged_files(tmp_path, counts, seed=20):
    """Shards whose label carries the global record number, so that a consumed stream can be audited."""
    from recsys_amd.input_pipeline import write_criteo_shard
    files, base = [], 0
    for k, n in enumerate(counts):
        label, cont, cat = _raw(n, seed=seed + k)
        label[:] = np.arange(n) + base
        base += n
        f = tmp_path / ("part-r-%05d" % k)
        write_criteo_shard(str(f), label, cont, cat)
        files.append(str(f))
    return files, base


def test_reader_shards_the_batch_stream_by_rank(tmp_path, layout):
    """MirroredStrategy semantics (fm/fm.py:184-194): batch b of the one stream goes to replica b % N.  Ranks consume
    disjoint records, the same number of equal-size batches each, and together every complete round of the epoch."""
    from recsys_amd.input_pipeline import criteo_input_fn
    files, total = _tagged_files(tmp_path, [70, 33, 58])           # 161 records
    bs = 8
    for world in (2, 3, 4):
        per_rank = []
        for rank in range(world):
            got = list(criteo_input_fn(files, bs, num_epochs=2, layout=layout, shard=(rank, world), num_parallel=3))
            assert all(b[1].shape[0] == bs for b in got)           # complete rounds only: no partial batch under DP
            per_rank.append([b[1].reshape(-1).astype(np.int64) for b in got])
        nb = {len(x) for x in per_rank}
        rounds = total // (bs * world)
        assert nb == {2 * rounds}                                  # same step count on every rank, both epochs
        for rank in range(world):
            for i, lab in enumerate(per_rank[rank]):
                r = i % rounds                                     # round within the epoch
                first = (r * world + rank) * bs
                assert np.array_equal(lab, np.arange(first, first + bs)), (world, rank, i)
        seen = np.concatenate([np.concatenate(x[:rounds]) for x in per_rank])
        assert len(set(seen.tolist())) == len(seen) == rounds * world * bs       # disjoint
    # EVALUATION under sharding (one pass, no shuffle): nothing is dropped -- leftover full batches of the last
    # round and the final partial batch go to rank b % world, and the ranks together cover every record exactly once
    for world in (2, 3, 4):
        per_rank = [list(criteo_input_fn(files, bs, num_epochs=1, layout=layout, shard=(rank, world), num_parallel=3,
                                         shard_tail=True))
                    for rank in range(world)]
        nbatches = (total + bs - 1) // bs
        for rank in range(world):
            mine = [b for b in range(nbatches) if b % world == rank]
            assert len(per_rank[rank]) == len(mine), (world, rank)
            for got, b in zip(per_rank[rank], mine):
                want = np.arange(b * bs, min(total, (b + 1) * bs))
                assert np.array_equal(got[1].reshape(-1).astype(np.int64), want), (world, rank, b)
        seen = np.concatenate([g[1].reshape(-1) for x in per_rank for g in x]).astype(np.int64)
        assert sorted(seen.tolist()) == list(range(total))
        # TRAIN-style call of the same files (shard_tail off = the DEFAULT, also for one unshuffled epoch): complete rounds only
        t0 = list(criteo_input_fn(files, bs, num_epochs=1, layout=layout, shard=(0, world)))
        assert len(t0) == total // (bs * world)
    # world 1 keeps the partial batch, drop handled by the caller's choice
    got = list(criteo_input_fn(files, bs, num_epochs=1, layout=layout, shard=(0, 1)))
    assert sum(b[1].shape[0] for b in got) == total and got[-1][1].shape[0] == total % bs


def test_reader_reports_corruption_in_stream_order(tmp_path, layout):
    from recsys_amd._lib import RsxError
    from recsys_amd.input_pipeline import criteo_input_fn
    files, total = _tagged_files(tmp_path, [64])
    raw = bytearray(open(files[0], "rb").read())
    # flip one payload byte of a record in the second half: batches before it are delivered, then the error surfaces
    raw[len(raw) * 3 // 4] ^= 0x40
    bad = tmp_path / "bad"
    bad.write_bytes(bytes(raw))
    it = criteo_input_fn([str(bad)], 8, num_epochs=1, layout=layout, num_parallel=2)
    n_ok = 0
    with pytest.raises(RsxError):
        for _f, lab in it:
            n_ok += lab.shape[0]
    assert 8 <= n_ok < 64
    # the same shard with verification off parses (the flipped byte sits inside a value) or reports malformed data --
    # but never crashes
    try:
        list(criteo_input_fn([str(bad)], 8, num_epochs=1, layout=layout, verify_crc=False))
    except RsxError:
        pass
    trunc = tmp_path / "trunc"
    trunc.write_bytes(bytes(open(files[0], "rb").read()[:-5]))
    with pytest.raises(RsxError):
        list(criteo_input_fn([str(trunc)], 8, num_epochs=1, layout=layout))
    with pytest.raises(RsxError):
        list(criteo_input_fn([str(tmp_path / "does-not-exist")], 8, num_epochs=1, layout=layout))


def test_reader_stops_when_the_consumer_leaves_early(tmp_path, layout):
    """evaluate(steps=...) and predict break out of an infinite stream: the reader's threads must go away with the
    iterator (the old Python prefetch thread leaked one thread + one shard image per evaluation)."""
    import threading
    from recsys_amd.input_pipeline import criteo_input_fn
    files, _ = _tagged_files(tmp_path, [200])
    before = threading.active_count()
    for _ in range(20):
        it = criteo_input_fn(files, 16, num_epochs=-1, need_shuffle=True, layout=layout, shuffle_buffer=4)
        for i, _b in enumerate(it):
            if i == 3:
                break
        it.close()
    assert threading.active_count() <= before          # C++ threads are not Python threads; and no Python thread is used
    import os
    n_threads = len(os.listdir("/proc/self/task"))
    assert n_threads < 64, n_threads                   # 20 readers x (1 scanner + 8 workers) would be 180 if leaked


def test_crc32c_hardware_and_table_agree():
    from recsys_amd import _lib
    import ctypes as C
    L = _lib.lib()
    rng = np.random.default_rng(5)
    assert L.rsx_crc32c_h(b"123456789", 9) == 0xE3069283 == L.rsx_crc32c_table_h(b"123456789", 9)
    for n in list(range(0, 40)) + [63, 64, 65, 801, 4096, 100003]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert L.rsx_crc32c_h(b, n) == L.rsx_crc32c_table_h(b, n), n


def test_din_reader_shards_and_keeps_order(tmp_path):
    from recsys_amd import synthetic
    from recsys_amd.input_pipeline import din_input_fn, write_din_shard
    b = synthetic.din_batch(np.random.default_rng(1), 50, P=12, n_item=300, n_cate=20)
    b["label"] = np.arange(50)
    p = tmp_path / "train2"
    write_din_shard(str(p), b)
    r0 = list(din_input_fn([str(p)], 8, num_epochs=1, hist_len=12, shard=(0, 2), ids_int32=True))        # default: complete rounds
    r1 = list(din_input_fn([str(p)], 8, num_epochs=1, hist_len=12, shard=(1, 2)))
    assert [g[1].tolist() for g in r0] == [list(range(0, 8)), list(range(16, 24)), list(range(32, 40))]
    assert [g[1].tolist() for g in r1] == [list(range(8, 16)), list(range(24, 32)), list(range(40, 48))]
    assert r0[0][0]["u_iid_seq"].dtype == np.int32 and r1[0][0]["u_iid_seq"].dtype == np.int64
    assert np.array_equal(r1[1][0]["u_iid_seq"], b["u_iid_seq"][24:32])
    # evaluation (shard_tail=True): the tail is delivered -- batch 6 (records 48, 49) belongs to rank 6 % 2 = 0
    e0 = list(din_input_fn([str(p)], 8, num_epochs=1, hist_len=12, shard=(0, 2), shard_tail=True))
    e1 = list(din_input_fn([str(p)], 8, num_epochs=1, hist_len=12, shard=(1, 2), shard_tail=True))
    assert [g[1].tolist() for g in e0] == [list(range(0, 8)), list(range(16, 24)), list(range(32, 40)), [48, 49]]
    assert [g[1].tolist() for g in e1] == [list(range(8, 16)), list(range(24, 32)), list(range(40, 48))]


def test_parse_serialized_examples_like_a_serving_request(layout):
    """serialized tf.train.Example strings as deepfm/grpc_client.py:50-76 builds them (no label feature)."""
    from recsys_amd._lib import RsxError
    from recsys_amd.input_pipeline import parse_criteo_examples, parse_din_examples
    label, cont, cat = _raw(9, seed=8)
    ser = []
    for r in range(9):
        ex = {"_c%d" % j: [float(cont[r, j - 1])] for j in range(1, 14)}
        ex.update({"_c%d" % j: [cat[r][j - 14]] for j in range(14, 40) if cat[r][j - 14] != b"NULL"})
        ser.append(tfrecord.encode_example(ex))
    feats, lab = parse_criteo_examples(ser, layout)
    assert np.array_equal(feats["ids"], criteo.transform_batch(cont, cat, c2_shift=4.0))
    assert np.array_equal(lab.reshape(-1), np.zeros(9, np.float32))
    with pytest.raises(RsxError):                      # a numeric feature is still required (FixedLenFeature, no default)
        parse_criteo_examples([tfrecord.encode_example({"_c1": [1.0]})], layout)
    with pytest.raises(RsxError):
        parse_criteo_examples([b"\xff\xff\xff garbage"], layout)
    ex = {"label": [1], "i_id": [7], "i_cate": [3], "u_iid_seq": [5, 6], "u_icat_seq": [2, 2]}
    f, y = parse_din_examples([tfrecord.encode_example(ex)], hist_len=4)
    assert f["u_iid_seq"].tolist() == [[5, 6, 0, 0]] and f["i_id"].tolist() == [7] and y.tolist() == [1]


def test_int64_key_columns_of_deepfm_py_as_committed(tmp_path):
    """deepfm/deepfm.py:28-51: u_id / i_id int64 features, categorical_column_with_hash_bucket(dtype=int64) = hash of the
    decimal string.  Host functions vs the oracle's FarmHash + codec."""
    import ctypes as C
    from oracle import hashing
    from recsys_amd._lib import RsxError, lib
    from recsys_amd.feature_columns import CriteoLayout, build_model_columns
    from recsys_amd.input_pipeline import uid_iid_input_fn
    lin, emb = build_model_columns(8)
    layout = CriteoLayout.from_columns(emb)
    assert [c.key for c in layout.columns] == ["i_id", "u_id"]            # input_layer sorts by column name (A-1)
    rng = np.random.default_rng(2)
    keys = np.concatenate([rng.integers(0, 1 << 40, 50), [0, 7, -5, 2 ** 62, -(2 ** 63)]]).astype(np.int64)
    out = np.empty(len(keys), np.int32)
    assert lib().rsx_hash_int64_keys_h(keys.ctypes.data_as(C.c_void_p), len(keys), 100000, out.ctypes.data_as(C.c_void_p)) == 0
    assert out.tolist() == [hashing.hash_bucket(str(int(k)).encode(), 100000) for k in keys]
    n = 23
    u, i, lab = rng.integers(1, 10 ** 9, n), rng.integers(1, 10 ** 6, n), rng.integers(0, 2, n)
    blob = b"".join(tfrecord.frame(tfrecord.encode_example({"label": [int(lab[r])], "u_id": [int(u[r])], "i_id": [int(i[r])]}))
                    for r in range(n))
    p = tmp_path / "part-r-00000"
    p.write_bytes(blob)
    got = list(uid_iid_input_fn([str(p)], 10, num_epochs=1, layout=layout))
    assert [g[1].shape[0] for g in got] == [10, 10, 3]
    ids = np.concatenate([g[0]["ids"] for g in got])
    assert ids[:, 0].tolist() == [hashing.hash_bucket(str(int(k)).encode(), 100000) for k in i]      # slot 0 = i_id
    assert ids[:, 1].tolist() == [hashing.hash_bucket(str(int(k)).encode(), 500000) for k in u]
    assert np.array_equal(np.concatenate([g[1] for g in got]), lab.astype(np.float32))
    bad = tmp_path / "bad"
    bad.write_bytes(tfrecord.frame(tfrecord.encode_example({"label": [1], "u_id": [3]})))       # i_id missing
    with pytest.raises(RsxError):
        list(uid_iid_input_fn([str(bad)], 4, num_epochs=1, layout=layout))


def test_packed_batch_adopts_the_readers_flat_buffer(tmp_path, layout):
    """The C++ reader assembles every batch in one flat host buffer laid out like estimator.PackedBatch's packing; the
    Estimator must wrap that buffer (no 40 KB copy per step) and see exactly the bytes a packed copy would hold.  Arrays that
    are not such views (copies, the last partial batch, torch tensors) take the copying path and give the same views."""
    from recsys_amd.estimator import PackedBatch
    from recsys_amd.input_pipeline import criteo_input_fn, write_criteo_shard
    p = str(tmp_path / "part-r-00000")
    label, cont, cat = _raw(100, seed=2)
    write_criteo_shard(p, label, cont, cat)
    batches = list(criteo_input_fn([p], 32, num_epochs=1, need_shuffle=False, layout=layout))
    assert [b[1].shape[0] for b in batches] == [32, 32, 32, 4]
    for feats, lab in batches:
        pb = PackedBatch(feats, lab)
        adopted = pb.flat.data_ptr() == lab.ctypes.data
        assert adopted == (lab.shape[0] == 32)                  # full batches are adopted, the partial one is copied
        ref = PackedBatch({k: v.copy() for k, v in feats.items()}, lab.copy())
        assert ref.flat.data_ptr() != lab.ctypes.data and ref.key() == pb.key() and ref.nbytes == pb.nbytes
        f1, l1 = pb.views()
        f2, l2 = ref.views()
        assert np.array_equal(l1.numpy(), lab) and np.array_equal(l2.numpy(), lab)
        for k in feats:
            assert np.array_equal(f1[k].numpy(), feats[k]) and np.array_equal(f2[k].numpy(), feats[k])
