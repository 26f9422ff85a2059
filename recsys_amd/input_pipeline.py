"""Input pipeline: the reference's `input_fn` (fm/fm.py:106-112, deepfm/deepfm.py:60-70, xdeepfm/xdeepfm.py:101-118,
dcn/dcn.py:106-112, din/din.py:61-80) on top of librsx.so's streaming TFRecord reader (csrc/tfrecord_reader.cpp: mmap, SSE4.2 CRC, parse and batch assembly in
C++ worker threads, data-parallel sharding of the batch stream).

Order of transformations is the reference's:
    TFRecordDataset(files) -> map(parse) -> batch(bs) -> [shuffle(buffer of BATCHES)] -> prefetch -> repeat(epochs)
i.e. shuffling permutes whole batches inside a sliding window and the final partial batch of an epoch is kept.
The host half of `input_layer` (FarmHash % bucket, bucketize(log(x+shift))) runs inside the C++ parse, so the
pipeline yields `features = {'ids': int32 [B,39], 'cont_log': float32 [B,13]}` and `labels` float32 [B,1].
"""
import ctypes as C
import queue
import threading

import numpy as np

from ._lib import RsxError, check, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def read_shard(path, verify_crc=True):
    """-> (buf uint8, offsets int64 [n], lengths int64 [n]).  Raises RsxError on a corrupt record (TF: DataLossError)."""
    buf = np.fromfile(path, dtype=np.uint8)
    L = lib()
    n = L.rsx_tfrecord_index_h(_p(buf), buf.size, None, None, 0, 0)
    if n < 0:
        raise RsxError("%s: %s" % (path, L.rsx_strerror(int(n)).decode()))
    offs, lens = np.empty(n, np.int64), np.empty(n, np.int64)
    n2 = L.rsx_tfrecord_index_h(_p(buf), buf.size, _p(offs), _p(lens), n, int(verify_crc))
    if n2 < 0:
        raise RsxError("%s: %s" % (path, L.rsx_strerror(int(n2)).decode()))
    return buf, offs, lens


class _CriteoParser:
    def __init__(self, layout, threads, label_optional=False):
        self.F = layout.F
        self.threads = int(threads) | (0x10000 if label_optional else 0)    # bit 16: a missing `_c0` parses as label 0
        self.slot_src = np.array([int(c.key[2:]) for c in layout.columns], np.int32)
        self.slot_rows = np.array([c.rows for c in layout.columns], np.int32)
        bnd, off = [], [0]
        for c in layout.columns:
            bnd += list(c.boundaries or [])
            off.append(len(bnd))
        self.bnd = np.array(bnd if bnd else [0.0], np.float32)
        self.bnd_off = np.array(off, np.int32)
        shift = np.ones(13, np.float32)
        for c in layout.columns:
            if c.boundaries is not None:
                shift[int(c.key[2:]) - 1] = c.log_shift
        self.shift = shift

    def __call__(self, buf, offs, lens):
        n = len(offs)
        label = np.empty((n, 1), np.float32)
        cont = np.empty((n, 13), np.float32)
        ids = np.empty((n, self.F), np.int32)
        check(lib().rsx_criteo_parse_h(_p(buf), _p(offs), _p(lens), n, _p(self.slot_src), _p(self.slot_rows), _p(self.bnd),
                                       _p(self.bnd_off), _p(self.shift), self.F, _p(label), _p(cont), _p(ids), self.threads),
              "rsx_criteo_parse_h")
        return {"ids": ids, "cont_log": cont}, label


def _batched(parse, filenames, batch_size, num_epochs, chunk_records=65536):
    """map(parse) -> batch: record streams of consecutive files are concatenated before batching, like TFRecordDataset."""
    epoch = 0
    while num_epochs < 0 or epoch < num_epochs:
        carry_f, carry_l = None, None
        for path in filenames:
            buf, offs, lens = read_shard(path)
            for s in range(0, len(offs), chunk_records):
                feats, lab = parse(buf, offs[s:s + chunk_records], lens[s:s + chunk_records])
                if carry_l is not None:
                    feats = {k: np.concatenate([carry_f[k], v]) for k, v in feats.items()}
                    lab = np.concatenate([carry_l, lab])
                n = lab.shape[0]
                full = (n // batch_size) * batch_size
                for b in range(0, full, batch_size):
                    yield {k: v[b:b + batch_size] for k, v in feats.items()}, lab[b:b + batch_size]
                carry_f = {k: v[full:] for k, v in feats.items()} if full < n else None
                carry_l = lab[full:] if full < n else None
        if carry_l is not None and carry_l.shape[0]:
            yield carry_f, carry_l                      # the final partial batch is kept
        epoch += 1


def _shuffled(it, buffer_size, seed):
    """tf.data shuffle(buffer_size) applied AFTER batch: a sliding window of whole batches."""
    rng = np.random.default_rng(seed)
    buf = []
    for x in it:
        if len(buf) < buffer_size:
            buf.append(x)
            continue
        i = int(rng.integers(0, buffer_size))
        out, buf[i] = buf[i], x
        yield out
    rng.shuffle(buf)
    yield from buf


def _prefetched(it, depth):
    """prefetch(buffer_size): a producer thread runs `it` up to `depth` batches ahead.  When the consumer stops early
    (evaluate(steps=...), predict's caller breaking out, garbage collection of the generator) the `finally` below sets
    the stop event; the producer notices within one put-timeout, closes the upstream generator (releasing the shard
    image it holds) and exits -- no thread or buffer outlives its consumer."""
    q = queue.Queue(maxsize=max(1, depth))
    end = object()
    stop = threading.Event()

    def put(x):
        while not stop.is_set():
            try:
                q.put(x, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for x in it:
                if not put(x):
                    break
            else:
                put(end)
        except BaseException as e:  # surfaced in the consumer
            put(e)
        finally:
            close = getattr(it, "close", None)
            if close is not None:
                close()

    t = threading.Thread(target=work, daemon=True, name="rsx-prefetch")
    t.start()
    try:
        while True:
            x = q.get()
            if x is end:
                return
            if isinstance(x, BaseException):
                raise x
            yield x
    finally:
        stop.set()
        t.join(timeout=5.0)


def _shard_from_env(shard):
    """(rank, world) of the data-parallel job this process belongs to; explicit `shard` wins."""
    if shard is not None:
        return int(shard[0]), int(shard[1])
    return 0, 1


class _Reader:
    """Owner of one C++ streaming reader (csrc/tfrecord_reader.cpp); closed by the generator's `finally`, i.e. also when
    the consumer stops early or drops the iterator -- no thread or mapping outlives its consumer."""

    def __init__(self, handle):
        if not handle:
            raise RsxError("could not open the TFRecord reader (bad arguments)")
        self.h = C.c_void_p(handle)

    def close(self):
        if self.h is not None:
            lib().rsx_reader_close_h(self.h)
            self.h = None

    def __del__(self):
        self.close()


def _paths_array(filenames):
    arr = (C.c_char_p * len(filenames))(*[str(f).encode() for f in filenames])
    return arr


class FlatFeatures(dict):
    """The features dict of one batch whose arrays (and labels) are views of ONE flat host buffer laid out like
    estimator.PackedBatch's packing: `flat` is that buffer (np.uint8), so the Estimator wraps it instead of re-packing."""
    __slots__ = ("flat", "pack_key")       # pack_key: one hashable per stream layout (batch size, fields)


def _criteo_batches(filenames, batch_size, num_epochs, layout, threads, shard, verify_crc, queue_batches, drop_remainder):
    ps = _CriteoParser(layout, threads)
    rank, world = shard
    paths = _paths_array(filenames)
    rd = _Reader(lib().rsx_criteo_reader_open_h(paths, len(filenames), _p(ps.slot_src), _p(ps.slot_rows), _p(ps.bnd),
                                                _p(ps.bnd_off), _p(ps.shift), ps.F, int(batch_size), int(num_epochs), rank,
                                                world, int(drop_remainder), int(threads), int(verify_crc), int(queue_batches)))
    F, bs = ps.F, int(batch_size)
    # ONE flat host buffer per batch, laid out like estimator.PackedBatch packs it (label | cont_log | ids, 16-byte aligned
    # parts), so the Estimator's single H2D copy can take it as it is (PackedBatch adopts it without a copy).  The parts are
    # handed to the reader by ADDRESS (one .ctypes.data per batch instead of three data_as casts: the Python side of a batch
    # cost 70 us, more than a DeepFM step).
    o_cont = (bs * 4 + 15) & ~15
    o_ids = (o_cont + bs * 52 + 15) & ~15
    nb = o_ids + bs * F * 4
    next_h = lib().rsx_criteo_reader_next_h
    VP = C.c_void_p
    pack_key = ("criteo", bs, F)
    try:
        while True:
            flat = np.empty(nb, np.uint8)
            a = flat.ctypes.data
            n = next_h(rd.h, VP(a), VP(a + o_cont), VP(a + o_ids))
            if n == 0:
                return
            if n < 0:
                raise RsxError("%s: %s" % (filenames[0] if len(filenames) == 1 else "TFRecord stream",
                                           lib().rsx_strerror(int(n)).decode()))
            label = flat[:bs * 4].view(np.float32).reshape(bs, 1)
            cont = flat[o_cont:o_cont + bs * 52].view(np.float32).reshape(bs, 13)
            ids = flat[o_ids:o_ids + bs * F * 4].view(np.int32).reshape(bs, F)
            feats = FlatFeatures(ids=ids, cont_log=cont)
            feats.flat, feats.pack_key = flat, pack_key
            if n < bs:
                label = label[:n]
                feats = {"ids": ids[:n], "cont_log": cont[:n]}
            yield feats, label
    finally:
        rd.close()


RSX_SHARD_TAIL = 2      # include/rsx.h: value of drop_remainder


def _shard_tail(shard_tail, num_epochs, need_shuffle):
    """Sharded EVALUATION has no collective inside its loop, so the ranks need not run equal step counts: with shard_tail=True
    (what run_main's eval input_fns pass) the leftover batches of the last round and the final partial batch are delivered
    too (csrc/tfrecord_reader.cpp RSX_SHARD_TAIL) -- together the ranks then evaluate exactly the examples a single replica
    would.  An explicit opt-in (default False): a data-parallel TRAIN stream that happened to be built with
    num_epochs=1 and no shuffle must keep complete rounds only, or its ranks run unequal step counts and deadlock in the
    step's collectives."""
    return RSX_SHARD_TAIL if shard_tail else 0


def criteo_input_fn(filenames, batch_size, num_epochs=-1, need_shuffle=False, num_parallel=8, layout=None,
                    shuffle_buffer=1000, prefetch=16, seed=0, shard=None, verify_crc=True, shard_tail=False):
    """fm/fm.py:106-112.  Returns an iterator of (features, labels).
    shard=(rank, world): this replica's share of the batch stream (see csrc/tfrecord_reader.cpp); each replica then
    shuffles its own sub-stream with its own seed.  `prefetch` = batches the C++ reader keeps in flight.
    shard_tail (default False; evaluation passes True): see _shard_tail."""
    if layout is None:
        from .feature_columns import CriteoLayout, build_feature_columns
        layout = CriteoLayout.from_columns(build_feature_columns(16)[1])
    rank, world = _shard_from_env(shard)
    it = _criteo_batches(list(filenames), batch_size, num_epochs, layout, num_parallel, (rank, world), verify_crc,
                         max(2, prefetch), drop_remainder=_shard_tail(shard_tail, num_epochs, need_shuffle) if world > 1 else 0)
    if need_shuffle:
        it = _shuffled(it, shuffle_buffer, seed + 7919 * rank)
    return it


def din_input_fn(filenames, batch_size, num_epochs=-1, need_shuffle=False, num_parallel=6, hist_len=100,
                 shuffle_buffer=1000, prefetch=16, seed=0, ids_int32=False, shard=None, verify_crc=True, shard_tail=False):
    """din/din.py:61-80: features {'i_id','i_cate' int64 [B]; 'u_iid_seq','u_icat_seq' int64 [B,P]}, labels int64 [B].
    ids_int32: narrow the id features to int32 on the HOST (the device kernels index with int32; like the Criteo parse,
    which emits int32 row ids) so that the training step has no per-feature cast launches."""
    P, bs = int(hist_len), int(batch_size)
    rank, world = _shard_from_env(shard)
    filenames = list(filenames)

    def gen():
        paths = _paths_array(filenames)
        tail = _shard_tail(shard_tail, num_epochs, need_shuffle) if world > 1 else 0
        rd = _Reader(lib().rsx_din_reader_open_h(paths, len(filenames), P, bs, int(num_epochs), rank, world, tail,
                                                 int(num_parallel), int(verify_crc), max(2, int(prefetch))))
        try:
            while True:
                lab, iid, icat = (np.empty(bs, np.int64) for _ in range(3))
                hi, hc = np.empty((bs, P), np.int64), np.empty((bs, P), np.int64)
                n = lib().rsx_din_reader_next_h(rd.h, _p(lab), _p(iid), _p(icat), _p(hi), _p(hc))
                if n == 0:
                    return
                if n < 0:
                    raise RsxError("DIN TFRecord stream: %s" % lib().rsx_strerror(int(n)).decode())
                if n < bs:
                    lab, iid, icat, hi, hc = lab[:n], iid[:n], icat[:n], hi[:n], hc[:n]
                # (id 0 is the history padding (din/din.py:56-57,107) AND an ordinary row for the target lookups, as in the
                # reference: the device side keys the padding entries to a dummy row, so a target id 0 trains normally)
                # Negative ids would index in front of the tables (tf.gather raises InvalidArgument on CPU): refuse them here.
                if min(int(iid.min()), int(icat.min()), int(hi.min()), int(hc.min())) < 0:
                    raise RsxError("DIN TFRecord stream: negative id in i_id / i_cate / u_iid_seq / u_icat_seq")
                if ids_int32:
                    iid, icat, hi, hc = (x.astype(np.int32) for x in (iid, icat, hi, hc))
                yield {"i_id": iid, "i_cate": icat, "u_iid_seq": hi, "u_icat_seq": hc}, lab
        finally:
            rd.close()

    it = gen()
    if need_shuffle:
        it = _shuffled(it, shuffle_buffer, seed + 7919 * rank)
    return it


def uid_iid_input_fn(filenames, batch_size=32, num_epochs=-1, need_shuffle=False, layout=None, shuffle_buffer=100, seed=0):
    """deepfm/deepfm.py:53-70 AS COMMITTED: records {label int64, u_id int64, i_id int64} -> features {'ids' int32 [B,2]}
    (hashed on the host like the Criteo categorical fields), labels float32 [B].  Small utility path (the BASELINE configs
    are the Criteo marriage): whole shards are indexed and parsed at once."""
    names = (C.c_char_p * 3)(b"label", b"u_id", b"i_id")

    def gen():
        epoch = 0
        while num_epochs < 0 or epoch < num_epochs:
            carry = None
            for path in filenames:
                buf, offs, lens = read_shard(path)
                n = len(offs)
                out = np.empty((3, n), np.int64)
                check(lib().rsx_int64_features_parse_h(_p(buf), _p(offs), _p(lens), n, names, 3, _p(out), 4),
                      "rsx_int64_features_parse_h")
                if carry is not None:
                    out = np.concatenate([carry, out], 1)
                full = (out.shape[1] // batch_size) * batch_size
                for b in range(0, full, batch_size):
                    sl = out[:, b:b + batch_size]
                    yield {"ids": layout.transform_int64({"u_id": sl[1], "i_id": sl[2]})}, sl[0].astype(np.float32)
                carry = out[:, full:] if full < out.shape[1] else None
            if carry is not None and carry.shape[1]:
                yield {"ids": layout.transform_int64({"u_id": carry[1], "i_id": carry[2]})}, carry[0].astype(np.float32)
            epoch += 1

    it = gen()
    if need_shuffle:
        it = _shuffled(it, shuffle_buffer, seed)
    return it


# ---- serialized tf.train.Example entry -------------------------------------------------------------
def _pack_serialized(serialized):
    """list of bytes -> (buf uint8, offsets int64 [n], lengths int64 [n])."""
    lens = np.fromiter((len(x) for x in serialized), np.int64, len(serialized))
    offs = np.zeros(len(serialized), np.int64)
    if len(serialized) > 1:
        np.cumsum(lens[:-1], out=offs[1:])
    buf = np.frombuffer(b"".join(serialized), np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    return buf, offs, lens


def parse_criteo_examples(serialized, layout, threads=1):
    """What the exported model's parsing signature does with the strings a serving client sends
    (deepfm/grpc_client.py:50-76 builds one serialized tf.train.Example per request row; the SavedModel parses them with
    the script's feature_description, deepfm/deepfm.py:213-233): -> (features, labels) like one input_fn batch.
    The label feature `_c0` is optional here (serving requests carry none)."""
    if not len(serialized):
        raise ValueError("no examples")
    buf, offs, lens = _pack_serialized(serialized)
    feats, label = _CriteoParser(layout, threads, label_optional=True)(buf, offs, lens)
    return feats, label


def parse_din_examples(serialized, hist_len=100, ids_int32=True):
    """din/din.py:44-57 parse spec applied to in-memory serialized Examples."""
    if not len(serialized):
        raise ValueError("no examples")
    buf, offs, lens = _pack_serialized(serialized)
    n, P = len(serialized), int(hist_len)
    lab, iid, icat = (np.empty(n, np.int64) for _ in range(3))
    hi, hc = np.empty((n, P), np.int64), np.empty((n, P), np.int64)
    check(lib().rsx_din_parse_h(_p(buf), _p(offs), _p(lens), n, P, _p(lab), _p(iid), _p(icat), _p(hi), _p(hc), 1),
          "rsx_din_parse_h")
    if ids_int32:
        iid, icat, hi, hc = (x.astype(np.int32) for x in (iid, icat, hi, hc))
    return {"i_id": iid, "i_cate": icat, "u_iid_seq": hi, "u_icat_seq": hc}, lab


# ---- synthetic shard writers (the reference's sample shard is a missing blob) ------------
def write_criteo_shard(path, label, cont, cat):
    """label [n] f32, cont [n,13] f32, cat: n lists of 26 bytes values ('NULL' entries are omitted on disk)."""
    from .synthetic import pack_cat
    label = np.ascontiguousarray(label, np.float32).reshape(-1)
    cont = np.ascontiguousarray(cont, np.float32)
    buf, offs = pack_cat(cat)
    L = lib()
    need = L.rsx_criteo_encode_h(_p(label), _p(cont), _p(buf), _p(offs), len(label), None, 0)
    cap = -int(need)
    out = np.empty(cap, np.uint8)
    n = L.rsx_criteo_encode_h(_p(label), _p(cont), _p(buf), _p(offs), len(label), _p(out), cap)
    if n < 0:
        raise RsxError("rsx_criteo_encode_h failed (%d)" % n)
    out[:n].tofile(path)
    return int(n)


def write_din_shard(path, batch, keep_padding=False):
    lab, iid, icat = (np.ascontiguousarray(batch[k], np.int64) for k in ("label", "i_id", "i_cate"))
    hi, hc = np.ascontiguousarray(batch["u_iid_seq"], np.int64), np.ascontiguousarray(batch["u_icat_seq"], np.int64)
    L = lib()
    n_ex, P = hi.shape
    need = L.rsx_din_encode_h(_p(lab), _p(iid), _p(icat), _p(hi), _p(hc), n_ex, P, int(keep_padding), None, 0)
    cap = -int(need)
    out = np.empty(cap, np.uint8)
    n = L.rsx_din_encode_h(_p(lab), _p(iid), _p(icat), _p(hi), _p(hc), n_ex, P, int(keep_padding), _p(out), cap)
    if n < 0:
        raise RsxError("rsx_din_encode_h failed (%d)" % n)
    out[:n].tofile(path)
    return int(n)
