"""Oracle: TFRecord framing + tf.train.Example codec, pure Python.

Follows: the file format `tf.data.TFRecordDataset` / `tf.parse_single_example`
read at fm/fm.py:100-112 (twins deepfm/deepfm.py:54-70, xdeepfm/xdeepfm.py:95-118,
dcn/dcn.py:100-112, din/din.py:52-80) and that xdeepfm/gen_tfrecords.py:31-40
writes through spark-tensorflow-connector.
Pinned by the CRC-32C / framing KATs of SURVEY.md Appendix B-2.

Test infrastructure only (see oracle/__init__.py).
"""
import struct

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def frame(payload: bytes) -> bytes:
    hdr = struct.pack("<Q", len(payload))
    return hdr + struct.pack("<I", masked_crc(hdr)) + payload + struct.pack("<I", masked_crc(payload))


def unframe(buf: bytes):
    """Yield payloads; raises ValueError on a corrupt record (like TF's DataLossError)."""
    p = 0
    while p < len(buf):
        if p + 12 > len(buf):
            raise ValueError("truncated record header")
        hdr = buf[p:p + 8]
        (n,) = struct.unpack("<Q", hdr)
        if struct.unpack_from("<I", buf, p + 8)[0] != masked_crc(hdr):
            raise ValueError("corrupted record length crc")
        if p + 12 + n + 4 > len(buf):
            raise ValueError("truncated record")
        payload = buf[p + 12:p + 12 + n]
        if struct.unpack_from("<I", buf, p + 12 + n)[0] != masked_crc(payload):
            raise ValueError("corrupted record data crc")
        yield payload
        p += 16 + n


# ---- minimal protobuf wire codec for tf.train.Example -----------------------
def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _rd_varint(buf, p):
    v = 0
    s = 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, p
        s += 7


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features: dict) -> bytes:
    """features: name -> bytes | list[bytes] | list[float] (np.float32) | list[int].

    Floats/ints are written packed (like TF / the Spark connector)."""
    entries = b""
    for name in features:
        val = features[name]
        if isinstance(val, (bytes, str)):
            val = [val]
        val = list(val)
        if len(val) and isinstance(val[0], (bytes, str)):
            inner = b"".join(_ld(1, v.encode() if isinstance(v, str) else v) for v in val)
            feat = _ld(1, inner)                      # bytes_list = 1
        elif len(val) and isinstance(val[0], float):
            inner = _ld(1, struct.pack("<%df" % len(val), *val))
            feat = _ld(2, inner)                      # float_list = 2
        else:
            inner = _ld(1, b"".join(_varint(int(v)) for v in val))
            feat = _ld(3, inner)                      # int64_list = 3
        entry = _ld(1, name.encode()) + _ld(2, feat)  # map entry: key=1, value=2
        entries += _ld(1, entry)                      # Features.feature = 1
    return _ld(1, entries)                            # Example.features = 1


def decode_example(buf: bytes) -> dict:
    """Inverse of encode_example; accepts packed and unpacked repeated scalars."""
    out = {}

    def fields(b):
        p = 0
        while p < len(b):
            tag, p = _rd_varint(b, p)
            wt = tag & 7
            if wt == 2:
                n, p = _rd_varint(b, p)
                yield tag >> 3, wt, b[p:p + n]
                p += n
            elif wt == 0:
                v, p = _rd_varint(b, p)
                yield tag >> 3, wt, v
            elif wt == 5:
                yield tag >> 3, wt, b[p:p + 4]
                p += 4
            elif wt == 1:
                yield tag >> 3, wt, b[p:p + 8]
                p += 8
            else:
                raise ValueError("unsupported wire type %d" % wt)

    for fno, _, feats in fields(buf):
        if fno != 1:
            continue
        for eno, _, entry in fields(feats):
            if eno != 1:
                continue
            key, feat = None, b""
            for kno, _, v in fields(entry):
                if kno == 1:
                    key = v.decode()
                elif kno == 2:
                    feat = v
            vals = []
            for kind, _, lst in fields(feat):
                for vno, wt, v in fields(lst):
                    if vno != 1:
                        continue
                    if kind == 1:
                        vals.append(bytes(v))
                    elif kind == 2:
                        if wt == 2:
                            vals.extend(struct.unpack("<%df" % (len(v) // 4), v))
                        else:
                            vals.append(struct.unpack("<f", v)[0])
                    elif kind == 3:
                        if wt == 2:
                            p = 0
                            while p < len(v):
                                x, p = _rd_varint(v, p)
                                vals.append(x - (1 << 64) if x >> 63 else x)
                        else:
                            vals.append(v - (1 << 64) if v >> 63 else v)
            out[key] = vals
    return out
