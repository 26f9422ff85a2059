#!/usr/bin/env python
"""Host ingest throughput: TFRecord framing + CRC, Example parse + FarmHash + bucketize by thread
count, and the whole `criteo_input_fn` (batch -> shuffle -> prefetch) in records/s.  Host only, no GPU.
usage: python scripts/ingest_bench.py [n_records=200000] [batch=256]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from recsys_amd import input_pipeline as ip
from recsys_amd import synthetic
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    d = tempfile.mkdtemp(prefix="rsx_ingest_")
    rng = np.random.default_rng(1)
    files = []
    per = 50000
    for k in range((n + per - 1) // per):
        p = os.path.join(d, "part-r-%05d" % k)
        ip.write_criteo_shard(p, *synthetic.criteo_raw_batch(rng, min(per, n - k * per)))
        files.append(p)
    size = sum(os.path.getsize(f) for f in files)
    print("cores=%d  records=%d  bytes=%d (%.0f B/record)" % (os.cpu_count(), n, size, size / n), flush=True)
    layout = CriteoLayout.from_columns(build_feature_columns(16)[1])
    for verify in (0, 1):
        t = time.perf_counter()
        for f in files:
            ip.read_shard(f, verify_crc=bool(verify))
        dt = time.perf_counter() - t
        print("read + index, verify_crc=%d: %.3f s  %.2f M rec/s  %.2f GB/s" % (verify, dt, n / dt / 1e6, size / dt / 1e9), flush=True)
    shards = [ip.read_shard(f, verify_crc=False) for f in files]
    for th in (1, 2, 4, 8, 16):
        if th > 2 * (os.cpu_count() or 1):
            break
        parse = ip._CriteoParser(layout, th)
        t = time.perf_counter()
        for buf, offs, lens in shards:
            parse(buf, offs, lens)
        dt = time.perf_counter() - t
        print("parse threads=%2d: %.3f s  %.2f M rec/s" % (th, dt, n / dt / 1e6), flush=True)
    for th in (1, os.cpu_count() or 1):
        t = time.perf_counter()
        cnt = 0
        for feats, lab in ip.criteo_input_fn(files, bs, 1, True, th, layout):
            cnt += lab.shape[0]
        dt = time.perf_counter() - t
        print("criteo_input_fn(batch=%d, shuffle, threads=%d): %d records in %.3f s = %.3f M ex/s" % (bs, th, cnt, dt, cnt / dt / 1e6), flush=True)
    for f in files:
        os.remove(f)
    os.rmdir(d)


if __name__ == "__main__":
    main()
