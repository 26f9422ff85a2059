#!/usr/bin/env python
"""Where the host side of a batch goes: the C++ reader's `next` alone (one preallocated buffer), the Python generator around
it, the shuffle buffer around that.  usage: python scripts/ingest_layers.py [n_records=600000] [workers=32]"""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from recsys_amd import input_pipeline as ip
from recsys_amd import synthetic
from recsys_amd._lib import lib
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else min(32, os.cpu_count())
    bs = 256
    lin, emb = build_feature_columns(16, "indicator_all")
    layout = CriteoLayout.from_columns(emb)
    with tempfile.TemporaryDirectory() as d:
        rng = np.random.default_rng(0)
        files = []
        for k in range(4):
            label, cont, cat = synthetic.criteo_raw_batch(rng, n // 4)
            p = os.path.join(d, "part-r-%05d" % k)
            ip.write_criteo_shard(p, label, cont, cat)
            files.append(p)
        steps = 2 * (n // bs)
        print("cores=%d workers=%d records=%d batch=%d" % (os.cpu_count(), workers, n, bs), flush=True)
        for w in sorted({8, 16, workers, 64} & set(range(1, os.cpu_count() + 1))):
            for q in (16, 64):
                ps = ip._CriteoParser(layout, w)
                paths = ip._paths_array(files)
                _p = ip._p
                h = lib().rsx_criteo_reader_open_h(paths, len(files), _p(ps.slot_src), _p(ps.slot_rows), _p(ps.bnd), _p(ps.bnd_off),
                                                   _p(ps.shift), ps.F, bs, -1, 0, 1, 0, w, 1, q)
                rd = ip._Reader(h)
                F = ps.F
                o_cont = (bs * 4 + 15) & ~15
                o_ids = (o_cont + bs * 52 + 15) & ~15
                flat = np.empty(o_ids + bs * F * 4, np.uint8)
                a = flat.ctypes.data
                nx = lib().rsx_criteo_reader_next_h
                args = (rd.h, C.c_void_p(a), C.c_void_p(a + o_cont), C.c_void_p(a + o_ids))
                for _ in range(200):
                    nx(*args)
                t0 = time.time()
                for _ in range(steps):
                    nx(*args)
                dt = time.time() - t0
                rd.close()
                print("  reader next() alone, %2d workers, queue %2d: %.2f M examples/s (%.1f us per batch)"
                      % (w, q, steps * bs / dt / 1e6, dt / steps * 1e6), flush=True)
        for shuffle in (False, True):
            it = iter(ip.criteo_input_fn(files, bs, num_epochs=-1, need_shuffle=shuffle, layout=layout, num_parallel=workers))
            for _ in range(200):
                next(it)
            t0 = time.time()
            for _ in range(steps):
                next(it)
            dt = time.time() - t0
            it.close()
            print("  criteo_input_fn (shuffle=%s): %.2f M examples/s (%.1f us per batch)" % (shuffle, steps * bs / dt / 1e6, dt / steps * 1e6),
                  flush=True)


if __name__ == "__main__":
    main()
