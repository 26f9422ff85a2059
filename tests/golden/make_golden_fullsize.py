#!/usr/bin/env python
"""Generates tests/golden/full_<config>.npz: the fp64 oracle's outputs for every BASELINE.json config AT ITS BATCH SIZE
(deepfm bs 256, xdeepfm bs 256 CIN 128-128, dcn bs 4096, din bs 1024 hist 100 K 32), two training steps each.
Inputs (54 MB of tables, ids) are NOT stored: both sides regenerate them from the seed (tests/fullsize.make_inputs);
the fixture pins them with a sha256 digest and stores only expected outputs (data, no code).

usage: python tests/golden/make_golden_fullsize.py [config ...]       (about two minutes in the build container)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests import fullsize  # noqa: E402


def main():
    names = sys.argv[1:] or list(fullsize.CONFIGS)
    for name in names:
        t0 = time.time()
        P, batches, digest = fullsize.make_inputs(name)
        out = fullsize.oracle_run(name, P, batches)
        path = os.path.join(HERE, "full_%s.npz" % name)
        np.savez_compressed(path, digest=np.array(digest), **out)
        print("%s: %.1f s, %d KB, losses %s" % (name, time.time() - t0, os.path.getsize(path) // 1024, out["losses"]), flush=True)


if __name__ == "__main__":
    main()
