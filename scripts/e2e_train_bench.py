#!/usr/bin/env python
"""The whole product path end to end on the GPU box: TFRecord shards on disk -> C++ reader (framing, CRC-32C, Example parse,
FarmHash / bucketize, batching) -> Estimator.train (optimizer windows, one H2D copy per batch, HIP graphs) for deepfm.py and
fm.py at batch 256.  Prints examples/s of `Estimator.train` itself, input pipeline included.
usage: python scripts/e2e_train_bench.py [n_records=600000]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recsys_amd import deepfm, fm, synthetic
from recsys_amd import input_pipeline as ip
from recsys_amd.estimator import Estimator, RunConfig
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
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
        print("cores=%d  records=%d  batch=%d" % (os.cpu_count(), n, bs), flush=True)
        for name, mfn in (("deepfm", deepfm.model_fn), ("fm", fm.model_fn)):
            params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
                      "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": bs}
            est = Estimator(mfn, None, params, RunConfig(device="cuda", seed=1, log_step_count_steps=1000000))
            fn = lambda: ip.criteo_input_fn(files, bs, num_epochs=-1, need_shuffle=True, layout=layout,
                                            num_parallel=min(32, os.cpu_count()))
            est.train(fn, steps=300)                       # build, warm-up, graph captures
            torch.cuda.synchronize()
            # every Estimator.train call starts its input pipeline afresh (reader threads, mmap, and the 1 000-batch shuffle
            # buffer has to fill before the first batch comes out): timed separately, and the runs are long enough to amortise it
            t0 = time.time()
            it = iter(fn())
            next(it)
            t_start = time.time() - t0
            it.close()
            print("%-7s input pipeline start-up (threads + shuffle buffer of 1000 batches) until the first batch: %.1f ms" % (name, t_start * 1e3), flush=True)
            steps = 4 * ((n // bs) // 8 * 8)
            for rep in range(3):
                t0 = time.time()
                est.train(fn, steps=steps)
                torch.cuda.synchronize()
                dt = time.time() - t0
                print("%-7s Estimator.train over TFRecord shards: %d steps in %.2f s = %.3f M examples/s (%.1f us per step; %.1f without the start-up)"
                      % (name, steps, dt, steps * bs / dt / 1e6, dt / steps * 1e6, (dt - t_start) / steps * 1e6), flush=True)
            # the same input pipeline alone (no training): what the host side can deliver
            t0 = time.time()
            it = iter(fn())
            for _ in range(steps):
                next(it)
            dt = time.time() - t0
            it.close()
            print("        input pipeline alone: %.3f M examples/s (%.1f us per batch)" % (steps * bs / dt / 1e6, dt / steps * 1e6), flush=True)


if __name__ == "__main__":
    main()
