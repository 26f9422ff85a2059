"""GPU box: host-side timeline of the streaming TRAIN loop (deepfm.py, batch 256, windows of 8): where a window's host time goes
(reader, packing, waiting for the staging buffer, enqueueing) against the GPU's time per window."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recsys_amd import deepfm, synthetic
from recsys_amd import input_pipeline as ip
from recsys_amd.estimator import Estimator, PackedBatch, RunConfig
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns

lin, emb = build_feature_columns(16, "indicator_all")
layout = CriteoLayout.from_columns(emb)
with tempfile.TemporaryDirectory() as d:
    rng = np.random.default_rng(0)
    files = []
    for k in range(2):
        label, cont, cat = synthetic.criteo_raw_batch(rng, 150000)
        p = os.path.join(d, "part-r-%05d" % k)
        ip.write_criteo_shard(p, label, cont, cat)
        files.append(p)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
              "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": 256}
    est = Estimator(deepfm.model_fn, None, params, RunConfig(device="cuda", seed=1, log_step_count_steps=1000000))
    workers = int(os.environ.get("WORKERS", "32"))
    fn = lambda: ip.criteo_input_fn(files, 256, num_epochs=-1, need_shuffle=True, layout=layout, num_parallel=workers)
    est.train(fn, steps=304)
    torch.cuda.synchronize()
    K = est._window_len()
    it = iter(fn())
    if os.environ.get("POOL"):            # no reader threads during the timed loop: a pool of batches fetched beforehand
        pool = []
        while len(pool) < 512:
            h = next(it)
            if h[1].shape[0] == 256:
                pool.append(h)
            it.close()
        it = iter(pool[i % 512] for i in range(10 ** 9))
    waits = []
    orig = torch.cuda.Event.synchronize

    def timed_sync(self):
        t = time.perf_counter()
        orig(self)
        waits.append(time.perf_counter() - t)

    torch.cuda.Event.synchronize = timed_sync
    gpu_ev = []
    orig_launch = est._launch_staged

    def timed_launch(st):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_launch(st)
        e1.record()
        gpu_ev.append((e0, e1))
        return r

    est._launch_staged = timed_launch
    T = {"read": 0.0, "pack": 0.0, "window": 0.0}
    nwin = 300
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    for w in range(nwin):
        t0 = time.perf_counter()
        held = []
        while len(held) < K:
            h = next(it)
            if h[1].shape[0] == 256:          # (the short last batch of an epoch is a step of its own in Estimator.train)
                held.append(h)
        t1 = time.perf_counter()
        pbs = [PackedBatch(*h) for h in held]
        t2 = time.perf_counter()
        est._train_window_packed(pbs)
        t3 = time.perf_counter()
        T["read"] += t1 - t0; T["pack"] += t2 - t1; T["window"] += t3 - t2
    t_host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print("windows of %d: host loop %.1f us per window (reader %.1f, PackedBatch %.1f, _train_window_packed %.1f of which waiting "
          "for the staging buffer %.1f), drained after %.1f us per window" %
          (K, t_host / nwin * 1e6, T["read"] / nwin * 1e6, T["pack"] / nwin * 1e6, T["window"] / nwin * 1e6,
           sum(waits) / nwin * 1e6, t_tot / nwin * 1e6))
    torch.cuda.Event.synchronize = orig
    durs = [a.elapsed_time(b) * 1e3 for a, b in gpu_ev[20:]]
    gaps = [gpu_ev[i][1].elapsed_time(gpu_ev[i + 1][0]) * 1e3 for i in range(20, len(gpu_ev) - 1)]
    span = gpu_ev[0][0].elapsed_time(gpu_ev[-1][1]) * 1e3
    print("GPU side: %d windows launched, first start -> last end %.1f us = %.1f us per window" % (len(gpu_ev), span, span / len(gpu_ev)))
    print("GPU side: window durations mean %.1f us, p90 %.1f, the 6 longest %s" % (float(np.mean(durs)), float(np.percentile(durs, 90)), [int(x) for x in sorted(durs)[-6:]]))
    print("GPU side: window %.1f us (median), idle between windows %.1f us (median; p90 %.1f; mean %.1f; the 6 longest: %s at windows %s)" %
          (float(np.median(durs)), float(np.median(gaps)), float(np.percentile(gaps, 90)), float(np.mean(gaps)),
           [int(x) for x in sorted(gaps)[-6:]], [int(i) + 20 for i in np.argsort(gaps)[-6:]]))
    getattr(it, 'close', lambda: None)()
