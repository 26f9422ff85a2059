"""GPU box: field-sort timings, LDS path (one workgroup per field) vs the multi-workgroup global radix path."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from recsys_amd.ops import EmbeddingArena
from kernel_roofline_util import criteo_row_off, synth_ids
row_off = criteo_row_off()
rng = np.random.default_rng(0)
for B in (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    res = []
    for force_large in (False, True):
        if not force_large and B > 16384:
            res.append(float("nan")); continue
        EmbeddingArena.LDS_SORT_MAX_B = 0 if force_large else 16384
        a = EmbeddingArena(row_off, 16, max(B, 513), "cuda", with_w1=True)
        ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
        for _ in range(3): a.field_sort(ids)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): a.field_sort(ids)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1000 / 10)
        del a
    print("field_sort B=%6d   LDS %.1f us   global radix %.1f us" % (B, res[0], res[1]))
