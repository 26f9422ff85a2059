"""GPU box: what the fused scatter + Adam launch (segsum_adam_k) costs by ingredient, at the batch sizes of the DeepFM step
(256 single, 2048 = 8 emulated ranks): dX only / + FM term / + first-order vector / + dense arena / + window pass.
usage: scatter_detail.py [B ...]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from recsys_amd.ops import AdamTF1, DenseArena, EmbeddingArena
from scripts.kernel_roofline_util import timeit
from kernel_roofline_util import criteo_row_off, synth_ids

row_off = criteo_row_off()
Bs = [int(x) for x in sys.argv[1:]] or [256, 2048]
for B in Bs:
    rng = np.random.default_rng(0)
    for w1 in (False, True):
        a = EmbeddingArena(row_off, 16, B, "cuda", with_w1=w1, w1_field_mask=(1 << 39) - 1)
        with torch.no_grad():
            a.tables.normal_()
            if w1:
                a.w1.normal_()
        nb = min(8 if B <= 1024 else 4, len(a.sortbufs))
        ids = [torch.from_numpy(synth_ids(rng, B, row_off)).cuda() for _ in range(nb)]
        dX = torch.randn(B, 39 * 16, device="cuda"); g1 = torch.randn(B, device="cuda"); g2 = torch.randn(B, device="cuda")
        E, S, _, _ = a.gather(ids[0], fm=True, first_order=w1)
        opt = AdamTF1(device="cuda")
        dense = DenseArena({"w": (73100,)}, "cuda")
        a.sort_window(ids[:nb])
        a.select(0)
        opt.window_sweep(a.adam_split_segments(window_k=nb)[0])     # (leaves the window's step sizes in the optimizer state)
        win = (nb, 0)
        res = {}
        if not w1:
            res["dX only, no window"] = timeit(lambda: a.segsum_adam(B, None, dX, None, None, opt, [], None))
            res["dX only + window pass"] = timeit(lambda: a.segsum_adam(B, None, dX, None, None, opt, [], None, window=win))
            res["dX + FM + window pass"] = timeit(lambda: a.segsum_adam(B, S, dX, None, g2, opt, [], None, window=win))
            res["dX + FM + dense + window pass"] = timeit(lambda: a.segsum_adam(B, S, dX, None, g2, opt, dense.adam_segments(), None, window=win))
        else:
            res["dX + FM + w1, no window (w1 swept in the launch)"] = timeit(lambda: a.segsum_adam(B, S, dX, g1, g2, opt, [], None))
            res["dX + FM + w1 + window pass"] = timeit(lambda: a.segsum_adam(B, S, dX, g1, g2, opt, [], None, window=win))
            res["dX + FM + w1 + dense + window pass (the DeepFM launch)"] = timeit(lambda: a.segsum_adam(B, S, dX, g1, g2, opt, dense.adam_segments(), None, window=win))
            for pos in (nb // 2, nb - 1):      # a middle position and the window's LAST step (the lazy pass finishes every earlier list there)
                a.select(pos)
                res["the same at window position %d of %d" % (pos, nb)] = timeit(
                    lambda: a.segsum_adam(B, S, dX, g1, g2, opt, dense.adam_segments(), None, window=(nb, pos)))
            a.select(0)
            if B > 1024:
                res["stage A alone (FM + w1)"] = timeit(lambda: a._stage_a(B, S, dX, g1, g2))
        for k, v in res.items():
            print("B=%d window=%d  %-60s %6.1f us" % (B, nb, k, v), flush=True)
        del a
        torch.cuda.empty_cache()
