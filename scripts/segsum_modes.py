"""GPU box: single-stage vs two-stage segment-sum (+ the sort) at a few batch sizes (HIP events over graph replays)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from recsys_amd.ops import AdamTF1, EmbeddingArena
from scripts.kernel_roofline_util import timeit
from kernel_roofline_util import criteo_row_off, synth_ids

row_off = criteo_row_off()
rng = np.random.default_rng(0)
for B in (512, 1024, 2048, 4096):
    ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
    dX = torch.randn(B, 39 * 16, device="cuda"); g1 = torch.randn(B, device="cuda"); g2 = torch.randn(B, device="cuda")
    out = []
    for mode, thr in (("two-stage", 512), ("single", 1 << 30)):
        EmbeddingArena.TWO_STAGE_MIN_B = thr
        a = EmbeddingArena(row_off, 16, B, "cuda", with_w1=True)
        a.tables.normal_(); a.w1.normal_()
        E, S, _, _ = a.gather(ids, fm=True, first_order=True)
        opt = AdamTF1(device="cuda")
        a.field_sort(ids)
        t_sort = timeit(lambda: a.field_sort(ids))
        t_seg = timeit(lambda: a.segsum(B, S, dX, g1, g2))
        t_sa = timeit(lambda: a.segsum_adam(B, S, dX, g1, g2, opt, [], None))
        out.append("%s: sort %.1f segsum %.1f segsum_adam %.1f us" % (mode, t_sort, t_seg, t_sa))
    print("B=%d  " % B + " | ".join(out), flush=True)
