"""GPU box: the large-batch scatter in parts (HIP events over graph replays): dedup sort, stage A (tiles), stage A + B
(segment-sum into G), the fused scatter + touched-row Adam, with and without the FM term, + the window pass.
usage: scatter_large.py [B ...]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from recsys_amd.ops import AdamTF1, DenseArena, EmbeddingArena
from scripts.kernel_roofline_util import timeit
from kernel_roofline_util import criteo_row_off, synth_ids

row_off = criteo_row_off()
Bs = [int(x) for x in sys.argv[1:]] or [2048, 4096, 16384, 65536]
for B in Bs:
    rng = np.random.default_rng(0)
    a = EmbeddingArena(row_off, 16, B, "cuda", with_w1=True, w1_field_mask=(1 << 39) - 1)
    with torch.no_grad():
        a.tables.normal_(); a.w1.normal_()
    nb = min(4, len(a.sortbufs))
    ids = [torch.from_numpy(synth_ids(rng, B, row_off)).cuda() for _ in range(nb)]
    dX = torch.randn(B, 39 * 16, device="cuda"); g1 = torch.randn(B, device="cuda"); g2 = torch.randn(B, device="cuda")
    E, S, _, _ = a.gather(ids[0], fm=True, first_order=True)
    opt = AdamTF1(device="cuda")
    dense = DenseArena({"w": (73100,)}, "cuda")
    a.field_sort(ids[0])
    nu = int(a.nuniq.sum().item())
    alg = B * 39 * (4 + 64) + nu * 64                    # scatter bwd bytes
    t_sort = timeit(lambda: a.field_sort(ids[0]))
    t_a = timeit(lambda: a._stage_a(B, None, dX, None, None))
    t_ab = timeit(lambda: a.segsum(B, None, dX, None, None))
    t_ab_fm = timeit(lambda: a.segsum(B, S, dX, g1, g2))
    t_sa = timeit(lambda: a.segsum_adam(B, None, dX, None, None, opt, [], None))
    t_sa_fm = timeit(lambda: a.segsum_adam(B, S, dX, g1, g2, opt, dense.adam_segments(), None))
    line = ("B=%d unique=%d  sort %.1f | stage A %.1f | A+B (dX only) %.1f us = %.1f%% of HBM | A+B (FM + first order) %.1f | "
            "scatter+Adam (dX only) %.1f | scatter+Adam (FM, w1, dense) %.1f" %
            (B, nu, t_sort, t_a, t_ab, alg / (t_ab * 1e-6) / 8e12 * 100, t_ab_fm, t_sa, t_sa_fm))
    if nb > 1 and B <= EmbeddingArena.LDS_SORT_MAX_B * 2:
        a.sort_window(ids[:nb])
        a.select(0)
        t_w = timeit(lambda: a.segsum_adam(B, None, dX, None, None, opt, dense.adam_segments(), None, window=(nb, 0)))
        line += " | + dense + window pass of %d: %.1f" % (nb, t_w)
    print(line, flush=True)
    del a
    torch.cuda.empty_cache()
