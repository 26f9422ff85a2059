"""GPU box: what the scatter launch of a DeepFM bs-256 step is made of (HIP events over graph replays): segment-sum only,
+ touched-row Adam, + the dense-variable Adam segment, + the window pass over 3 / 7 other lists."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from recsys_amd import _lib
from recsys_amd.ops import AdamTF1, DenseArena, EmbeddingArena
from scripts.kernel_roofline_util import timeit
from kernel_roofline_util import criteo_row_off, synth_ids

row_off = criteo_row_off()
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
a = EmbeddingArena(row_off, 16, B, "cuda", with_w1=True, w1_field_mask=(1 << 39) - 1)
with torch.no_grad():
    a.tables.normal_(); a.w1.normal_()
ids = [torch.from_numpy(synth_ids(rng, B, row_off)).cuda() for _ in range(8)]
dX = torch.randn(B, 39 * 16, device="cuda"); g1 = torch.randn(B, device="cuda"); g2 = torch.randn(B, device="cuda")
E, S, _, _ = a.gather(ids[0], fm=True, first_order=True)
opt = AdamTF1(device="cuda")
dense = DenseArena({"w": (73100,)}, "cuda")
a.sort_window(ids[:min(8, len(a.sortbufs))])
a.select(0)
print("B=%d" % B)
print("segsum only                          %.1f us" % timeit(lambda: a.segsum(B, S, dX, g1, g2)))
print("segsum + touched-row Adam            %.1f us" % timeit(lambda: a.segsum_adam(B, S, dX, g1, g2, opt, [], None)))
print("  + dense Adam (73 k floats)         %.1f us" % timeit(lambda: a.segsum_adam(B, S, dX, g1, g2, opt, dense.adam_segments(), None)))
for k in (2, 4, 8):
    if k <= len(a.sortbufs):
        print("  + window pass, window of %d steps   %.1f us" % (k, timeit(lambda: a.segsum_adam(B, S, dX, g1, g2, opt, dense.adam_segments(), None, window=(k, 0)))))
