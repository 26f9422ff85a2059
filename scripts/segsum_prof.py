"""GPU box: field_sort + two-stage segsum at one batch size, for rocprofv3 --kernel-trace --stats (per-kernel split)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from recsys_amd.ops import EmbeddingArena
from kernel_roofline_util import criteo_row_off, synth_ids

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
row_off = criteo_row_off()
rng = np.random.default_rng(0)
a = EmbeddingArena(row_off, 16, B, "cuda", with_w1=True)
a.tables.normal_(); a.w1.normal_()
ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
F = a.F
dX = torch.randn(B, F * 16, device="cuda"); g1 = torch.randn(B, device="cuda"); g2 = torch.randn(B, device="cuda")
E, S, _, _ = a.gather(ids, fm=True, first_order=True)
for _ in range(20):
    a.field_sort(ids)
    a.segsum(B, S, dX, g1, g2)
torch.cuda.synchronize()
nl = a.segid[F * a.stride:F * a.stride + 2 * F].cpu().numpy() if a.partials is not None else None
print("B", B, "nuniq", int(a.nuniq.sum()), "long/huge", None if nl is None else (int(nl[:F].sum()), int(nl[F:].sum())))
