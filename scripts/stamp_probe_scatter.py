#!/usr/bin/env python
"""Phase stamps (100 MHz wall clock) inside the large-batch scatter's stage A (segsum_tiles_k), -DRSX_STAMPS build:
workgroup 0 (a 4-row bucketized field) and workgroup 24 (a 100 000-row hashed field at B = 4096), dX only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSX_LIB_PATH"] = os.path.join(ROOT, "scripts", "_build", os.environ.get("RSX_STAMP_LIB", "librsx_stamps.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from recsys_amd.ops import EmbeddingArena  # noqa: E402
from scripts.kernel_roofline_util import criteo_row_off, synth_ids  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fm = len(sys.argv) > 2 and sys.argv[2] == "fm"
fn = C.CDLL(os.environ["RSX_LIB_PATH"]).rsx_dbg_stamps_embedding
row_off = criteo_row_off()
rng = np.random.default_rng(0)
a = EmbeddingArena(row_off, 16, B, "cuda", with_w1=True, w1_field_mask=(1 << 39) - 1)
with torch.no_grad():
    a.tables.normal_(); a.w1.normal_()
ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
dX = torch.randn(B, 39 * 16, device="cuda"); g1 = torch.randn(B, device="cuda"); g2 = torch.randn(B, device="cuda")
E, S, _, _ = a.gather(ids, fm=True, first_order=True)
a.field_sort(ids)
from recsys_amd.ops import AdamTF1  # noqa: E402
opt = AdamTF1(device="cuda")
acc = np.zeros(64)
reps = 0
win = None
dense_segs = []
if os.environ.get("RSX_STAMP_WINDOW"):       # the DeepFM launch: window pass over 7 other lists + the dense arena
    from recsys_amd.ops import DenseArena  # noqa: E402
    nb = min(8 if B <= 1024 else 4, len(a.sortbufs))
    a.sort_window([ids] + [torch.from_numpy(synth_ids(rng, B, row_off)).cuda() for _ in range(nb - 1)])
    a.select(0)
    opt.window_sweep(a.adam_split_segments(window_k=nb)[0])     # (leaves the window's step sizes in the optimizer state)
    win = (nb, 0)
    dense_segs = DenseArena({"w": (73100,)}, "cuda").adam_segments()
fz = C.CDLL(os.environ["RSX_LIB_PATH"]).rsx_dbg_stamps_embedding_zero
last = np.zeros(4)
for s in range(30):
    fz()
    if fm:
        a.segsum_adam(B, S, dX, g1, g2, opt, dense_segs, None, window=win)
    else:
        a.segsum_adam(B, None, dX, None, None, opt, [], None)
    torch.cuda.synchronize()
    if s >= 10:
        buf = (C.c_ulonglong * 64)()
        assert fn(buf) == 0
        t = np.array(list(buf), np.float64)
        acc[:32] += np.where(t[:32] > 0, t[:32] - t[0], 0)
        acc[32:] += np.where(t[32:] > 0, t[32:] - t[32], 0)
        last += np.where(t[59:63] > 0, t[59:63] - t[32], 0) * 0.01
        reps += 1
t = acc / reps * 0.01
names = ["entry", "tile indices in LDS", "chunk classified (LDS reads, masks)", "first batch of rows arrived",
         "main loop done (sums stored)", "extension done"]
for g, what in ((0, "workgroup 0 (field 0, tile 0)"), (16, "workgroup 24")):
    print("---- B = %d, %s: us since workgroup 0's entry" % (B, what))
    for k, n in enumerate(names):
        print("%-40s %8.2f" % (n, t[g + k]))
names = ["entry", "segment sum of the wave ready (segsum_wave returns)", "Adam update stored", "workgroup barrier passed",
         "arrival counter done"]
for g, what in ((32, "workgroup 0 (helpers of field 0's huge segments)"), (48, "workgroup 432 (row owners, 100 000-row field)")):
    print("---- stage B + Adam (segsum_adam_k), B = %d, %s: us since workgroup 0's entry" % (B, what))
    for k, n in enumerate(names):
        print("%-55s %8.2f" % (n, t[g + k]))

print("---- LAST exit per role (us since workgroup 0's entry): row owners %.2f, window pass %.2f, dense / riders %.2f | latest "
      "ENTRY of a window-pass workgroup %.2f" % tuple(last / reps))
if os.environ.get("RSX_STAMP_WINDOW"):
    print("---- first window-pass workgroup: entry %.2f, slot maps read %.2f, table rows stored %.2f, first-order done %.2f, "
          "counter done %.2f | last workgroup of the grid: entry %.2f, work done %.2f, counter done %.2f" %
          (t[40], t[41], t[42], t[43], t[44], t[58], t[56], t[57]))
