#!/usr/bin/env python
"""Phase stamps (100 MHz wall clock) of the several-workgroups-per-field sort (sort_device.h field_sort_split_block), -DRSX_STAMPS
build: slot 0 = workgroup 0's entry, slots 1.. = the LATEST time any workgroup of the launch passed each phase boundary."""
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
lib = C.CDLL(os.environ["RSX_LIB_PATH"])
fn, fz = lib.rsx_dbg_stamps_embedding, lib.rsx_dbg_stamps_embedding_zero
row_off = criteo_row_off()
rng = np.random.default_rng(0)
a = EmbeddingArena(row_off, 16, max(B, 1025), "cuda", with_w1=True)
ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
names = ["entry (wg 0)", "bitmap zeroed, ids + slot sweep done", "ids marked", "keys compacted, prefix popcounts",
         "radix passes, segment starts, lists classified", "global stores issued", "reservations read", "lists written, ticket"]
acc, reps = np.zeros(8), 0
for s in range(30):
    fz()
    a.field_sort(ids)
    torch.cuda.synchronize()
    if s >= 10:
        buf = (C.c_ulonglong * 64)()
        assert fn(buf) == 0
        t = np.array(list(buf)[:8], np.float64)
        acc += np.where(t > 0, t - t[0], 0)
        reps += 1
t = acc / reps * 0.01
print("field_sort_split_k, B = %d x 39 fields: latest arrival of any workgroup at each phase boundary (us after workgroup 0's entry)" % B)
prev = 0.0
for n, v in zip(names, t):
    print("  %-42s %7.2f  (+%.2f)" % (n, v, v - prev))
    prev = v
