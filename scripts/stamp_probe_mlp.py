#!/usr/bin/env python
"""Phase stamps (100 MHz wall clock) of workgroup 0 of the one-launch batch-norm-free tower (csrc/mlp_fused.hip), -DRSX_STAMPS build,
din.py's 'mlp_layer' shape: batch 1 024, 96 -> 100 -> 52 -> 20 -> 1."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSX_LIB_PATH"] = os.path.join(ROOT, "scripts", "_build", os.environ.get("RSX_STAMP_LIB", "librsx_stamps.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from recsys_amd.ops import DenseArena, FusedTower  # noqa: E402

B, k0, widths = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 96, [100, 52, 20]
fn = C.CDLL(os.environ["RSX_LIB_PATH"]).rsx_dbg_stamps_mlp
shapes, d = {}, k0
for i, n in enumerate(widths):
    shapes[f"mlp.W{i}"], shapes[f"mlp.b{i}"] = (d, n), (n,)
    d = n
shapes["mlp.Wout"], shapes["mlp.bout"] = (d, 1), (1,)
P = DenseArena(shapes, "cuda")
with torch.no_grad():
    P.flat.normal_(0, 0.1)
tw = FusedTower(P, "mlp", k0, widths, B, "cuda", batch_norm=False)
X, s0, y = torch.randn(B, k0, device="cuda"), torch.randn(B, device="cuda"), (torch.rand(B, device="cuda") < 0.4).float()
step = torch.tensor([3], dtype=torch.int32, device="cuda")
names = ["entry", "LDS zeroed (input rows requested)", "weights in LDS", "forward layer 0", "forward layer 1", "forward layer 2",
         "logit, loss, output-layer gradients, da of the last layer", "backward layer 2: tiles issued", "  .. stores drained (barrier)",
         "backward layer 1: tiles issued", "  .. stores drained (barrier)", "backward layer 0: tiles issued", "  .. stores drained (barrier)"]
acc, reps = np.zeros(13), 0
for s in range(30):
    tw.train_step(X, y, 0.5, step, s0=s0, head=("mlp.Wout", "mlp.bout", None, None), relu0=False, relu2=False)
    torch.cuda.synchronize()
    if s >= 10:
        buf = (C.c_ulonglong * 64)()
        assert fn(buf) == 0
        t = np.array(list(buf)[:13], np.float64)
        acc += t - t[0]
        reps += 1
t = acc / reps * 0.01
print("mlp_nobn_step_k, batch %d, %d -> %s -> 1: workgroup 0, us after its entry" % (B, k0, " -> ".join(map(str, widths))))
prev = 0.0
for n, v in zip(names, t):
    print("  %-62s %7.2f  (+%.2f)" % (n, v, v - prev))
    prev = v
