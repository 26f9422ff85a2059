#!/usr/bin/env python
"""Phase stamps (100 MHz wall clock) inside din_attn_bwd_k (-DRSX_STAMPS build, scripts/build_stamps.sh) at din.py's shapes:
B = 1024, P = 100, K = 32, MLP 80 / 40, about half of the history positions valid, dH accumulated; workgroup 0's SECOND
64-row block (steady state: weights staged, caches as warm as they get), wave 0.  Also the launch's average duration."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSX_LIB_PATH"] = os.path.join(ROOT, "scripts", "_build", os.environ.get("RSX_STAMP_LIB", "librsx_stamps.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from recsys_amd import _lib  # noqa: E402
from recsys_amd.ops import _ptr, _stream  # noqa: E402

B, P, K, N1, N2 = 1024, 100, 32, 80, 40
L = _lib.lib()
dbg = C.CDLL(os.environ["RSX_LIB_PATH"])
rng = np.random.default_rng(0)
lens = rng.integers(1, P + 1, B)
ids = np.zeros((B, P), np.int32)
for b in range(B):
    ids[b, :lens[b]] = rng.integers(1, 60000, lens[b])
ids_t = torch.from_numpy(ids).cuda()
M = B * P
f = lambda *s: torch.randn(*s, device="cuda")
H, q = f(M, K), f(B, K)
W0, W1, W2 = f(4 * K, N1) * 0.1, f(N1, N2) * 0.1, f(N2) * 0.1
a1, a2 = torch.relu(f(M, N1)), torch.relu(f(M, N2))
dw = f(M)
dH = torch.zeros(M, 2 * K, device="cuda")
rows = torch.zeros(M + 2 + (M + 1023) // 1024, dtype=torch.int32, device="cuda")
cnt = rows[M:]
_lib.check(L.rsx_din_valid_rows(_ptr(ids_t), B, P, _ptr(rows), _ptr(cnt), None, _stream()))
ws = torch.empty(int(L.rsx_din_attn_bwd_workspace_floats(B, P, K, N1, N2)), device="cuda")
step = torch.zeros(1, dtype=torch.int32, device="cuda")


def run():
    _lib.check(L.rsx_din_attn_bwd_nofinish(_ptr(H), _ptr(q), _ptr(W0), _ptr(W1), _ptr(W2), _ptr(a1), _ptr(a2), _ptr(dw), _ptr(dH),
                                           _ptr(ws), None, None, _ptr(step), 1, 0, 0.0, 1, _ptr(rows), _ptr(cnt), _ptr(ids_t),
                                           B, P, K, N1, N2, 2 * K, _stream()))


for _ in range(5):
    run()
torch.cuda.synchronize()
print("valid rows", int(cnt[0].item()), "of", M, "-> 64-row blocks", (int(cnt[0].item()) + 63) // 64)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
print("din_attn_bwd_k back to back: %.2f us per launch" % (e0.elapsed_time(e1) * 1000 / 50))
acc, reps = np.zeros(16), 0
buf = (C.c_ulonglong * 64)()
for s in range(20):
    dbg.rsx_dbg_stamps_attn_zero()
    run()
    torch.cuda.synchronize()
    assert dbg.rsx_dbg_stamps_attn(buf) == 0
    t = np.array(list(buf)[:16], np.float64)
    acc += (t - t[0]) * 0.01
    reps += 1
t = acc / reps
names = ["entry", "weights staged in LDS, first row index requested", "second block: start", "  g2 built (dw / a2 / a1 / h / q / dH arrived)",
         "  S3 dg1 = g2 . W1^T", "  S4 g1 tiles written", "  barrier", "  S5 dx = g1 . W0^T, dH / dq rows stored", "  S6 weight-gradient MFMAs",
         "  barrier (block done)", "all blocks done", "partials written (workgroup 0)", "last workgroup's exit"]
prev = 0.0
for k, n in enumerate(names):
    print("%-62s %7.2f us  (+%.2f)" % (n, t[k], t[k] - prev))
    prev = t[k]
