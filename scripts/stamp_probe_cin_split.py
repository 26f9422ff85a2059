#!/usr/bin/env python
"""Where does a stand-alone split-operand CIN launch (csrc/cin_split.hip, ns = 3, H = N = 128, batch 256) spend its microseconds?
Phase stamps (100 MHz wall clock) of workgroup (0, 0) on a -DRSX_STAMPS build of that one translation unit
(scripts/build_cin_split_variant.sh stamps), and first entry -> last exit over all workgroups."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RSX_LIB_PATH", os.path.join(ROOT, "scripts", "_build", "librsx_cs_stamps.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from recsys_amd.ops import _ptr, _stream, check, lib  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, H, N = 256, 128, 128
ns = int(os.environ.get("NS", "3"))
dev = "cuda"
X0 = torch.randn(B, 39, 16, device=dev) * 0.3
Xk = torch.randn(B, H, 16, device=dev) * 0.3
W = torch.randn(39 * H, N, device=dev) * 0.05
c = torch.zeros(N, device=dev)
out = torch.empty(B, N, 16, device=dev)
ws = torch.empty(int(lib().rsx_cin_split_weight_elems(39, H, N, ns)), dtype=torch.int16, device=dev)
Wh, wh = (C.c_void_p * 1)(W.data_ptr()), (C.c_void_p * 1)(ws.data_ptr())
Hh, Nh = (C.c_int32 * 1)(H), (C.c_int32 * 1)(N)
check(lib().rsx_cin_split_prep(Wh, wh, Hh, Nh, 1, 39, ns, _stream()))
dout = torch.randn(B, N, 16, device=dev)
dXk = torch.empty(B, H, 16, device=dev)
parts = torch.empty(int(lib().rsx_cin_bf16_dx0_parts_floats(B, 39, (H + 15) // 16 * 16)), device=dev)
wsb = torch.empty(int(lib().rsx_cin_split_bwd_workspace_bytes(B, N, ns)), dtype=torch.uint8, device=dev)


def run():
    if which == "fwd":
        check(lib().rsx_cin_split_fwd(_ptr(X0), _ptr(Xk), _ptr(ws), _ptr(c), _ptr(out), B, 39, H, N, 16, ns, _stream()))
    else:
        check(lib().rsx_cin_split_bwd_dx(_ptr(X0), _ptr(Xk), _ptr(ws), _ptr(out), _ptr(dout), None, None, _ptr(dXk), 0, _ptr(parts),
                                         _ptr(wsb), B, 39, H, N, 16, ns, _stream()))


L = C.CDLL(os.environ["RSX_LIB_PATH"])
acc, reps = np.zeros(64), 0
for s in range(40):
    L.rsx_dbg_stamps_cin_split_reset()
    run()
    torch.cuda.synchronize()
    if s >= 8:
        buf = (C.c_ulonglong * 64)()
        assert L.rsx_dbg_stamps_cin_split(buf) == 0
        t = np.array(list(buf), np.float64)
        z = t[0] if which == "fwd" else t[32]
        acc += np.where(t > 0, t - z, 0)
        reps += 1
acc /= reps * 100.0       # us
names = {32: "entry", 33: "prologue loads requested, operands split (+ dpre planes: tile 0 only)", 34: "prologue barrier",
         35: "field loop done", 36: "ring drained (barrier)", 37: "end", 48: "LAST workgroup's entry", 49: "LAST workgroup's exit"} if which != "fwd" else {0: "entry", 1: "prologue loads requested, operands split", 2: "prologue barrier (sX0 + first slots landed)", 3: "field loop done",
         4: "ring drained (barrier)", 5: "end", 16: "LAST workgroup's entry", 17: "LAST workgroup's exit"}
print("cin_split %s ns=%d H=N=128 B=256, workgroup (0,0) [dx: (1,0)], us from its entry:" % (which, ns))
for k in sorted(names):
    print("  %2d %-50s %8.2f" % (k, names[k], acc[k]))
