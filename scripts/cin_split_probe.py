#!/usr/bin/env python
"""Stand-alone timings of the CIN forward at the BASELINE config-3 shapes: fp32 MFMA (cin.hip), bf16 (cin_bf16.hip, two and
eight examples per workgroup) and the split-operand kernels (cin_split.hip, ns = 1 / 2 / 3)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recsys_amd.ops import _ptr, _stream, check, lib  # noqa: E402
from scripts.kernel_roofline_util import timeit  # noqa: E402

dev = "cuda"
for (B, H, N) in ((256, 39, 128), (256, 128, 128)):
    X0 = torch.randn(B, 39, 16, device=dev) * 0.3
    Xk = torch.randn(B, H, 16, device=dev) * 0.3
    W = torch.randn(39 * H, N, device=dev) * 0.05
    c = torch.zeros(N, device=dev)
    out = torch.empty(B, N, 16, device=dev)
    fl = 2.0 * B * 16 * 39 * H * N
    us = timeit(lambda: check(lib().rsx_cin_layer_fwd(_ptr(X0), _ptr(Xk), _ptr(W), _ptr(c), _ptr(out), B, 39, H, N, 16, None, _stream())))
    print("fwd fp32 MFMA        H=%3d %8.2f us %7.1f TF" % (H, us, fl / us / 1e6))
    w16 = torch.empty(int(lib().rsx_cin_bf16_weight_elems(39, H, N)), dtype=torch.int16, device=dev)
    check(lib().rsx_cin_prep_bf16(_ptr(W), _ptr(w16), 39, H, N, _stream()))
    us = timeit(lambda: check(lib().rsx_cin_layer_fwd_bf16(_ptr(X0), _ptr(Xk), _ptr(w16), _ptr(c), _ptr(out), B, 39, H, N, 16, None, _stream())))
    print("fwd bf16 (cin_wide=%s) H=%3d %8.2f us %7.1f TF" % (__import__("recsys_amd._lib", fromlist=["form"]).form("cin_wide"), H, us, fl / us / 1e6))
    for ns in (1, 2, 3, 4):
        ws = torch.empty(int(lib().rsx_cin_split_weight_elems(39, H, N, ns)), dtype=torch.int16, device=dev)
        Wh, wh = (C.c_void_p * 1)(W.data_ptr()), (C.c_void_p * 1)(ws.data_ptr())
        Hh, Nh = (C.c_int32 * 1)(H), (C.c_int32 * 1)(N)
        usp = timeit(lambda: check(lib().rsx_cin_split_prep(Wh, wh, Hh, Nh, 1, 39, ns, _stream())))
        us = timeit(lambda: check(lib().rsx_cin_split_fwd(_ptr(X0), _ptr(Xk), _ptr(ws), _ptr(c), _ptr(out), B, 39, H, N, 16, ns, _stream())))
        print("fwd split ns=%d       H=%3d %8.2f us %7.1f TF (fp32-equivalent flops)   prep %6.2f us" % (ns, H, us, fl / us / 1e6, usp), flush=True)
