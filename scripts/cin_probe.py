#!/usr/bin/env python
"""Stand-alone timings of the CIN kernels (forward, backward) at the BASELINE config-3 shapes; used by cin_variants.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recsys_amd.ops import _ptr, _stream, check, lib
from scripts.kernel_roofline_util import timeit

dev = "cuda"
for (B, H, N) in ((256, 39, 128), (256, 128, 128)):
    X0 = torch.randn(B, 39, 16, device=dev) * 0.3; Xk = torch.randn(B, H, 16, device=dev) * 0.3
    W = torch.randn(39 * H, N, device=dev) * 0.05; c = torch.zeros(N, device=dev); out = torch.empty(B, N, 16, device=dev)
    fl = 2.0 * B * 16 * 39 * H * N
    us = timeit(lambda: check(lib().rsx_cin_layer_fwd(_ptr(X0), _ptr(Xk), _ptr(W), _ptr(c), _ptr(out), B, 39, H, N, 16, None, _stream())))
    print("fwd  H=%3d %8.2f us %6.1f TF" % (H, us, fl / us / 1e6))
    g = torch.randn(B, N, 16, device=dev); dXk = torch.empty_like(Xk); dX0 = torch.empty_like(X0); dW = torch.empty_like(W); dc = torch.empty(N, device=dev); ws = torch.empty(int(lib().rsx_cin_bwd_workspace_floats(B, 39, H, N)), device=dev)
    us = timeit(lambda: check(lib().rsx_cin_layer_bwd(_ptr(X0), _ptr(Xk), _ptr(W), _ptr(out), _ptr(g), None, None, _ptr(dXk), 0, _ptr(dX0), 0, _ptr(dW), _ptr(dc), _ptr(ws), B, 39, H, N, 16, None, _stream())))
    print("bwd  H=%3d %8.2f us %6.1f TF" % (H, us, 2 * fl / us / 1e6), flush=True)
