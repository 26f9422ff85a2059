#!/usr/bin/env python
"""Per-kernel roofline table: each hand-written kernel timed stand-alone (HIP events over graph-captured repeats) at the
BASELINE sizes and at larger batches, against its bounding roofline (HBM 8 TB/s spec / fp32 MFMA 157.3 TF spec).
Algorithmic bytes / flops follow SURVEY.md section 8d.  Output: a text table (commit under profiles/)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from recsys_amd import _lib
from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
from recsys_amd.ops import AdamTF1, CinLayerFn, CrossLayers, DinPoolFn, EmbeddingArena, _ptr, _stream, check, lib
from kernel_roofline_util import synth_ids

HBM, MFMA32 = 8000.0, 157.3
dev = "cuda"


def timeit(fn, reps=20, inner=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (reps * inner) * 1e3      # us


rows = []


def rec(name, size, us, nbytes=None, flops=None):
    if nbytes is not None:
        ach = nbytes / us / 1e3
        rows.append("%-34s %-26s %9.2f us  %9.1f GB/s   %5.1f %% of HBM 8 TB/s   (%.2f MB alg.)" % (name, size, us, ach, 100 * ach / HBM, nbytes / 1e6))
    else:
        ach = flops / us / 1e6
        rows.append("%-34s %-26s %9.2f us  %9.2f TFLOP/s %5.1f %% of fp32 MFMA 157 TF (%.2f GF alg.)" % (name, size, us, ach, 100 * ach / MFMA32, flops / 1e9))
    print(rows[-1], flush=True)


lay = CriteoLayout.from_columns(build_feature_columns(16)[1])
rng = np.random.default_rng(0)
F, D = 39, 16
for B in (256, 4096, 16384, 65536):
    arena = EmbeddingArena(lay.row_off, D, B, dev, with_w1=True, tables=np.zeros((int(lay.row_off[-1]), D), np.float32) + 0.01,
                           w1=np.zeros(int(lay.row_off[-1]), np.float32))
    ids = torch.from_numpy(synth_ids(rng, B, lay.row_off)).to(dev)
    us = timeit(lambda: arena.gather(ids, fm=True, first_order=True))
    rec("gather_fm_fwd_k (+FM +1st order)", "B=%d F=39 D=16" % B, us, nbytes=B * F * (4 + 64 + 64))
    us = timeit(lambda: arena.field_sort(ids))
    rec("field_sort_k" if B <= 16384 else "field_sort_large (10 launches)", "B=%d" % B, us, nbytes=B * F * (4 + 16))
    E, S, y1, y2 = arena.gather(ids, fm=True, first_order=True)
    dX = torch.randn(B, F * D, device=dev); g1 = torch.randn(B, device=dev); g2 = torch.randn(B, device=dev)
    us = timeit(lambda: arena.segsum(B, S, dX, g1, g2))
    U = int(arena.nuniq.sum().item())
    rec("segsum (scatter%s)" % (", 2 launches" if B > 512 else ""), "B=%d U=%d" % (B, U), us, nbytes=B * F * (4 + 64) + U * 64)
    del arena
# optimizer sweep
arena = EmbeddingArena(lay.row_off, D, 256, dev, with_w1=True, tables=np.zeros((int(lay.row_off[-1]), D), np.float32), w1=np.zeros(int(lay.row_off[-1]), np.float32))
opt = AdamTF1(device=dev)
us = timeit(lambda: opt.step(arena.adam_segments()))
rec("adam_multi_k (TF-1 dense sweep)", "R=840646 D=16 + w1", us, nbytes=24 * (arena.R * D + arena.R))
# cross
for B in (256, 4096):
    op = CrossLayers(624, 3, B, dev)
    x0 = torch.randn(B, 624, device=dev); W = torch.randn(3, 624, device=dev) * 0.05; Bc = torch.randn(3, 624, device=dev) * 0.05
    us = timeit(lambda: op.forward(x0, W, Bc, None, want_xL=True))
    rec("cross_fwd_k (3 layers fused)", "B=%d dim=624" % B, us, nbytes=B * 2 * 2496 + 2 * 3 * 624 * 4)
    dW, dB, dX, g = torch.empty_like(W), torch.empty_like(Bc), torch.empty_like(x0), torch.randn(B, 624, device=dev)
    op.forward(x0, W, Bc, None, want_xL=True)
    us = timeit(lambda: op.backward(x0, W, Bc, dW, dB, dX, False, dxL=g))
    rec("cross_bwd4_k + cross_reduce_k", "B=%d dim=624" % B, us, nbytes=B * 3 * 2496 + 4 * 3 * 624 * 4)
# CIN
for (B, H, N) in ((256, 39, 128), (256, 128, 128)):
    X0 = torch.randn(B, 39, 16, device=dev) * 0.3; Xk = torch.randn(B, H, 16, device=dev) * 0.3
    W = torch.randn(39 * H, N, device=dev) * 0.05; c = torch.zeros(N, device=dev); out = torch.empty(B, N, 16, device=dev)
    fl = 2.0 * B * 16 * 39 * H * N
    us = timeit(lambda: check(lib().rsx_cin_layer_fwd(_ptr(X0), _ptr(Xk), _ptr(W), _ptr(c), _ptr(out), B, 39, H, N, 16, None, _stream())))
    rec("cin_fwd_k", "B=%d H=%d N=%d" % (B, H, N), us, flops=fl)
    g = torch.randn(B, N, 16, device=dev); dXk = torch.empty_like(Xk); dX0 = torch.empty_like(X0); dW = torch.empty_like(W); dc = torch.empty(N, device=dev); ws = torch.empty(int(lib().rsx_cin_bwd_workspace_floats(B, 39, H, N)), device=dev)
    us = timeit(lambda: check(lib().rsx_cin_layer_bwd(_ptr(X0), _ptr(Xk), _ptr(W), _ptr(out), _ptr(g), None, None, _ptr(dXk), 0, _ptr(dX0), 0, _ptr(dW), _ptr(dc), _ptr(ws), B, 39, H, N, 16, None, _stream())))
    rec("cin_bwd_dx_k + cin_bwd_dw_k", "B=%d H=%d N=%d" % (B, H, N), us, flops=2 * fl)
# DIN pooling
for B in (1024,):
    P, K = 100, 32
    Hh = torch.randn(B, P, K, device=dev); w = torch.randn(B, P, device=dev); idh = torch.randint(0, 3, (B, P), device=dev, dtype=torch.int32)
    out = torch.empty(B, K, device=dev)
    us = timeit(lambda: check(lib().rsx_din_pool_fwd(_ptr(Hh), _ptr(w), _ptr(idh), _ptr(out), B, P, K, _stream())))
    rec("din_pool_fwd_k", "B=%d P=100 K=32" % B, us, nbytes=B * (P * (4 + 4 * K + 4) + 4 * K))
open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kernel_roofline.txt"), "w").write(
    "# scripts/kernel_roofline.py: stand-alone kernel timings (HIP events over graph-captured repeats), 1x MI355X\n" + "\n".join(rows) + "\n")
