"""Micro-benchmark: adam_multi_k (TABLE_TF1) bandwidth vs footprint -- where do L2 (32 MB) / Infinity Cache (256 MB) help?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recsys_amd import _lib
from recsys_amd.ops import AdamTF1
dev = "cuda"
opt = AdamTF1(device=dev)
for rows in [20000, 80000, 160000, 320000, 840646, 1600000, 3200000, 6400000]:
    D = 16
    var = torch.randn(rows, D, device=dev); m = torch.zeros_like(var); v = torch.zeros_like(var)
    slot = torch.full((rows + 4,), -1, dtype=torch.int32, device=dev)
    G = torch.zeros(1024, D, device=dev)
    seg = [dict(kind=_lib.RSX_ADAM_TABLE_TF1, d=D, n=rows, var=var, m=m, v=v, g=G, slot=slot)]
    for _ in range(5): opt.step(seg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps): opt.step(seg)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    foot = rows * D * 4 * 3 / 1e6
    print("rows=%8d footprint=%7.1f MB  %.4f ms  %.0f GB/s (24 B/elem)" % (rows, foot, ms, rows * D * 24 / ms / 1e6), flush=True)
