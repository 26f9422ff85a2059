"""Probe: inside ONE HIP graph, does a long HBM sweep on a side stream overlap with a chain of short kernels on the
main stream?  (Decides whether the TF-1 dense Adam sweep of step t can hide behind step t+1's tower kernels.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recsys_amd import _lib
from recsys_amd.ops import AdamTF1
dev = "cuda"
rows, D = 840646, 16
var = torch.randn(rows, D, device=dev); m = torch.zeros_like(var); v = torch.zeros_like(var)
slot = torch.full((rows + 4,), -1, dtype=torch.int32, device=dev)
G = torch.zeros(1024, D, device=dev)
opt = AdamTF1(device=dev)
seg = [dict(kind=_lib.RSX_ADAM_TABLE_TF1, d=D, n=rows, var=var, m=m, v=v, g=G, slot=slot)]
x = torch.randn(256, 624, device=dev); W = torch.randn(624, 624, device=dev)
def chain(n=12):
    y = x
    for _ in range(n):
        y = torch.relu(y @ W) * 0.01
    return y
side = torch.cuda.Stream()
def body(mode):
    if mode == "serial":
        opt.step(seg); return chain()
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        opt.step(seg)
    y = chain()
    cur.wait_stream(side)
    return y
for mode in ("serial", "overlap"):
    for _ in range(3): body(mode)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8): out = body(mode)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print(mode, "us per (sweep + 12-kernel chain):", (time.perf_counter() - t0) / 50 / 8 * 1e6, flush=True)
# reference: each alone
for name, fn in (("sweep only", lambda: opt.step(seg)), ("chain only", chain)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(8): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print(name, "us:", (time.perf_counter() - t0) / 50 / 8 * 1e6, flush=True)
