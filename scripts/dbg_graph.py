import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recsys_amd.ops import EmbeddingArena, AdamTF1, DenseArena, gather_fm
which = sys.argv[1]
rng = np.random.default_rng(0)
row_off = np.array([0, 50, 60, 1000], np.int64)
B, D, F = 64, 16, 3
a = EmbeddingArena(row_off, D, B, "cuda", with_w1=True, tables=rng.standard_normal((1000, D)).astype(np.float32), w1=rng.standard_normal(1000).astype(np.float32))
ids = torch.from_numpy(np.stack([rng.integers(0, 50, B), rng.integers(0, 10, B), rng.integers(0, 940, B)], 1).astype(np.int32)).cuda()
dense = DenseArena({"w": (48, 4)}, "cuda")
opt = AdamTF1(device="cuda")
dX = torch.randn(B, F * D, device="cuda")
gy1 = torch.randn(B, device="cuda")

def body():
    if which == "gather":
        return a.gather(ids, True, True)[0]
    if which == "sort":
        a.field_sort(ids); return a.perm
    if which == "segsum":
        a.segsum(B, None, dX, gy1, None); return a.G
    if which == "adam":
        opt.step(a.adam_segments() + dense.adam_segments()); return a.tables
    if which == "autograd":
        a.field_sort(ids)
        E, y1, y2 = gather_fm(a, ids, True, True)
        loss = (E @ dense["w"]).sum() + y1.sum() + y2.sum()
        loss.backward()
        return loss
    if which == "all":
        a.field_sort(ids)
        E, y1, y2 = gather_fm(a, ids, True, True)
        loss = (E @ dense["w"]).sum() + y1.sum() + y2.sum()
        loss.backward()
        opt.step(a.adam_segments() + dense.adam_segments())
        return loss
a.field_sort(ids)
for _ in range(2):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
print(which, "captured", flush=True)
g.replay(); g.replay()
torch.cuda.synchronize()
print(which, "replayed ok", float(out.float().abs().sum()), flush=True)
