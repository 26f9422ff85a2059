import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import criteo
from recsys_amd.ops import EmbeddingArena
from tests.parity_util import synth_ids
row_off = criteo.row_offsets()
rng = np.random.default_rng(0)
for B in (256, 512, 1024, 2048, 4096, 8192, 16384):
    a = EmbeddingArena(row_off, 16, B, "cuda", with_w1=True)
    ids = torch.from_numpy(synth_ids(rng, B, row_off)).cuda()
    for _ in range(3): a.field_sort(ids)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): a.field_sort(ids)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("field_sort B=%6d  %.2f us" % (B, e0.elapsed_time(e1) * 1000 / 20))
