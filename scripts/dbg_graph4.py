import sys, os, faulthandler
faulthandler.enable()
import torch, torch.nn.functional as F
which = sys.argv[1]
B = 128
x = torch.randn(B, 80, device="cuda")
W = torch.randn(80, 32, device="cuda", requires_grad=True); b = torch.zeros(32, device="cuda", requires_grad=True)
W2 = torch.randn(32, 1, device="cuda", requires_grad=True); b2 = torch.zeros(1, device="cuda", requires_grad=True)
y = (torch.rand(B, device="cuda") > 0.5).float()
def body():
    if which == "addmm":
        h = torch.relu(torch.addmm(b, x, W))
        z = torch.addmm(b2, h, W2).reshape(-1)
    elif which == "matmul":
        h = torch.relu(x @ W + b)
        z = (h @ W2 + b2).reshape(-1)
    elif which == "cat":
        h = torch.relu(x @ W + b)
        z = torch.cat([h[:, :1], h[:, 1:2], h @ W2], -1).sum(1)
    loss = F.binary_cross_entropy_with_logits(z, y)
    loss.backward()
    return loss
for _ in range(2): body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
print(which, "captured", flush=True)
g.replay(); torch.cuda.synchronize()
print(which, "ok", float(out.detach()))
