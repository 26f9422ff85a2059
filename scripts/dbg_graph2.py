import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
which = sys.argv[1]
x = torch.randn(64, 32, device="cuda")
W = torch.randn(32, 16, device="cuda", requires_grad=True)
gam = torch.ones(16, device="cuda", requires_grad=True); bet = torch.zeros(16, device="cuda", requires_grad=True)
y = (torch.rand(64, device="cuda") > 0.5).float()
def body():
    h = torch.relu(x @ W)
    if which == "bn":
        h = F.batch_norm(h, None, None, gam, bet, True, 0.0, 1e-3)
    if which == "bnoff":
        with torch.backends.cudnn.flags(enabled=False):
            h = F.batch_norm(h, None, None, gam, bet, True, 0.0, 1e-3)
    if which == "drop":
        h = F.dropout(h, 0.5, True)
    z = h.sum(1)
    if which == "bce":
        loss = F.binary_cross_entropy_with_logits(z, y)
    else:
        loss = z.mean()
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
