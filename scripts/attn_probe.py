"""GPU box: fused DIN attention MLP (rsx_din_attn_fwd / _bwd) against torch fp64 autograd, plus timings at B=1024, P=100."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from recsys_amd._lib import lib, check
from recsys_amd.ops import _ptr, _stream
torch.manual_seed(0)
def run(B, P, K, rate, mk, time_it=False):
    M = B * P
    dev = 'cuda'
    H = torch.randn(M, K, device=dev); q = torch.randn(B, K, device=dev)
    W0 = torch.randn(4 * K, 80, device=dev) * 0.1; b0 = torch.randn(80, device=dev) * 0.1
    W1 = torch.randn(80, 40, device=dev) * 0.1; b1 = torch.randn(40, device=dev) * 0.1
    W2 = torch.randn(40, device=dev) * 0.1; b2 = torch.randn(1, device=dev)
    m1 = (torch.rand(M, 80, device=dev) > 0.5).float(); m2 = (torch.rand(M, 40, device=dev) > 0.5).float()
    dw = torch.randn(M, device=dev)
    a1 = torch.empty(M, 80, device=dev); a2 = torch.empty(M, 40, device=dev); w = torch.empty(M, device=dev)
    M1, M2 = (_ptr(m1), _ptr(m2)) if mk else (None, None)
    fwd = lambda: check(lib().rsx_din_attn_fwd(_ptr(H), _ptr(q), _ptr(W0), _ptr(b0), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(a1), _ptr(a2), _ptr(w),
                                               M1, M2, None, 0, 0, rate, None, None, B, P, K, 80, 40, _stream()), "fwd")
    fwd()
    dH = torch.empty(M, K, device=dev); dq = torch.empty(B, K, device=dev)
    n = 4 * K * 80 + 80 + 80 * 40 + 40 + 40 + 1
    grads = torch.empty(n, device=dev)
    ws = torch.empty(int(lib().rsx_din_attn_bwd_workspace_floats(B, P, K, 80, 40)), device=dev)
    bwd = lambda: check(lib().rsx_din_attn_bwd(_ptr(H), _ptr(q), _ptr(W0), _ptr(W1), _ptr(W2), _ptr(a1), _ptr(a2), _ptr(dw), _ptr(dH), _ptr(dq), _ptr(grads),
                                               _ptr(ws), M1, M2, None, 0, 0, rate, 0, None, None, None, B, P, K, 80, 40, _stream()), "bwd")
    bwd()
    torch.cuda.synchronize()
    if time_it:
        for name, f in (("fwd", fwd), ("bwd", bwd)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            print("  %s %.1f us" % (name, e0.elapsed_time(e1) * 1000 / 20))
        return
    t = [x.double().requires_grad_() for x in (H, q, W0, b0, W1, b1, W2, b2)]
    Hd, qd, W0d, b0d, W1d, b1d, W2d, b2d = t
    qq = qd[:, None, :].expand(B, P, K).reshape(M, K)
    x = torch.cat([Hd, qq, Hd * qq, Hd - qq], 1)
    r1 = torch.relu(x @ W0d + b0d); e1_ = r1 * (m1.double() * 2 if mk else 1)
    r2 = torch.relu(e1_ @ W1d + b1d); e2_ = r2 * (m2.double() * 2 if mk else 1)
    wr = e2_ @ W2d + b2d
    (wr * dw.double()).sum().backward()
    ref = torch.cat([W0d.grad.reshape(-1), b0d.grad, W1d.grad.reshape(-1), b1d.grad, W2d.grad, b2d.grad])
    sc = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    print(B, P, K, rate, "w %.1e  dH %.1e  dq %.1e  grads %.1e (dW0 %.1e dW1 %.1e dW2 %.1e)" % (
        sc(w, wr.detach()), sc(dH, Hd.grad), sc(dq, qd.grad), sc(grads, ref),
        sc(grads[:4 * K * 80], ref[:4 * K * 80]), sc(grads[4 * K * 80 + 80:4 * K * 80 + 80 + 3200], ref[4 * K * 80 + 80:4 * K * 80 + 80 + 3200]),
        sc(grads[-41:-1], ref[-41:-1])))
for cfg in ((3, 5, 16), (7, 13, 32), (64, 100, 32), (300, 70, 32)):
    for rate, mk in ((0.0, False), (0.5, True)):
        run(*cfg, rate, mk)
print("timing B=1024 P=100 K=32:")
run(1024, 100, 32, 0.5, False, time_it=True)
