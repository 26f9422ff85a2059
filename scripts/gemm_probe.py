import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
M = 102400
for lib in ("default", "hipblaslt", "cublas"):
    if lib != "default":
        torch.backends.cuda.preferred_blas_library(lib)
    for (K, N) in ((128, 80), (80, 40), (40, 1)):
        A = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda"); dY = torch.randn(M, N, device="cuda")
        Wt = W.t().contiguous(); b = torch.randn(N, device="cuda")
        print(lib, "K=%d N=%d" % (K, N),
              "fwd mm %.1f" % t(lambda: torch.mm(A, W)),
              "fwd linear(W^T stored) %.1f" % t(lambda: torch.nn.functional.linear(A, Wt, b)),
              "dX %.1f" % t(lambda: torch.mm(dY, W.t())),
              "dW %.1f" % t(lambda: torch.mm(A.t(), dY)), flush=True)
