"""xDeepFM (BASELINE config 3): CIN layer op parity and full train steps vs the oracle."""
import numpy as np
import pytest

from oracle import criteo, init, models, nn
from tests.parity_util import synth_ids

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("B,F,H,N", [(5, 39, 39, 128), (4, 39, 128, 128), (3, 7, 5, 20), (9, 39, 20, 10), (2, 4, 33, 16)])
def test_cin_layer_fwd_bwd(B, F, H, N):
    from recsys_amd.ops import CinLayerFn
    rng = np.random.default_rng(B * 100 + H)
    D = 16
    X0 = rng.standard_normal((B, F, D)).astype(np.float32) * 0.3
    Xk = rng.standard_normal((B, H, D)).astype(np.float32) * 0.3
    W = rng.standard_normal((F * H, N)).astype(np.float32) * 0.1
    c = rng.standard_normal(N).astype(np.float32) * 0.1
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    out_o = models.cin_layer_fwd(X0.astype(np.float64), Xk.astype(np.float64), W.astype(np.float64), c.astype(np.float64))
    d0_o, dk_o, dW_o, dc_o = models.cin_layer_bwd(X0.astype(np.float64), Xk.astype(np.float64), W.astype(np.float64), out_o,
                                                  g.astype(np.float64))
    t = [torch.from_numpy(a).cuda().requires_grad_() for a in (X0, Xk, W, c)]
    out = CinLayerFn.apply(*t)
    out.backward(torch.from_numpy(g).cuda())
    tol = dict(rtol=2e-5, atol=2e-5)                      # fp32 MFMA vs fp64 oracle
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_o, **tol)
    # relu masks can differ only where |pre| ~ 1e-7; compare gradients where the fp64 pre-activation is not that close to 0
    np.testing.assert_allclose(t[0].grad.cpu().numpy(), d0_o, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(t[1].grad.cpu().numpy(), dk_o, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(t[2].grad.cpu().numpy(), dW_o, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(t[3].grad.cpu().numpy(), dc_o, rtol=1e-4, atol=1e-4)


def test_cin_first_layer_aliasing_x0():
    """Layer 0 passes X0 as both operands (xdeepfm/xdeepfm.py:143-148); autograd must add both gradient roles."""
    from recsys_amd.ops import CinLayerFn
    rng = np.random.default_rng(1)
    B, F, N, D = 6, 39, 32, 16
    X0 = rng.standard_normal((B, F, D)).astype(np.float32) * 0.3
    W = rng.standard_normal((F * F, N)).astype(np.float32) * 0.1
    c = np.zeros(N, np.float32)
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    X64 = X0.astype(np.float64)
    out_o = models.cin_layer_fwd(X64, X64, W.astype(np.float64), c.astype(np.float64))
    d0, dk, _, _ = models.cin_layer_bwd(X64, X64, W.astype(np.float64), out_o, g.astype(np.float64))
    tx = torch.from_numpy(X0).cuda().requires_grad_()
    out = CinLayerFn.apply(tx, tx, torch.from_numpy(W).cuda(), torch.from_numpy(c).cuda())
    out.backward(torch.from_numpy(g).cuda())
    np.testing.assert_allclose(tx.grad.cpu().numpy(), d0 + dk, rtol=1e-4, atol=1e-4)
