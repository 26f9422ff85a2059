"""The one-launch tower without batch-norm (csrc/mlp_fused.hip, rsx_mlp_nobn_train_step; din/din.py:130-147 'mlp_layer')
against (a) the launch-per-layer kernels of the same FusedTower and (b) a plain torch fp64 autograd restatement with the
same keep masks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _towers(k0, widths, B, seed):
    from recsys_amd.ops import DenseArena, FusedTower
    g = torch.Generator().manual_seed(seed)
    out = []
    for fuse in (True, False):
        shapes, d = {}, k0
        for i, n in enumerate(widths):
            shapes[f"mlp.W{i}"], shapes[f"mlp.b{i}"] = (d, n), (n,)
            d = n
        shapes["mlp.Wout"], shapes["mlp.bout"] = (d, 1), (1,)
        P = DenseArena(shapes, "cuda")
        out.append((P, FusedTower(P, "mlp", k0, widths, B, "cuda", batch_norm=False)))
        out[-1][1]._mlp_ok = None if fuse else False
    vals = {k: (torch.randn(v.shape, generator=g) * (0.3 if k.endswith(("bout",)) or ".b" in k else 1.0 / np.sqrt(v.shape[0])))
            for k, v in out[0][0].params.items()}
    for P, _ in out:
        P.load({k: v.numpy() for k, v in vals.items()})
    assert out[0][1]._mlp_fused_ok() and not out[1][1]._mlp_fused_ok()
    return out, vals, g


@pytest.mark.parametrize("B,k0,widths,rate,explicit", [
    (1024, 96, (100, 52, 20), 0.5, False),      # din.py's mlp_layer at the BASELINE batch (hash dropout)
    (1000, 96, (100, 52, 20), 0.5, True),       # a ragged last tile, injected masks
    (37, 96, (100, 52, 20), 0.0, False),
    (256, 32, (64,), 0.3, True),
    (130, 112, (112, 4), 0.5, False),
    (16, 4, (4, 4, 4), 0.0, False),
])
def test_fused_mlp_equals_layerwise_and_fp64(B, k0, widths, rate, explicit):
    (pair, vals, g) = _towers(k0, list(widths), B, 11 * B + k0)
    X = torch.randn(B, k0, generator=g).cuda()
    s0 = (torch.randn(B, generator=g) * 0.2).cuda()
    y = (torch.rand(B, generator=g) < 0.4).float().cuda()
    step = torch.tensor([7], dtype=torch.int32, device="cuda")
    masks = None
    if explicit:
        masks = [(torch.rand(B, n, generator=g) >= rate).float().cuda() for n in widths]
    res = []
    for P, tw in pair:
        P.grad.zero_()
        loss, prob, dX, gs0, _ = tw.train_step(X, y, rate, step, s0=s0, head=("mlp.Wout", "mlp.bout", None, None), relu0=False,
                                               relu2=False, replicas=2, masks=masks, seed=0xD1AD)
        torch.cuda.synchronize()
        res.append(dict(loss=loss.clone(), prob=prob.clone(), dX=dX.clone(), gs0=gs0.clone(), grad=P.grad.clone()))
    a, b = res
    for k in ("loss", "prob", "dX", "gs0", "grad"):
        np.testing.assert_allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), rtol=2e-5, atol=2e-7, err_msg=k)
    if not explicit and rate > 0.0:
        return                       # (the hash masks are not visible from here: the layer-wise kernels are the reference)
    # fp64 autograd restatement (din/din.py:130-147), gradients scaled by 1 / (B * replicas) like the kernels'
    P = pair[0][0]
    W = {k: vals[k].double().cuda().requires_grad_() for k in vals}
    Xd = X.double().requires_grad_()
    s0d = s0.double().requires_grad_()
    h = Xd
    for i, n in enumerate(widths):
        h = torch.relu(h @ W[f"mlp.W{i}"] + W[f"mlp.b{i}"])
        if rate > 0.0:
            h = h * masks[i].double() / (1.0 - rate)
    z = (h @ W["mlp.Wout"]).reshape(-1) + W["mlp.bout"] + s0d
    ce = torch.clamp(z, min=0) - z * y.double() + torch.log1p(torch.exp(-z.abs()))
    (ce.sum() / (B * 2)).backward()
    np.testing.assert_allclose(a["loss"].cpu().numpy(), [float(ce.mean().detach())], rtol=1e-5)
    np.testing.assert_allclose(a["prob"].cpu().numpy(), torch.sigmoid(z).detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a["dX"].cpu().numpy(), Xd.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(a["gs0"].cpu().numpy(), s0d.grad.cpu().numpy(), rtol=1e-4, atol=1e-8)
    for k in vals:
        np.testing.assert_allclose(P[k].grad.cpu().numpy(), W[k].grad.cpu().numpy().reshape(P[k].shape), rtol=1e-4, atol=2e-7,
                                   err_msg=k)
