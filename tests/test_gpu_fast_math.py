"""csrc/adam_fast.h: the window sweep's packed square root and division must return the bits of sqrtf() and '/' on their
domains -- the sweep's claim of being the same arithmetic as k single TF-1 steps (tests/test_gpu_adam_window.py) rests on it.
The square root is checked on EVERY float of its domain, the division on all 2^46 pairs of mantissas (~30 s) and on 8e8
structured pseudo-random pairs over its exponent range; the guard
that keeps operands inside the domains is exercised with adversarial optimizer state in test_gpu_adam_window.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_packed_sqrt_and_div_are_correctly_rounded_on_their_domains():
    from recsys_amd import _lib
    lib = _lib.lib()
    counts = torch.zeros(4, dtype=torch.int64, device="cuda")
    rc = lib.rsx_adam_fast_math_selftest(counts.data_ptr(), 20260928, 400, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    sqrt_bad, div_bad, pairs, div_all_bad = (int(x) for x in counts.cpu())
    assert pairs == 2 * 4096 * 256 * 400
    assert (sqrt_bad, div_bad, div_all_bad) == (0, 0, 0)


def test_fused_touched_row_update_with_extreme_operands_is_the_bits_of_the_full_sweep():
    """segsum_adam_k (scatter + touched-row TF-1 update + first-order vector, the cold slice for the untouched rows) against
    segment-sum + ONE full sweep (adam_multi_k) on state and gradients drawn from extreme magnitudes -- zeros of both signs,
    denormals, 2^-100 .. 2^45, 1e38 -- in all combinations of (var, m, v, g): same bits over several steps wherever the plain
    path is finite, NaN / inf in the same places elsewhere.  (Round 3 tried adam_fast.h's packed sqrt / division behind an
    operand guard in this launch: no faster -- the update is not on its critical path -- and not kept; this test is what
    such a change has to pass.)"""
    import numpy as np
    import torch
    from recsys_amd.ops import AdamTF1, EmbeddingArena
    rows = (40, 1, 300, 7, 64)
    row_off = np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)
    R, F, D, B = int(row_off[-1]), len(rows), 16, 128
    edge = [0.0, -0.0, 1e-45, -1e-45, 1e-39, 2.0 ** -126, 2.0 ** -100, 2.0 ** -97, 2.0 ** -96, 2.0 ** -95, 2.0 ** -94, 2.0 ** -86,
            2.0 ** -60, 2.0 ** -31, 2.0 ** -30, 2.0 ** -29, 1e-8, 1e-4, 0.5, 1.0, 3.0, 2.0 ** 20, 2.0 ** 21, 2.0 ** 22, 2.0 ** 33,
            2.0 ** 34, 2.0 ** 35, 2.0 ** 40, 2.0 ** 41, 2.0 ** 42, 2.0 ** 45, 1e38]
    rng = np.random.default_rng(5)

    def pick(shape, signed):
        x = rng.choice(np.array(edge, np.float32), size=shape)
        x = x * rng.choice(np.array([1.0, 1.5, 0.75], np.float32), size=shape)       # mantissas off the powers of two
        if signed:
            x = x * rng.choice(np.array([1.0, -1.0], np.float32), size=shape)
        return x.astype(np.float32)

    for hyper in ((1e-3, 0.9, 0.999, 1e-8), (0.05, 0.85, 0.5, 1e-3)):
        arenas, opts = [], []
        st = dict(tables=pick((R, D), True), m=pick((R, D), True), v=np.abs(pick((R, D), False)),
                  w1=pick((R,), True), mw=pick((R,), True), vw=np.abs(pick((R,), False)))
        for _ in range(2):
            a = EmbeddingArena(row_off, D, B, "cuda", with_w1=True, w1_field_mask=(1 << F) - 1)
            with torch.no_grad():
                for t, k in ((a.tables, "tables"), (a.m_t, "m"), (a.v_t, "v"), (a.w1, "w1"), (a.m_w, "mw"), (a.v_w, "vw")):
                    t.copy_(torch.from_numpy(st[k]))
            arenas.append(a)
            opts.append(AdamTF1(*hyper, device="cuda"))
        for step in range(3):
            ids = np.stack([rng.integers(0, r, B) for r in rows], 1).astype(np.int32)
            ids_t = torch.from_numpy(ids).cuda()
            dX = torch.from_numpy(pick((B, F * D), True)).cuda()
            S = torch.from_numpy(pick((B, D), True)).cuda()
            g1 = torch.from_numpy(pick((B,), True)).cuda()
            g2 = torch.from_numpy(pick((B,), True)).cuda()
            fused, plain = arenas
            fused.field_sort(ids_t)
            cold, _ = fused.adam_split_segments()
            opts[0].run_slice(opts[0].cold_slices(cold, [1.0])[0])
            fused.segsum_adam(B, S, dX, g1, g2, opts[0], [], None)
            plain.field_sort(ids_t)
            plain.segsum(B, S, dX, g1, g2)
            opts[1].step(plain.adam_segments())
            torch.cuda.synchronize()
            for name in ("tables", "m_t", "v_t", "w1", "m_w", "v_w"):
                x, y = getattr(fused, name), getattr(plain, name)
                fin = torch.isfinite(y)
                assert torch.equal(torch.isnan(x), torch.isnan(y)) and torch.equal(torch.isinf(x), torch.isinf(y)), (hyper, step, name)
                bad = (x.view(torch.int32) != y.view(torch.int32)) & fin
                assert int(bad.sum()) == 0, (hyper, step, name, int(bad.sum()), int(fin.sum()))
