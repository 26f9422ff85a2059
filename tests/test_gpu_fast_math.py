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
