"""Data-parallel code path on ONE GPU (world_size 1 over RCCL): the all-gather / all-reduce plumbing, the global sort +
segment-sum on gathered blocks and HIP-graph capture of the collectives.  With one replica the numbers must equal
the single-process run, i.e. stay within the oracle tolerance.  (N > 1 is covered by tests/test_dist_gloo.py on CPU.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %r)
from tests.parity_util import deepfm_parity_run
# (injected dropout masks are per-step tensors, which a captured graph cannot follow: the graph case runs dropout 0)
for kind, graph, drop in (("deepfm", False, 0.5), ("deepfm", True, 0.0), ("dcn", False, 0.5), ("dcn", True, 0.0)):
    err, losses, perr = deepfm_parity_run(B=64, steps=5, seed=31, rows=(3, 7, 40, 11, 600), layers=(32, 16), return_all=True,
                                          kind=kind, dropout=drop, use_graph=graph, data_parallel=True)
    assert err < 1e-5, (kind, err)
    assert all(abs(a - b) < 1e-5 for a, b in losses), (kind, losses)
    assert max(perr.values()) < 5e-5, (kind, perr)
import torch.distributed as dist
dist.destroy_process_group()
print("DP_OK")
""" % ROOT


@pytest.mark.parametrize("capture_collectives", ["0", "1"])
def test_dp_world1_rccl_matches_oracle(capture_collectives):
    """'0': graph SEGMENTS with eager RCCL calls between them (default); '1': collectives captured into the graph."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               RSX_DP_CAPTURE=capture_collectives)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert "DP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
