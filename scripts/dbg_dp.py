import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
from tests.parity_util import deepfm_parity_run
for dp, tower, steps in ((False, "hip", 5), (False, "torch", 5), (True, "hip", 5)):
    err, losses, perr = deepfm_parity_run(B=64, steps=steps, seed=31, rows=(3, 7, 40, 11, 600), layers=(32, 16), return_all=True,
                                          kind="deepfm", dropout=0.5, data_parallel=dp, tower=tower)
    print("dp", dp, tower, "err %.3e" % err, "losses", [(round(a, 6), round(b, 6)) for a, b in losses])
    print({k: "%.1e" % v for k, v in perr.items() if v > 1e-6})
