import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tests.parity_util import deepfm_parity_run
for od in ("f32", "f64"):
    os.environ["RSX_TEST_ORACLE_DTYPE"] = od
    for B in (512, 4096):
        err, losses, perr = deepfm_parity_run(B=B, steps=2, seed=41, kind="dcn", dropout=0.5, return_all=True)
        print(od, B, "err", err, "losses", losses, "perr", {k: round(v, 8) for k, v in perr.items()}, flush=True)
