"""Debug aid: do cin_split 3 and 4 really run different arithmetic in the long-horizon harness?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import longrun
longrun.TRAIN_STEPS = 24
name = "xdeepfm_bs256_cin128"
P, train, ev, _ = longrun.make_inputs(name)
train = train[:24]
res = {}
for sp in (3, 4, 0):
    got = longrun.hip_run(name, P, train, ev[:2], {"cin_split": sp})
    res[sp] = got["train_losses"]
    print(sp, got["train_losses"][:4], got["eval_loss"])
print("3 vs 4 max |d|", np.abs(res[3] - res[4]).max(), " 3 vs 0", np.abs(res[3] - res[0]).max())
