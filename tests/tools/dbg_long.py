"""Debug aid: the long-horizon run of tests/longrun.py, per-step / per-variable differences against the committed fixture."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import longrun
name = sys.argv[1] if len(sys.argv) > 1 else "deepfm_bs256"
extra = {"cin_split": int(sys.argv[2])} if len(sys.argv) > 2 else None
g = np.load(os.path.join(ROOT, "tests", "golden", "long_%s.npz" % name))
P, train, ev, digest = longrun.make_inputs(name)
got = longrun.hip_run(name, P, train, ev, extra)
d = np.abs(got["train_losses"] - g["train_losses"])
print("train loss diffs every 10:", ["%.2e" % x for x in d[::10]])
print("first 12 hip", got["train_losses"][:12], "\nfirst 12 ora", g["train_losses"][:12])
for k in got["final_dense"]:
    if "final." + k in g.files:
        a, b = got["final_dense"][k].reshape(-1), g["final." + k].reshape(-1)
        i = int(np.abs(a - b).argmax())
        print("%-12s max|d| %.3e at %d hip %.6f ora %.6f  init %.6f" % (k, np.abs(a - b).max(), i, a[i], b[i], P[k].reshape(-1)[i]))
print("eval", got["eval_loss"], float(g["eval_loss"]), got["auc"], float(g["auc"]))
