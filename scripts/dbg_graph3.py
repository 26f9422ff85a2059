import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if sys.argv[1] == "nocudnn":
    torch.backends.cudnn.enabled = False
from tests.parity_util import deepfm_parity_run
if sys.argv[1] == "criteo":
    err, losses, perr = deepfm_parity_run(B=256, steps=3, seed=5, return_all=True)
    print(err, losses); print({k: v for k, v in perr.items()})
else:
    print(deepfm_parity_run(B=128, steps=6, seed=7, rows=(3, 7, 40, 11, 600), layers=(32, 16), use_graph=True))
