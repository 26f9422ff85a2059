"""Every RSX_* A/B knob of the training step flipped ONCE against the default schedule: a knob that only
changes how the same arithmetic is scheduled (which launch carries the sort, how long a window is, fused or separate gather,
LDS or register staging ...) must leave every variable and optimizer slot BIT-identical; a knob that changes a summation
order (tile shapes of the MFMA kernels) must stay within fp32 rounding of it.  One subprocess per setting: most knobs are read
once per process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_cache = {}


def _run(model, B, steps, env_extra, bf16=False):
    key = (model, B, steps, bf16, tuple(sorted(env_extra.items())))
    if key not in _cache:
        # launch-form toggles travel in ONE variable since round 6 (recsys_amd/_lib.py FORMS): "RSX_<NAME>" keys of the tables
        # below whose name is a launch form are folded into RSX_FORMS; the rest (kernel variants read by librsx.so, real knobs)
        # stay environment variables of their own
        from recsys_amd._lib import FORMS
        forms = {k[4:].lower(): v for k, v in env_extra.items() if k[4:].lower() in FORMS}
        rest = {k: v for k, v in env_extra.items() if k[4:].lower() not in FORMS}
        if forms:
            rest["RSX_FORMS"] = ",".join("%s=%s" % kv for kv in sorted(forms.items()))
        env = dict(os.environ, **rest)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "knob_worker.py"), model, str(B), str(steps), "1" if bf16 else "0"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        _cache[key] = json.loads(r.stdout.strip().splitlines()[-1])
    return _cache[key]


BIT_IDENTICAL = [
    ("deepfm", 256, {"RSX_FUSE_GATHER": "0"}),                # gather as its own launch
    ("deepfm", 256, {"RSX_ADAM_WINDOW": "1"}),                # no optimizer windows: every step sorts and carries its sweep slices
    ("deepfm", 256, {"RSX_ADAM_WINDOW": "4"}),
    ("deepfm", 256, {"RSX_ADAM_WINDOW": "1", "RSX_SORT_IN_GATHER": "1", "RSX_FUSE_GATHER": "0"}),   # the sort rides in the gather launch
    ("deepfm", 256, {"RSX_ADAM_WINDOW": "1", "RSX_SORT_RIDE_MAX": "0"}),                            # stand-alone sort launch
    ("deepfm", 256, {"RSX_ADAM_WINDOW": "1", "RSX_SWEEP_WEIGHTS": "0,1,1,2,2,1"}),                  # other shares of the sweep per launch
    ("deepfm", 256, {"RSX_WIN_SEPARATE_SORTS": "1"}),         # k single sorts instead of the multi-sort
    ("deepfm", 256, {"RSX_SEG_LDS": "0"}),                    # the scatter's register form
    ("fm", 256, {"RSX_ADAM_WINDOW": "1"}),
    ("dcn", 1024, {"RSX_ADAM_WINDOW": "1"}),
    ("dcn", 1024, {"RSX_TOWER_RTW": "1"}),                    # row tiles per d(input) workgroup
    ("xdeepfm", 128, {"RSX_ADAM_WINDOW": "1"}),
    ("xdeepfm", 128, {"RSX_XDFM_SORT_RIDE": "0"}),
    # round 4
    ("dcn", 4096, {"RSX_ADAM_WINDOW": "1"}),                  # stand-alone sort launches of 4 096 keys: several workgroups per field ..
    ("dcn", 4096, {"RSX_ADAM_WINDOW": "1", "RSX_SORT_SPLIT": "0"}),    # .. and one workgroup per field: the same bits
    ("din", 64, {"RSX_DIN_SIDE_SORT": "0"}),                  # din.py: the ids-only branch (sort + sweep) in line instead of on a side stream
    ("dcn", 1024, {"RSX_GATHER_CROSS": "0"}),                 # the lookup and the cross layers' forward as two launches
    ("dcn", 4096, {"RSX_CROSS_RIDE": "0"}),                   # round 6: the cross layers' backward as a launch of its own instead of riding in the second tower layer's backward launch
    ("dcn", 4096, {"RSX_SCATTER_RIDERS": "0"}),               # the dW / cross-gradient reduces as their own launches instead of riding in the scatter's stage A
    ("din", 64, {"RSX_SCATTER_RIDERS": "0"}),                 # the attention blocks' weight-gradient reduces inside the finish launch
    ("din", 64, {"RSX_DIN_GATHER_RIDE": "0"}),                # the six lookups as their own launch instead of riding in the two prepare launches
    ("din", 64, {"RSX_MLP_REDUCE_RIDE": "0"}),                # the mlp_layer's weight-gradient reduce as its own launch instead of riding in the pooling backward
    # round 5
    ("xdeepfm", 128, {"RSX_CIN_GATHER_RIDE": "0"}),           # xdeepfm.py's lookup as its own launch instead of riding in the CIN filter preparation
    ("xdeepfm", 128, {"RSX_CIN_DX0_RIDE": "0"}),              # the dX0 tile reduce as its own launch instead of riding in the CIN weight-gradient launch
]
ROUNDING = [
    ("fm", 256, {"RSX_FM_FUSE": "0"}),                        # fp64 reduction of the head's dense gradients instead of the grouped fp32 rows
    ("dcn", 1024, {"RSX_CROSS_BWD4": "0"}),                   # one-wave cross backward: its partials are added in another order
    ("deepfm", 256, {"RSX_TOWER_DXG": "0"}),                  # one-tile d(input) workgroups (the default until round 6): another order over the N outputs
    ("dcn", 1024, {"RSX_TOWER_BIG": "0"}),                    # the batch-256 tiles for the wide first layer
    ("dcn", 1024, {"RSX_TOWER_DXG_SPLIT": "0"}),
    ("dcn", 1024, {"RSX_TOWER_SB_ROWS": "256"}),              # dW row blocks of 256 instead of 512 rows
    ("xdeepfm", 128, {"RSX_CIN_SPLIT_DEFAULT": "0", "RSX_CIN_DX": "1"}),   # the fp32 MFMA CIN kernels, register form of their dX kernel
    # round 5: the CIN modes against the default (mode 4: two scaled fp16 planes forward / data gradients)
    ("xdeepfm", 128, {"RSX_CIN_SPLIT_DEFAULT": "0"}),         # the fp32 MFMA kernels of csrc/cin.hip
    ("xdeepfm", 128, {"RSX_CIN_SPLIT_DEFAULT": "3"}),         # three bf16 planes per operand, first-form kernels
    ("xdeepfm", 128, {"RSX_CIN_DX_FSPLIT": "0"}),             # the first CIN layer's data gradients: one workgroup per tile over all fields
    # round 4
    ("din", 64, {"RSX_MLP_FUSE": "0"}),                       # din.py's 'mlp_layer' as 8 launches instead of the one-launch form
    ("dcn", 4096, {"RSX_TOWER_BIG_MIN_K_BWD": "256"}),        # the batch-256 backward tiles for the 100-wide layer at batch 4 096
]


@pytest.mark.parametrize("model,B,env", BIT_IDENTICAL, ids=lambda x: "+".join("%s=%s" % kv for kv in x.items()) if isinstance(x, dict) else str(x))
def test_scheduling_knobs_leave_every_bit_unchanged(model, B, env):
    base = _run(model, B, 12, {})
    got = _run(model, B, 12, env)
    assert got["digest"] == base["digest"], (env, got["loss"], base["loss"], got["dense_abs_sum"], base["dense_abs_sum"])


@pytest.mark.parametrize("model,B,env", ROUNDING, ids=lambda x: "+".join("%s=%s" % kv for kv in x.items()) if isinstance(x, dict) else str(x))
def test_tiling_knobs_stay_within_fp32_rounding(model, B, env):
    base = _run(model, B, 12, {})
    got = _run(model, B, 12, env)
    assert abs(got["loss"] - base["loss"]) < 1e-5, (env, got["loss"], base["loss"])
    np.testing.assert_allclose(np.array(got["dense"]), np.array(base["dense"]), rtol=1e-4, atol=2e-6)
