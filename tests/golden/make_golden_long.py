#!/usr/bin/env python
"""Generates tests/golden/long_<config>.npz: the fp64 oracle's 200-step TRAIN run + EVAL over 20 held-out batches for deepfm
bs 256 and xdeepfm bs 256 CIN 128-128 (tests/longrun.py).  Inputs are regenerated from the seed on both sides and pinned by a
sha256 digest; the fixture stores expected outputs only (data, no code).

usage: python tests/golden/make_golden_long.py [--f32] [config ...]      (deepfm ~1 min, xdeepfm ~15 min in the build container)

--f32: the SAME oracle evaluated in float32 -> long_<config>_f32ref.npz (scalars + the 200 train losses only).  It is not an
expected output: it measures how far ANY float32 evaluation of these 200 steps lands from the fp64 one.  TF-1 Adam divides by
sqrt(v) + 1e-8, so a gradient component near the epsilon scale turns an absolute rounding difference of 1e-9 into a variable
difference of ~1e-5 in one step, and 200 steps compound it: the fp32 and fp64 oracles agree to 1e-7 on the first ~30 losses and
to ~1e-3 on the last ones.  tests/test_gpu_long.py holds the HIP path to the fp64 run within a small multiple of that floor.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests import longrun  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if a != "--f32"]
    f32 = "--f32" in sys.argv[1:]
    for name in argv or list(longrun.CONFIGS):
        t0 = time.time()
        P, train, ev, digest = longrun.make_inputs(name)
        if f32:
            out = longrun.oracle_run(name, P, train, ev, dtype=np.float32,
                                     progress=lambda i, l: print("  %s f32 step %d loss %.6f (%.0f s)" % (name, i, l, time.time() - t0), flush=True))
            path = os.path.join(HERE, "long_%s_f32ref.npz" % name)
            np.savez_compressed(path, digest=np.array(digest), train_losses=out["train_losses"], eval_losses=out["eval_losses"],
                                eval_loss=np.asarray(out["eval_loss"]), auc=np.asarray(out["auc"]), accuracy=np.asarray(out["accuracy"]))
            print("%s f32: %.1f s, eval loss %.7f AUC %.7f accuracy %.5f" % (name, time.time() - t0, out["eval_loss"], out["auc"], out["accuracy"]), flush=True)
            continue
        out = longrun.oracle_run(name, P, train, ev, progress=lambda i, l: print("  %s step %d loss %.6f (%.0f s)" % (name, i, l, time.time() - t0), flush=True))
        dense = out.pop("final_dense")
        path = os.path.join(HERE, "long_%s.npz" % name)
        np.savez_compressed(path, digest=np.array(digest), **{k: np.asarray(v) for k, v in out.items()},
                            **{"final." + k: v for k, v in dense.items()})
        print("%s: %.1f s, %d KB, eval loss %.7f AUC %.7f accuracy %.5f; train loss first/last %.5f / %.5f"
              % (name, time.time() - t0, os.path.getsize(path) // 1024, out["eval_loss"], out["auc"], out["accuracy"],
                 out["train_losses"][0], out["train_losses"][-1]), flush=True)


if __name__ == "__main__":
    main()
