"""Checkpoint / resume: what Estimator's RunConfig(save_checkpoints_steps, keep_checkpoint_max=5)
plus model_dir auto-resume give the reference scripts (fm/fm.py:187-194,204-209).  Files are
`model.ckpt-<global_step>.pt` holding tables, first-order weights, dense arena, Adam slots and beta powers."""
import glob
import os
import re

import torch


def _list(model_dir):
    out = []
    for p in glob.glob(os.path.join(model_dir, "model.ckpt-*.pt")):
        m = re.search(r"model\.ckpt-(\d+)\.pt$", p)
        if m:
            out.append((int(m.group(1)), p))
    return sorted(out)


def save(model_dir, store, global_step, keep_max=5):
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, "model.ckpt-%d.pt" % global_step)
    tmp = path + ".tmp.%d" % os.getpid()
    torch.save(store.state_dict(), tmp)
    os.replace(tmp, path)
    for _, p in _list(model_dir)[:-keep_max]:
        os.remove(p)
    return path


def latest(model_dir):
    ck = _list(model_dir) if model_dir and os.path.isdir(model_dir) else []
    return ck[-1][1] if ck else None


def restore_latest(model_dir, store):
    p = latest(model_dir)
    if p is None:
        return None
    store.load_state_dict(torch.load(p, map_location="cpu"))
    if store.dp is None or getattr(store.dp, "rank", 0) == 0:
        print("INFO:Restoring parameters from %s" % p, flush=True)
    return p
