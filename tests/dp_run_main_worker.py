"""Worker of tests/test_gpu_dp_run_main.py: one rank of a world-N run of the SCRIPT-LEVEL data-parallel path
(`recsys_amd.<model>.main` with --mirror true under torchrun-style env).  Records which records this rank consumed and a
digest of every variable after training."""
import hashlib
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    mod, data_dir, model_dir, out_dir = sys.argv[1:5]
    m = importlib.import_module("recsys_amd." + mod)
    from recsys_amd import deepfm as driver
    rank = int(os.environ["RANK"])
    seen = []
    real_input_fn = driver.input_fn

    def spy(filenames, batch_size, num_epochs=-1, need_shuffle=False, *a, **kw):
        it = real_input_fn(filenames, batch_size, num_epochs, need_shuffle, *a, **kw)
        train = need_shuffle
        for feats, lab in it:
            if train:            # c13's log value is unique per record in the synthetic shards: a record fingerprint
                seen.append(np.round(feats["cont_log"][:, 12].astype(np.float64) * 1e6).astype(np.int64))
            yield feats, lab

    driver.input_fn = spy
    import recsys_amd.estimator as E
    made = []
    real_init = E.Estimator.__init__

    def init_spy(self, *a, **kw):
        real_init(self, *a, **kw)
        made.append(self)

    E.Estimator.__init__ = init_spy
    res = m.main(["--train_path", data_dir, "--train_parts", "4", "--eval_parts", "1", "--batch_size", "64", "--model_dir",
                  model_dir, "--save_checkpoints_steps", "6", "--log_steps", "3", "--dropout", "0.0", "--learning_rate",
                  "0.01", "--task_type", "train", "--num_epochs", "2", "--mirror", "true"])
    # digest of every variable of this replica
    est = made[-1]
    h = hashlib.sha256()
    sd = est.store.state_dict()
    for k in sorted(sd):
        v = sd[k]
        items = sorted(v.items()) if isinstance(v, dict) else [("", v)]
        for kk, t in items:
            h.update(k.encode() + kk.encode())
            h.update(t.detach().cpu().contiguous().numpy().tobytes())
    json.dump({"rank": rank, "digest": h.hexdigest(), "seen": np.concatenate(seen).tolist() if seen else [],
               "batches": len(seen), "res": {k: float(v) for k, v in res.items()}, "global_step": int(est.global_step)},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
    print("WORKER_OK", rank, flush=True)


if __name__ == "__main__":
    main()
