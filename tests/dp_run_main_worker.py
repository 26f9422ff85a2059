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
    seen, trained = [], []
    real_input_fn = driver.input_fn

    def spy(filenames, batch_size, num_epochs=-1, need_shuffle=False, *a, **kw):
        it = real_input_fn(filenames, batch_size, num_epochs, need_shuffle, *a, **kw)
        train = need_shuffle
        for feats, lab in it:
            if train:            # c13's log value is unique per record in the synthetic shards: a record fingerprint
                seen.append(np.round(feats["cont_log"][:, 12].astype(np.float64) * 1e6).astype(np.int64))
                trained.append((np.array(feats["ids"]), np.array(lab)))      # what this rank's step len(trained) - 1 trains on
            yield feats, lab

    driver.input_fn = spy
    import recsys_amd.estimator as E
    made = []
    real_init = E.Estimator.__init__

    def init_spy(self, *a, **kw):
        real_init(self, *a, **kw)
        made.append(self)

    E.Estimator.__init__ = init_spy
    # the variables as model_fn creates them (before the first step): the parent replays the run through the oracle from here
    init_vars = {}
    real_build = E.VariableStore.build

    def build_spy(self, embeddings, *a, **kw):
        real_build(self, embeddings, *a, **kw)
        if mod == "deepfm" and not init_vars:
            ar = embeddings["input_layer"]
            init_vars.update(tables=ar.tables.cpu().numpy().copy(), w1=ar.w1.cpu().numpy().copy(),
                             **{k: p.detach().cpu().numpy().copy() for k, p in self.dense.params.items()})

    E.VariableStore.build = build_spy
    res = m.main(["--train_path", data_dir, "--train_parts", "4", "--eval_parts", "1", "--batch_size", "64", "--model_dir",
                  model_dir, "--save_checkpoints_steps", "6", "--log_steps", "3", "--dropout", "0.0", "--learning_rate",
                  "0.001", "--task_type", "train", "--num_epochs", "2", "--mirror", "true"])
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
    if mod == "deepfm":
        ar = est.store.embeddings["input_layer"]
        np.savez(os.path.join(out_dir, "vars_rank%d.npz" % rank),
                 **{"init." + k: v for k, v in init_vars.items()},
                 **{"final.tables": ar.tables.cpu().numpy(), "final.w1": ar.w1.cpu().numpy()},
                 **{"final." + k: p.detach().cpu().numpy() for k, p in est.store.dense.params.items()},
                 ids=np.stack([t[0] for t in trained]), labels=np.stack([t[1] for t in trained]))
    json.dump({"rank": rank, "digest": h.hexdigest(), "seen": np.concatenate(seen).tolist() if seen else [],
               "batches": len(seen), "res": {k: float(v) for k, v in res.items()}, "global_step": int(est.global_step)},
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
    print("WORKER_OK", rank, flush=True)


if __name__ == "__main__":
    main()
