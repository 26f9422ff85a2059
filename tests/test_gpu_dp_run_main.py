"""World-2 run of the SCRIPT-LEVEL data-parallel path on one GPU (two processes on cuda:0, gloo collectives -- RCCL
refuses two ranks on one device): `deepfm.main(--mirror true)` exactly as torchrun would start it.  Asserts what
VERDICT r1 found broken: the two ranks consume DISJOINT records (rank r gets batches r, r+2, ...), run the same number of
steps, only rank 0 writes checkpoints, and both replicas end with BIT-IDENTICAL variables; the summed eval counters
give both ranks the same AUC."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shards(d, n_files=4, per_file=400, seed=3):
    from recsys_amd import synthetic
    from recsys_amd.input_pipeline import write_criteo_shard
    rng = np.random.default_rng(seed)
    tag = 0
    for k in range(n_files):
        label, cont, cat = synthetic.criteo_raw_batch(rng, per_file)
        label = ((cont[:, 0] > 8) ^ (rng.random(per_file) < 0.1)).astype(np.float32)
        cont[:, 12] = np.arange(tag, tag + per_file) + 1.0           # _c13 = record number + 1 (unique fingerprint)
        tag += per_file
        write_criteo_shard(os.path.join(d, "part-r-%05d" % k), label, cont, cat)


@pytest.mark.parametrize("mod", ["deepfm", "xdeepfm"])
def test_run_main_world2_disjoint_records_identical_replicas(tmp_path, mod):
    d = str(tmp_path) + "/"
    _shards(d)
    out = tmp_path / "out"
    out.mkdir()
    model_dir = str(tmp_path / "model")
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK="0", RSX_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_run_main_worker.py"), mod, d, model_dir,
                                       str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all("WORKER_OK" in o for o in logs), "\n=====\n".join(o[-3000:] for o in logs)
    r = [json.load(open(out / ("rank%d.json" % k))) for k in range(2)]
    # same number of steps, disjoint records, and rank r holds batches r, r+2, ... of the 3 training shards (1200 records)
    assert r[0]["batches"] == r[1]["batches"] > 0 and r[0]["global_step"] == r[1]["global_step"] == r[0]["batches"]
    s0, s1 = set(r[0]["seen"]), set(r[1]["seen"])
    assert not (s0 & s1), "ranks trained on the same records"
    per_epoch = (1200 // (64 * 2)) * 64            # complete rounds only
    assert len(r[0]["seen"]) == len(r[1]["seen"]) == 2 * per_epoch
    assert len(s0) == per_epoch and len(s1) == per_epoch          # epoch 2 revisits the rank's own records
    # bit-identical replicas
    assert r[0]["digest"] == r[1]["digest"]
    # evaluate(): counters summed across ranks -> the same numbers everywhere
    assert r[0]["res"]["AUC"] == r[1]["res"]["AUC"] and r[0]["res"]["loss"] == r[1]["res"]["loss"]
    # only the chief wrote checkpoints, no stray tmp files
    assert glob.glob(model_dir + "/model.ckpt-*.pt") and not glob.glob(model_dir + "/*.tmp*")
    # rank 1 stayed quiet
    assert "INFO:loss" in logs[0] and "INFO:loss" not in logs[1]
    if mod != "deepfm":
        return
    # ... and the two real processes trained what the ORACLE trains from the same initial variables on the same per-rank batches
    # (oracle.models.train_step_dp: MirroredStrategy's step -- per-replica batch-norm, losses scaled 1/N, dense gradients summed,
    # IndexedSlices concatenated in replica order).  Two ranks that are identically WRONG pass the digest comparison above;
    # they do not pass this one.
    from oracle import criteo, models, nn
    v = [np.load(out / ("vars_rank%d.npz" % k)) for k in range(2)]
    for k in v[0].files:
        if k.startswith("init."):
            assert np.array_equal(v[0][k], v[1][k]), k            # same seed -> same initial variables on both ranks
    steps = v[0]["ids"].shape[0]
    assert steps == v[1]["ids"].shape[0] == r[0]["global_step"] and not np.array_equal(v[0]["ids"], v[1]["ids"])
    P = {k[5:]: v[0][k].astype(np.float64) for k in v[0].files if k.startswith("init.")}
    om = models.DeepFM(P, criteo.row_offsets(), 2, 0.0)
    opt = nn.AdamTF1(lr=1e-3, dtype=np.float64)
    for i in range(steps):
        models.train_step_dp(om, opt, [(v[0]["ids"][i],), (v[1]["ids"][i],)],
                             [v[0]["labels"][i].astype(np.float64).reshape(-1), v[1]["labels"][i].astype(np.float64).reshape(-1)])
    err = {k: float(np.abs(v[0]["final." + k].astype(np.float64).reshape(P[k].shape) - P[k]).max()) for k in P}
    assert max(err.values()) < 1e-4, err
    moved = float(np.abs(v[0]["final.tables"] - v[0]["init.tables"]).max())
    assert moved > 5e-3, moved                                     # (the variables did move: 18 steps at lr 1e-3)


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR",
                                                            "RSX_FORCE_DIST")}
    env.update(extra)
    return env


def test_bench_gpus2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the shape of the driver's N = 1 command): the process becomes the
    launcher of two ranks (one GPU here -> gloo, chosen by dist.spawn_env) and rank 0 prints the one JSON line (VERDICT r5
    item 2: this used to exit with 'launch with torchrun')."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no_cpu_baseline",
           "--repeats", "2"]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        # One failure in ten runs of the whole suite (round 6, not reproduced in 8 + 3 x 72 targeted runs): two ranks sharing ONE
        # GPU over gloo start through torch.distributed.run's rendezvous on a port picked a moment earlier.  Keep the evidence,
        # try once more -- a second failure fails the test.
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_gpus2_first_failure.txt"), "w") as f:
            f.write("rc %d\n---- stdout\n%s\n---- stderr\n%s\n" % (r.returncode, r.stdout[-6000:], r.stderr[-12000:]))
        print("bench.py --gpus 2: first attempt failed (rc %d), stderr tail:\n%s" % (r.returncode, r.stderr[-1500:]))
        r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["config"]["global_batch"] == 512
    assert len(d["rank_ms_per_step"]) == 2 and d["backend"] == "gloo" and d["ranks_per_gpu"] == 2
    assert d["ms_per_step"] == max(d["rank_ms_per_step"]) and d["value"] > 0


def test_mirror_true_spawns_one_rank_per_replica(tmp_path):
    """`python -m recsys_amd.fm --mirror true` with several local replicas (RSX_MIRROR_REPLICAS=2 stands for two visible GPUs on
    this one-GPU box): ONE command trains data-parallel, as `MirroredStrategy()` does (fm/fm.py:184-186)."""
    d = str(tmp_path) + "/"
    _shards(d)
    model_dir = str(tmp_path / "model")
    cmd = [sys.executable, "-m", "recsys_amd.fm", "--task_type", "train", "--train_path", d, "--train_parts", "4", "--eval_parts", "1",
           "--batch_size", "64", "--num_epochs", "1", "--model_dir", model_dir, "--mirror", "true", "--log_steps", "4"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(RSX_MIRROR_REPLICAS="2"), capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and not glob.glob(model_dir + "/model.ckpt-*.pt"):      # (see test_bench_gpus2_spawns_its_own_ranks)
        print("recsys_amd.fm --mirror true: first attempt failed before training (rc %d):\n%s" % (r.returncode, r.stderr[-1500:]))
        r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(RSX_MIRROR_REPLICAS="2"), capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "2 data-parallel ranks" in out and "INFO:loss" in out
    assert glob.glob(model_dir + "/model.ckpt-*.pt")
