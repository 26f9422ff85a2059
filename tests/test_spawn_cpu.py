"""One command -> all local GPUs (fm/fm.py:184-186 `tf.distribute.MirroredStrategy()`,
deepfm/readme.md:22-24): the launcher logic of recsys_amd/dist.py -- when a process becomes the launcher, the command it
starts, the environment of the ranks -- and a real world-2 launch on CPU (gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from recsys_amd import dist  # noqa: E402


def test_local_replica_count_rules():
    assert dist.local_replica_count(n_devices=8, env={}) == 8
    assert dist.local_replica_count(n_devices=1, env={}) == 0            # one GPU: train in-process
    assert dist.local_replica_count(n_devices=0, env={}) == 0
    assert dist.local_replica_count(n_devices=8, env={"WORLD_SIZE": "8"}) == 0     # already a rank of a launched job
    assert dist.local_replica_count(n_devices=8, env={"WORLD_SIZE": "1"}) == 0
    assert dist.local_replica_count(n_devices=8, env={"RSX_FORCE_DIST": "1"}) == 0
    assert dist.local_replica_count(n_devices=1, env={"RSX_MIRROR_REPLICAS": "2"}) == 2
    assert dist.local_replica_count(n_devices=8, env={"RSX_MIRROR_REPLICAS": "1"}) == 0


def test_spawn_command_and_env():
    cmd = dist.spawn_command(4, ["--gpus", 4, "--steps", 20], script="/x/bench.py", port=12345, python="py")
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                   "--master-port", "12345", "/x/bench.py", "--gpus", "4", "--steps", "20"]
    cmd = dist.spawn_command(2, ["--task_type", "train"], module="recsys_amd.deepfm", port=1, python="py")
    assert cmd[-4:] == ["-m", "recsys_amd.deepfm", "--task_type", "train"]
    with pytest.raises(AssertionError):
        dist.spawn_command(2, [], script="a", module="b")
    # more ranks than devices: RCCL refuses two ranks on one device -> gloo, unless the caller chose
    assert dist.spawn_env(2, n_devices=1, env={})["RSX_DIST_BACKEND"] == "gloo"
    assert "RSX_DIST_BACKEND" not in dist.spawn_env(8, n_devices=8, env={})
    assert dist.spawn_env(2, n_devices=1, env={"RSX_DIST_BACKEND": "nccl"})["RSX_DIST_BACKEND"] == "nccl"
    e = dist.spawn_env(2, n_devices=2, env={"RSX_MIRROR_REPLICAS": "2"})
    assert "RSX_MIRROR_REPLICAS" not in e and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    p, q = dist.free_port(), dist.free_port()
    assert 1024 < p < 65536 and 1024 < q < 65536


def test_spawn_two_ranks_gloo(tmp_path):
    """A real launch: 2 ranks on this CPU host, each joins the group through dist.init_process_group and sees world 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "RSX_FORCE_DIST")}
    code = ("import sys; sys.path.insert(0, %r); from recsys_amd import dist; "
            "sys.exit(dist.spawn_local_ranks(2, [%r, 'a', 'b'], script=%r))"
            % (ROOT, str(tmp_path), os.path.join(ROOT, "tests", "spawn_worker.py")))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    assert [s["rank"] for s in seen] == [0, 1] and all(s["world"] == 2 and s["sum"] == 3.0 for s in seen)
    assert all(s["args"] == ["a", "b"] and s["backend"] == "gloo" for s in seen)


def test_mirror_flag_spawns_only_when_asked(monkeypatch):
    """--mirror true + several local GPUs -> the script becomes the launcher; --mirror false or one GPU -> in-process."""
    calls = []
    monkeypatch.setattr(dist, "spawn_local_ranks", lambda n, args, script=None, module=None: calls.append((n, list(args), module)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RSX_FORCE_DIST", raising=False)

    class F:
        mirror = True
    monkeypatch.setenv("RSX_MIRROR_REPLICAS", "4")
    assert dist.maybe_spawn_mirror(F, "recsys_amd.dcn", ["--task_type", "train"]) == 0
    assert calls == [(4, ["--task_type", "train"], "recsys_amd.dcn")]
    F.mirror = False
    assert dist.maybe_spawn_mirror(F, "recsys_amd.dcn", []) is None
    F.mirror = True
    monkeypatch.setenv("RSX_MIRROR_REPLICAS", "1")
    assert dist.maybe_spawn_mirror(F, "recsys_amd.dcn", []) is None
    monkeypatch.setenv("RSX_MIRROR_REPLICAS", "4")
    monkeypatch.setenv("WORLD_SIZE", "4")                     # a rank never spawns
    assert dist.maybe_spawn_mirror(F, "recsys_amd.dcn", []) is None
    assert len(calls) == 1
