"""Worker of tests/test_spawn_cpu.py: one rank started by recsys_amd.dist.spawn_local_ranks.  Joins the process group the way
every rank of the product does (dist.init_process_group reads RANK / WORLD_SIZE / MASTER_*), all-reduces its rank and writes
what it saw."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

from recsys_amd import dist  # noqa: E402


def main():
    out_dir = sys.argv[1]
    assert dist.local_replica_count() == 0, "a spawned rank must never spawn again"
    dist.init_process_group()
    t = torch.tensor([float(td.get_rank() + 1)])
    td.all_reduce(t)
    with open(os.path.join(out_dir, "rank%d.json" % td.get_rank()), "w") as f:
        json.dump({"rank": td.get_rank(), "world": td.get_world_size(), "sum": float(t.item()), "backend": td.get_backend(),
                   "args": sys.argv[2:]}, f)
    td.destroy_process_group()


if __name__ == "__main__":
    main()
