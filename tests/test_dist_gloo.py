"""N>1 path on CPU: two gloo ranks shard a batch of frames with the bench's rule, never exchange frame
data, and reduce only their elapsed time (MAX) and frame counts (SUM)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %r)
from aswstereomatch_amd.dist import Group, frames_for_rank
g = Group(backend="gloo")
mine = frames_for_rank(13, g.rank, g.world)
g.barrier()
elapsed = 0.25 if g.rank == 0 else 0.75   # pretend rank 1 is the straggler
tmax = g.max_over_ranks(elapsed)
total = g.sum_over_ranks(len(mine))
print(json.dumps({"rank": g.rank, "world": g.world, "mine": mine, "tmax": tmax, "total": total}), flush=True)
g.close()
''' % ROOT


def test_two_rank_sharding_and_timing(tmp_path):
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda r: r["rank"])
    assert outs[0]["mine"] == [0, 2, 4, 6, 8, 10, 12] and outs[1]["mine"] == [1, 3, 5, 7, 9, 11]
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == list(range(13))       # every frame exactly once
    assert outs[0]["tmax"] == outs[1]["tmax"] == 0.75                          # MAX over ranks
    assert outs[0]["total"] == outs[1]["total"] == 13.0


def test_single_rank_is_a_noop():
    sys.path.insert(0, ROOT)
    from aswstereomatch_amd.dist import Group, frames_for_rank

    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        g = Group()
        assert g.world == 1 and g.max_over_ranks(1.5) == 1.5 and frames_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
        g.barrier()
        g.close()
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
