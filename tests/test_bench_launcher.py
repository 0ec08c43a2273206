"""bench.py --gpus N must run N real processes.  No GPU here: `--dry-run` swaps the HIP hot path for a
numpy checksum and the process group runs on gloo, so the launcher, the rank environment, the barrier / MAX-over-ranks
timing and the one-JSON-line contract are exercised exactly as on the GPU box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--dry-run", "--backend", "gloo", "--width", "96", "--height", "64", "--disp", "8", "--frames", "3", "--steps", "2",
         "--warmup", "1"]


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _json_line(stdout):
    lines = stdout.strip().splitlines()
    # rank 0 prints ONE JSON line and NOTHING else on stdout (the banners gloo / RCCL print while a process group is built are
    # sent to stderr, aswstereomatch_amd/dist.py); the other ranks print nothing
    assert len(lines) == 1 and lines[0].startswith("{"), stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    pids = out["config"]["rank_pids"]
    assert len(pids) == 2 and len(set(pids)) == 2 and os.getpid() not in pids  # two real child processes
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["higher_is_better"] is True
    # whole-job value: W*H * frames * steps * ranks / elapsed
    assert abs(out["value"] - 96 * 64 * 3 * 2 * 2 / (out["ms_per_step"] * 2 / 1e3) / 1e6) <= 0.02 * out["value"] + 1e-3


def test_gpus_8_dry_run_is_eight_processes_and_says_how_it_was_synchronised():
    # the shape of the first real 8-GPU line, rehearsed on CPU: eight ranks, eight distinct processes; a dry run never claims RCCL
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"] + SMALL, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["ranks"] == 8 and len(cfg["rank_pids"]) == 8 and len(set(cfg["rank_pids"])) == 8
    assert cfg["backend"] == "gloo" and cfg["rccl_ok"] is None and cfg["rccl_error"] is None  # gloo was asked for: nothing fell back
    assert cfg["devices"] == [-1] * 8  # no device in a dry run; on hardware: [0, 1, ..., 7]


def test_under_an_external_launcher_each_process_is_one_rank():
    # what `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` gives every process
    procs = []
    for rank in range(2):
        env = dict(_clean_env(), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert _json_line(outs[0][0])["n_gpus"] == 2
    assert "{" not in outs[1][0]  # rank 1 prints no JSON (gloo itself may print a connection notice)


def test_single_rank_default():
    r = subprocess.run([sys.executable, BENCH] + SMALL, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and len(out["config"]["rank_pids"]) == 1


def test_world_size_mismatch_fails_loudly():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_a_failing_rank_aborts_the_job():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--fail-rank", "1"] + SMALL, env=_clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "{" not in r.stdout  # no JSON line from a broken job


def test_without_a_gpu_the_real_bench_refuses():
    # the product path has no CPU fallback: on this GPU-less host the non-dry run must exit with a message, not a number
    import torch

    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0"], env=_clean_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "needs a GPU" in r.stderr and "{" not in r.stdout
