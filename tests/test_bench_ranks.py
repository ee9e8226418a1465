"""CPU: `python bench.py --gpus 2`, launched BARE as the driver launches the single-GPU bench, must come back as a two-rank run: the script re-executes
itself under torch.distributed.run, both ranks take part in the collective (the line's n_gpus is an all-reduce of ones), the index reaches rank 1 by
broadcast, and each rank times its own shard.  Here the collective backend is gloo and the device is the mock-runtime build of the product source (the
two switches bench.py reads from the environment for exactly this test); on the GPU box the same entry runs over RCCL."""
import json
import os
import subprocess
import sys

import hostsim_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, tmp_path, extra_env=None, timeout=900):
    env = dict(os.environ, BWA_AMD_BENCH_BACKEND="gloo", BWA_AMD_BENCH_LIB=hostsim_build.build(), BWA_AMD_CACHE=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


SMALL = ["--steps", "2", "--warmup", "1", "--reads", "8", "--genome-mbp", "0.05", "--streams", "1", "--dense-sa", "0",
         "--no-cpu-baseline", "--no-e2e", "--no-pmc", "--no-longread", "--variants", ""]


def test_bench_gpus_2_runs_two_ranks(tmp_path):
    p = _run(["--gpus", "2"] + SMALL, tmp_path)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints the one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"]["n"] == 2 and len(out["ranks"]["per_rank_Mreads_s"]) == 2
    assert all(v > 0 for v in out["ranks"]["per_rank_Mreads_s"])
    assert out["index_broadcast"]["ranks"] == 2
    assert out["scaling"] == "weak" and out["config"]["reads_per_gpu"] == 8 and "reads x2" in out["config"]["sharding"]
    # whole-job rate = both ranks' reads over the slowest rank's time: never more than the sum of the ranks' own rates
    assert out["value"] <= sum(out["ranks"]["per_rank_Mreads_s"]) * 1.001
    assert len(lines[0]) < 8192


def test_bench_world_size_must_equal_gpus(tmp_path):
    p = _run(["--gpus", "2"] + SMALL, tmp_path, extra_env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE is 1" in (p.stderr + p.stdout)


def test_bench_gpus_1_is_one_process(tmp_path):
    p = _run(["--gpus", "1"] + SMALL, tmp_path)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and "ranks" not in out and "index_broadcast" not in out
