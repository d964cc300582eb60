"""bench.py's N > 1 path (row blocks per rank, all-gather of the pool shards, all-reduce of the
conflict counts, max-over-ranks timing) rehearsed with two ranks on ONE card: the collectives go
through gloo (RCCL refuses two ranks per device), the row blocks through the HIP engine.  The merged
per-primer counts must equal the single-process run's."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _bench_line(cmd, env):
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]      # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("pool", ["4000", "3001"])        # 3001: odd, the two row blocks differ by one row
def test_two_ranks_equal_one_rank(pool):
    common = ["--steps", "1", "--warmup", "1", "--pool", pool, "--no-cpu-baseline"]
    env = dict(os.environ)
    one = _bench_line([sys.executable, "bench.py", "--gpus", "1"] + common, env)
    env2 = dict(env, MSSPE_BENCH_BACKEND="gloo", MSSPE_BENCH_DEVICE="0")
    port = 29600 + os.getpid() % 300
    two = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                       "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2"]
                      + common, env2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    for key in ("pool", "checks_per_step", "conflicts", "conflict_checksum"):
        assert one["config"][key] == two["config"][key], key
    assert one["config"]["conflicts"] > 0
    assert two["cpu_baseline"] is None          # rank 0 at N = 1 only
    assert two["value"] > 0 and two["roofline"]["launches"] >= 1
    km = two["roofline"]["kernel_ms_per_rank"]
    assert 0 < km["min"] <= km["max"]


def test_gpus_flag_must_match_the_launcher():
    """`--gpus 2` inside a one-rank launch is refused instead of reporting a one-GPU number as two."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--pool", "512",
                          "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
