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


def test_two_ranks_equal_one_rank():
    common = ["--steps", "1", "--warmup", "1", "--pool", "4000", "--no-cpu-baseline"]
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
