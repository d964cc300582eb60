"""The multi-rank pipeline harness (msspe_amd.pipeline_ranks; BASELINE.json configs[4]'s shape: stage A on
rank 0, stage B and stage C sharded, vertex cover and CSV on rank 0) must write the same CSV as the
single-process CLI, whatever the number of ranks.  Two and three ranks are rehearsed on ONE card with gloo
collectives (RCCL refuses several ranks per device); the sharding logic is the same."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 3])
def test_ranks_write_the_cli_csv(tmp_path, world):
    import msspe_amd
    g = msspe_amd.synth.aligned_genomes(40, 4200, seed=7)
    fasta = "".join(f">genome{i} synthetic\n{bytes(r).decode()}\n" for i, r in enumerate(g))
    fa, want_csv, got_csv = tmp_path / "in.fa", tmp_path / "cli.csv", tmp_path / "ranks.csv"
    fa.write_text(fasta)
    msspe_amd.load_library()
    host = C.CDLL(str(ROOT / "open-msspe-design_amd" / "libod_msspe_host.so"))
    flags = ["--max-iterations", "120", "--delta-g-threshold", "-6000"]
    args = ["od-msspe-hip", "-i", str(fa), "-o", str(want_csv), "--do-align", "false"] + flags
    arr = (C.c_char_p * len(args))(*[x.encode() for x in args])
    buf = C.create_string_buffer(1 << 20)
    assert host.odm_run_cli(len(args), arr, buf, 1 << 20) == 0, buf.value.decode()
    env = dict(os.environ, PYTHONPATH=str(ROOT / "open-msspe-design_amd"), MSSPE_BENCH_BACKEND="gloo", MSSPE_BENCH_DEVICE="0")
    mod = ["-m", "msspe_amd.pipeline_ranks", "-i", str(fa), "-o", str(got_csv)] + flags
    if world == 1:
        cmd = [sys.executable] + mod
    else:
        port = 29700 + os.getpid() % 200
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + mod
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert got_csv.read_text() == want_csv.read_text()
    assert want_csv.read_text().count("\n") > 5
