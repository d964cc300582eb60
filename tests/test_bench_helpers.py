"""bench.py's host-side helpers that need no GPU: the stage-A roofline imported from the committed rocprofv3 summary
(bytes / SUM of kernel time per direction, labelled as imported)."""
import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stage_a_roofline_comes_from_the_committed_kernel_stats():
    bench = load_bench()
    alg_bytes = 1190000 * (19.0 + 38 * 4.0 + 38 * 8.0 + 38 * 8.0 + 38 * 4.0)
    roof = bench.stage_a_kernel_roofline(alg_bytes)
    assert roof is not None, "profiles/r*_stage_a_kernel_stats.csv is tracked"
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and "imported" in roof["source"]
    # kernel time per direction: below the host wall time of the same workload, above the three sort passes alone
    assert 1.0 < roof["kernel_ms_per_direction"] < 20.0
    assert abs(roof["achieved"] - alg_bytes / (roof["kernel_ms_per_direction"] * 1e-3) / 1e9) < 1e-6
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9


def test_generated_scan_is_what_the_generator_writes():
    """open-msspe-design_amd/csrc/row_scan_pinned.inc (the row kernels' predecessor scan as inline asm) is generated:
    the committed file must be the generator's output, byte for byte."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_row_scan_asm.py")], capture_output=True, text=True, check=True)
    assert out.stdout == (ROOT / "open-msspe-design_amd" / "csrc" / "row_scan_pinned.inc").read_text()
