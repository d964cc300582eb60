"""A few fixed-seed cases of each randomised differential campaign under tools/ (pools / alignments / oligo sets /
blocks / whole CLI runs of random shape and parameters against the oracle): the long runs are one-off,
these keep the generators and a sample of what they cover in the suite.  In-process (one GPU process)."""
import runpy
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def run_tool(monkeypatch, name, *args):
    monkeypatch.setattr(sys, "argv", [name, *[str(a) for a in args]])
    with pytest.raises(SystemExit) as e:
        runpy.run_path(str(ROOT / "tools" / name), run_name="__main__")
    assert e.value.code == 0, f"{name} {args}: a case differs from the oracle (see the captured output)"


@pytest.mark.parametrize("tool,args", [
    ("random_campaign.py", (77, 3)),            # case 0 of this seed: a skewed pool with waves without any cell
    ("random_campaign.py", (5, 2, 17, 26)),
    ("random_campaign_stage_a.py", (11, 16)),
    ("random_campaign_stage_a_big.py", (5, 14)),       # the multi-winner loop's regime: hundreds of rows, duplicated blocks, clades
    ("random_campaign_stage_a_big.py", (5, 40, 35)),   # case 35: two winners of one iteration share a live segment (claimed once)
    ("random_campaign_stage_b.py", (3, 12)),
    ("random_campaign_blocks.py", (8, 14)),
    ("random_campaign_cli.py", (21, 8)),
])
def test_campaign_sample(monkeypatch, tool, args):
    run_tool(monkeypatch, tool, *args)
