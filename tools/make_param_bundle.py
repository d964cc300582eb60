#!/usr/bin/env python3
"""Consolidate a Primer3-format thermodynamic parameter directory into one bundle file.

The reference hands `ntthal` a directory of 16 `*.ds` / `*.dh` text tables
(od-msspe/src/delta_g.rs:90,107-108 -> `-path <cwd>/primer3_config/`).  Those tables are
published nearest-neighbour constants (SantaLucia 2004, Bommarito 2000), i.e. data.  The engine
must work on a GPU box where the reference tree does not exist, so this script folds the
directory into a single self-describing text bundle that ships with the package:

    @ <section> <ntokens>
    tok tok tok ...

Sections keep the file order of the source tables (index order is documented in DESIGN.md).
Run (in the build container only):
    python tools/make_param_bundle.py /root/reference/od-msspe/primer3_config \
        open-msspe-design_amd/data/nn_params.bundle
"""
import sys
from pathlib import Path

FOUR_INDEX = ["stack", "stackmm", "tstack2"]          # both .ds and .dh, 256 numbers each
SECTIONS = (
    [(f"{n}.ds", 256) for n in FOUR_INDEX]
    + [(f"{n}.dh", 256) for n in FOUR_INDEX]
    + [("tstack_tm_inf.ds", 256), ("tstack.dh", 256)]
    + [("dangle.ds", 128), ("dangle.dh", 128)]
    + [("loops.ds", 120), ("loops.dh", 120)]           # 30 lines x (size, interior, bulge, hairpin)
    + [("triloop.ds", None), ("triloop.dh", None)]     # key value pairs
    + [("tetraloop.ds", None), ("tetraloop.dh", None)]
)


def main(src: str, dst: str) -> None:
    srcdir = Path(src)
    out = ["# nearest-neighbour parameter bundle (consolidated from a Primer3 config directory)",
           "# generator: tools/make_param_bundle.py ; token 'inf' = not available"]
    for name, want in SECTIONS:
        toks = (srcdir / name).read_text().split()
        if want is not None and len(toks) != want:
            raise SystemExit(f"{name}: expected {want} tokens, found {len(toks)}")
        out.append(f"@ {name} {len(toks)}")
        for i in range(0, len(toks), 16):
            out.append(" ".join(toks[i:i + 16]))
    Path(dst).write_text("\n".join(out) + "\n")
    print(f"wrote {dst}: {len(SECTIONS)} sections")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
