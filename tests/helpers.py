"""Shared test helpers (pure Python, small inputs only)."""
from __future__ import annotations


def draw_dimer(oligo1: str, oligo2: str, ps1, ps2) -> list[str]:
    """ntthal's four alignment rows for a duplex (Primer3 2.6.1 thal.c drawDimer, restated).

    ps1[i-1] = partner in the REVERSED oligo 2 (0 = unpaired), ps2 likewise.  Rows are returned
    with ntthal's "SEQ\\t"/"STR\\t" prefixes; the reference's transcript
    (od-msspe/src/delta_g.rs:206-230) shows them tab-expanded.
    """
    o2 = oligo2[::-1]
    len1, len2 = len(oligo1), len(o2)
    d = ["", "", "", ""]
    n1 = 0
    while n1 < len1 and ps1[n1] == 0:
        n1 += 1
    n2 = 0
    while n2 < len2 and ps2[n2] == 0:
        n2 += 1
    if n1 >= n2:
        d[0] += oligo1[:n1]
        d[1] += " " * n1
        d[2] += " " * n1
        d[3] += " " * (n1 - n2) + o2[:n2]
    else:
        d[3] += o2[:n2]
        d[1] += " " * n2
        d[2] += " " * n2
        d[0] += " " * (n2 - n1) + oligo1[:n1]
    i, j = n1 + 1, n2 + 1
    while i <= len1:
        while i <= len1 and ps1[i - 1] != 0 and j <= len2 and ps2[j - 1] != 0:
            d[0] += " "
            d[1] += oligo1[i - 1]
            d[2] += o2[j - 1]
            d[3] += " "
            i += 1
            j += 1
        s1 = 0
        while i <= len1 and ps1[i - 1] == 0:
            d[0] += oligo1[i - 1]
            d[1] += " "
            s1 += 1
            i += 1
        s2 = 0
        while j <= len2 and ps2[j - 1] == 0:
            d[2] += " "
            d[3] += o2[j - 1]
            s2 += 1
            j += 1
        if s1 < s2:
            d[0] += "-" * (s2 - s1)
            d[1] += " " * (s2 - s1)
        elif s1 > s2:
            d[2] += " " * (s1 - s2)
            d[3] += "-" * (s1 - s2)
    return ["SEQ\t" + d[0], "SEQ\t" + d[1], "STR\t" + d[2], "STR\t" + d[3]]


def window_with_kmers(kmers: list[str], width: int) -> str:
    """A search window whose valid k-mers are exactly `kmers` in order: consecutive overlapping
    k-mers are merged, others are separated by '-' (which invalidates every k-mer covering it,
    od-msspe/src/main.rs:167)."""
    out = ""
    for km in kmers:
        if out and out[-(len(km) - 1):] == km[:-1]:
            out += km[-1]
        else:
            out += ("-" if out else "") + km
    assert len(out) <= width, (out, width)
    return out + "-" * (width - len(out))
