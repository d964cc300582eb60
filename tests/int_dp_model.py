"""Plain-Python restatement of the exact-integer recurrence of csrc/thal_pairs_int.hip (test
infrastructure: lets the CPU suite check the integer tables and the recurrence against the oracle
without a GPU).  Mirrors the kernel's indices one to one: cell_bases / visit_geometry /
visit_finish / the maxTM cross-multiplication / the deferral reasons."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

K_MAXSZ = 28
K_NB = 0
K_BU = K_NB + (K_MAXSZ - 1) * 64
K_BUSTRIDE = (K_MAXSZ + 1) * 4
K_TSC = K_BU + 4 * K_BUSTRIDE
K_MMC = K_TSC + 64
K_ZERO = K_MMC + 64
K_ZT = K_ZERO + 4
K_ENDL = K_ZT + 64
K_ENDR = K_ENDL + 100
K_WC = K_ENDR + 100
K_COUNT = K_WC + 16
K_ROWS = 14 * 17 + 1
K_VALID = 500000000

DEFER_TM, DEFER_LOOP_EQ, DEFER_LOOP_TIE, DEFER_BAD, DEFER_PICK = 1, 2, 4, 8, 16


@dataclass
class PairTables:
    S: np.ndarray
    H: np.ndarray
    g: np.ndarray
    T: np.ndarray
    init_S: float
    RC: float
    salt: float
    temp_k: float
    g_cut: float
    fast_ok: bool
    int_ok: bool


def load_tables(msspe_amd, threshold: float = -9000.0) -> PairTables:
    L = msspe_amd.capi.load_library()
    chem = msspe_amd.Chem.ntthal()
    S = np.zeros(K_COUNT)
    H = np.zeros(K_COUNT, dtype=np.int32)
    g = np.zeros(K_COUNT, dtype=np.int32)
    T = np.zeros(K_ROWS * 64, dtype=np.int32)
    consts = (C.c_double * 8)()
    L.msspe_host_pair_tables.argtypes = [C.c_char_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    rc = L.msspe_host_pair_tables(None, C.byref(chem), C.c_float(threshold), S.ctypes.data, H.ctypes.data,
                                  g.ctypes.data, T.ctypes.data, consts)
    assert rc == 0 and int(consts[7]) == K_COUNT
    return PairTables(S, H, g, T, consts[0], consts[1], consts[2], consts[3], consts[4],
                      consts[5] == 1.0, consts[6] == 1.0)


CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def cell_bases(s1, s2, im1, jm1):
    n = len(s1)
    a = s1[im1]
    oaL = s1[im1 - 1] if im1 > 0 else 4
    oaR = s1[im1 + 1] if im1 < n - 1 else 4
    obL = s2[jm1 - 1] if jm1 > 0 else 4
    obR = s2[jm1 + 1] if jm1 < n - 1 else 4
    ci = (((3 - a) * 4 + (obL & 3)) * 4 + (oaL & 3)) & 63
    return dict(a=a, idxL=K_ENDL + a * 25 + oaL * 5 + obL, idxR=K_ENDR + a * 25 + oaR * 5 + obR,
                wc=K_WC + (oaL & 3) * 4 + a, po=a | ((oaR & 3) << 2) | ((obR & 3) << 4),
                yTS=K_TSC + ci, yMM=K_MMC + ci, bBase=K_BU + a * K_BUSTRIDE)


def entropy_of(G, H):
    return (H * 2000.0 - G) * (1.0 / 620300.0)


def run_pair(tb: PairTables, a: str, b: str):
    """Returns (cells, defer) with cells = {(im1, jm1): (G, H, pred)} in the kernel's slot order."""
    s1 = [CODE[c] for c in a]
    s2 = [CODE[c] for c in reversed(b)]
    k = len(a)
    cells = {}
    order = []
    defer = 0
    for im1 in range(k):
        for jm1 in range(k):
            if s1[im1] + s2[jm1] != 3:
                continue
            cb = cell_bases(s1, s2, im1, jm1)
            cgeo = (im1 - 1) * 16 + (jm1 - 1)
            jm1p = jm1 - 1
            a4 = cb["a"] << 2
            yTS, yMM = int(tb.g[cb["yTS"]]), int(tb.g[cb["yMM"]])
            bestG, bestW, tie = K_VALID, None, False
            stk = None
            for (pi, pj) in order:
                Gp, Hp, _, po = cells[(pi, pj)]
                d = cgeo - (pi * 16 + pj)
                geo = pj <= jm1p and d >= 0
                if not geo:
                    continue
                if d == 0:
                    stk = (Gp, Hp)
                bulge = d < 16 or (d & 15) == 0
                pe = ((po & 3) | a4) if bulge else po
                idx = min(d * 64 + pe, K_ROWS * 64 - 1)
                y = yMM if d == 0x11 else (0 if bulge else yTS)
                cand = int(tb.T[idx]) + y + Gp
                if cand < bestG:
                    bestG, bestW, tie = cand, (pi, pj, po, Hp), False
                elif cand == bestG:
                    tie = True
            H0, G0, pred, flags = int(tb.H[cb["idxL"]]), int(tb.g[cb["idxL"]]), None, 0
            if stk is not None:
                rS, rH = float(tb.S[cb["idxR"]]), int(tb.H[cb["idxR"]])
                H1 = stk[1] + int(tb.H[cb["wc"]])
                G1 = stk[0] + int(tb.g[cb["wc"]])
                A0, A1 = float(H0 + 200 + rH), float(H1 + 200 + rH)
                B0 = ((entropy_of(G0, H0) + tb.init_S) + rS) + tb.RC
                B1 = ((entropy_of(G1, H1) + tb.init_S) + rS) + tb.RC
                lhs, rhs = A1 * B0, A0 * B1
                sure = B0 < 0 and B1 < 0 and abs(lhs - rhs) > 1e-9 * (abs(lhs) + abs(rhs))
                if not sure:
                    flags |= DEFER_TM
                if lhs > rhs:
                    H0, G0, pred = H1, G1, (im1 - 1, jm1 - 1)
            if bestG == G0:
                flags |= DEFER_LOOP_EQ
            if bestG < G0 and tie:
                flags |= DEFER_LOOP_TIE
            if bestG < G0:
                pi, pj, po, Hp = bestW
                l1, l2 = im1 - 1 - pi, jm1 - 1 - pj
                sz, t = l1 + l2, min(l1, l2)
                bulge = t == 0
                lx = (sz * 4 + (po & 3) + cb["bBase"]) if bulge else (sz * 64 + po + (K_NB - 2 * 64))
                lx = min(lx, K_COUNT - 1)
                yidx = K_ZERO if bulge else (cb["yMM"] if (l1, l2) == (1, 1) else cb["yTS"])
                Hw = int(tb.H[lx]) + int(tb.H[yidx]) + Hp
                if Hw > 0 and entropy_of(bestG, Hw) > -1e-6:
                    flags |= DEFER_BAD
                H0, G0, pred = Hw, bestG, (pi, pj)
            defer |= flags
            cells[(im1, jm1)] = (G0, H0, pred, cb["po"])
            order.append((im1, jm1))
    # terminal pick
    pickG, pick, ptie = None, None, False
    for (im1, jm1) in order:
        G0 = cells[(im1, jm1)][0]
        Gt = G0 + int(tb.g[cell_bases(s1, s2, im1, jm1)["idxR"]])
        if pickG is None or Gt < pickG:
            pickG, pick, ptie = Gt, (im1, jm1), False
        elif Gt == pickG:
            ptie = True
    if ptie:
        defer |= DEFER_PICK
    return cells, defer, pick
