"""Plain-Python restatement of the long-oligo integer recurrence of csrc/thal_pairs_split.hip (test
infrastructure, no GPU): 5-bit coordinates, the loop term in its two parts L[d] + X[xi]
(csrc/split_tables.hpp), one table for all the lanes of a group."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

MAXSZ = 30
K_NB = 0
K_BU = K_NB + (MAXSZ - 1) * 64
K_BUSTRIDE = (MAXSZ + 1) * 4
K_TSC = K_BU + 4 * K_BUSTRIDE
K_MMC = K_TSC + 64
K_ZERO = K_MMC + 64
K_ZT = K_ZERO + 4
K_ENDL = K_ZT + 64
K_ENDR = K_ENDL + 100
K_WC = K_ENDR + 100
K_COUNT = K_WC + 16
K_XP, K_XMM, K_XB1, K_XB2, K_XCOUNT = 0, 64, 128, 640, 1152
K_BIG, K_VALID = 600000000, 300000000


@dataclass
class SplitTables:
    S: np.ndarray
    H: np.ndarray
    g: np.ndarray
    L: np.ndarray
    X: np.ndarray
    usable: bool
    max_k: int


def load_tables(msspe_amd, params_path=None, max_loop=30) -> SplitTables:
    lib = msspe_amd.capi.load_library()
    chem = msspe_amd.Chem.ntthal()
    chem.max_loop = max_loop
    S = np.zeros(K_COUNT)
    H = np.zeros(K_COUNT, dtype=np.int32)
    g = np.zeros(K_COUNT, dtype=np.int32)
    L = np.zeros(1024, dtype=np.int32)
    X = np.zeros(K_XCOUNT, dtype=np.int32)
    info = (C.c_int32 * 4)()
    lib.msspe_host_split_tables.argtypes = [C.c_char_p, C.c_void_p] + [C.c_void_p] * 5 + [C.POINTER(C.c_int32)]
    rc = lib.msspe_host_split_tables(params_path.encode() if params_path else None, C.byref(chem),
                                     S.ctypes.data, H.ctypes.data, g.ctypes.data, L.ctypes.data,
                                     X.ctypes.data, info)
    assert rc == 0 and info[2] == K_COUNT and info[3] == K_XCOUNT
    return SplitTables(S, H, g, L, X, bool(info[0]), int(info[1]))


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


def run_pair(tb: SplitTables, init_S: float, RC: float, a: str, b: str):
    """Returns ({(im1, jm1): (G, H, po)}, hard) - every complementary cell's exact value in the
    kernel's order, and whether the kernel would hand the pair on for a tie inside the fill
    (Tm near-tie, unequal-enthalpy tie, rejected minimum)."""
    s1 = [CODE[c] for c in a]
    s2 = [CODE[c] for c in reversed(b)]
    k = len(a)
    cells, order, hard = {}, [], False
    for im1 in range(k):
        for jm1 in range(k):
            if s1[im1] + s2[jm1] != 3:
                continue
            cb = cell_bases(s1, s2, im1, jm1)
            cgeo = (im1 - 1) * 32 + (jm1 - 1)
            jm1p = jm1 - 1
            yTS, yMM = int(tb.g[cb["yTS"]]), int(tb.g[cb["yMM"]])
            bestG, winners, stk = K_VALID, [], None
            for (pi, pj) in order:
                Gp, Hp, po = cells[(pi, pj)]
                d = cgeo - (pi * 32 + pj)
                if not (pj <= jm1p and d >= 0):
                    continue
                if d == 0:
                    stk = (Gp, Hp)
                l1z, l2z = d < 32, pj == jm1p
                pe = (po & 3) | (cb["a"] << 2)
                if l1z or l2z:
                    xi = (K_XB1 + d * 16 + pe) if l1z else (K_XB2 + (d >> 5) * 16 + pe)
                    y = 0
                elif d == 0x21:
                    xi, y = K_XMM + po, yMM
                else:
                    xi, y = K_XP + po, yTS
                cand = int(tb.L[d]) + int(tb.X[xi]) + y + Gp
                if cand < bestG:
                    bestG, winners = cand, [(pi, pj, po, Hp)]
                elif cand == bestG:
                    winners.append((pi, pj, po, Hp))
            H0, G0 = int(tb.H[cb["idxL"]]), int(tb.g[cb["idxL"]])
            if stk is not None:
                rS, rH = float(tb.S[cb["idxR"]]), int(tb.H[cb["idxR"]])
                H1, G1 = stk[1] + int(tb.H[cb["wc"]]), stk[0] + int(tb.g[cb["wc"]])
                A0, A1 = float(H0 + 200 + rH), float(H1 + 200 + rH)
                B0 = ((entropy_of(G0, H0) + init_S) + rS) + RC
                B1 = ((entropy_of(G1, H1) + init_S) + rS) + RC
                lhs, rhs = A1 * B0, A0 * B1
                if not (B0 < 0 and B1 < 0 and abs(lhs - rhs) > 1e-9 * (abs(lhs) + abs(rhs))):
                    hard = True
                if lhs > rhs:
                    H0, G0 = H1, G1

            def enthalpy(w):
                pi, pj, po, Hp = w
                l1, l2 = im1 - 1 - pi, jm1 - 1 - pj
                sz = l1 + l2
                if min(l1, l2) == 0:
                    lx, yidx = sz * 4 + (po & 3) + cb["bBase"], K_ZERO
                else:
                    lx = sz * 64 + po + (K_NB - 2 * 64)
                    yidx = cb["yMM"] if (l1, l2) == (1, 1) else cb["yTS"]
                return int(tb.H[lx]) + int(tb.H[yidx]) + Hp

            if bestG < G0:
                hs = {enthalpy(w) for w in winners}
                if len(winners) > 2 or len(hs) > 1:
                    hard = True
                Hw = enthalpy(winners[0])
                if Hw > 0 and 2000 * Hw - bestG > -1000:
                    hard = True
                H0, G0 = Hw, bestG
            elif bestG == G0 and enthalpy(winners[0]) != H0:
                hard = True
            cells[(im1, jm1)] = (G0, H0, cb["po"])
            order.append((im1, jm1))
    return cells, hard
