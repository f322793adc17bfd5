"""A second, independent restatement of the binary decoder (numpy, vectorised over circulants and frames) against the C oracle.

The binary path is PARITY UNPINNED: the reference's arithmetic lives in CUDA kernels that cannot be built here, so
nothing below is a comparison with the reference itself.  What it does establish is that two restatements written separately
from SURVEY Appendix A / bldpc_实习/LDPC_Decoder.cu:172-398 -- oracle/bldpc_oracle.c (thread-per-node loops, the literal
two-pass sortQ) and this file (block-wise gathers, sort + argmax) -- agree bit for bit on hard bits AND a-posteriori sums after
1, 2, 3 and more iterations, for every matrix family the GPU tests use, including the long block and the reference's default
PON matrix, for which no reference-derived number exists at all.
"""
import os

import numpy as np
import pytest

from conftest import DATA

BL = os.path.join(DATA, "bldpc")


def np_minsum(H, Z, y, iters):
    """Flooding un-normalised min-sum on the intended circulants (row = (c - s) mod Z), fp32, in the reference's
    operation order: S = (((0 + R_0) + R_1) + ...) + y top to bottom (A.2); R_i = (float)(P * sg_i) * (i == first argmin ? min2 : min1)
    (A.3); the check-node pass after the last variable-node pass is not observable and not run.  Returns (D, S)."""
    J, L = H.shape
    F = y.shape[1]
    wc = (H != -1).sum(1)
    Wc = int(wc.max())
    RQ = np.zeros((J * Z * Wc, F), np.float32)
    pos = np.cumsum(H != -1, axis=1) - 1  # ordinal of a block inside its block row
    c = np.arange(Z)
    slots = {}  # (j, l) -> slot of every column position c
    for j in range(J):
        for l in range(L):
            if H[j, l] != -1:
                slots[(j, l)] = (j * Z + (c - H[j, l]) % Z) * Wc + pos[j, l]
    S_all = np.zeros((L * Z, F), np.float32)
    for it in range(1, iters + 1):
        for l in range(L):
            blocks = [slots[(j, l)] for j in range(J) if (j, l) in slots]
            R = [RQ[s] for s in blocks]
            S = np.zeros((Z, F), np.float32)
            for r in R:
                S = S + r
            S = S + y[l * Z:(l + 1) * Z]
            S_all[l * Z:(l + 1) * Z] = S
            for s, r in zip(blocks, R):
                RQ[s] = S - r
        if it == iters:
            break
        for j in range(J):
            w = int(wc[j])
            base = (j * Z + c) * Wc
            Q = np.stack([RQ[base + i] for i in range(w)])  # [w, Z, F]
            sg = np.where(Q < 0, -1, 1).astype(np.int32)
            a = np.where(Q < 0, -Q, Q)
            P = np.prod(sg, axis=0)
            srt = np.sort(a, axis=0)
            min1, min2 = srt[0], srt[1]
            idx = np.argmax(a == min1[None], axis=0)  # first index holding the minimum
            for i in range(w):
                RQ[base + i] = (P * sg[i]).astype(np.float32) * np.where(idx == i, min2, min1)
    return (S_all < 0).astype(np.int32), S_all


CASES = [  # (file, J, L, Z, F, Es/N0 dB, iteration counts)
    ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 8, 3.0, (1, 2, 3, 12)),
    ("J32_L64_Z64_BlockH.txt", 32, 64, 64, 4, -0.5, (1, 2, 3, 9)),
    ("J10_L60_Z160_BlockH.txt", 10, 60, 160, 2, 2.5, (1, 2, 5)),
    ("J15_L30_Z1280_BlockH.txt", 15, 30, 1280, 2, 0.0, (1, 2, 3, 6)),
    ("PON_LDPC.txt", 12, 69, 256, 2, 2.4, (1, 2, 3, 6)),
]


@pytest.mark.parametrize("name,J,L,Z,F,snr,its", CASES)
def test_numpy_restatement_equals_c_oracle(orc, name, J, L, Z, F, snr, its):
    code = orc.BinaryCode(os.path.join(BL, name), J, L, Z)
    seed = np.array([173, 173, 173], np.int32)
    y = orc.bldpc_awgn(seed, orc.bldpc_sigma(snr), code.N, F)
    H = code.H.reshape(J, L)
    for it in its:
        want = orc.bldpc_decode(code, y, F, it, early_exit=0, want_app=True)
        D, S = np_minsum(H, Z, y.reshape(code.N, F), it)
        assert np.array_equal(D.reshape(-1), want["D"][: code.N * F]), "hard bits, %d iterations" % it
        assert np.array_equal(S.reshape(-1).view(np.uint32), want["app"].view(np.uint32)), "a-posteriori bits, %d iterations" % it
