"""Experiment: per-iteration cost of k_nb_tmm (all frames forced to run maxIT iterations)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_ldpc_amd import nbldpc as nb
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nbd = os.path.join(ROOT, "data", "nb")
mul, _, _ = nb.GFInitial(64, os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
code = nb.NBCode(os.path.join(nbd, "BDS.576.288.GF.64.txt"), mul)
cw = np.loadtxt(os.path.join(nbd, "codeword_bds_gf64.txt"), dtype=np.int32)
seed = np.array([173, 173, 173], np.int32)
sigma = nb.sigma_of(-2.0, code.rate)
B = 16384
rx = nb.AWGNChannel_GPU(seed, sigma, code, torch.from_numpy(cw).cuda(), B)
Lch = nb.Demodulate(code, rx, sigma)
for layered in (False, True):
    for it in (2, 6):
        nb.Decoding_TMM(code, Lch, it, layered=layered)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = nb.Decoding_TMM(code, Lch, it, layered=layered)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("layered %d maxIT %2d: %.3f ms = %.1f us per frame per CU (mean it %.2f)" % (layered, it, dt * 1e3, dt * 1e6 / (B / 256), float(r["iter_number"].float().mean())))
