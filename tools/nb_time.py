"""Experiment: per-frame fixed cost and per-iteration cost of k_nb_ems (all frames forced to run maxIT iterations)."""
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
sigma = nb.sigma_of(-2.0, code.rate)  # hopeless channel: nobody converges
rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, cw) for _ in range(256)])
B = 16384
rxt = torch.from_numpy(rx).cuda().repeat(B // 256, 1).contiguous()
Lch = nb.Demodulate(code, rxt, sigma)
for it in (1, 2, 4, 8, 16):
    nb.Decoding_EMS(code, Lch, 2, 2, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = nb.Decoding_EMS(code, Lch, 2, 2, it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("maxIT %2d: %.3f ms per %d frames = %.1f us per frame per CU (mean it %.2f)" % (it, dt * 1e3, B, dt * 1e6 / (B / 256), float(r["iter_number"].float().mean())))
