"""Kernel time of k_nb_ems on the bench batch (16 384 frames, Eb/N0 3 dB) for maxIT = 1, 2, 3, 5, 10, 20: the slope is the cost of
an iteration, the intercept what a frame costs beyond its iterations.   usage: python tools/nb_fixed_cost.py   (GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_ldpc_amd import nbldpc as nb
NB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "nb")
mul, _, _ = nb.GFInitial(64, os.path.join(NB, "GF", "Arith.Table.GF.64.txt"))
code = nb.NBCode(os.path.join(NB, "BDS.576.288.GF.64.txt"), mul)
B = 16384
cw = torch.from_numpy(np.loadtxt(os.path.join(NB, "codeword_bds_gf64.txt"), dtype=np.int32)).cuda()
seed = np.array([173, 173, 173], np.int32)
sigma = nb.sigma_of(3.0, code.rate)
rx = nb.AWGNChannel_GPU(seed, sigma, code, cw, B)
Lch = nb.Demodulate(code, rx, sigma)
for snr2 in (1.0,):  # every frame fails: 20 iterations each, no spread
    sg = nb.sigma_of(snr2, code.rate)
    s2 = np.array([173, 173, 173], np.int32)
    L2 = nb.Demodulate(code, nb.AWGNChannel_GPU(s2, sg, code, cw, B), sg)
    for maxit in (1, 2, 20):
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = nb.Decoding_EMS(code, L2, 2, 2, maxit)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("Eb/N0 %.1f dB maxIT %2d: %.3f ms, mean iterations run %.2f, converged %.3f" % (snr2, maxit, best, r["iter_number"].float().mean().item(), r["ok"].float().mean().item()), flush=True)
for maxit in (1, 2, 3, 5, 10, 20):
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = nb.Decoding_EMS(code, Lch, 2, 2, maxit)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    it = r["iter_number"].float()
    print("maxIT %2d: %.3f ms, mean iterations run %.2f, converged %.3f" % (maxit, best, it.mean().item(), r["ok"].float().mean().item()), flush=True)
