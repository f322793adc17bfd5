"""Experiment (NB_STAMP build): cycles per half-step of k_nb_ems2 (workgroup 0), split at its synchronisation points, for the
first walking wave ("C") and the first A/S/B wave ("AB").  Every frame fails (Eb/N0 -2 dB): all run maxIT iterations."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuda_ldpc_amd import nbldpc as nb
nbd = os.path.join(ROOT, "data", "nb")
mul, _, _ = nb.GFInitial(64, os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
code = nb.NBCode(os.path.join(nbd, "BDS.576.288.GF.64.txt"), mul)
cw = np.loadtxt(os.path.join(nbd, "codeword_bds_gf64.txt"), dtype=np.int32)
seed = np.array([173, 173, 173], np.int32)
sigma = nb.sigma_of(-2.0, code.rate)
rx = np.stack([nb.AWGNChannel_CPU(seed, sigma, code, cw) for _ in range(256)])
for B in (2, 4096):
    rxt = torch.from_numpy(rx).cuda().repeat(max(1, B // 256), 1)[:B].contiguous()
    Lch = nb.Demodulate(code, rxt, sigma)
    r = nb.Decoding_EMS(code, Lch, 2, 2, 16, want_state="llr")
    torch.cuda.synchronize()
    t = r["LLR"].view(-1)[:256].view(torch.int64).cpu().numpy().reshape(16, 8)
    ncw = 3  # walking waves of the BDS code: ceil(48 rows x 4 edges / 64)
    for w in range(16):
        o = t[w]
        n = max(int(o[6]), 1)
        simd = (int(o[7]) >> 4) & 3
        if w < ncw:
            print("B=%d wave %2d (C,  SIMD %d): half-steps %d; cycles per half-step: loop top %d, wait at the barrier %d, walk %d, count-off %d"
                  % (B, w, simd, n, o[0] // n, o[1] // n, o[2] // n, o[3] // n))
        else:
            print("B=%d wave %2d (AB, SIMD %d): wait for the walk %d, stage 1a (pairs out, E in) %d, wait at the barrier %d, stage 1b (A) %d, "
                  "wait for the other sorting waves %d, stage 2 (S, B) %d" % (B, w, simd, o[0] // n, o[1] // n, o[2] // n, o[3] // n, o[4] // n, o[5] // n))
