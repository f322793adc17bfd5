"""Experiment (NB_STAMP build): cycles per half-step of k_nb_ems2 (workgroup 0): stage 1 work, wait at barrier 1, stage 2 work, wait at
barrier 2, for the first walking wave ("C") and the first A/S/B wave ("AB").  Every frame fails (Eb/N0 -2 dB): all run maxIT iterations."""
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
for B in (2, 512, 4096):
    rxt = torch.from_numpy(rx).cuda().repeat(max(1, B // 256), 1)[:B].contiguous()
    Lch = nb.Demodulate(code, rxt, sigma)
    r = nb.Decoding_EMS(code, Lch, 2, 2, 16, want_state="llr")
    torch.cuda.synchronize()
    t = r["LLR"].view(-1)[:32].view(torch.int64).cpu().numpy()
    for name, o in (("C ", t[0:5]), ("AB", t[8:13])):
        n = max(int(o[4]), 1)
        print("B=%d %s wave: half-steps %d; cycles per half-step: stage 1 %d, wait 1 %d, stage 2 %d, wait 2 (+ loop top) %d" % (B, name, n, o[0] // n, o[1] // n, o[2] // n, o[3] // n))
