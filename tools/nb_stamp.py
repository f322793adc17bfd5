"""Experiment (NB_STAMP build): cycles per phase of k_nb_ems for wave 0 of frame 0, one frame per CU resident."""
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
for B in (1, 256, 4096):
    rxt = torch.from_numpy(rx).cuda().repeat(max(1, B // 256), 1)[:B].contiguous()
    Lch = nb.Demodulate(code, rxt, sigma)
    its = 16
    r = nb.Decoding_EMS(code, Lch, 2, 2, its, want_state=True)
    torch.cuda.synchronize()
    t = r["L_c2v"].view(-1)[:12].view(torch.int64).cpu().numpy()
    names = ["A", "S", "B sort", "C checks", "barrier waits", "loop top"]
    print("B=%d: cycles per iteration (100 MHz s_memtime ticks x?):" % B, {n: int(v) // its for n, v in zip(names, t)})
