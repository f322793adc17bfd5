"""Time k_nb_ems_hbm on the reference's two heavy-row codes (frames/s at a given Eb/N0, channel samples from the device generator)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_ldpc_amd import nbldpc as nb
NB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "nb")
for mat, tab, q, B, snr in (("Tanner_74_9_Z128_GF16.txt", "Arith.Table.GF.16.txt", 16, 256, 5.0), ("LDPC_N576_K480_GF256_exp.txt", "Arith.Table.GF.256.txt", 256, 1024, 5.0)):
    mul, _, _ = nb.GFInitial(q, os.path.join(NB, "GF", tab))
    code = nb.NBCode(os.path.join(NB, mat), mul)
    sigma = nb.sigma_of(snr, code.rate)
    cw = torch.zeros(code.N, dtype=torch.int32, device="cuda")
    seed = np.array([173, 173, 173], np.int32)
    rx = nb.AWGNChannel_GPU(seed, sigma, code, cw, B)
    Lch = nb.Demodulate(code, rx, sigma)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        r = nb.Decoding_EMS(code, Lch, 2, 2, 20)
        torch.cuda.synchronize(); dt = time.time() - t0
    it = r["iter_number"].float().mean().item(); okf = r["ok"].float().mean().item()
    print("%s: B=%d Eb/N0 %.1f dB: %.3f s -> %.1f frames/s, mean iterations %.2f, converged %.3f" % (mat, B, snr, dt, B / dt, it, okf), flush=True)
