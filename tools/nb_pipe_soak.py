"""Soak of k_nb_ems2 (two frames in flight per workgroup, counted rendezvous instead of barriers): random batch sizes, noise levels
and iteration limits, every call compared bit for bit (symbols, iteration counts, flags, final LLRs) with k_nb_ems on the same
inputs and with a second call of itself (a race in the slot words, the counts or the column map would not reproduce).
usage: python tools/nb_pipe_soak.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuda_ldpc_amd import nbldpc as nb  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nbd = os.path.join(ROOT, "data", "nb")
mul, _, _ = nb.GFInitial(64, os.path.join(nbd, "GF", "Arith.Table.GF.64.txt"))
code = nb.NBCode(os.path.join(nbd, "BDS.576.288.GF.64.txt"), mul)
os.environ["NBLDPC_NO_PIPE"] = "1"  # read once, when a code object is created
plain = nb.NBCode(os.path.join(nbd, "BDS.576.288.GF.64.txt"), mul)
del os.environ["NBLDPC_NO_PIPE"]
cw = torch.from_numpy(np.loadtxt(os.path.join(nbd, "codeword_bds_gf64.txt"), dtype=np.int32)).cuda()
rng = np.random.default_rng(20261005)
t0, n, bad, frames = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    B = int(rng.choice([1, 2, 3, 5, 17, 255, 256, 257, 511, 1023, 2049, int(rng.integers(1, 20000))]))
    snr = float(rng.uniform(1.0, 5.0))
    maxit = int(rng.choice([1, 2, 3, 7, 20, 20, 20]))
    seed = rng.integers(1, 60000, 3).astype(np.int32)
    sigma = nb.sigma_of(snr, code.rate)
    Lch = nb.Demodulate(code, nb.AWGNChannel_GPU(seed, sigma, code, cw, B), sigma)
    if rng.random() < 0.2:
        Lch = torch.round(Lch)  # ties everywhere: the sort's repeat path, first-maximum decisions
    a = nb.Decoding_EMS(code, Lch, 2, 2, maxit, want_state="llr")
    b = nb.Decoding_EMS(plain, Lch, 2, 2, maxit, want_state="llr")
    c = nb.Decoding_EMS(code, Lch, 2, 2, maxit, want_state="llr")
    torch.cuda.synchronize()
    ok = all(torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]) for k in ("DecodeOutput", "iter_number", "ok"))
    ok = ok and torch.equal(a["LLR"].view(torch.int32), b["LLR"].view(torch.int32)) and torch.equal(a["LLR"].view(torch.int32), c["LLR"].view(torch.int32))
    n += 1
    frames += B
    if not ok:
        bad += 1
        print("MISMATCH B=%d Eb/N0 %.2f maxIT %d seed %s" % (B, snr, maxit, seed.tolist()), flush=True)
    if n % 25 == 0:
        print("%d calls, %d frames, %.0f s, %d mismatches" % (n, frames, time.time() - t0, bad), flush=True)
print("nb pipe soak: %d calls, %d frames, %s" % (n, frames, "OK" if bad == 0 else "%d MISMATCHES" % bad))
sys.exit(1 if bad else 0)
