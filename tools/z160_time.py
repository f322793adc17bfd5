"""Experiment: throughput of the Z = 160 family (compressed-state kernel) at 50 fixed iterations."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_ldpc_amd as C
F = 16384
for J in (10, 12, 15, 20, 24, 30, 36, 40, 48):
    p = os.path.join(ROOT, "data", "bldpc", "J%d_L60_Z160_BlockH.txt" % J)
    code = C.BinaryCode.from_blockh(p, J, 60, 160)
    y = torch.randn((code.N, F), device="cuda") * 0.5 + 1.0
    C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_FIXED)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_FIXED)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    E = code.nnz * 160
    print("J%d: %.0f codewords/s, %.2f T edge-iterations/s, %s" % (J, F / dt, F / dt * E * 50 / 1e12, code.last_kernel[:40]))
