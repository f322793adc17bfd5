"""Throughput of the per-frame exit (bldpc_decode_per_frame) against fixed iterations on the benchmark codes.
usage: python tools/per_frame_time.py   (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_ldpc_amd as C  # noqa: E402

CASES = [("J4_L24_Z96_BlockH.txt", 4, 24, 96, 4096, (3.0, 3.6, 4.2)), ("J4_L24_Z96_BlockH.txt", 4, 24, 96, 65536, (3.0, 3.6, 4.2)), ("J32_L64_Z64_BlockH.txt", 32, 64, 64, 65536, (0.0, -0.6)),
         ("J15_L30_Z1280_BlockH.txt", 15, 30, 1280, 8192, (0.0, -1.2)), ("PON_LDPC.txt", 12, 69, 256, 8192, (2.6,)),
         ("J10_L60_Z160_BlockH.txt", 10, 60, 160, 16384, (2.9, 3.2)), ("J4_L24_Z512_BlockH.txt", 4, 24, 512, 8192, (3.0,))]
if os.environ.get("PFT_CASES"):  # experiments: a subset, e.g. PFT_CASES=1,2
    CASES = [CASES[int(i)] for i in os.environ["PFT_CASES"].split(",")]
for fn, J, L, Z, F, snrs in CASES:
    code = C.BinaryCode.from_blockh(os.path.join(ROOT, "data", "bldpc", fn), J, L, Z)
    for snr in snrs:
        seed = np.array([173, 173, 173], np.int32)
        y = C.AWGNChannel_GPU(seed, C.sigma_of(snr), code.N, F)
        out = {}
        for name, mode in (("fixed", C.EXIT_FIXED), ("batch_global", C.EXIT_BATCH_GLOBAL), ("per_frame", C.EXIT_PER_FRAME)):
            r = C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                r = C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=mode)
            torch.cuda.synchronize()
            out[name] = F * 3 / (time.perf_counter() - t0)
            if name == "batch_global":
                stop = r["iteraTime"]
        it = r["iters"].float()
        print("%-26s Es/N0 %5.1f dB  F=%d  fixed-50 %9.0f cw/s   batch-global (reference rule, stops at %d) %9.0f cw/s   per-frame %10.0f cw/s (x%.1f)  mean iterations %.2f, %.3f%% at 50 [%s]"
              % (fn, snr, F, out["fixed"], stop, out["batch_global"], out["per_frame"], out["per_frame"] / out["fixed"], it.mean().item(),
                 100.0 * (r["iters"] == 50).float().mean().item(), code.last_kernel), flush=True)
