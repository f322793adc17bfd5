"""Fixed cost of the fused kernels per workgroup: kernel time at 1, 2, 5, 10, 50 iterations (HIP events around the kernel).
usage: python tools/prologue_time.py   (GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_ldpc_amd as C  # noqa: E402

for fn, J, L, Z, F, snr in (("J4_L24_Z96_BlockH.txt", 4, 24, 96, 65536, 3.0), ("J32_L64_Z64_BlockH.txt", 32, 64, 64, 65536, 0.0),
                            ("J15_L30_Z1280_BlockH.txt", 15, 30, 1280, 8192, 0.0)):
    code = C.BinaryCode.from_blockh(os.path.join(ROOT, "data", "bldpc", fn), J, L, Z)
    code.set_profiling(True)
    seed = np.array([173, 173, 173], np.int32)
    y = C.AWGNChannel_GPU(seed, C.sigma_of(snr), code.N, F)
    row = []
    for it in (1, 2, 5, 10, 50):
        ms = []
        for _ in range(4):
            C.LDPC_Decoder_GPU(code, y, max_iter=it, exit_mode=C.EXIT_FIXED)
            torch.cuda.synchronize()
            ms.append(code.last_kernel_ms())
        row.append("%d it %.3f ms" % (it, min(ms)))
    ms = []
    for _ in range(4):
        C.LDPC_Decoder_GPU(code, y, max_iter=50, exit_mode=C.EXIT_PER_FRAME)
        torch.cuda.synchronize()
        ms.append(code.last_kernel_ms())
    print("%-26s F=%d  %s | per-frame kernel %.3f ms" % (fn, F, ", ".join(row), min(ms)), flush=True)
